"""Backward kernels (training only): RoI pooling (roi_pooling_kernel.cu:137-248, roi_pool.py:40-50) and
Projection (projection.py:139-153).  Fixtures: roi_pool_backward_case.npz = the reference's own Python RoIPool forward + backward;
projection_backward_case.npz = the reference's own Projection.apply + backward run in place under ref_harness.legacy_data_alias
(oracle/make_golden.py --projection-backward)."""
import numpy as np
import pytest
import torch

from sis3d import synthetic


def test_oracle_roi_pool_backward_matches_reference_fixture(golden, oracle):
    g = golden("roi_pool_backward_case")
    feat, rois = torch.from_numpy(g["feat"]), torch.from_numpy(g["rois"])
    out, arg = oracle.roi_pool(feat, rois, (2, 2, 2), 0.25, want_argmax=True)
    assert np.array_equal(out.numpy(), g["out"])
    gin = oracle.roi_pool_backward(torch.from_numpy(g["grad_out"]), arg, feat.shape)
    assert np.array_equal(gin.numpy(), g["grad_in"])         # same accumulation order as the reference's loop: bit-identical


def _pb_cases(g):
    for name in [str(n) for n in g["names"]]:
        yield (name, tuple(int(v) for v in g[name + "_dims"]), torch.from_numpy(g[name + "_label"]), torch.from_numpy(g[name + "_i3d"]),
               torch.from_numpy(g[name + "_i2d"]), torch.from_numpy(g[name + "_grad_out"]), torch.from_numpy(g[name + "_grad_label"]))


def test_oracle_projection_backward_matches_reference_fixture(golden, oracle):
    """pins the oracle: bit-identical to the reference's own backward (duplicate pixels: last entry wins; untouched pixels keep
    the first C*h*w elements of the cloned volume gradient; empty list)"""
    g = golden("projection_backward_case")
    seen = 0
    for name, dims, label, i3d, i2d, gout, want in _pb_cases(g):
        got = oracle.projection_backward(gout, i3d, i2d)
        assert got.shape == want.shape == label.shape and torch.equal(got, want), name
        seen += 1
    assert seen == 3


def test_oracle_projection_backward_vs_live_reference(oracle):
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install()
    dims = (20, 16, 28)
    feats, i3d, i2d = synthetic.synth_views(21, n_views=1, n_per_view=2000, channels=4, image_hw=(32, 41), dims=dims)
    gout = torch.randn(4, dims[2], dims[1], dims[0], generator=torch.Generator().manual_seed(5))
    fwd, gl = rh.ref_projection_backward(ns, feats[0], i3d[0], i2d[0], dims, gout)
    assert torch.equal(fwd, oracle.projection(feats[0], i3d[0], i2d[0], dims))
    assert torch.equal(gl, oracle.projection_backward(gout, i3d[0], i2d[0]))
    # the shim is gone afterwards: `.data` is a shallow copy again
    x = torch.zeros(3)
    x.data.resize_(1)
    assert x.shape == (3,)


def test_oracle_projection_backward_semantics(oracle):
    dims = (12, 6, 10)
    feats, i3d, i2d = synthetic.synth_views(3, n_views=1, n_per_view=150, channels=5, image_hw=(32, 41), dims=dims)
    g = torch.randn(5, dims[2], dims[1], dims[0], generator=torch.Generator().manual_seed(1))
    gl = oracle.projection_backward(g, i3d[0], i2d[0])
    assert tuple(gl.shape) == (5, 32, 41)
    n = int(i3d[0, 0])
    src = g.reshape(5, -1)
    last = {}
    for k in range(n):
        last[int(i2d[0, 1 + k])] = int(i3d[0, 1 + k])
    for p, v in list(last.items())[:50]:
        assert torch.equal(gl.view(5, -1)[:, p], src[:, v])
    untouched = [p for p in range(32 * 41) if p not in last][:20]
    flat = torch.zeros(5 * 1312)
    flat[:g.numel()] = g.reshape(-1)[:5 * 1312]
    for p in untouched:
        assert torch.equal(gl.view(5, -1)[:, p], flat.view(5, -1)[:, p])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["ncdhw", "channels_last"])
def test_roi_pool_backward_gpu(golden, oracle, layout):
    from sis3d import ops
    from sis3d.layer_utils.roi_pooling.roi_pool import RoIPoolFunction
    g = golden("roi_pool_backward_case")
    feat, rois, gout = torch.from_numpy(g["feat"]), torch.from_numpy(g["rois"]), torch.from_numpy(g["grad_out"])
    f = feat.cuda()
    if layout == "channels_last":
        f = f.contiguous(memory_format=torch.channels_last_3d)
    fn = RoIPoolFunction(2, 2, 2, 0.25)
    out = fn(f, rois.cuda())
    assert np.array_equal(out.cpu().numpy(), g["out"])
    gin, grois = fn.backward(gout.cuda())
    assert tuple(gin.shape) == tuple(feat.shape) and float(grois.abs().sum()) == 0.0
    assert np.array_equal(gin.cpu().numpy(), g["grad_in"])     # r6: deterministic, in the reference's (RoI, bin) order: bit for bit
    # full-size, many overlapping RoIs, against the oracle
    gen = torch.Generator().manual_seed(3)
    big = torch.randn(1, 128, 24, 12, 24, generator=gen)
    lo = torch.rand(60, 3, generator=gen) * torch.tensor([70.0, 30.0, 70.0])
    r = torch.cat([lo, lo + torch.rand(60, 3, generator=gen) * 40.0 + 2.0], 1)
    fb = big.cuda().contiguous(memory_format=torch.channels_last_3d) if layout == "channels_last" else big.cuda()
    o, a = ops.roi_pool(fb, r.cuda(), (4, 4, 4), 0.25, want_argmax=True)
    go = torch.randn(o.shape, generator=gen)
    got = ops.roi_pool_backward(go.cuda(), a, big.shape, channels_last=(layout == "channels_last"))
    want = oracle.roi_pool_backward(go, a.cpu(), big.shape)
    assert torch.equal(got.cpu(), want)                       # same sums in the same order as the oracle's sequential adds
    # the cffi-level entry point accumulates into the caller's zeroed tensor
    from sis3d.dropin import roi_pooling_backward_cuda
    bottom = torch.zeros(1, 128, 24, 12, 24, device="cuda")
    assert roi_pooling_backward_cuda(4, 4, 4, 0.25, go.cuda(), r.cuda(), bottom, a) == 1
    assert torch.equal(bottom.cpu(), want)


@pytest.mark.gpu
def test_projection_backward_gpu_and_autograd(oracle, golden):
    from sis3d import ops
    from sis3d.layer_utils.projection import Projection
    # the reference's own backward (fixture): kernel and autograd.Function, bit for bit
    for name, dims, label, i3d, i2d, gout, want in _pb_cases(golden("projection_backward_case")):
        assert torch.equal(ops.projection_backward(gout.cuda(), i3d.cuda(), i2d.cuda()).cpu(), want), name
        lab = label.cuda().requires_grad_(True)
        Projection.apply(lab, i3d.cuda(), i2d.cuda(), dims).backward(gout.cuda())
        assert torch.equal(lab.grad.cpu(), want), name
    for dims, n in (((12, 6, 10), 150), ((96, 48, 96), 3000)):
        feats, i3d, i2d = synthetic.synth_views(5, n_views=1, n_per_view=n, channels=7, image_hw=(32, 41), dims=dims)
        g = torch.randn(7, dims[2], dims[1], dims[0], generator=torch.Generator().manual_seed(2))
        want = oracle.projection_backward(g, i3d[0], i2d[0])
        got = ops.projection_backward(g.cuda(), i3d[0].cuda(), i2d[0].cuda())
        assert torch.equal(got.cpu(), want)                    # pure copies: bit-exact, last list entry wins on shared pixels
        lab = feats[0].cuda().requires_grad_(True)
        out = Projection.apply(lab, i3d[0].cuda(), i2d[0].cuda(), dims)
        out.backward(g.cuda())
        assert torch.equal(lab.grad.cpu(), want)
