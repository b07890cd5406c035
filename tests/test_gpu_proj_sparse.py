"""csrc/proj_sparse.hip: the colour stem Conv3d(128, 64, k=2, s=2) + ReLU (+ the next Bottleneck's conv1) on a back-projected image
volume, computed only where a view sees something -- against the dense table kernel (sis3d_conv3d_chain_projected) and against a
float64 convolution of the materialised volume (lib/nets/network.py:216-239 + lib/nets/backbones.py:187,214)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import ops, synthetic  # noqa: E402


def _case(n_per_view, dims, n_views, kill, seed, bias):
    g = torch.Generator().manual_seed(seed)
    feats, i3d, i2d = synthetic.synth_views(seed, n_views=n_views, n_per_view=n_per_view, dims=dims)
    w = torch.randn(64, 128, 2, 2, 2, generator=g) * 0.03
    b = torch.randn(64, generator=g) * 0.1 if bias else None
    w1 = torch.randn(32, 64, 1, 1, 1, generator=g) * 0.1
    b1 = torch.randn(32, generator=g) * 0.1
    pv = ops.project_views_prepare(feats.cuda(), i3d.cuda(), i2d.cuda(), dims, kill)
    pc = ops.PackedConv(w.cuda(), b.cuda() if bias else None)
    pc1 = ops.PackedConv(w1.cuda(), b1.cuda())
    return pv, pc, pc1, (w, b, w1, b1)


@pytest.mark.parametrize("n_per_view,dims,n_views,kill,bias", [
    (3000, (96, 48, 96), 5, (), False),        # the bench workload's density (3.4 % of the voxels visible)
    (400, (64, 32, 48), 4, (1,), False),       # a killed view: 0 < cnt < nslots -> max(v, 0)
    (0, (32, 16, 24), 3, (), False),           # nothing visible: every row is the constant of a zero input
    (30000, (32, 16, 24), 3, (), True),        # dense visibility (every voxel seen by some view) and a bias
    (37, (18, 10, 14), 2, (), True),           # a list shorter than one workgroup's 64 voxels, odd brick counts
])
def test_sparse_colour_stem_vs_dense_kernel_and_float64(n_per_view, dims, n_views, kill, bias):
    nvox = dims[0] * dims[1] * dims[2]
    pv, pc, pc1, (w, b, w1, b1) = _case(min(n_per_view, nvox), dims, n_views, kill, 11 + n_per_view, bias)
    stage = [dict(pc=pc1, relu=True)]
    ops.set_sparse_projection(False)
    try:
        dm, dst = ops.conv3d_chain(pv, pc, 2, stage, relu=True, want_main=True)
    finally:
        ops.set_sparse_projection(True)
    sm, sst = ops.conv3d_chain(pv, pc, 2, stage, relu=True, want_main=True)
    sm2, sst2 = ops.conv3d_chain(pv, pc, 2, stage, relu=True, want_main=True)
    assert torch.equal(sm, sm2) and torch.equal(sst[0], sst2[0])                       # deterministic (no atomics)
    vol = pv.dense(channels_last=False).double().cpu()                                 # (1,128,X,Y,Z), the oracle-checked view max
    want = torch.relu(torch.nn.functional.conv3d(vol, w.double(), None if b is None else b.double(), stride=2))
    want1 = torch.relu(torch.nn.functional.conv3d(want, w1.double(), b1.double()))
    scale, scale1 = max(1.0, float(want.abs().max())), max(1.0, float(want1.abs().max()))
    e_m, e_s = float((sm.cpu().double() - want).abs().max()), float((sst[0].cpu().double() - want1).abs().max())
    d_m, d_s = float((sm - dm).abs().max()), float((sst[0] - dst[0]).abs().max())
    active = int((vol.abs().amax(1, keepdim=True) > 0).float().view(1, 1, dims[0] // 2, 2, dims[1] // 2, 2, dims[2] // 2, 2).amax((3, 5, 7)).sum())
    print("[proj sparse] %s n_per_view %d: %d of %d output voxels active; vs float64 %.2e / %.2e, vs dense kernel %.2e / %.2e"
          % (dims, n_per_view, active, nvox // 8, e_m, e_s, d_m, d_s))
    assert e_m <= 2e-5 * scale and e_s <= 2e-5 * scale1
    assert d_m <= 1e-5 * scale and d_s <= 1e-5 * scale1


def test_unsupported_channel_counts_fall_to_the_dense_kernel():
    dims = (32, 16, 24)
    feats, i3d, i2d = synthetic.synth_views(3, n_views=2, n_per_view=200, dims=dims)
    pv = ops.project_views_prepare(feats.cuda(), i3d.cuda(), i2d.cuda(), dims, ())
    pc = ops.PackedConv(torch.randn(32, 128, 2, 2, 2).cuda() * 0.03, None)             # 32 couts: not the colour stem
    assert ops._conv3d_chain_projected_sparse(pv, pc, [], True, True) is None
    pc64 = ops.PackedConv(torch.randn(64, 128, 2, 2, 2).cuda() * 0.03, None)
    wide = ops.PackedConv(torch.randn(48, 64, 1, 1, 1).cuda() * 0.1, torch.zeros(48).cuda())        # a 48-wide stage: not conv1 of Bottleneck(64,32)
    assert ops._conv3d_chain_projected_sparse(pv, pc64, [dict(pc=wide, relu=True)], True, True) is None
    main, outs = ops._conv3d_chain_projected_sparse(pv, pc64, [], True, True)                       # no stage at all is served (c2 = 0)
    assert tuple(main.shape) == (1, 64, 16, 8, 12) and outs == []
