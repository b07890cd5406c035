"""Winograd F(2x2x2, 3x3x3) k3 convolution in exact fp32 (3d-sis_amd/csrc/conv3d_wino.hip, sis3d_conv3d_k3wino) against
(a) torch's CPU nn.Conv3d arithmetic in float64 (the truth) and float32 (what the oracle / the reference's CPU path computes),
(b) its own restatement oracle.conv3d_winograd, (c) the direct fp32 MFMA kernel it replaces by default -- on the network's layer
shapes, odd grids (partial 2x2x2 blocks), cout that is not a multiple of 32, batched problems and channel-slice outputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _case(cin, cout, dims, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, cin, *dims, generator=g).clamp_(min=0)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    return x, w, b


@pytest.mark.parametrize("cin,cout,dims", [(128, 256, (24, 12, 24)), (128, 128, (24, 12, 24)), (64, 64, (24, 12, 24)),
                                           (32, 32, (48, 24, 48)), (64, 64, (13, 9, 11)), (32, 48, (7, 9, 13)), (96, 80, (6, 12, 18)),
                                           (8, 16, (2, 2, 2)), (32, 20, (17, 5, 9))])
def test_winograd_kernel_vs_float64_and_direct(oracle, cin, cout, dims):
    from sis3d import ops
    x, w, b = _case(cin, cout, dims, 11 * cin + cout)
    ref64 = F.conv3d(x.double(), w.double(), b.double(), padding=1).clamp(min=0)
    ref32 = F.conv3d(x, w, b, padding=1).clamp(min=0)
    pc = ops.PackedConv(w.cuda(), b.cuda())
    xc = ops.to_cl(x.cuda())
    got = ops.conv3d_k3wino([xc], [pc], relu=True)[0].cpu()
    assert got.shape == ref32.shape
    scale = max(1.0, float(ref64.abs().max()))
    e_w = float((got.double() - ref64).abs().max())
    e_d = float((ref32.double() - ref64).abs().max())
    print("[winograd] %d->%d %s: kernel vs float64 %.2e, oneDNN fp32 direct vs float64 %.2e (output scale %.2f)" % (cin, cout, dims, e_w, e_d, scale))
    assert e_w <= 2e-5 * scale                                  # fp32 summation noise, same class as the direct convolution
    assert float((got - ref32).abs().max()) <= TOL              # the path's stated tolerance against the CPU operator
    # its own restatement (same transforms, torch's summation order)
    own = oracle.conv3d_winograd(x, w, b, relu=True)
    assert float((got - own).abs().max()) <= 2e-5 * scale
    # the direct fp32 MFMA kernel it replaces (when that kernel takes the shape: cin % 32 == 0)
    if pc.packed_t16 is not None:
        ops.set_winograd(False)
        try:
            direct = ops.conv3d_k3t16([xc], [pc], relu=True)[0].cpu()
        finally:
            ops.set_winograd(True)
        assert float((got - direct).abs().max()) <= 2e-5 * scale


def test_winograd_batched_slice_and_no_relu():
    from sis3d import ops
    dims = (24, 12, 24)
    x1, w1, b1 = _case(128, 256, dims, 1)
    x2, w2, b2 = _case(128, 256, dims, 2)
    pcs = [ops.PackedConv(w1.cuda(), b1.cuda()), ops.PackedConv(w2.cuda(), b2.cuda())]
    xs = [ops.to_cl(x1.cuda()), ops.to_cl(x2.cuda())]
    both = ops.conv3d_k3wino(xs, pcs, relu=True)
    for x, w, b, o in ((x1, w1, b1, both[0]), (x2, w2, b2, both[1])):
        assert float((o.cpu() - F.conv3d(x, w, b, padding=1).clamp(min=0)).abs().max()) <= TOL
    one = ops.conv3d_k3wino(xs[1:], pcs[1:], relu=True)[0]
    assert torch.equal(one, both[1])                             # batching does not change a problem's arithmetic
    # no ReLU, no bias, output written into a channel slice of a wider tensor (the torch.cat of backbones.py:109)
    x, w, _ = _case(32, 32, (10, 6, 8), 3, bias=False)
    pc = ops.PackedConv(w.cuda(), None)
    wide = ops.new_act(96, (10, 6, 8), torch.device("cuda")).fill_(7.0)
    ops.conv3d_k3wino([ops.to_cl(x.cuda())], [pc], relu=False, outs=[wide], out_coff=32)
    want = F.conv3d(x, w, None, padding=1)
    assert float((wide[:, 32:64].cpu() - want).abs().max()) <= TOL
    assert bool((wide[:, :32] == 7.0).all()) and bool((wide[:, 64:] == 7.0).all())
    # deterministic: same launch twice, bit-identical
    a = ops.conv3d_k3wino(xs[:1], pcs[:1], relu=True)[0]
    assert torch.equal(a, both[0])


def test_winograd_is_the_default_route_and_can_be_switched_off():
    from sis3d import ops
    x, w, b = _case(128, 256, (24, 12, 24), 5)              # rpn_net: a layer sis3d_conv3d_k3wino_prefer takes
    assert ops.lib().sis3d_conv3d_k3wino_prefer(24, 12, 24, 128, 256, 1, 0) == 1
    assert ops.lib().sis3d_conv3d_k3wino_prefer(24, 12, 24, 64, 64, 1, 0) == 0       # too few work items: stays on the direct kernel
    pc = ops.PackedConv(w.cuda(), b.cuda())
    xc = ops.to_cl(x.cuda())
    assert ops.WINOGRAD
    via_default = ops.conv3d(xc, pc, relu=True)
    assert torch.equal(via_default, ops.conv3d_k3wino([xc], [pc], relu=True)[0])
    ops.set_winograd(False)
    try:
        direct = ops.conv3d(xc, pc, relu=True)
    finally:
        ops.set_winograd(True)
    assert not torch.equal(direct, via_default) and float((direct - via_default).abs().max()) <= 1e-5
