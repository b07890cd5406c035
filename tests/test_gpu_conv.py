"""fp32 parity of the HIP conv / pool kernels against torch-CPU conv3d (the oracle's operators).
Tolerance: 1e-4 absolute on O(1) activations (BASELINE.json north_star), reported tighter."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from sis3d import ops as o
    o.lib()
    return o


def cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last_3d)


def _w(cout, cin, k, g):
    fan = cin * k ** 3
    return (torch.rand(cout, cin, k, k, k, generator=g) * 2 - 1) / fan ** 0.5


CASES = [  # (cin, cout, k, dims)  -- every (shape class, tiling) the dispatcher can pick
    (32, 32, 1, (48, 24, 48)), (32, 32, 3, (48, 24, 48)), (32, 128, 2, (48, 24, 48)),
    (128, 32, 1, (24, 12, 24)), (32, 128, 1, (24, 12, 24)), (32, 32, 3, (24, 12, 24)),
    (128, 128, 3, (24, 12, 24)), (64, 64, 3, (24, 12, 24)), (128, 256, 3, (24, 12, 24)),
    (128, 64, 2, (24, 12, 16)), (64, 64, 2, (20, 12, 16)), (64, 64, 3, (13, 20, 11)), (64, 19, 1, (13, 20, 11)),
    (8, 64, 3, (9, 7, 10)), (64, 64, 3, (5, 3, 2)), (32, 32, 3, (50, 26, 46)),
]


@pytest.mark.parametrize("cin,cout,k,dims", CASES)
def test_conv_vs_torch_cpu(ops, cin, cout, k, dims):
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + k)
    x = torch.randn(1, cin, *dims, generator=g)
    w = _w(cout, cin, k, g)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(1, cout, *(d // (2 if k == 2 else 1) for d in dims), generator=g)
    stride, pad = (2, 0) if k == 2 else (1, k // 2)
    want = F.conv3d(x, w, b, stride=stride, padding=pad)
    pc = ops.PackedConv(w.cuda(), b.cuda())
    got = ops.conv3d(cl(x), pc, stride=stride)
    assert got.shape == want.shape
    err = (got.cpu() - want).abs().max().item()
    assert err <= TOL, err
    # fused epilogue: residual + ReLU
    got2 = ops.conv3d(cl(x), pc, stride=stride, relu=True, residual=cl(res))
    err2 = (got2.cpu() - F.relu(want + res)).abs().max().item()
    assert err2 <= TOL, err2
    # no-bias + sigmoid
    pc3 = ops.PackedConv(w.cuda(), None)
    got3 = ops.conv3d(cl(x), pc3, stride=stride, sigmoid=True)
    err3 = (got3.cpu() - torch.sigmoid(F.conv3d(x, w, None, stride=stride, padding=pad))).abs().max().item()
    assert err3 <= TOL, err3


def test_conv_channel_offset_concat(ops):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 32, 24, 12, 24, generator=g)
    w = _w(64, 32, 1, g)
    pc = ops.PackedConv(w.cuda(), None)
    out = ops.new_act(128, (24, 12, 24), torch.device("cuda"))
    out.fill_(-7.0)
    ops.conv3d(cl(x), pc, out=out, out_coff=64)
    want = F.conv3d(x, w)
    assert (out[:, 64:].cpu() - want).abs().max() <= TOL
    assert (out[:, :64] == -7.0).all()


@pytest.mark.parametrize("A", [3, 11])
def test_rpn_head_layout(ops, A):
    """fused cls+bbox 1x1x1 convs write (1,2,X,Y,Z,A) and (1,X,Y,Z,6A) exactly as network.py:541-543 permutes them"""
    g = torch.Generator().manual_seed(A)
    dims = (24, 12, 24)
    x = torch.randn(1, 256, *dims, generator=g)
    wc, bc = _w(2 * A, 256, 1, g), torch.randn(2 * A, generator=g)
    wb, bb = _w(6 * A, 256, 1, g), torch.randn(6 * A, generator=g)
    pc = ops.PackedConv(torch.cat([wc, wb]).cuda(), torch.cat([bc, bb]).cuda())
    score, bbox, prob = ops.conv3d(cl(x), pc, rpn_anchors=A)
    want_b = F.conv3d(x, wb, bb).permute(0, 2, 3, 4, 1).contiguous()
    want_s = F.conv3d(x, wc, bc).view(1, 2, A, *dims).permute(0, 1, 3, 4, 5, 2).contiguous()
    assert score.shape == want_s.shape and bbox.shape == want_b.shape
    assert score.is_contiguous() and bbox.is_contiguous()
    assert (score.cpu() - want_s).abs().max() <= TOL
    assert (bbox.cpu() - want_b).abs().max() <= TOL
    assert (prob.cpu() - torch.softmax(want_s, 1)).abs().max() <= 1e-5          # fused 2-way softmax (network.py:546)
    assert torch.equal(prob, ops.softmax2(score))                               # bitwise the standalone kernel


@pytest.mark.parametrize("k,cout,dims,window", [(2, 32, (96, 48, 96), None), (3, 64, (96, 48, 96), (10, 5, 20, 22, 25, 33)),
                                               (3, 64, (20, 10, 12), (0, 0, 0, 20, 10, 12)), (2, 64, (16, 8, 12), None)])
def test_planar2_first_layers(ops, k, cout, dims, window):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(1, 2, *dims, generator=g)
    w = _w(cout, 2, k, g)
    if k == 2:
        want = F.relu(F.conv3d(x, w, None, stride=2))
    else:
        x0, y0, z0, x1, y1, z1 = window
        want = F.relu(F.conv3d(x[:, :, x0:x1, y0:y1, z0:z1], w, None, padding=1))    # zero padding at the CROP border
    got = ops.conv3d_planar2(x.cuda(), w.cuda(), k, relu=True, window=window)
    assert got.shape == want.shape
    assert (got.cpu() - want).abs().max() <= 1e-5


def test_maxpool3(ops):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 128, 24, 12, 24, generator=g)
    got = ops.maxpool3(cl(x))
    assert torch.equal(got.cpu(), F.max_pool3d(x, 3, 1, 1))
    x2 = torch.randn(1, 64, 5, 1, 3, generator=g)
    assert torch.equal(ops.maxpool3(cl(x2)).cpu(), F.max_pool3d(x2, 3, 1, 1))


@pytest.mark.parametrize("dims,c", [((24, 12, 24), 128), ((13, 9, 11), 64), ((7, 20, 31), 32), ((48, 24, 48), 64)])
def test_maxpool3_separable_lds_form(ops, dims, c):
    """r4: the brick / LDS form (csrc/pool_misc.hip maxpool3_lds_kernel) on grids that are not multiples of its 6^3 brick, into a channel
    range of a wider tensor, incl. -inf / equal values: bit-identical to nn.MaxPool3d(3,1,1) (backbones.py:206,210,220) and to the
    tap-by-tap kernel"""
    g = torch.Generator().manual_seed(dims[0] + c)
    x = torch.randn(1, c, *dims, generator=g)
    x[0, :, 0, 0, 0] = float("-inf")
    x[0, 1] = 0.25
    want = F.max_pool3d(x, 3, 1, 1)
    assert torch.equal(ops.maxpool3(cl(x)).cpu(), want)
    wide = torch.full((1, c + 32, *dims), -7.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    assert ops.maxpool3(cl(x), wide, 16) is wide
    assert torch.equal(wide[:, 16:16 + c].cpu(), want)
    rest = torch.cat([wide[:, :16], wide[:, 16 + c:]], 1)
    assert float(rest.min()) == -7.0 and float(rest.max()) == -7.0


def test_layout_helpers(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 19, 13, 20, 11, generator=g)
    c = ops.to_cl(x.cuda())
    assert ops.is_cl(c) and torch.equal(c.cpu(), x)
    p = ops.to_planar(c)
    assert p.is_contiguous() and torch.equal(p.cpu(), x)


def test_batched_same_shape_convs(ops):
    g = torch.Generator().manual_seed(9)
    dims = (24, 12, 24)
    xs = [torch.randn(1, 128, *dims, generator=g) for _ in range(2)]
    ws = [_w(256, 128, 3, g) for _ in range(2)]
    bs = [torch.randn(256, generator=g) * 0.1 for _ in range(2)]
    pcs = [ops.PackedConv(w.cuda(), b.cuda()) for w, b in zip(ws, bs)]
    outs = ops.conv3d_batched([cl(x) for x in xs], pcs, relu=True)
    for x, w, b, o in zip(xs, ws, bs, outs):
        want = F.relu(F.conv3d(x, w, b, padding=1))
        assert (o.cpu() - want).abs().max() <= TOL
        # identical to the single launch, bit for bit (same tiling, same summation order)
        single = ops.conv3d(cl(x), ops.PackedConv(w.cuda(), b.cuda()), relu=True)
        assert torch.equal(o, single)


@pytest.mark.parametrize("planes,inplanes,nextp,dims,k", [(32, 32, 32, (48, 24, 48), 3), (32, 128, 32, (24, 12, 24), 3),
                                                         (64, 128, 64, (24, 12, 24), 3), (64, 64, 0, (13, 9, 11), 3),
                                                         (128, 0, 32, (48, 24, 48), 2), (64, 0, 32, (24, 12, 16), 2)])
def test_fused_bottleneck_chain(ops, planes, inplanes, nextp, dims, k):
    """one launch == conv2(k3)+ReLU -> conv3(1x1)+residual+ReLU -> next conv1(1x1)+ReLU  (backbones.py:27-40),
    or a stem conv + the following block's conv1"""
    g = torch.Generator().manual_seed(planes + inplanes + nextp)
    if k == 3:
        y1 = torch.randn(1, planes, *dims, generator=g)
        x = torch.randn(1, inplanes, *dims, generator=g)
        w2, b2 = _w(planes, planes, 3, g), torch.randn(planes, generator=g) * 0.1
        w3, b3 = _w(inplanes, planes, 1, g), torch.randn(inplanes, generator=g) * 0.1
        y2 = F.relu(F.conv3d(y1, w2, b2, padding=1))
        want_x = F.relu(F.conv3d(y2, w3, b3) + x)
        stages = [dict(pc=ops.PackedConv(w3.cuda(), b3.cuda()), relu=True, residual=cl(x), keep=True)]
        want_n = None
        if nextp:
            w1, b1 = _w(nextp, inplanes, 1, g), torch.randn(nextp, generator=g) * 0.1
            want_n = F.relu(F.conv3d(want_x, w1, b1))
            stages.append(dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True))
        main, outs = ops.conv3d_chain(cl(y1), ops.PackedConv(w2.cuda(), b2.cuda()), 1, stages)
        assert main is None
        assert (outs[0].cpu() - want_x).abs().max() <= TOL
        if nextp:
            assert (outs[1].cpu() - want_n).abs().max() <= TOL
    else:
        cin = 32
        xin = torch.randn(1, cin, *dims, generator=g)
        w0 = _w(planes, cin, 2, g)
        w1, b1 = _w(nextp, planes, 1, g), torch.randn(nextp, generator=g) * 0.1
        want0 = F.relu(F.conv3d(xin, w0, None, stride=2))
        want1 = F.relu(F.conv3d(want0, w1, b1))
        main, outs = ops.conv3d_chain(cl(xin), ops.PackedConv(w0.cuda(), None), 2,
                                      [dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True)], want_main=True)
        assert (main.cpu() - want0).abs().max() <= TOL
        assert (outs[0].cpu() - want1).abs().max() <= TOL
