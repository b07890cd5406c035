"""fp32 parity of the balanced k3 kernel (csrc/conv3d_t16.hip: 16x16x4 MFMA tiles, one workgroup per CU, the four waves
split the input channels) and of the pointwise Bottleneck half (sis3d_conv3d_pw_chain) against torch-CPU operators --
the arithmetic the reference's nn.Conv3d calls run (lib/nets/backbones.py:17-40).  Tolerance 1e-4 (north_star)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4
BRICKS = {0: (6, 6, 12), 1: (6, 6, 6), 2: (3, 6, 6), 3: (3, 3, 6), 4: (4, 4, 4), 5: (4, 4, 8), 6: (4, 8, 8)}


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


# every brick instantiation on a grid it tiles exactly, on a grid it does not, and on one smaller than the brick
@pytest.mark.parametrize("brick", sorted(BRICKS))
@pytest.mark.parametrize("dims", [(24, 12, 24), (13, 9, 11), (5, 3, 2)])
def test_every_brick_vs_torch_cpu(ops, brick, dims):
    g = torch.Generator().manual_seed(100 * brick + dims[0])
    cin, cout = 64, 48
    x = torch.randn(1, cin, *dims, generator=g)
    w, b = _w(cout, cin, 3, g), torch.randn(cout, generator=g) * 0.1
    want = F.relu(F.conv3d(x, w, b, padding=1))
    pc = ops.PackedConv(w.cuda(), b.cuda())
    assert pc.packed_t16 is not None
    got = ops.conv3d_k3t16([cl(x)], [pc], relu=True, brick=brick)[0]
    assert got.shape == want.shape and ops.is_cl(got)
    assert (got.cpu() - want).abs().max().item() <= TOL


# the network's own layer shapes through the automatic brick choice (what ops.conv3d now runs for k3)
@pytest.mark.parametrize("cin,cout,dims", [(128, 256, (24, 12, 24)), (128, 128, (24, 12, 24)), (64, 64, (24, 12, 24)),
                                           (32, 32, (24, 12, 24)), (32, 32, (48, 24, 48)), (64, 64, (30, 30, 36)),
                                           (32, 20, (7, 5, 9)), (96, 64, (12, 12, 12))])
def test_layer_shapes_auto_brick(ops, cin, cout, dims):
    g = torch.Generator().manual_seed(cin + cout + dims[2])
    x = torch.randn(1, cin, *dims, generator=g)
    w, b = _w(cout, cin, 3, g), torch.randn(cout, generator=g) * 0.1
    pc = ops.PackedConv(w.cuda(), b.cuda())
    got = ops.conv3d(cl(x), pc)                                   # no ReLU
    assert (got.cpu() - F.conv3d(x, w, b, padding=1)).abs().max().item() <= TOL
    pc0 = ops.PackedConv(w.cuda(), None)                          # no bias, ReLU
    got0 = ops.conv3d(cl(x), pc0, relu=True)
    assert (got0.cpu() - F.relu(F.conv3d(x, w, None, padding=1))).abs().max().item() <= TOL


def test_batched_problems_and_channel_offset(ops):
    g = torch.Generator().manual_seed(5)
    dims = (12, 6, 12)
    xs = [torch.randn(1, 32, *dims, generator=g) for _ in range(3)]
    ws = [_w(64, 32, 3, g) for _ in range(3)]
    bs = [torch.randn(64, generator=g) * 0.1 for _ in range(3)]
    pcs = [ops.PackedConv(w.cuda(), b.cuda()) for w, b in zip(ws, bs)]
    outs = ops.conv3d_batched([cl(x) for x in xs], pcs, relu=True)
    for x, w, b, o in zip(xs, ws, bs, outs):
        assert (o.cpu() - F.relu(F.conv3d(x, w, b, padding=1))).abs().max().item() <= TOL
    # write into a channel range of a wider tensor (torch.cat fusion), neighbours untouched
    wide = ops.new_act(128, dims, torch.device("cuda")).fill_(-3.0)
    ops.conv3d(cl(xs[0]), pcs[0], relu=True, out=wide, out_coff=32)
    assert (wide[:, 32:96].cpu() - F.relu(F.conv3d(xs[0], ws[0], bs[0], padding=1))).abs().max().item() <= TOL
    assert bool((wide[:, :32] == -3.0).all()) and bool((wide[:, 96:] == -3.0).all())


def test_matches_the_32x32_tile_kernel(ops, monkeypatch):
    """the two k3 kernels (conv3d.hip 32x32x2 tiles / conv3d_t16.hip 16x16x4 tiles) differ only in summation order"""
    g = torch.Generator().manual_seed(9)
    x = cl(torch.randn(1, 128, 24, 12, 24, generator=g))
    w, b = _w(128, 128, 3, g).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
    pc = ops.PackedConv(w, b)
    new = ops.conv3d(x, pc, relu=True)
    t16, pc.packed_t16 = pc.packed_t16, None
    old = ops.conv3d(x, pc, relu=True)
    pc.packed_t16 = t16
    assert (new - old).abs().max().item() <= 2e-5


@pytest.mark.parametrize("cin,cout,cnext,dims", [(32, 32, 32, (48, 24, 48)), (32, 128, 32, (24, 12, 24)), (64, 128, 64, (24, 12, 24)),
                                                 (32, 128, None, (24, 12, 24)), (64, 128, 64, (11, 7, 5)), (32, 64, 32, (9, 9, 9))])
def test_pointwise_chain_vs_torch_cpu(ops, cin, cout, cnext, dims):
    """relu(conv3(y2) + b3 + x) and relu(conv1_next(.) + b1) in one launch (backbones.py:33-40 + the next block's :29-31)"""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    y2 = torch.randn(1, cin, *dims, generator=g)
    res = torch.randn(1, cout, *dims, generator=g)
    w3, b3 = _w(cout, cin, 1, g), torch.randn(cout, generator=g) * 0.1
    want = F.relu(F.conv3d(y2, w3, b3) + res)
    pc3 = ops.PackedConv(w3.cuda(), b3.cuda())
    stage = None
    if cnext is not None:
        w1, b1 = _w(cnext, cout, 1, g), torch.randn(cnext, generator=g) * 0.1
        stage = dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True)
    try:
        main, so = ops.conv3d_pw_chain(cl(y2), pc3, residual=cl(res), relu=True, stage=stage)
    except ops.Sis3dUnsupported:
        pytest.skip("no fused pointwise tiling for %d->%d (callers fall back to separate launches)" % (cin, cout))
    assert (main.cpu() - want).abs().max().item() <= TOL
    if cnext is not None:
        assert (so.cpu() - F.relu(F.conv3d(want, w1, b1))).abs().max().item() <= TOL
    else:
        assert so is None


def test_bottleneck_sequence_split_equals_fused(ops):
    """FusedSequential of two Bottlenecks: the split path (k3t16 + pointwise chain) against the reference arithmetic"""
    from sis3d.nets import backbones as bb
    torch.manual_seed(3)
    seq = bb.FusedSequential(bb.Bottleneck(128, 64), bb.Bottleneck(128, 64)).eval()
    x = torch.randn(1, 128, 24, 12, 24)
    with torch.no_grad():
        y = x
        for blk in seq:
            z = F.relu(F.conv3d(y, blk.conv1.weight, blk.conv1.bias))
            z = F.relu(F.conv3d(z, blk.conv2.weight, blk.conv2.bias, padding=1))
            y = F.relu(F.conv3d(z, blk.conv3.weight, blk.conv3.bias) + y)
        got = seq.cuda()(cl(x))
    assert (got.cpu() - y).abs().max().item() <= TOL


PW16 = [(32, 32, 32), (32, 32, None), (32, 64, 32), (32, 64, None), (32, 128, 32), (32, 128, None), (64, 128, 64), (64, 128, None),
        (64, 64, None), (64, 32, None), (128, 64, None), (128, 32, None), (128, 128, None)]


@pytest.mark.parametrize("cin,cout,cnext", PW16)
@pytest.mark.parametrize("dims", [(24, 12, 24), (7, 5, 3)])
def test_register_chained_pointwise_vs_torch_cpu(ops, cin, cout, cnext, dims):
    """every instantiation of sis3d_conv3d_pw16 (csrc/pointwise.hip), with and without residual, ragged last voxel tile"""
    g = torch.Generator().manual_seed(cin * 13 + cout + (cnext or 0))
    y2 = torch.randn(1, cin, *dims, generator=g)
    res = torch.randn(1, cout, *dims, generator=g)
    w3, b3 = _w(cout, cin, 1, g), torch.randn(cout, generator=g) * 0.1
    pc3 = ops.PackedConv(w3.cuda(), b3.cuda())
    assert pc3.packed_pw16 is not None
    stage = None
    if cnext is not None:
        w1, b1 = _w(cnext, cout, 1, g), torch.randn(cnext, generator=g) * 0.1
        stage = dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True)
    want = F.relu(F.conv3d(y2, w3, b3) + res)
    main, so = ops.conv3d_pw16(cl(y2), pc3, residual=cl(res), relu=True, stage=stage)
    assert ops.is_cl(main) and (main.cpu() - want).abs().max().item() <= TOL
    if cnext is not None:
        assert (so.cpu() - F.relu(F.conv3d(want, w1, b1))).abs().max().item() <= TOL
    # no residual, no ReLU, no bias, written into a channel range of a wider tensor
    pc0 = ops.PackedConv(w3.cuda(), None)
    wide = ops.new_act(cout + 32, dims, torch.device("cuda")).fill_(5.0)
    ops.conv3d_pw16(cl(y2), pc0, relu=False, out=wide, out_coff=16)
    assert (wide[:, 16:16 + cout].cpu() - F.conv3d(y2, w3)).abs().max().item() <= TOL
    assert bool((wide[:, :16] == 5.0).all()) and bool((wide[:, 16 + cout:] == 5.0).all())


@pytest.mark.parametrize("cin,nc", [(64, 19), (128, 19), (64, 32), (64, 21)])
def test_pointwise_sigmoid_head_with_padded_couts_vs_torch_cpu(ops, cin, nc):
    """r6: the mask head's last layer (Conv3d(64, NUM_CLASSES, 1) + sigmoid, lib/nets/backbones.py:247 + network.py:312) on the
    register-chained pointwise kernel: couts padded to whole 16-row tiles (zero weight rows, zero bias), sigmoid in the epilogue;
    rows of the padded output beyond NUM_CLASSES hold sigmoid(0) = 0.5 and are never viewed"""
    g = torch.Generator().manual_seed(cin + nc)
    nvox = 1237                                                    # ragged last 16-voxel tile
    x = torch.randn(nvox, cin, generator=g)
    w, b = _w(nc, cin, 1, g), torch.randn(nc, generator=g) * 0.1
    pc = ops.PackedConv(w.cuda(), b.cuda(), pad_cout16=True)
    assert pc.packed_pw16 is not None and (nc % 16 == 0 or pc.bias16.numel() == 32)
    ncp = (nc + 15) // 16 * 16
    out = torch.full((nvox, ncp), 7.0, device="cuda")
    xc = x.cuda()
    rc = ops.lib().sis3d_conv3d_pw16(xc.data_ptr(), nvox, cin, cin, pc.packed_pw16.data_ptr(), (pc.bias16 if nc % 16 else pc.bias).data_ptr(),
                                     ncp, ops.EPI_SIGMOID, None, 0, out.data_ptr(), ncp, 0, None, None, 0, 0, None, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    want = torch.sigmoid(x @ w.view(nc, cin).t() + b)
    assert (out[:, :nc].cpu() - want).abs().max().item() <= TOL
    assert bool((out[:, nc:] == 0.5).all())
    # a sigmoid cannot feed a chained second stage
    assert ops.lib().sis3d_conv3d_pw16(xc.data_ptr(), nvox, 64, 64, pc.packed_pw16.data_ptr(), None, 32, ops.EPI_SIGMOID, None, 0, out.data_ptr(),
                                       ncp, 0, pc.packed_pw16.data_ptr(), None, 32, 0, out.data_ptr(), 32, None) != 0


def test_plain_k1_conv_routes_through_pw16_and_matches_legacy(ops):
    g = torch.Generator().manual_seed(77)
    x = cl(torch.randn(1, 128, 24, 12, 24, generator=g))
    w, b = _w(64, 128, 1, g).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    pc = ops.PackedConv(w, b)
    new = ops.conv3d(x, pc, relu=True)
    saved, pc.packed_pw16 = pc.packed_pw16, None
    old = ops.conv3d(x, pc, relu=True)                              # conv3d.hip's generic 1x1x1 path
    pc.packed_pw16 = saved
    assert (new - old).abs().max().item() <= 2e-5
    assert (new.cpu() - F.relu(F.conv3d(x.cpu(), w.cpu(), b.cpu()))).abs().max().item() <= TOL


@pytest.mark.parametrize("cin,cout,cnext,dims", [(32, 128, 32, (48, 24, 48)), (32, 128, None, (10, 6, 14)), (64, 64, 32, (20, 12, 16)),
                                                 (32, 64, 32, (9, 7, 11)), (64, 128, 32, (8, 8, 8))])
def test_k2s2_stem_register_chained_vs_torch_cpu(ops, cin, cout, cnext, dims):
    """Conv3d(k2, s2, bias=False) + ReLU [+ next conv1] (backbones.py:193,207 + :29-31); odd extents floor like nn.Conv3d"""
    g = torch.Generator().manual_seed(cin + cout + dims[0])
    x = torch.randn(1, cin, *dims, generator=g)
    w = _w(cout, cin, 2, g)
    pc = ops.PackedConv(w.cuda(), None)
    stage, w1, b1 = None, None, None
    if cnext is not None:
        w1, b1 = _w(cnext, cout, 1, g), torch.randn(cnext, generator=g) * 0.1
        stage = dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True)
    main, so = ops.conv3d_k2s2_pw16(cl(x), pc, relu=True, stage=stage)
    want = F.relu(F.conv3d(x, w, None, stride=2))
    assert main.shape == want.shape and (main.cpu() - want).abs().max().item() <= TOL
    if cnext is not None:
        assert (so.cpu() - F.relu(F.conv3d(want, w1, b1))).abs().max().item() <= TOL


@pytest.mark.parametrize("cout,cnext,dims", [(32, 32, (96, 48, 96)), (64, 32, (30, 22, 18)), (32, None, (7, 9, 5))])
def test_planar_stem_register_chained_vs_torch_cpu(ops, cout, cnext, dims):
    """geometry1[0] on the planar 2-channel grid + the first Bottleneck's conv1 (backbones.py:188 + :29-31)"""
    g = torch.Generator().manual_seed(cout + dims[2])
    x = torch.randn(1, 2, *dims, generator=g)
    w = _w(cout, 2, 2, g)
    stage, w1, b1 = None, None, None
    if cnext is not None:
        w1, b1 = _w(cnext, cout, 1, g), torch.randn(cnext, generator=g) * 0.1
        stage = dict(pc=ops.PackedConv(w1.cuda(), b1.cuda()), relu=True)
    pk = ops.pack_stem_planar2(w.cuda())
    x0, y1 = ops.stem_planar2(x.cuda(), pk, cout, relu=True, stage=stage)
    want = F.relu(F.conv3d(x, w, None, stride=2))
    assert ops.is_cl(x0) and x0.shape == want.shape and (x0.cpu() - want).abs().max().item() <= TOL
    if cnext is not None:
        assert (y1.cpu() - F.relu(F.conv3d(want, w1, b1))).abs().max().item() <= TOL
    # a window view of a larger grid (non-contiguous x / y strides), as the mask head and whole-scene callers pass
    big = torch.randn(1, 2, dims[0] + 4, dims[1] + 2, dims[2], generator=g).cuda()
    sub = big[:, :, 2:2 + dims[0], 1:1 + dims[1], :]
    x0b, _ = ops.stem_planar2(sub, pk, cout, relu=True, stage=None)
    assert (x0b.cpu() - F.relu(F.conv3d(sub.cpu(), w, None, stride=2))).abs().max().item() <= TOL


@pytest.mark.parametrize("A1,A2", [(3, 11), (3, 6)])
def test_rpn_heads_one_launch_vs_torch_cpu(ops, A1, A2):
    """network.py:541-549 for both levels: cls 1x1x1 -> view(1,2,A,..).permute(0,1,3,4,5,2), softmax over dim 1; bbox 1x1x1 -> permute"""
    g = torch.Generator().manual_seed(A1 * 100 + A2)
    dims = (24, 12, 24)
    outs, ins, pcs = [], [], []
    for A in (A1, A2):
        r = torch.randn(1, 256, *dims, generator=g).clamp_(min=0)
        wc, bc = _w(2 * A, 256, 1, g) * 4, torch.randn(2 * A, generator=g)
        wb, bb = _w(6 * A, 256, 1, g), torch.randn(6 * A, generator=g) * 0.1
        bbox = F.conv3d(r, wb, bb).permute(0, 2, 3, 4, 1).contiguous()
        score = F.conv3d(r, wc, bc).view(1, 2, A, *dims).permute(0, 1, 3, 4, 5, 2).contiguous()
        outs.append((score, bbox, F.softmax(score, dim=1)))
        ins.append(cl(r))
        pcs.append(ops.PackedConv(torch.cat([wc, wb], 0).cuda(), torch.cat([bc, bb], 0).cuda(), pad_cout16=True))
    got = ops.rpn_heads(ins[0], pcs[0], A1, ins[1], pcs[1], A2)
    for (s, b, p), (ws, wb_, wp) in zip(got, outs):
        assert s.shape == ws.shape and b.shape == wb_.shape and s.is_contiguous() and b.is_contiguous()
        assert (s.cpu() - ws).abs().max().item() <= TOL and (b.cpu() - wb_).abs().max().item() <= TOL
        assert (p.cpu() - wp).abs().max().item() <= TOL
