"""The one-launch Bottleneck body (csrc/bottleneck.hip, sis3d_bottleneck16): relu(conv3(relu(conv2(y1))) + x) and the next
block's conv1, against torch-CPU operators (lib/nets/backbones.py:17-40; tolerance 1e-4, north_star) and -- bit for bit --
against the two-launch path (sis3d_conv3d_k3t16 + sis3d_conv3d_pw16) whose summation orders it keeps."""
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


def _case(ops, planes, cio, c2, dims, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, cio, *dims, generator=g)
    y1 = torch.relu(torch.randn(1, planes, *dims, generator=g))
    w2, w3 = _w(planes, planes, 3, g), _w(cio, planes, 1, g)
    b2 = torch.randn(planes, generator=g) * 0.1 if bias else None
    b3 = torch.randn(cio, generator=g) * 0.1 if bias else None
    want = F.relu(F.conv3d(F.relu(F.conv3d(y1, w2, b2, padding=1)), w3, b3) + x)
    pc2 = ops.PackedConv(w2.cuda(), b2.cuda() if bias else None)
    pc3 = ops.PackedConv(w3.cuda(), b3.cuda() if bias else None)
    stage, wantn = None, None
    if c2:
        w1n, b1n = _w(c2, cio, 1, g), torch.randn(c2, generator=g) * 0.1
        stage = dict(pc=ops.PackedConv(w1n.cuda(), b1n.cuda()), relu=True)
        wantn = F.relu(F.conv3d(want, w1n, b1n))
    return x, y1, pc2, pc3, stage, want, wantn


SHAPES = [(32, 32, 32, (48, 24, 48)), (32, 32, 0, (48, 24, 48)), (32, 128, 32, (24, 12, 24)), (32, 128, 0, (24, 12, 24)),
          (32, 64, 0, (24, 12, 24)), (32, 64, 0, (48, 24, 48))]


@pytest.mark.parametrize("planes,cio,c2,dims", SHAPES)
def test_network_shapes_vs_torch_cpu_and_two_launch_path(ops, planes, cio, c2, dims, monkeypatch):
    monkeypatch.setattr(ops, "BNECK_WINO", False)                 # the direct-convolution body (the Winograd body: tests below)
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, planes, cio, c2, dims, planes + cio + c2 + dims[0])
    out, y1n = ops.bottleneck16(cl(y1), pc2, pc3, cl(x), stage=stage)
    assert ops.is_cl(out) and out.shape == want.shape
    assert (out.cpu() - want).abs().max().item() <= TOL
    y2 = ops.conv3d_k3t16([cl(y1)], [pc2], relu=True)[0]
    ref, refn = ops.conv3d_pw16(y2, pc3, residual=cl(x), relu=True, stage=stage)
    assert torch.equal(out, ref)                                  # same partial-sum order as the two launches
    if c2:
        assert (y1n.cpu() - wantn).abs().max().item() <= TOL
        assert (y1n - refn).abs().max().item() <= 1e-5
    else:
        assert y1n is None


# partial bricks, grids smaller than a brick, every brick forced; no bias; output into a channel range of a wider tensor
@pytest.mark.parametrize("brick", [0, 1, 2])
@pytest.mark.parametrize("dims", [(13, 9, 11), (5, 3, 2), (7, 6, 19)])
def test_ragged_grids_both_bricks(ops, brick, dims):
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, 32, 128, 32, dims, 7 * brick + dims[2], bias=(dims[0] != 5))
    wide = torch.full((1, 192, *dims), -7.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    out, y1n = ops.bottleneck16(cl(y1), pc2, pc3, cl(x), out=wide, out_coff=64, stage=stage, brick=brick)
    assert out is wide
    assert (wide[:, 64:].cpu() - want).abs().max().item() <= TOL
    assert float(wide[:, :64].min()) == -7.0 and float(wide[:, :64].max()) == -7.0
    assert (y1n.cpu() - wantn).abs().max().item() <= TOL


def test_unsupported_shapes_fall_back_loudly(ops):
    x, y1, pc2, pc3, stage, want, _ = _case(ops, 64, 128, 0, (24, 12, 24), 5)     # 64 planes: the two-launch path is faster
    with pytest.raises(ops.Sis3dUnsupported):
        ops.bottleneck16(cl(y1), pc2, pc3, cl(x))
    x, y1, pc2, pc3, stage, want, _ = _case(ops, 32, 96, 0, (12, 6, 12), 6)       # no (32, 96) instantiation
    with pytest.raises(ops.Sis3dUnsupported):
        ops.bottleneck16(cl(y1), pc2, pc3, cl(x))


def test_fused_sequential_uses_it_and_matches_split(ops, monkeypatch):
    from sis3d.nets import backbones as bb
    torch.manual_seed(3)
    seq = bb.FusedSequential(bb.Bottleneck(128, 32), bb.Bottleneck(128, 32)).cuda().eval()
    x = cl(torch.randn(1, 128, 24, 12, 24))
    calls = []
    real = ops.bottleneck16
    monkeypatch.setattr(ops, "bottleneck16", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        fused = seq(x)
        assert len(calls) == 2
        monkeypatch.setattr(ops, "BNECK_SPLIT", True)
        split = seq(x)
    assert (fused - split).abs().max().item() <= 1e-5


# ---- r4: the same body with conv2 on the Winograd kernel and the 1x1x1 tail in its epilogue (sis3d_bottleneck_wino)
WINO_SHAPES = [(32, 32, 32, (48, 24, 48)), (32, 32, 0, (48, 24, 48)), (32, 64, 0, (48, 24, 48))]


@pytest.mark.parametrize("planes,cio,c2,dims", WINO_SHAPES)
def test_winograd_body_network_shapes(ops, planes, cio, c2, dims):
    """vs torch-CPU (oneDNN fp32, tolerance 1e-4 = north_star), vs a float64 evaluation of the same block (2e-5 of the output scale:
    the bound test_gpu_conv_wino.py holds the kernel to) and vs the direct-convolution body; and it IS what bottleneck16 dispatches
    to on these grids."""
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, planes, cio, c2, dims, 11 + planes + cio + c2)
    assert ops.lib().sis3d_bottleneck_wino_prefer(*dims, planes, cio, c2, 0) == 1
    out, y1n = ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x), stage=stage)
    assert ops.is_cl(out) and out.shape == want.shape
    assert (out.cpu() - want).abs().max().item() <= TOL
    w2, b2 = pc2._w.cpu().double(), pc2.bias.cpu().double()
    g = torch.Generator().manual_seed(11 + planes + cio + c2)                 # re-draw the 1x1x1 weights of _case in float64
    _ = torch.randn(1, cio, *dims, generator=g); _ = torch.randn(1, planes, *dims, generator=g)
    w2r, w3r = _w(planes, planes, 3, g), _w(cio, planes, 1, g)
    b2r, b3r = torch.randn(planes, generator=g) * 0.1, torch.randn(cio, generator=g) * 0.1
    assert torch.equal(w2r.double(), w2)
    want64 = F.relu(F.conv3d(F.relu(F.conv3d(y1.double(), w2, b2, padding=1)), w3r.double(), b3r.double()) + x.double())
    scale = float(want64.abs().max())
    assert (out.cpu().double() - want64).abs().max().item() <= 2e-5 * scale
    via16, via16n = ops.bottleneck16(cl(y1), pc2, pc3, cl(x), stage=stage)   # default dispatch = this kernel
    assert torch.equal(via16, out)
    if c2:
        assert (y1n.cpu() - wantn).abs().max().item() <= TOL and torch.equal(via16n, y1n)
    else:
        assert y1n is None and via16n is None


@pytest.mark.parametrize("dims", [(13, 9, 11), (5, 3, 2), (16, 8, 16), (17, 4, 9)])
@pytest.mark.parametrize("cio,c2", [(32, 32), (64, 0)])
def test_winograd_body_partial_blocks_and_channel_range(ops, dims, cio, c2):
    """grids that are not multiples of the 8 x 4 x 8 block (clipped tiles, zero halo), no-bias layers, output written into a channel
    range of a wider tensor"""
    bias = dims[0] != 5
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, 32, cio, c2, dims, 3 * dims[0] + cio, bias=bias)
    wide = torch.full((1, cio + 96, *dims), -7.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    out, y1n = ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x), out=wide, out_coff=64, stage=stage)
    assert out is wide
    assert (wide[:, 64:64 + cio].cpu() - want).abs().max().item() <= TOL
    rest = torch.cat([wide[:, :64], wide[:, 64 + cio:]], 1)
    assert float(rest.min()) == -7.0 and float(rest.max()) == -7.0
    if c2:
        assert (y1n.cpu() - wantn).abs().max().item() <= TOL


def test_winograd_body_refuses_other_planes(ops):
    x, y1, pc2, pc3, stage, want, _ = _case(ops, 64, 128, 0, (24, 12, 24), 5)
    assert ops.lib().sis3d_bottleneck_wino_prefer(24, 12, 24, 64, 128, 0, 0) == 0
    with pytest.raises(ops.Sis3dUnsupported):
        ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x))
    assert ops.lib().sis3d_bottleneck_wino_prefer(24, 12, 24, 32, 128, 32, 0) == 0     # 27 blocks: the direct body serves the 24 x 12 x 24 maps


# ---- r4: dispatch on a SHARED chip (several chunks in flight) -- fewer, fatter Winograd work items.  r5: the regime is the calling
# thread's (ops.dispatch_regime) and reaches the library as per-call arguments; nothing process-wide is set or reset
@pytest.fixture
def shared_chip(ops):
    with ops.dispatch_regime(shared_chip=True):
        yield


@pytest.mark.parametrize("c2", [32, 0])
def test_winograd_body_128_channels_on_a_shared_chip(ops, c2, shared_chip):
    """the Bottleneck(128, 32) bodies of the 24 x 12 x 24 maps (geometry1.6 / .7, lib/nets/backbones.py:200-203) as 27 Winograd work items:
    only where the hint is set; same bounds as the planes-32 bodies of the 48 x 24 x 48 maps"""
    dims = (24, 12, 24)
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, 32, 128, c2, dims, 77 + c2)
    assert ops.lib().sis3d_bottleneck_wino_prefer(*dims, 32, 128, 0, 1) == 1
    assert ops.lib().sis3d_bottleneck_wino_prefer(*dims, 32, 128, 32, 1) == 0    # no instantiation holds both 128-channel weight sets of the tail
    out, none = ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x))
    assert none is None and (out.cpu() - want).abs().max().item() <= TOL
    if c2:
        with pytest.raises(ops.Sis3dUnsupported):
            ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x), stage=stage)
    ops.flop_tally(True)
    via16, via16n = ops.bottleneck16(cl(y1), pc2, pc3, cl(x), stage=stage)   # the dispatch: this kernel (+ the next conv1 as a pointwise launch)
    assert ops.flop_tally(False)["wino_launches"] == 1
    assert torch.equal(via16, out)
    if c2:
        assert (via16n.cpu() - wantn).abs().max().item() <= TOL
    else:
        assert via16n is None
    assert ops.lib().sis3d_bottleneck_wino_prefer(*dims, 32, 128, 0, 0) == 0
    with ops.dispatch_regime(shared_chip=False):
        direct, _ = ops.bottleneck16(cl(y1), pc2, pc3, cl(x), stage=stage)   # hint off: the direct-convolution body, same result within fp32 order
    assert (direct - out).abs().max().item() <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("dims", [(13, 9, 11), (24, 12, 24)])
def test_winograd_body_128_channels_partial_blocks(ops, dims):
    x, y1, pc2, pc3, stage, want, wantn = _case(ops, 32, 128, 0, dims, 5 * dims[0], bias=(dims[0] != 13))
    wide = torch.full((1, 128 + 96, *dims), -7.0).cuda().contiguous(memory_format=torch.channels_last_3d)
    out, y1n = ops.bottleneck_wino(cl(y1), pc2, pc3, cl(x), out=wide, out_coff=64)
    assert y1n is None
    assert (wide[:, 64:192].cpu() - want).abs().max().item() <= TOL
    rest = torch.cat([wide[:, :64], wide[:, 192:]], 1)
    assert float(rest.min()) == -7.0 and float(rest.max()) == -7.0


def test_shared_chip_sends_the_64_channel_convs_to_winograd(ops, shared_chip):
    """conv2 of geometry2's Bottleneck(128, 64) blocks (64 -> 64 on 24 x 12 x 24: 54 Winograd work items) and geometry2[0] with two cout
    tiles per workgroup: taken only on a shared chip; results equal the direct kernel's within the fp32-order bound"""
    lib = ops.lib()
    assert lib.sis3d_conv3d_k3wino_prefer(24, 12, 24, 64, 64, 1, 1) == 1
    assert lib.sis3d_conv3d_k3wino_prefer(12, 6, 12, 64, 64, 1, 1) == 0         # 8 work items: not even there
    g = torch.Generator().manual_seed(9)
    x = cl(torch.relu(torch.randn(1, 64, 24, 12, 24, generator=g)))
    w, b = _w(64, 64, 3, g), torch.randn(64, generator=g) * 0.1
    pc = ops.PackedConv(w.cuda(), b.cuda())
    ops.flop_tally(True)
    y = ops.conv3d_k3t16([x], [pc], relu=True)[0]
    assert ops.flop_tally(False)["wino_launches"] == 1
    assert lib.sis3d_conv3d_k3wino_prefer(24, 12, 24, 64, 64, 1, 0) == 0
    with ops.dispatch_regime(shared_chip=False):
        ops.flop_tally(True)
        d = ops.conv3d_k3t16([x], [pc], relu=True)[0]
        assert ops.flop_tally(False)["wino_launches"] == 0
    want = F.relu(F.conv3d(x.cpu().double(), w.double(), b.double(), padding=1))
    scale = float(want.abs().max())
    assert (y.cpu().double() - want).abs().max().item() <= 2e-5 * scale
    assert (y - d).abs().max().item() <= 2e-5 * scale
