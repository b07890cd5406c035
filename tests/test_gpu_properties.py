"""Size-independent properties at the full BASELINE sizes (96x48x96 chunks, 33k candidates, 6912-voxel x 128-channel
maps): idempotence, sortedness, linearity, permutation consistency -- checks that do not need the CPU oracle to finish."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import synthetic  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from sis3d import ops as o
    o.lib()
    return o


def _boxes(n, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(n, 3, generator=g) * torch.tensor([90.0, 44.0, 90.0])
    return torch.cat([lo, lo + 1 + torch.rand(n, 3, generator=g) * 24], 1).cuda()


@pytest.mark.parametrize("n", [1024, 6400])
def test_nms_idempotent_and_independent(ops, n):
    b = _boxes(n, n)
    keep = ops.nms(b, 0.35)
    kept = b[keep]
    assert torch.equal(ops.nms(kept, 0.35).cpu(), torch.arange(keep.numel()))          # survivors suppress nobody
    assert bool((keep[1:] > keep[:-1]).all()) and int(keep[0]) == 0                      # ascending, best box always kept
    # appending boxes AFTER the list never changes which of the earlier ones survive
    more = torch.cat([b, _boxes(500, 7)], 0)
    keep2 = ops.nms(more, 0.35)
    assert torch.equal(keep2[keep2 < n], keep)


def test_topk_is_sorted_permutation_prefix(ops):
    g = torch.Generator().manual_seed(3)
    s = torch.rand(40960, generator=g).round(decimals=3).cuda()                          # many ties
    v, idx = ops.topk_desc(s, 1024)
    assert bool((v[:-1] >= v[1:]).all()) and torch.equal(s[idx], v) and idx.unique().numel() == 1024
    tie = v[:-1] == v[1:]
    assert bool((idx[1:][tie] > idx[:-1][tie]).all())                                     # ties by ascending index (stable)
    assert float(v[-1]) >= float(torch.kthvalue(-s, 1024).values.neg())                  # nothing larger was left out


def test_conv_linearity_rpn_layer(ops):
    """conv(a*x + y) == a*conv(x) + conv(y) on the dominant layer's shape (128 -> 256, k3, 24x12x24), no bias / ReLU"""
    dims = (24, 12, 24)
    x = ops.new_act(128, dims, torch.device("cuda")).normal_()
    y = ops.new_act(128, dims, torch.device("cuda")).normal_()
    pc = ops.PackedConv(torch.randn(256, 128, 3, 3, 3, device="cuda") * 0.02, None)
    z = ops.new_act(128, dims, torch.device("cuda"))
    z.copy_(2.5 * x + y)
    lhs = ops.conv3d(z, pc)
    rhs = 2.5 * ops.conv3d(x, pc) + ops.conv3d(y, pc)
    assert float((lhs - rhs).abs().max()) <= 1e-4 * float(rhs.abs().max()) + 1e-5
    # a shifted input gives a shifted output away from the border (translation equivariance of the k3 p1 conv)
    xs = ops.new_act(128, dims, torch.device("cuda")).zero_()
    xs[:, :, 1:] = x[:, :, :-1]
    a, b = ops.conv3d(x, pc), ops.conv3d(xs, pc)
    assert float((a[:, :, 1:-2] - b[:, :, 2:-1]).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-6


def test_roi_pool_is_max_of_window_and_monotone(ops):
    g = torch.Generator().manual_seed(1)
    f = torch.randn(1, 128, 24, 12, 24, generator=g).cuda()
    rois = _boxes(200, 5)
    out = ops.roi_pool(f, rois, (4, 4, 4), 0.25)
    out = out[0] if isinstance(out, tuple) else out
    assert float(out.max()) <= float(f.max())
    out2 = ops.roi_pool(f + 1.0, rois, (4, 4, 4), 0.25)
    out2 = out2[0] if isinstance(out2, tuple) else out2
    nz = out != 0                                                # empty bins stay 0 (roi_pooling.c:95-97)
    assert torch.allclose(out2[nz], out[nz] + 1.0, atol=1e-6)


def test_projection_of_disjoint_views_is_their_sum(ops):
    """views that see disjoint voxel sets: the max with the implicit-zero rule equals relu-free sum only where one view sees
    the voxel and the others count as zero -> max(f, 0)"""
    dims = synthetic.CHUNK_DIMS
    feats, i3d, i2d = synthetic.synth_views(3, n_views=2, n_per_view=2000)
    nvox = dims[0] * dims[1] * dims[2]
    perm = torch.randperm(nvox, generator=torch.Generator().manual_seed(9))
    i3d[0, 1:2001] = perm[:2000].sort().values
    i3d[1, 1:2001] = perm[2000:4000].sort().values
    both = ops.project_views_max(feats.cuda(), i3d.cuda(), i2d.cuda(), dims, channels_last=True)
    one = [ops.project_views_max(feats[v:v + 1].cuda(), i3d[v:v + 1].cuda(), i2d[v:v + 1].cuda(), dims, channels_last=True) for v in range(2)]
    assert torch.equal(both, torch.clamp(one[0], min=0) + torch.clamp(one[1], min=0))
