"""Parity of the HIP kernels (through the C ABI / sis3d.ops) against the CPU oracle and the
reference-generated golden fixtures.  Bit-exact for NMS keep lists, RoI-pool values + argmax and
projection scatters; fp32 tolerances stated per test for decode / conv."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, synthetic  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from sis3d import ops as o
    o.lib()  # fails loudly if libsis3d_hip.so is missing
    return o


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------------------------ NMS
def test_nms_golden_bit_exact(ops, golden):
    g = golden("nms_cases")
    for name in sorted({k.split("/")[0] for k in g.files}):
        boxes = torch.from_numpy(g[name + "/boxes"])
        for th in (0.1, 0.35, 0.5):
            keep = ops.nms(dev(boxes), th).cpu().numpy()
            assert np.array_equal(keep, g["%s/keep_%g" % (name, th)]), (name, th)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 400, 1024, 1025, 2500, 6400])
def test_nms_random_vs_oracle(ops, oracle, n):
    g = torch.Generator().manual_seed(n)
    lo = torch.rand(n, 3, generator=g) * torch.tensor([380.0, 44.0, 760.0])
    sz = torch.rand(n, 3, generator=g) * 40.0 + 1.0
    boxes = torch.cat([lo, lo + sz], 1)
    for th in (0.1, 0.5):
        want = oracle.nms(boxes, th)
        got = ops.nms(dev(boxes), th).cpu()
        assert torch.equal(got, want), (n, th)
        # max_keep == truncation of the full list
        got_k = ops.nms(dev(boxes), th, max_keep=7).cpu()
        assert torch.equal(got_k, want[:7])
    if n >= 2:
        m = ops.nms_mask(dev(boxes), 0.3).cpu()
        want_m = oracle.nms_mask(boxes, 0.3)
        cb = (n + 63) // 64
        # only tiles on/above the diagonal are defined by the kernel contract
        for i in range(0, n, max(1, n // 50)):
            assert torch.equal(m[i, i // 64:], want_m[i, i // 64:]), i
        assert m.shape == (n, cb)


@pytest.mark.parametrize("path", [1, 2])
def test_nms_both_algorithms(ops, oracle, golden, path):
    """the one-workgroup sweep and the parallel resolve (csrc/nms.hip) forced at every size (r5: `path` is an argument of sis3d_nms, not library state): goldens, random sets, ties / NaN,
    and a suppression CHAIN (box k overlaps only k-1 and k+1: the resolve needs one sweep per link)"""
    if True:
        g = golden("nms_cases")
        for name in sorted({k.split("/")[0] for k in g.files}):
            boxes = torch.from_numpy(g[name + "/boxes"])
            for th in (0.1, 0.35, 0.5):
                assert np.array_equal(ops.nms(dev(boxes), th, path=path).cpu().numpy(), g["%s/keep_%g" % (name, th)]), (name, th)
        for n in (1, 2, 63, 64, 65, 400, 1025, 2500):
            gen = torch.Generator().manual_seed(100 + n)
            lo = torch.rand(n, 3, generator=gen) * torch.tensor([90.0, 44.0, 90.0])      # dense: long suppression chains
            boxes = torch.cat([lo, lo + torch.rand(n, 3, generator=gen) * 30.0 + 1.0], 1)
            for th in (0.05, 0.3):
                want = oracle.nms(boxes, th)
                assert torch.equal(ops.nms(dev(boxes), th, path=path).cpu(), want), (n, th)
                assert torch.equal(ops.nms(dev(boxes), th, max_keep=5, path=path).cpu(), want[:5])
        k = torch.arange(300, dtype=torch.float32)
        chain = torch.stack([6 * k, 0 * k, 0 * k, 6 * k + 9, 0 * k + 9, 0 * k + 9], 1)     # IoU(k, k+1) = 400/1600, others 0
        assert torch.equal(ops.nms(dev(chain), 0.2, path=path).cpu(), oracle.nms(chain, 0.2))
        assert ops.nms(dev(chain), 0.2, path=path).numel() == 150
        b = torch.tensor([[0.0, 0, 0, 10, 10, 10]] * 5 + [[float("nan"), 0, 0, 10, 10, 10]] + [[50.0, 20, 50, 60, 30, 60]])
        for th in (0.0, 0.1, 0.999, 1.0):
            assert torch.equal(ops.nms(dev(b), th, path=path).cpu(), oracle.nms(b, th)), th


def test_nms_ties_and_nan(ops, oracle):
    b = torch.tensor([[0.0, 0, 0, 10, 10, 10]] * 5 + [[float("nan"), 0, 0, 10, 10, 10]] + [[50.0, 20, 50, 60, 30, 60]])
    for th in (0.0, 0.1, 0.999, 1.0):
        assert torch.equal(ops.nms(dev(b), th).cpu(), oracle.nms(b, th)), th
    e = torch.zeros(0, 6)
    assert ops.nms(dev(e), 0.1).numel() == 0


def test_nms_select_matches_proposal_layer_tail(ops, oracle):
    g = torch.Generator().manual_seed(5)
    m = 3000
    lo = torch.rand(m, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
    boxes = torch.cat([lo, lo + torch.rand(m, 3, generator=g) * 20 + 1], 1)
    scores = torch.rand(m, generator=g)
    scores[100:110] = scores[100]                      # ties
    lv = (torch.rand(m, generator=g) > 0.5).float() + 1
    s_sorted, order = scores.sort(descending=True, stable=True)
    n_pre, k_out = 400, 200
    keep = oracle.nms(boxes[order[:n_pre]], 0.1)[:k_out]
    rois, rs, rl, kp, num = ops.nms_select(dev(boxes), dev(lv), dev(s_sorted), dev(order), n_pre, 0.1, k_out)
    n = int(num.item())
    assert n == keep.numel()
    assert torch.equal(kp[:n].cpu(), keep)
    assert torch.equal(rois[:n].cpu(), boxes[order[:n_pre]][keep])
    assert torch.equal(rs[:n].cpu(), s_sorted[:n_pre][keep])
    assert torch.equal(rl[:n].cpu(), lv[order[:n_pre]][keep])
    assert (rois[n:] == 0).all() and (rl[n:] == 0).all()


# ------------------------------------------------------------------------------- RoI pool
def test_roi_pool_golden_bit_exact(ops, golden):
    g = golden("roi_pool_cases")
    feat, rois = torch.from_numpy(g["feat"]), torch.from_numpy(g["rois"])
    out, arg = ops.roi_pool(dev(feat), dev(rois), (4, 4, 4), 0.25)
    assert np.array_equal(out.cpu().numpy(), g["out_c_4"])
    out2, arg2 = ops.roi_pool(dev(feat), dev(rois[:6]), (2, 2, 2), 0.25)
    assert np.array_equal(out2.cpu().numpy(), g["out_py_2"])


@pytest.mark.parametrize("layout", ["ncdhw", "channels_last"])
def test_roi_pool_vs_oracle_full_size(ops, oracle, layout):
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(1, 128, 24, 12, 24, generator=g)
    feat[0, :, 5, 5, 5] = feat[0, :, 5, 5, 6]            # equal values inside a window: first max must win
    R = 200
    lo = torch.rand(R, 3, generator=g) * torch.tensor([90.0, 44.0, 90.0])
    rois = torch.cat([lo, (lo + torch.rand(R, 3, generator=g) * 50).clamp(max=96)], 1)
    rois[0] = torch.tensor([0.0, 0, 0, 96, 48, 96])
    want, want_arg = oracle.roi_pool(feat, rois, (4, 4, 4), 0.25)
    f = dev(feat)
    if layout == "channels_last":
        f = f.contiguous(memory_format=torch.channels_last_3d)
    out, arg = ops.roi_pool(f, dev(rois), (4, 4, 4), 0.25, out_channels_last=(layout == "channels_last"))
    assert torch.equal(out.cpu(), want)
    assert torch.equal(arg.cpu(), want_arg)


def test_roi_pool_levels_matches_python_scatter(ops, oracle):
    g = torch.Generator().manual_seed(4)
    f1 = torch.randn(1, 128, 24, 12, 24, generator=g)
    f2 = torch.randn(1, 128, 24, 12, 24, generator=g)
    R = 64
    lo = torch.rand(R, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
    rois = torch.cat([lo, lo + torch.rand(R, 3, generator=g) * 30], 1)
    lv = (torch.rand(R, generator=g) > 0.4).float() + 1
    lv[-3:] = 0                                           # padded rows
    want = torch.zeros(R, 128, 4, 4, 4)
    for lid, f in ((1, f1), (2, f2)):
        idx = (lv == lid).nonzero()[:, 0]
        want[idx] = oracle.roi_pool(f, rois[idx], (4, 4, 4), 0.25, want_argmax=False)
    cl = lambda t: dev(t).contiguous(memory_format=torch.channels_last_3d)
    out = ops.roi_pool_levels(cl(f1), cl(f2), dev(rois), dev(lv), 4, 0.25, out_channels_last=True)
    assert torch.equal(out.cpu(), want)


# ----------------------------------------------------------------------------- projection
def test_projection_golden_bit_exact(ops, golden):
    g = golden("projection_cases")
    dims = tuple(int(v) for v in g["dims"])
    feats, i3d, i2d = (torch.from_numpy(g[k]) for k in ("feats", "i3d", "i2d"))
    for v in range(feats.shape[0]):
        out = ops.projection(dev(feats[v]), dev(i3d[v]), dev(i2d[v]), dims)
        assert np.array_equal(out.cpu().numpy(), g["out"][v])
    out2 = ops.projection(dev(feats[0, 0]), dev(i3d[0]), dev(i2d[0]), dims)
    assert np.array_equal(out2.cpu().numpy(), g["out2d"])


@pytest.mark.parametrize("n_per_view,kill", [(3000, ()), (60000, ()), (3000, (1, 3)), (0, ())])
def test_project_views_max_full_size(ops, oracle, n_per_view, kill):
    dims = synthetic.CHUNK_DIMS
    feats, i3d, i2d = synthetic.synth_views(0, n_views=5, n_per_view=n_per_view)
    want = oracle.project_views_max(feats, i3d, i2d, dims, kill)          # logical (1,C,X,Y,Z)
    for cl in (True, False):
        got = ops.project_views_max(dev(feats), dev(i3d), dev(i2d), dims, kill, channels_last=cl)
        assert got.shape == want.shape
        if not cl:
            assert got.stride() == want.stride()                          # the reference's (C,Z,Y,X) memory order
        assert torch.equal(got.cpu(), want), (cl, n_per_view, kill)


def test_project_views_single_view_is_plain_copy(ops, oracle):
    dims = (16, 8, 12)
    feats, i3d, i2d = synthetic.synth_views(1, n_views=1, n_per_view=100, channels=8, image_hw=(4, 5), dims=dims)
    want = oracle.project_views_max(feats, i3d, i2d, dims)
    got = ops.project_views_max(dev(feats), dev(i3d), dev(i2d), dims, channels_last=True)
    assert torch.equal(got.cpu(), want)
    assert (want < 0).any()                                               # negatives survive: no zero clamp with one view


# --------------------------------------------------------------- voxel -> pixel visibility
def _helper(dims):
    from sis3d.layer_utils.projection import ProjectionHelper
    c = config.scannet_benchmark_cfg()
    return c, ProjectionHelper(c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.DEPTH_SHAPE, list(dims), c.VOXEL_SIZE)


def test_compute_projection_golden_bit_exact(ops, golden):
    """HIP compute_projection == the reference's own output (fixtures from oracle/make_golden.py), None exits included,
    single-view reference signature and the batched form."""
    g = golden("compute_projection_cases")
    for name in ("small", "odd", "chunk"):
        dims = tuple(int(v) for v in g[name + "_dims"])
        _, helper = _helper(dims)
        depth, c2w, w2g = (torch.from_numpy(g[name + k]) for k in ("_depth", "_c2w", "_w2g"))
        counts = g[name + "_counts"]
        l3, l2 = helper.compute_projection_views(dev(depth), c2w, w2g)
        assert l3.shape == (len(counts), dims[0] * dims[1] * dims[2] + 1) and l3.dtype == torch.int64
        for v, n in enumerate(counts):
            one = helper.compute_projection(dev(depth[v]), dev(c2w[v]), dev(w2g[v]))
            assert int(l3[v, 0]) == n and int(l2[v, 0]) == n
            if n == 0:
                assert one is None
                continue
            assert one[0].is_cuda and torch.equal(one[0], l3[v]) and torch.equal(one[1], l2[v])
            assert np.array_equal(l3[v, 1:1 + n].cpu().numpy(), g["%s_l3_%d" % (name, v)])
            assert np.array_equal(l2[v, 1:1 + n].cpu().numpy(), g["%s_l2_%d" % (name, v)])
            assert not l3[v, 1 + n:].any() and not l2[v, 1 + n:].any()


@pytest.mark.parametrize("dims,cid,nv", [((96, 48, 96), 5, 8), ((33, 17, 1025), 6, 3), ((160, 64, 224), 7, 4)])
def test_compute_projection_vs_oracle(ops, oracle, dims, cid, nv):
    """fresh seeded rigs, incl. a whole-scene-sized grid (2.3 M voxels): bit-exact lists vs the CPU oracle"""
    c, helper = _helper(dims)
    depth, c2w, w2g = synthetic.synth_cameras(cid, nv, dims, c.VOXEL_SIZE)
    l3, l2 = helper.compute_projection_views(dev(depth), c2w, w2g)
    total = 0
    for v in range(nv):
        o = oracle.compute_projection(depth[v], c2w[v], w2g[v], c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX,
                                      c.DEPTH_SHAPE, dims, c.VOXEL_SIZE)
        if o is None:
            assert int(l3[v, 0]) == 0 and not l3[v].any() and not l2[v].any()
            continue
        total += int(o[0][0])
        assert torch.equal(l3[v].cpu(), o[0]) and torch.equal(l2[v].cpu(), o[1]), v
    assert total > 0


def test_compute_projection_feeds_project_views(ops, oracle):
    """row f-1 -> row a13: device-built lists drive the fused view max exactly like oracle-built ones"""
    dims = (64, 32, 48)
    c, helper = _helper(dims)
    depth, c2w, w2g = synthetic.synth_cameras(11, 3, dims, c.VOXEL_SIZE)
    feats = torch.randn(3, 16, 32, 41, generator=torch.Generator().manual_seed(5))
    l3, l2 = helper.compute_projection_views(dev(depth), c2w, w2g)
    got = ops.project_views_max(dev(feats), l3, l2, dims, channels_last=True)
    want = oracle.project_views_max(feats, l3.cpu(), l2.cpu(), dims)
    assert torch.equal(got.cpu(), want) and want.abs().sum() > 0


# --------------------------------------------------------------------------------- decode
def test_proposal_decode_and_softmax(ops, oracle):
    c = config.scannet_benchmark_cfg()
    dims = (96, 48, 96)
    g = torch.Generator().manual_seed(8)
    for lv, A in ((1, 3), (2, 11)):
        anchors = torch.from_numpy(oracle.generate_anchors((24, 12, 24), 4, config.anchor_sizes(c, lv)))
        score = torch.randn(1, 2, 24, 12, 24, A, generator=g) * 3
        bbox = torch.randn(1, 24, 12, 24, 6 * A, generator=g) * 0.5
        prob_want = torch.softmax(score, 1)
        prob = ops.softmax2(dev(score))
        assert (prob.cpu() - prob_want).abs().max() <= 1e-6
        inds = oracle.inside_anchor_inds(anchors, dims)
        want_b, want_s = oracle.decode_candidates(prob_want, bbox, anchors, inds, dims)
        n = inds.numel()
        ob, os_, ol = torch.empty(n, 6).cuda(), torch.empty(n).cuda(), torch.empty(n).cuda()
        ops.proposal_decode(dev(anchors), dev(bbox), dev(prob_want)[0, 1], dev(inds.int()), dims, lv, ob, os_, ol)
        assert (ob.cpu() - want_b).abs().max() <= 1e-4          # fp32 box tolerance (expf vs torch exp: <= 2 ulp)
        assert torch.equal(os_.cpu(), want_s[:, 0])
        assert (ol == lv).all()


@pytest.mark.parametrize("n,k", [(33394, 400), (5, 400), (400, 400), (1000, 1), (40960, 1024), (2048, 300)])
def test_topk_desc_matches_stable_sort(ops, n, k):
    g = torch.Generator().manual_seed(n + k)
    s = torch.rand(n, generator=g)
    if n > 100:
        s[torch.randint(0, n, (n // 3,), generator=g)] = 0.5          # many exact ties, also AT the k-th value
        s[7] = float("inf"); s[11] = -1.0; s[13] = -0.0; s[17] = 0.0
    s = (s * 64).round() / 64 if n == 2048 else s                       # heavy tie case
    ws, wo = torch.sort(s, descending=True, stable=True)
    kk = min(k, n)
    gs, go = ops.topk_desc(s.cuda(), k)
    assert torch.equal(go.cpu(), wo[:kk]) and torch.equal(gs.cpu(), ws[:kk])


# saturated scores: softmax outputs that round to exactly 1.0f (or NaN / +inf piles) overflow the rank-sort fast path; with >= k
# candidates on the single largest key the winners are the first k of them by index (the ballot-scan path of topk.hip), with
# fewer the exact selection runs.  Same oracle: torch's stable descending sort.
@pytest.mark.parametrize("n,k,top,frac", [(33394, 400, 1.0, 0.5), (33394, 400, 1.0, 0.02), (33394, 400, 1.0, 0.011), (40960, 1024, 1.0, 0.9),
                                          (33394, 400, float("inf"), 0.3), (33394, 400, float("nan"), 0.2), (4096, 400, 0.25, 1.0),
                                          (33394, 1, 1.0, 0.5)])
def test_topk_desc_saturated_scores(ops, n, k, top, frac):
    g = torch.Generator().manual_seed(n + k + int(frac * 1000))
    s = torch.rand(n, generator=g) * 0.999
    m = torch.rand(n, generator=g) < frac
    s[m] = top
    if top == 1.0:
        s[torch.rand(n, generator=g) < 0.3] = 0.99999994                 # a second pile one ulp below
    ws, wo = torch.sort(s, descending=True, stable=True)
    gs, go = ops.topk_desc(s.cuda(), k)
    assert torch.equal(go.cpu(), wo[:k])
    assert torch.equal(gs.cpu().view(torch.int32), ws[:k].view(torch.int32))     # bit compare (NaN == NaN)


# ------------------------------------------------------------------------- record packing
@pytest.mark.parametrize("num", [0, 1, 137, 200])
def test_pack_records(ops, num):
    """sis3d_pack_records == the torch glue it replaced (cat / gather / where) + the reference's host-side class-box decode"""
    from sis3d.utils.bbox_transform import bbox_transform_inv, clip_boxes
    g = torch.Generator().manual_seed(num)
    K, NC, dims = 200, 19, (96.0, 48.0, 96.0)
    lo = torch.rand(K, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
    rois = torch.cat([lo, lo + 2 + torch.rand(K, 3, generator=g) * 30], 1)
    d = dict(rois=rois, scores=torch.rand(K, generator=g), levels=torch.randint(1, 3, (K,), generator=g).float(),
             cls_pred=torch.randint(0, NC, (K,), generator=g), cls_prob=torch.rand(K, NC, generator=g),
             bbox_pred=torch.randn(K, 6 * NC, generator=g) * 0.3, num=torch.tensor([num], dtype=torch.int32))
    origin = torch.tensor([96.0, 0.0, 192.0])
    rec, blk = ops.pack_records({k: dev(v) for k, v in d.items()}, dims, dev(origin))
    rec, blk = rec.cpu(), blk.cpu()
    conf = d["cls_prob"].gather(1, d["cls_pred"].view(-1, 1))
    want10 = torch.cat([rois, d["scores"].view(-1, 1), d["levels"].view(-1, 1), d["cls_pred"].float().view(-1, 1), conf], 1)
    assert torch.equal(rec[:, :10], want10)
    reg = d["bbox_pred"].view(K, NC, 6)[torch.arange(K), d["cls_pred"]]
    fb = clip_boxes(bbox_transform_inv(rois, reg), dims)
    assert (rec[:, 10:] - fb).abs().max() <= 1e-4                       # expf vs torch exp: <= 2 ulp on boxes <= 96
    assert blk.numel() == 1 + K * ops.RECORD_WIDTH and blk[0] == num
    rows = blk[1:].view(K, ops.RECORD_WIDTH)
    o3 = origin.tolist()
    off = torch.tensor(o3 + o3 + [0.0] * 4 + o3 + o3)
    assert torch.equal(rows[:num], rec[:num] + off) and not rows[num:].any()
    # RPN-only engines (USE_CLASS off): class columns zero, final box = proposal box
    d2 = {k: dev(v) for k, v in d.items() if k in ("rois", "scores", "levels", "num")}
    rec2, _ = ops.pack_records(d2, dims, None, want_block=False)
    rec2 = rec2.cpu()
    assert torch.equal(rec2[:, :8], want10[:, :8]) and not rec2[:, 8:10].any() and torch.equal(rec2[:, 10:], rois)
