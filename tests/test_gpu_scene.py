"""Whole-scene path on one GPU (world size 1): per-chunk captured graph -> record blocks -> whole-scene NMS,
checked against the CPU oracle doing the same thing chunk by chunk."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, synthetic  # noqa: E402


def test_scene_of_four_chunks_vs_oracle(oracle):
    from sis3d import parallel
    from sis3d.engine import RECORD_WIDTH
    from sis3d.nets import backbones
    from sis3d.scene import SceneRunner
    cfg = config.scannet_benchmark_cfg()
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    net.cuda().eval()
    dims = (64, 32, 48)
    chunks = [(c, (float(dims[0] * (c % 2)), 0.0, float(dims[2] * (c // 2))), synthetic.synth_chunk(10 + c, dims)) for c in range(4)]
    runner = SceneRunner(net, dims)
    recs, keep = runner.infer(chunks)
    assert recs.shape[1] == RECORD_WIDTH and recs.shape[0] > 0
    s = recs[:, 6]
    assert bool((s[:-1] >= s[1:]).all())
    # the merge itself is exact: same records through the oracle NMS on the CPU give the same keep list
    assert torch.equal(keep.cpu(), oracle.nms(recs[:, :6].cpu().contiguous(), cfg.TEST.RPN_NMS_THRESH))
    # second run of the captured graph is bit-identical (static buffers, deterministic kernels)
    recs2, keep2 = runner.infer(chunks)
    assert torch.equal(recs, recs2) and torch.equal(keep, keep2)

    # oracle: same per-chunk pipeline + same packing / merge rules
    on = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))

    def odetect(data):
        o = on.forward(data)
        n = o["rois"][0].shape[0]
        k = cfg.TEST.RPN_POST_NMS_TOP_N
        rec = torch.zeros(k, RECORD_WIDTH)
        rec[:n, :6] = o["rois"][0]
        rec[:n, 6] = o["roi_scores"][0][:, 0]
        rec[:n, 7] = o["level_inds"][0]
        rec[:n, 8] = o["cls_pred"].float()
        rec[:n, 9] = o["cls_prob"].gather(1, o["cls_pred"].view(-1, 1))[:, 0]
        rec[:n, 10:16] = torch.from_numpy(oracle.class_boxes(o, data.shape[2:]))
        return rec, n
    orecs, okeep = parallel.infer_scene(chunks, odetect, oracle.nms, cfg.TEST.RPN_POST_NMS_TOP_N, cfg.TEST.RPN_NMS_THRESH)
    # the gathered record sets match one to one; an unmatched record must be a near-tie of the RPN scores
    from parity import assert_proposals_match
    allsc = torch.cat([on.forward(c[2])["_scores_sorted_all"] for c in chunks]).sort(descending=True).values
    assert_proposals_match(recs[:, :6].cpu(), recs[:, 6].cpu(), orecs[:, :6], orecs[:, 6], allsc, label="4-chunk scene records")
    assert recs.shape[0] == orecs.shape[0] and torch.equal(keep.cpu(), okeep)          # 0 near-ties asserted above
    got, want = recs[keep][:, :6].cpu(), orecs[okeep][:, :6]
    d = (got[None] - want[:, None]).abs().amax(-1)
    # boxes were shifted to scene coordinates
    assert float(recs[:, 3].max()) > dims[0] and float(recs[:, 5].max()) > dims[2]
    # class-regressed final boxes of the matched detections agree with the oracle's host-side decode
    m = d.argmin(1)
    ok = d.min(1).values <= 1e-3
    fb_got, fb_want = recs[keep][:, 10:16].cpu()[m[ok]], orecs[okeep][:, 10:16][ok]
    assert float((fb_got - fb_want).abs().max()) <= 5e-3

    # ---- instance masks of the survivors, computed on the chunk that produced each detection
    recs3, keep3, masks = runner.infer(chunks, with_masks=True)
    assert torch.equal(recs3, recs) and torch.equal(keep3, keep)
    kept = recs[keep].cpu()
    assert 0 < len(masks) <= kept.shape[0]
    # mask VALUES: the predicted class's sigmoid channel <= 1e-4 vs the oracle on the same crop; the binary mask is that channel
    # thresholded; voxels that flip at MASK_THRESH are counted and must sit within 1e-4 of it
    _, _, probs = runner.infer(chunks, with_masks=True, mask_values=True)
    assert probs.keys() == masks.keys()
    flips = 0
    for i, (w, mk) in masks.items():
        r = kept[i]
        assert r[9] > cfg.CLASS_THRESH and tuple(mk.shape) == (w[3] - w[0], w[4] - w[1], w[5] - w[2])
        c = [c for c in chunks if c[1][0] <= w[0] < c[1][0] + dims[0] and c[1][2] <= w[2] < c[1][2] + dims[2]][0]
        ox, oz = int(c[1][0]), int(c[1][2])
        crop = c[2][0:1, :, w[0] - ox:w[3] - ox, w[1]:w[4], w[2] - oz:w[5] - oz]
        want_p = on.mask_backbone(crop)[0, int(r[8])]
        pw, pr = probs[i]
        assert pw == w and float((pr.cpu() - want_p).abs().max()) <= 1e-4
        assert torch.equal(mk, (pr >= cfg.MASK_THRESH).float())
        diff = mk.cpu() != (want_p >= cfg.MASK_THRESH).float()
        flips += int(diff.sum())
        assert bool(((want_p[diff] - cfg.MASK_THRESH).abs() <= 1e-4).all())
    from parity import report
    report("4-chunk scene masks: %d masks <= 1e-4 on the sigmoid outputs, %d voxels flip at MASK_THRESH" % (len(masks), flips))


def test_engine_from_depth_maps_equals_loaded_lists(oracle):
    """ChunkEngine(from_depth=True): lists computed inside the captured graph == lists loaded from the oracle's
    compute_projection; same RPN maps bit for bit, and per-view counts are reported."""
    from sis3d import config, synthetic
    from sis3d.engine import ChunkEngine
    from sis3d.nets import backbones
    dims = (64, 32, 48)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = True
    cfg.USE_MASK = False
    net = getattr(backbones, cfg.NET)(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    net = net.cuda().eval()
    V = 3
    data = synthetic.synth_chunk(2, dims)
    feats = torch.randn(V, 128, 32, 41, generator=torch.Generator().manual_seed(3))
    depth, c2w, w2g = synthetic.synth_cameras(41, V, dims, cfg.VOXEL_SIZE)
    maps = [oracle.compute_projection(depth[v], c2w[v], w2g[v], cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX,
                                      cfg.DEPTH_SHAPE, dims, cfg.VOXEL_SIZE) for v in range(V)]
    assert all(m is not None for m in maps)
    i3d, i2d = torch.stack([m[0] for m in maps]), torch.stack([m[1] for m in maps])
    a = ChunkEngine(net, dims=dims, stage="rpn", n_views=V, from_depth=True)
    a.load_views(data, feats, depth, c2w, w2g)
    a.prepare()
    oa = {k: v.clone() for k, v in a.run().items()}
    assert a.view_counts() == [int(m[0][0]) for m in maps]
    assert torch.equal(a.i3d.cpu(), i3d) and torch.equal(a.i2d.cpu(), i2d)
    b = ChunkEngine(net, dims=dims, stage="rpn", n_views=V)
    b.load(data, feats, i3d, i2d)
    b.prepare()
    ob = b.run()
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k


@pytest.mark.parametrize("stage", ["rpn", "detect"])
def test_grouped_engine_equals_single_chunk_engines(stage):
    """ChunkEngine(group=2): two chunks per captured graph sharing ONE batched launch of their four RPN convs ==
    two single-chunk engines, bit for bit"""
    from sis3d import config, synthetic
    from sis3d.engine import ChunkEngine
    from sis3d.nets import backbones
    dims = (64, 32, 48)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    net = getattr(backbones, cfg.NET)(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    net = net.cuda().eval()
    chunks = [synthetic.synth_chunk(c, dims) for c in (3, 4)]
    pair = ChunkEngine(net, dims=dims, stage=stage, group=2)
    for g, c in enumerate(chunks):
        pair.load(c, slot=g)
        pair.set_origin((10.0 * g, 0.0, 5.0), slot=g)
    pair.prepare()
    got = [{k: v.clone() for k, v in d.items() if torch.is_tensor(v)} for d in pair.run()]
    for g, c in enumerate(chunks):
        one = ChunkEngine(net, dims=dims, stage=stage)
        one.load(c)
        one.set_origin((10.0 * g, 0.0, 5.0))
        one.prepare()
        want = one.run()
        keys = [k for k in got[g] if k in want]
        assert keys and ("block" in keys or stage == "rpn")
        for k in keys:
            assert torch.equal(got[g][k], want[k]), (g, k)


@pytest.mark.parametrize("stride", [96, 80])
def test_config5_scene_32_chunks_vs_oracle(oracle, stride):
    """BASELINE config 5 at its real size: 32 chunks of 96x48x96 laid out 4 x 1 x 8, per-chunk captured graph -> record
    blocks -> gather -> whole-scene NMS, against the oracle running the same chunks on the CPU + cpu_nms over the
    concatenation.  stride 96: tiles edge to edge (nothing crosses a border, the NMS keeps every
    record).  stride 80: neighbouring chunks OVERLAP by 16 voxels in x and z (what `bench.py`'s scene workload times),
    so detections of adjacent chunks do overlap and the whole-scene NMS -- the step the all-gather exists for --
    really suppresses: kept < gathered, and the kept SET must be the oracle's."""
    from parity import assert_proposals_match
    from sis3d import parallel
    from sis3d.engine import RECORD_WIDTH
    from sis3d.nets import backbones
    from sis3d.scene import SceneRunner
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    sd = synthetic.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    net.cuda().eval()
    dims = synthetic.CHUNK_DIMS
    chunks = [(c, (float(stride) * (c % 4), 0.0, float(stride) * (c // 4)), synthetic.synth_chunk(c)) for c in range(32)]
    runner = SceneRunner(net, dims)
    recs, keep = runner.infer([(c, o, d.cuda()) for c, o, d in chunks])
    on = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    k = cfg.TEST.RPN_POST_NMS_TOP_N
    allsc = []

    def odetect(data):
        o = on.forward(data)
        allsc.append(o["_scores_sorted_all"])
        n = o["rois"][0].shape[0]
        rec = torch.zeros(k, RECORD_WIDTH)
        rec[:n, :6], rec[:n, 6], rec[:n, 7] = o["rois"][0], o["roi_scores"][0][:, 0], o["level_inds"][0]
        rec[:n, 8] = o["cls_pred"].float()
        rec[:n, 9] = o["cls_prob"].gather(1, o["cls_pred"].view(-1, 1))[:, 0]
        rec[:n, 10:16] = torch.from_numpy(oracle.class_boxes(o, data.shape[2:]))
        return rec, n
    orecs, okeep = parallel.infer_scene(chunks, odetect, oracle.nms, k, cfg.TEST.RPN_NMS_THRESH)
    assert float(recs[:, 3].max()) > 3 * stride and float(recs[:, 5].max()) > 7 * stride   # scene coordinates of the 4 x 1 x 8 grid
    assert_proposals_match(recs[:, :6].cpu(), recs[:, 6].cpu(), orecs[:, :6], orecs[:, 6],
                           torch.cat(allsc).sort(descending=True).values, label="config 5: 32-chunk scene records, stride %d" % stride)
    # the whole-scene NMS itself is integer-exact on the device's own records
    assert torch.equal(keep.cpu(), oracle.nms(recs[:, :6].cpu().contiguous(), cfg.TEST.RPN_NMS_THRESH))
    # same records (the score-sorted ORDER may swap rows whose scores differ by ~1e-7 between the two fp32 pipelines):
    # level, class, class probability and the class-regressed final box of every matched record
    from parity import match_sets
    assert recs.shape[0] == orecs.shape[0] and keep.numel() == okeep.numel()
    pairs, uw, ug = match_sets(recs[:, :6].cpu(), orecs[:, :6])
    assert not uw and not ug
    iw = torch.tensor([i for i, _ in pairs])
    jg = torch.tensor([j for _, j in pairs])
    r, w = recs.cpu()[jg], orecs[iw]
    assert torch.equal(r[:, 7], w[:, 7]) and torch.equal(r[:, 8], w[:, 8])
    assert float((r[:, 9] - w[:, 9]).abs().max()) <= 1e-4 and float((r[:, 10:16] - w[:, 10:16]).abs().max()) <= 5e-3
    s_ = recs[:, 6]
    assert bool((s_[:-1] >= s_[1:]).all())
    # the survivors of the whole-scene NMS are the oracle's survivors, one to one
    kp, kuw, kug = match_sets(recs[keep][:, :6].cpu(), orecs[okeep][:, :6])
    assert not kuw and not kug and len(kp) == keep.numel()
    from parity import report
    report("config 5 stride %d: %d records gathered, %d kept after the whole-scene NMS (oracle: %d / %d), kept sets equal" % (
        stride, recs.shape[0], keep.numel(), orecs.shape[0], okeep.numel()))
    if stride < 96:
        assert keep.numel() < recs.shape[0]                   # suppression happened ...
        # ... and it is CROSS-chunk: a chunk's own records already survived its per-chunk NMS at the same threshold, so every
        # suppressed record must have a kept, higher-ranked suppressor (IoU > thresh) that another chunk produced
        from sis3d import ops
        from sis3d.scene import fused_merge
        blocks = parallel.gather_blocks(runner.run_chunks([(c, o, d.cuda()) for c, o, d in chunks]), 32, k)
        recs2, keep2, cids = parallel.merge_scene(blocks, k, ops.nms, cfg.TEST.RPN_NMS_THRESH, with_chunk_ids=True, merge_fn=fused_merge)
        assert torch.equal(recs2, recs) and torch.equal(keep2, keep)
        cids, kept = cids.cpu(), torch.zeros(recs.shape[0], dtype=torch.bool)
        kept[keep.cpu()] = True
        boxes = recs[:, :6].cpu()
        x0, x1 = torch.maximum(boxes[:, None, :3], boxes[None, :, :3]), torch.minimum(boxes[:, None, 3:], boxes[None, :, 3:])
        inter = (x1 - x0 + 1).clamp(min=0).prod(-1)
        vol = (boxes[:, 3:] - boxes[:, :3] + 1).prod(-1)
        over = inter / (vol[:, None] + vol[None, :] - inter) > cfg.TEST.RPN_NMS_THRESH          # (i, j): j overlaps i
        n_cross = 0
        for i in (~kept).nonzero().view(-1).tolist():
            sup = (over[i, :i] & kept[:i]).nonzero().view(-1)
            assert sup.numel() > 0 and bool((cids[sup] != cids[i]).all())
            n_cross += 1
        assert n_cross == recs.shape[0] - keep.numel() > 0
        report("config 5 stride %d: %d suppressed records, each by a kept box of ANOTHER chunk" % (stride, n_cross))


@pytest.mark.parametrize("n_chunks,k_rows", [(1, 200), (4, 200), (32, 200), (40, 200), (7, 33)])
def test_fused_scene_merge_equals_torch_path(oracle, n_chunks, k_rows):
    """sis3d_scene_merge (sort + gather + whole-scene NMS, one launch sequence) against parallel.merge_scene's torch code on
    random record tables: ragged counts incl. empty and full chunks, tied scores across chunks, boxes that overlap across
    chunk borders; both box / score column choices; keep list also against the CPU oracle's NMS."""
    from sis3d import ops, parallel
    from sis3d.engine import RECORD_WIDTH as W
    g = torch.Generator().manual_seed(1000 * n_chunks + k_rows)
    bf = 1 + k_rows * W
    blocks = torch.zeros(n_chunks, bf)
    counts = torch.randint(0, k_rows + 1, (n_chunks,), generator=g)
    counts[0] = k_rows
    if n_chunks > 2:
        counts[1] = 0
    for c in range(n_chunks):
        n = int(counts[c])
        rows = torch.zeros(k_rows, W)
        lo = torch.rand(n, 3, generator=g) * torch.tensor([96.0, 40.0, 96.0]) + torch.tensor([90.0 * (c % 4), 0.0, 90.0 * (c // 4)])
        rows[:n, 0:3] = lo
        rows[:n, 3:6] = lo + torch.rand(n, 3, generator=g) * 30 + 1
        rows[:n, 6] = (torch.rand(n, generator=g) * 50).round() / 50            # many exact ties, within and across chunks
        rows[:n, 9] = torch.rand(n, generator=g)
        rows[:n, 10:16] = rows[:n, 0:6] + torch.rand(n, 6, generator=g)
        rows[:n, 8] = torch.randint(1, 19, (n,), generator=g).float()
        rows[n:, 6] = 7.0                                                        # garbage past the count must be ignored
        blocks[c, 0] = n
        blocks[c, 1:] = rows.reshape(-1)
    dev = blocks.cuda()
    for score_col, box_cols in ((6, (0, 6)), (9, (10, 16))):
        for th, mk in ((0.1, 0), (0.3, 17)):
            want = parallel.merge_scene(dev, k_rows, ops.nms, th, score_col=score_col, max_keep=mk, with_chunk_ids=True, box_cols=box_cols)
            if n_chunks * k_rows > 8192:
                from sis3d.scene import fused_merge
                assert fused_merge(dev, k_rows, th, score_col, box_cols[0], mk) is None
                continue
            got = ops.scene_merge(dev, k_rows, th, score_col, box_cols[0], mk)
            for a, b in zip(got, want):
                assert a.dtype == b.dtype and torch.equal(a, b)
            okeep = oracle.nms(want[0][:, box_cols[0]:box_cols[1]].cpu().contiguous(), th)
            assert torch.equal(got[1].cpu(), okeep[:mk] if mk else okeep)
    empty = torch.zeros(3, bf).cuda()
    r, k, c = ops.scene_merge(empty, k_rows, 0.1)
    assert r.shape == (0, W) and k.numel() == 0 and c.numel() == 0


# ---- r4: one graph launch per scene share, lazy lengths, the full-table merge of a rank's share, and the RCCL branch
def _small_net():
    from sis3d.nets import backbones
    cfg = config.scannet_benchmark_cfg()
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS))
    return net.cuda().eval(), cfg


def test_round_graph_lazy_result_and_emulated_rank_share():
    """(a) a share of exactly one chunk per pipeline goes out as ONE graph launch (PipelinedEngines.capture_round) and gives the
    records of the per-chunk replays bit for bit, scene after scene; (b) a lazy SceneResult resolves to the same tensors;
    (c) SceneRunner(emulate=(r, W)) + infer(gathered=table): rank r's own chunks + the merge of the FULL gathered table equals the
    whole-scene result, for every r -- what bench.py times as share_of_one_rank_at_8."""
    from sis3d import parallel
    from sis3d.scene import SceneRunner, SceneResult
    net, cfg = _small_net()
    dims = (48, 24, 40)
    chunks = [(c, (40.0 * (c % 2), 0.0, 32.0 * (c // 2)), synthetic.synth_chunk(20 + c, dims).cuda()) for c in range(4)]   # overlapping
    base = SceneRunner(net, dims, inflight=3)                    # 4 chunks on 3 pipelines: the per-chunk replays
    recs, keep = base.infer(chunks)
    assert recs.shape[0] > 0 and keep.numel() < recs.shape[0]    # the whole-scene NMS really suppresses across chunks
    rnd = SceneRunner(net, dims, inflight=4, round_graph=True)
    for _ in range(3):
        r2, k2 = rnd.infer(chunks)
        assert rnd._round is not None
        assert torch.equal(r2, recs) and torch.equal(k2, keep)
    lz = [rnd.infer(chunks, lazy=True) for _ in range(3)]         # three scenes enqueued before the first length is read
    assert all(isinstance(x, SceneResult) for x in lz)
    for x in lz:
        r3, k3 = x.resolve()
        assert torch.equal(r3, recs) and torch.equal(k3, keep)
    # (c) the gathered table of the whole scene, then each emulated rank of a world of 2 on its own two chunks
    table = parallel.gather_blocks(base.run_chunks(chunks), 4, base.k_rows, solo=True).clone()
    for r in range(2):
        share = SceneRunner(net, dims, inflight=2, emulate=(r, 2), round_graph=True)
        stale = table.clone()
        stale[r::2] = 0                                          # this rank's rows must come from its own fresh pass
        only_mine = [(c, o, p if c % 2 == r else None) for c, o, p in chunks]
        for lazy in (False, True):
            out = share.infer(only_mine, gathered=stale, lazy=lazy)
            r4, k4 = out.resolve() if lazy else out
            assert torch.equal(r4, recs) and torch.equal(k4, keep)
        assert share._round is not None and float(stale[r::2].abs().max()) == 0.0     # the caller's table is not written


def test_rccl_branch_of_gather_blocks_on_one_gpu(tmp_path):
    """parallel.gather_blocks' `nccl` branch (one all_gather_into_tensor per scene) executed on real hardware: a world of ONE under
    RCCL with parallel.FORCE_COLLECTIVE, in a subprocess (a process group is process-wide state), against the same scene without
    a process group.  N > 1 needs an N-GPU node (the driver's); this pins the collective's shapes / ordering / stream semantics."""
    import os
    import subprocess
    import sys
    script = tmp_path / "rccl_world1.py"
    script.write_text('''
import os, sys, torch
import torch.distributed as dist
sys.path[:0] = %r
from sis3d import config, synthetic, parallel
from sis3d.nets import backbones
from sis3d.scene import SceneRunner
cfg = config.scannet_benchmark_cfg()
net = backbones.ScanNet_Backbone(cfg=cfg); net.init_modules()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict(synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)); net.cuda().eval()
dims = (48, 24, 40)
chunks = [(c, (40.0 * (c %% 2), 0.0, 32.0 * (c // 2)), synthetic.synth_chunk(20 + c, dims).cuda()) for c in range(4)]
runner = SceneRunner(net, dims, inflight=4)
want_r, want_k = runner.infer(chunks)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
parallel.FORCE_COLLECTIVE = True
calls = []
real = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(tuple(a[0].shape)), real(*a, **k))[1]
for _ in range(3):
    r, k = runner.infer(chunks)
    assert torch.equal(r, want_r) and torch.equal(k, want_k)
assert calls == [(1, 4, parallel.block_floats(runner.k_rows))] * 3, calls
# r5: the PIPELINED path (infer(lazy=True)): join, the collective and the merge are enqueued on the merge stream, three scenes in
# flight before the first length is read -- the same all_gather_into_tensor, on a stream that is not the caller's
lz = [runner.infer(chunks, lazy=True) for _ in range(3)]
for x in lz:
    r, k = x.resolve()
    assert torch.equal(r, want_r) and torch.equal(k, want_k)
assert len(calls) == 6 and calls[-1] == (1, 4, parallel.block_floats(runner.k_rows)), calls
host = [(c, o, p.cpu().contiguous().pin_memory()) for c, o, p in chunks]
r, k = runner.infer(host, lazy=True).resolve()
assert torch.equal(r, want_r) and torch.equal(k, want_k)
torch.cuda.synchronize(); dist.destroy_process_group()
print("RCCL_WORLD1_OK", len(calls), tuple(want_r.shape), int(want_k.numel()))
''' % ([p for p in sys.path if p],))
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK 7" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    from parity import report
    report("RCCL world-of-one: gather_blocks' nccl branch ran 3 eager + 3 pipelined + 1 host-fed scenes, %s" % p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("copy", ["mailbox", "kernel", "own", "per_pipeline"])
def test_streamed_inputs_equal_resident_inputs(copy):
    """r5 (the reference's forward owns the upload, lib/nets/network.py:191): chunks that arrive in PINNED HOST
    memory -- as the encoded grid or as the raw SDF block of a .chunk file -- give the outputs of the same chunks loaded into the
    static buffers, bit for bit, for every upload method: 'mailbox' (the default: engines captured with the upload as the first
    node of their graph, the host pointer travels in a ring of pinned slots, ONE call per chunk) and the eager variants of
    PipelinedEngines.feed / run_fed (upload kernel / hipMemcpyAsync on the pipeline's stream, copy stream per pipeline)."""
    from sis3d.engine import PipelinedEngines
    from sis3d import ops
    net, cfg = _small_net()
    dims = (48, 24, 40)
    n = 3
    ref = PipelinedEngines(net, n, dims=dims, stage="rpn").prepare()
    ids = [[30 + 10 * i + r for r in range(3)] for i in range(n)]
    want = {}
    for i in range(n):
        for cid in ids[i]:
            ref.load(i, synthetic.synth_chunk(cid, dims))
            out = ref.run(i)
            with torch.cuda.stream(ref.streams[i]):                  # the clone must sit behind the replay, on the pipeline's stream
                want[cid] = {k: v.clone() for k, v in out.items()}
            torch.cuda.synchronize()
    for mode in ("grid", "sdf"):
        host = {cid: (synthetic.synth_chunk(cid, dims) if mode == "grid" else synthetic.synth_sdf(cid, dims)).contiguous().pin_memory()
                for row in ids for cid in row}
        if copy == "mailbox":
            eng = PipelinedEngines(net, n, dims=dims, stage="rpn", mailbox=True, mail_input=mode).prepare()
            eng.enable_feed(mode)
        else:
            eng = ref
            eng.enable_feed(mode, copy=copy)
        for r in range(3):
            got = {}
            for i in range(n):
                o = eng.run_fed(i, host[ids[i][r]])
                with torch.cuda.stream(eng.streams[i]):              # outputs copied out behind the replay, before the look-ahead upload
                    got[i] = {k: v.clone() for k, v in o.items()}
                if copy not in ("own",) and r + 1 < 3:
                    assert eng.feed(i, host[ids[i][r + 1]])          # one chunk ahead
            torch.cuda.synchronize()
            for i in range(n):
                for k, v in want[ids[i][r]].items():
                    assert torch.equal(got[i][k], v), (mode, copy, i, r, k)
        if copy != "mailbox":
            for i in range(n):
                while eng.pending(i):
                    eng.consume(i, eng.streams[i])
        torch.cuda.synchronize()
        if copy == "mailbox":
            mb = eng.engines[0].mail
            assert mb.head == int(mb.progress[0]) and mb.head >= 3    # every slot written was consumed by exactly one pass
    with pytest.raises(ops._lib.Sis3dError):
        eng.feed(0, synthetic.synth_chunk(1, dims))                  # pageable host memory is refused
    print("[parity] streamed inputs (%s): run_fed grid + sdf bit-identical to loaded chunks" % copy)


@pytest.mark.parametrize("mailbox", ["1", "0"])
def test_scene_from_pinned_host_chunks_equals_resident_scene(mailbox, monkeypatch):
    """a whole scene whose chunks sit in PINNED HOST memory (per-chunk launches; one chunk per pipeline = the one-launch round;
    lazy / pipelined results) gives the resident scene's records bit for bit -- through the mailbox engines (default) and through
    engines without a mailbox (SIS3D_SCENE_MAILBOX=0: eager copies, the r4 path); pageable host chunks take the ordinary copy"""
    from sis3d.scene import SceneRunner
    monkeypatch.setenv("SIS3D_SCENE_MAILBOX", mailbox)
    net, cfg = _small_net()
    dims = (48, 24, 40)
    res_chunks = [(c, (40.0 * (c % 2), 0.0, 32.0 * (c // 2)), synthetic.synth_chunk(20 + c, dims).cuda()) for c in range(8)]
    host_chunks = [(c, o, p.cpu().contiguous().pin_memory()) for c, o, p in res_chunks]
    pageable = [(c, o, p.cpu()) for c, o, p in res_chunks]
    for nfl, nch in ((3, 8), (4, 4)):                                # per-chunk launches; one chunk per pipeline (round graph)
        runner = SceneRunner(net, dims, inflight=nfl, round_graph=(nch == nfl))
        assert (runner.pipes.engines[0].mail is not None) == (mailbox == "1")
        recs, keep = runner.infer(res_chunks[:nch])
        for chunks in (host_chunks, pageable):
            for _ in range(2):
                r2, k2 = runner.infer(chunks[:nch])
                assert torch.equal(r2, recs) and torch.equal(k2, keep), (mailbox, nfl)
        lz = [runner.infer(host_chunks[:nch], lazy=True) for _ in range(3)]
        for x in lz:
            r3, k3 = x.resolve()
            assert torch.equal(r3, recs) and torch.equal(k3, keep), (mailbox, nfl, "lazy")
        torch.cuda.synchronize()
    # without captured graphs (use_graph=False) the same launches go out eagerly -- the mailbox kernels included -- same records
    eager = SceneRunner(net, dims, inflight=2, use_graph=False)
    recs8, keep8 = SceneRunner(net, dims, inflight=3).infer(res_chunks)
    for chunks in (res_chunks, host_chunks):
        r5, k5 = eager.infer(chunks)
        assert torch.equal(r5, recs8) and torch.equal(k5, keep8), (mailbox, "eager")
    print("[parity] host-fed scenes (mailbox %s): per-chunk, round graph, pipelined and eager results bit-identical to the resident scene" % mailbox)


def test_mailbox_ring_wraps_and_never_laps_the_device():
    """a mailbox of FOUR slots, 30 passes submitted back to back without a host sync: the producer waits for a free slot instead of
    overwriting one the device has not read (progress word), every pass sees ITS chunk (device or pinned host source, origin, block
    row), resident passes (src=None) leave the input alone"""
    from sis3d.engine import ChunkEngine
    net, cfg = _small_net()
    dims = (48, 24, 40)
    plain = ChunkEngine(net, dims=dims, stage="detect")
    mbx = ChunkEngine(net, dims=dims, stage="detect", mailbox=True, mail_ring=4)
    cids = [60 + k for k in range(5)]
    grids = {c: synthetic.synth_chunk(c, dims) for c in cids}
    plain.prepare()
    want = {}
    for c in cids:
        plain.load(grids[c])
        plain.set_origin((float(c), 0.0, 2.0 * c))
        want[c] = plain.run()["block"].clone()
    torch.cuda.synchronize()
    mbx.prepare()
    srcs = {c: (grids[c].contiguous().pin_memory() if c % 2 else grids[c].cuda()) for c in cids}
    rows = torch.zeros(30, want[cids[0]].numel(), device="cuda")
    for k in range(30):
        c = cids[k % 5]
        mbx.submit(src=srcs[c], block_dst=rows[k], origin=(float(c), 0.0, 2.0 * c))
    torch.cuda.synchronize()
    for k in range(30):
        assert torch.equal(rows[k], want[cids[k % 5]]), k
    mb = mbx.mail
    assert mb.ring_size == 4 and mb.head == int(mb.progress[0]) >= 30
    # a resident pass: no source in the slot, the static input keeps the last chunk; no destination: nothing is written
    before = rows.clone()
    out = mbx.submit(src=None, block_dst=None, origin=(float(cids[4]), 0.0, 2.0 * cids[4]))
    torch.cuda.synchronize()
    assert torch.equal(out["block"], want[cids[4]]) and torch.equal(rows, before)


def test_stream_window_calibration_keeps_results():
    """PipelinedEngines.calibrate / SceneRunner.calibrate move the pipelines onto other streams of the pool: same graphs, same results"""
    from sis3d.scene import SceneRunner
    net, cfg = _small_net()
    dims = (48, 24, 40)
    chunks = [(c, (40.0 * (c % 2), 0.0, 32.0 * (c // 2)), synthetic.synth_chunk(20 + c, dims).cuda()) for c in range(4)]
    runner = SceneRunner(net, dims, inflight=4)
    recs, keep = runner.infer(chunks)
    info = runner.calibrate(chunks, reps=1, count=8)
    assert "stream_window" in info and len(info["ms_per_scene_by_window"]) >= 2 and "one_launch_round" in info
    assert len({s.cuda_stream for s in runner.pipes.streams}) == 4
    for lazy in (False, True):
        out = runner.infer(chunks, lazy=lazy)
        r2, k2 = out.resolve() if lazy else out
        assert torch.equal(r2, recs) and torch.equal(k2, keep)


@pytest.mark.parametrize("full", [False, True])
def test_stage_ahead_pulls_the_next_chunk_and_changes_nothing(full, monkeypatch):
    """r5: a mailbox slot may name the pipeline's NEXT pinned host chunk; one more row of workgroups of the pass's longest Winograd launch
    (sis3d_conv3d_k3wino_piggyback) pulls it into a staging buffer and the next pass's upload node copies it from there.  Results are those of
    resident passes bit for bit -- when the announcement is right, when the caller then submits ANOTHER chunk, when the chunk comes
    from device memory -- and the staged copy really is what the next pass reads (white box: the host block is changed after the
    staging pass, against the contract, and the pass still sees the staged contents)."""
    from sis3d import ops
    from sis3d.engine import ChunkEngine
    net, cfg = _small_net()
    dims, stage = ((96, 48, 96), "rpn") if full else ((48, 24, 40), "detect")
    if not full:
        monkeypatch.setattr(ops, "PIGGY_MIN_FLOPS", 0.0)      # small chunk: whichever Winograd launch comes first carries the upload
    key = "rpn_bbox_pred_level1" if stage == "rpn" else "block"
    plain = ChunkEngine(net, dims=dims, stage=stage)
    plain.prepare()
    cids = [70 + k for k in range(5)]
    grids = {c: synthetic.synth_chunk(c, dims) for c in cids}
    want = {}
    for c in cids:
        plain.load(grids[c])
        want[c] = plain.run()[key].clone()
    torch.cuda.synchronize()
    assert not torch.equal(want[cids[0]], want[cids[1]])
    host = {c: grids[c].contiguous().pin_memory() for c in cids}
    for use_graph in (True, False):
        mbx = ChunkEngine(net, dims=dims, stage=stage, mailbox=True, use_graph=use_graph)
        off = ChunkEngine(net, dims=dims, stage=stage, mailbox=True, use_graph=use_graph, stage_ahead=False)
        mbx.prepare()
        off.prepare()
        assert off._ahead is None and mbx._ahead is not None and not off.piggybacked
        if full:
            assert mbx.piggybacked, "the rpn_net launch of the full-size chunk must carry the upload"

        def one(eng, c, nxt):
            out = eng.submit(src=c if isinstance(c, torch.Tensor) else host[c], next_src=None if nxt is None else host[nxt])[key]
            torch.cuda.synchronize()
            return out.clone()
        # right announcements, back to back
        for k, c in enumerate(cids):
            nxt = cids[k + 1] if k + 1 < len(cids) else None
            assert torch.equal(one(mbx, c, nxt), want[c]), (use_graph, "announced", k)
            assert torch.equal(one(off, c, nxt), want[c]), (use_graph, "stage_ahead off", k)
        # announced cids[1], submitted cids[3]; then a device source; then the announced chunk after an unrelated pass
        assert torch.equal(one(mbx, cids[0], cids[1]), want[cids[0]])
        assert torch.equal(one(mbx, cids[3], cids[2]), want[cids[3]]), (use_graph, "another chunk than the announced one")
        assert torch.equal(one(mbx, grids[cids[2]].cuda(), None), want[cids[2]]), (use_graph, "device source")
        assert torch.equal(one(mbx, cids[2], None), want[cids[2]])
        if mbx.piggybacked:
            # white box: the pass below stages a copy of cids[4]; the host block then changes (a caller must not do this) and the next
            # pass still computes on the staged copy -- the link is not crossed again
            spare = host[cids[4]].clone().pin_memory()
            host["spare"] = spare
            one(mbx, cids[0], "spare")
            spare.copy_(host[cids[1]])
            assert torch.equal(one(mbx, spare, None), want[cids[4]]), (use_graph, "the staged copy is what the pass reads")
            # unannounced, the same block is read from the host again
            assert torch.equal(one(mbx, spare, None), want[cids[1]])
        # back-to-back passes without a host sync in between (the branch of pass k and the upload node of pass k + 1 are ordered)
        rows = []
        for k in range(12):
            c, nxt = cids[k % 5], cids[(k + 1) % 5]
            out = mbx.submit(src=host[c], next_src=host[nxt])[key]
            rows.append(out.clone())                  # on the stream of the pass: ordered behind it
        torch.cuda.synchronize()
        for k in range(12):
            assert torch.equal(rows[k], want[cids[k % 5]]), (use_graph, "back to back", k)
    print("[parity] stage-ahead upload (full=%s): staged / mis-announced / device / unannounced chunks all bit-identical to resident passes" % full)


def test_mailbox_rejects_stale_and_torn_slots_and_validates_sources():
    """r6 hardening: the fetch kernel accepts a slot only with the right sequence number and check
    word -- a replay without a fresh slot (stale) or a slot with a flipped pointer bit (torn) copies NOTHING and the producer raises;
    submit() refuses sources the upload kernel would read out of bounds; consumed slots release their tensors"""
    import pytest as _pytest
    from sis3d import _lib
    from sis3d.engine import ChunkEngine
    net, cfg = _small_net()
    dims = (48, 24, 40)
    eng = ChunkEngine(net, dims=dims, stage="detect", mailbox=True, mail_ring=8).prepare()
    g = synthetic.synth_chunk(7, dims).cuda()
    row = torch.zeros(eng.out["block"].numel(), device="cuda")
    eng.submit(src=g, block_dst=row, origin=(1.0, 2.0, 3.0))
    torch.cuda.synchronize()
    good = row.clone()
    assert good.abs().sum() > 0
    mb = eng.mail
    assert int(mb.progress[1]) == 0
    # consumed slots do not pin their tensors (all but the newest slot of the ring)
    assert sum(k is not None for k in mb.keep) <= 1
    # sources the upload kernel must not be handed
    for bad in (g.view(-1)[:-4], g.double(), g.permute(0, 1, 4, 3, 2)):
        with _pytest.raises(_lib.Sis3dError):
            eng.submit(src=bad, block_dst=row)
    # STALE: the graph replayed without a new slot -> the fetch sees the previous slot's sequence number
    row.zero_()
    eng.graph.replay()
    torch.cuda.synchronize()
    assert int(mb.progress[1]) == 1 and float(row.abs().sum()) == 0.0, "a stale slot must not be acted on"
    with _pytest.raises(_lib.Sis3dError):
        mb.check()
    # TORN: a slot whose destination pointer lost a bit after the stamp was computed
    eng2 = ChunkEngine(net, dims=dims, stage="detect", mailbox=True, mail_ring=8).prepare()
    mb2 = eng2.mail
    row2 = torch.zeros_like(row)
    mb2.write(src=g, dst=row2, origin=(1.0, 2.0, 3.0))
    k = (mb2.head - 1) % mb2.ring_size
    mb2.u64[k, 1] ^= 0x40                                       # plain CPU store into the pinned ring, as a racing writer would
    eng2.graph.replay()
    torch.cuda.synchronize()
    assert int(mb2.progress[1]) == 2 and float(row2.abs().sum()) == 0.0
    with _pytest.raises(_lib.Sis3dError):
        mb2.write(src=g, dst=row2)
    print("[parity] mailbox: stale slot -> code 1, torn slot -> code 2, nothing copied; submit() rejects 3 malformed sources")


def test_two_threads_prepare_engines_on_one_gpu():
    """prepare() warms up and captures on the device's shared capture stream -- two threads doing it at once used to
    record into each other's graph.  Serialised by engine.device_lock: both engines capture, both replay their own chunk."""
    import threading
    from sis3d.engine import ChunkEngine
    dims = (48, 24, 40)
    nets = [_small_net()[0], _small_net()[0]]
    engs = [ChunkEngine(nets[i], dims=dims, stage="detect", shared_chip=bool(i)) for i in range(2)]
    grids = [synthetic.synth_chunk(70 + i, dims).cuda() for i in range(2)]
    for e, g in zip(engs, grids):
        e.load(g)
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(2)

    def worker(i):
        try:
            start.wait()
            engs[i].prepare()
        except Exception as ex:                                 # pragma: no cover
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert all(e.graph is not None for e in engs)
    blocks = [e.run()["block"].clone() for e in engs]
    torch.cuda.synchronize()
    # each graph is the engine's own pass: the same chunk through a freshly prepared engine of the same regime gives the same block
    for i in range(2):
        ref = ChunkEngine(nets[i], dims=dims, stage="detect", shared_chip=bool(i))
        ref.load(grids[i])
        ref.prepare()
        assert torch.equal(ref.run()["block"], blocks[i]), i
    assert not torch.equal(blocks[0], blocks[1])


def test_pipelines_sit_on_verified_distinct_hardware_queues():
    """r6: the pipelines' streams are created by the engine (hipStreamCreateWithPriority -> ExternalStream) and
    their hardware queues PROBED: four pipelines get four streams of pairwise distinct queue classes, none of them torch's pool
    streams, without any timing calibration; the side roles take other own streams; results do not depend on the placement"""
    from sis3d import engine
    from sis3d.engine import PipelinedEngines
    o = engine.own_streams()
    assert len(o["streams"]) == engine.OWN_STREAMS == len(o["klass"])
    n_classes = len(set(o["klass"]))
    net, cfg = _small_net()
    dims = (48, 24, 40)
    pe = PipelinedEngines(net, 4, dims=dims, stage="detect")
    if n_classes < 4:
        # a process the runtime gives fewer than four hardware queues (GPU_MAX_HW_QUEUES < 4, or a probe that read shared queues):
        # the engine must SAY so -- prepare() then falls back to the timing calibration -- and this test has nothing to verify
        assert not pe.placement_verified
        pytest.skip("fewer than four distinct hardware queues in this process: %s (null %s)" % (o["klass"], o["null_class"]))
    assert pe.placement_verified
    idx = [o["streams"].index(s) for s in pe.streams]
    assert len({o["klass"][i] for i in idx}) == 4
    if n_classes > 4:
        assert all(o["klass"][i] != o["null_class"] for i in idx)      # the null stream's queue is taken last
    for i in range(4):
        pe.load(i, synthetic.synth_chunk(30 + i, dims))
    pe.prepare()
    assert not pe.stream_window_times, "no timing calibration when the placement is verified"
    outs = pe.run()
    torch.cuda.synchronize()                                              # the static outputs are written on the pipelines' streams
    a = [d["block"].clone() for d in outs]
    best, times = pe.calibrate(pe.run, reps=1, warm=1)                    # the fallback still works and changes nothing
    outs = pe.run()
    torch.cuda.synchronize()
    b = [d["block"].clone() for d in outs]
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and len(times) >= 2
    side = engine.pooled_stream("capture", 0)
    assert side in o["streams"]
