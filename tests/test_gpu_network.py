"""End-to-end parity of the HIP forward (sis3d.nets mirror of lib/nets) against the oracle and the
reference-generated golden fixtures, on the same seeded inputs and weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, synthetic  # noqa: E402
from parity import assert_proposals_match, report  # noqa: E402

TOL = 1e-4


def check_proposals(p, o, label):
    """set match at IoU = 1.  An unmatched box must be a near-tie of the RPN scores AND the committed seeds have none:
    parity.assert_proposals_match asserts near == 0, so everything after this call runs unconditionally."""
    return assert_proposals_match(p["rois"][0].cpu(), p["roi_scores"][0].cpu(), o["rois"][0], o["roi_scores"][0],
                                  o["_scores_sorted_all"], label=label)


def build(cfg, seed=0):
    from sis3d.nets import backbones
    net = getattr(backbones, cfg.NET)(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_state_dict(shapes, seed=seed, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    return net.cuda().eval(), sd


def blobs_for(data, feats=None, i3d=None, i2d=None):
    b = {"data": data, "id": ["syn0"], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]]}
    if feats is not None:
        b["nearest_images"] = {"images": [feats]}
        b["proj_ind_3d"] = [i3d]
        b["proj_ind_2d"] = [i2d]
    return b


@pytest.mark.parametrize("name,use_images", [("e2e_geometry_small", False), ("e2e_images_small", True),
                                             ("e2e_geometry_full", False), ("e2e_suncg_small", True), ("e2e_only_images_small", True)])
def test_forward_vs_oracle_and_golden(oracle, golden, name, use_images):
    g = golden(name)
    dims = tuple(int(v) for v in g["dims"])
    cfg = config.suncg_cfg() if "suncg" in name else config.scannet_benchmark_cfg()    # SUNCG_Backbone: 64-wide stems, 3/6 anchors, 26 classes
    cfg.USE_IMAGES = use_images
    cfg.ONLY_IMAGES = "only_images" in name                   # colour branch alone feeds level 1 (backbones.py:99-101)
    net, sd = build(cfg)
    data = synthetic.synth_chunk(int(g["chunk_id"]), dims)
    feats = i3d = i2d = None
    if use_images:
        feats, i3d, i2d = synthetic.synth_views(int(g["chunk_id"]), n_views=int(g["n_views"]), n_per_view=int(g["n_per_view"]), dims=dims)
    p = net.forward(blobs_for(data, feats, i3d, i2d), "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data, feats, i3d, i2d)
    s = int(g["sub"])
    # ---- stage outputs, fp32 tolerance
    l1, l2 = net._net_conv
    assert (l1.cpu() - o["level1"]).abs().max() <= TOL
    assert (l2.cpu() - o["level2"]).abs().max() <= TOL
    assert np.abs(l1.cpu()[0, :, ::s, ::s, ::s].numpy() - g["level1_sub"]).max() <= TOL      # vs the reference itself
    for lv in (1, 2):
        for k in ("rpn_cls_score_level%d", "rpn_cls_prob_level%d", "rpn_bbox_pred_level%d"):
            assert p[k % lv].shape == o[k % lv].shape
            assert (p[k % lv].cpu() - o[k % lv]).abs().max() <= TOL, k % lv
    if use_images:
        assert torch.equal(net._imageft.cpu(), o["imageft"])                                  # bit-exact gather
    # ---- proposals: same set up to near-tie reordering
    check_proposals(p, o, name)                                # asserts 0 near-ties
    # the device list is the reference's own list (fixture), row for row
    assert p["rois"][0].shape[0] == g["rois"].shape[0]
    assert np.abs(p["rois"][0].cpu().numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(p["roi_scores"][0].cpu().numpy() - g["roi_scores"]).max() <= TOL
    assert np.array_equal(p["level_inds"][0].cpu().numpy(), g["level_inds"])
    assert np.array_equal(p["cls_pred"].cpu().numpy(), g["cls_pred"])
    assert np.abs(p["cls_score"].cpu().numpy() - g["cls_score"]).max() <= TOL
    assert np.abs(p["bbox_pred"].cpu().numpy() - g["bbox_pred"]).max() <= TOL
    report("%s: %d rois / scores / levels / cls_pred / cls_score / bbox_pred equal to the reference fixture row for row" % (
        name, g["rois"].shape[0]))


def test_stage_isolated_heads_exact_inputs(oracle):
    """feed ORACLE tensors into the device proposal / RoI / classifier stages: integer outputs bit-exact,
    logits within 1e-4 (SURVEY.md 8c parity protocol (1)+(2))."""
    from sis3d import ops
    cfg = config.scannet_benchmark_cfg()
    net, sd = build(cfg)
    dims = synthetic.CHUNK_DIMS
    data = synthetic.synth_chunk(0, dims)
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data)
    net._scene_info = dims
    levels = []
    for lv in (1, 2):
        anchors = oracle.generate_anchors((24, 12, 24), 4, config.anchor_sizes(cfg, lv))
        levels.append((lv, o["rpn_cls_prob_level%d" % lv].cuda(), o["rpn_bbox_pred_level%d" % lv].cuda(), anchors))
    r = net._proposals.run(levels, dims)
    n = int(r["num"].item())
    assert torch.equal(r["order"][:r["n_pre"]].cpu(), o["_order"])          # stable sort, same tie rule
    assert torch.equal(r["keep"][:n].cpu(), o["_keep"])                     # NMS keep list bit-exact
    assert (r["rois"][:n].cpu() - o["rois"][0]).abs().max() <= TOL
    assert torch.equal(r["levels"][:n].cpu(), o["level_inds"][0])
    # RoI pooling + classifier on the oracle's rois / features
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last_3d)
    net._prop = dict(rois=o["rois"][0].cuda(), levels=o["level_inds"][0].cuda())
    outs = net._classify_rois(cl(o["level1"]), cl(o["level2"]))
    assert torch.equal(net._pool5.cpu(), o["pool5"])                        # pooled values bit-exact
    assert (outs[0].cpu() - o["cls_score"]).abs().max() <= TOL
    assert (outs[3].cpu() - o["bbox_pred"]).abs().max() <= TOL
    assert torch.equal(outs[1].cpu(), o["cls_pred"])


def test_mask_head_vs_oracle(oracle):
    cfg = config.scannet_benchmark_cfg()
    net, sd = build(cfg)
    on = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    data = synthetic.synth_chunk(2)
    for win in ((10, 5, 20, 22, 25, 33), (0, 0, 0, 8, 10, 9), (80, 30, 70, 96, 48, 96), (3, 3, 3, 4, 4, 4)):
        x0, y0, z0, x1, y1, z1 = win
        want = on.mask_backbone(data[:, :, x0:x1, y0:y1, z0:z1])
        got = net.mask_backbone(data.cuda(), None, window=win)
        assert got.shape == want.shape
        assert (got.cpu() - want).abs().max() <= TOL
        # the reference call form: a sliced view of the scene
        got2 = net.mask_backbone(data.cuda()[:, :, x0:x1, y0:y1, z0:z1], None)
        assert (got2.cpu() - want).abs().max() <= TOL


def test_full_forward_masks(oracle, golden):
    """BASELINE config 3 at full size, mask VALUES: every device mask against the oracle's mask of the same detection
    (same crop window and class), and the reference's own first four masks (fixture) against their device twins"""
    from sis3d.model.trainval import final_detections, mask_windows
    g = golden("e2e_geometry_full")
    cfg = config.scannet_benchmark_cfg()
    net, sd = build(cfg)
    data = synthetic.synth_chunk(0)
    p = net.forward(blobs_for(data), "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data)
    check_proposals(p, o, "config 3 full size")                # asserts 0 near-ties
    masks = p["mask_pred"][0]
    _, _, pred_box, keep = final_detections(p, data.shape[2:], cfg)
    wins = mask_windows(pred_box, keep)
    assert len(wins) == len(masks) and len(masks) > 0
    owins = [tuple(c) for c in o["_mask_aux"]["crops"]]
    omasks = o["mask_pred"][0]
    assert len(omasks) == int(g["n_masks"])
    assert wins == owins                                       # same detections, same integer crop windows, same order
    by_win = {w: m for w, m in zip(owins, omasks)}
    flips = checked = 0
    for w, m in zip(wins, masks):
        assert m.shape[1] == cfg.NUM_CLASSES and tuple(m.shape[2:]) == (w[3] - w[0], w[4] - w[1], w[5] - w[2])
        want = by_win[w]
        assert float((m.cpu() - want).abs().max()) <= TOL, w                # sigmoid outputs of all 19 class channels
        flips += int(((m.cpu() >= cfg.MASK_THRESH) != (want >= cfg.MASK_THRESH)).sum())
        checked += 1
    assert checked == len(owins) > 0
    report("config 3 masks: %d of %d compared at 1e-4, %d voxels flip at MASK_THRESH (|p - 0.5| < 1e-4)" % (checked, len(masks), flips))
    # the reference's own masks (fixture): its first four detections
    for i in range(min(4, int(g["n_masks"]))):
        ref_m = torch.from_numpy(g["mask_%d" % i])
        # the oracle is pinned to them bit for bit on the machine that generated the fixture (tests/test_oracle_pinning.py);
        # another host's oneDNN may pick a different summation order for the same convs
        assert float((omasks[i] - ref_m).abs().max()) <= 1e-5
        assert float((dict(zip(wins, masks))[owins[i]].cpu() - ref_m).abs().max()) <= TOL


@pytest.mark.parametrize("name,use_images", [("e2e_geometry_full", False), ("e2e_images_small", True)])
def test_forward_under_the_shared_chip_dispatch(oracle, golden, name, use_images):
    """r4: the dispatch that pipelines of several chunks in flight capture (ops.dispatch_regime(shared_chip=True): geometry2[0] with two cout
    tiles per Winograd workgroup, the 64 -> 64 convs and the Bottleneck(128,32) bodies on the Winograd kernel) -- i.e. the kernels that set the
    bench headline -- against the oracle and the reference's own fixture at the same tolerances and the same 0-near-tie rule as the default
    dispatch, and against the default dispatch itself (levels within 2e-5 of their scale)"""
    from sis3d import ops
    g = golden(name)
    dims = tuple(int(v) for v in g["dims"])
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = use_images
    net, sd = build(cfg)
    data = synthetic.synth_chunk(int(g["chunk_id"]), dims)
    feats = i3d = i2d = None
    if use_images:
        feats, i3d, i2d = synthetic.synth_views(int(g["chunk_id"]), n_views=int(g["n_views"]), n_per_view=int(g["n_per_view"]), dims=dims)
    net.forward(blobs_for(data, feats, i3d, i2d), "TEST", [])
    d1, d2 = [t.clone() for t in net._net_conv]
    ops.flop_tally(True)
    net.forward(blobs_for(data, feats, i3d, i2d), "TEST", [])
    base = ops.flop_tally(False)["wino_launches"]
    with ops.dispatch_regime(shared_chip=True):
        ops.flop_tally(True)
        p = net.forward(blobs_for(data, feats, i3d, i2d), "TEST", [])
        shared = ops.flop_tally(False)["wino_launches"]
    if dims == (96, 48, 96):
        assert shared > base, (shared, base)                   # more layers really took the Winograd kernel
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data, feats, i3d, i2d)
    l1, l2 = net._net_conv
    assert (l1.cpu() - o["level1"]).abs().max() <= TOL and (l2.cpu() - o["level2"]).abs().max() <= TOL
    assert (l1 - d1).abs().max().item() <= 2e-5 * float(d1.abs().max()) and (l2 - d2).abs().max().item() <= 2e-5 * float(d2.abs().max())
    for lv in (1, 2):
        for k in ("rpn_cls_score_level%d", "rpn_cls_prob_level%d", "rpn_bbox_pred_level%d"):
            assert (p[k % lv].cpu() - o[k % lv]).abs().max() <= TOL, k % lv
    check_proposals(p, o, name + " (shared-chip dispatch)")    # asserts 0 near-ties
    assert p["rois"][0].shape[0] == g["rois"].shape[0]
    assert np.abs(p["rois"][0].cpu().numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(p["roi_scores"][0].cpu().numpy() - g["roi_scores"]).max() <= TOL
    assert np.array_equal(p["level_inds"][0].cpu().numpy(), g["level_inds"])
    assert np.array_equal(p["cls_pred"].cpu().numpy(), g["cls_pred"])
    assert np.abs(p["cls_score"].cpu().numpy() - g["cls_score"]).max() <= TOL
    assert np.abs(p["bbox_pred"].cpu().numpy() - g["bbox_pred"]).max() <= TOL
    report("%s under the shared-chip dispatch (%d Winograd launches instead of %d): %d rois / scores / levels / cls_pred / cls_score / "
           "bbox_pred equal to the reference fixture row for row" % (name, shared, base, g["rois"].shape[0]))


def test_config4_full_size_vs_oracle(oracle):
    """BASELINE config 4 at 96x48x96 (feature maps handed in, USE_IMAGES_GT): RPN maps, logits, proposals vs the oracle"""
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES, cfg.USE_MASK = True, False
    net, sd = build(cfg)
    data = synthetic.synth_chunk(13)
    feats, i3d, i2d = synthetic.synth_views(13)                # 5 views, 3000 visible voxels each
    p = net.forward(blobs_for(data, feats, i3d, i2d), "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data, feats, i3d, i2d)
    l1, l2 = net._net_conv
    assert (l1.cpu() - o["level1"]).abs().max() <= TOL and (l2.cpu() - o["level2"]).abs().max() <= TOL
    assert torch.equal(net._imageft.cpu(), o["imageft"])
    for lv in (1, 2):
        for k in ("rpn_cls_score_level%d", "rpn_cls_prob_level%d", "rpn_bbox_pred_level%d"):
            assert (p[k % lv].cpu() - o[k % lv]).abs().max() <= TOL, k % lv
    check_proposals(p, o, "config 4 full size")                # asserts 0 near-ties
    assert (p["cls_score"].cpu() - o["cls_score"]).abs().max() <= TOL
    assert (p["bbox_pred"].cpu() - o["bbox_pred"]).abs().max() <= TOL
    assert torch.equal(p["cls_pred"].cpu(), o["cls_pred"])


def test_mask_head_batched_equals_per_box(oracle):
    """ragged one-launch-per-layer mask head == per-box launches == oracle (1e-4).  r3: a batch with >= 200 work items takes the
    Winograd kernel's ragged launch for its four 64->64 layers, a single box stays on the direct kernel: the two agree to fp32
    summation noise (1e-5 on the sigmoid outputs); with Winograd switched off the batched and per-box results differ only in the
    first (planar) and last (1x1x1) layers' kernels -- r6: MFMA tile GEMMs in the batch, FMA chains per box -- and agree to 1e-6."""
    from sis3d import ops
    cfg = config.scannet_benchmark_cfg()
    net, sd = build(cfg)
    on = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    data = synthetic.synth_chunk(4)
    wins = [(10, 5, 20, 22, 25, 33), (0, 0, 0, 8, 10, 9), (80, 30, 70, 96, 48, 96), (3, 3, 3, 4, 4, 4), (40, 10, 40, 71, 33, 59), (5, 6, 7, 7, 9, 12)]
    net.mask_backbone.eval()
    plan = ops.MaskPlan(wins, 64, cfg.NUM_CLASSES, torch.device("cuda"))
    assert plan.wino and plan.blocks_wino >= 200                   # this batch does take the Winograd ragged launch
    got = [t.clone() for t in net.mask_backbone.forward_batched(data.cuda(), wins)]
    assert len(got) == len(wins)
    ops.set_winograd(False)
    try:
        direct = [t.clone() for t in net.mask_backbone.forward_batched(data.cuda(), wins)]
    finally:
        ops.set_winograd(True)
    worst = 0.0
    for w, g, d in zip(wins, got, direct):
        x0, y0, z0, x1, y1, z1 = w
        want = on.mask_backbone(data[:, :, x0:x1, y0:y1, z0:z1])
        assert g.shape == want.shape
        assert (g.cpu() - want).abs().max() <= TOL and (d.cpu() - want).abs().max() <= TOL
        single = net.mask_backbone(data.cuda(), None, window=w)    # one box: direct kernel
        assert (d - single).abs().max() <= 1e-6
        worst = max(worst, float((g - single).abs().max()))
    assert worst <= 1e-5
    report("mask head, 6 crops: Winograd ragged batch vs direct per-box launches, max |diff| on the sigmoid outputs %.1e" % worst)
    # training mode returns the raw logits (backbones.py:286-287 applies the sigmoid only when not self.training): the last layer then
    # runs WITHOUT the sigmoid epilogue, per box (generic kernel: its pack is padded to 32 couts for the batch) and batched
    net.mask_backbone.train()
    try:
        raw1 = net.mask_backbone(data.cuda(), None, window=wins[0])
        rawb = net.mask_backbone.forward_batched(data.cuda(), wins)[0]
    finally:
        net.mask_backbone.eval()
    assert (torch.sigmoid(raw1) - got[0]).abs().max() <= 1e-5 and (torch.sigmoid(rawb) - got[0]).abs().max() <= 1e-5
    assert float(raw1.min()) < 0.0                                  # logits, not probabilities
    # r4: the batch above ran on MINI geometry (quads of 4 x 4 x 4 bricks); the 8 x 4 x 8-block launch computes the same tiles in another
    # grouping -> bit-identical outputs, with fewer work items
    assert plan.wino_mini and plan.items_mini < plan.blocks_wino
    ops.MASK_MINI = False
    try:
        plan8 = ops.MaskPlan(wins, 64, cfg.NUM_CLASSES, torch.device("cuda"))
        assert plan8.wino and not plan8.wino_mini
        blocks = [t.clone() for t in net.mask_backbone.forward_batched(data.cuda(), wins)]
    finally:
        ops.MASK_MINI = True
    assert all(torch.equal(a, b) for a, b in zip(got, blocks))
    report("mask head, 6 crops: %d work items on 4 x 4 x 4 minis vs %d on 8 x 4 x 8 blocks, outputs bit-identical" % (plan.items_mini, plan.blocks_wino))
    assert net.mask_backbone.forward_batched(data.cuda(), []) == []


@pytest.mark.parametrize("classes", [9, 26, 40])
def test_mask_head_last_layer_routes_by_class_count(oracle, classes):
    """the batched mask head's last layer: 17..32 classes (ScanNet 19, SUNCG 26) run on the pointwise kernel with rows padded to 32
    couts, fewer (9) on the generic ragged 1x1x1 kernel with dense rows, more (40) through ops.conv3d -- all three == the oracle"""
    from sis3d import ops
    cfg = config.scannet_benchmark_cfg()
    cfg.NUM_CLASSES = classes
    net, sd = build(cfg, seed=2)
    on = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    data = synthetic.synth_chunk(7)
    wins = [(10, 5, 20, 22, 25, 33), (0, 0, 0, 8, 10, 9), (80, 30, 70, 96, 48, 96), (40, 10, 40, 71, 33, 59)]
    plan = ops.MaskPlan(wins, 64, classes, torch.device("cuda"))
    assert (plan.out_pad is not None) == (16 < classes <= 32)
    net.mask_backbone.eval()
    got = net.mask_backbone.forward_batched(data.cuda(), wins)
    for w, g in zip(wins, got):
        want = on.mask_backbone(data[:, :, w[0]:w[3], w[1]:w[4], w[2]:w[5]])
        assert g.shape == want.shape and (g.cpu() - want).abs().max() <= TOL


def test_mask_head_batch_of_more_crops_than_the_lds_descriptor_table():
    """r6: the first (planar) layer's MFMA kernel keeps up to 128 crop descriptors in LDS and reads them from global memory beyond that:
    150 small crops in one batch == the same crops in two batches of 75 (same arithmetic per voxel: bit-identical)"""
    cfg = config.scannet_benchmark_cfg()
    net, _ = build(cfg)
    net.mask_backbone.eval()
    data = synthetic.synth_chunk(6).cuda()
    g = torch.Generator().manual_seed(11)
    wins = []
    for _ in range(150):
        d = torch.randint(1, 8, (3,), generator=g).tolist()
        o = [int(torch.randint(0, hi - dd + 1, (1,), generator=g)) for hi, dd in zip((96, 48, 96), d)]
        wins.append((o[0], o[1], o[2], o[0] + d[0], o[1] + d[1], o[2] + d[2]))
    whole = [t.clone() for t in net.mask_backbone.forward_batched(data, wins)]
    halves = [t.clone() for t in net.mask_backbone.forward_batched(data, wins[:75])] + \
             [t.clone() for t in net.mask_backbone.forward_batched(data, wins[75:])]
    assert len(whole) == 150 and all(torch.equal(a, b) for a, b in zip(whole, halves))


@pytest.mark.parametrize("dims", [(70, 46, 58), (128, 64, 40)])
def test_whole_scene_odd_grid_vs_oracle(oracle, dims):
    """the reference's benchmark mode convolves whole (non-chunked) scene grids of arbitrary size
    (lib/datasets/dataset.py:192-205): odd extents (floor at each stride-2 conv), bricks overhanging the volume"""
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    net, sd = build(cfg, seed=5)
    data = synthetic.synth_chunk(21, dims)
    p = net.forward(blobs_for(data), "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data)
    l1, l2 = net._net_conv
    assert l1.shape == o["level1"].shape and l2.shape == o["level2"].shape
    assert (l1.cpu() - o["level1"]).abs().max() <= TOL and (l2.cpu() - o["level2"]).abs().max() <= TOL
    for lv in (1, 2):
        assert (p["rpn_cls_prob_level%d" % lv].cpu() - o["rpn_cls_prob_level%d" % lv]).abs().max() <= TOL
        assert (p["rpn_bbox_pred_level%d" % lv].cpu() - o["rpn_bbox_pred_level%d" % lv]).abs().max() <= TOL
    check_proposals(p, o, "odd grid %s" % (dims,))             # asserts 0 near-ties
    assert (p["cls_score"].cpu() - o["cls_score"]).abs().max() <= TOL
    assert torch.equal(p["cls_pred"].cpu(), o["cls_pred"])


@pytest.mark.parametrize("kill_view", [None, 1])
def test_forward_from_depth_maps_vs_oracle(oracle, kill_view):
    """SURVEY 8f row 1 -> 8a: depth maps + poses -> device compute_projection -> forward, against the oracle running the
    reference's caller block (trainval.py:659-683) with its own compute_projection; incl. a view that sees nothing."""
    from sis3d.layer_utils.projection import prepare_projection
    dims = (64, 32, 48)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = True
    net, sd = build(cfg)
    data = synthetic.synth_chunk(4, dims)
    V = 4
    depth, c2w, w2g = synthetic.synth_cameras(31, V, dims, cfg.VOXEL_SIZE)
    if kill_view is not None:
        depth[kill_view] = 0.0
    feats = torch.randn(V, 128, 32, 41, generator=torch.Generator().manual_seed(17))
    # oracle side: per-view lists, None -> killing_inds, stack of the survivors
    maps = [oracle.compute_projection(depth[v], c2w[v], w2g[v], cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX,
                                      cfg.DEPTH_SHAPE, dims, cfg.VOXEL_SIZE) for v in range(V)]
    want_kill = [v for v, m in enumerate(maps) if m is None]
    o3 = torch.stack([m[0] for m in maps if m is not None])
    o2 = torch.stack([m[1] for m in maps if m is not None])
    blobs = {"data": data, "id": ["syn0"], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]],
             "nearest_images": {"images": [feats], "depths": [depth], "poses": [c2w], "world2grid": [w2g]}}
    kill = prepare_projection(blobs, cfg)
    assert kill == want_kill == ([] if kill_view is None else [kill_view])
    assert torch.equal(blobs["proj_ind_3d"][0].cpu(), o3) and torch.equal(blobs["proj_ind_2d"][0].cpu(), o2)
    p = net.forward(blobs, "TEST", kill)
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data, feats, o3, o2,
                                                                                                 killing_inds=want_kill)
    assert torch.equal(net._imageft.cpu(), o["imageft"]) and o["imageft"].abs().sum() > 0
    for lv in (1, 2):
        k = "rpn_cls_prob_level%d" % lv
        assert (p[k].cpu() - o[k]).abs().max() <= TOL
    check_proposals(p, o, "from depth maps, kill=%s" % kill_view)


@pytest.mark.parametrize("n_per_view,kill", [(400, ()), (3000, (1,)), (0, ())])
def test_fused_projection_equals_materialised_volume(oracle, n_per_view, kill):
    """colour stem reading the views through the voxel->pixel table (no 226 MB volume, empty bricks skipped) ==
    the same network on the materialised volume, bit for bit (dense table kernel) / within 1e-5 (sparse kernels, the default);
    and the lazily materialised `_imageft` is the oracle's."""
    from sis3d import ops
    dims = (96, 48, 96) if n_per_view == 3000 else (64, 32, 48)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = True
    cfg.USE_MASK = False
    net, sd = build(cfg)
    data = synthetic.synth_chunk(6, dims)
    feats, i3d, i2d = synthetic.synth_views(6, n_views=4, n_per_view=n_per_view, dims=dims)
    outs = []
    ops.set_sparse_projection(False)                  # the DENSE table kernel: same tiles, same summation order as the tensor path
    try:
        for fuse in (True, False):
            net.fuse_projection = fuse
            net.delete_intermediate_states()
            p = net.forward(blobs_for(data, feats, i3d, i2d), "TEST", list(kill))
            assert isinstance(net._image_input, ops.ProjectedVolume) == fuse
            outs.append((net._net_conv[0].clone(), net._net_conv[1].clone(), p["rpn_cls_prob_level1"].clone(), p["rois"][0].clone(),
                         net._imageft.clone()))
    finally:
        ops.set_sparse_projection(True)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # the default route, csrc/proj_sparse.hip (only output voxels with a visible input voxel are computed; tap-major summation):
    # the same network within fp32 summation noise, same proposals
    net.fuse_projection = True
    net.delete_intermediate_states()
    p = net.forward(blobs_for(data, feats, i3d, i2d), "TEST", list(kill))
    assert isinstance(net._image_input, ops.ProjectedVolume)
    for got, want in ((net._net_conv[0], outs[0][0]), (net._net_conv[1], outs[0][1]), (p["rpn_cls_prob_level1"], outs[0][2])):
        assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert p["rois"][0].shape == outs[0][3].shape and float((p["rois"][0] - outs[0][3]).abs().max()) <= 1e-3
    want = oracle.project_views_max(feats, i3d, i2d, dims, kill)
    assert torch.equal(outs[0][4].cpu(), want)


@pytest.mark.parametrize("switch", ["no_class", "level1_only", "level2_only", "sort_fallback", "allow_border"])
def test_forward_config_switches(oracle, switch):
    """cfg switches that change the TEST forward (lib/nets/network.py:241-282, proposal_layer.py:36-43,181-197): RPN-only,
    single pyramid level, > 1024 pre-NMS candidates (torch.sort instead of the top-k kernel), anchors allowed over the border"""
    dims = (64, 32, 48)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    if switch == "no_class":
        cfg.USE_CLASS = False
    elif switch == "level1_only":
        cfg.NUM_ANCHORS_LEVEL2 = 0
    elif switch == "level2_only":
        cfg.NUM_ANCHORS_LEVEL1 = 0
    elif switch == "sort_fallback":
        cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 2000, 300
    elif switch == "allow_border":
        cfg.ALLOW_BORDER = 8
    net, sd = build(cfg)
    data = synthetic.synth_chunk(8, dims)
    p = net.forward(blobs_for(data), "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data)
    for lv in (1, 2):
        k = "rpn_cls_prob_level%d" % lv
        assert (k in p) == (k in o)
        if k in o:
            assert (p[k].cpu() - o[k]).abs().max() <= TOL
    rois, want = p["rois"][0].cpu(), o["rois"][0]
    assert want.shape[0] > 0
    check_proposals(p, o, switch)
    lv_got = set(p["level_inds"][0].cpu().tolist())
    assert lv_got <= set(o["level_inds"][0].tolist()) | {1.0, 2.0}
    if switch == "level1_only":
        assert lv_got == {1.0}
    if switch == "level2_only":
        assert lv_got == {2.0}
    if switch == "no_class":
        assert "cls_prob" not in p and "cls_prob" not in o
    else:
        assert p["cls_prob"].shape == o["cls_prob"].shape
    if switch == "sort_fallback":
        assert want.shape[0] > 200 or rois.shape[0] <= 300
