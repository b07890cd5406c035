"""Pins the CPU oracle (oracle/sis3d_oracle.{py,c}) to the REFERENCE's own outputs.

(1) against the committed fixtures tests/golden/*.npz that oracle/make_golden.py
    produced by running the reference in place -- runs everywhere;
(2) against the reference executed live (only where /root/reference exists).
Bit-exactness is required throughout: the oracle performs the same binary32
operations in the same order as the reference's CPU path.
"""
import hashlib

import numpy as np
import pytest
import torch

from sis3d import config, synthetic


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_nms_matches_reference_cpu_nms(golden, oracle):
    g = golden("nms_cases")
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) >= 9
    for name in names:
        boxes = torch.from_numpy(g[name + "/boxes"])
        for th in (0.1, 0.35, 0.5):
            keep = oracle.nms(boxes, th).numpy()
            assert np.array_equal(keep, g["%s/keep_%g" % (name, th)]), (name, th)
            # the bit-matrix + sweep formulation (CUDA path of the reference) gives the same list
            m = oracle.nms_mask(boxes, th).numpy().view(np.uint64)
            n, cb = m.shape
            remv = np.zeros(cb, dtype=np.uint64)
            k2 = []
            for i in range(n):
                if not (int(remv[i // 64]) >> (i % 64)) & 1:
                    k2.append(i)
                    remv |= m[i]
            assert k2 == keep.tolist(), (name, th)


def test_roi_pool_matches_reference_c_and_python(golden, oracle):
    g = golden("roi_pool_cases")
    feat, rois = torch.from_numpy(g["feat"]), torch.from_numpy(g["rois"])
    out, arg = oracle.roi_pool(feat, rois, (4, 4, 4), 0.25)
    assert np.array_equal(out.numpy(), g["out_c_4"])
    out2, arg2 = oracle.roi_pool(feat, rois[:6], (2, 2, 2), 0.25)
    assert np.array_equal(out2.numpy(), g["out_py_2"])
    # argmax (CUDA kernel convention: linear index, -1 when empty) vs the Python RoIPool's (w,h,l)
    whl = g["argmax_whl_py_2"]
    C, W, H, L = feat.shape[1:]
    a = arg2.numpy()
    for idx in np.ndindex(*a.shape):
        w, h, l = whl[idx]
        if w < 0:
            assert a[idx] == -1
        else:
            assert a[idx] == (idx[1] * W + int(w)) * H * L + int(h) * L + int(l)
    # empty bins are 0 / -1
    assert (out[3] == 0).all() and (arg[3] == -1).all()


def test_projection_matches_reference(golden, oracle):
    g = golden("projection_cases")
    dims = tuple(int(v) for v in g["dims"])
    feats, i3d, i2d = (torch.from_numpy(g[k]) for k in ("feats", "i3d", "i2d"))
    for v in range(feats.shape[0]):
        assert np.array_equal(oracle.projection(feats[v], i3d[v], i2d[v], dims).numpy(), g["out"][v])
    assert np.array_equal(oracle.projection(feats[0, 0], i3d[0], i2d[0], dims).numpy(), g["out2d"])
    # fused multi-view rule == elementwise max of the zero-filled per-view volumes
    fused = oracle.project_views_max(feats, i3d, i2d, dims)
    ref = torch.from_numpy(g["out"]).max(0).values
    assert torch.equal(fused[0].permute(0, 3, 2, 1), ref)
    # killing_inds semantics (network.py:220-223): skipped positions simply do not take part
    fused_k = oracle.project_views_max(feats, i3d, i2d, dims, killing_inds=[1])
    ref_k = torch.from_numpy(g["out"][[0, 2]]).max(0).values
    assert torch.equal(fused_k[0].permute(0, 3, 2, 1), ref_k)


def _oracle_compute_projection(oracle, c, depth, c2w, w2g, dims):
    return oracle.compute_projection(depth, c2w, w2g, c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.DEPTH_SHAPE, dims,
                                     c.VOXEL_SIZE)


def test_compute_projection_matches_reference(golden, oracle):
    """ProjectionHelper.compute_projection (projection.py:52-121) incl. its three None exits."""
    g = golden("compute_projection_cases")
    c = config.scannet_benchmark_cfg()
    for name in ("small", "odd", "chunk"):
        dims = tuple(int(v) for v in g[name + "_dims"])
        depth, c2w, w2g = (torch.from_numpy(g[name + k]) for k in ("_depth", "_c2w", "_w2g"))
        for v, n in enumerate(g[name + "_counts"]):
            r = _oracle_compute_projection(oracle, c, depth[v], c2w[v], w2g[v], dims)
            if n == 0:
                assert r is None
                continue
            assert int(r[0][0]) == n and int(r[1][0]) == n
            assert np.array_equal(r[0][1:1 + n].numpy(), g["%s_l3_%d" % (name, v)])
            assert np.array_equal(r[1][1:1 + n].numpy(), g["%s_l2_%d" % (name, v)])
            assert not r[0][1 + n:].any() and not r[1][1 + n:].any()


def test_compute_projection_vs_live_reference(oracle):
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install()
    c = config.scannet_benchmark_cfg()
    dims = (52, 30, 44)
    depth, c2w, w2g = synthetic.synth_cameras(21, 6, dims, c.VOXEL_SIZE)
    hits = 0
    for v in range(6):
        r = rh.ref_compute_projection(ns, depth[v], c2w[v], w2g[v], dims)
        o = _oracle_compute_projection(oracle, c, depth[v], c2w[v], w2g[v], dims)
        assert (r is None) == (o is None)
        if r is not None:
            n = int(r[0][0])
            hits += n
            assert torch.equal(r[0][:n + 1], o[0][:n + 1]) and torch.equal(r[1][:n + 1], o[1][:n + 1])
    assert hits > 1000


def test_anchors_match_reference(golden, oracle):
    g = golden("anchors")
    c = config.scannet_benchmark_cfg()
    a1 = oracle.generate_anchors([3, 2, 4], 4, config.anchor_sizes(c, 1))
    a2 = oracle.generate_anchors([3, 2, 4], 4, config.anchor_sizes(c, 2))
    assert np.array_equal(a1, g["small_l1"]) and np.array_equal(a2, g["small_l2"])
    f1 = oracle.generate_anchors([24, 12, 24], 4, config.anchor_sizes(c, 1))
    f2 = oracle.generate_anchors([24, 12, 24], 4, config.anchor_sizes(c, 2))
    assert _sha(f1) == str(g["full_l1_sha"]) and _sha(f2) == str(g["full_l2_sha"])
    assert f1.shape == (20736, 6) and f2.shape == (76032, 6)
    # 11,412 + 21,982 anchors lie inside a 96x48x96 chunk
    n1 = oracle.inside_anchor_inds(f1, (96, 48, 96)).numel()
    n2 = oracle.inside_anchor_inds(f2, (96, 48, 96)).numel()
    assert (n1, n2) == (11412, 21982)


def _shapes_from_golden(g):
    # parameter shapes are fixed by SURVEY Appendix A; rebuild them from our own mirror net
    from sis3d.nets import backbones
    return backbones.state_dict_shapes


@pytest.mark.parametrize("name,use_images", [("e2e_geometry_small", False), ("e2e_images_small", True),
                                             ("e2e_geometry_full", False), ("e2e_suncg_small", True), ("e2e_only_images_small", True)])
def test_forward_matches_reference_golden(golden, oracle, name, use_images):
    from sis3d.nets.backbones import state_dict_shapes
    g = golden(name)
    dims = tuple(int(v) for v in g["dims"])
    c = config.suncg_cfg() if "suncg" in name else config.scannet_benchmark_cfg()     # second model family: SUNCG_Backbone
    c.USE_IMAGES = use_images
    c.ONLY_IMAGES = "only_images" in name                      # colour branch alone feeds level 1 (backbones.py:99-101)
    if "num_classes" in g.files:
        assert c.NUM_CLASSES == int(g["num_classes"])
    shapes = state_dict_shapes(c)
    assert sorted(shapes.keys()) == list(g["shapes_keys"])       # checkpoint contract
    sd = synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)
    data = synthetic.synth_chunk(int(g["chunk_id"]), dims)
    feats = i3d = i2d = None
    if use_images:
        feats, i3d, i2d = synthetic.synth_views(int(g["chunk_id"]), n_views=int(g["n_views"]),
                                                n_per_view=int(g["n_per_view"]), dims=dims)
    net = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2))
    o = net.forward(data, feats, i3d, i2d)
    s = int(g["sub"])
    assert _sha(o["level1"].numpy()) == str(g["level1_sha"])
    assert _sha(o["level2"].numpy()) == str(g["level2_sha"])
    assert np.array_equal(o["level1"][0, :, ::s, ::s, ::s].numpy(), g["level1_sub"])
    for lv in (1, 2):
        assert np.array_equal(o["rpn_cls_score_level%d" % lv][0, :, ::s, ::s, ::s].numpy(), g["rpn_cls_score_level%d_sub" % lv])
        assert np.array_equal(o["rpn_cls_prob_level%d" % lv][0, :, ::s, ::s, ::s].numpy(), g["rpn_cls_prob_level%d_sub" % lv])
        assert np.array_equal(o["rpn_bbox_pred_level%d" % lv][0, ::s, ::s, ::s].numpy(), g["rpn_bbox_pred_level%d_sub" % lv])
    assert np.array_equal(o["rois"][0].numpy(), g["rois"])
    assert np.array_equal(o["roi_scores"][0].numpy(), g["roi_scores"])
    assert np.array_equal(o["level_inds"][0].numpy(), g["level_inds"])
    for k in ("cls_score", "cls_pred", "cls_prob", "bbox_pred"):
        assert np.array_equal(o[k].numpy(), g[k]), k
    masks = o["mask_pred"][0]
    assert len(masks) == int(g["n_masks"])
    assert np.array_equal(np.array([list(m.shape[2:]) for m in masks]).reshape(-1, 3), g["mask_shapes"])
    for i in range(min(4, len(masks))):
        assert np.array_equal(masks[i].numpy(), g["mask_%d" % i])
    if use_images:
        ift = o["imageft"]
        assert tuple(ift.stride()) == tuple(int(v) for v in g["imageft_stride"])
        nz = ift[0].abs().sum(0).nonzero()
        assert np.array_equal(nz.numpy().astype(np.int32), g["imageft_nz_xyz"])
        assert np.array_equal(ift[0][:, nz[:, 0], nz[:, 1], nz[:, 2]].numpy(), g["imageft_nz_val"])


def test_oracle_vs_live_reference(oracle):
    """Only where the reference tree is mounted (build container)."""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    import os
    if not os.path.exists(os.path.join(os.path.dirname(rh.__file__), "_ref", "libref_roi_pooling.so")):
        pytest.skip("oracle/_ref not built")
    ns = rh.install()
    dims = (48, 24, 64)
    net = rh.build_net(ns, seed=0, use_images=True, use_mask=True)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_state_dict(shapes, seed=3, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    data = synthetic.synth_chunk(9, dims)
    feats, i3d, i2d = synthetic.synth_views(9, n_views=4, n_per_view=300, dims=dims)
    p = rh.forward(ns, net, rh.make_blobs(data, feats, i3d, i2d), killing_inds=[2])
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES = True
    o = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2)).forward(
        data, feats, i3d, i2d, killing_inds=[2])
    assert torch.equal(net._imageft, o["imageft"])
    for k in ("rpn_cls_prob_level1", "rpn_bbox_pred_level2", "cls_score", "cls_pred", "bbox_pred"):
        assert torch.equal(p[k], o[k]), k
    assert torch.equal(p["rois"][0], o["rois"][0]) and torch.equal(p["level_inds"][0], o["level_inds"][0])
    assert len(p["mask_pred"][0]) == len(o["mask_pred"][0])
    for a, b in zip(p["mask_pred"][0], o["mask_pred"][0]):
        assert torch.equal(a, b)
    # reference numpy cpu_nms == oracle C nms on the decoded proposals
    boxes = o["rois"][0]
    assert np.array_equal(ns.pth_nms.cpu_nms(boxes.numpy(), 0.3), oracle.nms(boxes, 0.3).numpy())


def test_nms_fuzz_vs_live_reference(oracle):
    """40 random box sets (ties, duplicates, zero-volume boxes, 1..700 boxes) through the reference's numpy cpu_nms"""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install()
    g = torch.Generator().manual_seed(123)
    for case in range(40):
        n = int(torch.randint(1, 700, (1,), generator=g))
        lo = torch.rand(n, 3, generator=g) * 60
        size = torch.rand(n, 3, generator=g) * (30 if case % 2 else 6)
        boxes = torch.cat([lo, lo + size], 1)
        if case % 5 == 0:
            boxes = boxes.round()                                   # integer coordinates -> exact IoU ties at the threshold
        if case % 7 == 0:
            boxes = torch.cat([boxes[:n // 2], boxes[:n - n // 2]], 0)   # exact duplicates
        if case % 11 == 0:
            boxes[::9, 3:] = boxes[::9, :3]                         # zero-volume boxes
        for thresh in (0.1, 0.5):
            want = ns.pth_nms.cpu_nms(boxes.numpy(), thresh)
            got = oracle.nms(boxes, thresh).numpy()
            assert np.array_equal(np.asarray(want, dtype=np.int64), got), (case, thresh)


def test_collate_vs_live_reference(tmp_path):
    """sis3d.datasets.dataset.collate_fn == lib/datasets/dataloader.py:collate_fn on the geometry side"""
    import os
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install(with_trainval=True)
    from lib.datasets.dataloader import collate_fn as ref_collate
    from sis3d.datasets.dataset import Dataset, collate_fn
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    lst = tmp_path / "l.txt"
    lst.write_text(os.path.join(golden_dir, "synthetic.chunk") + "\n")
    c = config.scannet_benchmark_cfg()
    c.LABEL_MAP, c.USE_IMAGES, c.USE_MASK, c.KEEP_THRESH = os.path.join(golden_dir, "synthetic_labels.csv"), False, True, 0.5
    item = Dataset(str(lst), "chunk", c)[0]
    cfg = ns.cfg
    saved = (cfg.USE_IMAGES,)
    try:
        cfg.USE_IMAGES = False
        want = ref_collate([item])
    finally:
        cfg.USE_IMAGES = saved[0]
    got = collate_fn([item])
    assert got["id"] == want["id"] and torch.equal(got["data"], want["data"])
    assert len(got["gt_box"]) == len(want["gt_box"]) and all(torch.equal(a, b) for a, b in zip(got["gt_box"], want["gt_box"]))
    assert len(got["gt_mask"]) == len(want["gt_mask"])
    for ma, mb in zip(got["gt_mask"], want["gt_mask"]):
        assert len(ma) == len(mb) and all(torch.equal(a, b) for a, b in zip(ma, mb))
    assert got["nearest_images"] == want["nearest_images"] == {}
