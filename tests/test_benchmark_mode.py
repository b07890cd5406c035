"""`benchmark` mode (lib/model/trainval.py:634-767): whole-scene forward + the result files vox2mesh.py consumes.

CPU: the oracle's restatement against the reference-generated fixture (tests/golden/benchmark_small.npz, made by
oracle/make_golden.py running the reference's own SolverWrapper.benchmark), the product's host post-processing fed
with oracle predictions against the same fixture, and the oracle against the live reference where it is mounted.
GPU: sis3d.model.trainval.SolverWrapper.benchmark end to end against the oracle, incl. the resume rule."""
import os
import pickle

import numpy as np
import pytest
import torch

from sis3d import config, synthetic

DIMS = (48, 24, 40)
SCENE_ID = "/data/scenes/scene0707_00__0.scene"


def inputs(tag, cfg):
    """the seeded scene of oracle/make_golden.py:benchmark_case"""
    data = synthetic.synth_chunk(9, DIMS)
    blobs = {"data": data, "id": [SCENE_ID], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]]}
    if tag == "img":
        depth, c2w, w2g = synthetic.synth_cameras(5, 4, DIMS, cfg.VOXEL_SIZE)
        depth[2] = 0
        feats = torch.randn(4, 128, 32, 41, generator=torch.Generator().manual_seed(1))
        blobs["nearest_images"] = {"images": [feats], "depths": [depth], "poses": [c2w], "world2grid": [w2g]}
    return blobs


def cfg_for(tag, g):
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES = tag == "img"
    c.CLASS_THRESH = float(g[tag + "_class_thresh"])
    return c


def state_dict(cfg):
    from sis3d.nets import backbones
    net = getattr(backbones, cfg.NET)(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    return net, synthetic.synth_state_dict(shapes, seed=3, gains=synthetic.DEFAULT_GAINS)


def oracle_forward(oracle, cfg, sd, blobs):
    feats = o3 = o2 = None
    kill = []
    if cfg.USE_IMAGES:
        ni = blobs["nearest_images"]
        maps = [oracle.compute_projection(d, p, w, cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE, DIMS,
                                          cfg.VOXEL_SIZE) for d, p, w in zip(ni["depths"][0], ni["poses"][0], ni["world2grid"][0])]
        kill = [v for v, m in enumerate(maps) if m is None]
        o3 = torch.stack([m[0] for m in maps if m is not None])
        o2 = torch.stack([m[1] for m in maps if m is not None])
        feats = ni["images"][0]
    net = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    return net.forward(blobs["data"], feats, o3, o2, killing_inds=kill), kill


def unpack_masks(g, tag):
    shapes = g[tag + "_mask_shapes"]
    bits = np.unpackbits(g[tag + "_mask_bits"])
    out, k = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(bits[k:k + n].reshape(s).astype(np.float32))
        k += n
    return out


def check_files(files, g, tag):
    import hashlib
    assert files["pred_class"].dtype == np.int64 and np.array_equal(files["pred_class"], g[tag + "_pred_class"])
    assert files["pred_conf"].dtype == np.float64 and np.array_equal(files["pred_conf"], g[tag + "_pred_conf"])
    assert files["pred_box"].dtype == np.float32 and np.array_equal(files["pred_box"], g[tag + "_pred_box"])
    assert [bool(v) for v in files["pred_mask_index"]] == [bool(v) for v in g[tag + "_keep"]]
    if "scene" in files:
        assert hashlib.sha256(np.ascontiguousarray(files["scene"]).tobytes()).hexdigest() == str(g[tag + "_scene_sha"])
    want = unpack_masks(g, tag)
    assert len(files["pred_mask"]) == len(want)
    for a, b in zip(files["pred_mask"], want):
        assert a.dtype == np.float32 and np.array_equal(a, b)


@pytest.mark.parametrize("tag", ["geo", "img"])
def test_oracle_and_host_postprocessing_match_reference_fixture(golden, oracle, tag):
    from sis3d.model import trainval as tv
    g = golden("benchmark_small")
    cfg = cfg_for(tag, g)
    _, sd = state_dict(cfg)
    blobs = inputs(tag, cfg)
    o, kill = oracle_forward(oracle, cfg, sd, blobs)
    assert kill == ([2] if tag == "img" else [])
    check_files(oracle.benchmark_files(o, blobs["data"], cfg), g, tag)            # oracle == reference
    # product host logic on the oracle's predictions == reference
    pred_class, pred_conf, pred_box, keep = tv.final_detections(o, DIMS, cfg)
    assert 0 < sum(keep) and (tag == "geo" or sum(keep) < len(keep))
    assert tv.mask_windows(pred_box, keep) == [tuple(w) for w in o["_mask_aux"]["crops"]]
    masks = tv.binarise_masks(o["mask_pred"][0], pred_class, keep, cfg)
    check_files(dict(pred_class=pred_class, pred_conf=pred_conf, pred_box=pred_box, pred_mask_index=keep, pred_mask=masks), g, tag)


def test_final_detections_keep_rule():
    """confidence at / below CLASS_THRESH and boxes that collapse after rounding to voxels are dropped (trainval.py:702-712)"""
    from sis3d.model import trainval as tv
    cfg = config.scannet_benchmark_cfg()
    rois = torch.tensor([[4., 4., 4., 12., 12., 12.], [4., 4., 4., 12., 12., 12.], [4., 4., 4., 4.4, 12., 12.],
                         [0.5, 4., 4., 1.5, 12., 12.]])
    nc = cfg.NUM_CLASSES
    prob = torch.full((4, nc), 0.0)
    prob[0, 3], prob[1, 5], prob[2, 1], prob[3, 2] = 0.9, 0.5, 0.99, 0.99        # row 1: not > 0.5
    pred = dict(cls_pred=prob.argmax(1), rois=[rois], bbox_pred=torch.zeros(4, 6 * nc), cls_prob=prob)
    pc, conf, box, keep = tv.final_detections(pred, (32, 32, 32), cfg)
    assert pc.tolist() == [3, 5, 1, 2] and conf.dtype == np.float64 and box.dtype == np.float32
    # row 1: 0.5 is not > CLASS_THRESH; row 2: round(4.0) == round(4.4); row 3: half-to-even, round(0.5)=0 < round(1.5)=2
    assert keep == [True, False, False, True]
    assert tv.mask_windows(box, keep) == [(4, 4, 4, 12, 12, 12), (0, 4, 4, 2, 12, 12)]


def test_oracle_benchmark_vs_live_reference(oracle, tmp_path):
    import ref_harness as rh
    if not rh.available() or not os.path.exists(os.path.join(os.path.dirname(rh.__file__), "_ref", "libref_roi_pooling.so")):
        pytest.skip("reference tree / oracle/_ref not present on this machine")
    ns = rh.install(with_trainval=True)
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = True
    net = rh.build_net(ns, seed=0, use_images=True, use_mask=True)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_state_dict(shapes, seed=11, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    blobs = inputs("img", cfg)
    o, _ = oracle_forward(oracle, cfg, sd, blobs)
    want = oracle.benchmark_files(o, blobs["data"], cfg)
    got = rh.ref_benchmark(ns, net, [dict(blobs)], str(tmp_path))["scene0707_00"]
    for k in ("pred_class", "pred_conf", "pred_box", "scene"):
        assert np.array_equal(got[k], want[k]), k
    assert got["pred_mask_index"] == want["pred_mask_index"] and len(got["pred_mask"]) == len(want["pred_mask"])
    for a, b in zip(got["pred_mask"], want["pred_mask"]):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["geo", "img"])
def test_benchmark_mode_end_to_end(golden, oracle, tmp_path, tag):
    from sis3d.model.trainval import SolverWrapper
    g = golden("benchmark_small")
    cfg = cfg_for(tag, g)
    cfg.TEST_SAVE_DIR = str(tmp_path)
    net, sd = state_dict(cfg)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    blobs = inputs(tag, cfg)
    dirs = SolverWrapper.benchmark(net, [blobs], None)
    assert dirs == [str(tmp_path) + "/scene0707_00"]
    d = dirs[0]
    got = {k: np.load("%s/%s.npy" % (d, k)) for k in ("pred_class", "pred_conf", "pred_box", "scene")}
    for k in ("pred_mask", "pred_mask_index"):
        with open("%s/%s" % (d, k), "rb") as f:
            got[k] = pickle.load(f)
    o, _ = oracle_forward(oracle, cfg, sd, blobs)
    want = oracle.benchmark_files(o, blobs["data"], cfg)
    assert np.array_equal(got["scene"], want["scene"]) and got["scene"].dtype == want["scene"].dtype
    for k in ("pred_class", "pred_conf", "pred_box"):
        assert got[k].dtype == want[k].dtype and got[k].ndim == want[k].ndim
    # proposals: one-to-one with the oracle's, 0 near-ties asserted (tests/parity.py) -> the files compare ROW FOR ROW
    from parity import assert_proposals_match, report
    p = net._predictions
    assert_proposals_match(p["rois"][0].cpu(), p["roi_scores"][0].cpu(), o["rois"][0], o["roi_scores"][0], o["_scores_sorted_all"],
                           label="benchmark mode %s" % tag)
    assert got["pred_class"].shape == want["pred_class"].shape and np.array_equal(got["pred_class"], want["pred_class"])
    assert np.abs(got["pred_conf"] - want["pred_conf"]).max() <= 1e-4
    assert np.abs(got["pred_box"] - want["pred_box"]).max() <= 2e-3
    # no confidence within the 1e-4 tolerance of CLASS_THRESH and no box edge within 2e-3 of a rounding boundary for this seed:
    # same keep list, same integer crop windows
    assert np.abs(want["pred_conf"] - cfg.CLASS_THRESH).min() > 1e-4
    assert [bool(v) for v in got["pred_mask_index"]] == [bool(v) for v in want["pred_mask_index"]]
    assert len(got["pred_mask"]) == len(want["pred_mask"]) > 0
    # mask VALUES: the sigmoid outputs behind the written files, every class channel, <= 1e-4; threshold flips counted
    dev_masks, ora_masks = p["mask_pred"][0], o["mask_pred"][0]
    assert len(dev_masks) == len(ora_masks) == len(want["pred_mask"])
    flips = 0
    kept_cls = [int(c) for c, s in zip(want["pred_class"], want["pred_mask_index"]) if s]
    for dm, om, a, b, k in zip(dev_masks, ora_masks, got["pred_mask"], want["pred_mask"], kept_cls):
        assert dm.shape == om.shape and a.shape == b.shape == tuple(om.shape[2:])
        assert float((dm.cpu() - om).abs().max()) <= 1e-4
        assert np.array_equal(a, (dm[0, k].cpu().numpy() >= cfg.MASK_THRESH).astype(np.float32))     # the file is that channel, thresholded
        diff = a != b
        flips += int(diff.sum())
        assert np.all(np.abs(om[0, k].numpy()[diff] - cfg.MASK_THRESH) <= 1e-4)                      # a flip only where p is within 1e-4 of 0.5
    report("benchmark mode %s: %d masks <= 1e-4 on the sigmoid outputs, %d voxels flip at MASK_THRESH" % (tag, len(dev_masks), flips))
    # resume rule: existing pred_box.npy -> detection is not recomputed, masks are rebuilt from the stored boxes
    t0 = os.path.getmtime(d + "/pred_box.npy")
    net.delete_intermediate_states()
    SolverWrapper.benchmark(net, [inputs(tag, cfg)], None)
    assert os.path.getmtime(d + "/pred_box.npy") == t0
    with open(d + "/pred_mask", "rb") as f:
        again = pickle.load(f)
    assert len(again) == len(got["pred_mask"]) and all(np.array_equal(a, b) for a, b in zip(again, got["pred_mask"]))
