"""`.chunk/.scene` container + Dataset mirror against what the reference's own
Dataset.__getitem__ (lib/datasets/dataset.py:45-218) returned for tests/golden/synthetic.chunk
(fixture made by oracle/make_golden.py:dataset_case), and the device TSDF encoder against the same."""
import os

import numpy as np
import pytest
import torch

from sis3d import config
from sis3d.datasets import scene_file
from sis3d.datasets.dataset import Dataset, collate_fn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHUNK = os.path.join(GOLDEN, "synthetic.chunk")


def make_cfg(keep):
    c = config.scannet_benchmark_cfg()
    c.LABEL_MAP = os.path.join(GOLDEN, "synthetic_labels.csv")
    c.USE_IMAGES, c.USE_MASK, c.KEEP_THRESH = False, True, keep
    return c


def listing(tmp_path):
    p = tmp_path / "list.txt"
    p.write_text(CHUNK + "\n")
    return str(p)


def test_container_sections():
    sf = scene_file.SceneFile(CHUNK)
    assert sf.dims == (20, 52, 12) and sf.sdf.shape == (20 * 52 * 12,) and sf.sdf.dtype == np.float32
    assert sf.bytes_read == os.path.getsize(CHUNK)
    assert sf.boxes.shape == (5, 6) and sf.box_labels.tolist() == [3, 5, 7, 4, 38]
    assert [lab for lab, _ in sf.masks] == [3, 5, 7, 4, 38] and sf.masks[0][1].dtype == np.uint16
    assert np.allclose(sf.part_in_volume, [1.0, 0.4, 1.0, 0.97, 1.0])
    assert sf.frame_ids.tolist() == [17, 420, 9000]
    w = sf.world2chunk
    assert w.shape == (4, 4) and np.allclose(w[:3, 3], [3.0, -2.0, 8.5]) and np.isclose(w[0, 0], 21.333)
    # x fastest: element (x,y,z) at x + X*(y + Y*z)
    g = sf.sdf_grid()
    assert g[0, 0, 0] == -1.0 and g[1, 0, 0] == 3.0 and g[2, 0, 0] == -3.0 and sf.sdf[1] == 3.0


def test_write_read_round_trip(tmp_path):
    g = np.random.default_rng(1)
    sdf = g.standard_normal((5, 7, 3)).astype(np.float32)
    m = g.integers(0, 2, (2, 3, 4))
    p = str(tmp_path / "a.scene")
    scene_file.write_scene_file(p, sdf, [[0, 1, 2, 3, 4, 5]], [9], [(9, m)], [0.5], np.arange(16).reshape(4, 4), [1, 2])
    sf = scene_file.SceneFile(p)
    assert np.array_equal(sf.sdf_grid(), sdf) and np.array_equal(sf.masks[0][1], m)
    assert np.array_equal(sf.world2chunk, np.arange(16).reshape(4, 4)) and sf.frame_ids.tolist() == [1, 2]
    # geometry-only file (what a USE_MASK=False reader expects) and a truncated one
    scene_file.write_scene_file(p, sdf, [[0, 1, 2, 3, 4, 5]], [9], None)
    assert scene_file.SceneFile(p, want_masks=False).boxes.shape == (1, 6)
    with pytest.raises(scene_file.SceneFileError):
        scene_file.SceneFile(p)                                   # asks for the mask section: not enough bytes
    with open(p, "r+b") as f:
        f.truncate(100)
    with pytest.raises(scene_file.SceneFileError):
        scene_file.SceneFile(p, want_masks=False)


@pytest.mark.parametrize("mode", ["chunk", "benchmark", "scene"])
def test_dataset_matches_reference_fixture(golden, tmp_path, mode):
    g = golden("dataset_cases")
    ds = Dataset(listing(tmp_path), mode, make_cfg(float(g[mode + "_keep_thresh"])))
    assert len(ds) == 1
    r = ds[0]
    assert r["id"] == CHUNK and r["data"].dtype == np.float32
    assert np.array_equal(r["data"], g[mode + "_data"])
    assert r["gt_box"].dtype == g[mode + "_gt_box"].dtype and np.array_equal(r["gt_box"], g[mode + "_gt_box"])
    assert len(r["gt_mask"]) == int(g[mode + "_n_mask"])
    for i, m in enumerate(r["gt_mask"]):
        assert m.dtype == np.uint8 and np.array_equal(m, g["%s_mask_%d" % (mode, i)])
    b = collate_fn([r])
    assert tuple(b["data"].shape) == (1,) + r["data"].shape and len(b["gt_box"]) == 1 and len(b["gt_mask"][0]) == len(r["gt_mask"])


def test_dataset_vs_live_reference(tmp_path):
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install(with_trainval=True)
    from lib.datasets.dataset import Dataset as RefDataset
    rng = np.random.default_rng(3)
    sdf = (rng.standard_normal((9, 50, 11)) * 2).astype(np.float32)
    boxes = np.array([[0.2, 0.3, 0.4, 5.5, 47.5, 6.5], [1, 1, 1, 4, 49.5, 4]], dtype=np.float32)
    masks = [(3, rng.integers(0, 3, (6, 48, 7))), (4, rng.integers(0, 3, (3, 49, 3)))]
    p = str(tmp_path / "s.chunk")
    scene_file.write_scene_file(p, sdf, boxes, [3, 4], masks, [1.0, 1.0])
    lst = tmp_path / "l.txt"
    lst.write_text(p + "\n")
    cfg = ns.cfg
    saved = (cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH, cfg.FLIP_TSDF)
    try:
        cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH = os.path.join(GOLDEN, "synthetic_labels.csv"), False, True, 0.9
        for flip in (False, True):
            cfg.FLIP_TSDF = flip
            want = RefDataset(str(lst), "chunk")[0]
            c = make_cfg(0.9)
            c.FLIP_TSDF = flip
            got = Dataset(str(lst), "chunk", c)[0]
            assert np.array_equal(got["data"], want["data"]) and np.array_equal(got["gt_box"], want["gt_box"])
            assert len(got["gt_mask"]) == len(want["gt_mask"]) and all(np.array_equal(a, b) for a, b in zip(got["gt_mask"], want["gt_mask"]))
    finally:
        cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH, cfg.FLIP_TSDF = saved


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["chunk", "benchmark"])
def test_device_tsdf_encode_matches_reference_fixture(golden, tmp_path, mode):
    from sis3d import ops
    g = golden("dataset_cases")
    ds = Dataset(listing(tmp_path), mode, make_cfg(float(g[mode + "_keep_thresh"])), device_encode=True)
    r = ds[0]
    want = g[mode + "_data"]
    assert r["data"].is_cuda and tuple(r["data"].shape) == want.shape
    assert np.array_equal(r["data"].cpu().numpy(), want)                       # bit-exact, -inf voxels included
    blobs = collate_fn([r])
    # planar (1,2,X,Y,Z) with contiguous z: what the first-layer kernels (geometry1.0, mask conv0) read
    assert blobs["data"].is_contiguous() and not ops.is_cl(blobs["data"])
    sf = scene_file.SceneFile(CHUNK)
    raw = torch.from_numpy(np.ascontiguousarray(sf.sdf)).cuda()
    planar = ops.tsdf_encode(raw, sf.dims, 3.0, "abs", want.shape[2], channels_last=False)
    assert planar.is_contiguous() and np.array_equal(planar[0].cpu().numpy(), want)
    flip = ops.tsdf_encode(raw, sf.dims, 3.0, "flip", want.shape[2])
    assert np.array_equal(flip[0, 0].cpu().numpy(), 3.0 - want[0]) and np.array_equal(flip[0, 1].cpu().numpy(), want[1])
    lg = ops.tsdf_encode(raw, sf.dims, 3.0, "log", want.shape[2])[0, 0].cpu().numpy()
    ref = np.log(want[0])
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(lg), fin) and np.abs(lg[fin] - ref[fin]).max() <= 1e-6   # logf vs np.log: <= 1 ulp near 1


@pytest.mark.gpu
def test_device_tsdf_encode_chunk_size():
    """full 96x48x96 chunk and an odd whole-scene grid vs numpy"""
    from sis3d import ops
    rng = np.random.default_rng(0)
    for dims, mh in (((96, 48, 96), 48), ((131, 70, 77), 64)):
        sdf = (rng.standard_normal(dims) * 3).astype(np.float32)
        want = np.concatenate([np.abs(np.clip(sdf[None], -3, 3)), np.greater(sdf[None], -1)], 0)[:, :, :mh, :]
        raw = torch.from_numpy(np.ascontiguousarray(sdf.reshape(-1, order="F"))).cuda()
        got = ops.tsdf_encode(raw, dims, 3.0, "abs", mh)
        assert np.array_equal(got[0].cpu().numpy(), want)


@pytest.mark.gpu
def test_device_encoded_sample_runs_through_the_network(tmp_path, oracle):
    """.chunk file -> Dataset(device_encode=True) -> collate_fn -> Network.forward on the GPU, against the oracle fed with
    the host-encoded sample of the same file (the device-encoded layout must be one the first layer accepts)"""
    from sis3d import config, synthetic
    from sis3d.nets import backbones
    rng = np.random.default_rng(11)
    sdf = (rng.standard_normal((32, 48, 24)) * 2).astype(np.float32)
    p = str(tmp_path / "n.chunk")
    scene_file.write_scene_file(p, sdf, np.zeros((0, 6), np.float32), [], [], [])
    lst = tmp_path / "l.txt"
    lst.write_text(p + "\n")
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    cfg.LABEL_MAP = ""
    cfg.KEEP_THRESH = 0
    host = collate_fn([Dataset(str(lst), "chunk", cfg)[0]])
    dev = collate_fn([Dataset(str(lst), "chunk", cfg, device_encode=True)[0]])
    assert dev["data"].is_cuda and torch.equal(dev["data"].cpu(), host["data"])
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    sd = synthetic.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    net.cuda().eval()
    pred = net.forward(dev, "TEST", [])
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(host["data"])
    l1, l2 = net._net_conv
    assert float((l1.cpu() - o["level1"]).abs().max()) <= 1e-4 and float((l2.cpu() - o["level2"]).abs().max()) <= 1e-4
    assert float((pred["rpn_bbox_pred_level2"].cpu() - o["rpn_bbox_pred_level2"]).abs().max()) <= 1e-4
    assert abs(pred["rois"][0].shape[0] - o["rois"][0].shape[0]) <= 2


# ------------------------------------------------------------------ per-frame files (dataset.py:136-190, 230-267)
FRAME_CASES = (("chunk_color", "chunk", "color", ".jpg", [328, 256], [41, 32], False),
               ("chunk_crop", "chunk", "color", ".jpg", [100, 90], [30, 30], False),
               ("scene_label", "scene", "label", ".png", [41, 32], [41, 32], True))


@pytest.mark.parametrize("case", FRAME_CASES, ids=[c[0] for c in FRAME_CASES])
def test_frames_match_reference_fixture(case, tmp_path):
    """Dataset(USE_IMAGES) + collate_fn against what the reference's own Dataset / collate_fn returned for the committed
    frame directory (oracle/make_golden.py:frames_case): depth maps, normalised colour images (or relabelled label
    images), poses, world2grid -- bit for bit, frames matched by id (os.listdir order is file-system dependent)."""
    import hashlib
    tag, mode, itype, ext, ishape, dshape, gt = case
    g = np.load(os.path.join(GOLDEN, "dataset_frames_cases.npz"))
    c = config.scannet_benchmark_cfg()
    c.LABEL_MAP = os.path.join(GOLDEN, "synthetic_labels.csv")
    c.BASE_IMAGE_PATH = os.path.join(GOLDEN, "frames_square")
    c.USE_IMAGES, c.USE_MASK, c.KEEP_THRESH, c.MODE, c.NUM_IMAGES = True, True, 0.0, "benchmark", 5
    c.IMAGE_TYPE, c.IMAGE_EXT, c.IMAGE_SHAPE, c.DEPTH_SHAPE, c.USE_IMAGES_GT = itype, ext, ishape, dshape, gt
    lst = tmp_path / "l.txt"
    lst.write_text(os.path.join(GOLDEN, "scene0000_00__0.chunk") + "\n")
    r = Dataset(str(lst), mode, c)[0]
    ni = r["nearest_images"]
    want_ids = [int(v) for v in g[tag + "_frameids"]]
    got_ids = [int(v) for v in ni["frameids"]]
    assert sorted(got_ids) == sorted(want_ids) == [17, 420, 9000]
    perm = [got_ids.index(f) for f in want_ids]                  # our position of the fixture's k-th frame
    assert np.array_equal(ni["world2grid"], g[tag + "_world2grid"]) and ni["world2grid"].dtype == g[tag + "_world2grid"].dtype
    assert np.array_equal(np.stack(ni["poses"])[perm], g[tag + "_poses"])
    depths = np.stack(ni["depths"])[perm]
    assert depths.dtype == np.float32 and np.array_equal(depths, g[tag + "_depths"])
    imgs = np.stack([np.asarray(i) for i in ni["images"]]).astype(np.float32)[perm]
    if tag + "_images" in g.files:
        assert np.array_equal(imgs, g[tag + "_images"])
    else:
        assert np.array_equal(imgs[:, :, ::16, ::16], g[tag + "_images_sub"])
        digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(imgs).tobytes()).digest(), dtype=np.uint8)
        assert np.array_equal(digest, g[tag + "_images_sha"])
    assert np.array_equal(r["gt_box"], g[tag + "_gt_box"])
    files = [os.path.relpath(p, GOLDEN) for p in r["image_files"]]
    assert [files[k] for k in perm] == [str(v) for v in g[tag + "_image_files"]]
    blobs = collate_fn([r], c)
    nb = blobs["nearest_images"]
    assert tuple(nb["images"][0].shape) == tuple(g[tag + "_blob_images_shape"]) and nb["images"][0].dtype == torch.float32
    assert np.array_equal(nb["world2grid"][0].numpy(), g[tag + "_blob_world2grid"])
    assert nb["depths"][0].shape == (3, dshape[1], dshape[0]) and nb["poses"][0].shape == (3, 4, 4)


def test_frame_loaders_edge_cases(tmp_path):
    from PIL import Image
    from sis3d.datasets import frames
    # same size: returned untouched; palette and bilevel images as scipy.misc.imread expands them
    a = (np.arange(12 * 8).reshape(8, 12) % 7).astype(np.uint8)
    assert frames.resize_crop_image(a, [12, 8]) is a
    p = str(tmp_path / "p.png")
    Image.fromarray(a).convert("P").save(p)
    assert frames.imread(p).shape == (8, 12, 3)
    Image.fromarray((a > 3)).save(p)
    assert frames.imread(p).dtype == np.uint8 and frames.imread(p).shape == (8, 12)
    # nearest resize picks source pixel floor((i + 0.5) * in / out); the crop is centred with round-half-even
    wide = np.tile(np.arange(40, dtype=np.uint8), (10, 1))
    out = frames.resize_crop_image(wide, [10, 5])                 # height 10 -> 5, width 40 -> 20, crop 10 around the centre
    assert out.shape == (5, 10) and out[0].tolist() == [2 * (5 + i) + 1 for i in range(10)]
    with open(str(tmp_path / "pose.txt"), "w") as f:
        f.write("1 0 0 0.5\n0 1 0 -2\n0 0 1 3 extra\n0 0 0 1\n")
    assert frames.load_pose(str(tmp_path / "pose.txt")).tolist() == [[1, 0, 0, 0.5], [0, 1, 0, -2], [0, 0, 1, 3], [0, 0, 0, 1]]
    with open(str(tmp_path / "bad.txt"), "w") as f:
        f.write("1 0 0 0\n")
    with pytest.raises(AssertionError):
        frames.load_pose(str(tmp_path / "bad.txt"))


@pytest.mark.gpu
def test_forward_from_frame_files(tmp_path, oracle):
    """Files on disk -> Dataset(USE_IMAGES) -> collate_fn -> prepare_projection -> Network.forward with the ENet encoder on
    the GPU (lib/model/trainval.py:659-686 flow), against the oracle fed with the same loaded blobs: depth PNGs + poses ->
    visibility lists bit for bit, colour JPEGs -> ENet -> back-projection -> backbone -> RPN -> class head."""
    from PIL import Image
    from sis3d import synthetic
    from sis3d.layer_utils.projection import prepare_projection
    from sis3d.nets import backbones
    from parity import assert_proposals_match
    dims, V = (64, 32, 48), 3
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES, cfg.USE_IMAGES_GT, cfg.USE_MASK, cfg.KEEP_THRESH, cfg.LABEL_MAP = True, False, False, 0.0, ""
    cfg.BASE_IMAGE_PATH = str(tmp_path / "frames_square")
    root = tmp_path / "frames_square" / "scene0042_00"
    for d in ("depth", "color", "pose"):
        os.makedirs(str(root / d))
    depth, c2w, w2g = synthetic.synth_cameras(31, V, dims, cfg.VOXEL_SIZE)
    w0 = w2g[0].double()
    ids = [5, 60, 700]
    rgb = (synthetic.synth_images(9, V) * 60 + 128).clamp(0, 255).byte()
    for v, fid in enumerate(ids):
        Image.fromarray((depth[v] * 1000).round().numpy().astype(np.uint16)).save(str(root / "depth" / ("%d.png" % fid)))
        Image.fromarray(rgb[v].permute(1, 2, 0).contiguous().numpy()).save(str(root / "color" / ("%d.jpg" % fid)), quality=90)
        pose = torch.linalg.inv(w0) @ w2g[v].double() @ c2w[v].double()        # the same camera in view 0's world frame
        with open(str(root / "pose" / ("%d.txt" % fid)), "w") as f:
            for r in pose.tolist():
                f.write(" ".join("%.9g" % x for x in r) + "\n")
    rng = np.random.default_rng(3)
    sdf = (rng.standard_normal(dims) * 2).astype(np.float32)
    chunk = str(tmp_path / "scene0042_00__3.chunk")
    scene_file.write_scene_file(chunk, sdf, np.zeros((0, 6), np.float32), [], [], [], torch.linalg.inv(w0).float().numpy().T, ids)
    lst = tmp_path / "l.txt"
    lst.write_text(chunk + "\n")
    blobs = collate_fn([Dataset(str(lst), "chunk", cfg)[0]], cfg)
    ni = blobs["nearest_images"]
    images, depths, poses, w2gs = ni["images"][0], ni["depths"][0], ni["poses"][0], ni["world2grid"][0]
    assert tuple(images.shape) == (V, 3, 256, 328) and tuple(depths.shape) == (V, 32, 41)
    assert (w2gs[0] - w2g[0]).abs().max() <= 1e-3
    maps = [oracle.compute_projection(depths[v], poses[v], w2gs[v], cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX,
                                      cfg.DEPTH_SHAPE, dims, cfg.VOXEL_SIZE) for v in range(V)]
    want_kill = [v for v, m in enumerate(maps) if m is None]
    o3 = torch.stack([m[0] for m in maps if m is not None])
    o2 = torch.stack([m[1] for m in maps if m is not None])
    assert int(o3[:, 0].sum()) > 500                                           # the rig really sees the chunk
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    sd = synthetic.synth_checkpoint({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=True)
    net.cuda().eval()
    kill = prepare_projection(blobs, cfg)
    assert kill == want_kill
    assert torch.equal(blobs["proj_ind_3d"][0].cpu(), o3) and torch.equal(blobs["proj_ind_2d"][0].cpu(), o2)
    p = net.forward(blobs, "TEST", kill)
    torch.cuda.synchronize()
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(blobs["data"], images, o3, o2,
                                                                                                   killing_inds=want_kill)
    l1, l2 = net._net_conv
    scale = max(1.0, float(o["level1"].abs().max()))
    assert float((l1.cpu() - o["level1"]).abs().max()) <= 1e-4 * scale and float((l2.cpu() - o["level2"]).abs().max()) <= 1e-4 * scale
    for lv in (1, 2):
        assert float((p["rpn_cls_prob_level%d" % lv].cpu() - o["rpn_cls_prob_level%d" % lv]).abs().max()) <= 1e-4
    assert_proposals_match(p["rois"][0].cpu(), p["roi_scores"][0].cpu(), o["rois"][0], o["roi_scores"][0], o["_scores_sorted_all"],
                           label="from frame files")
