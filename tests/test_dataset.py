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
