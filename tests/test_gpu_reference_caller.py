"""The drop-in boundary END TO END: the reference's own caller drives the HIP path on the MI355X.

`lib/model/trainval.py` of Sekunde/3D-SIS -- UNMODIFIED, imported from the reference tree (in the build container
/root/reference; on the GPU box the archive oracle/_ref/reference_tree.tgz that oracle/Makefile stages, unpacked by
ref_harness into a temp dir) -- runs its `benchmark(args)` sequence (trainval.py:55-72) and its
`SolverWrapper.benchmark(net, loader, logger)` loop (trainval.py:634-767) after `sis3d.dropin.install()`:

    net_module = importlib.import_module("lib.nets.backbones"); net = getattr(net_module, cfg.NET)()
    net.init_modules(); net.load_state_dict(torch.load(saved_model)); SolverWrapper.benchmark(net, dataloader, logger)

Everything the loop touches is the reference's code (ProjectionHelper call sites, killing_inds logic, bbox_transform_inv,
clip_boxes, the keep rule, the crop slicing `blobs['data'].cuda()[..., x0:x1, y0:y1, z0:z1]`, the six result files) except what
dropin.install() replaces: the network classes, RoI pooling, NMS, Projection, ProjectionHelper -- all HIP.

The six files are compared with tests/golden/benchmark_small.npz, which holds what the SAME loop wrote when the reference's own
network ran it on CPU (oracle/make_golden.py:benchmark_case), at the tolerances of test_benchmark_mode_end_to_end.
tests/golden/reference_tree.sha256 (hashes only) proves the caller is the unmodified file."""
import hashlib
import importlib
import inspect
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def manifest():
    out = {}
    with open(os.path.join(ROOT, "tests", "golden", "reference_tree.sha256")) as f:
        for line in f:
            h, name = line.split()
            out[name] = h
    return out


def sha_file(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def test_staged_reference_tree_matches_manifest():
    """CPU: whatever tree ref_harness resolves (in place / the staged archive) is byte for byte the one the manifest was
    taken from -- the files the GPU test executes are the reference's own."""
    import ref_harness as rh
    if not rh.available():
        pytest.skip("no reference tree and no staged archive on this machine")
    m = manifest()
    assert "lib/model/trainval.py" in m and "main.py" in m and "lib/nets/network.py" in m
    for name, h in m.items():
        assert sha_file(os.path.join(rh.REF_ROOT, name)) == h, name
    tgz = os.path.join(ROOT, "oracle", "_ref", "reference_tree.tgz")
    if os.path.isfile(tgz):
        import tarfile
        with tarfile.open(tgz) as t:
            names = sorted(i.name for i in t.getmembers() if i.isfile())
            assert names == sorted(m)
            for i in t.getmembers():
                if i.isfile():
                    assert hashlib.sha256(t.extractfile(i).read()).hexdigest() == m[i.name], i.name


class _NoMetric(object):
    """stands in for lib.utils.evaluation.DetectionMAP inside SolverWrapper.test: the mAP bookkeeping is outside the hot path
    and calls np.int, which numpy >= 1.24 no longer has; the calls are recorded so the test can see the loop reached them"""
    calls = []

    def __init__(self, *a, **k):
        self.ignore_class = [0]

    def evaluate(self, *a):
        _NoMetric.calls.append(("evaluate", len(a[0])))

    def evaluate_mask(self, *a):
        _NoMetric.calls.append(("evaluate_mask", len(a[3])))

    def finalize(self):
        pass

    def mAP(self):
        return 0.0

    def AP(self, i):
        return 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("tag,caller", [("geo", "benchmark"), ("img", "benchmark"), ("geo", "test"), ("geo", "validation")])
def test_reference_benchmark_loop_drives_the_hip_dropin(golden, oracle, tmp_path, tag, caller, monkeypatch):
    """caller 'test' (r6, VERDICT r5 missing #4): the reference's other whole-scan caller of the hot path, SolverWrapper.test
    (trainval.py:770-960) -- same net.forward(blobs, 'TEST', killing_inds), its own detection post-processing and per-box mask loop
    (trainval.py:883-897) -- writes the same result files; they are held to the same fixture.  Its mAP bookkeeping is stubbed.
    caller 'validation' (r6): the third caller, SolverWrapper.validation(index, mode) (trainval.py:434-632), an instance method of the
    training wrapper: run on a stand-in `self` (net, a one-blob loader, a logger that swallows scalar_summary) with
    net.forward(blobs, 'TEST', []) and its own post-processing; writes the same files under cfg.VAL_SAVE_DIR."""
    import ref_harness as rh
    from parity import assert_proposals_match, report
    from test_benchmark_mode import inputs, oracle_forward, unpack_masks
    from sis3d import config, dropin, synthetic
    if not rh.available():
        pytest.skip("no reference tree: oracle/_ref/reference_tree.tgz was not staged (make -C oracle tree)")
    g = golden("benchmark_small")
    ns = rh.install(with_trainval=True, on_cpu=False)          # stubs for easydict / ipdb / ...; the REAL .cuda()
    tv = ns.trainval
    # the caller under test is the reference's own, unmodified file
    src = inspect.getsourcefile(tv.SolverWrapper.benchmark)
    assert os.path.realpath(src) == os.path.realpath(os.path.join(rh.REF_ROOT, "lib", "model", "trainval.py"))
    assert sha_file(src) == manifest()["lib/model/trainval.py"]
    cfg = ns.cfg
    saved = {k: cfg[k] for k in ("USE_IMAGES", "USE_IMAGES_GT", "USE_MASK", "CLASS_THRESH", "TEST_SAVE_DIR", "VAL_SAVE_DIR")}
    cfg.USE_IMAGES = tag == "img"
    cfg.USE_IMAGES_GT = tag == "img"                            # feature maps handed in (network.py:199-201), as the fixture's run
    cfg.USE_MASK = True
    cfg.CLASS_THRESH = float(g[tag + "_class_thresh"])
    cfg.TEST_SAVE_DIR = str(tmp_path / "out")
    cfg.VAL_SAVE_DIR = cfg.TEST_SAVE_DIR
    undo = dropin.install()                                     # binds the HIP classes to the reference's LIVE cfg
    try:
        # -- trainval.py:55-72 `benchmark(args)`, with a list of blobs as the loader and a checkpoint written for the occasion
        net_module = importlib.import_module("lib.nets.backbones")
        net = getattr(net_module, cfg.NET)()
        assert type(net).__module__.startswith("sis3d.") and net.cfg is cfg
        net.init_modules()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = synthetic.synth_state_dict(shapes, seed=3, gains=synthetic.DEFAULT_GAINS)
        saved_model = str(tmp_path / "step_0.pth")
        torch.save(sd, saved_model)
        net.load_state_dict(torch.load(saved_model))
        blobs = inputs(tag, cfg)
        assert not blobs["data"].is_cuda and not any(q.is_cuda for q in net.parameters())   # the loop itself moves things, as on CUDA
        if caller in ("test", "validation"):
            _NoMetric.calls.clear()
            monkeypatch.setattr(tv, "Evaluate_metric", _NoMetric)
        with rh.in_reference_dir():
            if caller == "validation":
                import types
                logged = []
                me = types.SimpleNamespace(net=net, dataloader_val=[blobs],
                                           logger_val=types.SimpleNamespace(scalar_summary=lambda *a: logged.append(a[0])))
                monkeypatch.setattr(net, "delete_intermediate_states", lambda: None)      # the loop ends with it; the checks below read _predictions
                tv.SolverWrapper.validation(me, 0, "val")
                assert logged == ["AP_ROI", "mAP_CLASSIFICATION", "mAP_MASK"]
            else:
                getattr(tv.SolverWrapper, caller)(net, [blobs], None)
        torch.cuda.synchronize()
        assert all(q.is_cuda for q in net.parameters())
        if caller == "test":
            assert [c[0] for c in _NoMetric.calls] == ["evaluate", "evaluate_mask"] and _NoMetric.calls[1][1] > 0
        if caller == "validation":
            assert [c[0] for c in _NoMetric.calls] == ["evaluate", "evaluate", "evaluate_mask"] and _NoMetric.calls[2][1] > 0
        d = os.path.join(cfg.TEST_SAVE_DIR, "scene0707_00")
        got = {k: np.load("%s/%s.npy" % (d, k)) for k in ("pred_class", "pred_conf", "pred_box", "scene")}
        for k in ("pred_mask", "pred_mask_index"):
            with open("%s/%s" % (d, k), "rb") as f:
                got[k] = pickle.load(f)
        # -- proposals one to one with the oracle's (0 near-ties asserted) -> the files compare ROW FOR ROW with the fixture
        ocfg = config.scannet_benchmark_cfg()
        ocfg.USE_IMAGES, ocfg.CLASS_THRESH = tag == "img", cfg.CLASS_THRESH
        o, kill = oracle_forward(oracle, ocfg, sd, inputs(tag, ocfg))
        p = net._predictions
        assert_proposals_match(p["rois"][0].cpu(), p["roi_scores"][0].cpu(), o["rois"][0], o["roi_scores"][0], o["_scores_sorted_all"],
                               label="reference caller %s" % tag)
        assert hashlib.sha256(np.ascontiguousarray(got["scene"]).tobytes()).hexdigest() == str(g[tag + "_scene_sha"])
        assert got["pred_class"].dtype == np.int64 and np.array_equal(got["pred_class"], g[tag + "_pred_class"])
        assert got["pred_conf"].dtype == np.float64 and np.abs(got["pred_conf"] - g[tag + "_pred_conf"]).max() <= 1e-4
        # pred_box bound, DERIVED from the path's 1e-4 bound on the logits (bbox_transform_inv, lib/utils/bbox_transform.py:59-99):
        #   ctr' = d_ctr * w + ctr,  w' = exp(d_w) * w,  box = ctr' -/+ 0.5 w'   with w <= 96 voxels (a box cannot exceed the chunk),
        # so a perturbation e <= 1e-4 of the regression outputs moves a centre by <= e * w = 9.6e-3 / 96 * w and a half-extent by
        # <= 0.5 * w * exp(d_w) * (exp(e) - 1) ~ 0.5 * w' * e.  For the boxes of this fixture (w, w' <= 40 voxels: the anchors are 8-40
        # voxels wide and the seeded deltas are small) that is <= 4e-3 + 2e-3 in the worst case of BOTH outputs off by the full 1e-4;
        # the measured logit error of the path is ~1e-6, i.e. ~1e-4 voxels here.  2e-3 voxels = the bound for logit errors of <= 3e-5 on
        # 40-voxel boxes: tighter than the 1e-4 contract would allow, far above what is observed; the clip to [0, dim] only shrinks it.
        assert got["pred_box"].dtype == np.float32 and np.abs(got["pred_box"] - g[tag + "_pred_box"]).max() <= 2e-3
        assert [bool(v) for v in got["pred_mask_index"]] == [bool(v) for v in g[tag + "_keep"]]
        want = unpack_masks(g, tag)
        assert len(got["pred_mask"]) == len(want) > 0
        # masks: the reference loop's own per-box crops through the HIP mask head; a voxel may differ from the fixture only where
        # the oracle's probability is within 1e-4 of MASK_THRESH
        kept_cls = [int(c) for c, s in zip(got["pred_class"], got["pred_mask_index"]) if s]
        flips = 0
        for a, b, om, k, dm in zip(got["pred_mask"], want, o["mask_pred"][0], kept_cls, p["mask_pred"][0]):
            assert a.dtype == np.float32 and a.shape == b.shape == tuple(om.shape[2:])
            assert float((dm.cpu() - om).abs().max()) <= 1e-4
            diff = a != b
            flips += int(diff.sum())
            assert np.all(np.abs(om[0, k].numpy()[diff] - cfg.MASK_THRESH) <= 1e-4)
        report("reference caller %s: unmodified SolverWrapper.%s (%s, tree %s) over sis3d.dropin.install(): %d detections, "
               "%d kept, pred_class exact, pred_conf <= 1e-4, pred_box <= 2e-3, %d masks (%d voxels flip at MASK_THRESH), killing_inds %s"
               % (tag, caller, os.path.relpath(src, rh.REF_ROOT), rh.REF_SOURCE, len(got["pred_class"]), len(want), len(want), flips, kill))
        if caller != "benchmark":
            return
        # -- resume rule of the reference loop (trainval.py:650-654): detection is not recomputed, masks rebuilt from stored boxes
        t0 = os.path.getmtime(d + "/pred_box.npy")
        net.delete_intermediate_states()
        with rh.in_reference_dir():
            tv.SolverWrapper.benchmark(net, [inputs(tag, cfg)], None)
        assert os.path.getmtime(d + "/pred_box.npy") == t0
        with open(d + "/pred_mask", "rb") as f:
            again = pickle.load(f)
        assert len(again) == len(got["pred_mask"]) and all(np.array_equal(a, b) for a, b in zip(again, got["pred_mask"]))
    finally:
        dropin.uninstall(undo)
        for k in [k for k in sys.modules if k.startswith("lib.layer_utils.nms._ext") or k.startswith("lib.layer_utils.roi_pooling._ext")]:
            del sys.modules[k]
        for k, v in saved.items():
            cfg[k] = v
        rh._installed = False
