"""bench.py's `cpu_baseline` leg: the reference run in place (kind "reference") and the oracle port (kind "port") timed on the
GPU box's host cores.  The ONLY part of the bench that imports oracle/ -- as the thing timed beside the product, never as the
product (tests/test_abi_and_host.py checks where oracle/ is imported)."""
import json
import os
import subprocess
import sys
import time

from .constants import *  # noqa: F401,F403

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_threads_rule():
    """ONE stated rule for the CPU baseline's thread count (the 16/32/64/128 sweep moved the figure 2x between
    boxes): the physical cores of one socket of the host, capped at 64 -- oneDNN's 3D convolutions on a 0.1-GFLOP/voxel chunk stop
    scaling there, SMT siblings and the second socket only add contention.  SIS3D_CPU_THREADS overrides."""
    env = os.environ.get("SIS3D_CPU_THREADS")
    if env:
        return max(1, int(env)), "SIS3D_CPU_THREADS"
    cores = os.cpu_count() or 1
    try:
        phys, sockets = set(), set()
        with open("/proc/cpuinfo") as f:
            pid = cid = None
            for ln in f:
                if ln.startswith("physical id"):
                    pid = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":")[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                        sockets.add(pid)
                    pid = cid = None
        if phys:
            per_socket = max(1, len(phys) // max(1, len(sockets)))
            return min(64, per_socket), "physical cores of one socket (%d sockets x %d cores, %d logical), capped at 64" % (
                len(sockets), per_socket, cores)
    except Exception:
        pass
    return min(64, max(1, cores // 2)), "half of the logical CPUs, capped at 64 (no /proc/cpuinfo topology)"


def _median_runs(fn, budget_s, min_runs=10, max_runs=200):
    """median wall time of fn over >= min_runs runs (one untimed warm-up), stopping after budget_s once min_runs are in"""
    fn()
    ts, t_end = [], time.time() + budget_s
    while len(ts) < min_runs or (time.time() < t_end and len(ts) < max_runs):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    ts.sort()
    return ts[len(ts) // 2], len(ts), ts[0], ts[-1]


def cpu_baseline_reference(workload, sd, cfg, seconds, threads):
    """BASELINE config[0]: the REFERENCE's own `Network.forward(blobs, 'TEST', [])` (lib/nets/network.py:187-317) timed in place on
    the host cores -- the README's MAX_VOLUME=0 CPU path in full (`.cuda()` neutralised, the reference's own roi_pooling.c for
    RoIPoolFunction: oracle/ref_harness.py), same seeded weights and the same synthetic chunk as the GPU run.  Runs from
    /root/reference in the build container and from the staged archive oracle/_ref/reference_tree.tgz on the GPU box (verified
    against tests/golden/reference_tree.sha256 by ref_harness).  -> dict | None (reference not available)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import ref_harness as rh
    except Exception:
        return None
    if not rh.available() or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_roi_pooling.so")):
        return None
    from sis3d import synthetic
    use_images = workload == "images"
    ns = rh.install()
    try:
        net = rh.build_net(ns, seed=0, use_images=use_images, use_mask=False)
        missing = [k for k in net.state_dict() if k not in sd]
        if missing:
            return {"error": "reference net has parameters the synthetic checkpoint lacks: %s" % missing[:3]}
        net.load_state_dict({k: sd[k] for k in net.state_dict()})
        data = synthetic.synth_chunk(0)
        if use_images:
            feats, i3d, i2d = synthetic.synth_views(0)
            blobs = rh.make_blobs(data, feats, i3d, i2d)
        else:
            blobs = rh.make_blobs(data)
        torch.set_num_threads(threads)
        per, n, lo, hi = _median_runs(lambda: rh.forward(ns, net, blobs), seconds)
        rois = int(net._predictions["rois"][0].shape[0]) if "rois" in net._predictions else None
    finally:
        rh.restore_cuda()
    return {"value": VOXELS / per, "unit": "voxels/s", "cores": threads, "kind": "reference", "runs": n,
            "ms_per_chunk_median": per * 1e3, "ms_per_chunk_min_max": [lo * 1e3, hi * 1e3], "rois": rois,
            "reference_from": rh.REF_SOURCE,
            "what": "the reference's unmodified Network.forward TEST branch (backbone + RPN + proposal_layer/cpu_nms + RoI pooling (its own "
                    "roi_pooling.c) + classifier), torch-CPU operators, on one 96x48x96 synthetic chunk"}


def cpu_baseline(workload, sd, cfg, seconds):
    """CPU baseline beside the GPU number (reported, never the target).  kind "reference" = the reference itself run in place
    (cpu_baseline_reference) when its tree is available, with the oracle port's figure of the SAME workload beside it under `port`;
    kind "port" (the oracle, torch-CPU operators = what the reference's MAX_VOLUME=0 path runs) otherwise.  Thread count: one stated
    rule (cpu_threads_rule), median of >= 10 runs; the single-thread figure and a per-stage table from the
    port."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sis3d_oracle as orc
    from sis3d import config, synthetic
    cores = os.cpu_count() or 1
    threads, rule = cpu_threads_rule()
    threads = max(1, min(threads, cores))
    net = orc.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    data = synthetic.synth_chunk(0)
    feats = i3d = i2d = None
    if workload == "images":
        feats, i3d, i2d = synthetic.synth_views(0)
        if not cfg["USE_IMAGES_GT"]:
            feats = synthetic.synth_images(0, cfg["NUM_IMAGES"])

    views = feats                                            # what one() feeds the net (the stage table re-binds feats under --rgb)

    def one():
        with torch.no_grad():
            if workload in ("detect", "scene") or not cfg["USE_IMAGES_GT"]:
                net.forward(data, views, i3d, i2d)
            else:
                imageft = orc.project_views_max(feats, i3d, i2d, data.shape[2:]) if workload == "images" else None
                l1, l2 = net.backbone(data, imageft)
                net.rpn(l1, 1)
                net.rpn(l2, 2)

    def timed(fn, budget, max_n=400):
        fn()
        n, t0 = 0, time.time()
        while True:
            fn()
            n += 1
            if time.time() - t0 >= budget or n >= max_n:
                break
        return (time.time() - t0) / n, n

    ref = None
    try:
        ref = cpu_baseline_reference(workload, sd, cfg, seconds * 0.35, threads)
    except Exception as e:                                   # the reported baseline must never take the line down
        ref = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.set_num_threads(threads)
    per, n, lo, hi = _median_runs(one, seconds * (0.25 if ref and "value" in ref else 0.5))
    # per-stage table at the same thread count, then the whole forward at one thread
    stages = {}
    with torch.no_grad():
        if workload == "images" and not cfg["USE_IMAGES_GT"]:
            # --rgb: the views are RGB images; the stage table below starts from the encoder's feature maps (network.py:203-205)
            rgb = feats
            feats = orc.enet_features(sd, rgb)
            d, k = timed(lambda: orc.enet_features(sd, rgb), seconds * 0.05, 5)
            stages["enet_encoder_5_views"] = {"ms": d * 1e3, "threads": threads, "runs": k}
        l1, l2 = net.backbone(data, None) if workload != "images" else net.backbone(data, orc.project_views_max(feats, i3d, i2d, data.shape[2:]))
        o = None
        if not cfg["USE_IMAGES"]:
            o = net.forward(data)
        share = seconds * 0.25 / 6.0

        def st(name, fn, unit_work=VOXELS):
            d, k = timed(lambda: fn(), share, 50)
            stages[name] = {"ms": d * 1e3, "threads": threads, "runs": k}
        if workload != "images":
            st("backbone", lambda: net.backbone(data, None))
        st("rpn_convs_heads", lambda: (net.rpn(l1, 1), net.rpn(l2, 2)))
        if o is not None:
            levels = []
            for lid, feat in ((1, l1), (2, l2)):
                anchors = torch.from_numpy(orc.generate_anchors(feat.shape[2:], net.stride, net.anchor_sizes[lid]))
                levels.append((lid, o["rpn_cls_prob_level%d" % lid], o["rpn_bbox_pred_level%d" % lid], anchors))
            tc = cfg["TEST"]
            st("proposal_layer_cpu_nms", lambda: orc.proposal_layer(levels, tuple(data.shape[2:]), tc["RPN_PRE_NMS_TOP_N"],
                                                                    tc["RPN_POST_NMS_TOP_N"], tc["RPN_NMS_THRESH"], cfg["ALLOW_BORDER"]))
            rois, lv = o["rois"][0], o["level_inds"][0]
            st("roi_pool_c", lambda: net.roi_pool_layer(l1, l2, rois, lv))
            stages["roi_pool_c"]["rois"] = int(rois.shape[0])
            if "pool5" in o:
                st("classifier", lambda: net.classify(o["pool5"]))
            if any(k.startswith("mask_backbone") for k in net.sd):
                crop = data[:, :, 8:38, 6:36, 10:46].contiguous()
                st("mask_head_30x30x36_crop", lambda: net.mask_backbone(crop))
        if feats is not None:
            st("projection_view_max", lambda: orc.project_views_max(feats, i3d, i2d, data.shape[2:]))
    torch.set_num_threads(1)
    per1, n1 = timed(one, seconds * 0.15, 3)
    torch.set_num_threads(threads)
    port = {"value": VOXELS / per, "unit": "voxels/s", "cores": threads, "kind": "port", "runs": n,
            "ms_per_chunk_median": per * 1e3, "ms_per_chunk_min_max": [lo * 1e3, hi * 1e3],
            "what": "the pinned oracle (oracle/sis3d_oracle.py: the reference's CPU operators via torch-CPU/oneDNN) on the GPU line's own "
                    "workload (%s)" % workload}
    head = ref if (ref and "value" in ref) else port
    out = dict(value=head["value"], unit="voxels/s", cores=threads, kind=head["kind"], host_cores=cores, cpu=cpu_model(),
               threads_rule=rule, runs=head["runs"], ms_per_chunk_median=head["ms_per_chunk_median"],
               ms_per_chunk_min_max=head["ms_per_chunk_min_max"],
               single_thread={"value": VOXELS / per1, "unit": "voxels/s", "cores": 1, "runs": n1, "kind": "port"},
               stages=stages, port=port,
               sample=("%d forward passes of one 96x48x96 chunk; value = MEDIAN of the runs at %d threads (%s); "
                       % (head["runs"], threads, rule))
               + ("kind reference: the reference's own Network.forward TEST branch run in place (config[0], full detection pass); "
                  "`port` = the oracle on the GPU line's workload (%s); " % workload if head is ref else
                  "kind port: the oracle on the GPU line's workload (%s) -- the reference tree was not available here; " % workload)
               + "stages: per-stage means of the port at the same thread count; single_thread: %d passes of the port" % n1)
    if ref is not None:
        out["reference"] = ref
    return out


def cpu_compute_projection(depth, c2w, w2g, cfg, dims, views, seconds):
    """the CPU restatement of ProjectionHelper.compute_projection (oracle/sis3d_oracle.py, pinned to lib/layer_utils/projection.py:52-121
    by tests/test_oracle_pinning.py) on the same depth maps: microseconds for `views` views, median over the runs that fit `seconds`"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sis3d_oracle as orc
    threads, why = cpu_threads_rule()
    torch.set_num_threads(threads)

    def one():
        for v in range(views):
            orc.compute_projection(depth[v], c2w[v], w2g[v], cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE, dims,
                                   cfg.VOXEL_SIZE)
    per, n, lo, hi = _median_runs(one, max(0.5, seconds), min_runs=3, max_runs=50)
    return {"us": per * 1e6, "runs": n, "cores": threads, "kind": "port", "min_us": lo * 1e6, "max_us": hi * 1e6,
            "sample": "%d views over one %dx%dx%d grid, median of %d runs" % ((views,) + tuple(dims) + (n,))}
