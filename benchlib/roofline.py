"""Roofline accounting of bench.py: the dominant kernel timed live, executed-vs-algorithmic FLOPs of the Winograd layers, the stage
times of one chunk alone, the PMC traffic of the dominant kernel (committed file and live counter passes)."""
import json
import os
import subprocess
import sys
import time

from .constants import *  # noqa: F401,F403

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def time_dominant_kernel(net, iters=50):
    """mean duration of the rpn_net k3 128->256 conv launch (12.23 algorithmic GFLOP; the default route is the fp32 Winograd kernel,
    ops.set_winograd(False) = the direct fp32 MFMA kernel), HIP events on the launch (current) stream.
    Runs before any graph is captured, on its own input, so the timed launches have the chip to themselves.  (The round-1/2 fault of
    "eager launches between graph replays" was a HIP-graph memset node, removed in round 3: DESIGN.md section 7.)"""
    import torch
    from sis3d import ops
    x = ops.new_act(128, (24, 12, 24), torch.device("cuda"))
    x.normal_().clamp_(min=0)                      # post-ReLU-like activations
    conv = net.rpn_net_level1
    for _ in range(100):                           # bring the clocks up: measured cold the same launch is ~10 % slower
        conv(x)
    torch.cuda.synchronize()
    # five batches of `iters` back-to-back launches, HIP events around each batch; the MEDIAN batch mean is reported (a single
    # batch swings 96-106 us with the box's clock state; the rocprofv3 trace of the same launches is in profiles/)
    means = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            conv(x)
        e1.record()
        torch.cuda.synchronize()
        means.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(means)[len(means) // 2]


def executed_flops(algorithmic, wino_algorithmic):
    """FLOPs the matrix pipe executes: layers on the Winograd kernel issue 64 products per 2x2x2 output block instead of 216"""
    return algorithmic - wino_algorithmic * (1.0 - 1.0 / WINOGRAD_REDUCTION)


def wino_accounting(net, shared=False):
    """ALGORITHMIC FLOPs of the launches that take the Winograd kernel, in the backbone proper and in backbone + RPN of one chunk:
    one eager pass of each with ops.flop_tally on (whatever the dispatch rule sends there today is what gets counted).  shared: count
    under the shared-chip dispatch (ops.dispatch_regime(shared_chip=True)), which is what pipelines of several chunks in flight capture --
    more layers take the Winograd kernel there, so fewer FLOPs are executed."""
    import torch
    from sis3d import ops, synthetic
    scene = synthetic.synth_chunk(0).cuda().float()
    out = {}
    with torch.no_grad(), ops.dispatch_regime(shared_chip=shared, brick_cap=(108 if shared else 0)):
        for name, fn in (("backbone", net.backbone_only), ("backbone_rpn", net.backbone_rpn)):
            ops.flop_tally(True)
            try:
                fn(scene)
            finally:
                out[name] = ops.flop_tally(False)["wino_algorithmic_flops"]
    torch.cuda.synchronize()
    return out


def time_stages(net, wino, reps=60):
    """ONE chunk alone on the GPU: captured graph of the backbone proper and of backbone+RPN, `reps` back-to-back replays
    on one stream bracketed by HIP events -> ms per chunk.  The difference is the RPN (convs + heads + softmax)."""
    import torch
    from sis3d import synthetic
    from sis3d.engine import ChunkEngine
    out = {}
    data = synthetic.synth_chunk(0)
    for stage in ("backbone", "rpn"):
        eng = ChunkEngine(net, stage=stage)
        eng.load(data)
        eng.prepare(warmup=2)
        for _ in range(10):
            eng.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.run()
        e1.record()
        torch.cuda.synchronize()
        out[stage] = e0.elapsed_time(e1) / reps
        del eng
    b, full = out["backbone"], out["rpn"]
    r = max(full - b, 1e-6)
    wb, wf = wino["backbone"], wino["backbone_rpn"]

    def frac(ms, algo, wino_flops):
        ex = executed_flops(algo["flops"], wino_flops)
        return {"ms": ms, "fp32_frac": ex / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                "executed_gflop": ex / 1e9, "algorithmic_tflops": algo["flops"] / (ms * 1e-3) / 1e12,
                "hbm_frac": algo["bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "voxels_per_s": VOXELS / (ms * 1e-3)}
    return {"backbone": dict(frac(b, BACKBONE, wb), algo_gflop=BACKBONE["flops"] / 1e9, algo_mb=BACKBONE["bytes"] / 1e6),
            "rpn": dict(frac(r, RPN, wf - wb), algo_gflop=RPN["flops"] / 1e9, algo_mb=RPN["bytes"] / 1e6),
            "backbone_rpn": frac(full, ALGO["backbone_rpn"], wf),
            "how": "one chunk alone, captured graph, %d back-to-back replays; fp32_frac = EXECUTED MFMA FLOPs (Winograd layers: algorithmic "
                   "/ 3.375) / time / 157.3 TF; hbm_frac = algorithmic bytes / time / 8 TB/s" % reps}


PMC_FILES = {True: ("r02_pmc_rpn_net.json", "r01_pmc_rpn_net.json"),
             False: ("r05_pmc_rpn_net_winograd.json", "r04_pmc_rpn_net_winograd.json", "r03_pmc_rpn_net_winograd.json")}
DOMINANT_BYTES = (6912 * 128 + 6912 * 256 + 256 * 128 * 27) * 4.0      # in + out + weights once: 14.16 MB per launch


def pmc_traffic(direct=False):
    """(HBM bytes per launch of the dominant kernel, file it comes from): the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, KB -> B).  Counters cannot be read from inside the bench: the figure belongs to the round and the
    kernel revision the file names, NOT to this run."""
    for name in PMC_FILES[bool(direct)]:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["traffic_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


def live_pmc_traffic(timeout_s=90):
    """HBM bytes per launch of the dominant kernel measured IN THIS RUN: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE -- each in its
    own pass, with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over 40 eager launches of the rpn_net layer (tools/wino_pmc.py), as
    child processes of rank 0 after the timed regions.  -> (bytes | None, dict describing the collection)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, {"error": "rocprofv3 not found"}
    vals, info = {}, {"tool": "rocprofv3 --kernel-trace --pmc <counter> -- python tools/wino_pmc.py rpn", "launches": 40}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sis3d_pmc_", dir="/tmp")
        try:
            subprocess.run([rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.join(ROOT, "tools", "wino_pmc.py"), "rpn"], cwd="/tmp", env=env, timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            v = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k3wino" in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                        v.append(float(r["Counter_Value"]))
            if not v:
                return None, dict(info, error="no %s rows for the Winograd kernel" % ctr)
            v.sort()
            vals[ctr] = v[len(v) // 2]
        except Exception as e:
            return None, dict(info, error="%s pass: %s: %s" % (ctr, type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    info.update({"FETCH_SIZE_KB_median": vals["FETCH_SIZE"], "WRITE_SIZE_KB_median": vals["WRITE_SIZE"],
                 "formula": "2 x FETCH_SIZE (gfx950: the counter takes 64 B per 128 B request) + WRITE_SIZE, KB -> B"})
    return int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024), info


def roofline_entry(kt, kt_direct, winograd):
    """Dominant kernel against the fp32 MFMA roof.  `frac` / `achieved` / `flops_per_launch` are the FLOPs the matrix pipe EXECUTES
    (what a roofline fraction means: <= 1 by construction).  The Winograd kernel issues 3.375x fewer multiplications than the
    direct-convolution count of SURVEY 8d (12.23 GFLOP); that count and the rate it gives are flat sibling keys
    (`algorithmic_*`), never a fraction."""
    tb, tsrc = pmc_traffic(direct=not winograd)
    red = WINOGRAD_REDUCTION if winograd else 1.0
    ex = DOMINANT_FLOPS / red
    e = {"bound": "mfma",
         "kernel": ("rpn_net k3 128->256 conv, Winograd F(2x2x2,3x3x3) in exact fp32 (binary32 adds + fp32 MFMA, csrc/conv3d_wino.hip)"
                    if winograd else "rpn_net k3 128->256 conv, direct implicit GEMM (exact fp32 MFMA, csrc/conv3d_t16.hip)"),
         "achieved": ex / kt / 1e12, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ex / kt / 1e12 / FP32_PEAK_TF,
         "launch_us": kt * 1e6, "flops_per_launch": ex,
         "flops_are": "executed MFMA FLOPs (v_mfma_f32_16x16x4_f32 count x 512)",
         "algorithmic_gflop_per_launch": DOMINANT_FLOPS / 1e9, "algorithmic_tflops": DOMINANT_FLOPS / kt / 1e12,
         "algorithmic_speedup_vs_direct_count": red,
         "traffic": tb, "traffic_source": tsrc, "algorithmic_bytes_per_launch": DOMINANT_BYTES,
         "traffic_ratio": (tb / DOMINANT_BYTES) if tb else None}
    if winograd and kt_direct > 0:
        db, dsrc = pmc_traffic(direct=True)
        e["direct_kernel"] = {"launch_us": kt_direct * 1e6, "achieved": DOMINANT_FLOPS / kt_direct / 1e12,
                              "frac": DOMINANT_FLOPS / kt_direct / 1e12 / FP32_PEAK_TF, "traffic": db, "traffic_source": dsrc,
                              "what": "the same layer on the direct fp32 MFMA kernel (ops.set_winograd(False)), same run"}
    return e


def fracs_above_one(obj, path=""):
    """every key whose name contains 'frac' must be a fraction of a roof: -> list of (path, value) above 1 (tests assert it is empty)"""
    bad = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            q = path + "." + k if path else k
            if "frac" in k and isinstance(v, (int, float)) and v > 1.0:
                bad.append((q, v))
            bad += fracs_above_one(v, q)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            bad += fracs_above_one(v, "%s[%d]" % (path, i))
    return bad
