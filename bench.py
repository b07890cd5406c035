#!/usr/bin/env python
"""bench.py -- voxels/sec of the 3D-SIS forward hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload auto|backbone_rpn|detect|images|scene] [--masks]

One process per GPU.  Under `python -m torch.distributed.run` (how the driver starts N > 1) RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the env; started directly with `--gpus N`, N > 1, this script launches the N ranks
itself (re-exec under torch.distributed.run, rendezvous on 127.0.0.1) and FAILS if the box has fewer than N GPUs --
it never silently reports a smaller world.

Workloads.  `--workload auto` (the default, what the driver runs at N = 1, 2, 4, 8) makes `value` the SAME workload at every N --
backbone_rpn, BASELINE config[1], weak scaling -- and every line, at every N, carries BOTH workloads under the same two keys:
`chunk_pipeline` (config[1], weak: == `value` under auto) and `scene` (config[4], strong: the 32-chunk scene with its RCCL
all-gather and whole-scene NMS, `records_gathered`, `kept_after_scene_nms`, `ms_per_scene`; at N > 1 also rank 0 alone on the
same scene = `single_gpu` and `speedup_vs_1gpu`).  So value(N) / value(1) compares like with like, and the strong-scaling figure
of the collective path can be read off the `scene` key of the same lines.  `--workload scene` makes the scene the headline instead.
  backbone_rpn  BASELINE config[1]: one 96x48x96 geometry-only chunk per pipeline, HIP backbone + RPN; a step = one pass
                over `--inflight` (default 4 on the 8 hardware queues this script asks HIP for, profiles/r04_hw_queues.txt)
                independent chunks per GPU, each on its own HIP stream / captured graph, inputs resident in HBM.  Ranks share
                nothing (scaling: weak).
  detect        config[2]: + decode / top-k / NMS / RoI pooling / classifier (+ `--masks`: mask head on a fixed
                deterministic detection set).
  images        config[3]: 5-view back-projection + colour/geometry backbone + RPN.
  scene         config[4]: a 32-chunk scene (4 x 1 x 8 grid of 96x48x96 chunks whose origins sit `--scene-stride` = 80 voxels
                apart, i.e. neighbours OVERLAP by 16 voxels so that the whole-scene NMS really suppresses duplicates across chunk
                borders: kept_after_scene_nms < records_gathered), chunk c -> rank c mod N, per-chunk detection, ONE RCCL
                all_gather_into_tensor of the record blocks, whole-scene 3D NMS on every rank.  A step = one scene (strong).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- dominant kernel (the k3 128->256 RPN conv, 12.23 GFLOP/launch): achieved = algorithmic FLOPs / mean
                  launch duration measured live with HIP events on the launch stream; peak = 157.3 TF (fp32 MFMA).
  stages       -- N = 1: the backbone proper and backbone+RPN of ONE chunk alone on the GPU (captured graph, back-to-back
                  replays on one stream, HIP events): ms, fraction of the fp32 roof and of the 8 TB/s HBM roof.
  cpu_baseline -- the CPU oracle (torch-CPU operators = what the reference's MAX_VOLUME=0 path runs) timed on this box's
                  host cores on a bounded sample, per stage, at the best thread count and at 1 thread; rank 0 at N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
# HIP gives a process 4 hardware queues by default and maps its streams onto them round-robin: the pipelines' streams, the capture /
# side streams and the null stream then share queues, and a FOURTH chunk in flight serialises behind another one (round 4,
# profiles/r04_hw_queues.txt, same box: 4 in flight 1.80 G voxels/s on 4 queues, 2.28 G on >= 6; 3 in flight 2.16 G on either).  Read by the HIP
# runtime when it initialises, i.e. before torch is imported anywhere below; an integrator sets the same variable in his launcher.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

VOXELS = 96 * 48 * 96
# algorithmic work per chunk
BACKBONE = dict(bytes=201.6e6, flops=17.72e9)
RPN = dict(bytes=59.8e6, flops=24.86e9)
ALGO = {
    "backbone_rpn": dict(bytes=261.4e6, flops=42.58e9),
    "detect": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
    "images": dict(bytes=517e6 + 28.3e6 + 59.8e6 + 226.5e6, flops=29.1e9 + 24.86e9),
    "scene": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
}
DOMINANT_FLOPS = 2.0 * 6912 * 256 * 128 * 27        # rpn_net_level{1,2}: 12.23 GFLOP per launch (ALGORITHMIC = direct-convolution count)
WINOGRAD_REDUCTION = 27 * 8 / 64.0                  # F(2x2x2, 3x3x3): 64 products per 2x2x2 output block instead of 216
FP32_PEAK_TF = 157.3
HBM_PEAK_GBS = 8000.0
WORKLOAD_TEXT = {
    "backbone_rpn": "config[1]: one 96x48x96 chunk per pipeline, geometry-only, HIP 3D-conv backbone + RPN (convs, heads, "
                    "softmax), weights seeded synthetic",
    "detect": "config[2]: backbone + RPN + decode/sort/NMS + RoI pooling + classifier",
    "images": "config[3]: 5-view back-projection gather + colour/geometry backbone + RPN",
    "scene": "config[4]: 32-chunk scene sharded chunk->rank, per-chunk detection, one RCCL all-gather of record blocks, "
             "whole-scene 3D NMS on every rank",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=250.0,
                    help="untimed passes of the workload in front of the W warm-up steps, in ms of wall clock (clock ramp-up; 0 = off)")
    ap.add_argument("--workload", default="auto", choices=["auto", "backbone_rpn", "detect", "images", "scene"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--inflight", type=int, default=0, help="independent chunks in flight per GPU (HIP streams); 0 = the "
                    "measured best: 4 with GPU_MAX_HW_QUEUES >= 6 (this script sets 8 unless the variable is already set), else 3")
    ap.add_argument("--masks", action="store_true", help="detect: also run the mask head (config[2] in full) on a fixed "
                    "deterministic detection set; scene: mask the detections that survive the whole-scene NMS, each on the "
                    "chunk / rank that produced it")
    ap.add_argument("--mask-boxes", type=int, default=16, help="detect --masks: number of post-NMS RoIs taken as detections")
    ap.add_argument("--scene-chunks", type=int, default=32)
    ap.add_argument("--scene-stride", type=float, default=80.0, help="origin spacing of the scene's 4 x 1 x n/4 chunk grid in voxels "
                    "(96 = edge to edge: nothing to suppress; 80 = 16-voxel overlap: the whole-scene NMS removes cross-chunk duplicates)")
    ap.add_argument("--scene-steps", type=int, default=0, help="scenes timed for the `scene` side key (0 = min(steps, 20))")
    ap.add_argument("--group", type=int, default=1, help="chunks per captured graph (2: the pair's four RPN convs in one launch)")
    ap.add_argument("--from-depth", action="store_true", help="images workload: views arrive as depth maps + poses and the "
                    "voxel->pixel lists are computed on the device inside the timed step (sis3d_compute_projection)")
    ap.add_argument("--rgb", action="store_true", help="images workload: the views are RGB images and the ENet 2D encoder "
                    "(csrc/enet.hip, own captured graph) runs inside the timed step (BASELINE config[3] from pixels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 counter passes (FETCH_SIZE / WRITE_SIZE) over the "
                    "dominant kernel at the end of an N = 1 run; `roofline.traffic` then is the committed figure of profiles/")
    ap.add_argument("--split-line", action="store_true", help="also time the opt-in split-bf16 variant of the k3 convs (separately "
                    "reported, never the headline; off by default since r5: it does not beat the exact-fp32 path in throughput)")
    ap.add_argument("--no-calibrate", action="store_true", help="keep the pipelines on the first streams of the pool instead of choosing "
                    "the window of streams by measurement (PipelinedEngines.calibrate / SceneRunner.calibrate)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed-input variants (fresh chunks from pinned host memory)")
    ap.add_argument("--no-side-configs", action="store_true", help="N = 1 default run: skip the detect / detect_masks / images / "
                    "images_rgb sub-objects (BASELINE configs[2], [3])")
    ap.add_argument("--no-side-workloads", action="store_true", help="time only the headline workload (no chunk_pipeline / scene side keys)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--selftest-cpu", action="store_true", help=argparse.SUPPRESS)   # launch / rendezvous logic under gloo, no GPU
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launching N ranks --
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n, argv, port=None):
    """the command that starts n ranks of this script on this node (one per GPU, rendezvous on the loopback address)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: check the box, then start the N ranks ourselves"""
    if not args.selftest_cpu:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if os.environ.get("SIS3D_BENCH_SHARE_GPU") and have >= 1:
            have = args.gpus        # functional test hook: every rank on GPU 0, gloo instead of RCCL (see main())
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but this box exposes %d GPU(s); refusing to run a smaller world\n"
                             % (args.gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16) // args.gpus))))
    return subprocess.call(launch_command(args.gpus, argv), env=env)


def emit(line):
    """the JSON line must be the LAST thing on stdout: RCCL printf()s a version banner into C stdio's buffer, which would
    otherwise be flushed at exit, after our line"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


def scene_origin(c, stride):
    """origin of chunk c of the scene's 4 x 1 x n/4 chunk grid, in scene voxels"""
    return (float(stride) * (c % 4), 0.0, float(stride) * (c // 4))


def hw_queues():
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


def default_inflight(workload):
    """chunks in flight per GPU that measured best (profiles/r04_hw_queues.txt, 200-step runs after a clock pre-heat, same box): FOUR on
    >= 6 hardware queues for every workload -- backbone + RPN 2.28 G voxels/s (3: 2.16, 5: 2.00, 6: 1.94), detect 2.02 (3: 1.86),
    images 1.60 (3: 1.50) -- and three on HIP's default of 4 queues, where a fourth stream shares a queue with another one (1.80 G)."""
    return 4 if hw_queues() >= 6 else 3


def chunk_pipeline_entry(value, unit, ms_per_step, chunks_per_step_per_gpu, single_ms):
    return {"workload": WORKLOAD_TEXT["backbone_rpn"], "value": value, "unit": unit, "scaling": "weak", "ms_per_step": ms_per_step,
            "chunks_per_step_per_gpu": chunks_per_step_per_gpu, "single_chunk_latency_ms": single_ms}


def scene_entry(value, unit, ms_per_scene, steps, extra):
    return {"workload": WORKLOAD_TEXT["scene"], "value": value, "unit": unit, "scaling": "strong", "ms_per_scene": ms_per_scene,
            "steps": steps, **extra}


def selftest_cpu(args, rank, world):
    """The launch / rendezvous / sharding / max-over-ranks / line-assembly logic of the bench on CPU under gloo: both workloads with
    synthetic record blocks instead of GPU detections (chunk_pipeline = packing `inflight` blocks per rank, no collective; scene =
    the record all-gather + merge).  Used by tests/test_bench_launch.py; the line has the same two side keys as a GPU run."""
    import torch
    import torch.distributed as dist
    from sis3d import parallel
    if world > 1:
        dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    k_rows, n_chunks, nfl = 8, args.scene_chunks, 3

    def block(c):
        g = torch.Generator().manual_seed(c)
        rec = torch.rand(k_rows, parallel.RECORD_WIDTH, generator=g)
        return parallel.pack_block(rec, 3 + c % 5, scene_origin(c, args.scene_stride))
    # chunk_pipeline: rank-local work only
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for i in range(nfl):
            block(rank * nfl + i)
    barrier()
    cdt = max_over_ranks(time.perf_counter() - t0)
    # scene: shard, gather, merge
    local = [block(c) for c in parallel.shard_chunks(n_chunks, rank, world)]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        blocks = parallel.gather_blocks(local, n_chunks, k_rows)
        recs, keep = parallel.merge_scene(blocks, k_rows, lambda b, th: torch.arange(b.shape[0]), 0.1)
    barrier()
    sdt = max_over_ranks(time.perf_counter() - t0)
    barrier()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        cp = chunk_pipeline_entry(world * nfl * args.steps / cdt, "chunks/s", cdt / args.steps * 1e3, nfl, None)
        sc = scene_entry(n_chunks * args.steps / sdt, "chunks/s", sdt / args.steps * 1e3, args.steps,
                         {"scene_chunks": n_chunks, "scene_stride": args.scene_stride, "records_gathered": int(recs.shape[0]),
                          "kept_after_scene_nms": int(keep.numel())})
        head = sc if args.workload == "scene" else cp
        emit({"metric": "selftest (gloo, CPU): synthetic record blocks, no GPU work", "value": head["value"],
              "unit": "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "selftest": True,
              "scaling": head["scaling"], "config": {"workload": head["workload"], "scene_chunks": n_chunks, "records": int(recs.shape[0])},
              "chunk_pipeline": cp, "scene": sc})
    return 0


# ---------------------------------------------------------------------------------------------------- GPU workloads --
def build_net(workload, masks=False, rgb=False):
    from sis3d import config, synthetic
    from sis3d.nets import backbones
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = workload == "images"
    cfg.USE_IMAGES_GT = not (rgb and workload == "images")
    cfg.USE_MASK = bool(masks)
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_checkpoint(shapes, seed=0)
    net.load_state_dict(sd)
    return net.cuda().eval(), cfg, sd


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
            "how": "one chunk alone: captured graph, %d back-to-back replays on one stream, HIP events; rpn = backbone_rpn - backbone; "
                   "fp32_frac = EXECUTED MFMA FLOPs / time / 157.3 TF (layers on the Winograd kernel issue their algorithmic count / "
                   "3.375: executed_gflop; counted per launch by ops.flop_tally in one eager pass), algorithmic_tflops = the "
                   "direct-convolution count / time (not a fraction of the roof); hbm_frac is against 8 TB/s with the algorithmic "
                   "bytes; the binding roof is fp32 MFMA" % reps}


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

    def one():
        with torch.no_grad():
            if workload in ("detect", "scene") or not cfg["USE_IMAGES_GT"]:
                net.forward(data, feats, i3d, i2d)
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


def ops_mod():
    from sis3d import ops
    return ops


def preheat(step, ms):
    """untimed passes of the workload itself for `ms` of wall clock, in front of the W warm-up steps: a 20-step timed region is ~20 ms,
    and measured from a cold start (clocks at idle: the first thing a fresh process does) the same code reads 1.39-1.77 instead of
    1.95 G voxels/s (round 4, profiles/r04_inflight.txt) -- whatever ran before the timed region decided the number.  Synchronises
    every 16 passes so the launch queue stays shallow."""
    import torch
    if ms <= 0:
        return
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(16):
            step()
        torch.cuda.synchronize()


def time_streamed(net, stage, args, rank, nfl, barrier, mode):
    """The chunk pipelines on FRESH chunks from pinned host memory (the reference's forward owns the upload:
    `blobs['data'].cuda()`, lib/nets/network.py:191): every step every pipeline runs a chunk it has not seen in the previous RING - 1
    steps.  The pipelines are captured with a MAILBOX (PipelinedEngines(mailbox=True), ops.Mailbox): the first node of a pipeline's
    graph reads the chunk's host pointer from a ring of slots in pinned memory and pulls the chunk across PCIe itself
    (sis3d_mail_upload), so the host's only call per chunk is the graph launch.  mode 'grid': the encoded (1,2,96,48,96) float32 grid
    the reference's dataloader hands over (3.54 MB per chunk); 'sdf': the raw SDF block of a .chunk file (1.77 MB), TSDF-encoded by the
    graph's second node.  -> dict(dt, bytes_per_chunk, ...)"""
    import torch
    from sis3d import synthetic
    from sis3d.engine import PipelinedEngines
    RING = 4
    torch.cuda.synchronize()
    eng = PipelinedEngines(net, nfl, stage=stage, mailbox=True, mail_input=mode)
    ring = []
    for i in range(nfl):
        row = []
        for r in range(RING):
            cid = 1000 + (rank * nfl + i) * RING + r
            t = synthetic.synth_chunk(cid) if mode == "grid" else synthetic.synth_sdf(cid)
            row.append(t.contiguous().pin_memory())
        ring.append(row)
    eng.prepare(warmup=2)
    eng.enable_feed(mode)
    k = [0]

    # the loader runs ONE chunk ahead of the device (the reference's DataLoader prefetches whole batches): a pipeline knows its next
    # chunk when it starts a pass, so that chunk crosses the link during the pass (piggyback row of the rpn_net conv launch)
    for i in range(nfl):
        eng.feed(i, ring[i][0])

    def step():
        for i in range(nfl):
            eng.feed(i, ring[i][(k[0] + 1) % RING])
            eng.run_fed(i)
        k[0] += 1
    if not args.no_calibrate and nfl >= 2:
        preheat(step, min(args.preheat_ms, 60.0))
        eng.calibrate(step, reps=8, warm=2)
    preheat(step, args.preheat_ms)
    for _ in range(max(args.warmup, 4)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    return dict(dt=dt, bytes_per_chunk=ring[0][0].numel() * 4, ring=RING, host_ms_per_step=t_host / args.steps * 1e3, copy="mailbox",
                stream_window=eng.stream_window, stage_ahead=all(e.piggybacked for e in eng.engines))


def streamed_entry(st, resident_dt, steps, vox_per_step, nfl, world, mode):
    ms = st["dt"] / steps * 1e3
    chunks_per_s = vox_per_step / VOXELS * steps / st["dt"]
    return {"value": vox_per_step * steps / st["dt"], "unit": "voxels/s", "ms_per_step": ms,
            "ratio_to_resident": resident_dt / st["dt"], "host_bytes_per_chunk": st["bytes_per_chunk"],
            "h2d_gbs_per_gpu": chunks_per_s / world * st["bytes_per_chunk"] / 1e9,
            "host_enqueue_ms_per_step": st.get("host_ms_per_step"), "upload_by": st.get("copy"),
            "input": ("encoded (1,2,96,48,96) float32 grid = the reference's blobs['data'] (lib/nets/network.py:191)" if mode == "grid"
                      else "raw float32 SDF block in .chunk file order, TSDF-encoded on the device (sis3d_tsdf_encode; dataset.py:54-70)"),
            "how": "every step each of the %d pipelines runs a FRESH chunk from pinned host memory.  The pipelines are captured with a "
                   "mailbox: the first node of a pipeline's graph reads the chunk's host pointer from a ring of slots in pinned memory "
                   "(CPU stores by the host, no HIP call) and pulls the chunk across PCIe itself (sis3d_mail_upload: 8 workgroups, 256 KB "
                   "in flight; sdf: + sis3d_tsdf_encode as the second node), so the host's ONLY call per chunk is the graph launch -- a "
                   "command enqueued behind a graph launch that has not finished can block the host on this runtime.  The loader runs one "
                   "chunk ahead: a pass's slot also names the pipeline's NEXT chunk, which one more row of workgroups of the pass's rpn_net "
                   "conv launch pulls across the link into a staging buffer while the pass computes (sis3d_conv3d_k3wino_piggyback: "
                   "`stage_ahead`), so the next pass starts with a device copy instead of waiting on PCIe; ring of %d distinct host "
                   "chunks per pipeline; the timed region contains every upload" % (nfl, st["ring"]), "stage_ahead": st.get("stage_ahead")}


def run_chunk_pipeline(net, cfg, args, rank, world, workload, barrier, masks=False):
    """weak-scaling workloads: `inflight` independent chunks per GPU per step -> dict(dt, vox_per_step, single_ms, extra)"""
    import torch
    from sis3d import synthetic
    from sis3d.engine import PipelinedEngines
    stage = "rpn" if workload in ("backbone_rpn", "images") else "detect"
    nfl = max(1, args.inflight)
    from_depth = workload == "images" and args.from_depth
    grp = max(1, args.group)
    kw = {"from_depth": True} if from_depth else {}
    if masks:
        kw["mask_boxes"] = args.mask_boxes
    eng = PipelinedEngines(net, nfl, stage=stage, use_graph=not args.no_graph, group=grp, **kw)
    for i in range(nfl):
        for g in range(grp):
            cid = (rank * nfl + i) * grp + g
            data = synthetic.synth_chunk(cid)
            if from_depth:
                feats = synthetic.synth_views(cid, n_per_view=0)[0]
                depth, c2w, w2g = synthetic.synth_cameras(cid, feats.shape[0], voxel_size=cfg.VOXEL_SIZE)
                with torch.cuda.stream(eng.streams[i]):
                    eng.engines[i].load_views(data, feats, depth, c2w, w2g, slot=g)
            elif workload == "images" and args.rgb:
                _, i3d, i2d = synthetic.synth_views(cid)
                with torch.cuda.stream(eng.streams[i]):
                    eng.engines[i].load_rgb(data, synthetic.synth_images(cid, cfg.NUM_IMAGES), i3d, i2d, slot=g)
            elif workload == "images":
                feats, i3d, i2d = synthetic.synth_views(cid)
                eng.load(i, data, feats, i3d, i2d, slot=g)
            else:
                eng.load(i, data, slot=g)
    eng.prepare(warmup=2)
    if not args.no_calibrate and nfl >= 2 and not args.no_graph:
        preheat(eng.run, min(args.preheat_ms, 60.0))
        eng.calibrate(eng.run, reps=8, warm=2)
    preheat(eng.run, args.preheat_ms)
    for _ in range(args.warmup):
        eng.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run()
    barrier()
    dt = time.perf_counter() - t0
    extra = {"chunks_per_graph": grp, "streams_per_gpu": nfl}
    if eng.stream_window_times:
        extra["stream_window"] = eng.stream_window
        extra["stream_window_ms_per_step"] = {str(k): round(v, 4) for k, v in eng.stream_window_times.items()}
    streamed = {}
    if workload in ("backbone_rpn", "detect") and not masks and grp == 1 and not args.no_graph and not args.no_streamed:
        for mode in ("grid", "sdf"):
            try:
                streamed[mode] = time_streamed(net, stage, args, rank, nfl, barrier, mode)
            except Exception as e:                               # a side measurement must never take the headline down
                streamed[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.synchronize()
    # Winograd accounting of THIS configuration under the regime its pipelines captured: one eager pass of pipeline 0
    wino_flops = None
    if rank == 0 and grp == 1:
        ops_mod().flop_tally(True)
        try:
            with torch.no_grad(), torch.cuda.stream(eng.streams[0]):
                eng.engines[0]._step()
        finally:
            wino_flops = ops_mod().flop_tally(False)["wino_algorithmic_flops"]
        torch.cuda.synchronize()
    if workload == "images" and args.rgb:
        extra["views_from"] = ("RGB images (5 x 3 x 256 x 328) through the ENet encoder inside the step (5.2 GFLOP; csrc/enet.hip, one launch "
                               "per bottleneck: 25 launches)")
        extra["enet_impl"] = getattr(net, "enet_impl", "hip")
        e0 = eng.engines[0]
        extra["enet_graph_captured"] = e0.enet_graph is not None
        # the encoder alone on one pipeline: 30 passes (graph replays when captured) between HIP events on that pipeline's stream
        with torch.cuda.stream(eng.streams[0]), torch.no_grad():
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(30):
                if e0.enet_graph is not None:
                    e0.enet_graph.replay()
                else:
                    e0._encode_views()
            ev1.record()
            ev1.synchronize()
        extra["enet_ms_5_views"] = ev0.elapsed_time(ev1) / 30
    if from_depth:
        torch.cuda.synchronize()
        extra.update({"views_from": "depth maps + poses (lists computed on device inside the step)",
                      "visible_voxels_per_view": eng.engines[0].view_counts()})
    if masks:
        torch.cuda.synchronize()
        extra.update(eng.engines[0].mask_stats())
    single_ms = None
    if rank == 0:
        # latency of ONE chunk on an otherwise idle GPU (single stream, serialised on purpose).  Its own engine, captured the way a
        # caller with one chunk would (no shared-chip hint): the pipelines above took the dispatch for several chunks in flight --
        # fewer, fatter Winograd work items -- which is slower when a chunk has the chip alone
        torch.cuda.synchronize()
        if nfl >= 2 and grp == 1 and not args.no_graph and not (workload == "images" and from_depth):
            from sis3d.engine import ChunkEngine
            solo = ChunkEngine(net, stage=stage, **kw)
            src = eng.engines[0]
            solo.scenes[0].copy_(src.scenes[0])
            if solo.use_images:
                solo.feats_[0].copy_(src.feats_[0]); solo.i3d_[0].copy_(src.i3d_[0]); solo.i2d_[0].copy_(src.i2d_[0])
            if getattr(solo, "rgb", False):
                solo.images_[0].copy_(src.images_[0])          # its own 5-view encoder pass in front of its 3D graph
            solo.prepare(warmup=2)
            one = solo.run
        else:
            solo, one = None, (lambda: eng.run(0))
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            one()
            torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 50 * 1e3
        extra["single_chunk_latency_is"] = ("one chunk with the chip to itself, on an engine captured for that case (sis3d.engine.ChunkEngine)"
                                            if solo is not None else "pipeline 0's graph replayed alone")
        del solo
        if masks and eng.engines[0].mask_plan is not None:
            # the mask head alone on the same fixed detection set (one chunk, nothing else on the GPU): captured graph of the six
            # ragged launches, HIP events around 50 replays -> mask_head_ms / mask_head_tf (algorithmic FLOPs / time)
            e0 = eng.engines[0]
            with torch.no_grad():
                from sis3d.engine import pooled_stream
                side = pooled_stream("capture", 0)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    net.mask_backbone.forward_planned(e0.scenes[0], e0.mask_plan)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    net.mask_backbone.forward_planned(e0.scenes[0], e0.mask_plan)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(50):
                    g.replay()
                ev1.record()
                torch.cuda.synchronize()
            mh_ms = ev0.elapsed_time(ev1) / 50
            ops_mod().flop_tally(True)
            try:
                with torch.no_grad():
                    net.mask_backbone.forward_planned(e0.scenes[0], e0.mask_plan)
            finally:
                mw = ops_mod().flop_tally(False)["wino_algorithmic_flops"]
            torch.cuda.synchronize()
            extra["mask_head_ms"] = mh_ms
            extra["mask_head_algorithmic_tflops"] = e0.mask_plan.flops / (mh_ms * 1e-3) / 1e12
            extra["mask_head_executed_gflop"] = executed_flops(e0.mask_plan.flops, mw) / 1e9
            extra["mask_head_fp32_frac"] = executed_flops(e0.mask_plan.flops, mw) / (mh_ms * 1e-3) / 1e12 / FP32_PEAK_TF
            mp = e0.mask_plan
            extra["mask_head_kernel"] = (
                ("Winograd ragged launch on 4x4x4 minis for the four 64->64 k3 layers (%d work items; %d on 8x4x8 blocks)" % (mp.items_mini, mp.blocks_wino)
                 if getattr(mp, "wino_mini", False) else "Winograd ragged launch for the four 64->64 k3 layers (%d work items)" % mp.blocks_wino)
                if (ops_mod().WINOGRAD and mp.wino) else "direct balanced kernel, ragged")
    snap = None
    if rank == 0 and grp == 1:
        torch.cuda.synchronize()
        o0 = eng.engines[0].out
        snap = {k: o0[k].detach().clone() for k in o0 if k.startswith("rpn_") and torch.is_tensor(o0[k])} if isinstance(o0, dict) else None
    return dict(dt=dt, vox_per_step=world * nfl * grp * VOXELS, single_ms=single_ms, extra=extra, snap=snap, streamed=streamed,
                wino_flops=wino_flops, nfl=nfl)


def run_scene(net, args, rank, world, n_chunks, barrier, group=None, steps=None, inflight=None, emulate=None, gathered=None,
              want_table=False, streamed=False, runner=None):
    """BASELINE config 5: n_chunks chunks of one scene (4 x 1 x n/4 grid of 96x48x96 chunks, origins `--scene-stride` apart),
    chunk c -> rank c mod W,
    per-chunk detection, ONE all-gather of the record blocks, whole-scene NMS on every rank.  A step = one whole scene.
    The host reads a scene's two result lengths (8 bytes) one step LATE -- after the next scene has been enqueued -- like a consumer
    that double-buffers its results; every scene's lengths are read, the last one inside the timed region.
    emulate = (r, W) + gathered = a full scene's gathered table: rank r's share of a W-rank run on this GPU alone -- its own chunks
    (one graph launch when it owns one chunk per pipeline), its rows written over the table, the merge of the FULL table.
    streamed: the rank's chunks sit in PINNED HOST memory and every scene uploads all of them (the first node of each pipeline's graph
    pulls its chunk across PCIe: ChunkEngine.submit / ops.Mailbox).
    runner: reuse a SceneRunner (its captured graphs) from a previous call with the same sharding."""
    import torch
    from sis3d import parallel, synthetic
    from sis3d.scene import SceneRunner
    gw, gr = (1, 0) if group == "solo" else (world, rank)
    if emulate is not None:
        gr, gw = emulate
    n_local = len(range(gr, n_chunks, gw))
    nfl = inflight or (args.inflight if args.inflight > 0 else (n_local if n_local <= 4 else default_inflight("scene")))
    if runner is None:
        runner = SceneRunner(net, synthetic.CHUNK_DIMS, use_graph=not args.no_graph, inflight=max(1, nfl), solo=(group == "solo"), emulate=emulate)
    chunks = []
    for c in range(n_chunks):
        payload = None
        if c % gw == gr:                                                         # own shard only
            payload = synthetic.synth_chunk(c).contiguous().pin_memory() if streamed else synthetic.synth_chunk(c).cuda()
        chunks.append((c, scene_origin(c, args.scene_stride), payload))
    torch.cuda.synchronize()
    steps = steps or args.steps
    # results are read one scene LATE (SceneRunner.infer(lazy=True): join, gather and merge of scene k on the merge stream, scene k + 1's
    # chunks already enqueued).  r4 measured this slower for the 32-chunk scene (10.2 vs 8.65 ms: the host ran a scene ahead and hit the
    # runtime's blocking enqueue behind unfinished graph launches); with the mailbox engines of r5 a chunk is one graph launch and the
    # pipelined form wins there too (7.19 vs 7.48 ms, SIS3D_BENCH_SCENE_LAZY=0 restores the eager reads for shares of > 1 chunk per pipeline)
    lazy = not args.masks and not args.no_graph and (n_local == nfl or os.environ.get("SIS3D_BENCH_SCENE_LAZY", "1") == "1")
    if lazy and n_local == nfl:
        runner.prepare_round()              # the one-launch round graph is captured here, not inside the first timed / pipelined call
    if not args.no_calibrate and not args.no_graph and not args.masks and runner.calibration is None and not streamed:
        runner.infer(chunks, gathered=gathered)
        runner.calibrate(chunks, gathered=gathered, lazy=lazy)

    def one():
        return runner.infer(chunks, with_masks=args.masks, gathered=gathered, lazy=lazy)

    def done(r):
        return r.resolve() if lazy else r
    for _ in range(max(3, min(args.warmup, 10))):              # the first replays of freshly captured graphs cost milliseconds each
        res = done(one())
    preheat(lambda: done(one()), args.preheat_ms / 16.0)       # a scene is 4-32 chunks: ~1/16 of the passes
    barrier()
    t0 = time.perf_counter()
    prev = None
    for _ in range(steps):
        cur = one()
        if prev is not None:
            done(prev)
        prev = cur
    res = done(prev)
    barrier()
    dt = time.perf_counter() - t0
    recs, keep = res[0], res[1]
    extra = {"scene_chunks": n_chunks, "scene_stride": args.scene_stride, "chunks_on_this_rank": n_local, "streams_per_gpu": nfl,
             "records_gathered": int(recs.shape[0]), "kept_after_scene_nms": int(keep.numel()),
             "one_graph_launch_per_scene": runner._round is not None and runner._use_round and n_local == nfl,
             "results_read_one_scene_late": bool(lazy)}
    if runner.calibration:
        extra["calibration"] = runner.calibration
    if args.masks:
        extra["masks_on_this_rank"] = len(res[2])
        extra["mask_voxels_on_this_rank"] = int(sum(m.numel() for _, m in res[2].values()))
    out = dict(dt=dt, vox_per_step=n_chunks * VOXELS, single_ms=None, extra=extra, steps=steps, runner=runner)
    if want_table:
        with torch.no_grad():
            out["table"] = parallel.gather_blocks(runner.run_chunks(chunks), n_chunks, runner.k_rows, solo=True).clone()
        torch.cuda.synchronize()
    return out


def collective_latency_world_of_one(block_floats_per_rank, iters=50):
    """The one collective of a scene -- `all_gather_into_tensor` of a rank's record blocks -- timed on THIS GPU in an RCCL world of
    ONE (a one-GPU box has no peer: this is the call's launch + kernel latency, not an xGMI transfer; the payload at N = 8 is 8 x
    51 KB, far below any link limit, so the latency term is what a rank pays).  HIP events around `iters` back-to-back calls.
    -> (microseconds per call | None, how)"""
    import torch
    import torch.distributed as dist
    made = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(free_port())
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
            made = True
        send = torch.zeros(int(block_floats_per_rank), device="cuda")
        recv = torch.empty(dist.get_world_size() * int(block_floats_per_rank), device="cuda")
        for _ in range(5):
            dist.all_gather_into_tensor(recv, send)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_gather_into_tensor(recv, send)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        return us, "RCCL all_gather_into_tensor of %d floats in a world of %d on this GPU, %d back-to-back calls, HIP events" % (
            int(block_floats_per_rank), dist.get_world_size(), iters)
    except Exception as e:
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        if made:
            try:
                dist.destroy_process_group()
            except Exception:
                pass


ENET_FLOPS = 5.2e9                      # 5 views of 256 x 328 through the encoder


def config_entry(res, steps, workload, masks=False, rgb=False, what=None):
    """one BASELINE config as a sub-object of the line: throughput with `chunks_per_step_per_gpu` chunks in flight, latency of one
    chunk alone, and the two roofline fractions of the STEP (executed MFMA FLOPs / time / 157.3 TF; algorithmic bytes / time / 8 TB/s)"""
    ms = res["dt"] / steps * 1e3
    nchunk = res["vox_per_step"] / VOXELS
    algo = dict(ALGO[workload])
    if masks and "mask_head_gflop" in res["extra"]:
        algo["flops"] += res["extra"]["mask_head_gflop"] * 1e9
    if rgb:
        algo["flops"] += ENET_FLOPS
    e = {"workload": what or WORKLOAD_TEXT[workload], "value": res["vox_per_step"] * steps / res["dt"], "unit": "voxels/s",
         "ms_per_step": ms, "steps": steps, "chunks_per_step_per_gpu": nchunk, "single_chunk_latency_ms": res["single_ms"],
         "hbm_frac": algo["bytes"] * nchunk / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         "algorithmic_gflop_per_chunk": algo["flops"] / 1e9}
    if res.get("wino_flops") is not None:
        ex = executed_flops(algo["flops"], res["wino_flops"])
        e["fp32_frac"] = ex * nchunk / (ms * 1e-3) / 1e12 / FP32_PEAK_TF
        e["executed_gflop_per_chunk"] = ex / 1e9
    for k in ("mask_boxes", "mask_voxels", "mask_head_gflop", "mask_head_ms", "mask_head_fp32_frac", "mask_head_algorithmic_tflops",
              "enet_ms_5_views", "enet_impl", "views_from"):
        if k in res["extra"]:
            e[k] = res["extra"][k]
    return e


def side_configs(args, rank, world, barrier):
    """BASELINE configs[2] and [3] on the default N = 1 line: detect, detect + mask head, the 5-view image path
    from feature maps and from RGB pixels -- each with its own network (the config decides the architecture), the same pipelines /
    steps / warm-up as the headline, one after the other"""
    import argparse as _ap
    import gc
    import torch
    out = {}
    plan = [("detect", "detect", False, False, "config[2]: backbone + RPN + decode / top-k / 3D NMS + RoI pooling + classifier (no mask head)"),
            ("detect_masks", "detect", True, False, "config[2] in full: + mask head on a fixed deterministic detection set (the first "
                                                    "%d post-NMS RoIs stand in as detections)" % args.mask_boxes),
            ("images", "images", False, False, "config[3]: 5-view back-projection gather (feature maps given) + colour/geometry backbone + RPN"),
            ("images_rgb", "images", False, True, "config[3] from pixels: 5 RGB views through the ENet encoder (csrc/enet.hip) inside the step, "
                                                  "then as `images`")]
    for key, workload, masks, rgb, what in plan:
        a = _ap.Namespace(**vars(args))
        a.masks, a.rgb, a.from_depth, a.group = masks, rgb, False, 1
        a.inflight = default_inflight(workload)
        a.no_streamed = True
        t0 = time.perf_counter()
        try:
            net, cfg, _ = build_net(workload, masks=masks, rgb=rgb)
            res = run_chunk_pipeline(net, cfg, a, rank, world, workload, barrier, masks=masks)
            out[key] = config_entry(res, a.steps, workload, masks=masks, rgb=rgb, what=what)
            out[key]["bench_wall_s"] = time.perf_counter() - t0
        except Exception as e:                                   # a side config must never take the headline down
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        net = res = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a mismatched n_gpus\n"
                         % (args.gpus, world))
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)

    import torch
    import torch.distributed as dist
    # functional test hook (one-GPU boxes): SIS3D_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and swaps RCCL for gloo (RCCL refuses
    # two ranks on one device); everything else -- sharding, gather, merge, timing, the JSON line -- is the N-rank code path
    share = bool(os.environ.get("SIS3D_BENCH_SHARE_GPU"))
    if share:
        local = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        sys.stderr.write("bench.py: rank %d needs GPU %d, this box exposes %d\n" % (rank, local, torch.cuda.device_count()
                                                                                       if torch.cuda.is_available() else 0))
        return 2
    use_dist = world > 1 or bool(os.environ.get("SIS3D_FORCE_DIST"))     # FORCE: exercise the RCCL path on one GPU
    cpus = None
    if world > 1:
        # N ranks share one host: keep each rank's torch-CPU helpers (synthetic inputs, weight init) from spawning a thread
        # per core each, and every rank on its own block of cores (the launcher thread of a rank enqueues ~100 us of work per
        # chunk: eight of them must not migrate across each other); the timed path is GPU-only
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))
        from sis3d import parallel as _par
        cpus = _par.pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    torch.cuda.set_device(local)
    if use_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    workload = args.workload
    if workload == "auto":
        workload = "backbone_rpn"                  # the SAME headline workload at every N (config[1], weak scaling)
    # both workloads ride on every line of the default and the scene run, at every N, under the same two keys
    both = args.workload in ("auto", "scene") and not args.masks and not args.no_graph and not args.no_side_workloads
    if args.inflight <= 0 and workload != "scene":
        args.inflight = default_inflight(workload)

    from sis3d import ops
    ops.lib()
    net, cfg, sd = build_net(workload, masks=args.masks, rgb=args.rgb)
    kt = time_dominant_kernel(net) if rank == 0 else 0.0
    kt_direct = 0.0
    if rank == 0 and ops.WINOGRAD:
        ops.set_winograd(False)                     # the direct fp32 MFMA kernel on the same layer, for the record
        try:
            kt_direct = time_dominant_kernel(net)
        finally:
            ops.set_winograd(True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if not use_dist:
            return dt
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def chunk_pipeline_side():
        saved = args.inflight
        args.inflight = default_inflight("backbone_rpn")
        cp = run_chunk_pipeline(net, cfg, args, rank, world, "backbone_rpn", barrier)
        args.inflight = saved
        cp["dt"] = max_over_ranks(cp["dt"])
        return cp

    def scene_side(steps):
        saved = args.inflight
        if workload != "scene":
            args.inflight = 0                                  # the scene picks its own number of streams
        sc = run_scene(net, args, rank, world, args.scene_chunks, barrier, steps=steps, want_table=(world == 1 and rank == 0))
        args.inflight = saved
        sc["dt"] = max_over_ranks(sc["dt"])
        return sc

    stages = None
    wino = wino_accounting(net) if rank == 0 and workload != "images" else None                  # one chunk alone (stages)
    # the step's own accounting: pipelines of >= 2 chunks in flight capture the shared-chip dispatch
    nfl_step = max(1, args.inflight)
    wino_step = wino_accounting(net, shared=(nfl_step >= 2 and not args.no_graph)) if wino is not None else None
    if rank == 0 and world == 1 and workload == "backbone_rpn" and not args.no_stages and not args.no_graph:
        stages = time_stages(net, wino)
    side = {}
    scene_steps = args.scene_steps or max(1, min(args.steps, 20))
    cp = sc = None
    if workload == "scene":
        if both:
            cp = chunk_pipeline_side()
        res = sc = scene_side(args.steps if not args.scene_steps else args.scene_steps)
        res.setdefault("steps", args.steps)
        dt = res["dt"]
    else:
        res = run_chunk_pipeline(net, cfg, args, rank, world, workload, barrier, masks=args.masks)
        dt = res["dt"] = max_over_ranks(res["dt"])
        if both:
            cp = res
            sc = scene_side(scene_steps)
    if both:
        side["chunk_pipeline"] = chunk_pipeline_entry(cp["vox_per_step"] * args.steps / cp["dt"], "voxels/s", cp["dt"] / args.steps * 1e3,
                                                      cp["vox_per_step"] / VOXELS / world, cp["single_ms"])
        for mode, key in (("grid", "streamed"), ("sdf", "streamed_sdf")):
            st = (cp.get("streamed") or {}).get(mode)
            if st is not None:
                if "dt" in st:
                    st["dt"] = max_over_ranks(st["dt"])
                    side["chunk_pipeline"][key] = streamed_entry(st, cp["dt"], args.steps, cp["vox_per_step"], cp["nfl"], world, mode)
                else:
                    side["chunk_pipeline"][key] = st
        side["scene"] = scene_entry(sc["vox_per_step"] * sc["steps"] / sc["dt"], "voxels/s", sc["dt"] / sc["steps"] * 1e3, sc["steps"],
                                    sc["extra"])
        if not args.no_streamed and not args.masks and not args.no_graph:
            # the same scene with every chunk uploaded from pinned host memory inside the timed region (same runner: same graphs)
            try:
                saved = args.inflight
                if workload != "scene":
                    args.inflight = 0
                ss = run_scene(net, args, rank, world, args.scene_chunks, barrier, steps=sc["steps"], streamed=True, runner=sc.get("runner"))
                args.inflight = saved
                ss["dt"] = max_over_ranks(ss["dt"])
                n_local = ss["extra"]["chunks_on_this_rank"]
                side["scene"]["streamed"] = {
                    "value": ss["vox_per_step"] * ss["steps"] / ss["dt"], "unit": "voxels/s", "ms_per_scene": ss["dt"] / ss["steps"] * 1e3,
                    "ratio_to_resident": sc["dt"] / sc["steps"] / (ss["dt"] / ss["steps"]),
                    "host_bytes_per_scene_per_gpu": n_local * 2 * VOXELS * 4,
                    "records_gathered": ss["extra"]["records_gathered"], "kept_after_scene_nms": ss["extra"]["kept_after_scene_nms"],
                    "how": "every scene uploads this rank's %d chunks (3.54 MB each, pinned host memory): the first node of each "
                           "pipeline's captured graph pulls its chunk across PCIe (mailbox slot written by the host, sis3d_mail_upload); "
                           "the host's only call per chunk is the graph launch" % n_local}
            except Exception as e:
                side["scene"]["streamed"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world > 1 and rank == 0:
            # rank 0 alone on the same scene, same process: the 1-GPU reference point of the strong-scaling figure
            saved = args.inflight
            args.inflight = 0
            solo = run_scene(net, args, 0, 1, args.scene_chunks, lambda: torch.cuda.synchronize(), group="solo",
                             steps=max(3, min(10, scene_steps)))
            args.inflight = saved
            sv = solo["vox_per_step"] * solo["steps"] / solo["dt"]
            side["scene"]["single_gpu"] = {"value": sv, "unit": "voxels/s", "ms_per_scene": solo["dt"] / solo["steps"] * 1e3,
                                           "steps": solo["steps"], "how": "rank 0 alone, same scene, same process, no collective"}
            side["scene"]["speedup_vs_1gpu"] = side["scene"]["value"] / sv
        if world == 1 and rank == 0 and args.scene_chunks >= 8 and sc.get("table") is not None:
            # what ONE rank of an 8-GPU run does per scene, minus the collective: its own chunks (0, 8, 16, 24: one graph launch),
            # its rows written over a FULL scene's gathered table (the other 28 chunks' blocks come from the run above), the
            # whole-scene merge of all of it -- so the 1 -> 8 ceiling is on the N = 1 line
            saved_i = args.inflight
            args.inflight = 0
            try:
                sh = run_scene(net, args, 0, 1, args.scene_chunks, lambda: torch.cuda.synchronize(), group="solo", steps=scene_steps,
                               emulate=(0, 8), gathered=sc["table"])
            finally:
                args.inflight = saved_i
            try:
                from sis3d import parallel as _par
                coll_us, coll_how = collective_latency_world_of_one(sh["extra"]["chunks_on_this_rank"] * _par.block_floats(200))
            except Exception as e:
                coll_us, coll_how = None, "%s: %s" % (type(e).__name__, e)
            share_ms = sh["dt"] / sh["steps"] * 1e3
            total_ms = share_ms + (coll_us or 0.0) * 1e-3
            side["scene"]["share_of_one_rank_at_8"] = {
                "chunks": sh["extra"]["chunks_on_this_rank"], "ms": share_ms,
                "collective_us_world_of_one": coll_us, "collective_how": coll_how,
                "ms_with_collective": total_ms,
                "records_merged": sh["extra"]["records_gathered"], "kept_after_scene_nms": sh["extra"]["kept_after_scene_nms"],
                "one_graph_launch_per_scene": sh["extra"]["one_graph_launch_per_scene"],
                "ceiling_speedup_at_8": (sc["dt"] / sc["steps"] * 1e3) / total_ms,
                "ceiling_without_collective_term": (sc["dt"] / sc["steps"] * 1e3) / share_ms,
                "how": "this GPU alone as rank 0 of 8: its scene_chunks/8 chunks (one graph launch), then the merge of the FULL scene's "
                       "gathered table (its own rows fresh, the other ranks' rows from the 32-chunk run above), PLUS the latency of the "
                       "scene's one collective measured in an RCCL world of one on this GPU (collective_us_world_of_one; serial in "
                       "this sum, although on hardware it overlaps with the next scene's detect).  An emulation on one GPU, not a "
                       "measurement of an 8-GPU run: no xGMI transfer, no launch skew between ranks"}
    if sc is not None:
        sc.pop("runner", None)                       # the scene's engines and graphs are not needed any more
    if both and rank == 0 and cp.get("wino_flops") is not None:
        # the same two fractions every config sub-object carries (config_entry), for config[1] itself
        ce = config_entry(cp, args.steps, "backbone_rpn")
        side["chunk_pipeline"].update({k: ce[k] for k in ("fp32_frac", "hbm_frac", "executed_gflop_per_chunk", "algorithmic_gflop_per_chunk")
                                       if k in ce})
    if rank == 0 and world == 1 and args.workload == "auto" and not args.masks and not args.no_graph and not args.no_side_configs:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        side.update(side_configs(args, rank, world, barrier))
    if rank == 0 and world == 1 and not args.masks and args.split_line and not args.no_graph:
        # SEPARATELY REPORTED (never the headline): the headline workload with the balanced k3 convs on the bf16 matrix
        # pipe, operands split hi + lo (csrc/conv3d_b16.hip); `value` above stays on the exact-fp32 kernels
        try:
            ops.set_split_bf16(True)
            st2 = None
            try:
                if workload == "scene":
                    r2 = run_scene(net, args, rank, world, args.scene_chunks, barrier, steps=max(3, res["steps"] // 4))
                else:
                    r2 = run_chunk_pipeline(net, cfg, args, rank, world, workload, barrier, masks=args.masks)
                    r2["steps"] = args.steps
                    if stages is not None:
                        st2 = time_stages(net, {"backbone": 0.0, "backbone_rpn": 0.0})
            finally:
                ops.set_split_bf16(False)
            diffs = {}
            if res.get("snap") and r2.get("snap"):
                for k in sorted(res["snap"]):
                    if k in r2["snap"] and res["snap"][k].shape == r2["snap"][k].shape:
                        diffs[k] = float((res["snap"][k] - r2["snap"][k]).abs().max())
            v2 = r2["vox_per_step"] * r2["steps"] / r2["dt"]
            side["split_bf16"] = {
                "value": v2, "unit": "voxels/s", "ms_per_step": r2["dt"] / r2["steps"] * 1e3, "steps": r2["steps"],
                "single_chunk_latency_ms": r2["single_ms"],
                "speedup_vs_value": v2 / (res["vox_per_step"] * res.get("steps", args.steps) / res["dt"]),
                "max_abs_diff_vs_fp32_path": diffs,
                **({"records_gathered": r2["extra"]["records_gathered"], "kept_after_scene_nms": r2["extra"]["kept_after_scene_nms"]}
                   if workload == "scene" else {}),
                **({"stages": {k: (v if not isinstance(v, dict) else {"ms": v["ms"], "hbm_frac": v["hbm_frac"], "voxels_per_s": v["voxels_per_s"]})
                               for k, v in st2.items() if k != "how"}} if st2 else {}),
                "arithmetic": "k3 convs that run on conv3d_k3t16 (rpn_net x2, geometry2[0], Bottleneck conv2 of the unfused blocks): "
                              "v_mfma_f32_16x16x32_bf16 on operands split x = hi + lo, products ah*bh + ah*bl + al*bh, fp32 accumulate; "
                              "everything else exact fp32",
                "status": "opt-in (ops.set_split_bf16), NOT the headline: not the reference's fp32 arithmetic; parity tests at the "
                          "unchanged 1e-4 tolerances pass in this mode (tests/test_gpu_conv_b16.py)"}
        except Exception as e:                       # the separately reported line must never take the headline down
            ops.set_split_bf16(False)
            side["split_bf16"] = {"error": "%s: %s" % (type(e).__name__, e)}
    steps_timed = res.get("steps", args.steps)
    ms = dt / steps_timed * 1e3
    value = res["vox_per_step"] * steps_timed / dt

    line = None
    if rank == 0:
        nchunk_step = res["vox_per_step"] / VOXELS / world           # chunks per GPU per step
        algo = {k: v * nchunk_step for k, v in ALGO[workload].items()}
        line = {
            "metric": "voxels/sec forward on 96x48x96 chunks",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": steps_timed, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": ("strong" if workload == "scene" else "weak"), "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[workload] + (" + mask head" if args.masks else ""),
                       "value_is": ("chunk_pipeline" if workload == "backbone_rpn" else workload) + (
                           " (the same workload at every N; the `scene` key of this line is the config[4] figure)" if both and workload != "scene"
                           else ""),
                       "chunk": [96, 48, 96], "hip_graph": not args.no_graph, "parallelism": "chunk-dp%d" % world,
                       "hw_queues": hw_queues(),
                       **({"rank0_cpus": "%d logical CPUs (block of this rank: sis3d.parallel.pin_rank_to_cpus)" % len(cpus)} if cpus else {}),
                       **({"TEST_HOOK": "all ranks share GPU 0, gloo instead of RCCL: functional run, not a measurement"} if share else {}),
                       "chunks_per_step_per_gpu": nchunk_step, "single_chunk_latency_ms": res["single_ms"], **res["extra"]},
            "roofline": roofline_entry(kt, kt_direct, ops.WINOGRAD),
            "step_roofline": {"hbm_frac": algo["bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "hbm_gbs_algorithmic": algo["bytes"] / (ms * 1e-3) / 1e9,
                              **({"fp32_frac": executed_flops(algo["flops"], wino_step["backbone_rpn"] * nchunk_step) / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                                  "executed_gflop_per_step": executed_flops(algo["flops"], wino_step["backbone_rpn"] * nchunk_step) / 1e9,
                                  "fp32_frac_is": "executed MFMA FLOPs of backbone + RPN (Winograd layers: algorithmic / 3.375; counted under the "
                                                  "dispatch regime the step's pipelines captured) / time / 157.3 TF"}
                                 if wino_step is not None else {}),
                              "algorithmic_tflops": algo["flops"] / (ms * 1e-3) / 1e12,
                              "binding": "fp32 FLOPs (AI 163 FLOP/B >> 20 FLOP/B machine balance)"},
        }
        line.update(side)
        if stages is not None:
            line["stages"] = stages
        if world == 1 and ops.WINOGRAD and not args.no_live_pmc and not args.no_graph:
            # the dominant kernel's HBM traffic measured by this run's own counter passes; the committed figure stays beside it
            tb, how = live_pmc_traffic()
            r = line["roofline"]
            r["traffic_committed"], r["traffic_committed_source"] = r["traffic"], r["traffic_source"]
            if tb:
                r["traffic"], r["traffic_source"] = tb, "live: PMC passes of this run"
                r["traffic_ratio"] = tb / DOMINANT_BYTES
            r["traffic_live"] = how
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(workload, sd, cfg, args.cpu_seconds)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        bad = fracs_above_one(line)
        if bad:
            line["frac_errors"] = ["%s = %.3f is not a fraction of a roof" % b for b in bad]
        emit(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
