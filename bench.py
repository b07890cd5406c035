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

from benchlib.constants import *  # noqa: E402,F401,F403
from benchlib.launch import free_port, launch_command, self_launch, emit  # noqa: E402,F401
from benchlib.roofline import (time_dominant_kernel, executed_flops, wino_accounting, time_stages, pmc_traffic, live_pmc_traffic,  # noqa: E402,F401
                               roofline_entry, fracs_above_one, DOMINANT_BYTES)
from benchlib.cpu_baseline import cpu_model, cpu_threads_rule, _median_runs, cpu_baseline_reference, cpu_baseline  # noqa: E402,F401


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
    ap.add_argument("--calibrate", action="store_true", help="time every window of the candidate streams and keep the fastest (r5's default; "
                                                             "since r6 only the fallback when the pipelines' queues could not be verified)")
    ap.add_argument("--no-calibrate", action="store_true", help="keep the pipelines on the first streams of the pool instead of choosing "
                    "the window of streams by measurement (PipelinedEngines.calibrate / SceneRunner.calibrate)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed-input variants (fresh chunks from pinned host memory)")
    ap.add_argument("--no-side-configs", action="store_true", help="N = 1 default run: skip the detect / detect_masks / images / "
                    "images_rgb sub-objects (BASELINE configs[2], [3])")
    ap.add_argument("--no-side-workloads", action="store_true", help="time only the headline workload (no chunk_pipeline / scene side keys)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dump-scene", default=None, help="scene workload: rank 0 writes the gathered record table and the whole-scene "
                                                       "keep list of the last timed scene to this .npz (tests/test_gpu_multi.py)")
    ap.add_argument("--selftest-cpu", action="store_true", help=argparse.SUPPRESS)   # launch / rendezvous logic under gloo, no GPU
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launching N ranks --


def want_calibration(args, pipes):
    """r6: pipelines whose streams sit on VERIFIED distinct hardware queues (engine.distinct_queue_streams) are not re-placed by timing;
    --calibrate forces the r5 behaviour (time every window of the candidate streams), --no-calibrate forbids it"""
    if args.no_calibrate:
        return False
    return bool(args.calibrate) or not getattr(pipes, "placement_verified", False)


def stream_placement(pipes):
    from sis3d import engine
    try:
        o = engine.own_streams(pipes.engines[0].device)
        mine = [o["streams"].index(s) if s in o["streams"] else -1 for s in pipes.streams]
        return {"verified_distinct_queues": bool(pipes.placement_verified), "own_streams": len(o["streams"]),
                "queue_class_of_own_streams": o["klass"], "null_stream_class": o["null_class"], "pipelines_on_own_streams": mine,
                "calibrated": bool(pipes.stream_window_times)}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


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
    if want_calibration(args, eng) and nfl >= 2:
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
            "how": "every step each of the %d pipelines pulls a FRESH chunk from pinned host memory inside its captured graph (mailbox + "
                   "piggyback upload, DESIGN.md section 4); ring of %d host chunks per pipeline; every upload is inside the timed region"
                   % (nfl, st["ring"]), "stage_ahead": st.get("stage_ahead")}


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
    if want_calibration(args, eng) and nfl >= 2 and not args.no_graph:
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
    extra = {"chunks_per_graph": grp, "streams_per_gpu": nfl, "stream_placement": stream_placement(eng)}
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
    if want_calibration(args, runner.pipes) and not args.no_graph and not args.masks and runner.calibration is None and not streamed:
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
    if getattr(args, "dump_scene", None) and rank == 0 and group != "solo" and emulate is None and not streamed:
        # tests/test_gpu_multi.py: the gathered record table and the whole-scene keep list of the LAST timed scene, as rank 0 holds them
        import numpy as _np
        _np.savez(args.dump_scene, recs=recs.detach().cpu().numpy(), keep=keep.detach().cpu().numpy(), world=_np.int64(world),
                  n_chunks=_np.int64(n_chunks), stride=_np.float64(args.scene_stride))
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


def whole_scene_entry(steps=20, dims=(160, 64, 224)):
    """SURVEY 8(f) row f2 on the line: ONE non-chunked whole-scene grid -- what the reference's benchmark / test
    modes feed the network when a scene is not cut into chunks (lib/datasets/dataset.py:192-205) -- through the full detection
    pass (backbone + RPN + proposals + RoI pooling + classifier), captured graph, one grid alone on the GPU"""
    import torch
    from sis3d import synthetic
    from sis3d.engine import ChunkEngine
    net, cfg, _ = build_net("detect")
    eng = ChunkEngine(net, dims=dims, stage="detect")
    eng.load(synthetic.synth_chunk(5, dims))
    eng.prepare()
    for _ in range(3):
        eng.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nv = dims[0] * dims[1] * dims[2]
    n = int(eng.out["num"].item())
    return {"grid": list(dims), "voxels": nv, "ms": ms, "value": nv / (ms * 1e-3), "unit": "voxels/s", "steps": steps, "rois": n,
            "equivalent_chunks": nv / VOXELS, "how": "one %dx%dx%d grid, full detection pass, captured graph replayed back to back" % dims}


def compute_projection_entry(seconds=4.0, dims=(96, 48, 96), views=5):
    """SURVEY 8(f) row f1 on the line: the device compute_projection (lib/layer_utils/projection.py:52-121: frustum test, depth test,
    ordered index lists) of `views` depth maps over one chunk grid, beside the CPU restatement of the same function on this box"""
    import torch
    from sis3d import config, ops, synthetic
    from sis3d.layer_utils.projection import ProjectionHelper
    c = config.scannet_benchmark_cfg()
    h = ProjectionHelper(c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.DEPTH_SHAPE, list(dims), c.VOXEL_SIZE)
    depth, c2w, w2g = synthetic.synth_cameras(1, views, dims, c.VOXEL_SIZE)
    d = depth.cuda()
    params = torch.stack([h.view_params(c2w[v], w2g[v]) for v in range(views)]).cuda()
    nvox = dims[0] * dims[1] * dims[2]
    out = (torch.empty(views, nvox + 1, dtype=torch.int64, device="cuda"), torch.empty(views, nvox + 1, dtype=torch.int64, device="cuda"))
    run = lambda: ops.compute_projection(d, params, dims, c.DEPTH_SHAPE, c.INTRINSIC, c.PROJ_DEPTH_MIN, c.PROJ_DEPTH_MAX, c.VOXEL_SIZE, out=out)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    e = {"views": views, "grid": list(dims), "us": us, "visible_voxels_per_view": out[0][:, 0].tolist(),
         "list_write_gbs": 2 * views * (nvox + 1) * 8 / us / 1e3,
         "how": "%d views over one %dx%dx%d grid: all views in one launch sequence, no host sync, HIP events over 50 calls" % ((views,) + tuple(dims))}
    try:
        from benchlib.cpu_baseline import cpu_compute_projection
        e["cpu"] = cpu_compute_projection(depth, c2w, w2g, c, dims, views, seconds)
        e["speedup_vs_cpu"] = e["cpu"]["us"] / us
    except Exception as ex:                                      # the CPU leg must never take the line down
        e["cpu"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return e


def summary(line):
    """every scalar result of the line in ONE flat dict under config.summary: the driver keeps `config` whole but only the key NAMES
    of the other sub-objects and the last 8.8 KB of stdout"""
    def g(*path):
        o = line
        for p in path:
            if not isinstance(o, dict) or p not in o:
                return None
            o = o[p]
        return round(o, 4) if isinstance(o, float) else o
    s = {"value_G": round(line["value"] / 1e9, 4), "ms_per_step": round(line["ms_per_step"], 4),
         "dominant_kernel_us": g("roofline", "launch_us"), "dominant_kernel_frac": g("roofline", "frac"),
         "traffic_ratio": g("roofline", "traffic_ratio"),
         "step_fp32_frac": g("step_roofline", "fp32_frac"), "step_hbm_frac": g("step_roofline", "hbm_frac"),
         "backbone_ms": g("stages", "backbone", "ms"), "backbone_hbm_frac": g("stages", "backbone", "hbm_frac"),
         "streamed_ratio": g("chunk_pipeline", "streamed", "ratio_to_resident"),
         "streamed_sdf_ratio": g("chunk_pipeline", "streamed_sdf", "ratio_to_resident"),
         "scene_ms": g("scene", "ms_per_scene"), "scene_streamed_ratio": g("scene", "streamed", "ratio_to_resident"),
         "scene_kept": g("scene", "kept_after_scene_nms"), "scene_records": g("scene", "records_gathered"),
         "share_at_8_ms": g("scene", "share_of_one_rank_at_8", "ms_with_collective"),
         "ceiling_speedup_at_8": g("scene", "share_of_one_rank_at_8", "ceiling_speedup_at_8"),
         "whole_scene_ms": g("whole_scene", "ms"), "whole_scene_G": (round(line["whole_scene"]["value"] / 1e9, 4)
                                                                     if isinstance(line.get("whole_scene"), dict) and "value" in line["whole_scene"] else None),
         "compute_projection_us": g("compute_projection", "us"), "compute_projection_cpu_us": g("compute_projection", "cpu", "us"),
         "cpu_baseline_voxels_s": g("cpu_baseline", "value"), "cpu_port_voxels_s": g("cpu_baseline", "port", "value")}
    for k in ("detect", "detect_masks", "images", "images_rgb"):
        if isinstance(line.get(k), dict) and "value" in line[k]:
            s[k + "_G"] = round(line[k]["value"] / 1e9, 4)
            s[k + "_fp32_frac"] = g(k, "fp32_frac")
            s[k + "_single_ms"] = g(k, "single_chunk_latency_ms")
    s["mask_head_ms"] = g("detect_masks", "mask_head_ms")
    s["mask_head_fp32_frac"] = g("detect_masks", "mask_head_fp32_frac")
    s["enet_ms_5_views"] = g("images_rgb", "enet_ms_5_views")
    return {k: v for k, v in s.items() if v is not None}


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
                    "how": "every scene pulls this rank's %d chunks (3.54 MB each) from pinned host memory inside the pipelines' graphs" % n_local}
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
                "how": "EMULATION on one GPU of rank 0 of 8: its 4 chunks + the merge of the full gathered table + the all-gather's latency in "
                       "an RCCL world of one; no xGMI transfer, no skew between ranks"}
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
        for key, fn in (("whole_scene", lambda: whole_scene_entry(max(5, min(args.steps, 20)))),
                        ("compute_projection", lambda: compute_projection_entry(min(4.0, args.cpu_seconds) if not args.no_cpu_baseline else 0.0))):
            try:
                side[key] = fn()
            except Exception as e:                       # a side measurement must never take the headline down
                side[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            gc.collect()
            torch.cuda.empty_cache()
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
        line["config"]["summary"] = summary(line)
        bad = fracs_above_one(line)
        if bad:
            line["frac_errors"] = ["%s = %.3f is not a fraction of a roof" % b for b in bad]
        emit(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
