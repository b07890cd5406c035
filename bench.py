#!/usr/bin/env python
"""bench.py -- voxels/sec of the 3D-SIS forward hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload backbone_rpn|detect|images] [--no-graph]

One process per GPU (for N>1 the driver launches this under torch.distributed.run; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the env).  A step = one pass of the hot path over one batch of
`--inflight` (default 3) independent synthetic 96x48x96 chunks per rank, each on its own HIP stream / captured
graph, inputs already resident in HBM (static buffers of the ChunkEngine); weights
are seeded synthetic (no checkpoints exist offline).  Chunks are independent, so ranks share nothing
on the data path (scaling: weak); the per-scene proposal all-gather is exercised by `--workload scene`.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- dominant kernel (the k3 128->256 RPN conv, 12.2 GFLOP/launch): achieved = algorithmic FLOPs /
                  mean launch duration measured live with HIP events on the launch stream; peak = 157.3 TF
                  (fp32 MFMA == fp32 vector peak of gfx950).  The HBM-roof fraction of the whole step is given
                  beside it (the metric's "% HBM roofline"): algorithmic bytes per chunk / step time / 8 TB/s.
  cpu_baseline -- the CPU oracle (torch-CPU operators = what the reference's MAX_VOLUME=0 path runs) timed on
                  this box's host cores on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))

import torch  # noqa: E402

VOXELS = 96 * 48 * 96
# algorithmic work per chunk
ALGO = {
    "backbone_rpn": dict(bytes=261.4e6, flops=42.58e9),
    "detect": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
    "images": dict(bytes=517e6 + 28.3e6 + 59.8e6 + 226.5e6, flops=29.1e9 + 24.86e9),
    "scene": dict(bytes=261.4e6 + 2 * 3.54e6 + 200 * 32768 + 8.4e6, flops=42.58e9 + 0.9e9),
}
DOMINANT_FLOPS = 2.0 * 6912 * 256 * 128 * 27        # rpn_net_level{1,2}: 12.23 GFLOP per launch
FP32_PEAK_TF = 157.3
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="backbone_rpn", choices=["backbone_rpn", "detect", "images", "scene"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--inflight", type=int, default=0, help="independent chunks in flight per GPU (HIP streams); 0 = the "
                    "measured best: 3")
    ap.add_argument("--masks", action="store_true", help="scene workload: also run the mask head on the detections that survive "
                    "the whole-scene NMS (each on the chunk / rank that produced it)")
    ap.add_argument("--group", type=int, default=1, help="chunks per captured graph (2: the pair's four RPN convs in one launch)")
    ap.add_argument("--from-depth", action="store_true", help="images workload: views arrive as depth maps + poses and the "
                    "voxel->pixel lists are computed on the device inside the timed step (sis3d_compute_projection)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def build_net(workload, masks=False):
    from sis3d import config, synthetic
    from sis3d.nets import backbones
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_IMAGES = workload == "images"
    cfg.USE_MASK = bool(masks)
    net = backbones.ScanNet_Backbone(cfg=cfg)
    net.init_modules()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)
    net.load_state_dict(sd)
    return net.cuda().eval(), cfg, sd


def time_dominant_kernel(net, iters=50):
    """mean duration of the rpn_net k3 128->256 conv launch (12.23 GFLOP), HIP events on the launch (current) stream.
    Runs BEFORE any graph is captured, on its own input: on ROCm 7.2 eager launches of these kernels between replays
    of a captured graph were observed to fault the next replay, so the bench never interleaves them."""
    from sis3d import ops
    x = ops.new_act(128, (24, 12, 24), torch.device("cuda"))
    x.normal_().clamp_(min=0)                      # post-ReLU-like activations
    conv = net.rpn_net_level1
    for _ in range(100):                           # bring the clocks up: measured cold the same launch is ~10 % slower
        conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        conv(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, KB -> B; profiles/r01_pmc_rpn_net.json).  Counters cannot be read from inside the bench."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_rpn_net.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(workload, sd, cfg, seconds):
    """the oracle's backbone+RPN (torch CPU operators, as the reference's CPU path) on the host cores"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sis3d_oracle as orc
    from sis3d import config, synthetic
    cores = os.cpu_count() or 1
    net = orc.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2))
    data = synthetic.synth_chunk(0)
    feats = i3d = i2d = None
    if workload == "images":
        feats, i3d, i2d = synthetic.synth_views(0)

    def one():
        with torch.no_grad():
            if workload == "detect":
                net.forward(data, feats, i3d, i2d)
            else:
                imageft = orc.project_views_max(feats, i3d, i2d, data.shape[2:]) if workload == "images" else None
                l1, l2 = net.backbone(data, imageft)
                net.rpn(l1, 1)
                net.rpn(l2, 2)

    # oneDNN does not scale to hundreds of threads on a 0.02-GFLOP/voxel workload: pick the best of a short sweep
    best_t, best = None, None
    for t in [c for c in (16, 32, 64, 128) if c <= cores] or [cores]:
        torch.set_num_threads(t)
        one()
        t0 = time.time()
        one()
        one()
        d = (time.time() - t0) / 2
        if best is None or d < best:
            best_t, best = t, d
        if d > 2.0 * best:
            break
    torch.set_num_threads(best_t)
    n, t0 = 0, time.time()
    while True:
        one()
        n += 1
        if time.time() - t0 >= seconds or n >= 400:
            break
    dt = time.time() - t0
    cores_used = best_t
    return dict(value=VOXELS * n / dt, unit="voxels/s", cores=cores_used, kind="port", host_cores=cores,
                sample="%d forward passes of one 96x48x96 chunk (%s; oracle = the reference's CPU operators via torch-CPU/oneDNN, "
                       "%d threads, best of a 16/32/64/128 sweep) in %.1f s" % (n, workload, cores_used, dt))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    use_dist = world > 1 or bool(os.environ.get("SIS3D_FORCE_DIST"))     # FORCE: exercise the RCCL path on one GPU
    if world > 1:
        # N ranks share one host: keep each rank's torch-CPU helpers (synthetic inputs, weight init) from spawning a thread
        # per core each; the timed path is GPU-only
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    from sis3d import synthetic, ops
    from sis3d.engine import PipelinedEngines
    ops.lib()
    net, cfg, sd = build_net(args.workload, masks=args.masks)
    kt = time_dominant_kernel(net) if rank == 0 else 0.0
    dbg = bool(os.environ.get("SIS3D_BENCH_DEBUG"))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    single_ms = None
    if args.inflight <= 0:
        args.inflight = 3
    if args.workload == "scene":
        # BASELINE config 5: 32 chunks of one scene (4 x 1 x 8 grid of 96x48x96 tiles), chunk c -> rank c mod W, per-chunk
        # detection, ONE all-gather of the record blocks, whole-scene NMS on every rank.  A step = one whole scene.
        from sis3d.scene import SceneRunner
        n_chunks = 32
        runner = SceneRunner(net, synthetic.CHUNK_DIMS, use_graph=not args.no_graph, inflight=args.inflight)
        chunks = []
        for c in range(n_chunks):
            payload = synthetic.synth_chunk(c).cuda() if c % world == rank else None     # resident in HBM, own shard only
            chunks.append((c, (96.0 * (c % 4), 0.0, 96.0 * (c // 4)), payload))
        torch.cuda.synchronize()
        for _ in range(max(1, args.warmup // 10)):
            res = runner.infer(chunks, with_masks=args.masks)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = runner.infer(chunks, with_masks=args.masks)
        barrier()
        recs, keep = res[0], res[1]
        dt = time.perf_counter() - t0
        nfl, vox_per_step = 1, n_chunks * VOXELS
        extra_cfg = {"scene_chunks": n_chunks, "records": int(recs.shape[0]), "kept_after_scene_nms": int(keep.numel())}
        if args.masks:
            extra_cfg["masks_on_this_rank"] = len(res[2])
            extra_cfg["mask_voxels_on_this_rank"] = int(sum(m.numel() for _, m in res[2].values()))
    else:
        stage = "rpn" if args.workload in ("backbone_rpn", "images") else "detect"
        nfl = max(1, args.inflight)
        from_depth = args.workload == "images" and args.from_depth
        grp = max(1, args.group)
        eng = PipelinedEngines(net, nfl, stage=stage, use_graph=not args.no_graph, group=grp, **({"from_depth": True} if from_depth else {}))
        for i in range(nfl):
          for g in range(grp):
            cid = (rank * nfl + i) * grp + g
            data = synthetic.synth_chunk(cid)
            if from_depth:
                feats = synthetic.synth_views(cid, n_per_view=0)[0]
                depth, c2w, w2g = synthetic.synth_cameras(cid, feats.shape[0], voxel_size=cfg.VOXEL_SIZE)
                with torch.cuda.stream(eng.streams[i]):
                    eng.engines[i].load_views(data, feats, depth, c2w, w2g, slot=g)
            elif args.workload == "images":
                feats, i3d, i2d = synthetic.synth_views(cid)
                eng.load(i, data, feats, i3d, i2d, slot=g)
            else:
                eng.load(i, data, slot=g)
        eng.prepare(warmup=2)
        if dbg:
            print("[bench] prepared", file=sys.stderr, flush=True)
        for _ in range(args.warmup):
            eng.run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.run()
        barrier()
        dt = time.perf_counter() - t0
        vox_per_step = world * nfl * grp * VOXELS
        extra_cfg = {"chunks_per_graph": grp}
        if from_depth:
            torch.cuda.synchronize()
            extra_cfg = {"chunks_per_graph": grp, "views_from": "depth maps + poses (lists computed on device inside the step)",
                         "visible_voxels_per_view": eng.engines[0].view_counts()}
        if rank == 0:
            # latency of ONE chunk on an otherwise idle GPU (single stream, serialised on purpose)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                eng.run(0)
                torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t1) / 50 * 1e3
    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = vox_per_step * args.steps / dt

    if rank == 0:
        nchunk_step = vox_per_step / VOXELS / world           # chunks per GPU per step
        algo = {k: v * nchunk_step for k, v in ALGO[args.workload].items()}
        line = {
            "metric": "voxels/sec forward on 96x48x96 chunks",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": ("strong" if args.workload == "scene" else "weak"), "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"backbone_rpn": "config[1]: one 96x48x96 chunk per GPU, geometry-only, HIP 3D-conv backbone + RPN "
                                                    "(convs, heads, softmax), weights seeded synthetic",
                                    "detect": "config[2] minus mask head: backbone + RPN + decode/sort/NMS + RoI pooling + classifier",
                                    "images": "config[3]: 5-view back-projection gather + colour/geometry backbone + RPN",
                                    "scene": "config[4]: 32-chunk scene sharded chunk->rank, per-chunk detection, one RCCL all-gather of "
                                             "record blocks, whole-scene 3D NMS on every rank"}[args.workload],
                       "chunk": [96, 48, 96], "hip_graph": not args.no_graph, "parallelism": "chunk-dp%d" % world,
                       "chunks_per_step_per_gpu": nchunk_step, "streams_per_gpu": nfl, "single_chunk_latency_ms": single_ms, **extra_cfg},
            "roofline": {"bound": "mfma", "kernel": "conv3d_mfma_kernel<3,1,...> rpn_net 128->256 (fp32 v_mfma_f32_32x32x2_f32)",
                         "achieved": DOMINANT_FLOPS / kt / 1e12, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": DOMINANT_FLOPS / kt / 1e12 / FP32_PEAK_TF, "traffic": pmc_traffic(),
                         "launch_us": kt * 1e6},
            "step_roofline": {"hbm_frac": algo["bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "hbm_gbs_algorithmic": algo["bytes"] / (ms * 1e-3) / 1e9,
                              "fp32_frac": algo["flops"] / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                              "binding": "fp32 FLOPs (AI 163 FLOP/B >> 20 FLOP/B machine balance)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, sd, cfg, args.cpu_seconds)
        out_line = json.dumps(line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST thing on stdout: RCCL printf()s a version banner into C stdio's buffer, which
        # would otherwise be flushed at exit, after our line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
