#!/usr/bin/env python
"""Whole-scene merge (parallel.merge_scene) of a 32-chunk scene's gathered record blocks: time of its pieces and of the two
NMS algorithms behind sis3d_nms (one-workgroup sweep vs sparse suppressor table + parallel resolve), keep lists compared.
Usage (GPU box): python tools/merge_time.py [n_chunks]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sis3d import ops, parallel, synthetic  # noqa: E402
from sis3d.scene import SceneRunner  # noqa: E402


def ev_time(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    net, cfg, sd = bench.build_net("scene")
    runner = SceneRunner(net, synthetic.CHUNK_DIMS, inflight=3)
    chunks = [(c, (96.0 * (c % 4), 0.0, 96.0 * (c // 4)), synthetic.synth_chunk(c).cuda()) for c in range(n)]
    local = runner.run_chunks(chunks)
    blocks = parallel.gather_blocks(local, n, runner.k_rows).clone()
    torch.cuda.synchronize()
    recs, keep = parallel.merge_scene(blocks, runner.k_rows, ops.nms, 0.1)
    boxes = recs[:, 0:6].contiguous()
    print("records %d kept %d" % (recs.shape[0], keep.numel()))
    out = {}
    for path, name in ((1, "sweep"), (2, "resolve")):
        ops.nms_set_path(path)
        k = ops.nms(boxes, 0.1)
        out[name] = k
        print("nms %-8s %8.1f us   (n = %d, kept %d)" % (name, ev_time(lambda: ops.nms_raw(boxes, 0.1)), boxes.shape[0], k.numel()))
    ops.nms_set_path(0)
    print("keep lists equal:", torch.equal(out["sweep"], out["resolve"]))
    for th in (0.0, 0.02, 0.3):
        ops.nms_set_path(1); a = ops.nms(boxes, th)
        ops.nms_set_path(2); b = ops.nms(boxes, th)
        ops.nms_set_path(0)
        print("  thresh %.2f: equal %s kept %d" % (th, torch.equal(a, b), a.numel()))

    def host_timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    print("merge_scene (host wall, synced each call): %.1f us" % host_timed(lambda: parallel.merge_scene(blocks, runner.k_rows, ops.nms, 0.1)))
    a = ops.scene_merge(blocks, runner.k_rows, 0.1)
    b = parallel.merge_scene(blocks, runner.k_rows, ops.nms, 0.1, with_chunk_ids=True)
    print("fused == torch path:", all(torch.equal(x, y) for x, y in zip(a, b)))
    print("fused scene_merge (host wall, synced each call): %.1f us" % host_timed(lambda: ops.scene_merge(blocks, runner.k_rows, 0.1)))
    print("fused scene_merge device time: %.1f us" % ev_time(lambda: ops.scene_merge_raw(blocks, runner.k_rows, 0.1)))
    k_rows = runner.k_rows
    counts = blocks[:, 0].round().long().clamp(0, k_rows)
    rows = blocks[:, 1:].reshape(n * k_rows, -1)
    valid = (torch.arange(k_rows, device="cuda").view(1, -1) < counts.view(-1, 1)).reshape(-1)
    key = torch.where(valid, rows[:, 6], torch.full_like(rows[:, 6], float("-inf")))
    print("sort (stable, %d keys): %.1f us" % (key.numel(), ev_time(lambda: torch.sort(key, descending=True, stable=True))))
    print("count readback (.item): %.1f us" % host_timed(lambda: int(counts.sum().item())))
    order = torch.sort(key, descending=True, stable=True)[1][:recs.shape[0]]
    print("index_select: %.1f us" % ev_time(lambda: rows.index_select(0, order)))


if __name__ == "__main__":
    main()
