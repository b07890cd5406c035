#!/usr/bin/env python
"""Where does a scene's wall time go?  Host-side phase timers (with device syncs between phases) around SceneRunner.infer's
steps for a 4-chunk and a 32-chunk scene on one GPU.  Usage (GPU box): python tools/scene_profile.py [n_chunks ...]"""
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


def main():
    ns = [int(a) for a in sys.argv[1:]] or [4, 32]
    net, cfg, sd = bench.build_net("scene")
    for n in ns:
        for nfl in (3, 4):
            runner = SceneRunner(net, synthetic.CHUNK_DIMS, inflight=nfl)
            chunks = [(c, (96.0 * (c % 4), 0.0, 96.0 * (c // 4)), synthetic.synth_chunk(c).cuda()) for c in range(n)]
            for _ in range(3):
                runner.infer(chunks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                runner.infer(chunks)
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) / 20 * 1e3
            # phases
            T = {}
            for _ in range(10):
                torch.cuda.synchronize(); a = time.perf_counter()
                local = runner.run_chunks(chunks)
                torch.cuda.synchronize(); b = time.perf_counter()
                blocks = parallel.gather_blocks(local, n, runner.k_rows)
                torch.cuda.synchronize(); c = time.perf_counter()
                parallel.merge_scene(blocks, runner.k_rows, ops.nms, 0.1)
                torch.cuda.synchronize(); d = time.perf_counter()
                for k, v in (("chunks", b - a), ("gather", c - b), ("merge", d - c)):
                    T[k] = T.get(k, 0.0) + v * 1e3 / 10
            print("n_chunks %2d inflight %d: %.3f ms per scene (%.3f ms per chunk) phases %s" % (
                n, nfl, tot, tot / n, {k: round(v, 3) for k, v in T.items()}))


if __name__ == "__main__":
    main()
