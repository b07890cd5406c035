"""bench.py's streamed chunk pipeline by upload method, one process (same stream pool): resident step, streamed step, host enqueue"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    args = bench.parse(["--steps", "60", "--warmup", "10"])
    args.inflight = 4
    net, cfg, sd = bench.build_net("backbone_rpn")
    for copy in sys.argv[1:] or ["kernel", "own", "kernel"]:
        os.environ["SIS3D_FEED_COPY"] = copy
        r = bench.run_chunk_pipeline(net, cfg, args, 0, 1, "backbone_rpn", torch.cuda.synchronize)
        ms = r["dt"] / args.steps * 1e3
        out = "%-13s resident %.3f ms/step |" % (copy, ms)
        for mode in ("grid", "sdf"):
            s = r["streamed"][mode]
            out += (" %s %.3f ms (ratio %.3f, host %.3f)" % (mode, s["dt"] / args.steps * 1e3, r["dt"] / s["dt"], s["host_ms_per_step"])) if "dt" in s else " %s %s" % (mode, s)
        print(out, flush=True)


if __name__ == "__main__":
    main()
