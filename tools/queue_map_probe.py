"""Does the speed of four chunk pipelines depend on WHICH of torch's pool streams they run on (i.e. on the hardware queues HIP
mapped those streams to)?  The captured graphs are replayed on pool streams o .. o+3 for every offset o."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from sis3d import synthetic  # noqa: E402
from sis3d.engine import PipelinedEngines  # noqa: E402
from sis3d.scene import SceneRunner  # noqa: E402


def main():
    import sis3d.scene as _sc
    print("scene serial part: %s" % ("pipelined path, on the last pipeline's stream" if _sc.MERGE_STREAM and _sc.MERGE_ON_PIPELINE else
                                      "pipelined path, dedicated merge stream" if _sc.MERGE_STREAM else "on the caller's (null) stream"))
    torch.cuda.init()
    pool = [torch.cuda.Stream() for _ in range(32)]
    assert len({s.cuda_stream for s in pool}) == 32
    net, cfg, sd = bench.build_net("backbone_rpn")
    n = 4
    eng = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        eng.load(i, synthetic.synth_chunk(i))
    eng.prepare(warmup=2)
    bench.preheat(eng.run, 250.0)
    dnet, dcfg, _ = bench.build_net("scene")
    runner = SceneRunner(dnet, synthetic.CHUNK_DIMS, inflight=4)
    chunks = [(c, bench.scene_origin(c, 80.0), synthetic.synth_chunk(c).cuda()) for c in range(32)]
    for _ in range(3):
        runner.infer(chunks)
    torch.cuda.synchronize()
    print("offset  stream set                    chunk_pipeline ms/step   scene ms")
    sets = [("pooled(role)", list(eng.streams), list(runner.pipes.streams))]
    for o in list(range(0, 12)) + [16, 20, 24, 28]:
        sets.append(("pool[%d:%d]" % (o, o + 4), pool[o:o + 4], pool[o:o + 4]))
    sets.append(("pool[0,8,16,24] (same queue?)", pool[0:32:8], pool[0:32:8]))
    sets.append(("pool[0,2,4,6]", pool[0:8:2], pool[0:8:2]))
    sets.append(("pool[1,3,5,7]", pool[1:8:2], pool[1:8:2]))
    for name, s1, s2 in sets:
        eng.streams = s1
        for _ in range(10):
            eng.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.run()
        torch.cuda.synchronize()
        cp = (time.perf_counter() - t0) / 50 * 1e3
        runner.pipes.streams = s2
        for _ in range(2):
            runner.infer(chunks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            runner.infer(chunks)
        torch.cuda.synchronize()
        sc = (time.perf_counter() - t0) / 8 * 1e3
        print("%-32s %8.3f %12.3f" % (name, cp, sc))


if __name__ == "__main__":
    main()
