import sys, time, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "3d-sis_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from sis3d import synthetic
from sis3d.engine import ChunkEngine
net, cfg, sd = bench.build_net("backbone_rpn")
for nstreams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    engs = []
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            e = ChunkEngine(net, stage="rpn")
            e.load(synthetic.synth_chunk(i).cuda())
            e.prepare()
            engs.append(e)
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.run()
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("streams=%d  %.3f ms per chunk  %.1f Mvox/s" % (nstreams, dt / (100 * nstreams) * 1e3, 442368 * 100 * nstreams / dt / 1e6))
