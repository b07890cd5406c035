"""Where the one-workgroup top-k spends its time on the bench's real RPN scores: builds (here, on the CPU box:
`python tools/topk_phases.py --build`) a copy of csrc/topk.hip with -DTOPK_TIMING (100 MHz timestamps after each phase, written
to an extra buffer) and runs it on the score vector of one synthetic chunk."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_bin", "libtopk_timing.so")

if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTOPK_TIMING",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "3d-sis_amd", "csrc"), "-ffp-contract=off",
                           os.path.join(ROOT, "3d-sis_amd", "csrc", "topk.hip"), os.path.join(ROOT, "3d-sis_amd", "csrc", "api.hip"),
                           "-o", SO])
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sis3d import ops, synthetic  # noqa: E402

net, cfg, sd = bench.build_net("detect")
scene = synthetic.synth_chunk(0).cuda().float()
with torch.no_grad():
    net.detect(scene)
scores = net._prop["scores_all"].contiguous()
n = scores.numel()
srt = torch.sort(scores, descending=True).values
print("n = %d, max %.6f, 400th %.6f, 1536th %.6f, median %.3g; > 0.5: %d" % (n, srt[0], srt[399], srt[1535], srt[n // 2], int((scores > 0.5).sum())))
lib = ctypes.CDLL(SO)
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
out_s = torch.empty(400, device="cuda")
out_i = torch.empty(400, dtype=torch.int64, device="cuda")
for _ in range(3):
    rc = lib.sis3d_topk_desc_timing(ctypes.c_void_p(scores.data_ptr()), n, 400, ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
                                    ctypes.c_void_p(ts.data_ptr()), None)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    names = ["start", "loaded+minmax", "prefilter", "compacted", "ranked/end"]
    print("rc", rc, "m =", t[8], " ".join("%s +%.2f us" % (names[i], (t[i] - t[0]) * 0.01) for i in range(1, 5)))
want_s, want_i = ops.topk_desc(scores, 400)
print("matches product kernel:", bool(torch.equal(want_i, out_i)))
