"""us per launch of the Bottleneck(32,32) body on the 48x24x48 map: direct-convolution body (sis3d_bottleneck16) vs the Winograd body
(sis3d_bottleneck_wino), HIP events around 50 back-to-back launches, best of 5."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
from sis3d import ops  # noqa: E402


def cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last_3d)


def main():
    torch.manual_seed(0)
    for (pl, cio, c2, dims) in [(32, 32, 32, (48, 24, 48)), (32, 32, 0, (48, 24, 48)), (32, 64, 0, (48, 24, 48))]:
        x, y1 = cl(torch.randn(1, cio, *dims)), cl(torch.relu(torch.randn(1, pl, *dims)))
        pc2 = ops.PackedConv(torch.randn(pl, pl, 3, 3, 3).cuda() * 0.03, torch.randn(pl).cuda() * 0.1)
        pc3 = ops.PackedConv(torch.randn(cio, pl, 1, 1, 1).cuda() * 0.1, torch.randn(cio).cuda() * 0.1)
        stage = dict(pc=ops.PackedConv(torch.randn(c2, cio, 1, 1, 1).cuda() * 0.1, torch.randn(c2).cuda() * 0.1), relu=True) if c2 else None
        out = ops.new_act(cio, dims, x.device)
        for wino in (False, True):
            ops.BNECK_WINO = wino
            for _ in range(30):
                ops.bottleneck16(y1, pc2, pc3, x, out=out, stage=stage)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    ops.bottleneck16(y1, pc2, pc3, x, out=out, stage=stage)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
            print("planes %d cio %d next-conv1 %d @%s  %s body: %.1f us" % (pl, cio, c2, "x".join(map(str, dims)), "Winograd" if wino else "direct  ", best))


if __name__ == "__main__":
    main()
