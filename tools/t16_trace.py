#!/usr/bin/env python
"""Per-workgroup timeline of a k3t16 launch (sis3d_conv3d_k3t16_set_trace): which CU ran each workgroup, when, for how long,
and how many workgroups a CU held at once.  Usage (GPU box): python tools/t16_trace.py mask|rpn [brick]"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sis3d import ops, synthetic  # noqa: E402


def report(buf, nblocks, label):
    t = buf[:nblocks].cpu().numpy()
    st, en, hw = t[:, 0], t[:, 1], t[:, 2]
    ok = en > 0
    st, en, hw = st[ok], en[ok], hw[ok]
    t0 = st.min()
    us = lambda x: (x - t0) / 100.0                                   # wall_clock64: 100 MHz
    # HW_ID: wave_id [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]; XCD is not in it -> key by (se, sh, cu) + time overlap
    cu = ((hw >> 8) & 0xFF).astype(np.int64)
    slot = (hw & 0xF)
    dur = (en - st) / 100.0
    print("%s: %d workgroups, kernel span %.1f us, workgroup duration mean %.1f / min %.1f / max %.1f us" % (
        label, len(st), us(en.max()), dur.mean(), dur.min(), dur.max()))
    print("   start times: first wave of launches <= 1 us: %d; started later: %d" % ((us(st) <= 1.0).sum(), (us(st) > 1.0).sum()))
    print("   wave slot histogram (HW_ID.wave_id of wave 0):", np.bincount(slot, minlength=10).tolist())
    # concurrency on a (se,sh,cu) key: 8 XCDs alias onto the same key, so divide by 8
    keys = defaultdict(list)
    for s, e, c in zip(us(st), us(en), cu):
        keys[int(c)].append((s, e))
    conc = []
    for c, iv in keys.items():
        ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
        cur = area = 0
        last = ev[0][0]
        for x, d in ev:
            area += cur * (x - last)
            last = x
            cur += d
        conc.append(area / max(1e-9, ev[-1][0] - ev[0][0]))
    print("   mean workgroups in flight per (se,sh,cu) key %.2f -> per CU (8 XCDs share a key) %.2f" % (np.mean(conc), np.mean(conc) / 8))
    order = np.argsort(st)
    print("   first 12 starts (us, slot, dur):", [(round(float(us(st[i])), 1), int(slot[i]), round(float(dur[i]), 1)) for i in order[:12]])
    mid = order[len(order) // 2: len(order) // 2 + 8]
    print("   mid-launch starts (us, slot, dur):", [(round(float(us(st[i])), 1), int(slot[i]), round(float(dur[i]), 1)) for i in mid])


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "mask"
    cap = 8192
    buf = torch.zeros(cap, 3, dtype=torch.int64, device="cuda")
    lib = ops.lib()
    if what == "mask":
        from sis3d.engine import ChunkEngine
        if len(sys.argv) > 2:
            os.environ["SIS3D_MASK_BRICK"] = sys.argv[2]
        net, cfg, sd = bench.build_net("detect", masks=True)
        eng = ChunkEngine(net, stage="detect", use_graph=False, mask_boxes=16)
        data = synthetic.synth_chunk(0)
        eng.load(data)
        eng.prepare(warmup=1)
        torch.cuda.synchronize()
        plan = eng.mask_plan
        scene = data.cuda().float()
        mb = net.mask_backbone
        g = mb.geometry
        pcs = [g[i]._packed.get(g[i]) for i in (2, 4, 6, 8)]
        mb.forward_planned(scene, plan)
        torch.cuda.synchronize()
        # one k3 layer alone
        C = plan.C
        def layer():
            ops.check(lib.sis3d_conv3d_k3t16_ragged(ops._ptr(plan.a), C, C, ops._ptr(pcs[0].packed_t16), ops._ptr(pcs[0].bias), C, 1,
                                                    ops._ptr(plan.b), C, ops._ptr(plan.g3t), plan.n, plan.blocks_t16, plan.brick_t16,
                                                    ops._stream()), "ragged")
        nblocks = plan.blocks_t16
        label = "mask-head k3 layer, brick %d, %d workgroups" % (plan.brick_t16, nblocks)
    else:
        brick = int(sys.argv[2]) if len(sys.argv) > 2 else 0
        x = ops.new_act(128, (24, 12, 24), "cuda")
        x.normal_()
        ops.lib()
        w = torch.randn(256, 128, 3, 3, 3, device="cuda") * 0.02
        pc = ops.PackedConv(w, torch.zeros(256, device="cuda"))
        y = ops.new_act(256, (24, 12, 24), "cuda")

        def layer():
            ops.conv3d_k3t16([x], [pc], relu=True, outs=[y], brick=brick)
        nb = {0: 16, 1: 32, 2: 64, 3: 128, 4: 54, 5: 54}[brick]
        nblocks = nb * 16
        label = "rpn_net 128->256, brick %d, %d workgroups" % (brick, nblocks)
    for _ in range(3):
        layer()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        layer()
    b.record()
    torch.cuda.synchronize()
    print("untraced: %.1f us per launch" % (a.elapsed_time(b) / 20 * 1e3))
    lib.sis3d_conv3d_k3t16_set_trace(ops._ptr(buf), cap)
    layer()
    torch.cuda.synchronize()
    lib.sis3d_conv3d_k3t16_set_trace(None, 0)
    report(buf, min(cap, nblocks), label)


if __name__ == "__main__":
    main()
