"""VERDICT r5 item 3, route (a): the 2.2x-cheaper BATCHED ENet pass (20 views of four pipelines in one pass) on a CU-MASKED stream
(hipExtStreamCreateWithCUMask -> ExternalStream), overlapped with the four pipelines' 3D graphs, so that its workgroups cannot
squat on every CU.  Step = [batched encoder for the NEXT step's views on the masked stream] || [four 3D graphs reading THIS step's
feature maps]; the two meet once per step.  Compared with the shipped form (an encoder pass per pipeline inside its own stream).
    python tools/enet_cumask_probe.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def masked_stream(n_cus, total=256, spread=True):
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * (total // 32))()
    picks = [int(i * total / n_cus) for i in range(n_cus)] if spread else list(range(n_cus))
    for b in picks:
        words[b // 32] |= (1 << (b % 32))
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(total // 32), words)
    assert rc == 0 and h.value, rc
    return torch.cuda.ExternalStream(h.value)


def main():
    import bench
    from sis3d import synthetic
    from sis3d.engine import PipelinedEngines
    from sis3d.nets.enet_hip import HipEncoder
    n = 4
    steps = 60
    # (0) the shipped form: images from RGB, an encoder pass per pipeline
    net, cfg, _ = bench.build_net("images", rgb=True)
    pe = PipelinedEngines(net, n, stage="rpn")
    for i in range(n):
        data = synthetic.synth_chunk(i)
        feats, i3d, i2d = synthetic.synth_views(i)
        pe.engines[i].load_rgb(data, synthetic.synth_images(i, cfg.NUM_IMAGES), i3d, i2d)
    pe.prepare(warmup=2)

    def timed(fn, label):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("%-78s %.3f ms per step (%.3f G voxels/s)" % (label, ms, n * bench.VOXELS / ms / 1e6), flush=True)
        return ms
    timed(pe.run, "shipped: encoder pass per pipeline inside its stream (images_rgb)")
    del pe
    torch.cuda.empty_cache()
    # (1) 3D graphs from given feature maps + ONE batched encoder graph for the 20 views
    net2, cfg2, _ = bench.build_net("images", rgb=False)
    pe = PipelinedEngines(net2, n, stage="rpn")
    for i in range(n):
        data = synthetic.synth_chunk(i)
        feats, i3d, i2d = synthetic.synth_views(i)
        pe.load(i, data, feats, i3d, i2d)
    pe.prepare(warmup=2)
    timed(pe.run, "3D graphs only (feature maps given: images)")
    enc = HipEncoder(net.image_enet_fixed, net.image_enet_trainable)
    imgs = torch.cat([synthetic.synth_images(i, cfg.NUM_IMAGES) for i in range(n)]).cuda()
    nv = cfg.NUM_IMAGES
    feat_next = [torch.zeros_like(pe.engines[i].feats_[0]) for i in range(n)]

    def encode():
        f = enc(imgs)
        for i in range(n):
            feat_next[i].copy_(f[i * nv:(i + 1) * nv])
    for cus in (0, 32, 48, 64, 96, 128):
        st = masked_stream(cus) if cus else torch.cuda.Stream()
        with torch.cuda.stream(st), torch.no_grad():
            encode()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                encode()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) / 10
        with torch.cuda.stream(st), torch.no_grad():
            e0.record()
            for _ in range(10):
                encode()
            e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / 10
        print("    (%s: graph replay %.3f ms per pass, eager launches %.3f ms per pass)" % (("mask %d CUs" % cus) if cus else "no mask", alone, eager), flush=True)
        done = torch.cuda.Event()

        def step():
            # the encoder of the NEXT step's views runs beside the 3D graphs of this step; they meet at the end of the step
            with torch.cuda.stream(st):
                g.replay()
                done.record()
            pe.run()
            for i, s in enumerate(pe.streams):
                s.wait_event(done)
                with torch.cuda.stream(s):
                    pe.engines[i].feats_[0].copy_(feat_next[i], non_blocking=True)
        timed(step, "batched encoder (%d views) on %s: alone %.3f ms; overlapped with the 3D graphs" % (
            n * nv, ("a stream masked to %d CUs" % cus) if cus else "an UNMASKED side stream", alone))


if __name__ == "__main__":
    main()
