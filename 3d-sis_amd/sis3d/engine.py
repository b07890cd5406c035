"""ChunkEngine: the per-GPU inference loop over fixed-shape voxel chunks.

One engine = one network replica + static input buffers + ONE captured HIP graph of the whole
sync-free detection pass (~60 kernel launches: convs, heads, decode, sort, NMS, RoI pooling,
classifier).  Per chunk the host does: async H2D copy of the 3.5 MB grid into the static
buffer, one graph launch, (optionally) one D2H of the fixed-size record block.  Launch-bound
Python/ctypes overhead (~10 us per op) disappears from the steady state.
"""
import threading

import torch

from . import ops
from .synthetic import CHUNK_DIMS

RECORD_WIDTH = ops.RECORD_WIDTH     # proposal box (6), rpn score, level, class id, class prob, class-regressed final box (6)


# ---- r5: one small set of HIP streams per device, by ROLE.  `torch.cuda.Stream()` hands out streams from a pool of 32 per device
# round-robin and WRAPS AROUND: a process that has asked for more than 32 (a bench that builds one engine set per configuration, a
# server that rebuilds its runners) gets the SAME underlying stream for two of its chunk pipelines, which then serialise silently
# (measured: the four-chunk share of a scene 1.5 -> 3.0 ms, profiles/r05_stream_pool.txt).  Every engine set of a device therefore
# uses the same streams: pipeline i always runs on ("pipe", i), captures warm up on ("capture", 0), and so on.  Engine sets of one
# device do not run concurrently with each other (they share the chip), so sharing their streams costs nothing, and the mapping of
# streams onto the GPU_MAX_HW_QUEUES hardware queues stays the one the first engine set got.
_STREAM_POOL = {}
_POOL_LOCK = threading.RLock()          # guards _STREAM_POOL and _DEVICE_LOCKS
_DEVICE_LOCKS = {}


def device_lock(device=None):
    """ONE lock per device for everything that warms up or captures on the device's shared ("capture", 0) / ("pipe", i) streams
    (ChunkEngine.prepare, PipelinedEngines.prepare / capture_round / calibrate): two threads preparing engines on one GPU take turns
    instead of recording into each other's graph.  Re-entrant: PipelinedEngines.prepare holds it around its engines'."""
    dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    with _POOL_LOCK:
        lk = _DEVICE_LOCKS.get(dev)
        if lk is None:
            lk = _DEVICE_LOCKS[dev] = threading.RLock()
    return lk


# ---- r6: the streams are OURS, and their hardware queues are VERIFIED.  torch.cuda.Stream() hands out streams of a
# 32-entry pool whose mapping onto HIP's hardware queues depends on everything the process created before; r5 therefore TIMED every
# window of four pool streams at every start (calibrate()).  A stream created with hipStreamCreateWithPriority is placed by HIP at
# creation, in creation order; eight of them created together land on min(8, GPU_MAX_HW_QUEUES) queues in a fixed pattern
# (profiles/r06_own_streams.txt: 8 queues -> 8 classes, 4 queues -> the classes 0 1 2 3 3 2 1 0, identical in every process), and
# four pipelines on four DISTINCT queues run at the calibrated best (0.733 ms per step on 4 and on 8 queues; any shared queue: 0.89-
# 0.93).  So: create OWN_STREAMS streams once per device, find out which share a queue (a 1-ms spin on one, a tiny kernel on the
# other: it waits iff they share -- tools/queue_identity_probe.py's test, ~40 ms once per process), and hand the pipelines streams of
# pairwise distinct queues, the null stream's queue last.  calibrate() stays as the fallback for a process that cannot get enough
# queues (and as an A/B tool).
OWN_STREAMS = 8
_OWN = {}                # device -> {"streams": [...], "klass": [...], "null_class": int}


def _create_own_streams(dev, count):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")                  # the runtime torch has loaded
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    out = []
    with torch.cuda.device(dev):
        torch.cuda.current_stream()                      # HIP initialised, device current
        for _ in range(count):
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(lo.value))      # 1 = hipStreamNonBlocking
            if rc != 0 or not h.value:
                raise ops._lib.Sis3dError("hipStreamCreateWithPriority failed (%d)" % rc)
            out.append(torch.cuda.ExternalStream(h.value, device=dev))      # never destroyed: the process's streams
    return out


def _queue_classes(dev, streams):
    """group [null stream] + streams by hardware queue: b shares a's queue iff a tiny kernel on b cannot finish while a spins.
    Timed on the DEVICE (events on both streams), so a stalled host thread cannot fake a free queue: a kernel that finished inside the
    spin was not behind it.  The opposite reading -- finished after the spin -- is also what a host stall between the two launches
    looks like, so it only counts when three trials in a row say so."""
    with torch.cuda.device(dev):
        x = torch.zeros(64, device="cuda")
        null = torch.cuda.default_stream()
        torch.cuda.synchronize()

        def spin_ms(cycles):                                   # device-side duration of the spin kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda._sleep(cycles)
            e1.record()
            torch.cuda.synchronize()
            return max(e0.elapsed_time(e1), 1e-3)
        spin_ms(1000)                                          # the first launches load their code objects: not measurements
        x.add_(1.0)
        cyc = 200000
        for _ in range(4):                                     # ~1 ms, re-measured: a spin that came out short would make every reading marginal
            ms = spin_ms(cyc)
            if 0.7 <= ms <= 1.5:
                break
            cyc = max(1000, int(cyc / ms))

        def blocked(a, b):
            for _ in range(3):
                torch.cuda.synchronize()
                sa, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                with torch.cuda.stream(a):
                    sa.record()
                    torch.cuda._sleep(cyc)
                    ea.record()
                with torch.cuda.stream(b):
                    x.add_(1.0)
                    eb.record()
                torch.cuda.synchronize()
                if sa.elapsed_time(eb) < 0.5 * sa.elapsed_time(ea):
                    return False
            return True
        allst = [null] + list(streams)
        klass, reps = [], []
        for i, st in enumerate(allst):
            for k, r in enumerate(reps):
                if blocked(allst[r], st):
                    klass.append(k)
                    break
            else:
                klass.append(len(reps))
                reps.append(i)
    return klass[1:], klass[0]


def own_streams(device=None):
    """this device's own streams + the hardware-queue class of each (and of the null stream); created and probed on first use"""
    dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    with _POOL_LOCK:
        o = _OWN.get(dev)
        if o is None:
            with device_lock(dev):
                st = _create_own_streams(dev, OWN_STREAMS)
                for _ in range(2):                             # fewer classes than pipelines: probe once more before believing it
                    klass, null_class = _queue_classes(dev, st)
                    if len(set(klass)) >= 4:
                        break
            o = _OWN[dev] = {"streams": st, "klass": klass, "null_class": null_class}
        return o


def distinct_queue_streams(n, device=None):
    """n of the device's own streams on pairwise DISTINCT hardware queues (the null stream's queue is taken last), or as many as there
    are -> (streams, verified: bool)"""
    o = own_streams(device)
    order = sorted(range(len(o["streams"])), key=lambda i: (o["klass"][i] == o["null_class"], i))
    seen, pick = set(), []
    for i in order:
        if o["klass"][i] not in seen:
            seen.add(o["klass"][i])
            pick.append(i)
        if len(pick) == n:
            break
    if len(pick) < n:                                          # not enough queues: fill up in creation order (they will share)
        pick += [i for i in range(len(o["streams"])) if i not in pick][:n - len(pick)]
    return [o["streams"][i] for i in pick[:n]], len(seen) >= n


def candidate_streams(count=16, device=None):
    """the device's own streams in creation order: what calibrate() (the fallback) chooses windows from"""
    return list(own_streams(device)["streams"])[:int(count)]


def pooled_stream(role, index=0, device=None):
    """streams by ROLE, one set per device: ("pipe", i) = the i-th of the distinct-queue selection; the side roles ("capture", "round",
    "merge", "copy") take own streams from the END of the list (the queues the pipelines are least likely to sit on)"""
    dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    key = (dev, role, int(index))
    with _POOL_LOCK:
        st = _STREAM_POOL.get(key)
        if st is None:
            o = own_streams(dev)
            if role == "pipe":
                st = distinct_queue_streams(int(index) + 1, dev)[0][int(index)]
            else:
                side = [k for k in _STREAM_POOL if k[0] == dev and k[1] != "pipe"]
                st = o["streams"][-1 - (len(side) % len(o["streams"]))]
            _STREAM_POOL[key] = st
        return st


class ChunkEngine:
    def __init__(self, net, dims=CHUNK_DIMS, stage="detect", use_graph=True, n_views=0, device=None, from_depth=False, group=1,
                 mask_boxes=0, shared_chip=False, brick_cap=0, mailbox=False, mail_input="grid", truncated=3.0, mail_ring=256,
                 stage_ahead=None):
        """stage: 'backbone' (the two pyramid levels only), 'rpn' (backbone + RPN maps, BASELINE config 1) or 'detect'
        (+ proposals, RoI pooling, classifier).
        from_depth (USE_IMAGES): the chunk's views arrive as depth maps + poses (the dataloader's
        blobs['nearest_images'], lib/datasets/dataloader.py:17-38) and the voxel->pixel lists are computed inside the
        captured graph (sis3d_compute_projection) instead of being loaded; `view_counts()` reports views that saw
        nothing -- the caller's cue to take the reference's killing_inds route (layer_utils.projection.prepare_projection).
        group (1 or 2): chunks per captured graph.  With 2, the four 12-GFLOP RPN convs of the pair go out as one batched
        launch (Network.backbone_rpn_group); slots are addressed by the `slot` argument of load / set_origin and `run()`
        returns a list of per-chunk outputs.
        mask_boxes (detect stage, USE_MASK nets): BASELINE config 3 in full -- the mask head runs inside the captured graph on
        a FIXED detection set: with seeded weights no class probability passes CLASS_THRESH, so the first `mask_boxes`
        post-NMS RoIs of the chunk loaded at prepare() time stand in as detections; their crop
        windows (box rounded half-to-even, clipped, non-degenerate: trainval.py:702-712,742-745) are read back ONCE before
        the capture.  `mask_stats()` reports the crop volumes and the mask-head FLOPs.
        shared_chip / brick_cap: the engine's DISPATCH REGIME (ops.dispatch_regime): every launch this engine makes -- warm-up, capture
        and eager passes alike -- is dispatched for a chip shared with other chunks' kernels / with the direct k3 kernel's brick capped.
        An attribute of the engine, passed to the library per call (thread-local on the Python side): eager passes of engines of
        different regimes may run concurrently from different threads.  prepare() itself -- warm-up and capture on the device's
        shared capture stream -- is serialised per device (engine.device_lock): concurrent prepare() calls are safe, not parallel."""
        self.shared_chip, self.brick_cap = bool(shared_chip), int(brick_cap)
        # mailbox (r5): what changes per chunk -- the source of the input grid (pinned host or device memory), the chunk origin, the
        # row that receives the record block -- reaches the captured graph through a ring of slots in pinned host memory
        # (ops.Mailbox), read by kernels INSIDE the graph: `submit()` is CPU stores + ONE graph launch.  mail_input 'sdf': the source is
        # the raw SDF block of a .chunk file, TSDF-encoded by the graph's second node (lib/datasets/dataset.py:54-70).
        # stage_ahead (default on; SIS3D_MAIL_STAGE_AHEAD=0 / stage_ahead=False: off): a slot may also name the chunk of the pass AFTER
        # its own (submit(next_src=)); one more row of workgroups of the pass's longest conv launch (the rpn_net pair,
        # ops.piggyback / sis3d_conv3d_k3wino_piggyback) pulls it across PCIe into `_ahead` while the pass computes, and the next pass's
        # upload node copies it from there at HBM speed instead of waiting on the link.  (Measured alternatives, profiles/
        # r05_stage_ahead.txt: the same copy as a parallel BRANCH of the captured graph makes every replay 2.4x slower.)
        self.mail = None
        self.mail_input, self._trunc = mail_input, float(truncated)
        self._slot = None
        self._ahead, self._carry_ahead, self.piggybacked = None, True, False
        self.mask_boxes = int(mask_boxes)
        self.mask_plan = None
        self.net, self.dims, self.stage, self.use_graph = net, tuple(dims), stage, use_graph
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.group = int(group)
        cfg = net.cfg
        G = self.group
        self.scenes = [torch.zeros((1, 2) + self.dims, device=self.device) for _ in range(G)]
        self.use_images = bool(cfg.USE_IMAGES)
        self.from_depth = False
        self.rgb = False
        if self.use_images:
            nvox = self.dims[0] * self.dims[1] * self.dims[2]
            h, w = cfg.DEPTH_SHAPE[1], cfg.DEPTH_SHAPE[0]
            self.n_views = n_views or cfg.NUM_IMAGES
            self.feats_ = [torch.zeros(self.n_views, cfg.NUM_IMAGE_CHANNELS, h, w, device=self.device) for _ in range(G)]
            self.i3d_ = [torch.zeros(self.n_views, nvox + 1, dtype=torch.int64, device=self.device) for _ in range(G)]
            self.i2d_ = [torch.zeros(self.n_views, nvox + 1, dtype=torch.int64, device=self.device) for _ in range(G)]
            # RGB input (USE_IMAGES_GT=False): the 2D encoder (csrc/enet.hip, one launch per bottleneck) runs inside the step, in its own captured
            # graph in front of the 3D graph (falls back to eager launches if the capture of those library calls is refused)
            self.rgb = bool(cfg.USE_IMAGES and not cfg.USE_IMAGES_GT)
            if self.rgb:
                iw, ih = cfg.IMAGE_SHAPE
                self.images_ = [torch.zeros(self.n_views, 3, ih, iw, device=self.device) for _ in range(G)]
                self.enet_graph = None
            self.from_depth = bool(from_depth)
            if self.from_depth:
                from .layer_utils.projection import ProjectionHelper
                self.helper = ProjectionHelper(cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE,
                                               list(self.dims), cfg.VOXEL_SIZE)
                self.depths_ = [torch.zeros(self.n_views, h, w, device=self.device) for _ in range(G)]
                self.view_params_ = [torch.zeros(self.n_views, ops.VIEW_PARAM_FLOATS, device=self.device) for _ in range(G)]
        # chunk origin (x,y,z) in scene voxels, added to the boxes of the record block
        self.origins = [torch.zeros(3, device=self.device) for _ in range(G)]
        self.graph = None
        self.out = None
        self.records = None
        if mailbox:
            if G != 1 or mail_input not in ("grid", "sdf"):
                raise ops._lib.Sis3dError("mailbox engines: one chunk per graph, mail_input 'grid' or 'sdf'")
            self.mail = ops.Mailbox(self.device, ring=mail_ring)
            if mail_input == "sdf":
                self._sdf_stage = torch.zeros(self.dims[0] * self.dims[1] * self.dims[2], device=self.device)
            if stage_ahead is None:
                import os
                stage_ahead = os.environ.get("SIS3D_MAIL_STAGE_AHEAD", "1") not in ("0", "")
            if stage_ahead:
                self._ahead = torch.zeros_like(self._sdf_stage if mail_input == "sdf" else self.scenes[0].view(-1))

    # slot-0 views of the static buffers (the single-chunk API)
    scene = property(lambda self: self.scenes[0])
    feats = property(lambda self: self.feats_[0])
    i3d = property(lambda self: self.i3d_[0])
    i2d = property(lambda self: self.i2d_[0])
    origin = property(lambda self: self.origins[0])

    def _imageft(self, g):
        if not self.use_images:
            return None
        if self.from_depth:
            h = self.helper
            ops.compute_projection(self.depths_[g], self.view_params_[g], self.dims, h.image_dims, h.intrinsic, h.depth_min,
                                   h.depth_max, h.voxel_size, out=(self.i3d_[g], self.i2d_[g]))
        project = ops.project_views_prepare if getattr(self.net, "fuse_projection", False) else ops.project_views_max
        return project(self.feats_[g], self.i3d_[g], self.i2d_[g], self.dims, ())

    def _finish(self, d, g):
        """pack one chunk's detections: records + the fixed-size block in scene coordinates (one kernel)"""
        d["records"], d["block"] = ops.pack_records(d, self.dims, self.origins[g], mail=self.mail if self.mask_plan is None else None)
        return d

    def _mail_commit(self):
        src, dst, origin, nxt = self._slot or (None, None, None, None)
        self.mail.write(src, dst, origin, nxt if self.piggybacked else None)
        self._slot = None

    def _step(self):
        if self.mail is not None and not torch.cuda.is_current_stream_capturing():
            self._mail_commit()                      # every EXECUTED pass consumes one slot (a capture records the kernels, it runs none)
        with ops.dispatch_regime(self.shared_chip, self.brick_cap):
            if self.mail is not None:
                if self.mail_input == "sdf":
                    ops.mail_upload(self.mail, self._sdf_stage, self.origins[0], staged=self._ahead)
                    ops.tsdf_encode(self._sdf_stage, self.dims, self._trunc, "abs", None, out=self.scenes[0])
                else:
                    ops.mail_upload(self.mail, self.scenes[0].view(-1), self.origins[0], staged=self._ahead)
            # the pass's longest conv launch also pulls the NEXT chunk across the link (not inside the round graph: its slots announce
            # no next chunk); piggybacked: did a launch of this pass / of the captured graph take the upload on board
            with ops.piggyback(self.mail, self._ahead if self._carry_ahead else None) as took:
                out = self._step_body()
            if self._carry_ahead:
                self.piggybacked = bool(took[0])
            if self.mail is not None and not (isinstance(out, dict) and out.pop("_mail_posted", False)):
                ops.mail_post(self.mail, out["block"] if isinstance(out, dict) and "block" in out else None)
            return out

    def submit(self, src=None, block_dst=None, origin=None, next_src=None):
        """mailbox engines: one pass on the chunk `src` points at (pinned host or device tensor; None = the static input as it is),
        record block to `block_dst`, boxes shifted by `origin`: CPU stores into the next mailbox slot + one graph launch.
        next_src: the pinned host chunk this engine will be given NEXT, if the caller knows it -- pulled across the link while this pass
        computes (stage_ahead); it must stay unchanged until that pass has run.  Ignored when it is not pinned host memory."""
        if self.mail is None:
            raise ops._lib.Sis3dError("submit() needs an engine built with mailbox=True")
        # the upload kernel reads `n` floats from the raw pointer: a wrong-sized, strided or non-fp32 tensor would be an out-of-bounds
        # read, not an exception
        want = self._sdf_stage.numel() if self.mail_input == "sdf" else self.scenes[0].numel()
        if src is not None and ops.mail_source(src, want) is None:
            raise ops._lib.Sis3dError("submit(): src must be a contiguous float32 tensor of %d elements on the device or in pinned host "
                                      "memory (got %s)" % (want, "%s %s%s" % (tuple(src.shape), src.dtype, "" if src.is_contiguous() else " strided")
                                                           if isinstance(src, torch.Tensor) else type(src).__name__))
        if src is None and self.mail_input == "sdf":
            raise ops._lib.Sis3dError("submit(): an engine with mail_input='sdf' encodes its input from the slot's source on every pass; "
                                      "src=None would re-encode the previous chunk's SDF block over the input")
        if next_src is not None and (self._ahead is None or next_src.is_cuda or ops.mail_source(next_src, self._ahead.numel()) is None):
            next_src = None
        self._slot = (src, block_dst, origin, next_src)
        return self.run()

    def _step_body(self):
        net = self.net
        if self.rgb and (self.graph is None and not torch.cuda.is_current_stream_capturing()):
            self._encode_views()
        if self.group == 1:
            imageft = self._imageft(0)
            if self.stage == "backbone":
                # the backbone proper (geometry1 [+ color] + geometry2): what BASELINE.json's roofline target names
                l1, l2 = net.backbone_only(self.scenes[0], imageft)
                return {"level1": l1, "level2": l2}
            if self.stage == "rpn":
                net.backbone_rpn(self.scenes[0], imageft)
                return {k: v for k, v in net._predictions.items() if k.startswith("rpn_")}
            d = self._finish(net.detect(self.scenes[0], imageft), 0)
            if self.mask_plan is not None:
                d["mask_pred"] = net.mask_backbone.forward_planned(self.scenes[0], self.mask_plan)
            return d
        fts = [self._imageft(g) for g in range(self.group)] if self.use_images else None
        if self.stage == "rpn":
            return [{k: v for k, v in pred.items() if k.startswith("rpn_")}
                    for _, _, _, pred in net.backbone_rpn_group(self.scenes, fts)]
        return [self._finish(d, g) for g, d in enumerate(net.detect_group(self.scenes, fts))]

    def prepare(self, warmup=2):
        """warm caches (weight repack, anchor tables) and capture the graph"""
        with device_lock(self.device), torch.no_grad():
            self.net.eval()
            for _ in range(max(1, warmup)):
                self.out = self._step()
            torch.cuda.synchronize()
            if self.rgb and self.use_graph:
                try:
                    side = pooled_stream("capture", 0, self.device)
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._encode_views()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._encode_views()
                    self.enet_graph = g
                except Exception:                                   # library calls that cannot be captured: keep them eager
                    self.enet_graph = None
                torch.cuda.synchronize()
            if self.mask_boxes > 0 and self.stage == "detect" and self.group == 1:
                self.mask_plan = self._plan_masks()
                self.out = self._step()
                torch.cuda.synchronize()
            if self.use_graph:
                side = pooled_stream("capture", 0, self.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.out = self._step()          # one more eager pass on the capture stream
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = self._step()
                torch.cuda.synchronize()
        return self

    def _encode_views(self):
        """RGB views -> ENet feature maps, written into the static feature buffers the 3D graph reads"""
        with ops.dispatch_regime(self.shared_chip, self.brick_cap):
            for g in range(self.group):
                self.feats_[g].copy_(self.net.image_features(self.images_[g]))

    def load_rgb(self, data, images, i3d, i2d, slot=0):
        """rgb engines: grid, RGB views (V,3,H,W), packed index lists"""
        self._copy(self.scenes[slot], data)
        self._copy(self.images_[slot], images)
        self._copy(self.i3d_[slot], i3d)
        self._copy(self.i2d_[slot], i2d)

    def _plan_masks(self):
        """crop windows of the stand-in detections of the chunk currently loaded (one D2H read, before the capture)"""
        n = int(self.out["num"].item())
        rois = self.out["rois"][:min(n, self.mask_boxes)].detach().cpu()
        wins = []
        for r in rois.tolist():
            w = [int(round(v)) for v in r]
            w = [max(0, min(w[k], self.dims[k % 3])) for k in range(6)]
            if w[0] < w[3] and w[1] < w[4] and w[2] < w[5]:
                wins.append(tuple(w))
        if not wins:
            raise ops._lib.Sis3dError("mask_boxes: no non-degenerate RoI on this chunk")
        return self.net.mask_backbone.plan(wins, self.device)

    def mask_stats(self):
        p = self.mask_plan
        if p is None:
            return {}
        vols = sorted(dx * dy * dz for dx, dy, dz in p.dims)
        return {"mask_boxes": p.n, "mask_voxels": p.voxels, "mask_head_gflop": p.flops / 1e9,
                "mask_crop_voxels_min_median_max": [vols[0], vols[len(vols) // 2], vols[-1]]}

    @staticmethod
    def _copy(dst, src):
        # async only where it is safe: device sources or pinned host memory (a pageable source may be recycled by
        # the caller while an "async" staged copy is still reading it)
        dst.copy_(src, non_blocking=bool(src.is_cuda or src.is_pinned()))

    def set_origin(self, origin, slot=0):
        """chunk origin in scene voxels (host tuple): written into the static buffer the captured graph reads"""
        self.origins[slot].copy_(torch.tensor([float(origin[0]), float(origin[1]), float(origin[2])]))

    def load(self, data, feats=None, i3d=None, i2d=None, slot=0):
        self._copy(self.scenes[slot], data)
        if self.use_images:
            self._copy(self.feats_[slot], feats)
            self._copy(self.i3d_[slot], i3d)
            self._copy(self.i2d_[slot], i2d)

    def load_views(self, data, feats, depths, poses, world2grid, slot=0):
        """from_depth engines: grid, feature maps, depth maps (V,h,w), camera_to_world and world_to_grid (V,4,4)"""
        self._copy(self.scenes[slot], data)
        self._copy(self.feats_[slot], feats)
        self._copy(self.depths_[slot], depths)
        rows = torch.stack([self.helper.view_params(poses[v], world2grid[v]) for v in range(self.n_views)])
        self.view_params_[slot].copy_(rows)               # 40 floats per view of host geometry; blocking (pageable source)

    def view_counts(self, slot=0):
        """visible voxels per view of the last pass (host list; synchronises)"""
        return self.i3d_[slot][:, 0].cpu().tolist()

    def run(self):
        """one pass over the chunk(s) currently in the static buffers; returns the (static) output dict (list of dicts
        for group > 1)"""
        if self.mail is not None and self.mail_input == "sdf" and self.graph is not None and (self._slot is None or self._slot[0] is None):
            # an sdf-mode mailbox engine encodes its input from the slot's source on EVERY pass; load() + run() would
            # re-encode the previous chunk's staged SDF block over what load() wrote
            raise ops._lib.Sis3dError("run(): an engine with mail_input='sdf' takes its chunk through submit(src=...) / run_fed(); "
                                      "load() + run() is not a path it has")
        with torch.no_grad():
            if self.rgb and self.graph is not None:
                if self.enet_graph is not None:
                    self.enet_graph.replay()
                else:
                    self._encode_views()
            if self.graph is not None:
                if self.mail is not None:
                    self._mail_commit()
                self.graph.replay()
            else:
                self.out = self._step()
        return self.out


def hw_queues():
    """hardware queues HIP gives this process (GPU_MAX_HW_QUEUES, read by the runtime when it initialises; HIP's default is 4)"""
    import os
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


_QUEUE_WARNED = set()


def check_hw_queues(n_pipelines, strict=None, verified=None):
    """n pipelines want n DISTINCT hardware queues.  verified (r6: what engine.distinct_queue_streams found by probing the device's own
    streams): True -> nothing to check, whatever GPU_MAX_HW_QUEUES says (four own streams sit on four queues of HIP's default four);
    False -> the process cannot give the pipelines a queue each: two of them will share one and serialise (measured 0.89-0.93 ms per
    step instead of 0.73) -- RAISE with the fix spelled out when strict / SIS3D_STRICT_HW_QUEUES=1, else warn once per count and let
    PipelinedEngines.prepare() fall back to its timing calibration.  verified=None (no probe result: a caller that only wants the
    rule) falls back to the count of queues the environment asks HIP for.  `import sis3d` no longer edits os.environ (r6)."""
    import os
    import warnings
    have = hw_queues()
    if n_pipelines < 2 or verified is True:
        return True
    if verified is None and have >= (int(n_pipelines) + 2 if n_pipelines >= 4 else 4):
        return True
    msg = ("sis3d: %d chunk pipelines need %d distinct hardware queues and this process cannot give them one each (GPU_MAX_HW_QUEUES=%d; "
           "HIP's default is 4): pipelines will share a queue and serialise.  Fix: export GPU_MAX_HW_QUEUES=8 in the environment of the "
           "process BEFORE it starts (the HIP runtime reads it once, when it initialises), or use <= %d pipelines"
           % (n_pipelines, n_pipelines, have, max(1, min(have, n_pipelines) - 1)))
    if strict or (strict is None and os.environ.get("SIS3D_STRICT_HW_QUEUES", "0") not in ("", "0")):
        raise ops._lib.Sis3dError(msg)
    if n_pipelines not in _QUEUE_WARNED:
        _QUEUE_WARNED.add(n_pipelines)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return False


def default_pipelines():
    """chunk pipelines per GPU that measured best: four on >= 6 hardware queues, three on HIP's default of four"""
    return 4 if hw_queues() >= 6 else 3


class PipelinedEngines:
    """N ChunkEngines on N HIP streams sharing one set of weights: independent chunks in flight concurrently.

    The layers of this network have only 216..1728 output tiles, so no single kernel fills the 256 CUs evenly
    (a 432-workgroup launch leaves 80 CUs one workgroup short) and every dependent kernel boundary drains the
    chip.  Captured graphs replaying on separate streams fill each other's gaps: backbone+RPN 0.60 ms per chunk
    alone, 0.466 with two, 0.449 with three in flight; the detect pass (long single-workgroup tail kernels) gains
    11 % from the third stream; a fourth loses again.
    r4: "a fourth loses" was HIP's default of 4 hardware queues -- the pipelines' streams, the capture streams and the null stream share
    them round-robin, so a fourth pipeline serialises behind another one.  With GPU_MAX_HW_QUEUES >= 6 in the environment (read when the
    HIP runtime initialises; bench.py sets 8) four pipelines are the best count for every workload (backbone + RPN 2.16 -> 2.28 G voxels/s,
    detect 1.86 -> 2.02 G; five and six lose on any queue count: profiles/r04_hw_queues.txt)."""

    def __init__(self, net, n=2, brick_cap=None, **kw):
        """brick_cap: cap (voxels) on the k3 kernel's brick for the graphs captured here; default 108 when n >= 2 (small
        bricks = several workgroups per CU, so the pipelines' kernels interleave), none for a single pipeline"""
        cap = (108 if n >= 2 else 0) if brick_cap is None else int(brick_cap)
        self._brick_cap = cap
        self.streams, self.placement_verified = distinct_queue_streams(n)
        check_hw_queues(n, verified=self.placement_verified)
        self.stream_window, self.stream_window_times = 0, {}
        self.engines = []
        for s in self.streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                # the dispatch regime of pipelines that share the chip is a property of their engines (r5: per-call arguments of the
                # library, not a process-wide switch toggled around the captures)
                self.engines.append(ChunkEngine(net, shared_chip=(n >= 2), brick_cap=cap, **kw))

    def prepare(self, warmup=2, calibrate=None):
        """warm up and capture every pipeline, then (calibrate: default on for >= 2 captured pipelines, SIS3D_NO_AUTO_CALIBRATE=1 turns
        it off) choose the pipelines' streams by timing one pass of all pipelines on every window of the candidate streams -- ~40
        replays, tens of milliseconds; a caller with a more specific workload calls `calibrate(run_once)` again"""
        with device_lock(self.engines[0].device):
            for e, s in zip(self.engines, self.streams):
                with torch.cuda.stream(s):
                    e.prepare(warmup)
            torch.cuda.synchronize()
            if calibrate is None:
                # r6: pipelines on verified distinct hardware queues need no timing pass; the fallback is for a process that could
                # not get them (SIS3D_AUTO_CALIBRATE=1 forces it)
                import os
                calibrate = (not self.placement_verified and os.environ.get("SIS3D_NO_AUTO_CALIBRATE", "0") in ("", "0")) \
                    or os.environ.get("SIS3D_AUTO_CALIBRATE", "0") not in ("", "0")
            sdf_fed = any(e.mail is not None and e.mail_input == "sdf" for e in self.engines)      # no pass without a source: the caller calibrates on its feed
            if calibrate and len(self.engines) >= 2 and all(e.graph is not None for e in self.engines) and not sdf_fed:
                self.calibrate(self.run, reps=3, warm=1)
        return self

    def load(self, i, *a, wait=True, **kw):
        """copy a chunk (CPU or GPU tensors) into pipeline i's static buffers, on pipeline i's stream.  wait=False: the caller has
        already ordered the pipeline's stream behind the producer of the inputs (SceneRunner does it once per scene)"""
        s = self.streams[i]
        if wait:
            s.wait_stream(torch.cuda.current_stream())     # GPU inputs may still be in flight on the caller's stream
        with torch.cuda.stream(s):
            self.engines[i].load(*a, **kw)
        for t in a:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(s)                         # keep the allocator from recycling them under the copy

    # ---- streamed inputs (r5).  The reference's forward owns the upload (`blobs['data'].cuda()`, lib/nets/network.py:191): a chunk
    # arrives in HOST memory.  feed(i, host) enqueues its H2D copy on a dedicated copy stream into one of pipeline i's two staging
    # buffers; run_fed(i) makes pipeline i's stream wait for that copy, moves the grid into the static buffer its captured graph
    # reads (a 3.5 MB device copy, or sis3d_tsdf_encode when the host hands over the raw 1.77 MB SDF block of a .chunk file:
    # lib/datasets/dataset.py:54-70), releases the staging buffer and replays the graph.  Fed one chunk ahead, the upload of chunk
    # k + 1 runs under the compute of chunk k; nothing on the host waits.
    @property
    def mailbox(self):
        return self.engines[0].mail is not None

    def enable_feed(self, mode="grid", truncated=3.0, copy="kernel"):
        """mode 'grid': hosts hand over the encoded (1,2,X,Y,Z) float32 grid, as the reference's dataloader does; 'sdf': the raw
        float32 SDF block in file order (x fastest), encoded on the device.  Host tensors must be pinned.
        copy: how / where the upload is enqueued --
          'kernel'        (default) by a KERNEL on the pipeline's own stream that reads the pinned host memory across PCIe
                          (sis3d_upload_f32: 8 workgroups, 256 KB in flight -- grids go straight into the graph's static input, an SDF
                          block into a staging buffer sis3d_tsdf_encode reads).  An ordinary launch: it can be enqueued one chunk
                          ahead, right behind the previous replay, without ever blocking the host;
          'own'           on the pipeline's own stream, straight into the buffer the device copy / encode reads: no second stream, no
                          event handshake; the upload of chunk k + 1 sits behind the replay of chunk k in stream order, i.e. a
                          pipeline pauses for its own upload (71 us for 3.54 MB at the measured 50 GB/s) while the other pipelines
                          keep the chip busy;
          'per_pipeline'  on a copy stream of the pipeline's own, double-buffered: the upload of chunk k + 1 runs under chunk k
                          (nine live streams on eight hardware queues: measured slower, kept as a tested A/B).
        (ONE copy stream shared by all pipelines was measured too -- its event handshakes serialise the pipelines, 0.77 -> 1.13 ms per
        step before a single byte is copied, profiles/r05_stream_probe.txt -- and is not offered.)
        Measured (profiles/r05_stream_probe.txt, four backbone + RPN pipelines, ms per step): resident 0.770, own 0.831,
        per_pipeline 0.829, shared 1.38."""
        if mode not in ("grid", "sdf"):
            raise ValueError("feed mode must be 'grid' or 'sdf'")
        if self.mailbox:
            # engines built with mailbox=True take their chunks through the graph's own upload node (ChunkEngine.submit): nothing
            # to set up, no command per chunk besides the graph launch; the engines' mail_input fixes the format
            if any(e.mail_input != mode for e in self.engines):
                raise ops._lib.Sis3dError("these engines were captured for mail_input=%r" % self.engines[0].mail_input)
            self._feed_mode, self._feed_copy = mode, "mailbox"
            self._mail_q = [[] for _ in self.engines]
            return self
        if copy not in ("kernel", "own", "per_pipeline"):
            raise ValueError("copy must be 'kernel', 'own' or 'per_pipeline'")
        if any(e.group != 1 or e.use_images for e in self.engines):
            raise ops._lib.Sis3dError("streamed inputs: geometry-only engines of one chunk per graph")
        torch.cuda.synchronize()
        self._feed_mode, self._feed_trunc, self._feed_copy = mode, float(truncated), copy
        self._feed = []
        for i, e in enumerate(self.engines):
            X, Y, Z = e.dims
            shape = (1, 2, X, Y, Z) if mode == "grid" else (X * Y * Z,)
            # grids are uploaded straight into the graph's static input (no staging at all); an SDF block goes to a staging buffer the
            # encode kernel reads (an encode that reads the host block itself puts a link-bound wave on every CU: measured slower)
            direct = mode == "grid" and copy in ("kernel", "own")
            nbuf = 1 if copy in ("own", "kernel") else 2
            cs = self.streams[i] if copy in ("own", "kernel") else pooled_stream("copy", i)
            self._feed.append({"stage": [e.scenes[0]] if direct else [torch.empty(shape, device=e.device) for _ in range(nbuf)],
                               "direct": direct, "nbuf": nbuf, "stream": cs,
                               "ready": [torch.cuda.Event() for _ in range(nbuf)], "free": [torch.cuda.Event() for _ in range(nbuf)],
                               "tag": [None] * nbuf, "used": [False] * nbuf, "fed": 0, "run": 0})
        return self

    def feed(self, i, host):
        """enqueue the upload of one chunk for pipeline i -> False if the pipeline's staging buffers all hold chunks that have not
        been consumed yet (nothing is enqueued then)"""
        if self.mailbox:
            if host.is_cuda or not host.is_pinned():
                raise ops._lib.Sis3dError("feed: the chunk must sit in pinned host memory")
            self._mail_q[i].append(host)                # no GPU work here: the graph's upload node reads it when the pass runs
            return True
        f = self._feed[i]
        if f["fed"] - f["run"] >= f["nbuf"]:
            return False
        if host.is_cuda or not host.is_pinned():
            raise ops._lib.Sis3dError("feed: the chunk must sit in pinned host memory (an upload from pageable memory blocks the host)")
        s = f["fed"] % f["nbuf"]
        cs = f["stream"]
        own = self._feed_copy in ("own", "kernel")
        if not own and f["used"][s]:
            cs.wait_event(f["free"][s])                   # the chunk this buffer held has been moved into the static buffer
        with torch.cuda.stream(cs), torch.no_grad():
            if self._feed_copy == "kernel":
                ops.upload(host, f["stage"][s])
            else:
                f["stage"][s].copy_(host.view(f["stage"][s].shape), non_blocking=True)
            if not own:
                f["ready"][s].record(cs)
        f["tag"][s], f["used"][s] = host.data_ptr(), True
        f["fed"] += 1
        return True

    def is_fed(self, i, host):
        """is an upload of THIS host tensor outstanding on pipeline i (fed, not yet consumed)?"""
        if self.mailbox:
            return any(h.data_ptr() == host.data_ptr() for h in self._mail_q[i])
        f = self._feed[i]
        return any(f["tag"][k % f["nbuf"]] == host.data_ptr() for k in range(f["run"], f["fed"]))

    def pending(self, i):
        """chunks fed to pipeline i and not yet consumed"""
        if self.mailbox:
            return len(self._mail_q[i])
        f = self._feed[i]
        return f["fed"] - f["run"]

    def consume(self, i, stream):
        """on `stream`: wait for the upload of the next chunk pipeline i was fed, move it into the pipeline's static input buffer
        (device copy, or the TSDF encoding of a raw SDF block) and hand the staging buffer back to the copy stream"""
        f = self._feed[i]
        if f["fed"] == f["run"]:
            raise ops._lib.Sis3dError("streamed inputs: nothing was fed to pipeline %d" % i)
        s = f["run"] % f["nbuf"]
        e = self.engines[i]
        own = self._feed_copy in ("own", "kernel")
        with torch.cuda.stream(stream), torch.no_grad():
            if own:
                if stream is not f["stream"]:
                    stream.wait_stream(f["stream"])       # (the one-launch round consumes on its capture stream)
            else:
                stream.wait_event(f["ready"][s])
            if f["direct"]:
                pass                                      # the upload went straight into the static input
            elif self._feed_mode == "grid":
                e.scenes[0].copy_(f["stage"][s], non_blocking=True)
            else:
                ops.tsdf_encode(f["stage"][s], e.dims, self._feed_trunc, "abs", None, out=e.scenes[0])
            if not own:
                f["free"][s].record(stream)
        f["run"] += 1

    def run_fed(self, i, host=None):
        """run pipeline i on the next chunk it was fed (host given and nothing outstanding: fed now) -> the static output dict"""
        if host is not None and self.pending(i) == 0:
            self.feed(i, host)
        if self.mailbox:
            if not self._mail_q[i]:
                raise ops._lib.Sis3dError("streamed inputs: nothing was fed to pipeline %d" % i)
            with torch.cuda.stream(self.streams[i]):
                src = self._mail_q[i].pop(0)
                return self.engines[i].submit(src=src, next_src=self._mail_q[i][0] if self._mail_q[i] else None)
        st = self.streams[i]
        self.consume(i, st)
        with torch.cuda.stream(st):
            return self.engines[i].run()

    def run(self, i=None):
        """replay engine i (or all of them) on its own stream; returns the static output dict(s)"""
        idx = range(len(self.engines)) if i is None else [i]
        outs = []
        for k in idx:
            with torch.cuda.stream(self.streams[k]):
                outs.append(self.engines[k].run())
        return outs if i is None else outs[0]

    def capture_round(self, send, stagger=True):
        """ONE HIP graph = one pass of EVERY pipeline: the capture stream forks into the pipelines' streams, pipeline e runs its
        detection pass on the chunk in its static buffers and copies its record block into row e of `send`, and all of them join
        the capture stream again.  A rank that owns as many chunks of a scene as it has pipelines (4 at N = 8 on the 32-chunk
        scene) then issues ONE graph launch per scene: the chunks start together instead of one host launch (~30-40 us of
        hipGraphLaunch + copies) apart, and whatever follows on the launch stream (the collective, the whole-scene merge) is
        ordered behind all of them without a host-side join.  -> (graph, capture stream)"""
        if any(eng.group != 1 for eng in self.engines):
            raise ops._lib.Sis3dError("capture_round: engines of one chunk per graph only (group == 1)")
        main = pooled_stream("round", 0)
        main.wait_stream(torch.cuda.current_stream())
        with device_lock(self.engines[0].device), torch.no_grad():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                prev_ev = None
                for e, (eng, s) in enumerate(zip(self.engines, self.streams)):
                    s.wait_stream(main)
                    with torch.cuda.stream(s):
                        if prev_ev is not None and stagger:
                            # pipelines that start together run the same layers in lockstep and queue for the same CUs (the
                            # Winograd launches take a CU whole); started one stage apart they fill each other's gaps, as the
                            # free-running per-chunk replays do: pipeline e starts when pipeline e - 1 has finished level 1
                            s.wait_event(prev_ev)
                        ev = torch.cuda.Event()
                        eng.net._after_level1 = ev.record
                        eng._carry_ahead = False
                        try:
                            out = eng._step()
                        finally:
                            eng.net._after_level1 = None
                            eng._carry_ahead = True
                        prev_ev = ev
                        send[e].copy_(out["block"])
                for s in self.streams:
                    main.wait_stream(s)
        torch.cuda.synchronize()
        return g, main

    def calibrate(self, run_once, reps=3, warm=1, count=16, windows=None):
        """Choose the HIP streams the pipelines run on BY MEASUREMENT (r5).  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
        when they are created and does not say how; which queues four pipelines land on -- relative to each other, to the stream
        that issues the serial part of the work, and to the hardware pipes behind the queues -- moves a 32-chunk scene between 7.5 and
        12.5 ms and four backbone + RPN pipelines between 0.77 and 1.19 ms per step on the same graphs (profiles/r05_queue_map.txt:
        every window of four consecutive pool streams, same process).  The captured graphs replay on any stream, so: for every window
        of len(engines) consecutive candidate streams run `run_once()` (one pass of the caller's real workload over these pipelines,
        enqueued, not synchronised) warm + reps times, time the reps, keep the fastest window.  One-time cost: windows x (warm + reps)
        passes.  -> (best window index, {window index: ms per pass})"""
        import time
        n = len(self.engines)
        cands = candidate_streams(count, self.engines[0].device)
        if len(cands) < n + 1:
            return 0, {}
        wins = list(range(0, len(cands) - n + 1)) if windows is None else [w for w in windows if w + n <= len(cands)]
        times = {}
        for w in wins:
            self.streams = cands[w:w + n]
            for _ in range(warm):
                run_once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run_once()
            torch.cuda.synchronize()
            times[w] = (time.perf_counter() - t0) / reps * 1e3
        best = min(times, key=times.get)
        self.streams = cands[best:best + n]
        self.stream_window = best
        self.stream_window_times = times
        if hasattr(self, "_feed") and getattr(self, "_feed_copy", None) in ("own", "kernel"):
            for i, f in enumerate(self._feed):
                f["stream"] = self.streams[i]
        return best, times

    def join(self):
        """make the current stream wait for every pipeline"""
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
