"""Whole-scene inference over chunk grids (BASELINE config 5) on top of PipelinedEngines + parallel.

Each rank owns chunks c with c mod W == rank and runs the captured per-chunk graph on them, three chunks in flight on
three HIP streams (measured best for the detect pass: 2 -> 3 streams = +11 %); the fixed-size record blocks are all-gathered once per scene (RCCL) and every rank runs the same
whole-scene 3D NMS (HIP kernels)."""
import torch
import torch.distributed as dist

from . import ops, parallel
from .engine import PipelinedEngines, pooled_stream


import os as _os
# capture_round's staggered start (pipeline e waits for pipeline e - 1's level 1): measured and lost on the 4-chunk share
# (1.357 vs 1.331 ms in the same box run, profiles/r04_scene_share_pipelines.txt) -- off by default, kept as an A/B switch
ROUND_STAGGER = _os.environ.get("SIS3D_ROUND_STAGGER", "0") != "0"
# lazy results: join + gather + whole-scene merge on their own stream, scenes overlap (A/B switch; on by default since r5)
MERGE_STREAM = _os.environ.get("SIS3D_MERGE_STREAM", "1") != "0"
# ... on the last pipeline's stream (default) or on a dedicated fifth stream (SIS3D_MERGE_ON_PIPELINE=0: faster when that stream happens to
# land on a good hardware queue, 60 % slower when it does not)
MERGE_ON_PIPELINE = _os.environ.get("SIS3D_MERGE_ON_PIPELINE", "1") != "0"


def fused_merge(blocks, k_rows, thresh, score_col, box_col, max_keep):
    """parallel.merge_scene's merge_fn on the GPU: sis3d_scene_merge for tables its sort takes (<= 8192 rows), else None"""
    if not blocks.is_cuda or blocks.shape[0] * int(k_rows) > 8192:
        return None
    return ops.scene_merge(blocks, k_rows, thresh, score_col, box_col, max_keep)


class SceneResult(object):
    """What a scene's merge left on the device: padded tables + the two data-dependent lengths, still unread.  `resolve()` does the
    one 8-byte readback and slices; a pipelined caller launches the next scene first and resolves this one afterwards, so the
    host never stands between two scenes."""

    def __init__(self, recs, order, keep, counts, k_rows, with_chunk_ids, done=None):
        self.recs, self.order, self.keep, self.counts = recs, order, keep, counts
        self.k_rows, self.with_chunk_ids = int(k_rows), bool(with_chunk_ids)
        self._done = done               # event on the merge stream (the tables are written there, not on the caller's stream)
        self._out = None

    @classmethod
    def resolved(cls, out):
        """an eager result behind the lazy interface: `infer(lazy=True)` ALWAYS returns a SceneResult, also on the paths that had to
        read lengths on the host already (masks, tables beyond the fused merge's 8192 rows, CPU tables)"""
        r = cls(None, None, None, None, 1, False)
        r._out = tuple(out)
        return r

    def resolve(self):
        if self._out is None:
            if self._done is not None:
                self._done.synchronize()
                # later work of the caller's stream that touches the tables is ordered behind the merge
                cur = torch.cuda.current_stream()
                cur.wait_event(self._done)
                # ... and the tables were ALLOCATED on the merge stream: tell the caching allocator that the caller's stream uses them
                # too, or the blocks go back to the merge stream's pool when the caller drops its slices and the next scene's merge
                # (already enqueued in lazy mode) may overwrite them under a kernel of the caller's that is still reading
                for t in (self.recs, self.order, self.keep, self.counts):
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
            total, kept = self.counts.tolist()
            out = (self.recs[:total], self.keep[:kept])
            if self.with_chunk_ids:
                out = out + ((self.order[:total] // self.k_rows).long(),)
            self._out = out
        return self._out


class SceneRunner:
    def __init__(self, net, dims, use_graph=True, inflight=3, solo=False, emulate=None, round_graph=False):
        """solo: behave as a world of one even when a process group exists (the 1-GPU reference point bench.py takes on
        rank 0 inside an N-rank run); no collective is issued.
        emulate = (rank, world): shard the scene as that rank of that world WITHOUT a process group (implies solo); together with
        infer(gathered=...) this is one rank's share of an N-rank scene on a single GPU -- its own chunks, then the merge of a
        full-size gathered table (bench.py's share_of_one_rank_at_8)."""
        self.net = net
        self.emulate = (int(emulate[0]), int(emulate[1])) if emulate is not None else None
        self.solo = bool(solo) or self.emulate is not None
        self.k_rows = int(net.cfg.TEST.RPN_POST_NMS_TOP_N)
        self.use_graph = bool(use_graph)
        self.dims = tuple(int(d) for d in dims)
        # mailbox engines (r5): a chunk costs the host one graph launch and nothing else (ChunkEngine.submit)
        self.pipes = PipelinedEngines(net, inflight, dims=dims, stage="detect", use_graph=use_graph,
                                      mailbox=(_os.environ.get("SIS3D_SCENE_MAILBOX", "1") != "0")).prepare()
        self._origins = {}
        self._send = None
        self._round = None          # (graph, stream, send buffer): one launch for a share of exactly len(engines) chunks
        self._round_ok, self._round_error = True, None
        self._merge_stream = None
        self._sends, self._consumed, self._scene_no, self._send_slot = [None, None], [None, None, None], 0, 0
        # round_graph: a share of exactly one chunk per pipeline as ONE graph launch (PipelinedEngines.capture_round).  r4's default; since
        # r5 per-chunk launches are (a chunk is one call, the streams are placed by measurement: 0.98 against 1.06 ms for a rank's share at
        # N = 8) and the round graph is what calibrate() may still choose when it measures faster
        self._use_round = bool(round_graph)
        self.calibration = None

    def mask_fn(self, payload, windows, classes, values=False):
        """mask head of one chunk's surviving detections: one ragged launch per layer for all its boxes, then the predicted
        class's channel thresholded at MASK_THRESH (lib/model/trainval.py:736-759) -> list of {0,1} float32 GPU tensors
        (values=True: the channel's sigmoid outputs themselves, what the parity tests compare at 1e-4)"""
        data = payload[0] if isinstance(payload, (tuple, list)) else payload
        scene = data.cuda().float()
        with torch.no_grad():
            preds = self.net.mask_backbone.forward_batched(scene, windows)
        if values:
            return [p[0, k].clone() for p, k in zip(preds, classes)]
        t = float(self.net.cfg.MASK_THRESH)
        return [(p[0, k] >= t).float() for p, k in zip(preds, classes)]

    def _origin(self, origin):
        """device copy of a chunk origin, made once per distinct origin (a pageable H2D copy per chunk would stall the host)"""
        key = (float(origin[0]), float(origin[1]), float(origin[2]))
        t = self._origins.get(key)
        if t is None:
            t = torch.tensor(key, device=self.pipes.engines[0].device)
            self._origins[key] = t
        return t

    def _rank_world(self, group=None):
        if self.emulate is not None:
            return self.emulate
        live = dist.is_initialized() and not self.solo
        return (dist.get_rank(group), dist.get_world_size(group)) if live else (0, 1)

    def run_chunks(self, chunks, group=None, post=None):
        """this rank's chunks through the captured per-chunk graphs, `inflight` at a time -> (n_local, block_floats) tensor of
        record blocks in ascending chunk order (rows are written by async device copies on the pipelines' streams; the
        current stream waits for all of them before returning).
        post (r5, the pipelined path of infer(lazy=True)): a stream that takes the place of the current stream as the one ordered
        behind the pipelines -- the caller gathers and merges there -- so the CURRENT stream never waits for a scene's chunks and the
        next scene's chunks start on each pipeline as soon as that pipeline is free, not when the slowest pipeline of this scene has
        finished.  The record rows then go to one of two alternating send buffers (the previous scene's may still be being read)."""
        rank, world = self._rank_world(group)
        mine = parallel.shard_chunks(len(chunks), rank, world)
        bf = parallel.block_floats(self.k_rows)
        dev = self.pipes.engines[0].device
        if self.use_graph and self._use_round and len(mine) == len(self.pipes.engines) \
                and all(not isinstance(chunks[c][2], (tuple, list)) for c in mine) and self.prepare_round():
            return self._run_round(chunks, mine, bf, dev, post)
        if post is not None:
            b = self._scene_no & 1
            self._scene_no += 1
            if self._sends[b] is None or self._sends[b].shape[0] != max(1, len(mine)):
                old = self._sends[b]
                if old is not None:
                    # the buffer being replaced may still be read by the gather of the scene before last (on `post`) and written by
                    # the pipelines: keep its block out of the allocator's hands until those streams have passed this point
                    for st in list(self.pipes.streams) + [post]:
                        old.record_stream(st)
                self._sends[b] = torch.zeros(max(1, len(mine)), bf, device=dev)
                self._consumed[b] = None
            send = self._sends[b]
            self._send_slot = b
        else:
            if self._send is None or self._send.shape[0] != len(mine):
                self._send = torch.zeros(max(1, len(mine)), bf, device=dev)
            send = self._send
        n = len(self.pipes.engines)
        nfloat = 2 * self.dims[0] * self.dims[1] * self.dims[2]
        with torch.no_grad():
            if post is not None and self._consumed[self._send_slot] is not None:
                for st in self.pipes.streams:                 # the scene before last has been gathered out of this send buffer
                    st.wait_event(self._consumed[self._send_slot])
            # the pipelines' streams are ordered behind the caller's stream ONCE per scene (the chunks of a scene are all there when
            # infer() is called), not once per chunk: 4 cross-queue barriers per scene instead of 32
            cur = torch.cuda.current_stream()
            for st in self.pipes.streams:
                st.wait_stream(cur)
            for j, c in enumerate(mine):
                cid, origin, payload = chunks[c]
                e = j % n
                eng = self.pipes.engines[e]
                org = (float(origin[0]), float(origin[1]), float(origin[2]))
                if eng.mail is not None:
                    # r5: ONE call per chunk -- the graph launch.  Where the grid comes from (device memory or PINNED HOST memory: the
                    # reference's forward owns the upload, lib/nets/network.py:191), the chunk origin and the send-buffer row that
                    # receives the record block travel in a mailbox slot (CPU stores) and are applied by kernels inside the graph; a
                    # copy enqueued behind a graph launch that has not finished can block the host on this runtime
                    src = None
                    if isinstance(payload, (tuple, list)):
                        self.pipes.load(e, *payload, wait=False)          # image-path payloads: feature maps + index lists, copied eagerly
                    else:
                        src = ops.mail_source(payload, nfloat)
                        if src is None:
                            self.pipes.load(e, payload, wait=False)       # pageable / strided / non-fp32 grids: an ordinary copy
                    # the chunk this pipeline gets NEXT (n chunks on): if it sits in pinned host memory it is pulled across the link
                    # while this one computes (piggyback row of the rpn_net conv launch, ChunkEngine.submit)
                    nxt = chunks[mine[j + n]][2] if (src is not None and j + n < len(mine)) else None
                    nxt = nxt if isinstance(nxt, torch.Tensor) and not nxt.is_cuda else None
                    with torch.cuda.stream(self.pipes.streams[e]):
                        eng.submit(src=src, block_dst=send[j], origin=org, next_src=nxt)
                    continue
                if isinstance(payload, (tuple, list)):
                    self.pipes.load(e, *payload, wait=False)
                else:
                    self.pipes.load(e, payload, wait=False)
                with torch.cuda.stream(self.pipes.streams[e]):
                    eng.origins[0].copy_(self._origin(origin), non_blocking=True)
                    out = eng.run()
                    send[j].copy_(out["block"], non_blocking=True)     # out of the graph's static buffer before its next replay
            if post is not None:
                for st in self.pipes.streams:
                    if st is not post:
                        post.wait_stream(st)
            else:
                self.pipes.join()
        return send[:len(mine)]

    def calibrate(self, chunks, group=None, reps=2, count=16, gathered=None, lazy=False):
        """choose the pipelines' streams -- and, for a share of exactly one chunk per pipeline, between the one-launch round graph
        and per-chunk launches -- by timing infer() on `chunks` (PipelinedEngines.calibrate; one-time cost: ~13 x 3 scenes).
        -> dict describing the choice"""
        import time
        use_round_saved = self._use_round
        self._use_round = False

        pending = []

        def once():
            # the caller's mode: lazy results are resolved one scene late, as a pipelined consumer does
            r = self.infer(chunks, group=group, gathered=gathered, lazy=lazy)
            if lazy:
                pending.append(r)
                if len(pending) > 1:
                    pending.pop(0).resolve()
        best, times = self.pipes.calibrate(once, reps=reps, warm=1, count=count)
        while pending:
            pending.pop(0).resolve()
        out = {"stream_window": best, "ms_per_scene_by_window": {str(k): round(v, 3) for k, v in times.items()}}
        self._use_round = use_round_saved
        rank, world = self._rank_world(group)
        if self.use_graph and len(parallel.shard_chunks(len(chunks), rank, world)) == len(self.pipes.engines) and self.prepare_round():
            res = {}
            for flag in (False, True):
                self._use_round = flag
                for _ in range(2):
                    self.infer(chunks, group=group, lazy=True, gathered=gathered).resolve()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                prev = None
                for _ in range(6):
                    cur = self.infer(chunks, group=group, lazy=True, gathered=gathered)
                    if prev is not None:
                        prev.resolve()
                    prev = cur
                prev.resolve()
                torch.cuda.synchronize()
                res[flag] = (time.perf_counter() - t0) / 6 * 1e3
            self._use_round = res[True] <= res[False]
            out.update({"ms_per_scene_round_graph": round(res[True], 3), "ms_per_scene_per_chunk_launches": round(res[False], 3),
                        "one_launch_round": self._use_round})
        self.calibration = out
        return out

    def prepare_round(self):
        """capture the one-launch round graph now (outside any timed or pipelined call): a share of exactly one chunk per pipeline
        then costs ONE graph launch per scene.  On failure the per-chunk path keeps serving (returns False)."""
        if self._round is None and self.use_graph and self._round_ok:
            bf = parallel.block_floats(self.k_rows)
            dev = self.pipes.engines[0].device
            try:
                torch.cuda.synchronize()
                send = torch.zeros(len(self.pipes.engines), bf, device=dev)
                g, main = self.pipes.capture_round(send, stagger=ROUND_STAGGER)
                self._round = (g, main, send)
            except Exception as e:                      # a capture that fails must not take inference down: per-chunk graphs serve
                self._round_ok = False
                self._round_error = "%s: %s" % (type(e).__name__, e)
        return self._round is not None

    def _run_round(self, chunks, mine, bf, dev, post=None):
        """the share is exactly one chunk per pipeline: every pipeline's mailbox slot is written (grid source, origin), then ONE graph
        launch (PipelinedEngines.capture_round) runs all of them and leaves their record blocks in the send buffer; the caller's
        stream (or `post`) is ordered behind the graph, no host-side join"""
        g, main, send = self._round
        cur = torch.cuda.current_stream()
        main.wait_stream(cur)
        if post is not None:
            self._send_slot = 2
            if self._consumed[2] is not None:
                main.wait_event(self._consumed[2])         # the previous scene's rows have been gathered out of the graph's send buffer
        nfloat = 2 * self.dims[0] * self.dims[1] * self.dims[2]
        with torch.no_grad(), torch.cuda.stream(main):
            for e, c in enumerate(mine):
                cid, origin, payload = chunks[c]
                eng = self.pipes.engines[e]
                org = (float(origin[0]), float(origin[1]), float(origin[2]))
                if eng.mail is not None:
                    src = ops.mail_source(payload, nfloat)
                    if src is None:
                        eng._copy(eng.scenes[0], payload)
                    eng.mail.write(src, None, org)             # the round graph copies the blocks into its own send buffer
                else:
                    eng._copy(eng.scenes[0], payload)
                    if payload.is_cuda:
                        payload.record_stream(main)
                    eng.origins[0].copy_(self._origin(origin), non_blocking=True)
            g.replay()
        (post if post is not None else cur).wait_stream(main)
        return send

    def _infer_pipelined(self, chunks, thresh, group, max_keep, gathered):
        """infer(lazy=True) on tables the fused merge serves: everything behind the per-chunk detection -- join, gather (the scene's
        ONE collective), whole-scene merge -- is enqueued on the merge stream, ordered behind the pipelines only.  The current stream
        carries nothing of a scene, so scene k + 1's chunks do not wait for scene k's slowest pipeline, gather or merge: consecutive
        scenes overlap on the chip."""
        if MERGE_ON_PIPELINE:
            # the scene's serial part rides on the LAST pipeline's stream (that pipeline's next chunk queues ~85 us behind it): a fifth
            # stream that carries kernels while four pipelines run is at the mercy of the stream -> hardware-queue placement (measured:
            # the same scene 7.2 or 12.0 ms depending on where a dedicated merge stream landed), four streams are what the chip serves
            ms = self.pipes.streams[-1]
        else:
            if self._merge_stream is None:
                self._merge_stream = pooled_stream("merge", 0)
            ms = self._merge_stream
        n_chunks = len(chunks)
        with torch.no_grad():
            ms.wait_stream(torch.cuda.current_stream())        # `gathered` / process-group state produced on the caller's stream
            local = self.run_chunks(chunks, group, post=ms)
            with torch.cuda.stream(ms):
                if gathered is not None:
                    rank, world = self._rank_world(group)
                    rows = parallel._chunk_ids(tuple(parallel.shard_chunks(n_chunks, rank, world)), local.device)
                    blocks = gathered.index_copy(0, rows, local)
                else:
                    blocks = parallel.gather_blocks(local, n_chunks, self.k_rows, group, solo=self.solo)
                    if blocks.data_ptr() == local.data_ptr():
                        blocks = blocks.clone()                   # a world of one: the table must not alias the send buffer
                ev = torch.cuda.Event()
                ev.record(ms)
                self._consumed[self._send_slot] = ev
                recs, order, keep, counts = ops.scene_merge_raw(blocks, self.k_rows, thresh, 6, 0, max_keep)
                done = torch.cuda.Event()
                done.record(ms)
        return SceneResult(recs, order, keep, counts, self.k_rows, False, done=done)

    def infer(self, chunks, thresh=None, group=None, max_keep=0, with_masks=False, mask_values=False, gathered=None, lazy=False):
        """chunks: [(chunk_id, origin, data or (data, feats, i3d, i2d) or None)] for the whole scene (entries of other
        ranks' chunks may carry None).  -> (records (N,16) sorted by score, keep LongTensor) on the GPU, identical on
        every rank; with_masks adds this rank's {position in keep: (scene window, mask)} (parallel.scene_masks).
        gathered: a (n_chunks, block_floats) table standing in for the collective's result -- this rank's fresh rows are written
        over its own chunks' rows and the merge runs on the FULL table (bench.py: what one rank of an N-rank run does, minus
        the collective itself).  lazy: ALWAYS return a SceneResult -- lengths still on the device where the fused merge serves
        the table, already resolved otherwise (with_masks, > 8192 rows, CPU tables) -- whose resolve() gives the eager tuple."""
        thresh = float(self.net.cfg.TEST.RPN_NMS_THRESH) if thresh is None else thresh
        n_chunks = len(chunks)
        # every table the fused merge serves takes the pipelined path -- also for eager callers, who resolve at once: nothing of a scene
        # is enqueued on the caller's (usually the null) stream, whose hardware queue the pipelines may share (r5: with the serial part
        # on the null stream the same scene took 7.5 or 12 ms depending on the pipelines' stream window; on the pipelined path 10 of 13
        # windows are within 3 % of the best)
        pipelined = (MERGE_STREAM and not with_masks and n_chunks * self.k_rows <= 8192
                     and self.pipes.engines[0].device.type == "cuda")
        if pipelined:
            r = self._infer_pipelined(chunks, thresh, group, max_keep, gathered)
            return r if lazy else r.resolve()
        with torch.no_grad():
            local = self.run_chunks(chunks, group)
            if gathered is not None:
                rank, world = self._rank_world(group)
                rows = parallel._chunk_ids(tuple(parallel.shard_chunks(n_chunks, rank, world)), local.device)
                blocks = gathered.index_copy(0, rows, local)
            else:
                blocks = parallel.gather_blocks(local, n_chunks, self.k_rows, group, solo=self.solo)
            if lazy and not with_masks and blocks.is_cuda and blocks.shape[0] * self.k_rows <= 8192:
                recs, order, keep, counts = ops.scene_merge_raw(blocks, self.k_rows, thresh, 6, 0, max_keep)
                return SceneResult(recs, order, keep, counts, self.k_rows, False)
            if not with_masks:
                out = parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, merge_fn=fused_merge)
                return SceneResult.resolved(out) if lazy else out
            recs, keep, cids = parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, with_chunk_ids=True,
                                                    merge_fn=fused_merge)
            fn = (lambda p, w, k: self.mask_fn(p, w, k, values=True)) if mask_values else self.mask_fn
            masks = parallel.scene_masks(recs, keep, cids, chunks, fn, float(self.net.cfg.CLASS_THRESH), group, solo=self.solo)
            return SceneResult.resolved((recs, keep, masks)) if lazy else (recs, keep, masks)
