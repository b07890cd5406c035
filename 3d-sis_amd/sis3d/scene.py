"""Whole-scene inference over chunk grids (BASELINE config 5) on top of PipelinedEngines + parallel.

Each rank owns chunks c with c mod W == rank and runs the captured per-chunk graph on them, three chunks in flight on
three HIP streams (measured best for the detect pass: 2 -> 3 streams = +11 %); the fixed-size record blocks are all-gathered once per scene (RCCL) and every rank runs the same
whole-scene 3D NMS (HIP kernels)."""
import torch
import torch.distributed as dist

from . import ops, parallel
from .engine import PipelinedEngines


import os as _os
# capture_round's staggered start (pipeline e waits for pipeline e - 1's level 1): measured and lost on the 4-chunk share
# (1.357 vs 1.331 ms in the same box run, tools/r04_scene_share.sh) -- off by default, kept as an A/B switch
ROUND_STAGGER = _os.environ.get("SIS3D_ROUND_STAGGER", "0") != "0"


def fused_merge(blocks, k_rows, thresh, score_col, box_col, max_keep):
    """parallel.merge_scene's merge_fn on the GPU: sis3d_scene_merge for tables its sort takes (<= 8192 rows), else None"""
    if not blocks.is_cuda or blocks.shape[0] * int(k_rows) > 8192:
        return None
    return ops.scene_merge(blocks, k_rows, thresh, score_col, box_col, max_keep)


class SceneResult(object):
    """What a scene's merge left on the device: padded tables + the two data-dependent lengths, still unread.  `resolve()` does the
    one 8-byte readback and slices; a pipelined caller launches the next scene first and resolves this one afterwards, so the
    host never stands between two scenes."""

    def __init__(self, recs, order, keep, counts, k_rows, with_chunk_ids):
        self.recs, self.order, self.keep, self.counts = recs, order, keep, counts
        self.k_rows, self.with_chunk_ids = int(k_rows), bool(with_chunk_ids)
        self._out = None

    def resolve(self):
        if self._out is None:
            total, kept = self.counts.tolist()
            out = (self.recs[:total], self.keep[:kept])
            if self.with_chunk_ids:
                out = out + ((self.order[:total] // self.k_rows).long(),)
            self._out = out
        return self._out


class SceneRunner:
    def __init__(self, net, dims, use_graph=True, inflight=3, solo=False, emulate=None):
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
        self.pipes = PipelinedEngines(net, inflight, dims=dims, stage="detect", use_graph=use_graph).prepare()
        self._origins = {}
        self._send = None
        self._round = None          # (graph, stream, send buffer): one launch for a share of exactly len(engines) chunks

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

    def run_chunks(self, chunks, group=None):
        """this rank's chunks through the captured per-chunk graphs, `inflight` at a time -> (n_local, block_floats) tensor of
        record blocks in ascending chunk order (rows are written by async device copies on the pipelines' streams; the
        current stream waits for all of them before returning)"""
        rank, world = self._rank_world(group)
        mine = parallel.shard_chunks(len(chunks), rank, world)
        bf = parallel.block_floats(self.k_rows)
        dev = self.pipes.engines[0].device
        if self.use_graph and len(mine) == len(self.pipes.engines) and all(not isinstance(chunks[c][2], (tuple, list)) for c in mine):
            return self._run_round(chunks, mine, bf, dev)
        if self._send is None or self._send.shape[0] != len(mine):
            self._send = torch.zeros(max(1, len(mine)), bf, device=dev)
        send = self._send
        with torch.no_grad():
            for j, c in enumerate(mine):
                cid, origin, payload = chunks[c]
                e = j % len(self.pipes.engines)
                if isinstance(payload, (tuple, list)):
                    self.pipes.load(e, *payload)
                else:
                    self.pipes.load(e, payload)
                eng = self.pipes.engines[e]
                with torch.cuda.stream(self.pipes.streams[e]):
                    eng.origins[0].copy_(self._origin(origin), non_blocking=True)
                    out = eng.run()
                    send[j].copy_(out["block"], non_blocking=True)     # out of the graph's static buffer before its next replay
            self.pipes.join()
        return send[:len(mine)]

    def _run_round(self, chunks, mine, bf, dev):
        """the share is exactly one chunk per pipeline: inputs copied into the static buffers on ONE stream, then ONE graph launch
        (PipelinedEngines.capture_round) runs all of them and leaves their record blocks in the send buffer; the caller's stream
        is ordered behind the graph, no host-side join"""
        if self._round is None:
            send = torch.zeros(len(mine), bf, device=dev)
            g, main = self.pipes.capture_round(send, stagger=ROUND_STAGGER)
            self._round = (g, main, send)
        g, main, send = self._round
        cur = torch.cuda.current_stream()
        main.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(main):
            for e, c in enumerate(mine):
                cid, origin, payload = chunks[c]
                eng = self.pipes.engines[e]
                eng._copy(eng.scenes[0], payload)
                eng.origins[0].copy_(self._origin(origin), non_blocking=True)
                if payload.is_cuda:
                    payload.record_stream(main)
            g.replay()
        cur.wait_stream(main)
        return send

    def infer(self, chunks, thresh=None, group=None, max_keep=0, with_masks=False, mask_values=False, gathered=None, lazy=False):
        """chunks: [(chunk_id, origin, data or (data, feats, i3d, i2d) or None)] for the whole scene (entries of other
        ranks' chunks may carry None).  -> (records (N,16) sorted by score, keep LongTensor) on the GPU, identical on
        every rank; with_masks adds this rank's {position in keep: (scene window, mask)} (parallel.scene_masks).
        gathered: a (n_chunks, block_floats) table standing in for the collective's result -- this rank's fresh rows are written
        over its own chunks' rows and the merge runs on the FULL table (bench.py: what one rank of an N-rank run does, minus
        the collective itself).  lazy: return a SceneResult (lengths still on the device) instead of resolving it."""
        thresh = float(self.net.cfg.TEST.RPN_NMS_THRESH) if thresh is None else thresh
        n_chunks = len(chunks)
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
                return parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, merge_fn=fused_merge)
            recs, keep, cids = parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, with_chunk_ids=True,
                                                    merge_fn=fused_merge)
            fn = (lambda p, w, k: self.mask_fn(p, w, k, values=True)) if mask_values else self.mask_fn
            masks = parallel.scene_masks(recs, keep, cids, chunks, fn, float(self.net.cfg.CLASS_THRESH), group, solo=self.solo)
            return recs, keep, masks
