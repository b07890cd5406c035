"""Whole-scene inference over chunk grids (BASELINE config 5) on top of PipelinedEngines + parallel.

Each rank owns chunks c with c mod W == rank and runs the captured per-chunk graph on them, three chunks in flight on
three HIP streams (measured best for the detect pass: 2 -> 3 streams = +11 %); the fixed-size record blocks are all-gathered once per scene (RCCL) and every rank runs the same
whole-scene 3D NMS (HIP kernels)."""
import torch
import torch.distributed as dist

from . import ops, parallel
from .engine import PipelinedEngines


def fused_merge(blocks, k_rows, thresh, score_col, box_col, max_keep):
    """parallel.merge_scene's merge_fn on the GPU: sis3d_scene_merge for tables its sort takes (<= 8192 rows), else None"""
    if not blocks.is_cuda or blocks.shape[0] * int(k_rows) > 8192:
        return None
    return ops.scene_merge(blocks, k_rows, thresh, score_col, box_col, max_keep)


class SceneRunner:
    def __init__(self, net, dims, use_graph=True, inflight=3, solo=False):
        """solo: behave as a world of one even when a process group exists (the 1-GPU reference point bench.py takes on
        rank 0 inside an N-rank run); no collective is issued"""
        self.net = net
        self.solo = bool(solo)
        self.k_rows = int(net.cfg.TEST.RPN_POST_NMS_TOP_N)
        self.pipes = PipelinedEngines(net, inflight, dims=dims, stage="detect", use_graph=use_graph).prepare()
        self._origins = {}
        self._send = None

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

    def run_chunks(self, chunks, group=None):
        """this rank's chunks through the captured per-chunk graphs, `inflight` at a time -> (n_local, block_floats) tensor of
        record blocks in ascending chunk order (rows are written by async device copies on the pipelines' streams; the
        current stream waits for all of them before returning)"""
        live = dist.is_initialized() and not self.solo
        world = dist.get_world_size(group) if live else 1
        rank = dist.get_rank(group) if live else 0
        mine = parallel.shard_chunks(len(chunks), rank, world)
        bf = parallel.block_floats(self.k_rows)
        dev = self.pipes.engines[0].device
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

    def infer(self, chunks, thresh=None, group=None, max_keep=0, with_masks=False, mask_values=False):
        """chunks: [(chunk_id, origin, data or (data, feats, i3d, i2d) or None)] for the whole scene (entries of other
        ranks' chunks may carry None).  -> (records (N,16) sorted by score, keep LongTensor) on the GPU, identical on
        every rank; with_masks adds this rank's {position in keep: (scene window, mask)} (parallel.scene_masks)."""
        thresh = float(self.net.cfg.TEST.RPN_NMS_THRESH) if thresh is None else thresh
        n_chunks = len(chunks)
        with torch.no_grad():
            local = self.run_chunks(chunks, group)
            blocks = parallel.gather_blocks(local, n_chunks, self.k_rows, group, solo=self.solo)
            if not with_masks:
                return parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, merge_fn=fused_merge)
            recs, keep, cids = parallel.merge_scene(blocks, self.k_rows, ops.nms, thresh, max_keep=max_keep, with_chunk_ids=True,
                                                    merge_fn=fused_merge)
            fn = (lambda p, w, k: self.mask_fn(p, w, k, values=True)) if mask_values else self.mask_fn
            masks = parallel.scene_masks(recs, keep, cids, chunks, fn, float(self.net.cfg.CLASS_THRESH), group, solo=self.solo)
            return recs, keep, masks
