"""Whole-scene inference over chunk grids (BASELINE config 5) on top of ChunkEngine + parallel.

Each rank owns chunks c with c mod W == rank and runs the captured per-chunk graph on them; the fixed-size
record blocks are all-gathered once per scene and every rank runs the same whole-scene 3D NMS (HIP kernels)."""
import torch

from . import ops, parallel
from .engine import ChunkEngine


class SceneRunner:
    def __init__(self, net, dims, use_graph=True):
        self.net = net
        self.k_rows = int(net.cfg.TEST.RPN_POST_NMS_TOP_N)
        self.engine = ChunkEngine(net, dims=dims, stage="detect", use_graph=use_graph).prepare()

    def _detect(self, payload):
        if isinstance(payload, (tuple, list)):
            self.engine.load(*payload)
        else:
            self.engine.load(payload)
        out = self.engine.run()
        return out["records"], out["num"]            # static buffers: consumed (packed) before the next run, stream-ordered

    def infer(self, chunks, thresh=None, group=None, max_keep=0):
        """chunks: [(chunk_id, origin, data or (data, feats, i3d, i2d))] for the whole scene.
        -> (records (N,10) sorted by score, keep LongTensor) on the GPU, identical on every rank."""
        thresh = float(self.net.cfg.TEST.RPN_NMS_THRESH) if thresh is None else thresh
        with torch.no_grad():
            return parallel.infer_scene(chunks, self._detect, ops.nms, self.k_rows, thresh, group=group, max_keep=max_keep)
