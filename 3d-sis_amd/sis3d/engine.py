"""ChunkEngine: the per-GPU inference loop over fixed-shape voxel chunks.

One engine = one network replica + static input buffers + ONE captured HIP graph of the whole
sync-free detection pass (~60 kernel launches: convs, heads, decode, sort, NMS, RoI pooling,
classifier).  Per chunk the host does: async H2D copy of the 3.5 MB grid into the static
buffer, one graph launch, (optionally) one D2H of the fixed-size record block.  Launch-bound
Python/ctypes overhead (~10 us per op) disappears from the steady state.
"""
import torch

from . import ops
from .synthetic import CHUNK_DIMS

RECORD_WIDTH = 10     # x1,y1,z1,x2,y2,z2, rpn score, level, class id, class prob


class ChunkEngine:
    def __init__(self, net, dims=CHUNK_DIMS, stage="detect", use_graph=True, n_views=0, device=None):
        """stage: 'rpn' (backbone + RPN maps, BASELINE config 1) or 'detect' (+ proposals, RoI pooling, classifier)."""
        self.net, self.dims, self.stage, self.use_graph = net, tuple(dims), stage, use_graph
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        cfg = net.cfg
        self.scene = torch.zeros((1, 2) + self.dims, device=self.device)
        self.use_images = bool(cfg.USE_IMAGES)
        if self.use_images:
            nvox = self.dims[0] * self.dims[1] * self.dims[2]
            h, w = cfg.DEPTH_SHAPE[1], cfg.DEPTH_SHAPE[0]
            self.n_views = n_views or cfg.NUM_IMAGES
            self.feats = torch.zeros(self.n_views, cfg.NUM_IMAGE_CHANNELS, h, w, device=self.device)
            self.i3d = torch.zeros(self.n_views, nvox + 1, dtype=torch.int64, device=self.device)
            self.i2d = torch.zeros(self.n_views, nvox + 1, dtype=torch.int64, device=self.device)
        self.graph = None
        self.out = None
        self.records = None

    def _step(self):
        net = self.net
        imageft = None
        if self.use_images:
            imageft = ops.project_views_max(self.feats, self.i3d, self.i2d, self.dims, (), channels_last=True)
        if self.stage == "rpn":
            net.backbone_rpn(self.scene, imageft)
            return {k: v for k, v in net._predictions.items() if k.startswith("rpn_")}
        d = net.detect(self.scene, imageft)
        if "cls_pred" in d:
            conf = d["cls_prob"].gather(1, d["cls_pred"].view(-1, 1))[:, 0]
            rec = torch.cat([d["rois"], d["scores"].view(-1, 1), d["levels"].view(-1, 1),
                             d["cls_pred"].float().view(-1, 1), conf.view(-1, 1)], 1)
        else:
            z = torch.zeros_like(d["scores"]).view(-1, 1)
            rec = torch.cat([d["rois"], d["scores"].view(-1, 1), d["levels"].view(-1, 1), z, z], 1)
        d["records"] = rec
        return d

    def prepare(self, warmup=2):
        """warm caches (weight repack, anchor tables) and capture the graph"""
        with torch.no_grad():
            self.net.eval()
            for _ in range(max(1, warmup)):
                self.out = self._step()
            torch.cuda.synchronize()
            if self.use_graph:
                side = torch.cuda.Stream()
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

    def load(self, data, feats=None, i3d=None, i2d=None):
        self.scene.copy_(data, non_blocking=True)
        if self.use_images:
            self.feats.copy_(feats, non_blocking=True)
            self.i3d.copy_(i3d, non_blocking=True)
            self.i2d.copy_(i2d, non_blocking=True)

    def run(self):
        """one pass over the chunk currently in the static buffers; returns the (static) output dict"""
        with torch.no_grad():
            if self.graph is not None:
                self.graph.replay()
            else:
                self.out = self._step()
        return self.out
