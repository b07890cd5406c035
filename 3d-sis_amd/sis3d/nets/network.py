"""Mirror of lib/nets/network.py for the TEST branch (forward, lines 187-317).

Same public surface as the reference's `Network`: `init_modules()`, `forward(blobs, mode,
killing_inds)`, `_predictions{}`, `_scene_info`, `mask_backbone`, `delete_intermediate_states()`,
and the same parameter tree.  Host code stays Python; every hot op is a HIP kernel launch on
the current stream.  Between the H2D copy of the chunk and the single 4-byte read of the RoI
count there is NO host synchronisation (the reference has three: NMS mask D2H, .nonzero() in
_roi_pool_layer, .cpu().numpy() in the mask branch -- SURVEY.md 3.1).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..config import anchor_sizes, cfg as _default_cfg
from ..layer_utils.generate_anchors import anchors_for_level
from ..layer_utils.proposal_layer import ProposalEngine


class Network(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg or _default_cfg
        self._predictions = {}
        self._anchor_targets = {}
        self._proposal_targets = {}
        self._mask_targets = {}
        self._losses = {}
        self._proposals = ProposalEngine(self.cfg)
        self._head_cache = {}
        self.batch_rpn = True             # both RPN k3 convs in one batched launch
        self.batch_masks = True           # mask head of all detected boxes as one ragged launch per layer
        self.fuse_projection = True       # colour stem reads the views through a voxel->pixel table; no 226 MB volume
        self._image_input = None
        self._image_dense = self._image_dense_of = None

    @property
    def _imageft(self):
        """network.py:239 `self._imageft`: the back-projected volume, logical (1,C,X,Y,Z).  With fuse_projection it is
        only materialised when somebody asks for it."""
        v = self._image_input
        if isinstance(v, ops.ProjectedVolume):
            if self._image_dense_of is not v:
                self._image_dense, self._image_dense_of = v.dense(), v
            return self._image_dense
        return v

    @_imageft.setter
    def _imageft(self, value):
        self._image_input = value

    # network.py:35-64
    def init_modules(self):
        from . import backbones
        cfg = self.cfg
        self._init_backbone_classifier()
        if cfg.USE_RPN:
            for lv, ch in ((1, self._net_conv_level1_channels), (2, self._net_conv_level2_channels),
                           (3, self._net_conv_level3_channels)):
                A = cfg["NUM_ANCHORS_LEVEL%d" % lv]
                if A != 0:
                    setattr(self, "rpn_net_level%d" % lv, backbones.HipConv3d(ch, cfg.RPN_CHANNELS, 3, padding=1, fuse_relu=True))
                    setattr(self, "rpn_cls_score_net_level%d" % lv, nn.Sequential(nn.Conv3d(cfg.RPN_CHANNELS, A * 2, 1)))
                    setattr(self, "rpn_bbox_pred_net_level%d" % lv, nn.Conv3d(cfg.RPN_CHANNELS, A * 6, 1))
        if cfg.USE_CLASS:
            self.classifier_cls_score_net = nn.Linear(self._fc7_channels, cfg.NUM_CLASSES)
            self.classifier_bbox_pred_net = nn.Linear(self._fc7_channels, cfg.NUM_CLASSES * 6)
        if cfg.USE_MASK:
            self.mask_backbone = getattr(backbones, cfg.MASK_BACKBONE)(cfg=cfg)
        if cfg.USE_IMAGES and not cfg.USE_IMAGES_GT:
            # network.py:63-64: the 2D encoder, split into its frozen and trainable halves + classifier (same attribute
            # names = same state_dict keys).  The module tree holds the parameters; on the GPU its eval forward runs on csrc/enet.hip (image_features).
            from . import enet
            self.image_enet_fixed, self.image_enet_trainable, self.image_enet_classification = enet.create_enet_for_3d(
                cfg.NUM_2D_CLASSES, cfg.get("PRETRAINED_ENET_PATH", ""), cfg.NUM_CLASSES)

    def delete_intermediate_states(self):
        for d in (self._losses, self._predictions, self._anchor_targets, self._proposal_targets, self._mask_targets):
            for k in list(d):
                del d[k]

    # ------------------------------------------------------------------ pieces --
    def _rpn_head(self, lv):
        """cls (2A) and bbox (6A) 1x1x1 convs packed as ONE 8A-channel GEMM (network.py:41-42,541-543)."""
        cls = getattr(self, "rpn_cls_score_net_level%d" % lv)[0]
        box = getattr(self, "rpn_bbox_pred_net_level%d" % lv)
        ver = (cls.weight._version, cls.bias._version, box.weight._version, box.bias._version, cls.weight.data_ptr())
        hit = self._head_cache.get(lv)
        if hit is None or hit[0] != ver:
            w = torch.cat([cls.weight.detach(), box.weight.detach()], 0)
            b = torch.cat([cls.bias.detach(), box.bias.detach()], 0)
            hit = (ver, ops.PackedConv(w, b, pad_cout16=True))
            self._head_cache[lv] = hit
        return hit[1]

    # network.py:503-534 + backbones.py:92-96 + network.py:589-604, on the padded K rows
    def _classify_rois(self, l1, l2):
        cfg, p = self.cfg, self._prop
        ps = cfg.CLASS_POOLING_SIZE
        if self._feat_stride[0] != self._feat_stride[1]:
            # network.py:520,529 pool level 2 with 1/feat_stride[1]; the fused two-level launch takes one scale
            raise NotImplementedError("pyramid levels with different strides (every shipped backbone uses 4, 4)")
        pool5 = ops.roi_pool_levels(l1, l2, p["rois"], p["levels"], ps, 1.0 / self._feat_stride[0], out_channels_last=True)
        self._pool5 = pool5
        x = pool5.permute(0, 2, 3, 4, 1).reshape(pool5.shape[0], -1)           # memory order (R, bins, C): a view
        fcs = [self.classifier[0], self.classifier[2], self.classifier[4]]
        heads = [self.classifier_cls_score_net, self.classifier_bbox_pred_net]
        ver = tuple(q._version for m in fcs + heads for q in (m.weight, m.bias)) + (fcs[0].weight.data_ptr(),)
        hit = self._head_cache.get("mlp")
        if hit is None or hit.version != ver:
            hit = ops.PackedClassifier(fcs, heads[0], heads[1], pool5.shape[1], ps ** 3)
            self._head_cache["mlp"] = hit
        return ops.classifier_forward(x, hit, p.get("num"))      # device-side row count: dead 32-row tiles are skipped

    # network.py:283-317
    def _mask_branch(self, n):
        """network.py:290-317: mask head on the crops of the confident, non-degenerate final boxes"""
        from ..model.trainval import final_detections, mask_windows
        _, _, pred_box, keep = final_detections(self._predictions, self._scene_info, self.cfg)
        self.mask_backbone.eval()
        windows = mask_windows(pred_box, keep)
        if getattr(self.mask_backbone, "use_images", False):
            # network.py:307-316 with USE_IMAGES: every box's crop of the scene AND of the back-projected volume
            vol = self._imageft
            if self.batch_masks and hasattr(self.mask_backbone, "forward_batched"):
                try:
                    return [self.mask_backbone.forward_batched(self._scene, windows, vol)]  # r3: one launch per layer for ALL boxes
                except ops.Sis3dUnsupported:
                    pass                        # a layer without a ragged instantiation: the per-box launches below still serve it
            return [[self.mask_backbone(self._scene, vol, window=w) for w in windows]]
        if self.batch_masks and hasattr(self.mask_backbone, "forward_batched"):
            return [self.mask_backbone.forward_batched(self._scene, windows)]   # one launch per layer for ALL boxes
        return [[self.mask_backbone(self._scene, None, window=w) for w in windows]]

    # ------------------------------------------------------------------ forward --
    def _rpn_level(self, lv, feat, rpn=None, heads=None):
        """network.py:539-549 for one pyramid level: k3 conv + ReLU, fused cls/bbox 1x1x1 heads, 2-way softmax"""
        cfg = self.cfg
        A = cfg["NUM_ANCHORS_LEVEL%d" % lv]
        if heads is not None:
            score, bbox, prob = heads                      # both levels' heads came out of one launch (ops.rpn_heads)
        else:
            if rpn is None:
                rpn = getattr(self, "rpn_net_level%d" % lv)(feat)
            score, bbox, prob = ops.conv3d(rpn, self._rpn_head(lv), rpn_anchors=A)     # softmax fused into the head epilogue
        self._predictions["rpn_cls_score_level%d" % lv] = score
        self._predictions["rpn_cls_prob_level%d" % lv] = prob
        self._predictions["rpn_bbox_pred_level%d" % lv] = bbox
        anchors = anchors_for_level(feat.shape[2:], self._feat_stride[lv - 1], anchor_sizes(cfg, lv))
        setattr(self, "_anchors_level%d" % lv, anchors)
        return (lv, prob, bbox, anchors)

    def image_features(self, images):
        """network.py:203-205: `image_enet_trainable(image_enet_fixed(images))`, eval mode, no grad.
        Inference only: the reference trains image_enet_trainable with gradients; this path (no_grad, BatchNorm folded into
        detached weights) cannot, and says so instead of silently freezing that half."""
        if self.training:
            raise NotImplementedError("sis3d: the RGB image path is inference-only (mode 'TEST', net.eval()); training "
                                      "image_enet_trainable is out of scope (SURVEY.md section 2)")
        with torch.no_grad():
            self.image_enet_fixed.eval()
            self.image_enet_trainable.eval()
            impl = getattr(self, "enet_impl", "hip") if getattr(self, "fold_enet", True) else "modules"
            if impl == "hip":
                if not images.is_cuda:
                    raise ops._lib.Sis3dError("image_features: images must be on the GPU (sis3d has no CPU path; enet_impl = 'folded' / "
                                          "fold_enet = False run the module tree on PyTorch operators)")
                # csrc/enet.hip: one launch per bottleneck (nets/enet_hip.py), 25 launches for the 5 views instead of ~190 operators
                if getattr(self, "_enet_hip", None) is None:
                    from .enet_hip import HipEncoder
                    self._enet_hip = HipEncoder(self.image_enet_fixed, self.image_enet_trainable)
                return self._enet_hip(images)
            if impl == "folded":
                # same modules on PyTorch-ROCm operators, BatchNorm + eval-dropout scale folded into the convolutions (nets/enet_folded.py)
                if getattr(self, "_enet_folded", None) is None:
                    from .enet_folded import FoldedEncoder
                    self._enet_folded = FoldedEncoder(self.image_enet_fixed, self.image_enet_trainable)
                return self._enet_folded(images)
            return self.image_enet_trainable(self.image_enet_fixed(images.float())).contiguous()

    def backbone_only(self, scene, imageft=None):
        """Device-only: the backbone proper (backbones.py:98-113) -> (level1, level2)."""
        self._scene = scene
        self._scene_info = scene.shape[2:]
        if imageft is not None:
            self._imageft = imageft
        l1 = self._backbone_level1()
        l2 = self._backbone_level2(l1)
        self._net_conv = (l1, l2)
        return l1, l2

    def backbone_rpn(self, scene, imageft=None):
        """Device-only: backbone + RPN convs/heads/softmax (BASELINE config 1).  Fills the rpn_* predictions.
        The two 128->256 k3 RPN convs (12.2 GFLOP each, the largest layers) are independent and of identical
        shape: they go out as ONE batched launch (864 workgroups instead of 2 x 432, which evens out the
        432-over-256-CUs quantisation that costs each of them ~17 %)."""
        self._scene = scene
        self._scene_info = scene.shape[2:]
        if imageft is not None:
            self._imageft = imageft
        cfg = self.cfg
        if cfg.NUM_ANCHORS_LEVEL3 != 0:
            raise NotImplementedError("three pyramid levels are not used by any shipped config")
        l1 = self._backbone_level1()
        hook = getattr(self, "_after_level1", None)
        if hook is not None:
            hook()                       # engine.PipelinedEngines.capture_round: the next pipeline of a one-launch round starts here
        l2 = self._backbone_level2(l1)
        self._net_conv = (l1, l2)
        levels = []
        if cfg.NUM_ANCHORS_LEVEL1 != 0 and cfg.NUM_ANCHORS_LEVEL2 != 0 and self.batch_rpn and l1.shape == l2.shape:
            c1, c2 = self.rpn_net_level1, self.rpn_net_level2
            r1, r2 = ops.conv3d_batched([l1, l2], [c1._packed.get(c1), c2._packed.get(c2)], relu=True)
            try:
                h1, h2 = ops.rpn_heads(r1, self._rpn_head(1), cfg.NUM_ANCHORS_LEVEL1, r2, self._rpn_head(2), cfg.NUM_ANCHORS_LEVEL2)
            except ops.Sis3dUnsupported:
                h1 = h2 = None
            levels.append(self._rpn_level(1, l1, r1, h1))
            levels.append(self._rpn_level(2, l2, r2, h2))
        else:
            if cfg.NUM_ANCHORS_LEVEL1 != 0:
                levels.append(self._rpn_level(1, l1))
            if cfg.NUM_ANCHORS_LEVEL2 != 0:
                levels.append(self._rpn_level(2, l2))
        return l1, l2, levels

    def backbone_rpn_group(self, scenes, imagefts=None):
        """Several independent chunks in one pass: backbones one after the other, then ONE batched launch of all their RPN
        k3 convs (2 chunks x 2 levels = 4 problems = 1728 workgroups = 6.75 per CU instead of 3.375: the 12-GFLOP convs are
        where the CU-count quantisation costs most).  -> [(l1, l2, levels, predictions dict)] per chunk."""
        cfg = self.cfg
        n = len(scenes)
        c1, c2 = self.rpn_net_level1, self.rpn_net_level2
        feats = []
        for i, sc in enumerate(scenes):
            self._scene, self._scene_info = sc, sc.shape[2:]
            if imagefts is not None:
                self._imageft = imagefts[i]
            l1 = self._backbone_level1()
            feats.append((l1, self._backbone_level2(l1)))
        if not (self.batch_rpn and 2 * n <= 4 and cfg.NUM_ANCHORS_LEVEL1 != 0 and cfg.NUM_ANCHORS_LEVEL2 != 0
                and all(a.shape == b.shape == feats[0][0].shape for a, b in feats)):
            raise ops.Sis3dUnsupported("grouped RPN launch needs <= 2 chunks of identical shape and both pyramid levels")
        rs = ops.conv3d_batched([f for pair in feats for f in pair], [c1._packed.get(c1), c2._packed.get(c2)] * n, relu=True)
        outs = []
        for i, (l1, l2) in enumerate(feats):
            self._scene, self._scene_info = scenes[i], scenes[i].shape[2:]
            self._predictions = {}
            self._net_conv = (l1, l2)
            try:
                h1, h2 = ops.rpn_heads(rs[2 * i], self._rpn_head(1), cfg.NUM_ANCHORS_LEVEL1, rs[2 * i + 1], self._rpn_head(2),
                                       cfg.NUM_ANCHORS_LEVEL2)
            except ops.Sis3dUnsupported:
                h1 = h2 = None
            levels = [self._rpn_level(1, l1, rs[2 * i], h1), self._rpn_level(2, l2, rs[2 * i + 1], h2)]
            outs.append((l1, l2, levels, self._predictions))
        return outs

    def detect_group(self, scenes, imagefts=None):
        """`detect` for a group of chunks sharing the batched RPN launch -> list of output dicts"""
        res = []
        for l1, l2, levels, pred in self.backbone_rpn_group(scenes, imagefts):
            self._predictions = pred
            self._prop = self._proposals.run(levels, self._scene_info[:3], "TEST")
            out = dict(rois=self._prop["rois"], scores=self._prop["scores"], levels=self._prop["levels"], num=self._prop["num"])
            if self.cfg.USE_CLASS:
                out["cls_score"], out["cls_pred"], out["cls_prob"], out["bbox_pred"] = self._classify_rois(l1, l2)
            res.append(out)
        return res

    def detect(self, scene, imageft=None):
        """Device-only, fixed-shape, sync-free detection pass (graph-capturable): backbone -> RPN ->
        decode/sort/NMS -> two-level RoI pooling -> classifier, on K = RPN_POST_NMS_TOP_N padded rows.
        Returns a dict of padded device tensors + `num` (int32 [1]) = number of valid rows."""
        l1, l2, levels = self.backbone_rpn(scene, imageft)
        self._prop = self._proposals.run(levels, self._scene_info[:3], "TEST")
        out = dict(rois=self._prop["rois"], scores=self._prop["scores"], levels=self._prop["levels"], num=self._prop["num"])
        if self.cfg.USE_CLASS:
            out["cls_score"], out["cls_pred"], out["cls_prob"], out["bbox_pred"] = self._classify_rois(l1, l2)
        return out

    def _apply(self, fn, *args, **kwargs):
        # every re-placement of the module (cuda / to / cpu / float ...) invalidates forward()'s cached device check
        self._placement_version = getattr(self, "_placement_version", 0) + 1
        return super()._apply(fn, *args, **kwargs)

    def forward(self, blobs, mode="TRAIN", killing_inds=None):
        if mode != "TEST":
            raise NotImplementedError("forward-only build: mode must be 'TEST' (training is out of scope, SURVEY.md 2)")
        cfg = self.cfg
        if blobs["data"].shape[0] != 1:
            raise NotImplementedError("batch size 1 (as the reference's TEST path)")
        if not (cfg.USE_BACKBONE and cfg.USE_RPN):
            raise NotImplementedError("USE_BACKBONE/USE_RPN=False (ground-truth boxes as RoIs) are training/debug modes")
        self._scene_info = blobs["data"].shape[2:]
        self._id = blobs["id"][0]
        self.batch_size = 1
        self._mode = "TEST"
        dev = torch.device("cuda", torch.cuda.current_device())
        # network.py:75: the reference's forward moves the module itself.  Walking every parameter costs tens of microseconds per call,
        # so the walk is repeated only after something re-placed the module (`_apply`: .cuda() / .to() / .cpu() bump the counter)
        placed = (getattr(self, "_placement_version", 0), dev)
        if getattr(self, "_placement_checked", None) != placed:
            if any(q.device != dev for q in self.parameters()):
                self.cuda()
            self._placement_checked = (getattr(self, "_placement_version", 0), dev)
        with torch.no_grad():
            self.eval()
            scene = blobs["data"].to(dev, non_blocking=True).float()
            self._gt_bbox = blobs.get("gt_box")
            self._gt_mask = blobs.get("gt_mask") if cfg.USE_MASK else None
            imageft = None
            if cfg.USE_IMAGES:
                feats = blobs["nearest_images"]["images"][0].to(dev, non_blocking=True)
                if not cfg.USE_IMAGES_GT:
                    # network.py:203-205: RGB views (V,3,256,328) -> ENet features (V,128,32,41)
                    feats = self.image_features(feats)
                p3 = blobs["proj_ind_3d"][0].to(dev, non_blocking=True)
                p2 = blobs["proj_ind_2d"][0].to(dev, non_blocking=True)
                project = ops.project_views_prepare if self.fuse_projection else ops.project_views_max
                imageft = project(feats, p3, p2, self._scene_info, killing_inds or ())
            d = self.detect(scene, imageft)
            # the only host sync of the detection path: number of surviving RoIs
            n = int(d["num"].item())
            self._predictions["rois"] = [d["rois"][:n]]
            self._predictions["roi_scores"] = [d["scores"][:n].view(-1, 1)]
            self._predictions["level_inds"] = [d["levels"][:n]]
            if cfg.USE_CLASS:
                for k in ("cls_score", "cls_pred", "cls_prob", "bbox_pred"):
                    self._predictions[k] = d[k][:n]
                if cfg.USE_MASK:
                    self._predictions["mask_pred"] = self._mask_branch(n)
        return self._predictions

    def _init_backbone_classifier(self):
        raise NotImplementedError

    def _backbone(self):
        raise NotImplementedError
