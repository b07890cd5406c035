"""Mirror of lib/layer_utils/proposal_layer.py:11-204 for the TEST path, on the device.

`ProposalEngine` is the fixed-shape, sync-free pipeline the network uses
(decode -> stable descending sort -> top-N -> NMS -> first-K survivors, all on
the GPU, outputs padded to K rows + a device-side count).  `proposal_layer(...)`
keeps the reference's signature and list-of-variable-length-tensors return
(one 4-byte D2H read for the count).
"""
import hashlib

import numpy as np
import torch

from .. import ops
from ..config import cfg as _default_cfg


class ProposalEngine:
    def __init__(self, cfg=None):
        self.cfg = cfg or _default_cfg
        self._cache = {}
        self._ident = {}

    def _level_tables(self, anchors, dims, device):
        """device anchors + int32 inside-index list (proposal_layer.py:36-43), cached per shape."""
        # keyed on the CONTENT of the whole anchor table (a few hundred KB, hashed once per call: ~0.1 ms): neither
        # id() (recycled after garbage collection) nor a prefix of the rows identifies an anchor set
        a = anchors if isinstance(anchors, np.ndarray) else anchors.detach().cpu().numpy()
        a = np.ascontiguousarray(a, dtype=np.float32)
        memo = self._ident.get(id(anchors))
        if memo is not None and memo[0] is anchors and isinstance(anchors, np.ndarray) and not anchors.flags.writeable:
            digest = memo[1]                                  # read-only array object seen before: content cannot have changed
        else:
            digest = hashlib.blake2b(a.tobytes(), digest_size=16).digest()
            if isinstance(anchors, np.ndarray) and not anchors.flags.writeable:
                self._ident[id(anchors)] = (anchors, digest)  # holds a reference, so the id cannot be recycled
        key = (a.shape[0], digest, tuple(int(d) for d in dims), str(device), float(self.cfg["ALLOW_BORDER"]))
        if key not in self._cache:
            b = np.float32(self.cfg["ALLOW_BORDER"])
            inside = np.where((a[:, 0] >= -b) & (a[:, 1] >= -b) & (a[:, 2] >= -b) &
                              (a[:, 3] < np.float32(dims[0]) + b) & (a[:, 4] < np.float32(dims[1]) + b) &
                              (a[:, 5] < np.float32(dims[2]) + b))[0].astype(np.int32)
            self._cache[key] = (torch.tensor(a).to(device), torch.from_numpy(inside).to(device))
        return self._cache[key]

    def run(self, levels, dims, cfg_key="TEST"):
        """levels: [(level_id, cls_prob (1,2,X,Y,Z,A), bbox_pred (1,X,Y,Z,6A), anchors np/tensor (K*A,6))].
        Returns dict(rois (K,6), scores (K,), levels (K,), num (1,) int32, order, keep, n_pre)."""
        c = self.cfg[cfg_key]
        pre_n, post_n, thr = int(c["RPN_PRE_NMS_TOP_N"]), int(c["RPN_POST_NMS_TOP_N"]), float(c["RPN_NMS_THRESH"])
        device = levels[0][1].device
        tabs = [self._level_tables(lv[3], dims, device) for lv in levels]
        M = sum(int(t[1].numel()) for t in tabs)
        boxes = torch.empty(M, 6, device=device)
        scores = torch.empty(M, device=device)
        lvl = torch.empty(M, device=device)
        for lid, prob, bbox, _ in levels:
            if not prob.is_contiguous() or not bbox.is_contiguous():
                raise ops._lib.Sis3dError("rpn maps must be contiguous (1,2,X,Y,Z,A) / (1,X,Y,Z,6A)")
        if len(levels) == 2:
            # r5: both levels in one launch (a kernel boundary costs a pipeline a wait for a free CU when several chunks are in flight)
            (l1, p1, b1, _), (l2, p2, b2, _) = levels
            ops.proposal_decode2(tabs[0][0], b1, p1[0, 1], tabs[0][1], l1, tabs[1][0], b2, p2[0, 1], tabs[1][1], l2, dims, boxes, scores, lvl)
        else:
            off = 0
            for (lid, prob, bbox, _), (anc, inside) in zip(levels, tabs):
                n = int(inside.numel())
                ops.proposal_decode(anc, bbox, prob[0, 1], inside, dims, lid, boxes[off:off + n], scores[off:off + n], lvl[off:off + n])
                off += n
        # stable descending sort: the tie rule pinned in the oracle (SURVEY.md 7 'Sort tie order')
        n_pre = min(pre_n, M) if pre_n > 0 else M
        if 0 < n_pre <= 1024 and M <= 40960:
            s_sorted, order = ops.topk_desc(scores, n_pre)     # one launch instead of a full 33k-element sort
        else:
            s_sorted, order = torch.sort(scores, descending=True, stable=True)
        k_out = post_n if post_n > 0 else n_pre
        rois, r_scores, r_levels, keep, num = ops.nms_select(boxes, lvl, s_sorted, order, n_pre, thr, k_out)
        return dict(rois=rois, scores=r_scores, levels=r_levels, num=num, order=order, keep=keep, n_pre=n_pre,
                    boxes_all=boxes, scores_all=scores)


_engine = None


def proposal_layer(rpn_cls_prob_level1, rpn_bbox_pred_level1, all_anchors_level1,
                   rpn_cls_prob_level2, rpn_bbox_pred_level2, all_anchors_level2,
                   rpn_cls_prob_level3, rpn_bbox_pred_level3, all_anchors_level3,
                   scene_info, cfg_key,
                   anchors_filter_level1=None, anchors_filter_level2=None, anchors_filter_level3=None, cfg=None):
    """Reference signature (proposal_layer.py:11-15); batch size 1.  Returns
    (proposals_batch, scores_batch, levelInds_batch): lists with one (R,6) / (R,1) / (R,) tensor."""
    global _engine
    if anchors_filter_level1 is not None or anchors_filter_level2 is not None or anchors_filter_level3 is not None:
        raise NotImplementedError("FILTER_ANCHOR_LEVEL* (training-time overfitting aid) is outside the forward path")
    if _engine is None or (cfg is not None and _engine.cfg is not cfg):
        _engine = ProposalEngine(cfg)
    levels = []
    for lid, p, b, a in ((1, rpn_cls_prob_level1, rpn_bbox_pred_level1, all_anchors_level1),
                         (2, rpn_cls_prob_level2, rpn_bbox_pred_level2, all_anchors_level2),
                         (3, rpn_cls_prob_level3, rpn_bbox_pred_level3, all_anchors_level3)):
        if p is not None:
            levels.append((lid, p, b, a))
    r = _engine.run(levels, scene_info[:3], cfg_key)
    n = int(r["num"].item())
    return [r["rois"][:n]], [r["scores"][:n].view(-1, 1)], [r["levels"][:n]]
