"""Mirror of the reference's `benchmark` mode (lib/model/trainval.py:634-767, `SolverWrapper.benchmark`):
whole-scene forward on the GPU and the result files `tools/scannet_benchmark/vox2mesh.py:42-72` consumes.

Per scene directory `<TEST_SAVE_DIR>/<scene id[:12]>/`:
    pred_class.npy  int64 (R,)      arg-max class per RoI
    pred_conf.npy   float64 (R,)    its soft-max probability
    pred_box.npy    float32 (R,6)   class-specific regressed box, clipped to the scene
    scene.npy       int64 (X,Y,Z)   occupancy  (|tsdf| channel 0 <= 1)
    pred_mask       pickle: list of float32 {0,1} crops, one per kept detection   (USE_MASK)
    pred_mask_index pickle: list of bool (R,), which RoIs were kept             (USE_MASK)
A scene whose pred_box.npy already exists is not recomputed (resume rule, trainval.py:650-654).
"""
import os
import pickle

import numpy as np
import torch

from ..layer_utils.projection import prepare_projection
from ..utils.bbox_transform import bbox_transform_inv, clip_boxes


def final_detections(predictions, scene_info, cfg):
    """trainval.py:686-712: per-RoI class pick, box regression of that class, keep rule.
    -> pred_class int64 (R,), pred_conf float64 (R,), pred_box float32 (R,6), keep list[bool]."""
    cls_t, rois_t = predictions["cls_pred"].detach(), predictions["rois"][0].detach()
    reg_t, prob_t = predictions["bbox_pred"].detach(), predictions["cls_prob"].detach()
    if cls_t.is_cuda and cls_t.shape[0] > 0:
        # one D2H instead of four (each is a stream sync): class ids ride along as exact small floats
        nc = prob_t.shape[1]
        host = torch.cat([rois_t.float(), reg_t.float(), prob_t.float(), cls_t.view(-1, 1).float()], 1).cpu()
        rois_t, reg_t, prob_t = host[:, :6], host[:, 6:6 + 6 * nc], host[:, 6 + 6 * nc:6 + 7 * nc]
        cls_t = host[:, -1].long()
    pred_class = cls_t.cpu().numpy()
    rois = rois_t.cpu().contiguous()
    reg_all = reg_t.cpu().numpy()
    prob_all = prob_t.cpu().numpy()
    R = pred_class.shape[0]
    rows = np.arange(R)
    cols = pred_class[:, None] * 6 + np.arange(6)[None, :]
    box_reg = np.zeros((R, 6))
    box_reg[:, :] = reg_all[rows[:, None], cols] if R else 0
    pred_conf = np.zeros((R,))
    pred_conf[:] = prob_all[rows, pred_class] if R else 0
    pred_box = bbox_transform_inv(rois, torch.from_numpy(box_reg).float())
    pred_box = clip_boxes(pred_box, scene_info[:3]).numpy()
    keep = [bool(c > cfg.CLASS_THRESH) for c in pred_conf]
    for i, b in enumerate(pred_box):                          # degenerate after rounding to voxels (:709-712)
        if round(b[0]) >= round(b[3]) or round(b[1]) >= round(b[4]) or round(b[2]) >= round(b[5]):
            keep[i] = False
    return pred_class, pred_conf, pred_box, keep


def mask_windows(pred_box, keep):
    """integer crop windows of the kept boxes; Python round = half-to-even, as trainval.py:742-745"""
    return [tuple(int(round(b[k])) for k in range(6)) for b, s in zip(pred_box, keep) if s]


def binarise_masks(mask_pred, pred_class, keep, cfg):
    """trainval.py:751-759: the predicted class's channel of every kept detection, thresholded to {0,1} float32"""
    out = []
    it = iter(mask_pred)
    for cls, s in zip(pred_class, keep):
        if s:
            m = next(it)[0][int(cls)].detach().cpu().numpy()
            out.append(np.where(m >= cfg.MASK_THRESH, 1, 0).astype(np.float32))
    return out


def scene_dir(cfg, blobs):
    return "{}/{}".format(cfg.TEST_SAVE_DIR, blobs["id"][0].split("/")[-1][:12])


class SolverWrapper(object):
    @staticmethod
    def benchmark(net, data_loader, data_logger=None, cfg=None):
        """Same call as the reference (`SolverWrapper.benchmark(net, loader, logger)`); `cfg` defaults to net.cfg."""
        cfg = cfg if cfg is not None else net.cfg
        os.makedirs(cfg.TEST_SAVE_DIR, exist_ok=True)
        written = []
        for blobs in data_loader:
            out = scene_dir(cfg, blobs)
            done = os.path.isfile(out + "/pred_box.npy")
            if done:
                pred_class = np.load(out + "/pred_class.npy")
                pred_conf = np.load(out + "/pred_conf.npy")
                pred_box = np.load(out + "/pred_box.npy")
            else:
                killing_inds = None
                if cfg.USE_IMAGES:
                    killing_inds = prepare_projection(blobs, cfg)
                net.forward(blobs, "TEST", killing_inds)
                pred_class, pred_conf, pred_box, keep = final_detections(net._predictions, net._scene_info, cfg)
                os.makedirs(out, exist_ok=True)
                np.save(out + "/pred_class", pred_class)
                np.save(out + "/pred_conf", pred_conf)
                np.save(out + "/pred_box", pred_box)
                np.save(out + "/scene", np.where(blobs["data"][0, 0].cpu().numpy() <= 1, 1, 0))
            if cfg.USE_MASK:
                keep = [bool(c > cfg.CLASS_THRESH) for c in pred_conf]
                for i, b in enumerate(pred_box):
                    if round(b[0]) >= round(b[3]) or round(b[1]) >= round(b[4]) or round(b[2]) >= round(b[5]):
                        keep[i] = False
                if done or "mask_pred" not in net._predictions:
                    # resumed scene: only the mask head runs, on crops of the stored boxes (:736-749)
                    scene = blobs["data"].cuda().float()
                    net._predictions["mask_pred"] = [net.mask_backbone.forward_batched(scene, mask_windows(pred_box, keep))]
                pred_mask = binarise_masks(net._predictions["mask_pred"][0], pred_class, keep, cfg)
                with open(out + "/pred_mask", "wb") as f:
                    pickle.dump(pred_mask, f)
                with open(out + "/pred_mask_index", "wb") as f:
                    pickle.dump(keep, f)
            written.append(out)
        return written


benchmark = SolverWrapper.benchmark
