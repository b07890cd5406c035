"""Mirror of lib/utils/bbox_transform.py:4-21,59-99 (`clip_boxes`, `bbox_transform_inv`) as plain torch
ops for HOST-side callers (the mask branch decodes R<=200 class boxes on the CPU exactly like the
reference, network.py:293-294).  The RPN proposal decode on the device is sis3d_proposal_decode."""
import torch


def clip_boxes(boxes, scene_shape):
    return torch.stack([boxes[:, 0].clamp(0, scene_shape[0]), boxes[:, 1].clamp(0, scene_shape[1]),
                        boxes[:, 2].clamp(0, scene_shape[2]), boxes[:, 3].clamp(0, scene_shape[0]),
                        boxes[:, 4].clamp(0, scene_shape[1]), boxes[:, 5].clamp(0, scene_shape[2])], 1)


def bbox_transform_inv(boxes, deltas):
    if len(boxes) == 0:
        return deltas.detach() * 0
    w = (boxes[:, 3] - boxes[:, 0]).unsqueeze(1)
    h = (boxes[:, 4] - boxes[:, 1]).unsqueeze(1)
    l = (boxes[:, 5] - boxes[:, 2]).unsqueeze(1)
    cx = boxes[:, 0:1] + 0.5 * w
    cy = boxes[:, 1:2] + 0.5 * h
    cz = boxes[:, 2:3] + 0.5 * l
    pcx = deltas[:, 0::6] * w + cx
    pcy = deltas[:, 1::6] * h + cy
    pcz = deltas[:, 2::6] * l + cz
    pw = torch.exp(deltas[:, 3::6]) * w
    ph = torch.exp(deltas[:, 4::6]) * h
    pl = torch.exp(deltas[:, 5::6]) * l
    return torch.cat([pcx - 0.5 * pw, pcy - 0.5 * ph, pcz - 0.5 * pl, pcx + 0.5 * pw, pcy + 0.5 * ph, pcz + 0.5 * pl], 1)
