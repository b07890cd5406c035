"""Mask head with colour (lib/nets/backbones.py:253-284, MASK_USE_IMAGES / MASK_ONLY_IMAGES; VERDICT r1 missing #3).
Fixtures e2e_mask_{use,only}_images_small.npz: the reference's own forward with those switches (oracle/make_golden.py
--mask-images).  CPU: the oracle restatement is pinned to them and the product's parameter tree matches the reference's;
GPU: the product forward against the oracle, mask values included."""
import numpy as np
import pytest
import torch

from sis3d import config, synthetic


def _cfg(kind):
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_MASK = True, True
    c.MASK_USE_IMAGES = True
    c.MASK_ONLY_IMAGES = kind == "only"
    return c


def _inputs(g):
    dims = tuple(int(v) for v in g["dims"])
    cid = int(g["chunk_id"])
    data = synthetic.synth_chunk(cid, dims)
    feats, i3d, i2d = synthetic.synth_views(cid, n_views=int(g["n_views"]), n_per_view=int(g["n_per_view"]), dims=dims)
    return dims, data, feats, i3d, i2d


@pytest.mark.parametrize("kind", ["use", "only"])
def test_oracle_and_state_dict_match_reference_fixture(golden, oracle, kind):
    g = golden("e2e_mask_%s_images_small" % kind)
    from sis3d.nets.backbones import state_dict_shapes
    c = _cfg(kind)
    shapes = state_dict_shapes(c)
    assert sorted(shapes) == list(g["shapes_keys"])                   # mask_backbone.{geometry,color,combine}.N.weight
    assert shapes["mask_backbone.combine.0.weight"] == (128, 128, 3, 3, 3)
    assert shapes["mask_backbone.color.10.weight"] == ((19 if kind == "only" else 64), 64, 1, 1, 1)
    assert shapes["mask_backbone.geometry.10.weight"] == (64, 64, 1, 1, 1)
    sd = synthetic.synth_checkpoint(shapes, seed=0)
    dims, data, feats, i3d, i2d = _inputs(g)
    o = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2)).forward(data, feats, i3d, i2d)
    assert np.array_equal(o["rois"][0].numpy(), g["rois"]) and np.array_equal(o["cls_pred"].numpy(), g["cls_pred"])
    masks = o["mask_pred"][0]
    assert len(masks) == int(g["n_masks"]) > 0
    assert np.array_equal(np.array([list(m.shape[2:]) for m in masks]).reshape(-1, 3), g["mask_shapes"])
    for i in range(min(4, len(masks))):
        assert np.array_equal(masks[i].numpy(), g["mask_%d" % i])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["use", "only"])
def test_mask_head_with_colour_vs_oracle(golden, oracle, kind):
    from parity import assert_proposals_match
    from sis3d.model.trainval import final_detections, mask_windows
    from sis3d.nets import backbones
    g = golden("e2e_mask_%s_images_small" % kind)
    c = _cfg(kind)
    net = backbones.ScanNet_Backbone(cfg=c)
    net.init_modules()
    sd = synthetic.synth_checkpoint({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=True)
    net.cuda().eval()
    dims, data, feats, i3d, i2d = _inputs(g)
    blobs = {"data": data, "id": ["m"], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]], "nearest_images": {"images": [feats]},
             "proj_ind_3d": [i3d], "proj_ind_2d": [i2d]}
    p = net.forward(blobs, "TEST", [])
    o = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2)).forward(data, feats, i3d, i2d)
    assert_proposals_match(p["rois"][0].cpu(), p["roi_scores"][0].cpu(), o["rois"][0], o["roi_scores"][0],
                           o["_scores_sorted_all"], label="mask head, colour variant %s" % kind)     # asserts 0 near-ties
    _, _, pred_box, keep = final_detections(p, data.shape[2:], c)
    wins = mask_windows(pred_box, keep)
    masks = p["mask_pred"][0]
    owins = [tuple(b) for b in o["_mask_aux"]["crops"]]
    assert len(masks) == len(wins) > 0
    assert wins == owins
    by = dict(zip(owins, o["mask_pred"][0]))
    checked = 0
    for w, m in zip(wins, masks):
        assert tuple(m.shape) == (1, c.NUM_CLASSES, w[3] - w[0], w[4] - w[1], w[5] - w[2])
        assert float((m.cpu() - by[w]).abs().max()) <= 1e-4, w
        checked += 1
    assert checked == len(owins) > 0
    # r3: the masks above came out of the BATCHED colour-variant head (one ragged launch per layer for all boxes); the per-box
    # launches of round 2 give the same values (same kernels per layer up to the ragged / Winograd routing: fp32 noise)
    assert net.batch_masks
    vol = net._imageft
    for w, m in zip(wins, masks):
        single = net.mask_backbone(net._scene, vol, window=w)
        assert float((m - single).abs().max()) <= 1e-5, w
    # against the reference's own first masks
    for i in range(min(4, int(g["n_masks"]))):
        assert float((dict(zip(wins, masks))[owins[i]].cpu() - torch.from_numpy(g["mask_%d" % i])).abs().max()) <= 1e-4
