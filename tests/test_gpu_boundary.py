"""The drop-in boundary called BY THE REFERENCE'S NAMES AND SIGNATURES on the GPU:
`nms(dets, thresh)`, `RoIPoolFunction(pw, ph, pl, scale)(features, rois)` + `.argmax/.rois/.feature_size`,
`Projection.apply(label, i3d, i2d, dims)`, the 15-argument `proposal_layer(...)`, and the two cffi entry points with
caller-allocated outputs `gpu_nms(keep, num_out, boxes, thresh)` / `roi_pooling_forward_cuda(...)`.  Integer outputs are
bit-exact against the oracle; fixtures are the reference's own outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sis3d import config, synthetic  # noqa: E402


def _boxes(n, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(n, 3, generator=g) * torch.tensor([80.0, 40.0, 80.0])
    return torch.cat([lo, lo + torch.rand(n, 3, generator=g) * 30.0 + 1.0], 1)


def test_nms_wrapper_reference_signature(oracle, golden):
    from sis3d.layer_utils.nms_wrapper import nms                     # lib/layer_utils/nms_wrapper.py:7
    boxes = _boxes(400, 1)
    keep = nms(boxes.cuda(), 0.1)
    assert keep.dtype == torch.int64 and keep.is_cuda and torch.equal(keep.cpu(), oracle.nms(boxes, 0.1))
    g = golden("nms_cases")
    name = sorted({k.split("/")[0] for k in g.files})[0]
    assert np.array_equal(nms(torch.from_numpy(g[name + "/boxes"]).cuda(), 0.35).cpu().numpy(), g[name + "/keep_0.35"])
    with pytest.raises(Exception):
        nms(boxes, 0.1)                                                # CPU tensors: no fallback path


def test_gpu_nms_cffi_signature_caller_allocated(oracle):
    from sis3d.dropin import gpu_nms                                   # lib/layer_utils/nms/src/nms_cuda.h
    boxes = _boxes(1025, 2)
    keep = torch.full((boxes.shape[0],), -1, dtype=torch.int64)        # caller-allocated CPU LongTensors, as pth_nms.py:58-62
    num_out = torch.zeros(1, dtype=torch.int64)
    assert gpu_nms(keep, num_out, boxes.cuda(), 0.3) == 1
    want = oracle.nms(boxes, 0.3)
    n = int(num_out[0])
    assert n == want.numel() and torch.equal(keep[:n], want) and bool((keep[n:] == -1).all())


@pytest.mark.parametrize("layout", ["ncdhw", "channels_last"])
def test_roi_pool_function_reference_call_form(oracle, layout):
    from sis3d.layer_utils.roi_pooling.roi_pool import RoIPoolFunction  # lib/layer_utils/roi_pooling/roi_pool.py:9-38
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(1, 128, 24, 12, 24, generator=g)
    rois = _boxes(37, 4)
    f = feat.cuda()
    if layout == "channels_last":
        f = f.contiguous(memory_format=torch.channels_last_3d)
    fn = RoIPoolFunction(4, 4, 4, 0.25)                                 # network.py:511: RoIPoolFunction(ps, ps, ps, 1/stride)
    out = fn(f, rois.cuda())
    want, warg = oracle.roi_pool(feat, rois, (4, 4, 4), 0.25, want_argmax=True)
    assert tuple(out.shape) == (37, 128, 4, 4, 4) and torch.equal(out.cpu(), want)
    assert fn.argmax.dtype == torch.int32 and torch.equal(fn.argmax.cpu(), warg)
    assert fn.rois is not None and tuple(fn.feature_size) == (1, 128, 24, 12, 24)
    gin, grois = fn.backward(out)                                       # roi_pool.py:40-50: (grad_input, zeros for the rois)
    assert tuple(gin.shape) == tuple(feat.shape) and tuple(grois.shape) == tuple(rois.shape)
    want_g = oracle.roi_pool_backward(want, warg, feat.shape)
    assert float((gin.cpu() - want_g).abs().max()) <= 1e-4 * float(want_g.abs().max())


def test_roi_pooling_forward_cuda_cffi_signature(oracle):
    from sis3d.dropin import roi_pooling_forward_cuda                   # lib/layer_utils/roi_pooling/src/roi_pooling_cuda.h
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(1, 32, 12, 6, 12, generator=g)
    rois = _boxes(9, 6) * 0.5
    output = torch.zeros(9, 32, 4, 4, 4, device="cuda")                 # caller-allocated, roi_pool.py:26-30
    argmax = torch.zeros(9, 32, 4, 4, 4, dtype=torch.int32, device="cuda")
    assert roi_pooling_forward_cuda(4, 4, 4, 0.25, feat.cuda(), rois.cuda(), output, argmax) == 1
    want, warg = oracle.roi_pool(feat, rois, (4, 4, 4), 0.25, want_argmax=True)
    assert torch.equal(output.cpu(), want) and torch.equal(argmax.cpu(), warg)


def test_projection_apply_reference_signature(oracle, golden):
    from sis3d.layer_utils.projection import Projection                 # lib/layer_utils/projection.py:124-136
    g = golden("projection_cases")
    feats, i3d, i2d = torch.from_numpy(g["feats"]), torch.from_numpy(g["i3d"]), torch.from_numpy(g["i2d"])
    dims = tuple(int(v) for v in g["dims"])
    for v in range(feats.shape[0]):
        got = Projection.apply(feats[v].cuda(), i3d[v].cuda(), i2d[v].cuda(), dims)
        assert tuple(got.shape) == (feats.shape[1], dims[2], dims[1], dims[0])
        assert torch.equal(got.cpu(), oracle.projection(feats[v], i3d[v], i2d[v], dims))
        assert np.array_equal(got.cpu().numpy(), g["out"][v])              # the reference's own Projection.apply output
    got2 = Projection.apply(feats[0, 0].cuda(), i3d[0].cuda(), i2d[0].cuda(), dims)     # (h,w) label image form
    assert np.array_equal(got2.cpu().numpy(), g["out2d"])


def test_proposal_layer_fifteen_argument_signature(oracle):
    """lib/layer_utils/proposal_layer.py:11-15, called as network.py:551-563 calls it, on the ORACLE's RPN maps: the
    sorted order / keep list are integer outputs -> the returned lists equal the oracle's row for row"""
    from sis3d.layer_utils.proposal_layer import proposal_layer
    from sis3d.layer_utils.generate_anchors import generate_anchors
    cfg = config.scannet_benchmark_cfg()
    cfg.USE_MASK = False
    from sis3d.nets import backbones
    shapes = backbones.state_dict_shapes(cfg)
    sd = synthetic.synth_state_dict(shapes, seed=0, gains=synthetic.DEFAULT_GAINS)
    dims = (64, 32, 48)
    data = synthetic.synth_chunk(3, dims)
    o = oracle.OracleNet(sd, cfg, config.anchor_sizes(cfg, 1), config.anchor_sizes(cfg, 2)).forward(data)
    fs = tuple(d // 4 for d in dims)
    a1, a2, a3 = generate_anchors(fs, fs, fs, [4, 4, 4], cfg)
    rois, scores, levels = proposal_layer(o["rpn_cls_prob_level1"].cuda(), o["rpn_bbox_pred_level1"].cuda(), a1,
                                          o["rpn_cls_prob_level2"].cuda(), o["rpn_bbox_pred_level2"].cuda(), a2,
                                          None, None, a3, dims, "TEST", None, None, None, cfg=cfg)
    assert len(rois) == len(scores) == len(levels) == 1
    assert tuple(scores[0].shape) == (rois[0].shape[0], 1)
    assert rois[0].shape[0] == o["rois"][0].shape[0]
    assert float((rois[0].cpu() - o["rois"][0]).abs().max()) <= 1e-4            # decode: expf vs exp, <= 1 ulp-class
    assert torch.equal(scores[0].cpu(), o["roi_scores"][0]) and torch.equal(levels[0].cpu(), o["level_inds"][0])
