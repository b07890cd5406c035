"""ENet 2D encoder (reference lib/nets/enet.py:130-715, lib/nets/network.py:63-64,199-205).

Fixture `enet_cases.npz` holds what the reference's own `create_enet(41)`, split as `create_enet_for_3d` does, returns for
seeded weights / views (oracle/make_golden.py::enet_case).  CPU tests pin the oracle restatement and the product module
tree (state_dict contract, arithmetic); the GPU tests run the product encoder on PyTorch-ROCm and the whole config-4 forward
from RGB views."""
import numpy as np
import pytest
import torch

from sis3d import config, synthetic
from sis3d.nets import enet


def _shapes(g):
    return {str(k): tuple(int(d) for d in str(s).split(",")) if str(s) else () for k, s in zip(g["keys"], g["shapes"])}


def test_state_dict_contract_matches_reference(golden):
    g = golden("enet_cases")
    want = _shapes(g)
    got = {k: tuple(v.shape) for k, v in enet.create_enet(41).state_dict().items()}
    assert got == want                                       # every key and shape of the reference checkpoint
    # a reference-shaped checkpoint loads strict=True, before and after the 3D split
    sd = synthetic.synth_enet_state_dict(want, seed=0)
    m = enet.create_enet(41)
    m.load_state_dict(sd, strict=True)
    fixed, train, cls = enet.split_enet_for_3d(m)
    assert len(fixed) == 18 and len(train) == 8 and len(cls) == 1
    assert all(not p.requires_grad for p in fixed.parameters()) and all(p.requires_grad for p in train.parameters())
    assert sorted(cls.state_dict()) == ["0.0.weight"]


def test_oracle_enet_matches_reference_fixture(golden, oracle):
    g = golden("enet_cases")
    sd = synthetic.synth_enet_state_dict(_shapes(g), seed=0)
    small = synthetic.synth_images(7, 1, (64, 80))
    assert np.array_equal(oracle.enet_forward(sd, small, 0, 26).numpy(), g["small_out"])
    full = synthetic.synth_images(3, 2)
    mid = oracle.enet_forward(sd, full, 0, 18)
    assert np.array_equal(mid[:, :, ::4, ::4].numpy(), g["full_fixed_sub"])
    out = oracle.enet_forward(sd, mid, 18, 26)
    assert tuple(out.shape) == (2, 128, 32, 41)
    assert np.array_equal(out[:, :, ::4, ::4].numpy(), g["full_out_sub"])
    assert np.array_equal(oracle.enet_forward(sd, out, 26, 27)[:, :, ::4, ::4].numpy(), g["full_cls_sub"])


def test_product_enet_cpu_equals_oracle(golden, oracle):
    g = golden("enet_cases")
    sd = synthetic.synth_enet_state_dict(_shapes(g), seed=0)
    m = enet.create_enet(41)
    m.load_state_dict(sd)
    fixed, train, _ = enet.split_enet_for_3d(m)
    x = synthetic.synth_images(7, 1, (64, 80))
    with torch.no_grad():
        got = train.eval()(fixed.eval()(x))
    assert np.array_equal(got.numpy(), g["small_out"])       # same torch-CPU operators in the same order: bit-identical
    # torch7-style dropout: eval scales by (1 - p); train mode drops without the 1/(1-p) boost
    d = enet.ScaledDropout2d(0.1)
    assert torch.allclose(d.eval()(torch.ones(1, 4, 2, 2)), torch.full((1, 4, 2, 2), 0.9))
    t = d.train()(torch.ones(64, 64, 1, 1))
    assert set(np.unique(t.numpy()).round(4)) <= {0.0, 1.0}


def test_network_carries_the_encoder_under_the_reference_names(golden):
    """lib/nets/network.py:63-64: image_enet_fixed / image_enet_trainable / image_enet_classification"""
    g = golden("e2e_rgb_small")
    from sis3d.nets.backbones import state_dict_shapes
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_IMAGES_GT = True, False
    shapes = state_dict_shapes(c)
    assert sorted(shapes) == list(g["shapes_keys"])
    assert any(k.startswith("image_enet_trainable.7.") for k in shapes) and "image_enet_classification.0.0.weight" in shapes


def test_oracle_rgb_forward_matches_reference_fixture(golden, oracle):
    """whole TEST forward from RGB views (USE_IMAGES_GT=False): the reference ran its own ENet + projection + network"""
    g = golden("e2e_rgb_small")
    from sis3d.nets.backbones import state_dict_shapes
    dims = tuple(int(v) for v in g["dims"])
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_IMAGES_GT = True, False
    sd = synthetic.synth_checkpoint(state_dict_shapes(c), seed=0)
    cid = int(g["chunk_id"])
    data = synthetic.synth_chunk(cid, dims)
    _, i3d, i2d = synthetic.synth_views(cid, n_views=int(g["n_views"]), n_per_view=int(g["n_per_view"]), dims=dims)
    images = synthetic.synth_images(cid, int(g["n_views"]))
    o = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2)).forward(data, images, i3d, i2d)
    s = int(g["sub"])
    assert np.array_equal(o["level1"][0, :, ::s, ::s, ::s].numpy(), g["level1_sub"])
    assert np.array_equal(o["rpn_bbox_pred_level2"][0, ::s, ::s, ::s].numpy(), g["rpn_bbox_pred_level2_sub"])
    assert np.array_equal(o["rois"][0].numpy(), g["rois"]) and np.array_equal(o["cls_score"].numpy(), g["cls_score"])
    assert len(o["mask_pred"][0]) == int(g["n_masks"]) and np.array_equal(o["mask_pred"][0][0].numpy(), g["mask_0"])


def test_product_enet_vs_live_reference(oracle):
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    rh.install()
    from lib.nets import enet as renet
    torch.manual_seed(5)
    ref = renet.create_enet(41)                               # PyTorch default init, as SURVEY 8c prescribes
    g = torch.Generator().manual_seed(6)
    sd = ref.state_dict()
    for k, v in sd.items():                                   # non-trivial BatchNorm statistics
        if k.endswith("running_mean"):
            v.copy_(torch.rand(v.shape, generator=g) - 0.5)
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    mine = enet.create_enet(41)
    mine.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 256, 328, generator=g)
    n = len(ref)
    with torch.no_grad():
        want = torch.nn.Sequential(*[ref[i] for i in range(n - 9, n - 1)]).eval()(torch.nn.Sequential(*[ref[i] for i in range(n - 9)]).eval()(x))
        f, t, _ = enet.split_enet_for_3d(mine)
        got = t.eval()(f.eval()(x))
    assert torch.equal(got, want)
    assert torch.equal(oracle.enet_forward(sd, x, 0, 26), want)


# ------------------------------------------------------------------------------------------------------------- GPU
def _rel_err(got, want):
    return float((got - want).abs().max()) / max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_enet_on_gpu_five_views_vs_oracle(golden, oracle):
    """(V,3,256,328) -> (V,128,32,41) on PyTorch-ROCm against the CPU oracle: 1e-4 of the feature scale"""
    g = golden("enet_cases")
    sd = synthetic.synth_enet_state_dict(_shapes(g), seed=0)
    m = enet.create_enet(41)
    m.load_state_dict(sd)
    fixed, train, _ = enet.split_enet_for_3d(m)
    fixed.cuda().eval()
    train.cuda().eval()
    x = synthetic.synth_images(11, 5)
    with torch.no_grad():
        got = train(fixed(x.cuda())).cpu()
    want = oracle.enet_forward(sd, x, 0, 26)
    assert tuple(got.shape) == (5, 128, 32, 41)
    assert _rel_err(got, want) <= 1e-4
    full = synthetic.synth_images(3, 2)
    with torch.no_grad():
        sub = train(fixed(full.cuda())).cpu()[:, :, ::4, ::4]
    assert float((sub - torch.from_numpy(g["full_out_sub"])).abs().max()) <= 1e-4 * max(1.0, float(np.abs(g["full_out_sub"]).max()))


def _rgb_net(c):
    from sis3d.nets import backbones
    net = backbones.ScanNet_Backbone(cfg=c)
    net.init_modules()
    sd = synthetic.synth_checkpoint({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval(), sd


@pytest.mark.gpu
@pytest.mark.parametrize("dims,n_views,n_per_view,fixture", [((64, 32, 48), 3, 400, "e2e_rgb_small"), ((96, 48, 96), 5, 3000, None)])
def test_config4_forward_from_rgb(golden, oracle, dims, n_views, n_per_view, fixture):
    """BASELINE config 4 end to end: RGB views -> ENet -> back-projection -> colour/geometry backbone -> RPN -> heads,
    against the reference's own output (small fixture) and against the oracle at the full 96x48x96 chunk"""
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_IMAGES_GT, c.USE_MASK = True, False, False
    net, sd = _rgb_net(c)
    cid = 6 if fixture else 21
    data = synthetic.synth_chunk(cid, dims)
    _, i3d, i2d = synthetic.synth_views(cid, n_views=n_views, n_per_view=n_per_view, dims=dims)
    images = synthetic.synth_images(cid, n_views)
    blobs = {"data": data, "id": ["rgb"], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]],
             "nearest_images": {"images": [images]}, "proj_ind_3d": [i3d], "proj_ind_2d": [i2d]}
    p = net.forward(blobs, "TEST", [])
    torch.cuda.synchronize()
    o = oracle.OracleNet(sd, c, config.anchor_sizes(c, 1), config.anchor_sizes(c, 2)).forward(data, images, i3d, i2d)
    l1, l2 = net._net_conv
    scale = max(1.0, float(o["level1"].abs().max()))
    assert float((l1.cpu() - o["level1"]).abs().max()) <= 1e-4 * scale
    assert float((l2.cpu() - o["level2"]).abs().max()) <= 1e-4 * scale
    for lv in (1, 2):
        assert float((p["rpn_cls_prob_level%d" % lv].cpu() - o["rpn_cls_prob_level%d" % lv]).abs().max()) <= 1e-4
        assert float((p["rpn_bbox_pred_level%d" % lv].cpu() - o["rpn_bbox_pred_level%d" % lv]).abs().max()) <= 1e-4 * scale
    if fixture:
        g = golden(fixture)
        s = int(g["sub"])
        assert float((l1.cpu()[0, :, ::s, ::s, ::s] - torch.from_numpy(g["level1_sub"])).abs().max()) <= 1e-4 * scale


def test_folded_encoder_equals_module_tree_cpu():
    """nets/enet_folded.py (BatchNorm + eval-dropout scale folded into the convolutions) against the module tree it reads, with
    non-trivial BatchNorm statistics: equal to the rounding of the folded weights; re-folds when a parameter changes"""
    from sis3d.nets.enet_folded import FoldedEncoder
    torch.manual_seed(3)
    m = enet.create_enet(41)
    g = torch.Generator().manual_seed(4)
    for k, v in m.state_dict().items():
        if k.endswith("running_mean"):
            v.copy_(torch.rand(v.shape, generator=g) - 0.5)
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    fixed, train, _ = enet.split_enet_for_3d(m)
    fixed.eval()
    train.eval()
    x = torch.randn(2, 3, 64, 80, generator=g)
    enc = FoldedEncoder(fixed, train)
    with torch.no_grad():
        want, got = train(fixed(x)), enc(x)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
        fixed[2].weight.mul_(1.5)                                # a parameter update must invalidate the folded copy
        want2, got2 = train(fixed(x)), enc(x)
        assert float((want2 - want).abs().max()) > 1e-3 and float((got2 - want2).abs().max()) <= 1e-5 * max(1.0, float(want2.abs().max()))


# ------------------------------------------------------------------------------------------------- csrc/enet.hip executor
def _seeded_encoder(seed=0):
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "enet_cases.npz"))
    sd = synthetic.synth_enet_state_dict(_shapes(g), seed=seed)
    m = enet.create_enet(41)
    m.load_state_dict(sd)
    fixed, train, _ = enet.split_enet_for_3d(m)
    return sd, fixed.eval(), train.eval()


def test_hip_encoder_plan_reads_the_module_tree_cpu():
    """nets/enet_hip.py: the launch plan built from the reference-named modules (no GPU needed): 22 bottlenecks, the two stride-2
    blocks where enet.py has them, stages 2 and 3 = regular, dilated 2, asymmetric 5, dilated 4, regular, dilated 8, asymmetric 5,
    dilated 16; packed weights have the lane-order size; pack_pw is the documented permutation"""
    from sis3d.nets.enet_hip import HipEncoder, pack_pw, pack_taps
    sd, fixed, train = _seeded_encoder()
    plan = HipEncoder(fixed, train)._build()
    bl = plan["blocks"]
    assert len(bl) == 22
    assert [i for i, b in enumerate(bl) if b.down] == [0, 5]
    assert [(b.cin, b.c, b.mid) for b in bl[:6]] == [(16, 64, 16)] + [(64, 64, 16)] * 4 + [(64, 128, 32)]
    stage = [(0, 1), (0, 2), (1, 1), (0, 4), (0, 1), (0, 8), (1, 1), (0, 16)]
    assert [(b.kind, b.dil) for b in bl[6:]] == stage + stage
    for b in bl:
        assert b.w2.numel() == (5 if b.kind else 9) * b.mid * b.mid and b.w3.numel() == b.c * b.mid
        assert b.w1.numel() == (4 if b.down else 1) * b.mid * b.cin and (b.w2b is None) == (b.kind == 0)
    w = torch.arange(32 * 48, dtype=torch.float32).view(32, 48)
    p = pack_pw(w).view(2, 3, 64, 4)
    for ct, g_, lane, r in [(0, 0, 0, 0), (1, 2, 37, 3), (0, 1, 63, 1), (1, 0, 16, 2)]:
        assert float(p[ct, g_, lane, r]) == float(w[16 * ct + (lane & 15), 16 * g_ + 4 * (lane >> 4) + r])
    w4 = torch.randn(16, 16, 2, 2)
    assert torch.equal(pack_taps(w4).view(4, -1)[3], pack_pw(w4[:, :, 1, 1]))
    assert plan["init"][0].shape == (13, 3, 3, 3) and plan["init"][2].numel() == 3


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 3, 256, 328), (3, 3, 40, 56), (1, 3, 64, 80)])
def test_hip_encoder_vs_oracle_and_operators(oracle, shape):
    """csrc/enet.hip (one launch per bottleneck) against the CPU oracle (1e-4 of the feature scale, the bound of the operator path) and
    against the same folded weights on PyTorch-ROCm operators; 105- and 80-pixel maps exercise the partial last tile and the borders of
    the dilated / asymmetric convolutions (dilation 16 on an 5 x 7 map: only the centre tap is inside)"""
    from sis3d.nets.enet_folded import FoldedEncoder
    from sis3d.nets.enet_hip import HipEncoder
    sd, fixed, train = _seeded_encoder(seed=1)
    fixed.cuda()
    train.cuda()
    x = synthetic.synth_images(7, shape[0])[:, :, :shape[2], :shape[3]].contiguous()
    assert tuple(x.shape) == shape
    with torch.no_grad():
        got = HipEncoder(fixed, train)(x.cuda())
        ops_ = FoldedEncoder(fixed, train)(x.cuda())
    torch.cuda.synchronize()
    want = oracle.enet_forward(sd, x, 0, 26)
    assert tuple(got.shape) == tuple(want.shape) == (shape[0], 128, shape[2] // 8, shape[3] // 8) and got.is_contiguous()
    e_o, e_t = _rel_err(got.cpu(), want), _rel_err(got, ops_)
    print("[enet hip] %s: vs oracle %.2e, vs PyTorch-ROCm operators %.2e (feature scale %.2f)" % (shape, e_o, e_t, float(want.abs().max())))
    assert e_o <= 1e-4 and e_t <= 1e-4


@pytest.mark.gpu
def test_network_image_features_runs_the_hip_encoder():
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_IMAGES_GT, c.USE_MASK = True, False, False
    net, _ = _rgb_net(c)
    x = synthetic.synth_images(5, 2).cuda()
    a = net.image_features(x)
    assert net._enet_hip is not None and getattr(net, "_enet_folded", None) is None
    net.enet_impl = "folded"
    b = net.image_features(x)
    assert net._enet_folded is not None and _rel_err(a, b) <= 1e-4


def test_image_features_rejects_cpu_images_on_the_hip_route():
    """the default route of Network.image_features is csrc/enet.hip: a CPU tensor is an error, not a silent run on CPU operators"""
    from sis3d import _lib
    from sis3d.nets import backbones
    c = config.scannet_benchmark_cfg()
    c.USE_IMAGES, c.USE_IMAGES_GT, c.USE_MASK = True, False, False
    net = backbones.ScanNet_Backbone(cfg=c)
    net.init_modules()
    net.eval()
    with pytest.raises(_lib.Sis3dError):
        net.image_features(torch.zeros(1, 3, 64, 80))
    net.enet_impl = "folded"                                     # an explicit choice of the operator path runs wherever the tensors are
    assert tuple(net.image_features(torch.zeros(1, 3, 64, 80)).shape) == (1, 128, 8, 10)
