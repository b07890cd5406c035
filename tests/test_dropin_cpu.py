"""The drop-in boundary against the REAL reference tree (only where /root/reference is mounted; no GPU compute):
after sis3d.dropin.install() the reference's own import paths resolve to the HIP implementations, the
reference's construction sequence (lib/model/trainval.py:68-71) works on our classes, and a state_dict saved by
the reference's network loads into ours key-for-key."""
import pytest
import torch


def test_install_into_reference_tree():
    import ref_harness as rh
    if not rh.available():
        pytest.skip("reference tree not present on this machine")
    ns = rh.install()                                   # stubs for easydict/ipdb/... + cfg_from_file(benchmark.yml)
    cfg = ns.cfg
    cfg.USE_IMAGES = False
    cfg.USE_IMAGES_GT = True
    ref_cls = ns.backbones.ScanNet_Backbone
    with rh.in_reference_dir():
        torch.manual_seed(0)
        ref_net = ref_cls()
        ref_net.init_modules()
    ref_sd = ref_net.state_dict()

    from sis3d import dropin, _lib
    undo = dropin.install(cfg)
    try:
        import lib.nets.backbones as r_bb
        import lib.nets.network as r_net
        import lib.layer_utils.nms_wrapper as r_nmsw
        import lib.layer_utils.proposal_layer as r_prop
        from sis3d.layer_utils.roi_pooling.roi_pool import RoIPoolFunction
        assert r_net.RoIPoolFunction is RoIPoolFunction
        assert r_nmsw.nms is r_prop.nms
        # trainval.py:68-71:  net = getattr(backbones, cfg.NET)(); net.init_modules(); net.load_state_dict(...)
        net = getattr(r_bb, cfg.NET)()
        assert type(net).__name__ == "ScanNet_Backbone" and net.cfg is cfg
        net.init_modules()
        missing, unexpected = net.load_state_dict(ref_sd, strict=True)
        assert not missing and not unexpected
        assert hasattr(net, "mask_backbone") and hasattr(net, "_predictions") and hasattr(net, "delete_intermediate_states")
        # the cffi-shaped entry points exist with the reference's argument lists
        from lib.layer_utils.nms._ext import nms as ext_nms
        from lib.layer_utils.roi_pooling._ext import roi_pooling as ext_roi
        assert callable(ext_nms.gpu_nms) and callable(ext_roi.roi_pooling_forward_cuda)
        # no silent CPU fallback behind the boundary
        with pytest.raises(_lib.Sis3dError):
            r_nmsw.nms(torch.zeros(3, 6), 0.1)
        with pytest.raises(NotImplementedError):
            net.forward({"data": torch.zeros(1, 2, 8, 8, 8), "id": ["x"]}, "TRAIN", [])
    finally:
        dropin.uninstall(undo)
        import sys
        for k in [k for k in sys.modules if k.startswith("lib.layer_utils.nms._ext") or k.startswith("lib.layer_utils.roi_pooling._ext")]:
            del sys.modules[k]
        rh._installed = False
        rh.install()                                     # restore the harness's own stubs
