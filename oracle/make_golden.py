"""Generate tests/golden/*.npz FROM THE REFERENCE ITSELF (run in the build
container, where /root/reference exists):

    python oracle/make_golden.py

Test infrastructure.  Inputs come from the product's seeded synthetic helpers
(`sis3d.synthetic`); outputs are whatever the reference's own code returns for
them (imported in place by oracle/ref_harness.py, RoI pooling by the reference's
roi_pooling.c built into oracle/_ref/).  The fixtures are what pins the oracle
(tests/test_oracle_pinning.py) and, through it, the HIP path.
"""
import hashlib
import os
import subprocess
import tempfile
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "3d-sis_amd"))

import ref_harness as rh                       # noqa: E402
from sis3d import synthetic                    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def nms_cases(ns):
    g = torch.Generator().manual_seed(7)
    cases = {}
    for n in (1, 2, 63, 64, 65, 130, 400):
        lo = torch.rand(n, 3, generator=g) * torch.tensor([90.0, 44.0, 90.0])
        sz = torch.rand(n, 3, generator=g) * 30.0 + 1.0
        boxes = torch.cat([lo, lo + sz], 1)
        cases["rand%d" % n] = boxes
    # adversarial: duplicates, nested, touching (+1 convention makes touching boxes overlap)
    base = torch.tensor([[10.0, 10, 10, 20, 20, 20]])
    adv = torch.cat([base, base, base + 0.5, torch.tensor([[20.0, 10, 10, 30, 20, 20]]),
                     torch.tensor([[21.0, 10, 10, 31, 20, 20]]), torch.tensor([[12.0, 12, 12, 18, 18, 18]]),
                     torch.tensor([[0.0, 0, 0, 96, 48, 96]]), torch.tensor([[5.0, 5, 5, 5, 5, 5]]),
                     torch.tensor([[5.0, 5, 5, 5, 5, 5]])], 0)
    cases["adversarial"] = adv
    # integer-grid boxes produce exact IoU ties around simple fractions
    gi = torch.randint(0, 12, (200, 3), generator=g).float() * 4
    cases["grid200"] = torch.cat([gi, gi + torch.randint(1, 6, (200, 3), generator=g).float() * 4], 1)
    out = {}
    for name, b in cases.items():
        for th in (0.1, 0.35, 0.5):
            keep = ns.pth_nms.cpu_nms(b.numpy(), th)          # the reference's own numpy NMS
            out["%s/boxes" % name] = b.numpy()
            out["%s/keep_%g" % (name, th)] = np.asarray(keep, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "nms_cases.npz"), **out)
    print("nms_cases", len(out))


def roi_cases(ns):
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(1, 8, 12, 6, 10, generator=g)
    lo = torch.rand(24, 3, generator=g) * torch.tensor([44.0, 20.0, 36.0])
    sz = torch.rand(24, 3, generator=g) * 24.0
    rois = torch.cat([lo, lo + sz], 1)
    rois[0] = torch.tensor([0.0, 0, 0, 48, 24, 40])          # whole map
    rois[1] = torch.tensor([4.0, 4, 4, 4, 4, 4])             # degenerate -> forced 1x1x1
    rois[2] = torch.tensor([40.0, 20, 36, 60, 30, 50])       # runs past the border
    rois[3] = torch.tensor([60.0, 30, 50, 70, 40, 60])       # fully outside -> empty bins -> 0
    rois[4] = torch.tensor([3.9999, 7.5, 0.25, 12.0001, 8.5, 39.75])
    ref = rh.ref_roi_pool_c().forward(4, 4, 4, 0.25, feat, rois)       # reference roi_pooling.c
    # reference pure-Python RoIPool (roi_pool.py:53-120) on a subset (slow) for argmax coordinates
    rp = ns.roi_pool.RoIPool.__new__(ns.roi_pool.RoIPool)
    rp.pooled_width = rp.pooled_height = rp.pooled_length = 2
    rp.spatial_scale = 0.25
    py = ns.roi_pool.RoIPool.forward(rp, feat, rois[:6])
    np.savez_compressed(os.path.join(OUT, "roi_pool_cases.npz"), feat=feat.numpy(), rois=rois.numpy(),
                        out_c_4=ref.numpy(), out_py_2=py.numpy(), argmax_whl_py_2=rp.remember_for_backward.numpy())
    print("roi_pool_cases", ref.shape, py.shape)


def roi_backward_case(ns):
    """the reference's Python RoIPool (roi_pool.py:52-199): forward + backward on a small map; its backward accumulates
    grad_output through the remembered arg-max positions in (roi, c, pw, ph, pl) order"""
    from lib.layer_utils.roi_pooling.roi_pool import RoIPool
    g = torch.Generator().manual_seed(77)
    feat = torch.randn(1, 6, 9, 7, 8, generator=g)
    lo = torch.rand(7, 3, generator=g) * torch.tensor([20.0, 14.0, 16.0])
    rois = torch.cat([lo, lo + torch.rand(7, 3, generator=g) * 18.0 + 1.0], 1)
    rp = RoIPool(2, 2, 2, 0.25)
    out = rp.forward(feat, rois)
    gout = torch.randn(out.shape, generator=g)
    gin = rp.backward(gout)
    np.savez_compressed(os.path.join(OUT, "roi_pool_backward_case.npz"), feat=feat.numpy(), rois=rois.numpy(), out=out.numpy(),
                        grad_out=gout.numpy(), grad_in=gin.numpy())
    print("roi_pool_backward_case", float(gin.abs().sum()))


def projection_cases(ns):
    dims = (12, 6, 10)
    feats, i3d, i2d = synthetic.synth_views(3, n_views=3, n_per_view=150, channels=5, image_hw=(8, 9), dims=dims)
    outs = [ns.projection.Projection.apply(f, a, b, dims).numpy() for f, a, b in zip(feats, i3d, i2d)]
    lab2d = feats[0, 0]
    out2d = ns.projection.Projection.apply(lab2d, i3d[0], i2d[0], dims).numpy()
    np.savez_compressed(os.path.join(OUT, "projection_cases.npz"), feats=feats.numpy(), i3d=i3d.numpy(), i2d=i2d.numpy(),
                        dims=np.array(dims), out=np.stack(outs), out2d=out2d)
    print("projection_cases")


def projection_backward_case(ns):
    """Projection.backward of the reference (projection.py:139-153) run in place under ref_harness.legacy_data_alias (torch-0.4
    `.data` aliasing + `saved_variables`): volumes with >= 32*41 voxels (below that the reference's resize_ GROWS the clone's
    storage and reads uninitialised memory -- undefined, not pinned), duplicate pixels (the last list entry wins), an empty list."""
    out = {}
    cases = (("a", (24, 12, 20), 900, 3, 7), ("b", (32, 16, 24), 3000, 6, 8), ("empty", (16, 10, 12), 0, 2, 9))
    for name, dims, n, C, seed in cases:
        feats, i3d, i2d = synthetic.synth_views(seed, n_views=1, n_per_view=n, channels=C, image_hw=(32, 41), dims=dims)
        g = torch.randn(C, dims[2], dims[1], dims[0], generator=torch.Generator().manual_seed(seed))
        fwd, gl = rh.ref_projection_backward(ns, feats[0], i3d[0], i2d[0], dims, g)
        assert tuple(gl.shape) == (C, 32, 41)
        pix = i2d[0, 1:1 + int(i3d[0, 0])]
        out.update({name + "_dims": np.array(dims), name + "_label": feats[0].numpy(), name + "_i3d": i3d[0].numpy(),
                    name + "_i2d": i2d[0].numpy(), name + "_grad_out": g.numpy(), name + "_grad_label": gl.numpy(),
                    name + "_forward_sha": np.array(sha(fwd))})
        print("projection_backward", name, dims, "entries", int(i3d[0, 0]), "distinct pixels", int(pix.unique().numel()))
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "projection_backward_case.npz"), **out)


def anchors_case(ns):
    with rh.in_reference_dir():
        a1, a2, _ = ns.generate_anchors.generate_anchors([3, 2, 4], [3, 2, 4], [], [4, 4, 4])
        f1, f2, _ = ns.generate_anchors.generate_anchors([24, 12, 24], [24, 12, 24], [], [4, 4, 4])
    np.savez_compressed(os.path.join(OUT, "anchors.npz"), small_l1=a1, small_l2=a2,
                        full_l1_sha=np.array(sha(f1)), full_l2_sha=np.array(sha(f2)),
                        full_l1_head=f1[:64], full_l2_tail=f2[-64:])
    print("anchors", f1.shape, f2.shape)


def split_state_dict(shapes, seed=0):
    return synthetic.synth_checkpoint(shapes, seed=seed)


def enet_case(ns):
    """the reference's ENet (lib/nets/enet.py create_enet, split as create_enet_for_3d :701-705) with seeded weights:
    fixed -> trainable features on a small view (full output) and on two full-size views (every 4th pixel), plus the key /
    shape list of its checkpoint (the state_dict contract)"""
    from lib.nets import enet as renet
    ref = renet.create_enet(41)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ref.load_state_dict(synthetic.synth_enet_state_dict(shapes, seed=0))
    ref.eval()
    n = len(ref)
    fixed = torch.nn.Sequential(*[ref[i] for i in range(n - 9)])
    train = torch.nn.Sequential(*[ref[i] for i in range(n - 9, n - 1)])
    small = synthetic.synth_images(7, 1, (64, 80))
    full = synthetic.synth_images(3, 2)
    with torch.no_grad():
        f_small = train(fixed(small))
        mid_full = fixed(full)
        f_full = train(mid_full)
        cls_full = ref[n - 1](f_full)
    keys = sorted(shapes)
    np.savez_compressed(os.path.join(OUT, "enet_cases.npz"), keys=np.array(keys),
                        shapes=np.array([",".join(str(d) for d in shapes[k]) for k in keys]),
                        small_out=f_small.numpy(), full_fixed_sub=mid_full[:, :, ::4, ::4].numpy(), full_out_sub=f_full[:, :, ::4, ::4].numpy(),
                        full_out_sha=np.array(sha(f_full.numpy())), full_cls_sub=cls_full[:, :, ::4, ::4].numpy())
    print("enet", tuple(f_full.shape), "max %.3f" % float(f_full.abs().max()),
          "%.1f KB" % (os.path.getsize(os.path.join(OUT, "enet_cases.npz")) / 1024))


def e2e(ns, name, use_images, dims, chunk_id, n_views=5, n_per_view=3000, sub=3, only_images=False, rgb=False, mask_images=None):
    """rgb: the views are RGB images and the reference runs its own ENet on them (USE_IMAGES_GT=False, network.py:203-205);
    mask_images: 'use' / 'only' = MASK_USE_IMAGES / + MASK_ONLY_IMAGES (backbones.py:253-284); only the mask-related arrays
    are stored for those"""
    ns.cfg.ONLY_IMAGES = bool(only_images)
    ns.cfg.MASK_USE_IMAGES = mask_images in ("use", "only")
    ns.cfg.MASK_ONLY_IMAGES = mask_images == "only"
    ckpt = None
    if rgb:
        from lib.nets import enet as renet
        ckpt = os.path.join(tempfile.mkdtemp(), "enet_init.pth")
        torch.save(renet.create_enet(int(ns.cfg.NUM_2D_CLASSES)).state_dict(), ckpt)     # overwritten by the seeded weights below
    try:
        net = rh.build_net(ns, seed=0, use_images=use_images, use_mask=True, enet_ckpt=ckpt)
    finally:
        pass
    shapes = {k: v.shape for k, v in net.state_dict().items()}
    sd = split_state_dict(shapes, seed=0)
    net.load_state_dict(sd)
    data = synthetic.synth_chunk(chunk_id, dims)
    feats = i3d = i2d = None
    if use_images:
        feats, i3d, i2d = synthetic.synth_views(chunk_id, n_views=n_views, n_per_view=n_per_view, dims=dims)
        if rgb:
            feats = synthetic.synth_images(chunk_id, n_views)
    grabbed = {}
    bb = net._backbone

    def grab():
        r = bb()
        grabbed["l1"], grabbed["l2"] = r[0], r[1]
        return r
    net._backbone = grab
    p = rh.forward(ns, net, rh.make_blobs(data, feats, i3d, i2d))
    out = dict(num_classes=np.array(int(ns.cfg.NUM_CLASSES)), dims=np.array(dims), chunk_id=np.array(chunk_id), n_views=np.array(n_views), n_per_view=np.array(n_per_view),
               sub=np.array(sub), shapes_keys=np.array(sorted(shapes.keys())),
               level1_sub=grabbed["l1"][0, :, ::sub, ::sub, ::sub].numpy(), level2_sub=grabbed["l2"][0, :, ::sub, ::sub, ::sub].numpy(),
               level1_sha=np.array(sha(grabbed["l1"].numpy())), level2_sha=np.array(sha(grabbed["l2"].numpy())))
    for lv in (1, 2):
        out["rpn_cls_score_level%d_sub" % lv] = p["rpn_cls_score_level%d" % lv][0, :, ::sub, ::sub, ::sub].numpy()
        out["rpn_cls_prob_level%d_sub" % lv] = p["rpn_cls_prob_level%d" % lv][0, :, ::sub, ::sub, ::sub].numpy()
        out["rpn_bbox_pred_level%d_sub" % lv] = p["rpn_bbox_pred_level%d" % lv][0, ::sub, ::sub, ::sub].numpy()
    out["rois"] = p["rois"][0].numpy()
    out["roi_scores"] = p["roi_scores"][0].numpy()
    out["level_inds"] = p["level_inds"][0].numpy()
    for k in ("cls_score", "cls_pred", "cls_prob", "bbox_pred"):
        out[k] = p[k].numpy()
    masks = p["mask_pred"][0]
    out["n_masks"] = np.array(len(masks))
    for i, m in enumerate(masks[:4]):
        out["mask_%d" % i] = m.numpy()
    out["mask_shapes"] = np.array([list(m.shape[2:]) for m in masks]).reshape(-1, 3)
    if use_images:
        ift = net._imageft           # logical (1,C,X,Y,Z), memory (C,Z,Y,X)
        nz = ift[0].abs().sum(0).nonzero()
        out["imageft_nz_xyz"] = nz.numpy().astype(np.int32)
        out["imageft_nz_val"] = ift[0][:, nz[:, 0], nz[:, 1], nz[:, 2]].numpy()
        out["imageft_stride"] = np.array(ift.stride())
    ns.cfg.ONLY_IMAGES = False
    ns.cfg.USE_IMAGES_GT = True
    ns.cfg.MASK_USE_IMAGES = ns.cfg.MASK_ONLY_IMAGES = False
    if mask_images:
        keep = ("num_classes", "dims", "chunk_id", "n_views", "n_per_view", "sub", "shapes_keys", "rois", "cls_pred", "cls_prob", "bbox_pred",
                "n_masks", "mask_shapes", "mask_0", "mask_1", "mask_2", "mask_3")
        out = {k: v for k, v in out.items() if k in keep}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "R=%d" % out["rois"].shape[0], "masks=%d" % len(masks),
          "%.1f KB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))


def compute_projection_cases(ns):
    """ProjectionHelper.compute_projection of the reference (under torch-0.4 integer division, ref_harness) on seeded
    camera rigs; lists stored up to the count (the reference's tail is uninitialised memory)."""
    cfg = ns.cfg
    out = {}
    for name, dims, cid, nv in (("small", (40, 24, 32), 1, 6), ("odd", (31, 70, 17), 2, 4), ("chunk", (96, 48, 96), 3, 5)):
        depth, c2w, w2g = synthetic.synth_cameras(cid, nv, dims, cfg.VOXEL_SIZE, (cfg.DEPTH_SHAPE[1], cfg.DEPTH_SHAPE[0]))
        if name == "small":
            depth[4] = 0.0                                    # no valid depth   -> None (projection.py:105-107)
            c2w[5, :3, 2] = -c2w[5, :3, 2]                    # looks away       -> None (:75-77 or :95-97)
            c2w[5, :3, 0] = -c2w[5, :3, 0]
        counts = []
        for v in range(nv):
            r = rh.ref_compute_projection(ns, depth[v], c2w[v], w2g[v], dims)
            n = 0 if r is None else int(r[0][0])
            counts.append(n)
            out["%s_l3_%d" % (name, v)] = (r[0][1:1 + n] if n else torch.zeros(0)).numpy().astype(np.int32)
            out["%s_l2_%d" % (name, v)] = (r[1][1:1 + n] if n else torch.zeros(0)).numpy().astype(np.int32)
        out[name + "_dims"] = np.array(dims)
        out[name + "_depth"] = depth.numpy()
        out[name + "_c2w"] = c2w.numpy()
        out[name + "_w2g"] = w2g.numpy()
        out[name + "_counts"] = np.array(counts)
        print("compute_projection", name, counts)
    np.savez_compressed(os.path.join(OUT, "compute_projection_cases.npz"), **out)


def benchmark_case(ns_unused):
    """SolverWrapper.benchmark of the reference (trainval.py:634-767) on a small synthetic scene, geometry-only and
    with colour from depth maps + poses (one view sees nothing -> killing_inds)."""
    import tempfile
    ns = rh.install(with_trainval=True)
    dims = (48, 24, 40)
    out = {"dims": np.array(dims)}
    for tag, use_images in (("geo", False), ("img", True)):
        net = rh.build_net(ns, seed=0, use_images=use_images, use_mask=True)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(synthetic.synth_state_dict(shapes, seed=3, gains=synthetic.DEFAULT_GAINS))
        data = synthetic.synth_chunk(9, dims)
        blobs = rh.make_blobs(data, scene_id="/data/scenes/scene0707_00__0.scene")
        if use_images:
            depth, c2w, w2g = synthetic.synth_cameras(5, 4, dims, ns.cfg.VOXEL_SIZE)
            depth[2] = 0
            feats = torch.randn(4, 128, 32, 41, generator=torch.Generator().manual_seed(1))
            blobs["nearest_images"] = {"images": [feats], "depths": [depth], "poses": [c2w], "world2grid": [w2g]}
        thresh = 0.975 if use_images else 0.5             # 0.975: about half of the detections fall under the keep rule
        old, ns.cfg.CLASS_THRESH = ns.cfg.CLASS_THRESH, thresh
        try:
            with tempfile.TemporaryDirectory() as td:
                r = rh.ref_benchmark(ns, net, [blobs], td)["scene0707_00"]
        finally:
            ns.cfg.CLASS_THRESH = old
        out[tag + "_class_thresh"] = np.array(thresh)
        for k in ("pred_class", "pred_conf", "pred_box"):
            out[tag + "_" + k] = r[k]
        out[tag + "_keep"] = np.array(r["pred_mask_index"])
        out[tag + "_scene_sha"] = np.array(sha(r["scene"]))
        out[tag + "_mask_shapes"] = np.array([m.shape for m in r["pred_mask"]]).reshape(-1, 3)
        out[tag + "_mask_bits"] = np.packbits(np.concatenate([m.reshape(-1) for m in r["pred_mask"]]).astype(np.uint8))
        print("benchmark", tag, "R=%d kept=%d" % (len(r["pred_class"]), int(out[tag + "_keep"].sum())))
    np.savez_compressed(os.path.join(OUT, "benchmark_small.npz"), **out)


def dataset_case(ns_unused):
    """Dataset.__getitem__ of the reference (lib/datasets/dataset.py:45-218) on a synthetic .chunk container written by
    sis3d.datasets.scene_file.write_scene_file (committed as tests/golden/synthetic.chunk + synthetic_labels.csv)."""
    from sis3d.datasets import scene_file
    ns = rh.install(with_trainval=True)
    from lib.datasets.dataset import Dataset as RefDataset
    g = np.random.default_rng(5)
    dims = (20, 52, 12)                                            # Y > 48: exercises the chunk-mode height crop
    sdf = (g.standard_normal(dims) * 3).astype(np.float32)
    sdf[g.random(dims) < 0.2] = -np.inf                            # unobserved voxels
    sdf[0, 0, 0], sdf[1, 0, 0], sdf[2, 0, 0] = -1.0, 3.0, -3.0      # boundaries of `> -1` and the clamp
    boxes = np.array([[1.5, 2.2, 0.7, 9.1, 20.9, 7.5], [-3.0, 4.0, 2.0, 6.0, 30.0, 9.0], [4.0, 40.2, 1.0, 12.0, 51.5, 8.0],
                      [10.2, 1.0, 3.0, 18.8, 12.0, 11.0], [2.0, 2.0, 2.0, 5.0, 5.0, 5.0]], dtype=np.float32)
    labels = [3, 5, 7, 4, 38]                                       # 38: zero-weight class in the synthetic map
    masks = []
    for b, lab in zip(boxes, labels):
        md = tuple(int(np.ceil(b[3 + k]) - np.floor(b[k])) for k in range(3))
        m = g.integers(0, 2, md).astype(np.uint16)
        m.reshape(-1)[:3] = [2, 257, 256]                           # > 1 cleared; 257 wraps to 1, 256 to 0 (uint8 cast)
        masks.append((lab, m))
    part = [1.0, 0.4, 1.0, 0.97, 1.0]
    w2c = np.eye(4, dtype=np.float32) * 21.333
    w2c[:3, 3] = [3.0, -2.0, 8.5]
    w2c[3, 3] = 1
    path = os.path.join(OUT, "synthetic.chunk")
    scene_file.write_scene_file(path, sdf, boxes, labels, masks, part, w2c, [17, 420, 9000])
    csv_path = os.path.join(OUT, "synthetic_labels.csv")
    with open(csv_path, "w") as f:
        f.write("nyu40id,nyu40class,mappedIdConsecutive,weight\n")
        for nyu, cons, w in ((3, 1, 1.5), (4, 2, 0.8), (5, 3, 2.0), (7, 4, 1.1), (38, 5, 0.0)):
            f.write("%d,c%d,%d,%g\n" % (nyu, nyu, cons, w))
    lst = os.path.join(OUT, "synthetic_filelist.txt")
    with open(lst, "w") as f:
        f.write("synthetic.chunk\n")
    out = {}
    cfg = ns.cfg
    saved = (cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH)
    try:
        cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK = csv_path, False, True
        for mode, keep in (("chunk", 0.5), ("benchmark", 1.0), ("scene", 0.0)):
            cfg.KEEP_THRESH = keep
            old = os.getcwd()
            os.chdir(OUT)
            try:
                ds = RefDataset(lst, mode)
                r = ds[0]
            finally:
                os.chdir(old)
            out[mode + "_keep_thresh"] = np.array(keep)
            out[mode + "_data"] = r["data"]
            out[mode + "_gt_box"] = r["gt_box"]
            out[mode + "_n_mask"] = np.array(len(r["gt_mask"]))
            for i, m in enumerate(r["gt_mask"]):
                out["%s_mask_%d" % (mode, i)] = m
            print("dataset", mode, r["data"].shape, r["data"].dtype, r["gt_box"].shape, len(r["gt_mask"]))
    finally:
        cfg.LABEL_MAP, cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH = saved
    np.savez_compressed(os.path.join(OUT, "dataset_cases.npz"), **out)


FRAMES_SCENE = "scene0000_00"


def write_frames_fixture():
    """tests/golden/frames_square/scene0000_00/{depth,color,label,pose}/<id>.* + world2grid.txt + the container
    tests/golden/scene0000_00__0.chunk naming the three frames.  Small synthetic images (seeded), encoded with Pillow."""
    from PIL import Image
    from sis3d.datasets import scene_file
    root = os.path.join(OUT, "frames_square", FRAMES_SCENE)
    for d in ("depth", "color", "label", "pose"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    g = np.random.default_rng(11)
    ids = [17, 420, 9000]
    for k, fid in enumerate(ids):
        yy, xx = np.mgrid[0:60, 0:80]
        depth = (900 + 13 * xx + 7 * yy + 400 * k + g.integers(0, 50, (60, 80))).astype(np.uint16)     # millimetres
        depth[g.random((60, 80)) < 0.05] = 0
        Image.fromarray(depth).save(os.path.join(root, "depth", "%d.png" % fid))
        col = np.stack([(3 * xx + 40 * k) % 256, (5 * yy + 2 * xx) % 256, (xx * yy // 7 + 90 * k) % 256], -1)
        col = np.kron(col, np.ones((2, 2, 1))).astype(np.uint8)                                           # 120 x 160 blocks
        col = (col + g.integers(0, 12, col.shape)).clip(0, 255).astype(np.uint8)
        Image.fromarray(col).save(os.path.join(root, "color", "%d.jpg" % fid), quality=92)
        lab = g.integers(0, 45, (60, 80)).astype(np.uint8)                                               # nyu40 ids incl. > 40
        Image.fromarray(lab).save(os.path.join(root, "label", "%d.png" % fid))
        a = 0.3 * (k + 1)
        pose = np.array([[np.cos(a), 0, np.sin(a), 1.5 + k], [0, 1, 0, 0.25 * k], [-np.sin(a), 0, np.cos(a), -2.0], [0, 0, 0, 1]])
        with open(os.path.join(root, "pose", "%d.txt" % fid), "w") as f:
            for r in pose:
                f.write(" ".join("%.6f" % v for v in r) + "\n")
    w2g = np.array([[21.3333, 0, 0, 14.0], [0, 21.3333, 0, 19.5], [0, 0, 21.3333, 30.25], [0, 0, 0, 1]])
    with open(os.path.join(root, "world2grid.txt"), "w") as f:
        for r in w2g:
            f.write(" ".join("%.6f" % v for v in r) + "\n")
    dims = (12, 20, 10)
    sdf = (g.standard_normal(dims) * 3).astype(np.float32)
    boxes = np.array([[1.5, 2.2, 0.7, 9.1, 15.9, 7.5], [2.0, 2.0, 2.0, 5.0, 5.0, 5.0]], dtype=np.float32)
    labels = [3, 38]
    masks = [(lab, g.integers(0, 2, tuple(int(np.ceil(b[3 + k]) - np.floor(b[k])) for k in range(3))).astype(np.uint16))
             for b, lab in zip(boxes, labels)]
    w2c = np.eye(4, dtype=np.float32) * 21.333
    w2c[:3, 3] = [3.0, -2.0, 8.5]
    w2c[3, 3] = 1
    path = os.path.join(OUT, FRAMES_SCENE + "__0.chunk")
    scene_file.write_scene_file(path, sdf, boxes, labels, masks, [1.0, 1.0], w2c, ids)
    with open(os.path.join(OUT, "frames_filelist.txt"), "w") as f:
        f.write(FRAMES_SCENE + "__0.chunk\n")
    return path


def frames_case(ns_unused):
    """The reference's Dataset.__getitem__ WITH frames (lib/datasets/dataset.py:136-190,230-267) and its collate_fn
    (dataloader.py:8-49) on the fixture above: chunk mode with colour JPEGs at the benchmark shapes (328x256 / 41x32) and at
    a second, non-trivially cropped shape; scene mode (depth-directory listing, world2grid.txt minus padding) with label
    PNGs relabelled through the label map (USE_IMAGES_GT).  scipy.misc / torchvision are the Pillow restatements of
    ref_harness.install_image_stubs."""
    write_frames_fixture()
    ns = rh.install(with_trainval=True)
    rh.install_image_stubs()
    from lib.datasets.dataset import Dataset as RefDataset
    from lib.datasets.dataloader import collate_fn as ref_collate
    cfg = ns.cfg
    keys = ("LABEL_MAP", "USE_IMAGES", "USE_IMAGES_GT", "USE_MASK", "KEEP_THRESH", "BASE_IMAGE_PATH", "IMAGE_TYPE", "IMAGE_EXT",
            "IMAGE_SHAPE", "DEPTH_SHAPE", "COLOR_MEAN", "COLOR_STD", "MODE", "NUM_IMAGES")
    saved = {k: cfg[k] for k in keys}
    out = {}
    old = os.getcwd()
    os.chdir(OUT)
    try:
        cfg.LABEL_MAP = os.path.join(OUT, "synthetic_labels.csv")
        cfg.BASE_IMAGE_PATH = os.path.join(OUT, "frames_square")
        cfg.USE_IMAGES, cfg.USE_MASK, cfg.KEEP_THRESH, cfg.MODE, cfg.NUM_IMAGES = True, True, 0.0, "benchmark", 5
        for tag, mode, itype, ext, ishape, dshape, gt in (
                ("chunk_color", "chunk", "color", ".jpg", [328, 256], [41, 32], False),
                ("chunk_crop", "chunk", "color", ".jpg", [100, 90], [30, 30], False),
                ("scene_label", "scene", "label", ".png", [41, 32], [41, 32], True)):
            cfg.IMAGE_TYPE, cfg.IMAGE_EXT, cfg.IMAGE_SHAPE, cfg.DEPTH_SHAPE, cfg.USE_IMAGES_GT = itype, ext, ishape, dshape, gt
            r = RefDataset("frames_filelist.txt", mode)[0]
            ni = r["nearest_images"]
            blobs = ref_collate([r])
            fids = [int(v) for v in ni["frameids"]]
            out[tag + "_frameids"] = np.array(fids)
            out[tag + "_world2grid"] = np.asarray(ni["world2grid"])
            out[tag + "_poses"] = np.stack(ni["poses"])
            out[tag + "_depths"] = np.stack(ni["depths"])
            imgs = np.stack([np.asarray(i) for i in ni["images"]]).astype(np.float32)
            if imgs.size > 200000:
                out[tag + "_images_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(imgs).tobytes()).digest(), dtype=np.uint8)
                out[tag + "_images_sub"] = imgs[:, :, ::16, ::16]
            else:
                out[tag + "_images"] = imgs
            out[tag + "_gt_box"] = r["gt_box"]
            out[tag + "_image_files"] = np.array([os.path.relpath(p, OUT) for p in r["image_files"]])
            out[tag + "_blob_world2grid"] = blobs["nearest_images"]["world2grid"][0].numpy()
            out[tag + "_blob_images_shape"] = np.array(blobs["nearest_images"]["images"][0].shape)
            print("frames", tag, fids, imgs.shape, imgs.dtype, ni["depths"][0].shape, ni["depths"][0].dtype)
    finally:
        os.chdir(old)
        for k, v in saved.items():
            cfg[k] = v
    np.savez_compressed(os.path.join(OUT, "dataset_frames_cases.npz"), **out)


def suncg_case():
    """second model family (SUNCG_Backbone, experiments/cfgs/SUNCG/rpn_class_mask_5.yml: colour + geometry, 3 / 6 anchors,
    its own label map).  Run in a SUBPROCESS: the reference's cfg is a process-wide singleton."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.argv=['x']; import make_golden as mg, ref_harness as rh; "
            "ns = rh.install('experiments/cfgs/SUNCG/rpn_class_mask_5.yml'); "
            "print('SUNCG NUM_CLASSES', ns.cfg.NUM_CLASSES, flush=True); "
            "mg.e2e(ns, 'e2e_suncg_small', True, (64, 32, 48), 3, n_views=3, n_per_view=400, sub=2)" % HERE)
    subprocess.check_call([sys.executable, "-c", code])


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = rh.install()
    if "--backward" in sys.argv:
        roi_backward_case(ns)
        return
    if "--projection-backward" in sys.argv:      # round 3: pins the oracle's Projection.backward
        projection_backward_case(ns)
        return
    if "--mask-images" in sys.argv:              # round 2: the mask head's colour variants
        e2e(ns, "e2e_mask_use_images_small", True, (64, 32, 48), 8, n_views=3, n_per_view=2500, sub=2, mask_images="use")
        e2e(ns, "e2e_mask_only_images_small", True, (64, 32, 48), 8, n_views=3, n_per_view=2500, sub=2, mask_images="only")
        return
    if "--frames" in sys.argv:                   # round 2: per-frame image / depth / pose loading of the Dataset
        frames_case(ns)
        return
    if "--enet" in sys.argv:                     # only the round-2 additions (2D encoder + RGB end-to-end)
        enet_case(ns)
        e2e(ns, "e2e_rgb_small", True, (64, 32, 48), 6, n_views=3, n_per_view=400, sub=2, rgb=True)
        return
    nms_cases(ns)
    roi_cases(ns)
    projection_cases(ns)
    projection_backward_case(ns)
    anchors_case(ns)
    compute_projection_cases(ns)
    benchmark_case(ns)
    dataset_case(ns)
    enet_case(ns)
    e2e(ns, "e2e_rgb_small", True, (64, 32, 48), 6, n_views=3, n_per_view=400, sub=2, rgb=True)
    if "--only-new" in sys.argv:
        return
    e2e(ns, "e2e_geometry_full", False, (96, 48, 96), 0, sub=4)
    e2e(ns, "e2e_geometry_small", False, (64, 32, 48), 1, sub=2)
    e2e(ns, "e2e_images_small", True, (64, 32, 48), 2, n_views=3, n_per_view=400, sub=2)
    e2e(ns, "e2e_only_images_small", True, (64, 32, 48), 4, n_views=3, n_per_view=400, sub=2, only_images=True)
    suncg_case()


if __name__ == "__main__":
    main()
