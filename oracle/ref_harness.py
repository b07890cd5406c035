"""Run the REAL reference (Sekunde/3D-SIS, /root/reference) in place on CPU.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`3d-sis_amd/`) may import
this file.  It exists to (a) pin `oracle/sis3d_oracle.py` (our CPU restatement)
against the reference's own code and (b) generate the golden fixtures under
`tests/golden/` (see `oracle/make_golden.py`).  `/root/reference` only exists in
the build container; on the GPU box `available()` is False and every user of
this module must skip.

How the reference is made to run:
  * `sys.path` gets `/root/reference` and the cwd is switched there while the
    reference runs (anchor files are opened by relative path,
    lib/layer_utils/generate_anchors.py:19).
  * missing third-party modules are stubbed (easydict, ipdb, skimage, plyfile,
    the two cffi `_ext` packages ...).
  * `yaml.load` gets a SafeLoader (lib/utils/config.py:296 predates PyYAML 6).
  * `.cuda()` is neutralised so `Network.forward` (lib/nets/network.py:75,191)
    runs on CPU: this is the complete form of the README's "MAX_VOLUME=0 CPU
    path".
  * `RoIPoolFunction` (legacy instance-style autograd Function,
    lib/layer_utils/roi_pooling/roi_pool.py:9-38) is replaced by a callable
    that runs the reference's own `roi_pooling.c` compiled by
    `oracle/Makefile` into `oracle/_ref/libref_roi_pooling.so`.
No reference source is copied; it is imported / compiled where it lies.
"""
import contextlib
import ctypes
import os
import sys
import types

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(_HERE, "_ref", "libref_roi_pooling.so")
_REF_TGZ = os.path.join(_HERE, "_ref", "reference_tree.tgz")     # built by `make -C oracle tree`; git-ignored, travels to the GPU box


def _resolve_root():
    """Where the reference lives: $SIS3D_REFERENCE, else /root/reference (the build container), else the staged archive
    oracle/_ref/reference_tree.tgz unpacked into a per-archive temp dir (the GPU box: /root/reference does not exist there).
    The archive is a build output of oracle/Makefile taken from the reference where it lies -- it is never unpacked into the
    repo, never committed and never imported by the product."""
    env = os.environ.get("SIS3D_REFERENCE")
    if env:
        return env, "env"
    if os.path.isdir("/root/reference/lib/nets"):
        return "/root/reference", "in place"
    if os.path.isfile(_REF_TGZ):
        # unpacked into a FRESH private directory (mkdtemp: not a predictable path another user could have pre-populated), members
        # restricted to plain files / directories below it (tarfile's "data" filter), every file checked against the committed
        # manifest tests/golden/reference_tree.sha256 before anything is imported from it; removed again at exit
        import atexit
        import hashlib
        import shutil
        import tarfile
        import tempfile
        root = tempfile.mkdtemp(prefix="sis3d_reference_")
        atexit.register(shutil.rmtree, root, True)
        with tarfile.open(_REF_TGZ) as t:
            try:
                t.extractall(root, filter="data")
            except TypeError:                                  # Python without extraction filters: refuse anything but plain members
                for m in t.getmembers():
                    if not (m.isfile() or m.isdir()) or m.name.startswith(("/", "..")) or "/../" in m.name:
                        raise RuntimeError("reference archive holds an unsafe member: %s" % m.name)
                t.extractall(root)
        manifest = os.path.join(os.path.dirname(_HERE), "tests", "golden", "reference_tree.sha256")
        if os.path.isfile(manifest):
            with open(manifest) as f:
                for ln in f:
                    want, rel = ln.split(None, 1)
                    rel = rel.strip()
                    with open(os.path.join(root, rel), "rb") as fh:
                        if hashlib.sha256(fh.read()).hexdigest() != want:
                            raise RuntimeError("staged reference tree: %s does not match tests/golden/reference_tree.sha256" % rel)
        return root, "staged archive"
    return "/root/reference", "absent"


REF_ROOT, REF_SOURCE = _resolve_root()

_installed = False
_cuda_originals = None


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "nets"))


def neutralise_cuda():
    """`.cuda()` -> identity so the reference's forward (network.py:75,191) runs on CPU: the README's MAX_VOLUME=0 path in full."""
    global _cuda_originals
    if _cuda_originals is None:
        _cuda_originals = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.empty_cache, torch.cuda.synchronize)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.empty_cache = lambda: None
        torch.cuda.synchronize = lambda *a, **k: None


def restore_cuda():
    """undo neutralise_cuda(): the GPU test that runs the reference's caller over the HIP drop-in needs the real `.cuda()`"""
    global _cuda_originals
    if _cuda_originals is not None:
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.empty_cache, torch.cuda.synchronize = _cuda_originals
        _cuda_originals = None


class _EasyDict(dict):
    """20-line stand-in for `easydict.EasyDict` (attribute access + recursion)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


@contextlib.contextmanager
def in_reference_dir():
    old = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        yield
    finally:
        os.chdir(old)


class _RefRoiPoolC:
    """ctypes view of the reference's own roi_pooling.c (built by oracle/Makefile)."""

    class _THFloatTensor(ctypes.Structure):
        _fields_ = [("data", ctypes.POINTER(ctypes.c_float)), ("size", ctypes.c_long * 8)]

    def __init__(self):
        self.lib = ctypes.CDLL(_REF_SO)
        self.lib.roi_pooling_forward.restype = ctypes.c_int

    def _wrap(self, t):
        s = self._THFloatTensor()
        s.data = ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))
        for i, d in enumerate(t.shape):
            s.size[i] = d
        return s

    def forward(self, pw, ph, pl, scale, features, rois):
        features = features.contiguous().float()
        rois = rois.contiguous().float()
        out = torch.zeros(rois.shape[0], features.shape[1], pw, ph, pl)
        f, r, o = self._wrap(features), self._wrap(rois), self._wrap(out)
        rc = self.lib.roi_pooling_forward(
            ctypes.c_int(pw), ctypes.c_int(ph), ctypes.c_int(pl), ctypes.c_float(scale),
            ctypes.byref(f), ctypes.byref(r), ctypes.byref(o))
        assert rc == 1, "reference roi_pooling_forward returned %d" % rc
        return out


_ref_roi = None


def ref_roi_pool_c():
    global _ref_roi
    if _ref_roi is None:
        if not os.path.exists(_REF_SO):
            raise RuntimeError("build oracle/_ref first: make -C oracle ref")
        _ref_roi = _RefRoiPoolC()
    return _ref_roi


class RefRoIPoolCallable:
    """Stands in for `RoIPoolFunction(pw,ph,pl,scale)(features, rois)`."""

    def __init__(self, pw, ph, pl, scale):
        self.a = (int(pw), int(ph), int(pl), float(scale))

    def __call__(self, features, rois):
        return ref_roi_pool_c().forward(*self.a, features, rois)


def install(cfg_file="experiments/cfgs/ScanNet/benchmark.yml", with_trainval=False, on_cpu=True):
    """Import the reference with stubs; returns the module namespace we need.
    on_cpu=True (the oracle's use): `.cuda()` is neutralised so the reference's own forward runs on CPU.
    on_cpu=False (tests/test_gpu_reference_caller.py): the real `.cuda()` stays -- the reference's caller drives the HIP drop-in."""
    global _installed
    if not available():
        raise RuntimeError("reference not available at %s" % REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if not _installed:
        import yaml
        _orig_load = yaml.load

        def _load(stream, Loader=None, **kw):
            return _orig_load(stream, Loader=Loader or yaml.SafeLoader, **kw)
        yaml.load = _load

        _stub("easydict", EasyDict=_EasyDict)
        _stub("ipdb", set_trace=lambda *a, **k: None)
        sk = _stub("skimage")
        sk.transform = _stub("skimage.transform", resize=None)
        _stub("plyfile", PlyData=None, PlyElement=None)
        if "tools" not in sys.modules:
            tools = _stub("tools")
            tools.__path__ = []
            tools.visualization = _stub("tools.visualization", write_bbox=None, write_mask=None)
        # the two cffi extension packages (prebuilt cpython-36/sm_61 .so files are unusable)
        _stub("lib.layer_utils.roi_pooling._ext", roi_pooling=None).__path__ = []
        _stub("lib.layer_utils.nms._ext", nms=None).__path__ = []
        _installed = True
    # neutralise .cuda(): the reference forward calls it unconditionally
    if on_cpu:
        neutralise_cuda()
    else:
        restore_cuda()

    if with_trainval and "reprint" not in sys.modules:
        # import-time dependencies of lib/model/trainval.py that are not installable offline
        _stub("h5py")
        import scipy
        scipy.misc = _stub("scipy.misc")
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tn = _stub("torchnet")
        tn.meter = _stub("torchnet.meter")
        tn.meter.confusionmeter = _stub("torchnet.meter.confusionmeter", ConfusionMeter=None)
        _stub("reprint", output=None)
        _stub("tensorflow")

    with in_reference_dir():
        from lib.utils.config import cfg, cfg_from_file
        cfg_from_file(cfg_file)
        # main.py:41-50 derives NUM_CLASSES from the label map (19 for ScanNet v2)
        import csv
        weights_pre = {}
        with open(cfg.LABEL_MAP) as f:
            for row in csv.DictReader(f):
                weights_pre[int(row["mappedIdConsecutive"])] = float(row["weight"])
        # lib/datasets/dataset.py:269-283 (load_mapping): background weight first
        weights = [0.3280746813009404] + [weights_pre[k] for k in sorted(weights_pre)]
        weights = [w for w in weights if w > 0]
        cfg.NORMALIZE_WEIGHTS = weights
        cfg.NUM_CLASSES = len(weights)
        from lib.nets import backbones, network
        from lib.layer_utils import proposal_layer as pl_mod
        from lib.layer_utils import generate_anchors as ga_mod
        from lib.layer_utils import projection as proj_mod
        from lib.layer_utils.nms import pth_nms
        from lib.layer_utils.roi_pooling import roi_pool
        from lib.utils import bbox_transform
        network.RoIPoolFunction = RefRoIPoolCallable
        trainval = None
        if with_trainval:
            from lib.model import trainval
    ns = types.SimpleNamespace(trainval=trainval, cfg=cfg, backbones=backbones, network=network, proposal_layer=pl_mod,
                               generate_anchors=ga_mod, projection=proj_mod, pth_nms=pth_nms,
                               roi_pool=roi_pool, bbox_transform=bbox_transform)
    return ns


def install_image_stubs():
    """Give the reference's frame loaders (lib/datasets/dataset.py:237-267) the three third-party calls they make, restated
    over Pillow -- the library both absent packages delegate to.  Pinned versions (requirements.txt era): scipy 1.1
    `scipy.misc.imread` (scipy/misc/pilutil.py: Image.open + fromimage: palette -> RGB(A), '1' -> 'L', else numpy.array)
    and torchvision 0.2.1 transforms (functional.py: resize -> img.resize(size[::-1], interpolation); center_crop ->
    i = int(round((h - th) / 2.)), j = int(round((w - tw) / 2.)), img.crop((j, i, j + tw, i + th));
    normalize -> for t, m, s in zip(tensor, mean, std): t.sub_(m).div_(s)).  Test infrastructure."""
    import numpy as np
    from PIL import Image
    import scipy

    def imread(name, flatten=False, mode=None):
        im = Image.open(name)
        if im.mode == "P":
            im = im.convert("RGBA" if "transparency" in im.info else "RGB")
        elif im.mode == "1":
            im = im.convert("L")
        return np.array(im)

    class Resize(object):
        def __init__(self, size, interpolation=Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            return img.resize(tuple(self.size[::-1]), self.interpolation)

    class CenterCrop(object):
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            w, h = img.size
            th, tw = self.size
            i = int(round((h - th) / 2.))
            j = int(round((w - tw) / 2.))
            return img.crop((j, i, j + tw, i + th))

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, tensor):
            for t, m, sd in zip(tensor, self.mean, self.std):
                t.sub_(m).div_(sd)
            return tensor

    scipy.misc.imread = imread
    tv = sys.modules["torchvision.transforms"]
    tv.Resize, tv.CenterCrop, tv.Normalize = Resize, CenterCrop, Normalize


def make_blobs(data, images=None, proj3d=None, proj2d=None, scene_id="syn0"):
    blobs = {"data": data, "id": [scene_id], "gt_box": [torch.zeros(0, 7)], "gt_mask": [[]]}
    if images is not None:
        blobs["nearest_images"] = {"images": [images]}
        blobs["proj_ind_3d"] = [proj3d]
        blobs["proj_ind_2d"] = [proj2d]
    return blobs


def build_net(ns, seed=0, use_images=False, use_mask=True, enet_ckpt=None):
    """`getattr(backbones, cfg.NET)(); init_modules()` as lib/model/trainval.py:68-70 does.
    enet_ckpt: path of an ENet state_dict file -> USE_IMAGES_GT=False, the reference builds its 2D encoder from it
    (create_enet_for_3d always torch.load()s cfg.PRETRAINED_ENET_PATH, enet.py:699) and RGB views are the image input."""
    cfg = ns.cfg
    cfg.USE_IMAGES = bool(use_images)
    cfg.USE_IMAGES_GT = bool(use_images) and enet_ckpt is None   # True: feature maps supplied directly (network.py:199-201)
    if enet_ckpt is not None:
        cfg.PRETRAINED_ENET_PATH = enet_ckpt
    cfg.USE_MASK = bool(use_mask)
    torch.manual_seed(seed)
    with in_reference_dir():
        net = getattr(ns.backbones, cfg.NET)()
        net.init_modules()
    net.eval()
    return net


def forward(ns, net, blobs, killing_inds=()):
    with in_reference_dir(), torch.no_grad():
        net.forward(blobs, "TEST", list(killing_inds))
    return net._predictions


@contextlib.contextmanager
def legacy_int_division():
    """torch 0.4 semantics for `LongTensor / int` (floor division): the reference's compute_projection
    (lib/layer_utils/projection.py:68-70,80-82) relies on it; under torch >= 1.5 `/` is true division and the function
    returns None for every camera.  Patched only while the reference code runs -- the reference itself is unmodified."""
    orig = torch.Tensor.__truediv__

    def div(a, b):
        if isinstance(a, torch.Tensor) and not a.is_floating_point() and not isinstance(b, float) and \
                (not isinstance(b, torch.Tensor) or not b.is_floating_point()):
            return torch.div(a, b, rounding_mode="floor")
        return orig(a, b)
    torch.Tensor.__truediv__ = div
    try:
        yield
    finally:
        torch.Tensor.__truediv__ = orig


@contextlib.contextmanager
def legacy_data_alias():
    """torch 0.4 semantics for `Tensor.data` and `ctx.saved_variables`, which the reference's `Projection.backward`
    (lib/layer_utils/projection.py:139-153) relies on.  In torch 0.4.1 `x.data` wrapped the SAME underlying tensor, so
    `grad_label.data.resize_(C, 32, 41)` resized `grad_label` itself (keeping the first C*32*41 elements of its storage) and
    `grad_label.data.view(...).index_copy_(...)` wrote into it; since torch 1.2 `.data` is a shallow copy whose size changes no
    longer reach the original, so the unmodified backward returns a volume-shaped gradient and autograd rejects it.  Inside a
    backward (grad mode off, tensors that do not require grad) "the same underlying tensor" is the tensor itself: `.data` is
    patched to return `self` and `saved_variables` (removed in torch 2.x) to alias `saved_tensors`, only while the reference
    code runs -- the reference itself is unmodified (same approach as legacy_int_division)."""
    from torch.autograd.function import FunctionCtx
    had = "data" in torch.Tensor.__dict__
    orig = torch.Tensor.__dict__.get("data")
    torch.Tensor.data = property(lambda self: self)
    FunctionCtx.saved_variables = property(lambda self: self.saved_tensors)
    try:
        yield
    finally:
        if had:
            torch.Tensor.data = orig
        else:
            del torch.Tensor.data
        del FunctionCtx.saved_variables


def ref_projection_backward(ns, label, lin_indices_3d, lin_indices_2d, volume_dims, grad_output):
    """d(Projection.apply(label, ...)) / d(label) applied to grad_output, through the reference's own forward + backward run in
    place (CPU) under legacy_data_alias().  -> (output of the forward, grad_label)."""
    lab = label.clone().requires_grad_(True)
    out = ns.projection.Projection.apply(lab, lin_indices_3d, lin_indices_2d, volume_dims)
    with legacy_data_alias():
        out.backward(grad_output)
    return out.detach(), lab.grad.detach().clone()


def ref_compute_projection(ns, depth, camera_to_world, world_to_grid, volume_dims):
    """ProjectionHelper(...).compute_projection of the reference, run in place (CPU)."""
    cfg = ns.cfg
    helper = ns.projection.ProjectionHelper(cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE,
                                            list(volume_dims), cfg.VOXEL_SIZE)
    with legacy_int_division(), contextlib.redirect_stdout(None):
        return helper.compute_projection(depth, camera_to_world, world_to_grid)


def ref_benchmark(ns, net, blobs_list, save_dir):
    """SolverWrapper.benchmark of the reference (lib/model/trainval.py:634-767), run in place on CPU over a list of
    blobs; returns {scene_dir_name: {file: array / unpickled object}}.  `ns` must come from install(with_trainval=True)."""
    import pickle
    cfg = ns.cfg
    cfg.TEST_SAVE_DIR = save_dir
    with in_reference_dir(), legacy_int_division(), torch.no_grad(), contextlib.redirect_stdout(None):
        ns.trainval.SolverWrapper.benchmark(net, blobs_list, None)
    out = {}
    for name in sorted(os.listdir(save_dir)):
        d = os.path.join(save_dir, name)
        files = {}
        for f in sorted(os.listdir(d)):
            if f.endswith(".npy"):
                files[f[:-4]] = np.load(os.path.join(d, f))
            else:
                with open(os.path.join(d, f), "rb") as fh:
                    files[f] = pickle.load(fh)
        out[name] = files
    return out
