"""Mirror of lib/nets/backbones.py: same class names, module tree and parameter names / shapes
(the checkpoint contract), forward on the HIP kernels.

Modules are ordinary nn.Module parameter holders (so load_state_dict / state_dict work as in the
reference) whose forward enqueues hand-written gfx950 kernels on channels-last activations; ReLU,
bias, residual add and the colour/geometry concat are fused into the producing conv's epilogue.
"""
import torch
import torch.nn as nn

from .. import ops
from ..config import cfg as _default_cfg
from .network import Network


class _Packed(object):
    """lazy repack cache, refreshed when the parameter is modified or replaced"""

    def __init__(self):
        self.pc = None

    def get(self, conv, cin_pad=None):
        w, b = conv.weight, conv.bias
        ver = (w._version, None if b is None else b._version, w.data_ptr())
        if self.pc is None or self.pc.version != ver:
            # k1: also the pw16 pack with cout padded to whole 16-row tiles (r6: the mask head's last conv runs on the pointwise kernel)
            self.pc = ops.PackedConv(w, b, cin_pad, pad_cout16=(w.shape[2] == 1))
        return self.pc


class HipConv3d(nn.Conv3d):
    """nn.Conv3d parameters, HIP forward.  k in {1, 3 (pad 1), 2 (stride 2)}; fuse_relu folds the
    following nn.ReLU into the epilogue."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True, fuse_relu=False, fuse_sigmoid=False):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=padding, bias=bias)
        self.fuse_relu, self.fuse_sigmoid = fuse_relu, fuse_sigmoid
        self._packed = _Packed()

    def forward(self, x, residual=None, out=None, out_coff=0):
        k, s = self.kernel_size[0], self.stride[0]
        if self.in_channels == 2 and x.shape[1] == 2 and not ops.is_cl(x):
            y = ops.conv3d_planar2(x, self.weight, k, relu=self.fuse_relu)
            if self.bias is not None:
                raise ops._lib.Sis3dError("planar first-layer conv has no bias in the reference")
            return y
        x = ops.to_cl(x)
        return ops.conv3d(x, self._packed.get(self), stride=s, relu=self.fuse_relu, residual=residual,
                          sigmoid=self.fuse_sigmoid, out=out, out_coff=out_coff)


class FusedReLU(nn.Module):
    """Placeholder that keeps the nn.Sequential indices of the reference (geometry1.1, .5, ...);
    the ReLU itself runs in the producing conv's epilogue."""

    def forward(self, x):
        return x


class HipMaxPool3d(nn.Module):
    """nn.MaxPool3d(3,1,1)"""

    def forward(self, x, out=None, out_coff=0):
        return ops.maxpool3(ops.to_cl(x), out, out_coff)


class Bottleneck(nn.Module):
    """backbones.py:17-40: relu(conv3(relu(conv2(relu(conv1(x))))) + x)"""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv3d(inplanes, planes, 1, stride=stride, fuse_relu=True)
        self.conv2 = HipConv3d(planes, planes, 3, stride=1, padding=1, fuse_relu=True)
        self.conv3 = HipConv3d(planes, inplanes, 1, fuse_relu=True)   # ReLU after the residual add
        self.relu = FusedReLU()
        self.stride = stride

    def forward(self, x, out=None, out_coff=0):
        return self.forward_fused(x, None, None, out, out_coff)[0]

    def forward_fused(self, x, y1=None, nxt=None, out=None, out_coff=0):
        """ONE launch for conv2 + conv3 + residual + ReLU (+ the next block's conv1), the tile never leaves the CU
        between them (sis3d_conv3d_chain).  y1: this block's conv1 output if a previous launch already produced it.
        Returns (block output, next block's conv1 output or None)."""
        x = ops.to_cl(x)
        if y1 is None:
            y1 = self.conv1(x)
        pc2 = self.conv2._packed.get(self.conv2)
        if SPLIT_BOTTLENECK and pc2.packed_t16 is not None:
            # conv2 on the balanced k3 kernel (every SIMD of the chip gets the same MFMA count), then ONE pointwise launch
            # for conv3 + residual + ReLU (+ the next block's conv1 on the tile while it is on chip)
            pc3 = self.conv3._packed.get(self.conv3)
            stage = dict(pc=nxt.conv1._packed.get(nxt.conv1), relu=True) if nxt is not None else None
            try:
                # the whole body in ONE launch: the brick's conv2 tile stays on the CU for conv3 + residual + next conv1
                return ops.bottleneck16(y1, pc2, pc3, x, out=out, out_coff=out_coff, stage=stage)
            except ops.Sis3dUnsupported:
                pass
            y2 = ops.conv3d_k3t16([y1], [pc2], relu=True)[0]
            try:
                return ops.conv3d_pw_chain(y2, pc3, residual=x, relu=True, out=out, out_coff=out_coff, stage=stage)
            except ops.Sis3dUnsupported:
                return self.conv3(y2, residual=x, out=out, out_coff=out_coff), None
        if FUSE_BOTTLENECK:
            stages = [dict(pc=self.conv3._packed.get(self.conv3), relu=True, residual=x, out=out, out_coff=out_coff)]
            if nxt is not None:
                stages.append(dict(pc=nxt.conv1._packed.get(nxt.conv1), relu=True))
            try:
                _, outs = ops.conv3d_chain(y1, self.conv2._packed.get(self.conv2), 1, stages)
                return outs[0], (outs[1] if nxt is not None else None)
            except ops.Sis3dUnsupported:
                pass
        y2 = self.conv2(y1)
        return self.conv3(y2, residual=x, out=out, out_coff=out_coff), None


FUSE_BOTTLENECK = True
SPLIT_BOTTLENECK = not ops.K3_LEGACY     # conv2 through csrc/conv3d_t16.hip + one pointwise launch (default)


class FusedSequential(nn.Sequential):
    """nn.Sequential whose forward fuses across module boundaries: a stem conv's launch also computes the
    following Bottleneck's conv1, and every Bottleneck launch also computes the next Bottleneck's conv1."""

    def forward(self, x, last_out=None, last_coff=0):
        mods = list(self)

        def consumer(i):                                   # the Bottleneck that directly consumes module i's output
            j = i + 1
            while j < len(mods) and isinstance(mods[j], FusedReLU):
                j += 1
            return mods[j] if j < len(mods) and isinstance(mods[j], Bottleneck) else None

        y1 = None
        for i, m in enumerate(mods):
            last = i == len(mods) - 1
            if isinstance(m, Bottleneck):
                o, oc = (last_out, last_coff) if last else (None, 0)
                x, y1 = m.forward_fused(x, y1, consumer(i), o, oc)
            elif isinstance(m, FusedReLU):
                continue
            elif isinstance(m, HipMaxPool3d) and last and last_out is not None:
                x, y1 = m(x, last_out, last_coff), None
            elif isinstance(m, HipConv3d) and m.in_channels == 2 and m.kernel_size[0] == 2 and m.bias is None and m.fuse_relu \
                    and torch.is_tensor(x) and not ops.is_cl(x) and self._stem_planar(m, consumer(i), x) is not None:
                x, y1 = self._stem_out
            elif isinstance(m, HipConv3d) and m.kernel_size[0] == 2 and consumer(i) is not None and torch.is_tensor(x) and ops.is_cl(x) \
                    and self._stem_k2s2(m, consumer(i), x) is not None:
                x, y1 = self._stem_out
            elif isinstance(m, HipConv3d) and FUSE_BOTTLENECK and consumer(i) is not None and not (m.in_channels == 2 and not ops.is_cl(x)) \
                    and not (SPLIT_BOTTLENECK and m.kernel_size[0] == 3 and m._packed.get(m).packed_t16 is not None):
                nb = consumer(i)
                try:
                    x, outs = ops.conv3d_chain(ops.to_cl(x), m._packed.get(m), m.stride[0],
                                               [dict(pc=nb.conv1._packed.get(nb.conv1), relu=True)], relu=m.fuse_relu, want_main=True)
                    y1 = outs[0]
                except ops.Sis3dUnsupported:
                    if isinstance(x, ops.ProjectedVolume):
                        x = x.dense()
                    x, y1 = m(x), None
            else:
                if isinstance(x, ops.ProjectedVolume):
                    x = x.dense()
                x, y1 = m(x), None
        return x


def _stem_planar(self, m, nb, x):
    """geometry1[0] + the first Bottleneck's conv1 as one register-chained launch; None if this shape has no instantiation"""
    try:
        stage = dict(pc=nb.conv1._packed.get(nb.conv1), relu=True) if nb is not None else None
        ver = (m.weight._version, m.weight.data_ptr())
        if getattr(m, "_stem_pack", None) is None or m._stem_pack[0] != ver:      # lives on the module that owns the parameter
            m._stem_pack = (ver, ops.pack_stem_planar2(m.weight))
        self._stem_out = ops.stem_planar2(x, m._stem_pack[1], m.out_channels, relu=True, stage=stage)
    except ops.Sis3dUnsupported:
        return None
    return self._stem_out


def _stem_k2s2(self, m, nb, x):
    """a k2 s2 stem + the following Bottleneck's conv1 as one register-chained launch; None if unsupported"""
    try:
        self._stem_out = ops.conv3d_k2s2_pw16(x, m._packed.get(m), relu=m.fuse_relu, stage=dict(pc=nb.conv1._packed.get(nb.conv1), relu=True))
    except ops.Sis3dUnsupported:
        return None
    return self._stem_out


FusedSequential._stem_planar = _stem_planar
FusedSequential._stem_k2s2 = _stem_k2s2


def _conv_relu(cin, cout, k, stride=1, padding=0):
    return [HipConv3d(cin, cout, k, stride=stride, padding=padding, bias=False, fuse_relu=True), FusedReLU()]


class Base_Backbone(Network):
    def __init__(self, obbox=True, cfg=None):
        super().__init__(cfg)
        self._feat_stride = [4, 4, 4]
        self._fc7_channels = 128
        self._net_conv_level1_channels = 128
        self._net_conv_level2_channels = 128
        self._net_conv_level3_channels = 128

    def _make_classifier(self):
        ps = self.cfg.CLASS_POOLING_SIZE
        return nn.Sequential(nn.Linear(self._net_conv_level1_channels * ps * ps * ps, 256), nn.ReLU(True),
                             nn.Linear(256, 256), nn.ReLU(True), nn.Linear(256, 128), nn.ReLU(True))

    # backbones.py:98-113, split at level1 so the network can overlap the level-1 RPN branch with geometry2
    def _backbone_level1(self):
        cfg = self.cfg
        if cfg.USE_IMAGES and cfg.ONLY_IMAGES:
            return self.color(self._image_input)
        if cfg.USE_IMAGES:
            # torch.cat([color, geometry], 1) (backbones.py:109): the last geometry Bottleneck writes its
            # channel range of the concatenated tensor directly (conv epilogue channel offset)
            # torch.cat([color, geometry], 1): both branches write their channel range of l1 directly (the colour branch's
            # last module -- max-pool or Bottleneck -- and the last geometry Bottleneck); no concat copy
            cc, gc = self._branch_channels(self.color), self._branch_channels(self.geometry1)
            od = tuple(int(v) // 4 for v in self._scene.shape[2:])
            l1 = ops.new_act(cc + gc, od, self._scene.device)
            self.color(self._image_input, last_out=l1, last_coff=0)
            self.geometry1(self._scene, last_out=l1, last_coff=cc)
            return l1
        return self.geometry1(self._scene)

    @staticmethod
    def _branch_channels(seq):
        for m in reversed(list(seq)):
            if isinstance(m, Bottleneck):
                return m.conv3.out_channels
            if isinstance(m, HipConv3d):
                return m.out_channels
        raise ValueError("empty branch")

    def _backbone_level2(self, l1):
        return self.geometry2(l1)

    def _backbone(self):
        l1 = self._backbone_level1()
        return l1, self._backbone_level2(l1), None


class SUNCG_Backbone(Base_Backbone):
    """backbones.py:118-169"""

    def _init_backbone_classifier(self):
        cfg = self.cfg
        if not cfg.ONLY_IMAGES or not cfg.USE_IMAGES:
            self.geometry1 = FusedSequential(*_conv_relu(2, 64, 2, 2), Bottleneck(64, 32), *_conv_relu(64, 64, 2, 2), Bottleneck(64, 32))
        if cfg.USE_IMAGES:
            self.color = FusedSequential(*_conv_relu(cfg.NUM_IMAGE_CHANNELS, 64, 2, 2), Bottleneck(64, 32),
                                         *_conv_relu(64, 64, 2, 2), Bottleneck(64, 32))
        if cfg.USE_IMAGES and cfg.ONLY_IMAGES:
            cin = 64
        elif cfg.USE_IMAGES:
            cin = 128
        else:
            cin = 64
        self.geometry2 = FusedSequential(*_conv_relu(cin, 128, 3, 1, 1), Bottleneck(128, 64))
        self.classifier = self._make_classifier()


class ScanNet_Backbone(Base_Backbone):
    """backbones.py:171-231"""

    def _init_backbone_classifier(self):
        cfg = self.cfg
        if cfg.ONLY_IMAGES:
            gc, cc = 0, 128
        elif cfg.USE_IMAGES:
            gc, cc = 64, 64
        else:
            gc, cc = 128, 0
        if not cfg.ONLY_IMAGES or not cfg.USE_IMAGES:
            self.geometry1 = FusedSequential(*_conv_relu(2, 32, 2, 2), Bottleneck(32, 32), Bottleneck(32, 32),
                                             *_conv_relu(32, gc, 2, 2), Bottleneck(gc, 32), Bottleneck(gc, 32))
        if cfg.USE_IMAGES:
            self.color = FusedSequential(*_conv_relu(cfg.NUM_IMAGE_CHANNELS, 64, 2, 2), Bottleneck(64, 32), HipMaxPool3d(),
                                         *_conv_relu(64, cc, 2, 2), Bottleneck(cc, 32), HipMaxPool3d())
        self.geometry2 = FusedSequential(*_conv_relu(gc + cc, 128, 3, 1, 1), Bottleneck(128, 64), Bottleneck(128, 64), HipMaxPool3d())
        self.classifier = self._make_classifier()


class MaskBackbone(nn.Module):
    """backbones.py:236-287.  geometry: 5 x (Conv3d k3 + ReLU) on the 2-channel crop + a 1x1x1 conv; with MASK_USE_IMAGES a
    second stack `color` of the same shape on the crop of the back-projected image volume and `combine` (k3 128->128, ReLU,
    1x1x1 -> classes) on their concatenation; MASK_ONLY_IMAGES uses the colour stack alone.  All convs bias-free; sigmoid in
    eval mode.  Parameter names = the reference's (`geometry.N.weight`, `color.N.weight`, `combine.N.weight`)."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or _default_cfg
        self.use_images = bool(cfg.MASK_USE_IMAGES)
        self.only_images = bool(cfg.MASK_ONLY_IMAGES)
        if self.only_images and not self.use_images:
            raise ValueError("MASK_ONLY_IMAGES needs MASK_USE_IMAGES (the reference builds `color` only under it, backbones.py:253)")
        nc = cfg.NUM_CLASSES
        mods = _conv_relu(2, 64, 3, 1, 1)
        for _ in range(4):
            mods += _conv_relu(64, 64, 3, 1, 1)
        mods.append(HipConv3d(64, 64 if self.use_images else nc, 1, bias=False))
        self.geometry = nn.Sequential(*mods)
        if self.use_images:
            cm = _conv_relu(cfg.NUM_IMAGE_CHANNELS, 64, 3, 1, 1)
            for _ in range(4):
                cm += _conv_relu(64, 64, 3, 1, 1)
            cm.append(HipConv3d(64, nc if self.only_images else 64, 1, bias=False))
            self.color = nn.Sequential(*cm)
            self.combine = nn.Sequential(*_conv_relu(128, 128, 3, 1, 1), HipConv3d(128, nc, 1, bias=False))

    @staticmethod
    def _stack(seq, x, first=0, last_out=None, last_coff=0, sigmoid=False):
        """conv+ReLU pairs seq[first], seq[first+2], ... then the final 1x1x1 conv (optionally into a channel range / sigmoid)"""
        n = len(seq)
        for i in range(first, n - 1, 2):
            x = seq[i](x)
        last = seq[n - 1]
        last.fuse_sigmoid = bool(sigmoid)
        return last(x, out=last_out, out_coff=last_coff)

    def forward(self, scene, imageft=None, window=None):
        """scene: (1,2,dx,dy,dz) crop (any view of the planar grid with contiguous z), or the full grid +
        window=(x0,y0,z0,x1,y1,z1); imageft: the matching crop of the back-projected volume (1,C,dx,dy,dz) (or the full
        volume with `window`).  Returns logical (1,NUM_CLASSES,dx,dy,dz); sigmoid in eval mode."""
        g = self.geometry
        sig = not self.training
        if self.use_images:
            if imageft is None:
                raise ops._lib.Sis3dError("MASK_USE_IMAGES: the mask head needs the image volume crop")
            if isinstance(imageft, ops.ProjectedVolume):
                imageft = imageft.dense()
            if window is not None:
                x0, y0, z0, x1, y1, z1 = window
                imageft = imageft[:, :, x0:x1, y0:y1, z0:z1]
            # a crop of the channels-last volume is a strided view: one copy into a dense channels-last tile
            col_in = ops.to_cl(imageft.contiguous(memory_format=torch.channels_last_3d) if not ops.is_cl(imageft) else imageft)
            if self.only_images:
                return self._stack(self.color, col_in, sigmoid=sig)
            od = tuple(col_in.shape[2:])
            both = ops.new_act(128, od, col_in.device)       # torch.cat([geometry, color], 1) written in place (backbones.py:282)
            x = ops.conv3d_planar2(scene, g[0].weight, 3, relu=True, window=window)
            self._stack(g, x, first=2, last_out=both, last_coff=0)
            self._stack(self.color, col_in, last_out=both, last_coff=64)
            return self._stack(self.combine, both, sigmoid=sig)
        x = ops.conv3d_planar2(scene, g[0].weight, 3, relu=True, window=window)
        for i in (2, 4, 6, 8):
            x = g[i](x)
        last = g[10]
        last.fuse_sigmoid = sig
        return last(x)


def _mask_forward_batched(self, scene, windows, imageft=None):
    """all boxes at once: one launch per layer over the ragged batch of crops.  Geometry-only head: ops.mask_head_batched.
    r3: MASK_USE_IMAGES / MASK_ONLY_IMAGES (backbones.py:253-284) batched as well -- the image-volume crops are gathered into the
    same packing, the colour stack and `combine` run as ragged launches (ops.RaggedBatch), the geometry stack through the same
    MaskPlan as the geometry-only head with its last 1x1x1 conv producing the 64 features of the concatenation."""
    g = self.geometry
    sig = not self.training
    pcs = [g[i]._packed.get(g[i]) for i in (2, 4, 6, 8)]
    if not self.use_images:
        return ops.mask_head_batched(scene, windows, g[0].weight, pcs, g[10]._packed.get(g[10]), sigmoid=sig)
    if len(windows) == 0:
        return []
    if imageft is None:
        raise ops._lib.Sis3dError("MASK_USE_IMAGES: the mask head needs the image volume")
    if isinstance(imageft, ops.ProjectedVolume):
        imageft = imageft.dense()
    vol = ops.to_cl(imageft)
    rb = ops.RaggedBatch(windows, vol.device)
    c = self.color
    x = rb.gather(vol)
    for i in (0, 2, 4, 6, 8):
        x = rb.conv(x, c[i]._packed.get(c[i]), relu=True)
    if self.only_images:
        return rb.views(rb.conv(x, c[10]._packed.get(c[10]), sigmoid=sig))
    col = rb.conv(x, c[10]._packed.get(c[10]))
    plan = ops.MaskPlan(windows, g[2].out_channels, g[10].out_channels, vol.device)
    ops.mask_head_run(scene, plan, g[0].weight, pcs, g[10]._packed.get(g[10]), sigmoid=False)
    both = torch.cat([plan.out, col], 1)                       # torch.cat([geometry, color], 1) of backbones.py:282, all boxes at once
    m = self.combine
    y = rb.conv(both, m[0]._packed.get(m[0]), relu=True)
    return rb.views(rb.conv(y, m[2]._packed.get(m[2]), sigmoid=sig))


def _mask_plan(self, windows, device):
    """host-side part of forward_batched for a fixed window set (ops.MaskPlan): build once, run many times / capture"""
    g = self.geometry
    return ops.MaskPlan(windows, g[2].out_channels, g[10].out_channels, device)


def _mask_forward_planned(self, scene, plan):
    g = self.geometry
    pcs = [g[i]._packed.get(g[i]) for i in (2, 4, 6, 8)]
    return ops.mask_head_run(scene, plan, g[0].weight, pcs, g[10]._packed.get(g[10]), sigmoid=not self.training)


MaskBackbone.forward_batched = _mask_forward_batched
MaskBackbone.plan = _mask_plan
MaskBackbone.forward_planned = _mask_forward_planned


def state_dict_shapes(cfg=None):
    """{name: shape} of the checkpoint for cfg (meta-device build: no allocation, no RNG use)."""
    cfg = cfg or _default_cfg
    with torch.device("meta"):
        net = globals()[cfg.NET](cfg=cfg)
        net.init_modules()
    return {k: tuple(v.shape) for k, v in net.state_dict().items()}
