"""Inference-time executor of the ENet encoder (sis3d/nets/enet.py) on the HIP kernels of csrc/enet.hip: one launch per bottleneck.

Reads the SAME modules as nets/enet_folded.py (the reference checkpoint's parameter names, lib/nets/enet.py:130-694), folds BatchNorm
(eval) and the torch7-style dropout scale into the convolutions exactly as that executor does (`_fold_path`), repacks the folded
weights into the lane order of the 16x16x4 fp32 MFMA (`pack_pw`: [cout/16][cin/16][64][4], the layout of sis3d_conv_pw16_pack_weight)
and walks the 22 bottlenecks: (V,3,256,328) images -> (V,128,32,41) feature maps in 25 launches instead of ~190 library operators.
Activations between launches are pixels x channels rows (NHWC); the last launch writes the NCHW maps the back-projection reads.
Inference only (eval mode, no autograd), like the folded executor."""
import torch
import torch.nn as nn

from .. import _lib
from ..ops import _ptr, _stream, check, lib
from .enet import AppendZeroChannels, Branches, Join
from .enet_folded import _bn_affine, _fold_path


def pack_pw(w2d):
    """(Cout, Cin) -> [Cout/16][Cin/16][64][4]: lane l = 16 kq + i of tile (ct, g) holds W[16 ct + i][16 g + 4 kq + r], r = 0..3"""
    co, ci = w2d.shape
    if co % 16 or ci % 16:
        raise _lib.Sis3dError("enet_hip: channel counts must be multiples of 16, got %d x %d" % (co, ci))
    return w2d.reshape(co // 16, 16, ci // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def pack_taps(w):
    """(Cout, Cin, kh, kw) -> [kh * kw][Cout/16][Cin/16][64][4], tap = ky * kw + kx"""
    kh, kw = w.shape[2], w.shape[3]
    return torch.cat([pack_pw(w[:, :, ky, kx]) for ky in range(kh) for kx in range(kw)])


class _Block(object):
    __slots__ = ("cin", "c", "mid", "down", "kind", "dil", "w1", "b1", "s1", "w2", "b2", "s2", "w2b", "w3", "b3", "s3")


def _plan_block(m):
    if not (isinstance(m, nn.Sequential) and isinstance(m[0], Branches) and isinstance(m[1], Join) and m[1].how == "add" and isinstance(m[2], nn.PReLU)):
        raise TypeError("not an ENet bottleneck: %r" % (m,))
    ops = _fold_path(m[0][0])
    kinds = [o[0] for o in ops]
    b = _Block()
    if kinds == ["conv", "prelu", "conv", "prelu", "conv"]:
        c1, p1, c2, p2, c3 = ops
        b.kind, b.w2b = 0, None
        if tuple(c2[1].shape[2:]) != (3, 3) or c2[5][0] != c2[5][1] or tuple(c2[4]) != tuple(c2[5]) or tuple(c2[3]) != (1, 1):
            raise TypeError("enet_hip: conv2 must be 3x3, stride 1, padding = dilation")
        b.dil = int(c2[5][0])
        b.w2 = pack_taps(c2[1])
        b.b2 = c2[2].contiguous()
    elif kinds == ["conv", "prelu", "conv", "conv", "prelu", "conv"]:
        c1, p1, ca, cb, p2, c3 = ops
        if tuple(ca[1].shape[2:]) != (1, 5) or tuple(cb[1].shape[2:]) != (5, 1) or float(ca[2].abs().max()) != 0.0:
            raise TypeError("enet_hip: asymmetric pair must be (1,5) without bias then (5,1)")
        b.kind, b.dil = 1, 1
        b.w2, b.w2b, b.b2 = pack_taps(ca[1]), pack_taps(cb[1]), cb[2].contiguous()
    else:
        raise TypeError("enet_hip: unexpected conv path %r" % (kinds,))
    b.down = tuple(c1[1].shape[2:]) == (2, 2)
    if not b.down and tuple(c1[1].shape[2:]) != (1, 1):
        raise TypeError("enet_hip: conv1 must be 1x1 or 2x2 / stride 2")
    has_pad = any(isinstance(s, AppendZeroChannels) for s in m[0][1])
    if has_pad != b.down:
        raise TypeError("enet_hip: a down block pools and pads its skip path, the others pass it through")
    b.mid, b.cin, b.c = int(c1[1].shape[0]), int(c1[1].shape[1]), int(c3[1].shape[0])
    b.w1, b.b1, b.s1 = pack_taps(c1[1]), c1[2].contiguous(), p1[1].contiguous()
    b.s2 = p2[1].contiguous()
    b.w3, b.b3, b.s3 = pack_pw(c3[1][:, :, 0, 0]), c3[2].contiguous(), m[2].weight.detach().contiguous()
    for s in (b.s1, b.s2):
        if s.numel() != b.mid:
            raise TypeError("enet_hip: per-channel PReLU expected")
    return b


class HipEncoder(object):
    """fixed + trainable halves of the encoder (enet.split_enet_for_3d) on csrc/enet.hip: enc(images) -> (V,128,h,w)"""

    def __init__(self, fixed, trainable):
        self.entries = list(fixed) + list(trainable)
        self._key = None
        self._plan = None

    def _version(self):
        return tuple((p.data_ptr(), p._version) for m in self.entries for p in list(m.parameters()) + list(m.buffers()))

    def _build(self):
        e = self.entries
        if not (isinstance(e[0], Branches) and isinstance(e[1], Join) and e[1].how == "cat" and isinstance(e[2], nn.BatchNorm2d) and isinstance(e[3], nn.PReLU)):
            raise TypeError("not the ENet initial block")
        conv0 = e[0][0]
        g, h = _bn_affine(e[2])
        nc = conv0.out_channels
        if (nc, conv0.in_channels, tuple(conv0.kernel_size), tuple(conv0.stride), tuple(conv0.padding)) != (13, 3, (3, 3), (2, 2), (1, 1)):
            raise TypeError("enet_hip: initial block must be Conv2d(3, 13, 3, stride 2, padding 1) || MaxPool2d(2, 2)")
        init = ((conv0.weight.detach() * g[:nc].view(-1, 1, 1, 1)).contiguous(), (conv0.bias.detach() * g[:nc] + h[:nc]).contiguous(),
                g[nc:].contiguous(), h[nc:].contiguous(), e[3].weight.detach().contiguous())
        return {"init": init, "blocks": [_plan_block(m) for m in e[4:]]}

    def __call__(self, images):
        if not images.is_cuda:
            raise _lib.Sis3dError("enet_hip: images must be on the GPU (the product has no CPU path)")
        # ONE walk over the module tree per call: (data_ptr, version) of every parameter / buffer tells both whether the folded plan is
        # still valid and -- only when it is not -- that everything has to be checked for its device before raw pointers reach a kernel
        # (a Python error here, not a GPU fault there).  A tensor that moves device changes its data_ptr, so an unchanged key implies
        # an unchanged placement.
        key = (self._version(), images.device)
        if key != self._key:
            for m in self.entries:
                for q in list(m.parameters()) + list(m.buffers()):
                    if q.device != images.device:
                        raise _lib.Sis3dError("enet_hip: encoder weights are on %s but the images on %s (call net.cuda() first)"
                                              % (q.device, images.device))
            self._plan, self._key = self._build(), key
        L = lib()
        st = _stream()
        x = images.float().contiguous()
        V, _, hi, wi = x.shape
        dev = x.device
        w0, b0, ps, ph, s0 = self._plan["init"]
        H, W = hi // 2, wi // 2
        cur = torch.empty(V * H * W, 16, device=dev)
        check(L.sis3d_enet_initial(_ptr(x), V, hi, wi, _ptr(w0), _ptr(b0), _ptr(ps), _ptr(ph), _ptr(s0), _ptr(cur), st), "sis3d_enet_initial")
        blocks = self._plan["blocks"]
        y1 = None
        out = None
        for i, b in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            if b.down:
                if H % 2 or W % 2:
                    raise _lib.Sis3dError("enet_hip: a stride-2 block needs even feature-map sides, got %d x %d" % (H, W))
                H, W = H // 2, W // 2
                y1 = torch.empty(V * H * W, b.mid, device=dev)
                check(L.sis3d_enet_conv1(_ptr(cur), V, H, W, b.cin, b.mid, 4, _ptr(b.w1), _ptr(b.b1), _ptr(b.s1), _ptr(y1), st), "sis3d_enet_conv1")
            elif y1 is None:
                y1 = torch.empty(V * H * W, b.mid, device=dev)
                check(L.sis3d_enet_conv1(_ptr(cur), V, H, W, b.cin, b.mid, 1, _ptr(b.w1), _ptr(b.b1), _ptr(b.s1), _ptr(y1), st), "sis3d_enet_conv1")
            fuse = nxt is not None and not nxt.down
            last = nxt is None
            out = torch.empty((V, b.c, H, W) if last else (V * H * W, b.c), device=dev)
            y1n = torch.empty(V * H * W, nxt.mid, device=dev) if fuse else None
            check(L.sis3d_enet_block(_ptr(cur), _ptr(y1), V, H, W, b.c, b.mid, b.kind, b.dil, _ptr(b.w2), _ptr(b.b2), _ptr(b.s2), _ptr(b.w2b),
                                     _ptr(b.w3), _ptr(b.b3), _ptr(b.s3), b.cin if b.down else 0, _ptr(out), 1 if last else 0,
                                     _ptr(nxt.w1) if fuse else None, _ptr(nxt.b1) if fuse else None, _ptr(nxt.s1) if fuse else None,
                                     nxt.mid if fuse else 0, _ptr(y1n), st), "sis3d_enet_block")
            cur, y1 = out, y1n
        return out
