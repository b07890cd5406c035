"""Inference-time executor of the ENet encoder (sis3d/nets/enet.py) with BatchNorm and the eval-mode dropout scale folded
into the convolutions.

The module tree of enet.py keeps the reference checkpoint's parameter names (lib/nets/enet.py:130-694), which costs a
BatchNorm and a scale kernel after almost every convolution: ~350 launches per pass of the 5 views, most of them 3-5 us
of launch for ~0 work.  In eval mode  bn(conv(x)) = conv(x) * g + h  with  g = gamma / sqrt(var + eps),  h = beta - mean * g,
and the torch7-style dropout multiplies by (1 - p): both fold into the convolution's weight and bias.  The executor walks
the SAME modules, builds the folded tensors once per parameter version, and runs conv / PReLU / add / max-pool only
(~190 launches).  Results differ from the module tree by the rounding of the folded weights (~1e-6 of the feature scale;
the GPU test against the oracle keeps its 1e-4 bound).  Parameters stay where the checkpoint put them: this is a read-only view.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .enet import AppendZeroChannels, Branches, Join, Passthrough, ScaledDropout2d, Skip


def _bn_affine(bn):
    g = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    return g, bn.bias.detach() - bn.running_mean.detach() * g


def _fold_path(mods):
    """[Conv2d | BatchNorm2d | PReLU | ScaledDropout2d ...] -> list of ('conv', W, b, stride, padding, dilation) / ('prelu', slope)"""
    ops, i, mods = [], 0, list(mods)
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            w = m.weight.detach()
            b = m.bias.detach() if m.bias is not None else torch.zeros(m.out_channels, device=w.device, dtype=w.dtype)
            j = i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d):
                g, h = _bn_affine(mods[j])
                w, b = w * g.view(-1, 1, 1, 1), b * g + h
                j += 1
            if j < len(mods) and isinstance(mods[j], ScaledDropout2d):
                s = 1.0 - mods[j].p
                w, b = w * s, b * s
                j += 1
            ops.append(("conv", w.contiguous(), b.contiguous(), m.stride, m.padding, m.dilation))
            i = j
        elif isinstance(m, nn.PReLU):
            ops.append(("prelu", m.weight.detach()))
            i += 1
        else:
            raise TypeError("unexpected module in an ENet conv path: %r" % (m,))
    return ops


def _run(ops, x):
    for op in ops:
        if op[0] == "conv":
            x = F.conv2d(x, op[1], op[2], op[3], op[4], op[5])
        else:
            x = F.prelu(x, op[1])
    return x


class FoldedEncoder(object):
    """fixed + trainable halves of the encoder (enet.split_enet_for_3d) as one folded eval pass: enc(images) -> (V,128,h,w)"""

    def __init__(self, fixed, trainable):
        self.entries = list(fixed) + list(trainable)
        self._key = None
        self._plan = None

    def _version(self):
        return tuple((p.data_ptr(), p._version) for m in self.entries for p in list(m.parameters()) + list(m.buffers()))

    def _build(self):
        e = self.entries
        if not (isinstance(e[0], Branches) and isinstance(e[1], Join) and e[1].how == "cat" and isinstance(e[2], nn.BatchNorm2d)
                and isinstance(e[3], nn.PReLU)):
            raise TypeError("not the ENet initial block")
        conv0 = e[0][0]
        g, h = _bn_affine(e[2])
        nc = conv0.out_channels
        plan = {"init": (conv0.weight.detach() * g[:nc].view(-1, 1, 1, 1), conv0.bias.detach() * g[:nc] + h[:nc], conv0.stride, conv0.padding,
                         g[nc:].view(1, -1, 1, 1), h[nc:].view(1, -1, 1, 1), e[3].weight.detach()), "blocks": []}
        for m in e[4:]:
            if not (isinstance(m, nn.Sequential) and isinstance(m[0], Branches) and isinstance(m[1], Join) and m[1].how == "add"
                    and isinstance(m[2], nn.PReLU)):
                raise TypeError("not an ENet bottleneck: %r" % (m,))
            path, skip = m[0][0], m[0][1]
            pool = extra = None
            for s in skip:
                if isinstance(s, nn.MaxPool2d):
                    pool = (s.kernel_size, s.stride)
                elif isinstance(s, AppendZeroChannels):
                    extra = s.extra
                elif not isinstance(s, (Passthrough, Skip)):
                    raise TypeError("unexpected module on an ENet skip path: %r" % (s,))
            plan["blocks"].append((_fold_path(path), pool, extra, m[2].weight.detach()))
        return plan

    def __call__(self, images):
        key = self._version()
        if key != self._key:
            self._plan, self._key = self._build(), key
        w, b, stride, padding, gs, hs, slope = self._plan["init"]
        x = images.float()
        x = F.prelu(torch.cat([F.conv2d(x, w, b, stride, padding), F.max_pool2d(x, 2, 2) * gs + hs], 1), slope)
        for ops, pool, extra, slope in self._plan["blocks"]:
            y = _run(ops, x)
            s = x
            if pool is not None:
                s = F.max_pool2d(s, pool[0], pool[1])
            if extra:
                y[:, :s.shape[1]].add_(s)                     # the skip path's appended channels are zeros: nothing to add there
            else:
                y = y + s
            x = F.prelu(y, slope)
        return x.contiguous()
