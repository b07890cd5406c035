"""ENet 2D encoder for the colour branch (reference: lib/nets/enet.py:130-715, used at lib/nets/network.py:63-64,199-205).

The reference file is a torch7 -> PyTorch conversion: one flat `nn.Sequential` of 27 entries whose nesting (table /
reduce containers) fixes the checkpoint's parameter names -- `4.0.0.3.weight` is "entry 4, branch table, conv branch,
4th layer".  3D-SIS cuts it in three (`create_enet_for_3d`, enet.py:697-715):

    image_enet_fixed          entries 0..17   initial block, stage 1 (16 -> 64, 1/4 res), stage 2 (64 -> 128, 1/8 res)
    image_enet_trainable      entries 18..25  stage 3 (the second run of regular / dilated / asymmetric bottlenecks)
    image_enet_classification entry 26        1x1 classifier (not used by the TEST forward)

and runs fixed -> trainable on the (V,3,256,328) views to get the (V,128,32,41) feature maps that are back-projected
into the voxel grid.  This module rebuilds that tree from a compact stage table (same container nesting, hence the same
`state_dict` keys and shapes: a reference checkpoint loads with strict=True) with plain torch operators; on the GPU box
they run on PyTorch-ROCm (MIOpen) -- the 2D encoder is outside the 3D hot path that the hand-written kernels cover
(5.2 GFLOP for 5 views).

Semantics kept from the reference containers:
  * a bottleneck is  PReLU( conv_branch(x) + skip_branch(x) ); the initial block concatenates its two branches;
  * BatchNorm eps = 1e-3; PReLU has one slope per channel;
  * dropout is the torch7 "v1" kind: at eval time the activation is SCALED by (1 - p) (enet.py:80-95), p = 0.01 in
    stage 1 and 0.1 afterwards -- not the identity of nn.Dropout2d;
  * a down-sampling bottleneck's skip branch is max-pool 2x2 followed by zero channels appended up to the new width.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class Branches(nn.Sequential):
    """every child sees the same input -> list of their outputs (the converter's ConcatTable)"""

    def forward(self, x):
        return [m(x) for m in self]


class Join(nn.Module):
    """list -> tensor: channel concatenation ('cat', initial block) or element-wise sum ('add', bottlenecks)"""

    def __init__(self, how):
        super().__init__()
        self.how = how

    def forward(self, xs):
        if self.how == "cat":
            return torch.cat(xs, 1)
        out = xs[0]
        for t in xs[1:]:
            out = out + t
        return out


class Skip(nn.Sequential):
    """identity when empty (the regular bottleneck's skip path is an empty container holding a no-op)"""

    def forward(self, x):
        for m in self:
            x = m(x)
        return x


class Passthrough(nn.Module):
    def forward(self, x):
        return x


class AppendZeroChannels(nn.Module):
    """(N,C,H,W) -> (N,C+extra,H,W), zeros appended (enet.py:48-77 with dim=0, nInputDim=3, pad>0)"""

    def __init__(self, extra):
        super().__init__()
        self.extra = int(extra)

    def forward(self, x):
        return F.pad(x, (0, 0, 0, 0, 0, self.extra))


class ScaledDropout2d(nn.Dropout2d):
    """torch7-style dropout: the input is scaled by (1 - p) and THEN nn.Dropout2d is applied, so eval mode returns
    x * (1 - p) and train mode drops without the 1/(1-p) boost (enet.py:89-95)"""

    def forward(self, x):
        return super().forward(x * (1.0 - self.p))


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.1, affine=True)


def _bottleneck(cin, cout, mid, kind, arg, p_drop):
    """one ENet bottleneck as the nested container  Sequential(Branches(conv_path, skip_path), Join('add'), PReLU).
    kind: 'down' (2x2/s2 projection + pooled skip), 'regular', 'dilated' (arg = dilation), 'asym' (arg = kernel length)"""
    if kind == "down":
        first = nn.Conv2d(cin, mid, 2, stride=2, bias=False)
    else:
        first = nn.Conv2d(cin, mid, 1, bias=False)
    path = [first, _bn(mid), nn.PReLU(mid)]
    if kind == "asym":
        h = arg // 2
        path += [nn.Conv2d(mid, mid, (1, arg), padding=(0, h), bias=False), nn.Conv2d(mid, mid, (arg, 1), padding=(h, 0), bias=True)]
    else:
        d = arg if kind == "dilated" else 1
        path += [nn.Conv2d(mid, mid, 3, padding=d, dilation=d, bias=True)]
    path += [_bn(mid), nn.PReLU(mid), nn.Conv2d(mid, cout, 1, bias=False), _bn(cout), ScaledDropout2d(p_drop)]
    skip = [Passthrough()]
    if kind == "down":
        skip += [nn.MaxPool2d(2, 2), AppendZeroChannels(cout - cin)]
    return nn.Sequential(Branches(nn.Sequential(*path), Skip(*skip)), Join("add"), nn.PReLU(cout))


# stages 2 and 3 share this run (ENet paper, table 1): regular, dilated 2, asymmetric 5, dilated 4, regular, dilated 8,
# asymmetric 5, dilated 16
_STAGE23 = [("regular", 0), ("dilated", 2), ("asym", 5), ("dilated", 4), ("regular", 0), ("dilated", 8), ("asym", 5), ("dilated", 16)]


def create_enet(num_classes):
    """the 27-entry encoder + classifier, entry for entry as the reference's `create_enet` (enet.py:130-694)"""
    mods = [Branches(nn.Conv2d(3, 13, 3, stride=2, padding=1, bias=True), nn.MaxPool2d(2, 2)), Join("cat"), _bn(16), nn.PReLU(16)]
    mods.append(_bottleneck(16, 64, 16, "down", 0, 0.01))
    mods += [_bottleneck(64, 64, 16, "regular", 0, 0.01) for _ in range(4)]
    mods.append(_bottleneck(64, 128, 32, "down", 0, 0.1))
    for _ in range(2):
        mods += [_bottleneck(128, 128, 32, kind, arg, 0.1) for kind, arg in _STAGE23]
    mods.append(nn.Sequential(nn.Conv2d(128, num_classes, 1, bias=False)))
    return nn.Sequential(*mods)


def split_enet_for_3d(model):
    """enet.py:701-705: (fixed = entries [0, n-9), trainable = [n-9, n-1), classifier = [n-1]); the fixed part's
    parameters do not require grad"""
    n = len(model)
    fixed = nn.Sequential(*(model[i] for i in range(n - 9)))
    trainable = nn.Sequential(*(model[i] for i in range(n - 9, n - 1)))
    classifier = nn.Sequential(model[n - 1])
    for p in fixed.parameters():
        p.requires_grad = False
    return fixed, trainable, classifier


def create_enet_for_3d(num_2d_classes, model_path, num_3d_classes=None):
    """Reference signature (enet.py:697).  The reference always loads `model_path`; checkpoints cannot be fetched in this
    environment, so a missing / empty path leaves PyTorch's default initialisation in place (image path)."""
    model = create_enet(num_2d_classes)
    if model_path and os.path.isfile(model_path):
        model.load_state_dict(torch.load(model_path, map_location="cpu"))
    return split_enet_for_3d(model)
