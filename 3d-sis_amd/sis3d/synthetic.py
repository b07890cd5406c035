"""Seeded synthetic inputs and weights (no datasets or checkpoints exist offline).

Definitions follow SURVEY.md section 8(d):
  * chunk: |truncated sdf| + known-mask, as lib/datasets/dataset.py:66-68 builds
    ``blobs['data']`` (1,2,X,Y,Z) fp32;
  * weights: every state_dict entry drawn U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    (the bound PyTorch's default Conv/Linear init uses) from a generator seeded
    by the parameter NAME, so the reference net and ours get identical tensors
    through ``load_state_dict`` regardless of module construction order;
  * projection index lists: packed like ProjectionHelper.compute_projection
    returns them (lib/layer_utils/projection.py:108-121): int64 (nvox+1,), slot 0
    holds the count.
Pure torch-CPU; used by tests, bench.py and oracle/make_golden.py.
"""
import zlib

import torch

CHUNK_DIMS = (96, 48, 96)


def synth_chunk(chunk_id=0, dims=CHUNK_DIMS, truncated=3.0):
    g = torch.Generator().manual_seed(1234 + int(chunk_id))
    tsdf = 2.0 * torch.randn(*dims, generator=g)
    data = torch.zeros(1, 2, *dims)
    data[0, 0] = tsdf.clamp(-truncated, truncated).abs()
    data[0, 1] = (tsdf > -1).float()
    return data


def synth_sdf(chunk_id=0, dims=CHUNK_DIMS):
    """the raw signed-distance block behind synth_chunk(chunk_id), flat in .chunk FILE order (x fastest, then y, then z:
    datagen/SceneSampler/main.cpp:348-415): what a reader hands to ops.tsdf_encode / a pipeline fed in 'sdf' mode;
    encoding it (dataset.py:54-70) gives synth_chunk(chunk_id) bit for bit"""
    g = torch.Generator().manual_seed(1234 + int(chunk_id))
    tsdf = 2.0 * torch.randn(*dims, generator=g)
    return tsdf.permute(2, 1, 0).contiguous().view(-1)


def synth_state_dict(shapes, seed=0, gains=None):
    """shapes: {name: shape}; gains: {substring: factor} applied to matching names."""
    out = {}
    for name in shapes:
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * int(seed)) & 0x7FFFFFFF)
        if name.endswith("weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
        elif name.endswith("bias"):
            wname = name[:-4] + "weight"
            fan_in = 1
            for d in tuple(shapes[wname])[1:]:
                fan_in *= d
        else:
            fan_in = 1
        bound = 1.0 / (fan_in ** 0.5)
        t = (torch.rand(*shape, generator=g) * 2.0 - 1.0) * bound
        for key, f in (gains or {}).items():
            if key in name:
                t = t * f
        out[name] = t
    return out


DEFAULT_GAINS_LATER = object()      # placeholder default: DEFAULT_GAINS is defined further down


def synth_enet_state_dict(shapes, seed=0, prefix=""):
    """Seeded weights for the ENet tree (sis3d.nets.enet / lib/nets/enet.py): conv weights / biases as synth_state_dict,
    BatchNorm statistics in a sane range (running_var > 0), BN scale near 1, PReLU slopes in (0.1, 0.4).  Keys are drawn
    from generators seeded by the key WITHOUT `prefix`, so the stand-alone encoder and the `image_enet_*` copies inside a
    Network get the same tensors."""
    out = {}
    names = set(shapes)
    for name in shapes:
        shape = tuple(shapes[name])
        key = name[len(prefix):] if prefix and name.startswith(prefix) else name
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * int(seed) + 17) & 0x7FFFFFFF)
        stem = name.rsplit(".", 1)[0]
        is_bn = (stem + ".running_mean") in names
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_var":
            t = torch.rand(*shape, generator=g) + 0.5
        elif leaf == "running_mean":
            t = torch.rand(*shape, generator=g) - 0.5
        elif is_bn and leaf == "weight":
            t = torch.rand(*shape, generator=g) + 0.5
        elif is_bn and leaf == "bias":
            t = (torch.rand(*shape, generator=g) - 0.5) * 0.2
        elif leaf == "weight" and len(shape) == 1:             # PReLU slopes
            t = torch.rand(*shape, generator=g) * 0.3 + 0.1
        elif leaf == "weight":
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = (torch.rand(*shape, generator=g) * 2.0 - 1.0) / fan_in ** 0.5
        else:                                                   # conv bias
            t = (torch.rand(*shape, generator=g) - 0.5) * 0.2
        out[name] = t
    return out


def synth_checkpoint(shapes, seed=0, gains=DEFAULT_GAINS_LATER):
    """seeded weights for a whole 3D-SIS checkpoint: the `image_enet_*` entries (2D encoder: BatchNorm statistics, PReLU
    slopes) from synth_enet_state_dict, everything else from synth_state_dict"""
    e = {k: v for k, v in shapes.items() if k.startswith("image_enet_")}
    r = {k: v for k, v in shapes.items() if not k.startswith("image_enet_")}
    sd = synth_state_dict(r, seed=seed, gains=DEFAULT_GAINS if gains is DEFAULT_GAINS_LATER else gains)
    sd.update(synth_enet_state_dict(e, seed=seed))
    return sd


def synth_images(chunk_id=0, n_views=5, image_hw=(256, 328)):
    """seeded colour views (V,3,H,W), roughly mean/std normalised like the dataloader's output (dataloader.py:26-31)"""
    g = torch.Generator().manual_seed(2468 + int(chunk_id))
    return torch.randn(n_views, 3, *image_hw, generator=g)


# spreads RPN scores away from 0.5 and lets a few class scores pass CLASS_THRESH
DEFAULT_GAINS = {"rpn_cls_score_net": 8.0, "classifier_cls_score_net.weight": 150.0}


def synth_views(chunk_id=0, n_views=5, n_per_view=3000, channels=128, image_hw=(32, 41), dims=CHUNK_DIMS):
    """feature maps (V,C,h,w) + packed index lists (V,nvox+1) int64."""
    g = torch.Generator().manual_seed(4321 + int(chunk_id))
    nvox = dims[0] * dims[1] * dims[2]
    npix = image_hw[0] * image_hw[1]
    feats = torch.randn(n_views, channels, *image_hw, generator=g)
    i3d = torch.zeros(n_views, nvox + 1, dtype=torch.int64)
    i2d = torch.zeros(n_views, nvox + 1, dtype=torch.int64)
    for v in range(n_views):
        n = min(int(n_per_view), nvox)
        vox = torch.randperm(nvox, generator=g)[:n].sort().values
        pix = torch.randint(0, npix, (n,), generator=g)
        i3d[v, 0] = n
        i2d[v, 0] = n
        i3d[v, 1:1 + n] = vox
        i2d[v, 1:1 + n] = pix
    return feats, i3d, i2d


def synth_cameras(chunk_id=0, n_views=5, dims=CHUNK_DIMS, voxel_size=0.046875, image_hw=(32, 41), depth_range=(1.0, 3.5)):
    """Seeded camera rig for ProjectionHelper.compute_projection: depth maps (V,h,w), camera_to_world (V,4,4),
    world_to_grid (V,4,4).  Cameras stand 1-2.5 m outside a random face of the volume and look at a jittered point
    inside it; ~10 % of the depth pixels are invalid (0), as in sensor data."""
    g = torch.Generator().manual_seed(977 + int(chunk_id))
    ext = torch.tensor([float(d) for d in dims]) * voxel_size
    depths, c2ws, w2gs = [], [], []
    for _ in range(n_views):
        origin = (torch.rand(3, generator=g) - 0.5) * 2.0               # world position of grid voxel (0,0,0)
        w2g = torch.eye(4)
        w2g[0, 0] = w2g[1, 1] = w2g[2, 2] = 1.0 / voxel_size
        w2g[:3, 3] = -origin / voxel_size
        target = origin + ext * (0.25 + 0.5 * torch.rand(3, generator=g))
        axis = int(torch.randint(0, 3, (1,), generator=g))
        side = float(torch.randint(0, 2, (1,), generator=g)) * 2.0 - 1.0
        pos = origin + ext * torch.rand(3, generator=g)
        pos[axis] = origin[axis] + (ext[axis] if side > 0 else 0.0) + side * (1.0 + 1.5 * float(torch.rand(1, generator=g)))
        fwd = target - pos
        fwd = fwd / fwd.norm()
        up = torch.tensor([0.0, 1.0, 0.0]) if abs(float(fwd[1])) < 0.9 else torch.tensor([1.0, 0.0, 0.0])
        right = torch.linalg.cross(up, fwd)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        depth = depth_range[0] + (depth_range[1] - depth_range[0]) * torch.rand(*image_hw, generator=g)
        depth[torch.rand(*image_hw, generator=g) < 0.1] = 0.0
        depths.append(depth)
        c2ws.append(c2w)
        w2gs.append(w2g)
    return torch.stack(depths), torch.stack(c2ws), torch.stack(w2gs)
