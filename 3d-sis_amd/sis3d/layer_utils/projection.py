"""Mirror of lib/layer_utils/projection.py: `ProjectionHelper` (:6-121) and `Projection` (:124-136)."""
import torch

from .. import ops
from .._lib import Sis3dError


class ProjectionHelper(object):
    """Same constructor and `compute_projection(depth, camera_to_world, world_to_grid)` contract as the reference
    (projection.py:6-13, 52-121): returns (lin_indices_3d, lin_indices_2d), int64 (nvox+1,) tensors on the depth
    map's device with slot 0 = count, or None when no voxel projects into the view.

    The per-voxel work (coordinates, the two 4x4 transforms, pinhole projection, depth test, ordered compaction) runs
    in HIP kernels (csrc/frustum.hip).  The host prepares per view what the reference also prepares once per call:
    the two matrix inverses and the frustum's voxel AABB (8 corner points, :27-49) -- 40 floats.
    `compute_projection_views` is the batched form of the reference call sites' list comprehension
    (lib/model/trainval.py:464-465, 663-667): one launch sequence for all views of a chunk, no host sync.

    Integer `/` of projection.py:68-70,80-82 is floor division (torch 0.4.1 semantics, requirements.txt:1)."""

    def __init__(self, intrinsic, depth_min, depth_max, image_dims, volume_dims, voxel_size):
        self.intrinsic = intrinsic
        self.depth_min = depth_min
        self.depth_max = depth_max
        self.image_dims = image_dims
        self.volume_dims = volume_dims
        self.voxel_size = voxel_size

    # -- host geometry (tiny, torch CPU) -------------------------------------------------------------------------------
    def depth_to_skeleton(self, ux, uy, depth):
        k = self.intrinsic
        return torch.Tensor([depth * ((ux - k[0][2]) / k[0][0]), depth * ((uy - k[1][2]) / k[1][1]), depth])

    def skeleton_to_depth(self, p):
        k = self.intrinsic
        return torch.Tensor([(p[0] * k[0][0]) / p[2] + k[0][2], (p[1] * k[1][1]) / p[2] + k[1][2], p[2]])

    def compute_frustum_bounds(self, world_to_grid, camera_to_world):
        """Grid-space AABB of the view frustum between depth_min and depth_max (projection.py:27-49)."""
        c2w = camera_to_world.detach().float().cpu()
        w2g = world_to_grid.detach().float().cpu()
        w, h = self.image_dims[0] - 1, self.image_dims[1] - 1
        pix = ((0, 0), (w, 0), (w, h), (0, h))
        cam = torch.ones(8, 4, 1)
        for j, d in enumerate((self.depth_min, self.depth_max)):
            for i, (ux, uy) in enumerate(pix):
                cam[4 * j + i, :3, 0] = self.depth_to_skeleton(ux, uy, d)
        world = torch.bmm(c2w.expand(8, 4, 4).contiguous(), cam)
        w2g8 = w2g.expand(8, 4, 4).contiguous()
        lo = torch.round(torch.bmm(w2g8, torch.floor(world)))[:, :3, 0]
        hi = torch.round(torch.bmm(w2g8, torch.ceil(world)))[:, :3, 0]
        both = torch.cat([lo, hi], 0)
        return both.min(0)[0], both.max(0)[0]

    def view_params(self, camera_to_world, world_to_grid):
        """(40,) fp32 CPU row of include/sis3d.h's view_params for one view."""
        c2w = camera_to_world.detach().float().cpu()
        w2g = world_to_grid.detach().float().cpu()
        bmin, bmax = self.compute_frustum_bounds(w2g, c2w)
        dims = torch.tensor([float(v) for v in self.volume_dims])
        row = torch.zeros(ops.VIEW_PARAM_FLOATS)
        row[0:16] = torch.inverse(w2g).reshape(-1)          # grid_to_world  (:58)
        row[16:32] = torch.inverse(c2w).reshape(-1)         # world_to_camera (:57)
        row[32:35] = torch.maximum(bmin, torch.zeros(3))    # :60
        row[35:38] = torch.minimum(bmax, dims)              # :61
        return row

    # -- device ------------------------------------------------------------------------------------------------------
    def compute_projection_views(self, depths, cameras_to_world, worlds_to_grid, out=None):
        """depths (V,H,W) cuda; poses / world2grid (V,4,4) any device -> (lin3d, lin2d) (V,nvox+1) int64 cuda.
        A view with count 0 is the reference's None (lib/model/trainval.py:671-676 turns those into killing_inds)."""
        if not depths.is_cuda:
            raise Sis3dError("compute_projection: depth maps must be CUDA tensors (no CPU path)")
        V = depths.shape[0]
        rows = torch.stack([self.view_params(cameras_to_world[v], worlds_to_grid[v]) for v in range(V)])
        if torch.cuda.is_current_stream_capturing():
            raise Sis3dError("compute_projection_views uploads host geometry; call it outside graph capture")
        params = rows.to(depths.device)
        return ops.compute_projection(depths, params, self.volume_dims, self.image_dims, self.intrinsic, self.depth_min,
                                      self.depth_max, self.voxel_size, out=out)

    def compute_projection(self, depth, camera_to_world, world_to_grid):
        """A host depth map (the reference's big-volume branch, lib/model/trainval.py:664-665) is uploaded, processed
        on the GPU and the lists returned on the host: the result lives where `depth` lives, as in the reference."""
        d = depth if depth.is_cuda else depth.cuda()
        l3, l2 = self.compute_projection_views(d.reshape(1, self.image_dims[1], self.image_dims[0]),
                                               camera_to_world.reshape(1, 4, 4), world_to_grid.reshape(1, 4, 4))
        if int(l3[0, 0].item()) == 0:                       # the reference's three `return None` exits (:75-77,95-97,105-107)
            return None
        return (l3[0], l2[0]) if depth.is_cuda else (l3[0].cpu(), l2[0].cpu())


def prepare_projection(blobs, cfg, helper=None):
    """The colour-projection block every TEST/benchmark loop of the reference runs before net.forward
    (lib/model/trainval.py:659-683, 795-819): builds blobs['proj_ind_3d'/'proj_ind_2d'] from
    blobs['nearest_images']['depths'/'poses'/'world2grid'][0] and returns killing_inds (views with no visible voxel;
    the stacked lists hold only the surviving views, as in the reference).  All views go through one device launch
    sequence; one V-int readback replaces the reference's 3 x V `.any()` syncs."""
    grid_shape = [int(v) for v in blobs["data"].shape[-3:]]
    helper = helper or ProjectionHelper(cfg.INTRINSIC, cfg.PROJ_DEPTH_MIN, cfg.PROJ_DEPTH_MAX, cfg.DEPTH_SHAPE, grid_shape,
                                        cfg.VOXEL_SIZE)
    ni = blobs["nearest_images"]
    depths = ni["depths"][0]
    depths = depths if torch.is_tensor(depths) else torch.stack(list(depths))
    V = min(depths.shape[0], len(ni["poses"][0]), len(ni["world2grid"][0]))      # zip() truncation
    l3, l2 = helper.compute_projection_views(depths[:V].cuda(), ni["poses"][0], ni["world2grid"][0])
    counts = l3[:, 0].cpu().tolist()
    killing_inds = [v for v, n in enumerate(counts) if n == 0]
    if len(killing_inds) == V:
        raise Sis3dError("no view projects into the chunk (the reference fails in zip(*[]) here, trainval.py:680)")
    if killing_inds:
        keep = torch.tensor([v for v in range(V) if counts[v] > 0], device=l3.device)
        l3, l2 = l3.index_select(0, keep), l2.index_select(0, keep)
    blobs["proj_ind_3d"] = [l3]
    blobs["proj_ind_2d"] = [l2]
    return killing_inds


class Projection(torch.autograd.Function):
    """`Projection.apply(label, lin_indices_3d, lin_indices_2d, volume_dims)` -> (C,Z,Y,X) fp32, an autograd Function as in the
    reference (projection.py:124-153).

    label: (C,h,w) or (h,w) feature map; the index tensors are the packed int64 lists of
    ProjectionHelper.compute_projection (slot 0 = count).  backward: sis3d_projection_backward."""

    @staticmethod
    def forward(ctx, label, lin_indices_3d, lin_indices_2d, volume_dims):
        ctx.save_for_backward(lin_indices_3d, lin_indices_2d)
        ctx.image_hw = tuple(label.shape[-2:])
        ctx.label_dim = label.dim()
        return ops.projection(label, lin_indices_3d, lin_indices_2d, volume_dims)

    @staticmethod
    def backward(ctx, grad_output):
        i3d, i2d = ctx.saved_tensors
        g = ops.projection_backward(grad_output, i3d, i2d, ctx.image_hw)
        if ctx.label_dim == 2:
            g = g[0]
        return g, None, None, None
