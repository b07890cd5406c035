// 2D -> 3D back-projection of image features on gfx950.
//
// Replaces Projection.forward (lib/layer_utils/projection.py:124-136: zero-filled
// (C,Z,Y,X) volume + index_select/index_copy_) and, in the fused entry point, the
// per-view loop with pairwise stack -> view(C,-1,2) -> MaxPool1d(2) of
// lib/nets/network.py:216-239.
//
// The reference moves ~6.6 GB for a 5-view 96x48x96x128 volume (V zero fills, V
// scatters, V-1 stack+pool round trips).  Here the volume is written exactly once:
//   1. a V x nvox int32 table voxel->pixel (-1 = not visible) is filled from the packed
//      index lists (counts are read on the device: no host sync);
//   2. one pass over the OUTPUT in its own memory order computes, per voxel/channel,
//      max over the included views of (visible ? feature : 0) -- the implicit-zero rule
//      of the reference's max over zero-filled volumes -- and
//      stores it.  >98 % of the voxels are empty, so the pass is a pure streaming store
//      bound by HBM write bandwidth; features (3.4 MB) are transposed once to
//      pixel-major so a voxel's 128 channels are one coalesced 512 B read.
#include "common.h"
#include <float.h>

namespace {

constexpr int MAX_VIEWS = 64;
struct ViewIds { int n; int id[MAX_VIEWS]; };   // passed by value: no host->device copy, graph-capture safe

__global__ void proj_scatter_kernel(const float *__restrict__ feat, int C, int64_t npix, const int64_t *__restrict__ lin3d,
                                    const int64_t *__restrict__ lin2d, int64_t nvox, float *__restrict__ out)
{
    const int64_t n = lin3d[0];
    const int64_t total = n * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t % n, c = t / n;
        out[c * nvox + lin3d[1 + k]] = feat[c * npix + lin2d[1 + k]];
    }
}

// vox2pix[v][vox] = pixel index for every listed voxel of every included view
__global__ void proj_table_kernel(const int64_t *__restrict__ lin3d, const int64_t *__restrict__ lin2d, int64_t nvox,
                                  ViewIds view_ids, int32_t *__restrict__ vox2pix)
{
    const int slot = blockIdx.y;                       // position among the included views
    const int v = view_ids.id[slot];
    const int64_t *a = lin3d + (int64_t)v * (nvox + 1), *b = lin2d + (int64_t)v * (nvox + 1);
    const int64_t n = a[0];
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        vox2pix[(int64_t)slot * nvox + a[1 + k]] = (int32_t)b[1 + k];
}

// feats [V][C][npix] -> featT [slot][npix][C] for the included views
__global__ void proj_transpose_kernel(const float *__restrict__ feats, int C, int64_t npix, ViewIds view_ids,
                                      float *__restrict__ featT)
{
    __shared__ float tile[32][33];
    const int slot = blockIdx.z, v = view_ids.id[slot];
    const float *src = feats + (int64_t)v * C * npix;
    float *dst = featT + (int64_t)slot * npix * C;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j;
        const int64_t p = p0 + threadIdx.x;
        tile[j][threadIdx.x] = (c < C && p < npix) ? src[(int64_t)c * npix + p] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t p = p0 + j;
        const int c = c0 + threadIdx.x;
        if (c < C && p < npix) dst[p * C + c] = tile[threadIdx.x][j];
    }
}

// ---- single pass over the output volume --------------------------------------------
// Channels-last output (x,y,z,c): thread = one voxel x 4 channels (16 B store); lanes run over
// channels first, so a voxel's row is one coalesced segment and the vox2pix lookups are
// uniform across the 32 lanes of a voxel.  featT is pixel-major [slot][npix][C].
__global__ void proj_gather_cl_kernel(const float *__restrict__ featT, const int32_t *__restrict__ vox2pix, int nslots, int C,
                                      int64_t npix, int X, int Y, int Z, float *__restrict__ out, int64_t os_c, int64_t os_x,
                                      int64_t os_y, int64_t os_z)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int cq = C / 4;
    const int64_t total = nvox * cq;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % cq) * 4;
        const int64_t o = t / cq;                          // output-order voxel id (x*Y + y)*Z + z
        const int z = (int)(o % Z), y = (int)((o / Z) % Y), x = (int)(o / ((int64_t)Z * Y));
        const int64_t vox = ((int64_t)z * Y + y) * X + x;  // reference linear index z*X*Y + y*X + x
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        for (int s = 0; s < nslots; ++s) {
            const int32_t p = vox2pix[(int64_t)s * nvox + vox];
            if (p >= 0) {
                const float4 f = *reinterpret_cast<const float4 *>(featT + ((int64_t)s * npix + p) * C + c);
                if (cnt == 0) m = f;
                else { m.x = fmaxf(m.x, f.x); m.y = fmaxf(m.y, f.y); m.z = fmaxf(m.z, f.z); m.w = fmaxf(m.w, f.w); }
                ++cnt;
            }
        }
        if (cnt > 0 && cnt < nslots) { m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f); }
        float *dst = out + x * os_x + y * os_y + z * os_z + c * os_c;
        if (os_c == 1) *reinterpret_cast<float4 *>(dst) = m;
        else { dst[0] = m.x; dst[os_c] = m.y; dst[2 * os_c] = m.z; dst[3 * os_c] = m.w; }
    }
}

// Reference memory order (c, z, y, x): lanes run over x (the linear voxel index), so vox2pix reads
// and the stores are coalesced; features are read from the original [v][c][npix] maps.
__global__ void proj_gather_planar_kernel(const float *__restrict__ feats, ViewIds view_ids,
                                          const int32_t *__restrict__ vox2pix, int nslots, int C, int64_t npix, int X, int Y,
                                          int Z, float *__restrict__ out, int64_t os_c, int64_t os_x, int64_t os_y, int64_t os_z)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = nvox * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t vox = t % nvox;
        const int c = (int)(t / nvox);
        const int x = (int)(vox % X), y = (int)((vox / X) % Y), z = (int)(vox / ((int64_t)X * Y));
        float m = 0.f;
        int cnt = 0;
        for (int s = 0; s < nslots; ++s) {
            const int32_t p = vox2pix[(int64_t)s * nvox + vox];
            if (p >= 0) {
                const float f = feats[((int64_t)view_ids.id[s] * C + c) * npix + p];
                m = cnt == 0 ? f : fmaxf(m, f);
                ++cnt;
            }
        }
        if (cnt > 0 && cnt < nslots) m = fmaxf(m, 0.f);
        out[c * os_c + x * os_x + y * os_y + z * os_z] = m;
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

} // namespace

extern "C" int sis3d_projection_forward(const float *feat, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                                        int64_t nvox, float *out, sis3d_stream_t stream)
{
    if (!feat || !lin3d || !lin2d || !out || C <= 0 || npix <= 0 || nvox <= 0) return SIS3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (sis3d_fill32(out, 0u, sizeof(float) * (size_t)C * (size_t)nvox, st) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(proj_scatter_kernel, dim3(1024), dim3(256), 0, st, feat, C, npix, lin3d, lin2d, nvox, out);
    return sis3d_check_launch();
}

extern "C" size_t sis3d_project_views_workspace_bytes(int V, int C, int64_t npix, int64_t nvox)
{
    // vox2pix table + pixel-major feature copy
    return align256(sizeof(int32_t) * (size_t)V * (size_t)nvox) +
           align256(sizeof(float) * (size_t)V * (size_t)C * (size_t)npix);
}

extern "C" int sis3d_project_views_max(const float *feats, int V, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                                       const uint8_t *kill_host, int X, int Y, int Z, float *out, int64_t os_c, int64_t os_x,
                                       int64_t os_y, int64_t os_z, void *ws, size_t ws_bytes, sis3d_stream_t stream)
{
    if (!feats || !lin3d || !lin2d || !out || V <= 0 || C <= 0 || npix <= 0 || X <= 0 || Y <= 0 || Z <= 0) return SIS3D_EINVAL;
    const int64_t nvox = (int64_t)X * Y * Z;
    if (!ws || ws_bytes < sis3d_project_views_workspace_bytes(V, C, npix, nvox)) return SIS3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    ViewIds ids;
    ids.n = 0;
    for (int v = 0; v < V; ++v)
        if (!kill_host || !kill_host[v]) {
            if (ids.n >= MAX_VIEWS) return SIS3D_EUNSUPPORTED;
            ids.id[ids.n++] = v;
        }
    const int nslots = ids.n;
    if (nslots == 0) return SIS3D_EINVAL;                 // the reference would fail as well (sz undefined, network.py:237)
    unsigned char *base = (unsigned char *)ws;
    int32_t *vox2pix = (int32_t *)base;
    float *featT = (float *)(base + align256(sizeof(int32_t) * (size_t)V * (size_t)nvox));
    if (sis3d_fill32(vox2pix, 0xFFFFFFFFu, sizeof(int32_t) * (size_t)nslots * (size_t)nvox, st) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(proj_table_kernel, dim3(64, nslots), dim3(256), 0, st, lin3d, lin2d, nvox, ids, vox2pix);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    const bool cl = (os_c == 1) && (C % 4 == 0);
    if (cl) {
        hipLaunchKernelGGL(proj_transpose_kernel, dim3(cdiv(npix, 32), cdiv(C, 32), nslots), dim3(32, 8), 0, st, feats, C, npix,
                           ids, featT);
        rc = sis3d_check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(proj_gather_cl_kernel, dim3(4096), dim3(256), 0, st, featT, vox2pix, nslots, C, npix, X, Y, Z, out,
                           os_c, os_x, os_y, os_z);
    } else {
        hipLaunchKernelGGL(proj_gather_planar_kernel, dim3(4096), dim3(256), 0, st, feats, ids, vox2pix, nslots, C, npix, X, Y,
                           Z, out, os_c, os_x, os_y, os_z);
    }
    return sis3d_check_launch();
}

// Table + pixel-major feature rows only (no volume): the input of sis3d_conv3d_chain_projected.
extern "C" int sis3d_project_views_prepare(const float *feats, int V, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                                           const uint8_t *kill_host, int64_t nvox, int32_t *vox2pix, float *feat_rows, int *nslots_out,
                                           sis3d_stream_t stream)
{
    if (!feats || !lin3d || !lin2d || !vox2pix || !feat_rows || !nslots_out || V <= 0 || C <= 0 || npix <= 0 || nvox <= 0) return SIS3D_EINVAL;
    hipStream_t st = as_stream(stream);
    ViewIds ids;
    ids.n = 0;
    for (int v = 0; v < V; ++v)
        if (!kill_host || !kill_host[v]) {
            if (ids.n >= MAX_VIEWS) return SIS3D_EUNSUPPORTED;
            ids.id[ids.n++] = v;
        }
    *nslots_out = ids.n;
    if (ids.n == 0) return SIS3D_EINVAL;
    if (sis3d_fill32(vox2pix, 0xFFFFFFFFu, sizeof(int32_t) * (size_t)ids.n * (size_t)nvox, st) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(proj_table_kernel, dim3(64, ids.n), dim3(256), 0, st, lin3d, lin2d, nvox, ids, vox2pix);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(proj_transpose_kernel, dim3(cdiv(npix, 32), cdiv(C, 32), ids.n), dim3(32, 8), 0, st, feats, C, npix, ids,
                       feat_rows);
    return sis3d_check_launch();
}

// ---- backward of Projection (training only; lib/layer_utils/projection.py:139-153).  The reference clones grad_output, resizes
// the clone to (C, h, w) -- so pixels no voxel maps to keep the clone's LEADING elements, a quirk kept here -- and index_copy_()s
// grad_output[:, i3d[1:1+n]] into columns i2d[1:1+n]; when several voxels share a pixel the last list entry wins (the sequential
// CPU index_copy_).  Three small launches: init, winner = max list position per pixel, copy by the winners.
namespace {

__global__ __launch_bounds__(256) void projb_init_kernel(const float *__restrict__ gout, int64_t gout_elems, int64_t n_out, float *__restrict__ gl,
                                                         int32_t *__restrict__ winner, int64_t npix)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_out) gl[i] = i < gout_elems ? gout[i] : 0.0f;
    if (i < npix) winner[i] = -1;
}

__global__ __launch_bounds__(256) void projb_winner_kernel(const int64_t *__restrict__ i3d, const int64_t *__restrict__ i2d, int64_t nvox,
                                                           int64_t npix, int32_t *__restrict__ winner)
{
    const int64_t n = min(max(i3d[0], (int64_t)0), nvox);
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t p = i2d[1 + k];
    if (p >= 0 && p < npix) atomicMax(winner + p, (int32_t)k);
}

__global__ __launch_bounds__(256) void projb_copy_kernel(const float *__restrict__ gout, const int64_t *__restrict__ i3d,
                                                         const int64_t *__restrict__ i2d, int64_t nvox, int64_t npix, int C,
                                                         const int32_t *__restrict__ winner, float *__restrict__ gl)
{
    const int64_t n = min(max(i3d[0], (int64_t)0), nvox);
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t k = t / C;
    const int c = (int)(t % C);
    if (k >= n) return;
    const int64_t p = i2d[1 + k], v = i3d[1 + k];
    if (p < 0 || p >= npix || v < 0 || v >= nvox || winner[p] != (int32_t)k) return;
    gl[(int64_t)c * npix + p] = gout[(int64_t)c * nvox + v];
}

} // namespace

extern "C" size_t sis3d_projection_backward_workspace_bytes(int64_t npix) { return npix > 0 ? (size_t)npix * 4 : 0; }

extern "C" int sis3d_projection_backward(const float *grad_out, int C, int64_t nvox, const int64_t *lin3d, const int64_t *lin2d,
                                         int64_t npix, float *grad_label, void *workspace, size_t workspace_bytes, sis3d_stream_t stream)
{
    if (!grad_out || !lin3d || !lin2d || !grad_label || C <= 0 || nvox <= 0 || npix <= 0) return SIS3D_EINVAL;
    if (!workspace || workspace_bytes < (size_t)npix * 4) return SIS3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int64_t n_out = (int64_t)C * npix;
    int32_t *winner = (int32_t *)workspace;
    hipLaunchKernelGGL(projb_init_kernel, dim3((unsigned)((max(n_out, npix) + 255) / 256)), dim3(256), 0, st, grad_out, (int64_t)C * nvox, n_out,
                       grad_label, winner, npix);
    hipLaunchKernelGGL(projb_winner_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, lin3d, lin2d, nvox, npix, winner);
    hipLaunchKernelGGL(projb_copy_kernel, dim3((unsigned)((nvox * C + 255) / 256)), dim3(256), 0, st, grad_out, lin3d, lin2d, nvox, npix, C,
                       winner, grad_label);
    return sis3d_check_launch();
}
