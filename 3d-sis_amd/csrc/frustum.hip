// Voxel -> pixel visibility lists on gfx950: the device form of
// ProjectionHelper.compute_projection (lib/layer_utils/projection.py:52-121), all views of a
// chunk in one launch sequence.
//
// The reference materialises arange(nvox), a (4, nvox) coordinate matrix, two torch.mm products
// and five boolean-mask compactions per view (~25 temporaries, three .any() host syncs).  Here a
// voxel's whole test is ~40 flops in registers:
//   coords -> frustum AABB test -> q = grid_to_world * (x,y,z,1) -> p = world_to_camera * q
//   -> u = (p.x*fx)/p.z + cx, v likewise -> round-half-even -> image bounds -> depth lookup
//   -> depth_min <= d <= depth_max and |d - p.z| <= voxel_size
// and the ordered compaction (the lists are ascending in the linear voxel index) is
// count-per-1024-voxel-block -> one scan block per view -> recompute + scatter.  No host sync:
// the count lands in slot 0 of both lists exactly as the reference packs it.
//
// Bit-exactness: the matrix products are the fp32 fma chain in k order that torch's CPU mm
// performs for a (4x4)@(4xN) product (acc = a0*b0; acc = fma(a_k, b_k, acc)), the division is
// IEEE-correct, torch.round is rintf.  Compiled with -ffp-contract=off: every fma is explicit.
#include "common.h"

namespace {

constexpr int FR_THREADS = 256;
constexpr int FR_ITERS = 4;
constexpr int FR_BLOCK_VOX = FR_THREADS * FR_ITERS;     // 1024 voxels per block, iteration-major order
constexpr int FR_WAVES = FR_THREADS / 64;

struct FrustumArgs {
    const float *depth;        // [V][W*H]
    const float *view;         // [V][SIS3D_VIEW_PARAM_FLOATS]: g2w(16) w2c(16) bmin(3) bmax(3) pad(2)
    int V, X, Y, Z, W, H;
    float fx, fy, cx, cy, depth_min, depth_max, voxel_size;
    int64_t nvox;
    int nblk;
};

__device__ __forceinline__ float dot4(const float *__restrict__ m, float b0, float b1, float b2, float b3)
{
    float acc = m[0] * b0;
    acc = fmaf(m[1], b1, acc);
    acc = fmaf(m[2], b2, acc);
    acc = fmaf(m[3], b3, acc);
    return acc;
}

// pixel index of voxel `lin` in this view, or -1
__device__ __forceinline__ int voxel_pixel(const FrustumArgs &a, const float *__restrict__ vp, const float *__restrict__ depth,
                                           int64_t lin)
{
    const int64_t xy = (int64_t)a.X * a.Y;
    const int z = (int)(lin / xy);
    const int r = (int)(lin - (int64_t)z * xy);
    const int y = r / a.X, x = r - y * a.X;
    const float fx_ = (float)x, fy_ = (float)y, fz_ = (float)z;
    if (!(fx_ >= vp[32] && fy_ >= vp[33] && fz_ >= vp[34] && fx_ < vp[35] && fy_ < vp[36] && fz_ < vp[37])) return -1;
    const float q0 = dot4(vp + 0, fx_, fy_, fz_, 1.0f), q1 = dot4(vp + 4, fx_, fy_, fz_, 1.0f);
    const float q2 = dot4(vp + 8, fx_, fy_, fz_, 1.0f), q3 = dot4(vp + 12, fx_, fy_, fz_, 1.0f);
    const float p0 = dot4(vp + 16, q0, q1, q2, q3), p1 = dot4(vp + 20, q0, q1, q2, q3), p2 = dot4(vp + 24, q0, q1, q2, q3);
    const float u = rintf((p0 * a.fx) / p2 + a.cx);
    const float v = rintf((p1 * a.fy) / p2 + a.cy);
    if (!(u >= 0.0f && v >= 0.0f && u < (float)a.W && v < (float)a.H)) return -1;   // NaN/inf fall out here
    const int pix = (int)v * a.W + (int)u;
    const float d = depth[pix];
    if (!(d >= a.depth_min && d <= a.depth_max && fabsf(d - p2) <= a.voxel_size)) return -1;
    return pix;
}

// SCATTER=false: counts[view][blk] = visible voxels of the block.
// SCATTER=true : counts holds the exclusive scan; write the packed lists and zero the unused tail.
template <bool SCATTER>
__global__ void __launch_bounds__(FR_THREADS) frustum_kernel(FrustumArgs a, int64_t *__restrict__ counts,
                                                             int64_t *__restrict__ lin3d, int64_t *__restrict__ lin2d)
{
    __shared__ int wave_cnt[FR_ITERS][FR_WAVES];
    __shared__ float vp[SIS3D_VIEW_PARAM_FLOATS];
    const int view = blockIdx.y, blk = blockIdx.x;
    if (threadIdx.x < SIS3D_VIEW_PARAM_FLOATS) vp[threadIdx.x] = a.view[(int64_t)view * SIS3D_VIEW_PARAM_FLOATS + threadIdx.x];
    __syncthreads();
    const float *depth = a.depth + (int64_t)view * a.W * a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blk * FR_BLOCK_VOX;
    int pix[FR_ITERS];
    int before[FR_ITERS];                                   // visible voxels earlier in my wave, this iteration
#pragma unroll
    for (int it = 0; it < FR_ITERS; ++it) {
        const int64_t lin = base + it * FR_THREADS + threadIdx.x;
        pix[it] = lin < a.nvox ? voxel_pixel(a, vp, depth, lin) : -1;
        const unsigned long long bal = __ballot(pix[it] >= 0);
        before[it] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[it][wave] = __popcll(bal);
    }
    __syncthreads();
    if (!SCATTER) {
        if (threadIdx.x == 0) {
            int total = 0;
            for (int i = 0; i < FR_ITERS * FR_WAVES; ++i) total += (&wave_cnt[0][0])[i];
            counts[(int64_t)view * a.nblk + blk] = total;
        }
        return;
    }
    int64_t *o3 = lin3d + (int64_t)view * (a.nvox + 1), *o2 = lin2d + (int64_t)view * (a.nvox + 1);
    const int64_t blk_off = counts[(int64_t)view * a.nblk + blk];
    const int64_t n = o3[0];                                // written by the scan kernel
    int run = 0;
#pragma unroll
    for (int it = 0; it < FR_ITERS; ++it) {
        int pre = run;
        for (int w = 0; w < FR_WAVES; ++w) {
            if (w < wave) pre += wave_cnt[it][w];
            run += wave_cnt[it][w];
        }
        const int64_t lin = base + it * FR_THREADS + threadIdx.x;
        if (pix[it] >= 0) {
            const int64_t slot = 1 + blk_off + pre + before[it];
            o3[slot] = lin;
            o2[slot] = pix[it];
        }
        if (lin < a.nvox && lin >= n) { o3[1 + lin] = 0; o2[1 + lin] = 0; }   // tail: defined zeros (reference: uninitialised)
    }
}

// one block per view: in-place exclusive scan of the block counts, total -> slot 0 of both lists
__global__ void __launch_bounds__(1024) frustum_scan_kernel(int64_t *__restrict__ counts, int nblk, int64_t nvox,
                                                            int64_t *__restrict__ lin3d, int64_t *__restrict__ lin2d)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    int64_t *c = counts + (int64_t)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int64_t v = i < nblk ? c[i] : 0;
        int64_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t pre = carry;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (i < nblk) c[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lin3d[(int64_t)blockIdx.x * (nvox + 1)] = carry;
        lin2d[(int64_t)blockIdx.x * (nvox + 1)] = carry;
    }
}

} // namespace

extern "C" size_t sis3d_compute_projection_workspace_bytes(int V, int64_t nvox)
{
    const int64_t nblk = (nvox + FR_BLOCK_VOX - 1) / FR_BLOCK_VOX;
    return sizeof(int64_t) * (size_t)(V > 0 ? V : 0) * (size_t)nblk;
}

extern "C" int sis3d_compute_projection(const float *depth, const float *view_params, int V, int X, int Y, int Z, int W, int H,
                                        float fx, float fy, float cx, float cy, float depth_min, float depth_max,
                                        float voxel_size, int64_t *lin3d, int64_t *lin2d, void *ws, size_t ws_bytes,
                                        sis3d_stream_t stream)
{
    if (!depth || !view_params || !lin3d || !lin2d || V <= 0 || X <= 0 || Y <= 0 || Z <= 0 || W <= 0 || H <= 0)
        return SIS3D_EINVAL;
    const int64_t nvox = (int64_t)X * Y * Z;
    if (V > 65535 || nvox > ((int64_t)1 << 40)) return SIS3D_EUNSUPPORTED;
    if (!ws || ws_bytes < sis3d_compute_projection_workspace_bytes(V, nvox)) return SIS3D_EWORKSPACE;
    FrustumArgs a;
    a.depth = depth; a.view = view_params;
    a.V = V; a.X = X; a.Y = Y; a.Z = Z; a.W = W; a.H = H;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    a.depth_min = depth_min; a.depth_max = depth_max; a.voxel_size = voxel_size;
    a.nvox = nvox;
    a.nblk = (int)((nvox + FR_BLOCK_VOX - 1) / FR_BLOCK_VOX);
    hipStream_t st = as_stream(stream);
    int64_t *counts = (int64_t *)ws;
    hipLaunchKernelGGL(frustum_kernel<false>, dim3(a.nblk, V), dim3(FR_THREADS), 0, st, a, counts, lin3d, lin2d);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(frustum_scan_kernel, dim3(V), dim3(1024), 0, st, counts, a.nblk, nvox, lin3d, lin2d);
    rc = sis3d_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(frustum_kernel<true>, dim3(a.nblk, V), dim3(FR_THREADS), 0, st, a, counts, lin3d, lin2d);
    return sis3d_check_launch();
}
