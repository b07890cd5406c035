// 3D NMS on gfx950: wave64 ballot IoU bit-matrix + on-device greedy sweep.
//
// Replaces lib/layer_utils/nms/src/cuda/nms_kernel.cu:11-94 (bit matrix) and the
// HOST sweep of lib/layer_utils/nms/src/nms_cuda.c:44-59.  Arithmetic is the
// reference's binary32 sequence (devIoU, nms_kernel.cu:11-31 == cpu_nms,
// pth_nms.py:22-40); this file is built with -ffp-contract=off so no FMA is
// formed and keep lists are bit-identical to the CPU path.
//
// Design (MI355X): a 64x64 tile of the matrix is exactly one wavefront: lane =
// column box, the 64 row boxes are walked with the row box broadcast from LDS,
// and __ballot() of the 64 lanes IS the 64-bit mask word -- no per-thread bit
// loop.  The sweep is a single workgroup: per 64-box block the in-block
// decisions are resolved by one wave with v_readlane on the diagonal words (no
// memory traffic), then all lanes OR the kept rows into the running remove
// vector held in LDS.  For n <= SMALL_N the bit matrix never leaves LDS and the
// whole NMS is ONE launch with no global scratch.
#include "common.h"

namespace {

struct Box { float x1, y1, z1, x2, y2, z2, area, pad; };   // area = (x2-x1+1)*(y2-y1+1)*(z2-z1+1), computed once per box

__device__ __forceinline__ float iou3d(const Box &a, const Box &b)
{
    float left = fmaxf(a.x1, b.x1), top = fmaxf(a.y1, b.y1), front = fmaxf(a.z1, b.z1);
    float right = fminf(a.x2, b.x2), bottom = fminf(a.y2, b.y2), back = fminf(a.z2, b.z2);
    float w = fmaxf(right - left + 1.0f, 0.0f);
    float h = fmaxf(bottom - top + 1.0f, 0.0f);
    float l = fmaxf(back - front + 1.0f, 0.0f);
    float inter = w * h * l;
    return inter / (a.area + b.area - inter);              // same binary32 values as recomputing the areas per pair
}

template <bool INDIRECT>
__device__ __forceinline__ Box load_box(const float *boxes, const int64_t *order, int i)
{
    const float *p = boxes + 6 * (INDIRECT ? order[i] : (int64_t)i);
    Box b;
    b.x1 = p[0]; b.y1 = p[1]; b.z1 = p[2]; b.x2 = p[3]; b.y2 = p[4]; b.z2 = p[5];
    b.area = (b.x2 - b.x1 + 1.0f) * (b.y2 - b.y1 + 1.0f) * (b.z2 - b.z1 + 1.0f);
    b.pad = 0.0f;
    return b;
}

// One wave computes the 64 mask words of tile (rb, cb): word for row 64*rb+r has bit j set
// iff box 64*cb+j is suppressed by row box, j > i.  rows[] = the 64 row boxes in LDS.
__device__ __forceinline__ uint64_t tile_word(const Box *rows, const Box &col, bool col_valid, int rb, int cb, int n,
                                              float thresh, int lane)
{
    uint64_t mine = 0;
    const int col_idx = 64 * cb + lane;
    const int nrows = min(64, n - 64 * rb);
    for (int r = 0; r < nrows; ++r) {
        const Box a = rows[r];                       // LDS broadcast (same address in all lanes)
        const float v = iou3d(a, col);
        const bool sup = col_valid && (col_idx > 64 * rb + r) && !(v <= thresh);
        const uint64_t word = __ballot(sup);
        if (lane == r) mine = word;
    }
    return mine;
}

// ---------------------------------------------------------------- global-matrix path
template <bool INDIRECT>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order, int n,
                                                      float thresh, uint64_t *__restrict__ mask)
{
    const int cb = blockIdx.x, rb = blockIdx.y, lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    __shared__ Box rows[64];
    const int ri = 64 * rb + lane;
    if (cb < rb) {                                   // strictly-lower tiles are never read by the sweep
        if (ri < n) mask[(size_t)ri * col_blocks + cb] = 0;
        return;
    }
    if (ri < n) rows[lane] = load_box<INDIRECT>(boxes, order, ri);
    __syncthreads();
    const int ci = 64 * cb + lane;
    Box col = {0, 0, 0, 0, 0, 0, 1, 0};
    if (ci < n) col = load_box<INDIRECT>(boxes, order, ci);
    const uint64_t w = tile_word(rows, col, ci < n, rb, cb, n, thresh, lane);
    if (ri < n) mask[(size_t)ri * col_blocks + cb] = w;
}

// Greedy sweep, one workgroup of 256 threads.  remv[] (one word per column block) lives in LDS.
template <bool SELECT>
__global__ __launch_bounds__(256) void nms_sweep_kernel(const uint64_t *__restrict__ mask, int n, int max_keep,
                                                        int64_t *__restrict__ keep, int32_t *__restrict__ num_keep,
                                                        // SELECT outputs
                                                        const float *__restrict__ boxes_all, const float *__restrict__ level_all,
                                                        const float *__restrict__ scores_sorted, const int64_t *__restrict__ order,
                                                        float *__restrict__ rois, float *__restrict__ roi_scores,
                                                        float *__restrict__ roi_levels)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *remv = (uint64_t *)smem;                 // [col_blocks]
    __shared__ uint64_t s_kept;
    __shared__ int s_nk;
    const int col_blocks = (n + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int c = tid; c < col_blocks; c += blockDim.x) remv[c] = 0;
    if (tid == 0) s_nk = 0;
    __syncthreads();
    const int limit = max_keep > 0 ? max_keep : n;
    for (int b = 0; b < col_blocks; ++b) {
        if (wid == 0) {
            const int ri = 64 * b + lane;
            const uint64_t diag = ri < n ? mask[(size_t)ri * col_blocks + b] : 0;
            uint64_t cur = remv[b];
            const int nrows = min(64, n - 64 * b);
            const int nk0 = s_nk;
            int nk = nk0;
            uint64_t kept = 0;
            for (int i = 0; i < nrows && nk < limit; ++i) {
                if (!((cur >> i) & 1ULL)) {
                    kept |= 1ULL << i;
                    ++nk;
                    // v_readlane: broadcast the diagonal word of row i
                    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)diag, i);
                    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(diag >> 32), i);
                    cur |= ((uint64_t)hi << 32) | lo;
                }
            }
            // survivors of this block: lane i writes its own slot (prefix popcount gives the rank)
            if ((kept >> lane) & 1ULL) {
                const int rank = nk0 + __popcll(kept & ((1ULL << lane) - 1ULL));
                const int i = 64 * b + lane;
                keep[rank] = i;
                if (SELECT) {
                    const int64_t src = order[i];
                    for (int k = 0; k < 6; ++k) rois[6 * rank + k] = boxes_all[6 * src + k];
                    roi_scores[rank] = scores_sorted[i];
                    roi_levels[rank] = level_all[src];
                }
            }
            if (lane == 0) { s_kept = kept; s_nk = nk; }
        }
        __syncthreads();
        const uint64_t kept = s_kept;
        if (s_nk >= limit) break;
        // fold the kept rows of block b into remv for the later column blocks
        for (int c = b + 1 + tid; c < col_blocks; c += blockDim.x) {
            uint64_t acc = remv[c], k = kept;
            while (k) {
                const int i = __builtin_ctzll(k);
                k &= k - 1;
                acc |= mask[(size_t)(64 * b + i) * col_blocks + c];
            }
            remv[c] = acc;
        }
        __syncthreads();
    }
    __syncthreads();
    const int nk = s_nk;
    if (tid == 0) num_keep[0] = nk;
    if (SELECT) {                                      // zero-fill the padded rows
        for (int r = nk + tid; r < max_keep; r += blockDim.x) {
            for (int k = 0; k < 6; ++k) rois[6 * r + k] = 0.0f;
            roi_scores[r] = 0.0f;
            roi_levels[r] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------- single-launch path (matrix in LDS)
constexpr int SMALL_N = 960;      // 960 x 15 words x 8 B = 112 KiB matrix + 30 KiB boxes of the 160 KiB LDS

template <bool INDIRECT, bool SELECT>
__global__ __launch_bounds__(1024) void nms_small_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order, int n,
                                                        float thresh, int max_keep, int64_t *__restrict__ keep,
                                                        int32_t *__restrict__ num_keep, const float *__restrict__ level_all,
                                                        const float *__restrict__ scores_sorted, float *__restrict__ rois,
                                                        float *__restrict__ roi_scores, float *__restrict__ roi_levels)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int col_blocks = (n + 63) / 64;
    const int npad = col_blocks * 64;
    Box *sbox = (Box *)smem;                                        // [npad]
    uint64_t *smask = (uint64_t *)(smem + (size_t)npad * sizeof(Box)); // [npad][col_blocks]
    uint64_t *remv = smask + (size_t)npad * col_blocks;               // [col_blocks]
    __shared__ uint64_t s_kept;
    __shared__ int s_nk;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < npad; i += blockDim.x) {
        Box b = {0, 0, 0, 0, 0, 0, 1, 0};
        if (i < n) b = load_box<INDIRECT>(boxes, order, i);
        sbox[i] = b;
    }
    for (int c = tid; c < col_blocks; c += blockDim.x) remv[c] = 0;
    if (tid == 0) s_nk = 0;
    __syncthreads();
    // upper-triangular tiles, one wave each
    const int ntiles = col_blocks * (col_blocks + 1) / 2;
    for (int t = wid; t < ntiles; t += nw) {
        int rb = 0, rem = t;
        while (rem >= col_blocks - rb) { rem -= col_blocks - rb; ++rb; }
        const int cb = rb + rem;
        const Box col = sbox[64 * cb + lane];
        const uint64_t w = tile_word(sbox + 64 * rb, col, 64 * cb + lane < n, rb, cb, n, thresh, lane);
        smask[(size_t)(64 * rb + lane) * col_blocks + cb] = w;
    }
    __syncthreads();
    const int limit = max_keep > 0 ? max_keep : n;
    for (int b = 0; b < col_blocks; ++b) {
        if (wid == 0) {
            const uint64_t diag = smask[(size_t)(64 * b + lane) * col_blocks + b];
            uint64_t cur = remv[b];
            const int nrows = min(64, n - 64 * b);
            const int nk0 = s_nk;
            int nk = nk0;
            uint64_t kept = 0;
            for (int i = 0; i < nrows && nk < limit; ++i) {
                if (!((cur >> i) & 1ULL)) {
                    kept |= 1ULL << i;
                    ++nk;
                    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)diag, i);
                    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(diag >> 32), i);
                    cur |= ((uint64_t)hi << 32) | lo;
                }
            }
            if ((kept >> lane) & 1ULL) {
                const int rank = nk0 + __popcll(kept & ((1ULL << lane) - 1ULL));
                const int i = 64 * b + lane;
                keep[rank] = i;
                if (SELECT) {
                    const Box bx = sbox[i];
                    float *r = rois + 6 * rank;
                    r[0] = bx.x1; r[1] = bx.y1; r[2] = bx.z1; r[3] = bx.x2; r[4] = bx.y2; r[5] = bx.z2;
                    roi_scores[rank] = scores_sorted[i];
                    roi_levels[rank] = level_all[order[i]];
                }
            }
            if (lane == 0) { s_kept = kept; s_nk = nk; }
        }
        __syncthreads();
        const uint64_t kept = s_kept;
        if (s_nk >= limit) break;
        for (int c = b + 1 + tid; c < col_blocks; c += blockDim.x) {
            uint64_t acc = remv[c], k = kept;
            while (k) {
                const int i = __builtin_ctzll(k);
                k &= k - 1;
                acc |= smask[(size_t)(64 * b + i) * col_blocks + c];
            }
            remv[c] = acc;
        }
        __syncthreads();
    }
    __syncthreads();
    const int nk = s_nk;
    if (tid == 0) num_keep[0] = nk;
    if (SELECT) {
        for (int r = nk + tid; r < max_keep; r += blockDim.x) {
            for (int k = 0; k < 6; ++k) rois[6 * r + k] = 0.0f;
            roi_scores[r] = 0.0f;
            roi_levels[r] = 0.0f;
        }
    }
}

size_t small_lds_bytes(int n)
{
    const size_t cb = (n + 63) / 64, npad = cb * 64;
    return npad * sizeof(Box) + npad * cb * 8 + cb * 8;
}

template <bool INDIRECT, bool SELECT>
int launch_nms(const float *boxes, const int64_t *order, const float *level_all, const float *scores_sorted, int n,
               float thresh, int max_keep, int64_t *keep, int32_t *num_keep, float *rois, float *roi_scores,
               float *roi_levels, void *ws, size_t ws_bytes, hipStream_t st)
{
    if (n < 0 || !keep || !num_keep) return SIS3D_EINVAL;
    if (SELECT && max_keep <= 0) return SIS3D_EINVAL;
    if (n == 0) {
        // nothing to keep; still define the outputs
        hipLaunchKernelGGL((nms_small_kernel<INDIRECT, SELECT>), dim3(1), dim3(64), 64, st, boxes, order, 0, thresh, max_keep,
                           keep, num_keep, level_all, scores_sorted, rois, roi_scores, roi_levels);
        return sis3d_check_launch();
    }
    if (n <= SMALL_N) {
        const size_t lds = small_lds_bytes(n);
        auto kern = nms_small_kernel<INDIRECT, SELECT>;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, boxes, order, n, thresh, max_keep, keep, num_keep, level_all,
                           scores_sorted, rois, roi_scores, roi_levels);
        return sis3d_check_launch();
    }
    const int cb = (n + 63) / 64;
    if (ws_bytes < (size_t)n * cb * 8 || !ws) return SIS3D_EWORKSPACE;
    uint64_t *mask = (uint64_t *)ws;
    hipLaunchKernelGGL((nms_mask_kernel<INDIRECT>), dim3(cb, cb), dim3(64), 0, st, boxes, order, n, thresh, mask);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((nms_sweep_kernel<SELECT>), dim3(1), dim3(256), (size_t)cb * 8, st, mask, n, max_keep, keep, num_keep,
                       boxes, level_all, scores_sorted, order, rois, roi_scores, roi_levels);
    return sis3d_check_launch();
}

} // namespace

extern "C" size_t sis3d_nms_workspace_bytes(int n)
{
    if (n <= SMALL_N) return 0;
    return (size_t)n * ((n + 63) / 64) * 8;
}

extern "C" int sis3d_nms(const float *boxes, int n, float thresh, int max_keep, int64_t *keep, int32_t *num_keep, void *ws,
                         size_t ws_bytes, sis3d_stream_t stream)
{
    return launch_nms<false, false>(boxes, nullptr, nullptr, nullptr, n, thresh, max_keep, keep, num_keep, nullptr, nullptr,
                                    nullptr, ws, ws_bytes, as_stream(stream));
}

extern "C" int sis3d_nms_mask(const float *boxes, int n, float thresh, uint64_t *mask, sis3d_stream_t stream)
{
    if (n <= 0 || !mask) return n == 0 ? SIS3D_OK : SIS3D_EINVAL;
    const int cb = (n + 63) / 64;
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(cb, cb), dim3(64), 0, as_stream(stream), boxes, nullptr, n, thresh, mask);
    return sis3d_check_launch();
}

extern "C" int sis3d_nms_select(const float *boxes_all, const float *level_all, const float *scores_sorted, const int64_t *order,
                                int n, float thresh, int max_keep, float *rois, float *roi_scores, float *roi_levels,
                                int64_t *keep, int32_t *num_keep, void *ws, size_t ws_bytes, sis3d_stream_t stream)
{
    if (!boxes_all || !level_all || !scores_sorted || !order || !rois || !roi_scores || !roi_levels) return SIS3D_EINVAL;
    return launch_nms<true, true>(boxes_all, order, level_all, scores_sorted, n, thresh, max_keep, keep, num_keep, rois,
                                  roi_scores, roi_levels, ws, ws_bytes, as_stream(stream));
}
