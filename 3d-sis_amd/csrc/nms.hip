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
// loop; one tile per workgroup so the matrix spreads over as many CUs as it has
// tiles (a single workgroup doing all 28 tiles of n = 400 is VALU-bound: 28 us).
// The sweep is a single workgroup: per 64-box block the in-block decisions are
// resolved by one wave in SGPRs (s_ff1 over the not-yet-suppressed bits +
// v_readlane on the diagonal words, no memory traffic), then the kept rows are
// OR-folded into the later column blocks by the other waves with shuffles.  The
// matrix is staged in LDS when it fits.  Measured n = 400: 57 us -> see profiles/.
#include "common.h"
#include <stdlib.h>
#include <atomic>

namespace {

struct Box { float x1, y1, z1, x2, y2, z2, area, pad; };   // area = (x2-x1+1)*(y2-y1+1)*(z2-z1+1), computed once per box

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

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
                                              float thresh, int lane, int r0 = 0, int r1 = 64)
{
    uint64_t mine = 0;
    const int col_idx = 64 * cb + lane;
    const int nrows = min(r1, n - 64 * rb);
#pragma unroll 4
    for (int r = r0; r < nrows; ++r) {
        const Box a = rows[r];                       // LDS broadcast (same address in all lanes)
        const float v = iou3d(a, col);
        const bool sup = col_valid && (col_idx > 64 * rb + r) && !(v <= thresh);
        const uint64_t word = __ballot(sup);
        if (lane == r) mine = word;
    }
    return mine;
}

// ---------------------------------------------------------------- global-matrix path
// one 64x64 tile per workgroup of MASK_WAVES waves: wave w walks rows [R w, R (w+1)), R = 64 / MASK_WAVES (the 64
// dependent ballot steps of a tile were the latency of this kernel: 11.9 us at n = 400 with one wave per tile)
constexpr int MASK_WAVES = 4;

template <bool INDIRECT>
__global__ __launch_bounds__(64 * MASK_WAVES) void nms_mask_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order,
                                                                   int n, float thresh, uint64_t *__restrict__ mask)
{
    const int cb = blockIdx.x, rb = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col_blocks = (n + 63) / 64;
    constexpr int R = 64 / MASK_WAVES;
    __shared__ Box rows[64];
    const int ri = 64 * rb + lane;
    if (cb < rb) {                                   // strictly-lower tiles are never read by the sweep
        if (wave == 0 && ri < n) mask[(size_t)ri * col_blocks + cb] = 0;
        return;
    }
    if (wave == 0 && ri < n) rows[lane] = load_box<INDIRECT>(boxes, order, ri);
    __syncthreads();
    const int ci = 64 * cb + lane;
    Box col = {0, 0, 0, 0, 0, 0, 1, 0};
    if (ci < n) col = load_box<INDIRECT>(boxes, order, ci);
    const uint64_t w = tile_word(rows, col, ci < n, rb, cb, n, thresh, lane, R * wave, R * (wave + 1));
    if (lane >= R * wave && lane < R * (wave + 1) && ri < n) mask[(size_t)ri * col_blocks + cb] = w;
}

// Greedy sweep: ONE workgroup of 1024 threads over the bit matrix produced by nms_mask_kernel.
//  * per 64-box block, wave 0 resolves the in-block decisions with the running remove word in SGPRs; it visits only
//    the not-yet-suppressed boxes (s_ff1 on ~cur) and ORs in their diagonal words via v_readlane -- no memory traffic;
//  * the kept rows are then folded into the remove words of the later column blocks by the other waves in parallel:
//    wave w takes column block b+1+w, lane i contributes row i's word, 64-lane OR by shuffles;
//  * the matrix is staged into LDS when it fits (n <= ~1400), else read from L2; the candidates' level / score are staged
//    up front so the select variant never waits on a dependent global load inside the serial part.
constexpr size_t SWEEP_LDS_MASK_MAX = 120 * 1024;

template <bool SELECT>
__global__ __launch_bounds__(1024) void nms_sweep_kernel(const uint64_t *__restrict__ mask, int n, int max_keep,
                                                         int64_t *__restrict__ keep, int32_t *__restrict__ num_keep,
                                                         const float *__restrict__ boxes_all, const float *__restrict__ level_all,
                                                         const float *__restrict__ scores_sorted, const int64_t *__restrict__ order,
                                                         float *__restrict__ rois, float *__restrict__ roi_scores,
                                                         float *__restrict__ roi_levels, int stage_mask, int stage_meta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int col_blocks = (n + 63) / 64;
    uint64_t *remv = (uint64_t *)smem;                                   // [col_blocks]
    uint64_t *smask = remv + col_blocks;                                 // [n][col_blocks] if staged
    float *slev = (float *)(smask + (stage_mask ? (size_t)n * col_blocks : 0));   // [n] if staged
    float *sscr = slev + (stage_meta ? n : 0);
    int *skeep = (int *)(sscr + (stage_meta ? n : 0));                   // [min(n, max_keep)] kept candidates (SELECT)
    __shared__ uint64_t s_kept;
    __shared__ int s_nk;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    for (int c = tid; c < col_blocks; c += blockDim.x) remv[c] = 0;
    if (stage_mask)
        for (size_t i = tid; i < (size_t)n * col_blocks; i += blockDim.x) smask[i] = mask[i];
    if (SELECT && stage_meta)
        for (int i = tid; i < n; i += blockDim.x) { slev[i] = level_all[order[i]]; sscr[i] = scores_sorted[i]; }
    if (tid == 0) s_nk = 0;
    __syncthreads();
    const uint64_t *M = stage_mask ? smask : mask;
    const int limit = max_keep > 0 ? max_keep : n;
    for (int b = 0; b < col_blocks; ++b) {
        if (wid == 0) {
            const int ri = 64 * b + lane;
            const uint64_t diag = ri < n ? M[(size_t)ri * col_blocks + b] : 0;
            uint64_t cur = uniform64(remv[b]);
            const int nrows = min(64, n - 64 * b);
            const uint64_t valid = nrows == 64 ? ~0ULL : ((1ULL << nrows) - 1ULL);
            const int nk0 = s_nk;
            int nk = nk0;
            uint64_t kept = 0;
            uint64_t avail = ~cur & valid;                               // candidates not suppressed so far
            while (avail && nk < limit) {
                const int i = __builtin_ctzll(avail);
                kept |= 1ULL << i;
                ++nk;
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)diag, i);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(diag >> 32), i);
                cur |= ((uint64_t)hi << 32) | lo;
                const uint64_t above = i == 63 ? 0ULL : (~0ULL << (i + 1));
                avail = ~cur & valid & above;
            }
            if ((kept >> lane) & 1ULL) {
                const int rank = nk0 + __popcll(kept & ((1ULL << lane) - 1ULL));
                const int i = 64 * b + lane;
                keep[rank] = i;
                if (SELECT) skeep[rank] = i;             // the gather itself runs after the sweep, on all threads
            }
            if (lane == 0) { s_kept = kept; s_nk = nk; }
        }
        __syncthreads();
        const uint64_t kept = s_kept;
        if (s_nk >= limit) break;
        for (int c = b + 1 + wid; c < col_blocks; c += nw) {
            const int ri = 64 * b + lane;
            uint64_t v = (((kept >> lane) & 1ULL) && ri < n) ? M[(size_t)ri * col_blocks + c] : 0ULL;
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
                v |= ((uint64_t)hi << 32) | lo;
            }
            if (lane == 0) remv[c] |= v;
        }
        __syncthreads();
    }
    __syncthreads();
    const int nk = s_nk;
    if (tid == 0) num_keep[0] = nk;
    if (SELECT) {
        // surviving rois / scores / levels: one thread per survivor, its dependent chain is order[i] -> three 8-byte box
        // loads in flight together (this gather used to sit inside the serial sweep, seven awaited loads per kept row)
        for (int r = tid; r < nk; r += blockDim.x) {
            const int i = skeep[r];
            const int64_t src = order[i];
            const float2 *p = reinterpret_cast<const float2 *>(boxes_all + 6 * src);
            const float2 b0 = p[0], b1 = p[1], b2 = p[2];
            const float sc = stage_meta ? sscr[i] : scores_sorted[i];
            const float lv = stage_meta ? slev[i] : level_all[src];
            float2 *q = reinterpret_cast<float2 *>(rois + 6 * r);
            q[0] = b0; q[1] = b1; q[2] = b2;
            roi_scores[r] = sc;
            roi_levels[r] = lv;
        }
        // zero-fill the padded rows
        for (int r = nk + tid; r < max_keep; r += blockDim.x) {
            for (int k = 0; k < 6; ++k) rois[6 * r + k] = 0.0f;
            roi_scores[r] = 0.0f;
            roi_levels[r] = 0.0f;
        }
    }
}


// ---------------------------------------------------------------- large n: candidate lists + parallel resolve
// The single-workgroup sweep above reads every kept row's words one 64-box block after the other: at n = 6400 (a whole
// scene's records, BASELINE config 5) that is 100 dependent rounds over a 5 MB matrix -- ~0.4 ms for a decision that is
// almost empty (per-chunk NMS already ran; only boxes at chunk borders still meet).  Greedy NMS is a fixed point of
//     keep[i] = no j < i with IoU(j, i) > thresh is kept,
// so it is resolved here in parallel sweeps over a SPARSE suppressor table: box i is decided "suppressed" as soon as one of
// its suppressor candidates is known to be kept, "kept" as soon as all of them are known to be suppressed; boxes without
// candidates are kept in the first sweep.  The number of sweeps is the longest suppression chain (a handful for a scene),
// and the result is the sequential algorithm's keep list bit for bit (same IoU arithmetic, same strict comparison).
//   W[i][jb]  word of suppressor candidates of box i inside block jb (bit j: box 64 jb + j, j < i, IoU > thresh); written
//             only where non-zero
//   nz[i][.]  bitmap of the non-zero words of row i (zeroed by the launcher)
//   n_dev     when given, the box count is read on the device (<= n_cap, which fixes the table strides): the scene merge
//             sorts and suppresses in one stream-ordered sequence without a host readback in between
template <bool INDIRECT>
__global__ __launch_bounds__(64 * MASK_WAVES) void nms_cand_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ order,
                                                                   int n_cap, const int32_t *__restrict__ n_dev, int bstride,
                                                                   float thresh, uint64_t *__restrict__ W,
                                                                   unsigned long long *__restrict__ nz, int nzw)
{
    const int jb = blockIdx.x, ib = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (jb > ib) return;
    const int n = n_dev ? min(n_dev[0], n_cap) : n_cap;
    if (64 * ib >= n) return;
    const int col_blocks = (n_cap + 63) / 64;
    constexpr int R = 64 / MASK_WAVES;
    __shared__ Box rows[64];
    const int ii = 64 * ib + lane;
    auto box_at = [&](int i) {
        if (bstride == 6) return load_box<INDIRECT>(boxes, order, i);
        const float *p = boxes + (size_t)bstride * i;
        Box b;
        b.x1 = p[0]; b.y1 = p[1]; b.z1 = p[2]; b.x2 = p[3]; b.y2 = p[4]; b.z2 = p[5];
        b.area = (b.x2 - b.x1 + 1.0f) * (b.y2 - b.y1 + 1.0f) * (b.z2 - b.z1 + 1.0f);
        b.pad = 0.0f;
        return b;
    };
    if (wave == 0 && ii < n) rows[lane] = box_at(ii);
    __syncthreads();
    const int ji = 64 * jb + lane;
    Box col = {0, 0, 0, 0, 0, 0, 1, 0};
    if (ji < n) col = box_at(ji);
    uint64_t mine = 0;
    const int nrows = min(R * (wave + 1), n - 64 * ib);
#pragma unroll 4
    for (int r = R * wave; r < nrows; ++r) {
        const Box a = rows[r];
        const float v = iou3d(col, a);                 // suppressor first, as the sweep path (the value is symmetric anyway)
        const bool sup = ji < n && ji < 64 * ib + r && !(v <= thresh);
        const uint64_t word = __ballot(sup);
        if (lane == r) mine = word;
    }
    if (mine != 0 && ii < n) {
        W[(size_t)ii * col_blocks + jb] = mine;
        atomicOr(&nz[(size_t)ii * nzw + (jb >> 6)], 1ULL << (jb & 63));
    }
}

__global__ __launch_bounds__(1024) void nms_resolve_kernel(const uint64_t *__restrict__ W, const unsigned long long *__restrict__ nz,
                                                           int nzw, int n_cap, const int32_t *__restrict__ n_dev, int max_keep,
                                                           int64_t *__restrict__ keep, int32_t *__restrict__ num_keep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = n_dev ? min(n_dev[0], n_cap) : n_cap;
    const int cb = (n_cap + 63) / 64;                  // table stride; words past ceil(n/64) stay zero
    unsigned long long *Kc = (unsigned long long *)smem, *Dc = Kc + cb, *Kn = Dc + cb, *Dn = Kn + cb;   // snapshot / next
    int *pref = (int *)(Dn + cb);
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    for (int c = tid; c < cb; c += blockDim.x) { Kc[c] = Dc[c] = Kn[c] = Dn[c] = 0; }
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int sweep = 0; sweep < n; ++sweep) {
        bool changed = false;
        for (int i = tid; i < n; i += blockDim.x) {
            const unsigned long long bit = 1ULL << (i & 63);
            if (Dc[i >> 6] & bit) continue;
            bool any_kept = false, all_dec = true;
            for (int q = 0; q < nzw; ++q) {
                unsigned long long bits = nz[(size_t)i * nzw + q];
                while (bits) {
                    const int jb = 64 * q + __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const uint64_t w = W[(size_t)i * cb + jb];
                    any_kept |= (w & Kc[jb]) != 0;
                    all_dec &= (w & ~Dc[jb]) == 0;
                }
            }
            if (any_kept) {
                atomicOr(&Dn[i >> 6], bit);
                changed = true;
            } else if (all_dec) {
                atomicOr(&Kn[i >> 6], bit);
                atomicOr(&Dn[i >> 6], bit);
                changed = true;
            }
        }
        if (changed) s_changed = 1;
        __syncthreads();
        const int any = s_changed;
        for (int c = tid; c < cb; c += blockDim.x) { Kc[c] = Kn[c]; Dc[c] = Dn[c]; }
        __syncthreads();
        if (!any) break;                               // every box is decided (the first undecided box always resolves)
        if (tid == 0) s_changed = 0;
        __syncthreads();
    }
    // keep list = set bits of K in ascending order, cut at max_keep
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < cb; ++c) { pref[c] = run; run += __popcll(Kc[c]); }
        const int limit = max_keep > 0 ? max_keep : n;
        num_keep[0] = run < limit ? run : limit;
    }
    __syncthreads();
    const int limit = max_keep > 0 ? max_keep : n;
    for (int c = tid; c < cb; c += blockDim.x) {
        unsigned long long bits = Kc[c];
        int rank = pref[c];
        while (bits && rank < limit) {
            keep[rank++] = 64 * c + __builtin_ctzll(bits);
            bits &= bits - 1;
        }
    }
}


// ---------------------------------------------------------------- whole-scene merge (BASELINE config 5)
// One stream-ordered sequence for what parallel.merge_scene does with ~10 torch kernels and two host readbacks: the
// gathered per-chunk record blocks [n_chunks][1 + K * width] (slot 0 = valid row count) are flattened, the valid rows
// ordered by score with the stable descending rule (ties: chunk id, then row), the rows gathered in that order, and the
// whole-scene NMS run on them with the row count staying on the device.
__device__ __forceinline__ uint32_t merge_order_key(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;       // NaN first, as torch's descending sort
    if (u == 0x80000000u) u = 0;                                      // -0.0 ties with +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Order = RANK: the composite keys (score image << 32 | ~flat index) are unique, so the sorted position of a valid row is the
// number of keys above it.  scene_keys_kernel compacts the valid rows' keys (row r of chunk c sits at prefix[c] + r: no scan
// over rows is needed); scene_rank_kernel: one workgroup per 64 keys, all keys staged in its LDS, sixteen waves each scan a
// sixteenth of them with broadcast reads, then the row is gathered to its rank.  (A one-workgroup bitonic network over 8192
// slots took 115 us; torch.sort 31 us + gather 8 us.)
__device__ __forceinline__ int block_count(const float *blocks, int c, int bf, int k_rows)
{
    const int cnt = (int)rintf(blocks[(size_t)c * bf]);
    return cnt < 0 ? 0 : (cnt > k_rows ? k_rows : cnt);
}

__global__ __launch_bounds__(256) void scene_keys_kernel(const float *__restrict__ blocks, int n_chunks, int k_rows, int width,
                                                         int score_col, uint64_t *__restrict__ keys, int32_t *__restrict__ total_out)
{
    extern __shared__ int s_pref[];                                   // [n_chunks + 1]
    const int tid = threadIdx.x, bf = 1 + k_rows * width, T = n_chunks * k_rows;
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < n_chunks; ++c) { s_pref[c] = run; run += block_count(blocks, c, bf, k_rows); }
        s_pref[n_chunks] = run;
        if (blockIdx.x == 0) total_out[0] = run;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + tid;
    if (i >= T) return;
    const int c = i / k_rows, r = i - c * k_rows;
    if (r < s_pref[c + 1] - s_pref[c])
        keys[s_pref[c] + r] = ((uint64_t)merge_order_key(blocks[(size_t)c * bf + 1 + (size_t)r * width + score_col]) << 32) |
                              (0xffffffffu - (uint32_t)i);
}

__global__ __launch_bounds__(1024) void scene_rank_kernel(const float *__restrict__ blocks, const uint64_t *__restrict__ keysg,
                                                          const int32_t *__restrict__ total_dev, int k_rows, int width,
                                                          float *__restrict__ recs, int32_t *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = (uint64_t *)smem;                                // [total rounded up to 256], tail = 0
    __shared__ int s_rank[64];
    const int total = total_dev[0];
    if ((int)(blockIdx.x * 64) >= total) return;
    const int tid = threadIdx.x, bf = 1 + k_rows * width, Tp = (total + 255) & ~255;
    for (int j = 2 * tid; j < Tp; j += 2 * blockDim.x) {              // 16 B per lane, tail zeroed
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j + 1 < total) v = *reinterpret_cast<const uint4 *>(keysg + j);
        else if (j < total) { const uint64_t k = keysg[j]; v.x = (uint32_t)k; v.y = (uint32_t)(k >> 32); }
        *reinterpret_cast<uint4 *>(keys + j) = v;
    }
    if (tid < 64) s_rank[tid] = 0;
    __syncthreads();
    const int e = tid & 63, part = tid >> 6;
    const int i = blockIdx.x * 64 + e;
    const uint64_t my = i < total ? keys[i] : 0ULL;
    const int q = Tp / 16;                                            // a multiple of 16 keys
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + part * q);
    int above = 0;
    for (int j = 0; j < q / 2; j += 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k4[j + u];                 // same address in every lane: LDS broadcast
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            above += (((uint64_t)v[u].y << 32) | v[u].x) > my ? 1 : 0;
            above += (((uint64_t)v[u].w << 32) | v[u].z) > my ? 1 : 0;
        }
    }
    if (my != 0 && above) atomicAdd(&s_rank[e], above);
    __syncthreads();
    if (my != 0 && part < 4) {
        const int rk = s_rank[e];
        const int flat = (int)(0xffffffffu - (uint32_t)my);
        const int c = flat / k_rows, r = flat - c * k_rows;
        const float *src = blocks + (size_t)c * bf + 1 + (size_t)r * width;
        float *dst = recs + (size_t)rk * width;
        for (int f = part; f < width; f += 4) dst[f] = src[f];
        if (part == 0) order[rk] = flat;
    }
}

template <bool INDIRECT, bool SELECT>
int launch_nms(const float *boxes, const int64_t *order, const float *level_all, const float *scores_sorted, int n,
               float thresh, int max_keep, int64_t *keep, int32_t *num_keep, float *rois, float *roi_scores,
               float *roi_levels, void *ws, size_t ws_bytes, hipStream_t st, int path = 0)
{
    if (n < 0 || !keep || !num_keep || path < 0 || path > 2) return SIS3D_EINVAL;
    if (SELECT && max_keep <= 0) return SIS3D_EINVAL;
    const int cb = (n + 63) / 64;
    const size_t mask_bytes = (size_t)n * cb * 8;
    const int nzw = (cb + 63) / 64;
    const size_t nz_bytes = (size_t)n * nzw * 8;
    if (n > 0 && (ws_bytes < mask_bytes || !ws)) return SIS3D_EWORKSPACE;          // n * ceil(n/64) * 8: what the sweep paths need
    uint64_t *mask = (uint64_t *)ws;
    if (!SELECT && n > 0 && (path == 2 || (path == 0 && mask_bytes > SWEEP_LDS_MASK_MAX))) {
        // only this path reads the non-zero-word bitmap behind the matrix: sis3d_nms_workspace_bytes(n) covers both
        if (ws_bytes < mask_bytes + nz_bytes) return SIS3D_EWORKSPACE;
        // the matrix does not fit the sweep workgroup's LDS: sparse candidate table + parallel resolve
        const size_t lds = (size_t)cb * (4 * 8 + 4) + 16;
        if (lds > 160 * 1024 - 64) return SIS3D_EUNSUPPORTED;
        unsigned long long *nz = (unsigned long long *)((char *)ws + mask_bytes);
        if (sis3d_fill32(nz, 0u, nz_bytes, st) != SIS3D_OK) return SIS3D_ELAUNCH;
        hipLaunchKernelGGL((nms_cand_kernel<INDIRECT>), dim3(cb, cb), dim3(64 * MASK_WAVES), 0, st, boxes, order, n, nullptr, 6, thresh, mask, nz, nzw);
        int rc = sis3d_check_launch();
        if (rc) return rc;
        // granted once, at the largest size any launch may ask for (never lowered by a later, smaller launch)
        static Sis3dLdsOnce lds_once;
        if (sis3d_grant_lds(lds_once, (const void *)nms_resolve_kernel, 0) != SIS3D_OK) return SIS3D_ELAUNCH;
        hipLaunchKernelGGL(nms_resolve_kernel, dim3(1), dim3(1024), lds, st, mask, nz, nzw, n, nullptr, max_keep, keep, num_keep);
        return sis3d_check_launch();
    }
    if (n > 0) {
        // the bit matrix on as many CUs as it has 64x64 tiles (a single workgroup is VALU-bound: 28 us for n = 400)
        hipLaunchKernelGGL((nms_mask_kernel<INDIRECT>), dim3(cb, cb), dim3(64 * MASK_WAVES), 0, st, boxes, order, n, thresh, mask);
        int rc = sis3d_check_launch();
        if (rc) return rc;
    }
    constexpr size_t LDS_LIMIT = 160 * 1024 - 64;
    int stage_mask = mask_bytes <= SWEEP_LDS_MASK_MAX ? 1 : 0;
    int stage_meta = (SELECT && n <= 4096) ? 1 : 0;
    const size_t keep_b = SELECT ? (size_t)(n < max_keep ? n : max_keep) * 4 : 0;     // the survivors' positions (select path)
    auto need = [&] { return (size_t)cb * 8 + (stage_mask ? mask_bytes : 0) + (stage_meta ? (size_t)n * 8 : 0) + keep_b + 16; };
    // the optional stagings are dropped first; what remains (one remove word per column block + the keep positions) is
    // bounded by the caller's n / max_keep, not by this library: refuse instead of failing the launch
    if (need() > LDS_LIMIT) stage_meta = 0;
    if (need() > LDS_LIMIT) stage_mask = 0;
    if (need() > LDS_LIMIT) return SIS3D_EUNSUPPORTED;
    const size_t lds = need();
    auto kern = nms_sweep_kernel<SELECT>;
    // once per instantiation and device: not per launch, so that no attribute write can coincide with the enqueue of a captured
    // graph that contains this kernel
    static Sis3dLdsOnce lds_once;
    if (sis3d_grant_lds(lds_once, (const void *)kern, 0) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, mask, n, max_keep, keep, num_keep, boxes, level_all, scores_sorted, order,
                       rois, roi_scores, roi_levels, stage_mask, stage_meta);
    return sis3d_check_launch();
}

} // namespace

extern "C" size_t sis3d_nms_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    const size_t cb = (size_t)(n + 63) / 64;
    return (size_t)n * cb * 8 + (size_t)n * ((cb + 63) / 64) * 8;      // bit matrix + non-zero-word bitmap (resolve path)
}

extern "C" int sis3d_nms(const float *boxes, int n, float thresh, int max_keep, int64_t *keep, int32_t *num_keep, void *ws,
                         size_t ws_bytes, int path, sis3d_stream_t stream)
{
    return launch_nms<false, false>(boxes, nullptr, nullptr, nullptr, n, thresh, max_keep, keep, num_keep, nullptr, nullptr,
                                    nullptr, ws, ws_bytes, as_stream(stream), path);
}

extern "C" int sis3d_nms_mask(const float *boxes, int n, float thresh, uint64_t *mask, sis3d_stream_t stream)
{
    if (n <= 0 || !mask) return n == 0 ? SIS3D_OK : SIS3D_EINVAL;
    const int cb = (n + 63) / 64;
    hipLaunchKernelGGL((nms_mask_kernel<false>), dim3(cb, cb), dim3(64 * MASK_WAVES), 0, as_stream(stream), boxes, nullptr, n, thresh, mask);
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

extern "C" size_t sis3d_scene_merge_workspace_bytes(int n_chunks, int k_rows)
{
    const int64_t T = (int64_t)n_chunks * k_rows;
    return (n_chunks > 0 && k_rows > 0 && T <= 0x7fffffff) ? sis3d_nms_workspace_bytes((int)T) : 0;
}

extern "C" int sis3d_scene_merge(const float *blocks, int n_chunks, int k_rows, int width, int score_col, int box_col, float thresh,
                                 int max_keep, float *recs, int32_t *order, int64_t *keep, int32_t *counts, void *ws, size_t ws_bytes,
                                 sis3d_stream_t stream)
{
    if (!blocks || !recs || !order || !keep || !counts || n_chunks <= 0 || k_rows <= 0 || width < 6) return SIS3D_EINVAL;
    if (score_col < 0 || score_col >= width || box_col < 0 || box_col + 6 > width || max_keep < 0) return SIS3D_EINVAL;
    const int64_t T64 = (int64_t)n_chunks * k_rows;
    if (T64 > 8192) return SIS3D_EUNSUPPORTED;                     // the sort holds every candidate key in one workgroup's LDS
    const int T = (int)T64;
    const int cb = (T + 63) / 64, nzw = (cb + 63) / 64;
    const size_t mask_bytes = (size_t)T * cb * 8, nz_bytes = (size_t)T * nzw * 8;
    if (!ws || ws_bytes < mask_bytes + nz_bytes) return SIS3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    // the key list borrows the head of the bit-matrix area (>= T words): it is dead before the suppressor table is written
    uint64_t *mask = (uint64_t *)ws;
    hipLaunchKernelGGL(scene_keys_kernel, dim3((T + 255) / 256), dim3(256), (size_t)(n_chunks + 1) * sizeof(int), st, blocks, n_chunks,
                       k_rows, width, score_col, mask, counts);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    const size_t sort_lds = (size_t)((T + 255) & ~255) * 8;
    static Sis3dLdsOnce rank_lds_once;                               // once per device
    if (sis3d_grant_lds(rank_lds_once, (const void *)scene_rank_kernel, 0) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(scene_rank_kernel, dim3((T + 63) / 64), dim3(1024), sort_lds, st, blocks, mask, counts, k_rows, width, recs, order);
    rc = sis3d_check_launch();
    if (rc) return rc;
    unsigned long long *nz = (unsigned long long *)((char *)ws + mask_bytes);
    if (sis3d_fill32(nz, 0u, nz_bytes, st) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL((nms_cand_kernel<false>), dim3(cb, cb), dim3(64 * MASK_WAVES), 0, st, recs + box_col, nullptr, T, counts, width,
                       thresh, mask, nz, nzw);
    rc = sis3d_check_launch();
    if (rc) return rc;
    const size_t lds = (size_t)cb * (4 * 8 + 4) + 16;
    hipLaunchKernelGGL(nms_resolve_kernel, dim3(1), dim3(1024), lds, st, mask, nz, nzw, T, counts, max_keep, keep, counts + 1);
    return sis3d_check_launch();
}
