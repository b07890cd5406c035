// The colour stem on a back-projected image volume, SPARSE: color[0] = Conv3d(128, 64, k=2, s=2, bias=False) + ReLU of
// lib/nets/backbones.py:187,214 reading the view max of lib/nets/network.py:216-239 through the voxel->pixel table of
// sis3d_project_views_prepare, followed by the next Bottleneck's conv1 (1x1x1, 64 -> 32, bias, ReLU).
// A view sees at most a few thousand of the 442,368 voxels of a chunk (n_v ~ 3,000 per view; the depth image has 1,312
// pixels), so > 95 % of the input volume is exactly zero and, the conv having no bias, so is every output voxel none of whose eight
// input voxels is visible.  The dense kernel (sis3d_conv3d_chain_projected) multiplies those zeros: 7.25 GFLOP, 150 us.  Here:
//   1. proj_mask_kernel     one thread per OUTPUT voxel: which of its 8 taps has a visible input voxel (8-bit mask), workgroup counts;
//   2. proj_compact_kernel  deterministic compaction (ascending voxel order, prefix sums, no atomics) -> list of active output voxels;
//   3. proj_fill_kernel     every output row := the constant of an inactive voxel (main: relu(bias) = 0; stage: relu(W1 relu(bias) + b1));
//   4. proj_tile_kernel     64 active voxels per workgroup (4 waves x 16): per tap, weights staged through LDS once per workgroup, the
//                           wave gathers its visible voxels' feature rows (max over the views, an invisible view counts as 0) and runs
//                           the transposed tile GEMM of mfma16.h; a wave skips a tap none of its 16 voxels sees; ReLU, store, then the
//                           stage conv on the tile in registers.
// The grid of step 4 is sized for the worst case (every output voxel active) and workgroups past the list's end exit at once, so the
// launch sequence is the same for any visibility and captures into a HIP graph; with fully dense visibility it does the dense
// kernel's arithmetic at a lower efficiency -- real chunks and the synthetic workload are 1-4 % dense.
// Summation order differs from the dense kernel's (tap-major here), results agree to ~1e-6 of the output scale.
#include "common.h"
#include "mfma16.h"

namespace {

constexpr int CIN = 128, COUT = 64, KG = CIN / 16, CT = COUT / 16;
constexpr int MAXSL = 6;                                      // view slots the tile kernel holds table entries for
constexpr int TAP_FLOATS = CT * KG * 256;                     // one tap's weights in pw16 order: [CT][KG][64][4]

struct ProjArgs {
    const int32_t *tab;        // [nslots][nvox], voxel index (z * Y + y) * X + x, -1 = invisible
    const float *rows;         // [nslots][npix][CIN]
    int nslots;
    int64_t npix, nvox;
    int X, Y, Z;               // INPUT grid; output grid = X/2 x Y/2 x Z/2
    const float *w;            // pw16 pack of the (COUT, 8 * CIN) matrix, column = tap * CIN + ci, tap = (dx * 2 + dy) * 2 + dz
    const float *bias;         // may be null
    int relu;
    float *out;                // rows of COUT floats, voxel (ox * OY + oy) * OZ + oz
    const float *w1, *b1;      // stage: pw16 pack (C2, COUT), bias (may be null)
    float *y1;                 // rows of C2 floats
    uint8_t *mask;             // [nout] in table order (oz * OY + oy) * OX + ox
    int *counts;               // [nblocks]
    int *total;                // [1]
    int *list;                 // [nout]: out-row index | mask << 24
    int nout, nblocks;
};

__global__ __launch_bounds__(256) void proj_mask_kernel(const ProjArgs a)
{
    const int OX = a.X / 2, OY = a.Y / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    int m = 0;
    if (t < a.nout) {
        const int ox = t % OX, oy = (t / OX) % OY, oz = t / (OX * OY);
        // the two dx taps of a (dy, dz) pair are neighbours in the table (x fastest, 2 ox even, nvox even): one 8-byte read for both
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const int dy = yz >> 1, dz = yz & 1;
            const int64_t vox = ((int64_t)(2 * oz + dz) * a.Y + (2 * oy + dy)) * a.X + 2 * ox;
            int seen0 = 0, seen1 = 0;
            for (int sl = 0; sl < a.nslots; ++sl) {
                const int2 p = *reinterpret_cast<const int2 *>(a.tab + sl * a.nvox + vox);
                seen0 |= p.x >= 0;
                seen1 |= p.y >= 0;
            }
            m |= (seen0 << (dy * 2 + dz)) | (seen1 << (4 + dy * 2 + dz));      // tap = (dx * 2 + dy) * 2 + dz
        }
        a.mask[t] = (uint8_t)m;
    }
    const int n = __syncthreads_count(m != 0);
    if (threadIdx.x == 0) a.counts[blockIdx.x] = n;
}

__global__ __launch_bounds__(256) void proj_compact_kernel(const ProjArgs a)
{
    __shared__ int part[256];
    __shared__ int wsum[4];
    // base = number of active voxels in the workgroups before this one
    int s = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) s += a.counts[j];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    const int base = part[0];
    const int OX = a.X / 2, OY = a.Y / 2, OZ = a.Z / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t < a.nout ? a.mask[t] : 0;
    const unsigned long long b = __ballot(m != 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (m) {
        const int ox = t % OX, oy = (t / OX) % OY, oz = t / (OX * OY);
        const int row = (ox * OY + oy) * OZ + oz;
        a.list[off + __popcll(b & ((1ull << lane) - 1ull))] = row | (m << 24);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.total[0] = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// element (c2, k) of a pw16 pack with KG2 k-groups
__device__ __forceinline__ float pw16_at(const float *w, int kg2, int c2, int k)
{
    return w[(((size_t)(c2 >> 4) * kg2 + (k >> 4)) * 64 + (((k >> 2) & 3) * 16 + (c2 & 15))) * 4 + (k & 3)];
}

template <int C2>
__global__ __launch_bounds__(256) void proj_fill_kernel(const ProjArgs a)
{
    __shared__ float cm[COUT], cs[C2 > 0 ? C2 : 1];
    if (threadIdx.x < COUT) {
        float v = a.bias ? a.bias[threadIdx.x] : 0.f;
        cm[threadIdx.x] = a.relu ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
    if constexpr (C2 > 0) {
        if ((int)threadIdx.x < C2) {
            float acc = 0.f;
            for (int k = 0; k < COUT; ++k) acc = fmaf(pw16_at(a.w1, COUT / 16, threadIdx.x, k), cm[k], acc);
            cs[threadIdx.x] = fmaxf(acc + (a.b1 ? a.b1[threadIdx.x] : 0.f), 0.f);
        }
        __syncthreads();
    }
    const int64_t n_main = (int64_t)a.nout * (COUT / 4), n_st = (int64_t)a.nout * (C2 / 4);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n_main + n_st; i += (int64_t)gridDim.x * 256) {
        if (i < n_main) {
            const int c4 = (int)(i % (COUT / 4));
            reinterpret_cast<float4 *>(a.out)[i] = make_float4(cm[4 * c4], cm[4 * c4 + 1], cm[4 * c4 + 2], cm[4 * c4 + 3]);
        } else if constexpr (C2 > 0) {
            const int64_t j = i - n_main;
            const int c4 = (int)(j % (C2 / 4));
            reinterpret_cast<float4 *>(a.y1)[j] = make_float4(cs[4 * c4], cs[4 * c4 + 1], cs[4 * c4 + 2], cs[4 * c4 + 3]);
        }
    }
}

template <int R>
__device__ __forceinline__ float comp4(const float4 &v) { return R == 0 ? v.x : R == 1 ? v.y : R == 2 ? v.z : v.w; }

template <int C2>
__global__ __launch_bounds__(256, 1) void proj_tile_kernel(const ProjArgs a)
{
    __shared__ __attribute__((aligned(16))) float wl[2][TAP_FLOATS];              // two taps of weights, 32 KB each
    const int total = a.total[0];
    const int first = blockIdx.x * 64;
    if (first >= total) return;                                                     // uniform: the whole workgroup leaves
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e_idx = first + wave * 16 + li;
    const bool live = e_idx < total;
    const int e = a.list[live ? e_idx : first];
    const int row = e & 0xffffff, m = live ? (e >> 24) & 0xff : 0;
    const int OY = a.Y / 2, OZ = a.Z / 2;
    const int oz = row % OZ, oy = (row / OZ) % OY, ox = row / (OZ * OY);

    // one tap's weights = 32 chunks of 1 KB ([ct][g][64 lanes][16 B]); a wave moves 8 of them global -> LDS by LDS-DMA (no registers:
    // staged through a register array the copy went through scratch memory)
    auto stage_tap = [&](int tap, int buf) __attribute__((always_inline)) {
        static_for<0, 8>([&](auto J) {
            const int c = wave * 8 + decltype(J)::value, ct = c / KG, g = c % KG;
            const float *src = a.w + (((size_t)ct * (8 * KG) + (size_t)tap * KG + g) * 64 + lane) * 4;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                             (void __attribute__((address_space(3))) *)(wl[buf] + c * 256), 16, 0, 0);
        });
    };
    stage_tap(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[CT][2];
    static_for<0, CT>([&](auto N) { acc[decltype(N)::value][0] = acc[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // every table entry of the voxel's eight input voxels, requested at once (a lane whose tap bit is clear reads them too: no branch,
    // no dependent chain): the per-tap gather below then costs ONE memory round trip, and it runs one tap ahead of the MFMAs
    int pix[8][MAXSL];
    static_for<0, 8>([&](auto T) {
        constexpr int tap = decltype(T)::value, dx = tap >> 2, dy = (tap >> 1) & 1, dz = tap & 1;
        const int64_t vox = ((int64_t)(2 * oz + dz) * a.Y + (2 * oy + dy)) * a.X + (2 * ox + dx);
        static_for<0, MAXSL>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            pix[tap][sl] = sl < a.nslots ? a.tab[(sl < a.nslots ? sl : 0) * a.nvox + vox] : -1;
        });
    });
    // network.py:216-239 per voxel: max over the included views, an invisible view counts as 0
    auto gather = [&](auto T, float4 (&xv)[KG]) __attribute__((always_inline)) {
        constexpr int tap = decltype(T)::value;
        const bool mine = (m >> tap) & 1;
        static_for<0, KG>([&](auto G) { xv[decltype(G)::value] = zero4; });
        int cnt = 0;
        static_for<0, MAXSL>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            const int px = pix[tap][sl];
            if (mine && px >= 0) {
                const float *src = a.rows + ((size_t)sl * a.npix + px) * CIN + 4 * kq;
                static_for<0, KG>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    const float4 f = *reinterpret_cast<const float4 *>(src + 16 * g);
                    xv[g] = cnt == 0 ? f : make_float4(fmaxf(xv[g].x, f.x), fmaxf(xv[g].y, f.y), fmaxf(xv[g].z, f.z), fmaxf(xv[g].w, f.w));
                });
                ++cnt;
            }
        });
        if (cnt > 0 && cnt < a.nslots) {
            static_for<0, KG>([&](auto G) {
                constexpr int g = decltype(G)::value;
                xv[g] = make_float4(fmaxf(xv[g].x, 0.f), fmaxf(xv[g].y, 0.f), fmaxf(xv[g].z, 0.f), fmaxf(xv[g].w, 0.f));
            });
        }
    };
    float4 xa[KG], xb[KG];
    gather(std::integral_constant<int, 0>{}, xa);

    auto do_tap = [&](auto T, float4 (&xv)[KG], float4 (&xn)[KG]) __attribute__((always_inline)) {
        constexpr int tap = decltype(T)::value;
        if constexpr (tap + 1 < 8) {
            stage_tap(tap + 1, (tap + 1) & 1);                                      // that buffer was last read during tap - 1 (barrier since)
            gather(std::integral_constant<int, tap + 1>{}, xn);                     // in flight under this tap's MFMAs
        }
        if (__ballot((m >> tap) & 1) != 0ull) {                                     // wave-uniform: none of the 16 voxels sees this tap
            const float4 *wt = reinterpret_cast<const float4 *>(wl[tap & 1]) + lane;
            static_for<0, KG>([&](auto G) {
                constexpr int g = decltype(G)::value;
                float4 wv[CT];
                static_for<0, CT>([&](auto N) { wv[decltype(N)::value] = wt[(decltype(N)::value * KG + g) * 64]; });
                static_for<0, 4>([&](auto R) {
                    constexpr int r = decltype(R)::value;
                    static_for<0, CT>([&](auto N) {
                        constexpr int n = decltype(N)::value;
                        acc[n][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4<r>(wv[n]), comp4<r>(xv[g]), acc[n][r & 1], 0, 0, 0);
                    });
                });
            });
        }
        if constexpr (tap + 1 < 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's share of the next tap's weights has landed
            __syncthreads();
        }
    };
    static_for<0, 8>([&](auto T) {
        if constexpr (decltype(T)::value & 1) do_tap(T, xb, xa);
        else do_tap(T, xa, xb);
    });

    // ---- bias, ReLU, store; then the stage conv on the tile
    float4 mv[CT];
    static_for<0, CT>([&](auto N) {
        constexpr int n = decltype(N)::value;
        const float4 b = a.bias ? *reinterpret_cast<const float4 *>(a.bias + 16 * n + 4 * kq) : zero4;
        float4 v = make_float4(acc[n][0][0] + acc[n][1][0] + b.x, acc[n][0][1] + acc[n][1][1] + b.y, acc[n][0][2] + acc[n][1][2] + b.z,
                               acc[n][0][3] + acc[n][1][3] + b.w);
        mv[n] = relu4(v, a.relu != 0);
        if (live) *reinterpret_cast<float4 *>(a.out + (size_t)row * COUT + 16 * n + 4 * kq) = mv[n];
    });
    if constexpr (C2 > 0) {
        constexpr int NT = C2 / 16;
        const float4 *w1 = reinterpret_cast<const float4 *>(a.w1) + lane;
        f32x4 s[NT][2];
        static_for<0, NT>([&](auto N) { s[decltype(N)::value][0] = s[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
        static_for<0, CT>([&](auto G) {
            constexpr int g = decltype(G)::value;
            float4 wv[NT];
            static_for<0, NT>([&](auto N) { wv[decltype(N)::value] = w1[(decltype(N)::value * CT + g) * 64]; });
            static_for<0, 4>([&](auto R) {
                constexpr int r = decltype(R)::value;
                static_for<0, NT>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    s[n][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4<r>(wv[n]), comp4<r>(mv[g]), s[n][r & 1], 0, 0, 0);
                });
            });
        });
        if (live) {
            static_for<0, NT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                const float4 b = a.b1 ? *reinterpret_cast<const float4 *>(a.b1 + 16 * n + 4 * kq) : zero4;
                *reinterpret_cast<float4 *>(a.y1 + (size_t)row * C2 + 16 * n + 4 * kq) =
                    relu4(make_float4(s[n][0][0] + s[n][1][0] + b.x, s[n][0][1] + s[n][1][1] + b.y, s[n][0][2] + s[n][1][2] + b.z,
                                      s[n][0][3] + s[n][1][3] + b.w), true);
            });
        }
    }
}

} // namespace

extern "C" size_t sis3d_conv3d_k2s2_projected_sparse_workspace_bytes(int X, int Y, int Z)
{
    if (X <= 0 || Y <= 0 || Z <= 0) return 0;
    const size_t nout = (size_t)(X / 2) * (Y / 2) * (Z / 2);
    const size_t nb = (nout + 255) / 256;
    return ((nout + 15) & ~(size_t)15) + (nb + 4) * sizeof(int) + nout * sizeof(int) + 64;
}

extern "C" int sis3d_conv3d_k2s2_projected_sparse(const int32_t *vox2pix, const float *feat_rows, int nslots, int64_t npix, int X, int Y, int Z,
                                                  int cin, const float *w_pw16, const float *bias, int cout, int relu, float *out,
                                                  const float *w1_pw16, const float *b1, int c2, float *y1, void *workspace,
                                                  size_t workspace_bytes, sis3d_stream_t stream)
{
    if (!vox2pix || !feat_rows || !w_pw16 || !out || !workspace || nslots < 0 || npix <= 0 || X <= 0 || Y <= 0 || Z <= 0) return SIS3D_EINVAL;
    if ((X | Y | Z) & 1) return SIS3D_EINVAL;
    if (c2 < 0 || (c2 > 0 && (!w1_pw16 || !y1))) return SIS3D_EINVAL;
    if (cin != CIN || cout != COUT || (c2 != 0 && c2 != 32) || nslots > MAXSL) return SIS3D_EUNSUPPORTED;
    const int64_t nout64 = (int64_t)(X / 2) * (Y / 2) * (Z / 2);
    if (nout64 >= (1 << 24)) return SIS3D_EUNSUPPORTED;                            // the list packs the row index into 24 bits
    if (workspace_bytes < sis3d_conv3d_k2s2_projected_sparse_workspace_bytes(X, Y, Z)) return SIS3D_EWORKSPACE;
    ProjArgs a;
    a.tab = vox2pix; a.rows = feat_rows; a.nslots = nslots; a.npix = npix; a.nvox = (int64_t)X * Y * Z; a.X = X; a.Y = Y; a.Z = Z;
    a.w = w_pw16; a.bias = bias; a.relu = relu; a.out = out; a.w1 = w1_pw16; a.b1 = b1; a.y1 = y1;
    a.nout = (int)nout64; a.nblocks = (a.nout + 255) / 256;
    char *ws = (char *)workspace;
    a.mask = (uint8_t *)ws; ws += ((size_t)a.nout + 15) & ~(size_t)15;
    a.counts = (int *)ws; ws += (size_t)a.nblocks * sizeof(int);
    a.total = (int *)ws; ws += 4 * sizeof(int);
    a.list = (int *)ws;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(proj_mask_kernel, dim3(a.nblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(proj_compact_kernel, dim3(a.nblocks), dim3(256), 0, st, a);
    const int fill_blocks = 1024;
    const int tiles = (a.nout + 63) / 64;
    if (c2 == 32) {
        hipLaunchKernelGGL((proj_fill_kernel<32>), dim3(fill_blocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL((proj_tile_kernel<32>), dim3(tiles), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((proj_fill_kernel<0>), dim3(fill_blocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL((proj_tile_kernel<0>), dim3(tiles), dim3(256), 0, st, a);
    }
    return sis3d_check_launch();
}
