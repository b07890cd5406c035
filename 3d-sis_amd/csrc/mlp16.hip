// RoI classifier head on gfx950, latency form: Linear(8192,256)+ReLU -> Linear(256,256)+ReLU -> Linear(256,128)+ReLU ->
// {Linear(128,NC), Linear(128,6NC)} + softmax + argmax   (lib/nets/backbones.py:92-96,225-231, lib/nets/network.py:589-604).
//
// mlp.hip's pair (32x32x2 tiles, LDS-staged operands, 7 tail workgroups walking three layers of 64-cycle MFMAs) sits on
// every chunk's critical path for 15 + 34 us although the work is 0.9 GFLOP.  Same arithmetic here on 16x16x4 tiles in the
// transposed, register-chained form of pointwise.hip:
//   fc16_splitk_kernel   K = 8192 cut into slices of 256 channels x column groups of 32; a workgroup = 4 waves = 4 adjacent
//                        slices (summed through LDS), grid = 8 slice groups x 8 column groups x 4 row-tile groups = 256; every
//                        wave holds its 32 KB weight block in REGISTERS (2 x 16 fragments) and sweeps its 16-row tiles, activation rows loaded 16 B per lane straight from global, the next
//                        tile's rows in flight during the MFMAs; deterministic partial sums, no atomics.
//                        Row tiles past the device-side count of live proposals are not touched.
//   mlp16_tail_kernel    one workgroup of 8 waves per 16-row tile: slice sum + bias + ReLU into LDS, then the three small
//                        layers with one or two 16-column tiles per wave (a layer's weight fragments are requested while the
//                        previous layer multiplies), the
//                        activations handed from layer to layer through 16 KB of LDS; softmax / argmax per row.
// Weights: pw16 fragment order [cout/16][cin/16][64][4] (sis3d_conv_pw16_pack_weight).
#include "common.h"
#include "mfma16.h"
#include <float.h>

namespace {

constexpr int SLICE = 256;                 // channels per K-slice of the first layer
constexpr int SKG = SLICE / 16;

constexpr int SGRP = 4;                     // K-slices reduced inside one workgroup (one per wave)

// x [R][K] (row stride ldx), w1p pw16 [C1/16][K/16][64][4]; part [K / (SGRP * SLICE)][Rpad][C1].
// grid (slice groups, column groups of 32, 4 row-tile groups): wave w multiplies K-slice SGRP * sg + w, the four partial
// tiles of a row tile are summed through LDS in wave order (deterministic), so the tail reads 8 slabs, not 32.
__global__ __launch_bounds__(256) void fc16_splitk_kernel(const float *__restrict__ x, int R, int K, int ldx, const float *__restrict__ w1p,
                                                          int C1, float *__restrict__ part, int Rpad, const int32_t *__restrict__ nrows)
{
    __shared__ __attribute__((aligned(16))) float red[SGRP * 2 * 256];
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = blockIdx.x, cg = blockIdx.y, mg = blockIdx.z;
    const int slice = SGRP * sg + wave;
    const int live = nrows ? min(nrows[0], R) : R;
    const int nmt = (live + 15) / 16;
    if (mg >= nmt) return;
    float4 w[2][SKG];
    static_for<0, 2>([&](auto N) {
        constexpr int n = decltype(N)::value;
        static_for<0, SKG>([&](auto G) {
            constexpr int g = decltype(G)::value;
            w[n][g] = reinterpret_cast<const float4 *>(w1p)[((size_t)(2 * cg + n) * (K / 16) + slice * SKG + g) * 64 + lane];
        });
    });
    auto load_rows = [&](int mt, float4 (&a)[SKG]) {
        const int row = min(16 * mt + li, R - 1);
        const float *p = x + (size_t)row * ldx + slice * SLICE + 4 * q;
        static_for<0, SKG>([&](auto G) { a[decltype(G)::value] = *reinterpret_cast<const float4 *>(p + 16 * decltype(G)::value); });
    };
    float4 a[SKG];
    load_rows(mg, a);
    for (int mt = mg; mt < nmt; mt += 4) {
        float4 an[SKG];
        const bool more = mt + 4 < nmt;
        if (more) load_rows(mt + 4, an);
        f32x4 acc[2];
        gemm_t<2, SKG>(w, a, acc);
        *reinterpret_cast<f32x4 *>(red + ((wave * 2 + 0) * 64 + lane) * 4) = acc[0];
        *reinterpret_cast<f32x4 *>(red + ((wave * 2 + 1) * 64 + lane) * 4) = acc[1];
        __syncthreads();
        if (threadIdx.x < 128) {
            const int n = threadIdx.x >> 6;                // column tile 0 / 1 of this group, same lane layout as the accumulators
            const float4 *src = reinterpret_cast<const float4 *>(red) + n * 64 + lane;
            const float4 s0 = src[0], s1 = src[128], s2 = src[256], s3 = src[384];
            float4 v;
            v.x = (s0.x + s1.x) + (s2.x + s3.x); v.y = (s0.y + s1.y) + (s2.y + s3.y);
            v.z = (s0.z + s1.z) + (s2.z + s3.z); v.w = (s0.w + s1.w) + (s2.w + s3.w);
            *reinterpret_cast<float4 *>(part + ((size_t)sg * Rpad + 16 * mt + li) * C1 + 32 * cg + 16 * n + 4 * q) = v;
        }
        __syncthreads();
        if (more) static_for<0, SKG>([&](auto G) { a[decltype(G)::value] = an[decltype(G)::value]; });
    }
}

struct Tail16Args {
    const float *part;     // [S][Rpad][C1]
    int S, R, Rpad;
    const float *b1;
    const float *w2, *b2;  // C1 -> C2   (pw16)
    const float *w3, *b3;  // C2 -> C3
    const float *wh, *bh;  // C3 -> NC + 6NC (rows padded to a multiple of 16 with zeros)
    int NC;
    float *cls_score, *cls_prob, *bbox_pred;
    int64_t *cls_pred;
    const int32_t *nrows;
};

// NT out^T tiles (16 columns each; tile j of this wave = n0 + j * nstep, tiles >= ntiles are skipped) of
// act(in[16][CIN] * W^T + b): B operand rows from the LDS tile, result written back to LDS as 16 B per lane (the D^T layout
// is row = lane & 15, columns 4 (lane >> 4) + r)
template <int CIN, int NT>
__device__ __forceinline__ void dense16(const float *tin, int in_stride, const float4 (&w)[NT][CIN / 16], const float *bias, int n0,
                                        int nstep, int ntiles, int cout_valid, bool relu, float *tout, int out_stride, int lane)
{
    constexpr int KG = CIN / 16;
    const int li = lane & 15, q = lane >> 4;
    float4 xin[KG];
    static_for<0, KG>([&](auto G) {
        xin[decltype(G)::value] = *reinterpret_cast<const float4 *>(tin + li * in_stride + 16 * decltype(G)::value + 4 * q);
    });
    f32x4 acc[NT];
    gemm_t<NT, KG>(w, xin, acc);
    static_for<0, NT>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int n = n0 + j * nstep;
        if (n < ntiles) {
            const int c0 = 16 * n + 4 * q;
            float4 o;
            float *po = reinterpret_cast<float *>(&o);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[j][r] + ((bias && c0 + r < cout_valid) ? bias[c0 + r] : 0.0f);
                po[r] = relu ? fmaxf(v, 0.0f) : v;
            }
            *reinterpret_cast<float4 *>(tout + li * out_stride + c0) = o;
        }
    });
}

template <int CIN, int NT>
__device__ __forceinline__ void load_w16(float4 (&w)[NT][CIN / 16], const float *wp, int n0, int nstep, int ntiles, int lane)
{
    static_for<0, NT>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int n = min(n0 + j * nstep, ntiles - 1);
        static_for<0, CIN / 16>([&](auto G) {
            constexpr int g = decltype(G)::value;
            w[j][g] = reinterpret_cast<const float4 *>(wp)[((size_t)n * (CIN / 16) + g) * 64 + lane];
        });
    });
}

template <int C1, int C2, int C3>
__global__ __launch_bounds__(512) void mlp16_tail_kernel(const Tail16Args a)
{
    constexpr int WAVES = 8;
    constexpr int NT2 = (C2 / 16 + WAVES - 1) / WAVES, NT3 = (C3 / 16 + WAVES - 1) / WAVES, NTH = 2;
    constexpr int S1 = C1 + 4, S2 = C2 + 4, S3 = C3 + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 16;
    const int nh = a.NC * 7, nht = (nh + 15) / 16, SH = nht * 16 + 4;
    const int live = a.nrows ? min(a.nrows[0], a.R) : a.R;
    if (m0 >= live) {
        // padded rows only: defined (zero) outputs, no work
        for (int idx = tid; idx < 16 * nh; idx += 64 * WAVES) {
            const int row = m0 + idx / nh, c = idx % nh;
            if (row >= a.R) continue;
            if (c < a.NC) { a.cls_score[(size_t)row * a.NC + c] = 0.0f; a.cls_prob[(size_t)row * a.NC + c] = 0.0f; }
            else a.bbox_pred[(size_t)row * (6 * a.NC) + (c - a.NC)] = 0.0f;
        }
        if (tid < 16 && m0 + tid < a.R) a.cls_pred[m0 + tid] = 0;
        return;
    }
    float *t1 = lds, *t2 = t1 + 16 * S1, *t3 = t2 + 16 * S2, *th = t3 + 16 * S3;
    // the second layer's weight fragments are requested before anything else; the later layers' while the previous one multiplies
    float4 w2[NT2][C1 / 16];
    load_w16<C1, NT2>(w2, a.w2, wave, WAVES, C2 / 16, lane);
    // ---- fc1: sum of the K-slices (fixed order: deterministic) + bias + ReLU -> t1 [16][C1]
    for (int idx = tid; idx < 16 * (C1 / 4); idx += 64 * WAVES) {
        const int row = idx / (C1 / 4), c = (idx % (C1 / 4)) * 4;
        const float4 *src = reinterpret_cast<const float4 *>(a.part + ((size_t)m0 + row) * C1 + c);
        const size_t sstride = (size_t)a.Rpad * C1 / 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < a.S; s0 += 8) {
            float4 p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = src[(size_t)min(s0 + j, a.S - 1) * sstride];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < a.S) { v.x += p[j].x; v.y += p[j].y; v.z += p[j].z; v.w += p[j].w; }
        }
        const float4 bb = *reinterpret_cast<const float4 *>(a.b1 + c);
        *reinterpret_cast<float4 *>(t1 + row * S1 + c) = make_float4(fmaxf(v.x + bb.x, 0.f), fmaxf(v.y + bb.y, 0.f), fmaxf(v.z + bb.z, 0.f),
                                                                    fmaxf(v.w + bb.w, 0.f));
    }
    __syncthreads();
    float4 w3[NT3][C2 / 16];
    load_w16<C2, NT3>(w3, a.w3, wave, WAVES, C3 / 16, lane);
    dense16<C1, NT2>(t1, S1, w2, a.b2, wave, WAVES, C2 / 16, C2, true, t2, S2, lane);
    __syncthreads();
    float4 wh[NTH][C3 / 16];
    load_w16<C3, NTH>(wh, a.wh, wave, WAVES, nht, lane);
    dense16<C2, NT3>(t2, S2, w3, a.b3, wave, WAVES, C3 / 16, C3, true, t3, S3, lane);
    __syncthreads();
    dense16<C3, NTH>(t3, S3, wh, a.bh, wave, WAVES, nht, nh, false, th, SH, lane);
    __syncthreads();
    // heads out of the LDS tile: columns [0,NC) = class scores, [NC, 7NC) = box deltas
    for (int idx = tid; idx < 16 * nh; idx += 64 * WAVES) {
        const int row = idx / nh, c = idx % nh;
        if (m0 + row >= a.R) continue;
        const float v = th[row * SH + c];
        if (c < a.NC) a.cls_score[(size_t)(m0 + row) * a.NC + c] = v;
        else a.bbox_pred[(size_t)(m0 + row) * (6 * a.NC) + (c - a.NC)] = v;
    }
    // softmax (F.softmax, network.py:597) + argmax (torch.max(...)[1]: first maximum): one thread per row
    if (tid < 16 && m0 + tid < a.R) {
        const float *srow = th + tid * SH;
        float mx = srow[0];
        int am = 0;
        for (int c = 1; c < a.NC; ++c)
            if (srow[c] > mx) { mx = srow[c]; am = c; }
        float sum = 0.0f;
        for (int c = 0; c < a.NC; ++c) sum += expf(srow[c] - mx);
        for (int c = 0; c < a.NC; ++c) a.cls_prob[(size_t)(m0 + tid) * a.NC + c] = expf(srow[c] - mx) / sum;
        a.cls_pred[m0 + tid] = am;
    }
}

} // namespace

extern "C" size_t sis3d_classifier16_workspace_floats(int R, int K, int C1)
{
    if (R <= 0 || K <= 0 || C1 <= 0 || (K % (SGRP * SLICE))) return 0;
    return (size_t)(K / (SGRP * SLICE)) * ((R + 15) / 16 * 16) * C1;
}

extern "C" int sis3d_classifier16_forward(const float *x, int R, const int32_t *nrows_dev, int K, int ldx, const float *w1p,
                                          const float *b1, int C1, const float *w2p, const float *b2, int C2, const float *w3p,
                                          const float *b3, int C3, const float *whp, const float *bh, int NC, float *cls_score,
                                          float *cls_prob, int64_t *cls_pred, float *bbox_pred, float *workspace,
                                          size_t workspace_floats, sis3d_stream_t stream)
{
    if (R < 0) return SIS3D_EINVAL;
    if (R == 0) return SIS3D_OK;
    if (!x || !w1p || !b1 || !w2p || !b2 || !w3p || !b3 || !whp || !bh || !cls_score || !cls_prob || !cls_pred || !bbox_pred)
        return SIS3D_EINVAL;
    if ((K % (SGRP * SLICE)) || (ldx % 4) || NC <= 0 || 7 * NC > 16 * 16) return SIS3D_EUNSUPPORTED;      // heads: <= 2 tiles per wave
    if (!(C1 == 256 && C2 == 256 && C3 == 128)) return SIS3D_EUNSUPPORTED;      // the reference's classifier (backbones.py:225-231)
    const int Rpad = (R + 15) / 16 * 16, S = K / (SGRP * SLICE);
    if (!workspace || workspace_floats < (size_t)S * Rpad * C1) return SIS3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(fc16_splitk_kernel, dim3(S, C1 / 32, 4), dim3(256), 0, st, x, R, K, ldx, w1p, C1, workspace, Rpad, nrows_dev);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    Tail16Args a;
    a.part = workspace; a.S = S; a.R = R; a.Rpad = Rpad; a.b1 = b1; a.w2 = w2p; a.b2 = b2; a.w3 = w3p; a.b3 = b3; a.wh = whp; a.bh = bh;
    a.NC = NC; a.cls_score = cls_score; a.cls_prob = cls_prob; a.bbox_pred = bbox_pred; a.cls_pred = cls_pred; a.nrows = nrows_dev;
    const int nht = (7 * NC + 15) / 16;
    const size_t lds = (size_t)16 * ((256 + 4) + (256 + 4) + (128 + 4) + (nht * 16 + 4)) * sizeof(float);
    hipLaunchKernelGGL((mlp16_tail_kernel<256, 256, 128>), dim3(Rpad / 16), dim3(512), lds, st, a);
    return sis3d_check_launch();
}
