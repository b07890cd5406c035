// RoI classifier head on gfx950: Linear(8192,256)+ReLU -> Linear(256,256)+ReLU -> Linear(256,128)+ReLU ->
// {Linear(128,NC), Linear(128,6NC)} + softmax + argmax   (lib/nets/backbones.py:92-96,225-231, lib/nets/network.py:589-604).
//
// The reference runs five cuBLAS GEMMs + four elementwise kernels on R <= 200 rows; rocBLAS picks one-workgroup
// macro-tiles for M = 200 (55 us for the first layer alone).  Here:
//   fc_splitk_kernel  the 8192-deep first layer as a split-K GEMM on v_mfma_f32_32x32x2_f32: 7 row tiles x 2 column
//                     groups x 16 K-slices = 224 workgroups stream the 8 MB weight once; partial sums go to a small
//                     workspace (deterministic: no atomics).
//   mlp_tail_kernel   per 32-row tile: sum the K-slices + bias + ReLU, then the three remaining layers back to back with
//                     the activations held in LDS, softmax / argmax in registers.
// Weights use the same fragment-order packing as the convolutions (sis3d_conv_pack_weight with ksize 1).
#include "common.h"
#include <float.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FC_CK = 128;      // channels per LDS chunk
constexpr int FC_NW = 4;        // waves along N (32 cols each)
constexpr int FC_KW = 4;        // waves splitting the k-groups of a chunk

// x [R][K] row-major (row stride ldx), wp packed [N/32][K/8][64][4]; part [S][Rpad][N]
__global__ __launch_bounds__(64 * FC_NW *FC_KW) void fc_splitk_kernel(const float *__restrict__ x, int R, int K, int ldx,
                                                                     const float *__restrict__ wp, int N, int chunks_per_slice,
                                                                     float *__restrict__ part, int Rpad, const int32_t *__restrict__ nrows)
{
    if (nrows && (int)blockIdx.x * 32 >= nrows[0]) return;     // a row tile past the live rows: the tail writes its zeros
    constexpr int RS = FC_CK + 4, KGC = FC_CK / 8, KGW = KGC / FC_KW;
    __shared__ __attribute__((aligned(16))) float lds[32 * RS > (FC_KW - 1) * FC_NW * 1024 ? 32 * RS : (FC_KW - 1) * FC_NW * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % FC_NW, kw = wave / FC_NW;
    const int li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32, tile = blockIdx.y * FC_NW + nw, slice = blockIdx.z;
    const int ntiles = N / 32, kgtot = K / 8;
    const int q0 = slice * chunks_per_slice, q1 = q0 + chunks_per_slice;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *aptr = lds + li * RS + 4 * kh + 8 * KGW * kw;
    const int tl = min(tile, ntiles - 1);
    for (int q = q0; q < q1; ++q) {
        if (q > q0) __syncthreads();
        for (int idx = tid; idx < 32 * (FC_CK / 4); idx += 64 * FC_NW * FC_KW) {
            const int row = idx / (FC_CK / 4), c4 = idx % (FC_CK / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + row < R) v = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + row) * ldx + q * FC_CK + c4 * 4);
            *reinterpret_cast<float4 *>(lds + row * RS + c4 * 4) = v;
        }
        __syncthreads();
        float4 bq[KGW], aq[KGW];
#pragma unroll
        for (int g = 0; g < KGW; ++g) {
            bq[g] = reinterpret_cast<const float4 *>(wp)[((size_t)tl * kgtot + q * KGC + KGW * kw + g) * 64 + lane];
            aq[g] = *reinterpret_cast<const float4 *>(aptr + 8 * g);
        }
#pragma unroll
        for (int g = 0; g < KGW; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g].x, bq[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g].y, bq[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g].z, bq[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g].w, bq[g].w, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (kw > 0) {
        float *red = lds + (size_t)((kw - 1) * FC_NW + nw) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kw > 0 || tile >= ntiles) return;
#pragma unroll
    for (int k = 1; k < FC_KW; ++k) {
        const float *red = lds + (size_t)((k - 1) * FC_NW + nw) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
    }
    float *dst = part + ((size_t)slice * Rpad + m0) * N + tile * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = (r & 3) + 8 * (r >> 2) + 4 * kh;
        dst[(size_t)mm * N] = acc[r];
    }
}

struct TailArgs {
    const float *part;     // [S][Rpad][C1]
    int S, R, Rpad, C1;
    const float *b1;
    const float *w2, *b2;  // C1 -> C2
    int C2;
    const float *w3, *b3;  // C2 -> C3
    int C3;
    const float *wh, *bh;  // C3 -> NC + 6NC (padded to a multiple of 32 in the packed weight)
    int NC;
    float *cls_score, *cls_prob, *bbox_pred;   // [R][NC], [R][NC], [R][6NC]
    int64_t *cls_pred;                         // [R]
    const int32_t *nrows;                      // device count of live rows (may be NULL): tiles past it are zero-filled
};

constexpr int TAIL_WAVES = 8;

// one 32 x Cout layer on the LDS tile: out = act(in[32][Cin] * W + b)
__device__ __forceinline__ void dense32(const float *tin, int cin, const float *wp, const float *bias, int cout_valid, int cout_pad,
                                        bool relu, float *tout, int wave, int lane, f32x16 *keep, int keep_tile)
{
    const int li = lane & 31, kh = lane >> 5, kgs = cin / 8, ins = cin + 4, outs = cout_pad + 4;
    for (int nt = wave; nt < cout_pad / 32; nt += TAIL_WAVES) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
        const float *ap = tin + li * ins + 4 * kh;
        const float4 *bp = reinterpret_cast<const float4 *>(wp) + (size_t)nt * kgs * 64 + lane;
        // all weight fragments of (up to) 32 k-groups = 256 input channels are requested up front -- 128 VGPRs, which this
        // 2-waves-per-SIMD kernel has to spare -- so a layer exposes ONE L2 latency instead of one per k-group (the former
        // `#pragma unroll 16` loop was refused by the optimiser: a load, a wait and four MFMAs per iteration)
        for (int gb = 0; gb < kgs; gb += 32) {
            float4 bq[32];
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (gb + j < kgs) bq[j] = bp[(gb + j) * 64];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (gb + j >= kgs) break;
                const float4 av = *reinterpret_cast<const float4 *>(ap + 8 * (gb + j));
                const float4 bv = bq[j];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, c, 0, 0, 0);
            }
        }
        const int co = nt * 32 + li;
        const float bb = (co < cout_valid && bias) ? bias[co] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = c[r] + bb;
            if (relu) v = fmaxf(v, 0.0f);
            c[r] = v;
            tout[mm * outs + co] = v;
        }
        if (keep && nt == keep_tile) *keep = c;
    }
}

__global__ __launch_bounds__(64 * TAIL_WAVES) void mlp_tail_kernel(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 32;
    const int nh = a.NC * 7, nh_pad = (nh + 31) / 32 * 32;
    if (a.nrows && m0 >= a.nrows[0]) {
        // padded rows only (RPN_POST_NMS_TOP_N rows are allocated, `num` are proposals): defined, zero outputs, no work
        for (int idx = tid; idx < 32 * nh; idx += 64 * TAIL_WAVES) {
            const int row = idx / nh, c = idx % nh;
            if (m0 + row >= a.R) continue;
            if (c < a.NC) { a.cls_score[(size_t)(m0 + row) * a.NC + c] = 0.0f; a.cls_prob[(size_t)(m0 + row) * a.NC + c] = 0.0f; }
            else a.bbox_pred[(size_t)(m0 + row) * (6 * a.NC) + (c - a.NC)] = 0.0f;
        }
        if (tid < 32 && m0 + tid < a.R) a.cls_pred[m0 + tid] = 0;
        return;
    }
    const int wmax = max(max(a.C1, nh_pad), max(a.C2, a.C3));
    float *t0 = lds, *t1 = lds + 32 * (wmax + 4);
    // fc1: sum of the K-slices + bias + ReLU -> t0 [32][C1+4]
    const int c4n = a.C1 / 4;
    for (int idx = tid; idx < 32 * c4n; idx += 64 * TAIL_WAVES) {
        const int row = idx / c4n, c = (idx % c4n) * 4;
        const float4 *src = reinterpret_cast<const float4 *>(a.part + ((size_t)m0 + row) * a.C1 + c);
        const size_t sstride = (size_t)a.Rpad * a.C1 / 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int s = 0; s < a.S; ++s) {                   // fixed slice order: deterministic sum
            const float4 p = src[s * sstride];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const float4 bb = *reinterpret_cast<const float4 *>(a.b1 + c);
        float *dst = t0 + row * (a.C1 + 4) + c;
        dst[0] = fmaxf(v.x + bb.x, 0.0f); dst[1] = fmaxf(v.y + bb.y, 0.0f);
        dst[2] = fmaxf(v.z + bb.z, 0.0f); dst[3] = fmaxf(v.w + bb.w, 0.0f);
    }
    __syncthreads();
    dense32(t0, a.C1, a.w2, a.b2, a.C2, a.C2, true, t1, wave, lane, nullptr, -1);
    __syncthreads();
    dense32(t1, a.C2, a.w3, a.b3, a.C3, a.C3, true, t0, wave, lane, nullptr, -1);
    __syncthreads();
    dense32(t0, a.C3, a.wh, a.bh, nh, nh_pad, false, t1, wave, lane, nullptr, -1);
    __syncthreads();
    // heads out of the LDS tile: columns [0,NC) = class scores, [NC, 7NC) = box deltas
    const int hs = nh_pad + 4;
    for (int idx = tid; idx < 32 * nh; idx += 64 * TAIL_WAVES) {
        const int row = idx / nh, c = idx % nh;
        if (m0 + row >= a.R) continue;
        const float v = t1[row * hs + c];
        if (c < a.NC) a.cls_score[(size_t)(m0 + row) * a.NC + c] = v;
        else a.bbox_pred[(size_t)(m0 + row) * (6 * a.NC) + (c - a.NC)] = v;
    }
    // softmax (F.softmax, network.py:597) + argmax (torch.max(...)[1]: first maximum) per row: one thread per row
    if (tid < 32 && m0 + tid < a.R) {
        const float *srow = t1 + tid * hs;
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

extern "C" size_t sis3d_classifier_workspace_floats(int R, int K, int C1)
{
    const int Rpad = (R + 31) / 32 * 32;
    const int slices = K >= 4096 ? 16 : (K >= 1024 ? 4 : 1);
    return (size_t)slices * Rpad * C1;
}

extern "C" int sis3d_classifier_forward(const float *x, int R, int K, int ldx, const float *w1p, const float *b1, int C1,
                                        const float *w2p, const float *b2, int C2, const float *w3p, const float *b3, int C3,
                                        const float *whp, const float *bh, int NC, float *cls_score, float *cls_prob,
                                        int64_t *cls_pred, float *bbox_pred, float *workspace, size_t workspace_floats,
                                        sis3d_stream_t stream)
{
    return sis3d_classifier_forward_n(x, R, nullptr, K, ldx, w1p, b1, C1, w2p, b2, C2, w3p, b3, C3, whp, bh, NC, cls_score, cls_prob,
                                      cls_pred, bbox_pred, workspace, workspace_floats, stream);
}

extern "C" int sis3d_classifier_forward_n(const float *x, int R, const int32_t *nrows_dev, int K, int ldx, const float *w1p,
                                          const float *b1, int C1, const float *w2p, const float *b2, int C2, const float *w3p,
                                          const float *b3, int C3, const float *whp, const float *bh, int NC, float *cls_score,
                                          float *cls_prob, int64_t *cls_pred, float *bbox_pred, float *workspace,
                                          size_t workspace_floats, sis3d_stream_t stream)
{
    if (R < 0) return SIS3D_EINVAL;
    if (R == 0) return SIS3D_OK;
    if (!x || !w1p || !b1 || !w2p || !b2 || !w3p || !b3 || !whp || !bh || !cls_score || !cls_prob || !cls_pred || !bbox_pred)
        return SIS3D_EINVAL;
    if ((K % FC_CK) || (C1 % 32) || (C2 % 32) || (C3 % 32) || C1 > 512 || C2 > 512 || C3 > 512 || NC <= 0 || (ldx % 4)) return SIS3D_EUNSUPPORTED;
    const int Rpad = (R + 31) / 32 * 32;
    const int slices = K >= 4096 ? 16 : (K >= 1024 ? 4 : 1);
    if ((K / FC_CK) % slices) return SIS3D_EUNSUPPORTED;
    if (!workspace || workspace_floats < (size_t)slices * Rpad * C1) return SIS3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(fc_splitk_kernel, dim3(Rpad / 32, (C1 / 32 + FC_NW - 1) / FC_NW, slices), dim3(64 * FC_NW * FC_KW), 0, st, x, R,
                       K, ldx, w1p, C1, (K / FC_CK) / slices, workspace, Rpad, nrows_dev);
    int rc = sis3d_check_launch();
    if (rc) return rc;
    TailArgs a;
    a.part = workspace; a.S = slices; a.R = R; a.Rpad = Rpad; a.C1 = C1; a.b1 = b1;
    a.w2 = w2p; a.b2 = b2; a.C2 = C2; a.w3 = w3p; a.b3 = b3; a.C3 = C3; a.wh = whp; a.bh = bh; a.NC = NC;
    a.cls_score = cls_score; a.cls_prob = cls_prob; a.bbox_pred = bbox_pred; a.cls_pred = cls_pred; a.nrows = nrows_dev;
    const int cmax = C1 > C2 ? (C1 > C3 ? C1 : C3) : (C2 > C3 ? C2 : C3);
    const int nh_pad = (NC * 7 + 31) / 32 * 32;
    const int w = cmax > nh_pad ? cmax : nh_pad;
    const size_t lds = (size_t)2 * 32 * (w + 4) * sizeof(float);
    hipLaunchKernelGGL(mlp_tail_kernel, dim3(Rpad / 32), dim3(64 * TAIL_WAVES), lds, st, a);
    return sis3d_check_launch();
}
