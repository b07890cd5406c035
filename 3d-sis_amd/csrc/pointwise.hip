// 1x1x1 convolutions of the Bottleneck (lib/nets/backbones.py:27-40) as register-chained MFMA GEMMs on gfx950.
//
// Replaces the cuDNN / elementwise kernels behind
//     out  = relu(conv3(y2) + b3 + x)          Bottleneck.conv3 + residual + ReLU        (backbones.py:33-40)
//     out2 = relu(conv1_next(out) + b1)        the NEXT block's conv1 + ReLU              (backbones.py:29-31)
// and the stand-alone conv1 of the first block of a stage.  These layers move 10-28 MB for 0.1-0.2 GFLOP: they are
// HBM / L2-latency bound, and the generic implicit-GEMM kernel (conv3d.hip: LDS halo staging, barriers, split-K through
// LDS) spends 10-19 us on them.  Here a wave owns a tile of 16 voxels and NOTHING is staged:
//   * both GEMMs are computed transposed, D^T[cout][voxel] = W[cout][cin] * Y^T[cin][voxel], on v_mfma_f32_16x16x4_f32:
//     the B operand is the activation tile, lane (voxel = lane & 15, k = lane >> 4) loads 16 B = channels 16 g + 4 k .. +3
//     straight from global memory (K order permuted so the four MFMAs of a group consume the four floats of one load);
//   * the result comes out as lane (voxel = lane & 15; couts 16 n + 4 (lane >> 4) + r) -- exactly the B-operand layout of
//     the NEXT GEMM, so conv3 -> (+bias, +residual, ReLU) -> conv1_next chains through registers: no LDS, no shuffle;
//     residual loads and all stores are 16 B per lane;
//   * weights are repacked once into fragment order [cout/16][cin/16][lane][4] and live in registers;
//   * wide layers (cout >= 64) split the output channels over the 4 waves of a workgroup (the 24x12x24 grid has only
//     432 voxel tiles); the second GEMM is then split over its reduction (each wave holds its quarter of the
//     channels) and summed through 4-16 KB of LDS.
// FMA contraction irrelevant (MFMA = fmaf chain); tolerance 1e-4.
#include "common.h"
#include "mfma16.h"

namespace {

struct PwArgs {
    const float *in;
    int in_stride;
    const float *w1p, *b1;       // stage 1 (conv3 / a plain conv1): pw16-packed weights, bias (may be NULL)
    const float *res;            // residual rows (may be NULL)
    int res_stride;
    float *out;
    int out_stride, out_coff, flags1;
    const float *w2p, *b2;       // stage 2 (the next block's conv1); unused when C2 == 0
    float *out2;
    int out2_stride, flags2;
    int nvox;                    // OUTPUT voxels
    int iY, iZ, oY, oZ;          // TAPS == 8 (k2 s2): input grid (Y, Z) and output grid (OY, OZ) extents
};

// WS = 1: a wave owns whole voxel tiles (all C1 and C2 output channels); grid-stride loop with the weights in registers.
// WS = 4: the 4 waves of a workgroup share one voxel tile: wave w computes output channels [w C1/4, (w+1) C1/4) of stage
//         1, then its K-quarter of stage 2; the partial stage-2 tiles are summed through LDS.
// TAPS = 8: stage 1 is a Conv3d(C0, C1, k=2, s=2) (the stems geometry1[4] / color[4], backbones.py:193,207): its reduction runs
//           over the 2x2x2 input voxels of an output voxel, i.e. 8 gathered rows per lane instead of one -- the same GEMM with
//           K = 8 C0 (weights packed with K index = tap * C0 + ci, tap = 4 dx + 2 dy + dz)
template <int C0, int C1, int C2, int WS, int TAPS = 1>
__global__ __launch_bounds__(256) void pw16_kernel(const PwArgs a)
{
    static_assert(C0 % 16 == 0 && C1 % 16 == 0 && C2 % 16 == 0, "channel counts are multiples of 16");
    static_assert(WS == 1 || (WS == 4 && C1 % 64 == 0), "the 4-wave split needs cout % 64 == 0");
    static_assert(TAPS == 1 || TAPS == 8, "pointwise or k2/s2");
    constexpr int KGC = C0 / 16;              // channel groups per input row
    constexpr int KG1 = TAPS * KGC;           // reduction groups of stage 1
    constexpr bool PREFETCH = TAPS == 1;      // the gathered variant holds 8 rows per lane: no second operand set
    constexpr int NT1 = C1 / 16 / WS;         // stage-1 output tiles of this wave
    constexpr int NT2 = C2 / 16;              // stage-2 output tiles (all of them, partial sums when WS == 4)
    constexpr int NT2A = NT2 > 0 ? NT2 : 1;
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t1_0 = WS == 1 ? 0 : wave * NT1;                       // first stage-1 tile of this wave
    __shared__ __attribute__((aligned(16))) float red[(WS == 4 && NT2 > 0) ? 4 * NT2 * 256 : 4];

    // ---- weights and biases of this wave, once
    float4 w1[NT1][KG1], bb1[NT1];
    static_for<0, NT1>([&](auto N) {
        constexpr int n = decltype(N)::value;
        static_for<0, KG1>([&](auto G) {
            constexpr int g = decltype(G)::value;
            w1[n][g] = reinterpret_cast<const float4 *>(a.w1p)[((size_t)(t1_0 + n) * KG1 + g) * 64 + lane];
        });
        bb1[n] = a.b1 ? *reinterpret_cast<const float4 *>(a.b1 + 16 * (t1_0 + n) + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    float4 w2[NT2A][NT1], bb2[NT2A];
    if constexpr (NT2 > 0) {
        static_for<0, NT2>([&](auto N) {
            constexpr int n = decltype(N)::value;
            static_for<0, NT1>([&](auto G) {
                constexpr int g = decltype(G)::value;
                w2[n][g] = reinterpret_cast<const float4 *>(a.w2p)[((size_t)n * (C1 / 16) + t1_0 + g) * 64 + lane];
            });
            bb2[n] = a.b2 ? *reinterpret_cast<const float4 *>(a.b2 + 16 * n + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        });
    }
    const bool relu1 = a.flags1 & SIS3D_EPI_RELU, relu2 = a.flags2 & SIS3D_EPI_RELU;
    const bool sig1 = a.flags1 & SIS3D_EPI_SIGMOID;             // r6: the mask head's last 1x1x1 conv (backbones.py:286), stage 1 only
    const int nmt = (a.nvox + 15) / 16;
    const int first = WS == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    const int step = WS == 1 ? gridDim.x * 4 : gridDim.x;

    auto load_tile = [&](int mt, float4 (&y)[KG1], float4 (&r)[NT1]) {
        int v = 16 * mt + li;
        v = v < a.nvox ? v : a.nvox - 1;
        if constexpr (TAPS == 1) {
            const float *yp = a.in + (size_t)v * a.in_stride + 4 * q;
            static_for<0, KG1>([&](auto G) { y[decltype(G)::value] = *reinterpret_cast<const float4 *>(yp + 16 * decltype(G)::value); });
        } else {
            const int oz = v % a.oZ, oy = (v / a.oZ) % a.oY, ox = v / (a.oZ * a.oY);
            const float *yp = a.in + ((size_t)(2 * ox * a.iY + 2 * oy) * a.iZ + 2 * oz) * a.in_stride + 4 * q;
            static_for<0, TAPS>([&](auto T) {
                constexpr int t = decltype(T)::value;
                const float *tp = yp + (size_t)(((t >> 2) * a.iY + ((t >> 1) & 1)) * a.iZ + (t & 1)) * a.in_stride;
                static_for<0, KGC>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    y[t * KGC + g] = *reinterpret_cast<const float4 *>(tp + 16 * g);
                });
            });
        }
        if (a.res) {
            const float *rp = a.res + (size_t)v * a.res_stride + 16 * t1_0 + 4 * q;
            static_for<0, NT1>([&](auto N) { r[decltype(N)::value] = *reinterpret_cast<const float4 *>(rp + 16 * decltype(N)::value); });
        } else {
            static_for<0, NT1>([&](auto N) { r[decltype(N)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
        }
    };

    float4 y[KG1], r[NT1];
    if (first < nmt) load_tile(first, y, r);
    for (int mt = first; mt < nmt; mt += step) {
        // the next tile's operands are requested before this tile's MFMAs
        float4 yn[PREFETCH ? KG1 : 1], rn[PREFETCH ? NT1 : 1];
        const bool more = mt + step < nmt;
        if constexpr (PREFETCH) {
            if (more) load_tile(mt + step, yn, rn);
        }
        const int v = 16 * mt + li;
        const bool ok = v < a.nvox;
        f32x4 acc[NT1];
        gemm_t<NT1, KG1>(w1, y, acc);
        float4 z[NT1];
        static_for<0, NT1>([&](auto N) {
            constexpr int n = decltype(N)::value;
            float4 t;
            t.x = acc[n][0] + bb1[n].x + r[n].x; t.y = acc[n][1] + bb1[n].y + r[n].y;
            t.z = acc[n][2] + bb1[n].z + r[n].z; t.w = acc[n][3] + bb1[n].w + r[n].w;
            z[n] = relu4(t, relu1);
            if (sig1) z[n] = make_float4(1.0f / (1.0f + expf(-z[n].x)), 1.0f / (1.0f + expf(-z[n].y)), 1.0f / (1.0f + expf(-z[n].z)), 1.0f / (1.0f + expf(-z[n].w)));
            if (ok && a.out) *reinterpret_cast<float4 *>(a.out + (size_t)v * a.out_stride + a.out_coff + 16 * (t1_0 + n) + 4 * q) = z[n];
        });
        if constexpr (NT2 > 0) {
            f32x4 acc2[NT2];
            gemm_t<NT2, NT1>(w2, z, acc2);
            if constexpr (WS == 1) {
                static_for<0, NT2>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    float4 t;
                    t.x = acc2[n][0] + bb2[n].x; t.y = acc2[n][1] + bb2[n].y; t.z = acc2[n][2] + bb2[n].z; t.w = acc2[n][3] + bb2[n].w;
                    if (ok) *reinterpret_cast<float4 *>(a.out2 + (size_t)v * a.out2_stride + 16 * n + 4 * q) = relu4(t, relu2);
                });
            } else {
                static_for<0, NT2>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    *reinterpret_cast<f32x4 *>(red + ((wave * NT2 + n) * 64 + lane) * 4) = acc2[n];
                });
                __syncthreads();
                for (int n = wave; n < NT2; n += 4) {
                    const float4 *src = reinterpret_cast<const float4 *>(red) + n * 64 + lane;
                    const float4 s0 = src[0], s1 = src[NT2 * 64], s2 = src[2 * NT2 * 64], s3 = src[3 * NT2 * 64];
                    const float4 bn = a.b2 ? *reinterpret_cast<const float4 *>(a.b2 + 16 * n + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 t;
                    t.x = (s0.x + s1.x) + (s2.x + s3.x) + bn.x; t.y = (s0.y + s1.y) + (s2.y + s3.y) + bn.y;
                    t.z = (s0.z + s1.z) + (s2.z + s3.z) + bn.z; t.w = (s0.w + s1.w) + (s2.w + s3.w) + bn.w;
                    if (ok) *reinterpret_cast<float4 *>(a.out2 + (size_t)v * a.out2_stride + 16 * n + 4 * q) = relu4(t, relu2);
                }
                if (more) __syncthreads();
            }
        }
        if (more) {
            if constexpr (PREFETCH) {
                static_for<0, KG1>([&](auto G) { y[decltype(G)::value] = yn[decltype(G)::value]; });
                static_for<0, NT1>([&](auto N) { r[decltype(N)::value] = rn[decltype(N)::value]; });
            } else {
                load_tile(mt + step, y, r);
            }
        }
    }
}

// geometry1[0] = Conv3d(2, C1, k=2, s=2, bias=False) + ReLU on the PLANAR 2-channel grid (backbones.py:188), chained into the
// first Bottleneck's conv1 (C1 -> C2, + bias, ReLU; backbones.py:29-31).  K = 2 channels x 8 taps = 16 = ONE reduction
// group: lane (voxel, k) gathers (ci = k >> 1, dx = k & 1) x (dy, dz) -- four scalar loads from the planar grid -- and the
// result tile feeds the second GEMM from registers.  Both outputs are channels-last.
template <int C1, int C2>
__global__ __launch_bounds__(256) void stem_planar_kernel(const float *__restrict__ in, int64_t is_c, int64_t is_x, int64_t is_y, int oX,
                                                          int oY, int oZ, const float *__restrict__ w1p, int flags1,
                                                          float *__restrict__ out, int out_stride, const float *__restrict__ w2p,
                                                          const float *__restrict__ b2, int flags2, float *__restrict__ out2,
                                                          int out2_stride)
{
    constexpr int NT1 = C1 / 16, NT2 = C2 / 16, NT2A = NT2 > 0 ? NT2 : 1;
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = threadIdx.x >> 6;
    float4 w1[NT1][1];
    static_for<0, NT1>([&](auto N) { w1[decltype(N)::value][0] = reinterpret_cast<const float4 *>(w1p)[decltype(N)::value * 64 + lane]; });
    float4 w2[NT2A][NT1], bb2[NT2A];
    if constexpr (NT2 > 0) {
        static_for<0, NT2>([&](auto N) {
            constexpr int n = decltype(N)::value;
            static_for<0, NT1>([&](auto G) { w2[n][decltype(G)::value] = reinterpret_cast<const float4 *>(w2p)[(n * NT1 + decltype(G)::value) * 64 + lane]; });
            bb2[n] = b2 ? *reinterpret_cast<const float4 *>(b2 + 16 * n + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        });
    }
    const bool relu1 = flags1 & SIS3D_EPI_RELU, relu2 = flags2 & SIS3D_EPI_RELU;
    const int nvox = oX * oY * oZ, nmt = (nvox + 15) / 16;
    const int ci = q >> 1, dx = q & 1;
    for (int mt = blockIdx.x * 4 + wave; mt < nmt; mt += gridDim.x * 4) {
        const int v = 16 * mt + li;
        const bool ok = v < nvox;
        const int vc = ok ? v : nvox - 1;
        const int oz = vc % oZ, oy = (vc / oZ) % oY, ox = vc / (oZ * oY);
        const float *p = in + ci * is_c + (int64_t)(2 * ox + dx) * is_x + (int64_t)(2 * oy) * is_y + 2 * oz;
        float4 y[1];
        y[0].x = p[0]; y[0].y = p[1]; y[0].z = p[is_y]; y[0].w = p[is_y + 1];
        f32x4 acc[NT1];
        gemm_t<NT1, 1>(w1, y, acc);
        float4 z[NT1];
        static_for<0, NT1>([&](auto N) {
            constexpr int n = decltype(N)::value;
            z[n] = relu4(make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]), relu1);
            if (ok) *reinterpret_cast<float4 *>(out + (size_t)v * out_stride + 16 * n + 4 * q) = z[n];
        });
        if constexpr (NT2 > 0) {
            f32x4 acc2[NT2];
            gemm_t<NT2, NT1>(w2, z, acc2);
            static_for<0, NT2>([&](auto N) {
                constexpr int n = decltype(N)::value;
                float4 t;
                t.x = acc2[n][0] + bb2[n].x; t.y = acc2[n][1] + bb2[n].y; t.z = acc2[n][2] + bb2[n].z; t.w = acc2[n][3] + bb2[n].w;
                if (ok) *reinterpret_cast<float4 *>(out2 + (size_t)v * out2_stride + 16 * n + 4 * q) = relu4(t, relu2);
            });
        }
    }
}

// The two 1x1x1 RPN heads of one pyramid level as ONE GEMM (lib/nets/network.py:41-42,541-549): rows [0,2A) =
// rpn_cls_score_net, [2A,8A) = rpn_bbox_pred_net, K = 256 channels split over the 4 waves of a workgroup (one voxel tile of
// 16 per workgroup), partial tiles summed through LDS; the epilogue writes the reference's permuted layouts directly --
// score (2,X,Y,Z,A), prob = softmax over the two class planes, bbox (X,Y,Z,6A).  blockIdx.y selects the level (the
// two levels have different anchor counts, hence different NT).
struct HeadLevel {
    const float *in, *wp, *bias;
    float *score, *prob, *bbox;
    int A;
};
struct HeadArgs {
    HeadLevel lv[2];
    int nvox, in_stride;
};

template <int NT, int C0>
__device__ __forceinline__ void rpn_head_body(const HeadLevel &h, int nvox, int in_stride, float *lds)
{
    constexpr int KGW = C0 / 16 / 4;            // channel groups per wave
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 w[NT][KGW];
    static_for<0, NT>([&](auto N) {
        constexpr int n = decltype(N)::value;
        static_for<0, KGW>([&](auto G) {
            constexpr int g = decltype(G)::value;
            w[n][g] = reinterpret_cast<const float4 *>(h.wp)[((size_t)n * (C0 / 16) + KGW * wave + g) * 64 + lane];
        });
    });
    const int nmt = (nvox + 15) / 16;
    for (int mt = blockIdx.x; mt < nmt; mt += gridDim.x) {
    int v = 16 * mt + li;
    v = v < nvox ? v : nvox - 1;
    float4 y[KGW];
    const float *yp = h.in + (size_t)v * in_stride + 16 * KGW * wave + 4 * q;
    static_for<0, KGW>([&](auto G) { y[decltype(G)::value] = *reinterpret_cast<const float4 *>(yp + 16 * decltype(G)::value); });
    f32x4 acc[NT];
    gemm_t<NT, KGW>(w, y, acc);
    // partial tiles -> LDS as [wave][cout][voxel] (cout-major rows of 16 voxels + 1 pad)
    constexpr int RSZ = NT * 16 * 17;
    static_for<0, NT>([&](auto N) {
        constexpr int n = decltype(N)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[wave * RSZ + (16 * n + 4 * q + r) * 17 + li] = acc[n][r];
    });
    __syncthreads();
    const int A = h.A, nout = 8 * A;
    float *fin = lds + 4 * RSZ;                  // [cout][voxel] final sums (+ bias)
    for (int i = threadIdx.x; i < nout * 16; i += 256) {
        const int c = i >> 4, vv = i & 15;
        const int o = c * 17 + vv;
        fin[o] = (lds[o] + lds[RSZ + o]) + (lds[2 * RSZ + o] + lds[3 * RSZ + o]) + (h.bias ? h.bias[c] : 0.0f);
    }
    __syncthreads();
    // score / prob: element (plane p, voxel, anchor a) at (p * nvox + voxel) * A + a ; bbox: voxel * 6A + j
    for (int i = threadIdx.x; i < 16 * 2 * A; i += 256) {
        const int vv = i / (2 * A), c = i % (2 * A);
        const int vox = 16 * mt + vv;
        if (vox >= nvox) continue;
        const int pl = c / A, an = c % A;
        const float s = fin[c * 17 + vv], sp = fin[(pl ? c - A : c + A) * 17 + vv];
        const size_t o = ((size_t)pl * nvox + vox) * A + an;
        h.score[o] = s;
        if (h.prob) {
            const float mx = fmaxf(s, sp);
            const float e = expf(s - mx), ep = expf(sp - mx);
            h.prob[o] = e / (e + ep);
        }
    }
    for (int i = threadIdx.x; i < 16 * 6 * A; i += 256) {
        const int vv = i / (6 * A), j = i % (6 * A);
        const int vox = 16 * mt + vv;
        if (vox < nvox) h.bbox[(size_t)vox * (6 * A) + j] = fin[(2 * A + j) * 17 + vv];
    }
    __syncthreads();                             // the LDS tiles are rewritten by the next voxel tile
    }
}

template <int NTA, int NTB, int C0>
__global__ __launch_bounds__(256) void rpn_heads_kernel(const HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.y == 0) rpn_head_body<NTA, C0>(a.lv[0], a.nvox, a.in_stride, lds);
    else rpn_head_body<NTB, C0>(a.lv[1], a.nvox, a.in_stride, lds);
}


// (Cout,Cin[,1,1,1]) -> [cout/16][cin/16][lane 64][4]: lane (i = lane & 15, k = lane >> 4) holds W[16 n + i][16 g + 4 k + e]
__global__ __launch_bounds__(256) void pack_weight_pw16_kernel(const float *__restrict__ w, int cout, int cin, int nt, int kg,
                                                               float *__restrict__ packed)
{
    const int64_t total = (int64_t)nt * kg * 256;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const int64_t rest = idx >> 8;
        const int g = (int)(rest % kg), n = (int)(rest / kg);
        const int co = 16 * n + (lane & 15), ci = 16 * g + 4 * (lane >> 4) + e;
        packed[idx] = (co < cout && ci < cin) ? w[(int64_t)co * cin + ci] : 0.0f;
    }
}

template <int C0, int C1, int C2, int WS, int TAPS = 1>
int launch_pw(const PwArgs &a, hipStream_t st)
{
    const int nmt = (a.nvox + 15) / 16;
    if constexpr (TAPS != 1) {
        // two voxel tiles per workgroup: a wave's 32 KB of weight fragments (the PMC table showed 55 MB of L2 weight re-reads for
        // 11.6 MB of activations with one tile each) are fetched half as often
        const int nb = WS == 1 ? (nmt + 3) / 4 : (nmt + 1) / 2;
        hipLaunchKernelGGL((pw16_kernel<C0, C1, C2, WS, TAPS>), dim3(nb), dim3(256), 0, st, a);
        return sis3d_check_launch();
    }
    // WS == 1: ~2 waves per SIMD, every wave loops over its tiles with the weights in registers; WS == 4: one tile per
    // workgroup pass
    int blocks = WS == 1 ? (nmt + 3) / 4 : nmt;
    const int cap = WS == 1 ? 512 : 2048;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((pw16_kernel<C0, C1, C2, WS>), dim3(blocks), dim3(256), 0, st, a);
    return sis3d_check_launch();
}

template <int NTA, int NTB>
int launch_heads(const HeadArgs &a, hipStream_t st)
{
    constexpr int NTM = NTA > NTB ? NTA : NTB;
    const size_t lds = (size_t)5 * NTM * 16 * 17 * sizeof(float);
    const int nmt = (a.nvox + 15) / 16;
    hipLaunchKernelGGL((rpn_heads_kernel<NTA, NTB, 256>), dim3((nmt + 1) / 2, 2), dim3(256), lds, st, a);   // 2 tiles per workgroup
    return sis3d_check_launch();
}

} // namespace

extern "C" size_t sis3d_conv_pw16_packed_floats(int cout, int cin)
{
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cout + 15) / 16) * ((cin + 15) / 16) * 256;
}

extern "C" int sis3d_conv_pw16_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream)
{
    if (!w || !packed || cout <= 0 || cin <= 0) return SIS3D_EINVAL;
    const int nt = (cout + 15) / 16, kg = (cin + 15) / 16;
    const int64_t blocks = ((int64_t)nt * kg * 256 + 255) / 256;
    hipLaunchKernelGGL(pack_weight_pw16_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, as_stream(stream), w, cout, cin,
                       nt, kg, packed);
    return sis3d_check_launch();
}

extern "C" int sis3d_conv3d_pw16(const float *in, int64_t nvox, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                 int flags, const float *residual, int res_stride, float *out, int out_stride, int out_coff,
                                 const float *packed_w2, const float *bias2, int cout2, int flags2, float *out2, int out2_stride,
                                 sis3d_stream_t stream)
{
    if (!in || !packed_w || nvox <= 0 || nvox > 0x7fffffff || cin <= 0 || cout <= 0 || cout2 < 0) return SIS3D_EINVAL;
    if ((cin_stride % 4) || cin_stride < cin || (out_stride % 4) || (out_coff % 4) || (res_stride % 4)) return SIS3D_EINVAL;
    if (!out && cout2 == 0) return SIS3D_EINVAL;
    if (cout2 > 0 && (!packed_w2 || !out2 || (out2_stride % 4) || out2_stride < cout2)) return SIS3D_EINVAL;
    if ((flags & ~(SIS3D_EPI_RELU | SIS3D_EPI_RESIDUAL | SIS3D_EPI_SIGMOID)) || (flags2 & ~SIS3D_EPI_RELU)) return SIS3D_EUNSUPPORTED;
    if ((flags & SIS3D_EPI_SIGMOID) && cout2 > 0) return SIS3D_EUNSUPPORTED;          // the sigmoid ends a chain
    if ((flags & SIS3D_EPI_RESIDUAL) && !residual) return SIS3D_EINVAL;
    PwArgs a;
    a.in = in; a.in_stride = cin_stride; a.w1p = packed_w; a.b1 = bias;
    a.res = (flags & SIS3D_EPI_RESIDUAL) ? residual : nullptr; a.res_stride = res_stride;
    a.out = out; a.out_stride = out_stride; a.out_coff = out_coff; a.flags1 = flags;
    a.w2p = packed_w2; a.b2 = bias2; a.out2 = out2; a.out2_stride = out2_stride; a.flags2 = flags2;
    a.nvox = (int)nvox;
    hipStream_t st = as_stream(stream);
    const int key = cin * 1000000 + cout * 1000 + cout2;
    switch (key) {
    case 32 * 1000000 + 32 * 1000 + 32: return launch_pw<32, 32, 32, 1>(a, st);
    case 32 * 1000000 + 32 * 1000 + 0: return launch_pw<32, 32, 0, 1>(a, st);
    case 32 * 1000000 + 64 * 1000 + 32: return launch_pw<32, 64, 32, 4>(a, st);
    case 32 * 1000000 + 64 * 1000 + 0: return launch_pw<32, 64, 0, 4>(a, st);
    case 32 * 1000000 + 128 * 1000 + 32: return launch_pw<32, 128, 32, 4>(a, st);
    case 32 * 1000000 + 128 * 1000 + 0: return launch_pw<32, 128, 0, 4>(a, st);
    case 64 * 1000000 + 128 * 1000 + 64: return launch_pw<64, 128, 64, 4>(a, st);
    case 64 * 1000000 + 128 * 1000 + 0: return launch_pw<64, 128, 0, 4>(a, st);
    case 64 * 1000000 + 64 * 1000 + 0: return launch_pw<64, 64, 0, 4>(a, st);
    case 64 * 1000000 + 32 * 1000 + 0: return launch_pw<64, 32, 0, 1>(a, st);
    case 128 * 1000000 + 64 * 1000 + 0: return launch_pw<128, 64, 0, 4>(a, st);
    case 128 * 1000000 + 32 * 1000 + 0: return launch_pw<128, 32, 0, 1>(a, st);
    case 128 * 1000000 + 128 * 1000 + 0: return launch_pw<128, 128, 0, 4>(a, st);
    default: return SIS3D_EUNSUPPORTED;
    }
}

extern "C" int sis3d_conv3d_k2s2_pw16(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                                      int cout, int flags, float *out, int out_stride, int out_coff, const float *packed_w2,
                                      const float *bias2, int cout2, int flags2, float *out2, int out2_stride, sis3d_stream_t stream)
{
    if (!in || !packed_w || X < 2 || Y < 2 || Z < 2 || cin <= 0 || cout <= 0 || cout2 < 0) return SIS3D_EINVAL;
    if ((cin_stride % 4) || cin_stride < cin || (out_stride % 4) || (out_coff % 4)) return SIS3D_EINVAL;
    if (!out && cout2 == 0) return SIS3D_EINVAL;
    if (cout2 > 0 && (!packed_w2 || !out2 || (out2_stride % 4) || out2_stride < cout2)) return SIS3D_EINVAL;
    if ((flags & ~SIS3D_EPI_RELU) || (flags2 & ~SIS3D_EPI_RELU)) return SIS3D_EUNSUPPORTED;
    const int64_t nvox = (int64_t)(X / 2) * (Y / 2) * (Z / 2);
    if (nvox > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    PwArgs a;
    a.in = in; a.in_stride = cin_stride; a.w1p = packed_w; a.b1 = bias; a.res = nullptr; a.res_stride = 0;
    a.out = out; a.out_stride = out_stride; a.out_coff = out_coff; a.flags1 = flags;
    a.w2p = packed_w2; a.b2 = bias2; a.out2 = out2; a.out2_stride = out2_stride; a.flags2 = flags2;
    a.nvox = (int)nvox; a.iY = Y; a.iZ = Z; a.oY = Y / 2; a.oZ = Z / 2;
    hipStream_t st = as_stream(stream);
    const int key = cin * 1000000 + cout * 1000 + cout2;
    switch (key) {
    case 32 * 1000000 + 128 * 1000 + 32: return launch_pw<32, 128, 32, 4, 8>(a, st);
    case 32 * 1000000 + 128 * 1000 + 0: return launch_pw<32, 128, 0, 4, 8>(a, st);
    case 32 * 1000000 + 64 * 1000 + 32: return launch_pw<32, 64, 32, 4, 8>(a, st);
    case 32 * 1000000 + 64 * 1000 + 0: return launch_pw<32, 64, 0, 4, 8>(a, st);
    case 64 * 1000000 + 64 * 1000 + 32: return launch_pw<64, 64, 32, 4, 8>(a, st);
    case 64 * 1000000 + 64 * 1000 + 0: return launch_pw<64, 64, 0, 4, 8>(a, st);
    case 64 * 1000000 + 128 * 1000 + 32: return launch_pw<64, 128, 32, 4, 8>(a, st);
    default: return SIS3D_EUNSUPPORTED;
    }
}

extern "C" int sis3d_conv3d_stem_planar2(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, int X, int Y, int Z,
                                         const float *packed_w, int cout, int flags, float *out, int out_stride,
                                         const float *packed_w2, const float *bias2, int cout2, int flags2, float *out2,
                                         int out2_stride, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || X < 2 || Y < 2 || Z < 2 || (out_stride % 4) || out_stride < cout) return SIS3D_EINVAL;
    if (cout2 > 0 && (!packed_w2 || !out2 || (out2_stride % 4) || out2_stride < cout2)) return SIS3D_EINVAL;
    if ((flags & ~SIS3D_EPI_RELU) || (flags2 & ~SIS3D_EPI_RELU)) return SIS3D_EUNSUPPORTED;
    const int oX = X / 2, oY = Y / 2, oZ = Z / 2;
    const int64_t nmt = ((int64_t)oX * oY * oZ + 15) / 16;
    if (nmt > 0x7ffffff) return SIS3D_EUNSUPPORTED;
    int blocks = (int)((nmt + 3) / 4);
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = as_stream(stream);
#define STEM(C1, C2)                                                                                                              \
    hipLaunchKernelGGL((stem_planar_kernel<C1, C2>), dim3(blocks), dim3(256), 0, st, in, is_c, is_x, is_y, oX, oY, oZ, packed_w, flags, \
                       out, out_stride, packed_w2, bias2, flags2, out2, out2_stride);                                             \
    return sis3d_check_launch();
    if (cout == 32 && cout2 == 32) { STEM(32, 32) }
    if (cout == 32 && cout2 == 0) { STEM(32, 0) }
    if (cout == 64 && cout2 == 32) { STEM(64, 32) }
    if (cout == 64 && cout2 == 0) { STEM(64, 0) }
#undef STEM
    return SIS3D_EUNSUPPORTED;
}

extern "C" int sis3d_rpn_heads(const float *in1, const float *packed_w1, const float *bias1, int anchors1, float *score1, float *prob1,
                               float *bbox1, const float *in2, const float *packed_w2, const float *bias2, int anchors2, float *score2,
                               float *prob2, float *bbox2, int64_t nvox, int cin, int cin_stride, sis3d_stream_t stream)
{
    if (!in1 || !in2 || !packed_w1 || !packed_w2 || !score1 || !score2 || !bbox1 || !bbox2 || nvox <= 0 || nvox > 0x7fffffff)
        return SIS3D_EINVAL;
    if (cin != 256 || (cin_stride % 4) || cin_stride < cin || anchors1 <= 0 || anchors2 <= 0) return SIS3D_EUNSUPPORTED;
    HeadArgs a;
    a.lv[0] = HeadLevel{in1, packed_w1, bias1, score1, prob1, bbox1, anchors1};
    a.lv[1] = HeadLevel{in2, packed_w2, bias2, score2, prob2, bbox2, anchors2};
    a.nvox = (int)nvox; a.in_stride = cin_stride;
    hipStream_t st = as_stream(stream);
    const int nta = (8 * anchors1 + 15) / 16, ntb = (8 * anchors2 + 15) / 16;
    if (nta == 2 && ntb == 6) return launch_heads<2, 6>(a, st);       // ScanNet: 3 + 11 anchors
    if (nta == 2 && ntb == 3) return launch_heads<2, 3>(a, st);       // SUNCG: 3 + 6 anchors
    return SIS3D_EUNSUPPORTED;
}
