// The ENet 2D encoder of the RGB image path (lib/nets/enet.py:130-694 `create_enet`, run by lib/nets/network.py:203-205 as
// image_enet_trainable(image_enet_fixed(images))) in eval mode, on gfx950: 5 views of 256x328 pixels -> (V,128,32,41) feature maps.
// The arithmetic is small (~5 GFLOP per 5 views) and the layers are tiny; on library operators the pass is ~190 launches of
// 3-8 us.  Here one launch does a whole bottleneck:
//     y2  = prelu(conv2(y1) + b2)                    3x3 (dilation d) or the asymmetric 1x5 -> 5x1 pair, mid -> mid channels
//     out = prelu(conv3(y2) + b3 + skip(x))          1x1 mid -> C, skip = x or (down blocks) maxpool2x2(x) zero-padded in channels
//     y1n = prelu(conv1_next(out) + b1n)             the NEXT bottleneck's 1x1 reduction C -> mid', while `out` is in registers
// with BatchNorm (eval) and the torch7-style dropout scale folded into the weights on the host (sis3d/nets/enet_hip.py).
// Activations are pixels x channels (NHWC) rows; a wave owns 16 consecutive pixels and runs the transposed tile GEMM of mfma16.h
// (D^T[cout][pixel] = W[cout][k] X^T[k][pixel]) whose result is the operand layout of the next GEMM, so the three convolutions
// chain through registers.  Weights are A operands in the pw16 lane order [cout/16][cin/16][64][4]; since r4 a workgroup's four waves
// share ONE copy of them in LDS (LDS-DMA, one L2 round trip together with the tap / skip rows, biases and slopes: enet_block_kernel).
// The two stride-2 projections (2x2, stride 2) and the initial block (3x3 stride-2 conv || 2x2 max-pool, concatenated) have their
// own small kernels.  fp32 throughout; the summation order differs from MIOpen's, results agree to ~1e-6 of the feature scale.
#include "common.h"
#include "mfma16.h"
#include <stdlib.h>

namespace {

struct EnetBlockArgs {
    const float *x;            // block input (skip source): rows of C floats, or (pool_cin > 0) rows of pool_cin floats at 2H x 2W
    const float *y1;           // conv1 output, rows of MID floats at H x W
    const float *w2, *b2, *s2; // conv2 taps [ntaps][MID/16][MID/16][64][4] (9 taps, or 5 taps of the 1x5 conv), bias, PReLU slopes
    const float *w2b;          // asymmetric blocks: the 5 taps of the 5x1 conv
    const float *w3, *b3, *s3; // conv3 [C/16][MID/16][64][4], bias, slopes of the block's output PReLU
    float *out;                // rows of C floats, or NCHW (V,C,H,W) when nchw
    const float *w1n, *b1n, *s1n;
    float *y1n;                // rows of MIDN floats
    int V, H, W, npix;
    int kind, dil;             // kind 0: 3x3 with dilation dil; 1: 1x5 then 5x1
    int pool_cin, nchw;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float prelu1(float v, float s) { return v >= 0.f ? v : v * s; }
__device__ __forceinline__ float4 prelu4(float4 v, float4 s) { return make_float4(prelu1(v.x, s.x), prelu1(v.y, s.y), prelu1(v.z, s.z), prelu1(v.w, s.w)); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
template <int R>
__device__ __forceinline__ float comp(const float4 &v) { return R == 0 ? v.x : R == 1 ? v.y : R == 2 ? v.z : v.w; }

// inline-asm row loads with counted waits (see enet_block_kernel): 16 B per lane from a uniform base + a 32-bit byte offset
__device__ __forceinline__ void gload16(f32x4 &dst, int byte_off, const float *base)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
// an empty volatile asm that "modifies" a register the asm loads wrote: keeps every use of it behind the wait above
__device__ __forceinline__ void touch(f32x4 &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ float4 as_float4(const f32x4 &v) { return make_float4(v[0], v[1], v[2], v[3]); }

// acc[ct][r & 1] += W[ct][g] * X[g]: two accumulators per output tile and the tiles interleaved, so that consecutive MFMAs never
// wait on their own result
template <int NT, int KG>
__device__ __forceinline__ void gemm_acc(const float4 *__restrict__ w, const float4 (&x)[KG], f32x4 (&acc)[NT][2])
{
    float4 wv[NT][KG];
    static_for<0, NT>([&](auto N) { static_for<0, KG>([&](auto G) { wv[decltype(N)::value][decltype(G)::value] = w[(decltype(N)::value * KG + decltype(G)::value) * 64]; }); });
    static_for<0, KG>([&](auto G) {
        constexpr int g = decltype(G)::value;
        static_for<0, 4>([&](auto R) {
            constexpr int r = decltype(R)::value;
            static_for<0, NT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                acc[n][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp<r>(wv[n][g]), comp<r>(x[g]), acc[n][r & 1], 0, 0, 0);
            });
        });
    });
}

// r4: FOUR waves per workgroup (64 pixels) share one copy of the block's weights in LDS.  Round 3 ran one wave per workgroup with the
// weights as A operands straight from L2: 68 KB streamed per 16 pixels, one exposed L2 round trip per tap (12-16 us per launch for
// 272 MFMAs = 3.8 us of matrix work).  Now the workgroup copies conv2's taps, conv3 and the next conv1 into LDS once (<= 72 KB: two
// workgroups per CU), and every A operand is a ds_read_b128 that the MFMAs hide; L2 weight traffic per pixel drops 4x.
template <int C, int MID, int MIDN>
constexpr int enet_lds_float4(int ntaps) { return 64 * (ntaps * (MID / 16) * (MID / 16) + (C / 16) * (MID / 16) + (MIDN / 16) * (C / 16)); }

// ENET_WAVES = 1: the round-3 form (one wave per workgroup, weights as A operands straight from L2, no LDS) with the tap rows
// prefetched; kept as an A/B switch (SIS3D_ENET_WAVES=1)
// ASYM: the block's conv2 is the asymmetric 1x5 -> 5x1 pair (a.kind == 1); a template parameter so that only ONE of the two tap-row
// register sets exists in an instantiation
template <int C, int MID, int MIDN, int ENET_WAVES, bool ASYM>
__global__ __launch_bounds__(64 * ENET_WAVES) void enet_block_kernel(const EnetBlockArgs a)
{
    constexpr int MT = MID / 16, CT = C / 16, NT = MIDN / 16;
    extern __shared__ __attribute__((aligned(16))) float4 lw[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = (blockIdx.x * ENET_WAVES + wave) * 16 + li;
    const bool live = p < a.npix;
    const int pc = live ? p : a.npix - 1;
    const int W = a.W, H = a.H;
    const int x0 = pc % W, y0 = (pc / W) % H;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- the block's weights -> LDS, once per workgroup: [conv2 taps (9, or 5 + 5)][conv3][next conv1], fragment order kept
    constexpr int ntaps = ASYM ? 10 : 9;
    constexpr int n2 = (ASYM ? 5 : 9) * MT * MT * 64, n3 = CT * MT * 64;
    const float4 *l2, *l2b, *l3, *l1;
    if constexpr (ENET_WAVES == 1) {
        l2 = reinterpret_cast<const float4 *>(a.w2); l2b = reinterpret_cast<const float4 *>(a.w2b);
        l3 = reinterpret_cast<const float4 *>(a.w3); l1 = reinterpret_cast<const float4 *>(a.w1n);
    } else {
        l2 = lw; l2b = lw + n2; l3 = lw + ntaps * MT * MT * 64; l1 = l3 + n3;
    }
    if constexpr (ENET_WAVES > 1) {
        // LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KB fragment per instruction, no staging registers): a wave issues
        // its share of the <= 72 fragments back to back -- ONE L2 round trip for the lot (a load / store loop through registers pays
        // one per iteration: measured 0.57 instead of 0.37 ms per 5 views) -- and the single wait sits in front of the barrier below
        constexpr int f2 = (ASYM ? 5 : 9) * MT * MT, f2b = ASYM ? 5 * MT * MT : 0, f3 = CT * MT, f1 = NT * CT;
        const int nf = f2 + f2b + f3 + f1;
        for (int f = wave; f < nf; f += ENET_WAVES) {
            const float *src;
            if (f < f2) src = a.w2 + (size_t)f * 256;
            else if (f < f2 + f2b) src = a.w2b + (size_t)(f - f2) * 256;
            else if (f < f2 + f2b + f3) src = a.w3 + (size_t)(f - f2 - f2b) * 256;
            else src = a.w1n + (size_t)(f - f2 - f2b - f3) * 256;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + lane * 4),
                                             (void __attribute__((address_space(3))) *)(lw + (size_t)f * 64), 16, 0, 0);
        }
    }

    // ---- the skip rows travel while the weights land
    float4 skip[CT];
    if (a.pool_cin > 0) {
        // down blocks: MaxPool2d(2,2) of the input at 2H x 2W, zero channels appended up to C (enet.py's padding layer)
        const int v = pc / (W * H);
        const float *s00 = a.x + ((size_t)(v * 2 * H + 2 * y0) * (2 * W) + 2 * x0) * a.pool_cin + 4 * kq;
        const size_t rowstep = (size_t)2 * W * a.pool_cin;
        static_for<0, CT>([&](auto N) {
            constexpr int n = decltype(N)::value;
            if (16 * n < a.pool_cin) {
                const float *q = s00 + 16 * n;
                skip[n] = max4(max4(ld4(q), ld4(q + a.pool_cin)), max4(ld4(q + rowstep), ld4(q + rowstep + a.pool_cin)));
            } else {
                skip[n] = zero4;
            }
        });
    } else {
        static_for<0, CT>([&](auto N) { skip[decltype(N)::value] = ld4(a.x + (size_t)pc * C + 16 * decltype(N)::value + 4 * kq); });
    }
    // r4 (second half): everything else the launch reads from memory is requested HERE, in the same L2 round trip as the weights and
    // the skip rows: the biases / PReLU slopes of the three stages (they used to be loaded where they are used: three exposed round
    // trips of ~0.7 us each on a 14 us launch) and the tap rows of conv2 (they used to go out behind the barrier: a second round trip)
    float4 pb2[MT], ps2[MT], pb3[CT], ps3[CT];
    static_for<0, MT>([&](auto N) { pb2[decltype(N)::value] = ld4(a.b2 + 16 * decltype(N)::value + 4 * kq); ps2[decltype(N)::value] = ld4(a.s2 + 16 * decltype(N)::value + 4 * kq); });
    static_for<0, CT>([&](auto N) { pb3[decltype(N)::value] = ld4(a.b3 + 16 * decltype(N)::value + 4 * kq); ps3[decltype(N)::value] = ld4(a.s3 + 16 * decltype(N)::value + 4 * kq); });
    constexpr int NTA = NT > 0 ? NT : 1;
    float4 pb1[NTA], ps1[NTA];
    if constexpr (NT > 0)
        static_for<0, NT>([&](auto N) { pb1[decltype(N)::value] = ld4(a.b1n + 16 * decltype(N)::value + 4 * kq); ps1[decltype(N)::value] = ld4(a.s1n + 16 * decltype(N)::value + 4 * kq); });
    const int ybase = (int)(((size_t)pc * MID + 4 * kq) * 4);               // byte offset of this lane's 16 B of its own pixel row
    f32x4 xa[ASYM ? 1 : 9][MT];
    bool okt[9];
    f32x4 xr[ASYM ? 2 : 1][5][MT];
    [[maybe_unused]] auto issue_row = [&](auto DY, auto B) {
        constexpr int dy = decltype(DY)::value - 2, b = decltype(B)::value;
        const bool rowok = (unsigned)(y0 + dy) < (unsigned)H;
        static_for<0, 5>([&](auto DX) {
            constexpr int dx = decltype(DX)::value - 2;
            const bool ok = rowok && (unsigned)(x0 + dx) < (unsigned)W;
            const int off = ybase + (ok ? (dy * W + dx) * MID * 4 : 0);
            static_for<0, MT>([&](auto G) { gload16(xr[b][decltype(DX)::value][decltype(G)::value], off + 64 * decltype(G)::value, a.y1); });
        });
    };
    if constexpr (!ASYM) {
        const int d = a.dil;
        static_for<0, 9>([&](auto T) {
            constexpr int tap = decltype(T)::value, ky = tap / 3 - 1, kx = tap % 3 - 1;
            const int dy = ky * d, dx = kx * d;
            okt[tap] = (unsigned)(y0 + dy) < (unsigned)H && (unsigned)(x0 + dx) < (unsigned)W;
            const int off = ybase + (okt[tap] ? (dy * W + dx) * MID * 4 : 0);
            static_for<0, MT>([&](auto G) { gload16(xa[tap][decltype(G)::value], off + 64 * decltype(G)::value, a.y1); });
        });
    } else {
        issue_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    if constexpr (ENET_WAVES > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA above (the compiler does not count LDS-DMA writes)
        __syncthreads();
    }

    // ---- conv2 (zero padding: taps outside the image contribute nothing)
    f32x4 acc[MT][2];
    static_for<0, MT>([&](auto N) { acc[decltype(N)::value][0] = acc[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
    const float4 *w2 = l2 + lane;
    // The tap rows are loaded by inline asm, ALL of a 3x3 conv's nine taps (or one 5-tap row of the asymmetric pair, one row ahead)
    // before the first MFMA: left to itself hipcc turned `ok ? load : 0` into a branch around eight single-dword loads per tap and a
    // vmcnt(0) per tap, i.e. nine dependent L2 round trips (~1 us each) around 3.8 us of matrix work.  Addresses are clamped to the
    // pixel itself where the tap falls outside the image, the zero is a select after the (counted) wait.
    if constexpr (!ASYM) {
        wait_vm<0>(xa[0][0]);
        static_for<0, 9 * MT>([&](auto I) { touch(xa[decltype(I)::value / MT][decltype(I)::value % MT]); });
        static_for<0, 9>([&](auto T) {
            constexpr int tap = decltype(T)::value;
            float4 xv[MT];
            static_for<0, MT>([&](auto G) { xv[decltype(G)::value] = okt[tap] ? as_float4(xa[tap][decltype(G)::value]) : zero4; });
            gemm_acc<MT, MT>(w2 + tap * (MT * MT * 64), xv, acc);
        });
    } else {
        // enet.py's asymmetric pair: Conv2d(mid, mid, (1,5), padding (0,2), no bias) then Conv2d(mid, mid, (5,1), padding (2,0)): the
        // row y + dy of the intermediate is rebuilt per dy (5 x 5 taps); rows outside the image are the second conv's zero padding
        const float4 *w2b = l2b + lane;
        static_for<0, 5>([&](auto DY) {
            constexpr int dyi = decltype(DY)::value, dy = dyi - 2, b = dyi & 1;
            if constexpr (dyi + 1 < 5) {
                issue_row(std::integral_constant<int, dyi + 1>{}, std::integral_constant<int, (dyi + 1) & 1>{});
                wait_vm<5 * MT>(xr[b][0][0]);                       // everything but the row just requested has landed
            } else {
                wait_vm<0>(xr[b][0][0]);
            }
            static_for<0, 5 * MT>([&](auto I) { touch(xr[b][decltype(I)::value / MT][decltype(I)::value % MT]); });
            const bool rowok = (unsigned)(y0 + dy) < (unsigned)H;
            f32x4 t[MT][2];
            static_for<0, MT>([&](auto N) { t[decltype(N)::value][0] = t[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
            static_for<0, 5>([&](auto DX) {
                constexpr int dx = decltype(DX)::value - 2;
                const bool ok = rowok && (unsigned)(x0 + dx) < (unsigned)W;
                float4 xv[MT];
                static_for<0, MT>([&](auto G) { xv[decltype(G)::value] = ok ? as_float4(xr[b][decltype(DX)::value][decltype(G)::value]) : zero4; });
                gemm_acc<MT, MT>(w2 + decltype(DX)::value * (MT * MT * 64), xv, t);
            });
            float4 tv[MT];
            static_for<0, MT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                tv[n] = make_float4(t[n][0][0] + t[n][1][0], t[n][0][1] + t[n][1][1], t[n][0][2] + t[n][1][2], t[n][0][3] + t[n][1][3]);
            });
            gemm_acc<MT, MT>(w2b + dyi * (MT * MT * 64), tv, acc);
        });
    }
    float4 y2[MT];
    static_for<0, MT>([&](auto N) {
        constexpr int n = decltype(N)::value;
        const float4 b = pb2[n], s = ps2[n];
        y2[n] = prelu4(make_float4(acc[n][0][0] + acc[n][1][0] + b.x, acc[n][0][1] + acc[n][1][1] + b.y, acc[n][0][2] + acc[n][1][2] + b.z,
                                   acc[n][0][3] + acc[n][1][3] + b.w), s);
    });

    // ---- conv3 + skip + PReLU (weights from LDS, skip rows requested at the top of the kernel)
    const float4 *w3 = l3 + lane;
    float4 w3v[CT][MT];
    static_for<0, CT>([&](auto N) { static_for<0, MT>([&](auto G) { w3v[decltype(N)::value][decltype(G)::value] = w3[(decltype(N)::value * MT + decltype(G)::value) * 64]; }); });
    f32x4 o[CT];
    static_for<0, CT>([&](auto N) { o[decltype(N)::value] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
    static_for<0, MT>([&](auto G) {
        constexpr int g = decltype(G)::value;
        static_for<0, 4>([&](auto R) {
            constexpr int r = decltype(R)::value;
            static_for<0, CT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp<r>(w3v[n][g]), comp<r>(y2[g]), o[n], 0, 0, 0);
            });
        });
    });
    float4 ov[CT];
    static_for<0, CT>([&](auto N) {
        constexpr int n = decltype(N)::value;
        const float4 b = pb3[n], s = ps3[n];
        ov[n] = prelu4(make_float4((o[n][0] + b.x) + skip[n].x, (o[n][1] + b.y) + skip[n].y, (o[n][2] + b.z) + skip[n].z, (o[n][3] + b.w) + skip[n].w), s);
    });
    if (live) {
        if (a.nchw) {
            const int v = pc / (W * H);
            float *dst = a.out + ((size_t)v * C * H + y0) * W + x0;
            static_for<0, CT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                float *q = dst + (size_t)(16 * n + 4 * kq) * H * W;
                q[0] = ov[n].x; q[(size_t)H * W] = ov[n].y; q[(size_t)2 * H * W] = ov[n].z; q[(size_t)3 * H * W] = ov[n].w;
            });
        } else {
            static_for<0, CT>([&](auto N) { *reinterpret_cast<float4 *>(a.out + (size_t)pc * C + 16 * decltype(N)::value + 4 * kq) = ov[decltype(N)::value]; });
        }
    }

    // ---- the next bottleneck's conv1 (1x1, C -> MIDN) + PReLU on the block output while it is in registers
    if constexpr (NT > 0) {
        const float4 *w1 = l1 + lane;
        float4 w1v[NT][CT];
        static_for<0, NT>([&](auto N) { static_for<0, CT>([&](auto G) { w1v[decltype(N)::value][decltype(G)::value] = w1[(decltype(N)::value * CT + decltype(G)::value) * 64]; }); });
        f32x4 n1[NT][2];
        static_for<0, NT>([&](auto N) { n1[decltype(N)::value][0] = n1[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
        static_for<0, CT>([&](auto G) {
            constexpr int g = decltype(G)::value;
            static_for<0, 4>([&](auto R) {
                constexpr int r = decltype(R)::value;
                static_for<0, NT>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    n1[n][r & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp<r>(w1v[n][g]), comp<r>(ov[g]), n1[n][r & 1], 0, 0, 0);
                });
            });
        });
        if (live) {
            static_for<0, NT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                const float4 b = pb1[n], s = ps1[n];
                *reinterpret_cast<float4 *>(a.y1n + (size_t)pc * MIDN + 16 * n + 4 * kq) =
                    prelu4(make_float4(n1[n][0][0] + n1[n][1][0] + b.x, n1[n][0][1] + n1[n][1][1] + b.y, n1[n][0][2] + n1[n][1][2] + b.z,
                                       n1[n][0][3] + n1[n][1][3] + b.w), s);
            });
        }
    }
}

// conv1 of a down block: Conv2d(cin, mid, 2, stride 2) (+ folded BatchNorm) + PReLU; x rows of CIN floats at 2H x 2W -> y1 rows of
// MID floats at H x W.  Also serves a plain 1x1 reduction (taps == 1, x at H x W): the first block after a down block's K2 is fused
// into that launch, this entry is for callers that start a chain elsewhere.
template <int CIN, int MID>
__global__ __launch_bounds__(64) void enet_conv1_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                        const float *__restrict__ s, float *__restrict__ y1, int H, int W, int npix, int taps)
{
    constexpr int MT = MID / 16, KG = CIN / 16;
    const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x * 16 + li;
    const bool live = p < npix;
    const int pc = live ? p : npix - 1;
    const int x0 = pc % W, y0 = (pc / W) % H, v = pc / (W * H);
    f32x4 acc[MT][2];
    static_for<0, MT>([&](auto N) { acc[decltype(N)::value][0] = acc[decltype(N)::value][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
    const float4 *wp = reinterpret_cast<const float4 *>(w) + lane;
    if (taps == 4) {
        const float *s00 = x + ((size_t)(v * 2 * H + 2 * y0) * (2 * W) + 2 * x0) * CIN + 4 * kq;
        static_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            const float *src = s00 + ((size_t)(t >> 1) * 2 * W + (t & 1)) * CIN;
            float4 xv[KG];
            static_for<0, KG>([&](auto G) { xv[decltype(G)::value] = ld4(src + 16 * decltype(G)::value); });
            gemm_acc<MT, KG>(wp + t * (MT * KG * 64), xv, acc);
        });
    } else {
        float4 xv[KG];
        static_for<0, KG>([&](auto G) { xv[decltype(G)::value] = ld4(x + (size_t)pc * CIN + 16 * decltype(G)::value + 4 * kq); });
        gemm_acc<MT, KG>(wp, xv, acc);
    }
    if (live) {
        static_for<0, MT>([&](auto N) {
            constexpr int n = decltype(N)::value;
            const float4 bb = ld4(b + 16 * n + 4 * kq), ss = ld4(s + 16 * n + 4 * kq);
            *reinterpret_cast<float4 *>(y1 + (size_t)pc * MID + 16 * n + 4 * kq) =
                prelu4(make_float4(acc[n][0][0] + acc[n][1][0] + bb.x, acc[n][0][1] + acc[n][1][1] + bb.y, acc[n][0][2] + acc[n][1][2] + bb.z,
                                   acc[n][0][3] + acc[n][1][3] + bb.w), ss);
        });
    }
}

// enet.py's initial block: cat(Conv2d(3, 13, 3, stride 2, padding 1)(x), MaxPool2d(2, 2)(x)) -> BatchNorm2d(16) -> PReLU(16), NCHW
// images (V,3,Hi,Wi) -> rows of 16 floats at Hi/2 x Wi/2.  w: the 13 filters with the BatchNorm scale folded in, [13][3][3][3];
// b: their folded bias; ps / ph: BatchNorm scale / shift of the three pooled channels; slope: the 16 PReLU slopes.
__global__ __launch_bounds__(256) void enet_initial_kernel(const float *__restrict__ img, const float *__restrict__ w, const float *__restrict__ b,
                                                           const float *__restrict__ ps, const float *__restrict__ ph, const float *__restrict__ slope,
                                                           float *__restrict__ out, int V, int Hi, int Wi)
{
    __shared__ float sw[13 * 27 + 13 + 6 + 16];
    for (int i = threadIdx.x; i < 13 * 27 + 13 + 6 + 16; i += 256) {
        sw[i] = i < 351 ? w[i] : i < 364 ? b[i - 351] : i < 367 ? ps[i - 364] : i < 370 ? ph[i - 367] : slope[i - 370];
    }
    __syncthreads();
    const int H = Hi / 2, W = Wi / 2;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= V * H * W) return;
    const int x0 = p % W, y0 = (p / W) % H, v = p / (W * H);
    float patch[3][3][3], pool[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *pl = img + ((size_t)v * 3 + c) * Hi * Wi;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = 2 * y0 - 1 + ky, xx = 2 * x0 - 1 + kx;
                patch[c][ky][kx] = ((unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi) ? pl[(size_t)yy * Wi + xx] : 0.f;
            }
        }
        // pool window = rows 2y, 2y+1 / cols 2x, 2x+1 = patch entries [1..2][1..2] (always inside the image: Hi, Wi even)
        pool[c] = fmaxf(fmaxf(patch[c][1][1], patch[c][1][2]), fmaxf(patch[c][2][1], patch[c][2][2]));
    }
    float o[16];
#pragma unroll
    for (int f = 0; f < 13; ++f) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = fmaf(sw[f * 27 + c * 9 + ky * 3 + kx], patch[c][ky][kx], acc);
        o[f] = acc + sw[351 + f];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[13 + c] = pool[c] * sw[364 + c] + sw[367 + c];
#pragma unroll
    for (int f = 0; f < 16; ++f) o[f] = prelu1(o[f], sw[370 + f]);
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)p * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

template <int C, int MID, int MIDN, bool ASYM>
int launch_block_k(const EnetBlockArgs &a, hipStream_t st)
{
    static const int waves = [] { const char *e = getenv("SIS3D_ENET_WAVES"); return e && atoi(e) == 1 ? 1 : 4; }();      // A/B switch
    // 32-bit byte offsets in the inline-asm row loads (ybase + tap offsets): refuse maps whose conv1 rows do not fit
    if ((int64_t)a.npix * MID * 4 + (int64_t)16 * 4 * MID * (a.W + 1) * (a.kind == 0 ? a.dil : 2) >= 0x7fffffffLL) return SIS3D_EUNSUPPORTED;
    if (waves == 1) {
        hipLaunchKernelGGL((enet_block_kernel<C, MID, MIDN, 1, ASYM>), dim3((unsigned)cdiv(a.npix, 16)), dim3(64), 0, st, a);
        return sis3d_check_launch();
    }
    constexpr size_t lds = (size_t)enet_lds_float4<C, MID, MIDN>(ASYM ? 10 : 9) * sizeof(float4);
    auto kern = enet_block_kernel<C, MID, MIDN, 4, ASYM>;
    static Sis3dLdsOnce once;
    if (lds > 64 * 1024 && sis3d_grant_lds(once, (const void *)kern, (int)lds) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(cdiv(a.npix, 16), 4)), dim3(256), lds, st, a);
    return sis3d_check_launch();
}

template <int C, int MID, int MIDN>
int launch_block(const EnetBlockArgs &a, hipStream_t st)
{
    return a.kind == 0 ? launch_block_k<C, MID, MIDN, false>(a, st) : launch_block_k<C, MID, MIDN, true>(a, st);
}

} // namespace

extern "C" int sis3d_enet_initial(const float *images, int V, int Hi, int Wi, const float *w, const float *b, const float *pool_scale,
                                  const float *pool_shift, const float *slope, float *out, sis3d_stream_t stream)
{
    if (!images || !w || !b || !pool_scale || !pool_shift || !slope || !out || V <= 0 || Hi <= 0 || Wi <= 0 || (Hi & 1) || (Wi & 1)) return SIS3D_EINVAL;
    const int64_t n = (int64_t)V * (Hi / 2) * (Wi / 2);
    if (n > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    hipLaunchKernelGGL(enet_initial_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), images, w, b, pool_scale, pool_shift, slope,
                       out, V, Hi, Wi);
    return sis3d_check_launch();
}

extern "C" int sis3d_enet_conv1(const float *x, int V, int H, int W, int cin, int mid, int taps, const float *w, const float *b, const float *slope,
                                float *y1, sis3d_stream_t stream)
{
    if (!x || !w || !b || !slope || !y1 || V <= 0 || H <= 0 || W <= 0 || (taps != 1 && taps != 4)) return SIS3D_EINVAL;
    const int64_t n = (int64_t)V * H * W;
    if (n > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)cdiv(n, 16));
    if (cin == 16 && mid == 16) hipLaunchKernelGGL((enet_conv1_kernel<16, 16>), grid, dim3(64), 0, st, x, w, b, slope, y1, H, W, (int)n, taps);
    else if (cin == 64 && mid == 16) hipLaunchKernelGGL((enet_conv1_kernel<64, 16>), grid, dim3(64), 0, st, x, w, b, slope, y1, H, W, (int)n, taps);
    else if (cin == 64 && mid == 32) hipLaunchKernelGGL((enet_conv1_kernel<64, 32>), grid, dim3(64), 0, st, x, w, b, slope, y1, H, W, (int)n, taps);
    else if (cin == 128 && mid == 32) hipLaunchKernelGGL((enet_conv1_kernel<128, 32>), grid, dim3(64), 0, st, x, w, b, slope, y1, H, W, (int)n, taps);
    else return SIS3D_EUNSUPPORTED;
    return sis3d_check_launch();
}

extern "C" int sis3d_enet_block(const float *x, const float *y1, int V, int H, int W, int c, int mid, int kind, int dil, const float *w2,
                                const float *b2, const float *s2, const float *w2b, const float *w3, const float *b3, const float *s3, int pool_cin,
                                float *out, int out_nchw, const float *w1n, const float *b1n, const float *s1n, int midn, float *y1n,
                                sis3d_stream_t stream)
{
    if (!x || !y1 || !w2 || !b2 || !s2 || !w3 || !b3 || !s3 || !out || V <= 0 || H <= 0 || W <= 0) return SIS3D_EINVAL;
    if ((kind != 0 && kind != 1) || (kind == 0 && dil < 1) || (kind == 1 && !w2b)) return SIS3D_EINVAL;
    if (midn < 0 || (midn > 0 && (!w1n || !b1n || !s1n || !y1n))) return SIS3D_EINVAL;
    if (pool_cin < 0 || pool_cin > c || (pool_cin % 16)) return SIS3D_EINVAL;
    const int64_t n = (int64_t)V * H * W;
    if (n > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    EnetBlockArgs a;
    a.x = x; a.y1 = y1; a.w2 = w2; a.b2 = b2; a.s2 = s2; a.w2b = w2b; a.w3 = w3; a.b3 = b3; a.s3 = s3; a.out = out;
    a.w1n = w1n; a.b1n = b1n; a.s1n = s1n; a.y1n = y1n;
    a.V = V; a.H = H; a.W = W; a.npix = (int)n; a.kind = kind; a.dil = dil; a.pool_cin = pool_cin; a.nchw = out_nchw;
    hipStream_t st = as_stream(stream);
    if (c == 64 && mid == 16 && midn == 16) return launch_block<64, 16, 16>(a, st);
    if (c == 64 && mid == 16 && midn == 0) return launch_block<64, 16, 0>(a, st);
    if (c == 128 && mid == 32 && midn == 32) return launch_block<128, 32, 32>(a, st);
    if (c == 128 && mid == 32 && midn == 0) return launch_block<128, 32, 0>(a, st);
    return SIS3D_EUNSUPPORTED;
}
