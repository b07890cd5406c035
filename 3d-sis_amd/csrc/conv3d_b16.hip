// k3 / pad-1 3D convolution on the bf16 matrix pipe with SPLIT operands ("bf16x3"): an OPTIONAL, separately reported variant
// of conv3d_k3t16_kernel for the rpn_net layers (lib/nets/network.py:40,45: nn.Conv3d(128, 256, 3, padding=1) + ReLU).
//
// gfx950 has no xf32 / TF32 mode and its fp32 MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s), so the exact-fp32 conv
// stack is bound by the fp32 matrix pipe.  Here every fp32 operand x is split into two bf16 numbers
//     x = hi + lo (+ r),  hi = bf16(x),  lo = bf16(x - hi),  |r| <= 2^-17 |x|
// and a product a * b is taken as  ah*bh + ah*bl + al*bh  on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: three matrix
// instructions of 16 cycles replace eight fp32 instructions of 32 cycles for the same 32 input channels.  The dropped term
// al*bl and the split remainders bound the relative error of a product by ~2^-16 (1.5e-5); results are NOT bit-compatible with
// the fp32 path and the variant is therefore off by default (tests/test_gpu_conv_b16.py states the measured error; the headline
// bench line stays on the fp32 kernels).
//
// Structure (as conv3d_t16.hip unless noted): workgroup = 4 waves = a BX x BY x BZ brick x one 16-wide cout tile; the halo brick
// of one 32-channel chunk is staged in LDS as rows of [32 hi | 32 lo] bf16 (128 B + 16 B pad, the footprint of the fp32 image),
// converted on the way from the fp32 activations (v_cvt_pk_bf16_f32); a matrix instruction consumes all 32 channels of a chunk,
// so the four waves split the 27 TAPS (7 / 7 / 7 / 6) instead of the channels, and
// their partial tiles are summed through LDS at the end.  A tap is an immediate offset of two ds_read_b128 (hi, lo) that feed
// three MFMAs; weights are pre-split into fragment order ([tile][chunk][tap][hi|lo][lane][8]) and a chunk's seven taps are
// requested one chunk ahead.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int B16_MAXP = 4;
constexpr int CK = 32;                  // channels per chunk = K of one matrix instruction
constexpr int RSB = 2 * CK * 2 + 32;    // LDS row stride in BYTES: 64 B hi + 64 B lo + 32 B pad (160: ds_read_b128 of a tile's 16 rows takes
                                        // 8 LDS cycles on the real halo-brick rows; 144 would take 11.7, tools note in profiles/)
constexpr int TAPS = 27;
constexpr int TPW = 7;                  // taps per wave and chunk (the wave of rank 3 takes 6)

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ragged batch descriptor (the mask head: all detected boxes' crops in one launch); layout shared with conv3d_t16.hip
struct B16Ragged {
    int X, Y, Z;
    int nbx, nby, nbz;
    int block0;
    int pad;
    int64_t in_off, out_off;
};

struct B16Args {
    const float *in[B16_MAXP];
    const uint4 *wp[B16_MAXP];          // packed split weights
    const float *bias[B16_MAXP];
    float *out[B16_MAXP];
    int X, Y, Z;
    int cin_stride;
    int cout, ntiles, nq, flags;
    int out_stride, out_coff;
    int nbx, nby, nbz;
    const B16Ragged *rag;
    int nrag;
    long long *dbg;                     // tools/b16_phases.py: wall_clock64() of every wave at 16 phase boundaries, or NULL
    int dbg_cap;                        // workgroups the buffer holds
};

// fp32 pair -> packed bf16 pair (round to nearest even) and the bf16 pair of the remainders
__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    hi = *reinterpret_cast<const uint32_t *>(&h);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)ra, (__bf16)rb};
    lo = *reinterpret_cast<const uint32_t *>(&l);
}

// NTC = cout tiles per workgroup: with 2, an A fragment read from LDS feeds six matrix instructions instead of three -- the
// kernel with one tile per workgroup is bound by its LDS reads (two ds_read_b128 per three 16-cycle MFMAs on four waves)
// CLIP (ragged mask-head launches, as conv3d_k3t16_kernel): a brick that sticks out of its crop enumerates only the voxels inside
// and runs only the tile groups it has voxels for (one uniform branch per chunk picks the loop body).
template <int BX, int BY, int BZ, int NTC, bool CLIP = false>
__global__ __launch_bounds__(256, 1) void conv3d_k3b16_kernel(const B16Args a)
{
    constexpr int M = BX * BY * BZ, MT = (M + 15) / 16;
    constexpr int IBY = BY + 2, IBZ = BZ + 2, IBX = BX + 2, ROWS = IBX * IBY * IBZ;
    constexpr int ITEMS = ROWS * (CK / 4), NIT = (ITEMS + 255) / 256;
    constexpr int G = (MT % 3 == 0) ? 3 : (MT >= 4 ? 4 : MT);
    constexpr int NG = (MT + G - 1) / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];   // [ROWS][RSB]; reused as float [4 waves][MT][16][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    auto stamp = [&](int k) {
        const int fb = (int)(blockIdx.x + gridDim.x * blockIdx.y);
        if (a.dbg && lane == 0 && fb < a.dbg_cap) a.dbg[((size_t)fb * 4 + wave) * 16 + k] = (long long)wall_clock64();
    };
    stamp(0);
    int wid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int prob = blockIdx.y;
    const float *__restrict__ p_in = a.in[prob];
    int gX = a.X, gY = a.Y, gZ = a.Z, nby = a.nby, nbz = a.nbz;
    int64_t out_off = 0;
    if (a.nrag > 0) {
        int lo = 0, hi = a.nrag - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.rag[mid].block0 <= wid) lo = mid; else hi = mid - 1;
        }
        const B16Ragged d = a.rag[lo];
        wid -= d.block0;
        gX = d.X; gY = d.Y; gZ = d.Z; nby = d.nby; nbz = d.nbz;
        p_in += d.in_off;
        out_off = d.out_off;
    }
    const int ngrp = (a.ntiles + NTC - 1) / NTC;            // workgroups per brick
    const int nt = (wid % ngrp) * NTC;                      // first cout tile of this workgroup
    const int brick = wid / ngrp;
    const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
    const int ox0 = bx * BX, oy0 = by * BY, oz0 = bz * BZ;
    int cy = BY, cz = BZ, m_act = M, mt_act = MT;
    uint32_t inv_yz = 0, inv_z = 0;
    if constexpr (CLIP) {
        const int cx = min(BX, gX - ox0);
        cy = min(BY, gY - oy0);
        cz = min(BZ, gZ - oz0);
        m_act = cx * cy * cz;
        mt_act = __builtin_amdgcn_readfirstlane((m_act + 15) >> 4);
        inv_yz = (65536u + (uint32_t)(cy * cz) - 1u) / (uint32_t)(cy * cz);       // m / d by multiply-shift: exact for m < 512, d <= 144
        inv_z = (65536u + (uint32_t)cz - 1u) / (uint32_t)cz;
    }
    auto voxel_of = [&](int m, int &lx, int &ly, int &lz) {
        if constexpr (CLIP) {
            lx = (int)(((uint32_t)m * inv_yz) >> 16);
            const int rem = m - lx * cy * cz;
            ly = (int)(((uint32_t)rem * inv_z) >> 16);
            lz = rem - ly * cz;
        } else {
            lx = m / (BY * BZ); ly = (m / BZ) % BY; lz = m % BZ;
        }
    };

    // halo staging table (element offsets, -1 = outside the grid), as conv3d_k3t16_kernel
    int goff[NIT];
    {
        const int row0 = tid >> 3, c4 = tid & 7;
        int hz = row0 % IBZ, hy = (row0 / IBZ) % IBY, hx = row0 / (IBZ * IBY);
        constexpr int DZ = 32 % IBZ, DY = (32 / IBZ) % IBY, DX = 32 / (IBZ * IBY);
        static_for<0, NIT>([&](auto I) {
            constexpr int it = decltype(I)::value;
            const int gx = ox0 - 1 + hx, gy = oy0 - 1 + hy, gz = oz0 - 1 + hz;
            const bool ok = (tid + it * 256 < ITEMS) && (unsigned)gx < (unsigned)gX && (unsigned)gy < (unsigned)gY && (unsigned)gz < (unsigned)gZ;
            goff[it] = ok ? ((gx * gY + gy) * gZ + gz) * a.cin_stride + c4 * 4 : -1;
            hz += DZ;
            const int cz = hz >= IBZ;
            hz -= cz * IBZ;
            hy += DY + cz;
            const int cy = hy >= IBY;
            hy -= cy * IBY;
            hx += DX + cy;
        });
    }
    float4 sv[NIT];
    auto stage_load = [&](int q) {
        static_for<0, NIT>([&](auto I) {
            constexpr int it = decltype(I)::value;
            const int o = goff[it];
            const float4 v = *reinterpret_cast<const float4 *>(p_in + (size_t)(o < 0 ? 0 : o) + q * CK);
            sv[it] = o < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : v;
        });
    };
    // item = (row, 4 channels): 4 hi bf16 (8 B) into the hi half of the row, 4 lo bf16 into the lo half
    auto stage_store = [&]() {
        static_for<0, NIT>([&](auto I) {
            constexpr int it = decltype(I)::value;
            const int idx = tid + it * 256;
            if (idx < ITEMS) {
                uint2 h, l;
                split2(sv[it].x, sv[it].y, h.x, l.x);
                split2(sv[it].z, sv[it].w, h.y, l.y);
                unsigned char *row = ldsb + (idx >> 3) * RSB + (idx & 7) * 8;
                *reinterpret_cast<uint2 *>(row) = h;
                *reinterpret_cast<uint2 *>(row + 2 * CK) = l;
            }
        });
    };

    // B operand: wave w owns taps [TPW w, TPW w + 7) (wave 3: six taps -- a fixed 27/28 balance; rotating the short share with
    // the chunk made every refill address a run-time select and the compiler serialised those loads, see profiles/r02_split_bf16.md).
    // hi and lo fragment of every (tap, cout tile) = 16 B per lane each, packed [ntile][chunk][tap][2][64] uint4.  All of a chunk's
    // slots sit in registers; slot ts is refilled for the NEXT chunk right after its last matrix instruction (the last chunk
    // re-requests its own fragments: no branch around the loads).
    constexpr int TS = 2 * 64, QS = TAPS * TS;              // uint4 strides of a tap / a chunk
    const uint4 *bsrc[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) {
        const int tile = (nt + n < a.ntiles) ? nt + n : a.ntiles - 1;          // a surplus tile recomputes the last one, never stored
        const int tap0 = TPW * wave;
        bsrc[n] = a.wp[prob] + lane + (size_t)tile * a.nq * QS + (size_t)tap0 * TS;
    }
    uint4 bw[TPW][NTC][2];
    auto load_slot = [&](int q, auto T) {
        constexpr int t = decltype(T)::value;
        // wave 3's seventh slot (tap 27 does not exist) re-reads its sixth and is never used
        const int tt = (wave == 3 && t == TPW - 1) ? t - 1 : t;
        static_for<0, NTC>([&](auto N) {
            constexpr int n = decltype(N)::value;
            const uint4 *src = bsrc[n] + (size_t)q * QS + (size_t)tt * TS;
            bw[t][n][0] = src[0];
            bw[t][n][1] = src[64];
        });
    };

    f32x4 acc[MT][NTC];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NTC; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage_load(0);                                          // first: the LDS stores below wait for these only,
    static_for<0, TPW>([&](auto T) { load_slot(0, T); });   // the weight fragments keep arriving under the first taps
    __builtin_amdgcn_sched_barrier(0);
    int abase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = 16 * t + li;
        m = m < m_act ? m : m_act - 1;
        int lx, ly, lz;
        voxel_of(m, lx, ly, lz);
        abase[t] = ((lx * IBY + ly) * IBZ + lz) * RSB + kq * 16;          // bytes: channels 8 kq .. 8 kq + 7 of the hi half
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_store();
    stamp(1);
    __syncthreads();
    stamp(2);

    const int nq = a.nq;
    const int nga = CLIP ? (mt_act + G - 1) / G : NG;
    for (int q = 0; q < nq; ++q) {
        const bool more = q + 1 < nq;
        if (more) stage_load(q + 1);
        const int r = wave;
        const int qn = more ? q + 1 : q;                   // chunk whose fragments the slots are refilled with
        // taps [TPW R, TPW R + NTAP) x NG groups of G tiles; per group 2 G reads (hi, lo) and 3 G NTC matrix instructions
        auto run_taps = [&](auto R_, auto NGA_) {
            constexpr int R = decltype(R_)::value, NGA = decltype(NGA_)::value;      // NGA: tile groups this brick has (NG unless CLIP)
            constexpr int T0 = TPW * R, NTAP = (T0 + TPW <= TAPS) ? TPW : TAPS - T0;
            constexpr int NSTEP = NTAP * NGA;
            bf16x8 ah[2][G], al[2][G];
            auto read_group = [&](auto BUF, auto STEP) {
                constexpr int buf = decltype(BUF)::value, step = decltype(STEP)::value;
                constexpr int tap = T0 + step / NGA, g = step % NGA;
                constexpr int dz = tap % 3, dy = (tap / 3) % 3, dx = tap / 9;
                constexpr int toff = ((dx * IBY + dy) * IBZ + dz) * RSB;
                static_for<0, G>([&](auto J) {
                    constexpr int j = decltype(J)::value, t = g * G + j;
                    if constexpr (t < MT) {
                        ah[buf][j] = *reinterpret_cast<const bf16x8 *>(ldsb + abase[t] + toff);
                        al[buf][j] = *reinterpret_cast<const bf16x8 *>(ldsb + abase[t] + toff + 2 * CK);
                    }
                });
            };
            read_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, NSTEP>([&](auto STEP) {
                constexpr int step = decltype(STEP)::value;
                constexpr int ts = step / NGA, g = step % NGA;
                if constexpr (step + 1 < NSTEP) read_group(std::integral_constant<int, (step + 1) & 1>{}, std::integral_constant<int, step + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                // small terms first, the hi * hi product last; the instructions of one accumulator are G NTC instructions apart
                static_for<0, 3>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    static_for<0, G>([&](auto J) {
                        constexpr int j = decltype(J)::value, t = g * G + j;
                        if constexpr (t < MT)
                            static_for<0, NTC>([&](auto N) {
                                constexpr int n = decltype(N)::value;
                                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(p == 0 ? &al[step & 1][j] : &ah[step & 1][j]);
                                const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(&bw[ts][n][p == 1 ? 1 : 0]);
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[t][n], 0, 0, 0);
                            });
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g == NGA - 1) load_slot(qn, std::integral_constant<int, ts>{});     // this slot's last use is behind us
            });
        };
        static_for<0, 4>([&](auto R_) {
            if (r == decltype(R_)::value) {
                if constexpr (CLIP) {
                    static_for<1, NG + 1>([&](auto NGA_) {
                        if (nga == decltype(NGA_)::value) run_taps(R_, NGA_);
                    });
                } else {
                    run_taps(R_, std::integral_constant<int, NG>{});
                }
            }
        });
        if (q < 4) stamp(3 + 3 * q);                       // matrix loop of chunk q done
        __syncthreads();                                   // every wave is done with chunk q's image
        if (q < 4) stamp(4 + 3 * q);
        if (more) {
            stage_store();
            __syncthreads();
        }
        if (q < 4) stamp(5 + 3 * q);
    }

    // ---- cross-wave reduction + epilogue, as conv3d_k3t16_kernel: tile (t, n) of wave w at [w][t][n][16 voxels][16 couts]
    float *lds = reinterpret_cast<float *>(ldsb);
    constexpr int SLAB = MT * NTC * 256;
    float *red = lds + (size_t)wave * SLAB;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NTC; ++n)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(t * NTC + n) * 256 + (4 * kq + rr) * 16 + li] = acc[t][n][rr];
    __syncthreads();
    const int row = lane >> 2, c4 = lane & 3;
    float *__restrict__ p_out = a.out[prob] + out_off;
    for (int t = wave; t < mt_act; t += 4) {
        const int m = 16 * t + row;
        int lx, ly, lz;
        voxel_of(m < m_act ? m : 0, lx, ly, lz);
        const int ox = ox0 + lx, oy = oy0 + ly, oz = oz0 + lz;
        const bool inside = m < m_act && ox < gX && oy < gY && oz < gZ;
#pragma unroll
        for (int n = 0; n < NTC; ++n) {
            const int co = 16 * (nt + n) + 4 * c4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias[prob] && co < a.cout) bv = *reinterpret_cast<const float4 *>(a.bias[prob] + co);
            const float4 *src = reinterpret_cast<const float4 *>(lds + (t * NTC + n) * 256 + row * 16 + c4 * 4);
            const float4 s0 = src[0], s1 = src[SLAB / 4], s2 = src[2 * (SLAB / 4)], s3 = src[3 * (SLAB / 4)];
            float4 v;
            v.x = (s0.x + s1.x) + (s2.x + s3.x) + bv.x;
            v.y = (s0.y + s1.y) + (s2.y + s3.y) + bv.y;
            v.z = (s0.z + s1.z) + (s2.z + s3.z) + bv.z;
            v.w = (s0.w + s1.w) + (s2.w + s3.w) + bv.w;
            if (a.flags & SIS3D_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (inside && nt + n < a.ntiles && co < a.cout)
                *reinterpret_cast<float4 *>(p_out + ((size_t)(ox * gY + oy) * gZ + oz) * a.out_stride + a.out_coff + co) = v;
        }
    }
    stamp(15);
}

// (Cout,Cin,3,3,3) fp32 -> [cout/16][cin/32][tap 27][hi|lo][lane 64][8] bf16: lane (j = lane & 15, kb = lane >> 4), element e
// holds the split of W[16 tile + j][32 q + 8 kb + e][tap]
__global__ __launch_bounds__(256) void pack_weight_b16_kernel(const float *__restrict__ w, int cout, int cin, int ntiles, int nq,
                                                              uint16_t *__restrict__ packed)
{
    const int64_t total = (int64_t)ntiles * nq * TAPS * 64 * 8;           // (tile, q, tap, lane, e)
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        int64_t rest = idx >> 9;
        const int tap = (int)(rest % TAPS);
        rest /= TAPS;
        const int q = (int)(rest % nq), tile = (int)(rest / nq);
        const int co = tile * 16 + (lane & 15), ci = q * CK + 8 * (lane >> 4) + e;
        const float v = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * TAPS + tap] : 0.0f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        const int64_t base = (((int64_t)(tile * nq + q) * TAPS + tap) * 2) * 512 + lane * 8 + e;
        packed[base] = *reinterpret_cast<const uint16_t *>(&h);
        packed[base + 512] = *reinterpret_cast<const uint16_t *>(&l);
    }
}

std::atomic<long long *> g_b16_dbg{nullptr};   // sis3d_conv3d_k3b16_set_trace
std::atomic<int> g_b16_dbg_cap{0};

template <int BX, int BY, int BZ, int NTC, bool CLIP = false>
int launch_b16(B16Args &a, int nprob, hipStream_t st, int64_t ragged_blocks = 0)
{
    a.dbg = g_b16_dbg.load(std::memory_order_relaxed);
    a.dbg_cap = g_b16_dbg_cap.load(std::memory_order_relaxed);
    constexpr int M = BX * BY * BZ, MT = (M + 15) / 16;
    constexpr int ROWS = (BX + 2) * (BY + 2) * (BZ + 2);
    constexpr size_t img = (size_t)ROWS * RSB, red = (size_t)4 * MT * NTC * 256 * sizeof(float);
    constexpr size_t lds = img > red ? img : red;
    static_assert(lds <= 160 * 1024, "LDS brick too large");
    a.nbx = cdiv(a.X, BX); a.nby = cdiv(a.Y, BY); a.nbz = cdiv(a.Z, BZ);
    auto kern = conv3d_k3b16_kernel<BX, BY, BZ, NTC, CLIP>;
    static Sis3dLdsOnce lds_once;                                   // once per device, not per launch
    if (lds > 64 * 1024 && sis3d_grant_lds(lds_once, (const void *)kern, (int)lds) != SIS3D_OK) return SIS3D_ELAUNCH;
    const int64_t nwg = ragged_blocks > 0 ? ragged_blocks : (int64_t)a.nbx * a.nby * a.nbz * ((a.ntiles + NTC - 1) / NTC);
    if (nwg > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)nprob), dim3(256), lds, st, a);
    return sis3d_check_launch();
}

} // namespace

extern "C" int sis3d_conv3d_k3b16_set_trace(void *buf, int capacity_blocks)
{
    if (buf && capacity_blocks <= 0) return SIS3D_EINVAL;
    g_b16_dbg_cap.store(buf ? capacity_blocks : 0, std::memory_order_relaxed);
    g_b16_dbg.store((long long *)buf, std::memory_order_relaxed);
    return SIS3D_OK;
}

extern "C" size_t sis3d_conv_k3b16_packed_floats(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || cin % CK) return 0;
    // 2 bf16 per weight = one float's worth of bytes
    return (size_t)((cout + 15) / 16) * (cin / CK) * TAPS * 64 * 8;
}

extern "C" int sis3d_conv_k3b16_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream)
{
    if (!w || !packed || cout <= 0 || cin <= 0 || cin % CK) return SIS3D_EINVAL;
    const int ntiles = (cout + 15) / 16, nq = cin / CK;
    const int64_t total = (int64_t)ntiles * nq * TAPS * 512;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weight_b16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), w, cout, cin,
                       ntiles, nq, (uint16_t *)packed);
    return sis3d_check_launch();
}

extern "C" int sis3d_conv3d_k3b16(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                  const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                                  int out_stride, int out_coff, int brick, sis3d_stream_t stream)
{
    if (nprob < 1 || nprob > B16_MAXP || !ins || !packed_ws || !outs) return SIS3D_EINVAL;
    if (X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || cin_stride < cin || (cin_stride % 4)) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || (out_stride % 4) || (out_coff % 4) || out_stride < out_coff + cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    if ((int64_t)X * Y * Z * cin_stride > 0x7fffffffLL) return SIS3D_EUNSUPPORTED;
    B16Args a;
    for (int p = 0; p < B16_MAXP; ++p) {
        const int s = p < nprob ? p : 0;
        if (!ins[s] || !packed_ws[s] || !outs[s]) return SIS3D_EINVAL;
        a.in[p] = ins[s]; a.wp[p] = (const uint4 *)packed_ws[s]; a.bias[p] = biases ? biases[s] : nullptr; a.out[p] = outs[s];
    }
    a.X = X; a.Y = Y; a.Z = Z; a.cin_stride = cin_stride; a.cout = cout; a.ntiles = (cout + 15) / 16; a.nq = cin / CK;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = out_coff;
    a.rag = nullptr; a.nrag = 0;
    hipStream_t st = as_stream(stream);
    if (brick < 0) {
        // two cout tiles per workgroup (an A fragment feeds six instructions) whenever that still fills the chip, on the
        // larger brick first; else one tile per workgroup (measured on the four layer shapes of a 96x48x96 chunk, tools/b16_time.py)
        const int64_t b666 = (int64_t)cdiv(X, 6) * cdiv(Y, 6) * cdiv(Z, 6), b366 = (int64_t)cdiv(X, 3) * cdiv(Y, 6) * cdiv(Z, 6);
        const int64_t pairs = (a.ntiles + 1) / 2, quads = (a.ntiles + 3) / 4;
        if (a.ntiles % 4 == 0 && b366 * quads >= 224) brick = 5;        // rpn_net: 35.3 us against 36.7 for 6x6x6 x 2 tiles
        else if (b666 * pairs * nprob >= 224) brick = 3;
        else if (b366 * pairs * nprob >= 224) brick = 4;
        else if (b666 * a.ntiles * nprob >= 224) brick = 1;
        else brick = 2;
    }
    switch (brick) {
    case 1: return launch_b16<6, 6, 6, 1>(a, nprob, st);
    case 2: return launch_b16<3, 6, 6, 1>(a, nprob, st);
    case 3: return launch_b16<6, 6, 6, 2>(a, nprob, st);
    case 4: return launch_b16<3, 6, 6, 2>(a, nprob, st);
    case 5: return launch_b16<3, 6, 6, 4>(a, nprob, st);
    default: return SIS3D_EINVAL;
    }
}

// ---- ragged batch (the mask head, lib/nets/network.py:303-317): the caller tiles every crop with the brick of `brick`
// (2: 3x6x6 x 1 cout tile per workgroup, 4: 3x6x6 x 2 tiles) and numbers the workgroups crop by crop
extern "C" int sis3d_ragged_tiling_k3b16(int cin, int cout, int brick, int *bx, int *by, int *bz, int *ngroups)
{
    if (!bx || !by || !bz || !ngroups) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || (brick != 2 && brick != 4)) return SIS3D_EUNSUPPORTED;
    *bx = 3; *by = 6; *bz = 6;
    const int ntiles = (cout + 15) / 16;
    *ngroups = brick == 4 ? (ntiles + 1) / 2 : ntiles;
    return SIS3D_OK;
}

extern "C" int sis3d_conv3d_k3b16_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                         int flags, float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks,
                                         int brick, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || !desc_dev || ndesc <= 0 || total_blocks <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || (cin_stride % 4) || cin_stride < cin || (out_stride % 4) || out_stride < cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    B16Args a;
    for (int p = 0; p < B16_MAXP; ++p) { a.in[p] = in; a.wp[p] = (const uint4 *)packed_w; a.bias[p] = bias; a.out[p] = out; }
    a.X = a.Y = a.Z = 1; a.cin_stride = cin_stride; a.cout = cout; a.ntiles = (cout + 15) / 16; a.nq = cin / CK;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = 0;
    a.rag = (const B16Ragged *)desc_dev; a.nrag = ndesc;
    hipStream_t st = as_stream(stream);
    switch (brick) {
    case 2: return launch_b16<3, 6, 6, 1, true>(a, 1, st, total_blocks);
    case 4: return launch_b16<3, 6, 6, 2, true>(a, 1, st, total_blocks);
    default: return SIS3D_EUNSUPPORTED;
    }
}
