// k3 / pad-1 3D convolution, "one workgroup per CU, one wave per SIMD" formulation for gfx950 (CDNA4).
//
// Replaces the cuDNN calls behind nn.Conv3d(C, C', 3, padding=1) + bias + ReLU of lib/nets/backbones.py:20-22,188-231
// (Bottleneck.conv2, geometry2[0]) and lib/nets/network.py:40 (rpn_net_level*), where conv3d.hip's workgroup-granular
// decomposition leaves the chip quantised: its 32x32 tiles x tap-thirds give 5.06 wave tasks per SIMD (-> 6) on the
// 24x12x24 grid, and with every load removed it still stops at 69 % of the fp32 MFMA roof (profiles/README.md).
//
// Here the layer is cut so that EVERY SIMD of the chip gets the same number of matrix instructions:
//   * implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD like the 32x32x2 form, but 16-wide tiles:
//     a 96x48x96 chunk's layers have 432 / 3456 voxel tiles of 16, i.e. 27 * 2^k -- they divide over 256 CUs);
//   * a workgroup = 4 waves = one per SIMD, owns a BX x BY x BZ brick of output voxels (MT = ceil(vox/16) accumulator
//     tiles PER WAVE) x ONE 16-wide cout tile; the four waves split the reduction by input channel (wave w takes
//     channels 8w..8w+7 of every 32-channel chunk) and are summed through LDS at the end (deterministic);
//     e.g. rpn_net 128->256 on 24x12x24: 16 bricks of 6x6x12 x 16 cout tiles = 256 workgroups, 27 tiles per wave,
//     5832 MFMAs on every SIMD = exactly 1/1024 of the layer;
//   * the halo brick of one 32-channel chunk is staged once in LDS (rows padded to 36 floats); a tap is a constant
//     byte offset folded into the ds_read_b64 instruction; one read feeds two MFMAs (K order permuted: lane group k
//     holds channels 2k, 2k+1 of the wave's eight);
//   * weights are repacked once so that a wave's B operand for (chunk, tap) is one coalesced 8 B/lane load, fetched a
//     few taps ahead through a small register ring;
//   * with one wave per SIMD nothing else hides latency, so the NEXT chunk's halo brick is already in flight into
//     registers (up to 28 x 16 B per lane) while the current chunk is being multiplied; the switch is barrier ->
//     ds_write -> barrier;
//   * MFMAs on one accumulator are 40 cycles apart in dependency but issue every 32: the two MFMAs of a read are
//     interleaved over a group of G tiles.
// Epilogue: + bias, ReLU, 16 B stores.  FMA contraction is irrelevant here (the MFMA is an fmaf chain; tolerance 1e-4).
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int T16_MAXP = 4;    // independent same-shape problems per launch (the two RPN levels, chunk pairs)
constexpr int CK = 32;         // channels per LDS chunk
constexpr int RS = CK + 4;     // padded LDS row stride (floats): 144 B, 16 consecutive rows hit 16 distinct 8 B bank pairs
constexpr int TAPS = 27;

// compile-time loop: the body sees its index as a constant, so register arrays are never indexed dynamically (hipcc
// declines `#pragma unroll` on the 100-250-step bodies below and would push the arrays to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct T16Ragged {
    int X, Y, Z;          // grid of this problem
    int nbx, nby, nbz;    // bricks per axis for the launched brick
    int block0;           // first workgroup of this problem in the launch
    int pad;
    int64_t in_off, out_off;   // element offsets of this problem's activations inside the packed in / out buffers
};

struct T16Args {
    const float *in[T16_MAXP];
    const float *wp[T16_MAXP];
    const float *bias[T16_MAXP];
    float *out[T16_MAXP];
    int X, Y, Z;
    int cin_stride;
    int cout, ntiles;          // ntiles = ceil(cout/16)
    int nq;                    // cin / 32
    int flags;
    int out_stride, out_coff;
    int nbx, nby, nbz;
    // ragged batch (the mask head: one launch for all detected boxes' crops): problems of different grid sizes packed back
    // to back; descriptor layout shared with conv3d.hip's ragged launch
    const T16Ragged *rag;
    int nrag;
    // tools/t16_trace.py: per-workgroup {start, end (wall_clock64 ticks), HW_ID} of wave 0, or NULL
    long long *dbg;
    int dbg_cap;
    int xcd_tg, xcd_bpb;       // work-list order per XCD range: xcd_bpb bricks x xcd_tg cout tiles (0: tile-fastest order)
};

// CLIP (the ragged mask-head launches): a brick that sticks out of its crop enumerates only the voxels inside -- its tiles
// are cx*cy*cz / 16, not BX*BY*BZ / 16 -- and the MFMAs of the tiles it does not have are skipped by wave-uniform branches.
// Crops are 9-20 voxels across, so with fixed 108-voxel bricks 30 % of all tile slots were padding (mask head 76 TF).
template <int BX, int BY, int BZ, int G, int RB, bool CLIP = false>
__global__ __launch_bounds__(256, 1) void conv3d_k3t16_kernel(const T16Args a)
{
    constexpr int M = BX * BY * BZ, MT = (M + 15) / 16;
    constexpr int IBY = BY + 2, IBZ = BZ + 2, IBX = BX + 2, ROWS = IBX * IBY * IBZ;
    constexpr int ITEMS = ROWS * (CK / 4), NIT = (ITEMS + 255) / 256;
    constexpr int NG = (MT + G - 1) / G;                   // tile groups per tap
    constexpr int NSTEP = TAPS * NG;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [ROWS][RS]; reused as [4 waves][MT][16][16] for the reduction

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    const int flat_block = (int)(blockIdx.x + gridDim.x * blockIdx.y);
    if (a.dbg && tid == 0 && flat_block < a.dbg_cap) {
        a.dbg[3 * (size_t)flat_block] = (long long)wall_clock64();
        a.dbg[3 * (size_t)flat_block + 2] = (long long)__builtin_amdgcn_s_getreg((15 << 11) | 4);      // HW_REG_HW_ID[15:0]
    }
    // block -> (brick, cout tile): work list is brick-major / tile-minor and every XCD (block b runs on XCD b % 8, private
    // 4 MiB L2) takes one contiguous range of it: the workgroups that share a halo brick share an L2
    int wid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int prob = blockIdx.y;
    const float *__restrict__ p_in = a.in[prob];
    const float *__restrict__ p_wp = a.wp[prob];
    int gX = a.X, gY = a.Y, gZ = a.Z, nby = a.nby, nbz = a.nbz;
    int64_t out_off = 0;
    if (a.nrag > 0) {
        // find this workgroup's problem (block0 ascending): uniform -> scalar loads
        int lo = 0, hi = a.nrag - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.rag[mid].block0 <= wid) lo = mid; else hi = mid - 1;
        }
        const T16Ragged d = a.rag[lo];
        wid -= d.block0;
        gX = d.X; gY = d.Y; gZ = d.Z; nby = d.nby; nbz = d.nbz;
        p_in += d.in_off;
        out_off = d.out_off;
    }
    // (brick, cout tile) of work item wid.  Default: tile fastest.  With xcd_tg > 0 (launch_t16: full-grid launches whose
    // workgroup count splits evenly over the 8 XCDs) an XCD's range is xcd_bpb bricks x xcd_tg tiles instead of few bricks x all
    // tiles: its L2 then holds 1/ (ntiles / xcd_tg) of the weights (rpn_net: 1.75 MB weights + 1.65 MB halo bricks per 4 MB
    // L2 instead of 3.5 + 0.8 MB)
    int nt, brick;
    if (a.xcd_tg > 0) {
        const int L = a.xcd_tg * a.xcd_bpb, r = wid / L, within = wid - r * L, ngroups = a.ntiles / a.xcd_tg;
        const int bb = r / ngroups, tgp = r - bb * ngroups;
        brick = bb * a.xcd_bpb + within / a.xcd_tg;
        nt = tgp * a.xcd_tg + within % a.xcd_tg;
    } else {
        nt = wid % a.ntiles;
        brick = wid / a.ntiles;
    }
    const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
    const int ox0 = bx * BX, oy0 = by * BY, oz0 = bz * BZ;
    // clipped brick extents (CLIP): voxel m of the brick is (m / (cy cz), (m / cz) % cy, m % cz); small divisions by
    // multiply-shift (exact for m < 512, divisor <= 144)
    int cy = BY, cz = BZ, m_act = M, mt_act = MT;
    uint32_t inv_yz = 0, inv_z = 0;
    if constexpr (CLIP) {
        const int cx = min(BX, gX - ox0);
        cy = min(BY, gY - oy0);
        cz = min(BZ, gZ - oz0);
        m_act = cx * cy * cz;
        mt_act = __builtin_amdgcn_readfirstlane((m_act + 15) >> 4);
        inv_yz = (65536u + (uint32_t)(cy * cz) - 1u) / (uint32_t)(cy * cz);
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

    // halo staging: item = (row, 16 B piece); global element offset (-1: outside the grid -> zero).  Rows advance by 32 per
    // item: (hx, hy, hz) is carried incrementally instead of re-dividing (this address math sits in front of the very first
    // load of a workgroup that has nothing else to overlap it with)
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
    auto stage_store = [&]() {
        static_for<0, NIT>([&](auto I) {
            constexpr int it = decltype(I)::value;
            const int idx = tid + it * 256;
            if (idx < ITEMS) *reinterpret_cast<float4 *>(lds + (idx >> 3) * RS + (idx & 7) * 4) = sv[it];
        });
    };
    // B operand: packed [ntile][chunk][wave][tap][lane][2]
    const float2 *bp = reinterpret_cast<const float2 *>(p_wp) + ((size_t)(nt * a.nq) * 4 + wave) * (TAPS * 64) + lane;
    constexpr int QSTRIDE = 4 * TAPS * 64;                 // float2 elements per chunk
    float2 bq[RB];
    auto load_b = [&](int q, int tap) { return bp[(size_t)q * QSTRIDE + tap * 64]; };

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: first weight fragments, then chunk 0 straight into LDS
#pragma unroll
    for (int d = 0; d < RB - 1; ++d) bq[d] = load_b(0, d);
    stage_load(0);
    __builtin_amdgcn_sched_barrier(0);                     // the A-operand address math below runs under the loads' latency
    // A operand addressing: lane (li, kq) of tile t reads voxel m = 16 t + li, channels 8 wave + 2 kq, +1
    int abase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = 16 * t + li;
        m = m < m_act ? m : m_act - 1;                     // surplus rows of the last tile recompute a valid voxel, never stored
        int lx, ly, lz;
        voxel_of(m, lx, ly, lz);
        abase[t] = (((lx * IBY + ly) * IBZ + lz) * RS + 8 * wave + 2 * kq) * 4;
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_store();
    __syncthreads();

    const int nq = a.nq;
    const int nga = CLIP ? (mt_act + G - 1) / G : NG;          // tile groups with at least one voxel tile (uniform)
    for (int q = 0; q < nq; ++q) {
        const bool more = q + 1 < nq;
        if (more) stage_load(q + 1);                       // in flight while this chunk is multiplied
        // one chunk = TAPS x NGA steps; NGA = tile groups this brick really has (NG unless CLIP): the body is instantiated
        // per NGA and picked by ONE uniform branch per chunk (a branch per step cost more than the skipped MFMAs saved)
        auto run_chunk = [&](auto NGA_) {
            constexpr int NGA = decltype(NGA_)::value;
            constexpr int NSTEPA = TAPS * NGA;
            f32x2 ar[2][G];
            auto read_group = [&](auto BUF, auto STEP) {
                constexpr int buf = decltype(BUF)::value, step = decltype(STEP)::value;
                constexpr int tap = step / NGA, g = step % NGA;
                constexpr int dz = tap % 3, dy = (tap / 3) % 3, dx = tap / 9;
                constexpr int toff = ((dx * IBY + dy) * IBZ + dz) * RS * 4;
                static_for<0, G>([&](auto J) {
                    constexpr int j = decltype(J)::value, t = g * G + j;
                    if constexpr (t < MT)
                        ar[buf][j] = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(lds) + abase[t] + toff);
                });
            };
            read_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, NSTEPA>([&](auto STEP) {
                constexpr int step = decltype(STEP)::value;
                constexpr int tap = step / NGA, g = step % NGA;
                if constexpr (g == 0) {
                    // fetch the fragment RB-1 taps ahead (wraps into the next chunk's first taps)
                    constexpr int tn = tap + RB - 1;
                    if constexpr (tn < TAPS) bq[tn % RB] = load_b(q, tn);
                    else if (more) bq[tn % RB] = load_b(q + 1, tn - TAPS);
                }
                if constexpr (step + 1 < NSTEPA) read_group(std::integral_constant<int, (step + 1) & 1>{}, std::integral_constant<int, step + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                const float2 b = bq[tap % RB];
                static_for<0, G>([&](auto J) {
                    constexpr int j = decltype(J)::value, t = g * G + j;
                    if constexpr (t < MT) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[step & 1][j].x, b.x, acc[t], 0, 0, 0);
                });
                static_for<0, G>([&](auto J) {
                    constexpr int j = decltype(J)::value, t = g * G + j;
                    if constexpr (t < MT) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[step & 1][j].y, b.y, acc[t], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if constexpr (CLIP) {
            static_for<1, NG + 1>([&](auto NGA_) {
                if (nga == decltype(NGA_)::value) run_chunk(NGA_);
            });
        } else {
            run_chunk(std::integral_constant<int, NG>{});
        }
        __syncthreads();                                   // every wave is done with chunk q's image
        if (more) {
            stage_store();
            __syncthreads();
        }
    }

    // ---- cross-wave reduction: every wave publishes its MT partial tiles as [16 voxels][16 couts] (D layout: column =
    // lane & 15, rows 4 (lane >> 4) + r), then wave w finishes tiles w, w+4, ...: 16 B per lane = 4 couts of one voxel
    float *red = lds + (size_t)wave * (MT * 256);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[t * 256 + (4 * kq + r) * 16 + li] = acc[t][r];
    __syncthreads();
    const int row = lane >> 2, c4 = lane & 3;
    const int co = 16 * nt + 4 * c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias[prob] && co < a.cout) bv = *reinterpret_cast<const float4 *>(a.bias[prob] + co);
    float *__restrict__ p_out = a.out[prob] + out_off;
    for (int t = wave; t < mt_act; t += 4) {
        const float4 *src = reinterpret_cast<const float4 *>(lds + t * 256 + row * 16 + c4 * 4);
        const float4 s0 = src[0], s1 = src[MT * 64], s2 = src[2 * MT * 64], s3 = src[3 * MT * 64];
        float4 v;
        v.x = (s0.x + s1.x) + (s2.x + s3.x) + bv.x;
        v.y = (s0.y + s1.y) + (s2.y + s3.y) + bv.y;
        v.z = (s0.z + s1.z) + (s2.z + s3.z) + bv.z;
        v.w = (s0.w + s1.w) + (s2.w + s3.w) + bv.w;
        if (a.flags & SIS3D_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int m = 16 * t + row;
        int lx, ly, lz;
        voxel_of(m < m_act ? m : 0, lx, ly, lz);
        const int ox = ox0 + lx, oy = oy0 + ly, oz = oz0 + lz;
        if (m < m_act && ox < gX && oy < gY && oz < gZ && co < a.cout)
            *reinterpret_cast<float4 *>(p_out + ((size_t)(ox * gY + oy) * gZ + oz) * a.out_stride + a.out_coff + co) = v;
    }
    if (a.dbg && tid == 0 && flat_block < a.dbg_cap) a.dbg[3 * (size_t)flat_block + 1] = (long long)wall_clock64();
}

// (Cout,Cin,3,3,3) -> [cout/16][cin/32][wave 4][tap 27][lane 64][2]: lane (j = lane & 15, k = lane >> 4) holds
// W[16 tile + j][32 q + 8 wave + 2 k + e][tap], e = 0, 1
__global__ __launch_bounds__(256) void pack_weight_t16_kernel(const float *__restrict__ w, int cout, int cin, int ntiles, int nq,
                                                              float *__restrict__ packed)
{
    const int64_t total = (int64_t)ntiles * nq * 4 * TAPS * 128;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 1), lane = (int)((idx >> 1) & 63);
        int64_t rest = idx >> 7;
        const int tap = (int)(rest % TAPS);
        rest /= TAPS;
        const int wv = (int)(rest & 3);
        rest >>= 2;
        const int q = (int)(rest % nq), tile = (int)(rest / nq);
        const int co = tile * 16 + (lane & 15);
        const int ci = q * CK + 8 * wv + 2 * (lane >> 4) + e;
        packed[idx] = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * TAPS + tap] : 0.0f;
    }
}

std::atomic<long long *> g_dbg{nullptr};   // sis3d_conv3d_k3t16_set_trace
std::atomic<int> g_dbg_cap{0};

template <int BX, int BY, int BZ, bool CLIP = false>
int launch_t16(T16Args &a, int nprob, hipStream_t st, int64_t ragged_blocks = 0)
{
    constexpr int M = BX * BY * BZ, MT = (M + 15) / 16;
    constexpr int G = (MT % 3 == 0) ? 3 : (MT >= 4 ? 4 : MT);
    constexpr int RB = (MT >= 14) ? 3 : 9;                 // weight-fragment ring (taps); must divide 27 so the ring index carries across chunks
    constexpr int ROWS = (BX + 2) * (BY + 2) * (BZ + 2);
    constexpr size_t img = (size_t)ROWS * RS * sizeof(float), red = (size_t)4 * MT * 256 * sizeof(float);
    constexpr size_t lds = img > red ? img : red;
    static_assert(lds <= 160 * 1024, "LDS brick too large");
    a.nbx = cdiv(a.X, BX); a.nby = cdiv(a.Y, BY); a.nbz = cdiv(a.Z, BZ);
    a.xcd_tg = a.xcd_bpb = 0;
    if (ragged_blocks == 0) {
        // split of an XCD's range (nwg / 8 work items) into bricks x tiles that keeps the least data in its L2
        static const int tg_mode = [] { const char *e = getenv("SIS3D_T16_XCD_TG"); return e ? atoi(e) : -1; }();   // 0: off, >0: forced
        const int64_t nbr = (int64_t)a.nbx * a.nby * a.nbz, nwg_ = nbr * a.ntiles;
        if (tg_mode != 0 && nwg_ % 8 == 0) {
            const int L = (int)(nwg_ / 8);
            const double wbytes = (double)a.ntiles * 16 * a.nq * CK * TAPS * 4, bbytes = (double)ROWS * a.nq * CK * 4;
            double best = -1;
            for (int tg = 1; tg <= a.ntiles; ++tg) {
                if (a.ntiles % tg || L % tg || nbr % (L / tg) || 8 % (a.ntiles / tg)) continue;   // 8 ranges = brick blocks x tile groups
                if (tg_mode > 0 && tg != tg_mode) continue;
                const double cost = wbytes * tg / a.ntiles + bbytes * (L / tg);
                if (best < 0 || cost < best) { best = cost; a.xcd_tg = tg; a.xcd_bpb = L / tg; }
            }
            if (a.xcd_tg == a.ntiles) a.xcd_tg = a.xcd_bpb = 0;          // the default order already
        }
    }
    a.dbg = g_dbg.load(std::memory_order_relaxed);
    a.dbg_cap = g_dbg_cap.load(std::memory_order_relaxed);
    auto kern = conv3d_k3t16_kernel<BX, BY, BZ, G, RB, CLIP>;
    // once per instantiation (function-local static), never per launch: a launch that re-sets the attribute while replays of a
    // captured graph containing the same kernel are being enqueued touches state the graph launch reads
    static Sis3dLdsOnce lds_once;                                   // per instantiation; granted once per device
    if (lds > 64 * 1024 && sis3d_grant_lds(lds_once, (const void *)kern, (int)lds) != SIS3D_OK) return SIS3D_ELAUNCH;
    const int64_t nwg = ragged_blocks > 0 ? ragged_blocks : (int64_t)a.nbx * a.nby * a.nbz * a.ntiles;
    if (nwg > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)nprob), dim3(256), lds, st, a);
    return sis3d_check_launch();
}


struct Brick { int bx, by, bz; };
constexpr Brick BRICKS[] = {{6, 6, 12}, {6, 6, 6}, {3, 6, 6}, {3, 3, 6}, {4, 4, 4}, {4, 4, 8}, {4, 8, 8}};
constexpr int NBRICKS = sizeof(BRICKS) / sizeof(BRICKS[0]);

// estimated SIMD-cycles of the slowest CU: rounds of 256 workgroups x (MFMA issue + chunk switches + fixed cost)
int64_t t16_cost(const Brick &b, int X, int Y, int Z, int cin, int ntiles, int nprob)
{
    const int64_t nb = (int64_t)cdiv(X, b.bx) * cdiv(Y, b.by) * cdiv(Z, b.bz);
    const int64_t nwg = nb * ntiles * nprob;
    const int64_t rounds = (nwg + 255) / 256;
    const int mt = (b.bx * b.by * b.bz + 15) / 16;
    const int rows = (b.bx + 2) * (b.by + 2) * (b.bz + 2);
    const int64_t per_wg = (int64_t)mt * TAPS * (cin / 16) * 32 + (int64_t)(cin / CK) * (rows * 8 / 256 * 30 + 1200) + 3000;
    return rounds * per_wg;
}

} // namespace

extern "C" size_t sis3d_conv_k3t16_packed_floats(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || cin % CK) return 0;
    return (size_t)((cout + 15) / 16) * (cin / CK) * 4 * TAPS * 128;
}

extern "C" int sis3d_conv_k3t16_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream)
{
    if (!w || !packed || cout <= 0 || cin <= 0 || cin % CK) return SIS3D_EINVAL;
    const int ntiles = (cout + 15) / 16, nq = cin / CK;
    const int64_t total = (int64_t)ntiles * nq * 4 * TAPS * 128;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weight_t16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), w, cout, cin,
                       ntiles, nq, packed);
    return sis3d_check_launch();
}

extern "C" int sis3d_conv3d_k3t16_set_trace(void *buf, int capacity_blocks)
{
    g_dbg.store((long long *)buf, std::memory_order_relaxed);
    g_dbg_cap.store(buf ? capacity_blocks : 0, std::memory_order_relaxed);
    return SIS3D_OK;
}

extern "C" int sis3d_conv3d_k3t16_brick(int X, int Y, int Z, int cin, int cout, int nprob, int max_voxels)
{
    if (X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || nprob < 1 || max_voxels < 0) return SIS3D_EINVAL;
    // tuning hook: SIS3D_K3_MAXVOX caps the brick volume (smaller bricks = less LDS / fewer registers per workgroup, so
    // workgroups of other streams' kernels can share the CU)
    static const int env_maxvox = [] { const char *e = getenv("SIS3D_K3_MAXVOX"); return e ? atoi(e) : -1; }();
    const int maxvox = env_maxvox >= 0 ? env_maxvox : max_voxels;       // the cap is the caller's (an argument since r5: no library state)
    int best = -1;
    int64_t bc = -1;
    for (int i = 0; i < NBRICKS; ++i) {
        if (maxvox > 0 && BRICKS[i].bx * BRICKS[i].by * BRICKS[i].bz > maxvox) continue;
        const int64_t c = t16_cost(BRICKS[i], X, Y, Z, cin, (cout + 15) / 16, nprob);
        if (bc < 0 || c < bc) { bc = c; best = i; }
    }
    return best < 0 ? 3 : best;
}

extern "C" int sis3d_conv3d_k3t16(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                  const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                                  int out_stride, int out_coff, int brick, sis3d_stream_t stream)
{
    if (nprob < 1 || nprob > T16_MAXP || !ins || !packed_ws || !outs) return SIS3D_EINVAL;
    if (X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || cin_stride < cin || (cin_stride % 4)) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || (out_stride % 4) || (out_coff % 4) || out_stride < out_coff + cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    if ((int64_t)X * Y * Z * cin_stride > 0x7fffffffLL) return SIS3D_EUNSUPPORTED;      // 32-bit element offsets in the staging table
    T16Args a;
    for (int p = 0; p < T16_MAXP; ++p) {
        const int s = p < nprob ? p : 0;
        if (!ins[s] || !packed_ws[s] || !outs[s]) return SIS3D_EINVAL;
        a.in[p] = ins[s]; a.wp[p] = packed_ws[s]; a.bias[p] = biases ? biases[s] : nullptr; a.out[p] = outs[s];
    }
    a.X = X; a.Y = Y; a.Z = Z; a.cin_stride = cin_stride; a.cout = cout; a.ntiles = (cout + 15) / 16; a.nq = cin / CK;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = out_coff;
    a.rag = nullptr; a.nrag = 0;
    if (brick < 0) brick = sis3d_conv3d_k3t16_brick(X, Y, Z, cin, cout, nprob, 0);
    hipStream_t st = as_stream(stream);
    switch (brick) {
    case 0: return launch_t16<6, 6, 12>(a, nprob, st);
    case 1: return launch_t16<6, 6, 6>(a, nprob, st);
    case 2: return launch_t16<3, 6, 6>(a, nprob, st);
    case 3: return launch_t16<3, 3, 6>(a, nprob, st);
    case 4: return launch_t16<4, 4, 4>(a, nprob, st);
    case 5: return launch_t16<4, 4, 8>(a, nprob, st);
    case 6: return launch_t16<4, 8, 8>(a, nprob, st);
    default: return SIS3D_EINVAL;
    }
}

// ---- ragged batch: every detected box's mask-head crop (lib/nets/network.py:303-317) through ONE launch per k3 layer.
// The caller picks the brick (3x6x6, 3x3x6, 4x4x4 or 4x4x8) that wastes the fewest tile slots on its crops: detections are
// 10-40 voxels across, so the padding of partial bricks decides the efficiency (3x6x6 is best on a 30x30x36 crop -- 65 us
// vs 89 us for conv3d.hip's 32x32 tiles -- but covers a 14^3 crop with 56 % fill).
extern "C" int sis3d_ragged_tiling_k3t16(int cin, int cout, int brick, int *bx, int *by, int *bz, int *ngroups, int *tiles_per_wave)
{
    if (!bx || !by || !bz || !ngroups) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || brick < 0 || brick >= NBRICKS) return SIS3D_EUNSUPPORTED;
    *bx = BRICKS[brick].bx; *by = BRICKS[brick].by; *bz = BRICKS[brick].bz;
    *ngroups = (cout + 15) / 16;
    if (tiles_per_wave) *tiles_per_wave = (BRICKS[brick].bx * BRICKS[brick].by * BRICKS[brick].bz + 15) / 16;
    return SIS3D_OK;
}

extern "C" int sis3d_conv3d_k3t16_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                         int flags, float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks,
                                         int brick, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || !desc_dev || ndesc <= 0 || total_blocks <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % CK) || (cout % 4) || (cin_stride % 4) || cin_stride < cin || (out_stride % 4) || out_stride < cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    T16Args a;
    for (int p = 0; p < T16_MAXP; ++p) { a.in[p] = in; a.wp[p] = packed_w; a.bias[p] = bias; a.out[p] = out; }
    a.X = a.Y = a.Z = 1; a.cin_stride = cin_stride; a.cout = cout; a.ntiles = (cout + 15) / 16; a.nq = cin / CK;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = 0;
    a.rag = (const T16Ragged *)desc_dev; a.nrag = ndesc;
    hipStream_t st = as_stream(stream);
    static const bool noclip = [] { const char *e = getenv("SIS3D_T16_NOCLIP"); return e && atoi(e) != 0; }();   // A/B hook
    if (noclip) {
        switch (brick) {
        case 2: return launch_t16<3, 6, 6>(a, 1, st, total_blocks);
        case 3: return launch_t16<3, 3, 6>(a, 1, st, total_blocks);
        case 4: return launch_t16<4, 4, 4>(a, 1, st, total_blocks);
        case 5: return launch_t16<4, 4, 8>(a, 1, st, total_blocks);
        default: return SIS3D_EUNSUPPORTED;
        }
    }
    switch (brick) {
    case 2: return launch_t16<3, 6, 6, true>(a, 1, st, total_blocks);
    case 3: return launch_t16<3, 3, 6, true>(a, 1, st, total_blocks);
    case 4: return launch_t16<4, 4, 4, true>(a, 1, st, total_blocks);
    case 5: return launch_t16<4, 4, 8, true>(a, 1, st, total_blocks);
    case 1: return launch_t16<6, 6, 6, true>(a, 1, st, total_blocks);
    default: return SIS3D_EUNSUPPORTED;
    }
}
