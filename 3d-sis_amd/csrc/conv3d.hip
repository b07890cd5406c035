// Dense 3D convolution for the 3D-SIS backbone / RPN / mask head on gfx950 (CDNA4).
//
// Replaces the cuDNN calls behind nn.Conv3d(k in {1, 2/s2, 3/p1}) + bias + ReLU + residual
// add + sigmoid of lib/nets/backbones.py:17-40,171-287 and lib/nets/network.py:38-47,537-574.
//
// Formulation: implicit GEMM  D[voxel][cout] = sum_{tap,cin} A[voxel+tap][cin] * W[tap][cin][cout]
// on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the fp32
// roof of the chip, bitwise an fmaf chain -> well inside the 1e-4 logit tolerance).
//
//  * Activations are channels-last (x,y,z,c).  A workgroup owns a BX x BY x BZ brick of output
//    voxels; the input halo brick for one CK-channel chunk is staged ONCE in LDS with fully
//    coalesced 16 B loads and then serves all 27 (8, 1) taps: a tap is just a constant row offset
//    into the LDS image, so there is no im2col buffer and no re-read of the input from L2.
//  * A fragments: the MFMA wants A[i = lane&31][k = lane>>5]; the K order inside a group of 8
//    channels is permuted so that lane l consumes channels 4*(l>>5)..+3 in four consecutive
//    MFMAs -> one ds_read_b128 per four MFMAs.  LDS rows are padded to CK+4 floats: row stride
//    = 9 (17) x 16 B, odd, so the 16-lane groups of ds_read_b128 spread over all 16 bank slots.
//  * B fragments never touch LDS: weights are repacked once into "fragment order"
//    [cout/32][tap][cin/8][lane][4], so each wave reads exactly the 1 KiB it needs per group of
//    four MFMAs with one fully coalesced global_load_dwordx4 (L1/L2-resident: all waves of a
//    workgroup with the same cout tile hit the same lines), prefetched one tap ahead in
//    registers.  No barrier is needed per tap, only one pair per CK-channel chunk.
//  * Epilogue fused: + bias, + residual, ReLU / sigmoid, channel-offset write (torch.cat of the
//    colour and geometry branches), and the RPN head's permute/view into the
//    (2,X,Y,Z,A) score and (X,Y,Z,6A) bbox layouts.
//
// This file may use FMA freely (fp32 tolerance 1e-4 applies, not bit-exactness).
#include "common.h"
#include "mfma16.h"
#include <stdlib.h>
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct PwStage {
    const float *wp, *bias, *res;
    float *out;
    int cout, res_stride, out_stride, flags;
};

// ragged batch: many independent problems of DIFFERENT grid sizes in one launch (the per-box mask-head convs)
struct RaggedDesc {
    int X, Y, Z;          // grid of this problem (stride-1 convs: output grid == input grid)
    int nbx, nby, nbz;    // bricks per axis for the launched tiling
    int block0;           // first block of this problem in the launch
    int pad;
    int64_t in_off, out_off;   // element offsets of this problem's activations inside the packed in / out buffers
};

constexpr int MAXP = 4;   // independent same-shape problems per launch (e.g. the two RPN levels)

struct ConvArgs {
    const float *in;
    int X, Y, Z;          // input grid
    int cin, cin_stride;
    const float *wp;      // packed weights
    const float *bias;
    int cout, ntiles;     // ntiles = ceil(cout/32)
    int OX, OY, OZ;       // output grid
    int flags;
    const float *res;
    int res_stride;
    float *out;
    int out_stride, out_coff;
    float *out2;
    float *out3;          // RPN head: softmax over the two class planes, same layout as the score map (may be NULL)
    int anchors;
    int nbx, nby, nbz;    // bricks per axis
    int ngroups;          // cout groups per brick
    // fused pointwise (1x1x1) stages applied to the output tile while it is still on chip (Bottleneck conv3 +
    // residual + ReLU, and the NEXT block's conv1 + ReLU): see sis3d_conv3d_chain
    int npw;
    PwStage pw[2];
    // batched launch: problems 1..nprob-1 share every shape field and differ only in these pointers
    const RaggedDesc *rag;   // device array, nrag entries (nrag == 0: regular launch)
    int nrag;
    int64_t ragged_blocks;   // total blocks of a ragged launch
    int nprob;
    const float *b_in[MAXP], *b_wp[MAXP], *b_bias[MAXP], *b_res[MAXP];
    float *b_out[MAXP], *b_out2[MAXP];
    // projected input (sis3d_conv3d_chain_projected): the input volume is never materialised; a voxel's channel
    // vector is gathered while the halo brick is staged: max over the view slots of (visible ? feature row : 0)
    const int32_t *ptab = nullptr;   // [pslots][X*Y*Z] voxel -> pixel (-1 = not visible), linear voxel index z*X*Y + y*X + x
    const float *pfeat = nullptr;    // [pslots][pnpix][cin] pixel-major feature rows
    int pslots = 0;
    int64_t pnpix = 0;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// occupancy floor (waves per SIMD) of the fused-stage instantiations: two workgroups per CU for workgroups of <= 8 waves and of
// 12 waves, one otherwise -- without it the stage code's extra registers would halve the residency of the 8-wave kernels
constexpr int fused_min_waves(bool pw, int waves)
{
    return !pw ? 1 : waves <= 8 ? (2 * waves + 3) / 4 : waves == 12 ? 6 : (waves + 3) / 4;
}

// KS: kernel size (1,2,3); S: stride; brick BX*BY*BZ = 32*MW output voxels; NW waves along cout, each NTW
// 32-wide cout tiles; KW waves split the reduction (taps for k=2/3, channel groups for k=1) of the SAME
// output tile and are summed through LDS at the end (intra-workgroup split-K: no atomics, deterministic);
// CK channels per LDS chunk (k=1: CK == cin, a single chunk).
// PW: compile the fused pointwise stages in (separate instantiation so the plain kernels keep their register budget)
// PROJ: the input is a back-projected image volume given as (voxel->pixel table, pixel-major features), see ConvArgs::ptab
template <int KS, int S, int BX, int BY, int BZ, int MW, int NW, int KW, int NTW, int CK, bool PW = false, int PF = 4, bool PROJ = false>
__global__ __launch_bounds__(64 * MW *NW *KW, fused_min_waves(PW, MW *NW *KW)) void conv3d_mfma_kernel(const ConvArgs a)
{
    static_assert(BX * BY * BZ == 32 * MW, "brick must hold 32*MW voxels");
    constexpr int T = KS * KS * KS;
    constexpr bool SPLIT_TAPS = (KS != 1);
    static_assert(!SPLIT_TAPS || T % KW == 0, "taps must split evenly over the KW waves");
    constexpr int TPW = SPLIT_TAPS ? T / KW : 1;           // taps per wave
    constexpr int PAD = (KS == 3) ? 1 : 0;
    constexpr int IBX = (BX - 1) * S + KS, IBY = (BY - 1) * S + KS, IBZ = (BZ - 1) * S + KS;
    constexpr int ROWS = IBX * IBY * IBZ;
    constexpr int RS = CK + 4;                             // padded LDS row stride (floats)
    constexpr int KGC = CK / 8;                            // k-groups (8 channels) per chunk
    static_assert(SPLIT_TAPS || KGC % KW == 0, "channel groups must split evenly over the KW waves");
    constexpr int KGW = SPLIT_TAPS ? KGC : KGC / KW;       // k-groups this wave consumes per tap
    constexpr int NTHREADS = 64 * MW * NW * KW;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [ROWS][RS], later reused for the split-K reduction

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = wave % MW, nw = (wave / MW) % NW, kw = wave / (MW * NW);
    const int li = lane & 31, kh = lane >> 5;

    // block -> (brick, cout group).  XCD-aware: the dispatcher places block b on XCD b % 8, each with a private
    // 4 MiB L2; remap so every XCD owns one CONTIGUOUS range of the (group-major, brick-minor) work list, i.e.
    // touches one or two cout groups -> its weight working set (<= 0.9 MB per group) stays L2-resident instead of
    // all 8 L2s streaming the whole weight tensor.  Bijective for any grid size.  (+3 % on the k3 layers.)
    int bid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    // batched launch: the work list is problem-major; pick this block's pointer set (uniform -> scalar loads;
    // never copy the argument struct into a local, dynamic indexing would push it to scratch)
    const float *p_in = a.in, *p_wp = a.wp, *p_bias = a.bias, *p_res = a.res;
    float *p_out = a.out, *p_out2 = a.out2;
    int gX = a.X, gY = a.Y, gZ = a.Z, gOX = a.OX, gOY = a.OY, gOZ = a.OZ, nbx = a.nbx, nby = a.nby, nbz = a.nbz;
    if (a.nrag > 0) {
        // ragged batch: find this block's problem (block0 is ascending) -- all uniform, scalar loads
        int lo = 0, hi = a.nrag - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.rag[mid].block0 <= bid) lo = mid; else hi = mid - 1;
        }
        const RaggedDesc d = a.rag[lo];
        bid -= d.block0;
        gX = gOX = d.X; gY = gOY = d.Y; gZ = gOZ = d.Z;
        nbx = d.nbx; nby = d.nby; nbz = d.nbz;
        p_in += d.in_off;
        if (p_out) p_out += d.out_off;
    }
    const int nbricks = nbx * nby * nbz;
    if (a.nprob > 1) {
        const int per = nbricks * a.ngroups;
        const int prob = bid / per;
        bid -= prob * per;
        p_in = a.b_in[prob]; p_wp = a.b_wp[prob]; p_bias = a.b_bias[prob]; p_res = a.b_res[prob];
        p_out = a.b_out[prob]; p_out2 = a.b_out2[prob];
    }
    const int group = bid / nbricks;
    bid -= group * nbricks;
    const int bz = bid % nbz, by = (bid / nbz) % nby, bx = bid / (nbz * nby);
    const int ox0 = bx * BX, oy0 = by * BY, oz0 = bz * BZ;           // output brick origin
    const int ix0 = ox0 * S - PAD, iy0 = oy0 * S - PAD, iz0 = oz0 * S - PAD;

    // this lane's A row (output voxel m of the brick)
    const int m = 32 * mw + li;
    const int lx = m / (BY * BZ), ly = (m / BZ) % BY, lz = m % BZ;
    const int arow = ((S * lx) * IBY + S * ly) * IBZ + S * lz;
    const float *aptr = lds + arow * RS + 4 * kh + (SPLIT_TAPS ? 0 : 8 * KGW * kw);

    const int tile0 = (group * NW + nw) * NTW;                       // first 32-wide cout tile of this wave
    const int kgtot = a.cin / 8;
    const int nchunks = SPLIT_TAPS ? a.cin / CK : 1;

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // ---- software pipeline at k-group granularity.  One step = one group of 8 input channels of one tap =
    // one 16 B A fragment (LDS) + one 16 B B fragment per cout tile (global, fragment-order packed) + 4 MFMAs per
    // tile.  B fragments are fetched DB-1 steps ahead (L2 latency), A fragments one step ahead (LDS latency);
    // rings are indexed with compile-time constants (the step loop is fully unrolled) and the order is pinned
    // with sched_barrier so hipcc cannot sink the loads next to their use.  Small rings keep the kernel at
    // <= 64 VGPRs -> 8 waves/SIMD, which is what hides the per-chunk LDS refill of the other workgroups.
    constexpr int NS = TPW * KGW;                          // steps per chunk for this wave
    // PF = requested ring depth: 4 when many waves share a SIMD, 12 for the small layers that run 1-2 waves per SIMD
    constexpr int DB = (PF >= 12 && NS % 12 == 0) ? 12 : (PF >= 8 && NS % 8 == 0) ? 8 : (NS % 4 == 0) ? 4 : (NS % 3 == 0) ? 3 : (NS % 2 == 0) ? 2 : 1;
    constexpr int DA = 2;                                  // A ring restarts at slot 0 every chunk: no wrap constraint
    float4 bq[DB][NTW];
    float4 aq[DA];
    auto load_b = [&](float4(&dst)[NTW], int q, int s) {
        const int ti = s / KGW, g = s % KGW;
        const int tap = SPLIT_TAPS ? kw + KW * ti : 0;
        const int kg = SPLIT_TAPS ? q * KGC + g : KGW * kw + g;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = min(tile0 + t, a.ntiles - 1);           // clamp: surplus tiles recompute the last one, never stored
            dst[t] = reinterpret_cast<const float4 *>(p_wp)[((size_t)(tile * T + tap) * kgtot + kg) * 64 + lane];
        }
    };
    auto load_a = [&](float4 &dst, int s) {
        const int ti = s / KGW, g = s % KGW;
        const int tap = SPLIT_TAPS ? kw + KW * ti : 0;
        const int dz = tap % KS, dy = (tap / KS) % KS, dx = tap / (KS * KS);
        dst = *reinterpret_cast<const float4 *>(aptr + ((dx * IBY + dy) * IBZ + dz) * RS + 8 * g);
    };
    // projected input: >98 % of the voxels are seen by no view; a brick without a single visible voxel contributes
    // exactly zero, so its workgroup skips the whole reduction and goes straight to the epilogue (bias, ReLU, fused stages)
    bool brick_live = true;
    constexpr int PSL = 8;                                 // view slots whose per-row pixel index is cached in LDS
    __shared__ int32_t ppix[PROJ ? PSL * ROWS : 1];
    if constexpr (PROJ) {
        // one pass over the brick's (row, slot) table entries: cache them for the four channel-chunk refills, and find
        // out whether anything is visible at all
        const int64_t pnvox = (int64_t)gX * gY * gZ;
        int seen = 0;
        for (int idx = tid; idx < ROWS * a.pslots; idx += NTHREADS) {
            const int row = idx % ROWS, sl = idx / ROWS;
            const int hz = row % IBZ, hy = (row / IBZ) % IBY, hx = row / (IBZ * IBY);
            const int gx = ix0 + hx, gy = iy0 + hy, gz = iz0 + hz;
            int pix = -1;
            if (gx >= 0 && gx < gX && gy >= 0 && gy < gY && gz >= 0 && gz < gZ)
                pix = a.ptab[sl * pnvox + ((int64_t)gz * gY + gy) * gX + gx];
            if (sl < PSL) ppix[sl * ROWS + row] = pix;
            seen |= pix >= 0;
        }
        brick_live = __syncthreads_or(seen) != 0;
    }

    if (brick_live) {
#pragma unroll
    for (int d = 0; d < (DB > 1 ? DB - 1 : 1); ++d)
        if (d < NS) load_b(bq[d], 0, d);

    for (int q = 0; q < nchunks; ++q) {
        if (q) __syncthreads();
        // ---- stage the halo brick of chunk q: ROWS rows x CK floats, 16 B per thread, coalesced
        constexpr int ITEMS = ROWS * (CK / 4), NIT = (ITEMS + NTHREADS - 1) / NTHREADS;
        if constexpr (!PROJ && NIT <= 4 && (KS == 1 || PW)) {
            // all of a thread's (<= 4) 16-byte pieces are requested before the first is written to LDS: the rolled loop
            // compiles to load -> wait -> ds_write per piece, i.e. NIT exposed L2 latencies per chunk instead of one.  Used
            // for the latency-bound kernels (1x1x1 convs, fused Bottleneck launches: 42.7 -> 37.5 us per 128/32 pair); the
            // long plain k3 kernels hide that latency behind their six resident waves and measured 1-3 % slower with it
            float4 sv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NTHREADS;
                const int row = idx / (CK / 4), c4 = idx % (CK / 4);
                const int hz = row % IBZ, hy = (row / IBZ) % IBY, hx = row / (IBZ * IBY);
                const int gx = ix0 + hx, gy = iy0 + hy, gz = iz0 + hz;
                sv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < ITEMS && gx >= 0 && gx < gX && gy >= 0 && gy < gY && gz >= 0 && gz < gZ)
                    sv[it] = *reinterpret_cast<const float4 *>(p_in + ((size_t)(gx * gY + gy) * gZ + gz) * a.cin_stride + q * CK + c4 * 4);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NTHREADS;
                if (idx < ITEMS) *reinterpret_cast<float4 *>(lds + (idx / (CK / 4)) * RS + (idx % (CK / 4)) * 4) = sv[it];
            }
        } else
        for (int idx = tid; idx < ROWS * (CK / 4); idx += NTHREADS) {
            const int row = idx / (CK / 4), c4 = idx % (CK / 4);
            const int hz = row % IBZ, hy = (row / IBZ) % IBY, hx = row / (IBZ * IBY);
            const int gx = ix0 + hx, gy = iy0 + hy, gz = iz0 + hz;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gx >= 0 && gx < gX && gy >= 0 && gy < gY && gz >= 0 && gz < gZ) {
                if constexpr (PROJ) {
                    // network.py:216-239 per voxel: max over the views, an invisible view counts as 0
                    const int64_t pnvox = (int64_t)gX * gY * gZ, vox = ((int64_t)gz * gY + gy) * gX + gx;
                    int cnt = 0;
                    for (int sl = 0; sl < a.pslots; ++sl) {
                        const int pix = sl < PSL ? ppix[sl * ROWS + row] : a.ptab[sl * pnvox + vox];
                        if (pix >= 0) {
                            const float4 f = *reinterpret_cast<const float4 *>(a.pfeat + ((size_t)sl * a.pnpix + pix) * a.cin + q * CK + c4 * 4);
                            if (cnt == 0) v = f;
                            else { v.x = fmaxf(v.x, f.x); v.y = fmaxf(v.y, f.y); v.z = fmaxf(v.z, f.z); v.w = fmaxf(v.w, f.w); }
                            ++cnt;
                        }
                    }
                    if (cnt > 0 && cnt < a.pslots) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                } else {
                    v = *reinterpret_cast<const float4 *>(p_in + ((size_t)(gx * gY + gy) * gZ + gz) * a.cin_stride + q * CK + c4 * 4);
                }
            }
            *reinterpret_cast<float4 *>(lds + row * RS + c4 * 4) = v;
        }
        __syncthreads();
        load_a(aq[0], 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (DB > 1) {
                const int sn = s + DB - 1;                 // wraps into the next chunk's first steps
                if (sn < NS) load_b(bq[sn % DB], q, sn);
                else if (q + 1 < nchunks) load_b(bq[sn % DB], q + 1, sn - NS);
            }
            if (s + 1 < NS) load_a(aq[(s + 1) % DA], s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 av = aq[s % DA];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float4 bv = bq[s % DB][t];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }   // brick_live

    // ---- split-K reduction through LDS: waves kw>0 publish, waves kw==0 accumulate and run the epilogue
    if (KW > 1 && brick_live) {
        __syncthreads();                                   // everyone is done reading the A image
        if (kw > 0) {
            float *red = lds + ((size_t)((kw - 1) * (MW * NW) + nw * MW + mw) * NTW) * 1024;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (kw == 0)
#pragma unroll
        for (int k = 1; k < KW; ++k) {
            const float *red = lds + ((size_t)((k - 1) * (MW * NW) + nw * MW + mw) * NTW) * 1024;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
        }
    }

    // ---- epilogue.  D layout: column (cout) = lane&31, row (voxel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int64_t nvox_out = (int64_t)gOX * gOY * gOZ;
    constexpr int MROWS = 32 * MW;
    const int npw = PW ? a.npw : 0;
    const int c0s = a.cout + 4;                             // padded row stride of the on-chip output tile
    if (PW && npw > 0) __syncthreads();                     // A image / reduction buffers are dead before the tile overwrites them
    if (kw == 0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = tile0 + t;
            if (tile >= a.ntiles) continue;
            const int co = tile * 32 + li;
            if (co >= a.cout) continue;
            const float bv = p_bias ? p_bias[co] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = 32 * mw + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int ox = ox0 + mm / (BY * BZ), oy = oy0 + (mm / BZ) % BY, oz = oz0 + mm % BZ;
                const bool inside = ox < gOX && oy < gOY && oz < gOZ;
                const int64_t vox = ((int64_t)ox * gOY + oy) * gOZ + oz;
                float v = acc[t][r] + bv;
                if ((a.flags & SIS3D_EPI_RESIDUAL) && inside) v += p_res[vox * a.res_stride + co];
                if (a.flags & SIS3D_EPI_RELU) v = fmaxf(v, 0.0f);
                if (a.flags & SIS3D_EPI_SIGMOID) v = sigmoidf_(v);
                if (PW && npw > 0) lds[mm * c0s + co] = v;        // keep the tile on chip for the fused 1x1x1 stages
                if (a.flags & SIS3D_EPI_RPN_HEAD) {
                    // score channel c pairs with c +- A (background / foreground of the same anchor): both live in this
                    // wave's first tile, same accumulator row -> the 2-way softmax of network.py:546 is one lane exchange
                    const int A = a.anchors;
                    const int pl = co < A ? li + A : (co < 2 * A ? li - A : li);
                    const float pv = __shfl(v, (lane & 32) | pl);
                    if (!inside) continue;
                    if (co < 2 * A) {
                        const int64_t o = ((int64_t)(co / A) * nvox_out + vox) * A + (co % A);
                        p_out[o] = v;
                        if (a.out3) {
                            const float mx = fmaxf(v, pv);
                            const float e = expf(v - mx), ep = expf(pv - mx);
                            a.out3[o] = e / (e + ep);
                        }
                    } else {
                        p_out2[vox * (6 * A) + (co - 2 * A)] = v;
                    }
                    continue;
                }
                if (!inside || !p_out) continue;
                p_out[vox * a.out_stride + a.out_coff + co] = v;
            }
        }
    }
    if constexpr (PW) {
    if (npw == 0) return;
    // ---- fused pointwise stages: out_s[M][Cs] = act(tile[M][Cprev] * W_s + b_s (+ residual)); every wave of the
    // workgroup takes 32x32 tiles of the stage output.  A from the LDS tile, B from L2 (fragment-order pack).
    constexpr int NWAVES = MW * NW * KW;
    auto run_stage = [&](const PwStage &st, const float *tin, float *tout, int cprev, bool has_next) {
        __syncthreads();
        const int cs = st.cout, kgs = cprev / 8, ins = cprev + 4, outs = cs + 4;
        const int ntl = MW * (cs / 32);
        for (int tl = wave; tl < ntl; tl += NWAVES) {
            const int mt = tl % MW, nt = tl / MW;
            f32x16 c2;
#pragma unroll
            for (int r = 0; r < 16; ++r) c2[r] = 0.0f;
            const float *ap2 = tin + (32 * mt + li) * ins + 4 * kh;
            const float4 *bp2 = reinterpret_cast<const float4 *>(st.wp) + (size_t)nt * kgs * 64 + lane;
            // k-groups in blocks of 4 (channel counts are multiples of 32): the four weight fragments and the four A
            // fragments of a block are requested back to back, then consumed -- one exposed L2 latency per block instead of
            // one per k-group (a plain `#pragma unroll` is refused here: "loop not unrolled")
            for (int g0 = 0; g0 < kgs; g0 += 4) {
                float4 bv[4], av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bp2[(g0 + j) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const float4 *>(ap2 + 8 * (g0 + j));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].x, bv[j].x, c2, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].y, bv[j].y, c2, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].z, bv[j].z, c2, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].w, bv[j].w, c2, 0, 0, 0);
                }
            }
            const int co = nt * 32 + li;
            const float bb = st.bias ? st.bias[co] : 0.0f;
            auto row_of = [&](int r, int &mm, bool &inside, int64_t &vox) {
                mm = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int ox = ox0 + mm / (BY * BZ), oy = oy0 + (mm / BZ) % BY, oz = oz0 + mm % BZ;
                inside = ox < gOX && oy < gOY && oz < gOZ;
                vox = ((int64_t)ox * gOY + oy) * gOZ + oz;
            };
            // residual operands first, all 16 requests in flight together (they used to be loaded and awaited one by one)
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int mm; bool inside; int64_t vox;
                row_of(r, mm, inside, vox);
                rv[r] = ((st.flags & SIS3D_EPI_RESIDUAL) && inside) ? st.res[vox * st.res_stride + co] : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int mm; bool inside; int64_t vox;
                row_of(r, mm, inside, vox);
                float v = c2[r] + bb;
                v += rv[r];
                if (st.flags & SIS3D_EPI_RELU) v = fmaxf(v, 0.0f);
                if (has_next) tout[mm * outs + co] = v;
                if (inside && st.out) st.out[vox * st.out_stride + co] = v;
            }
        }
    };
    float *t1 = lds + MROWS * c0s;
    run_stage(a.pw[0], lds, t1, a.cout, npw > 1);
    if (npw > 1) run_stage(a.pw[1], t1, lds, a.pw[0].cout, false);
    }
}

// ---- weight repack: (Cout,Cin,k,k,k) -> [cout/32][tap][cin8/8][lane 64][4] -------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, int cout, int cin, int T, int ntiles,
                                                          int kgtot, float *__restrict__ packed)
{
    const int64_t total = (int64_t)ntiles * T * kgtot * 256;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        int64_t rest = idx >> 8;
        const int kg = (int)(rest % kgtot);
        rest /= kgtot;
        const int tap = (int)(rest % T), tile = (int)(rest / T);
        const int co = tile * 32 + (lane & 31);
        const int ci = 8 * kg + 4 * (lane >> 5) + e;
        float v = 0.0f;
        if (co < cout && ci < cin) v = w[((int64_t)co * cin + ci) * T + tap];
        packed[idx] = v;
    }
}

// ---- first layers on the planar 2-channel grid (VALU: K = 16 / 54 is too thin for the matrix pipe) ------
// thread = (output voxel, 16 consecutive couts): each of the 2*KS^3 input taps is loaded once per 16 outputs; weights
// transposed into LDS as [ci*T+tap][cout]
template <int KS, int S>
__global__ __launch_bounds__(256) void conv_planar2_kernel(const float *__restrict__ in, int64_t is_c, int64_t is_x, int64_t is_y,
                                                           int x0, int y0, int z0, int WX, int WY, int WZ, int OX, int OY, int OZ,
                                                           const float *__restrict__ w, int cout, int flags,
                                                           float *__restrict__ out, int out_stride)
{
    constexpr int T = KS * KS * KS, PAD = (KS == 3) ? 1 : 0, K = 2 * T, CPT = 16;
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [K][cout]
    for (int i = threadIdx.x; i < K * cout; i += blockDim.x) {
        const int co = i % cout, k = i / cout;                     // k = ci*T + tap
        wl[i] = w[(int64_t)co * K + k];
    }
    __syncthreads();
    const int cg = (cout + CPT - 1) / CPT;
    const int64_t total = (int64_t)OX * OY * OZ * cg;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(t % cg) * CPT;
        const int64_t v = t / cg;
        const int oz = (int)(v % OZ), oy = (int)((v / OZ) % OY), ox = (int)(v / ((int64_t)OZ * OY));
        float4 acc[CPT / 4];
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // input taps in batches (all 16 for k2, one (channel, dx) slice of 9 for k3): clamped addresses, padding applied as
        // a select, the batch's loads in flight together, then its FMAs.  Tap-by-tap the loads were awaited one at a time.
        constexpr int NB = (KS == 2) ? 1 : 2 * KS, BT = K / NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float xs[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const int k = b * BT + u, ci = k / T, tap = k % T;
                const int dx = tap / (KS * KS), dy = (tap / KS) % KS, dz = tap % KS;
                const int wx = ox * S + dx - PAD, wy = oy * S + dy - PAD, wz = oz * S + dz - PAD;   // window coords
                const bool ok = wx >= 0 && wx < WX && wy >= 0 && wy < WY && wz >= 0 && wz < WZ;
                const int cx = min(max(wx, 0), WX - 1), cy = min(max(wy, 0), WY - 1), cz = min(max(wz, 0), WZ - 1);
                const float t = in[ci * is_c + (int64_t)(x0 + cx) * is_x + (int64_t)(y0 + cy) * is_y + (z0 + cz)];
                xs[u] = ok ? t : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float xv = xs[u];
                const float *wr = wl + (b * BT + u) * cout + c0;
#pragma unroll
                for (int j = 0; j < CPT / 4; ++j) {
                    if (c0 + 4 * j >= cout) break;
                    const float4 wv = *reinterpret_cast<const float4 *>(wr + 4 * j);
                    acc[j].x = fmaf(xv, wv.x, acc[j].x); acc[j].y = fmaf(xv, wv.y, acc[j].y);
                    acc[j].z = fmaf(xv, wv.z, acc[j].z); acc[j].w = fmaf(xv, wv.w, acc[j].w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) {
            if (c0 + 4 * j >= cout) break;
            float4 r = acc[j];
            if (flags & SIS3D_EPI_RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            *reinterpret_cast<float4 *>(out + v * out_stride + c0 + 4 * j) = r;
        }
    }
}

template <int KS, int S, int BX, int BY, int BZ, int MW, int NW, int KW, int NTW, int CK, bool PW = false, int PF = 4, bool PROJ = false>
int launch_cfg_(ConvArgs &a, hipStream_t st)
{
    constexpr int IBX = (BX - 1) * S + KS, IBY = (BY - 1) * S + KS, IBZ = (BZ - 1) * S + KS;
    constexpr size_t tile_b = (size_t)IBX * IBY * IBZ * (CK + 4) * sizeof(float);
    constexpr size_t red_b = (size_t)(KW - 1) * MW * NW * NTW * 1024 * sizeof(float);
    constexpr size_t lds0 = tile_b > red_b ? tile_b : red_b;
    static_assert(lds0 <= 160 * 1024, "LDS brick too large");
    size_t lds = lds0;
    if (PW && a.npw > 0) {
        // on-chip output tile (+ the first stage's output tile when a second stage follows), padded rows
        if (cdiv(a.ntiles, NW * NTW) != 1) return SIS3D_EUNSUPPORTED;      // the workgroup must own every output channel
        const size_t mrows = 32 * MW;
        size_t need = mrows * (a.cout + 4) * sizeof(float);
        if (a.npw > 1) need += mrows * (a.pw[0].cout + 4) * sizeof(float);
        lds = need > lds ? need : lds;
        if (lds > 160 * 1024) return SIS3D_EUNSUPPORTED;
    }
    static_assert(64 * MW * NW * KW <= 1024, "workgroup too large");
    a.nbx = cdiv(a.OX, BX); a.nby = cdiv(a.OY, BY); a.nbz = cdiv(a.OZ, BZ);
    a.ngroups = cdiv(a.ntiles, NW * NTW);
    auto kern = conv3d_mfma_kernel<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, PW, PF, PROJ>;
    if (lds > 64 * 1024) {
        // per device and thread-safe: a relaxed atomic per (instantiation, device) remembers the largest size already granted
        static std::atomic<size_t> set_to[16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t> &slot = set_to[dev & 15];
        if (lds > slot.load(std::memory_order_relaxed)) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            size_t cur = slot.load(std::memory_order_relaxed);
            while (lds > cur && !slot.compare_exchange_weak(cur, lds, std::memory_order_relaxed)) {}
        }
    }
    const int64_t blocks = a.nrag > 0 ? a.ragged_blocks : (int64_t)a.nbx * a.nby * a.nbz * a.ngroups * (a.nprob > 1 ? a.nprob : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * MW * NW * KW), lds, st, a);
    return sis3d_check_launch();
}

template <int KS, int S, int BX, int BY, int BZ, int MW, int NW, int KW, int NTW, int CK, int PF = 4>
int launch_cfg(ConvArgs &a, hipStream_t st)
{
    static const int pf_override = [] { const char *e = getenv("SIS3D_PF"); return e ? atoi(e) : 0; }();
    if constexpr (KS == 2 && S == 2 && CK == 32 && NW * NTW <= 4) {
        if (a.ptab) {                                      // projected input: only the colour stem (k2 s2 + fused conv1) uses it
            if (a.npw == 0) return SIS3D_EUNSUPPORTED;
            return launch_cfg_<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, true, 4, true>(a, st);
        }
    } else {
        if (a.ptab) return SIS3D_EUNSUPPORTED;
    }
    if constexpr ((KS != 1 && CK == 32 && NW * NTW <= 4) || (KS == 1 && CK <= 64 && NW * NTW <= 4 && MW * NW * KW <= 8)) {
        if (a.npw > 0) {
            if (PF >= 12 && pf_override == 12) return launch_cfg_<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, true, 12>(a, st);
            return launch_cfg_<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, true, 4>(a, st);
        }
    } else {
        if (a.npw > 0) return SIS3D_EUNSUPPORTED;
    }
    if constexpr (PF >= 12) {
        if (pf_override == 12) return launch_cfg_<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, false, 12>(a, st);
    }
    return launch_cfg_<KS, S, BX, BY, BZ, MW, NW, KW, NTW, CK, false, 4>(a, st);
}

// ---- tiling choice.  The chip has 1024 SIMDs; a 32x32 output tile is the work quantum of one wave, and the
// layers of this network have only 216..1728 such tiles, so the reduction is split over KW waves per tile
// (taps for k=3/2, channel groups for k=1) to put >= ~2-5 waves on every SIMD.
template <int KS, int S>
int dispatch(ConvArgs &a, hipStream_t st)
{
    if (a.cin % 8) return SIS3D_EUNSUPPORTED;
    const int64_t nvox = (int64_t)a.OX * a.OY * a.OZ;
    const bool big = nvox >= 32768;            // 48x24x48-class layers
    if constexpr (KS == 1) {
        // k=1: one LDS chunk holds all input channels (CK == cin)
        switch (a.cin) {
        case 8:
            return launch_cfg<1, 1, 4, 4, 4, 2, 1, 1, 1, 8>(a, st);
        case 32:
            if (a.ntiles >= 4) return launch_cfg<1, 1, 2, 4, 4, 1, 4, 2, 1, 32>(a, st);
            if (big) {
                static const int v1 = [] { const char *e = getenv("SIS3D_K1_VARIANT"); return e ? atoi(e) : 0; }();
                switch (v1) {
                case 1: return launch_cfg<1, 1, 4, 4, 4, 2, 1, 1, 1, 32>(a, st);
                case 2: return launch_cfg<1, 1, 2, 4, 4, 1, 1, 1, 1, 32>(a, st);
                case 3: return launch_cfg<1, 1, 4, 4, 8, 4, 1, 1, 1, 32>(a, st);
                case 4: return launch_cfg<1, 1, 2, 4, 4, 1, 1, 2, 1, 32>(a, st);
                default: return launch_cfg<1, 1, 4, 4, 4, 2, 1, 2, 1, 32>(a, st);
                }
            }
            return launch_cfg<1, 1, 2, 4, 4, 1, 1, 4, 1, 32>(a, st);
        case 64:
            if (a.ntiles >= 4) return launch_cfg<1, 1, 2, 4, 4, 1, 4, 2, 1, 64>(a, st);
            if (a.ntiles >= 2) return launch_cfg<1, 1, 2, 4, 4, 1, 2, 4, 1, 64>(a, st);
            return launch_cfg<1, 1, 2, 4, 4, 1, 1, 4, 1, 64>(a, st);
        case 128:
            if (a.ntiles >= 2) return launch_cfg<1, 1, 2, 4, 4, 1, 2, 4, 1, 128>(a, st);
            return launch_cfg<1, 1, 2, 4, 4, 1, 1, 8, 1, 128>(a, st);
        case 256:
            if (a.ntiles >= 2) return launch_cfg<1, 1, 2, 4, 4, 1, 2, 4, 1, 256>(a, st);
            return launch_cfg<1, 1, 2, 4, 4, 1, 1, 8, 1, 256>(a, st);
        default:
            return SIS3D_EUNSUPPORTED;
        }
    } else if constexpr (S == 2) {
        // k2 s2: the input brick is 2x the output brick per axis -> small output bricks; 8 taps over KW=2/4 waves
        if (a.cin % 32 == 0) {
            if (a.ntiles >= 4) return launch_cfg<2, 2, 2, 4, 4, 1, 4, 2, 1, 32, 12>(a, st);
            if (a.ntiles >= 2) return launch_cfg<2, 2, 2, 4, 4, 1, 2, 4, 1, 32, 12>(a, st);
            return launch_cfg<2, 2, 2, 4, 4, 1, 1, 4, 1, 32, 12>(a, st);
        }
        return launch_cfg<2, 2, 2, 4, 4, 1, 2, 2, 1, 8>(a, st);
    } else {
        // k3 p1: 27 taps over KW=3 waves (9 each)
        if (a.cin % 32) {
            if (a.ntiles >= 2) return launch_cfg<3, 1, 4, 4, 4, 2, 2, 3, 1, 8>(a, st);
            return launch_cfg<3, 1, 4, 4, 4, 2, 1, 3, 1, 8>(a, st);
        }
        if (big) {
            if (a.ntiles >= 2) return launch_cfg<3, 1, 4, 4, 8, 4, 1, 3, 2, 32>(a, st);
            static const int vbig = [] { const char *e = getenv("SIS3D_K3BIG_VARIANT"); return e ? atoi(e) : 0; }();   // tuning hook
            if (vbig == 1) return launch_cfg<3, 1, 4, 4, 4, 2, 1, 3, 1, 32>(a, st);      // 64-voxel bricks, 6 waves, 864 workgroups
            if (vbig == 2) return launch_cfg<3, 1, 2, 4, 4, 1, 1, 3, 1, 32>(a, st);      // 32-voxel bricks, 3 waves, 1728 workgroups
            return launch_cfg<3, 1, 4, 4, 8, 4, 1, 3, 1, 32, 12>(a, st);
        }
        // tuning hook (tools/conv_tune.py): SIS3D_K3_VARIANT selects an alternative tiling for the small-volume k3 layers
        static const int variant = [] { const char *e = getenv("SIS3D_K3_VARIANT"); return e ? atoi(e) : 0; }();
        if (variant == 0 && a.ntiles >= 2 && a.ntiles <= 3 && a.npw == 0) return launch_cfg<3, 1, 2, 4, 4, 1, 2, 3, 1, 32, 12>(a, st);
        if (variant == 0 && a.npw > 0 && a.ntiles == 2) return launch_cfg<3, 1, 2, 4, 4, 1, 2, 3, 1, 32, 12>(a, st);
        if (a.ntiles >= 2) {
            switch (variant) {
            case 1: return launch_cfg<3, 1, 2, 4, 4, 1, 1, 9, 1, 32>(a, st);       // one tile per WG, taps over 9 waves
            case 2: return launch_cfg<3, 1, 2, 4, 4, 1, 2, 3, 1, 32>(a, st);       // 32 vox x 64 cout, 6 waves
            case 3: return launch_cfg<3, 1, 4, 4, 8, 4, 1, 3, 1, 32>(a, st);       // 128 vox x 32 cout, 12 waves (B shared 4x)
            case 4: return launch_cfg<3, 1, 2, 4, 4, 1, 1, 3, 1, 32>(a, st);       // one tile per WG, 3 waves
            default: return launch_cfg<3, 1, 4, 4, 4, 2, 2, 3, 1, 32>(a, st);
            }
        }
        if (variant == 2) return launch_cfg<3, 1, 4, 4, 4, 2, 1, 3, 1, 32>(a, st);
        return launch_cfg<3, 1, 2, 4, 4, 1, 1, 9, 1, 32, 12>(a, st);       // one tile per WG, 27 taps over 9 waves
    }
}

} // namespace

extern "C" size_t sis3d_conv_packed_floats(int cout, int cin, int ksize)
{
    const size_t ntiles = (cout + 31) / 32, kg = (cin + 7) / 8, T = (size_t)ksize * ksize * ksize;
    return ntiles * T * kg * 256;
}

extern "C" int sis3d_conv_pack_weight(const float *w, int cout, int cin, int ksize, float *packed, sis3d_stream_t stream)
{
    if (!w || !packed || cout <= 0 || cin <= 0 || ksize < 1 || ksize > 3) return SIS3D_EINVAL;
    const int T = ksize * ksize * ksize, ntiles = (cout + 31) / 32, kgtot = (cin + 7) / 8;
    const int64_t total = (int64_t)ntiles * T * kgtot * 256;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0,
                       as_stream(stream), w, cout, cin, T, ntiles, kgtot, packed);
    return sis3d_check_launch();
}

extern "C" int sis3d_conv3d(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                            int cout, int ksize, int stride, int flags, const float *residual, int res_stride, float *out,
                            int out_stride, int out_coff, float *out2, float *out3, int anchors, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % 8) || (cin_stride % 4) || cin_stride < cin) return SIS3D_EINVAL;
    if ((flags & SIS3D_EPI_RESIDUAL) && !residual) return SIS3D_EINVAL;
    if ((flags & SIS3D_EPI_RPN_HEAD) && (!out2 || anchors <= 0 || cout != 8 * anchors)) return SIS3D_EINVAL;
    ConvArgs a;
    a.nprob = 1;
    a.npw = 0;
    a.rag = nullptr; a.nrag = 0; a.ragged_blocks = 0;
    a.in = in; a.X = X; a.Y = Y; a.Z = Z; a.cin = cin; a.cin_stride = cin_stride;
    a.wp = packed_w; a.bias = bias; a.cout = cout; a.ntiles = (cout + 31) / 32;
    a.flags = flags; a.res = residual; a.res_stride = res_stride;
    a.out = out; a.out_stride = out_stride; a.out_coff = out_coff; a.out2 = out2; a.out3 = out3; a.anchors = anchors;
    if ((flags & SIS3D_EPI_RPN_HEAD) && 2 * anchors > 32) return SIS3D_EUNSUPPORTED;   // class pairs must share one 32-wide tile
    hipStream_t st = as_stream(stream);
    if (ksize == 1 && stride == 1) { a.OX = X; a.OY = Y; a.OZ = Z; return dispatch<1, 1>(a, st); }
    if (ksize == 3 && stride == 1) { a.OX = X; a.OY = Y; a.OZ = Z; return dispatch<3, 1>(a, st); }
    if (ksize == 2 && stride == 2) { a.OX = X / 2; a.OY = Y / 2; a.OZ = Z / 2; if (!a.OX || !a.OY || !a.OZ) return SIS3D_EINVAL; return dispatch<2, 2>(a, st); }
    return SIS3D_EUNSUPPORTED;
}

static int chain_common(ConvArgs &a, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                        int ksize, int stride, int flags, float *out, int out_stride, int nstages, const sis3d_pw_stage *stages,
                        sis3d_stream_t stream)
{
    if (!packed_w || X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || nstages < 1 || nstages > 2 || !stages) return SIS3D_EINVAL;
    if ((cin % 8) || (cin_stride % 4) || cin_stride < cin || (cout % 32) || cout > 128) return SIS3D_EINVAL;
    if (flags & (SIS3D_EPI_RPN_HEAD | SIS3D_EPI_RESIDUAL)) return SIS3D_EUNSUPPORTED;
    a.nprob = 1;
    a.rag = nullptr; a.nrag = 0; a.ragged_blocks = 0;
    a.npw = nstages;
    int cprev = cout;
    for (int i = 0; i < nstages; ++i) {
        const sis3d_pw_stage &s = stages[i];
        if (!s.packed_w || s.cout <= 0 || (s.cout % 32) || s.cout > 128 || s.cin != cprev) return SIS3D_EINVAL;
        if ((s.flags & SIS3D_EPI_RESIDUAL) && !s.residual) return SIS3D_EINVAL;
        if (!s.out && i + 1 == nstages) return SIS3D_EINVAL;
        a.pw[i].wp = s.packed_w; a.pw[i].bias = s.bias; a.pw[i].res = s.residual; a.pw[i].out = s.out;
        a.pw[i].cout = s.cout; a.pw[i].res_stride = s.res_stride; a.pw[i].out_stride = s.out_stride; a.pw[i].flags = s.flags;
        cprev = s.cout;
    }
    a.X = X; a.Y = Y; a.Z = Z; a.cin = cin; a.cin_stride = cin_stride;
    a.wp = packed_w; a.bias = bias; a.cout = cout; a.ntiles = cout / 32;
    a.flags = flags; a.res = nullptr; a.res_stride = 0;
    a.out = out; a.out_stride = out_stride; a.out_coff = 0; a.out2 = nullptr; a.out3 = nullptr; a.anchors = 0;
    hipStream_t st = as_stream(stream);
    if (ksize == 3 && stride == 1) { a.OX = X; a.OY = Y; a.OZ = Z; return dispatch<3, 1>(a, st); }
    if (ksize == 2 && stride == 2) { a.OX = X / 2; a.OY = Y / 2; a.OZ = Z / 2; if (!a.OX || !a.OY || !a.OZ) return SIS3D_EINVAL; return dispatch<2, 2>(a, st); }
    return SIS3D_EUNSUPPORTED;
}

extern "C" int sis3d_conv3d_chain(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                                  int cout, int ksize, int stride, int flags, float *out, int out_stride, int nstages,
                                  const sis3d_pw_stage *stages, sis3d_stream_t stream)
{
    if (!in) return SIS3D_EINVAL;
    ConvArgs a;
    a.in = in;
    return chain_common(a, X, Y, Z, cin, cin_stride, packed_w, bias, cout, ksize, stride, flags, out, out_stride, nstages, stages, stream);
}

extern "C" int sis3d_conv3d_pw_chain(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                                     int cout, int flags, const float *residual, int res_stride, float *out, int out_stride,
                                     int out_coff, int nstages, const sis3d_pw_stage *stages, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || X <= 0 || Y <= 0 || Z <= 0 || nstages < 0 || nstages > 1) return SIS3D_EINVAL;
    if ((cin != 32 && cin != 64) || (cin_stride % 4) || cin_stride < cin || (cout % 32) || cout > 128) return SIS3D_EUNSUPPORTED;
    if ((flags & SIS3D_EPI_RESIDUAL) && !residual) return SIS3D_EINVAL;
    if (flags & (SIS3D_EPI_RPN_HEAD | SIS3D_EPI_SIGMOID)) return SIS3D_EUNSUPPORTED;
    if (nstages == 0)
        return sis3d_conv3d(in, X, Y, Z, cin, cin_stride, packed_w, bias, cout, 1, 1, flags, residual, res_stride, out, out_stride,
                            out_coff, nullptr, nullptr, 0, stream);
    ConvArgs a;
    a.nprob = 1;
    a.rag = nullptr; a.nrag = 0; a.ragged_blocks = 0;
    a.npw = 1;
    const sis3d_pw_stage &s = stages[0];
    if (!s.packed_w || !s.out || s.cout <= 0 || (s.cout % 32) || s.cout > 128 || s.cin != cout) return SIS3D_EINVAL;
    if (s.flags & SIS3D_EPI_RESIDUAL) return SIS3D_EUNSUPPORTED;
    a.pw[0].wp = s.packed_w; a.pw[0].bias = s.bias; a.pw[0].res = nullptr; a.pw[0].out = s.out;
    a.pw[0].cout = s.cout; a.pw[0].res_stride = 0; a.pw[0].out_stride = s.out_stride; a.pw[0].flags = s.flags;
    a.in = in; a.X = X; a.Y = Y; a.Z = Z; a.cin = cin; a.cin_stride = cin_stride;
    a.wp = packed_w; a.bias = bias; a.cout = cout; a.ntiles = cout / 32;
    a.flags = flags; a.res = residual; a.res_stride = res_stride;
    a.out = out; a.out_stride = out_stride; a.out_coff = out_coff; a.out2 = nullptr; a.out3 = nullptr; a.anchors = 0;
    a.OX = X; a.OY = Y; a.OZ = Z;
    return dispatch<1, 1>(a, as_stream(stream));
}

extern "C" int sis3d_conv3d_chain_projected(const int32_t *vox2pix, const float *feat_rows, int nslots, int64_t npix, int X, int Y, int Z,
                                            int cin, const float *packed_w, const float *bias, int cout, int flags, float *out,
                                            int out_stride, int nstages, const sis3d_pw_stage *stages, sis3d_stream_t stream)
{
    if (!vox2pix || !feat_rows || nslots <= 0 || npix <= 0 || (cin % 32)) return SIS3D_EINVAL;
    ConvArgs a;
    a.in = feat_rows;                                    // never dereferenced as a volume
    a.ptab = vox2pix; a.pfeat = feat_rows; a.pslots = nslots; a.pnpix = npix;
    return chain_common(a, X, Y, Z, cin, cin, packed_w, bias, cout, 2, 2, flags, out, out_stride, nstages, stages, stream);
}

extern "C" int sis3d_conv3d_batched(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                    const float *const *packed_ws, const float *const *biases, int cout, int ksize, int stride,
                                    int flags, const float *const *residuals, int res_stride, float *const *outs, int out_stride,
                                    int out_coff, sis3d_stream_t stream)
{
    if (nprob < 1 || nprob > MAXP || !ins || !packed_ws || !outs) return SIS3D_EINVAL;
    if (X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || (cin % 8) || (cin_stride % 4) || cin_stride < cin) return SIS3D_EINVAL;
    if (flags & SIS3D_EPI_RPN_HEAD) return SIS3D_EUNSUPPORTED;
    ConvArgs a;
    a.npw = 0;
    a.rag = nullptr; a.nrag = 0; a.ragged_blocks = 0;
    a.nprob = nprob;
    for (int p = 0; p < nprob; ++p) {
        if (!ins[p] || !packed_ws[p] || !outs[p]) return SIS3D_EINVAL;
        if ((flags & SIS3D_EPI_RESIDUAL) && (!residuals || !residuals[p])) return SIS3D_EINVAL;
        a.b_in[p] = ins[p]; a.b_wp[p] = packed_ws[p]; a.b_bias[p] = biases ? biases[p] : nullptr;
        a.b_res[p] = residuals ? residuals[p] : nullptr; a.b_out[p] = outs[p]; a.b_out2[p] = nullptr;
    }
    a.in = ins[0]; a.X = X; a.Y = Y; a.Z = Z; a.cin = cin; a.cin_stride = cin_stride;
    a.wp = packed_ws[0]; a.bias = a.b_bias[0]; a.cout = cout; a.ntiles = (cout + 31) / 32;
    a.flags = flags; a.res = a.b_res[0]; a.res_stride = res_stride;
    a.out = outs[0]; a.out_stride = out_stride; a.out_coff = out_coff; a.out2 = nullptr; a.out3 = nullptr; a.anchors = 0;
    hipStream_t st = as_stream(stream);
    if (ksize == 1 && stride == 1) { a.OX = X; a.OY = Y; a.OZ = Z; return dispatch<1, 1>(a, st); }
    if (ksize == 3 && stride == 1) { a.OX = X; a.OY = Y; a.OZ = Z; return dispatch<3, 1>(a, st); }
    if (ksize == 2 && stride == 2) { a.OX = X / 2; a.OY = Y / 2; a.OZ = Z / 2; if (!a.OX || !a.OY || !a.OZ) return SIS3D_EINVAL; return dispatch<2, 2>(a, st); }
    return SIS3D_EUNSUPPORTED;
}

// ---- ragged batches (mask head: one launch for all detected boxes) ---------------------------------------------
struct PlanarDesc {
    int x0, y0, z0;       // window origin in the planar grid
    int dx, dy, dz;       // window (= output) size
    int pad0, pad1;
    int64_t t0;           // first work item (voxel x cout/4) of this problem
    int64_t out_off;      // element offset of this problem's output rows
};

// k3 p1 on windows of the planar 2-channel grid, zero padding at each WINDOW border (what nn.Conv3d on the sliced
// tensor computes, lib/nets/network.py:307-310 + backbones.py:241)
__global__ __launch_bounds__(256) void conv_planar2_ragged_kernel(const float *__restrict__ in, int64_t is_c, int64_t is_x, int64_t is_y,
                                                                  const PlanarDesc *__restrict__ desc, int ndesc, int64_t total,
                                                                  const float *__restrict__ w, int cout, int flags,
                                                                  float *__restrict__ out, int out_stride)
{
    // thread = one voxel x 16 output channels (4 float4 accumulators): the 54 input taps are loaded once per 16 outputs
    // (4x fewer global loads than one float4 per thread); desc[i].t0 counts (voxel, 4-channel) items as the host packs them
    constexpr int KS = 3, T = 27, K = 54, CPT = 16;
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [K][cout]
    for (int i = threadIdx.x; i < K * cout; i += blockDim.x) {
        const int co = i % cout, k = i / cout;
        wl[i] = w[(int64_t)co * K + k];
    }
    __syncthreads();
    const int cq = cout / 4, cg = (cout + CPT - 1) / CPT;          // float4 groups per voxel (host units), 16-channel groups per voxel
    const int64_t nvox_total = total / cq;
    const int64_t items = nvox_total * cg;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < items; t += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(t % cg) * CPT;
        const int64_t gv = t / cg;                                 // voxel index over the packed crops
        const int64_t key = gv * cq;
        int lo = 0, hi = ndesc - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (desc[mid].t0 <= key) lo = mid; else hi = mid - 1;
        }
        const PlanarDesc d = desc[lo];
        const int64_t v = gv - d.t0 / cq;
        const int oz = (int)(v % d.dz), oy = (int)((v / d.dz) % d.dy), ox = (int)(v / ((int64_t)d.dz * d.dy));
        float4 acc[CPT / 4];
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // one (channel, dx) slice of 9 taps per batch: loads in flight together (clamped addresses, padding as a select)
#pragma unroll
        for (int b = 0; b < 2 * KS; ++b) {
            float xs[KS * KS];
#pragma unroll
            for (int u = 0; u < KS * KS; ++u) {
                const int ci = b / KS, ddx = b % KS, ddy = u / KS, ddz = u % KS;
                const int wx = ox + ddx - 1, wy = oy + ddy - 1, wz = oz + ddz - 1;
                const bool ok = wx >= 0 && wx < d.dx && wy >= 0 && wy < d.dy && wz >= 0 && wz < d.dz;
                const int cx = min(max(wx, 0), d.dx - 1), cy = min(max(wy, 0), d.dy - 1), cz = min(max(wz, 0), d.dz - 1);
                const float t = in[ci * is_c + (int64_t)(d.x0 + cx) * is_x + (int64_t)(d.y0 + cy) * is_y + (d.z0 + cz)];
                xs[u] = ok ? t : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KS * KS; ++u) {
                const float xv = xs[u];
                const float *wr = wl + (b * KS * KS + u) * cout + c0;
#pragma unroll
                for (int j = 0; j < CPT / 4; ++j) {
                    if (c0 + 4 * j >= cout) break;
                    const float4 wv = *reinterpret_cast<const float4 *>(wr + 4 * j);
                    acc[j].x = fmaf(xv, wv.x, acc[j].x); acc[j].y = fmaf(xv, wv.y, acc[j].y);
                    acc[j].z = fmaf(xv, wv.z, acc[j].z); acc[j].w = fmaf(xv, wv.w, acc[j].w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) {
            if (c0 + 4 * j >= cout) break;
            float4 r = acc[j];
            if (flags & SIS3D_EPI_RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            *reinterpret_cast<float4 *>(out + d.out_off + v * out_stride + c0 + 4 * j) = r;
        }
    }
}

// r6: the same layer on the matrix pipe, for the mask head's 64 output channels.  The 54 taps of a voxel are the K dimension (padded to
// 56 = 14 MFMAs of K=4), a wave owns 16 consecutive packed voxels x all 64 couts: D^T[cout][voxel] = W[cout][k] * X^T[k][voxel], the
// transposed tile GEMM of mfma16.h, so a lane ends with 4 consecutive couts of one voxel (one 16 B store per cout tile).  A lane
// (voxel l16, kq) gathers 14 taps instead of the 54 of the FMA kernel above, and the weights sit in 56 registers per lane (read from
// LDS once per wave) instead of one ds_read_b128 per 4 FMAs -- that LDS traffic, not the FMAs, bounded the kernel above (22.2 us for
// the 16-box batch of bench.py --workload detect --masks; this one: profiles/r06_mask_head_first_last_layer_ab.txt).
constexpr int PL_MAXD = 128;
__global__ __launch_bounds__(256) void conv_planar2_ragged_mfma_kernel(const float *__restrict__ in, int64_t is_c, int64_t is_x, int64_t is_y,
                                                                       const PlanarDesc *__restrict__ desc, int ndesc, int64_t nvox_total,
                                                                       const float *__restrict__ w, int flags, float *__restrict__ out,
                                                                       int out_stride)
{
    constexpr int K = 54, KK = 14, CO = 64, WS = CO + 1;          // WS: padded row of the LDS weight table (k-major writes without conflicts)
    __shared__ float wl[4 * KK * WS];
    __shared__ PlanarDesc dl[PL_MAXD];
    __shared__ int2 tap[4 * KK];                                   // tap k: its 32-bit offset from the voxel's own cell, (ddx, ddy, ddz) packed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, q = lane >> 4;
    if (tid < 4 * KK) {
        const int kc = tid < K ? tid : K - 1;                      // k = 54, 55: zero weights, any valid tap
        const int r = kc % 27, ddx = r / 9, ddy = (r / 3) % 3, ddz = r % 3;
        tap[tid] = make_int2((kc / 27) * (int)is_c + (ddx - 1) * (int)is_x + (ddy - 1) * (int)is_y + (ddz - 1), ddx | (ddy << 2) | (ddz << 4));
    }
    for (int i = tid; i < CO * 4 * KK; i += 256) {                 // checkpoint layout (Cout, 2, 3, 3, 3): rows of 54, read along k
        const int co = i / (4 * KK), k = i % (4 * KK);
        wl[k * WS + co] = k < K ? w[co * K + k] : 0.0f;
    }
    const bool dlds = ndesc <= PL_MAXD;
    if (dlds) for (int i = tid; i < ndesc; i += 256) dl[i] = desc[i];
    __syncthreads();
    const int64_t ntiles = (nvox_total + 15) / 16;
    // decode the lane's voxel (32-bit arithmetic: the launcher checks the sizes) and issue its 14 tap loads
    auto gather = [&](int64_t tile, float (&x)[KK], int64_t &ooff, bool &live) {
        const int64_t gv0 = tile * 16 + l16;
        live = gv0 < nvox_total;
        const int64_t gv = live ? gv0 : nvox_total - 1;
        const int64_t key = gv * (CO / 4);                        // desc[i].t0 counts (voxel, 4-channel) items as the host packs them
        int lo = 0, hi = ndesc - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const int64_t t0 = dlds ? dl[mid].t0 : desc[mid].t0;
            if (t0 <= key) lo = mid; else hi = mid - 1;
        }
        const PlanarDesc d = dlds ? dl[lo] : desc[lo];
        const unsigned v = (unsigned)(gv - d.t0 / (CO / 4));
        const unsigned t = v / (unsigned)d.dz, uox = t / (unsigned)d.dy;
        const int oz = (int)(v - t * (unsigned)d.dz), oy = (int)(t - uox * (unsigned)d.dy), ox = (int)uox;
        ooff = d.out_off + (int64_t)v * out_stride;
        const float *centre = in + ((d.x0 + ox) * (int)is_x + (d.y0 + oy) * (int)is_y + (d.z0 + oz));
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int2 tp = tap[4 * kk + q];
            const unsigned wx = (unsigned)(ox + (tp.y & 3) - 1), wy = (unsigned)(oy + ((tp.y >> 2) & 3) - 1), wz = (unsigned)(oz + (tp.y >> 4) - 1);
            const bool ok = wx < (unsigned)d.dx && wy < (unsigned)d.dy && wz < (unsigned)d.dz;          // zero padding at the WINDOW border
            const float tv = centre[ok ? tp.x : 0];           // a padded tap reads the voxel's own cell (always valid) and drops it
            x[kk] = ok ? tv : 0.0f;
        }
    };
    float x[KK];
    int64_t ooff;
    bool live;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntiles) gather(tile, x, ooff, live);                // in flight while the weights come out of LDS
    float wr[KK][4];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int n = 0; n < 4; ++n) wr[kk][n] = wl[(4 * kk + q) * WS + 16 * n + l16];
    while (tile < ntiles) {
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kk][n], x[kk], acc[n], 0, 0, 0);
        float *o = out + ooff + 4 * q;
        const bool st = live;
        tile += (int64_t)gridDim.x * 4;
        if (tile < ntiles) gather(tile, x, ooff, live);            // the next tile's taps fly under the stores
        if (st) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float4 r = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
                *reinterpret_cast<float4 *>(o + 16 * n) = relu4(r, flags & SIS3D_EPI_RELU);
            }
        }
    }
}

extern "C" int sis3d_ragged_tiling(int cin, int cout, int ksize, int *bx, int *by, int *bz, int *ngroups)
{
    if (!bx || !by || !bz || !ngroups) return SIS3D_EINVAL;
    const int ntiles = (cout + 31) / 32;
    *bx = 2; *by = 4; *bz = 4;
    if (ksize == 3 && cin % 32 == 0 && ntiles <= 2) { *ngroups = 1; return SIS3D_OK; }
    if (ksize == 1 && cin == 64 && ntiles == 1) { *ngroups = 1; return SIS3D_OK; }
    return SIS3D_EUNSUPPORTED;
}

extern "C" int sis3d_conv3d_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                   int ksize, int flags, float *out, int out_stride, const void *desc_dev, int ndesc,
                                   int64_t total_blocks, sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || !desc_dev || ndesc <= 0 || total_blocks <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % 8) || (cin_stride % 4) || cin_stride < cin) return SIS3D_EINVAL;
    if (flags & (SIS3D_EPI_RPN_HEAD | SIS3D_EPI_RESIDUAL)) return SIS3D_EUNSUPPORTED;
    int bx, by, bz, ng;
    int rc = sis3d_ragged_tiling(cin, cout, ksize, &bx, &by, &bz, &ng);
    if (rc) return rc;
    ConvArgs a;
    a.nprob = 1; a.npw = 0;
    a.rag = (const RaggedDesc *)desc_dev; a.nrag = ndesc;
    a.in = in; a.X = a.Y = a.Z = a.OX = a.OY = a.OZ = 1; a.cin = cin; a.cin_stride = cin_stride;
    a.wp = packed_w; a.bias = bias; a.cout = cout; a.ntiles = (cout + 31) / 32;
    a.flags = flags; a.res = nullptr; a.res_stride = 0;
    a.out = out; a.out_stride = out_stride; a.out_coff = 0; a.out2 = nullptr; a.out3 = nullptr; a.anchors = 0;
    a.ragged_blocks = total_blocks;
    hipStream_t st = as_stream(stream);
    if (ksize == 3) return launch_cfg<3, 1, 2, 4, 4, 1, 2, 3, 1, 32>(a, st);
    return launch_cfg<1, 1, 2, 4, 4, 1, 1, 4, 1, 64>(a, st);
}

extern "C" int sis3d_conv3d_planar2_ragged(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, const void *desc_dev, int ndesc,
                                           int64_t total_items, const float *w, int cout, int flags, float *out, int out_stride,
                                           sis3d_stream_t stream)
{
    if (!in || !w || !out || !desc_dev || ndesc <= 0 || total_items <= 0 || cout <= 0 || (cout % 4) || (out_stride % 4)) return SIS3D_EINVAL;
    static const bool fma_only = [] { const char *e = getenv("SIS3D_PLANAR_FMA"); return e && atoi(e) != 0; }();      // A/B hook
    if (cout == 64 && !fma_only && 2 * is_c < 0x7fffffff && is_x < is_c && is_y < is_c && total_items / 16 < 0x7fffffff) {     // 32-bit offsets
        const int64_t nvox = total_items / 16, tiles = (nvox + 15) / 16;     // a wave per 16-voxel tile
        const unsigned blocks = (unsigned)((tiles + 3) / 4 < 768 ? (tiles + 3) / 4 : 768);       // 3 waves per SIMD (VGPRs)
        hipLaunchKernelGGL(conv_planar2_ragged_mfma_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, is_c, is_x, is_y,
                           (const PlanarDesc *)desc_dev, ndesc, nvox, w, flags, out, out_stride);
        return sis3d_check_launch();
    }
    const int64_t items = total_items / (cout / 4) * ((cout + 15) / 16);     // threads: one per (voxel, 16 output channels)
    const unsigned blocks = (unsigned)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(conv_planar2_ragged_kernel, dim3(blocks), dim3(256), sizeof(float) * 54 * cout, as_stream(stream), in, is_c, is_x,
                       is_y, (const PlanarDesc *)desc_dev, ndesc, total_items, w, cout, flags, out, out_stride);
    return sis3d_check_launch();
}

extern "C" int sis3d_conv3d_planar2(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, int X, int Y, int Z, int x0, int y0,
                                    int z0, int OX, int OY, int OZ, const float *w, int cout, int ksize, int flags, float *out,
                                    int out_stride, sis3d_stream_t stream)
{
    if (!in || !w || !out || OX <= 0 || OY <= 0 || OZ <= 0 || cout <= 0 || (cout % 4) || out_stride < cout || (out_stride % 4))
        return SIS3D_EINVAL;
    const int S = ksize == 2 ? 2 : 1;
    const int WX = OX * S, WY = OY * S, WZ = OZ * S;                 // window of the grid this call reads
    if (x0 < 0 || y0 < 0 || z0 < 0 || x0 + WX > X || y0 + WY > Y || z0 + WZ > Z) return SIS3D_EINVAL;
    const int64_t total = (int64_t)OX * OY * OZ * ((cout + 15) / 16);       // one thread per (voxel, 16 output channels)
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t st = as_stream(stream);
    if (ksize == 2) {
        hipLaunchKernelGGL((conv_planar2_kernel<2, 2>), dim3(blocks), dim3(256), sizeof(float) * 16 * cout, st, in, is_c, is_x, is_y,
                           x0, y0, z0, WX, WY, WZ, OX, OY, OZ, w, cout, flags, out, out_stride);
    } else if (ksize == 3) {
        hipLaunchKernelGGL((conv_planar2_kernel<3, 1>), dim3(blocks), dim3(256), sizeof(float) * 54 * cout, st, in, is_c, is_x, is_y,
                           x0, y0, z0, WX, WY, WZ, OX, OY, OZ, w, cout, flags, out, out_stride);
    } else {
        return SIS3D_EUNSUPPORTED;
    }
    return sis3d_check_launch();
}
