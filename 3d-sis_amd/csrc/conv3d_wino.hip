// k3 / pad-1 3D convolution by the Winograd minimal-filtering algorithm F(2x2x2, 3x3x3) in EXACT fp32, fused in one kernel
// for gfx950 (CDNA4).
//
// Replaces the same cuDNN calls as conv3d_t16.hip -- nn.Conv3d(C, C', 3, padding=1) + bias + ReLU of
// lib/nets/backbones.py:20-22,188-231 (Bottleneck.conv2, geometry2[0]) and lib/nets/network.py:40 (rpn_net_level*) -- with
// 3.375x fewer multiplications: a 2x2x2 block of outputs costs 64 products per (cin, cout) pair instead of 8 * 27 = 216.
// Every operation is an IEEE binary32 add or an fp32 MFMA (an fmaf chain); the transform matrices hold only 0, +-1, +-1/2, so
// the result differs from a direct fp32 convolution by summation order / association only (measured on the rpn_net layer:
// 2.7e-6 max abs error against a float64 convolution, the direct oneDNN fp32 convolution 2.0e-6; tolerance of the path: 1e-4).
// This is the algorithm class cuDNN -- the library under the reference's nn.Conv3d on a GPU -- picks for 3x3 filters itself.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      (nested over x, y, z)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// In the transformed domain the layer is 64 independent GEMMs  M_xi[tile][cout] = V_xi[tile][cin] * U_xi[cin][cout]
// (xi = the 4x4x4 positions of a transformed input tile, tile = a 2x2x2 output block).  Mapping on the chip:
//   * v_mfma_f32_16x16x4_f32: A = V_xi (16 tiles x 4 channels), B = U_xi (4 channels x 16 couts), one accumulator tile per xi.
//     All xi of a (tile, cout) pair must stay live over the whole channel loop (64 x 4 registers) and every MFMA needs a fresh A
//     and a fresh B operand.  A wave therefore takes HALF of the xi (xi_x in {2h, 2h+1}) for 16 tiles and NC = 2 (or 1) cout tiles:
//     128 NC accumulator registers (the AGPR half of the file), one wave per SIMD, and every transformed input value feeds NC MFMAs.
//   * workgroup = 4 waves (h, g) = a block of 4 x 2 x 4 Winograd tiles (8 x 4 x 8 output voxels; g = which 16 tiles) x NC cout
//     tiles.  rpn_net 128 -> 256 on 24 x 12 x 24: 27 blocks x 8 cout pairs = 216 workgroups, one per CU; geometry2[0] 128 -> 128:
//     NC = 1, 27 x 8 = 216.  sis3d_conv3d_k3wino_prefer says which layers have enough work items; the rest stay on conv3d_t16.hip.
//   * V is never materialised: the raw halo brick (10 x 6 x 10 voxels) of the 4 channels of a K-step is staged in LDS in a
//     planar layout [channel][x][y][z] whose strides (16, 100, 1040 floats) put the 32 lanes of a ds_read_b64 group on 32
//     distinct bank pairs; a lane (tile, channel) reads the 3 x-planes its xi half needs (24 x 8 B) and transforms them in
//     registers (96 adds) into the A operands of its 32 NC MFMAs -- for the NEXT step, as 28 small units issued one behind each
//     MFMA of the current step (struct NextV), ping-ponging between two register sets.
//   * U (the transformed weights, packed once: [cout tile][K-step][xi / 4][lane][4]) streams through a three-deep LDS ring filled
//     by LDS-DMA (global_load_lds_dwordx4: no staging registers) two steps ahead; a lane's B operands of a K-step are 8 NC x
//     ds_read_b128 through a four-quad register ring.
//   * gfx950 does not overlap a wave's fp32 MFMAs with its other instructions (measured: 36.7 cycles per MFMA + 4 per other
//     instruction), so the loop is built to issue as few as possible: addresses are uniform bases + fixed lane offsets, no branch
//     and no select in the block, the one barrier per step sits inside the MFMA block, every wait is counted.
//   * epilogue: output transform in registers (per lane: 32 xi -> 8 partial outputs per tile and cout tile), the two xi halves
//     meet through LDS (each wave finishes two of its four rows), + bias, ReLU, a wave-local LDS transpose, 16 B stores.
// The weight transform (G g G^T per axis) runs once at pack time in fp32.
#include "common.h"
#include "mfma16.h"
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// timing experiments (tools/wino_bench.cpp builds variants; results are WRONG with any bit set): 1 = no U fill, 2 = no raw staging,
// 4 = no barrier in the loop, 16 = no input transform, 32 = no output stores, 64 = phase timestamps (100 MHz) instead of the output,
// 128 = input transform without its packed adds, 256 = input transform without its LDS reads
#ifndef WN_EXP
#define WN_EXP 0
#endif
// XCD work order of the eight-cout-group layers: 1 = two groups x half of the blocks per XCD, 0 = one group x all blocks (r3)
#ifndef WN_XCD_PAIRS
#define WN_XCD_PAIRS 1
#endif
static constexpr bool XCD_PAIRS = WN_XCD_PAIRS != 0;

namespace {

constexpr int WN_MAXP = 4;
constexpr int TXB = 4, TYB = 2, TZB = 4;                        // Winograd tiles per workgroup block
constexpr int VX = 2 * TXB, VY = 2 * TYB, VZ = 2 * TZB;         // output voxels per block: 8 x 4 x 8
constexpr int HX = VX + 2, HY = VY + 2, HZ = VZ + 2;            // halo brick 10 x 6 x 10
constexpr int HZS = 16;                                         // LDS row stride (floats)
constexpr int PS = 100;                                         // x-plane stride: >= HY * HZS, == 4 (mod 32)
constexpr int CHS = 1040;                                       // channel stride: >= HX * PS, == 16 (mod 64)
constexpr int RAW_STAGE = 4 * CHS;                              // floats of one K-step's raw brick (4 channels)
constexpr int NRAW = 3;                                         // raw ring depth: step k + 1 is transformed during step k, k + 2 is being staged
constexpr int B_TILE = 64 * 64;                                 // floats of one (cout tile, K-step) block of U: [16][64][4]
constexpr int NBST = 3;                                         // U ring depth: LDS-DMA lands ~1 us after issue -> two steps ahead
constexpr int lds_floats(int nc) { return NRAW * RAW_STAGE + NBST * nc * B_TILE; }    // NC = 2: 37,056 floats = 148,224 B; NC = 1: 98 KB
constexpr int NVOX = HX * HY * HZ;                              // 600 staging items (voxel x 4 channels)
constexpr int NTHR = 256;                                        // 4 waves, one per SIMD
constexpr int NIT = (NVOX + NTHR - 1) / NTHR;

// r4 -- MINI geometry (the ragged mask-head launches): a workgroup's 32 Winograd tiles are FOUR independent 2x2x2-tile bricks
// ("minis": 4 x 4 x 4 output voxels, halo 6 x 6 x 6) instead of one 4x2x4-tile block, so the crops of the detected boxes
// (9..20 voxels per side) are covered with 4-voxel granularity on every axis: the 16-box bench set needs 924 minis = 231 workgroups
// per cout group where the 8 x 4 x 8 blocks need 304 (54 % of their voxel slots filled) -- 462 instead of 608 workgroups per layer
// = two rounds of the chip instead of three.  LDS layout of a K-step's raw stage: [channel][mini][hx][hy][hz] with strides
// (1352, 324, 48, 8, 1) floats: lane (tile = (mini bit, tx, ty, tz), channel kq) reads 8 B at float offset
// kq 1352 + mb 324 + 2 tx 48 + 2 ty 8 + 2 tz, i.e. bank pair (4 kq + 2 mb + 16 tx + 8 ty + tz) mod 32 -- the 32 lanes of a
// ds_read_b64 group on 32 distinct bank pairs, as in the block layout.  864 staging items (4 per thread instead of 3).
template <bool MINI> struct Geo;
template <> struct Geo<false> {
    static constexpr int hzs = HZS, ps = PS, chs = CHS, nvox = NVOX, nit = NIT, dump = HX * PS, ms = 0;
};
template <> struct Geo<true> {
    static constexpr int hzs = 8, ps = 48, ms = 324, chs = 1352, nvox = 4 * 216, nit = (4 * 216 + NTHR - 1) / NTHR, dump = 4 * 324;
};
template <bool MINI> constexpr int raw_stage() { return 4 * Geo<MINI>::chs; }
template <bool MINI> constexpr int lds_floats_g(int nc) { return NRAW * raw_stage<MINI>() + NBST * nc * B_TILE; }     // MINI, NC = 2: 163,200 B
static_assert(lds_floats_g<true>(2) * 4 <= 160 * 1024 && lds_floats_g<false>(2) == lds_floats(2), "LDS budget");

// ragged batch (the mask head: one launch per layer for all detected boxes' crops): problems of different grid sizes packed back to
// back in one activation buffer; same descriptor layout as conv3d_t16.hip / conv3d.hip's ragged launches (ops.MaskPlan)
struct WinoRagged {
    int X, Y, Z;
    int nbx, nby, nbz;         // 8 x 4 x 8 blocks per axis
    int block0;                // first work item (block x cout group) of this problem
    int pad;
    int64_t in_off, out_off;   // element offsets of this problem's activations inside the packed in / out buffers
};

struct WinoArgs {
    const float *in[WN_MAXP];
    const float *wp[WN_MAXP];
    const float *bias[WN_MAXP];
    float *out[WN_MAXP];
    int X, Y, Z;
    int cin_stride;
    int cout, ngroups;         // ngroups = ceil(ceil(cout/16) / NC): cout-tile groups, one per workgroup
    int nk;                    // cin / 4
    int flags;
    int out_stride, out_coff;
    int nbx, nby, nbz;
    const WinoRagged *rag;
    int nrag;
    // fused Bottleneck tail (template C3 > 0; lib/nets/backbones.py:33-40, 29-31): the k3 conv is Bottleneck.conv2 and its ReLU'd tile
    // goes through conv3 (1x1x1, C3 couts) + bias + residual + ReLU -> tout, and (C2N > 0) the NEXT block's conv1 + bias + ReLU -> out2
    const float *w3p, *b3;     // conv3: pw16-packed weights [C3/16][cout/16][64][4], bias
    const float *res;          // the block input (residual), channels-last rows of res_stride floats
    int res_stride;
    float *tout;
    int tout_stride, tout_coff;
    const float *w1n, *b1n;    // next conv1: pw16-packed [C2N/16][C3/16][64][4], bias
    float *out2;
    int out2_stride;
    // r5 -- PIGGYBACK upload (template PIGGY; sis3d_conv3d_k3wino_piggyback): the launch has one more ROW of workgroups than it has
    // problems (blockIdx.y == pig_row).  That row does not convolve: its first eight workgroups pull the pipeline's NEXT chunk (its
    // address sits in the mailbox slot the graph's first node fetched: pig_state[16..17], include/sis3d.h) into pig_dst while the real
    // workgroups compute, the others leave at once
    unsigned *pig_state;
    float *pig_dst;
    long long pig_n4;
    int pig_row;
};

// OFF: immediate byte offset of the instruction, added to BOTH the global and the LDS address (< 4096)
template <int OFF = 0>
__device__ __forceinline__ void glds16(const float *gsrc, float *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                     (void __attribute__((address_space(3))) *)lds_dst, 16, OFF, 0);
}

// fp32 MFMA with its accumulator pinned in AGPRs ("+a"): left to the register allocator, the 256 accumulator registers of the NC = 2
// kernel wander between the VGPR and AGPR halves whenever the loop body changes (dozens of v_accvgpr moves per step).  hipcc's hazard
// recogniser does not see inside asm: an accumulator is touched again 64 MFMAs later at the earliest, and a barrier stands between
// the loop and the epilogue's reads.
__device__ __forceinline__ void mfma_agpr(f32x4 &acc, float a, float b)
{
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

typedef const __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(3))) f32x2 lds_f32x2;
// byte address of a lane's raw rows inside the LDS (for the ds_read asm)
struct LdsRow {
    unsigned lo;
    __device__ __forceinline__ explicit LdsRow(const float *p) : lo((unsigned)(size_t)(lds_f32 *)p) {}
};
template <int OFF>
__device__ __forceinline__ void ds_read_b64_asm(f32x2 &dst, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// raw-stage loads and their counted waits as inline asm (see wino_wave): free functions, because clang rejects asm operands that
// name captured variables inside a generic lambda
__device__ __forceinline__ void load16_asm(f32x4 &dst, int byte_off, const float *base)
{
    // the base is uniform, but under SGPR pressure (the C3 = 128 Bottleneck tail keeps a dozen kernel-argument pointers live) hipcc parks
    // it in VGPRs and then hands the asm a VGPR pair for its "s" operand (an assembler error, not a readfirstlane); ask for the scalar
    // copy explicitly -- folded away wherever the value already sits in SGPRs.
    // r6 -- the "s_nop 4": an SGPR written by a VALU instruction (v_readfirstlane) must not be read as a VMEM address within the next
    // 5 wait states (gfx9 hazard).  hipcc pads that for instructions it knows, NOT for inline asm: with the readfirstlane pair one or two
    // instructions in front of the load, the load went out with a STALE low half of the base ("Memory access fault" at hi << 32 + offset
    // in every build of this kernel whose base needed the readfirstlane right there -- an offset kept in VGPRs, a spill-heavy build:
    // r5's unexplained faults of the piggyback variants; found with tools/wino_bench.cpp variants, r6).  Between MFMAs the five idle
    // issue cycles are free.
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const float *sbase = (const float *)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt(f32x4 &a, f32x4 &b, f32x4 &c)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt(f32x4 &a, f32x4 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_all(f32x4 (&v)[2]) { wait_vmcnt<N>(v[0], v[1]); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_all(f32x4 (&v)[3]) { wait_vmcnt<N>(v[0], v[1], v[2]); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_all(f32x4 (&v)[4]) { wait_vmcnt<N>(v[0], v[1], v[2], v[3]); }

// 1-D input transform B^T over 4 values, in place
#define WN_BT_INPLACE(v0, v1, v2, v3) \
    do { const float t0_ = (v0) - (v2), t1_ = (v1) + (v2), t2_ = (v2) - (v1), t3_ = (v1) - (v3); \
         v0 = t0_; v1 = t1_; v2 = t2_; v3 = t3_; } while (0)

// hipcc sinks pure arithmetic to its first use, i.e. below the MFMAs it is meant to hide behind (and a sched_barrier only pins
// the machine scheduler, not IR-level code motion): an empty volatile asm that "modifies" a unit's results keeps the unit
// where it is written
#define WN_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// packed fp32 adds (two IEEE binary32 adds per issue slot; a - b is a + (-b) exactly, so the results are those of the scalar adds).
// Inline asm: left to itself hipcc either does not pair the adds or pairs them with v_mov shuffles that cost more than they save;
// here every operand is a natural even-aligned pair (two consecutive z of one LDS row) and the z pass picks halves with op_sel.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 r;
    if constexpr (WN_EXP & 128) return a;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b)
{
    f32x2 r;
    if constexpr (WN_EXP & 128) return a;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The same operations for the epilogue: NOT volatile -- the output transform is a tree of short dependent chains, and a lone wave
// stalls on every back-to-back dependent pair unless hipcc is free to interleave the chains (volatile asm keeps program order)
__device__ __forceinline__ f32x2 epk_add(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 epk_sub(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// - a - b  =  (-a) + (-b), one rounding: the scalar code's  -a - b
__device__ __forceinline__ f32x2 epk_nsub(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a 16-byte store at (uniform base + 32-bit byte offset): asm, so that hipcc neither splits it nor folds it into the scalar fallback
__device__ __forceinline__ void store16_asm(f32x4 v, unsigned byte_off, float *base)
{
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    float *sbase = (float *)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(sbase) : "memory");      // s_nop: see load16_asm
}
// B^T along z on a row held as a = (t0, t1), b = (t2, t3):  (t0 - t2, t1 + t2)  and  (t2 - t1, t1 - t3)
__device__ __forceinline__ f32x2 pk_bt_lo(f32x2 a, f32x2 b)
{
    f32x2 r;
    if constexpr (WN_EXP & 128) return a;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_bt_hi(f32x2 a, f32x2 b)
{
    f32x2 r;
    if constexpr (WN_EXP & 128) return a;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// r6 -- A [slot][4 channels] RAW BRICK (what an LDS-DMA staging of the channels-last rows leaves in LDS; z pairs read with ds_read2_b32)
// was built at the end of the round, with LDS-DMA staging and with the register staging below, and is not in: bit-identical, 18-35 %
// slower either way -- the LDS serves ds_read2 as 16-lane groups on banks mod 32, where 16-byte slots leave the 16 tiles only 8 distinct
// banks (profiles/r06_raw_staging_dma_probe.txt).  The channel-major brick and its conflict-free ds_read_b64 stay.
// r6 -- a PERSISTENT WORK LOOP (a launch of fewer workgroups than work items, workgroup b taking items b, b + G, ...: VERDICT r5 item 2)
// was built and measured and is not in: same binary, same box, items strided over 216 workgroups against one workgroup per item
// (profiles/r06_wino_persistent_loop_ab.txt): rpn_net pair 96.8 against 95.7 us, four problems 177.9 / 178.6, 32 -> 32 @48x24x48 x2
// 36.1 / 36.7, x4 68.3 / 70.0 -- a wash (and the loop costs 13 more spilled registers).  The "~1.1 us between a workgroup's last store
// and its successor's first instruction" that suggested it is the same with the successor INSIDE the workgroup: it is the item
// decode (three integer divisions through v_rcp + readfirstlane), not a dispatch cost.  What would pay is the next item's loads in
// flight under this item's output transform; that needs the staging plan of two items live at once on a 512-register kernel.

// The input transform of the NEXT K-step as 28 units, issued behind MFMAs of the current step.  fp32 MFMA and fp32 VALU share
// the SIMD's fp32 lanes on gfx950 (their cycles ADD: measured, tools/wino_bench.cpp), so the transform is not hidden -- it is
// kept small instead: a wave transforms HALF of the xi (xi_x in {2H, 2H + 1}: 96 adds = 48 packed adds, + 24 LDS reads) and uses
// every transformed value for TWO cout tiles.  Everything is held as z PAIRS (the two halves of one ds_read_b64):
//   read_half_row (8):  half of raw row dy of the three x-planes this half needs: 3 x ds_read_b64, two rows in flight at most
//   x_pair   (8):  (dy, z pair) over x                2 packed adds
//   y_col    (4):  (i, z pair) over y                 4 packed adds, in place
//   z_row    (8):  row (i, xi_y) over z               2 packed adds (op_sel)  -> T[i][xi_y][z pair] = the A operands of the next step
template <int H, bool MINI = false>
struct NextV {
    f32x2 T[2][4][2];                                           // [xi_x - 2H][xi_y][xi_z pair]
    f32x2 d[2][3][2];                                           // [dy & 1][plane][lo / hi]

    __device__ __forceinline__ float a_operand(int i, int y, int e) const { return e & 1 ? T[i][y][e >> 1].y : T[i][y][e >> 1].x; }

    // One ds_read_b64 per z pair as inline asm, NOT the ds_read2_b64 hipcc fuses any two 8-byte reads off one base into: the LDS serves
    // ds_read2_b64 at a quarter of the rate (two accesses x four 16-lane groups on banks mod 32, where the ty lanes of a group
    // collide: 16 cycles per instruction, and the U reads queue behind it), a plain ds_read_b64 as two 32-lane groups on banks mod
    // 64, where this lane layout (2 tz + 8 txl + 32 ty + 16 kq) is conflict-free: 2 cycles.  hipcc does not count asm reads: the
    // bursts that consume the rows start with wait_rows().
    // half a row: reads 3 J .. 3 J + 2 of the six (plane p = read / 2, z pair = read & 1).  Three per MFMA gap: a gap hides ~32 cycles
    // of issue, and twelve reads issued back to back cost the matrix pipe ~85 ns per step
    template <int DY, int J>
    __device__ __forceinline__ void read_half_row(const LdsRow &r)
    {
        if constexpr (WN_EXP & 256) return;
        static_for<3 * J, 3 * J + 3>([&](auto R) {
            constexpr int p = decltype(R)::value >> 1, h = decltype(R)::value & 1, dx = p + H;
            ds_read_b64_asm<(dx * Geo<MINI>::ps + DY * Geo<MINI>::hzs + 2 * h) * 4>(d[DY & 1][p][h], r.lo);
        });
    }
    __device__ __forceinline__ void wait_rows()
    {
        // volatile asm statements keep their order: the reads stay in front of this wait, the packed adds (the only consumers of the
        // rows) behind it
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    template <int DY, int ZP>
    __device__ __forceinline__ void x_pair()
    {
        const f32x2 p0 = d[DY & 1][0][ZP], p1 = d[DY & 1][1][ZP], p2 = d[DY & 1][2][ZP];
        if constexpr (H == 0) {
            T[0][DY][ZP] = pk_sub(p0, p2);                       // xi_x = 0: d0 - d2
            T[1][DY][ZP] = pk_add(p1, p2);                       // xi_x = 1: d1 + d2
        } else {
            T[0][DY][ZP] = pk_sub(p1, p0);                       // xi_x = 2: d2 - d1   (planes held: 1, 2, 3)
            T[1][DY][ZP] = pk_sub(p0, p2);                       // xi_x = 3: d1 - d3
        }
    }
    template <int I, int ZP>
    __device__ __forceinline__ void y_col()
    {
        const f32x2 v0 = T[I][0][ZP], v1 = T[I][1][ZP], v2 = T[I][2][ZP], v3 = T[I][3][ZP];
        T[I][0][ZP] = pk_sub(v0, v2);
        T[I][1][ZP] = pk_add(v1, v2);
        T[I][2][ZP] = pk_sub(v2, v1);
        T[I][3][ZP] = pk_sub(v1, v3);
    }
    template <int I, int Y>
    __device__ __forceinline__ void z_row()
    {
        const f32x2 lo = pk_bt_lo(T[I][Y][0], T[I][Y][1]);
        T[I][Y][1] = pk_bt_hi(T[I][Y][0], T[I][Y][1]);
        T[I][Y][0] = lo;
    }
    // Unit M rides behind MFMA slot M (0..63) of the current step.  The packed adds go out in THREE BURSTS, not one or two per
    // MFMA: an fp32 VALU instruction between two MFMAs costs the matrix pipe ~11 cycles when it stands alone and 4 when it follows
    // another VALU instruction (tools/issue_overlap.hip: 48 packed adds cost 223 ns spread one per gap, 80 ns in bursts of >= 8),
    // while LDS reads, LDS writes and scalar instructions between MFMAs are free.
    template <int M>
    __device__ __forceinline__ void unit(const LdsRow &r)
    {
        // every burst sits in the LAST slot of a quad of MFMAs: the U reads of the quad were issued at its start, seven MFMAs earlier,
        // so the wait for the raw rows in front of a burst (hipcc emits lgkmcnt(0), not a counted wait) finds nothing young in flight
        if constexpr (M >= 1 && M <= 4) {                       // behind MFMAs 1..4 of a quad whose wait stood in front of MFMA 0
            this->template read_half_row<((M - 1) >> 1), ((M - 1) & 1)>(r);
        } else if constexpr (M == 15) {
            wait_rows();
            x_pair<0, 0>(); x_pair<0, 1>(); x_pair<1, 0>(); x_pair<1, 1>();
        } else if constexpr (M >= 17 && M <= 20) {
            this->template read_half_row<(2 + ((M - 17) >> 1)), ((M - 17) & 1)>(r);
        } else if constexpr (M == 31) {
            wait_rows();
            x_pair<2, 0>(); x_pair<2, 1>(); x_pair<3, 0>(); x_pair<3, 1>();
            y_col<0, 0>(); y_col<0, 1>(); y_col<1, 0>(); y_col<1, 1>();
        } else if constexpr (M == 39) {
            static_for<0, 8>([&](auto U) { z_row<(decltype(U)::value >> 2), (decltype(U)::value & 3)>(); });
        }
    }
};

// WC (r4): waves along the cout axis.  WC = 2 puts EIGHT waves in the workgroup -- two per SIMD -- that share one raw brick and one
// U ring: wave (wc, h, g) takes cout tile group wc (NC tiles), xi half h, tile group g.  Its 128 NC accumulator registers fit twice in
// a SIMD's file only with NC = 1; the input transform is then computed by both waves of a SIMD (each for its own cout tile); the idea
// was that while one of them issues VMEM / waits at the barrier, the other keeps the matrix pipe fed.  MEASURED (profiles/r04_wino_two_waves_per_simd.txt,
// -DWN_WC2_EXPERIMENT, rpn_net 128 -> 256): bit-identical output, 56.5 us against 52.3 -- the per-step costs beside the MFMAs (LDS-DMA
// and VALU issue) are costs of the SIMD, not of the wave, so a second wave hides none of them and the duplicated transform adds
// its 8 %.  Kept as a template parameter (the default WC = 1 is the shipped kernel); not instantiated in the library.
template <int H, int NC, int C3 = 0, int C2N = 0, bool MINI = false, int WC = 1>
__device__ __forceinline__ void wino_wave(const WinoArgs &a, float *lds, int prob, int brick, int grp, int gX, int gY, int gZ, int nby, int nbz,
                                          int64_t in_off, int64_t out_off, [[maybe_unused]] unsigned long long t_entry = 0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    [[maybe_unused]] unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tsp[4] = {0, 0, 0, 0}, tse[5] = {0, 0, 0, 0, 0};
    if constexpr (WN_EXP & 64) ts0 = wall_clock64();       // experiment: 100 MHz timestamps of the phases, written instead of the output
    const int g = wave & 1;
    const int wc = WC == 1 ? 0 : (wave >> 2);              // which NC cout tiles of the workgroup's NC WC
    const int li = lane & 15, kq = lane >> 4;
    using G_ = Geo<MINI>;
    constexpr int NTHR = 256 * WC;                          // shadows the file-level constant: threads of THIS instantiation
    constexpr int NVOX = G_::nvox, NIT = (NVOX + NTHR - 1) / NTHR, PS = G_::ps, HZS = G_::hzs, CHS = G_::chs, RAW_STAGE = raw_stage<MINI>();
    // block geometry: origin of the 8 x 4 x 8 block; MINI: origins of the workgroup's four 4 x 4 x 4 minis (minis 4 brick .. + 3 of the
    // problem's nmx x nmy x nmz grid, z fastest; a mini past the end sits far outside the grid: all zeros in, nothing stored)
    int ox0 = 0, oy0 = 0, oz0 = 0;
    [[maybe_unused]] int mox[4], moy[4], moz[4];
    if constexpr (MINI) {
        const int nmy = (gY + 3) >> 2, nmz = (gZ + 3) >> 2, nm = ((gX + 3) >> 2) * nmy * nmz;
        static_for<0, 4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int mi = 4 * brick + j;
            const bool dead = mi >= nm;
            mox[j] = dead ? (1 << 20) : 4 * (mi / (nmz * nmy));
            moy[j] = dead ? (1 << 20) : 4 * ((mi / nmz) % nmy);
            moz[j] = dead ? (1 << 20) : 4 * (mi % nmz);
        });
    } else {
        const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
        ox0 = bx * VX; oy0 = by * VY; oz0 = bz * VZ;
    }
    const float *__restrict__ p_in = a.in[prob] + in_off;
    const float *__restrict__ p_wp = a.wp[prob];
    // the launch's scalars: read from the kernel-argument segment ONCE, here.  Left alone, hipcc treats a kernel argument as free to
    // re-load and does so inside the staging plan's branches and in every iteration of the store tail -- each time a scalar-cache round
    // trip the wave sits out (r6, in-kernel timestamps: 2.35 us from entry to the first loads issued, 1.2 us for eight stores); the
    // empty asm makes the value opaque, so it stays in its SGPR
    int nk = a.nk, cin_stride = a.cin_stride;
    asm volatile("" : "+s"(nk), "+s"(cin_stride));

    float *raw = lds;                                   // [NRAW][4][CHS]
    constexpr int B_STAGE = NC * WC * B_TILE;           // floats of one K-step's U stage: NC WC cout tiles
    float *bst = lds + NRAW * RAW_STAGE;                // [NBST][NC][16][64][4]

    // U stage of K-step k: 16 NC WC wave-instructions of 1 KB, 4 NC per wave: block n = 4 NC wave + i = (cout tile n >> 4, xi quad n & 15)
    constexpr int NFILL = 4 * NC;
    const float *wbase = p_wp + (size_t)(NC * WC * grp) * nk * B_TILE;
    const int woff = lane * 4;
    // a wave's NFILL blocks are consecutive in the packed weights and in the stage (NFILL divides 16): one global base and one LDS
    // base (M0) per four instructions, the 1 KB steps in between as the instructions' immediate offsets
    const int n0 = wave * NFILL;
    const float *wwave = wbase + (size_t)(n0 >> 4) * nk * B_TILE + (n0 & 15) * 256;
    auto fill_b_item = [&](auto I, int k, int buf) {
        constexpr int i = decltype(I)::value;
        glds16<(i & 3) * 1024>(wwave + (size_t)k * B_TILE + (i >> 2) * 1024 + woff, bst + buf * B_STAGE + (n0 + (i >> 2) * 4) * 256);
    };
    // ---- prologue (r6 order): everything that crosses the memory system goes out FIRST -- U stages 0 and 1 need only the cout group,
    // the raw stages 0 and 1 of a staging item go out as soon as that item's offset is known -- and the work that needs no memory
    // (zero-filling the halo outside the grid, clearing 128 NC accumulator registers) runs under the loads' latency instead of in front
    // of them
    static_for<0, NFILL>([&](auto I) { fill_b_item(I, 0, 0); });
    [[maybe_unused]] unsigned long long tsq = 0;
    if constexpr (WN_EXP & 64) tsq = wall_clock64();       // U stages 0, 1 issued

    // ---- staging plan: item = halo voxel (4 channels = one float4 of its channels-last row).  Branch-free and VALU-free (an
    // fp32 VALU instruction costs the wave a 4-cycle issue slot that the matrix pipe cannot overlap): addresses are a uniform
    // base pointer that advances with k (scalar adds) plus a 32-bit per-lane offset fixed for the whole launch; a halo voxel
    // outside the grid (the same voxels in every K-step) is zeroed once in all three ring stages and its item loads from
    // offset 0 and stores to a dump slot in the pad behind the last x-plane of each channel, as do the items past the brick
    int goff[NIT], loff[NIT], zpos[NIT];
    // In the loop the loads are inline asm: hipcc waits vmcnt(0) -- i.e. also for every LDS-DMA in flight -- at the first use of an
    // ordinary load's result; here the only wait is the counted one in front of the LDS stores (step()).
    f32x4 sv[NIT], sw[NIT];
    const float *p_in1 = p_in + (nk > 1 ? 4 : 0);
    static_for<0, NIT>([&](auto I) {
        constexpr int it = decltype(I)::value;
        const int v = tid + it * NTHR;
        int gx, gy, gz, lpos;
        if constexpr (MINI) {
            const int mj = v / 216, r = v - mj * 216;               // mini, voxel of its 6 x 6 x 6 halo brick
            const int hz = r % 6, hy = (r / 6) % 6, hx = r / 36;
            const int sx = mj == 0 ? mox[0] : mj == 1 ? mox[1] : mj == 2 ? mox[2] : mox[3];
            const int sy = mj == 0 ? moy[0] : mj == 1 ? moy[1] : mj == 2 ? moy[2] : moy[3];
            const int sz = mj == 0 ? moz[0] : mj == 1 ? moz[1] : mj == 2 ? moz[2] : moz[3];
            gx = sx - 1 + hx; gy = sy - 1 + hy; gz = sz - 1 + hz;
            lpos = mj * G_::ms + hx * PS + hy * HZS + hz;
        } else {
            const int hz = v % HZ, hy = (v / HZ) % HY, hx = v / (HZ * HY);
            gx = ox0 - 1 + hx; gy = oy0 - 1 + hy; gz = oz0 - 1 + hz;
            lpos = hx * PS + hy * HZS + hz;
        }
        const bool inside = v < NVOX && (unsigned)gx < (unsigned)gX && (unsigned)gy < (unsigned)gY && (unsigned)gz < (unsigned)gZ;
        goff[it] = inside ? ((gx * gY + gy) * gZ + gz) * cin_stride : 0;
        loff[it] = 4 * (inside ? lpos : G_::dump + (lane & 31));    // BYTES; dump: 32 floats of the pad behind the last plane of a channel
        zpos[it] = (v < NVOX && !inside) ? lpos : -1;
        load16_asm(sv[it], goff[it] * 4, p_in);
        load16_asm(sw[it], goff[it] * 4, p_in1);
    });
    // U stage 1 LAST: step 0 needs it only at its sixth quad, and the step's own counted wait (slot 40) covers it -- the wait below
    // leaves these NFILL pieces in flight (a CU takes in ~24 B per clock: the 32 KB would hold the first MFMA up by ~0.5 us)
    static_for<0, NFILL>([&](auto I) { fill_b_item(I, nk > 1 ? 1 : 0, 1); });
    if constexpr (WN_EXP & 64) tsp[0] = wall_clock64();    // setup + loads issued
    static_for<0, NIT>([&](auto I) {
        constexpr int it = decltype(I)::value;
        if (zpos[it] >= 0) {
            static_for<0, NRAW * 4>([&](auto C) { raw[decltype(C)::value * CHS + zpos[it]] = 0.f; });
        }
    });
    auto stage_load_item = [&](auto I, int k) {
        constexpr int it = decltype(I)::value;
        const float *base = p_in + 4 * k;
        const int boff = goff[it] * 4;
        load16_asm(sv[it], boff, base);
    };
    auto stage_store_item = [&](auto I, int buf) {
        constexpr int it = decltype(I)::value;
        float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(raw) + (buf * (RAW_STAGE * 4) + loff[it]));   // one VALU add
        dst[0] = sv[it][0];
        dst[CHS] = sv[it][1];
        dst[2 * CHS] = sv[it][2];
        dst[3 * CHS] = sv[it][3];
    };

    f32x4 acc[NC][32];                                   // [cout tile c][xi of this half]
    static_for<0, 32 * NC>([&](auto I) { acc[decltype(I)::value >> 5][decltype(I)::value & 31] = (f32x4){0.f, 0.f, 0.f, 0.f}; });

    // lane (li, kq): tile li of group g = (txl, ty, tz), channel kq of the K-step
    const int txl = li >> 3, ty = (li >> 2) & 1, tz = li & 3;
    // MINI: tile li of group g = mini 2 g + (li >> 3), tile (tx, ty, tz) = bits 2, 1, 0 of li inside it
    const int rbase = MINI ? kq * CHS + (2 * g + (li >> 3)) * G_::ms + (2 * ((li >> 2) & 1)) * PS + (2 * ((li >> 1) & 1)) * HZS + 2 * (li & 1)
                           : kq * CHS + (2 * (2 * g + txl)) * PS + (2 * ty) * HZS + 2 * tz;
    const int bbase = wc * NC * B_TILE + H * 8 * 256 + lane * 4;

    if constexpr (WN_EXP & 64) tsp[3] = tsp[0];
    wait_vmcnt_all<NFILL>(sv);
    wait_vmcnt_all<NFILL>(sw);
    if constexpr (WN_EXP & 64) tsp[1] = wall_clock64();    // first loads landed
    static_for<0, NIT>([&](auto I) { stage_store_item(I, 0); });
    static_for<0, NIT>([&](auto I) { sv[decltype(I)::value] = sw[decltype(I)::value]; stage_store_item(I, 1); });
    // not __syncthreads(): its fence waits vmcnt(0), i.e. for U stage 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // raw stage 2 goes in flight now: in the loop, the loads of stage k + 3 are issued at the END of step k (behind the LDS stores of
    // stage k + 2) and waited for 54 MFMAs later, in front of the stores of step k + 1 -- an L2 round trip under load is longer than
    // the 35 MFMAs the loads had when they were issued at the start of the step that stores them
    if (!(WN_EXP & 2)) static_for<0, NIT>([&](auto I) { stage_load_item(I, nk > 2 ? 2 : nk - 1); });
    // V ping-pongs between two NextV objects (no register copies): step k multiplies with one while the other is being built
    NextV<H, MINI> va, vb;
    if constexpr (WN_EXP & 64) tsp[2] = wall_clock64();    // stores + barrier
    static_for<0, 64>([&](auto M) { va.template unit<decltype(M)::value>(LdsRow(raw + rbase)); });

    // ---- main loop.  Step k: the 64 MFMAs of step k (32 xi x 2 cout tiles); behind them, in the gaps between MFMAs: the LDS-DMA of
    // U stage k + 2 (slots 7..21, one per gap), the input transform of step k + 1 (NextV::unit: LDS reads + three bursts of packed
    // adds), the LDS stores of raw stage k + 2 (41..45) and the global loads of raw stage k + 3 (49..53).  A lone wave issues in
    // order: anything placed in front of or behind the MFMA block would hold the matrix pipe up by its whole issue time.
    // What the gaps cost (tools/issue_overlap.hip, one wave per SIMD): LDS reads, LDS writes and scalar instructions nothing; a VMEM
    // instruction ~11 ns wherever it stands; an fp32 VALU instruction ~11 cycles alone but 4 in a burst -- hence the bursts.
    // Waits: hipcc waits with lgkmcnt(0) in front of the first use of any freshly read register, so every LDS read is issued right
    // BEHIND an MFMA that carries such a wait, never in front of one (the wait would cover the read just issued: a full LDS round
    // trip per quad); the rows' own waits (NextV::wait_rows) sit at quad ends, seven MFMAs after the quad's U reads.
    // LDS-DMA lands ~1 us after issue under load (longer than a step), so U runs TWO stages ahead through a ring of three, and
    // no wait in the loop is a vmcnt(0): the one in front of the raw-stage stores leaves this step's eight DMA instructions
    // outstanding (vmcnt(8): the three loads issued at the end of the previous step and everything older have completed).
    // One barrier per step, and it sits INSIDE the MFMA block, after quad 5: by then every LDS read of U stage k is issued and
    // stage k + 1 has landed, so the barrier is followed by the 16 MFMAs of quads 6, 7 of step k, whose operands are already in
    // registers (and, behind the first of them, by the first U reads of step k + 1) -- the barrier skew and the LDS latency of
    // the next step's first operands hide behind 512 cycles of matrix work instead of idling the pipe.
    // Loads / DMA of the steps past the end are clamped to the last step (harmless duplicates, landed before the final barrier):
    // no branches in the block.
    static_assert(NIT == 2 || NIT == 3 || (NIT == 4 && NC == 2), "slots 41.. / 49.. hold the stores / loads of at most four staging items");
    int cur = 0;                                         // k % 3: raw stage of step k, U stage of step k
    f32x4 bq[4][NC];                                     // ring over xi quads (slot q & 3), every cout tile of the group
    auto read_b = [&](auto Q, const float *bs) {
        constexpr int q = decltype(Q)::value;
        static_for<0, NC>([&](auto C) { bq[q & 3][decltype(C)::value] = *reinterpret_cast<const f32x4 *>(bs + decltype(C)::value * B_TILE + q * 256); });
    };
    read_b(std::integral_constant<int, 0>{}, bst + bbase);
    read_b(std::integral_constant<int, 1>{}, bst + bbase);
    auto step = [&](int k, NextV<H, MINI> &vu, NextV<H, MINI> &vn) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        const int ks = k + 2 < nk ? k + 2 : nk - 1, ks3 = k + 3 < nk ? k + 3 : nk - 1;
        const float *bs = bst + cur * B_STAGE + bbase;
        const float *bs_next = bst + nxt * B_STAGE + bbase;
        const LdsRow rn(raw + nxt * RAW_STAGE + rbase);
        static_for<0, 8>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if constexpr (q == 6) {
                // everything but this step's DMA has completed (the counted wait at slot 40 stands); LDS traffic of this wave done
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(WN_EXP & 4)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            static_for<0, 4 * NC>([&](auto E) {
                constexpr int en = decltype(E)::value, e = en / NC, cc = en % NC, m = 4 * NC * q + en;
                mfma_agpr(acc[cc][4 * q + e], vu.a_operand(q >> 2, q & 3, e), bq[q & 3][cc][e]);
                // U reads of quad q + 2 go out BEHIND the first MFMA of quad q: hipcc waits with lgkmcnt(0) in front of the first use of
                // a freshly read register, i.e. in front of that MFMA -- anything issued just before it would be waited for in full
                if constexpr (en == 0 && q + 2 < 8) read_b(std::integral_constant<int, q + 2>{}, bs);
                if constexpr (en == 0 && q == 6) {              // first U operands of the next step (its stage landed before the barrier)
                    read_b(std::integral_constant<int, 0>{}, bs_next);
                    read_b(std::integral_constant<int, 1>{}, bs_next);
                }
                // the units are laid out on 64 virtual slots: one per MFMA with two cout tiles, two per MFMA with one
                static_for<0, 2 / NC>([&](auto V) {
                    constexpr int sl = m * (2 / NC) + decltype(V)::value;
                    if constexpr (!(WN_EXP & 16)) vn.template unit<sl>(rn);
                    if constexpr (sl >= 49 && sl < 49 + 2 * NIT && (sl & 1) == 1 && !(WN_EXP & 2)) stage_load_item(std::integral_constant<int, (sl - 49) / 2>{}, ks3);
                    if constexpr (sl >= 7 && sl < 7 + 2 * NFILL && (sl & 1) == 1 && !(WN_EXP & 1)) fill_b_item(std::integral_constant<int, (sl - 7) / 2>{}, ks, nn);
                    if constexpr (sl == 40 && !(WN_EXP & 2)) wait_vmcnt_all<(WN_EXP & 512) ? 0 : NFILL>(sv);
                    if constexpr (sl >= 41 && sl < 41 + 2 * NIT && (sl & 1) == 1 && !(WN_EXP & 2)) stage_store_item(std::integral_constant<int, (sl - 41) / 2>{}, nn);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        cur = nxt;
    };
    // nk is even (cin % 8 == 0) and BOTH steps of the loop body always transform: the last step's transform re-reads a stale
    // (valid) raw stage and its result is never used -- 96 wasted adds once per launch buy a loop with exactly two step shapes
    // and one register assignment for the 256 accumulator registers (hipcc otherwise moves them between the AGPR and VGPR halves
    // at every change of shape: 512 v_accvgpr moves per transition)
    if constexpr (WN_EXP & 64) ts1 = wall_clock64();
    for (int k = 0; k < nk; k += 2) {
        step(k, va, vb);
        step(k + 1, vb, va);
    }
    if constexpr (WN_EXP & 64) ts2 = wall_clock64();
    // the clamped duplicate DMA and raw-stage loads of the last steps.  The wait names sv: the loads are asm, so hipcc does not know
    // they are still writing those registers -- without the operands it hands them to the epilogue above this line
    wait_vmcnt_all<0>(sv);
    // the MFMAs are asm: hipcc's hazard recogniser does not know that the accumulators the epilogue is about to read were written by
    // the matrix pipe a few cycles ago (an 8-pass MFMA needs up to 11 wait states before a VALU read of its result)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();                                     // quads 6, 7 of the last step ran behind the last in-loop barrier
    if constexpr (WN_EXP & 64) tse[0] = wall_clock64();

    // ---- output transform A^T M A: per lane, cout tile c and row r (tile 4 (lane >> 4) + r, cout lane & 15): 32 xi -> 8 partial
    // outputs (this wave's xi_x half).  The halves meet through LDS (the U stages are free: the loop ended on a barrier, no DMA
    // is in flight): each wave FINISHES two of the four rows (H = 0: r = 0, 1; H = 1: r = 2, 3) and ships the partials of the
    // other two to its partner.
    // r6: the rows are handled as PAIRS (r, r + 1) -- the two halves of an accumulator quad -- with packed adds: half the VALU
    // instructions for bit-identical sums (same association, v_pk_add_f32 with neg modifiers = the scalar add / sub), 8-byte LDS
    // traffic for the partner exchange; the bias is requested before the transform so that its L2 round trip is not waited for in
    // front of the first finished row, and the launch's scalars come out of SGPRs (see the prologue)
    const int j = lane & 15, q4 = lane >> 4;
    float *__restrict__ p_out = a.out[prob] + out_off;
    int cout = a.cout, out_stride = a.out_stride, out_coff = a.out_coff, eflags = a.flags;
    asm volatile("" : "+s"(cout), "+s"(out_stride), "+s"(out_coff), "+s"(eflags));
    const int cobase = 16 * NC * (WC * grp + wc);
    float bv[NC];
    {
        const float *bp = a.bias[prob];
        static_for<0, NC>([&](auto C) {
            const int co = cobase + 16 * decltype(C)::value + j;
            bv[decltype(C)::value] = (bp && co < cout) ? bp[co] : 0.f;
        });
    }
    auto rows2 = [&](auto C, auto PP, f32x2 (&P)[8]) {      // the 8 partial outputs (ox oy oz) of row pair PP of cout tile C
        constexpr int cc = decltype(C)::value, pp = decltype(PP)::value;
        f32x2 t[2][4][2];
        static_for<0, 2>([&](auto I) {
            constexpr int i = decltype(I)::value;
            static_for<0, 4>([&](auto Y) {
                constexpr int y = decltype(Y)::value;
                const f32x4 q0 = acc[cc][i * 16 + y * 4 + 0], q1 = acc[cc][i * 16 + y * 4 + 1], q2 = acc[cc][i * 16 + y * 4 + 2], q3 = acc[cc][i * 16 + y * 4 + 3];
                const f32x2 m0 = {q0[2 * pp], q0[2 * pp + 1]}, m1 = {q1[2 * pp], q1[2 * pp + 1]}, m2 = {q2[2 * pp], q2[2 * pp + 1]},
                            m3 = {q3[2 * pp], q3[2 * pp + 1]};
                t[i][y][0] = epk_add(epk_add(m0, m1), m2);                 // (m0 + m1) + m2
                t[i][y][1] = epk_sub(epk_sub(m1, m2), m3);                 // (m1 - m2) - m3
            });
        });
        f32x2 s2[2][2][2];
        static_for<0, 2>([&](auto I) {
            constexpr int i = decltype(I)::value;
            static_for<0, 2>([&](auto Z) {
                constexpr int z = decltype(Z)::value;
                s2[i][0][z] = epk_add(epk_add(t[i][0][z], t[i][1][z]), t[i][2][z]);
                s2[i][1][z] = epk_sub(epk_sub(t[i][1][z], t[i][2][z]), t[i][3][z]);
            });
        });
        static_for<0, 4>([&](auto O) {
            constexpr int o = decltype(O)::value, oy = o >> 1, oz = o & 1;
            if constexpr (H == 0) {
                P[0 * 4 + o] = epk_add(s2[0][oy][oz], s2[1][oy][oz]);     // ox = 0: m0 + m1 (+ m2 from the other half)
                P[1 * 4 + o] = s2[1][oy][oz];                             // ox = 1: m1 (- m2 - m3 from the other half)
            } else {
                P[0 * 4 + o] = s2[0][oy][oz];                             // m2
                P[1 * 4 + o] = epk_nsub(s2[0][oy][oz], s2[1][oy][oz]);     // - m2 - m3
            }
        });
    };
    // partner scratch: [wave pair (wc, g)][cout tile][row pair][o][lane] x 8 B
    f32x2 *scr2 = reinterpret_cast<f32x2 *>(bst) + (size_t)((2 * wc + g) * NC) * 2 * (8 * 64) + lane;
    f32x2 P[NC][8];                                      // the row pair this wave finishes (pair H)
    static_for<0, NC>([&](auto C) {                      // the partner's pair first: its LDS stores drain under the own pair's adds
        constexpr int cc = decltype(C)::value;
        f32x2 S[8];
        rows2(C, std::integral_constant<int, 1 - H>{}, S);
        static_for<0, 8>([&](auto O) { scr2[((cc * 2 + (1 - H)) * 8 + decltype(O)::value) * 64] = S[decltype(O)::value]; });
    });
    static_for<0, NC>([&](auto C) { rows2(C, std::integral_constant<int, H>{}, P[decltype(C)::value]); });
    if constexpr (WN_EXP & 64) tse[1] = wall_clock64();    // output transform + partner stores
    __syncthreads();
    if constexpr (WN_EXP & 64) tse[2] = wall_clock64();
    // finish rows r = 2H, 2H + 1: + partner's partial, + bias, ReLU; transpose through LDS (second U stage) so that a lane
    // stores 16 B = four consecutive couts of one voxel: [cc][voxel = (r', tile quad q4, o)][cout 16] per wave
    float *tr = bst + B_STAGE + wave * (NC * 2 * 4 * 8 * 16);      // B_STAGE floats = the partner scratch above (2 WC NC x 4 rows x 512)
    static_for<0, NC>([&](auto C) {
        constexpr int cc = decltype(C)::value;
        const f32x2 bv2 = {bv[cc], bv[cc]};
        static_for<0, 8>([&](auto O) {
            constexpr int o = decltype(O)::value;
            P[cc][o] = epk_add(epk_add(P[cc][o], scr2[((cc * 2 + H) * 8 + o) * 64]), bv2);
        });
    });
    if (eflags & SIS3D_EPI_RELU) {                          // fmaxf(v, 0) as ONE instruction per value (hipcc's fmaxf: canonicalise + max + select)
        static_for<0, NC * 8>([&](auto I) {
            f32x2 &v = P[decltype(I)::value >> 3][decltype(I)::value & 7];
            asm("v_max_f32 %0, 0, %0" : "+v"(v.x));
            asm("v_max_f32 %0, 0, %0" : "+v"(v.y));
        });
    }
    static_for<0, NC>([&](auto C) {
        constexpr int cc = decltype(C)::value;
        static_for<0, 8>([&](auto O) {
            constexpr int o = decltype(O)::value;
            tr[(((cc * 2 + 0) * 4 + q4) * 8 + o) * 16 + j] = P[cc][o].x;
            tr[(((cc * 2 + 1) * 4 + q4) * 8 + o) * 16 + j] = P[cc][o].y;
        });
    });
    // wave-local exchange: every lane reads what other lanes of its own wave wrote
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (WN_EXP & 64) tse[3] = wall_clock64();    // rows finished, transposed in LDS
    if constexpr (C3 > 0) {
        // ---- fused Bottleneck tail, wave-local: this wave's 64 finished voxels x (16 NC = all) conv2 channels sit in `tr` as rows
        // [cout tile][voxel row][16]; a row IS the B operand of the transposed tile GEMM of mfma16.h (lane (voxel li, kq) reads the
        // 16 B = channels 16 g + 4 kq .. + 3), so conv3 -> (+ bias, + residual, ReLU) -> next conv1 chains through registers exactly
        // as in pointwise.hip's pw16_kernel, on four voxel tiles of 16.  Weights: pw16 fragment order, straight from L2 (2-8 KB).
        static_assert(NC == 2 && WC == 1 && C3 % 16 == 0 && C2N % 16 == 0, "the tail needs every conv2 channel in the workgroup (cout = 32 = 16 NC)");
        constexpr int NT3 = C3 / 16, NT2 = C2N / 16, NT2A = NT2 > 0 ? NT2 : 1;
        float4 w3[NT3][2], bb3[NT3];
        static_for<0, NT3>([&](auto N) {
            constexpr int n = decltype(N)::value;
            static_for<0, 2>([&](auto G) { w3[n][decltype(G)::value] = reinterpret_cast<const float4 *>(a.w3p)[(n * 2 + decltype(G)::value) * 64 + lane]; });
            bb3[n] = a.b3 ? *reinterpret_cast<const float4 *>(a.b3 + 16 * n + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
        });
        float4 w1n[NT2A][NT3], bb1n[NT2A];
        if constexpr (NT2 > 0) {
            static_for<0, NT2>([&](auto N) {
                constexpr int n = decltype(N)::value;
                static_for<0, NT3>([&](auto G) { w1n[n][decltype(G)::value] = reinterpret_cast<const float4 *>(a.w1n)[(n * NT3 + decltype(G)::value) * 64 + lane]; });
                bb1n[n] = a.b1n ? *reinterpret_cast<const float4 *>(a.b1n + 16 * n + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
            });
        }
        // residual rows: all four tiles' requested before the first GEMM (one L2 round trip for the wave) while they fit the register
        // file beside the weights (C3 <= 64: 4 x NT3 float4); the C3 = 128 blocks (NT3 = 8) keep two tiles' rows and request tile t + 1's
        // in front of tile t's GEMMs
        constexpr int RDEPTH = NT3 <= 4 ? 4 : 2;
        float4 rres[RDEPTH][NT3];
        bool okv[4];
        size_t vox[4];
        static_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            const int row = 16 * t + j;                              // voxel row (rr, tq, o) of this lane's B-operand column
            const int o = row & 7, tq = (row >> 3) & 3, rr = row >> 5;
            const int tl = 4 * tq + 2 * H + rr;
            const int x = ox0 + 2 * (2 * g + (tl >> 3)) + (o >> 2), y = oy0 + 2 * ((tl >> 2) & 1) + ((o >> 1) & 1), z = oz0 + 2 * (tl & 3) + (o & 1);
            okv[t] = x < gX && y < gY && z < gZ;
            vox[t] = okv[t] ? (size_t)(x * gY + y) * gZ + z : 0;
        });
        auto load_res = [&](auto T) {
            constexpr int t = decltype(T)::value;
            const float *rp = a.res + vox[t] * a.res_stride + 4 * q4;
            static_for<0, NT3>([&](auto N) { rres[t % RDEPTH][decltype(N)::value] = *reinterpret_cast<const float4 *>(rp + 16 * decltype(N)::value); });
        };
        static_for<0, (RDEPTH == 4 ? 4 : 1)>([&](auto T) { load_res(T); });
        static_for<0, 4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            if constexpr (RDEPTH == 2 && t + 1 < 4) load_res(std::integral_constant<int, t + 1>{});
            float4 yv[2];
            static_for<0, 2>([&](auto G) { yv[decltype(G)::value] = *reinterpret_cast<const float4 *>(tr + ((decltype(G)::value * 64 + 16 * t + j) * 16 + 4 * q4)); });
            f32x4 acc3[NT3];
            gemm_t<NT3, 2>(w3, yv, acc3);
            float4 zv[NT3];
            static_for<0, NT3>([&](auto N) {
                constexpr int n = decltype(N)::value;
                float4 v;
                constexpr int ts = t % RDEPTH;
                v.x = acc3[n][0] + bb3[n].x + rres[ts][n].x; v.y = acc3[n][1] + bb3[n].y + rres[ts][n].y;
                v.z = acc3[n][2] + bb3[n].z + rres[ts][n].z; v.w = acc3[n][3] + bb3[n].w + rres[ts][n].w;
                zv[n] = relu4(v, true);
                if (okv[t]) *reinterpret_cast<float4 *>(a.tout + vox[t] * a.tout_stride + a.tout_coff + 16 * n + 4 * q4) = zv[n];
            });
            if constexpr (NT2 > 0) {
                f32x4 acc1[NT2];
                gemm_t<NT2, NT3>(w1n, zv, acc1);
                static_for<0, NT2>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    float4 v;
                    v.x = acc1[n][0] + bb1n[n].x; v.y = acc1[n][1] + bb1n[n].y; v.z = acc1[n][2] + bb1n[n].z; v.w = acc1[n][3] + bb1n[n].w;
                    v = relu4(v, true);
                    if (okv[t]) *reinterpret_cast<float4 *>(a.out2 + vox[t] * a.out2_stride + 16 * n + 4 * q4) = v;
                });
            }
        });
    }
    if (C3 == 0 || a.out[prob] != nullptr) {
        // r6 store tail.  One instruction = 16 / NC voxel rows x (16 NC couts = 64 NC B contiguous: with two cout tiles a full 128-B line
        // per voxel); a lane's address = the output base (SGPRs) + a 32-bit offset = [lane part: its voxel's place inside a tile, its
        // cout piece] + [tile part: which tile of the block this instruction serves, uniform when NC = 2] -- both linear in (x, y, z), so
        // they are built once / on the scalar unit instead of 64-bit multiply-adds per store; the partial-tile / unaligned form is ONE
        // uniform branch, not a test per store
        constexpr int LPR = 4 * NC, RPI = 64 / LPR;                 // lanes per voxel row, voxel rows per instruction
        const int c4 = lane & 3, ccl = (lane >> 2) & (NC - 1), vr = lane / LPR;
        auto lin = [&](int x, int y, int z) { return ((x * gY + y) * gZ + z) * out_stride; };
        const bool fast = (((out_stride | out_coff) & 3) == 0) && cobase + 16 * NC <= cout;
        // every row of the transposed tile first (one LDS round trip for the lot), then the stores
        f32x4 rowv[4 * NC];
        static_for<0, 4 * NC>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            rowv[sidx] = *reinterpret_cast<const f32x4 *>(tr + (ccl * 64 + RPI * sidx + vr) * 16 + c4 * 4);
        });
        const int co = cobase + 16 * ccl + 4 * c4;
        static_for<0, 4 * NC>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            const int vrow = RPI * sidx + vr;                        // (rr, tq, o) of the wave's 64 finished voxels
            const int o = vrow & 7, tq = (vrow >> 3) & 3, rr = vrow >> 5;
            const int tl = 4 * tq + 2 * H + rr;
            int x, y, z;
            if constexpr (MINI) {
                const bool mb = (tl >> 3) & 1;                              // which of this wave's two minis (2 g, 2 g + 1)
                const int sx = g ? (mb ? mox[3] : mox[2]) : (mb ? mox[1] : mox[0]);
                const int sy = g ? (mb ? moy[3] : moy[2]) : (mb ? moy[1] : moy[0]);
                const int sz = g ? (mb ? moz[3] : moz[2]) : (mb ? moz[1] : moz[0]);
                x = sx + 2 * ((tl >> 2) & 1); y = sy + 2 * ((tl >> 1) & 1); z = sz + 2 * (tl & 1);
            } else {
                x = ox0 + 2 * (2 * g + (tl >> 3)); y = oy0 + 2 * ((tl >> 2) & 1); z = oz0 + 2 * (tl & 3);
            }
            const int dx = o >> 2, dy = (o >> 1) & 1, dz = o & 1;
            const bool ok = (int)(x + dx < gX) & (int)(y + dy < gY) & (int)(z + dz < gZ) & (int)(!(WN_EXP & 32) || rowv[sidx][0] == 123.456f);
            const unsigned boff = 4u * (unsigned)(lin(x, y, z) + lin(dx, dy, dz) + out_coff + co);
            if (fast) {
                if (ok) store16_asm(rowv[sidx], boff, p_out);
            } else if (ok) {
                float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(p_out) + boff);
                if (co < cout) dst[0] = rowv[sidx][0];
                if (co + 1 < cout) dst[1] = rowv[sidx][1];
                if (co + 2 < cout) dst[2] = rowv[sidx][2];
                if (co + 3 < cout) dst[3] = rowv[sidx][3];
            }
        });
    }
    if constexpr (WN_EXP & 64) {
        const unsigned long long ts3 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts4 = wall_clock64();       // stores drained
        if (tid == 0) {
            // 20 floats per workgroup (ns) BEHIND the real output: phases, sub-phases of prologue / epilogue, start and end on the 100 MHz
            // clock, which CU
            float *d = p_out + (size_t)gX * gY * gZ * out_stride + (size_t)(brick * a.ngroups + grp) * 20;
            d[0] = (float)(ts1 - ts0) * 10.f; d[1] = (float)(ts2 - ts1) * 10.f; d[2] = (float)(ts3 - ts2) * 10.f; d[3] = (float)(ts0 % 10000000ull) * 10.f;
            d[4] = (float)(tsp[0] - ts0) * 10.f; d[5] = (float)(tsp[3] - tsp[0]) * 10.f; d[6] = (float)(tsp[1] - tsp[3]) * 10.f;
            d[7] = (float)(tsp[2] - tsp[1]) * 10.f; d[8] = (float)(ts1 - tsp[2]) * 10.f;
            d[9] = (float)(tse[0] - ts2) * 10.f; d[10] = (float)(tse[1] - tse[0]) * 10.f; d[11] = (float)(tse[2] - tse[1]) * 10.f;
            d[12] = (float)(tse[3] - tse[2]) * 10.f; d[13] = (float)(ts3 - tse[3]) * 10.f; d[14] = (float)(ts4 - ts3) * 10.f;
            d[15] = (float)(ts4 % 10000000ull) * 10.f; d[17] = (float)(tsq - ts0) * 10.f;
            d[16] = (float)((__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xff00) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16));
        }
    }
}

typedef float pig4 __attribute__((ext_vector_type(4)));

template <int NC, int C3 = 0, int C2N = 0, bool MINI = false, int WC = 1, bool PIGGY = false>
__global__ __launch_bounds__(NTHR * WC, 1) void conv3d_k3wino_kernel(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    [[maybe_unused]] unsigned long long t_entry = 0;
    if constexpr (PIGGY) {
        if ((int)blockIdx.y == a.pig_row) {
            if (blockIdx.x >= 8) return;
            const unsigned long long sp = *reinterpret_cast<const unsigned long long *>(a.pig_state + 16);     // MailSlot.next_src
            // state[24..25]: whose chunk pig_dst holds when this launch has ended (0: nobody's)
            if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned long long *>(a.pig_state + 24) = sp;
            const pig4 *src = reinterpret_cast<const pig4 *>(sp);
            if (!src) return;
            pig4 *dst = reinterpret_cast<pig4 *>(a.pig_dst);
            constexpr int U = 8;
            const int64_t stride = (int64_t)8 * blockDim.x;
            int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
            for (; i + (U - 1) * stride < a.pig_n4; i += U * stride) {
                pig4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
                for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
            }
            for (; i < a.pig_n4; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
            return;
        }
    }
    // work list: cout group major, block minor; every XCD (block b runs on XCD b % 8, private L2) takes one contiguous range
    // of it, i.e. few cout groups x all blocks: its L2 holds 1/8 of U (8.4 MB for rpn_net) and the whole activation map
    int wid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    int gX = a.X, gY = a.Y, gZ = a.Z, nby = a.nby, nbz = a.nbz, grp, brick;
    int64_t in_off = 0, out_off = 0;
    if (a.nrag > 0) {
        // this workgroup's problem (block0 ascending): uniform -> scalar loads; work items of a problem are cout-group fastest
        int lo = 0, hi = a.nrag - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.rag[mid].block0 <= wid) lo = mid; else hi = mid - 1;
        }
        const WinoRagged d = a.rag[lo];
        wid -= d.block0;
        gX = d.X; gY = d.Y; gZ = d.Z; nby = d.nby; nbz = d.nbz;
        in_off = d.in_off; out_off = d.out_off;
        grp = wid % a.ngroups;
        brick = wid / a.ngroups;
    } else {
        const int nbr = a.nbx * a.nby * a.nbz;
        grp = wid / nbr;
        brick = wid - grp * nbr;
        // r4: with exactly eight cout groups (rpn_net: one per XCD under the rule above) an XCD takes TWO groups x HALF of the blocks
        // instead: its L2 then holds 2/8 of U (2.1 MB) and about half of the activation map (1.9 MB) instead of 1/8 and the whole map
        // (1.05 + 3.5 MB) -- less fetched per launch, same work per XCD.  Pair p = XCDs 2p, 2p + 1; the pair's 2 nbr items in the order
        // (group 2p, first half), (2p + 1, first half), (2p, second half), (2p + 1, second half), cut in two equal runs.
        if (XCD_PAIRS && a.ngroups == 8 && (int)gridDim.x == 8 * nbr) {
            const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;                 // idx in [0, nbr)
            const int p = xcd >> 1, j = (xcd & 1) * nbr + idx, hb = (nbr + 1) >> 1;
            if (j < hb) { grp = 2 * p; brick = j; }
            else if (j < 2 * hb) { grp = 2 * p + 1; brick = j - hb; }
            else if (j < hb + nbr) { grp = 2 * p; brick = j - hb; }
            else { grp = 2 * p + 1; brick = j - nbr; }
        }
    }
    // waves (wc, h, g): h = xi_x half, g = tile group, wc = cout tile group (WC = 2: waves w and w + 4 share a SIMD); each wave serves
    // NC cout tiles
    const int h = __builtin_amdgcn_readfirstlane((threadIdx.x >> 7) & 1);
    if (h == 0) wino_wave<0, NC, C3, C2N, MINI, WC>(a, lds, blockIdx.y, brick, grp, gX, gY, gZ, nby, nbz, in_off, out_off, t_entry);
    else wino_wave<1, NC, C3, C2N, MINI, WC>(a, lds, blockIdx.y, brick, grp, gX, gY, gZ, nby, nbz, in_off, out_off, t_entry);
}

// (Cout, Cin, 3, 3, 3) -> U = G g G^T per axis, packed [cout tile (even count)][K-step cin / 4][xi / 4][lane 64][4]:
// lane (j = lane & 15, kq = lane >> 4) holds U[xi = 4 xq + e][ci = 4 k + kq][co = 16 tile + j]
__global__ __launch_bounds__(256) void pack_weight_wino_kernel(const float *__restrict__ w, int cout, int cin, int ntp, int nk,
                                                               float *__restrict__ packed)
{
    const int64_t total = (int64_t)ntp * 16 * nk * 4;                      // (co padded, ci) pairs
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % (nk * 4)), co = (int)(idx / (nk * 4));
        float g[3][3][3];
        for (int t = 0; t < 27; ++t) (&g[0][0][0])[t] = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * 27 + t] : 0.f;
        // z axis
        float gz[3][3][4];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float w0 = g[i][j][0], w1 = g[i][j][1], w2 = g[i][j][2];
                gz[i][j][0] = w0; gz[i][j][1] = 0.5f * ((w0 + w2) + w1); gz[i][j][2] = 0.5f * ((w0 + w2) - w1); gz[i][j][3] = w2;
            }
        float gy[3][4][4];
        for (int i = 0; i < 3; ++i)
            for (int z = 0; z < 4; ++z) {
                const float w0 = gz[i][0][z], w1 = gz[i][1][z], w2 = gz[i][2][z];
                gy[i][0][z] = w0; gy[i][1][z] = 0.5f * ((w0 + w2) + w1); gy[i][2][z] = 0.5f * ((w0 + w2) - w1); gy[i][3][z] = w2;
            }
        const int tile = co >> 4, j = co & 15, k = ci >> 2, kq = ci & 3;
        float *dst = packed + ((int64_t)tile * nk + k) * B_TILE + (kq * 16 + j) * 4;
        for (int y = 0; y < 4; ++y)
            for (int z = 0; z < 4; ++z) {
                const float w0 = gy[0][y][z], w1 = gy[1][y][z], w2 = gy[2][y][z];
                const float u[4] = {w0, 0.5f * ((w0 + w2) + w1), 0.5f * ((w0 + w2) - w1), w2};
                for (int x = 0; x < 4; ++x) {
                    const int xi = x * 16 + y * 4 + z;
                    dst[(xi >> 2) * 256 + (xi & 3)] = u[x];
                }
            }
    }
}

} // namespace

extern "C" size_t sis3d_conv_k3wino_packed_floats(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || cin % 8) return 0;
    const int nt = (cout + 15) / 16, ntp = (nt + 1) & ~1;
    return (size_t)ntp * (cin / 4) * B_TILE;
}

extern "C" int sis3d_conv_k3wino_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream)
{
    if (!w || !packed || cout <= 0 || cin <= 0 || cin % 8) return SIS3D_EINVAL;
    const int nt = (cout + 15) / 16, ntp = (nt + 1) & ~1, nk = cin / 4;
    const int64_t total = (int64_t)ntp * 16 * nk * 4;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), w, cout, cin,
                       ntp, nk, packed);
    return sis3d_check_launch();
}

// Cout tiles per workgroup this kernel would use on the layer (2 or 1), or 0 when the direct kernel (sis3d_conv3d_k3t16) is
// expected to win.  A workgroup is a block of 8 x 4 x 8 voxels x 16 NC couts, one per CU (148 / 98 KB of LDS), so the layer needs
// >= ~200 (block, cout group) items to fill the chip -- with two cout tiles per wave if that still gives 200 (the input
// transform is shared by both), else with one -- and a channel loop long enough to amortise ~10 us of prologue + output
// transform.  Measured (tools/wino_bench.cpp, us, direct -> Winograd): rpn_net 128->256 @24x12x24: 102 -> 58 (pair 194 -> 112).
// shared_chip: the caller's dispatch regime (an argument since r5 -- no library state: launches / captures of the two regimes may run
// concurrently from different threads)
static int wino_nc(int X, int Y, int Z, int cin, int cout, bool shared_chip)
{
    if (X <= 0 || Y <= 0 || Z <= 0 || cin < 64 || cout < 64 || (cin % 8)) return 0;
    const int64_t blocks = (int64_t)cdiv(X, VX) * cdiv(Y, VY) * cdiv(Z, VZ);
    const int nt = (cout + 15) / 16;
    if (blocks * ((nt + 1) / 2) >= 200) return 2;
    // shared chip (several chunks in flight): what counts is CU-time, not the launch's own duration.  A Winograd workgroup of two
    // cout tiles does the work of ~6.75 direct-kernel workgroups' MFMAs, so even a layer with only ~50 work items (the 64 -> 64
    // convs of geometry2's Bottlenecks: 27 blocks x 2 cout pairs) costs the chip less on this kernel -- 54 CUs x ~27 us against
    // 256 x ~15 -- although alone it takes longer than the direct kernel's 17 us (env SIS3D_WINO_SHARED_MIN: work items needed)
    static const int shared_min = [] { const char *e = getenv("SIS3D_WINO_SHARED_MIN"); return e ? atoi(e) : 48; }();
    if (shared_chip && blocks * ((nt + 1) / 2) >= shared_min) return 2;
    if (blocks * nt >= 200 && cin >= 128) {
        // one cout tile per workgroup fills the chip when the launch has it alone (geometry2[0]: 216 work items, 34 us; on a shared
        // chip the rule above has already given it two: 108 work items x ~50 us, profiles/r04_shared_chip_rule.txt: 1.958 -> 1.992 G voxels/s with
        // three chunks in flight, one chunk alone 0.295 -> 0.308 ms)
        return 1;
    }
    return 0;
}

extern "C" int sis3d_conv3d_k3wino_prefer(int X, int Y, int Z, int cin, int cout, int nprob, int shared_chip)
{
    return nprob >= 1 && wino_nc(X, Y, Z, cin, cout, shared_chip != 0) > 0 ? 1 : 0;
}

template <int NC, int C3, int C2N, bool MINI = false, int WC = 1, bool PIGGY = false>
static int launch_wino_inst(const WinoArgs &a, int64_t nwg, int nprob, hipStream_t st)
{
    constexpr size_t lds = (size_t)lds_floats_g<MINI>(NC * WC) * sizeof(float);
    static Sis3dLdsOnce once;                                       // once per instantiation AND device
    auto kern = conv3d_k3wino_kernel<NC, C3, C2N, MINI, WC, PIGGY>;
    if (sis3d_grant_lds(once, (const void *)kern, (int)lds) != SIS3D_OK) return SIS3D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)nprob), dim3(NTHR * WC), lds, st, a);
    return sis3d_check_launch();
}

// nc = cout tiles per WAVE, wc = waves along cout (cout tiles per workgroup = nc wc)
static int launch_wino(WinoArgs &a, int nc, int wc, int64_t nwg, int nprob, hipStream_t st, unsigned *pig_state = nullptr,
                       float *pig_dst = nullptr, long long pig_n4 = 0)
{
    if (nwg <= 0 || nwg > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    a.w3p = a.b3 = a.res = a.w1n = a.b1n = nullptr; a.tout = a.out2 = nullptr;
    a.res_stride = a.tout_stride = a.tout_coff = a.out2_stride = 0;
    a.pig_state = pig_state; a.pig_dst = pig_dst; a.pig_n4 = pig_n4; a.pig_row = -1;
    if (pig_state) {
        // one more row of workgroups carries the upload; only the two-cout-tile form of the plain conv has the branch
        if (nc != 2 || wc != 1 || nwg < 8) return SIS3D_EUNSUPPORTED;
        a.pig_row = nprob;
        return launch_wino_inst<2, 0, 0, false, 1, true>(a, nwg, nprob + 1, st);
    }
#ifdef WN_WC2_EXPERIMENT    // measured and left out of the library (r4): 56.5 us against 52.3 on rpn_net, bit-identical output -- see wino_wave
    if (wc == 2) return nc == 1 ? launch_wino_inst<1, 0, 0, false, 2>(a, nwg, nprob, st) : SIS3D_EUNSUPPORTED;
#else
    if (wc != 1) return SIS3D_EUNSUPPORTED;
#endif
    return nc == 2 ? launch_wino_inst<2, 0, 0>(a, nwg, nprob, st) : launch_wino_inst<1, 0, 0>(a, nwg, nprob, st);
}

static int conv3d_k3wino_impl(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                              const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                              int out_stride, int out_coff, unsigned *pig_state, float *pig_dst, long long pig_n4, sis3d_stream_t stream)
{
    if (nprob < 1 || nprob > WN_MAXP || !ins || !packed_ws || !outs) return SIS3D_EINVAL;
    if (X <= 0 || Y <= 0 || Z <= 0 || cin <= 0 || cout <= 0 || cin_stride < cin || (cin_stride % 4)) return SIS3D_EINVAL;
    if ((cin % 8) || out_stride < out_coff + cout) return SIS3D_EUNSUPPORTED;        // K-steps of 4 channels, taken two at a time
    if (flags & ~(SIS3D_EPI_RELU | SIS3D_DISPATCH_SHARED_CHIP)) return SIS3D_EUNSUPPORTED;
    if ((int64_t)X * Y * Z * cin_stride > 0x7fffffffLL) return SIS3D_EUNSUPPORTED;      // 32-bit element offsets in the staging plan
    const bool shared_chip = (flags & SIS3D_DISPATCH_SHARED_CHIP) != 0;
    flags &= ~SIS3D_DISPATCH_SHARED_CHIP;
    WinoArgs a;
    for (int p = 0; p < WN_MAXP; ++p) {
        const int s = p < nprob ? p : 0;
        if (!ins[s] || !packed_ws[s] || !outs[s]) return SIS3D_EINVAL;
        a.in[p] = ins[s]; a.wp[p] = packed_ws[s]; a.bias[p] = biases ? biases[s] : nullptr; a.out[p] = outs[s];
    }
    static const int force_nc = [] { const char *e = getenv("SIS3D_WINO_NC"); return e ? atoi(e) : 0; }();      // tuning hook
    int nc = force_nc == 1 || force_nc == 2 ? force_nc : wino_nc(X, Y, Z, cin, cout, shared_chip);
    if (nc == 0) nc = 2;                                 // called directly on a layer the dispatch rule would not send here
    // two cout tiles per workgroup either as one wave with two tiles (nc 2, wc 1) or as two waves per SIMD with one each (nc 1, wc 2)
    int wc = 1;
#ifdef WN_WC2_EXPERIMENT
    static const int force_wc = [] { const char *e = getenv("SIS3D_WINO_WC"); return e ? atoi(e) : 0; }();      // tuning hook
    if (nc == 2 && force_wc == 2) { nc = 1; wc = 2; }
#endif
    a.X = X; a.Y = Y; a.Z = Z; a.cin_stride = cin_stride; a.cout = cout; a.ngroups = ((cout + 15) / 16 + nc * wc - 1) / (nc * wc); a.nk = cin / 4;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = out_coff;
    a.nbx = cdiv(X, VX); a.nby = cdiv(Y, VY); a.nbz = cdiv(Z, VZ);
    a.rag = nullptr; a.nrag = 0;
    return launch_wino(a, nc, wc, (int64_t)a.nbx * a.nby * a.nbz * a.ngroups, nprob, as_stream(stream), pig_state, pig_dst, pig_n4);
}

extern "C" int sis3d_conv3d_k3wino(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                   const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                                   int out_stride, int out_coff, sis3d_stream_t stream)
{
    return conv3d_k3wino_impl(nprob, ins, X, Y, Z, cin, cin_stride, packed_ws, biases, cout, flags, outs, out_stride, out_coff, nullptr, nullptr, 0,
                              stream);
}

// the same launch + the PIGGYBACK upload of a mailbox pipeline's next chunk (include/sis3d.h)
extern "C" int sis3d_conv3d_k3wino_piggyback(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                             const float *const *packed_ws, const float *const *biases, int cout, int flags,
                                             float *const *outs, int out_stride, int out_coff, uint32_t *mail_state, float *upload_dst,
                                             int64_t n, sis3d_stream_t stream)
{
    if (!mail_state || !upload_dst || n <= 0 || (n & 3) || ((uintptr_t)upload_dst & 15)) return SIS3D_EINVAL;
    return conv3d_k3wino_impl(nprob, ins, X, Y, Z, cin, cin_stride, packed_ws, biases, cout, flags, outs, out_stride, out_coff,
                              (unsigned *)mail_state, upload_dst, (long long)(n / 4), stream);
}

// ---- ragged batch: every detected box's mask-head crop (lib/nets/network.py:303-317, backbones.py:243-249) through ONE launch per k3
// layer.  Work items: (crop, 8 x 4 x 8 block, group of two cout tiles); the descriptor table is ops.MaskPlan's (block0 counts
// blocks x groups).  The caller asks sis3d_ragged_tiling_k3wino for the block and the group count first.
extern "C" int sis3d_ragged_tiling_k3wino(int cin, int cout, int *bx, int *by, int *bz, int *ngroups)
{
    if (!bx || !by || !bz || !ngroups) return SIS3D_EINVAL;
    if ((cin % 8) || cin <= 0 || cout <= 0) return SIS3D_EUNSUPPORTED;
    *bx = VX; *by = VY; *bz = VZ;
    *ngroups = ((cout + 15) / 16 + 1) / 2;
    return SIS3D_OK;
}

extern "C" int sis3d_conv3d_k3wino_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                          int flags, float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks,
                                          sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || !desc_dev || ndesc <= 0 || total_blocks <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % 8) || (cin_stride % 4) || cin_stride < cin || out_stride < cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    WinoArgs a;
    for (int p = 0; p < WN_MAXP; ++p) { a.in[p] = in; a.wp[p] = packed_w; a.bias[p] = bias; a.out[p] = out; }
    a.X = a.Y = a.Z = 1; a.cin_stride = cin_stride; a.cout = cout; a.ngroups = ((cout + 15) / 16 + 1) / 2; a.nk = cin / 4;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = 0;
    a.nbx = a.nby = a.nbz = 1;
    a.rag = (const WinoRagged *)desc_dev; a.nrag = ndesc;
    return launch_wino(a, 2, 1, total_blocks, 1, as_stream(stream));
}

// ---- ragged batch on MINI geometry (r4): work items = (crop, quad of 4 x 4 x 4 minis, group of two cout tiles).  Descriptor table as
// for sis3d_conv3d_k3wino_ragged, with nbx / nby / nbz = minis per axis (ceil(extent / 4)) and block0 counting quads x groups,
// quads = ceil(minis / 4).  sis3d_ragged_tiling_k3wino_mini gives the mini edge (4), the minis per work item (4) and the groups.
extern "C" int sis3d_ragged_tiling_k3wino_mini(int cin, int cout, int *mini_edge, int *minis_per_item, int *ngroups)
{
    if (!mini_edge || !minis_per_item || !ngroups) return SIS3D_EINVAL;
    if ((cin % 8) || cin <= 0 || cout <= 0) return SIS3D_EUNSUPPORTED;
    *mini_edge = 4; *minis_per_item = 4;
    *ngroups = ((cout + 15) / 16 + 1) / 2;
    return SIS3D_OK;
}

extern "C" int sis3d_conv3d_k3wino_ragged_mini(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                                               int flags, float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_items,
                                               sis3d_stream_t stream)
{
    if (!in || !packed_w || !out || !desc_dev || ndesc <= 0 || total_items <= 0 || cin <= 0 || cout <= 0) return SIS3D_EINVAL;
    if ((cin % 8) || (cin_stride % 4) || cin_stride < cin || out_stride < cout) return SIS3D_EUNSUPPORTED;
    if (flags & ~SIS3D_EPI_RELU) return SIS3D_EUNSUPPORTED;
    if (total_items > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    WinoArgs a;
    for (int p = 0; p < WN_MAXP; ++p) { a.in[p] = in; a.wp[p] = packed_w; a.bias[p] = bias; a.out[p] = out; }
    a.X = a.Y = a.Z = 1; a.cin_stride = cin_stride; a.cout = cout; a.ngroups = ((cout + 15) / 16 + 1) / 2; a.nk = cin / 4;
    a.flags = flags; a.out_stride = out_stride; a.out_coff = 0;
    a.nbx = a.nby = a.nbz = 1;
    a.rag = (const WinoRagged *)desc_dev; a.nrag = ndesc;
    a.w3p = a.b3 = a.res = a.w1n = a.b1n = nullptr; a.tout = a.out2 = nullptr;
    a.res_stride = a.tout_stride = a.tout_coff = a.out2_stride = 0;
    a.pig_state = nullptr; a.pig_dst = nullptr; a.pig_n4 = 0; a.pig_row = -1;
#ifdef WN_DEV_PLAIN_ONLY
    return SIS3D_EUNSUPPORTED;
#else
    return launch_wino_inst<2, 0, 0, true>(a, total_items, 1, as_stream(stream));
#endif
}

// ---- Bottleneck body on the Winograd kernel (lib/nets/backbones.py:17-40): conv2 = Conv3d(planes, planes, 3, padding=1) + bias + ReLU
// by F(2x2x2, 3x3x3), then -- on the output tile, before it leaves the CU -- conv3 (1x1x1) + bias + residual + ReLU and optionally the
// NEXT block's conv1 + bias + ReLU.  Same contract as sis3d_bottleneck16 (the direct-convolution form of the same fusion); serves the
// planes = 32 blocks, whose 32 conv2 channels are the two cout tiles of one workgroup.  sis3d_bottleneck_wino_prefer says where it is
// expected to win (enough 8 x 4 x 8 blocks to fill the chip: the 48 x 24 x 48 maps of geometry1 / color).
extern "C" int sis3d_bottleneck_wino_prefer(int X, int Y, int Z, int planes, int c3, int c2n, int shared_chip)
{
    if (planes != 32 || !(c3 == 32 || c3 == 64 || c3 == 128) || !(c2n == 0 || c2n == 32)) return 0;
    if (c3 != 32 && c2n != 0) return 0;
    const int64_t blocks = (int64_t)cdiv(X, VX) * cdiv(Y, VY) * cdiv(Z, VZ);
    // shared chip (the caller's regime, see sis3d_conv3d_k3wino_prefer): CU-time counts, not the launch's own duration -- the Bottleneck(128, 32) bodies
    // of the 24 x 12 x 24 maps as 27 fat work items instead of 256 thin ones (see wino_nc)
    if (shared_chip && blocks >= 24) return 1;
    if (c3 == 128) return 0;
    return blocks >= 200 ? 1 : 0;
}

extern "C" int sis3d_bottleneck_wino(const float *y1, int X, int Y, int Z, int planes, const float *w2_wino, const float *b2,
                                     const float *w3_pw16, const float *b3, int c3, const float *residual, int res_stride, float *out,
                                     int out_stride, int out_coff, const float *w1n_pw16, const float *b1n, int c2n, float *out2,
                                     sis3d_stream_t stream)
{
    if (!y1 || !w2_wino || !w3_pw16 || !residual || !out || X <= 0 || Y <= 0 || Z <= 0) return SIS3D_EINVAL;
    if (c2n < 0 || (c2n > 0 && (!w1n_pw16 || !out2))) return SIS3D_EINVAL;
    if (planes != 32 || res_stride < c3 || out_stride < out_coff + c3 || ((res_stride | out_stride | out_coff) & 3)) return SIS3D_EUNSUPPORTED;
    if ((int64_t)X * Y * Z * (planes > c3 ? planes : c3) > 0x7fffffffLL) return SIS3D_EUNSUPPORTED;
    WinoArgs a;
    for (int p = 0; p < WN_MAXP; ++p) { a.in[p] = y1; a.wp[p] = w2_wino; a.bias[p] = b2; a.out[p] = nullptr; }
    a.X = X; a.Y = Y; a.Z = Z; a.cin_stride = planes; a.cout = planes; a.ngroups = 1; a.nk = planes / 4;
    a.flags = SIS3D_EPI_RELU; a.out_stride = planes; a.out_coff = 0;
    a.nbx = cdiv(X, VX); a.nby = cdiv(Y, VY); a.nbz = cdiv(Z, VZ);
    a.rag = nullptr; a.nrag = 0;
    a.w3p = w3_pw16; a.b3 = b3; a.res = residual; a.res_stride = res_stride; a.tout = out; a.tout_stride = out_stride; a.tout_coff = out_coff;
    a.w1n = w1n_pw16; a.b1n = b1n; a.out2 = out2; a.out2_stride = c2n;
    a.pig_state = nullptr; a.pig_dst = nullptr; a.pig_n4 = 0; a.pig_row = -1;
    const int64_t nwg = (int64_t)a.nbx * a.nby * a.nbz;
    if (nwg > 0x7fffffff) return SIS3D_EUNSUPPORTED;
    const hipStream_t st = as_stream(stream);
#ifndef WN_DEV_PLAIN_ONLY                                   // (development builds of tools/wino_bench.cpp compile the plain kernels only)
    if (c3 == 32 && c2n == 0) return launch_wino_inst<2, 32, 0>(a, nwg, 1, st);
    if (c3 == 32 && c2n == 32) return launch_wino_inst<2, 32, 32>(a, nwg, 1, st);
    if (c3 == 64 && c2n == 0) return launch_wino_inst<2, 64, 0>(a, nwg, 1, st);
    // (128, 32) is not instantiated: with both weight sets of the tail live hipcc copies the kernel arguments to scratch and spills 43
    // registers (and the kernel faulted); the caller runs the next block's conv1 as its own pointwise launch instead (ops.bottleneck16)
    if (c3 == 128 && c2n == 0) return launch_wino_inst<2, 128, 0>(a, nwg, 1, st);
#endif
    (void)st;
    return SIS3D_EUNSUPPORTED;
}
