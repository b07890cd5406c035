// Stable descending top-k of the RPN proposal scores on gfx950 (one workgroup, one launch).
//
// Replaces `scores.sort(descending=True)` + `order[:pre_nms_topN]` of lib/layer_utils/proposal_layer.py:181-186:
// the reference sorts all ~33k candidates to keep 400.  Tie rule = the oracle's stable sort: equal scores keep
// ascending candidate index.  Method: 3-pass radix SELECT on the order-preserving integer image of the float
// (11+11+10 bits, LDS histograms) to find the k-th largest key, compaction of the <= k winners, then a bitonic
// sort of at most 1024 (key, index) pairs in LDS.  Integer-exact: the output order is bit-identical to
// torch.sort(stable=True, descending=True)[:k] for finite scores (NaN is ordered above +inf, as torch does).
// Fast path (the common case): after the first, data-adaptive bucketing pass the bucket holding the k-th score and
// everything above it usually amount to little more than k candidates; they are compacted and ordered by a RANK sort
// (every candidate counts the candidates ahead of it: one LDS sweep, no barrier ladder) -- 3 histogram passes and the
// 45-step bitonic network are skipped (39 -> ~12 us for 33k candidates, k = 400).
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t order_key(float f)
{
    // monotone map float -> uint (larger float => larger key); canonicalise NaN to the top like torch's descending sort
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    if (u == 0x80000000u) u = 0;                          // -0.0 == +0.0 (ties then break by index, as in torch)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}


// Pick the bucket that contains the `need`-th element counting from the TOP bucket down.  One wave: lane l owns the
// 32 bins [32l, 32l+32); lane partial sums -> suffix sums by shuffles -> the owning lane walks its 32 bins.
// (A single thread walking 2048 LDS bins serially costs ~50 us per pass.)  Returns through *b_out / *need_out.
__device__ __forceinline__ void pick_bucket(const uint32_t *hist, int nb, uint32_t need, uint32_t *b_out, uint32_t *need_out, int lane,
                                            uint32_t *cnt_out = nullptr)
{
    const int per = nb / 64;                            // 32 (2048 bins) or 16 (1024 bins)
    uint32_t part = 0;
    for (int j = 0; j < per; ++j) part += hist[lane * per + j];
    // above = number of elements in bins owned by HIGHER lanes
    uint32_t incl = part;                               // inclusive suffix sum over lanes >= lane
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(incl, o);
        if (lane + o < 64) incl += v;
    }
    const uint32_t above = incl - part;
    const bool mine = (above < need) && (incl >= need);
    // exactly one lane has `mine` unless the total is < need (then lane 0 takes the bottom bin)
    const uint64_t who = __ballot(mine);
    const int owner = who ? __builtin_ctzll(who) : 0;
    if (lane == owner) {
        uint32_t acc = above;
        int b = lane * per + per - 1;
        for (; b > lane * per; --b) {
            if (acc + hist[b] >= need) break;
            acc += hist[b];
        }
        if (!who) { b = 0; acc = incl - hist[0]; }
        *b_out = (uint32_t)b;
        *need_out = need - acc;
        if (cnt_out) *cnt_out = hist[b];
    }
}

constexpr int TPB = 1024;
constexpr int KMAX = 1024;
constexpr int FASTCAP = 1536;                         // candidates the rank-sort fast path orders (k plus one bucket's worth)
constexpr int EPT = 40;                                // elements per thread held in registers: n <= 40960

// scores [n]; out_scores [k], out_idx [k] (int64).  k <= KMAX.
#ifdef TOPK_TIMING      // tools/topk_phases.py: 100 MHz timestamps after each phase of the fast path
#define TOPK_TS(i) do { if (tid == 0) ts[i] = (long long)wall_clock64(); } while (0)
#define TOPK_TS_ARG , long long *ts
#else
#define TOPK_TS(i) do { } while (0)
#define TOPK_TS_ARG
#endif
__global__ __launch_bounds__(TPB) void topk_desc_kernel(const float *__restrict__ scores, int n, int k,
                                                        float *__restrict__ out_scores, int64_t *__restrict__ out_idx TOPK_TS_ARG)
{
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t s_prefix, s_need, s_cnt_gt, s_cnt_eq, s_bsel, s_eq_total;
    __shared__ uint64_t cand[KMAX];                    // (key << 32) | (~idx): sorting descending gives key desc, idx asc
    const int tid = threadIdx.x;
    TOPK_TS(0);
    if (k > n) k = n;
    // the whole score vector lives in registers (independent, fully pipelined loads); every later scan is on-chip.
    // A single workgroup re-reading global memory 6 times would pay ~200 dependent L2 round trips.
    float val[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TPB;
        val[j] = i < n ? scores[i] : 0.0f;
    }
    // ---- stage 0: data-adaptive linear bucketing.  RPN scores cluster (softmax outputs), so a radix digit taken from
    // the float's exponent would put most candidates in a handful of LDS histogram bins (atomic conflicts serialise
    // 64-fold).  Any MONOTONE bucket function is a valid first digit for selection: bucket = floor((s-lo)*scale)
    // over the observed finite [lo,hi] spreads the candidates over all 2048 bins; only the one bin that holds the
    // k-th value is then refined with exact radix passes on the integer keys.
    __shared__ float s_lo, s_hi;
    __shared__ uint32_t s_b1;
    {
        float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const float v = val[j];
            if (tid + j * TPB < n && fabsf(v) < __builtin_huge_valf()) { lo = fminf(lo, v); hi = fmaxf(hi, v); }    // finite only
        }
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        __shared__ float wlo[TPB / 64], whi[TPB / 64];
        if ((tid & 63) == 0) { wlo[tid >> 6] = lo; whi[tid >> 6] = hi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < TPB / 64; ++w) { lo = fminf(lo, wlo[w]); hi = fmaxf(hi, whi[w]); }
            s_lo = lo; s_hi = hi;
        }
        __syncthreads();
    }
    TOPK_TS(1);
    const float blo = s_lo, bhi = s_hi;
    const float bscale = (bhi > blo) ? 2047.0f / (bhi - blo) : 0.0f;
    auto bucket_of = [&](float v) -> uint32_t {
        if (!(v < bhi)) return 2047u;                     // >= hi, +inf, NaN
        if (!(v > blo)) return 0u;                        // <= lo, -inf
        const int b = (int)((v - blo) * bscale);
        return (uint32_t)(b < 0 ? 0 : (b > 2047 ? 2047 : b));
    };
    uint32_t need = (uint32_t)k;
    // ---- pre-filter: the k-th largest of the (<= 1024) per-thread maxima is a lower bound of the k-th largest score, so only
    // candidates in its bucket or above can be among the top k.  RPN scores pile up near zero: histogramming all 33k of them
    // means ~33k LDS atomics, most on a handful of bins (serialised); after the filter a few hundred remain.
    __shared__ uint32_t s_bt;
    {
        for (int i = tid; i < 2048; i += TPB) hist[i] = 0;
        if (tid == 0) s_bt = 0;
        __syncthreads();
        float tmax = -__builtin_huge_valf();                // NaNs are skipped by fmaxf: the bound only gets lower, still valid
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (tid + j * TPB < n) tmax = fmaxf(tmax, val[j]);
        if (tid < n) atomicAdd(&hist[bucket_of(tmax)], 1u);
        __syncthreads();
        uint32_t dummy_need;
        if (tid < 64 && (uint32_t)min(n, TPB) >= need) pick_bucket(hist, 2048, need, &s_bt, &s_need, tid);
        (void)dummy_need;
        __syncthreads();
    }
    const uint32_t bt = s_bt;
    __syncthreads();
    TOPK_TS(2);
    // ---- fast path: every element in bucket bt or above (one float compare per element -- this single workgroup's VALU time
    // over 33k elements is what the kernel costs) is compacted into LDS and ordered by a rank sort; >= k of them by construction
    {
        __shared__ uint32_t s_m;
        __shared__ uint64_t cand2[FASTCAP];
        if (tid == 0) s_m = 0;
        __syncthreads();
        // ANY superset of {bucket_of(v) >= bt} will do (the ordering below is exact on the composite keys), so the test is ONE float
        // compare against a threshold a whole bucket below bt's lower edge: v >= thr is implied by bucket_of(v) >= bt whatever
        // the rounding of (v - lo) * scale; NaN passes (!(v < thr)); bt <= 1 or a degenerate range takes everything.  r4: the
        // three-clause test this replaces cost the single workgroup 5.2 us over 33k elements (tools/topk_phases.py).
        const float thr = (bt <= 1 || !(bscale > 0.f)) ? -__builtin_huge_valf() : blo + ((float)bt - 1.0f) / bscale;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i = tid + j * TPB;
            const float v = val[j];
            if (i < n && !(v < thr)) {
                const uint32_t slot = atomicAdd(&s_m, 1u);
                if (slot < FASTCAP) cand2[slot] = ((uint64_t)order_key(v) << 32) | (uint32_t)(~(uint32_t)i);
            }
        }
        __syncthreads();
        const int m = (int)s_m;
        TOPK_TS(3);
#ifdef TOPK_TIMING
        if (tid == 0) ts[8] = m;
#endif
        if (m <= 1024) {
            // r4: bitonic network over the m candidates padded with zeros (smaller than every composite) to a power of two: 36-55
            // steps of one compare-exchange per thread; the rank sort it replaces makes every thread sweep m / P candidates with 64-bit
            // compares (8.7 us for m = 481 on one workgroup's VALUs)
            int p2 = 64;
            while (p2 < m) p2 <<= 1;
            for (int i = m + tid; i < p2; i += TPB) cand2[i] = 0;
            __syncthreads();
            for (int size = 2; size <= p2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    if (tid < p2 / 2) {
                        const int lo = 2 * tid - (tid & (stride - 1));
                        const int hi = lo + stride;
                        const bool desc = ((lo & size) == 0);
                        const uint64_t x = cand2[lo], y = cand2[hi];
                        if ((x < y) == desc) { cand2[lo] = y; cand2[hi] = x; }
                    }
                    __syncthreads();           // (wave-level fences for the stride <= 64 steps were measured: no gain, the steps are LDS-latency bound)
                }
            for (int i = tid; i < k; i += TPB) {
                const uint32_t idx = ~(uint32_t)(cand2[i] & 0xffffffffu);
                out_idx[i] = (int64_t)idx;
                out_scores[i] = scores[idx];
            }
            TOPK_TS(4);
            return;
        }
        if (m <= FASTCAP) {
            // composite keys are unique (index in the low word): rank = number of strictly larger composites.  P threads
            // share a candidate (each sweeps 1/P of the table, partial counts meet by lane shuffles)
            const int P = m <= 256 ? 4 : (m <= 512 ? 2 : 1);
            for (int base = 0; base < m; base += TPB / P) {
                const int i = base + tid / P, part = tid % P;
                const bool on = i < m;
                const uint64_t me = on ? cand2[i] : 0;
                const int j0 = (int)((int64_t)m * part / P), j1 = (int)((int64_t)m * (part + 1) / P);
                int rank = 0;
                int j = j0;
                for (; j + 8 <= j1; j += 8) {
                    const uint64_t c0 = cand2[j], c1 = cand2[j + 1], c2 = cand2[j + 2], c3 = cand2[j + 3];
                    const uint64_t c4 = cand2[j + 4], c5 = cand2[j + 5], c6 = cand2[j + 6], c7 = cand2[j + 7];
                    rank += (c0 > me) + (c1 > me) + (c2 > me) + (c3 > me) + (c4 > me) + (c5 > me) + (c6 > me) + (c7 > me);
                }
                for (; j < j1; ++j) rank += cand2[j] > me;
                if (P >= 2) rank += __shfl_xor(rank, 1);
                if (P >= 4) rank += __shfl_xor(rank, 2);
                if (on && part == 0 && rank < k) {
                    const uint32_t idx = ~(uint32_t)(me & 0xffffffffu);
                    out_idx[rank] = (int64_t)idx;
                    out_scores[rank] = scores[idx];
                }
            }
            TOPK_TS(4);
            return;
        }
        __syncthreads();
    }
    // ---- slow path (heavy ties / clustered scores: more than FASTCAP candidates share the top buckets): exact selection.
    // stage 0: histogram of the buckets >= bt, bucket b1 of the k-th score
    {
        for (int i = tid; i < 2048; i += TPB) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (tid + j * TPB < n) {
                const uint32_t b = bucket_of(val[j]);
                if (b >= bt) atomicAdd(&hist[b], 1u);
            }
        __syncthreads();
        if (tid < 64) pick_bucket(hist, 2048, need, &s_b1, &s_need, tid);
        __syncthreads();
        need = s_need;
    }
    const uint32_t b1 = s_b1;
    __syncthreads();
    // ---- stage 1: exact radix select (11+11+10 bits of the order-preserving key) inside bucket b1
    uint32_t prefix = 0, mask = 0;
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass], nb = 1 << bits[pass];
        for (int i = tid; i < 2048; i += TPB) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const float v = val[j];
            if (tid + j * TPB >= n || bucket_of(v) != b1) continue;
            const uint32_t key = order_key(v);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid < 64) pick_bucket(hist, nb, need, &s_bsel, &s_need, tid, &s_eq_total);
        __syncthreads();
        if (tid == 0) s_prefix = prefix | (s_bsel << sh);
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        mask |= (uint32_t)(nb - 1) << sh;
        __syncthreads();
    }
    const uint32_t kth = prefix;                       // exact key of the k-th largest element; `need` of them are equal to it
    // ---- compaction: every key > kth, plus the `need` lowest-index elements with key == kth
    if (tid == 0) { s_cnt_gt = 0; s_cnt_eq = 0; }
    __syncthreads();
    const uint32_t n_gt = (uint32_t)k - need;
    const uint32_t eq_total = s_eq_total;               // elements equal to the k-th key (last radix pass resolved all 32 bits)
    if (eq_total == need) {
        // common case (no tie straddles the cut): every key >= kth wins, order is fixed by the sort below
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i = tid + j * TPB;
            if (i >= n) break;
            const uint32_t key = order_key(val[j]);
            if (key >= kth) cand[atomicAdd(&s_cnt_gt, 1u)] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)i);
        }
    } else {
    // ties straddle the cut: equal-key elements must be taken in ascending index order, chunk by chunk
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int base = j * TPB;
        if (base >= n) break;
        const int i = base + tid;
        uint32_t key = 0;
        bool gt = false, eq = false;
        if (i < n) {
            key = order_key(val[j]);
            gt = key > kth;
            eq = key == kth;
        }
        if (gt) {
            const uint32_t slot = atomicAdd(&s_cnt_gt, 1u);
            cand[slot] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)i);
        }
        // ranks of the equal elements inside this chunk: wave ballot prefix + per-wave offsets through LDS
        const uint64_t bal = __ballot(eq);
        const int lane = tid & 63, wv = tid >> 6;
        __shared__ uint32_t wave_cnt[TPB / 64];
        if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (eq) {
            uint32_t off = s_cnt_eq;
            for (int w = 0; w < wv; ++w) off += wave_cnt[w];
            off += (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
            if (off < need) cand[n_gt + off] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)i);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int w = 0; w < TPB / 64; ++w) t += wave_cnt[w];
            s_cnt_eq += t;
        }
        __syncthreads();
    }
    }
    __syncthreads();
    // ---- bitonic sort (descending) of the k candidates, padded with zeros to a power of two
    int p2 = 1;
    while (p2 < k) p2 <<= 1;
    for (int i = k + tid; i < p2; i += TPB) cand[i] = 0;
    __syncthreads();
    for (int size = 2; size <= p2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < p2 / 2; t += TPB) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += TPB) {
        const uint32_t idx = ~(uint32_t)(cand[i] & 0xffffffffu);
        out_idx[i] = (int64_t)idx;
        out_scores[i] = scores[idx];
    }
}

} // namespace

extern "C" int sis3d_topk_desc(const float *scores, int n, int k, float *out_scores, int64_t *out_idx, sis3d_stream_t stream)
{
    if (n < 0 || k < 0 || k > KMAX) return SIS3D_EINVAL;
    if (n > EPT * TPB) return SIS3D_EUNSUPPORTED;        // caller falls back to a full sort
    if (n == 0 || k == 0) return SIS3D_OK;
    if (!scores || !out_scores || !out_idx) return SIS3D_EINVAL;
#ifdef TOPK_TIMING
    (void)stream;
    return SIS3D_EUNSUPPORTED;      // a timing build's kernel takes the timestamp buffer: this entry must not report success without a launch
#else
    hipLaunchKernelGGL(topk_desc_kernel, dim3(1), dim3(TPB), 0, as_stream(stream), scores, n, k, out_scores, out_idx);
    return sis3d_check_launch();
#endif
}

#ifdef TOPK_TIMING
extern "C" int sis3d_topk_desc_timing(const float *scores, int n, int k, float *out_scores, int64_t *out_idx, long long *ts, sis3d_stream_t stream)
{
    hipLaunchKernelGGL(topk_desc_kernel, dim3(1), dim3(TPB), 0, as_stream(stream), scores, n, k, out_scores, out_idx, ts);
    return sis3d_check_launch();
}
#endif
