// 3D RoI max pooling on gfx950.
//
// Replaces ROIPoolForward (lib/layer_utils/roi_pooling/src/cuda/roi_pooling_kernel.cu:15-109)
// and the CPU loop nest of lib/layer_utils/roi_pooling/src/roi_pooling.c:6-124; the
// two-level entry point also folds in the split-by-level / scatter-back Python loop of
// Network._roi_pool_layer (lib/nets/network.py:503-534).
//
// Bin geometry is the reference's binary32 arithmetic verbatim (floor/ceil of
// roi*scale, bin = len/pooled, windows clamped to the map); the max uses a strict
// '>' in w->h->l scan order so values AND argmax indices are bit-identical.  Built
// with -ffp-contract=off.
//
// Mapping (MI355X): the reference launches one thread per output element with the
// channel as the SLOWEST-varying index inside a RoI, so neighbouring lanes walk
// different windows of different channels.  Here a workgroup owns one (roi, bin):
// the window bounds are wave-uniform (SALU), lanes run across channels, and with
// channels-last features every window voxel is one contiguous, fully coalesced row
// read that all 64 lanes consume.  The 3.5 MB maps stay L2-resident.
#include "common.h"
#include <float.h>
#include <stdlib.h>

namespace {

struct RoiGeom { int rs_w, rs_h, rs_l; float bw, bh, bl; };

__device__ __forceinline__ RoiGeom roi_geom(const float *r, float scale, int pw_, int ph_, int pl_)
{
    RoiGeom g;
    g.rs_w = (int)floorf(r[0] * scale);
    g.rs_h = (int)floorf(r[1] * scale);
    g.rs_l = (int)floorf(r[2] * scale);
    const int re_w = (int)ceilf(r[3] * scale), re_h = (int)ceilf(r[4] * scale), re_l = (int)ceilf(r[5] * scale);
    const int rw = max(re_w - g.rs_w, 1), rh = max(re_h - g.rs_h, 1), rl = max(re_l - g.rs_l, 1);
    g.bw = (float)rw / (float)pw_;
    g.bh = (float)rh / (float)ph_;
    g.bl = (float)rl / (float)pl_;
    return g;
}

__device__ __forceinline__ void bin_range(int p, float bin, int start, int dim, int &s, int &e)
{
    s = (int)floorf((float)p * bin);
    e = (int)ceilf((float)(p + 1) * bin);
    s = min(max(s + start, 0), dim);
    e = min(max(e + start, 0), dim);
}

// grid: (bins, R); block: 64*k threads over channels
template <bool LEVELS>
__global__ void roi_pool_kernel(const float *__restrict__ f1, const float *__restrict__ f2, int C, int W, int H, int L,
                                int64_t fs_c, int64_t fs_w, int64_t fs_h, int64_t fs_l, const float *__restrict__ rois,
                                const float *__restrict__ levels, int PW, int PH, int PL, float scale, float *__restrict__ out,
                                int32_t *__restrict__ argmax, int64_t os_n, int64_t os_c, int64_t os_bin)
{
    const int bin = blockIdx.x, n = blockIdx.y;
    const int pl = bin % PL, ph = (bin / PL) % PH, pw = bin / (PL * PH);
    const float *feat = f1;
    bool live = true;
    if (LEVELS) {
        const float lv = levels[n];
        if (lv == 1.0f) feat = f1;
        else if (lv == 2.0f) feat = f2;
        else live = false;
    }
    int ws = 0, we = 0, hs = 0, he = 0, ls = 0, le = 0;
    if (live) {
        const RoiGeom g = roi_geom(rois + 6 * n, scale, PW, PH, PL);
        bin_range(pw, g.bw, g.rs_w, W, ws, we);
        bin_range(ph, g.bh, g.rs_h, H, hs, he);
        bin_range(pl, g.bl, g.rs_l, L, ls, le);
    }
    const bool empty = !live || (he <= hs) || (we <= ws) || (le <= ls);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float mx = empty ? 0.0f : -FLT_MAX;
        int mi = -1;
        if (!empty) {
            // the window is walked in w -> h -> l order as a flat index, eight voxels per step: their loads are requested
            // together (addresses clamped to the last voxel) and then compared in order with the strict '>' of the
            // reference -- one exposed L2 latency per eight voxels instead of one per voxel, same values and argmax
            const float *fc = feat + (int64_t)c * fs_c;
            const int nh = he - hs, nl = le - ls, nwin = (we - ws) * nh * nl;
            for (int i0 = 0; i0 < nwin; i0 += 8) {
                float v[8];
                int id[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = min(i0 + j, nwin - 1);
                    const int w = ws + i / (nh * nl), h = hs + (i / nl) % nh, l = ls + i % nl;
                    v[j] = fc[(int64_t)w * fs_w + (int64_t)h * fs_h + (int64_t)l * fs_l];
                    id[j] = (c * W + w) * H * L + h * L + l;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + j < nwin && v[j] > mx) { mx = v[j]; mi = id[j]; }
            }
        }
        const int64_t o = (int64_t)n * os_n + (int64_t)c * os_c + (int64_t)bin * os_bin;
        out[o] = mx;
        if (argmax) argmax[o] = mi;
    }
}

// ---- two-level pooling on channels-last maps, values only (the network's path: no argmax is kept in TEST mode).
// The bin-per-workgroup form above launches bins x R = 12,800 two-wave workgroups whose life is one or two dependent L2 round
// trips: the launch rate and the exposed latency, not bytes, set its 22 us.  Here a workgroup owns (RoI, pw): its 4 waves
// take the PH*PL bins of that slab four at a time, lanes hold V consecutive channels (C = 64 V), and the first voxels of all
// four windows are requested together (16 loads in flight per lane).  max() is order-independent, so the values equal the
// scan-order maximum of the reference bit for bit (up to the sign of a zero, which no comparison sees).
template <int V> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float2 T; };
template <> struct VecOf<4> { typedef float4 T; };

template <int V>
__global__ __launch_bounds__(256) void roi_pool_slab_kernel(const float *__restrict__ f1, const float *__restrict__ f2, int W, int H, int L,
                                                            int64_t fs_w, int64_t fs_h, int64_t fs_l, const float *__restrict__ rois,
                                                            const float *__restrict__ levels, int PW, int PH, int PL, float scale,
                                                            float *__restrict__ out, int64_t os_n, int64_t os_bin)
{
    typedef typename VecOf<V>::T vec;
    const int pw = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float lv = levels[n];
    const float *feat = lv == 1.0f ? f1 : f2;
    const bool live = lv == 1.0f || lv == 2.0f;
    const RoiGeom g = roi_geom(rois + 6 * n, scale, PW, PH, PL);   // read beside the level (one round trip, not two); unused when !live
    int ws = 0, we = 0;
    if (live) bin_range(pw, g.bw, g.rs_w, W, ws, we);
    const int nb = PH * PL;
    for (int b0 = 4 * wave; b0 < nb; b0 += 16) {
        int hs[4], ls[4], nh[4], nl[4], nwin[4];
        int maxwin = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + k;
            int he = 0, le = 0;
            hs[k] = ls[k] = 0;
            if (live && b < nb) {
                bin_range(b / PL, g.bh, g.rs_h, H, hs[k], he);
                bin_range(b % PL, g.bl, g.rs_l, L, ls[k], le);
            }
            nh[k] = max(he - hs[k], 0);
            nl[k] = max(le - ls[k], 0);
            nwin[k] = max(we - ws, 0) * nh[k] * nl[k];
            maxwin = max(maxwin, nwin[k]);
        }
        float mx[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) mx[k][e] = nwin[k] > 0 ? -FLT_MAX : 0.0f;
        // r6: the window is walked with three carried counters per bin (wave-uniform: scalar adds and compares) -- the flat-index
        // form divided three times per voxel, ~1,600 instructions per step of 16 loads, and THAT was the kernel's 11 us
        int cw[4], chh[4], cll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cw[k] = ws; chh[k] = hs[k]; cll[k] = ls[k]; }
        for (int i0 = 0; i0 < maxwin; i0 += 4) {
            vec v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = nwin[k] > 0;
                    v[k][j] = *reinterpret_cast<const vec *>(feat + (ok ? (int64_t)cw[k] * fs_w + (int64_t)chh[k] * fs_h + (int64_t)cll[k] * fs_l : 0) + V * lane);
                    if (i0 + j + 1 < nwin[k]) {                    // next voxel in w -> h -> l order; the last one is re-read, never passed
                        if (++cll[k] == ls[k] + nl[k]) {
                            cll[k] = ls[k];
                            if (++chh[k] == hs[k] + nh[k]) { chh[k] = hs[k]; ++cw[k]; }
                        }
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < nwin[k]) {
                        const float *pv = reinterpret_cast<const float *>(&v[k][j]);
#pragma unroll
                        for (int e = 0; e < V; ++e) mx[k][e] = fmaxf(mx[k][e], pv[e]);
                    }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + k;
            if (b >= nb) continue;
            vec o;
            float *po = reinterpret_cast<float *>(&o);
#pragma unroll
            for (int e = 0; e < V; ++e) po[e] = mx[k][e];
            *reinterpret_cast<vec *>(out + (int64_t)n * os_n + (int64_t)(pw * nb + b) * os_bin + V * lane) = o;
        }
    }
}

// ---- backward (training only).  Replaces ROIPoolBackward (roi_pooling_kernel.cu:137-248): the reference visits
// EVERY input element and scans all RoIs x bins for argmax == index, adding grad_output in (RoI, bin) ascending order.
// r6 -- DETERMINISTIC and in the reference's order (r2-r5 scattered with float atomics: right to 1e-5, order
// left to the hardware): a workgroup owns a SLAB of S consecutive voxels x all channels, held in LDS; every lane owns channels
// (lane, lane + 256, ...) and walks the whole (RoI, bin) list in ascending order -- coalesced argmax rows, eight entries in flight --
// adding the entries that point into its slab to ITS OWN LDS cells: no atomics, no two lanes on one cell, and each cell's sum is
// built in exactly the order of the reference's loop (roi_n ascending, pw / ph / pl ascending), so the result is the reference's
// bit for bit.  The slab is then ADDED to grad_in (the cffi entry point accumulates into the caller's tensor: dropin.py).
__global__ __launch_bounds__(256) void roi_pool_backward_slab_kernel(const float *__restrict__ gout, const int32_t *__restrict__ argmax,
                                                                     int64_t nent, int C, int nb, int64_t os_n, int64_t os_c, int64_t os_bin,
                                                                     int W, int H, int L, int S, float *__restrict__ gin, int64_t gs_c,
                                                                     int64_t gs_w, int64_t gs_h, int64_t gs_l)
{
    extern __shared__ float slab[];                           // [S][C]
    const int nsp = W * H * L;
    const int sp0 = blockIdx.x * S, sp1 = min(sp0 + S, nsp);
    for (int i = threadIdx.x; i < S * C; i += blockDim.x) slab[i] = 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int64_t oc = (int64_t)c * os_c;
        const int lo = c * nsp + sp0, hi = c * nsp + sp1;     // this channel's element indices inside the slab (argmax encodes the channel)
        constexpr int U = 8;
        int64_t e = 0;
        for (; e + U <= nent; e += U) {
            int idx[U];
            int64_t o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t n = (e + u) / nb;
                o[u] = n * os_n + oc + (int64_t)((e + u) - n * nb) * os_bin;
                idx[u] = argmax[o[u]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)                        // ascending entry order: the reference's summation order
                if (idx[u] >= lo && idx[u] < hi) slab[(idx[u] - lo) * C + c] += gout[o[u]];
        }
        for (; e < nent; ++e) {
            const int64_t n = e / nb;
            const int64_t o = n * os_n + oc + (int64_t)(e - n * nb) * os_bin;
            const int idx = argmax[o];
            if (idx >= lo && idx < hi) slab[(idx - lo) * C + c] += gout[o];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (sp1 - sp0) * C; i += blockDim.x) {
        const int c = i % C, sp = sp0 + i / C;
        const int l = sp % L, h = (sp / L) % H, w = sp / (L * H);
        float *dst = gin + (int64_t)c * gs_c + (int64_t)w * gs_w + (int64_t)h * gs_h + (int64_t)l * gs_l;
        *dst = *dst + slab[i];
    }
}

} // namespace

extern "C" int sis3d_roi_pool_backward(const float *grad_out, const int32_t *argmax, int R, int C, int pw, int ph, int pl, int64_t os_n,
                                       int64_t os_c, int64_t os_bin, int W, int H, int L, float *grad_in, int64_t gs_c, int64_t gs_w,
                                       int64_t gs_h, int64_t gs_l, sis3d_stream_t stream)
{
    if (R < 0 || C <= 0 || pw <= 0 || ph <= 0 || pl <= 0 || W <= 0 || H <= 0 || L <= 0) return SIS3D_EINVAL;
    if (R == 0) return SIS3D_OK;
    if (!grad_out || !argmax || !grad_in) return SIS3D_EINVAL;
    const int nb = pw * ph * pl;
    if ((int64_t)C * W * H * L > 0x7fffffffLL) return SIS3D_EUNSUPPORTED;         // argmax is a 32-bit element index
    // slab: as many voxels as 64 KB of LDS hold for C channels; one workgroup per slab
    int S = 16384 / C;
    if (S < 1) return SIS3D_EUNSUPPORTED;
    if (S > W * H * L) S = W * H * L;
    const int nslab = cdiv((int64_t)W * H * L, S);
    hipLaunchKernelGGL(roi_pool_backward_slab_kernel, dim3((unsigned)nslab), dim3(256), (size_t)S * C * sizeof(float), as_stream(stream), grad_out,
                       argmax, (int64_t)R * nb, C, nb, os_n, os_c, os_bin, W, H, L, S, grad_in, gs_c, gs_w, gs_h, gs_l);
    return sis3d_check_launch();
}

extern "C" int sis3d_roi_pool_forward(const float *features, int C, int W, int H, int L, int64_t fs_c, int64_t fs_w,
                                      int64_t fs_h, int64_t fs_l, const float *rois, int R, int pw, int ph, int pl,
                                      float scale, float *out, int32_t *argmax, int64_t os_n, int64_t os_c, int64_t os_bin,
                                      sis3d_stream_t stream)
{
    if (!features || !out || C <= 0 || W <= 0 || H <= 0 || L <= 0 || pw <= 0 || ph <= 0 || pl <= 0 || R < 0) return SIS3D_EINVAL;
    if (R == 0) return SIS3D_OK;
    if (!rois) return SIS3D_EINVAL;
    const int threads = C >= 256 ? 256 : (C > 64 ? 128 : 64);
    hipLaunchKernelGGL((roi_pool_kernel<false>), dim3(pw * ph * pl, R), dim3(threads), 0, as_stream(stream), features, nullptr, C,
                       W, H, L, fs_c, fs_w, fs_h, fs_l, rois, nullptr, pw, ph, pl, scale, out, argmax, os_n, os_c, os_bin);
    return sis3d_check_launch();
}

extern "C" int sis3d_roi_pool_levels(const float *f1, const float *f2, int C, int W, int H, int L, int64_t fs_c, int64_t fs_w,
                                     int64_t fs_h, int64_t fs_l, const float *rois, const float *levels, int R, int pooled,
                                     float scale, float *out, int64_t os_n, int64_t os_c, int64_t os_bin, sis3d_stream_t stream)
{
    if (!f1 || !f2 || !out || !levels || C <= 0 || W <= 0 || H <= 0 || L <= 0 || pooled <= 0 || R < 0) return SIS3D_EINVAL;
    if (R == 0) return SIS3D_OK;
    if (!rois) return SIS3D_EINVAL;
    // channels-last maps and rows (the network's layout): the slab kernel.  It moves C/64 floats per lane with one vector
    // access, so strides AND base pointers must be multiples of that vector (a channel-slice view need not be)
    static const bool legacy = getenv("SIS3D_ROIPOOL_LEGACY") != nullptr;
    const uintptr_t vec_bytes = (uintptr_t)(C / 64) * 4;
    const bool aligned = (((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)out) % (vec_bytes < 4 ? 4 : vec_bytes)) == 0;
    if (fs_c == 1 && os_c == 1 && (C == 64 || C == 128 || C == 256) && (fs_w % 4) == 0 && (fs_h % 4) == 0 && (fs_l % 4) == 0 &&
        (os_n % 4) == 0 && (os_bin % 4) == 0 && aligned && !legacy) {
        const dim3 grid(pooled, R), block(256);
        hipStream_t st = as_stream(stream);
        if (C == 64) hipLaunchKernelGGL((roi_pool_slab_kernel<1>), grid, block, 0, st, f1, f2, W, H, L, fs_w, fs_h, fs_l, rois, levels, pooled, pooled, pooled, scale, out, os_n, os_bin);
        else if (C == 128) hipLaunchKernelGGL((roi_pool_slab_kernel<2>), grid, block, 0, st, f1, f2, W, H, L, fs_w, fs_h, fs_l, rois, levels, pooled, pooled, pooled, scale, out, os_n, os_bin);
        else hipLaunchKernelGGL((roi_pool_slab_kernel<4>), grid, block, 0, st, f1, f2, W, H, L, fs_w, fs_h, fs_l, rois, levels, pooled, pooled, pooled, scale, out, os_n, os_bin);
        return sis3d_check_launch();
    }
    const int threads = C >= 256 ? 256 : (C > 64 ? 128 : 64);
    hipLaunchKernelGGL((roi_pool_kernel<true>), dim3(pooled * pooled * pooled, R), dim3(threads), 0, as_stream(stream), f1, f2, C,
                       W, H, L, fs_c, fs_w, fs_h, fs_l, rois, levels, pooled, pooled, pooled, scale, out, nullptr, os_n, os_c,
                       os_bin);
    return sis3d_check_launch();
}
