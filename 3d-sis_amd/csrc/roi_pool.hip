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

} // namespace

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
    const int threads = C >= 256 ? 256 : (C > 64 ? 128 : 64);
    hipLaunchKernelGGL((roi_pool_kernel<true>), dim3(pooled * pooled * pooled, R), dim3(threads), 0, as_stream(stream), f1, f2, C,
                       W, H, L, fs_c, fs_w, fs_h, fs_l, rois, levels, pooled, pooled, pooled, scale, out, nullptr, os_n, os_c,
                       os_bin);
    return sis3d_check_launch();
}
