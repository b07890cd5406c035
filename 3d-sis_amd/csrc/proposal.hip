// RPN proposal decoding on gfx950 (one launch per pyramid level).
//
// Replaces, for the TEST path, the torch op chain of
// lib/layer_utils/proposal_layer.py:96-103 (gather by inds_inside, view(-1,6)) plus
// bbox_transform_inv / clip_boxes (lib/utils/bbox_transform.py:59-99, 4-21): ~25 small
// kernels and their temporaries become one pass that reads each inside anchor's 6 deltas +
// 1 score and writes box / score / level id.  Arithmetic order is the reference's (separate
// multiply and add, no FMA: this file is built with -ffp-contract=off); expf differs from
// torch-CPU's vectorised exp by <= 2 ulp, inside the fp32 1e-4 box tolerance.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void decode_kernel(const float *__restrict__ anchors, const float *__restrict__ deltas,
                                                     const float *__restrict__ prob_fg, const int32_t *__restrict__ inside,
                                                     int n, float dx_, float dy_, float dz_, float level,
                                                     float *__restrict__ boxes, float *__restrict__ scores,
                                                     float *__restrict__ levels)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = inside[k];
    const float *a = anchors + 6 * (int64_t)i;
    const float *d = deltas + 6 * (int64_t)i;
    const float w = a[3] - a[0], h = a[4] - a[1], l = a[5] - a[2];
    const float cx = a[0] + 0.5f * w, cy = a[1] + 0.5f * h, cz = a[2] + 0.5f * l;
    const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pcz = d[2] * l + cz;
    const float pw = expf(d[3]) * w, ph = expf(d[4]) * h, pl = expf(d[5]) * l;
    float *o = boxes + 6 * (int64_t)k;
    o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), dx_);
    o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), dy_);
    o[2] = fminf(fmaxf(pcz - 0.5f * pl, 0.0f), dz_);
    o[3] = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), dx_);
    o[4] = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), dy_);
    o[5] = fminf(fmaxf(pcz + 0.5f * pl, 0.0f), dz_);
    scores[k] = prob_fg[i];
    levels[k] = level;
}

// r5: both pyramid levels in ONE launch (a kernel boundary costs a pipeline ~15-20 us of waiting for a CU when four chunks are in flight:
// every Winograd workgroup owns its CU): blocks [0, nb1) decode level 1 into rows [0, n1), the rest level 2 into rows [n1, n1 + n2)
struct Decode2Args {
    const float *anchors[2], *deltas[2], *prob_fg[2];
    const int32_t *inside[2];
    int n[2];
    float level[2];
};
__global__ __launch_bounds__(256) void decode2_kernel(const Decode2Args a, int nb1, float dx_, float dy_, float dz_, float *__restrict__ boxes,
                                                      float *__restrict__ scores, float *__restrict__ levels)
{
    const int lv = (int)blockIdx.x >= nb1 ? 1 : 0;
    const int kk = ((int)blockIdx.x - (lv ? nb1 : 0)) * blockDim.x + threadIdx.x;
    if (kk >= a.n[lv]) return;
    const int i = a.inside[lv][kk];
    const float *an = a.anchors[lv] + 6 * (int64_t)i;
    const float *d = a.deltas[lv] + 6 * (int64_t)i;
    const float w = an[3] - an[0], h = an[4] - an[1], l = an[5] - an[2];
    const float cx = an[0] + 0.5f * w, cy = an[1] + 0.5f * h, cz = an[2] + 0.5f * l;
    const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pcz = d[2] * l + cz;
    const float pw = expf(d[3]) * w, ph = expf(d[4]) * h, pl = expf(d[5]) * l;
    const int64_t k = kk + (lv ? a.n[0] : 0);
    float *o = boxes + 6 * k;
    o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), dx_);
    o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), dy_);
    o[2] = fminf(fmaxf(pcz - 0.5f * pl, 0.0f), dz_);
    o[3] = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), dx_);
    o[4] = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), dy_);
    o[5] = fminf(fmaxf(pcz + 0.5f * pl, 0.0f), dz_);
    scores[k] = a.prob_fg[lv][i];
    levels[k] = a.level[lv];
}

// F.softmax over the 2-class dim of (1,2,n) (lib/nets/network.py:546): same max-subtract form as torch
__global__ __launch_bounds__(256) void softmax2_kernel(const float *__restrict__ s, float *__restrict__ p, int64_t n)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = s[i], b = s[n + i];
    const float m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    const float sum = ea + eb;
    p[i] = ea / sum;
    p[n + i] = eb / sum;
}

// One detection record per padded RoI row (16 floats): proposal box, RPN score, level, class id, class probability and the
// class-specific regressed box of lib/model/trainval.py:686-700 / network.py:285-294 (box_reg row of the arg-max class ->
// bbox_transform_inv -> clip_boxes).  `records` holds chunk coordinates; `block` = [count, rows shifted by the chunk origin
// to scene coordinates, rows >= count zeroed] is the fixed-size unit of the per-scene all-gather.
struct PackArgs {
    const float *rois, *scores, *levels;
    const int64_t *cls_pred;
    const float *cls_prob, *bbox_pred;
    const int32_t *num;
    const float *origin;
    int K, NC;
    float dx, dy, dz;
    float *records, *block;
};

__device__ __forceinline__ void pack_row(const PackArgs &p, int k, int n)
{
    const float *a = p.rois + 6 * (int64_t)k;
    float r[SIS3D_RECORD_WIDTH];
#pragma unroll
    for (int j = 0; j < 6; ++j) r[j] = a[j];
    r[6] = p.scores[k];
    r[7] = p.levels[k];
    if (p.cls_pred) {
        const int c = (int)p.cls_pred[k];
        r[8] = (float)c;
        r[9] = p.cls_prob[(int64_t)k * p.NC + c];
        const float *d = p.bbox_pred + (int64_t)k * (6 * p.NC) + 6 * c;
        const float w = a[3] - a[0], h = a[4] - a[1], l = a[5] - a[2];
        const float cx = a[0] + 0.5f * w, cy = a[1] + 0.5f * h, cz = a[2] + 0.5f * l;
        const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pcz = d[2] * l + cz;
        const float pw = expf(d[3]) * w, ph = expf(d[4]) * h, pl = expf(d[5]) * l;
        r[10] = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), p.dx);
        r[11] = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), p.dy);
        r[12] = fminf(fmaxf(pcz - 0.5f * pl, 0.0f), p.dz);
        r[13] = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), p.dx);
        r[14] = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), p.dy);
        r[15] = fminf(fmaxf(pcz + 0.5f * pl, 0.0f), p.dz);
    } else {
        r[8] = r[9] = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) r[10 + j] = a[j];
    }
    if (p.records) {
#pragma unroll
        for (int j = 0; j < SIS3D_RECORD_WIDTH; ++j) p.records[(int64_t)k * SIS3D_RECORD_WIDTH + j] = r[j];
    }
    if (p.block) {
        const float ox = p.origin ? p.origin[0] : 0.0f, oy = p.origin ? p.origin[1] : 0.0f, oz = p.origin ? p.origin[2] : 0.0f;
        const float off[SIS3D_RECORD_WIDTH] = {ox, oy, oz, ox, oy, oz, 0.f, 0.f, 0.f, 0.f, ox, oy, oz, ox, oy, oz};
        float *b = p.block + 1 + (int64_t)k * SIS3D_RECORD_WIDTH;
#pragma unroll
        for (int j = 0; j < SIS3D_RECORD_WIDTH; ++j) b[j] = k < n ? r[j] + off[j] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void pack_records_kernel(const PackArgs p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = p.num[0];
    if (k == 0 && p.block) p.block[0] = (float)n;
    if (k < p.K) pack_row(p, k, n);
}

// the same + the LAST node of a mailbox pipeline (include/sis3d.h, sis3d_mail_post) in one launch: a single workgroup (K <= 256), so
// the finished block can be copied to the slot's destination row and the slot consumed behind one barrier
__global__ __launch_bounds__(256) void pack_records_post_kernel(const PackArgs p, unsigned *__restrict__ mail_state,
                                                                unsigned long long *__restrict__ mail_progress)
{
    const int k = threadIdx.x;
    const int n = p.num[0];
    if (k == 0) p.block[0] = (float)n;
    if (k < p.K) pack_row(p, k, n);
    __threadfence_block();
    __syncthreads();
    const unsigned long long dstp = *reinterpret_cast<const unsigned long long *>(mail_state + 8 + 2);       // MailSlot.dst (state[8..15])
    float *dst = reinterpret_cast<float *>(dstp);
    if (dst) {
        const int total = 1 + p.K * SIS3D_RECORD_WIDTH;
        for (int i = threadIdx.x; i < total; i += blockDim.x) dst[i] = p.block[i];
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned c = mail_state[0] + 1u;
        mail_state[0] = c;
        if (mail_progress) {
            // state[1]: the fetch rejected a slot (stale / lapped / torn: csrc/pool_misc.hip) -> progress[1], where the producer raises
            if (mail_state[1]) __hip_atomic_store(mail_progress + 1, (unsigned long long)mail_state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mail_progress, (unsigned long long)c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

} // namespace

extern "C" int sis3d_pack_records(const float *rois, const float *scores, const float *levels, const int64_t *cls_pred,
                                  const float *cls_prob, const float *bbox_pred, const int32_t *num, const float *origin, int K, int NC,
                                  float dim_x, float dim_y, float dim_z, float *records, float *block, sis3d_stream_t stream)
{
    if (K < 0) return SIS3D_EINVAL;
    if (!rois || !scores || !levels || !num || (!records && !block)) return SIS3D_EINVAL;
    if (cls_pred && (!cls_prob || !bbox_pred || NC <= 0)) return SIS3D_EINVAL;
    const PackArgs p = {rois, scores, levels, cls_pred, cls_prob, bbox_pred, num, origin, K, NC, dim_x, dim_y, dim_z, records, block};
    hipLaunchKernelGGL(pack_records_kernel, dim3(cdiv(K > 0 ? K : 1, 256)), dim3(256), 0, as_stream(stream), p);
    return sis3d_check_launch();
}

extern "C" int sis3d_pack_records_post(const float *rois, const float *scores, const float *levels, const int64_t *cls_pred,
                                       const float *cls_prob, const float *bbox_pred, const int32_t *num, const float *origin, int K, int NC,
                                       float dim_x, float dim_y, float dim_z, float *records, float *block, uint32_t *mail_state,
                                       uint64_t *mail_progress, sis3d_stream_t stream)
{
    if (K < 0 || K > 256) return K < 0 ? SIS3D_EINVAL : SIS3D_EUNSUPPORTED;        // one workgroup: the caller falls back to two launches
    if (!rois || !scores || !levels || !num || !block || !mail_state) return SIS3D_EINVAL;
    if (cls_pred && (!cls_prob || !bbox_pred || NC <= 0)) return SIS3D_EINVAL;
    const PackArgs p = {rois, scores, levels, cls_pred, cls_prob, bbox_pred, num, origin, K, NC, dim_x, dim_y, dim_z, records, block};
    hipLaunchKernelGGL(pack_records_post_kernel, dim3(1), dim3(256), 0, as_stream(stream), p, (unsigned *)mail_state,
                       (unsigned long long *)mail_progress);
    return sis3d_check_launch();
}

extern "C" int sis3d_proposal_decode2(const float *anchors1, const float *deltas1, const float *prob_fg1, const int32_t *inside1, int n1,
                                      float level1, const float *anchors2, const float *deltas2, const float *prob_fg2,
                                      const int32_t *inside2, int n2, float level2, float dim_x, float dim_y, float dim_z,
                                      float *out_boxes, float *out_scores, float *out_levels, sis3d_stream_t stream)
{
    if (n1 < 0 || n2 < 0) return SIS3D_EINVAL;
    if (n1 + n2 == 0) return SIS3D_OK;
    if ((n1 && (!anchors1 || !deltas1 || !prob_fg1 || !inside1)) || (n2 && (!anchors2 || !deltas2 || !prob_fg2 || !inside2))) return SIS3D_EINVAL;
    if (!out_boxes || !out_scores || !out_levels) return SIS3D_EINVAL;
    Decode2Args a;
    a.anchors[0] = anchors1; a.deltas[0] = deltas1; a.prob_fg[0] = prob_fg1; a.inside[0] = inside1; a.n[0] = n1; a.level[0] = level1;
    a.anchors[1] = anchors2; a.deltas[1] = deltas2; a.prob_fg[1] = prob_fg2; a.inside[1] = inside2; a.n[1] = n2; a.level[1] = level2;
    const int nb1 = cdiv(n1, 256), nb2 = cdiv(n2, 256);
    hipLaunchKernelGGL(decode2_kernel, dim3(nb1 + nb2), dim3(256), 0, as_stream(stream), a, nb1, dim_x, dim_y, dim_z, out_boxes, out_scores,
                       out_levels);
    return sis3d_check_launch();
}

extern "C" int sis3d_proposal_decode(const float *anchors, const float *deltas, const float *prob_fg, const int32_t *inside,
                                     int n_inside, float dim_x, float dim_y, float dim_z, float level_id, float *out_boxes,
                                     float *out_scores, float *out_levels, sis3d_stream_t stream)
{
    if (n_inside < 0) return SIS3D_EINVAL;
    if (n_inside == 0) return SIS3D_OK;
    if (!anchors || !deltas || !prob_fg || !inside || !out_boxes || !out_scores || !out_levels) return SIS3D_EINVAL;
    hipLaunchKernelGGL(decode_kernel, dim3(cdiv(n_inside, 256)), dim3(256), 0, as_stream(stream), anchors, deltas, prob_fg, inside,
                       n_inside, dim_x, dim_y, dim_z, level_id, out_boxes, out_scores, out_levels);
    return sis3d_check_launch();
}

extern "C" int sis3d_softmax2(const float *score, float *prob, int64_t n, sis3d_stream_t stream)
{
    if (n < 0) return SIS3D_EINVAL;
    if (n == 0) return SIS3D_OK;
    if (!score || !prob) return SIS3D_EINVAL;
    hipLaunchKernelGGL(softmax2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), score, prob, n);
    return sis3d_check_launch();
}
