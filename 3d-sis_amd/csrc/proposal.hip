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
__global__ __launch_bounds__(256) void pack_records_kernel(const float *__restrict__ rois, const float *__restrict__ scores,
                                                           const float *__restrict__ levels, const int64_t *__restrict__ cls_pred,
                                                           const float *__restrict__ cls_prob, const float *__restrict__ bbox_pred,
                                                           const int32_t *__restrict__ num, const float *__restrict__ origin, int K,
                                                           int NC, float dx_, float dy_, float dz_, float *__restrict__ records,
                                                           float *__restrict__ block)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = num[0];
    if (k == 0 && block) block[0] = (float)n;
    if (k >= K) return;
    const float *a = rois + 6 * (int64_t)k;
    float r[SIS3D_RECORD_WIDTH];
#pragma unroll
    for (int j = 0; j < 6; ++j) r[j] = a[j];
    r[6] = scores[k];
    r[7] = levels[k];
    if (cls_pred) {
        const int c = (int)cls_pred[k];
        r[8] = (float)c;
        r[9] = cls_prob[(int64_t)k * NC + c];
        const float *d = bbox_pred + (int64_t)k * (6 * NC) + 6 * c;
        const float w = a[3] - a[0], h = a[4] - a[1], l = a[5] - a[2];
        const float cx = a[0] + 0.5f * w, cy = a[1] + 0.5f * h, cz = a[2] + 0.5f * l;
        const float pcx = d[0] * w + cx, pcy = d[1] * h + cy, pcz = d[2] * l + cz;
        const float pw = expf(d[3]) * w, ph = expf(d[4]) * h, pl = expf(d[5]) * l;
        r[10] = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), dx_);
        r[11] = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), dy_);
        r[12] = fminf(fmaxf(pcz - 0.5f * pl, 0.0f), dz_);
        r[13] = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), dx_);
        r[14] = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), dy_);
        r[15] = fminf(fmaxf(pcz + 0.5f * pl, 0.0f), dz_);
    } else {
        r[8] = r[9] = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) r[10 + j] = a[j];
    }
    if (records) {
#pragma unroll
        for (int j = 0; j < SIS3D_RECORD_WIDTH; ++j) records[(int64_t)k * SIS3D_RECORD_WIDTH + j] = r[j];
    }
    if (block) {
        const float ox = origin ? origin[0] : 0.0f, oy = origin ? origin[1] : 0.0f, oz = origin ? origin[2] : 0.0f;
        const float off[SIS3D_RECORD_WIDTH] = {ox, oy, oz, ox, oy, oz, 0.f, 0.f, 0.f, 0.f, ox, oy, oz, ox, oy, oz};
        float *b = block + 1 + (int64_t)k * SIS3D_RECORD_WIDTH;
#pragma unroll
        for (int j = 0; j < SIS3D_RECORD_WIDTH; ++j) b[j] = k < n ? r[j] + off[j] : 0.0f;
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
    hipLaunchKernelGGL(pack_records_kernel, dim3(cdiv(K > 0 ? K : 1, 256)), dim3(256), 0, as_stream(stream), rois, scores, levels,
                       cls_pred, cls_prob, bbox_pred, num, origin, K, NC, dim_x, dim_y, dim_z, records, block);
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
