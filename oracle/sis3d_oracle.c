/* sis3d_oracle.c -- CPU restatement of the integer/index-exact parts of the
 * 3D-SIS forward path.  TEST INFRASTRUCTURE ONLY: linked/loaded solely by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (3d-sis_amd/) never calls into this file.
 *
 * Every function cites the reference file:line whose arithmetic it follows
 * (paths relative to the reference repo root).  All floating-point work is
 * done in binary32 with the same operation order as the reference so that the
 * integer outputs (keep lists, argmax indices, scatter results) are bit-exact.
 * Compiled with -ffp-contract=off (oracle/Makefile).
 *
 * Parity pinning: tests/test_oracle_pinning.py checks these functions against
 * (a) the committed golden vectors under tests/golden/ that were produced by
 * the reference itself (oracle/make_golden.py) and (b), in the build
 * container, against the reference run live (oracle/ref_harness.py +
 * oracle/_ref/libref_roi_pooling.so).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ NMS --
 * lib/layer_utils/nms/pth_nms.py:7-45 (cpu_nms, numpy on float32 arrays) and
 * lib/layer_utils/nms/src/cuda/nms_kernel.cu:11-31 (devIoU) +
 * lib/layer_utils/nms/src/nms_cuda.c:44-59 (greedy sweep).  Both reference
 * implementations compute, in binary32,
 *   area = (x2-x1+1)*(y2-y1+1)*(z2-z1+1)
 *   inter = max(0,min(x2)-max(x1)+1) * ... ; iou = inter/(area_i+area_j-inter)
 * and cpu_nms keeps j iff `ovr <= thresh` (pth_nms.py:42), i.e. suppresses iff
 * !(iou <= thresh) -- identical to the CUDA `iou > thresh` except for NaN,
 * where we follow the CPU path (the north star's parity target).
 * boxes: [n][6] score-sorted; keep: out, ascending indices; returns count. */
static float orc_iou(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), top = fmaxf(a[1], b[1]), front = fmaxf(a[2], b[2]);
    float right = fminf(a[3], b[3]), bottom = fminf(a[4], b[4]), back = fminf(a[5], b[5]);
    float w = fmaxf(right - left + 1.0f, 0.0f);
    float h = fmaxf(bottom - top + 1.0f, 0.0f);
    float l = fmaxf(back - front + 1.0f, 0.0f);
    float inter = w * h * l;
    float sa = (a[3] - a[0] + 1.0f) * (a[4] - a[1] + 1.0f) * (a[5] - a[2] + 1.0f);
    float sb = (b[3] - b[0] + 1.0f) * (b[4] - b[1] + 1.0f) * (b[5] - b[2] + 1.0f);
    return inter / (sa + sb - inter);
}

int orc_nms(const float *boxes, int n, float thresh, int64_t *keep)
{
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!dead[j] && !(orc_iou(boxes + 6 * i, boxes + 6 * j) <= thresh)) dead[j] = 1;
    }
    free(dead);
    return nk;
}

/* Bit-matrix form of the same thing (nms_kernel.cu:34-79 writes mask[i][cb]
 * bit j set iff iou(i, 64*cb+j) > thresh, only j>i on the diagonal block).
 * mask: [n][ceil(n/64)] u64.  Used to pin the device mask kernel itself. */
void orc_nms_mask(const float *boxes, int n, float thresh, uint64_t *mask)
{
    int cb = (n + 63) / 64;
    memset(mask, 0, sizeof(uint64_t) * (size_t)n * (size_t)cb);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (!(orc_iou(boxes + 6 * i, boxes + 6 * j) <= thresh))
                mask[(size_t)i * cb + j / 64] |= 1ULL << (j % 64);
}

/* --------------------------------------------------------- RoI pooling --
 * lib/layer_utils/roi_pooling/src/roi_pooling.c:6-124 (CPU values) and
 * lib/layer_utils/roi_pooling/src/cuda/roi_pooling_kernel.cu:15-109 (same
 * arithmetic + argmax = linear index (c*W+w)*H*L + h*L + l of the first
 * strict maximum in w->h->l scan order; empty bin -> value 0, argmax -1).
 * features: [C][W][H][L] (batch 1), rois: [R][6] scene coords,
 * out: [R][C][pw][ph][pl], argmax: same shape int32 or NULL. */
int orc_roi_pool(const float *feat, int C, int W, int H, int L,
                 const float *rois, int R, int pw_, int ph_, int pl_, float scale,
                 float *out, int32_t *argmax)
{
    for (int n = 0; n < R; ++n) {
        const float *r = rois + 6 * n;
        int rs_w = (int)floor(r[0] * scale), rs_h = (int)floor(r[1] * scale), rs_l = (int)floor(r[2] * scale);
        int re_w = (int)ceil(r[3] * scale), re_h = (int)ceil(r[4] * scale), re_l = (int)ceil(r[5] * scale);
        int rw = (int)fmaxf((float)(re_w - rs_w), 1.0f);
        int rh = (int)fmaxf((float)(re_h - rs_h), 1.0f);
        int rl = (int)fmaxf((float)(re_l - rs_l), 1.0f);
        float bw = (float)rw / (float)pw_, bh = (float)rh / (float)ph_, bl = (float)rl / (float)pl_;
        for (int c = 0; c < C; ++c)
            for (int pw = 0; pw < pw_; ++pw)
                for (int ph = 0; ph < ph_; ++ph)
                    for (int pl = 0; pl < pl_; ++pl) {
                        int ws = (int)floor((float)pw * bw), hs = (int)floor((float)ph * bh), ls = (int)floor((float)pl * bl);
                        int we = (int)ceil((float)(pw + 1) * bw), he = (int)ceil((float)(ph + 1) * bh), le = (int)ceil((float)(pl + 1) * bl);
                        ws = (int)fminf(fmaxf((float)(ws + rs_w), 0.0f), (float)W);
                        hs = (int)fminf(fmaxf((float)(hs + rs_h), 0.0f), (float)H);
                        ls = (int)fminf(fmaxf((float)(ls + rs_l), 0.0f), (float)L);
                        we = (int)fminf(fmaxf((float)(we + rs_w), 0.0f), (float)W);
                        he = (int)fminf(fmaxf((float)(he + rs_h), 0.0f), (float)H);
                        le = (int)fminf(fmaxf((float)(le + rs_l), 0.0f), (float)L);
                        int empty = (he <= hs) || (we <= ws) || (le <= ls);
                        float mx = empty ? 0.0f : -FLT_MAX;
                        int mi = -1;
                        for (int w = ws; w < we; ++w)
                            for (int h = hs; h < he; ++h)
                                for (int l = ls; l < le; ++l) {
                                    int bi = (c * W + w) * H * L + h * L + l;
                                    if (feat[bi] > mx) { mx = feat[bi]; mi = bi; }
                                }
                        size_t oi = ((((size_t)n * C + c) * pw_ + pw) * ph_ + ph) * pl_ + pl;
                        out[oi] = mx;
                        if (argmax) argmax[oi] = mi;
                    }
    }
    return 1;
}

/* ------------------------------------------------------ back-projection --
 * lib/layer_utils/projection.py:124-136 (Projection.forward):
 *   out = zeros(C, Z*Y*X); n = i3d[0];
 *   out[:, i3d[1..n]] = feat[:, i2d[1..n]]
 * feat: [C][npix]; out: [C][nvox] (nvox = X*Y*Z, linear index z*(X*Y)+y*X+x). */
void orc_projection(const float *feat, int C, int64_t npix, const int64_t *i3d, const int64_t *i2d,
                    int64_t nvox, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)C * (size_t)nvox);
    int64_t n = i3d[0];
    for (int c = 0; c < C; ++c)
        for (int64_t k = 1; k <= n; ++k)
            out[(size_t)c * nvox + i3d[k]] = feat[(size_t)c * npix + i2d[k]];
}

/* lib/nets/network.py:216-239 (TEST branch): views whose position is in
 * killing_inds are skipped; the first included view initialises the volume,
 * every further one is folded in with an elementwise max of the two
 * zero-filled volumes (stack -> view(C,-1,2) -> MaxPool1d(2)).
 * feats: [V][C][npix]; i3d,i2d: [V][nvox+1]; kill: [V] 0/1; out: [C][nvox]. */
void orc_project_views_max(const float *feats, int V, int C, int64_t npix, const int64_t *i3d,
                           const int64_t *i2d, const unsigned char *kill, int64_t nvox, float *out)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)C * (size_t)nvox);
    int init = 1;
    for (int v = 0; v < V; ++v) {
        if (kill && kill[v]) continue;
        const float *f = feats + (size_t)v * C * npix;
        const int64_t *a = i3d + (size_t)v * (nvox + 1), *b = i2d + (size_t)v * (nvox + 1);
        if (init) { orc_projection(f, C, npix, a, b, nvox, out); init = 0; continue; }
        orc_projection(f, C, npix, a, b, nvox, tmp);
        for (size_t e = 0; e < (size_t)C * (size_t)nvox; ++e)
            out[e] = tmp[e] > out[e] ? tmp[e] : out[e];
    }
    free(tmp);
}

/* ------------------------------------------------ inside-anchor filter --
 * lib/layer_utils/proposal_layer.py:36-43: keep anchors with all mins >= -border
 * and all maxes < dim + border.  Returns count; inds ascending. */
int64_t orc_inside_anchors(const float *anchors, int64_t n, const float *dims, float border, int64_t *inds)
{
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *a = anchors + 6 * i;
        if (a[0] >= -border && a[1] >= -border && a[2] >= -border &&
            a[3] < dims[0] + border && a[4] < dims[1] + border && a[5] < dims[2] + border)
            inds[m++] = i;
    }
    return m;
}
