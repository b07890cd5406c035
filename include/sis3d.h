/* sis3d.h -- C ABI of libsis3d_hip.so: the MI355X (gfx950) forward path of 3D-SIS.
 *
 * Drop-in boundary.  Conventions, mirroring the
 * reference's cffi layer ("caller allocates, int status") but with the status
 * actually meaningful:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - nothing here allocates, frees or synchronises; all work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: SIS3D_OK (0) or a negative SIS3D_E* code; launch failures
 *     are reported through the return value, never by exit() (the reference
 *     calls exit(-1): roi_pooling_kernel.cu:127-132);
 *   - tensors are fp32; index outputs are int32/int64 as stated;
 *   - voxel grids use the reference's axis naming (X=W, Y=H, Z=L), Z fastest in
 *     the reference's NCDHW tensors.  Entry points that take explicit element
 *     strides work on either memory order (NCDHW or channels-last X,Y,Z,C).
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference repo root).
 */
#ifndef SIS3D_H
#define SIS3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIS3D_OK 0
#define SIS3D_EINVAL (-1)   /* bad shape / argument (reference returned 0 and was ignored) */
#define SIS3D_ELAUNCH (-2)  /* hipGetLastError() != success after a launch */
#define SIS3D_EWORKSPACE (-3) /* workspace too small */
#define SIS3D_EUNSUPPORTED (-4)

typedef void *sis3d_stream_t; /* hipStream_t */

int sis3d_abi_version(void);
const char *sis3d_strerror(int code);
/* last HIP error string recorded by a failing launch on this thread */
const char *sis3d_last_hip_error(void);

/* ---------------------------------------------------------------- 3D NMS --
 * Replaces: int gpu_nms(THLongTensor* keep, THLongTensor* num_out, THCudaTensor* boxes,
 *           float thresh)          lib/layer_utils/nms/src/nms_cuda.h, nms_cuda.c:10-67
 *           + nms_kernel / _nms    lib/layer_utils/nms/src/cuda/nms_kernel.cu:34-94
 * boxes [n][6] score-sorted (x1,y1,z1,x2,y2,z2), IoU with +1 extents, suppress
 * iff !(iou <= thresh) (== cpu_nms, pth_nms.py:42).  keep [n] ascending kept
 * indices (entries >= *num_keep untouched), num_keep[0] = count.  The greedy
 * sweep runs ON THE DEVICE (the reference copies the mask to the host).
 * max_keep > 0 stops after that many survivors (== keep[:max_keep]).
 * workspace: sis3d_nms_workspace_bytes(n) bytes (bit matrix + the bitmap of its non-zero
 * words), any content.  n whose matrix fits the sweep workgroup's LDS (n <~ 990) take the
 * one-workgroup sweep; larger n (a whole scene's records) take a sparse suppressor table +
 * parallel fixed-point resolve -- same keep list. */
size_t sis3d_nms_workspace_bytes(int n);
/* path: which of the two algorithms this call uses: 0 = by size (what every product call passes), 1 = one-workgroup sweep,
 * 2 = parallel resolve (parity hook: the tests force both at every size; both give the same keep list).  A per-call argument
 * (r5; it was a process-wide switch before): nothing about the dispatch lives in library state, so calls on distinct streams
 * from distinct threads do not interact. */
int sis3d_nms(const float *boxes, int n, float thresh, int max_keep, int64_t *keep, int32_t *num_keep,
              void *workspace, size_t workspace_bytes, int path, sis3d_stream_t stream);
/* the IoU bit matrix alone: mask [n][ceil(n/64)] u64, bit j of word cb set iff
 * box 64*cb+j (> i) is suppressed by box i (nms_kernel.cu:34-79). */
int sis3d_nms_mask(const float *boxes, int n, float thresh, uint64_t *mask, sis3d_stream_t stream);

/* Fused proposal selection (proposal_layer.py:181-197): boxes_all [m][6],
 * scores_sorted [>=n] (descending), order [>=n] int64 indices into boxes_all,
 * level_all [m] (1.0/2.0/..).  Takes the first n entries of `order`, runs NMS
 * and writes the first min(count,max_keep) survivors to rois [max_keep][6],
 * roi_scores [max_keep], roi_levels [max_keep] (float), keep [n] positions in the
 * sorted list, num_keep[0].  Rows >= count are zero-filled (level 0). */
int sis3d_nms_select(const float *boxes_all, const float *level_all, const float *scores_sorted, const int64_t *order,
                     int n, float thresh, int max_keep, float *rois, float *roi_scores, float *roi_levels,
                     int64_t *keep, int32_t *num_keep, void *workspace, size_t workspace_bytes, sis3d_stream_t stream);

/* --------------------------------------------------------- 3D RoI pooling --
 * Replaces: int roi_pooling_forward_cuda(int pw,int ph,int pl,float scale, THCudaTensor* features,
 *           THCudaTensor* rois, THCudaTensor* output, THCudaIntTensor* argmax)
 *           lib/layer_utils/roi_pooling/src/roi_pooling_cuda.h, roi_pooling_cuda.c:7-52,
 *           ROIPoolForward  lib/layer_utils/roi_pooling/src/cuda/roi_pooling_kernel.cu:15-134
 * features: one scene, C channels on a W x H x L grid, element strides
 * (fs_c, fs_w, fs_h, fs_l).  rois [R][6] scene coordinates.  out element
 * (n,c,pw,ph,pl) is written at n*os_n + c*os_c + ((pw*PH+ph)*PL+pl)*os_bin.
 * argmax (may be NULL) uses the same addressing and holds the NCDHW linear
 * index (c*W+w)*H*L + h*L + l of the first strict maximum in w->h->l scan
 * order, -1 for an empty bin (value 0) -- independent of the memory order. */
int sis3d_roi_pool_forward(const float *features, int C, int W, int H, int L, int64_t fs_c, int64_t fs_w, int64_t fs_h,
                           int64_t fs_l, const float *rois, int R, int pooled_w, int pooled_h, int pooled_l,
                           float spatial_scale, float *out, int32_t *argmax, int64_t os_n, int64_t os_c, int64_t os_bin,
                           sis3d_stream_t stream);
/* Both pyramid levels + the row scatter of Network._roi_pool_layer
 * (lib/nets/network.py:503-534) in one launch: roi n pools from features1 if
 * levels[n]==1, features2 if ==2, otherwise its output rows are zero. */
int sis3d_roi_pool_levels(const float *features1, const float *features2, int C, int W, int H, int L, int64_t fs_c,
                          int64_t fs_w, int64_t fs_h, int64_t fs_l, const float *rois, const float *levels, int R,
                          int pooled, float spatial_scale, float *out, int64_t os_n, int64_t os_c, int64_t os_bin,
                          sis3d_stream_t stream);

/* backward (training).  Replaces int roi_pooling_backward_cuda(int,int,int,float, THCudaTensor* top_grad,
 * THCudaTensor* rois, THCudaTensor* bottom_grad, THCudaIntTensor* argmax)  (roi_pooling_cuda.h, ROIPoolBackward
 * roi_pooling_kernel.cu:137-248): grad_in[c,w,h,l] += grad_out[n,c,bin] for every (n,c,bin) whose argmax is (c,w,h,l).
 * grad_out / argmax addressed as in sis3d_roi_pool_forward; grad_in: CALLER-ZEROED map with element strides gs_*.
 * r6: deterministic -- every element's sum is built in the reference's order (RoI ascending, then pw / ph / pl ascending), no
 * atomics: bit-identical to ROIPoolBackward / the Python RoIPool.backward on the same inputs.  C <= 16384. */
int sis3d_roi_pool_backward(const float *grad_out, const int32_t *argmax, int R, int C, int pw, int ph, int pl, int64_t os_n,
                            int64_t os_c, int64_t os_bin, int W, int H, int L, float *grad_in, int64_t gs_c, int64_t gs_w,
                            int64_t gs_h, int64_t gs_l, sis3d_stream_t stream);

/* ------------------------------------------------- 2D -> 3D back-projection --
 * Replaces: Projection.forward   lib/layer_utils/projection.py:124-136
 * feat [C][npix]; lin3d/lin2d int64 [nvox+1], slot 0 = count n (read on the
 * device).  out [C][nvox] = 0 except out[c][lin3d[1+k]] = feat[c][lin2d[1+k]]. */
int sis3d_projection_forward(const float *feat, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                             int64_t nvox, float *out, sis3d_stream_t stream);
/* Replaces the per-view loop + pairwise MaxPool1d of lib/nets/network.py:216-239:
 * feats [V][C][npix], lin3d/lin2d [V][nvox+1], kill_host [V] (1 = view skipped,
 * may be NULL).  Per voxel and channel: max over the included views of
 * (visible ? feature : 0); with one included view a plain copy.  Volume dims
 * X,Y,Z (nvox = X*Y*Z, linear voxel index z*X*Y + y*X + x).  out element
 * (c,x,y,z) at c*os_c + x*os_x + y*os_y + z*os_z: the reference's memory order
 * is (os_c,os_x,os_y,os_z) = (nvox,1,X,X*Y); channels-last is (1,Y*Z*C,Z*C,C).
 * workspace: sis3d_project_views_workspace_bytes(V,C,npix,nvox). */
size_t sis3d_project_views_workspace_bytes(int V, int C, int64_t npix, int64_t nvox);
int sis3d_project_views_max(const float *feats, int V, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                            const uint8_t *kill_host, int X, int Y, int Z, float *out, int64_t os_c, int64_t os_x,
                            int64_t os_y, int64_t os_z, void *workspace, size_t workspace_bytes, sis3d_stream_t stream);

/* The same view max WITHOUT materialising the volume (226 MB at 128 ch x 96x48x96), for the colour
 * stem: sis3d_project_views_prepare builds vox2pix [nslots][nvox] int32 (-1 = invisible; nslots =
 * views not killed, returned through *nslots_out) and feat_rows [nslots][npix][C] (pixel-major);
 * sis3d_conv3d_chain_projected (below) is Conv3d(C, cout, k=2, s=2) + fused stages reading its
 * input voxels through that table: value(voxel, c) = max over slots of (visible ? row[c] : 0), and
 * workgroups whose input brick holds no visible voxel skip the reduction (lib/nets/network.py:216-239
 * + backbones.py color[0]).  Caller allocates vox2pix (V*nvox) and feat_rows (V*npix*C). */
int sis3d_project_views_prepare(const float *feats, int V, int C, int64_t npix, const int64_t *lin3d, const int64_t *lin2d,
                                const uint8_t *kill_host, int64_t nvox, int32_t *vox2pix, float *feat_rows, int *nslots_out,
                                sis3d_stream_t stream);

/* Projection.backward (lib/layer_utils/projection.py:139-153, training only): grad_label [C][npix] = the first C*npix elements
 * of grad_out's storage (the reference resizes a CLONE of grad_output: pixels no voxel maps to keep those values), then
 * grad_label[:, lin2d[1+k]] = grad_out[:, lin3d[1+k]] for k < lin3d[0], the last k winning on shared pixels.
 * grad_out [C][nvox].  workspace: sis3d_projection_backward_workspace_bytes(npix). */
size_t sis3d_projection_backward_workspace_bytes(int64_t npix);
int sis3d_projection_backward(const float *grad_out, int C, int64_t nvox, const int64_t *lin3d, const int64_t *lin2d, int64_t npix,
                              float *grad_label, void *workspace, size_t workspace_bytes, sis3d_stream_t stream);

/* Replaces ProjectionHelper.compute_projection (lib/layer_utils/projection.py:52-121) and its
 * call sites' per-view loop (lib/model/trainval.py:336-337,464-465,663-667,799-803) for V views of
 * one volume.  depth [V][W*H]; view_params [V][SIS3D_VIEW_PARAM_FLOATS] = grid_to_world (16,
 * row-major), world_to_camera (16), frustum voxel bounds min (3) and max (3) already clamped to
 * [0, dims] (projection.py:59-61), 2 pad -- both device pointers.  fx..cy = INTRINSIC[0][0],
 * [1][1], [0][2], [1][2]; W,H = DEPTH_SHAPE.  Outputs lin3d/lin2d [V][nvox+1] int64 in the
 * reference's packing: slot 0 = count n, slots 1..n = ascending linear voxel index
 * (z*X*Y + y*X + x) / pixel index (v*W + u); slots > n are zeroed (the reference leaves them
 * uninitialised).  n == 0 is the reference's `return None`.  No host synchronisation.
 * workspace: sis3d_compute_projection_workspace_bytes(V, nvox). */
#define SIS3D_VIEW_PARAM_FLOATS 40
size_t sis3d_compute_projection_workspace_bytes(int V, int64_t nvox);
int sis3d_compute_projection(const float *depth, const float *view_params, int V, int X, int Y, int Z, int W, int H,
                             float fx, float fy, float cx, float cy, float depth_min, float depth_max, float voxel_size,
                             int64_t *lin3d, int64_t *lin2d, void *workspace, size_t workspace_bytes,
                             sis3d_stream_t stream);

/* Replaces the TSDF encoding of Dataset.__getitem__ (lib/datasets/dataset.py:54-70) plus the
 * max-height crop (:196-211, data[:, :, :maxHeight, :]) and the upload: sdf is the raw f32 grid of
 * a .chunk/.scene file (x fastest, then y, then z; writer datagen/SceneSampler/main.cpp:348-415),
 * already on the device.  out element (c,x,y,z), c in {0,1}, y < Yout <= Y, at
 * c*os_c + x*os_x + y*os_y + z*os_z.  c0 = |clamp(v,-T,T)| (mode 0), T - that (mode 1, FLIP_TSDF),
 * log of it (mode 2, LOG_TSDF); c1 = v > -1 ? 1 : 0. */
int sis3d_tsdf_encode(const float *sdf, int X, int Y, int Z, int Yout, float truncated, int mode, float *out,
                      int64_t os_c, int64_t os_x, int64_t os_y, int64_t os_z, sis3d_stream_t stream);

/* ---- upload of a chunk by a KERNEL that reads host memory (r5) ---------------------------------------------------------------
 * Replaces the `blobs['data'].cuda()` at the head of the reference's forward (lib/nets/network.py:191) for chunk pipelines.
 * src: n floats in PINNED, device-mapped host memory (hipHostMalloc / torch pin_memory: the pointer is valid on the device);
 * dst: n floats of device memory; both 16-byte aligned, n % 4 == 0.  A grid-stride 16-byte copy on `stream`: the launch is an
 * ordinary kernel launch, so -- unlike hipMemcpyAsync, which was measured to BLOCK THE HOST when it is enqueued behind a captured
 * graph that has not drained -- it never stalls the enqueueing thread, needs no copy stream and no events, and the reads cross PCIe
 * at the link rate while the other pipelines' kernels keep the CUs busy.  sis3d_tsdf_encode accepts such a host pointer as `sdf`
 * too (upload and TSDF encoding in one pass over the 1.77 MB block, but with one short-lived wave per 32 x 32 tile on every CU: chunk pipelines upload with this
 * function and encode from the device copy).  workgroups: 0 = default (8: few workgroups with 8 x 16 B in flight per lane -- a
 * wave waiting on the link must not sit on many CUs, see csrc/pool_misc.hip). */
int sis3d_upload_f32(const float *src_host_mapped, float *dst, int64_t n, int workgroups, sis3d_stream_t stream);

/* ---- host -> graph MAILBOX (r5): per-chunk inputs of a captured pipeline without a single per-chunk command ----------------------
 * On this runtime a command enqueued on a stream whose last command is a graph launch that has not finished can BLOCK THE HOST
 * until that graph drains -- an upload kernel, a hipMemcpyAsync, even the 12-byte copy of a chunk origin (profiles/r05_feed_probe.txt:
 * 0.16 -> 1.45 ms of host time per step, bimodal).  A pipeline that replays a captured graph per chunk therefore takes everything
 * that changes from chunk to chunk through a ring of 64-byte slots in PINNED host memory, which kernels INSIDE the graph read:
 *     slot = { u64 src; u64 dst; f32 origin[3]; u32 flags; u64 next_src; u64 seq; u64 check; u64 pad }
 *     seq = number of slots written before this one; check = src ^ rotl(dst, 17) ^ rotl(next_src, 31) ^ (flags << 40)
 *           ^ (seq * 0x9E3779B97F4A7C15) ^ 0x5151D3D3 (r6): sis3d_mail_fetch accepts a slot only if seq == state[0] (a stale or lapped
 *           slot carries another number) and the check word matches (a torn slot); a rejected slot is parked with its pointers
 *           zeroed -- the pass runs on whatever the input buffer holds and copies nothing out -- and the reason (1 stale / lapped,
 *           2 torn) goes to state[1] (sticky) and, with the next post, to progress[1]
 *     flags bit 0: origin valid, bit 1: src is DEVICE memory, bit 2: src was the next_src of the PREVIOUS slot and has not changed since;
 *     next_src: the chunk of the pipeline's NEXT pass (0 = unknown) -- the extra row of workgroups of sis3d_conv3d_k3wino_piggyback
 *     pulls it into a staging buffer while this pass computes, and records whose chunk the buffer holds in state[24..25]
 * The host writes slot (k mod ring_size) with plain CPU stores and replays the graph: hipGraphLaunch is the ONLY call per chunk.
 *   sis3d_mail_fetch   first node of the graph: ONE read of slot k = state[0] (device counter of consumed slots) across PCIe into
 *                      state[8..23] -- small uncached reads over the link are slow and serialise, so no other kernel touches the ring.
 *   sis3d_mail_upload  second node: copies n floats from the fetched slot's src -- a pinned host pointer (pulled across PCIe by 8
 *                      workgroups with 256 KB in flight, as sis3d_upload_f32) or a device pointer (flags bit 1: copied by the whole
 *                      grid at HBM speed); 0 = the input buffer already holds the chunk -- to `input_dst` (n % 4 == 0, 16-byte aligned),
 *                      and the slot's origin to `origin_dst` (3 floats; may be NULL).  `staged` (may be NULL): the staging buffer of
 *                      sis3d_conv3d_k3wino_piggyback -- taken instead of src when flags bit 2 is set and state[24..25] == src, i.e. the
 *                      previous pass already pulled this chunk across the link.  Replaces `blobs['data'].cuda()`
 *                      (lib/nets/network.py:191).
 *   sis3d_mail_post    last node: copies n floats of `block_src` (the chunk's record block) to the slot's dst (0 = nowhere) and
 *                      consumes the slot: state[0] = k + 1, progress[0] = k + 1 (pinned host word the producer polls before it laps
 *                      the ring), progress[1] = state[1] when a slot was rejected.
 * state: 32 uint32 of device memory, zero-initialised, owned by the pipeline.  progress: TWO uint64 of pinned host memory. */
int sis3d_mail_fetch(const void *ring_host_mapped, int ring_size, uint32_t *state, sis3d_stream_t stream);
int sis3d_mail_upload(const uint32_t *state, float *input_dst, int64_t n, float *origin_dst, const float *staged, int workgroups,
                      sis3d_stream_t stream);
int sis3d_mail_post(uint32_t *state, const float *block_src, int64_t n, uint64_t *progress_host_mapped, sis3d_stream_t stream);

/* ------------------------------------------------------- proposal decoding --
 * Replaces proposal_layer.py:96-103 + bbox_transform_inv / clip_boxes
 * (lib/utils/bbox_transform.py:59-99,4-21) for one level:
 * for k < n_inside:  i = inside[k];  box = clip(decode(anchors[i], deltas[i]))
 *   out_boxes[k] = box; out_scores[k] = prob_fg[i]; out_levels[k] = level_id.
 * anchors [K*A][6], deltas = rpn_bbox_pred viewed (-1,6), prob_fg =
 * rpn_cls_prob[0,1] viewed (-1). */
int sis3d_proposal_decode(const float *anchors, const float *deltas, const float *prob_fg, const int32_t *inside,
                          int n_inside, float dim_x, float dim_y, float dim_z, float level_id, float *out_boxes,
                          float *out_scores, float *out_levels, sis3d_stream_t stream);
/* both pyramid levels in ONE launch (r5): level 1 -> rows [0, n1), level 2 -> rows [n1, n1 + n2) of the outputs: exactly what two calls
 * of sis3d_proposal_decode into the two slices write. */
int sis3d_proposal_decode2(const float *anchors1, const float *deltas1, const float *prob_fg1, const int32_t *inside1, int n1,
                           float level1, const float *anchors2, const float *deltas2, const float *prob_fg2, const int32_t *inside2,
                           int n2, float level2, float dim_x, float dim_y, float dim_z, float *out_boxes, float *out_scores,
                           float *out_levels, sis3d_stream_t stream);
/* Replaces `scores.sort(descending=True)` + `[:pre_nms_topN]` (proposal_layer.py:181-186): the k (<= 1024) largest
 * of scores [n] in descending order, ties by ascending index (== torch.sort(stable=True, descending=True)[:k]).
 * out_scores [k], out_idx [k] int64.  One launch (radix select + bitonic sort in LDS); k > n is clamped to n;
 * n > 40960 returns SIS3D_EUNSUPPORTED (the score vector is held in registers). */
int sis3d_topk_desc(const float *scores, int n, int k, float *out_scores, int64_t *out_idx, sis3d_stream_t stream);
/* One detection record per padded RoI row, SIS3D_RECORD_WIDTH floats: [0:6] proposal box, [6] RPN score, [7] pyramid
 * level, [8] arg-max class, [9] its probability, [10:16] the class-specific regressed box clipped to the chunk --
 * box_reg row of the predicted class -> bbox_transform_inv -> clip_boxes, lib/model/trainval.py:686-700 and
 * lib/nets/network.py:285-294 (there on the host in numpy).  rois (K,6), scores/levels (K), cls_pred (K) int64,
 * cls_prob (K,NC), bbox_pred (K,6NC) (the three may be NULL together: columns 8,9 = 0, 10:16 = proposal box), num int32[1],
 * origin float[3] (may be NULL).  records (K,W) in chunk coordinates (may be NULL); block [1 + K*W] = count, then the rows
 * shifted by `origin` to scene coordinates with rows >= count zeroed: the unit of the per-scene all-gather (may be NULL). */
#define SIS3D_RECORD_WIDTH 16
int sis3d_pack_records(const float *rois, const float *scores, const float *levels, const int64_t *cls_pred, const float *cls_prob,
                       const float *bbox_pred, const int32_t *num, const float *origin, int K, int NC, float dim_x, float dim_y,
                       float dim_z, float *records, float *block, sis3d_stream_t stream);
/* the same + sis3d_mail_post in one launch (r5; K <= 256, `block` required): the last node of a mailbox pipeline's detection graph --
 * the finished block goes to the fetched slot's destination row and the slot is consumed.  SIS3D_EUNSUPPORTED for K > 256 (the caller
 * launches sis3d_pack_records and sis3d_mail_post). */
int sis3d_pack_records_post(const float *rois, const float *scores, const float *levels, const int64_t *cls_pred, const float *cls_prob,
                            const float *bbox_pred, const int32_t *num, const float *origin, int K, int NC, float dim_x, float dim_y,
                            float dim_z, float *records, float *block, uint32_t *mail_state, uint64_t *mail_progress_host_mapped,
                            sis3d_stream_t stream);
/* softmax over dim 1 of (1,2,...) score maps (network.py:546): n = elements per class plane */
int sis3d_softmax2(const float *score, float *prob, int64_t n, sis3d_stream_t stream);

/* ------------------------------------------------------------ RoI classifier --
 * Replaces the five cuBLAS GEMMs + ReLU / softmax / max kernels of Base_Backbone._classifier and
 * Network._region_classification (lib/nets/backbones.py:92-96,225-231, lib/nets/network.py:589-604):
 *   fc7 = relu(L3(relu(L2(relu(L1(x))))));  cls_score = Lc(fc7);  bbox_pred = Lb(fc7);
 *   cls_prob = softmax(cls_score, 1);  cls_pred = argmax(cls_score, 1) (first maximum).
 * x [R][K] fp32 rows (row stride ldx); w?p = sis3d_conv_pack_weight(weight viewed (Cout,Cin,1,1,1), ksize 1);
 * whp packs the two heads stacked: rows [0,NC) = classifier_cls_score_net, [NC,7NC) = classifier_bbox_pred_net,
 * bh likewise.  K % 128 == 0, C1,C2,C3 % 32 == 0.  Two launches (split-K first layer, fused tail). */
size_t sis3d_classifier_workspace_floats(int R, int K, int C1);
int sis3d_classifier_forward(const float *x, int R, int K, int ldx, const float *w1p, const float *b1, int C1,
                             const float *w2p, const float *b2, int C2, const float *w3p, const float *b3, int C3,
                             const float *whp, const float *bh, int NC, float *cls_score, float *cls_prob,
                             int64_t *cls_pred, float *bbox_pred, float *workspace, size_t workspace_floats,
                             sis3d_stream_t stream);

/* the same with a DEVICE-side count of live rows (the padded-row detection pass: R = RPN_POST_NMS_TOP_N rows are
 * allocated, nrows_dev[0] <= R of them are proposals): 32-row tiles entirely past the count are not computed, their
 * outputs are zero-filled.  nrows_dev may be NULL (= all R rows live). */
int sis3d_classifier_forward_n(const float *x, int R, const int32_t *nrows_dev, int K, int ldx, const float *w1p, const float *b1,
                               int C1, const float *w2p, const float *b2, int C2, const float *w3p, const float *b3, int C3,
                               const float *whp, const float *bh, int NC, float *cls_score, float *cls_prob, int64_t *cls_pred,
                               float *bbox_pred, float *workspace, size_t workspace_floats, sis3d_stream_t stream);

/* The same classifier in its latency form (csrc/mlp16.hip: 16x16x4 tiles, weights of the first layer held in registers by
 * (K/256) x (C1/32) workgroups, a 16-row tail workgroup per tile).  Weight packs: sis3d_conv_pw16_pack_weight of the
 * (Cout, Cin) matrices (heads stacked as for sis3d_classifier_forward; rows beyond 7*NC zero).  K % 256 == 0;
 * (C1, C2, C3) = (256, 256, 128) (lib/nets/backbones.py:225-231), 7*NC <= 256.  workspace:
 * sis3d_classifier16_workspace_floats(R, K, C1) floats. */
size_t sis3d_classifier16_workspace_floats(int R, int K, int C1);
int sis3d_classifier16_forward(const float *x, int R, const int32_t *nrows_dev, int K, int ldx, const float *w1p, const float *b1,
                               int C1, const float *w2p, const float *b2, int C2, const float *w3p, const float *b3, int C3,
                               const float *whp, const float *bh, int NC, float *cls_score, float *cls_prob, int64_t *cls_pred,
                               float *bbox_pred, float *workspace, size_t workspace_floats, sis3d_stream_t stream);

/* ------------------------------------------------------------ 3D convolution --
 * Replaces the cuDNN calls behind nn.Conv3d / nn.MaxPool3d / nn.ReLU / residual
 * add in lib/nets/backbones.py:17-40,171-287 and lib/nets/network.py:38-47.
 * Activations are channels-last: element (x,y,z,c) at ((x*Y+y)*Z+z)*C + c
 * (torch.channels_last_3d view of a logical (1,C,X,Y,Z) tensor).
 *
 * Weights are repacked once (sis3d_conv_pack_weight) from the checkpoint layout
 * (Cout,Cin,kx,ky,kz) into MFMA fragment order; Cin is padded to a multiple of
 * 8 and Cout to a multiple of 32 with zeros.  Returns floats needed by
 * sis3d_conv_packed_floats. */
size_t sis3d_conv_packed_floats(int cout, int cin, int ksize);
int sis3d_conv_pack_weight(const float *w, int cout, int cin, int ksize, float *packed, sis3d_stream_t stream);

#define SIS3D_EPI_RELU 1      /* max(v,0) last */
#define SIS3D_EPI_RESIDUAL 2  /* v += residual[(vox)*res_stride + c] before ReLU */
#define SIS3D_EPI_SIGMOID 4   /* 1/(1+exp(-v)) last (MaskBackbone eval, backbones.py:286) */
#define SIS3D_EPI_RPN_HEAD 8  /* channels [0,2A) -> score (2,X,Y,Z,A); [2A,8A) -> bbox (X,Y,Z,6A) */

/* out[(x,y,z)][co] = epi( sum_{tap,ci} in[(S*x+dx-P, ...)][ci] * w[co][ci][tap] + bias[co] )
 * ksize/stride/pad in {(1,1,0),(3,1,1),(2,2,0)}.  in: (X,Y,Z,cin_stride>=cin) channels-last,
 * reading channels [0,cin).  out: (OX,OY,OZ) voxels, row stride out_stride floats, channel offset
 * out_coff (lets two convs write one concatenated tensor, backbones.py:109).
 * bias may be NULL.  For SIS3D_EPI_RPN_HEAD: out = score base, out2 = bbox base, out3 = softmax(score) over the two
 * class planes (F.softmax of lib/nets/network.py:546; may be NULL), anchors = A (2A <= 32). */
int sis3d_conv3d(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                 int cout, int ksize, int stride, int flags, const float *residual, int res_stride, float *out,
                 int out_stride, int out_coff, float *out2, float *out3, int anchors, sis3d_stream_t stream);

/* A k=3 (or k=2/s=2) convolution followed by up to two fused 1x1x1 convolutions applied to the output tile
 * while it is still on chip: the Bottleneck of lib/nets/backbones.py:27-40 as ONE launch --
 *     main  = relu(conv2(y1) + b2)                       (k3, `flags` = SIS3D_EPI_RELU; `out` may be NULL)
 *     stage0= relu(conv3(main) + b3 + x)                 (1x1x1 + residual; Bottleneck output)
 *     stage1= relu(conv1_next(stage0) + b1)              (optional: the NEXT block's conv1)
 * or a stem conv (k2 s2 / k3) + the following block's conv1.  Each stage: cin = previous cout, cout in
 * {32,64,96,128}; packed_w from sis3d_conv_pack_weight(ksize 1); out (may be NULL except for the last stage)
 * is channels-last with row stride out_stride.  Returns SIS3D_EUNSUPPORTED if no tiling keeps all main-conv
 * output channels in one workgroup (use separate launches then). */
typedef struct sis3d_pw_stage {
    const float *packed_w, *bias, *residual;
    float *out;
    int cin, cout, res_stride, out_stride, flags;
} sis3d_pw_stage;
int sis3d_conv3d_chain(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                       int cout, int ksize, int stride, int flags, float *out, int out_stride, int nstages,
                       const sis3d_pw_stage *stages_host, sis3d_stream_t stream);
/* sis3d_conv3d_chain for Conv3d(cin, cout, k=2, s=2) whose input is a back-projected image volume given as
 * (vox2pix, feat_rows) from sis3d_project_views_prepare instead of a tensor; X,Y,Z = volume dims, cin % 32 == 0. */
int sis3d_conv3d_chain_projected(const int32_t *vox2pix, const float *feat_rows, int nslots, int64_t npix, int X, int Y, int Z,
                                 int cin, const float *packed_w, const float *bias, int cout, int flags, float *out,
                                 int out_stride, int nstages, const sis3d_pw_stage *stages_host, sis3d_stream_t stream);

/* sis3d_conv3d_chain_projected for the colour stem proper -- color[0] = Conv3d(128, 64, k=2, s=2, bias) + ReLU (lib/nets/backbones.py:187,214)
 * on the view max of lib/nets/network.py:216-239, followed (c2 = 32) by the next Bottleneck's conv1 (1x1x1, 64 -> 32, + b1, ReLU) -- computed
 * SPARSELY (csrc/proj_sparse.hip): output voxels none of whose eight input voxels is seen by a view get the constant row of a zero input
 * (relu(bias); relu(W1 relu(bias) + b1)), the others are compacted (ascending order, no atomics) and computed 64 per workgroup.  The launch
 * sequence does not depend on the visibility (worst-case grid, workgroups past the list exit), so it captures into a HIP graph.
 * w_pw16: sis3d_conv_pw16_pack_weight of the (64, 8 * 128) matrix with column = tap * 128 + ci, tap = (dx * 2 + dy) * 2 + dz; w1_pw16: the
 * same pack of (32, 64).  out / y1: rows of 64 / 32 floats per output voxel (x-major, z fastest).  X, Y, Z: the INPUT grid (even).
 * Other channel counts -> SIS3D_EUNSUPPORTED (the dense kernel serves them). */
size_t sis3d_conv3d_k2s2_projected_sparse_workspace_bytes(int X, int Y, int Z);
int sis3d_conv3d_k2s2_projected_sparse(const int32_t *vox2pix, const float *feat_rows, int nslots, int64_t npix, int X, int Y, int Z, int cin,
                                       const float *w_pw16, const float *bias, int cout, int relu, float *out, const float *w1_pw16,
                                       const float *b1, int c2, float *y1, void *workspace, size_t workspace_bytes, sis3d_stream_t stream);

/* The pointwise half of the Bottleneck (lib/nets/backbones.py:33-40) as ONE launch when conv2 ran on its own
 * (sis3d_conv3d_k3t16):   main  = relu(conv3(y2) + b3 + x)      1x1x1 + residual; the block output, written to `out` at
 *                                                               channel offset out_coff of rows of out_stride floats
 *                         stage0= relu(conv1_next(main) + b1)   optional (nstages 0 or 1): the NEXT block's conv1
 * Same tile-on-chip mechanism and argument meaning as sis3d_conv3d_chain; cin in {32,64}, packed_w from
 * sis3d_conv_pack_weight(ksize 1). */
int sis3d_conv3d_pw_chain(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                          int cout, int flags, const float *residual, int res_stride, float *out, int out_stride, int out_coff,
                          int nstages, const sis3d_pw_stage *stages_host, sis3d_stream_t stream);

/* The same pointwise pair as register-chained 16x16x4 MFMA GEMMs (csrc/pointwise.hip): a wave owns 16 voxels, the
 * activation rows are read 16 B per lane straight from global memory, conv3's result tile is already in the operand
 * layout of conv1_next -- no LDS staging, no barrier (wide layers: 4 waves share a voxel tile, one LDS reduction).
 *     out  = epi(W1 in + b1 [+ residual])            cout channels at out_coff of rows of out_stride floats (out may be
 *                                                    NULL when only the second stage is wanted)
 *     out2 = epi2(W2 out + b2)                       optional (cout2 > 0)
 * in: nvox rows of cin_stride floats (channels-last activations of ANY grid, flattened).  packed_w / packed_w2 from
 * sis3d_conv_pw16_pack_weight (checkpoint layout (Cout,Cin[,1,1,1]) -> [cout/16][cin/16][64][4]).  flags: SIS3D_EPI_RELU |
 * SIS3D_EPI_RESIDUAL | SIS3D_EPI_SIGMOID (r6: applied last, only without a second stage); flags2: SIS3D_EPI_RELU.  SIS3D_EUNSUPPORTED for (cin, cout, cout2) combinations that are not
 * instantiated (callers fall back to sis3d_conv3d_pw_chain / sis3d_conv3d). */
size_t sis3d_conv_pw16_packed_floats(int cout, int cin);
int sis3d_conv_pw16_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream);
int sis3d_conv3d_pw16(const float *in, int64_t nvox, int cin, int cin_stride, const float *packed_w, const float *bias, int cout,
                      int flags, const float *residual, int res_stride, float *out, int out_stride, int out_coff,
                      const float *packed_w2, const float *bias2, int cout2, int flags2, float *out2, int out2_stride,
                      sis3d_stream_t stream);

/* Conv3d(cin, cout, k=2, s=2) (+ bias, ReLU) as the same register-chained GEMM with 8 gathered input rows per output voxel
 * (the stems geometry1[4] / color[4], lib/nets/backbones.py:193,207), optionally chained into the following Bottleneck's
 * conv1 (cout -> cout2).  in: (X,Y,Z) channels-last; output grid (X/2,Y/2,Z/2).  packed_w = sis3d_conv_pw16_pack_weight of
 * the weight viewed as (Cout, 8*Cin) with column index tap*Cin + ci, tap = 4 dx + 2 dy + dz. */
int sis3d_conv3d_k2s2_pw16(const float *in, int X, int Y, int Z, int cin, int cin_stride, const float *packed_w, const float *bias,
                           int cout, int flags, float *out, int out_stride, int out_coff, const float *packed_w2,
                           const float *bias2, int cout2, int flags2, float *out2, int out2_stride, sis3d_stream_t stream);
/* geometry1[0] = Conv3d(2, cout, k=2, s=2, bias=False) + ReLU on the PLANAR 2-channel grid (backbones.py:188; same input
 * addressing as sis3d_conv3d_planar2) chained into the first Bottleneck's conv1 (cout -> cout2, + bias2, ReLU).
 * packed_w = sis3d_conv_pw16_pack_weight of the weight viewed as (Cout, 16) (column = ci*8 + 4 dx + 2 dy + dz). */
int sis3d_conv3d_stem_planar2(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, int X, int Y, int Z,
                              const float *packed_w, int cout, int flags, float *out, int out_stride, const float *packed_w2,
                              const float *bias2, int cout2, int flags2, float *out2, int out2_stride, sis3d_stream_t stream);
/* Both RPN heads of BOTH pyramid levels in one launch (lib/nets/network.py:41-42,541-549): per level the 1x1x1 convs
 * rpn_cls_score_net (2A rows) and rpn_bbox_pred_net (6A rows) stacked into one (8A, 256) matrix (pw16 pack), outputs in the
 * reference's permuted layouts: score / prob (2,X,Y,Z,A) (prob = softmax over the two class planes; may be NULL),
 * bbox (X,Y,Z,6A).  in?: the level's rpn_net output, nvox rows of cin_stride floats, cin = 256. */
int sis3d_rpn_heads(const float *in1, const float *packed_w1, const float *bias1, int anchors1, float *score1, float *prob1,
                    float *bbox1, const float *in2, const float *packed_w2, const float *bias2, int anchors2, float *score2,
                    float *prob2, float *bbox2, int64_t nvox, int cin, int cin_stride, sis3d_stream_t stream);

/* Conv3d(cin, cout, 3, padding=1) + bias + ReLU in the balanced "one workgroup per CU, one wave per SIMD" form
 * (csrc/conv3d_t16.hip): v_mfma_f32_16x16x4_f32, workgroup = brick of voxels x one 16-wide cout tile, the four waves
 * split the input channels.  Replaces the same cuDNN calls as sis3d_conv3d(ksize 3) for cin % 32 == 0, cout % 4 == 0;
 * nprob (<= 4) independent same-shape problems per launch (host arrays of device pointers, as sis3d_conv3d_batched).
 * Weights: sis3d_conv_k3t16_pack_weight from the checkpoint layout (Cout,Cin,3,3,3) into
 * [cout/16][cin/32][4][27][64][2] (sis3d_conv_k3t16_packed_floats floats).  flags: 0 or SIS3D_EPI_RELU.
 * brick: index of the voxel brick (0: 6x6x12, 1: 6x6x6, 2: 3x6x6, 3: 3x3x6, 4: 4x4x4, 5: 4x4x8, 6: 4x8x8) or -1 =
 * sis3d_conv3d_k3t16_brick's choice (fewest SIMD-cycles on the busiest CU for this grid).  Any grid size; partial
 * bricks are masked. */
/* sis3d_conv3d_k3t16_brick(..., max_voxels): cap on the brick volume the choice may take (0 = none).  One chunk alone is fastest
 * on the largest brick (one workgroup per CU); with several chunks in flight on separate streams bricks of <= 108 voxels (46 KB of
 * LDS, 3 workgroups per CU) let the streams' kernels share the CUs (+3-4 % throughput, profiles/README.md).  The cap is an ARGUMENT
 * (r5; a process-wide setter before): the caller that knows its regime asks for the brick and passes it to sis3d_conv3d_k3t16. */
/* profiling hook (tools/t16_trace.py): every later k3t16 launch writes {start, end (100 MHz wall clock ticks), HW_ID} of each of its
 * first capacity_blocks workgroups into buf (3 x int64 per workgroup, device memory); NULL switches it off */
int sis3d_conv3d_k3t16_set_trace(void *buf, int capacity_blocks);
size_t sis3d_conv_k3t16_packed_floats(int cout, int cin);
int sis3d_conv_k3t16_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream);
int sis3d_conv3d_k3t16_brick(int X, int Y, int Z, int cin, int cout, int nprob, int max_voxels);
int sis3d_conv3d_k3t16(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                       const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                       int out_stride, int out_coff, int brick, sis3d_stream_t stream);

/* ---- k3 / pad-1 convolution by Winograd F(2x2x2, 3x3x3), exact fp32 (csrc/conv3d_wino.hip) --------------------------------
 * Replaces the same cuDNN calls as sis3d_conv3d_k3t16 (nn.Conv3d(C, C', 3, padding=1) + bias + ReLU: lib/nets/backbones.py:20-22,
 * 188-231, lib/nets/network.py:40) with 3.375x fewer multiplications: every operation is a binary32 add or an fp32 MFMA, the
 * transform matrices hold 0, +-1, +-1/2 only, so the result differs from a direct fp32 convolution by summation order /
 * association (same error class: ~2-3e-6 against float64 on the rpn_net layer for both).  1..4 same-shape problems per launch,
 * channels-last activations, any grid size (partial 2x2x2 output blocks are masked), cin % 8 == 0, output may be a channel
 * slice (out_stride, out_coff).  Weights: sis3d_conv_k3wino_pack_weight transforms (Cout,Cin,3,3,3) once (fp32) into
 * U[cout tile][cin / 4][xi / 4][lane 64][4] (sis3d_conv_k3wino_packed_floats floats).  flags: SIS3D_EPI_RELU |
 * SIS3D_DISPATCH_SHARED_CHIP. */
/* 1 when the Winograd kernel is expected to beat sis3d_conv3d_k3t16 on this layer (enough (block, cout pair) work items to fill
 * the chip; measured table in csrc/conv3d_wino.hip), else 0: the host-side dispatch rule */
/* shared_chip != 0 (and SIS3D_DISPATCH_SHARED_CHIP in sis3d_conv3d_k3wino's flags): the launch shares the chip with other streams'
 * kernels (several chunks in flight), so what counts is CU-time, not the launch's own duration: layers that would take ONE cout tile
 * per workgroup to fill the chip alone (geometry2[0]) take two -- fewer, longer work items -- and layers with >= 48 work items
 * (the 64 -> 64 convs) take this kernel at all.  A per-call argument (r5; a process-wide setter before): the two regimes can be
 * launched / captured concurrently from different threads.  For one layer the two cout-tile counts give bit-identical results;
 * a layer that changes KERNEL between the regimes (direct <-> Winograd) differs by summation order (~2e-5 relative). */
#define SIS3D_DISPATCH_SHARED_CHIP 0x100
int sis3d_conv3d_k3wino_prefer(int X, int Y, int Z, int cin, int cout, int nprob, int shared_chip);
size_t sis3d_conv_k3wino_packed_floats(int cout, int cin);
int sis3d_conv_k3wino_pack_weight(const float *w, int cout, int cin, float *packed, sis3d_stream_t stream);
int sis3d_conv3d_k3wino(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                        const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                        int out_stride, int out_coff, sis3d_stream_t stream);
/* the same launch with ONE MORE ROW of workgroups that do not convolve (r5): the first eight of them copy n floats from the `next_src`
 * of the mailbox slot this graph's first node fetched (mail_state, sis3d_mail_fetch) into upload_dst (the STAGING buffer
 * sis3d_mail_upload of the next pass is given) and record next_src in mail_state[24..25], the others leave at once -- the pipeline's
 * NEXT chunk is pulled across PCIe while the real workgroups compute, so the upload hides under the longest kernel of a chunk
 * instead of stalling the pipeline in front of it.  Only the two-cout-tile form has the branch (the rpn_net pair):
 * SIS3D_EUNSUPPORTED otherwise -- the caller launches sis3d_conv3d_k3wino and the next pass uploads at its head. */
int sis3d_conv3d_k3wino_piggyback(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                                  const float *const *packed_ws, const float *const *biases, int cout, int flags, float *const *outs,
                                  int out_stride, int out_coff, uint32_t *mail_state, float *upload_dst, int64_t n,
                                  sis3d_stream_t stream);
/* ragged batch (the mask head's crops, lib/nets/network.py:303-317): one launch per k3 layer for all boxes.  Work items are
 * (crop, 8 x 4 x 8 block, group of two cout tiles); desc_dev = ndesc descriptors {int X,Y,Z, nbx,nby,nbz, block0, pad; int64 in_off,
 * out_off} (the layout of sis3d_conv3d_k3t16_ragged) with block0 counting blocks x groups, total_blocks their sum.
 * sis3d_ragged_tiling_k3wino returns the block and the group count the caller sizes the table with. */
int sis3d_ragged_tiling_k3wino(int cin, int cout, int *bx, int *by, int *bz, int *ngroups);
int sis3d_conv3d_k3wino_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout, int flags,
                               float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks, sis3d_stream_t stream);

/* The same ragged batch on MINI geometry (r4): a work item is a QUAD of 4 x 4 x 4-voxel bricks (2 x 2 x 2 Winograd tiles each, taken
 * in z-fastest order from the crop's ceil(X/4) x ceil(Y/4) x ceil(Z/4) grid of them) x one group of two cout tiles, so a crop is
 * covered with 4-voxel granularity on every axis instead of 8 x 4 x 8 blocks.  Descriptors as above with nbx / nby / nbz = minis per
 * axis and block0 = work items (quads x groups, quads = ceil(minis / 4)) in front of the crop; total_items = their sum.  Same
 * arithmetic as sis3d_conv3d_k3wino_ragged (identical results: the tiles are the same, only their grouping differs). */
int sis3d_ragged_tiling_k3wino_mini(int cin, int cout, int *mini_edge, int *minis_per_item, int *ngroups);
int sis3d_conv3d_k3wino_ragged_mini(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout, int flags,
                                    float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_items, sis3d_stream_t stream);

/* nprob (<= 4) INDEPENDENT convolutions of identical shape in ONE launch (different input / weights / bias /
 * residual / output pointers; host arrays of device pointers, read at call time).  Used for the two RPN levels
 * (lib/nets/network.py:539,552): their 432 workgroups each leave 80 of the 256 CUs a workgroup short, a single
 * 864-workgroup grid evens that out.  Same argument meaning as sis3d_conv3d. */
int sis3d_conv3d_batched(int nprob, const float *const *ins, int X, int Y, int Z, int cin, int cin_stride,
                         const float *const *packed_ws, const float *const *biases, int cout, int ksize, int stride, int flags,
                         const float *const *residuals, int res_stride, float *const *outs, int out_stride, int out_coff,
                         sis3d_stream_t stream);

/* first layers: NCDHW (planar) 2-channel grid in, channels-last out
 * (geometry1.0: Conv3d(2,32,k2,s2), backbones.py:188 ; mask head conv0: Conv3d(2,64,k3,p1), backbones.py:241).
 * w in checkpoint layout (Cout,2,k,k,k).  in element (c,x,y,z) at c*is_c + x*is_x + y*is_y + z (z contiguous);
 * the output covers the window [x0,x0+OX*stride) etc of the grid (crop for the mask head). */
int sis3d_conv3d_planar2(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, int X, int Y, int Z, int x0, int y0,
                         int z0, int OX, int OY, int OZ, const float *w, int cout, int ksize, int flags, float *out,
                         int out_stride, sis3d_stream_t stream);

/* ---- ragged batches: the mask head of lib/nets/network.py:303-317 (one conv stack per detected box, each on a
 * different dx x dy x dz crop) as ONE launch per layer for all boxes.  Activations of all problems are packed back to
 * back in one channels-last buffer; `desc_dev` is a device array of
 *   struct { int32 X,Y,Z, nbx,nby,nbz, block0, pad; int64 in_off, out_off; }           (48 bytes, sis3d_conv3d_ragged)
 *   struct { int32 x0,y0,z0, dx,dy,dz, pad,pad; int64 t0, out_off; }                  (48 bytes, sis3d_conv3d_planar2_ragged)
 * built by the host: nb? = ceil(dim / b?) with the brick of sis3d_ragged_tiling, block0 = running sum of
 * nbx*nby*nbz*ngroups; t0 = running sum of dx*dy*dz*cout/4; offsets in elements. */
int sis3d_ragged_tiling(int cin, int cout, int ksize, int *bx, int *by, int *bz, int *ngroups);
int sis3d_conv3d_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout, int ksize,
                        int flags, float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks,
                        sis3d_stream_t stream);
int sis3d_conv3d_planar2_ragged(const float *in, int64_t is_c, int64_t is_x, int64_t is_y, const void *desc_dev, int ndesc,
                                int64_t total_items, const float *w, int cout, int flags, float *out, int out_stride,
                                sis3d_stream_t stream);

/* the k3 layers of the ragged mask-head batch on the balanced kernel (csrc/conv3d_t16.hip); same descriptor struct as
 * sis3d_conv3d_ragged, bricks / groups from sis3d_ragged_tiling_k3t16(brick in {2,3,4,5} of the sis3d_conv3d_k3t16 list; the
 * caller picks the one with the fewest padded tile slots over its crops); packed_w from sis3d_conv_k3t16_pack_weight */
int sis3d_ragged_tiling_k3t16(int cin, int cout, int brick, int *bx, int *by, int *bz, int *ngroups, int *tiles_per_wave);
int sis3d_conv3d_k3t16_ragged(const float *in, int cin, int cin_stride, const float *packed_w, const float *bias, int cout, int flags,
                              float *out, int out_stride, const void *desc_dev, int ndesc, int64_t total_blocks, int brick,
                              sis3d_stream_t stream);

/* The body of a Bottleneck after its conv1 (lib/nets/backbones.py:27-40) in ONE launch (csrc/bottleneck.hip):
 *     y2  = relu(conv2(y1) + b2)            Conv3d(planes, planes, 3, padding=1); y1 = rows of `planes` floats
 *     out = relu(conv3(y2) + b3 + x)        Conv3d(planes, cio, 1) + residual x (rows of res_stride floats) -> channels
 *                                           [out_coff, out_coff + cio) of rows of out_stride floats
 *     y1n = relu(conv1_next(out) + b1n)     optional (c2 > 0): the NEXT block's conv1, rows of c2 floats
 * The workgroup that owns a brick of voxels owns all `planes` channels of conv2 (the balanced k3 scheme of
 * sis3d_conv3d_k3t16), so the 1x1x1 tail runs on the brick while it is on the CU; `out` is bit-identical to
 * sis3d_conv3d_k3t16 followed by sis3d_conv3d_pw16.  w2_t16: sis3d_conv_k3t16_pack_weight(planes, planes); w3_pw16 / w1n_pw16:
 * sis3d_conv_pw16_pack_weight.  brick: 0 = 6x6x6, 1 = 3x3x3, 2 = 6x6x3 (two workgroups per CU), -1 = sis3d_bottleneck16_brick's choice (which is
 * -1 itself -> SIS3D_EUNSUPPORTED when the two-launch path is the better one: 64-plane blocks).  (planes, cio, c2)
 * instantiated: (32,32,{0,32}) (32,64,0) (32,128,{0,32}); others -> SIS3D_EUNSUPPORTED. */
int sis3d_bottleneck16_brick(int X, int Y, int Z, int planes);
int sis3d_bottleneck16(const float *y1, int X, int Y, int Z, int planes, const float *w2_t16, const float *b2, const float *w3_pw16,
                       const float *b3, int cio, const float *residual, int res_stride, float *out, int out_stride, int out_coff,
                       const float *w1n_pw16, const float *b1n, int c2, float *y1n, int brick, sis3d_stream_t stream);

/* The same Bottleneck body with conv2 on the Winograd kernel (csrc/conv3d_wino.hip; F(2x2x2, 3x3x3) in exact fp32: 3.375x fewer
 * multiplications than sis3d_bottleneck16's direct convolution, same binary32 arithmetic class) and the 1x1x1 tail on the output tile
 * in the kernel's epilogue:  y2 = relu(conv2(y1) + b2);  out = relu(conv3(y2) + b3 + x);  y1n = relu(conv1_next(out) + b1n) (c2 > 0).
 * Replaces the cuDNN calls behind Bottleneck.forward (lib/nets/backbones.py:27-40) for planes = 32 (the workgroup's two cout tiles
 * are all of conv2's channels).  w2_wino: sis3d_conv_k3wino_pack_weight(32, 32); w3_pw16 / w1n_pw16: sis3d_conv_pw16_pack_weight.
 * (planes, cio, c2) instantiated: (32,32,{0,32}) (32,64,0) (32,128,0); others -> SIS3D_EUNSUPPORTED.  sis3d_bottleneck_wino_prefer: 1 where this
 * launch is expected to beat sis3d_bottleneck16 (>= 200 blocks of 8 x 4 x 8 voxels: the 48 x 24 x 48 maps; with shared_chip != 0,
 * see sis3d_conv3d_k3wino_prefer, >= 24 blocks: the Bottleneck(128, 32) bodies of the 24 x 12 x 24 maps as 27 fat work items). */
int sis3d_bottleneck_wino_prefer(int X, int Y, int Z, int planes, int cio, int c2, int shared_chip);
int sis3d_bottleneck_wino(const float *y1, int X, int Y, int Z, int planes, const float *w2_wino, const float *b2, const float *w3_pw16,
                          const float *b3, int cio, const float *residual, int res_stride, float *out, int out_stride, int out_coff,
                          const float *w1n_pw16, const float *b1n, int c2, float *y1n, sis3d_stream_t stream);

/* nn.MaxPool3d(3,1,1) (backbones.py:206,210,220), channels-last, -inf padding.  The C output channels land at
 * [out_coff, out_coff + C) of rows of out_stride floats (out_stride = C, out_coff = 0: a plain tensor; otherwise a channel
 * range of a wider tensor = the torch.cat of backbones.py:109 done in place). */
int sis3d_maxpool3d_3x3x3(const float *in, int X, int Y, int Z, int C, float *out, int out_stride, int out_coff,
                          sis3d_stream_t stream);

/* layout helpers: planar (C,X,Y,Z) <-> channels-last (X,Y,Z,C) */
int sis3d_planar_to_cl(const float *in, int C, int64_t nvox, float *out, sis3d_stream_t stream);
int sis3d_cl_to_planar(const float *in, int C, int64_t nvox, float *out, sis3d_stream_t stream);

/* ---------------------------------------------------------------- whole-scene merge --
 * New (the reference never chunks a scene; BASELINE config 5 / SURVEY 8e): what follows the all-gather of the per-chunk
 * record blocks.  blocks [n_chunks][1 + k_rows*width] fp32, slot 0 of a block = its valid row count.  The valid rows are
 * ordered by column score_col, descending and STABLE (ties: chunk id, then row -- torch.sort(stable=True) on the
 * flattened table), gathered into recs [<= n_chunks*k_rows][width], and the 3D NMS of sis3d_nms runs on columns
 * box_col..box_col+5 of the sorted rows.  order[j] = flat index (chunk*k_rows + row) of sorted row j; keep = ascending
 * kept positions in recs; counts[0] = number of valid rows, counts[1] = number kept (max_keep > 0 cuts the list).
 * One stream-ordered launch sequence, no host readback.  n_chunks*k_rows <= 8192, else SIS3D_EUNSUPPORTED. */
size_t sis3d_scene_merge_workspace_bytes(int n_chunks, int k_rows);
int sis3d_scene_merge(const float *blocks, int n_chunks, int k_rows, int width, int score_col, int box_col, float thresh,
                      int max_keep, float *recs, int32_t *order, int64_t *keep, int32_t *counts, void *workspace,
                      size_t workspace_bytes, sis3d_stream_t stream);

/* ---- ENet 2D encoder of the RGB image path (lib/nets/enet.py:130-694 `create_enet`, run by lib/nets/network.py:203-205 as
 * image_enet_trainable(image_enet_fixed(images)), eval mode) -- csrc/enet.hip.  These replace the cuDNN / MIOpen operator calls behind
 * the nn.Conv2d / nn.BatchNorm2d / nn.PReLU / nn.MaxPool2d modules of that Sequential: BatchNorm (eval) and the torch7-style dropout
 * scale are folded into the weights by the caller (sis3d/nets/enet_hip.py), activations are rows of channels per pixel (NHWC),
 * weights are in the lane order of the 16x16x4 fp32 MFMA: [cout/16][cin/16][64][4] per filter tap (lane 16 kq + i of tile (ct, g)
 * holds W[16 ct + i][16 g + 4 kq + r], r = 0..3; the layout sis3d_conv_pw16_pack_weight writes), taps in (ky, kx) order.
 *   sis3d_enet_initial  cat(Conv2d(3,13,3,stride 2,padding 1)(x), MaxPool2d(2,2)(x)) -> BatchNorm2d(16) -> PReLU(16) (enet.py's first
 *                       four entries): NCHW images (V,3,Hi,Wi), Hi / Wi even -> rows of 16 floats at Hi/2 x Wi/2.  w (13,3,3,3) / b (13):
 *                       the filters with the BatchNorm affine folded in; pool_scale / pool_shift (3): the affine of the pooled channels.
 *   sis3d_enet_conv1    a bottleneck's first convolution + BatchNorm + PReLU: taps = 4: Conv2d(cin, mid, 2, stride 2) of a down block
 *                       (x at 2H x 2W -> y1 at H x W); taps = 1: the 1x1 reduction.  (cin, mid) in {(16,16), (64,16), (64,32), (128,32)}.
 *   sis3d_enet_block    the rest of a bottleneck in one launch: y2 = prelu(conv2(y1) + b2, s2) with conv2 = 3x3 / dilation `dil`
 *                       (kind 0, w2 = 9 taps) or the asymmetric Conv2d(1x5, no bias) -> Conv2d(5x1) pair (kind 1, w2 / w2b = 5 taps each);
 *                       out = prelu(conv3(y2) + b3 + skip, s3), skip = x (pool_cin = 0) or MaxPool2d(2,2)(x) with x = rows of pool_cin
 *                       floats at 2H x 2W, zero channels appended (down blocks); out = rows of c floats, or NCHW (V,c,H,W) if out_nchw;
 *                       midn > 0: y1n = prelu(conv1_next(out) + b1n, s1n), the next bottleneck's 1x1 reduction (rows of midn floats).
 *                       (c, mid, midn) in {(64,16,16), (64,16,0), (128,32,32), (128,32,0)}; others -> SIS3D_EUNSUPPORTED. */
int sis3d_enet_initial(const float *images, int V, int Hi, int Wi, const float *w, const float *b, const float *pool_scale,
                       const float *pool_shift, const float *slope, float *out, sis3d_stream_t stream);
int sis3d_enet_conv1(const float *x, int V, int H, int W, int cin, int mid, int taps, const float *w, const float *b, const float *slope,
                     float *y1, sis3d_stream_t stream);
int sis3d_enet_block(const float *x, const float *y1, int V, int H, int W, int c, int mid, int kind, int dil, const float *w2, const float *b2,
                     const float *s2, const float *w2b, const float *w3, const float *b3, const float *s3, int pool_cin, float *out, int out_nchw,
                     const float *w1n, const float *b1n, const float *s1n, int midn, float *y1n, sis3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIS3D_H */
