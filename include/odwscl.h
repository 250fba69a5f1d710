/*
 * odwscl.h -- C-ABI of libodwscl.so: the MI355X (gfx950) native operators of
 * OD-WSCL's proposal-feature hot path.
 *
 * This is the drop-in boundary.  The reference binds its native operators
 * through the pybind11 module `wetectron._C` (csrc/vision.cpp:9-25); every
 * entry point below names the reference interface it replaces.  INTEGRATION.md
 * shows the ctypes stub a wetectron maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch/ATen types; all pointers are DEVICE
 *     pointers unless a parameter is documented as host
 *   - the caller owns every buffer (outputs and workspaces are caller-allocated)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing synchronises (safe to call from the autograd thread)
 *   - return 0 on success, a negative ODW_E* code otherwise; odw_last_error()
 *     returns a thread-local message (the Python layer raises RuntimeError,
 *     mirroring AT_ASSERTM/AT_ERROR in the reference)
 *   - no global mutable state besides the thread-local error string
 *   - tensors are dense row-major fp32 unless stated; rois are (R,5) rows
 *     [batch_index, x1, y1, x2, y2] exactly as modeling/poolers.py:85-96 builds
 */
#ifndef ODWSCL_H_
#define ODWSCL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODW_OK 0
#define ODW_EINVAL (-1)   /* bad argument                        */
#define ODW_ELAUNCH (-2)  /* hipGetLastError() after a launch    */
#define ODW_EWORKSPACE (-3) /* workspace too small               */

const char* odw_last_error(void);
int odw_version(void);

/* ---- ROIPool --------------------------------------------------------------
 * replaces _C.roi_pool_forward / roi_pool_backward
 * (csrc/ROIPool.h:11-45 -> csrc/cuda/ROIPool_cuda.cu:17-108,110-202).
 * feat (B,C,H,W); out (R,C,PH,PW) fp32; argmax (R,C,PH,PW) int32 (h*W+w or -1).
 * workspace: odw_roi_pool_workspace(R,PH,PW) bytes (per-ROI bin tables). */
int64_t odw_roi_pool_workspace(int R, int PH, int PW);
/* Workspace of the faster (ROI, 64-channel) form of odw_roi_pool_forward (round 4: the bin table + an NHWC ordinal image of
 * the map; C % 8 == 0, PH * PW <= 64).  With only odw_roi_pool_workspace bytes the plane-resident kernels run. */
int64_t odw_roi_pool_forward_workspace(int B, int C, int H, int W, int R, int PH, int PW);
int odw_roi_pool_forward(const float* feat, const float* rois, float spatial_scale,
                         int B, int C, int H, int W, int R, int PH, int PW,
                         float* out, int32_t* argmax, void* workspace, int64_t workspace_bytes,
                         void* stream);
/* grad_in (B,C,H,W) is fully written (zero where nothing pools). */
int odw_roi_pool_backward(const float* grad_out, const int32_t* argmax, const float* rois,
                          int B, int C, int H, int W, int R, int PH, int PW,
                          float* grad_in, void* stream);
/* Deterministic form of the backward: fixed-point accumulation (one power-of-two scale per launch from an atomicMax
 * pre-pass, 64-bit integer LDS atomics) -- bit-identical from run to run, where the reference's float atomicAdd
 * (ROIPool_cuda.cu:101) is not.  workspace: >= 4 bytes of device memory. */
int odw_roi_pool_backward_det(const float* grad_out, const int32_t* argmax, const float* rois,
                              int B, int C, int H, int W, int R, int PH, int PW,
                              float* grad_in, void* workspace, int64_t workspace_bytes, void* stream);
/* ROIPool fused with the operand staging of the first head GEMM (same pooling semantics as odw_roi_pool_forward:
 * csrc/cuda/ROIPool_cuda.cu:17-108): X (2R x ld) bf16, row n = pooled features of ROI n flattened (C, PH, PW), row
 * R+n = ((x * keep[n][bin]) * R*PH*PW) / *keep_sum (DropBlock2D.forward, drop_block.py:45-50; keep = NULL: only
 * the R clean rows); argmax 16-bit (0xFFFF = empty bin; H*W < 65535).  _backward scatters the gradient of both halves
 * of X -- plus E parked fp32 gradient rows `extra` belonging to ROIs `extra_roi` (the sampled-row views of the
 * contrastive loss) -- through the argmax into grad_in (B, C, H, W) fp32; skip_clean: rows [0, R) of dX are not read
 * (row-sparse backward: their gradient arrives through `extra`). */
int odw_roi_pool_stack_forward(const float* feat, const float* rois, float spatial_scale, int B, int C, int H, int W,
                               int R, int PH, int PW, const float* keep, const float* keep_sum, void* X_bf16, int ld,
                               void* argmax_u16, void* workspace, int64_t workspace_bytes, void* stream);
/* The same forward (7x7 bins) reading the backbone's NHWC bf16 map (B, H, W, C), C % 64 == 0: one workgroup per
 * (ROI, 64 channels) = 6272 contiguous output bytes per row, full 16-byte stores. */
int64_t odw_roi_pool_stack_nhwc_workspace(int R, int B, int C, int H, int W);
int odw_roi_pool_stack_forward_nhwc(const void* feat_nhwc_bf16, const float* rois, float spatial_scale, int B, int C,
                                    int H, int W, int R, const float* keep, const float* keep_sum, void* X_bf16, int ld,
                                    void* argmax_u16, void* workspace, int64_t workspace_bytes, void* stream);
/* The same pooling on an fp32 NHWC map with the results written as bf16 PLANES (precision mode "bf16x2f": the split-
 * precision forward of the first head GEMM reads its operand as `T` column blocks of width `block` >= C*49 holding
 * plane pattern[t] (0 hi, 1 mid, 2 lo, 3 zeros; HOST array) of each value): X_planes (2R x ld) bf16, rows [0,R) the
 * pooled maxima, rows [R,2R) their DropBlock view (keep nullable: clean rows only); pooled_f32 (R x C*49, nullable)
 * the fp32 maxima themselves; argmax_u16 as above.  Replaces roi_pool_forward + stack_clean_aug_f32 + split_rows_bf16. */
int64_t odw_roi_pool_stack_nhwc_f32_workspace(int R, int B, int C, int H, int W);
int odw_roi_pool_stack_forward_nhwc_f32(const float* feat_nhwc, const float* rois, float spatial_scale, int B, int C,
                                        int H, int W, int R, const float* keep, const float* keep_sum,
                                        const int* pattern, int T, void* X_planes, int64_t ld, int block,
                                        float* pooled_f32, void* argmax_u16, void* workspace,
                                        int64_t workspace_bytes, void* stream);
/* ... and (X_cm non-NULL) the clean rows once more as the two CELL-MAJOR planes odw_gemm_nt_cm reads: X_cm (R x ld_cm)
 * bf16, X_cm[r][bin * C + c] = hi, X_cm[r][cm_mid + bin * C + c] = mid.  With the shared clean + DropBlock fc6 forward
 * X_planes then only carries what the BACKWARD reads (pattern {0}: the hi plane of both halves). */
int odw_roi_pool_stack_forward_nhwc_f32_cm(const float* feat_nhwc, const float* rois, float spatial_scale, int B, int C,
                                           int H, int W, int R, const float* keep, const float* keep_sum,
                                           const int* pattern, int T, void* X_planes, int64_t ld, int block,
                                           float* pooled_f32, void* argmax_u16, void* X_cm, int64_t ld_cm, int64_t cm_mid,
                                           void* workspace, int64_t workspace_bytes, void* stream);
int odw_roi_pool_stack_backward(const void* dX, int dx_is_f32, int ld, const void* argmax_u16, const float* rois,
                                const float* keep, const float* keep_sum, const float* extra, const int* extra_roi,
                                int E, int skip_clean, int B, int C, int H, int W, int R, int PH, int PW, float* grad_in,
                                void* stream);
/* Workspace form (>= 4 bytes): fixed-point accumulation -- deterministic, and 64-bit integer LDS atomics run at 10x
 * the rate of ds_add_f32 on gfx950 (csrc/odw_fixed.h).  NULL workspace = the float-atomic form above. */
int odw_roi_pool_stack_backward_ws(const void* dX, int dx_is_f32, int ld, const void* argmax_u16, const float* rois,
                                   const float* keep, const float* keep_sum, const float* extra, const int* extra_roi,
                                   int E, int skip_clean, int B, int C, int H, int W, int R, int PH, int PW,
                                   float* grad_in, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- ROIAlign -------------------------------------------------------------
 * replaces _C.roi_align_forward / roi_align_backward
 * (csrc/ROIAlign.h:11-45 -> csrc/cuda/ROIAlign_cuda.cu:65-254,
 *  csrc/cpu/ROIAlign_cpu.cpp:114-257).  Legacy (un-aligned) sampling;
 * sampling_ratio <= 0 means adaptive ceil(roi/pooled). */
int odw_roi_align_forward(const float* feat, const float* rois, float spatial_scale,
                          int B, int C, int H, int W, int R, int PH, int PW, int sampling_ratio,
                          float* out, void* stream);
int odw_roi_align_backward(const float* grad_out, const float* rois, float spatial_scale,
                           int B, int C, int H, int W, int R, int PH, int PW, int sampling_ratio,
                           float* grad_in, void* stream);
/* The same backward with table space from the caller (odw_roi_align_backward_workspace bytes): the separable form --
 * per ROI 14 axis weight vectors are built once and every plane spends one LDS atomic per touched cell instead of four
 * taps per sample (16.7 -> ~1 ms at P = 2000 on 76x76x512).  NULL / too small a workspace = the sample form above. */
int64_t odw_roi_align_backward_workspace(int R, int PH, int PW);
/* The forward in the same separable form (workspace of odw_roi_align_backward_workspace bytes: the per-ROI axis
 * vectors; they are staged in LDS chunk by chunk): (bin+2)^2 cell reads per bin instead of 4 taps per sample.  The
 * sample coordinates are the reference's, the order of the fp32 additions is not (1e-6 against ROIAlign_cpu.cpp). */
/* Workspace of the (ROI, 64-channel) form of the forward (round 4: axis vectors + an NHWC copy of the map; C % 8 == 0,
 * PH * PW <= 64, PH + PW <= 32): ~2x the plane-resident form, the same values. */
int64_t odw_roi_align_forward_workspace(int B, int C, int H, int W, int R, int PH, int PW);
int odw_roi_align_forward_ws(const float* feat, const float* rois, float spatial_scale, int B, int C, int H, int W,
                             int R, int PH, int PW, int sampling_ratio, float* out, void* workspace,
                             int64_t workspace_bytes, void* stream);
int odw_roi_align_backward_ws(const float* grad_out, const float* rois, float spatial_scale, int B, int C, int H, int W,
                              int R, int PH, int PW, int sampling_ratio, float* grad_in, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* ---- NMS --------------------------------------------------------------------
 * mode ODW_NMS_TV : torchvision.ops.nms semantics -- the live path
 *                   (structures/boxlist_ops.py:32,57): IoU without +1, suppress
 *                   when IoU > thr, keep[] in descending-score order.
 * mode ODW_NMS_WT_GE / ODW_NMS_WT_GT : wetectron `_C.nms` (csrc/nms.h:10-28):
 *                   +1 areas, suppress when >= thr (csrc/cpu/nms_cpu.cpp:60) or
 *                   > thr (csrc/cuda/nms.cu:60), keep[] ascending original index.
 * boxes (n,4) xyxy, scores (n).  keep: int64[n] device; n_keep: int32[1] device.
 * Score ties are ordered by ascending index (stable sort).  n <= ODW_NMS_MAX_N.
 * workspace: odw_nms_workspace(n) bytes. */
#define ODW_NMS_TV 0
#define ODW_NMS_WT_GE 1
#define ODW_NMS_WT_GT 2
#define ODW_NMS_MAX_N 8192
int64_t odw_nms_workspace(int n);
int odw_nms(const float* boxes, const float* scores, int n, float thr, int mode,
            int64_t* keep, int32_t* n_keep, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- box IoU ----------------------------------------------------------------
 * replaces structures/boxlist_ops.py:127-160 (boxlist_iou, TO_REMOVE=1).
 * a (N,4), b (M,4) -> iou (N,M). */
int odw_box_iou(const float* a, int N, const float* b, int M, float* iou, void* stream);

/* ---- proposal x proposal similarity -----------------------------------------
 * replaces `torch.mm(sim_feature, sim_feature.T)` of
 * roi_heads/weak_head/loss.py:319.  E (P,D) fp32, D % 4 == 0 -> S (P,P) fp32. */
int odw_pairwise_sim(const float* E, int P, int D, float* S, void* stream);
/* The same with a row pitch of ldS >= P elements for S (D = 128, S 16-byte aligned; columns P..ldS-1 are not written).
 * The kernel leaves S as 128-byte row segments: with a pitch that is a multiple of 16 floats they are whole cache lines and
 * leave as nontemporal stores; otherwise every segment straddles two lines and has to merge in L2 (plain stores), ~15%
 * slower (P = 5000: 33.9 us dense, 29.5 us with ldS = 5024) -- a caller that may choose the layout of S passes ldS = P
 * rounded up to 32 and reads the (P, P) view of it. */
int odw_pairwise_sim_ld(const float* E, int P, int D, float* S, int64_t ldS, void* stream);
/* Workspace form (D = 128): the products run on the bf16 matrix cores as six plane products per fp32-grade product and the
 * kernel is bound by the 4 P^2-byte write of S.  Below odw_pairwise_sim_planes_min() rows (5600; ODW_PAIRWISE_PLANES_MIN
 * overrides) the ONE-launch panel kernel runs whatever the workspace (it splits E in registers: 5-7 us ahead at P <= 4000);
 * from there on, given odw_pairwise_sim_workspace(P, D) bytes, the split kernel + LDS-DMA kernel pair, which is ahead at
 * P >= 6000 (43.4 / 72.6 us against 48.5 / 80.7 us at P = 6000 / 8000).  Bit-identical results either way. */
int odw_pairwise_sim_planes_min(void);
int64_t odw_pairwise_sim_workspace(int P, int D);
int odw_pairwise_sim_ws(const float* E, int P, int D, float* S, void* workspace, int64_t workspace_bytes, void* stream);
/* The same product from the PLANES of E (round 4): [3 planes][Ppad][128] bf16, Ppad = P rounded up to 32, padding rows
 * zero, hi + mid + lo == the fp32 value exactly -- odw_pairwise_split_planes writes them (odw_pairwise_sim_workspace(P, 128)
 * bytes), or the producer of E does.  The planes reach LDS by DMA; results are bit-identical to the one-launch form (and no
 * faster: this entry exists for callers that already hold the planes).  Reference: roi_heads/weak_head/loss.py:319. */
int odw_pairwise_split_planes(const float* E, int P, void* planes, void* stream);
int odw_pairwise_sim_planes(const void* planes, int P, float* S, void* stream);

/* ---- SupConLossV2 ------------------------------------------------------------
 * replaces roi_heads/sim_head/sim_loss.py:49-80 forward + its autograd
 * backward.  F (N,D) fp32 rows, labels (N) int32, w (N) fp32 (detached),
 * temperature tau.  loss: fp32[1] = mean_i( -log(A_i/B_i) * w_i ) (not yet
 * multiplied by lmda).  dF (N,D) = dloss/dF * grad_scale, or NULL to skip.
 * workspace: odw_supcon_workspace(N) bytes. */
int64_t odw_supcon_workspace(int N);
int odw_supcon_v2(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                  float grad_scale, float* loss, float* dF,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* ---- counter-based randomness --------------------------------------------------
 * The reference draws dropout masks and the noise view from torch's device
 * generator (modeling/backbone/vgg16.py:124,127,177-180) and DropBlock centres
 * from the CPU generator (modeling/dropblock/drop_block.py:42).  Here every
 * draw is a pure function of (key, element index); (k0,k1) = the 64-bit key of
 * a (seed, stream) pair (od_wscl_amd/utils/rng.py:stream_key).
 *   uniform : out[i] = U[0,1) of element (i + offset)
 *   normal  : Box-Muller pairs of the stream
 *   dropout : out = x * [u >= p] / (1 - p); its backward is the same call on the
 *             gradient with the same key (no mask is stored)
 *   noise_mul : out = x + N(0,1) * x */
int odw_rng_uniform(float* out, int64_t n, uint32_t k0, uint32_t k1, uint32_t offset, void* stream);
int odw_rng_normal(float* out, int64_t n, uint32_t k0, uint32_t k1, void* stream);
/* DropBlock2D's keep mask (modeling/dropblock/drop_block.py:38-47, :55-71): block centres = (u < gamma) over ONE uniform
 * draw of n*h*w values from stream (k0, k1) -- element i of odw_rng_uniform --, dilated by a block_size max-pool
 * (padding block_size / 2, cropped to h x w), inverted; keep (n, h, w) fp32 in {0, 1}, *keep_sum = its sum (exact). */
int odw_dropblock_keep_mask(int n, int h, int w, int block_size, float gamma, uint32_t k0, uint32_t k1, float* keep,
                            float* keep_sum, void* stream);
int odw_dropout(const float* x, float* out, int64_t n, uint32_t k0, uint32_t k1, float p, void* stream);
int odw_noise_mul(const float* x, float* out, int64_t n, uint32_t k0, uint32_t k1, void* stream);

/* ---- object-discovery pseudo-label assignment -------------------------------------
 * replaces the tail of od_layer.__call__ / oicr_layer.__call__
 * (roi_heads/weak_head/pseudo_label_generator.py:171-190): IoU(+1) of every
 * proposal against the G pseudo-GT boxes, row max and FIRST argmax (the reference
 * does this on the host with numpy, :176-177), labels (0 where max IoU <= fg_thresh),
 * loss weights = gt_scores[argmax], regression targets = BoxCoder.encode
 * (modeling/box_coder.py:22-50) with weights (wx,wy,ww,wh).
 * boxes (P,4), gt_boxes (G,4), gt_classes int64 (G), gt_scores (G) ->
 * labels int64 (P), weights (P), targets (P,4). */
int odw_od_assign(const float* boxes, int P, const float* gt_boxes, const int64_t* gt_classes,
                  const float* gt_scores, int G, float fg_thresh, float wx, float wy, float ww, float wh,
                  int64_t* labels, float* weights, float* targets, void* stream);
/* Same assignment with the pseudo-GT given the way the discovery kernels emit them: int32 indices into `boxes` and
 * int32 classes (no gather / widening kernels in between). */
int odw_od_assign_indexed(const float* boxes, int P, const int* gt_index, const int* gt_classes,
                          const float* gt_scores, int G, float fg_thresh, float wx, float wy, float ww, float wh,
                          int64_t* labels, float* weights, float* targets, void* stream);
/* the same with the pseudo-GT count read from device memory (written by odw_discover_sim): no host read in front of it;
 * g_cap = capacity of the index / class / score lists */
int odw_od_assign_indexed_dev(const float* boxes, int P, const int* gt_index, const int* gt_classes,
                              const float* gt_scores, const int* n_gt_dev, int g_cap, float fg_thresh, float wx, float wy,
                              float ww, float wh, int64_t* labels, float* weights, float* targets, void* stream);

/* ---- ROI-head GEMM on the bf16 matrix cores ---------------------------------------
 * C[M,N] (+)= epilogue( alpha * sum_k A[M,K] B[N,K] )   both operands bf16, K contiguous.
 * Serves the forward / dgrad / wgrad products of every Linear layer on the path
 * (modeling/backbone/vgg16.py:121-127 fc6/fc7, roi_heads/sim_head/sim_net.py:12-16,
 * roi_heads/weak_head/roi_weak_predictors.py:158-165 fused into one N=5C+12C GEMM) -- the
 * reference calls cuBLAS through torch.nn.Linear.
 *   lda, ldb : row strides in elements, multiples of 8; K rounded up to 8 must fit (zero padded)
 *   C        : bf16 (c_is_bf16) or fp32, row stride ldc; accumulate (fp32 only): C += result
 *   epilogue : + bias[N] (nullable), ReLU, dropout(drop_p) with counter-based keys:
 *              nseg row segments (seg_rows[i] = first row, seg_keys[2i..2i+1] = key); element
 *              (m,n) of segment s uses index (m - seg_rows[s]) * N + n   (HOST arrays, <= 8)
 * odw_linear_bwd_prep: dZ = dY * [Y != 0] * scale (Y = saved output, nullable), emitted
 *   row-major (ld_z) and transposed (N x ld_t), both zero padded; db[n] += column sums.
 *   dy_is_f32 is a bit set: bit 0 = dY is fp32 (else bf16), bit 1 = Y is fp32 (else bf16; the saved
 *   output of a split-precision forward, precision mode "bf16x2f").
 * odw_transpose_to_bf16 / odw_f32_to_bf16: layout + precision helpers for the operands.
 * odw_sgd_momentum: fused SGD step over flat fp32 buffers (solver/build.py:10-24 semantics),
 *   optionally refreshing the bf16 shadow the GEMMs read. */
int odw_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C, int ldc,
                     int c_is_bf16, const float* bias, int relu, float alpha, float drop_p, int nseg,
                     const int* seg_rows, const uint32_t* seg_keys, int accumulate, void* stream);
/* Stacked clean + DropBlock operand of the first head GEMM, and its gradient (ROIWeakRegHead.forward,
 * roi_heads/weak_head/weak_head.py:107-112 runs forward() and forward_dropblock()+forward_neck() on the same pooled
 * features; DropBlock2D.forward, modeling/dropblock/drop_block.py:45-50: out = x * block * numel / sum):
 *   pooled (P, C, S) fp32, block (P, S) keep mask, block_sum = sum(block) on the device, numel = P*S
 *   out  (2P x ld) bf16: row p = x, row P+p = ((x * block) * numel) / sum        (ld >= C*S, padding zeroed)
 *   bwd : dpooled (P, C, S) fp32 = dX[p] + ((dX[P+p] * block) * numel) / sum,   dX (2P x ld) bf16 or fp32 */
int odw_stack_clean_aug(const float* pooled, const float* block, const float* block_sum, int P, int C, int S,
                        void* out_bf16, int ld, void* stream);
int odw_unstack_clean_aug_bwd(const void* dX, int dx_is_f32, int ld, const float* block, const float* block_sum,
                              int P, int C, int S, float* dpooled, void* stream);
/* The two contrastive views of k sampled proposals of one (image, class) (loss.py:292-305: drop_pool and noise_pool
 * of pooled[rows], vgg16.py:169-180) written straight into the bf16 GEMM operand: rows [out_row0, +k) =
 * ((x * keep) * numel) / sum with keep = !(u < gamma), u the counter-based uniform draw (kd0,kd1) over (k, S) and
 * sum = sum(keep) (left in keep_sum, a device scalar), rows [out_row0 + k, +k) = z*x + x with z the normal draw
 * (kn0,kn1) over (k, C, S).  rows = int32 indices relative to row_base; pooled = fp32 (P, C, S), or (src_is_bf16) the
 * bf16 rows of a stacked operand with the output's row stride.  _bwd folds the gradient of those 2k rows
 * back: dpooled[row_base + rows[r]] += ...  (launches on one stream are ordered; rows of one call are distinct). */
int odw_rows_drop_noise(const void* pooled, int src_is_bf16, const int* rows, int row_base, int k, int C, int S, float gamma,
                        uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1, float* keep_sum, void* out_bf16,
                        int ld, int out_row0, void* stream);
int odw_rows_drop_noise_bwd(const void* dX, int dx_is_f32, int ld, int dx_row0, const int* rows, int row_base, int k,
                            int C, int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1,
                            const float* keep_sum, float* dpooled, void* stream);
/* The same with dpooled[row_base + rows[r]] WRITTEN instead of added to: for rows that belong to this launch alone (the entries
 * of the pooling node's side buffer) -- the buffer then needs no zero fill and is not read. */
int odw_rows_drop_noise_bwd_store(const void* dX, int dx_is_f32, int ld, int dx_row0, const int* rows, int row_base, int k,
                                  int C, int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1,
                                  const float* keep_sum, float* dpooled, void* stream);
/* Row-wise L2 normalisation of the (R x D) embeddings (Sim_Net.forward, sim_head/sim_net.py:25-26, F.normalize with
 * eps): y = x / max(||x||, eps), norm[r] = ||x_r||; bwd: dx = (g - y (g.y)) / max(||x||, eps). */
int odw_l2norm_rows(const float* x, int R, int D, float eps, float* y, float* norm, void* stream);
int odw_l2norm_rows_bwd(const float* g, const float* y, const float* norm, int R, int D, float eps, float* dx,
                        void* stream);
/* ---- fc6 over CELL-MAJOR planes: clean + DropBlock outputs from one pass (round 4) -------------------------
 * replaces the two fc6 evaluations of ROIWeakRegHead.forward (roi_heads/weak_head/weak_head.py:107-112: the pooled
 * features and their DropBlock view, modeling/dropblock/drop_block.py:38-50) in the split precision mode "bf16x2f".
 * Operands: A (M x lda), B (N x ldb) bf16, each row = two planes [hi at 0 | mid at a_mid / b_mid] of the fp32 values,
 * laid out k' = s * C + c (cell-major; odw_split_rows_cm writes it, odw_roi_pool_stack_forward_nhwc_f32 can); the three
 * plane products hi.hi + hi.mid + mid.hi are summed.  keep == NULL: Cout (M x N fp32) = epilogue(A B^T) with the
 * epilogue arguments of odw_gemm_nt_bf16_ws (workspace: odw_gemm_nt_cm_workspace(M, N, S) bytes lets small M split
 * the reduction over cells).  keep (M x S floats, 0 = dropped) + keep_sum (device scalar): rows [0, M) of Cout get
 * the clean result and rows [drop_row0, drop_row0 + M) the result for x * keep * (M S / keep_sum), both from ONE
 * sweep over A (summation by parts over the cells, csrc/gemm_bf16.hip); dropout then takes two segments
 * (rows 0 and drop_row0). */
int64_t odw_gemm_nt_cm_workspace(int M, int N, int S);
int64_t odw_gemm_nt_cm_pair_workspace(int M, int N, int S);      /* the same for the pair form (needs drop_row0 == M) */
int odw_gemm_nt_cm(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M, int N, int C, int S,
                   const float* keep, const float* keep_sum, int drop_row0, float* Cout, int ldc, const float* bias,
                   int relu, float drop_p, int nseg, const int* seg_rows, const uint32_t* seg_keys, const int* row_ids,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* fp32 rows (R x C*S, k = c * S + s: the reference's flattening of (C, 7, 7)) -> the two cell-major planes above:
 * out[r][s * C + c] = hi, out[r][mid_off + s * C + c] = mid.  C % 64 == 0, S <= 64. */
int odw_split_rows_cm(const float* in, int64_t ld_in, int R, int C, int S, void* out, int64_t ld_out, int64_t mid_off,
                      void* stream);
/* Which kernel odw_gemm_nt_bf16 will launch for this product: 0 register-staged 128x128, 1 LDS-DMA 128x128,
 * 2 LDS-DMA 256x128 ring, 3 256x256 (per-kernel timing in bench.py names its roofline entry from this). */
int odw_gemm_nt_bf16_variant(int M, int N, int K, int lda, int ldb, const void* C, int ldc, int c_is_bf16);
/* Split-K form: products with a small C and a long K (conv weight gradients, the fc6 pass over the sampled rows)
 * fill the chip only when K is split.  odw_gemm_nt_bf16_workspace returns the bytes of fp32 partials the planner
 * wants for this product (0 = no split) and the kernel variant it would use; odw_gemm_nt_bf16_ws takes that
 * buffer (16-byte aligned; NULL / too small = unsplit, identical to odw_gemm_nt_bf16).  The reduction pass applies
 * the same fused epilogue in a fixed summation order.  row_ids (device int32[M], nullable): A holds a GATHERED
 * subset of the rows of a larger pass -- the dropout draw of GEMM row m is the one logical row row_ids[m] had there
 * (segment 0's key), so re-evaluating a few rows reproduces their original masks. */
int64_t odw_gemm_nt_bf16_workspace(int M, int N, int K, int lda, int ldb, const void* C, int ldc, int c_is_bf16,
                                   int* variant_out);
int odw_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C, int ldc,
                        int c_is_bf16, const float* bias, int relu, float alpha, float drop_p, int nseg,
                        const int* seg_rows, const uint32_t* seg_keys, const int* row_ids, int accumulate,
                        void* workspace, int64_t workspace_bytes, void* stream);
int odw_linear_bwd_prep(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                        float scale, void* dZ, int ld_z, void* dZT, int ld_t, float* db, void* stream);
int odw_transpose_to_bf16(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                          void* stream);
/* "_part" forms: the transposed output is a column block [0, cols) (zero padded from R / M up to cols) of a wider
 * matrix with row stride ld -- several evaluations of one Linear lay their dZ^T / X^T blocks side by side so that
 * ONE weight-gradient GEMM (K = all their rows) replaces one read-modify-write pass over the gradient per evaluation. */
int odw_transpose_to_bf16_part(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                               int out_cols, void* stream);
int odw_linear_bwd_prep_part(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                             float scale, void* dZ, int ld_z, void* dZT, int ld_t, int t_cols, float* db, void* stream);
int odw_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int odw_sgd_momentum(float* p, const float* g, float* buf, void* shadow_bf16, int64_t n, float lr, float wd,
                     float momentum, float grad_scale, int first_step, void* stream);
/* The same update on at most `max_workgroups` workgroups (0 = as many as the buffer wants): an optimiser pass that runs
 * BESIDE other kernels (the head's 600 MB on a side stream during the backbone's backward) is paced so that its 3.4 GB
 * of HBM traffic spread over that time instead of starving whatever runs next to it. */
int odw_sgd_momentum_paced(float* p, const float* g, float* buf, void* shadow_bf16, int64_t n, float lr, float wd,
                           float momentum, float grad_scale, int first_step, int max_workgroups, void* stream);

/* ---- OD-WSCL selection logic on the device ------------------------------------------
 * replaces the Python loops of roi_heads/weak_head/loss.py:281-345 (IoU sampling, object
 * discovery) and the pseudo-GT bookkeeping of od_layer (pseudo_label_generator.py:143-166);
 * one workgroup per image, sets as P-bit masks in LDS, no host synchronisation inside.
 *   s0,s1,s2  : the three branch score matrices (sumP x C): final_score, softmax(ref1), softmax(ref2)
 *   img_off   : int32[n_img+1] first proposal of each image; pos_cls int32[n_img][maxpos] positive
 *               class indices c (0-based over foreground, ascending); n_pos int32[n_img]
 * discover_iou -> tops[n_img][3][maxpos] (first argmax of column c+1), masks uint32[n_img][maxpos][ceil(max_p/32)]
 *               (pgt_index bit sets), rows int32[n_img][maxpos][pstride] (ascending), counts[n_img][maxpos]
 * discover_sim : E (sumP x 128) embeddings; bank (rows x 128) class-major banks with bank_off/bank_cnt[C-1];
 *               updates masks; emits per (img,branch,class) NMS survivors in descending class score
 *               (inst_*), the not-yet-known survivors ascending (fresh_*), and per (img,branch) the od_layer
 *               pseudo-GT lists gt_idx/gt_cls/gt_score [n_img][3][maxpos*pstride] + gt_cnt[n_img][3]. */
int odw_discover_iou(const float* s0, const float* s1, const float* s2, int C, const float* boxes,
                     const int* img_off, int n_img, int max_p, const int* pos_cls, const int* n_pos, int maxpos,
                     float thres, int* tops, uint32_t* masks, int* rows, int pstride, int* counts, void* stream);
int odw_discover_sim(const float* E, const float* s0, const float* s1, const float* s2, int C, const float* boxes,
                     const int* img_off, int n_img, int max_p, const int* pos_cls, const int* n_pos, int maxpos,
                     const int* tops, uint32_t* masks, const float* bank, const int* bank_off, const int* bank_cnt,
                     float nms_thr, int pstride, int* inst_idx, int* inst_cnt, int* fresh_idx, int* fresh_cnt,
                     int* gt_idx, int* gt_cls, float* gt_score, int* gt_cnt, void* stream);

/* ---- inference tail of the ROI head ------------------------------------------------------------
 * replaces PostProcessor.forward + filter_results (modeling/roi_heads/box_head/inference.py:41-90,216-258; the
 * weak variant roi_heads/weak_head/inference.py with reg = NULL) reached from ROIWeakRegHead.testing_forward
 * (weak_head.py:124-145): BoxCoder.decode (box_coder.py:52-95, weights wx..wh, dw/dh clamped to xform_clip),
 * BoxList.clip_to_image (bounding_box.py:218-229), per class j >= 1: candidates prob[:, j] > score_thresh,
 * torchvision-semantics NMS (boxlist_ops.py:13-36).  One workgroup per (image, class).
 *   prob (sumP, C) fp32; reg (sumP, ld_reg) fp32 (4*C columns, or 4 with cls_agnostic, or NULL = no decoding);
 *   boxes (sumP, 4) xyxy; img_off (n_img+1) int32; img_wh (n_img, 2) fp32 = (width, height); max_p <= 4096
 *   out_* : (n_img, C-1, pstride[, 4]) survivors of each class in kept order (descending score), out_count
 *   (n_img, C-1).  The final "best max_det over all classes" (kthvalue, ties kept) is the caller's. */
int odw_detect_postprocess(const float* prob, int C, const float* reg, int ld_reg, int cls_agnostic,
                           const float* boxes, const int* img_off, const float* img_wh, int n_img, int max_p,
                           float wx, float wy, float ww, float wh, float xform_clip, float score_thresh,
                           float nms_thr, int pstride, float* out_boxes, float* out_scores, int* out_index,
                           int* out_count, void* stream);
/* The same tail in the two halves test-time augmentation needs (engine/bbox_aug.py:11-77):
 * odw_detect_decode = PostProcessor.forward with bbox_aug_enabled (inference.py:57-90): BoxCoder.decode + clip_to_image
 *   for every (proposal, class), nothing filtered -> out_boxes (sumP, C, 4);
 * odw_detect_filter = PostProcessor.filter_results (inference.py:216-258) on a merged boxlist: boxes_pc (sumP, C, 4) are
 *   used as given (no decode, no clip), prob (sumP, C); outputs as odw_detect_postprocess. */
int odw_detect_decode(const float* reg, int ld_reg, int cls_agnostic, const float* boxes, const int* img_off,
                      const float* img_wh, int n_img, int sum_p, int C, float wx, float wy, float ww, float wh,
                      float xform_clip, float* out_boxes, void* stream);
int odw_detect_filter(const float* prob, int C, const float* boxes_pc, const int* img_off, int n_img, int max_p,
                      float score_thresh, float nms_thr, int pstride, float* out_boxes, float* out_scores,
                      int* out_index, int* out_count, void* stream);

/* ---- backbone convolutions (NHWC bf16, implicit GEMM on the MFMA tile) ----------------------
 * replaces the cuDNN convolutions behind torch.nn.Conv2d in VGG_Base
 * (modeling/backbone/vgg16.py:34-36,58-83: 3x3, stride 1, padding = dilation).
 * odw_conv3x3_nhwc_bf16: Y[m][n] = act( sum_{tap,c} X[m+shift(tap)][c] * Wk[n][tap*C+c] + bias[n] ),
 *   X (n_pix x C) NHWC bf16 with C a power of two >= 8 or any multiple of 64, n_pix = B*H*W; Wk rows zero padded to a
 *   multiple of 64 (ldw); mirror=1 flips the taps (input gradient with the [ci][tap*Cout+co] copy);
 *   mask (bf16, n_pix x ldmask, nullable) zeroes the result where mask == 0 (ReLU backward);
 *   zero_page = >= 16 bytes of zeros in device memory (source of padding taps).
 * odw_conv_weight_prep / odw_conv_wgrad_unpack: torch (Cout,Cin,3,3) fp32 <-> the packed layouts.
 * odw_im2col_t_bf16: (9*C x ldm) pixel-contiguous patches for the weight-gradient GEMM.
 * odw_maxpool2x2_nhwc_bf16(_bwd): 2x2/2 max pool; backward routes to the first maximum and folds
 *   the ReLU mask of the pooled activation.
 * odw_nchw_f32_to_nhwc_bf16 / odw_nhwc_bf16_to_nchw_f32: the two ends of the backbone. */
int odw_conv3x3_nhwc_bf16(const void* X, int n_pix, int H, int W, int C, int dilation, int mirror, const void* Wk,
                          int ldw, int N, void* Y, int ldy, int y_is_bf16, const float* bias, int relu,
                          const void* mask, int ldmask, const void* zero_page, void* stream);
/* Workspace form: for the deep layers (few output tiles, K = 9*C long) the launcher splits K over the grid and
 * reduces fp32 partials with the same fused epilogue.  odw_conv3x3_workspace = bytes wanted (0 = no split). */
int64_t odw_conv3x3_workspace(int n_pix, int C, int N);
/* The same for a known geometry: covers the halo-tile kernel (conv3x3_halo_kernel: a 16x16 spatial tile of pixels x 128
 * output channels per workgroup, the input patch staged in LDS once per 64-channel block and shared by the nine taps),
 * which serves every layer with C % 64 == 0 and dilation 1 or 2 and slices the channel blocks of small maps. */
int64_t odw_conv3x3_workspace_hw(int n_pix, int H, int W, int C, int N, int dilation);
int odw_conv3x3_nhwc_bf16_ws(const void* X, int n_pix, int H, int W, int C, int dilation, int mirror, const void* Wk,
                             int ldw, int N, void* Y, int ldy, int y_is_bf16, const float* bias, int relu,
                             const void* mask, int ldmask, const void* zero_page, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* The forward convolution of the precision mode "bf16x2f" on TWO stored planes (round 5; conv3x3_halo2_kernel): the same
 * cuDNN convolution of modeling/backbone/vgg16.py:34-36,58-83 as the sum of the three bf16 plane products
 * x_hi w_hi + x_hi w_mid + x_mid w_hi, fp32 accumulation.
 *   X  : (n_pix x ldx) bf16, a pixel row = [hi plane: C channels | mid plane: C channels | padding], C % 32 == 0
 *   Wk : (N x ldw) bf16 packed by odw_conv_weight_prep_planes_batch with T = -2: per tap, per block of 32 channels,
 *        [hi 32 | mid 32]; ldw >= 18 C
 *   Y  : y_planes == 0: fp32 (n_pix x ldy);  y_planes == 1: the next layer's operand, bf16 planes [hi N | mid N] per pixel
 *        (ldy bf16 elements per row, the mid plane ldy / 2 elements in) -- bias and ReLU applied before the split;
 *        y_planes == 2: the same planes of the 2 x 2 / 2 max-pooled result (n_pix / 4 rows; H and W even): bias, ReLU, pool and
 *        split all in the epilogue, no fp32 activation (a pooled layer without a backward: the frozen conv1_2 / conv2_2)
 *   N % 64 == 0, dilation 1 or 2; the workspace (odw_conv3x3_planes2_workspace bytes; 0 = none) holds the fp32 partials
 *   of the K slices small maps are cut into. */
int64_t odw_conv3x3_planes2_workspace(int n_pix, int H, int W, int C, int N);
int odw_conv3x3_planes2_ws(const void* X, int ldx, int n_pix, int H, int W, int C, int dilation, const void* Wk, int ldw, int N,
                           void* Y, int ldy, int y_planes, const float* bias, int relu, const void* zero_page, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* 2x2 / 2 max pooling of an fp32 NHWC activation written as that operand (planes [hi C | mid C], row stride ldy) */
int odw_maxpool2x2_nhwc_f32_planes2(const float* X, int B, int H, int W, int C, void* Y, int ldy, void* stream);
int odw_conv_weight_prep(const float* w, int Co, int Ci, int Cp, void* wk, int ldk, void* wd, int ldd, void* stream);
int odw_conv_wgrad_unpack(const float* dwk, int ld, int Co, int Ci, int Cp, float* dw, void* stream);
/* every layer of a body in one launch: host arrays of length n (<= 32) of the arguments of odw_conv_weight_prep */
int odw_conv_weight_prep_batch(int n, const void* const* w, const int* Co, const int* Ci, const int* Cp,
                               void* const* wk, const int* ldk, void* const* wd, const int* ldd, void* stream);
/* the same with wk written as bf16 PLANES of the fp32 weights (the forward operand of the split-precision modes):
 * T[i] (0 = plain bf16, 1..4) blocks of Cp[i] channels per tap, block tt holding plane patterns[4 i + tt]
 * (0 hi, 1 mid, 2 lo, 3 zeros); ldk[i] >= 9 * T[i] * Cp[i]; wd stays single-plane bf16.  T = NULL: plain. */
int odw_conv_weight_prep_planes_batch(int n, const void* const* w, const int* Co, const int* Ci, const int* Cp,
                                      void* const* wk, const int* ldk, void* const* wd, const int* ldd,
                                      const int* T, const int* patterns, void* stream);
/* convolution weight gradient as one call: dw (Co,Ci,3,3 fp32) (+)= dZ^T im2col(X) from the operands
 * odw_linear_bwd_prep (dzt: Co x lda) and odw_im2col_t_bf16 (colt: 9*Cp x ldb) write, K = pixels: split-K partial
 * products into the workspace, then ONE pass that reduces the slices and unpacks [co][tap*Cp+ci] -> [co][ci][tap] */
int64_t odw_conv_wgrad_workspace(int Co, int Cp, int K, int lda, int ldb);
int odw_conv_wgrad_nt(const void* dzt, int lda, const void* colt, int ldb, int Co, int Ci, int Cp, int K, float* dw,
                      int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* the same gradient straight from the NHWC operands (dz: n_pix x ld_dz masked output gradient, X: n_pix x Cp layer input):
 * gemm_tn_bf16_ring_kernel reads both K-major through ds_read_b64_tr_b16 -- no dZ^T, no transposed im2col.
 * Cp a power of two >= 128. */
/* out[n] += sum over the M rows of X (bf16, fp32 sums): the bias gradient beside odw_conv_wgrad_tn */
int odw_colsum_bf16(const void* X, int ld, int M, int N, float* out, void* stream);
/* the same sums in ONE summation order (run-to-run identical bias gradients): row-chunk partials parked in the
 * workspace (odw_colsum_workspace(M, N) bytes of scratch), added in chunk order by a second small launch. */
int64_t odw_colsum_workspace(int M, int N);
int odw_colsum_bf16_ws(const void* X, int ld, int M, int N, float* out, void* workspace, int64_t workspace_bytes,
                       void* stream);
int64_t odw_conv_wgrad_tn_workspace(int Co, int Cp, int n_pix);
int odw_conv_wgrad_tn(const void* dz, int ld_dz, const void* X, int n_pix, int H, int W, int Cp, int dilation, int Co, int Ci,
                      float* dw, int accumulate, const void* zero_page, void* workspace, int64_t workspace_bytes, void* stream);
/* odw_conv_wgrad_tn plus the layer's bias gradient (the .bias half of the same cuDNN backward in the reference,
 * modeling/backbone/vgg16.py:58-83): db[co] += sum over pixels of dz[p][co], one summation order.  Channel counts that are
 * multiples of 64 (dilation 1 / 2) take the output-stationary halo kernel, which forms these sums in extra workgroups of
 * the same launch; otherwise the ring form + the two-launch column sum.  ldx = row stride of X in elements (the halo
 * kernel reads the hi plane of a planes operand in place: ldx = T * Cp; the ring form needs ldx == Cp).
 * Workspace: odw_conv_wgrad_tn_bias_workspace. */
int64_t odw_conv_wgrad_tn_bias_workspace(int Co, int Cp, int n_pix);
int odw_conv_wgrad_tn_bias(const void* dz, int ld_dz, const void* X, int ldx, int n_pix, int H, int W, int Cp, int dilation,
                           int Co, int Ci, float* dw, float* db, int accumulate, const void* zero_page, void* workspace,
                           int64_t workspace_bytes, void* stream);
int odw_im2col_t_bf16(const void* X, int n_pix, int H, int W, int C, int dilation, void* out, int ldm, void* stream);
/* column-block form: this call owns `cols` (>= n_pix, zero padded) columns of a wider (9*C x ldm) matrix */
int odw_im2col_t_bf16_part(const void* X, int n_pix, int H, int W, int C, int dilation, void* out, int ldm, int cols,
                           void* stream);
/* the same pooling / layout helpers on fp32 NHWC activations (bf16x3 precision mode, see odw_split_rows_bf16) */
int odw_maxpool2x2_nhwc_f32(const float* X, int B, int H, int W, int C, float* Y, void* stream);
int odw_maxpool2x2_nhwc_f32_bwd(const float* X, const float* dY, int B, int H, int W, int C, float* dX, void* stream);
/* precision mode "bf16x2f" (split forward, bf16 backward): fp32 pooled activation X, bf16 gradients */
int odw_maxpool2x2_nhwc_f32x_bf16_bwd(const float* X, const void* dY, int B, int H, int W, int C, void* dX, void* stream);
int odw_nchw_f32_to_nhwc_f32(const float* in, int B, int HW, int C, int Cp, float* out, void* stream);
int odw_nhwc_f32_to_nchw_f32(const float* in, int B, int HW, int C, int Cp, float* out, void* stream);
int odw_maxpool2x2_nhwc_bf16(const void* X, int B, int H, int W, int C, void* Y, void* stream);
int odw_maxpool2x2_nhwc_bf16_bwd(const void* X, const void* dY, int B, int H, int W, int C, void* dX, void* stream);
int odw_nhwc_bf16_to_nchw_f32(const void* in, int B, int HW, int C, float* out, void* stream);
int odw_nchw_f32_to_nhwc_bf16(const float* in, int B, int HW, int C, int Cp, void* out, void* stream);

/* ---- dense part of the OD-WSCL loss, forward + backward in two launches ----------------------
 * Y (sumP x ldy) fp32 = the fused predictor output; head_offsets int[8] (HOST) = first column of
 * cls, det, ref1, bbox1, ref2, bbox2, ref3, bbox3 (roi_weak_predictors.py:158-165), C classes.
 * odw_wsddn_scores : final_s = softmax_row(cls) * softmax_col-per-image(det) (loss.py:234-247), src1/src2 =
 *   softmax(ref1/ref2) (loss.py:357); colstat float[n_img][3][128] = det column max, det column sum-exp,
 *   column sums of final_s.
 * odw_refine_losses: given pseudo labels int64[3][sumP], loss weights [3][sumP], regression targets
 *   [3][sumP][4] (od_layer) and image label vectors lab [n_img][C]: out float[n_img][16] = per image
 *   {loss_img, cls0, reg0, cls1, reg1, cls2, reg2, acc_img, acc_ref0..2}/n_img (loss.py:349-409) and
 *   dY (sumP x ldy) = d(sum of the 7 losses)/dY. */
int64_t odw_refine_workspace(int n_img);
int odw_wsddn_scores(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img, int max_p,
                     float* final_s, float* src1, float* src2, float* colstat, void* workspace, int64_t workspace_bytes,
                     void* stream);
int odw_refine_losses(const float* Y, int ldy, const int* head_offsets, int C, const int* img_off, int n_img, int sum_p,
                      int max_p, const float* final_s, const float* colstat, const float* lab, const int64_t* pseudo,
                      const float* weights, const float* targets, const int* n_pos, float eps, float* out, float* dY,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- data boundary: decoded uint8 image -> its slot of the normalised, zero-padded batch -----------
 * Replaces, for the pixels, the reference's CPU-worker transform chain data/transforms/transforms.py:33-150
 * (Resize = torchvision F.resize on a PIL image = Pillow's Image.resize(BILINEAR), :62-70; RandomHorizontalFlip /
 * RandomVerticalFlip :72-98; ToTensor :116-118; Lighting :133-150; Normalize :120-131) and the padding copy of
 * structures/image_list.py:60-72, and engine/bbox_aug.py:81-137 at test time (one call per scale / flip).
 * rgb = DEVICE uint8 (in_h, in_w, 3), RGB order, as PIL decodes it; (out_h, out_w) = Resize.get_size();
 * lighting_rgb / mean / std = HOST float[3] (lighting_rgb may be NULL); out = DEVICE fp32 plane (3, Hp, Wp) of the
 * batch tensor, pixels outside (out_h, out_w) are written as 0.  The resampling is bit-identical to Pillow's 8-bit
 * path (22-bit fixed-point coefficients, horizontal pass rounded to uint8 first). */
int64_t odw_image_preprocess_workspace(int in_h, int in_w, int out_h, int out_w);
int odw_image_preprocess(const uint8_t* rgb, int in_h, int in_w, int out_h, int out_w, int hflip, int vflip,
                         const float* lighting_rgb, const float* mean, const float* std, int to_bgr255, float* out,
                         int Hp, int Wp, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- ResNet-C5 bodies (modeling/backbone/resnet.py:258-406) on NHWC bf16 activations: what is neither a GEMM (the
 * 1x1 convolutions run on odw_gemm_nt_bf16 with the frozen batch-norm folded into weight and bias) nor the 3x3
 * implicit GEMM above.  odw_add_relu_bf16: out = relu(a + b), the residual junction (:368-373), n % 8 == 0;
 * odw_relu_bwd_bf16: g = dout where out > 0.  odw_stem_conv7x7_bn_relu: the 7x7/2 stem on the fp32 NCHW image with
 * weight (Co,3,7,7) fp32, y = relu(conv * scale[co] + shift[co]) -> NHWC bf16 (B, (H+1)/2.., Co) (:381-403);
 * odw_maxpool3x3s2_nhwc_bf16: the 3x3/2 pad-1 max pool that follows it (:404). */
int odw_add_relu_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
int odw_relu_bwd_bf16(const void* dout, const void* out, void* g, int64_t n, void* stream);
int odw_stem_conv7x7_bn_relu(const float* img_nchw, const float* weight, const float* scale, const float* shift, int B,
                             int H, int W, int Co, void* out_nhwc_bf16, void* stream);
int odw_maxpool3x3s2_nhwc_bf16(const void* X, int B, int H, int W, int C, void* Y, void* stream);
/* fp32 NHWC forms of the four (split precision modes keep activations in fp32 between kernels) */
int odw_add_relu_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int odw_relu_bwd_f32(const float* dout, const float* out, float* g, int64_t n, void* stream);
int odw_stem_conv7x7_bn_relu_f32(const float* img_nchw, const float* weight, const float* scale, const float* shift, int B,
                                 int H, int W, int Co, float* out_nhwc, void* stream);
int odw_maxpool3x3s2_nhwc_f32(const float* X, int B, int H, int W, int C, float* Y, void* stream);
/* First layer of the VGG body when it is frozen (modeling/backbone/vgg16.py:58-60, FREEZE_CONV_BODY_AT >= 1): 3x3 / pad 1
 * convolution of the fp32 NCHW image, weight (Co,3,3,3) fp32, + bias + ReLU -> NHWC bf16 (B*H*W, Co), direct form. */
int odw_stem_conv3x3_bias_relu(const float* img_nchw, const float* weight, const float* bias, int B, int H, int W, int Co,
                               void* out_nhwc_bf16, void* stream);
/* the same layer with the result written as bf16 PLANES of the fp32 values (see odw_split_rows_bf16): out_planes
 * (B*H*W x ld) bf16, T column blocks of `block` >= Co channels holding plane pattern[t] -- the operand of the next
 * convolution in the split-precision modes, without an fp32 tensor and a split pass in between */
int odw_stem_conv3x3_bias_relu_planes(const float* img_nchw, const float* weight, const float* bias, int B, int H, int W,
                                      int Co, const int* pattern, int T, void* out_planes, int64_t ld, int block,
                                      void* stream);

/* ---- fp32-grade products on the bf16 matrix cores: operand splitting ("bf16x3") ---------------
 * replaces the fp32 cuBLAS / cuDNN arithmetic behind torch.nn.Linear / Conv2d in the reference
 * (config/defaults.py:559 DTYPE float32; modeling/backbone/vgg16.py:58-83,107-193).
 * An fp32 value is carried as three bf16 planes hi + mid + lo (24 significand bits) laid out along the
 * REDUCTION axis of a product; plane code 0/1/2 = hi/mid/lo, 3 = a block of zeros.
 *   odw_split_rows_bf16: out[r][t*block + c] = plane_{pattern[t]}(in[r][c]), c < Cc <= block (zero padded)
 *   odw_split_cols_bf16: out[c][t*block + r] = plane_{pattern[t]}(in[r][c]), r < R  <= block (zero padded)
 * pattern = T <= 8 plane codes (HOST array).  With pattern {0,0,0,1,1,2} on one operand and {0,1,2,0,1,0} on the
 * other, odw_gemm_nt_bf16 / odw_conv3x3_nhwc_bf16 over K' = T*block accumulate the six plane products of order <= 2.
 *   odw_linear_bwd_mask_f32: dZ = dY * [Y != 0] * scale (Y nullable: dZ = dY), db[n] += sum_m dZ[m][n]
 *   (two fixed-order stages: single writer per bias entry, deterministic; db = NULL skips it and needs no workspace). */
int odw_split_rows_bf16(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                        int64_t ld_out, int block, void* stream);
int odw_split_cols_bf16(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                        int64_t ld_out, int block, void* stream);
int64_t odw_linear_bwd_mask_workspace(int M, int N);     /* bytes the bias gradient's two-stage reduction wants */
int odw_linear_bwd_mask_f32(const float* dY, int64_t ld_dy, const void* Y, int y_is_bf16, int64_t ld_y, int M, int N,
                            float scale, float* dZ, int64_t ld_z, float* db, void* workspace, int64_t workspace_bytes,
                            void* stream);
/* fp32-output forms of the stacked-operand producers (same arithmetic, no bf16 rounding of the result) */
int odw_stack_clean_aug_f32(const float* pooled, const float* block, const float* block_sum, int P, int C, int S,
                            float* out, int ld, void* stream);
int odw_rows_drop_noise_f32(const float* pooled, const int* rows, int row_base, int k, int C, int S, float gamma,
                            uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1, float* keep_sum, float* out, int ld,
                            int out_row0, void* stream);
/* The two views as the OPERAND of the first head Linear in the "bf16x2f" mode: the sampled rows are read from the clean rows'
 * cell-major planes [hi | mid] (k' = cell * C + channel; src_cm, row stride ld_src, mid plane src_mid elements after the hi
 * plane: what odw_roi_pool_stack_forward_nhwc_f32_cm writes) as x = hi + mid, and the 2k view rows (drop rows [out_row0, +k),
 * noise rows [out_row0 + k, +k)) are written as cell-major planes (out_cm, ld_cm, cm_mid: the forward operand of
 * odw_gemm_nt_cm) and as their channel-major hi plane (out_hi, ld_hi: what the single-plane backward transposes).  Same
 * draws, same evaluation order as odw_rows_drop_noise; C a multiple of 64; gradient = odw_rows_drop_noise_bwd. */
int odw_rows_views_cm(const void* src_cm, int64_t ld_src, int64_t src_mid, const int* rows, int row_base, int k, int C, int S,
                      float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1, float* keep_sum, void* out_cm,
                      int64_t ld_cm, int64_t cm_mid, void* out_hi, int64_t ld_hi, int out_row0, void* stream);

/* ---- Device-resident control flow of the OD-WSCL loss (round 6) --------------------------------------------------------
 * replaces the host side of roi_heads/weak_head/loss.py:281-347: its two loops append to Python lists whose lengths depend
 * on the step's scores (pgt_collection, pgt_update, instance_diff).  Rounds 2-5 read the counts back twice per step and
 * assembled the gather lists with numpy; these entry points keep the lists AND their lengths on the device, so the host
 * thread never waits for the GPU inside a step.  Launches that consume a device-resident extent are sized for its
 * CAPACITY (the `_cap` arguments: what exists as memory) and read the live value from a device int (`*_dev`).
 *
 * odw_loss_lists_a: after odw_discover_iou.  grp = host table [G][16] int32 per (image, positive class) group in loop-1
 * order: {img, ci, cls, first proposal row of the image, k6 drop (2), k7 drop (2), k6 noise (2), k7 noise (2), -};
 * cls_order = group indices sorted by (cls, loop-1 order).  Writes scal[16] = {E1, V = 2 E1, r64(V), 3 E1, overflow, ...},
 * the entry prefix e0[G + 1], roi_index[E1] (proposal row of every sampled entry), the class banks (bank_index over the
 * virtual table [P proposal embeddings; V view embeddings], bank_off / bank_cnt per class: loss.py:307, Q2) and the per-row
 * dropout draws of the stacked views in fc6 / fc7 (row_tab6 / row_tab7: uint4 {row inside its pass, key0, key1, -}).
 * odw_loss_lists_b: after odw_discover_sim.  Writes scal[16] = {N, A, E = A + E1, overflow, P64 + r64(V),
 * P64 + r64(V) + r64(A), r64(V), r64(V) + r64(A), r64(A), pseudo-GT overflow, r64(N)}, the SupCon inputs in the reference's
 * orders (features class-major: feat_index into [A_cap re-attached clean rows; views], labels; weights in APPEND order = Q1,
 * = final_score[row, c + 1] / colstat[...]: Q12), the ascending unique proposal rows the features reference (act_rows) and
 * the pooling node's side-buffer entry list roi_index_all = [act_rows | roi_index].  sticky (optional, 2 ints the kernels only ever
 * set): [0] a capacity overflowed in some step, [1] more pseudo-GT boxes than gt_max -- read back at the caller's leisure. */
int odw_loss_lists_a(const int* grp, const int* cls_order, int G, const int* counts, const int* rows, int maxpos, int pstride,
                     int sum_p, int n_cls1, int e_cap, int* scal, int* e0, int* roi_index, int* bank_index, int* bank_off,
                     int* bank_cnt, void* row_tab6, void* row_tab7, void* stream);
int odw_loss_lists_b(const int* grp, const int* cls_order, int G, const int* img_off, const int* n_pos, const int* pos_cls,
                     int n_img, int maxpos, int pstride, int sum_p, const int* scal_a, const int* e0, const int* roi_index,
                     const int* bank_index, const int* bank_off, const int* bank_cnt, const int* fresh_idx, const int* fresh_cnt,
                     const int* gt_cnt, int gt_max, const float* final_score, int fs_cols, const float* colstat, int cs_ld,
                     int cs_off, int n_cap, int a_cap, int e_cap, int p64, int* scal, int* feat_index, int* labels, float* weights,
                     int* act_rows, int* roi_index_all, int* sticky, void* stream);
/* out[r] = index[r] < split ? t0[index[r]] : t1[index[r] - split], r < *n_dev (fp32 rows of D values); and its transpose
 * d0 / d1 += (*scale) * alpha * g (fp32 atomics; scale = a device scalar or NULL = 1). */
int odw_gather_rows2_dyn(const float* t0, const float* t1, int split, const int* index, const int* n_dev, int n_cap, int D,
                         float* out, void* stream);
int odw_scatter_rows2_dyn(const float* g, const int* index, const int* n_dev, int n_cap, int D, int split, const float* scale,
                          float alpha, float* d0, float* d1, void* stream);
/* out[r][0 : row_bytes) = src[index[r]][0 : row_bytes), r < *n_dev; rows [0, *n_dev) of p zeroed (byte counts % 16 == 0). */
int odw_gather_rows_dyn(const void* src, int64_t ld_src_bytes, const int* index, const int* n_dev, int n_cap, int64_t row_bytes,
                        void* out, int64_t ld_out_bytes, void* stream);
int odw_zero_rows_dyn(void* p, int64_t ld_bytes, int64_t row_bytes, const int* n_dev, int n_cap, void* stream);
/* odw_gemm_nt_bf16_ws with device-resident extents: M_cap / K_cap bound the launch, *m_dev rows and a reduction of *k_dev
 * (a multiple of 8; operand columns [*k_dev, r64(*k_dev)) zero) are computed; either may be NULL (= the capacity is exact).
 * row_tab: per-row dropout draw (odw_loss_lists_a).  The hints pick the kernel variant and the K slices only; the workspace
 * query takes M_hint <= 0 for "the rows are not device-resident" (m_dev == NULL: the static form's tail-column split applies). */
int64_t odw_gemm_nt_bf16_dyn_workspace(int M_cap, int M_hint, int N, int K_cap, int K_hint, int lda, int ldb, const void* C, int ldc,
                                       int c_is_bf16, int* variant_out);
int odw_gemm_nt_bf16_dyn(const void* A, int lda, const void* B, int ldb, int M_cap, int N, int K_cap, void* C, int ldc,
                         int c_is_bf16, const float* bias, int relu, float alpha, float drop_p, const void* row_tab,
                         const int* m_dev, int M_hint, const int* k_dev, int K_hint, int accumulate, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* the plain form of odw_gemm_nt_cm (cell-major planes, the first head Linear over the sampled-row views) with *m_dev rows */
int64_t odw_gemm_nt_cm_dyn_workspace(int M_cap, int M_hint, int N, int S);
int odw_gemm_nt_cm_dyn(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M_cap, int N, int C, int S,
                       float* Cout, int ldc, const float* bias, int relu, float drop_p, const void* row_tab, const int* m_dev,
                       int M_hint, void* workspace, int64_t workspace_bytes, void* stream);
/* odw_transpose_to_bf16_part / odw_linear_bwd_prep_part / odw_split_rows_bf16 / odw_l2norm_rows(_bwd) over *r_dev rows;
 * the transposed outputs land *col_off_dev columns into `out` / dZT, zero padded to r64(*r_dev) (the column block of a
 * weight-gradient batch); src_rows / y_rows: the input / mask row of row r is row src_rows[r] of a larger table. */
int odw_transpose_to_bf16_dyn(const void* in, int in_is_f32, int ld_in, int R_cap, int Cc, void* out, int ld_out, const int* r_dev,
                              const int* col_off_dev, const int* src_rows, void* stream);
int odw_linear_bwd_prep_dyn(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M_cap, int N, float scale,
                            void* dZ, int ld_z, void* dZT, int ld_t, float* db, const int* m_dev, const int* tcol_off_dev,
                            const int* y_rows, void* stream);
int odw_split_rows_bf16_dyn(const float* in, int64_t ld_in, int R_cap, int Cc, const int* pattern, int T, void* out, int64_t ld_out,
                            int block, const int* r_dev, void* stream);
int odw_l2norm_rows_dyn(const float* x, int R_cap, int D, float eps, float* y, float* norm, const int* r_dev, void* stream);
int odw_l2norm_rows_bwd_dyn(const float* g, const float* y, const float* norm, int R_cap, int D, float eps, float* dx,
                            const int* r_dev, void* stream);
/* odw_rows_views_cm / odw_rows_drop_noise_bwd_store for EVERY (image, class) group of the step in one launch (loss.py:292-305):
 * group sizes from the entry prefix e0 (odw_loss_lists_a), keys (G x 4: kd0 kd1 kn0 kn1) from the host; entry e's
 * gradient is stored at row *dst_off + e of the side buffer. */
int odw_rows_views_cm_grouped(const void* src_cm, int64_t ld_src, int64_t src_mid, int G, int E_cap, const int* n_entries,
                              const int* e0, const uint32_t* keys, const int* src_row, int C, int S, float gamma, float* keep_sum,
                              void* out_cm, int64_t ld_cm, int64_t cm_mid, void* out_hi, int64_t ld_hi, void* stream);
int odw_rows_views_bwd_store_grouped(const void* dX, int dx_is_f32, int ld, int G, int E_cap, const int* n_entries, const int* e0,
                                     const uint32_t* keys, const float* keep_sum, int C, int S, float gamma, const int* dst_off,
                                     float* extra, void* stream);
/* odw_supcon_v2 over the first *n_dev rows (N_cap sizes the launch and the workspace: odw_supcon_dyn_workspace) */
int64_t odw_supcon_dyn_workspace(int N_cap);
int odw_supcon_v2_dyn(const float* F, const int32_t* labels, const float* w, int N_cap, int D, float tau, float grad_scale,
                      float* loss, float* dF, const int* n_dev, void* workspace, int64_t workspace_bytes, void* stream);
/* odw_roi_pool_stack_backward_ws with *e_dev side-buffer entries (E_cap exist as memory) */
int odw_roi_pool_stack_backward_dyn(const void* dX, int dx_is_f32, int ld, const void* argmax_u16, const float* rois,
                                    const float* keep, const float* keep_sum, const float* extra, const int* extra_roi, int E_cap,
                                    const int* e_dev, int skip_clean, int B, int C, int H, int W, int R, int PH, int PW,
                                    float* grad_in, void* workspace, int64_t workspace_bytes, void* stream);
/* The input gradient of the first Linear and the maximum ROI pooling's backward scales by, from one pass.  The reference's
 * ROIPool backward (csrc/cuda/ROIPool_cuda.cu:80-108) adds floats with atomicAdd; this path adds in fixed point
 * (deterministic), which needs max |dX| first -- a 200 MB read of its own until round 6.
 * odw_gemm_nt_bf16_absmax: C (fp32, N % 4 == 0) = alpha A B^T as odw_gemm_nt_bf16_ws computes it (same plan, same workspace
 * query), and *absmax (a zeroed 4-byte device word) = max(*absmax, bits of max |C|).
 * odw_roi_pool_stack_backward_scaled: odw_roi_pool_stack_backward_ws / _dyn (e_dev NULL: E_cap is exact) with that word already
 * in workspace[0..4): only the side buffer's rows are scanned. */
int odw_gemm_nt_bf16_absmax(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* C, int ldc, float alpha,
                            void* absmax, void* workspace, int64_t workspace_bytes, void* stream);
int odw_roi_pool_stack_backward_scaled(const void* dX, int dx_is_f32, int ld, const void* argmax_u16, const float* rois,
                                       const float* keep, const float* keep_sum, const float* extra, const int* extra_roi, int E_cap,
                                       const int* e_dev, int skip_clean, int B, int C, int H, int W, int R, int PH, int PW,
                                       float* grad_in, void* workspace, int64_t workspace_bytes, void* stream);

/* A stream confined to a subset of the compute units (bit i of mask[i / 32] = CU i usable); the caller destroys it.  No
 * reference counterpart: the reference runs its loss on one stream (weak_head/loss.py:233-411). */
int odw_stream_create_cu_mask(int n_words, const uint32_t* mask, void** stream_out);
int odw_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODWSCL_H_ */
