/*
 * myolo_hip_internal.h -- stage-level entry points of libmyolo_hip.so used by the engine (mask-yolo_amd/myolo/engine.py) to
 * compose its fused pipelines: the Winograd transform / multiply stages, conv gradients that form a BatchNorm gradient lazily,
 * row-sparse helpers of the exact-sparsity backward, batched frozen-BatchNorm coefficients.  Same C-ABI conventions as
 * myolo_hip.h (caller-owned buffers, explicit stream, status codes), but the V / M / U / Q buffers exchanged between these
 * calls have LIBRARY-OWNED layouts ("opaque"): they are only meaningful to the matching stage of the same library build.
 * A maintainer replacing a Keras layer binds the operators of myolo_hip.h, not these.
 */
#ifndef MYOLO_HIP_INTERNAL_H
#define MYOLO_HIP_INTERNAL_H

#include "myolo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- frozen BatchNorm coefficients of many layers in one launch (inference trunk) ---- */
/* the same coefficients for many layers in one launch: table [nlayers][6] int64 (device) = element offsets of gamma, beta into params,
 * of the moving mean, moving variance into stats, of the layer's output into coeffs (scale[C] then shift[C]), and C */
int myolo_bn_frozen_coeffs_batched(const float* params, const float* stats, const int64_t* table, int nlayers, float* coeffs,
                                   void* stream);

/* frozen BatchNormalization (training=False, model.py:696,702,708) + ReLU/ReLU6 backward from the POST-activation tensor
 * a = act(gamma*xhat + beta): where the activation passes gradient xhat = (a - beta)/gamma, so the pre-BN tensor is not needed.
 * dx = scale * dy * [act passes], dgamma = sum dz*xhat, dbeta = sum dz.  ws as bn_act_bwd. */
int myolo_bn_act_bwd_frozen_post(const float* dy, const float* a_post, const float* gamma, const float* beta, const float* scale,
                                 float* dx, float* dgamma, float* dbeta, int64_t M, int C, int act,
                                 void* ws, size_t ws_bytes, void* stream);

/* ---- training-mode BatchNorm (+ activation) backward in ONE launch (same results as myolo_bn_act_bwd(batch_stats = 1) up to the order in which the row
 * slabs' partial sums are added; bit-reproducible): sums, a grid-wide barrier, dx.  sync: 64 int32 counters, zero before their first use, never reset by
 * the caller, and used by launches of ONE stream only (two launches in flight on one counter would wait for each other's arrivals); ws:
 * myolo_bn_act_bwd_fused_ws_bytes (0 = shape not supported: use myolo_bn_act_bwd).  The grid is at most 256 workgroups, all of which must become
 * resident: do not call from a stream whose progress another stream's running kernel waits for. ---- */
size_t myolo_bn_act_bwd_fused_ws_bytes(int64_t M, int C);
int myolo_bn_act_bwd_fused(const float* dy, const float* x, const float* mean, const float* var, const float* scale, const float* shift, float* dx,
                           float* dgamma, float* dbeta, int64_t M, int C, int act, int32_t* sync, void* ws, size_t ws_bytes, void* stream);

/* ---- prepared-weights registry: the weight-only re-layouts that entry points run in front of their kernels (transposes, bf16x6 piece splits, Winograd
 * filter transforms) recorded once and re-run by the owner on a stream of its choice, off the training step's critical chain (csrc/myolo_common.h).
 * arena: a 256-byte aligned device buffer the slots are carved from (entries that do not fit stay on the in-place path).  activate(h) makes h the
 * registry the entry points consult (NULL: none); refresh re-runs entries [first, last) on `stream` for the current weight generation and returns
 * how many it launched; invalidate: the weights changed.  One launch thread per process. ---- */
int myolo_wprep_create(void* arena, size_t arena_bytes, void** handle);
int myolo_wprep_destroy(void* h);
int myolo_wprep_activate(void* h);
int myolo_wprep_count(void* h);
int myolo_wprep_invalidate(void* h);
int myolo_wprep_refresh(void* h, int first, int last, int max_idle, void* stream);
int myolo_wprep_stats(void* h, long long* hits, long long* misses, long long* bytes_used);
int myolo_wprep_overflows(void* h, long long* n);      /* resolve() calls whose new site found no room in the arena (made in place every step, never recorded) */

/* Exact-sparsity helpers for the mask head backward (build_mask_graph model.py:690-708: bn1 is the only
 * batch-statistics layer, so behind it only ROIs with a positive target carry non-zero gradient):
 * gather_groups: dst[i] = src[idx[i]] for groups of group_elems floats (one ROI's rows);
 * bn_act_bwd_rowsparse: training-mode BN+activation backward where the upstream gradient is given only
 *   for the n_groups row groups idx[] (compact dy [n_groups*group_rows, C]; inv[group] = slot or -1);
 *   results are identical to bn_act_bwd on the zero-padded dense gradient. */
int myolo_gather_groups(const float* src, const int32_t* idx, float* dst, int n, int64_t group_elems, void* stream);
/* the index those helpers take, built on the device from the per-image positive counts of myolo_mask_targets (an image's positives are its first
 * n_pos[b] ROIs, model.py:593): flags [B*R] (NULL: not written) = 1 for a positive ROI, inv [B*R] = its compact slot or -1, idx [slot] = flat ROI
 * (the first sum(n_pos) entries of a [B*R] buffer are written), total [1] (NULL: not written) = sum(n_pos).  No host round trip. */
int myolo_positive_index(const int32_t* n_pos, int B, int R, int32_t* flags, int32_t* idx, int32_t* inv, int32_t* total, void* stream);
/* myolo_deconv2x2s2_mask_fwd (include/myolo_hip.h) that ALSO writes relu(deconv + bias) -- exactly what myolo_deconv2x2s2_fwd(ACT_RELU) gives -- for the images (ROIs) the training step will
 * differentiate: keep_inv [N] = slot (0 <= slot < keep_cap) or -1 (myolo_positive_index), keep_d [keep_cap][2H][2W][Cout]; images whose slot is -1 or
 * >= keep_cap are not kept.  Needs Cout % 256 == 0 (the matrix-pipe kernels of csrc/wino_mm.hip). */
int myolo_deconv2x2s2_mask_fwd_keep(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                                    int N, int H, int W, int Cin, int Cout, int ncls, const int32_t* keep_inv, float* keep_d, int keep_cap,
                                    void* ws, size_t ws_bytes, void* stream);
/* the same gather of n groups of group_rows x C floats, fused with the per-channel affine map + activation that follows it in the compacted mask-head
 * backward: dst_pre (NULL: not written) = the gathered rows, dst_act = act(row * scale + shift) as myolo_bn_apply_act gives.  C / 4 must divide 256. */
int myolo_gather_groups_affine_act(const float* src, const int32_t* idx, const float* scale, const float* shift, int act, float* dst_pre,
                                   float* dst_act, int n, int64_t group_rows, int C, void* stream);
int myolo_bn_act_bwd_rowsparse(const float* dy_compact, const float* x, const int32_t* idx, const int32_t* inv,
                               const float* mean, const float* var, const float* scale, const float* shift,
                               float* dx, float* dgamma, float* dbeta,
                               int64_t M, int C, int n_groups, int group_rows, int act,
                               void* ws, size_t ws_bytes, void* stream);

/* ---- Winograd F(4,3)/F(2,3) tiling: buffer sizes and the forward's stages (csrc/wino_kernels.hip) ---- */
/* elements of the 36 transformed planes V (or M) of an [N,H,W,C] tensor (<= 36*N*ceil(H/4)*ceil(W/4)*C: with mixed tiling --
 * F(2,3) on a last tile row / column that holds <= 2 outputs, e.g. 14 = 4+4+4+2 -- reduced tiles have no row in the planes of
 * the points they do not use; at 14x14 that is 484 instead of 576 point-tiles per image) */
size_t myolo_wino_plane_elems(int N, int H, int W, int C);
/* floats to allocate for U of myolo_wino_weight_transform (1.5 x 36*Cin*Cout: room for the split-bf16 layout of option "wino_x6") */
size_t myolo_wino_u_elems(int Cin, int Cout);

/* the forward's four stages, callable on their own: U = 36 planes of Cin*Cout transformed filter taps (flip=1: of the rotated
 * filter with the channel roles exchanged, for the data gradient: call the multiply with (Cout, Cin) then).  U is OPAQUE between
 * myolo_wino_weight_transform and myolo_wino_multiply: the element order inside a plane is the one the multiply kernel chosen
 * for (Cin, Cout) wants ([K][N], or [N][K] for csrc/wino_mm.hip when K % 16 == 0 and N % 256 == 0).  V and M = 36 planes of myolo_wino_plane_elems(N,H,W,C) elements in total (a buffer of 36*T*C floats,
 * T = N*ceil(H/4)*ceil(W/4), always suffices); planes are ordered by point group, see csrc/wino_kernels.hip */
int myolo_wino_weight_transform(const float* w, float* U, int Cin, int Cout, int flip, void* stream);
int myolo_wino_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream);
int myolo_wino_multiply(const float* V, const float* U, float* M, int N, int H, int W, int Cin, int Cout, void* stream);
/* weight_transform (flip 0) + multiply; U_scratch is written only when the filters are not already prepared for this step (myolo_wprep_*) */
int myolo_wino_multiply_w(const float* V, const float* w, float* U_scratch, float* M, int N, int H, int W, int Cin, int Cout, void* stream);
int myolo_wino_output_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                int N, int H, int W, int C, int act, void* stream);
/* ROIAlign (myolo_crop_and_resize_fwd: same boxes / box_ind / sampling) fused into the input transform of the conv that
 * consumes the crops: V [36][nb*ceil(crop_h/4)*ceil(crop_w/4)][C] directly from the feature map [B,FH,FW,C] */
int myolo_wino_input_transform_roialign(const float* feature, const float* boxes, const int32_t* box_ind, float* V, int B, int FH, int FW,
                                        int C, int nb, int crop_h, int crop_w, void* stream);
/* input transform with the producing layer's BatchNorm apply + activation folded into the load (scale/shift per channel) */
int myolo_wino_input_transform_affine(const float* x, const float* scale, const float* shift, int act, float* V, int N, int H, int W,
                                      int C, void* stream);
/* output transform (+bias) that also produces the training-mode BatchNorm statistics of what it writes: same outputs as
 * myolo_bn_stats (mean, var, folded scale/shift, moving averages; model.py:690).  C/4 must divide 256. */
size_t myolo_wino_output_transform_bn_ws_bytes(int C);
int myolo_wino_output_transform_bn_stats(const float* M, const float* bias, float* y, int N, int H, int W, int C, const float* gamma,
                                         const float* beta, float* mean, float* var, float* scale, float* shift, float* moving_mean,
                                         float* moving_var, void* ws, size_t ws_bytes, void* stream);
/* conv gradients whose incoming gradient sits behind a training-mode BatchNorm + activation with a row-sparse upstream
 * gradient (bn1 of the mask head, model.py:690 -- only the positive ROIs carry gradient, myolo_mask_loss_graph
 * model.py:739-746): bn_bwd_rowsparse_coeffs reduces dgamma / dbeta and leaves the per-channel terms ka, kb of
 * dx = scale*dz + ka + kb*x; the *_lazybn gradients form dx while loading the pre-BN tensor, so it is never written. */
int myolo_bn_bwd_rowsparse_coeffs(const float* dy_compact, const float* x, const int32_t* idx, const float* mean, const float* var,
                                  const float* scale, const float* shift, float* dgamma, float* dbeta, float* ka, float* kb, int64_t M,
                                  int C, int n_groups, int group_rows, int act, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino_bwd_data_lazybn(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale,
                                       const float* shift, const float* ka, const float* kb, int act, const float* w, float* dx, int N,
                                       int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino_bwd_weight_lazybn(const float* v_saved, const float* y_pre, const float* dy_compact, const int32_t* inv,
                                         const float* scale, const float* shift, const float* ka, const float* kb, int act, float* dw,
                                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* layer boundary between two Winograd convs in one pass per image: M of conv_i -> (+bias, affine, act) -> V of conv_{i+1}
 * through LDS; the activation itself goes to y only for images with flags[img] != 0 (flags NULL: all; y NULL: none).
 * Needs C % 32 == 0 and ceil(H/4)*ceil(W/4) <= 32 (14x14: 16 tiles). */
int myolo_wino_output_input_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                      const int32_t* flags, float* V_next, int N, int H, int W, int C, int act, void* stream);
/* ... writing the conv's PRE-BatchNorm output (A^T m A + bias) to ypre where flags[img] != 0 (NULL: everywhere) instead of the activation */
int myolo_wino_output_input_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* ypre,
                                               const int32_t* flags, float* V_next, int N, int H, int W, int C, int act, void* stream);

/* ---- trunk backward: BatchNorm-backward sums from the producer of the BatchNorm's output gradient (csrc/mem_kernels.hip) ---- */
/* The same data gradient when dx is the gradient reaching a TRAINING-mode BatchNorm + activation (the conv_pw_{b-1}_bn / conv1_bn in front of this
 * depthwise conv, model.py:51,68-77): xbn = that BatchNorm's pre-BN tensor [N,H,W,C], scale / shift / mean / var = its forward coefficients.  The
 * kernel also leaves the BatchNorm backward's sums (sum dz, sum dz * xhat) as `rows` rows of [2][C] double partials in `part`;
 * myolo_bn_act_bwd_from_partials finishes them (fixed order) and forms the BatchNorm's dx -- the pass over (dx, xbn) myolo_bn_act_bwd starts with
 * is gone.  rows = myolo_dwconv3x3_bwd_data_bnsums_rows(...); 0 = not available for these sizes (use the two plain calls). */
int myolo_dwconv3x3_bwd_data_bnsums_rows(int N, int H, int W, int C, int stride);
int myolo_dwconv3x3_bwd_data_bnsums(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int stride, const float* xbn,
                                    const float* scale, const float* shift, const float* mean, const float* var, int act, double* part, int rows,
                                    void* stream);
/* myolo_bn_act_bwd(batch_stats = 1) from sums a producer already left as nblk rows of [2][C] double partials (myolo_dwconv3x3_bwd_data_bnsums):
 * finish in row order -> dgamma, dbeta; then dx.  ws: 2 * C doubles. */
int myolo_bn_act_bwd_from_partials(const float* dy, const float* x, const float* mean, const float* var, const float* scale, const float* shift,
                                   float* dx, float* dgamma, float* dbeta, int64_t M, int C, int act, const double* part, int nblk,
                                   void* ws, size_t ws_bytes, void* stream);

/* ---- F(6,3)/F(4,3) tiling of 14x14 maps: buffer sizes and stages (csrc/wino63_kernels.hip) ---- */
size_t myolo_wino63_plane_elems(int N, int C);
size_t myolo_wino63_u_elems(int Cin, int Cout);

int myolo_wino63_weight_transform(const float* w, float* U, int Cin, int Cout, void* stream);
int myolo_wino63_multiply(const float* V, const float* U, float* M, int N, int Cin, int Cout, void* stream);
/* weight_transform + multiply in one call (w = the layer's [3,3,Cin,Cout] kernel): U_scratch (myolo_wino63_u_elems floats) is written only when the
 * transformed filters are not already prepared for this step (prepared-weights registry, myolo_wprep_* above) */
int myolo_wino63_multiply_w(const float* V, const float* w, float* U_scratch, float* M, int N, int Cin, int Cout, void* stream);
/* x [N,14,14,C] -> act(x*scale + shift) (scale NULL: identity) -> V; the activation also goes to y (NULL: nowhere) where flags[img] != 0
 * (flags NULL: everywhere) */
int myolo_wino63_input_transform(const float* x, const float* scale, const float* shift, int act, float* y, const int32_t* flags, float* V,
                                 int N, int C, void* stream);
/* the flagged outputs of the three keeping calls (input_transform's y, *_keep_pre's ypre) written in COMPACT order for the sparse mask-head backward:
 * slots [N] = compact slot of an image or -1 (myolo_positive_index); image n's rows go to block slots[n] when 0 <= slots[n] < cap -- no gather later */
int myolo_wino63_input_transform_slots(const float* x, const float* scale, const float* shift, int act, float* y_compact, const int32_t* slots, int cap,
                                       float* V, int N, int C, void* stream);
int myolo_wino63_output_input_transform_keep_pre_slots(const float* M, const float* bias, const float* scale, const float* shift, float* ypre_compact,
                                                       const int32_t* slots, int cap, float* Vn, int N, int C, int act, void* stream);
int myolo_wino63_output_transform_keep_pre_slots(const float* M, const float* bias, const float* scale, const float* shift, float* y, float* ypre_compact,
                                                 const int32_t* slots, int cap, int N, int C, int act, void* stream);
/* layer boundary in one kernel: M_i -> act((A^T m A + bias)*scale + shift) -> V_{i+1}; y / flags as above */
int myolo_wino63_output_input_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                        const int32_t* flags, float* Vn, int N, int C, int act, void* stream);
int myolo_wino63_output_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y, int N, int C, int act,
                                  void* stream);
/* ... with the conv's PRE-BatchNorm output (A^T m A + bias) kept for the flagged ROIs (ypre written where flags[img] != 0, NULL: everywhere; y of
 * output_transform_keep_pre -- the activation -- written for every ROI, may be NULL): the exact-sparsity backward reads bn2-4's backward off it */
int myolo_wino63_output_input_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* ypre,
                                                 const int32_t* flags, float* Vn, int N, int C, int act, void* stream);
int myolo_wino63_output_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* y, float* ypre,
                                           const int32_t* flags, int N, int C, int act, void* stream);
/* conv1 of the mask head on this tiling: ROIAlign fused into the input transform (myolo_wino_input_transform_roialign), the output
 * transform that also yields the training-mode BatchNorm statistics (myolo_wino_output_transform_bn_stats), and the weight gradient
 * from the kept V planes and the lazily formed output gradient (myolo_conv3x3_wino_bwd_weight_lazybn) */
int myolo_wino63_input_transform_roialign(const float* feature, const float* boxes, const int32_t* box_ind, float* V, int B, int FH, int FW,
                                          int C, int nb, void* stream);
size_t myolo_wino63_output_transform_bn_ws_bytes(int N, int C);
int myolo_wino63_output_transform_bn_stats(const float* M, const float* bias, float* y, int N, int C, const float* gamma, const float* beta,
                                           float* mean, float* var, float* scale, float* shift, float* moving_mean, float* moving_var,
                                           void* ws, size_t ws_bytes, void* stream);
size_t myolo_wino63_bwd_weight_ws_bytes(int N, int Cin, int Cout);
int myolo_wino63_bwd_weight_lazybn(const float* v_saved, const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale,
                                   const float* shift, const float* ka, const float* kb, int act, float* dw, int N, int Cin, int Cout,
                                   void* ws, size_t ws_bytes, void* stream);
/* conv1's backward with ONE pass over y_pre: the lazily formed gradient of the conv's output is transformed both ways in one kernel --
 * V (operand of the data gradient) and Q (operand of the weight gradient), myolo_wino63_plane_elems(N, C) floats each -- and the two
 * gradients are finished by the calls below (on different streams if the caller likes: they share nothing but read-only inputs).
 * Same results as myolo_wino63_bwd_data_lazybn + myolo_wino63_bwd_weight_lazybn, bit for bit. */
int myolo_wino63_lazybn_transforms(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale, const float* shift,
                                   const float* ka, const float* kb, int act, float* V, float* Q, int N, int C, void* stream);
size_t myolo_wino63_bwd_data_from_v_ws_bytes(int N, int Cin, int Cout);
int myolo_wino63_bwd_data_from_v(const float* V, const float* w, float* dx, int N, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
size_t myolo_wino63_bwd_weight_from_q_ws_bytes(int N, int Cin, int Cout);
int myolo_wino63_bwd_weight_from_q(const float* v_saved, const float* Q, float* dw, int N, int Cin, int Cout, void* ws, size_t ws_bytes,
                                   void* stream);

/* myolo_conv3x3_wino_bwd_data_lazybn on this tiling (same operands; needs myolo_wino63_ok(14, 14, Cout, Cin)) */
size_t myolo_wino63_bwd_data_ws_bytes(int N, int Cin, int Cout);
int myolo_wino63_bwd_data_lazybn(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale, const float* shift,
                                 const float* ka, const float* kb, int act, const float* w, float* dx, int N, int Cin, int Cout, void* ws,
                                 size_t ws_bytes, void* stream);
/* ---- training-mode BatchNorm fusion of the trunk (model.py:42-79, 249-278 with every BatchNormalization on batch statistics):
 * "*_bnstats_fwd" = the conv AND the batch statistics of its output (mean, biased variance, folded scale / shift, Keras moving averages:
 * exactly what myolo_bn_stats produces) -- the statistics are reduced from partial sums the conv kernel leaves in its epilogue, so the
 * output is not re-read; "in_scale / in_shift / in_act" = the PRODUCING layer's BatchNorm apply + activation, performed on the load of
 * its pre-BN output, so the normalised activation is never written (NULL: the input is used as it is).  "phases": 3 = both launches
 * (the conv, then the statistics finish); 1 / 2 = only the first / second, for callers that bracket the conv kernel with events.  The "*_bwd_weight_affine_in"
 * gradients re-normalise the same pre-BN tensor on load.  Results equal the unfused sequences up to fp32 summation order. ---- */
/* inference: act(conv(x) * scale + shift) in one launch: the frozen conv1_bn + ReLU6 folded into the conv's store (scale / shift from
 * myolo_bn_frozen_coeffs[_batched]); bit-identical to myolo_conv3x3s2_c3_fwd + myolo_bn_apply_act (model.py:45-52, BatchNormalization in inference mode) */
int myolo_conv3x3s2_c3_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y,
                                      int N, int H, int W, int Cout, void* stream);
size_t myolo_conv3x3s2_c3_bnstats_ws_bytes(int N, int H, int W, int Cout);
int myolo_conv3x3s2_c3_bnstats_fwd(const float* x, const float* w, float* y, const float* gamma, const float* beta, float* mean, float* var,
                                   float* scale, float* shift, float* moving_mean, float* moving_var, int N, int H, int W, int Cout, int phases,
                                   void* ws, size_t ws_bytes, void* stream);
size_t myolo_dwconv3x3_bnstats_ws_bytes(int N, int H, int W, int C, int stride);
int myolo_dwconv3x3_bnstats_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y,
                                const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                                float* moving_mean, float* moving_var, int N, int H, int W, int C, int stride, int phases,
                                void* ws, size_t ws_bytes, void* stream);
int myolo_dwconv3x3_bwd_weight_affine_in(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                                         int N, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream);
int    myolo_pwconv1x1_bnstats_ok(int Cin, int Cout);
size_t myolo_pwconv1x1_bnstats_ws_bytes(int64_t M, int Cin, int Cout);
int myolo_pwconv1x1_bnstats_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y,
                                const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                                float* moving_mean, float* moving_var, int64_t M, int Cin, int Cout, int phases, void* ws, size_t ws_bytes, void* stream);
int myolo_pwconv1x1_bwd_weight_affine_in(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                                         int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- HBM stream-copy microbenchmark (the measured copy bandwidth printed beside the nominal 8 TB/s): dst = src over
 * nbytes (multiple of 16), hand-written float4 kernel, 4 loads in flight per thread.  variant 0 default cache policy, 1 non-temporal
 * stores, 2 non-temporal loads + stores, 3 read only, 4 write only; blocks <= 0: 8 workgroups per CU. ---- */
int myolo_stream_copy(const void* src, void* dst, size_t nbytes, int variant, int blocks, void* stream);
/* Matrix-pipe ceiling probe: `blocks` workgroups of four waves (<= 0: 512 = two per CU), each wave `iters` rounds of eight independent
 * accumulator blocks from register operands.  kind 0: v_mfma_f32_32x32x16_bf16 (32768 flop each), kind 1: v_mfma_f32_32x32x2_f32 (4096 flop each),
 * kind 2 / 3: bf16 with the eight issues of a round on two alternating / one accumulator block (dependent-issue chains).
 * flop per launch = blocks * 4 * iters * 8 * flop-per-instruction.  `out` (blocks * 256 floats) is practically never written. */
int myolo_mfma_probe(int kind, int iters, int blocks, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MYOLO_HIP_INTERNAL_H */
