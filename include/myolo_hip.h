/*
 * myolo_hip.h -- C-ABI of libmyolo_hip.so: the MI355X (gfx950) kernels of the Mask-YOLO
 * forward/backward hot path.
 *
 * The reference (jianing-sun/Mask-YOLO) exposes no FFI: its arithmetic is TensorFlow/Keras
 * op invocations inside myolo/model.py.  Each entry point below replaces the op invocation
 * cited next to it (paths relative to the reference root); INTEGRATION.md shows the ctypes
 * binding a maintainer would add.
 *
 * This header is the OPERATOR API: one entry point per reference op (forward / data gradient / weight gradient), housekeeping
 * and the RCCL wrappers -- what a maintainer replacing a Keras layer binds.  The stage-level entry points the engine
 * (myolo/engine.py) composes its fused pipelines from (Winograd transform / multiply stages, lazy-BatchNorm gradients, row-sparse
 * helpers, producer-fused BatchNorm statistics) live in myolo_hip_internal.h; their buffers have library-owned layouts.
 *
 * Conventions
 *   - every tensor is NHWC, contiguous, float32 unless stated; 2-D views are [rows, channels];
 *   - the CALLER owns every buffer (kernels never allocate); `ws` is caller-provided scratch of
 *     at least myolo_workspace_bytes(...) bytes (forward convolutions accept ws = NULL: they then skip
 *     the split-K path used for small problems);
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no implicit synchronisation,
 *     re-entrant; the only process state is the set of integer tuning switches changed by
 *     myolo_set_option() (all 0 by default) -- nothing in the launch path reads the environment;
 *   - returns 0 on success, a negative MYOLO_E* code otherwise (myolo_last_error_string()).
 */
#ifndef MYOLO_HIP_H
#define MYOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MYOLO_OK            0
#define MYOLO_EINVAL       -1   /* bad argument (shape / alignment / null pointer)  */
#define MYOLO_EWORKSPACE   -2   /* workspace too small                               */
#define MYOLO_ELAUNCH      -3   /* HIP launch error                                  */
#define MYOLO_ECOMM        -4   /* RCCL unavailable or a collective failed           */

#define MYOLO_ACT_NONE   0
#define MYOLO_ACT_RELU   1
#define MYOLO_ACT_RELU6  2

int         myolo_version(void);
const char* myolo_last_error_string(void);
/* upper bound of scratch any single call below needs for a problem with `rows` rows and
 * `cols` output channels (weights-gradient split-K partials dominate). */
size_t      myolo_workspace_bytes(int64_t rows, int cin, int cout);

/* Tuning / ablation switches (process-wide ints, default 0 = shipped behaviour).  Names: "no_nt", "gemm_generic",
 * "no_splitk", "gemm_w256", "wino_nt", "wino_w256", "bf16_regstage", "bf16_no256", "bf16_force256", "crop_bwd_nolds",
 * "tune0", "dw_rows1", "dw_legacy", "dw_bwd_legacy", "dw_min_wg", "wino_no_mixed", "wino_no_bt", "wino_x6", "w63_order", "w63_legacy" (the one-unit Winograd transform kernels of rounds 3-5 instead of the
 * persistent ones; same bits), "w63_wgs" (persistent workgroups per CU, 0 = 1), "deconv_mask_legacy" (myolo_deconv2x2s2_mask_fwd[_keep] under "wino_x6": 1 = the
 * untransposed tile with the per-class butterfly epilogue of rounds 3-5, 2 = the transposed tile with partial logits + the finish launch -- same bits as 0),
 * "bf16_mask_nofin" (myolo_deconv2x2s2_mask_bf16_fwd: partial logits + the finish launch; same bits), "crop_bf16_legacy" (myolo_crop_and_resize_bf16_fwd: four corner
 * loads per output element instead of the column walk; same bits), "pw_no_smallm" (pointwise convs with few rows and K >= 256: the split-K pair of launches of
 * rounds 2-5 instead of the one-launch small-M kernel; another fp32 summation order), "pw_skinny_nw4" (conv_23: four waves per workgroup also from K = 512 up; another
 * fp32 summation order).  Unknown name -> MYOLO_EINVAL.  Every switch selects between
 * kernels with the same contract.  One switch changes NUMERICS within the bf16 inference path: "bf16_mask_valu" = 1 keeps the deconv output and the 1x1 mask kernel of
 * myolo_deconv2x2s2_mask_bf16_fwd in fp32 (rounds 3-5, VALU epilogue); the default rounds both to bf16 like every other activation / weight of that path and runs the
 * 1x1 conv on the matrix pipe (256-row kernels, i.e. >= 1536 row tiles of 256 channels; smaller problems keep the fp32 form).
 * One semantic switch: "bn_fused_tf_variance" (default 1) -- the BatchNormalization moving-variance update of bn_stats
 * restates Keras 2.2.x on TensorFlow 1.x's fused path (tf.nn.fused_batch_norm hands Keras the Bessel-corrected batch
 * variance, Keras multiplies by n/(n-(1+eps)) on top); 0 = Keras' factor on the biased variance (non-fused backend). */
int myolo_set_option(const char* name, int value);
int myolo_get_option(const char* name, int* value);

/* ---- gradient exchange of the data-parallel step (the reference has none: GPU_COUNT = 0,
 *      config.py:47).  Thin RCCL wrappers, librccl.so resolved lazily (MYOLO_ECOMM if it cannot be loaded).
 *      rank 0 creates the 128-byte id and ships it to the other ranks by any side channel; every rank then calls
 *      comm_init on its own HIP device; allreduce is in place, asynchronous on `stream`. ---- */
#define MYOLO_COMM_ID_BYTES 128
int myolo_comm_unique_id(void* id_out_128_bytes);
int myolo_comm_init(int rank, int nranks, const void* unique_id_128_bytes, void** comm_out);
int myolo_comm_size(void* comm, int* nranks_out);
int myolo_allreduce_sum_f32(float* buf, int64_t n, void* comm, void* stream);
int myolo_comm_destroy(void* comm);

/* ---- conv1: ZeroPad(1,1) + Conv2D 3x3 stride 2 valid, Cin=3, no bias -- model.py:45-50 ---- */
int myolo_conv3x3s2_c3_fwd(const float* x, const float* w, float* y,
                           int N, int H, int W, int Cout, void* stream);
int myolo_conv3x3s2_c3_bwd_weight(const float* x, const float* dy, float* dw,
                                  int N, int H, int W, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- DepthwiseConv2D 3x3, stride 1 'same' | stride 2 pad-bottom-right 'valid', no bias
 *      (keras_applications _depthwise_conv_block, model.py:19,68-77,256-268) ---- */
int myolo_dwconv3x3_fwd(const float* x, const float* w, float* y,
                        int N, int H, int W, int C, int stride, void* stream);
/* inference: act(dwconv(x) * scale + shift) in one launch: the frozen BatchNorm after the depthwise conv folded into its epilogue
 * (scale / shift from myolo_bn_frozen_coeffs[_batched]); bit-identical to myolo_dwconv3x3_fwd + myolo_bn_apply_act
 * (keras_applications mobilenet _depthwise_conv_block with BatchNormalization in inference mode) */
int myolo_dwconv3x3_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y,
                                   int N, int H, int W, int C, int stride, void* stream);
int myolo_dwconv3x3_bwd_data(const float* dy, const float* w, float* dx,
                             int N, int H, int W, int C, int stride, void* stream);
size_t myolo_dwconv3x3_bwd_weight_ws_bytes(int N, int H, int W, int C, int stride);   /* scratch every path of the call below accepts */
int myolo_dwconv3x3_bwd_weight(const float* x, const float* dy, float* dw,
                               int N, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream);

/* ---- pointwise Conv2D 1x1 (conv_pw_N, conv_23 model.py:271) : y[M,Cout] = x[M,Cin] w[Cin,Cout] (+bias) ---- */
/* fwd: ws may be NULL (ws_bytes 0); with scratch of >= 8*M*Cout*4 bytes the small deep layers run split-K (same result up to summation order) */
int myolo_pwconv1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                        int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* inference: act((x w) * scale + shift) in one launch (the affine in the GEMM / split-K epilogue), bit-identical to
 * myolo_pwconv1x1_fwd + myolo_bn_apply_act; act: MYOLO_ACT_NONE | RELU | RELU6 */
int myolo_pwconv1x1_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y,
                                   int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_pwconv1x1_bwd_data(const float* dy, const float* w, float* dx,
                             int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_pwconv1x1_bwd_weight(const float* x, const float* dy, float* dw,
                               int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- Conv2D 3x3 'same' + bias (feature_map model.py:848; myolo_mask_conv1-4 model.py:688-709) ---- */
int myolo_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y,
                      int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* same convolution with a per-channel affine (a folded frozen BatchNormalization: scale/shift from
 * myolo_bn_frozen_coeffs) and optional ReLU applied in the epilogue: y = act((conv + bias)*scale + shift)
 * -- TimeDistributed(Conv2D) + BatchNormalization(training=False) + ReLU, model.py:693-709 */
int myolo_conv3x3_affine_act_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                                 float* y, int N, int H, int W, int Cin, int Cout, int act,
                                 void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_bwd_data(const float* dy, const float* w, float* dx,
                           int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_bwd_weight(const float* x, const float* dy, float* dw,
                             int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- Conv2DTranspose 2x2 stride 2 + bias (+ReLU) (myolo_mask_deconv model.py:711-712);
 *      kernel layout [2,2,Cout,Cin] ---- */
int myolo_deconv2x2s2_fwd(const float* x, const float* w, const float* bias, float* y,
                          int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, void* stream);
int myolo_deconv2x2s2_bwd_data(const float* dy, const float* w, float* dx,
                               int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_deconv2x2s2_bwd_weight(const float* x, const float* dy, float* dw,
                                 int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- per-channel column sums over rows: bias gradients ---- */
int myolo_colsum(const float* x, float* out, int64_t M, int C, void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNormalization(axis=-1, eps=1e-3, momentum=0.99) (model.py:51,690-708) ----
 * bn_stats: batch mean / biased variance of x[M,C]; writes mean,var and the fused affine
 *   scale = gamma*rsqrt(var+eps), shift = beta - mean*scale; if moving_* non-null updates them
 *   Keras-style (momentum 0.99, variance rescaled by M/(M-(1+eps)), on top of TF's fused-path M/(M-1): see
 *   myolo_set_option "bn_fused_tf_variance").
 * bn_frozen_coeffs: same scale/shift from the moving statistics (inference / training=False).
 * bn_apply_act: y = act(x*scale + shift).
 * bn_act_bwd: dy is the gradient wrt act(...) output; recomputes the activation mask from x;
 *   batch_stats=1 -> training-mode BN backward (needs mean, var); 0 -> frozen affine.
 *   Writes dx, dgamma, dbeta. */
int myolo_bn_stats(const float* x, const float* gamma, const float* beta,
                   float* mean, float* var, float* scale, float* shift,
                   float* moving_mean, float* moving_var,
                   int64_t M, int C, void* ws, size_t ws_bytes, void* stream);
int myolo_bn_frozen_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, float* scale, float* shift, int C, void* stream);
int myolo_bn_apply_act(const float* x, const float* scale, const float* shift, float* y,
                       int64_t M, int C, int act, void* stream);
/* frozen BatchNorm + activation in one launch: myolo_bn_frozen_coeffs followed by myolo_bn_apply_act (same results; scale / shift are
 * written as well); needs C/4 to divide 256 */
int myolo_bn_frozen_apply_act(const float* x, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                              float* scale, float* shift, float* y, int64_t M, int C, int act, void* stream);
int myolo_bn_act_bwd(const float* dy, const float* x, const float* gamma,
                     const float* mean, const float* var, const float* scale, const float* shift,
                     float* dx, float* dgamma, float* dbeta,
                     int64_t M, int C, int act, int batch_stats, void* ws, size_t ws_bytes, void* stream);



/* ---- tf.image.crop_and_resize, bilinear, extrapolation 0 (PyramidROIAlign model.py:385-387) ----
 * boxes [nb,4] = (y1,x1,y2,x2) normalised as crop_and_resize reads them; box_ind [nb] int32. */
int myolo_crop_and_resize_fwd(const float* image, const float* boxes, const int32_t* box_ind, float* out,
                              int B, int H, int W, int C, int nb, int crop_h, int crop_w, void* stream);
int myolo_crop_and_resize_bwd_image(const float* dout, const float* boxes, const int32_t* box_ind, float* dimage,
                                    int B, int H, int W, int C, int nb, int crop_h, int crop_w, void* stream);

/* Same gradient for the layout build_mask_graph uses (model.py:684-685: boxes [batch, R, 4] flattened,
 * box b*R+r samples image b): gather formulation, deterministic, no atomics, writes every element. */
int myolo_roialign_bwd_grouped(const float* dout, const float* boxes, float* dimage,
                               int B, int H, int W, int C, int R, int crop_h, int crop_w, void* stream);

/* ---- DecodeYOLOLayer model.py:1442-1473 / DetectionsLayer model.py:1493-1538 ----
 * y_pred [B,G,G,A,5+C]; anchors [2A]; proposals [B,G*G*A,4]; detections [B,G*G*A,6]. */
int myolo_yolo_decode(const float* y_pred, const float* anchors, float* proposals,
                      int B, int G, int A, int C, void* stream);
int myolo_yolo_detections(const float* y_pred, const float* anchors, float* detections,
                          int B, int G, int A, int C, void* stream);

/* ---- yolo_custom_loss model.py:86-242, forward + gradient wrt y_pred in one call ----
 * y_true [B,G,G,A,5+C] f32; true_boxes [B,T,4] f32 (grid units cx,cy,w,h); class_weights [C];
 * out_terms[8] = {loss, loss_xy, loss_wh, loss_conf, loss_class, recall, n_coord, n_conf};
 * grad = d(loss*loss_weight)/d y_pred. */
int myolo_yolo_loss(const float* y_true, const float* y_pred, const float* true_boxes,
                    const float* anchors, const float* class_weights,
                    float object_scale, float no_object_scale, float coord_scale, float class_scale,
                    float loss_weight, float* out_terms, float* grad,
                    int B, int G, int A, int C, int T, void* ws, size_t ws_bytes, void* stream);
/* the same with the warm-up branch of model.py:193-207 selectable (warmup != 0: taken by the reference while its `seen` counter, incremented once per
 * evaluation of the loss, is below config.WARM_UP_BATCHES): predictors without a box are pulled to their cell centre and anchor size, every
 * predictor's coordinate terms count with weight 1.  myolo_yolo_loss == warmup 0. */
int myolo_yolo_loss_warmup(const float* y_true, const float* y_pred, const float* true_boxes,
                           const float* anchors, const float* class_weights,
                           float object_scale, float no_object_scale, float coord_scale, float class_scale,
                           float loss_weight, int warmup, float* out_terms, float* grad,
                           int B, int G, int A, int C, int T, void* ws, size_t ws_bytes, void* stream);

/* ---- DetectMaskTargetLayer / detect_mask_target_graph model.py:457-661 (+norm_boxes_graph
 *      :1394-1408, trim_zeros_graph :1411-1420, overlaps_graph :420-454), one block per image ----
 * proposals [B,R,4] (x1,y1,x2,y2); gt_class_ids [B,T] i32; gt_boxes_px [B,T,4] i32 (x1,y1,x2,y2);
 * gt_masks [B,H,W,T] uint8 (0/1).  Outputs: rois [B,R,4], target_class_ids [B,R] i32,
 * target_masks [B,R,mh,mw] f32 in {0,1}, n_pos [B] i32. */
int myolo_mask_targets(const float* proposals, const int32_t* gt_class_ids, const int32_t* gt_boxes_px,
                       const uint8_t* gt_masks, float* rois, int32_t* target_class_ids, float* target_masks,
                       int32_t* n_pos, int B, int R, int T, int H, int W, int mh, int mw, void* stream);

/* ---- Shapes input producer: from per-image shape specifications to the six training inputs
 *      of model.py:896-897 on the device -- ShapesDataset.load_image / load_mask / draw_shape
 *      (example/shapes/dataset_shapes.py:80-135), load_image_gt's empty-instance filter + extract_bboxes
 *      (myolo_utils.py:247-271,346-352) and BatchGenerator.__getitem__'s target encoding (myolo_utils.py:753-844).
 *      spec [B, spec_stride] int32: bg r,g,b, n_shapes, then 13 ints per shape (type 1/2/3 = class id, r,g,b, x,y,s,
 *      triangle vertices ax,ay,bx,by,cx,cy).  lut[256] = float32(v / 255.).  anchors double [2A]. ---- */
int myolo_shapes_batch(const int32_t* spec, int spec_stride, const double* anchors, const float* lut,
                       float* images, uint8_t* gt_masks, int32_t* gt_boxes, int32_t* gt_class_ids,
                       float* y_true, float* true_boxes,
                       int B, int H, int W, int S, int T, int G, int A, int C,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- inference post-processing: unmold_mask for all detections of one image (myolo_utils.py:883-912 as
 *      called from MaskYOLO.decode_masks model.py:1355-1389).  masks [N,mh,mw,C] post-sigmoid, detections [N,6]
 *      (x1,y1,x2,y2,score,class) normalised; full_masks [H,W,N] uint8 0/1 (class channel picked, order-1 resize to
 *      the clamped pixel window, threshold 0.5, paste).  The resize is skimage.transform.resize(order=1, mode='constant',
 *      cval=0, clip=True, anti_aliasing=False) as the reference's wrapper calls it (myolo_utils.py:433-447): zero border,
 *      output clipped to the mask's value range.  ws: N * 4 bytes. ---- */
int myolo_unmold_masks(const float* masks, const float* detections, uint8_t* full_masks,
                       int N, int mh, int mw, int C, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* ---- final mask conv 1x1 + bias + sigmoid (myolo_mask model.py:713-714), C small ---- */
int myolo_mask_head_out_fwd(const float* x, const float* w, const float* bias, float* p,
                            int64_t M, int Cin, int C, void* stream);
/* backward of the above given dz [M,C] (gradient wrt the pre-sigmoid logits):
 * dx[M,Cin] = (dz w^T) * (x > 0)   (ReLU of the deconv output folded in), dw[Cin,C], db[C]. */
int myolo_mask_head_out_bwd(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db,
                            int64_t M, int Cin, int C, void* ws, size_t ws_bytes, void* stream);

/* ---- myolo_mask_loss_graph model.py:718-754 (K.binary_crossentropy), forward + gradient ----
 * target_masks [NR,h,w]; ids [NR] i32; pred [NR,h,w,C] post-sigmoid.
 * loss_out[2] = {loss, n_positive_rois}; dz [NR,h,w,C] = d(loss*loss_weight)/d logits. */
int myolo_mask_bce(const float* target_masks, const int32_t* target_class_ids, const float* pred,
                   float loss_weight, float* loss_out, float* dz,
                   int NR, int h, int w, int C, void* ws, size_t ws_bytes, void* stream);

/* ---- Keras Adam (model.py:1071-1075) over a flat parameter buffer ----
 * g is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int myolo_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                    float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* ---- myolo_mask_deconv + ReLU + myolo_mask 1x1 + sigmoid (model.py:711-714) in one pass: the [N,2H,2W,Cout]
 * activation is never written.  w [2,2,Cout,Cin], w2 [Cout][ncls], p_out [N,2H,2W,ncls] probabilities.
 * Needs Cout % 128 == 0, Cin % 16 == 0, ncls <= 4; results equal deconv2x2s2_fwd + mask_head_out_fwd up to fp32
 * summation order (partial sums over 64-channel slabs are added in a fixed order: bit-reproducible). ---- */
size_t myolo_deconv2x2s2_mask_ws_bytes(int N, int H, int W, int Cin, int Cout, int ncls);
int myolo_deconv2x2s2_mask_fwd(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                               int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream);


/* ---- Winograd F(4x4,3x3) form of the dense 3x3/s1/SAME convolutions (same contracts as myolo_conv3x3_* above:
 * myolo_mask_conv1-4 model.py:687-709 and their gradients); fp32 operands and accumulation, 4x fewer multiplications
 * per full tile.  ws_bytes(which): 0 forward, 1 data gradient, 2 weight gradient.
 * fwd: optional scale/shift = folded frozen BatchNorm applied after the bias (as conv3x3_affine_act_fwd); v_keep
 * (nullable) receives the transformed input [36][N*ceil(H/4)*ceil(W/4)][Cin] so bwd_weight can reuse it (v_saved). ---- */
size_t myolo_conv3x3_wino_ws_bytes(int N, int H, int W, int Cin, int Cout, int which);
int myolo_conv3x3_wino_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y,
                           int N, int H, int W, int Cin, int Cout, int act, float* v_keep, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                                void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino_bwd_weight(const float* x, const float* v_saved, const float* dy, float* dw, int N, int H, int W,
                                  int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- F(6,3)/F(4,3) tiling for 14x14 maps (csrc/wino63_kernels.hip): 14 = 6+4+4 per direction, 400 point-tiles per image in 64 planes
 * (the F(4,3) tiles use the 36 planes of the points they share with F(6,3)).  Stages as above; V / M buffers hold
 * myolo_wino63_plane_elems floats, U myolo_wino63_u_elems (opaque).  Needs myolo_wino63_ok(H, W, Cin, Cout):
 * H = W = 14, channels multiples of 64, Cin % 16 == 0, Cout % 256 == 0.  The reference op: the conv2-4 / bn / ReLU stages of
 * build_mask_graph (model.py:693-709). ---- */
int    myolo_wino63_ok(int H, int W, int Cin, int Cout);
/* the three conv operators as single calls, mirroring myolo_conv3x3_wino_{fwd,bwd_data,bwd_weight} (which = 0, 1, 2 for the scratch size) */
size_t myolo_conv3x3_wino63_ws_bytes(int N, int Cin, int Cout, int which);
int myolo_conv3x3_wino63_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int N,
                             int Cin, int Cout, int act, float* v_keep, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino63_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int myolo_conv3x3_wino63_bwd_weight(const float* x, const float* v_saved, const float* dy, float* dw, int N, int Cin, int Cout, void* ws,
                                    size_t ws_bytes, void* stream);

/* ---- bf16 inference path of the mask head (BASELINE.json configs[3]: Rice 416x416, bf16, inference-only) ----
 * Activations are bf16 (uint16_t bit patterns, NHWC), accumulation fp32 on v_mfma_f32_32x32x16_bf16.  All four
 * BatchNorm layers of build_mask_graph are frozen in inference (model.py:690-708), so they are folded into the
 * packed weights.  pack_weights: w fp32 [K][N] (w_is_nk=0; HWIO kernel flattened) or [N][K] (w_is_nk=1; the
 * Conv2DTranspose kernel [2,2,Co,Ci] flattened) -> wt bf16 [N][K]; gamma==NULL packs without folding. */
int myolo_pack_weights_bf16(const float* w, int K, int N, int w_is_nk, const float* bias, const float* gamma, const float* beta,
                            const float* mean, const float* var, uint16_t* wt, float* bias_out, void* stream);
/* tf.image.crop_and_resize (model.py:385-387) with bf16 output; same sampling arithmetic as myolo_crop_and_resize_fwd */
int myolo_crop_and_resize_bf16_fwd(const float* image, const float* boxes, const int32_t* box_ind, uint16_t* out,
                                   int B, int H, int W, int C, int nb, int crop_h, int crop_w, void* stream);
/* myolo_mask_conv1-4 + folded bn + relu (model.py:687-709): x [N,H,W,Cin] bf16, wt [Cout][9*Cin] bf16 */
int myolo_conv3x3_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, uint16_t* y,
                           int N, int H, int W, int Cin, int Cout, int act, void* stream);
/* myolo_mask_deconv (model.py:711-712): wt [4*Cout][Cin] bf16, y [N,2H,2W,Cout] bf16 */
int myolo_deconv2x2s2_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, uint16_t* y,
                               int N, int H, int W, int Cin, int Cout, int act, void* stream);
/* deconv + ReLU + myolo_mask 1x1 + sigmoid in one pass from bf16 activations (see myolo_deconv2x2s2_mask_fwd); ws holds
 * (Cout/128)*2 partial-logit slabs of 4*N*H*W*ncls floats */
int myolo_deconv2x2s2_mask_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, const float* w2, const float* b2, float* p_out,
                                    int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream);
/* myolo_mask 1x1 + sigmoid (model.py:713-714) from bf16 activations; fp32 weights and probabilities */
int myolo_mask_head_out_bf16_fwd(const uint16_t* x, const float* w, const float* bias, float* p,
                                 int64_t M, int Cin, int C, void* stream);

/* ---- plain fp32 matrix product with the way its products are formed chosen per call (csrc/wino_mm.hip) ----
 * C [M][N] = A [M][K] * B, B given as [N][K] (b_is_nk = 1) or [K][N] (b_is_nk = 0); K % 16 == 0, N % 256 == 0, 16-byte aligned.
 * MYOLO_PRODUCTS_NATIVE: v_mfma_f32_32x32x2_f32.  MYOLO_PRODUCTS_BF16X6: every fp32 operand split EXACTLY into three bf16
 * pieces, each fp32 product accumulated in fp32 from its six piece products >= 2^-24 relative (cfg.FP32_MATMUL, DESIGN.md
 * section 8): fp32-level error; an Inf operand gives NaN (Inf - Inf in the split) where the native product gives +-Inf. */
#define MYOLO_PRODUCTS_NATIVE 0
#define MYOLO_PRODUCTS_BF16X6 1
size_t myolo_matmul_f32_ws_bytes(int K, int N, int b_is_nk, int products);
int myolo_matmul_f32(const float* A, const float* B, float* C, int64_t M, int K, int N, int b_is_nk, int products,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- small elementwise helpers ---- */
int myolo_add_inplace(float* a, const float* b, int64_t n, void* stream);      /* a += b */
int myolo_fill(float* a, float value, int64_t n, void* stream);
/* y[i] = (float)(x[i] / 255.0): the generator's `image / 255.` (myolo_utils.py:824) done on the device, so that a training batch crosses
 * PCIe as bytes (a quarter of the float32 image) -- same bits as the host expression. */
int myolo_u8_to_unit_f32(const uint8_t* x, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MYOLO_HIP_H */
