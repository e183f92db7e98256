/*
 * occnet_amd.h — C ABI of the MI355X (gfx950) OccNet / BEVFormer-occ forward hot path.
 *
 * Every entry point takes plain device pointers + sizes (no torch types) and a
 * hipStream_t passed as void*.  All tensors are device-resident, contiguous, row-major.
 * Work is enqueued on `stream`; no call synchronises the device.  Every function returns
 * 0 on success and a negative OCC_E_* code on failure; occ_last_error() returns a
 * thread-local human readable message (the Python mirror turns it into RuntimeError, the
 * analogue of the TORCH_CHECK exceptions the reference's extension raises).
 *
 * Reference interfaces replaced (paths relative to the OccNet tree,
 *   P/ = projects/mmdet3d_plugin/):
 *   occ_ms_deform_attn_forward_f32   <- mmcv._ext.ms_deform_attn_forward, bound at
 *        P/bevformer/modules/multi_scale_deformable_attn_function.py:10-12, called :42-48,:118-124
 *   occ_ms_deform_attn_backward_f32  <- mmcv._ext.ms_deform_attn_backward, same file :74-84,:150-160
 *   occ_point_sampling_f32           <- BEVFormerEncoder.point_sampling, P/bevformer/modules/encoder.py:92-151
 *   occ_sca_fused_forward_f32        <- SpatialCrossAttention.forward :136-173 (rebatch, scatter-add,
 *        visible-camera mean) fused with MSDeformableAttention3D.forward :338-396 (softmax,
 *        offset normalisation, z-anchor add, deformable gather), P/bevformer/modules/spatial_cross_attention.py
 *   occ_tsa_fused_forward_f32        <- TemporalSelfAttention.forward :206-262 (softmax, locations,
 *        gather, mean over the 2-deep BEV queue), P/bevformer/modules/temporal_self_attention.py
 *   occ_conv3d_pack_weight_f32, occ_conv3d_bn_relu_f32 <- TransformerOcc.forward lifter view +
 *        decoder ConvModule(Conv3d k3 + BN3d + ReLU) x2 + permute, P/bevformer/modules/transformer_occ.py
 *        :305-308 (modules :106-126)
 *   occ_occ_heads_f32                <- predicter / flow_predicter MLPs, transformer_occ.py:132-141,318-319
 *   occ_linear_f32                   <- the nn.Linear / FFN / LayerNorm call sites of a BEVFormerLayer
 *   occ_dvr_render_forward_f32       <- dvr.render_forward, tools/ray_iou/lib/dvr/dvr.cu:70-388
 */
#ifndef OCCNET_AMD_H_
#define OCCNET_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCC_OK 0
#define OCC_E_INVALID (-1)     /* bad argument / unsupported shape                */
#define OCC_E_LAUNCH (-2)      /* hipLaunch / runtime error                       */
#define OCC_E_UNSUPPORTED (-3) /* shape is valid but has no fused kernel: caller  */
                               /* must use the unfused HIP path (never a CPU one) */

/* ABI version: bumped whenever a signature below changes. */
int occ_abi_version(void);
/* Thread-local message describing the last failure in this thread ("" if none). */
const char* occ_last_error(void);
/* Test hook: q[i] = the quotient a[i] / d[i] as the gather kernels compute it (reciprocal + Newton step + residual correction,
 * csrc/common.h occ::fdiv — NOT the compiler's IEEE division expansion, whose results are unreliable on gfx950 while an
 * MFMA-issuing wave shares the SIMD: DESIGN.md section 8d).  a, d, q: n DEVICE floats; d normal and non-zero. */
int occ_selftest_fdiv_f32(const float* a, const float* d, float* q, int64_t n, void* stream);

/* ==========================================================================================
 * PART I — THE REFERENCE INTERFACE.  The entry points a maintainer of the reference binds: one per interface the
 * reference itself loads from native code (mmcv._ext for the model, the dvr extension for the metric).  INTEGRATION.md
 * section 3 shows the binding stubs; nothing else in this header is needed for a drop-in.
 *   occ_ms_deform_attn_forward_f32, occ_ms_deform_attn_backward_f32 (+ _workspace_bytes / _ws_f32), occ_dvr_render_forward_f32
 * ========================================================================================== */

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward (mmcv op semantics).
 *   value            (B, S, M, D)        f32   S = sum_l H_l*W_l
 *   spatial_shapes   (L, 2)              i64   (H_l, W_l), device memory
 *   level_start_index(L)                 i64   device memory
 *   sampling_loc     (B, Lq, M, L, P, 2) f32   (x, y) in [0,1] image-normalised
 *   attn_weight      (B, Lq, M, L, P)    f32
 *   out              (B, Lq, M*D)        f32   fully overwritten
 * out[b,q,m,:] = sum_l sum_p attn[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc*(W,H)-0.5),
 * zero padding, align_corners=False.  im2col_step keeps mmcv's contract:
 * step = min(B, im2col_step) must divide B (otherwise OCC_E_INVALID).
 */
int occ_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes,
                                   const int64_t* level_start_index, const float* sampling_loc,
                                   const float* attn_weight, float* out, int B, int S, int M, int D,
                                   int L, int Lq, int P, int im2col_step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, backward (mmcv op semantics).  Shapes as in the forward;
 *   grad_output (B, Lq, M*D) f32 in;  grad_value (B, S, M, D), grad_sampling_loc (B, Lq, M, L, P, 2),
 *   grad_attn_weight (B, Lq, M, L, P) f32 out — PRE-ZEROED BY THE CALLER (the reference's autograd
 *   Function allocates them with zeros_like, multi_scale_deformable_attn_function.py:146-148);
 *   grad_value: for D == 32 (almost) WITHOUT floating-point atomics (counting sort of the bilinear row items into
 *   32-pixel bins, replayed by owner blocks from a device-built work list, csrc/msda_backward.hip): a bin with
 *   more than 2048 items is split over several blocks that combine with f32 atomics (a few dozen per bin); the
 *   item order inside a bin follows integer-atomic slot order.  The f32 summation order — and the last bits of
 *   grad_value — may therefore differ from run to run, as they do with mmcv's atomicAdd.  Other D (and
 *   OCC_MSDA_BWD_ATOMICS=1): f32 atomics throughout.
 */
int occ_ms_deform_attn_backward_f32(const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start_index, const float* sampling_loc,
                                    const float* attn_weight, const float* grad_output,
                                    float* grad_value, float* grad_sampling_loc,
                                    float* grad_attn_weight, int B, int S, int M, int D, int L,
                                    int Lq, int P, int im2col_step, void* stream);
/* The same with CALLER-PROVIDED scratch for the atomic-free grad_value path (D == 32): `workspace` of at least
 * occ_ms_deform_attn_backward_workspace_bytes(...) bytes, 256-byte aligned, uninitialised (1.0-1.2 GB per SCA call
 * at the base config: the Python operator module takes it from torch's caching allocator).  workspace == NULL: the
 * library uses hipMallocAsync and, if that fails, falls back to float atomics with ONE warning on stderr.
 * ..._workspace_bytes returns 0 when the path does not apply (D != 32 / index ranges): workspace is then ignored. */
int64_t occ_ms_deform_attn_backward_workspace_bytes(int B, int S, int M, int D, int L, int Lq, int P);
int occ_ms_deform_attn_backward_ws_f32(const float* value, const int64_t* spatial_shapes,
                                       const int64_t* level_start_index, const float* sampling_loc,
                                       const float* attn_weight, const float* grad_output,
                                       float* grad_value, float* grad_sampling_loc,
                                       float* grad_attn_weight, int B, int S, int M, int D, int L,
                                       int Lq, int P, int im2col_step, void* workspace,
                                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray casting through an occupancy grid (RayIoU metric) — replaces the reference's `dvr.render_forward`
 * (tools/ray_iou/lib/dvr/dvr.cpp:68-72 binding, dvr.cu:70-388), called at
 * projects/mmdet3d_plugin/datasets/ray_metrics.py:116-123.
 *   sigma (N, T, Z, Y, X) f32 occupancy ; origin (N, T, 3) f32 and points (N, M, point_stride >= 3) f32
 *   in VOXEL units ; tindex (N, M) f32 time index per ray (< 0 = padded ray, skipped)
 *   pred_dist, gt_dist (N, M) f32 and coord_index (N, M, 3) f32 out, fully written
 *   (-1, -1, (0,0,0) for rays that never enter the grid) ; train_phase 0 = "test", 1 = "train".
 */
int occ_dvr_render_forward_f32(const float* sigma, const float* origin, const float* points,
                               const float* tindex, float* pred_dist, float* gt_dist,
                               float* coord_index, int N, int T, int Z, int Y, int X, int M,
                               int point_stride, int train_phase, void* stream);

/* ==========================================================================================
 * PART II — FUSED MI355X ENTRY POINTS.  No counterpart in mmcv._ext: each replaces a stretch of the reference's
 * Python / ATen code on the hot path with a gfx950 kernel (cited per entry); occnet_amd/plugin/ calls them behind the
 * reference's module / registry surface.  Grouped: (a) the encoder — projection of the pillars, the two fused gathers,
 * the Linear kernels and chains; (b) the decoder — Conv3d, heads, decode; (c) training partners; (d) the inference
 * plan of the stock image backbone (outside the SURVEY.md section 8 scope; builder conveniences such as
 * occ_mfma_pack_b_frag_bf16 / occ_bias_act_nhwc_bf16 live here).
 * ========================================================================================== */

/* ------------------------------------------------------------------------------------------
 * Pillar reference points -> per-camera image coordinates + visibility.
 *   ref_3d     (B, Z, Nq, 3) f32  normalised (x, y, z) in [0,1]
 *   lidar2img  (B, NC, 4, 4) f32 ; ego2lidar (4, 4) f32 ; pc_range[6] host floats
 *   ref_cam    (NC, B, Nq, Z, 2) f32 out ; bev_mask (NC, B, Nq, Z) u8 out (0/1)
 *   vis_bits   (B, Nq) u32 out: bit c set iff any z-anchor of query q is visible in camera c
 *              (may be NULL).  NC <= 32.
 */
int occ_point_sampling_f32(const float* ref_3d, const float* lidar2img, const float* ego2lidar,
                           const float* pc_range, float img_h, float img_w, float* ref_cam,
                           uint8_t* bev_mask, uint32_t* vis_bits, int B, int NC, int Nq, int Z,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused spatial cross-attention gather (one encoder layer, all cameras):
 *   slots[b,q,:] = (1/max(1,count[b,q])) * sum_{c : query q visible in camera c (batch 0's mask)}
 *                  MSDA( value[b*NC+c], softmax(logits[b,q]), ref_cam[c,b,q,z(p)] + offs[b,q]/(W_l,H_l) )
 *   value     (B*NC, S, M, D) f32 ; spatial_shapes (L,2) i64 ; level_start_index (L) i64
 *   offs      (B, Nq, M*L*P*2) f32 raw sampling_offsets Linear output, row stride offs_stride floats
 *   logits    (B, Nq, M*L*P)   f32 raw attention_weights Linear output, row stride logits_stride
 *   ref_cam   (NC, B, Nq, Z, 2) f32 ; vis_bits (B, Nq) u32 (see occ_point_sampling_f32)
 *   order     (Nq) i32 processing order of the queries (permutation; NULL = identity); only
 *             affects cache locality, never results
 *   slots     (B, Nq, M*D) f32 out, fully overwritten
 *   stats     NULL, or 2 x u64 device counters (pre-zeroed): [0] += visible (camera,query) rows,
 *             [1] += bilinear corners that fall inside their map (the N_in of the roofline formula)
 * Fused kernels exist for M=8, D=32, (L,P) in {(4,8),(4,4),(2,8),(1,8)}, Z | P; other shapes
 * return OCC_E_UNSUPPORTED.
 */
int occ_sca_fused_forward_f32(const float* value, const int64_t* spatial_shapes,
                              const int64_t* level_start_index, const float* offs,
                              int64_t offs_stride, const float* logits, int64_t logits_stride,
                              const float* ref_cam, const uint32_t* vis_bits, const int32_t* order,
                              float* slots, uint64_t* stats, int B, int NC, int S, int M, int D,
                              int L, int P, int Z, int Nq, void* stream);

/* The same gather over fp16 VALUE maps (SURVEY.md §8d's e_v = 2 variant), as written by occ_value_proj_bf16_f16pairs /
 * occ_value_proj_bf16_planes.  One head row of a pixel is 64 bytes = 4 lanes x 16 bytes, so a wave load fetches 16 rows
 * instead of 8 (the texture path retires wave loads, not bytes: csrc/sca_fused.hip).  LAYOUT: pixel PAIRS —
 * value_f16[b * NC + c][pix >> 1][head (M)][pix & 1][D] fp16, pix = level_start + y * W + x, so the two x-neighbours
 * (2k, 2k + 1) of one head share one 128-byte line; S = pixel rows per camera entry INCLUDING padding to an even count.
 * Sampling arithmetic, attention weights and accumulation stay fp32 (v_fma_mix_f32); the value elements carry 11
 * significant bits.  value_scale: NULL, or a DEVICE float s (a power of two): the maps hold s * value (what the value
 * projections write under out_scale = occ_value_range_scale_bf16's result, so that no finite feature map can pass the
 * fp16 limit); the kernel divides its fp32 sums by count * s — exact, the result does not depend on s.  (abi 2) */
int occ_sca_fused_forward_f16v(const void* value_f16, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* offs, int64_t offs_stride,
                               const float* logits, int64_t logits_stride, const float* ref_cam,
                               const uint32_t* vis_bits, const int32_t* order, float* slots, uint64_t* stats,
                               int B, int NC, int S, int M, int D, int L, int P, int Z, int Nq,
                               const float* value_scale, void* stream);

/* The same gather over q16 VALUE maps (round 6): the fp16 maps' geometry, pair layout and byte count, but BLOCK FLOATING
 * POINT — every 16-byte piece (8 channels of one head of one pixel) holds 8 two's-complement int16 mantissas q_j under one
 * 4-bit exponent E:  value_j * s = q_j * 2^(E - 15), E = the binary exponent of the piece's largest |value * s| (0 .. 15),
 * stored in the two low bits of elements 0 and 1 (E & 3, E >> 2; those two are rounded to the nearest value with these
 * low bits, i.e. to 14 bits).  The reference keeps these rows in fp32 (spatial_cross_attention.py:75,387-390); fp16 rows
 * round every element to 2^-12 relative, q16 rounds the elements that dominate the gather's sums to 2^-16 relative —
 * measured against the CPU oracle: DESIGN.md section 2.  Written by occ_value_proj_bf16_planes(out_f16 = 2),
 * occ_value_proj_bf16_q16pairs or occ_sca_rows_encode_q16.  value_scale: the plane's range scale s (or NULL = 1). */
int occ_sca_fused_forward_q16v(const void* value_q16, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* offs, int64_t offs_stride,
                               const float* logits, int64_t logits_stride, const float* ref_cam,
                               const uint32_t* vis_bits, const int32_t* order, float* slots, uint64_t* stats,
                               int B, int NC, int S, int M, int D, int L, int P, int Z, int Nq,
                               const float* value_scale, void* stream);
/* fp32 value rows -> q16 pixel pairs.  v (groups, S, C) f32 contiguous, C % 32 == 0; out (groups, S + (S & 1), C) int16 in
 * the pair order (the pad row of an odd S is not written); scale: NULL or one DEVICE float s, a power of two with
 * max|v| * s <= 2^15. */
int occ_sca_rows_encode_q16(const float* v, void* out, const float* scale, int64_t groups, int S, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused temporal self-attention gather over the 2-deep BEV queue (single level):
 *   out[b,q,:] = 0.5 * sum_{t in {0,1}} MSDA( value[b*2+t], softmax_p(logits[b,q,m,t,:]),
 *                                            ref_2d[b*2+t,q] + offs[b,q,m,t,p]/(W,H) )
 *   value    f32, queue entry t of batch b at value + (b*2+t)*value_bt_stride floats, each (Nq_v, M, D)
 *            with Nq_v = bev_h*bev_w (pass value_bt_stride such that both entries alias one buffer
 *            when there is no history BEV: the reference stacks the same tensor twice)
 *   offs     (B, Nq, M*2*P*2) f32, row stride offs_stride ; logits (B, Nq, M*2*P), stride logits_stride
 *   ref_2d   (B*2, Nq, 1, 2) f32
 *   out      (B, Nq, M*D) f32
 * Fused kernel exists for M=8, D=32, P=4, one level; otherwise OCC_E_UNSUPPORTED.
 */
int occ_tsa_fused_forward_f32(const float* value, int64_t value_bt_stride, const float* offs,
                              int64_t offs_stride, const float* logits, int64_t logits_stride,
                              const float* ref_2d, const int32_t* order, float* out, int B, int Nq,
                              int bev_h, int bev_w, int M, int D, int P, void* stream);

/* ------------------------------------------------------------------------------------------
 * Lifter + Conv3d(k=3, pad=1, stride=1, no bias) + BatchNorm3d(eval) + ReLU, implicit GEMM on the f32
 * matrix cores (exact f32).  Voxel grid (Z, Y, X) = torch's (D, H, W).
 *   in        in_layout 0: (B, Y, X, Z, Cin) f32, channels innermost (the layout this op writes)
 *             in_layout 1: (B, Y, X, Cin, Z) f32 = the lifter view of a (B, Y*X, Cin*Z) BEV embedding,
 *                          channel c = ci*Z + z (transformer_occ.py:305-307)
 *   w_packed  weight packed by occ_conv3d_pack_weight_f32 from torch's (Cout, Cin, 3, 3, 3) layout
 *             (Cout*Cin*27 floats)
 *   scale, shift (Cout) f32: y = conv*scale + shift  (BN eval: scale = gamma/sqrt(var+eps),
 *             shift = beta - mean*scale; a conv bias folds into shift)
 *   out       element (b, y, x, z, co) at b*out_stride_b + y*out_stride_y + x*out_stride_x + z*Cout + co
 *             (strides in floats): (Y,X)-major for the next conv, (X,Y)-major for the reference's
 *             permute(0,4,3,2,1) output order (transformer_occ.py:308)
 * Kernels exist for Cout = 32, Cin % 8 == 0, Z in {4, 8, 16, 32}; otherwise OCC_E_UNSUPPORTED.
 * occ_conv3d_channel_block(Cin) = input channels contracted per LDS phase (16, 8, or 0 = unsupported).
 */
int occ_conv3d_channel_block(int Cin);
int occ_conv3d_pack_weight_f32(const float* weight, float* packed, int Cin, int Cout, void* stream);
int occ_conv3d_bn_relu_f32(const float* in, const float* w_packed, const float* scale,
                           const float* shift, float* out, int B, int Z, int Y, int X, int Cin,
                           int Cout, int in_layout, int64_t out_stride_b, int64_t out_stride_y,
                           int64_t out_stride_x, int relu, void* stream);

/* bf16x3 variant of the two calls above (the default decoder kernel): operands split into hi + lo bf16,
 * a.w ~= al.wh + ah.wl + ah.wh accumulated in f32 by v_mfma_f32_32x32x16_bf16 (product error <= 2^-16; 5.3x less
 * matrix-pipe time than the exact-f32 instruction).  Same arguments; packed = 2 * Cout*Cin*27 16-bit words
 * ([phase][tap][hi, lo][k half][co][8]); needs Cout == 32, Cin % 16 == 0, Z in {4, 8, 16, 32}.
 * Cin == 8 (BASELINE configs[4]: 256 / 32 channels per voxel; round 4): two TAPS per 16-k MFMA step — lanes 0-31
 * contract tap 2s, lanes 32-63 tap 2s + 1, 14 steps; packed = 14 * 2 * 2 * 32 * 8 16-bit words
 * ([step][hi, lo][tap parity][co][8], the 28th tap zero).
 */
int occ_conv3d_pack_weight_bf16x3(const float* weight, void* packed, int Cin, int Cout, void* stream);
int occ_conv3d_bn_relu_bf16x3_f32(const float* in, const void* w_packed, const float* scale, const float* shift,
                                  float* out, int B, int Z, int Y, int X, int Cin, int Cout, int in_layout,
                                  int64_t out_stride_b, int64_t out_stride_y, int64_t out_stride_x, int relu,
                                  void* stream);

/* Second decoder convolution + BatchNorm(eval) + ReLU + BOTH occupancy heads + the class decode in ONE kernel
 * (csrc/conv3d_mfma.hip conv3d_heads_x3_kernel; reference transformer_occ.py:304-321, bevformer_occ_head.py:210-212):
 * the (B, Y, X, Z, 32) activations of the second convolution never reach HBM.  in (B, Y, X, Z, Cin = 32) f32 (layout 0 of
 * occ_conv3d_bn_relu_*), w_packed = occ_conv3d_pack_weight_bf16x3, scale / shift (32) as there; heads_packed =
 * occ_conv3d_heads_pack(...) (occ_conv3d_heads_pack_bytes() bytes: the eight head tensors of occ_occ_heads_f32 as bf16
 * hi/lo MFMA fragments + biases, C = 32, hidden = 64).  Outputs in the reference's (B, X, Y, Z, .) order:
 * occ_out (.., num_classes), flow_out (.., 2), occ_cls_out (..) int64 or NULL — the same values occ_conv3d_bn_relu_bf16x3_f32
 * (out (X, Y)-major) followed by occ_occ_heads_decode_f32(exact_f32 = 0) produce.  Cin == 32; Z == 16 (2 x 8 pillars per
 * block) or Z == 32 (2 x 4 pillars, BASELINE configs[4]; round 4). */
int64_t occ_conv3d_heads_pack_bytes(void);
int occ_conv3d_heads_pack(const float* w1_occ, const float* b1_occ, const float* w2_occ, const float* b2_occ,
                          const float* w1_flow, const float* b1_flow, const float* w2_flow, const float* b2_flow,
                          void* packed, int C, int hidden, int num_classes, void* stream);
int occ_conv3d_heads_decode_bf16x3_f32(const float* in, const void* w_packed, const float* scale, const float* shift,
                                       const void* heads_packed, float* occ_out, float* flow_out,
                                       int64_t* occ_cls_out, int B, int Z, int Y, int X, int Cin, int num_classes,
                                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Occupancy heads on every voxel feature row:
 *   occ  = Linear(hidden, num_classes)( Softplus( Linear(C, hidden)(feat) ) )     (predicter)
 *   flow = Linear(hidden, 2)( ReLU( Linear(C, hidden)(feat) ) )                   (flow_predicter)
 *   feat (n_rows, C) f32; weights in torch Linear layout (out_features, in_features), f32
 *   occ_out (n_rows, num_classes) f32 ; flow_out (n_rows, 2) f32
 * Fused kernel for C = 32, hidden = 64, num_classes <= 30; otherwise OCC_E_UNSUPPORTED.
 */
int occ_occ_heads_f32(const float* feat, const float* w1_occ, const float* b1_occ,
                      const float* w2_occ, const float* b2_occ, const float* w1_flow,
                      const float* b1_flow, const float* w2_flow, const float* b2_flow,
                      float* occ_out, float* flow_out, int64_t n_rows, int C, int hidden,
                      int num_classes, void* stream);
/* same + the decoded class per voxel, occ_cls[row] = argmax_c occ[row, c] (first index on ties) as int64 — the
 * reference's get_occ, P/bevformer/dense_heads/bevformer_occ_head.py:210-212 (softmax(-1).argmax(-1); softmax is
 * monotonic) — written by the same pass; occ_cls_out may be NULL.  exact_f32 != 0: the f32 matrix instruction
 * (occ_occ_heads_f32's kernel); 0: bf16x3 arithmetic (hi/lo-split operands on the bf16 MFMA, f32 accumulation,
 * product error <= 2^-16) — 5x less matrix-pipe time. */
int occ_occ_heads_decode_f32(const float* feat, const float* w1_occ, const float* b1_occ, const float* w2_occ,
                             const float* b2_occ, const float* w1_flow, const float* b1_flow,
                             const float* w2_flow, const float* b2_flow, float* occ_out, float* flow_out,
                             int64_t* occ_cls_out, int64_t n_rows, int C, int hidden, int num_classes,
                             int exact_f32, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.Linear on the f32 matrix cores (exact f32) with the encoder's elementwise tail fused:
 *   out = LayerNorm( residual + act( [A1 | A2 (+ A2add)] @ W^T + bias ) )
 *   a1 (M, K1) row stride lda1 ; optional second K segment a2 (M, K2) row stride lda2 with an optional
 *   addend a2_add of the same shape/stride (the TSA query `cat([value, query + query_pos], -1)`,
 *   temporal_self_attention.py:197) ; weight (N, K1+K2) torch Linear layout ; bias (N) or NULL ;
 *   act 0 = none, 1 = ReLU ; residual (M, N) row stride ldres or NULL (added after act) ;
 *   ln_gamma / ln_beta (N) or both NULL, ln_eps: LayerNorm over the N outputs ; out (M, N) row stride ldo.
 * Requires K1, K2 multiples of 16, N % 4 == 0, 16-byte aligned rows, N <= 256 with LayerNorm;
 * otherwise OCC_E_UNSUPPORTED (the caller keeps the library GEMM).
 * Call sites replaced: value_proj / sampling_offsets / attention_weights / output_proj Linears
 * (spatial_cross_attention.py:334-341,173 ; temporal_self_attention.py:198-209,266), mmcv FFN + the three
 * LayerNorms of a BEVFormerLayer (encoder.py:377-404).
 */
int occ_linear_f32(const float* a1, int64_t lda1, int K1, const float* a2, const float* a2_add,
                   int64_t lda2, int K2, const float* weight, const float* bias, int act,
                   const float* residual, int64_t ldres, const float* ln_gamma, const float* ln_beta,
                   float ln_eps, float* out, int64_t ldo, int M, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * occ_linear_f32's fast variant on the bf16 matrix cores ("bf16x3"): every f32 operand is split into
 * hi + lo bf16 and A.W^T ~= Ah.Wh^T + Ah.Wl^T + Al.Wh^T is accumulated in f32 (relative error of a product
 * <= 2^-16).  Same arguments, except that the weight is given PACKED by occ_linear_pack_weight_bf16x3 in
 * MFMA fragment order: packed[K/16][ceil(N/32)][hi, lo][lane][8] bf16 = ceil(N/32)*32 * K * 2 16-bit words
 * (columns zero-padded to a multiple of 32; K % 16 == 0).
 */
int occ_linear_pack_weight_bf16x3(const float* weight, void* packed, int N, int K, void* stream);
int occ_linear_bf16x3_f32(const float* a1, int64_t lda1, int K1, const float* a2, const float* a2_add,
                          int64_t lda2, int K2, const void* weight_packed, const float* bias, int act,
                          const float* residual, int64_t ldres, const float* ln_gamma,
                          const float* ln_beta, float ln_eps, float* out, int64_t ldo, int M, int N,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-local Linear CHAINS of a BEVFormerLayer, one launch each (csrc/linear_chain_x3.hip; bf16x3 arithmetic as
 * occ_linear_bf16x3_f32; embed_dims = 256).  A 64-row block keeps the LayerNorm'd tile in LDS as the next Linear's
 * operand; the weights of all stages of a chain are ONE buffer in consumption order:
 *   occ_linear_chain_pack_bf16x3(W (N, K)) -> for every 256-row group of W: [K/16][8 tiles][hi | lo][lane][8 bf16]
 *   (rows beyond N are zero), occ_linear_chain_packed_bytes(N, K) bytes; the caller concatenates the packs of the stages.
 *
 * occ_linear_ln_chain_bf16x3_f32 — "program A":   y = LayerNorm(a.W1^T + b1 + residual)      (N = K = 256)
 *                                                 z = act2(y.W2^T + b2)                      (n2 columns, n2 % 32 == 0)
 *   w_chain = [pack(W1) | pack(W2)], bias_chain = [b1 (256) | b2 zero-padded to ceil(n2 / 256) * 256].
 *   Call sites replaced: TemporalSelfAttention.output_proj + residual (temporal_self_attention.py:266-272) + the layer's
 *   first LayerNorm (encoder.py:377-404 `norm`) + MSDeformableAttention3D.sampling_offsets | attention_weights
 *   (spatial_cross_attention.py:334-341) — z is what occ_sca_fused_forward_* reads.
 *
 * occ_encoder_ffn_chain_bf16x3_f32 — "program B": x2 = LayerNorm1(a.Wo^T + bo + residual)
 *                                                 y  = LayerNorm2(relu(x2.W1^T + b1).W2^T + b2 + x2)   (hidden = 512)
 *                                   optional tail: zq = y.Wq^T + q_term (nq <= 256 columns, nq % 64 == 0),
 *                                                  zv = y.Wv^T + bv     (256 columns)
 *   w_chain = [pack(Wo) | pack(W1) | pack(W2) | pack(Wq) | pack(Wv)] (the last two only with a tail),
 *   bias_chain = [bo | b1 (512) | b2 | 256 zeros | bv].  zq == zv == NULL: no tail.  `y` doubles as scratch (x2 is parked
 *   in the block's own rows before y is written).
 *   Call sites replaced: SpatialCrossAttention.output_proj + residual (spatial_cross_attention.py:173-175), the second
 *   LayerNorm, mmcv FFN (custom_base_transformer_layer.py:144-160) + the third LayerNorm, and — tail — the NEXT layer's
 *   TemporalSelfAttention sampling_offsets | attention_weights on cat([query, query + pos]) (temporal_self_attention.py
 *   :197-209; without history Wq = W[:, :256] + W[:, 256:], q_term = pos.W[:, 256:]^T + b) and its value_proj (:198).
 */
int64_t occ_linear_chain_packed_bytes(int N, int K);
int occ_linear_chain_pack_bf16x3(const float* weight, void* packed, int N, int K, void* stream);
int occ_linear_ln_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres,
                                   const void* w_chain, const float* bias_chain, const float* ln_gamma,
                                   const float* ln_beta, float ln_eps, float* y, int64_t ldy, float* z,
                                   int64_t ldz, int n2, int act2, int M, void* stream);
int occ_encoder_ffn_chain_bf16x3_f32(const float* a, int64_t lda, const float* residual, int64_t ldres,
                                     const void* w_chain, const float* bias_chain, const float* ln1_gamma,
                                     const float* ln1_beta, float ln1_eps, const float* ln2_gamma,
                                     const float* ln2_beta, float ln2_eps, float* y, int64_t ldy,
                                     const float* q_term, int64_t ldq_term, float* zq, int64_t ldzq, int nq,
                                     float* zv, int64_t ldzv, int M, void* stream);

/* Program C: the tail stage alone — two Linears of the SAME 256-wide rows in one launch (the first encoder layer's TSA
 * query Linears and value projection, temporal_self_attention.py:197-209,239-240, straight from the BEV queries):
 *   zq (M, nq) = a . Wq^T + q_term (or + 0),   zv (M, 256) = a . Wv^T + bv.
 * w_chain = chain packs of [Wq (nq <= 256 rows, zero-padded to 256), Wv]; bias_chain = [256 zeros | bv].
 * nq % 64 == 0; rows 16-byte aligned. */
int occ_linear_pair_chain_bf16x3_f32(const float* a, int64_t lda, const void* w_chain, const float* bias_chain,
                                     const float* q_term, int64_t ldq_term, float* zq, int64_t ldzq, int nq,
                                     float* zv, int64_t ldzv, int M, void* stream);

/* Training path of SpatialCrossAttention, query side (csrc/sca_prep.hip; reference spatial_cross_attention.py:338-373
 * applied to the per-camera rebatched rows): from proj (B, Q, >= 3*M*L*P) = [sampling_offsets | attention_weights]
 * Linear outputs per BEV query, row_to_query (R) (-1 = padded row) and the rebatched reference points ref_rb
 * (B, R, Z, 2):  attn (B, R, M, L, P) = softmax over L*P,  loc (B, R, M, L, P, 2) = ref_rb[.., p % Z, :] +
 * offsets / (W_l, H_l) — the layouts ms_deform_attn_forward takes.  The backward accumulates, per BEV query over the
 * <= Kq rows of query_to_rows (Q, Kq) (-1 = none), d proj (B, Q, proj_ld) from grad_loc / grad_attn (every row of
 * dproj is written).  M = 8, L = 4, P = 8 only (OCC_E_UNSUPPORTED otherwise: the caller keeps the ATen ops). */
int occ_sca_prep_forward_f32(const float* proj, int64_t proj_batch_stride, int proj_ld, const int64_t* row_to_query,
                             const float* ref_rb, const int64_t* spatial_shapes, float* loc, float* attn, int B,
                             int64_t R, int M, int L, int P, int Z, void* stream);
int occ_sca_prep_backward_f32(const float* grad_loc, const float* grad_attn, const float* attn,
                              const int64_t* query_to_rows, int Kq, const int64_t* spatial_shapes, float* dproj,
                              int proj_ld, int B, int64_t R, int64_t Q, int M, int L, int P, void* stream);

/* Row gather-sum (csrc/rows_index.hip): out[b][r][:] = sum over k < K with index[r*K + k] >= 0 of x[b][index[r*K + k]][:].
 * x (B, rows_in, F) with batch stride x_batch_stride (floats), index (rows_out, K) int64 (-1 = no row), out
 * (B, rows_out, F) contiguous; F % 4 == 0, 16-byte aligned.  K = 1: SpatialCrossAttention's per-camera rebatch of the
 * visible BEV queries (spatial_cross_attention.py:145-153); index = the inverse map: the scatter back into the BEV
 * slots (:165-167); each is the other's gradient, so the training path needs no float-atomic index_add. */
int occ_rows_gather_sum_f32(const float* x, int64_t x_batch_stride, const int64_t* index, int K, float* out, int B,
                            int64_t rows_out, int64_t rows_in, int F, void* stream);

/* Training partner of occ_linear_bf16x3_f32 (csrc/linear_wgrad.hip): weight and bias gradient of a Linear,
 *     dw[n][k] = sum_m dy[m][n] * x[m][k]      db[n] = sum_m dy[m][n]      (db may be NULL)
 * on the bf16 matrix cores with the same hi/lo operand split (product error <= 2^-16), f32 accumulation.
 * The row dimension is reduced in chunks into `workspace` (occ_linear_wgrad_workspace_bytes(M, N, K) bytes, owned by
 * the caller, 16-byte aligned) and summed in a fixed order: the result is deterministic, no float atomics.
 * dy (M, N) with row stride lddy, x (M, K) with row stride ldx, dw (N, K) contiguous (overwritten, not accumulated).
 * Replaces: ATen's addmm backward behind the nn.Linear call sites listed at occ_linear_f32.  The input gradient
 * needs no entry point: dx = occ_linear_bf16x3_f32(dy, pack(W^T)). */
int64_t occ_linear_wgrad_workspace_bytes(int M, int N, int K);
int occ_linear_wgrad_bf16x3_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                                void* workspace, int64_t workspace_bytes, int M, int N, int K, void* stream);

/* Training partner of occ_conv3d_bn_relu_bf16x3_f32 (csrc/linear_wgrad.hip linear_wgrad_n32_kernel; round 4): weight
 * gradient of a 3x3x3 / stride 1 / pad 1 Conv3d with 32 output channels — the decoder's two convolutions, reference
 * transformer_occ.py:106-126 (autograd of nn.Conv3d).  dy_pad (B, Y+2, X+2, Z+2, 32) and x_pad (B, Y+2, X+2, Z+2, Cin) are
 * ZERO-PADDED channels-last copies of the output gradient and the input: in the padded grid a tap is a constant row
 * offset, dW = dy_pad^T . im2col(x_pad) in one launch.  dw (32, 27, Cin): dW[co][kz*9 + ky*3 + kx][ci].  The input
 * gradient needs no entry point: dx = occ_conv3d_bn_relu_bf16x3_f32(dy, pack(flipped, transposed W), scale 1, shift 0). */
int64_t occ_conv3d_wgrad_workspace_bytes(int B, int Z, int Y, int X, int Cin);
int occ_conv3d_wgrad_bf16x3_f32(const float* dy_pad, const float* x_pad, float* dw, void* workspace,
                                int64_t workspace_bytes, int B, int Z, int Y, int X, int Cin, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backbone tail (outside the hand-written hot path): in place x = relu?(x + bias[c] (+ residual)) on an
 * NHWC bf16 activation of `rows` = N*H*W pixels x C channels (C % 8 == 0, 16-byte aligned).  Replaces the
 * BatchNorm / ReLU / residual-add launches after each MIOpen convolution of the ResNet-50 bottlenecks once
 * eval-mode BatchNorm is folded into the convolution weights.
 */
int occ_bias_act_nhwc_bf16(void* x, const float* bias, const void* residual, int64_t rows, int C, int relu,
                           void* stream);

/* Backward of that tail for the training step's autocast backbone (the reference trains its ResNet with eval-mode
 * BatchNorm, bevformer_base_occ.py:55; conv -> + bias (+ residual) -> ReLU is one autograd node in plugin/backbone.py):
 *   g = relu ? (y > 0 ? grad_y : 0) : grad_y   (bf16 NHWC; relu == 0: g is not written and may be NULL, y is not read)
 *   bias_grad[c] = sum over rows of g[., c]     (fp32, deterministic: per-block partial sums added in a fixed order)
 * in one pass over the activation.  partial: occ_bias_act_bwd_partial_floats(rows, C) floats of scratch (0 = unsupported C:
 * C % 8 != 0 or C > 2048). */
int64_t occ_bias_act_bwd_partial_floats(int64_t rows, int C);
int occ_bias_act_bwd_nhwc_bf16(const void* grad_y, const void* y, void* g, float* partial, float* bias_grad,
                               int64_t rows, int C, int relu, void* stream);

/* Weight side of the same training nodes: eval-mode BatchNorm folded into a convolution weight, and the fold's chain rule,
 * one launch each (conv_bn_folded's arithmetic, plugin/backbone.py; rstd = 1 / sqrt(running_var + eps), mean_rstd =
 * running_mean * rstd are constants of the frozen statistics).  weight / w_folded / grad_weight: (Cout, Cin, KH, KW) fp32
 * contiguous; w_folded_bf16_nhwc: the same values as bf16 in channels_last order [Cout][KH][KW][Cin]; bias[o] = beta[o] -
 * gamma[o] * mean_rstd[o].  Backward: grad_w_bf16 = the convolution's bf16 weight gradient with the given ELEMENT strides;
 * grad_weight = grad_w * gamma * rstd; grad_gamma[o] = rstd[o] * sum(grad_w[o] * weight[o]) - mean_rstd[o] * grad_bias[o]
 * (grad_beta = grad_bias). */
int occ_conv_bn_fold_fwd_f32(const float* weight, const float* gamma, const float* beta, const float* rstd,
                             const float* mean_rstd, float* w_folded, void* w_folded_bf16_nhwc, float* bias, int Cout,
                             int Cin, int KH, int KW, void* stream);
int occ_conv_bn_fold_bwd_f32(const void* grad_w_bf16, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w,
                             const float* weight, const float* gamma, const float* rstd, const float* mean_rstd,
                             const float* grad_bias, float* grad_weight, float* grad_gamma, int Cout, int Cin, int KH, int KW,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step glue of a BEVFormerLayer (reference: encoder.py:377-404 with the module tails
 * spatial_cross_attention.py:173-175, temporal_self_attention.py:270-272 and mmcv's FFN): y = LayerNorm(dropout(x) +
 * residual) and its backward, one launch each, C = 256 (OCC_E_UNSUPPORTED otherwise).
 *   forward : z = x * keep / (1 - p) + residual ; y = (z - mean) * rstd * gamma + beta ; stats[r] = (mean, rstd)
 *   backward: grad_residual = LayerNorm's input gradient ; grad_x = grad_residual * keep / (1 - p) (p == 0: not written, may
 *             be NULL — it IS grad_residual) ; grad_gamma_beta = (sum_rows gy * zhat | sum_rows gy), 2 C floats, by a fixed-order
 *             two-stage sum (partial: occ_dropout_add_ln_bwd_partial_floats(rows, C) floats of scratch).
 * keep = occ_ln_dropout_hash(element index, seed_lo, seed_hi) >= (uint32)(p * 2^32): the mask is a pure function of (seed,
 * index) and is never stored; the caller draws `seed` per call and hands the same one to the backward. */
unsigned occ_ln_dropout_hash(unsigned index, unsigned seed_lo, unsigned seed_hi);
int64_t occ_dropout_add_ln_bwd_partial_floats(int64_t rows, int C);
int occ_dropout_add_ln_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta, float eps,
                               float p_drop, uint64_t seed, float* z, float* y, float* stats, int64_t rows, int C,
                               void* stream);
int occ_dropout_add_ln_bwd_f32(const float* grad_y, const float* z, const float* stats, const float* gamma, float p_drop,
                               uint64_t seed, float* grad_x, float* grad_residual, float* partial, float* grad_gamma_beta,
                               int64_t rows, int C, void* stream);

/* SCA value projection straight off the backbone's bf16 feature maps (rows A3/A8: replaces the reference's
 * fp32 feature flatten + embedding adds, transformer_occ.py:204-222, AND MSDeformableAttention3D.value_proj,
 * spatial_cross_attention.py:366), all FPN levels in one launch.  For segment (= level) s, row m = g*rpg_s + i:
 *   out[(g*out_group_rows + out_row0[s] + i)*ldo + n] = sum_k a[s][m*lda[s] + k] * W[n][k]
 *                                                      + group_bias[s][(g % bias_groups)*N + n]
 * n_segments <= 8 ; a[s] (rows[s], K) bf16 device pointers, lda[s] row strides ; the arrays a / lda / rows /
 * rows_per_group / out_row0 / group_bias are HOST arrays of n_segments entries (read at launch) ;
 * weight_packed = occ_linear_pack_weight_bf16x3(W (N, K)) (hi/lo split: products exact to 2^-17, two MFMAs per
 * k-step) ; group_bias[s] (bias_groups, N) f32 device pointers, or group_bias == NULL — for the SCA:
 * (cams_embeds[cam] + level_embeds[s]) . W^T + b, one row per camera ; out f32.
 * Needs K % 32 == 0, N % 4 == 0, lda % 8 == 0, ldo % 4 == 0.
 */
int occ_value_proj_bf16_f32(int n_segments, const void* const* a, const int64_t* lda, const int64_t* rows,
                            const int64_t* rows_per_group, const int64_t* out_row0,
                            const float* const* group_bias, int bias_groups, const void* weight_packed,
                            float* out, int64_t ldo, int K, int N, int64_t out_group_rows, void* stream);
/* same with the output written as fp16 for occ_sca_fused_forward_f16v, IN ITS PIXEL-PAIR ORDER: row r = out_row0[s] + i of
 * group g's block lands at out[(g*out_group_rows + (r & ~1)) * ldo + (n / 32) * 64 + (r & 1) * 32 + n % 32].  Needs
 * ldo == N, N % 32 == 0, out_group_rows even (pad an odd pixel count by one row).  out_scale: NULL, or a DEVICE float:
 * out = fp16(out_scale[0] * (a . W^T + group_bias)) — the range scale of occ_value_range_scale_bf16 (abi 2). */
int occ_value_proj_bf16_f16pairs(int n_segments, const void* const* a, const int64_t* lda, const int64_t* rows,
                            const int64_t* rows_per_group, const int64_t* out_row0,
                            const float* const* group_bias, int bias_groups, const void* weight_packed,
                            void* out, int64_t ldo, int K, int N, int64_t out_group_rows, const float* out_scale,
                            void* stream);
/* same with q16 pixel pairs (occ_sca_fused_forward_q16v); exists on the activation-resident kernel only (K == 256,
 * N % 256 == 0, rows_per_group >= 128): OCC_E_UNSUPPORTED otherwise — project to f32 and use occ_sca_rows_encode_q16. */
int occ_value_proj_bf16_q16pairs(int n_segments, const void* const* a, const int64_t* lda, const int64_t* rows,
                            const int64_t* rows_per_group, const int64_t* out_row0,
                            const float* const* group_bias, int bias_groups, const void* weight_packed,
                            void* out, int64_t ldo, int K, int N, int64_t out_group_rows, const float* out_scale,
                            void* stream);

/* Range-safe fp16 value rows (csrc/value_range.hip).  The reference keeps the SCA value rows in fp32
 * (spatial_cross_attention.py:75,387-390, @force_fp32); stored as fp16 a plane whose values pass 65 504 would be clamped.
 * This call derives, on the device and per call, one POWER-OF-TWO scale per projection plane from an a-priori bound:
 *     |a . W_p^T + gb_p|  <=  max|a| * row_l1[p] + bias_max[p],      row_l1[p] = max_n sum_k |W_p[n][k]|,
 *     scale_out[p] = 2^(15 - e) with bound_p = m 2^e, m in [0.5, 1)   (bound_p * scale <= 2^15 = half of the fp16 limit)
 * so no finite input can saturate; a zero / Inf / NaN bound gives 1.  max|a| is measured over every segment (bf16 rows,
 * K % 8 == 0, lda % 8 == 0); row_l1 / bias_max: DEVICE arrays of n_planes (<= 8) floats (abi 3: no host round trip, the
 * call is graph-safe and follows the live weights).
 * scale_out: DEVICE floats [0, n_planes) scales, [n_planes] max|a|, [n_planes + 1, 2 n_planes + 1) the bounds.
 * work: two DEVICE words of scratch, zeroed by the call itself (calls sharing `work` must be stream-ordered).  Scaling by
 * a power of two is exact both ways: results equal the unscaled fp16-row path's wherever that one does not saturate. */
int occ_value_range_scale_bf16(int n_segments, const void* const* a, const int64_t* lda, const int64_t* rows, int K,
                               int n_planes, const float* row_l1, const float* bias_max, float* scale_out,
                               uint32_t* work, void* stream);
/* The same scales when the PRODUCER of the maps already holds max|a|: amax8 = 8 DEVICE words, the largest sign-stripped
 * bf16 pattern each shard of the producer's blocks stored (occ_conv3x3_nhwc_bf16_amax; the maximum over the 8 words is
 * max|a|).  One 64-thread launch instead of a pass over the maps (52 us at the base config). */
int occ_value_range_scale_from_amax(const uint32_t* amax8, int n_planes, const float* row_l1, const float* bias_max,
                                    float* scale_out, void* stream);

/* Several projections of the SAME rows in one launch — the four encoder layers' SCA value projections depend on the
 * camera features only (spatial_cross_attention.py:366 in each of the 4 layers): weight_packed = pack of the
 * (n_planes * plane_cols, K) stacked weights, group_bias[s] (bias_groups, n_planes * plane_cols); projection p writes
 * plane p of `out` (planes plane_stride elements apart, rows of ldo elements, plane_cols columns; fp16 when out_f16) just
 * as the single-projection calls above would.  The column blocks of a row block are dealt to one XCD back to back: the
 * feature maps are read from HBM once instead of once per layer.  plane_cols % 256 == 0, otherwise OCC_E_UNSUPPORTED.
 * out_scale: NULL, or n_planes DEVICE floats: plane p is multiplied by out_scale[p] before it is stored (abi 2).
 * out_f16: 0 = f32 rows, 1 = fp16 pixel pairs, 2 = q16 pixel pairs (abi 3). */
int occ_value_proj_bf16_planes(int n_segments, const void* const* a, const int64_t* lda, const int64_t* rows,
                               const int64_t* rows_per_group, const int64_t* out_row0,
                               const float* const* group_bias, int bias_groups, const void* weight_packed, void* out,
                               int out_f16, int64_t ldo, int K, int n_planes, int plane_cols, int64_t plane_stride,
                               int64_t out_group_rows, const float* out_scale, void* stream);

/* ResNet stem in one launch (outside the hand-written hot path):
 *   out = max_pool2d(relu(conv2d(x, W 7x7, stride 2, pad 3) + bias), kernel 3, stride 2, pad 1)
 * x (batch, 3, H, W) f32 NCHW (rounded to bf16 while staged) ; weight_frag = occ_mfma_pack_b_frag_bf16 of the
 * (64, 224) matrix W2[co][ky*32 + kx*4 + c] = W[co][c][ky][kx] (zero at kx = 7 and c = 3) ; bias (64) f32 ;
 * out (batch, Hp, Wp, 64) bf16 NHWC with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).
 */
int occ_stem_conv7x7_pool_f32_bf16(const float* x, const void* weight_frag, const float* bias, void* out,
                                   int batch, int H, int W, void* stream);
/* The same stem fed with the RAW camera images: x (batch, Hs, Ws, 3) uint8 HWC; the pipeline's
 * NormalizeMultiviewImage ((x[to_rgb ? 2-c : c] - mean[c]) / std[c], float32) and PadMultiViewImage (zeros to
 * H x W) — reference P/datasets/pipelines/transform_3d.py:31-45,82-94 — run while the input tile is staged.
 * mean, std: 3 HOST floats each (network channel order). */
int occ_stem_conv7x7_pool_u8_bf16(const uint8_t* x, const void* weight_frag, const float* bias, void* out,
                                  int batch, int Hs, int Ws, int H, int W, const float* mean,
                                  const float* std, int to_rgb, void* stream);

/* Stem tail in one pass: out = max_pool2d(relu(y + bias), kernel 3, stride 2, padding 1) on NHWC bf16
 * (outside the hand-written hot path).  y (batch, H, W, C) bf16 raw convolution output ; bias (C) f32 ;
 * out (batch, (H-1)/2+1, (W-1)/2+1, C) bf16.  Needs C % 8 == 0.
 */
int occ_bias_relu_maxpool_nhwc_bf16(const void* y, const float* bias, void* out, int batch, int H, int W, int C,
                                    void* stream);

/* Backbone 1x1 convolution on NHWC bf16 (outside the hand-written hot path, like the call above):
 * out[(n,yo,xo), co] = relu?( sum_ci x[(n, yo*stride, xo*stride), ci] * weight[co, ci] + bias[co]
 *                             (+ residual[(n,yo,xo), co]) ), bf16 in, f32 accumulate, bf16 out.
 *   x (batch, Hin, Win, Cin) bf16 ; weight = occ_mfma_pack_b_frag_bf16 of the (Cout, Cin) matrix (MFMA
 *   B-fragment order, read straight from global memory) ; bias (Cout) f32 ; residual / out
 *   (batch, Hout, Wout, Cout) bf16 with Hout = (Hin-1)/stride + 1.  Needs Cin % 32 == 0, Cout % 32 == 0.
 *   residual_upsample2 != 0: residual is (batch, Hout/2, Wout/2, Cout) and is added nearest-upsampled x2 (the FPN
 *   top-down path: lateral conv + F.interpolate(coarser, nearest) in one launch; Hout, Wout even).
 */
int occ_conv1x1_nhwc_bf16(const void* x, const void* weight, const float* bias, const void* residual,
                          void* out, int batch, int Hin, int Win, int Cin, int Cout, int stride, int relu,
                          int residual_upsample2, void* stream);

/* Backbone 3x3 pad-1 convolution, stride 1 or 2, on NHWC bf16 with bias (+ ReLU) fused (outside the
 * hand-written hot path).  x (batch, H, W, Cin) bf16 ; weight packed by occ_conv3x3_pack_weight_bf16 from
 * torch's (Cout, Cin, 3, 3) f32 layout to [Cin/32][tap][co][32] bf16 ; bias (Cout) f32 ;
 * out (batch, (H-1)/stride+1, (W-1)/stride+1, Cout) bf16.
 * Needs Cin % 32 == 0, Cout % 128 == 0, stride in {1, 2}, otherwise OCC_E_UNSUPPORTED (the caller keeps MIOpen).
 */
int occ_conv3x3_pack_weight_bf16(const float* weight, void* packed, int Cout, int Cin, void* stream);
int occ_conv3x3_nhwc_bf16(const void* x, const void* weight_packed, const float* bias, void* out, int batch,
                          int H, int W, int Cin, int Cout, int stride, int relu, void* stream);
/* The same convolution; additionally folds max|out| (the sign-stripped bf16 pattern of every element it stores) into the
 * 8 DEVICE words amax8 with atomic maxima.  The words ACCUMULATE across launches: zero them once in front of the FPN's
 * output convolutions (the maps the reference flattens at transformer_occ.py:204-222), then hand them to
 * occ_value_range_scale_from_amax. */
int occ_conv3x3_nhwc_bf16_amax(const void* x, const void* weight_packed, const float* bias, void* out, int batch,
                               int H, int W, int Cin, int Cout, int stride, int relu, uint32_t* amax8, void* stream);

/* MFMA B-operand packing: f32 row-major (N, K) matrix -> bf16 in v_mfma_f32_32x32x16_bf16 fragment order
 * packed[((ks * N/32 + nt) * 64 + lane) * 8 + j] = w[nt*32 + (lane & 31)][ks*16 + (lane >> 5)*8 + j], so a wave
 * loads its operand of (k-step ks, column tile nt) as one coalesced 1 KB read.  Needs N % 32 == 0, K % 16 == 0.
 */
int occ_mfma_pack_b_frag_bf16(const float* weight, void* packed, int N, int K, void* stream);

/* One whole 64-mid-channel ResNet bottleneck (mmdet Bottleneck, style='pytorch', stride 1, eval BatchNorm
 * folded; outside the hand-written hot path) on NHWC bf16 in ONE launch — the 64-channel intermediates stay in
 * LDS:  out = relu( conv1x1( relu(conv3x3( relu(conv1x1(x,W1)+b1), W2)+b2), W3) + b3 + identity ).
 *   x (batch, H, W, Cin) bf16 ; out (batch, H, W, 256) bf16 ; b1, b2 (64) f32 ; b3 (256) f32
 *   w1_frag = pack_b_frag(W1 (64, Cin)) ; w2_frag = pack_b_frag(W2 as (64, 9*64) with k = (ky*3+kx)*64 + ci)
 *   downsample == 0 (Cin == 256): identity = x,  w3_frag = pack_b_frag(W3 (256, 64))
 *   downsample == 1 (Cin == 64):  identity = conv1x1(x, Wds) + bds,  w3_frag = pack_b_frag([W3 | Wds] (256, 128)),
 *                                 b3 := b3 + bds
 * Other channel counts: OCC_E_UNSUPPORTED (the caller keeps the per-layer kernels).
 */
int occ_bottleneck64_nhwc_bf16(const void* x, const void* w1_frag, const float* b1, const void* w2_frag,
                               const float* b2, const void* w3_frag, const float* b3, void* out, int batch,
                               int H, int W, int Cin, int downsample, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCCNET_AMD_H_ */
