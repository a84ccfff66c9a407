/*
 * slotvps_hip.h - C ABI of the MI355X (gfx950) slot-retriever decode path.
 *
 * The reference (SAITPublic/SlotVPS) implements this path in pure PyTorch; it has no FFI of its
 * own. Every entry point below therefore names the reference *Python* function whose body it
 * replaces (file:line relative to the reference tree). The host-side mirror of the reference's
 * module interface lives in slotvps_amd/ and reaches these symbols through ctypes
 * (slotvps_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Rules common to all entry points
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise
 *   - the caller owns all buffers, including the workspace; nothing is allocated or freed here
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work,
 *     they never synchronise (safe under hipGraph capture)
 *   - return 0 on success, a positive hipError_t value if the runtime refused a launch, or one of
 *     the negative SVPS_ERR_* codes for argument errors (nothing is enqueued in that case)
 *   - thread-safe for distinct workspaces / output buffers
 */
#ifndef SLOTVPS_HIP_H_
#define SLOTVPS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVPS_ABI_VERSION 1

#define SVPS_ERR_BAD_ARG (-1)   /* null pointer / inconsistent argument */
#define SVPS_ERR_BAD_SHAPE (-2) /* shape outside what the kernels are built for */
#define SVPS_ERR_WORKSPACE (-3) /* workspace smaller than svps_*_workspace_bytes() */

/* flags of svps_slot_attn_fwd */
#define SVPS_FLAG_SPLIT_P 1 /* carry softmax probabilities as bf16 hi+lo (16-bit mantissa) */

/* flags of svps_mask_decode_fwd */
#define SVPS_FLAG_OUT_BF16 1 /* write mask logits as bf16 instead of fp32 */
#define SVPS_FLAG_MAP_F16 2  /* the fused map `feat` is fp16, not bf16 (MultiScaleDynamicMaskHead.map_dtype = "fp16") */

/* kernel ids of the launch hook (svps_set_launch_hook) */
#define SVPS_KERNEL_SLOT_ATTN 0
#define SVPS_KERNEL_SLOT_ATTN_FINISH 1
#define SVPS_KERNEL_MASK_DECODE 2
#define SVPS_KERNEL_POS_EMBED 3
#define SVPS_KERNEL_KV_PROJECT 4
#define SVPS_KERNEL_LEVEL_FUSE 5
#define SVPS_KERNEL_PANOPTIC_POST 6
#define SVPS_KERNEL_DEFORM_CONV 7
#define SVPS_KERNEL_RETR_STATS 8
#define SVPS_KERNEL_RETR_ATTN 9
#define SVPS_KERNEL_RETR_FINISH 10
#define SVPS_KERNEL_COUNT 11

int svps_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * K1 slot <-> pixel retriever.
 * Replaces the tensor program of MaskDynamicConv.forward after the three projections
 * (mmdet/models/detectors/dynamic_mask_head.py:435-459):
 *     attn = einsum("b l c, b h w c -> b l h w", q, k)      :435   (unscaled)
 *     attn = softmax(attn, dim=1)                            :446   (over the L slots)
 *     out  = einsum("b l h w, b h w c -> b l c", attn, v)    :456   (sum over all pixels)
 *     out  = ReLU(LayerNorm(out))                            :458-459
 * for T frames in one launch (the reference loops over frames, dynamic_mask_head.py:302).
 *
 *   q    [T, L, D]   bf16, already = norm_q(to_q(slots))               (:431)
 *   k    [T, HW, D]  bf16, already = norm_k(to_k(feat + pos)), row = pixel h*W + w   (:432)
 *   v    [T, HW, D]  bf16, already = norm_v(to_v(feat))                 (:433)
 *   ln_w, ln_b [D]   fp32, inst_interact.norm1 affine, ln_eps its epsilon (1e-5)
 *   out  [T, L, D]   fp32
 *   out_pre_ln       optional [T, L, D] fp32: the pixel sum before LayerNorm (NULL to skip)
 *   D must be 256; 1 <= L <= 256; 1 <= HW <= 4 Mi (any value, tiles are masked; byte offsets inside a frame are 32-bit)
 *   chunks: workgroups per frame, 0 = choose (one resident workgroup per CU)
 *   workspace: svps_slot_attn_workspace_bytes() bytes = the per-workgroup partial sums [T, chunks, L, D] fp32
 *              (+ [T, HW] x 8 B of per-pixel softmax statistics when L > 128: two-kernel path)
 * ------------------------------------------------------------------------------------------- */
size_t svps_slot_attn_workspace_bytes(int T, int L, int HW, int chunks);
int svps_slot_attn_plan(int T, int L, int HW, int chunks, int* out_chunks, int* out_tiles_per_chunk,
                        int* out_tile_px);
int svps_slot_attn_fwd(const void* q, const void* k, const void* v, const float* ln_w,
                       const float* ln_b, float ln_eps, void* workspace, size_t workspace_bytes,
                       float* out, float* out_pre_ln, int T, int L, int HW, int D, int flags,
                       int chunks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2 slot -> mask dot-product decode.
 * Replaces VPS_Temporal_Slots.generate_final_outputs, main branch
 * (mmdet/models/detectors/vps_temporal_slots.py:144-160):
 *     f = feat_bn(f)  (eval BatchNorm2d: per-channel affine)           :146
 *     f = F.normalize(f, p=2, dim=1)  (eps 1e-12)                       :147
 *     m = einsum("n c h w, n l c -> n l h w", f, slot_embed)            :149
 *     m = fg_bn(m)    (eval BatchNorm2d(1) with slots as batch: one scalar affine)  :153
 *
 *   feat     [T, HW, D] bf16 finest-level fused feature map, row = pixel
 *   embed    [T, L, D]  fp32 last-stage slot embeddings
 *   bn_scale, bn_shift [D] fp32: gamma/sqrt(var+eps), beta - mean*gamma/sqrt(var+eps)
 *   fg_scale, fg_shift: the scalar affine of fg_bn folded the same way
 *   out      [T, L, HW] fp32 (or bf16 with SVPS_FLAG_OUT_BF16); NULL = argmax-only mode (slot_argmax required): the logits are
 *            computed exactly as otherwise but not stored - 512 B in, 1 B out per pixel
 *   slot_argmax optional [T, HW] uint8: argmax over slots of the logits (first max wins), NULL to skip
 * ------------------------------------------------------------------------------------------- */
int svps_mask_decode_fwd(const void* feat, const float* embed, const float* bn_scale,
                         const float* bn_shift, float fg_scale, float fg_shift, void* out,
                         uint8_t* slot_argmax, int T, int L, int HW, int D, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sine position embedding, PositionEmbeddingSine.forward with normalize=True, scale 2*pi,
 * temperature 1e4, all-False mask (mmdet/models/detectors/position_encoding.py:236-256).
 *   out [H*W, D] fp32, pixel-major (the NHWC view the retriever consumes, dynamic_mask_head.py:430);
 *   channels [0, D/2) encode y, [D/2, D) encode x, sin on even / cos on odd channels.
 * ------------------------------------------------------------------------------------------- */
int svps_pos_embed_sine(float* out, int H, int W, int D, void* stream);
/* The same embedding in its separable form: ytab [H, D/2] (channels [0, D/2) of row y), xtab [W, D/2]
 * (channels [D/2, D) of column x); pos[h*W + w] = concat(ytab[h], xtab[w]) exactly. */
int svps_pos_embed_sine_tables(float* ytab, float* xtab, int H, int W, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3 key / value producer of K1: the pixel-side projections of MaskDynamicConv.forward
 * (mmdet/models/detectors/dynamic_mask_head.py:432-433) for all T frames of a stage:
 *     k = norm_k(to_k(features + pos))   v = norm_v(to_v(features))
 *   feat [T, H*W, D] bf16 pixel-major fused level map; pos_y [H, D/2], pos_x [W, D/2] fp32 tables of
 *   svps_pos_embed_sine_tables (both NULL: no position embedding);
 *   wk, wv [D, D] bf16 (nn.Linear.weight layout, row = output channel, rounded to bf16 by the caller),
 *   bk, bv [D] fp32;
 *   lnk_*, lnv_* the LayerNorm affines and epsilons; k_out, v_out [T, H*W, D] bf16.
 * ------------------------------------------------------------------------------------------- */
int svps_kv_project_fwd(const void* feat, const float* pos_y, const float* pos_x, const void* wk,
                        const float* bk, const float* lnk_w, const float* lnk_b, float lnk_eps,
                        const void* wv, const float* bv, const float* lnv_w, const float* lnv_b,
                        float lnv_eps, void* k_out, void* v_out, int T, int H, int W, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4 level fusion of the multi-scale head (mmdet/models/detectors/dynamic_mask_head.py:171-188), all T
 * frames of a level per launch:
 *     level i > 0:  f_i = conv_trans( cat( interpolate(f_{i-1}, x2, bilinear, align_corners=False), x_i ) )
 *     level 0   :   f_0 = conv_trans( cat( x_0, x_0, x_0 ) )               (prev == NULL)
 *   cur   the incoming 128-channel map: [T, 128, H, W] fp32 NCHW (cur_flags & 1, the reference's
 *         layout) or [T, H*W, 128] 16-bit pixel-major in the element type of the conv's operands: bf16, or fp16 with
 *         cur_flags == 2 (round 4: the semantic tower hands its own output over, conv_trans folded into wc / bc by the caller)
 *   prev  [T, (H/2)*(W/2), 256] bf16 pixel-major fused map of the previous (coarser) level, or NULL
 *   wc    [256, 384] bf16 conv_trans weight (row = output channel), bc [256] fp32 bias
 *   out   [T, H*W, 256] bf16 pixel-major
 *   cur_flags & 2: prev, wc and out are FP16: the conv and the bilinear blend run on fp16 operands - the
 *         same bytes with three more mantissa bits; every consumer of the map then takes SVPS_FLAG_MAP_F16.
 *   cur_flags & 4 (together with & 2): the bf16 storage POLICY in the fp16 ENCODING - wc is bf16, the conv runs on bf16 operands exactly as
 *         in the bf16 form, every value of `out` is rounded to bf16 first and stored as the fp16 number it equals (prev likewise holds bf16
 *         values in the fp16 encoding): bit-identical maps above fp16's subnormal range, and no consumer converts a tile any more.
 * ------------------------------------------------------------------------------------------- */
int svps_level_fuse_fwd(const void* cur, int cur_flags, const void* prev, const void* wc,
                        const float* bc, void* out, int T, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5 fused slot-side row step (rows of D = 256 fp32 values), one launch instead of add + LayerNorm +
 * ReLU + cast kernels between the small GEMMs of the slot update
 * (mmdet/models/detectors/dynamic_mask_head.py:356-358, :374-376, :384-385, :394-397, :431, :515-525, :555-570):
 *     y = LayerNorm(x [+ pre]) * w[g] + b[g];  y = relu(y) if relu;  y += post if post;  g = row / rows_per_group
 *   x, pre, post [rows, D] fp32 (pre / post may be NULL); w, b [G, D] fp32 with G = ceil(rows / rows_per_group);
 *   out_f32 [rows, D] and / or out_bf16 [rows, D] (at least one non-NULL).
 * ------------------------------------------------------------------------------------------- */
int svps_row_ln(const float* x, const float* pre, const float* post, const float* w, const float* b, float eps,
                int relu, int rows, int rows_per_group, int D, float* out_f32, void* out_bf16, void* stream);

/* Softmax over the last index of x [rows, cols] fp32 -> y (y may be x): the temporal retriever's softmax over the query axis
 * (mmdet/models/detectors/dynamic_mask_head.py:559-567: F.softmax(attn, dim=1) of [1, Lq, Lk], applied here to the transposed
 * logits [Lk, Lq]); max / exp / sum / divide in fp32 like torch.softmax. */
int svps_row_softmax(const float* x, float* y, int rows, int cols, void* stream);
/* the same times `scale` (> 0, a power of two): probabilities for a consumer that carries them as fp16 hi + lo (svps_bgemm_f16 with
 * alpha = 1 / scale) - keeps near-zero probabilities out of fp16's subnormal range (precision "fp16x2", the temporal retriever) */
int svps_row_softmax_scaled(const float* x, float* y, int rows, int cols, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K7' deformable convolution forward without a column buffer (slotvps_amd/csrc/deform_conv_fused.hip): replaces
 * deform_conv_forward_cuda (mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-258 = deformable_im2col,
 * deform_conv_cuda_kernel.cu:190-241, + addmm_) in one kernel: the 3 x 3 bilinear taps of 128 output pixels are gathered
 * chunk by chunk into LDS as matrix-core operands; operands as bf16 hi + lo, three products, fp32 accumulation (fp32-class).
 *   x_nhwc [N, H, W, C] fp32; offset [N, 18, Ho, Wo] fp32 (channel 2t = dy, 2t+1 = dx of tap t = 3i + j);
 *   wpack: the weight [O, C, 3, 3] as [O/32][9C/16][2 (hi, lo)][64 lanes][8] bf16 in MFMA A-fragment order with k = t*C + c:
 *          element (ob, ks, part, 32h + r, j) = part(W[32 ob + r, c, i, j']) at k = 16 ks + 8 h + j  (slotvps_amd/dcn.py packs it)
 *   out [N, Ho*Wo, O] fp32 pixel-major.   3 x 3 kernels, one deformable group, C % 64 == 0, O = 128 or 256.
 * ------------------------------------------------------------------------------------------- */
int svps_deform_conv_fused_fwd(const float* x_nhwc, const float* offset, const void* wpack, float* out, int N, int C, int H,
                               int W, int O, int kh, int kw, int pad, int stride, int dil, int Ho, int Wo, void* stream);
/* The same with the per-channel sums and sums of squares of the result for the GroupNorm behind the layer (round 4: the statistics pass
 * re-read the whole result): gn_partial [N, chunks, 2, O] fp32, chunks = svps_deform_conv_fused_stats_chunks(N, O, Ho, Wo) per frame (the launch picks its tile shape from N, O, Ho Wo),
 * written deterministically (one lane per entry); hand it to svps_group_norm_relu_stats_fwd. NULL: no statistics. */
int svps_deform_conv_fused_stats_chunks(int N, int O, int Ho, int Wo);
int svps_deform_conv_fused_stats_fwd(const float* x_nhwc, const float* offset, const void* wpack, float* out, float* gn_partial, int N, int C,
                                     int H, int W, int O, int kh, int kw, int pad, int stride, int dil, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm + ReLU on pixel-major fp32 activations (slotvps_amd/csrc/gn_relu.hip): the normalisation between the deformable
 * convolutions of the semantic tower (mmdet/models/panoptic/upsnetFPN.py:36-49: nn.GroupNorm(32, C) + nn.ReLU), in the layout
 * svps_deform_conv_fused_fwd reads and writes.
 *   x, y [N, HW, C] fp32 pixel-major; y_nchw [N, C, HW] fp32 or NULL (the same result in the layout the framework's
 *   convolutions take); gamma, beta [C]; C % 4 == 0, C % groups == 0, C in {32, 64, 128, 256, 512, 1024}.
 *   workspace: svps_group_norm_relu_workspace_bytes(N, HW, C).
 * ------------------------------------------------------------------------------------------- */
size_t svps_group_norm_relu_workspace_bytes(int N, int HW, int C);
int svps_group_norm_relu_fwd(const float* x, const float* gamma, const float* beta, int groups, float eps, float* y,
                             float* y_nchw, void* workspace, size_t workspace_bytes, int N, int HW, int C, void* stream);
/* The same with a third optional result: y16 [N, HW, C] 16-bit pixel-major - bf16 (y16_is_fp16 = 0), fp16 (1, saturating) or TWO fp16
 * planes hi + lo [2, N, HW, C] (2: hi = fp16(x), lo = fp16(x - hi), what svps_level_fuse_hl_pm_fwd takes) - the form in which the tower's
 * LAST layer hands its output to svps_level_fuse_fwd / svps_level_fuse_hl_pm_fwd (16-bit pixel-major `cur`; the linear 1x1 conv_trans
 * between them, mmdet/models/detectors/vps_capsule.py:76-79, folded into K4's weights by the caller); y, y_nchw, y16: any subset. */
int svps_group_norm_relu16_fwd(const float* x, const float* gamma, const float* beta, int groups, float eps, float* y, float* y_nchw,
                               void* y16, int y16_is_fp16, void* workspace, size_t workspace_bytes, int N, int HW, int C, void* stream);
/* ... with the statistics supplied by the producer of x (partial [N, chunks, 2, C] as svps_deform_conv_fused_stats_fwd writes them;
 * NULL: computed here) - the pass over x for the moments disappears. Same workspace size. */
int svps_group_norm_relu_stats_fwd(const float* x, const float* partial, int chunks, const float* gamma, const float* beta, int groups,
                                   float eps, float* y, float* y_nchw, void* y16, int y16_is_fp16, void* workspace, size_t workspace_bytes,
                                   int N, int HW, int C, void* stream);
/* [N, C, HW] fp32 (NCHW) -> [N, HW, C] fp32 pixel-major: the layout copy in front of the semantic tower's first layer (C % 4 == 0,
 * 256 % (C / 4) == 0). */
int svps_nchw_to_pixel_major(const float* x, float* y, int N, int C, int HW, void* stream);
/* The prediction layer of the semantic head in one kernel (mmdet/models/panoptic/upsnetFPN.py, forward):
 * out [N, K, H, W] = conv1x1( cat( p0, up2(p1), up4(p2), up8(p3) ) ) + bias, p_i [N, C, H >> i, W >> i] fp32 NCHW, bilinear
 * upsampling with align_corners = False (torch's arithmetic), weight [K, 4 C] (K <= 32), H % 8 == W % 8 == 0. */
int svps_semantic_pred_fwd(const float* p0, const float* p1, const float* p2, const float* p3, const float* weight, const float* bias,
                           float* out, int N, int C, int K, int H, int W, void* stream);
/* The offset-producing convolution of DeformConvWithOffset (mmdet/ops/dcn/deform_conv.py: conv_offset - 3 x 3, stride 1, padding 1,
 * C -> O <= 32 channels) on pixel-major activations: x_nhwc [N, H, W, C] fp32, wpack = the [32, 9 C] matrix (row o < O: w[o, c, tap] at
 * k = tap C + c; rows >= O zero) in B-fragment order with bf16 hi + lo halves (ops.pack_b_fragments), bias [O] or NULL ->
 * out [N, O, H, W] fp32 (the layout svps_deform_conv_fused_fwd reads its offsets in). C % 32 == 0. fp32-class. */
int svps_conv3x3_pm_small_fwd(const float* x_nhwc, const void* wpack, const float* bias, float* out, int N, int C, int H, int W, int O,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Slot-side helpers (slotvps_amd/csrc/row_ops.hip), rows of D = 256 fp32 values.
 *
 * svps_retr_query_prep: the query side of the fused retriever from x = to_q(slots) [T, L, D] (dynamic_mask_head.py:431):
 *     q = LN(x; lnq)   gp[t, l] = q * lnk_w   c3[t, l] = log2(e) q . lnk_b   a1[t, l] = gp[t, l] . bck
 *     rows l >= L: gp = 0, a1 = 0, c3 = -1e30 (the padding convention of svps_retr_attn_fwd)
 *     gp [T, LP, D] is the operand of Q'' = gp W~_k;  c3, a1 [T, LP];  LP = 128 or 256 >= L;  bck [D] = centred to_k bias
 * svps_retr_split: q2 [n] fp32 -> hi = fp16(q2), lo = fp16(q2 - hi) (n a multiple of 4): the FP16 operand pair of svps_retr_attn_fwd
 * svps_slot_self_attn: softmax(q k^T / sqrt(head_dim)) v per (frame, head) on the packed projection
 *     qkv [T, L, 3, nheads, head_dim] fp32 -> out [T, L, nheads * head_dim]; head_dim = 32, L <= 256
 *     (the attention of nn.MultiheadAttention between in_proj and out_proj, dynamic_mask_head.py:346-355)
 * ------------------------------------------------------------------------------------------- */
int svps_retr_query_prep(const float* x, const float* lnq_w, const float* lnq_b, float lnq_eps, const float* lnk_w,
                         const float* lnk_b, const float* bck, float* gp, float* c3, float* a1, int T, int L, int LP, int D,
                         void* stream);
int svps_retr_split(const float* q2, void* hi, void* lo, size_t n, void* stream);
int svps_slot_self_attn(const float* qkv, float* out, int T, int L, int nheads, int head_dim, void* stream);
/* the same with fp16 hi + lo operands (22 bits of mantissa instead of 16: the slot side of the reference-precision mode, round 5) */
int svps_slot_self_attn_f16(const float* qkv, float* out, int T, int L, int nheads, int head_dim, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K8 dense layers of the slot update (slotvps_amd/csrc/slot_gemm.hip): y = act(x W^T + bias), the nn.Linear layers of
 * MaskRCNNHead / TemporalSlotsHead / MaskDynamicConv.to_q (dynamic_mask_head.py:342-400, :494-572, :431) on [M, K] rows.
 * Matrix cores with both operands as bf16 hi + lo, three products, fp32 accumulation (fp32-class; the reference is fp32).
 *   x [M, K] fp32; wpack: W [N, K] as [N/32][K/16][2 (hi, lo)][64 lanes][8] bf16 in MFMA B-fragment order
 *   (element (cb, ks, part, 32h + r, j) = part(W[32 cb + r, 16 ks + 8 h + j]); slotvps_amd/ops.py pack_b_fragments);
 *   bias [N] fp32 or NULL; act 0 none / 1 ReLU / 2 GELU (erf form); y [M, N] fp32.  K % 16 == 0, N % 256 == 0.
 * ------------------------------------------------------------------------------------------- */
int svps_slot_gemm(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, int act, void* stream);
/* The same product with both operands split into fp16 hi + lo (22 bits of mantissa for the same three MFMAs; |x|, |w| < 65 504):
 * the query side of the fused retriever, MaskDynamicConv.forward's to_q (dynamic_mask_head.py:431) and the folded key projection.
 * wpack: fp16 fragments (pack_b_fragments(weight, split="fp16")). No activation. */
int svps_slot_gemm_f16(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, void* stream);
/* The fp16-split form with the epilogues of the bf16-split form (round 4; the slot side of precision "fp16x2"): act 0 none, 1 ReLU, 2
 * GELU (erf) as in svps_slot_gemm; the LayerNorm step of svps_slot_gemm_ln (N = 256; same arguments). */
int svps_slot_gemm_f16_act(const float* x, const void* wpack, const float* bias, float* y, int M, int K, int N, int act, void* stream);
int svps_slot_gemm_ln_f16(const float* x, const void* wpack, const float* bias, const float* pre, const float* post, const float* gamma,
                          const float* beta, float eps, int relu, float* y, int M, int K, void* stream);
/* the same product for N = 256 with the step that follows most dense layers of the slot update fused into the launch
 * (dynamic_mask_head.py:356-358, :374-376, :384-385, :394-397, :458-459, :515-525):
 *     y = LN(x W^T + bias [+ pre]) * gamma + beta  (+ReLU if relu)  (+ post)
 * LayerNorm over the 256 columns, biased variance, two-pass - the arithmetic of svps_row_ln operation for operation, so the
 * result is bitwise the one of svps_slot_gemm followed by svps_row_ln. pre / post [M, 256] fp32 or NULL. */
int svps_slot_gemm_ln(const float* x, const void* wpack, const float* bias, const float* pre, const float* post,
                      const float* gamma, const float* beta, float eps, int relu, float* y, int M, int K, void* stream);
/* The feed-forward block of a stage in ONE launch (slotvps_amd/csrc/slot_ffn.hip; dynamic_mask_head.py:379-385 and :519-525):
 *     y = LN( pre + W2 act(W1 x + b1) + b2 ) * gamma + beta  (+ post)
 * x, pre, post, y [M, 256] fp32 (pre / post may be NULL; the reference's residual is pre = x); w1pack / w2pack = the fragment-order
 * packs of W1 [H, 256] and W2 [256, H] (H % 256 == 0); act 1 ReLU / 2 GELU (erf). The hidden tensor never leaves the CU; the
 * arithmetic is svps_slot_gemm (act) followed by svps_slot_gemm_ln operation for operation - bitwise the same result. */
int svps_slot_ffn(const float* x, const void* w1pack, const float* b1, const void* w2pack, const float* b2, const float* pre,
                  const float* post, const float* gamma, const float* beta, float eps, int act, float* y, int M, int H, void* stream);
/* The same block in the fp16-split form (precision "fp16x2"): weights packed with split "fp16", activations split into fp16 hi + lo. */
int svps_slot_ffn_f16(const float* x, const void* w1pack, const float* b1, const void* w2pack, const float* b2, const float* pre,
                      const float* post, const float* gamma, const float* beta, float eps, int act, float* y, int M, int H, void* stream);

/* A chain of up to 6 layers  y_l = LN(src_l W_l^T + b_l [+ pre_l]) * gamma_l + beta_l (+ReLU if relu_l) (+ post_l)  on the same
 * [M, 256] rows in ONE launch (slotvps_amd/csrc/slot_chain.hip): the class / embedding towers (dynamic_mask_head.py:394-397), the
 * q / k / v projections of the temporal retriever (:555-557), out_proj + norm1 followed by to_q + norm_q (:356-358, :431).
 * src_l = 0: the chain's input x; 1: the result of layer l - 1 (kept on the CU as the next operand). All arguments are HOST arrays
 * of n_layers entries: wpack_l = the fragment-order pack of W_l [256, 256]; bias_l / pre_l / post_l / out_l may be NULL (out_l NULL:
 * the result only feeds the next layer; the last layer must have an out). Arithmetic: svps_slot_gemm_ln per layer, bit for bit. */
int svps_slot_chain(const float* x, int M, int n_layers, const void* const* wpack, const float* const* bias,
                    const float* const* gamma, const float* const* beta, const float* eps, const int* relu,
                    const float* const* pre, const float* const* post, float* const* out, const int* src, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K9 small batched products of the slot side (slotvps_amd/csrc/bgemm.hip), the products K8 does not take because their B
 * operand is an activation: the slot <-> slot retriever of the temporal head, k q^T and softmax(.)^T v
 * (dynamic_mask_head.py:550-572), the separable position terms of the fused retriever, the 20-column class projection (:398).
 *     C[b, m, n] = alpha * sum_k A[b, m, k] B[b, n, k] (+ bias[b, n])
 * fp32 in / out, split-bf16 products on the matrix cores with fp32 accumulation (fp32-class, as K8).
 *   sa = {batch, m, k}, sb = {batch, n, k}, sc = {batch, m, n}, sbias = {batch, n}: ELEMENT strides (a transposed operand is a
 *   stride pattern, batch stride 0 a shared operand); bias / sbias may be NULL. 1 <= batch <= 65535, any M, N, K >= 1.
 * ------------------------------------------------------------------------------------------- */
int svps_bgemm(const float* a, const long long* sa, const float* b, const long long* sb, const float* bias,
               const long long* sbias, float* c, const long long* sc, int batch, int M, int N, int K, float alpha, void* stream);
/* The same product with both operands split into fp16 hi + lo (22 bits of mantissa, |a|, |b| < 65 504): the separable position terms
 * Cy, Cx of the fused retriever (sine tables x folded queries), which end up inside logits that cancel. */
int svps_bgemm_f16(const float* a, const long long* sa, const float* b, const long long* sb, const float* bias,
                   const long long* sbias, float* c, const long long* sc, int batch, int M, int N, int K, float alpha, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K6 full-resolution panoptic post-process (PostProcessPanopticInstances.mask_removal / get_ids_area,
 * mmdet/models/detectors/vps_temporal_slots.py:564-657, :697-698, :724-757; argmax + relabel of simple_test
 * :411-435). Bilinear upsampling to H x W is fused into both kernels; the order-dependent part of
 * mask_removal runs on the small integer tables they produce (slotvps_amd/postprocess.py).
 *
 * svps_panoptic_candidates: masks [K, h, w] fp32 low-resolution logits of the K kept slots in DESCENDING score
 *   order, is_thing [K] uint8. Per full-resolution pixel the (at most two) thing slots whose softmax-over-K
 *   probability reaches pixel_threshold -> cand [H*W, 2] uint8 (255 = none); counts [K] int32 += pixels per
 *   slot, pairs [K, K] int32 += pixels per candidate pair (row = earlier slot). counts / pairs zeroed by caller.
 * svps_panoptic_argmax: per pixel, first-max argmax over the n slots sel[j] (indices into the K list) of the
 *   masks after removal (stuff: upsampled logit; thing: upsampled logit where it is the first kept candidate,
 *   0 elsewhere; kept [K] uint8), id = lut[j] -> out_ids [H*W] uint8 (nullable), hist [256] int32 += pixels
 *   per id (nullable, zeroed by caller), out_masks [n, H*W] fp32 the masks themselves (nullable).
 * ------------------------------------------------------------------------------------------- */
int svps_panoptic_candidates(const float* masks, const uint8_t* is_thing, int K, int h, int w, int H, int W,
                             float pixel_threshold, uint8_t* cand, int* counts, int* pairs, void* stream);
int svps_panoptic_argmax(const float* masks, const uint8_t* sel, const uint8_t* sel_thing, int n,
                         const uint8_t* kept, const uint8_t* cand, const uint8_t* lut, int h, int w, int H, int W,
                         uint8_t* out_ids, int* hist, float* out_masks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K6c: the same post-process for the T frames of a CLIP with the decisions on the device (slotvps_amd/csrc/panoptic_clip.hip;
 * round 4): mask_removal's keep / drop loop (vps_temporal_slots.py:601-640), the stuff de-duplication of get_ids_area (:724-757),
 * the small-area loop (:760-790) and the relabel table of simple_test (:420-433) run in single-workgroup kernels on a per-frame state
 * block, the pixel passes evaluate the x4 upsampling per 4 x 4 output block. Requires H == 4 h, W == 4 w.
 *   masks [T, Ks, h, w] fp32 (frame_stride = Ks h w): logits of each frame's kept slots in descending score order;
 *   state [T, SVPS_PPC_STATE_INTS] int32: the caller fills K, THING[K], CL[K] (class ids) per frame and zeroes the rest;
 *   pairs [T, pair_stride] int32 zeroed (pair_stride >= max K^2); cand [T, H W, 2] uint8 scratch; out_ids [T, H W] uint8;
 *   small_option: 0 = "4", 1 = "4_256", 2 = "4096_256" (filter_small_option, :762-776);
 *   stages: bit 0 candidates + decisions, bit 1 `rounds` x (area pass, step), bit 2 the id pass. A frame is finished when
 *   state[SVPS_PPC_PHASE] == 2: N surviving slots CUR[N] (indices into its K list, stuff first), AREA[N], the relabel table LUT2;
 *   if a frame is not, call again with stages = 2 | 4 (finished frames ignore further rounds).
 * ------------------------------------------------------------------------------------------- */
#define SVPS_PPC_K 0
#define SVPS_PPC_N 1
#define SVPS_PPC_PHASE 2
#define SVPS_PPC_ROUNDS 3
#define SVPS_PPC_LUT_IDENT 4
#define SVPS_PPC_THING 16
#define SVPS_PPC_CL (SVPS_PPC_THING + 256)
#define SVPS_PPC_COUNTS (SVPS_PPC_CL + 256)
#define SVPS_PPC_KEPT (SVPS_PPC_COUNTS + 256)
#define SVPS_PPC_CUR (SVPS_PPC_KEPT + 256)
#define SVPS_PPC_LUT (SVPS_PPC_CUR + 256)
#define SVPS_PPC_HIST (SVPS_PPC_LUT + 256)
#define SVPS_PPC_AREA (SVPS_PPC_HIST + 256)
#define SVPS_PPC_LUT2 (SVPS_PPC_AREA + 256)
#define SVPS_PPC_SLOT (SVPS_PPC_LUT2 + 256)
#define SVPS_PPC_SCORE (SVPS_PPC_SLOT + 256)
#define SVPS_PPC_STATE_INTS (SVPS_PPC_SCORE + 256)
int svps_panoptic_clip_state_ints(void);
/* The score filter (:684-691) and the descending-score order (:580) of every frame on the device, so that nothing of the clip's
 * post-process waits for the host: scores / classes [T, L] = softmax(class logits).max(-1) (fp32 / int64), L <= 255; nc = number of
 * class logits. Fills K, SLOT[K] (kept slot ids, best score first), SCORE[K] (float bits), CL[K], THING[K] of the zeroed state and
 * index [T, L] int64 (the same ids padded with 0: the gather index of the decode). Equal scores among the kept slots of a frame
 * have no defined order in the reference (np.argsort's unstable default); here: descending slot id, numpy's scalar path. */
int svps_panoptic_clip_select(const float* scores, const long long* classes, int T, int L, int nc, int num_classes, int num_stuff,
                              float threshold, long long* index, int* state, void* stream);
/* frame_stride / (h * w) = slot rows per frame of `masks`: a frame whose K exceeds it is skipped (the caller decoded only the first rows
 * of the score order, sees K after its wait and runs the clip again with all rows). */
int svps_panoptic_clip(const float* masks, long long frame_stride, int T, int h, int w, int H, int W, int* state, int* pairs,
                       int pair_stride, uint8_t* cand, uint8_t* out_ids, float pixel_threshold, double fraction_threshold,
                       int small_option, int stuff_num, int rounds, int stages, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K7 deformable convolution forward, sampling half: deformable im2col of DCNv1
 * (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:82-114, :190-241; host wrapper deform_conv_cuda.cpp:152-258).
 *   x_nhwc [N, H, W, C] fp32 pixel-major input; offset [N, dg*2*kh*kw, Ho, Wo] fp32 in the reference's layout
 *   (channel 2*(i*kw+j) = dy, +1 = dx); cols [N, Ho*Wo, C*kh*kw] fp32, column index c*kh*kw + i*kw + j, so the
 *   convolution is cols @ weight.view(O, -1)^T (the reference's addmm, done by the caller).
 * ------------------------------------------------------------------------------------------- */
int svps_deform_im2col(const float* x_nhwc, const float* offset, float* cols, int N, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                       int deformable_groups, int Ho, int Wo, void* stream);
/* bf16 operand form (matrix-core GEMM with fp32 accumulation): x_nhwc [N, H, W, C] bf16; cols [N, Ho*Wo, kh*kw, C] bf16,
 * TAP-major - the matching GEMM operand is weight.permute(0, 2, 3, 1).reshape(O, kh*kw*C). Sampling and validity rules
 * as above, blend in fp32. (C / deformable_groups) % 8 == 0. */
int svps_deform_im2col_bf16(const void* x_nhwc, const float* offset, void* cols, int N, int C, int H, int W, int kh,
                            int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                            int deformable_groups, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch hook (round 6): the one trace point of the product library. Every entry point that launches one of the kernels named by
 * SVPS_KERNEL_* calls svps_prof_mark(id, 0, stream) before and svps_prof_mark(id, 1, stream) behind its launches; with no hook
 * installed (the default) that is one relaxed load and a branch. A profiler installs a callback with svps_set_launch_hook - the
 * diagnostics library (include/slotvps_hip_diag.h, libslotvps_hip_diag.so: HIP events on the launch stream, bench.py's roofline leg)
 * is one; the event bookkeeping, the hardware probes and the ablation / stamp builds are NOT part of this library.
 * ------------------------------------------------------------------------------------------- */
typedef void (*svps_launch_hook_t)(int kernel_id, int is_end, void* stream);
void svps_set_launch_hook(svps_launch_hook_t hook);   /* NULL removes it */
void svps_prof_mark(int kernel_id, int is_end, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Statistics-fused retriever (K3' + K1'): MaskDynamicConv.forward (dynamic_mask_head.py:423-461) without k / v tensors.
 * Both pixel-side LayerNorms are "one scalar per pixel times an affine map":
 *     norm_k(to_k(x_p)) = gamma_k * rstd_k(p) * (W~_k x_p + b~_k) + beta_k,   W~ = (I - 11^T / 256) W,  b~ = b - mean(b)
 * so the pixel side only has to produce rstd_k(p), rstd_v(p) (svps_retr_stats_fwd), the key projection is folded into the
 * queries and the value projection is applied after the pixel sum (svps_retr_attn_fwd + a [L, 272] x [272, 256] product on the
 * slot side). The fused map is read once per kernel; nothing of size [HW, 256] is written.
 *
 * svps_retr_stats_fwd   (:432-433, the LayerNorm statistics only)
 *   feat  [T, HW, 256] bf16 fused map
 *   rk, rv [256, 256] FP16: the UPPER-TRIANGULAR factor R of  [W~ | b~] = Q [R | r]  for to_k / to_v (host, float64 QR; the
 *       kernel converts the map tile to fp16 in LDS - exact - and runs fp16 x fp16: rstd_k needs the three extra mantissa bits);
 *   rbk, rbv [256] fp32: the column r.   |R x + r|^2 = |W~ x + b~|^2 = 256 * var.
 *   ty [H, 256], tx [W, 256] fp32 or both NULL: the separable sine tables ALREADY multiplied by the key factor,
 *       ty[y] = R_k[:, :128] pos_y[y], tx[x] = R_k[:, 128:] pos_x[x] (R_k (f + pos) = R_k f + ty[y] + tx[x]: the position term
 *       enters the key accumulators in fp32 and never passes through 16 bits)
 *   out: aux [T, HW, 8] 16-bit words = ONE 16-byte row per pixel: {1, hi(1/rstd_v), lo(1/rstd_v), 0} FP16, {rstd_k, rstd_v} fp32
 *       (rstd = 1 / sqrt(var + eps)) - the row svps_retr_attn_fwd stages with every pixel. Nothing else is written: 16 B per
 *       pixel, whole memory lines (rows written in part cost a third of the kernel's streaming rate in read-modify-write traffic).
 *
 * svps_retr_attn_fwd    (:435-456)
 *   The slot axis of the inputs is padded to LP = 128 rows (L <= 128) or 256 rows (L <= 256), rows >= L zero:
 *   qh, ql [T, LP, 256] FP16: hi / lo halves of Q'' = (q * gamma_k) W~_k   (q = norm_q(to_q(slots)), :431; svps_retr_split)
 *   cy [T, H, LP], cx [T, W, LP] fp32: Q''[:, :128] . pos_y[y] + (q * gamma_k) . b~_k  and  Q''[:, 128:] . pos_x[x]
 *   c3 [T, LP] fp32: log2(e) * q . beta_k, and <= -1e30 in the padded rows l >= L (that, with their zero Q'' / cy / cx,
 *       is what removes them from the softmax: the kernel applies no mask)
 *   out_ext [T, L, 272] fp32: { A_l = sum_p P rstd_v f_p (256), s1_l = sum_p P rstd_v, s0_l = sum_p P, 0 x 14 };
 *       pre-LayerNorm output (:456) = out_ext @ [ (gamma_v * W~_v)^T ; gamma_v * b~_v ; beta_v ; 0 ]
 *   aux: the rows of svps_retr_stats_fwd
 *   1 <= L <= 256; workspace svps_retr_attn_workspace_bytes() = per-workgroup partials [T, chunks, L, 260] fp32
 *   (+ [T, tiles, 16 KiB] when L > 128, tiles = ceil(W / 32) * H: the softmax runs over all 256 slot rows of a pixel, which one
 *   workgroup cannot hold together with their accumulators - a first kernel writes P * rstd_v of every slot block as fp16 into
 *   the workspace, a second one forms sum_p P f from it: every logit is computed once, the map is read twice)
 * ------------------------------------------------------------------------------------------- */
int svps_retr_stats_fwd(const void* feat, const float* ty, const float* tx, const void* rk, const float* rbk,
                        float lnk_eps, const void* rv, const float* rbv, float lnv_eps, void* aux, int T, int H, int W, int D,
                        int flags, void* stream);
/* svps_retr_stats_level_fwd: the statistics of ALL retriever stages of one pyramid level (n_stages = 1 or 2; the stages of a level
 * read the same fused map, MultiScaleDynamicMaskHead.forward :190-215) in ONE read of the map (slotvps_amd/csrc/retr_stats2.hip:
 * eight waves, stage s on waves 4 s .. 4 s + 3, factors in AGPRs). Arguments as svps_retr_stats_fwd, as HOST arrays of n_stages device
 * pointers / values; ty, tx must be given (zero tables for "no position embedding"); aux[s] receives stage s's rows, bit-for-bit
 * the format svps_retr_stats_fwd writes. */
int svps_retr_stats_level_fwd(const void* feat, int n_stages, const float* const* ty, const float* const* tx,
                              const void* const* rk, const float* const* rbk, const float* lnk_eps,
                              const void* const* rv, const float* const* rbv, const float* lnv_eps,
                              void* const* aux, int T, int H, int W, int D, int flags, void* stream);
size_t svps_retr_attn_workspace_bytes(int T, int L, int H, int W, int chunks);
int svps_retr_attn_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                       const void* feat, const void* aux, void* workspace,
                       size_t workspace_bytes, float* out_ext, int T, int L, int H, int W, int D, int chunks,
                       int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reference precision ON THE MATRIX CORES (round 4; head.set_mode("fp16x2")): the reference runs this path in fp32
 * (fp16_enabled = False, mmdet/models/detectors/vps_temporal_slots.py:55). gfx950's fp32 matrix instructions run at the vector
 * rate, so these entry points carry every 16-bit matrix operand as FP16 hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 bits of
 * mantissa, |x| < 65 504) and spend three MFMAs per product (hi hi + lo hi + hi lo) into one fp32 accumulator. The fused level maps
 * are TWO fp16 planes [T, H*W, 256] (hi, lo): 1 KiB per pixel, the size of an fp32 map, in the operand form of the consumers.
 * Against the reference's own fp32 outputs the free-running head then sits at the reference's own reproducibility (mask logits
 * <= 1e-4, slot argmax identical wherever decidable: tests/test_refprec_gpu.py) - the bounds only the vector-ALU exact mode below met.
 *   svps_level_fuse_hl_fwd   = svps_level_fuse_fwd (dynamic_mask_head.py:171-188). A 1x1 conv commutes with bilinear interpolation; with
 *                              W = [W_a | W_b] and G^(m)_i = f_i (W_a^m)^T the level recursion is
 *                                  G^(m)_i = up( G^(m+1)_{i-1} ) + (W_a^m W_b) x_i + W_a^m b      (level 0: W_a^m (W_1 + W_2 + W_3) x_0)
 *                              - ONE call computes out = up(gprev) + wb cur + bc: cur [T, 128, H, W] fp32 NCHW; gprev [T, (H/2)(W/2), 256]
 *                              fp32 or NULL (no upsampled term); wb_hi / wb_lo [256, 128] fp16 (the host composes W_a^m W_b in float64);
 *                              out_hi / out_lo the planes (both or neither: m = 0), out_f32 the same values as fp32 [T, H*W, 256] or NULL
 *                              (m >= 1: the operand of the finer levels); at least one output. Level i of n is n - i calls.
 *   svps_retr_stats_hl_fwd   = svps_retr_stats_fwd (:432-433, the two LayerNorm statistics) with hi + lo factors on the planes, both projections from
 *                              ONE read (slotvps_amd/csrc/retr_stats_hl.hip); same aux rows. The fp32 tables come in ACCUMULATOR order
 *                              (column 32 B + 16 h + 4 g + j = factor row 32 B + 8 g + 4 h + j): tyk [ty_rows, 256] = Ty + r_k, txk
 *                              [tx_rows, 256] = Tx (ty_rows = H or 1, tx_rows = W or 1: no position term), rbv [256] = r_v; tx_tiled
 *                              (W % 32 == 0): txk re-ordered to [W / 32][8 B][4 g][2 h][32 pixels][4 j], so that a wave's load for the
 *                              32 pixels of a tile is one contiguous KiB
 *   svps_retr_attn_hl_fwd    = svps_retr_attn_fwd (:435-456) with hi + lo probabilities on the planes, 1 <= L <= 256 (more than 128 slots: the
 *                              per-pixel softmax statistics over all slots first, 8 B per pixel in the workspace, then the retriever once per
 *                              half of the slots); tiles of 32 pixels, a hi tile and a lo tile per LDS stage (round 6:
 *                              slotvps_amd/csrc/retr_attn_hl32.hip); workspace svps_retr_attn_hl_workspace_bytes()
 *   svps_mask_decode_hl_fwd  = svps_mask_decode_fwd (vps_temporal_slots.py:144-160) on the planes: fp32 logits [T, L, HW] (required),
 *                              optional fused slot argmax [T, HW]; any HW, L <= 256
 * Same rules as everywhere: no allocation, no synchronisation, launched on `stream`, 0 or an error code.
 * ------------------------------------------------------------------------------------------- */
int svps_level_fuse_hl_fwd(const float* cur, const float* gprev, const void* wb_hi, const void* wb_lo, const float* bc,
                           void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream);
/* The same function with the incoming map as fp16 hi + lo PIXEL-MAJOR planes cur_hi / cur_lo [T, H*W, 128] (round 6: the semantic
 * tower's own output rows as svps_group_norm_relu_stats_fwd(.., y16_is_fp16 = 2) writes them, with the linear conv_trans
 * (vps_capsule.py:76-79) composed into wb / bc by the caller) - no NCHW detour, no split in the kernel: the operand tile is the rows. */
int svps_level_fuse_hl_pm_fwd(const void* cur_hi, const void* cur_lo, const float* gprev, const void* wb_hi, const void* wb_lo, const float* bc,
                              void* out_hi, void* out_lo, float* out_f32, int T, int H, int W, void* stream);
/* ALL orders m = 0 .. n - 1 of a level in ONE launch (round 6; 1 <= n <= 4: level i of 4 asks for n = 4 - i): HOST arrays of n device pointers
 * gprev (all NULL at level 0), wb_hi / wb_lo (the composed weights W_a^m W_b), bc, out_f32 (entry 0 ignored: order 0 writes the planes out_hi /
 * out_lo, orders >= 1 their fp32 G^(m)). cur: fp32 NCHW (cur_lo NULL) or the hi plane of pixel-major fp16 rows (cur_lo: the lo plane). The n
 * workgroups of a chunk of tiles meet on one XCD, so the incoming map crosses HBM -> L2 once; results bit-identical to n calls of
 * svps_level_fuse_hl_fwd / _pm_fwd (slotvps_amd/csrc/level_fuse_hl.hip, level_fuse_hl_multi_kernel). dynamic_mask_head.py:171-188. */
int svps_level_fuse_hl_multi_fwd(const void* cur, const void* cur_lo, int n, const float* const* gprev, const void* const* wb_hi,
                                 const void* const* wb_lo, const float* const* bc, void* out_hi, void* out_lo, float* const* out_f32,
                                 int T, int H, int W, void* stream);
int svps_retr_stats_hl_fwd(const void* feat_hi, const void* feat_lo, const float* tyk, int ty_rows, const float* txk, int tx_rows, int tx_tiled,
                           const void* rk_hi, const void* rk_lo, float lnk_eps, const void* rv_hi, const void* rv_lo,
                           const float* rbv, float lnv_eps, void* aux, int T, int H, int W, int D, void* stream);
size_t svps_retr_attn_hl_workspace_bytes(int T, int L, int H, int W, int chunks);
int svps_retr_attn_hl_fwd(const void* qh, const void* ql, const float* cy, const float* cx, const float* c3,
                          const void* feat_hi, const void* feat_lo, const void* aux, void* workspace, size_t workspace_bytes,
                          float* out_ext, int T, int L, int H, int W, int D, int chunks, void* stream);
int svps_mask_decode_hl_fwd(const void* feat_hi, const void* feat_lo, const float* embed, const float* bn_scale,
                            const float* bn_shift, float fg_scale, float fg_shift, float* out, uint8_t* slot_argmax,
                            int T, int L, int HW, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Exact mode: fp32 storage and fp32 arithmetic for the whole pixel side (slotvps_amd/csrc/exact_f32.hip).
 * The reference runs this path in fp32 (fp16_enabled = False, mmdet/models/detectors/vps_temporal_slots.py:55);
 * these entry points take and return `float` where their bf16 counterparts above take bf16, round nothing
 * below fp32, and exist so that the head can be compared free-running with the reference's fp32 outputs.
 * Same layouts (pixel-major [T, HW, 256]), same argument meaning, same error codes.
 *
 *   svps_level_fuse_f32_fwd    = svps_level_fuse_fwd    (dynamic_mask_head.py:171-188)
 *        cur [T, 128, H, W] fp32 NCHW; prev [T, (H/2)(W/2), 256] fp32 or NULL (level 0);
 *        wT [384, 256] fp32 = conv_trans.conv.weight[256, 384, 1, 1] TRANSPOSED; bc [256]; out [T, HW, 256] fp32
 *   svps_kv_project_f32_fwd    = svps_kv_project_fwd    (dynamic_mask_head.py:428-433)
 *        feat [T, HW, 256] fp32; wkT / wvT [256 in, 256 out] fp32 = to_k / to_v weight TRANSPOSED; k_out / v_out fp32
 *   svps_slot_attn_f32_fwd     = svps_slot_attn_fwd     (dynamic_mask_head.py:435-459); q, k, v fp32
 *        workspace: svps_slot_attn_f32_workspace_bytes(T, L, HW)
 *   svps_mask_decode_f32_fwd   = svps_mask_decode_fwd   (vps_temporal_slots.py:144-160); feat fp32, out [T, L, HW] fp32
 * ------------------------------------------------------------------------------------------- */
int svps_level_fuse_f32_fwd(const float* cur, const float* prev, const float* wT, const float* bc, float* out, int T,
                            int H, int W, void* stream);
int svps_kv_project_f32_fwd(const float* feat, const float* pos_y, const float* pos_x, const float* wkT,
                            const float* bk, const float* lnk_w, const float* lnk_b, float lnk_eps, const float* wvT,
                            const float* bv, const float* lnv_w, const float* lnv_b, float lnv_eps, float* k_out,
                            float* v_out, int T, int H, int W, int D, void* stream);
size_t svps_slot_attn_f32_workspace_bytes(int T, int L, int HW);
int svps_slot_attn_f32_fwd(const float* q, const float* k, const float* v, const float* ln_w, const float* ln_b,
                           float ln_eps, void* workspace, size_t workspace_bytes, float* out, float* out_pre_ln, int T,
                           int L, int HW, int D, void* stream);
int svps_mask_decode_f32_fwd(const float* feat, const float* embed, const float* bn_scale, const float* bn_shift,
                             float fg_scale, float fg_shift, float* out, int T, int L, int HW, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLOTVPS_HIP_H_ */
