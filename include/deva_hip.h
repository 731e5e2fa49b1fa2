/*
 * deva_hip.h -- C ABI of libdeva_hip.so: the MI355X (gfx950) kernels behind DEVA's per-frame
 * temporal-propagation path.
 *
 * The reference (hkchengrex/Tracking-Anything-with-DEVA) is pure Python/PyTorch and has no FFI;
 * each entry point below states the reference code (file:line under the reference root) whose
 * arithmetic it replaces.  The Python host side that mirrors the reference's module interface
 * (tracking-anything-with-deva_amd/deva/...) binds these symbols with ctypes; INTEGRATION.md shows
 * the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - activations are NCHW, exactly as the reference passes them;
 *   - memory banks are TOKEN-MAJOR arenas: row n holds the C channels of memory token n
 *     (the reference keeps [C, N] tensors and re-allocates them on every append,
 *     kv_memory_store.py:97-116);
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *   - no entry point allocates, frees or synchronises; all work is stream-ordered;
 *   - return value 0 = ok, non-zero = error, text via deva_hip_last_error().
 */
#ifndef DEVA_HIP_H
#define DEVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEVA_HIP_ABI_VERSION 9

int deva_hip_version(void);
const char* deva_hip_last_error(void);

/* Diagnostic (no counterpart in the reference): one launch of a register-only v_mfma_f32_32x32x2_f32 loop, 4 waves per
 * SIMD on every CU, `iters` x 16 MFMAs per wave.  Returns the flop of the launch (< 0 on error); the caller times it
 * with events on `stream`.  bench.py reports the rate beside the convolution roofline: under dense MFMA work the chip
 * clocks to its power budget, so the sustained rate sits below the data-sheet 157.3 TFLOP/s.  `operands`: >= 1024
 * floats (their values set the switching activity: random vs zeros), `sink`: as many writable floats. */
int64_t deva_probe_mfma_f32(const float* operands, int64_t operand_elems, float* sink, int iters, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32), fused prologue/epilogue.
 * Replaces every nn.Conv2d / GConv2D call of the path (big_modules.py:42-51,103-113,164-201;
 * modules.py:19-38,73-78,135-169; group_modules.py:41-67; resnet.py:46-114) with eval-mode
 * BatchNorm folded into the packed weights.
 *
 * Input  = virtual channel-concat of up to two NCHW tensors (replaces torch.cat in
 *          group_modules.py:119-121, modules.py:139,163, big_modules.py:180), each with its own
 *          batch stride (0 = broadcast over the batch, replaces .expand()).
 * Weight = packed [K][cout_pad] (cout contiguous, cout_pad = cout rounded up to 32, zero filled)
 *          with the K index ordered by `k_layout`:
 *            DEVA_KLAYOUT_TAP_MAJOR   k = tap*(C0+C1) + c                      (any channel count)
 *            DEVA_KLAYOUT_CHUNK32     k = ((c/32)*KH*KW + tap)*32 + c%32       (C0, C1 multiples of 32)
 *          CHUNK32 walks all taps of a 32-channel slab before moving on, so the 9 shifted
 *          re-reads of a 3x3 convolution hit the same L2 lines back to back.
 * out[b][m][oh][ow] = act( sum_k W[k][m] * in(k, b, oh, ow) + bias[m] + residual ) */
enum {
  DEVA_KLAYOUT_TAP_MAJOR = 0,
  DEVA_KLAYOUT_CHUNK32 = 1,
  /* flag, ORed to one of the K orders above: element (k, m) is stored at ((k/4)*cout_pad + m)*4 + k%4
   * ("k-quad interleaved", K rounded up to a multiple of 4 with zeros): the lean-loop kernels
   * (csrc/conv_mfma.hip) read four consecutive k of one output channel with one 16-byte load. */
  DEVA_KLAYOUT_Q4 = 16
};

enum {
  DEVA_ACT_NONE = 0,
  DEVA_ACT_RELU = 1,
  DEVA_ACT_SIGMOID = 2,
  DEVA_ACT_SQUARE_PLUS_ONE = 3 /* x*x+1: shrinkage head, modules.py:75 */
};

typedef struct deva_conv_desc {
  const float* in0;
  const float* in1; /* may be NULL when c1 == 0 */
  int64_t in0_batch_stride; /* elements; 0 broadcasts in0 over the batch */
  int64_t in1_batch_stride;
  int32_t c0, c1;
  int32_t batch, height, width; /* input spatial size */
  const float* weight; /* packed, see above */
  const float* bias;   /* [cout] or NULL */
  int32_t cout, cout_pad;
  int32_t k_layout;
  int32_t kh, kw, stride, pad;
  int32_t relu_in; /* apply max(x,0) to the inputs while loading (F.relu before the conv) */
  const float* residual; /* [batch or 1][cout][OH][OW] or NULL, added before the activation */
  int64_t residual_batch_stride; /* elements; 0 broadcasts */
  int32_t act;
  float* out; /* [batch][cout][OH][OW] */
  /* Number of float elements that are allocated and readable immediately before the first and
   * after the last element of BOTH inputs (0 if unknown).  With >= pad*(W+1)+4 the stride-1
   * 'same' convolutions gather 4 consecutive pixels per 16-byte load (reads may touch the guard
   * band, the values are masked); otherwise every element is gathered with a clamped scalar load. */
  int32_t in_guard_elems;
  /* optional scratch for split-K (layers with too few output tiles to fill the GPU accumulate
   * K ranges in parallel and a second kernel reduces them in a fixed order); NULL disables it */
  float* workspace;
  int64_t workspace_elems;
  /* opt-in fp16 OPERANDS with fp32 accumulation (the reference's --amp: deva/inference/eval_args.py:17,
   * evaluation/eval_vos.py:137 wrap the frame loop in fp16 autocast): amp != 0 and weight_f16 != NULL run
   * v_mfma_f32_32x32x16_f16 on the inputs rounded to fp16 while they are staged and on fp16 weights packed by
   * deva_conv_pack_f16 (layout DEVA_KLAYOUT_H8: element (k, m) at ((k/8)*cout_pad + m)*8 + k%8; K tap-major for 1x1,
   * 64-channel slabs otherwise); bias / residual / activation / output stay fp32.  Shapes the fp16 kernels do not
   * cover (stride 2, channel counts that are not multiples of 64, single-channel heads, unguarded inputs) run the
   * fp32 kernels on `weight` as if amp were 0. */
  const void* weight_f16;
  int32_t amp;
  /* amp == 2: fp32-ACCURATE convolution on the f16 matrix pipes (the arithmetic of the reference's fp32 nn.Conv2d,
   * big_modules.py:54-212, modules.py:81-169, to fp32 round-off; opt-in, csrc/conv_f16.hip).  weight_f16 then points to
   * the hi / lo fp16 planes packed by deva_conv_pack_split (layout: element (k, plane, m) at
   * (((k/8)*2 + plane)*cout_pad + m)*8 + k%8, K tap-major for 1x1, 32-channel slabs otherwise) of the weights scaled
   * by 2^split_scale_log2; every activation is split into hi = fp16(x), lo = fp16(x - hi) while it is staged, each
   * K-block runs hi.hi + hi.lo + lo.hi with fp32 accumulation, and the accumulators are scaled back exactly.
   * ERROR MODEL (tests/test_split_arithmetic_cpu.py): per product |x w - (hi.hi + hi.lo + lo.hi)| <= 2^-21 |x w| PLUS
   * an ABSOLUTE floor of 2^-25 |w| -- activations are not pre-scaled, so below |x| ~ 2^-3 the lo plane enters the fp16
   * subnormals (spacing 2^-24) and the representation error of x stops shrinking with x.  With activations of ordinary
   * magnitude (max |x| of the layer >= ~0.1) the floor is below the fp32 kernels' own accumulation round-off; a layer whose
   * inputs are ALL tiny (|x| ~ 1e-4) is 11..12-bit accurate in x (fp16-operand class), not 22-bit.  "fp32-accurate" in this
   * file and in the flags' help means: to fp32 round-off of max(|x|, 2^-3) |w| per product.
   * `out` must not overlap in0 / in1 / residual for the split kernels to run (the fp32 re-run behind them reads those
   * after `out` has been written): an overlapping call -- an in-place residual add -- runs the fp32 kernels instead.
   * split_flag: one device int the caller has zeroed; the kernel sets it when an input lay beyond the fp16 range
   * (|x| > 65504 or non-finite), and the fp32 kernels -- launched behind the split kernel on the same stream, gated on
   * that int -- then produce the output.  Shapes the split kernels do not cover (stride 2, 3x3 layers whose channel counts
   * are not multiples of 32, kernels other than 1x1 / 3x3, single-channel heads, unguarded inputs) run the fp32 kernels
   * on `weight` directly.  1x1 layers may have any channel count behind a first source that is a multiple of 32
   * (513 = 512 + 1, 257 = 256 + 1: a partial last K step). */
  int32_t split_scale_log2;
  int32_t* split_flag;
  /* optional: the weights of a 3x3 / stride 1 / pad 1 layer transformed for Winograd F(2x2, 3x3) by deva_conv_pack_wino
   * (NULL: the direct kernels).  With amp == 0, even width (>= 4), c0 and c1 multiples of 8, cout >= 32,
   * guard-banded inputs and enough 2x2 output tiles to fill the chip (>= 160 workgroups of 64 channels x 64 tiles), the
   * layer runs csrc/conv_wino.hip: 16 instead of 36 multiply-adds per input channel and 2x2 outputs on the fp32 matrix
   * pipes, transforms with constants 0, +-1, +-1/2 in fp32 -- the arithmetic of the reference's nn.Conv2d
   * (big_modules.py:54-212, modules.py:81-169, resnet.py:78-152) with a quarter of the accumulated terms: measured
   * round-off below the direct kernels' on the network's layer shapes (the same 2e-5 gate in tests/test_gpu_a_conv.py).
   * Everything else runs the direct kernels on `weight`. */
  const float* weight_wino;
} deva_conv_desc;

int deva_conv2d(const deva_conv_desc* desc, void* stream);
/* fp16 weights of the amp path (HOST pointers, model load): -> number of uint16 elements (out == NULL: size query),
 * -1 when the layer is not eligible (cin % 64 != 0) or on bad arguments */
int64_t deva_conv_pack_f16(const float* w_oihw, uint16_t* out, int cout, int cin, int kh, int kw, int* cout_pad);
/* hi / lo fp16 planes of the split path (HOST pointers, model load): -> number of uint16 elements (out == NULL: size
 * query; *scale_log2 is set either way), -1 when the layer is not eligible (kh*kw > 1 and cin % 32 != 0; a 1x1 layer of any
 * cin is packed with its K rows padded with zeros to a multiple of 32), holds a non-finite weight,
 * or on bad arguments.  *scale_log2 = e with max|w| * 2^e in [2^13, 2^14): pass it as deva_conv_desc.split_scale_log2. */
int64_t deva_conv_pack_split(const float* w_oihw, uint16_t* out, int cout, int cin, int kh, int kw, int* cout_pad,
                             int* scale_log2);

/* The 7x7 stride-2 pad-3 stems (resnet.py:117-122 conv1 + bn1 (+ relu) of the key encoder; big_modules.py:58-61,103-107
 * conv1 + bn1 of the value encoder over cat(image, mask)) as a direct convolution on the f16 matrix pipes with the hi / lo
 * operand split of amp == 2 above (same error model; opt-in: --f16_split runs the value encoder's stem on it,
 * --f16_split_key_encoder the key encoder's).  in0 [1 or batch][c0 = 3][H][W] (in0_batch_stride 0 broadcasts the image
 * over the objects), in1 [batch][c1 = 0 or 1][H][W] (the object masks) or NULL; H, W even; out [batch][64][H/2][W/2]
 * = act(conv + bias), act = ReLU when relu != 0.  planes / w32 / scale_log2: from deva_stem_pack (DEVICE copies).
 * An input beyond the fp16 range (|x| > 65504, non-finite) does not saturate: the workgroup that meets one recomputes its
 * tile in fp32 from w32 and sets *flag (a device int the caller has zeroed; may be NULL) -- no second launch. */
int deva_stem7x7(const float* in0, int64_t in0_batch_stride, int c0, const float* in1, int64_t in1_batch_stride, int c1,
                 int batch, int height, int width, const uint16_t* planes, const float* w32, int scale_log2,
                 const float* bias, int relu, float* out, int32_t* flag, void* stream);
/* weights of a stem (HOST pointers, model load): w_oihw [64][cin][7][7] with BatchNorm folded, cin = 3 or 4 ->
 * planes: cin*8*2*64*8 uint16 (hi / lo fp16 of w * 2^e; k = ((c*4 + dy/2)*2 + dy%2)*8 + dx, element (k, plane, m) at
 * (((k/8)*2 + plane)*64 + m)*8 + k%8, taps dy = 7 / dx = 7 zero), w32: cin*49*64 floats ([(c*7 + dy)*7 + dx][m]).
 * Returns the number of uint16 elements of planes (planes == NULL: size query; *scale_log2 is set either way), -1 on error. */
int64_t deva_stem_pack(const float* w_oihw, int cin, uint16_t* planes, float* w32, int* scale_log2);

/* Winograd-transformed weights (HOST pointers, model load): w_oihw [cout][cin][3][3] (BatchNorm folded), cin % 8 == 0 ->
 * U = G g G^T computed in fp64 and rounded once, element (c, p = 4 i + l, m) at ((((c/8)*16 + p)*2 + c%2)*cout_pad64 + m)*4 +
 * (c%8)/2 with cout padded to a multiple of 64 (zeros).  Returns the number of floats (out == NULL: size query), -1 when
 * the layer is not eligible (cin % 8 != 0) or on bad arguments. */
int64_t deva_conv_pack_wino(const float* w_oihw, float* out, int cout, int cin);

/* Host-side packing of one convolution's weights (HOST pointers; model load, not the frame path):
 * w_oihw [cout][cin][kh][kw] (BatchNorm already folded) -> out in the layout named by *k_layout / *cout_pad
 * (32-channel slabs when kh*kw > 1 and cin % 32 == 0, tap-major otherwise; k-quad interleaved when want_q4 != 0
 * and cout > 1).  Returns the number of floats of the packed weight (out == NULL: size query only), -1 on error.
 * Replaces nothing in the reference (its nn.Conv2d weights stay [cout][cin][kh][kw], resnet.py:46-114). */
int64_t deva_conv_pack(const float* w_oihw, float* out, int cout, int cin, int kh, int kw, int want_q4,
                       int* k_layout, int* cout_pad);

/* ------------------------------------------------------------------------------------------
 * Pooling / resampling / pointwise blocks */

/* Zero padding of the last two dimensions (tensor_utils.py:7-22 pad_divide_by -> F.pad): in [planes][height][width] ->
 * out [planes][out_height][out_width] with the input at (top, left), zeros elsewhere; elements of 1, 4 or 8 bytes (uint8
 * / fp32 / int64 masks: bits are moved).  One launch instead of ATen's fill + copy. */
int deva_pad2d(const void* in, void* out, int elem_bytes, int64_t planes, int height, int width, int top, int left,
               int out_height, int out_width, void* stream);

/* The taps of a stride-2 convolution as channels: in [batch][channels][H][W] -> out [batch][k*k*channels][OH][OW],
 * out[b][t*C + c][oh][ow] = in[b][c][2 oh + dy - pad][2 ow + dx - pad] (zero outside), t = dy*k + dx, k = 1 (pad 0) or 3
 * (pad 1).  The stride-2 convolutions of the ResNets (resnet.py:46-114) then run as 1x1 stride-1 convolutions over
 * k*k*channels channels (weights [cout][t*C + c]) on the vector-gather kernels of deva_conv2d. */
int deva_gather_s2(const float* in, float* out, int batch, int channels, int height, int width, int kernel, void* stream);

/* nn.MaxPool2d(3, stride 2, pad 1) (resnet.py:122), optional fused ReLU after the pool
 * (MaskEncoder order, big_modules.py:107-110).  in [planes][H][W] -> out [planes][OH][OW]. */
int deva_maxpool3x3s2(const float* in, float* out, int64_t planes, int height, int width,
                      int relu_after, void* stream);

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) over [batch][C][h][w]
 * (group_modules.py:26-30) fused with the broadcast skip add of MaskUpsampleBlock
 * (modules.py:88-92): out = skip[c] + up(in[b][c]); skip is [C][2h][2w] or NULL. */
int deva_upsample2x_add(const float* in, const float* skip, float* out, int batch, int channels,
                        int height, int width, void* stream);
/* The same, and area_downsample(in, 2) of the INPUT from the same pass: ds2 [batch][channels][height/2][width/2] (height,
 * width even).  The decoder reads p8 once for the x2 up-sampling towards p4 (big_modules.py:164-201) and once more for the
 * 1/16 copy the sensory update takes (modules.py:121-151: downsample_groups(g[1], ratio=1/2)); this entry serves both.
 * ds2 is bit-identical to deva_area_downsample(in, ..., 2). */
int deva_upsample2x_add_ds2(const float* in, const float* skip, float* out, float* ds2, int batch, int channels, int height,
                            int width, void* stream);

/* F.interpolate(mode='area') for an integer shrink factor (network.py:117,
 * group_modules.py:33-38): mean over factor x factor boxes.  in [planes][H][W]. */
int deva_area_downsample(const float* in, float* out, int64_t planes, int height, int width,
                         int factor, void* stream);

/* Input head in one pass: uint8 [height][width][3] frame (device memory) -> fp32 [3][out_height][out_width],
 * (u/255 - mean)/std per channel and, when the size differs, the resize of the readers:
 * antialias != 0: transforms.Resize(..., BILINEAR, antialias=True) (deva/inference/data/video_reader.py:139-144,
 * detection_video_reader.py:63-71); antialias == 0: F.interpolate(bilinear, align_corners=False)
 * (deva/inference/demo_utils.py:10-19).  mean3 / std3 are HOST pointers to three floats.
 * The zero padding of pad_divide_by (deva/utils/tensor_utils.py:7-22) is fused: out is
 * [3][pad_top + out_height + pad_bottom][pad_left + out_width + pad_right] with a zero border. */
int deva_input_head(const unsigned char* image_hwc, int height, int width, const float* mean3,
                    const float* std3, int antialias, float* out, int out_height, int out_width,
                    int pad_left, int pad_right, int pad_top, int pad_bottom, void* stream);

/* DEVA.aggregate (network.py:33-40) over `num` object planes of `pixels` each:
 * p = apply_sigmoid ? sigmoid(in) : in;  out[0] = logit(clamp(prod(1-p)));  out[i+1] = logit(clamp(p_i)).
 * in may be fp32 or (in_is_u8 != 0) uint8/bool one-hot planes (inference_core.py:273-277).
 * num == 0 (nothing tracked yet) is allowed, in may then be NULL: out is the background-only plane. */
int deva_aggregate(const void* in, int in_is_u8, int apply_sigmoid, float* out, int num,
                   int64_t pixels, void* stream);

/* softmax over the channel axis of [channels][pixels] (inference_core.py:279) */
int deva_softmax_channels(const float* in, float* out, int channels, int64_t pixels, void* stream);

/* network.py:167-168: x4 bilinear upsample (align_corners=False) of [channels][h][w] logits and
 * the channel softmax, one pass.  Writes both logits_up and prob ([channels][4h][4w]). */
int deva_upsample4x_softmax(const float* in, float* logits_up, float* prob, int channels,
                            int height, int width, void* stream);

/* CBAM (cbam.py:21-76) on x [batch][C][hw], three small kernels + the fused apply.
 *  (1) global average and max per (b,c) plane;
 *  (2) scale = sigmoid(mlp(avg) + mlp(max)), mlp = Linear(C,hidden) -> ReLU -> Linear(hidden,C);
 *  (3) pooled[b][0] = max_c(x*scale), pooled[b][1] = mean_c(x*scale)   (ChannelPool)
 *  (4) after the 7x7 gate conv (deva_conv2d):  out = x + (x*scale)*sigmoid(gate)
 *      -- the "+ x" is the g + r of group_modules.py:149 fused in. */
int deva_global_avgmax(const float* x, float* avg, float* mx, int64_t planes, int hw, void* stream);
int deva_cbam_mlp(const float* avg, const float* mx, const float* w1, const float* b1,
                  const float* w2, const float* b2, float* scale, int batch, int channels,
                  int hidden, void* stream);
int deva_cbam_channel_pool(const float* x, const float* scale, float* pooled, int batch,
                           int channels, int hw, void* stream);
int deva_cbam_apply(const float* x, const float* scale, const float* gate, float* out, int batch,
                    int channels, int hw, void* stream);

/* GRU-style sensory update (modules.py:141-149,162-169): values [batch][3C][hw], h [batch][C][hw]
 * new_h = sig(v[:C]) * h * (1 - sig(v[C:2C])) + sig(v[C:2C]) * tanh(v[2C:]). */
int deva_gru_update(const float* values, const float* h, float* new_h, int batch, int channels,
                    int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory read: anisotropic-L2 similarity -> top-k -> softmax (-> usage), nothing materialised.
 * Replaces get_similarity + do_softmax(top_k, inplace, return_usage) (memory_utils.py:6-76) as
 * called from MemoryManager.match_memory (memory_manager.py:106-152).
 *
 * The bank is the virtual concatenation of two token-major segments (long-term rows first, then
 * working rows: memory_manager.py:110-113 without the torch.cat): keys [n][64], shrinkage [n].
 * Queries are channel-major as produced by transform_key: qk, qe [64][hw].
 *
 * Step 1 (deva_affinity_topk): grid = query tiles x `splits` token ranges; every wave filters the
 *   scores of its range for 32 queries against a running k-th-best threshold and hands over the
 *   surviving candidates (a superset of the range's top-k, at most 64 per query) as 64-bit keys
 *   (order-preserving score bits << 32 | ~token index) in part_keys [splits][hw][64], followed by the
 *   32-bit list lengths [splits][hw] (only the first `length` keys of a list are written).
 * Step 2 (deva_affinity_finalize): one wave per query selects the exact top-k over all ranges,
 *   w = exp(v)/sum exp(v) (NO max subtraction, memory_utils.py:59-60), writes idx/w [hw][k] sorted by
 *   descending score and, if usage_fix != NULL, adds w * 2^40 to usage_fix[token] (uint64 fixed
 *   point: integer atomics make the usage sum order-independent, hence deterministic).
 * Total order used for ties: higher score first, then lower token index.
 * Requires CK == 64, 1 <= k <= 32, n_long + n_work >= k, 1 <= splits <= 32. */
int deva_affinity_topk(const float* key_long, const float* shr_long, int n_long,
                       const float* key_work, const float* shr_work, int n_work,
                       const float* qk, const float* qe, int hw, int k, int splits,
                       uint64_t* part_keys, void* stream);
int deva_affinity_finalize(const uint64_t* part_keys, int hw, int k, int splits, int32_t* idx,
                           float* weight, uint64_t* usage_fix, void* stream);
/* Token-sharded bank (one shard of the memory per GPU, SURVEY.md 8e "shard the bank"):
 * deva_affinity_select turns a shard's candidate lists (output of deva_affinity_topk run on that shard's
 *   rows) into the shard's own sorted top-k in the same hand-over format -- out_keys [hw][64] (first k
 *   entries of each list live), out_counts [hw] -- with every token index shifted by token_offset (the
 *   shard's first row in the global long-then-work index space).
 * deva_affinity_merge is step 2 on explicit buffers: keys [lists][hw][64], counts [lists][hw] -- the
 *   all-gathered out_keys / out_counts of every shard.  Since each shard's top-k contains every member of
 *   the global top-k that lives on it, the merged result is bit-identical to the unsharded read. */
int deva_affinity_select(const uint64_t* part_keys, int hw, int k, int splits, int64_t token_offset,
                         uint64_t* out_keys, uint32_t* out_counts, void* stream);
int deva_affinity_merge(const uint64_t* keys, const uint32_t* counts, int hw, int k, int lists, int32_t* idx,
                        float* weight, uint64_t* usage_fix, void* stream);
/* Tuning / test hook: force one of the kernel shapes of deva_affinity_topk (1, 6: per-wave candidate lists, one / two
 * workgroups per CU with early key prefetch; 2: two per CU, late prefetch; 3: key tiles shared through LDS; 4, 5:
 * workgroup-shared lists, two / one 4-wave workgroup per CU; 7: ping-pong phases; 8: workgroup-shared lists, one 8-wave
 * workgroup per CU); 0 = automatic choice by frame and bank size (the default; the environment
 * variable DEVA_AFFINITY_SHAPE sets the initial value).  All shapes give bit-identical results.  The product library
 * carries shapes 2, 4 and 8 (the ones the automatic choice uses); 1, 3, 5, 6, 7 are A/B variants of `make PROBES=1`
 * builds and are refused otherwise.  Call between
 * deva_affinity_default_splits / deva_affinity_workspace / deva_affinity_topk sequences, not inside one. */
int deva_affinity_force_shape(int shape);
/* workspace query: number of uint64 elements part_keys must hold */
int64_t deva_affinity_workspace(int hw, int k, int splits);
/* splits the library would pick for a bank/query size (>= 1) */
int deva_affinity_default_splits(int n_total, int hw);

/* The same read for 32 < k <= 64 (eval_args.py:40 leaves --top_k free; the list kernels behind deva_affinity_topk /
 * deva_affinity_read hand over at most 32 entries per range): ONE kernel, lane = query, the bank streamed through
 * wave-uniform rows, scores by the same natural-order fp32 FMA chains (bit-identical to the other kernels, hence the same
 * selection wherever both apply), each query's k best kept in LDS, then exp / normalise / usage like
 * deva_affinity_finalize.  VALU-bound (~4 ms at 10 000 x 8 160): a correct path for a rare setting.  deva_affinity_read
 * routes k > 32 here (idx / weight outputs only: no hand-over format, so no token-sharded bank).  Also accepts k <= 32
 * (tests hold it against the list kernels bit for bit). */
int deva_affinity_dense(const float* key_long, const float* shr_long, int n_long, const float* key_work,
                        const float* shr_work, int n_work, const float* qk, const float* qe, int hw, int k,
                        int32_t* idx, float* weight, uint64_t* usage_fix, void* stream);
/* The whole read in one call, with the fp16 pre-filter where it pays (banks of >= 4 096 tokens AND >= 8 000 000
 * (token, query) scores: PF_MIN_TOKENS / PF_MIN_SCORES of csrc/affinity.hip; deva_affinity_prefilter_enabled tells):
 *   every (token, query) score is first bounded from both sides with v_mfma_f32_32x32x16_f16 on fp16 copies of the
 *   operands (1/16 of the fp32 matrix time) under a rigorous error bound, a filter threshold is derived from group
 *   maxima of the lower bounds, and only the tokens whose upper bound reaches it (~k + 5 per query) are re-scored with
 *   the natural-order fp32 FMA chain of deva_affinity_topk and selected exactly -- indices, weights, usage counters and
 *   shard keys are bit-identical to deva_affinity_topk + deva_affinity_finalize / deva_affinity_select.  Inputs the
 *   bound does not cover (negative selection, non-finite operands, flat "near-tie" banks that overflow the candidate
 *   lists) raise a device-side flag and the fp32 kernels produce the result in the same stream.
 * Output: either idx + weight [hw][k] (+ usage_fix), or out_keys [hw][64] + out_counts [hw] (the shard format of
 *   deva_affinity_select, token indices shifted by token_offset).  scratch: deva_affinity_read_scratch(...) uint64
 *   elements, private to the stream until the call's kernels have run.
 * deva_affinity_force_prefilter(0) routes every read to the fp32 kernels (A/B measurements, tests), 1 = automatic
 *   (default; the environment variable DEVA_AFFINITY_PREFILTER=0 sets the initial value to 0).
 * deva_affinity_read_flag: test hook -- copies the fall-back flag of the last read on `scratch` to the host
 *   (synchronises the stream): 0 = the pre-filter produced the result. */
int deva_affinity_read(const float* key_long, const float* shr_long, int n_long,
                       const float* key_work, const float* shr_work, int n_work,
                       const float* qk, const float* qe, int hw, int k, uint64_t* scratch,
                       int32_t* idx, float* weight, uint64_t* usage_fix,
                       uint64_t* out_keys, uint32_t* out_counts, int64_t token_offset, void* stream);
/* The same read with the bank side of the pre-filter kept between reads.  The reference re-derives nothing either while
 * the bank stands still: a bucket's bank changes only when a memory frame is added or consolidated
 * (deva/inference/memory_manager.py:171-218: every mem_every-th frame), the frames between read it unchanged
 * (memory_manager.py:91-169).  bank_prep: deva_affinity_bank_prep_bytes(n_long + n_work) bytes of device memory that the
 * CALLER keeps per bank (NULL: everything lives in `scratch`, like deva_affinity_read); the call writes the bank's mean
 * key, operand scales and fp16 MFMA fragments there.  bank_prep_valid != 0 is the caller's promise that bank_prep was
 * filled by an earlier call on the SAME bank contents (same rows in both segments, same n_long and n_work): the three
 * bank kernels (mean key, operand statistics, fragment preparation) are skipped.  A stale promise is a silent wrong
 * answer -- keep a version counter beside every bank (deva/inference/kv_memory_store.py: version()).  Reads that the
 * pre-filter does not serve (deva_affinity_prefilter_enabled == 0, k > 32) ignore both arguments.  Results are
 * bit-identical to deva_affinity_read. */
int deva_affinity_read_prepared(const float* key_long, const float* shr_long, int n_long,
                                const float* key_work, const float* shr_work, int n_work,
                                const float* qk, const float* qe, int hw, int k, uint64_t* scratch,
                                int32_t* idx, float* weight, uint64_t* usage_fix,
                                uint64_t* out_keys, uint32_t* out_counts, int64_t token_offset,
                                uint64_t* bank_prep, int bank_prep_valid, void* stream);
int64_t deva_affinity_bank_prep_bytes(int n_total);
int64_t deva_affinity_read_scratch(int n_total, int hw, int k);
int deva_affinity_prefilter_enabled(int n_total, int hw, int k);
int deva_affinity_force_prefilter(int mode);
int deva_affinity_read_flag(const uint64_t* scratch, void* stream);
/* test / tuning hook (synchronises): out[5] = {fall-back flag, largest candidate sub-list, largest candidate count of a
 * query, mean candidates per query x 1000, token ranges} of the last pre-filtered read on `scratch` */
int deva_affinity_read_stats(const uint64_t* scratch, int n_total, int hw, int k, int64_t* out, void* stream);

/* Usage counters of freshly appended tokens (kv_memory_store.py:93-95): use[i] = 0, life[i] = 1e-7 for i < count. */
int deva_usage_init(float* use, float* life, int count, void* stream);

/* KeyValueMemoryStore.update_bucket_usage (kv_memory_store.py:118-125) for one segment:
 * use[i] += usage_fix[offset+i] * 2^-40 (if use != NULL), life[i] += 1 (if life != NULL), and
 * usage_fix[offset .. offset+n) is zeroed for the next frame. */
int deva_usage_update(uint64_t* usage_fix, int64_t offset, float* use, float* life, int n,
                      void* stream);

/* MemoryManager._readout (memory_manager.py:64-75) in sparse form, one object per call:
 * out[c][q] = sum_j weight[q][j] * value(idx[q][j])[c], value rows token-major [n][cv] in the
 * same long-then-work index space.  out is [cv][hw] (an NCHW plane stack).  Only tokens in
 * [tok_lo, tok_hi) contribute (pass 0, INT_MAX for the whole bank): a bank shard adds its own terms
 * and the partial read-outs are summed over the shards.
 * map_long / map_work (NULL = identity): value-sharded storage -- the value arenas hold only the rows this rank
 * owns; map[token within the segment] = local row, < 0 = the row lives on another rank (its owner adds that term). */
int deva_readout_sparse(const int32_t* idx, const float* weight, int hw, int k,
                        const float* val_long, int n_long, const float* val_work, int cv,
                        float* out, int tok_lo, int tok_hi, const int32_t* map_long,
                        const int32_t* map_work, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bank maintenance (KeyValueMemoryStore.add / sieve_by_range / remove_obsolete_features,
 * kv_memory_store.py:35-185; MemoryManager.consolidation, memory_manager.py:251-276). */

/* append one frame: src channel-major [channels][count] -> rows dst_row0.. of a token-major arena */
int deva_bank_append(const float* src, float* arena, int64_t dst_row0, int channels, int count,
                     void* stream);
/* dst[i][:] = src[rows[i]][:] (rows = int32 device array, NULL = identity/contiguous copy) */
int deva_bank_gather_rows(const float* src, const int32_t* rows, float* dst, int count,
                          int channels, void* stream);
/* token-major rows -> channel-major [channels][count] (for API views of the bank) */
int deva_bank_export(const float* arena, float* dst, int channels, int count, void* stream);

/* rank[i] = position of x[i] in the order (descending ? larger first : smaller first), ties by
 * lower index first; a permutation of 0..n-1.  If use/life given (life != NULL) x = use/life
 * (KeyValueMemoryStore.get_usage, kv_memory_store.py:187-192) is computed on the fly and also
 * written to x_out when non-NULL. */
int deva_rank(const float* x, const float* life, float* x_out, int n, int descending,
              int32_t* rank, void* stream);
/* topk via ranks: out_idx[rank[i]] = i for rank[i] < k  (torch.topk(..., sorted=True)) */
int deva_rank_select(const int32_t* rank, int n, int k, int32_t* out_idx, void* stream);
/* eviction (kv_memory_store.py:170-174): with ascending ranks, thr = x at rank n_remove-1;
 * survivors = x > thr.  Writes survivor row indices in order to out_idx and their number to
 * out_count[0] (device int32). */
int deva_evict_select(const float* x, const int32_t* rank_asc, int n, int n_remove,
                      int32_t* out_idx, int32_t* out_count, void* stream);

/* consolidation, potentiation step: dense similarity of all candidates against the prototypes
 * (queries = prototype key + its stored selection, rows proto_idx of the candidate arrays):
 * sim[n*ld + p], token-major candidates key/sel [n_cand][64], shr [n_cand]. */
int deva_similarity_dense(const float* key, const float* shr, const float* sel,
                          const int32_t* proto_idx, int n_cand, int n_proto, int ld, float* sim,
                          void* stream);
/* do_softmax without top-k (memory_utils.py:67-70): softmax over rows n of each column of
 * x[n][ld] (first p columns), in place.  With ld = p rounded up to 32 and zero pad columns the
 * result is directly the packed "weight" of a 1x1 deva_conv2d, which performs the prototype
 * value / shrinkage readout (memory_manager.py:270-274) as a GEMM on the matrix cores. */
int deva_softmax_columns(float* x, int n, int p, int ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Detection merging (match_and_merge / merge_by_iou, deva/inference/segment_merging.py:17-143;
 * caller: DEVAInferenceCore.incorporate_detection, inference_core.py:137-198).
 *
 * deva_label_histogram: joint histogram of the propagated index mask `ours` (tmp ids 0..n_our, int64)
 *   and the detection index mask `news` (arbitrary int64 ids; column j = position in new_ids, column
 *   n_new = anything else): counts[t*(n_new+1)+j] += 1 per pixel (int32, must be zeroed by the caller).
 *   Every intersection / area / union of _get_iou (segment_merging.py:17-22) is an entry or a row /
 *   column sum of this matrix -- one device pass and one small copy instead of one sync per pair.
 *   Any number of label pairs: tables up to 60 KiB are accumulated per workgroup in LDS, larger ones
 *   with global integer atomics.
 * deva_merge_paint: replays the area-ordered repaint (segment_merging.py:62-85): source t (propagated)
 *   and source j (detection) each carry (order, label), order < 0 = not painted; a pixel takes the
 *   label of the source painted last; the result is written as one-hot planes out[o][pixel] =
 *   (label == out_ids[o])   (ObjectManager.make_one_hot, object_manager.py:133-141). */
int deva_label_histogram(const int64_t* ours, const int64_t* news, const int64_t* new_ids, int n_our,
                         int n_new, int64_t pixels, int32_t* counts, void* stream);
/* index-mask relabelling out[i] = lut[in[i]] (0 outside 0..n-1): ObjectManager.tmp_to_obj_cls
 * (object_manager.py:112-117, called in the timed region of evaluation/eval_vos.py:181) in one pass
 * instead of one masked assignment per object */
int deva_lut_remap(const int64_t* in, const int64_t* lut, int n, int64_t pixels, int64_t* out,
                   void* stream);
/* Output tail in one pass (evaluation/eval_vos.py:170-181 + result_utils.py:98-102 +
 * object_manager.py:112-117): bilinear resize of the [channels][height][width] probabilities to
 * (out_height, out_width) with F.interpolate(align_corners=False) arithmetic when the size differs,
 * argmax over channels (first maximum), then out = lut ? lut[argmax] (0 beyond n_lut) : argmax. */
int deva_index_mask(const float* prob, int channels, int height, int width, int out_height,
                    int out_width, const int64_t* lut, int n_lut, int64_t* out, void* stream);
int deva_merge_paint(const int64_t* ours, const int64_t* news, const int64_t* new_ids, int n_our,
                     int n_new, const int32_t* our_order, const int64_t* our_label,
                     const int32_t* new_order, const int64_t* new_label, const int64_t* out_ids,
                     int n_out, int64_t pixels, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEVA_HIP_H */
