/* svcmi.h -- C ABI of libsvcmi.so, the MI355X (gfx950) kernels of the SVC inference hot path.
 *
 * The reference (PlayVoice/whisper-vits-svc) has NO native/FFI interface: the path sits behind
 * Python callables.  This header is therefore the boundary a maintainer
 * would bind (ctypes stub in INTEGRATION.md).  Each entry point cites the reference code whose
 * arithmetic it replaces.  Conventions:
 *   - extern "C", plain pointers and sizes, no torch / C++ types, no exceptions, no hidden
 *     allocation; the caller owns every buffer and keeps it alive until the stream has drained.
 *   - every pointer is a DEVICE pointer (HBM); `stream` is a hipStream_t passed as void*.
 *   - activations are fp32, time-major / channels-last: element (b, t, c) of a tensor lives at
 *     base + b*bstride + t*ld + c.   (The reference is NCL [B,C,T]; the Python facade converts
 *     at the API edge.)  Leading dimensions are multiples of 4 floats and bases 16-byte aligned
 *     unless a function says otherwise.
 *   - `lengths` (int32[batch], may be NULL = all rows valid) carries the reference's x_mask
 *     (vits/commons.py:147-151): rows t >= lengths[b] are "masked".
 *   - return value: 0 = launched; <0 = svcmi_status (argument error, nothing launched);
 *     >0 = hipError_t from the launch.  Kernels are asynchronous on `stream`.
 */
#ifndef SVCMI_H
#define SVCMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVCMI_ABI_VERSION 22

enum svcmi_status { SVCMI_OK = 0, SVCMI_EINVAL = -1, SVCMI_EUNSUPPORTED = -2, SVCMI_EALIGN = -3 };

enum svcmi_act { SVCMI_ACT_NONE = 0, SVCMI_ACT_RELU = 1, SVCMI_ACT_GELU = 2, SVCMI_ACT_MISH = 3, SVCMI_ACT_TANH = 4, SVCMI_ACT_SIGMOID = 5 };

enum svcmi_conv_flags {
    SVCMI_CONV_ACCUMULATE = 1, /* y += result instead of y = result                          */
    SVCMI_CONV_MASK_IN = 2,    /* input rows >= lengths[b] read as zero   (conv(x * x_mask))  */
    SVCMI_CONV_MASK_OUT = 4,   /* output rows >= lengths[b] written as zero  (... * x_mask)   */
    SVCMI_CONV_PARTIALS = 8,   /* write exactly split_k raw partial slabs [batch][split_k][t_out][n_out] to `workspace` and stop:   */
                               /* no reduction, no epilogue, y untouched (consumer: svcmi_splitk_layernorm_f32)                     */
    /* tuning knob (bits 8-11): force the block tile (time x channels); 0 = library heuristic       */
    SVCMI_CONV_TILE_64x64 = 0x100,
    SVCMI_CONV_TILE_128x64 = 0x200,
    SVCMI_CONV_TILE_128x128 = 0x300,
    /* 16x16x4-MFMA tiles whose N spans the whole (narrow) output: n_out <= 48 / 80 / 160, c_in % 4 == 0 */
    SVCMI_CONV_TILE_P16_64x48 = 0x400,
    SVCMI_CONV_TILE_P16_128x48 = 0x500,
    SVCMI_CONV_TILE_P16_64x80 = 0x600,
    SVCMI_CONV_TILE_P16_128x80 = 0x700,
    SVCMI_CONV_TILE_P16_64x160 = 0x800,
    SVCMI_CONV_TILE_64x128 = 0x900, /* reduced-precision entry points only */
    /* ABI 22: the 64x80 wave tile on EIGHT-wave (512-thread) blocks: 128 x 80 per block = one block per CU at M = 500 x N = 5120 with two
     * waves per SIMD, 26 instead of 36 KB of LDS fill per CU and K-step; fp32 only, vector gathers; bit-identical to SVCMI_CONV_TILE_P16_64x80 */
    SVCMI_CONV_TILE_P16W8_128x80 = 0xA00,
    SVCMI_CONV_TILE_MASK = 0xF00,
    /* tuning knob: 2-deep operand ring instead of the 3-deep one of the 64-row fp32 tiles (64x64, P16 64x48 / 64x80): 33 / 41 KB of LDS
     * per block instead of 49 / 61 KB = one more resident block per CU and room for other streams' blocks beside them.  Slower for a
     * launch that has the chip to itself (one K-step of prefetch less), faster when several streams share it (clips in flight).  Results
     * are bit-identical (the ring depth does not touch the summation order).  Ignored by launches that have no such instantiation. */
    SVCMI_CONV_RING2 = 0x1000
};

int svcmi_abi_version(void);
/* "hip:gfx950" for the product build; the CPU SIMT emulator used by the unit tests says "emu". */
const char* svcmi_build_info(void);

/* ---------------------------------------------------------------------------------------------
 * 1-D convolution / linear layer as an implicit GEMM on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32, exact fp32).  Replaces every nn.Conv1d / nn.Linear / ConvTranspose1d
 * (polyphase-packed) on the path:
 *   whisper/model.py:34-45 (Linear, Conv1d), :144-163 (conv stem), :66-86,:118-129 (q/k/v/out, mlp);
 *   vits/models.py:26-37,44-46,49 (pre, hub, proj); vits/attentions.py:193-196 (conv_q/k/v/o),
 *   :390-403 (FFN); vits/modules.py:148-176,186,196 (WN in/res_skip), :273,281,285 (pre, post, snac);
 *   vits_decoder/generator.py:36-47 (adapter linears), :60,:69-99,:108 (conv_pre, ups, noise_convs,
 *   conv_post); vits_decoder/bigv.py:22-39 (AMP convs).
 *
 *   y[b,t,n] = epilogue( sum_{k<ksize} sum_{ci<c_in} x[b, (t*stride + k*dilation - pad) >> x_row_shift, ci]
 *                                                   * w[n, k*c_in + ci] )
 *   epilogue(v): v += bias[n]; v = act(v); v += res[b,t,n]; v *= alpha; (ACCUMULATE) v += y_old;
 *                (MASK_OUT) v = t < lengths[b] ? v : 0.
 * Input rows outside [0, t_in) (and, with MASK_IN, >= lengths[b]) read as zero.  x_row_shift = 1
 * fuses the np.repeat(ppg, 2, 0) of svc_inference.py:175-182 into the load (t_in is then the
 * repeated length).  w is [n_out][ldw] with ldw >= ksize*c_in, ldw % 4 == 0, zero padded.
 * c_in % 4 == 0 and ldx % 4 == 0 select 16-byte loads; any other c_in/ldx uses scalar loads.
 * y may alias res (in-place residual).  bias / res / lengths may be NULL.
 * Split-K: when the (time x channel) tile grid is too small to fill the GPU the K range can be cut into
 * `split_k` slices whose partial tiles go through `workspace` (caller-owned, >= batch*split_k*t_out*n_out
 * floats, any contents) and are summed in fixed slice order -- by the last block to finish each tile when `counters` is
 * given, by a second kernel otherwise -- deterministic either way (the only atomic is the arrival ticket).
 * split_k = 0 lets the library choose (never more than the workspace allows), 1 disables, workspace = NULL
 * disables.
 */
typedef struct svcmi_conv_desc {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    const int32_t* lengths;
    int64_t x_bstride, y_bstride, res_bstride; /* in floats */
    int32_t batch, t_in, t_out, c_in, ldx, n_out, ldw, ldy, ldr;
    int32_t ksize, stride, dilation, pad, x_row_shift;
    int32_t act;   /* enum svcmi_act */
    int32_t flags; /* enum svcmi_conv_flags */
    float alpha;
    int32_t split_k;
    float* workspace;
    int64_t workspace_floats;
    int32_t* counters;     /* optional: >= one int32 per output tile, ALL ZERO on entry (the library leaves them zero).  With it   */
    int64_t counters_len;  /* the slices are combined inside the GEMM launch by the last-arriving block of each tile (fixed slice */
                           /* order: still deterministic); without it a second kernel does the reduction.                         */
    void* y16;             /* optional: a 16-bit copy of y (same values rounded to y16_format = SVCMI_PREC_BF16 | SVCMI_PREC_F16 | _BF16X3 = split), element (b,t,n) */
    int64_t y16_bstride;   /* at y16 + b*y16_bstride + t*ldy16 + n (in 16-bit elements): the A operand of a following _A16 launch, written by    */
    int32_t ldy16;         /* THIS launch's epilogue instead of being rounded in that launch's registers.  Needs the float4 epilogue (n_out,    */
    int32_t y16_format;    /* ldy % 4 == 0, aligned operands), no split-K.                                                                       */
} svcmi_conv_desc;

int svcmi_conv_gemm_f32(const svcmi_conv_desc* d, void* stream);

/* Up to 3 convolutions of one geometry (same batch, t_out, n_out, c_in and gather alignment; taps, dilation, padding, operands
 * and epilogues free) in ONE launch -- the three AMP blocks of a generator stage at the same step (vits_decoder/generator.py:
 * 188-194 runs them one after the other).  The grid holds the blocks of all problems, longest K first, so medium-sized
 * problems fill the chip several blocks deep without multi-stream concurrency.  No split-K, no SVCMI_CONV_PARTIALS, no
 * x_row_shift; tile override bits of descs[0] apply to all.  Outputs of different problems must not overlap. */
int svcmi_conv_gemm_group_f32(const svcmi_conv_desc* descs, int32_t count, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reduced-precision variants of the same convolution: 16-bit MULTIPLICANDS, fp32 accumulation, fp32 activations in
 * HBM on both sides (x, res, y, bias, workspace exactly as above).  The reference drops to fp16 on an accelerator
 * (whisper/inference.py:22-23,43-44: model.half(), mel.half()); here the choice is per call and opt-in:
 *   SVCMI_PREC_BF16X3  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) on both operands, acc += hi*hi + lo*hi + hi*lo:
 *                      three bf16 MFMAs (16x the fp32 matrix rate each) for ~2^-17 relative error per product -- fp32-class
 *                      results (waveform within the 1e-3 parity bar, tests/test_gpu_precision.py) at up to ~5x the fp32 rate;
 *   SVCMI_PREC_BF16    operands rounded to bf16 (8-bit mantissa), one MFMA;
 *   SVCMI_PREC_F16     operands rounded to fp16 (11-bit mantissa, range 65504), one MFMA -- what `.half()` does.
 * Weights are converted ONCE by svcmi_pack_weights_lp from the fp32 operand [n][ldw] svcmi_conv_gemm_f32 takes to a 16-bit
 * image: row n = ldw16 values (K order permuted inside every block of 32 to match the kernel's fragment reads; bf16x3: hi row
 * followed by lo row, 2*ldw16 values per n); ldw16 % 32 == 0, ldw16 >= ldw, `out` 16-byte aligned and n*ldw16*2 (x2 for bf16x3)
 * bytes.  svcmi_conv_gemm_lp / svcmi_conv_gemm_group_lp take the SAME descriptor with d->w = that image and d->ldw = ldw16;
 * everything else (epilogue, masks, split-K, SVCMI_CONV_PARTIALS, grouping rules) is unchanged.  Activations are rounded in
 * registers inside the kernel, so no other kernel of the path changes.  Not supported (SVCMI_EUNSUPPORTED): per-element
 * gathers (c_in % 4 != 0 or unaligned x) and the in-launch split-K combine (`counters` is ignored). */
enum svcmi_precision { SVCMI_PREC_F32 = 0, SVCMI_PREC_BF16X3 = 1, SVCMI_PREC_BF16 = 2, SVCMI_PREC_F16 = 3,
                       /* 16-bit ACTIVATIONS as well: d->x is a bf16 / fp16 tensor (ldx, x_bstride in 16-bit elements; c_in, ldx, x_bstride % 8 == 0, x 16-byte
                        * aligned, no x_row_shift) that the producing kernel wrote (y16 of a convolution, svcmi_splitk_layernorm_f32, svcmi_layernorm_f32,
                        * svcmi_attention_f32, svcmi_snake_alias_group_f32), so nothing is rounded in this launch's registers and the operand tile in LDS is
                        * half the size.  Same products as SVCMI_PREC_BF16 / _F16 (the same fp32 values, rounded the same way); the weight image is packed
                        * by svcmi_pack_weights_lp with the _A16 code (natural k order instead of the permuted order of the other modes).
                        * SVCMI_PREC_BF16X3_A16: the split-bf16 products of SVCMI_PREC_BF16X3 on 16-bit activations: a row of d->x is [hi: ldx/2 values |
                        * lo: ldx/2 values] (what the producers write for y16_format = SVCMI_PREC_BF16X3; ldx % 16 == 0, c_in <= ldx/2), the weight image
                        * [hi row | lo row] in natural k order. */
                       SVCMI_PREC_BF16_A16 = 4, SVCMI_PREC_F16_A16 = 5, SVCMI_PREC_BF16X3_A16 = 6,
                       /* model-level only (svcmi_synth_model.precision): every layer CLASS of the synthesizer runs in its own mode,
                        * svcmi_synth_model.class_prec[] -- the per-layer mixed policy that keeps the 16-bit waveform error inside the parity bar */
                       SVCMI_PREC_MIXED = 7,
                       /* fp16 activations x SPLIT fp16 weights (round 4): a mode of a model / layer class that behaves like SVCMI_PREC_F16 (fp16 activation
                        * rows from the producers, fp16 attention) except that a convolution reading 16-bit activations runs as SVCMI_PREC_F16W2_A16: the
                        * static weight w = hi + lo with hi = fp16(w), lo = fp16(w - hi) (exact to 2^-22; the matrix cores honour the subnormal lo values:
                        * scripts/probes/mfma_f16_subnormal.hip), acc += a*hi + a*lo -- two MFMAs per fragment pair, the error of the product is the
                        * ACTIVATION rounding alone (0.6-0.7x of plain fp16, scripts/precision_sensitivity.py --operands a).  Weight image of the _A16 code:
                        * rows [hi: ldw16 | lo: ldw16] in natural k order; the non-_A16 launches of such a layer use the plain SVCMI_PREC_F16 image. */
                       SVCMI_PREC_F16W2 = 8, SVCMI_PREC_F16W2_A16 = 9 };
int svcmi_pack_weights_lp(const float* w, int32_t n, int32_t ldw, int32_t precision, void* out, int32_t ldw16, void* stream);
int svcmi_conv_gemm_lp(const svcmi_conv_desc* d, int32_t precision, void* stream);
int svcmi_conv_gemm_group_lp(const svcmi_conv_desc* descs, int32_t count, int32_t precision, void* stream);

/* LayerNorm over the channel dim of time-major rows, optional pre-add:
 *   y[r,:] = (v - mean(v)) / sqrt(var(v) + eps) * gamma + beta,   v = x[r,:] + (res ? res[r,:] : 0)
 * rows = batch*rows_per_batch contiguous rows of stride ldx/ldr/ldy.  gamma/beta (NULL = 1/0) are
 * indexed [ (r / rows_per_batch) * gb_bstride + c ] so a per-utterance affine can be applied.
 * Replaces whisper/model.py:28-31 (attn_ln, mlp_ln, ln_post), vits/modules.py:19-22 fused with the
 * `x + y` of vits/attentions.py:66,70, and the speaker-conditioned LN of
 * vits_decoder/generator.py:36-47 (gamma/beta = W_scale(spk)/W_bias(spk), gb_bstride = c).
 * c % 4 == 0, c <= 2048. */
int svcmi_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                        int32_t batch, int32_t rows_per_batch, int32_t c, int32_t ldx, int32_t ldr, int32_t ldy,
                        int32_t gb_bstride, float eps, void* y16, int32_t ldy16, int32_t y16_format, void* stream);
/* (y16 / ldy16 / y16_format here and below: an optional second output, the same rows rounded to bf16 / fp16 (SVCMI_PREC_BF16 |
 * SVCMI_PREC_F16) with leading dimension ldy16 in 16-bit elements -- the A operand of a following SVCMI_PREC_*_A16 convolution.
 * SVCMI_PREC_BF16X3: split rows, hi = bf16(v) at [c] and lo = bf16(v - hi) at [ldy16/2 + c] (ldy16 % 8 == 0, ldy16/2 >= the row
 * length) for SVCMI_PREC_BF16X3_A16.  NULL = none.) */

/* Per-channel normalisation over time + GELU: GroupNorm(num_groups = c, c) followed by exact-erf GELU, the first layer of
 * HuBERT's feature extractor (hubert/hubert_model.py:78,88): y[b,t,ch] = gelu((x[b,t,ch] - mean_t) / sqrt(var_t + eps) *
 * gamma[ch] + beta[ch]) with mean / biased variance over the t axis of batch item b.  x, y: [batch][t][ld] time-major
 * (y may alias x); scratch: >= batch * 129 * c doubles.  c % 4 == 0. */
int svcmi_channel_norm_gelu_f32(const float* x, const float* gamma, const float* beta, float* y, double* scratch,
                                int32_t batch, int32_t t, int32_t c, int32_t ldx, int32_t ldy, float eps, void* stream);

/* Split-K tail fused with the residual update and the LayerNorm that follows it (whisper/model.py:118-129:
 * `x = x + attn(...)` / `x = x + mlp(...)` then the next `ln(x)`):
 *   x[r,:] += bias + sum_{s<split} partials[b, s, t, :]   (slice order fixed);   y[r,:] = LayerNorm(x[r,:]) * gamma + beta
 * partials: the slabs a SVCMI_CONV_PARTIALS launch of svcmi_conv_gemm_f32 left in its workspace ([batch][split][t][c]).
 * rows r = b*rows_per_batch + t.  c % 4 == 0, c <= 2048. */
int svcmi_splitk_layernorm_f32(const float* partials, int32_t split, const float* bias, float* x, const float* gamma,
                               const float* beta, float* y, int32_t batch, int32_t rows_per_batch, int32_t c,
                               int32_t ldx, int32_t ldy, float eps, void* y16, int32_t ldy16, int32_t y16_format, void* stream);

/* Multi-head self-attention with exact (fp32, online) softmax.
 *   S[i,j] = scale * ( q_i . k_j  +  (|j-i| <= window ? q_i . rel_k[j-i+window] : 0) )
 *   S[i,j] = -1e4 where i >= lengths[b] or j >= lengths[b]           (masked_fill, attentions.py:249)
 *   P = softmax_j(S);  o_i = sum_j P[i,j] v_j  +  sum_{|j-i|<=window} P[i,j] rel_v[j-i+window]
 * q/k/v/o: element (b, t, h, d) at base + b*bstride + t*ld + h*head_dim + d (so q,k,v can point into
 * one fused [T, 3C] projection).  rel_k/rel_v: [2*window+1][head_dim] shared by heads, or NULL
 * (then window is ignored).  head_dim in {16, 32, 64, 96}.
 * Replaces whisper/model.py:88-101 (scale = head_dim^-0.5, no rel, no mask) and
 * vits/attentions.py:225-274 incl. the relative-position skew helpers :276-347. */
int svcmi_attention_f32(const float* q, const float* k, const float* v, float* o,
                        int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                        int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                        int32_t batch, int32_t t, int32_t heads, int32_t head_dim, float scale,
                        const float* rel_k, const float* rel_v, int32_t window,
                        const int32_t* lengths, void* o16, int32_t ldo16, int64_t o16_bstride, int32_t o16_format, void* stream);

/* The same attention on the 16-bit matrix cores, for the bf16 / f16 modes (rel_k / rel_v / window as above, fp32 tables, or NULL): q / k / v are 16-bit tensors
 * (`format` = SVCMI_PREC_BF16 | SVCMI_PREC_F16; element (b, t, h, d) at base + b*bstride16 + t*ld16 + h*head_dim + d, in 16-bit elements --
 * the y16 copy of the fused QKV projection), products on v_mfma_f32_16x16x32, softmax statistics / rescale / accumulation in fp32, P
 * rounded to `format` before the PV product.  K / V tiles of a head are staged once per block through LDS (V transposed on the way in).
 * Outputs: o (fp32, may be NULL) and / or o16 (16-bit copy in `format`, may be NULL): the out-projection's A operand.  head_dim in {32, 64}
 * without the band, {32, 96} with it.  Replaces whisper/model.py:88-101 under `.half()` (whisper/inference.py:22-23) and, in the 16-bit
 * modes of the synthesizer, vits/attentions.py:225-274. */
int svcmi_attention16(const void* q, const void* k, const void* v, int32_t ld16, int64_t bstride16, float* o, int32_t ldo,
                      int64_t o_bstride, void* o16, int32_t ldo16, int64_t o16_bstride, int32_t batch, int32_t t, int32_t heads,
                      int32_t head_dim, float scale, const float* rel_k, const float* rel_v, int32_t window,
                      const int32_t* lengths, int32_t format, void* stream);

/* Anti-aliased SnakeBeta (vits_decoder/alias/act.py:124-129): 2x Kaiser-sinc polyphase upsample with
 * replicate padding (resample.py:25-33), x + sin^2(x*e^alpha)/(e^beta + 1e-9) (act.py:79-92), 12-tap
 * low-pass + 2x decimation (filter.py:86-95).  x,y: [batch][len][ld] time-major, c <= ld channels;
 * alpha_log/beta_log: [c]; filt: the 12 taps (filter.py:28-57).  SURVEY.md A.5.  One batch item is addressed with 32-bit byte
 * offsets: len * ld * 4 >= 2^31 returns SVCMI_EUNSUPPORTED (a 32 kHz x 10-channel tensor of that size is 1.5 hours of audio). */
int svcmi_snake_alias_f32(const float* x, float* y, const float* alpha_log, const float* beta_log,
                          const float* filt, int32_t batch, int32_t len, int32_t c, int32_t ld, void* stream);
/* The same activation for up to 3 tensors of one shape in one launch (x[i] -> y[i] with alpha_log[i] / beta_log[i]): the AMP
 * blocks of a generator stage at the same step.  The pointer arrays are host arrays read during the call. */
int svcmi_snake_alias_group_f32(const float* const* x, float* const* y, const float* const* alpha_log,
                                const float* const* beta_log, const float* filt, int32_t count, int32_t batch,
                                int32_t len, int32_t c, int32_t ld, void* const* y16, int32_t y16_format, void* stream);
/* (y16 != NULL: tensor i is written as bf16 / fp16 rows (y16_format, same ld) to y16[i] INSTEAD of fp32 to y[i] -- the activation only
 * feeds the following convolution, which then runs as an SVCMI_PREC_*_A16 launch; y / y[i] may be NULL then.  y16_format =
 * SVCMI_PREC_BF16X3: split rows of 2*ld values, [hi: ld | lo: ld].) */
/* y[i] = ((xs[0][i] + xs[1][i]) + xs[2][i]) / count, count <= 3: the `xs / num_kernels` of vits_decoder/generator.py:188-194
 * when the AMP blocks ran side by side in grouped launches.  n % 4 == 0, 16-byte aligned pointers; y may alias xs[0]. */
int svcmi_block_mean_f32(const float* const* xs, int32_t count, float* y, int64_t n, void* stream);

/* Fused AMP half-step for the narrow generator stages (vits_decoder/bigv.py:50-58: `xt = act(x); xt = conv(xt)`):
 *   y[b,t,n] = alpha * ( bias[n] + sum_{k<ksize} sum_{ci<c} w[n, k*ld + ci] * S[b, t + (k - (ksize-1)/2)*dilation, ci]
 *                        + res[b,t,n] )  (+ y_old[b,t,n] when accumulate != 0),      n < c
 * with S = SnakeAlias(x) exactly as svcmi_snake_alias_f32 computes it and S rows outside [0, len) read as zero
 * (the convolution's 'same' zero padding, stride 1).  x, y, res: [batch][len][ld] time-major; channels c..ld-1
 * of y are written as zero.  w: [ld][ldw] packed like svcmi_conv_gemm_f32's (tap-major, ld channels per tap).
 * res may alias y; x must not.  Supported shapes: (c, ld) in {(10,12), (20,20), (40,40)}, ksize in {3,7,11},
 * dilation 1..5 -- svcmi_snake_conv_supported() tells; other shapes return SVCMI_EUNSUPPORTED and the caller
 * uses svcmi_snake_alias_f32 + svcmi_conv_gemm_f32. */
int svcmi_snake_conv_supported(int32_t c, int32_t ld, int32_t ksize, int32_t dilation);
/* 1 where the fused kernel is also the FASTER choice on MI355X (narrowest stages); the facade follows it. */
int svcmi_snake_conv_preferred(int32_t c, int32_t ld, int32_t ksize, int32_t dilation);
int svcmi_snake_conv_f32(const float* x, const float* w, const float* bias, const float* res, float* y,
                         const float* alpha_log, const float* beta_log, const float* filt,
                         int32_t batch, int32_t len, int32_t c, int32_t ld, int32_t ldw, int32_t ksize,
                         int32_t dilation, float alpha, int32_t accumulate, void* stream);

/* The same half-step for up to 3 AMP blocks of a stage in one launch (own taps / dilation / weights / activation
 * parameters / tensors; shared shape batch x len x ld and channel count c).  Problems are run most-taps-first.  Outputs of
 * different problems must not overlap.  Same support matrix as svcmi_snake_conv_f32. */
typedef struct svcmi_snake_conv_desc {
    const float* x; const float* w; const float* bias; const float* res; float* y;
    const float* alpha_log; const float* beta_log;
    int32_t ldw, ksize, dilation, accumulate;
    float alpha;
} svcmi_snake_conv_desc;
int svcmi_snake_conv_group_f32(const svcmi_snake_conv_desc* descs, int32_t count, const float* filt, int32_t batch,
                               int32_t len, int32_t c, int32_t ld, void* stream);

/* The grouped half-step with its convolution on the fp16 matrix cores -- the narrow stages' member of the per-layer mixed-precision
 * policy (classes amp3 / amp4).  `precision`: SVCMI_PREC_F16 (weights rounded to fp16, one MFMA per tile) or SVCMI_PREC_F16W2 (weights as
 * hi + lo fp16, two MFMAs: only the activated input is rounded).  x / res / y / bias / weights are the SAME fp32 tensors the _f32 entry
 * takes (the block converts its problem's weights to MFMA fragment order in LDS); accumulation, epilogue and SnakeAlias are fp32.
 * 10 and 20 channels; y, res and bias 16-byte aligned. */
int svcmi_snake_conv_lp_supported(int32_t c, int32_t ld, int32_t ksize, int32_t dilation, int32_t precision);
int svcmi_snake_conv_group_lp(const svcmi_snake_conv_desc* descs, int32_t count, const float* filt, int32_t batch,
                              int32_t len, int32_t c, int32_t ld, int32_t precision, void* stream);

/* Stage entry of the narrow generator stages in one launch (vits_decoder/generator.py:183-186):
 *   y[b, u*q + r, co] = b_up[r*cp+co] + sum_{k<taps} sum_{ci<c_in} x[b, q + k - pad, ci] * w_up[r*cp+co, k*c_in + ci]      (ups[i], polyphase)
 *                     + b_nz[co] + sum_{k<nz_k} src[b, (u*q+r)*nz_stride - nz_pad + k] * w_nz[co, k]                       (noise_convs[i])
 * x: [batch][t_in][c_in]; w_up / b_up: the polyphase operands svcmi_conv_gemm_f32 would take ([u*cp][ldw_up], rows outside
 * the tensor read as zero); src: [batch][src_len]; y: [batch][t_in*u][cp].  x == NULL selects the noise-only mode: y already
 * holds ups[i](x) and only the second line is added (20-channel stage: its transposed convolution is faster on the
 * matrix cores).  Supported: u == 2, cp in {12, 20}, c_in % 4 == 0
 * (svcmi_upsample_noise_supported); other shapes use two svcmi_conv_gemm_f32 launches. */
int svcmi_upsample_noise_supported(int32_t u, int32_t cp, int32_t c_in);
int svcmi_upsample_noise_f32(const float* x, const float* w_up, const float* b_up, const float* src, const float* w_nz,
                             const float* b_nz, float* y, int32_t batch, int32_t t_in, int32_t c_in, int32_t ldw_up,
                             int32_t taps, int32_t pad, int32_t u, int32_t cp, int64_t src_len, int32_t nz_k,
                             int32_t nz_stride, int32_t nz_pad, int32_t ldw_nz, void* stream);

/* Output layer of the generator in one launch (vits_decoder/generator.py:196-199: activation_post, conv_post, tanh):
 *   y[b, t] = tanh( sum_{k < ksize} sum_{ci < c} w[k*ld + ci] * S[b, t + k - (ksize-1)/2, ci] ),  S = SnakeAlias(x) zero-padded
 * x: [batch][len][ld] time-major, w: conv_post.weight packed as one row (tap-major, ld floats per tap; no bias),
 * y: [batch][len].  Supported: c == 10, ld == 12, ksize == 7 (svcmi_snake_post_supported); other widths use
 * svcmi_snake_alias_f32 + svcmi_conv_gemm_f32. */
int svcmi_snake_post_supported(int32_t c, int32_t ld, int32_t ksize);
int svcmi_snake_post_f32(const float* x, const float* w, float* y, const float* alpha_log, const float* beta_log,
                         const float* filt, int32_t batch, int32_t len, int32_t c, int32_t ld, int32_t ksize, void* stream);

/* Development knobs for the tuning scripts (kernel SHAPE choices only: results never depend on them beyond fp32 re-association):
 *   "amp_tt"   {0 = default, 1, 2, 4}   time steps per thread of svcmi_snake_conv_f32
 *   "amp_u"    {0 = measured choice, -1 = never, 1 = wherever it exists}  grouped fused narrow-stage kernel with the up-sampled activation tile in LDS
 *   "amp_grouped" {1 = default, 0}      0 forces the one-launch-per-AMP-block fallback of the generator stages (stage host)
 *   "group_nst" {0 = default, 2, 3}     LDS ring depth of the grouped implicit-GEMM launches
 *   "attn_ns"  {0 = heuristic, 1, 2, 4, 8}  key-split waves per block of svcmi_attention_f32
 *   "attn_q32" {-1 = heuristic, 0, 1}   two query tiles per wave (band-free attention, head_dim <= 64)
 *   "attn16"   {0 = heuristic, 41, 42, 44, 81, 82; with the band: 14, 21, 24, 42, 44}  block shape (10 * query tiles + key ranges) of svcmi_attention16
 *   "attn_lds" {0 = heuristic, -1 = never, 1, 10*QT+KS}  band-free attention with K / V tiles staged through LDS and shared by QT in
 *              {2, 4, 8} query tiles x KS in {1, 2, 4} key ranges per block (1 = 4 x 2; compiled shapes 21 22 24 41 42 44 81 82)
 * Process-wide, not thread-safe against concurrent launches.  Returns 0, or SVCMI_EINVAL for an unknown name / value. */
int svcmi_tune_set(const char* name, int32_t value);
/* ABI 22: the current value of "ring2" / "amp_grouped" / "amp_lp" / "group_nst", and "last_conv_ring" = which instantiation the last
 * svcmi_conv_gemm_f32 launch took: 2 = the SVCMI_CONV_RING2 one, 3 = the tile's default (tests assert which kernel a flag selected).  0, or SVCMI_EINVAL. */
int svcmi_tune_get(const char* name, int32_t* value);

/* WaveNet gate, vits/commons.py:126-133 with input_b == 0 (vits/modules.py:190-193):
 *   out[b, t, c] = tanh(v[b, t, c]) * sigmoid(v[b, t, h + c]),  c < h,  v = bias + sum_s a[b][s][t][:]
 * a: `splits` slabs per batch item, [batch][splits][t][lda] -- the raw split-K partials of the in_layer convolution
 * (SVCMI_CONV_PARTIALS, lda == 2h) summed here in slab order with the layer bias, so that the reduce pass, the bias and the
 * gate are one launch; or the finished activations with splits == 1 (bias may be NULL). */
int svcmi_wn_gate_f32(const float* a, const float* bias, float* out, int32_t batch, int32_t t, int32_t h, int32_t lda,
                      int32_t ldo, int32_t splits, void* stream);

/* WaveNet residual/skip bookkeeping, vits/modules.py:196-203, from rs = res_skip_layer(acts):
 *   !last: x = (x + rs[:, :h]) * mask ; skip (+)= rs[:, h:2h]        last: skip (+)= rs[:, :h]; skip *= mask
 * `first` makes skip = instead of += (output = zeros_like(x) at :179). */
int svcmi_wn_update_f32(const float* rs, float* x, float* skip, const int32_t* lengths,
                        int32_t batch, int32_t t, int32_t h, int32_t ldrs, int32_t first, int32_t last, void* stream);

/* Speaker-normalised coupling, reverse direction (vits/modules.py:288-321 with mean_only).
 * ms_vs: [batch][2*half] = snac(spk) laid out (m_s | v_s), already permuted for flipped layers.
 * pre :  out[b,t,c] = (x[b,t,x0_off+c] - m_s[c]) * exp(-v_s[c]) * mask
 * post:  x1 = x[b,t,x1_off+c];  x1 = (x1 - m[b,t,c]) * mask;  x[b,t,x1_off+c] = (m_s[c] + x1*exp(v_s[c])) * mask
 * torch.flip (modules.py:225-229) is folded into the x0/x1 offsets and weight permutations at load. */
int svcmi_coupling_pre_f32(const float* x, int32_t ldx, int32_t x0_off, const float* ms_vs, float* out, int32_t ldo,
                           const int32_t* lengths, int32_t batch, int32_t t, int32_t half, void* stream);
int svcmi_coupling_post_f32(float* x, int32_t ldx, int32_t x1_off, const float* m, int32_t ldm, const float* ms_vs,
                            const int32_t* lengths, int32_t batch, int32_t t, int32_t half, void* stream);

/* x[b,t,:] = (x[b,t,:] + emb[f0_to_coarse(pit[b,t]), :]) * mask   -- vits/utils.py:20-33 (Hz -> bin 1..255)
 * + nn.Embedding lookup and the sum/mask of vits/models.py:44-48.  emb: [256][c]. */
int svcmi_embed_pitch_f32(float* x, int32_t ldx, const float* pit, const float* emb, const int32_t* lengths,
                          int32_t batch, int32_t t, int32_t c, void* stream);

/* z[b,t,c] = (m + noise[b,c,t] * exp(logs)) * mask with (m | logs) = stats[b,t,0:i | i:2i]
 * (vits/models.py:49-51).  noise is the reference's randn_like(m) in ITS layout [batch][i][t]. */
int svcmi_sample_prior_f32(const float* stats, int32_t lds, const float* noise_ncl, const int32_t* lengths,
                           float* z, int32_t ldz, int32_t batch, int32_t t, int32_t i, void* stream);

/* Layout bridges at the API edge.  ncl_to_nlc: y[b,t,c] = x[b,c,t] + add_scale*add[b,c,t] (add may be NULL;
 * used for `mel + 0.1*randn`, whisper/inference.py:46).  nlc_to_ncl: y[b,c,t] = x[b,t,c]. */
int svcmi_ncl_to_nlc_f32(const float* x, const float* add, float add_scale, float* y,
                         int32_t batch, int32_t c, int32_t t, int32_t ldy, void* stream);
int svcmi_nlc_to_ncl_f32(const float* x, int32_t ldx, float* y, int32_t batch, int32_t c, int32_t t, void* stream);

/* Harmonic source (vits_decoder/generator.py:160-165 -> nsf.py:223-253,284-316,383-394).
 * Step 1: per-frame phase prefix.  rad[f,k] = fmodf(f0[b,f]*(k+1)/sr, 1) in fp32 exactly as :228; the sample-level
 *   double cumsum of :246-253 collapses (F0 is piecewise constant over a hop) to
 *   prefix[b,f,k] = frac(rand_ini[b,k] + hop * sum_{f'<f} rad[f',k]) accumulated in fp64.  rand_ini column 0 is
 *   treated as 0 (:235).
 * Step 2: out[b, f*hop+j] = tanh(sum_k merge_w[k]*(0.1*sin(2*pi*(prefix + (j+1)*rad))*uv + amp*noise[b,t,k]) + merge_b),
 *   uv = f0 > 0, amp = uv ? 0.003 : 0.1/3 (:305-314); noise: [batch][t*hop][11] as drawn by randn_like(:311). */
int svcmi_pitch_prefix_f64(const float* f0, const float* rand_ini, double* prefix,
                           int32_t batch, int32_t t, int32_t hop, float sr, void* stream);
int svcmi_pitch_source_f32(const float* f0, const double* prefix, const float* noise, const float* merge_w,
                           float merge_b, float* out, int32_t batch, int32_t t, int32_t hop, float sr, void* stream);

/* Whisper log-mel front-end (whisper/audio.py:68-100), the glue around two svcmi_conv_gemm_f32 launches (the windowed DFT
 * is a stride-160 / 400-tap / 1-channel convolution with a [hann*cos | -hann*sin] basis, the mel projection a linear
 * layer):
 *   reflect_pad:     y[b, i] = x[b, reflect(i - pad)], i < n + 2*pad          (torch.stft center=True, pad_mode="reflect")
 *   power_spectrum:  p[r, f] = ri[r, f]^2 + ri[r, half + f]^2, f < nbins; p[r, nbins..ldp) = 0
 *   logmel_finish:   v = log10(max(mel_power, 1e-10)); v = max(v, max_b(v) - 8); out[b, c, t] = (v[b, t, c] + 4) / 4
 *                    mel_power [batch][t][c] is overwritten; scratch >= 64*batch floats; out is NCL like the reference. */
int svcmi_reflect_pad_f32(const float* x, float* y, int32_t batch, int64_t n, int32_t pad, void* stream);
int svcmi_power_spectrum_f32(const float* ri, float* p, int64_t rows, int32_t nbins, int32_t half, int32_t ldri, int32_t ldp, void* stream);
int svcmi_logmel_finish_f32(float* mel_power, float* scratch, float* out, int32_t batch, int32_t t, int32_t c, void* stream);

/* CREPE F0 extractor glue (row N3; the six convolutions and the classifier are svcmi_conv_gemm_f32 launches):
 *   crepe_frames: crepe/core.py:664-703 -- frame f = samples [f*hop - 512, f*hop + 512) of the waveform (zeros outside
 *                 [0, n)), minus its mean, divided by max(1e-10, unbiased std).  Written as rows of `ld` (>= 1532, % 4 == 0)
 *                 floats: 254 zeros, the 1024 samples, zeros -- the zero padding of the first convolution (model.py:119),
 *                 placed so that its stride-4 / 512-tap window reads 16-byte aligned groups of 4 samples, i.e. the first
 *                 layer is a stride-1, 128-tap convolution over rows of 4 "channels".
 *   bn_maxpool2:  y[r, c] = max(x[2r, c], x[2r+1, c] as a' = a*scale[c] + shift[c]) -- eval-mode BatchNorm2d applied after
 *                 the ReLU, then max_pool2d((2,1)) (model.py:128-134); rows_out = rows_in / 2, rows never straddle frames
 *                 because every per-frame length is even. */
int svcmi_crepe_frames_f32(const float* audio, int64_t n, int32_t hop, int32_t frame0, int32_t frames, float* out, int32_t ld, void* stream);
int svcmi_bn_maxpool2_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows_out, int32_t c,
                          int32_t ldx, int32_t ldy, void* y16, int32_t ldy16, int32_t y16_format, void* stream);
/* (y16: optional 16-bit copy of the pooled rows for the next layer's SVCMI_PREC_*_A16 convolution, formats as for svcmi_layernorm_f32;
 * y may be NULL when y16 is given.) */
/* Viterbi decoding of the 360-bin pitch posteriorgram (crepe/decode.py:53-80; librosa.sequence.viterbi semantics: uniform
 * prior, log domain): prob [frames][360] = the network's sigmoid outputs; bins outside [minidx, maxidx) are excluded
 * (crepe/core.py:597-598); softmax over the rest in fp32, DP in fp64, independently per batch of `batch_frames` frames
 * (crepe/core.py:683-686).  log_trans: [360][360] doubles = log(transition + tiny).  band > 0 declares the matrix banded:
 * every entry with |i - j| > band equals log_trans[0][359] (CREPE's transition: band = 11); the DP then evaluates 2*band + 1
 * predecessors plus the best out-of-band one per state instead of all 360 (same result, ties to the lowest index).  band = 0:
 * dense.  lp_scratch: frames*360 floats, ptr_scratch: frames*360 int16.  path: [frames] decoded bins. */
int svcmi_viterbi_decode(const float* prob, const double* log_trans, float* lp_scratch, int16_t* ptr_scratch, int32_t* path,
                         int32_t frames, int32_t batch_frames, int32_t minidx, int32_t maxidx, int32_t band, void* stream);

/* Feature retrieval (row N4; feature_retrieval/index.py:57-94 -- faiss search_and_reconstruct + the RVC weighting):
 *   row_sqnorm: out[r] = sum_c x[r, c]^2 (the |b|^2 term of the bank, computed once per index).
 *   knn_blend:  for every query row i < t: the k (<= 8) stored vectors with the smallest squared L2 distance, ranked by
 *               |x|^2 + bank_sq[j] - 2*dots[i, j] with dots = X * Bank^T from svcmi_conv_gemm_f32 (ties -> smaller j);
 *               the 8 best-ranked candidates are re-measured as sum_c (x - b)^2 and the k nearest of those kept (so fp32
 *               cancellation in the ranking cannot swap near-equidistant neighbours), weight_q = (1/dist_q)^2 / sum_q' (1/dist_q')^2, and
 *               out[i, :] = (1 - ratio) * x[i, :] + ratio * sum_q weight_q * bank[nn_q, :].
 *               An exact search, i.e. faiss IVF-Flat with nprobe = nlist (the reference's nprobe = 1 is its approximation). */
int svcmi_row_sqnorm_f32(const float* x, int32_t ldx, int64_t rows, int32_t d, float* out, void* stream);
int svcmi_knn_blend_f32(const float* x, int32_t ldx, const float* bank, int32_t ldb, const float* dots, int64_t ldd,
                        const float* bank_sq, float* out, int32_t ldo, int32_t t, int32_t n, int32_t d, int32_t k,
                        float ratio, void* stream);

/* The reference's own index type: faiss IVF-Flat searched with nprobe = 1 (feature_retrieval/index.py:145-151 builds
 * index_factory("IVF{n},Flat", METRIC_L2) and sets nprobe = 1; :57-62 search_and_reconstruct; :163-166 read_index).  faiss 1.7.4 is
 * pinned by the reference's requirements.txt and not vendored; its published algorithm is restated (parity unpinned: no faiss here):
 *   ivf_assign:   coarse quantizer = IndexFlatL2 over the nlist centroids: assign[i] = argmin_j max(0, |x_i|^2 + cent_sq[j] -
 *                 2*dots[i, j]) with dots = X * Centroids^T from svcmi_conv_gemm_f32 (ties -> smaller j); dist (optional) = that
 *                 minimum.  Also the assignment step of the k-means that trains the centroids (Clustering.cpp).
 *   ivf_blend:    the inverted list of cell assign[i] = rows [list_off[a], list_off[a+1]) of `bank` (stored vectors grouped by cell,
 *                 insertion order inside a cell) is scanned with exact distances sum_c (x - b)^2; the k (<= 8) nearest, ascending (ties ->
 *                 earlier row), get weight (1/dist)^2 normalised and out[i] = (1 - ratio) * x[i] + ratio * sum_q weight_q * bank[nn_q].
 *                 A cell with fewer than k vectors uses the ones it has, an empty cell leaves x[i] unchanged (faiss pads with label -1
 *                 and a NaN reconstruction there, which makes the reference emit a NaN frame).  out_idx [t][k] (optional): the chosen
 *                 rows of `bank`, -1 padded; out_dist [t][k] (optional): their distances, +inf padded.
 *   segment_mean: k-means centroid update: out[c] = mean of x[order[r]], r in [seg_off[c], seg_off[c+1]), summed in that order;
 *                 an empty segment leaves out[c] untouched. */
int svcmi_ivf_assign_f32(const float* x, int32_t ldx, const float* dots, int64_t ldd, const float* cent_sq, int32_t t,
                         int32_t nlist, int32_t d, int32_t* assign, float* dist, void* stream);
int svcmi_ivf_blend_f32(const float* x, int32_t ldx, const int32_t* assign, const int32_t* list_off, const float* bank,
                        int32_t ldb, float* out, int32_t ldo, int32_t t, int32_t d, int32_t k, float ratio,
                        int32_t* out_idx, float* out_dist, void* stream);
int svcmi_segment_mean_f32(const float* x, int32_t ldx, const int32_t* order, const int32_t* seg_off, float* out, int32_t ldo,
                           int32_t segments, int32_t d, void* stream);

/* int16 side output, vits_decoder/generator.py:167-173: clamp(32768*x, -32768, 32767) truncated to short. */
int svcmi_source2wav_i16(const float* x, int16_t* y, int64_t n, void* stream);


/* =============================================================================================
 * Stage-level entry points: the composition of the kernels above into the
 * reference's forward passes, done by a C++ host INSIDE this library -- no Python in the loop, so a
 * non-Python caller can run a stage, and the Python facade (svcmi.SynthesizerInfer.inference,
 * svcmi.whisper AudioEncoder) is one ctypes call per stage.
 *
 * Models are plain structs of DEVICE pointers to operands already in the packed layouts the kernels
 * take (weight-norm folded, Conv1d [N,Cin,K] -> [N][K*Cin], ConvTranspose1d polyphase, q|k|v fused,
 * Flip folded into permutations: svcmi/weights.py does this once at load; `python -m svcmi.tools
 * pack` writes the same arena to a file a non-Python host can mmap / upload).  The structs are read
 * on the host during the call only.
 *
 * Memory: every intermediate lives in ONE caller-owned device workspace; svcmi_*_workspace_bytes
 * runs the same allocation plan without launching and returns the size needed for a shape.  Nothing
 * is allocated, nothing synchronises, every launch goes to `stream`: a call can be captured into a
 * hipGraph.  Return value as above (first failing launch aborts the stage).
 */
#define SVCMI_MAX_WHISPER_BLOCKS 32
#define SVCMI_MAX_ENC_LAYERS 8
#define SVCMI_MAX_FLOWS 8
#define SVCMI_MAX_WN_LAYERS 8
#define SVCMI_MAX_STAGES 6
#define SVCMI_MAX_AMP_BLOCKS 3
#define SVCMI_MAX_AMP_DILATIONS 3

/* one packed GEMM operand: w [n][ldw] fp32 as svcmi_conv_gemm_f32 takes it; bias [n] or NULL; w16 = the svcmi_pack_weights_lp
 * image for the model's `precision` (NULL = this layer always runs fp32), ldw16 its leading dimension */
typedef struct svcmi_weight {
    const float* w;
    const float* bias;
    const void* w16;
    const void* w16a;     /* the natural-k-order image of the SVCMI_PREC_*_A16 kernels (same ldw16), or NULL: 16-bit activations are not used here */
    int32_t n, ldw, ldw16;
    int32_t prec16;       /* enum svcmi_precision the w16 / w16a images were packed for (0 = unknown: trusted to match the model's mode); a launch
                           * whose mode differs runs on the fp32 operand instead of misreading the image */
} svcmi_weight;

/* ---- Whisper audio encoder, truncated as whisper/inference.py:11-29 does (n_layers = the kept blocks) */
typedef struct svcmi_whisper_block {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;     /* attn_ln, mlp_ln (whisper/model.py:113-116) */
    svcmi_weight qkv, o, m1, m2;                    /* q|k|v fused [3S][S] (key has no bias: zeros), out, mlp.0, mlp.2 */
} svcmi_whisper_block;

typedef struct svcmi_whisper_model {
    int32_t n_state, n_heads, n_layers, n_mels, n_ctx;
    int32_t precision;                              /* enum svcmi_precision of the linear layers (w16 images must match) */
    svcmi_weight conv1, conv2;                      /* whisper/model.py:144-145 */
    const float* pos;                               /* positional_embedding [n_ctx][n_state] */
    const float *lnp_g, *lnp_b;                     /* ln_post */
    svcmi_whisper_block blocks[SVCMI_MAX_WHISPER_BLOCKS];
    /* tuning (0 = the library's table for MI355X): K slices of the two N = n_state projections and tile overrides (>> 8) for
     * single windows (M <= small_m_rows rows); batched windows use the heuristic of svcmi_conv_gemm_f32 */
    int32_t split_o, split_mlp, tile_qkv, tile_o, tile_mlp1, tile_mlp2, small_m_rows, reserved;
    float lp_min_flops;                             /* launches below this stay fp32 in a 16-bit mode (0 = 1.5e9) */
    int32_t reserved2;
} svcmi_whisper_model;

int64_t svcmi_whisper_workspace_bytes(const svcmi_whisper_model* m, int32_t batch, int32_t n_frames);
/* AudioEncoder.forward, whisper/model.py:147-163, on mel [batch][n_mels][n_frames] (the reference's NCL layout) with the
 * `mel + noise_scale * noise` of whisper/inference.py:46,58 fused into the layout change (noise: same layout, or NULL):
 * conv1 + GELU, conv2 (stride 2) + GELU, + positional_embedding, n_layers x ResidualAttentionBlock (:118-129), ln_post.
 * out: [batch][tw][n_state] time-major, tw = (n_frames - 1) / 2 + 1 <= n_ctx (else SVCMI_EINVAL: "incorrect audio shape", :156). */
int svcmi_whisper_encoder_fwd(const svcmi_whisper_model* m, const float* mel, const float* noise, float noise_scale,
                              int32_t batch, int32_t n_frames, float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- SynthesizerInfer (vits/models.py:211-256) */
/* layer classes of the per-layer mixed-precision policy: prior encoder (enc_p: pre / hub / attention + FFN layers / proj, and its attention
 * kernel), flow (pre / in / res_skip / post of every coupling layer), the generator's trunk (conv_pre + every ups[i]), and the AMP-block
 * convolutions of generator stage i (SVCMI_CLASS_AMP0 + i; stages on the fused vector-ALU kernels compute in fp32 whatever this says) */
#define SVCMI_AMP_CLASSES 5                         /* amp0 .. amp4; a generator with a sixth stage (SVCMI_MAX_STAGES) runs it in amp4's mode */
enum svcmi_prec_class { SVCMI_CLASS_ENC = 0, SVCMI_CLASS_FLOW = 1, SVCMI_CLASS_UPS = 2, SVCMI_CLASS_AMP0 = 3 /* .. AMP0 + SVCMI_AMP_CLASSES - 1 */,
                        /* the prior encoder's ATTENTION kernel: SVCMI_PREC_BF16 / _F16 = both products on the 16-bit matrix cores
                         * (svcmi_attention16, from a 16-bit copy of the QKV projection's output), anything else = the fp32 kernel */
                        SVCMI_CLASS_ENC_ATTN = 8 };
#define SVCMI_PREC_CLASSES 12
#ifdef __cplusplus
static_assert(SVCMI_CLASS_AMP0 + SVCMI_AMP_CLASSES <= SVCMI_CLASS_ENC_ATTN, "AMP classes must not reach the encoder-attention class");
#endif
typedef struct svcmi_enc_layer {                    /* attentions.Encoder layer i, vits/attentions.py:36-72 */
    svcmi_weight qkv, o, f1, f2;                    /* conv_q|k|v fused, conv_o, FFN conv_1 / conv_2 */
    const float *rel_k, *rel_v;                     /* emb_rel_k / emb_rel_v [2*window+1][H/heads] */
    const float *g1, *b1, *g2, *b2;                 /* norm_layers_1 / _2 */
} svcmi_enc_layer;

typedef struct svcmi_wn_layer { svcmi_weight in, rs; } svcmi_wn_layer;   /* WN in_layers[l], res_skip_layers[l] (modules.py:148-176) */

typedef struct svcmi_flow_layer {                   /* one ResidualCouplingLayer in EXECUTION order of reverse=True, Flip folded in */
    int32_t x0_off, x1_off, n_wn, reserved;
    svcmi_weight pre, post, snac;                   /* pre writes (h | skip) rows of 2H (upper H output channels zero) */
    svcmi_wn_layer wn[SVCMI_MAX_WN_LAYERS];
} svcmi_flow_layer;

typedef struct svcmi_amp_block {                    /* AMPBlock, vits_decoder/bigv.py:22-58 */
    int32_t k, n_dil;
    int32_t dil[SVCMI_MAX_AMP_DILATIONS];
    int32_t reserved;
    svcmi_weight c1[SVCMI_MAX_AMP_DILATIONS], c2[SVCMI_MAX_AMP_DILATIONS];
    const float *a1_alpha[SVCMI_MAX_AMP_DILATIONS], *a1_beta[SVCMI_MAX_AMP_DILATIONS];   /* activations[2q]   (log-scale alpha / beta) */
    const float *a2_alpha[SVCMI_MAX_AMP_DILATIONS], *a2_beta[SVCMI_MAX_AMP_DILATIONS];   /* activations[2q+1] */
} svcmi_amp_block;

typedef struct svcmi_gen_stage {                    /* ups[i] + noise_convs[i] + resblocks[i*nk .. i*nk+nk) (generator.py:183-194) */
    int32_t u, c, cp, up_taps, up_pad, nz_k, nz_stride, nz_pad, n_blocks, reserved;
    svcmi_weight up, nz;                            /* polyphase ConvTranspose1d operand [u*cp][taps*cin_p]; noise conv [cp][>= nz_k] */
    svcmi_amp_block blocks[SVCMI_MAX_AMP_BLOCKS];
} svcmi_gen_stage;

typedef struct svcmi_synth_model {
    int32_t hidden, inter, n_heads, enc_window, enc_ffn_kernel, flow_kernel, n_enc, n_flow, n_stages;
    int32_t ppg_dim, vec_dim, spk_dim, upsample_input, hop;
    int32_t precision;                              /* enum svcmi_precision of prior encoder / flow / generator GEMMs, or SVCMI_PREC_MIXED */
    float lp_min_flops;                             /* 0 = 1.5e9 */
    float sampling_rate, merge_b;
    /* SVCMI_PREC_MIXED: the mode (SVCMI_PREC_F32 / _BF16X3 / _BF16 / _F16) of each layer class -- enum svcmi_prec_class; the w16 images of a
     * class's weights must be packed for its mode.  Ignored in every other `precision`. */
    int32_t class_prec[SVCMI_PREC_CLASSES];
    svcmi_weight pre, hub, proj;                    /* enc_p.pre / hub / proj (vits/models.py:26-37) */
    const float* pit_emb;                           /* enc_p.pit [256][hidden] */
    svcmi_enc_layer enc[SVCMI_MAX_ENC_LAYERS];
    svcmi_flow_layer flow[SVCMI_MAX_FLOWS];
    svcmi_weight adapter, conv_pre, post;           /* SpeakerAdapter W_scale|W_bias fused [2U][spk]; conv_pre; conv_post (1 row, no bias) */
    const float* merge_w;                           /* NSF merge weights [11] (nsf.py:378-381) */
    const float* filt;                              /* the 12 Kaiser-sinc taps */
    const float *post_alpha, *post_beta;            /* activation_post */
    svcmi_gen_stage stages[SVCMI_MAX_STAGES];
} svcmi_synth_model;

enum svcmi_synth_stop { SVCMI_STOP_NONE = 0, SVCMI_STOP_PRIOR = 1, SVCMI_STOP_FLOW = 2, SVCMI_STOP_GEN_PRE = 3, SVCMI_STOP_STAGE0 = 4 /* + i */ };

typedef struct svcmi_synth_io {
    /* inputs, all device pointers */
    const float* ppg;        /* [batch][t >> ppg_row_shift][ppg_dim] (+ ppg_bstride): shift 1 = the 50 fps Whisper output, x2 repeat fused */
    const float* vec;        /* [batch][t][vec_dim] */
    const float* pit;        /* [batch][t] Hz */
    const float* spk;        /* [batch][spk_dim] */
    const int32_t* lengths;  /* [batch] */
    const float* source;     /* [batch][t*hop] harmonic source (svcmi_pitch2source_fwd) */
    const float* noise;      /* [batch][inter][t]: the randn_like(m) of vits/models.py:51 in the reference's layout */
    int64_t ppg_bstride;     /* floats; 0 = dense */
    int32_t ppg_row_shift, batch, t;
    int32_t stream_frames;   /* 0 = the generator sees the whole chunk; N = time tiles of N frames + 32-frame halos (bit-identical for every N: */
                             /* the kernels' per-row arithmetic does not depend on the launch size, tile policies are pinned by stage width) */
    int32_t stop_after;      /* enum svcmi_synth_stop: timing aid (scripts/stage_times.py): the pipeline stops there, `wave` is not written */
    int32_t reserved;
    /* outputs */
    float* wave;             /* [batch][t*hop] */
    float* z_p;              /* optional [batch][t][inter] time-major copies of the prior sample / flow output (NULL = skip) */
    float* z;
} svcmi_synth_io;

int64_t svcmi_synth_workspace_bytes(const svcmi_synth_model* m, int32_t batch, int32_t t, int32_t stream_frames);
/* Generator.pitch2source, vits_decoder/generator.py:160-165: svcmi_pitch_prefix_f64 + svcmi_pitch_source_f32.  workspace >= batch*t*11 doubles. */
int svcmi_pitch2source_fwd(const svcmi_synth_model* m, const float* f0, const float* rand_ini, const float* noise, int32_t batch,
                           int32_t t, float* source, void* workspace, int64_t workspace_bytes, void* stream);
/* TextEncoder.forward (vits/models.py:39-52 + attentions.py:60-72): z_p [batch][t][inter] time-major */
int svcmi_text_encoder_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, float* z_p, void* workspace, int64_t workspace_bytes, void* stream);
/* ResidualCouplingBlock.forward(reverse=True) (vits/models.py:89-94): x [batch][t][inter] updated in place */
int svcmi_flow_reverse_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, float* x, void* workspace, int64_t workspace_bytes, void* stream);
/* Generator.inference (vits_decoder/generator.py:175-200) on z [batch][t][inter] time-major -> io->wave */
int svcmi_generator_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, const float* z, void* workspace, int64_t workspace_bytes, void* stream);
/* SynthesizerInfer.inference (vits/models.py:251-256): the three stages above on one workspace */
int svcmi_synth_infer_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, void* workspace, int64_t workspace_bytes, void* stream);

/* y[r][0:cols] = x[r][0:cols] for `rows` rows of leading dimensions ldy / ldx (floats): the strided device copies of the host
 * above (time tiles of the streaming decoder, optional z_p / z outputs) as a kernel, so that they are stream-ordered and graph-capturable. */
int svcmi_copy2d_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, void* stream);

/* Per-launch timing of the stage entry points (bench.py's roofline leg): between svcmi_trace_begin and svcmi_trace_end every
 * kernel launch the host makes on the calling thread is bracketed by HIP events on its stream.  svcmi_trace_end waits for the last one and
 * fills `out` (up to `cap` records, in launch order); returns the number of launches seen.  Not for production use. */
typedef struct svcmi_trace_record { int32_t op; float ms; double flops, bytes; } svcmi_trace_record;
/* (op: bits 0-7 = the entry point, svcmi_trace_op_name(op); bits 8+ = the enum svcmi_precision code a reduced-precision GEMM launch ran
 * with, so a test can tell an _A16 launch from one that rounds in registers) */
int svcmi_trace_begin(int32_t max_records);
int svcmi_trace_end(svcmi_trace_record* out, int32_t cap);
const char* svcmi_trace_op_name(int32_t op);

/* Packed-model files: a whole model for a host without Python.  `python -m svcmi.tools pack --config cfg.yaml --model sovits5.0.pth
 * --out synth.svcmi` (or --whisper large-v2.pt) writes header + relocation table + the model struct with every pointer stored as
 * (byte offset into the arena + 1; 0 = NULL) + the flat fp32 weight arena in the packed layouts.  The host reads / mmaps the file,
 * asks svcmi_packed_model_info where the arena lies (kind: 1 = svcmi_synth_model, 2 = svcmi_whisper_model), uploads those
 * arena_bytes to device memory (256-byte aligned) and calls svcmi_packed_model_bind, which fills `model_out` (model_bytes = sizeof of the
 * struct of that kind) with device pointers.  examples/stage_host.cpp does exactly this and runs svcmi_synth_infer_fwd. */
int svcmi_packed_model_info(const void* file, int64_t file_bytes, int32_t* kind, int64_t* arena_offset, int64_t* arena_bytes);
int svcmi_packed_model_bind(const void* file, int64_t file_bytes, const void* device_arena, void* model_out, int64_t model_bytes);

/* Layout check for FFI bindings: out[i] = sizeof of svcmi_weight, svcmi_whisper_model, svcmi_synth_model, svcmi_synth_io,
 * svcmi_trace_record, svcmi_conv_desc, svcmi_snake_conv_desc (in this order, up to `cap`); returns the count written.  A binding
 * compares them with its own struct sizes before it trusts a filled struct (svcmi/_lib.py does at load). */
int svcmi_struct_sizes(int64_t* out, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SVCMI_H */
