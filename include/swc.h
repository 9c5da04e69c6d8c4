/*
 * swc.h — C-ABI of libswc_hip.so: the MI355X (gfx950) kernels behind the
 * SimWhisper-Codec encode -> quantize -> vocode hot path.
 *
 * The reference has no native layer: every entry point below replaces a run of
 * stock ATen calls inside one reference function (cited per entry, paths
 * relative to the reference repo).  The Python host (simwhisper_codec_amd/codec.py)
 * binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *    unless a parameter is documented as host.
 *  - activations are "frame-major": a (B, C, T) reference tensor is stored as
 *    [B][T][C] (channels contiguous), so a row is one frame/token.
 *  - every function only ENQUEUES work on `stream` and returns: 0 on success,
 *    a negative SWC_E_* code otherwise (swc_last_error() gives the text).
 *    Nothing allocates, synchronises or throws; the caller owns all memory.
 *  - `stream` is a hipStream_t passed as void*.
 *  - dtype codes: SWC_F32 = 0, SWC_BF16 = 1, SWC_F16S = 2 (split f16), SWC_FP8 = 3 (OCP e4m3fn); defined below.
 */
#ifndef SWC_H_
#define SWC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWC_OK 0
#define SWC_E_ARG (-1)     /* bad shape / alignment / null pointer */
#define SWC_E_LAUNCH (-2)  /* hipLaunch error */
#define SWC_E_NODEV (-3)   /* no HIP device */

#define SWC_F32 0
#define SWC_BF16 1
/*
 * SWC_F16S — "split f16": an f32 matrix X[rows][K] (K % 32 == 0) stored as fp16 [rows][2K]; every block of 32
 * logical elements k = 32q .. 32q+31 is 32 hi halves followed by 32 lo halves, hi = f16(x * 2^s),
 * lo = f16(x * 2^s - hi): 22 significand bits per element at 4 bytes, i.e. the byte layout and strides of the
 * f32 matrix.  swc_gemm contracts two such operands with three f16 MFMAs per k-step (hi*hi + hi*lo + lo*hi,
 * f32 accumulate): f32-class accuracy (the dropped lo*lo term and the 2^-22 representation error are below the
 * f32 accumulation rounding of a K >= 64 dot product) at 3/16 of the exact-f32 MFMA cycles.  `lda/ldw/ldc`
 * stay in LOGICAL elements.  The power-of-two scales are the caller's: alpha un-does them.
 */
#define SWC_F16S 2
/* scale 2^s of every split-f16 ACTIVATION written by swc_layernorm / swc_snake_aa / swc_attention_ex:
 * |x| up to 1023 without saturation, lo halves normal down to |x| ~ 2e-3 */
#define SWC_F16S_ACT_SCALE 64.0f
/*
 * SWC_FP8 — OCP e4m3fn bytes (gfx950's native fp8; max 448, 3 mantissa bits), one byte per element, for the
 * "fp8 Whisper-encoder GEMMs" preset (BASELINE.json configs[4]): swc_gemm contracts two fp8 operands with the
 * block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, K = 128 per instruction, f32 accumulate: twice the bf16
 * rate per clock; the non-scaled v_mfma_f32_16x16x32_fp8_fp8 runs at the bf16 rate and is not used).  Values are stored
 * pre-multiplied by a power of two chosen by
 * the caller (activations: SWC_FP8_ACT_SCALE; weights: per tensor) and un-done by `alpha`; conversions saturate
 * at +-448.  This preset trades the bit-exact indices for speed: its tolerance is stated in DESIGN.md section 4.
 */
#define SWC_FP8 3
#define SWC_FP8_ACT_SCALE 16.0f
/* SWC_F16 — plain IEEE half precision (11 significand bits, |x| <= 65504), one 16-bit element per value.  Exists as the
 * operand type INSIDE swc_convnext_block (operand_dtype): no tensor of the path is stored in it. */
#define SWC_F16 4

#define SWC_ACT_NONE 0
#define SWC_ACT_GELU 1 /* exact erf GELU == nn.GELU() / ACT2FN["gelu"] */

int swc_version(void);
const char* swc_last_error(void);
/* number of visible HIP devices, or SWC_E_NODEV */
int swc_device_count(void);

/* A one-wave kernel that occupies `stream` for `us` microseconds (<= 100 ms) and exits: phase shift between two chains
 * of launches on two streams (no reference counterpart; the reference runs one stream). */
int swc_delay_us(int32_t us, void* stream);

/*
 * Range guard of the reduced-range operand formats.  The reference computes in fp32 and has no such limit
 * (modules.py:214-232 only clamps fp16 / bf16 infinities); here split-f16 activations clip at |x| = 65504 / 64 = 1023
 * and fp8 activations at |x| = 448 / 16 = 28.  `counters` is a caller-owned DEVICE array of two uint32:
 * [0] += 1 for every thread of a producer kernel (swc_layernorm, swc_snake_aa, swc_gemm epilogue, swc_cast_*) that
 * clipped at least one split-f16 element, [1] likewise for fp8.  The pointer is kept PER CALLING THREAD (the only
 * state the library holds besides the error text) and applies to every later call of that thread until replaced;
 * NULL (the default) turns the accounting off.  The caller zeroes and reads the array; a non-zero count after
 * an encode means the codes may differ from the reference's and the stage must be re-run on exact-f32 operands.
 */
int swc_set_saturation_counter(uint32_t* counters);

/*
 * GEMM / implicit-GEMM Conv1d on MFMA:  C = epi(A (*) W^T)
 *
 * Replaces nn.Linear / nn.Conv1d calls: audiocodec/nn/modules.py:159-161,185
 * (q/k/v/out proj), :225-226 (fc1+GELU, fc2), :316-319 (conv1, conv2),
 * :544,548 (in_proj,to_latent), :43,45 (ResidualUnit convs), :615,621,
 * :464-465 (deconv as GEMM, see swc_deconv_col2im), :1494 (embed),
 * :1240-1244 (pwconv1+GELU, pwconv2*gamma+res), :1064 (head.out), and the
 * DFT / mel / inverse-DFT contractions of feature_extractor.py:99-103 and
 * modules.py:861.
 *
 * A: [rows_in][lda]  element type a_dtype.  W: [N][ldw] (taps*K columns, tap-major),
 * same element type as A.  C: [M][ldc] of c_dtype.
 * Output row r = b*t_out + t reads, for tap j, input row
 *   b*t_in + t*stride + j*dil - pad   (zero if outside [0, t_in)).
 * A plain GEMM is taps=1, stride=1, dil=1, pad=0, t_in=t_out=M.
 * epilogue: v = alpha * acc + bias[n]; v = act(v); v *= gamma[n]; v += residual[r][n];
 *           C = (c_dtype == F16S) ? split(v * out_scale) : (c_dtype == FP8 ? e4m3(v * out_scale) : v).
 *           (alpha == 0 is read as 1)
 * K must be a multiple of 4 (f32) / 8 (bf16) / 32 (f16s) / 16 (fp8); lda, ldw keep rows 16-byte aligned.
 * a_dtype F16S: A and W are both split-f16; c_dtype F16S needs N % 32 == 0.
 * a_dtype FP8: A and W are both e4m3 bytes; c_dtype FP8 is available for a_dtype FP8 only.
 */
typedef struct swc_gemm_args {
    const void* A;
    const void* W;
    void* C;
    const float* bias;     /* [N] or NULL */
    const float* gamma;    /* [N] or NULL; f32 outputs only (c_dtype == SWC_F32) */
    const float* residual; /* [M][ldr] f32 or NULL (may alias C when c_dtype == F32) */
    int64_t lda, ldw, ldc, ldr;
    int32_t M, N, K;
    int32_t taps, dil, stride, pad, t_in, t_out;
    int32_t a_dtype, c_dtype, act;
    float alpha;     /* multiplies the accumulator (undoes operand scales); 0 means 1 */
    float out_scale; /* F16S / FP8 outputs only: 2^s applied before the split / conversion; 0 means 1 */
} swc_gemm_args;
int swc_gemm(const swc_gemm_args* args, void* stream);

/*
 * Varlen multi-head self-attention, head_dim 64, non-causal, keys >= len masked.
 * Replaces VarLenAttention score/softmax/PV (modules.py:164-182) incl. the
 * additive mask of :111-143.  qkv: [B][T][3*H*64] (q | k | v, q pre-scaled by
 * 64^-0.5 through the packed weights), out: [B][T][H*64].  Rows t >= lens[b]
 * produce finite don't-care values (the caller masks them, modules.py:358,460).
 * dtype is the element type of qkv and out.
 */
int swc_attention(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T,
                  int32_t H, int32_t dtype, void* stream);
/*
 * 16-bit operand attention (same semantics): dtype BF16 = bf16 q/k/v in, bf16 out, on the bf16 MFMA;
 * dtype F16S = split-f16 q/k/v in (SWC_F16S_ACT_SCALE), split-f16 out, three f16 MFMAs per product
 * (f32-class scores and outputs).  128 queries per workgroup, K/V tiles double-buffered by LDS-DMA,
 * V^T operand through ds_read_b64_tr_b16.
 * row_start (NULL = padded layout, utterance b at rows b*T ..): VALID-TOKEN PACKING — utterance b's lens[b] rows of qkv and
 * out start at row row_start[b] and nothing follows them but the next utterance: ragged batches then cost their valid
 * tokens, not B x the longest row (the reference pads every row, modules.py:111-143 masks afterwards).
 */
int swc_attention16(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T, int32_t H,
                    int32_t dtype, const int32_t* row_start, void* stream);
/* f32 qkv in and the output written as out_dtype (F32 | F16S at SWC_F16S_ACT_SCALE), exact-f32 MFMA */
int swc_attention_ex(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T, int32_t H,
                     int32_t out_dtype, void* stream);

/*
 * LayerNorm over the last dim.  Replaces nn.LayerNorm calls modules.py:216,224,
 * 353,457 (eps 1e-5) and :1239,1499,1503 (eps 1e-6).  x: [B][t_in][C] f32;
 * y: [B][t_out][C] (y_dtype: F32 | BF16 | F16S at SWC_F16S_ACT_SCALE | FP8 at SWC_FP8_ACT_SCALE).  Every output row is written:
 * rows t >= t_in (t_out > t_in: the zero extension the down-sampler reads) are zeros, and when lens != NULL rows
 * t >= lens[b] are zeros as well (torch.where(mask, h, 0), modules.py:358,460).
 */
int swc_layernorm(const float* x, void* y, const float* w, const float* b, const int32_t* lens,
                  int32_t B, int32_t t_in, int32_t t_out, int32_t C, float eps, int32_t y_dtype,
                  const int32_t* row_start, void* stream);
/* row_start != NULL (needs lens): x is PACKED — utterance b's lens[b] rows start at row row_start[b]; y stays padded
 * [B][t_out][C] (this is how the packed token stream of a transformer returns to the padded layout of the convolutions). */

/* padded [B][T][row_bytes] -> packed: the first lens[b] rows of utterance b are copied to row row_start[b] of dst. */
int swc_pack_rows(const void* src, void* dst, const int32_t* row_start, const int32_t* lens, int32_t B, int32_t T,
                  int64_t row_bytes, void* stream);

/*
 * ConvNeXt back half in one kernel (modules.py:1241-1247, 24x per Vocos call = half of all FLOPs of the path):
 *     x[M][C] += gamma * ( GELU( y W1^T + b1 ) W2^T + b2 )          pwconv1 -> nn.GELU -> pwconv2 -> gamma -> residual
 * y: [M][C] bf16 (swc_dwconv7_ln output), x: [M][C] f32 residual stream, updated in place; b1 [I], b2 [C], gamma [C] f32.
 * The [M][I] intermediate stays on chip (registers -> LDS -> MFMA operand).  Built for C = 512, I % 128 == 0 (the
 * shipped Vocos: 512 / 4096); other geometries return SWC_E_ARG and the caller runs two swc_gemm calls instead.
 * `w_stream` is the pair (pwconv1.weight [I][C], pwconv2.weight [C][I]) in bf16, re-ordered ONCE at load by
 * swc_convnext_pack into the order in which each wave consumes 1 KiB MFMA operand fragments
 * (swc_convnext_stream_bytes(C, I) bytes, 0 for an unsupported geometry).  bf16 operands, f32 accumulation; GELU is the
 * refit sigmoid form of the bf16 swc_gemm epilogue (|error| <= 2.7e-4).  The pack step takes the block's gamma: builds with
 * -DCX_RES_ACC=1 (a measured build option, not the default) fold it into pwconv2's rows and start pwconv2's accumulators at
 * x + gamma * b2; the gamma / b2 arguments of swc_convnext_mlp / swc_convnext_block must be the ones the stream was packed with.
 */
int64_t swc_convnext_stream_bytes(int32_t C, int32_t I);
int swc_convnext_pack(const void* w1_16bit, const void* w2_16bit, const float* gamma, void* w_stream, int32_t C, int32_t I,
                      void* stream);
int swc_convnext_mlp(const void* y, const void* w_stream, const float* b1, const float* b2, const float* gamma,
                     float* x, int32_t M, int32_t C, int32_t I, void* stream);
/*
 * The WHOLE ConvNeXtBlock.forward (modules.py:1229-1248) in one kernel: the workgroup also computes the front half
 * (depthwise Conv1d k7 pad 3 + LayerNorm(eps), the arithmetic of swc_dwconv7_ln) of its 128 frames, so the bf16 copy of the
 * normalised activations never exists in memory either: per block the residual stream x [B][T][C] f32 is read (+6 halo rows
 * per 128) and x_out written once.  NOT in place (x_out != x: tiles read halo rows of their neighbours, which a finished
 * neighbour would already have updated); the caller ping-pongs two buffers over the 24 blocks.  dw_w7: [7][C], taps do
 * not cross utterances (rows b * T + t).  Same geometry limits.
 * t_limit (optional, device int32 [B]; ragged batches): frames t >= t_limit[b] of utterance b need not be computed — a
 * 128-frame tile that lies wholly at or beyond the limits of the utterances it touches returns at once and leaves its rows
 * of x_out undefined.  The caller's limit must cover what it keeps plus the receptive field of the remaining blocks
 * (3 frames per block); no reference counterpart (the reference computes every padded frame, model.py:327-333).
 * operand_dtype = SWC_BF16 | SWC_F16: the type of the block's INTERNAL MFMA operands — the LayerNorm output, the GELU output and
 * the two weight matrices of `w_stream` (swc_convnext_pack re-orders 16-bit elements whatever their format: pass it
 * pwconv1 / pwconv2 rounded to the same type).  SWC_F16 keeps 11 significand bits instead of bf16's 8 at the same MFMA rate;
 * the values it holds are bounded by the block itself (|LayerNorm output| <= sqrt(C) |ln_w| + |ln_b|, hidden activations, weights)
 * far inside +-65504.  x / x_out stay f32 either way.
 */
int swc_convnext_block(const float* x, float* x_out, const float* dw_w7, const float* dw_bias, const float* ln_w,
                       const float* ln_b, float eps, const void* w_stream, const float* b1, const float* b2,
                       const float* gamma, int32_t B, int32_t T, int32_t C, int32_t I, const int32_t* t_limit,
                       int32_t operand_dtype, void* stream);

/*
 * EXPERIMENT (round 4, not on the product path): swc_convnext_mlp on 64-frame tiles with two workgroups
 * per CU (2 waves per SIMD, 80 KiB of LDS each) and an optional start stagger of the second workgroup of every CU
 * (`stagger_cycles` shader cycles, 0 = none), so that one workgroup's HBM phases fall into the other's slice loop.  Same
 * arithmetic, operand types and arguments as swc_convnext_mlp; its own packed stream (swc_convnext64_pack).
 */
int64_t swc_convnext64_stream_bytes(int32_t C, int32_t I);
int swc_convnext64_pack(const void* w1_bf16, const void* w2_bf16, void* w_stream, int32_t C, int32_t I, void* stream);
int swc_convnext64_mlp(const void* y, const void* w_stream, const float* b1, const float* b2, const float* gamma, float* x,
                       int32_t M, int32_t C, int32_t I, int32_t stagger_cycles, void* stream);

/*
 * The MLP sub-block of one OmniWhisperTransformerLayer in one kernel (modules.py:224-232: final_layer_norm -> fc1 ->
 * exact GELU -> fc2 -> + residual), 12x per encoder / decoder call:
 *     x_out[M][D] = x + ( GELU( LayerNorm(x; ln_w, ln_b, eps) W1^T + b1 ) W2^T + b2 )
 *     y_next[M][D] = LayerNorm(x_out; next_ln_w, next_ln_b, eps) in bf16      (optional: y_next == NULL skips it)
 * x / x_out: f32 residual stream (x_out may be x: rows are independent); y_next is the operand of the NEXT layer's q/k/v
 * projection (self_attn_layer_norm, modules.py:216), so neither LayerNorm of the layer is a launch of its own and the
 * [M][F] hidden activations never exist in memory (two swc_gemm calls write and re-read them: 98 MB per layer at
 * 32 x 10 s).  bf16 operands, f32 accumulation; GELU is the refit sigmoid form of the bf16 swc_gemm epilogue
 * (|error| <= 2.7e-4).  Built for D = 768, F % 256 == 0 (the shipped 768 / 3072); other geometries return SWC_E_ARG and the
 * caller runs swc_layernorm + two swc_gemm calls instead.  `w_stream` is the pair (fc1.weight [F][D], fc2.weight [D][F]) in
 * bf16, re-ordered ONCE at load by swc_mlp_pack into the order in which each wave consumes 1 KiB MFMA operand fragments
 * (swc_mlp_stream_bytes(D, F) bytes, 0 for an unsupported geometry).  Padded rows are computed like any other row, as in
 * the reference (the layer masks keys, not rows: modules.py:169-177).
 */
int64_t swc_mlp_stream_bytes(int32_t D, int32_t F);
int swc_mlp_pack(const void* w1_bf16, const void* w2_bf16, void* w_stream, int32_t D, int32_t F, void* stream);
int swc_mlp_block(const float* x, float* x_out, const float* ln_w, const float* ln_b, float eps, const void* w_stream,
                  const float* b1, const float* b2, const float* next_ln_w, const float* next_ln_b, void* y_next,
                  int32_t M, int32_t D, int32_t F, void* stream);

/*
 * Everything of one OmniWhisperTransformerLayer behind the attention (modules.py:219-232: out_proj -> + residual ->
 * final_layer_norm -> fc1 -> GELU -> fc2 -> + residual) in one kernel, plus the NEXT layer's self_attn_layer_norm:
 *     x'           = x + attn Wo^T + bo                      attn: [M][D] bf16, the attention output (swc_attention16)
 *     x_out[M][D]  = x' + ( GELU( LayerNorm(x'; ln_w, ln_b, eps) W1^T + b1 ) W2^T + b2 )
 *     y_next[M][D] = LayerNorm(x_out; next_ln_w, next_ln_b, eps) in bf16                       (optional, as swc_mlp_block)
 * swc_mlp_block with the out-projection in front of it: the workgroup's 64-token tile of attn goes to LDS by DMA, x' is
 * accumulated in the registers that afterwards accumulate fc2 — it never exists in memory, its LayerNorm is taken in the
 * accumulator layout (row statistics exchanged between the 4 waves through LDS), and the MLP's residual add costs nothing.
 * A decoder layer is then three launches: q/k/v GEMM, attention, this.  Same geometry limits, operand types and GELU as
 * swc_mlp_block; `w_stream` comes from swc_layer_tail_pack(out_proj.weight [D][D], fc1.weight [F][D], fc2.weight [D][F])
 * (swc_layer_tail_stream_bytes(D, F, fc1_dtype) bytes).  x_out may be x.
 * fc1_dtype = SWC_FP8 (preset fp8_fc1): fc1 alone runs on the block-scaled fp8 MFMA — fc1.weight is then e4m3 bytes [F][D] at a
 * per-tensor power-of-two scale sw, the kernel writes LayerNorm(x') to LDS as e4m3 at SWC_FP8_ACT_SCALE (clipping is counted:
 * swc_set_saturation_counter) and multiplies fc1's accumulators by fc1_alpha = 1 / (SWC_FP8_ACT_SCALE * sw); everything else stays
 * bf16.  fc1_dtype = SWC_BF16: fc1_alpha is ignored.  The stream must have been packed for the same fc1_dtype.
 * operand_dtype = SWC_BF16 | SWC_F16 (with fc1_dtype SWC_BF16 = "16-bit"): the type of the kernel's INTERNAL MFMA operands.  SWC_F16:
 * the attention tile is converted bf16 -> f16 in LDS (exact), LayerNorm(x') and GELU(h) are written as f16 and the stream holds the
 * three weight matrices rounded to f16 (swc_layer_tail_pack re-orders 16-bit elements whatever their format): 11 significand bits
 * instead of 8 at the same MFMA rate; conversions saturate at +-65504.  attn in and y_next out stay bf16, x / x_out f32.
 */
int64_t swc_layer_tail_stream_bytes(int32_t D, int32_t F, int32_t fc1_dtype);
int swc_layer_tail_pack(const void* wo_bf16, const void* w1, const void* w2_bf16, void* w_stream, int32_t D, int32_t F,
                        int32_t fc1_dtype, void* stream);
int swc_layer_tail(const void* attn, const float* x, float* x_out, const void* w_stream, const float* bo, const float* ln_w,
                   const float* ln_b, float eps, const float* b1, const float* b2, const float* next_ln_w,
                   const float* next_ln_b, void* y_next, int32_t M, int32_t D, int32_t F, int32_t fc1_dtype, float fc1_alpha,
                   int32_t operand_dtype, void* stream);

/*
 * A split-f16 projection onto the residual stream with the LayerNorm behind it in one kernel — the `mixed` encoder's
 * `x = x + out_proj(attn)` followed by final_layer_norm, and `x = x + fc2(h)` followed by the NEXT layer's
 * self_attn_layer_norm (modules.py:214-232), where the two-launch form is swc_gemm (f32 output, residual) + swc_layernorm:
 *     x_out[M][N]  = x + alpha * (A W^T) + bias          A: [M][lda >= K] split-f16 at SWC_F16S_ACT_SCALE, W: [N][K] split-f16
 *     y_next[M][N] = LayerNorm(x_out; ln_w, ln_b, eps) as split-f16 at SWC_F16S_ACT_SCALE   (optional: y_next == NULL skips it)
 * Three f16 MFMAs per product as in swc_gemm (hi*hi + hi*lo + lo*hi, f32 accumulation); the epilogue's element arithmetic is
 * swc_gemm's and the LayerNorm's is swc_layernorm's, so y_next is bit-identical to swc_layernorm of the x_out this call stored
 * (x_out itself differs from swc_gemm's in the last bits: another MFMA shape, another order of the k sum).  Clipping of y_next is
 * counted (swc_set_saturation_counter).  Built for N = 768 and K % 64 == 0 (the shipped 768 / 3072); other geometries return
 * SWC_E_ARG and the caller runs the two launches.  `w_stream`: W re-ordered ONCE at load by swc_proj_ln_pack into the order in
 * which each wave consumes 1 KiB MFMA operand fragments (swc_proj_ln_stream_bytes(N, K) bytes, 0 for an unsupported geometry).
 * x_out may be x.  lda in logical columns.
 * Measured slower than the two launches at the path's shapes (profiles/r04_proj_ln_split_f16.txt: 85 / 222 us against 78 / 194 us at
 * 16 000 tokens, K = 768 / 3072): AudioCodec does not call it; a caller with other shapes may.
 */
int64_t swc_proj_ln_stream_bytes(int32_t N, int32_t K);
int swc_proj_ln_pack(const void* w_f16s, void* w_stream, int32_t N, int32_t K, void* stream);
int swc_proj_ln(const void* a_f16s, int64_t lda, const void* w_stream, const float* bias, float alpha, const float* x,
                float* x_out, const float* ln_w, const float* ln_b, float eps, void* y_next, int32_t M, int32_t N, int32_t K,
                void* stream);

/*
 * ConvNeXt front half: depthwise Conv1d(k=7, pad=3, groups=C) + LayerNorm(eps)
 * (modules.py:1233-1239).  x: [B][T][C] f32, w: [7][C], y: [B][T][C] (y_dtype).
 */
int swc_dwconv7_ln(const float* x, void* y, const float* w, const float* bias, const float* ln_w,
                   const float* ln_b, int32_t B, int32_t T, int32_t C, float eps, int32_t y_dtype,
                   void* stream);

/*
 * Anti-aliased SnakeBeta (Activation1d): replicate-pad, 2x kaiser-sinc upsample,
 * x + sin^2(x*a)/(b+1e-9), 2x low-pass downsample.  Replaces
 * alias_free_torch/act.py:23-28 + resample.py:25-33,46-49 + filter.py:83-92 +
 * activations.py:107-120.  x: [B][T][C] f32; alpha/beta: [C] already exp()'d
 * (alpha_logscale); filt: host pointer to the 12 taps; y: [B][T][C] (y_dtype).
 */
int swc_snake_aa(const float* x, void* y, const float* alpha, const float* beta,
                 const float* filt_host12, int32_t B, int32_t T, int32_t C, int32_t y_dtype,
                 void* stream);

/*
 * Grouped finite scalar quantiser, levels [8,7,6,6] per group of 4 channels.
 * Replaces GroupFiniteScalarQuantizer.forward / .decode (quantizer.py:273-318,
 * 129-157,169-200,207-224).
 * encode: z [B][T][ldz>=4G] f32 -> zq [B][t_pad][4G] f32 (zeros for t >= lens[b] and t >= T),
 *         codes [G][B][t_pad] int32 (same masking).
 * decode: codes [G][B][T] int64 -> zq [B][T][ldq] f32, cols >= 4G zero, masked by lens.
 * consts_host12 (HOST pointer): scale[4] | offset[4] | shift[4] as f32, computed by the
 * caller with the reference's own formula (quantizer.py:131-137) so that the constants
 * are bit-identical to the ones the reference derives at run time.
 */
int swc_fsq_encode(const float* z, int64_t ldz, float* zq, int32_t* codes, const int32_t* lens,
                   const float* consts_host12, int32_t B, int32_t T, int32_t t_pad, int32_t G,
                   void* stream);
int swc_fsq_decode(const int64_t* codes, float* zq, int64_t ldq, const int32_t* lens, int32_t B,
                   int32_t T, int32_t G, void* stream);
/* the same with the levels of the four channels of a group given by the caller (HOST pointer, 4 values in 2..1024 whose product
 * fits an int32 code: quantizer.py:47-120 is config-driven); swc_fsq_encode / swc_fsq_decode are these with {8, 7, 6, 6} */
int swc_fsq_encode_levels(const float* z, int64_t ldz, float* zq, int32_t* codes, const int32_t* lens,
                          const float* consts_host12, const int32_t* levels_host4, int32_t B, int32_t T,
                          int32_t t_pad, int32_t G, void* stream);
int swc_fsq_decode_levels(const int64_t* codes, float* zq, int64_t ldq, const int32_t* lens,
                          const int32_t* levels_host4, int32_t B, int32_t T, int32_t G, void* stream);

/*
 * Whisper log-mel front end (feature_extractor.py:86-112,207-214), three steps
 * around two swc_gemm calls (windowed DFT, mel filter bank):
 *  frames: wav [B][ld_wav] f32 with n[b] valid samples, virtually zero-padded to
 *          n_pad (480000) then reflect-padded by 200 -> frames [B][T][400]
 *          (frame t = padded samples t*160-200 .. +399).
 *  power : dft [rows][ld] (re 0..200 | im 201..401) -> pw [rows][ldp] (|X|^2, cols >= 201 zero)
 *  logmax: mel [B][T][ld] (first n_mel cols valid) -> in-place log10(max(x,1e-10)),
 *          per-utterance max into umax[b] (caller presets umax to -10 when padding
 *          frames beyond T exist, else -inf)
 *  final : y = (max(x, umax[b]-8)+4)/4 written to out [B][T][ldo] (out dtype),
 *          cols n_mel..ldo zeroed.
 */
int swc_mel_frames(const float* wav, int64_t ld_wav, const int32_t* n, int32_t n_pad,
                   float* frames, int32_t B, int32_t T, void* stream);
int swc_mel_power(const float* dft, int64_t ld, float* pw, int64_t ldp, int64_t rows,
                  void* stream);
int swc_mel_logmax(float* mel, int64_t ld, float* umax, int32_t B, int32_t T, int32_t n_mel,
                   void* stream);
int swc_mel_final(const float* mel, int64_t ld, const float* umax, void* out, int64_t ldo,
                  int32_t B, int32_t T, int32_t n_mel, int32_t out_dtype, void* stream);

/*
 * ConvTranspose1d(k=3, stride s, pad 0) tail: y3 [B][T][3][C] (the GEMM of the
 * input against the three taps) -> out [B][t_out][ldo] = bias + sum_j y3[t][j] at
 * position s*t + j (modules.py:464-465).  t_out <= (T-1)*s + 3; cols >= C zeroed.
 */
int swc_deconv_col2im(const float* y3, const float* bias, void* out, int64_t ldo, int32_t B,
                      int32_t T, int32_t C, int32_t s, int32_t t_out, int32_t out_dtype,
                      void* stream);

/*
 * ISTFT head (modules.py:1065-1081, 861-884):
 *  spec: h [rows][ldh] (mag 0..320 | phase 321..641) -> s [rows][lds]:
 *        re = min(exp(mag),100)*cos(p) at col k, im at col 321+k, pad zero.  s_dtype F32 | BF16 | F16S (split-f16 at
 *        SWC_F16S_ACT_SCALE, lds % 32 == 0: what swc_cast_f32_f16s would write for the same values, |S| <= 100).
 *  ola : frames [B][T][640] (windowed inverse DFT, a swc_gemm) -> wav [B][T*160]:
 *        overlap-add hop 160, crop 240, divide by the hann^2 envelope.
 */
int swc_istft_spec(const float* h, int64_t ldh, void* s, int64_t lds, int64_t rows,
                   int32_t s_dtype, void* stream);
int swc_istft_ola(const float* frames, const float* window_sq, float* wav, int32_t B, int32_t T,
                  void* stream);

/*
 * Code bitstream (the reference keeps codes in memory only, model.py:302): 8 groups x 11 bits
 * = 11 bytes per 12.5 Hz frame (1100 bit/s).  codes [8][ldg] int32 (row g = group g, values < 2048) <-> bytes [11*T];
 * frame t = bytes 11t..11t+10, group g = bits 11g..11g+10 of the frame, LSB first.
 */
int swc_codes_pack(const int32_t* codes, int64_t ldg, void* bytes, int32_t T, void* stream);
int swc_codes_unpack(const void* bytes, int32_t* codes, int64_t ldg, int32_t T, void* stream);

/* f32 -> bf16 cast (weight packing / mode switches) */
int swc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
/* f32 [rows][ldx] (first K columns) -> split-f16 [rows][K] logical (K % 32 == 0), x scaled by `scale` */
int swc_cast_f32_f16s(const float* x, int64_t ldx, void* y, int64_t rows, int32_t K, float scale,
                      void* stream);
/* x (x_dtype F32 | BF16, n elements, n % 4 == 0) * scale -> e4m3 bytes, saturating (fp8 preset: weights at load,
 * attention outputs) */
int swc_cast_fp8(const void* x, int32_t x_dtype, void* y, int64_t n, float scale, void* stream);

/*
 * Batch assembly: n_rows separate device buffers (row i: nbytes[i] bytes at src[i], 4-byte aligned, nbytes % 4 == 0)
 * -> one [n_rows][ld_bytes] buffer, zero-filled beyond each row's length.  Replaces the per-utterance copy loops of
 * model.py:258-262,322-326 (the reference pads on the host); `src` and `nbytes` are DEVICE arrays.
 */
int swc_gather_rows(const void* const* src, const int64_t* nbytes, void* out, int64_t ld_bytes, int32_t n_rows,
                    void* stream);

/*
 * PCM16 <-> f32 on the device (the file loop around the path; utils/helpers.py:77-104 leaves both to
 * torchaudio.load / torchaudio.save on the host).  Files cross PCIe as 16-bit samples, half the bytes of f32, and the host
 * threads of inference.py only read and write bytes.
 *   pcm16_to_f32: out[i] = pcm[i] * 2^-15                          (torchaudio.load's normalisation; exact)
 *   f32_to_pcm16: pcm[i] = round_half_even(clip(x[i], -1, 1) * 32767), every step in f32 — the values
 *                 simwhisper_codec_amd.wavio.save_audio computes on the host (bit-exact; tests/test_kernels_gpu.py)
 * n samples, any alignment (16-byte aligned spans take the vector path).
 */
int swc_pcm16_to_f32(const int16_t* pcm, float* out, int64_t n, void* stream);
int swc_f32_to_pcm16(const float* x, int16_t* pcm, int64_t n, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* SWC_H_ */
