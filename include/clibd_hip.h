/*
 * clibd_hip.h — C ABI of libclibd_hip.so: the MI355X (gfx950) kernels behind the CLIBD contrastive
 * training step (scripts/train_cl.py -> bioscanclip/epoch/train_epoch.py:21-63 in the reference).
 *
 * The reference has NO native/FFI boundary of its own (it is pure Python on PyTorch), so
 * every entry point below cites the reference *Python* symbol whose arithmetic it replaces.  The Python
 * mirror of `bioscanclip.model` (package `clibd_amd.model`) binds these through ctypes.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers owned by the caller; kernels never allocate, free or synchronise;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); NULL = the null stream;
 *   - return 0 on success, a negative CLIBD_E* code otherwise; clibd_last_error() gives a thread-local
 *     message; shapes are validated on the host BEFORE any launch (a bad shape never reaches the GPU);
 *   - "bf16" buffers are uint16 bit patterns; activations row-major, leading dimension in ELEMENTS;
 *   - re-entrant: no mutable global state, safe on different streams / devices concurrently.
 */
#ifndef CLIBD_HIP_H
#define CLIBD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIBD_OK 0
#define CLIBD_EINVAL (-1)  /* bad shape / null pointer / unsupported size */
#define CLIBD_ELAUNCH (-2) /* hipLaunch / runtime error */

const char* clibd_last_error(void);
int clibd_abi_version(void);
/* 16 hex digits identifying the kernel sources this library was built from (clibd_amd.build.csrc_hash()); "unknown" for a
 * build outside clibd_amd/build.py.  The ctypes binding compares it with the sources beside it and refuses a stale library. */
const char* clibd_build_hash(void);

/* ------------------------------------------------------------------------------------------------
 * K2/K4/K5: bf16 MFMA GEMM  out = epilogue(A[M,K] · W[N,K]^T)   (nn.Linear layout)
 * Replaces every torch.nn.Linear on the path: timm Attention.qkv/proj, Mlp.fc1/fc2
 * (reference call sites model/image_encoder.py:40-46,106-107), HF BertSelfAttention.query/key/value,
 * BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense (model/dna_encoder.py:137,
 * model/language_encoder.py:89) and their dgrad in backward.  fp32 accumulate.
 *
 * Epilogue, applied in this order to acc[m,n] (every pointer optional unless noted):
 *   v = acc + rank_u[m,0:8]·rank_v[n,0:8]   (LoRA rank-(4+4) update, bf16 operands, as one extra MFMA k-step;
 *                                            reference: _LoRA_qkv_timm.forward image_encoder.py:40-46,
 *                                            _LoRALayer.forward dna_encoder.py:75-77)
 *   v += bias[n]
 *   if out_pre_bf16:  out_pre_bf16[m,n] = bf16(v); v = float(bf16(v))           (pre-activation, saved for bwd)
 *   if act == CLIBD_ACT_GELU:       v = gelu_erf(v)                             (timm Mlp.act / HF "gelu")
 *   if act == CLIBD_ACT_GELU_GRAD:  v = v * gelu'(aux_bf16[m,n])                (dgrad through GELU)
 *   if act == CLIBD_ACT_GELU_SAVE_GRAD (needs out_pre_bf16): x = float(bf16(v)); out_pre_bf16[m,n] = bf16(gelu'(x));
 *                                   v = gelu_erf(x)    (training forward: the backward then needs no transcendental)
 *   if act == CLIBD_ACT_MUL_AUX:    v = v * aux_bf16[m,n]                       (dgrad through GELU with saved gelu')
 *   if act == CLIBD_ACT_ADD_AUX:    v = v + aux_bf16[m,n]                       (dgrad joining a bf16 residual-gradient stream)
 *   if act == CLIBD_ACT_GELU_SAVE_GRAD_U8 / CLIBD_ACT_MUL_AUX_U8: the two forms above with gelu' kept as ONE BYTE per element:
 *                                   out_pre_bf16 / aux_bf16 then point to uint8 [M,N] (ld_pre / ld_aux in bytes, % 16),
 *                                   code = rint((gelu'(x) + 0.1328125) * 255 / 1.265625), decoded as code * 1.265625 / 255 - 0.1328125
 *                                   (gelu' lies in [-0.129, 1.129]; |error| <= 2.5e-3, the spacing of bf16 in [0.5, 1) is 3.9e-3)
 *   if act == CLIBD_ACT_GELU_SAVE_GRAD_E12 / CLIBD_ACT_MUL_AUX_E12 (ABI 5): the same two forms with gelu' kept as TWELVE bits per element, "e4m7":
 *                                   sign, 4-bit exponent, bf16's 7 mantissa bits — the bf16 value of gelu' with its exponent re-biased to the binades
 *                                   [2^-14, 2) (code e4 = biased bf16 exponent - 112, e4 = 0: zero).  Every bf16 gelu' of magnitude >= 2^-14 is kept BIT FOR
 *                                   BIT (gelu' lies in [-0.129, 1.129]); smaller magnitudes become a signed zero.  out_pre_bf16 / aux_bf16 then point to
 *                                   bytes [M, 3N/2]: eight adjacent columns = three dwords w0 = c0 | c1 << 12 | c2 << 24, w1 = c2 >> 8 | c3 << 4 | c4 << 16 |
 *                                   c5 << 28, w2 = c5 >> 4 | c6 << 8 | c7 << 20; ld_pre / ld_aux in BYTES, % 4, >= 3N/2; N % 8 == 0.
 *   if residual_f32:  v += residual_f32[m,n]
 *   out_bf16[m,n] = bf16(v) ; out_f32[m,n] = v   (either or both)
 *   split_k > 1: only out_f32 allowed; partial sums are atomically added into a caller-zeroed out_f32.
 * Constraints: K % 64 == 0, lda/ldw % 8 == 0, N % 16 == 0, all ld_* % 8 == 0, 16-byte aligned pointers.
 * ------------------------------------------------------------------------------------------------ */
enum { CLIBD_ACT_NONE = 0, CLIBD_ACT_GELU = 1, CLIBD_ACT_GELU_GRAD = 2, CLIBD_ACT_GELU_SAVE_GRAD = 3, CLIBD_ACT_MUL_AUX = 4,
       CLIBD_ACT_ADD_AUX = 5, CLIBD_ACT_GELU_SAVE_GRAD_U8 = 6, CLIBD_ACT_MUL_AUX_U8 = 7,
       CLIBD_ACT_GELU_SAVE_GRAD_E12 = 8, CLIBD_ACT_MUL_AUX_E12 = 9 /* ABI 5 */ };

typedef struct clibd_gemm_epilogue {
    const float* bias;          /* [N] fp32 */
    const void* rank_u;         /* [M,8] bf16, row stride ld_rank_u (>= 8, % 8) */
    const void* rank_v;         /* [N,8] bf16, row stride 8 */
    const void* aux_bf16;       /* [M,N] bf16 (GELU_GRAD input), ld = ld_aux */
    const float* residual_f32;  /* [M,N] fp32, ld = ld_res */
    void* out_pre_bf16;         /* [M,N] bf16, ld = ld_pre */
    void* out_bf16;             /* [M,N] bf16, ld = ld_out_bf16 */
    float* out_f32;             /* [M,N] fp32, ld = ld_out_f32 */
    int32_t act;
    int32_t ld_rank_u, ld_aux, ld_res, ld_pre, ld_out_bf16, ld_out_f32;
    int32_t split_k;            /* >= 1 */
    /* dropout on (acc + rank update + bias), before activation / residual: v *= keep(seed, m*drop_ld + n) / (1-p);
     * drop_thr16 = round(p * 65536) (0 = off), drop_scale = 1/(1-p), drop_ld = N of the forward GEMM.  See clibd_dropout_*. */
    uint32_t drop_seed;
    int32_t drop_thr16;
    float drop_scale;
    int32_t drop_ld;
    /* ABI 3 — the algebraic LayerNorm -> Linear fold of a pre-LN block (norm2 -> mlp.fc1 of timm's Block, image_encoder.py:106-107):
     *   fc1(LN(x)) = rstd_m * ( x . (gamma o W)^T  -  mean_m * s_n ) + b'_n,   s_n = sum_k bf16(gamma_k W[n,k]),  b' = b + W beta.
     * PRODUCER form (the GEMM that writes x: the attention projection): with row_sums != NULL, act NONE, residual_f32, out_f32 AND
     * out_bf16 given, the epilogue also stores out_bf16 = bf16(out_f32 value) and, per 128-column slice j of the output,
     * row_sums[(j * M + m) * 2 + {0, 1}] = sum / sum of squares of row m's fp32 values over that slice (N / 128 slices; 256x256 kernel only).
     * CONSUMER form (fc1 on A = that bf16 copy, W = bf16(gamma o W)): with row_stats != NULL (fp32 [M, 2]: mean, rstd — what
     * clibd_layernorm_fwd's `stats` holds, here from clibd_rowsum_finalize) and col_sum_w = s (fp32 [N]), act GELU_SAVE_GRAD, bias = b':
     * v = rstd_m * (acc - mean_m * s_n) + b'_n before the activation (256x256 kernel only). */
    float* row_sums;
    const float* row_stats;
    const float* col_sum_w;
} clibd_gemm_epilogue;

int clibd_gemm_bf16_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                       const clibd_gemm_epilogue* ep, void* stream);
/* ABI 5 — the same with a STREAM-K TAIL workspace.  The 256x256 kernel runs one persistent workgroup per CU; a launch whose last tile round is at most
 * half full (591 tiles on 256 CUs: 79 in the third round) and whose contraction is long (K >= 1536) cuts that round's tiles into 2-4 K-slices, one per
 * otherwise idle CU: slices 1.. store fp32 partial tiles to the workspace and raise a flag, slice 0 waits, adds them in a fixed order and runs the
 * epilogue.  Results are deterministic; rows of those tiles differ from the plain launch's by fp32 summation order only.  workspace:
 * clibd_gemm_tail_workspace_bytes(M, N, K) bytes (0: this shape has no use for it; <= 48 MiB + 1 KiB), 16-byte aligned, whose first 1 KiB the caller
 * zeroes ONCE (every launch leaves it zero); one workspace per stream.  NULL workspace, or a kind without the form: the plain launch. */
size_t clibd_gemm_tail_workspace_bytes(int M, int N, int K);
int clibd_gemm_bf16_nt_ws(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                          const clibd_gemm_epilogue* ep, void* workspace, size_t workspace_bytes, void* stream);
/* Same product with the K range [hole_k0, hole_k0 + hole_len) of BOTH operands skipped (multiples of 64; 128x128 kernel):
 * for an A whose column segment meets all-zero weights — the adapters' dt projection reads the q and v segments of dqkv
 * and never touches the k segment (a third of the bytes of a latency/HBM-bound skinny product). */
int clibd_gemm_bf16_nt_khole(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int hole_k0, int hole_len,
                             const clibd_gemm_epilogue* ep, void* stream);

/* fp8-forward mode (BASELINE.json configs[4], "fp8 MFMA GEMM path"): the same product with OCP e4m3 operands on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), twice the bf16 MFMA rate.  A [M,K] and W [N,K] are fp8 bytes
 * (lda, ldw, K in bytes; % 16, K % 256 == 0, K >= 512, N % 256 == 0); v = acc * col_scale[n] (the product of the
 * activation's and the weight row's dequantisation factors) then the epilogue of clibd_gemm_bf16_nt, restricted to the
 * forward forms of a transformer layer, all with a bias:  [rank update] -> out_bf16 | GELU_SAVE_GRAD -> out_pre_bf16 =
 * bf16(gelu') and out_bf16 := fp8(gelu(x) * out_fp8_scale) (ld_out_bf16 in bytes; out_fp8_scale > 0 exactly for this form)
 * | [dropout] + residual_f32 -> out_f32.  Replaces the same nn.Linear call sites as clibd_gemm_bf16_nt in forward only; backward stays bf16. */
int clibd_gemm_fp8_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* col_scale,
                      float out_fp8_scale, const clibd_gemm_epilogue* ep, void* stream);

/* Operand image of a frozen nn.Linear weight for clibd_gemm_fp8_nt: per output channel n, s_n = 448 / max_k |w[n,k]|,
 * w_fp8[n,k] = e4m3(w[n,k] * s_n), col_scale[n] = 1 / (s_n * act_scale) where act_scale is the (per-tensor) factor the
 * producer of the activation operand applied before its e4m3 conversion.  w fp32 [N,K] dense, K % 4 == 0. */
int clibd_quantize_rows_fp8(const float* w, int N, int K, float act_scale, void* w_fp8, float* col_scale, void* stream);

/* 8-bit dgrad (ABI 4; BASELINE.json configs[4], numerics switch dgrad = "fp8"): the activation-gradient products dX = dY . W of a frozen
 * nn.Linear (the backward of the same call sites as clibd_gemm_bf16_nt: timm Mlp.fc1 / fc2, Attention.proj, image_encoder.py:106-107; HF
 * BertIntermediate / BertOutput / BertSelfOutput .dense, dna_encoder.py:137) on e4m3 operands.
 *   A [M,K]  = the gradient dY as e4m3 bytes with ONE power-of-two scale per row: A[m,k] = e4m3(dY[m,k] * s_m), a_row_dequant[m] = 1 / s_m
 *              (fp32 [M]; written by clibd_layernorm_bwd_fp8, or inherited from the form below);
 *   W [N,K]  = the TRANSPOSED weight (row n = input channel n of the layer) as clibd_quantize_rows_fp8_bf16 writes it, col_scale[n] its
 *              row's dequantisation factor (times 1 / out_fp8_scale of the producer when A came from the third form).
 * Forms (bias-free, no adapters; M % 4 == 0, N % 256 == 0, K % 256 == 0, K >= 512; lda, ldw, K in bytes):
 *   act NONE                : out_bf16 = bf16(acc * col_scale[n] * a_row_dequant[m])
 *   act ADD_AUX + aux_bf16  : out_bf16 = bf16(acc * col_scale[n] * a_row_dequant[m] + aux[m,n])
 *   act MUL_AUX[_U8] + aux  : (aux = gelu' as bf16, or its one-byte code as in clibd_gemm_bf16_nt's MUL_AUX_U8, ld_aux in bytes % 16)
 *                             out_bf16 := e4m3(acc * col_scale[n] * aux[m,n] * out_fp8_scale) BYTES (ld_out_bf16 in bytes; a_row_dequant
 *                             unused): the dgrad through GELU writes the next dgrad's A operand, which keeps A's row scales; out_fp8_scale
 *                             (> 0 exactly for this form) is a power of two <= 448 / (256 * 1.13 * l1max) with l1max from
 *                             clibd_quantize_rows_fp8_bf16, so that no value saturates (|gelu'| <= 1.13, scaled row maxima < 256).
 *                             ABI 5: with ep->out_pre_bf16 (+ ld_pre, and a_row_dequant, whose values must be powers of two as
 *                             clibd_layernorm_bwd_fp8 writes them) the form ALSO writes out_pre_bf16[m,n] = bf16(acc * col_scale[n] * aux[m,n] *
 *                             a_row_dequant[m]), the true d(fc1 out): under full fine-tuning (trainable base weights) the bf16 weight
 *                             gradient of fc1 contracts it with the layer input while the next dgrad takes the e4m3 bytes. */
int clibd_gemm_fp8_dgrad_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* col_scale,
                            const float* a_row_dequant, float out_fp8_scale, const clibd_gemm_epilogue* ep, void* stream);
/* Operand image of a bf16 matrix (the transposed weight shadow of the bf16 dgrad) with a POWER-OF-TWO scale per row:
 * s_n = 2^(7 - floor(log2 max_k |w[n,k]|)) (1 for an all-zero row), w_fp8[n,k] = e4m3(w[n,k] * s_n), col_scale[n] = 1 / (s_n * act_scale);
 * *l1max = max(*l1max, max_n sum_k |dequantised w8[n,k]|) (fp32 scalar, caller zeroes; may be NULL). */
int clibd_quantize_rows_fp8_bf16(const void* w_bf16, int N, int K, float act_scale, void* w_fp8, float* col_scale, float* l1max, void* stream);

/* bf16 transpose with zero padding: out[C, ld_out] (ld_out >= R) = in[R, C]^T; columns R..ld_out-1 zero.
 * Used to feed the weight-gradient GEMMs (contraction over the token dimension). */
int clibd_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream);
/* same, and colsum[c] += sum_r in[r, c] (fp32, accumulates): the bias gradient of a linear layer rides along with the transpose
 * of dy that its weight gradient needs (full fine-tune mode).  Needs C, ld_in, ld_out multiples of 8, 16-byte aligned bases. */
int clibd_transpose_colsum_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum, void* stream);
/* ABI 5 — the same with a partials workspace (clibd_transpose_colsum_workspace_bytes(ld_out, C) bytes): every row block writes its
 * column sums to the workspace and a second kernel adds them to colsum in row-block order, so the bias gradient repeats bit for bit from
 * run to run (the form above issues one float atomic per block and column).  The trainable heads of the LoRA step take this form. */
size_t clibd_transpose_colsum_workspace_bytes(int ld_out, int C);
int clibd_transpose_colsum_bf16_ws(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* fp32 -> bf16 cast of a contiguous buffer (weights are kept fp32 in the state dict, bf16 shadow copies
 * feed the MFMA path, as torch.autocast does per call in the reference, epoch/train_epoch.py:43). */
int clibd_cast_f32_to_bf16(const float* in, void* out, size_t n, void* stream);
/* out[C,R] bf16 = transpose(in[R,C] fp32) */
int clibd_cast_transpose_f32_to_bf16(const float* in, int R, int C, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (timm Block.norm1/norm2/norm eps=1e-6; HF BertEmbeddings/BertSelfOutput/BertOutput
 * LayerNorm eps=1e-12), fp32 statistics.  x fp32 [M,H] -> y (bf16 and/or fp32), optional saved
 * (mean, rstd) fp32 [M,2], optional LoRA down-projection t[m,0:8] = bf16(y[m,:]) · lora_a[8,H]^T
 * (reference: linear_a_q/linear_a_v image_encoder.py:41-42, w_a dna_encoder.py:76) emitted as bf16 [M,8].
 * H % 64 == 0, H <= 1024.
 * ------------------------------------------------------------------------------------------------ */
int clibd_layernorm_fwd(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                        void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                        void* stream);
/* Dropout (HF BERT train mode, p = 0.1: BertEmbeddings.dropout, BertSelfOutput.dropout, BertOutput.dropout,
 * BertSelfAttention.dropout on the probabilities).  Masks are a pure function of (seed, element index) — lowbias32 hash of
 * (index >> 1) ^ seed, 16 bits per element, keep iff bits >= round(p*65536) — so nothing is stored and the backward
 * recomputes them.  Element index: row*H + col for [M,H] activations; ((b*heads+h)*S + q)*256 + key for attention.
 * _drop variants: y = dropout(LN(x)) (embeddings);  backward: the bf16 output only is masked (it is the gradient that
 * enters the dgrad GEMM of the dropped dense output; the fp32 output is the residual-path gradient). */
/* fp8-forward mode: additionally (or only: y_bf16 / y_f32 may both be NULL) y_fp8[m,c] = e4m3(y * fp8_scale), saturating
 * at +-448 — the operand of the next clibd_gemm_fp8_nt.  The LoRA down-projection still reads bf16(y). */
int clibd_layernorm_fwd_fp8(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                            void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                            uint32_t drop_seed, int drop_thr16, float drop_scale, void* y_fp8, float fp8_scale, void* stream);
int clibd_layernorm_fwd_drop(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                             void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                             uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
/* dx = LN'(dy) [+ dres]; dy is bf16 (dy_bf16) or fp32 (dy_f32), exactly one non-null.
 * Outputs dx_f32 and/or dx_bf16. gamma is frozen on the LoRA path, so no dgamma/dbeta here. */
int clibd_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                        const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                        void* dx_bf16, void* stream);
int clibd_layernorm_bwd_drop(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                             const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                             void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
/* The same backward with the RESIDUAL GRADIENT carried in bf16 (frozen-base / LoRA mode): dx = LN'(dy) [+ dres_bf16];
 * dx_res_bf16 (optional) = bf16(dx), the gradient of the residual sum handed to the next block; dx_bf16 (optional) = bf16(dx x the
 * dense branch's dropout mask) when drop_thr16 > 0, else the same values.  10 instead of 16 bytes per element of a pre-LN block
 * (the reference's autograd keeps this stream in fp32: the rounding it adds is budgeted in DESIGN.md §4). */
int clibd_layernorm_bwd_res16(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                              int M, int H, const void* dres_bf16, void* dx_res_bf16, void* dx_bf16, uint32_t drop_seed,
                              int drop_thr16, float drop_scale, void* stream);
/* The general form (round 4; full fine-tune on the bf16 residual-gradient stream): every optional operand of the three entry
 * points above in one call — residual gradient in fp32 (dres_f32) OR bf16 (dres_bf16), outputs dx_f32 / dx_res_bf16 / dx_bf16
 * (at least one), dropout on the dx_bf16 copy, and dgamma / dbeta (both or neither; accumulated).  Same kernel, same arithmetic. */
int clibd_layernorm_bwd_any(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                            int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                            void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, float* dgamma, float* dbeta,
                            void* stream);
/* 8-bit dgrad (ABI 4): clibd_layernorm_bwd_any without parameter gradients, and additionally the copy the dense branch's dgrad consumes
 * (dropout mask applied, as dx_bf16 — which may now be NULL) as e4m3 bytes with one power-of-two scale per row:
 *   s_m = 2^(7 - floor(log2 max_c |v[m,c]|))  (1 for an all-zero row),  dx_fp8[m,c] = e4m3(v[m,c] * s_m),  row_dequant[m] = 1 / s_m.
 * dx_fp8 uint8 [M,H], row_dequant fp32 [M]: the A operand of clibd_gemm_fp8_dgrad_nt. */
int clibd_layernorm_bwd_fp8(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                            int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                            void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* dx_fp8, float* row_dequant,
                            void* stream);
/* ABI 5 — the same with the LayerNorm parameter gradients (dgamma[c] += sum_rows dy xhat, dbeta[c] += sum_rows dy, as clibd_layernorm_bwd_pg):
 * the 8-bit dgrad under full fine-tuning (model_config.disable_lora, the reference's final BIOSCAN-1M / 5M recipe): the dgrad takes the e4m3
 * rows, the bf16 weight gradient takes dx_bf16, both written in this one pass. */
int clibd_layernorm_bwd_fp8_pg(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                               int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                               void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* dx_fp8, float* row_dequant,
                               float* dgamma, float* dbeta, void* stream);
/* full fine-tune mode: the same backward that also accumulates the parameter gradients it has the operands for
 *   dgamma[c] += sum_m dy[m,c] * xhat[m,c],  dbeta[c] += sum_m dy[m,c]     (fp32 [H], caller zeroes once per step). */
int clibd_layernorm_bwd_pg(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                           int M, int H, const float* dres_f32, float* dx_f32, void* dx_bf16, uint32_t drop_seed, int drop_thr16,
                           float drop_scale, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3: multi-head attention, head dim 64, whole sequence per workgroup (S <= 256: ViT 197, BarcodeBERT 133,
 * BERT-small 20).  qkv bf16 [B*S, 3*H] packed [q | k | v] per row (timm Attention.forward layout
 * image_encoder.py:20-24; HF BertSelfAttention with q/k/v weights concatenated).
 * key_mask (optional) int32 [B,S], 1 = attend, 0 = masked (HF extended attention mask,
 * language_encoder.py:89 via BertModel).  out bf16 [B*S, H].
 * Query prefix: only query rows [0, nq) of every sequence are evaluated (nq = S: everything; nq = 1: the [CLS]-only
 * attention of the last ViT block, whose other outputs the reference computes and discards, image_encoder.py:107 ->
 * timm pools token 0).  out / dout hold `out_seq` (>= nq) rows per sequence: row (b*out_seq + q).  The backward writes
 * the full dqkv [B*S,3H]: dq of rows >= nq is zero, dk/dv collect the active queries only.
 * ------------------------------------------------------------------------------------------------ */
int clibd_attention_fwd(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out,
                        int nq, int out_seq, void* stream);
int clibd_attention_bwd(const void* qkv, const void* dout, int B, int S, int nheads, const int32_t* key_mask,
                        void* dqkv, int nq, int dout_seq, void* stream);

/* Single-pass backward (frozen-base training, full sequences): the forward also saves, per (head, query), the log2-domain
 * log-sum-exp of the un-dropped scores — lse fp32 [B * nheads, S] — and the bf16 residual of its output rounding — o_lo bf16
 * [B * S, H], out + o_lo = the fp32 output to 2^-17 — so that the backward evaluates every score, exponential and dP ONCE:
 *   P = exp2(c2 s - lse), delta = dO . (out + o_lo), dS = P o (dP - delta), dV += P^T dO, dK += dS^T Q, dQ += dS K
 * (one 8-wave workgroup per head, K / V / Q / dO resident in LDS, dK / dV in the key-owning wave's accumulators, dQ from a dS
 * exchange through LDS: no atomics, deterministic).  Same results as clibd_attention_bwd up to rounding points.
 * clibd_attention_fwd_save: nq = S; key_mask allowed in the forward.  clibd_attention_bwd_sp: no key mask, S <= 224. */
int clibd_attention_fwd_save(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out, uint32_t drop_seed,
                             int drop_thr16, float drop_scale, float* lse, void* o_lo, void* stream);
int clibd_attention_bwd_sp(const void* qkv, const void* dout, const void* out, const void* o_lo, const float* lse, int B, int S,
                           int nheads, void* dqkv, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
/* fp8-forward mode: the attention output leaves as e4m3(o * out_fp8_scale) bytes [B*out_seq, H] (operand of the projection
 * clibd_gemm_fp8_nt; the backward recomputes what it needs from qkv, so no bf16 copy is kept). */
int clibd_attention_fwd_fp8(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out_fp8,
                            int nq, int out_seq, uint32_t drop_seed, int drop_thr16, float drop_scale, float out_fp8_scale,
                            void* stream);
/* same with dropout on the attention probabilities (see clibd_layernorm_fwd_drop for the mask definition) */
int clibd_attention_fwd_drop(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out,
                             int nq, int out_seq, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
int clibd_attention_bwd_drop(const void* qkv, const void* dout, int B, int S, int nheads, const int32_t* key_mask,
                             void* dqkv, int nq, int dout_seq, uint32_t drop_seed, int drop_thr16, float drop_scale,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6: LoRA (rank 4 on q and v; reference image_encoder.py:13-46, dna_encoder.py:68-77,
 * language_encoder.py:24-33).  Adapter parameters: a_q,a_v fp32 [4,H] (nn.Linear(H,4).weight),
 * b_q,b_v fp32 [H,4] (nn.Linear(4,H).weight); no alpha/r scaling.
 *
 * clibd_lora_pack (per step, parameters are trainable) builds the bf16 operand images:
 *   v_fwd [3H,8]: rank_v of the QKV GEMM   q rows (B_q[n,:],0000) | k rows 0 | v rows (0000,B_v[n-2H,:])
 *   v_bwd [H,8] : rank_v of the QKV dgrad  (A_q[0:4,k], A_v[0:4,k])
 *   a_cat [8,H] : [A_q; A_v], the LayerNorm-fused down projection t = x·a_cat^T
 *   w_dt  [16,3H]: GEMM weight giving dt[m,0:8] = [dq·B_q | dv·B_v] (cols 8..15 zero) from dqkv[M,3H]
 * clibd_lora_wgrad: parameter gradients, contraction over the M tokens (accumulates, caller zeroes):
 *   dB_q[n,r] += sum_m dq[m,n] t[m,r]      dB_v[n,r] += sum_m dv[m,n] t[m,4+r]
 *   dA_q[r,k] += sum_m dt[m,r] x[m,k]      dA_v[r,k] += sum_m dt[m,4+r] x[m,k]
 *   with dq = dqkv[:,0:H], dv = dqkv[:,2H:3H], x = adapter input bf16 [M,H], t bf16 [M,8], dt bf16 [M,ld_dt].
 * ------------------------------------------------------------------------------------------------ */
int clibd_lora_pack(const float* a_q, const float* a_v, const float* b_q, const float* b_v, int H,
                    void* v_fwd_bf16, void* v_bwd_bf16, void* a_cat_bf16, void* w_dt_bf16, void* stream);
/* t[M,8] = bf16(x[M,H] . a_cat[8,H]^T): the adapters' down-projection on its own (the LayerNorm kernels fuse it for the first
 * rank-(4+4) slot; adapters with 4 < r <= 8 — the reference accepts any r > 0, image_encoder.py:50-53, dna_encoder.py:80-88 —
 * carry ranks 5..8 in a second slot whose t comes from here).  x bf16 row stride ld_x (% 8), H % 8 == 0. */
int clibd_lora_down_proj(const void* x_bf16, int ld_x, const void* a_cat_bf16, int M, int H, void* t_bf16, void* stream);
/* workspace (ABI 3; may be NULL): clibd_lora_workspace_bytes(M, H) bytes, 16-byte aligned.  With it the MFMA forms (M % 32 == 0,
 * H % 128 == 0) write per-workgroup partials there and a last kernel adds them in a fixed order: no contended float atomics (38-54 us
 * of a 286-377 us backward at M = 403 456 / 272 384) and gradients that are bit-reproducible run to run; NULL keeps the float atomics. */
size_t clibd_lora_workspace_bytes(int M, int H);
int clibd_lora_wgrad(const void* dqkv, int ld_dqkv, const void* x_bf16, const void* t_bf16, const void* dt_bf16,
                     int ld_dt, int M, int H, float* dA_q, float* dA_v, float* dB_q, float* dB_v,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The adapters' whole backward in one call: dt[M, 0:16] = dqkv . w_dt^T (bf16; columns 8..15 zero: the rank-8 operand of the QKV
 * dgrad), then the four parameter gradients of clibd_lora_wgrad (accumulating).  For large M (whole 32-token slabs, H % 128 == 0)
 * dq and dv are read ONCE: a first kernel produces dt and dB together, a second one dA from x and dt; otherwise it is the skinny
 * GEMM (clibd_gemm_bf16_nt_khole against w_dt) followed by clibd_lora_wgrad.  Replaces the autograd of _LoRA_qkv_timm.forward /
 * _LoRALayer.forward (image_encoder.py:40-46, dna_encoder.py:75-77) with respect to the adapter weights. */
int clibd_lora_backward(const void* dqkv, int ld_dqkv, const void* x_bf16, const void* t_bf16, const void* w_dt_bf16,
                        void* dt_bf16, int ld_dt, int M, int H, float* dA_q, float* dA_v, float* dB_q, float* dB_v,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embeddings.
 * K1 patch gather (timm PatchEmbed: Conv2d(3,H,16,16) == GEMM over 16x16x3 patches):
 *   patches bf16 [B*196, 768] with k = c*256 + py*16 + px  from image fp32 [B,3,224,224].
 * ViT token assembly: tok fp32 [B,S,H]: row 0 = cls + pos[0]; rows 1.. = patch_proj[b*(S-1)+p] + pos[1+p]
 *   (timm VisionTransformer._pos_embed).
 * BERT embeddings: word[id] + position[s] + token_type[tt] (HF BertEmbeddings) -> fp32 [B*S,H] (pre-LN sum).
 * ------------------------------------------------------------------------------------------------ */
int clibd_patchify(const float* image, int B, void* patches_bf16, void* stream);
/* The same from the dataset's image BYTES (uint8 [B,3,224,224], 8-byte aligned): value = fp32(u8) / 255 (IEEE division, what the
 * reference's ToTensor computes on the host, util/dataset.py:185-195), so the result equals clibd_patchify(u8.float() / 255) bit
 * for bit while a quarter of the bytes cross PCIe (train_epoch.py:26-32 copies the fp32 tensor every step).  ABI version 3.
 * Scope: this is the ARITHMETIC of ToTensor only.  The reference's pipeline continues on the host after ToTensor — Resize(256,
 * antialias), RandomResizedCrop / CenterCrop(224), flips and rotation on float tensors (util/dataset.py) — so the tensor it sends is not u8 / 255
 * of any stored byte image; the uint8 path applies when the augmentation pipeline emits uint8 224 x 224 crops (or moves to the device), and is not a
 * drop-in for the reference's float pipeline. */
int clibd_patchify_u8(const unsigned char* image, int B, void* patches_bf16, void* stream);
/* LayerNorm -> Linear fold (ABI 3; see clibd_gemm_epilogue.row_sums / row_stats).  clibd_rowsum_finalize: stats[m] = (mean, rstd) of
 * row m (eps as the LayerNorm's) from the producer GEMM's per-slice sums row_sums[slices][M][2], H = 128 * slices — the `stats` of
 * clibd_layernorm_fwd, consumed by the fold's consumer GEMM and by the unchanged clibd_layernorm_bwd.  clibd_ln_fold_weights (once per
 * weight version): wg = bf16(w o gamma) [N,K], col_sum_w[n] = sum_k float(wg[n,k]), bias_folded = bias + w beta (bias may be NULL).
 * Replaces timm Block's `self.mlp.fc1(self.norm2(x))` pair (vision_transformer.py via image_encoder.py:106-107) without an HBM pass for norm2. */
int clibd_rowsum_finalize(const float* row_sums, int slices, int M, int H, float eps, float* stats, void* stream);
int clibd_ln_fold_weights(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K, void* wg_bf16,
                          float* col_sum_w, float* bias_folded, void* stream);
int clibd_vit_assemble_tokens(const float* proj, const float* cls, const float* pos, int B, int S, int H, float* tok,
                              void* stream);
int clibd_bert_embed(const int64_t* ids, const int64_t* token_type, int B, int S, int H, int vocab,
                     const float* word, const float* pos, const float* type, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K7: DNA head tail (model/dna_encoder.py:137): out[b,:] = mean_s softmax(logits[b,s,:]).
 * logits bf16 [B*S, C]; out fp32 [B,C].  bwd: dlogits bf16 [B*S,C] from dout fp32 [B,C].  C % 64 == 0, C<=1024.
 * ------------------------------------------------------------------------------------------------ */
int clibd_softmax_mean_fwd(const void* logits, int B, int S, int C, float* out, void* stream);
int clibd_softmax_mean_bwd(const void* logits, const float* dout, int B, int S, int C, void* dlogits,
                           void* stream);
/* token mean (model/language_encoder.py:89 `.last_hidden_state.mean(dim=1)`): x fp32 [B,S,H] -> bf16/fp32 [B,H] */
int clibd_token_mean_fwd(const float* x, int B, int S, int H, void* out_bf16, void* stream);
int clibd_token_mean_bwd(const float* dout, int B, int S, int H, float* dx, void* stream);
/* dx = dy * gelu'(pre), bf16 in/out, n % 4 == 0 (HF BertPredictionHeadTransform: dense -> gelu -> LayerNorm) */
int clibd_gelu_bwd_bf16(const void* dy, const void* pre, size_t n, void* dx, void* stream);
/* column sums of a bf16 [M,N] matrix into fp32 [N] (bias gradients of the trainable heads; accumulates) */
int clibd_colsum_bf16(const void* x, int ld, int M, int N, float* out, void* stream);
/* gather / scatter the [CLS] rows: x fp32 [B,S,H] row 0 <-> [B,H] */
int clibd_gather_rows(const float* x, int B, int S, int H, float* out, void* stream);
int clibd_scatter_rows_bf16(const float* dcls, int B, int S, int H, void* dx_bf16, float* dx_f32, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8: row L2 normalisation (F.normalize(p=2, dim=-1, eps=1e-12), model/simple_clip.py:45,58,60).
 * ------------------------------------------------------------------------------------------------ */
int clibd_l2norm_fwd(const float* x, int N, int D, float* y, float* inv_norm, void* stream);
int clibd_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int N, int D, float* dx,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9: all-pairs similarity + soft-target cross-entropy, row-block form, one direction of one modality pair
 * (model/loss_func.py:41-69 ContrastiveLoss.forward, :138-201 ClipLoss.forward, :19-22 targets):
 *   S[i,j] = scale * <x_i, y_j>    x: fp32 [Nx,D] = the rows this rank owns (global row offset row0),
 *                                   y: fp32 [N,D]  = all rows of the other modality (unit-norm inputs)
 *   T[i,j] = (labels[row0+i] == labels[j])                              (never materialised)
 *   *loss_sum += sum_i ( LSE_j S[i,:] * sum_j T[i,j] - sum_j T[i,j] S[i,j] )   (= sum_i CE(S[i,:], T[i,:]);
 *                the reference's nn.CrossEntropyLoss() mean is loss_sum / N, the caller applies it)
 * The product runs on bf16 MFMA with split operands (hi+lo, three partial products in one K=3D GEMM), which
 * is ~fp32-accurate like the reference's fp32 matmul outside autocast.  Full N x N semantics: Nx=N,row0=0,
 * called once per direction (x=a,y=b) and (x=b,y=a).  Data-parallel: each rank passes its own row block.
 * bwd must follow fwd on the same workspace:  g = weight * (*weight_scale) * dloss_sum/dS;
 *   dx [Nx,D] += scale * g·y,  dy [N,D] += scale * g^T·x  (fp32, ACCUMULATED),  *dscale += sum g∘(x·y^T).
 * D % 64 == 0; any Nx, N.
 * ABI 5: *loss_sum and *dscale are accumulated from per-row terms (kept in the workspace) by a one-workgroup kernel in a fixed order —
 * no float atomics on the loss path, so the loss value and d(logit_scale) repeat bit for bit from run to run.
 * ------------------------------------------------------------------------------------------------ */
size_t clibd_softce_workspace_bytes(int Nx, int N, int D);
int clibd_softce_rows_fwd(const float* x, const float* y, const int64_t* labels, int Nx, int N, int D, int row0,
                          const float* scale /* device scalar */, float* loss_sum, void* workspace, size_t workspace_bytes, void* stream);
int clibd_softce_rows_bwd(const int64_t* labels, int Nx, int N, int D, int row0, const float* scale, float weight,
                          const float* weight_scale /* optional device scalar multiplied into weight */, float* dx, float* dy, float* dscale, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * K10 ("next" row §8f-2): exact fp32 inner-product top-k — faiss.IndexFlatIP.search(query, k) in the reference's eval
 * path (util/util.py:521-528, make_prediction; callers L2-normalise first, clibd_l2norm_fwd).  q [Q,D], keys [Nk,D]
 * fp32; out_idx int64 [Q,k] (ties -> lower key index), out_sim fp32 [Q,k]; 1 <= k <= 8; D % 4 == 0.  Scores use the
 * fp32-input MFMA (exact fp32 products, fmaf chain) and go from the accumulators into running top-8 lists in registers: the
 * [Q,Nk] score matrix is never written.  workspace: clibd_topk_ip_workspace_bytes(Q, Nk) (per-key-split lists, <= 16 KiB per query).
 * ------------------------------------------------------------------------------------------------ */
size_t clibd_topk_ip_workspace_bytes(int Q, int Nk);
int clibd_topk_ip(const float* q, const float* keys, int Q, int Nk, int D, int k, int64_t* out_idx, float* out_sim,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The same search, pre-filtered (round 4): indices and similarities BIT-IDENTICAL to clibd_topk_ip at the bf16 MFMA rate.
 * clibd_topk_prepare_keys: once per key bank: keys_bf16 [Nk, D] (the bf16 image) and max_norm[0] = max_n ||key_n||.
 * clibd_topk_ip_fast: approximate scores bf16(q) . bf16(key) streamed into running top-8 lists (no score matrix), then every listed key
 * whose approximate score lies within 2 eps of the k-th largest one — eps = 0.0045 ||q|| max ||key|| bounds |approximate - exact| —
 * is re-scored with the exact kernel's arithmetic (k-ordered fp32 fmaf chain) and the top k of those are returned (ties -> lower
 * index).  The true top k are provably among the re-scored keys unless a list was full above that line: such queries are flagged
 * overflow[q] = 1 (their outputs are then unspecified) and the caller re-runs them through clibd_topk_ip.  D % 64 == 0, Nk < 2^24.
 * Replaces the same faiss.IndexFlatIP(...).search call (reference util/util.py:521-528). */
int clibd_topk_prepare_keys(const float* keys, int Nk, int D, void* keys_bf16, float* max_norm, void* stream);
size_t clibd_topk_ip_fast_workspace_bytes(int Q, int Nk, int D);
int clibd_topk_ip_fast(const float* q, const float* keys, const void* keys_bf16, const float* max_norm, int Q, int Nk, int D, int k,
                       int64_t* out_idx, float* out_sim, int32_t* overflow, void* workspace, size_t workspace_bytes, void* stream);

/* f1 batch contract: k-mer tokenisation (model/dna_encoder.py:53-63 get_sequence_pipeline, util/util.py:77-98).
 * seq_u8 [B,L] ASCII, already truncated / 'N'-padded to L (660); out int64 [B, 1 + L/k]: leading 0, then 3 + base-4 value
 * (A0 C1 G2 T3, product('ACGT', repeat=k) order) or 2 (<UNK>) for a k-mer with any other character. */
int clibd_kmer_tokenize(const void* seq_u8, int B, int L, int k, int64_t* out, void* stream);

/* Split-K GEMM through a partials workspace: out[M,N] (fp32, dense, ld_out == N) = (accumulate ? out : 0) + A[M,K] . W[N,K]^T.
 * For products with few output tiles and a very long contraction — the weight gradients dW = dY^T X of the full
 * fine-tune mode (autograd of every nn.Linear on the path), A = dY^T [N_w, tokens], W = X^T [K_w, tokens].
 * The 256x256 kernel runs one (tile, K-slice) work item per CU and stores fp32 partial tiles (deterministic: no atomics);
 * a second kernel sums the slices.  N % 256 == 0, K % 128 == 0, K >= 512; workspace >= clibd_gemm_splitk_workspace_bytes(M,N). */
size_t clibd_gemm_splitk_workspace_bytes(int M, int N);
int clibd_gemm_bf16_nt_splitk(const void* A, int lda, const void* W, int ldw, int M, int N, int K, float* out_f32, int ld_out,
                              int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The same product with both operands read in place: out[Na,Nb] (+)= A[M,Na]^T · B[M,Nb] (bf16, contraction over the token
 * rows through transposing LDS reads; the weight gradient dW = dY^T X without materialising dY^T and X^T).
 * colsum_a (optional, fp32 [Na], accumulates atomically): column sums of A in the same pass (the bias gradient db = dY^T 1).
 * M % 128 == 0, M >= 256, Na % 256 == 0, Nb % 256 == 0, lda / ldb % 8 == 0; workspace >= clibd_gemm_splitk_workspace_bytes(Na, Nb). */
int clibd_gemm_bf16_tn_splitk(const void* A, int lda, const void* B, int ldb, int M, int Na, int Nb, float* out_f32, int ld_out,
                              int accumulate, float* colsum_a, void* workspace, size_t workspace_bytes, void* stream);

/* ---- full fine-tune mode (model_config.disable_lora): parameter gradients that are not GEMM-shaped.
 * Every output ACCUMULATES (atomicAdd) into fp32 buffers the caller zeroes once per step.
 * Replaces the autograd of nn.LayerNorm (timm Block.norm1/2, VisionTransformer.norm; HF Bert*LayerNorm), of nn.Embedding
 * (HF BertEmbeddings) and of the position / class-token parameters (timm VisionTransformer._pos_embed). */
/* dgamma[c] += sum_m dy[m,c] * (x[m,c]-mean[m]) * rstd[m];  dbeta[c] += sum_m dy[m,c].  dy [M,H] bf16 or fp32 (dy_is_f32),
 * row stride ld_dy; x fp32 [M,H]; stats fp32 [M,2]; drop_thr16 > 0: dy is first multiplied by the dropout factor of
 * element m*H+c (LayerNorm whose output went through dropout: HF BertEmbeddings).  H <= 1024. */
int clibd_layernorm_param_grads(const void* dy, int dy_is_f32, int ld_dy, const float* x, const float* stats, int M, int H,
                                float* dgamma, float* dbeta, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
/* y[i] = x[i] * dropout_factor(seed, i) (x, y fp32 [n], may alias): gradient through a dropout whose mask index is the flat
 * element index (y = dropout(LN(e)) of HF BertEmbeddings, when the embedding tables are trainable). */
int clibd_dropout_apply_f32(const float* x, size_t n, float* y, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream);
/* out[r] += sum_b x[b, r]  (x fp32 [B, R]): position-embedding and class-token gradients. */
int clibd_batch_sum_f32(const float* x, int B, size_t R, float* out, void* stream);
/* dword[ids[m], :] += de[m, :];  dtype[token_type[m] (0 if NULL), :] += de[m, :]   (de fp32 [M,H]; either table may be NULL). */
int clibd_bert_embed_bwd(const int64_t* ids, const int64_t* token_type, const float* de, int M, int H, int vocab, int type_vocab,
                         float* dword, float* dtype, void* stream);
/* out bf16 [B*(s1-s0), H] = rows s0..s1-1 of every sequence of x fp32 [B,S,H] (patch rows of the ViT token gradient). */
int clibd_slice_rows_cast_bf16(const float* x, int B, int S, int H, int s0, int s1, void* out, void* stream);

/* fused AdamW step on a flat fp32 parameter bucket (torch.optim.AdamW semantics, scripts/train_cl.py:221):
 * p,g,m,v [n]; g is multiplied by grad_scale first (1/world_size folding etc.). */
int clibd_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIBD_HIP_H */
