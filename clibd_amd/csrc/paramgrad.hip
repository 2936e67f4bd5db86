// Parameter gradients of the full fine-tune mode (model_config.disable_lora): everything the LoRA step never
// needs because the base encoders are frozen there.  Weight gradients dW = dY^T X go through the NT GEMM on transposed
// operands (host side: clibd_amd/engine.py::linear_wgrad); this file holds the reductions that are not GEMM-shaped:
//   * LayerNorm gamma / beta gradients (replaces autograd of nn.LayerNorm in timm Block / HF BertLayer),
//   * sums over the batch of a [B, R] fp32 tensor (position-embedding / class-token gradients),
//   * the scatter of the embedding gradient into the word / token-type tables (autograd of nn.Embedding),
//   * a row-range slice + bf16 cast (patch rows of the ViT token gradient, feeding the patch-embedding weight gradient).
// All outputs ACCUMULATE (atomicAdd) into fp32 buffers the caller zeroes once per step (the flat gradient bucket).
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

__device__ __forceinline__ float ld_as_f32(const float* p) { return *p; }
__device__ __forceinline__ float ld_as_f32(const unsigned short* p) { return bf2f(*p); }

constexpr int PG_ROWS = 64;

// one block: PG_ROWS rows x all H columns (H <= 1024: up to 4 columns per thread); coalesced along columns
template <typename DY>
__global__ __launch_bounds__(256) void ln_param_grads_kernel(const DY* __restrict__ dy, int ld_dy, const float* __restrict__ x,
                                                             const float* __restrict__ stats, int M, int H,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, unsigned drop_seed,
                                                             int drop_thr16, float drop_scale) {
    const int r0 = blockIdx.x * PG_ROWS, r1 = min(r0 + PG_ROWS, M);
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < H) {
                float g = ld_as_f32(dy + (size_t)r * ld_dy + c);
                if (drop_thr16 > 0) g *= drop_one(drop_seed, (unsigned)r * (unsigned)H + (unsigned)c, (unsigned)drop_thr16, drop_scale);
                sg[j] += g * ((x[(size_t)r * H + c] - mean) * rstd);
                sb[j] += g;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < H) {
            atomicAdd(dgamma + c, sg[j]);
            atomicAdd(dbeta + c, sb[j]);
        }
    }
}

// out[r] += sum over b in this block's batch chunk of x[b, r]
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, int B, size_t R, float* __restrict__ out, int bchunk) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int b0 = blockIdx.y * bchunk, b1 = min(b0 + bchunk, B);
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += x[(size_t)b * R + r];
    atomicAdd(out + r, s);
}

// word table: dword[ids[m], :] += de[m, :] (scatter);  token-type table (vocabulary 2, HF BERT): block-level partial sums
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const long long* __restrict__ ids, const long long* __restrict__ tt,
                                                             const float* __restrict__ de, int M, int H, int vocab, int type_vocab,
                                                             float* __restrict__ dword, float* __restrict__ dtype) {
    const int r0 = blockIdx.x * PG_ROWS, r1 = min(r0 + PG_ROWS, M);
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
        long long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        long long ty = tt ? tt[r] : 0;
        ty = ty < 0 ? 0 : (ty >= type_vocab ? type_vocab - 1 : ty);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < H) {
                const float g = de[(size_t)r * H + c];
                if (dword) atomicAdd(dword + (size_t)id * H + c, g);
                if (dtype) {
                    if (ty == 0) t0[j] += g;
                    else if (ty == 1) t1[j] += g;
                    else atomicAdd(dtype + (size_t)ty * H + c, g);
                }
            }
        }
    }
    if (dtype) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < H) {
                atomicAdd(dtype + c, t0[j]);
                if (type_vocab > 1) atomicAdd(dtype + (size_t)H + c, t1[j]);
            }
        }
    }
}

// out[b * (s1 - s0) + (s - s0), :] = bf16(x[b, s, :]) for s in [s0, s1)
__global__ __launch_bounds__(256) void slice_rows_cast_kernel(const float* __restrict__ x, int B, int S, int H, int s0, int s1,
                                                              unsigned short* __restrict__ out) {
    const int ns = s1 - s0;
    const size_t total = (size_t)B * ns * (H / 2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int hp = (int)(i % (H / 2));
        const size_t row = i / (H / 2);
        const int b = (int)(row / ns), s = (int)(row % ns) + s0;
        const float2 v = *(const float2*)(x + ((size_t)b * S + s) * H + 2 * hp);
        *(unsigned*)(out + row * H + 2 * hp) = pack2bf(v.x, v.y);
    }
}

// y[i] = x[i] * dropout_factor(seed, i): the gradient through y = dropout(.) of an [M,H] activation (element index row*H+col)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y, unsigned seed,
                                                            unsigned thr16, float scale) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * blockDim.x * 2) {
        float f0, f1;
        drop_pair(seed, (unsigned)i, thr16, scale, f0, f1);
        y[i] = x[i] * f0;
        if (i + 1 < n) y[i + 1] = x[i + 1] * f1;
    }
}

}  // namespace clibd

using namespace clibd;

extern "C" int clibd_layernorm_param_grads(const void* dy, int dy_is_f32, int ld_dy, const float* x, const float* stats, int M, int H,
                                           float* dgamma, float* dbeta, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    if (!dy || !x || !stats || !dgamma || !dbeta) return set_error(CLIBD_EINVAL, "layernorm_param_grads: null pointer");
    if (M <= 0 || H <= 0 || H > 1024 || ld_dy < H) return set_error(CLIBD_EINVAL, "layernorm_param_grads: bad shape (H <= 1024)");
    const dim3 grid((M + PG_ROWS - 1) / PG_ROWS);
    if (dy_is_f32)
        hipLaunchKernelGGL(ln_param_grads_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, ld_dy, x, stats, M, H,
                           dgamma, dbeta, drop_seed, drop_thr16, drop_scale);
    else
        hipLaunchKernelGGL(ln_param_grads_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy, ld_dy, x,
                           stats, M, H, dgamma, dbeta, drop_seed, drop_thr16, drop_scale);
    return check_launch("layernorm_param_grads");
}

extern "C" int clibd_batch_sum_f32(const float* x, int B, size_t R, float* out, void* stream) {
    if (!x || !out || B <= 0 || R == 0) return set_error(CLIBD_EINVAL, "batch_sum: bad args");
    const int chunks = B >= 64 ? 8 : 1;
    const int bchunk = (B + chunks - 1) / chunks;
    const size_t gx = (R + 255) / 256;
    if (gx > 0x7fffffffull) return set_error(CLIBD_EINVAL, "batch_sum: R too large");
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)gx, (unsigned)((B + bchunk - 1) / bchunk)), dim3(256), 0, (hipStream_t)stream, x, B, R,
                       out, bchunk);
    return check_launch("batch_sum");
}

extern "C" int clibd_bert_embed_bwd(const int64_t* ids, const int64_t* token_type, const float* de, int M, int H, int vocab, int type_vocab,
                                    float* dword, float* dtype, void* stream) {
    if (!ids || !de || (!dword && !dtype)) return set_error(CLIBD_EINVAL, "bert_embed_bwd: null pointer");
    if (M <= 0 || H <= 0 || H > 1024 || vocab <= 0 || type_vocab <= 0) return set_error(CLIBD_EINVAL, "bert_embed_bwd: bad shape (H <= 1024)");
    hipLaunchKernelGGL(bert_embed_bwd_kernel, dim3((M + PG_ROWS - 1) / PG_ROWS), dim3(256), 0, (hipStream_t)stream, (const long long*)ids,
                       (const long long*)token_type, de, M, H, vocab, type_vocab, dword, dtype);
    return check_launch("bert_embed_bwd");
}

extern "C" int clibd_slice_rows_cast_bf16(const float* x, int B, int S, int H, int s0, int s1, void* out, void* stream) {
    if (!x || !out || B <= 0 || S <= 0 || H <= 0 || (H & 1) || s0 < 0 || s1 > S || s0 >= s1) return set_error(CLIBD_EINVAL, "slice_rows_cast: bad args");
    if (!aligned16(x) || !aligned16(out)) return set_error(CLIBD_EINVAL, "slice_rows_cast: alignment");
    hipLaunchKernelGGL(slice_rows_cast_kernel, dim3(grid_for((size_t)B * (s1 - s0) * (H / 2))), dim3(256), 0, (hipStream_t)stream, x, B, S, H,
                       s0, s1, (unsigned short*)out);
    return check_launch("slice_rows_cast");
}

extern "C" int clibd_dropout_apply_f32(const float* x, size_t n, float* y, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    if (!x || !y || n == 0 || n >= (1ull << 32)) return set_error(CLIBD_EINVAL, "dropout_apply: bad args (n < 2^32)");
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "dropout_apply: bad dropout threshold");
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, x, n, y, drop_seed,
                       (unsigned)drop_thr16, drop_scale);
    return check_launch("dropout_apply");
}
