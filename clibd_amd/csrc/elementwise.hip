// HBM/latency-bound helpers of the CLIBD step: embeddings, head tails, L2 normalisation, reductions, AdamW.
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

// ---- K1 front: image fp32 [B,3,224,224] -> patch matrix bf16 [B*196, 768], k = c*256 + py*16 + px ----------
// one thread = 4 consecutive px (16 B read, 8 B write); a wave covers 4 (c,py) lines of one patch row-block.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B,
                                                       unsigned short* __restrict__ out) {
    const size_t total = (size_t)B * 196 * 192;  // quads
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        const int kq = (int)(q % 192);           // quad index inside the patch row: k = 4*kq
        const size_t pr = q / 192;               // b*196 + p
        const int p = (int)(pr % 196);
        const int b = (int)(pr / 196);
        const int k = kq * 4;
        const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
        const int gy = (p / 14) * 16 + py, gx = (p % 14) * 16 + px;
        const f32x4 v = *(const f32x4*)(img + (((size_t)b * 3 + c) * 224 + gy) * 224 + gx);
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)(out + pr * 768 + k) = o;
    }
}

// The same gather from the image bytes the dataset holds (uint8 [B,3,224,224]; the reference's ToTensor turns them into fp32
// u8 / 255 on the host and copies 4 bytes per value across PCIe, util/dataset.py:185-195, train_epoch.py:26-32): the division is
// the same IEEE fp32 division, so the bf16 patch matrix equals patchify_kernel(image_u8.float() / 255) bit for bit.
// one thread = 8 consecutive px (8 B read, 16 B write).
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ img, int B, unsigned short* __restrict__ out) {
    const size_t total = (size_t)B * 196 * 96;   // octets
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        const int ko = (int)(q % 96);
        const size_t pr = q / 96;                // b*196 + p
        const int p = (int)(pr % 196);
        const int b = (int)(pr / 196);
        const int k = ko * 8;
        const int c = k >> 8, py = (k >> 4) & 15, px = k & 15;
        const int gy = (p / 14) * 16 + py, gx = (p % 14) * 16 + px;
        const uint2 v = *(const uint2*)(img + (((size_t)b * 3 + c) * 224 + gy) * 224 + gx);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = __fdiv_rn((float)((v.x >> (8 * e)) & 255u), 255.0f);
            f[4 + e] = __fdiv_rn((float)((v.y >> (8 * e)) & 255u), 255.0f);
        }
        uint4 o;
        o.x = pack2bf(f[0], f[1]);
        o.y = pack2bf(f[2], f[3]);
        o.z = pack2bf(f[4], f[5]);
        o.w = pack2bf(f[6], f[7]);
        *(uint4*)(out + pr * 768 + k) = o;
    }
}

// ---- LayerNorm -> Linear fold (round 5) ------------------------------------------------------------------------------------------
// stats[m] = (mean, rstd) of row m from the per-slice partial sums the producing GEMM's epilogue wrote (clibd_gemm_epilogue.row_sums:
// [S][M][2] = sum, sum of squares over slice s of the H = 128 S columns): what clibd_layernorm_fwd's `stats` holds, for the consumer GEMM
// and for the LayerNorm BACKWARD, which is unchanged.  The per-slice (sum, sum of squares) are added and differenced in fp64, clamped at 0.
__global__ __launch_bounds__(256) void rowsum_finalize_kernel(const float* __restrict__ sums, int S, int M, float inv_h, float eps, float* __restrict__ stats) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    // Round 6: sums and the difference E[x^2] - mean^2 in fp64, so the merge adds no cancellation of its own (in fp32 the subtraction
    // itself lost (|mean| / sigma)^2 x 6e-8 once more).  What remains is the rounding of the producer's fp32 slice sums: a relative error of
    // about 1.2e-7 (|mean| / sigma)^2 on the variance — negligible for a pre-LN residual stream (|mean| << sigma), 1 % at |mean| / sigma = 300;
    // rows beyond that (single-channel outliers of some pretrained ViTs) want the LayerNorm pass (ln_fold stays opt-in, default off).
    double t = 0.0, q = 0.0;
    for (int j = 0; j < S; ++j) {
        const float2 v = *(const float2*)(sums + ((size_t)j * M + m) * 2);
        t += (double)v.x;
        q += (double)v.y;
    }
    const double mean_d = t * (double)inv_h;
    const float mean = (float)mean_d;
    const float var = fmaxf((float)((q - mean_d * t) * (double)inv_h), 0.f);
    *(float2*)(stats + (size_t)m * 2) = make_float2(mean, rsqrtf(var + eps));
}

// Operand image of a frozen Linear behind a LayerNorm(gamma, beta) for the fold: wg[n,k] = bf16(w[n,k] gamma[k]), s[n] = sum_k float(wg[n,k])
// (the sum of exactly what the MFMA multiplies), bp[n] = b[n] + sum_k w[n,k] beta[k] (fp32).  One wave per output row, once per weight version.
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ b, int N, int K, unsigned short* __restrict__ wg,
                                                              float* __restrict__ s, float* __restrict__ bp) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float as = 0.f, ab = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 wv = *(const f32x4*)(w + (size_t)n * K + k);
        const f32x4 gv = *(const f32x4*)(gamma + k), bv = *(const f32x4*)(beta + k);
        const unsigned short q0 = f2bf(wv[0] * gv[0]), q1 = f2bf(wv[1] * gv[1]), q2 = f2bf(wv[2] * gv[2]), q3 = f2bf(wv[3] * gv[3]);
        *(uint2*)(wg + (size_t)n * K + k) = make_uint2((unsigned)q0 | ((unsigned)q1 << 16), (unsigned)q2 | ((unsigned)q3 << 16));
        as += (bf2f(q0) + bf2f(q1)) + (bf2f(q2) + bf2f(q3));
        ab += (wv[0] * bv[0] + wv[1] * bv[1]) + (wv[2] * bv[2] + wv[3] * bv[3]);
    }
    as = wave_sum(as);
    ab = wave_sum(ab);
    if (lane == 0) { s[n] = as; bp[n] = (b != nullptr ? b[n] : 0.f) + ab; }
}

// tok[b,0,:] = cls + pos[0,:];  tok[b,1+p,:] = proj[b*P+p,:] + pos[1+p,:]   (timm VisionTransformer._pos_embed)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ proj, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, int B, int S, int H,
                                                           float* __restrict__ tok) {
    const int hq = H / 4;
    const size_t total = (size_t)B * S * hq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % hq) * 4;
        const size_t row = i / hq;
        const int s = (int)(row % S);
        const size_t b = row / S;
        const f32x4 p = *(const f32x4*)(pos + (size_t)s * H + c);
        const f32x4 v = (s == 0) ? *(const f32x4*)(cls + c) : *(const f32x4*)(proj + (b * (S - 1) + (s - 1)) * H + c);
        *(f32x4*)(tok + row * H + c) = (f32x4){v[0] + p[0], v[1] + p[1], v[2] + p[2], v[3] + p[3]};
    }
}

// dx = dy * gelu'(pre)   (bf16 in/out; the dense -> GELU -> LayerNorm transform of the BERT MLM head)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ pre,
                                                       size_t n4, unsigned short* __restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 d = *(const uint2*)(dy + 4 * i);
        const uint2 x = *(const uint2*)(pre + 4 * i);
        uint2 o;
        o.x = pack2bf(bf2f((unsigned short)(d.x & 0xffff)) * gelu_grad_f(bf2f((unsigned short)(x.x & 0xffff))),
                      bf2f((unsigned short)(d.x >> 16)) * gelu_grad_f(bf2f((unsigned short)(x.x >> 16))));
        o.y = pack2bf(bf2f((unsigned short)(d.y & 0xffff)) * gelu_grad_f(bf2f((unsigned short)(x.y & 0xffff))),
                      bf2f((unsigned short)(d.y >> 16)) * gelu_grad_f(bf2f((unsigned short)(x.y >> 16))));
        *(uint2*)(dx + 4 * i) = o;
    }
}

// out[b*S+s, :] = word[id] + pos[s] + type[tt]
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tt,
                                                         int B, int S, int H, int vocab,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type, float* __restrict__ out) {
    const int hq = H / 4;
    const size_t total = (size_t)B * S * hq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % hq) * 4;
        const size_t row = i / hq;
        const int s = (int)(row % S);
        long long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // host validates; clamp keeps a bad id from faulting
        const long long t = tt ? (tt[row] != 0 ? 1 : 0) : 0;
        const f32x4 w = *(const f32x4*)(word + (size_t)id * H + c);
        const f32x4 p = *(const f32x4*)(pos + (size_t)s * H + c);
        const f32x4 y = *(const f32x4*)(type + (size_t)t * H + c);
        *(f32x4*)(out + row * H + c) = (f32x4){w[0] + p[0] + y[0], w[1] + p[1] + y[1], w[2] + p[2] + y[2], w[3] + p[3] + y[3]};
    }
}

// ---- K7 tail: out[b,:] = mean_s softmax(logits[b,s,:])  (one block per b, one wave per token, LDS reduce) -------
template <int NCH>  // NCH = C/256 rounded up; lane owns chunks of 4 columns: c = 4*(lane+64j)
__global__ __launch_bounds__(256) void softmax_mean_fwd_kernel(const unsigned short* __restrict__ logits, int S, int C,
                                                               float* __restrict__ out) {
    __shared__ float part[4][1024];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int s = wave; s < S; s += 4) {
        const unsigned short* row = logits + ((size_t)b * S + s) * C;
        f32x4 v[NCH];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = 4 * (lane + 64 * j);
            if (c < C) {
                const uint2 pk = *(const uint2*)(row + c);
                v[j][0] = bf2f((unsigned short)(pk.x & 0xffff)); v[j][1] = bf2f((unsigned short)(pk.x >> 16));
                v[j][2] = bf2f((unsigned short)(pk.y & 0xffff)); v[j][3] = bf2f((unsigned short)(pk.y >> 16));
                mx = fmaxf(mx, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
            } else {
                v[j] = (f32x4){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] = __expf(v[j][e] - mx);
                sum += v[j][e];
            }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] += v[j][e] * inv;
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < C) *(f32x4*)(&part[wave][c]) = acc[j];
    }
    __syncthreads();
    const float invS = 1.0f / (float)S;
    for (int c = threadIdx.x; c < C; c += 256)
        out[(size_t)b * C + c] = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) * invS;
}

// dlogits[b,s,c] = p[s,c] * (g[c] - sum_c' p[s,c'] g[c']),  g = dout[b,:] / S
template <int NCH>
__global__ __launch_bounds__(256) void softmax_mean_bwd_kernel(const unsigned short* __restrict__ logits,
                                                               const float* __restrict__ dout, int B, int S, int C,
                                                               unsigned short* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const size_t rows = (size_t)B * S;
    const float invS = 1.0f / (float)S;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4) {
        const int b = (int)(r / S);
        const unsigned short* row = logits + r * C;
        f32x4 v[NCH], g[NCH];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = 4 * (lane + 64 * j);
            if (c < C) {
                const uint2 pk = *(const uint2*)(row + c);
                v[j][0] = bf2f((unsigned short)(pk.x & 0xffff)); v[j][1] = bf2f((unsigned short)(pk.x >> 16));
                v[j][2] = bf2f((unsigned short)(pk.y & 0xffff)); v[j][3] = bf2f((unsigned short)(pk.y >> 16));
                g[j] = *(const f32x4*)(dout + (size_t)b * C + c);
                mx = fmaxf(mx, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
            } else {
                v[j] = (f32x4){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
                g[j] = (f32x4){0, 0, 0, 0};
            }
        }
        mx = wave_max(mx);
        float sum = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] = __expf(v[j][e] - mx);
                sum += v[j][e];
                dot += v[j][e] * g[j][e];
            }
        sum = wave_sum(sum);
        dot = wave_sum(dot);
        const float inv = 1.0f / sum;
        const float pd = dot * inv;  // sum_c p_c g_c
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = 4 * (lane + 64 * j);
            if (c < C) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[j][e] * inv * (g[j][e] - pd) * invS;
                uint2 pk;
                pk.x = pack2bf(o[0], o[1]);
                pk.y = pack2bf(o[2], o[3]);
                *(uint2*)(dlogits + r * C + c) = pk;
            }
        }
    }
}

// out[b,:] = bf16(mean_s x[b,s,:])
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float* __restrict__ x, int S, int H,
                                                             unsigned short* __restrict__ out_bf16) {
    const int b = blockIdx.x;
    const float invS = 1.0f / (float)S;
    for (int h = threadIdx.x; h < H; h += 256) {
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += x[((size_t)b * S + t) * H + h];
        out_bf16[(size_t)b * H + h] = f2bf(s * invS);
    }
}
// dx[b,s,:] = dout[b,:] / S
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dout, int B, int S, int H,
                                                             float* __restrict__ dx) {
    const size_t total = (size_t)B * S * H;
    const float invS = 1.0f / (float)S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const size_t b = i / ((size_t)S * H);
        dx[i] = dout[b * H + h] * invS;
    }
}

// column sums of bf16 [M,N] into fp32 out[N] (atomic accumulate): block = 256 rows chunk x 64 columns
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short* __restrict__ x, int ld, int M, int N,
                                                          float* __restrict__ out) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * 256;
    float s = 0.f;
    if (c < N) {
        const int r1 = min(r0 + 256, M);
        for (int r = r0 + wave; r < r1; r += 4) s += bf2f(x[(size_t)r * ld + c]);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) atomicAdd(out + c, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int B, int S, int H,
                                                          float* __restrict__ out) {
    const int total = B * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, h = i % H;
        out[i] = x[(size_t)b * S * H + h];
    }
}
// dx[b,0,:] = dcls[b,:], all other rows zero
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dcls, int B, int S, int H,
                                                           unsigned short* __restrict__ dx_bf16,
                                                           float* __restrict__ dx_f32) {
    const size_t total = (size_t)B * S * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const size_t row = i / H;
        const int s = (int)(row % S);
        const size_t b = row / S;
        const float v = (s == 0) ? dcls[b * H + h] : 0.f;
        if (dx_f32) dx_f32[i] = v;
        if (dx_bf16) dx_bf16[i] = f2bf(v);
    }
}

// ---- f1: 5-mer tokenisation of padded nucleotide strings (model/dna_encoder.py:53-63, util/util.py:77-98) ----------------
// seq uint8 [B,L] ('N'-padded / truncated by the host), out int64 [B, 1 + L/k]: out[b,0] = 0 (<MASK> id leads every
// sequence in the reference pipeline), out[b,1+t] = 3 + base-4 value of chars [k*t, k*t+k) with A0 C1 G2 T3, or 2 (<UNK>)
// when any char is not ACGT.
__global__ __launch_bounds__(256) void kmer_tokenize_kernel(const unsigned char* __restrict__ seq, int B, int L, int k,
                                                            long long* __restrict__ out) {
    const int T = L / k;
    const size_t total = (size_t)B * (T + 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % (T + 1));
        const size_t b = i / (T + 1);
        long long id = 0;
        if (t > 0) {
            const unsigned char* c = seq + b * L + (size_t)(t - 1) * k;
            int v = 0;
            bool ok = true;
            for (int j = 0; j < k; ++j) {
                const unsigned char ch = c[j];
                const int d = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
                ok = ok && d >= 0;
                v = v * 4 + (d < 0 ? 0 : d);
            }
            id = ok ? 3 + v : 2;
        }
        out[i] = id;
    }
}

// ---- K8: y = x / max(||x||, eps) -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int N, int D,
                                                         float* __restrict__ y, float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < N; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
        s = wave_sum(s);
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = xr[c] * inv;
        if (lane == 0 && inv_norm) inv_norm[row] = inv;
    }
}
// dx = inv * (dy - y * <dy, y>)   (for ||x|| > eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv_norm, int N, int D,
                                                         float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < N; row += gridDim.x * 4) {
        const float* g = dy + (size_t)row * D;
        const float* yr = y + (size_t)row * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += g[c] * yr[c];
        s = wave_sum(s);
        const float inv = inv_norm[row];
        for (int c = lane; c < D; c += 64) dx[(size_t)row * D + c] = inv * (g[c] - yr[c] * s);
    }
}

// ---- fused AdamW on a flat bucket (torch.optim.AdamW: decoupled decay, bias-corrected) ---------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i];
        pi *= (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}


}  // namespace clibd

using namespace clibd;

// ---- fp8-forward mode: per-output-channel quantisation of a frozen weight matrix ------------------------------------------
namespace clibd {
// one wave per row: s = 448 / max|w[n,:]|, w8[n,k] = e4m3(w[n,k] * s), col_scale[n] = 1 / (s * act_scale)
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const float* __restrict__ w, int N, int K, float act_scale,
                                                               unsigned char* __restrict__ w8, float* __restrict__ col_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* wr = w + (size_t)row * K;
    float amax = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 v = *(const f32x4*)(wr + k);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? 448.f / amax : 1.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 v = *(const f32x4*)(wr + k);
        *(unsigned*)(w8 + (size_t)row * K + k) = pack4fp8(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    }
    if (lane == 0) col_scale[row] = 1.f / (s * act_scale);
}
// The image of a bf16 matrix (the transposed weight shadows the dgrad GEMMs already read) under a power-of-two row scale, plus the largest row l1 norm of the
// DE-QUANTISED image: l1max = max_n sum_k |e4m3 value| / s_n (atomic max on the bits of a non-negative float; caller zeroes it).
// It bounds the 8-bit dgrad's output, |sum_k a_k w_nk| <= max|a| * l1max, from which the fc2 dgrad's fixed output scale is derived.
__global__ __launch_bounds__(256) void quantize_rows_fp8_bf16_kernel(const unsigned short* __restrict__ w, int N, int K, float act_scale,
                                                                    unsigned char* __restrict__ w8, float* __restrict__ col_scale,
                                                                    float* __restrict__ l1max) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const unsigned short* wr = w + (size_t)row * K;
    float amax = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const uint2 pk = *(const uint2*)(wr + k);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(bf2f((unsigned short)(pk.x & 0xffff))), fabsf(bf2f((unsigned short)(pk.x >> 16))))),
                     fmaxf(fabsf(bf2f((unsigned short)(pk.y & 0xffff))), fabsf(bf2f((unsigned short)(pk.y >> 16)))));
    }
    amax = wave_max(amax);
    // a power-of-two scale, s = 2^(7 - floor(log2 amax)) (the row-scale rule of clibd_layernorm_bwd_fp8: scaled maximum in [128, 256)):
    // exact in every arithmetic, so the image does not depend on how a division rounds — bf16 inputs sit ON e4m3 rounding ties
    // under the 448 / amax scale of the fp32 quantiser (w / amax is a ratio of 8-bit integers)
    float s = 1.f;
    {
        unsigned eb = (__float_as_uint(amax) >> 23) & 0xffu;
        if (amax > 0.f && eb != 255u) s = __uint_as_float((261u - (eb < 8u ? 8u : eb)) << 23);
    }
    float l1 = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const uint2 pk = *(const uint2*)(wr + k);
        const unsigned q = pack4fp8(bf2f((unsigned short)(pk.x & 0xffff)) * s, bf2f((unsigned short)(pk.x >> 16)) * s,
                                    bf2f((unsigned short)(pk.y & 0xffff)) * s, bf2f((unsigned short)(pk.y >> 16)) * s);
        *(unsigned*)(w8 + (size_t)row * K + k) = q;
        l1 += (fabsf(__builtin_amdgcn_cvt_f32_fp8((int)q, 0)) + fabsf(__builtin_amdgcn_cvt_f32_fp8((int)q, 1))) +
              (fabsf(__builtin_amdgcn_cvt_f32_fp8((int)q, 2)) + fabsf(__builtin_amdgcn_cvt_f32_fp8((int)q, 3)));
    }
    l1 = wave_sum(l1) / s;
    if (lane == 0) {
        col_scale[row] = 1.f / (s * act_scale);
        if (l1max != nullptr) atomicMax((unsigned*)l1max, __float_as_uint(l1));
    }
}
}  // namespace clibd

extern "C" int clibd_quantize_rows_fp8_bf16(const void* w_bf16, int N, int K, float act_scale, void* w_fp8, float* col_scale, float* l1max,
                                            void* stream) {
    if (!w_bf16 || !w_fp8 || !col_scale) return set_error(CLIBD_EINVAL, "quantize_rows_fp8_bf16: null pointer");
    if (N <= 0 || K <= 0 || (K & 3) || !(act_scale > 0.f) || ((uintptr_t)w_bf16 & 7) || ((uintptr_t)w_fp8 & 3) || ((uintptr_t)l1max & 3))
        return set_error(CLIBD_EINVAL, "quantize_rows_fp8_bf16: bad args (K % 4, act_scale > 0, alignment)");
    hipLaunchKernelGGL(clibd::quantize_rows_fp8_bf16_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_bf16, N, K, act_scale, (unsigned char*)w_fp8, col_scale, l1max);
    return check_launch("quantize_rows_fp8_bf16");
}

extern "C" int clibd_quantize_rows_fp8(const float* w, int N, int K, float act_scale, void* w_fp8, float* col_scale, void* stream) {
    if (!w || !w_fp8 || !col_scale) return set_error(CLIBD_EINVAL, "quantize_rows_fp8: null pointer");
    if (N <= 0 || K <= 0 || (K & 3) || !(act_scale > 0.f) || !aligned16(w) || ((uintptr_t)w_fp8 & 3)) return set_error(CLIBD_EINVAL, "quantize_rows_fp8: bad args (K % 4, act_scale > 0, alignment)");
    hipLaunchKernelGGL(clibd::quantize_rows_fp8_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, N, K, act_scale,
                       (unsigned char*)w_fp8, col_scale);
    return check_launch("quantize_rows_fp8");
}

extern "C" int clibd_patchify(const float* image, int B, void* patches_bf16, void* stream) {
    if (!image || !patches_bf16 || B <= 0) return set_error(CLIBD_EINVAL, "patchify: bad args");
    if (!aligned16(image) || !aligned16(patches_bf16)) return set_error(CLIBD_EINVAL, "patchify: alignment");
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((size_t)B * 196 * 192)), dim3(256), 0, (hipStream_t)stream, image, B,
                       (unsigned short*)patches_bf16);
    return check_launch("patchify");
}

extern "C" int clibd_patchify_u8(const unsigned char* image, int B, void* patches_bf16, void* stream) {
    if (!image || !patches_bf16 || B <= 0) return set_error(CLIBD_EINVAL, "patchify_u8: bad args");
    if (((uintptr_t)image & 7) || !aligned16(patches_bf16)) return set_error(CLIBD_EINVAL, "patchify_u8: alignment");
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(grid_for((size_t)B * 196 * 96)), dim3(256), 0, (hipStream_t)stream, image, B,
                       (unsigned short*)patches_bf16);
    return check_launch("patchify_u8");
}

extern "C" int clibd_rowsum_finalize(const float* row_sums, int slices, int M, int H, float eps, float* stats, void* stream) {
    if (!row_sums || !stats || slices <= 0 || M <= 0 || H != 128 * slices) return set_error(CLIBD_EINVAL, "rowsum_finalize: need H == 128 * slices");
    if (((uintptr_t)row_sums & 7) || ((uintptr_t)stats & 7)) return set_error(CLIBD_EINVAL, "rowsum_finalize: alignment");
    hipLaunchKernelGGL(rowsum_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row_sums, slices, M, 1.0f / (float)H, eps, stats);
    return check_launch("rowsum_finalize");
}

extern "C" int clibd_ln_fold_weights(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K, void* wg_bf16,
                                     float* col_sum_w, float* bias_folded, void* stream) {
    if (!w || !gamma || !beta || !wg_bf16 || !col_sum_w || !bias_folded) return set_error(CLIBD_EINVAL, "ln_fold_weights: null pointer");
    if (N <= 0 || K <= 0 || K % 4 != 0) return set_error(CLIBD_EINVAL, "ln_fold_weights: K must be a positive multiple of 4");
    if (!aligned16(w) || !aligned16(gamma) || !aligned16(beta) || ((uintptr_t)wg_bf16 & 7)) return set_error(CLIBD_EINVAL, "ln_fold_weights: alignment");
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, bias, N, K,
                       (unsigned short*)wg_bf16, col_sum_w, bias_folded);
    return check_launch("ln_fold_weights");
}

extern "C" int clibd_vit_assemble_tokens(const float* proj, const float* cls, const float* pos, int B, int S, int H, float* tok,
                                         void* stream) {
    if (!proj || !cls || !pos || !tok || B <= 0 || S <= 1 || H <= 0 || H % 4 != 0) return set_error(CLIBD_EINVAL, "vit_assemble_tokens: bad args");
    if (!aligned16(proj) || !aligned16(cls) || !aligned16(pos) || !aligned16(tok)) return set_error(CLIBD_EINVAL, "vit_assemble_tokens: alignment");
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for((size_t)B * S * (H / 4))), dim3(256), 0, (hipStream_t)stream, proj, cls, pos, B, S, H, tok);
    return check_launch("vit_assemble_tokens");
}

extern "C" int clibd_gelu_bwd_bf16(const void* dy, const void* pre, size_t n, void* dx, void* stream) {
    if (!dy || !pre || !dx) return set_error(CLIBD_EINVAL, "gelu_bwd: null pointer");
    if (n % 4 != 0) return set_error(CLIBD_EINVAL, "gelu_bwd: n must be a multiple of 4");
    if (n == 0) return CLIBD_OK;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                       (const unsigned short*)pre, n / 4, (unsigned short*)dx);
    return check_launch("gelu_bwd");
}

extern "C" int clibd_bert_embed(const int64_t* ids, const int64_t* token_type, int B, int S, int H, int vocab,
                                const float* word, const float* pos, const float* type, float* out, void* stream) {
    if (!ids || !word || !pos || !type || !out) return set_error(CLIBD_EINVAL, "bert_embed: null pointer");
    if (B <= 0 || S <= 0 || H <= 0 || H % 4 != 0 || vocab <= 0) return set_error(CLIBD_EINVAL, "bert_embed: bad shape");
    if (!aligned16(word) || !aligned16(pos) || !aligned16(type) || !aligned16(out)) return set_error(CLIBD_EINVAL, "bert_embed: alignment");
    hipLaunchKernelGGL(bert_embed_kernel, dim3(grid_for((size_t)B * S * (H / 4))), dim3(256), 0, (hipStream_t)stream, ids,
                       token_type, B, S, H, vocab, word, pos, type, out);
    return check_launch("bert_embed");
}

extern "C" int clibd_softmax_mean_fwd(const void* logits, int B, int S, int C, float* out, void* stream) {
    if (!logits || !out || B <= 0 || S <= 0) return set_error(CLIBD_EINVAL, "softmax_mean_fwd: bad args");
    if (C <= 0 || C % 64 != 0 || C > 1024) return set_error(CLIBD_EINVAL, "softmax_mean_fwd: C must be a multiple of 64, <= 1024");
    const int nch = (C + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(N) hipLaunchKernelGGL(softmax_mean_fwd_kernel<N>, dim3(B), dim3(256), 0, st, (const unsigned short*)logits, S, C, out)
    switch (nch) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); break; }
#undef LAUNCH
    return check_launch("softmax_mean_fwd");
}

extern "C" int clibd_softmax_mean_bwd(const void* logits, const float* dout, int B, int S, int C, void* dlogits,
                                      void* stream) {
    if (!logits || !dout || !dlogits || B <= 0 || S <= 0) return set_error(CLIBD_EINVAL, "softmax_mean_bwd: bad args");
    if (C <= 0 || C % 64 != 0 || C > 1024) return set_error(CLIBD_EINVAL, "softmax_mean_bwd: C must be a multiple of 64, <= 1024");
    const int nch = (C + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(grid_for((size_t)B * S, 4));
#define LAUNCH(N) hipLaunchKernelGGL(softmax_mean_bwd_kernel<N>, grid, dim3(256), 0, st, (const unsigned short*)logits, dout, B, S, C, (unsigned short*)dlogits)
    switch (nch) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); break; }
#undef LAUNCH
    return check_launch("softmax_mean_bwd");
}

extern "C" int clibd_token_mean_fwd(const float* x, int B, int S, int H, void* out_bf16, void* stream) {
    if (!x || !out_bf16 || B <= 0 || S <= 0 || H <= 0) return set_error(CLIBD_EINVAL, "token_mean_fwd: bad args");
    hipLaunchKernelGGL(token_mean_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, S, H, (unsigned short*)out_bf16);
    return check_launch("token_mean_fwd");
}
extern "C" int clibd_token_mean_bwd(const float* dout, int B, int S, int H, float* dx, void* stream) {
    if (!dout || !dx || B <= 0 || S <= 0 || H <= 0) return set_error(CLIBD_EINVAL, "token_mean_bwd: bad args");
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid_for((size_t)B * S * H)), dim3(256), 0, (hipStream_t)stream, dout, B, S, H, dx);
    return check_launch("token_mean_bwd");
}

extern "C" int clibd_colsum_bf16(const void* x, int ld, int M, int N, float* out, void* stream) {
    if (!x || !out || M <= 0 || N <= 0 || ld < N) return set_error(CLIBD_EINVAL, "colsum: bad args");
    dim3 grid((N + 63) / 64, (M + 255) / 256);
    hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ld, M, N, out);
    return check_launch("colsum_bf16");
}

extern "C" int clibd_gather_rows(const float* x, int B, int S, int H, float* out, void* stream) {
    if (!x || !out || B <= 0 || S <= 0 || H <= 0) return set_error(CLIBD_EINVAL, "gather_rows: bad args");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)B * H)), dim3(256), 0, (hipStream_t)stream, x, B, S, H, out);
    return check_launch("gather_rows");
}
extern "C" int clibd_scatter_rows_bf16(const float* dcls, int B, int S, int H, void* dx_bf16, float* dx_f32, void* stream) {
    if (!dcls || (!dx_bf16 && !dx_f32) || B <= 0 || S <= 0 || H <= 0) return set_error(CLIBD_EINVAL, "scatter_rows: bad args");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((size_t)B * S * H)), dim3(256), 0, (hipStream_t)stream, dcls, B, S, H,
                       (unsigned short*)dx_bf16, dx_f32);
    return check_launch("scatter_rows");
}

extern "C" int clibd_kmer_tokenize(const void* seq_u8, int B, int L, int k, int64_t* out, void* stream) {
    if (!seq_u8 || !out || B <= 0 || L <= 0 || k <= 0 || k > 12 || L % k != 0) return set_error(CLIBD_EINVAL, "kmer_tokenize: bad args (L % k == 0, k <= 12)");
    hipLaunchKernelGGL(kmer_tokenize_kernel, dim3(grid_for((size_t)B * (L / k + 1))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)seq_u8, B, L, k, (long long*)out);
    return check_launch("kmer_tokenize");
}

extern "C" int clibd_l2norm_fwd(const float* x, int N, int D, float* y, float* inv_norm, void* stream) {
    if (!x || !y || N <= 0 || D <= 0) return set_error(CLIBD_EINVAL, "l2norm_fwd: bad args");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid_for((size_t)N, 4)), dim3(256), 0, (hipStream_t)stream, x, N, D, y, inv_norm);
    return check_launch("l2norm_fwd");
}
extern "C" int clibd_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int N, int D, float* dx, void* stream) {
    if (!dy || !y || !inv_norm || !dx || N <= 0 || D <= 0) return set_error(CLIBD_EINVAL, "l2norm_bwd: bad args");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid_for((size_t)N, 4)), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, N, D, dx);
    return check_launch("l2norm_bwd");
}

extern "C" int clibd_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v) return set_error(CLIBD_EINVAL, "adamw: null pointer");
    if (step < 1) return set_error(CLIBD_EINVAL, "adamw: step must be >= 1");
    if (n == 0) return CLIBD_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrtf(bc2), grad_scale);
    return check_launch("adamw");
}
