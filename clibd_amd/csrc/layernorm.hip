// LayerNorm forward / backward for gfx950: one wave64 per row, fp32 statistics, 16-byte vector I/O.
// HBM-bound: forward reads 4H bytes/row and writes 2H (bf16) [+4H fp32]; the optional LoRA down-projection
// t = bf16(y) · A_cat^T (8 outputs per row) rides along in registers so the adapter never re-reads y.
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

constexpr int LN_MAX_CHUNKS = 4;  // H <= 1024: each lane owns up to 4 float4 chunks (chunk j = cols 4*(lane + 64 j))

// Reduce 8 per-lane partials across the wave: a butterfly that halves the live values at each of the first three steps.
// v_permlane32_swap / v_permlane16_swap exchange the lane halves / the 16-lane rows (one swap replaces two selects and a
// ds_bpermute), DPP row_ror:8 the 8-lane halves of a row, then row_half_mirror and two quad permutes sum the aligned group of
// 8 lanes.  No LDS-crossbar instruction (common.h explains why).  On return every lane holds the total of value index
// ((lane>>3)&7).
__device__ __forceinline__ float wave_reduce8(float v[8], int lane) {
    float w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // lanes 0-31: v[i] summed over both halves; lanes 32-63: v[i+4]
        float a = v[i], b = v[i + 4];
        permlane32_swap(a, b);
        w4[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // even 16-lane rows: w4[i] summed over the row pair; odd rows: w4[i+2]
        float a = w4[i], b = w4[i + 2];
        permlane16_swap(a, b);
        w2[i] = a + b;
    }
    const bool b3 = lane & 8;
    const float keep = b3 ? w2[1] : w2[0];
    const float send = b3 ? w2[0] : w2[1];
    float w1 = keep + dpp_mov<DPP_ROW_ROR8>(send);   // lane ^ 8 inside a 16-lane row
    w1 += dpp_mov<DPP_ROW_HALF_MIRROR>(w1);
    w1 += dpp_mov<DPP_QUAD_XOR1>(w1);
    w1 += dpp_mov<DPP_QUAD_XOR2>(w1);
    return w1;  // index = 4*b5 + 2*b4 + b3
}

// LORA: the adapter matrix A_cat [8,H] sits in LDS as fp32 (conflict-free 16-byte reads), so the kernel stays at
// high occupancy (HBM-bound: what matters is bytes in flight per CU, i.e. resident waves).
// ROWS adjacent rows per wave and iteration: twice the bytes in flight per wave, and with LORA every adapter vector read from
// LDS serves ROWS rows (the down-projection is LDS-read bound: 24 KB of adapter reads per row at ROWS = 1).
template <int NCH, bool LORA, int ROWS>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int M, int H,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            unsigned short* __restrict__ y_bf16,
                                                            float* __restrict__ y_f32, float* __restrict__ stats,
                                                            const unsigned short* __restrict__ lora_a,
                                                            unsigned short* __restrict__ t_bf16, unsigned drop_seed,
                                                            int drop_thr16, float drop_scale,
                                                            unsigned char* __restrict__ y_fp8, float fp8_scale) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[];  // [8][H] when LORA
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    if (LORA) {
        for (int i = threadIdx.x; i < 8 * H; i += 256) a_lds[i] = bf2f(lora_a[i]);
        __syncthreads();
    }
    f32x4 g[NCH], b[NCH];
    bool act[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = 4 * (lane + 64 * j);
        act[j] = c < H;
        g[j] = act[j] ? *(const f32x4*)(gamma + c) : (f32x4){0, 0, 0, 0};
        b[j] = act[j] ? *(const f32x4*)(beta + c) : (f32x4){0, 0, 0, 0};
    }
    const float invH = 1.0f / (float)H;
    for (int row0 = wave_in_grid * ROWS; row0 < M; row0 += nwaves * ROWS) {
        f32x4 v[ROWS][NCH];
        float mean[ROWS], rstd[ROWS];
        bool live[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            live[rr] = row0 + rr < M;
            const float* xr = x + (size_t)min(row0 + rr, M - 1) * H;
#pragma unroll
            for (int j = 0; j < NCH; ++j) v[rr][j] = act[j] ? *(const f32x4*)(xr + 4 * (lane + 64 * j)) : (f32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) s += (v[rr][j][0] + v[rr][j][1]) + (v[rr][j][2] + v[rr][j][3]);
            mean[rr] = wave_sum(s) * invH;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (act[j]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[rr][j][e] - mean[rr];
                        q += d * d;
                    }
                }
            }
            const float var = wave_sum(q) * invH;
            rstd[rr] = rsqrtf(var + eps);
            if (stats != nullptr && lane == 0 && live[rr]) *(float2*)(stats + 2 * (size_t)(row0 + rr)) = make_float2(mean[rr], rstd[rr]);
        }
        float tp[ROWS][8];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
            for (int r = 0; r < 8; ++r) tp[rr][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!act[j]) continue;
            const int c = 4 * (lane + 64 * j);
            float yb[ROWS][4];   // the bf16-rounded outputs, as the GEMM will read them (LORA)
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int row = row0 + rr;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[rr][j][e] - mean[rr]) * rstd[rr] * g[j][e] + b[j][e];
                if (drop_thr16 > 0) {  // y = dropout(LN(x)): HF BertEmbeddings
                    const unsigned base = (unsigned)row * (unsigned)H + (unsigned)c;
                    float f0, f1, f2, f3;
                    drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                    drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                    y[0] *= f0; y[1] *= f1; y[2] *= f2; y[3] *= f3;
                }
                uint2 pk;
                pk.x = pack2bf(y[0], y[1]);
                pk.y = pack2bf(y[2], y[3]);
                if (live[rr]) {
                    if (y_f32 != nullptr) *(f32x4*)(y_f32 + (size_t)row * H + c) = y;
                    if (y_bf16 != nullptr) *(uint2*)(y_bf16 + (size_t)row * H + c) = pk;
                    if (y_fp8 != nullptr)  // fp8-forward mode: the next GEMM's operand, quantised from the fp32 value
                        *(unsigned*)(y_fp8 + (size_t)row * H + c) = pack4fp8(y[0] * fp8_scale, y[1] * fp8_scale, y[2] * fp8_scale, y[3] * fp8_scale);
                }
                yb[rr][0] = bf2f((unsigned short)(pk.x & 0xffff)); yb[rr][1] = bf2f((unsigned short)(pk.x >> 16));
                yb[rr][2] = bf2f((unsigned short)(pk.y & 0xffff)); yb[rr][3] = bf2f((unsigned short)(pk.y >> 16));
            }
            if (LORA) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const f32x4 a = *(const f32x4*)(a_lds + r * H + c);
#pragma unroll
                    for (int rr = 0; rr < ROWS; ++rr)
                        tp[rr][r] += (yb[rr][0] * a[0] + yb[rr][1] * a[1]) + (yb[rr][2] * a[2] + yb[rr][3] * a[3]);
                }
            }
        }
        if (LORA) {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const float tv = wave_reduce8(tp[rr], lane);
                if ((lane & 7) == 0 && live[rr]) t_bf16[(size_t)(row0 + rr) * 8 + (lane >> 3)] = f2bf(tv);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma ;  optional + dres
// The residual gradient may travel as bf16 instead of fp32 (clibd_layernorm_bwd_res16): dres_b16 is then the incoming stream
// and dx_res_b16 an un-dropped bf16 copy of the result (the gradient of the residual sum; dx_bf16 carries the dropout mask of
// the dense branch when one is active) — 10 instead of 16 bytes per element for a pre-LN block.
// PG (full fine-tune mode): also dgamma[c] += sum_rows dy * xhat, dbeta[c] += sum_rows dy — the kernel has dy and xhat in
// registers anyway; every wave keeps its columns' partial sums over the rows it walks, the block combines its four waves in
// LDS and issues ONE float atomic per column and parameter (the grid is capped so that these stay a few microseconds).
// F8 (8-bit dgrad, round 5): the copy the dense branch's dgrad consumes (dropout mask applied, as dx_bf16) also leaves as e4m3 bytes with
// ONE power-of-two scale per row, s = 2^(7 - floor(log2 max|row|)) (the scaled row maximum lies in [128, 256), e4m3's top binade but one;
// an all-zero row takes s = 1), dx_fp8[row, c] = e4m3(value * s), row_dequant[row] = 1 / s — the A operand of clibd_gemm_fp8_dgrad_nt.
// A wave owns whole rows, so the row maximum is a DPP reduction of values it already holds: no second pass, no history of maxima.
template <int NCH, bool PG, int ROWS, bool F8>
__device__ __forceinline__ void layernorm_bwd_body(const unsigned short* __restrict__ dy_bf16,
                                                   const float* __restrict__ dy_f32,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ stats,
                                                   const float* __restrict__ gamma, int M, int H,
                                                   const float* __restrict__ dres,
                                                   float* __restrict__ dx_f32,
                                                   unsigned short* __restrict__ dx_bf16, unsigned drop_seed,
                                                   int drop_thr16, float drop_scale, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, const unsigned short* __restrict__ dres_b16,
                                                   unsigned short* __restrict__ dx_res_b16, unsigned char* __restrict__ dx_fp8,
                                                   float* __restrict__ row_dequant) {
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    f32x4 g[NCH];
    f32x4 pgam[PG ? NCH : 1], pbet[PG ? NCH : 1];
    bool act[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = 4 * (lane + 64 * j);
        act[j] = c < H;
        g[j] = act[j] ? *(const f32x4*)(gamma + c) : (f32x4){0, 0, 0, 0};
        if (PG) { pgam[j] = (f32x4){0, 0, 0, 0}; pbet[j] = (f32x4){0, 0, 0, 0}; }
    }
    const float invH = 1.0f / (float)H;
    // ROWS adjacent rows per wave and iteration, and EVERY input of a row — the incoming residual gradient included — requested
    // before the first reduction: one memory round trip per iteration instead of two per row.
    for (int row0 = wave_in_grid * ROWS; row0 < M; row0 += nwaves * ROWS) {
        // RS16 (the two-row e4m3 form: launched only with a bf16 incoming stream and no fp32 one): the residual gradient waits for the
        // reductions as the packed bf16 pairs it arrived in — 2 registers per chunk instead of 4, 12 fewer at H = 768, which is what
        // the 128-register bound of four waves per SIMD was short of (round 5: 4 spilled; round 6: none)
        constexpr bool RS16 = F8 && ROWS == 2;
        f32x4 gy[ROWS][NCH], xh[ROWS][NCH], rs[RS16 ? 1 : ROWS][RS16 ? 1 : NCH];
        uint2 rsp[RS16 ? ROWS : 1][RS16 ? NCH : 1];
        float rstd[ROWS], m1[ROWS], m2[ROWS];
        bool live[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            live[rr] = row0 + rr < M;
            const size_t row = (size_t)min(row0 + rr, M - 1);
            const float mean = stats[2 * row];
            rstd[rr] = stats[2 * row + 1];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                gy[rr][j] = (f32x4){0, 0, 0, 0};
                xh[rr][j] = (f32x4){0, 0, 0, 0};
                if constexpr (RS16) rsp[rr][j] = make_uint2(0u, 0u);
                else rs[rr][j] = (f32x4){0, 0, 0, 0};
                if (!act[j]) continue;
                const int c = 4 * (lane + 64 * j);
                f32x4 d;
                if (dy_f32 != nullptr) {
                    d = *(const f32x4*)(dy_f32 + row * H + c);
                } else {
                    const uint2 pk = *(const uint2*)(dy_bf16 + row * H + c);
                    d[0] = bf2f((unsigned short)(pk.x & 0xffff)); d[1] = bf2f((unsigned short)(pk.x >> 16));
                    d[2] = bf2f((unsigned short)(pk.y & 0xffff)); d[3] = bf2f((unsigned short)(pk.y >> 16));
                }
                const f32x4 xv = *(const f32x4*)(x + row * H + c);
                if constexpr (RS16) {
                    if (dres_b16 != nullptr) rsp[rr][j] = *(const uint2*)(dres_b16 + row * H + c);
                } else {
                    if (dres != nullptr) rs[rr][j] = *(const f32x4*)(dres + row * H + c);
                    if (dres_b16 != nullptr) {
                        const uint2 pk = *(const uint2*)(dres_b16 + row * H + c);
                        rs[rr][j][0] += bf2f((unsigned short)(pk.x & 0xffff)); rs[rr][j][1] += bf2f((unsigned short)(pk.x >> 16));
                        rs[rr][j][2] += bf2f((unsigned short)(pk.y & 0xffff)); rs[rr][j][3] += bf2f((unsigned short)(pk.y >> 16));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gy[rr][j][e] = d[e] * g[j][e];
                    xh[rr][j][e] = (xv[e] - mean) * rstd[rr];
                    if (PG && live[rr]) { pgam[j][e] += d[e] * xh[rr][j][e]; pbet[j][e] += d[e]; }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += gy[rr][j][e];
                    s2 += gy[rr][j][e] * xh[rr][j][e];
                }
            m1[rr] = wave_sum(s1) * invH;
            m2[rr] = wave_sum(s2) * invH;
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            if (!live[rr]) continue;
            const size_t row = (size_t)(row0 + rr);
            float amax = 0.f;   // F8: the masked values are kept in gy[rr][j] (dead once o is formed) until the row maximum is known
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (!act[j]) continue;
                const int c = 4 * (lane + 64 * j);
                f32x4 o, rv;
                if constexpr (RS16) {
                    const uint2 pk = rsp[rr][j];
                    rv[0] = bf2f((unsigned short)(pk.x & 0xffff)); rv[1] = bf2f((unsigned short)(pk.x >> 16));
                    rv[2] = bf2f((unsigned short)(pk.y & 0xffff)); rv[3] = bf2f((unsigned short)(pk.y >> 16));
                } else {
                    rv = rs[rr][j];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd[rr] * (gy[rr][j][e] - m1[rr] - xh[rr][j][e] * m2[rr]) + rv[e];
                if (dx_f32 != nullptr) *(f32x4*)(dx_f32 + row * H + c) = o;
                if (dx_res_b16 != nullptr) {
                    uint2 pk;
                    pk.x = pack2bf(o[0], o[1]);
                    pk.y = pack2bf(o[2], o[3]);
                    *(uint2*)(dx_res_b16 + row * H + c) = pk;
                }
                if (dx_bf16 != nullptr || F8) {
                    if (drop_thr16 > 0) {  // this copy is d(dense out) = d(sum) * mask / (1-p) of the forward's hidden dropout
                        const unsigned base = (unsigned)row * (unsigned)H + (unsigned)c;
                        float f0, f1, f2, f3;
                        drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                        drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                        o[0] *= f0; o[1] *= f1; o[2] *= f2; o[3] *= f3;
                    }
                    if (dx_bf16 != nullptr) {
                        uint2 pk;
                        pk.x = pack2bf(o[0], o[1]);
                        pk.y = pack2bf(o[2], o[3]);
                        *(uint2*)(dx_bf16 + row * H + c) = pk;
                    }
                    if (F8) {
                        gy[rr][j] = o;
                        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    }
                }
            }
            if (F8) {
                amax = wave_max(amax);
                // biased exponent of the row maximum -> s = 2^(134 - eb), 1 / s = 2^(eb - 134); rows below 2^-119 (eb < 8) are scaled as if
                // their maximum were 2^-119 (they quantise to zeros), a non-finite maximum (eb = 255) takes s = 1
                unsigned eb = (__float_as_uint(amax) >> 23) & 0xffu;
                float sc = 1.f, sinv = 1.f;
                if (amax > 0.f && eb != 255u) {
                    eb = eb < 8u ? 8u : eb;
                    sc = __uint_as_float((261u - eb) << 23);
                    sinv = __uint_as_float((eb - 7u) << 23);
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    if (!act[j]) continue;
                    const int c = 4 * (lane + 64 * j);
                    *(unsigned*)(dx_fp8 + row * H + c) = pack4fp8(gy[rr][j][0] * sc, gy[rr][j][1] * sc, gy[rr][j][2] * sc, gy[rr][j][3] * sc);
                }
                if (lane == 0) row_dequant[row] = sinv;
            }
        }
    }
    if (PG) {
        __shared__ __attribute__((aligned(16))) float pg_lds[2][4][1024];   // [gamma|beta][wave][column]
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!act[j]) continue;
            const int c = 4 * (lane + 64 * j);
            *(f32x4*)(&pg_lds[0][wv][c]) = pgam[j];
            *(f32x4*)(&pg_lds[1][wv][c]) = pbet[j];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256) {
            atomicAdd(dgamma + c, (pg_lds[0][0][c] + pg_lds[0][1][c]) + (pg_lds[0][2][c] + pg_lds[0][3][c]));
            atomicAdd(dbeta + c, (pg_lds[1][0][c] + pg_lds[1][1][c]) + (pg_lds[1][2][c] + pg_lds[1][3][c]));
        }
    }
}

template <int NCH, bool PG, int ROWS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const unsigned short* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                                            const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, int M, int H, const float* __restrict__ dres,
                                                            float* __restrict__ dx_f32, unsigned short* __restrict__ dx_bf16, unsigned drop_seed,
                                                            int drop_thr16, float drop_scale, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const unsigned short* __restrict__ dres_b16,
                                                            unsigned short* __restrict__ dx_res_b16) {
    layernorm_bwd_body<NCH, PG, ROWS, false>(dy_bf16, dy_f32, x, stats, gamma, M, H, dres, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma,
                                             dbeta, dres_b16, dx_res_b16, nullptr, nullptr);
}
// (four waves per SIMD, as the plain kernel reaches at H = 768: without the bound the two-row form takes 131 registers — three waves)
#ifndef CLIBD_LNB8_WAVES
#define CLIBD_LNB8_WAVES 4
#endif
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CLIBD_LNB8_WAVES, 8))) void layernorm_bwd_fp8_kernel(const unsigned short* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                                                const float* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, int M, int H, const float* __restrict__ dres,
                                                                float* __restrict__ dx_f32, unsigned short* __restrict__ dx_bf16, unsigned drop_seed,
                                                                int drop_thr16, float drop_scale, const unsigned short* __restrict__ dres_b16,
                                                                unsigned short* __restrict__ dx_res_b16, unsigned char* __restrict__ dx_fp8,
                                                                float* __restrict__ row_dequant) {
    layernorm_bwd_body<NCH, false, ROWS, true>(dy_bf16, dy_f32, x, stats, gamma, M, H, dres, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, nullptr,
                                               nullptr, dres_b16, dx_res_b16, dx_fp8, row_dequant);
}

// 8-bit dgrad with TRAINABLE base weights (round 6, full fine-tune): the e4m3 rows for the dgrad, the bf16 copy for the weight gradient and
// the LayerNorm parameter gradients, one row per wave and iteration (the PG form's grid and float atomics)
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_fp8_pg_kernel(const unsigned short* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                                                   const float* __restrict__ x, const float* __restrict__ stats,
                                                                   const float* __restrict__ gamma, int M, int H, const float* __restrict__ dres,
                                                                   float* __restrict__ dx_f32, unsigned short* __restrict__ dx_bf16, unsigned drop_seed,
                                                                   int drop_thr16, float drop_scale, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   const unsigned short* __restrict__ dres_b16, unsigned short* __restrict__ dx_res_b16,
                                                                   unsigned char* __restrict__ dx_fp8, float* __restrict__ row_dequant) {
    layernorm_bwd_body<NCH, true, 1, true>(dy_bf16, dy_f32, x, stats, gamma, M, H, dres, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma,
                                           dbeta, dres_b16, dx_res_b16, dx_fp8, row_dequant);
}

static inline int ln_grid(int M) {
    int blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;  // 8 blocks x 4 waves per CU: every wave slot of the chip, grid-stride over rows
    return blocks < 1 ? 1 : blocks;
}

}  // namespace clibd

using namespace clibd;

// rows per wave and iteration (A/B at M = 403 456, H = 768: with the down-projection 479 -> 392 us at 2 rows, 391 at 4; without it
// 352-363 -> 335-347 us at 2, 330 at 4; profiles/r03_exp_layernorm_rows.log)
constexpr int LN_FWD_ROWS_LORA = 2, LN_FWD_ROWS_PLAIN = 4;
static int layernorm_fwd_impl(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                              void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                              uint32_t drop_seed, int drop_thr16, float drop_scale, void* y_fp8, float fp8_scale, void* stream) {
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "layernorm_fwd: bad dropout threshold");
    if (!x || !gamma || !beta) return set_error(CLIBD_EINVAL, "layernorm_fwd: null pointer");
    if (M <= 0 || H <= 0 || H % 64 != 0 || H > 1024) return set_error(CLIBD_EINVAL, "layernorm_fwd: H must be a multiple of 64, <= 1024");
    if (!y_bf16 && !y_f32 && !y_fp8) return set_error(CLIBD_EINVAL, "layernorm_fwd: no output");
    if (y_fp8 && (!(fp8_scale > 0.f) || ((uintptr_t)y_fp8 & 3))) return set_error(CLIBD_EINVAL, "layernorm_fwd: fp8 output needs a positive scale and 4-byte alignment");
    if ((lora_a_bf16 == nullptr) != (t_bf16 == nullptr)) return set_error(CLIBD_EINVAL, "layernorm_fwd: lora_a/t must come together");
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (y_f32 && !aligned16(y_f32)) ||
        (y_bf16 && !aligned16(y_bf16)) || (lora_a_bf16 && !aligned16(lora_a_bf16)))
        return set_error(CLIBD_EINVAL, "layernorm_fwd: alignment");
    const int nch = (H + 255) / 256;
    dim3 grid(ln_grid(M)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool lora = lora_a_bf16 != nullptr;
    const size_t lds = lora ? (size_t)8 * H * sizeof(float) : 0;
#define LAUNCH(N)                                                                                              \
    do {                                                                                                       \
        if (lora)                                                                                              \
            hipLaunchKernelGGL((layernorm_fwd_kernel<N, true, LN_FWD_ROWS_LORA>), grid, block, lds, st, x, M, H, gamma, beta, eps, \
                               (unsigned short*)y_bf16, y_f32, stats, (const unsigned short*)lora_a_bf16,      \
                               (unsigned short*)t_bf16, drop_seed, drop_thr16, drop_scale, (unsigned char*)y_fp8, fp8_scale); \
        else                                                                                                   \
            hipLaunchKernelGGL((layernorm_fwd_kernel<N, false, LN_FWD_ROWS_PLAIN>), grid, block, 0, st, x, M, H, gamma, beta, eps, \
                               (unsigned short*)y_bf16, y_f32, stats, (const unsigned short*)nullptr,          \
                               (unsigned short*)nullptr, drop_seed, drop_thr16, drop_scale, (unsigned char*)y_fp8, fp8_scale); \
    } while (0)
    switch (nch) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return check_launch("layernorm_fwd");
}

extern "C" int clibd_layernorm_fwd_drop(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                                        void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                                        uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    return layernorm_fwd_impl(x, M, H, gamma, beta, eps, y_bf16, y_f32, stats, lora_a_bf16, t_bf16, drop_seed, drop_thr16, drop_scale, nullptr, 0.f, stream);
}

extern "C" int clibd_layernorm_fwd(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                                   void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16,
                                   void* t_bf16, void* stream) {
    return layernorm_fwd_impl(x, M, H, gamma, beta, eps, y_bf16, y_f32, stats, lora_a_bf16, t_bf16, 0u, 0, 1.0f, nullptr, 0.f, stream);
}

extern "C" int clibd_layernorm_fwd_fp8(const float* x, int M, int H, const float* gamma, const float* beta, float eps,
                                       void* y_bf16, float* y_f32, float* stats, const void* lora_a_bf16, void* t_bf16,
                                       uint32_t drop_seed, int drop_thr16, float drop_scale, void* y_fp8, float fp8_scale, void* stream) {
    if (!y_fp8) return set_error(CLIBD_EINVAL, "layernorm_fwd_fp8: null y_fp8");
    return layernorm_fwd_impl(x, M, H, gamma, beta, eps, y_bf16, y_f32, stats, lora_a_bf16, t_bf16, drop_seed, drop_thr16, drop_scale, y_fp8, fp8_scale, stream);
}

static int layernorm_bwd_impl(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                              const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                              void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, float* dgamma, float* dbeta, void* stream,
                              const void* dres_b16 = nullptr, void* dx_res_b16 = nullptr, void* dx_fp8 = nullptr, float* row_dequant = nullptr) {
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "layernorm_bwd: bad dropout threshold");
    if (!x || !stats || !gamma) return set_error(CLIBD_EINVAL, "layernorm_bwd: null pointer");
    if ((dy_bf16 == nullptr) == (dy_f32 == nullptr)) return set_error(CLIBD_EINVAL, "layernorm_bwd: exactly one of dy_bf16/dy_f32");
    if (M <= 0 || H <= 0 || H % 64 != 0 || H > 1024) return set_error(CLIBD_EINVAL, "layernorm_bwd: H must be a multiple of 64, <= 1024");
    if (!dx_f32 && !dx_bf16 && !dx_res_b16 && !dx_fp8) return set_error(CLIBD_EINVAL, "layernorm_bwd: no output");
    if ((dx_fp8 == nullptr) != (row_dequant == nullptr)) return set_error(CLIBD_EINVAL, "layernorm_bwd: dx_fp8 / row_dequant must come together");
    if (dx_fp8 && (((uintptr_t)dx_fp8 & 3) || (H & 3))) return set_error(CLIBD_EINVAL, "layernorm_bwd: the fp8 output needs 4-byte alignment");
    if (dres_f32 && dres_b16) return set_error(CLIBD_EINVAL, "layernorm_bwd: the residual gradient is either fp32 or bf16");
    if (((uintptr_t)dres_b16 & 7) || ((uintptr_t)dx_res_b16 & 7)) return set_error(CLIBD_EINVAL, "layernorm_bwd: alignment");
    if ((dgamma == nullptr) != (dbeta == nullptr)) return set_error(CLIBD_EINVAL, "layernorm_bwd: dgamma/dbeta must come together");
    const int nch = (H + 255) / 256;
    const bool pg = dgamma != nullptr;
    // parameter-gradient mode: at most 4 blocks per CU-slot (1024 blocks): 2 x H float atomics per block stay ~1.5 M per launch
    dim3 grid(min((M + 3) / 4, pg ? 1024 : 4096)), block(256);
    hipStream_t st = (hipStream_t)stream;
    // rows per wave and iteration: two pay for the long launches of the bf16 residual-gradient form (M = 403 456: 716-722 -> 652 us;
    // at M = 50 432 one row is the faster form: 97 -> 85-87 us against 91); the fp32 / parameter-gradient forms are indifferent
    // and keep one (profiles/r03_exp_layernorm_rows.log)
    // the e4m3-row form: two rows (bounded to four waves per SIMD; round 6: the residual gradient held packed, 118 registers, none spilled) for the pre-LN call (incoming bf16 stream: 569 ->
    // 635 us at M = 403 456, the 11 / 10 byte ratio; one row: 683), one row for the post-LN call, whose masked copy it replaces (591 -> 599 us;
    // two rows: 561 -> 591): tools/bench_ln_bwd_fp8.py, profiles/r05_exp_ln_bwd_fp8_rows.log
    const bool two_rows = !pg && dy_f32 == nullptr && dres_f32 == nullptr && dx_f32 == nullptr && M >= 131072 && (dx_fp8 == nullptr || (nch <= 3 && dres_b16 != nullptr));
#define LAUNCH_R(N, R)                                                                                         \
    do {                                                                                                       \
        if (dx_fp8 && pg)                                                                                      \
            hipLaunchKernelGGL((layernorm_bwd_fp8_pg_kernel<N>), grid, block, 0, st, (const unsigned short*)dy_bf16, dy_f32, x, \
                               stats, gamma, M, H, dres_f32, dx_f32, (unsigned short*)dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma, dbeta, \
                               (const unsigned short*)dres_b16, (unsigned short*)dx_res_b16, (unsigned char*)dx_fp8, row_dequant); \
        else if (dx_fp8)                                                                                       \
            hipLaunchKernelGGL((layernorm_bwd_fp8_kernel<N, (N <= 3 ? R : 1)>), grid, block, 0, st, (const unsigned short*)dy_bf16, dy_f32, x, /* (two rows only up to H = 768: see two_rows) */ \
                               stats, gamma, M, H, dres_f32, dx_f32, (unsigned short*)dx_bf16, drop_seed, drop_thr16, drop_scale, \
                               (const unsigned short*)dres_b16, (unsigned short*)dx_res_b16, (unsigned char*)dx_fp8, row_dequant); \
        else if (pg)                                                                                            \
            hipLaunchKernelGGL((layernorm_bwd_kernel<N, true, 1>), grid, block, 0, st, (const unsigned short*)dy_bf16, dy_f32, x, \
                               stats, gamma, M, H, dres_f32, dx_f32, (unsigned short*)dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma, dbeta, \
                               (const unsigned short*)dres_b16, (unsigned short*)dx_res_b16);                  \
        else                                                                                                   \
            hipLaunchKernelGGL((layernorm_bwd_kernel<N, false, R>), grid, block, 0, st, (const unsigned short*)dy_bf16, dy_f32, x, \
                               stats, gamma, M, H, dres_f32, dx_f32, (unsigned short*)dx_bf16, drop_seed, drop_thr16, drop_scale, \
                               (float*)nullptr, (float*)nullptr, (const unsigned short*)dres_b16, (unsigned short*)dx_res_b16); \
    } while (0)
#define LAUNCH(N)                    \
    do {                             \
        if (two_rows) LAUNCH_R(N, 2); \
        else LAUNCH_R(N, 1);         \
    } while (0)
    switch (nch) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
#undef LAUNCH_R
    return check_launch("layernorm_bwd");
}

extern "C" int clibd_layernorm_bwd_drop(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                                        const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                                        void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, nullptr,
                              nullptr, stream);
}

extern "C" int clibd_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                                   const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                                   void* dx_bf16, void* stream) {
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, 0u, 0, 1.0f, nullptr, nullptr, stream);
}

extern "C" int clibd_layernorm_bwd_pg(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                                      const float* gamma, int M, int H, const float* dres_f32, float* dx_f32,
                                      void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, float* dgamma, float* dbeta,
                                      void* stream) {
    if (!dgamma || !dbeta) return set_error(CLIBD_EINVAL, "layernorm_bwd_pg: null dgamma/dbeta");
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma, dbeta,
                              stream);
}

extern "C" int clibd_layernorm_bwd_res16(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats,
                                         const float* gamma, int M, int H, const void* dres_bf16, void* dx_res_bf16,
                                         void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, nullptr, nullptr, dx_bf16, drop_seed, drop_thr16, drop_scale, nullptr,
                              nullptr, stream, dres_bf16, dx_res_bf16);
}

extern "C" int clibd_layernorm_bwd_fp8(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                                       int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                                       void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* dx_fp8, float* row_dequant,
                                       void* stream) {
    if (!dx_fp8 || !row_dequant) return set_error(CLIBD_EINVAL, "layernorm_bwd_fp8: null dx_fp8 / row_dequant");
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, nullptr,
                              nullptr, stream, dres_bf16, dx_res_bf16, dx_fp8, row_dequant);
}

extern "C" int clibd_layernorm_bwd_fp8_pg(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                                          int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                                          void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, void* dx_fp8, float* row_dequant,
                                          float* dgamma, float* dbeta, void* stream) {
    if (!dx_fp8 || !row_dequant || !dgamma || !dbeta) return set_error(CLIBD_EINVAL, "layernorm_bwd_fp8_pg: null dx_fp8 / row_dequant / dgamma / dbeta");
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma,
                              dbeta, stream, dres_bf16, dx_res_bf16, dx_fp8, row_dequant);
}

extern "C" int clibd_layernorm_bwd_any(const void* dy_bf16, const float* dy_f32, const float* x, const float* stats, const float* gamma,
                                       int M, int H, const float* dres_f32, const void* dres_bf16, float* dx_f32, void* dx_res_bf16,
                                       void* dx_bf16, uint32_t drop_seed, int drop_thr16, float drop_scale, float* dgamma, float* dbeta,
                                       void* stream) {
    return layernorm_bwd_impl(dy_bf16, dy_f32, x, stats, gamma, M, H, dres_f32, dx_f32, dx_bf16, drop_seed, drop_thr16, drop_scale, dgamma,
                              dbeta, stream, dres_bf16, dx_res_bf16);
}
