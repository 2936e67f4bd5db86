// Multi-head attention forward / backward for short sequences (S <= 256, head dim 64) on gfx950.
//
// One workgroup (4 waves) per (batch, head).  The whole K/V (forward) or K,V then Q,dO (backward) of that
// head lives in LDS as [S_pad][64] bf16 tiles (128-B rows, 16-B chunk index XOR (row&7)), filled by 16-byte
// LDS-DMA with the swizzle applied on the per-lane source address.  Scores for one 16-query tile against
// ALL keys fit in registers (<= 16 key tiles x 4 fp32), so softmax is an exact full-row softmax: no online
// rescaling, no saved LSE.
//
// Orientation ("key on the MFMA row, query on the lane"): S^T = K·Q^T via
// v_mfma_f32_16x16x32_bf16(A = K rows, B = Q rows) leaves each lane with ONE query (lane&15) and keys
// 16*kt + 4*(lane>>4) + reg.  Those accumulators, packed to bf16, are directly the B operand of the next
// product that contracts over keys (P·V, dS·K) with the k-slot -> key map
//      kappa(g, j) = 32*s + 16*(j>>2) + 4*g + (j&3)        (g = lane>>4, j = 0..7)
// and the other operand (V^T / K^T rows = head-dim) is fetched with ds_read_b64_tr_b16 using the same map.
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"
#include <stdlib.h>

namespace clibd {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int ATT_THREADS = 256;
constexpr int ATT_WAVES = 4;
constexpr int DH = 64;
constexpr int MAX_KT = 16;  // S_pad <= 256

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// stage rows [0,S_pad) x 64 bf16 of one head (global row stride `ld` elements) into an LDS tile
__device__ __forceinline__ void stage_head_tile(char* lds_tile, const unsigned short* gbase, size_t ld, int S, int S_pad,
                                                int wave, int lane, int nwaves = ATT_WAVES) {
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    for (int p = wave; p < (S_pad >> 3); p += nwaves) {
        const int row = min(p * 8 + prow, S - 1);
        glds16(gbase + (size_t)row * ld + chunk * 8, lds_tile + p * 1024);
    }
}

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int row, int ks, int g) {
    return *(const bf16x8*)(tile + tile_off(row, 4 * ks + g));
}

// transposed fragment: rows d = 16*dt + (lane&15), k-slots (g,j) -> tile rows kappa(g,j) = 32*s+16*(j>>2)+4*g+(j&3)
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int s, int dt, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int q = i >> 2, p = i & 3;
    const int r0 = 32 * s + 4 * g + q;
    const int ch = 2 * dt + (p >> 1);
    const int a0 = tile_off(r0, ch) + 8 * (p & 1);
    const int a1 = tile_off(r0 + 16, ch) + 8 * (p & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a1));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
    r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
    return r;
}

// across the 4 lane groups (lane>>4) at fixed lane&15: v_permlane swaps, not ds_bpermute (common.h)
__device__ __forceinline__ float group4_sum(float v) { return rows4_sum(v); }
__device__ __forceinline__ float group4_max(float v) { return rows4_max(v); }

// One 16-row x 64-column output tile of a wave: lane (i = lane & 15, g = lane >> 4) holds v[dt][r] = element (row i, column
// 16 dt + 4 g + r) — 8 bytes per (lane, dt).  Written that way (four dwordx2 per lane) a store instruction covers 32-byte pieces
// and the output tail of the attention kernels is store-issue bound.  A v_permlane16_swap between the lane rows g and g ^ 1 (a
// 2 x 2 transpose of the (dt, dt + 1) chunks) leaves every lane 16 contiguous bytes, and each of the TWO dwordx4 stores of a tile
// then covers 64 contiguous bytes of all 16 rows.  Every lane must call this (the swap is a full-wave operation); `on` guards
// the stores only (it is a function of the row, identical in the lanes that exchange).  Values, and so results, are unchanged.
__device__ __forceinline__ void permlane16_swap_u32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void store_rows16(unsigned short* row, const f32x4 (&v)[4], float mul, bool on, int g) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        unsigned a0 = pack2bf(v[2 * pr][0] * mul, v[2 * pr][1] * mul), a1 = pack2bf(v[2 * pr][2] * mul, v[2 * pr][3] * mul);
        unsigned b0 = pack2bf(v[2 * pr + 1][0] * mul, v[2 * pr + 1][1] * mul), b1 = pack2bf(v[2 * pr + 1][2] * mul, v[2 * pr + 1][3] * mul);
        permlane16_swap_u32(a0, b0);   // even g: (a, b) = columns 4g .. 4g+7 of tile dt = 2 pr;  odd g: columns 4(g-1) .. 4(g-1)+7 of dt = 2 pr + 1
        permlane16_swap_u32(a1, b1);
        if (on) *(uint4*)(row + 16 * (2 * pr + (g & 1)) + 8 * (g >> 1)) = make_uint4(a0, a1, b0, b1);
    }
}

// the (dt, dt + 1) = (2 pr, 2 pr + 1) half of store_rows16, for a wave that holds only those two column tiles
__device__ __forceinline__ void store_rows16_pair(unsigned short* row, const f32x4& va, const f32x4& vb, float mul, bool on, int g, int pr) {
    unsigned a0 = pack2bf(va[0] * mul, va[1] * mul), a1 = pack2bf(va[2] * mul, va[3] * mul);
    unsigned b0 = pack2bf(vb[0] * mul, vb[1] * mul), b1 = pack2bf(vb[2] * mul, vb[3] * mul);
    permlane16_swap_u32(a0, b0);
    permlane16_swap_u32(a1, b1);
    if (on) *(uint4*)(row + 16 * (2 * pr + (g & 1)) + 8 * (g >> 1)) = make_uint4(a0, a1, b0, b1);
}
// Training forward (single-pass backward, attention_bwd_sp_kernel): beside out = bf16(o * mul) also the bf16 residual of that
// rounding, o_lo = bf16(o * mul - float(bf16(o * mul))), so that the backward can take delta = dO . (o_hi + o_lo) with fp32-class
// accuracy (delta from the bf16 output alone costs the q-adapter gradients 4-7 %: DESIGN.md §3.2).
__device__ __forceinline__ void store_rows16_lo(unsigned short* row_lo, const f32x4 (&v)[4], float mul, bool on, int g) {
    f32x4 lo[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = v[dt][r] * mul;
            lo[dt][r] = x - bfround(x);
        }
    store_rows16(row_lo, lo, 1.0f, on, g);
}

constexpr float NEG_BIG = -1.0e30f;

#ifdef CLIBD_GEMM_DIAG
// diagnostic build only (python -m clibd_amd.build --diag; tools/att_stamps.py): s_memtime at the phase boundaries of the backward
__device__ long long* g_att_stamps = nullptr;   // [workgroup][8]
#define ATT_STAMP(k)                                                                                      \
    do {                                                                                                  \
        if (g_att_stamps != nullptr && threadIdx.x == 0) g_att_stamps[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
// per-wave stamps of the single-pass backward: [workgroup][wave][16]
#define SP_STAMP(k)                                                                                       \
    do {                                                                                                  \
        if (g_att_stamps != nullptr && (threadIdx.x & 63) == 0)                                           \
            g_att_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ATT_STAMP(k) do { } while (0)
#define SP_STAMP(k) do { } while (0)
#endif

// ============================================ forward ==========================================================
// MASK: a key-padding mask is present (text tower).  Compile-time, because the per-key mask loads and their divergent
// select code — emitted for every (key tile, register) pair when the test is a run-time pointer check — cost the unmasked
// towers ~400 SGPR-spill v_readlane/v_writelane and ~450 exec-mask scalar ops per query tile (ISA of the round-1 kernel).
// DROP: dropout on the probabilities (HF BERT train mode); compile-time for the same reason (its counter hash is ~7 VALU per score).
// NW: waves per workgroup (4; 3 for nine query tiles — S in (128, 144], the DNA tower — where four waves would take 3 + 2 + 2 + 2 tiles
// one by one or, two at a time, 2 + 1 + 1 + 1 pairs of which the last is half padding: three waves take 3 + 3 + 3 single tiles, capped at
// three waves per SIMD so that four such workgroups share a CU).
// IMG: rows per LDS image (S_pad; 144 in the three-wave form: the all-padding tile 9 is never read from the K image, and rows 144 .. 159 of the
// V image, read by the last k-slot against probabilities that are exactly 0, are a zeroed 2-KiB pad: 38 912 B per workgroup, four per CU)
template <int NKT, bool PAIR, bool MASK, bool DROP, int NW = ATT_WAVES, int IMG = 16 * NKT>  // NKT: number of 16-key tiles, even (S_pad = 16*NKT, multiple of 32); PAIR: two query tiles per sweep
__global__ __launch_bounds__(64 * NW, (NW == 3 ? 3 : 2)) void attention_fwd_kernel(const unsigned short* __restrict__ qkv, int S,
                                                                    int nheads, const int* __restrict__ key_mask,
                                                                    unsigned short* __restrict__ out, float scale,
                                                                    int nq, int out_seq, unsigned drop_seed,
                                                                    int drop_thr16, float drop_scale, float out_fp8_scale,
                                                                    float* __restrict__ lse, unsigned short* __restrict__ o_lo) {
    // out_fp8_scale > 0 (fp8-forward mode): `out` holds OCP e4m3 bytes, fp8(o * out_fp8_scale) — the projection GEMM's operand
    // lse / o_lo (optional, training forward): per (head, query) log2-domain log-sum-exp c2 * max + log2(sum) of the UN-dropped
    // scores, [B * nheads, S] fp32, and the bf16 residual of the output rounding (store_rows16_lo), laid out like `out`
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S_pad = 16 * NKT;
    static_assert(IMG == S_pad || (IMG == S_pad - 16 && !PAIR), "a short image drops exactly the all-padding last tile (one query tile per sweep only)");
    char* kt_lds = smem;
    char* vt_lds = smem + IMG * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / nheads, h = blockIdx.x % nheads;
    const int H = nheads * DH;
    const size_t ld = (size_t)3 * H;
    const unsigned short* qbase = qkv + (size_t)b * S * ld + h * DH;
    const int g = lane >> 4, i = lane & 15;
    const int nqt = (nq + 15) >> 4;  // only the first nq query rows are evaluated (nq = S normally; 1 = [CLS]-only last ViT block)
    const float c2 = scale * 1.4426950408889634f;  // p = exp2(c2 * s - c2 * max): one FMA + one v_exp per score
    // NKT is even (k-slots of 32 keys); when S <= 16 (NKT - 1) the last key tile is all padding (S = 197 -> 13 live tiles of
    // 14, S = 133 -> 9 of 10): its score MFMAs and exponentials are skipped (probabilities exactly 0, as the mask gives)
    const bool last_live = (IMG == S_pad) && S > 16 * (NKT - 1);   // (a short image is only launched with S <= 16 (NKT - 1): compile-time false)
    // this wave's first Q fragment rides along with the K/V staging; later ones are prefetched a tile ahead
    bf16x8 qf[2];
    {
        const int qc0 = min(wave * 16 + i, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(qbase + (size_t)qc0 * ld + 32 * ks + 8 * g);
    }
    stage_head_tile(kt_lds, qbase + H, ld, S, IMG, wave, lane, NW);
    stage_head_tile(vt_lds, qbase + 2 * H, ld, S, IMG, wave, lane, NW);
    if constexpr (IMG < S_pad) {
        for (int r = threadIdx.x; r < (S_pad - IMG) * 8; r += 64 * NW) *(uint4*)(vt_lds + IMG * 128 + 16 * r) = make_uint4(0u, 0u, 0u, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if constexpr (PAIR) {
        // Two query tiles per wave and sweep: every K row fragment and every transposed V fragment read from LDS feeds two
        // MFMAs (the one-tile sweep moves 1 KiB of LDS per MFMA and is bound by it).  Arithmetic per element is unchanged.
        for (int p = wave; p < ((nqt + 1) >> 1); p += NW) {
            bf16x8 qf2[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qc = min((2 * p + t) * 16 + i, S - 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qf2[t][ks] = *(const bf16x8*)(qbase + (size_t)qc * ld + 32 * ks + 8 * g);
            }
            f32x4 sc[2][NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                sc[0][kt] = (f32x4){0, 0, 0, 0};
                sc[1][kt] = (f32x4){0, 0, 0, 0};
                if (kt < NKT - 1 || last_live) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8 kfr = lds_row_frag(kt_lds, kt * 16 + i, ks, g);
                        sc[0][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf2[0][ks], sc[0][kt], 0, 0, 0);
                        sc[1][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf2[1][ks], sc[1][kt], 0, 0, 0);
                    }
                }
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);  // at most four K fragments in flight: 2 x NKT score quads leave no room for more
            }
            float inv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float mx = NEG_BIG;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    if (kt == NKT - 1 && !last_live) continue;
                    if (MASK || (kt >= NKT - 2 && kt * 16 + 15 >= S)) {   // S > 16 (NKT - 2): only the last two tiles can hold padding keys
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = kt * 16 + 4 * g + r;
                            bool ok = key < S;
                            if (MASK) ok = ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
                            if (!ok) sc[t][kt][r] = NEG_BIG;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][kt][r]);
                }
                mx = group4_max(mx);
                const float mc = mx * c2;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    if (kt == NKT - 1 && !last_live) continue;   // sc stays exactly 0: the probabilities of an all-padding tile
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc[t][kt][r], c2, -mc));
                        sc[t][kt][r] = e;
                        sum += e;
                    }
                }
                sum = group4_sum(sum);
                inv[t] = 1.0f / sum;
                if (lse != nullptr && g == 0) {
                    const int qq = (2 * p + t) * 16 + i;
                    if (qq < nq) lse[(size_t)blockIdx.x * S + qq] = mc + __builtin_amdgcn_logf(sum);
                }
            }
            f32x4 o[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NKT / 2; ++s) {
                bf16x8 pf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (DROP) {
                        const unsigned q = (unsigned)((2 * p + t) * 16 + i);
                        const unsigned base = (((unsigned)blockIdx.x * (unsigned)S + q) << 8) + 32u * s + 4u * g;
                        float f0, f1, f2, f3;
                        drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                        drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                        sc[t][2 * s][0] *= f0; sc[t][2 * s][1] *= f1; sc[t][2 * s][2] *= f2; sc[t][2 * s][3] *= f3;
                        drop_pair(drop_seed, base + 16, (unsigned)drop_thr16, drop_scale, f0, f1);
                        drop_pair(drop_seed, base + 18, (unsigned)drop_thr16, drop_scale, f2, f3);
                        sc[t][2 * s + 1][0] *= f0; sc[t][2 * s + 1][1] *= f1; sc[t][2 * s + 1][2] *= f2; sc[t][2 * s + 1][3] *= f3;
                    }
                    pf[t] = pack_frag(sc[t][2 * s], sc[t][2 * s + 1]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vtr = lds_tr_frag(vt_lds, s, dt, lane);
                    o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtr, pf[0], o[0][dt], 0, 0, 0);
                    o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtr, pf[1], o[1][dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = (2 * p + t) * 16 + i;
                const size_t oidx = ((size_t)b * out_seq + min(q, nq - 1)) * H + h * DH;
                if (out_fp8_scale > 0.f) {
                    if (q < nq) {
                        unsigned char* orow8 = (unsigned char*)out + oidx;
                        const float sc8 = inv[t] * out_fp8_scale;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            *(unsigned*)(orow8 + 16 * dt + 4 * g) = pack4fp8(o[t][dt][0] * sc8, o[t][dt][1] * sc8, o[t][dt][2] * sc8, o[t][dt][3] * sc8);
                    }
                } else {
                    store_rows16(out + oidx, o[t], inv[t], q < nq, g);
                    if (o_lo != nullptr) store_rows16_lo(o_lo + oidx, o[t], inv[t], q < nq, g);
                }
            }
        }
        return;
    }
    for (int qt = wave; qt < nqt; qt += NW) {
        const int q = qt * 16 + i;
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            sc[kt] = (f32x4){0, 0, 0, 0};
            if (kt == NKT - 1 && !last_live) continue;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(kt_lds, kt * 16 + i, ks, g), qf[ks], sc[kt], 0, 0, 0);
        }
        {   // prefetch the next tile's Q fragment (clamped; unused after the last tile)
            const int qn = min((qt + NW) * 16 + i, S - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(qbase + (size_t)qn * ld + 32 * ks + 8 * g);
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt == NKT - 1 && !last_live) continue;
            if (MASK || (kt >= NKT - 2 && kt * 16 + 15 >= S)) {  // only tiles that can hold masked keys pay for the test
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    bool ok = key < S;
                    if (MASK) ok = ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
                    if (!ok) sc[kt][r] = NEG_BIG;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
        }
        mx = group4_max(mx);
        const float mc = mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt == NKT - 1 && !last_live) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c2, -mc));  // masked scores underflow to exactly 0
                sc[kt][r] = e;
                sum += e;
            }
        }
        sum = group4_sum(sum);
        const float inv = 1.0f / sum;
        if (lse != nullptr && g == 0 && q < nq) lse[(size_t)blockIdx.x * S + q] = mc + __builtin_amdgcn_logf(sum);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
            if (DROP) {  // dropout on the probabilities (HF BertSelfAttention.dropout); index ((b*nh+h)*S+q)*256+key
                const unsigned base = (((unsigned)blockIdx.x * (unsigned)S + (unsigned)q) << 8) + 32u * s + 4u * g;
                float f0, f1, f2, f3;
                drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                sc[2 * s][0] *= f0; sc[2 * s][1] *= f1; sc[2 * s][2] *= f2; sc[2 * s][3] *= f3;
                drop_pair(drop_seed, base + 16, (unsigned)drop_thr16, drop_scale, f0, f1);
                drop_pair(drop_seed, base + 18, (unsigned)drop_thr16, drop_scale, f2, f3);
                sc[2 * s + 1][0] *= f0; sc[2 * s + 1][1] *= f1; sc[2 * s + 1][2] *= f2; sc[2 * s + 1][3] *= f3;
            }
            const bf16x8 pf = pack_frag(sc[2 * s], sc[2 * s + 1]);  // un-normalised probabilities (<= 1/(1-p)); 1/sum is applied to O
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(vt_lds, s, dt, lane), pf, o[dt], 0, 0, 0);
        }
        {
            const size_t oidx = ((size_t)b * out_seq + min(q, nq - 1)) * H + h * DH;
            if (out_fp8_scale > 0.f) {
                if (q < nq) {
                    unsigned char* orow8 = (unsigned char*)out + oidx;
                    const float sc8 = inv * out_fp8_scale;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *(unsigned*)(orow8 + 16 * dt + 4 * g) = pack4fp8(o[dt][0] * sc8, o[dt][1] * sc8, o[dt][2] * sc8, o[dt][3] * sc8);
                }
            } else {
                store_rows16(out + oidx, o, inv, q < nq, g);
                if (o_lo != nullptr) store_rows16_lo(o_lo + oidx, o, inv, q < nq, g);
            }
        }
    }
}

// ============================================ forward, persistent =================================================
// Long sequences (S > 160: ViT 197, and anything up to 256).  The per-head kernel above spends 39 of its 88 us (ViT, b = 256) waiting
// for its K / V tiles: the two workgroups of a CU are launched together and last equally long, so they stage together and
// compute together (tools/exp_attn_parts.py: the time against the number of query tiles evaluated is a 39-us floor plus the
// compute, not their maximum).  Here ONE workgroup of 16 waves per CU walks heads blockIdx.x, + gridDim.x, ...: the K / V image
// is double-buffered (2 x 2 x S_pad x 128 B <= 112 KiB), the LDS-DMA of the NEXT head is issued before the current head's
// query tiles are evaluated, and one barrier per head hands the buffers over.  A head has at most 16 query tiles: wave w takes
// tile (w + 3 i) mod 16 of the workgroup's i-th head (the idle slots of a 13-tile head rotate over the SIMDs), one 16-query tile
// per wave — four waves per SIMD (<= 128 VGPRs) hide each other's LDS and MFMA latencies, so the two-tile sweep of the kernel
// above is not needed (the LDS array is 8 % busy in that kernel: SQ_LDS_IDX_ACTIVE).  Arithmetic per element, and therefore
// every result bit, is that of the per-head kernel.
constexpr int ATTP_WAVES = 16;
template <int NKT, bool MASK, bool DROP>
__global__ __launch_bounds__(64 * ATTP_WAVES) void attention_fwd_persistent_kernel(const unsigned short* __restrict__ qkv, int S, int nheads,
                                                                               int total_heads, const int* __restrict__ key_mask,
                                                                               unsigned short* __restrict__ out, float scale, int nq,
                                                                               int out_seq, unsigned drop_seed, int drop_thr16,
                                                                               float drop_scale, float out_fp8_scale,
                                                                               float* __restrict__ lse, unsigned short* __restrict__ o_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S_pad = 16 * NKT;
    constexpr int BUF = 2 * S_pad * 128;   // K then V of one head
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = nheads * DH;
    const size_t ld = (size_t)3 * H;
    const int g = lane >> 4, i = lane & 15;
    const int nqt = (nq + 15) >> 4;
    const float c2 = scale * 1.4426950408889634f;
    const bool last_live = S > 16 * (NKT - 1);

    auto head_base = [&](int head) { return qkv + (size_t)(head / nheads) * S * ld + (size_t)(head % nheads) * DH; };
    auto stage = [&](int head, char* buf) {
        const unsigned short* qb = head_base(head);
        stage_head_tile(buf, qb + H, ld, S, S_pad, wave, lane, ATTP_WAVES);
        stage_head_tile(buf + S_pad * 128, qb + 2 * H, ld, S, S_pad, wave, lane, ATTP_WAVES);
    };

    int head = blockIdx.x;
    if (head >= total_heads) return;
    stage(head, smem);
    // Query fragments are requested one head ahead, BEFORE the current head's output stores: vmcnt retires in issue order, so the
    // wait at the top of a head (this wave's LDS-DMA pieces) can leave the output stores (two 16-byte stores per lane, four 4-byte
    // ones in fp8 mode) in flight; the compiler's own wait in front of the first MFMA covers the fragments.
    auto load_q = [&](int hd, int it_, bf16x8 (&qf_)[2]) {
        const int qt_ = (wave + 3 * it_) & 15;
        if (qt_ < nqt) {
            const unsigned short* qb = head_base(hd);
            const int qc = min(qt_ * 16 + i, S - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf_[ks] = *(const bf16x8*)(qb + (size_t)qc * ld + 32 * ks + 8 * g);
        }
    };
    bf16x8 qf[2] = {}, qn[2] = {};
    load_q(head, 0, qf);
    bool stored = false;   // wave-uniform: this wave issued output stores in the previous head
    for (int it = 0; head < total_heads; ++it, head += gridDim.x) {
        char* kt_lds = smem + (it & 1) * BUF;
        char* vt_lds = kt_lds + S_pad * 128;
        const int b = head / nheads, h = head - b * nheads;
        const int qt = (wave + 3 * it) & 15;
        const bool mine = qt < nqt;
        const int q = qt * 16 + i;
        // this head's image: issued one head ago (or just above); the wave's own pieces are awaited, the barrier publishes the rest
        if (stored && lse == nullptr) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (the training forward stores more per head: full wait)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int next = head + (int)gridDim.x;
        if (next < total_heads) {
            stage(next, smem + ((it + 1) & 1) * BUF);  // flies under this head's arithmetic
            load_q(next, it + 1, qn);
        }
        stored = mine;
        if (mine) {
            f32x4 sc[NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                sc[kt] = (f32x4){0, 0, 0, 0};
                if (kt == NKT - 1 && !last_live) continue;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(kt_lds, kt * 16 + i, ks, g), qf[ks], sc[kt], 0, 0, 0);
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt == NKT - 1 && !last_live) continue;
                if (MASK || (kt >= NKT - 2 && kt * 16 + 15 >= S)) {  // only tiles that can hold masked keys pay for the test
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + 4 * g + r;
                        bool ok = key < S;
                        if (MASK) ok = ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
                        if (!ok) sc[kt][r] = NEG_BIG;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
            }
            mx = group4_max(mx);
            const float mc = mx * c2;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt == NKT - 1 && !last_live) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c2, -mc));  // masked scores underflow to exactly 0
                    sc[kt][r] = e;
                    sum += e;
                }
            }
            sum = group4_sum(sum);
            const float inv = 1.0f / sum;
            if (lse != nullptr && g == 0 && q < nq) lse[(size_t)head * S + q] = mc + __builtin_amdgcn_logf(sum);
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NKT / 2; ++s) {
                if (DROP) {  // dropout on the probabilities (HF BertSelfAttention.dropout); index ((b*nh+h)*S+q)*256+key
                    const unsigned base = (((unsigned)head * (unsigned)S + (unsigned)q) << 8) + 32u * s + 4u * g;
                    float f0, f1, f2, f3;
                    drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                    drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                    sc[2 * s][0] *= f0; sc[2 * s][1] *= f1; sc[2 * s][2] *= f2; sc[2 * s][3] *= f3;
                    drop_pair(drop_seed, base + 16, (unsigned)drop_thr16, drop_scale, f0, f1);
                    drop_pair(drop_seed, base + 18, (unsigned)drop_thr16, drop_scale, f2, f3);
                    sc[2 * s + 1][0] *= f0; sc[2 * s + 1][1] *= f1; sc[2 * s + 1][2] *= f2; sc[2 * s + 1][3] *= f3;
                }
                const bf16x8 pf = pack_frag(sc[2 * s], sc[2 * s + 1]);  // un-normalised probabilities (<= 1/(1-p)); 1/sum is applied to O
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(vt_lds, s, dt, lane), pf, o[dt], 0, 0, 0);
            }
            {
                const size_t oidx = ((size_t)b * out_seq + min(q, nq - 1)) * H + h * DH;
                if (out_fp8_scale > 0.f) {
                    if (q < nq) {
                        unsigned char* orow8 = (unsigned char*)out + oidx;
                        const float sc8 = inv * out_fp8_scale;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            *(unsigned*)(orow8 + 16 * dt + 4 * g) = pack4fp8(o[dt][0] * sc8, o[dt][1] * sc8, o[dt][2] * sc8, o[dt][3] * sc8);
                    }
                } else {
                    store_rows16(out + oidx, o, inv, q < nq, g);
                    if (o_lo != nullptr) store_rows16_lo(o_lo + oidx, o, inv, q < nq, g);
                }
            }
        }
        qf[0] = qn[0]; qf[1] = qn[1];
    }
}

// ============================================ backward =========================================================
// PAIR: phase 2 sweeps two key tiles per wave (long sequences: halves the LDS traffic per MFMA; costs ~60 VGPRs, so the
// short-sequence instantiations, which fit three waves per SIMD without it, keep the one-tile sweep)
// NW: waves per workgroup.  4 everywhere except S in (128, 144] (the DNA tower's 133 tokens = 9 tiles of 16): four waves take
// 3 + 2 + 2 + 2 of the nine query tiles in phase 1 and of the nine key tiles in phase 2, so the workgroup lives for 6 tile sweeps while
// its average wave has 4.5 to do; THREE waves take 3 + 3 + 3.  The register cap stays at three waves per SIMD, i.e. FOUR such
// workgroups per CU, and for four of them to fit the LDS each image is 144 rows instead of S_pad = 160 (IMG): rows 144 .. 159 of the
// first image then alias rows 0 .. 15 of the second, those of the second a zeroed 2-KiB pad.  Every such row is only ever multiplied
// by an exact zero (key tile 9 / query tile 9 are skipped: `last_live` is false, `nqt` = 9), so finite bytes are all that is needed.
// -DCLIBD_ATT_BWD_LONG_WAVES=3 (A/B knob, round 6): the register cap of the long-sequence instantiations (NKT > 10: the ViT's 14 tiles, 222
// registers at two waves per SIMD) forced to three waves per SIMD = 168 registers, whatever that spills (profiles/r06_exp_attention_vit_waves.log)
#ifndef CLIBD_ATT_BWD_LONG_WAVES
#define CLIBD_ATT_BWD_LONG_WAVES 2
#endif
// waves per SIMD the register allocator is held to: three for the short forms (NKT <= 10), two for the long ones.  The four-wave ten-tile form
// without a key mask (S in (144, 160]) spills 6 registers at the three-wave cap; round 6 measured the spill-free alternative (two waves per SIMD, 170-174
// registers: -DCLIBD_ATT_BWD_S160_WAVES=2): 177-182 us against 163-164 us per launch at S = 150, b = 256 — the spilling form is 10 % FASTER and stays
// (profiles/r06_exp_attention_spills.log).
#ifndef CLIBD_ATT_BWD_S160_WAVES
#define CLIBD_ATT_BWD_S160_WAVES 3
#endif
constexpr int att_bwd_min_waves(int NKT, bool MASK, int NW) {
    return NKT > 10 ? CLIBD_ATT_BWD_LONG_WAVES : (NKT == 10 && NW == 4 && !MASK) ? CLIBD_ATT_BWD_S160_WAVES : 3;
}
template <int NKT, bool PAIR, bool MASK, bool DROP, int NW = ATT_WAVES, int IMG = 16 * NKT>
__global__ __launch_bounds__(64 * NW, att_bwd_min_waves(NKT, MASK, NW)) void attention_bwd_kernel(const unsigned short* __restrict__ qkv,
                                                                    const unsigned short* __restrict__ dout, int S,
                                                                    int nheads, const int* __restrict__ key_mask,
                                                                    unsigned short* __restrict__ dqkv, float scale,
                                                                    int nq, int dout_seq, unsigned drop_seed,
                                                                    int drop_thr16, float drop_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S_pad = 16 * NKT;
    static_assert(IMG == S_pad || (IMG == S_pad - 16 && !PAIR), "a short image drops exactly the all-padding last tile, one-tile phase 2 only");
    constexpr int PADB = (IMG < S_pad) ? (S_pad - IMG) * 128 : 0;   // zeroed rows IMG .. S_pad-1 of the second image
    char* t0 = smem;                      // phase 1: K   | phase 2: Q
    char* t1 = smem + IMG * 128;          // phase 1: V   | phase 2: dO
    // per query row: m' = c2 * max + log2(row sum), so P = exp2(c2 * s - m') is the normalised probability with one FMA and one
    // v_exp; rows that carry no gradient (q >= nq) hold m' = +BIG, i.e. P = 0 exactly, and need no per-element test in phase 2
    float* st_m = (float*)(smem + 2 * IMG * 128 + PADB);
    float* st_d = st_m + S_pad;                       // delta = sum_k P dP
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / nheads, h = blockIdx.x % nheads;
    const int H = nheads * DH;
    const size_t ld = (size_t)3 * H;
    const unsigned short* qbase = qkv + (size_t)b * S * ld + h * DH;
    const unsigned short* dobase = dout + (size_t)b * dout_seq * H + h * DH;  // dO holds rows [0, nq) of every sequence
    unsigned short* dqbase = dqkv + (size_t)b * S * ld + h * DH;
    const int g = lane >> 4, i = lane & 15;

    const float c2 = scale * 1.4426950408889634f;
    bf16x8 qf[2], dof[2];
    {
        const int qc0 = min(wave * 16 + i, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *(const bf16x8*)(qbase + (size_t)qc0 * ld + 32 * ks + 8 * g);
            dof[ks] = *(const bf16x8*)(dobase + (size_t)min(qc0, nq - 1) * H + 32 * ks + 8 * g);
        }
    }
    ATT_STAMP(0);
    stage_head_tile(t0, qbase + H, ld, S, IMG, wave, lane, NW);
    stage_head_tile(t1, qbase + 2 * H, ld, S, IMG, wave, lane, NW);
    for (int r = threadIdx.x; r < S_pad; r += 64 * NW) { st_m[r] = -NEG_BIG; st_d[r] = 0.f; }
    if constexpr (PADB > 0) {
        for (int r = threadIdx.x; r < PADB / 16; r += 64 * NW) *(uint4*)(t1 + IMG * 128 + 16 * r) = make_uint4(0u, 0u, 0u, 0u);
    }
    // S <= 16 (NKT - 1): the last key tile is all padding and is skipped (see forward).  With a short image (IMG < S_pad: the host takes
    // that form only for such S) this is known at compile time, and sc[NKT-1] / dp[NKT-1] never exist: the 8 registers the three-wave
    // form was spilling (6 VGPRs, 28 bytes of scratch: VERDICT r4 weak 8)
    const bool last_live = (IMG == S_pad) && S > 16 * (NKT - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    ATT_STAMP(1);
    // ---------------- phase 1: per 16-query tile: softmax statistics, dS, dQ ----------------
    const int nqt = (nq + 15) >> 4;      // query tiles that carry a gradient (rows >= nq have dO = 0)
    const int nqt_all = (S + 15) >> 4;
    for (int qt = nqt + wave; qt < nqt_all; qt += NW) {  // dQ of the inactive query rows is exactly zero
        const int q = qt * 16 + i;
        if (q < S) {
            unsigned short* orow = dqbase + (size_t)q * ld;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(uint2*)(orow + 16 * dt + 4 * g) = make_uint2(0u, 0u);
        }
    }
    for (int qt = wave; qt < nqt; qt += NW) {
        const int q = qt * 16 + i;
        if (q >= nq) {  // rows of an active tile beyond nq: their dO is zero
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dof[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        f32x4 sc[NKT], dp[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            sc[kt] = (f32x4){0, 0, 0, 0};
            dp[kt] = (f32x4){0, 0, 0, 0};
            if (kt == NKT - 1 && !last_live) continue;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(t0, kt * 16 + i, ks, g), qf[ks], sc[kt], 0, 0, 0);
                dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(t1, kt * 16 + i, ks, g), dof[ks], dp[kt], 0, 0, 0);
            }
        }
        if (DROP) {  // O = (P o M / (1-p)) V  =>  dP = (dO V^T) o M / (1-p)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt == NKT - 1 && !last_live) continue;
                const unsigned base = (((unsigned)blockIdx.x * (unsigned)S + (unsigned)q) << 8) + 16u * kt + 4u * g;
                float f0, f1, f2, f3;
                drop_pair(drop_seed, base, (unsigned)drop_thr16, drop_scale, f0, f1);
                drop_pair(drop_seed, base + 2, (unsigned)drop_thr16, drop_scale, f2, f3);
                dp[kt][0] *= f0; dp[kt][1] *= f1; dp[kt][2] *= f2; dp[kt][3] *= f3;
            }
        }
        {   // prefetch the next tile's Q / dO fragments
            const int qn = min((qt + NW) * 16 + i, S - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[ks] = *(const bf16x8*)(qbase + (size_t)qn * ld + 32 * ks + 8 * g);
                dof[ks] = *(const bf16x8*)(dobase + (size_t)min(qn, nq - 1) * H + 32 * ks + 8 * g);
            }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt == NKT - 1 && !last_live) continue;
            if (MASK || (kt >= NKT - 2 && kt * 16 + 15 >= S)) {  // only tiles that can hold masked keys pay for the test
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    bool ok = key < S;
                    if (MASK) ok = ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
                    if (!ok) sc[kt][r] = NEG_BIG;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
        }
        mx = group4_max(mx);
        const float mc = mx * c2;
        float sum = 0.f, dl = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt == NKT - 1 && !last_live) continue;   // sc, dp stay exactly 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c2, -mc));  // un-normalised probability
                sc[kt][r] = e;
                sum += e;
                dl += e * dp[kt][r];
            }
        }
        sum = group4_sum(sum);
        dl = group4_sum(dl);
        const float inv = 1.0f / sum;
        dl *= inv;                                   // delta = sum_k P dP with P = e / sum (the fp32 softmax output)
        // dS = P (dP - delta) scale = e (dP - delta) (scale / sum): the per-query factor scale / sum is applied to the dQ
        // accumulators (query on the lane: one constant per lane) instead of to every score
        const float sinv = scale * inv;
        if (g == 0) { st_m[q] = (q < nq) ? mc + __builtin_amdgcn_logf(sum) : -NEG_BIG; st_d[q] = dl; }  // q < S_pad always
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NKT / 2; ++s) {
            f32x4 d0, d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0[r] = sc[2 * s][r] * (dp[2 * s][r] - dl);
                d1[r] = sc[2 * s + 1][r] * (dp[2 * s + 1][r] - dl);
            }
            const bf16x8 dsf = pack_frag(d0, d1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(t0, s, dt, lane), dsf, dq[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] *= sinv;
        store_rows16(dqbase + (size_t)min(q, S - 1) * ld, dq, 1.0f, q < S, g);
    }
    ATT_STAMP(2);
    __syncthreads();  // every wave is done reading K/V tiles; statistics are visible
    ATT_STAMP(3);

    if constexpr (PAIR) {
    static_assert(!PAIR || NW == ATT_WAVES, "the two-key-tile phase 2 deals pairs to four waves");
    // ---------------- phase 2: dV, dK — TWO 16-key tiles per wave (query on the MFMA row, key on the lane) ----------------
    // Every Q / dO fragment read from LDS (row form for S and dP, transposed form for dK and dV) feeds two MFMAs, one per
    // key tile: half the LDS traffic per MFMA of a one-tile sweep (this phase was LDS-bandwidth bound: 1 KiB per MFMA).
    // Arithmetic per element is unchanged.
    // This wave's key tiles (at most two pairs: S_pad <= 256, four waves) as MFMA row fragments, taken from the K / V images before
    // they are released — not re-read from global memory (a quarter of the kernel's fetches, and their latency at the head of a pair)
    const int nkt = (S + 15) >> 4;
    bf16x8 kfs[2][2][2], vfs[2][2][2];   // [pair of this wave][key tile][k-step]
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        if (wave + ATT_WAVES * pi < ((nkt + 1) >> 1)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kr = min((2 * (wave + ATT_WAVES * pi) + j) * 16 + i, S_pad - 1);   // rows S .. S_pad-1 are copies of row S-1 (never stored)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    kfs[pi][j][ks] = lds_row_frag(t0, kr, ks, g);
                    vfs[pi][j][ks] = lds_row_frag(t1, kr, ks, g);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave holds its fragments: the images may be overwritten
    stage_head_tile(t0, qbase, ld, S, S_pad, wave, lane);
    stage_head_tile(t1, dobase, (size_t)H, nq, S_pad, wave, lane);  // rows >= nq are clamped copies, masked below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ATT_STAMP(4);

#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int p = wave + ATT_WAVES * pi;
        if (p >= ((nkt + 1) >> 1)) break;
        bf16x8 kf[2][2], vf[2][2];
        bool key_ok[2];
        int keyv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = (2 * p + j) * 16 + i;
            keyv[j] = key;
            bool ok = key < S;
            if (MASK) ok = ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
            key_ok[j] = ok;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { kf[j][ks] = kfs[pi][j][ks]; vf[j][ks] = vfs[pi][j][ks]; }
        }
        f32x4 dv[2][4], dk[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dv[j][dt] = (f32x4){0, 0, 0, 0}; dk[j][dt] = (f32x4){0, 0, 0, 0}; }
        const bool live1 = (2 * p + 1) * 16 < S;   // the second key tile of the last pair can be all padding: skipped
        constexpr bool masked = MASK;
#pragma unroll 1
        for (int s = 0; s < (nqt + 1) / 2; ++s) {  // only query tiles that carry a gradient
            const bool q1 = 2 * s + 1 < nqt;       // the second query tile of the last pair can be past the last live tile
            bf16x8 qr[2][2], dor[2][2];
            float mr[2][4], dlr[2][4];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                if (hq == 1 && !q1) continue;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qr[hq][ks] = lds_row_frag(t0, (2 * s + hq) * 16 + i, ks, g);
                    dor[hq][ks] = lds_row_frag(t1, (2 * s + hq) * 16 + i, ks, g);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = (2 * s + hq) * 16 + 4 * g + r;
                    mr[hq][r] = st_m[qq]; dlr[hq][r] = st_d[qq];
                }
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pf[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                dsf[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (j == 1 && !live1) continue;
                f32x4 pp[2], dd[2];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    pp[hq] = (f32x4){0, 0, 0, 0};
                    dd[hq] = (f32x4){0, 0, 0, 0};
                    if (hq == 1 && !q1) continue;
                    f32x4 sv = (f32x4){0, 0, 0, 0}, dpv = (f32x4){0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[hq][ks], kf[j][ks], sv, 0, 0, 0);
                        dpv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dor[hq][ks], vf[j][ks], dpv, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // P = exp2(c2 s - m'): normalised; 0 for rows without a gradient (m' = +BIG); rounded to bf16 only inside dV's operand
                        float pr = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -mr[hq][r]));
                        if (masked && !key_ok[j]) pr = 0.f;   // padding keys (key >= S) need no test: their dK / dV rows are never stored
                        float fm = 1.0f;
                        if (DROP) {
                            const int qq = (2 * s + hq) * 16 + 4 * g + r;
                            fm = drop_one(drop_seed, (((unsigned)blockIdx.x * (unsigned)S + (unsigned)qq) << 8) + (unsigned)keyv[j], (unsigned)drop_thr16, drop_scale);
                        }
                        pp[hq][r] = pr * fm;                           // dV = (P o M/(1-p))^T dO
                        dd[hq][r] = pr * (dpv[r] * fm - dlr[hq][r]);   // dS / scale = P o (dP - delta); scale is applied to dK at the end
                    }
                }
                pf[j] = pack_frag(pp[0], pp[1]);
                dsf[j] = pack_frag(dd[0], dd[1]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 trdo = lds_tr_frag(t1, s, dt, lane);
                const bf16x8 trq = lds_tr_frag(t0, s, dt, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !live1) continue;
                    dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trdo, pf[j], dv[j][dt], 0, 0, 0);
                    dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trq, dsf[j], dk[j][dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            {
                const size_t roff = (size_t)min(keyv[j], S - 1) * ld;
                store_rows16(dqbase + H + roff, dk[j], scale, keyv[j] < S, g);
                store_rows16(dqbase + 2 * H + roff, dv[j], 1.0f, keyv[j] < S, g);
            }
        }
    }
    ATT_STAMP(5);
    } else {
    // ---------------- phase 2: per 16-key tile: dV, dK (query on the MFMA row, key on the lane) ----------------
    bf16x8 kf[2], vf[2], kfn[2], vfn[2];
#ifdef CLIBD_ATT_NO_REFETCH
    // TIMING-ONLY variant (tools/build_variant.sh, wrong results): the upper bound of removing this kernel's SECOND fetch of q, k, v and
    // dO — phase 2 takes its K / V fragments from the LDS images and sweeps the K / V images as if they were Q / dO (no re-staging)
    {
        const int kc0 = min(wave * 16 + i, IMG - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[ks] = lds_row_frag(t0, kc0, ks, g); vf[ks] = lds_row_frag(t1, kc0, ks, g); }
    }
    __syncthreads();
#else
    {
        const int kc0 = min(wave * 16 + i, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = *(const bf16x8*)(qbase + H + (size_t)kc0 * ld + 32 * ks + 8 * g);
            vf[ks] = *(const bf16x8*)(qbase + 2 * H + (size_t)kc0 * ld + 32 * ks + 8 * g);
        }
    }
    stage_head_tile(t0, qbase, ld, S, IMG, wave, lane, NW);
    stage_head_tile(t1, dobase, (size_t)H, nq, IMG, wave, lane, NW);  // rows >= nq are clamped copies, masked below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif

    const int nkt = (S + 15) >> 4;
    for (int kt = wave; kt < nkt; kt += NW) {
        const int key = kt * 16 + i;
        bool key_ok = key < S;
        if (MASK) key_ok = key_ok && key_mask[(size_t)b * S + min(key, S - 1)] != 0;
        {   // next key tile's K / V fragments fly during this tile's sweep over the queries
#ifdef CLIBD_ATT_NO_REFETCH
            const int kn = min((kt + NW) * 16 + i, IMG - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { kfn[ks] = lds_row_frag(t0, kn, ks, g); vfn[ks] = lds_row_frag(t1, kn, ks, g); }
#else
            const int kn = min((kt + NW) * 16 + i, S - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kfn[ks] = *(const bf16x8*)(qbase + H + (size_t)kn * ld + 32 * ks + 8 * g);
                vfn[ks] = *(const bf16x8*)(qbase + 2 * H + (size_t)kn * ld + 32 * ks + 8 * g);
            }
#endif
        }
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
        constexpr bool masked = MASK;
#pragma unroll 1
        for (int s = 0; s < (nqt + 1) / 2; ++s) {  // only query tiles that carry a gradient
            f32x4 pp[2], dd[2];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const int qt = 2 * s + hq;
                pp[hq] = (f32x4){0, 0, 0, 0};
                dd[hq] = (f32x4){0, 0, 0, 0};
                if (hq == 1 && qt >= nqt) continue;   // past the last live query tile
                f32x4 sv = (f32x4){0, 0, 0, 0}, dpv = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(t0, qt * 16 + i, ks, g), kf[ks], sv, 0, 0, 0);
                    dpv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(t1, qt * 16 + i, ks, g), vf[ks], dpv, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = qt * 16 + 4 * g + r;
                    const float m = st_m[qq], dl = st_d[qq];
                    float p = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -m));  // normalised; 0 for rows without a gradient (m' = +BIG); rounded to bf16 only inside dV's operand
                    if (masked && !key_ok) p = 0.f;   // padding keys need no test: their dK / dV rows are never stored
                    float fm = 1.0f;
                    if (DROP)
                        fm = drop_one(drop_seed, (((unsigned)blockIdx.x * (unsigned)S + (unsigned)qq) << 8) + (unsigned)key, (unsigned)drop_thr16, drop_scale);
                    pp[hq][r] = p * fm;                  // dV = (P o M/(1-p))^T dO
                    dd[hq][r] = p * (dpv[r] * fm - dl);  // dS / scale = P o (dP - delta), dP masked as in phase 1; scale goes on dK at the end
                }
            }
            const bf16x8 pf = pack_frag(pp[0], pp[1]);
            const bf16x8 dsf = pack_frag(dd[0], dd[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(t1, s, dt, lane), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(t0, s, dt, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
        {
            const size_t roff = (size_t)min(key, S - 1) * ld;
            store_rows16(dqbase + H + roff, dk, scale, key < S, g);
            store_rows16(dqbase + 2 * H + roff, dv, 1.0f, key < S, g);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[ks] = kfn[ks]; vf[ks] = vfn[ks]; }
    }
    }
}

// ============================================ backward, single pass ===============================================
// The two-phase kernel above evaluates every score, every exponential and every dP twice (once per query tile for dQ and the
// softmax statistics, once per key tile for dK / dV) because it derives the row statistics itself.  Given the forward's
// log-sum-exp and delta = dO . O (from the saved output and its rounding residual, store_rows16_lo), one sweep over the
// (key tile, query tile) pairs is enough:
//     P = exp2(c2 s - lse),  dS = P o (dP - delta),  dV += P^T dO,  dK += dS^T Q,  dQ += dS K.
// One workgroup of 8 waves per head, all four operand images (K, V, Q, dO: [32 NP][64] bf16 each) resident in LDS.  Key tiles
// are OWNED by waves (tile w and w + 8): dK and dV of a tile stay in that wave's accumulators for the whole head.  dQ is a sum
// over every wave's keys; instead of atomics the waves exchange dS: per block of 64 queries each wave writes the bf16 dS of its
// key tiles into a fifth LDS image laid out [key][query] — the layout of the other images, so the hardware-transposing read
// (ds_read_b64_tr_b16) hands it back as the "query on the lane, keys in the k-slots" MFMA operand — and after a barrier the
// eight waves compute that block's dQ = dS K, one (query tile, half of the head dim) each, contracting over ALL keys.
// Deterministic, no float atomics; 2 barriers per 64 queries.  MFMAs per head at S = 197: 1 790 against 2 660, exponentials and
// score arithmetic once instead of twice.  Needs nq = S, no key mask, S <= 224 (LDS: 5 images of 32 NP rows + statistics).
constexpr int ATTB_WAVES = 8;

template <bool DROP, int NPT>   // NPT: number of 32-row k-slots (7 at S = 197, 5 at S = 133) when known at compile time, 0 = run-time
__global__ __launch_bounds__(64 * ATTB_WAVES) void attention_bwd_sp_kernel(const unsigned short* __restrict__ qkv,
                                                                          const unsigned short* __restrict__ dout,
                                                                          const unsigned short* __restrict__ o_hi,
                                                                          const unsigned short* __restrict__ o_lo,
                                                                          const float* __restrict__ lse, int S, int nheads,
                                                                          unsigned short* __restrict__ dqkv, float scale, unsigned drop_seed,
                                                                          int drop_thr16, float drop_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = (S + 15) >> 4;                        // 16-row tiles of the sequence
    const int NP = NPT > 0 ? NPT : ((NT + 1) >> 1);      // 32-row k-slots
    const int R = 32 * NP;                               // rows of every image
    char* kimg = smem;
    char* vimg = kimg + R * 128;
    char* qimg = vimg + R * 128;
    char* doimg = qimg + R * 128;
    char* dsimg = doimg + R * 128;         // [key][64 queries of the current block] bf16
    float* st_m = (float*)(dsimg + R * 128);
    float* st_d = st_m + R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / nheads, h = blockIdx.x % nheads;
    const int H = nheads * DH;
    const size_t ld = (size_t)3 * H;
    const unsigned short* qbase = qkv + (size_t)b * S * ld + h * DH;
    const size_t obase = (size_t)b * S * H + h * DH;
    const unsigned short* dobase = dout + obase;
    unsigned short* dqbase = dqkv + (size_t)b * S * ld + h * DH;
    const int g = lane >> 4, i = lane & 15;
    const float c2 = scale * 1.4426950408889634f;

    SP_STAMP(0);
    // delta[q] = dO[q] . (o_hi[q] + o_lo[q]) and the row statistic; query rows >= S get lse = +BIG, i.e. P = 0 exactly.
    // Eight lanes per row, 64 rows per pass.  ALL passes' rows are requested before the operand images: the in-order vmcnt queue
    // then lets the arithmetic below start as soon as these loads are back, with the LDS-DMA of the images still in flight
    // (the first version issued each pass's loads after the previous pass's reduction: four global round trips behind the
    // DMA queue = 41 % of the workgroup's lifetime, tools/att_sp_stamps.py; now 28 %).
    constexpr int DPASS = 4;                 // 4 x 64 = 256 rows >= R
    bf16x8 d8[DPASS], h8[DPASS], l8[DPASS];
    float lse_r[DPASS];
    {
        const int sub = threadIdx.x & 7;
#pragma unroll
        for (int ps = 0; ps < DPASS; ++ps) {
            const int row = min(64 * ps + (int)(threadIdx.x >> 3), S - 1);
            const size_t off = (size_t)row * H + 8 * sub;
            d8[ps] = *(const bf16x8*)(dobase + off);
            h8[ps] = *(const bf16x8*)(o_hi + obase + off);
            l8[ps] = *(const bf16x8*)(o_lo + obase + off);
            lse_r[ps] = lse[(size_t)blockIdx.x * S + row];
        }
    }
    stage_head_tile(kimg, qbase + H, ld, S, R, wave, lane, ATTB_WAVES);
    stage_head_tile(vimg, qbase + 2 * H, ld, S, R, wave, lane, ATTB_WAVES);
    stage_head_tile(qimg, qbase, ld, S, R, wave, lane, ATTB_WAVES);
    stage_head_tile(doimg, dobase, (size_t)H, S, R, wave, lane, ATTB_WAVES);
#pragma unroll
    for (int ps = 0; ps < DPASS; ++ps) {
        const int row = 64 * ps + (int)(threadIdx.x >> 3), sub = threadIdx.x & 7;
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            part += bf2f((unsigned short)d8[ps][e]) * (bf2f((unsigned short)h8[ps][e]) + bf2f((unsigned short)l8[ps][e]));
        part += dpp_mov<DPP_QUAD_XOR1>(part);
        part += dpp_mov<DPP_QUAD_XOR2>(part);
        part += dpp_mov<DPP_ROW_HALF_MIRROR>(part);   // the 8 lanes of a row
        if (sub == 0 && row < R) {
            st_d[row] = row < S ? part : 0.f;
            st_m[row] = row < S ? lse_r[ps] : -NEG_BIG;
        }
    }
    // the key rows of the exchange image that no wave owns (the padding tile of an odd tile count) must read as zeros
    for (int e = threadIdx.x; e < (R - 16 * NT) * 8; e += 64 * ATTB_WAVES) *(uint4*)(dsimg + 16 * NT * 128 + e * 16) = make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    SP_STAMP(1);
    // ---- this wave's key tiles (0, 1 or 2 of them) as MFMA row fragments, for the whole head
    const int nown = (wave < NT ? 1 : 0) + (wave + ATTB_WAVES < NT ? 1 : 0);
    bf16x8 kf[2][2], vf[2][2];
    int keyv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kt = min(wave + ATTB_WAVES * j, 2 * NP - 1);
        keyv[j] = (wave + ATTB_WAVES * j) * 16 + i;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[j][ks] = lds_row_frag(kimg, kt * 16 + i, ks, g);
            vf[j][ks] = lds_row_frag(vimg, kt * 16 + i, ks, g);
        }
    }
    f32x4 dv[2][4], dk[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[j][dt] = (f32x4){0, 0, 0, 0}; dk[j][dt] = (f32x4){0, 0, 0, 0}; }

    const int nqb = (NT + 3) >> 2;   // blocks of 64 queries
#pragma unroll 1
    for (int it = 0; it < nqb; ++it) {
        // ------------ sweep: this wave's key tiles x the block's two query pairs
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int s = 2 * it + p;          // query pair = k-slot of the Q / dO images
            if (s < NP) {
                const bool q1 = 2 * s + 1 < NT;
                bf16x8 qr[2][2], dor[2][2];
                float mr[2][4], dlr[2][4];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        qr[hq][ks] = lds_row_frag(qimg, (2 * s + hq) * 16 + i, ks, g);
                        dor[hq][ks] = lds_row_frag(doimg, (2 * s + hq) * 16 + i, ks, g);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qq = (2 * s + hq) * 16 + 4 * g + r;
                        mr[hq][r] = st_m[qq];
                        dlr[hq][r] = st_d[qq];
                    }
                }
                bf16x8 pf[2], dsf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    pf[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    dsf[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (j >= nown) continue;
                    const bool key_live = keyv[j] < S;   // padding keys of the last live tile: no probability, no dS (dQ sums over keys)
                    f32x4 pp[2], dd[2];
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        pp[hq] = (f32x4){0, 0, 0, 0};
                        dd[hq] = (f32x4){0, 0, 0, 0};
                        if (hq == 1 && !q1) continue;
                        f32x4 sv = (f32x4){0, 0, 0, 0}, dpv = (f32x4){0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[hq][ks], kf[j][ks], sv, 0, 0, 0);
                            dpv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dor[hq][ks], vf[j][ks], dpv, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float pr = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -mr[hq][r]));   // normalised; 0 for query rows >= S
                            if (!key_live) pr = 0.f;
                            float fm = 1.0f;
                            if (DROP) {
                                const int qq = (2 * s + hq) * 16 + 4 * g + r;
                                fm = drop_one(drop_seed, (((unsigned)blockIdx.x * (unsigned)S + (unsigned)qq) << 8) + (unsigned)keyv[j], (unsigned)drop_thr16, drop_scale);
                            }
                            pp[hq][r] = pr * fm;                           // dV = (P o M/(1-p))^T dO
                            dd[hq][r] = pr * (dpv[r] * fm - dlr[hq][r]);   // dS / scale
                        }
                    }
                    pf[j] = pack_frag(pp[0], pp[1]);
                    dsf[j] = pack_frag(dd[0], dd[1]);
                    // exchange image: row = key, columns 32 p + 16 hq + 4 g .. + 3 of the block (8 bytes per query tile)
                    const uint4 w4 = __builtin_bit_cast(uint4, dsf[j]);
                    const int krow = (wave + ATTB_WAVES * j) * 16 + i;
                    *(uint2*)(dsimg + tile_off(krow, 4 * p + (g >> 1)) + 8 * (g & 1)) = make_uint2(w4.x, w4.y);
                    *(uint2*)(dsimg + tile_off(krow, 4 * p + 2 + (g >> 1)) + 8 * (g & 1)) = make_uint2(w4.z, w4.w);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 trdo = lds_tr_frag(doimg, s, dt, lane);
                    const bf16x8 trq = lds_tr_frag(qimg, s, dt, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j >= nown) continue;
                        dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trdo, pf[j], dv[j][dt], 0, 0, 0);
                        dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trq, dsf[j], dk[j][dt], 0, 0, 0);
                    }
                }
            } else {
                // a query pair past the end (the block's second pair when NP is odd): its columns of the exchange image are read by no one
            }
        }
        if (it < 4) SP_STAMP(2 + 3 * it);
        __syncthreads();   // the block's dS is complete
        if (it < 4) SP_STAMP(3 + 3 * it);
        // ------------ dQ of the block: wave -> (query tile 4 it + (wave >> 1), head-dim half wave & 1), all keys
        {
            const int qtl = wave >> 1, pr2 = wave & 1;
            const int qt = 4 * it + qtl;
            if (qt < NT) {
                f32x4 dq0 = (f32x4){0, 0, 0, 0}, dq1 = (f32x4){0, 0, 0, 0};
                if constexpr (NPT > 0) {
                    // every transposing read of the block's dS column and of the two K^T fragments is independent of the MFMAs:
                    // fully unrolled, the reads of the later k-slots fly under the earlier products
#pragma unroll
                    for (int sp = 0; sp < NPT; ++sp) {
                        const bf16x8 dsb = lds_tr_frag(dsimg, sp, qtl, lane);
                        dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(kimg, sp, 2 * pr2, lane), dsb, dq0, 0, 0, 0);
                        dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(kimg, sp, 2 * pr2 + 1, lane), dsb, dq1, 0, 0, 0);
                    }
                } else {
#pragma unroll 1
                    for (int sp = 0; sp < NP; ++sp) {
                        const bf16x8 dsb = lds_tr_frag(dsimg, sp, qtl, lane);
                        dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(kimg, sp, 2 * pr2, lane), dsb, dq0, 0, 0, 0);
                        dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(kimg, sp, 2 * pr2 + 1, lane), dsb, dq1, 0, 0, 0);
                    }
                }
                const int q = qt * 16 + i;
                store_rows16_pair(dqbase + (size_t)min(q, S - 1) * ld, dq0, dq1, scale, q < S, g, pr2);
            }
        }
        if (it < 4) SP_STAMP(4 + 3 * it);
        __syncthreads();   // before the next block's dS overwrites the exchange image
    }
    SP_STAMP(14);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j >= nown) continue;
        const size_t roff = (size_t)min(keyv[j], S - 1) * ld;
        store_rows16(dqbase + H + roff, dk[j], scale, keyv[j] < S, g);
        store_rows16(dqbase + 2 * H + roff, dv[j], 1.0f, keyv[j] < S, g);
    }
    SP_STAMP(15);
}

static int att_check(const void* qkv, int B, int S, int nheads, const char* who) {
    if (!qkv) return set_error(CLIBD_EINVAL, "attention: null pointer");
    if (B <= 0 || S <= 0 || nheads <= 0) return set_error(CLIBD_EINVAL, "attention: non-positive shape");
    if (S > 16 * MAX_KT) return set_error(CLIBD_EINVAL, "attention: S must be <= 256");
    if ((long long)B * nheads > 0x7fffffffLL) return set_error(CLIBD_EINVAL, "attention: grid too large");
    if (!aligned16(qkv)) return set_error(CLIBD_EINVAL, "attention: alignment");
    (void)who;
    return 0;
}

}  // namespace clibd

using namespace clibd;

#ifndef CLIBD_ATT_BWD_S256_DROP_ONE_TILE
#define CLIBD_ATT_BWD_S256_DROP_ONE_TILE 0
#endif
#ifndef CLIBD_ATT_FWD_PERSISTENT_SPILLING
#define CLIBD_ATT_FWD_PERSISTENT_SPILLING 1
#endif
#define ATT_DISPATCH(NKT_EXPR, MACRO) \
    switch (NKT_EXPR) {               \
        case 2: MACRO(2); break;      \
        case 4: MACRO(4); break;      \
        case 6: MACRO(6); break;      \
        case 8: MACRO(8); break;      \
        case 10: MACRO(10); break;    \
        case 12: MACRO(12); break;    \
        case 14: MACRO(14); break;    \
        default: MACRO(16); break;    \
    }

static int attention_fwd_impl(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out,
                              int nq, int out_seq, uint32_t drop_seed, int drop_thr16, float drop_scale, float out_fp8_scale, void* stream,
                              float* lse = nullptr, void* o_lo = nullptr) {
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "attention_fwd: bad dropout threshold");
    if (drop_thr16 > 0 && (unsigned long long)B * nheads * S * 256ull >= (1ull << 32)) return set_error(CLIBD_EINVAL, "attention_fwd: dropout index overflow");
    if (int e = att_check(qkv, B, S, nheads, "fwd")) return e;
    if (!out) return set_error(CLIBD_EINVAL, "attention_fwd: null out");
    if (nq < 1 || nq > S || out_seq < nq) return set_error(CLIBD_EINVAL, "attention_fwd: need 1 <= nq <= S and out_seq >= nq");
    const int nkt = 2 * ((S + 31) / 32);
    const size_t lds = (size_t)2 * nkt * 16 * 128;
    const float scale = 0.125f;  // 1/sqrt(64)
    hipStream_t st = (hipStream_t)stream;
    static const int num_cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        return n;
    }();
    // many heads of a long sequence: the persistent kernel (K / V of the next head land under the current head's arithmetic)
    const int total = B * nheads;
    const size_t ldsp = (size_t)2 * lds;
    static const bool three_waves = [] { const char* e = getenv("CLIBD_ATTN_FWD_WAVES"); return !(e && e[0] == '4'); }();
    const size_t lds3 = (size_t)2 * 144 * 128 + 16 * 128;   // three-wave form: two 144-row images + the zeroed pad
#define LAUNCH_M(N, MSK, DRP)                                                                                     \
    do {                                                                                                          \
        /* The dropout forms of the persistent kernel that spill 2-10 registers at its 128-register cap (dropout with a key mask, dropout at sixteen tiles) were measured   */ \
        /* against the spill-free per-head kernel in round 6 (knob = 0): 207 / 242 us persistent against 265-269 / 299 us per head at S = 220 / 250: the persistent forms stay */ \
        constexpr bool PERSIST_OK = CLIBD_ATT_FWD_PERSISTENT_SPILLING || !(DRP && (MSK || N >= 16));               \
        if (PERSIST_OK && N >= 12 && total >= 2 * num_cus) {   /* S > 160: at S = 133 only 9 of the 16 waves have a tile and the per-head kernel wins */ \
            constexpr int NP = (PERSIST_OK && N >= 12) ? N : 12;   /* (only the long-sequence, spill-free forms are instantiated) */             \
            hipFuncSetAttribute((const void*)attention_fwd_persistent_kernel<NP, MSK, DRP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp); \
            hipLaunchKernelGGL((attention_fwd_persistent_kernel<NP, MSK, DRP>), dim3(num_cus), dim3(64 * ATTP_WAVES), ldsp, st, \
                               (const unsigned short*)qkv, S, nheads, total, (const int*)key_mask, (unsigned short*)out, scale, nq, out_seq, \
                               drop_seed, drop_thr16, drop_scale, out_fp8_scale, lse, (unsigned short*)o_lo);      \
        } else if (N == 10 && S <= 144 && S > 128 && nq > 128 && three_waves) {   /* nine query tiles: 3 + 3 + 3 on three waves */ \
            hipFuncSetAttribute((const void*)attention_fwd_kernel<10, false, MSK, DRP, 3, 144>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
            hipLaunchKernelGGL((attention_fwd_kernel<10, false, MSK, DRP, 3, 144>), dim3(B * nheads), dim3(192), lds3, st,   \
                               (const unsigned short*)qkv, S, nheads, (const int*)key_mask, (unsigned short*)out, scale, nq, out_seq, \
                               drop_seed, drop_thr16, drop_scale, out_fp8_scale, lse, (unsigned short*)o_lo);      \
        } else {                                                                                                  \
        hipFuncSetAttribute((const void*)attention_fwd_kernel<N, (N >= 10), MSK, DRP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attention_fwd_kernel<N, (N >= 10), MSK, DRP>), dim3(B * nheads), dim3(ATT_THREADS), lds, st,       \
                           (const unsigned short*)qkv, S, nheads, (const int*)key_mask, (unsigned short*)out, scale, nq, out_seq, \
                           drop_seed, drop_thr16, drop_scale, out_fp8_scale, lse, (unsigned short*)o_lo);         \
        }                                                                                                         \
    } while (0)
#define LAUNCH(N)                                                        \
    do {                                                                 \
        if (key_mask != nullptr && drop_thr16 > 0) LAUNCH_M(N, true, true);   \
        else if (key_mask != nullptr) LAUNCH_M(N, true, false);          \
        else if (drop_thr16 > 0) LAUNCH_M(N, false, true);               \
        else LAUNCH_M(N, false, false);                                  \
    } while (0)
    ATT_DISPATCH(nkt, LAUNCH)
#undef LAUNCH
#undef LAUNCH_M
    return check_launch("attention_fwd");
}

// Training forward for the single-pass backward: the same kernels, which also write lse [B * nheads, S] (fp32, log2 domain) and
// o_lo (bf16 [B * S, H]: the rounding residual of `out`).  nq = S only (the backward that consumes them needs every query row).
extern "C" int clibd_attention_fwd_save(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out, uint32_t drop_seed,
                                        int drop_thr16, float drop_scale, float* lse, void* o_lo, void* stream) {
    if (!lse || !o_lo) return set_error(CLIBD_EINVAL, "attention_fwd_save: null lse / o_lo");
    if (!aligned16(o_lo) || ((uintptr_t)lse & 3)) return set_error(CLIBD_EINVAL, "attention_fwd_save: alignment");
    return attention_fwd_impl(qkv, B, S, nheads, key_mask, out, S, S, drop_seed, drop_thr16, drop_scale, 0.f, stream, lse, o_lo);
}

extern "C" int clibd_attention_bwd_sp(const void* qkv, const void* dout, const void* out, const void* o_lo, const float* lse, int B, int S,
                                      int nheads, void* dqkv, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "attention_bwd_sp: bad dropout threshold");
    if (drop_thr16 > 0 && (unsigned long long)B * nheads * S * 256ull >= (1ull << 32)) return set_error(CLIBD_EINVAL, "attention_bwd_sp: dropout index overflow");
    if (int e = att_check(qkv, B, S, nheads, "bwd_sp")) return e;
    if (!dout || !dqkv || !out || !o_lo || !lse) return set_error(CLIBD_EINVAL, "attention_bwd_sp: null pointer");
    if (S > 224) return set_error(CLIBD_EINVAL, "attention_bwd_sp: S must be <= 224 (five LDS images); use clibd_attention_bwd");
    if (!aligned16(dout) || !aligned16(dqkv) || !aligned16(out) || !aligned16(o_lo)) return set_error(CLIBD_EINVAL, "attention_bwd_sp: alignment");
    const int np = ((S + 15) / 16 + 1) / 2;
    const size_t lds = (size_t)5 * 32 * np * 128 + (size_t)2 * 32 * np * sizeof(float);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_SP(DRP, NPT_)                                                                                                   \
    do {                                                                                                                      \
        hipFuncSetAttribute((const void*)attention_bwd_sp_kernel<DRP, NPT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attention_bwd_sp_kernel<DRP, NPT_>), dim3(B * nheads), dim3(64 * ATTB_WAVES), lds, st, (const unsigned short*)qkv, \
                           (const unsigned short*)dout, (const unsigned short*)out, (const unsigned short*)o_lo, lse, S, nheads, \
                           (unsigned short*)dqkv, scale, drop_seed, drop_thr16, drop_scale);                                  \
    } while (0)
    // the two sequence lengths of the training step get their k-slot count at compile time (unrolled dQ stage); the rest share a run-time form
    if (drop_thr16 > 0) {
        if (np == 5) LAUNCH_SP(true, 5);
        else if (np == 7) LAUNCH_SP(true, 7);
        else LAUNCH_SP(true, 0);
    } else {
        if (np == 5) LAUNCH_SP(false, 5);
        else if (np == 7) LAUNCH_SP(false, 7);
        else LAUNCH_SP(false, 0);
    }
#undef LAUNCH_SP
    return check_launch("attention_bwd_sp");
}

extern "C" int clibd_attention_fwd_drop(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out,
                                        int nq, int out_seq, uint32_t drop_seed, int drop_thr16, float drop_scale, void* stream) {
    return attention_fwd_impl(qkv, B, S, nheads, key_mask, out, nq, out_seq, drop_seed, drop_thr16, drop_scale, 0.f, stream);
}

extern "C" int clibd_attention_fwd_fp8(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out_fp8,
                                       int nq, int out_seq, uint32_t drop_seed, int drop_thr16, float drop_scale, float out_fp8_scale,
                                       void* stream) {
    if (!(out_fp8_scale > 0.f)) return set_error(CLIBD_EINVAL, "attention_fwd_fp8: scale must be positive");
    return attention_fwd_impl(qkv, B, S, nheads, key_mask, out_fp8, nq, out_seq, drop_seed, drop_thr16, drop_scale, out_fp8_scale, stream);
}

extern "C" int clibd_attention_fwd(const void* qkv, int B, int S, int nheads, const int32_t* key_mask, void* out,
                                   int nq, int out_seq, void* stream) {
    return clibd_attention_fwd_drop(qkv, B, S, nheads, key_mask, out, nq, out_seq, 0u, 0, 1.0f, stream);
}

extern "C" int clibd_attention_bwd_drop(const void* qkv, const void* dout, int B, int S, int nheads, const int32_t* key_mask,
                                        void* dqkv, int nq, int dout_seq, uint32_t drop_seed, int drop_thr16, float drop_scale,
                                        void* stream) {
    if (drop_thr16 < 0 || drop_thr16 > 65535) return set_error(CLIBD_EINVAL, "attention_bwd: bad dropout threshold");
    if (drop_thr16 > 0 && (unsigned long long)B * nheads * S * 256ull >= (1ull << 32)) return set_error(CLIBD_EINVAL, "attention_bwd: dropout index overflow");
    if (int e = att_check(qkv, B, S, nheads, "bwd")) return e;
    if (!dout || !dqkv) return set_error(CLIBD_EINVAL, "attention_bwd: null pointer");
    if (nq < 1 || nq > S || dout_seq < nq) return set_error(CLIBD_EINVAL, "attention_bwd: need 1 <= nq <= S and dout_seq >= nq");
    if (!aligned16(dout) || !aligned16(dqkv)) return set_error(CLIBD_EINVAL, "attention_bwd: alignment");
    const int nkt = 2 * ((S + 31) / 32);
    const size_t lds = (size_t)2 * nkt * 16 * 128 + (size_t)2 * nkt * 16 * sizeof(float);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    // S in (128, 144] — the DNA tower's 133 tokens: nine tiles on THREE waves per workgroup, four workgroups per CU (see the kernel)
    static const bool three_waves = [] { const char* e = getenv("CLIBD_ATTN_BWD_WAVES"); return !(e && e[0] == '4'); }();
    if (nkt == 10 && S <= 144 && key_mask == nullptr && three_waves) {
        constexpr int IMG3 = 144;
        const size_t lds3 = (size_t)2 * IMG3 * 128 + 16 * 128 + (size_t)2 * 160 * sizeof(float);
#define LAUNCH_3(DRP)                                                                                             \
    do {                                                                                                          \
        hipFuncSetAttribute((const void*)attention_bwd_kernel<10, false, false, DRP, 3, IMG3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
        hipLaunchKernelGGL((attention_bwd_kernel<10, false, false, DRP, 3, IMG3>), dim3(B * nheads), dim3(192), lds3, st, \
                           (const unsigned short*)qkv, (const unsigned short*)dout, S, nheads, (const int*)nullptr,   \
                           (unsigned short*)dqkv, scale, nq, dout_seq, drop_seed, drop_thr16, drop_scale);        \
    } while (0)
        if (drop_thr16 > 0) LAUNCH_3(true);
        else LAUNCH_3(false);
#undef LAUNCH_3
        return check_launch("attention_bwd");
    }
#define LAUNCH_M(N, MSK, DRP)                                                                                     \
    do {                                                                                                          \
        /* PAIR (two key tiles per wave in phase 2) for the long forms.  Sixteen tiles with dropout and no mask spill 10 registers at 256; the one-tile form (knob = 1) spills 4 and runs the same 384 us at S = 250: PAIR kept */ \
        constexpr bool PR = (N >= 12) && !(CLIBD_ATT_BWD_S256_DROP_ONE_TILE && N == 16 && DRP && !MSK);             \
        hipFuncSetAttribute((const void*)attention_bwd_kernel<N, PR, MSK, DRP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attention_bwd_kernel<N, PR, MSK, DRP>), dim3(B * nheads), dim3(ATT_THREADS), lds, st,       \
                           (const unsigned short*)qkv, (const unsigned short*)dout, S, nheads, (const int*)key_mask, \
                           (unsigned short*)dqkv, scale, nq, dout_seq, drop_seed, drop_thr16, drop_scale);        \
    } while (0)
#define LAUNCH(N)                                                        \
    do {                                                                 \
        if (key_mask != nullptr && drop_thr16 > 0) LAUNCH_M(N, true, true);   \
        else if (key_mask != nullptr) LAUNCH_M(N, true, false);          \
        else if (drop_thr16 > 0) LAUNCH_M(N, false, true);               \
        else LAUNCH_M(N, false, false);                                  \
    } while (0)
    ATT_DISPATCH(nkt, LAUNCH)
#undef LAUNCH
#undef LAUNCH_M
    return check_launch("attention_bwd");
}

extern "C" int clibd_attention_bwd(const void* qkv, const void* dout, int B, int S, int nheads, const int32_t* key_mask,
                                   void* dqkv, int nq, int dout_seq, void* stream) {
    return clibd_attention_bwd_drop(qkv, dout, B, S, nheads, key_mask, dqkv, nq, dout_seq, 0u, 0, 1.0f, stream);
}

#ifdef CLIBD_GEMM_DIAG
extern "C" void clibd_debug_set_att_stamps(void* device_buffer) {
    long long* p = (long long*)device_buffer;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(clibd::g_att_stamps), &p, sizeof(p));
}
#endif
