// Shared pieces of the bf16 NT GEMM kernels (gemm.hip: 128x128 tile; gemm256.hip: 256x256 8-phase tile).
#pragma once
#include "common.h"
#include "../../include/clibd_hip.h"

namespace clibd {

struct GemmParams {
    const unsigned short* A;
    const unsigned short* W;
    int M, N, K, lda, ldw;
    int tiles_m, tiles_n, ktiles_per_split;
    clibd_gemm_epilogue ep;
    // gemm256 split-K with a partials workspace (weight gradients: few output tiles, very long contraction): work item =
    // (output tile, split); split s covers K-tiles [s*nk_split, min((s+1)*nk_split, K/64)) and stores its fp32 partial tile
    // to ep.out_f32 + s*split_stride.  splits == 1: the ordinary GEMM.
    int splits, nk_split;
    long long split_stride;
    int band;  // gemm256 tile order: ids run m-fastest inside bands of `band` m-tiles (tile_coords)
    // 128x128 kernel only: K-tiles [hole_kt, hole_kt + hole_nkt) of both operands are skipped (a column segment of A that
    // meets all-zero weights: the k segment of dqkv in the adapters' dt projection).  hole_nkt == 0: none.
    int hole_kt, hole_nkt;
    // gemm256 fp8 mode (clibd_gemm_fp8_nt): A / W are OCP e4m3 bytes, K / lda / ldw count bytes; acc * col_scale[n] dequantises.
    // out_fp8_scale > 0 (fc1 form, EPI_GELU_SAVE): ep.out_bf16 receives fp8(gelu(x) * out_fp8_scale), ld_out_bf16 in bytes.
    int fp8;
    const float* col_scale;
    float out_fp8_scale;
    // 8-bit dgrad forms (clibd_gemm_fp8_dgrad_nt): fp32 [M], the reciprocal of the per-row scale the producer of A applied
    const float* a_row_dequant;
    // 8-bit dgrad under full fine-tune (round 6): the MUL_AUX forms also write the DE-SCALED value as bf16 (the weight gradient's operand)
    unsigned short* dual_bf16;
    int ld_dual;
    // Stream-K tail (round 6, gemm256 SK instantiations; clibd_gemm_bf16_nt_ws): the tiles of the LAST, partial round of the persistent grid are cut
    // into sk_parts K-slices, one work item each.  Work items sk_first .. sk_first + sk_tail * sk_parts - 1 are those slices, non-owners first
    // (slot j % sk_tail, part 1 + j / sk_tail), owners (part 0) last: a workgroup's id is also its dispatch order, so an owner's partners are
    // always dispatched before it.  Part q covers K-tiles [q * sk_nk_part, min((q + 1) * sk_nk_part, K / 64)); parts >= 1 store their fp32 partial
    // tile to sk_ws[((q - 1) * sk_tail + slot) * 65536] and add 1 to sk_flags[slot]; part 0 waits for sk_parts - 1, adds the partials in part
    // order (fixed order: deterministic), clears the flag and runs the tile's epilogue.  sk_parts == 0: off.
    int sk_parts, sk_tail, sk_first, sk_nk_part;
    float* sk_ws;
    unsigned* sk_flags;
};

// Host-side plan of the stream-K tail: (parts, tail tiles, first tail item, K-tiles per part); parts == 0 when the launch has no use for it
// (no partial round, a partial round that is more than half full, or K too short for the slices to pay for their partial stores).
struct SkPlan { int parts, tail, first, nk_part; };
inline SkPlan plan_stream_k_tail(long long tiles, int num_cus, int nk) {
    SkPlan s{0, 0, 0, 0};
    if (tiles <= 0 || num_cus <= 0 || nk < 24 || (nk & 1)) return s;     // K >= 1536: below that a slice's partial store costs what it saves
    const long long full = (tiles / num_cus) * num_cus;
    const int tail = (int)(tiles - full);
    if (tail == 0 || 2 * tail > num_cus) return s;
    int P = num_cus / tail;
    if (P > 4) P = 4;                                                    // (bounds the owner's merge and the workspace: tail * (P - 1) * 256 KiB <= 48 MiB)
    for (; P >= 2; --P) {
        int nkp = (nk + P - 1) / P;
        nkp += nkp & 1;
        if (nkp < 4) nkp = 4;
        const int parts = (nk + nkp - 1) / nkp;
        const int last = nk - (parts - 1) * nkp;
        if (parts >= 2 && last >= 4 && (long long)parts * tail <= num_cus) { s.parts = parts; s.tail = tail; s.first = (int)full; s.nk_part = nkp; return s; }
    }
    return s;
}

// Inside a wave's 64 output columns, MFMA n-tile t (0..3), MFMA row i (0..15) carries tile-local column
// 16*(i>>2) + 4*t + (i&3): lane group g = lane>>4 then owns the 16 CONTIGUOUS columns 16g .. 16g+15 (e = 4t + reg).
__device__ __forceinline__ int w_col_of(int t, int i) { return 16 * (i >> 2) + 4 * t + (i & 3); }

// LDS tile addressing: 128-byte rows (64 bf16), 16-byte chunk index XOR (row & 7): conflict-free ds_read_b128 fragments
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// XCD-aware, L2-friendly tile order: blocks sharing an XCD (id % 8) walk a contiguous range of tile ids; ids run
// m-fastest inside bands of `band` m-tiles, so a band's A panels stay hot while its W panels stream through.
// band < 0 (the product's order since round 5): W-stationary — n-groups of -band column tiles; inside a group m runs over ALL row tiles with the group's column
// tiles fastest, so a group's W panels (six: 2.4 MB at K = 768) stay in the XCD's L2 while every A panel streams past once per group: 15-25 % fewer fabric
// reads than the m-bands (round 3, profiles/r03_exp_tile_order.log: at equal time on isolated launches); inside the two-stream step, where both towers share the
// fabric under the power cap, that is worth 0.8 % of the step (profiles/r05_exp_gemm_ws_in_step.log).
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int band, int& tile_m, int& tile_n) {
    const int ntiles = tiles_m * tiles_n;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (band < 0) {   // W-stationary order (round 5): n-groups of -band column tiles; inside a group m runs over ALL row tiles, the group's column tiles fastest
        const int G = -band;
        const int g = bid / (tiles_m * G);
        const int rem = bid - g * tiles_m * G;
        const int gw = min(G, tiles_n - g * G);
        tile_m = rem / gw;
        tile_n = g * G + rem - tile_m * gw;
        return;
    }
    const int band_id = bid / (band * tiles_n);
    const int band_m0 = band_id * band;
    const int band_h = min(band, tiles_m - band_m0);
    const int in_band = bid - band_id * band * tiles_n;
    tile_m = band_m0 + in_band % band_h;
    tile_n = in_band / band_h;
}

__device__ __forceinline__ void load_bias16(const clibd_gemm_epilogue& ep, int nb, bool on, float bias[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) bias[e] = 0.f;
    if (ep.bias != nullptr && on) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *(const f32x4*)(ep.bias + nb + 4 * q);
            bias[4 * q + 0] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
        }
    }
}

// One output row m, 16 contiguous columns nb..nb+15 held by this lane: v = acc + bias already applied by the caller.
__device__ __forceinline__ void store_row16(const clibd_gemm_epilogue& ep, int m, int nb, float v[16]) {
    if (ep.split_k > 1) {
        float* o = ep.out_f32 + (size_t)m * ep.ld_out_f32 + nb;
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(o + e, v[e]);
        return;
    }
    if (ep.drop_thr16 > 0) {
        const unsigned base = (unsigned)m * (unsigned)ep.drop_ld + (unsigned)nb;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            float f0, f1;
            drop_pair(ep.drop_seed, base + e, (unsigned)ep.drop_thr16, ep.drop_scale, f0, f1);
            v[e] *= f0;
            v[e + 1] *= f1;
        }
    }
    if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_E12) {
        float dg[16];
        gelu_and_grad_rows<16>(v, dg);
        gelu12_row8* o = (gelu12_row8*)((unsigned char*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + (nb >> 1) * 3);   // 1.5 bytes per column
        o[0] = gelu12_pack8(dg);
        o[1] = gelu12_pack8(dg + 8);
    } else if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_U8) {
        float dg[16];
        gelu_and_grad_rows<16>(v, dg);
        uint4 c;
        c.x = geluq_pack4(dg[0], dg[1], dg[2], dg[3]);   c.y = geluq_pack4(dg[4], dg[5], dg[6], dg[7]);
        c.z = geluq_pack4(dg[8], dg[9], dg[10], dg[11]); c.w = geluq_pack4(dg[12], dg[13], dg[14], dg[15]);
        store16_stream((unsigned char*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb, c);
    } else if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD) {
        float dg[16];
        gelu_and_grad_rows<16>(v, dg);
        uint4 lo, hi;
        lo.x = pack2bf(dg[0], dg[1]);   lo.y = pack2bf(dg[2], dg[3]);   lo.z = pack2bf(dg[4], dg[5]);   lo.w = pack2bf(dg[6], dg[7]);
        hi.x = pack2bf(dg[8], dg[9]);   hi.y = pack2bf(dg[10], dg[11]); hi.z = pack2bf(dg[12], dg[13]); hi.w = pack2bf(dg[14], dg[15]);
        uint4* o = (uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb);
        store16_stream(o, lo); store16_stream(o + 1, hi);
    } else if (ep.out_pre_bf16 != nullptr) {
        uint4 lo, hi;
        lo.x = pack2bf(v[0], v[1]);   lo.y = pack2bf(v[2], v[3]);   lo.z = pack2bf(v[4], v[5]);   lo.w = pack2bf(v[6], v[7]);
        hi.x = pack2bf(v[8], v[9]);   hi.y = pack2bf(v[10], v[11]); hi.z = pack2bf(v[12], v[13]); hi.w = pack2bf(v[14], v[15]);
        uint4* o = (uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb);
        store16_stream(o, lo); store16_stream(o + 1, hi);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = bfround(v[e]);
    }
    if (ep.act == CLIBD_ACT_GELU) {
        gelu_rows<16>(v);
    } else if (ep.act == CLIBD_ACT_GELU_GRAD) {
        const uint4* ax = (const uint4*)((const unsigned short*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        const uint4 x0 = ax[0], x1 = ax[1];
        const unsigned xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[2 * e] *= gelu_grad_f(bf2f((unsigned short)(xs[e] & 0xffffu)));
            v[2 * e + 1] *= gelu_grad_f(bf2f((unsigned short)(xs[e] >> 16)));
        }
    }
    if (ep.act == CLIBD_ACT_MUL_AUX_E12) {
        const gelu12_row8* ax = (const gelu12_row8*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + (nb >> 1) * 3);
        const gelu12_row8 a0 = ax[0], a1 = ax[1];
        float d[16];
        gelu12_unpack8(a0.w0, a0.w1, a0.w2, d);
        gelu12_unpack8(a1.w0, a1.w1, a1.w2, d + 8);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] *= d[e];
    }
    if (ep.act == CLIBD_ACT_MUL_AUX_U8) {
        const uint4 c = *(const uint4*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        const unsigned cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float d[4];
            geluq_unpack4(cs[q], d);
            v[4 * q] *= d[0]; v[4 * q + 1] *= d[1]; v[4 * q + 2] *= d[2]; v[4 * q + 3] *= d[3];
        }
    }
    if (ep.act == CLIBD_ACT_MUL_AUX || ep.act == CLIBD_ACT_ADD_AUX) {
        const uint4* ax = (const uint4*)((const unsigned short*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        const uint4 x0 = ax[0], x1 = ax[1];
        const unsigned xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a0 = bf2f((unsigned short)(xs[e] & 0xffffu)), a1 = bf2f((unsigned short)(xs[e] >> 16));
            if (ep.act == CLIBD_ACT_MUL_AUX) { v[2 * e] *= a0; v[2 * e + 1] *= a1; }
            else { v[2 * e] += a0; v[2 * e + 1] += a1; }
        }
    }
    if (ep.residual_f32 != nullptr) {
        const f32x4* rs = (const f32x4*)(ep.residual_f32 + (size_t)m * ep.ld_res + nb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 r4 = rs[q];
            v[4 * q + 0] += r4[0]; v[4 * q + 1] += r4[1]; v[4 * q + 2] += r4[2]; v[4 * q + 3] += r4[3];
        }
    }
    if (ep.out_f32 != nullptr) {
        f32x4* o = (f32x4*)(ep.out_f32 + (size_t)m * ep.ld_out_f32 + nb);
#pragma unroll
        for (int q = 0; q < 4; ++q) store16_stream(o + q, __builtin_bit_cast(uint4, (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}));
    }
    if (ep.out_bf16 != nullptr) {
        uint4 lo, hi;
        lo.x = pack2bf(v[0], v[1]);   lo.y = pack2bf(v[2], v[3]);   lo.z = pack2bf(v[4], v[5]);   lo.w = pack2bf(v[6], v[7]);
        hi.x = pack2bf(v[8], v[9]);   hi.y = pack2bf(v[10], v[11]); hi.z = pack2bf(v[12], v[13]); hi.w = pack2bf(v[14], v[15]);
        uint4* o = (uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb);
        store16_stream(o, lo); store16_stream(o + 1, hi);
    }
}

__device__ __forceinline__ void load_bias8(const clibd_gemm_epilogue& ep, int nb, float bias[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (ep.bias != nullptr) {
        const f32x4 b0 = *(const f32x4*)(ep.bias + nb), b1 = *(const f32x4*)(ep.bias + nb + 4);
        bias[0] = b0[0]; bias[1] = b0[1]; bias[2] = b0[2]; bias[3] = b0[3];
        bias[4] = b1[0]; bias[5] = b1[1]; bias[6] = b1[2]; bias[7] = b1[3];
    }
}

__device__ __forceinline__ uint4 pack8bf(const float v[8]) {
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    return o;
}

// Epilogue kinds of the 256x256 kernel: the epilogue runs once per output row (16 rows per lane and tile), so the
// wave-uniform flag tests of the generic form (~10 scalar branches per row) are resolved ONCE per launch on the host
// and the common shapes get straight-line code.
enum : int {
    EPI_GENERIC = 0,    // anything (store_row8<EPI_GENERIC> tests every field)
    EPI_BF16 = 1,       // [bias] -> out_bf16                                   (qkv + LoRA, dgrad outputs)
    EPI_GELU_SAVE = 2,  // [bias] -> gelu -> out_bf16, gelu' -> out_pre_bf16    (fc1 forward)
    EPI_MUL_AUX = 3,    // * aux_bf16 -> out_bf16                               (fc2 dgrad x gelu')
    EPI_RES_F32 = 4,    // [bias] + residual_f32 -> out_f32                     (proj / fc2 forward)
    EPI_RES_F32_DROP = 5,  // [bias] -> dropout -> + residual_f32 -> out_f32    (BERT proj / fc2 forward, train mode)
    EPI_SPLITK_F32 = 6,    // plain fp32 store of this split's partial tile (split-K workspace mode)
    EPI_ADD_AUX = 7,       // + aux_bf16 -> out_bf16                            (dgrad joining a bf16 residual-gradient stream)
    EPI_GELU_SAVE_U8 = 8,  // EPI_GELU_SAVE with gelu' as one byte per element     (fc1 forward; opt-in: numerics gelu_grad="u8")
    EPI_MUL_AUX_U8 = 9,    // EPI_MUL_AUX reading those bytes                      (fc2 dgrad x gelu'; opt-in, same switch)
    EPI_RES_F32_COPY = 10,    // EPI_RES_F32 + a bf16 copy of the result + per-slice row sums (LN -> Linear fold, producer: ViT projection)
    EPI_ROWNORM_GELU = 11,    // rstd_m (acc - mean_m s_n) + b'_n -> EPI_GELU_SAVE                  (LN -> Linear fold, consumer: ViT fc1)
    EPI_GELU_SAVE_12 = 12,    // EPI_GELU_SAVE with gelu' as the 12-bit e4m7 form of its bf16 value (common.h gelu12_*; numerics gelu_grad="e4m7")
    EPI_MUL_AUX_12 = 13,      // EPI_MUL_AUX reading that form
    EPI_GELU = 14,            // [bias] -> gelu -> out_bf16: fc1 of the NO-GRAD forward (eval: no gelu' to save); round 6 — it ran on the generic kind before
    EPI_F32 = 15,             // [bias] -> out_f32: patch embedding, the MLM head's dgrad into the fp32 residual-gradient entry (generic kind before round 6)
    EPI_NUM_KINDS = 16,
};
constexpr bool epi_aux_kind(int kind) { return kind == EPI_MUL_AUX || kind == EPI_ADD_AUX || kind == EPI_MUL_AUX_U8 || kind == EPI_MUL_AUX_12; }
constexpr bool epi_mul_aux_kind(int kind) { return kind == EPI_MUL_AUX || kind == EPI_MUL_AUX_U8 || kind == EPI_MUL_AUX_12; }

__host__ __device__ inline int epilogue_kind(const clibd_gemm_epilogue& ep) {
    if (ep.split_k > 1) return EPI_GENERIC;
    const bool drop = ep.drop_thr16 > 0;
    if (ep.row_sums != nullptr)
        return (ep.act == CLIBD_ACT_NONE && !drop && !ep.out_pre_bf16 && ep.residual_f32 && ep.out_f32 && ep.out_bf16 && !ep.rank_u && ep.bias) ? EPI_RES_F32_COPY : -1;
    if (ep.row_stats != nullptr)
        return (ep.act == CLIBD_ACT_GELU_SAVE_GRAD && !drop && ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16 && !ep.rank_u && ep.bias && ep.col_sum_w) ? EPI_ROWNORM_GELU : -1;
    if (ep.act == CLIBD_ACT_NONE && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16) return EPI_BF16;
    if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD && !drop && ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16) return EPI_GELU_SAVE;
    if (ep.act == CLIBD_ACT_GELU && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16) return EPI_GELU;
    if (ep.act == CLIBD_ACT_MUL_AUX && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16 && ep.aux_bf16) return EPI_MUL_AUX;
    if (ep.act == CLIBD_ACT_ADD_AUX && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16 && ep.aux_bf16) return EPI_ADD_AUX;
    if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_U8 && !drop && ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16) return EPI_GELU_SAVE_U8;
    if (ep.act == CLIBD_ACT_MUL_AUX_U8 && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16 && ep.aux_bf16) return EPI_MUL_AUX_U8;
    if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_E12 && !drop && ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16) return EPI_GELU_SAVE_12;
    if (ep.act == CLIBD_ACT_MUL_AUX_E12 && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && !ep.out_f32 && ep.out_bf16 && ep.aux_bf16) return EPI_MUL_AUX_12;
    if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_E12 || ep.act == CLIBD_ACT_MUL_AUX_E12) return -1;   // any other combination: the 128x128 kernel (the 256x256 kernel declines kind < 0)
    if (ep.act == CLIBD_ACT_NONE && !ep.out_pre_bf16 && ep.residual_f32 && ep.out_f32 && !ep.out_bf16) return drop ? EPI_RES_F32_DROP : EPI_RES_F32;
    if (ep.act == CLIBD_ACT_NONE && !drop && !ep.out_pre_bf16 && !ep.residual_f32 && ep.out_f32 && !ep.out_bf16) return EPI_F32;
    return EPI_GENERIC;
}

// One output row m, 8 contiguous columns nb..nb+7 held by this lane (gemm256: the 16 lanes of a row cover 128 contiguous
// columns): v = acc + bias already applied by the caller.  Same operation order as store_row16; no split-K.
template <int KIND>
__device__ __forceinline__ void store_row8(const clibd_gemm_epilogue& ep, int m, int nb, float v[8]) {
    if (KIND == EPI_BF16) {
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_F32) {
        f32x4* o = (f32x4*)(ep.out_f32 + (size_t)m * ep.ld_out_f32 + nb);
        o[0] = (f32x4){v[0], v[1], v[2], v[3]};
        o[1] = (f32x4){v[4], v[5], v[6], v[7]};
        return;
    }
    if (KIND == EPI_GELU) {   // the generic form's arithmetic (gelu of the fp32 value, no bf16 rounding of the pre-activation), on packed pairs
        gelu_rows<8>(v);
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_GELU_SAVE) {
        float dg[8];
        gelu_and_grad_rows<8>(v, dg);
        *(uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) = pack8bf(dg);
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_GELU_SAVE_U8) {
        float dg[8];
        gelu_and_grad_rows<8>(v, dg);
        *(uint2*)((unsigned char*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) =
            make_uint2(geluq_pack4(dg[0], dg[1], dg[2], dg[3]), geluq_pack4(dg[4], dg[5], dg[6], dg[7]));
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_MUL_AUX_U8) {
        const uint2 c = *(const uint2*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        float d0[4], d1[4];
        geluq_unpack4(c.x, d0);
        geluq_unpack4(c.y, d1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_GELU_SAVE_12) {
        float dg[8];
        gelu_and_grad_rows<8>(v, dg);
        *(gelu12_row8*)((unsigned char*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + (nb >> 1) * 3) = gelu12_pack8(dg);
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_MUL_AUX_12) {
        const gelu12_row8 a = *(const gelu12_row8*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + (nb >> 1) * 3);
        float d[8];
        gelu12_unpack8(a.w0, a.w1, a.w2, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= d[e];
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_MUL_AUX || KIND == EPI_ADD_AUX) {
        const uint4 x0 = *(const uint4*)((const unsigned short*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        const unsigned xs[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf2f((unsigned short)(xs[e] & 0xffffu)), a1 = bf2f((unsigned short)(xs[e] >> 16));
            if (KIND == EPI_MUL_AUX) { v[2 * e] *= a0; v[2 * e + 1] *= a1; }
            else { v[2 * e] += a0; v[2 * e + 1] += a1; }
        }
        *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
        return;
    }
    if (KIND == EPI_GENERIC || KIND == EPI_RES_F32_DROP) {
        if (KIND == EPI_RES_F32_DROP || ep.drop_thr16 > 0) {
            const unsigned base = (unsigned)m * (unsigned)ep.drop_ld + (unsigned)nb;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float f0, f1;
                drop_pair(ep.drop_seed, base + e, (unsigned)ep.drop_thr16, ep.drop_scale, f0, f1);
                v[e] *= f0;
                v[e + 1] *= f1;
            }
        }
    }
    if (KIND == EPI_GENERIC) {
        // (the e4m7 forms have no GENERIC path in the 256x256 kernel: with them the generic epilogue's accumulators went to 528 bytes of scratch per lane
        //  and every generic launch — patch embedding, heads, the whole no-grad forward — ran 3 x slower; epilogue_kind() sends their odd combinations to the 128x128 kernel)
        if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD_U8) {
            float dg[8];
            gelu_and_grad_rows<8>(v, dg);
            *(uint2*)((unsigned char*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) =
                make_uint2(geluq_pack4(dg[0], dg[1], dg[2], dg[3]), geluq_pack4(dg[4], dg[5], dg[6], dg[7]));
        } else if (ep.act == CLIBD_ACT_GELU_SAVE_GRAD) {
            float dg[8];
            gelu_and_grad_rows<8>(v, dg);
            *(uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) = pack8bf(dg);
        } else if (ep.out_pre_bf16 != nullptr) {
            *(uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) = pack8bf(v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bfround(v[e]);
        }
        if (ep.act == CLIBD_ACT_GELU) {
            gelu_rows<8>(v);
        } else if (ep.act == CLIBD_ACT_MUL_AUX_U8) {
            const uint2 c = *(const uint2*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
            float d0[4], d1[4];
            geluq_unpack4(c.x, d0);
            geluq_unpack4(c.y, d1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
        } else if (ep.act == CLIBD_ACT_GELU_GRAD || ep.act == CLIBD_ACT_MUL_AUX || ep.act == CLIBD_ACT_ADD_AUX) {
            const uint4 x0 = *(const uint4*)((const unsigned short*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
            const unsigned xs[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = bf2f((unsigned short)(xs[e] & 0xffffu)), a1 = bf2f((unsigned short)(xs[e] >> 16));
                if (ep.act == CLIBD_ACT_GELU_GRAD) {
                    v[2 * e] *= gelu_grad_f(a0);
                    v[2 * e + 1] *= gelu_grad_f(a1);
                } else if (ep.act == CLIBD_ACT_MUL_AUX) {
                    v[2 * e] *= a0;
                    v[2 * e + 1] *= a1;
                } else {
                    v[2 * e] += a0;
                    v[2 * e + 1] += a1;
                }
            }
        }
    }
    if (KIND == EPI_RES_F32 || KIND == EPI_RES_F32_DROP || ep.residual_f32 != nullptr) {
        const f32x4* rs = (const f32x4*)(ep.residual_f32 + (size_t)m * ep.ld_res + nb);
        const f32x4 r0 = rs[0], r1 = rs[1];
        v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
        v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
    }
    if (KIND == EPI_RES_F32 || KIND == EPI_RES_F32_DROP || ep.out_f32 != nullptr) {
        f32x4* o = (f32x4*)(ep.out_f32 + (size_t)m * ep.ld_out_f32 + nb);
        o[0] = (f32x4){v[0], v[1], v[2], v[3]};
        o[1] = (f32x4){v[4], v[5], v[6], v[7]};
    }
    if (KIND == EPI_GENERIC && ep.out_bf16 != nullptr) *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
}

// fp8-forward fc1: gelu' saved as bf16 for the backward, the activation itself leaves as fp8 (fc2's operand)
__device__ __forceinline__ void store_row8_gelu_fp8(const clibd_gemm_epilogue& ep, int m, int nb, float v[8], float out_scale) {
    float dg[8];
    gelu_and_grad_rows<8>(v, dg);
    *(uint4*)((unsigned short*)ep.out_pre_bf16 + (size_t)m * ep.ld_pre + nb) = pack8bf(dg);
    *(uint2*)((unsigned char*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8fp8(v, out_scale);
}

// Pass 1 of the two-pass epilogues (gemm256): everything of store_row8<KIND> up to, not including, the stores, on operands
// the caller has already loaded (ax: 8 bf16 of aux for EPI_MUL_AUX; r0, r1: 8 fp32 of the residual otherwise).
template <int KIND>
__device__ __forceinline__ void fold_row8_in(const clibd_gemm_epilogue& ep, int m, int nb, float v[8], const uint4& ax, const f32x4& r0,
                                             const f32x4& r1) {
    if (KIND == EPI_MUL_AUX_U8) {   // ax.x, ax.y: the row's eight codes
        float d0[4], d1[4];
        geluq_unpack4(ax.x, d0);
        geluq_unpack4(ax.y, d1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
    } else if (KIND == EPI_MUL_AUX_12) {   // ax.x, ax.y, ax.z: the row's eight 12-bit values
        float d[8];
        gelu12_unpack8(ax.x, ax.y, ax.z, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= d[e];
    } else if (KIND == EPI_MUL_AUX || KIND == EPI_ADD_AUX) {
        const unsigned xs[4] = {ax.x, ax.y, ax.z, ax.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf2f((unsigned short)(xs[e] & 0xffffu)), a1 = bf2f((unsigned short)(xs[e] >> 16));
            if (KIND == EPI_MUL_AUX) { v[2 * e] *= a0; v[2 * e + 1] *= a1; }
            else { v[2 * e] += a0; v[2 * e + 1] += a1; }
        }
    } else {  // EPI_RES_F32 / EPI_RES_F32_DROP
        if (KIND == EPI_RES_F32_DROP) {
            const unsigned base = (unsigned)m * (unsigned)ep.drop_ld + (unsigned)nb;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float f0, f1;
                drop_pair(ep.drop_seed, base + e, (unsigned)ep.drop_thr16, ep.drop_scale, f0, f1);
                v[e] *= f0;
                v[e + 1] *= f1;
            }
        }
        v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
        v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
    }
}

// LN -> Linear fold, consumer: v[e] = rstd * (acc[e] - mean * s[e]) + b'[e]   (st = (mean, rstd) of the row)
__device__ __forceinline__ void fold_rownorm8(float v[8], float mean, float rstd, const float s[8], const float b[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(rstd, fmaf(-mean, s[e], v[e]), b[e]);
}

// sum over the 16 lanes of a DPP row (the lanes that hold one output row's 128 columns in gemm256): every lane gets the total
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_mov<DPP_QUAD_XOR1>(x);
    x += dpp_mov<DPP_QUAD_XOR2>(x);
    x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
    x += dpp_mov<DPP_ROW_MIRROR>(x);
    return x;
}

// the same with the loads
// The 16-byte aux load of the MUL_AUX / ADD_AUX epilogues (gelu' / a residual-gradient stream: read ONCE, never again) carries the non-temporal
// hint (round 6), so that the stream does not displace the A / W panels the workgroups of an XCD share in L2: FETCH_SIZE of the fc2 dgrad x gelu'
// launch 6.37 -> 5.76 GB, step -0.14 % in an interleaved same-box A/B (profiles/r06_exp_nt_aux_loads.log).  -DCLIBD_PLAIN_AUX_LOADS builds the A/B partner.
// NT = false: the ADD_AUX stream (a residual gradient the preceding LayerNorm backward has just written: per-kernel times put the hint at +3 % there).
template <bool NT>
__device__ __forceinline__ uint4 load_aux16(const void* p) {
    typedef unsigned aux_u32x4 __attribute__((ext_vector_type(4)));
#ifndef CLIBD_PLAIN_AUX_LOADS
    const aux_u32x4 v = NT ? __builtin_nontemporal_load((const aux_u32x4*)p) : *(const aux_u32x4*)p;
#else
    const aux_u32x4 v = *(const aux_u32x4*)p;
#endif
    return make_uint4(v[0], v[1], v[2], v[3]);
}
#ifndef CLIBD_NT_ADD_AUX
#define CLIBD_NT_ADD_AUX 0
#endif
constexpr bool epi_aux_nt(int kind) { return kind != EPI_ADD_AUX || CLIBD_NT_ADD_AUX; }

// the 12-byte row piece of the e4m7 gelu' form (eight columns), non-temporal like load_aux16
__device__ __forceinline__ uint4 load_aux12(const void* p) {
    typedef unsigned aux_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
#ifndef CLIBD_PLAIN_AUX_LOADS
    const aux_u32x3 v = __builtin_nontemporal_load((const aux_u32x3*)p);
#else
    const aux_u32x3 v = *(const aux_u32x3*)p;
#endif
    return make_uint4(v[0], v[1], v[2], 0u);
}

template <int KIND>
__device__ __forceinline__ void fold_row8(const clibd_gemm_epilogue& ep, int m, int nb, float v[8]) {
    uint4 ax = make_uint4(0u, 0u, 0u, 0u);
    f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = r0;
    if (KIND == EPI_MUL_AUX_U8) {
        const uint2 c = *(const uint2*)((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
        ax.x = c.x; ax.y = c.y;
    } else if (KIND == EPI_MUL_AUX_12) {
        ax = load_aux12((const unsigned char*)ep.aux_bf16 + (size_t)m * ep.ld_aux + (nb >> 1) * 3);
    } else if (KIND == EPI_MUL_AUX || KIND == EPI_ADD_AUX) {
        ax = load_aux16<epi_aux_nt(KIND)>((const unsigned short*)ep.aux_bf16 + (size_t)m * ep.ld_aux + nb);
    } else {
        const f32x4* rs = (const f32x4*)(ep.residual_f32 + (size_t)m * ep.ld_res + nb);
        r0 = rs[0];
        r1 = rs[1];
    }
    fold_row8_in<KIND>(ep, m, nb, v, ax, r0, r1);
}

// K-slice plan shared by the two split-K launchers: nk (even, >= 4) 64-deep K-tiles over about one round of work items
// (tiles x splits ~ CUs); every slice is an even number of K-tiles and the LAST one keeps >= 4 (the kernels' pipeline
// prologue needs two K-tile pairs).  Always succeeds: nks grows by 2 until the tail is long enough, and nks = nk
// (one slice) ends the walk at the latest.  Returns the number of slices, *nks_out = K-tiles per slice.
inline int plan_k_slices(int nk, int tiles, int num_cus, int* nks_out) {
    int splits = num_cus / (tiles > 0 ? tiles : 1);
    if (splits < 1) splits = 1;
    if (splits > nk / 4) splits = nk / 4;
    if (splits < 1) splits = 1;
    int nks = (nk + splits - 1) / splits;
    nks += nks & 1;
    if (nks < 4) nks = 4;
    if (nks > nk) nks = nk;
    splits = (nk + nks - 1) / nks;
    while (nk - (splits - 1) * nks < 4 && nks < nk) {
        nks += 2;
        if (nks > nk) nks = nk;
        splits = (nk + nks - 1) / nks;
    }
    *nks_out = nks;
    return splits;
}

// host side (gemm256.hip): returns true when the 256x256 kernel took the launch
bool gemm256_try_launch(const GemmParams& p, hipStream_t stream);
// bytes of the stream-K tail workspace a launch of this shape can use (0: none): partial tiles + flags
size_t gemm256_tail_workspace_bytes(int M, int N, int K);
// fp8 operands (p.fp8, p.col_scale set; K / lda / ldw in bytes): true when launched
bool gemm256_fp8_launch(const GemmParams& p, hipStream_t stream);
bool gemm256_fp8_dgrad_launch(const GemmParams& p, hipStream_t stream);
// split-K with a partials workspace; returns the number of splits (0: shape not taken)
int gemm256_splitk_launch(const GemmParams& p, float* partials, size_t partials_elems, hipStream_t stream);

// rows-contracting ("TN") split-M GEMM, gemm256_tn.hip: partials[s] (fp32 [Na, Nb]) = A[slice s, :Na]^T · B[slice s, :Nb];
// returns the number of slices (0: shape not taken)
int gemm256_tn_splitk_launch(const unsigned short* A, int lda, const unsigned short* B, int ldb, int M, int Na, int Nb, float* partials,
                             size_t partials_elems, float* colsum, hipStream_t stream);

}  // namespace clibd
