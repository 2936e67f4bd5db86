// Shared device helpers for the CLIBD gfx950 kernels (wave64, MFMA bf16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clibd {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (one MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// round-to-nearest-even f32 -> bf16 (plain cast keeps NaN a NaN; lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short h) {
    return __builtin_bit_cast(float, ((unsigned)h) << 16);
}
// two f32 -> packed bf16 pair (lo in bits 0..15): the vector conversion lowers to ONE v_cvt_pk_bf16_f32; converting the
// halves separately costs the same instruction twice plus and / shift / or
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t r = __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bfround(float x) { return bf2f(f2bf(x)); }

// ---- OCP fp8 e4m3 (gfx950's v_cvt_pk_fp8_f32; max finite 448) for the fp8-forward mode: values are clamped first, so the
// result never depends on the conversion's overflow mode.  pack4fp8: byte i = fp8(v_i).
__device__ __forceinline__ unsigned pack4fp8(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ uint2 pack8fp8(const float v[8], float s) {
    uint2 o;
    o.x = pack4fp8(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    o.y = pack4fp8(v[4] * s, v[5] * s, v[6] * s, v[7] * s);
    return o;
}

// 16-byte LDS-DMA: each lane supplies its own global source; LDS dest = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// 16-byte store with the non-temporal hint: GEMM outputs are written once and re-read by a later kernel, long after
// they would have been evicted; keeping them out of the way leaves the XCD L2 to the A / W panels other workgroups share
__device__ __forceinline__ void store16_stream(void* p, uint4 v) {
#ifdef CLIBD_NT_STORES
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((u32x4_t){v.x, v.y, v.z, v.w}, (u32x4_t*)p);
#else
    *(uint4*)p = v;
#endif
}

// ---- counter-based dropout (HF BERT hidden / attention-probability dropout in train mode) --------------------------------
// One 32-bit hash (lowbias32) of (element index >> 1) ^ seed serves an even/odd element PAIR with 16 bits each:
// keep iff bits >= thr16 = round(p * 65536).  The mask is a pure function of (seed, index): the backward recomputes it.
__device__ __forceinline__ unsigned drop_hash(unsigned seed, unsigned pair) {
    unsigned x = pair ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// scale factors (0 or 1/(1-p)) of elements idx (even) and idx+1
__device__ __forceinline__ void drop_pair(unsigned seed, unsigned idx_even, unsigned thr16, float scale, float& f0, float& f1) {
    const unsigned h = drop_hash(seed, idx_even >> 1);
    f0 = (h & 0xffffu) >= thr16 ? scale : 0.f;
    f1 = (h >> 16) >= thr16 ? scale : 0.f;
}
__device__ __forceinline__ float drop_one(unsigned seed, unsigned idx, unsigned thr16, float scale) {
    const unsigned h = drop_hash(seed, idx >> 1);
    return (((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr16) ? scale : 0.f;
}

// ---- cross-lane reductions on the VALU data path only (DPP + v_permlane{16,32}_swap): no LDS-crossbar instruction.
// Round 2 finding (tools/stress_streams.py, tools/stress_ln.py; regression: tests/test_streams_gpu.py): layernorm_fwd's LoRA down-projection, whose reduction used
// ds_bpermute_b32 (what __shfl_xor lowers to) right after its ds_read of the adapter matrix, returned wrong sums in a few rows
// per launch whenever an attention-forward kernel of ANOTHER stream was co-resident on the CU (correct alone, correct beside
// GEMM / LayerNorm kernels; draining every ds_bpermute with lgkmcnt(0) did not help).  The reductions below never touch the
// LDS unit and are also cheaper (the permlane swap replaces two selects and a ds_bpermute of the butterfly).
// Inline asm for the swaps: ROCm 7.2's hipcc returned the two results of __builtin_amdgcn_permlane32_swap in ONE register when
// they were added.  The s_nop pads are the VALU-write -> permlane-read wait states hipcc puts around its own swaps.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140, DPP_ROW_ROR8 = 0x128;
// after the swap: a = {a[0:31], b[0:31]}, b = {a[32:63], b[32:63]}  (tools/micro/permlane_test.hip)
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// 16-lane rows r0..r3: a = {a.r0, b.r0, a.r2, b.r2}, b = {a.r1, b.r1, a.r3, b.r3}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// v + (lane ^ 16) + (lane ^ 32) + (lane ^ 48): the four 16-lane rows combined at a fixed position in the row
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    permlane16_swap(a, b);   // a + b = even+odd row of each pair, in every lane of the pair
    v = a + b;
    a = v; b = v;
    permlane32_swap(a, b);
    return a + b;
}
__device__ __forceinline__ float rows4_max(float v) {
    float a = v, b = v;
    permlane16_swap(a, b);
    v = fmaxf(a, b);
    a = v; b = v;
    permlane32_swap(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float wave_sum(float v) {   // every lane returns the total of the 64 lanes
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);   // lane i <-> 7 - i of its group of 8: quads 0 and 1 combined
    v += dpp_mov<DPP_ROW_MIRROR>(v);        // lane i <-> 15 - i: the other half of the 16-lane row
    return rows4_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return rows4_max(v);
}

// erf with |abs err| <= 1.5e-7 (Abramowitz-Stegun 7.1.26); outputs are rounded to bf16 downstream
__device__ __forceinline__ float fast_erf(float x) {
    float ax = fabsf(x);
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float y = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    float e = __expf(-ax * ax);
    float r = 1.0f - y * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) {
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f));
}
// gelu(x) and gelu'(x) from ONE erf evaluation: exp(-(x/sqrt2)^2) inside the erf is also the Gaussian of phi(x)
__device__ __forceinline__ void gelu_and_grad_f(float x, float& y, float& dy) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = __expf(-ax * ax);                       // = exp(-x^2 / 2)
    const float erfv = copysignf(1.0f - poly * e, x);
    const float cdf = 0.5f * (1.0f + erfv);
    y = x * cdf;
    dy = cdf + x * (0.3989422804014327f * e);
}
// The same for two elements, written on 2-vectors: the multiplies / fmas lower to v_pk_mul_f32 / v_pk_fma_f32 (packed fp32 issues
// two lanes' worth per instruction).  tools/exp_fc1_epilogue.py, ViT fc1 at b=2048: the GELU form costs 2450 us against 1650 us
// for the same GEMM with one plain bf16 store; with the arithmetic stubbed out 2110 us, i.e. ~450 us is the second 128-KB store
// per tile and ~290 us the arithmetic.  Constants are folded to leave per element |x|, copysign, v_rcp, v_exp and 14
// packed-pair multiplies / fmas (2400 us).
//   exp(-x^2/2) = exp2(-(K|x|)^2), K = sqrt(log2(e)/2);  A&S 7.1.26 argument p*|x|/sqrt2 = P * (K|x|);  0.5 folded into the polynomial.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_grad_x2(f32x2 x, f32x2& y, f32x2& dy) {
    constexpr float K = 0.84932180028801907f;
    constexpr float P = 0.3275911f * 0.70710678118654752f / K;
    f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    ax = ax * K;
    const f32x2 d = ax * P + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 poly = ((((0.5307027145f * t - 0.7265760135f) * t + 0.7107068705f) * t - 0.142248368f) * t + 0.127414796f) * t;
    const f32x2 s = ax * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(-s[0]), __builtin_amdgcn_exp2f(-s[1])};   // exp(-x^2/2)
    const f32x2 h = 0.5f - poly * e;                                                    // erf(|x|/sqrt2) / 2 >= 0
    const f32x2 hs = {copysignf(h[0], x[0]), copysignf(h[1], x[1])};
    const f32x2 cdf = hs + 0.5f;
    y = x * cdf;
    dy = (x * e) * 0.3989422804014327f + cdf;
}
// two f32 rounded to bf16 and widened again (one v_cvt_pk_bf16_f32 + shift + mask)
__device__ __forceinline__ f32x2 bfround2(float a, float b) {
    const unsigned pk = pack2bf(a, b);
    return (f32x2){__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
// v[0..N) -> gelu(bf16(v)), dg = gelu'(bf16(v)), pairwise
template <int N>
__device__ __forceinline__ void gelu_and_grad_rows(float* v, float* dg) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        f32x2 y, d;
        gelu_and_grad_x2(bfround2(v[e], v[e + 1]), y, d);
        v[e] = y[0]; v[e + 1] = y[1];
        dg[e] = d[0]; dg[e + 1] = d[1];
    }
}
// v[0..N) -> gelu(v), pairwise on the packed form (no bf16 rounding of the argument: CLIBD_ACT_GELU's contract); the unused gelu' folds away
template <int N>
__device__ __forceinline__ void gelu_rows(float* v) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        f32x2 y, d;
        gelu_and_grad_x2((f32x2){v[e], v[e + 1]}, y, d);
        v[e] = y[0]; v[e + 1] = y[1];
    }
}
// gelu'(x) lies in [-0.1290, 1.1290]: kept for the backward as ONE BYTE, code = rint((g' - LO) / STEP) over [LO, LO + 255 STEP]
// (|error| <= STEP / 2 = 2.5e-3; bf16's SPACING is 3.9e-3 in [0.5, 1) and 7.8e-3 in [1, 2), i.e. a rounding error of at most half of
// that: 2.0e-3 / 3.9e-3 — the figures engine.py quotes).  CLIBD_ACT_GELU_SAVE_GRAD_U8 writes
// the codes, CLIBD_ACT_MUL_AUX_U8 multiplies by the decoded value: half the bytes of the bf16 form on both sides.
constexpr float GELUQ_LO = -0.1328125f;
constexpr float GELUQ_STEP = 1.265625f / 255.0f;
constexpr float GELUQ_INV = 255.0f / 1.265625f;
__device__ __forceinline__ unsigned geluq_pack4(float a, float b, float c, float d) {   // v_cvt_pk_u8_f32: round + saturate + insert byte
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(a, GELUQ_INV, -GELUQ_LO * GELUQ_INV), 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, GELUQ_INV, -GELUQ_LO * GELUQ_INV), 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(c, GELUQ_INV, -GELUQ_LO * GELUQ_INV), 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, GELUQ_INV, -GELUQ_LO * GELUQ_INV), 3, w);
    return w;
}
__device__ __forceinline__ void geluq_unpack4(unsigned w, float out[4]) {               // v_cvt_f32_ubyte0..3 + fma
    out[0] = fmaf((float)(w & 0xffu), GELUQ_STEP, GELUQ_LO);
    out[1] = fmaf((float)((w >> 8) & 0xffu), GELUQ_STEP, GELUQ_LO);
    out[2] = fmaf((float)((w >> 16) & 0xffu), GELUQ_STEP, GELUQ_LO);
    out[3] = fmaf((float)(w >> 24), GELUQ_STEP, GELUQ_LO);
}

// gelu' as TWELVE bits, "e4m7" (round 6): sign, a 4-bit exponent and bf16's 7 mantissa bits — the bf16 value itself with the exponent re-biased
// to the sixteen binades gelu' lives in.  e4 = 0 is zero; e4 = 1 .. 15 are the binades [2^-14, 2^-13) .. [1, 2).  Every bf16 value of
// magnitude in [2^-14, 2) — gelu' lies in [-0.129, 1.129] — survives BIT FOR BIT; smaller magnitudes (|gelu'| < 6.1e-5: pre-activations
// below about -4.55) become a signed zero.  So this is the bf16 form at 1.5 bytes per element, not a coarser code: the decoded operand of
// the fc2 dgrad differs from the bf16 form's only where |gelu'| < 2^-14.  Eight adjacent columns = 12 bytes = three dwords per lane and row.
__device__ __forceinline__ unsigned gelu12_code_bits(unsigned b) {   // b: the bf16 bit pattern of gelu'
    const unsigned t = b & 0x7fffu;
    unsigned u = t >= (113u << 7) ? t - (112u << 7) : 0u;
    u = u > 0x7ffu ? 0x7ffu : u;                       // (|g| >= 2 cannot happen for gelu': saturate instead of wrapping)
    return ((b >> 4) & 0x800u) | u;
}
__device__ __forceinline__ float gelu12_val(unsigned c) {
    const unsigned u = c & 0x7ffu;
    const unsigned t = u ? u + (112u << 7) : 0u;
    return __uint_as_float((((c & 0x800u) << 4) | t) << 16);
}
struct __attribute__((packed, aligned(4))) gelu12_row8 { unsigned w0, w1, w2; };
__device__ __forceinline__ gelu12_row8 gelu12_pack8(const float* g) {
    unsigned c[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned pk = pack2bf(g[e], g[e + 1]);       // the bf16 form's rounding, one v_cvt_pk per pair
        c[e] = gelu12_code_bits(pk & 0xffffu);
        c[e + 1] = gelu12_code_bits(pk >> 16);
    }
    gelu12_row8 r;
    r.w0 = c[0] | (c[1] << 12) | (c[2] << 24);
    r.w1 = (c[2] >> 8) | (c[3] << 4) | (c[4] << 16) | (c[5] << 28);
    r.w2 = (c[5] >> 4) | (c[6] << 8) | (c[7] << 20);
    return r;
}
__device__ __forceinline__ void gelu12_unpack8(unsigned w0, unsigned w1, unsigned w2, float* g) {
    g[0] = gelu12_val(w0 & 0xfffu);
    g[1] = gelu12_val((w0 >> 12) & 0xfffu);
    g[2] = gelu12_val((w0 >> 24) | ((w1 & 0xfu) << 8));
    g[3] = gelu12_val((w1 >> 4) & 0xfffu);
    g[4] = gelu12_val((w1 >> 16) & 0xfffu);
    g[5] = gelu12_val((w1 >> 28) | ((w2 & 0xffu) << 4));
    g[6] = gelu12_val((w2 >> 8) & 0xfffu);
    g[7] = gelu12_val(w2 >> 20);
}

// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

}  // namespace clibd
