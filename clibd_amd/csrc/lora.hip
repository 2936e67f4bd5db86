// LoRA (rank 4 on q and v) helper kernels for gfx950.
//
// Forward/backward rank-8 updates themselves ride inside the GEMM (one extra MFMA k-step, gemm.hip); the
// down-projection t = x·A^T rides inside LayerNorm (layernorm.hip).  What is left here:
//   * lora_pack:  fp32 adapter parameters -> the bf16 operand images those kernels consume (tiny, per step);
//   * lora_wgrad: dA, dB — contractions over the token dimension M; HBM-bound (reads dq, dv, x once).
//     dt = dq·B_q | dv·B_v is produced by the MFMA GEMM against the packed [16, 3H] image (w_dt).
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

// a_q,a_v: [4,H] (nn.Linear(H,4).weight); b_q,b_v: [H,4] (nn.Linear(4,H).weight)
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* __restrict__ a_q, const float* __restrict__ a_v,
                                                        const float* __restrict__ b_q, const float* __restrict__ b_v, int H,
                                                        unsigned short* __restrict__ v_fwd,   // [3H, 8]
                                                        unsigned short* __restrict__ v_bwd,   // [H, 8]
                                                        unsigned short* __restrict__ a_cat,   // [8, H]
                                                        unsigned short* __restrict__ w_dt) {  // [16, 3H]
    const int n = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 3H-1
    if (n >= 3 * H) return;
    const int seg = n / H, c = n - seg * H;
    unsigned short vf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (seg == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[r] = f2bf(b_q[(size_t)c * 4 + r]);
    } else if (seg == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[4 + r] = f2bf(b_v[(size_t)c * 4 + r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v_fwd[(size_t)n * 8 + r] = vf[r];
    // w_dt[r, n]: rows 0-3 = B_q[:, r]^T over the q segment, rows 4-7 = B_v[:, r-4]^T over the v segment, rest 0
#pragma unroll
    for (int r = 0; r < 16; ++r) w_dt[(size_t)r * 3 * H + n] = (r < 8) ? vf[r] : (unsigned short)0;
    if (seg == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned short aq = f2bf(a_q[(size_t)r * H + c]), av = f2bf(a_v[(size_t)r * H + c]);
            v_bwd[(size_t)c * 8 + r] = aq;
            v_bwd[(size_t)c * 8 + 4 + r] = av;
            a_cat[(size_t)r * H + c] = aq;
            a_cat[(size_t)(4 + r) * H + c] = av;
        }
    }
}

// cross-group (4 row groups) reduction of one 4x4 accumulator family through LDS; the reduced 4H values are then laid
// out in OUTPUT order in LDS (padded against bank conflicts) and added to global memory with wave-contiguous atomics
// (256 B per wave-instruction: scattered float atomics run an order of magnitude slower on gfx950).
// IS_A: output layout dA[r, k] (k = c + e) else dB[n, r] (n = c + e).
__device__ __forceinline__ int lw_pad(int i) { return i + (i >> 5); }

template <bool IS_A>
__device__ __forceinline__ void lw_reduce_emit(float (&acc)[4][4], float* red, float* outl, int grp, int ct, int cthreads,
                                               float* __restrict__ out, int c, int H) {
    if (grp > 0) {
        float* dst = red + ((size_t)(grp - 1) * cthreads + ct) * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e) *(f32x4*)(dst + 4 * e) = (f32x4){acc[e][0], acc[e][1], acc[e][2], acc[e][3]};
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int gsrc = 0; gsrc < 3; ++gsrc) {
            const float* src = red + ((size_t)gsrc * cthreads + ct) * 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 v = *(const f32x4*)(src + 4 * e);
                acc[e][0] += v[0]; acc[e][1] += v[1]; acc[e][2] += v[2]; acc[e][3] += v[3];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) outl[lw_pad(IS_A ? r * H + c + e : (c + e) * 4 + r)] = acc[e][r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) atomicAdd(out + i, outl[lw_pad(i)]);
    __syncthreads();
}

// dB_q[n,r] += sum_m dq[m,n] t[m,r]      dB_v[n,r] += sum_m dv[m,n] t[m,4+r]
// dA_q[r,k] += sum_m dt[m,r] x[m,k]      dA_v[r,k] += sum_m dt[m,4+r] x[m,k]
// block = 4 row-groups x (H/4) column threads; a thread owns 4 adjacent columns; a block owns ROWS_PER_BLOCK rows.
constexpr int LW_ROWS = 256;

__global__ __launch_bounds__(1024) void lora_wgrad_kernel(const unsigned short* __restrict__ dqkv, int ld,
                                                          const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ t,
                                                          const unsigned short* __restrict__ dt, int ld_dt, int M, int H,
                                                          float* __restrict__ dA_q, float* __restrict__ dA_v,
                                                          float* __restrict__ dB_q, float* __restrict__ dB_v) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3][H/4][16] reduction slabs, then [4H + 4H/32] output image
    float* outl = red + 3 * (H >> 2) * 16;
    const int cthreads = H >> 2;
    const int grp = threadIdx.x / cthreads;
    const int ct = threadIdx.x - grp * cthreads;
    const int c = ct * 4;
    const int m0 = blockIdx.x * LW_ROWS;
    const int m1 = min(m0 + LW_ROWS, M);
    float accBq[4][4], accBv[4][4], accAq[4][4], accAv[4][4];  // [col e][r]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accBq[e][r] = 0.f; accBv[e][r] = 0.f; accAq[e][r] = 0.f; accAv[e][r] = 0.f; }

    // 4 rows in flight per thread (independent loads issued before any use): the kernel is pure HBM streaming
    constexpr int UNR = 4;
    for (int mb = m0 + grp; mb < m1; mb += 4 * UNR) {
      uint2 dq2u[UNR], dv2u[UNR], x2u[UNR];
      uint4 t4u[UNR], d4u[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int m = min(mb + 4 * u, m1 - 1);  // clamped rows are skipped below
        dq2u[u] = *(const uint2*)(dqkv + (size_t)m * ld + c);
        dv2u[u] = *(const uint2*)(dqkv + (size_t)m * ld + 2 * H + c);
        x2u[u] = *(const uint2*)(x + (size_t)m * H + c);
        t4u[u] = *(const uint4*)(t + (size_t)m * 8);
        d4u[u] = *(const uint4*)(dt + (size_t)m * ld_dt);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (mb + 4 * u >= m1) break;
        const uint2 dq2 = dq2u[u], dv2 = dv2u[u], x2 = x2u[u];
        const uint4 t4 = t4u[u], d4 = d4u[u];
        const float dq[4] = {bf2f((unsigned short)(dq2.x & 0xffff)), bf2f((unsigned short)(dq2.x >> 16)),
                             bf2f((unsigned short)(dq2.y & 0xffff)), bf2f((unsigned short)(dq2.y >> 16))};
        const float dv[4] = {bf2f((unsigned short)(dv2.x & 0xffff)), bf2f((unsigned short)(dv2.x >> 16)),
                             bf2f((unsigned short)(dv2.y & 0xffff)), bf2f((unsigned short)(dv2.y >> 16))};
        const float xv[4] = {bf2f((unsigned short)(x2.x & 0xffff)), bf2f((unsigned short)(x2.x >> 16)),
                             bf2f((unsigned short)(x2.y & 0xffff)), bf2f((unsigned short)(x2.y >> 16))};
        const float tq[4] = {bf2f((unsigned short)(t4.x & 0xffff)), bf2f((unsigned short)(t4.x >> 16)),
                             bf2f((unsigned short)(t4.y & 0xffff)), bf2f((unsigned short)(t4.y >> 16))};
        const float tv[4] = {bf2f((unsigned short)(t4.z & 0xffff)), bf2f((unsigned short)(t4.z >> 16)),
                             bf2f((unsigned short)(t4.w & 0xffff)), bf2f((unsigned short)(t4.w >> 16))};
        const float gq[4] = {bf2f((unsigned short)(d4.x & 0xffff)), bf2f((unsigned short)(d4.x >> 16)),
                             bf2f((unsigned short)(d4.y & 0xffff)), bf2f((unsigned short)(d4.y >> 16))};
        const float gv[4] = {bf2f((unsigned short)(d4.z & 0xffff)), bf2f((unsigned short)(d4.z >> 16)),
                             bf2f((unsigned short)(d4.w & 0xffff)), bf2f((unsigned short)(d4.w >> 16))};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accBq[e][r] += dq[e] * tq[r];
                accBv[e][r] += dv[e] * tv[r];
                accAq[e][r] += gq[r] * xv[e];
                accAv[e][r] += gv[r] * xv[e];
            }
      }
    }
    lw_reduce_emit<false>(accBq, red, outl, grp, ct, cthreads, dB_q, c, H);
    lw_reduce_emit<false>(accBv, red, outl, grp, ct, cthreads, dB_v, c, H);
    lw_reduce_emit<true>(accAq, red, outl, grp, ct, cthreads, dA_q, c, H);
    lw_reduce_emit<true>(accAv, red, outl, grp, ct, cthreads, dA_v, c, H);
}

// ---- MFMA form of the same contractions (large M): Z[3H, 16] = L^T R with L = [dq | dv | x] (one row per token) and
// R = [t (8) | dt (8)]: rows dq x cols 0-3 = dB_q, rows dv x cols 4-7 = dB_v, rows x x cols 8-11 / 12-15 = dA_q^T / dA_v^T.
// The VALU kernel above spends 16 FMAs and ~10 conversions per loaded element and streams at 4 TB/s; here a 32-token slab of
// one COLUMN SEGMENT of L (H/2 columns: blockIdx.y = 2 x {dq, dv, x} + half) is loaded with plain 16-byte loads (three slabs in
// flight per lane: LDS-DMA, one 8-row piece per instruction, tops out near 4 TB/s on this part), written to LDS as [32][64] bf16
// tiles (R beside it), and every 16-column group costs ONE v_mfma_f32_16x16x32_bf16 per slab on transposed fragments
// (ds_read_b64_tr_b16, the attention kernels' tile image): the kernel is its loads.  A workgroup keeps its segment of Z in
// accumulators over all its slabs and adds it to the outputs once.
typedef __attribute__((ext_vector_type(4))) short lw_s16x4;
typedef __attribute__((address_space(3))) lw_s16x4 lw_lds_s16x4;
__device__ __forceinline__ int lwm_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// fragment whose MFMA row index is the image COLUMN 16 dt + (lane & 15) and whose k-slots walk the 32 image rows
__device__ __forceinline__ bf16x8 lwm_tr_frag(const char* tile, int dt, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int q = i >> 2, pp = i & 3;
    const int r0 = 4 * g + q;
    const int ch = 2 * dt + (pp >> 1);
    const lw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lw_lds_s16x4*)(tile + lwm_tile_off(r0, ch) + 8 * (pp & 1)));
    const lw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lw_lds_s16x4*)(tile + lwm_tile_off(r0 + 16, ch) + 8 * (pp & 1)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int LWM_ROWS = 32;
template <int TILES>   // 64-column tiles per segment = H / 128
__global__ __launch_bounds__(256) void lora_wgrad_mfma_kernel(const unsigned short* __restrict__ dqkv, int ld,
                                                              const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ t,
                                                              const unsigned short* __restrict__ dt, int ld_dt, int M, int H,
                                                              float* __restrict__ dA_q, float* __restrict__ dA_v,
                                                              float* __restrict__ dB_q, float* __restrict__ dB_v, int which0,
                                                              float* __restrict__ ws) {
    // ws (round 5): per-workgroup partials instead of float atomics — copy blockIdx.x of [dB_q | dB_v | dA_q | dA_v] (16 H floats);
    // this workgroup writes its own column segment of its own matrix there (plain stores: every element of a copy has one writer), and
    // lora_reduce_kernel adds the copies in a fixed order: no 256-way contended atomics (38-54 us of a 286-377 us backward at b = 2048:
    // profiles/r05_exp_lora_atomics_bound.log) and a gradient that is bit-reproducible.  nullptr: the float atomics of rounds 1-4.
    // which0: first matrix of the launch (0: dq, dv and x segments, grid.y = 6;  2: the x segments only, grid.y = 2 — dB then
    // comes from lora_dt_db_kernel)
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    constexpr int BUF = (TILES + 1) * 4096;    // TILES tiles of L, then the R tile ([32][64] images, only chunks 0, 1 of R are used)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int which = which0 + (blockIdx.y >> 1), half = blockIdx.y & 1;
    const int col0 = half * (H >> 1);          // first column of this segment inside its matrix
    const unsigned short* Lbase = (which == 2) ? x + col0 : dqkv + (which == 1 ? 2 * H : 0) + col0;
    const size_t ldl = (which == 2) ? (size_t)H : (size_t)ld;
    const int nslab = M / LWM_ROWS;            // host-checked: M % 32 == 0
    // this lane's share of a slab: row 8 wave + (lane >> 3), 16-byte chunk lane & 7 of every tile; of R, chunk 0 (t) or 1 (dt)
    const int prow = lane >> 3, chk = lane & 7;
    const int lrow = 8 * wave + prow;
    const int lds_off = lwm_tile_off(lrow, chk);
    constexpr int MYT = (TILES + 3) / 4;       // tiles of this wave: wave, wave + 4, ...
    f32x4 acc[MYT][4];
#pragma unroll
    for (int a = 0; a < MYT; ++a)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[a][d] = (f32x4){0, 0, 0, 0};
    // (macros, not lambdas over a struct: the slab registers must stay scalar-replaced, a by-reference aggregate goes to scratch)
#define LWM_FETCH(slab_, L_, R_)                                                                                      \
    do {                                                                                                              \
        const size_t m_ = (size_t)min((slab_), nslab - 1) * LWM_ROWS + lrow; /* slabs past the end re-read the last one and are not accumulated */ \
        const unsigned short* src_ = Lbase + m_ * ldl + chk * 8;                                                      \
        _Pragma("unroll") for (int tl = 0; tl < TILES; ++tl) L_[tl] = *(const bf16x8*)(src_ + 64 * tl);               \
        R_ = (chk == 1) ? *(const bf16x8*)(dt + m_ * ld_dt) : *(const bf16x8*)(t + m_ * 8);                             \
    } while (0)
    // one slab: registers -> LDS image (buffer kb), barrier, refill the registers with the slab three ahead, fragments + MFMAs
#define LWM_STEP(slab_, kb_, L_, R_)                                                                                  \
    do {                                                                                                              \
        char* b_ = lsm + (kb_) * BUF;                                                                                 \
        _Pragma("unroll") for (int tl = 0; tl < TILES; ++tl) *(bf16x8*)(b_ + tl * 4096 + lds_off) = L_[tl];           \
        if (chk < 2) *(bf16x8*)(b_ + TILES * 4096 + lds_off) = R_;                                                     \
        __syncthreads(); /* the image is complete; every wave has finished the reads of the image two slabs back (same buffer) */ \
        LWM_FETCH((slab_) + 3 * G, L_, R_);                                                                           \
        if ((slab_) < nslab) {                                                                                        \
            const bf16x8 rf_ = lwm_tr_frag(b_ + TILES * 4096, 0, lane);                                               \
            _Pragma("unroll") for (int a = 0; a < MYT; ++a) {                                                         \
                const int tl = wave + 4 * a;                                                                          \
                if (tl < TILES) {                                                                                     \
                    _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                     \
                        acc[a][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lwm_tr_frag(b_ + tl * 4096, d, lane), rf_, acc[a][d], 0, 0, 0); \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    const int first = blockIdx.x, G = gridDim.x;
    if (first >= nslab) return;
    bf16x8 l0[TILES], l1[TILES], l2[TILES], r0, r1, r2;   // three slabs in flight per lane (registers are the prefetch queue; the LDS image only transposes)
    LWM_FETCH(first, l0, r0);
    LWM_FETCH(first + G, l1, r1);
    LWM_FETCH(first + 2 * G, l2, r2);
    for (int slab = first; slab < nslab; slab += 6 * G) {   // six steps per trip: register set (mod 3) and LDS buffer (mod 2) are compile-time
        LWM_STEP(slab, 0, l0, r0);
        LWM_STEP(slab + G, 1, l1, r1);
        LWM_STEP(slab + 2 * G, 0, l2, r2);
        LWM_STEP(slab + 3 * G, 1, l0, r0);
        LWM_STEP(slab + 4 * G, 0, l1, r1);
        LWM_STEP(slab + 5 * G, 1, l2, r2);
    }
#undef LWM_STEP
#undef LWM_FETCH
    // lane (c = lane & 15, g = lane >> 4) holds Z[64 tl + 16 d + 4 g + r][c]
    const int c = lane & 15, g = lane >> 4;
    const bool use = (which == 0) ? (c < 4) : (which == 1) ? (c >= 4 && c < 8) : (c >= 8);
    if (use) {
#pragma unroll
        for (int a = 0; a < MYT; ++a) {
            const int tl = wave + 4 * a;
            if (tl < TILES) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = col0 + 64 * tl + 16 * d + 4 * g + r;
                        if (ws != nullptr) {
                            const size_t idx = (which == 0) ? (size_t)col * 4 + c
                                             : (which == 1) ? (size_t)4 * H + (size_t)col * 4 + (c - 4)
                                             : (c < 12)     ? (size_t)8 * H + (size_t)(c - 8) * H + col
                                                            : (size_t)12 * H + (size_t)(c - 12) * H + col;
                            ws[(size_t)blockIdx.x * 16 * H + idx] = acc[a][d][r];
                        } else {
                            float* dst = (which == 0) ? dB_q + (size_t)col * 4 + c
                                       : (which == 1) ? dB_v + (size_t)col * 4 + (c - 4)
                                       : (c < 12)     ? dA_q + (size_t)(c - 8) * H + col
                                                      : dA_v + (size_t)(c - 12) * H + col;
                            atomicAdd(dst, acc[a][d][r]);
                        }
                    }
            }
        }
    }
}

// ---- dt AND dB in one pass over dq, dv (large M): the backward needs dt = [dq B_q | dv B_v] (rank-8 operand of the QKV dgrad
// and of dA) before anything else, and used to get it from a skinny GEMM (N = 16) that streams dq, dv once — which the weight
// gradient kernel then streamed again.  Here a workgroup owns 32-token slabs and walks their dq, dv columns in 256-column chunks
// (registers -> [32][64] LDS tiles, four chunks in flight per lane): every chunk feeds (a) dB += chunk^T t (transposed
// fragments, as above) and (b) dt += chunk W_dt^T (row fragments against the [16, 2H] image of w_dt resident in LDS); at the end of
// a slab the four waves' dt partials are summed in a fixed order, rounded to bf16 and stored.  dq, dv cross the fabric once;
// results do not depend on the grid (dB: float atomics once per workgroup, as before).
template <int NCH>   // 256-column chunks per slab = 2H / 256 (even: H % 256 == 0)
__global__ __launch_bounds__(256) void lora_dt_db_kernel(const unsigned short* __restrict__ dqkv, int ld,
                                                         const unsigned short* __restrict__ t,
                                                         const unsigned short* __restrict__ w_dt,   // [16, 3H]
                                                         unsigned short* __restrict__ dt, int ld_dt, int M, int H,
                                                         float* __restrict__ dB_q, float* __restrict__ dB_v, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    constexpr int IMG = 4 * 4096;                 // one chunk: four [32][64] tiles
    char* img = lsm;                              // 2 x IMG
    char* rimg = lsm + 2 * IMG;                   // 2 x 4096: R tile (chunk 0 of its rows = t[m, 0:8])
    char* wimg = rimg + 2 * 4096;                 // 4 NCH tiles of [16][64]: w_dt's q then v segment
    float* red = (float*)(wimg + 4 * NCH * 2048); // [4 waves][2 row tiles][64 lanes][4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int nslab = M / LWM_ROWS;
    const int prow = lane >> 3, chk = lane & 7;
    const int lrow = 8 * wave + prow;
    const int lds_off = lwm_tile_off(lrow, chk);
    // w_dt image: tile T (64 columns of the q segment for T < 2 NCH, of the v segment after) row r at lwm_tile_off(r, chunk)
    for (int e = threadIdx.x; e < 4 * NCH * 16 * 8; e += 256) {
        const int T = e >> 7, r = (e >> 3) & 15, ch = e & 7;
        const int col = (T < 2 * NCH ? 0 : H) + 64 * T + 8 * ch;   // (v segment starts at 2H: T - 2 NCH tiles into it)
        *(bf16x8*)(wimg + T * 2048 + lwm_tile_off(r, ch)) = *(const bf16x8*)(w_dt + (size_t)r * 3 * H + col);
    }
    f32x4 accB[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) accB[c][d] = (f32x4){0, 0, 0, 0};
    f32x4 dtp[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}};
    const int first = blockIdx.x, G = gridDim.x;
    if (first >= nslab) return;
    constexpr int RING = 4;
    constexpr int LCM = (NCH % 4 == 0) ? NCH : (NCH % 2 == 0 ? 2 * NCH : 4 * NCH);   // steps after which ring slot, chunk and buffer repeat
    bf16x8 ring[RING][5];
    // chunk stream j of this workgroup: slab first + (j / NCH) G, chunk j % NCH (columns 256 c of [dq | dv])
#define LDB_FETCH(j_, slot_)                                                                                          \
    do {                                                                                                              \
        const int sj_ = min(first + ((j_) / NCH) * G, nslab - 1);   /* chunks past the end re-read the last slab and are not accumulated */ \
        const int cj_ = (j_) % NCH;                                                                                   \
        const size_t m_ = (size_t)sj_ * LWM_ROWS + lrow;                                                              \
        const unsigned short* src_ = dqkv + m_ * (size_t)ld + (2 * cj_ < NCH ? 0 : H) + 256 * cj_ + chk * 8;          \
        _Pragma("unroll") for (int tl = 0; tl < 4; ++tl) ring[slot_][tl] = *(const bf16x8*)(src_ + 64 * tl);          \
        ring[slot_][4] = *(const bf16x8*)(t + m_ * 8);                                                                \
    } while (0)
    int jbase = 0;
    LDB_FETCH(0, 0); LDB_FETCH(1, 1); LDB_FETCH(2, 2); LDB_FETCH(3, 3);
    const int nchunks = ((nslab - first + G - 1) / G) * NCH;
    __syncthreads();   // the w_dt image
    for (; jbase < nchunks; jbase += LCM) {
#pragma unroll
        for (int u = 0; u < LCM; ++u) {
            constexpr int dummy = 0; (void)dummy;
            const int j = jbase + u;
            const int slot = u % RING, c = u % NCH, kb = u & 1;   // compile-time after unrolling (LCM is a multiple of 4, NCH and 2)
            char* b = img + kb * IMG;
            char* rb = rimg + kb * 4096;
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) *(bf16x8*)(b + tl * 4096 + lds_off) = ring[slot][tl];
            if (chk == 0) *(bf16x8*)(rb + lds_off) = ring[slot][4];
            __syncthreads();   // images complete; every wave has finished the reads of the images two chunks back (same buffers)
            LDB_FETCH(j + RING, slot);
            if (j < nchunks) {
                const char* tile = b + wave * 4096;   // this wave's 64 columns of the chunk
                // (a) dB: Z[col][0:8] += tile^T t
                const bf16x8 rf = lwm_tr_frag(rb, 0, lane);
#pragma unroll
                for (int d = 0; d < 4; ++d) accB[c][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lwm_tr_frag(tile, d, lane), rf, accB[c][d], 0, 0, 0);
                // (b) dt[row][0:16] += tile (rows) . w_dt[:, these 64 columns]^T
                const char* wt = wimg + (4 * c + wave) * 2048;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 wf = *(const bf16x8*)(wt + lwm_tile_off(i, 4 * ks + g));
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        dtp[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tile + lwm_tile_off(16 * rt + i, 4 * ks + g)), wf, dtp[rt], 0, 0, 0);
                }
                if (c == NCH - 1) {   // the slab is complete: fixed-order sum of the four waves' partials, bf16, store
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        *(f32x4*)(red + ((wave * 2 + rt) * 64 + lane) * 4) = dtp[rt];
                        dtp[rt] = (f32x4){0, 0, 0, 0};
                    }
                }
            }
            if (c == NCH - 1) {
                __syncthreads();
                if (j < nchunks) {
                    const int row = threadIdx.x >> 3, cp = (threadIdx.x & 7) * 2;   // 32 rows x 16 columns, two columns per thread
                    const int rt = row >> 4, rl = row & 15;
                    float v0 = 0.f, v1 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        v0 += red[((w * 2 + rt) * 64 + (rl >> 2) * 16 + cp) * 4 + (rl & 3)];
                        v1 += red[((w * 2 + rt) * 64 + (rl >> 2) * 16 + cp + 1) * 4 + (rl & 3)];
                    }
                    const size_t m = (size_t)(first + (j / NCH) * G) * LWM_ROWS + row;
                    *(unsigned*)(dt + m * ld_dt + cp) = pack2bf(v0, v1);
                }
            }
        }
    }
#undef LDB_FETCH
    // lane (cc = lane & 15, g) holds Z[256 c + 64 wave + 16 d + 4 g + r][cc] of [dq | dv]
    const int cc = lane & 15;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool isv = 2 * c >= NCH;
        const bool use = isv ? (cc >= 4 && cc < 8) : (cc < 4);
        if (use) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 256 * c - (isv ? H : 0) + 64 * wave + 16 * d + 4 * g + r;
                    if (ws != nullptr) ws[(size_t)blockIdx.x * 16 * H + (isv ? (size_t)4 * H : 0) + (size_t)col * 4 + (isv ? cc - 4 : cc)] = accB[c][d][r];   // (see lora_wgrad_mfma_kernel)
                    else atomicAdd((isv ? dB_v : dB_q) + (size_t)col * 4 + (isv ? cc - 4 : cc), accB[c][d][r]);
                }
        }
    }
}

// dst[i] += sum over copies x < G of ws[x][i], copies added in a fixed order (deterministic: bit-reproducible gradients).  Element i of the copy layout
// [dB_q (4H) | dB_v (4H) | dA_q (4H) | dA_v (4H)]; the dB half was written by g_b workgroups, the dA half by g_a (0: that half is skipped).
__global__ __launch_bounds__(256) void lora_reduce_kernel(const float* __restrict__ ws, int H, int g_b, int g_a, float* __restrict__ dB_q,
                                                          float* __restrict__ dB_v, float* __restrict__ dA_q, float* __restrict__ dA_v) {
    // 64 elements per workgroup, wave w adds the copies x = w, w + 4, ... in four chains; the four waves' sums meet in LDS in a fixed order.
    // (Round 5, later: one element per THREAD with all G copies in sequence was 48 workgroups and 64 dependent loads deep — 32 us per layer
    // at every batch size, 2 % of the b = 256 step.)
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;               // 16 H is a multiple of 64, and so is 4 H: a workgroup stays inside one matrix
    const int which = i / (4 * H);
    const int G = which < 2 ? g_b : g_a;
    if (G <= 0) return;                              // (uniform per workgroup)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const size_t stride = (size_t)16 * H;
    int x = w;
    for (; x + 12 < G; x += 16) {
        s0 += ws[(size_t)x * stride + i];
        s1 += ws[(size_t)(x + 4) * stride + i];
        s2 += ws[(size_t)(x + 8) * stride + i];
        s3 += ws[(size_t)(x + 12) * stride + i];
    }
    for (; x < G; x += 4) s0 += ws[(size_t)x * stride + i];
    part[w][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0) {
        float* dst = which == 0 ? dB_q : which == 1 ? dB_v : which == 2 ? dA_q : dA_v;
        dst[i - which * 4 * H] += (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    }
}

// Standalone down-projection t[m, 0:8] = bf16(x[m, :] . a_cat[0:8, :]^T) (bf16 operands, fp32 accumulation) — the arithmetic the
// LayerNorm kernel fuses for the first rank-(4+4) slot.  Used for the SECOND slot of adapters with 4 < r <= 8 (the reference
// accepts any r > 0, image_encoder.py:50-53): a rare configuration, one wave per row, not tuned.
__global__ __launch_bounds__(256) void lora_down_proj_kernel(const unsigned short* __restrict__ x, int ld, const unsigned short* __restrict__ a_cat,
                                                             int M, int H, unsigned short* __restrict__ t) {
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += nwaves) {
        float tp[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tp[r] = 0.f;
        for (int c = 8 * lane; c < H; c += 512) {
            const bf16x8 xv = *(const bf16x8*)(x + (size_t)row * ld + c);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bf16x8 av = *(const bf16x8*)(a_cat + (size_t)r * H + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) tp[r] += bf2f((unsigned short)xv[e]) * bf2f((unsigned short)av[e]);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float sum = wave_sum(tp[r]);
            if (lane == r) t[(size_t)row * 8 + r] = f2bf(sum);
        }
    }
}

}  // namespace clibd

using namespace clibd;

extern "C" int clibd_lora_pack(const float* a_q, const float* a_v, const float* b_q, const float* b_v, int H,
                               void* v_fwd_bf16, void* v_bwd_bf16, void* a_cat_bf16, void* w_dt_bf16, void* stream) {
    if (!a_q || !a_v || !b_q || !b_v || !v_fwd_bf16 || !v_bwd_bf16 || !a_cat_bf16 || !w_dt_bf16)
        return set_error(CLIBD_EINVAL, "lora_pack: null pointer");
    if (H <= 0 || H % 64 != 0) return set_error(CLIBD_EINVAL, "lora_pack: H must be a multiple of 64");
    hipLaunchKernelGGL(lora_pack_kernel, dim3((3 * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, a_q, a_v, b_q, b_v, H,
                       (unsigned short*)v_fwd_bf16, (unsigned short*)v_bwd_bf16, (unsigned short*)a_cat_bf16,
                       (unsigned short*)w_dt_bf16);
    return check_launch("lora_pack");
}

static int lora_num_cus() {
    static const int n = [] {
        int dev = 0, c = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c = prop.multiProcessorCount;
        return c;
    }();
    return n;
}

// Partials workspace of the adapters' gradient kernels (ABI 3): one copy of [dB_q | dB_v | dA_q | dA_v] (16 H floats) per workgroup
// column of the MFMA forms; 0 for shapes that take the VALU kernel (ragged M), which keeps its per-block float atomics.
extern "C" size_t clibd_lora_workspace_bytes(int M, int H) {
    if (M <= 0 || H <= 0 || M % 32 != 0 || H % 128 != 0) return 0;
    const int g = lora_num_cus() < M / 32 ? lora_num_cus() : M / 32;
    return (size_t)g * 16 * (size_t)H * sizeof(float);
}

static int lora_reduce_launch(const float* ws, int H, int g_b, int g_a, float* dA_q, float* dA_v, float* dB_q, float* dB_v, hipStream_t st) {
    hipLaunchKernelGGL(lora_reduce_kernel, dim3((unsigned)(16 * H / 64)), dim3(256), 0, st, ws, H, g_b, g_a, dB_q, dB_v, dA_q, dA_v);
    return check_launch("lora_reduce");
}

extern "C" int clibd_lora_wgrad(const void* dqkv, int ld_dqkv, const void* x_bf16, const void* t_bf16, const void* dt_bf16,
                                int ld_dt, int M, int H, float* dA_q, float* dA_v, float* dB_q, float* dB_v,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!dqkv || !x_bf16 || !t_bf16 || !dt_bf16 || !dA_q || !dA_v || !dB_q || !dB_v)
        return set_error(CLIBD_EINVAL, "lora_wgrad: null pointer");
    if (M <= 0 || H <= 0 || H % 64 != 0 || H > 1024) return set_error(CLIBD_EINVAL, "lora_wgrad: H must be a multiple of 64, <= 1024");
    if (ld_dqkv < 3 * H || ld_dqkv % 8 || ld_dt < 8 || ld_dt % 8) return set_error(CLIBD_EINVAL, "lora_wgrad: bad leading dimension");
    if (!aligned16(dqkv) || !aligned16(x_bf16) || !aligned16(t_bf16) || !aligned16(dt_bf16))
        return set_error(CLIBD_EINVAL, "lora_wgrad: alignment");
    // large M: the MFMA form (needs whole 32-token slabs, 128-column segments and the dt rows at 16-byte pitch); round 6: with a partials
    // workspace ANY whole-slab M takes it (its sums are fixed-order; the VALU kernel below ends in float atomics), so that a step at the
    // small batches of the parity / fidelity tests (32 x 197 = 6 304 rows) is as bit-reproducible as one at the metric's batch
    if (M % LWM_ROWS == 0 && (M >= 8192 || workspace != nullptr) && H % 128 == 0 && (H / 128 == 3 || H / 128 == 4 || H / 128 == 6 || H / 128 == 8) && ld_dt % 8 == 0) {
        const int num_cus = lora_num_cus();
        const int tiles = H / 128;
        const int want = (2 * num_cus + 5) / 6;   // x 6 segments: two workgroups (2 x 56 KiB of LDS at H = 768) per CU
        const int groups = want < M / LWM_ROWS ? want : M / LWM_ROWS;
        const size_t ldsm = (size_t)2 * (tiles + 1) * 4096;
        // partials instead of float atomics when the caller brought the workspace (nullptr: the atomics of rounds 1-4)
        float* ws = nullptr;
        if (workspace != nullptr) {
            if (!aligned16(workspace) || workspace_bytes < (size_t)groups * 16 * (size_t)H * sizeof(float))
                return set_error(CLIBD_EINVAL, "lora_wgrad: workspace too small or misaligned (clibd_lora_workspace_bytes)");
            ws = (float*)workspace;
        }
#define LWM_LAUNCH(T)                                                                                                 \
    do {                                                                                                              \
        hipFuncSetAttribute((const void*)lora_wgrad_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);   \
        hipLaunchKernelGGL((lora_wgrad_mfma_kernel<T>), dim3(groups, 6), dim3(256), ldsm, (hipStream_t)stream,        \
                           (const unsigned short*)dqkv, ld_dqkv, (const unsigned short*)x_bf16, (const unsigned short*)t_bf16, \
                           (const unsigned short*)dt_bf16, ld_dt, M, H, dA_q, dA_v, dB_q, dB_v, 0, ws);               \
    } while (0)
        if (tiles == 3) LWM_LAUNCH(3);
        else if (tiles == 4) LWM_LAUNCH(4);
        else if (tiles == 6) LWM_LAUNCH(6);
        else LWM_LAUNCH(8);
#undef LWM_LAUNCH
        if (int e = check_launch("lora_wgrad")) return e;
        return ws ? lora_reduce_launch(ws, H, groups, groups, dA_q, dA_v, dB_q, dB_v, (hipStream_t)stream) : CLIBD_OK;
    }
    const int blocks = (M + LW_ROWS - 1) / LW_ROWS;
    const size_t lds = ((size_t)3 * (H / 4) * 16 + (size_t)4 * H + (size_t)(4 * H) / 32 + 32) * sizeof(float);
    hipLaunchKernelGGL(lora_wgrad_kernel, dim3(blocks), dim3(H), lds, (hipStream_t)stream, (const unsigned short*)dqkv, ld_dqkv,
                       (const unsigned short*)x_bf16, (const unsigned short*)t_bf16, (const unsigned short*)dt_bf16, ld_dt, M, H,
                       dA_q, dA_v, dB_q, dB_v);
    return check_launch("lora_wgrad");
}

extern "C" int clibd_gemm_bf16_nt_khole(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int hole_k0, int hole_len,
                                        const clibd_gemm_epilogue* ep, void* stream);

extern "C" int clibd_lora_backward(const void* dqkv, int ld_dqkv, const void* x_bf16, const void* t_bf16, const void* w_dt_bf16,
                                   void* dt_bf16, int ld_dt, int M, int H, float* dA_q, float* dA_v, float* dB_q, float* dB_v,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!dqkv || !x_bf16 || !t_bf16 || !w_dt_bf16 || !dt_bf16 || !dA_q || !dA_v || !dB_q || !dB_v)
        return set_error(CLIBD_EINVAL, "lora_backward: null pointer");
    if (M <= 0 || H <= 0 || H % 64 != 0 || H > 1024) return set_error(CLIBD_EINVAL, "lora_backward: H must be a multiple of 64, <= 1024");
    if (ld_dqkv < 3 * H || ld_dqkv % 8 || ld_dt < 16 || ld_dt % 8) return set_error(CLIBD_EINVAL, "lora_backward: bad leading dimension");
    if (!aligned16(dqkv) || !aligned16(x_bf16) || !aligned16(t_bf16) || !aligned16(w_dt_bf16) || !aligned16(dt_bf16))
        return set_error(CLIBD_EINVAL, "lora_backward: alignment");
    const int nch = 2 * H / 256;
    // fused: whole slabs, a 256-column chunk must not straddle dq | dv, and enough slabs per workgroup (>= 16 on a 256-CU part) to
    // amortise its w_dt image and its 6 k float atomics at the end (M = 50 432: 157 us fused against 91 us in two calls;
    // M = 403 456: 384 against 613 us)
    const bool fused = M % LWM_ROWS == 0 && M >= 131072 && H % 256 == 0 && (nch == 4 || nch == 6 || nch == 8);
    if (!fused) {   // small or ragged M: the skinny GEMM (k segment of dqkv skipped) and the VALU / per-segment weight-gradient kernel
        clibd_gemm_epilogue ep{};
        ep.out_bf16 = dt_bf16;
        ep.ld_out_bf16 = ld_dt;
        ep.split_k = 1;
        if (int e = clibd_gemm_bf16_nt_khole(dqkv, ld_dqkv, w_dt_bf16, 3 * H, M, 16, 3 * H, H, H, &ep, stream)) return e;
        return clibd_lora_wgrad(dqkv, ld_dqkv, x_bf16, t_bf16, dt_bf16, ld_dt, M, H, dA_q, dA_v, dB_q, dB_v, workspace, workspace_bytes, stream);
    }
    const int num_cus = lora_num_cus();
    const int nslab = M / LWM_ROWS;
    float* ws = nullptr;
    if (workspace != nullptr) {
        if (!aligned16(workspace) || workspace_bytes < clibd_lora_workspace_bytes(M, H))
            return set_error(CLIBD_EINVAL, "lora_backward: workspace too small or misaligned (clibd_lora_workspace_bytes)");
        ws = (float*)workspace;
    }
    int g_b = 0, g_a = 0;
    {   // dt and dB: one workgroup per CU
        const int groups = num_cus < nslab ? num_cus : nslab;
        g_b = groups;
        const size_t lds1 = (size_t)2 * 4 * 4096 + 2 * 4096 + (size_t)4 * nch * 2048 + 4 * 2 * 64 * 16;
#define LDB_LAUNCH(N_)                                                                                                \
    do {                                                                                                              \
        hipFuncSetAttribute((const void*)lora_dt_db_kernel<N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);       \
        hipLaunchKernelGGL((lora_dt_db_kernel<N_>), dim3(groups), dim3(256), lds1, (hipStream_t)stream,              \
                           (const unsigned short*)dqkv, ld_dqkv, (const unsigned short*)t_bf16, (const unsigned short*)w_dt_bf16, \
                           (unsigned short*)dt_bf16, ld_dt, M, H, dB_q, dB_v, ws);                                    \
    } while (0)
        if (nch == 4) LDB_LAUNCH(4);
        else if (nch == 6) LDB_LAUNCH(6);
        else LDB_LAUNCH(8);
#undef LDB_LAUNCH
        if (int e = check_launch("lora_backward (dt, dB)")) return e;
    }
    {   // dA from x and the finished dt: the x segments of the per-segment kernel
        const int tiles = H / 128;
        const int want = (2 * num_cus + 1) / 2;
        const int groups = want < nslab ? want : nslab;
        g_a = groups;
        const size_t ldsm = (size_t)2 * (tiles + 1) * 4096;
#define LWM_LAUNCH2(T)                                                                                                \
    do {                                                                                                              \
        hipFuncSetAttribute((const void*)lora_wgrad_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);   \
        hipLaunchKernelGGL((lora_wgrad_mfma_kernel<T>), dim3(groups, 2), dim3(256), ldsm, (hipStream_t)stream,        \
                           (const unsigned short*)dqkv, ld_dqkv, (const unsigned short*)x_bf16, (const unsigned short*)t_bf16, \
                           (const unsigned short*)dt_bf16, ld_dt, M, H, dA_q, dA_v, dB_q, dB_v, 2, ws);               \
    } while (0)
        if (tiles == 3) LWM_LAUNCH2(3);
        else if (tiles == 4) LWM_LAUNCH2(4);
        else if (tiles == 6) LWM_LAUNCH2(6);
        else LWM_LAUNCH2(8);
#undef LWM_LAUNCH2
    }
    if (int e = check_launch("lora_backward (dA)")) return e;
    return ws ? lora_reduce_launch(ws, H, g_b, g_a, dA_q, dA_v, dB_q, dB_v, (hipStream_t)stream) : CLIBD_OK;
}

extern "C" int clibd_lora_down_proj(const void* x_bf16, int ld_x, const void* a_cat_bf16, int M, int H, void* t_bf16, void* stream) {
    if (!x_bf16 || !a_cat_bf16 || !t_bf16) return set_error(CLIBD_EINVAL, "lora_down_proj: null pointer");
    if (M <= 0 || H <= 0 || H % 8 != 0 || ld_x < H || ld_x % 8 != 0) return set_error(CLIBD_EINVAL, "lora_down_proj: bad shape");
    if (!aligned16(x_bf16) || !aligned16(a_cat_bf16) || !aligned16(t_bf16)) return set_error(CLIBD_EINVAL, "lora_down_proj: alignment");
    int blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lora_down_proj_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_bf16, ld_x,
                       (const unsigned short*)a_cat_bf16, M, H, (unsigned short*)t_bf16);
    return check_launch("lora_down_proj");
}
