// bf16 MFMA GEMM, 256x256x64 tile, 8 waves, one workgroup per CU, 8-phase software pipeline (gfx950).
//
//   out = epilogue(A[M,K] · W[N,K]^T), same contract and epilogue as gemm.hip (gemm_common.h).
//
// Geometry: wave (wm, wn) = (wave>>2, wave&3) owns a 128(M) x 64(N) sub-tile = 8 x 4 v_mfma_f32_16x16x32_bf16 tiles
// (128 accumulator VGPRs).  A K-tile (BK = 64) is staged as FOUR 16-KiB half-tiles, cut along the *phase* structure
// rather than along the wave grid:
//     A_even : tile rows {128wm + 0..63}     (the "hm0" m-half of every wave)      needed at phase 0 of the K-tile
//     W_hn0  : tile cols {64wn + n-tiles 0,1} (the "hn0" n-half of every wave)      needed at phase 0
//     W_hn1  : tile cols {64wn + n-tiles 2,3}                                        needed at phase 1
//     A_odd  : tile rows {128wm + 64..127}                                           needed at phase 2
// and a K-tile is computed as 4 phases of 16 MFMAs per wave — quadrants (hm0,hn0) (hm0,hn1) (hm1,hn1) (hm1,hn0) — so
// each phase loads at most one new operand half into registers (12 / 4 / 8 / 0 ds_read_b128) and every half-tile is
// dead in LDS right after the phase that read it.  With 2 stages (8 half-tile slots, 128 KiB) that early death lets
// half-tile L_i be issued SIX phases before it is needed:   at phase p issue L_{p+6};  L_{4t+j} = half-tile j of K-tile t.
//
// Synchronisation (LDS-DMA data is ordered for a ds_read only by the issuing wave's counted vmcnt + a barrier):
//   * at the end of every load segment: s_waitcnt vmcnt(8)  (4 half-tiles x 2 pieces may stay in flight) => L_{<=p+2}
//     has landed for this wave's pieces; the barrier that ends the segment publishes it; phase p+1 reads it.
//   * WAR: slot of L_i is re-filled by L_{i+8}, issued >= 2 phases after the last read of L_i.
//   * two wave groups (wm = 0 / 1) run staggered by one barrier: while one group issues MFMAs the other is in its
//     load segment (ds_read + LDS-DMA issue), so the matrix pipe and the LDS/VMEM paths overlap inside one workgroup.
#include "gemm_common.h"
#include "host_util.h"

namespace clibd {

constexpr int T_M = 256, T_N = 256, T_K = 64;
constexpr int HALF_BYTES = 128 * T_K * 2;      // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // 64 KiB
constexpr int G256_THREADS = 512;
constexpr int G256_LDS = 2 * STAGE_BYTES;      // 128 KiB

#define CLIBD_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(G256_THREADS) void gemm256_bf16_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tile_m, tile_n;
    tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, 4, tile_m, tile_n);
    const int m0 = tile_m * T_M, n0 = tile_n * T_N;
    const int nk = p.K / T_K;  // even, >= 2 (host-checked)

    // ---- per-lane LDS-DMA sources: half-tile type j (0 A_even, 1 W_hn0, 2 W_hn1, 3 A_odd) x this wave's 2 pieces
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    // 32-bit byte offsets from the (wave-uniform) matrix bases: half the registers of 64-bit pointers, and the
    // LDS-DMA can use the saddr + voffset addressing form (operands are < 4 GiB, host-checked)
    unsigned src[4][2];
    const char* const baseA = (const char*)p.A;
    const char* const baseW = (const char*)p.W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 8 + prow;      // LDS row inside the half-tile, 0..127
        const int ra = (r & 63) + 128 * (r >> 6);     // A: wave-half (r>>6) -> tile rows 128*wm' + (r&63)
        const int gm_e = min(m0 + ra, p.M - 1);
        const int gm_o = min(m0 + ra + 64, p.M - 1);
        const int wq = r >> 5, tq = (r >> 4) & 1, iq = r & 15;  // W: wave-quarter, n-tile inside the half, MFMA row
        const int gn_0 = min(n0 + 64 * wq + w_col_of(tq, iq), p.N - 1);
        const int gn_1 = min(n0 + 64 * wq + w_col_of(2 + tq, iq), p.N - 1);
        src[0][i] = (unsigned)gm_e * (unsigned)(p.lda * 2) + chunk * 16;
        src[3][i] = (unsigned)gm_o * (unsigned)(p.lda * 2) + chunk * 16;
        src[1][i] = (unsigned)gn_0 * (unsigned)(p.ldw * 2) + chunk * 16;
        src[2][i] = (unsigned)gn_1 * (unsigned)(p.ldw * 2) + chunk * 16;
    }
    // issue half-tile j of K-tile u into stage (u & 1)
#define ISSUE(u, j, stage)                                                                              \
    do {                                                                                                \
        const unsigned koff_ = (unsigned)(u) * (T_K * 2);                                               \
        const char* const base_ = ((j) == 0 || (j) == 3) ? baseA : baseW;                               \
        char* dst_ = smem + (stage) * STAGE_BYTES + (j) * HALF_BYTES + (2 * wave) * 1024;               \
        glds16(base_ + (src[j][0] + koff_), dst_);                                                      \
        glds16(base_ + (src[j][1] + koff_), dst_ + 1024);                                               \
    } while (0)

    // ---- fragment read offsets inside a half-tile
    const int frow = lane & 15, fch = lane >> 4;
    int a_off[4], w_off[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_off[t] = tile_off(64 * wm + 16 * t + frow, fch);
#pragma unroll
    for (int t = 0; t < 2; ++t) w_off[t] = tile_off(32 * wn + 16 * t + frow, fch);

    f32x4 acc[2][4][2][2];  // [hm][mt][hn][nt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 aF[4][2], w0F[2][2], w1F[2][2];  // [tile][kk]

#define LOAD_A(stage, j)                                                                                     \
    do {                                                                                                     \
        const char* b_ = smem + (stage) * STAGE_BYTES + (j) * HALF_BYTES;                                    \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                      \
            aF[t][0] = *(const bf16x8*)(b_ + a_off[t]);                                                      \
            aF[t][1] = *(const bf16x8*)(b_ + (a_off[t] ^ 64));                                               \
        }                                                                                                    \
    } while (0)
#define LOAD_W(dstF, stage, j)                                                                               \
    do {                                                                                                     \
        const char* b_ = smem + (stage) * STAGE_BYTES + (j) * HALF_BYTES;                                    \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                      \
            dstF[t][0] = *(const bf16x8*)(b_ + w_off[t]);                                                    \
            dstF[t][1] = *(const bf16x8*)(b_ + (w_off[t] ^ 64));                                             \
        }                                                                                                    \
    } while (0)
#define MMA(hm, hn, wF)                                                                                      \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                    acc[hm][t][hn][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wF[n][kk], aF[t][kk], acc[hm][t][hn][n], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
#define BARRIER()                                   \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");              \
    } while (0)

    // One phase.  Q8 = phase index inside the 2-K-tile iteration (compile time), kt = first K-tile of the iteration.
    // issue_ok: whether L_{p+6} exists;  WAITN: vmcnt literal for the end of the load segment.
#define PHASE(Q8, issue_ok, WAITN)                                                                           \
    do {                                                                                                     \
        constexpr int st_ = ((Q8) >> 2) & 1;          /* stage being read: K-tile kt + (Q8>>2), kt even */   \
        constexpr int dl_ = (Q8) & 3;                                                                        \
        if (dl_ == 0) { LOAD_W(w0F, st_, 1); LOAD_A(st_, 0); }                                               \
        else if (dl_ == 1) { LOAD_W(w1F, st_, 2); }                                                          \
        else if (dl_ == 2) { LOAD_A(st_, 3); }                                                               \
        if (issue_ok) {                                                                                      \
            constexpr int ju_ = ((Q8) + 2) & 3;                                                              \
            constexpr int du_ = ((Q8) + 6) >> 2;                                                             \
            ISSUE(kt + du_, ju_, du_ & 1);                                                                   \
        }                                                                                                    \
        CLIBD_WAIT_VMCNT(WAITN);                                                                             \
        BARRIER();                                                                                           \
        if (dl_ == 0) MMA(0, 0, w0F);                                                                        \
        else if (dl_ == 1) MMA(0, 1, w1F);                                                                   \
        else if (dl_ == 2) MMA(1, 1, w1F);                                                                   \
        else MMA(1, 0, w0F);                                                                                 \
        BARRIER();                                                                                           \
    } while (0)

    // ---- prologue: L_0 .. L_5 = the whole K-tile 0 + A_even, W_hn0 of K-tile 1
    ISSUE(0, 0, 0); ISSUE(0, 1, 0); ISSUE(0, 2, 0); ISSUE(0, 3, 0);
    ISSUE(1, 0, 1); ISSUE(1, 1, 1);
    CLIBD_WAIT_VMCNT(8);  // L_0, L_1 landed (this wave's pieces)
    BARRIER();
    if (wm == 1) BARRIER();  // stagger: group 1 runs one barrier interval behind group 0

    int kt = 0;
    for (; kt < nk - 2; kt += 2) {
        PHASE(0, true, 8); PHASE(1, true, 8); PHASE(2, true, 8); PHASE(3, true, 8);
        PHASE(4, true, 8); PHASE(5, true, 8); PHASE(6, true, 8); PHASE(7, true, 8);
    }
    // last iteration (kt = nk-2): only L_{4nk-2}, L_{4nk-1} are left to issue; drain with exact counts
    PHASE(0, true, 8); PHASE(1, true, 8); PHASE(2, false, 6); PHASE(3, false, 4);
    PHASE(4, false, 2); PHASE(5, false, 0); PHASE(6, false, 0); PHASE(7, false, 0);
    if (wm == 0) BARRIER();  // group 0 matches group 1's extra barrier

    const clibd_gemm_epilogue& ep = p.ep;
    // ---- LoRA rank-8 update: one extra zero-padded k-step (lanes with k-chunk 0 carry U[m,0:8] / V[n,0:8])
    if (ep.rank_u != nullptr) {
        bf16x8 vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vf[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (fch == 0) {
                const int gn = min(n0 + 64 * wn + w_col_of(t, frow), p.N - 1);
                vf[t] = *(const bf16x8*)((const unsigned short*)ep.rank_v + (size_t)gn * 8);
            }
        }
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 uf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (fch == 0) {
                    const int gm = min(m0 + 128 * wm + 64 * hm + 16 * t + frow, p.M - 1);
                    uf = *(const bf16x8*)((const unsigned short*)ep.rank_u + (size_t)gm * ep.ld_rank_u);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[hm][t][n >> 1][n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[n], uf, acc[hm][t][n >> 1][n & 1], 0, 0, 0);
            }
    }

    // ---- epilogue: lane owns rows m0 + 128wm + 64hm + 16t + (lane&15), columns nb .. nb+15 (e = 4*ntile + reg)
    const int nb = n0 + 64 * wn + 16 * fch;
    if (nb >= p.N) return;
    float bias[16];
    load_bias16(ep, nb, true, bias);
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = m0 + 128 * wm + 64 * hm + 16 * t + frow;
            if (m >= p.M) continue;
            float v[16];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * n + r] = acc[hm][t][n >> 1][n & 1][r] + bias[4 * n + r];
            store_row16(ep, m, nb, v);
        }
}

bool gemm256_try_launch(const GemmParams& p, hipStream_t stream) {
    const int nk = p.K / T_K;
    if (p.K % T_K != 0 || nk < 2 || (nk & 1)) return false;
    if (p.N % T_N != 0) return false;
    if (p.ep.split_k > 1) return false;
    const long long tiles = (long long)((p.M + T_M - 1) / T_M) * (p.N / T_N);
    if (p.M < 1024 || tiles < 128) return false;
    if (p.N < 1024 && p.K < 1536) return false;     // short-K, narrow-N (e.g. 768x768 projections): HBM-bound, two 128^2 blocks per CU overlap better (measured)
    if ((unsigned long long)p.M * p.lda * 2ull >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2ull >= (1ull << 32)) return false;  // too few 256x256 tiles to fill 256 CUs: the 128x128 kernel wins
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)gemm256_bf16_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) ==
               hipSuccess;
    }();
    if (!attr_ok) return false;
    GemmParams q = p;
    q.tiles_m = (p.M + T_M - 1) / T_M;
    q.tiles_n = p.N / T_N;
    hipLaunchKernelGGL(gemm256_bf16_nt_kernel, dim3((unsigned)tiles), dim3(G256_THREADS), G256_LDS, stream, q);
    return true;
}

}  // namespace clibd
