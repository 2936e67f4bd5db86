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
//
// Persistent tiles: the grid is one workgroup per CU; each workgroup walks tiles id, id+grid, ...  Before the epilogue of
// tile i it already issues the first six half-tiles of tile i+1 (LDS is idle during the epilogue), so the next tile's
// HBM/L2 latency and this tile's output stores overlap; vmcnt counts stores too (in issue order), so the first four
// phases of a tile wait with vmcnt(8 + stores of the previous epilogue) instead of vmcnt(8).  Workgroups start with a
// small per-XCD-slot time skew so the chip's store bursts do not line up.
#include "gemm_common.h"
#include "host_util.h"

namespace clibd {

constexpr int T_M = 256, T_N = 256, T_K = 64;
constexpr int HALF_BYTES = 128 * T_K * 2;      // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // 64 KiB
constexpr int G256_THREADS = 512;
constexpr int G256_LDS = 2 * STAGE_BYTES;      // 128 KiB

#define CLIBD_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(G256_THREADS) void gemm256_bf16_nt_kernel(GemmParams p, int ntiles, int skew_ticks, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nk = p.K / T_K;  // even, >= 2 (host-checked)
    int tile = blockIdx.x;
    int m0, n0;          // tile being computed
    int nm0 = 0, nn0 = 0;  // tile whose loads are being issued (== m0,n0 until the last issue of the current tile)

    // ---- per-lane LDS-DMA sources: half-tile type j (0 A_even, 1 W_hn0, 2 W_hn1, 3 A_odd) x this wave's 2 pieces
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    // Sources are NOT kept as pointers (8 x 64-bit per lane spilled): every half-tile piece of this lane derives from two
    // tile-local rows + constants, and the 32-bit byte offset is rebuilt at issue time (3 VALU per LDS-DMA):
    //   piece i (0/1) of A_even: tile row rowA + 8i        A_odd: + 64
    //   piece i       of W_hn0 : tile col rowW + 32i       W_hn1: + 8
    // (operands are < 4 GiB, host-checked, so base + 32-bit offset addresses them)
    const int r0 = 16 * wave + prow;                              // LDS row of piece 0 inside a half-tile
    const int rowA = (r0 & 63) + 128 * (r0 >> 6);
    const int rowW = 64 * (wave >> 1) + w_col_of(wave & 1, prow);  // w_col_of(tq, iq): iq = prow (piece 0)
    const unsigned chunk16 = (unsigned)chunk * 16u;
    const unsigned lda2 = (unsigned)p.lda * 2u, ldw2 = (unsigned)p.ldw * 2u;
    const char* const baseA = (const char*)p.A;
    const char* const baseW = (const char*)p.W;
    auto set_sources = [&](int tile_id) {
        int tm, tn;
        tile_coords(tile_id, p.tiles_m, p.tiles_n, 4, tm, tn);
        nm0 = tm * T_M;
        nn0 = tn * T_N;
    };
    // issue half-tile j (0 A_even, 1 W_hn0, 2 W_hn1, 3 A_odd) of K-tile u into stage (u & 1)
#define ISSUE(u, j, stage)                                                                              \
    do {                                                                                                \
        const unsigned koff_ = (unsigned)(u) * (T_K * 2) + chunk16;                                     \
        char* dst_ = smem + (stage) * STAGE_BYTES + (j) * HALF_BYTES + (2 * wave) * 1024;               \
        int ra_ = rowA, rw_ = rowW;                                                                     \
        asm volatile("" : "+v"(ra_), "+v"(rw_)); /* opaque: keeps the 8 per-tile offsets from being hoisted into (spilled) registers */ \
        if ((j) == 0 || (j) == 3) {                                                                     \
            const int c_ = nm0 + ((j) == 3 ? 64 : 0);                                                   \
            const unsigned g0_ = (unsigned)min(c_ + ra_, p.M - 1), g1_ = (unsigned)min(c_ + ra_ + 8, p.M - 1); \
            glds16(baseA + (g0_ * lda2 + koff_), dst_);                                                 \
            glds16(baseA + (g1_ * lda2 + koff_), dst_ + 1024);                                          \
        } else {                                                                                        \
            const int c_ = nn0 + ((j) == 2 ? 8 : 0);                                                    \
            const unsigned g0_ = (unsigned)min(c_ + rw_, p.N - 1), g1_ = (unsigned)min(c_ + rw_ + 32, p.N - 1); \
            glds16(baseW + (g0_ * ldw2 + koff_), dst_);                                                 \
            glds16(baseW + (g1_ * ldw2 + koff_), dst_ + 1024);                                          \
        }                                                                                               \
    } while (0)

    // ---- fragment read offsets inside a half-tile
    const int frow = lane & 15, fch = lane >> 4;
    int a_off[4], w_off[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_off[t] = tile_off(64 * wm + 16 * t + frow, fch);
#pragma unroll
    for (int t = 0; t < 2; ++t) w_off[t] = tile_off(32 * wn + 16 * t + frow, fch);

    f32x4 acc[2][4][2][2];  // [hm][mt][hn][nt]
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

    // wait at the end of a load segment in the first four phases of a tile (and before its phase 0): the previous
    // epilogue's stores sit between the prologue LDS-DMA and this tile's later issues in the in-order vmcnt queue
#define WAIT_HEAD()                                         \
    do {                                                    \
        if (allow_case == 0) CLIBD_WAIT_VMCNT(8);           \
        else if (allow_case == 1) CLIBD_WAIT_VMCNT(24);     \
        else if (allow_case == 2) CLIBD_WAIT_VMCNT(40);     \
        else if (allow_case == 3) CLIBD_WAIT_VMCNT(56);     \
        else CLIBD_WAIT_VMCNT(63);                          \
    } while (0)
#define PHASE_HEAD(Q8)                                                                                       \
    do {                                                                                                     \
        constexpr int dl_ = (Q8) & 3;                                                                        \
        if (dl_ == 0) { LOAD_W(w0F, 0, 1); LOAD_A(0, 0); }                                                   \
        else if (dl_ == 1) { LOAD_W(w1F, 0, 2); }                                                            \
        else if (dl_ == 2) { LOAD_A(0, 3); }                                                                 \
        {                                                                                                    \
            constexpr int ju_ = ((Q8) + 2) & 3;                                                              \
            constexpr int du_ = ((Q8) + 6) >> 2;                                                             \
            ISSUE(du_, ju_, du_ & 1);                                                                        \
        }                                                                                                    \
        WAIT_HEAD();                                                                                         \
        BARRIER();                                                                                           \
        if (dl_ == 0) MMA(0, 0, w0F);                                                                        \
        else if (dl_ == 1) MMA(0, 1, w1F);                                                                   \
        else if (dl_ == 2) MMA(1, 1, w1F);                                                                   \
        else MMA(1, 0, w0F);                                                                                 \
        BARRIER();                                                                                           \
    } while (0)
#define PROLOGUE_ISSUE()                                                     \
    do {                                                                     \
        ISSUE(0, 0, 0); ISSUE(0, 1, 0); ISSUE(0, 2, 0); ISSUE(0, 3, 0);      \
        ISSUE(1, 0, 1); ISSUE(1, 1, 1);                                      \
    } while (0)

    // ---- start-up skew (see header) and the first tile's prologue: L_0 .. L_5 = K-tile 0 + A_even, W_hn0 of K-tile 1
    if (skew_ticks > 0) {
        const int cls = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)cls * skew_ticks) __builtin_amdgcn_s_sleep(8);
    }
    set_sources(tile);
    PROLOGUE_ISSUE();
    int allow_case = 0;
    const clibd_gemm_epilogue& ep = p.ep;
    const int stores_case = (ep.out_pre_bf16 ? 1 : 0) + (ep.out_bf16 ? 1 : 0) + (ep.out_f32 ? 2 : 0);  // (16-B stores per row) / 2

    // diagnostic time stamps (only when a stamp buffer is installed: clibd_debug_set_gemm_stamps): [wg][tile_i][8]
    int tile_i = 0;
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        if (stamps != nullptr && tile_i < 16 && (wave == 0 || wave == 4) && lane == 0)                    \
            stamps[(((size_t)blockIdx.x * 16 + tile_i) * 2 + (wave >> 2)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

    while (true) {
        m0 = nm0;
        n0 = nn0;
        const int next = tile + (int)gridDim.x;  // static round-robin: tile ids of one workgroup stay on one XCD
        const bool has_next = next < ntiles;
        WAIT_HEAD();  // L_0, L_1 of this tile have landed (this wave's pieces)
        BARRIER();
        if (wm == 1) BARRIER();  // stagger: group 1 runs one barrier interval behind group 0
        STAMP(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int kt = 0;
        if (nk >= 4) {
            PHASE_HEAD(0); PHASE_HEAD(1); PHASE_HEAD(2); PHASE_HEAD(3);
            STAMP(1);
            PHASE(4, true, 8); PHASE(5, true, 8); PHASE(6, true, 8); PHASE(7, true, 8);
            STAMP(2);
            for (kt = 2; kt < nk - 2; kt += 2) {
                PHASE(0, true, 8); PHASE(1, true, 8); PHASE(2, true, 8); PHASE(3, true, 8);
                PHASE(4, true, 8); PHASE(5, true, 8); PHASE(6, true, 8); PHASE(7, true, 8);
            }
        }
        // last iteration (kt = nk-2): only L_{4nk-2}, L_{4nk-1} are left to issue; drain with exact counts
        STAMP(3);
        PHASE(0, true, 8); PHASE(1, true, 8);
        if (has_next) set_sources(next);  // no more issues for this tile: the source registers now describe the next one
        PHASE(2, false, 6); PHASE(3, false, 4);
        PHASE(4, false, 2); PHASE(5, false, 0); PHASE(6, false, 0); PHASE(7, false, 0);
        if (wm == 0) BARRIER();  // group 0 matches group 1's extra barrier; every LDS read of this tile is complete
        STAMP(4);
        if (has_next) PROLOGUE_ISSUE();  // next tile's first six half-tiles fly while this tile's epilogue runs

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
        {
            const int nb = n0 + 64 * wn + 16 * fch;  // < N: N % 256 == 0 (host-checked)
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
        STAMP(5);
        ++tile_i;
        if (!has_next) break;
        // a full tile issued exactly 8 rows x (2*stores_case) stores per lane; a ragged one fewer: be conservative there
        allow_case = (m0 + T_M <= p.M) ? stores_case : 0;
        tile = next;
    }
}

static long long* g_stamp_buffer = nullptr;  // diagnostic only (tools/), never set on the product path

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
    static const int num_cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        }
        return n;
    }();
    GemmParams q = p;
    q.tiles_m = (p.M + T_M - 1) / T_M;
    q.tiles_n = p.N / T_N;
    const int grid = (int)(tiles < num_cus ? tiles : num_cus);
    // start-up skew between the 4 classes of workgroups: a quarter of a tile's duration each (s_memrealtime = 100 MHz);
    // only worth it when a workgroup runs several tiles
    const float tile_us = 1.4f * nk + 5.0f;
    // Measured (tools/gemm_stamps.py): the epilogue of a 256x256 bf16 tile takes ~10-13k cycles = the per-CU store-path
    // rate (~10.7 B/clk/CU), not a chip-wide burst: neither a start-up skew, a dynamic tile queue nor full-line-coalesced
    // stores (lane exchange) shortened it, so the schedule stays static and the skew is off.
    const int skew_ticks = 0;
    (void)tile_us;
    hipLaunchKernelGGL(gemm256_bf16_nt_kernel, dim3((unsigned)grid), dim3(G256_THREADS), G256_LDS, stream, q, (int)tiles, skew_ticks, g_stamp_buffer);
    return true;
}

}  // namespace clibd

// Diagnostic hook for tools/: device buffer of [256 workgroups][16 tiles][2 wave groups][8] int64 s_memtime stamps.
extern "C" void clibd_debug_set_gemm_stamps(void* device_buffer) { clibd::g_stamp_buffer = (long long*)device_buffer; }
