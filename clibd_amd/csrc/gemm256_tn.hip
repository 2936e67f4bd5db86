// bf16 MFMA GEMM contracting over the ROWS of both operands ("TN"): C[Na, Nb] = A[M, Na]^T · B[M, Nb], fp32, split over M.
//
// This is the weight gradient of a linear layer in full fine-tune mode, dW[N, K] = dY[M, N]^T · X[M, K], read IN PLACE: the
// NT kernel (gemm256.hip) needs both operands transposed first (two extra passes over dY and X per layer and matrix:
// 9.7 of 77 ms of kernel time per step at b=256).  Same 256x256 tile, 8 waves, 8-phase schedule and persistent work items as
// gemm256.hip (read its header first; phase / wait / barrier structure is identical and not re-explained here); what differs:
//
//   * Operand roles.  P (MFMA column operand) = B columns = output columns; Q (row operand) = A columns = output rows.  The
//     k index of the MFMA is the token row m.  A K-tile is 64 rows of m.
//   * LDS image.  A half-tile is 64 k-rows x 128 columns (256 bytes per k-row), filled by 16 LDS-DMA instructions (two per
//     wave): instruction n (= "chunk" n) fetches the four k-rows n, n+16, n+32, n+48, 256 contiguous bytes each (lane ->
//     row lane>>4, 16-byte piece lane&15), and lands as 1 KiB at chunk base n * 1056 — 32 bytes of skew per chunk.
//   * Fragments come from ds_read_b64_tr_b16 (the hardware's 4 x 16 transposing read): lane (g = lane>>4, q = (lane&15)>>2,
//     p = lane&3) addresses k-row 16 a + (4g + q), columns 16 t + 4p..4p+3 and receives column 16 t + (lane&15) of k-rows
//     16 a + 4g .. +3; two such reads (a = 2kk, 2kk+1) make one 8-deep k fragment.  k-rows 4g + q for q = 0..3 live in four
//     different chunks, whose 1056-byte pitch puts them in four different 32-byte bank windows (and g, g+1 128 bytes apart):
//     conflict-free without an XOR swizzle, so tile t, k-step kk and half-tile j are all IMMEDIATE offsets of ONE per-lane
//     address per operand and stage (4 address VGPRs; the NT kernel needs 8).  Both operands use the same k permutation.
//   * P half hm = columns [128 hm, 128 hm + 128) of the tile, wave (wm, .) owns 64 wm + 16 t (t = 0..3) of it; Q half hn =
//     rows-of-output [128 hn, +128), wave (., wn) owns 32 wn + 16 n (n = 0, 1): every half-tile is contiguous in memory.
//   * Epilogue: fp32 partial tile of this (tile, M-slice) work item, dword stores (a lane holds one column of four rows);
//     the slices are summed by reduce_splits_kernel (gemm.hip), as for the NT split-K mode.  The next item's loads are issued
//     after the stores have drained (128 stores per lane exceed what vmcnt can count past).
//   * CS: colsum[n] += sum_m A[m, n] (the bias gradient db = dY^T 1) as one more MFMA per Q fragment against an all-ones P
//     fragment, in the two phases of a K-tile that touch P half 0 (each Q half exactly once): +8 MFMAs per wave and K-tile
//     (6 %), no extra pass over dY (a separate column-sum kernel cost 7.3 ms per step).  Work items of output-column tile 0
//     add their slice's sums atomically.
//
// M % 128 == 0, Na % 256 == 0, Nb % 256 == 0, every slice >= 4 K-tiles.
#include "gemm_common.h"
#include "host_util.h"

namespace clibd {

constexpr int TN_CHUNK = 1056;            // one LDS-DMA instruction's 1 KiB + 32 B of bank skew
constexpr int TN_HALF = 16 * TN_CHUNK;    // 16896 B
constexpr int TN_STAGE = 4 * TN_HALF;     // 67584 B
constexpr int TN_LDS = 2 * TN_STAGE;      // 135168 B
constexpr int TN_THREADS = 512;

struct TnParams {
    const unsigned short* A;   // [M, lda]  -> output rows
    const unsigned short* B;   // [M, ldb]  -> output columns
    int M, Na, Nb, lda, ldb;
    int tiles_a, tiles_b;      // Na / 256, Nb / 256
    int splits, nk_split;      // K-tiles (64 rows) per slice; even, >= 4
    float* partials;           // [splits][Na][Nb]
    long long split_stride;    // Na * Nb
    float* colsum;             // optional [Na]: += column sums of A (the bias gradient rides along, see CS below)
};

typedef int i32x2 __attribute__((ext_vector_type(2)));

#define TN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool CS>
__global__ __launch_bounds__(TN_THREADS) void gemm256_tn_kernel(TnParams p, int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk_total = p.M / 64;
    const int tiles_out = p.tiles_a * p.tiles_b;

    int a0 = 0, b0 = 0, kb_issue = 0, nk_issue = 0, split_issue = 0;
    bool first_col_tile = false;
    const unsigned ldA2 = (unsigned)p.lda * 2u, ldB2 = (unsigned)p.ldb * 2u;
    const char* const baseA = (const char*)p.A;
    const char* const baseB = (const char*)p.B;
    unsigned offP = 0, offQ = 0;   // per-lane byte offsets inside a chunk's source: k-row (lane>>4)*16, 16-byte piece lane&15
    auto set_sources = [&](int item) {
        split_issue = item / tiles_out;
        const int tile_id = item - split_issue * tiles_out;
        kb_issue = split_issue * p.nk_split;
        nk_issue = min(p.nk_split, nk_total - kb_issue);
        const int tb = tile_id / p.tiles_a, ta = tile_id - tb * p.tiles_a;
        a0 = ta * 256;
        b0 = tb * 256;
        first_col_tile = tb == 0;
        offP = (unsigned)(lane >> 4) * 16u * ldB2 + (unsigned)(lane & 15) * 16u + (unsigned)b0 * 2u;
        offQ = (unsigned)(lane >> 4) * 16u * ldA2 + (unsigned)(lane & 15) * 16u + (unsigned)a0 * 2u;
    };
    // this wave fills chunks 2*wave and 2*wave+1 of every half-tile (k-rows n, n+16, n+32, n+48 of the K-tile for chunk n)
    const unsigned lds_dma0 = (unsigned)(size_t)(lds_void*)smem + (unsigned)(2 * wave) * (unsigned)TN_CHUNK;
#define TN_GLDS_PAIR(off, sbase0, sbase1, ldsdst)                                                       \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"                   \
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"                         \
                 :: "v"(off), "s"(sbase0), "s"(sbase1), "s"(ldsdst), "s"((ldsdst) + (unsigned)TN_CHUNK) : "memory", "m0")
#define TN_ISSUE(u, j, stage)                                                                           \
    do {                                                                                                \
        const unsigned dst_ = lds_dma0 + (unsigned)((stage) * TN_STAGE + (j) * TN_HALF);                \
        const size_t krow_ = (size_t)(unsigned)((u) + kb_issue) * 64u + (size_t)(2 * wave);             \
        if ((j) == 0 || (j) == 3) {                                                                     \
            const char* s0_ = baseB + krow_ * ldB2 + ((j) == 3 ? 256 : 0);                              \
            TN_GLDS_PAIR(offP, s0_, s0_ + ldB2, dst_);                                                  \
        } else {                                                                                        \
            const char* s0_ = baseA + krow_ * ldA2 + ((j) == 2 ? 256 : 0);                              \
            TN_GLDS_PAIR(offQ, s0_, s0_ + ldA2, dst_);                                                  \
        }                                                                                               \
    } while (0)

    // ---- fragment addresses (see header): one VGPR per operand and stage, everything else is an immediate
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    const unsigned aB0 = lds0 + (unsigned)((4 * g + q) * TN_CHUNK + 128 * wm + 8 * pp);   // P operand, stage 0
    const unsigned aB1 = aB0 + (unsigned)TN_STAGE;
    const unsigned wB0 = lds0 + (unsigned)((4 * g + q) * TN_CHUNK + 64 * wn + 8 * pp);    // Q operand, stage 0
    const unsigned wB1 = wB0 + (unsigned)TN_STAGE;

    f32x4 acc[2][4][2][2];      // [hm][t][hn][n]
    f32x4 csum[2][2];           // CS: [hn][n], every column of D holds the same row sums
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    i32x2 aH[4][2][2];          // P fragments as read: [tile][kk][half of the 8 k-slots]
    i32x2 w0H[2][2][2], w1H[2][2][2];
    bf16x8 aF[4][2], w0F[2][2], w1F[2][2];

#define TN_TR(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define TN_LOAD_A(stage, j)                                                                                  \
    do {                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                        \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                               \
                TN_TR(aH[t][kk][0], (stage) ? aB1 : aB0, (j) * TN_HALF + (2 * kk) * 256 + 32 * t);           \
                TN_TR(aH[t][kk][1], (stage) ? aB1 : aB0, (j) * TN_HALF + (2 * kk + 1) * 256 + 32 * t);       \
            }                                                                                                \
    } while (0)
#define TN_LOAD_W(dstH, stage, j)                                                                            \
    do {                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                        \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                               \
                TN_TR(dstH[t][kk][0], (stage) ? wB1 : wB0, (j) * TN_HALF + (2 * kk) * 256 + 32 * t);         \
                TN_TR(dstH[t][kk][1], (stage) ? wB1 : wB0, (j) * TN_HALF + (2 * kk + 1) * 256 + 32 * t);     \
            }                                                                                                \
    } while (0)
#define TN_JOIN(lo, hi) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))
#define TN_WAIT_A()                                                                                          \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aH[0][0][0]), "+v"(aH[0][0][1]), "+v"(aH[0][1][0]), "+v"(aH[0][1][1]), \
                     "+v"(aH[1][0][0]), "+v"(aH[1][0][1]), "+v"(aH[1][1][0]), "+v"(aH[1][1][1]),              \
                     "+v"(aH[2][0][0]), "+v"(aH[2][0][1]), "+v"(aH[2][1][0]), "+v"(aH[2][1][1]),              \
                     "+v"(aH[3][0][0]), "+v"(aH[3][0][1]), "+v"(aH[3][1][0]), "+v"(aH[3][1][1]));             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                        \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) aF[t][kk] = TN_JOIN(aH[t][kk][0], aH[t][kk][1]); \
    } while (0)
#define TN_WAIT_W(wH, wF)                                                                                    \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wH[0][0][0]), "+v"(wH[0][0][1]), "+v"(wH[0][1][0]), "+v"(wH[0][1][1]), \
                     "+v"(wH[1][0][0]), "+v"(wH[1][0][1]), "+v"(wH[1][1][0]), "+v"(wH[1][1][1]));             \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                        \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) wF[t][kk] = TN_JOIN(wH[t][kk][0], wH[t][kk][1]); \
    } while (0)
#define TN_MMA(hm, hn, wF)                                                                                   \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                    acc[hm][t][hn][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wF[n][kk], aF[t][kk], acc[hm][t][hn][n], 0, 0, 0); \
        if constexpr (CS) {                                                                                  \
            if ((hm) == 0) {                                                                                 \
                _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                             \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                            \
                        csum[hn][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wF[n][kk], ones, csum[hn][n], 0, 0, 0); \
            }                                                                                                \
        }                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
#define TN_BARRIER()                                \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");              \
    } while (0)
#define TN_PHASE(Q8, issue_ok, WAITN)                                                                        \
    do {                                                                                                     \
        constexpr int st_ = ((Q8) >> 2) & 1;                                                                 \
        constexpr int dl_ = (Q8) & 3;                                                                        \
        if (dl_ == 0) { TN_LOAD_W(w0H, st_, 1); TN_LOAD_A(st_, 0); }                                         \
        else if (dl_ == 1) { TN_LOAD_W(w1H, st_, 2); }                                                       \
        else if (dl_ == 2) { TN_LOAD_A(st_, 3); }                                                            \
        if (issue_ok) {                                                                                      \
            constexpr int ju_ = ((Q8) + 2) & 3;                                                              \
            constexpr int du_ = ((Q8) + 6) >> 2;                                                             \
            TN_ISSUE(kt + du_, ju_, du_ & 1);                                                                \
        }                                                                                                    \
        TN_WAIT_VMCNT(WAITN);                                                                                \
        TN_BARRIER();                                                                                        \
        if (dl_ == 0) { TN_WAIT_W(w0H, w0F); TN_WAIT_A(); }                                                  \
        else if (dl_ == 1) { TN_WAIT_W(w1H, w1F); }                                                          \
        else if (dl_ == 2) { TN_WAIT_A(); }                                                                  \
        if (dl_ == 0) TN_MMA(0, 0, w0F);                                                                     \
        else if (dl_ == 1) TN_MMA(0, 1, w1F);                                                                \
        else if (dl_ == 2) TN_MMA(1, 1, w1F);                                                                \
        else TN_MMA(1, 0, w0F);                                                                              \
        TN_BARRIER();                                                                                        \
    } while (0)
#define TN_PROLOGUE()                                                        \
    do {                                                                     \
        TN_ISSUE(0, 0, 0); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(0, 1, 0); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(0, 2, 0); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(0, 3, 0); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(1, 0, 1); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(1, 1, 1); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(1, 2, 1); __builtin_amdgcn_sched_barrier(0);                \
        TN_ISSUE(1, 3, 1); __builtin_amdgcn_sched_barrier(0);                \
    } while (0)

    for (int item = blockIdx.x; item < nitems; item += (int)gridDim.x) {
        set_sources(item);
        TN_PROLOGUE();
        const int nk = nk_issue;
        TN_WAIT_VMCNT(8);   // L_0 .. L_3 of this item have landed (this wave's pieces)
        TN_BARRIER();
        if (wm == 1) TN_BARRIER();   // stagger: group 1 runs one barrier interval behind group 0
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (CS) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) csum[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int kt = 0;   // nk >= 4, even (host-checked)
        TN_PHASE(0, false, 8); TN_PHASE(1, false, 8);   // L_6, L_7 were part of the prologue
        TN_PHASE(2, true, 8); TN_PHASE(3, true, 8); TN_PHASE(4, true, 8); TN_PHASE(5, true, 8); TN_PHASE(6, true, 8); TN_PHASE(7, true, 8);
        for (kt = 2; kt < nk - 2; kt += 2) {
            TN_PHASE(0, true, 8); TN_PHASE(1, true, 8); TN_PHASE(2, true, 8); TN_PHASE(3, true, 8);
            TN_PHASE(4, true, 8); TN_PHASE(5, true, 8); TN_PHASE(6, true, 8); TN_PHASE(7, true, 8);
        }
        // last iteration (kt = nk - 2): two half-tiles left to issue, then drain with exact counts
        TN_PHASE(0, true, 8); TN_PHASE(1, true, 8);
        TN_PHASE(2, false, 6); TN_PHASE(3, false, 4);
        TN_PHASE(4, false, 2); TN_PHASE(5, false, 0); TN_PHASE(6, false, 0); TN_PHASE(7, false, 0);
        if (wm == 0) TN_BARRIER();   // group 0 matches group 1's extra barrier; every LDS read of this item is complete

        // ---- epilogue: lane (c = lane&15, g) holds output rows a0 + 128 hn + 32 wn + 16 n + 4 g + r, column b0 + 128 hm + 64 wm + 16 t + c
        {
            int ec = lane & 15, eg = lane >> 4;
            asm volatile("" : "+v"(ec), "+v"(eg));   // keep the 128 store addresses out of the persistent loop's live ranges
            float* outp = p.partials + (size_t)split_issue * (size_t)p.split_stride;
            const int col0 = b0 + 64 * wm + ec;
            const int row0 = a0 + 32 * wn + 4 * eg;
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* orow = outp + (size_t)(row0 + 128 * hn + 16 * n + r) * (size_t)p.Nb + col0;
#pragma unroll
                        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                            for (int t = 0; t < 4; ++t) orow[128 * hm + 16 * t] = acc[hm][t][hn][n][r];
                    }
            if constexpr (CS) {
                if (first_col_tile && wm == 0 && ec == 0) {
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + row0 + 128 * hn + 16 * n + r, csum[hn][n][r]);
                }
            }
        }
        TN_WAIT_VMCNT(0);   // the next item's LDS-DMA must not queue behind 128 stores that vmcnt cannot count past
    }
}

// Returns the number of M-slices used (>= 1) or 0 when the shape is not one this kernel takes.
int gemm256_tn_splitk_launch(const unsigned short* A, int lda, const unsigned short* B, int ldb, int M, int Na, int Nb, float* partials,
                             size_t partials_elems, float* colsum, hipStream_t stream) {
    if (M <= 0 || M % 128 != 0 || Na % 256 != 0 || Nb % 256 != 0 || Na <= 0 || Nb <= 0) return 0;
    const int nk = M / 64;
    if (nk < 4) return 0;
    TnParams q{};
    q.A = A; q.B = B; q.M = M; q.Na = Na; q.Nb = Nb; q.lda = lda; q.ldb = ldb;
    q.tiles_a = Na / 256;
    q.tiles_b = Nb / 256;
    const int tiles = q.tiles_a * q.tiles_b;
    static const int num_cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        }
        return n;
    }();
    int nks = 0;
    const int splits = plan_k_slices(nk, tiles, num_cus, &nks);   // nk is even (M % 128 == 0) and >= 4: never refuses
    if ((size_t)splits * (size_t)Na * (size_t)Nb > partials_elems) return 0;
    q.splits = splits;
    q.nk_split = nks;
    q.partials = partials;
    q.split_stride = (long long)Na * Nb;
    q.colsum = colsum;
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm256_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS) == hipSuccess &&
                                hipFuncSetAttribute((const void*)gemm256_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS) == hipSuccess;
    if (!attr_ok) return 0;
    int nitems = tiles * splits;
    const int grid = nitems < num_cus ? nitems : num_cus;
    void* args[] = {(void*)&q, (void*)&nitems};
    const void* fn = colsum != nullptr ? (const void*)gemm256_tn_kernel<true> : (const void*)gemm256_tn_kernel<false>;
    if (hipLaunchKernel(fn, dim3((unsigned)grid), dim3(TN_THREADS), args, TN_LDS, stream) != hipSuccess) return 0;
    return splits;
}

}  // namespace clibd
