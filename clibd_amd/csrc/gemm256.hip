// bf16 MFMA GEMM, 256x256x64 tile, 8 waves, one workgroup per CU, 8-phase software pipeline (gfx950).
//
//   out = epilogue(A[M,K] · W[N,K]^T), same contract and epilogue as gemm.hip (gemm_common.h).
//
// Geometry: wave (wm, wn) = (wave>>2, wave&3) owns 128 output COLUMNS x 64 output ROWS = 8 x 4 v_mfma_f32_16x16x32_bf16
// tiles (128 accumulator VGPRs).  The two operands are named by their role in the MFMA, not by the matrix:
//     P ("column operand", MFMA B / D columns) = W rows  = output columns: 8 tiles per wave, two halves hm0 / hm1
//     Q ("row operand",    MFMA A / D rows)    = A rows  = output rows   : 4 tiles per wave, two halves hn0 / hn1
// so a lane (c = lane&15, g = lane>>4) ends up with D[4g + r][c] of every tile; P tile nt = 4hm + t carries tile-local
// column 8c + nt, i.e. the lane owns 8 CONTIGUOUS output columns of rows 4g + r: one 16-byte bf16 store per row, and the
// 16 lanes of a row group write 256 contiguous bytes (tools/micro/store_patterns: 68 B/clk/CU against 28 B/clk/CU for
// the row-per-lane shape; adjacent LANES must be adjacent in memory for the store path to coalesce).
// A K-tile (BK = 64) is staged as FOUR 16-KiB half-tiles, cut along the *phase* structure rather than the wave grid:
//     P_hm0 : P tiles 0..3 of every wave (LDS rows 64wm + 16t + c)              needed at phase 0 of the K-tile
//     Q_hn0 : Q tiles 0,1 of every wave  (tile rows 64wn + 0..31)               needed at phase 0
//     Q_hn1 : Q tiles 2,3 of every wave  (tile rows 64wn + 32..63)              needed at phase 1
//     P_hm1 : P tiles 4..7 of every wave                                        needed at phase 2
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
// tile i it already issues the first EIGHT half-tiles of tile i+1 (all slots are free at a tile boundary), so the next
// tile's HBM/L2 latency and this tile's output stores overlap.  vmcnt retires in issue order and counts stores, so a wait
// for a half-tile issued AFTER the stores (L_8, first needed at phase 6) also waits for the stores: the first six phases
// of a tile wait with vmcnt(8 + stores of the previous epilogue) and the stores get six phases to drain.
#include <cstdlib>
#include "gemm_common.h"
#include "host_util.h"

namespace clibd {

constexpr int T_M = 256, T_N = 256, T_K = 64;
constexpr int HALF_BYTES = 128 * T_K * 2;      // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // 64 KiB
constexpr int G256_THREADS = 512;
constexpr int G256_LDS = 2 * STAGE_BYTES;      // 128 KiB
// stream-K tail workspace (caller-owned, zeroed ONCE by the caller): 256 flag words at a FIXED place in front — every launch leaves them zero again, whatever
// shape used the buffer last — then the fp32 partial tiles
constexpr size_t G256_SK_FLAG_BYTES = 1024;
// tile order (gemm_common.h tile_coords): W-stationary n-groups of 6 column tiles for every launch (round 5: in-step A/B against the m-bands of 4 of
// rounds 1-4: -0.8 % of the b = 2048 step, -1.3 % at b = 256, -0.7 % in the fp8 modes; profiles/r05_exp_gemm_ws_in_step.log).  The -D knobs build the A/B variants.
#ifndef CLIBD_WS_MIN_TILES_N
#define CLIBD_WS_MIN_TILES_N 1
#endif
#ifndef CLIBD_WS_GROUP
#define CLIBD_WS_GROUP 6
#endif
constexpr int G256_WS_MIN_TILES_N = CLIBD_WS_MIN_TILES_N, G256_WS_GROUP = CLIBD_WS_GROUP;

#define CLIBD_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// FP8: both operands are OCP e4m3 bytes (K counted in bytes; a K-tile is still 128 bytes per row, i.e. 128 k), the same
// LDS image and DMA schedule, and each (P tile, Q tile) pair takes ONE v_mfma_scale_f32_16x16x128_f8f6f4 with unit block
// scales on the lane's two 16-byte k-chunks (32 cycles for 128 k against 2 x 16 cycles for 64 k in bf16).  The accumulators
// are dequantised by p.col_scale[n] right after the K loop, before the (bf16) LoRA rank update and the epilogue.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x8 cat_frag(bf16x8 lo, bf16x8 hi) {
    return __builtin_shufflevector(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// DG (8-bit dgrad, round 5; FP8 only): the A operand is a GRADIENT as e4m3 bytes with one power-of-two scale per ROW (written by
// clibd_layernorm_bwd_fp8), W the transposed frozen weight as e4m3 with one scale per row (= per input channel of the layer).
//   EPI_BF16 / EPI_ADD_AUX: v = acc * col_scale[n] * a_row_dequant[m]  [+ aux]  -> out_bf16
//   EPI_MUL_AUX[_U8]      : out := e4m3(acc * col_scale[n] * aux[m,n] * out_fp8_scale) bytes (aux = gelu' as bf16, or as the one-byte code of §4) — the row scale of A passes THROUGH to the
//                           output (the next dgrad's A operand, dequantised by the same a_row_dequant; 1 / out_fp8_scale rides in its col_scale).
//   DG == 2 (round 6, full fine-tune; MUL_AUX kinds): additionally p.dual_bf16[m,n] = bf16(acc * col_scale[n] * aux[m,n] * a_row_dequant[m]) — the
//                           true d(fc1 out), which the bf16 weight gradient of fc1 contracts with its input.  The row factors are powers of two
//                           (clibd_layernorm_bwd_fp8 writes 2^(e - 134)): a lane keeps the four exponents of a row group in ONE register.
// SK (round 6): the stream-K tail — the last, partial round's tiles cut into K-slices over the idle CUs (GemmParams.sk_*); bf16 kinds only.
template <int KIND, bool LORA, bool BIAS, bool DIAG, bool FP8, int DG, bool SK = false>
__device__ __forceinline__ void gemm256_body(const GemmParams& p, int ntiles, int skew_ticks, long long* stamps) {
    static_assert(!DG || (FP8 && !LORA && !BIAS && !DIAG), "the 8-bit dgrad forms are fp8, bias-free and adapter-free");
    static_assert(!SK || (!FP8 && !DIAG && KIND != EPI_SPLITK_F32), "the stream-K tail exists for the bf16 epilogue kinds");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    constexpr int ESZ = FP8 ? 1 : 2;             // bytes per operand element
    const int nk_total = p.K * ESZ / (T_K * 2);  // 128-byte K-tiles
    const int tiles_out = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;  // work item: (output tile, split) = (item % tiles_out, item / tiles_out); splits == 1: item = tile
    int m0, n0;          // tile being computed
    int nm0 = 0, nn0 = 0;  // tile whose loads are being issued (== m0,n0 until the last issue of the current tile)
    int kb_issue = 0, nk_issue = nk_total, split_issue = 0;  // K-tile base / K-tile count (even, >= 4) / split of that item
    int part_issue = 0, slot_issue = -1;                     // SK: K-slice and tail slot of that item (slot -1: a whole tile)

    // ---- per-lane LDS-DMA sources: half-tile type j (0 P_hm0, 1 Q_hn0, 2 Q_hn1, 3 P_hm1) x this wave's 2 pieces
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ prow;
    // This wave fills LDS rows 16*wave + 8i + prow (piece i = 0/1) of every half-tile:
    //   P half: LDS row = 64wm' + 16t + c  ->  wm' = wave>>2, t = wave&3, c = 8i + prow;  tile col = 128wm' + 8c + 4hm + t
    //           piece i of P_hm0: tile col rowP + 64i        P_hm1: + 4
    //   Q half: LDS row = 32wn' + 16t + c  ->  wn' = wave>>1, t = wave&1, c = 8i + prow;  tile row = 64wn' + 32hn + 16t + c
    //           piece i of Q_hn0: tile row rowQ + 8i         Q_hn1: + 32
    // The load segment of a phase is on the critical path (it must fit under the other wave group's 256-cycle MFMA burst),
    // so it carries NO per-issue address arithmetic: the six per-lane byte offsets of a tile (row * ld + swizzled chunk;
    // operands are < 4 GiB, host-checked) are computed once per tile, and the K offset rides in the scalar base address.
    const int rowP = 128 * (wave >> 2) + 8 * prow + (wave & 3);
    const int rowQ = 64 * (wave >> 1) + 16 * (wave & 1) + prow;
    const unsigned chunk16 = (unsigned)chunk * 16u;
    const unsigned lda2 = (unsigned)p.lda * (unsigned)ESZ, ldw2 = (unsigned)p.ldw * (unsigned)ESZ;  // row strides in bytes
    const char* const baseA = (const char*)p.A;
    const char* const baseW = (const char*)p.W;
    unsigned offP0 = 0, offP1 = 0;                          // P_hm0 pieces (P_hm1 = + 4 rows: scalar)
    unsigned offQ00 = 0, offQ01 = 0, offQ10 = 0, offQ11 = 0;  // Q_hn0 / Q_hn1 pieces (rows clamp individually on a ragged tile)
    auto set_sources = [&](int item) {
        int tm, tn;
        int tile_id = item;
        if (KIND == EPI_SPLITK_F32) {
            split_issue = item / tiles_out;
            tile_id = item - split_issue * tiles_out;
            kb_issue = split_issue * p.nk_split;
            nk_issue = min(p.nk_split, nk_total - kb_issue);
        }
        if constexpr (SK) {
            part_issue = 0; slot_issue = -1; kb_issue = 0; nk_issue = nk_total;
            if (item >= p.sk_first) {   // a K-slice of a tail tile: non-owners (parts 1 ..) first, owners (part 0) last
                const int j = item - p.sk_first;
                const int nown = p.sk_tail * (p.sk_parts - 1);
                if (j < nown) { part_issue = 1 + j / p.sk_tail; slot_issue = j - (part_issue - 1) * p.sk_tail; }
                else { part_issue = 0; slot_issue = j - nown; }
                tile_id = p.sk_first + slot_issue;
                kb_issue = part_issue * p.sk_nk_part;
                nk_issue = min(p.sk_nk_part, nk_total - kb_issue);
            }
        }
        tile_coords(tile_id, p.tiles_m, p.tiles_n, p.band, tm, tn);
        nm0 = tm * T_M;
        nn0 = tn * T_N;
        offP0 = (unsigned)(nn0 + rowP) * ldw2 + chunk16;
        offP1 = (unsigned)(nn0 + rowP + 64) * ldw2 + chunk16;
        offQ00 = (unsigned)min(nm0 + rowQ, p.M - 1) * lda2 + chunk16;
        offQ01 = (unsigned)min(nm0 + rowQ + 8, p.M - 1) * lda2 + chunk16;
        offQ10 = (unsigned)min(nm0 + rowQ + 32, p.M - 1) * lda2 + chunk16;
        offQ11 = (unsigned)min(nm0 + rowQ + 40, p.M - 1) * lda2 + chunk16;
    };
    // issue half-tile j (0 P_hm0, 1 Q_hn0, 2 Q_hn1, 3 P_hm1) of K-tile u into stage (u & 1): two LDS-DMA instructions in the
    // scalar-base + 32-bit-VGPR-offset form (inline asm: left to the compiler, the loop-invariant per-lane parts become
    // 64-bit VGPR pointers re-added every issue).  These loads are invisible to the compiler's waitcnt pass; every wait
    // for them is an explicit CLIBD_WAIT_VMCNT (the counter still retires in issue order across asm and compiler ops).
    const unsigned lds_dma0 = (unsigned)(size_t)(lds_void*)smem + (unsigned)(2 * wave) * 1024u;
#define GLDS_PAIR(off0, off1, sbase, ldsdst)                                                            \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"                   \
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                         \
                 :: "v"(off0), "v"(off1), "s"(sbase), "s"(ldsdst), "s"((ldsdst) + 1024u) : "memory", "m0")
#define ISSUE(u, j, stage)                                                                              \
    do {                                                                                                \
        const unsigned dst_ = lds_dma0 + (unsigned)((stage) * STAGE_BYTES + (j) * HALF_BYTES);          \
        const size_t ks_ = (size_t)(unsigned)((u) + kb_issue) * (T_K * 2);  /* wave-uniform: folded into the scalar base */ \
        if ((j) == 0) GLDS_PAIR(offP0, offP1, baseW + ks_, dst_);                                       \
        else if ((j) == 3) GLDS_PAIR(offP0, offP1, baseW + ks_ + (size_t)ldw2 * 4, dst_);               \
        else if ((j) == 1) GLDS_PAIR(offQ00, offQ01, baseA + ks_, dst_);                                \
        else GLDS_PAIR(offQ10, offQ11, baseA + ks_, dst_);                                              \
    } while (0)

    // ---- fragment reads: tile t of a wave sits 16 rows = 2048 bytes after tile 0 and the swizzle term (row & 7) does not
    // depend on t, so one per-lane LDS address per (operand, stage, k-chunk) - 8 VGPRs - addresses every fragment as
    // base + IMMEDIATE offset.  The reads are inline asm: hipcc otherwise materialises ~20 slot variants of these addresses
    // as loop-invariant VGPRs (spilled), or re-adds constants in the load segment.  asm results are invisible to the
    // compiler's waitcnt pass, so every consumer is preceded by WAIT_FRAGS (s_waitcnt lgkmcnt(0) carrying the fragments as
    // in/out operands, which orders the MFMAs behind it).
    const int frow = lane & 15, fch = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    const unsigned aB00 = lds0 + (unsigned)tile_off(64 * wm + frow, fch), aB01 = aB00 ^ 64u;   // stage 0, k-chunk 0 / 1
    const unsigned aB10 = aB00 + STAGE_BYTES, aB11 = aB01 + STAGE_BYTES;                        // stage 1
    const unsigned wB00 = lds0 + (unsigned)tile_off(32 * wn + frow, fch), wB01 = wB00 ^ 64u;
    const unsigned wB10 = wB00 + STAGE_BYTES, wB11 = wB01 + STAGE_BYTES;

    f32x4 acc[2][4][2][2];  // [hm][mt][hn][nt]
    bf16x8 aF[4][2], w0F[2][2], w1F[2][2];  // [tile][kk]
    i32x8 aF8[4], w0F8[2], w1F8[2];         // FP8: [tile], both 16-byte k-chunks of the lane as ONE 8-VGPR MFMA operand
    // (w0F / w0F8 etc. are selected by token pasting in the macros below: the unused set never materialises)

#define DS_READ128(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
// FP8: the two halves are joined right at the read (a REG_SEQUENCE the coalescer folds into the asm outputs: the ISA must
// show no v_mov between a ds_read and its WAIT_FRAGS - tools/check_gemm256_isa.sh)
#define LOAD_A(stage, j)                                                                                     \
    do {                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                      \
            DS_READ128(aF[t][0], (stage) ? aB10 : aB00, (j) * HALF_BYTES + 2048 * t);                        \
            DS_READ128(aF[t][1], (stage) ? aB11 : aB01, (j) * HALF_BYTES + 2048 * t);                        \
        }                                                                                                    \
    } while (0)
#define LOAD_W(dstF, stage, j)                                                                               \
    do {                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                      \
            DS_READ128(dstF[t][0], (stage) ? wB10 : wB00, (j) * HALF_BYTES + 2048 * t);                      \
            DS_READ128(dstF[t][1], (stage) ? wB11 : wB01, (j) * HALF_BYTES + 2048 * t);                      \
        }                                                                                                    \
    } while (0)
#define WAIT_FRAGS_A()                                                                                       \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aF[0][0]), "+v"(aF[0][1]), "+v"(aF[1][0]), "+v"(aF[1][1]),  \
                     "+v"(aF[2][0]), "+v"(aF[2][1]), "+v"(aF[3][0]), "+v"(aF[3][1]));                         \
        if constexpr (FP8) { _Pragma("unroll") for (int t = 0; t < 4; ++t) aF8[t] = cat_frag(aF[t][0], aF[t][1]); } \
    } while (0)
#define WAIT_FRAGS_W(wF)                                                                                     \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wF[0][0]), "+v"(wF[0][1]), "+v"(wF[1][0]), "+v"(wF[1][1])); \
        if constexpr (FP8) { wF##8[0] = cat_frag(wF[0][0], wF[0][1]); wF##8[1] = cat_frag(wF[1][0], wF[1][1]); } \
    } while (0)
#define MMA(hm, hn, wF)                                                                                      \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        if constexpr (FP8) {                                                                                 \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                    /* inline asm with the accumulator TIED in and out: left to the builtin, hipcc gave the MFMAs of the peeled */ \
                    /* tail phases a destination different from their source accumulator and spilled 13-28 registers around   */ \
                    /* them, each reload behind a vmcnt(0) that also drains the LDS-DMA ring.  Unscaled form */ \
                    /* = unit block scales, e4m3 x e4m3 (cbsz = blgp = 0).  Invisible to the hazard recognizer: FP8_MFMA_DRAIN.  */ \
                    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[hm][t][hn][n]) : "v"(wF##8[n]), "v"(aF8[t])); \
        } else {                                                                                             \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                 \
                _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                            \
                        acc[hm][t][hn][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wF[n][kk], aF[t][kk], acc[hm][t][hn][n], 0, 0, 0); \
        }                                                                                                    \
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
        if (dl_ == 0) { WAIT_FRAGS_W(w0F); WAIT_FRAGS_A(); }                                                 \
        else if (dl_ == 1) { WAIT_FRAGS_W(w1F); }                                                            \
        else if (dl_ == 2) { WAIT_FRAGS_A(); }                                                               \
        if (dl_ == 0) MMA(0, 0, w0F);                                                                        \
        else if (dl_ == 1) MMA(0, 1, w1F);                                                                   \
        else if (dl_ == 2) MMA(1, 1, w1F);                                                                   \
        else MMA(1, 0, w0F);                                                                                 \
        BARRIER();                                                                                           \
    } while (0)

    // phase Q8 (2..7) of the LAST K-tile pair of a tile: issues half-tile Q8-2 of the next tile if there is one
#define PHASE_TAIL(Q8, WAITN_DRAIN)                                                                          \
    do {                                                                                                     \
        constexpr int st_ = ((Q8) >> 2) & 1;                                                                 \
        constexpr int dl_ = (Q8) & 3;                                                                        \
        if (dl_ == 0) { LOAD_W(w0F, st_, 1); LOAD_A(st_, 0); }                                               \
        else if (dl_ == 1) { LOAD_W(w1F, st_, 2); }                                                          \
        else if (dl_ == 2) { LOAD_A(st_, 3); }                                                               \
        if (has_next) {                                                                                      \
            constexpr int ju_ = ((Q8) + 2) & 3;                                                              \
            constexpr int du_ = ((Q8) + 6) >> 2;                                                             \
            ISSUE(du_ - 2, ju_, du_ & 1);                                                                    \
            CLIBD_WAIT_VMCNT(8);                                                                             \
        } else {                                                                                             \
            CLIBD_WAIT_VMCNT(WAITN_DRAIN);                                                                   \
        }                                                                                                    \
        BARRIER();                                                                                           \
        if (dl_ == 0) { WAIT_FRAGS_W(w0F); WAIT_FRAGS_A(); }                                                 \
        else if (dl_ == 1) { WAIT_FRAGS_W(w1F); }                                                            \
        else if (dl_ == 2) { WAIT_FRAGS_A(); }                                                               \
        if (dl_ == 0) MMA(0, 0, w0F);                                                                        \
        else if (dl_ == 1) MMA(0, 1, w1F);                                                                   \
        else if (dl_ == 2) MMA(1, 1, w1F);                                                                   \
        else MMA(1, 0, w0F);                                                                                 \
        BARRIER();                                                                                           \
    } while (0)

    // wait at the end of a load segment in the first six phases of a tile (and before its phase 0): the previous
    // epilogue's stores sit between the prologue LDS-DMA and this tile's later issues in the in-order vmcnt queue
#define WAIT_HEAD()                                         \
    do {                                                    \
        if (allow_case == 0) CLIBD_WAIT_VMCNT(8);           \
        else if (allow_case == 1) CLIBD_WAIT_VMCNT(24);     \
        else if (allow_case == 2) CLIBD_WAIT_VMCNT(40);     \
        else if (allow_case == 3) CLIBD_WAIT_VMCNT(56);     \
        else CLIBD_WAIT_VMCNT(63);                          \
    } while (0)
#define PHASE_HEAD(Q8)  /* phase Q8 (0..7) of the first K-tile pair of a tile (kt = 0), store-aware wait */     \
    do {                                                                                                     \
        constexpr int st_ = ((Q8) >> 2) & 1;                                                                 \
        constexpr int dl_ = (Q8) & 3;                                                                        \
        if (dl_ == 0) { LOAD_W(w0F, st_, 1); LOAD_A(st_, 0); }                                               \
        else if (dl_ == 1) { LOAD_W(w1F, st_, 2); }                                                          \
        else if (dl_ == 2) { LOAD_A(st_, 3); }                                                               \
        if ((Q8) >= 2) { /* L_6, L_7 belong to the prologue: all 8 slots are free at a tile boundary */      \
            constexpr int ju_ = ((Q8) + 2) & 3;                                                              \
            constexpr int du_ = ((Q8) + 6) >> 2;                                                             \
            ISSUE(du_, ju_, du_ & 1);                                                                        \
        }                                                                                                    \
        WAIT_HEAD();                                                                                         \
        BARRIER();                                                                                           \
        if (dl_ == 0) { WAIT_FRAGS_W(w0F); WAIT_FRAGS_A(); }                                                 \
        else if (dl_ == 1) { WAIT_FRAGS_W(w1F); }                                                            \
        else if (dl_ == 2) { WAIT_FRAGS_A(); }                                                               \
        if (dl_ == 0) MMA(0, 0, w0F);                                                                        \
        else if (dl_ == 1) MMA(0, 1, w1F);                                                                   \
        else if (dl_ == 2) MMA(1, 1, w1F);                                                                   \
        else MMA(1, 0, w0F);                                                                                 \
        BARRIER();                                                                                           \
    } while (0)
#define PROLOGUE_ISSUE()                                                     \
    do {                                                                     \
        /* sched_barrier: one ISSUE's address arithmetic at a time (all six hoisted together cost ~24 VGPRs -> acc spills) */ \
        ISSUE(0, 0, 0); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(0, 1, 0); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(0, 2, 0); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(0, 3, 0); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(1, 0, 1); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(1, 1, 1); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(1, 2, 1); __builtin_amdgcn_sched_barrier(0);                   \
        ISSUE(1, 3, 1); __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)

    // ---- optional start-up skew (diagnostic build only) and the first tile's prologue: L_0 .. L_7 = K-tiles 0 and 1
    if (DIAG && skew_ticks > 0) {
        const int cls = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)cls * skew_ticks) __builtin_amdgcn_s_sleep(8);
    }
    set_sources(tile);
    PROLOGUE_ISSUE();
    int allow_case = 0;
    const clibd_gemm_epilogue& ep = p.ep;
    const int stores_case = (ep.out_pre_bf16 ? 1 : 0) + (ep.out_bf16 ? 1 : 0) + (ep.out_f32 ? 2 : 0) + (DG == 2 ? 1 : 0);  // (16-B stores per row) / 2

    // diagnostic time stamps (only when a stamp buffer is installed: clibd_debug_set_gemm_stamps): [wg][tile_i][8]
    int tile_i = 0;
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        if (DIAG && stamps != nullptr && tile_i < 16 && (wave == 0 || wave == 4) && lane == 0)                    \
            stamps[(((size_t)blockIdx.x * 16 + tile_i) * 2 + (wave >> 2)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

    while (true) {
        m0 = nm0;
        n0 = nn0;
        const int nk = nk_issue;         // K-tiles of this work item (== K/64 unless split-K)
        const int split_cur = split_issue;
        const int part_cur = part_issue, slot_cur = slot_issue;   // SK
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

        int kt = 0;  // nk >= 4 (host-checked): the head phases always run and consume the literal-zero accumulators
        PHASE_HEAD(0); PHASE_HEAD(1); PHASE_HEAD(2); PHASE_HEAD(3);
        STAMP(1);
        PHASE_HEAD(4); PHASE_HEAD(5);
        // from phase 6 on the awaited half-tile (L_8 ...) was issued AFTER the stores: in-order vmcnt => plain vmcnt(8)
        PHASE(6, true, 8); PHASE(7, true, 8);
        STAMP(2);
        for (kt = 2; kt < nk - 2; kt += 2) {
            PHASE(0, true, 8); PHASE(1, true, 8); PHASE(2, true, 8); PHASE(3, true, 8);
            PHASE(4, true, 8); PHASE(5, true, 8); PHASE(6, true, 8); PHASE(7, true, 8);
        }
        // last iteration (kt = nk-2): only L_{4nk-2}, L_{4nk-1} are left to issue; drain with exact counts
        STAMP(3);
        PHASE(0, true, 8); PHASE(1, true, 8);
        // The ring does not stop at the tile boundary: L_{4nk+i} is half-tile i of the NEXT tile (nk is even, so its K-tile 0
        // falls on stage 0), issued six phases ahead like any other.  Phases 2..7 carry L'_0..L'_5 through a memory pipe that
        // would otherwise idle there, and the epilogue below shares it with two half-tiles instead of eight.  Without a next
        // tile the phases drain with exact counts.  (One code path with a scalar branch around the issue: two copies of the
        // phases, one per value of has_next, put the 128 accumulators behind phis and spill 180 registers.)
        if (has_next) set_sources(next);  // no more issues for this tile: the source registers now describe the next one
        PHASE_TAIL(2, 6); PHASE_TAIL(3, 4); PHASE_TAIL(4, 2); PHASE_TAIL(5, 0); PHASE_TAIL(6, 0); PHASE_TAIL(7, 0);
        if (wm == 0) BARRIER();  // group 0 matches group 1's extra barrier; every LDS read of this tile is complete
        // FP8_MFMA_DRAIN: the fp8 MFMAs are inline asm, so hipcc inserts no wait states between the last of them (8 passes) and the
        // first VALU read of an accumulator below: 18 are required, give 32.  The drain is TIED to the accumulators: every
        // reader after the K loop — the dequantisation by col_scale, the DG row scales, the rank update, the epilogue — takes acc[] from
        // these statements' outputs, so no refactor of what is loaded when can hoist a read above them (an untied asm with a memory clobber
        // ordered the reads only through the col_scale load).  Two statements: an asm takes at most 30 operands, the accumulators are 32.
        if constexpr (FP8) {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
                asm volatile("s_nop 15\n\ts_nop 15"
                             : "+v"(acc[hm][0][0][0]), "+v"(acc[hm][0][0][1]), "+v"(acc[hm][0][1][0]), "+v"(acc[hm][0][1][1]),
                               "+v"(acc[hm][1][0][0]), "+v"(acc[hm][1][0][1]), "+v"(acc[hm][1][1][0]), "+v"(acc[hm][1][1][1]),
                               "+v"(acc[hm][2][0][0]), "+v"(acc[hm][2][0][1]), "+v"(acc[hm][2][1][0]), "+v"(acc[hm][2][1][1]),
                               "+v"(acc[hm][3][0][0]), "+v"(acc[hm][3][0][1]), "+v"(acc[hm][3][1][0]), "+v"(acc[hm][3][1][1])
                             :: "memory");
        }
        STAMP(4);
        if constexpr (SK) {
            if (slot_cur >= 0) {   // a K-slice of a tail tile (always this workgroup's last item: the tail is one round)
                const int srow = 64 * wn + 4 * fch, scol = 128 * wm + 8 * frow;   // this lane's 16 rows x 8 columns inside the 256 x 256 tile
                if (part_cur > 0) {
                    // non-owner: the fp32 partial tile goes to the workspace, then ONE release increment of the slot's flag
                    // The partials travel with the sc0 sc1 bits (write-through / read-around the per-XCD L2, which is not coherent with the other seven): a
                    // release / acquire pair at agent scope would write back and invalidate a whole L2 per slice — measured: the first form of this tail,
                    // built on __ATOMIC_RELEASE / __ATOMIC_ACQUIRE, was 2.4 % SLOWER in-step at b = 256.  Here: coherent stores, vmcnt(0), barrier, ONE
                    // relaxed agent-scope increment; the owner spins on a relaxed agent-scope load and reads the partials with coherent loads.
                    float* outp = p.sk_ws + ((size_t)(part_cur - 1) * (size_t)p.sk_tail + (size_t)slot_cur) * (size_t)(T_M * T_N);
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* o = outp + (size_t)(srow + 32 * hn + 16 * n + r) * T_N + scol;
                                const f32x4 v0 = (f32x4){acc[0][0][hn][n][r], acc[0][1][hn][n][r], acc[0][2][hn][n][r], acc[0][3][hn][n][r]};
                                const f32x4 v1 = (f32x4){acc[1][0][hn][n][r], acc[1][1][hn][n][r], acc[1][2][hn][n][r], acc[1][3][hn][n][r]};
                                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" :: "v"(o), "v"(v0), "v"(v1) : "memory");
                            }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    BARRIER();                                      // every wave's stores are acknowledged at the coherence point
                    if (tid == 0) __hip_atomic_fetch_add(p.sk_flags + slot_cur, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                // owner: wait for the partners (dispatched before this workgroup: see GemmParams), add their partials in part order
                if (tid == 0) {
                    while (__hip_atomic_load(p.sk_flags + slot_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p.sk_parts - 1)) __builtin_amdgcn_s_sleep(16);
                    __hip_atomic_store(p.sk_flags + slot_cur, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
                }
                BARRIER();
                for (int q = 1; q < p.sk_parts; ++q) {
                    const float* inp = p.sk_ws + ((size_t)(q - 1) * (size_t)p.sk_tail + (size_t)slot_cur) * (size_t)(T_M * T_N);
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            f32x4 t0[4], t1[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float* o = inp + (size_t)(srow + 32 * hn + 16 * n + r) * T_N + scol;
                                asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1" : "=&v"(t0[r]), "=&v"(t1[r]) : "v"(o) : "memory");
                            }
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3]));
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int t = 0; t < 4; ++t) { acc[0][t][hn][n][r] += t0[r][t]; acc[1][t][hn][n][r] += t1[r][t]; }
                        }
                }
            }
        }
        // lane coordinates made opaque per tile: everything the epilogue derives from them (16 row offsets x several
        // leading dimensions) is rebuilt here instead of being hoisted out of the persistent loop into spilled registers
        int erow = frow, egrp = fch;
        asm volatile("" : "+v"(erow), "+v"(egrp));
        int nb = n0 + 128 * wm + 8 * erow;  // < N: N % 256 == 0 (host-checked)
        int mb = m0 + 64 * wn + 4 * egrp;
        // ---- Epilogue operands are requested BEFORE the next tile's prologue and awaited with an exact count after it.
        // vmcnt retires in issue order: a compiler-generated wait for a load issued after the 16 prologue LDS-DMA loads
        // waits for all of those to land first, and (worse) a wait inside each guarded row block — where hipcc put the
        // first use of the bias — is a vmcnt(0) that also waits for the previous row's store: one store in flight per
        // wave, 16 (fc1: 32) serial write round trips per tile.  The bias is therefore an inline-asm load (invisible to the
        // waitcnt pass), and the first row group of the two-pass kinds is loaded up here as well.
        constexpr bool RES_KIND = (KIND == EPI_RES_F32 || KIND == EPI_RES_F32_DROP || KIND == EPI_RES_F32_COPY);
        constexpr bool TWO_PASS = (epi_aux_kind(KIND) || RES_KIND || KIND == EPI_ROWNORM_GELU);
        constexpr bool PRELOAD_ROWS = TWO_PASS && !LORA && !FP8;  // (the LoRA / fp8 instantiations have no room for a row group)
        f32x4 bias_q[2];
        uint4 in_aux[8];
        f32x4 in_res[8][2];
        float2 in_st[8];      // EPI_ROWNORM_GELU: (mean, rstd) of the first row group
        f32x4 col_s[2];       // EPI_ROWNORM_GELU: s_n of this lane's eight columns
#define GLOAD128(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off ; EPI_OPERAND_LOAD" : "=&v"(dst) : "v"(ptr) : "memory")
        constexpr bool BIAS_ASM = BIAS && !FP8;  // (the fp8 instantiations spill with 8 more registers live across the dequantisation)
        if constexpr (BIAS_ASM) {
            const float* bp = ep.bias + nb;
            GLOAD128(bias_q[0], bp);
            GLOAD128(bias_q[1], bp + 4);
        }
        if constexpr (PRELOAD_ROWS) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mc = min(mb + 16 * n + r, p.M - 1);
                    if constexpr (KIND == EPI_ROWNORM_GELU) {
                        in_st[4 * n + r] = *(const float2*)(ep.row_stats + (size_t)mc * 2);
                    } else if constexpr (KIND == EPI_MUL_AUX_U8) {
                        const uint2 c8 = *(const uint2*)((const unsigned char*)ep.aux_bf16 + (size_t)mc * ep.ld_aux + nb);
                        in_aux[4 * n + r] = make_uint4(c8.x, c8.y, 0u, 0u);
                    } else if constexpr (KIND == EPI_MUL_AUX_12) {
                        in_aux[4 * n + r] = load_aux12((const unsigned char*)ep.aux_bf16 + (size_t)mc * ep.ld_aux + (nb >> 1) * 3);
                    } else if constexpr (epi_aux_kind(KIND)) {
                        in_aux[4 * n + r] = load_aux16<epi_aux_nt(KIND)>((const unsigned short*)ep.aux_bf16 + (size_t)mc * ep.ld_aux + nb);
                    } else {
                        const f32x4* rs = (const f32x4*)(ep.residual_f32 + (size_t)mc * ep.ld_res + nb);
                        in_res[4 * n + r][0] = rs[0];
                        in_res[4 * n + r][1] = rs[1];
                    }
                }
        }
        if constexpr (KIND == EPI_ROWNORM_GELU) {
            col_s[0] = *(const f32x4*)(ep.col_sum_w + nb);
            col_s[1] = *(const f32x4*)(ep.col_sum_w + nb + 4);
        }
        if (has_next) {  // L'_6, L'_7: ahead of the stores in the in-order queue (L'_0..L'_5 went out in phases 2..7)
            ISSUE(1, 2, 1); __builtin_amdgcn_sched_barrier(0);
            ISSUE(1, 3, 1); __builtin_amdgcn_sched_barrier(0);
        }
        // (rebuilt, so that only the requested operands — not the coordinates — are live across the prologue issue)
        erow = frow; egrp = fch;
        asm volatile("" : "+v"(erow), "+v"(egrp));
        nb = n0 + 128 * wm + 8 * erow;
        mb = m0 + 64 * wn + 4 * egrp;
        // queue now: [bias / LoRA asm loads] [first row group: 8 (aux) or 16 (residual) loads] [4 LDS-DMA loads if has_next]
#define EPI_TIED_WAIT(N)                                                                                             \
    do {                                                                                                             \
        if constexpr (BIAS_ASM) asm volatile("s_waitcnt vmcnt(" #N ") ; EPI_OPERAND_WAIT" : "+v"(bias_q[0]), "+v"(bias_q[1])); \
    } while (0)
        // ONE tied wait statement on every path (two of them, one per branch, make hipcc merge their results through copies it
        // places BEFORE the wait: stale bias).  Without a next tile there is no LDS-DMA behind the operands: an untied wait for
        // the operands themselves comes first and the tied one is then a no-op that only carries the data dependence.
        if constexpr (KIND == EPI_ROWNORM_GELU) {   // queue: [bias asm 2] [8 row-stat loads] [2 s_n loads] [4 LDS-DMA]
            if (!has_next) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            EPI_TIED_WAIT(14);
        } else if constexpr (PRELOAD_ROWS && epi_aux_kind(KIND)) {
            if (!has_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            EPI_TIED_WAIT(12);
        } else if constexpr (PRELOAD_ROWS) {
            if (!has_next) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            EPI_TIED_WAIT(20);
        } else {
            if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            EPI_TIED_WAIT(4);
        }
#undef EPI_TIED_WAIT
#undef GLOAD128
        // ---- FP8: dequantise in place, before the (bf16, unscaled) rank update: acc[.., column e] *= col_scale[nb + e]
        unsigned rdexp[2][2];   // DG == 2 only
        if constexpr (FP8) {
            const float* csp = p.col_scale + n0 + 128 * wm + 8 * erow;
            const f32x4 c0 = *(const f32x4*)csp, c1 = *(const f32x4*)(csp + 4);
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float c = hm ? c1[t] : c0[t];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[hm][t][q >> 1][q & 1] *= c;
                }
            if constexpr (DG == 2) {   // the four row exponents of every row group, packed (read here: no store of this tile is in the queue yet)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 rd = *(const f32x4*)(p.a_row_dequant + min(mb + 32 * hn + 16 * n, p.M - 4));
                        rdexp[hn][n] = ((__float_as_uint(rd[0]) >> 23) & 0xffu) | (((__float_as_uint(rd[1]) >> 23) & 0xffu) << 8) |
                                       (((__float_as_uint(rd[2]) >> 23) & 0xffu) << 16) | (((__float_as_uint(rd[3]) >> 23) & 0xffu) << 24);
                    }
            }
            if constexpr (DG && !epi_mul_aux_kind(KIND)) {   // the A operand's row scales (M % 4 == 0, host-checked: a group of 4 rows is all in or all out)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 rd = *(const f32x4*)(p.a_row_dequant + min(mb + 32 * hn + 16 * n, p.M - 4));
#pragma unroll
                        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[hm][t][hn][n] *= rd;
                    }
            }
        }
        // ---- LoRA rank-8 update: one extra zero-padded k-step (lanes with k-chunk 0 carry U[m,0:8] / V[n,0:8])
        // (compile-time flag: a run-time test here puts all 128 accumulators behind a phi the register allocator
        //  cannot coalesce -> 26 spilled VGPRs and vmcnt(0) drains around their reloads)
        // The operand loads sit behind the prologue LDS-DMA in the in-order queue (their wait also waits for those to land):
        // requesting them ahead of it, as the bias is, costs 48 live registers that the LoRA instantiations do not have.
        if (LORA) {
            bf16x8 uf[2][2];  // Q side (output rows): [hn][n]
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    uf[hn][n] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (egrp == 0) {
                        const int gm = min(m0 + 64 * wn + 32 * hn + 16 * n + erow, p.M - 1);
                        uf[hn][n] = *(const bf16x8*)((const unsigned short*)ep.rank_u + (size_t)gm * ep.ld_rank_u);
                    }
                }
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bf16x8 vf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};  // P side (output columns)
                    if (egrp == 0) {
                        const int gn = nb + 4 * hm + t;
                        vf = *(const bf16x8*)((const unsigned short*)ep.rank_v + (size_t)gn * 8);
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[hm][t][n >> 1][n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[n >> 1][n & 1], vf, acc[hm][t][n >> 1][n & 1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);  // keep the epilogue's aux / residual loads below the rank-update operands
        }

        // ---- epilogue: lane (c = frow, g = fch) owns rows m0 + 64wn + 32hn + 16n + 4g + r, columns nb .. nb+7 (e = 4hm + t)
        {
            float bias[8];
            if constexpr (BIAS_ASM) {  // compile-time: without a bias the 128 adds (and the moves pairing them) vanish
#pragma unroll
                for (int e = 0; e < 4; ++e) { bias[e] = bias_q[0][e]; bias[4 + e] = bias_q[1][e]; }
            } else if constexpr (BIAS) {
                load_bias8(ep, nb, bias);
                // one wait HERE (it also waits for the prologue loads to land), not one vmcnt(0) in every guarded row block
                if constexpr (!TWO_PASS) {  // (the two-pass kinds use the bias in their load-only pass: no store waits behind it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bias[e]));
                }
            }
#define EPV(hm, t) (BIAS ? acc[hm][t][hn][n][r] + bias[4 * (hm) + (t)] : acc[hm][t][hn][n][r])
#define FOR_ROWS_HN(HN0, HN1, ...)                                                                           \
    _Pragma("unroll") for (int hn = (HN0); hn < (HN1); ++hn)                                                 \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                        \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                  \
                const int m = mb + 32 * hn + 16 * n + r;                                                     \
                __VA_ARGS__                                                                                  \
            }
#define FOR_ROWS(...) FOR_ROWS_HN(0, 2, __VA_ARGS__)
            // pin the folded values of row group hn: left alone, LLVM sinks the adds / multiplies into pass 2's row predicates
            // and keeps every loaded vector alive instead
#define PIN_GROUP(hn_)                                                                                       \
    _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                        \
            _Pragma("unroll") for (int n2 = 0; n2 < 2; ++n2) asm volatile("" : "+v"(acc[hm][t][hn_][n2]))
            if (KIND == EPI_SPLITK_F32) {
                float* outp = ep.out_f32 + (size_t)split_cur * (size_t)p.split_stride;
                FOR_ROWS({
                    if (m < p.M) {
                        f32x4* o = (f32x4*)(outp + (size_t)m * ep.ld_out_f32 + nb);
                        o[0] = (f32x4){acc[0][0][hn][n][r], acc[0][1][hn][n][r], acc[0][2][hn][n][r], acc[0][3][hn][n][r]};
                        o[1] = (f32x4){acc[1][0][hn][n][r], acc[1][1][hn][n][r], acc[1][2][hn][n][r], acc[1][3][hn][n][r]};
                    }
                })
            } else if (TWO_PASS) {
                // Two passes so that no store sits between the epilogue's loads: vmcnt retires in order, and a load
                // waited behind earlier stores would pay their HBM write latency once per row (16 serial round trips).
                // Pass 1 folds bias / dropout / aux / residual into the accumulators in place (loads only, rows clamped):
                // row group hn = 0 from the operands requested before the prologue, group hn = 1 loads its own (8 aux /
                // 16 residual 16-B loads in flight); pass 2 only stores.
                float cs8[8];
                if constexpr (KIND == EPI_ROWNORM_GELU) {
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) { cs8[e] = col_s[0][e]; cs8[4 + e] = col_s[1][e]; }
                }
                FOR_ROWS_HN(0, 1, {
                    const int mc = min(m, p.M - 1);
                    float v[8];
                    _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) v[4 * hm + t] = (KIND == EPI_ROWNORM_GELU) ? acc[hm][t][hn][n][r] : EPV(hm, t);
                    if constexpr (KIND == EPI_ROWNORM_GELU) fold_rownorm8(v, in_st[4 * n + r].x, in_st[4 * n + r].y, cs8, bias);
                    else if constexpr (PRELOAD_ROWS) fold_row8_in<KIND>(ep, mc, nb, v, in_aux[4 * n + r], in_res[4 * n + r][0], in_res[4 * n + r][1]);
                    else fold_row8<KIND>(ep, mc, nb, v);
                    _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[hm][t][hn][n][r] = v[4 * hm + t];
                })
                PIN_GROUP(0);
                FOR_ROWS_HN(1, 2, {
                    const int mc = min(m, p.M - 1);
                    float v[8];
                    _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) v[4 * hm + t] = (KIND == EPI_ROWNORM_GELU) ? acc[hm][t][hn][n][r] : EPV(hm, t);
                    if constexpr (KIND == EPI_ROWNORM_GELU) {
                        const float2 st = *(const float2*)(ep.row_stats + (size_t)mc * 2);
                        fold_rownorm8(v, st.x, st.y, cs8, bias);
                    } else {
                        fold_row8<KIND>(ep, mc, nb, v);
                    }
                    _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[hm][t][hn][n][r] = v[4 * hm + t];
                })
                PIN_GROUP(1);
                float keep1 = 0.f, keep2 = 0.f;   // EPI_RES_F32_COPY: lane c keeps the sums of row j = c of its group's sixteen rows
                FOR_ROWS({
                    if (m < p.M) {
                        float v[8];
                        _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                            _Pragma("unroll") for (int t = 0; t < 4; ++t) v[4 * hm + t] = acc[hm][t][hn][n][r];
                        if (KIND == EPI_ROWNORM_GELU) {
                            store_row8<EPI_GELU_SAVE>(ep, m, nb, v);
                        } else if (DG && epi_mul_aux_kind(KIND)) {
                            *(uint2*)((unsigned char*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8fp8(v, p.out_fp8_scale);
                            if constexpr (DG == 2) {
                                const float rdf = __uint_as_float(((rdexp[hn][n] >> (8 * r)) & 0xffu) << 23);
                                float w8[8];
                                _Pragma("unroll") for (int e = 0; e < 8; ++e) w8[e] = v[e] * rdf;
                                *(uint4*)(p.dual_bf16 + (size_t)m * p.ld_dual + nb) = pack8bf(w8);
                            }
                        } else if (epi_aux_kind(KIND)) {
                            *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
                        } else {
                            f32x4* o = (f32x4*)(ep.out_f32 + (size_t)m * ep.ld_out_f32 + nb);
                            o[0] = (f32x4){v[0], v[1], v[2], v[3]};
                            o[1] = (f32x4){v[4], v[5], v[6], v[7]};
                            if constexpr (KIND == EPI_RES_F32_COPY) {
                                *(uint4*)((unsigned short*)ep.out_bf16 + (size_t)m * ep.ld_out_bf16 + nb) = pack8bf(v);
                                float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                                float s2 = 0.f;
                                _Pragma("unroll") for (int e = 0; e < 8; ++e) s2 = fmaf(v[e], v[e], s2);
                                s1 = row16_sum(s1);   // (the 16 lanes of a DPP row share m: all of them are here)
                                s2 = row16_sum(s2);
                                if (erow == 8 * hn + 4 * n + r) { keep1 = s1; keep2 = s2; }
                            }
                        }
                    }
                })
                if constexpr (KIND == EPI_RES_F32_COPY) {   // ONE store per wave and tile: 64 rows x (sum, sum of squares) of this 128-column slice
                    const int jm = mb + 32 * (erow >> 3) + 16 * ((erow >> 2) & 1) + (erow & 3);
                    const int slice = (n0 >> 7) + wm;
                    if (jm < p.M) *(float2*)(ep.row_sums + ((size_t)slice * (size_t)p.M + (size_t)jm) * 2) = make_float2(keep1, keep2);
                }
            } else {
                FOR_ROWS({
                    if (m < p.M) {
                        float v[8];
                        _Pragma("unroll") for (int hm = 0; hm < 2; ++hm)
                            _Pragma("unroll") for (int t = 0; t < 4; ++t) v[4 * hm + t] = EPV(hm, t);
                        if (FP8 && KIND == EPI_GELU_SAVE) store_row8_gelu_fp8(ep, m, nb, v, p.out_fp8_scale);
                        else store_row8<KIND>(ep, m, nb, v);
                    }
                })
            }
#undef PIN_GROUP
#undef FOR_ROWS_HN
#undef FOR_ROWS
#undef EPV
        }
        STAMP(5);
        if (DIAG) ++tile_i;
        if (!has_next) break;
        // a full tile issued exactly 16 rows x stores_case 16-B stores per lane; a ragged one fewer: be conservative there
        allow_case = (m0 + T_M <= p.M) ? stores_case : 0;
        tile = next;
    }
}

template <int KIND, bool LORA, bool BIAS, bool DIAG, bool FP8>
__global__ __launch_bounds__(G256_THREADS) void gemm256_bf16_nt_kernel(GemmParams p, int ntiles, int skew_ticks, long long* stamps) {
    gemm256_body<KIND, LORA, BIAS, DIAG, FP8, 0>(p, ntiles, skew_ticks, stamps);
}
template <int KIND, bool LORA, bool BIAS>
__global__ __launch_bounds__(G256_THREADS) void gemm256_bf16_nt_sk_kernel(GemmParams p, int ntiles, int skew_ticks, long long* stamps) {
    gemm256_body<KIND, LORA, BIAS, false, false, 0, true>(p, ntiles, skew_ticks, stamps);
}
// the stream-K tail is instantiated for the kinds whose launches have long contractions and few column tiles (N = 768, K = 2304 / 3072: fc2 forward,
// fc1 dgrad, QKV dgrad of both tower kinds)
static const void* kernel_ptr_sk(int kind, bool lora, bool bias) {
    switch (kind) {
        case EPI_BF16: return bias ? nullptr : lora ? (const void*)gemm256_bf16_nt_sk_kernel<EPI_BF16, true, false> : (const void*)gemm256_bf16_nt_sk_kernel<EPI_BF16, false, false>;
        case EPI_ADD_AUX: return bias ? nullptr : lora ? (const void*)gemm256_bf16_nt_sk_kernel<EPI_ADD_AUX, true, false> : (const void*)gemm256_bf16_nt_sk_kernel<EPI_ADD_AUX, false, false>;
        case EPI_RES_F32: return (bias && !lora) ? (const void*)gemm256_bf16_nt_sk_kernel<EPI_RES_F32, false, true> : nullptr;
        case EPI_RES_F32_DROP: return (bias && !lora) ? (const void*)gemm256_bf16_nt_sk_kernel<EPI_RES_F32_DROP, false, true> : nullptr;
        default: return nullptr;
    }
}
template <int KIND, int DG = 1>
__global__ __launch_bounds__(G256_THREADS) void gemm256_fp8_dgrad_kernel(GemmParams p, int ntiles, int skew_ticks, long long* stamps) {
    gemm256_body<KIND, false, false, false, true, DG>(p, ntiles, skew_ticks, stamps);
}

// The product library has no mutable state.  The s_memtime stamp buffer and the start-up skew knob of
// tools/gemm_stamps.py exist only in a diagnostic build: hipcc -DCLIBD_GEMM_DIAG (python -m clibd_amd.build --diag).
#ifdef CLIBD_GEMM_DIAG
static long long* g_stamp_buffer = nullptr;
static int skew_env_value() {
    static const int v = [] { const char* e = getenv("CLIBD_GEMM_SKEW"); return e ? atoi(e) : 0; }();
    return v;
}
static int band_env_value() {
    static const int v = [] { const char* e = getenv("CLIBD_GEMM_BAND"); return e ? atoi(e) : 0; }();
    return v;
}
#define CLIBD_DIAG_KERNEL(KIND) (const void*)gemm256_bf16_nt_kernel<KIND, false, true, true, false>
#else
static constexpr long long* g_stamp_buffer = nullptr;
static int skew_env_value() { return 0; }
static int band_env_value() { return 0; }
#define CLIBD_DIAG_KERNEL(KIND) (const void*)nullptr
#endif

static const void* kernel_ptr(int kind, bool lora, bool bias, bool diag) {
#define K256(KIND)                                                                                                    \
    (diag ? CLIBD_DIAG_KERNEL(KIND)                                                                                   \
          : lora ? (bias ? (const void*)gemm256_bf16_nt_kernel<KIND, true, true, false, false>                        \
                         : (const void*)gemm256_bf16_nt_kernel<KIND, true, false, false, false>)                      \
                 : (bias ? (const void*)gemm256_bf16_nt_kernel<KIND, false, true, false, false>                       \
                         : (const void*)gemm256_bf16_nt_kernel<KIND, false, false, false, false>))
    switch (kind) {
        case EPI_BF16: return K256(EPI_BF16);
        case EPI_GELU_SAVE: return K256(EPI_GELU_SAVE);
        case EPI_GELU: return K256(EPI_GELU);
        case EPI_F32: return K256(EPI_F32);
        case EPI_MUL_AUX: return K256(EPI_MUL_AUX);
        case EPI_ADD_AUX: return K256(EPI_ADD_AUX);
        case EPI_GELU_SAVE_U8: return K256(EPI_GELU_SAVE_U8);
        case EPI_MUL_AUX_U8: return K256(EPI_MUL_AUX_U8);
        case EPI_GELU_SAVE_12: return K256(EPI_GELU_SAVE_12);
        case EPI_MUL_AUX_12: return K256(EPI_MUL_AUX_12);
        case EPI_RES_F32: return K256(EPI_RES_F32);
        case EPI_RES_F32_DROP: return K256(EPI_RES_F32_DROP);
        case EPI_SPLITK_F32: return (const void*)gemm256_bf16_nt_kernel<EPI_SPLITK_F32, false, false, false, false>;
        case EPI_RES_F32_COPY: return (lora || !bias || diag) ? nullptr : (const void*)gemm256_bf16_nt_kernel<EPI_RES_F32_COPY, false, true, false, false>;
        case EPI_ROWNORM_GELU: return (lora || !bias || diag) ? nullptr : (const void*)gemm256_bf16_nt_kernel<EPI_ROWNORM_GELU, false, true, false, false>;
        default: return K256(EPI_GENERIC);
    }
#undef K256
}

// fp8 operands: only the four forward shapes of a transformer layer are instantiated (bias always present there)
static const void* kernel_ptr_fp8(int kind, bool lora) {
    switch (kind) {
        case EPI_BF16: return lora ? (const void*)gemm256_bf16_nt_kernel<EPI_BF16, true, true, false, true>
                                   : (const void*)gemm256_bf16_nt_kernel<EPI_BF16, false, true, false, true>;
        case EPI_GELU_SAVE: return lora ? nullptr : (const void*)gemm256_bf16_nt_kernel<EPI_GELU_SAVE, false, true, false, true>;
        case EPI_RES_F32: return lora ? nullptr : (const void*)gemm256_bf16_nt_kernel<EPI_RES_F32, false, true, false, true>;
        case EPI_RES_F32_DROP: return lora ? nullptr : (const void*)gemm256_bf16_nt_kernel<EPI_RES_F32_DROP, false, true, false, true>;
        default: return nullptr;
    }
}

static int device_cus() {
    static const int num_cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        }
        return n;
    }();
    return num_cus;
}

// fp8 launch: p.K / lda / ldw in bytes; the epilogue must be one of the instantiated forms and carry a bias.
bool gemm256_fp8_launch(const GemmParams& p, hipStream_t stream) {
    const int nk = p.K / 128;
    if (!p.fp8 || p.col_scale == nullptr || p.K % 128 != 0 || nk < 4 || (nk & 1) || p.N % T_N != 0 || p.M < 1 || p.ep.split_k > 1) return false;
    if (p.ep.bias == nullptr) return false;
    if ((unsigned long long)p.M * p.lda >= (1ull << 32) || (unsigned long long)p.N * p.ldw >= (1ull << 32)) return false;
    const int kind = epilogue_kind(p.ep);
    const bool lora = p.ep.rank_u != nullptr;
    const void* fn = kernel_ptr_fp8(kind, lora);
    if (fn == nullptr) return false;
    if ((p.out_fp8_scale > 0.f) != (kind == EPI_GELU_SAVE)) return false;
    static const bool attr_ok = [] {
        bool ok = true;
        for (int k = 0; k < EPI_NUM_KINDS; ++k)
            for (int l = 0; l < 2; ++l) {
                const void* f = kernel_ptr_fp8(k, l != 0);
                if (f != nullptr) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
            }
        return ok;
    }();
    if (!attr_ok) return false;
    GemmParams q = p;
    q.tiles_m = (p.M + T_M - 1) / T_M;
    q.tiles_n = p.N / T_N;
    q.splits = 1;
    q.nk_split = nk;
    q.split_stride = 0;
    q.band = q.tiles_n >= G256_WS_MIN_TILES_N ? -G256_WS_GROUP : 4;
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    const int grid = (int)(tiles < device_cus() ? tiles : device_cus());
    int ntiles_i = (int)tiles, skew = 0;
    long long* stamp = nullptr;
    void* args[] = {(void*)&q, (void*)&ntiles_i, (void*)&skew, (void*)&stamp};
    return hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G256_THREADS), args, G256_LDS, stream) == hipSuccess;
}

// 8-bit dgrad launch (clibd_gemm_fp8_dgrad_nt): the three bias-free backward forms
bool gemm256_fp8_dgrad_launch(const GemmParams& p, hipStream_t stream) {
    const int nk = p.K / 128;
    if (!p.fp8 || p.col_scale == nullptr || p.K % 128 != 0 || nk < 4 || (nk & 1) || p.N % T_N != 0 || p.M < 4 || (p.M & 3) || p.ep.split_k > 1) return false;
    if (p.ep.bias != nullptr || p.ep.rank_u != nullptr || p.ep.drop_thr16 > 0) return false;
    if ((unsigned long long)p.M * p.lda >= (1ull << 32) || (unsigned long long)p.N * p.ldw >= (1ull << 32)) return false;
    const int kind = epilogue_kind(p.ep);
    const bool dual = p.dual_bf16 != nullptr;
    const void* fn = kind == EPI_BF16 ? (const void*)gemm256_fp8_dgrad_kernel<EPI_BF16>
                   : kind == EPI_MUL_AUX ? (dual ? (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX, 2> : (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX>)
                   : kind == EPI_MUL_AUX_U8 ? (dual ? (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_U8, 2> : (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_U8>)
                   : kind == EPI_MUL_AUX_12 ? (dual ? (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_12, 2> : (const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_12>)
                   : kind == EPI_ADD_AUX ? (const void*)gemm256_fp8_dgrad_kernel<EPI_ADD_AUX> : nullptr;
    if (fn == nullptr) return false;
    const bool mul = epi_mul_aux_kind(kind);
    if ((p.out_fp8_scale > 0.f) != mul) return false;
    if ((!mul || dual) && p.a_row_dequant == nullptr) return false;
    if (dual && (!mul || (p.ld_dual & 7) || p.ld_dual < p.N || ((uintptr_t)p.dual_bf16 & 15))) return false;
    static const bool attr_ok = [] {
        bool ok = hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_U8>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_ADD_AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_U8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_12>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm256_fp8_dgrad_kernel<EPI_MUL_AUX_12, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
        return ok;
    }();
    if (!attr_ok) return false;
    GemmParams q = p;
    q.tiles_m = (p.M + T_M - 1) / T_M;
    q.tiles_n = p.N / T_N;
    q.splits = 1;
    q.nk_split = nk;
    q.split_stride = 0;
    q.band = q.tiles_n >= G256_WS_MIN_TILES_N ? -G256_WS_GROUP : 4;
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    const int grid = (int)(tiles < device_cus() ? tiles : device_cus());
    int ntiles_i = (int)tiles, skew = 0;
    long long* stamp = nullptr;
    void* args[] = {(void*)&q, (void*)&ntiles_i, (void*)&skew, (void*)&stamp};
    return hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G256_THREADS), args, G256_LDS, stream) == hipSuccess;
}

bool gemm256_try_launch(const GemmParams& p, hipStream_t stream) {
    const int nk = p.K / T_K;
    if (p.K % T_K != 0 || nk < 4 || (nk & 1)) return false;
    if (p.N % T_N != 0) return false;
    if (p.ep.split_k > 1) return false;
    const long long tiles = (long long)((p.M + T_M - 1) / T_M) * (p.N / T_N);
    if (p.M < 1024 || tiles < 128) return false;
    if ((unsigned long long)p.M * p.lda * 2ull >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2ull >= (1ull << 32)) return false;  // too few 256x256 tiles to fill 256 CUs: the 128x128 kernel wins
    static const bool attr_ok = [] {
        bool ok = true;
        for (int v = 0; v < 5; ++v)
            for (int k = 0; k < EPI_NUM_KINDS; ++k) {
                const void* fn = kernel_ptr(k, (v & 1) != 0, (v & 2) != 0, v == 4);
                if (fn != nullptr) ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
            }
        return ok;
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
    q.splits = 1;
    q.nk_split = nk;
    q.split_stride = 0;
    q.band = band_env_value() != 0 ? band_env_value() : (q.tiles_n >= G256_WS_MIN_TILES_N ? -G256_WS_GROUP : 4);
    int grid = (int)(tiles < num_cus ? tiles : num_cus);
#ifdef CLIBD_BALANCED_GRID
    // A/B knob (round 6): the SMALLEST grid that needs the same number of tile rounds (591 tiles: 3 rounds on 256 CUs -> 197 workgroups x 3 tiles
    // each), so that a launch with a mostly empty last round leaves whole CUs to the other tower's kernel from the start instead of for one round
    if (tiles > num_cus) {
        const long long rounds = (tiles + num_cus - 1) / num_cus;
        grid = (int)((tiles + rounds - 1) / rounds);
    }
#endif
    const int kind = epilogue_kind(p.ep);
    int ntiles_i = (int)tiles;
    int skew_arg = skew_env_value();
    long long* stamp_arg = (p.ep.rank_u != nullptr) ? nullptr : g_stamp_buffer;
    // Diagnostics (tools/gemm_stamps.py): a stamp buffer selects the instrumented instantiation; CLIBD_GEMM_SKEW=<ticks of
    // the 100 MHz s_memrealtime> adds a start-up skew between 4 classes of workgroups there.  Never used on the product path.
    const bool lora = p.ep.rank_u != nullptr;
    const bool diag = g_stamp_buffer != nullptr && !lora;
    void* args[] = {(void*)&q, (void*)&ntiles_i, (void*)&skew_arg, (void*)&stamp_arg};
    if (kind < 0) return false;                                   // an inconsistent fold epilogue (gemm_impl reports it)
    const void* fn = kernel_ptr(kind, lora, p.ep.bias != nullptr, diag);
    if (fn == nullptr) return false;
    // Stream-K tail (round 6): with a workspace from the caller (clibd_gemm_bf16_nt_ws), a launch whose last round is at most half full and whose
    // contraction is long cuts that round's tiles into K-slices over the idle CUs
    q.sk_parts = 0;
    if (p.sk_ws != nullptr && p.sk_flags != nullptr && !diag) {
        const SkPlan sk = plan_stream_k_tail(tiles, num_cus, nk);
        const void* fsk = sk.parts >= 2 ? kernel_ptr_sk(kind, lora, p.ep.bias != nullptr) : nullptr;
        if (fsk != nullptr) {
            static const bool sk_attr_ok = [] {
                bool ok = true;
                for (int k = 0; k < EPI_NUM_KINDS; ++k)
                    for (int v = 0; v < 4; ++v) {
                        const void* f = kernel_ptr_sk(k, (v & 1) != 0, (v & 2) != 0);
                        if (f != nullptr) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS) == hipSuccess;
                    }
                return ok;
            }();
            if (sk_attr_ok) {
                q.sk_parts = sk.parts; q.sk_tail = sk.tail; q.sk_first = sk.first; q.sk_nk_part = sk.nk_part;
                ntiles_i = sk.first + sk.tail * sk.parts;                      // work items: whole tiles, then the tail's slices
                const int gsk = ntiles_i < num_cus ? ntiles_i : num_cus;
                if (hipLaunchKernel(fsk, dim3((unsigned)gsk), dim3(G256_THREADS), args, G256_LDS, stream) != hipSuccess) return false;
                return true;
            }
        }
    }
    if (hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G256_THREADS), args, G256_LDS, stream) != hipSuccess) return false;
    return true;
}

size_t gemm256_tail_workspace_bytes(int M, int N, int K) {
    if (M < 1024 || N <= 0 || K <= 0 || N % T_N != 0 || K % T_K != 0) return 0;
    const long long tiles = (long long)((M + T_M - 1) / T_M) * (N / T_N);
    if (tiles < 128) return 0;
    const SkPlan sk = plan_stream_k_tail(tiles, device_cus(), K / T_K);
    if (sk.parts < 2) return 0;
    return G256_SK_FLAG_BYTES + (size_t)sk.tail * (size_t)(sk.parts - 1) * (size_t)(T_M * T_N) * sizeof(float);
}

// Split-K with a partials workspace: partials[s] (fp32 [M, N], row stride N) = A[:, ks] . W[:, ks]^T for K-slice s.
// Returns the number of splits used (>= 1) or 0 when the shape is not one this kernel takes.
int gemm256_splitk_launch(const GemmParams& p0, float* partials, size_t partials_elems, hipStream_t stream) {
    GemmParams q = p0;
    const int nk = q.K / T_K;
    if (q.K % T_K != 0 || nk < 8 || (nk & 1) || q.N % T_N != 0 || q.M < 1) return 0;
    if ((unsigned long long)q.M * q.lda * 2ull >= (1ull << 32) || (unsigned long long)q.N * q.ldw * 2ull >= (1ull << 32)) return 0;
    q.tiles_m = (q.M + T_M - 1) / T_M;
    q.tiles_n = q.N / T_N;
    const int tiles = q.tiles_m * q.tiles_n;
    static const int num_cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        }
        return n;
    }();
    int nks = 0;
    const int splits = plan_k_slices(nk, tiles, num_cus, &nks);   // nk is even and >= 8: never refuses
    if ((size_t)splits * (size_t)q.M * (size_t)q.N > partials_elems) return 0;
    q.splits = splits;
    q.nk_split = nks;
    q.band = 4;
    q.split_stride = (long long)q.M * q.N;
    q.ep = clibd_gemm_epilogue{};
    q.ep.out_f32 = partials;
    q.ep.ld_out_f32 = q.N;
    q.ep.split_k = 1;
    static const bool attr_ok = hipFuncSetAttribute(kernel_ptr(EPI_SPLITK_F32, false, false, false), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    G256_LDS) == hipSuccess;
    if (!attr_ok) return 0;
    int nitems = tiles * splits, skew = 0;
    long long* stamp = nullptr;
    const int grid = nitems < num_cus ? nitems : num_cus;
    void* args[] = {(void*)&q, (void*)&nitems, (void*)&skew, (void*)&stamp};
    if (hipLaunchKernel(kernel_ptr(EPI_SPLITK_F32, false, false, false), dim3((unsigned)grid), dim3(G256_THREADS), args, G256_LDS, stream) !=
        hipSuccess)
        return 0;
    return splits;
}

}  // namespace clibd

#ifdef CLIBD_GEMM_DIAG
// Diagnostic hook for tools/ (diagnostic build only): device buffer of [256 workgroups][16 tiles][2 wave groups][8] int64 stamps.
extern "C" void clibd_debug_set_gemm_stamps(void* device_buffer) { clibd::g_stamp_buffer = (long long*)device_buffer; }
#endif
