// Experiment: a SPLIT-ROLE bf16 GEMM workgroup — four COMPUTE waves (ds_read + MFMA + barrier: no
// vector-memory instruction in their K loop) and four MEMORY waves (every LDS-DMA of the ring, nothing else) on a 256 x 128 tile —
// measured against the product's eight-wave 256 x 256 kernel (libclibd_hip.so, clibd_gemm_bf16_nt, epilogue kind 1) in ONE process,
// interleaved, on the step's two extreme shapes.  DESIGN.md §7 item 1 sketches the design; this program answers its open question:
// does a main loop whose LDS-DMA runs on waves of its own hold the MFMA rate of the eight-wave kernel although the smaller tile
// needs 48 KiB of operands per 1024 MFMA cycles (47 B/clk/CU against 32)?
//
// Geometry.  Tile 256 rows (A rows: the MFMA "Q" operand) x 128 columns (W rows: "P").  Compute wave wn (0..3) owns rows
// 64 wn .. 64 wn + 63 and all 128 columns = 4 Q tiles x 8 P tiles = 128 accumulators (AGPRs): exactly one wave group of gemm256.hip.
// P tile nt = 4 hm + t carries tile-local column 8 c + nt, so a lane (c = lane & 15, g = lane >> 4) owns 8 contiguous columns of rows
// 4 g + r.  A K-tile (64 k = 128-byte rows, 16-byte chunk index XOR (row & 7)) is staged as four half-tiles
//     j = 0  P_hm0   64 rows   8 KiB   LDS row 16 t + c  <- W row n0 + 8 c + t
//     j = 1  Q_hn0  128 rows  16 KiB   LDS row 32 wn + 16 t + c <- A row m0 + 64 wn + 16 t + c        (t = 0, 1)
//     j = 2  Q_hn1  128 rows  16 KiB   the same + 32 rows
//     j = 3  P_hm1   64 rows   8 KiB   W row n0 + 8 c + 4 + t
// two stages = 96 KiB (the 64 KiB left are the bf16 stash of step B).  A K-tile is four phases of 16 MFMAs per compute wave:
// quadrants (hm0,hn0) (hm0,hn1) (hm1,hn1) (hm1,hn0); every phase reads ONE operand half for the NEXT phase into registers
// (4 / 8 / 8 / 4 ds_read_b128, one after each of the phase's first MFMAs): Q_hn1, P_hm1, the next K-tile's P_hm0, its Q_hn0 (which
// goes into the registers Q_hn1 just vacated: 96 fragment registers in all).
//
// Half-tile stream L_h (h = 4 u + j, running on across this workgroup's tiles).  ONE barrier B_g per phase g, joined by all eight
// waves.  Memory wave: before B_g it waits (counted vmcnt) until ITS pieces of L_{<= g+2} have landed; after B_g it issues L_{g+D}
// (D = 8: the slot it refills, L_{g}'s, was last read in phase g - 2, and those reads were awaited before B_{g-1}).  Compute wave:
// awaits the fragments it requested in phase g - 1, passes B_g, then requests phase g + 1's fragments (half-tile L_{g+2}: landed
// and published by B_g) between the MFMAs of phase g.  A memory wave carries 2 pieces of a P half and 4 of a Q half.
//
//   hipcc --offload-arch=gfx950 -O3 -o gemmsr gemmsr.hip -ldl && ./gemmsr <path to libclibd_hip.so> [seconds per arm]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/time.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include <glob.h>
#include <unistd.h>
#include <cctype>
#include "../../include/clibd_hip.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int PH_B = 8192, QH_B = 16384, STG_B = 49152;
constexpr int OFF_P0 = 0, OFF_Q0 = 8192, OFF_Q1 = 24576, OFF_P1 = 40960;
constexpr int STASH_OFF = 2 * STG_B;          // 96 KiB
constexpr int LDS_PLAIN = 2 * STG_B, LDS_STASH = 2 * STG_B + 65536;

struct PS {
    const char* A; const char* W; unsigned short* out;
    int M, N, K, lda2, ldw2, ldo;      // lda2 / ldw2: row strides in BYTES; ldo in elements
    int tiles_m, tiles_n, ntiles, band;
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, b2));
}
// the product kernel's order: an XCD (id & 7) walks a contiguous range of tile ids; ids run m-fastest inside bands of `band` m-tiles
__device__ __forceinline__ void tile_xy(const PS& p, int bid, int& m0, int& n0) {
    const int q = p.ntiles >> 3;               // ntiles % 8 == 0 (host-checked)
    bid = (bid & 7) * q + (bid >> 3);
    const int band_id = bid / (p.band * p.tiles_n);
    const int band_m0 = band_id * p.band;
    const int band_h = min(p.band, p.tiles_m - band_m0);
    const int in_band = bid - band_id * p.band * p.tiles_n;
    m0 = (band_m0 + in_band % band_h) * 256;
    n0 = (in_band / band_h) * 128;
}

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// pieces a memory wave has in flight for half-tiles L_{A} .. L_{B} (j = h & 3: P halves PP pieces, Q halves QP)
template <int A, int B, int PP, int QP> constexpr int pieces() {
    int n = 0;
    for (int h = A; h <= B; ++h) n += ((h & 3) == 1 || (h & 3) == 2) ? QP : PP;
    return n;
}

constexpr int FLAG_BYTES = 2048;     // SYNC 2: landed[4] at +0 (one dword per DMA wave), consumed: 4 x 256 B at +256 (lane-spread writes)
constexpr int SPIN_LIMIT = 1 << 16;  // a flag wait gives up after this many polls (wrong results instead of a hung GPU)

// MODE 0: plain epilogue (the compute waves store their rows after the K loop, no MFMA under them — the product kernel's scheme)
// MODE 1: the same with s_memtime stamps (wave 0 = compute, wave 4 = memory; [wg][tile < 16][role][4])
// MODE 2: step B, one-output kind: the compute waves drop the finished tile as bf16 into a 64-KiB LDS stash (16 ds_write_b128 per
//         wave) and go on; memory waves 2, 3 turn into STORE waves that drain the stash under the next tile's MFMAs (ds_read_b128 +
//         global_store_dwordx4, whole 256-byte row segments), memory waves 0, 1 carry all the LDS-DMA (4 / 8 pieces per half).
// SYNC 0: one workgroup barrier B_g per phase.   Before B_g a DMA wave waits for its pieces of L_{<= g+2}; after it, issues L_{g+8}.
// SYNC 1: one barrier per TWO phases (g even): before it L_{<= g+3} must have landed; after it L_{g+8}, L_{g+9} go out.
// SYNC 2: NO barrier in the K loop.  DMA wave mw publishes landed[mw] = number of half-tiles whose pieces it has seen land; a compute
//         wave polls the four counters one phase ahead (the ds_read rides behind the phase's fragment reads; its result is tested two
//         MFMAs into the next phase, before that phase's fragment reads) and publishes consumed[wn] = g once the reads of the phases
//         before g have returned; a DMA wave refills L_h's slot (with L_{h+8}) only when all four consumed counters are >= h - 1.
template <int MODE, int SYNC, int DK = 0, int OPT = 0>   // DK 4 (timing only): every LDS-DMA piece reads 1 KiB CONTIGUOUS (operands as if stored in [rows / 8][K / 64][8][64] tiles).  // OPT bit 0: compute waves at s_setprio 3; bit 1: fragment reads start after the phase's FIRST MFMA (not the third).  // DK: how the memory waves move operands: 0 LDS-DMA, 1 registers (global_load_dwordx4 -> ds_write_b128), 2 not at all (timing only)
__global__ __launch_bounds__(512, 1) void gemmsr_kernel(PS p, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 8;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nk = p.K / 64;                       // even, >= 4 (host-checked)
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) ++my_tiles;
    constexpr bool STASH = (MODE == 2);
    static_assert(!(STASH && SYNC == 2), "the flag area and the stash do not fit the LDS together in this experiment");
    constexpr int NDMA = STASH ? 2 : 4;            // waves that carry LDS-DMA
    constexpr int PP = 8 / NDMA, QP = 16 / NDMA;   // pieces per wave of a P / Q half
    const unsigned flag0 = lds0 + (unsigned)(2 * STG_B);   // SYNC 2 (plain epilogue only: the stash would sit here)
    if constexpr (SYNC == 2) {
        *(unsigned*)(smem + 2 * STG_B + 4 * threadIdx.x) = 0u;   // 512 dwords = FLAG_BYTES
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#define STAMP(role, k)                                                                                            \
    do {                                                                                                          \
        if (MODE == 1 && stamps != nullptr && tile_i < 16 && lane == 0 && (w == 0 || w == 4))                     \
            stamps[(((size_t)blockIdx.x * 16 + tile_i) * 2 + (role)) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

    if (w >= 4 && (w - 4) < NDMA) {
        // =============================================================================================== DMA wave
        const int mw = w - 4;
        const int prow = lane >> 3;
        const unsigned chunk16 = (unsigned)((lane & 7) ^ prow) * 16u;
        unsigned offP[PP], offQ[QP];
        int tile_is = blockIdx.x, u_is = 0, im0, in0;
        auto set_src = [&]() {
            tile_xy(p, tile_is, im0, in0);
            if constexpr (DK == 3) { im0 = 256 * (int)(blockIdx.x & 7); in0 = 0; }   // timing only: every workgroup re-reads ONE L2-resident A panel per XCD and one W panel
            if constexpr (DK == 4) {   // timing only: 1-KiB-tiled operands — piece q of a half = row group (8 rows) x K-tile, contiguous
                const unsigned sw16 = (unsigned)prow * 128u + chunk16;
#pragma unroll
                for (int i = 0; i < PP; ++i) offP[i] = (unsigned)(in0 / 8 + PP * mw + i) * (unsigned)nk * 1024u + sw16;
#pragma unroll
                for (int i = 0; i < QP; ++i) {
                    const int q = QP * mw + i;
                    offQ[i] = (unsigned)(im0 / 8 + 8 * (q >> 2) + (q & 3)) * (unsigned)nk * 1024u + sw16;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < PP; ++i) {   // P piece q = PP mw + i: LDS rows 8 q + prow = 16 t + c
                const int row = 8 * (PP * mw + i) + prow;
                offP[i] = (unsigned)(in0 + 8 * (row & 15) + (row >> 4)) * (unsigned)p.ldw2 + chunk16;
            }
#pragma unroll
            for (int i = 0; i < QP; ++i) {   // Q piece q = QP mw + i: LDS rows 8 q + prow = 32 wn + 16 t + c <- A row 64 wn + 16 t + c
                const int row = 8 * (QP * mw + i) + prow;
                offQ[i] = (unsigned)min(im0 + 64 * (row >> 5) + (row & 31), p.M - 1) * (unsigned)p.lda2 + chunk16;
            }
        };
        set_src();
#define PIECE(voff, sbase, dst)                                                                                   \
    do { if constexpr (DK == 0 || DK == 3 || DK == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0"); } while (0)
#define ISSUE(J)                                                                                                  \
    do {                                                                                                          \
        const unsigned st_ = lds0 + (unsigned)((u_is & 1) * STG_B);                                               \
        if ((J) == 0 || (J) == 3) {                                                                               \
            const char* sb_ = (DK == 4) ? p.W + (size_t)u_is * 1024 + ((J) == 3 ? (size_t)8 * nk * 1024 : 0)      \
                                        : p.W + (size_t)u_is * 128 + ((J) == 3 ? (size_t)p.ldw2 * 4 : 0);          \
            const unsigned d_ = st_ + (unsigned)((J) == 3 ? OFF_P1 : OFF_P0) + (unsigned)(mw * PP) * 1024u;       \
            _Pragma("unroll") for (int i = 0; i < PP; ++i) PIECE(offP[i], sb_, d_ + 1024u * i);                   \
        } else {                                                                                                  \
            const char* sb_ = (DK == 4) ? p.A + (size_t)u_is * 1024 + ((J) == 2 ? (size_t)4 * nk * 1024 : 0)      \
                                        : p.A + (size_t)u_is * 128 + ((J) == 2 ? (size_t)p.lda2 * 32 : 0);         \
            const unsigned d_ = st_ + (unsigned)((J) == 2 ? OFF_Q1 : OFF_Q0) + (unsigned)(mw * QP) * 1024u;       \
            _Pragma("unroll") for (int i = 0; i < QP; ++i) PIECE(offQ[i], sb_, d_ + 1024u * i);                   \
        }                                                                                                         \
        if ((J) == 3) {                                                                                           \
            if (++u_is == nk) {   /* the stream runs on into this workgroup's next tile; at the very end it re-fetches the last */ \
                u_is = 0;         /* tile (the slots it fills are dead and every wait count stays the same) */    \
                const int nxt_ = tile_is + (int)gridDim.x;                                                        \
                if (nxt_ < p.ntiles) tile_is = nxt_;                                                              \
                set_src();                                                                                        \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
        // SYNC 2 helpers: publish the number of half-tiles seen to land; wait until every compute wave has consumed phase `need`
        const unsigned my_landed = flag0 + 4u * (unsigned)mw;
        const unsigned cons_poll = flag0 + 256u + 256u * (unsigned)(lane & 3);
#define PUBLISH_LANDED(V)                                                                                         \
    do { if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(my_landed), "v"((unsigned)(V)) : "memory"); } while (0)
#define WAIT_CONSUMED(NEED)                                                                                       \
    do {                                                                                                          \
        const int need_ = (NEED);                                                                                 \
        if (need_ > 0) {                                                                                          \
            for (int it_ = 0; it_ < SPIN_LIMIT; ++it_) {                                                          \
                unsigned c_;                                                                                      \
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c_) : "v"(cons_poll) : "memory");   \
                const int m_ = min(min(__builtin_amdgcn_readlane((int)c_, 0), __builtin_amdgcn_readlane((int)c_, 1)),  \
                                   min(__builtin_amdgcn_readlane((int)c_, 2), __builtin_amdgcn_readlane((int)c_, 3)));  \
                if (m_ >= need_) break;                                                                           \
                __builtin_amdgcn_s_sleep(1);                                                                      \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
        if constexpr (DK == 1) {
            // ---- register staging (SYNC 0, plain epilogue): half-tile L_h is requested with global_load_dwordx4 at step h - 12, sits in
            // registers (slot h & 3) for four steps, and is written to its LDS slot with ds_write_b128 at step h - 8 (after B_{h-8}, when
            // L_{h-8}'s reads are over); the writes are awaited before B_{h-7}; the compute waves read L_h in phase h - 2.
            static_assert(SYNC == 0 && !STASH, "register staging: barrier per phase, plain epilogue");
            u32x4 buf[4][QP];
            const unsigned lane16 = (unsigned)lane * 16u;
            int uw = 0;   // K-tile parity of the half-tile being WRITTEN (the load stream's u_is runs ahead)
#define GLOAD(J, SLOT)                                                                                            \
    do {                                                                                                          \
        if ((J) == 0 || (J) == 3) {                                                                               \
            const char* sb_ = (DK == 4) ? p.W + (size_t)u_is * 1024 + ((J) == 3 ? (size_t)8 * nk * 1024 : 0)      \
                                        : p.W + (size_t)u_is * 128 + ((J) == 3 ? (size_t)p.ldw2 * 4 : 0);          \
            _Pragma("unroll") for (int i = 0; i < PP; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(buf[SLOT][i]) : "v"(offP[i]), "s"(sb_) : "memory"); \
        } else {                                                                                                  \
            const char* sb_ = (DK == 4) ? p.A + (size_t)u_is * 1024 + ((J) == 2 ? (size_t)4 * nk * 1024 : 0)      \
                                        : p.A + (size_t)u_is * 128 + ((J) == 2 ? (size_t)p.lda2 * 32 : 0);         \
            _Pragma("unroll") for (int i = 0; i < QP; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(buf[SLOT][i]) : "v"(offQ[i]), "s"(sb_) : "memory"); \
        }                                                                                                         \
        if ((J) == 3) {                                                                                           \
            if (++u_is == nk) { u_is = 0; const int nxt_ = tile_is + (int)gridDim.x; if (nxt_ < p.ntiles) tile_is = nxt_; set_src(); } \
        }                                                                                                         \
    } while (0)
#define LWRITE(J, SLOT)                                                                                           \
    do {                                                                                                          \
        const unsigned st_ = lds0 + (unsigned)(uw * STG_B) + lane16;                                              \
        if ((J) == 0 || (J) == 3) {                                                                               \
            const unsigned d_ = st_ + (unsigned)((J) == 3 ? OFF_P1 : OFF_P0) + (unsigned)(mw * PP) * 1024u;       \
            _Pragma("unroll") for (int i = 0; i < PP; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"(d_ + 1024u * i), "v"(buf[SLOT][i]) : "memory"); \
        } else {                                                                                                  \
            const unsigned d_ = st_ + (unsigned)((J) == 2 ? OFF_Q1 : OFF_Q0) + (unsigned)(mw * QP) * 1024u;       \
            _Pragma("unroll") for (int i = 0; i < QP; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"(d_ + 1024u * i), "v"(buf[SLOT][i]) : "memory"); \
        }                                                                                                         \
        if ((J) == 3) uw ^= 1;                                                                                    \
    } while (0)
#define TIE(SLOT) asm volatile("" : "+v"(buf[SLOT][0]), "+v"(buf[SLOT][1]), "+v"(buf[SLOT][QP - 2]), "+v"(buf[SLOT][QP - 1]))
            // prologue: L_0 .. L_7 through the registers one after the other, then L_8 .. L_11 in flight
            GLOAD(0, 0); WAITV(0); TIE(0); LWRITE(0, 0);  GLOAD(1, 1); WAITV(0); TIE(1); LWRITE(1, 1);
            GLOAD(2, 2); WAITV(0); TIE(2); LWRITE(2, 2);  GLOAD(3, 3); WAITV(0); TIE(3); LWRITE(3, 3);
            GLOAD(0, 0); WAITV(0); TIE(0); LWRITE(0, 0);  GLOAD(1, 1); WAITV(0); TIE(1); LWRITE(1, 1);
            GLOAD(2, 2); WAITV(0); TIE(2); LWRITE(2, 2);  GLOAD(3, 3); WAITV(0); TIE(3); LWRITE(3, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GLOAD(0, 0); GLOAD(1, 1); GLOAD(2, 2); GLOAD(3, 3);
            BARRIER();   // B_pre
            const int phases = my_tiles * 4 * nk;
#pragma unroll 1
            for (int g = 0; g < phases; g += 4) {
                // step g + GQ: L_{g+GQ+8} (slot GQ) must have arrived in registers: younger loads L_{g+GQ+9 .. +11}
                WAITV((pieces<1, 3, PP, QP>())); TIE(0); BARRIER(); LWRITE(0, 0); GLOAD(0, 0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WAITV((pieces<2, 4, PP, QP>())); TIE(1); BARRIER(); LWRITE(1, 1); GLOAD(1, 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WAITV((pieces<3, 5, PP, QP>())); TIE(2); BARRIER(); LWRITE(2, 2); GLOAD(2, 2); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WAITV((pieces<4, 6, PP, QP>())); TIE(3); BARRIER(); LWRITE(3, 3); GLOAD(3, 3); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef GLOAD
#undef LWRITE
#undef TIE
        } else {
        // prologue: L_0 .. L_7; then L_0, L_1 must have landed (in flight: L_2 .. L_7)
        ISSUE(0); ISSUE(1); ISSUE(2); ISSUE(3); ISSUE(0); ISSUE(1); ISSUE(2); ISSUE(3);
        WAITV((pieces<2, 7, PP, QP>()));
        BARRIER();   // B_pre
        const int phases = my_tiles * 4 * nk;
        int tile_i = 0, ph_in_tile = 0;
        long long acc_v = 0, acc_b = 0;
#pragma unroll 1
        for (int g = 0; g < phases; g += 4) {
            if (ph_in_tile == 0) STAMP(1, 0);
            if constexpr (MODE == 1 && SYNC == 0) {   // the same with the cycles spent waiting for data / at the barrier summed per tile
#define TIMED_STEP(GQ)                                                                                            \
    do {                                                                                                          \
        const long long a_ = (long long)__builtin_amdgcn_s_memtime();                                             \
        WAITV((pieces<(GQ) + 3, (GQ) + 7, PP, QP>()));                                                            \
        const long long b_ = (long long)__builtin_amdgcn_s_memtime();                                             \
        BARRIER();                                                                                                \
        const long long c_ = (long long)__builtin_amdgcn_s_memtime();                                             \
        acc_v += b_ - a_; acc_b += c_ - b_;                                                                       \
        ISSUE(GQ);                                                                                                \
    } while (0)
                TIMED_STEP(0); TIMED_STEP(1); TIMED_STEP(2); TIMED_STEP(3);
#undef TIMED_STEP
            } else if constexpr (SYNC == 0) {
                WAITV((pieces<0 + 3, 0 + 7, PP, QP>())); BARRIER(); ISSUE(0);
                WAITV((pieces<1 + 3, 1 + 7, PP, QP>())); BARRIER(); ISSUE(1);
                WAITV((pieces<2 + 3, 2 + 7, PP, QP>())); BARRIER(); ISSUE(2);
                WAITV((pieces<3 + 3, 3 + 7, PP, QP>())); BARRIER(); ISSUE(3);
            } else if constexpr (SYNC == 1) {
                WAITV((pieces<0 + 4, 0 + 7, PP, QP>())); BARRIER(); ISSUE(0); ISSUE(1);
                WAITV((pieces<2 + 4, 2 + 7, PP, QP>())); BARRIER(); ISSUE(2); ISSUE(3);
            } else {
                WAITV((pieces<0 + 3, 0 + 7, PP, QP>())); PUBLISH_LANDED(g + 3); WAIT_CONSUMED(g - 1); ISSUE(0);
                WAITV((pieces<1 + 3, 1 + 7, PP, QP>())); PUBLISH_LANDED(g + 4); WAIT_CONSUMED(g);     ISSUE(1);
                WAITV((pieces<2 + 3, 2 + 7, PP, QP>())); PUBLISH_LANDED(g + 5); WAIT_CONSUMED(g + 1); ISSUE(2);
                WAITV((pieces<3 + 3, 3 + 7, PP, QP>())); PUBLISH_LANDED(g + 6); WAIT_CONSUMED(g + 2); ISSUE(3);
            }
            ph_in_tile += 4;
            if (ph_in_tile == 4 * nk) {
                STAMP(1, 1);
                if (MODE == 1 && stamps != nullptr && tile_i < 16 && lane == 0 && w == 4) {
                    stamps[(((size_t)blockIdx.x * 16 + tile_i) * 2 + 1) * 4 + 2] = acc_v;
                    stamps[(((size_t)blockIdx.x * 16 + tile_i) * 2 + 1) * 4 + 3] = acc_b;
                }
                acc_v = 0; acc_b = 0;
                ph_in_tile = 0; ++tile_i;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the stream ran 8 half-tiles ahead: drain before the LDS is released
        if constexpr (STASH) BARRIER();                    // F: matches the compute / store waves' final barrier
        }
#undef ISSUE
#undef PIECE
    } else if (w >= 4) {
        // =============================================================================================== STORE wave (MODE 2 only)
        // Stash image: [256 rows][128 columns] bf16, plain row-major 256-byte rows (a row spans all 64 banks once: the compute waves'
        // ds_write_b128 — 8 lanes = 128 contiguous bytes per LDS cycle — and these ds_read_b128 are conflict-free as they stand).
        // Store wave sw drains rows 128 sw .. 128 sw + 127 of the PREVIOUS tile: in each of the next tile's first 32 phases one
        // instruction pair = 4 rows x 256 bytes (lane -> row 4 ph + (lane >> 4), chunk lane & 15).  The tile's last barrier
        // (phase >= 32) orders these reads before the compute waves overwrite the stash.
        const int sw = w - 4 - NDMA;
        const int phases = my_tiles * 4 * nk;
        int pm0 = 0, pn0 = 0, tile = blockIdx.x, ph_in_tile = 0;
        bool has_prev = false;
        BARRIER();   // B_pre
#pragma unroll 1
        for (int g = 0; g < phases; ++g) {
            if (SYNC == 0 || (g & 1) == 0) BARRIER();
            if (has_prev && ph_in_tile < 32) {   // (the compute waves awaited their stash writes before this tile's B_0)
                const int row = 128 * sw + 4 * ph_in_tile + (lane >> 4);
                const int ch = lane & 15;
                u32x4 d_;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d_) : "v"(lds0 + (unsigned)STASH_OFF + (unsigned)(row * 256 + (ch << 4))));
                *(u32x4*)(p.out + (size_t)(pm0 + row) * p.ldo + pn0 + 8 * ch) = d_;
            }
            if (++ph_in_tile == 4 * nk) {
                ph_in_tile = 0;
                tile_xy(p, tile, pm0, pn0);
                has_prev = true;
                tile += (int)gridDim.x;
            }
        }
        if (has_prev) {   // the last tile has no next tile to hide under
            BARRIER();    // F: the compute waves' last stash writes have landed
#pragma unroll 1
            for (int q = 0; q < 32; ++q) {
                const int row = 128 * sw + 4 * q + (lane >> 4);
                const int ch = lane & 15;
                u32x4 d_;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d_) : "v"(lds0 + (unsigned)STASH_OFF + (unsigned)(row * 256 + (ch << 4))));
                *(u32x4*)(p.out + (size_t)(pm0 + row) * p.ldo + pn0 + 8 * ch) = d_;
            }
        }
    } else {
        // =============================================================================================== compute wave
        const int wn = w;
        const int frow = lane & 15, fch = lane >> 4;
        unsigned aP[2][2], aQ[2][2];   // [stage][kk]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                aP[s][kk] = lds0 + (unsigned)(s * STG_B) + (unsigned)tile_off(frow, 4 * kk + fch);
                aQ[s][kk] = lds0 + (unsigned)(s * STG_B + OFF_Q0) + (unsigned)tile_off(32 * wn + frow, 4 * kk + fch);
            }
        f32x4 acc[2][4][2][2];         // [hm][t][hn][n]: 128 AGPRs
        bf16x8 Pa[4][2], Pb[4][2], X[2][2], Y[2][2];   // [tile][kk]: P_hm0, P_hm1, and the two Q register sets (roles alternate per K-tile)
        // SYNC 2
        u32x4 poll = (u32x4){0u, 0u, 0u, 0u};
        const unsigned my_cons = flag0 + 256u + 256u * (unsigned)wn + 4u * (unsigned)lane;
        int gph = 0;                   // global phase index of this workgroup
#define DSR(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define MFMA_ASM(ACC, QA, PB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(QA), "v"(PB))
#define MFMA_ASM0(ACC, QA, PB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(QA), "v"(PB))
        // load #i of a P half (i = 0..7: tile i >> 1, kk i & 1) / of a Q half (i = 0..3)
#define LOADP_I(DST, S, OFF, i) DSR(DST[(i) >> 1][(i) & 1], aP[S][(i) & 1], (OFF) + 2048 * ((i) >> 1))
#define LOADQ_I(DST, S, OFF, i) DSR(DST[(i) >> 1][(i) & 1], aQ[S][(i) & 1], (OFF) + 2048 * ((i) >> 1))
#define WAITP(F)                                                                                                  \
    do {                                                                                                          \
        if constexpr (SYNC == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1]), "+v"(poll)); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1])); \
    } while (0)
#define WAITQ(F)                                                                                                  \
    do {                                                                                                          \
        if constexpr (SYNC == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(poll)); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]));      \
    } while (0)
#define WAITPOLL() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(poll))
#define POLL_MIN() (int)min(min(poll[0], poll[1]), min(poll[2], poll[3]))
        // One phase (PQ = 0..3 inside the K-tile): 16 MFMAs on (PF, QF) into quadrant (hm, hn); LK: which half is requested for the
        // next phase (1 = Q: 4 reads, 2 = P: 8 reads), one read after each MFMA from the third on.  The caller has awaited this
        // phase's fragments (and the flag poll issued in the previous phase).
#define PHASE(PQ, hm, hn, PF, QF, ZC, LK, LDST, LS, LOFF)                                                         \
    do {                                                                                                          \
        if (SYNC == 0 || (SYNC == 1 && ((PQ) & 1) == 0)) BARRIER();                                               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
                _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                   \
                    if ((ZC) && kk == 0) MFMA_ASM0(acc[hm][t][hn][n], QF[n][kk], PF[t][kk]);                      \
                    else MFMA_ASM(acc[hm][t][hn][n], QF[n][kk], PF[t][kk]);                                       \
                    const int i_ = 8 * kk + 2 * t + n;                                                            \
                    if (SYNC == 2 && i_ == 0) {   /* the reads of every earlier phase have returned: publish; then test the poll */ \
                        asm volatile("ds_write_b32 %0, %1" :: "v"(my_cons), "v"((unsigned)gph) : "memory");       \
                    }                                                                                             \
                    if (SYNC == 2 && i_ == 1) {                                                                   \
                        int have_ = __builtin_amdgcn_readfirstlane(POLL_MIN());                                   \
                        if (__builtin_expect(have_ < gph + 3, 0)) {                                               \
                            for (int it_ = 0; it_ < SPIN_LIMIT && have_ < gph + 3; ++it_) {                       \
                                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(poll) : "v"(flag0) : "memory"); \
                                have_ = __builtin_amdgcn_readfirstlane(POLL_MIN());                               \
                            }                                                                                     \
                        }                                                                                         \
                    }                                                                                             \
                    constexpr int r0_ = (OPT & 2) ? 0 : 2;                                                        \
                    if ((LK) == 1 && i_ >= r0_ && i_ < r0_ + 4) LOADQ_I(LDST, LS, LOFF, i_ - r0_);                \
                    if ((LK) == 2 && i_ >= r0_ && i_ < r0_ + 8) LOADP_I(LDST, LS, LOFF, i_ - r0_);                \
                    if (SYNC == 2 && i_ == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(poll) : "v"(flag0));     \
                }                                                                                                 \
        ++gph;                                                                                                    \
    } while (0)
        // K-tile of stage S whose Q_hn0 sits in QA: Q_hn1 goes to QB, and after phase 2 (QB's last use) the next K-tile's Q_hn0
        // goes to QB as well — so the caller alternates (X, Y), (Y, X)
#define KTILE(S, QA, QB, ZC)                                                                                      \
    do {                                                                                                          \
        WAITP(Pa); WAITQ(QA);                                                                                     \
        PHASE(0, 0, 0, Pa, QA, ZC, 1, QB, S, OFF_Q1 - OFF_Q0);              /* (hm0,hn0); request Q_hn1 -> QB */   \
        WAITQ(QB);                                                                                                \
        PHASE(1, 0, 1, Pa, QB, ZC, 2, Pb, S, OFF_P1);                       /* (hm0,hn1); request P_hm1 -> Pb */   \
        WAITP(Pb);                                                                                                \
        PHASE(2, 1, 1, Pb, QB, ZC, 2, Pa, (S) ^ 1, OFF_P0);                 /* (hm1,hn1); request next P_hm0 -> Pa */ \
        if (SYNC == 2) WAITPOLL();   /* (phases 1..3 of the flag protocol: the poll must have returned before it is tested) */ \
        PHASE(3, 1, 0, Pb, QA, ZC, 1, QB, (S) ^ 1, 0);                      /* (hm1,hn0); request next Q_hn0 -> QB */ \
    } while (0)
        int tile = blockIdx.x, m0, n0, tile_i = 0;
        tile_xy(p, tile, m0, n0);
        BARRIER();   // B_pre: L_0, L_1 have landed
        if constexpr ((OPT & 1) != 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < 8; ++i) LOADP_I(Pa, 0, OFF_P0, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) LOADQ_I(X, 0, 0, i);
        if constexpr (SYNC == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(poll) : "v"(flag0));
        while (true) {
            STAMP(0, 0);
            KTILE(0, X, Y, true);
            KTILE(1, Y, X, false);
#pragma unroll 1
            for (int u = 2; u < nk; u += 2) {
                KTILE(0, X, Y, false);
                KTILE(1, Y, X, false);
            }
            STAMP(0, 1);
            // inline-asm MFMAs are invisible to hipcc's hazard recognizer: the accumulator reads below need the last MFMA's passes done
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            // ---- epilogue: lane (c = frow, g = fch) owns rows m0 + 64 wn + 32 hn + 16 n + 4 g + r, columns n0 + 8 c + (4 hm + t)
            if constexpr (STASH) {
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 64 * wn + 32 * hn + 16 * n + 4 * fch + r;
                            u32x4 d_;
                            d_[0] = pack2bf(acc[0][0][hn][n][r], acc[0][1][hn][n][r]);
                            d_[1] = pack2bf(acc[0][2][hn][n][r], acc[0][3][hn][n][r]);
                            d_[2] = pack2bf(acc[1][0][hn][n][r], acc[1][1][hn][n][r]);
                            d_[3] = pack2bf(acc[1][2][hn][n][r], acc[1][3][hn][n][r]);
                            asm volatile("ds_write_b128 %0, %1" :: "v"(lds0 + (unsigned)STASH_OFF + (unsigned)(row * 256 + (frow << 4))), "v"(d_) : "memory");
                        }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = m0 + 64 * wn + 32 * hn + 16 * n + 4 * fch + r;
                            u32x4 d_;
                            d_[0] = pack2bf(acc[0][0][hn][n][r], acc[0][1][hn][n][r]);
                            d_[1] = pack2bf(acc[0][2][hn][n][r], acc[0][3][hn][n][r]);
                            d_[2] = pack2bf(acc[1][0][hn][n][r], acc[1][1][hn][n][r]);
                            d_[3] = pack2bf(acc[1][2][hn][n][r], acc[1][3][hn][n][r]);
                            if (row < p.M) *(u32x4*)(p.out + (size_t)row * p.ldo + n0 + 8 * frow) = d_;
                        }
            }
            STAMP(0, 2);
            ++tile_i;
            const int nxt = tile + (int)gridDim.x;
            if (nxt >= p.ntiles) break;
            tile = nxt;
            tile_xy(p, tile, m0, n0);
        }
        if constexpr (STASH) BARRIER();   // F: publishes the last tile's stash to the store waves
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Pa[0][0]), "+v"(X[0][0]), "+v"(Y[0][0]));   // the fragment reads ran one K-tile ahead
        if constexpr (SYNC == 2) WAITPOLL();
        if constexpr (SYNC == 2) {   // release DMA waves still waiting to refill slots nobody will read
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(my_cons), "v"(0x7fffffffu) : "memory");
        }
#undef KTILE
#undef PHASE
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec + 1e-6 * t.tv_usec; }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Board {   // sysfs hwmon of THIS process's device (matched by PCI bus id): power1_input / power1_average (uW), freq1_input (Hz)
    std::string dir, pn;
    Board() {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) return;
        for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
        glob_t g;
        if (glob("/sys/class/drm/card*/device", 0, nullptr, &g) == 0) {
            for (size_t i = 0; i < g.gl_pathc && dir.empty(); ++i) {
                FILE* f = fopen((std::string(g.gl_pathv[i]) + "/uevent").c_str(), "r");
                if (!f) continue;
                char line[256]; bool mine = false;
                while (fgets(line, sizeof(line), f)) {
                    for (char* c = line; *c; ++c) *c = (char)tolower(*c);
                    if (!strncmp(line, "pci_slot_name=", 14) && !strncmp(line + 14, bus, strlen(bus))) mine = true;
                }
                fclose(f);
                if (!mine) continue;
                glob_t h;
                if (glob((std::string(g.gl_pathv[i]) + "/hwmon/hwmon*").c_str(), 0, nullptr, &h) == 0) {
                    for (size_t k = 0; k < h.gl_pathc && dir.empty(); ++k)
                        for (const char* n : {"power1_input", "power1_average"})
                            if (rd(std::string(h.gl_pathv[k]) + "/" + n) > 0) { dir = h.gl_pathv[k]; pn = n; break; }
                    globfree(&h);
                }
            }
            globfree(&g);
        }
    }
    static double rd(const std::string& path) {
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return -1;
        double v = -1;
        if (fscanf(f, "%lf", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
    double watts() const { return dir.empty() ? -1 : rd(dir + "/" + pn) / 1e6; }
    double mhz() const { return dir.empty() ? -1 : rd(dir + "/freq1_input") / 1e6; }
};

typedef int (*gemm_fn)(const void*, int, const void*, int, int, int, int, const clibd_gemm_epilogue*, void*);

typedef void (*kern_t)(PS, long long*);
struct Arm { const char* name; kern_t fn; int lds; };

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <libclibd_hip.so> [seconds per arm] [M]\n", argv[0]); return 2; }
    const double secs = argc > 2 ? atof(argv[2]) : 2.0;
    const int M = argc > 3 ? atoi(argv[3]) : 403456;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    gemm_fn ref = (gemm_fn)dlsym(lib, "clibd_gemm_bf16_nt");
    if (!ref) { fprintf(stderr, "clibd_gemm_bf16_nt not found\n"); return 2; }
    Board board;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int ncu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
    const Arm arms[] = {
        {"8-wave 256x256 product kernel (kind 1)", nullptr, 0},
        {"split-role, plain epilogue, barrier per phase", gemmsr_kernel<0, 0>, LDS_PLAIN},
        {"split-role, plain epilogue, barrier per 2 phases", gemmsr_kernel<0, 1>, LDS_PLAIN},
        {"split-role, plain epilogue, LDS flags (no barrier)", gemmsr_kernel<0, 2>, LDS_PLAIN + FLAG_BYTES},
        {"split-role, plain, barrier per phase, compute waves at s_setprio 3", gemmsr_kernel<0, 0, 0, 1>, LDS_PLAIN},
        {"split-role, plain, barrier per phase, reads from the first MFMA on", gemmsr_kernel<0, 0, 0, 2>, LDS_PLAIN},
        {"split-role, plain, barrier per 2, setprio 3 + early reads", gemmsr_kernel<0, 1, 0, 3>, LDS_PLAIN},
        {"split-role, plain, barrier per phase, REGISTER staging", gemmsr_kernel<0, 0, 1>, LDS_PLAIN},
        {"split-role, plain, barrier per phase, NO operand traffic (timing only)", gemmsr_kernel<0, 0, 2>, LDS_PLAIN},
        {"split-role, plain, barrier per phase, operands from ONE L2-hot tile (timing only)", gemmsr_kernel<0, 0, 3>, LDS_PLAIN},
        {"split-role, plain, barrier per phase, 1-KiB-TILED operand layout (timing only)", gemmsr_kernel<0, 0, 4>, LDS_PLAIN},
        {"split-role, stash + store waves, barrier per phase", gemmsr_kernel<2, 0>, LDS_STASH},
        {"split-role, stash + store waves, barrier per 2", gemmsr_kernel<2, 1>, LDS_STASH},
    };
    const Arm stamp_arms[] = {
        {"barrier per phase", gemmsr_kernel<1, 0>, LDS_PLAIN},
        {"barrier per 2 phases", gemmsr_kernel<1, 1>, LDS_PLAIN},
        {"LDS flags", gemmsr_kernel<1, 2>, LDS_PLAIN + FLAG_BYTES},
        {"barrier per phase, register staging", gemmsr_kernel<1, 0, 1>, LDS_PLAIN},
        {"barrier per phase, no operand traffic", gemmsr_kernel<1, 0, 2>, LDS_PLAIN},
    };
    constexpr int NA = sizeof(arms) / sizeof(arms[0]);
    for (const Arm& a : arms) if (a.fn) CK(hipFuncSetAttribute((const void*)a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds));
    for (const Arm& a : stamp_arms) CK(hipFuncSetAttribute((const void*)a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds));
    const int shapes[2][2] = {{768, 3072}, {3072, 768}};   // (N, K)
    for (int si = 0; si < 2; ++si) {
        const int N = shapes[si][0], K = shapes[si][1];
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        unsigned s = 12345u + si;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd();
        unsigned short *dA, *dW, *dRef, *dOut;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2));
        CK(hipMalloc(&dRef, (size_t)M * N * 2)); CK(hipMalloc(&dOut, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        if (((M / 256) * (N / 128)) % 8 || M % 256 || N % 256 || (K / 64) % 2 || K < 256 || 4 * (K / 64) < 34) { fprintf(stderr, "shape outside the experiment\n"); return 2; }
        PS p{(const char*)dA, (const char*)dW, dOut, M, N, K, K * 2, K * 2, N, M / 256, N / 128, (M / 256) * (N / 128), 4};
        clibd_gemm_epilogue ep;
        memset(&ep, 0, sizeof(ep));
        ep.split_k = 1; ep.ld_out_bf16 = N;
        long long* dstamps = nullptr;
        CK(hipMalloc(&dstamps, (size_t)ncu * 16 * 2 * 4 * sizeof(long long)));
        auto run = [&](int arm) {
            long long* nul = nullptr;
            if (arm == 0) { ep.out_bf16 = dRef; if (ref(dA, K, dW, K, M, N, K, &ep, st) != 0) { fprintf(stderr, "reference gemm failed\n"); exit(1); } }
            else hipLaunchKernelGGL(arms[arm].fn, dim3(ncu), dim3(512), arms[arm].lds, st, p, nul);
        };
        // ---- correctness: the product kernel against a host fp64 dot product on sampled entries, every arm against the product kernel on everything
        run(0);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned short> hR((size_t)M * N), hO((size_t)M * N);
        CK(hipMemcpy(hR.data(), dRef, hR.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
            s = s * 1664525u + 1013904223u; const size_t m = (t < 8) ? (size_t)(t < 4 ? t * 85 : M - 1 - (t - 4) * 77) : (s >> 4) % M;
            s = s * 1664525u + 1013904223u; const size_t n = (s >> 4) % N;
            double d = 0;
            for (int k = 0; k < K; ++k) d += (double)bf2f(hA[m * K + k]) * (double)bf2f(hW[n * K + k]);
            worst = std::max(worst, std::fabs((double)bf2f(hR[m * N + n]) - d) / (std::fabs(d) + 1.0));
        }
        printf("shape M=%d N=%d K=%d: product kernel max rel error vs fp64 on 4000 entries %.2e; elements differing from it:", M, N, K, worst);
        for (int arm = 1; arm < NA; ++arm) {
            CK(hipMemsetAsync(dOut, 0xff, (size_t)M * N * 2, st));
            run(arm);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            CK(hipMemcpy(hO.data(), dOut, hO.size() * 2, hipMemcpyDeviceToHost));
            size_t diff = 0, unwritten = 0;
            for (size_t i = 0; i < hO.size(); ++i) { diff += hO[i] != hR[i]; unwritten += (hO[i] == 0xffff && hR[i] != 0xffff); }
            printf(" [%d] %zu", arm, diff);
            if (diff) {   // where: histogram over (tile row block, k-phase-independent) — first mismatching tiles and their spread
                printf(" (unwritten %zu;", unwritten);
                int shown = 0;
                size_t colhist[16] = {0}, rowhist[16] = {0};
                for (size_t i = 0; i < hO.size(); ++i)
                    if (hO[i] != hR[i]) {
                        const size_t m = i / N, n = i % N;
                        ++colhist[n % 8]; ++rowhist[(m % 64) / 4];
                        if (shown < 8) { printf(" m=%zu n=%zu got %04x want %04x;", m, n, hO[i], hR[i]); ++shown; }
                    }
                printf(" by column%%8:"); for (int c = 0; c < 8; ++c) printf(" %zu", colhist[c]);
                printf(" by (row%%64)/4:"); for (int c = 0; c < 16; ++c) printf(" %zu", rowhist[c]);
                printf(")");

            }
        }
        printf(" of %zu\n", hR.size());
        fflush(stdout);
        // ---- stamps: cycles per phase of a compute wave, epilogue cycles
        for (const Arm& sa : stamp_arms) {
            CK(hipMemset(dstamps, 0, (size_t)ncu * 16 * 2 * 4 * sizeof(long long)));
            hipLaunchKernelGGL(sa.fn, dim3(ncu), dim3(512), sa.lds, st, p, dstamps);
            CK(hipStreamSynchronize(st));
            std::vector<long long> hs((size_t)ncu * 16 * 2 * 4);
            CK(hipMemcpy(hs.data(), dstamps, hs.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double kl = 0, epi = 0, gap = 0; int cnt = 0, gcnt = 0;
            for (int wg = 0; wg < ncu; ++wg)
                for (int ti = 1; ti < 12; ++ti) {   // steady tiles
                    const long long* c = &hs[(((size_t)wg * 16 + ti) * 2 + 0) * 4];
                    if (c[0] == 0 || c[2] == 0) continue;
                    kl += (double)(c[1] - c[0]); epi += (double)(c[2] - c[1]); ++cnt;
                    const long long* nx = &hs[(((size_t)wg * 16 + ti + 1) * 2 + 0) * 4];
                    if (nx[0]) { gap += (double)(nx[0] - c[2]); ++gcnt; }
                }
            {
                double wv = 0, wb = 0, tt = 0; int c2 = 0;
                for (int wg = 0; wg < ncu; ++wg)
                    for (int ti = 1; ti < 12; ++ti) {
                        const long long* c = &hs[(((size_t)wg * 16 + ti) * 2 + 1) * 4];
                        if (c[0] == 0 || c[1] == 0 || (c[2] == 0 && c[3] == 0)) continue;
                        wv += (double)c[2]; wb += (double)c[3]; tt += (double)(c[1] - c[0]); ++c2;
                    }
                if (c2) printf("  DMA wave 4 [%s]: of %.0f cycles per tile, %.0f (%.2f) waiting for its LDS-DMA to land (s_waitcnt vmcnt), %.0f (%.2f) at the barrier waiting for the compute waves\n",
                               sa.name, tt / c2, wv / c2, wv / tt, wb / c2, wb / tt);
            }
            if (cnt) printf("  stamps [%s] (s_memtime = shader clocks, mean over %d steady tiles): K loop %.0f cycles = %.1f per 16-MFMA phase (256 of MFMA issue; %d phases), epilogue %.0f, tile-to-tile gap %.0f; epilogue share %.3f\n",
                            sa.name, cnt, kl / cnt, kl / cnt / (4.0 * (K / 64)), 4 * (K / 64), epi / cnt, gcnt ? gap / gcnt : 0.0, (epi / cnt) / ((kl + epi) / cnt));
            fflush(stdout);
        }
        // ---- timing: interleaved arms, `secs` of back-to-back launches each, two rounds; board power / clock sampled at 20 Hz
        for (int round = 0; round < 2; ++round)
            for (int arm = 0; arm < NA; ++arm) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 3; ++i) run(arm);
                CK(hipStreamSynchronize(st));
                std::atomic<bool> stop{false};
                double wsum = 0, fsum = 0; int ns = 0;
                std::thread sampler([&] { while (!stop.load()) { const double wv = board.watts(), fv = board.mhz(); if (wv > 0) { wsum += wv; fsum += fv; ++ns; } usleep(50000); } });
                const double t0 = now();
                int launches = 0;
                CK(hipEventRecord(e0, st));
                while (now() - t0 < secs) { for (int i = 0; i < 10; ++i) run(arm); launches += 10; CK(hipStreamSynchronize(st)); }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                stop.store(true); sampler.join();
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / launches;
                const double mhz = ns ? fsum / ns : -1.0;
                // LDS-DMA bytes per clock per CU: operand bytes the launch stages / (CUs x launch time x shader clock)
                const double tile_cols = arm == 0 ? 256.0 : 128.0;
                const double staged = (double)(M / 256) * (N / tile_cols) * (K / 64) * (256 + tile_cols) * 128.0;
                printf("  round %d  %-84s %8.1f us/launch  %7.1f TFLOP/s  board %6.0f W  sclk %5.0f MHz  LDS-DMA %5.1f B/clk/CU  (%d launches)\n", round, arms[arm].name, us,
                       2.0 * M * N * K / (us * 1e-6) / 1e12, ns ? wsum / ns : -1.0, mhz, mhz > 0 ? staged / (ncu * us * 1e-6 * mhz * 1e6) : -1.0, launches);
                fflush(stdout);
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dstamps)); CK(hipFree(dRef)); CK(hipFree(dOut));
    }
    return 0;
}
