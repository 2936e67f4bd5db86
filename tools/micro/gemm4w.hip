// Experiment: the 256x256x64 bf16 GEMM tile on FOUR waves x 512 registers, measured against the product's
// eight-wave kernel (libclibd_hip.so, clibd_gemm_bf16_nt, epilogue kind 1: one bf16 output, no bias / adapters) in ONE process,
// interleaved, on the step's two extreme shapes.
//
// Why four waves: one wave per SIMD owns the whole 512-entry register file, a wave's output block is 128 x 128 (256 accumulator
// registers, hipcc puts them in AGPRs) and every operand fragment read from LDS feeds 4 MFMAs instead of 2.  What it loses: the
// partner wave that issues LDS / memory instructions while this wave issues MFMAs (the eight-wave kernel's two wave groups).  So
// the single instruction stream must carry its loads BETWEEN its MFMAs:
//   * fragments are read one phase ahead into a second register set (8 ds_read_b128 per 32 MFMAs, evenly: Q_hn1, P_hm1, then the
//     next K-tile's Q_hn0 and P_hm0);
//   * global -> LDS staging, two forms:  STAGE=0  LDS-DMA (global_load_lds_dwordx4; ~60 issue cycles per 1-KiB piece among bare
//     MFMAs, MI355X_MICROARCH.md), STAGE=1  register staging (global_load_dwordx4 now, ds_write_b128 three phases later: 64 more
//     live registers).
// Register count with the fragments double-buffered: 256 (accumulators) + 160 (fragments) [+ 64 staging] + addressing = 440-500 of
// 512: there is NO room for the 128-register bf16 stash that would let a finished tile's stores drain under the next tile's MFMAs
// (DESIGN.md §3.1 had priced that design at frac <= 0.47); this program measures what the design can be: the four-wave main loop
// with a plain epilogue.  If that main loop does not beat the eight-wave one there is nothing for a hidden epilogue to add to.
//
// Geometry (same LDS image as gemm256.hip: 128-B rows, 16-B chunk index XOR (row & 7), four 16-KiB half-tiles per K-tile, two
// stages): wave (wm, wn) = (w >> 1, w & 1) owns output columns n0 + 128 wm + [0, 128) (P: W rows, 8 tiles, tile t holds rows
// 8 c + t so that a lane owns 8 contiguous output columns) and rows m0 + 128 wn + [0, 128) (Q: A rows, 8 tiles).  Half-tiles
// P_hm0 | Q_hn0 | Q_hn1 | P_hm1 = tiles 0-3 / 4-7 of every wave.  A K-tile is four phases of 32 MFMAs: quadrants (hm0,hn0)
// (hm0,hn1) (hm1,hn1) (hm1,hn0).  Half-tile stream L_i (i = 4 u + j); phase p: wait until L_{<= p+3} has landed, barrier, issue
// L_{p+7} (its slot, L_{p-1}'s, was last read in phase p - 2), read next phase's fragments, 32 MFMAs.
//
//   hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip -ldl && ./gemm4w <path to libclibd_hip.so> [seconds per arm]
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

constexpr int HALF = 16384, STAGEB = 65536, LDS_BYTES = 131072;

struct P4 {
    const char* A; const char* W; unsigned short* out;
    int M, N, K, lda2, ldw2, ldo;      // lda2 / ldw2: row strides in BYTES; ldo in elements
    int tiles_m, tiles_n, ntiles;
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, b2));
}

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// DEFER = false: plain epilogue after the K loop (all 32 row stores of a wave back to back, no MFMA under them).
// DEFER = true : NO epilogue.  Quadrants run in the order (hm0,hn0) (hm1,hn0) (hm1,hn1) (hm0,hn1), so the rows of hn0 are final after
//   phase 1 of the last K-tile and are stored — straight from the accumulators: 8 v_accvgpr_read + 4 v_cvt_pk + one 16-byte store per
//   row, no stash — between the MFMAs of that K-tile's phases 2 and 3; the rows of hn1 are final after phase 3 and go out under the
//   NEXT tile's phases 0 and 1, which only write hn0 accumulators (its first K-step takes C = 0 instead of zeroed accumulators).
//   Eight stores per phase, one after every fourth MFMA.  vmcnt retires in order and counts stores, so the LDS-DMA waits of the
//   phases around a tile boundary carry the stores that are younger than the awaited half-tile (table at WAITS).
template <bool DEFER>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(P4 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int prow = lane >> 3;
    const unsigned chunk16 = (unsigned)((lane & 7) ^ prow) * 16u;
    const int nk = p.K / 64;                       // even, >= 4 (host-checked)

    // staging: piece i (0..3) of a half-tile = image rows 32 w + 8 i + prow.  Half-tile stream order j = 0 P_hm0, 1 Q_hn0, 2 P_hm1, 3 Q_hn1
    unsigned offP[4], offQ[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = 2 * (w & 1) + (i >> 1), c = 8 * (i & 1) + prow;
        offP[i] = (unsigned)(128 * (w >> 1) + 8 * c + t) * (unsigned)p.ldw2 + chunk16;       // + 4 rows for hm1 (scalar)
        offQ[i] = (unsigned)(128 * (w >> 1) + 16 * t + c) * (unsigned)p.lda2 + chunk16;      // + 64 rows for hn1 (scalar)
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    const unsigned stage_dst = (unsigned)w * 4096u;                 // this wave's 4 KiB of every half-tile
    const int frow = lane & 15, fch = lane >> 4;
    const int aP0 = tile_off(64 * wm + frow, fch), aP1 = tile_off(64 * wm + frow, 4 + fch);
    const int aQ0 = tile_off(64 * wn + frow, fch), aQ1 = tile_off(64 * wn + frow, 4 + fch);
    // output: lane (c = frow, g = fch) owns rows m0 + 128 wn + 64 hn + 16 qt + 4 g + r, columns n0 + 128 wm + 8 c + (4 hm + pt)
    const unsigned voff_out = (unsigned)((4 * fch) * p.ldo + 128 * wm + 8 * frow) * 2u;

    auto tile_xy = [&](int id, int& m0_, int& n0_) {   // XCD-aware: blocks b and b + 8 share an L2; give an XCD a contiguous range
        const int per = (p.ntiles + 7) >> 3;
        const int logical = (id & 7) * per + (id >> 3);   // a bijection: ntiles % 8 == 0 (host-checked)
        m0_ = (logical / p.tiles_n) * 256;
        n0_ = (logical % p.tiles_n) * 256;
    };

    int tile = blockIdx.x;
    int m0, n0, pm0 = 0, pn0 = 0;
    tile_xy(tile, m0, n0);
    int im0 = m0, in0 = n0;      // origin of the tile whose half-tiles are being issued
    int u_is = 0, tile_is = tile;
    const char* sbase_cur = nullptr;
    unsigned dst_cur = 0;
#define ISSUE_BEGIN(J)                                                                                            \
    do {                                                                                                          \
        constexpr bool isP_ = ((J) == 0 || (J) == 2);                                                             \
        sbase_cur = isP_ ? p.W + (size_t)(in0 + ((J) == 2 ? 4 : 0)) * p.ldw2 + (size_t)u_is * 128                 \
                         : p.A + (size_t)(im0 + ((J) == 3 ? 64 : 0)) * p.lda2 + (size_t)u_is * 128;               \
        dst_cur = lds0 + (unsigned)((u_is & 1) * STAGEB + (J) * HALF) + stage_dst;                                \
    } while (0)
#define ISSUE_PIECE(J, I)                                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                 \
                 :: "v"(((J) == 0 || (J) == 2) ? offP[I] : offQ[I]), "s"(sbase_cur), "s"(dst_cur + 1024u * (I)) : "memory", "m0")
#define ISSUE_END(J)                                                                                              \
    do {                                                                                                          \
        if ((J) == 3) {                                                                                           \
            if (++u_is == nk) {   /* the stream runs on into this workgroup's next tile (at the very end it re-fetches the last one: */ \
                u_is = 0;         /* the slots it fills are dead, and every wait count stays the same) */          \
                const int nxt_ = tile_is + (int)gridDim.x;                                                        \
                if (nxt_ < p.ntiles) tile_is = nxt_;                                                              \
                tile_xy(tile_is, im0, in0);                                                                       \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define ISSUE_HALF(J) do { ISSUE_BEGIN(J); ISSUE_PIECE(J, 0); ISSUE_PIECE(J, 1); ISSUE_PIECE(J, 2); ISSUE_PIECE(J, 3); ISSUE_END(J); } while (0)

    f32x4 acc[2][2][4][4];       // [hm][hn][P tile][Q tile]: 256 AGPRs
    bf16x8 aF0a[4][2], aF0b[4][2], aF1[4][2], w0[4][2], w1[4][2];   // [tile][kk]: P_hm0 (two sets), P_hm1, Q_hn0, Q_hn1

#define LOADF(dst, half_slot, a0, a1)                                                                 \
    do {                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                               \
            dst[t][0] = *(const bf16x8*)(smem + (half_slot) + (a0) + 2048 * t);                       \
            dst[t][1] = *(const bf16x8*)(smem + (half_slot) + (a1) + 2048 * t);                       \
        }                                                                                             \
    } while (0)
#define MFMA_ASM(ACC, QA, PB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(QA), "v"(PB))
#define MFMA_ASM0(ACC, QA, PB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(QA), "v"(PB))
    // one output row of this lane (16 bytes), straight from the accumulators
#define STORE_ROW(HN, QT, R, TM0, TN0)                                                                \
    do {                                                                                              \
        const char* rb_ = (const char*)p.out + ((size_t)((TM0) + 128 * wn + 64 * (HN) + 16 * (QT) + (R)) * p.ldo + (TN0)) * 2; \
        u32x4 d_;                                                                                     \
        d_[0] = pack2bf(acc[0][HN][0][QT][R], acc[0][HN][1][QT][R]);                                  \
        d_[1] = pack2bf(acc[0][HN][2][QT][R], acc[0][HN][3][QT][R]);                                  \
        d_[2] = pack2bf(acc[1][HN][0][QT][R], acc[1][HN][1][QT][R]);                                  \
        d_[3] = pack2bf(acc[1][HN][2][QT][R], acc[1][HN][3][QT][R]);                                  \
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff_out), "v"(d_), "s"(rb_) : "memory"); \
    } while (0)
    // One quadrant = 32 MFMAs in program order; after every fourth: an LDS-DMA piece of half-tile J (groups 0, 2, 4, 6) and, where the
    // phase carries stores (SP: 0 none, 1 = rows (SHN, SQB + (g >> 2), g & 3) of the tile at (SM0, SN0), under the run-time condition SC)
#define MMA(hm, hn, PF, QF, J, ZC, SP, SHN, SQB, SM0, SN0, SC)                                        \
    do {                                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                              \
            _Pragma("unroll") for (int pt = 0; pt < 4; ++pt) {                                        \
                _Pragma("unroll") for (int qt = 0; qt < 4; ++qt) {                                    \
                    if ((ZC) && kk == 0) MFMA_ASM0(acc[hm][hn][pt][qt], QF[qt][kk], PF[pt][kk]);      \
                    else MFMA_ASM(acc[hm][hn][pt][qt], QF[qt][kk], PF[pt][kk]);                       \
                }                                                                                     \
                if ((pt & 1) == 0) ISSUE_PIECE(J, 2 * kk + (pt >> 1));                                \
                if (SP) {                                                                             \
                    if (SC) STORE_ROW(SHN, (SQB) + ((4 * kk + pt) >> 2), (4 * kk + pt) & 3, SM0, SN0); \
                    __builtin_amdgcn_sched_barrier(0);                                                \
                }                                                                                     \
            }                                                                                         \
    } while (0)
#define WAITSEL(COND, NA, NB) do { if (COND) asm volatile("s_waitcnt vmcnt(" #NA ")" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #NB ")" ::: "memory"); } while (0)
    // phase head (phase p = 4 u + Q): wait for L_{<= p+3} — in flight behind it: L_{p+4..p+6} = 12 LDS-DMA plus the stores issued in
    // the last three phases —, barrier, start L_{p+7} (its pieces go out between this phase's MFMAs)
#define HEAD(Q, COND, NA, NB) do { WAITSEL(COND, NA, NB); BARRIER(); ISSUE_BEGIN(((Q) + 3) & 3); } while (0)
#define TAIL(Q) ISSUE_END(((Q) + 3) & 3)

    // ---- prologue: L_0 .. L_6; first fragments: P_hm0, Q_hn0 of K-tile 0 (stage 0)
    ISSUE_HALF(0); ISSUE_HALF(1); ISSUE_HALF(2); ISSUE_HALF(3); ISSUE_HALF(0); ISSUE_HALF(1); ISSUE_HALF(2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // L_0 .. L_3 landed
    BARRIER();
    LOADF(aF0a, 0 * HALF, aP0, aP1);
    LOADF(w0, 1 * HALF, aQ0, aQ1);

    bool has_prev = false;
    // WAITS.  DEFER: stores sit in phases L2, L3 (last K-tile) and N0, N1 (first K-tile of the next tile), 8 each:
    //   L3: 12 + 8 = 20 | N0: 12 + 16 = 28 | N1: 36 | N2: 36 | N3: 28 | N4: 20 | else 12   (first tile of a workgroup: no N-phase stores)
    // plain epilogue (32 stores between L3 and N0): N0 .. N3: 12 + 32 = 44, else 12
    // K-tile of stage S with P_hm0 in ACUR, the next K-tile's P_hm0 going to ANEXT; ZC: first K-tile of a tile; WN0..WN3 / SPx: see callers
#define KTILE(S, ACUR, ANEXT, ZC, W0A, W0B, W1A, W1B, W2A, W2B, W3A, W3B, SP0, SP1, SP2, SP3)                                    \
    do {                                                                                                                         \
        HEAD(0, has_prev, W0A, W0B); LOADF(aF1, (S) * STAGEB + 2 * HALF, aP0, aP1);                                              \
        MMA(0, 0, ACUR, w0, 3, ZC, SP0, 1, 0, pm0, pn0, has_prev); TAIL(0);                                                      \
        HEAD(1, has_prev, W1A, W1B); LOADF(w1, (S) * STAGEB + 3 * HALF, aQ0, aQ1);                                               \
        MMA(1, 0, aF1, w0, 0, ZC, SP1, 1, 2, pm0, pn0, has_prev); TAIL(1);                                                       \
        HEAD(2, has_prev, W2A, W2B); LOADF(w0, ((S) ^ 1) * STAGEB + 1 * HALF, aQ0, aQ1);                                         \
        MMA(1, 1, aF1, w1, 1, ZC, SP2, 0, 0, m0, n0, true); TAIL(2);                                                             \
        HEAD(3, has_prev, W3A, W3B); LOADF(ANEXT, ((S) ^ 1) * STAGEB + 0 * HALF, aP0, aP1);                                      \
        MMA(0, 1, ACUR, w1, 2, ZC, SP3, 0, 2, m0, n0, true); TAIL(3);                                                            \
    } while (0)
    while (true) {
        // ---- first K-tile pair of the tile
        if constexpr (DEFER) {
            KTILE(0, aF0a, aF0b, true, 28, 12, 36, 12, 36, 12, 28, 12, 1, 1, 0, 0);
            KTILE(1, aF0b, aF0a, false, 20, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
        } else {
            KTILE(0, aF0a, aF0b, true, 44, 12, 44, 12, 44, 12, 44, 12, 0, 0, 0, 0);
            KTILE(1, aF0b, aF0a, false, 12, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
        }
        // ---- steady K-tile pairs
#pragma unroll 1
        for (int u = 2; u < nk - 2; u += 2) {
            KTILE(0, aF0a, aF0b, false, 12, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
            KTILE(1, aF0b, aF0a, false, 12, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
        }
        // ---- last K-tile pair: DEFER stores the hn0 rows of this tile under phases 2 and 3 of the last K-tile
        KTILE(0, aF0a, aF0b, false, 12, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
        if constexpr (DEFER) {
            KTILE(1, aF0b, aF0a, false, 12, 12, 12, 12, 12, 12, 20, 20, 0, 0, 1, 1);
        } else {
            KTILE(1, aF0b, aF0a, false, 12, 12, 12, 12, 12, 12, 12, 12, 0, 0, 0, 0);
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { STORE_ROW(hn, qt, r, m0, n0); __builtin_amdgcn_sched_barrier(0); }
        }
        const int nxt = tile + (int)gridDim.x;
        if (nxt >= p.ntiles) break;
        pm0 = m0; pn0 = n0;
        tile = nxt;
        tile_xy(tile, m0, n0);
        has_prev = true;
    }
    if constexpr (DEFER) {   // the hn1 rows of this workgroup's last tile have no next tile to hide under
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { STORE_ROW(1, qt, r, m0, n0); __builtin_amdgcn_sched_barrier(0); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the stream ran 7 half-tiles ahead: drain before the LDS is released
#undef KTILE
#undef HEAD
#undef TAIL
#undef WAITSEL
#undef MMA
#undef STORE_ROW
#undef MFMA_ASM
#undef MFMA_ASM0
#undef LOADF
#undef ISSUE_HALF
#undef ISSUE_BEGIN
#undef ISSUE_PIECE
#undef ISSUE_END
}

// ------------------------------------------------------------------------------------------------------------------ host
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec + 1e-6 * t.tv_usec; }

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Board {   // sysfs hwmon of THIS process's device (matched by PCI bus id): power1_input / power1_average (uW), freq1_input (Hz)
    std::string dir, pn;
    Board() {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) return;      // "0000:dc:00.0": this process's device
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

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <libclibd_hip.so> [seconds per arm]\n", argv[0]); return 2; }
    const double secs = argc > 2 ? atof(argv[2]) : 2.0;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    gemm_fn ref = (gemm_fn)dlsym(lib, "clibd_gemm_bf16_nt");
    if (!ref) { fprintf(stderr, "clibd_gemm_bf16_nt not found\n"); return 2; }
    Board board;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int ncu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int M = 403456;
    const int shapes[2][2] = {{768, 3072}, {3072, 768}};   // (N, K)
    for (int si = 0; si < 2; ++si) {
        const int N = shapes[si][0], K = shapes[si][1];
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        unsigned s = 12345u + si;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) % 2001 - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd();
        unsigned short *dA, *dW, *dO[3];
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2));
        for (auto& o : dO) { CK(hipMalloc(&o, (size_t)M * N * 2)); CK(hipMemset(o, 0xff, (size_t)M * N * 2)); }
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        if (((M / 256) * (N / 256)) % 8 || M % 256 || N % 256 || (K / 64) % 2 || K < 256) { fprintf(stderr, "shape outside the experiment\n"); return 2; }
        P4 p{(const char*)dA, (const char*)dW, nullptr, M, N, K, K * 2, K * 2, N, M / 256, N / 256, (M / 256) * (N / 256)};
        clibd_gemm_epilogue ep;
        memset(&ep, 0, sizeof(ep));
        ep.split_k = 1; ep.ld_out_bf16 = N;
        auto run = [&](int arm) {
            if (arm == 0) { ep.out_bf16 = dO[0]; if (ref(dA, K, dW, K, M, N, K, &ep, st) != 0) { fprintf(stderr, "reference gemm failed\n"); exit(1); } }
            else if (arm == 1) { p.out = dO[1]; hipLaunchKernelGGL(gemm4w_kernel<false>, dim3(ncu), dim3(256), LDS_BYTES, st, p); }
            else { p.out = dO[2]; hipLaunchKernelGGL(gemm4w_kernel<true>, dim3(ncu), dim3(256), LDS_BYTES, st, p); }
        };
        // ---- correctness: each arm against a host fp64 dot product on sampled entries, and arm against arm on everything
        for (int arm = 0; arm < 3; ++arm) run(arm);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        std::vector<unsigned short> hO[3];
        for (int a = 0; a < 3; ++a) { hO[a].resize((size_t)M * N); CK(hipMemcpy(hO[a].data(), dO[a], hO[a].size() * 2, hipMemcpyDeviceToHost)); }
        double worst[3] = {0, 0, 0};
        for (int t = 0; t < 4000; ++t) {
            s = s * 1664525u + 1013904223u; const size_t m = (t < 8) ? (size_t)(t < 4 ? t * 85 : M - 1 - (t - 4) * 77) : (s >> 4) % M;
            s = s * 1664525u + 1013904223u; const size_t n = (s >> 4) % N;
            double d = 0;
            for (int k = 0; k < K; ++k) d += (double)bf2f(hA[m * K + k]) * (double)bf2f(hW[n * K + k]);
            for (int a = 0; a < 3; ++a) worst[a] = std::max(worst[a], std::fabs((double)bf2f(hO[a][m * N + n]) - d) / (std::fabs(d) + 1.0));
        }
        size_t diff1 = 0, diff2 = 0;
        for (size_t i = 0; i < hO[0].size(); ++i) { diff1 += hO[1][i] != hO[0][i]; diff2 += hO[2][i] != hO[0][i]; }
        printf("shape M=%d N=%d K=%d: max rel error vs fp64 on 4000 entries: 8-wave %.2e, 4-wave plain epilogue %.2e, 4-wave stores under MFMAs %.2e; "
               "elements differing from the 8-wave kernel: %zu / %zu of %zu\n", M, N, K, worst[0], worst[1], worst[2], diff1, diff2, hO[0].size());
        fflush(stdout);
        // ---- timing: interleaved arms, `secs` of back-to-back launches each, three rounds; board power / clock sampled at 20 Hz
        const char* names[3] = {"8-wave product kernel (kind 1)", "4-wave x 512 regs, plain epilogue", "4-wave x 512 regs, stores under MFMAs"};
        for (int round = 0; round < 3; ++round)
            for (int arm = 0; arm < 3; ++arm) {
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
                printf("  round %d  %-36s %8.1f us/launch  %7.1f TFLOP/s  board %6.0f W  sclk %5.0f MHz  (%d launches)\n", round, names[arm], us,
                       2.0 * M * N * K / (us * 1e-6) / 1e12, ns ? wsum / ns : -1.0, ns ? fsum / ns : -1.0, launches);
                fflush(stdout);
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
        CK(hipFree(dA)); CK(hipFree(dW));
        for (auto& o : dO) CK(hipFree(o));
    }
    return 0;
}
