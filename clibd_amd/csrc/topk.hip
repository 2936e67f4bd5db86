// K10: exact fp32 inner-product search, top-k per query — the arithmetic of
// faiss.IndexFlatIP.search(query, k) in the reference's eval path (bioscanclip/util/util.py:521-528).
//
// ONE streaming kernel; the [Q, Nk] score matrix never exists (64 GB at 40 k x 400 k).  A workgroup owns 64 queries and
// sweeps its range of keys 64 at a time: the 64 x 64 score tile is accumulated on v_mfma_f32_32x32x2_f32 (exact fp32
// products, a k-ordered fmaf chain per score — no bf16 rounding, so the integer top-k indices equal those of an fp32
// reference wherever its scores are not within an ulp of a tie), and goes straight from the accumulators into running
// top-8 lists kept in registers.  The MFMA's D layout gives every lane 16 scores of ONE query (query = lane & 31, keys by
// register), so a lane owns a sorted (value, index) list for its query over the keys it sees; after the warm-up almost no
// score beats a list's tail and the insertion is skipped by one wave vote per tile.  A query has four such lists (two
// lanes x two waves over disjoint keys); they are merged through LDS at the end (ties -> lower key index, like a stable
// sort of the exact scores).  With few queries the keys are split over blockIdx.y for occupancy; every split writes its
// per-query top-8 to a small workspace (split x Q x 8 pairs) and a second kernel merges them.
//
// Bound: fp32 MFMA (2 Q Nk D FLOP at 157 TFLOP/s), not HBM: the key bank is read once per 64-query block and is
// L2 / Infinity-Cache resident across blocks.
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

constexpr int TK_BK = 32;
constexpr int TK_LD = TK_BK + 1;  // +1 float: conflict-free ds_read_b32 column reads
constexpr int TK_KMAX = 8;
constexpr float TK_NEG = -3.0e38f;
constexpr int TK_NOIDX = 0x7fffffff;

// (v, id) goes before (bv, bi) in the result order: larger score first, equal scores by ascending key index
__device__ __forceinline__ bool tk_before(float v, int id, float bv, int bi) { return (v > bv) || (v == bv && id < bi); }

__device__ __forceinline__ void tk_insert(float v, int id, float (&bv)[TK_KMAX], int (&bi)[TK_KMAX]) {
#pragma unroll
    for (int j = 0; j < TK_KMAX; ++j) {
        const bool better = tk_before(v, id, bv[j], bi[j]);
        const float tv = better ? bv[j] : v;
        const int ti = better ? bi[j] : id;
        bv[j] = better ? v : bv[j];
        bi[j] = better ? id : bi[j];
        v = tv;
        id = ti;
    }
}

// part_v / part_i: [nsplit][Q][8] per-split lists (nsplit > 1), else the final out_idx / out_sim [Q][k] are written
__global__ __launch_bounds__(256) void topk_ip_stream_kernel(const float* __restrict__ q, const float* __restrict__ keys, int Q, int Nk,
                                                             int D, int tiles_per_split, int nsplit, int k,
                                                             long long* __restrict__ out_idx, float* __restrict__ out_sim,
                                                             float* __restrict__ part_v, int* __restrict__ part_i) {
    __shared__ float qa[2][64 * TK_LD];   // two stages: the next 32-deep chunk is fetched and written while the current one is multiplied
    __shared__ float kb[2][64 * TK_LD];
    __shared__ float cand_v[64][4 * TK_KMAX + 1];
    __shared__ int cand_i[64][4 * TK_KMAX + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 64;
    const int split = blockIdx.y;
    const int tile_beg = split * tiles_per_split;
    const int ntiles = (Nk + 63) / 64;
    const int tile_end = min(tile_beg + tiles_per_split, ntiles);
    const int wq = (wave >> 1) * 32, wk = (wave & 1) * 32;
    float bv[TK_KMAX];
    int bi[TK_KMAX];
#pragma unroll
    for (int j = 0; j < TK_KMAX; ++j) { bv[j] = TK_NEG; bi[j] = TK_NOIDX; }
    // staging: thread t loads rows (t>>3) and (t>>3)+32 of both tiles, 4 consecutive k at 4*(t&7)
    const int srow = threadIdx.x >> 3, scol = (threadIdx.x & 7) * 4;
    const int nchunks = (D + TK_BK - 1) / TK_BK;
    // one (key tile, k chunk) step = fetch -> registers -> LDS stage; the fetch of step n + 1 is issued before the MFMAs of step n
    float4 fa[2], fb[2];
    auto fetch = [&](int tile, int kk) {
        const int k0 = tile * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = srow + 32 * h;
            const int gq = min(q0 + r, Q - 1), gk = min(k0 + r, Nk - 1);
            fa[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            fb[h] = fa[h];
            if (kk + scol + 3 < D) {
                fa[h] = *(const float4*)(q + (size_t)gq * D + kk + scol);
                fb[h] = *(const float4*)(keys + (size_t)gk * D + kk + scol);
            } else {
                float ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; ++e)
                    if (kk + scol + e < D) { ta[e] = q[(size_t)gq * D + kk + scol + e]; tb[e] = keys[(size_t)gk * D + kk + scol + e]; }
                fa[h] = make_float4(ta[0], ta[1], ta[2], ta[3]);
                fb[h] = make_float4(tb[0], tb[1], tb[2], tb[3]);
            }
        }
    };
    auto stash = [&](int stage) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = srow + 32 * h;
            float* pa = qa[stage] + r * TK_LD + scol;
            float* pb = kb[stage] + r * TK_LD + scol;
            pa[0] = fa[h].x; pa[1] = fa[h].y; pa[2] = fa[h].z; pa[3] = fa[h].w;
            pb[0] = fb[h].x; pb[1] = fb[h].y; pb[2] = fb[h].z; pb[3] = fb[h].w;
        }
    };
    int step = 0;
    if (tile_beg < tile_end) {
        fetch(tile_beg, 0);
        stash(0);
    }
    __syncthreads();
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        const int k0 = tile * 64;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int c = 0; c < nchunks; ++c, ++step) {
            const int cur = step & 1;
            // the step after this one: next chunk of this tile, or the first chunk of the next tile
            const bool more = (c + 1 < nchunks) || (tile + 1 < tile_end);
            if (more) fetch(c + 1 < nchunks ? tile : tile + 1, c + 1 < nchunks ? (c + 1) * TK_BK : 0);
            // mfma_f32_32x32x2f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; A = keys tile (D rows),
            // B = query tile (D columns): D[key][query]
#pragma unroll
            for (int s = 0; s < TK_BK; s += 2) {
                const float av = kb[cur][(wk + (lane & 31)) * TK_LD + s + (lane >> 5)];
                const float bvq = qa[cur][(wq + (lane & 31)) * TK_LD + s + (lane >> 5)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvq, acc, 0, 0, 0);
            }
            if (more) stash(cur ^ 1);   // the other stage was last read one step ago, before the barrier below
            __syncthreads();
        }
        // C/D layout of 32x32: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (key)
        const int kbase = k0 + wk + 4 * (lane >> 5);
        bool any_in = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            any_in = any_in || (key < Nk && tk_before(acc[r], key, bv[TK_KMAX - 1], bi[TK_KMAX - 1]));
        }
        if (__any(any_in)) {   // wave vote: after the first tiles a score rarely beats the tail of a list
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key < Nk && tk_before(acc[r], key, bv[TK_KMAX - 1], bi[TK_KMAX - 1])) tk_insert(acc[r], key, bv, bi);
            }
        }
    }
    // ---- merge the four lists of every query: holder h = 2 * (wave & 1) + (lane >> 5)
    {
        const int ql = wq + (lane & 31);
        const int h = 2 * (wave & 1) + (lane >> 5);
#pragma unroll
        for (int j = 0; j < TK_KMAX; ++j) {
            cand_v[ql][h * TK_KMAX + j] = bv[j];
            cand_i[ql][h * TK_KMAX + j] = bi[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ql = threadIdx.x;
        const int gq = q0 + ql;
        if (gq < Q) {
            float pv = 3.0e38f;
            int pi = -1;
            const int rounds = nsplit > 1 ? TK_KMAX : k;
            for (int r = 0; r < rounds; ++r) {
                // the best candidate strictly after the previous pick in the result order (key indices are distinct)
                float cv = TK_NEG;
                int ci = TK_NOIDX;
                for (int c = 0; c < 4 * TK_KMAX; ++c) {
                    const float v = cand_v[ql][c];
                    const int id = cand_i[ql][c];
                    if (id != TK_NOIDX && tk_before(pv, pi, v, id) && tk_before(v, id, cv, ci)) { cv = v; ci = id; }
                }
                pv = cv;
                pi = ci;
                if (nsplit > 1) {
                    part_v[((size_t)split * Q + gq) * TK_KMAX + r] = cv;
                    part_i[((size_t)split * Q + gq) * TK_KMAX + r] = ci;
                } else {
                    out_idx[(size_t)gq * k + r] = (long long)ci;
                    out_sim[(size_t)gq * k + r] = cv;
                }
                if (ci == TK_NOIDX) { pv = TK_NEG; pi = TK_NOIDX; }   // lists exhausted (fewer keys than slots in this split)
            }
        }
    }
}

// one thread per query: k rounds of selection over the nsplit x 8 candidates of the splits
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i, int Q, int nsplit,
                                                         int k, long long* __restrict__ out_idx, float* __restrict__ out_sim) {
    const int gq = blockIdx.x * 256 + threadIdx.x;
    if (gq >= Q) return;
    float pv = 3.0e38f;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        float cv = TK_NEG;
        int ci = TK_NOIDX;
        for (int s = 0; s < nsplit; ++s) {
            const float* v8 = part_v + ((size_t)s * Q + gq) * TK_KMAX;
            const int* i8 = part_i + ((size_t)s * Q + gq) * TK_KMAX;
#pragma unroll
            for (int j = 0; j < TK_KMAX; ++j) {
                const float v = v8[j];
                const int id = i8[j];
                if (id != TK_NOIDX && tk_before(pv, pi, v, id) && tk_before(v, id, cv, ci)) { cv = v; ci = id; }
            }
        }
        pv = cv;
        pi = ci;
        out_idx[(size_t)gq * k + r] = (long long)ci;
        out_sim[(size_t)gq * k + r] = cv;
    }
}

// key splits: enough workgroups for ~2 per CU; a split keeps >= 2 key tiles, at most 256 splits (the merge reads 8 pairs per split)
static int topk_splits(int Q, int Nk) {
    const int qblocks = (Q + 63) / 64;
    const int ntiles = (Nk + 63) / 64;
    int nsplit = (512 + qblocks - 1) / qblocks;
    if (nsplit > ntiles / 2) nsplit = ntiles / 2;
    if (nsplit > 256) nsplit = 256;
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}


// ================================================================================================================================
// Pre-filtered search (round 4): the same top-k, indices and similarities bit-identical to clibd_topk_ip, at the bf16 MFMA rate.
//   1. topk_prepare_keys_kernel: the key bank once as bf16 [Nk, D] + max_n ||key_n|| (the bank is fixed across query batches).
//   2. topk_bf16_stream_kernel: APPROXIMATE scores s~ = bf16(q) . bf16(key) on v_mfma_f32_32x32x16_bf16 (1/16 of the fp32 MFMA's
//      cycles per k), 128 queries x 64 keys per workgroup step, running top-8 lists in registers exactly like the exact kernel
//      (two lists per query and key split).
//   3. topk_rescore_kernel (one wave per query): T = k-th largest approximate score over the query's lists.  With
//        |s~ - s| <= eps = TK_C * ||q|| * max ||key||        (bf16 rounds each operand by <= 2^-9 relative, Cauchy-Schwarz; TK_C =
//        0.0045 > 2^-8 leaves 6e-4 for both accumulations)
//      every true top-k key has s~ >= T - 2 eps  [k keys have s~ >= T, hence s >= T - eps, so the k-th exact score E_k >= T - eps;
//      a true top-k key has s >= E_k, hence s~ >= s - eps >= T - 2 eps].  The wave re-scores every listed key above that line
//      with the exact kernel's arithmetic — a k-ordered fp32 fmaf chain, which is what v_mfma_f32_32x32x2_f32 evaluates — and
//      selects the top k by (score descending, index ascending).  If a LIST is full above the line (its 8th entry >= T - 2 eps) a
//      key above the line may have been pushed out of it: the query is flagged in `overflow` and the caller re-runs it through
//      clibd_topk_ip (exactness never depends on the data; only the speed does).
// Random unit vectors at 1 k x 410 k x 768: ~12 keys above the line per query, ~3 per list.
// wave-wide best (score, index) pair under tk_before, in every lane: the maximum score, then the lowest index among the lanes that
// hold it.  DPP / v_permlane reductions only (common.h: no LDS-crossbar instruction); key indices are < 2^24, exact as floats.
__device__ __forceinline__ void tk_wave_best(float& v, int& id) {
    const float m = wave_max(v);
    const float cand = (v == m && id != TK_NOIDX) ? -(float)id : -3.0e38f;
    const float best = wave_max(cand);
    v = m;
    id = best > -3.0e38f ? (int)(-best) : TK_NOIDX;
}

constexpr float TK_C = 0.0045f;
constexpr int TK_FAST_MAX_KEYS = 1 << 24;   // candidate lists carry key indices as floats: exact below 2^24
constexpr int TK_FAST_MAX_D = 2048;          // 2 D 2^-23 <= 4.9e-4 < the 5.9e-4 of headroom TK_C leaves for fp32 accumulation
constexpr int TB_Q = 128, TB_K = 64, TB_BK = 64;      // queries / keys per workgroup step, k per LDS chunk
constexpr int TB_LD = TB_BK * 2 + 16;                  // LDS row stride in bytes (144: conflict-free ds_read_b128 over 16 rows)

__global__ __launch_bounds__(256) void topk_prepare_keys_kernel(const float* __restrict__ keys, int Nk, int D, unsigned short* __restrict__ keys_bf16,
                                                                 float* __restrict__ max_norm) {
    // one wave per key row: bf16 image + ||key||; the block's maximum goes out as one float atomicMax (on the bit pattern: norms >= 0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float best = 0.f;
    for (int n = blockIdx.x * 4 + wave; n < Nk; n += gridDim.x * 4) {
        float ss = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *(const float4*)(keys + (size_t)n * D + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            *(uint2*)(keys_bf16 + (size_t)n * D + c) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
        }
        ss = wave_sum(ss);
        best = fmaxf(best, sqrtf(ss));
    }
    if (lane == 0) atomicMax((unsigned*)max_norm, __float_as_uint(best * 1.000001f));
}

// queries as bf16 + ||q|| (fp32)
__global__ __launch_bounds__(256) void topk_prepare_queries_kernel(const float* __restrict__ q, int Q, int D, unsigned short* __restrict__ q_bf16,
                                                                   float* __restrict__ q_norm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = blockIdx.x * 4 + wave; n < Q; n += gridDim.x * 4) {
        float ss = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *(const float4*)(q + (size_t)n * D + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            *(uint2*)(q_bf16 + (size_t)n * D + c) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
        }
        ss = wave_sum(ss);
        if (lane == 0) q_norm[n] = sqrtf(ss) * 1.000001f;
    }
}

// part_v / part_i: [nsplit][Q][2 lists][8] approximate (score, key) pairs, each list sorted best first
__global__ __launch_bounds__(256) void topk_bf16_stream_kernel(const unsigned short* __restrict__ q, const unsigned short* __restrict__ keys, int Q,
                                                               int Nk, int D, int tiles_per_split, float* __restrict__ part_v,
                                                               int* __restrict__ part_i) {
    extern __shared__ __attribute__((aligned(16))) char tb_smem[];
    // two stages of [128 query rows | 64 key rows] x 64 k (bf16), row stride TB_LD
    constexpr int STAGE = (TB_Q + TB_K) * TB_LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * TB_Q;
    const int split = blockIdx.y;
    const int ntiles = (Nk + TB_K - 1) / TB_K;
    const int tile_beg = split * tiles_per_split;
    const int tile_end = min(tile_beg + tiles_per_split, ntiles);
    const int wq = wave * 32;                      // this wave's 32 queries; it sees all 64 keys of a tile (two 32 x 32 accumulators)
    float bv[TK_KMAX];
    int bi[TK_KMAX];
#pragma unroll
    for (int j = 0; j < TK_KMAX; ++j) { bv[j] = TK_NEG; bi[j] = TK_NOIDX; }
    // staging: 192 rows x 8 pieces of 16 bytes per chunk = 1536 pieces, 6 per thread: piece p = threadIdx.x + 256 i -> row p >> 3, piece p & 7
    uint4 st[6];
    const int nchunks = D / TB_BK;
    // (macros, not lambdas: captured by reference the six staging registers stayed in scratch memory)
#define TB_FETCH(tile_, c_)                                                                                                     \
    do {                                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                         \
            const int pc = threadIdx.x + 256 * i;                                                                               \
            const int row = pc >> 3, piece = pc & 7;                                                                            \
            const unsigned short* src = row < TB_Q ? q + (size_t)min(q0 + row, Q - 1) * D                                       \
                                                   : keys + (size_t)min((tile_) * TB_K + row - TB_Q, Nk - 1) * D;               \
            st[i] = *(const uint4*)(src + (c_) * TB_BK + piece * 8);                                                            \
        }                                                                                                                       \
    } while (0)
#define TB_STASH(stage_)                                                                                                        \
    do {                                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                         \
            const int pc = threadIdx.x + 256 * i;                                                                               \
            *(uint4*)(tb_smem + (stage_) * STAGE + (pc >> 3) * TB_LD + (pc & 7) * 16) = st[i];                                  \
        }                                                                                                                       \
    } while (0)
    int step = 0;
    if (tile_beg < tile_end) { TB_FETCH(tile_beg, 0); TB_STASH(0); }
    __syncthreads();
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        f32x16 acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][i] = 0.f;
        for (int c = 0; c < nchunks; ++c, ++step) {
            const int cur = step & 1;
            const bool more = (c + 1 < nchunks) || (tile + 1 < tile_end);
            // (unconditional: after the last step it re-fetches the current chunk into the stage nobody reads again — a conditional
            // fetch left the six staging registers behind a phi that hipcc kept in scratch memory)
            TB_FETCH(more ? (c + 1 < nchunks ? tile : tile + 1) : tile, more ? (c + 1 < nchunks ? c + 1 : 0) : c);
            const char* sq = tb_smem + cur * STAGE;
            const char* sk = sq + TB_Q * TB_LD;
            // mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 (l >> 5) + j] and B[k = 8 (l >> 5) + j][j' = l & 31]: A = key rows, B = query rows
#pragma unroll
            for (int s = 0; s < TB_BK / 16; ++s) {
                const bf16x8 bq = *(const bf16x8*)(sq + (wq + (lane & 31)) * TB_LD + s * 32 + (lane >> 5) * 16);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x8 ak = *(const bf16x8*)(sk + (32 * h + (lane & 31)) * TB_LD + s * 32 + (lane >> 5) * 16);
                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, bq, acc[h], 0, 0, 0);
                }
            }
            TB_STASH(cur ^ 1);
            __syncthreads();
        }
        // C/D layout of 32x32: col = lane & 31 (query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (key)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kbase = tile * TB_K + 32 * h + 4 * (lane >> 5);
            bool any_in = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                any_in = any_in || (key < Nk && tk_before(acc[h][r], key, bv[TK_KMAX - 1], bi[TK_KMAX - 1]));
            }
            if (__any(any_in)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + (r & 3) + 8 * (r >> 2);
                    if (key < Nk && tk_before(acc[h][r], key, bv[TK_KMAX - 1], bi[TK_KMAX - 1])) tk_insert(acc[h][r], key, bv, bi);
                }
            }
        }
    }
    const int gq = q0 + wq + (lane & 31);
    if (gq < Q) {
        const size_t base = (((size_t)split * Q + gq) * 2 + (lane >> 5)) * TK_KMAX;
#pragma unroll
        for (int j = 0; j < TK_KMAX; ++j) { part_v[base + j] = bv[j]; part_i[base + j] = bi[j]; }
    }
}

#undef TB_FETCH
#undef TB_STASH

// one wave per query
__global__ __launch_bounds__(256) void topk_rescore_kernel(const float* __restrict__ q, const float* __restrict__ keys, const float* __restrict__ q_norm,
                                                           const float* __restrict__ max_norm, const float* __restrict__ part_v,
                                                           const int* __restrict__ part_i, int Q, int D, int nsplit, int k,
                                                           long long* __restrict__ out_idx, float* __restrict__ out_sim, int* __restrict__ overflow) {
    __shared__ float cv[4][64];
    __shared__ int ci[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gq = blockIdx.x * 4 + wave;
    if (gq >= Q) return;
    const int nlists = 2 * nsplit;
    // ---- T: the k-th largest approximate score.  Lane-strided scan of all entries; k rounds of a wave-wide maximum (k <= 8)
    const int nent = nlists * TK_KMAX;
    auto ent_v = [&](int e) { return part_v[((size_t)(e / (2 * TK_KMAX)) * Q + gq) * 2 * TK_KMAX + e % (2 * TK_KMAX)]; };
    auto ent_i = [&](int e) { return part_i[((size_t)(e / (2 * TK_KMAX)) * Q + gq) * 2 * TK_KMAX + e % (2 * TK_KMAX)]; };
    float pv = 3.0e38f;
    int pi = -1;
    float T = TK_NEG;
    for (int r = 0; r < k; ++r) {
        float bvv = TK_NEG;
        int bii = TK_NOIDX;
        for (int e = lane; e < nent; e += 64) {
            const float v = ent_v(e);
            const int id = ent_i(e);
            if (id != TK_NOIDX && tk_before(pv, pi, v, id) && tk_before(v, id, bvv, bii)) { bvv = v; bii = id; }
        }
        tk_wave_best(bvv, bii);
        pv = bvv;
        pi = bii;
        T = bvv;
        if (bii == TK_NOIDX) break;
    }
    const float line = T - 2.0f * TK_C * q_norm[gq] * max_norm[0];
    // ---- candidates above the line; a list that is full above the line may have lost one
    int ovf = 0;
    int ncand = 0;
    for (int e0 = 0; e0 < nent; e0 += 64) {
        const int e = e0 + lane;
        float v = TK_NEG;
        int id = TK_NOIDX;
        if (e < nent) { v = ent_v(e); id = ent_i(e); }
        const bool in = id != TK_NOIDX && v >= line;
        if (in && (e % TK_KMAX) == TK_KMAX - 1) ovf = 1;
        const unsigned long long m = __ballot(in);
        const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < 64) { cv[wave][pos] = v; ci[wave][pos] = id; }
        ncand += __popcll(m);
    }
    ovf = __any(ovf) ? 1 : 0;
    if (ncand > 64) { ovf = 1; ncand = 64; }
    // ---- exact scores of the candidates: lane c re-scores candidate c with the exact kernel's k-ordered fmaf chain
    float ev = TK_NEG;
    int eid = TK_NOIDX;
    if (lane < ncand) {
        eid = ci[wave][lane];
        const float* kr = keys + (size_t)eid * D;
        const float* qr = q + (size_t)gq * D;
        float a = 0.f;
        for (int c = 0; c < D; c += 4) {
            const float4 kk = *(const float4*)(kr + c);
            const float4 qq = *(const float4*)(qr + c);
            a = __builtin_fmaf(kk.x, qq.x, a); a = __builtin_fmaf(kk.y, qq.y, a); a = __builtin_fmaf(kk.z, qq.z, a); a = __builtin_fmaf(kk.w, qq.w, a);
        }
        ev = a;
    }
    // ---- top k of the exact scores by (score descending, index ascending)
    pv = 3.0e38f;
    pi = -1;
    for (int r = 0; r < k; ++r) {
        float bvv = TK_NEG;
        int bii = TK_NOIDX;
        if (eid != TK_NOIDX && tk_before(pv, pi, ev, eid)) { bvv = ev; bii = eid; }
        tk_wave_best(bvv, bii);
        pv = bvv;
        pi = bii;
        if (lane == 0) { out_idx[(size_t)gq * k + r] = (long long)bii; out_sim[(size_t)gq * k + r] = bvv; }
        if (bii == TK_NOIDX) { if (lane == 0) ovf = 1; }
    }
    if (lane == 0) overflow[gq] = ovf;
}

static int topk_fast_splits(int Q, int Nk) {
    const int qblocks = (Q + TB_Q - 1) / TB_Q;
    const int ntiles = (Nk + TB_K - 1) / TB_K;
    int nsplit = (512 + qblocks - 1) / qblocks;
    if (nsplit > ntiles / 4) nsplit = ntiles / 4;
    if (nsplit > 128) nsplit = 128;
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}

}  // namespace clibd

using namespace clibd;

extern "C" size_t clibd_topk_ip_workspace_bytes(int Q, int Nk) {
    if (Q <= 0 || Nk <= 0) return 0;
    const int nsplit = topk_splits(Q, Nk);
    return nsplit > 1 ? (size_t)nsplit * (size_t)Q * TK_KMAX * (sizeof(float) + sizeof(int)) : 16;
}

extern "C" int clibd_topk_ip(const float* q, const float* keys, int Q, int Nk, int D, int k, int64_t* out_idx, float* out_sim,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !keys || !out_idx || !out_sim || !workspace) return set_error(CLIBD_EINVAL, "topk_ip: null pointer");
    if (Q <= 0 || Nk <= 0 || D <= 0) return set_error(CLIBD_EINVAL, "topk_ip: non-positive shape");
    if (k < 1 || k > TK_KMAX || k > Nk) return set_error(CLIBD_EINVAL, "topk_ip: need 1 <= k <= min(8, Nk)");
    if (D % 4 != 0) return set_error(CLIBD_EINVAL, "topk_ip: D must be a multiple of 4");
    if (!aligned16(q) || !aligned16(keys) || !aligned16(workspace)) return set_error(CLIBD_EINVAL, "topk_ip: alignment");
    if (workspace_bytes < clibd_topk_ip_workspace_bytes(Q, Nk)) return set_error(CLIBD_EINVAL, "topk_ip: workspace too small");
    const int nsplit = topk_splits(Q, Nk);
    const int ntiles = (Nk + 63) / 64;
    const int tiles_per_split = (ntiles + nsplit - 1) / nsplit;
    hipStream_t st = (hipStream_t)stream;
    float* part_v = (float*)workspace;
    int* part_i = (int*)(part_v + (size_t)nsplit * Q * TK_KMAX);
    dim3 grid((unsigned)((Q + 63) / 64), (unsigned)nsplit);
    hipLaunchKernelGGL(topk_ip_stream_kernel, grid, dim3(256), 0, st, q, keys, Q, Nk, D, tiles_per_split, nsplit, k, (long long*)out_idx,
                       out_sim, part_v, part_i);
    if (int e = check_launch("topk_ip_stream")) return e;
    if (nsplit > 1) {
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, (const float*)part_v, (const int*)part_i, Q,
                           nsplit, k, (long long*)out_idx, out_sim);
        return check_launch("topk_merge");
    }
    return CLIBD_OK;
}

// ---- pre-filtered search: C ABI
extern "C" int clibd_topk_prepare_keys(const float* keys, int Nk, int D, void* keys_bf16, float* max_norm, void* stream) {
    if (!keys || !keys_bf16 || !max_norm) return set_error(CLIBD_EINVAL, "topk_prepare_keys: null pointer");
    if (Nk <= 0 || D <= 0 || D % 64 != 0) return set_error(CLIBD_EINVAL, "topk_prepare_keys: D must be a positive multiple of 64");
    if (Nk >= TK_FAST_MAX_KEYS || D > TK_FAST_MAX_D) return set_error(CLIBD_EINVAL, "topk_prepare_keys: the pre-filtered search takes Nk < 2^24 keys and D <= 2048 (use clibd_topk_ip)");
    if (!aligned16(keys) || !aligned16(keys_bf16)) return set_error(CLIBD_EINVAL, "topk_prepare_keys: alignment");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(max_norm, 0, sizeof(float), st) != hipSuccess) return set_error(CLIBD_ELAUNCH, "topk_prepare_keys: memset");
    hipLaunchKernelGGL(topk_prepare_keys_kernel, dim3((unsigned)min((Nk + 3) / 4, 4096)), dim3(256), 0, st, keys, Nk, D, (unsigned short*)keys_bf16, max_norm);
    return check_launch("topk_prepare_keys");
}

extern "C" size_t clibd_topk_ip_fast_workspace_bytes(int Q, int Nk, int D) {
    if (Q <= 0 || Nk <= 0 || D <= 0) return 0;
    const int nsplit = topk_fast_splits(Q, Nk);
    size_t b = (size_t)Q * D * 2;                       // bf16 queries
    b = (b + 255) / 256 * 256;
    b += (size_t)Q * sizeof(float);                    // ||q||
    b = (b + 255) / 256 * 256;
    b += (size_t)nsplit * Q * 2 * TK_KMAX * (sizeof(float) + sizeof(int));
    return b + 256;
}

extern "C" int clibd_topk_ip_fast(const float* q, const float* keys, const void* keys_bf16, const float* max_norm, int Q, int Nk, int D, int k,
                                  int64_t* out_idx, float* out_sim, int32_t* overflow, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !keys || !keys_bf16 || !max_norm || !out_idx || !out_sim || !overflow || !workspace) return set_error(CLIBD_EINVAL, "topk_ip_fast: null pointer");
    if (Q <= 0 || Nk <= 0 || D <= 0 || D % 64 != 0) return set_error(CLIBD_EINVAL, "topk_ip_fast: D must be a positive multiple of 64");
    if (k < 1 || k > TK_KMAX || k > Nk) return set_error(CLIBD_EINVAL, "topk_ip_fast: need 1 <= k <= min(8, Nk)");
    // The bit-identity with clibd_topk_ip rests on two bounds: the candidate lists carry key indices as floats (exact
    // below 2^24), and the band TK_C leaves 5.9e-4 of headroom for the two fp32 accumulations, whose worst case grows as 2 D 2^-23.
    if (Nk >= TK_FAST_MAX_KEYS || D > TK_FAST_MAX_D) return set_error(CLIBD_EINVAL, "topk_ip_fast: needs Nk < 2^24 and D <= 2048 (use clibd_topk_ip)");
    if (!aligned16(q) || !aligned16(keys) || !aligned16(keys_bf16) || !aligned16(workspace)) return set_error(CLIBD_EINVAL, "topk_ip_fast: alignment");
    if (workspace_bytes < clibd_topk_ip_fast_workspace_bytes(Q, Nk, D)) return set_error(CLIBD_EINVAL, "topk_ip_fast: workspace too small");
    const int nsplit = topk_fast_splits(Q, Nk);
    const int ntiles = (Nk + TB_K - 1) / TB_K;
    const int tiles_per_split = (ntiles + nsplit - 1) / nsplit;
    char* w = (char*)workspace;
    unsigned short* q16 = (unsigned short*)w;
    size_t off = ((size_t)Q * D * 2 + 255) / 256 * 256;
    float* qn = (float*)(w + off);
    off = (off + (size_t)Q * sizeof(float) + 255) / 256 * 256;
    float* part_v = (float*)(w + off);
    int* part_i = (int*)(part_v + (size_t)nsplit * Q * 2 * TK_KMAX);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(topk_prepare_queries_kernel, dim3((unsigned)min((Q + 3) / 4, 4096)), dim3(256), 0, st, q, Q, D, q16, qn);
    if (int e = check_launch("topk_prepare_queries")) return e;
    constexpr int LDS = 2 * (TB_Q + TB_K) * TB_LD;
    hipFuncSetAttribute((const void*)topk_bf16_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(topk_bf16_stream_kernel, dim3((unsigned)((Q + TB_Q - 1) / TB_Q), (unsigned)nsplit), dim3(256), LDS, st, (const unsigned short*)q16,
                       (const unsigned short*)keys_bf16, Q, Nk, D, tiles_per_split, part_v, part_i);
    if (int e = check_launch("topk_bf16_stream")) return e;
    hipLaunchKernelGGL(topk_rescore_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, q, keys, (const float*)qn, max_norm, (const float*)part_v,
                       (const int*)part_i, Q, D, nsplit, k, (long long*)out_idx, out_sim, (int*)overflow);
    return check_launch("topk_rescore");
}
