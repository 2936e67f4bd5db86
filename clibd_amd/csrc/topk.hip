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
