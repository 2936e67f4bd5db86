// K10: exact fp32 inner-product search, top-k per query — the arithmetic of
// faiss.IndexFlatIP.search(query, k) in the reference's eval path (bioscanclip/util/util.py:521-528).
//
// Scores use v_mfma_f32_32x32x2_f32: exact fp32 products accumulated as a k-ordered fmaf chain (no bf16 rounding), so
// the integer top-k indices equal those of an fp32 reference wherever its scores are not within an ulp of a tie.
// Two kernels: (1) 64x64-tile score GEMM S[Q,Nk] = q · keys^T (fp32 operands staged through padded LDS, one 32x32
// accumulator tile per wave); (2) one wave per query row: per-lane sorted top-k insertion over a strided sweep, then k
// rounds of wave arg-max (ties -> lower key index, like a stable sort of the exact scores).
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

constexpr int TK_BK = 32;
constexpr int TK_LD = TK_BK + 1;  // +1 float: conflict-free ds_read_b32 column reads

__global__ __launch_bounds__(256) void ip_scores_f32_kernel(const float* __restrict__ q, const float* __restrict__ keys, int Q,
                                                            int Nk, int D, float* __restrict__ S, int ldS) {
    __shared__ float qa[64 * TK_LD];
    __shared__ float kb[64 * TK_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int wq = (wave >> 1) * 32, wk = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // staging: thread t loads rows (t>>3) and (t>>3)+32 of both tiles, 4 consecutive k at 4*(t&7)
    const int srow = threadIdx.x >> 3, scol = (threadIdx.x & 7) * 4;
    for (int kk = 0; kk < D; kk += TK_BK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = srow + 32 * h;
            const int gq = min(q0 + r, Q - 1), gk = min(k0 + r, Nk - 1);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (kk + scol + 3 < D) {
                a = *(const float4*)(q + (size_t)gq * D + kk + scol);
                b = *(const float4*)(keys + (size_t)gk * D + kk + scol);
            } else {
                float ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; ++e)
                    if (kk + scol + e < D) { ta[e] = q[(size_t)gq * D + kk + scol + e]; tb[e] = keys[(size_t)gk * D + kk + scol + e]; }
                a = make_float4(ta[0], ta[1], ta[2], ta[3]);
                b = make_float4(tb[0], tb[1], tb[2], tb[3]);
            }
            float* pa = qa + r * TK_LD + scol;
            float* pb = kb + r * TK_LD + scol;
            pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
            pb[0] = b.x; pb[1] = b.y; pb[2] = b.z; pb[3] = b.w;
        }
        __syncthreads();
        // mfma_f32_32x32x2f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D[i][j] rows on regs
        // here A = keys tile (rows -> D rows), B = query tile (cols): D[key][query]; lane: query = l&31, keys by register
#pragma unroll
        for (int s = 0; s < TK_BK; s += 2) {
            const float av = kb[(wk + (lane & 31)) * TK_LD + s + (lane >> 5)];
            const float bv = qa[(wq + (lane & 31)) * TK_LD + s + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of 32x32: col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (key)
    const int qq = q0 + wq + (lane & 31);
    if (qq < Q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + wk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (key < Nk) S[(size_t)qq * ldS + key] = acc[r];
        }
    }
}

template <int KMAX>
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ S, int ldS, int Q, int Nk, int k,
                                                       long long* __restrict__ out_idx, float* __restrict__ out_sim) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const float* sr = S + (size_t)row * ldS;
    float bv[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { bv[j] = -3.0e38f; bi[j] = 0x7fffffff; }
    for (int c = lane; c < Nk; c += 64) {
        float v = sr[c];
        int id = c;
        // sorted insertion (descending value, ascending index on ties); c increases, so "<=" keeps the earlier index first
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const bool better = (v > bv[j]) || (v == bv[j] && id < bi[j]);
            const float tv = better ? bv[j] : v;
            const int ti = better ? bi[j] : id;
            bv[j] = better ? v : bv[j];
            bi[j] = better ? id : bi[j];
            v = tv;
            id = ti;
        }
    }
    // k rounds: wave arg-max over the lanes' current heads, winner pops its head
    for (int r = 0; r < k; ++r) {
        float v = bv[0];
        int id = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(id, o, 64);
            const bool take = (ov > v) || (ov == v && oi < id);
            v = take ? ov : v;
            id = take ? oi : id;
        }
        if (lane == 0) {
            out_idx[(size_t)row * k + r] = (long long)id;
            out_sim[(size_t)row * k + r] = v;
        }
        if (bi[0] == id) {  // unique winner (indices are distinct): shift this lane's list
#pragma unroll
            for (int j = 0; j + 1 < KMAX; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
            bv[KMAX - 1] = -3.0e38f;
            bi[KMAX - 1] = 0x7fffffff;
        }
    }
}

}  // namespace clibd

using namespace clibd;

extern "C" size_t clibd_topk_ip_workspace_bytes(int Q, int Nk) {
    if (Q <= 0 || Nk <= 0) return 0;
    return (size_t)Q * (size_t)((Nk + 3) / 4 * 4) * sizeof(float);
}

extern "C" int clibd_topk_ip(const float* q, const float* keys, int Q, int Nk, int D, int k, int64_t* out_idx, float* out_sim,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !keys || !out_idx || !out_sim || !workspace) return set_error(CLIBD_EINVAL, "topk_ip: null pointer");
    if (Q <= 0 || Nk <= 0 || D <= 0) return set_error(CLIBD_EINVAL, "topk_ip: non-positive shape");
    if (k < 1 || k > 8 || k > Nk) return set_error(CLIBD_EINVAL, "topk_ip: need 1 <= k <= min(8, Nk)");
    if (D % 4 != 0) return set_error(CLIBD_EINVAL, "topk_ip: D must be a multiple of 4");
    if (!aligned16(q) || !aligned16(keys) || !aligned16(workspace)) return set_error(CLIBD_EINVAL, "topk_ip: alignment");
    if (workspace_bytes < clibd_topk_ip_workspace_bytes(Q, Nk)) return set_error(CLIBD_EINVAL, "topk_ip: workspace too small");
    const int ldS = (Nk + 3) / 4 * 4;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((Nk + 63) / 64, (Q + 63) / 64);
    if (grid.y > 65535) return set_error(CLIBD_EINVAL, "topk_ip: chunk the queries (Q <= 4M per call)");
    hipLaunchKernelGGL(ip_scores_f32_kernel, grid, dim3(256), 0, st, q, keys, Q, Nk, D, (float*)workspace, ldS);
    if (int e = check_launch("ip_scores_f32")) return e;
    hipLaunchKernelGGL(row_topk_kernel<8>, dim3((Q + 3) / 4), dim3(256), 0, st, (const float*)workspace, ldS, Q, Nk, k,
                       (long long*)out_idx, out_sim);
    return check_launch("row_topk");
}
