// bf16 MFMA GEMM for gfx950:  out = epilogue(A[M,K] · W[N,K]^T), fp32 accumulate.
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Both operands are K-contiguous ("NT"), staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) into a 2-deep ring; the LDS image is lane-linear, the
// XOR swizzle (16-B chunk ^= row&7 inside 128-B rows) is applied on the per-lane SOURCE address and
// again on the ds_read_b128 address, which makes every fragment read bank-conflict free.
//
// MFMA orientation: the W tile is the MFMA "A" operand (rows = n) and the activation tile the "B"
// operand (cols = m), so each lane ends up with 4 consecutive n per accumulator tile for ONE output
// row m.  W rows are additionally permuted when they are placed in LDS (row n_local = 16g+4t+r sits at
// LDS row 16t+4g+r) so the 4 n-tiles of a lane are 16 CONTIGUOUS output columns: 32-byte bf16 /
// 64-byte fp32 stores per lane, and full 128-B lines per row across a wave.
#include <stdlib.h>

#include "gemm_common.h"
#include "host_util.h"

namespace clibd {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage
constexpr int GEMM_THREADS = 256;

// LDS row r (0..127) of the W tile holds tile-local output column perm(r)
__device__ __forceinline__ int w_row_perm(int r) {
    const int r6 = r & 63;
    return (r & 64) + w_col_of(r6 >> 4, r6 & 15);
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> (split, tile) mapping, XCD-aware: blocks that share an XCD (id % 8) get a contiguous
    // chunk of tile ids, n fastest, so neighbouring tiles share the A panel and W stays in that XCD's L2.
    const int ntiles = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    const int split = bid / ntiles;
    bid -= split * ntiles;
    int tile_m, tile_n;
    tile_coords(bid, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kt0 = split * p.ktiles_per_split;
    const int nk = min(p.ktiles_per_split, p.K / BK - p.hole_nkt - kt0);  // K-tiles this block walks (the hole is not counted)

    // ---- per-lane source pointers for this wave's 4 A pieces and 4 W pieces (1 KiB = 8 rows each)
    const int prow = lane >> 3;                 // row inside the piece
    const int chunk = (lane & 7) ^ prow;        // logical 16-B chunk stored at LDS slot (lane&7) of that row
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + prow;  // LDS row 0..127
        const int gm = min(m0 + r, p.M - 1);
        const int gn = min(n0 + w_row_perm(r), p.N - 1);
        a_src[j] = (const char*)(p.A + (size_t)gm * p.lda) + chunk * 16;
        w_src[j] = (const char*)(p.W + (size_t)gn * p.ldw) + chunk * 16;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes inside a tile): row*128 + ((chunk ^ (row&7)) << 4)
    const int frow = lane & 15, fch = lane >> 4;
    int a_off[4], w_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + frow;
        const int rw = wn * 64 + t * 16 + frow;
        a_off[t] = ra * 128 + ((fch ^ (ra & 7)) << 4);
        w_off[t] = rw * 128 + ((fch ^ (rw & 7)) << 4);
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * (2 * TILE_BYTES);
        int kta = kt0 + kt;
        if (kta >= p.hole_kt) kta += p.hole_nkt;   // skip the hole (hole_nkt == 0: no-op)
        const size_t koff = (size_t)kta * (BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src[j] + koff, base + (wave * 4 + j) * 1024);
            glds16(w_src[j] + koff, base + TILE_BYTES + (wave * 4 + j) * 1024);
        }
    };

    if (nk > 0) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // emits s_waitcnt vmcnt(0): tile kt has landed for every wave; buffer (kt+1)&1 is free
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* ab = smem + (kt & 1) * (2 * TILE_BYTES);
        const char* wb = ab + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(ab + (a_off[t] ^ (kk << 6)));
                wf[t] = *(const bf16x8*)(wb + (w_off[t] ^ (kk << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    const clibd_gemm_epilogue& ep = p.ep;
    // ---- LoRA rank-8 update as one extra (zero-padded) k-step: lanes with k-chunk 0 carry U[m,0:8] / V[n,0:8]
    if (ep.rank_u != nullptr && split == 0) {
        bf16x8 uf[4], vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uf[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            vf[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (fch == 0) {
                const int gm = min(m0 + wm * 64 + t * 16 + frow, p.M - 1);
                const int gn = min(n0 + w_row_perm(wn * 64 + t * 16 + frow), p.N - 1);
                uf[t] = *(const bf16x8*)((const unsigned short*)ep.rank_u + (size_t)gm * ep.ld_rank_u);
                vf[t] = *(const bf16x8*)((const unsigned short*)ep.rank_v + (size_t)gn * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j], uf[i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: lane owns row m = m0 + 64wm + 16i + (lane&15), columns nb .. nb+15 (e = 4*j + reg)
    const int nb = n0 + wn * 64 + 16 * fch;
    if (nb >= p.N) return;  // N % 16 == 0, so a lane's 16 columns are all in or all out
    float bias[16];
    load_bias16(ep, nb, split == 0, bias);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + frow;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] + bias[4 * j + r];
        store_row16(ep, m, nb, v);
    }
}

// ---- bf16 transpose with zero padding ---------------------------------------------------------------
// general form (any alignment): 2-byte accesses through a padded LDS tile
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* in, int ld_in, int R, int C,
                                                             unsigned short* out, int ld_out) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;   // out row = c, out col = r
        if (c < C && r < ld_out) out[(size_t)c * ld_out + r] = tile[tx][i];
    }
}

// fast form (C % 8 == 0, ld_in % 8 == 0, ld_out % 8 == 0, 16-byte aligned bases): 16-byte global loads and stores on both
// sides, a (64 RB) x 64 tile in LDS with a 132-byte row stride (column reads of 8 rows hit 8 different banks), and — optionally —
// the column sums of the input (the bias gradient of y = x W^T + b is colsum(dy); the weight gradient needs dy^T anyway, so
// the full fine-tune backward gets db for free instead of a second pass over dy).  Rows R .. ld_out-1 of the output are zero.
// RB = 4 (256-row tiles): 8 independent 16-byte loads per thread in flight before the barrier and 512 contiguous bytes per
// output row and block; the 64-row tile of the first version (2 loads in flight, 16 KB per block) ran at 3.2 TB/s.
template <int RB>
__global__ __launch_bounds__(256) void transpose_colsum_bf16_kernel(const unsigned short* __restrict__ in, int ld_in, int R, int C,
                                                                    unsigned short* __restrict__ out, int ld_out,
                                                                    float* __restrict__ colsum, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * RB * 66];
    const int r0 = blockIdx.y * (64 * RB), c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    uint4 v[2 * RB];
#pragma unroll
    for (int k = 0; k < 2 * RB; ++k) {
        const int row = (t >> 3) + 32 * k, ch = t & 7;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + row < R && c0 + 8 * ch < C) v[k] = *(const uint4*)(in + (size_t)(r0 + row) * ld_in + c0 + 8 * ch);
    }
#pragma unroll
    for (int k = 0; k < 2 * RB; ++k) {
        const int row = (t >> 3) + 32 * k, ch = t & 7;
        unsigned* d = (unsigned*)(tile + row * 66 + 8 * ch);   // 4-byte aligned: 132 * row + 16 * ch
        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = (t >> 3) + 32 * k, rch = t & 7;
        float sum = 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tile[(64 * rb + 8 * rch + j) * 66 + c];
            if (c0 + c < C && r0 + 64 * rb + 8 * rch < ld_out) {
                uint4 o;
                o.x = (unsigned)e[0] | ((unsigned)e[1] << 16); o.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
                o.z = (unsigned)e[4] | ((unsigned)e[5] << 16); o.w = (unsigned)e[6] | ((unsigned)e[7] << 16);
                *(uint4*)(out + (size_t)(c0 + c) * ld_out + r0 + 64 * rb + 8 * rch) = o;
            }
            if (colsum != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += bf2f(e[j]);
            }
        }
        if (colsum != nullptr) {   // 8 adjacent lanes (rch = 0..7) hold the 64 RB rows of column c
            sum += dpp_mov<DPP_QUAD_XOR1>(sum);
            sum += dpp_mov<DPP_QUAD_XOR2>(sum);
            sum += dpp_mov<DPP_ROW_HALF_MIRROR>(sum);
            if (rch == 0 && c0 + c < C) {
                if (partials != nullptr) partials[(size_t)blockIdx.y * C + c0 + c] = sum;   // summed in row-block order by colsum_partials_kernel
                else atomicAdd(colsum + c0 + c, sum);
            }
        }
    }
}

// colsum[c] += sum_b partials[b, c] in the fixed order b = 0, 1, ... (four interleaved chains, combined in a fixed order): the bias
// gradient of the workspace form of the transpose (clibd_transpose_colsum_bf16_ws) repeats bit for bit from run to run.
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partials, int nblk, int C, float* __restrict__ colsum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nblk; b += 4) {
        s0 += partials[(size_t)b * C + c];
        s1 += partials[(size_t)(b + 1) * C + c];
        s2 += partials[(size_t)(b + 2) * C + c];
        s3 += partials[(size_t)(b + 3) * C + c];
    }
    for (; b < nblk; ++b) s0 += partials[(size_t)b * C + c];
    colsum[c] += (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* in, unsigned short* out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const f32x4 v = *(const f32x4*)(in + i);
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)(out + i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n & ~(size_t)3; j < n; ++j) out[j] = f2bf(in[j]);
}

__global__ __launch_bounds__(256) void cast_transpose_f32_bf16_kernel(const float* in, int R, int C,
                                                                      unsigned short* out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * R + r] = f2bf(tile[tx][i]);
    }
}

}  // namespace clibd

using namespace clibd;

static int gemm_impl(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int hole_k0, int hole_len,
                     const clibd_gemm_epilogue* ep, void* stream, void* tail_ws = nullptr, size_t tail_ws_bytes = 0) {
    if (!A || !W || !ep) return set_error(CLIBD_EINVAL, "gemm: null pointer");
    if (hole_len < 0 || hole_k0 < 0 || hole_k0 % BK || hole_len % BK || hole_k0 + hole_len > K || (hole_len > 0 && hole_len >= K))
        return set_error(CLIBD_EINVAL, "gemm: bad K hole (multiples of 64 inside [0, K))");
    if (M <= 0 || N <= 0 || K <= 0) return set_error(CLIBD_EINVAL, "gemm: non-positive shape");
    if (K % BK != 0) return set_error(CLIBD_EINVAL, "gemm: K must be a multiple of 64");
    if (N % 16 != 0) return set_error(CLIBD_EINVAL, "gemm: N must be a multiple of 16");
    if (lda % 8 != 0 || ldw % 8 != 0 || lda < K || ldw < K) return set_error(CLIBD_EINVAL, "gemm: bad lda/ldw");
    if (!aligned16(A) || !aligned16(W)) return set_error(CLIBD_EINVAL, "gemm: operands must be 16-byte aligned");
    if (!ep->out_bf16 && !ep->out_f32 && !ep->out_pre_bf16) return set_error(CLIBD_EINVAL, "gemm: no output");
    if ((ep->rank_u == nullptr) != (ep->rank_v == nullptr)) return set_error(CLIBD_EINVAL, "gemm: rank_u/rank_v");
    if (ep->rank_u && (ep->ld_rank_u < 8 || ep->ld_rank_u % 8)) return set_error(CLIBD_EINVAL, "gemm: ld_rank_u");
    if ((ep->act == CLIBD_ACT_GELU_GRAD || ep->act == CLIBD_ACT_MUL_AUX || ep->act == CLIBD_ACT_ADD_AUX) &&
        (!ep->aux_bf16 || ep->ld_aux % 8 || ep->ld_aux < N))
        return set_error(CLIBD_EINVAL, "gemm: GELU_GRAD / MUL_AUX / ADD_AUX need aux_bf16 with ld_aux >= N, % 8");
    if (ep->act == CLIBD_ACT_MUL_AUX_U8 && (!ep->aux_bf16 || ep->ld_aux % 16 || ep->ld_aux < N))
        return set_error(CLIBD_EINVAL, "gemm: MUL_AUX_U8 needs aux (one byte per element) with ld_aux >= N, % 16");
    const bool e12 = ep->act == CLIBD_ACT_GELU_SAVE_GRAD_E12 || ep->act == CLIBD_ACT_MUL_AUX_E12;
    if (e12 && (N % 8 != 0)) return set_error(CLIBD_EINVAL, "gemm: the e4m7 gelu' forms need N % 8 == 0");
    if (ep->act == CLIBD_ACT_MUL_AUX_E12 && (!ep->aux_bf16 || ep->ld_aux % 4 || ep->ld_aux < 3 * (N / 2)))
        return set_error(CLIBD_EINVAL, "gemm: MUL_AUX_E12 needs aux (1.5 bytes per element) with ld_aux >= 3N/2 bytes, % 4");
    if (ep->act == CLIBD_ACT_GELU_SAVE_GRAD_E12 && (!ep->out_pre_bf16 || ep->ld_pre % 4 || ep->ld_pre < 3 * (N / 2)))
        return set_error(CLIBD_EINVAL, "gemm: GELU_SAVE_GRAD_E12 needs out_pre (1.5 bytes per element) with ld_pre >= 3N/2 bytes, % 4");
    if ((ep->act == CLIBD_ACT_GELU_SAVE_GRAD || ep->act == CLIBD_ACT_GELU_SAVE_GRAD_U8) && !ep->out_pre_bf16)
        return set_error(CLIBD_EINVAL, "gemm: GELU_SAVE_GRAD needs out_pre_bf16");
    if (ep->act < 0 || ep->act > CLIBD_ACT_MUL_AUX_E12) return set_error(CLIBD_EINVAL, "gemm: bad act");
    if (ep->residual_f32 && (ep->ld_res % 4 || ep->ld_res < N)) return set_error(CLIBD_EINVAL, "gemm: ld_res");
    if (ep->out_pre_bf16 && (ep->ld_pre % (ep->act == CLIBD_ACT_GELU_SAVE_GRAD_U8 ? 16 : 8) || ep->ld_pre < N)) return set_error(CLIBD_EINVAL, "gemm: ld_pre");
    if (ep->out_bf16 && (ep->ld_out_bf16 % 8 || ep->ld_out_bf16 < N)) return set_error(CLIBD_EINVAL, "gemm: ld_out_bf16");
    if (ep->out_f32 && (ep->ld_out_f32 % 4 || ep->ld_out_f32 < N)) return set_error(CLIBD_EINVAL, "gemm: ld_out_f32");
    if (ep->drop_thr16 < 0 || ep->drop_thr16 > 65535 || (ep->drop_thr16 > 0 && (ep->drop_ld < N || (ep->drop_ld & 1) || ep->split_k > 1)))
        return set_error(CLIBD_EINVAL, "gemm: bad dropout parameters");
    const int split = ep->split_k < 1 ? 1 : ep->split_k;
    if (split > 1 && (ep->out_bf16 || ep->out_pre_bf16 || ep->act != CLIBD_ACT_NONE || ep->residual_f32 || !ep->out_f32))
        return set_error(CLIBD_EINVAL, "gemm: split_k > 1 supports only a plain fp32 accumulate output");
    const void* ptrs[] = {ep->bias, ep->rank_u, ep->rank_v, ep->aux_bf16, ep->residual_f32, ep->out_pre_bf16,
                          ep->out_bf16, ep->out_f32};
    for (const void* q : ptrs)
        if (q && !aligned16(q)) return set_error(CLIBD_EINVAL, "gemm: epilogue pointers must be 16-byte aligned");

    GemmParams p{};
    p.A = (const unsigned short*)A; p.W = (const unsigned short*)W;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    const int ktiles = (K - hole_len) / BK;
    p.ktiles_per_split = (ktiles + split - 1) / split;
    p.ep = *ep;
    p.ep.split_k = split;
    p.splits = 1; p.nk_split = 0; p.split_stride = 0;
    p.hole_kt = hole_len > 0 ? hole_k0 / BK : 0x7fffffff;
    p.hole_nkt = hole_len / BK;
    if (tail_ws != nullptr) {   // stream-K tail workspace: [256 flag words | fp32 partial tiles]; ignored when the shape has no use for it
        const size_t need = gemm256_tail_workspace_bytes(M, N, K);
        if (need > 0) {
            if (tail_ws_bytes < need || !aligned16(tail_ws)) return set_error(CLIBD_EINVAL, "gemm: tail workspace too small or misaligned (clibd_gemm_tail_workspace_bytes)");
            p.sk_flags = (unsigned*)tail_ws;
            p.sk_ws = (float*)((char*)tail_ws + 1024);
        }
    }
    // kernel choice (CLIBD_GEMM_KERNEL=1 forces the 128x128 kernel: tuning aid).  A third shape — 256x128x32 tiles, 3-stage
    // ring, two workgroups per CU so epilogues overlap across workgroups — was built and measured: 800 TF at K=768 and
    // 920 TF at K=3072 against 944 / 1300 TF for the 256x256 8-phase kernel, so it was dropped.
    static const int forced = [] { const char* e = getenv("CLIBD_GEMM_KERNEL"); return e ? atoi(e) : 0; }();
    const bool fold_epi = ep->row_sums != nullptr || ep->row_stats != nullptr;   // LayerNorm -> Linear fold forms: 256x256 kernel only
    if (fold_epi) {
        if (epilogue_kind(p.ep) < 0)
            return set_error(CLIBD_EINVAL, "gemm: LN-fold epilogue: row_sums needs bias + residual_f32 + out_f32 + out_bf16 (act NONE); row_stats needs "
                                           "col_sum_w + bias + GELU_SAVE_GRAD with out_pre_bf16 + out_bf16; no rank update, dropout or split-K");
        if ((ep->row_sums && !aligned16(ep->row_sums)) || (ep->row_stats && !aligned16(ep->row_stats)) || (ep->col_sum_w && !aligned16(ep->col_sum_w)))
            return set_error(CLIBD_EINVAL, "gemm: LN-fold epilogue pointers must be 16-byte aligned");
        if (forced == 1 || hole_len != 0 || !gemm256_try_launch(p, (hipStream_t)stream))
            return set_error(CLIBD_EINVAL, "gemm: the LN-fold epilogues exist in the 256x256 kernel only (M >= 1024, N % 256 == 0, K % 128 == 0, >= 128 tiles)");
        return check_launch("gemm256_bf16_nt");
    }
    if (forced != 1 && hole_len == 0 && gemm256_try_launch(p, (hipStream_t)stream)) return check_launch("gemm256_bf16_nt");
    const long long nblocks = (long long)p.tiles_m * p.tiles_n * split;
    if (nblocks > 0x7fffffffLL) return set_error(CLIBD_EINVAL, "gemm: grid too large");
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   4 * TILE_BYTES) == hipSuccess;
    }();
    (void)attr_ok;
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3((unsigned)nblocks), dim3(GEMM_THREADS), 4 * TILE_BYTES,
                       (hipStream_t)stream, p);
    return check_launch("gemm_bf16_nt");
}

extern "C" int clibd_gemm_bf16_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                                  const clibd_gemm_epilogue* ep, void* stream) {
    return gemm_impl(A, lda, W, ldw, M, N, K, 0, 0, ep, stream);
}

extern "C" size_t clibd_gemm_tail_workspace_bytes(int M, int N, int K) { return gemm256_tail_workspace_bytes(M, N, K); }

extern "C" int clibd_gemm_bf16_nt_ws(const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                                     const clibd_gemm_epilogue* ep, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_impl(A, lda, W, ldw, M, N, K, 0, 0, ep, stream, workspace, workspace_bytes);
}

extern "C" int clibd_gemm_bf16_nt_khole(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int hole_k0, int hole_len,
                                        const clibd_gemm_epilogue* ep, void* stream) {
    return gemm_impl(A, lda, W, ldw, M, N, K, hole_k0, hole_len, ep, stream);
}

extern "C" int clibd_gemm_fp8_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* col_scale,
                                 float out_fp8_scale, const clibd_gemm_epilogue* ep, void* stream) {
    if (!A || !W || !ep || !col_scale) return set_error(CLIBD_EINVAL, "gemm_fp8: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K) return set_error(CLIBD_EINVAL, "gemm_fp8: bad shape");
    if ((lda & 15) || (ldw & 15) || !aligned16(A) || !aligned16(W) || !aligned16(col_scale)) return set_error(CLIBD_EINVAL, "gemm_fp8: alignment (lda, ldw % 16)");
    if (ep->ld_out_bf16 & 7 || ep->ld_out_f32 & 3 || ep->ld_res & 3 || ep->ld_pre & 7 || ep->ld_rank_u & 7)
        return set_error(CLIBD_EINVAL, "gemm_fp8: leading dimensions");
    if ((ep->out_bf16 && !aligned16(ep->out_bf16)) || (ep->out_f32 && !aligned16(ep->out_f32)) || (ep->out_pre_bf16 && !aligned16(ep->out_pre_bf16)) ||
        (ep->residual_f32 && !aligned16(ep->residual_f32)) || (ep->rank_u && !aligned16(ep->rank_u)) || (ep->rank_v && !aligned16(ep->rank_v)) ||
        (ep->bias && !aligned16(ep->bias)))
        return set_error(CLIBD_EINVAL, "gemm_fp8: pointer alignment");
    if ((ep->rank_u == nullptr) != (ep->rank_v == nullptr)) return set_error(CLIBD_EINVAL, "gemm_fp8: rank_u / rank_v must come together");
    GemmParams p{};
    p.A = (const unsigned short*)A; p.W = (const unsigned short*)W;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    p.ep = *ep;
    if (p.ep.split_k < 1) p.ep.split_k = 1;
    p.fp8 = 1; p.col_scale = col_scale; p.out_fp8_scale = out_fp8_scale;
    if (!gemm256_fp8_launch(p, (hipStream_t)stream))
        return set_error(CLIBD_EINVAL, "gemm_fp8: shape / epilogue not supported (N % 256, K % 256, K >= 512, bias required; forms: "
                                       "[lora] bias->bf16 | bias->gelu->fp8 + gelu' | bias[->dropout]+residual->f32)");
    return check_launch("gemm256_fp8_nt");
}

extern "C" int clibd_gemm_fp8_dgrad_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* col_scale,
                                       const float* a_row_dequant, float out_fp8_scale, const clibd_gemm_epilogue* ep, void* stream) {
    if (!A || !W || !ep || !col_scale) return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K) return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: bad shape");
    if ((lda & 15) || (ldw & 15) || !aligned16(A) || !aligned16(W) || !aligned16(col_scale) || (a_row_dequant && !aligned16(a_row_dequant)))
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: alignment (lda, ldw % 16)");
    if (ep->ld_out_bf16 & 7 || ep->ld_aux & 7) return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: leading dimensions");
    if (ep->act == CLIBD_ACT_MUL_AUX_U8 && (!ep->aux_bf16 || ep->ld_aux % 16 || ep->ld_aux < N))
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: MUL_AUX_U8 needs aux (one byte per element) with ld_aux >= N, % 16");
    if (!ep->out_bf16 || !aligned16(ep->out_bf16) || (ep->aux_bf16 && !aligned16(ep->aux_bf16))) return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: out_bf16 / aux alignment");
    if (ep->out_f32 || ep->residual_f32 || ep->bias || ep->rank_u || ep->rank_v || ep->row_sums || ep->row_stats || ep->col_sum_w)
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: only out_bf16 [+ aux_bf16] epilogues");
    // ABI 5: out_pre_bf16 on the MUL_AUX forms = a second, bf16 output of the de-scaled value (the weight gradient's operand under full fine-tune)
    const bool mul_form = ep->act == CLIBD_ACT_MUL_AUX || ep->act == CLIBD_ACT_MUL_AUX_U8 || ep->act == CLIBD_ACT_MUL_AUX_E12;
    if (ep->act == CLIBD_ACT_MUL_AUX_E12 && (!ep->aux_bf16 || ep->ld_aux % 4 || ep->ld_aux < 3 * (N / 2)))
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: MUL_AUX_E12 needs aux (1.5 bytes per element) with ld_aux >= 3N/2 bytes, % 4");
    if (ep->out_pre_bf16 && (!mul_form || !a_row_dequant || !aligned16(ep->out_pre_bf16) || (ep->ld_pre & 7) || ep->ld_pre < N))
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: out_pre_bf16 (bf16 copy) comes with MUL_AUX[_U8] and a_row_dequant only; ld_pre >= N, % 8; 16-byte aligned");
    GemmParams p{};
    p.A = (const unsigned short*)A; p.W = (const unsigned short*)W;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    p.ep = *ep;
    if (p.ep.split_k < 1) p.ep.split_k = 1;
    p.fp8 = 1; p.col_scale = col_scale; p.out_fp8_scale = out_fp8_scale; p.a_row_dequant = a_row_dequant;
    p.dual_bf16 = (unsigned short*)p.ep.out_pre_bf16; p.ld_dual = p.ep.ld_pre;
    p.ep.out_pre_bf16 = nullptr; p.ep.ld_pre = 0;
    if (!gemm256_fp8_dgrad_launch(p, (hipStream_t)stream))
        return set_error(CLIBD_EINVAL, "gemm_fp8_dgrad: shape / epilogue not supported (M % 4, N % 256, K % 256, K >= 512; forms: -> bf16 | + aux_bf16 -> bf16 "
                                       "(both need a_row_dequant) | x aux (bf16 gelu' or its one-byte code) -> fp8 (needs out_fp8_scale > 0))");
    return check_launch("gemm256_fp8_dgrad_nt");
}

// ---- split-K with a partials workspace (weight gradients of the full fine-tune mode) -----------------------------------
namespace clibd {
// out[i] (+)= sum_s partials[s * n + i]
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ partials, int splits, size_t n4,
                                                            float* __restrict__ out, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = accumulate ? ((const f32x4*)out)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += ((const f32x4*)partials)[(size_t)k * n4 + i];
        ((f32x4*)out)[i] = s;
    }
}
}  // namespace clibd

extern "C" size_t clibd_gemm_splitk_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)256 * (size_t)M * (size_t)N * sizeof(float) / (size_t)(((M + 255) / 256) * ((N + 255) / 256));  // <= 256 work items per round
}

extern "C" int clibd_gemm_bf16_nt_splitk(const void* A, int lda, const void* W, int ldw, int M, int N, int K, float* out_f32, int ld_out,
                                         int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !W || !out_f32 || !workspace) return set_error(CLIBD_EINVAL, "gemm_splitk: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ld_out != N) return set_error(CLIBD_EINVAL, "gemm_splitk: bad shape (out must be dense [M,N])");
    if ((lda & 7) || (ldw & 7) || !aligned16(A) || !aligned16(W) || !aligned16(out_f32) || !aligned16(workspace) || (N & 3))
        return set_error(CLIBD_EINVAL, "gemm_splitk: alignment");
    GemmParams p{};
    p.A = (const unsigned short*)A; p.W = (const unsigned short*)W;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    const int splits = gemm256_splitk_launch(p, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
    if (splits <= 0) return set_error(CLIBD_EINVAL, "gemm_splitk: shape not supported (need N % 256 == 0, K % 128 == 0, K >= 512, workspace)");
    if (int e = check_launch("gemm256_splitk")) return e;
    const size_t n4 = (size_t)M * N / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits, n4,
                       out_f32, accumulate);
    return check_launch("reduce_splits");
}

extern "C" int clibd_gemm_bf16_tn_splitk(const void* A, int lda, const void* B, int ldb, int M, int Na, int Nb, float* out_f32, int ld_out,
                                         int accumulate, float* colsum_a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || !out_f32 || !workspace) return set_error(CLIBD_EINVAL, "gemm_tn_splitk: null pointer");
    if (M <= 0 || Na <= 0 || Nb <= 0 || lda < Na || ldb < Nb || ld_out != Nb) return set_error(CLIBD_EINVAL, "gemm_tn_splitk: bad shape (out must be dense [Na,Nb])");
    if ((lda & 7) || (ldb & 7) || !aligned16(A) || !aligned16(B) || !aligned16(out_f32) || !aligned16(workspace))
        return set_error(CLIBD_EINVAL, "gemm_tn_splitk: alignment");
    const int splits = gemm256_tn_splitk_launch((const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, Na, Nb, (float*)workspace,
                                                workspace_bytes / sizeof(float), colsum_a, (hipStream_t)stream);
    if (splits <= 0) return set_error(CLIBD_EINVAL, "gemm_tn_splitk: shape not supported (need M % 128 == 0, M >= 256, Na % 256 == 0, Nb % 256 == 0, workspace)");
    if (int e = check_launch("gemm256_tn")) return e;
    const size_t n4 = (size_t)Na * Nb / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits, n4,
                       out_f32, accumulate);
    return check_launch("reduce_splits");
}

static inline int transpose_row_tile(int ld_out) { return ld_out >= 1024 ? 256 : 64; }

static int transpose_impl(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum, void* stream,
                          float* partials = nullptr, size_t partials_bytes = 0) {
    if (!in || !out || R <= 0 || C <= 0 || ld_in < C || ld_out < R) return set_error(CLIBD_EINVAL, "transpose: bad args");
    dim3 grid((C + 63) / 64, (ld_out + 63) / 64);
    if (grid.y > 65535u) return set_error(CLIBD_EINVAL, "transpose: too many rows for one launch");
    const bool fast = (C % 8 == 0) && (ld_in % 8 == 0) && (ld_out % 8 == 0) && aligned16(in) && aligned16(out);
    if (fast) {
        const int tile_rows = transpose_row_tile(ld_out);
        const int nblk = (ld_out + tile_rows - 1) / tile_rows;
        if (partials != nullptr && (partials_bytes < (size_t)nblk * C * sizeof(float) || ((uintptr_t)partials & 3)))
            return set_error(CLIBD_EINVAL, "transpose_colsum: workspace too small (clibd_transpose_colsum_workspace_bytes)");
        if (tile_rows == 256) {   // long row dimension (activations): 256-row tiles
            dim3 grid4((C + 63) / 64, nblk);
            hipLaunchKernelGGL(transpose_colsum_bf16_kernel<4>, grid4, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, ld_in, R, C,
                               (unsigned short*)out, ld_out, colsum, partials);
        } else {
            hipLaunchKernelGGL(transpose_colsum_bf16_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, ld_in, R, C,
                               (unsigned short*)out, ld_out, colsum, partials);
        }
        if (int e = check_launch("transpose_colsum_bf16")) return e;
        if (partials != nullptr && colsum != nullptr) {
            hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)partials, nblk, C, colsum);
            return check_launch("colsum_partials");
        }
        return CLIBD_OK;
    }
    if (colsum != nullptr) return set_error(CLIBD_EINVAL, "transpose_colsum: needs C, ld_in, ld_out multiples of 8 and 16-byte aligned bases");
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in,
                       ld_in, R, C, (unsigned short*)out, ld_out);
    return check_launch("transpose_bf16");
}

extern "C" int clibd_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream) {
    return transpose_impl(in, ld_in, R, C, out, ld_out, nullptr, stream);
}

extern "C" int clibd_transpose_colsum_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum, void* stream) {
    if (!colsum) return set_error(CLIBD_EINVAL, "transpose_colsum: null colsum");
    return transpose_impl(in, ld_in, R, C, out, ld_out, colsum, stream);
}

extern "C" size_t clibd_transpose_colsum_workspace_bytes(int ld_out, int C) {
    if (ld_out <= 0 || C <= 0) return 0;
    const int tile_rows = transpose_row_tile(ld_out);
    return (size_t)((ld_out + tile_rows - 1) / tile_rows) * (size_t)C * sizeof(float);
}

extern "C" int clibd_transpose_colsum_bf16_ws(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    if (!colsum || !workspace) return set_error(CLIBD_EINVAL, "transpose_colsum_ws: null colsum / workspace");
    return transpose_impl(in, ld_in, R, C, out, ld_out, colsum, stream, (float*)workspace, workspace_bytes);
}

extern "C" int clibd_cast_f32_to_bf16(const float* in, void* out, size_t n, void* stream) {
    if (!in || !out) return set_error(CLIBD_EINVAL, "cast: null pointer");
    if (n == 0) return CLIBD_OK;
    if (!aligned16(in) || ((uintptr_t)out & 7)) return set_error(CLIBD_EINVAL, "cast: alignment");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in,
                       (unsigned short*)out, n);
    return check_launch("cast_f32_to_bf16");
}

extern "C" int clibd_cast_transpose_f32_to_bf16(const float* in, int R, int C, void* out, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return set_error(CLIBD_EINVAL, "cast_transpose: bad args");
    dim3 grid((C + 63) / 64, (R + 63) / 64);
    hipLaunchKernelGGL(cast_transpose_f32_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, R, C,
                       (unsigned short*)out);
    return check_launch("cast_transpose_f32_to_bf16");
}
