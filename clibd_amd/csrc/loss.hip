// K9: all-pairs similarity + soft-target (label-equality) cross-entropy, row-block form, for gfx950.
//
//   S[i,j] = scale * <x_i, y_j>          x: [Nx,D] rows owned by this rank (global row offset row0), y: [N,D] all rows
//   T[i,j] = (labels[row0+i] == labels[j])                    (never materialised)
//   loss   = sum_i ( LSE_j(S[i,:]) * sum_j T[i,j]  -  sum_j T[i,j] S[i,j] )      == sum_i CE(S[i,:], T[i,:])
//
// The similarity runs on the bf16 MFMA GEMM with the split-bf16 trick: x = hi + lo, y = hi + lo (both bf16),
// S ~ hi·hi + hi·lo + lo·hi, expressed as ONE NT GEMM with K = 3D over the images [hi|hi|lo] and [hi|lo|hi]
// (error ~2^-16 relative instead of 2^-8: the reference evaluates this product in fp32, loss_func.py:189-190).
// Row statistics use one wave64 per row with DPP / permlane reductions.  The Nx x N similarity block of THIS rank's rows is
// the kernel's fp32 scratch (caller-owned workspace: 16 MiB at Nx = N = 2048, 32 MiB at Nx = 1024, N = 8192); what is never
// materialised is the N x N target matrix, the log-softmax temporaries and the second (transposed) logits of the reference.
// Backward forms the coefficient matrix G = w * (tsum_i * softmax(S)_ij - T_ij) once, split like the forward operands
// (G = Ghi + Glo), and feeds it to the same GEMM with the K-concatenated images
//     dX = scale * [Ghi | Ghi | Glo] . [Yhi | Ylo | Yhi]^T          dY = scale * [Ghi^T | Glo^T | Ghi^T] . [Xhi | Xhi | Xlo]^T
// so the feature gradients carry the forward's ~2^-16 relative accuracy (the reference computes loss and gradients in fp32).
#include "common.h"
#include "../../include/clibd_hip.h"
#include "host_util.h"

namespace clibd {

// img3[r, 0:D] = hi, img3[r, D:2D] = MID, img3[r, 2D:3D] = LAST  with (MID,LAST) = (hi,lo) for x, (lo,hi) for y
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ in, int R, int D, int x_side,
                                                     unsigned short* __restrict__ out) {
    const size_t total = (size_t)R * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / D;
        const int c = (int)(i - r * D);
        const float v = in[i];
        const unsigned short hi = f2bf(v);
        const unsigned short lo = f2bf(v - bf2f(hi));
        unsigned short* o = out + r * 3 * (size_t)D;
        o[c] = hi;
        o[D + c] = x_side ? hi : lo;
        o[2 * D + c] = x_side ? lo : hi;
    }
}

struct RowStat {
    float m, s, tsum, tdot;
};

// one wave per row: online log-sum-exp + target sums
__global__ __launch_bounds__(256) void softce_rows_fwd_kernel(const float* __restrict__ S, int ldS, int Nx, int N,
                                                              const int64_t* __restrict__ labels, int row0,
                                                              const float* __restrict__ scale_ptr,
                                                              float* __restrict__ lse, float* __restrict__ tsum_out,
                                                              float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Nx) return;
    const int64_t lab = labels[row0 + row];
    const float scale = *scale_ptr;
    const float* sr = S + (size_t)row * ldS;
    float m = -3.0e38f, s = 0.f, ts = 0.f, td = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float v = sr[j] * scale;
        if (v > m) {
            s = s * __expf(m - v) + 1.0f;
            m = v;
        } else {
            s += __expf(v - m);
        }
        if (labels[j] == lab) {
            ts += 1.0f;
            td += v;
        }
    }
    const float mw = wave_max(m);
    s = wave_sum(s * __expf(m - mw));
    ts = wave_sum(ts);
    td = wave_sum(td);
    if (lane == 0) {
        const float l = mw + __logf(s);
        lse[row] = l;
        tsum_out[row] = ts;
        row_loss[row] = l * ts - td;   // summed in a fixed order by reduce_rows_add_kernel (round 6: no float atomics on the loss path)
    }
}

// out[0] += sum_i v[i], one workgroup, a fixed summation order (thread t takes i = t, t + 256, ...; then an LDS tree): the loss value and
// the temperature gradient repeat bit for bit from run to run.  Rounds 1-5 accumulated both with one float atomic per row, whose order —
// and with it the last bits of d(logit_scale), which AdamW turns into a different trajectory — changed from launch to launch.
__global__ __launch_bounds__(256) void reduce_rows_add_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += part[0];
}

// g_ij = w * (tsum_i * exp(scale*R_ij - lse_i) - T_ij) with R = raw similarities;
// scale * g_ij = hi + lo (bf16 each) is written as the row image [hi | hi | lo] of width 3 * Np (zero padded past N);
// row_ds[i] = sum_j g_ij * R_ij   (dscale += their fixed-order sum, reduce_rows_add_kernel)
__global__ __launch_bounds__(256) void softce_rows_bwd_kernel(const float* __restrict__ S, int ldS, int Nx, int N,
                                                              const int64_t* __restrict__ labels, int row0,
                                                              const float* __restrict__ lse, const float* __restrict__ tsum,
                                                              float w, const float* __restrict__ wscale_ptr,
                                                              const float* __restrict__ scale_ptr,
                                                              unsigned short* __restrict__ G,
                                                              int Np, float* __restrict__ row_ds) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Nx) return;
    const int64_t lab = labels[row0 + row];
    const float l = lse[row], ts = tsum[row];
    const float scale = *scale_ptr;
    if (wscale_ptr != nullptr) w *= *wscale_ptr;
    const float* sr = S + (size_t)row * ldS;
    unsigned short* gr = G + (size_t)row * 3 * Np;
    float ds = 0.f;
    for (int j = lane; j < Np; j += 64) {
        float g = 0.f;
        if (j < N) {
            const float v = sr[j];
            g = w * (ts * __expf(v * scale - l) - (labels[j] == lab ? 1.0f : 0.0f));
            ds += g * v;
        }
        const float gs = g * scale;
        const unsigned short hi = f2bf(gs);
        gr[j] = hi;
        gr[Np + j] = hi;
        gr[2 * Np + j] = f2bf(gs - bf2f(hi));
    }
    ds = wave_sum(ds);
    if (lane == 0) row_ds[row] = ds;
}

// out[c, k * Rp + r] = in[r, seg(k) * W + c] for the three column segments of a [R, 3W] image (seg(k) = 2 bits of `map` each),
// c < C <= W, r < Rp with zeros for r >= R: the K-concatenated operand images of the backward GEMMs, transposed in one launch.
__global__ __launch_bounds__(256) void transpose3_bf16_kernel(const unsigned short* __restrict__ in, int R, int W, int C, int map,
                                                              unsigned short* __restrict__ out, int Rp) {
    __shared__ unsigned short tile[64][66];
    const int k = blockIdx.z;
    const int seg = (map >> (2 * k)) & 3;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * 3 * W + (size_t)seg * W + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(size_t)c * 3 * Rp + (size_t)k * Rp + r] = tile[tx][i];
    }
}

static int launch_transpose3(const unsigned short* in, int R, int W, int C, int map, unsigned short* out, int Rp, hipStream_t st) {
    dim3 grid((C + 63) / 64, (Rp + 63) / 64, 3);
    hipLaunchKernelGGL(transpose3_bf16_kernel, grid, dim3(256), 0, st, in, R, W, C, map, out, Rp);
    return check_launch("softce transpose3");
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int pad64(int v) { return (v + 63) / 64 * 64; }
static inline int pad16(int v) { return (v + 15) / 16 * 16; }

struct LossWs {
    unsigned short *x3, *y3;   // [Nx,3D], [N,3D]
    float *S, *lse, *tsum;     // [Nx,N], [Nx], [Nx]
    float *rows;               // [Nx]: per-row loss terms (forward), then per-row d(scale) terms (backward)
    unsigned short *G, *GT;    // [Nx,3Np] = [hi|hi|lo],  [N,3Nxp] = [hi^T|lo^T|hi^T]
    unsigned short *xT, *yT;   // [D,3Nxp] = [hi^T|hi^T|lo^T],  [D,3Np] = [hi^T|lo^T|hi^T]
    size_t total;
};

static LossWs carve(void* base, int Nx, int N, int D) {
    LossWs w;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* q = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return q;
    };
    const int Np = pad64(N), Nxp = pad64(Nx);
    w.x3 = (unsigned short*)take((size_t)Nx * 3 * D * 2);
    w.y3 = (unsigned short*)take((size_t)pad16(N) * 3 * D * 2);   // rows N..N16-1 zero
    w.S = (float*)take((size_t)Nx * pad16(N) * 4);              // ld = N16, columns N.. ignored
    w.lse = (float*)take((size_t)Nx * 4);
    w.tsum = (float*)take((size_t)Nx * 4);
    w.rows = (float*)take((size_t)Nx * 4);
    w.G = (unsigned short*)take((size_t)Nx * 3 * Np * 2);
    w.GT = (unsigned short*)take((size_t)N * 3 * Nxp * 2);
    w.xT = (unsigned short*)take((size_t)D * 3 * Nxp * 2);
    w.yT = (unsigned short*)take((size_t)D * 3 * Np * 2);
    w.total = off;
    return w;
}

static int loss_check(const float* x, const float* y, const int64_t* labels, int Nx, int N, int D, int row0) {
    if (!x || !y || !labels) return set_error(CLIBD_EINVAL, "softce: null pointer");
    if (Nx <= 0 || N <= 0 || D <= 0) return set_error(CLIBD_EINVAL, "softce: non-positive shape");
    if (D % 64 != 0) return set_error(CLIBD_EINVAL, "softce: D must be a multiple of 64");
    if (row0 < 0 || row0 + Nx > N) return set_error(CLIBD_EINVAL, "softce: row block outside the global batch");
    return 0;
}

}  // namespace clibd

using namespace clibd;

extern "C" size_t clibd_softce_workspace_bytes(int Nx, int N, int D) {
    if (Nx <= 0 || N <= 0 || D <= 0) return 0;
    return carve(nullptr, Nx, N, D).total;
}

extern "C" int clibd_softce_rows_fwd(const float* x, const float* y, const int64_t* labels, int Nx, int N, int D, int row0,
                                     const float* scale, float* loss_sum, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = loss_check(x, y, labels, Nx, N, D, row0)) return e;
    if (!loss_sum || !workspace || !scale) return set_error(CLIBD_EINVAL, "softce_fwd: null pointer");
    if (!aligned16(workspace)) return set_error(CLIBD_EINVAL, "softce_fwd: workspace must be 16-byte aligned");
    const LossWs w = carve(workspace, Nx, N, D);
    if (workspace_bytes < w.total) return set_error(CLIBD_EINVAL, "softce_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(split3_kernel, dim3(1024), dim3(256), 0, st, x, Nx, D, 1, w.x3);
    hipLaunchKernelGGL(split3_kernel, dim3(1024), dim3(256), 0, st, y, N, D, 0, w.y3);
    if (int e = check_launch("softce split")) return e;
    const int N16 = pad16(N);
    if (N16 != N && hipMemsetAsync(w.y3 + (size_t)N * 3 * D, 0, (size_t)(N16 - N) * 3 * D * 2, st) != hipSuccess)
        return set_error(CLIBD_ELAUNCH, "softce_fwd: memset failed");
    clibd_gemm_epilogue ep = {};
    ep.out_f32 = w.S;  // raw similarities <x_i, y_j>
    ep.ld_out_f32 = N16;
    ep.split_k = 1;
    if (int e = clibd_gemm_bf16_nt(w.x3, 3 * D, w.y3, 3 * D, Nx, N16, 3 * D, &ep, stream)) return e;
    hipLaunchKernelGGL(softce_rows_fwd_kernel, dim3((Nx + 3) / 4), dim3(256), 0, st, w.S, N16, Nx, N, labels, row0, scale,
                       w.lse, w.tsum, w.rows);
    if (int e = check_launch("softce_rows_fwd")) return e;
    hipLaunchKernelGGL(reduce_rows_add_kernel, dim3(1), dim3(256), 0, st, w.rows, Nx, loss_sum);
    return check_launch("softce_rows_fwd (loss sum)");
}

// Must follow clibd_softce_rows_fwd on the same workspace (reuses its similarity matrix and row statistics).
extern "C" int clibd_softce_rows_bwd(const int64_t* labels, int Nx, int N, int D, int row0, const float* scale, float weight,
                                     const float* weight_scale, float* dx, float* dy, float* dscale, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!labels || !dx || !dy || !workspace || !scale) return set_error(CLIBD_EINVAL, "softce_bwd: null pointer");
    if (Nx <= 0 || N <= 0 || D <= 0 || D % 64 != 0 || row0 < 0 || row0 + Nx > N)
        return set_error(CLIBD_EINVAL, "softce_bwd: bad shape");
    if (!aligned16(workspace) || !aligned16(dx) || !aligned16(dy)) return set_error(CLIBD_EINVAL, "softce_bwd: alignment");
    const LossWs w = carve(workspace, Nx, N, D);
    if (workspace_bytes < w.total) return set_error(CLIBD_EINVAL, "softce_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Np = pad64(N), Nxp = pad64(Nx);   // multiples of 64: every K segment of the backward GEMMs is whole K-tiles
    hipLaunchKernelGGL(softce_rows_bwd_kernel, dim3((Nx + 3) / 4), dim3(256), 0, st, w.S, pad16(N), Nx, N, labels, row0, w.lse,
                       w.tsum, weight, weight_scale, scale, w.G, Np, w.rows);
    if (int e = check_launch("softce_rows_bwd")) return e;
    if (dscale != nullptr) {
        hipLaunchKernelGGL(reduce_rows_add_kernel, dim3(1), dim3(256), 0, st, w.rows, Nx, dscale);
        if (int e = check_launch("softce_rows_bwd (dscale sum)")) return e;
    }
    // operand images, zero padded along the contraction: segment order pairs hi.hi + hi.lo + lo.hi (see the file header)
    if (int e = launch_transpose3(w.G, Nx, Np, N, /*hi, lo, hi*/ 0 | (2 << 2) | (0 << 4), w.GT, Nxp, st)) return e;
    if (int e = launch_transpose3(w.x3, Nx, D, D, /*hi, hi, lo*/ 0 | (1 << 2) | (2 << 4), w.xT, Nxp, st)) return e;
    if (int e = launch_transpose3(w.y3, N, D, D, /*hi, lo, hi*/ 0 | (1 << 2) | (2 << 4), w.yT, Np, st)) return e;
    clibd_gemm_epilogue ep = {};
    ep.split_k = 1;
    // dx[i,:] += sum_j G[i,j] y[j,:]      (accumulate in place through the residual path)
    ep.out_f32 = dx; ep.ld_out_f32 = D; ep.residual_f32 = dx; ep.ld_res = D;
    if (int e = clibd_gemm_bf16_nt(w.G, 3 * Np, w.yT, 3 * Np, Nx, D, 3 * Np, &ep, stream)) return e;
    // dy[j,:] += sum_i G[i,j] x[i,:]
    ep.out_f32 = dy; ep.residual_f32 = dy;
    if (int e = clibd_gemm_bf16_nt(w.GT, 3 * Nxp, w.xT, 3 * Nxp, N, D, 3 * Nxp, &ep, stream)) return e;
    return CLIBD_OK;
}
