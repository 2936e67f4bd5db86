// Micro-benchmark: what the board delivers at its 1400 W power cap (run beside tools/power_sampler.py; tools/power_report.py joins them)
//   mode 0: bare MFMA    v_mfma_f32_16x16x32_bf16 on register operands (two alternating operand sets of random bf16), 2 waves / SIMD,
//                        128 accumulator registers per wave - the matrix pipe with nothing else running
//   mode 1: MFMA + LDS   the same, operands re-read from LDS in gemm256's ratio (12 ds_read_b128 per 32 MFMAs)
//   mode 2: copy         16-byte loads + 16-byte stores over 2 x 4 GiB (HBM both ways)
//   mode 3: read         16-byte loads over 4 GiB
//   mode 4: write        16-byte stores over 4 GiB
//   mode 5: MFMA + copy  waves 0-3 of every workgroup run mode 0, waves 4-7 stream a copy: both at once under one power cap
//   mode 6: bare MFMA    v_mfma_f32_32x32x16_bf16 (same flops per cycle, half the operand-register reads per flop), otherwise as mode 0
// Every mode runs for <seconds> of wall clock and prints its rate and its wall-clock window.
//   hipcc --offload-arch=gfx950 -O3 -o power_roof power_roof.hip && ./power_roof [seconds] [mode ...]
#include <hip/hip_runtime.h>
#include <sys/time.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// eight random bf16 in (-1, 1): sign and mantissa random, exponent 120..126
__device__ __forceinline__ bf16x8 rand_frag(unsigned seed) {
    u32x4 w;
    for (int i = 0; i < 4; ++i) {
        const unsigned h = hash32(seed * 4u + i);
        const unsigned lo = (h & 0x807fu) | ((120u + ((h >> 8) % 7u)) << 7);
        const unsigned hi = ((h >> 16) & 0x807fu) | ((120u + ((h >> 24) % 7u)) << 7);
        w[i] = lo | (hi << 16);
    }
    return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ void mfma_block(f32x4 (&acc)[8][4], const bf16x8 (&P)[8], const bf16x8 (&Q)[4]) {
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Q[q], P[p], acc[p][q], 0, 0, 0);
}

__device__ __forceinline__ void stream_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, size_t tid, size_t nthreads, int passes,
                                            bool do_read, bool do_write, u32x4& sinkv) {
    for (int ps = 0; ps < passes; ++ps)
        for (size_t i = tid; i + 3 * nthreads < n16; i += 4 * nthreads) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = do_read ? src[i + j * nthreads] : (u32x4){(unsigned)i, (unsigned)j, 3u, (unsigned)ps};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (do_write) dst[i + j * nthreads] = v[j];
                else sinkv ^= v[j];
            }
        }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k32(int iters, unsigned* sink) {
    f32x16 acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][q][e] = 0.f;
    bf16x8 P0[4], Q0[2], P1[4], Q1[2];
    const unsigned s = (blockIdx.x * 512u + threadIdx.x) * 64u;
#pragma unroll
    for (int p = 0; p < 4; ++p) { P0[p] = rand_frag(s + p); P1[p] = rand_frag(s + 16 + p); }
#pragma unroll
    for (int q = 0; q < 2; ++q) { Q0[q] = rand_frag(s + 32 + q); Q1[q] = rand_frag(s + 40 + q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)   // two k16 steps per operand set: the same flops per iteration as mode 0
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Q0[q], P0[p], acc[p][q], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Q1[q], P1[p], acc[p][q], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[p][q][e];
    if (t == 1234.5f) sink[1] = 1;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    __shared__ int copy_done;
    if (MODE == 5) {
        if (threadIdx.x == 0) copy_done = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 sinkv = {0, 0, 0, 0};
    if (MODE == 0 || MODE == 1 || (MODE == 5 && wave < 4)) {
        f32x4 acc[8][4];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) {
            for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 512) *(bf16x8*)(lds + i * 16) = rand_frag(blockIdx.x * 8192u + i);
            __syncthreads();
            const char* base = lds + lane * 16;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 P[8], Q[4];
                    const int o = ((2 * it + h) & 3) * 12288 + (wave & 3) * 1024;
#pragma unroll
                    for (int p = 0; p < 8; ++p) P[p] = *(const bf16x8*)(base + o + p * 1024);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Q[q] = *(const bf16x8*)(base + o + 8192 + q * 1024);
                    mfma_block(acc, P, Q);
                }
            }
        } else {
            bf16x8 P0[8], Q0[4], P1[8], Q1[4];
            const unsigned s = (blockIdx.x * 512u + threadIdx.x) * 64u;
#pragma unroll
            for (int p = 0; p < 8; ++p) { P0[p] = rand_frag(s + p); P1[p] = rand_frag(s + 16 + p); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { Q0[q] = rand_frag(s + 32 + q); Q1[q] = rand_frag(s + 40 + q); }
            if (MODE == 5) {  // run until this workgroup's copy waves are through; count the iterations for the host
                unsigned long long n = 0;
                while (__builtin_amdgcn_readfirstlane(*(volatile int*)&copy_done) < 4) {
                    for (int j = 0; j < 16; ++j) {
                        mfma_block(acc, P0, Q0);
                        mfma_block(acc, P1, Q1);
                    }
                    n += 16;
                }
                if (lane == 0) atomicAdd((unsigned long long*)(sink + 2), n);
            } else {
                for (int it = 0; it < iters; ++it) {
                    mfma_block(acc, P0, Q0);
                    mfma_block(acc, P1, Q1);
                }
            }
        }
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) t += acc[p][q][0] + acc[p][q][1] + acc[p][q][2] + acc[p][q][3];
        if (t == 1234.5f) sink[1] = 1;
    } else {
        const bool rd = MODE != 4, wr = MODE != 3;
        if (MODE == 5) {
            const size_t nthreads = (size_t)gridDim.x * 256;
            const size_t tid = (size_t)blockIdx.x * 256 + (threadIdx.x - 256);
            stream_copy(src, dst, n16, tid, nthreads, iters, true, true, sinkv);   // iters = passes over the buffer
            __threadfence_block();
            if (lane == 0) atomicAdd(&copy_done, 1);
        } else {
            const size_t nthreads = (size_t)gridDim.x * 512;
            const size_t tid = (size_t)blockIdx.x * 512 + threadIdx.x;
            stream_copy(src, dst, n16, tid, nthreads, iters, rd, wr, sinkv);
        }
    }
    if (sinkv[0] == 0x12345678u && sinkv[1] == 7u) sink[0] = sinkv[2] ^ sinkv[3];
}

static double now() {
    timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    std::vector<int> modes;
    for (int i = 2; i < argc; ++i) modes.push_back(atoi(argv[i]));
    if (modes.empty()) modes = {0, 6, 1, 2, 3, 4, 5};
    const size_t bytes = (size_t)4 << 30, n16 = bytes / 16;
    u32x4 *src, *dst;
    unsigned* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 0x5a, bytes)); CK(hipMemset(dst, 0, bytes)); CK(hipMemset(sink, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"mfma", "mfma+lds", "copy", "read", "write", "mfma+copy", "mfma32x32"};
    for (int mode : modes) {
        // per-launch work: ~10 ms
        const int grid = (mode <= 1 || mode >= 5) ? 256 : 2048;
        int iters = (mode <= 1 || mode == 6) ? 20000 : 1;
        auto launch = [&](int it) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, src, dst, n16, it, sink); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, src, dst, n16, it, sink); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, src, dst, n16, it, sink); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, src, dst, n16, it, sink); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, src, dst, n16, it, sink); break;
                case 6: hipLaunchKernelGGL(k32, dim3(grid), dim3(512), 0, 0, it, sink); break;
                default: break;
            }
        };
        double flops_per_launch = 0, bytes_per_launch = 0;
        if (mode <= 1 || mode == 6) flops_per_launch = (double)grid * 8 * iters * 64 * 16384.0;
        if (mode == 2) bytes_per_launch = 2.0 * bytes;
        if (mode == 3 || mode == 4) bytes_per_launch = (double)bytes;
        if (mode == 5) bytes_per_launch = 2.0 * bytes;  // (flops: the MFMA waves count their iterations into sink[2..3])
        for (int w = 0; w < 3; ++w) {
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, src, dst, n16, 1, sink);
            else launch(iters);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemset(sink, 0, 64));
        const double t0 = now();
        int launches = 0;
        CK(hipEventRecord(e0, 0));
        while (now() - t0 < secs) {
            for (int j = 0; j < 4; ++j) {
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, src, dst, n16, 1, sink);
                else launch(iters);
            }
            launches += 4;
            CK(hipDeviceSynchronize());
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        const double t1 = now();
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double s = ms * 1e-3;
        if (mode == 5) {
            unsigned long long n = 0;
            CK(hipMemcpy(&n, sink + 2, 8, hipMemcpyDeviceToHost));
            flops_per_launch = (double)n * 64 * 16384.0 / launches;
        }
        printf("%-10s %8.3f ms/launch  %8.1f TFLOP/s  %7.2f TB/s  window %.3f %.3f\n", names[mode], ms / launches, flops_per_launch * launches / s / 1e12,
               bytes_per_launch * launches / s / 1e12, t0, t1);
        fflush(stdout);
    }
    return 0;
}
