// Micro-benchmark: how many bytes per clock can one CU pull from L2 (panels shared by many CUs, as the GEMM's A / W panels are)
//   mode 0: LDS-DMA      (global_load_lds_dwordx4: 1 KiB per wave instruction, straight into LDS — what gemm256 uses)
//   mode 1: plain loads  (global_load_dwordx4 into VGPRs, 8 in flight per lane, results xor-ed so nothing is optimised away)
//   mode 2: plain loads + ds_write_b128 of every loaded vector into LDS (the register-staged pipeline of a 512-register GEMM)
//   mode 3: 16-byte STORES, 1 KiB contiguous per wave instruction, to distinct addresses of a 4-GiB buffer (HBM)
//   mode 4: 16-byte STORES in gemm256's epilogue shape: 16 lanes x 16 B = 256 contiguous bytes, the four 16-lane rows of a wave
//           6144 B apart (rows of an [M, 3072] bf16 output), 16 instructions walk 64 rows
// One workgroup of 8 waves per CU; every workgroup re-reads the same 8 MiB (L2 / Infinity-Cache resident) buffer.
//   hipcc --offload-arch=gfx950 -O3 -o load_paths load_paths.hip && ./load_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t bytes, int iters, unsigned* sink, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_iter = 512 * 16 * 8;   // bytes per workgroup and iteration (8 instructions x 8 waves x 1 KiB)
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    size_t off = ((size_t)blockIdx.x * 65536) % bytes;
    for (int it = 0; it < iters; ++it) {
        const char* p = src + off + (size_t)threadIdx.x * 16;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(p + j * 8192 - (size_t)lane * 16 + (size_t)lane * 16), (lds_void*)(lds + (wave * 8 + j) * 1024), 16, 0, 0);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 3 || MODE == 4) {
            char* q = const_cast<char*>(src) + ((size_t)blockIdx.x * iters + it) * (MODE == 3 ? per_iter : (size_t)64 * 6144 * 8 / 8) % (bytes - (MODE == 3 ? per_iter : (size_t)512 * 6144));
            const u32x4 v = {(unsigned)it, (unsigned)lane, 3u, 4u};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 3) *(u32x4*)(q + (size_t)threadIdx.x * 16 + j * 8192) = v;
                else *(u32x4*)(q + (size_t)(64 * wave + 4 * j + (lane >> 4) + 32 * (it & 1)) * 6144 + 16 * (lane & 15) + 256 * ((it >> 1) % 24)) = v;
            }
        } else {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(p + j * 8192);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 2) *(u32x4*)(lds + ((wave * 8 + j) * 64 + lane) * 16) = v[j];
                else acc ^= v[j];
            }
        }
        off += per_iter;
        if (off + per_iter > bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 0 || MODE == 2) acc = *(const u32x4*)(lds + threadIdx.x * 16);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[1] == 7u) sink[0] = acc[2] ^ acc[3];
}

int main(int argc, char** argv) {
    size_t bytes = (size_t)8 << 20;
    int iters = 2000;
    char* src; unsigned* sink; long long* cyc;
    hipMalloc(&src, ((size_t)4 << 30) + (1 << 20)); hipMemset(src, 1, ((size_t)4 << 30) + (1 << 20));
    hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * sizeof(long long));
    for (int grid : {1, 32, 256}) {
        for (int mode = 0; mode < 5; ++mode) {
            bytes = mode >= 3 ? (size_t)4 << 30 : (size_t)8 << 20;
            iters = mode >= 3 ? 400 : 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 65536, 0, src, bytes, iters, sink, cyc);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 65536, 0, src, bytes, iters, sink, cyc);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 65536, 0, src, bytes, iters, sink, cyc);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 65536, 0, src, bytes, iters, sink, cyc);
                else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 65536, 0, src, bytes, iters, sink, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c[256]; hipMemcpy(c, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)c[i]; avg /= grid;
            const double per_wg = (double)iters * 512 * 16 * 8;
            printf("grid %3d mode %d (%s): %8.3f ms, %7.0f cycles per workgroup, %6.1f B/clk/CU, %6.2f TB/s aggregate\n", grid, mode,
                   mode == 0 ? "LDS-DMA" : mode == 1 ? "plain loads" : mode == 2 ? "plain loads + ds_write" : mode == 3 ? "stores, contiguous" : "stores, gemm256 shape", ms, avg, per_wg / avg, per_wg * grid / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
