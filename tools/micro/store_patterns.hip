// Micro-benchmark: how fast can ONE CU (and the whole chip) retire the epilogue stores of a 256x256 bf16 tile, as a
// function of the lane -> address pattern of each 16-B store instruction?   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int P>
__global__ __launch_bounds__(512) void k(unsigned short* out, int M, int N, int tiles_m, int tiles_n, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles_m * tiles_n;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tm = tile % tiles_m, tn = tile / tiles_m;
        const int m0 = tm * 256 + 128 * wm, n0 = tn * 256 + 64 * wn;
        uint4 v = make_uint4(tile, lane, wave, 7);
        if (P == 4) {  // dwordx2, 16 lanes x 8 B = one 128-B row segment, 4 rows per instruction, 32 instructions
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int m = m0 + 4 * j + (lane >> 4);
                if (m < M) *(uint2*)((char*)(out + (size_t)m * N + n0) + 8 * (lane & 15)) = make_uint2(v.x, v.y);
            }
            continue;
        }
        if (P == 6 || P == 7) {  // wave covers 64 rows x 128 cols (wp = wave>>2, wq = wave&3); lane (c = lane&15, g = lane>>4)
            const int mm0 = tm * 256 + 64 * (wave & 3), nn0 = tn * 256 + 128 * (wave >> 2);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = mm0 + 4 * j + (lane >> 4);
                if (P == 7) {  // bf16: 16 lanes x 16 B = 256 contiguous bytes per row
                    if (m < M) *(uint4*)((char*)(out + (size_t)m * N + nn0) + 16 * (lane & 15)) = v;
                } else {       // fp32-like (same byte count: half the columns): 16-B pieces at 32-B stride, 2 instructions fill a row
                    if (m < M) *(uint4*)((char*)(out + (size_t)m * N + nn0) + 32 * (lane & 15) + 16 * (j & 1)) = v;
                }
            }
            continue;
        }
        if (P == 5) {  // dwordx2 in the row-per-lane shape: row = lane&15 ... (16 rows x 4 x 8 B), 32 instructions
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int m = m0 + 16 * (j >> 2) + (lane & 15);
                if (m < M) *(uint2*)((char*)(out + (size_t)m * N + n0) + 32 * (lane >> 4) + 8 * (j & 3)) = make_uint2(v.x, v.y);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int row, off;
                if (P == 0) { row = 16 * j + (lane & 15); off = 32 * (lane >> 4) + 16 * h; }           // current epilogue
                else if (P == 1) { row = 16 * j + (lane & 15); off = 64 * h + 16 * (lane >> 4); }       // 64-B sector per row
                else if (P == 2) { row = 16 * j + 8 * h + (lane >> 3); off = 16 * (lane & 7); }         // 128-B line per row
                else { row = 16 * j + (lane >> 2); off = 64 * h + 16 * (lane & 3); }                   // 64-B, row = lane>>2
                const int m = m0 + row;
                if (m < M) *(uint4*)((char*)(out + (size_t)m * N + n0) + off) = v;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int P>
void run(unsigned short* out, int M, int N, int grid, long long* cyc) {
    const int tm = (M + 255) / 256, tn = N / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, out, M, N, tm, tn, cyc);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, out, M, N, tm, tn, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (auto c : h) mx = c > mx ? c : mx;
    const double bytes = (double)M * N * 2;
    const double tiles_per_wg = (double)tm * tn / grid;
    printf("P%d grid %3d: %8.1f us  %6.2f TB/s  max %lld cyc/wg  -> %.0f cyc per tile (128 KB) = %.1f B/clk/CU\n", P, grid, ms * 1e3,
           bytes / (ms * 1e-3) / 1e12, mx, mx / tiles_per_wg, 131072.0 / (mx / tiles_per_wg));
}

int main() {
    const int M = 50432, N = 3072;
    unsigned short* out; long long* cyc;
    hipMalloc(&out, (size_t)M * N * 2);
    hipMalloc(&cyc, 256 * 8);
    for (int grid : {256, 8}) {
        run<0>(out, M, N, grid, cyc);
        run<1>(out, M, N, grid, cyc);
        run<2>(out, M, N, grid, cyc);
        run<3>(out, M, N, grid, cyc);
        run<4>(out, M, N, grid, cyc);
        run<5>(out, M, N, grid, cyc);
        run<6>(out, M, N, grid, cyc);
        run<7>(out, M, N, grid, cyc);
    }
    return 0;
}
