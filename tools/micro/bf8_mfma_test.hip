// Which modifier of v_mfma_f32_16x16x128_f8f6f4 selects the format of which source?  (round 5: e5m2 gradients x e4m3 weights for the
// 8-bit dgrad.)  Every byte of src0 is 0x38 (e4m3 1.0; as e5m2 it reads 0.5), every byte of src1 is 0x3C (e5m2 1.0; as e4m3 it reads 1.5):
// K = 128 products give 128 when src0 is taken as e4m3 and src1 as e5m2, 192 with both as e4m3, 64 with both as e5m2, 96 when swapped.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/bf8_mfma_test.hip -o build_ab/bf8_mfma_test && build_ab/bf8_mfma_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float* out) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x3C3C3C3C; }
    f32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    if (MODE == 1) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 blgp:1\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    if (MODE == 2) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:1\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    if (MODE == 3) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:1 blgp:1\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    if (threadIdx.x == 0) out[MODE] = acc[0];
}

int main() {
    float* d;
    hipMalloc(&d, 16);
    k<0><<<1, 64>>>(d); k<1><<<1, 64>>>(d); k<2><<<1, 64>>>(d); k<3><<<1, 64>>>(d);
    float h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("no modifier %.0f | blgp:1 %.0f | cbsz:1 %.0f | cbsz:1 blgp:1 %.0f   (src0 e4m3 x src1 e5m2 = 128)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
