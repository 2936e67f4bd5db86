// What does a lane's scale operand of v_mfma_scale_f32_16x16x128_f8f6f4 scale?  (round 5, for DESIGN §7's "what is left of configs[4]": an 8-bit QKV dgrad
// needs one scale per (token row, head) = per 64 consecutive k — the hardware's E8M0 block scales.)  Every operand byte is e4m3 1.0; scales are E8M0 bytes
// (127 = 2^0).  Cases: all 127; lanes 0-15 of src0 at 128 (x 2); lane 0 of src0 at 129 (x 4); lane 17 of src1 at 130 (x 8).  Prints the 16 x 16 result.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_scale_test.hip -o build_ab/mfma_scale_test && build_ab/mfma_scale_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float* out, int mode) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    int sa = 127, sb = 127;
    if (mode == 1 && lane < 16) sa = 128;
    if (mode == 2 && lane == 0) sa = 129;
    if (mode == 3 && lane == 17) sb = 130;
    if (mode == 4) sa = 127 | (130 << 8);       // byte 1 set, byte 0 = 1.0: is byte 0 the one that counts?
    f32x4 acc = {0, 0, 0, 0};
    asm volatile("s_nop 15\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    for (int r = 0; r < 4; ++r) out[mode * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

int main() {
    float* d;
    (void)hipMalloc(&d, 5 * 256 * sizeof(float));
    for (int m = 0; m < 5; ++m) k<<<1, 64>>>(d, m);
    static float h[5 * 256];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* what[5] = {"all scales 2^0", "src0 lanes 0-15 (k-group 0, every row) x 2", "src0 lane 0 (row 0, k-group 0) x 4", "src1 lane 17 (row 1, k-group 1) x 8",
                           "src0 scale register = 127 | 130 << 8 (is it byte 0?)"};
    for (int m = 0; m < 5; ++m) {
        printf("== %s\n", what[m]);
        for (int i = 0; i < 3; ++i) {
            printf("  out row %d:", i);
            for (int j = 0; j < 4; ++j) printf(" %6.0f", h[m * 256 + i * 16 + j]);
            printf(" ...\n");
        }
    }
    return 0;
}
