// Semantics probe for v_permlane32_swap / v_permlane16_swap / DPP controls on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/micro/permlane_test.hip -o /tmp/permlane_test && /tmp/permlane_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 1000 + lane, b = 2000 + lane;
    const u32x2_t r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const u32x2_t r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[0 * 64 + lane] = r32[0]; out[1 * 64 + lane] = r32[1];
    out[2 * 64 + lane] = r16[0]; out[3 * 64 + lane] = r16[1];
    out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(0, (int)lane, 0x128, 0xF, 0xF, false);   // row_ror:8
    out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(0, (int)lane, 0x141, 0xF, 0xF, false);   // row_half_mirror
    out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(0, (int)lane, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    out[7 * 64 + lane] = __builtin_amdgcn_update_dpp(0, (int)lane, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    out[8 * 64 + lane] = __builtin_amdgcn_update_dpp(0, (int)lane, 0x140, 0xF, 0xF, false);   // row_mirror
}
int main() {
    unsigned* d; hipMalloc(&d, 9 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"permlane32_swap(a,b)[0]", "permlane32_swap(a,b)[1]", "permlane16_swap(a,b)[0]", "permlane16_swap(a,b)[1]", "dpp row_ror:8", "dpp row_half_mirror", "dpp quad[1,0,3,2]", "dpp quad[2,3,0,1]", "dpp row_mirror"};
    for (int k = 0; k < 9; ++k) { printf("%-26s:", names[k]); for (int l = 0; l < 64; ++l) printf(" %u", h[k * 64 + l]); printf("\n"); }
    return 0;
}
