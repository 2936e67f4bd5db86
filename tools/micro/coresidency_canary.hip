// Co-residency reproducer for DESIGN.md §3.3: what does a kernel of ANOTHER stream do to a workgroup that shares
// its CU with libclibd_hip.so's attention forward kernels (whose K / V images are written by LDS-DMA through M0-based addresses)?
//
//   canary      : every workgroup fills its 24-KiB dynamic LDS allocation (the size of LayerNorm's adapter matrix) with a pattern,
//                 then re-reads and re-checks it `iters` times; any word that changed is counted and the first one recorded.
//                 Nothing but this workgroup may ever write that memory: a hit means a foreign write into its LDS.
//   ln_t<BPERM> : the LayerNorm kernel's LoRA down-projection in isolation — t[row, 0:8] = x[row, :] . A^T with A [8, H] in LDS
//                 (fp32), one wave per row, per-lane partial sums, then the 8-way butterfly across the wave either on
//                 __shfl_xor = ds_bpermute_b32 (BPERM = true: the round-1 kernel that returned wrong sums beside attention_fwd) or
//                 on DPP / v_permlane swaps (BPERM = false: what ships).  Checked against a host fp64 reference.
//
// Each is run alone and then beside back-to-back clibd_attention_fwd launches on a second stream (per-head kernel: B x heads <
// 2 x CUs; persistent kernel: more), called through the C ABI of the library given on the command line.
//
//   hipcc --offload-arch=gfx950 -O3 -o coresidency_canary coresidency_canary.hip -ldl
//   ./coresidency_canary /path/to/libclibd_hip.so [rounds]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int H = 768;
constexpr int LDS_WORDS = 8 * H;   // 24 KiB, as layernorm_fwd_kernel<3, true>

__device__ __forceinline__ unsigned pattern(unsigned block, unsigned idx) {
    unsigned x = block * 0x9E3779B1u + idx * 0x85EBCA6Bu + 0x12345u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}

__global__ __launch_bounds__(256) void canary(int iters, unsigned long long* errors, unsigned* first_bad) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < LDS_WORDS; i += 256) lds[i] = pattern(blockIdx.x, i);
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < LDS_WORDS; i += 256) {
            const unsigned v = lds[i];
            if (v != pattern(blockIdx.x, i)) {
                if (bad == 0 && atomicAdd(errors, 1ull) == 0) { first_bad[0] = blockIdx.x; first_bad[1] = i; first_bad[2] = v; first_bad[3] = pattern(blockIdx.x, i); }
                ++bad;
                lds[i] = pattern(blockIdx.x, i);   // repair, so that a second hit on the same word is seen as a second event
            }
        }
        __builtin_amdgcn_s_sleep(32);
        __syncthreads();
    }
    if (bad > 1) atomicAdd(errors, (unsigned long long)(bad - 1));
}

// ---- cross-lane pieces
__device__ __forceinline__ float reduce8_bpermute(float v[8], int lane) {   // the round-1 butterfly (10 ds_bpermute_b32)
    float w4[4], w2[2], w1;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float keep = b5 ? v[i + 4] : v[i], send = b5 ? v[i] : v[i + 4]; w4[i] = keep + __shfl_xor(send, 32, 64); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const float keep = b4 ? w4[i + 2] : w4[i], send = b4 ? w4[i] : w4[i + 2]; w2[i] = keep + __shfl_xor(send, 16, 64); }
    { const float keep = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1]; w1 = keep + __shfl_xor(send, 8, 64); }
    w1 += __shfl_xor(w1, 4, 64);
    w1 += __shfl_xor(w1, 2, 64);
    w1 += __shfl_xor(w1, 1, 64);
    return w1;  // value index 4*b5 + 2*b4 + b3
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
// clibd_amd/csrc/common.h wave_sum: DPP inside the 16-lane rows, v_permlane swaps across them — no LDS-crossbar instruction
__device__ __forceinline__ float wave_sum_nolds(float v) {
    v += dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);      // row_half_mirror
    v += dpp<0x140>(v);      // row_mirror
    float a = v, b = v;
    swap16(a, b);
    v = a + b;
    a = v; b = v;
    swap32(a, b);
    return a + b;
}

template <bool BPERM>
__global__ __launch_bounds__(256) void ln_t(const float* __restrict__ x, const float* __restrict__ A, int M, float* __restrict__ t) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[];   // [8][H]
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * H; i += 256) a_lds[i] = A[i];
    __syncthreads();
    const int nwaves = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += nwaves) {
        float tp[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tp[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = 4 * (lane + 64 * j);
            const float4 y = *(const float4*)(x + (size_t)row * H + c);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 a = *(const float4*)(a_lds + r * H + c);
                tp[r] += (y.x * a.x + y.y * a.y) + (y.z * a.z + y.w * a.w);
            }
        }
        if (BPERM) {
            const float tv = reduce8_bpermute(tp, lane);
            if ((lane & 7) == 0) t[(size_t)row * 8 + (lane >> 3)] = tv;
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float s = wave_sum_nolds(tp[r]);
                if (lane == 0) t[(size_t)row * 8 + r] = s;
            }
        }
    }
}

typedef int (*attn_fwd_fn)(const void*, int, int, int, const int*, void*, int, int, void*);

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s libclibd_hip.so [rounds]\n", argv[0]); return 1; }
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    attn_fwd_fn attn = (attn_fwd_fn)dlsym(lib, "clibd_attention_fwd");
    if (!attn) { printf("clibd_attention_fwd not found\n"); return 1; }
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    CHECK(hipFuncSetAttribute((const void*)canary, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WORDS * 4));

    // attention operands: S = 197, 12 heads; B = 32 -> 384 heads (per-head kernel), B = 256 -> 3072 heads (persistent kernel)
    const int S = 197, NH = 12, HA = 64 * NH, BMAX = 256;
    unsigned short* qkv; unsigned short* aout;
    CHECK(hipMalloc(&qkv, (size_t)BMAX * S * 3 * HA * 2));
    CHECK(hipMalloc(&aout, (size_t)BMAX * S * HA * 2));
    {
        std::vector<unsigned short> h((size_t)BMAX * S * 3 * HA);
        unsigned seed = 7;
        for (auto& v : h) { float f = frand(seed) * 1.5f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
        CHECK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    // LayerNorm-like operands
    const int M = 50432;
    std::vector<float> hx((size_t)M * H), hA(8 * H);
    unsigned seed = 11;
    for (auto& v : hx) v = frand(seed) * 4.f;
    for (auto& v : hA) v = frand(seed) * 0.1f;
    std::vector<double> tref((size_t)M * 8);
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < 8; ++r) { double acc = 0; for (int c = 0; c < H; ++c) acc += (double)hx[(size_t)m * H + c] * hA[r * H + c]; tref[(size_t)m * 8 + r] = acc; }
    float *dx, *dA, *dt;
    CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dt, (size_t)M * 8 * 4));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    unsigned long long* derr; unsigned* dbad;
    CHECK(hipMalloc(&derr, 8)); CHECK(hipMalloc(&dbad, 16));
    std::vector<float> ht((size_t)M * 8);

    auto bad_rows = [&]() {
        CHECK(hipMemcpy(ht.data(), dt, ht.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int m = 0; m < M; ++m) {
            bool b = false;
            for (int r = 0; r < 8; ++r) b = b || std::fabs((double)ht[(size_t)m * 8 + r] - tref[(size_t)m * 8 + r]) > 1e-3 * (1.0 + std::fabs(tref[(size_t)m * 8 + r]));
            bad += b;
        }
        return bad;
    };
    for (int mode = 0; mode < 3; ++mode) {   // 0: alone, 1: beside the per-head attention kernel, 2: beside the persistent one
        const int B = mode == 1 ? 32 : 256;
        const char* tag = mode == 0 ? "alone" : mode == 1 ? "beside attention_fwd (per-head kernel, 384 heads)" : "beside attention_fwd (persistent kernel, 3072 heads)";
        const int per_round = mode == 1 ? 6 : 1;
        // ---- canary
        CHECK(hipMemset(derr, 0, 8)); CHECK(hipMemset(dbad, 0, 16));
        for (int r = 0; r < rounds; ++r) {
            if (mode) for (int k = 0; k < per_round; ++k) { if (attn(qkv, B, S, NH, nullptr, aout, S, S, (void*)s2) != 0) { printf("attention launch failed\n"); return 1; } }
            hipLaunchKernelGGL(canary, dim3(2048), dim3(256), LDS_WORDS * 4, s1, 24, derr, dbad);
        }
        CHECK(hipDeviceSynchronize());
        unsigned long long herr; unsigned hbad[4];
        CHECK(hipMemcpy(&herr, derr, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hbad, dbad, 16, hipMemcpyDeviceToHost));
        printf("canary      %-58s: %llu changed LDS words in %d launches", tag, herr, rounds);
        if (herr) printf("  (first: block %u word %u read %08x expected %08x)", hbad[0], hbad[1], hbad[2], hbad[3]);
        printf("\n");
        // ---- LoRA down-projection, both reductions
        for (int bperm = 1; bperm >= 0; --bperm) {
            int total_bad = 0, launches_bad = 0;
            for (int r = 0; r < rounds; ++r) {
                CHECK(hipMemsetAsync(dt, 0, (size_t)M * 8 * 4, s1));
                if (mode) for (int k = 0; k < per_round; ++k) attn(qkv, B, S, NH, nullptr, aout, S, S, (void*)s2);
                if (bperm) hipLaunchKernelGGL(ln_t<true>, dim3(2048), dim3(256), LDS_WORDS * 4, s1, dx, dA, M, dt);
                else hipLaunchKernelGGL(ln_t<false>, dim3(2048), dim3(256), LDS_WORDS * 4, s1, dx, dA, M, dt);
                CHECK(hipDeviceSynchronize());
                const int b = bad_rows();
                total_bad += b;
                launches_bad += b > 0;
            }
            printf("ln_t %-7s%-58s: %d wrong rows in %d of %d launches\n", bperm ? "bperm" : "dpp", tag, total_bad, launches_bad, rounds);
        }
    }
    return 0;
}
