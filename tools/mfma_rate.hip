// Cycles per MFMA as a single wave per SIMD issues them (s_memtime around a loop), for the instruction forms and register
// patterns the kernels of this library use.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed) {
    const int tid = threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + tid * 0.001f + i; b[i] = seed * 0.5f + tid * 0.002f - i; }
    u32x4 ha, hb;
    for (int i = 0; i < 4; ++i) { ha[i] = 0x3F803F80u | (tid * 2654435761u >> (i + 3) & 0x007F007F); hb[i] = 0x3F803F80u | (tid * 40503u >> (i + 1) & 0x007F007F); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 big[2] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // f32 16x16x4, 4 accumulators round robin (conv_small: 2 x 2 blocks), 16 per iteration
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(q >> 1) * 4 + e], b[(q & 1) * 4 + e], acc[q], 0, 0, 0);
        } else if constexpr (MODE == 1) {   // f32 16x16x4, 16 accumulators
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 7], b[(q >> 1) & 7], acc[q], 0, 0, 0);
        } else if constexpr (MODE == 2) {   // f32 16x16x4, one accumulator (dependent chain)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 7], b[(q >> 1) & 7], acc[0], 0, 0, 0);
        } else if constexpr (MODE == 3) {   // f32 32x32x2, two accumulators, 16 per iteration (same MACs per instruction pair... 2048 MACs each)
#pragma unroll
            for (int q = 0; q < 16; ++q) big[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 7], b[(q >> 1) & 7], big[q & 1], 0, 0, 0);
        } else if constexpr (MODE == 4) {   // bf16 16x16x32, 16 accumulators
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ha), __builtin_bit_cast(bf16x8, hb), acc[q], 0, 0, 0);
        } else if constexpr (MODE == 5) {   // bf16 16x16x32, 4 accumulators
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ha), __builtin_bit_cast(bf16x8, hb), acc[q & 3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
    if ((tid & 63) == 0 || s == 1.234e-30f) out[(blockIdx.x * 4 + (tid >> 6))] = t1 - t0;
}

template <int MODE> void run(const char* name, int wgs) {
    unsigned long long* d;
    hipMalloc(&d, wgs * 4 * 8);
    const int iters = 2000;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f + r);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 4);
    hipMemcpy(h.data(), d, wgs * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %3d WGs: %.1f cycles per MFMA (median wave)\n", name, wgs, (double)h[h.size() / 2] / (iters * 16.0));
    hipFree(d);
}

int main() {
    for (int wgs : {1, 256}) {
        run<0>("v_mfma_f32_16x16x4_f32, 4 accumulators round robin", wgs);
        run<1>("v_mfma_f32_16x16x4_f32, 16 accumulators", wgs);
        run<2>("v_mfma_f32_16x16x4_f32, 1 accumulator (dependent)", wgs);
        run<3>("v_mfma_f32_32x32x2_f32, 2 accumulators", wgs);
        run<4>("v_mfma_f32_16x16x32_bf16, 16 accumulators", wgs);
        run<5>("v_mfma_f32_16x16x32_bf16, 4 accumulators", wgs);
    }
    return 0;
}
