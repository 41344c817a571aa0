// How fast can a kernel READ a buffer that is not in the Infinity Cache, as a function of the bytes it keeps in flight per CU?
// (round-5 brief, item 2: tools/mall_probe.hip issues ONE 16-byte load per thread per loop trip and was quoted as "the part gives a
// reader 3.9-4.7 TB/s"; MI355X_MICROARCH.md reports 6.0-6.1 TB/s for an in-order sweep with 72 KiB in flight per CU.)
//   hipcc --offload-arch=gfx950 -O3 tools/read_ceiling.hip -o /tmp/read_ceiling && /tmp/read_ceiling
// A 2 GiB buffer (8 x the Infinity Cache) is swept once per launch by a persistent grid: W workgroups of 256 threads per CU, every
// thread keeps U independent 16-byte loads in flight per loop trip (registers; LDSDMA = 1: the same bytes by global_load_lds_dwordx4
// into LDS).  Bytes in flight per CU = W x 256 x 16 x U.  Three address orders: (a) each WORKGROUP sweeps its own contiguous 1/grid of
// the buffer (the static-stage kernels' pattern), (b) workgroups interleave at 4 KiB x U granularity (the whole grid reads one moving
// window), (c) as (a) with an XCD-contiguous mapping (workgroup b of XCD x reads the x-th eighth).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U, int ORDER>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    const size_t nblk = gridDim.x;
    size_t blk = blockIdx.x;
    if (ORDER == 2) {                                   // XCD-contiguous: workgroups of one XCD (blockIdx % 8) own a contiguous eighth
        const size_t per_xcd = nblk / 8;
        blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    }
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (ORDER == 1) {
        const size_t stride = nblk * 256 * U;
        for (size_t i = blk * 256 * U + threadIdx.x; i + 256 * (U - 1) < n16; i += stride) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[i + 256 * u];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    } else {
        const size_t per = n16 / nblk, lo = blk * per, hi = lo + per;
        for (size_t i = lo + threadIdx.x; i + 256 * (U - 1) < hi; i += 256 * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[i + 256 * u];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = acc[0];
}

// the same sweep by LDS-DMA (no VGPR staging): every wave keeps U 1-KiB pieces in flight into its own LDS slots
template <int U>
__global__ __launch_bounds__(256) void rd_dma(const unsigned char* __restrict__ p, size_t bytes, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * U * 1024];
    const size_t per = bytes / gridDim.x, lo = blockIdx.x * per;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)lds + wave * U * 1024;
    for (size_t off = (size_t)wave * U * 1024; off + U * 1024 <= per; off += 4 * U * 1024) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned char* src = p + lo + off + u * 1024 + lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + u * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lds[threadIdx.x] == 0x5a && lds[threadIdx.x + 1] == 0xa5 && lds[threadIdx.x + 2] == 0x77) out[0] = 1;
}

template <int U, int ORDER>
static float run(const u32x4* p, size_t n16, int wpc, unsigned* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); rd<U, ORDER><<<256 * wpc, 256>>>(p, n16, out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[1];
}
template <int U>
static float run_dma(const unsigned char* p, size_t bytes, int wpc, unsigned* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a); rd_dma<U><<<256 * wpc, 256>>>(p, bytes, out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[1];
}

int main() {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    u32x4* p; hipMalloc(&p, bytes); hipMemset(p, 1, bytes);
    unsigned* out; hipMalloc(&out, 4);
    printf("| loads in flight per thread | workgroups per CU | KiB in flight per CU | own range TB/s | interleaved TB/s | XCD-contiguous TB/s | LDS-DMA, own range TB/s |\n|---|---|---|---|---|---|---|\n");
    const int wpcs[] = {1, 2, 4, 8};
#define ROW(U)                                                                                                     \
    for (int wpc : wpcs) {                                                                                         \
        const float a = run<U, 0>(p, n16, wpc, out), b = run<U, 1>(p, n16, wpc, out), c = run<U, 2>(p, n16, wpc, out); \
        const float d = (U <= 8 && wpc * 4 * U <= 160) ? run_dma<U>((const unsigned char*)p, bytes, wpc, out) : 0.f;    \
        printf("| %d | %d | %d | %.2f | %.2f | %.2f | %s |\n", U, wpc, wpc * 256 * 16 * U / 1024, bytes / a / 1e9, bytes / b / 1e9,  \
               bytes / c / 1e9, d > 0 ? (std::to_string(bytes / d / 1e9).substr(0, 4)).c_str() : "-");             \
    }
    ROW(1) ROW(2) ROW(4) ROW(8)
    return 0;
}
