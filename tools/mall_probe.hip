// Does the 256 MB Infinity Cache keep what a kernel has just WRITTEN, and does the traversal order of the next
// kernel matter?  (MI355X, build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe)
// Kernel W writes a buffer front to back (blocks in launch order); kernel R then reads it front to back, or back to
// front (most recently written lines first).  Sizes: 154 / 308 / 616 MB (the tensors of the static stage).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void wr(uint4* p, size_t n16, unsigned v) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;             // block b owns a contiguous range: ranges in launch order
    const size_t lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) p[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}
__global__ __launch_bounds__(256) void rd(const uint4* p, size_t n16, int reverse, unsigned* out) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const size_t lo = b * per, hi = lo + per < n16 ? lo + per : n16;
    unsigned acc = 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t sizes[] = {154u << 20, 308u << 20, 616u << 20, 1232u << 20};
    unsigned* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t bytes : sizes) {
        uint4* p; hipMalloc(&p, bytes);
        const size_t n16 = bytes / 16;
        const int blocks = 256 * 64;                                      // many short-lived blocks: dispatch order = address order
        for (int rev = 0; rev < 2; ++rev) {
            float best = 1e9f, wbest = 1e9f;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(a); wr<<<blocks, 256>>>(p, n16, it); hipEventRecord(b); hipEventSynchronize(b);
                float wm; hipEventElapsedTime(&wm, a, b); if (wm < wbest) wbest = wm;
                hipEventRecord(a); rd<<<blocks, 256>>>(p, n16, rev, out); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("%5zu MB  write %.1f us (%.0f GB/s)  read %s %.1f us (%.0f GB/s)\n", bytes >> 20, wbest * 1e3, bytes / wbest / 1e6,
                   rev ? "back-to-front" : "front-to-back", best * 1e3, bytes / best / 1e6);
        }
        hipFree(p);
    }
    return 0;
}
