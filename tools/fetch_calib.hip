// What does rocprofv3's FETCH_SIZE count for SCATTERED narrow reads?  MI355X_MICROARCH.md calibrates it for wide coalesced
// streaming reads only (the counter reports half of those bytes, hence the "x 2" in tools/summarize_profile.py) and calls other
// access widths uncalibrated.  K1 (equi2cube) gathers 12 bytes per lane; its "1.85 x algorithmic" traffic of round 2 rests on
// that x 2.  Three kernels over a 1 GiB buffer (4 x the Infinity Cache), each touching every 128-byte line exactly once:
//   stream : 16 B per lane, lanes consecutive            (the calibrated case: 1 GiB really read)
//   word128: one dword per 128-byte line                 (full lines or 64-byte sectors?  time tells: HBM-bound either way)
//   word64 : one dword per 64-byte half line             (touches both halves of every line)
//   k1like : 12 B per lane at a 7-byte lane stride (2.3 px x 3 B), rows 2.3 lines apart - K1's own pattern
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib      (counter per kernel)
//   /tmp/fetch_calib                                                                              (times)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void stream(const uint4* p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int STRIDE>
__global__ __launch_bounds__(256) void word(const unsigned char* p, size_t bytes, unsigned* out) {
    unsigned acc = 0;
    const size_t n = bytes / STRIDE;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += *reinterpret_cast<const unsigned*>(p + i * STRIDE);
    if (acc == 0x12345678u) out[0] = acc;
}
// lane l of "row" r reads 12 bytes at r * 6144 * 2 + (l * 7 & ~3): every other 6 KiB row (as K1's taps skip rows), 7-byte lane pitch
__global__ __launch_bounds__(256) void k1like(const unsigned char* p, size_t bytes, unsigned* out) {
    unsigned acc = 0;
    const size_t rows = bytes / 6144;
    const size_t items = rows / 2 * 768;                                  // 768 lanes cover one 6 KiB row at 8-byte effective pitch
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const size_t r = i / 768, l = i - r * 768;
        const uint3 v = *reinterpret_cast<const uint3*>(p + (2 * r) * 6144 + ((l * 8) & ~(size_t)3));
        acc += v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    unsigned char* p; hipMalloc(&p, bytes); hipMemset(p, 1, bytes);
    unsigned* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 32;
    for (int k = 0; k < 4; ++k) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(a);
            if (k == 0) stream<<<blocks, 256>>>((const uint4*)p, bytes / 16, out);
            else if (k == 1) word<128><<<blocks, 256>>>(p, bytes, out);
            else if (k == 2) word<64><<<blocks, 256>>>(p, bytes, out);
            else k1like<<<blocks, 256>>>(p, bytes, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const char* nm[] = {"stream 16 B/lane (1 GiB read)", "one dword per 128-B line   ", "one dword per 64-B half line", "k1like: 12 B / 8-B pitch, every other 6 KiB row"};
        printf("%-50s %8.1f us  -> %6.0f GB/s if full 128-B lines of the touched region are fetched\n", nm[k], best * 1e3,
               (k == 3 ? bytes / 2 : bytes) / best / 1e6);
    }
    return 0;
}
