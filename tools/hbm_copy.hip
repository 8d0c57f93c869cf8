// measured HBM ceiling of the box with hand-written kernels: 16-byte-per-lane copy (read + write) and read-only sum,
// grid-stride over 4 GiB.  hipcc --offload-arch=gfx950 -O3 tools/hbm_copy.hip -o tools/bin/hbm_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy(uint4 *dst, const uint4 *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_sum(const uint4 *src, size_t n, unsigned long long *out) {
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 0x123456789ull) *out = a;                       // keeps the loads alive, practically never written
}
int main(int argc, char **argv) {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    uint4 *a, *b; unsigned long long *o;
    CHK(hipMalloc((void **)&a, bytes)); CHK(hipMalloc((void **)&b, bytes)); CHK(hipMalloc((void **)&o, 8));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 32768, 131072}) {
        for (int which = 0; which < 2; which++) {
            for (int w = 0; w < 2; w++) { if (which == 0) k_copy<<<grid, 256>>>(b, a, n); else k_sum<<<grid, 256>>>(a, n, o); }
            CHK(hipEventRecord(e0));
            for (int r = 0; r < 10; r++) { if (which == 0) k_copy<<<grid, 256>>>(b, a, n); else k_sum<<<grid, 256>>>(a, n, o); }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (which == 0 ? 2.0 : 1.0) * (double)bytes * 10 / (ms * 1e-3) / 1e9;
            printf("%s grid %6d: %.0f GB/s\n", which == 0 ? "copy (read + write)" : "sum  (read)        ", grid, gbs);
        }
    }
    return 0;
}
