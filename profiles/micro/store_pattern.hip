// Microbenchmark (not product code): store throughput of the inverse FFT kernels' copy-out pattern.  A 512-thread workgroup
// writes SLOTS tiles of 256 ring points x 16 levels x 8 B = 32 KB with 16-byte non-temporal stores per lane, either
//   MODE 0  as today: the tile is 256 pieces of 128 bytes at a stride of zDim x 8 = 512 bytes (the four level-chunk workgroups of a
//           (unit, variable) interleave their pieces: layout [point][64 levels]), or
//   MODE 1  as one contiguous 32 KB block (layout [level chunk][point][16 levels]).
// Same bytes, same instruction count.   hipcc --offload-arch=gfx950 -O3 -o /tmp/sp profiles/micro/store_pattern.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512) k(double *out, int slots, long plane) {
    // blockIdx.x = 4 * unit + level chunk; a unit's plane of a slot: 256 points x 64 levels
    const int unit = blockIdx.x >> 2, zc = blockIdx.x & 3;
    const int zp = threadIdx.x & 7;
    for (int s = 0; s < slots; s++) {
        double *base = out + (long)s * plane + (long)unit * 256 * 64;
        for (int l = threadIdx.x >> 3; l < 256; l += 64) {
            d2 v; v.x = l + s; v.y = zp;
            double *p = MODE == 0 ? base + (long)l * 64 + zc * 16 + 2 * zp : base + ((long)zc * 256 + l) * 16 + 2 * zp;
            __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
        }
    }
}

int main() {
    const int units = 132 * 5, slots = 3;                 // 132 nodes x 5 variables, ~3 planes each: 0.32 GB
    const long plane = (long)units * 256 * 64;
    double *d;
    CK(hipMalloc(&d, sizeof(double) * plane * slots));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++)
        for (int mode = 0; mode < 2; mode++) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 20; it++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(units * 4), dim3(512), 0, 0, d, slots, plane);
                else hipLaunchKernelGGL(k<1>, dim3(units * 4), dim3(512), 0, 0, d, slots, plane);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d (%s): %.4f ms per launch, %.0f GB/s\n", mode, mode == 0 ? "128-byte pieces at 512-byte stride" : "contiguous 32 KB per workgroup",
                   ms / 20, 8.0 * plane * slots / (ms / 20 * 1e-3) / 1e9);
        }
    return 0;
}
