// Microbenchmark (not product code): memory skeleton of a PERSISTENT, software-pipelined equation-set kernel.
// A 256-thread workgroup owns one block of 4 azimuths x 64 levels and walks NC radial cells; per cell it needs NL fresh
// 2-KB chunks (14 node transforms of the one new spline node + 30 history values) and writes NS (33).  The loads of cell
// c + 1 are issued before cell c is finished, so their latency is covered by the stores (and, in the real kernel, the
// arithmetic) of cell c.  Compared with the one-shot form (all loads, then all stores, one cell per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pl profiles/micro/pipeline.hip && /tmp/pl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double dbl2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_pair_nt(double *pa, double *pb, int lane, double xa, double xb) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__double2loint(xa), __double2loint(xb), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__double2hiint(xa), __double2hiint(xb), false, false);
    dbl2v t; t.x = __hiloint2double(r1[0], r0[0]); t.y = __hiloint2double(r1[1], r0[1]);
    __builtin_nontemporal_store(t, reinterpret_cast<dbl2v *>(lane < 32 ? pa : pb));
}

// element offset of (cell, lam block) inside a plane: cell-major, 64 lam blocks of 256 doubles per cell
__device__ __forceinline__ int64_t chunk(int cell, int lb) { return ((int64_t)cell * 64 + lb) * 256; }

template <int NL, int NS, bool PIPE, bool WSTORE, int WORK>
__global__ void __launch_bounds__(256, 2) k(const double *__restrict__ in, double *__restrict__ out, int64_t plane, int ncell_seg, int ncells) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int lb = blockIdx.x & 63, seg = blockIdx.x >> 6;
    const int c0 = seg * ncell_seg, c1 = min(ncells, c0 + ncell_seg);
    const int eoff = WSTORE ? (tid & ~63) + 2 * (lane & 31) : tid;
    double cur[NL], nxt[NL];
    if (PIPE) {
#pragma unroll
        for (int p = 0; p < NL; p++) cur[p] = __builtin_nontemporal_load((const volatile double *)(in + (int64_t)p * plane + chunk(c0, lb) + tid));
    }
    for (int c = c0; c < c1; c++) {
        if (PIPE) {
            if (c + 1 < c1) {
#pragma unroll
                for (int p = 0; p < NL; p++) nxt[p] = __builtin_nontemporal_load((const volatile double *)(in + (int64_t)p * plane + chunk(c + 1, lb) + tid));
            }
            asm volatile("" ::: "memory");      // every load of the next cell is issued before this cell's work starts
        } else {
#pragma unroll
            for (int p = 0; p < NL; p++) cur[p] = __builtin_nontemporal_load((const volatile double *)(in + (int64_t)p * plane + chunk(c, lb) + tid));
            asm volatile("" ::: "memory");      // all loads in flight at once, as in the real kernel (every value is needed later)
        }
        // the first use depends on the LAST load issued (volatile loads keep their order; the memory counter retires them in
        // order), so every load is in flight before anything is consumed - as in the real kernel, where all values are needed
        const double gate = cur[NL - 1] * 0.0;
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < NL; p++) s += cur[p] + gate;
        // stand-in for the arithmetic of a cell: WORK dependent fp64 FMAs per thread
        double w = s;
#pragma unroll 8
        for (int i = 0; i < WORK; i++) w = w * 1.0000001 + 1e-9;
        s = w;
        if (sm[0] == 123.456) s += sm[tid];       // never true: keeps the LDS allocation alive
        if (WSTORE) {
#pragma unroll
            for (int p = 0; p + 1 < NS; p += 2)
                store_pair_nt(out + (int64_t)p * plane + chunk(c, lb) + eoff, out + (int64_t)(p + 1) * plane + chunk(c, lb) + eoff, lane, s + p, s + p + 1);
            if (NS & 1) __builtin_nontemporal_store(s + NS - 1, out + (int64_t)(NS - 1) * plane + chunk(c, lb) + tid);
        } else {
#pragma unroll
            for (int p = 0; p < NS; p++) __builtin_nontemporal_store(s + p, out + (int64_t)p * plane + chunk(c, lb) + tid);
        }
        if (PIPE) {
#pragma unroll
            for (int p = 0; p < NL; p++) cur[p] = nxt[p];
        }
    }
}

template <int NL, int NS, bool PIPE, bool WSTORE, int WORK>
static int run(const double *in, double *out, int64_t plane, int ncells, int nseg, size_t lds, const char *tag) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k<NL, NS, PIPE, WSTORE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ncell_seg = (ncells + nseg - 1) / nseg;
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k<NL, NS, PIPE, WSTORE, WORK>), dim3(64 * nseg), dim3(256), lds, 0, in, out, plane, ncell_seg, ncells);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k<NL, NS, PIPE, WSTORE, WORK>), dim3(64 * nseg), dim3(256), lds, 0, in, out, plane, ncell_seg, ncells);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double bytes = (double)(NL + NS) * ncells * 64 * 256 * 8;
    printf("%-64s NL=%d NS=%d work=%4d segs=%3d lds=%3zuK  %.3f ms  %.0f GB/s\n", tag, NL, NS, WORK, nseg, lds / 1024, ms, bytes / ms / 1e6);
    return 0;
}

int main() {
    const int ncells = 128;
    const int64_t plane = (int64_t)ncells * 64 * 256;
    const int NLmax = 86, NSmax = 34;
    double *in, *out;
    CK(hipMalloc(&in, sizeof(double) * plane * NLmax));
    CK(hipMalloc(&out, sizeof(double) * plane * NSmax));
    CK(hipMemset(in, 0, sizeof(double) * plane * NLmax));
    const size_t K64 = 64 * 1024, K48 = 50 * 1024;
    // one cell per workgroup (today's structure): nseg = ncells
    if (run<86, 33, false, false, 0>(in, out, plane, ncells, 128, K64, "one-shot, 86 loads (today)")) return 1;
    if (run<40, 33, false, false, 0>(in, out, plane, ncells, 128, K64, "one-shot, 40 loads")) return 1;
    if (run<40, 33, false, true, 0>(in, out, plane, ncells, 128, K64, "one-shot, 40 loads, 16-B stores")) return 1;
    // persistent walk, no prefetch
    if (run<40, 33, false, false, 0>(in, out, plane, ncells, 8, K48, "walk 16 cells, no prefetch")) return 1;
    // persistent walk with prefetch of the next cell
    if (run<40, 33, true, false, 0>(in, out, plane, ncells, 8, K48, "walk 16 cells, prefetch")) return 1;
    if (run<40, 33, true, true, 0>(in, out, plane, ncells, 8, K48, "walk 16 cells, prefetch, 16-B stores")) return 1;
    if (run<40, 33, true, true, 0>(in, out, plane, ncells, 16, K48, "walk 8 cells, prefetch, 16-B stores (4 WG/CU worth of grid)")) return 1;
    if (run<40, 33, true, true, 0>(in, out, plane, ncells, 4, K48, "walk 32 cells, prefetch, 16-B stores (1 WG/CU)")) return 1;
    // with a stand-in for the per-cell arithmetic (dependent FMAs: ~4 clk each per wave)
    if (run<40, 33, true, true, 1000>(in, out, plane, ncells, 8, K48, "walk 16 cells, prefetch, 16-B stores, +1000 FMA")) return 1;
    if (run<40, 33, true, true, 3000>(in, out, plane, ncells, 8, K48, "walk 16 cells, prefetch, 16-B stores, +3000 FMA")) return 1;
    if (run<40, 33, false, true, 1000>(in, out, plane, ncells, 128, K64, "one-shot, 40 loads, 16-B stores, +1000 FMA")) return 1;
    if (run<40, 33, false, true, 3000>(in, out, plane, ncells, 128, K64, "one-shot, 40 loads, 16-B stores, +3000 FMA")) return 1;
    if (run<86, 33, false, false, 1000>(in, out, plane, ncells, 128, K64, "one-shot, 86 loads, +1000 FMA")) return 1;
    if (run<86, 33, false, false, 3000>(in, out, plane, ncells, 128, K64, "one-shot, 86 loads, +3000 FMA")) return 1;
    return 0;
}
