// Microbenchmark (not product code): does the access WIDTH per lane bound the equation-set kernel?
// Emulates k_phys_hrbl_cell's memory behaviour without its arithmetic: a 256-thread workgroup (2 resident per CU, forced
// by a 64 KB LDS block) reads NL planes and writes NS planes, 2 KB contiguous per plane and workgroup.
//   variant 0: 8 B per lane  (global_load/store_dwordx2), what the kernel does today
//   variant 1: 16 B per lane: lanes 0-31 take plane p, lanes 32-63 plane p+1, then v_permlane32_swap hands every
//              lane its own level of both planes
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/lw profiles/micro/lane_width.hip && /tmp/lw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NL, int NS, int VAR>
__global__ void __launch_bounds__(256, 2) k(const double *__restrict__ in, double *__restrict__ out, int64_t plane, int lds_words) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * 256;
    double v[NL];
    if (VAR == 0) {
#pragma unroll
        for (int p = 0; p < NL; p++) v[p] = __builtin_nontemporal_load(in + (int64_t)p * plane + base + tid);
    } else {
        const int half = lane >> 5, i = lane & 31;
#pragma unroll
        for (int p = 0; p < NL; p += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(in + (int64_t)(p + half) * plane + base + wave * 64 + 2 * i);
            const unsigned xl = __double2loint(x.x), xh = __double2hiint(x.x), yl = __double2loint(x.y), yh = __double2hiint(x.y);
            auto r0 = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
            v[p] = __hiloint2double(r1[0], r0[0]);
            v[p + 1] = __hiloint2double(r1[1], r0[1]);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < NL; p++) s += v[p];
    if (lds_words < 0) sm[tid] = s;        // never: keeps the LDS allocation
    if (VAR == 0) {
#pragma unroll
        for (int p = 0; p < NS; p++) __builtin_nontemporal_store(s + p, out + (int64_t)p * plane + base + tid);
    } else {
        const int half = lane >> 5, i = lane & 31;
#pragma unroll
        for (int p = 0; p < NS; p += 2) {
            const double a = s + p, b = s + p + 1;
            const unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
            auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
            double2 y = make_double2(__hiloint2double(r1[0], r0[0]), __hiloint2double(r1[1], r0[1]));
            *reinterpret_cast<double2 *>(out + (int64_t)(p + half) * plane + base + wave * 64 + 2 * i) = y;
        }
    }
}

// plain full-occupancy copy, W doubles per lane
template <int W>
__global__ void __launch_bounds__(256) kcopy(const double *__restrict__ in, double *__restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * W;
    if (i + W <= n) {
        if (W == 1) out[i] = in[i];
        else *reinterpret_cast<double2 *>(out + i) = *reinterpret_cast<const double2 *>(in + i);
    }
}

template <int NL, int NS, int VAR>
static int run(const double *in, double *out, int nwg, int64_t plane, const char *tag, std::vector<double> *check) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t lds = 64 * 1024;
    CK(hipFuncSetAttribute((const void *)k<NL, NS, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k<NL, NS, VAR>), dim3(nwg), dim3(256), lds, 0, in, out, plane, 0);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k<NL, NS, VAR>), dim3(nwg), dim3(256), lds, 0, in, out, plane, 0);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double bytes = (double)(NL + NS) * nwg * 256 * 8;
    printf("%-28s NL=%d NS=%d  %.3f ms  %.0f GB/s  (%.1f B/cyc/CU at 2.4 GHz)\n", tag, NL, NS, ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 256 / 2.4e9);
    if (check) {
        check->resize(4096);
        CK(hipMemcpy(check->data(), out + plane * (NS - 1) + 256 * 7, 4096 * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int main() {
    const int nwg = 8256;
    const int64_t plane = (int64_t)nwg * 256;
    const int NLmax = 86, NSmax = 34;
    double *in, *out;
    CK(hipMalloc(&in, sizeof(double) * plane * NLmax));
    CK(hipMalloc(&out, sizeof(double) * plane * NSmax));
    std::vector<double> h((size_t)plane);
    for (int p = 0; p < NLmax; p++) {
        for (int64_t i = 0; i < plane; i++) h[i] = (double)((i * 7 + p * 13) % 1000) * 1e-3;
        CK(hipMemcpy(in + plane * p, h.data(), sizeof(double) * plane, hipMemcpyHostToDevice));
    }
    std::vector<double> c0, c1;
    if (run<86, 34, 0>(in, out, nwg, plane, "8 B/lane", &c0)) return 1;
    if (run<86, 34, 1>(in, out, nwg, plane, "16 B/lane + permlane32_swap", &c1)) return 1;
    double md = 0; for (size_t i = 0; i < c0.size(); i++) md = fmax(md, fabs(c0[i] - c1[i]));
    printf("max |variant0 - variant1| on a sample: %g (must be 0)\n", md);
    if (run<30, 34, 0>(in, out, nwg, plane, "8 B/lane (no node planes)", nullptr)) return 1;
    if (run<30, 34, 1>(in, out, nwg, plane, "16 B/lane (no node planes)", nullptr)) return 1;
    if (run<86, 2, 0>(in, out, nwg, plane, "8 B/lane loads only", nullptr)) return 1;
    if (run<86, 2, 1>(in, out, nwg, plane, "16 B/lane loads only", nullptr)) return 1;
    if (run<2, 34, 0>(in, out, nwg, plane, "8 B/lane stores only", nullptr)) return 1;
    if (run<2, 34, 1>(in, out, nwg, plane, "16 B/lane stores only", nullptr)) return 1;
    // reference: plain copies at full occupancy
    const int64_t n = plane * 30;
    for (int w = 1; w <= 2; w++) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const int64_t nb = (n / w + 255) / 256;
        for (int it = 0; it < 2; it++) { if (w == 1) hipLaunchKernelGGL(kcopy<1>, dim3(nb), dim3(256), 0, 0, in, out, n); else hipLaunchKernelGGL(kcopy<2>, dim3(nb), dim3(256), 0, 0, in, out, n); }
        CK(hipEventRecord(a));
        for (int it = 0; it < 10; it++) { if (w == 1) hipLaunchKernelGGL(kcopy<1>, dim3(nb), dim3(256), 0, 0, in, out, n); else hipLaunchKernelGGL(kcopy<2>, dim3(nb), dim3(256), 0, 0, in, out, n); }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
        printf("plain copy %2d B/lane: %.3f ms  %.0f GB/s (read + write)\n", 8 * w, ms, 2.0 * n * 8 / ms / 1e6);
    }
    return 0;
}
