// Microbenchmark (not product code): what one CU sustains in v_mfma_f64_16x16x4_f64 when the operands come from LDS the
// way the native-ring DFT kernels fetch them - 8 waves per CU (one 512-thread workgroup), NACC independent accumulators per
// wave, per K step one 16-byte twiddle read (random or linear position) + two 8-byte coefficient reads feeding NACC MFMAs.
//   MODE 0: MFMAs only (operands in registers)          MODE 1: + linear (conflict-free) LDS reads
//   MODE 2: + 16-byte reads at pseudo-random positions (MODE 4: requested one step ahead)    MODE 3: twiddle by rotation in registers (4 f64 VALU ops), 8-byte reads only
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mf profiles/micro/mfma_f64_rate.hip && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(512) k(double *out, long long *cyc, int steps, int L) {
    extern __shared__ double sm[];
    double2 *twl = reinterpret_cast<double2 *>(sm);          // [L]
    double *Cc = sm + 2 * L, *Cs = Cc + 128 * 17;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    for (int m = tid; m < L; m += 512) twl[m] = make_double2(1.0 / (m + 1), 0.5 / (m + 2));
    for (int e = tid; e < 2 * 128 * 17; e += 512) Cc[e] = 1e-3 * e;
    __syncthreads();
    d4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = d4{0.0, 0.0, 0.0, 0.0};
    const int l = (tid >> 6) * 16 + i;
    int m = (MODE == 2) ? (l * 37 + kk * 11) % L : lane;
    const int sm8 = (MODE == 2) ? (8 * l * 13 + 5) % L : 64;
    double2 t = twl[m];
    const double2 w = make_double2(0.999, 0.0447);
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 4) {
        // MODE 2 with the operands of step js + 1 requested before the MFMAs of step js (two register sets)
        auto ld = [&](int js_, int m_, double2 &t_, double &bc_, double &bs_) {
            t_ = twl[m_];
            bc_ = Cc[((js_ & 15) * 8 + 2 * kk) * 17 + i];
            bs_ = Cs[((js_ & 15) * 8 + 2 * kk) * 17 + i];
        };
        auto adv = [&](int m_) { m_ += sm8; return m_ >= L ? m_ - L : m_; };
        auto mm = [&](const double2 &t_, double bc_, double bs_) {
#pragma unroll
            for (int a = 0; a < NACC; a += 2) {
                acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(t_.x, bc_, acc[a], 0, 0, 0);
                acc[a + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(t_.y, bs_, acc[a + 1], 0, 0, 0);
            }
        };
        double2 tA, tB; double bcA, bsA, bcB, bsB;
        int mA = m;
        ld(0, mA, tA, bcA, bsA);
        for (int js = 0; js < steps; js += 2) {
            const int mB = adv(mA);
            ld(js + 1, mB, tB, bcB, bsB);
            mm(tA, bcA, bsA);
            mA = adv(mB);
            ld(js + 2, mA, tA, bcA, bsA);
            mm(tB, bcB, bsB);
        }
    } else
    for (int js = 0; js < steps; js++) {
        double bc = 1.0, bs = 2.0;
        if (MODE == 1 || MODE == 2) {
            t = twl[m];
            m += sm8; if (m >= L) m -= L;
        }
        if (MODE >= 1) {
            bc = Cc[((js & 15) * 8 + 2 * kk) * 17 + i];
            bs = Cs[((js & 15) * 8 + 2 * kk) * 17 + i];
        }
        if (MODE == 3) t = make_double2(t.x * w.x - t.y * w.y, t.x * w.y + t.y * w.x);
#pragma unroll
        for (int a = 0; a < NACC; a += 2) {
            acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(t.x, bc, acc[a], 0, 0, 0);
            acc[a + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(t.y, bs, acc[a + 1], 0, 0, 0);
        }
    }
    d4 s = acc[0];
#pragma unroll
    for (int a = 1; a < NACC; a++) s += acc[a];
    const long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NACC>
static int run(const char *name, double *out, long long *cyc, int nwg) {
    const int steps = 4096, L = 1020;
    const size_t lds = sizeof(double) * (2 * L + 2 * 128 * 17);      // 51 KB: up to three workgroups per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(nwg), dim3(512), lds, 0, out, cyc, steps, L);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(nwg), dim3(512), lds, 0, out, cyc, steps, L);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> h(nwg);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * nwg, hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long c : h) mean += (double)c / nwg;
    // per SIMD: 2 waves x NACC MFMAs per step
    const double mfma_per_simd = 2.0 * NACC * steps;                // per workgroup; nwg / 256 workgroups share a CU
    const double tflops = (double)nwg * 8 * NACC * steps * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%d waves/CU  %-58s accumulators %d: %7.1f cycles per MFMA per SIMD, %6.1f TFLOP/s, clock %.2f GHz (%.3f ms)\n", nwg / 256 * 8, name, NACC, mean / mfma_per_simd / (nwg / 256), tflops,
           mean / (ms * 1e-3) / 1e9, ms);
    return 0;
}

int main() {
    const int nwg = 256;                   // one workgroup of 8 waves per CU
    double *out; long long *cyc;
    CK(hipMalloc(&out, sizeof(double) * nwg * 3 * 512));
    CK(hipMalloc(&cyc, sizeof(long long) * nwg * 3));
    for (int wpc = 2; wpc <= 3; wpc++) {         // 16 / 24 waves per CU
        if (run<2, 2>("+ LDS operands, 16-byte reads at scattered positions", out, cyc, 256 * wpc)) return 1;
        if (run<2, 4>("+ LDS operands, 16-byte reads at scattered positions", out, cyc, 256 * wpc)) return 1;
        if (run<3, 2>("twiddle by rotation in registers, 8-byte LDS reads only", out, cyc, 256 * wpc)) return 1;
    }
    if (run<0, 2>("MFMAs only", out, cyc, nwg)) return 1;
    if (run<0, 6>("MFMAs only", out, cyc, nwg)) return 1;
    if (run<1, 2>("+ LDS operands, linear 16-byte reads", out, cyc, nwg)) return 1;
    if (run<2, 2>("+ LDS operands, 16-byte reads at scattered positions", out, cyc, nwg)) return 1;
    if (run<2, 4>("+ LDS operands, 16-byte reads at scattered positions", out, cyc, nwg)) return 1;
    if (run<2, 6>("+ LDS operands, 16-byte reads at scattered positions", out, cyc, nwg)) return 1;
    if (run<4, 2>("scattered 16-byte reads, requested one step ahead", out, cyc, nwg)) return 1;
    if (run<4, 6>("scattered 16-byte reads, requested one step ahead", out, cyc, nwg)) return 1;
    if (run<3, 2>("twiddle by rotation in registers, 8-byte LDS reads only", out, cyc, nwg)) return 1;
    if (run<3, 6>("twiddle by rotation in registers, 8-byte LDS reads only", out, cyc, nwg)) return 1;
    return 0;
}
