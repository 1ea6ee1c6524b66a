// Phase timeline of the 32-points-per-thread FIR block kernel (fft_fir32.h; development tool): one steady-state iteration of workgroup 7,
// thread 0, in cycles of the shader clock counter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPF_FIR32_DEBUG -I pffft_amd/csrc -I include tools/fir32_timeline.hip -o tools/_bin/fir32_timeline
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../include/pffft_hip.h"
#include "fft_fir32.h"
using namespace pf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int PREF>
static int run(const float* x, float* y, const vec4<float>* HP, const cx<float>* dtw, const cx<float>* dtwr, unsigned* ctr, long L, int taps, int cus, int nsig) {
    const int n = Fir32::n, Nfft = 2 * n, step = Nfft - taps + 1;
    const long nblk = (L - taps + 1 + step - 1) / step;
    const int lastOut = (int)(L - taps + 1 - (nblk - 1) * step);
    auto k = fastconv_fused32_kernel<PREF>;
    int per_cu = 1;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fir32::LDS_BYTES));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, Fir32::WG, Fir32::LDS_BYTES));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipMemset(ctr, 0, 64));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(cus * per_cu), dim3(Fir32::WG), Fir32::LDS_BYTES, 0, x, y, HP, (int)nblk, step, (int)L, lastOut, dtw, dtwr, ctr, nsig, (size_t)L, (size_t)L, 1);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        if (rep == 11) {
            ms = best;
            long long d[64];
            CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(pf_f32dbg), sizeof d));
            printf("PREF %d, %d taps, %d x 2^%d: %.1f us, fraction of the 8 B / sample roofline %.3f, %d workgroups per CU, %.0f ns per block and CU\n  stamps (cycles since the top of the iteration; 100 MHz counter x clock ratio):",
                   PREF, taps, nsig, (int)log2((double)L), ms * 1e3, 8.0 * nsig * (L - taps + 1) / (ms * 1e-3) / 8e12, per_cu, ms * 1e6 / ((double)nblk * nsig / cus));
            for (int i = 1; i <= 13; ++i) printf(" [%d] %lld", i, d[i] - d[0]);
            printf("\n");
        }
    }
    return 0;
}

int main() {
    const int n = Fir32::n, Nfft = 2 * n;
    const long L = 1L << 20; const int nsig = 256;
    std::vector<cx<float>> tw(n), twr(n / 2 + 1), H(n);
    for (int j = 0; j < n; ++j) { double a = -2 * M_PI * j / n; tw[j].x = cos(a); tw[j].y = sin(a); }
    for (int k = 0; k <= n / 2; ++k) { double a = -2 * M_PI * k / Nfft; twr[k].x = cos(a); twr[k].y = sin(a); }
    for (int k = 0; k < n; ++k) { H[k].x = 1.0f / Nfft; H[k].y = 0; }
    float *x, *y; cx<float> *dtw, *dtwr, *dH; unsigned* ctr; vec4<float>* HP;
    CK(hipMalloc(&x, L * nsig * 4)); CK(hipMalloc(&y, L * nsig * 4)); CK(hipMemset(x, 0, L * nsig * 4)); CK(hipMalloc(&ctr, 64));
    CK(hipMalloc(&dtw, n * 8)); CK(hipMalloc(&dtwr, (n / 2 + 1) * 8)); CK(hipMalloc(&dH, n * 8)); CK(hipMalloc(&HP, n * 16));
    CK(hipMemcpy(dtw, tw.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwr, twr.data(), (n / 2 + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, H.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fir32_coef_kernel, dim3(1), dim3(Fir32::WG), 0, 0, dH, dtwr, HP);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("clock rate %d kHz, %d CUs\n", prop.clockRate, cus);
    for (int taps : {4096}) {
        if (run<2>(x, y, HP, dtw, dtwr, ctr, L, taps, cus, nsig)) return 1;
    }
    return 0;
}
