// Phase timeline of the fused FIR kernel (development tool): hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -ffp-contract=off -DPF_FIR_DEBUG -I pffft_amd/csrc tools/fir_timeline.hip -o build/fir_timeline
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fft_fir.h"
using namespace pf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class C> static int run(const char* name) {
    const int n = C::n, Nfft = 2 * n, taps = 4096, L = 1 << 20, step = Nfft - taps + 1;
    const int nblk = (L - taps + 1 + step - 1) / step;
    std::vector<cx<float>> tw(n), twr(n / 2 + 1), H(n);
    for (int j = 0; j < n; ++j) { double a = -2 * M_PI * j / n; tw[j].x = cos(a); tw[j].y = sin(a); }
    for (int k = 0; k <= n / 2; ++k) { double a = -2 * M_PI * k / Nfft; twr[k].x = cos(a); twr[k].y = sin(a); }
    for (int k = 0; k < n; ++k) { H[k].x = 1.0f / Nfft; H[k].y = 0; }
    float *x, *y; cx<float> *dtw, *dtwr, *dH;
    CK(hipMalloc(&x, L * 4)); CK(hipMalloc(&y, L * 4)); CK(hipMemset(x, 0, L * 4));
    CK(hipMalloc(&dtw, n * 8)); CK(hipMalloc(&dtwr, (n / 2 + 1) * 8)); CK(hipMalloc(&dH, n * 8));
    CK(hipMemcpy(dtw, tw.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwr, twr.data(), (n / 2 + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, H.data(), n * 8, hipMemcpyHostToDevice));
    auto k = fastconv_fused_kernel<C>;
    printf("%s: %d threads per block, %d blocks\n", name, C::WG_THREADS, nblk);
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(C::WG_THREADS), C::LDS_BYTES, 0, x, y, dH, nblk, step, L, L - taps + 1 - (nblk - 1) * step, dtw, dtwr, (unsigned*)nullptr, 1, (size_t)0, (size_t)0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        long long d[64]; CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(pf_dbg), sizeof d));
        printf("rep %d: %.1f us total; cycles since start:", rep, ms * 1e3);
        for (int i = 1; i <= 8; ++i) printf(" [%d] %lld", i, d[i] - d[0]);
        printf("\n");
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(dtw); (void)hipFree(dtwr); (void)hipFree(dH);
    return 0;
}
int main() {
    if (run<FirCfg::C4096>("C4096 (256 threads, 16 points per thread, four stages)")) return 1;
    if (run<FirCfg::C4096m>("C4096m (512 threads, 8 points per thread, five stages)")) return 1;
    return 0;
}
