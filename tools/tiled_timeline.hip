// Phase timeline of one workgroup of fft_tiled_kernel in steady state (development tool).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fft_tiled.h"
using namespace pf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#ifndef CFG
#define CFG TiledAltF32::C8192
#endif
#ifndef REALV
#define REALV 1
#endif
#ifndef FLAGS
#define FLAGS 2
#endif
int main() {
    typedef CFG C;
    const int n = C::n; const size_t batch = (size_t)(1ull << 32) / (n * 8) / 2;
    std::vector<cx<float>> tw(n), twr(n / 2 + 1);
    for (int j = 0; j < n; ++j) { double a = -2 * M_PI * j / n; tw[j].x = cos(a); tw[j].y = sin(a); }
    for (int k = 0; k <= n / 2; ++k) { double a = -2 * M_PI * k / (2 * n); twr[k].x = cos(a); twr[k].y = sin(a); }
    float *x, *y; cx<float> *dtw, *dtwr; unsigned* ctr;
    CK(hipMalloc(&x, batch * n * 8)); CK(hipMalloc(&y, batch * n * 8)); CK(hipMemset(x, 0x3c, batch * n * 8));
    CK(hipMalloc(&dtw, n * 8)); CK(hipMalloc(&dtwr, (n / 2 + 1) * 8)); CK(hipMalloc(&ctr, 8)); CK(hipMemset(ctr, 0, 8));
    CK(hipMemcpy(dtw, tw.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwr, twr.data(), (n / 2 + 1) * 8, hipMemcpyHostToDevice));
    auto k = fft_tiled_kernel<C, FWD, REALV>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, C::WG_THREADS, C::LDS_BYTES));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(256 * per_cu), dim3(C::WG_THREADS), C::LDS_BYTES, 0, x, y, (unsigned)batch, FLAGS, dtw, dtwr, ctr);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        long long d[64]; CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(pf_tdbg), sizeof d));
        printf("rep %d: %.3f ms %.0f GB/s (wg/cu %d); deltas:", rep, ms, 2.0 * batch * n * 8 / ms / 1e6, per_cu);
        for (int i = 1; i <= 15; ++i) printf(" %d:%lld", i, d[i] - d[i - 1]);
        printf(" total %lld\n", d[15] - d[0]);
    }
    return 0;
}
