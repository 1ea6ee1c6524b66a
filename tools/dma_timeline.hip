// Phase timeline of the FIR block kernels of fft_dma.h (development tool): one steady-state iteration of workgroup 7, wave 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPF_DMA_DEBUG -I pffft_amd/csrc tools/dma_timeline.hip -o tools/_bin/dma_timeline
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "fft_dma.h"
#include "fft_split.h"
using namespace pf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static void report(const char* name, float us, long nblk, int cus, int per_cu) {
    long long d[64];
    if (hipMemcpyFromSymbol(d, HIP_SYMBOL(pf_ddbg), sizeof d) != hipSuccess) return;
    printf("%s: %.1f us total, %.2f us per block and workgroup slot (%d x %d slots)\n  stamps (x 10 ns since iteration start):", name, us,
           us / ((double)nblk / (cus * per_cu)), cus, per_cu);
    for (int i = 1; i <= 40; ++i) if (d[i] > d[0]) printf(" [%d] %lld", i, d[i] - d[0]);
    printf("\n");
    long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pf_ddbg), z, sizeof z);
}

template <class C, bool ONE, int SPREAD = 0>
static int run(const char* name, const float* x, float* y, const cx<float>* dH, const cx<float>* dtw, const cx<float>* dtwr, unsigned* ctr,
               long L, int taps, int cus) {
    const int n = C::n, Nfft = 2 * n, step = Nfft - taps + 1;
    const long nblk = (L - taps + 1 + step - 1) / step;
    const int lastOut = (int)(L - taps + 1 - (nblk - 1) * step);
    size_t lds; int per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(ctr, 0, 8));
        if constexpr (ONE) {
            return 1;   // (the one-image / two-workgroup kernel of round 4 was removed: git history)
        } else {
            auto k = fastconv_dma_kernel<C, SPREAD>;
            lds = DmaGeom<C>::LDS_BYTES;
            per_cu = 1;
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(cus), dim3(C::WG_THREADS), lds, 0, x, y, dH, (int)nblk, step, (int)L, lastOut, dtw, dtwr, ctr, 1, (size_t)0, (size_t)0);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 3) {
            printf("[lds %zu B, %d workgroups per CU, fraction of the 8 B / sample roofline %.3f] ", lds, per_cu, 8.0 * (L - taps + 1) / (ms * 1e-3) / 8e12);
            report(name, ms * 1e3f, nblk, cus, per_cu);
        }
    }
    return 0;
}

template <int PSYNC, int SPREAD, int W = 8>
static int run_split(const float* x, float* y, const cx<float>* dH, const cx<float>* dtw, const cx<float>* dtw1024, const cx<float>* dtwr,
                     unsigned* ctr, long L, int taps, int cus) {
    typedef SplitFirT<W> SF;
    const int n = SF::n, Nfft = 2 * n, step = Nfft - taps + 1;
    const long nblk = (L - taps + 1 + step - 1) / step;
    const int lastOut = (int)(L - taps + 1 - (nblk - 1) * step);
    auto k = fastconv_split_kernel<PSYNC, SPREAD, W>;
    int per_cu = 1;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, SF::WG, SF::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF::LDS_BYTES));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(ctr, 0, 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(cus * per_cu), dim3(SF::WG), SF::LDS_BYTES, 0, x, y, dH, (int)nblk, step, (int)L, lastOut, dtw, dtw1024, dtwr, ctr, 1, (size_t)0, (size_t)0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 3) {
            printf("[lds %zu B, fraction of the 8 B / sample roofline %.3f] ", (size_t)SF::LDS_BYTES, 8.0 * (L - taps + 1) / (ms * 1e-3) / 8e12);
            report(PSYNC ? (SPREAD ? "split, pairwise flags, pieces spread" : "split, pairwise flags around the mirror exchange") : (SPREAD ? "split, pieces spread over B and B'" : "split: cross-wave radix 8 + wave-local 1024-point transforms"), ms * 1e3f, nblk, cus, per_cu);
        }
    }
    return 0;
}

static int run_one(const float* x, float* y, int cus) {
    // the stated C4 call: 2^20 samples, 4096 taps, 255 blocks of 8192 samples (n = 4096), one launch at a time
    const int n = 4096, Nfft = 8192, taps = 4096; const long L = 1 << 20; const int step = Nfft - taps + 1;
    const int nblk = (int)((L - taps + 1 + step - 1) / step), lastOut = (int)(L - taps + 1 - (long)(nblk - 1) * step);
    std::vector<cx<float>> tw(n), twr(n / 2 + 1), H(n), t5(512);
    for (int j = 0; j < n; ++j) { double a = -2 * M_PI * j / n; tw[j].x = cos(a); tw[j].y = sin(a); }
    for (int k = 0; k <= n / 2; ++k) { double a = -2 * M_PI * k / Nfft; twr[k].x = cos(a); twr[k].y = sin(a); }
    for (int k = 0; k < n; ++k) { H[k].x = 1.0f / Nfft; H[k].y = 0; }
    for (int j = 0; j < 512; ++j) { double a = -2 * M_PI * j / 512; t5[j].x = cos(a); t5[j].y = sin(a); }
    cx<float>*dtw, *dtwr, *dH, *d5;
    CK(hipMalloc(&dtw, n * 8)); CK(hipMalloc(&dtwr, (n / 2 + 1) * 8)); CK(hipMalloc(&dH, n * 8)); CK(hipMalloc(&d5, 512 * 8));
    CK(hipMemcpy(dtw, tw.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dtwr, twr.data(), (n / 2 + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, H.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d5, t5.data(), 512 * 8, hipMemcpyHostToDevice));
    vec4<float>* dAB; CK(hipMalloc(&dAB, n * 16));
    hipLaunchKernelGGL(fastconv_split1_coef_kernel<8>, dim3(1), dim3(512), 0, 0, dH, dtwr, dAB);
    auto k = fastconv_split1_kernel<8>;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(512), SplitOneT<8>::LDS_BYTES, 0, x, y, dAB, nblk, step, (int)L, lastOut, dtw, d5, 1, (size_t)0, (size_t)0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 3) report("one-shot split kernel, the stated C4 call (stamps: 1 gather issued, 2 constants, 3 A, 4 barrier, 5 B, 6 mirror, 7 B', 8 barrier, 9 A' + stores)", ms * 1e3f, nblk, cus, 1);
    }
    return 0;
}

int main() {
    const int n = 8192, Nfft = 2 * n, taps = 4096;
    const long L = 1L << 26;
    std::vector<cx<float>> tw(n), twr(n / 2 + 1), H(n);
    for (int j = 0; j < n; ++j) { double a = -2 * M_PI * j / n; tw[j].x = cos(a); tw[j].y = sin(a); }
    for (int k = 0; k <= n / 2; ++k) { double a = -2 * M_PI * k / Nfft; twr[k].x = cos(a); twr[k].y = sin(a); }
    for (int k = 0; k < n; ++k) { H[k].x = 1.0f / Nfft; H[k].y = 0; }
    float *x, *y; cx<float> *dtw, *dtwr, *dH; unsigned* ctr;
    CK(hipMalloc(&x, L * 4)); CK(hipMalloc(&y, L * 4)); CK(hipMemset(x, 0, L * 4)); CK(hipMalloc(&ctr, 8));
    CK(hipMalloc(&dtw, n * 8)); CK(hipMalloc(&dtwr, (n / 2 + 1) * 8)); CK(hipMalloc(&dH, n * 8));
    CK(hipMemcpy(dtw, tw.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwr, twr.data(), (n / 2 + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, H.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<cx<float>> t1k(1024);
    for (int j = 0; j < 1024; ++j) { double a = -2 * M_PI * j / 1024; t1k[j].x = cos(a); t1k[j].y = sin(a); }
    cx<float>* dtw1024; CK(hipMalloc(&dtw1024, 1024 * 8)); CK(hipMemcpy(dtw1024, t1k.data(), 1024 * 8, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("clock rate %d kHz, %d CUs\n", prop.clockRate, cus);
    if (run_one(x, y, cus)) return 1;
    if (run_split<0, 0>(x, y, dH, dtw, dtw1024, dtwr, ctr, L, taps, cus)) return 1;
    if (run_split<0, 1>(x, y, dH, dtw, dtw1024, dtwr, ctr, L, taps, cus)) return 1;
    if (run_split<1, 1>(x, y, dH, dtw, dtw1024, dtwr, ctr, L, taps, cus)) return 1;
    {   // four wavefronts, 8192-sample blocks, 2048 taps (the W_n table of n = 4096: every second entry of the 8192 one)
        std::vector<cx<float>> t4(4096);
        for (int j = 0; j < 4096; ++j) { double a = -2 * M_PI * j / 4096; t4[j].x = cos(a); t4[j].y = sin(a); }
        std::vector<cx<float>> tr4(2049);
        for (int k = 0; k <= 2048; ++k) { double a = -2 * M_PI * k / 8192; tr4[k].x = cos(a); tr4[k].y = sin(a); }
        cx<float>*d4, *dr4; CK(hipMalloc(&d4, 4096 * 8)); CK(hipMalloc(&dr4, 2049 * 8));
        CK(hipMemcpy(d4, t4.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dr4, tr4.data(), 2049 * 8, hipMemcpyHostToDevice));
        printf("(2048 taps:) ");
        if (run_split<1, 1, 4>(x, y, dH, d4, dtw1024, dr4, ctr, L, 2048, cus)) return 1;
        printf("(2048 taps, 16384-sample blocks:) ");
        if (run_split<1, 1, 8>(x, y, dH, dtw, dtw1024, dtwr, ctr, L, 2048, cus)) return 1;
    }
    if (run<DmaCfgF32::D8192, false>("two images, 512 threads (D8192)", x, y, dH, dtw, dtwr, ctr, L, taps, cus)) return 1;
    if (run<DmaCfgF32::D8192, false, 1>("two images, 512 threads, pieces spread over the phases", x, y, dH, dtw, dtwr, ctr, L, taps, cus)) return 1;
    return 0;
}
