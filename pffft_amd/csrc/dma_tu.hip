// libpffft_hip.so, translation unit of the LDS-DMA staged FIR block kernel (fft_dma.h): instantiations + launcher.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_dma.h"
#include "fft_split.h"
#include "fft_fir32.h"
#include <cmath>

namespace pf {


template <class C>
static int fir_dma_cfg(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen,
                       int lastOut, hipStream_t st, const FcBatch& fb) {
    ps = for_device(ps);
    typedef DmaGeom<C> G;
    auto k = fastconv_dma_kernel<C>;
    int rc = allow_big_lds(k, G::LDS_BYTES);
    if (rc) return rc;
    const size_t groups = ((size_t)nblk * fb.nsig + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus();
    if (grid > groups) grid = groups;
    unsigned* ctr = groups <= grid ? nullptr : take_counters(ps, st);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), G::LDS_BYTES, st, d_x, d_y, (const cx<float>*)d_Hc,
                       nblk, step, inputLen, lastOut, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, ctr,
                       fb.nsig, fb.xstride, fb.ystride);
    PF_CHECK(hipGetLastError());
    return 0;
}

// W_m^j (m = 1024, 512) for the wave-local sub-transforms of the split kernels: one table per device and length, generated in
// extended precision
static int split_sub_table(const cx<float>** out, int m = 1024) {
    static std::mutex mu;
    static std::map<long long, cx<float>*> tabs;
    int dev = 0;
    PF_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    const long long key = (long long)dev * 65536 + m;
    auto it = tabs.find(key);
    if (it == tabs.end()) {
        std::vector<cx<float>> tw(m);
        for (int j = 0; j < m; ++j) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)m;
            tw[j].x = (float)cosl(a); tw[j].y = (float)sinl(a);
        }
        cx<float>* d = nullptr;
        PF_CHECK(hipMalloc((void**)&d, sizeof(cx<float>) * m));
        PF_CHECK(hipMemcpy(d, tw.data(), sizeof(cx<float>) * m, hipMemcpyHostToDevice));
        it = tabs.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

template <int W>
static int fir_split1(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                      hipStream_t st, const FcBatch& fb, void** ab_cache) {
    typedef SplitOneT<W> S;
    ps = for_device(ps);
    auto k = fastconv_split1_kernel<W>;
    int rc = allow_big_lds(k, S::LDS_BYTES);
    if (rc) return rc;
    if (!*ab_cache) {
        // the folded coefficients of this filter: built ONCE and COMPLETE before the pointer is published - null stream + synchronisation,
        // like every other lazily built table (fc_ensure_big, split_sub_table).  Built on the caller's stream (round 4) a second call on
        // another non-blocking stream saw the pointer set and could read a half-written table: the setup's mutex orders hosts, not streams.
        void* ab = nullptr;
        PF_CHECK(hipMalloc(&ab, sizeof(float) * 4 * (size_t)S::n));
        hipLaunchKernelGGL(fastconv_split1_coef_kernel<W>, dim3(1), dim3(S::WG), 0, nullptr, (const cx<float>*)d_Hc, (const cx<float>*)ps->d_twr,
                           (vec4<float>*)ab);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) { (void)hipFree(ab); return fail(e, "fastconv_split1_coef_kernel"); }
        *ab_cache = ab;
    }
    const cx<float>* tw512 = nullptr;
    if ((rc = split_sub_table(&tw512, S::M))) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), S::WG, S::LDS_BYTES, &per_cu))) return rc;
    const size_t groups = (size_t)nblk * fb.nsig;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(S::WG), S::LDS_BYTES, st, d_x, d_y, (const vec4<float>*)*ab_cache, nblk, step, inputLen,
                       lastOut, (const cx<float>*)ps->d_tw, tw512, fb.nsig, fb.xstride, fb.ystride, env().fir_xcd);
    PF_CHECK(hipGetLastError());
    return 0;
}

// calls with few blocks on reference-sized blocks of 2 ps->n samples: W wavefronts x 512-point wave-local transforms (fft_split.h);
// -1: no such kernel for this length
int launch_fir_split1(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                      hipStream_t st, const FcBatch& fb, void** ab_cache) {
    switch (ps->n) {
        case 4096: return fir_split1<8>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, ab_cache);
        case 2048: return fir_split1<4>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, ab_cache);
        default: return -1;
    }
}

template <int PSYNC, int SPREAD, int W = 8>
static int fir_split(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen,
                     int lastOut, hipStream_t st, const FcBatch& fb) {
    typedef SplitFirT<W> S;
    ps = for_device(ps);
    auto k = fastconv_split_kernel<PSYNC, SPREAD, W>;
    int rc = allow_big_lds(k, S::LDS_BYTES);
    if (rc) return rc;
    const cx<float>* tw1024 = nullptr;
    if ((rc = split_sub_table(&tw1024))) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), S::WG, S::LDS_BYTES, &per_cu))) return rc;
    const size_t groups = (size_t)nblk * fb.nsig;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    // (the per-XCD ranges of the in-order case are pulled by the workgroups of that XCD only: every XCD needs one - ADVICE r05)
    const int xmode = (env().fir_xcd && grid >= 8 && groups < 0xfffffff0ull) ? 1 : 0;
    unsigned* ctr = groups <= grid ? nullptr : take_counters(ps, st, xmode ? 5 : 1);   // (per-XCD counters: nine words)
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(S::WG), S::LDS_BYTES, st, d_x, d_y, (const cx<float>*)d_Hc,
                       nblk, step, inputLen, lastOut, (const cx<float>*)ps->d_tw, tw1024, (const cx<float>*)ps->d_twr, ctr,
                       fb.nsig, fb.xstride, fb.ystride, xmode);
    PF_CHECK(hipGetLastError());
    return 0;
}

// 16384-sample blocks on 256 threads, 32 points per thread (fft_fir32.h).  hp_cache: the thread-major copy of the filter spectrum, built
// once per filter - on the null stream and COMPLETE before the pointer is published, like every lazily built table - and owned by
// the caller's pffastconv setup (hipFree)
template <int PREF>
static int fir32(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                 hipStream_t st, const FcBatch& fb, void** hp_cache) {
    ps = for_device(ps);
    auto k = fastconv_fused32_kernel<PREF>;
    int rc = allow_big_lds(k, Fir32::LDS_BYTES);
    if (rc) return rc;
    if (!*hp_cache) {
        void* hp = nullptr;
        PF_CHECK(hipMalloc(&hp, sizeof(float) * 4 * (size_t)Fir32::n));
        hipLaunchKernelGGL(fir32_coef_kernel, dim3(1), dim3(Fir32::WG), 0, nullptr, (const cx<float>*)d_Hc, (const cx<float>*)ps->d_twr, (vec4<float>*)hp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) { (void)hipFree(hp); return fail(e, "fir32_coef_kernel"); }
        *hp_cache = hp;
    }
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), Fir32::WG, Fir32::LDS_BYTES, &per_cu))) return rc;
    const size_t groups = (size_t)nblk * fb.nsig;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    // (the per-XCD ranges need every XCD to have a workgroup: ADVICE r05)
    const int xmode = (env().fir_xcd && grid >= 8 && groups < 0xfffffff0ull) ? 1 : 0;
    unsigned* ctr = groups <= grid ? nullptr : take_counters(ps, st, xmode ? 5 : 1);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(Fir32::WG), Fir32::LDS_BYTES, st, d_x, d_y, (const vec4<float>*)*hp_cache,
                       nblk, step, inputLen, lastOut, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, ctr,
                       fb.nsig, fb.xstride, fb.ystride, xmode);
    PF_CHECK(hipGetLastError());
    return 0;
}
int launch_fir32(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                 hipStream_t st, const FcBatch& fb, void** hp_cache, int pref) {
    if (ps->n != Fir32::n) return -1;
    if (pref == 2) return fir32<2>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, hp_cache);
#ifdef PFFFT_HIP_VARIANTS
    if (pref == 1) return fir32<1>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, hp_cache);
    if (pref == 0) return fir32<0>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, hp_cache);
#endif
    return fir32<2>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, hp_cache);
}

// the overlap-save block kernel on a real setup of length Nfft = 2 ps->n; -1 when the size has no DMA kernel
int launch_fir_dma(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                   hipStream_t st, const FcBatch& fb) {
    switch (ps->n) {
        case 2048: return fir_dma_cfg<DmaCfgF32::D2048>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        case 4096:
            // (the split kernel on FOUR wavefronts and 8192-sample blocks, two workgroups per CU - fft_split.h W = 4, tools/dma_timeline.hip -
            //  moves 6 % more samples per CU and second than the eight-wavefront one: not enough to pay for 75 % instead of 87 % overlap-save
            //  efficiency at 2048 taps (0.394 against 0.428); not instantiated here)
            return fir_dma_cfg<DmaCfgF32::D4096>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        case 8192:
            // 16384-sample blocks: cross-wave radix 8 + wave-local 1024-point transforms (fft_split.h).  Development build: AB_FIR_LOCKSTEP =
            // the lock-step LDS-DMA kernel it replaced, AB_FIR_SPLIT_PLAIN = without the pairwise flags and the spread pieces
#ifdef PFFFT_HIP_VARIANTS
            if (ab().is(AB_FIR_LOCKSTEP)) return fir_dma_cfg<DmaCfgF32::D8192>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            if (ab().is(AB_FIR_SPLIT_PLAIN)) return fir_split<0, 0>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
#endif
            return fir_split<1, 1>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        default: return -1;
    }
}


}  // namespace pf
