// libpffft_hip.so, translation unit of the LDS-DMA staged kernels (fft_dma.h): instantiations + launchers.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_dma.h"
#include "fft_split.h"

namespace pf {

template <class C, int COUNTED>
static int launch_dma_cfg(Setup* s, const float* in, float* out, size_t batch, int dir, int ordered, hipStream_t st) {
    typedef DmaGeom<C> G;
    const int real = s->transform == PFFFT_REAL;
    void (*fn)(const float*, float*, unsigned, int, const cx<float>*, const cx<float>*, unsigned*);
    if (dir == PFFFT_FORWARD) fn = real ? fft_dma_kernel<C, FWD, 1, COUNTED> : fft_dma_kernel<C, FWD, 0, COUNTED>;
    else fn = real ? fft_dma_kernel<C, BWD, 1, COUNTED> : fft_dma_kernel<C, BWD, 0, COUNTED>;
    int rc = allow_big_lds(fn, G::LDS_BYTES);
    if (rc) return rc;
    const size_t groups = (batch + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus();
    if (grid > groups) grid = groups;
    const int flags = (((dir == PFFFT_BACKWARD) && !ordered) ? 1 : 0) | (((dir == PFFFT_FORWARD) && !ordered) ? 2 : 0);
    unsigned* ctr = groups <= grid ? nullptr : s->d_ctr + 2 * (s->ctr_slot.fetch_add(1) % CTR_RING);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(C::WG_THREADS), G::LDS_BYTES, st, in, out, (unsigned)batch, flags,
                       (const cx<float>*)s->d_tw, (const cx<float>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

// mode: 1 = counted vmcnt, 2 = vmcnt(0), 3 = vmcnt(0) + every twiddle in registers (n = 8192 only)
int launch_dma(Setup* s, const float* in, float* out, size_t batch, int dir, int ordered, hipStream_t st, int mode) {
    const bool counted = mode == 1;
    if (mode == 3 && s->n == 8192) return launch_dma_cfg<DmaCfgF32::D8192t0, 0>(s, in, out, batch, dir, ordered, st);
    switch (s->n) {
        case 2048: return counted ? launch_dma_cfg<DmaCfgF32::D2048, 1>(s, in, out, batch, dir, ordered, st) : launch_dma_cfg<DmaCfgF32::D2048, 0>(s, in, out, batch, dir, ordered, st);
        case 4096: return counted ? launch_dma_cfg<DmaCfgF32::D4096, 1>(s, in, out, batch, dir, ordered, st) : launch_dma_cfg<DmaCfgF32::D4096, 0>(s, in, out, batch, dir, ordered, st);
        case 8192: return counted ? launch_dma_cfg<DmaCfgF32::D8192, 1>(s, in, out, batch, dir, ordered, st) : launch_dma_cfg<DmaCfgF32::D8192, 0>(s, in, out, batch, dir, ordered, st);
        default: return -1;
    }
}

template <class C>
static int fir_dma_cfg(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen,
                       int lastOut, hipStream_t st, const FcBatch& fb) {
    typedef DmaGeom<C> G;
    auto k = fastconv_dma_kernel<C>;
    int rc = allow_big_lds(k, G::LDS_BYTES);
    if (rc) return rc;
    const size_t groups = ((size_t)nblk * fb.nsig + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus();
    if (grid > groups) grid = groups;
    unsigned* ctr = groups <= grid ? nullptr : ps->d_ctr + 2 * (ps->ctr_slot.fetch_add(1) % CTR_RING);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), G::LDS_BYTES, st, d_x, d_y, (const cx<float>*)d_Hc,
                       nblk, step, inputLen, lastOut, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, ctr,
                       fb.nsig, fb.xstride, fb.ystride);
    PF_CHECK(hipGetLastError());
    return 0;
}

// the overlap-save block kernel on a real setup of length Nfft = 2 ps->n; -1 when the size has no DMA kernel
int launch_fir_dma(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                   hipStream_t st, const FcBatch& fb) {
    switch (ps->n) {
        case 2048: return fir_dma_cfg<DmaCfgF32::D2048>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        case 4096: return fir_dma_cfg<DmaCfgF32::D4096>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        case 8192: return fir_dma_cfg<DmaCfgF32::D8192>(ps, d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
        default: return -1;
    }
}


int launch_split(Setup* s, const float* in, float* out, size_t batch, int dir, int ordered, hipStream_t st, int prefetch) {
    if (s->is_double || s->transform != PFFFT_REAL || s->n != SplitC3::n || dir != PFFFT_FORWARD) return -1;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->d_tw_sub) {
            std::vector<cx<float>> tw(SplitC3::M);
            for (int j = 0; j < SplitC3::M; ++j) {
                long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)SplitC3::M;
                tw[j].x = (float)cosl(a); tw[j].y = (float)sinl(a);
            }
            PF_CHECK(hipMalloc(&s->d_tw_sub, sizeof(cx<float>) * SplitC3::M));
            PF_CHECK(hipMemcpy(s->d_tw_sub, tw.data(), sizeof(cx<float>) * SplitC3::M, hipMemcpyHostToDevice));
        }
    }
    // one workgroup per CU, 180 VGPRs (two workgroups per CU need <= 128: 192 bytes of scratch per lane, 0.49-0.51)
    auto k = prefetch ? fft_split_real_fwd_kernel<1, 2> : fft_split_real_fwd_kernel<0, 2>;
    int rc = allow_big_lds(k, SplitC3::LDS_BYTES);
    if (rc) return rc;
    size_t grid = (size_t)num_cus();
    if (grid > batch) grid = batch;
    unsigned* ctr = batch <= grid ? nullptr : s->d_ctr + 2 * (s->ctr_slot.fetch_add(1) % CTR_RING);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(SplitC3::WG), SplitC3::LDS_BYTES, st, in, out, (unsigned)batch,
                       ordered ? 0 : 2, (const cx<float>*)s->d_tw, (const cx<float>*)s->d_tw_sub, (const cx<float>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

}  // namespace pf
