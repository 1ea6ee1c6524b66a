// libpffft_hip.so, translation unit of the LDS-DMA staged FIR block kernel (fft_dma.h): instantiations + launcher.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_dma.h"

namespace pf {

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


}  // namespace pf
