// libpffft_hip.so, translation unit of the fused spectral convolution kernels (fft_conv.h): instantiations + launcher.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_conv.h"

namespace pf {

template <typename T, class C>
static int conv_launch(Setup* s, const T* in, const T* H, T* out, size_t batch, T scaling, int accumulate, hipStream_t st) {
    const bool real = s->transform == PFFFT_REAL;
    void (*k)(const T*, const T*, T*, unsigned, T, int, const cx<T>*, const cx<T>*, unsigned*) =
        real ? fft_conv_kernel<C, 1> : fft_conv_kernel<C, 0>;
    int rc = allow_big_lds(k, C::LDS_BYTES);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), C::WG_THREADS, C::LDS_BYTES, &per_cu))) return rc;
    const size_t groups = (batch + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus() * per_cu;
    if (groups <= 4 * grid) grid = groups;        // (short launches: one group per workgroup in dispatch order - the rule of launch_tiled)
    if (grid > groups) grid = groups;
    unsigned* ctr = groups <= grid ? nullptr : take_counters(s, st);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), C::LDS_BYTES, st, in, H, out, (unsigned)batch, scaling, accumulate,
                       (const cx<T>*)s->d_tw, (const cx<T>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
static int conv_dispatch(Setup* s, const T* in, const T* H, T* out, size_t batch, T scaling, int accumulate, hipStream_t st) {
    typedef ConvPick<T> P;
    switch (s->n) {
        case 16: return conv_launch<T, typename P::C16>(s, in, H, out, batch, scaling, accumulate, st);
        case 32: return conv_launch<T, typename P::C32>(s, in, H, out, batch, scaling, accumulate, st);
        case 64: return conv_launch<T, typename P::C64>(s, in, H, out, batch, scaling, accumulate, st);
        case 128: return conv_launch<T, typename P::C128>(s, in, H, out, batch, scaling, accumulate, st);
        case 256: return conv_launch<T, typename P::C256>(s, in, H, out, batch, scaling, accumulate, st);
        case 512: return conv_launch<T, typename P::C512>(s, in, H, out, batch, scaling, accumulate, st);
        case 1024: return conv_launch<T, typename P::C1024>(s, in, H, out, batch, scaling, accumulate, st);
        case 2048: return conv_launch<T, typename P::C2048>(s, in, H, out, batch, scaling, accumulate, st);
        case 4096: return conv_launch<T, typename P::C4096>(s, in, H, out, batch, scaling, accumulate, st);
        case 8192: if constexpr (sizeof(T) == 4) return conv_launch<T, typename P::C8192>(s, in, H, out, batch, scaling, accumulate, st); break;
        default: break;
    }
    return -1;
}

// forward -> x H (one filter spectrum in the internal layout for the whole batch) -> backward in one kernel; -1: this size has no
// fused kernel (the caller composes the three launches)
int launch_conv_fused(Setup* s, const void* in, const void* H, void* out, size_t batch, double scaling, int accumulate, hipStream_t st) {
    if (s->kernel != K_TILED && s->kernel != K_C1024_F32) return -1;
    if (batch >= (1ull << 32)) return -1;
    if (s->is_double) return conv_dispatch<double>(s, (const double*)in, (const double*)H, (double*)out, batch, scaling, accumulate, st);
    return conv_dispatch<float>(s, (const float*)in, (const float*)H, (float*)out, batch, (float)scaling, accumulate, st);
}

}  // namespace pf
