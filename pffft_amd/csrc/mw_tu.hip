// libpffft_hip.so, translation unit of the multi-wave register-tiled configurations (fft_tiled.h TiledMwF32: 1024 threads per
// vector, eight points per thread): instantiations + launcher, compiled on their own so that they can be iterated on in seconds.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_tiled.h"

namespace pf {

template <typename T, class C>
static int mw_launch(Setup* s, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    const bool real = s->transform == PFFFT_REAL, fwd = dir == PFFFT_FORWARD;
    void (*k)(const T*, T*, unsigned, int, const cx<T>*, const cx<T>*, unsigned*);
    if (fwd) k = real ? fft_tiled_kernel<C, FWD, 1> : fft_tiled_kernel<C, FWD, 0>;
    else k = real ? fft_tiled_kernel<C, BWD, 1> : fft_tiled_kernel<C, BWD, 0>;
    int rc = allow_big_lds(k, C::LDS_BYTES);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), C::WG_THREADS, C::LDS_BYTES, &per_cu))) return rc;
    const size_t groups = (batch + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    const int flags = ((!fwd && !ordered) ? 1 : 0) | ((fwd && !ordered) ? 2 : 0);
    unsigned* ctr = groups <= grid ? nullptr : take_counters(s, st);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), C::LDS_BYTES, st, in, out, (unsigned)batch, flags,
                       (const cx<T>*)s->d_tw, (const cx<T>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

// which: 0 = the adopted configuration; 1 .. = alternatives (development build / A/B).  -1: no such configuration for this size
int launch_tiled_mw(Setup* s, const void* in, void* out, size_t batch, int dir, int ordered, hipStream_t st, int which) {
    if (s->is_double) return -1;
    const float* i = (const float*)in;
    float* o = (float*)out;
    // measured slower than the adopted configurations everywhere: development build only, nothing routes here
#ifdef PFFFT_HIP_VARIANTS
    if (s->n == 8192) {
        switch (which) {
            case 0: return mw_launch<float, TiledMwF32::M8192>(s, i, o, batch, dir, ordered, st);
            case 1: return mw_launch<float, TiledMwF32::M8192np>(s, i, o, batch, dir, ordered, st);
            case 2: return mw_launch<float, TiledMwF32::M8192x2>(s, i, o, batch, dir, ordered, st);
            case 3: return mw_launch<float, TiledMwF32::M8192x2g>(s, i, o, batch, dir, ordered, st);
            case 4: return mw_launch<float, TiledMwF32::M8192x2p>(s, i, o, batch, dir, ordered, st);
            case 7: return mw_launch<float, TiledMwF32::M8192l>(s, i, o, batch, dir, ordered, st);
            default: return -1;
        }
    }
#else
    (void)i; (void)o; (void)batch; (void)dir; (void)ordered; (void)st; (void)which;
#endif
    return -1;
}

}  // namespace pf
