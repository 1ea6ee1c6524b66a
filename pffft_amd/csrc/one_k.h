// the kernels of fft_one.h, one translation unit per flag set (one_k<flags>_tu.hip)
#pragma once
namespace pf {
#define PF_ONE_K_DECL(F) const void* one_kernel_##F(bool is_double);
PF_ONE_K_DECL(0) PF_ONE_K_DECL(2) PF_ONE_K_DECL(4) PF_ONE_K_DECL(5) PF_ONE_K_DECL(8) PF_ONE_K_DECL(10) PF_ONE_K_DECL(12) PF_ONE_K_DECL(13)
#undef PF_ONE_K_DECL
#define PF_ONE_K_TU(F)                                                                                         \
    const void* one_kernel_##F(bool is_double) {                                                               \
        return is_double ? reinterpret_cast<const void*>(fft_one_kernel<double, F>)                            \
                         : reinterpret_cast<const void*>(fft_one_kernel<float, F>);                            \
    }
}  // namespace pf
