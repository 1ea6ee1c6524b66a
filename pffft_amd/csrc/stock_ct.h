// Lookups of the Stockham kernels instantiated on compile-time plans; defined in the generated translation units
// stock_ct_*_gen.hip (tools/gen_stock_plans.hip), two per precision and transform so that they build in parallel.
#pragma once
#include "fft_stock.h"

namespace pf {

template <typename T>
using StockCtFn = void (*)(const T*, T*, size_t, const cx<T>*, const cx<T>*, unsigned*, unsigned);

// (two translation units per precision and transform, the sizes alternating between them)
#define PF_CT_DECL(T, tag)                                                                  \
    StockCtFn<T> stock_ct_lookup_##tag##_a(const StockPlan& p, int flags, bool wl);         \
    StockCtFn<T> stock_ct_lookup_##tag##_b(const StockPlan& p, int flags, bool wl);         \
    inline StockCtFn<T> stock_ct_lookup_##tag(const StockPlan& p, int flags, bool wl) {     \
        StockCtFn<T> f = stock_ct_lookup_##tag##_a(p, flags, wl);                           \
        return f ? f : stock_ct_lookup_##tag##_b(p, flags, wl);                             \
    }
PF_CT_DECL(float, f32c)
PF_CT_DECL(float, f32r)
PF_CT_DECL(double, f64c)
PF_CT_DECL(double, f64r)
#undef PF_CT_DECL

inline StockCtFn<float> stock_ct_lookup(const StockPlan& p, int flags, bool wl, const float*) {
    return (flags & 8) ? stock_ct_lookup_f32r(p, flags, wl) : stock_ct_lookup_f32c(p, flags, wl);
}
inline StockCtFn<double> stock_ct_lookup(const StockPlan& p, int flags, bool wl, const double*) {
    return (flags & 8) ? stock_ct_lookup_f64r(p, flags, wl) : stock_ct_lookup_f64c(p, flags, wl);
}

}  // namespace pf
