// Lookups of the Stockham kernels instantiated on compile-time plans; defined in the generated translation units
// stock_ct_*_gen.hip (tools/gen_stock_plans.hip), one per precision and transform so that they build in parallel.
#pragma once
#include "fft_stock.h"

namespace pf {

template <typename T>
using StockCtFn = void (*)(const T*, T*, size_t, const cx<T>*, const cx<T>*, unsigned*, unsigned);

StockCtFn<float> stock_ct_lookup_f32c(const StockPlan& p, int flags, bool wl);
StockCtFn<float> stock_ct_lookup_f32r(const StockPlan& p, int flags, bool wl);
StockCtFn<double> stock_ct_lookup_f64c(const StockPlan& p, int flags, bool wl);
StockCtFn<double> stock_ct_lookup_f64r(const StockPlan& p, int flags, bool wl);

inline StockCtFn<float> stock_ct_lookup(const StockPlan& p, int flags, bool wl, const float*) {
    return (flags & 8) ? stock_ct_lookup_f32r(p, flags, wl) : stock_ct_lookup_f32c(p, flags, wl);
}
inline StockCtFn<double> stock_ct_lookup(const StockPlan& p, int flags, bool wl, const double*) {
    return (flags & 8) ? stock_ct_lookup_f64r(p, flags, wl) : stock_ct_lookup_f64c(p, flags, wl);
}

}  // namespace pf
