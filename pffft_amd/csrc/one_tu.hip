// libpffft_hip.so, translation unit of the single-image kernel (fft_one.h): its planner and launcher.
#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_one.h"
#include "one_k.h"

namespace pf {

// Plan of the single-image kernel for n complex points: 2-4 stages from the radices dftR has, the stage next to HBM on either side large,
// paddings from the bank model of stock_plan.h.  false: not a size for this kernel (it fits two images, or not even one, or has no plan).
bool one_build(int n, bool is_double, bool real, StockPlan out[2], size_t lds_max, bool narrow) {
    const int esz = is_double ? 16 : 8;
    const int nmax = is_double ? one_nmax<double>() : one_nmax<float>();
    if (n > nmax || n < 2048 || n % 16) return false;
    static const std::vector<int> setf = {32, 27, 25, 24, 16, 15, 12, 10, 9, 8, 6, 5, 4, 3};
    static const std::vector<int> setd = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3};
    // a radix is usable in a position where its n / R butterflies fit the trips the kernel is compiled for there (fft_one.h one_trips_of: the
    // first and last stage hold 48 / 24 complex operands per thread, a middle one 64 / 32); the stages next to the layout image exist for
    // radices from 8 (one_run).  Three stages where a triple fits (smallest radix sum; float: radices 20 and 30 join), else the fewest stages
    // from the radices that fit anywhere.
    auto fits = [&](int R, bool middle) { return n % R == 0 && n / R <= one_trips_of(is_double, R, middle) * ONE_WG; };
    std::vector<int> r;
    {
        std::vector<int> set, cur, best;
        for (int R : (is_double ? setd : setf)) if (fits(R, false)) set.push_back(R);
        sk_search(n, 0, set, cur, best);
        if (!narrow && (best.size() > 3 || best.size() < 2)) {
            // no plan within three stages on the end-stage budgets: a triple whose MIDDLE radix takes two trips (n = 14400 = 24 x 25 x 24 ... 18432 =
            // 24 x 32 x 24, float; four-stage plans measured 0.39-0.48 of the roofline where three-stage ones run 0.52-0.57)
            static const std::vector<int> tri_f = {32, 30, 27, 25, 24, 20, 16, 15, 12, 10, 9, 8};
            static const std::vector<int> tri_d = {16, 15, 12, 10, 9, 8};
            const std::vector<int>& S = is_double ? tri_d : tri_f;
            int best_sum = 1 << 30;
            for (int f : S) for (int m : S) {
                if (n % (f * m)) continue;
                const int l = n / (f * m);
                if (std::find(S.begin(), S.end(), l) == S.end()) continue;
                if (!fits(f, false) || !fits(m, true) || !fits(l, false)) continue;
                if (f == 20 || f == 30 || l == 20 || l == 30) continue;          // (radices 20 and 30 exist as middle stages only)
                if (f + m + l < best_sum && f >= l) { best_sum = f + m + l; r = {f, m, l}; }   // (the larger end radix first: it reads HBM)
            }
        }
        if (r.empty()) {
            if (best.size() < 2 || best.size() > (size_t)SK_MAX_STAGES) return false;
            // order: largest radix first (the stage that reads HBM), second largest last (the one that writes it), the rest ascending in between
            std::sort(best.begin(), best.end());
            r.push_back(best.back()); best.pop_back();
            const int last = best.back(); best.pop_back();
            for (int x : best) r.push_back(x);
            r.push_back(last);
        }
    }
    if (r.front() < 8 || r.back() < 8) return false;     // (the stages next to the layout image exist for radices from 8: fft_one.h one_run)
    for (int dir = 0; dir < 2; ++dir) {
        StockPlan& p = out[dir];
        memset(&p, 0, sizeof p);
        p.n = n; p.ns = (int)r.size(); p.G = 1; p.C = ONE_WG; p.P = 0;
        int ctab = 0;
        { int Ns = 1; for (int s = 0; s < p.ns; ++s) { if (s) ctab += Ns; Ns *= r[s]; } }
        const long long fixed = (long long)(ctab + 1) * esz + (real ? (long long)(64 + n / 128 + 2) * esz : 0) + 16;
        const long long spare = (long long)lds_max - (long long)n * esz - fixed - 64;
        if (spare < 0) return false;
        const int maxextra = (int)std::min<long long>(n / 4, spare / esz);
        int Ns = 1, img = n, prevpad = 0, toff = 0;
        for (int s = 0; s < p.ns; ++s) {
            StockStage& st = p.st[s];
            const int R = r[s], nb = n / R;
            int pad = 0;
            if (s + 1 < p.ns) {
                pad = sk_pick_pad(n, esz, 1, ONE_WG, Ns, R, r[s + 1], maxextra);
                img = std::max(img, n + (n / (Ns * R)) * pad);
            }
            st.R = R; st.nb = nb; st.Ns = Ns;
            st.rpad = prevpad;
            st.rstride = nb + (s ? (nb / Ns) * prevpad : 0);
            st.wblk = Ns * R + pad;
            st.twstep = n / (Ns * R);
            st.m_nb = sk_magic(nb); st.m_Ns = sk_magic(Ns);
            st.tw_off = toff; if (s) toff += Ns;
            prevpad = pad;
            Ns *= R;
        }
        p.ctab = ctab;
        p.img = (img + 3) / 2 * 2;
        p.twmode = 2;
        p.twr_lds = real ? 2 : 0;
        p.ibs = 32;
        p.m_n4 = sk_magic(n / 4); p.m_per = sk_magic(n / 2 + 1); p.m_nchk = sk_magic(n * esz / 16);
        const size_t tot = is_double ? one_lds<double>(p, real).total : one_lds<float>(p, real).total;
        if (tot > lds_max) return false;
    }
    return true;
}

size_t one_lds_bytes(const StockPlan& p, bool is_double, bool real) {
    return is_double ? one_lds<double>(p, real).total : one_lds<float>(p, real).total;
}

// flags: bit 0 input in the internal layout (backward unordered), bit 1 output in it (forward unordered), bit 2 backward, bit 3 real.
// One translation unit per flag set (one_k<flags>_tu.hip): they build in parallel, and sixteen instantiations in one module left 240-620 B of
// scratch per lane in kernels that have none when compiled by themselves.
template <typename T>
using OneFn = void (*)(const T*, T*, size_t, StockPlan, const cx<T>*, const cx<T>*, unsigned*);
template <typename T> static OneFn<T> one_kernel(int flags) {
    const void* f = nullptr;
    switch (flags) {
        case 0: f = one_kernel_0(sizeof(T) == 8); break;
        case 2: f = one_kernel_2(sizeof(T) == 8); break;
        case 4: f = one_kernel_4(sizeof(T) == 8); break;
        case 5: f = one_kernel_5(sizeof(T) == 8); break;
        case 8: f = one_kernel_8(sizeof(T) == 8); break;
        case 10: f = one_kernel_10(sizeof(T) == 8); break;
        case 12: f = one_kernel_12(sizeof(T) == 8); break;
        default: f = one_kernel_13(sizeof(T) == 8); break;
    }
    return reinterpret_cast<OneFn<T>>(const_cast<void*>(f));
}

template <typename T>
static int launch_one_t(Setup* s, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    const bool bwd = dir == PFFFT_BACKWARD, real = s->transform == PFFFT_REAL;
    const int flags = (real ? 8 : 0) | (bwd ? 4 : 0) | (!ordered ? (bwd ? 1 : 2) : 0);
    const int pi = (flags == 5 && sizeof(T) == 4) ? 2 : bwd ? 1 : 0;      // (the float complex backward transform from the layout: a plan of its own)
    const StockPlan& p = s->one[pi];
    const size_t lds = one_lds<T>(p, real).total;
    auto k = one_kernel<T>(flags);
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    unsigned long long grid = (unsigned long long)num_cus();
    if (grid > batch) grid = batch;
    // in order from the counter (one grab per 80-144 KiB vector); a launch that the resident workgroups cover in one go needs none
    unsigned* ctr = (batch > grid && batch < 0xfffffff0ull) ? take_counters(s, st, 1) : nullptr;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(ONE_WG), lds, st, in, out, batch, p, (const cx<T>*)(pi == 2 ? s->d_one_tw2 : s->d_twc[pi]),
                       (const cx<T>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

int launch_one(Setup* s, const void* in, void* out, size_t batch, int dir, int ordered, hipStream_t st) {
    if (s->is_double) return launch_one_t<double>(s, (const double*)in, (double*)out, batch, dir, ordered, st);
    return launch_one_t<float>(s, (const float*)in, (float*)out, batch, dir, ordered, st);
}

const void* one_kernel_ptr(bool is_double, int flags) {
    return is_double ? reinterpret_cast<const void*>(one_kernel<double>(flags)) : reinterpret_cast<const void*>(one_kernel<float>(flags));
}

}  // namespace pf
