// libpffft_hip.so, translation unit of the tile passes with a run-time mixed-radix plan (fft_tileg.h): the plan of a tile length and the launch.
#include <map>
#include <mutex>

#include "tile_host.h"
#include "fft_tileg.h"

namespace pf {

static unsigned tg_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1); }   // x div d for x < 65536

// radices of a tile length: the fewest stages out of {12, 10, 9, 8, 6, 5, 4, 3, 2}; among those the plan with the largest smallest
// radix (balanced stages), the larger radices first (the last stage of a column pass carries the four-step twiddles: the smallest radix)
static bool tg_factor(int L, int maxst, int (&out)[TG_MAX_STAGES], int& ns) {
    static const int RAD[] = {12, 10, 9, 8, 6, 5, 4, 3, 2};
    int best[TG_MAX_STAGES], bn = 0, bmin = 0, cur[TG_MAX_STAGES];
    auto rec = [&](auto&& self, int rem, int depth, int first) -> void {
        if (rem == 1) {
            int mn = 1 << 30;
            for (int i = 0; i < depth; ++i) mn = cur[i] < mn ? cur[i] : mn;
            if (!bn || depth < bn || (depth == bn && mn > bmin)) { bn = depth; bmin = mn; for (int i = 0; i < depth; ++i) best[i] = cur[i]; }
            return;
        }
        if (depth == maxst || (bn && depth >= bn)) return;
        for (int ri = first; ri < (int)(sizeof(RAD) / sizeof(RAD[0])); ++ri)
            if (rem % RAD[ri] == 0) { cur[depth] = RAD[ri]; self(self, rem / RAD[ri], depth + 1, ri); }
    };
    rec(rec, L, 0, 0);
    if (!bn) return false;
    ns = bn;
    for (int i = 0; i < bn; ++i) out[i] = best[i];
    return true;
}

bool tile_gen_length_ok(int L, bool is_double) {
    if (L < 32 || L > TileGenGeom<float, 1024>::LMAX) return false;
    if (!is_double && (L & 1)) return false;                 // float: a 16-byte unit is two sequences of the OTHER pass
    int m = L;
    for (int q : {2, 3, 5}) while (m % q == 0) m /= q;
    if (m != 1) return false;
    int r[TG_MAX_STAGES], ns;
    return tg_factor(L, TG_MAX_STAGES, r, ns);
}

static const TileGenPlan* tg_plan(int L) {
    static std::mutex mu;
    static std::map<int, TileGenPlan> tab;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find(L);
    if (it == tab.end()) {
        TileGenPlan P{};
        int r[TG_MAX_STAGES], ns = 0;
        if (!tg_factor(L, TG_MAX_STAGES, r, ns)) return nullptr;
        P.L = L; P.ns = ns; P.m_L = tg_magic(L);
        int Ns = 1;
        for (int s = 0; s < ns; ++s) {
            P.R[s] = r[s]; P.nb[s] = L / r[s]; P.Ns[s] = Ns; P.tws[s] = L / (Ns * r[s]); P.m_Ns[s] = tg_magic(Ns);
            Ns *= r[s];
        }
        it = tab.emplace(L, P).first;
    }
    return &it->second;
}

template <typename T, int WG>
static int tile_gen_launch(const TileGenPlan& P, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                           bool out_int, bool in_int) {
    typedef TileGenGeom<T, WG> G;
    const size_t lds = G::lds_bytes(P.L, D.M > (1ull << (2 * G::WB)) ? 3 : 2);
    void (*k)(const cx<T>*, cx<T>*, unsigned long long, TileDesc, TileGenPlan, unsigned*);
    const bool fw = dir == PFFFT_FORWARD;
    if (D.seq_contig && in_int && !fw) k = tileg_kernel<T, WG, BWD, 1, 0, 1>;
    else if (D.seq_contig) k = fw ? tileg_kernel<T, WG, FWD, 1> : tileg_kernel<T, WG, BWD, 1>;
    else if (out_int && fw) k = tileg_kernel<T, WG, FWD, 0, 1, 0>;
    else k = fw ? tileg_kernel<T, WG, FWD, 0> : tileg_kernel<T, WG, BWD, 0>;
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), WG, lds, &per_cu))) return rc;
    unsigned long long grid = (unsigned long long)num_cus() * per_cu;
    const size_t tile_bytes = (size_t)P.L * G::C * sizeof(cx<T>);
    // (as tile_host.h: small tiles on a static stride, three per workgroup; tiles of 60 KiB and more in order from the counter)
    static const int its_env = dev_env("PFFFT_HIP_TILE_ITS", 3);
    if (its_env > 0 && tile_bytes < 60 * 1024) {
        const unsigned long long want = (ntiles + its_env - 1) / its_env;
        if (want > grid) grid = want;
    }
    if (grid > ntiles) grid = ntiles;
    const bool want_dyn = tile_bytes >= 60 * 1024;
    if (want_dyn && ntiles <= 4 * grid) grid = ntiles;     // (short launches of in-order tiles: one tile per workgroup, the rule of tile_host.h)
    static const int xctr_env = dev_env("PFFFT_HIP_TILE_XCTR", 1);
    const bool dynm = !(ntiles <= grid || !want_dyn || ntiles >= 0xfffffff0ull);
    const bool xctr = dynm && xctr_env && grid % 8 == 0 && ntiles >= 64;
    // (per-XCD counters: nine words = five {next, done} pairs of the ring, which is allocated with that much room past its end)
    unsigned* ctr = !dynm ? nullptr : take_counters(s, st, xctr ? 5 : 1);
    static const int xcd_env = dev_env("PFFFT_HIP_TILE_XCD", 1);
    // (the XCD-contiguous tile map of the static stride is a bijection of workgroup index to tile only on a grid of whole eights: rounded
    //  up, the workgroups beyond the tiles retire at once)
    if (!ctr && xcd_env) grid = (grid + 7) / 8 * 8;
    TileDesc D2 = D;
    D2.xmode = (xcd_env ? 1u : 0u) | (xctr ? 2u : 0u);     // (TileDesc::group keeps its one meaning: tiles per grab)
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(WG), lds, st, in, out, ntiles, D2, P, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

int tile_gen_pass(bool is_double, int L, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                  bool out_int, bool in_int) {
    const TileGenPlan* P = tg_plan(L);
    if (!P || L > TileGenGeom<float, 1024>::LMAX) { g_last_error = "pffft_hip: tile pass length out of range"; return (int)hipErrorInvalidValue; }
    if (((out_int && dir == PFFFT_FORWARD) || (in_int && dir == PFFFT_BACKWARD)) && (L % 4 != 0)) {
        g_last_error = "pffft_hip: internal layout on a tile length that is not a multiple of 4";
        return (int)hipErrorInvalidValue;
    }
    static const int wg128 = dev_env("PFFFT_HIP_TILE_WG128", 1);
    const int wg = (wg128 && L <= TileGenGeom<float, 128>::LMAX) ? 128 : L <= TileGenGeom<float, 256>::LMAX ? 256 : L <= TileGenGeom<float, 512>::LMAX ? 512 : 1024;
#define PF_TG(T, WG) tile_gen_launch<T, WG>(*P, (const cx<T>*)in, (cx<T>*)out, ntiles, D, dir, st, s, out_int, in_int)
    if (is_double) return wg == 128 ? PF_TG(double, 128) : wg == 256 ? PF_TG(double, 256) : wg == 512 ? PF_TG(double, 512) : PF_TG(double, 1024);
    return wg == 128 ? PF_TG(float, 128) : wg == 256 ? PF_TG(float, 256) : wg == 512 ? PF_TG(float, 512) : PF_TG(float, 1024);
#undef PF_TG
}

}  // namespace pf
