// libpffft_hip.so, translation unit of the two / three-pass tile kernels beyond LDS (fft_tile.h): the power-of-two tile
// lengths, the plans and the pass descriptors.  The tile lengths with an odd first stage are instantiated in tile_mr*_tu.hip.
#include <algorithm>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "tile_host.h"
#include "tile_plan_gen.h"

namespace pf {

static int g_tile_pp = dev_env("PFFFT_HIP_TILE_PP", 0);   // A/B: force 4 or 8
static int g_tile_pf = dev_env("PFFFT_HIP_TILE_PF", -1);  // A/B: prefetch off / on

template <typename T, int PP>
static int tile_dispatch(int logl, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                         bool out_int = false, bool in_int = false) {
    // (64-byte runs, PP = 4, for the lengths that have 128-byte runs: development build only - PFFFT_HIP_TILE_PP=4 A/B)
#ifdef PFFFT_HIP_VARIANTS
    constexpr bool SHORT = true;
#else
    constexpr bool SHORT = PP == 8;
#endif
    switch (logl) {
        case 6: if constexpr (SHORT) return tile_pass<T, 6, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int, g_tile_pf); break;
        case 7: if constexpr (SHORT) return tile_pass<T, 7, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int, g_tile_pf); break;
        case 8: if constexpr (SHORT) return tile_pass<T, 8, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int, g_tile_pf); break;
        case 9: if constexpr (SHORT) return tile_pass<T, 9, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int, g_tile_pf); break;
        case 10: if constexpr (PP == 4) return tile_pass<T, 10, 4>(in, out, ntiles, D, dir, st, s, out_int, in_int, g_tile_pf); break;
        default: break;
    }
    g_last_error = "pffft_hip: tile pass length out of range";
    return (int)hipErrorInvalidValue;
}

template <typename T> static int pick_pp(int logl) {
    // L = 1024: a padded image of 1024 x 144 bytes plus the tables does not fit LDS twice: 64-byte runs (PP = 4), three
    // workgroups of 512 threads per CU.  Measured in round 3 and dropped: the same tile with 128-byte runs in float (PP = 8:
    // 147 KiB image + W_L + three levels of 2^7 four-step twiddles = 158 KiB, ONE workgroup of 1024 threads per CU):
    // N = 2^20 0.197-0.199 against 0.206-0.239 - the denser runs do not pay for the lost overlap between workgroups.
    if (logl == 10) return 4;
#ifdef PFFFT_HIP_VARIANTS
    if (g_tile_pp == 4 || g_tile_pp == 8) return g_tile_pp;
#else
    (void)g_tile_pp;
#endif
    return 8;                                 // 128-byte runs: 64-byte runs measured 0.18 against 0.30 of the roofline
}

// pass A: `nvec` vectors of len = L x cols complex points; length-L transforms over the columns (stride cols), times
// W_len^(k col); same layout out (in place allowed)
// in_int (backward only, the first pass of a transform): the columns are read from the pffft-internal layout (tile_fft_kernel IINT)
// one tile length L = r0 2^logl (r0 = 1: power of two)
// (gen: a length of fft_tileg.h - run-time mixed-radix plan, r0 = L, logl = 0)
struct TileLen {
    int r0, logl;
    bool gen = false;
    bool alt = false;      // a run-time plan for a length that also has a register-tiled kernel: considered only where that kernel's stride is misaligned
    unsigned long long len() const { return (unsigned long long)r0 << logl; }
};

template <typename T>
static int tile_any(const TileLen& tl, int pp, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                    Setup* s, bool out_int, bool in_int) {
    if (tl.gen) return tile_gen_pass(sizeof(T) == 8, (int)tl.len(), in, out, ntiles, D, dir, st, s, out_int, in_int);
    switch (tl.r0) {
        case 1: return pp == 8 ? tile_dispatch<T, 8>(tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int)
                               : tile_dispatch<T, 4>(tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 3: return tile_mr_pass_3(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 5: return tile_mr_pass_5(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 9: return tile_mr_pass_9(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 15: return tile_mr_pass_15(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 25: return tile_mr_pass_25(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 27: return tile_mr_pass_27(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 45: return tile_mr_pass_45(sizeof(T) == 8, tl.logl, in, out, ntiles, D, dir, st, s, out_int, in_int);
        default: break;
    }
    g_last_error = "pffft_hip: tile pass radix out of range";
    return (int)hipErrorInvalidValue;
}

template <typename T> static int pick_pp(const TileLen& tl) { return (tl.r0 == 1 && !tl.gen) ? pick_pp<T>(tl.logl) : 8; }

template <typename T>
static int pass_columns(Setup* s, const cx<T>* in, cx<T>* out, unsigned long long nvec, TileLen tl, unsigned long long cols, int dir, hipStream_t st,
                        bool in_int = false) {
    const int pp = pick_pp<T>(tl), C = pp * TileUnit<T>::S;
    TileDesc D{};
    D.TA = (unsigned)((cols + C - 1) / C); D.TB = 1;
    D.last_units = (unsigned)((cols % C) / TileUnit<T>::S);      // ragged last tile (float, cols = 8 mod 16): 0 = none
    D.vstride = tl.len() * cols;
    D.in_a = C; D.out_a = C; D.ips = cols; D.iss = 1; D.ops = cols;
    D.col_a = (unsigned)C; D.M = D.vstride; D.seq_contig = 1;
    // (adjacent column tiles with 64-byte runs taken in PAIRS per grab - TileDesc::group = 2, built in round 4 against the 1.456 x traffic of
    //  this pass - measured slower, DESIGN.md appendix A.10: one tile per grab)
    D.group = 1u;
    const unsigned long long ntiles = nvec * D.TA;
    return tile_any<T>(tl, pp, in, out, ntiles, D, dir, st, s, false, in_int);
}

// pass B: rows of length L = 2^logl; row (vec, o, i) [o < outer, i < inner] sits at vec vlen + (o inner + i) L and its
// spectrum goes to X[vec vlen + (k inner + i) outer + o]   (outer index fastest: runs of C adjacent o)
// out_int (forward only): the spectrum is stored in the pffft-internal layout (tile_fft_kernel OINT)
template <typename T>
static int pass_rows(Setup* s, const cx<T>* in, cx<T>* out, unsigned long long nvec, TileLen tl, unsigned long long outer,
                     unsigned long long inner, int dir, hipStream_t st, bool out_int = false) {
    const int pp = pick_pp<T>(tl), C = pp * TileUnit<T>::S;
    const unsigned long long L = tl.len();
    TileDesc D{};
    D.TA = (unsigned)((outer + C - 1) / C); D.TB = (unsigned)inner;
    D.last_units = (unsigned)((outer % C) / TileUnit<T>::S);
    D.vstride = outer * inner * L;
    D.in_a = (unsigned long long)C * inner * L; D.in_b = L; D.ips = 1; D.iss = inner * L;
    D.out_a = C; D.out_b = outer; D.ops = outer * inner;
    D.M = 0; D.seq_contig = 0;
    const unsigned long long ntiles = nvec * D.TA * D.TB;
    return tile_any<T>(tl, pp, in, out, ntiles, D, dir, st, s, out_int, false);
}

// the LAST pass of a real forward transform on its complex core n = N1 L: row tiles closed under k1 -> N1 - k1, pair pass in the tile, canonical
// half-complex spectrum out (fft_tile.h RMODE 3)
template <typename T>
static int pass_rows_real(Setup* s, const cx<T>* in, cx<T>* out, unsigned long long nvec, TileLen tl, unsigned long long N1, hipStream_t st) {
    const int C = 8 * TileUnit<T>::S;
    const unsigned long long L = tl.len();
    TileDesc D{};
    D.TA = (unsigned)(N1 / C); D.TB = 1;
    D.vstride = N1 * L; D.ovstride = N1 * L;
    D.ips = 1; D.iss = L; D.ops = N1;
    D.M = 0;
    D.seq_contig = 0; D.rn1 = (unsigned)N1;
    return tile_any<T>(tl, 8, in, out, nvec * D.TA, D, PFFFT_FORWARD, st, s, false, false);
}
// can the last pass of n's two-pass plan be that one?  A register-tiled row length on 128-byte runs (not the L = 1024 geometry, not a run-time
// plan), whole mirror-closed tiles of the column length
template <typename T> static bool real_rows_ok(const TileLen& ta, const TileLen& tb) {
    const unsigned long long C = 8 * TileUnit<T>::S;
    return !tb.gen && pick_pp<T>(tb) == 8 && ta.len() % C == 0 && ta.len() >= 2 * C && (ta.len() * tb.len()) * 2 <= (1ull << 27);
}

// Two-pass plan of a size with factors 3 and / or 5: n = L1 L2, L = R0 2^b with R0 in {1, 3, 5, 9, 15, 25, 27, 45} and a tile
// length that is instantiated: power of two 64 .. 512, odd-stage lengths 48 .. 768 (tile_host.h: mr_min_logl / mr_max_logl; image
// <= 110 KiB).  Which length is the column pass and which the row pass is decided by the measured cost of each (us per GiB of
// vectors, float and double alike within 10 %, tile_len_times.py (earlier-round tool, git history) on MI355X): column tiles 105-125, but 165-180 from L = 640
// (ten and more wavefronts per workgroup: 168 registers); row tiles 97-127, 125-140 from L = 576.  Three streaming passes cost
// ~285 (five ~480 where the row length of that route is itself beyond LDS): plans above that are refused.  Tile lengths with two
// odd stages (25, 27, 45): columns 118-128 (L = 720: 176-192), rows 102-138 (L = 720: 165-169).  false: no plan - the three streaming passes of fft_big.h.
static int g_mr_min = dev_env("PFFFT_HIP_TILE_MRMIN", 48);    // A/B: shortest / longest
static int g_mr_max = dev_env("PFFFT_HIP_TILE_MRMAX", 768);   // tile length of a plan
static int g_wide_cost = dev_env("PFFFT_HIP_TILE_WIDECOST", 340);   // A/B: 0 = off
static const int g_gen_cost = env().tile_plans;   // PFFFT_HIP_TILE_PLANS=0: no run-time plans (fft_tileg.h) - those sizes take the streaming passes
// `stride`: the element stride between the points of the pass's strided side - the column count of a column pass (loads and stores), the
// row count (outer) of a row pass (stores).  Costs in the unit of the table above (~ us per 0.5 GiB of float vectors / 2.1), round 4,
// N = 10800 / 11664 / 250000 / 600000 on forced plans (r4_gen_force.sh (earlier-round tool, git history)):
//   run-time plans (fft_tileg.h): rows 230 us where the stride is whole 128-byte lines, 250-275 on half lines, 250-290 else (L > 432: 280-310);
//   columns 285-318 on half lines, 320-340 else, 414 for L = 60 (L > 432: 330-365);
//   register-tiled kernels on strides that are not half lines (float, the other length = 2, 4, 6 mod 8): 475-550 for either pass
static int tile_cost(const TileLen& t, bool columns, unsigned long long stride, bool is_double) {
    const long long L = t.len();
    const unsigned long long line = is_double ? 8 : 16;
    if (t.gen) {
        // (+ 10: a plan that ties with a register-tiled one on this model measured 1-9 % slower - N = 144000 .. 307200, r4_gen_scan.sh (earlier-round tool, git history) changed)
        const int dbl = (is_double ? 8 : 0) + 10;         // (double: 233-324 / 292-390 on the same plans)
        if (columns) return (L > 432 ? 165 : stride % (line / 2) == 0 ? 143 : 158) + (L < 80 ? 40 : 0) + dbl;
        return (L > 432 ? 140 : stride % line == 0 ? 110 : stride % (line / 2) == 0 ? 125 : 132) + (L < 80 ? 20 : 0) + dbl;
    }
    if (stride % (line / 2)) return 240;
    if (t.r0 == 1) return columns ? 103 : 97;
    // Register-tiled COLUMN tiles of L >= 576 (576, 640, 720, 768): a workgroup of L / 8 x 8 threads is nine to twelve wavefronts, three of them on
    // one SIMD - 168 VGPRs at most - and the kernels spill (-Rpass-analysis=kernel-resource-usage: 12-172 B of scratch per lane in the plain passes,
    // 152-364 B in the internal-layout input variant IINT, the first pass of a backward unordered transform, which then runs at HALF the rate:
    // N = 82944 = 576 x 144 double 0.30 / 0.31 / 0.30 / 0.19 over the four combinations, float 0.34 / 0.33 / 0.34 / 0.23).  A quarter of that pass's
    // extra cost, since all four combinations share one plan: N = 82944 double as 384 x 216 0.365 / 0.360 / 0.365 / 0.327, N = 294912 as 512 x 576
    // 0.328 / 0.324 / 0.330 / 0.318 (was 0.315 / 0.315 / 0.315 / 0.205; tools/r5_force_scan.sh, profiles/r05_tile_plans_576.txt)
    // (float 35: at 50 N = 460800 / 497664 would move from 576 x 800 / 576 x 864 to run-time column tiles - 0.31 / 0.31 / 0.31 / 0.22 -> 0.27 / 0.26 / 0.27 / 0.25)
    const int iint_dbl = (columns && L >= 576) ? (is_double ? 50 : 35) : 0;
    if (t.r0 >= 25) {                                    // two odd stages: one more exchange
        if (columns) return (L >= 600 ? 185 : 125) + iint_dbl;
        return L >= 600 ? 167 : L >= 256 ? 133 : 108;
    }
    if (columns) return (L >= 600 ? 172 : L <= 100 ? 125 : 115) + iint_dbl;
    return L >= 560 ? 135 : L >= 256 ? 122 : 105;
}
// cost factor in percent of a pass whose sequence count `seqs` is not a multiple of the tile width
static int ragged_pct(unsigned long long seqs, bool is_double) {
    const unsigned long long C = is_double ? 8 : 16;
    return (int)(100 * ((seqs + C - 1) / C * C) / seqs);
}
// the tile lengths of one precision, ascending
static const std::vector<TileLen>& tile_lengths(bool is_double) {
    static std::once_flag once;
    static std::vector<TileLen> f32, f64;
    std::call_once(once, [] {
        for (int dbl = 0; dbl < 2; ++dbl) {
            std::vector<TileLen>& v = dbl ? f64 : f32;
            for (int l = 6; l <= 9; ++l) v.push_back(TileLen{1, l});
            for (int r0 : {3, 5, 9, 15, 25, 27, 45})
                for (int l = mr_min_logl(r0, dbl != 0); l <= mr_max_logl(r0); ++l)
                    if ((r0 << l) >= 48) v.push_back(TileLen{r0, l});
            // every other length 2^a 3^b 5^c (float: even) up to 864 on the run-time plans of fft_tileg.h
            if (g_gen_cost > 0) {
                const size_t fixed = v.size();
                for (int L = 32; L <= 864; ++L) {
                    if (!tile_gen_length_ok(L, dbl != 0)) continue;
                    bool have = false;
                    for (size_t i = 0; i < fixed; ++i) have = have || (long long)v[i].len() == L;
                    static const int alt_env = dev_env("PFFFT_HIP_TILE_ALT", 1);   // A/B: 0 = off
                    if (!have || alt_env) v.push_back(TileLen{L, 0, true, have});
                }
            }
            std::sort(v.begin(), v.end(), [](const TileLen& x, const TileLen& y) { return x.len() < y.len(); });
        }
    });
    return is_double ? f64 : f32;
}
static bool tile_len_ok(const TileLen& t) { return t.gen || ((long long)t.len() >= g_mr_min && (long long)t.len() <= g_mr_max); }

// PFFFT_HIP_TILE_FORCE="L1[g],L2[g]" (A/B): the two tile lengths by value, g = the run-time plan even where a register-tiled kernel exists
static bool forced_lengths(long long n, bool is_double, TileLen& a, TileLen& b) {
    // parsed once: "L1[g],L2[g]" by hand (round 4 used sscanf("%d%c%d%c"), which stops at the comma behind a leading "g": a plan like
    // "162g,72g" was dropped silently and the production route ran in its place - ADVICE r04)
    struct Forced { bool set = false, ok = false; int l1 = 0, l2 = 0; bool g1 = false, g2 = false; };
    static const Forced F = [] {
        Forced f;
        const char* e = env().tile_force;
        if (!e || !*e) return f;
        f.set = true;
        char* end = nullptr;
        f.l1 = (int)strtol(e, &end, 10);
        if (end == e) return f;
        if (*end == 'g') { f.g1 = true; ++end; }
        if (*end != ',') return f;
        const char* p2 = end + 1;
        f.l2 = (int)strtol(p2, &end, 10);
        if (end == p2) return f;
        if (*end == 'g') { f.g2 = true; ++end; }
        f.ok = *end == 0 && f.l1 > 0 && f.l2 > 0;
        if (!f.ok) fprintf(stderr, "pffft_hip: PFFFT_HIP_TILE_FORCE=%s is not \"L1[g],L2[g]\": ignored\n", e);
        return f;
    }();
    if (!F.ok || (long long)F.l1 * F.l2 != n) return false;
    auto pick = [&](int L, bool gen, TileLen& t) {
        if (!gen)
            for (const TileLen& v : tile_lengths(is_double)) if ((long long)v.len() == L && !v.gen) { t = v; return true; }
        if (!tile_gen_length_ok(L, is_double)) return false;
        t = TileLen{L, 0, true};
        return true;
    };
    if (pick(F.l1, F.g1, a) && pick(F.l2, F.g2, b)) {
        // (the planner's own legality rule: the register-tiled double kernels are built without the ragged last tile - next to one of them the
        //  OTHER length is a multiple of 8; a forced 750 x 768 in double computed garbage before this check)
        if (!(is_double && ((!a.gen && b.len() % 8) || (!b.gen && a.len() % 8)))) return true;
    }
    static bool warned = false;
    if (!warned) { warned = true; fprintf(stderr, "pffft_hip: PFFFT_HIP_TILE_FORCE=%s names a tile length without a kernel: ignored\n", env().tile_force); }
    return false;
}

// is (columns ta, rows tb) a pair of tile lengths the kernels can run for n = ta tb?  (the rules of the search below, in one place: the
// measured plan table and the tuner's candidates obey them too)
static bool tile_pair_legal(long long n, bool is_double, const TileLen& ta, const TileLen& tb) {
    if (!tile_len_ok(ta) || !tile_len_ok(tb) || (long long)ta.len() * (long long)tb.len() != n) return false;
    // (double: the register-tiled kernels are built without the ragged last tile - the OTHER length must be a multiple of 8)
    if (is_double && ((!ta.gen && tb.len() % 8) || (!tb.gen && ta.len() % 8))) return false;
    // (a length with a register-tiled kernel runs on the run-time plan only where the strided 128-byte runs of that kernel would not be
    //  half lines - 475-550 us per pass against 250-330, r4_gen_force.sh (earlier-round tool, git history): N = 12000 = 100 x 120, 21600 = 108 x 200 ...)
    const unsigned long long half = is_double ? 4 : 8;
    if ((ta.alt && tb.len() % half == 0) || (tb.alt && ta.len() % half == 0)) return false;
    return true;
}
static int tile_pair_cost(bool is_double, const TileLen& ta, const TileLen& tb) {
    // (float: a length that is 8 mod 16 leaves the OTHER pass a half-empty last tile of 8 sequences)
    int c = tile_cost(ta, true, tb.len(), is_double) * ragged_pct(tb.len(), is_double) / 100 +
            tile_cost(tb, false, ta.len(), is_double) * ragged_pct(ta.len(), is_double) / 100;
    // (lengths that are not multiples of 4 cannot carry the internal layout: a reorder sweep, ~130, on the unordered half of the calls)
    if (ta.len() % 4 || tb.len() % 4) c += 40;
    return c;
}

// Plans that were MEASURED to beat the cost model's choice (tile_plan_gen.h, written by tools/tune_tile_plans.py on MI355X: every legal pair
// of every legal size beyond LDS timed in the four direction x layout combinations), and the tuner's run-time override of one size.
static std::mutex g_override_mu;
static std::map<long long, std::pair<TileLen, TileLen>> g_override;     // key n * 2 + is_double; first.r0 == 0: no tile plan
static bool find_len(bool is_double, int L, bool gen, TileLen& t) {
    for (const TileLen& v : tile_lengths(is_double)) if ((long long)v.len() == L && v.gen == gen) { t = v; return true; }
    return false;
}
// 1: a measured / overriding plan exists (a, b set), 0: that entry says "no tile plan", -1: no entry
static int measured_plan(long long n, bool is_double, TileLen& a, TileLen& b) {
    {
        std::lock_guard<std::mutex> lk(g_override_mu);
        auto it = g_override.find(n * 2 + (is_double ? 1 : 0));
        if (it != g_override.end()) {
            if (it->second.first.r0 == 0) return 0;
            a = it->second.first; b = it->second.second;
            return 1;
        }
    }
    for (const TilePlanEnt& e : kTilePlans)
        if (e.n == n && (e.is_double != 0) == is_double) {
            if (e.l1 == 0) return 0;
            if (find_len(is_double, e.l1, e.g1 != 0, a) && find_len(is_double, e.l2, e.g2 != 0, b) && tile_pair_legal(n, is_double, a, b)) return 1;
            return -1;       // (a table written for other tile lengths than this build has: the model decides)
        }
    return -1;
}

// mode: 0 = the streaming route of this size is three sweeps, complex transform; 1 = it is five (deep); 2 = three sweeps, the core of a REAL
// transform (no wide threshold: real N = 2n measured 1-4 % slower on those plans - the pair sweep dominates either way)
static bool tile_plan_search(long long n, bool is_double, int mode, TileLen& a, TileLen& b) {
    const bool deep = mode == 1;
    if (forced_lengths(n, is_double, a, b)) return true;
    const std::vector<TileLen>& V = tile_lengths(is_double);
    TileLen ma{1, 0}, mb{1, 0};
    const int measured = measured_plan(n, is_double, ma, mb);
    static const int maxcost_env = dev_env("PFFFT_HIP_TILE_MAXCOST", 286);   // A/B
    // one round over the pairs of tile lengths; with_alt: lengths that have a register-tiled kernel also on their run-time plan
    auto search = [&](bool with_alt) -> bool {
        int best = deep ? 460 : maxcost_env;             // (deep: the streaming route takes five sweeps, ~480)
        int wide_best = g_wide_cost;
        bool found = false, have_wide = false;
        TileLen wa{1, 0}, wb{1, 0};
        for (const TileLen& ta : V) {
            if (!tile_len_ok(ta) || n % (long long)ta.len()) continue;
            const long long L2 = n / (long long)ta.len();
            for (const TileLen& tb : V) {
                if ((long long)tb.len() != L2 || !tile_pair_legal(n, is_double, ta, tb)) continue;
                if ((ta.alt || tb.alt) && !with_alt) continue;
                const int c = tile_pair_cost(is_double, ta, tb);
                // (float, not deep: a plan with a run-time length that carries the internal layout is taken up to 340 - the sizes with 2^4 / 2^5
                //  and a large odd part, whose streaming route cannot read the internal layout in its column pass (R odd): four combinations
                //  0.23 / 0.24 / 0.25 / 0.18 -> 0.21 / 0.24 / 0.24 / 0.24, r4_gen_scan.sh (earlier-round tool, git history) wide; double: the run-time passes are 5-17 % behind)
                const bool wide = mode == 0 && !is_double && (ta.gen || tb.gen) && ta.len() % 4 == 0 && tb.len() % 4 == 0;
                if (c < best) { best = c; found = true; a = ta; b = tb; }
                else if (wide && c < wide_best) { wide_best = c; wa = ta; wb = tb; have_wide = true; }
            }
        }
        if (!found && have_wide) { a = wa; b = wb; found = true; }
        return found;
    };
    // the plans without `alt` lengths first; those with them only where that finds nothing (N = 12000 = 100 x 120 ... 200000: the minimum
    // over the four combinations 0.18 -> 0.25, mean +10 ... +17 %; as equal competitors they displaced better plans: N = 108000 -7 %)
    const bool found = search(false) || search(true);
    // a measured plan replaces the model's: for the complex transforms always (mode 0: including "the streaming passes win"), for the core of a
    // real transform (mode 2: the plans were timed on complex transforms) only WHICH tile lengths run where the model plans tiles at all
    if (measured == 1 && (mode != 2 || found)) { a = ma; b = mb; return true; }
    if (measured == 0 && mode == 0) return false;
    return found;
}

// Three tile passes n = L1 (L2 L3) (the shape of the power-of-two sizes beyond 2^20) for the sizes without a two-pass plan whose
// streaming route would take five sweeps (its row length is itself beyond LDS: `deep`); ~330 us per GiB against ~450-550.
static bool tile_plan3_search(long long n, bool is_double, TileLen& a, TileLen& b, TileLen& c) {
    const std::vector<TileLen>& V = tile_lengths(is_double);
    int best = 1 << 30;
    for (const TileLen& ta : V) {
        if (!tile_len_ok(ta) || n % (long long)ta.len()) continue;
        const long long rem = n / (long long)ta.len();
        for (const TileLen& tb : V) {
            if (!tile_len_ok(tb) || rem % (long long)tb.len()) continue;
            const long long L3 = rem / (long long)tb.len();
            for (const TileLen& tc : V) {
                if ((long long)tc.len() != L3 || !tile_len_ok(tc)) continue;
                if (is_double && ((!ta.gen && (tb.len() * tc.len()) % 8) || (!tb.gen && tc.len() % 8) || (!tc.gen && ta.len() % 8))) continue;
                if (ta.alt || tb.alt || tc.alt) continue;       // (three passes: the register-tiled kernel of a length where there is one)
                const int cst = tile_cost(ta, true, tb.len() * tc.len(), is_double) * ragged_pct(tb.len() * tc.len(), is_double) / 100 +
                                tile_cost(tb, true, tc.len(), is_double) * ragged_pct(tc.len(), is_double) / 100 +
                                tile_cost(tc, false, ta.len(), is_double) * ragged_pct(ta.len(), is_double) / 100;
                if (cst < best) { best = cst; a = ta; b = tb; c = tc; }
            }
        }
    }
    return best != (1 << 30);
}

// The plan of a size is a pure function of (n, precision, deep): searched once, then served from a table (launch_big asked
// twice per transform, under the setup's lock; the three-pass search walks ~26^3 length triples)
struct PlanEntry { int passes = 0; TileLen t[3] = {{1, 0}, {1, 0}, {1, 0}}; };
static std::mutex g_plan_mu;
static std::map<long long, PlanEntry> g_plan_tab;
static void drop_cached_plans(long long n, bool is_double) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (int mode = 0; mode < 3; ++mode) g_plan_tab.erase(n * 8 + (is_double ? 4 : 0) + mode);
}
static PlanEntry plan_of(long long n, bool is_double, int mode) {
    const bool deep = mode == 1;
    std::mutex& mu = g_plan_mu;
    std::map<long long, PlanEntry>& tab = g_plan_tab;
    const long long key = n * 8 + (is_double ? 4 : 0) + mode;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find(key);
    if (it == tab.end()) {
        PlanEntry e;
        if (tile_plan_search(n, is_double, mode, e.t[0], e.t[1])) e.passes = 2;
        else if (deep && n <= (1ll << 27) && tile_plan3_search(n, is_double, e.t[0], e.t[1], e.t[2])) e.passes = 3;
        it = tab.emplace(key, e).first;
    }
    return it->second;
}
static bool tile_plan(long long n, bool is_double, int mode, TileLen& a, TileLen& b) {
    const PlanEntry e = plan_of(n, is_double, mode);
    if (e.passes != 2) return false;
    a = e.t[0]; b = e.t[1];
    return true;
}
static bool tile_plan3(long long n, bool is_double, TileLen& a, TileLen& b, TileLen& c) {
    const PlanEntry e = plan_of(n, is_double, 1);
    if (e.passes != 3) return false;
    a = e.t[0]; b = e.t[1]; c = e.t[2];
    return true;
}

bool tile_has_plan(long long n, bool is_double, int mode) {
    const bool deep = mode == 1;
    if (n > 0 && (n & (n - 1)) == 0) return n >= (1 << 12) && n <= (1ll << 27);
    if (ab().is(AB_BIG_NO_MR_TILES)) return false;       // the streaming passes for these sizes (the second route of tests/test_gpu_round3.py)
    TileLen a, b, c;
    return tile_plan(n, is_double, mode, a, b) || (deep && n <= (1ll << 27) && tile_plan3(n, is_double, a, b, c));
}

// bit 0: the last pass can store the internal layout, bit 1: the first pass can read it.  The layout is blocks of four adjacent bins of
// the four quarters of the spectrum: the pass's tile length (the quarters) and its sequence count (the other lengths) are multiples of 4
int tile_plan_layouts(long long n, bool is_double, int mode) {
    const bool deep = mode == 1;
    if (n > 0 && (n & (n - 1)) == 0) return 3;
    const PlanEntry e = plan_of(n, is_double, mode);
    if (e.passes == 2) return ((e.t[1].len() % 4 || e.t[0].len() % 4) ? 0 : 1) | ((e.t[0].len() % 4 || e.t[1].len() % 4) ? 0 : 2);
    if (deep) {
        const PlanEntry e3 = plan_of(n, is_double, 1);
        if (e3.passes == 3)
            return ((e3.t[2].len() % 4 || e3.t[0].len() % 4 || e3.t[1].len() % 4) ? 0 : 1) | ((e3.t[0].len() % 4 || (e3.t[1].len() * e3.t[2].len()) % 4) ? 0 : 2);
    }
    return 0;
}

int tile_plan_lengths(long long n, bool is_double, int mode, int lengths[3]) {
    const bool deep = mode == 1;
    lengths[0] = lengths[1] = lengths[2] = 0;
    if (n <= 0) return 0;
    TileLen a, b, c;
    if ((n & (n - 1)) == 0) {
        int logn = 0;
        while ((1ll << logn) < n) ++logn;
        if (logn < 12 || logn > 27) return 0;
        if (logn <= 20) {
            TileLen ma{1, 0}, mb{1, 0};
            if (measured_plan(n, is_double, ma, mb) == 1) { lengths[0] = (int)ma.len(); lengths[1] = (int)mb.len(); return 2; }   // (a measured split)
            lengths[0] = 1 << (logn / 2); lengths[1] = 1 << (logn - logn / 2);
            return 2;
        }
        const int l1 = logn / 3, rem = logn - l1;
        lengths[0] = 1 << l1; lengths[1] = 1 << (rem / 2); lengths[2] = 1 << (rem - rem / 2);
        return 3;
    }
    if (ab().is(AB_BIG_NO_MR_TILES)) return 0;
    if (tile_plan(n, is_double, mode, a, b)) { lengths[0] = (int)a.len(); lengths[1] = (int)b.len(); return 2; }
    if (deep && n <= (1ll << 27) && tile_plan3(n, is_double, a, b, c)) {
        lengths[0] = (int)a.len(); lengths[1] = (int)b.len(); lengths[2] = (int)c.len();
        return 3;
    }
    return 0;
}

// canonical complex transform of `batch` vectors of n points: in -> out through ONE work buffer of the same size
// (in may equal out; work must differ from both).  Returns -1 when the size is outside the tile plans.
// out_int: forward transform straight into the internal layout (the last pass stores it: no reorder sweep)
template <typename T>
static int tile_fft(Setup* s, const cx<T>* in, cx<T>* work, cx<T>* out, size_t batch, long long n, int dir, hipStream_t st, bool out_int, bool in_int,
                    int mode, bool real_rows = false) {
    const bool deep = mode == 1;
    int rc;
    if (n & (n - 1)) {
        TileLen a, b, c;
        if (ab().is(AB_BIG_NO_MR_TILES)) return -1;
        if (tile_plan(n, sizeof(T) == 8, mode, a, b)) {
            if (real_rows && !real_rows_ok<T>(a, b)) return -1;
            if ((rc = pass_columns<T>(s, in, work, batch, a, b.len(), dir, st, in_int))) return rc;
            if (real_rows) return pass_rows_real<T>(s, work, out, batch, b, a.len(), st);
            return pass_rows<T>(s, work, out, batch, b, a.len(), 1, dir, st, out_int);
        }
        if (real_rows) return -1;
        if (!deep || n > (1ll << 27) || !tile_plan3(n, sizeof(T) == 8, a, b, c)) return -1;
        if ((rc = pass_columns<T>(s, in, work, batch, a, b.len() * c.len(), dir, st, in_int))) return rc;
        if ((rc = pass_columns<T>(s, work, work, batch * a.len(), b, c.len(), dir, st))) return rc;
        return pass_rows<T>(s, work, out, batch, c, a.len(), b.len(), dir, st, out_int);
    }
    int logn = 0;
    while ((1ll << logn) < n) ++logn;
    const int minlog = 12;
    if (logn < minlog || logn > 27) return -1;
    if (logn <= 20) {
        // the balanced split 2^(logn / 2) x 2^(logn - logn / 2), unless another split measured faster (tile_plan_gen.h: round 5)
        TileLen ta{1, logn / 2}, tb{1, logn - logn / 2};
        TileLen ma{1, 0}, mb{1, 0};
        if (measured_plan(n, sizeof(T) == 8, ma, mb) == 1) { ta = ma; tb = mb; }
        if (real_rows && !real_rows_ok<T>(ta, tb)) return -1;
        if ((rc = pass_columns<T>(s, in, work, batch, ta, tb.len(), dir, st, in_int))) return rc;
        if (real_rows) return pass_rows_real<T>(s, work, out, batch, tb, ta.len(), st);
        return pass_rows<T>(s, work, out, batch, tb, ta.len(), 1, dir, st, out_int);
    }
    if (real_rows) return -1;
    const int l1 = logn / 3, rem = logn - l1, l2 = rem / 2, l3 = rem - l2;
    // n = L1 n', n' = L2 L3:  A over L1 (columns n'), then per row of length n': A over L2 (in place), B over L3 with the
    // scatter X[(k3 L2 + k2) L1 + k1]
    if ((rc = pass_columns<T>(s, in, work, batch, TileLen{1, l1}, 1ull << rem, dir, st, in_int))) return rc;
    if ((rc = pass_columns<T>(s, work, work, batch << l1, TileLen{1, l2}, 1ull << l3, dir, st))) return rc;
    return pass_rows<T>(s, work, out, batch, TileLen{1, l3}, 1ull << l1, 1ull << l2, dir, st, out_int);
}

// every legal pair of tile lengths of n (columns, rows) with the model's cost: out[5 i ..] = {L1, gen1, L2, gen2, cost}; returns the count
int tile_plan_candidates(long long n, bool is_double, int* out, int max) {
    const std::vector<TileLen>& V = tile_lengths(is_double);
    int cnt = 0;
    for (const TileLen& ta : V) {
        if (n % (long long)ta.len()) continue;
        for (const TileLen& tb : V) {
            if (!tile_pair_legal(n, is_double, ta, tb)) continue;
            if (cnt < max) {
                out[5 * cnt] = (int)ta.len(); out[5 * cnt + 1] = ta.gen ? 1 : 0; out[5 * cnt + 2] = (int)tb.len(); out[5 * cnt + 3] = tb.gen ? 1 : 0;
                out[5 * cnt + 4] = tile_pair_cost(is_double, ta, tb);
            }
            ++cnt;
        }
    }
    return cnt;
}
// the tuner's hook: l1 > 0 - the plan of n from now on (0 = accepted, -1 = not a legal pair); l1 == 0 - "no tile plan"; l1 < 0 - remove the
// override.  Set it BEFORE any setup of n exists (the tuner creates a fresh setup per candidate): a live setup's route - which pass reads /
// stores the internal layout, the sweeps - was frozen at pffft_new_setup for the lengths of that time, while the tile passes look the
// lengths up on every launch; a live setup of n would run the new lengths under its old route.  The cached model plans of n
// are dropped.
int tile_plan_override(long long n, bool is_double, int l1, int g1, int l2, int g2) {
    const long long key = n * 2 + (is_double ? 1 : 0);
    TileLen a{0, 0}, b{0, 0};
    if (l1 > 0 && !(find_len(is_double, l1, g1 != 0, a) && find_len(is_double, l2, g2 != 0, b) && tile_pair_legal(n, is_double, a, b))) return -1;
    {
        std::lock_guard<std::mutex> lk(g_override_mu);
        if (l1 < 0) g_override.erase(key);
        else g_override[key] = std::make_pair(a, b);
    }
    drop_cached_plans(n, is_double);
    return 0;
}

// layout: 1 = forward, spectrum out in the internal layout; 2 = backward, spectrum in from the internal layout
int launch_tile_fft(Setup* s, const void* in, void* work, void* out, size_t batch, long long n, int dir, hipStream_t st, int layout, int mode) {
    const bool out_int = layout == 1, in_int = layout == 2, real_rows = layout == 3;
    if ((out_int && dir != PFFFT_FORWARD) || (in_int && dir != PFFFT_BACKWARD) || (real_rows && dir != PFFFT_FORWARD)) return -1;
    if (s->is_double) return tile_fft<double>(s, (const cx<double>*)in, (cx<double>*)work, (cx<double>*)out, batch, n, dir, st, out_int, in_int, mode, real_rows);
    return tile_fft<float>(s, (const cx<float>*)in, (cx<float>*)work, (cx<float>*)out, batch, n, dir, st, out_int, in_int, mode, real_rows);
}

// does the two-pass plan of the complex core n of a real transform end in a row pass that can carry the pair pass (layout 3)?
bool tile_real_rows_plan(long long n, bool is_double, int mode) {
    if (n <= 0) return false;
    TileLen a{1, 0}, b{1, 0};
    if ((n & (n - 1)) == 0) {
        int logn = 0;
        while ((1ll << logn) < n) ++logn;
        if (logn < 12 || logn > 20) return false;
        a = TileLen{1, logn / 2}; b = TileLen{1, logn - logn / 2};
        TileLen ma{1, 0}, mb{1, 0};
        if (measured_plan(n, is_double, ma, mb) == 1) { a = ma; b = mb; }
    } else {
        if (ab().is(AB_BIG_NO_MR_TILES) || !tile_plan(n, is_double, mode, a, b)) return false;
    }
    return is_double ? real_rows_ok<double>(a, b) : real_rows_ok<float>(a, b);
}

}  // namespace pf
