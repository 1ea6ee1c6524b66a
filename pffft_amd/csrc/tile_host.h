// Host side of one tile pass (fft_tile.h): kernel choice, LDS opt-in, grid, in-order counter.  Shared by tile_tu.hip (power-of-two
// tile lengths) and tile_mr*_tu.hip (tile lengths R0 2^b with an odd first stage) so that the instantiations compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_tile.h"

namespace pf {

// PFSEL: -1 = both prefetch variants are instantiated and chosen at run time (power-of-two lengths, PFFFT_HIP_TILE_PF A/B);
//         0 / 1 = only that one (odd-stage lengths: the choice is a function of the geometry, half the kernels)
template <typename T, int LOGL, int PP, int R0, int PFSEL, int PFSEL_B, int RAG>
static int tile_pass_impl(const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                          bool out_int, bool in_int, int pf_force) {
    typedef TileGeom<T, LOGL, PP, R0> G;
    const bool fw0 = dir == PFFFT_FORWARD;
    // the kernels that keep an unpadded image (fft_tile.h IMG_BYTES_PLAIN): column passes off the canonical layout, and - round 6 - every pass
    // of the swizzled L = 1024 geometry
    const bool plain = G::SWZ || (D.seq_contig && !(in_int && !fw0));
    if (D.M > (1ull << (3 * G::WB))) { g_last_error = "pffft_hip: four-step modulus beyond the tile's twiddle table"; return (int)hipErrorInvalidValue; }
    const size_t lds = G::lds_bytes(D.M > (1ull << (2 * G::WB)) ? 3 : 2, plain);
    void (*k)(const cx<T>*, cx<T>*, unsigned long long, TileDesc, unsigned*);
    const bool fw = dir == PFFFT_FORWARD;
    if constexpr (PFSEL < 0) {
        // register prefetch of the next tile where one or two workgroups fill a CU (images of 40 KiB and more)
        const bool pf = pf_force >= 0 ? pf_force != 0 : lds > 40 * 1024;
        if (D.seq_contig && in_int && !fw) k = pf ? tile_fft_kernel<T, LOGL, PP, BWD, 1, 1, 0, 1, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 0, 0, 1, R0, RAG>;
        else if (D.seq_contig) k = pf ? (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 1, 1, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 1, 0, 0, R0, RAG>)
                                 : (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 1, 0, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 0, 0, 0, R0, RAG>);
        else if (out_int && fw) k = pf ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 1, 1, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, FWD, 0, 0, 1, 0, R0, RAG>;
        else k = pf ? (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 1, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 0, 1, 0, 0, R0, RAG>)
                    : (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 0, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 0, 0, 0, 0, R0, RAG>);
    } else {
        constexpr int PF = PFSEL, PFB = PFSEL_B;
        if (D.seq_contig && in_int && !fw) k = tile_fft_kernel<T, LOGL, PP, BWD, 1, PF, 0, 1, R0, RAG>;
        else if (D.seq_contig) k = fw ? tile_fft_kernel<T, LOGL, PP, FWD, 1, PF, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 1, PF, 0, 0, R0, RAG>;
        else if (out_int && fw) k = tile_fft_kernel<T, LOGL, PP, FWD, 0, PFB, 1, 0, R0, RAG>;
        else k = fw ? tile_fft_kernel<T, LOGL, PP, FWD, 0, PFB, 0, 0, R0, RAG> : tile_fft_kernel<T, LOGL, PP, BWD, 0, PFB, 0, 0, R0, RAG>;
    }
    // the last pass of a real forward transform: mirror-closed row tiles with the pair pass inside (fft_tile.h RMODE 3; TileDesc::rn1 set)
    if (D.rn1 && !D.seq_contig) {
        if constexpr (PP == 8 && RAG == 0) {
            if (!fw || out_int || in_int) { g_last_error = "pffft_hip: the fused real row pass is a forward pass into the canonical spectrum"; return (int)hipErrorInvalidValue; }
            if constexpr (PFSEL < 0) {
                const bool pf = pf_force >= 0 ? pf_force != 0 : lds > 40 * 1024;
                k = pf ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 1, 0, 0, R0, 0, 3> : tile_fft_kernel<T, LOGL, PP, FWD, 0, 0, 0, 0, R0, 0, 3>;
            } else {
                k = tile_fft_kernel<T, LOGL, PP, FWD, 0, PFSEL_B, 0, 0, R0, 0, 3>;
            }
        } else {
            g_last_error = "pffft_hip: no fused real row pass for this tile geometry";
            return (int)hipErrorInvalidValue;
        }
    }
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), G::WG, lds, &per_cu))) return rc;
    unsigned long long grid = (unsigned long long)num_cus() * per_cu;
    const unsigned long long ngroups = D.group > 1 ? (ntiles + D.group - 1) / D.group : ntiles;   // the counter hands out groups of tiles
    // Static-stride passes (tiles below 60 KiB) launch ntiles / 3 workgroups - three tiles each - instead of the resident set (round 4:
    // what the static-stride Stockham kernels showed, tools/tune_stock_grid.py, holds here too - N = 2^15 complex float 0.31-0.33 ->
    // 0.33-0.37, N = 15360 0.31 -> 0.36, 61440 0.33 -> 0.36, double 2^15 0.31-0.33 -> 0.34-0.36 at 2 .. 4 tiles per workgroup, back
    // to the old figures from 16 on).  PFFFT_HIP_TILE_ITS=<k> sets the count, 0 = the resident set (A/B)
    // ... but only for tiles up to 32 KiB: steady scans over every size (profiles/r04_scan_beyond_lds_*.txt against r03's) showed the plans with
    // a 36-54 KiB column tile (L = 288 ... 432) 9-15 % SLOWER at three tiles per workgroup than on the resident set - their prefetch pipeline
    // wants a long run of tiles - and the L = 144 tiles (radix 9 first, 144 threads: columns -11 ... -15 %, rows -4 %) likewise; L = 64 ... 256 gain 7-18 %
    static const int its_env = dev_env("PFFFT_HIP_TILE_ITS", 3);
    const bool its_ok = (size_t)G::L * G::C * sizeof(cx<T>) <= 32 * 1024 && !(R0 == 9 && LOGL == 4);
    if (its_env > 0 && its_ok) {
        const unsigned long long want = (ntiles + its_env - 1) / its_env;
        if (want > grid) grid = want;
    }
    if (grid > ngroups) grid = ngroups;
    // in-order tiles only where a tile is 64 KiB or more: one counter address serves ~80 M atomics/s, so 16-32 KiB tiles
    // are throttled by the grab (2^15: 0.30 static, 0.20 in order; 2^18 .. 2^20: 0.27-0.31 / 0.19 static, 0.29-0.32 / 0.24 in
    // order).  PFFFT_HIP_TILE_DYN=0/1 forces it (A/B).
    static const int dyn_env = dev_env("PFFFT_HIP_TILE_DYN", -1);
    const bool want_dyn = dyn_env >= 0 ? dyn_env != 0 : (size_t)G::L * G::C * sizeof(cx<T>) >= 60 * 1024;   // (L = 480: 60 KiB)
    // (in-order tiles: a launch of up to four tiles per resident workgroup runs one tile per workgroup in dispatch order - N = 2^18 complex at 32 MiB
    //  of vectors 56 -> 38 us per transform, tools/r4_small_batch.py; launch_tiled has the rule's measurements.  PFFFT_HIP_ONESHOT=<k>, 0 = off)
    const unsigned long long oneshot_env = (unsigned long long)env().oneshot;
    if (want_dyn && oneshot_env && ngroups <= oneshot_env * grid) grid = ngroups;
    // XCD-aware tile order (TileDesc::xmode, round 4): PFFFT_HIP_TILE_XMODE = 0 off, 1 static map only, 2 per-XCD counters only, 3 both (A/B).
    // OFF for these kernels: their strides are whole or half lines, and measured (r4_xmode.sh (earlier-round tool, git history)) the static map costs 0-4 %, the per-XCD
    // counters N = 2^20 0.20-0.23 -> 0.17-0.19 (one in-order sweep over the whole batch is what HBM rewards there); they pay only on the
    // strides of fft_tileg.h that are neither
    // Round 6, bit 4 (on): the column passes with 64-byte runs (PP = 4: two adjacent tiles share every 128-byte line) take their tiles from
    // per-XCD counters that deal tile PAIRS round robin (fft_tile.h xpair) - both tiles of a line through one L2 while all XCDs sweep the batch
    // together: N = 2^20 complex float 0.239 -> 0.245, double 0.250 -> 0.262, 2^19 +-1 % (tools/r6_quick.py, same box, alternating)
    static const int xmode_env = dev_env("PFFFT_HIP_TILE_XMODE", 16);
    const bool dynm = !(ngroups <= grid || !want_dyn || ntiles >= 0xfffffff0ull);
    // (bit 2: per-XCD counters for the column passes with 64-byte runs only - adjacent tiles share every line there)
    const bool xpair = (xmode_env & 16) && PP == 4 && D.seq_contig;      // pairs of column tiles with 64-byte runs dealt per XCD (fft_tile.h)
    const bool xctr = dynm && ((xmode_env & 2) || ((xmode_env & 4) && PP == 4 && D.seq_contig) || xpair || (D.rn1 && !D.seq_contig)) && grid % 8 == 0 && ngroups >= 64 && D.group <= 1;
    // (per-XCD counters: nine words = five {next, done} pairs of the ring, which is allocated with that much room past its end)
    unsigned* ctr = !dynm ? nullptr : take_counters(s, st, xctr ? 5 : 1);
    TileDesc D2 = D;
    static const int cstart_env = dev_env("PFFFT_HIP_TILE_CSTART", 0);   // A/B: start-up grabs from the counter
    // (the fused real row pass: the mirror runs of neighbouring tiles share their lines - XCD-contiguous tiles, static map or per-XCD counters)
    const bool rrows = D.rn1 && !D.seq_contig;
    D2.xmode = (xctr ? 2u : 0u) | ((xctr && xpair) ? 4u : 0u) | ((!dynm && ((xmode_env & 1) || rrows) && D.group <= 1) ? 1u : 0u) | (cstart_env ? 8u : 0u);
    if (D2.xmode & 1u) grid = (grid + 7) / 8 * 8;      // (the static map is a bijection on a grid of whole eights; the surplus workgroups retire at once)
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(G::WG), lds, st, in, out, ntiles, D2, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T, int LOGL, int PP, int R0 = 1, int PFSEL = -1, int PFSEL_B = PFSEL>
static int tile_pass(const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                     bool out_int = false, bool in_int = false, int pf_force = -1) {
    // a ragged last tile exists in float only (a double tile is 8 sequences, every tile length a multiple of 8)
    if constexpr (sizeof(T) == 4) {
        if (D.last_units) return tile_pass_impl<T, LOGL, PP, R0, PFSEL, PFSEL_B, 1>(in, out, ntiles, D, dir, st, s, out_int, in_int, pf_force);
    }
    return tile_pass_impl<T, LOGL, PP, R0, PFSEL, PFSEL_B, 0>(in, out, ntiles, D, dir, st, s, out_int, in_int, pf_force);
}

// tile lengths with an odd first stage: L = R0 2^logl, 128-byte runs (PP = 8)
template <typename T, int LOGL, int R0>
static int tile_pass_mr(const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                        bool out_int, bool in_int) {
    // always with the prefetch of the next tile: measured 96-270 us -> 95-130 us per GiB on the short row tiles (L <= 192), 120-217 -> 105 on the
    // short column tiles in double
    return tile_pass<T, LOGL, 8, R0, 1, 1>(in, out, ntiles, D, dir, st, s, out_int, in_int);
}

// the odd-stage tile lengths that are instantiated: L = R0 2^logl, logl in [mr_min_logl(R0), mr_max_logl(R0)]  (48 <= L <= 768:
// image <= 110 KiB).  A tile is C = 16 (float) / 8 (double) adjacent columns or rows; every tile length is a multiple of 8, and in
// float a length that is 8 mod 16 leaves the other pass a ragged last tile of 8 sequences (TileDesc::last_units).
constexpr int mr_max_logl(int r0) { return r0 == 3 ? 8 : r0 == 5 ? 7 : r0 == 9 ? 6 : r0 == 15 ? 5 : (r0 == 25 || r0 == 27 || r0 == 45) ? 4 : 0; }
constexpr int mr_min_logl(int r0, bool) { return r0 >= 9 ? 3 : 4; }

template <typename T, int R0>
static int tile_dispatch_mr(int logl, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                            Setup* s, bool out_int, bool in_int) {
    switch (logl) {
        case 3: if constexpr (mr_min_logl(R0, sizeof(T) == 8) == 3) return tile_pass_mr<T, 3, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int); break;
        case 4: return tile_pass_mr<T, 4, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 5: if constexpr (mr_max_logl(R0) >= 5) return tile_pass_mr<T, 5, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int); break;
        case 6: if constexpr (mr_max_logl(R0) >= 6) return tile_pass_mr<T, 6, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int); break;
        case 7: if constexpr (mr_max_logl(R0) >= 7) return tile_pass_mr<T, 7, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int); break;
        case 8: if constexpr (mr_max_logl(R0) >= 8) return tile_pass_mr<T, 8, R0>(in, out, ntiles, D, dir, st, s, out_int, in_int); break;
        default: break;
    }
    g_last_error = "pffft_hip: tile pass length out of range";
    return (int)hipErrorInvalidValue;
}

// tileg_tu.hip: tile lengths on a run-time mixed-radix plan (fft_tileg.h), canonical layouts
int tile_gen_pass(bool is_double, int L, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                  bool out_int, bool in_int);
bool tile_gen_length_ok(int L, bool is_double);

// one entry per odd radix, each in its own translation unit (tile_mr<R0>_tu.hip)
int tile_mr_pass_3(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                   Setup* s, bool out_int, bool in_int);
int tile_mr_pass_5(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                   Setup* s, bool out_int, bool in_int);
int tile_mr_pass_9(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                   Setup* s, bool out_int, bool in_int);
int tile_mr_pass_15(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                    Setup* s, bool out_int, bool in_int);
int tile_mr_pass_25(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                    Setup* s, bool out_int, bool in_int);
int tile_mr_pass_27(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                    Setup* s, bool out_int, bool in_int);
int tile_mr_pass_45(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st,
                    Setup* s, bool out_int, bool in_int);

#define PF_TILE_MR_TU(R0)                                                                                                              \
    int tile_mr_pass_##R0(bool is_double, int logl, const void* in, void* out, unsigned long long ntiles, const TileDesc& D, int dir,    \
                          hipStream_t st, Setup* s, bool out_int, bool in_int) {                                                        \
        if (is_double) return tile_dispatch_mr<double, R0>(logl, (const cx<double>*)in, (cx<double>*)out, ntiles, D, dir, st, s, out_int, in_int); \
        return tile_dispatch_mr<float, R0>(logl, (const cx<float>*)in, (cx<float>*)out, ntiles, D, dir, st, s, out_int, in_int);        \
    }

}  // namespace pf
