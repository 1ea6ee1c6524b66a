// libpffft_hip.so, translation unit of the two / three-pass tile kernels for power-of-two sizes beyond LDS (fft_tile.h).
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_tile.h"

namespace pf {

static int g_tile_pp = [] { const char* e = getenv("PFFFT_HIP_TILE_PP"); return e ? atoi(e) : 0; }();   // A/B: force 4 or 8
static int g_tile_pf = [] { const char* e = getenv("PFFFT_HIP_TILE_PF"); return e ? atoi(e) : -1; }();  // A/B: prefetch off / on

template <typename T, int LOGL, int PP>
static int tile_pass(const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                     bool out_int = false, bool in_int = false) {
    typedef TileGeom<T, LOGL, PP> G;
    const size_t lds = G::lds_bytes(D.M > (1ull << (2 * G::WB)) ? 3 : 2);
    void (*k)(const cx<T>*, cx<T>*, unsigned long long, TileDesc, unsigned*);
    // register prefetch of the next tile where one or two workgroups fill a CU (images of 40 KiB and more)
    const bool pf = g_tile_pf >= 0 ? g_tile_pf != 0 : lds > 40 * 1024;
    const bool fw = dir == PFFFT_FORWARD;
    if (D.seq_contig && in_int && !fw) k = pf ? tile_fft_kernel<T, LOGL, PP, BWD, 1, 1, 0, 1> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 0, 0, 1>;
    else if (D.seq_contig) k = pf ? (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 1, 1> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 1>)
                             : (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 1, 0> : tile_fft_kernel<T, LOGL, PP, BWD, 1, 0>);
    else if (out_int && fw) k = pf ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 1, 1> : tile_fft_kernel<T, LOGL, PP, FWD, 0, 0, 1>;
    else k = pf ? (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 1> : tile_fft_kernel<T, LOGL, PP, BWD, 0, 1>)
                : (fw ? tile_fft_kernel<T, LOGL, PP, FWD, 0, 0> : tile_fft_kernel<T, LOGL, PP, BWD, 0, 0>);
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    PF_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), G::WG, lds));
    if (per_cu < 1) per_cu = 1;
    unsigned long long grid = (unsigned long long)num_cus() * per_cu;
    if (grid > ntiles) grid = ntiles;
    // in-order tiles only where a tile is 64 KiB or more: one counter address serves ~80 M atomics/s, so 16-32 KiB tiles
    // are throttled by the grab (2^15: 0.30 static, 0.20 in order; 2^18 .. 2^20: 0.27-0.31 / 0.19 static, 0.29-0.32 / 0.24 in
    // order).  PFFFT_HIP_TILE_DYN=0/1 forces it (A/B).
    static const int dyn_env = [] { const char* e = getenv("PFFFT_HIP_TILE_DYN"); return e ? atoi(e) : -1; }();
    const bool want_dyn = dyn_env >= 0 ? dyn_env != 0 : (size_t)G::L * G::C * sizeof(cx<T>) >= 64 * 1024;
    unsigned* ctr = (ntiles <= grid || !want_dyn || ntiles >= 0xfffffff0ull) ? nullptr : s->d_ctr + 2 * (s->ctr_slot.fetch_add(1) % CTR_RING);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(G::WG), lds, st, in, out, ntiles, D, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T, int PP>
static int tile_dispatch(int logl, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, int dir, hipStream_t st, Setup* s,
                         bool out_int = false, bool in_int = false) {
    switch (logl) {
        case 6: return tile_pass<T, 6, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 7: return tile_pass<T, 7, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 8: return tile_pass<T, 8, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 9: return tile_pass<T, 9, PP>(in, out, ntiles, D, dir, st, s, out_int, in_int);
        case 10: if constexpr (PP == 4) return tile_pass<T, 10, 4>(in, out, ntiles, D, dir, st, s, out_int, in_int);
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
    if (g_tile_pp == 4 || g_tile_pp == 8) return g_tile_pp;
    return 8;                                 // 128-byte runs: 64-byte runs measured 0.18 against 0.30 of the roofline
}

// pass A: `nvec` vectors of len = L x cols complex points; length-L transforms over the columns (stride cols), times
// W_len^(k col); same layout out (in place allowed)
// in_int (backward only, the first pass of a transform): the columns are read from the pffft-internal layout (tile_fft_kernel IINT)
template <typename T>
static int pass_columns(Setup* s, const cx<T>* in, cx<T>* out, unsigned long long nvec, int logl, unsigned long long cols, int dir, hipStream_t st,
                        bool in_int = false) {
    const int pp = pick_pp<T>(logl), C = pp * TileUnit<T>::S;
    TileDesc D{};
    D.TA = (unsigned)(cols / C); D.TB = 1;
    D.vstride = ((unsigned long long)1 << logl) * cols;
    D.in_a = C; D.out_a = C; D.ips = cols; D.iss = 1; D.ops = cols;
    D.col_a = (unsigned)C; D.M = D.vstride; D.seq_contig = 1;
    const unsigned long long ntiles = nvec * D.TA;
    return pp == 8 ? tile_dispatch<T, 8>(logl, in, out, ntiles, D, dir, st, s, false, in_int) : tile_dispatch<T, 4>(logl, in, out, ntiles, D, dir, st, s, false, in_int);
}

// pass B: rows of length L = 2^logl; row (vec, o, i) [o < outer, i < inner] sits at vec vlen + (o inner + i) L and its
// spectrum goes to X[vec vlen + (k inner + i) outer + o]   (outer index fastest: runs of C adjacent o)
// out_int (forward only): the spectrum is stored in the pffft-internal layout (tile_fft_kernel OINT)
template <typename T>
static int pass_rows(Setup* s, const cx<T>* in, cx<T>* out, unsigned long long nvec, int logl, unsigned long long outer,
                     unsigned long long inner, int dir, hipStream_t st, bool out_int = false) {
    const int pp = pick_pp<T>(logl), C = pp * TileUnit<T>::S;
    const unsigned long long L = (unsigned long long)1 << logl;
    TileDesc D{};
    D.TA = (unsigned)(outer / C); D.TB = (unsigned)inner;
    D.vstride = outer * inner * L;
    D.in_a = (unsigned long long)C * inner * L; D.in_b = L; D.ips = 1; D.iss = inner * L;
    D.out_a = C; D.out_b = outer; D.ops = outer * inner;
    D.M = 0; D.seq_contig = 0;
    const unsigned long long ntiles = nvec * D.TA * D.TB;
    return pp == 8 ? tile_dispatch<T, 8>(logl, in, out, ntiles, D, dir, st, s, out_int) : tile_dispatch<T, 4>(logl, in, out, ntiles, D, dir, st, s, out_int);
}

// canonical complex transform of `batch` vectors of n = 2^logn points: in -> out through ONE work buffer of the same size
// (in may equal out; work must differ from both).  Returns -1 when the size is outside the tile plans.
// out_int: forward transform straight into the internal layout (the last pass stores it: no reorder sweep)
template <typename T>
static int tile_fft(Setup* s, const cx<T>* in, cx<T>* work, cx<T>* out, size_t batch, int logn, int dir, hipStream_t st, bool out_int, bool in_int) {
    const int minlog = 12;
    if (logn < minlog || logn > 27) return -1;
    int rc;
    if (logn <= 20) {
        const int l1 = logn / 2, l2 = logn - l1;
        if ((rc = pass_columns<T>(s, in, work, batch, l1, 1ull << l2, dir, st, in_int))) return rc;
        return pass_rows<T>(s, work, out, batch, l2, 1ull << l1, 1, dir, st, out_int);
    }
    const int l1 = logn / 3, rem = logn - l1, l2 = rem / 2, l3 = rem - l2;
    // n = L1 n', n' = L2 L3:  A over L1 (columns n'), then per row of length n': A over L2 (in place), B over L3 with the
    // scatter X[(k3 L2 + k2) L1 + k1]
    if ((rc = pass_columns<T>(s, in, work, batch, l1, 1ull << rem, dir, st, in_int))) return rc;
    if ((rc = pass_columns<T>(s, work, work, batch << l1, l2, 1ull << l3, dir, st))) return rc;
    return pass_rows<T>(s, work, out, batch, l3, 1ull << l1, 1ull << l2, dir, st, out_int);
}

// layout: 1 = forward, spectrum out in the internal layout; 2 = backward, spectrum in from the internal layout
int launch_tile_fft(Setup* s, const void* in, void* work, void* out, size_t batch, int logn, int dir, hipStream_t st, int layout) {
    const bool out_int = layout == 1, in_int = layout == 2;
    if ((out_int && dir != PFFFT_FORWARD) || (in_int && dir != PFFFT_BACKWARD)) return -1;
    if (s->is_double) return tile_fft<double>(s, (const cx<double>*)in, (cx<double>*)work, (cx<double>*)out, batch, logn, dir, st, out_int, in_int);
    return tile_fft<float>(s, (const cx<float>*)in, (cx<float>*)work, (cx<float>*)out, batch, logn, dir, st, out_int, in_int);
}

}  // namespace pf
