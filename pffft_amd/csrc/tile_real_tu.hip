// libpffft_hip.so, translation unit of the REAL transforms beyond LDS in two sweeps (fft_tile.h RMODE, round 4): instantiations + host side.
// Reference: rfftf1_ps / rfftb1_ps + real_finalize / real_preprocess (src/pffft_priv_impl.h:809-901, :1330-1462) sweep the vector
// once per radix pass plus once for the pair pass; here N = N1 N2 real points go through TWO tile passes, the half-spectrum split
// inside the column tiles and the Hermitian symmetry in the row tiles' stores.
#include <hip/hip_runtime.h>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_tile.h"

namespace pf {

template <typename T, int LOGL, int PP, int SEQC, int RMODE>
static int rtile_pass(const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, hipStream_t st, Setup* s) {
    typedef TileGeom<T, LOGL, PP, 1> G;
    const size_t lds = G::lds_bytes(D.M > (1ull << (2 * G::WB)) ? 3 : 2);
    constexpr int PF = (G::IMG_BYTES > 40 * 1024) ? 1 : 0;      // the rule of tile_host.h: prefetch where one or two workgroups fill a CU
    auto k = tile_fft_kernel<T, LOGL, PP, FWD, SEQC, PF, 0, 0, 1, 0, RMODE>;
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), G::WG, lds, &per_cu))) return rc;
    unsigned long long grid = (unsigned long long)num_cus() * per_cu;
    if (grid > ntiles) grid = ntiles;
    const bool want_dyn = (size_t)G::L * G::C * sizeof(cx<T>) >= 60 * 1024;
    // pass B (RMODE 2) takes its tiles from one work counter PER XCD, each over a contiguous eighth of the tiles (TileDesc::xmode bit 1): the
    // partner-bin runs of a row tile are off by one element against the 128-byte lines, every line is shared with the neighbour tile, and
    // neighbours then meet in one L2 - double N = 2^17 .. 2^20 0.226 / 0.303 / 0.273 / 0.220 -> 0.285 / 0.339 / 0.315 / 0.261, float 2^18 /
    // 2^19 0.216 / 0.227 -> 0.242 / 0.246 (r4_rfft_x.py (earlier-round tool, git history); dropping the streaming hint of those stores: no change).
    // PFFFT_HIP_RFFT_X=0: one counter (A/B)
    static const int x_env = dev_env("PFFFT_HIP_RFFT_X", 2);
    const bool dynm = !(ntiles <= grid || !want_dyn || ntiles >= 0xfffffff0ull);
    const bool xctr = RMODE == 2 && dynm && (x_env & 2) && grid % 8 == 0 && ntiles >= 64;
    unsigned* ctr = !dynm ? nullptr : take_counters(s, st, xctr ? 5 : 1);
    TileDesc D2 = D;
    D2.xmode = xctr ? 2u : 0u;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(G::WG), lds, st, in, out, ntiles, D2, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T, int SEQC, int RMODE>
static int rtile_dispatch(int logl, const cx<T>* in, cx<T>* out, unsigned long long ntiles, const TileDesc& D, hipStream_t st, Setup* s) {
    switch (logl) {
        case 8: return rtile_pass<T, 8, 8, SEQC, RMODE>(in, out, ntiles, D, st, s);
        case 9: return rtile_pass<T, 9, 8, SEQC, RMODE>(in, out, ntiles, D, st, s);
        case 10: return rtile_pass<T, 10, 4, SEQC, RMODE>(in, out, ntiles, D, st, s);
        default: break;
    }
    g_last_error = "pffft_hip: real tile pass length out of range";
    return (int)hipErrorInvalidValue;
}

// the split N = N1 N2 of a real power-of-two length: both tile lengths in 256 .. 1024, the column pass the shorter one
static bool rtile_split(long long N, int* l1, int* l2) {
    if (N <= 0 || (N & (N - 1))) return false;
    int logn = 0;
    while ((1ll << logn) < N) ++logn;
    if (logn < 16 || logn > 20) return false;
    *l1 = logn / 2; *l2 = logn - *l1;
    return true;
}

// adopted: true -> only the lengths where the two sweeps pay for BOTH layouts of the forward transform.  pffft_transform_ordered ==
// pffft_zreorder(pffft_transform) holds bit for bit, so the unordered transform takes the same two sweeps and then the one-sweep
// permutation big_block_kernel<5> - against tile passes + pair-and-layout sweep of the three-sweep route.  MI355X, 1 GiB per launch,
// fraction of 8 TB/s, two sweeps ordered / unordered against three (r4_rfft_x.py (earlier-round tool, git history)): float 2^16 0.232 / 0.192 against 0.250 / 0.247,
// 2^17 0.254 / 0.198 : 0.238 / 0.243, 2^18 0.243 / 0.187 : 0.237 / 0.234, 2^19 0.234 / 0.189 : 0.253 / 0.249, 2^20 0.230 / 0.186 : 0.214 / 0.225;
// double 2^16 0.223 / 0.182 : 0.247 / 0.246, 2^17 0.288 / 0.220 : 0.245 / 0.247, 2^18 0.342 / 0.253 : 0.239 / 0.256, 2^19 0.318 / 0.233 :
// 0.248 / 0.263, 2^20 0.262 / 0.201 : 0.237 / 0.241.  The Hermitian partner bins N - k of a row tile sit off by one element against the
// 128-byte grid; those partial-line stores (0.5 GiB in ~200 us with one work counter, r4_real_prof.py (earlier-round tool, git history)) ate most of the saved sweep
// until the row tiles were taken per XCD
bool tile_rfft_has_plan(long long N, bool is_double, bool adopted) {
    int l1, l2;
    if (!rtile_split(N, &l1, &l2)) return false;
    if (!adopted) return true;
    const int logn = l1 + l2;
    return is_double && (logn == 18 || logn == 19);
}

// complex elements of the work buffer per vector: rows k1 = 0 .. N1/2 of N2 values, the row count rounded up to whole row tiles
size_t tile_rfft_work_elems(long long N, bool is_double) {
    int l1, l2;
    if (!rtile_split(N, &l1, &l2)) return 0;
    const int ppB = l2 == 10 ? 4 : 8, CB = ppB * (is_double ? 1 : 2);
    const long long N1 = 1ll << l1, N2 = 1ll << l2;
    const long long rows = (N1 / 2 + 1 + CB - 1) / CB * CB;
    return (size_t)(rows * N2);
}

template <typename T>
static int rfft_fwd(Setup* s, const T* in, cx<T>* work, cx<T>* out, size_t batch, long long N, hipStream_t st) {
    int l1, l2;
    if (!rtile_split(N, &l1, &l2)) return -1;
    const unsigned long long N1 = 1ull << l1, N2 = 1ull << l2, M2 = N2 / 2, n = (unsigned long long)N / 2;
    const unsigned long long wstride = tile_rfft_work_elems(N, sizeof(T) == 8);
    {   // pass A: the N1 x M2 complex matrix, column transforms of length N1, split + twiddle, rows k1 <= N1/2 of N2 values
        const int pp = l1 == 10 ? 4 : 8, C = pp * TileUnit<T>::S;
        TileDesc D{};
        D.TA = (unsigned)(M2 / C); D.TB = 1;
        D.vstride = n; D.ovstride = wstride;
        D.in_a = C; D.ips = M2; D.iss = 1;
        D.col_a = (unsigned)C; D.M = (unsigned long long)N; D.seq_contig = 1;
        int rc = rtile_dispatch<T, 1, 1>(l1, (const cx<T>*)in, work, batch * D.TA, D, st, s);
        if (rc) return rc;
    }
    {   // pass B: rows k1 <= N1/2 (contiguous, N2 values), transposing Hermitian store into the canonical half spectrum
        const int pp = l2 == 10 ? 4 : 8, C = pp * TileUnit<T>::S;
        TileDesc D{};
        D.TA = (unsigned)((N1 / 2 + 1 + C - 1) / C); D.TB = 1;
        D.vstride = wstride; D.ovstride = n;
        D.in_a = (unsigned long long)C * N2; D.ips = 1; D.iss = N2;
        D.M = 0; D.seq_contig = 0; D.rn1 = (unsigned)N1;
        return rtile_dispatch<T, 0, 2>(l2, work, out, batch * D.TA, D, st, s);
    }
}

// real forward transform of `batch` vectors of N points into the canonical half-complex spectrum, two sweeps; -1: no plan
int launch_tile_rfft(Setup* s, const void* in, void* work, void* out, size_t batch, long long N, int dir, hipStream_t st) {
    if (dir != PFFFT_FORWARD) return -1;
    if (s->is_double) return rfft_fwd<double>(s, (const double*)in, (cx<double>*)work, (cx<double>*)out, batch, N, st);
    return rfft_fwd<float>(s, (const float*)in, (cx<float>*)work, (cx<float>*)out, batch, N, st);
}

}  // namespace pf
