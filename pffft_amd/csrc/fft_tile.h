// Power-of-two transforms beyond LDS in TWO (n <= 2^20) or THREE (n <= 2^27) passes over HBM.
//
// Reference: what the reference does for every size — one sweep over main memory per radix pass (cfftf1_ps,
// src/pffft_priv_impl.h:1004-1048; N up to 2^26 accepted, :1069) — reduced to the minimum a workgroup's LDS allows.
// Replaces, for power-of-two n, the three-to-five-sweep composition of fft_big.h (0.10-0.24 of the
// roofline, >= 6 x vector bytes moved against 2 x algorithmic).
//
// n = N1 N2:   x[n1 N2 + n2] --pass A--> Y[k1 N2 + n2] = W_n^(k1 n2) sum_n1 x[..] W_N1^(n1 k1)
//              Y[k1 N2 + n2] --pass B--> X[k1 + N1 k2] = sum_n2 Y[..] W_N2^(n2 k2)
// Both passes are the same kernel on a TILE of C sequences x L points held in one LDS image [point][sequence]
// (sequence fastest, 16-byte units = 2 float / 1 double sequences, rows padded by one unit):
//   pass A: the C sequences are ADJACENT COLUMNS — every global access is a run of C complex numbers (128 bytes), the
//           loads go straight into the stage-0 operand registers;
//   pass B: the C sequences are C rows (contiguous over their points): coalesced loads, transposed into the image,
//           and the spectrum leaves from the image in runs of C adjacent k1 — the transpose is the store.
// Radix-8 Stockham stages (a leading radix 2 / 4 for odd log2 L), 8 points per thread and sequence, stage twiddles
// W_L^k and the four-step twiddles W_M^m (three-level 512-entry tables, two products per twiddle) built in LDS at kernel
// start by exact-argument sincospi in double: no host tables.
// n > 2^20: n = L1 (L2 L3): pass A over L1, then passes A / B on the rows of length L2 L3 with the scatter of the last pass
// carrying both outer indices (TileDesc strides).
// Sizes with factors 3 and 5 (round 3): tile lengths L = R0 2^b, R0 = 3, 5, 9, 15 - the same kernel with ONE odd Stockham stage
// (radix R0 in registers, cxmath.h) in front of the power-of-two stages; n = L1 L2 with the odd part of n split over the two
// passes (tile_tu.hip: tile_plan).  Replaces three streaming passes (fft_big.h) by two for those n.
#pragma once
#include <type_traits>

#include "cxmath.h"
#include "fft_big.h"   // pair_root: the W_N^k of every real pair pass beyond LDS

namespace pf {

struct TileDesc {
    unsigned TA, TB;                                  // tiles per vector: TA x TB; tile id = (vec TA + a) TB + b
    unsigned long long vstride;                       // complex elements per vector
    unsigned long long in_a, in_b, out_a, out_b;      // tile base = vec vstride + a x_a + b x_b
    unsigned long long ips, iss;                      // input strides between points / between sequences
    unsigned long long ops;                           // output stride between points (sequences are adjacent)
    unsigned col_a, col_b;                            // four-step twiddle column of sequence c: a col_a + b col_b + c
    unsigned long long M;                             // modulus of the four-step twiddle, output (k, c) *= W_M^(k col); 0 = none
    int seq_contig;                                   // 1: pass A tile (iss == 1), 0: pass B tile (ips == 1)
    // ragged last tile along a (float only: a tile is 16 sequences, a tile length may be 8 mod 16): 16-byte units of sequences
    // that exist in tile a = TA - 1 (every other tile: all of them).  0 = not ragged.
    unsigned last_units;
    // tiles per grab of the work counter (1 or 2).  2 for the column tiles with 64-byte runs (8 float / 4 double columns): two
    // adjacent tiles share every 128-byte line of their point rows, and taken by DIFFERENT workgroups they are fetched through
    // two XCDs' L2s - rocprofv3 counted 1.456 x the algorithmic bytes on pass A of N = 2^20 (profiles/r03_pmc.md); one workgroup
    // taking both, back to back, finds the second half in its own L2
    unsigned group;
    // real transforms in two sweeps (RMODE, round 4): output vector stride when it differs from the input's (0 = the same), and the
    // column length N1 of the real four-step split (rows k1 <= N1 / 2 exist)
    unsigned long long ovstride;
    unsigned rn1;
    // round 4, which tiles a workgroup takes (workgroup b runs on XCD b mod 8): bit 0 - static stride: XCD x takes a CONTIGUOUS eighth of
    // every sweep of the grid (the grid is a whole number of eights); bit 1 - in order: one work counter per XCD over a contiguous eighth
    // of the tiles (ctr[0 .. 7] next, ctr[8] done).  Tiles that share 128-byte lines then meet in one L2, and eight counter addresses
    // serve the start-up burst of three grabs per workgroup (fft_tileg.h has the measurements)
    unsigned xmode;
};

template <typename T> struct TileUnit;                // one 16-byte LDS / global unit
template <> struct TileUnit<float> {
    typedef __attribute__((ext_vector_type(4))) float U;
    static constexpr int S = 2;
    static __device__ __forceinline__ cx<float> get(const U& u, int s) { return s ? mk<float>(u.z, u.w) : mk<float>(u.x, u.y); }
    static __device__ __forceinline__ void set(U& u, int s, cx<float> v) { if (s) { u.z = v.x; u.w = v.y; } else { u.x = v.x; u.y = v.y; } }
};
template <> struct TileUnit<double> {
    typedef __attribute__((ext_vector_type(2))) double U;
    static constexpr int S = 1;
    static __device__ __forceinline__ cx<double> get(const U& u, int) { return mk<double>(u.x, u.y); }
    static __device__ __forceinline__ void set(U& u, int, cx<double> v) { u.x = v.x; u.y = v.y; }
};

// R0 > 1 (round 3, sizes with factors 3 and 5 beyond LDS): L = R0 2^LOGL, an odd first stage of radix R0 in front of the
// power-of-two stages
// R0 = 25, 27, 45: TWO odd stages (5 x 5, 9 x 3, 9 x 5), the second one with twiddles like every later stage
template <typename T, int LOGL, int PP, int R0 = 1> struct TileGeom {
    static constexpr int L = R0 << LOGL, TPT = L / 8, WG = TPT * PP, S = TileUnit<T>::S, C = PP * S;
    static constexpr int PITCH = PP + 1;                                  // 16-byte units per point row
    static constexpr int RA = R0 == 25 ? 5 : R0 == 27 ? 9 : R0 == 45 ? 9 : R0, RB = R0 / RA;   // the odd stages
    static constexpr int NODD = (RA > 1) + (RB > 1);
    static constexpr int NS = (LOGL + 2) / 3 + NODD;
    // internal-layout output (OINT): the last stage's rows are skewed by QSKEW units per spectrum quarter so that the
    // block-gather of the store loop reads conflict-free (+ 3 QSKEW units at the end of the image)
    static constexpr int QSKEW = S == 2 ? 4 : 1;
    static constexpr size_t IMG_BYTES = (size_t)L * PITCH * 16 + 256;
    // a plain column pass (SEQC && !IINT) keeps its image WITHOUT the padding unit (the kernel's PITCH = PP): round 6 gives it the LDS of
    // that image only - L = 1024 on 64-byte runs: 64 KiB + tables = 75 KiB instead of 100, L = 512 on 128-byte runs 76 instead of 84:
    // TWO workgroups per CU where one ran alone with nothing to overlap its barrier-separated phases
    static constexpr size_t IMG_BYTES_PLAIN = (size_t)L * PP * 16 + 256;
    static constexpr bool SWZ = LOGL == 10 && PP == 4 && R0 == 1;         // XOR-swizzled unpadded image (tile_swz) for the plain passes of this geometry
    // + W_L^k (L entries) + `levels` x 2^WB entries of the four-step twiddle table.  WB = 9 (two levels reach M = 2^18, three
    // 2^27); the one tile that fills LDS - L = 1024 with 128-byte runs (PP = 8): 147 KiB of image - takes three levels of 2^7
    // (M <= 2^21), 3 KiB instead of 12
    static constexpr int WB = (LOGL == 10 && R0 == 1) ? 7 : 9;
    // (the swizzled geometry in double keeps HALF the W_L table - W_L^(k + L/2) = -W_L^k: 8 KiB instead of 16, which is what lets two of its
    //  workgroups share a CU's 160 KiB)
    static constexpr bool HALFW = SWZ && sizeof(T) == 8;
    __host__ __device__ static constexpr size_t lds_bytes(int levels, bool plain = false) {
        return (plain ? IMG_BYTES_PLAIN : IMG_BYTES) + ((size_t)((plain && HALFW) ? L / 2 : L) + ((size_t)levels << WB)) * 2 * sizeof(T) + 16;
    }
    __host__ __device__ static constexpr int rad(int s) {
        if (s < NODD) return s == 0 ? RA : RB;
        s -= NODD;
        return (LOGL % 3 == 0 || s > 0) ? 8 : (1 << (LOGL % 3));
    }
    __host__ __device__ static constexpr int nsprod(int s) { int p = 1; for (int i = 0; i < s; ++i) p *= rad(i); return p; }
};

template <typename T> __device__ __forceinline__ cx<T> tile_unit_root(double turns) {   // exp(-2 pi i turns)
    double sn, cs;
    sincospi(2.0 * turns, &sn, &cs);
    return mk<T>((T)cs, (T)-sn);
}

// `levels`-level table of W_M^m with 2^WB entries per level: w3[l][d] = W_M^(d 2^(WB l)); W_M^idx = product of its digits' entries
template <int WB, typename CX> __device__ __forceinline__ CX tile_w3(const CX* w3, unsigned idx, bool lv3) {
    constexpr unsigned MSK = (1u << WB) - 1;
    CX f = cmul(w3[idx & MSK], w3[(1u << WB) + ((idx >> WB) & MSK)]);
    if (lv3) f = cmul(f, w3[(2u << WB) + (idx >> (2 * WB))]);
    return f;
}

// unit index of (point pt, unit p) in the swizzled image of four units per point (L = 1024): the unit's slot inside its 8-unit window (two
// points) is XORed with a mask that is linear in bits 1, 2, 3, 8, 9 of the point index - masks 5, 2, 4, 5, 3, found by exhaustive search with
// tools/lds_sim.py (tools/swz_search.py): every access pattern of the kernel, float and double, costs the ideal number of LDS cycles - the
// transposing write of a row pass (eight lanes on the same unit of eight consecutive even / odd / plain points), the stage reads (sixteen lanes on
// four points), the stage writes (two lanes on points 2 t + d and 2 t + 2 + d in the radix-2 first stage, on adjacent points later), the copy-out,
// and the quarter gathers of the internal layout (points ptq + m L/4: bits 8, 9) in both directions
__device__ __forceinline__ int tile_swz(int pt, int p) {
    const int b18 = ((pt >> 1) ^ (pt >> 8)) & 1;
    return ((pt << 2) | p) ^ (b18 * 5) ^ ((pt >> 1) & 2) ^ ((pt >> 1) & 4) ^ (((pt >> 9) & 1) * 3);
}

// SEQC = 1: pass A (adjacent columns, four-step twiddle)   SEQC = 0: pass B (rows in, transposing store)
// PF = 1: the loads of the workgroup's next tile fly while the current one is transformed (for the tiles so large that
//         one workgroup fills a CU: nothing else would overlap its load latency)
// OINT = 1 (pass B of a forward transform only): the spectrum leaves in the pffft-internal layout instead of the canonical one
//   (reference: pffft_transform's output, src/pffft_priv_impl.h:1195-1237 cplx_finalize; SURVEY.md appendix A:
//   internal[32 b + 8 m + 4 p + l] = part p of X[m n/4 + 4 b + l]).  A pass-B tile holds, for its C adjacent outer indices,
//   ALL points k of the rows, i.e. the four quarters m of every bin group: per point k' < L/4 and group of 4 adjacent outer
//   indices the image yields one whole 128-byte (float) / 256-byte (double) block of the layout, per k' a contiguous run of
//   C/4 blocks - the canonical -> internal reorder costs no extra sweep over HBM.
// IINT = 1 (the first pass A of a backward transform only): the mirror image - the tile's C adjacent columns are read from the
//   internal layout (pffft_transform backward takes it, :1423-1462 / cplx_preprocess): per point row n1' < L/4 the four
//   quarters of the columns are a run of C/4 whole blocks, loaded as dense 16-byte units; a lane ^ 1 (float) / lane ^ 2 (double)
//   DPP exchange turns the (re group, im group) units into the image's (re, im) sequence units.
// RAG = 1 (float only): the last tile along a may be ragged (TileDesc::last_units); RAG = 0 compiles the predicates away (they cost
// the L = 512 kernels 4 % when left to run time)
// RMODE (round 4, power-of-two L): REAL transforms of N = N1 N2 points in TWO sweeps.  The real vector is the
// N1 x N2/2 complex matrix z[j1][m] = (x[j1 N2 + 2m], x[j1 N2 + 2m + 1]): adjacent real columns in pairs.
//   1 (pass A, forward): column transforms of length L = N1 as always; the spectra of the two real columns of every complex column
//     are split off INSIDE the tile - E = (C[k1] + conj C[N1 - k1]) / 2, O = -i (C[k1] - conj C[N1 - k1]) / 2, the mirror is in the same
//     column -, twiddled by W_N^(k1 j2) and stored as rows k1 = 0 .. N1/2 of N2 complex values (runs of 2 C adjacent values)
//   2 (pass B, forward): row transforms of length L = N2 of the rows k1 <= N1/2: X[k1 + N1 k2]; k2 < N2/2 goes straight to its bin, k2 >=
//     N2/2 conjugated to bin N - k = (N1 - k1) + N1 (N2 - 1 - k2) (Hermitian symmetry: the rows k1 > N1/2 are never computed) - every
//     tile stores to two places instead of needing its mirror tile; bin 0 carries (DC, Nyquist) (include/pffft/pffft.h:144-155)
//   3 (round 6, the LAST pass of a real forward transform on the complex core n = N/2 - any tile length, 128-byte runs): row tiles whose ROW SET IS
//     CLOSED UNDER THE MIRROR k1 -> N1 - k1 - tile a = rows [aH, aH + H) and (N1 - aH - H, N1 - aH], H = C/2; tile 0 takes row N1/2 in place of
//     the second copy of row 0 - so that both bins of every pair (k, n - k) = (k1 + N1 k2, (N1 - k1) + N1 (L - 1 - k2)) lie in ONE image: the pair pass
//     X[k] = S + D, X[n-k] = conj(S - D) runs in LDS, in place, and the tile stores the canonical half-complex spectrum itself - two runs of H
//     adjacent k1 per point (64 bytes; the mirror run sits one element off the grid).  The pair sweep over HBM of the three-sweep route is gone.
template <typename T, int LOGL, int PP, int DIR, int SEQC, int PF, int OINT = 0, int IINT = 0, int R0 = 1, int RAG = 0, int RMODE = 0>
__global__ void __launch_bounds__((R0 << LOGL) / 8 * PP, PF ? 2 : 3)
tile_fft_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, unsigned long long ntiles, TileDesc D, unsigned* ctr) {
    typedef cx<T> CX;
    typedef TileGeom<T, LOGL, PP, R0> G;
    typedef TileUnit<T> TU;
    typedef typename TU::U U;
    constexpr int L = G::L, TPT = G::TPT, WG = G::WG, S = G::S, C = G::C, NS = G::NS;
    // 16-byte units per point row of the image.  The padding unit is for the TRANSPOSING write of a row pass (and the quarter rows of the
    // internal-layout input); a plain column pass touches the image only in whole unit rows - lanes (t, p), p fastest - and those are
    // conflict-free exactly WITHOUT it: ds_read_b128 of four consecutive rows at a pitch of 9 units costs 8 cycles against 4 (tools/lds_sim.py;
    // rocprofv3 had 0.56 conflict cycles per active cycle on pass A of N = 2^20).  The LDS size stays that of the padded image.
    // SWZ (round 6): the L = 1024 tiles (64-byte runs, PP = 4) keep an UNPADDED image whose 16-byte units are XOR-swizzled by their point index
    // (tile_swz): every access pattern of the kernel - the transposing write of a row pass (16-byte units now: a thread loads two adjacent points
    // of every row), the stage reads and writes of every Ns, the copy-out, the gathers of the internal layout - is conflict-free in
    // tools/lds_sim.py, where the padded image (pitch 5 units) cost 2.33 x the ideal LDS cycles (rocprofv3: 0.567 conflict cycles per active cycle on pass B of N = 2^20), and the image is 64 KiB
    // instead of 80: TWO workgroups per CU where the row pass ran alone
    constexpr bool SWZ = G::SWZ && RMODE == 0;
    constexpr int PITCH = (SWZ || (SEQC && !IINT)) ? PP : G::PITCH;
    // odd first stage on a pass-A tile (not from the internal layout): the loads ARE its operands, point j + q 2^LOGL of butterfly
    // j = t + TPT u (u < UB0, predicated on j < 2^LOGL)
    constexpr int RA = G::RA, RB = G::RB, NODD = G::NODD;
    constexpr int NB0 = L / RA, UB0 = (8 + RA - 1) / RA;
    constexpr bool ODD_DIRECT = R0 > 1 && SEQC && !IINT;
    constexpr bool SWZ_ROWS2 = SWZ && !SEQC && S == 2;                    // row pass, float: 16-byte loads of two adjacent points per row
    constexpr int NLD = ODD_DIRECT ? UB0 * RA : SEQC ? 8 : SWZ_ROWS2 ? C : C * L / WG;   // loads per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    U* img = reinterpret_cast<U*>(smem);
    constexpr size_t IMGB = (SWZ || (SEQC && !IINT && RMODE == 0)) ? G::IMG_BYTES_PLAIN : G::IMG_BYTES;
    auto ua = [](int pt, int pu) -> int { if constexpr (SWZ) return tile_swz(pt, pu); else return pt * PITCH + pu; };
    CX* wl = reinterpret_cast<CX*>(smem + IMGB);
    constexpr bool HALFW = SWZ && G::HALFW;
    constexpr int WLN = HALFW ? L / 2 : L;
    CX* w3 = wl + WLN;
    auto wl_at = [&](int idx) -> CX {
        if constexpr (HALFW) { const CX w = wl[idx & (L / 2 - 1)]; return (idx & (L / 2)) ? mk<T>(-w.x, -w.y) : w; }
        else return wl[idx];
    };
    const int tid = threadIdx.x, t = tid / PP, p = tid % PP;

    for (int i = tid; i < WLN; i += WG) wl[i] = tile_unit_root<T>((double)i / (double)L);
    constexpr int WB = G::WB;
    const bool lv3 = D.M > (1ull << (2 * WB));
    if (SEQC) {
        const double invM = 1.0 / (double)D.M;
        for (int i = tid; i < ((lv3 ? 3 : 2) << WB); i += WG) {
            const int lvl = i >> WB, m = i & ((1 << WB) - 1);
            w3[i] = tile_unit_root<T>((double)m * (double)(1u << (WB * lvl)) * invM);
        }
    }
    __syncthreads();
    // four-step twiddle of output k = t + d L/8 of column col0 + c:  W^(t col0) W^(t c) [W^((L/8) col0) W^((L/8) c)]^d :
    // the c-dependent factors are per-thread constants, the col0-dependent ones one table product per tile
    CX c0[S], c1[S];
    if constexpr (SEQC) {
#pragma unroll
        for (int sq = 0; sq < S; ++sq) {
            c0[sq] = tile_w3<WB>(w3, (unsigned)t * (unsigned)(S * p + sq), lv3);
            c1[sq] = tile_w3<WB>(w3, (unsigned)(L / 8) * (unsigned)(S * p + sq), lv3);
        }
    }

    // (eb: OINT only - canonical index, inside its vector, of the tile's first output element; dst is then the vector's base)
    // (pv: the 16-byte sequence units of the tile that exist - PP but for a ragged last tile)
    auto tile_bases = [&](unsigned long long tile, const CX*& src, CX*& dst, unsigned& col0, unsigned long long& eb, int& pv) {
        const unsigned b = (unsigned)(tile % D.TB);
        const unsigned long long rest = tile / D.TB;
        const unsigned a = (unsigned)(rest % D.TA);
        if constexpr (RAG) pv = (D.last_units && a == D.TA - 1) ? (int)D.last_units : PP;
        else pv = PP;
        const unsigned long long vec = rest / D.TA;
        if constexpr (IINT) { eb = a * D.in_a + b * D.in_b; src = in + vec * D.vstride; }
        else src = in + vec * D.vstride + a * D.in_a + b * D.in_b;
        if constexpr (OINT) { eb = a * D.out_a + b * D.out_b; dst = out + vec * D.vstride; }
        else if constexpr (RMODE != 0) { eb = a; dst = out + vec * D.ovstride; }               // the epilogue addresses the vector itself
        else dst = out + vec * D.vstride + a * D.out_a + b * D.out_b;
        col0 = a * D.col_a + b * D.col_b;
    };
    // pass A: unit p of points t + TPT m straight into the stage-0 operand registers;
    // pass B: elements g = tid + i WG of the [sequence][point] tile (coalesced over the points of a row)
    // RMODE 3: row of sequence seq of tile a (mirror-closed row sets, H = C / 2)
    auto mirror_row = [&](unsigned a, int seq) -> unsigned {
        constexpr int H = C / 2;
        if (seq < H) return a * H + (unsigned)seq;
        const unsigned r = D.rn1 - a * H - (unsigned)(C - 1 - seq);
        return r == D.rn1 ? D.rn1 / 2 : r;
    };
    typedef typename std::conditional<SEQC != 0 || SWZ_ROWS2, U, CX>::type LD;
    constexpr int UPB_ = 2 * (int)sizeof(T), UPP_ = (C / 4) * UPB_;   // 16-byte units per block / per point row of the internal layout
    auto issue_loads = [&](const CX* src, LD (&r)[NLD], unsigned long long eb, int pv) {
        if constexpr (IINT) {
            static_assert(!IINT || (SEQC && ((L / 4) * UPP_) == 8 * WG), "internal-layout input: eight units per thread");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = tid + i * WG, ptq = g / UPP_, rr = g % UPP_, bb = rr / UPB_;
                if (bb * 4 < pv * S)
                    r[i] = __builtin_nontemporal_load(reinterpret_cast<const U*>(src + 4 * (eb + (unsigned long long)ptq * D.ips + 4 * bb)) + rr % UPB_);
            }
        } else if constexpr (ODD_DIRECT) {
#pragma unroll
            for (int u = 0; u < UB0; ++u) {
                const int j = t + TPT * u;
                if (j < NB0 && p < pv) {
#pragma unroll
                    for (int q = 0; q < RA; ++q)
                        r[u * RA + q] = __builtin_nontemporal_load(reinterpret_cast<const U*>(src + (unsigned long long)(j + q * NB0) * D.ips + S * p));
                }
            }
        } else if constexpr (SEQC) {
            if (p < pv) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    r[m] = __builtin_nontemporal_load(reinterpret_cast<const U*>(src + (unsigned long long)(t + TPT * m) * D.ips + S * p));
            }
        } else if constexpr (SWZ_ROWS2) {
            static_assert(!SWZ_ROWS2 || 2 * WG == L, "two points per thread");
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                if (i < pv * S) r[i] = __builtin_nontemporal_load(reinterpret_cast<const U*>(src + (unsigned long long)i * D.iss + 2 * tid));
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int g = tid + i * WG, seq = WG == L ? i : g / L, pt = WG == L ? tid : g % L;   // (128-byte runs: WG == L)
                if constexpr (RMODE == 3) r[i] = __builtin_nontemporal_load(src + (unsigned long long)mirror_row((unsigned)eb, seq) * D.iss + pt);
                else if (seq < pv * S) r[i] = __builtin_nontemporal_load(src + (unsigned long long)seq * D.iss + pt);
            }
        }
    };

    // Tiles are taken IN ORDER from an atomic counter (ctr != nullptr): the workgroups in flight then sweep neighbouring
    // column bands / row blocks together, which HBM rewards; the grab runs two tiles ahead so that the
    // prefetch knows its tile.  ctr == nullptr: static stride.
    unsigned* s_next = reinterpret_cast<unsigned*>(w3 + ((lv3 ? 3 : 2) << WB));
    const bool dyn = ctr != nullptr;
    // ids of the counter / the static stride are GROUPS of K consecutive tiles: gcur = the group in work, gnext = the one after
    const unsigned K = D.group > 1 ? D.group : 1u;
    const bool xmap = !dyn && (D.xmode & 1u), xctr = dyn && (D.xmode & 2u);
    const unsigned long long gstride = xmap ? 8ull * ((gridDim.x + 7) / 8) : gridDim.x;
    unsigned long long gcur = xmap ? (unsigned long long)(blockIdx.x % 8) * ((gridDim.x + 7) / 8) + blockIdx.x / 8 : blockIdx.x, gnext = gcur + gstride;
    const unsigned long long ngroups = (ntiles + K - 1) / K, xper = (ngroups + 7) / 8, xbase = xctr ? (blockIdx.x % 8) * xper : 0;
    const unsigned long long xend = xctr ? (xbase + xper < ngroups ? xbase + xper : ngroups) : ngroups;
    unsigned* cnext = ctr + (xctr ? blockIdx.x % 8 : 0);
    // The first TWO groups of a workgroup are static (its index among the workgroups of its counter, and that plus their number): the counter
    // hands out what follows.  (Round 3 took all three start-up grabs from the counter: on a short launch - 512 tiles on 256 workgroups - the
    // first 170 workgroups to arrive took three tiles each and the rest none: 34 us per pass against 10 us for 256 tiles; and the start-up
    // burst of three atomics per workgroup on one address is off the critical path now.)
    const unsigned long long g0 = xctr ? gridDim.x / 8 : gridDim.x, lid = xctr ? blockIdx.x / 8 : blockIdx.x;
    // xmode bit 2 (round 6, with bit 1): the XCD's counter deals tile PAIRS round robin - local index l of XCD x is tile 16 (l / 2) + 2 x + l % 2 -
    // instead of a contiguous eighth: the two column tiles that share every 128-byte line of their point rows (64-byte runs) are taken back to
    // back by two workgroups of ONE XCD and meet in its L2, while all eight XCDs still sweep the batch together in order (the contiguous
    // eighths of bit 1 alone cost the register-tiled passes 10-15 %: eight distant fronts in HBM)
    const bool xpair = xctr && (D.xmode & 4u);
    auto ranged = [&](unsigned long long local) -> unsigned long long {
        if (xpair) { const unsigned long long g = 16ull * (local >> 1) + 2ull * (blockIdx.x % 8) + (local & 1ull); return g < ngroups ? g : ngroups; }
        const unsigned long long g = xbase + local; return g < xend ? g : ngroups;
    };
    unsigned pend = 0;
    const bool cstart = dyn && (D.xmode & 8u);       // (A/B, PFFFT_HIP_TILE_CSTART=1: the three start-up grabs of round 3 - 1-3 % slower at 1 GiB, N = 2^18 .. 393216)
    const unsigned long long goff = cstart ? 0 : 2 * g0;
    auto grabbed = [&](unsigned v) -> unsigned long long { return ranged(goff + v); };
    if (cstart) {
        if (tid == 0) { s_next[0] = atomicAdd(cnext, 1u); s_next[1] = atomicAdd(cnext, 1u); pend = atomicAdd(cnext, 1u); }
        __syncthreads();
        gcur = ranged(s_next[0]); gnext = ranged(s_next[1]);
        __syncthreads();
    } else if (dyn) {
        gcur = ranged(lid); gnext = ranged(lid + g0);
        if (tid == 0) pend = atomicAdd(cnext, 1u);
    }
    unsigned sub = 0;                                 // tile of the group in work
    unsigned long long tile = gcur * K, tile1 = K > 1 ? tile + 1 : gnext * K;
    LD nxt[NLD];
    if constexpr (PF) {
        if (tile < ntiles) { const CX* s0; CX* d0; unsigned cc; unsigned long long e0 = 0; int pv0; tile_bases(tile, s0, d0, cc, e0, pv0); issue_loads(s0, nxt, e0, pv0); }
    }
    unsigned gi = 0;                                  // groups begun so far
    for (unsigned it = 0; tile < ntiles; ++it) {
        if (dyn && tid == 0 && sub == 0) {
            s_next[gi & 1] = pend;                   // the group after the next one, read by everyone after the first barrier below
            pend = atomicAdd(cnext, 1u);
        }
        const CX* src; CX* dst; unsigned col0; unsigned long long ebase = 0;
        int pv;
        tile_bases(tile, src, dst, col0, ebase, pv);
        LD cur[NLD];
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) cur[i] = nxt[i];
            if (tile1 < ntiles) { const CX* s1; CX* d1; unsigned cc; unsigned long long e1 = 0; int pv1; tile_bases(tile1, s1, d1, cc, e1, pv1); issue_loads(s1, nxt, e1, pv1); }
        } else {
            issue_loads(src, cur, ebase, pv);
        }
        U v[8];   // v[m] = unit p of point t + TPT m
        if constexpr (IINT) {
            // (re group, im group) units of the layout -> (re, im) sequence units of the image
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = tid + i * WG, ptq = g / UPP_, rr = g % UPP_, bb = rr / UPB_, m = (rr / (UPB_ / 4)) % 4, sub = rr % (UPB_ / 4);
                U* dp = img + (ptq + m * (L / 4)) * PITCH + m * G::QSKEW + bb * (4 / S);   // quarter rows skewed: conflict-free
                [[maybe_unused]] const int prow = ptq + m * (L / 4), pun = bb * (4 / S);
                const U x = cur[i];
                if constexpr (S == 2) {           // sub = part: even lane re0..3, odd lane im0..3 -> sequences (0, 1) / (2, 3)
                    const T s0 = dpp_xor1(sub ? x.x : x.z), s1 = dpp_xor1(sub ? x.y : x.w);
                    U o;
                    if (sub) { o.x = s0; o.y = x.z; o.z = s1; o.w = x.w; }
                    else { o.x = x.x; o.y = s0; o.z = x.y; o.w = s1; }
                    if constexpr (SWZ) img[ua(prow, pun + sub)] = o; else dp[sub] = o;
                } else {                          // sub = 2 part + (l / 2): lanes (re01, re23, im01, im23) -> sequences 0, 2, 1, 3
                    const T sv = dpp_xor2(sub >> 1 ? x.x : x.y);
                    U o;
                    if (sub >> 1) { o.x = sv; o.y = x.y; }
                    else { o.x = x.x; o.y = sv; }
                    if constexpr (SWZ) img[ua(prow, pun + 2 * (sub & 1) + (sub >> 1))] = o; else dp[2 * (sub & 1) + (sub >> 1)] = o;
                }
            }
            __syncthreads();
            if constexpr (R0 == 1) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = SWZ ? img[ua(t + TPT * m, p)] : img[(t + TPT * m) * PITCH + p + (m / 2) * G::QSKEW];   // row t + m L/8 lies in quarter m / 2
                __syncthreads();
            }
        } else if constexpr (SEQC && R0 == 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = cur[m];
        } else if constexpr (SEQC) {
            // odd first stage: cur[] holds its operands
        } else if constexpr (SWZ_ROWS2) {
            // cur[seq] = points 2 tid, 2 tid + 1 of row seq: whole units (sequences 2u, 2u + 1) of those two points
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                U a, b;
                a.x = cur[2 * u].x; a.y = cur[2 * u].y; a.z = cur[2 * u + 1].x; a.w = cur[2 * u + 1].y;
                b.x = cur[2 * u].z; b.y = cur[2 * u].w; b.z = cur[2 * u + 1].z; b.w = cur[2 * u + 1].w;
                img[ua(2 * tid, u)] = a;
                img[ua(2 * tid + 1, u)] = b;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = img[ua(t + TPT * m, p)];
            __syncthreads();
        } else {
            // C rows, contiguous over their points: transposed into the [point][sequence] image
            CX* imgc = reinterpret_cast<CX*>(img);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int g = tid + i * WG, seq = WG == L ? i : g / L, pt = WG == L ? tid : g % L;
                if constexpr (SWZ) img[ua(pt, seq)] = *reinterpret_cast<const U*>(&cur[i]);      // (double: an element is a unit)
                else imgc[pt * (PITCH * S) + seq] = cur[i];
            }
            __syncthreads();
            if constexpr (R0 == 1) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = img[ua(t + TPT * m, p)];
                __syncthreads();
            }
        }
        // ---- odd first stage (R0 > 1): butterflies j < L / RA on the operands j + q L / RA, 8 / RA per thread (predicated);
        //      outputs to j RA + d
        if constexpr (R0 > 1) {
            constexpr int NB = NB0, UB = UB0;
            U opnd[UB][RA];
            if constexpr (ODD_DIRECT) {
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < RA; ++q) opnd[u][q] = cur[u * RA + q];
            } else {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = t + TPT * u;
                    if (j < NB) {
#pragma unroll
                        for (int q = 0; q < RA; ++q) {
                            const int pt = j + q * NB;
                            opnd[u][q] = img[pt * PITCH + p + (IINT ? (pt / (L / 4)) * G::QSKEW : 0)];
                        }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = t + TPT * u;
                if (j < NB) {
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) {
                        CX o[RA];
#pragma unroll
                        for (int q = 0; q < RA; ++q) o[q] = TU::get(opnd[u][q], sq);
                        dftR<RA, DIR>(o);
#pragma unroll
                        for (int q = 0; q < RA; ++q) TU::set(opnd[u][q], sq, o[q]);
                    }
#pragma unroll
                    for (int d = 0; d < RA; ++d) img[(j * RA + d) * PITCH + p] = opnd[u][d];
                }
            }
        }
        // ---- second odd stage (R0 = 25, 27, 45): radix RB, Ns = RA: operands j + q L / RB times W_(RA RB)^(q (j mod RA)),
        //      outputs to (j div RA) RA RB + (j mod RA) + d RA
        if constexpr (RB > 1) {
            constexpr int NB = L / RB, UB = (8 + RB - 1) / RB;
            static_assert(TPT % RA == 0, "stage shape");
            U opnd[UB][RB];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = t + TPT * u;
                if (j < NB) {
#pragma unroll
                    for (int q = 0; q < RB; ++q) opnd[u][q] = img[(j + q * NB) * PITCH + p];
                }
            }
            __syncthreads();
            const int tk = t % RA, tq = t / RA;
            CX w[RB];
#pragma unroll
            for (int q = 1; q < RB; ++q) w[q] = wl_at((q * tk) * (L / (RA * RB)));
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = t + TPT * u;
                if (j < NB) {
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) {
                        CX o[RB];
#pragma unroll
                        for (int q = 0; q < RB; ++q) {
                            o[q] = TU::get(opnd[u][q], sq);
                            if (q) o[q] = twmul<DIR>(o[q], w[q]);
                        }
                        dftR<RB, DIR>(o);
#pragma unroll
                        for (int q = 0; q < RB; ++q) TU::set(opnd[u][q], sq, o[q]);
                    }
                    const int pbase = (tq + u * (TPT / RA)) * (RA * RB) + tk;
#pragma unroll
                    for (int d = 0; d < RB; ++d) img[(pbase + d * RA) * PITCH + p] = opnd[u][d];
                }
            }
        }
        // ---- Stockham stages; stage s, butterfly u (point j = t + TPT u): operands v[u + q B], outputs to
        //      (j div Ns) Ns R + (j mod Ns) + d Ns
        auto stage = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int R = G::rad(s), B = 8 / R, Ns = G::nsprod(s);
            if constexpr (s > 0) {
                static_assert(Ns * R <= L && L % (Ns * R) == 0, "stage shape");
                __syncthreads();                     // every thread wrote its outputs of the stage before
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = img[ua(t + TPT * m, p)];
                __syncthreads();                     // ... and read its operands: the image is free again
            }
            // Ns divides TPT (Ns R <= L, R <= 8): (t + TPT u) mod Ns = t mod Ns, (t + TPT u) div Ns = t div Ns + u TPT / Ns
            static_assert(TPT % Ns == 0, "stage shape");
            const int tk = t % Ns, tq = t / Ns;
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int j = t + TPT * u;
                CX w[R];
                if constexpr (s > 0) {
                    const int k = tk;
#pragma unroll
                    for (int q = 1; q < R; ++q) w[q] = wl_at((q * k) * (L / (Ns * R)));
                }
                CX o[S][R];
#pragma unroll
                for (int sq = 0; sq < S; ++sq) {
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        o[sq][q] = TU::get(v[u + q * B], sq);
                        if constexpr (s > 0) if (q) o[sq][q] = twmul<DIR>(o[sq][q], w[q]);
                    }
                    dftR<R, DIR>(o[sq]);
                }
                if constexpr (SEQC && s == NS - 1 && NS > 1 && RMODE != 1) {
                    // the last stage is a radix 8 with one butterfly per thread: outputs k = t + d L/8
                    static_assert(R == 8 && B == 1 && Ns == L / 8, "last stage shape");
                    const CX A = tile_w3<WB>(w3, (unsigned)t * col0, lv3), Bc = tile_w3<WB>(w3, (unsigned)(L / 8) * col0, lv3);
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) {
                        CX f[8];
                        f[0] = cmul(A, c0[sq]);
                        const CX st = cmul(Bc, c1[sq]), st2 = cmul(st, st), st4 = cmul(st2, st2);
                        f[1] = cmul(f[0], st); f[2] = cmul(f[0], st2); f[3] = cmul(f[1], st2);
                        f[4] = cmul(f[0], st4); f[5] = cmul(f[1], st4); f[6] = cmul(f[2], st4); f[7] = cmul(f[3], st4);
#pragma unroll
                        for (int d = 0; d < 8; ++d) o[sq][d] = twmul<DIR>(o[sq][d], f[d]);
                    }
                }
                const int pbase = (tq + u * (TPT / Ns)) * (Ns * R) + tk;
#pragma unroll
                for (int d = 0; d < R; ++d) {
                    U x;
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) TU::set(x, sq, o[sq][d]);
                    if constexpr (OINT && s == NS - 1 && NS > 1) {
                        static_assert(!OINT || (R == 8 && B == 1 && Ns == L / 8), "last stage shape");
                        if constexpr (SWZ) img[ua(pbase + d * Ns, p)] = x;
                        else img[(pbase + d * Ns) * PITCH + p + (d / 2) * G::QSKEW] = x;   // row j + d L/8 lies in quarter d / 2
                    } else {
                        img[ua(pbase + d * Ns, p)] = x;
                    }
                }
            }
        };
        if constexpr (NODD == 0) stage(std::integral_constant<int, 0>{});
        if constexpr (NS > 1 && NODD <= 1) stage(std::integral_constant<int, 1>{});
        if constexpr (NS > 2) stage(std::integral_constant<int, 2>{});
        if constexpr (NS > 3) stage(std::integral_constant<int, 3>{});
        if constexpr (NS > 4) stage(std::integral_constant<int, 4>{});
        __syncthreads();
        // ---- the image holds the spectrum [k][sequence]: runs of C adjacent sequences per point
        if constexpr (RMODE == 1) {
            // one item = (row k1 <= L/2, unit pu): the unit's S complex columns are 2 S real columns -> 2 S values of output row k1,
            // 32 bytes per lane, consecutive lanes consecutive: runs of 2 C values
            static_assert(RMODE != 1 || (SEQC && DIR == FWD && R0 == 1 && !OINT && !IINT), "real split: forward column pass");
            constexpr int ITEMS = (L / 2 + 1) * PP;
            const unsigned long long N2 = 2ull * D.ips;                      // values per output row (D.ips = complex columns of the input)
#pragma unroll
            for (int i = 0; i < (ITEMS + WG - 1) / WG; ++i) {
                const int g = tid + i * WG;
                if (g >= ITEMS) break;
                const int k1 = g / PP, pu = g % PP;
                const U u = img[k1 * PITCH + pu], m = img[((L - k1) & (L - 1)) * PITCH + pu];
                const unsigned mc = col0 + (unsigned)(S * pu);              // first complex column of the unit
                CX tw = tile_w3<WB>(w3, (unsigned)k1 * (2u * mc), lv3);     // W_N^(k1 j2), j2 = 2 mc, then j2 + 1, ...
                const CX wk = tile_w3<WB>(w3, (unsigned)k1, lv3);
                CX y[2 * S];
#pragma unroll
                for (int sq = 0; sq < S; ++sq) {
                    const CX ck = TU::get(u, sq), cm = conj(TU::get(m, sq));
                    const CX e = (ck + cm) * (T)0.5, dm = (ck - cm) * (T)0.5;
                    const CX o = mk<T>(dm.y, -dm.x);                         // -i dm
                    y[2 * sq] = cmul(e, tw); tw = cmul(tw, wk);
                    y[2 * sq + 1] = cmul(o, tw); tw = cmul(tw, wk);
                }
                U o0, o1;
                if constexpr (S == 2) { TU::set(o0, 0, y[0]); TU::set(o0, 1, y[1]); TU::set(o1, 0, y[2]); TU::set(o1, 1, y[3]); }
                else { TU::set(o0, 0, y[0]); TU::set(o1, 0, y[1]); }
                U* dp = reinterpret_cast<U*>(dst + (unsigned long long)k1 * N2 + 2ull * mc);
                __builtin_nontemporal_store(o0, dp);
                __builtin_nontemporal_store(o1, dp + 1);
            }
        } else if constexpr (RMODE == 3) {
            static_assert(RMODE != 3 || (!SEQC && DIR == FWD && !OINT && !IINT && PP == 8 && !RAG && !SWZ), "real pair pass in the tile: forward row pass on 128-byte tiles");
            constexpr int H = C / 2;
            const unsigned a = (unsigned)ebase, N1 = D.rn1;
            CX* imgc = reinterpret_cast<CX*>(img);
            auto at = [&](int pt, int seq) -> CX& { return imgc[pt * (PITCH * S) + seq]; };
            // packed spectrum Z -> half-complex X (fft_stock.h / fft_one.h): S = (A + conj B) / 2, D = -(i/2) W_N^k (A - conj B), A = Z[k], B = Z[n-k]
            // The arithmetic of the pair sweeps of the three-sweep route, operation for operation (fft_big.h big_block_kernel / real_pair_kernel): the
            // pair (k, n - k) is evaluated from its SMALLER index, W_N^k through the same exact-argument evaluation (pair_root) - so that this pass
            // and complex core + pair sweep give the same bits, and the unordered transform may keep its own three sweeps (pair + internal layout in
            // one): pffft_transform_ordered == pffft_zreorder(pffft_transform) bit for bit (benchmarks/bench_pffft.c:343-349) whichever route runs
            const long long nn = (long long)N1 * L;
            auto pairx = [&](unsigned k, CX& Zk, CX& Zm) {               // Zk = Z[k] -> X[k], Zm = Z[n - k] -> X[n - k]
                const bool low = 2ull * k <= (unsigned long long)nn;
                const long long ks = low ? (long long)k : nn - (long long)k;
                const CX A = low ? Zk : Zm, Bn = low ? Zm : Zk;
                const CX wk = pair_root<T>(ks, nn);
                const CX Sm = add_conj(A, Bn) * (T)0.5, Dm = cmul(sub_conj(A, Bn) * (T)0.5, wk);
                const CX Xa = add_rot<FWD>(Sm, Dm), Xb = conj(sub_rot<FWD>(Sm, Dm));
                Zk = low ? Xa : Xb; Zm = low ? Xb : Xa;
            };
            static_assert(RMODE != 3 || ((H * L) % WG == 0 && WG % H == 0), "pairs per thread, one sequence per thread");
            const int c = tid % H;
            const unsigned k1 = a * H + (unsigned)c;
#pragma unroll
            for (int i = 0; i < H * L / WG; ++i) {
                const int k2 = tid / H + i * (WG / H);
                if (a == 0 && c == 0) {
                    // row 0 pairs with itself, (0, k2) <-> (0, L - k2); bin 0 carries (DC, Nyquist) (include/pffft/pffft.h:144-152), bin n/2 is conj Z
                    if (k2 == 0) { const CX Z = at(0, 0); at(0, 0) = mk<T>(Z.x + Z.y, Z.x - Z.y); }
                    else if (2 * k2 < L) { CX Zk = at(k2, 0), Zm = at(L - k2, 0); pairx(N1 * (unsigned)k2, Zk, Zm); at(k2, 0) = Zk; at(L - k2, 0) = Zm; }
                    else if (2 * k2 == L) at(k2, 0) = conj(at(k2, 0));
                    // ... and so does row N1/2 (the last sequence of tile 0): (N1/2, k2) <-> (N1/2, L - 1 - k2), bin N1/2 + N1 k2
                    if (2 * k2 < L) { CX Zk = at(k2, C - 1), Zm = at(L - 1 - k2, C - 1); pairx(N1 / 2 + N1 * (unsigned)k2, Zk, Zm); at(k2, C - 1) = Zk; at(L - 1 - k2, C - 1) = Zm; }
                } else {
                    const int ps = C - 1 - c;                          // the sequence that holds row N1 - k1
                    CX Zk = at(k2, c), Zm = at(L - 1 - k2, ps);
                    pairx(k1 + N1 * (unsigned)k2, Zk, Zm);
                    at(k2, c) = Zk; at(L - 1 - k2, ps) = Zm;
                }
            }
            __syncthreads();
            // bins k1 + N1 k2: per point two runs of H adjacent k1 (no streaming hint: the mirror run shares its lines with the neighbour tile's).
            // Float: the first run as 16-byte units; the mirror run starts one element off the 16-byte grid - its inner pairs as units, the two end
            // elements alone (eight-byte stores of all sixteen: 16 instead of 9 store instructions per thread)
            if constexpr (S == 2) {
                constexpr int IPP = 4 + 3 + 2;                          // items per point
#pragma unroll
                for (int i = 0; i < (IPP * L + WG - 1) / WG; ++i) {
                    const int g = tid + i * WG;
                    if (g < IPP * L) {
                        const int pt = g / IPP, j = g % IPP;
                        CX* row = dst + (unsigned long long)pt * N1;
                        if (j < 4) {                                    // sequences 2j, 2j + 1 -> bins aH + 2j, + 1
                            *reinterpret_cast<U*>(row + a * H + 2 * j) = img[pt * PITCH + j];
                        } else if (j < 7) {                             // sequences 9 + 2 (j - 4), + 1 -> an aligned pair inside the mirror run
                            const int sq = 9 + 2 * (j - 4);
                            if (a == 0 && sq + 1 == C - 1) {            // (tile 0: the last sequence is row N1/2, not the run's end)
                                row[mirror_row(a, sq)] = at(pt, sq);
                            } else {
                                U o; TU::set(o, 0, at(pt, sq)); TU::set(o, 1, at(pt, sq + 1));
                                *reinterpret_cast<U*>(row + mirror_row(a, sq)) = o;
                            }
                        } else if (j == 7) {
                            row[mirror_row(a, 8)] = at(pt, 8);
                        } else {
                            row[mirror_row(a, C - 1)] = at(pt, C - 1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < C * L / WG; ++i) {
                    const int g = tid + i * WG, pt = g / C, seq = g % C;
                    dst[(unsigned long long)pt * N1 + mirror_row(a, seq)] = at(pt, seq);
                }
            }
        } else if constexpr (RMODE == 2) {
            static_assert(RMODE != 2 || (!SEQC && DIR == FWD && R0 == 1 && !OINT && !IINT), "real Hermitian store: forward row pass");
            const unsigned N1 = D.rn1, k10 = (unsigned)ebase * (unsigned)C;   // ebase = tile index a along the rows
            const CX* imgc = reinterpret_cast<const CX*>(img);
            // direct half, k2 < L/2: bin k1 + N1 k2, units of S adjacent k1
#pragma unroll
            for (int i = 0; i < (L / 2) * PP / WG; ++i) {
                const int g = tid + i * WG, pt = g / PP, pu = g % PP;
                U x = img[pt * PITCH + pu];
                const unsigned k1 = k10 + (unsigned)(S * pu);
                if (k1 == 0 && pt == 0) {            // bin 0 = (DC, Nyquist): X[0] and X[N/2] = row 0, k2 = L/2, both real
                    const CX ny = imgc[(L / 2) * (PITCH * S)];
                    CX b0 = TU::get(x, 0);
                    b0.y = ny.x;
                    TU::set(x, 0, b0);
                }
                CX* dp = dst + (unsigned long long)pt * N1 + k1;
                if (k1 + (S - 1) <= N1 / 2) __builtin_nontemporal_store(x, reinterpret_cast<U*>(dp));
                else if (k1 <= N1 / 2) *dp = TU::get(x, 0);                  // (the last row N1/2 of the last tile: its unit partner does not exist)
            }
            // mirrored half, k2 >= L/2, rows 0 < k1 < N1/2: conj X -> bin (N1 - k1) + N1 (L - 1 - k2).  The run of a tile's C rows is
            // DESCENDING in k1 and off by one element against the 16-byte grid (N - k): float stores the C/2 - 1 aligned pairs inside
            // it as 16-byte units and only the two end elements alone (8-byte stores of all 16 measured the pass at 630 us per GiB
            // against 441 for the plain row pass); double: an element is a unit
            if (k10 + (unsigned)C <= N1 / 2 + (unsigned)C - 1 && k10 < N1 / 2) {      // (the last tile holds row N1/2 only: nothing to mirror)
                if constexpr (S == 2) {
                    constexpr int IPP = C / 2 + 1;                                   // items per point: C/2 - 1 units + 2 end elements
#pragma unroll
                    for (int i = 0; i < ((L / 2) * IPP + WG - 1) / WG; ++i) {
                        const int g = tid + i * WG;
                        if (g >= (L / 2) * IPP) break;
                        const int pt = L / 2 + g / IPP, j = g % IPP;
                        const CX* row = imgc + pt * (PITCH * S);
                        CX* rbase = dst + (unsigned long long)N1 * (unsigned)(L - 1 - pt) + (N1 - k10 - (unsigned)(C - 1));   // bin of sequence C - 1
                        if (j < C / 2 - 1) {                                          // bins rbase + 1 + 2j, + 2 + 2j = sequences C - 2 - 2j, C - 3 - 2j
                            const unsigned sa = (unsigned)(C - 2 - 2 * j);
                            if (k10 + sa - 1 < N1 / 2) {                              // (both rows below N1/2; k10 + sa - 1 >= 1 always)
                                U o;
                                TU::set(o, 0, conj(row[sa])); TU::set(o, 1, conj(row[sa - 1]));
                                __builtin_nontemporal_store(o, reinterpret_cast<U*>(rbase + 1 + 2 * j));
                            } else if (k10 + sa < N1 / 2) {
                                rbase[1 + 2 * j] = conj(row[sa]);
                            }
                        } else if (j == C / 2 - 1) {                                  // sequence C - 1: the run's first bin
                            if (k10 + (unsigned)(C - 1) < N1 / 2) rbase[0] = conj(row[C - 1]);
                        } else {                                                      // sequence 0: the run's last bin (row 0 has no mirror)
                            if (k10 >= 1) rbase[C - 1] = conj(row[0]);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < (L / 2) * C / WG; ++i) {
                        const int g = tid + i * WG, pt = L / 2 + g / C, sq = g % C;
                        const unsigned k1 = k10 + (unsigned)sq;
                        if (k1 >= 1 && k1 < N1 / 2)
                            dst[(unsigned long long)(N1 - k1) + (unsigned long long)N1 * (unsigned)(L - 1 - pt)] = conj(imgc[pt * (PITCH * S) + sq]);
                    }
                }
            }
        } else if constexpr (OINT) {
            // one item = ONE 16-byte unit of the layout, consecutive lanes = consecutive units: every store instruction is
            // dense (two stores per lane, 16 bytes each at a 32-byte stride, measured 0.25 against 0.31 for the canonical store).  The units of a point k' < L/4 form a run of C/4 whole blocks.
            constexpr int BPT = C / 4, UPB = 2 * (int)sizeof(T);               // blocks per point; 16-byte units per block: 8 (float) / 16 (double)
            constexpr int UPP = BPT * UPB, UPQ = UPB / 4, UPS = 4 / S;        // units per point / per (group, quarter) slot; image units per group
            static_assert(((L / 4) * UPP) % WG == 0, "store units per thread");
#pragma unroll
            for (int i = 0; i < (L / 4) * UPP / WG; ++i) {
                const int g = tid + i * WG, ptq = g / UPP, r = g % UPP, bb = r / UPB, m = (r / UPQ) % 4, sub = r % UPQ;
                if (bb * 4 >= pv * S) continue;                                      // (ragged last tile: these blocks do not exist)
                const U* sp = img + (ptq + m * (L / 4)) * PITCH + m * G::QSKEW + bb * UPS;
                // internal layout in complex units: 4 t + 4 m + 2 p (+ l / 2), t = canonical index of the group's first bin in quarter 0
                U* dp = reinterpret_cast<U*>(dst + 4 * (ebase + (unsigned long long)ptq * D.ops + 4 * bb)) + r % UPB;
                U o;
                if constexpr (S == 2) {           // sub = part p: (re or im) of the four sequences of the group
                    const U u0 = SWZ ? img[ua(ptq + m * (L / 4), bb * UPS)] : sp[0], u1 = SWZ ? img[ua(ptq + m * (L / 4), bb * UPS + 1)] : sp[1];
                    if (sub) { o.x = u0.y; o.y = u0.w; o.z = u1.y; o.w = u1.w; }
                    else { o.x = u0.x; o.y = u0.z; o.z = u1.x; o.w = u1.z; }
                } else {                          // sub = 2 p + (l / 2): part p of sequences 2 (l/2), 2 (l/2) + 1
                    const U u0 = SWZ ? img[ua(ptq + m * (L / 4), bb * UPS + 2 * (sub & 1))] : sp[2 * (sub & 1)];
                    const U u1 = SWZ ? img[ua(ptq + m * (L / 4), bb * UPS + 2 * (sub & 1) + 1)] : sp[2 * (sub & 1) + 1];
                    if (sub >> 1) { o.x = u0.y; o.y = u1.y; }
                    else { o.x = u0.x; o.y = u1.x; }
                }
                __builtin_nontemporal_store(o, dp);
            }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = tid + i * WG, pt = g / PP, pu = g % PP;
            if (pu < pv) __builtin_nontemporal_store(img[ua(pt, pu)], reinterpret_cast<U*>(dst + (unsigned long long)pt * D.ops + S * pu));
        }
        }
        // next tile: the group's next one, or the first of the next group (whose successor was published at this group's start)
        unsigned long long gnn = gnext;
        if (sub + 1 == K) gnn = dyn ? grabbed(s_next[gi & 1]) : gnext + gstride;
        __syncthreads();
        tile = tile1;
        if (++sub == K) { sub = 0; gcur = gnext; gnext = gnn; ++gi; }
        tile1 = sub + 1 < K ? tile + 1 : gnext * K;
    }
    if (dyn && tid == 0) {
        __threadfence();
        const unsigned dn = atomicAdd(&ctr[xctr ? 8 : 1], 1u);
        if (dn == gridDim.x - 1) {
            if (xctr) { for (int i = 0; i < 9; ++i) atomicExch(&ctr[i], 0u); }
            else { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
        }
    }
}

}  // namespace pf
