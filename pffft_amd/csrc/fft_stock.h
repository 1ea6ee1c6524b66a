// Mixed-radix Stockham kernel with a run-time plan: every n = 2^a 3^b 5^c whose two ping-pong images fit in
// LDS (n <= 9216 complex float points, 4608 double), real or complex, forward or backward, canonical or
// pffft-internal layout.  It takes over the non-power-of-two sizes from the in-place radix-2..5 kernel of
// fft_generic.h (whose passes cost one LDS sweep + one barrier each: 4-7 of them for the sizes of the
// reference's benchmark list, benchmarks/bench_pffft.c:445).
//
// Same reference functions as fft_generic.h / fft_tiled.h (src/pffft_priv_impl.h:122-901 butterflies and
// pass drivers, :1158-1462 reorder / finalize / preprocess, :1465-1532 transform_internal).
//
// Shape (the design of fft_tiled.h with the plan as data instead of template parameters):
//   * radices 3,4,5,6,8,10,12,15,16 (6,10,12,15 as twiddle-free Good-Thomas products, cxmath.h) -> 2-4
//     stages for every size up to 9216;
//   * Stockham autosort, twiddles on the inputs:  stage s (radix R, Ns = product of earlier radices),
//     butterfly j < n/R reads x[j + q n/R] W_{Ns R}^(q (j mod Ns)), writes y[(j div Ns) Ns R + (j mod Ns) + d Ns];
//     the last stage writes the canonical spectrum straight to HBM (8/16-byte accesses, consecutive lanes on
//     consecutive points);
//   * exchanges ping-pong between two LDS images -> one barrier per stage; the image written by stage s is
//     padded by pad[s] points per block of Ns R points (chosen on the host by simulating the bank
//     conflicts of the write and of the following read);
//   * WAVE SPECIALISATION: the workgroup is C compute threads + P producer wavefronts.  A producer issues the
//     16-byte loads of the NEXT group of vectors at the top of an iteration, sits through the iteration's
//     barriers with the loads in flight, and deposits them into the free image just before the closing
//     barrier.  (First version, every wave loading its own stage-0 operands: the loads were in flight for
//     one phase out of four, ~24 KB per CU on average = 2.4 TB/s by Little's law, measured 2.3-2.6.)
//   * pffft-internal layout through an image of the layout itself (32-scalar blocks padded to IBS scalars):
//     linear 16-byte chunks on the HBM side, scalar picks on the butterfly side (complex) or in the pair
//     pass (real);
//   * the backward transform is conj o forward o conj, so there is one set of stage bodies per precision;
//   * groups of G consecutive vectors are pulled in order from the work counter (see fft_c1024.h).
#pragma once
#include <type_traits>
#include "cxmath.h"
#include "fft_generic.h"  // fdiv

namespace pf {

constexpr int SK_MAX_STAGES = 4;
constexpr int SK_NCHP = 16;  // 16-byte chunks a producer lane holds in registers

// everything a stage needs, precomputed on the host (stock_plan.h): the kernel keeps the structs of all stages
// in SGPRs for its whole life (first version: derived per stage per iteration with integer divisions and
// indexed kernarg loads - 900 SALU + 1400 VALU instructions per 1024-point transform, 62 % of the wave
// cycles waiting)
struct StockStage {
    int R, nb, Ns;           // radix, butterflies per vector (n / R), product of the earlier radices
    int rpad, rstride;       // LDS source: operand q of butterfly j at j + (j div Ns) rpad + q rstride
    int wblk;                // LDS destination: result d at (j div Ns) wblk + (j mod Ns) + d Ns
    int twstep;              // W_{Ns R}^(q jm) = W_n^(q jm twstep)
    unsigned m_nb, m_Ns;     // magic multipliers: x div d = umulhi(x, m) for x < 65536
    int tw_off;              // twmode 2: this stage's Ns base twiddles W_{Ns R}^jm start here in the compact table
};

struct StockPlan {
    int n, ns;       // complex points per transform, stages
    int G;           // transforms per workgroup pass
    int img;         // complex points per image slot (largest padded image)
    int twmode;      // 0: every twiddle from the padded W_n^j table in LDS; 2: base twiddle W_{Ns R}^jm from the compact
                     // per-stage table in LDS (ctab entries: sum of Ns over the stages), powers recomputed (<= 4
                     // products deep); 1: base twiddle from the global W_n^j table, powers recomputed
    int ctab;        // entries of the compact table
    int ibs;         // scalars per padded 32-scalar block of the internal-layout image (36 float / 34 double; 32 when LDS is tight)
    int twr_lds;     // W_N^k of the real pair pass: 1 = the whole table (n/2 + 1 entries) in LDS; 2 = two small tables in LDS,
                     // W_N^k = W_N^(k mod 64) W_N^(k - k mod 64) (64 + n/128 + 1 entries, one product per pair: where the whole
                     // table does not fit next to the images); 0 = read from the global table (L2) per pair
    int C, P;        // compute threads, producer wavefronts (blockDim = C + 64 P)
    unsigned m_n4, m_per, m_nchk;  // magic multipliers for n/4, n/2 + 1, 16-byte chunks per vector
    int sym, sym_items; unsigned m_sym; // real transforms: symmetric spectrum-side stage (pairs in registers) with (nb + 1) / 2
                                   // work items - or, sym == 0, a separate pair phase over a natural image
    StockStage st[SK_MAX_STAGES];
};

template <typename T> struct StockLds { size_t buf = 0, tab = 0, twr = 0, next = 0, total = 0; };
template <typename T> __host__ __device__ constexpr StockLds<T> stock_lds(const StockPlan& p) {
    StockLds<T> l{};
    size_t o = 0;
    l.buf = o; o += (size_t)2 * p.G * p.img * sizeof(cx<T>);
    l.tab = o; if (p.twmode == 0) o += ((size_t)p.n + (p.n >> 5) + 1) * sizeof(cx<T>);
    if (p.twmode == 2) o += (size_t)(p.ctab + 1) * sizeof(cx<T>);
    l.twr = o; if (p.twr_lds == 1) o += ((size_t)p.n / 2 + 1) * sizeof(cx<T>);
    if (p.twr_lds == 2) o += ((size_t)64 + p.n / 128 + 2) * sizeof(cx<T>);
    l.next = o; o += 16;
    l.total = o;
    return l;
}

enum { SK_G = 0, SK_L = 1, SK_I = 2 };  // operand source / result destination: HBM, LDS image, LDS internal-layout image

template <typename T> struct SkIbs { static constexpr int v = 32 + (sizeof(T) == 4 ? 4 : 2); };

__device__ __forceinline__ int udiv(int x, unsigned m) { return (int)__umulhi((unsigned)x, m); }

template <typename T> struct SkArgs {
    int tid, nthr;               // worker thread index / count (workgroup's compute threads, or one wavefront)
    int slot0;                   // first image slot / vector of the group this worker owns
    int n, total, maxtotal;      // total = vectors of this worker * nb butterflies (<= maxtotal, the plan's bound:
                                 // the loops run to the plan's bound under a predicate so that a compile-time plan
                                 // gives compile-time trip counts)
    unsigned m_n4;
    int img;                     // slot stride of the LDS images (complex points)
    int twmode, ibs;
    bool iswz;                   // internal-layout image: unpadded blocks with swizzled groups (sk_igrp)
    bool cj_in, cj_out;          // conjugate on the way in (first stage) / out (HBM store): backward transform
    // LDS is addressed as (one base pointer) + integer offsets: a pointer selected between the two images or
    // between an LDS and a global table becomes a generic pointer and every access a FLAT instruction
    cx<T>* lds;                  // start of dynamic LDS
    int src_off, dst_off, tab_off;   // source image, destination image, W_n^j table (complex points)
    cx<T>* gdst;                 // vector 0 of the group (stride n)
    const cx<T>* twg;            // global W_n^j table (twmode 1; in twmode 2 the kernel is handed the compact table)
    // real transforms: W_N^k of the pair pass (LDS copy at twr_off, or the global table), work items of the symmetric stage
    const cx<T>* twrg; int twr_off; int twr_lds; int sym_items; unsigned m_sym; int cnt, maxcnt;
};

__device__ __forceinline__ int tpad(int i) { return i + (i >> 5); }

// The internal-layout image: blocks of 32 scalars = eight groups of 4 [quarter qq of the spectrum: real parts in group 2 qq, imaginary parts in
// 2 qq + 1], a block padded to ibs scalars (36 float / 34 double) so that the scalar scatter - four lanes per block, sixteen blocks per wavefront -
// walks the banks.  Where LDS has no room for the padding (ibs == 32: n = 7200 ... 10000 float, 4608 double; sk_iswz) every lane of such an access met the
// same four banks until round 5 (complex n = 9216 float forward 0.75 ordered, 0.65 unordered); there the groups of block b now sit at group ^ (b & 7).
__device__ __forceinline__ int sk_igrp(int ibs, bool swz, int blk, int grp) { return ibs * blk + 4 * (swz ? (grp ^ (blk & 7)) : grp); }
// ... and the imaginary part of a scalar whose real part sits at ire: the next group - under the swizzle the group index with bit 0 flipped
__device__ __forceinline__ int sk_iim(bool swz, int ire) { return swz ? (ire ^ 4) : ire + 4; }
// (not for the one double plan that also reads its stage twiddles from L2 - n = 4800, LDS filled to the last KiB, 168 VGPRs: the index arithmetic of the
//  swizzle costs it 40 B more scratch per lane and 0.04-0.05, measured)
template <typename T> __host__ __device__ constexpr bool sk_iswz(const StockPlan& p) { return p.ibs == 32 && !(sizeof(T) == 8 && p.twmode == 1); }
// linear 16-byte chunk cc of the layout -> chunk offset inside the block image
template <typename T> __device__ __forceinline__ int sk_ichunk(int cc, int ibs, bool swz) {
    constexpr int CH = 16 / (int)sizeof(T), CPB = 32 / CH, GPC = 4 / CH;   // scalars per chunk, chunks per block, chunks per group of 4 scalars
    const int b = cc / CPB, c = cc % CPB;
    const int cs = swz ? (((c / GPC) ^ (b & 7)) * GPC + c % GPC) : c;
    return b * (ibs / CH) + cs;
}

template <typename T, int R, int SRC, int DST>
__device__ __forceinline__ void sk_stage(const StockStage& st, const SkArgs<T>& a) {
    typedef cx<T> CX;
    const int IBS = a.ibs;
    const int n4 = a.n >> 2;
#pragma unroll
    for (int i0 = 0; i0 < a.maxtotal; i0 += a.nthr) {
        const int i = i0 + a.tid;
        if (i >= a.total) continue;
        const int gl = udiv(i, st.m_nb), j = i - gl * st.nb, g = a.slot0 + gl;
        int jd = j, jm = 0;
        if (st.Ns > 1) { jd = udiv(j, st.m_Ns); jm = j - jd * st.Ns; }
        CX v[R];
        // ---- operands
        if constexpr (SRC == SK_L) {
            const CX* p = a.lds + a.src_off + g * a.img + j + jd * st.rpad;
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = p[q * st.rstride];
        } else {  // complex spectrum in the internal layout: point P = j + q nb sits in quarter P div n/4
            const T* p = reinterpret_cast<const T*>(a.lds + a.src_off + g * a.img);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                int qq, r;
                if constexpr (R % 4 == 0) { qq = q / (R / 4); r = j + (q % (R / 4)) * st.nb; }
                else { const int P = j + q * st.nb; qq = udiv(P, a.m_n4); r = P - qq * n4; }
                const int ip = sk_igrp(IBS, a.iswz, r >> 2, 2 * qq) + (r & 3);
                v[q] = mk<T>(p[ip], p[sk_iim(a.iswz, ip)]);
            }
        }
        if (a.cj_in) {
#pragma unroll
            for (int q = 0; q < R; ++q) v[q].y = -v[q].y;
        }
        // ---- twiddles W_{Ns R}^(q jm) = W_n^(q jm twstep)
        if (st.Ns > 1) {
            const int k = jm * st.twstep;
            if (a.twmode == 0) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], a.lds[a.tab_off + tpad(q * k)]);
            } else if constexpr (R > 16) {
                // radix 32: w^q = (w^4)^(q div 4) w^(q mod 4), five live values instead of R
                CX p1;
                if (a.twmode == 2) p1 = a.lds[a.tab_off + st.tw_off + jm];
                else { p1 = a.twg[k]; asm volatile(""); }
                const CX w2 = cmul(p1, p1), w3 = cmul(w2, p1), w4 = cmul(w2, w2);
                CX blk = w4;
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    const int aa = q >> 2, b = q & 3;
                    const CX wb = b == 1 ? p1 : (b == 2 ? w2 : w3);
                    CX t;
                    if (aa == 0) t = wb;
                    else t = b == 0 ? blk : cmul(blk, wb);
                    v[q] = cmul(v[q], t);
                    if (aa >= 1 && b == 3) blk = cmul(blk, w4);
                }
            } else {
                CX p[R < 4 ? 4 : R];
                // (the empty asm keeps the two loads in separate blocks: merged into one load through a selected
                // pointer they become FLAT instructions)
                if (a.twmode == 2) p[1] = a.lds[a.tab_off + st.tw_off + jm];
                else { p[1] = a.twg[k]; asm volatile(""); }
                p[2] = cmul(p[1], p[1]);
                if constexpr (R > 3) p[3] = cmul(p[2], p[1]);
                if constexpr (R > 4) p[4] = cmul(p[2], p[2]);
                if constexpr (R > 5) p[5] = cmul(p[4], p[1]);
                if constexpr (R > 6) p[6] = cmul(p[3], p[3]);
                if constexpr (R > 7) p[7] = cmul(p[4], p[3]);
                if constexpr (R > 8) p[8] = cmul(p[4], p[4]);
                if constexpr (R > 9) p[9] = cmul(p[8], p[1]);
                if constexpr (R > 10) p[10] = cmul(p[5], p[5]);
                if constexpr (R > 11) p[11] = cmul(p[8], p[3]);
                if constexpr (R > 12) p[12] = cmul(p[6], p[6]);
                if constexpr (R > 13) p[13] = cmul(p[8], p[5]);
                if constexpr (R > 14) p[14] = cmul(p[7], p[7]);
                if constexpr (R > 15) p[15] = cmul(p[8], p[7]);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], p[q]);
            }
        }
        dftR<R, FWD>(v);
        // ---- results
        if constexpr (DST == SK_G) {  // last stage: Ns = nb, result d is bin j + d nb
            CX* p = a.gdst + (size_t)g * a.n + j;
            if (a.cj_out) {
#pragma unroll
                for (int d = 0; d < R; ++d) v[d].y = -v[d].y;
            }
#pragma unroll
            for (int d = 0; d < R; ++d) __builtin_nontemporal_store(v[d], p + d * st.nb);
        } else if constexpr (DST == SK_L) {
            CX* p = a.lds + a.dst_off + g * a.img + jd * st.wblk + jm;
#pragma unroll
            for (int d = 0; d < R; ++d) p[d * st.Ns] = v[d];
        } else {  // last stage, forward, complex: bin j + d nb into the internal-layout image
            T* p = reinterpret_cast<T*>(a.lds + a.dst_off + g * a.img);
#pragma unroll
            for (int d = 0; d < R; ++d) {
                int qq, r;
                if constexpr (R % 4 == 0) { qq = d / (R / 4); r = j + (d % (R / 4)) * st.nb; }
                else { const int P = j + d * st.nb; qq = udiv(P, a.m_n4); r = P - qq * n4; }
                const int ip = sk_igrp(IBS, a.iswz, r >> 2, 2 * qq) + (r & 3);
                p[ip] = v[d].x;
                p[sk_iim(a.iswz, ip)] = v[d].y;
            }
        }
    }
}

// scalar index (real part; imaginary part + 4) of half-complex bin k of a REAL transform inside the
// internal-layout image: odd quarters run backwards (bin_of, fft_generic.h)
// (padded block images only, ibs != 32 - fft_aux.h: the imaginary part sits 4 scalars behind the real part)
template <typename T> __device__ __forceinline__ int sk_iposr(int k, int n4, unsigned m_n4, int ibs) {
    const int qq = udiv(k, m_n4), r = k - qq * n4;
    const int tt = (qq & 1) ? (r ? n4 - r : 0) : r;
    return ibs * (tt >> 2) + 8 * qq + (tt & 3);
}
template <typename T> __device__ __forceinline__ void sk_iposr(int k, int n4, unsigned m_n4, int ibs, bool swz, int& ire, int& iim) {
    const int qq = udiv(k, m_n4), r = k - qq * n4;
    const int tt = (qq & 1) ? (r ? n4 - r : 0) : r;
    ire = sk_igrp(ibs, swz, tt >> 2, 2 * qq) + (tt & 3);
    iim = sk_iim(swz, ire);
}

// operands of butterfly j of a stage whose source is an LDS image, twiddled and transformed (the body of sk_stage)
template <typename T, int R>
__device__ __forceinline__ void sk_bfly(const StockStage& st, const SkArgs<T>& a, int g, int j, cx<T> (&v)[R]) {
    typedef cx<T> CX;
    int jd = j, jm = 0;
    if (st.Ns > 1) { jd = udiv(j, st.m_Ns); jm = j - jd * st.Ns; }
    const CX* p = a.lds + a.src_off + g * a.img + j + jd * st.rpad;
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = p[q * st.rstride];
    if (st.Ns > 1) {
        const int k = jm * st.twstep;
        if (a.twmode == 0) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul(v[q], a.lds[a.tab_off + tpad(q * k)]);
        } else {
            CX p1;
            if (a.twmode == 2) p1 = a.lds[a.tab_off + st.tw_off + jm];
            else { p1 = a.twg[k]; asm volatile(""); }
            // w^q = (w^4)^(q div 4) w^(q mod 4): five live values instead of R (two butterflies' worth of registers
            // are in use here), every power at most 5 products deep
            const CX w2 = cmul(p1, p1), w3 = cmul(w2, p1), w4 = cmul(w2, w2);
            CX blk = w4;
#pragma unroll
            for (int q = 1; q < R; ++q) {
                const int aa = q >> 2, b = q & 3;
                const CX wb = b == 1 ? p1 : (b == 2 ? w2 : w3);
                CX t;
                if (aa == 0) t = wb;
                else t = b == 0 ? blk : cmul(blk, wb);
                v[q] = cmul(v[q], t);
                if (aa >= 1 && b == 3) blk = cmul(blk, w4);
            }
        }
    }
    dftR<R, FWD>(v);
}

// W_N^k of the pair pass, 0 <= k <= n/2 (StockPlan::twr_lds).  Mode 2: the sizes whose whole table does not fit next to the two
// images (real N >= 15360 float, 7776 double) read it from L2 once per pair until round 5 - a dependent global load in the middle of
// an LDS phase, the latency of which nothing covers: N = 17280 float real 0.46-0.52 against 0.76-0.78 for its complex core.
template <typename T> __device__ __forceinline__ cx<T> sk_twr(const cx<T>* lds, int twr_off, int mode, const cx<T>* twrg, int k) {
    if (mode == 1) return lds[twr_off + k];
    if (mode == 2) return cmul(lds[twr_off + (k & 63)], lds[twr_off + 64 + (k >> 6)]);
    cx<T> w = twrg[k];
    asm volatile("");
    return w;
}
template <typename T> __device__ __forceinline__ void sk_twr_fill(cx<T>* lds, int twr_off, int mode, const cx<T>* twrg, int n, int tid, int nthreads) {
    if (mode == 1)
        for (int i = tid; i <= n / 2; i += nthreads) lds[twr_off + i] = twrg[i];
    if (mode == 2)
        for (int i = tid; i < 64 + (n >> 7) + 1; i += nthreads) lds[twr_off + i] = twrg[i < 64 ? i : (i - 64) << 6];
}

template <typename T> __device__ __forceinline__ cx<T> sk_wN(const SkArgs<T>& a, int k) {   // W_N^k, 0 < k < n, N = 2n
    const int kk = 2 * k <= a.n ? k : a.n - k;
    const cx<T> w = sk_twr<T>(a.lds, a.twr_off, a.twr_lds, a.twrg, kk);
    return 2 * k <= a.n ? w : mk<T>(-w.x, w.y);   // W_N^k = -conj(W_N^(n-k))
}

// Real forward, last stage, SYMMETRIC assignment: a work item owns butterflies j and nb - j (item 0: butterfly 0 and,
// when nb is even, nb/2), i.e. bin k = j + d nb together with its mirror n - k = (nb - j) + (R-1-d) nb, so the pair pass
//   X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k]
// runs on registers and the half-complex spectrum goes straight to its destination (HBM, or the internal-layout
// image) - no natural image, no separate pair phase, one barrier less.
template <typename T, int R, int DST>
__device__ __forceinline__ void sk_last_real(const StockStage& st, const SkArgs<T>& a) {
    typedef cx<T> CX;
    const int n = a.n, nb = st.nb, n4 = n >> 2, items = a.sym_items;
    auto pairf = [&](CX A, CX Bn, int k, CX& Xa, CX& Xb) {
        const CX wk = sk_wN<T>(a, k);
        const CX S = add_conj(A, Bn) * (T)0.5, Dm = cmul(sub_conj(A, Bn) * (T)0.5, wk);   // B = conj(Bn); D = -i Dm rides on the add
        Xa = add_rot<FWD>(S, Dm);
        Xb = conj(sub_rot<FWD>(S, Dm));
    };
#pragma unroll
    for (int i0 = 0; i0 < a.maxcnt * items; i0 += a.nthr) {
        const int i = i0 + a.tid;
        if (i >= a.cnt * items) continue;
        const int gl = udiv(i, a.m_sym), it = i - gl * items, g = a.slot0 + gl;
        const bool even = (nb & 1) == 0;
        const int j1 = it, j2 = it ? nb - it : (even ? nb >> 1 : 0);
        CX v1[R], v2[R];
        sk_bfly<T, R>(st, a, g, j1, v1);
        sk_bfly<T, R>(st, a, g, j2, v2);
        auto put = [&](int k, CX X) {
            if constexpr (DST == SK_G) {
                __builtin_nontemporal_store(X, a.gdst + (size_t)g * n + k);
            } else {
                T* pd = reinterpret_cast<T*>(a.lds + a.dst_off + g * a.img);
                int ire, iim;
                sk_iposr<T>(k, n4, a.m_n4, a.ibs, a.iswz, ire, iim);
                pd[ire] = X.x; pd[iim] = X.y;
            }
        };
        if (it) {
#pragma unroll
            for (int d = 0; d < R; ++d) {
                const int k = j1 + d * nb;
                CX Xa, Xb;
                pairf(v1[d], v2[R - 1 - d], k, Xa, Xb);
                put(k, Xa);
                put(n - k, Xb);
            }
        } else {
            put(0, mk<T>(v1[0].x + v1[0].y, v1[0].x - v1[0].y));   // (DC, Nyquist): include/pffft/pffft.h:144-152
#pragma unroll
            for (int d = 1; 2 * d < R; ++d) {
                CX Xa, Xb;
                pairf(v1[d], v1[R - d], d * nb, Xa, Xb);
                put(d * nb, Xa);
                put(n - d * nb, Xb);
            }
            if constexpr (R % 2 == 0) put(n >> 1, conj(v1[R / 2]));
            if (even) {
#pragma unroll
                for (int d = 0; 2 * d < R - 1; ++d) {
                    const int k = (nb >> 1) + d * nb;
                    CX Xa, Xb;
                    pairf(v2[d], v2[R - 1 - d], k, Xa, Xb);
                    put(k, Xa);
                    put(n - k, Xb);
                }
                if constexpr (R % 2 == 1) put(n >> 1, conj(v2[(R - 1) / 2]));
            }
        }
    }
}

// Real backward, first stage, symmetric assignment: the operands of butterflies j and nb - j are the mirror pairs
//   Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A + B, D = i conj(W_N^k) (A - B), A = X[k], B = conj X[n-k]
// built in registers from the deposited half-complex spectrum (natural or internal-layout image), conjugated because
// the stages run a forward transform (the store conjugates back).
template <typename T, int R, int SRC>
__device__ __forceinline__ void sk_first_real(const StockStage& st, const SkArgs<T>& a) {
    typedef cx<T> CX;
    const int n = a.n, nb = st.nb, n4 = n >> 2, items = a.sym_items;
#pragma unroll
    for (int i0 = 0; i0 < a.maxcnt * items; i0 += a.nthr) {
        const int i = i0 + a.tid;
        if (i >= a.cnt * items) continue;
        const int gl = udiv(i, a.m_sym), it = i - gl * items, g = a.slot0 + gl;
        const bool even = (nb & 1) == 0;
        const int j1 = it, j2 = it ? nb - it : (even ? nb >> 1 : 0);
        auto get = [&](int k) -> CX {
            if constexpr (SRC == SK_L) {
                return a.lds[a.src_off + g * a.img + k];
            } else {
                const T* ps = reinterpret_cast<const T*>(a.lds + a.src_off + g * a.img);
                int ire, iim;
                sk_iposr<T>(k, n4, a.m_n4, a.ibs, a.iswz, ire, iim);
                return mk<T>(ps[ire], ps[iim]);
            }
        };
        // conj Z'[k] -> za, conj Z'[n-k] -> zb
        auto pairb = [&](int k, CX& za, CX& zb) {
            const CX A = get(k), Bn = get(n - k), wk = sk_wN<T>(a, k);
            const CX S = add_conj(A, Bn), Dm = cmulc(sub_conj(A, Bn), wk);               // B = conj(Bn); D = i Dm rides on the add
            za = conj(add_rot<BWD>(S, Dm));
            zb = sub_rot<BWD>(S, Dm);
        };
        CX v1[R], v2[R];
        if (it) {
#pragma unroll
            for (int q = 0; q < R; ++q) pairb(j1 + q * nb, v1[q], v2[R - 1 - q]);
        } else {
            { const CX A = get(0); v1[0] = mk<T>(A.x + A.y, -(A.x - A.y)); }
#pragma unroll
            for (int q = 1; 2 * q < R; ++q) pairb(q * nb, v1[q], v1[R - q]);
            if constexpr (R % 2 == 0) { const CX A = get(n >> 1); v1[R / 2] = mk<T>((T)2 * A.x, (T)2 * A.y); }
#pragma unroll
            for (int q = 0; q < R; ++q) v2[q] = v1[q];   // (nb odd: butterfly 0 twice, the second copy is not stored)
            if (even) {
#pragma unroll
                for (int q = 0; 2 * q < R - 1; ++q) pairb((nb >> 1) + q * nb, v2[q], v2[R - 1 - q]);
                if constexpr (R % 2 == 1) { const CX A = get(n >> 1); v2[(R - 1) / 2] = mk<T>((T)2 * A.x, (T)2 * A.y); }
            }
        }
        dftR<R, FWD>(v1);
        dftR<R, FWD>(v2);
        CX* p1 = a.lds + a.dst_off + g * a.img + j1 * st.wblk;   // first stage: Ns = 1
#pragma unroll
        for (int d = 0; d < R; ++d) p1[d] = v1[d];
        if (it || even) {
            CX* p2 = a.lds + a.dst_off + g * a.img + j2 * st.wblk;
#pragma unroll
            for (int d = 0; d < R; ++d) p2[d] = v2[d];
        }
    }
}

// RC > 0: the radix as a compile-time constant (compile-time plans: one instantiation per stage); RC == 0: run-time dispatch
template <typename T, int SRC, int DST, int RC = 0>
__device__ __forceinline__ void sk_run(const StockStage& st, const SkArgs<T>& a) {
    if constexpr (RC > 0) {
        if constexpr (sizeof(T) == 4 || RC <= 12) sk_stage<T, RC, SRC, DST>(st, a);
        return;
    }
    switch (st.R) {
        case 3: sk_stage<T, 3, SRC, DST>(st, a); break;
        case 4: sk_stage<T, 4, SRC, DST>(st, a); break;
        case 5: sk_stage<T, 5, SRC, DST>(st, a); break;
        case 6: sk_stage<T, 6, SRC, DST>(st, a); break;
        case 8: sk_stage<T, 8, SRC, DST>(st, a); break;
        case 9: sk_stage<T, 9, SRC, DST>(st, a); break;
        case 10: sk_stage<T, 10, SRC, DST>(st, a); break;
        case 12: sk_stage<T, 12, SRC, DST>(st, a); break;
        default:
            if constexpr (sizeof(T) == 4) {  // double stops at radix 12 (register budget)
                if (st.R == 15) sk_stage<T, 15, SRC, DST>(st, a);
                else if (st.R == 24) sk_stage<T, 24, SRC, DST>(st, a);
                else if (st.R == 32) sk_stage<T, 32, SRC, DST>(st, a);
                else sk_stage<T, 16, SRC, DST>(st, a);
            }
            break;
    }
}


template <typename T, int DST, int RC = 0>
__device__ __forceinline__ void sk_run_last_real(const StockStage& st, const SkArgs<T>& a) {
    if constexpr (RC > 0) {
        if constexpr (sizeof(T) == 4 || RC <= 12) sk_last_real<T, RC, DST>(st, a);
        return;
    }
    switch (st.R) {
        case 3: sk_last_real<T, 3, DST>(st, a); break;
        case 4: sk_last_real<T, 4, DST>(st, a); break;
        case 5: sk_last_real<T, 5, DST>(st, a); break;
        case 6: sk_last_real<T, 6, DST>(st, a); break;
        case 8: sk_last_real<T, 8, DST>(st, a); break;
        case 9: sk_last_real<T, 9, DST>(st, a); break;
        case 10: sk_last_real<T, 10, DST>(st, a); break;
        case 12: sk_last_real<T, 12, DST>(st, a); break;
        default:
            if constexpr (sizeof(T) == 4) {
                if (st.R == 15) sk_last_real<T, 15, DST>(st, a);
                else sk_last_real<T, 16, DST>(st, a);
            }
            break;
    }
}
template <typename T, int SRC, int RC = 0>
__device__ __forceinline__ void sk_run_first_real(const StockStage& st, const SkArgs<T>& a) {
    if constexpr (RC > 0) {
        if constexpr (sizeof(T) == 4 || RC <= 12) sk_first_real<T, RC, SRC>(st, a);
        return;
    }
    switch (st.R) {
        case 3: sk_first_real<T, 3, SRC>(st, a); break;
        case 4: sk_first_real<T, 4, SRC>(st, a); break;
        case 5: sk_first_real<T, 5, SRC>(st, a); break;
        case 6: sk_first_real<T, 6, SRC>(st, a); break;
        case 8: sk_first_real<T, 8, SRC>(st, a); break;
        case 9: sk_first_real<T, 9, SRC>(st, a); break;
        case 10: sk_first_real<T, 10, SRC>(st, a); break;
        case 12: sk_first_real<T, 12, SRC>(st, a); break;
        default:
            if constexpr (sizeof(T) == 4) {
                if (st.R == 15) sk_first_real<T, 15, SRC>(st, a);
                else sk_first_real<T, 16, SRC>(st, a);
            }
            break;
    }
}

// One pass of the compute side over `cnt` vectors (image slots slot0 .. slot0 + cnt - 1) whose input sits in
// image w ^ 1: (real backward pair pass) -> stages -> (real forward pair pass) -> (internal-layout copy-out).
// WL = false: the worker is the workgroup's compute threads, phases are separated by __syncthreads.
// WL = true : the worker is ONE wavefront that owns its slots, phases are separated by wave-local fences.
// Every LDS-writing phase flips w; on return image w is free.
template <bool WL> __device__ __forceinline__ void sk_sync() {
    if constexpr (WL) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

template <typename T> struct SkCtx {
    cx<T>* lds;
    int bufsz, tab_off, twr_off;
    bool in_int, out_int, bwd, real;
    int twr_lds;
    const cx<T>* twg;
    const cx<T>* twrg;
};

template <int I> struct SkIdx { static constexpr int value = I; };
// radix of stage S of a compile-time plan (-1: the plan has no such stage); 0 for a run-time plan (PT = void)
template <class PT, int S> constexpr int sk_ct_radix() {
    if constexpr (std::is_void<PT>::value) return 0;
    else return S < PT::value.ns ? PT::value.st[S].R : -1;
}

// DF = 1: the first stage ran outside (operands straight from HBM, sk_df_body) and its results sit in image w ^ 1;
// DF = 2: the pair pass of the real backward transform ran outside, the packed spectrum sits in image w ^ 1
template <typename T, bool WL, class PT = void, int DF = 0>
__device__ __forceinline__ void sk_iteration(const StockPlan& p, const StockStage& st0, const StockStage& st1,
                                             const StockStage& st2, const StockStage& st3, const SkCtx<T>& c,
                                             int wtid, int wn, int slot0, int cnt, int maxcnt, cx<T>* gout, int& w,
                                             int stid = -1, int sn = 0) {
    typedef cx<T> CX;
    typedef vec4<float> chunk16;
    // worker index / count of the phases that STORE TO HBM (sk_df_body: the wavefronts that hold prefetched loads never
    // store, so their s_waitcnt vmcnt for the loads does not wait for store acknowledgements); default: all of them
    if (sn == 0) { stid = wtid; sn = wn; }
    CX* const lds = c.lds;
    const int n = p.n, ns = p.ns, bufsz = c.bufsz;
    const int n4 = n >> 2, half = n >> 1, per = half + 1;
    const int nchk = (int)((size_t)n * sizeof(CX) / 16);       // 16-byte chunks per vector
    const int img16 = (int)((size_t)p.img * sizeof(CX) / 16);  // slot stride in chunks (img is even)
    const bool in_int = c.in_int, out_int = c.out_int, bwd = c.bwd, real = c.real;
    // Wave-local kernels, canonical output (forward ordered, both backward transforms): the last phase writes a natural-order
    // LDS image and the result leaves as linear 16-byte chunks, like the internal layout does - dense 1 KiB store instructions
    // instead of 8-byte stores in runs of n / R (complex) or ascending + descending runs (real pair phase) that are short for
    // the small vectors these kernels serve (round 3: the internal-layout path, with its extra phase, had measured FASTER than
    // the direct stores - real N = 96 .. 384 forward 0.67-0.68 ordered against 0.73-0.77 unordered).  Measured (1 GiB per
    // launch): float real N = 128 / 160 / 192 forward ordered 0.66-0.68 -> 0.74-0.75, float complex backward +0.02-0.06, double
    // forward ordered +0.02-0.07; the double BACKWARD transforms lose 0.03-0.05 (a complex double is a 16-byte store per lane
    // already) and keep their direct stores.
    const bool via_img = WL && !out_int && !(real && !bwd && p.sym) && (sizeof(T) == 4 || !bwd);

    // ---- real backward: half-complex spectrum X -> conj of the packed spectrum Z' (the stages then run
    //      a forward transform; the final conjugation happens on the store):
    //      Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A + B, D = i conj(W_N^k) (A - B), A = X[k], B = conj X[n-k]
    if (DF != 2 && real && bwd && !p.sym) {
#pragma unroll
        for (int id0 = 0; id0 < maxcnt * per; id0 += wn) {
            const int id = id0 + wtid;
            if (id >= cnt * per) continue;
            const int gl = udiv(id, p.m_per), k = id - gl * per, g = slot0 + gl;
            CX A, Bn;
            if (in_int) {
                const T* ps = reinterpret_cast<const T*>(lds + (w ^ 1) * bufsz + g * p.img);
                int ire, iim;
                sk_iposr<T>(k, n4, p.m_n4, p.ibs, sk_iswz<T>(p), ire, iim);
                A = mk<T>(ps[ire], ps[iim]);
                if (k != 0 && k != half) {
                    sk_iposr<T>(n - k, n4, p.m_n4, p.ibs, sk_iswz<T>(p), ire, iim);
                    Bn = mk<T>(ps[ire], ps[iim]);
                } else Bn = A;
            } else {
                const CX* ps = lds + (w ^ 1) * bufsz + g * p.img;
                A = ps[k];
                Bn = (k != 0 && k != half) ? ps[n - k] : A;
            }
            CX* pd = lds + w * bufsz + g * p.img;
            if (k == 0) {
                pd[0] = mk<T>(A.x + A.y, -(A.x - A.y));
            } else if (k == half) {
                pd[half] = mk<T>((T)2 * A.x, (T)2 * A.y);  // conj(2 conj(A))
            } else {
                const CX wk = sk_twr<T>(lds, c.twr_off, c.twr_lds, c.twrg, k);
                const CX S = add_conj(A, Bn), Dm = cmulc(sub_conj(A, Bn), wk);   // B = conj(Bn); D = i Dm rides on the add
                pd[k] = conj(add_rot<BWD>(S, Dm));
                pd[n - k] = sub_rot<BWD>(S, Dm);  // conj(conj(S - D))
            }
        }
        w ^= 1;
        sk_sync<WL>();
    }
    // ---- stages (unrolled over the at most SK_MAX_STAGES stage structs held in SGPRs)
    {
        SkArgs<T> a;
        a.tid = wtid; a.nthr = wn; a.slot0 = slot0; a.n = n; a.m_n4 = p.m_n4; a.img = p.img; a.twmode = p.twmode; a.ibs = p.ibs; a.iswz = sk_iswz<T>(p);
        a.lds = lds; a.tab_off = c.tab_off; a.gdst = gout; a.twg = c.twg;
        a.cj_out = bwd;
        a.twrg = c.twrg; a.twr_off = c.twr_off; a.twr_lds = c.twr_lds; a.sym_items = p.sym_items; a.m_sym = p.m_sym;
        a.cnt = cnt; a.maxcnt = maxcnt;
        // RC: the stage's radix where the plan is a compile-time constant (one instantiation per stage instead of one per
        // radix: with radices 20 / 24 / 25 the run-time dispatch outgrew the unroller's budget), 0 = run-time dispatch
        auto stage = [&](auto RCt, int s) __attribute__((always_inline)) {
            constexpr int RC = decltype(RCt)::value;
            if constexpr (RC >= 0) {
                if (s < ns) {
                    const StockStage& st = s == 0 ? st0 : s == 1 ? st1 : s == 2 ? st2 : st3;
                    a.total = cnt * st.nb; a.maxtotal = maxcnt * st.nb;
                    a.src_off = (w ^ 1) * bufsz; a.dst_off = w * bufsz;
                    a.cj_in = (s == 0) && bwd && !real;   // real backward: the pair pass / symmetric first stage conjugates
                    if (s < ns - 1) {
                        if (s == 0 && real && bwd && p.sym) {   // half-complex spectrum -> first butterflies, pairs in registers
                            if (in_int) sk_run_first_real<T, SK_I, RC>(st, a); else sk_run_first_real<T, SK_L, RC>(st, a);
                        }
                        else if (s == 0 && in_int && !real) sk_run<T, SK_I, SK_L, RC>(st, a);
                        else sk_run<T, SK_L, SK_L, RC>(st, a);
                        w ^= 1;
                        sk_sync<WL>();
                    } else if (real && !bwd && p.sym) {    // last butterflies -> half-complex spectrum, pairs in registers
                        if (out_int) { sk_run_last_real<T, SK_I, RC>(st, a); w ^= 1; sk_sync<WL>(); }
                        else { a.tid = stid; a.nthr = sn; sk_run_last_real<T, SK_G, RC>(st, a); a.tid = wtid; a.nthr = wn; }
                    } else if (real && !bwd) {             // natural image for the pair phase below
                        sk_run<T, SK_L, SK_L, RC>(st, a); w ^= 1; sk_sync<WL>();
                    } else {
                        if (out_int) { sk_run<T, SK_L, SK_I, RC>(st, a); w ^= 1; sk_sync<WL>(); }
                        else if (via_img) { sk_run<T, SK_L, SK_L, RC>(st, a); w ^= 1; sk_sync<WL>(); }   // natural image, copy-out below
                        else { a.tid = stid; a.nthr = sn; sk_run<T, SK_L, SK_G, RC>(st, a); a.tid = wtid; a.nthr = wn; }
                    }
                }
            }
        };
        static_assert(SK_MAX_STAGES == 4, "one call per stage below");
        if constexpr (std::is_void<PT>::value) {
#pragma unroll
            for (int s = 0; s < SK_MAX_STAGES; ++s) stage(SkIdx<0>{}, s);
        } else {
            if constexpr (DF != 1) stage(SkIdx<sk_ct_radix<PT, 0>()>{}, 0);
            stage(SkIdx<sk_ct_radix<PT, 1>()>{}, 1);
            stage(SkIdx<sk_ct_radix<PT, 2>()>{}, 2); stage(SkIdx<sk_ct_radix<PT, 3>()>{}, 3);
        }
    }
    // ---- real forward: packed spectrum Z (natural image) -> half-complex spectrum X:
    //      X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k]
    if (real && !bwd && !p.sym) {
        const int ptid = out_int ? wtid : stid, pn = out_int ? wn : sn;   // canonical output: this phase stores to HBM
#pragma unroll
        for (int id0 = 0; id0 < maxcnt * per; id0 += pn) {
            const int id = id0 + ptid;
            if (id >= cnt * per) continue;
            const int gl = udiv(id, p.m_per), k = id - gl * per, g = slot0 + gl;
            const CX* ps = lds + (w ^ 1) * bufsz + g * p.img;
            CX Xa, Xb;
            if (k == 0) {
                const CX A = ps[0];
                Xa = mk<T>(A.x + A.y, A.x - A.y);  // (DC, Nyquist): include/pffft/pffft.h:144-152
                Xb = Xa;
            } else if (k == half) {
                Xa = conj(ps[half]);
                Xb = Xa;
            } else {
                const CX A = ps[k], Bn = ps[n - k];
                const CX wk = sk_twr<T>(lds, c.twr_off, c.twr_lds, c.twrg, k);
                const CX S = add_conj(A, Bn) * (T)0.5, Dm = cmul(sub_conj(A, Bn) * (T)0.5, wk);   // B = conj(Bn); D = -i Dm rides on the add
                Xa = add_rot<FWD>(S, Dm);
                Xb = conj(sub_rot<FWD>(S, Dm));
            }
            if (out_int) {
                T* pd = reinterpret_cast<T*>(lds + w * bufsz + g * p.img);
                int ire, iim;
                sk_iposr<T>(k, n4, p.m_n4, p.ibs, sk_iswz<T>(p), ire, iim);
                pd[ire] = Xa.x; pd[iim] = Xa.y;
                if (k != 0 && k != half) {
                    sk_iposr<T>(n - k, n4, p.m_n4, p.ibs, sk_iswz<T>(p), ire, iim);
                    pd[ire] = Xb.x; pd[iim] = Xb.y;
                }
            } else if (via_img) {
                CX* pd = lds + w * bufsz + g * p.img;
                pd[k] = Xa;
                if (k != 0 && k != half) pd[n - k] = Xb;
            } else {
                CX* pd = gout + (size_t)g * n;
                __builtin_nontemporal_store(Xa, pd + k);
                if (k != 0 && k != half) __builtin_nontemporal_store(Xb, pd + (n - k));
            }
        }
        if (out_int || via_img) { w ^= 1; sk_sync<WL>(); }
    }
    // ---- internal-layout output (padded block image) / canonical output of the wave-local kernels (natural image): the
    //      image leaves as linear 16-byte chunks; the backward transforms conjugate here (the stages ran a forward transform)
    if (out_int || via_img) {
        const chunk16* s16 = reinterpret_cast<const chunk16*>(lds + (w ^ 1) * bufsz);
        chunk16* d16 = reinterpret_cast<chunk16*>(gout);
#pragma unroll
        for (int cb = 0; cb < maxcnt * nchk; cb += sn) {
            const int cc0 = cb + stid;
            if (cc0 >= cnt * nchk) continue;
            const int gl = udiv(cc0, p.m_nchk), cc = cc0 - gl * nchk, g = slot0 + gl;
            chunk16 v = s16[g * img16 + (out_int ? sk_ichunk<T>(cc, p.ibs, sk_iswz<T>(p)) : cc)];
            if (via_img && bwd) {
                if constexpr (sizeof(T) == 4) { v.y = -v.y; v.w = -v.w; }
                else {
                    vec2<double> dv = __builtin_bit_cast(vec2<double>, v);
                    dv.y = -dv.y;
                    v = __builtin_bit_cast(chunk16, dv);
                }
            }
            __builtin_nontemporal_store(v, d16 + (size_t)g * nchk + cc);
        }
    }
}

// Work distribution.  ctr == nullptr: static, group = blockIdx + it * gridDim.  Otherwise CHUNKS of K consecutive
// groups are pulled in order from the counter by one owner thread, one chunk ahead, through a ring of three LDS
// slots (published at the top of the iteration at whose end it is consumed).  K keeps the atomic rate down: all
// workgroups hit ONE address, which serves ~80 M atomics/s - with one 19 KiB group per atomic the n = 2400 kernel
// ran at 0.37 of the roofline, statically scheduled at 0.64.
struct SkSched {
    unsigned gcur, gnx, sub, rslot, wslot, pend, K;
    bool dyn;
    unsigned* ctr;
    unsigned* s_next;
    // (Round 4: the register-tiled kernels take their first groups statically - fft_tiled.h, no start-up burst of atomics, C3 0.71 -> 0.74.  The
    //  same change HERE measured the in-order plans 10-12 % slower at 1 GiB per launch - n = 8192 complex float 0.79 -> 0.68, double n = 2048 /
    //  4096 0.77-0.80 -> 0.68-0.72 - and neither a run-time switch back to these three grabs nor a staggered start brought it back, only this
    //  source: the start-up grabs stay.)
    __device__ __forceinline__ void grab(bool owner) {   // before the first barrier
        if (dyn && owner) {
            s_next[0] = atomicAdd(&ctr[0], 1u);
            s_next[1] = atomicAdd(&ctr[0], 1u);
            pend = atomicAdd(&ctr[0], 1u);
        }
    }
    __device__ __forceinline__ void start() {            // after it
        if (!dyn) { gcur = blockIdx.x; gnx = gcur + gridDim.x; sub = 0; rslot = wslot = 0; return; }
        gcur = K * s_next[0];
        if (K > 1) { gnx = gcur + 1; sub = 1; rslot = 1; }
        else { gnx = s_next[1]; sub = 0; rslot = 2; }
        wslot = 2;
    }
    __device__ __forceinline__ void top(bool owner) {    // top of an iteration
        if (dyn && sub + 1 >= K) {
            if (owner) { s_next[wslot] = pend; pend = atomicAdd(&ctr[0], 1u); }
            wslot = wslot == 2 ? 0 : wslot + 1;
        }
    }
    __device__ __forceinline__ void advance() {          // after the iteration's closing barrier
        gcur = gnx;
        if (!dyn) gnx += gridDim.x;
        else if (sub + 1 < K) { ++gnx; ++sub; }
        else { gnx = K * s_next[rslot]; rslot = rslot == 2 ? 0 : rslot + 1; sub = 0; }
    }
    __device__ __forceinline__ void finish(bool owner) {
        if (dyn && owner) {
            __threadfence();
            unsigned d = atomicAdd(&ctr[1], 1u);
            if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
        }
    }
};

// linear 16-byte chunk c of a group -> chunk offset inside the images (natural image, or internal-layout block image)
template <typename T> __device__ __forceinline__ int sk_chunk_off(int g, int cc, int img16, bool in_int, int ibs, bool swz) {
    return g * img16 + (in_int ? sk_ichunk<T>(cc, ibs, swz) : cc);
}

// flags: bit0 input in internal layout, bit1 output in internal layout, bit2 backward, bit3 real
//
// Workgroup-phase kernel: C compute threads share every stage of the G vectors of a group; P producer wavefronts
// hold the next group in registers across the iteration's barriers.
template <typename T, class PT = void>
__device__ __forceinline__ void sk_wg_body(const T* in, T* out, size_t batch, const StockPlan& p, int flags,
                                           const cx<T>* __restrict__ twg, const cx<T>* __restrict__ twrg, unsigned* ctr,
                                           unsigned kchunk) {
    typedef cx<T> CX;
    typedef vec4<float> chunk16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const StockLds<T> L = stock_lds<T>(p);
    SkCtx<T> c;
    c.lds = reinterpret_cast<CX*>(smem_raw);
    CX* const lds = c.lds;
    c.bufsz = p.G * p.img;          // image b starts at complex offset b * bufsz (L.buf == 0)
    c.tab_off = (int)(L.tab / sizeof(CX)); c.twr_off = (int)(L.twr / sizeof(CX));
    c.in_int = flags & 1; c.out_int = flags & 2; c.bwd = flags & 4; c.real = flags & 8;
    c.twg = twg; c.twrg = twrg;
    c.twr_lds = c.real ? p.twr_lds : 0;
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + L.next);
    const int n = p.n, G = p.G, ns = p.ns, C = p.C;
    const int tid = threadIdx.x, nthreads = p.C + 64 * p.P;
    if (p.twmode == 0)
        for (int i = tid; i < n; i += nthreads) lds[c.tab_off + tpad(i)] = twg[i];
    if (p.twmode == 2)
        for (int i = tid; i < p.ctab; i += nthreads) lds[c.tab_off + i] = twg[i];
    sk_twr_fill<T>(lds, c.twr_off, c.twr_lds, twrg, n, tid, nthreads);
    const int nchk = (int)((size_t)n * sizeof(CX) / 16);
    const int img16 = (int)((size_t)p.img * sizeof(CX) / 16);
    // LDS-writing compute phases per iteration (each followed by one barrier); + the closing barrier
    const int Fc = p.sym ? (ns - 1) + (c.out_int ? 1 : 0)
                         : ((c.real && c.bwd) ? 1 : 0) + (ns - 1) + (((c.real && !c.bwd) || c.out_int) ? 1 : 0) +
                               ((c.real && !c.bwd && c.out_int) ? 1 : 0);

    SkSched sch;
    sch.dyn = ctr != nullptr; sch.ctr = ctr; sch.s_next = s_next; sch.K = kchunk; sch.pend = 0;
    sch.grab(tid == C);
    __syncthreads();
    sch.start();
    __syncthreads();
    int w = 0;  // image the next LDS-writing phase writes; the phase after it reads image w ^ 1 (ping-pong)

    if (tid >= C) {
        // ======================================================================= producer wavefronts
        const int pl = tid - C, NPT = 64 * p.P;
        const chunk16* s16 = reinterpret_cast<const chunk16*>(in);
        chunk16 raw[SK_NCHP];
        // unconditional loads (a predicated load costs a branch and pins its address): chunks beyond the
        // group's vectors re-read its last chunk, groups beyond the batch re-read chunk 0 of the input
        auto issue = [&](unsigned grp) {
            const size_t t0 = (size_t)grp * G;
            const int cnt = t0 < batch ? (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G) : 0;
            const int lastc = cnt ? cnt * nchk - 1 : 0;
            const chunk16* s = cnt ? s16 + t0 * nchk : s16;
#pragma unroll
            for (int i = 0; i < SK_NCHP; ++i) {
                const int cix = pl + NPT * i;
                raw[i] = __builtin_nontemporal_load(s + (cix < lastc ? cix : lastc));
            }
        };
        auto deposit = [&](int boff) {
            chunk16* d16 = reinterpret_cast<chunk16*>(lds + boff);
            const int tot = G * nchk;  // chunks beyond the group's vectors hold stale values nobody reads
#pragma unroll
            for (int i = 0; i < SK_NCHP; ++i) {
                const int cix = pl + NPT * i;
                if (cix < tot) {
                    const int g = udiv(cix, p.m_nchk), cc = cix - g * nchk;
                    d16[sk_chunk_off<T>(g, cc, img16, c.in_int, p.ibs, sk_iswz<T>(p))] = raw[i];
                }
            }
        };
        issue(sch.gcur);
        deposit(w * c.bufsz);
        w ^= 1;
        __syncthreads();
        while ((size_t)sch.gcur * G < batch) {
            issue(sch.gnx);
            sch.top(tid == C);
            for (int b = 0; b < Fc; ++b) __syncthreads();
            w ^= (Fc & 1);
            deposit(w * c.bufsz);
            w ^= 1;
            __syncthreads();
            sch.advance();
        }
        sch.finish(tid == C);
        return;
    }

    // =========================================================================== compute wavefronts
    const StockStage st0 = p.st[0], st1 = p.st[1], st2 = p.st[2], st3 = p.st[3];
    w ^= 1;
    __syncthreads();  // first deposit
    while ((size_t)sch.gcur * G < batch) {
        const size_t t0 = (size_t)sch.gcur * G;
        sch.top(false);
        const int g_here = (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G);
        CX* gout = reinterpret_cast<CX*>(out) + t0 * n;
        sk_iteration<T, false, PT>(p, st0, st1, st2, st3, c, tid, C, 0, g_here, G, gout, w);
        // ---- closing barrier: the producers have deposited the next group into image w
        w ^= 1;
        __syncthreads();
        sch.advance();
    }
}

// Wave-local kernel (small n): every wavefront owns G / (waves per workgroup) image slots and runs ALL phases on
// them between wave-local fences - no workgroup barrier inside a transform, wavefronts drift apart and hide each
// other's LDS / HBM latency.  Each wavefront prefetches its own slots of the next group into SK_NCHW registers
// per lane.  One __syncthreads per iteration hands the next group index over.
constexpr int SK_NCHW = 8;
constexpr int SK_WL_WAVES = 4;  // wavefronts per workgroup of the wave-local kernel
template <typename T, class PT = void>
__device__ __forceinline__ void sk_wl_body(const T* in, T* out, size_t batch, const StockPlan& p, int flags,
                                           const cx<T>* __restrict__ twg, const cx<T>* __restrict__ twrg, unsigned* ctr,
                                           unsigned kchunk) {
    typedef cx<T> CX;
    typedef vec4<float> chunk16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const StockLds<T> L = stock_lds<T>(p);
    SkCtx<T> c;
    c.lds = reinterpret_cast<CX*>(smem_raw);
    CX* const lds = c.lds;
    c.bufsz = p.G * p.img;
    c.tab_off = (int)(L.tab / sizeof(CX)); c.twr_off = (int)(L.twr / sizeof(CX));
    c.in_int = flags & 1; c.out_int = flags & 2; c.bwd = flags & 4; c.real = flags & 8;
    c.twg = twg; c.twrg = twrg;
    c.twr_lds = c.real ? p.twr_lds : 0;
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + L.next);
    const int n = p.n, G = p.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Gw = G / SK_WL_WAVES, slot0 = wave * Gw;
    if (p.twmode == 0)
        for (int i = tid; i < n; i += 64 * SK_WL_WAVES) lds[c.tab_off + tpad(i)] = twg[i];
    if (p.twmode == 2)
        for (int i = tid; i < p.ctab; i += 64 * SK_WL_WAVES) lds[c.tab_off + i] = twg[i];
    sk_twr_fill<T>(lds, c.twr_off, c.twr_lds, twrg, n, tid, 64 * SK_WL_WAVES);
    const int nchk = (int)((size_t)n * sizeof(CX) / 16);
    const int img16 = (int)((size_t)p.img * sizeof(CX) / 16);
    const StockStage st0 = p.st[0], st1 = p.st[1], st2 = p.st[2], st3 = p.st[3];

    SkSched sch;
    sch.dyn = ctr != nullptr; sch.ctr = ctr; sch.s_next = s_next; sch.K = kchunk; sch.pend = 0;
    sch.grab(tid == 0);
    __syncthreads();
    sch.start();
    __syncthreads();

    const chunk16* s16 = reinterpret_cast<const chunk16*>(in);
    chunk16 raw[SK_NCHW];
    // this wavefront's vectors of group grp: count, and their chunks into registers (clamped, unconditional)
    auto mine = [&](unsigned grp) -> int {
        const size_t t0 = (size_t)grp * G + slot0;
        return t0 < batch ? (int)((batch - t0) < (size_t)Gw ? (batch - t0) : (size_t)Gw) : 0;
    };
    auto issue = [&](unsigned grp) {
        const int cnt = mine(grp);
        const int lastc = cnt ? cnt * nchk - 1 : 0;
        const chunk16* s = cnt ? s16 + ((size_t)grp * G + slot0) * nchk : s16;
#pragma unroll
        for (int i = 0; i < SK_NCHW; ++i) {
            const int cix = lane + 64 * i;
            raw[i] = __builtin_nontemporal_load(s + (cix < lastc ? cix : lastc));
        }
    };
    auto deposit = [&](int boff) {
        chunk16* d16 = reinterpret_cast<chunk16*>(lds + boff);
        const int tot = Gw * nchk;
#pragma unroll
        for (int i = 0; i < SK_NCHW; ++i) {
            const int cix = lane + 64 * i;
            if (cix < tot) {
                const int gl = udiv(cix, p.m_nchk), cc = cix - gl * nchk;
                d16[sk_chunk_off<T>(slot0 + gl, cc, img16, c.in_int, p.ibs, sk_iswz<T>(p))] = raw[i];
            }
        }
    };
    int w = 0;
    issue(sch.gcur);
    deposit(w * c.bufsz);
    w ^= 1;
    sk_sync<true>();
    while ((size_t)sch.gcur * G < batch) {
        sch.top(tid == 0);
        issue(sch.gnx);
        const int cnt = mine(sch.gcur);
        CX* gout = reinterpret_cast<CX*>(out) + (size_t)sch.gcur * G * n;
        sk_iteration<T, true, PT>(p, st0, st1, st2, st3, c, lane, 64, slot0, cnt, Gw, gout, w);
        sk_sync<true>();          // the last phase's LDS reads are done before ...
        deposit(w * c.bufsz);     // ... the next group lands in the free image
        w ^= 1;
        // orders the deposit against the next iteration's reads; the workgroup barrier is only needed to hand the
        // next chunk index over (statically scheduled wavefronts never meet inside the loop)
        if (sch.dyn) __syncthreads(); else sk_sync<true>();
        sch.advance();
    }
    sch.finish(tid == 0);
}


// ---- "direct first stage" variants (compile-time plans, natural-layout input, not the real backward transform):
// every work item of the FIRST stage loads its R operands straight from HBM into registers - one group ahead, the loads
// fly during the remaining phases - and the first stage runs from those registers.  Against the bodies above this saves
// the producer wavefronts, the deposit into LDS, the first stage's LDS reads and two of the workgroup barriers per
// iteration (deposit hand-over and closing barrier: the next first stage writes the image the last phase does not read).
// The real backward transform (canonical input) does the same with its pair pass, which is its first phase.
// Consecutive work items read consecutive points (one 8 / 16-byte load per lane, 512 / 1024 contiguous bytes per
// wavefront and operand as long as n / R >= 64).  Static group assignment only.
// threads of the first stage that carry prefetched loads when the roles are split (0: not split): single-vector groups of
// the workgroup kernel, one round, not the real backward transform (its pair pass loads on every wavefront)
__host__ __device__ constexpr int sk_df_loaders(const StockPlan& p, int flags, bool wl) {
    if (wl || p.G != 1 || ((flags & 8) && (flags & 4))) return 0;
    const int s0 = (p.st[0].nb + 63) / 64 * 64;
    return (s0 <= p.C && s0 + 128 <= 1024) ? s0 : 0;
}
// workgroup size of the direct-first-stage kernels
__host__ __device__ constexpr int sk_df_threads(const StockPlan& p, int flags, bool wl) {
    if (wl) return 64 * SK_WL_WAVES;
    const int s0 = sk_df_loaders(p, flags, wl);
    if (!s0) return p.C;
    // loaders + as many storing threads as the last stage has butterflies (at least the plan's compute threads)
    const int want = s0 + (p.st[p.ns - 1].nb + 63) / 64 * 64;
    return want > 1024 ? 1024 : (want < p.C ? p.C : want);
}

template <typename T, class PT, bool WL, int flags>
__device__ __forceinline__ void sk_df_body(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twg,
                                           const cx<T>* __restrict__ twrg) {
    typedef cx<T> CX;
    constexpr StockPlan p = PT::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr StockLds<T> L = stock_lds<T>(p);
    SkCtx<T> c;
    c.lds = reinterpret_cast<CX*>(smem_raw);
    CX* const lds = c.lds;
    c.bufsz = p.G * p.img;
    c.tab_off = (int)(L.tab / sizeof(CX)); c.twr_off = (int)(L.twr / sizeof(CX));
    c.in_int = false; c.out_int = flags & 2; c.bwd = flags & 4; c.real = flags & 8;
    c.twg = twg; c.twrg = twrg;
    c.twr_lds = c.real ? p.twr_lds : 0;
    constexpr int n = p.n, G = p.G, R0 = p.st[0].R, nb0 = p.st[0].nb, wblk0 = p.st[0].wblk;
    constexpr int NT = sk_df_threads(p, flags, WL);         // threads of the workgroup
    constexpr int W = WL ? 64 : NT;                        // threads of one worker (a wavefront / the workgroup)
    constexpr int GV = WL ? G / SK_WL_WAVES : G;           // vectors per worker and group
    constexpr int ROUNDS = (GV * nb0 + W - 1) / W;
    // split roles (workgroup kernel, single-vector groups): the wavefronts of the first stage's work items - the ones with
    // loads in flight - never store to HBM; every storing phase runs on the others.  Otherwise the compiler's s_waitcnt for
    // the prefetched operands (stores are conditional, so it must assume none were issued: vmcnt(0)) waits for the previous
    // iteration's store acknowledgements at the top of every iteration (n = 4000: 0.48 against 0.59 with producers).
    constexpr int S0 = sk_df_loaders(p, flags, WL);
    const int stid = S0 ? (threadIdx.x >= S0 ? (int)threadIdx.x - S0 : (1 << 28)) : -1;
    const int sn = S0 ? NT - S0 : 0;
    const int tid = threadIdx.x;
    const int wtid = WL ? (tid & 63) : tid;
    const int slot0 = WL ? (tid >> 6) * GV : 0;
    if (p.twmode == 0)
        for (int i = tid; i < n; i += NT) lds[c.tab_off + tpad(i)] = twg[i];
    if (p.twmode == 2)
        for (int i = tid; i < p.ctab; i += NT) lds[c.tab_off + i] = twg[i];
    sk_twr_fill<T>(lds, c.twr_off, c.twr_lds, twrg, n, tid, NT);
    const StockStage st0 = p.st[0], st1 = p.st[1], st2 = p.st[2], st3 = p.st[3];
    const CX* gin = reinterpret_cast<const CX*>(in);
    const bool cj = c.bwd && !c.real;

    // this worker's vectors of group grp
    auto mine = [&](size_t grp) -> int {
        const size_t t0 = grp * G + slot0;
        return t0 < batch ? (int)((batch - t0) < (size_t)GV ? (batch - t0) : (size_t)GV) : 0;
    };
    size_t gcur = blockIdx.x;
    const size_t gstep = gridDim.x;
    int w = 0;
    if (c.real && c.bwd) {
        // ---- real backward, canonical half-complex input: the work items of the PAIR pass load X[k] and X[n-k] straight from
        // HBM (ascending / descending runs) one group ahead; the pass writes the packed spectrum into image w, all stages follow
        constexpr int per = n / 2 + 1, half = n / 2;
        constexpr int RP = (GV * per + W - 1) / W;
        CX pa[RP], pb[RP];
        auto issue = [&](size_t grp) {
            const int cnt = mine(grp);
            const CX* base = cnt ? gin + (grp * G + slot0) * n : gin;
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                const int i = wtid + r * W;
                const int ii = i < cnt * per ? i : 0;
                const int g = ii / per, k = ii - g * per;
                pa[r] = __builtin_nontemporal_load(base + g * n + k);
                pb[r] = __builtin_nontemporal_load(base + g * n + (k ? n - k : 0));
            }
        };
        auto pairs = [&](int cnt, int w) {
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                const int i = wtid + r * W;
                if (i < cnt * per) {
                    const int g = i / per, k = i - g * per;
                    const CX A = pa[r];
                    CX* pd = lds + w * c.bufsz + (slot0 + g) * p.img;
                    if (k == 0) {
                        pd[0] = mk<T>(A.x + A.y, -(A.x - A.y));
                    } else if (k == half) {
                        pd[half] = mk<T>((T)2 * A.x, (T)2 * A.y);
                    } else {
                        const CX wk = sk_twr<T>(lds, c.twr_off, c.twr_lds, twrg, k);
                        const CX S = add_conj(A, pb[r]), Dm = cmulc(sub_conj(A, pb[r]), wk);   // B = conj(pb); D = i Dm rides on the add
                        pd[k] = conj(add_rot<BWD>(S, Dm));
                        pd[n - k] = sub_rot<BWD>(S, Dm);
                    }
                }
            }
        };
        issue(gcur);
        __syncthreads();   // tables
        while (gcur * G < batch) {
            const int cnt = mine(gcur);
            CX* gout = reinterpret_cast<CX*>(out) + gcur * G * n;
            pairs(cnt, w);
            issue(gcur + gstep);
            w ^= 1;
            sk_sync<WL>();
            sk_iteration<T, WL, PT, 2>(p, st0, st1, st2, st3, c, wtid, W, slot0, cnt, GV, gout, w, stid, sn);
            gcur += gstep;
        }
        return;
    }
    CX pre[ROUNDS][R0];
    // unconditional loads: work items beyond the group's vectors re-read operand 0 of its first vector, groups beyond
    // the batch the first vector of the input
    auto issue = [&](size_t grp) {
        if (S0 && (int)threadIdx.x >= S0) return;   // split roles: the storing wavefronts carry no loads
        const int cnt = mine(grp);
        const CX* base = cnt ? gin + (grp * G + slot0) * n : gin;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = wtid + r * W;
            const int ii = i < cnt * nb0 ? i : 0;
            const int g = ii / nb0, j = ii - g * nb0;
            const CX* src = base + g * n + j;
#pragma unroll
            for (int q = 0; q < R0; ++q) pre[r][q] = __builtin_nontemporal_load(src + q * nb0);
        }
    };
    auto stage0 = [&](int cnt, int w) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int i = wtid + r * W;
            if (i < cnt * nb0) {
                const int g = i / nb0, j = i - g * nb0;
                CX v[R0];
#pragma unroll
                for (int q = 0; q < R0; ++q) { v[q] = pre[r][q]; if (cj) v[q].y = -v[q].y; }
                dftR<R0, FWD>(v);
                CX* pd = lds + w * c.bufsz + (slot0 + g) * p.img + j * wblk0;
#pragma unroll
                for (int d = 0; d < R0; ++d) pd[d] = v[d];
            }
        }
    };
    issue(gcur);
    __syncthreads();   // tables
    while (gcur * G < batch) {
        const int cnt = mine(gcur);
        CX* gout = reinterpret_cast<CX*>(out) + gcur * G * n;
        stage0(cnt, w);
        issue(gcur + gstep);
        w ^= 1;
        sk_sync<WL>();
        sk_iteration<T, WL, PT, 1>(p, st0, st1, st2, st3, c, wtid, W, slot0, cnt, GV, gout, w, stid, sn);
        gcur += gstep;
    }
}

template <typename T>
__global__ void __launch_bounds__(1024)
fft_stock_kernel(const T* in, T* out, size_t batch, StockPlan p, int flags, const cx<T>* __restrict__ twg,
                 const cx<T>* __restrict__ twrg, unsigned* ctr, unsigned kchunk) {
    sk_wg_body<T>(in, out, batch, p, flags, twg, twrg, ctr, kchunk);
}

template <typename T>
__global__ void __launch_bounds__(256)
fft_stock_wl_kernel(const T* in, T* out, size_t batch, StockPlan p, int flags, const cx<T>* __restrict__ twg,
                    const cx<T>* __restrict__ twrg, unsigned* ctr, unsigned kchunk) {
    sk_wl_body<T>(in, out, batch, p, flags, twg, twrg, ctr, kchunk);
}

// The same body with the plan and the flags as compile-time constants (stock_plans_gen.h, written by
// tools/gen_stock_plans): radix dispatch, index arithmetic, LDS offsets and loop trip counts fold away, what is
// left per butterfly is its loads, twiddle products, the DFT and its stores.  (The run-time plan costs ~100
// VALU + ~60 SALU instructions per point and is issue-bound at 0.3 of the HBM roofline whatever the
// organisation - workgroup phases, producer wavefronts or wave-local all measured 0.26-0.44.)
// wavefronts per SIMD the register allocator should leave room for: what LDS lets a CU hold, capped at 5 (<= 96 VGPRs)
// (a kernel that spills at that budget gets a lower cap in stock_wpe_gen.h, written by tools/tune_stock_wpe.py from
// the compiler's resource remarks: squeezed into spilling, N = 640 fell from 0.60 to 0.55)
template <class PT, int FLAGS> struct SkWpeCap { static constexpr int v = 5; };
template <typename T> constexpr int sk_waves_per_simd(const StockPlan& p, int threads, int cap) {
    const size_t lds = stock_lds<T>(p).total;
    int wgs = (int)(160 * 1024 / lds);
    wgs = wgs > 8 ? 8 : (wgs < 1 ? 1 : wgs);
    const int wpe = wgs * (threads / 64) / 4;
    return wpe > cap ? cap : (wpe < 1 ? 1 : wpe);
}

template <typename T, class PT, int FLAGS>
__global__ void __launch_bounds__(256, sk_waves_per_simd<T>(PT::value, 256, SkWpeCap<PT, FLAGS>::v))
fft_stock_wl_ct_kernel(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twg,
                       const cx<T>* __restrict__ twrg, unsigned* ctr, unsigned kchunk) {
    constexpr StockPlan p = PT::value;
    sk_wl_body<T, PT>(in, out, batch, p, FLAGS, twg, twrg, ctr, kchunk);
}

template <typename T, class PT, int FLAGS>
__global__ void __launch_bounds__(PT::value.C + 64 * PT::value.P, sk_waves_per_simd<T>(PT::value, PT::value.C + 64 * PT::value.P, SkWpeCap<PT, FLAGS>::v))
fft_stock_ct_kernel(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twg,
                    const cx<T>* __restrict__ twrg, unsigned* ctr, unsigned kchunk) {
    constexpr StockPlan p = PT::value;
    sk_wg_body<T, PT>(in, out, batch, p, FLAGS, twg, twrg, ctr, kchunk);
}

// direct-first-stage variants: FLAGS carries bit 4 (16) so that the occupancy overrides of stock_wpe_gen.h tell them apart
template <typename T, class PT, int FLAGS>
__global__ void __launch_bounds__(256, sk_waves_per_simd<T>(PT::value, 256, SkWpeCap<PT, FLAGS>::v))
fft_stock_wl_df_ct_kernel(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twg,
                          const cx<T>* __restrict__ twrg, unsigned*, unsigned) {
    sk_df_body<T, PT, true, (FLAGS & 15)>(in, out, batch, twg, twrg);
}

template <typename T, class PT, int FLAGS>
__global__ void __launch_bounds__(sk_df_threads(PT::value, FLAGS & 15, false),
                                  sk_waves_per_simd<T>(PT::value, sk_df_threads(PT::value, FLAGS & 15, false), SkWpeCap<PT, FLAGS>::v))
fft_stock_df_ct_kernel(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twg,
                       const cx<T>* __restrict__ twrg, unsigned*, unsigned) {
    sk_df_body<T, PT, false, (FLAGS & 15)>(in, out, batch, twg, twrg);
}

}  // namespace pf
