// One HBM pass for the vectors that fill LDS ONCE but not twice (round 6): mixed-radix n with 80 KiB < vector <= 144 KiB - complex float
// n = 10240 ... 18432, double 5120 ... 9216, and the real transforms on those cores - which the two-image Stockham kernels of fft_stock.h
// cannot hold and which until now ran the two / three tile passes beyond LDS (0.27-0.38 of the roofline, real 0.17-0.25).
//
// Reference: the same functions as fft_stock.h - the radix passes and their drivers (src/pffft_priv_impl.h:122-901, cfftf1_ps :1004-1048,
// rfftf1_ps / rfftb1_ps :809-901), real finalize / preprocess (:1330-1462), the reorder of pffft_zreorder (:1158-1193) - as ONE sweep.
//
// Shape: one 512-thread workgroup per vector and CU, ONE image in LDS, every exchange IN PLACE:
//   * Stockham autosort in 2-4 stages of radix 3 ... 32 (cxmath.h dftR; the plan is data: StockStage of fft_stock.h, paddings from the
//     same bank model), a stage's butterflies dealt round robin to the threads; a thread reads ALL its operands, the workgroup meets at a
//     barrier, then it writes its results - the image a stage writes may be padded differently from the one it read;
//   * canonical complex input goes from HBM straight into the first stage's operand registers, canonical complex output from the last
//     stage's results straight to HBM (8 / 16-byte accesses, consecutive lanes on consecutive points);
//   * the pffft-internal layout enters / leaves as an image OF THE LAYOUT (linear 16-byte chunks on the HBM side: dense 1 KiB instructions;
//     the 4-scalar groups of block b sit at group ^ (b & 7), fft_stock.h sk_igrp, so that the scalar picks of the butterfly side walk the
//     banks without padding): the first stage of a complex backward transform reads its operands from it, the last stage of a forward one
//     writes its results into it;
//   * real transforms: the pair pass runs between the natural image and the layout image (all pairs of a thread in registers across a
//     barrier) or, where the canonical half-complex spectrum is the input / output, between HBM and the natural image;
//   * backward = conj o forward o conj (one set of stage bodies); base twiddles W_(Ns R)^jm from a compact table in LDS, powers recomputed;
//   * vectors are pulled in order from a work counter, the first one of a workgroup static.
#pragma once
#include "fft_stock.h"

namespace pf {

constexpr int ONE_WG = 512;
template <typename T> constexpr int one_nmax() { return sizeof(T) == 4 ? 18432 : 9216; }
// trips of a stage of radix R: butterflies per thread (compile-time bound, the run-time count is predicated): as many as keep a thread's
// operands within 48 (float) / 24 (double) complex registers - the planner only uses a radix where n / R butterflies fit (one_build)
// (float, radices 24 ... 32 as a MIDDLE stage - image to image: 32-bit LDS addresses, no 64-bit pair per HBM access - take TWO trips: that
//  gives n = 14400 ... 18432 three-stage plans, 24 x 25 x 24 ... 24 x 32 x 24; every other (radix, position) keeps the 48 / 24 budget - a wider
//  one for all of them spilled 50-350 B per lane in the kernels with a pair pass)
// NARROW: the one kernel without them - the float complex backward transform from the internal layout, whose layout-image first stage (two scalar
// picks and their index arithmetic per operand) is the kernel closest to its 256 registers: with the two-trip instantiations next to it its
// three-stage sizes lost 13 % (n = 12000: 0.46 -> 0.40); it keeps a plan of its own (one_tu.hip: Setup::one[2])
__host__ __device__ constexpr int one_trips_of(bool is_double, int R, bool middle = false, bool narrow = false) {
    const int budget = is_double ? 24 : 48;
    if (!is_double && middle && !narrow && R >= 24) return 2;
    return budget / R < 1 ? 1 : budget / R;
}
template <typename T, int R, int SRC, int DST, bool NARROW> constexpr int one_trips() { return one_trips_of(sizeof(T) == 8, R, SRC == 1 && DST == 1, NARROW); }

template <typename T> struct OneLds { size_t tab = 0, twr = 0, next = 0, total = 0; };
template <typename T> __host__ __device__ constexpr OneLds<T> one_lds(const StockPlan& p, bool real) {
    OneLds<T> l{};
    size_t o = (size_t)p.img * sizeof(cx<T>);
    l.tab = o; o += (size_t)(p.ctab + 1) * sizeof(cx<T>);
    l.twr = o; if (real) o += ((size_t)64 + p.n / 128 + 2) * sizeof(cx<T>);
    l.next = o; o += 16;
    l.total = o;
    return l;
}

template <typename T> struct OneCtx {
    cx<T>* img;            // the image
    const cx<T>* tab;      // compact base twiddles
    const cx<T>* gsrc;     // canonical complex input vector (first stage from HBM)
    cx<T>* gdst;           // canonical complex output vector (last stage to HBM)
    int tid;
    bool cj_in, cj_out;
    int n4; unsigned m_n4;   // n / 4 and its magic multiplier (layout image)
};

// scalar index of the real part of complex bin k inside the layout image (the imaginary part: index ^ 4): block b = 32 scalars = eight
// groups of four, group g of block b at g ^ (b & 7)     (internal[32 b + 8 m + 4 p + l] = part p of X[m n/4 + 4 b + l])
__device__ __forceinline__ int one_lpos(int m, int t) { const int b = t >> 2; return 32 * b + 4 * ((2 * m) ^ (b & 7)) + (t & 3); }
// ... of half-complex bin k of a REAL transform: odd quarters run backwards (bin_of, fft_generic.h)
__device__ __forceinline__ int one_lposr(int k, int n4, unsigned m_n4) {
    const int qq = udiv(k, m_n4), r = k - qq * n4;
    const int tt = (qq & 1) ? (r ? n4 - r : 0) : r;
    return one_lpos(qq, tt);
}
// linear 16-byte chunk cc of the layout -> chunk of the image
template <typename T> __device__ __forceinline__ int one_lchunk(int cc) {
    if constexpr (sizeof(T) == 4) { const int b = cc >> 3; return (cc & ~7) | ((cc & 7) ^ (b & 7)); }
    else { const int b = cc >> 4, g = (cc >> 1) & 7; return (cc & ~15) | (((g ^ (b & 7)) << 1) | (cc & 1)); }
}

// operand q of a stage times W^(q jm), W = p1 (the schemes of fft_stock.h sk_stage: powers at most five products deep)
template <typename T, int R> __device__ __forceinline__ void one_twiddle(cx<T> (&v)[R], cx<T> p1) {
    typedef cx<T> CX;
    if constexpr (R > 16) {
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
        p[1] = p1;
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

// One stage, in place.  SRC / DST: 0 = HBM (the canonical complex vector), 1 = the LDS image, 2 = the image of the internal layout (complex
// transforms: SRC of the first stage, DST of the last one).
//   operand q of butterfly j: logical point j + q nb of the image the previous stage wrote (blocks of Ns points padded by rpad);
//   result d goes to logical point (j div Ns) Ns R + (j mod Ns) + d Ns of this stage's image (blocks of Ns R points padded to wblk).
// Barriers: between the reads and the writes of an LDS -> LDS stage; before the writes of an HBM -> LDS stage (the previous vector's last
// readers); after every stage that wrote the image.
template <typename T, int R, int SRC, int DST, bool NARROW>
__device__ __forceinline__ void one_stage(const StockStage& st, const OneCtx<T>& c) {
    typedef cx<T> CX;
    constexpr int K = one_trips<T, R, SRC, DST, NARROW>();
    CX v[K][R];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = c.tid + k * ONE_WG;
        if (j < st.nb) {
            if constexpr (SRC == 1) {
                const int jd = st.Ns > 1 ? udiv(j, st.m_Ns) : j;
                const CX* p = c.img + j + jd * st.rpad;
#pragma unroll
                for (int q = 0; q < R; ++q) v[k][q] = p[q * st.rstride];
            } else if constexpr (SRC == 2) {   // first stage: operand q is bin j + q nb, which sits in quarter (j + q nb) div n/4
                const T* ps = reinterpret_cast<const T*>(c.img);
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    int m, t;
                    if constexpr (R % 4 == 0) { m = q / (R / 4); t = j + (q % (R / 4)) * st.nb; }
                    else { const int P = j + q * st.nb; m = udiv(P, c.m_n4); t = P - m * c.n4; }
                    const int ip = one_lpos(m, t);
                    v[k][q] = mk<T>(ps[ip], ps[ip ^ 4]);
                }
                __builtin_amdgcn_sched_barrier(0);   // (one trip's index arithmetic at a time: interleaved across trips it spills)
            } else {
                // (uniform base + 32-bit element offset: one address register per load instead of a 64-bit pair)
#pragma unroll
                for (int q = 0; q < R; ++q) v[k][q] = __builtin_nontemporal_load(c.gsrc + (unsigned)(j + q * st.nb));
            }
        }
    }
    if constexpr (DST != 0) __syncthreads();
    int tid_w = c.tid;
    asm volatile("" : "+v"(tid_w));          // (the write indices are derived after the barrier: hoisted above it they live in scratch)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = tid_w + k * ONE_WG;
        if (j < st.nb) {
            int jd = j, jm = 0;
            if (st.Ns > 1) { jd = udiv(j, st.m_Ns); jm = j - jd * st.Ns; }
            if (c.cj_in) {
#pragma unroll
                for (int q = 0; q < R; ++q) v[k][q].y = -v[k][q].y;
            }
            if (st.Ns > 1) one_twiddle<T, R>(v[k], c.tab[st.tw_off + jm]);
            dftR<R, FWD>(v[k]);
            if constexpr (DST == 1) {
                CX* p = c.img + jd * st.wblk + jm;
#pragma unroll
                for (int d = 0; d < R; ++d) p[d * st.Ns] = v[k][d];
            } else if constexpr (DST == 2) {   // last stage, forward: bin j + d nb into the layout image
                T* ps = reinterpret_cast<T*>(c.img);
#pragma unroll
                for (int d = 0; d < R; ++d) {
                    int m, t;
                    if constexpr (R % 4 == 0) { m = d / (R / 4); t = j + (d % (R / 4)) * st.nb; }
                    else { const int P = j + d * st.nb; m = udiv(P, c.m_n4); t = P - m * c.n4; }
                    const int ip = one_lpos(m, t);
                    ps[ip] = v[k][d].x;
                    ps[ip ^ 4] = v[k][d].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {   // last stage: Ns = nb, result d is bin j + d nb
                if (c.cj_out) {
#pragma unroll
                    for (int d = 0; d < R; ++d) v[k][d].y = -v[k][d].y;
                }
#pragma unroll
                for (int d = 0; d < R; ++d) __builtin_nontemporal_store(v[k][d], c.gdst + (unsigned)(j + d * st.nb));
            }
        }
    }
    if constexpr (DST != 0) __syncthreads();
}

// (the stages next to the layout image are the first / last one of a plan: radices from 8 - one_build; the small radices there, twelve and
//  more trips of index arithmetic, were the kernels' only spills)
template <typename T, int SRC, int DST, bool NARROW>
__device__ __forceinline__ void one_run(const StockStage& st, const OneCtx<T>& c) {
    switch (st.R) {
        case 3: if constexpr (SRC != 2 && DST != 2) one_stage<T, 3, SRC, DST, NARROW>(st, c); break;
        case 4: if constexpr (SRC != 2 && DST != 2) one_stage<T, 4, SRC, DST, NARROW>(st, c); break;
        case 5: if constexpr (SRC != 2 && DST != 2) one_stage<T, 5, SRC, DST, NARROW>(st, c); break;
        case 6: if constexpr (SRC != 2 && DST != 2) one_stage<T, 6, SRC, DST, NARROW>(st, c); break;
        case 8: one_stage<T, 8, SRC, DST, NARROW>(st, c); break;
        case 9: one_stage<T, 9, SRC, DST, NARROW>(st, c); break;
        case 10: one_stage<T, 10, SRC, DST, NARROW>(st, c); break;
        case 12: one_stage<T, 12, SRC, DST, NARROW>(st, c); break;
        case 15: one_stage<T, 15, SRC, DST, NARROW>(st, c); break;
        case 16: one_stage<T, 16, SRC, DST, NARROW>(st, c); break;
        default:
            if constexpr (sizeof(T) == 4) {
                switch (st.R) {
                    case 20: if constexpr (SRC == 1 && DST == 1) one_stage<T, 20, SRC, DST, NARROW>(st, c); break;   // (20, 30: middle stages only - one_build)
                    case 24: one_stage<T, 24, SRC, DST, NARROW>(st, c); break;
                    case 25: one_stage<T, 25, SRC, DST, NARROW>(st, c); break;
                    case 27: one_stage<T, 27, SRC, DST, NARROW>(st, c); break;
                    case 30: if constexpr (SRC == 1 && DST == 1) one_stage<T, 30, SRC, DST, NARROW>(st, c); break;
                    case 32: one_stage<T, 32, SRC, DST, NARROW>(st, c); break;
                    default: break;
                }
            }
            break;
    }
}

// flags: bit 0 input in the internal layout (backward), bit 1 output in the internal layout (forward), bit 2 backward, bit 3 real
template <typename T, int FLAGS>
__global__ void __launch_bounds__(ONE_WG, 1)
fft_one_kernel(const T* __restrict__ in, T* __restrict__ out, size_t batch, StockPlan p, const cx<T>* __restrict__ twc,
               const cx<T>* __restrict__ twrg, unsigned* ctr) {
    typedef cx<T> CX;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool in_int = FLAGS & 1, out_int = FLAGS & 2, bwd = FLAGS & 4, real = FLAGS & 8;
    const OneLds<T> L = one_lds<T>(p, real);
    CX* const img = reinterpret_cast<CX*>(smem_raw);
    CX* const tab = reinterpret_cast<CX*>(smem_raw + L.tab);
    CX* const lds0 = img;                                            // sk_twr addresses the pair-pass tables from the start of LDS
    const int twr_off = (int)(L.twr / sizeof(CX));
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + L.next);
    const int tid = threadIdx.x, n = p.n, ns = p.ns, half = n >> 1, per = half + 1;
    const size_t vs = real ? (size_t)2 * n : (size_t)2 * n;         // scalars per vector: N = 2n (real) / 2N (complex)
    for (int i = tid; i < p.ctab; i += ONE_WG) tab[i] = twc[i];
    if constexpr (real) sk_twr_fill<T>(lds0, twr_off, 2, twrg, n, tid, ONE_WG);
    constexpr bool first_from_g = !in_int && !(real && bwd);
    constexpr bool last_to_g = !out_int && !(real && !bwd);
    __syncthreads();

    // half-complex bins k, n - k -> the CONJUGATE of the packed spectrum (the stages then run a forward transform and the store conjugates):
    // Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A + B, D = i conj(W_N^k) (A - B), A = X[k], B = conj X[n-k]   (fft_stock.h, real backward)
    auto pair_bwd = [&](int k, CX A, CX Bn, CX& Pk, CX& Pm) {
        if (k == 0) { Pk = mk<T>(A.x + A.y, -(A.x - A.y)); Pm = Pk; }
        else if (k == half) { Pk = mk<T>((T)2 * A.x, (T)2 * A.y); Pm = Pk; }
        else {
            const CX wk = sk_twr<T>(lds0, twr_off, 2, twrg, k);
            const CX S = add_conj(A, Bn), Dm = cmulc(sub_conj(A, Bn), wk);
            Pk = conj(add_rot<BWD>(S, Dm));
            Pm = sub_rot<BWD>(S, Dm);
        }
    };
    // packed spectrum Z -> half-complex X: X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k]
    auto pair_fwd = [&](int k, CX A, CX Bn, CX& Xa, CX& Xb) {
        if (k == 0) { Xa = mk<T>(A.x + A.y, A.x - A.y); Xb = Xa; }        // (DC, Nyquist): include/pffft/pffft.h:144-152
        else if (k == half) { Xa = conj(A); Xb = Xa; }
        else {
            const CX wk = sk_twr<T>(lds0, twr_off, 2, twrg, k);
            const CX S = add_conj(A, Bn) * (T)0.5, Dm = cmul(sub_conj(A, Bn) * (T)0.5, wk);
            Xa = add_rot<FWD>(S, Dm);
            Xb = conj(sub_rot<FWD>(S, Dm));
        }
    };

    typedef vec4<float> chunk16;
    const int nchk = (int)((size_t)n * sizeof(CX) / 16);
    constexpr int NC = (int)(((size_t)one_nmax<T>() * sizeof(CX) / 16 + ONE_WG - 1) / ONE_WG);   // 16-byte chunks per thread
    constexpr int NP = (one_nmax<T>() / 2 + 1 + ONE_WG - 1) / ONE_WG;                              // pair items per thread
    T* const imgs = reinterpret_cast<T*>(img);
    chunk16* const img16 = reinterpret_cast<chunk16*>(img);

    size_t cur = blockIdx.x;
    for (unsigned it = 0; cur < batch; ++it) {
        if (ctr && tid == 0) s_next[it & 1] = atomicAdd(ctr, 1u);      // read by everyone at the end of this iteration, barriers in between
        // (every index of an iteration is derived from an opaque copy of the thread index: the positions of a thread's pairs, chunks and
        //  operands are the same for every vector, and hoisted out of this loop they lived in scratch - 240-770 B per lane)
        int tix = tid;
        asm volatile("" : "+v"(tix));
        const T* gin = in + cur * vs;
        T* gout = out + cur * vs;
        constexpr bool TOUCH = (real && bwd) || (in_int && sizeof(T) == 8);
        [[maybe_unused]] int touched = 0;
        constexpr int NT = (int)(((size_t)one_nmax<T>() * sizeof(CX) / 128 + ONE_WG - 1) / ONE_WG);   // 128-byte lines per thread
        OneCtx<T> c;
        c.img = img; c.tab = tab; c.tid = tix; c.n4 = n >> 2; c.m_n4 = p.m_n4;
        c.gsrc = reinterpret_cast<const CX*>(gin); c.gdst = reinterpret_cast<CX*>(gout);
        c.cj_in = false; c.cj_out = false;

        // ---------------------------------------------------------------- input phases
        constexpr bool RIN_NAT = in_int && real;                       // real backward, internal layout in: straight into the NATURAL image
        constexpr bool ROUT_NAT = out_int && real && sizeof(T) == 4;   // real forward, internal layout out, float: gathered from the natural image
        if constexpr (RIN_NAT) {
            // (real: measured against the layout image + pair pass across a barrier, which holds every pair of a thread in registers: 0.27-0.35 of the
            //  roofline against 0.15-0.19; float complex through the same deposit: 0.39-0.44 flat against 0.46-0.53 through
            //  the layout image for n <= 13824 and 0.29-0.39 above: not adopted) item i = (block b, quarter q) is eight scalars at offset 8 i of the layout - two dense 16 / 32-byte loads per
            //  lane - and four bins of the natural image (bin_of, fft_generic.h); the pair pass then runs in place: bins k and n - k belong to one item
            typedef vec4<T> V4;
            constexpr int NI = (one_nmax<T>() / 4 + ONE_WG - 1) / ONE_WG;
            const int items4 = n >> 2;
            V4 re[NI], im[NI];
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                const unsigned i = (unsigned)tix + (unsigned)r * ONE_WG;
                if ((int)i < items4) {
                    re[r] = __builtin_nontemporal_load(reinterpret_cast<const V4*>(gin + 8 * i));
                    im[r] = __builtin_nontemporal_load(reinterpret_cast<const V4*>(gin + 8 * i + 4));
                }
            }
            __syncthreads();                                           // the previous vector's last readers of the image
            int tid_c = tix;
            asm volatile("" : "+v"(tid_c));
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                const int i = tid_c + r * ONE_WG;
                if (i < items4) {
                    const T sg = real ? (T)1 : (T)-1;                  // (complex backward: conjugated on the way in)
                    img[bin_of(2 * i, 0, n, real)] = mk<T>(re[r].x, sg * im[r].x);
                    img[bin_of(2 * i, 1, n, real)] = mk<T>(re[r].y, sg * im[r].y);
                    img[bin_of(2 * i, 2, n, real)] = mk<T>(re[r].z, sg * im[r].z);
                    img[bin_of(2 * i, 3, n, real)] = mk<T>(re[r].w, sg * im[r].w);
                }
            }
            __syncthreads();
            if constexpr (real) {
#pragma unroll 2
                for (int k = tid_c; k < per; k += ONE_WG) {
                    const CX A = img[k], Bn = (k != 0 && k != half) ? img[n - k] : A;
                    CX Pk, Pm;
                    pair_bwd(k, A, Bn, Pk, Pm);
                    img[k] = Pk;
                    if (k != 0 && k != half) img[n - k] = Pm;
                }
                __syncthreads();
            }
        } else if constexpr (in_int) {
            // the vector arrives in the internal layout: linear 16-byte chunks into the layout image
            const chunk16* g16 = reinterpret_cast<const chunk16*>(gin);
            chunk16 raw[NC];
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                const unsigned cc = (unsigned)tix + (unsigned)r * ONE_WG;
                if ((int)cc < nchk) raw[r] = __builtin_nontemporal_load(g16 + cc);
            }
            __syncthreads();                                           // the previous vector's last readers of the image
            int tid_c = tix;
            asm volatile("" : "+v"(tid_c));
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                const int cc = tid_c + r * ONE_WG;
                if (cc < nchk) img16[one_lchunk<T>(cc)] = raw[r];
            }
            __syncthreads();
        }
        if constexpr (real && bwd && !RIN_NAT) {
            // half-complex spectrum X -> conj of the packed spectrum in the natural image: bins k and n - k belong to one item; all items of a
            // thread are read (HBM: ascending / descending runs) before the barrier, written after it
            const CX* gx = reinterpret_cast<const CX*>(gin);
            CX pa[NP], pb[NP];
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int k = tix + r * ONE_WG;
                if (k < per) {
                    pa[r] = __builtin_nontemporal_load(gx + (unsigned)k);
                    pb[r] = __builtin_nontemporal_load(gx + (unsigned)(k ? n - k : 0));
                }
            }
            __syncthreads();
            int tid_w = tix;
            asm volatile("" : "+v"(tid_w));
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int k = tid_w + r * ONE_WG;
                if (k < per) {
                    CX Pk, Pm;
                    pair_bwd(k, pa[r], pb[r], Pk, Pm);
                    img[k] = Pk;
                    if (k != 0 && k != half) img[n - k] = Pm;
                }
                if (r & 1) __builtin_amdgcn_sched_barrier(0);   // (two pairs at a time: all of them interleaved spill)
            }
            __syncthreads();
        }

        // ---------------------------------------------------------------- stages (one call site per kind of stage)
        constexpr bool NARROW = (FLAGS & 15) == 5 && sizeof(T) == 4;
        constexpr int SRC0 = first_from_g ? 0 : (in_int && !RIN_NAT) ? 2 : 1;   // (real: the pair pass left the natural image)
        constexpr int DSTL = last_to_g ? 0 : (out_int && !real) ? 2 : 1;
#pragma unroll 1
        for (int si = 0; si < ns; ++si) {
            const StockStage st = p.st[si];
            const bool first = si == 0, last = si == ns - 1;
            c.cj_in = first && bwd && !real;          // complex backward: conj o forward o conj (real backward: the pair pass conjugates)
            c.cj_out = last && bwd;
            if constexpr (TOUCH) if (last) asm volatile("" ::"v"(touched));   // (consumed BEFORE the last stage: behind its stores the wait would cover them too)
            if (first && SRC0 != 1) { if constexpr (SRC0 != 1) one_run<T, SRC0, 1, NARROW>(st, c); }
            else if (last && DSTL != 1) { if constexpr (DSTL != 1) one_run<T, 1, DSTL, NARROW>(st, c); }
            else one_run<T, 1, 1, NARROW>(st, c);
            if constexpr (TOUCH) if (first) {
                // One image leaves no room to hold the next vector, and a stage's operands leave no registers: its lines are TOUCHED instead - one
                // 4-byte load per 128-byte line, issued behind the first stage's barriers (everyone knows the next vector by then), consumed at
                // the end of the iteration - so that they travel from HBM to L2 / the Infinity Cache while the stages run and the next
                // iteration's loads find them there.  Measured per flow (same box, alternating): real backward 0.33-0.39 -> 0.38-0.49, double complex
                // backward from the layout 0.37 -> 0.43; every flow whose first stage loads from HBM itself -3 ... -8 % (wherever the value is
                // consumed), the float complex backward one from the layout -14 %: those stay untouched
                const size_t nxv = ctr ? (size_t)gridDim.x + (unsigned)__builtin_amdgcn_readfirstlane((int)s_next[it & 1]) : cur + gridDim.x;
                if (nxv < batch) {
                    const char* pn = reinterpret_cast<const char*>(in + nxv * vs);
                    const unsigned vbytes = (unsigned)(vs * sizeof(T));
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        const unsigned off = ((unsigned)tix + (unsigned)r * ONE_WG) * 128u;
                        if (off < vbytes) touched ^= *reinterpret_cast<const volatile int*>(pn + off);
                    }
                }
            }
        }

        // ---------------------------------------------------------------- output phases
        if constexpr (real && !bwd) {
            // packed spectrum Z (natural image) -> half-complex X: to HBM, or - all pairs of a thread in registers across a barrier - into the layout image
            if constexpr (!out_int) {
                CX* gx = reinterpret_cast<CX*>(gout);
#pragma unroll 2
                for (int k = tix; k < per; k += ONE_WG) {
                    const CX A = img[k], Bn = (k != 0 && k != half) ? img[n - k] : A;
                    CX Xa, Xb;
                    pair_fwd(k, A, Bn, Xa, Xb);
                    __builtin_nontemporal_store(Xa, gx + (unsigned)k);
                    if (k != 0 && k != half) __builtin_nontemporal_store(Xb, gx + (unsigned)(n - k));
                }
            } else if constexpr (ROUT_NAT) {
                // float: pair pass in place, then item i = (b, q) gathers its four bins from the natural image (two 16-byte stores per lane);
                // measured 0.32-0.38 against 0.22-0.30 through the layout image (double: 0.23-0.26 against 0.34-0.37 - its 32-byte halves at a
                // 64-byte stride cost more than the extra exchange)
#pragma unroll 2
                for (int k = tix; k < per; k += ONE_WG) {
                    const CX A = img[k], Bn = (k != 0 && k != half) ? img[n - k] : A;
                    CX Xa, Xb;
                    pair_fwd(k, A, Bn, Xa, Xb);
                    img[k] = Xa;
                    if (k != 0 && k != half) img[n - k] = Xb;
                }
                __syncthreads();
                typedef vec4<T> V4;
#pragma unroll 2
                for (int i = tix; i < (n >> 2); i += ONE_WG) {
                    const CX x0 = img[bin_of(2 * i, 0, n, 1)], x1 = img[bin_of(2 * i, 1, n, 1)];
                    const CX x2 = img[bin_of(2 * i, 2, n, 1)], x3 = img[bin_of(2 * i, 3, n, 1)];
                    V4 re, im;
                    re.x = x0.x; re.y = x1.x; re.z = x2.x; re.w = x3.x;
                    im.x = x0.y; im.y = x1.y; im.z = x2.y; im.w = x3.y;
                    __builtin_nontemporal_store(re, reinterpret_cast<V4*>(gout + 8 * (unsigned)i));
                    __builtin_nontemporal_store(im, reinterpret_cast<V4*>(gout + 8 * (unsigned)i + 4));
                }
            } else {
                CX pa[NP], pb[NP];
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const int k = tix + r * ONE_WG;
                    if (k < per) { pa[r] = img[k]; pb[r] = (k != 0 && k != half) ? img[n - k] : pa[r]; }
                }
                __syncthreads();
                int tid_w = tix;
                asm volatile("" : "+v"(tid_w));
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const int k = tid_w + r * ONE_WG;
                    if (k < per) {
                        CX Xa, Xb;
                        pair_fwd(k, pa[r], pb[r], Xa, Xb);
                        const int ia = one_lposr(k, c.n4, c.m_n4);
                        imgs[ia] = Xa.x; imgs[ia ^ 4] = Xa.y;
                        if (k != 0 && k != half) { const int ib = one_lposr(n - k, c.n4, c.m_n4); imgs[ib] = Xb.x; imgs[ib ^ 4] = Xb.y; }
                    }
                    if (r & 1) __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
        }
        if constexpr (out_int && !ROUT_NAT) {
            chunk16* g16 = reinterpret_cast<chunk16*>(gout);
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                const unsigned cc = (unsigned)tix + (unsigned)r * ONE_WG;
                if ((int)cc < nchk) __builtin_nontemporal_store(img16[one_lchunk<T>((int)cc)], g16 + cc);
            }
        }
        // (read back as a wave-uniform value: the vector's base pointers then live in scalar registers)
        const size_t nx = ctr ? (size_t)gridDim.x + (unsigned)__builtin_amdgcn_readfirstlane((int)s_next[it & 1]) : cur + gridDim.x;
        cur = nx;
    }
    if (ctr && tid == 0) {
        __threadfence();
        const unsigned dn = atomicAdd(&ctr[1], 1u);
        if (dn == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

}  // namespace pf
