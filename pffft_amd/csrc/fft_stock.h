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
//     the first stage reads the vector straight from HBM, the last one writes the canonical spectrum
//     straight back (both 8/16-byte accesses, consecutive lanes on consecutive points);
//   * exchanges ping-pong between two LDS images -> one barrier per stage; the image written by stage s is
//     padded by pad[s] points per block of Ns R points (chosen on the host by simulating the bank
//     conflicts of the write and of the following read);
//   * pffft-internal layout through an image of the layout itself (32-scalar blocks padded to IBS scalars):
//     linear 16-byte chunks on the HBM side, scalar picks on the butterfly side (complex) or in the pair
//     pass (real);
//   * the backward transform is conj o forward o conj (sign flips on the HBM-side accesses), so there is
//     one set of stage bodies per precision;
//   * groups of G consecutive vectors are pulled in order from the work counter (see fft_c1024.h).
#pragma once
#include "cxmath.h"
#include "fft_generic.h"  // fdiv

namespace pf {

constexpr int SK_MAX_STAGES = 6;

struct StockPlan {
    int n, ns;       // complex points per transform, stages
    int G;           // transforms per workgroup pass
    int img;         // complex points per image slot (largest padded image)
    int twmode;      // 0: padded W_n^j table in LDS, 1: base twiddle from the global table, powers recomputed
    int twr_lds;     // W_N^k table of the real pair pass in LDS
    unsigned char radix[SK_MAX_STAGES];
    unsigned char pad[SK_MAX_STAGES];  // pad[s]: points added after every block of Ns_s R_s points of the image stage s writes
};

template <typename T> struct StockLds { size_t buf, tab, twr, next, total; };
template <typename T> __host__ __device__ inline StockLds<T> stock_lds(const StockPlan& p) {
    StockLds<T> l;
    size_t o = 0;
    l.buf = o; o += (size_t)2 * p.G * p.img * sizeof(cx<T>);
    l.tab = o; if (p.twmode == 0) o += ((size_t)p.n + (p.n >> 5) + 1) * sizeof(cx<T>);
    l.twr = o; if (p.twr_lds) o += ((size_t)p.n / 2 + 1) * sizeof(cx<T>);
    l.next = o; o += 16;
    l.total = o;
    return l;
}

enum { SK_G = 0, SK_L = 1, SK_I = 2 };  // operand source / result destination: HBM, LDS image, LDS internal-layout image

template <typename T> struct SkIbs { static constexpr int v = 32 + (sizeof(T) == 4 ? 4 : 2); };

template <typename T> struct SkArgs {
    int n, nb, Ns, total;        // total = vectors in this group * nb butterflies
    float inv_nb, inv_Ns, inv_n4;
    int img;                     // slot stride of the LDS images (complex points)
    int rpad, rstride;           // LDS source: operand q of butterfly j at j + (j div Ns) rpad + q rstride
    int wblk;                    // LDS destination: result d at (j div Ns) wblk + (j mod Ns) + d Ns
    int twstep, twmode;
    T sgn_in, sgn_out;           // -1 conjugates on the HBM-side access (backward transform)
    const cx<T>* lsrc; cx<T>* ldst;
    const cx<T>* gsrc; cx<T>* gdst;   // vector 0 of the group (stride n)
    const cx<T>* tw;
};

__device__ __forceinline__ int tpad(int i) { return i + (i >> 5); }

template <typename T, int R, int SRC, int DST>
__device__ __forceinline__ void sk_stage(const SkArgs<T>& a) {
    typedef cx<T> CX;
    constexpr int IBS = SkIbs<T>::v;
    const int n4 = a.n >> 2;
    for (int i = threadIdx.x; i < a.total; i += blockDim.x) {
        const int g = fdiv(i, a.nb, a.inv_nb), j = i - g * a.nb;
        int jd = j, jm = 0;
        if (a.Ns > 1) { jd = fdiv(j, a.Ns, a.inv_Ns); jm = j - jd * a.Ns; }
        CX v[R];
        // ---- operands
        if constexpr (SRC == SK_G) {
            const CX* p = a.gsrc + (size_t)g * a.n + j;
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = __builtin_nontemporal_load(p + q * a.nb);
#pragma unroll
            for (int q = 0; q < R; ++q) v[q].y *= a.sgn_in;
        } else if constexpr (SRC == SK_L) {
            const CX* p = a.lsrc + g * a.img + j + jd * a.rpad;
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = p[q * a.rstride];
        } else {  // complex spectrum in the internal layout: point P = j + q nb sits in quarter P div n/4
            const T* p = reinterpret_cast<const T*>(a.lsrc + g * a.img);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                int qq, r;
                if constexpr (R % 4 == 0) { qq = q / (R / 4); r = j + (q % (R / 4)) * a.nb; }
                else { const int P = j + q * a.nb; qq = fdiv(P, n4, a.inv_n4); r = P - qq * n4; }
                const int ip = IBS * (r >> 2) + 8 * qq + (r & 3);
                v[q] = mk<T>(p[ip], p[ip + 4] * a.sgn_in);
            }
        }
        // ---- twiddles W_{Ns R}^(q jm) = W_n^(q jm twstep)
        if (a.Ns > 1) {
            const int k = jm * a.twstep;
            if (a.twmode == 0) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = cmul(v[q], a.tw[tpad(q * k)]);
            } else {
                CX p[R < 4 ? 4 : R];
                p[1] = a.tw[k];
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
#pragma unroll
            for (int d = 0; d < R; ++d) {
                CX o = v[d];
                o.y *= a.sgn_out;
                __builtin_nontemporal_store(o, p + d * a.nb);
            }
        } else if constexpr (DST == SK_L) {
            CX* p = a.ldst + g * a.img + jd * a.wblk + jm;
#pragma unroll
            for (int d = 0; d < R; ++d) p[d * a.Ns] = v[d];
        } else {  // last stage, forward, complex: bin j + d nb into the internal-layout image
            T* p = reinterpret_cast<T*>(a.ldst + g * a.img);
#pragma unroll
            for (int d = 0; d < R; ++d) {
                int qq, r;
                if constexpr (R % 4 == 0) { qq = d / (R / 4); r = j + (d % (R / 4)) * a.nb; }
                else { const int P = j + d * a.nb; qq = fdiv(P, n4, a.inv_n4); r = P - qq * n4; }
                const int ip = IBS * (r >> 2) + 8 * qq + (r & 3);
                p[ip] = v[d].x;
                p[ip + 4] = v[d].y;
            }
        }
    }
}

template <typename T, int SRC, int DST>
__device__ __forceinline__ void sk_run(int R, const SkArgs<T>& a) {
    switch (R) {
        case 3: sk_stage<T, 3, SRC, DST>(a); break;
        case 4: sk_stage<T, 4, SRC, DST>(a); break;
        case 5: sk_stage<T, 5, SRC, DST>(a); break;
        case 6: sk_stage<T, 6, SRC, DST>(a); break;
        case 8: sk_stage<T, 8, SRC, DST>(a); break;
        case 10: sk_stage<T, 10, SRC, DST>(a); break;
        case 12: sk_stage<T, 12, SRC, DST>(a); break;
        default:
            if constexpr (sizeof(T) == 4) {  // double stops at radix 12 (register budget)
                if (R == 15) sk_stage<T, 15, SRC, DST>(a);
                else sk_stage<T, 16, SRC, DST>(a);
            }
            break;
    }
}

// scalar index (real part; imaginary part + 4) of half-complex bin k of a REAL transform inside the
// internal-layout image: odd quarters run backwards (bin_of, fft_generic.h)
template <typename T> __device__ __forceinline__ int sk_iposr(int k, int n4, float inv_n4) {
    const int qq = fdiv(k, n4, inv_n4), r = k - qq * n4;
    const int tt = (qq & 1) ? (r ? n4 - r : 0) : r;
    return SkIbs<T>::v * (tt >> 2) + 8 * qq + (tt & 3);
}

// flags: bit0 input in internal layout, bit1 output in internal layout, bit2 backward, bit3 real
template <typename T>
__global__ void __launch_bounds__(1024)
fft_stock_kernel(const T* in, T* out, size_t batch, StockPlan p, int flags, const cx<T>* __restrict__ twg,
                 const cx<T>* __restrict__ twrg, unsigned* ctr) {
    typedef cx<T> CX;
    typedef vec4<float> chunk16;
    constexpr int IBS = SkIbs<T>::v, CH = 16 / (int)sizeof(T), CPB = 32 / CH, BCH = IBS / CH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const StockLds<T> L = stock_lds<T>(p);
    CX* bufs[2];
    bufs[0] = reinterpret_cast<CX*>(smem_raw + L.buf);
    bufs[1] = bufs[0] + (size_t)p.G * p.img;
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + L.next);
    const int n = p.n, G = p.G, ns = p.ns;
    const bool in_int = flags & 1, out_int = flags & 2, bwd = flags & 4, real = flags & 8;
    const CX* tw = twg;
    const CX* twr = twrg;
    if (p.twmode == 0) {
        CX* t = reinterpret_cast<CX*>(smem_raw + L.tab);
        for (int i = threadIdx.x; i < n; i += blockDim.x) t[tpad(i)] = twg[i];
        tw = t;
    }
    if (p.twr_lds && real) {
        CX* t = reinterpret_cast<CX*>(smem_raw + L.twr);
        for (int i = threadIdx.x; i <= n / 2; i += blockDim.x) t[i] = twrg[i];
        twr = t;
    }
    const int n4 = n >> 2, half = n >> 1, per = half + 1;
    const int nchk = (int)((size_t)n * sizeof(CX) / 16);  // 16-byte chunks per vector
    const float inv_n4 = 1.0f / (float)n4, inv_per = 1.0f / (float)per, inv_nchk = 1.0f / (float)nchk;
    const int img16 = (int)((size_t)p.img * sizeof(CX) / 16);  // slot stride in chunks (img is even)

    const bool dyn = ctr != nullptr;
    unsigned pend = 0, g0 = blockIdx.x;
    if (dyn && threadIdx.x == 0) {
        s_next[0] = atomicAdd(&ctr[0], 1u);
        pend = atomicAdd(&ctr[0], 1u);
    }
    __syncthreads();
    if (dyn) g0 = s_next[0];
    int w = 0;  // image the next LDS-writing phase writes; the phase after it reads bufs[w ^ 1] ... (ping-pong)
    for (unsigned it = 0; (size_t)g0 * G < batch; ++it) {
        if (dyn && threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = atomicAdd(&ctr[0], 1u);
        }
        const size_t t0 = (size_t)g0 * G;
        const int g_here = (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G);
        const CX* gin = reinterpret_cast<const CX*>(in) + t0 * n;
        CX* gout = reinterpret_cast<CX*>(out) + t0 * n;
        unsigned gn = g0 + gridDim.x;
        bool have_gn = !dyn;
#define SK_SYNC() do { __syncthreads(); if (!have_gn) { gn = s_next[(it + 1) & 1]; have_gn = true; } } while (0)

        // ---- internal-layout input: linear 16-byte chunks into the padded block image
        if (in_int) {
            const chunk16* s16 = reinterpret_cast<const chunk16*>(gin);
            chunk16* d16 = reinterpret_cast<chunk16*>(bufs[w]);
            for (int c = threadIdx.x; c < g_here * nchk; c += blockDim.x) {
                const int g = fdiv(c, nchk, inv_nchk), cc = c - g * nchk;
                d16[g * img16 + (cc / CPB) * BCH + (cc % CPB)] = __builtin_nontemporal_load(s16 + c);
            }
            w ^= 1;
            SK_SYNC();
        }
        // ---- real backward: half-complex spectrum X -> conj of the packed spectrum Z' (the stages then run
        //      a forward transform; the final conjugation happens on the store):
        //      Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A + B, D = i conj(W_N^k) (A - B), A = X[k], B = conj X[n-k]
        if (real && bwd) {
            const T* si = reinterpret_cast<const T*>(bufs[w ^ 1]);
            for (int id = threadIdx.x; id < g_here * per; id += blockDim.x) {
                const int g = fdiv(id, per, inv_per), k = id - g * per;
                CX A, Bn;
                if (in_int) {
                    const T* ps = si + (size_t)g * p.img * 2;
                    const int ia = sk_iposr<T>(k, n4, inv_n4);
                    A = mk<T>(ps[ia], ps[ia + 4]);
                    if (k != 0 && k != half) {
                        const int ib = sk_iposr<T>(n - k, n4, inv_n4);
                        Bn = mk<T>(ps[ib], ps[ib + 4]);
                    } else Bn = A;
                } else {
                    const CX* ps = gin + (size_t)g * n;
                    A = ps[k];
                    Bn = (k != 0 && k != half) ? ps[n - k] : A;
                }
                CX* pd = bufs[w] + g * p.img;
                if (k == 0) {
                    pd[0] = mk<T>(A.x + A.y, -(A.x - A.y));
                } else if (k == half) {
                    pd[half] = mk<T>((T)2 * A.x, (T)2 * A.y);  // conj(2 conj(A))
                } else {
                    const CX B = conj(Bn);
                    const CX S = A + B, Dm = cmulc(A - B, twr[k]);
                    const CX D = mk<T>(-Dm.y, Dm.x);
                    pd[k] = conj(S + D);
                    pd[n - k] = S - D;  // conj(conj(S - D))
                }
            }
            w ^= 1;
            SK_SYNC();
        }
        // ---- stages
        {
            int Ns = 1;
            for (int s = 0; s < ns; ++s) {
                const int R = p.radix[s];
                SkArgs<T> a;
                a.n = n; a.nb = n / R; a.Ns = Ns; a.total = g_here * a.nb;
                a.inv_nb = 1.0f / (float)a.nb; a.inv_Ns = 1.0f / (float)Ns; a.inv_n4 = inv_n4;
                a.img = p.img;
                a.rpad = s ? p.pad[s - 1] : 0;
                a.rstride = a.nb + (s ? (a.nb / Ns) * p.pad[s - 1] : 0);
                a.wblk = Ns * R + p.pad[s];
                a.twstep = n / (Ns * R); a.twmode = p.twmode;
                a.sgn_in = bwd ? (T)-1 : (T)1; a.sgn_out = a.sgn_in;
                a.lsrc = bufs[w ^ 1]; a.ldst = bufs[w];
                a.gsrc = gin; a.gdst = gout;
                a.tw = tw;
                if (s == 0) {
                    if (real && bwd) sk_run<T, SK_L, SK_L>(R, a);
                    else if (in_int) sk_run<T, SK_I, SK_L>(R, a);
                    else sk_run<T, SK_G, SK_L>(R, a);
                    w ^= 1;
                    SK_SYNC();
                } else if (s < ns - 1) {
                    sk_run<T, SK_L, SK_L>(R, a);
                    w ^= 1;
                    SK_SYNC();
                } else {
                    if (real && !bwd) { a.wblk = n; sk_run<T, SK_L, SK_L>(R, a); w ^= 1; SK_SYNC(); }
                    else if (out_int) { sk_run<T, SK_L, SK_I>(R, a); w ^= 1; SK_SYNC(); }
                    else sk_run<T, SK_L, SK_G>(R, a);
                }
                Ns *= R;
            }
        }
        // ---- real forward: packed spectrum Z (natural image) -> half-complex spectrum X:
        //      X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k]
        if (real && !bwd) {
            T* di = reinterpret_cast<T*>(bufs[w]);
            for (int id = threadIdx.x; id < g_here * per; id += blockDim.x) {
                const int g = fdiv(id, per, inv_per), k = id - g * per;
                const CX* ps = bufs[w ^ 1] + g * p.img;
                CX Xa, Xb;
                if (k == 0) {
                    const CX A = ps[0];
                    Xa = mk<T>(A.x + A.y, A.x - A.y);  // (DC, Nyquist): include/pffft/pffft.h:144-152
                    Xb = Xa;
                } else if (k == half) {
                    Xa = conj(ps[half]);
                    Xb = Xa;
                } else {
                    const CX A = ps[k], B = conj(ps[n - k]);
                    const CX S = (A + B) * (T)0.5, Dm = cmul((A - B) * (T)0.5, twr[k]);
                    const CX D = mk<T>(Dm.y, -Dm.x);
                    Xa = S + D;
                    Xb = conj(S - D);
                }
                if (out_int) {
                    T* pd = di + (size_t)g * p.img * 2;
                    const int ia = sk_iposr<T>(k, n4, inv_n4);
                    pd[ia] = Xa.x; pd[ia + 4] = Xa.y;
                    if (k != 0 && k != half) {
                        const int ib = sk_iposr<T>(n - k, n4, inv_n4);
                        pd[ib] = Xb.x; pd[ib + 4] = Xb.y;
                    }
                } else {
                    CX* pd = gout + (size_t)g * n;
                    __builtin_nontemporal_store(Xa, pd + k);
                    if (k != 0 && k != half) __builtin_nontemporal_store(Xb, pd + (n - k));
                }
            }
            if (out_int) { w ^= 1; SK_SYNC(); }
        }
        // ---- internal-layout output: the padded block image leaves as linear 16-byte chunks
        if (out_int) {
            const chunk16* s16 = reinterpret_cast<const chunk16*>(bufs[w ^ 1]);
            chunk16* d16 = reinterpret_cast<chunk16*>(gout);
            for (int c = threadIdx.x; c < g_here * nchk; c += blockDim.x) {
                const int g = fdiv(c, nchk, inv_nchk), cc = c - g * nchk;
                __builtin_nontemporal_store(s16[g * img16 + (cc / CPB) * BCH + (cc % CPB)], d16 + c);
            }
        }
#undef SK_SYNC
        g0 = gn;
    }
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

}  // namespace pf
