// Register-tiled power-of-two FFT kernels: n = 512 ... 16384 complex points per transform, float or
// double, complex or real (real length N = 2n), forward/backward, canonical or pffft-internal layout.
//
// Replaces for these sizes the same reference functions as fft_generic.h (cfftf1_ps/rfftf1_ps/rfftb1_ps
// pass drivers, passf*/radf*/radb* butterflies, cplx/real finalize+preprocess, zreorder:
// src/pffft_priv_impl.h:122-901, :1158-1462) with ONE pass over HBM per vector.
//
// Shape: TPT threads own one transform, E = n/TPT points each, in registers.  Stockham autosort with the
// twiddles on the INPUTS of every stage but the first:
//   stage s (radix R, Ns = product of earlier radices), butterfly j < n/R:
//     in  q : x[j + q n/R] * W_{Ns R}^(q (j mod Ns))          out d : y[(j div Ns) Ns R + (j mod Ns) + d Ns]
// so the first stage reads the vector straight from HBM with 16-byte coalesced loads, the last stage
// writes the canonical spectrum straight back the same way, and only the exchanges between stages go
// through LDS (one image of n points per transform, reused in place).  Layout / real-transform work is
// done by adapters around that core, also through the LDS image:
//   complex, internal layout : natural-order image <-> linear 16-byte chunks of the pffft layout (bin_of)
//   real forward             : Z = FFT_n(x[2j] + i x[2j+1]); pairs (k, n-k) in place:
//                              X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B),
//                              A = Z[k], B = conj Z[n-k]
//   real backward            : pairs in place: Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A+B,
//                              D = i conj(W_N^k) (A-B), A = X[k], B = conj X[n-k];  x = IFFT_n(Z') (= N x)
// The recipe that makes it HBM-bound is the one measured on fft_c1024.h: persistent 512-thread workgroups
// (one per CU) that pull transforms IN ORDER from an atomic counter, every twiddle a thread ever needs
// resident in its registers (they depend on the thread id only), and the 16-byte loads of the NEXT
// transform issued as soon as the current one has left its input registers.
#pragma once
#include "cxmath.h"
#include "fft_generic.h"  // bin_of

namespace pf {

// TWMODE: 0 = every twiddle in registers, 1 = W_n^j table in LDS, 2 = global table (L2),
//         3 = one base twiddle per butterfly in registers, its powers w^2..w^(R-1) recomputed (<= 4 products deep)
//         4 = the same base twiddles from a compact LDS table (TiledCfg::ctw_off), nothing resident in registers
template <typename T, int LOGN_, int TPT_, int NS_, int R0_, int R1_, int R2_, int R3_, int PAD0_, int PADN_,
          int TWMODE_, int PREFETCH_, int WGT_ = 512, int OCC_ = 2, int R4_ = 1>
struct TiledCfg {
    typedef T real_t;
    static constexpr int LOGN = LOGN_, n = 1 << LOGN_, TPT = TPT_, E = n / TPT_, NS = NS_;
    static constexpr int VEC = 16 / (2 * (int)sizeof(T));  // complex points per 16 bytes: 2 float, 1 double
    static constexpr int CH = 16 / (int)sizeof(T);         // scalars per 16-byte chunk: 4 float, 2 double
    static constexpr int NCH = E * 2 / CH;                  // 16-byte chunks per thread
    static constexpr int PAD0 = PAD0_, PADN = PADN_, TWMODE = TWMODE_, PREFETCH = PREFETCH_, OCC = OCC_;
    __host__ __device__ static constexpr int rad(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : s == 2 ? R2_ : s == 3 ? R3_ : R4_; }
    __host__ __device__ static constexpr int ns(int s) {
        int p = 1;
        for (int i = 0; i < s; ++i) p *= rad(i);
        return p;
    }
    // register twiddles: stage s >= 1 holds (E / R_s) * (R_s - 1) of them
    __host__ __device__ static constexpr int tw_off(int s) {
        int o = 0;
        for (int i = 1; i < s; ++i) o += (E / rad(i)) * (TWMODE_ == 3 ? 1 : rad(i) - 1);
        return o;
    }
    static constexpr int TW_COUNT = TWMODE_ == 4 ? 0 : tw_off(NS_);
    // TWMODE 4: one base twiddle per butterfly from a COMPACT table in LDS - W_{Ns R}^k, k < Ns, of every stage s >= 1, one
    // after the other (n = 8192 as 8 x 8 x 16 x 8: 8 + 64 + 1024 entries = 8.6 KiB against 64 KiB for the whole W_n^j) -, its
    // powers recomputed as in mode 3: no twiddle lives in a register across the persistent loop
    __host__ __device__ static constexpr int ctw_off(int s) {
        int o = 0;
        for (int i = 1; i < s; ++i) o += ns(i);
        return o;
    }
    static constexpr int CTW_COUNT = ctw_off(NS_);
    static constexpr int IMG_NAT = n + PADN_ * (n / 64);
    static constexpr int IMG_TRN = R0_ * (n / R0_ + PAD0_);
    // image of the pffft-internal layout: blocks of 32 scalars padded to IBS scalars (bank-conflict-free
    // scalar scatter on one side, linear 16-byte accesses on the other)
    static constexpr int IBS = 32 + (sizeof(T) == 4 ? 4 : 2);
    static constexpr int IMG_INT = (n / 16) * IBS / 2 + 2;  // in points
    static constexpr int IMG_A = IMG_NAT > IMG_TRN ? IMG_NAT : IMG_TRN;
    static constexpr int IMG = (IMG_A > IMG_INT ? IMG_A : IMG_INT) + 8;
    static constexpr int WG_THREADS = TPT > WGT_ ? TPT : WGT_;
    static constexpr int T_PER_WG = WG_THREADS / TPT;
    static constexpr size_t TABLE_BYTES = TWMODE_ == 1 ? (size_t)n * 2 * sizeof(T)
                                          : TWMODE_ == 4 ? ((size_t)CTW_COUNT * 2 * sizeof(T) + 15) / 16 * 16 : 0;
    static constexpr size_t LDS_BYTES = TABLE_BYTES + (size_t)T_PER_WG * IMG * 2 * sizeof(T) + 16;
};

template <class C, int S> struct StageInfo {
    static constexpr int R = C::rad(S), Ns = C::ns(S), B = C::E / R;
    static constexpr bool PAIR = (C::VEC == 2) && (B % 2 == 0);
    static __device__ __forceinline__ int j(int t, int u) {
        if (PAIR) return 2 * t + (u & 1) + 2 * C::TPT * (u >> 1);
        return t + C::TPT * u;
    }
};

template <class C> __device__ __forceinline__ int phys_nat(int P) { return P + C::PADN * (P >> 6); }
template <class C> __device__ __forceinline__ int phys_trn(int P) {  // image after stage 0: P = j*R0 + d -> row d, column j
    constexpr int R0 = C::rad(0);
    return (P & (R0 - 1)) * (C::n / R0 + C::PAD0) + (P / R0);
}

// LDS accesses of one complex point as ONE typed vector access (struct copies of cx<T> through memory
// left type-punned private allocas behind = scratch traffic in the hot loop)
template <typename V> __device__ __forceinline__ void lds_st(V* p, V v) { *p = v; }
template <typename V> __device__ __forceinline__ V lds_ld(const V* p) { return *p; }

// two adjacent float points in one 16-byte access
__device__ __forceinline__ void lds_st2(cx<float>* p, cx<float> a, cx<float> b) {
    vec4<float> x; x.x = a.x; x.y = a.y; x.z = b.x; x.w = b.y;
    *reinterpret_cast<vec4<float>*>(p) = x;
}
__device__ __forceinline__ vec4<float> lds_ld2(const cx<float>* p) {  // by value: reference out-parameters into
    return *reinterpret_cast<const vec4<float>*>(p);                     // the register array left private allocas behind
}
__device__ __forceinline__ void lds_st2(cx<double>*, cx<double>, cx<double>) {}
__device__ __forceinline__ vec4<float> lds_ld2(const cx<double>*) { return vec4<float>(); }

typedef vec4<float> chunk16;  // a 16-byte register quantum, reinterpreted per precision

// Round 6: ONE 8-byte LDS read per instruction.  Left to the compiler, neighbouring 8-byte reads of an exchange are merged into
// ds_read2_b64 / ds_read2st64_b64, which the LDS serves in 8 cycles per wave-instruction where two ds_read_b64 take 2 + 2
// (MI355X_MICROARCH.md, LDS table: 128 against 256 B per clock and CU; measured on the FIR block kernel of fft_fir32.h: +5 %).
#ifndef PF_NO_SINGLE_LDSRD
template <int OFF> __device__ __forceinline__ cx<float> lds_ld_c(const cx<float>* p) {
    // volatile: the load/store optimizer leaves volatile accesses alone (no ds_read2 pairing), and - unlike an inline-asm ds_read_b64 - the
    // compiler still counts the read in lgkmcnt and knows when its value is there (an asm read's result can be copied before it has landed)
    // (the explicit LDS address space: address-space inference leaves volatile accesses on the flat path)
    typedef __attribute__((address_space(3))) const volatile vec2<float>* LP;
    const vec2<float> r = *(LP)((__attribute__((address_space(3))) const char*)p + OFF);
    return mk<float>(r.x, r.y);
}
#else
template <int OFF> __device__ __forceinline__ cx<float> lds_ld_c(const cx<float>* p) { return *reinterpret_cast<const cx<float>*>(reinterpret_cast<const char*>(p) + OFF); }
#endif
__device__ __forceinline__ void lds_rd_wait() {}
template <int OFF> __device__ __forceinline__ cx<double> lds_ld_c(const cx<double>* p) { return *reinterpret_cast<const cx<double>*>(reinterpret_cast<const char*>(p) + OFF); }
// operands q = Q .. NQ - 1 of one butterfly: v[VOFF + q] = p[q STRIDE] (STRIDE in points)
template <int Q, int NQ, int STRIDE, int VOFF, typename T, int E> struct LdsRdSeq {
    static __device__ __forceinline__ void run(cx<T> (&v)[E], const cx<T>* p) {
        v[VOFF + Q] = lds_ld_c<Q * STRIDE * (int)sizeof(cx<T>)>(p);
        if constexpr (Q + 1 < NQ) LdsRdSeq<Q + 1, NQ, STRIDE, VOFF, T, E>::run(v, p);
    }
};

template <typename T> struct ChunkOps;
template <> struct ChunkOps<float> {
    static __device__ __forceinline__ float get(const chunk16& c, int i) { return c[i]; }
    static __device__ __forceinline__ void set(chunk16& c, int i, float v) { c[i] = v; }
};
template <> struct ChunkOps<double> {
    static __device__ __forceinline__ double get(const chunk16& c, int i) {
        return __builtin_bit_cast(vec2<double>, c)[i];
    }
    static __device__ __forceinline__ void set(chunk16& c, int i, double v) {
        vec2<double> d = __builtin_bit_cast(vec2<double>, c);
        d[i] = v;
        c = __builtin_bit_cast(chunk16, d);
    }
};

template <class C, int DIR, int REAL>
struct Tiled {
    typedef typename C::real_t T;
    typedef cx<T> CX;
    typedef ChunkOps<T> CO;
    typedef StageInfo<C, 0> S0;
    typedef StageInfo<C, C::NS - 1> SL;
    static constexpr int n = C::n, E = C::E, TPT = C::TPT, VEC = C::VEC, CH = C::CH, NCH = C::NCH;
    static constexpr int R0 = S0::R, RL = SL::R;
    static constexpr bool REGTW = C::TWMODE == 0 || C::TWMODE == 3;
    // Real transforms: the stage next to the half-complex spectrum (last stage forward, first stage
    // backward) uses the SYMMETRIC butterfly assignment  thread t -> butterflies { t, n/R - t }  (thread 0:
    // { 0, n/(2R) }).  Bin k = j + d n/R and its mirror n - k = (n/R - j) + (R-1-d) n/R then live in the same
    // thread, so the pair pass  X[k], X[n-k] <-> Z[k], Z[n-k]  runs in registers, with no LDS round trip.
    static constexpr int RS = (DIR == FWD) ? RL : R0;                 // radix of that stage
    static constexpr int SYM_STAGE = REAL ? ((DIR == FWD) ? C::NS - 1 : 0) : -1;
    static_assert(!REAL || E / RS == 2, "real transforms need exactly two butterflies per thread in the spectrum-side stage");
    static constexpr int NPT = REAL ? RS : 1;  // pair-pass twiddles per thread
    template <int S> static __device__ __forceinline__ int jm(int t, int u) {
        if constexpr (S == SYM_STAGE) {
            constexpr int nb = n / C::rad(S);
            return u == 0 ? t : (t == 0 ? nb / 2 : nb - t);
        } else {
            return StageInfo<C, S>::j(t, u);
        }
    }

    struct Tw {
        CX r[C::TW_COUNT > 0 ? C::TW_COUNT : 1];
        CX p[NPT];
    };

    static __device__ __forceinline__ void xsync() {
        if (TPT > 64) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }

    template <int S> static __device__ __forceinline__ void load_tw_stage(Tw& w, int t, const CX* __restrict__ twg) {
        if constexpr (S < C::NS) {
            typedef StageInfo<C, S> SI;
            constexpr int R = SI::R, Ns = SI::Ns, step = n / (Ns * R), off = C::tw_off(S);
#pragma unroll
            for (int u = 0; u < SI::B; ++u) {
                const int k = jm<S>(t, u) & (Ns - 1);
                if constexpr (C::TWMODE == 3) w.r[off + u] = twg[k * step];
                else {
#pragma unroll
                    for (int q = 1; q < R; ++q) w.r[off + u * (R - 1) + (q - 1)] = twg[(q * k) * step];
                }
            }
            load_tw_stage<S + 1>(w, t, twg);
        }
    }
    static __device__ __forceinline__ void load_tw(Tw& w, int t, const CX* __restrict__ twg, const CX* __restrict__ twrg) {
        if constexpr (REGTW) {
            load_tw_stage<1>(w, t, twg);
        }
        if constexpr (REAL) load_pair_tw(w.p, t, twrg);
    }
    // W_N^k of the RS mirror pairs this thread owns (pair_regs): t != 0: k = t + d n/RS;
    // thread 0: butterfly 0 pairs d = 1..RS/2-1 at k = d n/RS, butterfly 1 pairs d = 0..RS/2-1 at
    // k = (2d+1) n/(2 RS).  The table holds k <= n/2; W_N^k = -conj(W_N^(n-k)) beyond.
    static __device__ __forceinline__ void load_pair_tw(CX (&p)[NPT], int t, const CX* __restrict__ twrg) {
#pragma unroll
        for (int d = 0; d < RS; ++d) {
            int k = t + d * (n / RS);
            if (t == 0) k = d < RS / 2 ? d * (n / RS) : (2 * (d - RS / 2) + 1) * (n / (2 * RS));
            p[d] = k <= n / 2 ? twrg[k] : conj(twrg[n - k]) * (T)-1;
        }
    }
    template <int S> static __device__ __forceinline__ CX stage_tw(const Tw& w, int t, int u, int q, const CX* tab) {
        typedef StageInfo<C, S> SI;
        if constexpr (C::TWMODE == 0) return w.r[C::tw_off(S) + u * (SI::R - 1) + (q - 1)];
        else {
            const int k = jm<S>(t, u) & (SI::Ns - 1);
            return tab[(q * k) * (n / (SI::Ns * SI::R))];
        }
    }

    template <int S>
    static __device__ __forceinline__ void butterflies(CX (&v)[E], int t, const Tw& w, const CX* tab) {
        typedef StageInfo<C, S> SI;
        constexpr int R = SI::R;
#pragma unroll
        for (int u = 0; u < SI::B; ++u) {
            CX a[R];
#pragma unroll
            for (int q = 0; q < R; ++q) a[q] = v[u * R + q];
            if constexpr (S > 0 && (C::TWMODE == 3 || C::TWMODE == 4)) {
                CX p[R < 16 ? R : 16];  // p[q] = w^q, every power at most 4 products away from the table value
                if constexpr (C::TWMODE == 3) p[1] = w.r[C::tw_off(S) + u];
                else p[1] = lds_ld(tab + C::ctw_off(S) + (jm<S>(t, u) & (SI::Ns - 1)));
                // opaque per iteration: otherwise the powers are hoisted out of the persistent loop and
                // pinned in registers again (which is TWMODE 0 and spills)
                asm volatile("" : "+v"(p[1].x), "+v"(p[1].y));
                if constexpr (R > 2) { p[2] = cmul(p[1], p[1]); p[3] = cmul(p[2], p[1]); }
                if constexpr (R > 4) { p[4] = cmul(p[2], p[2]); p[5] = cmul(p[4], p[1]); p[6] = cmul(p[3], p[3]); p[7] = cmul(p[4], p[3]); }
                if constexpr (R > 8) {
                    p[8] = cmul(p[4], p[4]); p[9] = cmul(p[8], p[1]); p[10] = cmul(p[5], p[5]); p[11] = cmul(p[8], p[3]);
                    p[12] = cmul(p[6], p[6]); p[13] = cmul(p[8], p[5]); p[14] = cmul(p[7], p[7]); p[15] = cmul(p[8], p[7]);
                }
#pragma unroll
                for (int q = 1; q < (R < 16 ? R : 16); ++q) a[q] = twmul<DIR>(a[q], p[q]);
                if constexpr (R > 16) {   // radix 32: w^16 and w^(16+k) = w^16 w^k, five products deep
                    const CX p16 = cmul(p[8], p[8]);
                    a[16] = twmul<DIR>(a[16], p16);
#pragma unroll
                    for (int q = 1; q < 16; ++q) a[16 + q] = twmul<DIR>(a[16 + q], cmul(p16, p[q]));
                }
            } else if constexpr (S > 0) {
#pragma unroll
                for (int q = 1; q < R; ++q) a[q] = twmul<DIR>(a[q], stage_tw<S>(w, t, u, q, tab));
            }
            dftR<R, DIR>(a);
#pragma unroll
            for (int d = 0; d < R; ++d) v[u * R + d] = a[d];
        }
    }

    // The LDS addresses below are written as (one base per butterfly) + (compile-time offset per
    // operand), so that every access is a ds_read/ds_write with an immediate offset and the persistent
    // loop keeps a handful of address registers instead of one per access.  The split is exact:
    //   natural image : phys(P) = P + PADN*(P>>6); for P = H + a + c with H a multiple of min(Ns R, 64)...
    //                    floor((H+a+c)/64) = floor(H/64) + floor(c/64) for the operand
    //                   strides c used here (multiples of Ns, resp. of n/R >= 64)
    //   transposed    : phys(P) = (P mod R0)*ROW + P div R0
    static __device__ __forceinline__ constexpr int nat_off(int c) { return c + C::PADN * (c >> 6); }
    // stages whose butterflies come in adjacent pairs (j, j+1) move two points per 16-byte LDS access
    template <int S> static constexpr bool pair_stage() {
        return StageInfo<C, S>::PAIR && S != SYM_STAGE && (C::PAD0 % 2 == 0) && (C::PADN % 2 == 0);
    }
    template <int S> static __device__ __forceinline__ void xwrite(const CX (&v)[E], int t, CX* img) {
        typedef StageInfo<C, S> SW;
        constexpr int R = SW::R, Ns = SW::Ns, ROW = n / R0 + C::PAD0;
        if constexpr (pair_stage<S>() && (S == 0 || Ns >= 2)) {
#pragma unroll
            for (int ii = 0; ii < SW::B / 2; ++ii) {
                const int j = jm<S>(t, 2 * ii);  // even; j + 1 is the partner
                if constexpr (S == 0) {
                    CX* p = img + j;
#pragma unroll
                    for (int d = 0; d < R; ++d) lds_st2(p + d * ROW, v[(2 * ii) * R + d], v[(2 * ii + 1) * R + d]);
                } else {
                    const int Ha = (j / Ns) * (Ns * R) + (j & (Ns - 1));
                    CX* p = img + Ha + C::PADN * (Ha >> 6);
#pragma unroll
                    for (int d = 0; d < R; ++d) lds_st2(p + nat_off(d * Ns), v[(2 * ii) * R + d], v[(2 * ii + 1) * R + d]);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < SW::B; ++u) {
            const int j = jm<S>(t, u);
            if constexpr (S == 0) {  // P = j*R0 + d -> row d, column j
                CX* p = img + j;
#pragma unroll
                for (int d = 0; d < R; ++d) lds_st(p + d * ROW, v[u * R + d]);
            } else {
                const int Ha = (j / Ns) * (Ns * R) + (j & (Ns - 1));
                CX* p = img + Ha + C::PADN * (Ha >> 6);
#pragma unroll
                for (int d = 0; d < R; ++d) lds_st(p + nat_off(d * Ns), v[u * R + d]);
            }
        }
    }
    template <int S> static __device__ __forceinline__ void xread(CX (&v)[E], int t, const CX* img) {
        typedef StageInfo<C, S + 1> SR;
        constexpr int R2 = SR::R, ROW = n / R0 + C::PAD0;
        static_assert((n / R2) % 64 == 0 || C::PADN == 0, "operand stride must be a multiple of the padding period");
        static_assert((n / R2) % R0 == 0, "operand stride must be a multiple of R0");
        if constexpr (S > 0 && pair_stage<S + 1>()) {
#pragma unroll
            for (int ii = 0; ii < SR::B / 2; ++ii) {
                const int j = jm<S + 1>(t, 2 * ii);
                const CX* p = img + j + C::PADN * (j >> 6);
#pragma unroll
                for (int q = 0; q < R2; ++q) {
                    const vec4<float> x = lds_ld2(p + nat_off(q * (n / R2)));
                    v[(2 * ii) * R2 + q] = mk<T>((T)x.x, (T)x.y);
                    v[(2 * ii + 1) * R2 + q] = mk<T>((T)x.z, (T)x.w);
                }
            }
            return;
        }
        xread_u<S, 0>(v, t, img);
        if constexpr (sizeof(T) == 4) lds_rd_wait();
    }
    // butterfly u (and the following ones) of the unpaired read: compile-time operand offsets from one base per butterfly
    template <int S, int U> static __device__ __forceinline__ void xread_u(CX (&v)[E], int t, const CX* img) {
        typedef StageInfo<C, S + 1> SR;
        constexpr int R2 = SR::R, ROW = n / R0 + C::PAD0;
        const int j = jm<S + 1>(t, U);
        if constexpr (S == 0) {  // P = j + q n/R2 -> row P mod R0 = j mod R0, column j div R0 + q n/(R2 R0)
            const CX* p = img + (j & (R0 - 1)) * ROW + (j / R0);
            LdsRdSeq<0, R2, n / (R2 * R0), U * R2, T, E>::run(v, p);
        } else {
            const CX* p = img + j + C::PADN * (j >> 6);
            // nat_off(q c) = q nat_off(c) for the strides used here (multiples of 64, or no padding)
            LdsRdSeq<0, R2, nat_off(n / R2), U * R2, T, E>::run(v, p);
        }
        if constexpr (U + 1 < SR::B) xread_u<S, U + 1>(v, t, img);
    }

    // chunk index (16-byte units inside one vector) of raw slot i for the two load patterns
    static __device__ __forceinline__ int plain_chunk(int t, int i) {  // first-stage operand order
        if constexpr (VEC == 2) {  // slot i = (pair index ii, input q): ii * R0 + q
            const int ii = i / R0, q = i % R0;
            return t + TPT * ii + q * (n / (2 * R0));
        } else {
            const int u = i / R0, q = i % R0;
            return t + TPT * u + q * (n / R0);
        }
    }
    static __device__ __forceinline__ void load_raw(chunk16 (&raw)[NCH], const T* src, int t, bool plain) {
        const chunk16* s = reinterpret_cast<const chunk16*>(src);
        if (plain) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) raw[i] = __builtin_nontemporal_load(s + plain_chunk(t, i));
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) raw[i] = __builtin_nontemporal_load(s + t + TPT * i);
        }
    }

    // scalar index, inside the internal-layout image, of the real part of bin k = j + d n/RR (imaginary part: + 4).
    // RR/4 operands per spectrum quarter; odd quarters of a real spectrum run backwards (bin_of, fft_generic.h).
    template <int RR> static __device__ __forceinline__ int ipos(int j, int d) {
        constexpr int per = RR / 4;
        const int qq = d / per;
        const int r = j + (d % per) * (n / RR);
        int tt = r;
        if (REAL && (qq & 1)) tt = (n / 4 - r) & (n / 4 - 1);
        return C::IBS * (tt >> 2) + 8 * qq + (tt & 3);
    }

    // one mirror pair: a holds the bin k, b the bin n-k (spectrum X forward-out / backward-in, packed Z on the other side)
    struct Pair { CX a, b; };
    static __device__ __forceinline__ Pair pair1(CX A, CX Bin, CX wk) {
        // forward:  X[k]  = S + D, S = (A+B)/2, D = -(i/2) W_N^k (A-B);   backward: Z'[k] = S + D, S = A+B, D = i conj(W_N^k) (A-B)
        // with B = conj(Bin); D = rot<DIR>(m): the quarter turn rides on the packed add (cxmath.h add_rot / sub_rot)
        const CX su = add_conj(A, Bin), di = sub_conj(A, Bin);
        CX S, m;
        if (DIR == FWD) { S = su * (T)0.5; m = cmul(di * (T)0.5, wk); }
        else { S = su; m = cmulc(di, wk); }
        Pair r;
        r.a = add_rot<DIR>(S, m);
        r.b = conj(sub_rot<DIR>(S, m));
        return r;
    }
    // in-register pair pass on the symmetric stage's operands: v[u*RS + d] = bin jm(t,u) + d n/RS.
    // Branch-free: every thread evaluates the regular pairing (bin of butterfly 0 with its mirror in
    // butterfly 1) and the pairing of thread 0 (both of its butterflies are self-mirrored), then selects.
    static __device__ __forceinline__ CX sel(bool c, CX a, CX b) { return mk<T>(c ? a.x : b.x, c ? a.y : b.y); }
    static __device__ __forceinline__ void pair_regs(CX (&v)[E], int t, const Tw& w) { pair_regs_p(v, t, w.p); }
    static __device__ __forceinline__ void pair_regs_p(CX (&v)[E], int t, const CX (&wp)[NPT]) {
        CX r0[2 * RS], r1[2 * RS];
#pragma unroll
        for (int d = 0; d < RS; ++d) {
            const Pair r = pair1(v[d], v[RS + (RS - 1 - d)], wp[d]);
            r0[d] = r.a; r0[RS + (RS - 1 - d)] = r.b;
        }
        r1[0] = mk<T>(v[0].x + v[0].y, v[0].x - v[0].y);  // bin 0 <-> (DC, Nyquist), the same map in both directions
#pragma unroll
        for (int d = 1; d < RS / 2; ++d) {
            const Pair r = pair1(v[d], v[RS - d], wp[d]);
            r1[d] = r.a; r1[RS - d] = r.b;
        }
        r1[RS / 2] = DIR == FWD ? conj(v[RS / 2]) : mk<T>((T)2 * v[RS / 2].x, (T)-2 * v[RS / 2].y);  // k = n/2
#pragma unroll
        for (int d = 0; d < RS / 2; ++d) {
            const Pair r = pair1(v[RS + d], v[RS + (RS - 1 - d)], wp[RS / 2 + d]);
            r1[RS + d] = r.a; r1[RS + (RS - 1 - d)] = r.b;
        }
        const bool first = (t == 0);
#pragma unroll
        for (int i = 0; i < 2 * RS; ++i) v[i] = sel(first, r1[i], r0[i]);
    }
};

// compact base-twiddle table of TWMODE 4 (entry ctw_off(S) + k = W_{Ns R}^k = W_n^(k n / (Ns R))), filled by the whole workgroup
template <class C, int S = 1>
__device__ __forceinline__ void fill_ctw(cx<typename C::real_t>* tab, const cx<typename C::real_t>* __restrict__ twg, int tid, int nthreads) {
    if constexpr (S < C::NS) {
        constexpr int Ns = C::ns(S), step = C::n / (Ns * C::rad(S));
        for (int k = tid; k < Ns; k += nthreads) tab[C::ctw_off(S) + k] = twg[k * step];
        fill_ctw<C, S + 1>(tab, twg, tid, nthreads);
    }
}

#ifdef PF_TILED_DEBUG
__device__ long long pf_tdbg[64];
#define PF_TSTAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0 && it == 3) pf_tdbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PF_TSTAMP(i) do { } while (0)
#endif

// flags: bit0 = input in internal layout, bit1 = output in internal layout
template <class C, int DIR, int REAL>
__global__ void __launch_bounds__(C::WG_THREADS, C::WG_THREADS >= 1024 ? (C::OCC > 4 ? C::OCC : 4) : C::OCC)
fft_tiled_kernel(const typename C::real_t* in, typename C::real_t* out, unsigned batch, int flags,
                 const cx<typename C::real_t>* __restrict__ twg, const cx<typename C::real_t>* __restrict__ twrg,
                 unsigned* ctr) {
    typedef typename C::real_t T;
    typedef cx<T> CX;
    typedef Tiled<C, DIR, REAL> K;
    typedef typename K::S0 S0;
    typedef typename K::SL SL;
    typedef ChunkOps<T> CO;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, VEC = C::VEC, CH = C::CH, NCH = C::NCH;
    constexpr int R0 = K::R0, RL = K::RL;
    static_assert(VEC == 1 || (S0::PAIR && SL::PAIR), "float configs need an even butterfly count in the first/last stage");
    static_assert(!REAL || ((n / RL) % 64 == 0 && (n / R0) % 64 == 0) || C::PADN == 0, "pad period vs operand stride");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* tab = reinterpret_cast<CX*>(smem_raw);  // W_n^j table (TWMODE 1), else unused
    CX* img = reinterpret_cast<CX*>(smem_raw + C::TABLE_BYTES) + (size_t)slot * C::IMG;
    T* imgs = reinterpret_cast<T*>(img);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + C::TABLE_BYTES + (size_t)C::T_PER_WG * C::IMG * sizeof(CX));
    const bool in_int = flags & 1, out_int = flags & 2;
    const bool plain_in = REAL ? (DIR == FWD) : !in_int;   // operand order of stage 0 straight from HBM
    const bool plain_out = REAL ? (DIR == BWD) : !out_int;

    typename K::Tw w;
    K::load_tw(w, t, twg, twrg);
    const CX* twt = twg;
    if constexpr (C::TWMODE == 1) {
        for (int i = threadIdx.x; i < n; i += C::WG_THREADS) tab[i] = twg[i];
        twt = tab;
    }
    if constexpr (C::TWMODE == 4) {
        fill_ctw<C>(tab, twg, threadIdx.x, C::WG_THREADS);
        twt = tab;
    }
    // ctr == nullptr: static assignment (the grid covers every group exactly once): small batches are
    // latency-bound and skip the atomics; otherwise groups are pulled in order from the counter
    const bool dyn = ctr != nullptr;
    unsigned pend = 0;
    unsigned g = blockIdx.x;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    const size_t last = (size_t)batch - 1;
    chunk16 raw[NCH];
    {
        size_t t0 = (size_t)g * C::T_PER_WG + slot;
        K::load_raw(raw, in + (t0 < last ? t0 : last) * 2 * (size_t)n, t, plain_in);
    }
    for (unsigned it = 0; (size_t)g * C::T_PER_WG < batch; ++it) {
        if (dyn && threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        const size_t tr = (size_t)g * C::T_PER_WG + slot;
        const bool active = tr < batch;  // inactive slots recompute the last vector and never store
        T* dst = out + (active ? tr : last) * 2 * (size_t)n;
        CX v[E];
        // The irregular layout maps (bin_of) are recomputed per iteration from an opaque copy of the thread
        // index: hoisted out of the persistent loop they would pin one address register per scalar.
        int tl = t;
        asm volatile("" : "+v"(tl));

        PF_TSTAMP(0);
        // ------------------------------------------------------------------ input
        if (plain_in) {
            if constexpr (VEC == 2) {
#pragma unroll
                for (int ii = 0; ii < S0::B / 2; ++ii)
#pragma unroll
                    for (int q = 0; q < R0; ++q) {
                        const chunk16 c = raw[ii * R0 + q];
                        v[(2 * ii) * R0 + q] = mk<T>(c.x, c.y);
                        v[(2 * ii + 1) * R0 + q] = mk<T>(c.z, c.w);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < NCH; ++i) v[i] = mk<T>(CO::get(raw[i], 0), CO::get(raw[i], 1));
            }
        } else {
            if (in_int) {
                // internal layout: linear 16-byte chunks into the padded block image, then every thread picks
                // the scalars of its own stage-0 operands
                chunk16* im16 = reinterpret_cast<chunk16*>(imgs);
                constexpr int CPB = 32 / CH;  // chunks per 32-scalar block
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = t + TPT * i;
                    im16[(c / CPB) * (C::IBS / CH) + (c % CPB)] = raw[i];
                }
                K::xsync();
#pragma unroll
                for (int u = 0; u < S0::B; ++u)
#pragma unroll
                    for (int q = 0; q < R0; ++q) {
                        const int ip = K::template ipos<R0>(K::template jm<0>(t, u), q);
                        v[u * R0 + q] = mk<T>(imgs[ip], imgs[ip + 4]);
                    }
                K::xsync();
                if constexpr (REAL) K::pair_regs(v, t, w);
            } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tl + TPT * i;
                {       // canonical: CH/2 consecutive bins
                    if constexpr (VEC == 2) {
                        lds_st(img + phys_nat<C>(2 * c), mk<T>(raw[i].x, raw[i].y));
                        lds_st(img + phys_nat<C>(2 * c + 1), mk<T>(raw[i].z, raw[i].w));
                    } else {
                        lds_st(img + phys_nat<C>(c), mk<T>(CO::get(raw[i], 0), CO::get(raw[i], 1)));
                    }
                }
            }
            K::xsync();
#pragma unroll
            for (int u = 0; u < S0::B; ++u)
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int j = K::template jm<0>(t, u);
                    v[u * R0 + q] = lds_ld(img + j + C::PADN * (j >> 6) + K::nat_off(q * (n / R0)));
                }
            K::xsync();
            if constexpr (REAL) K::pair_regs(v, t, w);  // half-complex spectrum -> packed spectrum, in registers
            }
        }

        PF_TSTAMP(1);
        // ------------------------------------------------------------------ transform
        K::template butterflies<0>(v, t, w, twt);
        PF_TSTAMP(2);
        if constexpr (C::NS > 1) K::template xwrite<0>(v, t, img);
        __syncthreads();  // publishes s_next; first half of exchange 0
        PF_TSTAMP(3);
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        if constexpr (C::PREFETCH) {  // the loads of the next transform fly while this one is finished
            const size_t tn = (size_t)gn * C::T_PER_WG + slot;
            K::load_raw(raw, in + (tn < last ? tn : last) * 2 * (size_t)n, t, plain_in);
        }
        if constexpr (C::NS > 1) { K::template xread<0>(v, t, img); K::xsync(); PF_TSTAMP(4); K::template butterflies<1>(v, t, w, twt); PF_TSTAMP(5); }
        if constexpr (C::NS > 2) { K::template xwrite<1>(v, t, img); K::xsync(); PF_TSTAMP(6); K::template xread<1>(v, t, img); K::xsync(); PF_TSTAMP(7); K::template butterflies<2>(v, t, w, twt); PF_TSTAMP(8); }
        if constexpr (C::NS > 3) { K::template xwrite<2>(v, t, img); K::xsync(); PF_TSTAMP(9); K::template xread<2>(v, t, img); K::xsync(); PF_TSTAMP(10); K::template butterflies<3>(v, t, w, twt); PF_TSTAMP(11); }
        if constexpr (C::NS > 4) { K::template xwrite<3>(v, t, img); K::xsync(); K::template xread<3>(v, t, img); K::xsync(); K::template butterflies<4>(v, t, w, twt); }

        // ------------------------------------------------------------------ output
        if (plain_out) {
            if (active) {
                chunk16* d16 = reinterpret_cast<chunk16*>(dst);
                if constexpr (VEC == 2) {
#pragma unroll
                    for (int ii = 0; ii < SL::B / 2; ++ii)
#pragma unroll
                        for (int d = 0; d < RL; ++d) {
                            const CX a = v[(2 * ii) * RL + d], b = v[(2 * ii + 1) * RL + d];
                            chunk16 x; x.x = a.x; x.y = a.y; x.z = b.x; x.w = b.y;
                            __builtin_nontemporal_store(x, d16 + t + TPT * ii + d * (n / (2 * RL)));
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < SL::B; ++u)
#pragma unroll
                        for (int d = 0; d < RL; ++d) {
                            chunk16 x;
                            CO::set(x, 0, v[u * RL + d].x); CO::set(x, 1, v[u * RL + d].y);
                            __builtin_nontemporal_store(x, d16 + t + TPT * u + d * (n / RL));
                        }
                }
            }
        } else {
            // (real: pair pass in registers) canonical spectrum -> LDS image -> linear chunks of the output layout
            if constexpr (REAL) K::pair_regs(v, t, w);
            PF_TSTAMP(12);
            if (out_int) {
                // scatter re / im scalars into the padded internal-layout image, read it back linearly
#pragma unroll
                for (int u = 0; u < SL::B; ++u)
#pragma unroll
                    for (int d = 0; d < RL; ++d) {
                        const int ip = K::template ipos<RL>(K::template jm<C::NS - 1>(t, u), d);
                        imgs[ip] = v[u * RL + d].x;
                        imgs[ip + 4] = v[u * RL + d].y;
                    }
                K::xsync();
                PF_TSTAMP(13);
                const chunk16* im16 = reinterpret_cast<const chunk16*>(imgs);
                chunk16* d16o = reinterpret_cast<chunk16*>(dst);
                constexpr int CPB = 32 / CH;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = t + TPT * i;
                    const chunk16 o = im16[(c / CPB) * (C::IBS / CH) + (c % CPB)];
                    if (active) __builtin_nontemporal_store(o, d16o + c);
                }
                PF_TSTAMP(14);
                K::xsync();
            } else {
#pragma unroll
            for (int u = 0; u < SL::B; ++u)
#pragma unroll
                for (int d = 0; d < RL; ++d) {
                    const int j = K::template jm<C::NS - 1>(t, u);
                    lds_st(img + j + C::PADN * (j >> 6) + K::nat_off(d * (n / RL)), v[u * RL + d]);
                }
            K::xsync();
            PF_TSTAMP(13);
            chunk16* d16 = reinterpret_cast<chunk16*>(dst);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tl + TPT * i;
                chunk16 o;
                if (out_int) {
                    const int gi = (c * CH) >> 2, l0 = (c * CH) & 3, part = gi & 1;
#pragma unroll
                    for (int s = 0; s < CH; ++s) CO::set(o, s, imgs[2 * phys_nat<C>(bin_of(gi, l0 + s, n, REAL)) + part]);
                } else {
                    if constexpr (VEC == 2) {
                        const CX a = lds_ld(img + phys_nat<C>(2 * c)), b = lds_ld(img + phys_nat<C>(2 * c + 1));
                        o.x = a.x; o.y = a.y; o.z = b.x; o.w = b.y;
                    } else {
                        const CX a = lds_ld(img + phys_nat<C>(c));
                        CO::set(o, 0, a.x); CO::set(o, 1, a.y);
                    }
                }
                if (active) __builtin_nontemporal_store(o, d16 + c);
            }
            PF_TSTAMP(14);
            K::xsync();
            }
        }
        PF_TSTAMP(15);
        if constexpr (!C::PREFETCH) {
            const size_t tn = (size_t)gn * C::T_PER_WG + slot;
            K::load_raw(raw, in + (tn < last ? tn : last) * 2 * (size_t)n, t, plain_in);
        }
        g = gn;
    }
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// ---- configurations: <T, log2 n, threads/transform, stages, R0..R3, PAD0, PADN, TWMODE, PREFETCH> ----
template <typename T> struct TiledPick;
template <> struct TiledPick<float> {
    // small transforms: several per wavefront (TPT < 64), wave-local exchanges only
    typedef TiledCfg<float, 4, 2, 2, 4, 4, 1, 1, 1, 0, 0, 1> C16;
    typedef TiledCfg<float, 5, 4, 3, 4, 2, 4, 1, 1, 0, 0, 1> C32;
    typedef TiledCfg<float, 6, 4, 2, 8, 8, 1, 1, 1, 0, 0, 1> C64;
    typedef TiledCfg<float, 7, 8, 3, 8, 2, 8, 1, 1, 0, 0, 1> C128;
    typedef TiledCfg<float, 8, 16, 3, 8, 4, 8, 1, 2, 0, 0, 1> C256;
    typedef TiledCfg<float, 9, 32, 3, 8, 8, 8, 1, 4, 4, 3, 1> C512;
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 1> C1024;
    typedef TiledCfg<float, 11, 128, 4, 8, 4, 8, 8, 4, 8, 3, 1> C2048;
    typedef TiledCfg<float, 12, 256, 4, 8, 8, 8, 8, 4, 8, 3, 1> C4096;
    typedef TiledCfg<float, 13, 512, 4, 8, 8, 16, 8, 4, 8, 3, 1> C8192;
    typedef TiledCfg<float, 14, 1024, 4, 8, 16, 16, 8, 4, 8, 3, 0> C16384;
};
// Double-precision alternatives measured against TiledPick<double> (tools/c5_ab.py):
//   A: one base twiddle per butterfly in registers, powers recomputed (no LDS twiddle table)
//   B: LDS table + register prefetch of the next vector          C: both
// n = 1024 (BASELINE configs[4], fraction of 8 TB/s, fwd internal / bwd internal / fwd canonical / bwd canonical):
//   pick 0.70 / 0.72 / 0.71 / 0.72,  A 0.75 / 0.75 / 0.76 / 0.76,  B 0.77 / 0.78 / 0.80 / 0.80,  C 0.82 / 0.80 / 0.79 / 0.79
//   real N = 2048: pick 0.63 / 0.62 / 0.70 / 0.63,  A 0.70 / 0.68 / 0.73 / 0.71,  B 0.60 / 0.73 / 0.74 / 0.76,  C 0.57 / 0.75 / 0.69 / 0.78
// n = 512: complex pick 0.73-0.77, C 0.80-0.82; real N = 1024 backward pick 0.66-0.68, C 0.81-0.83
// n = 256 / 128: the gains are in the real backward transforms (N = 512: 0.58-0.64 -> 0.77-0.81, N = 256: 0.62-0.64 -> 0.78-0.80)
// n = 2048 / 4096 (were on the Stockham kernel at 0.65-0.73): complex C 0.76-0.80; real N = 4096 A/C 0.70-0.76, N = 8192 forward 0.70-0.73
// The planner's table (pffft_hip.hip tiled_pick) picks per size, layout and direction.
struct TiledAltF64 {
    typedef TiledCfg<double, 10, 64, 3, 8, 16, 8, 1, 4, 0, 3, 0> A1024;
    typedef TiledCfg<double, 10, 64, 3, 8, 16, 8, 1, 4, 0, 1, 1> B1024;
    typedef TiledCfg<double, 10, 64, 3, 8, 16, 8, 1, 4, 0, 3, 1> C1024;
    typedef TiledCfg<double, 9, 32, 3, 8, 8, 8, 1, 4, 0, 3, 0> A512;
    typedef TiledCfg<double, 9, 32, 3, 8, 8, 8, 1, 4, 0, 1, 1> B512;
    typedef TiledCfg<double, 9, 32, 3, 8, 8, 8, 1, 4, 0, 3, 1> C512;
    typedef TiledCfg<double, 8, 16, 3, 8, 4, 8, 1, 2, 0, 3, 0, 256> A256;
    typedef TiledCfg<double, 8, 16, 3, 8, 4, 8, 1, 2, 0, 0, 1, 256> B256;
    typedef TiledCfg<double, 8, 16, 3, 8, 4, 8, 1, 2, 0, 3, 1, 256> C256;
    typedef TiledCfg<double, 11, 128, 4, 8, 4, 8, 8, 4, 0, 3, 0, 256> A2048;
    typedef TiledCfg<double, 11, 128, 4, 8, 4, 8, 8, 4, 0, 1, 1, 256> B2048;
    typedef TiledCfg<double, 11, 128, 4, 8, 4, 8, 8, 4, 0, 3, 1, 256> C2048;
    typedef TiledCfg<double, 12, 256, 4, 8, 8, 8, 8, 4, 0, 3, 0, 256> A4096;
    typedef TiledCfg<double, 12, 256, 4, 8, 8, 8, 8, 4, 0, 1, 1, 256> B4096;
    typedef TiledCfg<double, 12, 256, 4, 8, 8, 8, 8, 4, 0, 3, 1, 256> C4096;
    typedef TiledCfg<double, 7, 8, 3, 8, 2, 8, 1, 1, 0, 3, 0, 256> A128;
    typedef TiledCfg<double, 7, 8, 3, 8, 2, 8, 1, 1, 0, 0, 1, 256> B128;
    typedef TiledCfg<double, 7, 8, 3, 8, 2, 8, 1, 1, 0, 3, 1, 256> C128;
};
// float n = 64 real forward (N = 128): base twiddle + recomputed powers measured 0.70-0.71 against 0.62-0.65 for the
// all-in-registers table; for every other float size <= 512 the TiledPick entries won the A/B of round 1
// (recomputed powers 0.55-0.83, no prefetch 0.59-0.76, pick 0.62-0.83)
struct TiledAltF32b {
    typedef TiledCfg<float, 6, 4, 2, 8, 8, 1, 1, 1, 0, 3, 1> A64;
    // n = 8192 in THREE stages, 16 x 32 x 16 with 32 points per thread: one exchange less than 8 x 8 x 16 x 8 (LDS cycles
    // of the exchanges 2304 -> 1536, conflict-free with PAD0 = 2, PADN = 0: tools/tiled_lds_search.py)
    typedef TiledCfg<float, 13, 256, 3, 16, 32, 16, 1, 2, 0, 3, 1, 256, 2> T8192;
    typedef TiledCfg<float, 13, 256, 3, 16, 32, 16, 1, 2, 0, 3, 0, 256, 2> T8192np;
    // every twiddle resident in registers (two workgroups of 256 threads per CU leave 256 VGPRs per lane): no power
    // recomputation in the loop
    typedef TiledCfg<float, 13, 256, 3, 16, 32, 16, 1, 2, 0, 0, 0, 256, 2> T8192np0;
    // the same idea one and two sizes down: 32 points per thread, three stages (with / without prefetch)
    typedef TiledCfg<float, 11, 64, 3, 16, 8, 16, 1, 2, 0, 3, 1> T2048;            // ONE wavefront per transform
    typedef TiledCfg<float, 11, 64, 3, 16, 8, 16, 1, 2, 0, 3, 0> T2048np;
    typedef TiledCfg<float, 12, 128, 3, 16, 16, 16, 1, 2, 0, 3, 1, 256, 2> T4096;
    typedef TiledCfg<float, 12, 128, 3, 16, 16, 16, 1, 2, 0, 3, 0, 256, 2> T4096np;
    // n = 16384 (one 128 KiB image, one workgroup per CU): 32 points per thread, 512 threads -> 256 VGPRs per lane, room for
    // the register prefetch of the next vector that the 1024-thread configuration (128 VGPRs) cannot hold
    typedef TiledCfg<float, 14, 512, 4, 16, 8, 8, 16, 2, 0, 3, 1, 512, 2> T16384;
    typedef TiledCfg<float, 14, 512, 4, 16, 8, 8, 16, 2, 0, 3, 0, 512, 2> T16384np;
    typedef TiledCfg<float, 14, 512, 4, 16, 8, 8, 16, 4, 8, 3, 1, 512, 2> T16384b;   // the paddings of the 1024-thread one
};
// (round 4 built 1024-thread configurations of n = 8192 on this engine - eight points per thread, five stages 4 x 8 x 8 x 8 x 4, TiledCfg's
//  fifth radix - and measured them slower than the adopted ones everywhere: DESIGN.md appendix A.9; the configurations are in the git history)

template <> struct TiledPick<double> {
    typedef TiledCfg<double, 4, 2, 2, 4, 4, 1, 1, 1, 0, 0, 0, 256> C16;
    typedef TiledCfg<double, 5, 4, 3, 4, 2, 4, 1, 1, 0, 0, 0, 256> C32;
    typedef TiledCfg<double, 6, 4, 2, 8, 8, 1, 1, 1, 0, 0, 0, 256> C64;
    typedef TiledCfg<double, 7, 8, 3, 8, 2, 8, 1, 1, 0, 0, 0, 256> C128;
    typedef TiledCfg<double, 8, 16, 3, 8, 4, 8, 1, 2, 0, 0, 0, 256> C256;
    typedef TiledCfg<double, 9, 32, 3, 8, 8, 8, 1, 4, 0, 1, 0> C512;
    typedef TiledCfg<double, 10, 64, 3, 8, 16, 8, 1, 4, 0, 1, 0> C1024;
    typedef TiledCfg<double, 11, 128, 4, 8, 4, 8, 8, 4, 0, 1, 0, 256> C2048;
    typedef TiledCfg<double, 12, 256, 4, 8, 8, 8, 8, 4, 0, 1, 0, 256> C4096;
    typedef TiledCfg<double, 13, 512, 4, 8, 8, 16, 8, 4, 0, 3, 0> C8192;
    typedef TiledCfg<double, 14, 1024, 4, 8, 16, 16, 8, 4, 0, 2, 0> C16384;  // (LDS too small: never dispatched)
};

}  // namespace pf
