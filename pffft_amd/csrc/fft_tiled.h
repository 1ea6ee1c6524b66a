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
// through LDS (one image of n points, reused in place; TPT <= 64: wave-local fences, TPT > 64: one
// workgroup per transform and __syncthreads).  Layout / real-transform work is done by adapters around
// that core, also through the LDS image:
//   complex, internal layout : split re/im planes <-> linear 4-scalar groups (bin_of, fft_generic.h)
//   real forward             : Z = FFT_n(x[2j] + i x[2j+1]);  X[k] = (Z[k]+conj Z[n-k])/2 - i/2 W_N^k (Z[k]-conj Z[n-k])
//   real backward            : Z'[k] = (X[k]+conj X[n-k]) + i conj(W_N^k) (X[k]-conj X[n-k]);  x = IFFT_n(Z')  (= N x)
// Workgroups are persistent and pull transforms in order from an atomic counter (see fft_c1024.h for why).
#pragma once
#include "cxmath.h"
#include "fft_generic.h"  // bin_of

namespace pf {

template <typename T, int LOGN_, int TPT_, int NS_, int R0_, int R1_, int R2_, int R3_, int PAD0_ = 0, int PADN_ = 0>
struct TiledCfg {
    typedef T real_t;
    static constexpr int LOGN = LOGN_, n = 1 << LOGN_, TPT = TPT_, E = n / TPT_, NS = NS_;
    static constexpr int VEC = 16 / (2 * (int)sizeof(T));  // complex points per 16 bytes: 2 float, 1 double
    static constexpr int PAD0 = PAD0_;  // row padding (points) of the transposed image after stage 0
    static constexpr int PADN = PADN_;  // padding (points) per 64 points of the natural images
    __host__ __device__ static constexpr int rad(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : s == 2 ? R2_ : R3_; }
    __host__ __device__ static constexpr int ns(int s) {  // product of the radices before stage s
        int p = 1;
        for (int i = 0; i < s; ++i) p *= rad(i);
        return p;
    }
    // LDS image of one transform, in points (complex<T>): natural layout padded per 64, or the transposed
    // image after stage 0 (R0 rows of n/R0 + PAD0), whichever is larger
    static constexpr int IMG_NAT = n + PADN * (n / 64);
    static constexpr int IMG_TRN = R0_ * (n / R0_ + PAD0);
    static constexpr int IMG = (IMG_NAT > IMG_TRN ? IMG_NAT : IMG_TRN) + 8;
    // transforms per workgroup: one when a transform spans several waves, else as many as fit 512
    // threads and ~80 KiB of LDS (two workgroups per CU)
    __host__ __device__ static constexpr int t_per_wg() {
        if (TPT > 64) return 1;
        int m = 512 / TPT;
        while (m > 1 && (size_t)m * IMG * 2 * sizeof(T) > 80 * 1024) m /= 2;
        return m;
    }
    static constexpr int T_PER_WG = t_per_wg();
    static constexpr int WG_THREADS = TPT > 64 ? TPT : (T_PER_WG * TPT < 64 ? 64 : T_PER_WG * TPT);
    static constexpr size_t LDS_BYTES = (size_t)T_PER_WG * IMG * 2 * sizeof(T) + 16;
};

// 16-byte global access = VEC complex points
template <typename T> struct unit16;
template <> struct unit16<float> {
    typedef vec4<float> type;
    static __device__ __forceinline__ void unpack(type v, cx<float>& a, cx<float>& b) { a = mk<float>(v.x, v.y); b = mk<float>(v.z, v.w); }
    static __device__ __forceinline__ type pack(cx<float> a, cx<float> b) { type v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y; return v; }
};
template <> struct unit16<double> {
    typedef vec2<double> type;
};

template <class C, int S> struct StageInfo {
    static constexpr int R = C::rad(S), Ns = C::ns(S), B = C::E / R;
    static constexpr bool PAIR = (C::VEC == 2) && (B % 2 == 0);
    // butterfly index of thread t, slot u
    static __device__ __forceinline__ int j(int t, int u) {
        if (PAIR) return 2 * t + (u & 1) + 2 * C::TPT * (u >> 1);
        return t + C::TPT * u;
    }
};

// physical LDS index (points) of logical position P
template <class C> __device__ __forceinline__ int phys_nat(int P) { return P + C::PADN * (P >> 6); }
template <class C> __device__ __forceinline__ int phys_trn(int P) {  // image after stage 0: P = j*R0 + d -> row d, column j
    constexpr int R0 = C::rad(0);
    return (P & (R0 - 1)) * (C::n / R0 + C::PAD0) + (P / R0);
}

template <class C, int DIR>
struct TiledCore {
    typedef typename C::real_t T;
    typedef cx<T> CX;

    template <int TPT> static __device__ __forceinline__ void xsync() {
        if (TPT > 64) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }

    // butterflies of stage S on v (inputs v[u*R+q] -> outputs v[u*R+d]); twiddles on inputs for S > 0
    template <int S>
    static __device__ __forceinline__ void butterflies(CX (&v)[C::E], int t, const CX* __restrict__ tw) {
        typedef StageInfo<C, S> SI;
        constexpr int R = SI::R, Ns = SI::Ns, B = SI::B;
#pragma unroll
        for (int u = 0; u < B; ++u) {
            CX a[R];
#pragma unroll
            for (int q = 0; q < R; ++q) a[q] = v[u * R + q];
            if (S > 0) {
                const int k = SI::j(t, u) & (Ns - 1);
                constexpr int step = C::n / (Ns * R);
#pragma unroll
                for (int q = 1; q < R; ++q) a[q] = twmul<DIR>(a[q], tw[(q * k) * step]);
            }
            dftR<R, DIR>(a);
#pragma unroll
            for (int d = 0; d < R; ++d) v[u * R + d] = a[d];
        }
    }

    // exchange between stage S and S+1 through the LDS image `img`
    template <int S>
    static __device__ __forceinline__ void exchange(CX (&v)[C::E], int t, CX* img) {
        typedef StageInfo<C, S> SW;
        typedef StageInfo<C, S + 1> SR;
        constexpr int R = SW::R, Ns = SW::Ns;
        constexpr bool TRN = (S == 0);
        // write: y[(j div Ns) Ns R + (j mod Ns) + d Ns]
#pragma unroll
        for (int u = 0; u < SW::B; ++u) {
            const int j = SW::j(t, u);
            const int base = (j / Ns) * (Ns * R) + (j & (Ns - 1));
#pragma unroll
            for (int d = 0; d < R; ++d) {
                const int P = base + d * Ns;
                img[TRN ? phys_trn<C>(P) : phys_nat<C>(P)] = v[u * R + d];
            }
        }
        xsync<C::TPT>();
        // read: x[j' + q n/R']
        constexpr int R2 = SR::R;
#pragma unroll
        for (int u = 0; u < SR::B; ++u) {
            const int j = SR::j(t, u);
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const int P = j + q * (C::n / R2);
                v[u * R2 + q] = img[TRN ? phys_trn<C>(P) : phys_nat<C>(P)];
            }
        }
        xsync<C::TPT>();
    }

    // the whole transform on registers: in  v[u*R0+q] = x[j0(t,u) + q n/R0],  out v[u*RL+d] = X[jL(t,u) + d n/RL]
    static __device__ __forceinline__ void run(CX (&v)[C::E], int t, CX* img, const CX* __restrict__ tw) {
        butterflies<0>(v, t, tw);
        if constexpr (C::NS > 1) { exchange<0>(v, t, img); butterflies<1>(v, t, tw); }
        if constexpr (C::NS > 2) { exchange<1>(v, t, img); butterflies<2>(v, t, tw); }
        if constexpr (C::NS > 3) { exchange<2>(v, t, img); butterflies<3>(v, t, tw); }
    }
};

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
// flags: bit0 = input in internal layout, bit1 = output in internal layout
template <class C, int DIR, int REAL>
__global__ void __launch_bounds__(C::WG_THREADS)
fft_tiled_kernel(const typename C::real_t* in, typename C::real_t* out, unsigned batch, int flags,
                 const cx<typename C::real_t>* __restrict__ tw, const cx<typename C::real_t>* __restrict__ twr,
                 unsigned* ctr) {
    typedef typename C::real_t T;
    typedef cx<T> CX;
    typedef TiledCore<C, DIR> Core;
    typedef StageInfo<C, 0> S0;
    typedef StageInfo<C, C::NS - 1> SL;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, VEC = C::VEC;
    constexpr int R0 = S0::R, RL = SL::R;
    static_assert(VEC == 1 || (S0::PAIR && SL::PAIR), "float configs need an even number of butterflies in the first and last stage");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* img = reinterpret_cast<CX*>(smem_raw) + (size_t)slot * C::IMG;
    T* imgs = reinterpret_cast<T*>(img);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + (size_t)C::T_PER_WG * C::IMG * sizeof(CX));
    const bool in_int = flags & 1, out_int = flags & 2;
    constexpr int GROUPS = (2 * n / 4) / TPT;  // 4-scalar groups per thread (E/2)
    typedef vec4<T> G4;

    unsigned pend = 0;
    if (threadIdx.x == 0) {
        s_next[0] = atomicAdd(&ctr[0], 1u);
        pend = atomicAdd(&ctr[0], 1u);
    }
    __syncthreads();
    unsigned g = s_next[0];
    for (unsigned it = 0; (size_t)g * C::T_PER_WG < batch; ++it) {
        if (threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = atomicAdd(&ctr[0], 1u);
        }
        const size_t tr = (size_t)g * C::T_PER_WG + slot;
        const bool active = tr < batch;
        const size_t trc = active ? tr : (size_t)batch - 1;  // inactive slots recompute the last vector, never store
        const T* src = in + trc * 2 * (size_t)n;
        T* dst = out + trc * 2 * (size_t)n;
        CX v[E];

        // ------------------------------------------------------------------ input adapters
        const bool plain_in = REAL ? (DIR == FWD) : !in_int;  // time-domain input or canonical complex spectrum
        if (plain_in) {
            if constexpr (VEC == 2) {
                const vec4<float>* s4 = reinterpret_cast<const vec4<float>*>(src);
#pragma unroll
                for (int i = 0; i < S0::B / 2; ++i)
#pragma unroll
                    for (int q = 0; q < R0; ++q) {
                        vec4<float> x = __builtin_nontemporal_load(s4 + t + TPT * i + q * (n / (2 * R0)));
                        v[(2 * i) * R0 + q] = mk<T>(x.x, x.y);
                        v[(2 * i + 1) * R0 + q] = mk<T>(x.z, x.w);
                    }
            } else {
                const vec2<double>* s2 = reinterpret_cast<const vec2<double>*>(src);
#pragma unroll
                for (int u = 0; u < S0::B; ++u)
#pragma unroll
                    for (int q = 0; q < R0; ++q) {
                        vec2<double> x = __builtin_nontemporal_load(s2 + t + TPT * u + q * (n / R0));
                        v[u * R0 + q] = mk<T>(x.x, x.y);
                    }
            }
        } else {
            // spectrum input that needs the LDS image first: linear 4-scalar groups -> natural-order bins
            const G4* s4 = reinterpret_cast<const G4*>(src);
            G4 gv[GROUPS];
#pragma unroll
            for (int i = 0; i < GROUPS; ++i) gv[i] = __builtin_nontemporal_load(s4 + t + TPT * i);
#pragma unroll
            for (int i = 0; i < GROUPS; ++i) {
                const int gi = t + TPT * i;
                if (in_int) {
                    const int part = gi & 1;
                    imgs[2 * phys_nat<C>(bin_of(gi, 0, n, REAL)) + part] = gv[i].x;
                    imgs[2 * phys_nat<C>(bin_of(gi, 1, n, REAL)) + part] = gv[i].y;
                    imgs[2 * phys_nat<C>(bin_of(gi, 2, n, REAL)) + part] = gv[i].z;
                    imgs[2 * phys_nat<C>(bin_of(gi, 3, n, REAL)) + part] = gv[i].w;
                } else {
                    img[phys_nat<C>(2 * gi)] = mk<T>(gv[i].x, gv[i].y);
                    img[phys_nat<C>(2 * gi + 1)] = mk<T>(gv[i].z, gv[i].w);
                }
            }
            Core::template xsync<TPT>();
#pragma unroll
            for (int u = 0; u < S0::B; ++u)
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    const int k = S0::j(t, u) + q * (n / R0);
                    if (!REAL) {
                        v[u * R0 + q] = img[phys_nat<C>(k)];
                    } else {  // real backward: Z'[k] = (A+B) + i conj(W_N^k) (A-B), A = X[k], B = conj X[n-k]
                        CX A = img[phys_nat<C>(k)];
                        if (k == 0) {
                            v[u * R0 + q] = mk<T>(A.x + A.y, A.x - A.y);
                        } else {
                            CX Bc = conj(img[phys_nat<C>(n - k)]);
                            CX S = A + Bc, Dm = A - Bc;
                            // conj(W_N^k): k <= n/2 -> conj(twr[k]);  k > n/2 -> -twr[n-k]
                            CX m = (k <= n / 2) ? cmulc(Dm, twr[k]) : cmul(Dm, twr[n - k]) * (T)-1;
                            v[u * R0 + q] = mk<T>(S.x - m.y, S.y + m.x);  // S + i*m
                        }
                    }
                }
            Core::template xsync<TPT>();
        }

        // ------------------------------------------------------------------ the transform
        Core::run(v, t, img, tw);

        __syncthreads();  // publishes s_next (and is a workgroup barrier for TPT > 64 images)
        const unsigned gn = s_next[(it + 1) & 1];

        // ------------------------------------------------------------------ output adapters
        const bool plain_out = REAL ? (DIR == BWD) : !out_int;
        if (plain_out) {
            if (active) {
                if constexpr (VEC == 2) {
                    vec4<float>* d4 = reinterpret_cast<vec4<float>*>(dst);
#pragma unroll
                    for (int i = 0; i < SL::B / 2; ++i)
#pragma unroll
                        for (int d = 0; d < RL; ++d) {
                            CX a = v[(2 * i) * RL + d], b = v[(2 * i + 1) * RL + d];
                            vec4<float> x; x.x = a.x; x.y = a.y; x.z = b.x; x.w = b.y;
                            __builtin_nontemporal_store(x, d4 + t + TPT * i + d * (n / (2 * RL)));
                        }
                } else {
                    vec2<double>* d2 = reinterpret_cast<vec2<double>*>(dst);
#pragma unroll
                    for (int u = 0; u < SL::B; ++u)
#pragma unroll
                        for (int d = 0; d < RL; ++d) {
                            vec2<double> x; x.x = v[u * RL + d].x; x.y = v[u * RL + d].y;
                            __builtin_nontemporal_store(x, d2 + t + TPT * u + d * (n / RL));
                        }
                }
            }
        } else {
            // canonical spectrum -> LDS image (natural order), then 4-scalar groups in output order
#pragma unroll
            for (int u = 0; u < SL::B; ++u)
#pragma unroll
                for (int d = 0; d < RL; ++d) img[phys_nat<C>(SL::j(t, u) + d * (n / RL))] = v[u * RL + d];
            Core::template xsync<TPT>();
            G4* d4 = reinterpret_cast<G4*>(dst);
#pragma unroll
            for (int i = 0; i < GROUPS; ++i) {
                const int gi = t + TPT * i;
                G4 o;
                if (!REAL) {  // complex, internal layout: part p of 4 consecutive bins
                    const int part = gi & 1;
                    o.x = imgs[2 * phys_nat<C>(bin_of(gi, 0, n, 0)) + part];
                    o.y = imgs[2 * phys_nat<C>(bin_of(gi, 1, n, 0)) + part];
                    o.z = imgs[2 * phys_nat<C>(bin_of(gi, 2, n, 0)) + part];
                    o.w = imgs[2 * phys_nat<C>(bin_of(gi, 3, n, 0)) + part];
                } else {      // real forward: X[k] = S + D, S = (A+B)/2, D = -(i/2) W_N^k (A-B)
                    T r[4];
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int k = out_int ? bin_of(gi, l, n, 1) : (2 * gi + (l >> 1));
                        const int part = out_int ? (gi & 1) : (l & 1);
                        CX A = img[phys_nat<C>(k)];
                        CX X;
                        if (k == 0) {
                            X = mk<T>(A.x + A.y, A.x - A.y);
                        } else {
                            CX Bc = conj(img[phys_nat<C>(n - k)]);
                            CX S = (A + Bc) * (T)0.5, Dm = (A - Bc) * (T)0.5;
                            // W_N^k: k <= n/2 -> twr[k];  k > n/2 -> -conj(twr[n-k])
                            CX m = (k <= n / 2) ? cmul(Dm, twr[k]) : cmulc(Dm, twr[n - k]) * (T)-1;
                            X = mk<T>(S.x + m.y, S.y - m.x);  // S - i*m
                        }
                        r[l] = part ? X.y : X.x;
                    }
                    o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
                }
                if (active) __builtin_nontemporal_store(o, d4 + t + TPT * i);
            }
            Core::template xsync<TPT>();
        }
        g = gn;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// ---- configurations: <T, log2 n, threads/transform, stages, radices...> ----
template <typename T> using Tiled512 = TiledCfg<T, 9, 32, 3, 8, 8, 8, 1>;      // E = 16, two transforms per wave
template <typename T> using Tiled1024 = TiledCfg<T, 10, 64, 3, 8, 16, 8, 1>;   // E = 16
template <typename T> using Tiled2048 = TiledCfg<T, 11, 128, 4, 8, 4, 8, 8>;   // E = 16
template <typename T> using Tiled4096 = TiledCfg<T, 12, 128, 3, 16, 16, 16, 1>;   // E = 32
template <typename T> using Tiled8192 = TiledCfg<T, 13, 512, 4, 8, 8, 16, 8>;     // E = 16
template <typename T> using Tiled16384 = TiledCfg<T, 14, 512, 4, 16, 8, 8, 16>;  // E = 32

}  // namespace pf
