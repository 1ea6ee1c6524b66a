// Transforms whose working set exceeds LDS (up to the reference's limit N = 2^26,
// src/pffft_priv_impl.h:1069): n = N1 x N2 "four-step" through an HBM workspace.
//   step A : N2 column transforms of length N1 (stride N2), times W_n^(k1 n2)        x[n1 N2 + n2] -> Y[k1 N2 + n2]
//   step B : N1 row transforms of length N2, stored transposed                        Y[k1 N2 + n2] -> X[k1 + N1 k2]
// This is the analogue of what the reference does for every size — one sweep over main memory per
// radix pass (cfftf1_ps, src/pffft_priv_impl.h:1011-1045) — reduced to two sweeps.  Both steps run the
// LDS-resident mixed-radix passes of fft_generic.h on G adjacent columns / rows per workgroup so that
// global accesses stay G*8 (or G*16) bytes contiguous.  Real transforms and the internal layout are
// composed from elementwise passes (pair pass below, zreorder_kernel): this path is about coverage
// (pffft_new_setup accepts these sizes, tests/test_pffft.c goes to N = 65536), not about the roofline.
#pragma once
#include "fft_generic.h"

namespace pf {

struct StridedPlan {
    int n;          // sub-transform length
    int nstages;
    int G;          // transforms per workgroup pass
    unsigned char radix[MAX_STAGES];
    long long count;        // transforms per vector (N2 for step A, N1 for step B)
    long long estride_in, tstride_in, estride_out, tstride_out;  // complex units
    long long vec;          // complex points per vector (batch stride)
    long long twN;          // step A: multiply output element k of transform c by W_twN^(k c); 0 = no twiddle
};

template <typename T> __device__ __forceinline__ cx<T> unit_root(long long m, long long N, int dir) {
    // exp(-/+ 2 pi i m / N) with the angle reduced in double precision
    double s, c;
    sincospi(2.0 * (double)(m % N) / (double)N, &s, &c);
    return mk<T>((T)c, (T)(dir == FWD ? -s : s));
}

// W_N^k = exp(-i pi k / n), N = 2n: the twiddle of the real pair passes below.  float: the angle is reduced in double, the
// sincos runs in float with a first-order correction for the rounding of the angle (error < 1e-8; the exact-argument double
// sincospi cost the one-sweep block kernel 0.64 instead of ~0.75 of the roofline); double: exact-argument sincospi.  BOTH real pair
// kernels use this one function: pffft_transform_ordered == pffft_zreorder(pffft_transform) holds bit for bit.
template <typename T> __device__ __forceinline__ cx<T> pair_root(long long k, long long n);
template <> __device__ __forceinline__ cx<float> pair_root<float>(long long k, long long n) {
    const double turns = (double)k / (double)(2 * n);          // 0 <= turns <= 1/2
    const float th = (float)turns;
    const float d = 6.28318530717958647692f * (float)(turns - (double)th);
    float sn, cs;
    sincospif(2.0f * th, &sn, &cs);
    return mk<float>(cs - sn * d, -(sn + cs * d));
}
template <> __device__ __forceinline__ cx<double> pair_root<double>(long long k, long long n) {
    double sn, cs;
    sincospi((double)k / (double)n, &sn, &cs);
    return mk<double>(cs, -sn);
}

__device__ __forceinline__ int pos_of_sp(int k, const StridedPlan& p) {
    int pos = 0, m = p.n;
    for (int s = 0; s < p.nstages; ++s) {
        int R = p.radix[s], d;
        switch (R) {
            case 2: d = k & 1; k >>= 1; m >>= 1; break;
            case 4: d = k & 3; k >>= 2; m >>= 2; break;
            case 3: d = k % 3; k /= 3; m /= 3; break;
            default: d = k % 5; k /= 5; m /= 5; break;
        }
        pos += d * m;
    }
    return pos;
}

template <typename T, int DIR>
__global__ void __launch_bounds__(1024)
fft_strided_kernel(const cx<T>* in, cx<T>* out, long long batch, StridedPlan p, const cx<T>* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cx<T>* z = reinterpret_cast<cx<T>*>(smem_raw);
    const int n = p.n, G = p.G;
    const long long groups_per_vec = (p.count + G - 1) / G;
    const long long total_groups = batch * groups_per_vec;
    for (long long grp = blockIdx.x; grp < total_groups; grp += gridDim.x) {
        const long long b = grp / groups_per_vec;
        const long long c0 = (grp - b * groups_per_vec) * G;
        const int g_here = (int)((p.count - c0) < G ? (p.count - c0) : G);
        const cx<T>* src = in + b * p.vec + c0 * p.tstride_in;
        cx<T>* dst = out + b * p.vec + c0 * p.tstride_out;
        // load: the faster-varying global index goes to consecutive threads
        const bool t_fast_in = p.tstride_in < p.estride_in;
        for (int i = threadIdx.x; i < g_here * n; i += blockDim.x) {
            int g, e;
            if (t_fast_in) { e = i / g_here; g = i - e * g_here; } else { g = i / n; e = i - g * n; }
            z[gpad(g * n + e)] = src[(long long)g * p.tstride_in + (long long)e * p.estride_in];
        }
        __syncthreads();
        int Ls = n;
        const int total = g_here * n;
        for (int s = 0; s < p.nstages; ++s) {
            const int R = p.radix[s];
            const int tws = n / Ls;
            switch (R) {
                case 2: stage<T, 2, DIR>(z, total, Ls, tws, tw); break;
                case 3: stage<T, 3, DIR>(z, total, Ls, tws, tw); break;
                case 4: stage<T, 4, DIR>(z, total, Ls, tws, tw); break;
                default: stage<T, 5, DIR>(z, total, Ls, tws, tw); break;
            }
            Ls /= R;
            __syncthreads();
        }
        const bool t_fast_out = p.tstride_out < p.estride_out;
        for (int i = threadIdx.x; i < g_here * n; i += blockDim.x) {
            int g, k;
            if (t_fast_out) { k = i / g_here; g = i - k * g_here; } else { g = i / n; k = i - g * n; }
            cx<T> v = z[gpad(g * n + pos_of_sp(k, p))];
            if (p.twN) v = cmul(v, unit_root<T>((long long)k * (c0 + g), p.twN, DIR));
            dst[(long long)g * p.tstride_out + (long long)k * p.estride_out] = v;
        }
        __syncthreads();
    }
}

// real <-> packed-complex pair pass on canonical vectors of n complex bins in global memory (in place)
template <typename T, int DIR>
__global__ void real_pair_kernel(cx<T>* data, long long batch, long long n) {
    const long long per = n / 2 + 1;
    const long long total = batch * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / per, k = i - b * per;
        cx<T>* z = data + b * n;
        if (k == 0) {
            cx<T> a = z[0];
            z[0] = mk<T>(a.x + a.y, a.x - a.y);
        } else if (2 * k == n) {
            cx<T> a = z[k];
            z[k] = DIR == FWD ? conj(a) : mk<T>((T)2 * a.x, (T)-2 * a.y);
        } else {
            const cx<T> wk = pair_root<T>(k, n);          // W_N^k, N = 2n
            const cx<T> A = z[k], Bc = conj(z[n - k]);
            cx<T> S, D;
            if (DIR == FWD) {
                S = (A + Bc) * (T)0.5;
                const cx<T> m = cmul((A - Bc) * (T)0.5, wk);
                D = mk<T>(m.y, -m.x);
            } else {
                S = A + Bc;
                const cx<T> m = cmulc(A - Bc, wk);
                D = mk<T>(-m.y, m.x);
            }
            z[k] = S + D;
            z[n - k] = conj(S - D);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Layout / pair sweeps of the vectors beyond LDS, ONE sweep each (round 3).  Reference: pffft_zreorder
// (src/pffft_priv_impl.h:1158-1193), real finalize / preprocess (:1330-1372, :1423-1462), which the reference also runs as
// separate sweeps over the vector.  Before, a real transform into / out of the internal layout cost two elementwise sweeps
// around the complex core (in-place pair pass + zreorder_kernel with 4-byte gathers: 205 + 280 us per GiB against 170 for a
// copy); here both are one wave-local kernel whose every global access is a dense run:
//   a wavefront owns a TILE of 64 consecutive positions t of the spectrum quarters (16 blocks of the internal layout,
//   SURVEY.md appendix A); lane i = position t0 + i loads / stores the four canonical bins that share that position - runs of
//   64 consecutive bins (512 bytes float) per quarter, ascending or descending - and the tile's 16 blocks pass through a
//   wave-private LDS image (32-scalar blocks padded to 36) whose other side is linear 16-byte units (1 KiB per instruction).
//   MODE 0  complex  canonical -> internal            MODE 1  complex  internal -> canonical
//   MODE 2  real forward:  packed spectrum Z of the complex core -> half-complex spectrum X in the internal layout
//           X[k] = S + D, X[n-k] = conj(S - D), S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k]
//   MODE 3  real backward: X in the internal layout -> packed spectrum Z' for the complex core
//           Z'[k] = S + D, Z'[n-k] = conj(S - D), S = A+B, D = i conj(W_N^k) (A-B), A = X[k], B = conj X[n-k]
//   MODE 4  real backward, canonical input: half-complex X -> Z' out of place (the copy + in-place pair pass it replaces were two sweeps)
//   Real quarters (bin(q, t) of the appendix): position t holds bins t, n/2 - t, n/2 + t, n - t, i.e. the two mirror pairs
//   (t, n - t) and (n/2 - t, n/2 + t) - a lane computes both.
//   t = 0 holds the self-paired bins 0 = (DC, Nyquist) and n/2 and the pair (n/4, 3n/4).
// ------------------------------------------------------------------------------------------------------------------
constexpr int BLK_WAVES = 4;                                    // wavefronts per workgroup (each owns its tiles)
template <typename T> struct BlkGeom {
    static constexpr int CH = 16 / (int)sizeof(T);              // scalars per 16-byte unit
    static constexpr int IBS = 36;                              // scalars per padded 32-scalar block
    static constexpr int IMG = 16 * IBS;                        // scalars per tile image (16 blocks)
    static constexpr int UNITS = 16 * 32 / CH;                  // 16-byte units per tile: 128 (float) / 256 (double)
    static constexpr size_t LDS_BYTES = (size_t)BLK_WAVES * IMG * sizeof(T);
};

template <typename T, int MODE>
__global__ void __launch_bounds__(BLK_WAVES * 64)
big_block_kernel(const T* __restrict__ in, T* __restrict__ out, long long batch, long long n, int kchunk) {
    typedef cx<T> CX;
    typedef BlkGeom<T> G;
    typedef vec4<float> U16;                                    // a 16-byte register quantum
    __shared__ __attribute__((aligned(16))) T img_all[BLK_WAVES * G::IMG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* img = img_all + wave * G::IMG;
    const long long n4 = n >> 2, half = n >> 1;
    const long long tpv = (n4 + 63) >> 6;                       // tiles per vector
    const long long ntiles = batch * tpv;
    const long long gw = (long long)blockIdx.x * BLK_WAVES + wave, nw = (long long)gridDim.x * BLK_WAVES;
    // scalar position, inside the padded tile image, of (quarter q, part p) of this lane's position
    const int ipos = G::IBS * (lane >> 2) + (lane & 3);
    auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // a wavefront takes CHUNKS of kchunk consecutive tiles (kchunk = 1: tile gw, gw + nw, ...)
    for (long long task = gw; task * kchunk < ntiles; task += nw)
    for (long long tile = task * kchunk; tile < (task + 1) * kchunk && tile < ntiles; ++tile) {
        const long long vec = tile / tpv, t0 = (tile - vec * tpv) << 6, t = t0 + lane;
        const bool act = t < n4;
        const CX* cin = reinterpret_cast<const CX*>(in) + vec * n;
        CX* cout = reinterpret_cast<CX*>(out) + vec * n;
        const int nblk = (int)((n4 - t0) < 64 ? (n4 - t0) >> 2 : 16);            // whole blocks of this tile
        const int nunits = nblk * (32 / G::CH);
        const long long ubase = (vec * 2 * n + 8 * t0) / G::CH;                   // first 16-byte unit of the tile in the internal layout
        CX q[4];                                                                  // the four quarters' values at position t
        if constexpr (MODE == 0 || MODE == 2 || MODE == 5) {
            // ---- canonical side in: dense runs of 64 bins per quarter
            if constexpr (MODE == 5) {
                // the canonical half-complex spectrum of a real transform: the four bins of position t (the mirror image of MODE 4) - a pure
                // permutation, for the real forward transforms whose spectrum is computed canonically (two tile sweeps, tile_real_tu.hip)
                const long long ta = act ? t : 0;
                q[0] = __builtin_nontemporal_load(cin + ta);
                q[1] = __builtin_nontemporal_load(cin + (ta ? half - ta : n4));
                q[2] = __builtin_nontemporal_load(cin + half + ta);
                q[3] = __builtin_nontemporal_load(cin + (ta ? n - ta : n - n4));
            } else if constexpr (MODE == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) q[m] = act ? __builtin_nontemporal_load(cin + m * n4 + t) : mk<T>((T)0, (T)0);
            } else {
                const long long ta = act ? t : 0;
                const long long k1 = ta ? ta : n4;                                // pair (k1, n - k1); t = 0 carries (n/4, 3n/4)
                const CX A1 = __builtin_nontemporal_load(cin + k1), B1 = __builtin_nontemporal_load(cin + (n - k1));
                const long long k2 = ta ? half - ta : 0;                          // pair (k2, n - k2); t = 0: the self-paired bins 0 and n/2
                const CX A2 = __builtin_nontemporal_load(cin + k2), B2 = __builtin_nontemporal_load(cin + (ta ? half + ta : half));
                // every W_N^k through the same exact-argument evaluation as real_pair_kernel: pffft_transform_ordered ==
                // pffft_zreorder(pffft_transform) must hold bit for bit (benchmarks/bench_pffft.c:343-349)
                const CX w1 = pair_root<T>(k1, n), w2 = pair_root<T>(k2, n);
                auto pairf = [](CX A, CX Bn, CX wk, CX& Xa, CX& Xb) {
                    const CX S = add_conj(A, Bn) * (T)0.5, Dm = cmul(sub_conj(A, Bn) * (T)0.5, wk);
                    Xa = add_rot<FWD>(S, Dm);
                    Xb = conj(sub_rot<FWD>(S, Dm));
                };
                CX Xa1, Xb1, Xa2, Xb2;
                pairf(A1, B1, w1, Xa1, Xb1);
                pairf(A2, B2, w2, Xa2, Xb2);
                if (ta) { q[0] = Xa1; q[3] = Xb1; q[1] = Xa2; q[2] = Xb2; }       // bins t, n - t, n/2 - t, n/2 + t
                else { q[0] = mk<T>(A2.x + A2.y, A2.x - A2.y); q[1] = Xa1; q[2] = conj(B2); q[3] = Xb1; }   // 0, n/4, n/2, 3n/4
            }
            if (act) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { img[ipos + 8 * m] = q[m].x; img[ipos + 8 * m + 4] = q[m].y; }
            }
            fence();
            // ---- internal side out: the tile's blocks as linear 16-byte units
            U16* o16 = reinterpret_cast<U16*>(out) + ubase;
#pragma unroll
            for (int h = 0; h < G::UNITS / 64; ++h) {
                const int u = lane + 64 * h;
                if (u < nunits) {
                    const int sc = u * G::CH, b = sc >> 5, r = sc & 31;
                    __builtin_nontemporal_store(*reinterpret_cast<const U16*>(img + G::IBS * b + r), o16 + u);
                }
            }
            fence();
        } else {
            if constexpr (MODE == 4) {
                // ---- canonical half-complex spectrum in: the four bins of position t (dense ascending / descending runs)
                const long long ta = act ? t : 0;
                q[0] = __builtin_nontemporal_load(cin + ta);
                q[1] = __builtin_nontemporal_load(cin + (ta ? half - ta : n4));
                q[2] = __builtin_nontemporal_load(cin + half + ta);
                q[3] = __builtin_nontemporal_load(cin + (ta ? n - ta : n - n4));
            } else {
            // ---- internal side in
            const U16* i16 = reinterpret_cast<const U16*>(in) + ubase;
#pragma unroll
            for (int h = 0; h < G::UNITS / 64; ++h) {
                const int u = lane + 64 * h;
                if (u < nunits) {
                    const int sc = u * G::CH, b = sc >> 5, r = sc & 31;
                    *reinterpret_cast<U16*>(img + G::IBS * b + r) = __builtin_nontemporal_load(i16 + u);
                }
            }
            fence();
            if (act) {
#pragma unroll
                for (int m = 0; m < 4; ++m) q[m] = mk<T>(img[ipos + 8 * m], img[ipos + 8 * m + 4]);
            }
            fence();
            }
            if (act) {
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) __builtin_nontemporal_store(q[m], cout + m * n4 + t);
                } else {
                    auto pairb = [](CX A, CX Bn, CX wk, CX& Za, CX& Zb) {
                        const CX S = add_conj(A, Bn), Dm = cmulc(sub_conj(A, Bn), wk);
                        Za = add_rot<BWD>(S, Dm);
                        Zb = conj(sub_rot<BWD>(S, Dm));
                    };
                    const long long k1 = t ? t : n4;
                    const CX w1 = pair_root<T>(k1, n), w2 = pair_root<T>(half - t, n);
                    if (t) {
                        CX Za1, Zb1, Za2, Zb2;
                        pairb(q[0], q[3], w1, Za1, Zb1);                          // X[t], X[n - t]
                        pairb(q[1], q[2], w2, Za2, Zb2);                          // X[n/2 - t], X[n/2 + t]
                        __builtin_nontemporal_store(Za1, cout + t);
                        __builtin_nontemporal_store(Zb1, cout + (n - t));
                        __builtin_nontemporal_store(Za2, cout + (half - t));
                        __builtin_nontemporal_store(Zb2, cout + (half + t));
                    } else {
                        CX Za, Zb;
                        pairb(q[1], q[3], w1, Za, Zb);                            // X[n/4], X[3n/4]
                        cout[n4] = Za; cout[n - n4] = Zb;
                        cout[0] = mk<T>(q[0].x + q[0].y, q[0].x - q[0].y);
                        cout[half] = mk<T>((T)2 * q[2].x, (T)-2 * q[2].y);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// n = R x N2 with R <= 32 and N2 small enough for the LDS-resident batched kernels (complex N up to 32 x 8192): three
// streaming passes instead of the strided mixed-radix kernel above,
//   1. big_col_kernel   : x[n1 N2 + n2] -> length-R transform over n1 IN REGISTERS (one column per thread, every access
//                         coalesced over n2), times W_n^(k1 n2) (one exact sincos per thread, powers by products <= 5 deep)
//   2. the batched kernel of size N2 on the R * batch rows (fft_tiled.h / fft_stock.h, in place)
//   3. big_transpose_kernel : B[k1 N2 + k2] -> X[k2 R + k1] through an LDS tile (coalesced on both sides)
// ------------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ cx<T> big_unit(double turns);   // exp(-2 pi i turns), 0 <= turns < 1
template <> __device__ __forceinline__ cx<float> big_unit<float>(double turns) {
    turns -= rint(turns);
    const float th = (float)turns;
    const float d = 6.28318530717958647692f * (float)(turns - (double)th);
    float sn, cs;
    sincospif(2.0f * th, &sn, &cs);
    return mk<float>(cs - sn * d, -(sn + cs * d));
}
template <> __device__ __forceinline__ cx<double> big_unit<double>(double turns) {
    double sn, cs;
    sincospi(2.0 * turns, &sn, &cs);
    return mk<double>(cs, -sn);
}

template <typename T, int R, int DIR>
__global__ void __launch_bounds__(256)
big_col_kernel(const cx<T>* in, cx<T>* out, long long total, int N2, double inv_n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / N2;
    const int n2 = (int)(i - b * N2);
    const cx<T>* src = in + b * (long long)R * N2 + n2;
    cx<T>* dst = out + b * (long long)R * N2 + n2;
    cx<T> a[R];
#pragma unroll
    for (int q = 0; q < R; ++q) a[q] = src[(long long)q * N2];
    dftR<R, DIR>(a);
    cx<T> p[R];                      // p[k] = W_n^(k n2) (forward sign); twmul conjugates it for the backward transform
    p[1] = big_unit<T>((double)n2 * inv_n);
#pragma unroll
    for (int k = 2; k < R; ++k) p[k] = cmul(p[k >> 1], p[k - (k >> 1)]);
    dst[0] = a[0];
#pragma unroll
    for (int k = 1; k < R; ++k) dst[(long long)k * N2] = twmul<DIR>(a[k], p[k]);
}

// The column pass reading the pffft-internal layout (backward complex transforms with R a multiple of 4, round 3; reference:
// cplx_preprocess on pffft_transform's backward input, src/pffft_priv_impl.h:1239-1270).  Row q of the R x N2 matrix lies in
// quarter m = q div (R/4), at positions t = (q mod R/4) N2 + n2: a workgroup takes 256 adjacent columns; for every q' < R/4
// the four rows q' + m R/4 are the 64 whole blocks of positions q' N2 + n2 - one contiguous range, loaded as linear 16-byte
// units and regrouped through an LDS tile [row][column]; the R-point transforms and the twiddle then run as in
// big_col_kernel.  It replaces a separate reorder sweep.
template <typename T, int R, int DIR>
__global__ void __launch_bounds__(256)
big_col_int_kernel(const T* in, cx<T>* out, long long batch, int N2, double inv_n) {
    static_assert(R % 4 == 0, "quarters must be whole rows");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                   // [R rows][256 + 1 columns] complex, as scalars
    typedef vec4<float> U16;
    constexpr int CH = 16 / (int)sizeof(T), UPB = 32 / CH, UPQ = UPB / 4, PITCH = 257;
    const int tiles_per_vec = (N2 + 255) / 256;
    const long long b = blockIdx.x / tiles_per_vec;
    const int n20 = (int)(blockIdx.x - b * tiles_per_vec) * 256;
    const int w = (N2 - n20) < 256 ? (N2 - n20) : 256;          // columns of this tile (a multiple of 16)
    const int t = threadIdx.x;
    const long long vbase = b * 2 * (long long)R * N2;           // scalars
    const int units = (w / 4) * UPB;
#pragma unroll 1
    for (int qq = 0; qq < R / 4; ++qq) {
        const U16* src = reinterpret_cast<const U16*>(in) + (vbase + 8 * ((long long)qq * N2 + n20)) / CH;
        for (int u = t; u < units; u += 256) {
            const U16 v = __builtin_nontemporal_load(src + u);
            const int blk = u / UPB, r = u - blk * UPB, m = r / UPQ, sub = r - m * UPQ;
            T* row = tile + (size_t)(qq + m * (R / 4)) * PITCH * 2;
            if constexpr (sizeof(T) == 4) {                     // sub = part p of positions 4 blk .. 4 blk + 3
                row[(4 * blk + 0) * 2 + sub] = v.x; row[(4 * blk + 1) * 2 + sub] = v.y;
                row[(4 * blk + 2) * 2 + sub] = v.z; row[(4 * blk + 3) * 2 + sub] = v.w;
            } else {                                            // sub = 2 p + h: part p of positions 4 blk + 2 h, + 1
                const vec2<double> d2 = __builtin_bit_cast(vec2<double>, v);
                const int h = sub & 1, p = sub >> 1;
                row[(4 * blk + 2 * h + 0) * 2 + p] = d2.x; row[(4 * blk + 2 * h + 1) * 2 + p] = d2.y;
            }
        }
    }
    __syncthreads();
    if (t >= w) return;
    const int n2 = n20 + t;
    cx<T>* dst = out + b * (long long)R * N2 + n2;
    cx<T> a[R];
#pragma unroll
    for (int q = 0; q < R; ++q) a[q] = mk<T>(tile[((size_t)q * PITCH + t) * 2], tile[((size_t)q * PITCH + t) * 2 + 1]);
    dftR<R, DIR>(a);
    cx<T> p[R];
    p[1] = big_unit<T>((double)n2 * inv_n);
#pragma unroll
    for (int k = 2; k < R; ++k) p[k] = cmul(p[k >> 1], p[k - (k >> 1)]);
    dst[0] = a[0];
#pragma unroll
    for (int k = 1; k < R; ++k) dst[(long long)k * N2] = twmul<DIR>(a[k], p[k]);
}

// tile of 256 consecutive k2 per workgroup: rows k1 = 0..R-1 in (coalesced over k2), R*256 consecutive outputs out
template <typename T, int R>
__global__ void __launch_bounds__(256)
big_transpose_kernel(const cx<T>* in, cx<T>* out, long long batch, int N2) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cx<T>* tile = reinterpret_cast<cx<T>*>(smem_raw);          // [256][R + 1]
    const int tiles_per_vec = (N2 + 255) / 256;
    const long long b = blockIdx.x / tiles_per_vec;
    const int k20 = (int)(blockIdx.x - b * tiles_per_vec) * 256;
    const int w = (N2 - k20) < 256 ? (N2 - k20) : 256;          // columns in this tile
    const cx<T>* src = in + b * (long long)R * N2 + k20;
    cx<T>* dst = out + b * (long long)R * N2 + (long long)k20 * R;
    const int t = threadIdx.x;
    if (t < w) {
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) tile[t * (R + 1) + k1] = src[(long long)k1 * N2 + t];
    }
    __syncthreads();
    const int tot = w * R;
    for (int j = t; j < tot; j += 256) {
        const int c = j / R, k1 = j - c * R;
        dst[j] = tile[c * (R + 1) + k1];
    }
}

// The same transpose storing the pffft-internal layout (forward complex transforms, round 3; reference: cplx_finalize +
// the layout pffft_transform leaves, src/pffft_priv_impl.h:1195-1237).  Output bin k = k2 R + k1 lies in
// quarter m = k2 div (N2/4): a workgroup takes 64 columns k2 of EACH quarter (four runs of 64 bins per row k1) and owns the
// 16 R whole blocks of the layout that they fill - one contiguous range, stored as linear 16-byte units.  It replaces the
// canonical transpose + a separate reorder sweep.
template <typename T, int R>
__global__ void __launch_bounds__(256)
big_transpose_int_kernel(const cx<T>* in, T* out, long long batch, int N2) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cx<T>* tile = reinterpret_cast<cx<T>*>(smem_raw);          // [4 quarters][64 columns][R + 1]
    typedef vec4<float> U16;
    constexpr int CH = 16 / (int)sizeof(T), UPB = 32 / CH, UPQ = UPB / 4;   // scalars per unit; units per block / per quarter slot
    const int Q = N2 >> 2;                                      // columns per quarter (a multiple of 4)
    const int tiles_per_vec = (Q + 63) / 64;
    const long long b = blockIdx.x / tiles_per_vec;
    const int c0 = (int)(blockIdx.x - b * tiles_per_vec) * 64;
    const int w = (Q - c0) < 64 ? (Q - c0) : 64;                // columns of this tile per quarter
    const cx<T>* src = in + b * (long long)R * N2;
    const int t = threadIdx.x, m0 = t >> 6, c = t & 63;
    if (c < w) {
        cx<T>* row = tile + (m0 * 64 + c) * (R + 1);
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) row[k1] = src[(long long)k1 * N2 + m0 * Q + c0 + c];
    }
    __syncthreads();
    const int units = (w * R / 4) * UPB;
    U16* dst = reinterpret_cast<U16*>(out) + (b * 2 * (long long)R * N2 + 8LL * c0 * R) / CH;
    for (int u = t; u < units; u += 256) {
        const int blk = u / UPB, r = u - blk * UPB, m = r / UPQ, sub = r - m * UPQ;
        const cx<T>* q = tile + m * 64 * (R + 1);
        const int j = 4 * blk;                                  // first of the block's four positions (k2' R + k1, relative to c0 R)
        U16 o;
        if constexpr (sizeof(T) == 4) {                         // sub = part p
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int cc = (j + i) / R, k1 = (j + i) - cc * R; const cx<T> v = q[cc * (R + 1) + k1]; e[i] = sub ? v.y : v.x; }
            o.x = e[0]; o.y = e[1]; o.z = e[2]; o.w = e[3];
        } else {                                                // sub = 2 p + h: part p of positions 2 h, 2 h + 1
            const int h = sub & 1, p = sub >> 1;
            double e[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int jj = j + 2 * h + i, cc = jj / R, k1 = jj - cc * R; const cx<T> v = q[cc * (R + 1) + k1]; e[i] = p ? v.y : v.x; }
            vec2<double> d2; d2.x = e[0]; d2.y = e[1];
            o = __builtin_bit_cast(U16, d2);
        }
        __builtin_nontemporal_store(o, dst + u);
    }
}

}  // namespace pf
