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
            const cx<T> wk = unit_root<T>(k, 2 * n, FWD);  // W_N^k, N = 2n
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

}  // namespace pf
