// Shared pieces of the LDS / streaming kernels that survive from the first coverage kernel (an in-place radix 2-5 transform in
// one LDS image; every size it served now runs a Stockham plan or the streaming passes, and the kernel itself was removed in
// round 3): the radix plan struct the setup keeps, bin_of() - the internal-layout map of SURVEY.md appendix A -, the in-place
// DIF stage used by the strided fallback of fft_big.h, and the direct (no LDS image) zreorder / zconvolve kernels that serve
// vectors beyond LDS and double precision (pffft_zreorder src/pffft_priv_impl.h:1158-1193, pffft_zconvolve_* :1534-1684).
#pragma once
#include "cxmath.h"

namespace pf {

constexpr int MAX_STAGES = 28;

struct GenericPlan {
    int n;        // complex length held in LDS: N (complex) or N/2 (real)
    int nstages;
    int is_real;
    int G;        // transforms per workgroup pass
    unsigned char radix[MAX_STAGES];
};

// canonical bin k (natural order) -> LDS position after the DIF passes (mixed-radix digit reversal)
__device__ __forceinline__ int pos_of(int k, const GenericPlan& p) {
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

// Spectrum bin stored at slot l (0..3) of 4-scalar group v of one vector in the pffft-internal
// layout (SIMD_SZ == 4).  Closed forms checked against the reference's own pffft_zreorder
// (src/pffft_priv_impl.h:1158-1193) in tests/test_oracle.py and tests/test_gpu_parity.py:
//   complex: internal[32 b + 8 m + 4 p + l] = part p of X[m n/4 + 4 b + l]
//   real   : internal[32 b + 8 q + 4 p + l] = part p of X[bin(q, t = 4 b + l)] with
//            bin(0,t)=t, bin(2,t)=n/2+t, bin(1,t)= t ? n/2-t : n/4, bin(3,t)= t ? n-t : 3n/4
//            (n = N/2 bins; bin 0 carries (DC, Nyquist))
__device__ __forceinline__ int bin_of(int v, int l, int n, int is_real) {
    int b = v >> 3, q = (v >> 1) & 3, t = 4 * b + l;
    if (!is_real) return q * (n >> 2) + t;
    switch (q) {
        case 0: return t;
        case 2: return (n >> 1) + t;
        case 1: return t ? (n >> 1) - t : (n >> 2);
        default: return t ? n - t : 3 * (n >> 2);
    }
}

// Physical LDS index of logical point e: one pad point per 32 breaks the power-of-two (and multiple-of-32)
// strides of the in-place passes — measured 69 % of the LDS cycles of the unpadded image were bank conflicts.
__device__ __forceinline__ int gpad(int e) { return e + (e >> 5); }

// x / d for 0 <= x < 2^23 through one float multiply and a +-1 correction (integer division by a run-time
// divisor costs ~40 instructions on gfx950 and sat in front of every butterfly)
__device__ __forceinline__ int fdiv(int x, int d, float inv) {
    int q = (int)((float)x * inv);
    const int r = x - q * d;
    if (r >= d) ++q;
    else if (r < 0) --q;
    return q;
}

template <typename T, int R, int DIR>
__device__ __forceinline__ void stage(cx<T>* z, int total, int Ls, int tw_stride, const cx<T>* __restrict__ tw) {
    const int m = Ls / R;
    const int nb = total / R;
    const float inv_m = 1.0f / (float)m;
    for (int id = threadIdx.x; id < nb; id += blockDim.x) {
        int sub = fdiv(id, m, inv_m);
        int i = id - sub * m;
        const int e0 = sub * Ls + i;
        cx<T> a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = z[gpad(e0 + q * m)];
        dftR<R, DIR>(a);
        z[gpad(e0)] = a[0];
#pragma unroll
        for (int d = 1; d < R; ++d) z[gpad(e0 + d * m)] = twmul<DIR>(a[d], tw[(i * d) * tw_stride]);
    }
}

// ------------------------------------------------------------------------------------------------
// Spectral helpers in the internal layout (batched, elementwise, HBM-bound).
//   zreorder      : src/pffft_priv_impl.h:1158-1193  (pure permutation internal <-> canonical)
//   zconvolve_*   : src/pffft_priv_impl.h:1534-1684  (ab (+)= a*b*scaling; for real transforms the
//                   scalars at internal index 0 (DC) and 4 (Nyquist) multiply as reals, :1626-1629)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void zreorder_kernel(const T* __restrict__ in, T* __restrict__ out, size_t batch, int n, int is_real,
                                int to_canonical) {
    const int nv = n >> 1;
    const size_t total = batch * (size_t)nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i / nv;
        int v = (int)(i - t * nv);
        const size_t base = t * 2 * (size_t)n;
        int part = v & 1;
        if (to_canonical) {  // gather: read one internal group, scatter 4 scalars (write side strided)
            vec4<T> val = reinterpret_cast<const vec4<T>*>(in)[i];
            out[base + 2 * (size_t)bin_of(v, 0, n, is_real) + part] = val.x;
            out[base + 2 * (size_t)bin_of(v, 1, n, is_real) + part] = val.y;
            out[base + 2 * (size_t)bin_of(v, 2, n, is_real) + part] = val.z;
            out[base + 2 * (size_t)bin_of(v, 3, n, is_real) + part] = val.w;
        } else {
            vec4<T> val;
            val.x = in[base + 2 * (size_t)bin_of(v, 0, n, is_real) + part];
            val.y = in[base + 2 * (size_t)bin_of(v, 1, n, is_real) + part];
            val.z = in[base + 2 * (size_t)bin_of(v, 2, n, is_real) + part];
            val.w = in[base + 2 * (size_t)bin_of(v, 3, n, is_real) + part];
            reinterpret_cast<vec4<T>*>(out)[i] = val;
        }
    }
}

// One thread handles a (re-group, im-group) pair = 4 complex products.
template <typename T, int ACCUMULATE>
__global__ void zconvolve_kernel(const T* a, const T* b, T* ab, size_t batch, int n, int is_real, T scaling,
                                 size_t a_stride, size_t b_stride) {
    const int npairs = n >> 2;  // pairs of 4-scalar groups per vector
    const size_t total = batch * (size_t)npairs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i / npairs;
        int pr = (int)(i - t * npairs);
        const vec4<T>* a4 = reinterpret_cast<const vec4<T>*>(a + t * a_stride) + 2 * pr;
        const vec4<T>* b4 = reinterpret_cast<const vec4<T>*>(b + t * b_stride) + 2 * pr;
        vec4<T>* ab4 = reinterpret_cast<vec4<T>*>(ab + t * 2 * (size_t)n) + 2 * pr;
        vec4<T> ar = a4[0], ai = a4[1], br = b4[0], bi = b4[1];
        vec4<T> pr_ = ar * br - ai * bi, pi_ = ar * bi + ai * br;
        if (is_real && pr == 0) {  // DC and Nyquist are both real: multiply separately
            pr_.x = ar.x * br.x;
            pi_.x = ai.x * bi.x;
        }
        if (ACCUMULATE) {
            vec4<T> cr = ab4[0], ci = ab4[1];
            ab4[0] = cr + pr_ * scaling;
            ab4[1] = ci + pi_ * scaling;
        } else {
            ab4[0] = pr_ * scaling;
            ab4[1] = pi_ * scaling;
        }
    }
}

}  // namespace pf
