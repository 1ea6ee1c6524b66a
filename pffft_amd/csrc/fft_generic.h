// Generic LDS-resident batched FFT kernel: any n = 2^a 3^b 5^c that fits in LDS, float or double,
// real or complex, forward or backward, canonical ("ordered") or pffft-internal ("unordered")
// spectrum layout.  This is the coverage kernel behind every pffft_* / pffftd_* entry; the
// size-specialised kernels (fft_c1024.h ...) replace it on the headline shapes.
//
// What it replaces in the reference (all in src/pffft_priv_impl.h):
//   cfftf1_ps :1004-1048 + passf2/3/4/5_ps :122-321   -> stage<R>() in-place DIF passes in LDS
//   rfftf1_ps/rfftb1_ps :809-901 + radf*/radb* :323-807 -> complex FFT of length N/2 + real_post/real_pre
//   pffft_cplx_finalize/_preprocess :1195-1270, pffft_real_finalize/_preprocess :1330-1462
//                                                       -> absorbed: one length-n transform, no 4-lane stitch
//   pffft_zreorder :1158-1193                           -> bin_of(): the layout is applied on the global
//                                                          load / store address, never as a separate sweep
//   pffft_transform_internal :1465-1532                 -> the kernel body (load, passes, store)
//
// Structure (one workgroup handles G transforms per pass, grid-stride over the batch):
//   L  global -> LDS, coalesced 4-scalar loads, scatter to natural order (undoing the internal layout)
//   P  (real backward) half-complex spectrum -> packed complex spectrum, pairs (k, n-k) in place
//   C  radix-2/3/4/5 decimation-in-frequency passes, in place, output left digit-reversed in LDS
//   Q  (real forward) packed complex spectrum -> half-complex spectrum, pairs in place
//   S  LDS -> global, coalesced 4-scalar stores, gather through digit reversal + layout map
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

// LDS carve-up (bytes) of the generic kernel: data image, then (when they fit) the W_n^j table, the W_N^k
// table of the real pair pass and the digit-reversal table — read thousands of times per transform, and a
// global (L2) load in a barrier-separated pass is what such a kernel waits for.
struct GenericLds {
    size_t tw, twr, pos, next, total;
};
template <typename T>
__host__ __device__ inline GenericLds generic_lds(int n, int G, int is_real, int tables) {
    GenericLds l;
    size_t o = (((size_t)G * n + ((size_t)G * n >> 5) + 2) * sizeof(cx<T>) + 15) / 16 * 16;
    l.tw = o;  if (tables) o += (size_t)n * sizeof(cx<T>);
    l.twr = o; if (tables && is_real) o += ((size_t)n / 2 + 1) * sizeof(cx<T>);
    l.pos = o; if (tables) o += (((size_t)n * 2 + 15) / 16) * 16;
    l.next = o; o += 16;
    l.total = o;
    return l;
}

template <typename T, int DIR>
__global__ void __launch_bounds__(1024)
fft_generic_kernel(const T* in, T* out, size_t batch, GenericPlan p, int in_internal, int out_internal,
                   const cx<T>* __restrict__ twg, const cx<T>* __restrict__ twrg, int tables, unsigned* ctr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cx<T>* z = reinterpret_cast<cx<T>*>(smem_raw);
    T* zs = reinterpret_cast<T*>(smem_raw);
    const int n = p.n, G = p.G;
    const GenericLds L = generic_lds<T>(n, G, p.is_real, tables);
    const cx<T>* tw = twg;
    const cx<T>* twr = twrg;
    const unsigned short* lpos = nullptr;
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + L.next);
    if (tables) {
        cx<T>* ltw = reinterpret_cast<cx<T>*>(smem_raw + L.tw);
        for (int i = threadIdx.x; i < n; i += blockDim.x) ltw[i] = twg[i];
        tw = ltw;
        if (p.is_real) {
            cx<T>* ltwr = reinterpret_cast<cx<T>*>(smem_raw + L.twr);
            for (int i = threadIdx.x; i <= n / 2; i += blockDim.x) ltwr[i] = twrg[i];
            twr = ltwr;
        }
        unsigned short* lp = reinterpret_cast<unsigned short*>(smem_raw + L.pos);
        for (int i = threadIdx.x; i < n; i += blockDim.x) lp[i] = (unsigned short)pos_of(i, p);
        lpos = lp;
    }
    auto POS = [&](int k) -> int { return lpos ? (int)lpos[k] : pos_of(k, p); };
    const int nv = n >> 1;  // 4-scalar groups per vector (2n scalars complex, N = 2n scalars real)
    const float inv_nv = 1.0f / (float)nv, inv_per = 1.0f / (float)(nv + 1);
    const vec4<T>* in4 = reinterpret_cast<const vec4<T>*>(in);
    vec4<T>* out4 = reinterpret_cast<vec4<T>*>(out);

    // groups of G consecutive vectors are pulled in order from the counter (ctr == nullptr: one group per
    // workgroup, static) — see fft_c1024.h for what in-order sweeping buys on HBM
    const bool dyn = ctr != nullptr;
    unsigned pend = 0, g0 = blockIdx.x;
    if (dyn && threadIdx.x == 0) {
        s_next[0] = atomicAdd(&ctr[0], 1u);
        pend = atomicAdd(&ctr[0], 1u);
    }
    __syncthreads();
    if (dyn) g0 = s_next[0];
    for (unsigned it = 0; (size_t)g0 * G < batch; ++it) {
        if (dyn && threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = atomicAdd(&ctr[0], 1u);
        }
        const size_t t0 = (size_t)g0 * G;
        const int g_here = (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G);
        const int totv = g_here * nv;
        // ---- L: load (4 independent 16-byte loads in flight per thread before the first LDS write) ----
        for (int base = threadIdx.x; base < totv; base += 4 * blockDim.x) {
            vec4<T> vals[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iv = base + u * blockDim.x;
                if (iv < totv) vals[u] = __builtin_nontemporal_load(in4 + t0 * nv + iv);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iv = base + u * blockDim.x;
                if (iv >= totv) continue;
                const vec4<T> val = vals[u];
                if (!in_internal) {
                    z[gpad(2 * iv)] = mk<T>(val.x, val.y);
                    z[gpad(2 * iv + 1)] = mk<T>(val.z, val.w);
                } else {
                    int g = fdiv(iv, nv, inv_nv), v = iv - g * nv, part = v & 1;
                    const int eb = g * n;
                    zs[2 * gpad(eb + bin_of(v, 0, n, p.is_real)) + part] = val.x;
                    zs[2 * gpad(eb + bin_of(v, 1, n, p.is_real)) + part] = val.y;
                    zs[2 * gpad(eb + bin_of(v, 2, n, p.is_real)) + part] = val.z;
                    zs[2 * gpad(eb + bin_of(v, 3, n, p.is_real)) + part] = val.w;
                }
            }
        }
        __syncthreads();
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g0 + gridDim.x;
        // ---- P: real backward pre-processing: Z'[k] = (A+B) + i w (A-B), Z'[n-k] = conj((A+B) - i w (A-B)),
        //         A = X[k], B = conj X[n-k], w = exp(+2 pi i k / N)  (gives N*x after the unscaled inverse) ----
        if (p.is_real && DIR == BWD) {
            const int half = n >> 1, per = half + 1;
            for (int id = threadIdx.x; id < g_here * per; id += blockDim.x) {
                int g = fdiv(id, per, inv_per), k = id - g * per;
                const int eb = g * n;
                if (k == 0) {
                    cx<T> a = z[gpad(eb)];
                    z[gpad(eb)] = mk<T>(a.x + a.y, a.x - a.y);
                } else if (k == half) {
                    cx<T> a = z[gpad(eb + half)];
                    z[gpad(eb + half)] = mk<T>((T)2 * a.x, (T)-2 * a.y);
                } else {
                    cx<T> A = z[gpad(eb + k)], B = conj(z[gpad(eb + n - k)]);
                    cx<T> S = A + B, Dm = cmulc(A - B, twr[k]);  // (A-B) * conj(W_N^k)
                    cx<T> D = mk<T>(-Dm.y, Dm.x);                // * i
                    z[gpad(eb + k)] = S + D;
                    z[gpad(eb + n - k)] = conj(S - D);
                }
            }
            __syncthreads();
        }
        // ---- C: in-place DIF passes ----
        {
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
        }
        // ---- Q: real forward post-processing: X[k] = S + D, X[n-k] = conj(S - D),
        //         S = (A+B)/2, D = -(i/2) W_N^k (A-B), A = Z[k], B = conj Z[n-k] ----
        if (p.is_real && DIR == FWD) {
            const int half = n >> 1, per = half + 1;
            for (int id = threadIdx.x; id < g_here * per; id += blockDim.x) {
                int g = fdiv(id, per, inv_per), k = id - g * per;
                const int eb = g * n;
                if (k == 0) {
                    cx<T> a = z[gpad(eb)];
                    z[gpad(eb)] = mk<T>(a.x + a.y, a.x - a.y);  // (DC, Nyquist): include/pffft/pffft.h:144-152
                } else if (k == half) {
                    int pk = gpad(eb + POS(half));
                    z[pk] = conj(z[pk]);
                } else {
                    int pk = gpad(eb + POS(k)), pn = gpad(eb + POS(n - k));
                    cx<T> A = z[pk], B = conj(z[pn]);
                    cx<T> S = (A + B) * (T)0.5, Dm = cmul(A - B, twr[k]) * (T)0.5;
                    cx<T> D = mk<T>(Dm.y, -Dm.x);  // * (-i)
                    z[pk] = S + D;
                    z[pn] = conj(S - D);
                }
            }
            __syncthreads();
        }
        // ---- S: store ----
        for (int iv = threadIdx.x; iv < totv; iv += blockDim.x) {
            int g = fdiv(iv, nv, inv_nv), v = iv - g * nv;
            const int eb = g * n;
            vec4<T> val;
            if (!out_internal) {
                cx<T> a = z[gpad(eb + POS(2 * v))], b = z[gpad(eb + POS(2 * v + 1))];
                val.x = a.x; val.y = a.y; val.z = b.x; val.w = b.y;
            } else {
                const int part = v & 1;
                val.x = zs[2 * gpad(eb + POS(bin_of(v, 0, n, p.is_real))) + part];
                val.y = zs[2 * gpad(eb + POS(bin_of(v, 1, n, p.is_real))) + part];
                val.z = zs[2 * gpad(eb + POS(bin_of(v, 2, n, p.is_real))) + part];
                val.w = zs[2 * gpad(eb + POS(bin_of(v, 3, n, p.is_real))) + part];
            }
            __builtin_nontemporal_store(val, out4 + t0 * nv + iv);
        }
        __syncthreads();
        g0 = gn;
    }
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
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
