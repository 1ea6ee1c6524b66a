// The smallest transforms (n = 16 or 32 complex points: complex N = 16, 32; real N = 32, 64 — the reference's minimum
// sizes, src/pffft_priv_impl.h:78-89) with ONE THREAD PER TRANSFORM.
//
// Same reference functions as fft_tiled.h.  At these sizes a vector is 128-512 bytes: any scheme that spreads a
// transform over lanes spends its time exchanging (the wave-local Stockham kernel that served them: 0.50-0.70 of the
// roofline).  Here a wavefront moves 64 consecutive vectors: 16-byte loads linear over the 64 vectors (fully coalesced),
// one transposing trip through a padded LDS image so that lane L ends up with vector L in registers, the whole transform
// as ONE in-register butterfly (dft16 / dft32, no twiddles at all), the real pair pass and the pffft-internal layout as
// compile-time register renaming, and the mirror-image trip back.  No workgroup barrier, no twiddle table except the
// n/2 pair-pass factors W_N^k of real transforms (wave-uniform scalar loads).
#pragma once
#include "cxmath.h"

namespace pf {

// (bin, part) of scalar i of the internal layout (fft_generic.h bin_of), as compile-time functions
template <int n, int REAL> __host__ __device__ constexpr int tiny_bin(int i) {
    const int n4 = n / 4, b = i / 32, q = (i % 32) / 8, l = i % 4, t = 4 * b + l;
    return (REAL && (q & 1)) ? q * n4 + (t ? n4 - t : 0) : q * n4 + t;
}
__host__ __device__ constexpr int tiny_part(int i) { return (i % 8) / 4; }

template <typename T, int n, int DIR, int REAL, int IN_INT, int OUT_INT>
__global__ void __launch_bounds__(256)
fft_tiny_kernel(const T* in, T* out, size_t batch, const cx<T>* __restrict__ twrg) {
    typedef cx<T> CX;
    typedef vec4<float> chunk16;
    constexpr int CH = 16 / (int)sizeof(T);            // scalars per 16-byte chunk
    constexpr int CPV = 2 * n / CH;                    // chunks per vector
    constexpr int ROW = CPV + 1;                       // padded row of the per-wave image: conflict-free both ways
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    chunk16* img = reinterpret_cast<chunk16*>(smem_raw) + (size_t)wave * 64 * ROW;
    const chunk16* in16 = reinterpret_cast<const chunk16*>(in);
    chunk16* out16 = reinterpret_cast<chunk16*>(out);
    const size_t ngroups = (batch + 63) / 64;
    // pair-pass factors W_N^k, k = 1 .. n/2 - 1 (uniform)
    CX wk[REAL ? n / 2 : 1];
    if constexpr (REAL != 0) {
#pragma unroll
        for (int k = 1; k < n / 2; ++k) wk[k] = twrg[k];
    }
    const size_t last_chunk = batch * (size_t)CPV - 1;
    const size_t gstride = (size_t)gridDim.x * waves;
    chunk16 r[CPV];                                    // the NEXT group's chunks travel in registers while this one is transformed
    auto load_group = [&](size_t grp) {                // clamped: unconditional loads
        const size_t c0 = grp * 64 * CPV + lane;
#pragma unroll
        for (int i = 0; i < CPV; ++i) {
            const size_t c = c0 + 64 * (size_t)i;
            r[i] = __builtin_nontemporal_load(in16 + (c < last_chunk ? c : last_chunk));
        }
    };
    {
        const size_t g0 = (size_t)blockIdx.x * waves + wave;
        load_group(g0 < ngroups ? g0 : ngroups - 1);
    }
    for (size_t grp = (size_t)blockIdx.x * waves + wave; grp < ngroups; grp += gstride) {
        const size_t v0 = grp * 64;
        const int cnt = (int)((batch - v0) < 64 ? (batch - v0) : 64);
        const int tot = cnt * CPV;
        // ---- in: the prefetched 16-byte chunks -> image rows
#pragma unroll
        for (int i = 0; i < CPV; ++i) {
            const int c = lane + 64 * i;
            img[(c / CPV) * ROW + (c % CPV)] = r[i];
        }
        load_group(grp + gstride < ngroups ? grp + gstride : ngroups - 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        T s[2 * n];                                    // the vector as scalars, in its input order
#pragma unroll
        for (int i = 0; i < CPV; ++i) {
            const chunk16 c = img[lane * ROW + i];
            if constexpr (sizeof(T) == 4) { s[4 * i] = c.x; s[4 * i + 1] = c.y; s[4 * i + 2] = c.z; s[4 * i + 3] = c.w; }
            else { const vec2<double> d = __builtin_bit_cast(vec2<double>, c); s[2 * i] = (T)d.x; s[2 * i + 1] = (T)d.y; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- unpack
        CX a[n];
        if constexpr (IN_INT != 0) {
#pragma unroll
            for (int i = 0; i < 2 * n; ++i) {
                if (tiny_part(i) == 0) a[tiny_bin<n, REAL>(i)].x = s[i]; else a[tiny_bin<n, REAL>(i)].y = s[i];
            }
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j) a[j] = mk<T>(s[2 * j], s[2 * j + 1]);
        }
        // ---- transform
        if constexpr (REAL != 0 && DIR == BWD) {       // half-complex spectrum -> packed spectrum (fft_big.h real_pair_kernel)
            const CX x0 = a[0];
            a[0] = mk<T>(x0.x + x0.y, x0.x - x0.y);
            a[n / 2] = mk<T>((T)2 * a[n / 2].x, (T)-2 * a[n / 2].y);
#pragma unroll
            for (int k = 1; k < n / 2; ++k) {
                const CX A = a[k], Bc = conj(a[n - k]);
                const CX S = A + Bc, m = cmulc(A - Bc, wk[k]);
                const CX D = mk<T>(-m.y, m.x);
                a[k] = S + D;
                a[n - k] = conj(S - D);
            }
        }
        dftR<n, DIR>(a);
        if constexpr (REAL != 0 && DIR == FWD) {
            const CX z0 = a[0];
            a[0] = mk<T>(z0.x + z0.y, z0.x - z0.y);   // (DC, Nyquist): include/pffft/pffft.h:144-152
            a[n / 2] = conj(a[n / 2]);
#pragma unroll
            for (int k = 1; k < n / 2; ++k) {
                const CX A = a[k], Bc = conj(a[n - k]);
                const CX S = (A + Bc) * (T)0.5, m = cmul((A - Bc) * (T)0.5, wk[k]);
                const CX D = mk<T>(m.y, -m.x);
                a[k] = S + D;
                a[n - k] = conj(S - D);
            }
        }
        // ---- pack
        if constexpr (OUT_INT != 0) {
#pragma unroll
            for (int i = 0; i < 2 * n; ++i) s[i] = tiny_part(i) == 0 ? a[tiny_bin<n, REAL>(i)].x : a[tiny_bin<n, REAL>(i)].y;
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j) { s[2 * j] = a[j].x; s[2 * j + 1] = a[j].y; }
        }
#pragma unroll
        for (int i = 0; i < CPV; ++i) {
            chunk16 c;
            if constexpr (sizeof(T) == 4) { c.x = s[4 * i]; c.y = s[4 * i + 1]; c.z = s[4 * i + 2]; c.w = s[4 * i + 3]; }
            else { vec2<double> d; d.x = s[2 * i]; d.y = s[2 * i + 1]; c = __builtin_bit_cast(chunk16, d); }
            img[lane * ROW + i] = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- out: image rows -> linear 16-byte stores
        {
            chunk16* dst = out16 + v0 * CPV;
#pragma unroll
            for (int i = 0; i < CPV; ++i) {
                const int c = lane + 64 * i;
                if (c < tot) __builtin_nontemporal_store(img[(c / CPV) * ROW + (c % CPV)], dst + c);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace pf
