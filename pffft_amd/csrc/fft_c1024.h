// N = 1024 complex single-precision: one wavefront per transform, 16 points per lane.
//
// This is the headline kernel (BASELINE.json: "batched N=1024 cplx-float fwd").  It replaces the
// reference's whole forward/backward complex pipeline for this size — uninterleave, 4x passf4_ps,
// pffft_cplx_finalize, optional pffft_zreorder (src/pffft_priv_impl.h:1490-1499, :185-251,
// :1195-1237, :1181-1185) — with ONE pass over HBM: 8 KiB in, 8 KiB out per transform.
//
// Decomposition 1024 = 8 x 16 x 8, decimation in frequency, all butterflies in registers:
//   load   lane L gets float4 j (j = 0..7) of the vector: x[128 j + 2L + e], e = 0,1   (1 KiB / wave-instr)
//   S1     2 radix-8 butterflies over j  -> y[k1][r = 2L+e] * W1024^(k1 r)
//   X1     LDS exchange: write (k1, r) rows (b128), read lane (k1 = L>>3, c = L&7): r = 8a + c, a = 0..15
//   S2     1 radix-16 butterfly over a   -> z[k1][ka][c] * W128^(c ka)
//   X2     LDS exchange: write b64, read lane (k1 = 2(L&3)+e, ka = L>>2): c = 0..7 (b128)
//   S3     2 radix-8 butterflies over c  -> X[2L + e + 128 kc], kc = 0..7
//   store  canonical: float4 (X[2L+128kc], X[2L+1+128kc]) at 1 KiB stride   (1 KiB / wave-instr)
//          internal : X3 exchange through split re/im planes so that lane L, step s stores the
//                     4-scalar group v = 64 s + L of the pffft layout (cxmath/bin_of) — again 1 KiB / instr.
// Backward = conjugated twiddles and the mirror-image X0 exchange when the input is in internal layout.
//
// LDS: 8 KiB twiddle table W1024^j per workgroup + 8960 B per wave.  Row strides (136 complex) and
// the X2 pair swizzle / X3 plane paddings come from tools/lds_sim.py searches (bank-conflict free
// except the X2 b128 read, 2x).  No barriers in the loop: a wave only talks to itself.
#pragma once
#include "cxmath.h"

namespace pf {

constexpr int C1024_WAVES = 8;                 // waves per workgroup
constexpr int C1024_S1 = 136;                  // complex row stride of the X1/X2 images
constexpr int C1024_PLANE = 1120;              // floats per re/im plane of the X0/X3 image
constexpr int C1024_WAVE_BYTES = 2 * C1024_PLANE * 4;  // 8960 >= 8*136*8 = 8704
constexpr int C1024_LDS_BYTES = 8192 + C1024_WAVES * C1024_WAVE_BYTES;

__device__ __forceinline__ int c1024_plane_addr(int part, int k) {  // float index inside the wave image
    return part * C1024_PLANE + k + 16 * (k >> 8);
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int DIR, int IN_INTERNAL, int OUT_INTERNAL>
__global__ void __launch_bounds__(C1024_WAVES * 64, 4)
fft_c1024_f32_kernel(const float* in, float* out, unsigned batch, const cx<float>* __restrict__ twg) {
    typedef cx<float> C;
    typedef vec4<float> V4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* tw = reinterpret_cast<C*>(smem_raw);
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    char* wbase = smem_raw + 8192 + wave * C1024_WAVE_BYTES;
    C* wl = reinterpret_cast<C*>(wbase);
    float* wf = reinterpret_cast<float*>(wbase);

    for (int i = threadIdx.x; i < 1024; i += C1024_WAVES * 64) tw[i] = twg[i];
    __syncthreads();

    const unsigned nwaves = gridDim.x * C1024_WAVES;
    for (unsigned t = blockIdx.x * C1024_WAVES + wave; t < batch; t += nwaves) {
        const V4* src = reinterpret_cast<const V4*>(in) + (size_t)t * 512;
        V4* dst = reinterpret_cast<V4*>(out) + (size_t)t * 512;
        C a[8][2];
        if (!IN_INTERNAL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                V4 v = __builtin_nontemporal_load(src + 64 * j + L);
                a[j][0] = mk<float>(v.x, v.y);
                a[j][1] = mk<float>(v.z, v.w);
            }
        } else {
            // X0: linear load of the internal layout, scatter into re/im planes, read back canonical
            V4 v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = __builtin_nontemporal_load(src + 64 * s + L);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int k0 = 256 * ((L >> 1) & 3) + 4 * (8 * s + (L >> 3));
                *reinterpret_cast<V4*>(wf + c1024_plane_addr(L & 1, k0)) = v[s];
            }
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int k = 128 * j + 2 * L;
                vec2<float> re = *reinterpret_cast<const vec2<float>*>(wf + c1024_plane_addr(0, k));
                vec2<float> im = *reinterpret_cast<const vec2<float>*>(wf + c1024_plane_addr(1, k));
                a[j][0] = mk<float>(re.x, im.x);
                a[j][1] = mk<float>(re.y, im.y);
            }
            wave_lds_fence();
        }
        // ---- S1: radix-8 over j, twiddle W1024^(k1 * r) ----
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            C b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = a[j][e];
            dft8<DIR>(b);
            const int r = 2 * L + e;
            a[0][e] = b[0];
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) a[k1][e] = twmul<DIR>(b[k1], tw[k1 * r]);
        }
        // ---- X1 ----
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            V4 v; v.x = a[k1][0].x; v.y = a[k1][0].y; v.z = a[k1][1].x; v.w = a[k1][1].y;
            *reinterpret_cast<V4*>(wl + k1 * C1024_S1 + 2 * L) = v;
        }
        wave_lds_fence();
        C m[16];
        {
            const C* rd = wl + (L >> 3) * C1024_S1 + (L & 7);
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = rd[8 * q];
        }
        wave_lds_fence();
        // ---- S2: radix-16 over a, twiddle W128^(c * ka) = W1024^(8 c ka) ----
        dft16<DIR>(m);
        {
            const int c8 = 8 * (L & 7);
#pragma unroll
            for (int ka = 1; ka < 16; ++ka) m[ka] = twmul<DIR>(m[ka], tw[c8 * ka]);
        }
        // ---- X2: image (k1, ka, c) at k1*S1 + 8 ka + 2*((c>>1) ^ (ka&3)) + (c&1) ----
        {
            const int k1 = L >> 3, c = L & 7;
#pragma unroll
            for (int ka = 0; ka < 16; ++ka)
                wl[k1 * C1024_S1 + 8 * ka + 2 * ((c >> 1) ^ (ka & 3)) + (c & 1)] = m[ka];
        }
        wave_lds_fence();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k1 = 2 * (L & 3) + e, ka = L >> 2;
            const C* rd = wl + k1 * C1024_S1 + 8 * ka;
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                V4 v = *reinterpret_cast<const V4*>(rd + 2 * (cp ^ (ka & 3)));
                a[2 * cp][e] = mk<float>(v.x, v.y);
                a[2 * cp + 1][e] = mk<float>(v.z, v.w);
            }
        }
        // ---- S3: radix-8 over c -> X[2L + e + 128 kc] ----
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            C b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = a[q][e];
            dft8<DIR>(b);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q][e] = b[q];
        }
        if (!OUT_INTERNAL) {
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                V4 v; v.x = a[kc][0].x; v.y = a[kc][0].y; v.z = a[kc][1].x; v.w = a[kc][1].y;
                __builtin_nontemporal_store(v, dst + 64 * kc + L);
            }
        } else {
            wave_lds_fence();
            // ---- X3: canonical -> internal layout through split planes ----
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                int k = 2 * L + 128 * kc;
                vec2<float> re, im;
                re.x = a[kc][0].x; re.y = a[kc][1].x; im.x = a[kc][0].y; im.y = a[kc][1].y;
                *reinterpret_cast<vec2<float>*>(wf + c1024_plane_addr(0, k)) = re;
                *reinterpret_cast<vec2<float>*>(wf + c1024_plane_addr(1, k)) = im;
            }
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int k0 = 256 * ((L >> 1) & 3) + 4 * (8 * s + (L >> 3));
                V4 v = *reinterpret_cast<const V4*>(wf + c1024_plane_addr(L & 1, k0));
                __builtin_nontemporal_store(v, dst + 64 * s + L);
            }
        }
        wave_lds_fence();
    }
}

}  // namespace pf
