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
//                     4-scalar group v = 64 s + L of the pffft layout (fft_generic.h bin_of) — again 1 KiB / instr.
// Backward = conjugated twiddles and the mirror-image X0 exchange when the input is in internal layout.
//
// Work distribution (the part that decides the HBM rate, tools/membench.hip): persistent workgroups of
// 8 waves pull the next group of 8 consecutive transforms from an atomic counter, so the whole chip
// sweeps the batch IN ORDER through a window of a few MiB (DRAM-page friendly; a copy with this
// pattern reaches 6.9 TB/s where a static grid-stride assignment of the same chunks reaches 5.3),
// and every wave issues the loads of its NEXT transform before it finishes the current one.
//
// LDS: 8960 B per wave (exchange images only; twiddles live in registers).  Row strides (136 complex) and
// the X2 pair swizzle / X3 plane paddings come from tools/lds_sim.py searches (bank-conflict free
// except the X2 b128 read, 2x).  A wave only talks to itself through LDS; the one __syncthreads per
// iteration only publishes the next work index.
#pragma once
#include "cxmath.h"
#include "fft_tiled.h"   // lds_ld_c

namespace pf {

constexpr int C1024_WAVES = 8;                 // waves per workgroup
constexpr int C1024_S1 = 136;                  // complex row stride of the X1/X2 images
constexpr int C1024_PLANE = 1120;              // floats per re/im plane of the X0/X3 image
constexpr int C1024_WAVE_BYTES = 2 * C1024_PLANE * 4;  // 8960 >= 8*136*8 = 8704
constexpr int C1024_LDS_BYTES = C1024_WAVES * C1024_WAVE_BYTES + 16;

// float index inside the wave image of the split re/im planes.  Two paddings (tools/search_x0.py): the X3
// image (b64 writes, b128 reads) is conflict-free with plane 1120 / +16 per 256 bins; the X0 image (b128
// writes, b64 reads) with plane 1116 / +8 — the X3 padding costs it 4x on the writes.
template <int X0> __device__ __forceinline__ int c1024_plane_addr(int part, int k) {
    return X0 ? part * 1116 + k + 8 * (k >> 8) : part * C1024_PLANE + k + 16 * (k >> 8);
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef vec4<float> C1024V4;

__device__ __forceinline__ void c1024_load(C1024V4 (&raw)[8], const float* in, size_t t, int L) {
    const C1024V4* src = reinterpret_cast<const C1024V4*>(in) + t * 512 + L;
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = __builtin_nontemporal_load(src + 64 * j);
}

// part A: consume the 8 loaded float4, S1, write the X1 image.  After it `raw` is dead.
// MIX (forward, canonical input only): the batch is one stream of samples that is frequency-shifted before the
// transform, x'[g] = x[g] exp(j 2 pi (phase0 + step g)), g = 1024 t + 128 j + 2L + e (the mixer
// fused into the load stage).  The phasor factors into  F(t) * G[j] * R[L,e]:  R does not depend on j, commutes
// with the radix-8 over j and is folded into the S1 twiddles (w1 *= R, plus `r0` for the k1 = 0 output);
// G[j] = exp(j 2 pi step 128 j) is a kernel argument (scalar registers); F(t) is one sincos per transform.
struct C1024Mix {
    double step;       // turns per sample
    double phase0;     // turns at sample 0 of the batch
    float g[7][2];     // G[1..7], computed in double on the host
};

struct C1024NoMix {};
template <int MIX> struct C1024MixArg { typedef C1024NoMix type; };
template <> struct C1024MixArg<1> { typedef C1024Mix type; };
// per-wave state of the fused mixer: R[L,e] (registers, set once) and F(t) (per transform)
struct C1024MixState { cx<float> r0[2]; cx<float> f0; };

template <int DIR, int IN_INTERNAL, int MIX = 0>
__device__ __forceinline__ void c1024_part_a(const C1024V4 (&raw)[8], cx<float>* wl, float* wf,
                                             const cx<float> (&w1)[7][2], int L,
                                             const C1024MixState* ms = nullptr, const C1024Mix* mix = nullptr) {
    typedef cx<float> C;
    typedef C1024V4 V4;
    C a[8][2];
    if (!IN_INTERNAL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j][0] = mk<float>(raw[j].x, raw[j].y);
            a[j][1] = mk<float>(raw[j].z, raw[j].w);
        }
        if (MIX) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const C fj = j == 0 ? ms->f0 : cmul(ms->f0, mk<float>(mix->g[j - 1][0], mix->g[j - 1][1]));
                a[j][0] = cmul(a[j][0], fj);
                a[j][1] = cmul(a[j][1], fj);
            }
        }
    } else {
        // X0: the internal layout arrives in linear order; scatter into re/im planes, read back canonical
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int k0 = 256 * ((L >> 1) & 3) + 4 * (8 * s + (L >> 3));
            *reinterpret_cast<V4*>(wf + c1024_plane_addr<1>(L & 1, k0)) = raw[s];
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = 128 * j + 2 * L;
            vec2<float> re = *reinterpret_cast<const vec2<float>*>(wf + c1024_plane_addr<1>(0, k));
            vec2<float> im = *reinterpret_cast<const vec2<float>*>(wf + c1024_plane_addr<1>(1, k));
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
        if (MIX) a[0][e] = cmul(b[0], ms->r0[e]); else a[0][e] = b[0];
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) a[k1][e] = twmul<DIR>(b[k1], w1[k1 - 1][e]);
    }
    // ---- X1 write ----
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        V4 v; v.x = a[k1][0].x; v.y = a[k1][0].y; v.z = a[k1][1].x; v.w = a[k1][1].y;
        *reinterpret_cast<V4*>(wl + k1 * C1024_S1 + 2 * L) = v;
    }
    wave_lds_fence();
}

template <int Q> __device__ __forceinline__ void c1024_rd16(cx<float> (&m)[16], const cx<float>* rd) {
    m[Q] = lds_ld_c<Q * 64>(rd);
    if constexpr (Q + 1 < 16) c1024_rd16<Q + 1>(m, rd);
}

// part B: X1 read, S2, X2, S3, store
template <int DIR, int OUT_INTERNAL>
__device__ __forceinline__ void c1024_part_b(float* out, size_t t, cx<float>* wl, float* wf,
                                             const cx<float> (&w2)[15], int L) {
    typedef cx<float> C;
    typedef C1024V4 V4;
    V4* dst = reinterpret_cast<V4*>(out) + t * 512 + L;
    C m[16];
    {
        const C* rd = wl + (L >> 3) * C1024_S1 + (L & 7);
        c1024_rd16<0>(m, rd);                                     // sixteen single ds_read_b64 (fft_tiled.h lds_ld_c: no ds_read2 pairing)
    }
    wave_lds_fence();
    // ---- S2: radix-16 over a, twiddle W128^(c * ka) = W1024^(8 c ka) ----
    dft16<DIR>(m);
#pragma unroll
    for (int ka = 1; ka < 16; ++ka) m[ka] = twmul<DIR>(m[ka], w2[ka - 1]);
    // ---- X2: image (k1, ka, c) at k1*S1 + 8 ka + 2*((c>>1) ^ (ka&3)) + (c&1) ----
    {
        const int k1 = L >> 3, c = L & 7;
#pragma unroll
        for (int ka = 0; ka < 16; ++ka)
            wl[k1 * C1024_S1 + 8 * ka + 2 * ((c >> 1) ^ (ka & 3)) + (c & 1)] = m[ka];
    }
    wave_lds_fence();
    C a[8][2];
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
            __builtin_nontemporal_store(v, dst + 64 * kc);
        }
    } else {
        wave_lds_fence();
        // ---- X3: canonical -> internal layout through split planes ----
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            int k = 2 * L + 128 * kc;
            vec2<float> re, im;
            re.x = a[kc][0].x; re.y = a[kc][1].x; im.x = a[kc][0].y; im.y = a[kc][1].y;
            *reinterpret_cast<vec2<float>*>(wf + c1024_plane_addr<0>(0, k)) = re;
            *reinterpret_cast<vec2<float>*>(wf + c1024_plane_addr<0>(1, k)) = im;
        }
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int k0 = 256 * ((L >> 1) & 3) + 4 * (8 * s + (L >> 3));
            V4 v = *reinterpret_cast<const V4*>(wf + c1024_plane_addr<0>(L & 1, k0));
            __builtin_nontemporal_store(v, dst + 64 * s);
        }
    }
    wave_lds_fence();
}

// The 29 twiddles a lane needs depend only on its lane id: W1024^(k1 (2L+e)) for S1 and
// W1024^(8 (L&7) ka) for S2.  They are fetched once per wave and live in registers (58 VGPRs) for
// the whole persistent loop: no twiddle table in LDS, no LDS latency in the butterflies.
__device__ __forceinline__ void c1024_load_twiddles(const cx<float>* __restrict__ twg, int L, cx<float> (&w1)[7][2],
                                                    cx<float> (&w2)[15]) {
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
        w1[k1 - 1][0] = twg[k1 * (2 * L)];
        w1[k1 - 1][1] = twg[k1 * (2 * L + 1)];
    }
#pragma unroll
    for (int ka = 1; ka < 16; ++ka) w2[ka - 1] = twg[8 * (L & 7) * ka];
}

// ---- dynamic in-order distribution + prefetch (default) ----
// ctr[0] = next group of 8 transforms, ctr[1] = workgroups finished (the last one re-arms both)
// exp(j 2 pi frac(turns)): the reduction in double, the residue of the float conversion as a first-order rotation
__device__ __forceinline__ cx<float> c1024_unit(double turns) {
    turns -= rint(turns);
    const float th = (float)turns;
    const float d = 6.28318530717958647692f * (float)(turns - (double)th);
    float sn, cs;
    sincospif(2.0f * th, &sn, &cs);
    return mk<float>(cs - sn * d, sn + cs * d);
}

template <int DIR, int IN_INTERNAL, int OUT_INTERNAL, int MIX>
__device__ __forceinline__ void c1024_dyn_body(const float* in, float* out, unsigned batch, const cx<float>* __restrict__ twg,
                                               unsigned* ctr, const typename C1024MixArg<MIX>::type& mix) {
    typedef cx<float> C;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    char* wbase = smem_raw + wave * C1024_WAVE_BYTES;
    C* wl = reinterpret_cast<C*>(wbase);
    float* wf = reinterpret_cast<float*>(wbase);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + C1024_WAVES * C1024_WAVE_BYTES);

    C w1[7][2], w2[15];
    c1024_load_twiddles(twg, L, w1, w2);
    C1024MixState ms;
    if constexpr (MIX != 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ms.r0[e] = c1024_unit(mix.step * (double)(2 * L + e));
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) w1[k1 - 1][e] = cmul(w1[k1 - 1][e], ms.r0[e]);
        }
    }
    unsigned pend = 0;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    unsigned g = blockIdx.x;
    const size_t last = (size_t)batch - 1;
    C1024V4 raw[8];
    {   // clamped: always a valid address, so the loads are unconditional (no phi copies, no early waits)
        size_t t0 = (size_t)g * C1024_WAVES + wave;
        c1024_load(raw, in, t0 < last ? t0 : last, L);
    }
    for (unsigned it = 0; (size_t)g * C1024_WAVES < batch; ++it) {
        if (threadIdx.x == 0) {  // publish the index of iteration it+1, grab the one of it+2
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        const size_t t = (size_t)g * C1024_WAVES + wave;
        const bool active = t < batch;  // wave-uniform
        if constexpr (MIX != 0) {
            ms.f0 = c1024_unit(mix.phase0 + mix.step * (double)(1024ull * (unsigned long long)(active ? t : 0)));
            c1024_part_a<DIR, IN_INTERNAL, 1>(raw, wl, wf, w1, L, &ms, &mix);
        } else {
            c1024_part_a<DIR, IN_INTERNAL>(raw, wl, wf, w1, L);
        }
        __syncthreads();
        const unsigned gn = s_next[(it + 1) & 1];
        const size_t tn = (size_t)gn * C1024_WAVES + wave;
        c1024_load(raw, in, tn < last ? tn : last, L);  // in flight while this transform finishes
        if (active) c1024_part_b<DIR, OUT_INTERNAL>(out, t, wl, wf, w2, L);
        g = gn;
    }
    if (threadIdx.x == 0) {
        __threadfence();  // my last (unused) grab has landed before I report done
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

template <int DIR, int IN_INTERNAL, int OUT_INTERNAL>
__global__ void __launch_bounds__(C1024_WAVES * 64, 2)
fft_c1024_f32_dyn_kernel(const float* in, float* out, unsigned batch, const cx<float>* __restrict__ twg,
                         unsigned* ctr) {
    c1024_dyn_body<DIR, IN_INTERNAL, OUT_INTERNAL, 0>(in, out, batch, twg, ctr, C1024NoMix());
}

// forward transform of the frequency-shifted stream (C1024Mix above)
template <int OUT_INTERNAL>
__global__ void __launch_bounds__(C1024_WAVES * 64, 2)
fft_c1024_f32_mix_kernel(const float* in, float* out, unsigned batch, const cx<float>* __restrict__ twg,
                         unsigned* ctr, C1024Mix mix) {
    c1024_dyn_body<FWD, 0, OUT_INTERNAL, 1>(in, out, batch, twg, ctr, mix);
}

// ---- short launches: ONE transform per wavefront, W wavefronts per workgroup, workgroups in hardware dispatch order ----
// A launch of a few transforms per resident wavefront never reaches the steady state the persistent loop is built for: its 256
// workgroups of 8 wavefronts hold 64 KiB of loads per CU in flight and every lane opens with its 29 twiddle loads before the first
// butterfly (batch 2^12: 24 us against 8.4 us of HBM time).  Here 4 workgroups x 4 wavefronts are resident per CU (35 KiB of LDS, 84
// VGPRs each), every wavefront requests its vector FIRST and its twiddles behind it (L1 / L2 hits after the first workgroup of a CU),
// and a retiring workgroup is replaced by the dispatcher.  Same part A / part B as the loop: bit-identical spectra whatever the launch
// shape (tests/test_gpu_round4.py).
template <int DIR, int IN_INTERNAL, int OUT_INTERNAL, int W>
__global__ void __launch_bounds__(W * 64, 4)
fft_c1024_f32_once_kernel(const float* in, float* out, unsigned batch, const cx<float>* __restrict__ twg) {
    typedef cx<float> C;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    const size_t t = (size_t)blockIdx.x * W + wave;
    if (t >= batch) return;                                   // wave-uniform; the kernel has no workgroup barrier
    char* wbase = smem_raw + wave * C1024_WAVE_BYTES;
    C* wl = reinterpret_cast<C*>(wbase);
    float* wf = reinterpret_cast<float*>(wbase);
    C1024V4 raw[8];
    c1024_load(raw, in, t, L);
    C w1[7][2], w2[15];
    c1024_load_twiddles(twg, L, w1, w2);
    c1024_part_a<DIR, IN_INTERNAL>(raw, wl, wf, w1, L);
    c1024_part_b<DIR, OUT_INTERNAL>(out, t, wl, wf, w2, L);
}

}  // namespace pf
