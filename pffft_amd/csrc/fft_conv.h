// Fused spectral convolution for the register-tiled power-of-two sizes:   out = backward( forward(in) . H ) * scaling
// in ONE kernel - one HBM read of the vector, one write - instead of three launches and seven vector passes.
//
// Reference: the sequence every FFT convolution with pffft runs (README "convolution" use; src/pffastconv.c:235-254 is one
// instance): pffft_transform(FORWARD) :1465-1532, pffft_zconvolve_no_accu / _accumulate :1534-1684, pffft_transform(BACKWARD).
// H is what pffft_transform(…, PFFFT_FORWARD) produced for the filter: the INTERNAL (unordered) layout, one vector for the whole
// batch.  It is staged once per workgroup through the block image of fft_tiled.h (the same scalar picks as a transform that
// reads the internal layout) and then lives in registers for the whole persistent loop, scaled.
//
// Built from the two Tiled<> halves exactly like the FIR block kernels (fft_fir.h): because the spectrum-side stage of both
// directions uses the same butterfly assignment (R0 == RL; real transforms: the symmetric one), the bins a thread holds after
// the forward transform are the first-stage operands of the inverse - forward pair pass, product and backward pair pass run in
// registers, no layout and no exchange between the transforms.  Loads, work distribution and stores are those of
// fft_tiled_kernel (persistent workgroups, in-order pull, next vector in flight).
#pragma once
#include "fft_tiled.h"

namespace pf {

template <class C, int REAL>
__global__ void __launch_bounds__(C::WG_THREADS, C::WG_THREADS >= 1024 ? 4 : C::OCC)
fft_conv_kernel(const typename C::real_t* in, const typename C::real_t* __restrict__ H, typename C::real_t* out, unsigned batch,
                typename C::real_t scaling, int accumulate,
                const cx<typename C::real_t>* __restrict__ twg, const cx<typename C::real_t>* __restrict__ twrg, unsigned* ctr) {
    typedef typename C::real_t T;
    typedef cx<T> CX;
    typedef Tiled<C, FWD, REAL> KF;
    typedef Tiled<C, BWD, REAL> KB;
    typedef typename KF::S0 S0;
    typedef ChunkOps<T> CO;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, VEC = C::VEC, CH = C::CH, NCH = C::NCH, NS = C::NS;
    constexpr int R0 = C::rad(0), RL = C::rad(NS - 1);
    static_assert(R0 == RL, "the forward transform must leave the bins where the inverse takes its first operands");
    static_assert(VEC == 1 || S0::PAIR, "float configs need an even butterfly count in the first / last stage");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* tab = reinterpret_cast<CX*>(smem_raw);  // W_n^j table (TWMODE 1), else unused
    CX* img = reinterpret_cast<CX*>(smem_raw + C::TABLE_BYTES) + (size_t)slot * C::IMG;
    T* imgs = reinterpret_cast<T*>(img);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + C::TABLE_BYTES + (size_t)C::T_PER_WG * C::IMG * sizeof(CX));

    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, t, twg, twrg);
    KB::load_tw(wb, t, twg, twrg);
    const CX* twt = twg;
    if constexpr (C::TWMODE == 1) {
        for (int i = threadIdx.x; i < n; i += C::WG_THREADS) tab[i] = twg[i];
        twt = tab;
    }
    // ---- the filter spectrum of this thread's bins: linear 16-byte chunks of the internal layout into the padded block image
    //      (every slot stages its own copy), then the scalar picks of ipos (fft_tiled.h), times `scaling`
    CX h[E];
    {
        const chunk16* H16 = reinterpret_cast<const chunk16*>(H);
        chunk16* im16 = reinterpret_cast<chunk16*>(imgs);
        constexpr int CPB = 32 / CH;  // chunks per 32-scalar block
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = t + TPT * i;
            im16[(c / CPB) * (C::IBS / CH) + (c % CPB)] = H16[c];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E / RL; ++u)
#pragma unroll
            for (int d = 0; d < RL; ++d) {
                const int ip = KF::template ipos<RL>(KF::template jm<NS - 1>(t, u), d);
                h[u * RL + d] = mk<T>(imgs[ip] * scaling, imgs[ip + 4] * scaling);
            }
        __syncthreads();
    }

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
        KF::load_raw(raw, in + (t0 < last ? t0 : last) * 2 * (size_t)n, t, true);
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
        // ---- input: first-stage operand order straight from the raw chunks
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
        // ---- forward transform
        KF::template butterflies<0>(v, t, wf, twt);
        if constexpr (NS > 1) KF::template xwrite<0>(v, t, img);
        __syncthreads();  // publishes s_next; first half of exchange 0
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        if constexpr (C::PREFETCH) {  // the loads of the next vector fly while this one is finished
            const size_t tn = (size_t)gn * C::T_PER_WG + slot;
            KF::load_raw(raw, in + (tn < last ? tn : last) * 2 * (size_t)n, t, true);
        }
        if constexpr (NS > 1) { KF::template xread<0>(v, t, img); KF::xsync(); KF::template butterflies<1>(v, t, wf, twt); }
        if constexpr (NS > 2) { KF::template xwrite<1>(v, t, img); KF::xsync(); KF::template xread<1>(v, t, img); KF::xsync(); KF::template butterflies<2>(v, t, wf, twt); }
        if constexpr (NS > 3) { KF::template xwrite<2>(v, t, img); KF::xsync(); KF::template xread<2>(v, t, img); KF::xsync(); KF::template butterflies<3>(v, t, wf, twt); }
        if constexpr (NS > 4) { KF::template xwrite<3>(v, t, img); KF::xsync(); KF::template xread<3>(v, t, img); KF::xsync(); KF::template butterflies<4>(v, t, wf, twt); }
        // ---- spectrum . H  (real: packed spectrum -> X[k] first; bin 0 carries (DC, Nyquist): two real products,
        //      src/pffft_priv_impl.h:1626-1629 / :1680-1683)
        if constexpr (REAL) KF::pair_regs(v, t, wf);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const CX p = cmul(v[i], h[i]);
            if (REAL && i == 0) v[0] = KF::sel(t == 0, mk<T>(v[0].x * h[0].x, v[0].y * h[0].y), p);
            else v[i] = p;
        }
        if constexpr (REAL) KB::pair_regs(v, t, wb);
        // ---- backward transform (its first-stage operands are in place)
        KB::template butterflies<0>(v, t, wb, twt);
        if constexpr (NS > 1) { KB::template xwrite<0>(v, t, img); KB::xsync(); KB::template xread<0>(v, t, img); KB::xsync(); KB::template butterflies<1>(v, t, wb, twt); }
        if constexpr (NS > 2) { KB::template xwrite<1>(v, t, img); KB::xsync(); KB::template xread<1>(v, t, img); KB::xsync(); KB::template butterflies<2>(v, t, wb, twt); }
        if constexpr (NS > 3) { KB::template xwrite<2>(v, t, img); KB::xsync(); KB::template xread<2>(v, t, img); KB::xsync(); KB::template butterflies<3>(v, t, wb, twt); }
        if constexpr (NS > 4) { KB::template xwrite<3>(v, t, img); KB::xsync(); KB::template xread<3>(v, t, img); KB::xsync(); KB::template butterflies<4>(v, t, wb, twt); }
        // ---- output: the last stage's natural order, 16-byte units in lane order (accumulate: out += result)
        if (active) {
            chunk16* d16 = reinterpret_cast<chunk16*>(dst);
            if constexpr (VEC == 2) {
#pragma unroll
                for (int ii = 0; ii < S0::B / 2; ++ii)
#pragma unroll
                    for (int d = 0; d < RL; ++d) {
                        const CX a = v[(2 * ii) * RL + d], b = v[(2 * ii + 1) * RL + d];
                        chunk16 x; x.x = a.x; x.y = a.y; x.z = b.x; x.w = b.y;
                        chunk16* p = d16 + t + TPT * ii + d * (n / (2 * RL));
                        if (accumulate) { const chunk16 o = *p; x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w; }
                        __builtin_nontemporal_store(x, p);
                    }
            } else {
#pragma unroll
                for (int u = 0; u < E / RL; ++u)
#pragma unroll
                    for (int d = 0; d < RL; ++d) {
                        chunk16* p = d16 + t + TPT * u + d * (n / RL);
                        T re = v[u * RL + d].x, im = v[u * RL + d].y;
                        if (accumulate) { const chunk16 o = *p; re += CO::get(o, 0); im += CO::get(o, 1); }
                        chunk16 x;
                        CO::set(x, 0, re); CO::set(x, 1, im);
                        __builtin_nontemporal_store(x, p);
                    }
            }
        }
        KB::xsync();
        if constexpr (!C::PREFETCH) {
            const size_t tn = (size_t)gn * C::T_PER_WG + slot;
            KF::load_raw(raw, in + (tn < last ? tn : last) * 2 * (size_t)n, t, true);
        }
        g = gn;
    }
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// configurations: the register-tiled ones of fft_tiled.h with one base twiddle per butterfly in registers (two transforms and
// the filter share the register file: no table in LDS, the next vector prefetched only where 16 points per thread leave room)
template <typename T> struct ConvPick;
template <> struct ConvPick<float> {
    typedef TiledCfg<float, 4, 2, 2, 4, 4, 1, 1, 1, 0, 0, 1> C16;
    typedef TiledCfg<float, 5, 4, 3, 4, 2, 4, 1, 1, 0, 0, 1> C32;
    typedef TiledCfg<float, 6, 4, 2, 8, 8, 1, 1, 1, 0, 3, 1> C64;
    typedef TiledCfg<float, 7, 8, 3, 8, 2, 8, 1, 1, 0, 3, 1> C128;
    typedef TiledCfg<float, 8, 16, 3, 8, 4, 8, 1, 2, 0, 3, 1> C256;
    typedef TiledCfg<float, 9, 32, 3, 8, 8, 8, 1, 4, 4, 3, 1> C512;
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 1> C1024;
    typedef TiledCfg<float, 11, 128, 4, 8, 4, 8, 8, 4, 8, 3, 1> C2048;
    typedef TiledCfg<float, 12, 256, 4, 8, 8, 8, 8, 4, 8, 3, 0> C4096;
    typedef TiledCfg<float, 13, 512, 4, 8, 8, 16, 8, 4, 8, 3, 0> C8192;
};
// double: the filter and a transform's points fill the register file; the W_n^j table sits in LDS (TWMODE 1) and nothing is prefetched
template <> struct ConvPick<double> {
    typedef TiledCfg<double, 4, 2, 2, 4, 4, 1, 1, 1, 0, 0, 0, 256> C16;
    typedef TiledCfg<double, 5, 4, 3, 4, 2, 4, 1, 1, 0, 0, 0, 256> C32;
    typedef TiledCfg<double, 6, 4, 2, 8, 8, 1, 1, 1, 0, 1, 0, 256> C64;
    typedef TiledCfg<double, 7, 8, 3, 8, 2, 8, 1, 1, 0, 1, 0, 256> C128;
    typedef TiledCfg<double, 8, 16, 3, 8, 4, 8, 1, 2, 0, 1, 0, 256> C256;
    typedef TiledCfg<double, 9, 32, 3, 8, 8, 8, 1, 4, 0, 1, 0> C512;
    typedef TiledCfg<double, 10, 64, 3, 8, 16, 8, 1, 4, 0, 1, 0> C1024;
    typedef TiledCfg<double, 11, 128, 4, 8, 4, 8, 8, 4, 0, 1, 0, 256> C2048;
    typedef TiledCfg<double, 12, 256, 4, 8, 8, 8, 8, 4, 0, 1, 0, 256> C4096;
};

}  // namespace pf
