// Overlap-save FIR block kernel with 32 points per thread (round 6) - the throughput regime of pffastconv for filters beyond
// 850 taps (BASELINE configs[3]; reference: the block loop of pffastconv_apply, src/pffastconv.c:207-261, on 16384-sample internal
// blocks: what a caller observes is the number of samples produced and the values of the convolution).
//
// Why: the split kernel (fft_split.h) is bound by its LDS pipe and barriers - per 16384-sample block 448 KiB of LDS stores and
// 512 KiB of loads in SEVEN exchanges plus the landing buffer, 16 200 cycles per block and CU against an LDS floor of ~7 800
// (tools/dma_timeline.hip, profiles/r05_pmc.md: HBM traffic 1.004 x algorithmic at 0.30 of the roofline).  Here a block is ONE
// 256-thread workgroup with 32 points per thread, n = 8192 = 16 x 32 x 16 (fft_tiled.h, the configuration C3 runs on):
//   gather    16 x 16 bytes per thread straight from HBM into the stage-0 operands (no landing buffer, no DMA)
//   forward   radix 16 -> exchange -> radix 32 -> exchange -> radix 16 (symmetric butterfly assignment: bins k and n - k in one thread)
//   pair pass + x Hf / Nfft + pair pass in registers (no mirror exchange)
//   backward  radix 16 -> exchange -> radix 32 -> exchange -> radix 16 -> 16-byte stores of the block's first `numOut` samples
// FOUR exchanges per block (256 KiB of stores + 256 KiB of loads), TWO workgroups per CU (two barrier domains on four wavefronts
// each, 256 VGPRs per lane).  What made this organisation spill in rounds 2 and 4 (868 B of scratch per lane) and how it fits now:
//   * nothing of the filter is resident: pair pass, product and pair pass are folded into Z'[k] = A_k Z[k] + B_k conj Z[n - k] with per-bin
//     coefficients read per block from a thread-major table (fir32_coef_kernel: 32 coalesced 16-byte loads per thread, L2 hits) around
//     the last forward stage - 8 packed operations per pair of bins instead of 20, no pair-pass twiddles at all;
//   * thread 0 (both of its butterflies are self-mirrored) used to evaluate a second pairing and select (3 x 64 live registers):
//     now lane 0 permutes its 24 affected registers through 192 bytes of LDS so that the regular pairing applies to it too,
//     and back after the second pair pass; only slot 15 (bins 0 and n/2) keeps a select;
//   * with the spectrum dead, the NEXT block's samples are requested in four pieces between the phases of the inverse transform.
// Measured and dropped (tools/fir32_timeline.hip, min of 12 launches, 256 x 2^20 samples, 4096 taps): the 31 twiddles of the radix-32 stage
// from a 4 KiB LDS table instead of 30 products per transform: 0.428 against 0.441 - the LDS pipe is the scarcer resource here.
#pragma once
#include "fft_dma.h"

namespace pf {

struct Fir32 {
    typedef TiledCfg<float, 13, 256, 3, 16, 32, 16, 1, 2, 0, 3, 0, 256, 2> C;
    static constexpr int n = C::n, WG = 256, NB = n / 16;                        // NB: butterflies of the radix-16 stages
    static constexpr int IMG = (C::IMG_NAT > C::IMG_TRN ? C::IMG_NAT : C::IMG_TRN) + 8;   // points (no internal-layout image here)
    static constexpr size_t LDS_BYTES = (size_t)IMG * 8 + 32 * 8 + 16;           // image + lane 0's permutation scratch + next-group slots
    // thread 0 holds butterflies 0 and NB/2: register i < 16 = bin i NB, register 16 + i = bin NB/2 + i NB.  PERM[i] = the register whose
    // value sits in slot i while the pair passes run (so that slot d pairs with slot 31 - d like in every other thread)
    __host__ __device__ static constexpr int perm(int i) { return i < 8 ? 16 + i : i < 15 ? i - 7 : i == 15 ? 0 : i < 24 ? i - 8 : i; }
    // bin held by slot i of thread t during the pair passes / the product
    __host__ __device__ static constexpr int bin0(int r) { return r < 16 ? r * NB : NB / 2 + (r - 16) * NB; }   // thread 0, register r
    __host__ __device__ static int bin(int t, int i) {
        if (t == 0) return bin0(perm(i));
        return i < 16 ? t + i * NB : (NB - t) + (i - 16) * NB;
    }
};

// Folded per-bin coefficients, once per filter like the reference's own transform of the filter (src/pffastconv.c:108): real finalize,
// x Hf / Nfft and real preprocess of a mirror pair collapse into   Z'[k] = A_k Z[k] + B_k conj Z[n - k]   (derivation: fft_fir.h
// fastconv_part_kernel; the split kernels use the same form; bin 0 = (DC, Nyquist) and bin n/2 are their own mirrors) - 8 packed
// operations per pair of bins where pair pass + two products + pair pass took 20.  Thread-major: AB[(2 d + h) WG + t] = (A, B) of the bin
// in slot d (h = 0) / slot 31 - d (h = 1) of thread t: 32 coalesced 16-byte loads per thread and block, L2 hits.
__global__ void __launch_bounds__(Fir32::WG) fir32_coef_kernel(const cx<float>* __restrict__ Hc, const cx<float>* __restrict__ twr,
                                                               vec4<float>* __restrict__ AB) {
    typedef float T;
    typedef cx<T> CX;
    constexpr int n = Fir32::n;
    const int t = threadIdx.x;
    for (int i = 0; i < 32; ++i) {
        const int slot = (i & 1) ? 31 - (i >> 1) : (i >> 1);
        const int k = Fir32::bin(t, slot);
        const int km = (n - k) & (n - 1);
        const CX w = k <= n / 2 ? twr[k] : conj(twr[n - k]) * (T)-1;
        const CX Hk = Hc[k], Hm = Hc[km];
        const CX iw = mk<T>(-w.y, w.x), iwc = mk<T>(w.y, w.x);   // i w, i conj(w)
        const CX al = mk<T>(0.5f * (1.f - iw.x), -0.5f * iw.y), be = mk<T>(0.5f * (1.f + iw.x), 0.5f * iw.y);
        const CX ga = mk<T>(1.f + iwc.x, iwc.y), de = mk<T>(1.f - iwc.x, -iwc.y);
        const CX gH = cmul(ga, Hk), dHm = cmul(de, conj(Hm));
        CX a = cmul(gH, al) + cmul(dHm, be), b = cmul(gH, be) + cmul(dHm, al);
        if (k == 0) { a = mk<T>(Hk.x + Hk.y, 0.f); b = mk<T>(0.f, Hk.x - Hk.y); }
        if (k == n / 2) { a = mk<T>(2.f * Hk.x, -2.f * Hk.y); b = mk<T>(0.f, 0.f); }
        vec4<float> o; o.x = a.x; o.y = a.y; o.z = b.x; o.w = b.y;
        AB[i * Fir32::WG + t] = o;
    }
}

// (b.y m.y + c.x, -b.x m.y + c.y): the second half of b conj(m) on top of an accumulator (cxmath.h pk_mul_yx_yy_n as a fused multiply-add)
__device__ __forceinline__ vec2<float> pk_fma_yx_yy_n(vec2<float> b, vec2<float> m, vec2<float> c) {
    vec2<float> r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(b), "v"(m), "v"(c));
    return r;
}
// a z + b conj(m) in four packed operations
__device__ __forceinline__ cx<float> fir32_fold(cx<float> a, cx<float> z, cx<float> b, cx<float> m) {
    return pk_fma_xy_xx(b, m, pk_fma_yx_yy_n(b, m, pk_fma_xx_xy(a, z, pk_mul_yy_yx_n(a, z))));
}

// One 8-byte LDS read that the compiler cannot merge with its neighbour into ds_read2_b64: the pairs cost 8 LDS cycles per KiB where two
// ds_read_b64 cost 2 + 2 (MI355X_MICROARCH.md, LDS table: 128 against 256 B per clock and CU).  The compiler does not count these in
// lgkmcnt: every caller waits (wg_sync_raw) before it uses the values.
template <int OFF> __device__ __forceinline__ cx<float> lds_ld64_asm(const cx<float>* p) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    return lds_ld_c<OFF>(p);
}

template <int Q, int NQ, int STRIDE, int VOFF> struct Fir32Rd {
    static __device__ __forceinline__ void run(cx<float> (&v)[32], const cx<float>* p) {
        v[VOFF + Q] = lds_ld64_asm<Q * STRIDE>(p);
        if constexpr (Q + 1 < NQ) Fir32Rd<Q + 1, NQ, STRIDE, VOFF>::run(v, p);
    }
};

#ifdef PF_FIR32_DEBUG
__device__ long long pf_f32dbg[64];
#define PF_FSTAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0 && it == 3) pf_f32dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#elif defined(PF_FIR32_MARK)
#define PF_FSTAMP(i) asm volatile("; FIR32MARK %0" ::"n"(i))
#else
#define PF_FSTAMP(i) do { } while (0)
#endif

// PREF: the next block's samples are requested right after the product (else after the output stores)
template <int PREF>   // 0: after the stores; 1: all at once after the product; 2: in four pieces between the phases of the inverse transform
__global__ void __launch_bounds__(Fir32::WG, 2)
fastconv_fused32_kernel(const float* __restrict__ x, float* __restrict__ y, const vec4<float>* __restrict__ HP,
                        int nblk, int step, int inputLen, int lastOut,
                        const cx<float>* __restrict__ twg, const cx<float>* __restrict__ twrg, unsigned* ctr,
                        int nsig, size_t xstride, size_t ystride, int xmode) {
    typedef float T;
    typedef cx<T> CX;
    typedef Fir32::C C;
    typedef Tiled<C, FWD, 1> KF;
    typedef Tiled<C, BWD, 1> KB;
    constexpr int n = C::n, E = C::E, R = 16, WG = Fir32::WG;
    static_assert(E == 32 && C::rad(0) == 16 && C::rad(2) == 16 && C::NS == 3, "16 x 32 x 16 on 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    CX* img = reinterpret_cast<CX*>(smem_raw);
    CX* pscr = img + Fir32::IMG;                                  // lane 0's permutation scratch (32 points)
    unsigned* s_next = reinterpret_cast<unsigned*>(pscr + 32);
    const int t = threadIdx.x;
    const bool first = t == 0;
    const bool wave0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;

    // ---- work distribution: the first two groups of a workgroup are static, the counter hands out what follows (fft_tiled.h);
    //      xmode: every XCD works on CONTIGUOUS blocks - adjacent overlap-save blocks share taps - 1 samples and then meet in one L2
    //      (fft_fir.h xcd_local; ctr[0 .. 7] next per XCD, ctr[8] done)
    const bool dyn = ctr != nullptr;
    const long long nblk_all = (long long)nblk * nsig;
    const unsigned xl = xmode ? xcd_local(blockIdx.x, gridDim.x) : blockIdx.x;
    unsigned g = xl, pend = xl + gridDim.x;
    const unsigned xcd = blockIdx.x & 7u;
    const long long rest = nblk_all - 2ll * gridDim.x, xper = rest > 0 ? (rest + 7) / 8 : 0;
    const long long xbase = 2ll * gridDim.x + xcd * xper, xend = xbase + xper < nblk_all ? xbase + xper : nblk_all;
    auto grabbed = [&](unsigned v) -> unsigned {
        if (!xmode) return 2u * gridDim.x + v;
        const long long gg = xbase + v;
        return gg < xend ? (unsigned)gg : 0xffffffffu;
    };

    // ---- gather of block grp: stage-0 operand order, zero beyond the end of the signal (src/pffastconv.c:231-233)
    typedef vec4<float> F4;
    F4 raw[R];
    // (t: an opaque copy of the thread index per call - addresses hoisted out of the block loop spill; q0 .. q1: the rows of 1024 samples to
    //  request - a wavefront that issues all 16 loads at once sits at the issue of the last ones until the memory pipeline has taken them)
    auto gather = [&](unsigned grp, int t, int q0 = 0, int q1 = R) {
        long long ba = (long long)grp;
        if (ba >= nblk_all) ba = nblk_all - 1;
        int sg, bk;
        fc_split(ba, nblk, nsig, sg, bk);
        const float* src = x + (size_t)sg * xstride + (long)bk * step;
        const long avail = (long)inputLen - (long)bk * step;      // samples of this block that exist (wave-uniform)
        if (avail >= 2 * n) {                                     // every block but the last ones of a signal: no predicates
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (q < q0 || q >= q1) continue;
                const F4u q4 = *reinterpret_cast<const F4u*>(src + 4 * (t + q * (n / (2 * R))));  // 16 bytes, 4-byte aligned
                F4 r; r.x = q4.a; r.y = q4.b; r.z = q4.c; r.w = q4.d;
                raw[q] = r;
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (q < q0 || q >= q1) continue;
                const int e0 = 4 * (t + q * (n / (2 * R)));      // first of 4 consecutive samples
                F4 r;
                if (e0 + 3 < avail) {
                    const F4u q4 = *reinterpret_cast<const F4u*>(src + e0);
                    r.x = q4.a; r.y = q4.b; r.z = q4.c; r.w = q4.d;
                } else {
                    r.x = e0 < avail ? src[e0] : 0.f; r.y = e0 + 1 < avail ? src[e0 + 1] : 0.f;
                    r.z = e0 + 2 < avail ? src[e0 + 2] : 0.f; r.w = e0 + 3 < avail ? src[e0 + 3] : 0.f;
                }
                raw[q] = r;
            }
        }
    };
    gather(g, t);

    // ---- per-thread constants: one base twiddle per butterfly of stages 1 and 2 (fft_tiled.h TWMODE 3), W_N^t of the pair passes
    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::template load_tw_stage<1>(wf, t, twg);
    KB::template load_tw_stage<1>(wb, t, twg);
    if (dyn && t == 0) { s_next[0] = 0u; s_next[1] = 0u; }
    __syncthreads();

    // the 8-byte exchange reads (fft_tiled.h xread, the stages whose butterflies are not adjacent pairs) as single ds_read_b64:
    //   after stage 0 (both directions): butterfly j = t of the radix-32 stage, operand q at row j mod 16, column j div 16 + 16 q
    //   forward, after stage 1: butterflies jm<2>(t, u) of the symmetric radix-16 stage, operand q at j + 512 q
    constexpr int ROW0 = n / 16 + C::PAD0;
    static_assert(C::PADN == 0, "natural image without padding");
    const CX* rd0 = img + (t & 15) * ROW0 + (t >> 4);
    const CX* rd1a = img + KF::template jm<2>(t, 0);
    const CX* rd1b = img + KF::template jm<2>(t, 1);
    auto xread0 = [&](CX (&v)[E]) {
#ifdef PF_FIR32_NO_ASMRD
        KF::template xread<0>(v, t, img);
#else
        Fir32Rd<0, 32, 16 * 8, 0>::run(v, rd0);
#endif
    };
    auto xread1f = [&](CX (&v)[E]) {
#ifdef PF_FIR32_NO_ASMRD
        KF::template xread<1>(v, t, img);
#else
        Fir32Rd<0, 16, 512 * 8, 0>::run(v, rd1a);
        Fir32Rd<0, 16, 512 * 8, 16>::run(v, rd1b);
#endif
    };
    for (unsigned it = 0; (long long)g < nblk_all; ++it) {
        if (dyn && t == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = grabbed(atomicAdd(ctr + (xmode ? xcd : 0u), 1u));
        }
        int sig, blk;
        fc_split((long long)g, nblk, nsig, sig, blk);
        const long off = (long)blk * step;                        // first input / output sample of the block
        const int numOut = (blk == nblk - 1) ? lastOut : step;
        float* dst = y + (size_t)sig * ystride + off;
        CX v[E];
        int tl = t;                                               // re-derived per block: 48 hoisted 64-bit addresses would live in scratch
        asm volatile("" : "+v"(tl));
        PF_FSTAMP(0);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[q] = mk<T>(raw[q].x, raw[q].y);
            v[R + q] = mk<T>(raw[q].z, raw[q].w);
        }
        // ================= forward transform
        KF::template butterflies<0>(v, t, wf, twg);
        PF_FSTAMP(1);
        KF::template xwrite<0>(v, t, img);
        wg_sync_raw();                                            // (publishes s_next)
        const unsigned gn = __builtin_amdgcn_readfirstlane(dyn ? s_next[(it + 1) & 1] : g + gridDim.x);   // (wave-uniform: the block's offsets and bounds live in SGPRs)
        xread0(v); wg_sync_raw();
        PF_FSTAMP(2);
        KF::template butterflies<1>(v, t, wf, twg);
        PF_FSTAMP(3);
        KF::template xwrite<1>(v, t, img); wg_sync_raw();
        xread1f(v); wg_sync_raw();
        PF_FSTAMP(4);
        // the folded coefficients of this thread's first eight slot pairs: in flight during the last stage (the other eight follow below)
        F4 ab[32];
#pragma unroll
        for (int c = 0; c < 16; ++c) ab[c] = HP[c * WG + tl];
        KF::template butterflies<2>(v, t, wf, twg);
        PF_FSTAMP(5);
        // ================= lane 0: its registers into the slot order of the regular pairing (through LDS: no selects, no registers)
        if (wave0) {
            if (first) {
#pragma unroll
                for (int i = 0; i < 24; ++i) lds_st(pscr + i, v[i]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (first) {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = lds_ld(pscr + Fir32::perm(i));
            }
        }
        // ================= Z'[k] = A_k Z[k] + B_k conj Z[n - k]: real finalize, x Hf / Nfft and real preprocess folded per bin (fir32_coef_kernel);
        //                   slot d holds bin k, slot 31 - d its mirror n - k
#pragma unroll
        for (int c = 16; c < 32; ++c) ab[c] = HP[c * WG + tl];
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const CX zA = v[d], zB = v[31 - d];
            const F4 ca = ab[2 * d], cb = ab[2 * d + 1];
            // thread 0, slot 15: bin 0 = (DC, Nyquist) is its own mirror (slot 16 = bin n/2 has B = 0)
            const CX mA = d == 15 ? KF::sel(first, zA, zB) : zB;
            v[d] = fir32_fold(mk<T>(ca.x, ca.y), zA, mk<T>(ca.z, ca.w), mA);
            v[31 - d] = fir32_fold(mk<T>(cb.x, cb.y), zB, mk<T>(cb.z, cb.w), zA);
        }
        PF_FSTAMP(6);
        if constexpr (PREF == 1) gather(gn, tl);
        if constexpr (PREF == 2) gather(gn, tl, 0, 4);                         // the spectrum registers are free: the next block lands during the inverse
        if (wave0) {
            if (first) {
#pragma unroll
                for (int i = 0; i < 24; ++i) lds_st(pscr + Fir32::perm(i), v[i]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (first) {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = lds_ld(pscr + i);
            }
        }
        // ================= backward transform (its first-stage operands are in place)
        KB::template butterflies<0>(v, t, wb, twg);
        PF_FSTAMP(7);
        if constexpr (PREF == 2) gather(gn, tl, 4, 8);
        KB::template xwrite<0>(v, t, img); wg_sync_raw();
        xread0(v); wg_sync_raw();
        PF_FSTAMP(8);
        KB::template butterflies<1>(v, t, wb, twg);
        PF_FSTAMP(9);
        if constexpr (PREF == 2) gather(gn, tl, 8, 12);
        KB::template xwrite<1>(v, t, img); wg_sync_raw();
        KB::template xread<1>(v, t, img); wg_sync_raw();
        PF_FSTAMP(10);
        if constexpr (PREF == 2) gather(gn, tl, 12, 16);
        KB::template butterflies<2>(v, t, wb, twg);
        PF_FSTAMP(11);
        // ================= the first numOut samples of the block (src/pffastconv.c:255)
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const int e0 = 4 * (tl + d * (n / (2 * R)));
            const CX a = v[d], b = v[R + d];
            const int lim = numOut - 4 * d * (n / (2 * R));       // samples of this row of 1024 that are output (wave-uniform)
            if (lim >= 4 * (n / (2 * R))) {
#ifndef PF_FIR32_NO_NT
                typedef float F4nt __attribute__((ext_vector_type(4), aligned(4)));
                F4nt q4; q4.x = a.x; q4.y = a.y; q4.z = b.x; q4.w = b.y;
                __builtin_nontemporal_store(q4, reinterpret_cast<F4nt*>(dst + e0));   // (outputs are written once and not read back: streaming)
#else
                F4u q4; q4.a = a.x; q4.b = a.y; q4.c = b.x; q4.d = b.y;
                *reinterpret_cast<F4u*>(dst + e0) = q4;
#endif
            } else if (lim > 0) {
                if (e0 + 3 < numOut) {
                    F4u q4; q4.a = a.x; q4.b = a.y; q4.c = b.x; q4.d = b.y;
                    *reinterpret_cast<F4u*>(dst + e0) = q4;
                } else {
                    if (e0 < numOut) dst[e0] = a.x;
                    if (e0 + 1 < numOut) dst[e0 + 1] = a.y;
                    if (e0 + 2 < numOut) dst[e0 + 2] = b.x;
                }
            }
        }
        PF_FSTAMP(12);
        if constexpr (!PREF) { if ((long long)gn < nblk_all) gather(gn, tl); }
        g = gn;
    }
    if (dyn && t == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[xmode ? 8 : 1], 1u);
        if (d == gridDim.x - 1) {
            if (xmode) { for (int i = 0; i < 9; ++i) atomicExch(&ctr[i], 0u); }
            else { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
        }
    }
}

}  // namespace pf
