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
//   * the filter spectrum of a thread's 32 bins is NOT resident (64 VGPRs): it is read per block from a thread-major copy of the
//     table (fir32_coef_kernel: 16 coalesced 16-byte loads per thread, L2 hits) right before the last forward stage and is dead after
//     the product;
//   * the 16 pair-pass twiddles W_N^(t + d n/16) are one base W_N^t times the constants W_32^d (30 VGPRs);
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
    // W_32^d, rounded from double
    static constexpr float W32[16][2] = {
        {1.0f, -0.0f}, {0.9807852506637573f, -0.19509032368659973f}, {0.9238795042037964f, -0.3826834261417389f},
        {0.8314695954322815f, -0.5555702447891235f}, {0.7071067690849304f, -0.7071067690849304f}, {0.5555702447891235f, -0.8314695954322815f},
        {0.3826834261417389f, -0.9238795042037964f}, {0.19509032368659973f, -0.9807852506637573f}, {0.0f, -1.0f},
        {-0.19509032368659973f, -0.9807852506637573f}, {-0.3826834261417389f, -0.9238795042037964f}, {-0.5555702447891235f, -0.8314695954322815f},
        {-0.7071067690849304f, -0.7071067690849304f}, {-0.8314695954322815f, -0.5555702447891235f}, {-0.9238795042037964f, -0.3826834261417389f},
        {-0.9807852506637573f, -0.19509032368659973f}};
};

// Thread-major copy of the filter spectrum (canonical half-complex spectrum x 1 / Nfft, bin 0 = (DC, Nyquist)), once per filter like
// the reference's own transform of the filter (src/pffastconv.c:108): HP[(c WG + t) ] = (H[bin(t, 2c)], H[bin(t, 2c + 1)]).
__global__ void __launch_bounds__(Fir32::WG) fir32_coef_kernel(const cx<float>* __restrict__ Hc, vec4<float>* __restrict__ HP) {
    const int t = threadIdx.x;
    for (int c = 0; c < 16; ++c) {
        // x 1/2: the real finalize X[k] = ((A + conj B) -+ i W (A - conj B)) / 2 leaves its factor here (not for thread 0's self-mirrored
        // bins 0 and n/2, slots 15 and 16, which take no such step)
        const float ha = (t == 0 && 2 * c == 16) ? 1.f : 0.5f, hb = (t == 0 && 2 * c + 1 == 15) ? 1.f : 0.5f;
        const cx<float> a = Hc[Fir32::bin(t, 2 * c)] * ha, b = Hc[Fir32::bin(t, 2 * c + 1)] * hb;
        vec4<float> o; o.x = a.x; o.y = a.y; o.z = b.x; o.w = b.y;
        HP[c * Fir32::WG + t] = o;
    }
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
    // W_N^k of slot d (N = 2 n): every thread but 0: k = t + d n/16, i.e. W_N^t W_32^d.  Thread 0 (permuted slots): d < 8: k = (2 d + 1) n/32,
    // i.e. W_64 W_32^d; 8 <= d < 15: k = (d - 7) n/16, i.e. W_32^(-7) W_32^d (slot 15 is the self-mirrored pair, no twiddle): the same
    // constants W_32^d on two bases, no select per slot
    const CX pb = twrg[t];                                        // W_N^t (t <= n/2)
    const CX pbase_lo = KF::sel(first, mk<T>(0.9951847195625305f, -0.0980171412229538f), pb);
    const CX pbase_hi = KF::sel(first, mk<T>(0.19509032368659973f, 0.9807852506637573f), pb);
    auto pair_tw = [&](int d) -> CX {
        const CX c = mk<T>(Fir32::W32[d][0], Fir32::W32[d][1]);
        return d == 0 ? pbase_lo : cmul(d < 8 ? pbase_lo : pbase_hi, c);
    };
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
        // the filter spectrum of this thread's bins: in flight during the last stage and the pair pass
        F4 hh[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) hh[c] = HP[c * WG + tl];
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
        // ================= packed spectrum -> X[k] (real finalize), x Hf / Nfft, X'[k] -> packed spectrum of the inverse (real preprocess):
        //                   slot d holds bin k, slot 31 - d its mirror n - k
#pragma unroll
        for (int d = 0; d < R; ++d) {
            const CX w = pair_tw(d);
            const CX A = v[d], B = v[31 - d];
            // real finalize without its factor 1/2 (folded into HP): X[k] = S + D, X[n - k] = conj(S - D), S = A + conj B, D = -i W (A - conj B)
            typename KF::Pair f;
            {
                const CX su = add_conj(A, B), m = cmul(sub_conj(A, B), w);
                f.a = add_rot<FWD>(su, m);
                f.b = conj(sub_rot<FWD>(su, m));
            }
            const F4 ha4 = hh[d >> 1], hb4 = hh[(31 - d) >> 1];
            const CX ha = (d & 1) ? mk<T>(ha4.z, ha4.w) : mk<T>(ha4.x, ha4.y);
            const CX hb = ((31 - d) & 1) ? mk<T>(hb4.z, hb4.w) : mk<T>(hb4.x, hb4.y);
            CX xa = cmul(f.a, ha), xb = cmul(f.b, hb);
            typename KB::Pair r = KB::pair1(xa, xb, w);
            if (d == 15) {
                // thread 0: slot 15 = bin 0 = (DC, Nyquist) packed in one complex (two real products, src/pffft_priv_impl.h:1680-1683),
                // slot 16 = bin n/2 (its own mirror: X = conj Z, Z' = 2 conj X')
                const CX x0 = mk<T>(A.x + A.y, A.x - A.y);
                const CX p0 = mk<T>(x0.x * ha.x, x0.y * ha.y);
                const CX z0 = mk<T>(p0.x + p0.y, p0.x - p0.y);
                const CX xh = cmul(conj(B), hb);
                const CX zh = mk<T>((T)2 * xh.x, (T)-2 * xh.y);
                r.a = KF::sel(first, z0, r.a);
                r.b = KF::sel(first, zh, r.b);
            }
            v[d] = r.a; v[31 - d] = r.b;
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
