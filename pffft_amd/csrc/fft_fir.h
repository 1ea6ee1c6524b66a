// Fused overlap-save FIR block kernel (pffastconv): for every block of the signal
//     gather (zero padded) -> real FFT -> x Hf * 1/Nfft -> inverse real FFT -> scatter of the valid samples
// in ONE kernel, one HBM read of the block and one write of its valid outputs.
//
// Reference: the block loop of pffastconv_apply, src/pffastconv.c:207-261 (copy + zero pad :231-233,
// pffft_transform forward :235, pffft_zconvolve_no_accu :238, pffft_transform backward :254, copy of numOut
// samples :255), which walks the blocks one at a time through three separate sweeps over Nfft floats.
//
// Built from the register-tiled transform (fft_tiled.h) with n = Nfft/2 complex points.  Because the
// spectrum-side stage of BOTH directions uses the symmetric butterfly assignment, a thread that finishes
// the forward transform holds exactly the bins (k and n-k pairs) whose products the first backward
// butterflies need: forward pair pass, multiplication with the filter spectrum and backward pair pass all
// happen in registers — no layout, no LDS exchange between the two transforms.  The filter spectrum
// (canonical order, pre-scaled by 1/Nfft) sits in registers for the whole persistent loop.
#pragma once
#include "fft_tiled.h"

namespace pf {

#ifdef PF_FIR_DEBUG
__device__ long long pf_dbg[64];
#define PF_STAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) pf_dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PF_STAMP(i) do { } while (0)
#endif

// block index over all signals of a call -> (signal, block in the signal).  One signal (the reference's entry) needs no division;
// otherwise 32-bit whenever the index fits: the 64-bit division is a ~200-instruction loop, and the LDS-DMA kernel ran it once PER
// PIECE (tools/dma_timeline.hip: 350 cycles per piece, 3 300 of the 24 000 cycles of an iteration)
__device__ __forceinline__ void fc_split(long long ba, int nblk, int nsig, int& sig, int& blk) {
    if (nsig == 1) { sig = 0; blk = (int)ba; return; }
    if (ba <= 0xffffffffll) {
        const unsigned s = (unsigned)ba / (unsigned)nblk;
        sig = (int)s; blk = (int)((unsigned)ba - s * (unsigned)nblk);
        return;
    }
    sig = (int)(ba / nblk); blk = (int)(ba - (long long)sig * nblk);
}

// Position of workgroup b in a sweep of `grid` consecutive blocks such that every XCD takes a CONTIGUOUS run of the sweep: workgroup b
// runs on XCD b mod 8 (observed placement, MI355X_MICROARCH.md - used for speed only, any placement is correct), XCD x takes positions
// [x base + min(x, r), ...) with base = grid / 8, r = grid mod 8 - a bijection on [0, grid).  Adjacent overlap-save blocks share
// taps - 1 input samples; taken by workgroups of different XCDs those are fetched through two L2s (rocprofv3, round 4: 1.13 x the
// algorithmic bytes on the long signals, 1.59 x on the stated C4 call), inside one XCD the second reader hits.
__device__ __forceinline__ unsigned xcd_local(unsigned b, unsigned grid) {
    const unsigned base = grid >> 3, r = grid & 7u, x = b & 7u, q = b >> 3;
    return x * base + (x < r ? x : r) + q;
}

struct __attribute__((packed, aligned(4))) F4u { float a, b, c, d; };  // block offsets are multiples of 4 bytes only

template <class C>
__global__ void __launch_bounds__(C::WG_THREADS, 2)
fastconv_fused_kernel(const float* __restrict__ x, float* __restrict__ y, const cx<float>* __restrict__ Hc,
                      int nblk, int step, int inputLen, int lastOut,
                      const cx<float>* __restrict__ twg, const cx<float>* __restrict__ twrg, unsigned* ctr,
                      int nsig, size_t xstride, size_t ystride) {
    // nsig independent signals (pffastconv_hip_apply_batch): block index = sig * nblk + block-in-signal
    typedef float T;
    typedef cx<T> CX;
    typedef Tiled<C, FWD, 1> KF;
    typedef Tiled<C, BWD, 1> KB;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, NS = C::NS;
    constexpr int R0 = C::rad(0), RL = C::rad(NS - 1);
    static_assert(R0 == RL && E / R0 == 2 && C::VEC == 2, "fused FIR needs R0 == RL, two butterflies per thread, float");
    constexpr int Nfft = 2 * n;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* img = reinterpret_cast<CX*>(smem_raw + C::TABLE_BYTES) + (size_t)slot * C::IMG;
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + C::TABLE_BYTES + (size_t)C::T_PER_WG * C::IMG * sizeof(CX));

    PF_STAMP(0);
    // ctr == nullptr: static assignment (grid covers every block group once) — no atomics on the latency path
    const bool dyn = ctr != nullptr;
    unsigned pend = 0;
    unsigned g = blockIdx.x;
    // (the first two groups of a workgroup are static - its index, and that plus the grid -, the counter hands out what follows: fft_tiled.h)
    pend = blockIdx.x + gridDim.x;
    const long long nblk_all = (long long)nblk * nsig;
    // ---- gather of block group grp: stage-0 operand order, zero beyond the end of the signal (src/pffastconv.c:231-233).
    //      The FIRST block's samples are requested before anything else: the twiddle / filter tables below are cold on a call
    //      with few blocks, and their misses then overlap the block's instead of preceding it (tools/fir_timeline.hip: the
    //      prologue cost 5 000-6 000 of the 21 000 cycles of the stated C4 call before the first sample was even requested)
    typedef vec4<float> F4;
    F4 raw[R0];
    auto gather = [&](unsigned grp) {
        const long long ba = (long long)grp * C::T_PER_WG + slot;
        int sg = nsig - 1, bk = nblk - 1;
        if (ba < nblk_all) fc_split(ba, nblk, nsig, sg, bk);
        const float* src = x + (size_t)sg * xstride + (long)bk * step;
        const long avail = (long)inputLen - (long)bk * step;  // samples of this block that exist
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const int e0 = 4 * (t + q * (n / (2 * R0)));      // first of 4 consecutive samples
            F4 r;
            if (e0 + 3 < avail) {
                const F4u q4 = *reinterpret_cast<const F4u*>(src + e0);  // 16 bytes, 4-byte aligned
                r.x = q4.a; r.y = q4.b; r.z = q4.c; r.w = q4.d;
            } else {
                r.x = e0 < avail ? src[e0] : 0.f; r.y = e0 + 1 < avail ? src[e0 + 1] : 0.f;
                r.z = e0 + 2 < avail ? src[e0 + 2] : 0.f; r.w = e0 + 3 < avail ? src[e0 + 3] : 0.f;
            }
            raw[q] = r;
        }
    };
    gather(g);
    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, t, twg, twrg);
    if constexpr (KB::REGTW) KB::template load_tw_stage<1>(wb, t, twg);   // (the pair twiddles W_N^k are those of the forward transform: one set, wf.p)
    // filter spectrum of the bins this thread owns after the forward transform: k = jm(t,u) + d n/R
    CX h[E];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < RL; ++d) h[u * RL + d] = Hc[KF::template jm<NS - 1>(t, u) + d * (n / RL)];
    for (unsigned it = 0; (long long)g * C::T_PER_WG < nblk_all; ++it) {
        if (dyn && threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        const long long blk_all = (long long)g * C::T_PER_WG + slot;
        const bool active = blk_all < nblk_all;
        int sig = nsig - 1, blk = nblk - 1;
        if (active) fc_split(blk_all, nblk, nsig, sig, blk);
        const long off = (long)blk * step;  // first input / output sample of the block
        const int numOut = (active && blk == nblk - 1) ? lastOut : step;
        float* ys = y + (size_t)sig * ystride;
        CX v[E];
        PF_STAMP(1);
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            v[q] = mk<T>(raw[q].x, raw[q].y);
            v[R0 + q] = mk<T>(raw[q].z, raw[q].w);
        }
        // ---- forward transform ----
        KF::template butterflies<0>(v, t, wf, twg);
        PF_STAMP(2);
        KF::template xwrite<0>(v, t, img);
        __syncthreads();
        PF_STAMP(3);
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        KF::template xread<0>(v, t, img); KF::xsync(); KF::template butterflies<1>(v, t, wf, twg);
        if constexpr (NS > 2) { KF::template xwrite<1>(v, t, img); KF::xsync(); KF::template xread<1>(v, t, img); KF::xsync(); KF::template butterflies<2>(v, t, wf, twg); }
        if constexpr (NS > 3) { KF::template xwrite<2>(v, t, img); KF::xsync(); KF::template xread<2>(v, t, img); KF::xsync(); KF::template butterflies<3>(v, t, wf, twg); }
        if constexpr (NS > 4) { KF::template xwrite<3>(v, t, img); KF::xsync(); KF::template xread<3>(v, t, img); KF::xsync(); KF::template butterflies<4>(v, t, wf, twg); }
        PF_STAMP(4);
        KF::pair_regs(v, t, wf);                       // packed spectrum -> half-complex spectrum X[k]
        // ---- X[k] * H[k] (already scaled by 1/Nfft, src/pffastconv.c:97,238); bin 0 carries (DC, Nyquist),
        //      two real products (src/pffft_priv_impl.h:1680-1683) ----
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const CX p = cmul(v[i], h[i]);
            if (i == 0) {
                const CX r = mk<T>(v[0].x * h[0].x, v[0].y * h[0].y);
                v[0] = KF::sel(t == 0, r, p);
            } else {
                v[i] = p;
            }
        }
        PF_STAMP(5);
        KB::pair_regs_p(v, t, wf.p);                   // half-complex spectrum -> packed spectrum of the inverse
        PF_STAMP(6);
        // ---- backward transform (first-stage operands are already in place) ----
        KB::template butterflies<0>(v, t, wb, twg);
        KB::template xwrite<0>(v, t, img); KB::xsync();
        KB::template xread<0>(v, t, img); KB::xsync(); KB::template butterflies<1>(v, t, wb, twg);
        if constexpr (NS > 2) { KB::template xwrite<1>(v, t, img); KB::xsync(); KB::template xread<1>(v, t, img); KB::xsync(); KB::template butterflies<2>(v, t, wb, twg); }
        if constexpr (NS > 3) { KB::template xwrite<2>(v, t, img); KB::xsync(); KB::template xread<2>(v, t, img); KB::xsync(); KB::template butterflies<3>(v, t, wb, twg); }
        if constexpr (NS > 4) { KB::template xwrite<3>(v, t, img); KB::xsync(); KB::template xread<3>(v, t, img); KB::xsync(); KB::template butterflies<4>(v, t, wb, twg); }
        PF_STAMP(7);
        // ---- scatter the first numOut samples (src/pffastconv.c:255) ----
        if (active) {
            float* dst = ys + off;
#pragma unroll
            for (int ii = 0; ii < 1; ++ii)
#pragma unroll
                for (int d = 0; d < RL; ++d) {
                    const int e0 = 4 * (t + TPT * ii + d * (n / (2 * RL)));
                    const CX a = v[(2 * ii) * RL + d], b = v[(2 * ii + 1) * RL + d];
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
        KB::xsync();
        PF_STAMP(8);
        g = gn;
        if ((long long)g * C::T_PER_WG < nblk_all) gather(g);
    }
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
    (void)Nfft;
}

// scale a canonical half-complex spectrum (n bins) into the table the fused kernel multiplies with
static __global__ void fastconv_scale_kernel(const float* __restrict__ in, float* __restrict__ out, int count, float s) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) out[i] = in[i] * s;
}

// FIR configurations: WG = one block (TPT threads) so that a call with few blocks still spreads over the chip
// (The three-stage / 32-points-per-thread configurations that won for the plain transforms, fft_tiled.h TiledAltF32b,
//  were tried here too: with the filter spectrum and both twiddle sets resident they spill ~890 B per lane and the
//  C4 call goes from 11.4 us to 48 us.  The four-stage ones stay.)
struct FirCfg {
    typedef TiledCfg<float, 9, 32, 3, 8, 8, 8, 1, 4, 4, 3, 0, 64, 2> C512;
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 0, 64, 2> C1024;
    typedef TiledCfg<float, 11, 128, 4, 8, 4, 8, 8, 4, 8, 3, 0, 128, 2> C2048;
    typedef TiledCfg<float, 12, 256, 4, 8, 8, 8, 8, 4, 8, 3, 0, 256, 2> C4096;
    typedef TiledCfg<float, 13, 512, 4, 8, 8, 16, 8, 4, 8, 3, 0, 512, 2> C8192;
    // round 4, calls with few blocks (the stated C4 call: 255 blocks of 8192 samples on 256 CUs): twice the threads per block,
    // eight points per thread - two wavefronts per SIMD and half the dependent chain per thread, one more exchange per transform
    typedef TiledCfg<float, 12, 512, 5, 4, 8, 8, 4, 8, 4, 3, 0, 512, 2, 4> C4096m;
    typedef TiledCfg<float, 11, 256, 5, 4, 8, 4, 4, 8, 4, 3, 0, 256, 2, 4> C2048m;
};

}  // namespace pf

namespace pf {

// ------------------------------------------------------------------------------------------------------------------
// Uniformly partitioned overlap-save in ONE kernel (round 2) — long filters on long signals / many signals.
//
// Reference: pffastconv_apply's block loop (src/pffastconv.c:207-261) transforms Nfft = 2 next_pow2(taps - 1) samples per
// block; pffft_zconvolve_accumulate exists for exactly the partitioned form ("multiple convolutions ... accumulate",
// README.md:273-275).  What a caller observes is only how many samples a call produces (fc_schedule) and the values of the
// exact convolution — so the same outputs are computed here with the filter cut into P partitions of B = n taps:
//     y[kB + m] = sum_p IFFT( X_{k+p} . H_p )[m],  m < B,   X_s = FFT_2B( x[sB .. sB + 2B) ),  H_p = FFT_2B of partition p's
// correlation image.  Why: the 8192-point transforms of the long-block kernels run in workgroup-wide lock-step phases
// (~33 k cycles per block and CU whatever the filter), while a 2048-sample block is ONE WAVEFRONT's
// transform — wave-local exchanges, no workgroup barrier in the loop, wavefronts de-phase and overlap VALU with LDS like
// the headline kernel.  Each wavefront owns a run of consecutive output blocks of one signal:
//   * HBM traffic is the algorithmic 8 bytes per output sample: the first half of a block's input is the second half of
//     the previous block's and is kept in registers (the operand layout makes them the same thread's chunks), the new
//     half is prefetched one block ahead;
//   * the last P packed spectra live in registers (P x 16 bins per lane);
//   * the two pair passes and the product with H_p between the transforms are folded into two coefficients per bin and
//     partition (see the kernel), built once per workgroup in LDS in the threads' own bin order (conflict-free 8-byte
//     reads): ~1900 -> ~1150 VALU instructions per block for P = 1;
//   * per output block: one forward and one inverse 1024-point complex transform + P x 16 x 2 complex multiply-adds per lane.
// A run of K output blocks needs K + P - 1 forward transforms (the first P - 1 fill the ring).
template <class C, int P, int OCC = (P > 2 ? 1 : 2)>
__global__ void __launch_bounds__(C::WG_THREADS, OCC)
fastconv_part_kernel(const float* __restrict__ x, float* __restrict__ y, const cx<float>* __restrict__ Hp,
                     int nblk, int inputLen, int lastOut, int kchunk,
                     const cx<float>* __restrict__ twg, const cx<float>* __restrict__ twrg,
                     int nsig, size_t xstride, size_t ystride) {
    typedef float T;
    typedef cx<T> CX;
    typedef Tiled<C, FWD, 1> KF;
    typedef Tiled<C, BWD, 1> KB;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, NS = C::NS, WAVES = C::T_PER_WG;
    constexpr int R0 = C::rad(0), RL = C::rad(NS - 1), HALF = R0 / 2, RS = RL;
    constexpr int B = n;                                      // output samples per block, taps per partition
    static_assert(TPT == 64 && R0 == RL && E / R0 == 2 && C::VEC == 2, "one wavefront per block, two butterflies per thread, float");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* img = reinterpret_cast<CX*>(smem_raw) + (size_t)slot * C::IMG;
    CX* AB = reinterpret_cast<CX*>(smem_raw) + (size_t)WAVES * C::IMG;          // [P][E][2][64]: coefficients A, B per lane

    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, t, twg, twrg);
    KB::load_tw(wb, t, twg, twrg);
    // ---- coefficients.  Between the two transforms the reference does: packed spectrum Z -> half-complex X (real finalize,
    // src/pffft_priv_impl.h:1330-1372), X . H (zconvolve, :1632-1684), half-complex -> packed Z' (real preprocess, :1423-1462).
    // All three are linear in the mirror pair (Z[k], conj Z[n-k]) that one thread owns, so they fold into
    //     Z'[k]   = A  Z[k]   + B  conj Z[n-k]          A  = g Hk a + d conj(Hm) b      B  = g Hk b + d conj(Hm) a
    //     Z'[n-k] = A' Z[n-k] + B' conj Z[k]            A' = conj(d Hk b + g conj(Hm) a)   B' = conj(d Hk a + g conj(Hm) b)
    // with a = (1 - i w)/2, b = (1 + i w)/2, g = 1 + i conj(w), d = 1 - i conj(w), w = W_N^k, Hk = H[k], Hm = H[n-k];
    // bin 0 = (DC, Nyquist): A = Hx + Hy, B = i (Hx - Hy) on itself; bin n/2: A = 2 conj(H), B = 0.
    // Computed once per workgroup by its first wavefront (the lane -> bin map is the same in every wavefront).
    if (slot == 0) {
        auto bin_of_slot = [&](int i) { return KF::template jm<NS - 1>(t, i / RL) + (i % RL) * (n / RL); };
        auto put = [&](int p, int i, CX a, CX b) { AB[((p * E + i) * 2 + 0) * 64 + t] = a; AB[((p * E + i) * 2 + 1) * 64 + t] = b; };
        auto pair_coef = [&](int p, int ia, int ib, CX w) {       // slot ia holds bin k, slot ib its mirror n - k
            const CX Hk = Hp[(size_t)p * n + bin_of_slot(ia)], Hm = Hp[(size_t)p * n + bin_of_slot(ib)];
            const CX iw = mk<T>(-w.y, w.x), iwc = mk<T>(w.y, w.x);   // i w, i conj(w)
            const CX al = mk<T>(0.5f * (1.f - iw.x), -0.5f * iw.y), be = mk<T>(0.5f * (1.f + iw.x), 0.5f * iw.y);
            const CX ga = mk<T>(1.f + iwc.x, iwc.y), de = mk<T>(1.f - iwc.x, -iwc.y);
            const CX gH = cmul(ga, Hk), dHm = cmul(de, conj(Hm)), dH = cmul(de, Hk), gHm = cmul(ga, conj(Hm));
            put(p, ia, cmul(gH, al) + cmul(dHm, be), cmul(gH, be) + cmul(dHm, al));
            put(p, ib, conj(cmul(dH, be) + cmul(gHm, al)), conj(cmul(dH, al) + cmul(gHm, be)));
        };
#pragma unroll 1
        for (int p = 0; p < P; ++p) {
            if (t != 0) {
                for (int d = 0; d < RS; ++d) pair_coef(p, d, RS + (RS - 1 - d), wf.p[d]);
            } else {
                const CX H0 = Hp[(size_t)p * n], Hh = Hp[(size_t)p * n + n / 2];
                put(p, 0, mk<T>(H0.x + H0.y, 0.f), mk<T>(0.f, H0.x - H0.y));
                put(p, RS / 2, mk<T>(2.f * Hh.x, -2.f * Hh.y), mk<T>(0.f, 0.f));
                for (int d = 1; d < RS / 2; ++d) pair_coef(p, d, RS - d, wf.p[d]);
                for (int d = 0; d < RS / 2; ++d) pair_coef(p, RS + d, RS + (RS - 1 - d), wf.p[RS / 2 + d]);
            }
        }
    }
    __syncthreads();

    const long ntask_sig = (nblk + kchunk - 1) / kchunk;
    const long ntask = ntask_sig * nsig;
    const long gw = (long)blockIdx.x * WAVES + slot, nw = (long)gridDim.x * WAVES;
    typedef vec4<float> F4;
    // 4 consecutive samples of the signal starting at e (zero beyond the end: src/pffastconv.c:231-233)
    auto load4 = [&](const float* xs, long e) -> F4 {
        F4 r;
        if (e + 3 < inputLen) {
            const F4u q4 = *reinterpret_cast<const F4u*>(xs + e);
            r.x = q4.a; r.y = q4.b; r.z = q4.c; r.w = q4.d;
        } else {
            r.x = e < inputLen ? xs[e] : 0.f; r.y = e + 1 < inputLen ? xs[e + 1] : 0.f;
            r.z = e + 2 < inputLen ? xs[e + 2] : 0.f; r.w = e + 3 < inputLen ? xs[e + 3] : 0.f;
        }
        return r;
    };
    const bool first = t == 0;
    for (long task = gw; task < ntask; task += nw) {
        const int sig = (int)(task / ntask_sig);
        const int k0 = (int)(task - (long)sig * ntask_sig) * kchunk;
        const int k1 = k0 + kchunk < nblk ? k0 + kchunk : nblk;
        const float* xs = x + (size_t)sig * xstride;
        float* ys = y + (size_t)sig * ystride;
        CX ring[P][E];                                        // packed spectra Z of the last P blocks
        // chunk q of a block: samples 4 (t + q n / (2 R0)) .. + 3; chunks HALF .. R0-1 are the block's new half and become
        // chunks 0 .. HALF-1 of the next block
        F4 hi[HALF], nx[HALF];
#pragma unroll
        for (int q = 0; q < HALF; ++q) hi[q] = load4(xs, (long)k0 * B + 4 * (t + q * (n / (2 * R0))));
#pragma unroll
        for (int q = 0; q < HALF; ++q) nx[q] = load4(xs, (long)(k0 + 1) * B + 4 * (t + q * (n / (2 * R0))));
        for (int s = k0; s < k1 + P - 1; ++s) {
            CX v[E];
#pragma unroll
            for (int q = 0; q < HALF; ++q) {
                v[q] = mk<T>(hi[q].x, hi[q].y); v[R0 + q] = mk<T>(hi[q].z, hi[q].w);
                v[HALF + q] = mk<T>(nx[q].x, nx[q].y); v[R0 + HALF + q] = mk<T>(nx[q].z, nx[q].w);
                hi[q] = nx[q];
            }
#pragma unroll
            for (int q = 0; q < HALF; ++q) nx[q] = load4(xs, (long)(s + 2) * B + 4 * (t + q * (n / (2 * R0))));   // next block's new half
            // ---- forward transform of x[sB .. sB + 2B) -> packed spectrum Z_s (no pair pass: folded into A, B)
            KF::template butterflies<0>(v, t, wf, twg);
            KF::template xwrite<0>(v, t, img); KF::xsync();
            KF::template xread<0>(v, t, img); KF::xsync(); KF::template butterflies<1>(v, t, wf, twg);
            if constexpr (NS > 2) { KF::template xwrite<1>(v, t, img); KF::xsync(); KF::template xread<1>(v, t, img); KF::xsync(); KF::template butterflies<2>(v, t, wf, twg); }
            if constexpr (NS > 3) { KF::template xwrite<2>(v, t, img); KF::xsync(); KF::template xread<2>(v, t, img); KF::xsync(); KF::template butterflies<3>(v, t, wf, twg); }
#pragma unroll
            for (int p = 0; p + 1 < P; ++p)
#pragma unroll
                for (int i = 0; i < E; ++i) ring[p][i] = ring[p + 1][i];
#pragma unroll
            for (int i = 0; i < E; ++i) ring[P - 1][i] = v[i];
            if (s - k0 < P - 1) continue;                     // the ring is still filling
            // ---- Z'_k = sum_p (A_p Z_{k+p} + B_p conj(mirror of Z_{k+p})): the mirror of slot i sits in slot pi(i) of the same
            //      thread — RS + (RS-1-d) <-> d, except in thread 0, whose two butterflies are self-mirrored
            const int k = s - (P - 1);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                constexpr int dummy = 0; (void)dummy;
                const int u = i / RS, d = i % RS;
                const int pi1 = u == 0 ? RS + (RS - 1 - d) : (RS - 1 - d);                       // threads 1 .. 63
                const int pi0 = u == 0 ? (d == 0 || d == RS / 2 ? d : RS - d) : RS + (RS - 1 - d);   // thread 0
                CX acc = mk<T>(0.f, 0.f);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const CX a = AB[((p * E + i) * 2 + 0) * 64 + t], bq = AB[((p * E + i) * 2 + 1) * 64 + t];
                    const CX z = ring[p][i], zm = KF::sel(first, ring[p][pi0], ring[p][pi1]);
                    // acc += a z + bq conj(zm)
                    acc = mk<T>(fma_(a.x, z.x, fma_(-a.y, z.y, fma_(bq.x, zm.x, fma_(bq.y, zm.y, acc.x)))),
                                fma_(a.x, z.y, fma_(a.y, z.x, fma_(bq.y, zm.x, fma_(-bq.x, zm.y, acc.y)))));
                }
                v[i] = acc;
            }
            // ---- inverse transform; its first B samples are the block's outputs (src/pffastconv.c:255)
            KB::template butterflies<0>(v, t, wb, twg);
            KB::template xwrite<0>(v, t, img); KB::xsync();
            KB::template xread<0>(v, t, img); KB::xsync(); KB::template butterflies<1>(v, t, wb, twg);
            if constexpr (NS > 2) { KB::template xwrite<1>(v, t, img); KB::xsync(); KB::template xread<1>(v, t, img); KB::xsync(); KB::template butterflies<2>(v, t, wb, twg); }
            if constexpr (NS > 3) { KB::template xwrite<2>(v, t, img); KB::xsync(); KB::template xread<2>(v, t, img); KB::xsync(); KB::template butterflies<3>(v, t, wb, twg); }
            const int numOut = (k == nblk - 1) ? lastOut : B;
            float* dst = ys + (size_t)k * B;
#pragma unroll
            for (int d = 0; d < HALF; ++d) {
                const int e0 = 4 * (t + d * (n / (2 * RL)));
                const CX a = v[d], bb = v[RL + d];
                if (e0 + 3 < numOut) {
                    F4u q4; q4.a = a.x; q4.b = a.y; q4.c = bb.x; q4.d = bb.y;
                    *reinterpret_cast<F4u*>(dst + e0) = q4;
                } else {
                    if (e0 < numOut) dst[e0] = a.x;
                    if (e0 + 1 < numOut) dst[e0 + 1] = a.y;
                    if (e0 + 2 < numOut) dst[e0 + 2] = bb.x;
                }
            }
            KB::xsync();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// One wavefront per overlap-save block with the block step the FILTER allows (round 3).  fastconv_part_kernel advances by
// B = Nfft/2 samples per block whatever the filter: the partitioned form needs that, but a filter of 200 taps wastes 45 % of
// every inverse transform (1024 of its 1849 valid samples are kept).  Here P = 1 and step = Nfft - taps + 1 (rounded down to
// a multiple of 4 for the 16-byte stores): src/pffastconv.c:204-261 with the reference's own arithmetic of valid samples,
// on 2048-sample blocks whatever Nfft the reference would have chosen - what a caller observes is the number of samples
// produced (fc_schedule) and the values of the exact convolution.  Same transform, same folded coefficients
//     Z'[k] = A Z[k] + B conj Z[n-k]                      (fastconv_part_kernel above; one partition)
// the whole block is loaded (the overlap of taps - 1 samples is re-read: L2 / HBM traffic Nfft / step of the ideal, the
// kernel is VALU-bound), one block ahead.
template <class C, int OCC = 3>
__global__ void __launch_bounds__(C::WG_THREADS, OCC)
fastconv_wave_kernel(const float* __restrict__ x, float* __restrict__ y, const cx<float>* __restrict__ Hp,
                     int nblk, int step, int inputLen, int lastOut, int kchunk,
                     const cx<float>* __restrict__ twg, const cx<float>* __restrict__ twrg,
                     int nsig, size_t xstride, size_t ystride) {
    typedef float T;
    typedef cx<T> CX;
    typedef Tiled<C, FWD, 1> KF;
    typedef Tiled<C, BWD, 1> KB;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, NS = C::NS, WAVES = C::T_PER_WG;
    constexpr int R0 = C::rad(0), RL = C::rad(NS - 1), RS = RL;
    static_assert(TPT == 64 && R0 == RL && E / R0 == 2 && C::VEC == 2, "one wavefront per block, two butterflies per thread, float");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    CX* img = reinterpret_cast<CX*>(smem_raw) + (size_t)slot * C::IMG;
    CX* AB = reinterpret_cast<CX*>(smem_raw) + (size_t)WAVES * C::IMG;          // [E][2][64]: coefficients A, B per lane
    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, t, twg, twrg);
    KB::load_tw(wb, t, twg, twrg);
    if (slot == 0) {   // the folded coefficients of fastconv_part_kernel, one partition
        auto bin_of_slot = [&](int i) { return KF::template jm<NS - 1>(t, i / RL) + (i % RL) * (n / RL); };
        auto put = [&](int i, CX a, CX b) { AB[(i * 2 + 0) * 64 + t] = a; AB[(i * 2 + 1) * 64 + t] = b; };
        auto pair_coef = [&](int ia, int ib, CX w) {
            const CX Hk = Hp[bin_of_slot(ia)], Hm = Hp[bin_of_slot(ib)];
            const CX iw = mk<T>(-w.y, w.x), iwc = mk<T>(w.y, w.x);
            const CX al = mk<T>(0.5f * (1.f - iw.x), -0.5f * iw.y), be = mk<T>(0.5f * (1.f + iw.x), 0.5f * iw.y);
            const CX ga = mk<T>(1.f + iwc.x, iwc.y), de = mk<T>(1.f - iwc.x, -iwc.y);
            const CX gH = cmul(ga, Hk), dHm = cmul(de, conj(Hm)), dH = cmul(de, Hk), gHm = cmul(ga, conj(Hm));
            put(ia, cmul(gH, al) + cmul(dHm, be), cmul(gH, be) + cmul(dHm, al));
            put(ib, conj(cmul(dH, be) + cmul(gHm, al)), conj(cmul(dH, al) + cmul(gHm, be)));
        };
        if (t != 0) {
            for (int d = 0; d < RS; ++d) pair_coef(d, RS + (RS - 1 - d), wf.p[d]);
        } else {
            const CX H0 = Hp[0], Hh = Hp[n / 2];
            put(0, mk<T>(H0.x + H0.y, 0.f), mk<T>(0.f, H0.x - H0.y));
            put(RS / 2, mk<T>(2.f * Hh.x, -2.f * Hh.y), mk<T>(0.f, 0.f));
            for (int d = 1; d < RS / 2; ++d) pair_coef(d, RS - d, wf.p[d]);
            for (int d = 0; d < RS / 2; ++d) pair_coef(RS + d, RS + (RS - 1 - d), wf.p[RS / 2 + d]);
        }
    }
    __syncthreads();

    const long ntask_sig = (nblk + kchunk - 1) / kchunk;
    const long ntask = ntask_sig * nsig;
    const long gw = (long)blockIdx.x * WAVES + slot, nw = (long)gridDim.x * WAVES;
    typedef vec4<float> F4;
    auto load4 = [&](const float* xs, long e) -> F4 {     // zero beyond the end of the signal: src/pffastconv.c:231-233
        F4 r;
        if (e + 3 < inputLen) {
            const F4u q4 = *reinterpret_cast<const F4u*>(xs + e);
            r.x = q4.a; r.y = q4.b; r.z = q4.c; r.w = q4.d;
        } else {
            r.x = e < inputLen ? xs[e] : 0.f; r.y = e + 1 < inputLen ? xs[e + 1] : 0.f;
            r.z = e + 2 < inputLen ? xs[e + 2] : 0.f; r.w = e + 3 < inputLen ? xs[e + 3] : 0.f;
        }
        return r;
    };
    const bool first = t == 0;
    for (long task = gw; task < ntask; task += nw) {
        const int sig = (int)(task / ntask_sig);
        const int k0 = (int)(task - (long)sig * ntask_sig) * kchunk;
        const int k1 = k0 + kchunk < nblk ? k0 + kchunk : nblk;
        const float* xs = x + (size_t)sig * xstride;
        float* ys = y + (size_t)sig * ystride;
        F4 nx[R0];                                            // chunk q of a block: samples 4 (t + q n / (2 R0)) .. + 3
#pragma unroll
        for (int q = 0; q < R0; ++q) nx[q] = load4(xs, (long)k0 * step + 4 * (t + q * (n / (2 * R0))));
        for (int k = k0; k < k1; ++k) {
            CX v[E];
#pragma unroll
            for (int q = 0; q < R0; ++q) { v[q] = mk<T>(nx[q].x, nx[q].y); v[R0 + q] = mk<T>(nx[q].z, nx[q].w); }
            if (k + 1 < k1) {
#pragma unroll
                for (int q = 0; q < R0; ++q) nx[q] = load4(xs, (long)(k + 1) * step + 4 * (t + q * (n / (2 * R0))));   // the next block flies
            }
            KF::template butterflies<0>(v, t, wf, twg);
            KF::template xwrite<0>(v, t, img); KF::xsync();
            KF::template xread<0>(v, t, img); KF::xsync(); KF::template butterflies<1>(v, t, wf, twg);
            if constexpr (NS > 2) { KF::template xwrite<1>(v, t, img); KF::xsync(); KF::template xread<1>(v, t, img); KF::xsync(); KF::template butterflies<2>(v, t, wf, twg); }
            if constexpr (NS > 3) { KF::template xwrite<2>(v, t, img); KF::xsync(); KF::template xread<2>(v, t, img); KF::xsync(); KF::template butterflies<3>(v, t, wf, twg); }
            // Z' = A Z + B conj(mirror of Z): the mirror of slot i sits in slot pi(i) of the same thread
            CX z[E];
#pragma unroll
            for (int i = 0; i < E; ++i) z[i] = v[i];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int u = i / RS, d = i % RS;
                const int pi1 = u == 0 ? RS + (RS - 1 - d) : (RS - 1 - d);                           // threads 1 .. 63
                const int pi0 = u == 0 ? (d == 0 || d == RS / 2 ? d : RS - d) : RS + (RS - 1 - d);   // thread 0
                const CX a = AB[(i * 2 + 0) * 64 + t], bq = AB[(i * 2 + 1) * 64 + t];
                const CX zz = z[i], zm = KF::sel(first, z[pi0], z[pi1]);
                v[i] = mk<T>(fma_(a.x, zz.x, fma_(-a.y, zz.y, fma_(bq.x, zm.x, bq.y * zm.y))),
                             fma_(a.x, zz.y, fma_(a.y, zz.x, fma_(bq.y, zm.x, -(bq.x * zm.y)))));
            }
            KB::template butterflies<0>(v, t, wb, twg);
            KB::template xwrite<0>(v, t, img); KB::xsync();
            KB::template xread<0>(v, t, img); KB::xsync(); KB::template butterflies<1>(v, t, wb, twg);
            if constexpr (NS > 2) { KB::template xwrite<1>(v, t, img); KB::xsync(); KB::template xread<1>(v, t, img); KB::xsync(); KB::template butterflies<2>(v, t, wb, twg); }
            if constexpr (NS > 3) { KB::template xwrite<2>(v, t, img); KB::xsync(); KB::template xread<2>(v, t, img); KB::xsync(); KB::template butterflies<3>(v, t, wb, twg); }
            const int numOut = (k == nblk - 1) ? lastOut : step;   // the first numOut samples are the block's outputs (:255)
            float* dst = ys + (size_t)k * step;
#pragma unroll
            for (int d = 0; d < RL; ++d) {
                const int e0 = 4 * (t + d * (n / (2 * RL)));
                const CX a = v[d], bb = v[RL + d];
                if (e0 + 3 < numOut) {
                    F4u q4; q4.a = a.x; q4.b = a.y; q4.c = bb.x; q4.d = bb.y;
                    *reinterpret_cast<F4u*>(dst + e0) = q4;
                } else {
                    if (e0 < numOut) dst[e0] = a.x;
                    if (e0 + 1 < numOut) dst[e0 + 1] = a.y;
                    if (e0 + 2 < numOut) dst[e0 + 2] = bb.x;
                }
            }
            KB::xsync();
        }
    }
}

struct FirPartCfg {
    // n = 1024 complex points = 2048-sample blocks, one wavefront per block, 4 wavefronts per workgroup
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 0, 256, 1> C1024;
};

}  // namespace pf
