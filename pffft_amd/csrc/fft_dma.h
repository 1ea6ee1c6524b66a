// LDS-DMA staged overlap-save block kernel (pffastconv, src/pffastconv.c:207-261) for the blocks that need a whole
// workgroup (Nfft/2 = 2048 ... 8192 complex points in float: 16 ... 64 KiB per block), built from the register-tiled
// transform of fft_tiled.h.  (Round 2 also had plain-transform kernels of this organisation; they measured 0.55-0.73
// against 0.68-0.80 for the register-staged ones and were removed in round 3 - DESIGN.md appendix.)
//
// ONE persistent 512-thread workgroup per CU owns TWO LDS images.  While image A is transformed in place (stage-0
// operands are read straight from it, the exchanges of the later stages reuse it), the next group of vectors lands in
// image B by `global_load_lds_dwordx4` — asynchronous global -> LDS copies that occupy no VGPR and need no ds_write
// pass.  The probe tools/dma_probe.hip measured this skeleton (64 KiB groups pulled in order from an atomic counter,
// counted `s_waitcnt vmcnt`) at 0.78-0.80 of the 8 TB/s roofline with up to four conflict-free LDS exchanges and ~600
// VALU instructions per thread per group fully hidden behind the DMA.
//
// Ordering rules of the DMA (MI355X_MICROARCH.md "Two waves per SIMD" item 7, cdna_hip_programming.md §LDS-DMA):
//   * a landed piece may be read only after the ISSUING wave's `s_waitcnt vmcnt` has retired it AND a barrier the reader
//     passed afterwards: top of every iteration = counted vmcnt -> lgkmcnt(0) -> s_barrier, the reads come after it;
//   * the pieces of group i+1 are issued after that same barrier, i.e. after every wave finished reading image B in
//     iteration i-1 (write-after-read), and are older than the global stores of iteration i, so the wait at the top of
//     iteration i+1 is `vmcnt(<stores per thread>)`: the stores stay in flight;
//   * barriers inside the iteration are raw `s_barrier` + `lgkmcnt(0)` (never a fence that drains vmcnt).
#pragma once
#include "fft_tiled.h"
#include "fft_fir.h"

namespace pf {

// one 1 KiB piece: lane L copies 16 bytes from gsrc to LDS byte address lds_dst + 16 L (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_sync_raw() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

#ifdef PF_DMA_DEBUG
__device__ long long pf_ddbg[64];
#define PF_DSTAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0 && it == 3) pf_ddbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PF_DSTAMP(i) do { } while (0)
#endif

template <class C> struct DmaGeom {
    typedef typename C::real_t T;
    static constexpr int VEC_BYTES = C::n * 2 * (int)sizeof(T);
    static constexpr int GROUP_BYTES = C::T_PER_WG * VEC_BYTES;       // contiguous in HBM
    static constexpr int PIECES = GROUP_BYTES / 1024, WAVES = C::WG_THREADS / 64, PPW = PIECES / WAVES;
    static constexpr int PPV = VEC_BYTES / 1024;                      // pieces per vector
    static constexpr int IMG_BYTES = C::IMG * 2 * (int)sizeof(T);
    static constexpr int BUF_BYTES = C::T_PER_WG * IMG_BYTES;
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES + 16;
    static constexpr int NSTORE = C::NCH;                             // 16-byte global stores per thread and iteration
    static_assert(PIECES % WAVES == 0 && PPW >= 1, "a group must split into whole 1 KiB pieces per wavefront");
    static_assert(VEC_BYTES % 1024 == 0, "a vector must be a whole number of 1 KiB pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "two images must fit LDS");
};

// ------------------------------------------------------------------------------------------------------------------
// Overlap-save FIR block kernel on the same skeleton (the register-staged one: fft_fir.h; reference: the block loop of
// pffastconv_apply, src/pffastconv.c:207-261).  A block's Nfft input floats land by DMA (block offsets are multiples
// of 4 bytes only: the copies are dword-aligned 16-byte transfers), forward real FFT, x Hf, inverse real FFT in
// registers / one LDS image, stores of the valid samples; the next group of blocks lands meanwhile.
// The store count per wave varies (ragged last chunk), so the landing wait is vmcnt(0) (measured equal to the counted
// wait on the copy skeleton, tools/dma_probe.hip).
// SPREAD: the pieces of the next group are issued one share per butterfly phase instead of all at the top of the iteration
// (tools/dma_timeline.hip: the eight back-to-back pieces per wave stall their wave for 3 300 cycles of a 24 000-cycle iteration -
// the vector-memory queue takes a piece per ~50 cycles and CU - and skew the waves into the next barrier by another 1 100)
template <class C, int SPREAD = 0>
__global__ void __launch_bounds__(C::WG_THREADS, 1)
fastconv_dma_kernel(const float* __restrict__ x, float* __restrict__ y, const cx<float>* __restrict__ Hc,
                    int nblk, int step, int inputLen, int lastOut,
                    const cx<float>* __restrict__ twg, const cx<float>* __restrict__ twrg, unsigned* ctr,
                    int nsig, size_t xstride, size_t ystride) {
    typedef float T;
    typedef cx<T> CX;
    typedef Tiled<C, FWD, 1> KF;
    typedef Tiled<C, BWD, 1> KB;
    typedef DmaGeom<C> G;
    constexpr int n = C::n, E = C::E, TPT = C::TPT, NS = C::NS;
    constexpr int R0 = C::rad(0), RL = C::rad(NS - 1);
    static_assert(R0 == RL && E / R0 == 2 && C::VEC == 2, "fused FIR needs R0 == RL, two butterflies per thread, float");
    static_assert(C::TWMODE == 3 || C::TWMODE == 0, "register twiddles only (no LDS table next to two images)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int slot = threadIdx.x / TPT, t = threadIdx.x % TPT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + 2 * (size_t)G::BUF_BYTES);

    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, t, twg, twrg);
    KB::load_tw(wb, t, twg, twrg);
    // Between the two transforms the reference does real finalize -> X . H -> real preprocess (src/pffft_priv_impl.h:1330-1372,
    // :1632-1684, :1423-1462).  All three are linear in the mirror pair (Z[k], conj Z[n-k]) this thread owns, so they fold
    // into two coefficients per bin (derivation: fft_fir.h fastconv_part_kernel):  Z'[k] = A Z[k] + B conj Z[n-k].
    constexpr int RS = RL;
    CX cA[E], cB[E];
    {
        auto bin_of_slot = [&](int i) { return KF::template jm<NS - 1>(t, i / RL) + (i % RL) * (n / RL); };
        auto pair_coef = [&](int ia, int ib, CX w) {              // slot ia holds bin k, slot ib its mirror n - k
            const CX Hk = Hc[bin_of_slot(ia)], Hm = Hc[bin_of_slot(ib)];
            const CX iw = mk<T>(-w.y, w.x), iwc = mk<T>(w.y, w.x);   // i w, i conj(w)
            const CX al = mk<T>(0.5f * (1.f - iw.x), -0.5f * iw.y), be = mk<T>(0.5f * (1.f + iw.x), 0.5f * iw.y);
            const CX ga = mk<T>(1.f + iwc.x, iwc.y), de = mk<T>(1.f - iwc.x, -iwc.y);
            const CX gH = cmul(ga, Hk), dHm = cmul(de, conj(Hm)), dH = cmul(de, Hk), gHm = cmul(ga, conj(Hm));
            cA[ia] = cmul(gH, al) + cmul(dHm, be); cB[ia] = cmul(gH, be) + cmul(dHm, al);
            cA[ib] = conj(cmul(dH, be) + cmul(gHm, al)); cB[ib] = conj(cmul(dH, al) + cmul(gHm, be));
        };
        // every thread evaluates the regular pairing; thread 0 (both butterflies self-mirrored) overrides its own
#pragma unroll
        for (int d = 0; d < RS; ++d) pair_coef(d, RS + (RS - 1 - d), wf.p[d]);
        if (t == 0) {
            const CX H0 = Hc[0], Hh = Hc[n / 2];
            cA[0] = mk<T>(H0.x + H0.y, 0.f); cB[0] = mk<T>(0.f, H0.x - H0.y);                 // (DC, Nyquist) on itself
            cA[RS / 2] = mk<T>(2.f * Hh.x, -2.f * Hh.y); cB[RS / 2] = mk<T>(0.f, 0.f);        // bin n/2
#pragma unroll
            for (int d = 1; d < RS / 2; ++d) pair_coef(d, RS - d, wf.p[d]);
#pragma unroll
            for (int d = 0; d < RS / 2; ++d) pair_coef(RS + d, RS + (RS - 1 - d), wf.p[RS / 2 + d]);
        }
    }
    const bool first = t == 0;

    const bool dyn = ctr != nullptr;
    unsigned pend = 0;
    unsigned g = blockIdx.x;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    const long long nblk_all = (long long)nblk * nsig;
    // the last 16 bytes a lane may read: copies that would run past the signal are clamped there (their samples are
    // replaced by the zero padding of src/pffastconv.c:231-233 when the operands are picked up)
    constexpr int NPH = 2 * NS;                                   // butterfly phases per iteration
    auto issue = [&](unsigned grp, int b, int lo = 0, int hi = G::PPW) {
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            const int p = wave + G::WAVES * i;
            const int sl = p / G::PPV, pv = p % G::PPV;
            long long ba = (long long)grp * C::T_PER_WG + sl;
            if (ba >= nblk_all) ba = nblk_all - 1;
            int sig, blk;
            fc_split(ba, nblk, nsig, sig, blk);
            long e = (long)blk * step + pv * 256 + lane * 4;     // first of this lane's 4 floats
            if (e > (long)inputLen - 4) e = inputLen >= 4 ? (long)inputLen - 4 : 0;
            glds16(x + (size_t)sig * xstride + e, lds0 + (unsigned)(b * G::BUF_BYTES + sl * G::IMG_BYTES + pv * 1024));
        }
    };
    issue(g, 0);
    int b = 0;
    for (unsigned it = 0; (long long)g * C::T_PER_WG < nblk_all; ++it) {
        if (dyn && threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        PF_DSTAMP(0);
        wait_vmcnt<0>();
        wg_sync_raw();
        PF_DSTAMP(1);
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        if constexpr (!SPREAD) issue(gn, b ^ 1);
        PF_DSTAMP(2);
        const long long blk_all = (long long)g * C::T_PER_WG + slot;
        const bool active = blk_all < nblk_all;
        int sig = nsig - 1, blk = nblk - 1;
        if (active) fc_split(blk_all, nblk, nsig, sig, blk);
        const long off = (long)blk * step;
        const int numOut = (active && blk == nblk - 1) ? lastOut : step;
        float* ys = y + (size_t)sig * ystride;
        CX* img = reinterpret_cast<CX*>(smem_raw + (size_t)b * G::BUF_BYTES) + (size_t)slot * C::IMG;
        const chunk16* land16 = reinterpret_cast<const chunk16*>(img);
        CX v[E];
        // ---- stage-0 operands from the landed block, zero beyond the end of the signal (src/pffastconv.c:231-233).
        //      A clamped copy (issue) holds x[inputLen-4 .. inputLen-1] in place of the lane's own 4 floats: lanes whose
        //      first float is past inputLen-4 rebuild their samples from it.
        {
            const long avail = (long)inputLen - off;  // samples of this block that exist
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int c = t + q * (n / (2 * R0));
                const int e0 = 4 * c;
                const chunk16 cc = land16[c];
                float f0 = cc.x, f1 = cc.y, f2 = cc.z, f3 = cc.w;
                if (e0 + 3 >= avail) {
                    // the copy of this chunk was clamped to the last 4 floats of the signal (issue): its element k is
                    // sample avail - 4 + k of the block, so sample e0 + k = element k + sh, zero padding beyond
                    const long sh = (long)e0 - (avail - 4);      // >= 1
                    f0 = sh == 1 ? cc.y : sh == 2 ? cc.z : sh == 3 ? cc.w : 0.f;
                    f1 = sh == 1 ? cc.z : sh == 2 ? cc.w : 0.f;
                    f2 = sh == 1 ? cc.w : 0.f;
                    f3 = 0.f;
                }
                v[q] = mk<T>(f0, f1);
                v[R0 + q] = mk<T>(f2, f3);
            }
        }
        // ---- forward transform ----
        PF_DSTAMP(3);
        KF::template butterflies<0>(v, t, wf, twg);
#define PF_DMA_SHARE(PH) if constexpr (SPREAD) issue(gn, b ^ 1, (PH) * G::PPW / NPH, ((PH) + 1) * G::PPW / NPH)
        PF_DMA_SHARE(0);
        PF_DSTAMP(4);
        wg_sync_raw();
        PF_DSTAMP(5);
#define PF_DMA_STAGE(K, W, S, B, PH)                                                                            \
        if constexpr (NS > S + 1) {                                                                              \
            K::template xwrite<S>(v, t, img); wg_sync_raw(); PF_DSTAMP(B); K::template xread<S>(v, t, img);      \
            wg_sync_raw(); PF_DSTAMP(B + 1); K::template butterflies<S + 1>(v, t, W, twg); PF_DMA_SHARE(PH + S + 1); PF_DSTAMP(B + 2); \
        }
        PF_DMA_STAGE(KF, wf, 0, 6, 0) PF_DMA_STAGE(KF, wf, 1, 9, 0) PF_DMA_STAGE(KF, wf, 2, 12, 0) PF_DMA_STAGE(KF, wf, 3, 15, 0)
        // ---- Z'[k] = A Z[k] + B conj Z[mirror]: the mirror of slot i sits in slot pi(i) of the same thread
        {
            CX z[E];
#pragma unroll
            for (int i = 0; i < E; ++i) z[i] = v[i];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int u = i / RS, d = i % RS;
                const int pi1 = u == 0 ? RS + (RS - 1 - d) : (RS - 1 - d);                           // threads != 0
                const int pi0 = u == 0 ? (d == 0 || d == RS / 2 ? d : RS - d) : RS + (RS - 1 - d);   // thread 0
                const CX zm = KF::sel(first, z[pi0], z[pi1]), a = cA[i], bq = cB[i], zz = z[i];
                v[i] = mk<T>(fma_(a.x, zz.x, fma_(-a.y, zz.y, fma_(bq.x, zm.x, bq.y * zm.y))),
                             fma_(a.x, zz.y, fma_(a.y, zz.x, fma_(bq.y, zm.x, -(bq.x * zm.y)))));
            }
        }
        // ---- backward transform (first-stage operands are already in place) ----
        PF_DSTAMP(20);
        KB::template butterflies<0>(v, t, wb, twg);
        PF_DMA_SHARE(NS);
        PF_DSTAMP(21);
        PF_DMA_STAGE(KB, wb, 0, 22, NS) PF_DMA_STAGE(KB, wb, 1, 25, NS) PF_DMA_STAGE(KB, wb, 2, 28, NS) PF_DMA_STAGE(KB, wb, 3, 31, NS)
#undef PF_DMA_STAGE
#undef PF_DMA_SHARE
        // ---- the first numOut samples (src/pffastconv.c:255) ----
        if (active) {
            float* dst = ys + off;
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
        }
        PF_DSTAMP(40);
        g = gn;
        b ^= 1;
    }
    wait_vmcnt<0>();
    if (dyn && threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ---- configurations: <T, log2 n, threads/transform, stages, R0..R3, PAD0, PADN, TWMODE, PREFETCH(unused), WG threads, OCC> ----
// 16 points per thread, 512-thread workgroups: n = 8192 one vector per workgroup iteration, 4096 two, 2048 four.
struct DmaCfgF32 {
    typedef TiledCfg<float, 13, 512, 4, 8, 8, 16, 8, 4, 8, 3, 0, 512, 1> D8192;
    typedef TiledCfg<float, 12, 256, 4, 8, 8, 8, 8, 4, 8, 3, 0, 512, 1> D4096;
    typedef TiledCfg<float, 11, 128, 4, 8, 4, 8, 8, 4, 8, 3, 0, 512, 1> D2048;
    // (round 4: 1024 threads per 16384-sample block, eight points per thread, five stages - four wavefronts per SIMD and half the
    //  per-thread chain - measured 0.22 / 0.26 against 0.26 / 0.32 for D8192: one more exchange per transform)
};

}  // namespace pf
