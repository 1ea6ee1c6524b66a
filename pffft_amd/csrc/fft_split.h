// Overlap-save FIR block kernel with WAVE-LOCAL middle stages (round 4) - the throughput regime of pffastconv for filters
// beyond 1024 taps (BASELINE configs[3]; reference: the block loop of pffastconv_apply, src/pffastconv.c:207-261, on
// 16384-sample internal blocks: what a caller observes is the number of samples produced and the values of the convolution).
//
// Why: tools/dma_timeline.hip put numbers on fastconv_dma_kernel (fft_dma.h) - of the 22 400 cycles a block costs its CU,
// the butterflies of both 8192-point transforms take 5 600; the SIX workgroup-wide LDS exchanges (write, barrier, read, barrier:
// 13 barriers per block) take 11 500, and because all eight wavefronts move through every phase in lock step nothing overlaps
// (LDS stores run at 79 B/clk and CU, MI355X_MICROARCH.md LDS table: an exchange of a 64 KiB image cannot cost less than 1 100).
// Here n = 8 x 1024, decimation in frequency for the forward transform and its mirror image for the inverse:
//   A    all 8 wavefronts: radix-8 butterflies over z[j + 1024 q] straight from the landed block, times W_n^(j d), into row d
//                                                                                                        -> barrier
//   B    wavefront d: 1024-point transform of row d (8 x 16 x 8, two WAVE-LOCAL exchanges, no workgroup barrier): Z[d + 8 k2]
//   M    Z into row d in natural order                                                                   -> barrier
//        every thread fetches the mirrors conj Z[n - k] of its 16 bins (row 8 - d, index 1023 - k2)        -> barrier
//        Z'[k] = A Z[k] + B conj Z[n - k]   (real finalize, x Hf / Nfft, real preprocess folded per bin: fft_fir.h)
//   B'   wavefront d: inverse 1024-point transform of Z'[d + 8 k2], wave-local, into row d               -> barrier
//   A'   radix-8 across the rows, times conj W_n^(j d): z'[j + 1024 q] = the block's output samples, 16-byte stores
// Five workgroup barriers per block instead of thirteen, and during B, B' - two thirds of the arithmetic and four of the seven
// exchanges - every wavefront runs on its own: the LDS traffic of one overlaps the butterflies of another.
// The block lands by LDS-DMA (fft_dma.h glds16) in ONE buffer: stage A consumes it at the top of the iteration, the next block's
// pieces are issued right behind A's barrier and have the rest of the iteration to land.
#pragma once
#include "fft_dma.h"

namespace pf {

// W = 8: 16384-sample blocks, one 512-thread workgroup per CU;  W = 4: 8192-sample blocks on 256 threads, TWO workgroups per CU
// (67 KiB of LDS each) - two barrier domains, one's waits covered by the other's work - for the filters the shorter block still
// serves at 75 % overlap-save efficiency (<= 2048 taps)
template <int W> struct SplitFirT {
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 0, 512, 1> Sub;   // the wave-local 1024-point transform
    static constexpr int WAVES = W, M = 1024, n = W * M, WG = 64 * W;
    static constexpr int CPT = 8 / W;                                          // 16-byte chunks (two adjacent j) per thread and row
    static constexpr int ROW = (Sub::IMG_NAT > Sub::IMG_TRN ? Sub::IMG_NAT : Sub::IMG_TRN) + 8;   // points per wavefront region
    static constexpr int LAND_BYTES = n * 8;                                   // the landed block: 2 n floats, linear
    static constexpr size_t LDS_BYTES = (size_t)LAND_BYTES + (size_t)WAVES * ROW * 8 + 16 + 64;   // + next-group slots + pair flags
    static constexpr int PPW = (LAND_BYTES / 1024) / WAVES;                    // 1 KiB pieces per wavefront
    static_assert(PPW == 8, "the spread schedule places eight pieces");
    static_assert((ROW * 8) % 16 == 0, "rows must keep 16-byte alignment");
    static_assert(W == 8 || W == 4, "cross-wave radix 8 or 4");
};
typedef SplitFirT<8> SplitFir;

// pairwise hand-over through LDS flags (PSYNC): the mirror exchange couples wavefront d with wavefront 8 - d only (0 and 4 with
// themselves), so the two workgroup barriers around it become waits for ONE partner
__device__ __forceinline__ void lds_flag_set(unsigned* f, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // everything this wave wrote / read before is done
    *(volatile unsigned*)f = v;
}
__device__ __forceinline__ void lds_flag_wait(const unsigned* f, unsigned v) {
    while (*(volatile const unsigned*)f != v) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int PSYNC, int SPREAD = 0, int W = 8>
__global__ void __launch_bounds__(SplitFirT<W>::WG, 2)
fastconv_split_kernel(const float* __restrict__ x, float* __restrict__ y, const cx<float>* __restrict__ Hc,
                      int nblk, int step, int inputLen, int lastOut,
                      const cx<float>* __restrict__ twn,      // W_n^j, j < n
                      const cx<float>* __restrict__ tw1024,   // W_1024^j
                      const cx<float>* __restrict__ twr,      // W_N^k, k <= n/2, N = 2n
                      unsigned* ctr, int nsig, size_t xstride, size_t ystride, int xmode) {
    typedef float T;
    typedef cx<T> CX;
    typedef SplitFirT<W> S;
    typedef Tiled<typename S::Sub, FWD, 0> KF;
    typedef Tiled<typename S::Sub, BWD, 0> KB;
    constexpr int T_ = S::WG, CPT = S::CPT;
    constexpr int n = S::n, ROW = S::ROW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const chunk16* land16 = reinterpret_cast<const chunk16*>(smem_raw);
    CX* rows = reinterpret_cast<CX*>(smem_raw + S::LAND_BYTES);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + S::LAND_BYTES + (size_t)S::WAVES * ROW * 8);
    unsigned* flagZ = s_next + 4;                            // [8] spectrum of wavefront d is in its row (tag = iteration + 1)
    unsigned* flagR = s_next + 12;                           // [8] wavefront d has read its partner's row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    CX* row = rows + (size_t)wave * ROW;
    if (tid < 16) s_next[4 + tid] = 0u;
    const bool paired = wave != 0 && wave != W / 2;          // 0 and W/2 are their own partners

    // ---- per-thread constants
    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, lane, tw1024, nullptr);
    KB::load_tw(wb, lane, tw1024, nullptr);
    CX wa[2 * CPT];                                          // W_n^j of this thread's stage-A butterflies, j = 2 (tid + T cc) + u
#pragma unroll
    for (int i = 0; i < 2 * CPT; ++i) wa[i] = twn[2 * (tid + T_ * (i >> 1)) + (i & 1)];
    // folded coefficients of this thread's 16 bins k = wave + 8 k2, k2 = 2 lane + u + 128 d (slot u * 8 + d): derivation in
    // fft_fir.h (fastconv_part_kernel); bin 0 = (DC, Nyquist) and bin n/2 are their own mirrors
    CX cA[16], cB[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = wave + W * (2 * lane + (i >> 3) + 128 * (i & 7));
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
        cA[i] = a; cB[i] = b;
    }

    const bool dyn = ctr != nullptr;
    unsigned pend = 0;
    // the first TWO groups of a workgroup are static (its position in the sweep, and that plus the grid); the counter hands out what follows.
    // (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the launch and the last
    // workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    // xmode (round 5): every XCD works on CONTIGUOUS blocks (fft_fir.h xcd_local) - the static sweeps through the XCD-contiguous map, the
    // rest [2 grid, nblk_all) in eight contiguous ranges with one counter each (ctr[0 .. 7] next, ctr[8] done)
    const unsigned xl = xmode ? xcd_local(blockIdx.x, gridDim.x) : blockIdx.x;
    unsigned g = xl;
    pend = xl + gridDim.x;
    __syncthreads();
    const long long nblk_all = (long long)nblk * nsig;
    const unsigned xcd = blockIdx.x & 7u;
    const long long rest = nblk_all - 2ll * gridDim.x, xper = rest > 0 ? (rest + 7) / 8 : 0;
    const long long xbase = 2ll * gridDim.x + xcd * xper, xend = xbase + xper < nblk_all ? xbase + xper : nblk_all;
    unsigned* cnext = dyn ? ctr + (xmode ? xcd : 0u) : nullptr;      // (formed only when there is a counter: no arithmetic on a null pointer)
    auto grabbed = [&](unsigned v) -> unsigned {
        if (!xmode) return 2u * gridDim.x + v;
        const long long gg = xbase + v;
        return gg < xend ? (unsigned)gg : 0xffffffffu;
    };
    // copies that would run past the signal are clamped to its last 16 bytes (their samples are replaced by the zero padding of
    // src/pffastconv.c:231-233 when the operands are picked up)
    const float* nx_src = x;                                 // the next block's signal and first sample for this lane (issue_pieces)
    long nx_base = 0;
    auto issue_prep = [&](unsigned grp) {
        long long ba = (long long)grp;
        if (ba >= nblk_all) ba = nblk_all - 1;
        int sig, blk;
        fc_split(ba, nblk, nsig, sig, blk);
        nx_src = x + (size_t)sig * xstride;
        nx_base = (long)blk * step + lane * 4;
    };
    // pieces lo .. hi-1 of this wavefront's share (piece pv = wave + 8 i covers floats 256 pv .. of the block)
    auto issue_pieces = [&](int lo, int hi) {
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            const int pv = wave + S::WAVES * i;
            long e = nx_base + pv * 256;                     // first of this lane's 4 floats
            if (e > (long)inputLen - 4) e = inputLen >= 4 ? (long)inputLen - 4 : 0;
            glds16(nx_src + e, lds0 + (unsigned)(pv * 1024));
        }
    };
    issue_prep(g);
    issue_pieces(0, S::PPW);
    for (unsigned it = 0; (long long)g < nblk_all; ++it) {
        if (dyn && tid == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = grabbed(atomicAdd(cnext, 1u));
        }
        PF_DSTAMP(0);
        wait_vmcnt<0>();
        wg_sync_raw();                                       // (1) the block has landed; the rows are free (A' of the block before is done)
        PF_DSTAMP(1);
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        int sig, blk;
        fc_split((long long)g, nblk, nsig, sig, blk);
        const long off = (long)blk * step;
        const int numOut = (blk == nblk - 1) ? lastOut : step;
        float* ys = y + (size_t)sig * ystride;
        // ================= A: radix W across the wavefronts, operands z[j + 1024 q], j = 2 c + u, c = tid + T cc: chunk c + 512 q of the landed block
        // W_n^(j d), d = 1 .. W - 1, from W_n^j by products at most three deep (recomputed in A': the registers would stay pinned)
        auto powers = [&](int cc, CX (&p0)[W], CX (&p1)[W]) {
            p0[1] = wa[2 * cc]; p1[1] = wa[2 * cc + 1];
            asm volatile("" : "+v"(p0[1].x), "+v"(p0[1].y), "+v"(p1[1].x), "+v"(p1[1].y));
            p0[2] = cmul(p0[1], p0[1]); p0[3] = cmul(p0[2], p0[1]);
            p1[2] = cmul(p1[1], p1[1]); p1[3] = cmul(p1[2], p1[1]);
            if constexpr (W == 8) {
                p0[4] = cmul(p0[2], p0[2]); p0[5] = cmul(p0[4], p0[1]); p0[6] = cmul(p0[3], p0[3]); p0[7] = cmul(p0[4], p0[3]);
                p1[4] = cmul(p1[2], p1[2]); p1[5] = cmul(p1[4], p1[1]); p1[6] = cmul(p1[3], p1[3]); p1[7] = cmul(p1[4], p1[3]);
            }
        };
        {
            const long avail = (long)inputLen - off;         // samples of this block that exist
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                CX a0[W], a1[W];
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const int c = tid + T_ * cc + 512 * q;
                    const int e0 = 4 * c;
                    const chunk16 ck = land16[c];
                    float f0 = ck.x, f1 = ck.y, f2 = ck.z, f3 = ck.w;
                    if (e0 + 3 >= avail) {     // clamped copy (issue): element k is sample avail - 4 + k of the block
                        const long sh = (long)e0 - (avail - 4);      // >= 1
                        f0 = sh == 1 ? ck.y : sh == 2 ? ck.z : sh == 3 ? ck.w : 0.f;
                        f1 = sh == 1 ? ck.z : sh == 2 ? ck.w : 0.f;
                        f2 = sh == 1 ? ck.w : 0.f;
                        f3 = 0.f;
                    }
                    a0[q] = mk<T>(f0, f1);
                    a1[q] = mk<T>(f2, f3);
                }
                dftR<W, FWD>(a0);
                dftR<W, FWD>(a1);
                CX p0[W], p1[W];
                powers(cc, p0, p1);
#pragma unroll
                for (int d = 1; d < W; ++d) { a0[d] = cmul(a0[d], p0[d]); a1[d] = cmul(a1[d], p1[d]); }
#pragma unroll
                for (int d = 0; d < W; ++d) lds_st2(rows + (size_t)d * ROW + 2 * (tid + T_ * cc), a0[d], a1[d]);
            }
        }
        PF_DSTAMP(2);
        wg_sync_raw();                                       // (2) rows complete; the landing buffer is free
        PF_DSTAMP(3);
        // The next block's pieces go out ONE AT A TIME between the steps of B and B': issued back to back here, the 64 pieces of a
        // workgroup pass the CU's vector-memory queue one after the other (~50 cycles each) and the wavefront served last started B
        // 2 500 cycles behind the first - every later barrier waited for it (tools/dma_timeline.hip)
        issue_prep(gn);
        if constexpr (!SPREAD) issue_pieces(0, S::PPW);
        PF_DSTAMP(4);
#define PF_SPLIT_PIECE(I) do { if constexpr (SPREAD) issue_pieces(I, (I) + 1); } while (0)
        // ================= B: wavefront `wave` transforms row `wave`, wave-local
        CX v[16];
        {
            const chunk16* r16 = reinterpret_cast<const chunk16*>(row);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const chunk16 c = r16[lane + 64 * q];        // points 2 lane + 128 q, + 1: operands q of butterflies 2 lane, 2 lane + 1
                v[q] = mk<T>(c.x, c.y);
                v[8 + q] = mk<T>(c.z, c.w);
            }
        }
        KF::xsync();
        KF::template butterflies<0>(v, lane, wf, tw1024);
        PF_SPLIT_PIECE(0);
        KF::template xwrite<0>(v, lane, row); KF::xsync();
        KF::template xread<0>(v, lane, row); KF::xsync();
        PF_SPLIT_PIECE(1);
        KF::template butterflies<1>(v, lane, wf, tw1024);
        PF_SPLIT_PIECE(2);
        KF::template xwrite<1>(v, lane, row); KF::xsync();
        KF::template xread<1>(v, lane, row); KF::xsync();
        PF_SPLIT_PIECE(3);
        KF::template butterflies<2>(v, lane, wf, tw1024);
        PF_SPLIT_PIECE(4);
        PF_DSTAMP(5);
        // ================= M: Z[wave + W k2], k2 = 2 lane + u + 128 d -> row[k2]; mirrors conj Z[n - k] from row W - wave
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st2(row + 2 * lane + 128 * d, v[d], v[8 + d]);
        const unsigned tag = it + 1;
        if constexpr (PSYNC) {
            if (lane == 0) lds_flag_set(flagZ + wave, tag);
            if (paired) lds_flag_wait(flagZ + (W - wave), tag);
        } else {
            wg_sync_raw();                                   // (3) the whole packed spectrum sits in the rows
        }
        PF_DSTAMP(6);
        CX zm[16];
        if (wave != 0) {
            // n - k = (W - wave) + W (1023 - k2): the unit at point 1022 - 2 lane - 128 d holds the mirrors of u = 1, u = 0 in this order
            const CX* mrow = rows + (size_t)(W - wave) * ROW;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const vec4<float> m = lds_ld2(mrow + 1022 - 2 * lane - 128 * d);
                zm[8 + d] = mk<T>(m.x, m.y);
                zm[d] = mk<T>(m.z, m.w);
            }
        } else {
            // wave 0: n - W k2 = W (1024 - k2): the same row, index (1024 - k2) mod 1024
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                zm[d] = lds_ld(row + ((1024 - 2 * lane - 128 * d) & 1023));
                zm[8 + d] = lds_ld(row + (1023 - 2 * lane - 128 * d));
            }
        }
        if constexpr (PSYNC) {
            if (lane == 0) lds_flag_set(flagR + wave, tag);
        } else {
            wg_sync_raw();                                   // (3') every mirror is read: the rows are exchange buffers again
        }
        PF_DSTAMP(7);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const CX a = cA[i], bq = cB[i], zz = v[i], m = zm[i];
            v[i] = mk<T>(fma_(a.x, zz.x, fma_(-a.y, zz.y, fma_(bq.x, m.x, bq.y * m.y))),
                         fma_(a.x, zz.y, fma_(a.y, zz.x, fma_(bq.y, m.x, -(bq.x * m.y)))));
        }
        // ================= B': inverse transform of Z'[wave + 8 k2] (first-stage operands are in place), wave-local
        KB::template butterflies<0>(v, lane, wb, tw1024);
        if constexpr (PSYNC) { if (paired) lds_flag_wait(flagR + (W - wave), tag); }   // the partner is done with this row
        PF_SPLIT_PIECE(5);
        KB::template xwrite<0>(v, lane, row); KB::xsync();
        KB::template xread<0>(v, lane, row); KB::xsync();
        KB::template butterflies<1>(v, lane, wb, tw1024);
        PF_SPLIT_PIECE(6);
        KB::template xwrite<1>(v, lane, row); KB::xsync();
        KB::template xread<1>(v, lane, row); KB::xsync();
        PF_SPLIT_PIECE(7);
        KB::template butterflies<2>(v, lane, wb, tw1024);
#undef PF_SPLIT_PIECE
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st2(row + 2 * lane + 128 * d, v[d], v[8 + d]);   // b_wave[j], j = 2 lane + u + 128 d
        PF_DSTAMP(8);
        wg_sync_raw();                                       // (4)
        PF_DSTAMP(9);
        // ================= A': z'[j + 1024 q] = sum_d W_W^(-q d) conj(W_n^(j d)) b_d[j], j = 2 (tid + T cc) + u: the block's output samples
        {
            float* dst = ys + off;
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                CX a0[W], a1[W];
#pragma unroll
                for (int d = 0; d < W; ++d) {
                    const vec4<float> c = lds_ld2(rows + (size_t)d * ROW + 2 * (tid + T_ * cc));
                    a0[d] = mk<T>(c.x, c.y);
                    a1[d] = mk<T>(c.z, c.w);
                }
                CX p0[W], p1[W];
                powers(cc, p0, p1);
#pragma unroll
                for (int d = 1; d < W; ++d) { a0[d] = cmulc(a0[d], p0[d]); a1[d] = cmulc(a1[d], p1[d]); }
                dftR<W, BWD>(a0);
                dftR<W, BWD>(a1);
                // ---- the first numOut samples (src/pffastconv.c:255)
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const int e0 = 4 * (tid + T_ * cc + 512 * q);
                    if (e0 + 3 < numOut) {
                        F4u q4; q4.a = a0[q].x; q4.b = a0[q].y; q4.c = a1[q].x; q4.d = a1[q].y;
                        *reinterpret_cast<F4u*>(dst + e0) = q4;
                    } else {
                        if (e0 < numOut) dst[e0] = a0[q].x;
                        if (e0 + 1 < numOut) dst[e0 + 1] = a0[q].y;
                        if (e0 + 2 < numOut) dst[e0 + 2] = a1[q].x;
                    }
                }
            }
        }
        PF_DSTAMP(40);
        g = gn;
    }
    wait_vmcnt<0>();
    if (dyn && tid == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[xmode ? 8 : 1], 1u);
        if (d == gridDim.x - 1) {
            if (xmode) { for (int i = 0; i < 9; ++i) atomicExch(&ctr[i], 0u); }
            else { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same organisation for calls with FEW blocks (the stated C4 call: 2^20 samples, 4096 taps = 255 reference-sized blocks of
// 8192 samples, one per CU): latency, not throughput.  n = W x 512: W wavefronts, EIGHT points per lane - A: radix W across the
// wavefronts straight from HBM (no landing buffer: a workgroup has one block), B / B': wave-local 512-point transforms (8 x 8 x 8, two
// wave-local exchanges), M: mirror exchange through the rows.  Against the lock-step kernel on 512 threads (fft_fir.h FirCfg::C4096m:
// five stages, eight workgroup-wide exchanges, tools/fir_timeline.hip: 12 400 cycles from the first butterfly to the last store) the
// dependent chain of a wavefront is two short transforms and four barriers.
template <int W> struct SplitOneT {
    typedef TiledCfg<float, 9, 64, 3, 8, 8, 8, 1, 4, 8, 3, 0, 512, 1> Sub;     // the wave-local 512-point transform, 8 points per lane
    static constexpr int WAVES = W, M = 512, n = W * M, WG = 64 * W;
    static constexpr int JPT = 8 / W;                                          // stage-A butterflies (points j) per thread
    static constexpr int ROW = (Sub::IMG_NAT > Sub::IMG_TRN ? Sub::IMG_NAT : Sub::IMG_TRN) + 8;
    static constexpr size_t LDS_BYTES = (size_t)WAVES * ROW * 8 + 16;
    static_assert(W == 8 || W == 4, "cross-wave radix 8 or 4");
};

// Folded per-bin coefficients of fastconv_split1_kernel, once per filter (the reference transforms the filter once per setup too:
// src/pffastconv.c:108): AB[d T + tid] = (A, B) of bin k = wave + W (lane + 64 d) - real finalize, x Hf / Nfft and real preprocess as
// Z'[k] = A Z[k] + B conj Z[n - k] (derivation: fft_fir.h fastconv_part_kernel; bin 0 = (DC, Nyquist) and bin n/2 are their own
// mirrors).  In the kernel's prologue this was 8 bins x 4 scattered table reads + 40 flops per thread, and 255 workgroups did it at
// the same time: 6 000 of the 17 800 cycles of the stated C4 call (tools/dma_timeline.hip); now it is 8 coalesced 16-byte loads.
template <int W>
__global__ void __launch_bounds__(SplitOneT<W>::WG)
fastconv_split1_coef_kernel(const cx<float>* __restrict__ Hc, const cx<float>* __restrict__ twr, vec4<float>* __restrict__ AB) {
    typedef float T;
    typedef cx<T> CX;
    typedef SplitOneT<W> S;
    constexpr int n = S::n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int d = 0; d < 8; ++d) {
        const int k = wave + W * (lane + 64 * d);
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
        AB[d * S::WG + tid] = o;
    }
}

template <int W>
__global__ void __launch_bounds__(SplitOneT<W>::WG, 2)
fastconv_split1_kernel(const float* __restrict__ x, float* __restrict__ y, const vec4<float>* __restrict__ AB,
                       int nblk, int step, int inputLen, int lastOut,
                       const cx<float>* __restrict__ twn,      // W_n^j, j < n
                       const cx<float>* __restrict__ tw512,    // W_512^j
                       int nsig, size_t xstride, size_t ystride, int xmode) {
    typedef float T;
    typedef cx<T> CX;
    typedef SplitOneT<W> S;
    typedef Tiled<typename S::Sub, FWD, 0> KF;
    typedef Tiled<typename S::Sub, BWD, 0> KB;
    constexpr int n = S::n, ROW = S::ROW, T_ = S::WG, JPT = S::JPT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    CX* rows = reinterpret_cast<CX*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    CX* row = rows + (size_t)wave * ROW;
    const long long nblk_all = (long long)nblk * nsig;
    struct __attribute__((packed, aligned(4))) F2u { float a, b; };   // block offsets are multiples of 4 bytes only
    typedef vec2<float> F2;
    // ---- stage-A operands of block group grp: z[j + 512 q] = the float pair at 2 (j + 512 q), zero beyond the end of the signal
    //      (src/pffastconv.c:231-233); requested before the tables (their misses overlap)
    F2 raw[JPT][W];
    auto gather = [&](long long ba) {
        int sg, bk;
        fc_split(ba, nblk, nsig, sg, bk);
        const float* src = x + (size_t)sg * xstride + (long)bk * step;
        const long avail = (long)inputLen - (long)bk * step;
#pragma unroll
        for (int jj = 0; jj < JPT; ++jj)
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int e0 = 2 * (tid + T_ * jj + 512 * q);
                F2 r;
                if (e0 + 1 < avail) { const F2u q2 = *reinterpret_cast<const F2u*>(src + e0); r.x = q2.a; r.y = q2.b; }   // 8 bytes, 4-byte aligned
                else { r.x = e0 < avail ? src[e0] : 0.f; r.y = 0.f; }
                raw[jj][q] = r;
            }
    };
    const unsigned it = 3; (void)it;      // (PF_DSTAMP's iteration filter: one-shot kernel)
    PF_DSTAMP(0);
    // (xmode: sweep positions through the XCD-contiguous map - adjacent blocks share taps - 1 samples and meet in one L2)
    long long g = xmode ? xcd_local(blockIdx.x, gridDim.x) : blockIdx.x;
    if (g < nblk_all) gather(g);
    PF_DSTAMP(1);
    // ---- per-thread constants
    typename KF::Tw wf;
    typename KB::Tw wb;
    KF::load_tw(wf, lane, tw512, nullptr);
    KB::load_tw(wb, lane, tw512, nullptr);
    CX wa[JPT];
#pragma unroll
    for (int jj = 0; jj < JPT; ++jj) wa[jj] = twn[tid + T_ * jj];
    // folded coefficients of this thread's 8 bins k = wave + W k2, k2 = lane + 64 d: from the table of fastconv_split1_coef_kernel
    CX cA[8], cB[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const vec4<float> ab = AB[d * T_ + tid];
        cA[d] = mk<T>(ab.x, ab.y); cB[d] = mk<T>(ab.z, ab.w);
    }
    auto powers = [&](int jj, CX (&p)[W]) {                  // W_n^(j d), d = 1 .. W - 1
        p[1] = wa[jj];
        asm volatile("" : "+v"(p[1].x), "+v"(p[1].y));
        p[2] = cmul(p[1], p[1]); p[3] = cmul(p[2], p[1]);
        if constexpr (W == 8) { p[4] = cmul(p[2], p[2]); p[5] = cmul(p[4], p[1]); p[6] = cmul(p[3], p[3]); p[7] = cmul(p[4], p[3]); }
    };
    PF_DSTAMP(2);
    for (; g < nblk_all; g += gridDim.x) {
        int sig, blk;
        fc_split(g, nblk, nsig, sig, blk);
        const long off = (long)blk * step;
        const int numOut = (blk == nblk - 1) ? lastOut : step;
        float* dst = y + (size_t)sig * ystride + off;
        // ================= A: radix W across the wavefronts, times W_n^(j d), into row d
#pragma unroll
        for (int jj = 0; jj < JPT; ++jj) {
            CX a[W];
#pragma unroll
            for (int q = 0; q < W; ++q) a[q] = raw[jj][q];
            dftR<W, FWD>(a);
            CX p[W];
            powers(jj, p);
#pragma unroll
            for (int d = 1; d < W; ++d) a[d] = cmul(a[d], p[d]);
#pragma unroll
            for (int d = 0; d < W; ++d) lds_st(rows + (size_t)d * ROW + tid + T_ * jj, a[d]);
        }
        PF_DSTAMP(3);
        wg_sync_raw();                                       // (1) rows complete
        PF_DSTAMP(4);
        // ================= B: wavefront `wave` transforms row `wave` (512 points, operands lane + 64 q), wave-local
        CX v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lds_ld(row + lane + 64 * q);
        KF::xsync();
        KF::template butterflies<0>(v, lane, wf, tw512);
        KF::template xwrite<0>(v, lane, row); KF::xsync();
        KF::template xread<0>(v, lane, row); KF::xsync();
        KF::template butterflies<1>(v, lane, wf, tw512);
        KF::template xwrite<1>(v, lane, row); KF::xsync();
        KF::template xread<1>(v, lane, row); KF::xsync();
        KF::template butterflies<2>(v, lane, wf, tw512);
        PF_DSTAMP(5);
        // ================= M: Z[wave + W k2], k2 = lane + 64 d -> row[k2]; mirrors conj Z[n - k] from row W - wave, index 511 - k2
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st(row + lane + 64 * d, v[d]);
        wg_sync_raw();                                       // (2) the whole packed spectrum sits in the rows
        CX zm[8];
        if (wave != 0) {
            const CX* mrow = rows + (size_t)(W - wave) * ROW;
#pragma unroll
            for (int d = 0; d < 8; ++d) zm[d] = lds_ld(mrow + 511 - lane - 64 * d);
        } else {
#pragma unroll
            for (int d = 0; d < 8; ++d) zm[d] = lds_ld(row + ((512 - lane - 64 * d) & 511));
        }
        wg_sync_raw();                                       // (3) every mirror is read: the rows are exchange buffers again
        PF_DSTAMP(6);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const CX a = cA[i], bq = cB[i], zz = v[i], m = zm[i];
            v[i] = mk<T>(fma_(a.x, zz.x, fma_(-a.y, zz.y, fma_(bq.x, m.x, bq.y * m.y))),
                         fma_(a.x, zz.y, fma_(a.y, zz.x, fma_(bq.y, m.x, -(bq.x * m.y)))));
        }
        // ================= B': inverse transform of Z'[wave + W k2] (first-stage operands are in place), wave-local
        KB::template butterflies<0>(v, lane, wb, tw512);
        KB::template xwrite<0>(v, lane, row); KB::xsync();
        KB::template xread<0>(v, lane, row); KB::xsync();
        KB::template butterflies<1>(v, lane, wb, tw512);
        KB::template xwrite<1>(v, lane, row); KB::xsync();
        KB::template xread<1>(v, lane, row); KB::xsync();
        KB::template butterflies<2>(v, lane, wb, tw512);
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st(row + lane + 64 * d, v[d]);        // b_wave[j], j = lane + 64 d
        PF_DSTAMP(7);
        wg_sync_raw();                                       // (4)
        PF_DSTAMP(8);
        const long long gnext = g + gridDim.x;
        // ================= A': z'[j + 512 q] = sum_d W_W^(-q d) conj(W_n^(j d)) b_d[j]: output samples 2 (j + 512 q), + 1
#pragma unroll
        for (int jj = 0; jj < JPT; ++jj) {
            CX a[W];
#pragma unroll
            for (int d = 0; d < W; ++d) a[d] = lds_ld(rows + (size_t)d * ROW + tid + T_ * jj);
            CX p[W];
            powers(jj, p);
#pragma unroll
            for (int d = 1; d < W; ++d) a[d] = cmulc(a[d], p[d]);
            dftR<W, BWD>(a);
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int e0 = 2 * (tid + T_ * jj + 512 * q);
                if (e0 + 1 < numOut) { F2u q2; q2.a = a[q].x; q2.b = a[q].y; *reinterpret_cast<F2u*>(dst + e0) = q2; }
                else if (e0 < numOut) dst[e0] = a[q].x;
            }
        }
        PF_DSTAMP(9);
        if (gnext < nblk_all) gather(gnext);                 // (a call with more blocks than workgroups: the next block's operands; every
                                                             //  thread re-writes exactly the row slots it has just read: no barrier)
    }
}

}  // namespace pf
