// Streaming versions of the spectrum helpers: pffft_zreorder (src/pffft_priv_impl.h:1158-1193) and
// pffft_zconvolve_accumulate / _no_accu (:1534-1684) on device-resident batches.
//
// Both are pure HBM streams.  The first versions (fft_generic.h: grid-stride, one 64-bit division per element,
// scalar 4-byte gathers straight from global memory for zreorder) measured, on MI355X (tools/aux_bench.py):
// zconvolve 0.65-0.71 of 8 TB/s = the rate of a plain grid-stride copy, zreorder 0.41-0.61.  Here:
//   * zreorder goes through an LDS image of the internal layout (the padded block image of fft_stock.h): linear
//     16-byte chunks on both HBM sides, the permutation is done by 4-byte LDS accesses -> 0.61-0.70;
//   * zconvolve issues all loads of two pairs per thread before the first product, remainders in 32 bits -> 0.69-0.70;
//   * both can pull chunks in order from the work counter (SkSched, fft_stock.h) - measured SLOWER here
//     (0.47-0.62 / 0.57-0.60; with one 16 KiB group per atomic even 0.16: one counter address serves ~80 M atomics/s),
//     so the launchers assign statically and keep the in-order path as the alternative route of the tests (AB_INORDER_SMALL, pf_route.h).
#pragma once
#include "fft_stock.h"
#include "fft_tiled.h"  // ChunkOps

namespace pf {

constexpr int ZC_THREADS = 512;
constexpr int ZC_U = 2;                                  // (re, im) group pairs per thread and chunk
constexpr int ZC_CHUNK = ZC_THREADS * ZC_U;              // pairs per chunk: 32 KiB of a, of b and of ab in float

// One thread handles a (re-group, im-group) pair = 4 complex products.  total = batch * npairs, npairs = n / 4.
template <typename T, int ACC>
__global__ void __launch_bounds__(ZC_THREADS)
zconvolve_stream_kernel(const T* a, const T* b, T* ab, size_t total, unsigned npairs, int is_real, T scaling,
                        int b_broadcast, unsigned* ctr, unsigned kchunk) {
    __shared__ unsigned s_next[4];
    SkSched sch;
    sch.dyn = ctr != nullptr; sch.ctr = ctr; sch.s_next = s_next; sch.K = kchunk; sch.pend = 0;
    sch.grab(threadIdx.x == 0);
    __syncthreads();
    sch.start();
    __syncthreads();
    const vec4<T>* a4 = reinterpret_cast<const vec4<T>*>(a);
    const vec4<T>* b4 = reinterpret_cast<const vec4<T>*>(b);
    vec4<T>* ab4 = reinterpret_cast<vec4<T>*>(ab);
    const bool need_pr = is_real || b_broadcast;
    const float inv_np = 1.0f / (float)npairs;
    while ((size_t)sch.gcur * ZC_CHUNK < total) {
        sch.top(threadIdx.x == 0);
        const size_t base = (size_t)sch.gcur * ZC_CHUNK;
        unsigned base_pr = 0;
        if (need_pr) base_pr = (unsigned)(base % npairs);   // once per chunk; the per-pair remainder is 32-bit
        vec4<T> ar[ZC_U], ai[ZC_U], br[ZC_U], bi[ZC_U], cr[ZC_U], ci[ZC_U];
        unsigned pr[ZC_U];
#pragma unroll
        for (int u = 0; u < ZC_U; ++u) {
            const unsigned k = threadIdx.x + ZC_THREADS * u;
            size_t i = base + k;
            if (i >= total) i = total - 1;   // clamped, unconditional loads; the store is predicated
            unsigned x = base_pr + k;        // < npairs + ZC_CHUNK
            if (need_pr) {
                if (npairs >= (unsigned)ZC_CHUNK) x = x >= npairs ? x - npairs : x;
                else x -= (unsigned)fdiv((int)x, (int)npairs, inv_np) * npairs;
            }
            pr[u] = x;
            const size_t ib = b_broadcast ? (size_t)x : i;
            ar[u] = __builtin_nontemporal_load(a4 + 2 * i); ai[u] = __builtin_nontemporal_load(a4 + 2 * i + 1);
            br[u] = b_broadcast ? b4[2 * ib] : __builtin_nontemporal_load(b4 + 2 * ib);
            bi[u] = b_broadcast ? b4[2 * ib + 1] : __builtin_nontemporal_load(b4 + 2 * ib + 1);
            if (ACC) { cr[u] = __builtin_nontemporal_load(ab4 + 2 * i); ci[u] = __builtin_nontemporal_load(ab4 + 2 * i + 1); }
        }
#pragma unroll
        for (int u = 0; u < ZC_U; ++u) {
            const size_t i = base + threadIdx.x + ZC_THREADS * u;
            vec4<T> pre = ar[u] * br[u] - ai[u] * bi[u], pim = ar[u] * bi[u] + ai[u] * br[u];
            if (is_real && pr[u] == 0) {  // DC and Nyquist are both real: multiply separately (:1626-1629, :1680-1683)
                pre.x = ar[u].x * br[u].x;
                pim.x = ai[u].x * bi[u].x;
            }
            vec4<T> o0, o1;
            if (ACC) { o0 = cr[u] + pre * scaling; o1 = ci[u] + pim * scaling; }
            else { o0 = pre * scaling; o1 = pim * scaling; }
            if (i < total) {
                __builtin_nontemporal_store(o0, ab4 + 2 * i);
                __builtin_nontemporal_store(o1, ab4 + 2 * i + 1);
            }
        }
        __syncthreads();
        sch.advance();
    }
    sch.finish(threadIdx.x == 0);
}

// ---- zconvolve for long batches: the streaming organisation of the mixer kernel (pfdsp_mix.h) -----------------
// Persistent workgroups of 8 wavefronts pull groups of 8 consecutive chunks from an atomic counter (grabbed two
// iterations ahead), a wave moves 8 rows of 64 x 16 bytes of EVERY stream per chunk — every global access is a fully
// coalesced 1 KiB (float) wave instruction — and the loads of the next chunk are in flight while the current one is
// multiplied and stored.  In the internal layout 4-scalar re-groups and im-groups alternate, so with one 16-byte unit
// per lane the two halves of a complex product sit in neighbouring lanes: they are exchanged with DPP quad swaps
// (even lane: re-group, odd lane: im-group):
//     P = A*B (own), X = A*B_partner;   even: out = P - P_partner (Re),   odd: out = X + X_partner (Im)
// q = 16-byte unit index over the whole batch, nq = units per vector (n/2); DC/Nyquist of a real spectrum are the .x
// of units 0 and 1 of a vector and multiply as reals (src/pffft_priv_impl.h:1626-1629, :1680-1683): out.x = P.x.
constexpr int ZD_WAVES = 8;
// One 16-byte unit per lane in BOTH precisions (a 32-byte unit per lane made every wave instruction touch half of each
// 128-byte line: double ran at 0.66 of the roofline against 0.79 for float).  float: a unit is a whole 4-scalar re- or
// im-group, the partner group sits in lane ^ 1; double: a unit is HALF a group (2 scalars), the matching half of the
// partner group sits in lane ^ 2.  DC / Nyquist of a real spectrum are the .x of the first re-unit and the first im-unit
// of a vector: units 0, 1 (float) / 0, 2 (double).
template <typename T> __device__ __forceinline__ T dpp_swap1(T v);   // lane ^ 1 inside every quad
template <> __device__ __forceinline__ float dpp_swap1<float>(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <typename T> struct Zd;
template <> struct Zd<float> {
    typedef vec4<float> V;
    static constexpr int ROWS = 8, UPG = 1;                  // 8 KiB of every stream per wave chunk; units per 4-scalar group
    static constexpr unsigned CHUNK = 64 * ROWS;             // units per wave chunk
    static __device__ __forceinline__ float swap1(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    }
    static __device__ __forceinline__ V swap(V v) { V o; o.x = swap1(v.x); o.y = swap1(v.y); o.z = swap1(v.z); o.w = swap1(v.w); return o; }
};
template <> struct Zd<double> {
    typedef vec2<double> V;
    static constexpr int ROWS = 8, UPG = 2;
    static constexpr unsigned CHUNK = 64 * ROWS;
    static __device__ __forceinline__ double swap1(double v) {                                                          // lane ^ 2
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, 0x4E, 0xF, 0xF, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), 0x4E, 0xF, 0xF, true);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    static __device__ __forceinline__ V swap(V v) { V o; o.x = swap1(v.x); o.y = swap1(v.y); return o; }
};

template <typename T, int ACC, int BCAST>
__global__ void __launch_bounds__(ZD_WAVES * 64)
zconvolve_dyn_kernel(const T* a, const T* b, T* ab, unsigned long long Q, unsigned nq, int is_real, T scaling, unsigned* ctr) {
    typedef typename Zd<T>::V V;
    constexpr int UPG = Zd<T>::UPG;
    constexpr int ZD_ROWS = Zd<T>::ROWS;
    constexpr unsigned ZD_CHUNK = Zd<T>::CHUNK;
    __shared__ unsigned s_next[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool odd = (lane / UPG) & 1;               // this lane holds (part of) an im-group
    const V* a4 = reinterpret_cast<const V*>(a);
    const V* b4 = reinterpret_cast<const V*>(b);
    V* c4 = reinterpret_cast<V*>(ab);
    const unsigned long long nchunks = (Q + ZD_CHUNK - 1) / ZD_CHUNK;
    const bool need_rem = BCAST || is_real;
    unsigned pend = 0;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    unsigned g = blockIdx.x;
    V xa[ZD_ROWS], xb[ZD_ROWS], xc[ZD_ROWS];
    unsigned rem[ZD_ROWS];
    // all loads of one chunk; indices clamped to the last unit so that they are unconditional
    auto load_chunk = [&](unsigned long long c, V (&la)[ZD_ROWS], V (&lb)[ZD_ROWS], V (&lc)[ZD_ROWS], unsigned (&lr)[ZD_ROWS]) {
        if (c >= nchunks) c = nchunks - 1;
        const unsigned long long q0 = c * ZD_CHUNK + lane;
        unsigned r0 = 0;
        if (need_rem) r0 = (unsigned)((c * ZD_CHUNK) % nq);          // wave-uniform, once per chunk
#pragma unroll
        for (int r = 0; r < ZD_ROWS; ++r) {
            unsigned long long q = q0 + 64 * r;
            if (q >= Q) q = Q - 1;
            unsigned x = 0;
            if (need_rem) x = (r0 + (unsigned)(64 * r + lane)) % nq;   // 32-bit
            lr[r] = x;
            la[r] = __builtin_nontemporal_load(a4 + q);
            lb[r] = BCAST ? b4[x] : __builtin_nontemporal_load(b4 + q);
            if (ACC) lc[r] = __builtin_nontemporal_load(c4 + q);
        }
    };
    load_chunk((unsigned long long)g * ZD_WAVES + wave, xa, xb, xc, rem);
    for (unsigned it = 0; (unsigned long long)g * ZD_WAVES < nchunks; ++it) {
        if (threadIdx.x == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        __syncthreads();
        const unsigned gn = s_next[(it + 1) & 1];
        const unsigned long long c = (unsigned long long)g * ZD_WAVES + wave;
        V na[ZD_ROWS], nb[ZD_ROWS], nc[ZD_ROWS];
        unsigned nrem[ZD_ROWS];
        load_chunk((unsigned long long)gn * ZD_WAVES + wave, na, nb, nc, nrem);
        if (c < nchunks) {
            const unsigned long long q0 = c * ZD_CHUNK + lane;
#pragma unroll
            for (int r = 0; r < ZD_ROWS; ++r) {
                const V A = xa[r], B = xb[r];
                const V Bp = Zd<T>::swap(B);
                const V P = A * B, X = A * Bp;
                const V S = odd ? P : X;            // what the partner needs
                const V R = Zd<T>::swap(S);
                V o = odd ? X + R : P - R;
                if (is_real && (rem[r] == 0 || rem[r] == UPG)) o.x = P.x;
                if (ACC) o = xc[r] + o * scaling; else o = o * scaling;
                const unsigned long long q = q0 + 64 * r;
                if (q < Q) __builtin_nontemporal_store(o, c4 + q);
            }
        }
#pragma unroll
        for (int r = 0; r < ZD_ROWS; ++r) { xa[r] = na[r]; xb[r] = nb[r]; if (ACC) xc[r] = nc[r]; rem[r] = nrem[r]; }
        g = gn;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// zreorder through the padded block image: G vectors per pass.  to_canonical: internal -> canonical (PFFFT_FORWARD).
constexpr int ZR_THREADS = 256;
template <typename T>
__global__ void __launch_bounds__(ZR_THREADS)
zreorder_lds_kernel(const T* in, T* out, size_t batch, int n, int is_real, int to_canonical, int G, unsigned m_n4,
                    unsigned m_nchk, unsigned* ctr, unsigned kchunk) {
    typedef vec4<float> chunk16;
    constexpr int IBS = SkIbs<T>::v, CH = 16 / (int)sizeof(T), CPB = 32 / CH, BCH = IBS / CH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* const lds = reinterpret_cast<T*>(smem_raw);
    chunk16* const lds16 = reinterpret_cast<chunk16*>(smem_raw);
    const int nchk = 2 * n / CH;                  // 16-byte chunks per vector
    const int img16 = (n / 16) * BCH + 1;         // block image per vector, in chunks
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + (size_t)G * img16 * 16);
    const int n4 = n >> 2, tid = threadIdx.x;
    SkSched sch;
    sch.dyn = ctr != nullptr; sch.ctr = ctr; sch.s_next = s_next; sch.K = kchunk; sch.pend = 0;
    sch.grab(tid == 0);
    __syncthreads();
    sch.start();
    __syncthreads();
    const chunk16* in16 = reinterpret_cast<const chunk16*>(in);
    chunk16* out16 = reinterpret_cast<chunk16*>(out);
    // scalar index inside a vector's block image of scalar e of canonical chunk cc
    auto ipos_of = [&](int cc, int e) -> int {
        const int s = cc * CH + e, bin = s >> 1, part = s & 1;
        const int ip = is_real ? sk_iposr<T>(bin, n4, m_n4, IBS)
                               : (IBS * ((bin - udiv(bin, m_n4) * n4) >> 2) + 8 * udiv(bin, m_n4) + ((bin - udiv(bin, m_n4) * n4) & 3));
        return ip + 4 * part;
    };
    while ((size_t)sch.gcur * G < batch) {
        sch.top(tid == 0);
        const size_t t0 = (size_t)sch.gcur * G;
        const int cnt = (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G);
        const int tot = cnt * nchk;
        const chunk16* src = in16 + t0 * nchk;
        chunk16* dst = out16 + t0 * nchk;
        // ---- phase 1: 4 independent 16-byte loads in flight per thread, then into the image
        for (int c0 = tid; c0 < tot; c0 += 4 * ZR_THREADS) {
            chunk16 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * ZR_THREADS;
                v[u] = __builtin_nontemporal_load(src + (c < tot ? c : tot - 1));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * ZR_THREADS;
                if (c >= tot) continue;
                const int g = udiv(c, m_nchk), cc = c - g * nchk;
                if (to_canonical) {
                    lds16[g * img16 + (cc / CPB) * BCH + (cc % CPB)] = v[u];
                } else {
                    T* img = lds + (size_t)g * img16 * CH;
#pragma unroll
                    for (int e = 0; e < CH; ++e) img[ipos_of(cc, e)] = ChunkOps<T>::get(v[u], e);
                }
            }
        }
        __syncthreads();
        // ---- phase 2: out of the image, linear 16-byte stores
        for (int c = tid; c < tot; c += ZR_THREADS) {
            const int g = udiv(c, m_nchk), cc = c - g * nchk;
            chunk16 o;
            if (to_canonical) {
                const T* img = lds + (size_t)g * img16 * CH;
#pragma unroll
                for (int e = 0; e < CH; ++e) ChunkOps<T>::set(o, e, img[ipos_of(cc, e)]);
            } else {
                o = lds16[g * img16 + (cc / CPB) * BCH + (cc % CPB)];
            }
            __builtin_nontemporal_store(o, dst + c);
        }
        __syncthreads();
        sch.advance();
    }
    sch.finish(tid == 0);
}

// ---- zreorder for long batches: the same image, the streaming organisation of the mixer kernel (pfdsp_mix.h) ----
// One persistent workgroup of 8 wavefronts per CU pulls groups of G vectors (<= 64 KiB) IN ORDER from an atomic counter
// (grabbed two iterations ahead); the 16-byte loads of the NEXT group sit in registers (8 per thread) while the current
// group leaves the LDS image.  Same index maps as zreorder_lds_kernel above.
constexpr int ZRD_THREADS = 512, ZRD_U = 8;
constexpr size_t ZRD_GROUP_BYTES = (size_t)ZRD_THREADS * ZRD_U * 16;   // 64 KiB

// internal -> canonical goes through an image of the CANONICAL vector instead: the re-group and the im-group of the same
// four bins sit in neighbouring lanes (float: lane ^ 1, double: lane ^ 2), one DPP swap interleaves them into whole
// (re, im) bins, and each lane writes 16 bytes of canonical spectrum (two 8-byte bins where a real spectrum's odd quarters
// run backwards) — no scalar LDS access on either side.  Quarter q of the image is shifted by q * ZrdPad chunks so that
// the four quarters a wave writes at once fall into different banks.
template <typename T> struct ZrdPad { static constexpr int v = sizeof(T) == 4 ? 2 : 4; };   // 16-byte chunks per quarter
template <typename T> __host__ __device__ constexpr int zrd_canon_img16(int n) { return 2 * n * (int)sizeof(T) / 16 + 4 * ZrdPad<T>::v; }

__device__ __forceinline__ float dpp_swap2(float v) {   // lane ^ 2 inside every quad
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ double dpp_swap2(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, 0x4E, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), 0x4E, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <typename T, int TO_CANON>
__global__ void __launch_bounds__(ZRD_THREADS)
zreorder_dyn_kernel(const T* in, T* out, size_t batch, int n, int is_real, int G, unsigned m_n4, unsigned m_nchk,
                    unsigned* ctr) {
    typedef vec4<float> chunk16;
    constexpr int IBS = SkIbs<T>::v, CH = 16 / (int)sizeof(T), CPB = 32 / CH, BCH = IBS / CH, PADC = ZrdPad<T>::v;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* const lds = reinterpret_cast<T*>(smem_raw);
    chunk16* const lds16 = reinterpret_cast<chunk16*>(smem_raw);
    const int nchk = 2 * n / CH;                  // 16-byte chunks per vector
    const int img16 = TO_CANON ? zrd_canon_img16<T>(n) : (n / 16) * BCH + 1;   // image per vector, in chunks
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + (size_t)G * img16 * 16);
    const int n4 = n >> 2, tid = threadIdx.x;
    const chunk16* in16 = reinterpret_cast<const chunk16*>(in);
    chunk16* out16 = reinterpret_cast<chunk16*>(out);
    auto ipos_of = [&](int cc, int e) -> int {
        const int sc = cc * CH + e, bin = sc >> 1, part = sc & 1;
        const int ip = is_real ? sk_iposr<T>(bin, n4, m_n4, IBS)
                               : (IBS * ((bin - udiv(bin, m_n4) * n4) >> 2) + 8 * udiv(bin, m_n4) + ((bin - udiv(bin, m_n4) * n4) & 3));
        return ip + 4 * part;
    };
    // canonical bin of element t of quarter q (fft_generic.h bin_of): odd quarters of a real spectrum run backwards
    auto bin_qt = [&](int q, int t) -> int {
        if (is_real && (q & 1)) return q * n4 + (t ? n4 - t : 0);
        return q * n4 + t;
    };
    const size_t ngroups = (batch + G - 1) / G;
    const size_t last_chunk = batch * (size_t)nchk - 1;
    const size_t gchunks = (size_t)G * nchk;
    unsigned pend = 0;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    unsigned g = blockIdx.x;
    chunk16 v[ZRD_U];
    auto load_group = [&](size_t grp) {           // clamped: unconditional loads
        if (grp >= ngroups) grp = ngroups - 1;
        const size_t c0 = grp * gchunks + tid;
#pragma unroll
        for (int u = 0; u < ZRD_U; ++u) {
            size_t c = c0 + (size_t)u * ZRD_THREADS;
            v[u] = __builtin_nontemporal_load(in16 + (c < last_chunk ? c : last_chunk));
        }
    };
    load_group(g);
    for (unsigned it = 0; g < ngroups; ++it) {
        if (tid == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        const size_t t0 = (size_t)g * G;
        const int cnt = (int)((batch - t0) < (size_t)G ? (batch - t0) : (size_t)G);
        const int tot = cnt * nchk;
        // ---- phase 1: the prefetched chunks into the image
#pragma unroll
        for (int u = 0; u < ZRD_U; ++u) {
            const int c = tid + u * ZRD_THREADS;
            const int gg = udiv(c, m_nchk), cc = c - gg * nchk;
            if constexpr (TO_CANON && sizeof(T) == 4) {
                // cc = 8 b + 2 q + p: lane pair (p = 0: re-group, p = 1: im-group) of block b, quarter q
                const bool odd = cc & 1;
                const vec4<float> x = v[u];
                const float s0 = dpp_swap1<float>(odd ? x.x : x.z), s1 = dpp_swap1<float>(odd ? x.y : x.w);
                vec4<float> o;                     // even: bins t0, t0+1 = (re0, im0, re1, im1); odd: (re2, im2, re3, im3)
                if (odd) { o.x = s0; o.y = x.z; o.z = s1; o.w = x.w; } else { o.x = x.x; o.y = s0; o.z = x.y; o.w = s1; }
                if (c >= tot) continue;
                const int b = cc >> 3, q = (cc >> 1) & 3, t = 4 * b + (odd ? 2 : 0);
                float* img = reinterpret_cast<float*>(lds16 + gg * img16 + PADC * q);
                if (is_real && (q & 1)) {
                    vec2<float> lo, hi; lo.x = o.x; lo.y = o.y; hi.x = o.z; hi.y = o.w;
                    *reinterpret_cast<vec2<float>*>(img + 2 * bin_qt(q, t)) = lo;
                    *reinterpret_cast<vec2<float>*>(img + 2 * bin_qt(q, t + 1)) = hi;
                } else {
                    *reinterpret_cast<vec4<float>*>(img + 2 * (q * n4 + t)) = o;
                }
            } else if constexpr (TO_CANON) {
                // cc = 16 b + 4 q + 2 p + h: lanes (p, h) hold (re|im)_{2h}, (re|im)_{2h+1}; partner = lane ^ 2
                const int p = (cc >> 1) & 1, h = cc & 1;
                const vec2<double> x = __builtin_bit_cast(vec2<double>, v[u]);
                const double r = dpp_swap2(p ? x.x : x.y);
                vec2<double> o;                    // p = 0: bin 2h = (re, im); p = 1: bin 2h+1
                if (p) { o.x = r; o.y = x.y; } else { o.x = x.x; o.y = r; }
                if (c >= tot) continue;
                const int b = cc >> 4, q = (cc >> 2) & 3, t = 4 * b + 2 * h + p;
                lds16[gg * img16 + PADC * q + bin_qt(q, t)] = __builtin_bit_cast(chunk16, o);
            } else {
                if (c >= tot) continue;
                T* img = lds + (size_t)gg * img16 * CH;
#pragma unroll
                for (int e = 0; e < CH; ++e) img[ipos_of(cc, e)] = ChunkOps<T>::get(v[u], e);
            }
        }
        __syncthreads();
        const unsigned gn = s_next[(it + 1) & 1];
        load_group(gn);                           // in flight during phase 2
        // ---- phase 2: out of the image, linear 16-byte stores
        chunk16* dst = out16 + t0 * nchk;
        for (int c = tid; c < tot; c += ZRD_THREADS) {
            const int gg = udiv(c, m_nchk), cc = c - gg * nchk;
            chunk16 o;
            if constexpr (TO_CANON) o = lds16[gg * img16 + cc + PADC * udiv(cc * (CH / 2), m_n4)];
            else o = lds16[gg * img16 + (cc / CPB) * BCH + (cc % CPB)];
            __builtin_nontemporal_store(o, dst + c);
        }
        __syncthreads();
        g = gn;
    }
    if (tid == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

}  // namespace pf
