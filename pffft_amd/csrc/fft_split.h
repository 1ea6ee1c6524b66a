// Real forward transform of N = 16384 floats (n = 8192 packed complex points) with WAVE-LOCAL middle stages — BASELINE
// configs[2] (C3).  Same reference functions as fft_tiled.h (rfftf1_ps + radf* passes, src/pffft_priv_impl.h:809-854,
// :469-550; pffft_real_finalize :1330-1372; zreorder :1158-1193 folded in), one pass over HBM per vector.
//
// Why: the register-tiled kernel moves 256 threads through every stage in lock step (exchange, barrier, butterflies,
// barrier ...), so VALU, LDS and the wait for HBM add up instead of overlapping; the FIR block kernel showed that the
// same arithmetic organised as ONE WAVEFRONT per 1024-point transform — wave-local exchanges, no workgroup barrier —
// runs at more than twice the instruction rate.  n = 8 x 1024 (decimation in frequency):
//   A   all 8 wavefronts: radix-8 butterflies over z[j + 1024 q], straight from HBM (prefetched registers), times
//       W_n^(j d), into row d of the LDS image                                             -> ONE workgroup barrier
//   B   wavefront d: 1024-point transform of row d (radix 8 x 16 x 8, two wave-local exchanges), Z[d + 8 k2] back
//       into row d in natural order                                                        -> ONE workgroup barrier
//   C   thread T gathers the four runs of four bins that make up block T of the pffft-internal layout together with
//       their mirrors — Z[t], Z[n-t], Z[n/2-t], Z[n/2+t], t = 4T .. 4T+3 (bin 0 / n/2 / n/4 / 3n/4 for T = 0) —, runs the
//       pair pass in registers and holds one whole 128-byte block of the internal layout (or four 32-byte runs of the
//       canonical spectrum): no layout exchange; a wave-local transposition through LDS makes the stores coalesced.
// Four workgroup barriers per transform instead of six, and between (2) and (3) — two thirds of the arithmetic — every
// wavefront runs on its own.
// MEASURED (MI355X, batch 2^16, tools/split_check.py): 0.69-0.71 of the roofline with register prefetch (0.53 without),
// against 0.70-0.73 for the three-stage register-tiled kernel on the same boxes: on par, not better — stage C's scattered
// 8-byte LDS gathers and the store transposition cost what the lock-step phases cost the other kernel.  It stays an opt-in
// variant (pffft_hip_set_variant(89)); parity: tests/test_gpu_round2.py.
#pragma once
#include "fft_tiled.h"

namespace pf {

struct SplitC3 {
    typedef TiledCfg<float, 10, 64, 3, 8, 16, 8, 1, 4, 4, 3, 0, 512, 1> Sub;   // the wave-local 1024-point transform
    static constexpr int n = 8192, R0 = 8, M = 1024, WAVES = 8, WG = 512;
    static constexpr int ROW = Sub::IMG;                                  // complex points per wavefront region (row + exchange paddings)
    static constexpr size_t LDS_BYTES = (size_t)WAVES * ROW * 8 + 64;
};

// flags bit1: output in the internal layout (else canonical half-complex spectrum)
template <int PREFETCH, int OCC>
__global__ void __launch_bounds__(SplitC3::WG, OCC)
fft_split_real_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned batch, int flags,
                          const cx<float>* __restrict__ twn,      // W_n^j, j < n
                          const cx<float>* __restrict__ tw1024,   // W_1024^j
                          const cx<float>* __restrict__ twr,      // W_N^k, k <= n/2, N = 2n
                          unsigned* ctr) {
    typedef float T;
    typedef cx<T> CX;
    typedef SplitC3 S;
    typedef Tiled<S::Sub, FWD, 0> K;
    constexpr int n = S::n, M = S::M, ROW = S::ROW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    CX* lds = reinterpret_cast<CX*>(smem_raw);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem_raw + (size_t)S::WAVES * ROW * 8);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool out_int = flags & 2;

    // ---- per-thread constants
    typename K::Tw w;                                  // base twiddles of the wave-local transform
    K::load_tw(w, lane, tw1024, nullptr);
    const CX wa0 = twn[2 * tid], wa1 = twn[2 * tid + 1];   // W_n^j of this thread's two stage-A butterflies, j = 2 tid + u
    // pair pass: t = 4 tid + jj;  wq0[jj] = W_N^t (bins t, n - t);  wq1[jj] = W_N^(n/2 - t) (bins n/2 - t, n/2 + t)
    CX wq0[4], wq1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int t = 4 * tid + jj;
        wq0[jj] = twr[t];
        wq1[jj] = twr[n / 2 - t];
    }
    if (tid == 0) wq1[0] = twr[n / 4];                  // block 0: the pair (n/4, 3n/4) takes the place of (n/2 - 0, n/2 + 0)

    const bool dyn = ctr != nullptr;
    unsigned pend = 0, g = blockIdx.x;
    if (dyn && tid == 0) {
        s_next[0] = atomicAdd(&ctr[0], 1u);
        pend = atomicAdd(&ctr[0], 1u);
    }
    __syncthreads();
    if (dyn) g = s_next[0];
    const unsigned last = batch - 1;
    typedef chunk16 C16;
    // stage-A operands: 16-byte chunk tid + 512 q of the vector = packed points z[2 tid + 1024 q], z[2 tid + 1 + 1024 q]
    auto load_vec = [&](unsigned vec, C16 (&r)[8]) {
        const C16* src = reinterpret_cast<const C16*>(in + (size_t)(vec < last ? vec : last) * 2 * n);
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = __builtin_nontemporal_load(src + tid + 512 * q);
    };
    C16 raw[8];
    load_vec(g, raw);
    for (unsigned it = 0; g < batch; ++it) {
        if (dyn && tid == 0) {
            s_next[(it + 1) & 1] = pend;
            pend = atomicAdd(&ctr[0], 1u);
        }
        // ================= A: radix-8 across the wavefronts
        CX a0[8], a1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0[q] = mk<T>(raw[q].x, raw[q].y); a1[q] = mk<T>(raw[q].z, raw[q].w); }
        dft8<FWD>(a0);
        dft8<FWD>(a1);
        {
            // W_n^(j d), d = 1 .. 7, from W_n^j by products at most three deep
            CX p0[8], p1[8];
            p0[1] = wa0; p1[1] = wa1;
            asm volatile("" : "+v"(p0[1].x), "+v"(p0[1].y), "+v"(p1[1].x), "+v"(p1[1].y));
            p0[2] = cmul(p0[1], p0[1]); p0[3] = cmul(p0[2], p0[1]); p0[4] = cmul(p0[2], p0[2]); p0[5] = cmul(p0[4], p0[1]);
            p0[6] = cmul(p0[3], p0[3]); p0[7] = cmul(p0[4], p0[3]);
            p1[2] = cmul(p1[1], p1[1]); p1[3] = cmul(p1[2], p1[1]); p1[4] = cmul(p1[2], p1[2]); p1[5] = cmul(p1[4], p1[1]);
            p1[6] = cmul(p1[3], p1[3]); p1[7] = cmul(p1[4], p1[3]);
#pragma unroll
            for (int d = 1; d < 8; ++d) { a0[d] = cmul(a0[d], p0[d]); a1[d] = cmul(a1[d], p1[d]); }
        }
        __syncthreads();                               // (1) the rows are free: everyone finished stage C of the vector before
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st2(lds + (size_t)d * ROW + 2 * tid, a0[d], a1[d]);
        __syncthreads();                               // (2) rows complete; also publishes s_next
        const unsigned gn = dyn ? s_next[(it + 1) & 1] : g + gridDim.x;
        if (PREFETCH) load_vec(gn, raw);               // the next vector flies while this one is transformed
        // ================= B: wavefront `wave` transforms row `wave` (1024 points), wave-local
        CX* row = lds + (size_t)wave * ROW;
        CX v[16];
        {
            const C16* r16 = reinterpret_cast<const C16*>(row);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const C16 c = r16[lane + 64 * q];      // points 2 lane + 128 q, + 1: operands q of butterflies 2 lane, 2 lane + 1
                v[q] = mk<T>(c.x, c.y);
                v[8 + q] = mk<T>(c.z, c.w);
            }
        }
        K::xsync();
        K::template butterflies<0>(v, lane, w, tw1024);
        K::template xwrite<0>(v, lane, row); K::xsync();
        K::template xread<0>(v, lane, row); K::xsync();
        K::template butterflies<1>(v, lane, w, tw1024);
        K::template xwrite<1>(v, lane, row); K::xsync();
        K::template xread<1>(v, lane, row); K::xsync();
        K::template butterflies<2>(v, lane, w, tw1024);
        // Z[wave + 8 k2], k2 = 2 lane + e + 128 dd  ->  row[k2] (natural order)
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) lds_st2(row + 2 * lane + 128 * dd, v[dd], v[8 + dd]);
        __syncthreads();                               // (3) the whole packed spectrum Z sits in the rows: Z[k] at row k mod 8, column k div 8
        // ================= C: pair pass on the four runs of block `tid`
        auto Zat = [&](int k) -> CX { return lds[(size_t)(k & 7) * ROW + (k >> 3)]; };
        CX x0[4], x1[4], x2[4], x3[4];                 // quarters of the internal layout: bins t, n/2 - t, n/2 + t, n - t
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int t = 4 * tid + jj;
            const bool special = (jj == 0) && (tid == 0);
            const CX A0 = Zat(t), B0 = Zat((n - t) & (n - 1));                      // t = 0: both bin 0
            const CX A1 = Zat(special ? n / 4 : n / 2 - t), B1 = Zat(special ? 3 * n / 4 : n / 2 + t);
            const typename K::Pair r0 = K::pair1(A0, B0, wq0[jj]);
            const typename K::Pair r1 = K::pair1(A1, B1, wq1[jj]);
            x0[jj] = r0.a; x3[jj] = r0.b; x1[jj] = r1.a; x2[jj] = r1.b;
            if (jj == 0) {
                // block 0: bin 0 = (DC, Nyquist), bin n/2 = conj(Z[n/2]); the pair (n/4, 3n/4) went through r1
                const CX z0 = A0, zh = Zat(n / 2);
                const CX dc = mk<T>(z0.x + z0.y, z0.x - z0.y), hf = conj(zh);
                x0[0] = K::sel(special, dc, x0[0]);
                x2[0] = K::sel(special, hf, x2[0]);
                x1[0] = K::sel(special, r1.a, x1[0]);
                x3[0] = K::sel(special, r1.b, x3[0]);
            }
        }
        const bool active = g < batch;
        float* dst = out + (size_t)(active ? g : last) * 2 * n;
        __syncthreads();                               // (4) everyone has its bins: the rows may be reused for the store transposition
        C16* st16 = reinterpret_cast<C16*>(lds) + (size_t)wave * (64 * 9);   // wave-local: 64 blocks x 8 units, pitch 9 units
        if (out_int) {
            // block tid of the internal layout: units [q0 re][q0 im][q1 re][q1 im][q2 re][q2 im][q3 re][q3 im]
            C16 u[8];
            u[0] = C16{x0[0].x, x0[1].x, x0[2].x, x0[3].x}; u[1] = C16{x0[0].y, x0[1].y, x0[2].y, x0[3].y};
            u[2] = C16{x1[0].x, x1[1].x, x1[2].x, x1[3].x}; u[3] = C16{x1[0].y, x1[1].y, x1[2].y, x1[3].y};
            u[4] = C16{x2[0].x, x2[1].x, x2[2].x, x2[3].x}; u[5] = C16{x2[0].y, x2[1].y, x2[2].y, x2[3].y};
            u[6] = C16{x3[0].x, x3[1].x, x3[2].x, x3[3].x}; u[7] = C16{x3[0].y, x3[1].y, x3[2].y, x3[3].y};
#pragma unroll
            for (int i = 0; i < 8; ++i) st16[lane * 9 + i] = u[i];
            K::xsync();
            // the wavefront's 64 blocks = 8 KiB contiguous in HBM: store instruction i covers units 64 i .. 64 i + 63
            C16* d16 = reinterpret_cast<C16*>(dst) + (size_t)wave * 512;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int uidx = 64 * i + lane;        // unit index inside the wavefront's 8 KiB
                const C16 val = st16[(uidx >> 3) * 9 + (uidx & 7)];
                if (active) __builtin_nontemporal_store(val, d16 + uidx);
            }
        } else {
            // canonical: bins t .. t+3 ascending at 4 tid, n/2 - t descending, n/2 + t ascending, n - t descending
            // (bin 0 carries (DC, Nyquist)); as 16-byte units of two bins
            C16* d16 = reinterpret_cast<C16*>(dst);
            const int t = 4 * tid;
            auto st2 = [&](int bin, CX a, CX b) { if (active) __builtin_nontemporal_store(C16{a.x, a.y, b.x, b.y}, d16 + (bin >> 1)); };
            st2(t, x0[0], x0[1]); st2(t + 2, x0[2], x0[3]);
            st2(n / 2 + t, x2[0], x2[1]); st2(n / 2 + t + 2, x2[2], x2[3]);
            // descending runs: bins n/2 - t - 3 .. n/2 - t and n - t - 3 .. n - t; their first element (jj = 0) belongs to the
            // 16-byte unit of the NEXT lower thread's run: write 8-byte bins
            CX* d8 = reinterpret_cast<CX*>(dst);
            if (active) {
                if (tid != 0) { d8[n / 2 - t] = x1[0]; d8[n - t] = x3[0]; }
                else { d8[n / 4] = x1[0]; d8[3 * n / 4] = x3[0]; }
#pragma unroll
                for (int jj = 1; jj < 4; ++jj) { d8[n / 2 - t - jj] = x1[jj]; d8[n - t - jj] = x3[jj]; }
            }
        }
        g = gn;
        if (!PREFETCH) load_vec(g, raw);
    }
    if (dyn && tid == 0) {
        __threadfence();
        const unsigned dn = atomicAdd(&ctr[1], 1u);
        if (dn == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

}  // namespace pf
