// Tile pass with a RUN-TIME mixed-radix plan (round 4): the column / row pass of fft_tile.h for tile lengths L = 2^a 3^b 5^c that the
// register-tiled kernels there do not cover - their lengths are R0 2^b with b >= 3 and R0 in {1, 3, 5, 9, 15, 25, 27, 45}, so a size with few
// factors of two and a large odd part (10800 = 2^4 3^3 5^2, 600000 = 2^6 3 5^5: 76 float / 82 double legal sizes up to 600 000) had no two-pass
// plan and went through the three-to-five streaming sweeps of fft_big.h at 0.08-0.18 of the roofline.
//
// Reference: the same functions as fft_tile.h - one sweep over memory per radix pass for every size the reference accepts (cfftf1_ps,
// src/pffft_priv_impl.h:1004-1048; the accepted sizes: :91-114, tests/test_fft_factors.c:36-61) - reduced to two.
//
// Same tile as fft_tile.h: C = 16 (float) / 8 (double) sequences x L points in ONE LDS image [point][sequence], 16-byte units, rows padded by
// one unit; the same pass descriptor (TileDesc), the same four-step twiddle tables, so that a plan may mix the two families (tile_tu.hip).
// What differs is the stage engine: radices 2 .. 12 (cxmath.h dftR; 15 and 16 would hold 64 registers of operands next to the 28 of the prefetch) in up to five Stockham stages taken from the plan, a stage's work items
// (butterfly j, unit p) dealt to the threads round robin; all operands of a thread are read before the barrier, all results written after
// it (one image: L = 864 is 122 KiB).  One thread per 6.75 image units: 128 threads up to L = 108, 256 up to 216, 512 up to 432, 1024 beyond - a stage then
// holds at most 16 units per thread and the kernels stay within 128 registers.
#pragma once
#include <type_traits>
#include "fft_tile.h"

namespace pf {

constexpr int TG_MAX_STAGES = 5;

struct TileGenPlan {
    int L, ns;
    int R[TG_MAX_STAGES], nb[TG_MAX_STAGES], Ns[TG_MAX_STAGES], tws[TG_MAX_STAGES];   // radix, L / R, product of the earlier radices, L / (Ns R)
    unsigned m_Ns[TG_MAX_STAGES];                                                       // x div Ns = umulhi(x, m) for x < 65536
    unsigned m_L;
};

template <typename T, int WG> struct TileGenGeom {
    static constexpr int PP = 8, S = TileUnit<T>::S, C = PP * S, PITCH = PP + 1;
    static constexpr int LMAX = WG == 1024 ? 864 : WG == 512 ? 432 : WG == 256 ? 216 : 108;
    static constexpr int KU = (LMAX * PP + WG - 1) / WG;          // 16-byte units of the image per thread (7)
    static constexpr int KE = (LMAX * C + WG - 1) / WG;           // elements of a row tile per thread (14 float / 7 double)
    static constexpr int WB = 9;
    __host__ __device__ static constexpr size_t img_bytes(int L) { return (size_t)L * PITCH * 16 + 256; }
    __host__ __device__ static constexpr size_t lds_bytes(int L, int levels) { return img_bytes(L) + ((size_t)L + ((size_t)levels << WB)) * 2 * sizeof(T) + 16; }
};

// one stage: radix R, every thread up to K work items
template <typename T, int WG, int R, int DIR, int PITCH>
__device__ __forceinline__ void tg_stage(typename TileUnit<T>::U* img, const cx<T>* wl, int nb, int Ns, unsigned mNs, int tws, int tid) {
    typedef cx<T> CX;
    typedef TileGenGeom<T, WG> G;
    typedef TileUnit<T> TU;
    typedef typename TU::U U;
    constexpr int K = (G::KU + R - 1) / R, S = G::S;
    const int items = nb * G::PP;
    U op[K][R];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = tid + k * WG;
        if (i < items) {
            const int p = i & 7, j = i >> 3;
#pragma unroll
            for (int q = 0; q < R; ++q) op[k][q] = img[(j + q * nb) * PITCH + p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = tid + k * WG;
        if (i < items) {
            const int p = i & 7, j = i >> 3;
            int jd = j, jm = 0;
            if (Ns > 1) {
                jd = (int)__umulhi((unsigned)j, mNs); jm = j - jd * Ns;
                // operand q times W_(Ns R)^(q jm): the table entry serves the unit's S sequences
                const int step = jm * tws;
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    const CX w = wl[q * step];
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) TU::set(op[k][q], sq, twmul<DIR>(TU::get(op[k][q], sq), w));
                }
            }
#pragma unroll
            for (int sq = 0; sq < S; ++sq) {
                CX o[R];
#pragma unroll
                for (int q = 0; q < R; ++q) o[q] = TU::get(op[k][q], sq);
                dftR<R, DIR>(o);
#pragma unroll
                for (int q = 0; q < R; ++q) TU::set(op[k][q], sq, o[q]);
            }
            const int pbase = jd * Ns * R + jm;
#pragma unroll
            for (int d = 0; d < R; ++d) img[(pbase + d * Ns) * PITCH + p] = op[k][d];
        }
    }
    __syncthreads();
}

// SEQC = 1: pass A (adjacent columns, four-step twiddle)   SEQC = 0: pass B (rows in, transposing store); canonical layouts only
// No streaming (nontemporal) hint on the global accesses: where a stride is not whole 128-byte lines adjacent tiles share lines, and the half
// a tile does not use must stay in L2 for its neighbour (N = 10800 on 108 x 100: 350 / 307 -> 332 / 277 us per pass); where the runs ARE whole
// lines the hint measured no difference (N = 384000, 409600), so there is one kind of kernel
// OINT (row pass of a forward transform) / IINT (column pass of a backward one): the spectrum leaves / arrives in the pffft-internal layout,
// as in fft_tile.h (the layout: SURVEY.md appendix A, src/pffft_priv_impl.h:1195-1237): L and the sequence count of the pass multiples of 4
template <typename T, int WG, int DIR, int SEQC, int OINT = 0, int IINT = 0>
__global__ void __launch_bounds__(WG, 4)
tileg_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, unsigned long long ntiles, TileDesc D, TileGenPlan P, unsigned* ctr) {
    typedef cx<T> CX;
    typedef TileGenGeom<T, WG> G;
    typedef TileUnit<T> TU;
    typedef typename TU::U U;
    constexpr int PP = G::PP, S = G::S, C = G::C, WB = G::WB, KU = G::KU, KE = G::KE;
    constexpr int PITCH = (SEQC && !IINT) ? PP : G::PITCH;        // (a plain column pass is conflict-free without the padding unit: fft_tile.h)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = P.L, tid = threadIdx.x;
    U* img = reinterpret_cast<U*>(smem);
    CX* wl = reinterpret_cast<CX*>(smem + G::img_bytes(L));
    CX* w3 = wl + L;
    for (int i = tid; i < L; i += WG) wl[i] = tile_unit_root<T>((double)i / (double)L);
    const bool lv3 = D.M > (1ull << (2 * WB));
    if (SEQC) {
        const double invM = 1.0 / (double)D.M;
        for (int i = tid; i < ((lv3 ? 3 : 2) << WB); i += WG) {
            const int lvl = i >> WB, m = i & ((1 << WB) - 1);
            w3[i] = tile_unit_root<T>((double)m * (double)(1u << (WB * lvl)) * invM);
        }
    }
    unsigned* s_next = reinterpret_cast<unsigned*>(w3 + ((lv3 ? 3 : 2) << WB));
    const bool dyn = ctr != nullptr;
    const int units = L * PP;
    typedef typename std::conditional<SEQC != 0, U, CX>::type LD;
    constexpr int NLD = SEQC ? KU : KE;
    static_assert(!OINT || (!SEQC && DIR == FWD), "internal layout out: forward row pass");
    static_assert(!IINT || (SEQC && DIR == BWD), "internal layout in: backward column pass");
    constexpr int UPB = 2 * (int)sizeof(T), UPQ = UPB / 4, UPS = 4 / S;   // 16-byte units per block of the layout / per (group, quarter); image units per group of 4
    static_assert((C / 4) * UPB == 32, "32 units of the layout per point row of a tile");
    struct Tile { const CX* src; CX* dst; unsigned col0; int pv; unsigned long long eb; };
    auto tile_of = [&](unsigned long long tile) {
        // tile id = (vec TA + a) TB + b
        const unsigned b = (unsigned)(tile % D.TB);
        const unsigned long long rest = tile / D.TB;
        const unsigned a = (unsigned)(rest % D.TA);
        const unsigned long long vec = rest / D.TA;
        Tile t;
        t.pv = (D.last_units && a == D.TA - 1) ? (int)D.last_units : PP;          // 16-byte sequence units that exist
        // (internal layout: eb = canonical index, inside its vector, of the tile's first element; src / dst the vector's base)
        t.eb = 0;
        if constexpr (IINT) { t.eb = a * D.in_a + b * D.in_b; t.src = in + vec * D.vstride; }
        else t.src = in + vec * D.vstride + a * D.in_a + b * D.in_b;
        if constexpr (OINT) { t.eb = a * D.out_a + b * D.out_b; t.dst = out + vec * D.vstride; }
        else t.dst = out + vec * D.vstride + a * D.out_a + b * D.out_b;
        t.col0 = a * D.col_a + b * D.col_b;
        return t;
    };
    // (tid is re-read through an empty asm per phase: the index arithmetic of the load, exchange and store loops is the same for every tile,
    //  and hoisted out of the tile loop it lived in scratch - 37 eight-byte reloads per tile and lane in the first build)
    auto issue_loads = [&](const Tile& t, LD (&r)[NLD]) {
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        if constexpr (IINT) {
            // per point row n1' < L/4 the four quarters of the tile's columns are a run of C/4 whole blocks: 32 dense 16-byte units
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int g = tid_l + k * WG, ptq = g >> 5, rr = g & 31, bb = rr / UPB;
                if (g < units && bb * 4 < t.pv * S) {
                    const U* gp = reinterpret_cast<const U*>(t.src + 4 * (t.eb + (unsigned long long)ptq * D.ips + 4 * bb)) + rr % UPB;
                    r[k] = *gp;
                }
            }
        } else if constexpr (SEQC) {
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int i = tid_l + k * WG, pt = i >> 3, pu = i & 7;
                if (i < units && pu < t.pv) {
                    const U* gp = reinterpret_cast<const U*>(t.src + (unsigned long long)pt * D.ips + S * pu);
                    r[k] = *gp;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KE; ++k) {
                const int g = tid_l + k * WG, seq = (int)__umulhi((unsigned)g, P.m_L), pt = g - seq * L;
                if (g < C * L && seq < t.pv * S) {
                    const CX* gp = t.src + (unsigned long long)seq * D.iss + pt;
                    r[k] = *gp;
                }
            }
        }
    };
    // Tiles in order from the counter (ctr != nullptr, tiles of 60 KiB and more) or on a static stride; the grab runs two tiles ahead so that
    // the prefetch knows its tile (the protocol of fft_tile.h)
    // static stride: workgroup b runs on XCD b mod 8 (round-robin dispatch); XCD x takes the CONTIGUOUS tiles x per .. (x + 1) per - 1 of every
    // sweep of the grid, so that tiles which share 128-byte lines (a stride between points that is not a multiple of 16 elements) meet in
    // one L2 (TileDesc::xmode bit 0; off: tile = workgroup index)
    const unsigned per = (gridDim.x + 7) / 8;
    const bool xmap = (D.xmode & 1u) != 0;
    const unsigned long long first = xmap ? (unsigned long long)(blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    const unsigned long long sweep = xmap ? 8ull * per : gridDim.x;
    unsigned long long tile = first, tile1 = first + sweep;
    unsigned pend = 0;
    // in order from the counter: ONE counter per XCD (workgroup b runs on XCD b mod 8), each over a contiguous eighth of the tiles, so that
    // the tiles in flight on an XCD are neighbours and the 128-byte lines two of them share are fetched through one L2 (ctr[0 .. 7] next,
    // ctr[8] done; TileDesc::xmode bit 1, off: one counter for all)
    const bool xctr = dyn && (D.xmode & 2u);
    const unsigned long long xper = (ntiles + 7) / 8, xbase = xctr ? (blockIdx.x % 8) * xper : 0;
    const unsigned long long xend = xctr ? (xbase + xper < ntiles ? xbase + xper : ntiles) : ntiles;
    unsigned* cnext = ctr + (xctr ? blockIdx.x % 8 : 0);
    // the first two tiles of a workgroup are static (its index among the workgroups of its counter, and that plus their number); the counter
    // hands out what follows (fft_tile.h: three start-up grabs per workgroup left a short launch unbalanced)
    const unsigned long long g0 = xctr ? gridDim.x / 8 : gridDim.x, lid = xctr ? blockIdx.x / 8 : blockIdx.x;
    auto ranged = [&](unsigned long long local) -> unsigned long long { const unsigned long long t = xbase + local; return t < xend ? t : ntiles; };
    auto grabbed = [&](unsigned v) -> unsigned long long { return ranged(2 * g0 + v); };
    if (dyn) {
        tile = ranged(lid); tile1 = ranged(lid + g0);
        if (tid == 0) pend = atomicAdd(cnext, 1u);
    }
    __syncthreads();
    LD nxt[NLD];
    if (tile < ntiles) issue_loads(tile_of(tile), nxt);
    for (unsigned it = 0; tile < ntiles; ++it) {
        if (dyn && tid == 0) { s_next[it & 1] = pend; pend = atomicAdd(cnext, 1u); }   // the tile after the next one: read after this tile's barriers
        const Tile tl = tile_of(tile);
        const int pv = tl.pv;
        CX* dst = tl.dst;
        const unsigned col0 = tl.col0;
        {
            int tid_w = tid;
            asm volatile("" : "+v"(tid_w));
            if constexpr (IINT) {
                // (re group, im group) units of the layout -> (re, im) sequence units of the image: a lane ^ 1 (float) / lane ^ 2 (double) exchange
#pragma unroll
                for (int k = 0; k < KU; ++k) {
                    const int g = tid_w + k * WG, ptq = g >> 5, rr = g & 31, bb = rr / UPB, m = (rr / UPQ) % 4, sub = rr % UPQ;
                    if (g < units && bb * 4 < pv * S) {
                        U* dp = img + (ptq + m * (L >> 2)) * PITCH + bb * UPS;
                        const U x = nxt[k];
                        if constexpr (S == 2) {           // sub = part: even lane re0..3, odd lane im0..3 -> sequences (0, 1) / (2, 3)
                            const T s0 = dpp_xor1(sub ? x.x : x.z), s1 = dpp_xor1(sub ? x.y : x.w);
                            U o;
                            if (sub) { o.x = s0; o.y = x.z; o.z = s1; o.w = x.w; }
                            else { o.x = x.x; o.y = s0; o.z = x.y; o.w = s1; }
                            dp[sub] = o;
                        } else {                          // sub = 2 part + (l / 2): lanes (re01, re23, im01, im23) -> sequences 0, 2, 1, 3
                            const T sv = dpp_xor2(sub >> 1 ? x.x : x.y);
                            U o;
                            if (sub >> 1) { o.x = sv; o.y = x.y; }
                            else { o.x = x.x; o.y = sv; }
                            dp[2 * (sub & 1) + (sub >> 1)] = o;
                        }
                    }
                }
            } else if constexpr (SEQC) {
#pragma unroll
                for (int k = 0; k < KU; ++k) {
                    const int i = tid_w + k * WG, pt = i >> 3, pu = i & 7;
                    if (i < units && pu < pv) img[pt * PITCH + pu] = nxt[k];
                }
            } else {
                CX* imgc = reinterpret_cast<CX*>(img);
#pragma unroll
                for (int k = 0; k < KE; ++k) {
                    const int g = tid_w + k * WG, seq = (int)__umulhi((unsigned)g, P.m_L), pt = g - seq * L;
                    if (g < C * L && seq < pv * S) imgc[pt * (PITCH * S) + seq] = nxt[k];
                }
            }
        }
        // the next tile's loads fly through the stages of this one
        if (tile1 < ntiles) issue_loads(tile_of(tile1), nxt);
        __syncthreads();
        for (int s = 0; s < P.ns; ++s) {
            const int nb = P.nb[s], Ns = P.Ns[s], tws = P.tws[s];
            const unsigned mNs = P.m_Ns[s];
            switch (P.R[s]) {
                case 2: tg_stage<T, WG, 2, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 3: tg_stage<T, WG, 3, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 4: tg_stage<T, WG, 4, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 5: tg_stage<T, WG, 5, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 6: tg_stage<T, WG, 6, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 8: tg_stage<T, WG, 8, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 9: tg_stage<T, WG, 9, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 10: tg_stage<T, WG, 10, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                case 12: tg_stage<T, WG, 12, DIR, PITCH>(img, wl, nb, Ns, mNs, tws, tid); break;
                default: break;
            }
        }
        // the image holds the spectrum [k][sequence]: runs of C adjacent sequences per point.  Column pass: point k of column col times the
        // four-step twiddle W_M^(k col) on the way out - a thread's units are one unit column pu and the points tid / 8 + k WG / 8: two table
        // products per sequence and tile, one multiplication per step
        int tid_s = tid;
        asm volatile("" : "+v"(tid_s));
        CX ftw[S], fst[S];
        if constexpr (SEQC) {
#pragma unroll
            for (int sq = 0; sq < S; ++sq) {
                const unsigned col = col0 + (unsigned)(S * (tid_s & 7) + sq);
                ftw[sq] = tile_w3<WB>(w3, (unsigned)(tid_s >> 3) * col, lv3);
                fst[sq] = tile_w3<WB>(w3, (unsigned)(WG / 8) * col, lv3);
            }
        }
        if constexpr (OINT) {
            // one item = ONE 16-byte unit of the layout, consecutive lanes consecutive units; the units of a point k' < L/4 are a run of C/4 blocks
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int g = tid_s + k * WG, ptq = g >> 5, r = g & 31, bb = r / UPB, m = (r / UPQ) % 4, sub = r % UPQ;
                if (g < units && bb * 4 < pv * S) {
                    const U* sp = img + (ptq + m * (L >> 2)) * PITCH + bb * UPS;
                    U* gp = reinterpret_cast<U*>(dst + 4 * (tl.eb + (unsigned long long)ptq * D.ops + 4 * bb)) + r % UPB;
                    U o;
                    if constexpr (S == 2) {           // sub = part p: (re or im) of the four sequences of the group
                        const U u0 = sp[0], u1 = sp[1];
                        if (sub) { o.x = u0.y; o.y = u0.w; o.z = u1.y; o.w = u1.w; }
                        else { o.x = u0.x; o.y = u0.z; o.z = u1.x; o.w = u1.z; }
                    } else {                          // sub = 2 p + (l / 2): part p of sequences 2 (l/2), 2 (l/2) + 1
                        const U u0 = sp[2 * (sub & 1)], u1 = sp[2 * (sub & 1) + 1];
                        if (sub >> 1) { o.x = u0.y; o.y = u1.y; }
                        else { o.x = u0.x; o.y = u1.x; }
                    }
                    *gp = o;
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const int i = tid_s + k * WG, pt = i >> 3, pu = i & 7;
            if (i < units) {
                U x = img[pt * PITCH + pu];
                if constexpr (SEQC) {
#pragma unroll
                    for (int sq = 0; sq < S; ++sq) {
                        TU::set(x, sq, twmul<DIR>(TU::get(x, sq), ftw[sq]));
                        ftw[sq] = cmul(ftw[sq], fst[sq]);
                    }
                }
                if (pu < pv) {
                    U* gp = reinterpret_cast<U*>(dst + (unsigned long long)pt * D.ops + S * pu);
                    *gp = x;
                }
            }
        }
        }
        const unsigned long long tile2 = dyn ? grabbed(s_next[it & 1]) : tile1 + sweep;
        __syncthreads();
        tile = tile1; tile1 = tile2;
    }
    if (dyn && tid == 0) {
        __threadfence();
        const unsigned dn = atomicAdd(&ctr[xctr ? 8 : 1], 1u);
        if (dn == gridDim.x - 1) {
            if (xctr) { for (int i = 0; i < 9; ++i) atomicExch(&ctr[i], 0u); }
            else { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
        }
    }
}

}  // namespace pf
