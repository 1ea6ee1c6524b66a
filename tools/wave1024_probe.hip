// VERDICT r04 4(b), probe: the wave-local 1024-point transforms of the split FIR kernel (fft_split.h, phases B and B') with their two
// exchanges per transform IN REGISTERS - lane-bit / register-bit swaps by v_permlane32_swap / v_permlane16_swap (lane distance 32 / 16), DPP
// row shifts under bank masks (8 / 4) and DPP quad permutations + selects (2 / 1) - against the same transforms on the LDS exchanges of the
// register-tiled engine (fft_tiled.h), in the FIR kernel's setting: eight wavefronts per workgroup, one workgroup per CU, every wavefront:
// operands from its LDS row, forward transform, spectrum to the row in natural order + mirror read (phase M), inverse transform, result to the row.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pffft_amd/csrc tools/wave1024_probe.hip -o tools/_bin/wave1024_probe
// 1024 = 8 x 16 x 8, decimation in frequency; point n = u + 2 lane + 128 q sits in register (u, q) of lane `lane`:
//   S1  radix 8 over q (registers), times W_1024^(ka (u + 2 lane))
//   T1  lane bits 5, 4, 3, 2 <-> register bits (u, ka2, ka1, ka0): the registers now index m[6:3] of the remaining 128-point transforms
//   S2  radix 16 over the registers, times W_128^(kb mlo), mlo = 4 lane1 + 2 lane0 + lane5
//   T2  lane bits 1, 0, 5 <-> register bits kb2, kb1, kb0
//   S3  radix 8 over the registers (twice: kb3 = 0, 1)
// bin k2 = ka + 8 kb + 128 kc ends in register (kb3, kc) of the lane with bits (5: kb0, 4..2: ka, 1: kb2, 0: kb1).  The inverse runs the same
// steps backwards with conjugated twiddles.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "fft_split.h"
using namespace pf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef cx<float> C;

__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float bitsf(unsigned u) { return __builtin_bit_cast(float, u); }

// lane bit J <-> the register bit that tells a from b: a's lanes with bit J set trade places with b's lanes with bit J clear
template <int J> __device__ __forceinline__ void bitswap(float& a, float& b, int lane) {
    if constexpr (J == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(fbits(a), fbits(b), false, false);
        a = bitsf(r[0]); b = bitsf(r[1]);
    } else if constexpr (J == 4) {
        auto r = __builtin_amdgcn_permlane16_swap(fbits(a), fbits(b), false, false);
        a = bitsf(r[0]); b = bitsf(r[1]);
    } else if constexpr (J == 3) {
        const int t = (int)fbits(b);
        const int nb = __builtin_amdgcn_update_dpp((int)fbits(b), (int)fbits(a), 0x108 /* row_shl:8 */, 0xf, 0x3, false);
        const int na = __builtin_amdgcn_update_dpp((int)fbits(a), t, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
        a = bitsf((unsigned)na); b = bitsf((unsigned)nb);
    } else if constexpr (J == 2) {
        const int t = (int)fbits(b);
        const int nb = __builtin_amdgcn_update_dpp((int)fbits(b), (int)fbits(a), 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        const int na = __builtin_amdgcn_update_dpp((int)fbits(a), t, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
        a = bitsf((unsigned)na); b = bitsf((unsigned)nb);
    } else {
        constexpr int QP = J == 1 ? 0x4e /* [2,3,0,1] */ : 0xb1 /* [1,0,3,2] */;
        const int xa = __builtin_amdgcn_update_dpp(0, (int)fbits(a), QP, 0xf, 0xf, false);     // a of lane ^ (1 << J)
        const int xb = __builtin_amdgcn_update_dpp(0, (int)fbits(b), QP, 0xf, 0xf, false);
        const bool hi = (lane >> J) & 1;
        const float na = hi ? bitsf((unsigned)xb) : a, nb = hi ? b : bitsf((unsigned)xa);
        a = na; b = nb;
    }
}
template <int J> __device__ __forceinline__ void bitswap(C& a, C& b, int lane) {
    float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    bitswap<J>(ax, bx, lane); bitswap<J>(ay, by, lane);
    a = mk<float>(ax, ay); b = mk<float>(bx, by);
}
// over all register pairs that differ in bit RB of the 4-bit register index
template <int J, int RB> __device__ __forceinline__ void bitswap_all(C (&v)[16], int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (!(i & (1 << RB))) bitswap<J>(v[i], v[i | (1 << RB)], lane);
}

struct RegTw { C w1[7][2]; C w2[15]; };
__device__ __forceinline__ void reg_load_tw(RegTw& w, int lane, const C* __restrict__ tw1024) {
#pragma unroll
    for (int ka = 1; ka < 8; ++ka)
#pragma unroll
        for (int u = 0; u < 2; ++u) w.w1[ka - 1][u] = tw1024[(ka * (u + 2 * lane)) & 1023];
    // after T1 lane bit 5 of THIS lane's data is u of ... the lane that sent it: the twiddle of stage 2 belongs to the data, i.e. to the
    // lane that holds it after T1: mlo = 4 lane1 + 2 lane0 + lane5
    const int mlo = 4 * ((lane >> 1) & 1) + 2 * (lane & 1) + ((lane >> 5) & 1);
#pragma unroll
    for (int kb = 1; kb < 16; ++kb) w.w2[kb - 1] = tw1024[(8 * kb * mlo) & 1023];
}

// register index: v[u * 8 + q]: bit 3 = u, bits 2..0 = q
template <int DIR> __device__ __forceinline__ void reg_fft1024(C (&v)[16], int lane, const RegTw& w) {
    if constexpr (DIR == FWD) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            C b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = v[u * 8 + q];
            dft8<FWD>(b);
            v[u * 8] = b[0];
#pragma unroll
            for (int ka = 1; ka < 8; ++ka) v[u * 8 + ka] = cmul(b[ka], w.w1[ka - 1][u]);
        }
        bitswap_all<5, 3>(v, lane); bitswap_all<4, 2>(v, lane); bitswap_all<3, 1>(v, lane); bitswap_all<2, 0>(v, lane);   // T1
        dft16<FWD>(v);
#pragma unroll
        for (int kb = 1; kb < 16; ++kb) v[kb] = cmul(v[kb], w.w2[kb - 1]);
        bitswap_all<1, 2>(v, lane); bitswap_all<0, 1>(v, lane); bitswap_all<5, 0>(v, lane);                               // T2
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            C b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = v[h * 8 + q];
            dft8<FWD>(b);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[h * 8 + q] = b[q];
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            C b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = v[h * 8 + q];
            dft8<BWD>(b);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[h * 8 + q] = b[q];
        }
        bitswap_all<5, 0>(v, lane); bitswap_all<0, 1>(v, lane); bitswap_all<1, 2>(v, lane);                               // T2 back
#pragma unroll
        for (int kb = 1; kb < 16; ++kb) v[kb] = cmulc(v[kb], w.w2[kb - 1]);
        dft16<BWD>(v);
        bitswap_all<2, 0>(v, lane); bitswap_all<3, 1>(v, lane); bitswap_all<4, 2>(v, lane); bitswap_all<5, 3>(v, lane);   // T1 back
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            C b[8];
            b[0] = v[u * 8];
#pragma unroll
            for (int ka = 1; ka < 8; ++ka) b[ka] = cmulc(v[u * 8 + ka], w.w1[ka - 1][u]);
            dft8<BWD>(b);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[u * 8 + q] = b[q];
        }
    }
}
// the bin k2 = ka + 8 kb + 128 kc of register i = (kb3, kc) in this lane
__device__ __forceinline__ int reg_bin(int i, int lane) {
    const int ka = (lane >> 2) & 7, kb = ((i >> 3) << 3) | (((lane >> 1) & 1) << 2) | ((lane & 1) << 1) | ((lane >> 5) & 1), kc = i & 7;
    return ka + 8 * kb + 128 * kc;
}

typedef SplitFirT<8> S;
typedef Tiled<typename S::Sub, FWD, 0> KF;
typedef Tiled<typename S::Sub, BWD, 0> KB;

// MODE 0: the LDS exchanges of the engine (what fft_split.h runs today); MODE 1: exchanges in registers.  `iters` rounds of
// {load, forward, phase-M store + mirror read, inverse, store}; the last forward spectrum and the last result go to global memory.
template <int MODE>
__global__ void __launch_bounds__(512, 2) probe(const C* __restrict__ x, C* __restrict__ spec, C* __restrict__ back, const C* __restrict__ tw1024, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROW = S::ROW;
    C* rows = reinterpret_cast<C*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    C* row = rows + (size_t)wave * ROW;
    const C* src = x + ((size_t)blockIdx.x * 8 + wave) * 1024;
    for (int i = lane; i < 1024; i += 64) row[i] = src[i];
    typename KF::Tw wf; typename KB::Tw wb;
    RegTw rw;
    if constexpr (MODE == 0) { KF::load_tw(wf, lane, tw1024, nullptr); KB::load_tw(wb, lane, tw1024, nullptr); }
    else reg_load_tw(rw, lane, tw1024);
    __syncthreads();
    C v[16];
    for (int it = 0; it < iters; ++it) {
        {
            const chunk16* r16 = reinterpret_cast<const chunk16*>(row);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const chunk16 c = r16[lane + 64 * q];
                v[q] = mk<float>(c.x, c.y);
                v[8 + q] = mk<float>(c.z, c.w);
            }
        }
        KF::xsync();
        C zm[16];
        if constexpr (MODE == 0) {
            KF::template butterflies<0>(v, lane, wf, tw1024);
            KF::template xwrite<0>(v, lane, row); KF::xsync();
            KF::template xread<0>(v, lane, row); KF::xsync();
            KF::template butterflies<1>(v, lane, wf, tw1024);
            KF::template xwrite<1>(v, lane, row); KF::xsync();
            KF::template xread<1>(v, lane, row); KF::xsync();
            KF::template butterflies<2>(v, lane, wf, tw1024);
#pragma unroll
            for (int d = 0; d < 8; ++d) lds_st2(row + 2 * lane + 128 * d, v[d], v[8 + d]);
            KF::xsync();
#pragma unroll
            for (int d = 0; d < 8; ++d) {                       // (the mirror read of phase M, here from the wave's own row)
                const vec4<float> m = lds_ld2(row + 1022 - 2 * lane - 128 * d);
                zm[8 + d] = mk<float>(m.x, m.y); zm[d] = mk<float>(m.z, m.w);
            }
            KF::xsync();
        } else {
            reg_fft1024<FWD>(v, lane, rw);
#pragma unroll
            for (int i = 0; i < 16; ++i) lds_st(row + reg_bin(i, lane), v[i]);
            KF::xsync();
#pragma unroll
            for (int i = 0; i < 16; ++i) zm[i] = lds_ld(row + ((1024 - reg_bin(i, lane)) & 1023));
            KF::xsync();
        }
        if (it == iters - 1) {
            C* sp = spec + ((size_t)blockIdx.x * 8 + wave) * 1024;
            KF::xsync();
            for (int i = lane; i < 1024; i += 64) sp[i] = row[i];
            KF::xsync();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = mk<float>(v[i].x + 1e-30f * zm[i].x, v[i].y + 1e-30f * zm[i].y);   // keep the mirror read alive
        if constexpr (MODE == 0) {
            KB::template butterflies<0>(v, lane, wb, tw1024);
            KB::template xwrite<0>(v, lane, row); KB::xsync();
            KB::template xread<0>(v, lane, row); KB::xsync();
            KB::template butterflies<1>(v, lane, wb, tw1024);
            KB::template xwrite<1>(v, lane, row); KB::xsync();
            KB::template xread<1>(v, lane, row); KB::xsync();
            KB::template butterflies<2>(v, lane, wb, tw1024);
        } else {
            reg_fft1024<BWD>(v, lane, rw);
        }
        const float sc = 1.0f / 1024.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) lds_st2(row + 2 * lane + 128 * d, mk<float>(v[d].x * sc, v[d].y * sc), mk<float>(v[8 + d].x * sc, v[8 + d].y * sc));
        KF::xsync();
    }
    C* bk = back + ((size_t)blockIdx.x * 8 + wave) * 1024;
    for (int i = lane; i < 1024; i += 64) bk[i] = row[i];
}

int main() {
    const int cus = 256, nvec = cus * 8, iters = 200;
    std::vector<std::complex<float>> hx((size_t)nvec * 1024), htw(1024);
    srand(5);
    for (auto& v : hx) v = std::complex<float>(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    for (int j = 0; j < 1024; ++j) { const long double a = -2.0L * 3.14159265358979323846264338327950288L * j / 1024.0L; htw[j] = std::complex<float>((float)cosl(a), (float)sinl(a)); }
    C *dx, *dspec, *dback, *dtw;
    CK(hipMalloc((void**)&dx, hx.size() * 8)); CK(hipMalloc((void**)&dspec, hx.size() * 8)); CK(hipMalloc((void**)&dback, hx.size() * 8)); CK(hipMalloc((void**)&dtw, 8192));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dtw, htw.data(), 8192, hipMemcpyHostToDevice));
    // float64 DFT of vector 0 and of the last one
    auto check = [&](const char* name, int mode) -> int {
        std::vector<std::complex<float>> hs(hx.size()), hb(hx.size());
        CK(hipMemcpy(hs.data(), dspec, hs.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dback, hb.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0, worstb = 0;
        for (int vec : {0, 777, nvec - 1}) {
            // after `iters` rounds the row holds x again (forward, inverse, / 1024): the last forward spectrum is that of x up to rounding
            double mx = 0;
            std::vector<std::complex<double>> X(1024);
            for (int k = 0; k < 1024; ++k) {
                std::complex<double> s = 0;
                for (int n = 0; n < 1024; ++n) s += std::complex<double>(hx[(size_t)vec * 1024 + n]) * std::polar(1.0, -2.0 * M_PI * ((long)k * n % 1024) / 1024.0);
                X[k] = s; mx = std::max(mx, std::abs(s));
            }
            for (int k = 0; k < 1024; ++k) worst = std::max(worst, std::abs(std::complex<double>(hs[(size_t)vec * 1024 + k]) - X[k]) / mx);
            for (int n = 0; n < 1024; ++n) worstb = std::max(worstb, (double)std::abs(hb[(size_t)vec * 1024 + n] - hx[(size_t)vec * 1024 + n]));
        }
        printf("%s: forward spectrum rel err %.2e (after %d round trips), round-trip abs err %.2e  %s\n", name, worst, iters - 1, worstb, worst < 1e-4 && worstb < 1e-3 ? "ok" : "WRONG");
        (void)mode;
        return 0;
    };
    const size_t lds = (size_t)8 * S::ROW * 8 + 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode) {
        auto k = mode ? probe<1> : probe<0>;
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k));
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(cus), dim3(512), lds, 0, dx, dspec, dback, dtw, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%s: %.2f us per round (8 wavefronts per CU: one 1024-point forward + phase M + inverse each) = %.0f cycles at 2.1 GHz;  %d VGPRs, %zu B scratch\n",
               mode ? "exchanges in registers (permlane swaps + DPP)" : "exchanges through LDS (fft_tiled.h engine)      ", best * 1e3f / iters, best * 1e3 / iters * 2100, fa.numRegs,
               (size_t)fa.localSizeBytes);
        if (check(mode ? "  registers" : "  LDS      ", mode)) return 1;
    }
    return 0;
}
