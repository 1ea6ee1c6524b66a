// The mixer kernel shared by libpfdsp_hip.so (pfdsp_hip.hip: the reference's pf_mixer.h entries) and
// libpffft_hip.so (pffft_hip.hip: pffft_hip_shift_transform_batch for the sizes without a fused kernel):
//
//     out[i] = in[i] * S[i mod LANES] * exp(j 2 pi frac(step * (i div LANES)))
//
// S[] = up to 8 lane phasors, step = angle of one block step in turns (double: the reduction is exact at any
// stream position).  Pure streaming op: 8 B in + 8 B out per sample, HBM-bound; algorithmic bytes 16 B / sample.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>

namespace pfmix {

static thread_local std::string last_error;
static bool mix_force_static() {   // PFDSP_HIP_STATIC=1: never use the streaming kernel (A/B measurements)
    static const bool v = [] { const char* e = getenv("PFDSP_HIP_STATIC"); return e && e[0] == '1'; }();
    return v;
}

static int mix_fail(hipError_t e, const char* what) {
    last_error = std::string("pfdsp mix kernel: ") + what + " failed: " + hipGetErrorString(e);
    return (int)e;
}
#define PFMIX_CHECK(expr)                                     \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return mix_fail(_e, #expr);     \
    } while (0)

constexpr double MIX_TWO_PI = 6.28318530717958647692528676655900577;

constexpr int MIX_DYN_ROWS = 8;   // rows of 1 KiB per wave chunk of the streaming kernel

struct MixArgs {
    float2 S[8];     // lane phasors
    double step;     // turns per block of LANES samples
    float2 rot1;     // LANES == 1: rotation by one sample (second sample of a pair)
    float2 rowrot[MIX_DYN_ROWS - 1];  // streaming kernel: rotation between rows of a wave chunk, exp(j 2 pi step (128 / LANES) r), r = 1..7
};

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: nontemporal 16-byte accesses

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// exp(j 2 pi frac(step * k)), the product reduced in double (v_fract-exact), the sincos in float
__device__ __forceinline__ float2 rot_of(double step, unsigned long long k) {
    double t = step * (double)k;
    t -= rint(t);  // [-0.5, 0.5]
    // float(t) alone would cost up to 2e-7 rad: rotate by the rounding residue (first order, < 4e-8 rad) as well
    const float th = (float)t;
    const float d = 6.28318530717958647692f * (float)(t - (double)th);
    float s, c;
    sincospif(2.0f * th, &s, &c);
    return make_float2(c - s * d, s + c * d);
}

constexpr int MIX_THREADS = 256;
constexpr int MIX_UNROLL = 4;  // 16-byte accesses per thread: a workgroup streams 16 KiB in, 16 KiB out

// two samples (16 bytes) per access; mix_single_kernel below takes pointers that are only 8-byte aligned and odd tails
template <int LANES, bool GEN>
__global__ void __launch_bounds__(MIX_THREADS)
mix_pairs_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, unsigned long long first,
                 unsigned long long npairs, MixArgs a) {
    // `first` is a multiple of 512 pairs, so the lane phasor rule below still holds
    const unsigned long long base = first + (unsigned long long)blockIdx.x * (MIX_THREADS * MIX_UNROLL) + threadIdx.x;
    // 2p mod LANES depends on the thread only (every other term of p is a multiple of 256): lane phasors once
    const int l0 = (2 * (int)threadIdx.x) % LANES;
    const float2 s0 = a.S[l0], s1 = a.S[LANES == 1 ? 0 : l0 + 1];
    f32x4 x[MIX_UNROLL];
#pragma unroll
    for (int u = 0; u < MIX_UNROLL; ++u) {
        const unsigned long long p = base + (unsigned long long)u * MIX_THREADS;
        if (!GEN && p < npairs) x[u] = __builtin_nontemporal_load(in + p);
        else x[u] = f32x4{1.f, 0.f, 1.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < MIX_UNROLL; ++u) {
        const unsigned long long p = base + (unsigned long long)u * MIX_THREADS;
        if (p >= npairs) continue;
        const unsigned long long i0 = 2 * p;
        float2 w0, w1;
        if (LANES == 1) {
            const float2 r0 = rot_of(a.step, i0);
            w0 = cmulf(s0, r0);
            w1 = cmulf(w0, a.rot1);
        } else {
            const float2 r = rot_of(a.step, i0 / LANES);
            w0 = cmulf(s0, r);
            w1 = cmulf(s1, r);
        }
        const float2 y0 = cmulf(make_float2(x[u].x, x[u].y), w0);
        const float2 y1 = cmulf(make_float2(x[u].z, x[u].w), w1);
        __builtin_nontemporal_store(f32x4{y0.x, y0.y, y1.x, y1.y}, out + p);
    }
}

// Streaming kernel for long streams: the work distribution of the headline FFT kernel (fft_c1024.h, tools/membench.hip):
// persistent workgroups of 8 wavefronts pull groups of 8 consecutive 8 KiB chunks from an atomic counter, so the chip
// sweeps the stream in order (a copy with this pattern: 6.9 TB/s; hardware dispatch order: 5.5-6.0).  A wave moves 8 rows
// of 1 KiB (one 16-byte access per lane and row); the phase is reduced once per chunk and lane, the other rows follow by
// constant rotations.  ctr[0] = next group, ctr[1] = workgroups finished (the last one re-arms the pair).
constexpr int MIX_DYN_WAVES = 8;
constexpr unsigned long long MIX_CHUNK_PAIRS = 64ull * MIX_DYN_ROWS;   // 512 pairs = 1024 samples = 8 KiB

template <int LANES, bool GEN>
__global__ void __launch_bounds__(MIX_DYN_WAVES * 64)
mix_dyn_kernel(const f32x4* in, f32x4* out, unsigned nchunks, MixArgs a, unsigned* ctr) {
    __shared__ unsigned s_next[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l0 = (2 * lane) % LANES;
    const float2 s0 = a.S[l0], s1 = a.S[LANES == 1 ? 0 : l0 + 1];
    unsigned pend = 0;
    // the first TWO groups of a workgroup are static (its index, and that plus the grid); the counter hands out what follows: value v =
    // group 2 grid + v.  (Every workgroup used to open with two grabs: ~2 000 atomics on one address, served at ~80 M/s, stood between the
    // launch and the last workgroup's first load - 25-35 us of every launch, tools/r4_small_batch.py.)
    pend = blockIdx.x + gridDim.x;
    __syncthreads();
    unsigned g = blockIdx.x;
    const unsigned long long lastc = (unsigned long long)nchunks - 1;
    f32x4 x[MIX_DYN_ROWS];
    {   // clamped: always a valid address, so the loads are unconditional
        const unsigned long long c0 = (unsigned long long)g * MIX_DYN_WAVES + wave;
        const f32x4* src = in + (c0 < lastc ? c0 : lastc) * MIX_CHUNK_PAIRS + lane;
#pragma unroll
        for (int r = 0; r < MIX_DYN_ROWS; ++r) x[r] = GEN ? f32x4{1.f, 0.f, 1.f, 0.f} : __builtin_nontemporal_load(src + 64 * r);
    }
    for (unsigned it = 0; (unsigned long long)g * MIX_DYN_WAVES < nchunks; ++it) {
        if (threadIdx.x == 0) {   // publish the group of iteration it+1, grab the one of it+2 (latency never exposed)
            s_next[(it + 1) & 1] = pend;
            pend = 2u * gridDim.x + atomicAdd(&ctr[0], 1u);
        }
        __syncthreads();
        const unsigned gn = s_next[(it + 1) & 1];
        const unsigned long long c = (unsigned long long)g * MIX_DYN_WAVES + wave;
        const unsigned long long cn = (unsigned long long)gn * MIX_DYN_WAVES + wave;
        f32x4 xn[MIX_DYN_ROWS];
        {   // the next chunk is in flight while this one is rotated and stored
            const f32x4* src = in + (cn < lastc ? cn : lastc) * MIX_CHUNK_PAIRS + lane;
#pragma unroll
            for (int r = 0; r < MIX_DYN_ROWS; ++r) xn[r] = GEN ? f32x4{1.f, 0.f, 1.f, 0.f} : __builtin_nontemporal_load(src + 64 * r);
        }
        if (c < nchunks) {
            const unsigned long long p0 = c * MIX_CHUNK_PAIRS + lane;
            const float2 rb = rot_of(a.step, (2 * p0) / LANES);
            const float2 b0 = cmulf(s0, rb);
            const float2 b1 = LANES == 1 ? cmulf(b0, a.rot1) : cmulf(s1, rb);
#pragma unroll
            for (int r = 0; r < MIX_DYN_ROWS; ++r) {
                const float2 w0 = r ? cmulf(b0, a.rowrot[r - 1]) : b0;
                const float2 w1 = r ? cmulf(b1, a.rowrot[r - 1]) : b1;
                const float2 y0 = cmulf(make_float2(x[r].x, x[r].y), w0);
                const float2 y1 = cmulf(make_float2(x[r].z, x[r].w), w1);
                __builtin_nontemporal_store(f32x4{y0.x, y0.y, y1.x, y1.y}, out + p0 + 64 * r);
            }
        }
#pragma unroll
        for (int r = 0; r < MIX_DYN_ROWS; ++r) x[r] = xn[r];
        g = gn;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        unsigned d = atomicAdd(&ctr[1], 1u);
        if (d == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

// one ring of {next, done} counter pairs per device, created at the first long stream
constexpr unsigned MIX_CTR_RING = 1024;
static unsigned* mix_counters(int* cus_out) {
    static std::mutex mu;
    static unsigned* ring[64] = {};
    static int cus[64] = {};
    static std::atomic<unsigned> slot{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!ring[dev]) {
            unsigned* p = nullptr;
            if (hipMalloc((void**)&p, sizeof(unsigned) * 2 * MIX_CTR_RING) != hipSuccess ||
                hipMemset(p, 0, sizeof(unsigned) * 2 * MIX_CTR_RING) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            hipDeviceProp_t prop;
            cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            ring[dev] = p;
        }
    }
    *cus_out = cus[dev];
    return ring[dev] + 2 * (slot.fetch_add(1) % MIX_CTR_RING);
}

template <int LANES, bool GEN>
__global__ void __launch_bounds__(MIX_THREADS)
mix_single_kernel(const float2* __restrict__ in, float2* __restrict__ out, unsigned long long first,
                  unsigned long long n, MixArgs a) {
    const unsigned long long i = first + (unsigned long long)blockIdx.x * MIX_THREADS + threadIdx.x;
    if (i >= n) return;
    const float2 x = GEN ? make_float2(1.f, 0.f) : in[i];
    const float2 r = rot_of(a.step, i / LANES);
    out[i] = cmulf(x, cmulf(a.S[i % LANES], r));
}

template <int LANES, bool GEN>
static int launch_mix_t(const float2* in, float2* out, size_t n, const MixArgs& a, hipStream_t st) {
    if (n == 0) return 0;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    size_t done = 0;
    if (aligned && n >= 2) {
        const unsigned long long npairs = n / 2;
        unsigned long long first = 0;
        // long streams (>= 4 rounds of the whole chip): the in-order streaming kernel over the full 8 KiB chunks
        const unsigned long long nchunks = npairs / MIX_CHUNK_PAIRS;
        int cus = 0;
        unsigned* ctr = nullptr;
        if (!mix_force_static() && nchunks >= 8192 && nchunks < 0xffffffffULL && (ctr = mix_counters(&cus)) != nullptr) {
            static const int wgs_per_cu = [] { const char* e = getenv("PFDSP_HIP_WGS"); return e ? atoi(e) : 1; }();
            hipLaunchKernelGGL((mix_dyn_kernel<LANES, GEN>), dim3((unsigned)(cus * wgs_per_cu)), dim3(MIX_DYN_WAVES * 64), 0, st,
                               reinterpret_cast<const f32x4*>(in), reinterpret_cast<f32x4*>(out), (unsigned)nchunks, a, ctr);
            PFMIX_CHECK(hipGetLastError());
            first = nchunks * MIX_CHUNK_PAIRS;
        }
        if (first < npairs) {
            const unsigned long long per_block = MIX_THREADS * MIX_UNROLL;
            const unsigned long long blocks = (npairs - first + per_block - 1) / per_block;
            if (blocks > 0x7fffffffULL) { last_error = "pfdsp_hip: stream too long for one launch"; return (int)hipErrorInvalidValue; }
            hipLaunchKernelGGL((mix_pairs_kernel<LANES, GEN>), dim3((unsigned)blocks), dim3(MIX_THREADS), 0, st,
                               reinterpret_cast<const f32x4*>(in), reinterpret_cast<f32x4*>(out), first, npairs, a);
            PFMIX_CHECK(hipGetLastError());
        }
        done = 2 * (size_t)npairs;
    }
    if (done < n) {
        const unsigned long long rest = n - done;
        const unsigned long long blocks = (rest + MIX_THREADS - 1) / MIX_THREADS;
        if (blocks > 0x7fffffffULL) { last_error = "pfdsp_hip: stream too long for one launch"; return (int)hipErrorInvalidValue; }
        hipLaunchKernelGGL((mix_single_kernel<LANES, GEN>), dim3((unsigned)blocks), dim3(MIX_THREADS), 0, st, in, out,
                           (unsigned long long)done, (unsigned long long)n, a);
        PFMIX_CHECK(hipGetLastError());
    }
    return 0;
}

// out[i] = in[i] * S[i % lanes] * exp(j 2 pi frac(step_turns * (i / lanes))), device pointers
static int launch_mix(const float2* in, float2* out, size_t n, int lanes, const double (*S)[2], double step_turns,
                      bool gen, hipStream_t st) {
    MixArgs a;
    for (int l = 0; l < 8; ++l) a.S[l] = make_float2(l < lanes ? (float)S[l][0] : 1.f, l < lanes ? (float)S[l][1] : 0.f);
    step_turns -= std::rint(step_turns);
    a.step = step_turns;
    a.rot1 = make_float2((float)std::cos(MIX_TWO_PI * step_turns), (float)std::sin(MIX_TWO_PI * step_turns));
    for (int r = 1; r < MIX_DYN_ROWS; ++r) {
        double t = step_turns * (128.0 / lanes) * r;
        t -= std::rint(t);
        a.rowrot[r - 1] = make_float2((float)std::cos(MIX_TWO_PI * t), (float)std::sin(MIX_TWO_PI * t));
    }
    switch (lanes * 2 + (gen ? 1 : 0)) {
        case 2: return launch_mix_t<1, false>(in, out, n, a, st);
        case 3: return launch_mix_t<1, true>(in, out, n, a, st);
        case 8: return launch_mix_t<4, false>(in, out, n, a, st);
        case 9: return launch_mix_t<4, true>(in, out, n, a, st);
        case 16: return launch_mix_t<8, false>(in, out, n, a, st);
        case 17: return launch_mix_t<8, true>(in, out, n, a, st);
    }
    last_error = "pfdsp_hip: bad lane count";
    return (int)hipErrorInvalidValue;
}

}  // namespace pfmix
