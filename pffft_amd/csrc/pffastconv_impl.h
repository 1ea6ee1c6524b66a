// pffastconv on the GPU: overlap-save FIR with ALL blocks of a call batched into one pass.
//
// Reference: src/pffastconv.c — setup :58-116 (Nfft choice :62-80, flipped / zero-stuffed filter image
// :99-106, one forward transform into Hf :108), apply :133-263 (block loop :207-261: copy + zero pad,
// forward rFFT, pffft_zconvolve_no_accu with scale 1/Nfft, backward rFFT, copy numOut samples).
// The reference walks the blocks one by one on one core; here every block of the call is gathered
// into a [nblk][Nfft] device image and pushed through the batched kernels in place:
//   gather -> rFFT fwd (internal layout) -> x Hf * 1/Nfft -> rFFT bwd -> scatter.
// (included at the end of pffft_hip.hip)
#pragma once
#include "fft_fir.h"

namespace pf {

struct FastConv {
    uint32_t magic;
    PFFFT_Setup* st;
    int filterLen;  // effective length (2*len-1 for the single-FFT complex mode, src/pffastconv.c:93-94)
    int Nfft, flags, cplxFactor;
    float scale;
    std::vector<float> h_filter_image;  // time-domain image the reference builds in Xt (:99-106)
    std::mutex mu;
    bool ready = false;     // set only after EVERY step of fc_ensure_device succeeded
    int device = -1;
    float* d_Hf = nullptr;
    float* d_Hc = nullptr;  // canonical-order filter spectrum * 1/Nfft for the fused kernel
    // throughput regime (many blocks): the same outputs through LARGER internal blocks (better overlap-save efficiency)
    PFFFT_Setup* st_big = nullptr; float* d_Hc_big = nullptr; int Nfft_big = 0;
    // partitioned path (fft_fir.h fastconv_part_kernel): P spectra of 1024-tap partitions on 2048-sample blocks
    PFFFT_Setup* st_part = nullptr; float* d_Hp = nullptr; int part_P = 0;
    std::vector<float> h_td;  // y[m] = sum_i h_td[i] x[m + i]: the filter as the time-domain kernel applies it (zero padded to 8)
    float* d_td = nullptr;
    void* d_fir32_hp = nullptr;    // thread-major filter spectrum of the 32-points-per-thread block kernel (fft_fir32.h), of d_Hc_big
    void* d_fir32_hp_ref = nullptr;   // ... of d_Hc (filters whose reference block length is 16384 samples itself)
    void* d_split1_ab = nullptr;   // folded per-bin coefficients of the few-block split kernel (fft_split.h), built on first use
    // work image of the composed path: one per stream (two streams running one setup must not share scratch)
    struct Work { float* p = nullptr; size_t floats = 0; unsigned long long last_use = 0; bool captured = false; };   // captured: pf_host.h Scratch
    std::vector<float*> retired;
    unsigned long long work_clock = 0;
    std::map<hipStream_t, Work> work;
    float* d_x = nullptr; size_t x_floats = 0;   // staging of pffastconv_apply's host pointers (large signals)
    float* d_y = nullptr; size_t y_floats = 0;
    float* h_x = nullptr; size_t hx_floats = 0;  // pinned host images the kernels read / write directly (zero copy)
    float* h_y = nullptr; size_t hy_floats = 0;
};
constexpr uint32_t FC_MAGIC = 0x46434e56u;

// mode 0: real stream (also the single-FFT complex mode, which is a real stream of 2x length)
// mode 1: interleaved complex input processed as two real streams (part = block & 1)
__global__ void fastconv_gather_kernel(const float* __restrict__ x, float* __restrict__ blocks, int nblk, int Nfft,
                                       int step, int inputLen, int mode) {
    const size_t total = (size_t)nblk * Nfft;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int b = (int)(i / Nfft), j = (int)(i - (size_t)b * Nfft);
        float v = 0.f;
        if (mode == 0) {
            long src = (long)b * step + j;
            if (src < inputLen) v = x[src];
        } else {
            int part = b & 1;
            long src = (long)(b >> 1) * step + j;
            if (src < inputLen) v = x[2 * src + part];
        }
        blocks[i] = v;
    }
}

__global__ void fastconv_scatter_kernel(const float* __restrict__ blocks, float* __restrict__ y, int nblk, int Nfft,
                                        int step, int lastOut, int mode) {
    const int nb_time = mode == 0 ? nblk : nblk >> 1;
    const size_t total = (size_t)nblk * step;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int b = (int)(i / step), j = (int)(i - (size_t)b * step);
        int tb = mode == 0 ? b : (b >> 1);
        int numOut = (tb == nb_time - 1) ? lastOut : step;
        if (j >= numOut) continue;
        float v = blocks[(size_t)b * Nfft + j];
        if (mode == 0) y[(size_t)tb * step + j] = v;
        else y[2 * ((size_t)tb * step + j) + (b & 1)] = v;
    }
}

// ---- short filters: time domain ------------------------------------------------------------------------------------
// The reference always goes through Nfft = max(32, 2 next_pow2(len - 1)) transforms.  For <= TD_MAX_TAPS taps that is two
// transforms per (Nfft - len + 1) outputs of a kernel family that is latency-bound at these sizes (64 taps: 36
// Gsamples/s, 512 taps: 77; this kernel: 378 and 84 - the crossover is near 550 taps); the same outputs  y[m] = sum_i c_i x[m + i]  (the circular product of :99-108 and :238-255 written
// out: c_i = filter[len-1-i], or filter[i] with PFFASTCONV_CORRELATION) cost len FMAs each.  A workgroup stages
// 2048 + len inputs in LDS, a thread owns 8 consecutive outputs and slides a 16-value register window over them, 8 taps
// per step (two 16-byte LDS reads for 64 FMAs), the taps come through scalar loads.  The block schedule the caller can
// observe (how many samples a call produces, src/pffastconv.c:156-166,204-210) is computed as before.
constexpr int TD_THREADS = 256, TD_PER = 8, TD_TILE = TD_THREADS * TD_PER, TD_MAX_TAPS = 512;

// STRIDE 2: interleaved complex samples filtered by a real filter = the same sum over every second float,
// y[f] = sum_i c_i x[f + 2 i] — both complex modes of the reference (two real transforms per block, or one transform with
// the zero-stuffed filter, src/pffastconv.c:84-106) are this on the float stream; they differ in the block schedule only.
template <int STRIDE>
__global__ void __launch_bounds__(TD_THREADS)
fastconv_td_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ c, int flen8, long produced,
                   long inputLen, size_t xstride, size_t ystride) {
    extern __shared__ __attribute__((aligned(16))) float sx[];
    x += (size_t)blockIdx.y * xstride;   // blockIdx.y = signal of a batch (pffastconv_hip_apply_batch)
    y += (size_t)blockIdx.y * ystride;
    const int tid = threadIdx.x;
    const long o0 = (long)blockIdx.x * TD_TILE;
    const int outs = (produced - o0) < TD_TILE ? (int)(produced - o0) : TD_TILE;
    const int need = outs + STRIDE * (flen8 - 1);      // produced + STRIDE (len - 1) <= inputLen (fc_schedule); the zero taps of
    for (int i = tid; i < TD_TILE + STRIDE * flen8 + 8; i += TD_THREADS)   // the padding may point past the input: guarded
        sx[i] = (i < need && o0 + i < inputLen) ? x[o0 + i] : 0.f;
    __syncthreads();
    const float* w = sx + tid * TD_PER;
    float acc[TD_PER];
#pragma unroll
    for (int k = 0; k < TD_PER; ++k) acc[k] = 0.f;
    vec4<float> lo0 = *reinterpret_cast<const vec4<float>*>(w), lo1 = *reinterpret_cast<const vec4<float>*>(w + 4);
    constexpr int TAPS_PER_STEP = 8 / STRIDE;          // the window advances 8 floats per step
    for (int j = 0, wo = 8; j < flen8; j += TAPS_PER_STEP, wo += 8) {
        const vec4<float> hi0 = *reinterpret_cast<const vec4<float>*>(w + wo);
        const vec4<float> hi1 = *reinterpret_cast<const vec4<float>*>(w + wo + 4);
        const float win[16] = {lo0.x, lo0.y, lo0.z, lo0.w, lo1.x, lo1.y, lo1.z, lo1.w,
                               hi0.x, hi0.y, hi0.z, hi0.w, hi1.x, hi1.y, hi1.z, hi1.w};
#pragma unroll
        for (int jj = 0; jj < TAPS_PER_STEP; ++jj) {
            const float cj = c[j + jj];                // wave-uniform: scalar load
#pragma unroll
            for (int k = 0; k < TD_PER; ++k) acc[k] = __builtin_fmaf(cj, win[STRIDE * jj + k], acc[k]);
        }
        lo0 = hi0; lo1 = hi1;
    }
    const int m0 = tid * TD_PER;
    float* dst = y + o0 + m0;
    if (m0 + TD_PER <= outs && (((uintptr_t)dst) & 15) == 0) {
        vec4<float> a, b;
        a.x = acc[0]; a.y = acc[1]; a.z = acc[2]; a.w = acc[3]; b.x = acc[4]; b.y = acc[5]; b.z = acc[6]; b.w = acc[7];
        *reinterpret_cast<vec4<float>*>(dst) = a;
        *reinterpret_cast<vec4<float>*>(dst + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < TD_PER; ++k) if (m0 + k < outs) dst[k] = acc[k];
    }
}

static int fc_grow(float** p, size_t* have, size_t want) {
    if (*have >= want) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    PF_CHECK(hipMalloc((void**)p, want * sizeof(float)));
    *have = want;
    return 0;
}

static void fc_free_device(FastConv* s) {
    for (float** p : {&s->d_Hf, &s->d_Hc}) if (*p) { (void)hipFree(*p); *p = nullptr; }
}

// everything this setup holds on its device: the filter tables of every route, per-stream work images, staging (not the pinned host
// images, not the inner pffft setups - those keep state per device themselves, for_device)
static void fc_release_device(FastConv* s) {
    fc_free_device(s);
    for (float** p : {&s->d_Hc_big, &s->d_Hp, &s->d_td, &s->d_x, &s->d_y}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    for (void** p : {&s->d_split1_ab, &s->d_fir32_hp, &s->d_fir32_hp_ref}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    for (auto& kv : s->work) if (kv.second.p) (void)hipFree(kv.second.p);
    s->work.clear();
    for (float* p : s->retired) if (p) (void)hipFree(p);
    s->retired.clear();
    s->Nfft_big = 0; s->part_P = 0; s->x_floats = 0; s->y_floats = 0;
    s->ready = false;
}

static int fc_init_device(FastConv* s) {
    PF_CHECK(hipMalloc((void**)&s->d_Hf, sizeof(float) * s->Nfft));
    PF_CHECK(hipMemcpy(s->d_Hf, s->h_filter_image.data(), sizeof(float) * s->Nfft, hipMemcpyHostToDevice));
    int rc = transform_batch<float>(s->st, s->d_Hf, s->d_Hf, 1, PFFFT_FORWARD, 0, nullptr);  // :108
    if (rc) return rc;
    // fused path: canonical order (pffft_zreorder) and the 1/Nfft scale folded into the table
    PF_CHECK(hipMalloc((void**)&s->d_Hc, sizeof(float) * s->Nfft));
    rc = zreorder_batch<float>(s->st, s->d_Hf, s->d_Hc, 1, PFFFT_FORWARD, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(fastconv_scale_kernel, dim3(64), dim3(256), 0, nullptr, s->d_Hc, s->d_Hc, s->Nfft, s->scale);
    PF_CHECK(hipGetLastError());
    PF_CHECK(hipStreamSynchronize(nullptr));
    return 0;
}

// A PFFASTCONV_Setup is used by one thread at a time (the reference's is "not shareable", include/pffft/pffastconv.h:77-80,142-143) and
// holds the filter's tables on ONE device: the calling thread's current one.  Round 6: a call from another device no longer fails - the
// tables are released and rebuilt there (the filter is kept on the host), i.e. the setup follows its user from device to device; a caller
// that alternates between devices call by call should hold one setup per device.
static int fc_ensure_device(FastConv* s) {
    int dev = -1;
    int rc = current_device_key(&dev);
    if (rc) return rc;
    if (s->ready) {
        if (dev == s->device) return 0;
        fc_release_device(s);
    }
    rc = fc_init_device(s);
    if (rc) { fc_free_device(s); return rc; }   // a later call starts over instead of launching on half-built tables
    s->device = dev;
    s->ready = true;
    return 0;
}

template <class C>
static int fc_launch_fused(FastConv* s, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                           hipStream_t st, const FcBatch& fb, PFFFT_Setup* pst = nullptr, const float* d_Hc = nullptr) {
    auto k = fastconv_fused_kernel<C>;
    int rc = allow_big_lds(k, C::LDS_BYTES);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), C::WG_THREADS, C::LDS_BYTES, &per_cu))) return rc;
    size_t groups = ((size_t)nblk * fb.nsig + C::T_PER_WG - 1) / C::T_PER_WG;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    Setup* ps = for_device(pst ? pst : s->st);
    if (!d_Hc) d_Hc = s->d_Hc;
    unsigned* ctr = groups <= grid ? nullptr : take_counters(ps, st);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), C::LDS_BYTES, st, d_x, d_y, (const cx<float>*)d_Hc,
                       nblk, step, inputLen, lastOut, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, ctr,
                       fb.nsig, fb.xstride, fb.ystride);
    PF_CHECK(hipGetLastError());
    return 0;
}

// filters of up to this many taps take the wave kernel (from g_fir_wave_min on, below): beyond, the 16384-sample split kernel (fft_split.h, round 4) is faster - 800 taps 0.44 / 0.465 (wave) against
// 0.41 / 0.46 (split), 900 taps 0.40 / 0.43 against 0.41 / 0.46, 1024 taps 0.37 / 0.39 against 0.40 / 0.45 (tools/fir_shapes.py)
static const int g_fir_wave_max = dev_env("PFFASTCONV_HIP_WAVE_MAX", 850);

// Internal block length for the throughput regime.  What a caller can observe of the reference's blocks is only HOW MANY
// samples a call produces (fc_schedule); the values are those of the exact convolution whatever the block length, so
// when a call has many blocks the same `produced` outputs are computed with longer internal blocks: overlap-save
// efficiency (Nfft - len + 1) / Nfft goes from ~0.5 (the reference's Nfft = 2 next_pow2(len-1), src/pffastconv.c:62-63)
// to 0.75-0.94.  PFFASTCONV_HIP_NFFT=<n> forces a length (A/B), =0 switches this off.
static int fc_big_nfft(const FastConv* s, long produced, int nsig) {
    const int forced = env().fir_nfft;
    if (forced == 0) return 0;
    const int taps = s->filterLen;
    // measured table (MI355X, tools/fir_quick.py with PFFASTCONV_HIP_NFFT forced; fraction of the 8 B / sample roofline on
    // 2^26 samples / on 256 signals of 2^20; r02):      Nfft   2048          4096          8192          16384
    //    200 taps (FFT route forced)                         0.12 / 0.15   0.23 / 0.30   0.26 / 0.31   0.26 / 0.30
    //    600 taps                                            0.10 / 0.12   0.22 / 0.27   0.25 / 0.30   0.27 / 0.29
    //   1024 taps                                            0.08 / 0.09   0.20 / 0.24   0.24 / 0.28   0.26 / 0.29
    //   2048 taps                                                 -        0.14 / 0.17   0.22 / 0.25   0.25 / 0.27
    //   4096 taps                                                 -             -        0.16 / 0.19   0.21 / 0.24
    // -> 16384 from 1024 taps on, 8192 below (the time-domain kernel wins up to ~128 taps).  All block kernels saturate
    // near 0.26-0.30: two 8192-point transforms per block cost ~33 k cycles per CU whatever the filter.
    const int want = forced > 0 ? forced : (taps <= g_fir_wave_max ? 8192 : 16384);
    if (want <= s->Nfft || want > 16384 || (want & (want - 1)) || want < 2 * taps) return 0;
    if (forced > 0) return want;
    if (taps <= 128) return 0;
    const long bblk = (produced + (want - taps)) / (want - taps + 1);
    return bblk * nsig >= num_cus() ? want : 0;          // fewer blocks: latency regime, one reference-sized block per CU is faster
}

static int fc_ensure_big(FastConv* s, int Nfft_big) {
    if (s->Nfft_big == Nfft_big) return 0;
    if (s->st_big) { pffft_destroy_setup(s->st_big); s->st_big = nullptr; }
    if (s->d_Hc_big) { (void)hipFree(s->d_Hc_big); s->d_Hc_big = nullptr; }
    if (s->d_fir32_hp) { (void)hipFree(s->d_fir32_hp); s->d_fir32_hp = nullptr; }
    s->Nfft_big = 0;
    s->st_big = pffft_new_setup(Nfft_big, PFFFT_REAL);
    if (!s->st_big) { g_last_error = "pffastconv: internal setup failed"; return (int)hipErrorInvalidValue; }
    std::vector<float> img((size_t)Nfft_big, 0.f);
    const int flen = s->filterLen;
    for (int i = 0; i < flen; ++i) img[(Nfft_big - i) & (Nfft_big - 1)] = s->h_td[i];   // :100-106 with the longer block
    float* d_tmp = nullptr;
    PF_CHECK(hipMalloc((void**)&d_tmp, sizeof(float) * Nfft_big));
    PF_CHECK(hipMalloc((void**)&s->d_Hc_big, sizeof(float) * Nfft_big));
    PF_CHECK(hipMemcpy(d_tmp, img.data(), sizeof(float) * Nfft_big, hipMemcpyHostToDevice));
    int rc = transform_batch<float>(s->st_big, d_tmp, d_tmp, 1, PFFFT_FORWARD, 0, nullptr);
    if (!rc) rc = zreorder_batch<float>(s->st_big, d_tmp, s->d_Hc_big, 1, PFFFT_FORWARD, nullptr);
    if (!rc) {
        hipLaunchKernelGGL(fastconv_scale_kernel, dim3(64), dim3(256), 0, nullptr, s->d_Hc_big, s->d_Hc_big, Nfft_big, 1.0f / (float)Nfft_big);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) rc = (int)hipErrorUnknown;
    }
    (void)hipFree(d_tmp);
    if (rc) return rc;
    s->Nfft_big = Nfft_big;
    return 0;
}

// ---- partitioned path: the filter as P partitions of PART_B taps on blocks of 2 PART_B samples (fft_fir.h) ----
constexpr int PART_B = 1024, PART_MAXP = 4;
static int fc_ensure_part(FastConv* s) {
    const int P = (s->filterLen + PART_B - 1) / PART_B;
    if (s->part_P == P && s->d_Hp) return 0;
    const int Nfft = 2 * PART_B;
    if (!s->st_part) s->st_part = pffft_new_setup(Nfft, PFFFT_REAL);
    if (!s->st_part) { g_last_error = "pffastconv: internal setup failed"; return (int)hipErrorInvalidValue; }
    // correlation image of partition p (src/pffastconv.c:100-106 with the partition's taps): g[(Nfft - i) mod Nfft] = c[pB + i]
    std::vector<float> img((size_t)P * Nfft, 0.f);
    for (int i = 0; i < s->filterLen; ++i) img[(size_t)(i / PART_B) * Nfft + ((Nfft - i % PART_B) & (Nfft - 1))] = s->h_td[i];
    float* d_tmp = nullptr;
    if (s->d_Hp) { (void)hipFree(s->d_Hp); s->d_Hp = nullptr; s->part_P = 0; }
    PF_CHECK(hipMalloc((void**)&d_tmp, sizeof(float) * P * Nfft));
    PF_CHECK(hipMalloc((void**)&s->d_Hp, sizeof(float) * P * Nfft));
    PF_CHECK(hipMemcpy(d_tmp, img.data(), sizeof(float) * P * Nfft, hipMemcpyHostToDevice));
    int rc = transform_batch<float>(s->st_part, d_tmp, d_tmp, P, PFFFT_FORWARD, 0, nullptr);
    if (!rc) rc = zreorder_batch<float>(s->st_part, d_tmp, s->d_Hp, P, PFFFT_FORWARD, nullptr);
    if (!rc) {
        hipLaunchKernelGGL(fastconv_scale_kernel, dim3(64), dim3(256), 0, nullptr, s->d_Hp, s->d_Hp, P * Nfft, 1.0f / (float)Nfft);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) rc = (int)hipErrorUnknown;
    }
    (void)hipFree(d_tmp);
    if (rc) { (void)hipFree(s->d_Hp); s->d_Hp = nullptr; return rc; }
    s->part_P = P;
    return 0;
}

#ifdef PFFFT_HIP_VARIANTS
template <int P, int OCC = (P > 2 ? 1 : 2)>
static int fc_launch_part(FastConv* s, const float* d_x, float* d_y, long produced, int inputLen, hipStream_t st, const FcBatch& fb) {
    typedef FirPartCfg::C1024 C;
    auto k = fastconv_part_kernel<C, P, OCC>;
    const size_t lds = ((size_t)C::T_PER_WG * C::IMG + (size_t)P * C::E * 2 * 64) * sizeof(cx<float>);
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), C::WG_THREADS, lds, &per_cu))) return rc;
    const int nblk = (int)((produced + PART_B - 1) / PART_B);
    const int lastOut = (int)(produced - (long)(nblk - 1) * PART_B);
    const long waves = (long)num_cus() * per_cu * C::T_PER_WG;
    // runs of consecutive output blocks per wavefront task: ~2 tasks per wavefront (measured on 2^26 samples: runs of 8-16
    // blocks 0.34, 4: 0.32, 32: 0.30, 64: 0.22; on 256 signals of 2^20: 32-64 best — both ~1.3-2.7 tasks per wavefront; the
    // first loads of a run are not prefetched), never so short that filling the ring (P - 1 extra forward transforms per
    // run) costs more than ~10 %
    long kchunk = ((long)nblk * fb.nsig + 2 * waves - 1) / (2 * waves);
    if (kchunk < 10 * (P - 1)) kchunk = 10 * (P - 1);
    if (kchunk < 8) kchunk = 8;
    if (dev_env("PFFASTCONV_HIP_PART_K", 0) > 0) kchunk = dev_env("PFFASTCONV_HIP_PART_K", 0);
    if (kchunk > nblk) kchunk = nblk;
    const long ntask = ((nblk + kchunk - 1) / kchunk) * fb.nsig;
    long grid = (ntask + C::T_PER_WG - 1) / C::T_PER_WG;
    if (grid > (long)num_cus() * per_cu) grid = (long)num_cus() * per_cu;
    Setup* ps = for_device(s->st_part);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), lds, st, d_x, d_y, (const cx<float>*)s->d_Hp, nblk, inputLen,
                       lastOut, (int)kchunk, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, fb.nsig, fb.xstride, fb.ystride);
    PF_CHECK(hipGetLastError());
    return 0;
}
#endif

// one wavefront per 2048-sample block, step = 2048 - taps + 1 (fft_fir.h fastconv_wave_kernel): filters up to 1024 taps
static int fc_launch_wave(FastConv* s, const float* d_x, float* d_y, long produced, int inputLen, hipStream_t st, const FcBatch& fb) {
    typedef FirPartCfg::C1024 C;
    auto k = fastconv_wave_kernel<C, 3>;
    const size_t lds = ((size_t)C::T_PER_WG * C::IMG + (size_t)C::E * 2 * 64) * sizeof(cx<float>);
    int rc = allow_big_lds(k, lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(k), C::WG_THREADS, lds, &per_cu))) return rc;
    const int step = (2 * PART_B - s->filterLen + 1) & ~3;        // valid samples per block, 16-byte store units
    const int nblk = (int)((produced + step - 1) / step);
    const int lastOut = (int)(produced - (long)(nblk - 1) * step);
    const long waves = (long)num_cus() * per_cu * C::T_PER_WG;
    long kchunk = ((long)nblk * fb.nsig + 2 * waves - 1) / (2 * waves);   // ~2 tasks per wavefront (fc_launch_part)
    if (kchunk < 4) kchunk = 4;
    if (kchunk > nblk) kchunk = nblk;
    const long ntask = ((nblk + kchunk - 1) / kchunk) * fb.nsig;
    long grid = (ntask + C::T_PER_WG - 1) / C::T_PER_WG;
    if (grid > (long)num_cus() * per_cu) grid = (long)num_cus() * per_cu;
    Setup* ps = for_device(s->st_part);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::WG_THREADS), lds, st, d_x, d_y, (const cx<float>*)s->d_Hp, nblk, step, inputLen,
                       lastOut, (int)kchunk, (const cx<float>*)ps->d_tw, (const cx<float>*)ps->d_twr, fb.nsig, fb.xstride, fb.ystride);
    PF_CHECK(hipGetLastError());
    return 0;
}
// Filters from this many taps up take the wave kernel when a call has many blocks (PFFASTCONV_HIP_WAVE_MIN, A/B); below: time
// domain.  Measured (MI355X, tools/fir_quick.py, fraction of the 8 B / sample roofline on 2^26 samples / 256 signals of 2^20;
// time domain or partitioned kernel -> wave kernel): 8-24 taps 0.47 / 0.50-0.52 -> 0.47-0.49 / 0.56 (a tie: the time-domain
// kernel stays), 32 taps 0.45 / 0.49 -> 0.52 / 0.57, 64 taps 0.38 / 0.43 -> 0.52 / 0.55, 128 taps 0.25 / 0.30 -> 0.52 / 0.55,
// 200 taps 0.31 / 0.38 -> 0.54 / 0.54, 600 taps 0.31 / 0.38 -> 0.39 / 0.47, 800 taps -> 0.35 / 0.43, 1024 taps 0.30 / 0.37 (equal).
static const int g_fir_wave_min = dev_env("PFFASTCONV_HIP_WAVE_MIN", 32);

// Block schedule of src/pffastconv.c:156-166 / :204-210.  Returns the number of time blocks and the
// number of outputs of the last one; *produced = value returned by pffastconv_apply (in real samples).
static int fc_schedule(const FastConv* s, int inputLen, int flush, int* lastOut, int* produced) {
    const int Nfft = s->Nfft, flen = s->filterLen;
    const int step_full = s->cplxFactor == 2 ? ((Nfft - flen + 1) & ~1) : (Nfft - flen + 1);
    const long maxOff = flush ? ((long)inputLen - flen + 1) : ((long)inputLen - Nfft + 1);
    if (s->cplxFactor == 1) {
        // closed form of the loop below (it runs once per block: a million iterations for a short filter on a long signal):
        // full blocks while off + Nfft <= inputLen, then - flushing only - one partial block with the remaining outputs
        const long nfull = inputLen >= Nfft ? ((long)inputLen - Nfft) / step_full + 1 : 0;
        long off = nfull * step_full;
        int nb = (int)nfull, lst = nfull ? step_full : 0;
        if (flush && off < maxOff) { lst = (int)(inputLen - off - flen + 1); off += lst; ++nb; }
        *lastOut = lst;
        *produced = (int)off;
        return nb;
    }
    int nblk = 0, last = 0;
    long off = 0;
    while (off < maxOff) {
        long remain = inputLen - off;
        int procLen = remain >= Nfft ? Nfft : (int)remain;
        int numOut = procLen - flen + 1;
        if (s->cplxFactor == 2) { numOut &= ~1; if (!numOut) break; }
        ++nblk; last = numOut; off += numOut;
        if (numOut != step_full) break;  // a partial block is always the last one
    }
    *lastOut = last;
    *produced = (int)off;
    return nblk;
}

static int fc_apply_device(FastConv* s, const float* d_x, int cplxInputLen, float* d_y, int flush, hipStream_t st,
                           int* produced_out, const FcBatch& fb = FcBatch{1, 0, 0}) {
    std::lock_guard<std::mutex> lk(s->mu);
    int rc = fc_ensure_device(s);
    if (rc) return rc;
    const int inputLen = s->cplxFactor * cplxInputLen;  // :141
    const int mode = ((s->flags & PFFASTCONV_HIP_CPLX_INP_OUT) && s->cplxFactor == 1) ? 1 : 0;
    int lastOut = 0, produced = 0;
    const int nbt = fc_schedule(s, inputLen, flush, &lastOut, &produced);
    *produced_out = produced / s->cplxFactor;  // :200 / :261
    if (nbt == 0) return 0;
    const int nblk = mode ? 2 * nbt : nbt;
    const int Nfft = s->Nfft;
    const int step = s->cplxFactor == 2 ? ((Nfft - s->filterLen + 1) & ~1) : (Nfft - s->filterLen + 1);
    const AbSel sel = ab();
    if (mode == 0 && s->cplxFactor == 1 && !sel.any() && s->filterLen >= g_fir_wave_min && s->filterLen <= PART_B && s->filterLen <= g_fir_wave_max && produced > 0) {
        // many blocks of a filter of up to 1024 taps: one wavefront per 2048-sample block, step = 2048 - taps + 1 (round 3;
        // the partitioned kernel below advances 1024 samples per block whatever the filter)
        const int wstep = (2 * PART_B - s->filterLen + 1) & ~3;
        if (((long)produced + wstep - 1) / wstep * fb.nsig >= 8L * num_cus()) {
            if ((rc = fc_ensure_part(s))) return rc;
            if (s->part_P == 1) return fc_launch_wave(s, d_x, d_y, produced, inputLen, st, fb);
        }
    }
#ifdef PFFFT_HIP_VARIANTS
    if (mode == 0 && s->cplxFactor == 1 && s->filterLen > TD_MAX_TAPS / 4 && s->filterLen <= PART_B * PART_MAXP &&
        sel.is(AB_FIR_PARTITIONED) && produced > 0) {
        // development build: the uniformly partitioned one-wavefront-per-block kernel (fft_fir.h fastconv_part_kernel, round 2;
        // AB_FIR_PARTITIONED forces it).  Measured (fraction of the 8 B / sample roofline, 2^26 samples / 256
        // signals of 2^20): one partition 0.30 / 0.37 - superseded by the wave kernel above, which advances by the samples the
        // filter leaves valid instead of 1024; two partitions (2048 taps) 0.25 / 0.26 lose to the long-block DMA kernel; three
        // and four keep their ring in > 256 registers and run one wavefront per SIMD (0.12-0.23).
        if ((rc = fc_ensure_part(s))) return rc;
        switch (s->part_P) {
            case 1: return fc_launch_part<1, 3>(s, d_x, d_y, produced, inputLen, st, fb);
            case 2: return fc_launch_part<2, 2>(s, d_x, d_y, produced, inputLen, st, fb);
            case 3: return fc_launch_part<3, 1>(s, d_x, d_y, produced, inputLen, st, fb);
            case 4: return fc_launch_part<4, 1>(s, d_x, d_y, produced, inputLen, st, fb);
            default: break;
        }
    }
#endif
    if (mode == 0 && s->cplxFactor == 1) {
        const int nbig = fc_big_nfft(s, produced, fb.nsig);
        if (nbig) {
            if ((rc = fc_ensure_big(s, nbig))) return rc;
            const int bstep = nbig - s->filterLen + 1;
            const int bblk = (int)(((long)produced + bstep - 1) / bstep);
            const int blast = (int)(produced - (long)(bblk - 1) * bstep);
            // 16384-sample blocks: the LDS-DMA staged split kernel (fft_split.h; the lock-step one it replaced: development build).  Shorter
            // internal blocks: the register-staged fused kernel (tools/dma_ab.py: a tie with the DMA kernel at Nfft 8192, 0.22-0.29 both)
            // 16384-sample blocks: 256 threads, 32 points per thread, four exchanges per block (fft_fir32.h, round 6); AB_FIR_SPLIT: the split
            // kernel it replaced (fft_split.h: the second route of tests/test_gpu_round6.py), development build: AB_FIR_LOCKSTEP / AB_FIR_SPLIT_PLAIN
            if (nbig == 16384 && !sel.is(AB_FIR_SPLIT) && !sel.is(AB_FIR_LOCKSTEP) && !sel.is(AB_FIR_SPLIT_PLAIN)) {
                rc = launch_fir32(s->st_big, s->d_Hc_big, d_x, d_y, bblk, bstep, inputLen, blast, st, fb, &s->d_fir32_hp,
                                  sel.is(AB_FIR_FUSED32_PF1) ? 1 : sel.is(AB_FIR_FUSED32_NOPF) ? 0 : 2);
                if (rc != -1) return rc;
            }
            if (nbig == 16384) {
                rc = launch_fir_dma(s->st_big, s->d_Hc_big, d_x, d_y, bblk, bstep, inputLen, blast, st, fb);
                if (rc != -1) return rc;
            }
            switch (nbig / 2) {
                case 1024: return fc_launch_fused<FirCfg::C1024>(s, d_x, d_y, bblk, bstep, inputLen, blast, st, fb, s->st_big, s->d_Hc_big);
                case 2048: return fc_launch_fused<FirCfg::C2048>(s, d_x, d_y, bblk, bstep, inputLen, blast, st, fb, s->st_big, s->d_Hc_big);
                case 4096: return fc_launch_fused<FirCfg::C4096>(s, d_x, d_y, bblk, bstep, inputLen, blast, st, fb, s->st_big, s->d_Hc_big);
                case 8192: return fc_launch_fused<FirCfg::C8192>(s, d_x, d_y, bblk, bstep, inputLen, blast, st, fb, s->st_big, s->d_Hc_big);
                default: break;
            }
        }
    }
    const int taps = s->cplxFactor == 2 ? (s->filterLen + 1) / 2 : s->filterLen;   // the caller's filter length
    if (taps <= TD_MAX_TAPS) {
        // short real filter: time domain; the complex modes are the stride-2 sum over the float stream
        if (!s->d_td) {
            PF_CHECK(hipMalloc((void**)&s->d_td, sizeof(float) * s->h_td.size()));
            PF_CHECK(hipMemcpy(s->d_td, s->h_td.data(), sizeof(float) * s->h_td.size(), hipMemcpyHostToDevice));
        }
        const int flen8 = (int)s->h_td.size();
        const bool cplx = mode == 1 || s->cplxFactor == 2;
        const long out_f = mode == 1 ? 2L * produced : produced, in_f = mode == 1 ? 2L * inputLen : inputLen;
        const size_t lds = sizeof(float) * (TD_TILE + (cplx ? 2 : 1) * flen8 + 8);
        // the signal index is blockIdx.y (<= 65535): longer batches go out in slices of 65535 signals on the same stream
        for (int s0 = 0; s0 < fb.nsig; s0 += 65535) {
            const int ns = fb.nsig - s0 < 65535 ? fb.nsig - s0 : 65535;
            const dim3 grid((unsigned)((out_f + TD_TILE - 1) / TD_TILE), (unsigned)ns);
            const float* xs = d_x + (size_t)s0 * fb.xstride;
            float* ys = d_y + (size_t)s0 * fb.ystride;
            if (cplx) hipLaunchKernelGGL(fastconv_td_kernel<2>, grid, dim3(TD_THREADS), lds, st, xs, ys, (const float*)s->d_td, flen8, out_f, in_f, fb.xstride, fb.ystride);
            else hipLaunchKernelGGL(fastconv_td_kernel<1>, grid, dim3(TD_THREADS), lds, st, xs, ys, (const float*)s->d_td, flen8, out_f, in_f, fb.xstride, fb.ystride);
        }
        PF_CHECK(hipGetLastError());
        return 0;
    }
    if (mode == 0 && Nfft == 16384 && (long)nblk * fb.nsig >= 2L * num_cus()) {
        if (!sel.is(AB_FIR_SPLIT) && !sel.is(AB_FIR_LOCKSTEP) && !sel.is(AB_FIR_SPLIT_PLAIN)) {
            rc = launch_fir32(s->st, s->d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, &s->d_fir32_hp_ref, 2);
            if (rc != -1) return rc;
        }
        rc = launch_fir_dma(s->st, s->d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);   // many reference-sized blocks
        if (rc != -1) return rc;
    }
    if (mode == 0 && !sel.is(AB_FIR_FEW_16PT) && !sel.is(AB_FIR_FEW_LOCKSTEP) && (Nfft == 8192 || Nfft == 4096)) {
        // few blocks of 8192 / 4096 samples: cross-wave radix 8 / 4 + wave-local 512-point transforms (fft_split.h
        // fastconv_split1_kernel, round 4); AB_FIR_FEW_LOCKSTEP = the lock-step kernel on 512 / 256 threads (the second route of
        // tests/test_gpu_round4.py), AB_FIR_FEW_16PT (development build) = on 256 / 128
        rc = launch_fir_split1(s->st, s->d_Hc, d_x, d_y, nblk, step, inputLen, lastOut, st, fb, &s->d_split1_ab);
        if (rc != -1) return rc;
    }
    if (mode == 0) {  // one real stream: the fused one-kernel path when Nfft/2 has a tiled kernel
        switch (Nfft / 2) {
            case 512: return fc_launch_fused<FirCfg::C512>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            case 1024: return fc_launch_fused<FirCfg::C1024>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            // 2048 / 4096 points: twice the threads per block, eight points per thread (FirCfg::C*m, round 4): the stated C4 call
            // 10.9 -> 9.5 us, 2048 taps on 2^19 samples 8.8 -> 7.3 us; n = 8192 on 1024 threads measured slower (15.9 -> 17.9 us).
            case 2048:
#ifdef PFFFT_HIP_VARIANTS
                if (sel.is(AB_FIR_FEW_16PT)) return fc_launch_fused<FirCfg::C2048>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
#endif
                return fc_launch_fused<FirCfg::C2048m>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            case 4096:
#ifdef PFFFT_HIP_VARIANTS
                if (sel.is(AB_FIR_FEW_16PT)) return fc_launch_fused<FirCfg::C4096>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
#endif
                return fc_launch_fused<FirCfg::C4096m>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            case 8192: return fc_launch_fused<FirCfg::C8192>(s, d_x, d_y, nblk, step, inputLen, lastOut, st, fb);
            default: break;
        }
    }
    // composed path (complex-I/O modes with long filters): signal by signal on the same stream through one work image
    if (s->work.size() >= 8 && !s->work.count(st)) {   // the stream that used this setup longest ago gives up its image (hipFree waits for its kernels)
        auto victim = s->work.end();                   // (never an image a HIP graph has recorded: a replay dereferences the frozen pointer)
        for (auto it = s->work.begin(); it != s->work.end(); ++it)
            if (!it->second.captured && (victim == s->work.end() || it->second.last_use < victim->second.last_use)) victim = it;
        if (victim != s->work.end()) {
            if (victim->second.p) (void)hipFree(victim->second.p);
            s->work.erase(victim);
        }
    }
    FastConv::Work& wk = s->work[st];
    wk.last_use = ++s->work_clock;
    if (stream_capturing(st)) wk.captured = true;
    if (wk.captured && wk.p && wk.floats < (size_t)nblk * Nfft) { s->retired.push_back(wk.p); wk.p = nullptr; wk.floats = 0; }   // outgrown: retired, not freed
    rc = fc_grow(&wk.p, &wk.floats, (size_t)nblk * Nfft);
    if (rc) return rc;
    float* const d_work = wk.p;
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)nblk * Nfft + 255) / 256, (size_t)num_cus() * 16);
    for (int sig = 0; sig < fb.nsig; ++sig) {
        const float* xs = d_x + (size_t)sig * fb.xstride;
        float* ys = d_y + (size_t)sig * fb.ystride;
        hipLaunchKernelGGL(fastconv_gather_kernel, dim3(grid), dim3(256), 0, st, xs, d_work, nblk, Nfft, step, inputLen,
                           mode);
        PF_CHECK(hipGetLastError());
        rc = transform_batch<float>(s->st, d_work, d_work, nblk, PFFFT_FORWARD, 0, st);            // :235
        if (rc) return rc;
        rc = zconvolve_batch<float>(s->st, d_work, s->d_Hf, d_work, s->scale, nblk, 0, 1, st);     // :238
        if (rc) return rc;
        rc = transform_batch<float>(s->st, d_work, d_work, nblk, PFFFT_BACKWARD, 0, st);           // :254
        if (rc) return rc;
        hipLaunchKernelGGL(fastconv_scatter_kernel, dim3(grid), dim3(256), 0, st, d_work, ys, nblk, Nfft, step, lastOut,
                           mode);
        PF_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace pf

struct PFFASTCONV_Setup : pf::FastConv {};

PF_EXPORT PFFASTCONV_Setup* pffastconv_new_setup(const float* filterCoeffs, int filterLen, int* blockLen, int flags) {
    using namespace pf;
    // src/pffastconv.c:61-82
    const int cplxFactor = ((flags & PFFASTCONV_HIP_CPLX_INP_OUT) && (flags & PFFASTCONV_HIP_CPLX_SINGLE_FFT)) ? 2 : 1;
    const int minFftLen = 2 * SIMD * SIMD;
    if (!filterCoeffs || filterLen <= 0 || !blockLen) return nullptr;
    int Nfft = 2 * next_pow2(filterLen - 1);
    if (Nfft < minFftLen) Nfft = minFftLen;
    if (flags & PFFASTCONV_HIP_CPLX_FILTER) return nullptr;  // :71-72 "not implemented yet"
    if (*blockLen > Nfft) Nfft = next_pow2(*blockLen);
    *blockLen = Nfft;
    Nfft *= cplxFactor;
    PFFFT_Setup* st = pffft_new_setup(Nfft, PFFFT_REAL);
    if (!st) return nullptr;
    PFFASTCONV_Setup* s = new PFFASTCONV_Setup();
    s->magic = FC_MAGIC;
    s->st = st;
    s->filterLen = cplxFactor == 2 ? 2 * filterLen - 1 : filterLen;
    s->Nfft = Nfft; s->flags = flags; s->cplxFactor = cplxFactor;
    s->scale = (float)(1.0 / Nfft);
    s->h_filter_image.assign(Nfft, 0.f);
    s->h_td.assign((size_t)(filterLen + 7) / 8 * 8, 0.f);
    for (int i = 0; i < filterLen; ++i) {  // :100-106
        float c = (flags & PFFASTCONV_HIP_CORRELATION) ? filterCoeffs[i] : filterCoeffs[filterLen - 1 - i];
        s->h_filter_image[(Nfft - cplxFactor * i) & (Nfft - 1)] = c;
        s->h_td[i] = c;
    }
    return s;
}

PF_EXPORT void pffastconv_destroy_setup(PFFASTCONV_Setup* s) {
    if (!s) return;
    pffft_destroy_setup(s->st);
    if (s->st_big) pffft_destroy_setup(s->st_big);
    for (float* p : {s->d_Hf, s->d_Hc, s->d_Hc_big, s->d_Hp, s->d_td, s->d_x, s->d_y}) if (p) (void)hipFree(p);
    if (s->st_part) pffft_destroy_setup(s->st_part);
    for (auto& kv : s->work) if (kv.second.p) (void)hipFree(kv.second.p);
    for (float* p : s->retired) if (p) (void)hipFree(p);
    if (s->d_split1_ab) (void)hipFree(s->d_split1_ab);
    if (s->d_fir32_hp) (void)hipFree(s->d_fir32_hp);
    if (s->d_fir32_hp_ref) (void)hipFree(s->d_fir32_hp_ref);
    for (float* p : {s->h_x, s->h_y}) if (p) (void)hipHostFree(p);
    s->magic = 0;
    delete s;
}

PF_EXPORT int pffastconv_hip_apply_device(PFFASTCONV_Setup* s, const float* d_input, int inputLen, float* d_output,
                                          int applyFlush, void* stream) {
    if (!s || s->magic != pf::FC_MAGIC) return -1;
    int produced = 0;
    int rc = pf::fc_apply_device(s, d_input, inputLen, d_output, applyFlush, (hipStream_t)stream, &produced);
    return rc ? -1 : produced;
}

PF_EXPORT int pffastconv_hip_apply_batch(PFFASTCONV_Setup* s, const float* d_input, int inputLen, size_t inputStride,
                                         float* d_output, size_t outputStride, int nsignals, int applyFlush, void* stream) {
    if (!s || s->magic != pf::FC_MAGIC || nsignals < 0) return -1;
    const size_t fl = (size_t)inputLen * ((s->flags & PFFASTCONV_HIP_CPLX_INP_OUT) ? 2 : 1);
    if (nsignals > 1 && (inputStride < fl || outputStride == 0)) { pf::g_last_error = "pffastconv_hip_apply_batch: stride smaller than a signal"; return -1; }
    int produced = 0;
    if (nsignals > 1) {   // rows of the output must not overlap: outputStride >= the floats one signal produces
        int lastOut = 0, prod = 0;
        (void)pf::fc_schedule(s, s->cplxFactor * inputLen, applyFlush, &lastOut, &prod);
        const size_t out_floats = (size_t)(prod / s->cplxFactor) * ((s->flags & PFFASTCONV_HIP_CPLX_INP_OUT) ? 2 : 1);
        if (outputStride < out_floats) { pf::g_last_error = "pffastconv_hip_apply_batch: outputStride smaller than the samples one signal produces"; return -1; }
    }
    if (nsignals == 0) {   // nothing to do, but the count a call would produce is still defined
        int lastOut = 0;
        (void)pf::fc_schedule(s, s->cplxFactor * inputLen, applyFlush, &lastOut, &produced);
        return produced / s->cplxFactor;
    }
    int rc = pf::fc_apply_device(s, d_input, inputLen, d_output, applyFlush, (hipStream_t)stream, &produced,
                                 pf::FcBatch{nsignals, inputStride, outputStride});
    return rc ? -1 : produced;
}

namespace pf {
static int fc_pinned(float** p, size_t* have, size_t want) {
    if (*have >= want) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *have = 0;
    PF_CHECK(hipHostMalloc((void**)p, want * sizeof(float), hipHostMallocDefault));
    *have = want;
    return 0;
}
}  // namespace pf

// Host pointers (the reference's calling convention): up to FC_ZC_LIMIT bytes per signal there is no DMA copy — the signal
// is copied by the CPU into a pinned host image that the kernel reads over PCIe directly, the kernel writes its outputs
// into another pinned image, one stream synchronisation, CPU copy out (the same scheme as the transform entries,
// pffft_hip.hip legacy_run).  Larger signals are staged through device buffers.  Failure: fail-soft like the transform
// entries (stderr, pffft_hip_last_error(), error counter) and the return value -1, abort() only under PFFFT_HIP_ABORT=1.
PF_EXPORT int pffastconv_apply(PFFASTCONV_Setup* s, const float* input, int cplxInputLen, float* output, int applyFlush) {
    using namespace pf;
    if (!s || s->magic != FC_MAGIC) {
        g_last_error = "pffastconv_apply: bad setup";
        legacy_fatal((int)hipErrorInvalidHandle, "pffastconv_apply", nullptr, 0, true);
        return -1;
    }
    constexpr size_t FC_ZC_LIMIT = (size_t)64 << 20;
    const bool in_dev = is_device_ptr(input), out_dev = is_device_ptr(output);
    const int cpl = (s->flags & PFFASTCONV_HIP_CPLX_INP_OUT) ? 2 : 1;
    const size_t in_floats = (size_t)cplxInputLen * cpl;
    const float* d_in = input; float* d_out = output;
    int rc = 0, produced = 0;
    const bool zc = zero_copy_enabled() && in_floats * sizeof(float) <= FC_ZC_LIMIT;
    bool out_pinned = false;
    do {
        if (!in_dev) {
            if (zc && fc_pinned(&s->h_x, &s->hx_floats, in_floats ? in_floats : 1) == 0) {
                if (in_floats) memcpy(s->h_x, input, in_floats * sizeof(float));
                d_in = s->h_x;
            } else {
                if ((rc = fc_grow(&s->d_x, &s->x_floats, in_floats ? in_floats : 1))) break;
                if (in_floats && hipMemcpy(s->d_x, input, in_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = -1; break; }
                d_in = s->d_x;
            }
        }
        if (!out_dev) {
            if (zc && fc_pinned(&s->h_y, &s->hy_floats, in_floats ? in_floats : 1) == 0) { d_out = s->h_y; out_pinned = true; }
            else {
                if ((rc = fc_grow(&s->d_y, &s->y_floats, in_floats ? in_floats : 1))) break;
                d_out = s->d_y;
            }
        }
        if ((rc = fc_apply_device(s, d_in, cplxInputLen, d_out, applyFlush, nullptr, &produced))) break;
        if (!out_dev && !out_pinned) {
            if (produced > 0 && hipMemcpy(output, d_out, (size_t)produced * cpl * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
        } else {
            if (hipStreamSynchronize(nullptr) != hipSuccess) { rc = -1; break; }
            if (out_pinned && produced > 0) memcpy(output, d_out, (size_t)produced * cpl * sizeof(float));
        }
    } while (0);
    if (rc) {
        // The reference cannot fail here.  Fail soft: NaN over the part of `output` the reference's contract guarantees to be
        // writable - inputLen - filterLen + 1 samples (include/pffft/pffastconv.h:159), never the whole input length - and
        // -1 as the return value: 0 would be indistinguishable from "not enough input yet" and a streaming loop waiting for
        // progress would spin forever (documented in include/pffft_hip.h).
        const int taps = s->cplxFactor == 2 ? (s->filterLen + 1) / 2 : s->filterLen;   // the caller's filter length
        const long writable = (long)cplxInputLen - taps + 1;
        legacy_fatal(rc, "pffastconv_apply", output, writable > 0 ? (size_t)writable * cpl * sizeof(float) : 0, !out_dev);
        return -1;
    }
    return produced;
}

PF_EXPORT void* pffastconv_malloc(size_t nb) { return pf::aligned_malloc64(nb); }
PF_EXPORT void pffastconv_free(void* p) { pf::aligned_free64(p); }
PF_EXPORT int pffastconv_simd_size(void) { return pf::SIMD; }
