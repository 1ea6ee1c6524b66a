// Host-side declarations shared by the translation units of libpffft_hip.so: the plan ("PFFFT_Setup":
// src/pffft_priv_impl.h:1051-1060), error plumbing and launch helpers.  pffft_hip.hip owns the definitions; dma_tu.hip
// (the LDS-DMA staged kernels, compiled on their own so that they can be iterated on in seconds) uses them.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <mutex>
#include <string>

#include "fft_generic.h"
#include "fft_big.h"
#include "stock_plan.h"
#include "pf_route.h"

namespace pf {

extern thread_local std::string g_last_error;
int fail(hipError_t e, const char* what);
#define PF_CHECK(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return fail(_e, #expr);    \
    } while (0)

int num_cus();

// is `st` recording into a HIP graph right now (hipStreamIsCapturing; errors read as "no")
bool stream_capturing(hipStream_t st);

// opt-in to more than 64 KiB of dynamic LDS: once per (kernel, size) - the attribute call sat on every launch
int allow_big_lds_impl(const void* kernel, size_t bytes);
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) return allow_big_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
    return 0;
}

// hipOccupancyMaxActiveBlocksPerMultiprocessor is constant per (kernel, threads, LDS bytes): asked once, then served from a table
// (it sat on the launch path of every tile pass and FIR call: pure host latency for the one-vector legacy entries)
int cached_occupancy(const void* kernel, int threads, size_t lds, int* per_cu);

enum Kernel { K_GENERIC = 0, K_C1024_F32 = 1, K_TILED = 2, K_BIG = 3 };
constexpr size_t LDS_MAX = 160 * 1024;
constexpr unsigned CTR_RING = 4096;
constexpr unsigned CTR_CAPTURED = 512;   // counter pairs set aside for launches recorded into a HIP graph (take_counters)

struct Setup {
    uint32_t magic;
    int N, transform, is_double;
    int n;           // complex length of the device transform
    size_t vec_scalars;  // scalars per vector: N (real) / 2N (complex)
    Kernel kernel;
    // what runs for [direction][ordered], decided at pffft_new_setup (plan_routes); pffft_hip_describe() prints it
    Route route[2][2];
    GenericPlan gp;
    int gthreads;
    size_t glds;
    // mixed-radix Stockham plans (fft_stock.h): [0] forward order, [1] backward order of the same radices
    StockPlan sk[2], skw[2];          // workgroup-phase plans; wave-local plans (small n)
    int sk_threads = 0, skw_threads = 0;
    bool sk_ok = false, skw_ok = false;
    // single-image plan (fft_one.h, round 6): the sizes that fill LDS once but not twice - [0] forward, [1] backward
    StockPlan one[3];          // [2]: the float complex backward transform from the internal layout (fft_one.h NARROW: no two-trip middle stages)
    bool one_ok = false;
    void* d_one_tw2 = nullptr; // compact twiddles of one[2]
    // device state (lazy: creating a setup never touches the GPU)
    std::mutex mu;        // guards the lazy device initialisation
    std::mutex stage_mu;  // guards the staging buffers of the legacy host-pointer entries
    bool dev_ready = false;
    // The device the tables / counters / scratch of THIS object live on: bound at first use.  A call from a thread whose current device is
    // another one is served by a replica of the plan with device state of its own (for_device; round 6): one PFFFT_Setup may be shared by
    // threads on different devices like the reference's immutable setup (include/pffft/pffft.h:102-105)
    std::atomic<int> device{-1};
    std::map<int, Setup*> replicas;   // (under mu) device key -> replica; owned, destroyed with the setup
    bool is_replica = false;
    void* d_tw = nullptr;   // W_n^j, j < n
    void* d_twr = nullptr;  // W_N^k, k <= n/2 (real only)
    void* d_twc[2] = {nullptr, nullptr};  // compact per-stage base twiddles of the Stockham plans (forward / backward order)
    unsigned* d_ctr = nullptr;             // ring of {next, done} work counters for the dynamic kernels (+ the captured region behind it)
    std::atomic<unsigned> ctr_slot{0}, cap_slot{0};
    // sizes beyond LDS with a small factor (fft_big.h, three streaming passes): n = bigR x sub->n
    int bigR = 0;
    Setup* sub = nullptr;
    // sizes beyond LDS (K_BIG): n = bigN[0] x bigN[1], one strided plan + twiddle table per factor
    StridedPlan bigp[2];
    void* d_bigtw[2] = {nullptr, nullptr};
    // HBM work buffers of the beyond-LDS path: one pair PER STREAM (kernels of one stream serialise; two streams running
    // the same setup concurrently must not share scratch).  big_mu is held while a call enqueues its passes (launch_big).
    // captured: a launch using these buffers was recorded into a HIP graph - a replay dereferences the pointers it froze, so such an
    // entry is never evicted and a buffer it outgrows is retired (freed with the setup) instead of freed
    struct Scratch { void* buf[2] = {nullptr, nullptr}; size_t bytes[2] = {0, 0}; unsigned long long last_use = 0; bool captured = false; };
    std::mutex retired_mu;
    std::vector<void*> retired;
    unsigned long long scratch_clock = 0;      // (under big_mu) orders the streams' last uses: the idlest one is evicted first
    std::mutex big_mu;
    std::map<hipStream_t, Scratch> big_scratch;
    // spectrum image of the composed pffft_hip_convolve_batch route: one per stream, conv_mu held while a call enqueues
    std::mutex conv_mu;
    std::map<hipStream_t, Scratch> conv_scratch;
    unsigned long long conv_clock = 0;
    void* d_stage[3] = {nullptr, nullptr, nullptr};  // staging for host-pointer legacy calls
    size_t stage_bytes[3] = {0, 0, 0};
    void* h_stage[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned host images the kernels read / write directly (small vectors)
    size_t hstage_bytes[4] = {0, 0, 0, 0};
};
constexpr uint32_t MAGIC = 0x50464654u;  // "PFFT"

// The {next, done} work-counter pairs of ONE launch of an in-order kernel (`pairs` consecutive pairs: the per-XCD kernels take five).  A
// direct launch takes the next slot of the setup's ring: reused CTR_RING launches later, i.e. at most CTR_RING launches of one setup in
// flight at once (include/pffft_hip.h).  A launch that is being RECORDED INTO A GRAPH freezes its counter address for every replay: it gets
// a slot of a region of its own, outside the ring, so that no later direct launch can share counters with a replay running on another
// stream (tiles skipped or run twice).  CTR_CAPTURED captured launches per setup have private counters; beyond that the region wraps, which
// is safe as long as the launches that share a slot do not run concurrently (nodes of one graph on one stream never do).
struct Setup;
unsigned* take_counters(Setup* s, hipStream_t st, unsigned pairs = 1);

// The object that holds `s`'s device state for the calling thread's CURRENT device: `s` itself on the device it is bound to (or binds to
// now), else the replica of that device, created on first use.  Every entry that takes a setup resolves it first; the launchers that read
// device pointers of a setup (d_tw, counters) are handed the resolved object.  NULL / foreign handles pass through (the entry reports them).
Setup* for_device(Setup* s);
// devices `s` holds state on right now (its own binding first): fills out[0 .. max), returns the count (pffft_hip_setup_devices)
int setup_devices(Setup* s, int* out, int max);

struct FcBatch { int nsig; size_t xstride, ystride; };   // signals of one pffastconv call (1 for the reference entries)

// dma_tu.hip: the LDS-DMA staged FIR block kernel (fft_dma.h); -1 when the block length has no such kernel
int launch_fir_dma(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                   hipStream_t st, const FcBatch& fb);

// dma_tu.hip: 16384-sample FIR blocks on 256 threads with 32 points per thread (fft_fir32.h); -1: not this block length
int launch_fir32(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                 hipStream_t st, const FcBatch& fb, void** hp_cache, int pref);

// tile_real_tu.hip: REAL transforms beyond LDS in two sweeps (fft_tile.h RMODE): N real points -> canonical half spectrum through a
// work buffer of tile_rfft_work_elems(N) complex elements per vector; -1: no plan for this length / direction
int launch_tile_rfft(Setup* s, const void* in, void* work, void* out, size_t batch, long long N, int dir, hipStream_t st);
bool tile_rfft_has_plan(long long N, bool is_double, bool adopted = true);   // adopted: only where it measured faster
size_t tile_rfft_work_elems(long long N, bool is_double);

// one_tu.hip: the single-image kernel (fft_one.h) - one HBM pass for the vectors between 80 and 144 KiB that have no two-image Stockham plan
bool one_build(int n, bool is_double, bool real, StockPlan out[2], size_t lds_max, bool narrow = false);
size_t one_lds_bytes(const StockPlan& p, bool is_double, bool real);
int launch_one(Setup* s, const void* in, void* out, size_t batch, int dir, int ordered, hipStream_t st);
const void* one_kernel_ptr(bool is_double, int flags);

// conv_tu.hip: forward -> x H (one filter spectrum, internal layout) -> backward in ONE kernel (fft_conv.h); -1: no fused kernel
int launch_conv_fused(Setup* s, const void* in, const void* H, void* out, size_t batch, double scaling, int accumulate, hipStream_t st);

// dma_tu.hip: few-block calls on reference-sized blocks (fft_split.h fastconv_split1_kernel); -1: no such kernel for this length
// (ab_cache: the filter's folded coefficient table, built on first use and owned by the caller's pffastconv setup: hipFree)
int launch_fir_split1(Setup* ps, const float* d_Hc, const float* d_x, float* d_y, int nblk, int step, int inputLen, int lastOut,
                      hipStream_t st, const FcBatch& fb, void** ab_cache);

// tile_tu.hip: power-of-two sizes beyond LDS in two / three passes (fft_tile.h); canonical complex, in -> out through `work`
// (same size, distinct from both; in may equal out).  -1 when the size has no tile plan.  layout 3 (round 6, forward only, the complex core
// of a REAL transform): the last pass carries the pair pass and stores the canonical half-complex spectrum.  layout 1 (forward only): the
// spectrum is stored in the pffft-internal layout by the last pass; layout 2 (backward only): it is read from that layout
// by the first pass.
// (n complex points; -1 = no tile plan for n.  tile_has_plan: the same answer without launching)
// deep: the streaming route of this n would take five sweeps - three tile passes are allowed
int launch_tile_fft(Setup* s, const void* in, void* work, void* out, size_t batch, long long n, int dir, hipStream_t st, int layout = 0, int mode = 0);
bool tile_has_plan(long long n, bool is_double, int mode = 0);
bool tile_real_rows_plan(long long n, bool is_double, int mode);   // layout 3 of launch_tile_fft: real forward, the pair pass inside the last (row) pass   // mode: 0 complex / three streaming sweeps, 1 five (deep), 2 real core / three
int tile_plan_lengths(long long n, bool is_double, int mode, int lengths[3]);
int tile_plan_candidates(long long n, bool is_double, int* out, int max);   // every legal pair {L1, gen1, L2, gen2, model cost}
int tile_plan_override(long long n, bool is_double, int l1, int g1, int l2, int g2);   // the tuner's hook (tools/tune_tile_plans.py)
int tile_plan_layouts(long long n, bool is_double, int mode);   // bit 0: internal layout out of the last pass, bit 1: into the first   // 0 / 2 / 3 passes (pffft_hip_tile_plan)

}  // namespace pf
