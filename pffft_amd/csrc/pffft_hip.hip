// libpffft_hip.so — host side (planner, dispatch, C ABI) of the MI355X pffft drop-in.
// The ABI is declared in include/pffft_hip.h; each entry cites the reference line it replaces.
// There is NO CPU arithmetic path here: every transform runs as a HIP kernel on gfx950.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pffft_hip.h"
#include "pf_host.h"
#include "fft_c1024.h"
#include "fft_generic.h"
#include "fft_tiled.h"
#include "fft_big.h"
#include "fft_stock.h"
#include "stock_plan.h"
#include "stock_ct.h"
#include "stock_df_gen.h"
#include "stock_grid_gen.h"
#include "fft_aux.h"
#include "fft_tiny.h"
#include "pfdsp_mix.h"

namespace pf {

#ifdef PFFFT_HIP_VARIANTS
constexpr bool PF_HAS_VARIANTS = true;
#else
constexpr bool PF_HAS_VARIANTS = false;
#endif

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_last_error;
static thread_local int g_ab_raw = 0;   // pffft_hip_set_variant(): per calling thread
AbSel ab() { AbSel a; a.raw = g_ab_raw; return a; }

// every environment switch of the product build, read once (pf_route.h)
const Env& env() {
    static const Env e = [] {
        auto num = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        Env r;
        r.abort_on_error = num("PFFFT_HIP_ABORT", 0) == 1;
        r.zero_copy = num("PFFFT_HIP_NO_ZEROCOPY", 0) != 1;
        r.oneshot = num("PFFFT_HIP_ONESHOT", 4);
        r.c1024_rounds = num("PFFFT_HIP_C1024_ONCE", 4);
        r.tile_plans = num("PFFFT_HIP_TILE_PLANS", 1);
        r.tile_force = getenv("PFFFT_HIP_TILE_FORCE");
        r.fir_nfft = num("PFFASTCONV_HIP_NFFT", -1);
        r.fir_xcd = num("PFFASTCONV_HIP_XCD", 1);
        if (r.oneshot < 0) r.oneshot = 0;
        if (r.c1024_rounds < 0) r.c1024_rounds = 0;
        return r;
    }();
    return e;
}

int fail(hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "pffft_hip: %s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
    g_last_error = buf;
    return (int)e;
}

// Legacy void entries have no error channel (include/pffft/pffft.h:159).  A drop-in must not kill its caller where the
// reference could not fail: the default is FAIL-SOFT — one line on stderr (the first 8 failures per process, then every
// 2^k-th: a long-running caller never goes fully silent), the text in pffft_hip_last_error(), the failure counted in
// pffft_hip_error_count(), and the output vector filled with NaN (all-ones bytes; host or device memory alike) so that a
// failed call can never be mistaken for a spectrum.  A call on an INVALID HANDLE (null, destroyed, wrong precision) writes
// nothing: the vector length would have to be read from the very object that failed validation.
// PFFFT_HIP_ABORT=1 restores fail-fast (abort()).
static std::atomic<unsigned> g_error_count{0};
static bool abort_on_error() { return env().abort_on_error; }
static void legacy_fatal(int code, const char* entry, void* out, size_t out_bytes, bool out_is_host) {
    const unsigned nth = g_error_count.fetch_add(1);
    const unsigned seq = nth + 1;
    if (nth < 8 || (seq & (seq - 1)) == 0 || abort_on_error())
        fprintf(stderr, "%s: HIP path failed (%d) [failure #%u of this process]: %s%s\n", entry, code, seq, g_last_error.c_str(),
                abort_on_error() ? "" : (out && out_bytes) ? " -- output filled with NaN (PFFFT_HIP_ABORT=1 aborts instead)"
                                                          : " -- output left untouched (PFFFT_HIP_ABORT=1 aborts instead)");
    if (abort_on_error()) abort();
    if (out && out_bytes) {
        if (out_is_host) memset(out, 0xFF, out_bytes);  // all-ones = NaN pattern
        else if (hipMemset(out, 0xFF, out_bytes) != hipSuccess) (void)hipGetLastError();
    }
}

// ------------------------------------------------------------------------------------------------
// size helpers — semantics of src/pffft_priv_impl.h:76-116 and src/pffft_common.c:25-43
// ------------------------------------------------------------------------------------------------
constexpr int SIMD = 4;  // the internal layout is the reference's SIMD_SZ == 4 layout (SURVEY.md finding 2)

static int min_fft_size(int transform) {
    if (transform == PFFFT_REAL) return 2 * SIMD * SIMD;
    if (transform == PFFFT_COMPLEX) return SIMD * SIMD;
    return 1;
}
static int is_valid_size(int N, int transform) {
    const int nmin = min_fft_size(transform);
    int r = N;
    while (r >= 5 * nmin && r % 5 == 0) r /= 5;
    while (r >= 3 * nmin && r % 3 == 0) r /= 3;
    while (r >= 2 * nmin && r % 2 == 0) r /= 2;
    return r == nmin;
}
static int nearest_size(int N, int transform, int higher) {
    const int nmin = min_fft_size(transform);
    if (N < nmin) N = nmin;
    const int d = higher ? nmin : -nmin;
    N = higher ? nmin * ((N + nmin - 1) / nmin) : nmin * (N / nmin);
    for (;; N += d)
        if (is_valid_size(N, transform)) return N;
}
static int next_pow2(int N) {
    unsigned v = (unsigned)N;
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return (int)(v + 1);
}
static int is_pow2(int N) { return N && !(N & (N - 1)); }

static void* aligned_malloc64(size_t nb) {  // src/pffft_common.c:12-22: 64-byte aligned, raw pointer kept at p[-1]
    void* p0 = malloc(nb + 63 + sizeof(void*));
    if (!p0) return nullptr;
    uintptr_t p = ((uintptr_t)p0 + 63 + sizeof(void*)) & ~(uintptr_t)63;
    ((void**)p)[-1] = p0;
    return (void*)p;
}
static void aligned_free64(void* p) {
    if (p) free(((void**)p)[-1]);
}

// ------------------------------------------------------------------------------------------------
// the plan ("PFFFT_Setup": src/pffft_priv_impl.h:1051-1060)
// ------------------------------------------------------------------------------------------------

static void destroy_setup(Setup* s);
static void plan_routes(Setup* s);

// does the canonical forward transform of this (sub-)setup run on one of the fast kernels: register-tiled power-of-two
// sizes, or a Stockham plan that exists as a compile-time constant (the run-time-plan twin runs at 0.3)
static bool sub_is_fast(const Setup* q) {
    if (q->kernel == K_TILED || q->kernel == K_C1024_F32) return true;
    if (q->kernel != K_GENERIC) return false;
    const int flags = q->transform == PFFFT_REAL ? 8 : 0;   // forward, canonical layouts (fft_stock.h)
    // (the product build instantiates ONE of the two variants of a plan - deposit or direct first stage, bit 4 -, see
    //  tools/gen_stock_plans.hip: a plan exists when either is there)
    auto has = [&](const StockPlan& p, bool wl) {
        if (q->is_double) return stock_ct_lookup(p, flags, wl, (const double*)nullptr) || stock_ct_lookup(p, flags | 16, wl, (const double*)nullptr);
        return stock_ct_lookup(p, flags, wl, (const float*)nullptr) || stock_ct_lookup(p, flags | 16, wl, (const float*)nullptr);
    };
    return (q->skw_ok && has(q->skw[0], true)) || (q->sk_ok && has(q->sk[0], false));
}

static Setup* new_setup(int N, int transform, int is_double) {
    // validation: src/pffft_priv_impl.h:1066-1078 and :1105-1109
    if (N <= 0 || N > (1 << 26)) return nullptr;
    if (transform != PFFFT_REAL && transform != PFFFT_COMPLEX) return nullptr;
    const int mult = transform == PFFFT_REAL ? 2 * SIMD * SIMD : SIMD * SIMD;
    if (N % mult) return nullptr;
    {   // N / SIMD must factor into 2, 3, 5 (what decompose() + the product check enforce)
        int r = N / SIMD;
        for (int f : {2, 3, 5}) while (r % f == 0) r /= f;
        if (r != 1) return nullptr;
    }
    Setup* s = new Setup();
    s->magic = MAGIC;
    s->N = N; s->transform = transform; s->is_double = is_double;
    s->n = transform == PFFFT_REAL ? N / 2 : N;
    s->vec_scalars = transform == PFFFT_REAL ? (size_t)N : 2 * (size_t)N;
    // radix schedule for the in-place DIF kernel: 5s, 3s, then 4s, then at most one 2
    GenericPlan& gp = s->gp;
    memset(&gp, 0, sizeof gp);
    gp.n = s->n; gp.is_real = transform == PFFFT_REAL;
    int r = s->n, ns = 0;
    while (r % 5 == 0) { gp.radix[ns++] = 5; r /= 5; }
    while (r % 3 == 0) { gp.radix[ns++] = 3; r /= 3; }
    while (r % 4 == 0) { gp.radix[ns++] = 4; r /= 4; }
    if (r % 2 == 0) { gp.radix[ns++] = 2; r /= 2; }
    gp.nstages = ns;
    const size_t esz = is_double ? 16 : 8;
    gp.G = s->n >= 2048 ? 1 : 2048 / s->n;
    s->glds = (((size_t)gp.G * s->n + ((size_t)gp.G * s->n >> 5) + 2) * esz + 15) / 16 * 16 + 16;  // padded image (gpad)
    int th = (int)(((size_t)gp.G * s->n / 8 + 63) / 64 * 64);
    s->gthreads = th < 64 ? 64 : (th > 1024 ? 1024 : th);
    s->kernel = K_GENERIC;
    // mixed-radix Stockham plans (fft_stock.h) for every size whose two exchange images fit LDS
    {
        bool wl = false;
        s->sk_ok = sk_build(s->n, is_double != 0, transform == PFFFT_REAL, s->sk, &s->sk_threads, &wl, false, LDS_MAX);
        s->skw_ok = sk_build(s->n, is_double != 0, transform == PFFFT_REAL, s->skw, &s->skw_threads, &wl, true, LDS_MAX) && wl;
    }
    const bool pow2_tiled = (s->n & (s->n - 1)) == 0 && s->n >= 16 && s->n <= 16384 && (size_t)s->n * esz <= 128 * 1024;
    // sizes with ONE image in LDS (two do not fit: complex float n = 9600 .. 20480): as R x N2 with the rows on a fast kernel
    // the three streaming passes below measure 0.17-0.24
    // (the same holds for a size whose Stockham plan fits but has no compile-time twin: the run-time-plan kernel measured 0.07
    //  at n = 9600)
    const bool single_image = s->glds <= LDS_MAX && !pow2_tiled && !sub_is_fast(s) && s->n >= 2048;
    if (s->glds > LDS_MAX || single_image) {
        // four-step plan: split the prime factors of n into two balanced products
        s->kernel = K_BIG;
        // ... unless n = R x N2 with a register-sized R and an N2 the LDS-resident batched kernels take (fft_big.h)
        // (the first R whose N2 runs on a fast kernel: N = 20480 as 4 x 5120 put the rows on the run-time-plan Stockham
        //  kernel and measured 0.04 of the roofline, as 32 x 640 it has a compile-time plan)
        // (multiples of 4 first: their column pass can read the internal layout itself, big_col_int_kernel)
        for (int R : {4, 8, 12, 16, 32, 2, 3, 5, 6, 9, 10, 15, 25, 27}) {
            if (s->n % R) continue;
            const int N2 = s->n / R;
            if ((size_t)N2 * esz > 80 * 1024 || N2 % (SIMD * SIMD)) continue;   // (Stockham plans reach n = 10000 float)
            Setup* sub = new_setup(N2, PFFFT_COMPLEX, is_double);
            if (!sub) continue;
            if (sub->kernel == K_BIG || (sub->kernel == K_GENERIC && !sub->sk_ok)) { destroy_setup(sub); continue; }
            const bool fast = sub_is_fast(sub);
            if (s->sub && !fast) { destroy_setup(sub); continue; }      // keep the first usable one as the fallback
            if (s->sub) destroy_setup(s->sub);
            s->sub = sub; s->bigR = R;
            if (fast) break;
        }
        if (single_image && !(s->sub && sub_is_fast(s->sub))) {
            // no factorisation with fast rows: the balanced two-pass strided plan below (launch_strided) - decided HERE, at setup
            // time (the in-place kernel this branch used to fall back to is gone: such a setup would have been created and then
            // failed on every transform).  No legal size reaches this today (tests/test_generated_sources.py: every size has a
            // compile-time plan); the route exists so that a gap in the generated tables costs speed, not correctness.
            if (s->sub) destroy_setup(s->sub);
            s->sub = nullptr; s->bigR = 0;
        }
        // larger still: peel the largest register-sized factor and recurse (n = R x (R' x N2')): five passes, seven, ...
        if (!s->sub && s->kernel == K_BIG) {
            for (int R : {32, 16, 15, 12, 10, 8, 6, 5, 4, 3, 2}) {
                if (s->n % R) continue;
                const int N2 = s->n / R;
                if (N2 % (SIMD * SIMD)) continue;
                Setup* sub = new_setup(N2, PFFFT_COMPLEX, is_double);
                if (!sub) continue;
                if (!(sub->kernel == K_BIG && sub->bigR)) { destroy_setup(sub); continue; }
                s->sub = sub; s->bigR = R;
                break;
            }
        }
        std::vector<int> f;
        int rr = s->n;
        for (int q : {5, 3, 2}) while (rr % q == 0) { f.push_back(q); rr /= q; }
        long long a = 1, b = 1;
        for (int q : f) { if (a <= b) a *= q; else b *= q; }
        const long long sub[2] = {a, b};
        for (int i = 0; i < 2; ++i) {
            StridedPlan& sp = s->bigp[i];
            memset(&sp, 0, sizeof sp);
            sp.n = (int)sub[i];
            int r2 = sp.n, k = 0;
            while (r2 % 5 == 0) { sp.radix[k++] = 5; r2 /= 5; }
            while (r2 % 3 == 0) { sp.radix[k++] = 3; r2 /= 3; }
            while (r2 % 4 == 0) { sp.radix[k++] = 4; r2 /= 4; }
            if (r2 % 2 == 0) { sp.radix[k++] = 2; r2 /= 2; }
            sp.nstages = k;
            long long g = (long long)(92 * 1024) / ((long long)sp.n * (long long)esz);
            sp.G = (int)(g < 1 ? 1 : (g > 32 ? 32 : g));
            sp.vec = s->n;
        }
        // step A: columns (length N1 = sub[0], stride N2), twiddled;  step B: rows (length N2), transposed store
        s->bigp[0].count = sub[1]; s->bigp[0].estride_in = sub[1]; s->bigp[0].tstride_in = 1;
        s->bigp[0].estride_out = sub[1]; s->bigp[0].tstride_out = 1; s->bigp[0].twN = s->n;
        s->bigp[1].count = sub[0]; s->bigp[1].estride_in = 1; s->bigp[1].tstride_in = sub[1];
        s->bigp[1].estride_out = sub[0]; s->bigp[1].tstride_out = 1; s->bigp[1].twN = 0;
    }
    if (!is_double && transform == PFFFT_COMPLEX && N == 1024) s->kernel = K_C1024_F32;
    else if ((s->n & (s->n - 1)) == 0 && s->n >= 16 && s->n <= 16384 &&
             (size_t)s->n * esz <= 128 * 1024)
        s->kernel = K_TILED;  // power-of-two sizes: register-tiled kernels (fft_tiled.h)
    // every other size that fits: mixed-radix Stockham kernel (fft_stock.h); the in-place kernel of
    // fft_generic.h keeps the sizes whose two images exceed LDS
    if (s->kernel == K_BIG) s->sk_ok = s->skw_ok = false;
    // ... of those, the vectors that fill LDS once (80-144 KiB): ONE pass on the single-image kernel (fft_one.h, round 6); the plan above stays
    // as the route of the helpers (zreorder, zconvolve) and of AB_NO_ONE_IMAGE
    if (s->kernel == K_BIG) {
        s->one_ok = one_build(s->n, is_double != 0, transform == PFFFT_REAL, s->one, LDS_MAX);
        if (s->one_ok) {
            StockPlan narrow[2];
            if (one_build(s->n, is_double != 0, transform == PFFFT_REAL, narrow, LDS_MAX, true)) s->one[2] = narrow[1];
            else s->one_ok = false;
        }
    }
    if (s->kernel == K_GENERIC && !PF_HAS_VARIANTS && !sub_is_fast(s)) {
        // product build: only compile-time Stockham plans exist.  A size without one (none today) must not yield a setup whose
        // every transform fails: refuse it here, where the reference reports unsupported sizes too (NULL, src/pffft_priv_impl.h:1105-1109)
        g_last_error = "pffft_hip: no kernel plan for this size in the product build";
        destroy_setup(s);
        return nullptr;
    }
    plan_routes(s);       // what runs for every (direction, layout): decided here, printed by pffft_hip_describe()
    for (int d = 0; d < 2; ++d)
        for (int o = 0; o < 2; ++o)
            if (s->route[d][o].fam == FAM_NONE) {
                g_last_error = "pffft_hip: no kernel plan for this size";
                destroy_setup(s);
                return nullptr;
            }
    return s;
}

static void destroy_setup(Setup* s) {
    if (!s) return;
    for (auto& kv : s->replicas) destroy_setup(kv.second);   // (hipFree takes a pointer of any device, whatever the current one is)
    if (s->dev_ready) {
        if (s->d_tw) (void)hipFree(s->d_tw);
        if (s->d_twr) (void)hipFree(s->d_twr);
        for (void* q : s->d_twc) if (q) (void)hipFree(q);
        if (s->d_one_tw2) (void)hipFree(s->d_one_tw2);
    }
    if (s->d_ctr) (void)hipFree(s->d_ctr);
    if (s->sub) destroy_setup(s->sub);
    for (void* p : s->d_bigtw) if (p) (void)hipFree(p);
    for (auto& kv : s->big_scratch) for (void* p : kv.second.buf) if (p) (void)hipFree(p);
    for (auto& kv : s->conv_scratch) for (void* p : kv.second.buf) if (p) (void)hipFree(p);
    for (void* p : s->retired) if (p) (void)hipFree(p);
    for (void* p : s->d_stage) if (p) (void)hipFree(p);
    for (void* p : s->h_stage) if (p) (void)hipHostFree(p);
    s->magic = 0;
    delete s;
}

// The key a thread's current device goes by: the HIP device index; AB_FAKE_DEVICE moves the calling thread to a key of its own (the
// replica path on a one-GPU box).
static int current_device_key(int* key) {
    int dev = -1;
    PF_CHECK(hipGetDevice(&dev));
    *key = ab().is(AB_FAKE_DEVICE) ? dev + 64 : dev;
    return 0;
}

// One setup, any device (round 6; the reference's setup is immutable and shareable, include/pffft/pffft.h:102-105).  The object binds
// to the device of its first use; every other device gets a replica of the plan - the same (N, transform, precision), hence the same
// routes - with tables, counter ring, per-stream scratch and staging of its own, built lazily under the setup's mutex and destroyed with it.
Setup* for_device(Setup* s) {
    if (!s || s->magic != MAGIC || s->is_replica) return s;
    int key = -1;
    if (current_device_key(&key)) { (void)hipGetLastError(); return s; }   // no usable device: the entry's own checks report it
    if (s->device.load(std::memory_order_acquire) == key) return s;
    std::lock_guard<std::mutex> lk(s->mu);
    int bound = s->device.load(std::memory_order_relaxed);
    if (bound < 0) { s->device.store(key, std::memory_order_release); return s; }
    if (bound == key) return s;
    auto it = s->replicas.find(key);
    if (it != s->replicas.end()) return it->second;
    Setup* r = new_setup(s->N, s->transform, s->is_double);
    if (!r) return s;                                                       // (cannot happen: the same arguments made `s`)
    r->is_replica = true;
    r->device.store(key, std::memory_order_release);
    s->replicas[key] = r;
    return r;
}

int setup_devices(Setup* s, int* out, int max) {
    if (!s || s->magic != MAGIC) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    int n = 0;
    const int bound = s->device.load();
    if (bound >= 0) { if (out && n < max) out[n] = bound; ++n; }
    for (auto& kv : s->replicas) { if (out && n < max) out[n] = kv.first; ++n; }
    return n;
}

// The device state of an object lives on ONE device; for_device() hands every entry the object of the calling thread's device, so a
// mismatch here means the caller switched devices between resolving and launching (or handed a replica around): an error, not a fault
// inside a kernel.
static int check_device(Setup* s) {
    int key = -1;
    int rc = current_device_key(&key);
    if (rc) return rc;
    int bound = s->device.load();
    if (bound < 0) { s->device.store(key); bound = key; }
    if (bound != key) {
        char buf[160];
        snprintf(buf, sizeof buf, "pffft_hip: setup state is bound to device %d but the calling thread's current device is %d",
                 bound, key);
        g_last_error = buf;
        return (int)hipErrorInvalidDevice;
    }
    return 0;
}

static int alloc_counter_ring(Setup* s) {
    // Each launch of a dynamic kernel takes its own {next, done} counter pair from this ring; the
    // kernel re-arms the pair when its last workgroup retires.  A pair is reused only CTR_RING
    // launches later, i.e. at most CTR_RING launches of one setup may be in flight at once
    // (stated in include/pffft_hip.h; launches on one stream serialise, so this bounds concurrent streams x depth).
    if (s->d_ctr) return 0;
    // (+ 16 after each region: a launch may take several consecutive pairs from the last slot)
    constexpr size_t words = 2 * (size_t)CTR_RING + 16 + 2 * (size_t)CTR_CAPTURED + 16;
    PF_CHECK(hipMalloc((void**)&s->d_ctr, sizeof(unsigned) * words));
    PF_CHECK(hipMemset(s->d_ctr, 0, sizeof(unsigned) * words));
    return 0;
}

bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusActive;
}

unsigned* take_counters(Setup* s, hipStream_t st, unsigned pairs) {
    if (stream_capturing(st)) return s->d_ctr + 2 * (size_t)CTR_RING + 16 + 2 * (s->cap_slot.fetch_add(pairs) % CTR_CAPTURED);
    return s->d_ctr + 2 * (s->ctr_slot.fetch_add(pairs) % CTR_RING);
}

template <typename T>
static int ensure_device(Setup* s) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->dev_ready) return check_device(s);
    int rc = check_device(s);
    if (rc) return rc;
    if ((rc = alloc_counter_ring(s))) return rc;   // every kernel family may take the in-order path (zconvolve on K_BIG too)
    if (s->kernel == K_BIG) {
        for (int i = 0; i < 2; ++i) {
            const int m = s->bigp[i].n;
            if (((size_t)m + (m >> 5) + 2) * sizeof(cx<T>) > LDS_MAX) {
                g_last_error = "pffft_hip: N too large even for the four-step path in this precision";
                return (int)hipErrorInvalidValue;
            }
            std::vector<cx<T>> tw(m);
            for (int j = 0; j < m; ++j) {
                long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)m;
                tw[j].x = (T)cosl(a); tw[j].y = (T)sinl(a);
            }
            PF_CHECK(hipMalloc(&s->d_bigtw[i], sizeof(cx<T>) * m));
            PF_CHECK(hipMemcpy(s->d_bigtw[i], tw.data(), sizeof(cx<T>) * m, hipMemcpyHostToDevice));
        }
        if (s->one_ok) {   // single-image kernel: compact base twiddles per direction, W_N^k of the pair pass
            const long double PI2 = 2.0L * 3.14159265358979323846264338327950288L;
            for (int d = 0; d < 3; ++d) {
                const StockPlan& sp = s->one[d];
                std::vector<cx<T>> tc(sp.ctab + 1);
                for (int st = 1; st < sp.ns; ++st) {
                    const StockStage& g = sp.st[st];
                    for (int jm = 0; jm < g.Ns; ++jm) {   // W_{Ns R}^jm
                        const long double a = -PI2 * (long double)jm / (long double)(g.Ns * g.R);
                        tc[g.tw_off + jm].x = (T)cosl(a); tc[g.tw_off + jm].y = (T)sinl(a);
                    }
                }
                void** dst = d < 2 ? &s->d_twc[d] : &s->d_one_tw2;
                PF_CHECK(hipMalloc(dst, sizeof(cx<T>) * tc.size()));
                PF_CHECK(hipMemcpy(*dst, tc.data(), sizeof(cx<T>) * tc.size(), hipMemcpyHostToDevice));
            }
            if (s->transform == PFFFT_REAL) {
                const int m = s->n / 2 + 1;
                std::vector<cx<T>> twr(m);
                for (int k = 0; k < m; ++k) {
                    const long double a = -PI2 * (long double)k / (long double)(2 * s->n);
                    twr[k].x = (T)cosl(a); twr[k].y = (T)sinl(a);
                }
                PF_CHECK(hipMalloc(&s->d_twr, sizeof(cx<T>) * m));
                PF_CHECK(hipMemcpy(s->d_twr, twr.data(), sizeof(cx<T>) * m, hipMemcpyHostToDevice));
            }
        }
        s->dev_ready = true;
        return 0;
    }
    const int n = s->n;
    std::vector<cx<T>> tw(n);
    for (int j = 0; j < n; ++j) {  // tables are generated in extended precision and rounded once
        long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)n;
        tw[j].x = (T)cosl(a); tw[j].y = (T)sinl(a);
    }
    PF_CHECK(hipMalloc(&s->d_tw, sizeof(cx<T>) * n));
    PF_CHECK(hipMemcpy(s->d_tw, tw.data(), sizeof(cx<T>) * n, hipMemcpyHostToDevice));
    if (s->transform == PFFFT_REAL) {
        const int m = n / 2 + 1;
        std::vector<cx<T>> twr(m);
        for (int k = 0; k < m; ++k) {
            long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)(2 * n);
            twr[k].x = (T)cosl(a); twr[k].y = (T)sinl(a);
        }
        PF_CHECK(hipMalloc(&s->d_twr, sizeof(cx<T>) * m));
        PF_CHECK(hipMemcpy(s->d_twr, twr.data(), sizeof(cx<T>) * m, hipMemcpyHostToDevice));
    }
    for (int d = 0; d < 2; ++d) {
        const StockPlan* sp = (s->sk_ok && s->sk[d].twmode == 2) ? &s->sk[d] : (s->skw_ok && s->skw[d].twmode == 2) ? &s->skw[d] : nullptr;
        if (!sp) continue;
        std::vector<cx<T>> tc(sp->ctab + 1);
        for (int st = 1; st < sp->ns; ++st) {
            const StockStage& g = sp->st[st];
            for (int jm = 0; jm < g.Ns; ++jm) {   // W_{Ns R}^jm
                long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)jm / (long double)(g.Ns * g.R);
                tc[g.tw_off + jm].x = (T)cosl(a); tc[g.tw_off + jm].y = (T)sinl(a);
            }
        }
        PF_CHECK(hipMalloc(&s->d_twc[d], sizeof(cx<T>) * tc.size()));
        PF_CHECK(hipMemcpy(s->d_twc[d], tc.data(), sizeof(cx<T>) * tc.size(), hipMemcpyHostToDevice));
    }
    s->dev_ready = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// (both tables are per DEVICE: the attribute is a property of the function on one device, and one process may drive several)
int allow_big_lds_impl(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;           // largest size already granted per (device, kernel)
    int dev = 0;
    PF_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(dev, kernel);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return 0;
    PF_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[key] = bytes;
    return 0;
}

int cached_occupancy(const void* kernel, int threads, size_t lds, int* per_cu) {
    struct Key {
        int dev; const void* k; int th; size_t lds;
        bool operator<(const Key& o) const {
            return dev != o.dev ? dev < o.dev : k != o.k ? k < o.k : th != o.th ? th < o.th : lds < o.lds;
        }
    };
    static std::mutex mu;
    static std::map<Key, int> tab;
    int dev = 0;
    PF_CHECK(hipGetDevice(&dev));
    const Key key{dev, kernel, threads, lds};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = tab.find(key);
        if (it != tab.end()) { *per_cu = it->second; return 0; }
    }
    int v = 0;
    PF_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, threads, lds));
    if (v < 1) v = 1;
    std::lock_guard<std::mutex> lk(mu);
    tab[key] = v;
    *per_cu = v;
    return 0;
}

int num_cus() {
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int W>
static int launch_c1024_once(Setup* s, const float* in, float* out, size_t batch, int dir, int ordered, hipStream_t st) {
    const unsigned grid = (unsigned)((batch + W - 1) / W);
    const size_t lds = (size_t)W * C1024_WAVE_BYTES;
    const cx<float>* tw = (const cx<float>*)s->d_tw;
#define PF_LAUNCH_C1024_ONCE(D, I, O)                                                                 \
    do {                                                                                              \
        auto k = fft_c1024_f32_once_kernel<D, I, O, W>;                                               \
        int rc = allow_big_lds(k, lds);                                                               \
        if (rc) return rc;                                                                            \
        hipLaunchKernelGGL(k, dim3(grid), dim3(W * 64), lds, st, in, out, (unsigned)batch, tw);       \
    } while (0)
    if (dir == PFFFT_FORWARD) {
        if (ordered) PF_LAUNCH_C1024_ONCE(FWD, 0, 0); else PF_LAUNCH_C1024_ONCE(FWD, 0, 1);
    } else {
        if (ordered) PF_LAUNCH_C1024_ONCE(BWD, 0, 0); else PF_LAUNCH_C1024_ONCE(BWD, 1, 0);
    }
#undef PF_LAUNCH_C1024_ONCE
    PF_CHECK(hipGetLastError());
    return 0;
}

// resident wavefronts per CU of the short-launch kernel: 4 workgroups x C1024_ONCE_W wavefronts (35 KiB of LDS and 84 VGPRs each)
constexpr int C1024_ONCE_W = 4, C1024_ONCE_RESIDENT = 16;

static int launch_c1024(Setup* s, const Route& r, const float* in, float* out, size_t batch, int dir, int ordered, hipStream_t st) {
    // Short launches - up to r.oneshot resident sets of 16 wavefronts per CU - run one transform per wavefront in dispatch order
    // (fft_c1024.h once kernel; tools/r5_c1024_batch.py, us per launch, loop -> once: batch 2^10 12.5 -> 7.7, 2^11 19.3 -> 8.2, 2^12 24.1 -> 14.0,
    // 2^13 34.8 -> 27.6, 2^14 55.2 -> 50.8; from 2^15 on the loop wins: 96 against 99 us; 2- and 8-wavefront workgroups within 1 us of these).
    if (r.oneshot > 0 && batch <= (size_t)r.oneshot * C1024_ONCE_RESIDENT * (size_t)num_cus())
        return launch_c1024_once<C1024_ONCE_W>(s, in, out, batch, dir, ordered, st);
    // the persistent in-order loop: ONE 8-wavefront workgroup per CU (two per CU measured 2^12 .. 2^17 10-50 % slower, 2^18 on equal)
    const unsigned wgs_needed = (unsigned)((batch + C1024_WAVES - 1) / C1024_WAVES);
    unsigned grid = (unsigned)num_cus();
    if (grid > wgs_needed) grid = wgs_needed;
    const dim3 blk(C1024_WAVES * 64);
    const size_t lds = C1024_LDS_BYTES;
    const cx<float>* tw = (const cx<float>*)s->d_tw;
    const unsigned b = (unsigned)batch;
    unsigned* ctr = take_counters(s, st);
#define PF_LAUNCH_C1024(D, I, O)                                                                      \
    do {                                                                                              \
        auto k = fft_c1024_f32_dyn_kernel<D, I, O>;                                                   \
        int rc = allow_big_lds(k, lds);                                                               \
        if (rc) return rc;                                                                            \
        hipLaunchKernelGGL(k, dim3(grid), blk, lds, st, in, out, b, tw, ctr);                         \
    } while (0)
    if (dir == PFFFT_FORWARD) {
        if (ordered) PF_LAUNCH_C1024(FWD, 0, 0); else PF_LAUNCH_C1024(FWD, 0, 1);
    } else {
        if (ordered) PF_LAUNCH_C1024(BWD, 0, 0); else PF_LAUNCH_C1024(BWD, 1, 0);
    }
#undef PF_LAUNCH_C1024
    PF_CHECK(hipGetLastError());
    return 0;
}

// forward transform of the frequency-shifted stream (fused mixer, fft_c1024.h C1024Mix)
static int launch_c1024_mix(Setup* s, const float* in, float* out, size_t batch, int ordered, double step_turns,
                            double phase_turns, hipStream_t st) {
    const unsigned wgs_needed = (unsigned)((batch + C1024_WAVES - 1) / C1024_WAVES);
    unsigned grid = (unsigned)num_cus();
    if (grid > wgs_needed) grid = wgs_needed;
    const dim3 blk(C1024_WAVES * 64);
    const size_t lds = C1024_LDS_BYTES;
    const cx<float>* tw = (const cx<float>*)s->d_tw;
    const unsigned b = (unsigned)batch;
    unsigned* ctr = take_counters(s, st);
    C1024Mix mix;
    step_turns -= std::rint(step_turns);
    phase_turns -= std::rint(phase_turns);
    mix.step = step_turns;
    mix.phase0 = phase_turns;
    for (int j = 1; j < 8; ++j) {
        double a = step_turns * 128.0 * j;
        a -= std::rint(a);
        mix.g[j - 1][0] = (float)std::cos(pfmix::MIX_TWO_PI * a);
        mix.g[j - 1][1] = (float)std::sin(pfmix::MIX_TWO_PI * a);
    }
    if (ordered) {
        auto k = fft_c1024_f32_mix_kernel<0>;
        int rc = allow_big_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), blk, lds, st, in, out, b, tw, ctr, mix);
    } else {
        auto k = fft_c1024_f32_mix_kernel<1>;
        int rc = allow_big_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), blk, lds, st, in, out, b, tw, ctr, mix);
    }
    PF_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// register-tiled power-of-two family (fft_tiled.h): which configuration serves (n, direction, transform, layout)
// ------------------------------------------------------------------------------------------------
template <typename T>
using TiledFn = void (*)(const T*, T*, unsigned, int, const cx<T>*, const cx<T>*, unsigned*);

template <typename T, class C>
static TiledSel tiled_sel(int dir, int real, const char* name) {
    TiledSel e;
    e.lds = C::LDS_BYTES; e.wg = C::WG_THREADS; e.t_per_wg = C::T_PER_WG; e.cfg = name;
    TiledFn<T> fn;
    if (dir == PFFFT_FORWARD) fn = real ? fft_tiled_kernel<C, FWD, 1> : fft_tiled_kernel<C, FWD, 0>;
    else fn = real ? fft_tiled_kernel<C, BWD, 1> : fft_tiled_kernel<C, BWD, 0>;
    e.fn = reinterpret_cast<const void*>(fn);
    return e;
}
#define PF_TSEL(C) tiled_sel<T, C>(dir, real, #C)
#define PF_TSELP(C) tiled_sel<T, typename TiledPick<T>::C>(dir, real, "TiledPick::" #C)

// A measured table.  Constraint: both layouts of a direction share one configuration wherever the configurations differ in their twiddle
// arithmetic, because pffft_transform_ordered == pffft_zreorder(pffft_transform) holds bit for bit (benchmarks/bench_pffft.c:343-349).
template <typename T>
static bool tiled_pick(int n, int dir, int real, int ordered, TiledSel* e) {
    const bool fwd = dir == PFFFT_FORWARD;
    if constexpr (sizeof(T) == 4) {
        // round 3, after the packed-arithmetic change (tools/route_ab.py, 1 GiB per launch, sum of both layouts of a direction):
        //   n = 8192: real forward (C3) three-stage T8192np 0.657 / 0.706; complex forward runs the Stockham plan (0.70 / 0.76 against
        //             0.68 / 0.68), complex backward the four-stage TiledPick (0.77 / 0.71 against 0.70 / 0.70)
        //   n = 4096: forward three-stage (complex 0.72 / 0.73 against 0.69 / 0.71; real N = 8192 0.66 / 0.71 against 0.59 / 0.66), backward
        //             the four-stage TiledPick (complex 0.79 / 0.74 against 0.73 / 0.73; real 0.70 / 0.71 against 0.71 / 0.66)
        //   n = 2048: complex forward three-stage, one wavefront per transform (0.76 / 0.76 against 0.75 / 0.74); complex backward and real
        //             (N = 4096) the four-stage TiledPick (0.81 / 0.76 against 0.75 / 0.76)
        //   n = 16384: 512 threads x 32 points with the register prefetch (tools/c16k_quick.py, 1024-thread configuration -> this one):
        //             complex fwd canonical 0.66 -> 0.69, bwd 0.63 / 0.67 -> 0.68 / 0.70, real N = 32768 bwd 0.52 / 0.59 -> 0.54 / 0.64; real
        //             forward spills into the internal layout (0.55 -> 0.45) and stays on TiledPick
        if (n == 8192 && real && fwd) { *e = PF_TSEL(TiledAltF32b::T8192np); return true; }
        if (n == 2048 && !real && fwd) { *e = PF_TSEL(TiledAltF32b::T2048); return true; }
        if (n == 4096 && fwd) {
            *e = (!real && ordered) ? PF_TSEL(TiledAltF32b::T4096) : PF_TSEL(TiledAltF32b::T4096np);
            return true;
        }
        if (n == 16384 && (!real || !fwd)) { *e = PF_TSEL(TiledAltF32b::T16384); return true; }
    }
    if constexpr (sizeof(T) == 8) {
        // alt: 0 = TiledPick, 1 = A (register base twiddles), 3 = C (A + prefetch); fft_tiled.h TiledAltF64.  tools/c5_ab.py on MI355X;
        // 2048 / 4096: these beat the Stockham kernel that had taken double >= 2048 over (0.65-0.73 -> 0.70-0.80)
        int alt = 0;
        if (n == 1024) alt = (real && fwd) ? 1 : 3;
        else if (n == 512) alt = (real && fwd && !ordered) ? 1 : 3;
        else if (n == 256) alt = !real ? (fwd ? 1 : 3) : (fwd ? 0 : 3);
        else if (n == 128) alt = (real && !fwd) ? 3 : 0;
        else if (n == 2048) alt = (real && fwd && !ordered) ? 1 : 3;
        else if (n == 4096) alt = !real ? ((!fwd && ordered) ? 1 : 3) : ((fwd && !ordered) ? 1 : 3);
#define PF_ALT64(N)                                                                   \
        case N:                                                                       \
            if (alt == 1) { *e = PF_TSEL(TiledAltF64::A##N); return true; }           \
            if (alt == 3) { *e = PF_TSEL(TiledAltF64::C##N); return true; }           \
            break;
        switch (n) { PF_ALT64(128) PF_ALT64(256) PF_ALT64(512) PF_ALT64(1024) PF_ALT64(2048) PF_ALT64(4096) }
#undef PF_ALT64
    }
    switch (n) {
        case 16: *e = PF_TSELP(C16); return true;
        case 32: *e = PF_TSELP(C32); return true;
        case 64: *e = PF_TSELP(C64); return true;
        case 128: *e = PF_TSELP(C128); return true;
        case 256: *e = PF_TSELP(C256); return true;
        case 512: *e = PF_TSELP(C512); return true;
        case 1024: *e = PF_TSELP(C1024); return true;
        case 2048: *e = PF_TSELP(C2048); return true;
        case 4096: *e = PF_TSELP(C4096); return true;
        case 8192: *e = PF_TSELP(C8192); return true;
        case 16384: *e = PF_TSELP(C16384); return true;
    }
    return false;
}
#undef PF_TSEL
#undef PF_TSELP

template <typename T>
static int launch_tiled(Setup* s, const Route& r, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    const TiledSel& e = r.tiled;
    TiledFn<T> fn = reinterpret_cast<TiledFn<T>>(const_cast<void*>(e.fn));
    int rc = allow_big_lds(fn, e.lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(e.fn, e.wg, e.lds, &per_cu))) return rc;
    size_t groups = (batch + e.t_per_wg - 1) / e.t_per_wg;
    size_t grid = (size_t)num_cus() * per_cu;
    // LR_INORDER: launches of up to r.oneshot groups per resident workgroup run as ONE group per workgroup in hardware dispatch order instead of
    // the persistent in-order loop, which pays for itself only over a long run of groups (tools/r4_small_batch.py, us per call, loop ->
    // dispatch order: C3's kernel at 64 / 128 MiB of vectors 46 / 72 -> 30 / 60, N = 4096 complex 44 / 69 -> 24 / 49, N = 256 at 32 MiB 25 -> 14,
    // N = 1024 double at 64 MiB 34 -> 27; from 8 groups per workgroup on the loop wins: N = 1024 double at 256 MiB 98 against 110)
    if (r.oneshot > 0 && groups <= (size_t)r.oneshot * grid && groups < 0x7fffffffull) grid = groups;
    if (grid > groups) grid = groups;
    const int flags = (((dir == PFFFT_BACKWARD) && !ordered) ? 1 : 0) | (((dir == PFFFT_FORWARD) && !ordered) ? 2 : 0);
    unsigned* ctr = groups <= grid ? nullptr : take_counters(s, st);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(e.wg), e.lds, st, in, out, (unsigned)batch, flags,
                       (const cx<T>*)s->d_tw, (const cx<T>*)s->d_twr, ctr);
    PF_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// mixed-radix Stockham plans (fft_stock.h): kernel twin, organisation and launch rule of a (direction, layout)
// ------------------------------------------------------------------------------------------------
// false: the product build has no kernel for this plan (a development build runs the run-time-plan kernel: r.stock.fn == nullptr)
template <typename T>
static bool plan_stock(const Setup* s, int dir, int ordered, const AbSel& sel, Route& r) {
    const bool bwd = dir == PFFFT_BACKWARD, real = s->transform == PFFFT_REAL;
    StockSel& k = r.stock;
    k.wl = s->skw_ok && !(PF_HAS_VARIANTS && sel.is(AB_STOCK_WORKGROUP));   // (workgroup plans of the small sizes have run-time twins only)
    if (!k.wl && !s->sk_ok) return false;
    const StockPlan& sp = k.wl ? s->skw[bwd ? 1 : 0] : s->sk[bwd ? 1 : 0];
    k.lds = stock_lds<T>(sp).total;
    k.flags = ((bwd && !ordered) ? 1 : 0) | ((!bwd && !ordered) ? 2 : 0) | (bwd ? 4 : 0) | (real ? 8 : 0);
    // Static stride by default: these kernels sit at ~0.6-0.75 of the roofline on latency, not on the HBM access order, and the hand-over
    // of the next chunk costs more than the ordering buys (0.63 static against 0.54 in-order chunks on N = 96 .. 800).  The workgroup-phase
    // kernels of the LARGE plans - one workgroup per CU - do gain from the order (round 3, 1 GiB per launch, tools/scan_variant.py): complex
    // float n = 4320 .. 5760 +0.01 .. +0.04, n = 8192 forward 0.68 -> 0.77 / 0.80, n = 8640 0.65 -> 0.77; double n = 2160 .. 4800 +0.02 .. +0.09,
    // n = 4096 forward 0.69 -> 0.81 / 0.79; real transforms are neutral up to 64 KiB and gain 0 .. +0.04 beyond.
    const size_t vbytes = (size_t)sp.n * sizeof(cx<T>);
    bool auto_dyn = false;
    if (!k.wl) auto_dyn = real ? vbytes >= 65536 : vbytes >= (sizeof(T) == 8 ? 32u : 34u) * 1024u;
    const bool dyn = sel.is(AB_INORDER_SMALL) || (auto_dyn && !sel.is(AB_STATIC_LARGE));
    r.rule = dyn ? LR_INORDER : LR_STATIC;
    r.oneshot = dyn ? env().oneshot : 0;
    // the kernel body instantiated on this very plan as a compile-time constant; per plan, direction and layout either the deposit
    // variant or the direct first stage (operands straight from HBM into registers, fft_stock.h sk_df_body) is adopted from a measured
    // table (stock_df_gen.h, tools/tune_stock_df.py): the product build instantiates the adopted one only
    const bool rt_forced = PF_HAS_VARIANTS && sel.is(AB_STOCK_RUNTIME);
    const bool df_ok = !(k.flags & 1) && !rt_forced;
    const bool want_df = df_ok && (sel.is(AB_STOCK_DF_ON) ? true : sel.is(AB_STOCK_DF_OFF) ? false
                                   : stock_df_adopted(sizeof(T) == 8, real, sp.n, (k.flags & 2) != 0, bwd));
    auto cf = want_df ? stock_ct_lookup(sp, k.flags | 16, k.wl, (const T*)nullptr) : nullptr;
    k.df = cf != nullptr;
    if (!cf && !rt_forced) cf = stock_ct_lookup(sp, k.flags, k.wl, (const T*)nullptr);
    if (!cf && !rt_forced && df_ok) { cf = stock_ct_lookup(sp, k.flags | 16, k.wl, (const T*)nullptr); k.df = cf != nullptr; }
    k.fn = reinterpret_cast<const void*>(cf);
    k.threads = (cf && k.df) ? sk_df_threads(sp, k.flags & 15, k.wl) : (k.wl ? s->skw_threads : s->sk_threads);
    // grid of the static stride: groups per workgroup from the measured table (stock_grid_gen.h, tools/tune_stock_grid.py: 2-3 for the
    // complex plans, 3-4 for the real ones, N = 384 / 768 complex float 0.71-0.75 -> 0.78-0.80) or the size rule - 16 x the resident set
    // for vectors <= 4 KiB, 8 x up to 20 KiB, the resident set beyond (tools/stock_bench2.py: N = 96 .. 480 0.63-0.69 -> 0.71-0.77)
    k.groups_per_wg = stock_grid_its(sizeof(T) == 8, real, sp.n);
    k.grid_mul = vbytes <= 4096 ? 16 : vbytes <= 20480 ? 8 : 1;
    return cf != nullptr || PF_HAS_VARIANTS;
}

template <typename T>
static int launch_stock(Setup* s, const Route& r, const T* in, T* out, size_t batch, int dir, hipStream_t st) {
    const StockSel& k = r.stock;
    const bool bwd = dir == PFFFT_BACKWARD;
    const StockPlan& sp = k.wl ? s->skw[bwd ? 1 : 0] : s->sk[bwd ? 1 : 0];
    const cx<T>* twp = (const cx<T>*)(sp.twmode == 2 ? s->d_twc[bwd ? 1 : 0] : s->d_tw);
    const size_t groups = (batch + sp.G - 1) / sp.G;
    const bool dyn = r.rule == LR_INORDER;
    // in-order groups are pulled in chunks of ONE group from 44 KiB per group on (double n = 3072 .. 4000 0.71-0.75 -> 0.78-0.82 against two),
    // below that of as many as keep one counter address under its ~80 M atomics/s; never so large that a workgroup sees fewer than ~8 chunks
    const size_t gbytes = (size_t)sp.G * sp.n * sizeof(cx<T>);
    auto chunk_for = [&](size_t grid) -> unsigned {
        size_t kk = (44000 + gbytes - 1) / gbytes, cap = groups / (8 * grid);
        if (kk > cap) kk = cap;
        return (unsigned)(kk < 1 ? 1 : (kk > 64 ? 64 : kk));
    };
    if (k.fn) {
        StockCtFn<T> cf = reinterpret_cast<StockCtFn<T>>(const_cast<void*>(k.fn));
        int rc = allow_big_lds(cf, k.lds);
        if (rc) return rc;
        int per_cu = 0;
        if ((rc = cached_occupancy(k.fn, k.threads, k.lds, &per_cu))) return rc;
        size_t grid = (size_t)num_cus() * per_cu;
        if (k.groups_per_wg > 0) {
            const size_t want = (groups + (size_t)k.groups_per_wg - 1) / (size_t)k.groups_per_wg;
            if (want > grid) grid = want;                       // (never below the resident set)
        } else {
            grid *= (size_t)k.grid_mul;
        }
#ifdef PFFFT_HIP_VARIANTS
        {   // the tuner's knob (tools/tune_stock_grid.py): selectors 210 + i force SK_ITS[i] groups per workgroup, 208 the size rule alone
            static const int SK_ITS[12] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
            const int v = ab().raw;
            if (v >= 210 && v <= 221) { grid = (size_t)num_cus() * per_cu; const size_t want = (groups + SK_ITS[v - 210] - 1) / SK_ITS[v - 210]; if (want > grid) grid = want; }
            if (v == 208) grid = (size_t)num_cus() * per_cu * (size_t)k.grid_mul;
        }
#endif
        if (dyn && r.oneshot > 0 && groups <= (size_t)r.oneshot * grid) grid = groups;   // (n = 8192 complex float at 32 MiB of vectors 25 -> 16 us)
        if (grid > groups) grid = groups;
        if (grid > 0x7fffffffu) grid = 0x7fffffffu;
        unsigned* ctr = (groups <= grid || !dyn) ? nullptr : take_counters(s, st);
        hipLaunchKernelGGL(cf, dim3((unsigned)grid), dim3(k.threads), k.lds, st, in, out, batch, twp, (const cx<T>*)s->d_twr, ctr, chunk_for(grid));
        PF_CHECK(hipGetLastError());
        return 0;
    }
#ifdef PFFFT_HIP_VARIANTS
    // the same bodies on the run-time plan (0.3 of the roofline: issue-bound).  Every legal size has a compile-time plan
    // (tests/test_generated_sources.py), so the product build does not carry these kernels (2 MB); AB_STOCK_RUNTIME forces them here.
    auto kf = k.wl ? fft_stock_wl_kernel<T> : fft_stock_kernel<T>;
    int rc = allow_big_lds(kf, k.lds);
    if (rc) return rc;
    int per_cu = 0;
    if ((rc = cached_occupancy(reinterpret_cast<const void*>(kf), k.threads, k.lds, &per_cu))) return rc;
    size_t grid = (size_t)num_cus() * per_cu;
    if (grid > groups) grid = groups;
    unsigned* ctr = (groups <= grid || !dyn) ? nullptr : take_counters(s, st);
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(k.threads), k.lds, st, in, out, batch, sp, k.flags, twp, (const cx<T>*)s->d_twr, ctr, chunk_for(grid));
    PF_CHECK(hipGetLastError());
    return 0;
#else
    (void)twp; (void)chunk_for;
    g_last_error = "pffft_hip: no compile-time Stockham plan for this size (product build)";
    return (int)hipErrorInvalidValue;
#endif
}

template <typename T> static int zreorder_batch(Setup* s, const T* in, T* out, size_t batch, int dir, hipStream_t st);

template <typename T>
static int launch_strided(Setup* s, int which, const cx<T>* in, cx<T>* out, size_t batch, int dir, hipStream_t st) {
    const StridedPlan& sp = s->bigp[which];
    const size_t lds = ((size_t)sp.G * sp.n + ((size_t)sp.G * sp.n >> 5) + 2) * sizeof(cx<T>);  // padded image (gpad)
    long long groups = (long long)batch * ((sp.count + sp.G - 1) / sp.G);
    long long grid = (long long)num_cus() * 4;
    if (grid > groups) grid = groups;
    int th = (int)(((size_t)sp.G * sp.n / 8 + 63) / 64 * 64);
    th = th < 64 ? 64 : (th > 1024 ? 1024 : th);
    auto kf = fft_strided_kernel<T, FWD>;
    auto kb = fft_strided_kernel<T, BWD>;
    int rc = allow_big_lds(dir == PFFFT_FORWARD ? kf : kb, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dir == PFFFT_FORWARD ? kf : kb, dim3((unsigned)grid), dim3(th), lds, st, in, out, (long long)batch, sp,
                       (const cx<T>*)s->d_bigtw[which]);
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T> static int transform_batch(Setup* s, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st);

// n = R x N2: columns in registers -> batched LDS-resident rows -> tiled transpose (fft_big.h)
template <typename T>
static int big_small_factor(Setup* s, const cx<T>* in, cx<T>* work, cx<T>* out, size_t batch, int dir, hipStream_t st, bool out_int = false,
                            bool in_int = false) {
    // in_int (backward only, R a multiple of 4): the column pass reads the internal layout itself (big_col_int_kernel)
    // out_int (forward only): the transpose stores the internal layout itself (big_transpose_int_kernel)
    const int R = s->bigR, N2 = s->sub->n;
    const unsigned tgrid_int = (unsigned)(batch * (size_t)((N2 / 4 + 63) / 64));
    const long long total = (long long)batch * N2;
    const unsigned grid = (unsigned)((total + 255) / 256);
    const double inv_n = 1.0 / (double)s->n;
    const unsigned tgrid = (unsigned)(batch * (size_t)((N2 + 255) / 256));
#define PF_BIG_R(RR)                                                                                               \
    case RR: {                                                                                                     \
        if constexpr (RR % 4 == 0) {                                                                               \
            if (in_int) {                                                                                          \
                auto kc = big_col_int_kernel<T, RR, BWD>;                                                          \
                const size_t ldsc = (size_t)RR * 257 * sizeof(cx<T>);                                              \
                int rcc = allow_big_lds(kc, ldsc);                                                                 \
                if (rcc) return rcc;                                                                               \
                hipLaunchKernelGGL(kc, dim3(tgrid), dim3(256), ldsc, st, (const T*)in, work, (long long)batch, N2, inv_n); \
            }                                                                                                      \
        }                                                                                                          \
        if (in_int && RR % 4 == 0) {                                                                               \
        } else if (dir == PFFFT_FORWARD) hipLaunchKernelGGL((big_col_kernel<T, RR, FWD>), dim3(grid), dim3(256), 0, st, in, work, total, N2, inv_n); \
        else hipLaunchKernelGGL((big_col_kernel<T, RR, BWD>), dim3(grid), dim3(256), 0, st, in, work, total, N2, inv_n);  \
        PF_CHECK(hipGetLastError());                                                                               \
        int rc = transform_batch<T>(s->sub, (const T*)work, (T*)work, batch * (size_t)RR, dir, 1, st);             \
        if (rc) return rc;                                                                                         \
        const size_t lds = (size_t)256 * (RR + 1) * sizeof(cx<T>);                                                 \
        if (out_int) {                                                                                             \
            auto ki = big_transpose_int_kernel<T, RR>;                                                             \
            if ((rc = allow_big_lds(ki, lds))) return rc;                                                          \
            hipLaunchKernelGGL(ki, dim3(tgrid_int), dim3(256), lds, st, (const cx<T>*)work, (T*)out, (long long)batch, N2); \
        } else {                                                                                                   \
            auto k = big_transpose_kernel<T, RR>;                                                                  \
            if ((rc = allow_big_lds(k, lds))) return rc;                                                           \
            hipLaunchKernelGGL(k, dim3(tgrid), dim3(256), lds, st, (const cx<T>*)work, out, (long long)batch, N2); \
        }                                                                                                          \
        PF_CHECK(hipGetLastError());                                                                               \
        return 0;                                                                                                  \
    }
    switch (R) {
        PF_BIG_R(2) PF_BIG_R(3) PF_BIG_R(4) PF_BIG_R(5) PF_BIG_R(6) PF_BIG_R(8) PF_BIG_R(9) PF_BIG_R(10) PF_BIG_R(12) PF_BIG_R(15) PF_BIG_R(16)
        PF_BIG_R(25) PF_BIG_R(27) PF_BIG_R(32)
    }
#undef PF_BIG_R
    g_last_error = "pffft_hip: unsupported small factor";
    return (int)hipErrorInvalidValue;
}

// one-sweep layout / pair kernels of the beyond-LDS path (fft_big.h big_block_kernel): mode 0 complex canonical -> internal,
// 1 complex internal -> canonical, 2 real forward Z -> X (internal), 3 real backward X (internal) -> Z', 4 real backward
// X (canonical) -> Z', 5 real X (canonical) -> X (internal), a pure permutation.  in != out.
template <typename T>
static int launch_block(Setup* s, int mode, const T* in, T* out, size_t batch, hipStream_t st) {
    const long long n = s->n, tiles = (long long)batch * ((n / 4 + 63) / 64);
    long long grid = (tiles + BLK_WAVES - 1) / BLK_WAVES;
    // ONE tile per wavefront in hardware dispatch order (grid = every tile): real N = 2^18 forward unordered 0.210 -> 0.224 of the
    // roofline for the whole transform, backward 0.200-0.204 -> 0.213-0.214, against persistent wavefronts on a static stride
    // or chunks of 4 .. 32 consecutive tiles per wavefront (no better) - the order of the accesses again
    const int kchunk = 1;
    if (grid > 0x7fffffffll) grid = 0x7fffffffll;
    const dim3 g((unsigned)grid), b(BLK_WAVES * 64);
    switch (mode) {
        case 0: hipLaunchKernelGGL((big_block_kernel<T, 0>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
        case 1: hipLaunchKernelGGL((big_block_kernel<T, 1>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
        case 2: hipLaunchKernelGGL((big_block_kernel<T, 2>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
        case 3: hipLaunchKernelGGL((big_block_kernel<T, 3>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
        case 4: hipLaunchKernelGGL((big_block_kernel<T, 4>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
        default: hipLaunchKernelGGL((big_block_kernel<T, 5>), g, b, 0, st, in, out, (long long)batch, n, kchunk); break;
    }
    PF_CHECK(hipGetLastError());
    return 0;
}

// Per-stream scratch of a setup (held under the setup's scratch lock).  More than SCRATCH_STREAMS streams: ONE entry goes - the stream that
// used this setup longest ago (hipFree waits for its kernels) - not the whole map: a caller cycling through nine streams would otherwise
// free and re-allocate every stream's buffers on every call.  An entry a HIP graph has recorded (Scratch::captured) is never the victim,
// and a buffer it outgrows is retired instead of freed: a replay dereferences the pointers it froze at capture time.
constexpr size_t SCRATCH_STREAMS = 8;
static int stream_scratch(std::map<hipStream_t, Setup::Scratch>& tab, unsigned long long& clock, hipStream_t st, Setup::Scratch** out) {
    if (tab.size() >= SCRATCH_STREAMS && !tab.count(st)) {
        auto victim = tab.end();
        for (auto it = tab.begin(); it != tab.end(); ++it)
            if (!it->second.captured && (victim == tab.end() || it->second.last_use < victim->second.last_use)) victim = it;
        if (victim != tab.end()) {
            for (void* p : victim->second.buf) if (p) (void)hipFree(p);
            tab.erase(victim);
        }
    }
    Setup::Scratch& sc = tab[st];
    sc.last_use = ++clock;
    if (stream_capturing(st)) sc.captured = true;
    *out = &sc;
    return 0;
}
static int scratch_grow(Setup* s, Setup::Scratch& sc, int i, size_t bytes) {
    if (sc.bytes[i] >= bytes) return 0;
    // (hipFree waits for the device: kernels of this stream still using the old buffer finish first)
    if (sc.buf[i]) {
        if (sc.captured) { std::lock_guard<std::mutex> lk(s->retired_mu); s->retired.push_back(sc.buf[i]); }
        else (void)hipFree(sc.buf[i]);
    }
    sc.buf[i] = nullptr; sc.bytes[i] = 0;
    PF_CHECK(hipMalloc(&sc.buf[i], bytes));   // (while the stream is capturing this fails: warm the setup up with the largest batch first)
    sc.bytes[i] = bytes;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// n beyond LDS: the sweeps over HBM of a (direction, layout), planned once (BigPlan), executed by launch_big
// ------------------------------------------------------------------------------------------------
static void plan_big(const Setup* s, int dir, int ordered, const AbSel& sel, BigPlan& b) {
    const bool real = s->transform == PFFFT_REAL, fwd = dir == PFFFT_FORWARD, dbl = s->is_double != 0;
    const bool strided = sel.is(AB_BIG_STRIDED), no_tiles = strided || sel.is(AB_BIG_NO_TILES);
    b = BigPlan();
    // real forward into the canonical spectrum in TWO sweeps where they measured faster (tile_real_tu.hip RMODE; AB_RFFT_THREE = always the
    // complex core + pair sweep, AB_RFFT_TWO = two sweeps wherever the length splits).  Ordered and unordered take the SAME route -
    // pffft_transform_ordered == pffft_zreorder(pffft_transform) bit for bit -: the unordered spectrum is the canonical one of the two
    // sweeps through the one-sweep permutation big_block_kernel<5>
    if (real && fwd && !sel.is(AB_RFFT_THREE) && !no_tiles && tile_rfft_has_plan(2LL * s->n, dbl, !sel.is(AB_RFFT_TWO))) {
        b.core = BIG_RFFT2;
        b.post = ordered ? -1 : 5;
        b.sweeps = ordered ? 2 : 3;
        return;
    }
    const bool blk = !sel.is(AB_BIG_SEPARATE_SWEEPS), fuse_ok = !sel.is(AB_BIG_SEPARATE_LAYOUT);
    // two (three beyond 2^20) tile passes: power-of-two n and the n whose odd part splits over two tile lengths (tile_tu.hip)
    // (deep: the row length of the streaming route is itself beyond LDS, or there is no streaming plan - five sweeps)
    const bool deep = !s->bigR || !s->sub || s->sub->kernel == K_BIG;
    b.tmode = deep ? 1 : real ? 2 : 0;
    bool tiled = !no_tiles && tile_has_plan(s->n, dbl, b.tmode);
    const int tlay = tiled ? tile_plan_layouts(s->n, dbl, b.tmode) : 0;
    // a complex plan with a run-time tile pass that cannot carry the internal layout costs two passes + a reorder sweep against the three
    // streaming passes, which fuse it (measured 0.13-0.15 against 0.18-0.24) - and ordered / unordered must run the SAME arithmetic:
    // such a plan is used for both layouts or for none; it stays where the streaming route would take five sweeps (deep)
    if (tiled && !real && !deep && s->bigR && tlay != 3) tiled = false;
    // backward from the internal layout: the first tile pass / the column pass of the streaming route (R a multiple of 4) reads it itself
    b.fuse_in = !fwd && !ordered && !real && tiled && (tlay & 2) && fuse_ok;
    b.col_in = !fwd && !ordered && !real && !tiled && s->bigR && s->bigR % 4 == 0 && !strided && fuse_ok;
    if (!b.fuse_in && !b.col_in) {
        if (!fwd && !ordered) b.pre = real ? 3 : 1;      // internal -> canonical (complex) / -> packed spectrum of the inverse (real)
        else if (!fwd && real) b.pre = 4;                // canonical half-complex spectrum -> packed spectrum
        b.pre_separate = b.pre >= 0 && !blk;
    }
    // real forward ORDERED on a two-pass plan whose row pass is a register-tiled one on 128-byte runs: the pair pass runs INSIDE that pass (mirror-
    // closed row tiles, fft_tile.h RMODE 3) - two sweeps into the canonical spectrum instead of three.  Its arithmetic is the pair sweeps' operation
    // for operation, so the unordered transform keeps its three sweeps (pair pass + internal layout in one) and pffft_transform_ordered ==
    // pffft_zreorder(pffft_transform) still holds bit for bit.  Adopted in double (0.24-0.25 -> 0.28-0.32 of the roofline); in float the exact-
    // argument W_N^k per pair (a double-precision division + sincospif) costs the small tiles more than the sweep saves - 128-point rows 0.25 ->
    // 0.23, 512-point rows 0.25 -> 0.26 - and AB_RFFT_TWO runs it.  AB_RFFT_THREE: the complex core + pair sweep (the second route of the tests)
    if (tiled && real && fwd && ordered && blk && !sel.is(AB_RFFT_THREE) && (dbl || sel.is(AB_RFFT_TWO)) && tile_real_rows_plan(s->n, dbl, b.tmode)) {
        b.core = BIG_TILES;
        b.rfuse = true;
        b.sweeps = tile_plan_lengths(s->n, dbl, b.tmode, b.lens);
        return;
    }
    if (tiled) {
        b.core = BIG_TILES;
        b.fuse_out = fwd && !ordered && !real && (tlay & 1) && fuse_ok;      // the last tile pass stores the internal layout
        b.sweeps = tile_plan_lengths(s->n, dbl, b.tmode, b.lens);
    } else if (s->bigR && !strided) {
        b.core = BIG_STREAM;
        b.fuse_out = fwd && !ordered && !real && fuse_ok;                    // the transpose pass stores it
        b.lens[0] = s->bigR; b.lens[1] = s->sub->n;
        b.sweeps = s->sub->kernel == K_BIG ? 5 : 3;
    } else {
        b.core = BIG_STRIDED;
        b.lens[0] = s->bigp[0].n; b.lens[1] = s->bigp[1].n;
        b.sweeps = 2;
    }
    if (fwd && !ordered && !b.fuse_out) { b.post = real ? 2 : 0; b.post_separate = !blk; }   // (real: pair pass +) canonical -> internal
    else if (fwd && real) b.pair_after = true;                                                 // real forward ordered: in-place pair pass
    b.sweeps += (b.pre >= 0 ? (b.pre_separate && b.pre != 1 ? 2 : 1) : 0) + (b.post >= 0 ? (b.post_separate && b.post == 2 ? 2 : 1) : 0) + (b.pair_after ? 1 : 0);
}

template <typename T>
static int launch_big(Setup* s, const Route& r, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    const BigPlan& b = r.big;
    size_t bytes = batch * (size_t)s->n * sizeof(cx<T>);
    // (the work rows of the real two-sweep route - k1 <= N1/2, whole row tiles - need a little more than n)
    if (b.core == BIG_RFFT2) bytes = std::max(bytes, batch * tile_rfft_work_elems(2LL * s->n, s->is_double != 0) * sizeof(cx<T>));
    cx<T>*bufA, *bufB;
    // big_mu is held until EVERY pass of this call is enqueued: it guards host-side enqueue only, and a second thread growing the same
    // stream's scratch (hipFree synchronises with the device) can then never free buffers whose kernels are not yet in the stream
    std::lock_guard<std::mutex> lk(s->big_mu);
    {
        Setup::Scratch* scp = nullptr;
        int rcs = stream_scratch(s->big_scratch, s->scratch_clock, st, &scp);
        if (rcs) return rcs;
        for (int i = 0; i < 2; ++i)
            if ((rcs = scratch_grow(s, *scp, i, bytes))) return rcs;
        bufA = (cx<T>*)scp->buf[0];
        bufB = (cx<T>*)scp->buf[1];
    }
    const bool real = s->transform == PFFFT_REAL, fwd = dir == PFFFT_FORWARD;
    int rc;
    if (b.core == BIG_RFFT2) {
        rc = launch_tile_rfft(s, in, bufB, ordered ? (void*)out : (void*)bufA, batch, 2LL * s->n, dir, st);
        if (rc < 0) { g_last_error = "pffft_hip: the planned two-sweep real route has no kernel"; return (int)hipErrorInvalidValue; }
        if (rc) return rc;
        return b.post == 5 ? launch_block<T>(s, 5, (const T*)bufA, out, batch, st) : 0;
    }
    // in-place pair pass: one pair per thread, every workgroup once, in dispatch order
    const size_t pair_wgs = (batch * ((size_t)s->n / 2 + 1) + 255) / 256;
    const unsigned egrid = (unsigned)std::min<size_t>(pair_wgs, (size_t)0x7fffffff);
    // ---- before the core
    const cx<T>* cur = (const cx<T>*)in;
    if (b.pre >= 0 && !b.pre_separate) {
        if ((rc = launch_block<T>(s, b.pre, in, (T*)bufA, batch, st))) return rc;
        cur = bufA;
    } else if (b.pre >= 0) {
        if (b.pre != 4) {     // internal -> canonical
            if ((rc = zreorder_batch<T>(s, in, (T*)bufA, batch, PFFFT_FORWARD, st))) return rc;
            cur = bufA;
        }
        if (real) {           // half-complex spectrum -> packed spectrum (in place, never on the caller's input)
            if (cur != bufA) { PF_CHECK(hipMemcpyAsync(bufA, cur, bytes, hipMemcpyDeviceToDevice, st)); cur = bufA; }
            hipLaunchKernelGGL((real_pair_kernel<T, BWD>), dim3(egrid), dim3(256), 0, st, bufA, (long long)batch, (long long)s->n);
            PF_CHECK(hipGetLastError());
        }
    }
    // ---- the core: canonical complex transform cur -> dest (or straight into `out` in the internal layout)
    cx<T>* dest = (b.post >= 0) ? bufA : (cx<T>*)out;
    switch (b.core) {
        case BIG_TILES:
            rc = launch_tile_fft(s, cur, bufB, dest, batch, (long long)s->n, dir, st, b.rfuse ? 3 : b.fuse_out ? 1 : b.fuse_in ? 2 : 0, b.tmode);
            if (rc < 0) { g_last_error = "pffft_hip: the planned tile passes have no kernel"; return (int)hipErrorInvalidValue; }
            if (rc) return rc;
            break;
        case BIG_STREAM:
            if ((rc = big_small_factor<T>(s, cur, bufB, dest, batch, dir, st, b.fuse_out, b.col_in))) return rc;
            break;
        default:
            if ((rc = launch_strided<T>(s, 0, cur, bufB, batch, dir, st))) return rc;
            if ((rc = launch_strided<T>(s, 1, bufB, dest, batch, dir, st))) return rc;
            break;
    }
    // ---- after the core
    if (b.post >= 0 && !b.post_separate) return launch_block<T>(s, b.post, (const T*)bufA, out, batch, st);
    if (b.pair_after || (b.post == 2 && b.post_separate)) {
        hipLaunchKernelGGL((real_pair_kernel<T, FWD>), dim3(egrid), dim3(256), 0, st, dest, (long long)batch, (long long)s->n);
        PF_CHECK(hipGetLastError());
    }
    if (b.post >= 0) return zreorder_batch<T>(s, (const T*)bufA, out, batch, PFFFT_BACKWARD, st);   // canonical -> internal
    (void)fwd;
    return 0;
}

// n = 16 / 32: one thread per transform (fft_tiny.h)
template <typename T, int n>
static int launch_tiny(Setup* s, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    constexpr int CPV = 2 * n * (int)sizeof(T) / 16;
    const int waves = CPV >= 32 ? 2 : 4;
    const size_t lds = (size_t)waves * 64 * (CPV + 1) * 16;
    const size_t groups = (batch + 63) / 64;
    size_t grid = (groups + waves - 1) / waves;
    const size_t cap = (size_t)num_cus() * (LDS_MAX / lds > 8 ? 8 : LDS_MAX / lds);
    // (one group of 64 vectors per wavefront in hardware dispatch order: plan_route has the measurement)
    (void)cap;
    const bool real = s->transform == PFFFT_REAL, fwd = dir == PFFFT_FORWARD;
    const cx<T>* twr = (const cx<T>*)s->d_twr;
#define PF_TINY(D, R, I, O)                                                                                        \
    do {                                                                                                           \
        auto k = fft_tiny_kernel<T, n, D, R, I, O>;                                                                \
        int rc = allow_big_lds(k, lds);                                                                            \
        if (rc) return rc;                                                                                         \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves * 64), lds, st, in, out, batch, twr);               \
    } while (0)
    if (!real) {
        if (fwd) { if (ordered) PF_TINY(FWD, 0, 0, 0); else PF_TINY(FWD, 0, 0, 1); }
        else { if (ordered) PF_TINY(BWD, 0, 0, 0); else PF_TINY(BWD, 0, 1, 0); }
    } else {
        if (fwd) { if (ordered) PF_TINY(FWD, 1, 0, 0); else PF_TINY(FWD, 1, 0, 1); }
        else { if (ordered) PF_TINY(BWD, 1, 0, 0); else PF_TINY(BWD, 1, 1, 0); }
    }
#undef PF_TINY
    PF_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the planner: one Route per (direction, layout), computed at pffft_new_setup (plan_routes) - and again, into a temporary, only for a
// call made under a non-default A/B selector
// ------------------------------------------------------------------------------------------------
// power-of-two sizes where the Stockham kernel on its compile-time plan measured faster than the register-tiled one (1 GiB of vectors,
// tools/stock_ab.py, tools/route_ab.py):
//   float   complex n = 16: 0.46 vs 0.26, 32: 0.70 vs 0.59, 64: 0.70 vs 0.65, 128: 0.65 -> 0.71-0.74, 8192: 0.68-0.73 vs 0.62-0.70 (both directions since
//           the large plans are pulled in order: backward 0.78 / 0.72 -> 0.77 / 0.78); real N = 32: 0.38 vs 0.26, 64: 0.66 vs 0.60, N = 128 0.66-0.69 ->
//           0.62-0.75, N = 16384 backward 0.61-0.68 vs 0.51-0.60 (symmetric spectrum-side stage)
//   double  complex n = 16: 0.50 vs 0.26, 32: 0.70 vs 0.48, 64: 0.69 vs 0.47, 128 / 256 0.68 -> 0.75, 2048 0.70-0.72 -> 0.75-0.77, 4096 forward
//           0.70 -> 0.73 / 0.75, backward 0.71 -> 0.77; real N = 32: 0.42 vs 0.29, 64: 0.66 vs 0.55, 128 0.39-0.50 -> 0.66-0.76, N = 256 backward 0.70 ->
//           0.76, N = 512 0.70-0.73 -> 0.73-0.75, N = 4096 0.69-0.72 -> 0.71-0.75, N = 8192 forward 0.68 / 0.69 -> 0.73 / 0.70
template <typename T>
static bool pow2_prefers_stock(int n, bool cplx, bool fwd) {
    if (sizeof(T) == 4) return cplx ? (n <= 64 || n == 128 || n == 8192) : (n <= 64 || (n == 8192 && !fwd));
    return cplx ? (n <= 256 || n >= 2048) : (n <= 64 || (n == 128 && !fwd) || n == 256 || n >= 2048);
}

template <typename T>
static Route plan_route(const Setup* s, int dir, int ordered, const AbSel& sel) {
    Route r;
    const bool real = s->transform == PFFFT_REAL, fwd = dir == PFFFT_FORWARD;
    // n = 16 / 32: one thread per transform (fft_tiny.h), one group of 64 vectors per wavefront in dispatch order (measured faster than
    // persistent wavefronts on a static stride: 256-byte vectors 0.75-0.78 against 0.67-0.72)
    if (!sel.is(AB_NO_TINY) && (s->n == 16 || (sizeof(T) == 4 && s->n == 32))) { r.fam = FAM_TINY; r.rule = LR_DISPATCH; return r; }
    const bool stock_all = sel.is(AB_STOCK_FOR_TILED) && s->sk_ok;
    if (sizeof(T) == 4 && s->kernel == K_C1024_F32 && !stock_all) {
        r.fam = FAM_C1024; r.rule = LR_INORDER; r.oneshot = env().c1024_rounds;
        return r;
    }
    if (s->kernel == K_TILED && !stock_all) {
        const bool stock = s->sk_ok && pow2_prefers_stock<T>(s->n, !real, fwd);
        if (!stock && tiled_pick<T>(s->n, dir, real ? 1 : 0, ordered, &r.tiled)) {
            r.fam = FAM_TILED; r.rule = LR_INORDER;
            // (N = 4096 complex float alone prefers the dispatch order up to SIXTEEN groups per workgroup: 256 / 512 MiB 111 / 193 -> 91 / 178 us; every
            //  other size measured loses there - 4096 real 104 -> 154 us at 256 MiB, 16384 complex 217 -> 283 at 512 MiB)
            r.oneshot = (env().oneshot == 4 && sizeof(T) == 4 && s->n == 4096 && !real) ? 16 : env().oneshot;
            return r;
        }
    }
    if (s->kernel == K_BIG && s->one_ok && !sel.is(AB_NO_ONE_IMAGE)) { r.fam = FAM_ONE; r.rule = LR_INORDER; return r; }
    if (s->kernel == K_BIG) { r.fam = FAM_BIG; r.rule = LR_INORDER; plan_big(s, dir, ordered, sel, r.big); return r; }
    if ((s->sk_ok || s->skw_ok) && plan_stock<T>(s, dir, ordered, sel, r)) { r.fam = FAM_STOCK; return r; }
    r.fam = FAM_NONE;   // (every legal size is routed above: new_setup sends whatever has no Stockham plan to the streaming passes, K_BIG)
    return r;
}

static void plan_routes(Setup* s) {
    // The stored routes are the DEFAULT ones whatever selector the creating thread has set: the tile planner's helpers
    // (tile_has_plan, tile_plan_lengths) read the calling thread's selector themselves (AB_BIG_NO_MR_TILES), so it is cleared for the
    // planning and restored (a setup created under set_variant(83) had the streaming route baked in as its default)
    const int keep = g_ab_raw;
    g_ab_raw = 0;
    const AbSel none;
    for (int d = 0; d < 2; ++d)
        for (int o = 0; o < 2; ++o)
            s->route[d][o] = s->is_double ? plan_route<double>(s, d, o, none) : plan_route<float>(s, d, o, none);
    g_ab_raw = keep;
}

template <typename T>
static int transform_batch(Setup* s, const T* in, T* out, size_t batch, int dir, int ordered, hipStream_t st) {
    if (!s || s->magic != MAGIC || s->is_double != (sizeof(T) == 8)) {
        g_last_error = "pffft_hip: bad setup handle";
        return (int)hipErrorInvalidHandle;
    }
    if ((dir != PFFFT_FORWARD && dir != PFFFT_BACKWARD)) { g_last_error = "pffft_hip: bad direction"; return (int)hipErrorInvalidValue; }
    if (batch == 0) return 0;
    s = for_device(s);        // the object that holds this setup's tables on the calling thread's device
    int rc = ensure_device<T>(s);
    if (rc) return rc;
    const AbSel sel = ab();
    Route tmp;
    const Route* r = &s->route[dir][ordered ? 1 : 0];
    if (sel.any()) { tmp = plan_route<T>(s, dir, ordered ? 1 : 0, sel); r = &tmp; }
    switch (r->fam) {
        case FAM_TINY:
            if (s->n == 16) return launch_tiny<T, 16>(s, in, out, batch, dir, ordered, st);
            if constexpr (sizeof(T) == 4) return launch_tiny<T, 32>(s, in, out, batch, dir, ordered, st);
            break;
        case FAM_C1024:
        case FAM_TILED: {
            // (these kernels count vectors in 32 bits: longer batches go out in slices on the same stream)
            constexpr size_t SLICE = (size_t)3 << 30;
            for (size_t b0 = 0; b0 < batch; b0 += SLICE) {
                const size_t nb = batch - b0 < SLICE ? batch - b0 : SLICE;
                const T* pi = in + b0 * s->vec_scalars;
                T* po = out + b0 * s->vec_scalars;
                if constexpr (sizeof(T) == 4) {
                    if (r->fam == FAM_C1024) { if ((rc = launch_c1024(s, *r, pi, po, nb, dir, ordered, st))) return rc; continue; }
                }
                if ((rc = launch_tiled<T>(s, *r, pi, po, nb, dir, ordered, st))) return rc;
            }
            return 0;
        }
        case FAM_STOCK: return launch_stock<T>(s, *r, in, out, batch, dir, st);
        case FAM_BIG: return launch_big<T>(s, *r, in, out, batch, dir, ordered, st);
        case FAM_ONE: return launch_one(s, in, out, batch, dir, ordered ? 1 : 0, st);
        default: break;
    }
    g_last_error = "pffft_hip: no kernel for this size";
    return (int)hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// pffft_hip_describe: the routes of a setup as text
// ------------------------------------------------------------------------------------------------
const char* family_name(Family f) {
    switch (f) {
        case FAM_TINY: return "tiny";
        case FAM_C1024: return "c1024_f32";
        case FAM_TILED: return "tiled";
        case FAM_STOCK: return "stockham";
        case FAM_BIG: return "fourstep";
        case FAM_ONE: return "oneimage";
        default: return "none";
    }
}
const char* launch_rule_name(LaunchRule r) {
    switch (r) {
        case LR_DISPATCH: return "dispatch-order";
        case LR_STATIC: return "static-stride";
        default: return "in-order";
    }
}
static int describe_route(const Setup* s, const Route& r, char* buf, size_t len) {
    switch (r.fam) {
        case FAM_TINY: return snprintf(buf, len, "tiny: one thread per transform, %s", launch_rule_name(r.rule));
        case FAM_C1024:
            return snprintf(buf, len, "c1024_f32: loop 8 waves/wg x 1 wg/CU %s; <= %d resident sets: once kernel %d waves/wg %s",
                            launch_rule_name(r.rule), r.oneshot, C1024_ONCE_W, launch_rule_name(LR_DISPATCH));
        case FAM_TILED:
            return snprintf(buf, len, "tiled: cfg %s wg %d vec/wg %d lds %zu %s oneshot<=%d groups/wg", r.tiled.cfg, r.tiled.wg, r.tiled.t_per_wg,
                            r.tiled.lds, launch_rule_name(r.rule), r.oneshot);
        case FAM_STOCK: {
            char grid[48];
            if (r.stock.groups_per_wg > 0) snprintf(grid, sizeof grid, "%d groups/wg (table)", r.stock.groups_per_wg);
            else snprintf(grid, sizeof grid, "%d x resident set", r.stock.grid_mul);
            return snprintf(buf, len, "stockham: %s%s%s threads %d lds %zu %s grid %s%s", r.stock.wl ? "wave-local" : "workgroup",
                            r.stock.df ? " direct-first-stage" : " deposit", r.stock.fn ? "" : " run-time-plan", r.stock.threads, r.stock.lds,
                            launch_rule_name(r.rule), r.rule == LR_INORDER ? "resident set" : grid,
                            r.rule == LR_INORDER ? (r.oneshot ? " oneshot" : "") : "");
        }
        case FAM_BIG: {
            const BigPlan& b = r.big;
            char core[96];
            switch (b.core) {
                case BIG_RFFT2: snprintf(core, sizeof core, "real two-sweep tiles"); break;
                case BIG_TILES:
                    if (b.lens[2]) snprintf(core, sizeof core, "tiles %d x %d x %d (mode %d)", b.lens[0], b.lens[1], b.lens[2], b.tmode);
                    else snprintf(core, sizeof core, "tiles %d x %d (mode %d)%s", b.lens[0], b.lens[1], b.tmode, b.rfuse ? " real-rows" : "");
                    break;
                case BIG_STREAM: snprintf(core, sizeof core, "streaming %d x %d%s", b.lens[0], b.lens[1], (s->sub && s->sub->kernel == K_BIG) ? " (rows beyond LDS)" : ""); break;
                default: snprintf(core, sizeof core, "strided %d x %d", b.lens[0], b.lens[1]); break;
            }
            return snprintf(buf, len, "fourstep: %s; pre %d%s fuse_in %d col_in %d fuse_out %d post %d%s pair_after %d; %d sweeps", core, b.pre,
                            b.pre_separate ? "(separate)" : "", (int)b.fuse_in, (int)b.col_in, (int)b.fuse_out, b.post, b.post_separate ? "(separate)" : "",
                            (int)b.pair_after, b.sweeps);
        }
        case FAM_ONE: {
            const int pi = (&r == &s->route[1][0] && !s->is_double && s->transform == PFFFT_COMPLEX) ? 2 : (&r == &s->route[1][0] || &r == &s->route[1][1]) ? 1 : 0;
            const StockPlan& sp = s->one[pi];
            char rad[48] = "";
            for (int i = 0; i < sp.ns; ++i) snprintf(rad + strlen(rad), sizeof rad - strlen(rad), i ? " x %d" : "%d", sp.st[i].R);
            return snprintf(buf, len, "oneimage: one %d-thread workgroup per vector, stages %s in place, lds %zu %s; 1 sweep", sp.C, rad,
                            one_lds_bytes(sp, s->is_double != 0, s->transform == PFFFT_REAL), launch_rule_name(r.rule));
        }
        default: return snprintf(buf, len, "none");
    }
}
// the family a SIZE is built for (pffft_hip_kernel_name); a (direction, layout) of a power of two may still run its Stockham plan
static const char* setup_family(const Setup* s) {
    if (!ab().is(AB_NO_TINY) && (s->n == 16 || (s->n == 32 && !s->is_double))) return "tiny";
    switch (s->kernel) {
        case K_C1024_F32: return "c1024_f32";
        case K_TILED: return "tiled";
        case K_BIG: return "fourstep";
        // "stockham_rt": the plan has no compile-time twin and runs the run-time-plan kernel (0.3 of the roofline and less)
        default: return s->sk_ok ? (sub_is_fast(s) ? "stockham" : "stockham_rt") : "none";
    }
}
static int describe_setup(const Setup* s, char* buf, size_t len) {
    std::string out;
    char line[512];
    snprintf(line, sizeof line, "pffft_hip setup N=%d %s %s: core n=%d, family %s\n", s->N, s->transform == PFFFT_REAL ? "real" : "complex",
             s->is_double ? "f64" : "f32", s->n, setup_family(s));
    out += line;
    static const char* dn[2] = {"forward ", "backward"};
    static const char* on[2] = {"unordered", "ordered  "};
    for (int d = 0; d < 2; ++d)
        for (int o = 1; o >= 0; --o) {
            char body[400];
            describe_route(s, s->route[d][o], body, sizeof body);
            snprintf(line, sizeof line, "  %s %s: %s\n", dn[d], on[o], body);
            out += line;
        }
    if (buf && len) { const size_t c = out.size() < len - 1 ? out.size() : len - 1; memcpy(buf, out.data(), c); buf[c] = 0; }
    return (int)out.size();
}

// frequency shift (src/pf_mixer.cpp) immediately followed by the forward FFT, the usual SDR
// chain.  The batch is ONE stream of batch*N complex samples, sample g gets exp(j (phase_rad + 2 pi rate g)).
// N = 1024: fused into the load stage of the headline kernel (one pass over HBM); other sizes: mixer kernel
// into `out`, then the transform in place (two passes).
static int shift_transform_batch(Setup* s, const float* in, float* out, size_t batch, int ordered, double rate,
                                 double phase_rad, hipStream_t st) {
    if (!s || s->magic != MAGIC || s->is_double || s->transform != PFFFT_COMPLEX) {
        g_last_error = "pffft_hip: shift_transform_batch needs a complex single-precision setup";
        return (int)hipErrorInvalidHandle;
    }
    if (batch == 0) return 0;
    s = for_device(s);
    int rc = ensure_device<float>(s);
    if (rc) return rc;
    const double phase_turns = phase_rad / pfmix::MIX_TWO_PI;
    if (s->kernel == K_C1024_F32 && !ab().is(AB_AUX_DIRECT) && batch < (1ull << 32))   // AB_AUX_DIRECT: the two-pass composition (second route of the tests)
        return launch_c1024_mix(s, in, out, batch, ordered, rate, phase_turns, st);
    const double S[1][2] = {{std::cos(phase_rad), std::sin(phase_rad)}};
    rc = pfmix::launch_mix(reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(out), batch * (size_t)s->N, 1, S,
                           rate, false, st);
    if (rc) { g_last_error = pfmix::last_error; return rc; }
    return transform_batch<float>(s, out, out, batch, PFFFT_FORWARD, ordered, st);
}

template <typename T>
static int zreorder_batch(Setup* s, const T* in, T* out, size_t batch, int dir, hipStream_t st) {
    if (!s || s->magic != MAGIC) return (int)hipErrorInvalidHandle;
    if (batch == 0) return 0;
    s = for_device(s);
    // through an LDS image of the internal layout when a vector fits (fft_aux.h); AB_AUX_DIRECT = the direct kernel
    constexpr int CH = 16 / (int)sizeof(T), BCH = SkIbs<T>::v / CH;
    const size_t vimg = ((size_t)(s->n / 16) * BCH + 1) * 16;   // block image of one vector, bytes
    const size_t vbytes = s->vec_scalars * sizeof(T);
    // long batches of vectors <= 64 KiB: in-order streaming kernel with next-group prefetch (fft_aux.h); AB_AUX_NO_STREAM = off
    const AbSel sel = ab();
    const bool direct = sel.is(AB_AUX_DIRECT), inorder_small = sel.is(AB_INORDER_SMALL);
    if (vbytes <= ZRD_GROUP_BYTES && batch * vbytes >= ((size_t)64 << 20) && !direct && !sel.is(AB_AUX_NO_STREAM) && !inorder_small && in != out) {
        int rc = ensure_device<T>(s);
        if (rc) return rc;
        const int G = (int)(ZRD_GROUP_BYTES / vbytes);
        const bool to_canon = dir == PFFFT_FORWARD;
        const size_t img = to_canon ? (size_t)zrd_canon_img16<T>(s->n) * 16 : vimg;
        const size_t lds = (size_t)G * img + 16;
        if (lds <= LDS_MAX) {
            unsigned* ctr = take_counters(s, st);
            const int nchk = 2 * s->n / CH;
            const dim3 grid((unsigned)num_cus()), blk(ZRD_THREADS);
            const int real = s->transform == PFFFT_REAL;
            if (to_canon) {
                auto k = zreorder_dyn_kernel<T, 1>;
                if ((rc = allow_big_lds(k, lds))) return rc;
                hipLaunchKernelGGL(k, grid, blk, lds, st, in, out, batch, s->n, real, G, sk_magic(s->n / 4), sk_magic(nchk), ctr);
            } else {
                auto k = zreorder_dyn_kernel<T, 0>;
                if ((rc = allow_big_lds(k, lds))) return rc;
                hipLaunchKernelGGL(k, grid, blk, lds, st, in, out, batch, s->n, real, G, sk_magic(s->n / 4), sk_magic(nchk), ctr);
            }
            PF_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (vimg <= 128 * 1024 && !direct && in != out) {
        int rc = ensure_device<T>(s);
        if (rc) return rc;
        int G = (int)(16384 / vimg);
        if (G < 1) G = 1;
        const size_t lds = (size_t)G * vimg + 16;
        auto k = zreorder_lds_kernel<T>;
        if ((rc = allow_big_lds(k, lds))) return rc;
        size_t per_cu = LDS_MAX / lds;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        const size_t groups = (batch + G - 1) / G;
        size_t grid = (size_t)num_cus() * per_cu;
        if (grid > groups) grid = groups;
        // >= 128 KiB per atomic (one counter address serves ~80 M atomics/s), >= 8 chunks per workgroup
        size_t kc = (131072 + G * vimg - 1) / (G * vimg), cap = groups / (8 * grid);
        if (kc > cap) kc = cap;
        if (kc > 64) kc = 64;
        // static by default: 0.61-0.70 of the roofline against 0.47-0.62 with in-order chunks (AB_INORDER_SMALL) and
        // 0.41-0.61 for the direct kernel (tools/aux_bench.py)
        unsigned* ctr = (kc < 1 || !inorder_small) ? nullptr : take_counters(s, st);
        if (kc < 1) kc = 1;
        const int nchk = 2 * s->n / CH;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(ZR_THREADS), lds, st, in, out, batch, s->n,
                           (int)(s->transform == PFFFT_REAL), (int)(dir == PFFFT_FORWARD), G, sk_magic(s->n / 4),
                           sk_magic(nchk), ctr, (unsigned)kc);
        PF_CHECK(hipGetLastError());
        return 0;
    }
    size_t total = batch * (size_t)(s->n / 2);
    size_t grid = (total + 255) / 256;
    if (grid > (size_t)num_cus() * 16) grid = (size_t)num_cus() * 16;
    hipLaunchKernelGGL((zreorder_kernel<T>), dim3((unsigned)grid), dim3(256), 0, st, in, out, batch, s->n,
                       (int)(s->transform == PFFFT_REAL), (int)(dir == PFFFT_FORWARD));
    PF_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
static int zconvolve_batch(Setup* s, const T* a, const T* b, T* ab, T scaling, size_t batch, int accumulate,
                           int b_broadcast, hipStream_t st) {
    if (!s || s->magic != MAGIC) return (int)hipErrorInvalidHandle;
    if (batch == 0) return 0;
    s = for_device(s);
    const AbSel sel = ::pf::ab();      // (`ab` is also this function's output vector)
    const bool direct = sel.is(AB_AUX_DIRECT), inorder_small = sel.is(AB_INORDER_SMALL);
    size_t total = batch * (size_t)(s->n / 4);
    // float: streaming kernel, two pairs per thread with all loads issued first (fft_aux.h): 0.69-0.70 against 0.65-0.70
    // for the grid-stride kernel, which stays for double (0.65 vs 0.41) and as AB_AUX_DIRECT; in-order chunks (AB_INORDER_SMALL)
    // measured 0.57-0.60
    // long batches (>= 64 MiB per stream): in-order streaming kernel with DPP pair exchange (fft_aux.h); AB_AUX_NO_STREAM = off
    {
        const unsigned long long Q = 2ull * total * Zd<T>::UPG;   // 16-byte units in the batch
        if (!direct && !sel.is(AB_AUX_NO_STREAM) && !inorder_small && Q / Zd<T>::CHUNK >= 8192u &&
            Q / Zd<T>::CHUNK < 0xffffffffull) {
            int rc = ensure_device<T>(s);
            if (rc) return rc;
            unsigned* ctr = take_counters(s, st);
            const int real = s->transform == PFFFT_REAL;
            const dim3 grid((unsigned)num_cus()), blk(ZD_WAVES * 64);
            const unsigned nq = (unsigned)(s->n / 2) * Zd<T>::UPG;   // units per vector
#define PF_ZD(ACC, BC) hipLaunchKernelGGL((zconvolve_dyn_kernel<T, ACC, BC>), grid, blk, 0, st, a, b, ab, Q, nq, real, scaling, ctr)
            if (accumulate) { if (b_broadcast) PF_ZD(1, 1); else PF_ZD(1, 0); }
            else { if (b_broadcast) PF_ZD(0, 1); else PF_ZD(0, 0); }
#undef PF_ZD
            PF_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (!direct && sizeof(T) == 4) {
        int rc = ensure_device<T>(s);
        if (rc) return rc;
        const size_t chunks = (total + ZC_CHUNK - 1) / ZC_CHUNK;
        size_t grid = (size_t)num_cus() * 4;
        if (grid > chunks) grid = chunks;
        size_t kc = 4, cap = chunks / (8 * grid);
        if (kc > cap) kc = cap;
        unsigned* ctr = (kc < 1 || !inorder_small) ? nullptr : take_counters(s, st);
        if (kc < 1) kc = 1;
        const int real = s->transform == PFFFT_REAL;
        if (accumulate)
            hipLaunchKernelGGL((zconvolve_stream_kernel<T, 1>), dim3((unsigned)grid), dim3(ZC_THREADS), 0, st, a, b, ab, total,
                               (unsigned)(s->n / 4), real, scaling, b_broadcast, ctr, (unsigned)kc);
        else
            hipLaunchKernelGGL((zconvolve_stream_kernel<T, 0>), dim3((unsigned)grid), dim3(ZC_THREADS), 0, st, a, b, ab, total,
                               (unsigned)(s->n / 4), real, scaling, b_broadcast, ctr, (unsigned)kc);
        PF_CHECK(hipGetLastError());
        return 0;
    }
    size_t grid = (total + 255) / 256;
    if (grid > (size_t)num_cus() * 16) grid = (size_t)num_cus() * 16;
    const size_t vs = s->vec_scalars;
    const int is_real = s->transform == PFFFT_REAL;
    if (accumulate)
        hipLaunchKernelGGL((zconvolve_kernel<T, 1>), dim3((unsigned)grid), dim3(256), 0, st, a, b, ab, batch, s->n,
                           is_real, scaling, vs, b_broadcast ? (size_t)0 : vs);
    else
        hipLaunchKernelGGL((zconvolve_kernel<T, 0>), dim3((unsigned)grid), dim3(256), 0, st, a, b, ab, batch, s->n,
                           is_real, scaling, vs, b_broadcast ? (size_t)0 : vs);
    PF_CHECK(hipGetLastError());
    return 0;
}

// out += x, 16-byte units (the accumulate leg of the composed convolution)
template <typename T>
__global__ void vec_add_kernel(const T* __restrict__ x, T* __restrict__ out, size_t units) {
    const vec4<float>* x16 = reinterpret_cast<const vec4<float>*>(x);
    vec4<float>* o16 = reinterpret_cast<vec4<float>*>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
        const vec4<float> a = x16[i], b = o16[i];
        if constexpr (sizeof(T) == 4) {
            vec4<float> r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w;
            o16[i] = r;
        } else {
            const vec2<double> da = __builtin_bit_cast(vec2<double>, a), db = __builtin_bit_cast(vec2<double>, b);
            vec2<double> r; r.x = da.x + db.x; r.y = da.y + db.y;
            o16[i] = __builtin_bit_cast(vec4<float>, r);
        }
    }
}

// pffft_hip_convolve_batch: out (+)= backward(forward(in) . H) scaling.  One kernel where fft_conv.h has one (conv_tu.hip);
// otherwise the three batched entries through a per-stream spectrum image (AB_CONV_COMPOSED forces the composition: the second route of the tests).
template <typename T>
static int convolve_batch(Setup* s, const T* in, const T* H, T* out, T scaling, size_t batch, int accumulate, int h_broadcast,
                          hipStream_t st) {
    if (!s || s->magic != MAGIC || s->is_double != (sizeof(T) == 8)) {
        g_last_error = "pffft_hip: bad setup handle";
        return (int)hipErrorInvalidHandle;
    }
    if (batch == 0) return 0;
    s = for_device(s);
    int rc = ensure_device<T>(s);
    if (rc) return rc;
    if (h_broadcast && !ab().is(AB_CONV_COMPOSED)) {
        rc = launch_conv_fused(s, in, H, out, batch, (double)scaling, accumulate, st);
        if (rc != -1) return rc;
    }
    const size_t bytes = batch * s->vec_scalars * sizeof(T);
    std::lock_guard<std::mutex> lk(s->conv_mu);
    Setup::Scratch* scp = nullptr;
    if ((rc = stream_scratch(s->conv_scratch, s->conv_clock, st, &scp))) return rc;
    Setup::Scratch& sc = *scp;
    if ((rc = scratch_grow(s, sc, 0, bytes))) return rc;
    T* X = (T*)sc.buf[0];
    if ((rc = transform_batch<T>(s, in, X, batch, PFFFT_FORWARD, 0, st))) return rc;
    if ((rc = zconvolve_batch<T>(s, X, H, X, scaling, batch, 0, h_broadcast, st))) return rc;
    if (!accumulate) return transform_batch<T>(s, X, out, batch, PFFFT_BACKWARD, 0, st);
    if ((rc = transform_batch<T>(s, X, X, batch, PFFFT_BACKWARD, 0, st))) return rc;
    const size_t units = bytes / 16;
    const unsigned grid = (unsigned)std::min<size_t>((units + 255) / 256, (size_t)num_cus() * 16);
    hipLaunchKernelGGL((vec_add_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)X, out, units);
    PF_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// legacy single-vector entries: host pointers are staged, device pointers are used in place
// ------------------------------------------------------------------------------------------------
static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static int stage_buf(Setup* s, int slot, size_t bytes, void** out) {
    if (s->stage_bytes[slot] < bytes) {
        if (s->d_stage[slot]) (void)hipFree(s->d_stage[slot]);
        s->d_stage[slot] = nullptr; s->stage_bytes[slot] = 0;
        PF_CHECK(hipMalloc(&s->d_stage[slot], bytes));
        s->stage_bytes[slot] = bytes;
    }
    *out = s->d_stage[slot];
    return 0;
}

static int pinned_buf(Setup* s, int slot, size_t bytes, void** out) {
    if (s->hstage_bytes[slot] < bytes) {
        if (s->h_stage[slot]) (void)hipHostFree(s->h_stage[slot]);
        s->h_stage[slot] = nullptr; s->hstage_bytes[slot] = 0;
        PF_CHECK(hipHostMalloc(&s->h_stage[slot], bytes, hipHostMallocDefault));
        s->hstage_bytes[slot] = bytes;
    }
    *out = s->h_stage[slot];
    return 0;
}

// Host-pointer calls on small vectors: no DMA copies at all.  The vector is copied (CPU memcpy, < 1 us) into a pinned host
// image that the kernel reads over PCIe directly, the kernel writes its result into another pinned image, one stream
// synchronisation, CPU memcpy out: one launch + one sync per call instead of two synchronous hipMemcpy around them
// (measured: tools/legacy_bench.py).  Vectors above ZC_LIMIT keep the device staging (kernels there may sweep `out`
// more than once).  PFFFT_HIP_NO_ZEROCOPY=1 switches it off (A/B).
constexpr size_t ZC_LIMIT = 256 * 1024;
static bool zero_copy_enabled() { return env().zero_copy; }

// run `fn(d_in..., d_out)` with up to 3 inputs + 1 output vector of `bytes` bytes each
template <typename T, typename F>
static int legacy_run(Setup* s, const T* const* ins, int nin, T* out, bool out_is_inout, F&& fn) {
    if (!s || s->magic != MAGIC || s->is_double != (sizeof(T) == 8)) {
        g_last_error = "pffft_hip: bad setup handle";
        return (int)hipErrorInvalidHandle;
    }
    s = for_device(s);        // (the staging buffers of the calling thread's device)
    const size_t bytes = s->vec_scalars * sizeof(T);
    // the staging buffers belong to the setup; the mutex keeps concurrent callers correct
    // (the reference allows a setup to be shared between threads, include/pffft/pffft.h:102-105)
    std::lock_guard<std::mutex> lk(s->stage_mu);
    if (bytes <= ZC_LIMIT && zero_copy_enabled() && s->kernel != K_BIG) {
        bool any_dev = is_device_ptr(out);
        for (int i = 0; i < nin && !any_dev; ++i) any_dev = is_device_ptr(ins[i]);
        // the pinned images are allocated up front; if the host cannot pin memory the device staging below still works
        void* pins[4] = {nullptr, nullptr, nullptr, nullptr};
        bool pinned_ok = !any_dev;
        for (int k = 0; k <= nin && pinned_ok; ++k) pinned_ok = pinned_buf(s, k, bytes, &pins[k]) == 0;
        if (!any_dev && pinned_ok) {
            const T* h_in[3] = {nullptr, nullptr, nullptr};
            void* po; int rc = pinned_buf(s, 0, bytes, &po); if (rc) return rc;
            T* h_out = (T*)po;
            bool out_loaded = false;
            if (out_is_inout) { memcpy(h_out, out, bytes); out_loaded = true; }
            int slot = 1;
            for (int i = 0; i < nin; ++i) {
                if (ins[i] == out) { if (!out_loaded) { memcpy(h_out, out, bytes); out_loaded = true; } h_in[i] = h_out; continue; }
                bool dup = false;
                for (int j = 0; j < i; ++j) if (ins[j] == ins[i]) { h_in[i] = h_in[j]; dup = true; break; }
                if (dup) continue;
                void* p; rc = pinned_buf(s, slot++, bytes, &p); if (rc) return rc;
                memcpy(p, ins[i], bytes);
                h_in[i] = (const T*)p;
            }
            rc = fn(h_in, h_out);
            if (rc) return rc;
            PF_CHECK(hipStreamSynchronize(nullptr));
            memcpy(out, h_out, bytes);
            return 0;
        }
    }
    const T* d_in[3] = {nullptr, nullptr, nullptr};
    T* d_out = nullptr;
    const bool out_dev = is_device_ptr(out);
    int slot = 0;
    if (out_dev) d_out = out;
    else {
        void* p; int rc = stage_buf(s, slot++, bytes, &p); if (rc) return rc;
        d_out = (T*)p;
        if (out_is_inout) PF_CHECK(hipMemcpy(d_out, out, bytes, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < nin; ++i) {
        if (ins[i] == out) { d_in[i] = d_out; if (!out_dev && !out_is_inout) PF_CHECK(hipMemcpy(d_out, out, bytes, hipMemcpyHostToDevice)); continue; }
        bool dup = false;
        for (int j = 0; j < i; ++j) if (ins[j] == ins[i]) { d_in[i] = d_in[j]; dup = true; break; }
        if (dup) continue;
        if (is_device_ptr(ins[i])) d_in[i] = ins[i];
        else {
            void* p; int rc = stage_buf(s, slot++, bytes, &p); if (rc) return rc;
            PF_CHECK(hipMemcpy(p, ins[i], bytes, hipMemcpyHostToDevice));
            d_in[i] = (const T*)p;
        }
    }
    int rc = fn(d_in, d_out);
    if (rc) return rc;
    if (!out_dev) PF_CHECK(hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost));
    else PF_CHECK(hipStreamSynchronize(nullptr));
    return 0;
}

// bytes of the caller's output vector the fail-soft path may overwrite: none unless the handle itself is valid
template <typename T>
static size_t legacy_out_bytes(const Setup* s) {
    return (s && s->magic == MAGIC && s->is_double == (sizeof(T) == 8)) ? s->vec_scalars * sizeof(T) : 0;
}

template <typename T>
static void legacy_transform(Setup* s, const T* in, T* out, int dir, int ordered, const char* name) {
    const T* ins[1] = {in};
    int rc = legacy_run<T>(s, ins, 1, out, false, [&](const T* const* di, T* dout) {
        return transform_batch<T>(s, di[0], dout, 1, dir, ordered, nullptr);
    });
    if (rc) legacy_fatal(rc, name, out, legacy_out_bytes<T>(s), !is_device_ptr(out));
}

template <typename T>
static void legacy_zreorder(Setup* s, const T* in, T* out, int dir, const char* name) {
    const T* ins[1] = {in};
    int rc = legacy_run<T>(s, ins, 1, out, false, [&](const T* const* di, T* dout) {
        return zreorder_batch<T>(s, di[0], dout, 1, dir, nullptr);
    });
    if (rc) legacy_fatal(rc, name, out, legacy_out_bytes<T>(s), !is_device_ptr(out));
}

template <typename T>
static void legacy_zconvolve(Setup* s, const T* a, const T* b, T* ab, T scaling, int accumulate, const char* name) {
    const T* ins[2] = {a, b};
    int rc = legacy_run<T>(s, ins, 2, ab, accumulate != 0, [&](const T* const* di, T* dout) {
        return zconvolve_batch<T>(s, di[0], di[1], dout, scaling, 1, accumulate, 0, nullptr);
    });
    if (rc) legacy_fatal(rc, name, ab, legacy_out_bytes<T>(s), !is_device_ptr(ab));
}

// Layout self-test standing in for validate_pffft_simd_ex (src/pffft_priv_impl.h:1889-2225, which
// unit-tests the SIMD macros): checks on the host that the internal-layout map used by the kernels
// is a permutation with the documented fixed points.  Returns the number of errors.
static int host_bin_of(int v, int l, int n, int is_real) {
    int b = v >> 3, q = (v >> 1) & 3, t = 4 * b + l;
    if (!is_real) return q * (n >> 2) + t;
    switch (q) {
        case 0: return t;
        case 2: return (n >> 1) + t;
        case 1: return t ? (n >> 1) - t : (n >> 2);
        default: return t ? n - t : 3 * (n >> 2);
    }
}
static int validate_layout(FILE* dbg) {
    int errors = 0;
    for (int is_real = 0; is_real < 2; ++is_real)
        for (int n : {16, 32, 48, 80, 1024}) {
            std::vector<int> seen(2 * n, 0);
            for (int v = 0; v < n / 2; ++v)
                for (int l = 0; l < 4; ++l) {
                    int idx = 2 * host_bin_of(v, l, n, is_real) + (v & 1);
                    if (idx < 0 || idx >= 2 * n || seen[idx]++) ++errors;
                }
            if (host_bin_of(0, 0, n, is_real) != 0) ++errors;
            if (dbg) fprintf(dbg, "pffft_hip layout check n=%d real=%d: errors so far %d\n", n, is_real, errors);
        }
    return errors;
}

}  // namespace pf

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
struct PFFFT_Setup : pf::Setup {};
struct PFFFTD_Setup : pf::Setup {};

#define PF_EXPORT extern "C" __attribute__((visibility("default")))

#define PF_DEFINE_API(PFX, SETUP, T, ISD, ARCHSTR)                                                                  \
    PF_EXPORT SETUP* PFX##_new_setup(int N, pffft_transform_t tr) {                                                 \
        return static_cast<SETUP*>(pf::new_setup(N, (int)tr, ISD));                                                 \
    }                                                                                                               \
    PF_EXPORT void PFX##_destroy_setup(SETUP* s) { pf::destroy_setup(s); }                                          \
    PF_EXPORT void PFX##_transform(SETUP* s, const T* in, T* out, T* work, pffft_direction_t d) {                   \
        (void)work; pf::legacy_transform<T>(s, in, out, (int)d, 0, #PFX "_transform");                              \
    }                                                                                                               \
    PF_EXPORT void PFX##_transform_ordered(SETUP* s, const T* in, T* out, T* work, pffft_direction_t d) {           \
        (void)work; pf::legacy_transform<T>(s, in, out, (int)d, 1, #PFX "_transform_ordered");                      \
    }                                                                                                               \
    PF_EXPORT void PFX##_zreorder(SETUP* s, const T* in, T* out, pffft_direction_t d) {                             \
        pf::legacy_zreorder<T>(s, in, out, (int)d, #PFX "_zreorder");                                               \
    }                                                                                                               \
    PF_EXPORT void PFX##_zconvolve_accumulate(SETUP* s, const T* a, const T* b, T* ab, T sc) {                      \
        pf::legacy_zconvolve<T>(s, a, b, ab, sc, 1, #PFX "_zconvolve_accumulate");                                  \
    }                                                                                                               \
    PF_EXPORT void PFX##_zconvolve_no_accu(SETUP* s, const T* a, const T* b, T* ab, T sc) {                         \
        pf::legacy_zconvolve<T>(s, a, b, ab, sc, 0, #PFX "_zconvolve_no_accu");                                     \
    }                                                                                                               \
    PF_EXPORT int PFX##_simd_size(void) { return pf::SIMD; }                                                        \
    PF_EXPORT const char* PFX##_simd_arch(void) { return ARCHSTR; }                                                 \
    PF_EXPORT int PFX##_min_fft_size(pffft_transform_t tr) { return pf::min_fft_size((int)tr); }                    \
    PF_EXPORT int PFX##_is_valid_size(int N, pffft_transform_t tr) { return pf::is_valid_size(N, (int)tr); }        \
    PF_EXPORT int PFX##_nearest_transform_size(int N, pffft_transform_t tr, int higher) {                           \
        return pf::nearest_size(N, (int)tr, higher);                                                                \
    }                                                                                                               \
    PF_EXPORT int PFX##_next_power_of_two(int N) { return pf::next_pow2(N); }                                       \
    PF_EXPORT int PFX##_is_power_of_two(int N) { return pf::is_pow2(N); }                                           \
    PF_EXPORT void* PFX##_aligned_malloc(size_t nb) { return pf::aligned_malloc64(nb); }                            \
    PF_EXPORT void PFX##_aligned_free(void* p) { pf::aligned_free64(p); }                                           \
    PF_EXPORT int validate_##PFX##_simd_ex(void* dbg) { return pf::validate_layout((FILE*)dbg); }                   \
    PF_EXPORT int validate_##PFX##_simd(void) { return pf::validate_layout(nullptr); }                              \
    PF_EXPORT int PFX##_hip_transform_batch(SETUP* s, const T* in, T* out, size_t batch, pffft_direction_t d,       \
                                            int ordered, void* stream) {                                            \
        return pf::transform_batch<T>(s, in, out, batch, (int)d, ordered, (hipStream_t)stream);                     \
    }                                                                                                               \
    PF_EXPORT int PFX##_hip_zreorder_batch(SETUP* s, const T* in, T* out, size_t batch, pffft_direction_t d,        \
                                           void* stream) {                                                          \
        return pf::zreorder_batch<T>(s, in, out, batch, (int)d, (hipStream_t)stream);                               \
    }                                                                                                               \
    PF_EXPORT int PFX##_hip_zconvolve_batch(SETUP* s, const T* a, const T* b, T* ab, T sc, size_t batch,            \
                                            int accumulate, int b_broadcast, void* stream) {                        \
        return pf::zconvolve_batch<T>(s, a, b, ab, sc, batch, accumulate, b_broadcast, (hipStream_t)stream);        \
    }                                                                                                               \
    PF_EXPORT int PFX##_hip_convolve_batch(SETUP* s, const T* in, const T* H, T* out, T sc, size_t batch,           \
                                           int accumulate, int h_broadcast, void* stream) {                         \
        return pf::convolve_batch<T>(s, in, H, out, sc, batch, accumulate, h_broadcast, (hipStream_t)stream);       \
    }

// Batch shards over several devices from ONE host thread (independent units, no exchange step): part p is transformed by
// setups[p] on devices[p] - hipSetDevice, then the batched entry on streams[p] (NULL: that device's default stream); every launch is
// asynchronous, so the devices run concurrently.  The caller's current device is restored.  Round 6: setups[p] may be the SAME setup in
// every slot (for_device: a setup holds device state per device it is used on) or a setup of its own per part, as before.  Returns the
// first error (0 = all enqueued).
template <typename T, typename SETUP>
static int transform_batch_multi(int nparts, const int* devices, SETUP* const* setups, const T* const* in, T* const* out, const size_t* batches,
                                 int dir, int ordered, void* const* streams) {
    if (nparts < 0 || (nparts > 0 && (!devices || !setups || !in || !out || !batches))) {
        pf::g_last_error = "pffft_hip: transform_batch_multi needs devices, setups, in, out and batches";
        return (int)hipErrorInvalidValue;
    }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    int rc = 0;
    for (int p = 0; p < nparts && !rc; ++p) {
        hipError_t e = hipSetDevice(devices[p]);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = pf::fail(e, "hipSetDevice"); break; }   // (the runtime's sticky error is consumed here: the next launch checks it)
        rc = pf::transform_batch<T>(setups[p], in[p], out[p], batches[p], dir, ordered, (hipStream_t)(streams ? streams[p] : nullptr));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
PF_EXPORT int pffft_hip_transform_batch_multi(int nparts, const int* devices, PFFFT_Setup* const* setups, const float* const* in, float* const* out,
                                              const size_t* batches, pffft_direction_t d, int ordered, void* const* streams) {
    return transform_batch_multi<float, PFFFT_Setup>(nparts, devices, setups, in, out, batches, (int)d, ordered, streams);
}
PF_EXPORT int pffftd_hip_transform_batch_multi(int nparts, const int* devices, PFFFTD_Setup* const* setups, const double* const* in, double* const* out,
                                               const size_t* batches, pffft_direction_t d, int ordered, void* const* streams) {
    return transform_batch_multi<double, PFFFTD_Setup>(nparts, devices, setups, in, out, batches, (int)d, ordered, streams);
}

PF_DEFINE_API(pffft, PFFFT_Setup, float, 0, "HIP-gfx950")
PF_DEFINE_API(pffftd, PFFFTD_Setup, double, 1, "HIP-gfx950")

PF_EXPORT int pffft_hip_shift_transform_batch(PFFFT_Setup* s, const float* in, float* out, size_t batch, int ordered,
                                              double rate, double phase_rad, void* stream) {
    return pf::shift_transform_batch(reinterpret_cast<pf::Setup*>(s), in, out, batch, ordered, rate, phase_rad,
                                     (hipStream_t)stream);
}

PF_EXPORT const char* pffft_hip_kernel_name(const void* setup) {
    const pf::Setup* s = static_cast<const pf::Setup*>(setup);
    if (!s || s->magic != pf::MAGIC) return "invalid";
    return pf::setup_family(s);
}
// resident workgroups per CU of the LDS-resident kernel a (direction, layout) runs on, as the launcher sees it (the occupancy query of
// the runtime for the route's kernel, threads and LDS bytes); 0 where the route has no single persistent kernel.  Needs a device.
PF_EXPORT int pffft_hip_route_occupancy(const void* setup, int dir, int ordered) {
    const pf::Setup* s = static_cast<const pf::Setup*>(setup);
    if (!s || s->magic != pf::MAGIC || dir < 0 || dir > 1) return -1;
    const pf::Route& r = s->route[dir][ordered ? 1 : 0];
    int per_cu = 0;
    if (r.fam == pf::FAM_TILED) { if (pf::allow_big_lds_impl(r.tiled.fn, r.tiled.lds) || pf::cached_occupancy(r.tiled.fn, r.tiled.wg, r.tiled.lds, &per_cu)) return -1; }
    else if (r.fam == pf::FAM_STOCK && r.stock.fn) { if (pf::allow_big_lds_impl(r.stock.fn, r.stock.lds) || pf::cached_occupancy(r.stock.fn, r.stock.threads, r.stock.lds, &per_cu)) return -1; }
    else if (r.fam == pf::FAM_ONE) {
        const bool real = s->transform == PFFFT_REAL;
        const int flags = (real ? 8 : 0) | (dir ? 4 : 0) | (!ordered ? (dir ? 1 : 2) : 0);
        const void* fn = pf::one_kernel_ptr(s->is_double != 0, flags);
        const int pi = (flags == 5 && !s->is_double) ? 2 : dir;
        const size_t lds = pf::one_lds_bytes(s->one[pi], s->is_double != 0, real);
        if (pf::allow_big_lds_impl(fn, lds) || pf::cached_occupancy(fn, s->one[pi].C, lds, &per_cu)) return -1;
    }
    return per_cu;
}
PF_EXPORT int pffft_hip_describe(const void* setup, char* buf, size_t len) {
    const pf::Setup* s = static_cast<const pf::Setup*>(setup);
    if (!s || s->magic != pf::MAGIC) { if (buf && len) buf[0] = 0; return -1; }
    return pf::describe_setup(s, buf, len);
}
PF_EXPORT int pffft_hip_tile_plan(long long n, int is_double, int deep, int lengths[3]) {
    if (!lengths) return 0;
    return pf::tile_plan_lengths(n, is_double != 0, deep < 0 || deep > 2 ? 1 : deep, lengths);
}
PF_EXPORT int pffft_hip_tile_candidates(long long n, int is_double, int* out, int max) {
    return (out && max > 0) ? pf::tile_plan_candidates(n, is_double != 0, out, max) : 0;
}
PF_EXPORT int pffft_hip_tile_override(long long n, int is_double, int l1, int g1, int l2, int g2) {
    return pf::tile_plan_override(n, is_double != 0, l1, g1, l2, g2);
}
PF_EXPORT const char* pffft_hip_last_error(void) { return pf::g_last_error.c_str(); }
PF_EXPORT unsigned pffft_hip_error_count(void) { return pf::g_error_count.load(); }
// devices the setup holds tables / counters / scratch on right now (the device it bound to first, then its replicas; a key >= 64 is the
// test hook AB_FAKE_DEVICE); fills out[0 .. max), returns the count
PF_EXPORT int pffft_hip_setup_devices(const void* setup, int* out, int max) {
    return pf::setup_devices(const_cast<pf::Setup*>(static_cast<const pf::Setup*>(setup)), out, max < 0 ? 0 : max);
}
PF_EXPORT int pffft_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
PF_EXPORT void pffft_hip_set_variant(int v) { pf::g_ab_raw = v; }
PF_EXPORT int pffft_hip_has_variants(void) {
#ifdef PFFFT_HIP_VARIANTS
    return 1;
#else
    return 0;
#endif
}

#include "pffastconv_impl.h"
