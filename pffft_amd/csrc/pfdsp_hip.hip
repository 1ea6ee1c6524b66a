// libpfdsp_hip.so — the PFDSP frequency-shift mixers (reference: src/pf_mixer.cpp, API
// include/pffft/pf_mixer.h:61-280) for MI355X.  ABI: include/pfdsp_hip.h.
//
// The reference has ten algorithms (A..J) because a CPU cannot afford sinf/cosf per sample: they trade
// accuracy for speed with tables, angle-addition recurrences and recursive oscillators, all of them
// SEQUENTIAL in the sample index.  On the GPU the mixer is a pure streaming op (8 B in, 8 B out per
// sample, HBM-bound) with ~100 free VALU slots per sample, so every entry runs ONE closed-form kernel:
//
//     out[i] = in[i] * S[i mod LANES] * exp(j 2 pi frac(step * (i div LANES)))
//
// where S[] are the (up to 8) lane phasors the algorithm keeps as its state and `step` is the angle of
// one block step in turns, held in double so the reduction frac() is exact for any stream position.
// The host wrappers below translate each algorithm's state/return contract to (S, step) and advance the
// state in double precision.  There is NO CPU arithmetic path: the wrappers only do O(LANES) state math.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/pfdsp_hip.h"
#include "pfdsp_mix.h"

#define PD_EXPORT extern "C" __attribute__((visibility("default")))

namespace pd {

static thread_local std::string g_last_error;

static int fail(hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "pfdsp_hip: %s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
    g_last_error = buf;
    return (int)e;
}
#define PD_CHECK(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return fail(_e, #expr);    \
    } while (0)

constexpr double PI_D = 3.14159265358979323846264338327950288;
constexpr double TWO_PI_D = 2.0 * PI_D;
// the reference's own pi: #define PI ((float)3.14159265358979323846) (src/pf_mixer.cpp:40).  Phase
// increments are formed from it in float there (e.g. :147 rate*PI), so the wrappers do the same.
constexpr float PI_F = (float)3.14159265358979323846;

using pfmix::launch_mix;

// ------------------------------------------------------------------------------------------------
// legacy entries: host pointers are staged through one grow-only device buffer, device pointers are
// used in place; the call returns when the result is where the caller expects it (CPU-library semantics)
// ------------------------------------------------------------------------------------------------
static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static std::mutex g_stage_mu;
static void* g_stage = nullptr;
static size_t g_stage_bytes = 0;

// blocks up to ZC_LIMIT on host pointers: CPU memcpy into one pinned host image, the kernel rotates it in place over
// PCIe, one synchronisation, CPU memcpy out — no DMA copies (same scheme as the legacy FFT entries, pffft_hip.hip)
constexpr size_t ZC_LIMIT = 256 * 1024;
static void* g_pinned = nullptr;
static size_t g_pinned_bytes = 0;

static int legacy_mix(const complexf* in, complexf* out, size_t n, int lanes, const double (*S)[2], double step_turns,
                      bool gen) {
    if (n == 0) return 0;
    const size_t bytes = n * sizeof(complexf);
    const bool out_dev = is_device_ptr(out);
    const bool in_dev = gen || is_device_ptr(in);
    std::unique_lock<std::mutex> lk(g_stage_mu, std::defer_lock);
    static const bool zc_on = [] { const char* e = getenv("PFFFT_HIP_NO_ZEROCOPY"); return !(e && e[0] == '1'); }();
    if (zc_on && bytes <= ZC_LIMIT && !out_dev && (gen || !is_device_ptr(in))) {
        lk.lock();
        if (g_pinned_bytes < bytes) {
            if (g_pinned) (void)hipHostFree(g_pinned);
            g_pinned = nullptr; g_pinned_bytes = 0;
            PD_CHECK(hipHostMalloc(&g_pinned, ZC_LIMIT, hipHostMallocDefault));
            g_pinned_bytes = ZC_LIMIT;
        }
        float2* img = reinterpret_cast<float2*>(g_pinned);
        if (!gen) memcpy(img, in, bytes);
        int rc = launch_mix(gen ? nullptr : img, img, n, lanes, S, step_turns, gen, nullptr);
        if (rc) { g_last_error = pfmix::last_error; return rc; }
        PD_CHECK(hipStreamSynchronize(nullptr));
        memcpy(out, img, bytes);
        return 0;
    }
    const float2* d_in = reinterpret_cast<const float2*>(in);
    float2* d_out = reinterpret_cast<float2*>(out);
    if (!out_dev || !in_dev) {
        lk.lock();
        if (g_stage_bytes < bytes) {
            if (g_stage) (void)hipFree(g_stage);
            g_stage = nullptr; g_stage_bytes = 0;
            PD_CHECK(hipMalloc(&g_stage, bytes));
            g_stage_bytes = bytes;
        }
        // one staging image serves both sides: the kernel is elementwise, in place is legal
        if (!in_dev) {
            PD_CHECK(hipMemcpy(g_stage, in, bytes, hipMemcpyHostToDevice));
            d_in = reinterpret_cast<const float2*>(g_stage);
            if (out_dev) d_out = reinterpret_cast<float2*>(out);
        }
        if (!out_dev) d_out = reinterpret_cast<float2*>(g_stage);
    }
    int rc = launch_mix(d_in, d_out, n, lanes, S, step_turns, gen, nullptr);
    if (rc) { g_last_error = pfmix::last_error; return rc; }
    if (!out_dev) PD_CHECK(hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost));
    else PD_CHECK(hipStreamSynchronize(nullptr));
    return 0;
}

// The reference's mixer entries cannot fail; like the transform entries of libpffft_hip.so they FAIL SOFT here: one line
// on stderr (the first 8 per process), the output samples filled with NaN, the failure counted in
// pfdsp_hip_error_count().  PFFFT_HIP_ABORT=1 turns a failure into abort().
static std::atomic<unsigned> g_error_count{0};
static void legacy_fatal(int code, const char* entry, complexf* out, size_t n) {
    static const bool fail_fast = [] { const char* e = getenv("PFFFT_HIP_ABORT"); return e && e[0] == '1'; }();
    const unsigned nth = g_error_count.fetch_add(1);
    if (nth < 8 || fail_fast)
        fprintf(stderr, "%s: HIP path failed (%d): %s%s\n", entry, code, g_last_error.c_str(),
                fail_fast ? "" : " -- output filled with NaN (PFFFT_HIP_ABORT=1 aborts instead)");
    if (fail_fast) abort();
    if (out && n) {
        if (!is_device_ptr(out)) memset(out, 0xFF, n * sizeof(complexf));
        else if (hipMemset(out, 0xFF, n * sizeof(complexf)) != hipSuccess) (void)hipGetLastError();
    }
}

static void mix_or_die(const char* entry, const complexf* in, complexf* out, long n, int lanes, const double (*S)[2],
                       double step_turns, bool gen = false) {
    if (n <= 0) return;
    int rc = legacy_mix(in, out, (size_t)n, lanes, S, step_turns, gen);
    if (rc) legacy_fatal(rc, entry, out, (size_t)n);
}

// ---- state arithmetic (double) ----
static inline void unit(double ang, double (&p)[2]) { p[0] = std::cos(ang); p[1] = std::sin(ang); }
static inline void rot(const double (&s)[2], double ang, double (&o)[2]) {
    const double c = std::cos(ang), sn = std::sin(ang);
    const double a = s[0] * c - s[1] * sn, b = s[0] * sn + s[1] * c;
    o[0] = a; o[1] = b;
}
// angle (radians, reduced to [-pi, pi]) of n steps of inc
static inline double reduced(double inc, double n) {
    double t = (inc / TWO_PI_D) * n;
    t -= std::rint(t);
    return t * TWO_PI_D;
}
// the reference's wrap loops, applied to an exactly accumulated phase
// (with the true 2 pi, not the reference's float one: the returned phase must continue the samples just written)
static float wrap_0_2pi(double ph) {      // while(phase>2*PI) phase-=2*PI; while(phase<0) phase+=2*PI;  (:162-163)
    const double tp = TWO_PI_D;
    if (ph > tp) ph -= tp * std::ceil((ph - tp) / tp);
    if (ph < 0) ph += tp * std::ceil(-ph / tp);
    return (float)ph;
}
static float wrap_pm_pi(double ph) {      // while(p>PI) p-=2*PI; while(p<-PI) p+=2*PI;  (:281-283)
    const double p = PI_D, tp = TWO_PI_D;
    if (ph > p) ph -= tp * std::ceil((ph - p) / tp);
    if (ph < -p) ph += tp * std::ceil((-p - ph) / tp);
    return (float)ph;
}

// effective rotation angle of one step of the recursive oscillator with the GIVEN (rounded) constants:
// the update u' = (u - k1 v) - k1 v', v' = v + k2 (u - k1 v)  (src/pf_mixer.cpp:962-967) is a product of
// three shears with trace 2 - 2 k1 k2 = 2 cos(theta)  ->  sin(theta/2) = sqrt(k1 k2 / 2)
static double osc_angle(float k1, float k2) {
    double h = 0.5 * (double)k1 * (double)k2;
    if (h < 0) h = 0;
    if (h > 1) h = 1;
    const double th = 2.0 * std::asin(std::sqrt(h));
    return k1 < 0 ? -th : th;
}

template <int NL, typename ConfT, typename StateT>
static void osc_update_rate(float rate, ConfT* conf, StateT* state) {   // :898-921 / :1043-1066, float math as there
    const float inc_s = rate * PI_F;
    const float k1 = tanf(0.5f * inc_s);
    const float k2 = 2 * k1 / (1 + k1 * k1);
    for (int j = 1; j < NL; ++j) {
        state->u_cos[j] = state->u_cos[j - 1];
        state->v_sin[j] = state->v_sin[j - 1];
        float tmp = state->u_cos[j] - k1 * state->v_sin[j];
        state->v_sin[j] += k2 * tmp;
        state->u_cos[j] = tmp - k1 * state->v_sin[j];
    }
    float inc_b = inc_s * NL;
    while (inc_b > PI_F) inc_b -= 2 * PI_F;
    while (inc_b < -PI_F) inc_b += 2 * PI_F;
    conf->k1 = tanf(0.5f * inc_b);
    conf->k2 = 2 * conf->k1 / (1 + conf->k1 * conf->k1);
}

template <int NL, typename ConfT, typename StateT>
static void osc_init(float rate, float starting_phase, ConfT* conf, StateT* state) {  // :923-936
    if (starting_phase != 0.0F) { state->u_cos[0] = cosf(starting_phase); state->v_sin[0] = sinf(starting_phase); }
    else { state->u_cos[0] = 1.0F; state->v_sin[0] = 0.0F; }
    osc_update_rate<NL>(rate, conf, state);
}

template <int NL, typename ConfT, typename StateT>
static void osc_run(const char* entry, const complexf* in, complexf* out, int size, const ConfT* conf, StateT* state,
                    bool gen) {
    const long nblk = size / NL;              // the reference loops over size / SIMD_SZ full blocks
    if (nblk <= 0) return;
    double S[8][2];
    for (int j = 0; j < NL; ++j) { S[j][0] = state->u_cos[j]; S[j][1] = state->v_sin[j]; }
    const double th = osc_angle(conf->k1, conf->k2);
    mix_or_die(entry, in, out, nblk * NL, NL, S, th / TWO_PI_D, gen);
    const double adv = reduced(th, (double)nblk);
    for (int j = 0; j < NL; ++j) {
        double o[2]; rot(S[j], adv, o);
        state->u_cos[j] = (float)o[0]; state->v_sin[j] = (float)o[1];
    }
}

// the four lane phasors of algorithms F/G/H: start the kernel from them, advance them by n*inc
template <typename D>
static void lanes4_run(const char* entry, complexf* in_out, int n, D* d) {
    if (n <= 0) return;
    double S[8][2];
    for (int k = 0; k < 4; ++k) { S[k][0] = d->phase_state_i[k]; S[k][1] = d->phase_state_q[k]; }
    const double inc = (double)d->phase_increment;
    mix_or_die(entry, in_out, in_out, n, 4, S, 4.0 * inc / TWO_PI_D);
    const double adv = reduced(inc, (double)n);
    for (int k = 0; k < 4; ++k) {
        double o[2]; rot(S[k], adv, o);
        const double m = std::hypot(o[0], o[1]);      // the reference renormalises every 128 samples (:600-612)
        d->phase_state_i[k] = (float)(o[0] / m); d->phase_state_q[k] = (float)(o[1] / m);
    }
}

// The float phase accumulator of the reference's table builders: one increment, then wrapped into (-pi, pi] by whole turns.  The tables
// must carry the reference's float rounding (they are compared bit for bit), so the accumulation stays in float, step by step.
static inline float phase_step(float phase, float inc) {
    phase += inc;
    for (; phase > PI_F; phase -= 2 * PI_F) {}
    for (; phase < -PI_F; phase += 2 * PI_F) {}
    return phase;
}

template <typename D>
static void lanes4_init_state(D* out, float relative_freq, float phase_start_rad) {   // :519-557 (and :637, :750)
    out->phase_increment = 2 * relative_freq * PI_F;
    out->dcos_blk = 0.0F; out->dsin_blk = 0.0F;
    float ph = phase_start_rad;                       // lane i starts i increments ahead
    for (int lane = 0; lane < 4; ++lane, ph = phase_step(ph, out->phase_increment)) {
        out->phase_state_i[lane] = cosf(ph);
        out->phase_state_q[lane] = sinf(ph);
    }
}
// table entry g (g = 0..32) of F/G/H: phasor of 4*(g+1) increments, the phase accumulated in float as there
template <typename F>
static void lanes4_tables(float inc, F&& put) {
    constexpr int LANES = PF_SHIFT_LIMITED_SIMD_SZ, ENTRIES = (PF_SHIFT_LIMITED_UNROLL_SIZE + LANES) / LANES;
    float ph = 0.0F;
    for (int g = 0; g < ENTRIES; ++g) {
        for (int k = 0; k < LANES; ++k) ph = phase_step(ph, inc);
        put(g, cosf(ph), sinf(ph));
    }
}

}  // namespace pd

using namespace pd;

// ================================================================================================
// C ABI — PART 1
// ================================================================================================
PD_EXPORT int have_sse_shift_mixer_impl(void) { return 1; }

// ---- A ----
PD_EXPORT float shift_math_cc(const complexf* input, complexf* output, int input_size, float rate, float starting_phase) {
    rate *= 2;
    const float inc = rate * PI_F;                       // :146-147
    double S[1][2]; unit((double)starting_phase, S[0]);
    mix_or_die("shift_math_cc", input, output, input_size, 1, S, (double)inc / TWO_PI_D);
    if (input_size <= 0) return starting_phase;
    return wrap_0_2pi((double)starting_phase + (double)inc * (double)input_size);
}

// ---- B ----
PD_EXPORT shift_table_data_t shift_table_init(int table_size) {     // :171-181 (the table is kept for callers that read it)
    shift_table_data_t o;
    o.table = (float*)malloc(sizeof(float) * (table_size > 0 ? table_size : 1));
    o.table_size = table_size;
    for (int i = 0; i < table_size; ++i) o.table[i] = sinf(((float)i / table_size) * (PI_F / 2));
    return o;
}
PD_EXPORT void shift_table_deinit(shift_table_data_t table_data) { free(table_data.table); }
PD_EXPORT float shift_table_cc(complexf* input, complexf* output, int input_size, float rate, shift_table_data_t,
                               float starting_phase) {
    rate *= 2;
    const float inc = rate * PI_F;                       // :192-195
    double S[1][2]; unit((double)starting_phase, S[0]);
    mix_or_die("shift_table_cc", input, output, input_size, 1, S, (double)inc / TWO_PI_D);
    if (input_size <= 0) return starting_phase;
    return wrap_0_2pi((double)starting_phase + (double)inc * (double)input_size);
}

// ---- C ----
PD_EXPORT shift_addfast_data_t shift_addfast_init(float rate) {     // :232-242
    shift_addfast_data_t o;
    o.phase_increment = 2 * rate * PI_F;
    for (int i = 0; i < 4; ++i) {
        o.dsin[i] = sinf(o.phase_increment * (i + 1));
        o.dcos[i] = cosf(o.phase_increment * (i + 1));
    }
    return o;
}
static float addfast_run(const char* entry, const complexf* in, complexf* out, int n, shift_addfast_data_t* d, float ph) {
    const long m = (long)(n / 4) * 4;                    // the reference processes n/4 groups of 4 (:264)
    const double inc = (double)d->phase_increment;
    double S[1][2]; unit((double)ph + inc, S[0]);        // sample 0 is already rotated by one increment (:266-270)
    mix_or_die(entry, in, out, m, 1, S, inc / TWO_PI_D);
    return wrap_pm_pi((double)ph + (double)n * inc);     // :281-284
}
PD_EXPORT float shift_addfast_cc(complexf* input, complexf* output, int input_size, shift_addfast_data_t* d, float starting_phase) {
    return addfast_run("shift_addfast_cc", input, output, input_size, d, starting_phase);
}
PD_EXPORT float shift_addfast_inp_c(complexf* in_out, int N_cplx, shift_addfast_data_t* d, float starting_phase) {
    return addfast_run("shift_addfast_inp_c", in_out, in_out, N_cplx, d, starting_phase);
}

// ---- D ----
// Tables of algorithms D and E: entry i = (cos, sin) of the phase accumulator after i + 1 additions of the increment.
// The accumulator is a FLOAT that is brought back into [-pi, pi] after every addition (src/pf_mixer.cpp:341-347,
// :419-425), so entry i is not cos((i + 1) * inc): the rounding of every partial sum is part of the table, and the
// structs have to come out bit-identical to the reference's (tests/test_pfdsp.py).  One builder for both.
static void wrapped_accumulator_table(float inc, int count, float* tcos, float* tsin) {
    const float two_pi = 2 * PI_F;
    float acc = 0.0f;
    for (int i = 0; i < count; ++i) {
        acc += inc;
        for (; acc > PI_F; acc -= two_pi) {}
        for (; acc < -PI_F; acc += two_pi) {}
        tcos[i] = cosf(acc);
        tsin[i] = sinf(acc);
    }
}

PD_EXPORT shift_unroll_data_t shift_unroll_init(float rate, int size) {   // :333-350
    shift_unroll_data_t o;
    const size_t cap = sizeof(float) * (size_t)(size > 0 ? size : 1);
    o.size = size;
    o.phase_increment = 2 * rate * PI_F;
    o.dcos = (float*)malloc(cap);
    o.dsin = (float*)malloc(cap);
    wrapped_accumulator_table(o.phase_increment, size, o.dcos, o.dsin);
    return o;
}
PD_EXPORT void shift_unroll_deinit(shift_unroll_data_t* d) {
    if (!d) return;
    free(d->dsin); free(d->dcos);
    d->dsin = nullptr; d->dcos = nullptr;
}
static float unroll_run(const char* entry, const complexf* in, complexf* out, int n, shift_unroll_data_t* d, float ph) {
    const double inc = (double)d->phase_increment;
    double S[1][2]; unit((double)ph, S[0]);
    mix_or_die(entry, in, out, n, 1, S, inc / TWO_PI_D);
    return wrap_pm_pi((double)ph + (double)n * inc);     // :377-380
}
PD_EXPORT float shift_unroll_cc(complexf* input, complexf* output, int size, shift_unroll_data_t* d, float starting_phase) {
    return unroll_run("shift_unroll_cc", input, output, size, d, starting_phase);
}
PD_EXPORT float shift_unroll_inp_c(complexf* in_out, int size, shift_unroll_data_t* d, float starting_phase) {
    return unroll_run("shift_unroll_inp_c", in_out, in_out, size, d, starting_phase);
}

// ---- E ----
PD_EXPORT shift_limited_unroll_data_t shift_limited_unroll_init(float rate) {   // :413-429
    shift_limited_unroll_data_t o;
    o.phase_increment = 2 * rate * PI_F;
    wrapped_accumulator_table(o.phase_increment, PF_SHIFT_LIMITED_UNROLL_SIZE, o.dcos, o.dsin);
    o.complex_phase.i = 1.0F;   // unit phasor: the state limited_run advances block by block
    o.complex_phase.q = 0.0F;
    return o;
}
static void limited_run(const char* entry, const complexf* in, complexf* out, int n, shift_limited_unroll_data_t* d) {
    if (n <= 0) return;
    const double inc = (double)d->phase_increment;
    double S[1][2] = {{(double)d->complex_phase.i, (double)d->complex_phase.q}};
    mix_or_die(entry, in, out, n, 1, S, inc / TWO_PI_D);
    double o[2]; rot(S[0], reduced(inc, (double)n), o);
    const double m = std::hypot(o[0], o[1]);             // "starts := vals / |vals|" after every block (:452-456)
    d->complex_phase.i = (float)(o[0] / m);
    d->complex_phase.q = (float)(o[1] / m);
}
PD_EXPORT void shift_limited_unroll_cc(const complexf* input, complexf* output, int size, shift_limited_unroll_data_t* d) {
    limited_run("shift_limited_unroll_cc", input, output, size, d);
}
PD_EXPORT void shift_limited_unroll_inp_c(complexf* in_out, int size, shift_limited_unroll_data_t* d) {
    limited_run("shift_limited_unroll_inp_c", in_out, in_out, size, d);
}

// ---- F, G, H ----
PD_EXPORT shift_limited_unroll_A_sse_data_t shift_limited_unroll_A_sse_init(float relative_freq, float phase_start_rad) {
    shift_limited_unroll_A_sse_data_t o;
    lanes4_init_state(&o, relative_freq, phase_start_rad);
    lanes4_tables(o.phase_increment, [&](int g, float c, float s) {       // :527-541: 4 copies of cos, 4 of sin
        for (int k = 0; k < 4; ++k) { o.dcos[4 * g + k] = c; o.dsin[4 * g + k] = s; }
    });
    return o;
}
PD_EXPORT void shift_limited_unroll_A_sse_inp_c(complexf* in_out, int N_cplx, shift_limited_unroll_A_sse_data_t* d) {
    lanes4_run("shift_limited_unroll_A_sse_inp_c", in_out, N_cplx, d);
}
PD_EXPORT shift_limited_unroll_B_sse_data_t shift_limited_unroll_B_sse_init(float relative_freq, float phase_start_rad) {
    shift_limited_unroll_B_sse_data_t o;
    lanes4_init_state(&o, relative_freq, phase_start_rad);
    lanes4_tables(o.phase_increment, [&](int g, float c, float s) {       // :645-657: cos, sin, cos, sin
        o.dtrig[4 * g + 0] = c; o.dtrig[4 * g + 1] = s; o.dtrig[4 * g + 2] = c; o.dtrig[4 * g + 3] = s;
    });
    return o;
}
PD_EXPORT void shift_limited_unroll_B_sse_inp_c(complexf* in_out, int N_cplx, shift_limited_unroll_B_sse_data_t* d) {
    lanes4_run("shift_limited_unroll_B_sse_inp_c", in_out, N_cplx, d);
}
PD_EXPORT shift_limited_unroll_C_sse_data_t shift_limited_unroll_C_sse_init(float relative_freq, float phase_start_rad) {
    shift_limited_unroll_C_sse_data_t o;
    lanes4_init_state(&o, relative_freq, phase_start_rad);
    lanes4_tables(o.phase_increment, [&](int g, float c, float s) {       // :758-773: 4 cos then 4 sin, interleaved
        for (int k = 0; k < 4; ++k) { o.dinterl_trig[8 * g + k] = c; o.dinterl_trig[8 * g + 4 + k] = s; }
    });
    return o;
}
PD_EXPORT void shift_limited_unroll_C_sse_inp_c(complexf* in_out, int N_cplx, shift_limited_unroll_C_sse_data_t* d) {
    lanes4_run("shift_limited_unroll_C_sse_inp_c", in_out, N_cplx, d);
}

// ---- I ----
PD_EXPORT void shift_recursive_osc_update_rate(float rate, shift_recursive_osc_conf_t* conf, shift_recursive_osc_t* state) {
    osc_update_rate<PF_SHIFT_RECURSIVE_SIMD_SZ>(rate, conf, state);
}
PD_EXPORT void shift_recursive_osc_init(float rate, float starting_phase, shift_recursive_osc_conf_t* conf, shift_recursive_osc_t* state) {
    osc_init<PF_SHIFT_RECURSIVE_SIMD_SZ>(rate, starting_phase, conf, state);
}
PD_EXPORT void shift_recursive_osc_cc(const complexf* input, complexf* output, int size, const shift_recursive_osc_conf_t* conf,
                                      shift_recursive_osc_t* state) {
    osc_run<PF_SHIFT_RECURSIVE_SIMD_SZ>("shift_recursive_osc_cc", input, output, size, conf, state, false);
}
PD_EXPORT void shift_recursive_osc_inp_c(complexf* in_out, int size, const shift_recursive_osc_conf_t* conf, shift_recursive_osc_t* state) {
    osc_run<PF_SHIFT_RECURSIVE_SIMD_SZ>("shift_recursive_osc_inp_c", in_out, in_out, size, conf, state, false);
}
PD_EXPORT void gen_recursive_osc_c(complexf* output, int size, const shift_recursive_osc_conf_t* conf, shift_recursive_osc_t* state) {
    osc_run<PF_SHIFT_RECURSIVE_SIMD_SZ>("gen_recursive_osc_c", nullptr, output, size, conf, state, true);
}

// ---- J ----
PD_EXPORT void shift_recursive_osc_sse_update_rate(float rate, shift_recursive_osc_sse_conf_t* conf, shift_recursive_osc_sse_t* state) {
    osc_update_rate<PF_SHIFT_RECURSIVE_SIMD_SSE_SZ>(rate, conf, state);
}
PD_EXPORT void shift_recursive_osc_sse_init(float rate, float starting_phase, shift_recursive_osc_sse_conf_t* conf, shift_recursive_osc_sse_t* state) {
    osc_init<PF_SHIFT_RECURSIVE_SIMD_SSE_SZ>(rate, starting_phase, conf, state);
}
PD_EXPORT void shift_recursive_osc_sse_inp_c(complexf* in_out, int N_cplx, const shift_recursive_osc_sse_conf_t* conf,
                                             shift_recursive_osc_sse_t* state_ext) {
    osc_run<PF_SHIFT_RECURSIVE_SIMD_SSE_SZ>("shift_recursive_osc_sse_inp_c", in_out, in_out, N_cplx, conf, state_ext, false);
}

// ================================================================================================
// C ABI — PART 2
// ================================================================================================
PD_EXPORT int pfdsp_hip_shift_device(const complexf* d_in, complexf* d_out, size_t n_cplx, double rate, double phase_rad, void* stream) {
    double S[1][2]; unit(phase_rad, S[0]);
    int rc = launch_mix(reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out), n_cplx, 1, S, rate,
                        d_in == nullptr, (hipStream_t)stream);
    if (rc) pd::g_last_error = pfmix::last_error;
    return rc;
}
PD_EXPORT const char* pfdsp_hip_last_error(void) { return pd::g_last_error.c_str(); }
PD_EXPORT unsigned pfdsp_hip_error_count(void) { return pd::g_error_count.load(); }
