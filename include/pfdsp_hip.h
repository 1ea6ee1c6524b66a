/* pfdsp_hip.h — C ABI of libpfdsp_hip.so: the PFDSP frequency-shift mixers
 * on MI355X (gfx950).
 *
 * PART 1 declares, with identical names, struct layouts, argument meaning and state/return
 * behaviour, every symbol of the reference's mixer API (include/pffft/pf_mixer.h:61-280,
 * implemented in src/pf_mixer.cpp).  A program compiled against the reference's pf_mixer.h links
 * against libpfdsp_hip.so unchanged.
 *
 * All ten reference algorithms are CPU approximations of ONE function,
 *        out[i] = in[i] * exp(j * (phase0 + i * 2*pi*rate)),
 * that differ in how they avoid calling sinf/cosf per sample (tables, angle-addition recurrences,
 * recursive oscillators with periodic renormalisation) and in where they keep the phase between
 * calls (returned float, complex phasor in the struct, 4 or 8 lane phasors).  A sequential
 * recurrence has no place on a GPU: here every entry runs the same closed-form kernel
 * (pffft_amd/csrc/pfdsp_hip.hip: lane phasor x rotation by an exactly reduced phase), and the host
 * side keeps each algorithm's own contract — which sample gets which phase, what is returned and
 * how the state struct is advanced — so that calls can be chained exactly as with the reference.
 * ACCURACY DIFFERS FROM THE REFERENCE BY DESIGN, and there is no switch to reproduce the reference's
 * drift: every reference algorithm accumulates float rounding along the stream (measured against a
 * float64 oscillator: 1e-5 .. 3e-5 after 256 samples, 1e-4 .. 5e-4 after 4096, 3e-3 .. 8e-3 after
 * 65536 for algorithms A, D-H; C, I, J stay near 1e-5), while this library's per-sample error stays
 * below 1e-6 absolute at any stream position.  Outputs therefore agree with the reference's to within
 * ITS drift bound 1e-6 + 2e-7 n, not bit for bit: a comparison against RECORDED reference output of a long
 * stream must allow that bound (tests/test_pfdsp.py holds both sides against the float64 oscillator).
 *
 * Pointer rule: `complexf*` arguments may be host pointers (staged through the device) or
 * device / managed pointers (used in place).  No CPU arithmetic path: without a usable HIP device (or on a
 * HIP error) a mixer call FAILS SOFT — one line on stderr, NaN-filled output, pfdsp_hip_error_count()
 * incremented; PFFFT_HIP_ABORT=1 aborts instead.
 * The *_init / *_deinit / *_update_rate entries are pure host code and work without a GPU.
 *
 * PART 2 is the additive device/stream entry.  The fused "shift, then FFT" entry lives with the
 * transforms: pffft_hip_shift_transform_batch in include/pffft_hip.h.
 */
#ifndef PFDSP_HIP_H
#define PFDSP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PART 1: reference ABI ---- */

#ifndef PFDSP_HIP_NO_COMPLEXF
typedef struct complexf_s { float i; float q; } complexf;      /* include/pffft/pf_cplx.h:44 */
#endif

int have_sse_shift_mixer_impl(void);   /* pf_mixer.h:61 — 1: the F/G/H/J entries below are functional */

/* A  shift_math_cc (src/pf_mixer.cpp:142-165): sample i gets starting_phase + i*2*pi*rate;
 *    returns the phase of sample input_size wrapped the way the reference wraps it ([0, 2*pi]). */
float shift_math_cc(const complexf *input, complexf *output, int input_size, float rate, float starting_phase);

/* B  shift_table_cc (:171-225): same contract as A.  The reference's lookup index is computed as
 *    (int)(x) * table_size (:202), i.e. always 0 — a quadrant-resolution oscillator; implemented here
 *    is the documented function (a complex mixer), the table argument only carries its size. */
typedef struct shift_table_data_s { float *table; int table_size; } shift_table_data_t;
shift_table_data_t shift_table_init(int table_size);
void shift_table_deinit(shift_table_data_t table_data);
float shift_table_cc(complexf *input, complexf *output, int input_size, float rate,
                     shift_table_data_t table_data, float starting_phase);

/* C  shift_addfast (:232-321): sample i gets starting_phase + (i+1)*phase_increment (the
 *    reference rotates BEFORE multiplying, :241-246,266-278); returns starting_phase + n*inc in [-pi, pi]. */
typedef struct shift_addfast_data_s { float dsin[4]; float dcos[4]; float phase_increment; } shift_addfast_data_t;
shift_addfast_data_t shift_addfast_init(float rate);
float shift_addfast_cc(complexf *input, complexf *output, int input_size, shift_addfast_data_t *d, float starting_phase);
float shift_addfast_inp_c(complexf *in_out, int N_cplx, shift_addfast_data_t *d, float starting_phase);

/* D  shift_unroll (:333-404): sample i gets starting_phase + i*inc; returns as C. */
typedef struct shift_unroll_data_s { float *dsin; float *dcos; float phase_increment; int size; } shift_unroll_data_t;
shift_unroll_data_t shift_unroll_init(float rate, int size);
void shift_unroll_deinit(shift_unroll_data_t *d);
float shift_unroll_cc(complexf *input, complexf *output, int size, shift_unroll_data_t *d, float starting_phase);
float shift_unroll_inp_c(complexf *in_out, int size, shift_unroll_data_t *d, float starting_phase);

/* E  shift_limited_unroll (:413-508): phase kept as the unit phasor d->complex_phase, advanced by
 *    size*inc per call.  size must be a multiple of 4. */
#define PF_SHIFT_LIMITED_UNROLL_SIZE 128
#define PF_SHIFT_LIMITED_SIMD_SZ 4
typedef struct shift_limited_unroll_data_s {
  float dcos[PF_SHIFT_LIMITED_UNROLL_SIZE]; float dsin[PF_SHIFT_LIMITED_UNROLL_SIZE];
  complexf complex_phase; float phase_increment;
} shift_limited_unroll_data_t;
shift_limited_unroll_data_t shift_limited_unroll_init(float rate);
void shift_limited_unroll_cc(const complexf *input, complexf *output, int size, shift_limited_unroll_data_t *d);
void shift_limited_unroll_inp_c(complexf *in_out, int size, shift_limited_unroll_data_t *d);

/* F, G, H  shift_limited_unroll_{A,B,C}_sse (:519-857): four lane phasors phase_state_{i,q}[k] =
 *    exp(j*(phase + k*inc)), all advanced by N_cplx*inc per call.  N_cplx must be a multiple of 4. */
typedef struct shift_limited_unroll_A_sse_data_s {
  float dcos[PF_SHIFT_LIMITED_UNROLL_SIZE + PF_SHIFT_LIMITED_SIMD_SZ];
  float dsin[PF_SHIFT_LIMITED_UNROLL_SIZE + PF_SHIFT_LIMITED_SIMD_SZ];
  float phase_state_i[PF_SHIFT_LIMITED_SIMD_SZ]; float phase_state_q[PF_SHIFT_LIMITED_SIMD_SZ];
  float dcos_blk; float dsin_blk; float phase_increment;
} shift_limited_unroll_A_sse_data_t;
shift_limited_unroll_A_sse_data_t shift_limited_unroll_A_sse_init(float relative_freq, float phase_start_rad);
void shift_limited_unroll_A_sse_inp_c(complexf *in_out, int N_cplx, shift_limited_unroll_A_sse_data_t *d);

typedef struct shift_limited_unroll_B_sse_data_s {
  float dtrig[PF_SHIFT_LIMITED_UNROLL_SIZE + PF_SHIFT_LIMITED_SIMD_SZ];
  float phase_state_i[PF_SHIFT_LIMITED_SIMD_SZ]; float phase_state_q[PF_SHIFT_LIMITED_SIMD_SZ];
  float dcos_blk; float dsin_blk; float phase_increment;
} shift_limited_unroll_B_sse_data_t;
shift_limited_unroll_B_sse_data_t shift_limited_unroll_B_sse_init(float relative_freq, float phase_start_rad);
void shift_limited_unroll_B_sse_inp_c(complexf *in_out, int N_cplx, shift_limited_unroll_B_sse_data_t *d);

typedef struct shift_limited_unroll_C_sse_data_s {
  float dinterl_trig[2 * (PF_SHIFT_LIMITED_UNROLL_SIZE + PF_SHIFT_LIMITED_SIMD_SZ)];
  float phase_state_i[PF_SHIFT_LIMITED_SIMD_SZ]; float phase_state_q[PF_SHIFT_LIMITED_SIMD_SZ];
  float dcos_blk; float dsin_blk; float phase_increment;
} shift_limited_unroll_C_sse_data_t;
shift_limited_unroll_C_sse_data_t shift_limited_unroll_C_sse_init(float relative_freq, float phase_start_rad);
void shift_limited_unroll_C_sse_inp_c(complexf *in_out, int N_cplx, shift_limited_unroll_C_sse_data_t *d);

/* I  recursive quadrature oscillator, 8 lanes (:898-1030).  NOTE the reference's per-sample
 *    increment here is rate*pi, not 2*pi*rate (:901) — kept.  Lane j holds the phasor of sample j;
 *    a call advances every lane by (size/8) block steps of angle 2*atan(conf->k1).
 *    size must be a multiple of 8. */
#define PF_SHIFT_RECURSIVE_SIMD_SZ 8
typedef struct shift_recursive_osc_s { float u_cos[PF_SHIFT_RECURSIVE_SIMD_SZ]; float v_sin[PF_SHIFT_RECURSIVE_SIMD_SZ]; } shift_recursive_osc_t;
typedef struct shift_recursive_osc_conf_s { float k1; float k2; } shift_recursive_osc_conf_t;
void shift_recursive_osc_init(float rate, float starting_phase, shift_recursive_osc_conf_t *conf, shift_recursive_osc_t *state);
void shift_recursive_osc_update_rate(float rate, shift_recursive_osc_conf_t *conf, shift_recursive_osc_t *state);
void shift_recursive_osc_cc(const complexf *input, complexf *output, int size,
                            const shift_recursive_osc_conf_t *conf, shift_recursive_osc_t *state);
void shift_recursive_osc_inp_c(complexf *output, int size, const shift_recursive_osc_conf_t *conf, shift_recursive_osc_t *state);
void gen_recursive_osc_c(complexf *output, int size, const shift_recursive_osc_conf_t *conf, shift_recursive_osc_t *state);

/* J  the same oscillator with 4 lanes (:1043-1126); N_cplx must be a multiple of 4. */
#define PF_SHIFT_RECURSIVE_SIMD_SSE_SZ 4
typedef struct shift_recursive_osc_sse_s { float u_cos[PF_SHIFT_RECURSIVE_SIMD_SSE_SZ]; float v_sin[PF_SHIFT_RECURSIVE_SIMD_SSE_SZ]; } shift_recursive_osc_sse_t;
typedef struct shift_recursive_osc_sse_conf_s { float k1; float k2; } shift_recursive_osc_sse_conf_t;
void shift_recursive_osc_sse_init(float rate, float starting_phase, shift_recursive_osc_sse_conf_t *conf, shift_recursive_osc_sse_t *state);
void shift_recursive_osc_sse_update_rate(float rate, shift_recursive_osc_sse_conf_t *conf, shift_recursive_osc_sse_t *state);
void shift_recursive_osc_sse_inp_c(complexf *in_out, int N_cplx, const shift_recursive_osc_sse_conf_t *conf, shift_recursive_osc_sse_t *state_ext);

/* ------------------------------------------------- PART 2: device / stream extension --------- */
/* out[i] = in[i] * exp(j*(phase_rad + i*2*pi*rate)) for i < n_cplx on DEVICE pointers (8-byte
 * aligned; in == out allowed), asynchronously on `stream` (hipStream_t, NULL = default).  rate and
 * phase are doubles: the phase of sample i is reduced exactly, whatever n_cplx.  in == NULL writes
 * the oscillator itself.  Returns 0 or a hipError_t value (pfdsp_hip_last_error() has the text). */
int pfdsp_hip_shift_device(const complexf *d_in, complexf *d_out, size_t n_cplx, double rate, double phase_rad, void *stream);
const char *pfdsp_hip_last_error(void);
/* number of mixer entries that failed soft in this process (no device / HIP error: stderr line, NaN-filled output) */
unsigned pfdsp_hip_error_count(void);

#ifdef __cplusplus
}
#endif
#endif /* PFDSP_HIP_H */
