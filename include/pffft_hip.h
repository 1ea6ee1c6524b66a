/* pffft_hip.h — C ABI of libpffft_hip.so, the MI355X (gfx950) drop-in for the pffft hot path.
 *
 * PART 1 declares, with identical names, argument meaning and error behaviour, every symbol the
 * reference exports for this path (reference headers: include/pffft/pffft.h:124-250,
 * include/pffft/pffft_double.h:124-246, include/pffft/pffastconv.h:145-180, plus the two
 * validate_* self-test entries the reference's test programs declare by hand,
 * tests/test_pffft.c:269-272).  A program compiled against the reference's own headers links
 * against libpffft_hip.so unchanged; this header exists so that the ABI is written down in this
 * repository (tests/test_abi.py checks that every name below is exported).
 *
 * PART 2 is the additive batched / device-pointer extension the throughput metric is measured on
 * (the reference API transforms one vector per call, include/pffft/pffft.h:159,168).
 *
 * Pointer rule for PART 1: `float*` / `double*` arguments may be ordinary host pointers (staged
 * through the device: correct, PCIe-bound) or device / managed pointers (used in place).  There is
 * no CPU arithmetic path in this library: without a usable HIP device (or on any HIP error) a
 * transform entry FAILS SOFT — one line on stderr, the output vector filled with NaN, the failure
 * counted in pffft_hip_error_count() and described by pffft_hip_last_error(); PFFFT_HIP_ABORT=1 in
 * the environment turns that into abort().  A call on an invalid handle (NULL, destroyed, wrong precision) writes
 * nothing.  `work` is accepted and ignored (reference: scratch of N / 2N scalars or NULL, include/pffft/pffft.h:137-142).
 *
 * COST OF THE LEGACY SINGLE-VECTOR ENTRIES.  One call = one kernel launch + one stream synchronisation: 17-29 us per call
 * with host pointers (N = 64 ... 16384; the reference's SSE path: 0.1-30 us), i.e. a program that links this library and
 * keeps calling pffft_transform() vector by vector runs 10-20 x SLOWER than on the CPU for N <= 4096.  The throughput of
 * this library is in PART 2 (batched entries on device-resident data): move the loop over vectors into `batch`.
 *
 * ACCURACY OF pffftd_* AGAINST THE REFERENCE.  The reference's double build keeps float-suffixed radix-3 / radix-5
 * constants (src/pffft_priv_impl.h:154,259-262,389,431-432,636-639 survive `#define float double`), so for every N with a
 * factor 3 or 5 the reference's own pffftd_* result is only ~1e-8 accurate.  This library uses full-precision constants:
 * for those sizes it agrees with a float64 DFT to 1e-12 and with the reference to ~1e-8 (tests: 2e-7); power-of-two sizes
 * agree with the reference to 1e-15.
 */
#ifndef PFFFT_HIP_H
#define PFFFT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PART 1: reference ABI ---- */

typedef struct PFFFT_Setup PFFFT_Setup;   /* include/pffft/pffft.h:106        */
typedef struct PFFFTD_Setup PFFFTD_Setup; /* include/pffft/pffft_double.h:111 */
typedef struct PFFASTCONV_Setup PFFASTCONV_Setup; /* include/pffft/pffastconv.h:81 */

#ifndef PFFFT_COMMON_ENUMS
#define PFFFT_COMMON_ENUMS
typedef enum { PFFFT_FORWARD, PFFFT_BACKWARD } pffft_direction_t; /* pffft.h:112 */
typedef enum { PFFFT_REAL, PFFFT_COMPLEX } pffft_transform_t;     /* pffft.h:115 */
#endif

/* float — src/pffft.c:101-129 maps these onto src/pffft_priv_impl.h */
PFFFT_Setup *pffft_new_setup(int N, pffft_transform_t transform);            /* impl :1062-1112; NULL on invalid N */
void pffft_destroy_setup(PFFFT_Setup *);                                     /* impl :1115-1120; NULL-safe */
void pffft_transform(PFFFT_Setup *, const float *in, float *out, float *work, pffft_direction_t);         /* :1816 */
void pffft_transform_ordered(PFFFT_Setup *, const float *in, float *out, float *work, pffft_direction_t); /* :1820 */
void pffft_zreorder(PFFFT_Setup *, const float *in, float *out, pffft_direction_t);                       /* :1158 */
void pffft_zconvolve_accumulate(PFFFT_Setup *, const float *a, const float *b, float *ab, float scaling); /* :1534 */
void pffft_zconvolve_no_accu(PFFFT_Setup *, const float *a, const float *b, float *ab, float scaling);    /* :1632 */
int pffft_simd_size(void);                                 /* :76  — always 4: selects the 4-lane internal layout */
const char *pffft_simd_arch(void);                         /* :116 — "HIP-gfx950" */
int pffft_min_fft_size(pffft_transform_t transform);       /* :78  */
int pffft_is_valid_size(int N, pffft_transform_t cplx);    /* :91  */
int pffft_nearest_transform_size(int N, pffft_transform_t cplx, int higher); /* :100 */
int pffft_next_power_of_two(int N);                        /* src/pffft_common.c:25 */
int pffft_is_power_of_two(int N);                          /* src/pffft_common.c:39 */
void *pffft_aligned_malloc(size_t nb_bytes);               /* src/pffft_common.c:12 — 64-byte aligned */
void pffft_aligned_free(void *);
int validate_pffft_simd(void);                             /* impl :2227 — layout self-test, 0 = ok */
int validate_pffft_simd_ex(void *dbg_file);                /* impl :1889 (FILE* or NULL) */

/* double — src/pffft_double.c:113-142 */
PFFFTD_Setup *pffftd_new_setup(int N, pffft_transform_t transform);
void pffftd_destroy_setup(PFFFTD_Setup *);
void pffftd_transform(PFFFTD_Setup *, const double *in, double *out, double *work, pffft_direction_t);
void pffftd_transform_ordered(PFFFTD_Setup *, const double *in, double *out, double *work, pffft_direction_t);
void pffftd_zreorder(PFFFTD_Setup *, const double *in, double *out, pffft_direction_t);
void pffftd_zconvolve_accumulate(PFFFTD_Setup *, const double *a, const double *b, double *ab, double scaling);
void pffftd_zconvolve_no_accu(PFFFTD_Setup *, const double *a, const double *b, double *ab, double scaling);
int pffftd_simd_size(void);
const char *pffftd_simd_arch(void);
int pffftd_min_fft_size(pffft_transform_t transform);
int pffftd_is_valid_size(int N, pffft_transform_t cplx);
int pffftd_nearest_transform_size(int N, pffft_transform_t cplx, int higher);
int pffftd_next_power_of_two(int N);
int pffftd_is_power_of_two(int N);
void *pffftd_aligned_malloc(size_t nb_bytes);
void pffftd_aligned_free(void *);
int validate_pffftd_simd(void);
int validate_pffftd_simd_ex(void *dbg_file);

/* fast convolution — src/pffastconv.c:58-263; flag values of include/pffft/pffastconv.h:83-134 */
enum {
  PFFASTCONV_HIP_CPLX_INP_OUT = 1, PFFASTCONV_HIP_CPLX_FILTER = 2, PFFASTCONV_HIP_DIRECT_INP = 4,
  PFFASTCONV_HIP_DIRECT_OUT = 8, PFFASTCONV_HIP_CPLX_SINGLE_FFT = 16, PFFASTCONV_HIP_SYMMETRIC = 32,
  PFFASTCONV_HIP_CORRELATION = 64
};
PFFASTCONV_Setup *pffastconv_new_setup(const float *filterCoeffs, int filterLen, int *blockLen, int flags);
void pffastconv_destroy_setup(PFFASTCONV_Setup *);
/* Returns the number of output samples produced, as the reference does (src/pffastconv.c:133-263).  ADDITION: -1 when the
 * HIP path failed (no device, HIP error, invalid setup) - the reference cannot fail here, and 0 would be indistinguishable
 * from "not enough input yet" for a streaming caller.  On failure at most inputLen - filterLen + 1 samples of `output`
 * (what include/pffft/pffastconv.h:159 guarantees to be writable) are filled with NaN. */
int pffastconv_apply(PFFASTCONV_Setup *, const float *input, int inputLen, float *output, int applyFlush);
void *pffastconv_malloc(size_t nb_bytes);
void pffastconv_free(void *);
int pffastconv_simd_size(void);

/* ------------------------------------------------- PART 2: batched / device extension -------- */
/* Concurrency bounds of the batched entries.  (i) ONE SETUP, ANY DEVICE (round 6): a PFFFT_Setup / PFFFTD_Setup is immutable and may
 * be shared by concurrent threads like the reference's (include/pffft/pffft.h:102-105) - also by threads whose current HIP devices
 * differ: the twiddle tables, work counters, per-stream scratch and staging buffers are kept per device (built on a device's first
 * call under the setup's mutex, released by destroy_setup; pffft_hip_setup_devices lists them).  The pointers of a call must be usable on
 * the calling thread's current device.  A PFFASTCONV_Setup - not shareable between threads in the reference either
 * (include/pffft/pffastconv.h:77-80) - holds its filter tables on ONE device and rebuilds them when its user's device changes.
 * (ii) Kernels that
 * pull their work in order draw a {next, done} counter pair from a ring of 4096 pairs per setup and re-arm it
 * when they retire: at most 4096 launches of ONE setup may be in flight at the same time (summed over all
 * streams).  Launches on one stream serialise, so only > 4096 concurrently RUNNING launches could collide.
 * (iii) Scratch of the sizes beyond LDS is kept per stream: the same setup may run on several streams at once.
 *
 * All return 0 on success, otherwise a hipError_t value (pffft_hip_last_error() has the text).
 * `in`, `out`, `a`, `b`, `ab` are DEVICE pointers to `batch` contiguous vectors (N scalars for a
 * real setup, 2N for a complex one), 16-byte (float) / 32-byte (double) aligned.  `stream` is a
 * hipStream_t (NULL = default stream).  Calls are asynchronous with respect to the host.  in == out
 * is allowed for the transforms (include/pffft/pffft.h:157) and all of a/b/ab may alias for
 * zconvolve (:194); zreorder needs in != out (:180).
 *   ordered = 0 -> pffft_transform semantics (spectrum in the internal layout)
 *   ordered = 1 -> pffft_transform_ordered semantics (canonical interleaved spectrum) */
int pffft_hip_transform_batch(PFFFT_Setup *, const float *in, float *out, size_t batch,
                              pffft_direction_t direction, int ordered, void *stream);
int pffft_hip_zreorder_batch(PFFFT_Setup *, const float *in, float *out, size_t batch,
                             pffft_direction_t direction, void *stream);
/* ab[i] (+)= a[i] * b[i or 0] * scaling; b_broadcast != 0 reuses ONE spectrum b for every vector
 * (the FIR case, src/pffastconv.c:238) */
int pffft_hip_zconvolve_batch(PFFFT_Setup *, const float *a, const float *b, float *ab, float scaling,
                              size_t batch, int accumulate, int b_broadcast, void *stream);

int pffftd_hip_transform_batch(PFFFTD_Setup *, const double *in, double *out, size_t batch,
                               pffft_direction_t direction, int ordered, void *stream);
int pffftd_hip_zreorder_batch(PFFFTD_Setup *, const double *in, double *out, size_t batch,
                              pffft_direction_t direction, void *stream);
int pffftd_hip_zconvolve_batch(PFFFTD_Setup *, const double *a, const double *b, double *ab, double scaling,
                               size_t batch, int accumulate, int b_broadcast, void *stream);

/* Batch shards over several devices from ONE host thread (round 5; SURVEY.md 8(e): the batch shards with no exchange step).  Part p -
 * batches[p] vectors at in[p] / out[p], device memory of devices[p] - is transformed by setups[p] on devices[p]: hipSetDevice, then the
 * batched entry on streams[p] (streams == NULL or streams[p] == NULL: that device's default stream).  Every launch is asynchronous, so the
 * devices work concurrently; the caller's current device is restored before the call returns.  setups[p] may be THE SAME SETUP in every
 * slot (round 6: a setup keeps its device state per device, above) or a setup per part.  Returns the first error, 0 when every part is
 * enqueued.  The reference has no counterpart: one of its setups serves any number of threads of one CPU
 * (include/pffft/pffft.h:102-105) - which is what one setup for all devices restores. */
int pffft_hip_transform_batch_multi(int nparts, const int *devices, PFFFT_Setup *const *setups, const float *const *in, float *const *out,
                                    const size_t *batches, pffft_direction_t direction, int ordered, void *const *streams);
int pffftd_hip_transform_batch_multi(int nparts, const int *devices, PFFFTD_Setup *const *setups, const double *const *in, double *const *out,
                                     const size_t *batches, pffft_direction_t direction, int ordered, void *const *streams);

/* Spectral convolution in one call (round 4):   out[i] (+)= backward( forward(in[i]) . H[i or 0] ) * scaling
 * - the sequence pffft_transform(FORWARD), pffft_zconvolve_no_accu, pffft_transform(BACKWARD) of every FFT convolution
 * (src/pffft_priv_impl.h:1465-1532, :1632-1684; src/pffastconv.c:235-254 is one instance), with `in` / `out` in the TIME domain
 * and H a spectrum in the INTERNAL layout (what pffft_transform(…, PFFFT_FORWARD) produced for the filter).  The transforms are
 * unscaled like the reference's: pass scaling = 1/N for a true circular convolution.  accumulate != 0 adds the result to `out`
 * (by linearity the same values as accumulating spectra with pffft_zconvolve_accumulate before one inverse transform).
 * h_broadcast != 0: ONE filter spectrum for the whole batch - for the power-of-two sizes up to N = 8192 complex / 16384 real
 * (float; 4096 / 8192 double) this runs as ONE kernel, one read and one write of every vector instead of seven vector passes;
 * every other case (per-vector spectra, sizes with factors 3 / 5, sizes beyond LDS) is composed from the three batched entries
 * on `stream` through a per-stream scratch image of the batch.  in == out is allowed; H must not alias out. */
int pffft_hip_convolve_batch(PFFFT_Setup *, const float *in, const float *H, float *out, float scaling, size_t batch,
                             int accumulate, int h_broadcast, void *stream);
int pffftd_hip_convolve_batch(PFFFTD_Setup *, const double *in, const double *H, double *out, double scaling, size_t batch,
                              int accumulate, int h_broadcast, void *stream);

/* Frequency shift fused into the forward transform (the mixers themselves:
 * include/pfdsp_hip.h, reference src/pf_mixer.cpp:142-165).  `in` is ONE stream of batch*N interleaved
 * complex samples; sample g is multiplied by exp(j*(phase_rad + 2*pi*rate*g)) — shift_math_cc's
 * function with an exactly reduced phase — and every N consecutive shifted samples are forward-
 * transformed as pffft_hip_transform_batch would (ordered as there).  Complex float setups only.
 * N = 1024 runs as one kernel (the shift costs no HBM traffic); other N as mixer + in-place transform. */
int pffft_hip_shift_transform_batch(PFFFT_Setup *, const float *in, float *out, size_t batch, int ordered,
                                    double rate, double phase_rad, void *stream);

/* Overlap-save FIR on device-resident signal/output (same block schedule as pffastconv_apply,
 * src/pffastconv.c:204-261): returns the number of output samples written, or -1 on error. */
int pffastconv_hip_apply_device(PFFASTCONV_Setup *, const float *d_input, int inputLen, float *d_output,
                                int applyFlush, void *stream);
/* The same filter over `nsignals` independent signals of `inputLen` samples each (complex I/O: complex samples), signal
 * i at d_input + i*inputStride and its output at d_output + i*outputStride (strides in floats; inputStride >= the signal's
 * floats, outputStride >= the floats one signal produces - both checked, -1 otherwise; any nsignals).
 * Every signal is processed exactly as one pffastconv_hip_apply_device call would (src/pffastconv.c:133-263 per signal,
 * same block schedule, same number of outputs — the return value, per signal); all blocks of all signals share one
 * launch so that reference-sized calls (BASELINE configs[3]: 255 blocks) fill the chip.  -1 on error. */
int pffastconv_hip_apply_batch(PFFASTCONV_Setup *, const float *d_input, int inputLen, size_t inputStride,
                               float *d_output, size_t outputStride, int nsignals, int applyFlush, void *stream);

/* Name of the kernel family a setup dispatches to ("c1024_f32", "tiled", "tiny", "stockham", "stockham_rt" = the same kernel on a
 * run-time plan because the size has no generated compile-time plan (no legal size today), "fourstep" = tile / streaming
 * passes beyond LDS): for tests/bench. */
const char *pffft_hip_kernel_name(const void *setup);
/* The routes of a setup as text: one line for the setup and one per (direction, layout) with the kernel family, its configuration,
 * the launch rule (dispatch order / static stride / in-order loop and the bound below which a launch runs in dispatch order) and,
 * beyond LDS, the sweeps over HBM (tile lengths, which pass reads / stores the internal layout).  Decided once, at pffft_new_setup
 * (reference: the ifac[] / twiddle plan of struct PFFFT_Setup, src/pffft_priv_impl.h:1051-1060, is fixed at setup time too).  Writes at
 * most len - 1 characters + a terminating 0 into buf and returns the length of the whole text (snprintf convention), -1 for an
 * invalid handle.  Works for PFFFT_Setup and PFFFTD_Setup handles; no device needed. */
int pffft_hip_describe(const void *setup, char *buf, size_t len);
/* The devices a setup holds tables / counters / scratch on right now: the device it was first used on, then one entry per further
 * device (HIP device indices; values >= 64 belong to the test hook pffft_hip_set_variant(130)).  Fills devices[0 .. max) and returns the
 * count (which may exceed max); 0 for a setup no transform has run on, or an invalid handle.  PFFFT_Setup and PFFFTD_Setup handles. */
int pffft_hip_setup_devices(const void *setup, int *devices, int max);
/* Resident workgroups per CU of the kernel a (direction, layout) of an LDS-resident setup runs on, as the launcher sizes its grid (the
 * runtime's occupancy query for the route's kernel, workgroup size and LDS bytes); 0 for routes without one persistent kernel (beyond
 * LDS, the minimum sizes), -1 on error.  Needs a device.  For tests and tools. */
int pffft_hip_route_occupancy(const void *setup, int direction, int ordered);
/* The tile plan of a complex core transform of n points beyond LDS (n = N for complex setups, N / 2 for real ones): returns the
 * number of tile passes over HBM — 2 or 3 — and their tile lengths in `lengths` (column pass(es) first, the row pass last), or 0
 * when the size runs on the streaming passes (or is LDS-resident: the planner is not consulted then).  `deep` = 1: the size's
 * streaming route would take five sweeps (its row length is itself beyond LDS), which admits costlier tile plans; 0: it takes three
 * and n is a complex transform; 2: it takes three and n is the core of a real transform of 2n points (the plans of 0 minus those
 * that pay only for complex transforms).  Pure host arithmetic, no device needed: for tests and for callers that want to know what
 * a size costs. */
int pffft_hip_tile_plan(long long n, int is_double, int deep, int lengths[3]);
/* The planner's tuning interface (tools/tune_tile_plans.py writes pffft_amd/csrc/tile_plan_gen.h with it).  pffft_hip_tile_candidates: every
 * legal pair of tile lengths of n, out[5 i ..] = {L1, gen1, L2, gen2, cost of the model}, returns the count (may exceed max).
 * pffft_hip_tile_override: from now on setups of n are planned with this pair (l1 > 0; -1 when it is not a legal pair), with no tile plan
 * (l1 == 0: the streaming passes) or by the tables again (l1 < 0).  Process-wide.  Call it while NO setup of n exists and destroy the setups
 * of n before changing it again: a setup's route is fixed at pffft_new_setup for the lengths of that moment, the passes read the lengths
 * per launch. */
int pffft_hip_tile_candidates(long long n, int is_double, int *out, int max);
int pffft_hip_tile_override(long long n, int is_double, int l1, int g1, int l2, int g2);
const char *pffft_hip_last_error(void);
/* Number of legacy (void) entries that failed in this process so far.  The legacy entries have no error channel
 * (include/pffft/pffft.h:159); a failed call — no device, HIP error — prints one line on stderr, fills its output
 * vector with NaN (all-ones bytes) and increments this counter.  PFFFT_HIP_ABORT=1 makes it abort() instead. */
unsigned pffft_hip_error_count(void);
int pffft_hip_device_count(void);
/* 0 = the planner's routes; the other values (enum AbValue, pffft_amd/csrc/pf_route.h) select an ALTERNATIVE ROUTE to the same
 * result - the streaming passes instead of the tile passes beyond LDS, the composed convolution instead of the fused kernel ... -
 * which the parity tests hold to the same bar as the default one; a development build (-DPFFFT_HIP_VARIANTS) knows a few more.  A
 * value the build does not know runs the default route.  The selector is THREAD-LOCAL: it affects only calls made by the thread
 * that set it. */
void pffft_hip_set_variant(int variant);
/* 1 when the library was built with -DPFFFT_HIP_VARIANTS (development build: both variants of every Stockham plan are
 * instantiated for A/B measurements), 0 for the product build (the adopted variant only; a selector that asks for the
 * other one gets the same arithmetic from the kernel that exists). */
int pffft_hip_has_variants(void);

#ifdef __cplusplus
}
#endif
#endif /* PFFFT_HIP_H */
