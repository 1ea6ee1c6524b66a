/* Timed CPU baseline driver — TEST INFRASTRUCTURE (bench.py "cpu_baseline" leg only).
 *
 * Runs the REAL reference (oracle/_ref/libpffft_ref.so = marton78/pffft compiled from its own
 * sources) the only way its API allows: a loop of single-vector pffft_transform() calls
 * (include/pffft/pffft.h:159 — there is no batch entry), on `threads` host threads that share one
 * read-only PFFFT_Setup and own one `work` buffer each (allowed by include/pffft/pffft.h:102-105).
 * Wall clock (clock_gettime), not the reference bench's user-time clock()
 * (benchmarks/bench_pffft.c:286-287), which would sum over threads.
 *
 * Only prototypes are declared here; nothing is copied from the reference.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct PFFFT_Setup PFFFT_Setup;
extern PFFFT_Setup *pffft_new_setup(int N, int transform);
extern void pffft_destroy_setup(PFFFT_Setup *);
extern void pffft_transform(PFFFT_Setup *, const float *in, float *out, float *work, int direction);
extern void pffft_transform_ordered(PFFFT_Setup *, const float *in, float *out, float *work, int direction);
extern void *pffft_aligned_malloc(size_t);
extern void pffft_aligned_free(void *);
typedef struct PFFFTD_Setup PFFFTD_Setup;
extern PFFFTD_Setup *pffftd_new_setup(int N, int transform);
extern void pffftd_destroy_setup(PFFFTD_Setup *);
extern void pffftd_transform(PFFFTD_Setup *, const double *in, double *out, double *work, int direction);

typedef struct PFFASTCONV_Setup PFFASTCONV_Setup;
extern PFFASTCONV_Setup *pffastconv_new_setup(const float *filterCoeffs, int filterLen, int *blockLen, int flags);
extern void pffastconv_destroy_setup(PFFASTCONV_Setup *);
extern int pffastconv_apply(PFFASTCONV_Setup *, const float *input, int inputLen, float *output, int applyFlush);

typedef struct {
  void *setup; const char *in; char *out; size_t vec_bytes; long first, count; int reps, direction, ordered, is_double;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  void *work = pffft_aligned_malloc(j->vec_bytes);
  for (int r = 0; r < j->reps; ++r)
    for (long i = j->first; i < j->first + j->count; ++i) {
      const char *src = j->in + (size_t)i * j->vec_bytes;
      char *dst = j->out + (size_t)i * j->vec_bytes;
      if (j->is_double) pffftd_transform((PFFFTD_Setup *)j->setup, (const double *)src, (double *)dst, (double *)work, j->direction);
      else if (j->ordered) pffft_transform_ordered((PFFFT_Setup *)j->setup, (const float *)src, (float *)dst, (float *)work, j->direction);
      else pffft_transform((PFFFT_Setup *)j->setup, (const float *)src, (float *)dst, (float *)work, j->direction);
    }
  pffft_aligned_free(work);
  return 0;
}

/* Transforms `batch` vectors (in -> out, both host, 64-byte aligned, contiguous) `reps` times on
 * `threads` threads.  Returns elapsed wall seconds of the timed region, < 0 on error. */
double cpu_baseline_run(int N, int transform, int is_double, int direction, int ordered, const void *in, void *out,
                        long batch, int reps, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  void *setup = is_double ? (void *)pffftd_new_setup(N, transform) : (void *)pffft_new_setup(N, transform);
  if (!setup) return -1.0;
  size_t vec_bytes = (size_t)N * (transform == 1 ? 2 : 1) * (is_double ? 8 : 4);
  pthread_t th[256];
  job_t jobs[256];
  long per = (batch + threads - 1) / threads;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int started = 0;
  for (int t = 0; t < threads; ++t) {
    long first = (long)t * per, cnt = batch - first < per ? batch - first : per;
    if (cnt <= 0) break;
    jobs[t] = (job_t){setup, (const char *)in, (char *)out, vec_bytes, first, cnt, reps, direction, ordered, is_double};
    pthread_create(&th[t], 0, worker, &jobs[t]);
    ++started;
  }
  for (int t = 0; t < started; ++t) pthread_join(th[t], 0);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (is_double) pffftd_destroy_setup((PFFFTD_Setup *)setup); else pffft_destroy_setup((PFFFT_Setup *)setup);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* FIR baseline (BASELINE configs[3]): `reps` calls of pffastconv_apply(setup, x, len, y, 1) over the whole signal on
 * each of `threads` threads.  A PFFASTCONV_Setup is NOT shareable (include/pffft/pffastconv.h:77-80), so every thread
 * owns its setup and its output buffer (y + t*len floats); the signal is shared read-only.  Returns elapsed wall
 * seconds, < 0 on error; *produced = output samples of ONE call. */
typedef struct { const float *h; int taps; const float *x; int len; float *y; int reps; int produced; } firjob_t;

static void *fir_worker(void *arg) {
  firjob_t *j = (firjob_t *)arg;
  int bl = 0;
  PFFASTCONV_Setup *s = pffastconv_new_setup(j->h, j->taps, &bl, 0);
  if (!s) { j->produced = -1; return 0; }
  for (int r = 0; r < j->reps; ++r) j->produced = pffastconv_apply(s, j->x, j->len, j->y, 1);
  pffastconv_destroy_setup(s);
  return 0;
}

double cpu_baseline_fir(const float *h, int taps, const float *x, int len, float *y, int reps, int threads, int *produced) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  firjob_t jobs[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (firjob_t){h, taps, x, len, y + (size_t)t * (size_t)len, reps, 0};
    pthread_create(&th[t], 0, fir_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (produced) *produced = jobs[0].produced;
  for (int t = 0; t < threads; ++t) if (jobs[t].produced < 0) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
