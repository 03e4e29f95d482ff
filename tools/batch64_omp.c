/* batch64_omp.c -- BASELINE config 2 as written, from C: the reference's loop shape (#pragma omp parallel for schedule(dynamic), scrappie_raw.c:355,387) with a body
 * that hands 64 reads at a time to scrappie_hip_basecall_batch on ONE engine.  bench.py builds and runs it (Python threads reach the queue over milliseconds; OpenMP
 * threads arrive together, as a maintainer's loop would).
 *     batch64_omp MODEL.scrm SIGNALS.f32 NREADS NSAMPLES NTHREADS [REPS]
 * SIGNALS.f32: NREADS x NSAMPLES normalised float32 samples.  Prints one line per repetition: "threads T calls C reads R wall_s W samples_per_s S engine_calls E". */
#include "scrappie_hip.h"
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: batch64_omp MODEL.scrm SIGNALS.f32 NREADS NSAMPLES NTHREADS [REPS]\n"); return 2; }
    const size_t n = (size_t)atol(argv[3]), ns = (size_t)atol(argv[4]);
    const int nthr = atoi(argv[5]), reps = argc > 6 ? atoi(argv[6]) : 3;
    const size_t per = 64, ncall = n / per;
    float *sig = malloc(n * ns * sizeof(float));
    FILE *fh = fopen(argv[2], "rb");
    if (!sig || !fh || fread(sig, sizeof(float), n * ns, fh) != n * ns) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    fclose(fh);
    scrappie_hip_engine *e = scrappie_hip_engine_create(0);
    if (!e) { fprintf(stderr, "%s\n", scrappie_hip_last_error()); return 1; }
    const int model = scrappie_hip_load_model(e, "rgrgr_r94", argv[1]);
    if (model < 0) { fprintf(stderr, "%s\n", scrappie_hip_last_error()); return 1; }
    const scrappie_hip_params p = scrappie_hip_default_params();
    raw_table *rts = calloc(n, sizeof *rts);
    scrappie_hip_call *out = calloc(n, sizeof *out);
    for (size_t i = 0; i < n; i++) rts[i] = (raw_table){ NULL, ns, 0, ns, sig + i * ns };
    omp_set_num_threads(nthr);
    double *tr = getenv("BATCH64_TRACE") ? calloc(2 * ncall, sizeof(double)) : NULL;
    if (tr) scrappie_hip_set_profiling(e, 1);
    for (int rep = 0; rep < reps + 1; rep++) {          /* (the first repetition warms the arenas and is not printed) */
        unsigned long long s0[3], s1[3];
        scrappie_hip_batch_coalescer_stats(s0);
        int failed = 0;
        const double t0 = now_s();
#pragma omp parallel for schedule(dynamic)
        for (size_t k = 0; k < ncall; k++) {
            if (tr) tr[2 * k] = now_s() - t0;
            if (scrappie_hip_basecall_batch(e, model, rts + k * per, per, &p, out + k * per) != 0) {
#pragma omp atomic write
                failed = 1;
            }
            if (tr) tr[2 * k + 1] = now_s() - t0;
        }
        const double dt = now_s() - t0;
        if (tr && rep) {      /* BATCH64_TRACE=1: when each caller went in and came out (ms), in order of return */
            for (size_t a = 0; a < ncall; a++) for (size_t b = a + 1; b < ncall; b++) if (tr[2 * b + 1] < tr[2 * a + 1]) {
                double x = tr[2 * a]; tr[2 * a] = tr[2 * b]; tr[2 * b] = x; x = tr[2 * a + 1]; tr[2 * a + 1] = tr[2 * b + 1]; tr[2 * b + 1] = x; }
            for (size_t k = 0; k < ncall; k++) fprintf(stderr, "%s%.2f>%.2f", k % 8 ? "  " : "\n  ", 1e3 * tr[2 * k], 1e3 * tr[2 * k + 1]);
            fprintf(stderr, "\n");
        }
        scrappie_hip_batch_coalescer_stats(s1);
        if (failed) { fprintf(stderr, "%s\n", scrappie_hip_last_error()); return 1; }
        scrappie_hip_free_calls(out, ncall * per);
        if (tr && rep) {
            scrappie_hip_timing tm;
            if (scrappie_hip_get_timing(e, &tm) == 0)
                fprintf(stderr, "last launch group on the device: conv %.2f gru %.2f (%d launches) decode %.2f walk %.2f stitch %.2f total %.2f ms\n", tm.conv_ms, tm.gru_ms, tm.n_gru_launches, tm.decode_ms, tm.backtrace_ms, tm.stitch_ms, tm.total_ms);
        }
        if (rep) printf("threads %d calls %zu reads %zu wall_s %.6f samples_per_s %.6e engine_calls %llu\n", nthr, ncall, ncall * per, dt, (double)(ncall * per * ns) / dt, s1[0] - s0[0]);
    }
    scrappie_hip_engine_destroy(e);
    free(rts); free(out); free(sig);
    return 0;
}
