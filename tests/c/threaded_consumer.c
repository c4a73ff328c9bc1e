/* The reference's calling pattern seen from a host WITHOUT a global interpreter lock — the Julia
 * shim under Threads.@spawn (reference src/forecasting.jl:131-159), a C++ service: T threads, each
 * with the P particles of its own scenario clone on the same dates, each making K gradient calls one
 * after another (the leapfrog steps of its HMC moves, src/forecasting.jl:145-148).  Times the same
 * work with combining off (every call waits for the context and runs alone) and on (include/ngp.h
 * "concurrent callers"), checks that every thread gets — to rounding: batch size decides launch
 * shapes — what it gets alone, and checks thread 0's first item against the C oracle.
 * Built and run by tests/test_combine_gpu.py::test_threaded_c_host; prints one line per size that
 * the test parses.  usage: threaded_consumer T K n P [combining mode: 1 (default) | 2 = never wait for company] */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/ngp.h"
#include "../../oracle/ngp_oracle.h"

typedef struct {
    ngp_ctx *ctx;
    pthread_barrier_t *gate;
    int id, K, n, P, resident;
    const double *t;
    double *y;          /* [n] this scenario's observations */
    ngp_kernel *ks;     /* [P] */
    double *lm, *grad;  /* results of the LAST call */
    int32_t *info;
    int ngrad, failed;
} task_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *work(void *arg) {
    task_t *w = (task_t *)arg;
    pthread_barrier_wait(w->gate);
    for (int k = 0; k < w->K; ++k) {
        ngp_status st = ngp_logml_grad_batch(w->ctx, w->P, w->ks, w->n, w->t, w->y, 0, w->lm, w->grad, w->info);
        if (st != NGP_OK) { w->failed = 1; printf("thread %d call %d: %s\n", w->id, k, ngp_strerror(st)); break; }
    }
    return NULL;
}

static double run_all(task_t *tk, int T) {
    pthread_t th[64];
    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)T + 1);
    for (int i = 0; i < T; ++i) { tk[i].gate = &gate; pthread_create(&th[i], NULL, work, &tk[i]); }
    pthread_barrier_wait(&gate);
    const double t0 = now_s();
    for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
    const double dt = now_s() - t0;
    pthread_barrier_destroy(&gate);
    return dt;
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, K = argc > 2 ? atoi(argv[2]) : 20,
              n = argc > 3 ? atoi(argv[3]) : 208, P = argc > 4 ? atoi(argv[4]) : 24,
              mode = argc > 5 ? atoi(argv[5]) : 1;   /* ngp_set_combining mode of the "together" run */
    if (T < 1 || T > 64 || K < 1 || n < 2 || P < 1) return 2;
    ngp_ctx *ctx = NULL;
    if (ngp_ctx_create(0, &ctx) != NGP_OK) { printf("no device\n"); return 2; }
    /* four tree shapes, cycled over the particles; parameters differ per (thread, particle) */
    static int32_t ops_a[] = {2, 5, 6}, ops_b[] = {4, 3, 8}, ops_c[] = {3}, ops_d[] = {5};
    const int32_t *shapes[4] = {ops_a, ops_b, ops_c, ops_d};
    const int nops[4] = {3, 3, 1, 1}, npar[4] = {6, 7, 2, 3};
    const double base_a[] = {0.2, 0.1, 0.5, 0.9, 0.3, 0.7}, base_b[] = {0.4, 1.3, 0.9, 0.2, 0.7, 0.5, 0.1},
                 base_c[] = {0.3, 0.8}, base_d[] = {0.9, 0.25, 0.6};
    const double *bases[4] = {base_a, base_b, base_c, base_d};
    double *t = malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) t[i] = (double)i / (n - 1);
    task_t *tk = calloc((size_t)T, sizeof(task_t));
    for (int i = 0; i < T; ++i) {
        task_t *w = &tk[i];
        w->ctx = ctx; w->id = i; w->K = K; w->n = n; w->P = P; w->t = t;
        w->y = malloc(sizeof(double) * (size_t)n);
        for (int j = 0; j < n; ++j)
            w->y[j] = sin(9.0 * t[j]) + 0.3 * cos(31.0 * t[j]) + 0.1 * (double)((j * 7919 + i * 13) % 13 - 6) / 6.0;
        w->ks = calloc((size_t)P, sizeof(ngp_kernel));
        w->ngrad = 0;
        for (int p = 0; p < P; ++p) {
            const int s = p % 4;
            double *par = malloc(sizeof(double) * (size_t)npar[s]);
            for (int q = 0; q < npar[s]; ++q) par[q] = bases[s][q] * (1.0 + 0.01 * i + 0.003 * p);
            w->ks[p].n_ops = nops[s]; w->ks[p].n_params = npar[s]; w->ks[p].ops = shapes[s];
            w->ks[p].params = par; w->ks[p].noise = 0.02 + 0.001 * p;
            w->ngrad += npar[s] + 1;
        }
        w->lm = malloc(sizeof(double) * (size_t)P);
        w->grad = malloc(sizeof(double) * (size_t)w->ngrad);
        w->info = malloc(sizeof(int32_t) * (size_t)P);
    }
    int fails = 0;
    /* alone: combining off */
    ngp_set_combining(ctx, 0);
    (void)run_all(tk, T);                             /* warm-up */
    const double off_s = run_all(tk, T);
    double **ref_lm = malloc(sizeof(double *) * (size_t)T), **ref_g = malloc(sizeof(double *) * (size_t)T);
    for (int i = 0; i < T; ++i) {
        ref_lm[i] = malloc(sizeof(double) * (size_t)P);
        ref_g[i] = malloc(sizeof(double) * (size_t)tk[i].ngrad);
        memcpy(ref_lm[i], tk[i].lm, sizeof(double) * (size_t)P);
        memcpy(ref_g[i], tk[i].grad, sizeof(double) * (size_t)tk[i].ngrad);
        for (int p = 0; p < P; ++p) if (tk[i].info[p] != 0) ++fails;
    }
    /* together */
    ngp_set_combining(ctx, mode);
    (void)run_all(tk, T);
    int64_t st4[6];
    ngp_combine_stats(ctx, st4, 1);
    const double on_s = run_all(tk, T);
    ngp_combine_stats(ctx, st4, 1);
    double worst_lm = 0.0, worst_g = 0.0;
    for (int i = 0; i < T; ++i) {
        if (tk[i].failed) ++fails;
        for (int p = 0; p < P; ++p) {
            worst_lm = fmax(worst_lm, fabs(tk[i].lm[p] - ref_lm[i][p]) / fabs(ref_lm[i][p]));
            if (tk[i].info[p] != 0) ++fails;
        }
        int off = 0;
        for (int p = 0; p < P; ++p) {   /* per item, relative to its largest component */
            double num = 0.0, den = 0.0;
            for (int q = 0; q <= tk[i].ks[p].n_params; ++q) {
                num = fmax(num, fabs(tk[i].grad[off + q] - ref_g[i][off + q]));
                den = fmax(den, fabs(ref_g[i][off + q]));
            }
            worst_g = fmax(worst_g, num / fmax(den, 1e-300));
            off += tk[i].ks[p].n_params + 1;
        }
    }
    /* thread 0, item 0 against the C oracle */
    ngp_spec spec;
    ngp_default_spec(&spec);
    /* (the scalar oracle's forward-mode gradient is O(n^3) per parameter: above a thousand points the
     * check is left to the Python tests of that size, which use the LAPACK oracle) */
    double olm = 0.0, og[16], oerr = 0.0, gnum = 0.0, gden = 1.0;
    if (n <= 1200) {
        const int oinfo = ngpo_logml_grad(&spec, &tk[0].ks[0], n, t, tk[0].y, &olm, og);
        oerr = fabs(tk[0].lm[0] - olm) / fabs(olm);
        gden = 0.0;
        for (int q = 0; q <= tk[0].ks[0].n_params; ++q) {
            gnum = fmax(gnum, fabs(tk[0].grad[q] - og[q]));
            gden = fmax(gden, fabs(og[q]));
        }
        if (oinfo != 0) ++fails;
    }
    printf("threaded_consumer T=%d K=%d n=%d P=%d alone_s=%.6f together_s=%.6f ratio=%.3f requests=%lld "
           "sequences=%lld largest_group=%lld shared=%lld worst_logml_diff=%.3e worst_grad_diff=%.3e "
           "oracle_logml_err=%.3e oracle_grad_err=%.3e fails=%d\n",
           T, K, n, P, off_s, on_s, on_s / off_s, (long long)st4[0], (long long)st4[1], (long long)st4[2],
           (long long)st4[3], worst_lm, worst_g, oerr, gnum / fmax(gden, 1e-300), fails);
    ngp_ctx_destroy(ctx);
    return fails ? 1 : 0;
}
