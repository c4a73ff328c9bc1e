/* A consumer of include/ngp.h written in plain C — what a non-Python host (the Julia shim's
 * ccall, a C++ service) sees: create a context, run the batched entry points on a small ensemble,
 * compare every output with the C oracle, exercise the error returns.  Built and run by
 * tests/test_gpu_parity.py::test_plain_c_consumer_of_the_abi; links libngp.so (the product) and
 * libngp_oracle.so (the checker).  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ngp.h"
#include "../../oracle/ngp_oracle.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static double relerr(const double *a, const double *b, int n) {
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i) {
        num = fmax(num, fabs(a[i] - b[i]));
        den = fmax(den, fabs(b[i]));
    }
    return num / fmax(den, 1e-300);
}

int main(void) {
    /* three particles: Linear + Periodic;  ChangePoint(GammaExp, SqExp);  Times(Linear, Periodic) */
    int32_t ops0[] = {2, 5, 6}, ops1[] = {4, 3, 8}, ops2[] = {2, 5, 7};
    double par0[] = {0.2, 0.1, 0.5, 0.9, 0.3, 0.7};
    double par1[] = {0.4, 1.3, 0.9, 0.2, 0.7, 0.5, 0.1};
    double par2[] = {0.1, 0.3, 0.8, 1.1, 0.21, 0.4};
    ngp_kernel ks[3] = {{3, 6, ops0, par0, 0.05}, {3, 7, ops1, par1, 0.02}, {3, 6, ops2, par2, 0.1}};
    enum { N = 150, D = 4, DD = 2, M = 5, P = 3 };
    double t[N], y[N], t_add[DD], y_add[D * DD], t_new[M];
    for (int i = 0; i < N; ++i) {
        t[i] = (double)i / (N - 1);
        y[i] = sin(9.0 * t[i]) + 0.3 * cos(31.0 * t[i]) + 0.1 * (double)((i * 7919) % 13 - 6) / 6.0;
    }
    for (int a = 0; a < DD; ++a) t_add[a] = 1.0 + (double)(a + 1) / (N - 1);
    for (int s = 0; s < D; ++s)
        for (int a = 0; a < DD; ++a) y_add[s * DD + a] = 0.2 * s - 0.1 * a;
    for (int i = 0; i < M; ++i) t_new[i] = t_add[DD - 1] + (double)(i + 1) / (N - 1);

    ngp_ctx *ctx = NULL;
    ngp_status st = ngp_ctx_create(0, &ctx);
    if (st != NGP_OK) {
        printf("ngp_ctx_create: %s\n", ngp_strerror(st));
        return 2;
    }
    printf("libngp %s\n", ngp_version());
    ngp_spec spec;
    ngp_default_spec(&spec);

    /* --- nowcast fan-out: one call, all particles x scenarios --------------------------- */
    double lb[P], lf[P * D], mu[P * D * M], sg[P * M * M];
    int32_t info[P];
    st = ngp_nowcast_batch(ctx, P, ks, N, t, y, DD, t_add, D, y_add, M, t_new, 1, lb, lf, mu, sg, info);
    CHECK(st == NGP_OK, "ngp_nowcast_batch: %s", ngp_strerror(st));
    for (int p = 0; p < P; ++p) {
        double rlb, rlf[D], rmu[D * M], rsg[M * M];
        int oi = ngpo_nowcast(&spec, &ks[p], N, t, y, DD, t_add, D, y_add, M, t_new, 1, &rlb, rlf, rmu, rsg);
        CHECK(oi == 0 && info[p] == 0, "particle %d info %d / oracle %d", p, (int)info[p], oi);
        CHECK(relerr(&lb[p], &rlb, 1) < 1e-9, "logml_base[%d] %.15g vs %.15g", p, lb[p], rlb);
        CHECK(relerr(&lf[p * D], rlf, D) < 1e-9, "logml_full[%d]", p);
        CHECK(relerr(&mu[p * D * M], rmu, D * M) < 1e-7, "mu[%d] err %.3g", p, relerr(&mu[p * D * M], rmu, D * M));
        CHECK(relerr(&sg[p * M * M], rsg, M * M) < 1e-7, "sigma[%d]", p);
    }

    /* --- logml + gradient ---------------------------------------------------------------- */
    double lm[P], grad[7 + 8 + 7];
    st = ngp_logml_grad_batch(ctx, P, ks, N, t, y, 0, lm, grad, info);
    CHECK(st == NGP_OK, "ngp_logml_grad_batch: %s", ngp_strerror(st));
    for (int p = 0, off = 0; p < P; off += ks[p].n_params + 1, ++p) {
        double rl, rg[8];
        ngpo_logml_grad(&spec, &ks[p], N, t, y, &rl, rg);
        CHECK(relerr(&lm[p], &rl, 1) < 1e-9, "grad logml[%d]", p);
        CHECK(relerr(&grad[off], rg, ks[p].n_params + 1) < 1e-6, "gradient[%d] err %.3g", p,
              relerr(&grad[off], rg, ks[p].n_params + 1));
    }

    /* --- the same through a resident gradient job: one-shot = stage + run, new parameters ---- */
    {
        ngp_grad_job *gj = NULL;
        st = ngp_grad_stage(ctx, P, ks, N, t, y, 0, &gj);
        CHECK(st == NGP_OK && gj, "ngp_grad_stage: %s", ngp_strerror(st));
        if (gj) {
            double lm2[P], grad2[7 + 8 + 7];
            st = ngp_grad_job_run(gj, lm2, grad2, info);
            CHECK(st == NGP_OK, "ngp_grad_job_run: %s", ngp_strerror(st));
            CHECK(relerr(lm2, lm, P) == 0.0 && relerr(grad2, grad, 7 + 8 + 7) == 0.0,
                  "the resident job's first run differs from the one-shot call");
            /* new parameters for the same trees (what a leapfrog step sends), back to the old ones */
            double flat[6 + 7 + 6], nz[P], flat0[6 + 7 + 6], nz0[P];
            int o = 0;
            for (int p = 0; p < P; ++p) {
                for (int q = 0; q < ks[p].n_params; ++q, ++o) {
                    flat0[o] = ks[p].params[q];
                    flat[o] = ks[p].params[q] * 1.01;
                }
                nz0[p] = ks[p].noise;
                nz[p] = ks[p].noise * 0.9;
            }
            CHECK(ngp_grad_job_set_params(gj, flat, nz) == NGP_OK, "ngp_grad_job_set_params");
            CHECK(ngp_grad_job_run(gj, lm2, grad2, info) == NGP_OK, "run with new parameters");
            CHECK(relerr(lm2, lm, P) > 0.0, "new parameters did not reach the device");
            CHECK(ngp_grad_job_set_params(gj, flat0, nz0) == NGP_OK, "ngp_grad_job_set_params back");
            CHECK(ngp_grad_job_run(gj, lm2, grad2, info) == NGP_OK, "run with the old parameters");
            CHECK(relerr(lm2, lm, P) == 0.0 && relerr(grad2, grad, 7 + 8 + 7) == 0.0,
                  "the old parameters do not give the old answer");
            ngp_grad_job_destroy(gj);
        }
        /* the storage option: value jobs bit for bit the same without it */
        double lb3[P], lf3[P * D], mu3[P * D * M], sg3[P * M * M];
        CHECK(ngp_set_structured_storage(ctx, 0) == NGP_OK, "ngp_set_structured_storage off");
        st = ngp_nowcast_batch(ctx, P, ks, N, t, y, DD, t_add, D, y_add, M, t_new, 1, lb3, lf3, mu3, sg3, info);
        CHECK(st == NGP_OK && relerr(lf3, lf, P * D) == 0.0 && relerr(mu3, mu, P * D * M) == 0.0 &&
              relerr(sg3, sg, P * M * M) == 0.0, "structured storage changes a value job's bits");
        CHECK(ngp_set_structured_storage(ctx, 1) == NGP_OK, "ngp_set_structured_storage on");
    }

    /* --- resident factor: same answers without refactorising ------------------------------ */
    ngp_factor *f = NULL;
    st = ngp_factor_create(ctx, P, ks, N, t, y, 0, &f);
    CHECK(st == NGP_OK && f, "ngp_factor_create: %s", ngp_strerror(st));
    if (f) {
        double lb2[P], lf2[P * D], mu2[P * D * M], sg2[P * M * M];
        st = ngp_factor_nowcast(f, DD, t_add, D, y_add, M, t_new, 1, lb2, lf2, mu2, sg2, info);
        CHECK(st == NGP_OK, "ngp_factor_nowcast: %s", ngp_strerror(st));
        CHECK(relerr(lf2, lf, P * D) < 1e-10 && relerr(mu2, mu, P * D * M) < 1e-9 &&
              relerr(sg2, sg, P * M * M) < 1e-9, "resident factor differs from one-shot");
        ngp_factor_destroy(f);
    }

    /* --- weights, sampler ------------------------------------------------------------------ */
    double w[P], ess, lnorm, rw[P], ress, rln;
    ngp_weights_normalize(P, lb, w, &ess, &lnorm);
    ngpo_weights_normalize(P, lb, rw, &ress, &rln);
    CHECK(relerr(w, rw, P) < 1e-12 && fabs(ess - ress) < 1e-9, "weights_normalize");
    double wS[D * P], draws[D * 8 * M];
    int32_t comp[D * 8];
    for (int s = 0; s < D; ++s) memcpy(&wS[s * P], w, sizeof w);
    st = ngp_mixture_sample(ctx, P, D, M, wS, mu, sg, 8, 42ull, draws, comp, info);
    CHECK(st == NGP_OK, "ngp_mixture_sample: %s", ngp_strerror(st));
    for (int i = 0; i < D * 8; ++i) CHECK(comp[i] >= 0 && comp[i] < P, "component %d", (int)comp[i]);
    for (int i = 0; i < D * 8 * M; ++i) CHECK(isfinite(draws[i]), "draw %d not finite", i);

    /* --- error returns: never a crash ------------------------------------------------------- */
    int32_t bad_ops[] = {6};
    ngp_kernel bad = {1, 0, bad_ops, NULL, 0.1};
    CHECK(ngp_kernel_check(&bad) == NGP_ERR_PROGRAM, "malformed program accepted");
    CHECK(ngp_logml_batch(ctx, 1, &bad, N, t, y, 0, lm, info) == NGP_ERR_PROGRAM, "bad program ran");
    CHECK(ngp_logml_batch(ctx, 0, ks, N, t, y, 0, lm, info) == NGP_ERR_ARG, "B = 0 accepted");
    CHECK(ngp_logml_batch(NULL, 1, ks, N, t, y, 0, lm, info) == NGP_ERR_ARG, "NULL ctx accepted");
    double t_dup[N];
    memcpy(t_dup, t, sizeof t);
    t_dup[70] = t_dup[69];                       /* duplicate time, zero noise -> singular */
    ngp_kernel se = {1, 2, (int32_t[]){3}, (double[]){0.5, 1.0}, 0.0};
    ngp_spec nojit = spec;
    nojit.jitter = 0.0;
    ngp_set_spec(ctx, &nojit);
    st = ngp_logml_batch(ctx, 1, &se, N, t_dup, y, 0, lm, info);
    CHECK(st == NGP_OK && info[0] > 0, "singular matrix: status %d info %d", (int)st, (int)info[0]);
    ngp_set_spec(ctx, &spec);

    ngp_ctx_destroy(ctx);
    printf(fails ? "%d check(s) FAILED\n" : "all checks passed%.0d\n", fails);
    return fails ? 1 : 0;
}
