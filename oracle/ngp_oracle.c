/*
 * ngp_oracle.c — CPU restatement of the GP arithmetic on the NowcastAutoGP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (nowcastautogp_amd/,
 * libngp.so) may link, import or call this file; only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py use it, as the checker.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the third-party Julia
 * package AutoGP.jl (reference Project.toml:7, compat "0.1.13" at Project.toml:15),
 * whose source is not in the reference checkout and not on this machine; Julia is
 * not installed; the reference's own tests assert only types, shapes, finiteness
 * and signs at this boundary (test/test_model_fitting.jl:31-124,
 * test/test_forecasting.jl:37-115, test/test_nowcast_functions.jl:150-294), so no
 * golden vector for logml / mu / Sigma exists.  This file therefore restates the
 * published GP identities and AutoGP's kernel grammar as recalled (SURVEY.md
 * Appendix B); every recalled choice is an ngp_spec flag.  It is cross-checked
 * against a second, independently written numpy/scipy implementation
 * (oracle/oracle_np.py) and against closed forms in tests/test_oracle.py.
 *
 * What each function follows:
 *   ngpo_kernel_eval / ngpo_cov  kernel grammar, opcodes 1..8:
 *        docs/src/vignettes/setting-priors.md:229-236 (numbering),
 *        SURVEY.md Appendix B (formulas, [RECALLED]).
 *   ngpo_logml     the per-particle evaluation behind fit_smc!/add_data!
 *        (src/make_and_fit_model.jl:91, src/forecasting.jl:135):
 *        logml = -1/2 y'K^-1 y - sum log L_ii - n/2 log 2pi, K = k(t,t)+(noise+jitter)I.
 *   ngpo_predict   the per-particle conditional MVN behind predict_mvn
 *        (src/forecasting.jl:46,66): mu = K21 K11^-1 y, S = K22 - K21 K11^-1 K12.
 *   ngpo_nowcast   one scenario task of forecast_with_nowcasts with
 *        n_mcmc = n_hmc = 0 (src/forecasting.jl:133-155), done the way the
 *        reference does it: an independent full factorisation at n+d per scenario.
 *   ngpo_weights_normalize   maybe_resample! arithmetic (src/forecasting.jl:138-141).
 *   ngpo_logml_grad   d logml / d theta = 1/2 tr((aa' - K^-1) dK/dtheta), the quantity
 *        HMC needs inside mcmc_parameters! (src/forecasting.jl:65,148).
 */
#include "ngp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const int k_nparams[9] = {0, 1, 3, 2, 3, 3, 0, 0, 2};

int ngpo_kernel_check(const ngp_kernel *k) {
    if (!k || !k->ops || k->n_ops <= 0 || k->n_ops > NGP_MAX_OPS) return NGP_ERR_PROGRAM;
    int depth = 0, np = 0;
    for (int i = 0; i < k->n_ops; ++i) {
        int op = k->ops[i];
        if (op < 1 || op > 8) return NGP_ERR_PROGRAM;
        np += k_nparams[op];
        if (op >= NGP_OP_PLUS) {
            if (depth < 2) return NGP_ERR_PROGRAM;
            depth -= 1;
        } else {
            depth += 1;
            if (depth > NGP_MAX_STACK) return NGP_ERR_TOO_LARGE;
        }
    }
    if (depth != 1) return NGP_ERR_PROGRAM;
    if (np != k->n_params || np > NGP_MAX_PARAMS) return NGP_ERR_PROGRAM;
    if (np > 0 && !k->params) return NGP_ERR_PROGRAM;
    return NGP_OK;
}

static double cp_sigma(const ngp_spec *s, double x, double loc, double scale) {
    double u = s->cp_form ? (x - loc) / scale : (loc - x) / scale;
    return 0.5 * (1.0 + tanh(u));
}

/* value only */
double ngpo_kernel_eval(const ngp_spec *s, const ngp_kernel *k, double t1, double t2) {
    double st[NGP_MAX_STACK];
    int sp = 0;
    const double *p = k->params;
    for (int i = 0; i < k->n_ops; ++i) {
        switch (k->ops[i]) {
        case NGP_OP_CONSTANT: st[sp++] = p[0]; p += 1; break;
        case NGP_OP_LINEAR:
            st[sp++] = p[1] + p[2] * (t1 - p[0]) * (t2 - p[0]); p += 3; break;
        case NGP_OP_SQEXP: {
            double d = t1 - t2;
            double den = s->se_form ? p[0] : p[0] * p[0];
            st[sp++] = p[1] * exp(-0.5 * d * d / den); p += 2; break;
        }
        case NGP_OP_GAMMAEXP: {
            double d = fabs(t1 - t2);
            st[sp++] = p[2] * exp(-pow(d / p[0], p[1])); p += 3; break;
        }
        case NGP_OP_PERIODIC: {
            double d = fabs(t1 - t2);
            double sn = sin(M_PI * d / p[1]);
            double c = s->periodic_form ? 2.0 / p[0] : 2.0 / (p[0] * p[0]);
            st[sp++] = p[2] * exp(-c * sn * sn); p += 3; break;
        }
        case NGP_OP_PLUS:  sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
        case NGP_OP_TIMES: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
        case NGP_OP_CHANGEPOINT: {
            double s1 = cp_sigma(s, t1, p[0], p[1]), s2 = cp_sigma(s, t2, p[0], p[1]);
            sp--;
            st[sp - 1] = s1 * st[sp - 1] * s2 + (1.0 - s1) * st[sp] * (1.0 - s2);
            p += 2; break;
        }
        default: return NAN;
        }
    }
    return st[0];
}

/* value + gradient w.r.t. every parameter (forward mode) */
static double kernel_eval_grad(const ngp_spec *s, const ngp_kernel *k, double t1, double t2,
                               double *grad /* n_params */) {
    const int NP = k->n_params;
    double st[NGP_MAX_STACK];
    /* gradient stack: each entry has NP slots */
    double *gs = (double *)calloc((size_t)NGP_MAX_STACK * (size_t)(NP > 0 ? NP : 1), sizeof(double));
    int sp = 0, pi = 0;
    const double *p = k->params;
#define G(sl, j) gs[(size_t)(sl) * (size_t)NP + (size_t)(j)]
    for (int i = 0; i < k->n_ops; ++i) {
        int op = k->ops[i];
        if (op < NGP_OP_PLUS) {
            for (int j = 0; j < NP; ++j) G(sp, j) = 0.0;
        }
        switch (op) {
        case NGP_OP_CONSTANT:
            st[sp] = p[pi]; G(sp, pi) = 1.0; sp++; pi += 1; break;
        case NGP_OP_LINEAR: {
            double c = p[pi], a1 = t1 - c, a2 = t2 - c;
            st[sp] = p[pi + 1] + p[pi + 2] * a1 * a2;
            G(sp, pi) = p[pi + 2] * (-a1 - a2);
            G(sp, pi + 1) = 1.0;
            G(sp, pi + 2) = a1 * a2;
            sp++; pi += 3; break;
        }
        case NGP_OP_SQEXP: {
            double d = t1 - t2, l = p[pi], a = p[pi + 1];
            double den = s->se_form ? l : l * l;
            double e = exp(-0.5 * d * d / den);
            st[sp] = a * e;
            G(sp, pi) = s->se_form ? a * e * 0.5 * d * d / (l * l) : a * e * d * d / (l * l * l);
            G(sp, pi + 1) = e;
            sp++; pi += 2; break;
        }
        case NGP_OP_GAMMAEXP: {
            double d = fabs(t1 - t2), l = p[pi], g = p[pi + 1], a = p[pi + 2];
            double r = d / l, u = pow(r, g), e = exp(-u);
            st[sp] = a * e;
            G(sp, pi) = a * e * g * u / l;
            G(sp, pi + 1) = (d > 0.0) ? -a * e * u * log(r) : 0.0;
            G(sp, pi + 2) = e;
            sp++; pi += 3; break;
        }
        case NGP_OP_PERIODIC: {
            double d = fabs(t1 - t2), l = p[pi], per = p[pi + 1], a = p[pi + 2];
            double ang = M_PI * d / per, sn = sin(ang), cs = cos(ang);
            double c = s->periodic_form ? 2.0 / l : 2.0 / (l * l);
            double e = exp(-c * sn * sn);
            st[sp] = a * e;
            G(sp, pi) = s->periodic_form ? a * e * 2.0 * sn * sn / (l * l)
                                         : a * e * 4.0 * sn * sn / (l * l * l);
            G(sp, pi + 1) = a * e * c * 2.0 * sn * cs * M_PI * d / (per * per);
            G(sp, pi + 2) = e;
            sp++; pi += 3; break;
        }
        case NGP_OP_PLUS:
            sp--;
            st[sp - 1] += st[sp];
            for (int j = 0; j < NP; ++j) G(sp - 1, j) += G(sp, j);
            break;
        case NGP_OP_TIMES: {
            sp--;
            double a = st[sp - 1], b = st[sp];
            for (int j = 0; j < NP; ++j) G(sp - 1, j) = G(sp - 1, j) * b + a * G(sp, j);
            st[sp - 1] = a * b;
            break;
        }
        case NGP_OP_CHANGEPOINT: {
            double loc = p[pi], sc = p[pi + 1];
            double sgn = s->cp_form ? 1.0 : -1.0; /* u = sgn*(x-loc)/sc */
            double u1 = sgn * (t1 - loc) / sc, u2 = sgn * (t2 - loc) / sc;
            double th1 = tanh(u1), th2 = tanh(u2);
            double s1 = 0.5 * (1.0 + th1), s2 = 0.5 * (1.0 + th2);
            double q1 = 0.5 * (1.0 - th1 * th1), q2 = 0.5 * (1.0 - th2 * th2); /* dsigma/du */
            /* du/dloc = -sgn/sc ; du/dscale = -u/sc */
            double d1l = q1 * (-sgn / sc), d2l = q2 * (-sgn / sc);
            double d1s = q1 * (-u1 / sc), d2s = q2 * (-u2 / sc);
            sp--;
            double kl = st[sp - 1], kr = st[sp];
            for (int j = 0; j < NP; ++j)
                G(sp - 1, j) = s1 * s2 * G(sp - 1, j) + (1.0 - s1) * (1.0 - s2) * G(sp, j);
            G(sp - 1, pi) += d1l * kl * s2 + s1 * kl * d2l - d1l * kr * (1.0 - s2) - (1.0 - s1) * kr * d2l;
            G(sp - 1, pi + 1) += d1s * kl * s2 + s1 * kl * d2s - d1s * kr * (1.0 - s2) - (1.0 - s1) * kr * d2s;
            st[sp - 1] = s1 * kl * s2 + (1.0 - s1) * kr * (1.0 - s2);
            pi += 2; break;
        }
        default: break;
        }
    }
    for (int j = 0; j < NP; ++j) grad[j] = G(0, j);
#undef G
    double v = st[0];
    free(gs);
    return v;
}

int ngpo_cov(const ngp_spec *s, const ngp_kernel *k, int n1, const double *t1, int n2,
             const double *t2, int add_diag, double *out) {
    int st = ngpo_kernel_check(k);
    if (st) return st;
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            double v = ngpo_kernel_eval(s, k, t1[i], t2[j]);
            if (add_diag && i == j) v += k->noise + s->jitter;
            out[(size_t)i * (size_t)n2 + (size_t)j] = v;
        }
    return NGP_OK;
}

/* in-place lower Cholesky, row-major, leading dimension lda; LAPACK potrf info */
int ngpo_chol(int n, double *a, int lda) {
    for (int i = 0; i < n; ++i) {
        double *ai = a + (size_t)i * (size_t)lda;
        for (int j = 0; j <= i; ++j) {
            const double *aj = a + (size_t)j * (size_t)lda;
            double sum = ai[j];
            for (int p = 0; p < j; ++p) sum -= ai[p] * aj[p];
            if (i == j) {
                if (!(sum > 0.0)) return i + 1;
                ai[j] = sqrt(sum);
            } else {
                ai[j] = sum / aj[j];
            }
        }
    }
    return 0;
}

/* solve L z = b in place (forward substitution) */
static void fwd_solve(int n, const double *l, int lda, double *b) {
    for (int i = 0; i < n; ++i) {
        const double *li = l + (size_t)i * (size_t)lda;
        double sum = b[i];
        for (int p = 0; p < i; ++p) sum -= li[p] * b[p];
        b[i] = sum / li[i];
    }
}
/* solve L' x = b in place (back substitution) */
static void bwd_solve(int n, const double *l, int lda, double *b) {
    for (int i = n - 1; i >= 0; --i) {
        double sum = b[i];
        for (int p = i + 1; p < n; ++p) sum -= l[(size_t)p * (size_t)lda + (size_t)i] * b[p];
        b[i] = sum / l[(size_t)i * (size_t)lda + (size_t)i];
    }
}

static double logml_from_factor(int n, const double *l, const double *z) {
    double quad = 0.0, ld = 0.0;
    for (int i = 0; i < n; ++i) {
        quad += z[i] * z[i];
        ld += log(l[(size_t)i * (size_t)n + (size_t)i]);
    }
    return -0.5 * quad - ld - 0.5 * (double)n * log(2.0 * M_PI);
}

int ngpo_logml(const ngp_spec *s, const ngp_kernel *k, int n, const double *t, const double *y,
               double *logml) {
    int st = ngpo_kernel_check(k);
    if (st) return st;
    double *K = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    ngpo_cov(s, k, n, t, n, t, 1, K);
    int info = ngpo_chol(n, K, n);
    if (info == 0) {
        memcpy(z, y, sizeof(double) * (size_t)n);
        fwd_solve(n, K, n, z);
        *logml = logml_from_factor(n, K, z);
    } else {
        *logml = NAN;
    }
    free(K); free(z);
    return info;
}

int ngpo_predict(const ngp_spec *s, const ngp_kernel *k, int n, const double *t, const double *y,
                 int m, const double *t_new, int noise_on_new, double *mu, double *sigma,
                 double *logml) {
    int st = ngpo_kernel_check(k);
    if (st) return st;
    double *K = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *K21 = (double *)malloc(sizeof(double) * (size_t)m * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    ngpo_cov(s, k, n, t, n, t, 1, K);
    int info = ngpo_chol(n, K, n);
    if (info) {
        for (int i = 0; i < m; ++i) mu[i] = NAN;
        for (int i = 0; i < m * m; ++i) sigma[i] = NAN;
        if (logml) *logml = NAN;
        free(K); free(a); free(K21); free(v);
        return info;
    }
    memcpy(a, y, sizeof(double) * (size_t)n);
    fwd_solve(n, K, n, a);
    if (logml) *logml = logml_from_factor(n, K, a);
    bwd_solve(n, K, n, a); /* a = K^-1 y */
    ngpo_cov(s, k, m, t_new, n, t, 0, K21);
    ngpo_cov(s, k, m, t_new, m, t_new, 0, sigma);
    for (int i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int p = 0; p < n; ++p) acc += K21[(size_t)i * (size_t)n + (size_t)p] * a[p];
        mu[i] = acc;
    }
    /* sigma -= K21 K11^-1 K12, one column of K12 at a time */
    for (int j = 0; j < m; ++j) {
        memcpy(v, K21 + (size_t)j * (size_t)n, sizeof(double) * (size_t)n);
        fwd_solve(n, K, n, v);
        bwd_solve(n, K, n, v); /* v = K11^-1 K12[:,j] */
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
            for (int p = 0; p < n; ++p) acc += K21[(size_t)i * (size_t)n + (size_t)p] * v[p];
            sigma[(size_t)i * (size_t)m + (size_t)j] -= acc;
        }
    }
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < i; ++j) {
            double sym = 0.5 * (sigma[(size_t)i * (size_t)m + j] + sigma[(size_t)j * (size_t)m + i]);
            sigma[(size_t)i * (size_t)m + j] = sym;
            sigma[(size_t)j * (size_t)m + i] = sym;
        }
        if (noise_on_new) sigma[(size_t)i * (size_t)m + i] += k->noise + s->jitter;
    }
    free(K); free(a); free(K21); free(v);
    return 0;
}

int ngpo_nowcast(const ngp_spec *s, const ngp_kernel *k, int n, const double *t, const double *y,
                 int d, const double *t_add, int D, const double *y_add, int m,
                 const double *t_new, int noise_on_new, double *logml_base, double *logml_full,
                 double *mu, double *sigma) {
    int info = 0;
    if (logml_base) {
        int i0 = ngpo_logml(s, k, n, t, y, logml_base);
        if (i0) info = i0;
    }
    int nn = n + d;
    double *tt = (double *)malloc(sizeof(double) * (size_t)nn);
    double *yy = (double *)malloc(sizeof(double) * (size_t)nn);
    double *sg = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m * m : 1));
    double *mm = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    memcpy(tt, t, sizeof(double) * (size_t)n);
    memcpy(tt + n, t_add, sizeof(double) * (size_t)d);
    memcpy(yy, y, sizeof(double) * (size_t)n);
    for (int sc = 0; sc < D; ++sc) {
        memcpy(yy + n, y_add + (size_t)sc * (size_t)d, sizeof(double) * (size_t)d);
        double lf = NAN;
        int i1;
        if (m > 0) {
            i1 = ngpo_predict(s, k, nn, tt, yy, m, t_new, noise_on_new, mm, sg, &lf);
            if (mu) memcpy(mu + (size_t)sc * (size_t)m, mm, sizeof(double) * (size_t)m);
            if (sigma && sc == 0) memcpy(sigma, sg, sizeof(double) * (size_t)m * (size_t)m);
        } else {
            i1 = ngpo_logml(s, k, nn, tt, yy, &lf);
        }
        if (i1 && !info) info = i1;
        if (logml_full) logml_full[sc] = lf;
    }
    free(tt); free(yy); free(sg); free(mm);
    return info;
}

int ngpo_logml_grad(const ngp_spec *s, const ngp_kernel *k, int n, const double *t,
                    const double *y, double *logml, double *grad) {
    int st = ngpo_kernel_check(k);
    if (st) return st;
    const int NP = k->n_params;
    size_t nn = (size_t)n * (size_t)n;
    double *L = (double *)malloc(sizeof(double) * nn);
    double *Kinv = (double *)malloc(sizeof(double) * nn);
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *col = (double *)malloc(sizeof(double) * (size_t)n);
    double *g = (double *)malloc(sizeof(double) * (size_t)(NP > 0 ? NP : 1));
    ngpo_cov(s, k, n, t, n, t, 1, L);
    int info = ngpo_chol(n, L, n);
    if (info) {
        *logml = NAN;
        for (int j = 0; j <= NP; ++j) grad[j] = NAN;
        free(L); free(Kinv); free(a); free(col); free(g);
        return info;
    }
    memcpy(a, y, sizeof(double) * (size_t)n);
    fwd_solve(n, L, n, a);
    *logml = logml_from_factor(n, L, a);
    bwd_solve(n, L, n, a);
    for (int j = 0; j < n; ++j) {
        memset(col, 0, sizeof(double) * (size_t)n);
        col[j] = 1.0;
        fwd_solve(n, L, n, col);
        bwd_solve(n, L, n, col);
        for (int i = 0; i < n; ++i) Kinv[(size_t)i * (size_t)n + (size_t)j] = col[i];
    }
    for (int j = 0; j <= NP; ++j) grad[j] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double w = 0.5 * (a[i] * a[j] - Kinv[(size_t)i * (size_t)n + (size_t)j]);
            kernel_eval_grad(s, k, t[i], t[j], g);
            for (int q = 0; q < NP; ++q) grad[q] += w * g[q];
            if (i == j) grad[NP] += w;
        }
    free(L); free(Kinv); free(a); free(col); free(g);
    return 0;
}

int ngpo_weights_normalize(int P, const double *logw, double *w_norm, double *ess,
                           double *log_norm) {
    if (P <= 0 || !logw) return NGP_ERR_ARG;
    /* non-finite entries (a failed particle: -inf, or NaN) carry weight 0 */
    double mx = -INFINITY;
    for (int i = 0; i < P; ++i) if (isfinite(logw[i]) && logw[i] > mx) mx = logw[i];
    if (!(mx > -INFINITY)) { /* nothing finite: undefined weights */
        if (ess) *ess = NAN;
        if (log_norm) *log_norm = -INFINITY;
        if (w_norm) for (int i = 0; i < P; ++i) w_norm[i] = NAN;
        return NGP_OK;
    }
    double sum = 0.0;
    for (int i = 0; i < P; ++i) sum += isfinite(logw[i]) ? exp(logw[i] - mx) : 0.0;
    double sq = 0.0;
    for (int i = 0; i < P; ++i) {
        double w = (isfinite(logw[i]) ? exp(logw[i] - mx) : 0.0) / sum;
        if (w_norm) w_norm[i] = w;
        sq += w * w;
    }
    if (ess) *ess = 1.0 / sq;
    if (log_norm) *log_norm = mx + log(sum);
    return NGP_OK;
}

void ngpo_default_spec(ngp_spec *s) {
    s->se_form = 0;
    s->periodic_form = 0;
    s->cp_form = 0;
    s->precision = 0;
    s->mixed_tau = 1e-6;
    s->refine_tol = 1e-9;
    s->refine_max = 3;
    s->reserved = 0;
    s->jitter = 1e-5;
}
