// Concurrent callers of the C-ABI against the mock HIP runtime (see mock_hip.cpp): several
// threads share one context (as Threads.@spawn tasks share the library in the reference,
// src/forecasting.jl:131-132), another thread creates and destroys contexts of its own, specs are
// flipped while jobs are staged and run.  Built with -fsanitize=thread and with
// -fsanitize=address,undefined by tests/test_host_sanitizers.py; exit code 0 and a silent sanitizer
// are the test.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../include/ngp.h"

extern "C" long mock_hip_launches(void);
extern "C" long mock_hip_live_allocations(void);
extern "C" long mock_hip_errors(void);

static std::atomic<int> fails{0};
#define CHECK(c, what) do { if (!(c)) { ++fails; std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what); } } while (0)

struct Ensemble {
    int32_t ops0[3] = {2, 5, 6}, ops1[3] = {4, 3, 8}, ops2[5] = {2, 5, 7, 3, 6};
    double par0[6] = {0.2, 0.1, 0.5, 0.9, 0.3, 0.7};
    double par1[7] = {0.4, 1.3, 0.9, 0.2, 0.7, 0.5, 0.1};
    double par2[8] = {0.1, 0.3, 0.8, 1.1, 0.21, 0.4, 0.3, 0.6};
    ngp_kernel ks[3];
    Ensemble() {
        ks[0] = {3, 6, ops0, par0, 0.05};
        ks[1] = {3, 7, ops1, par1, 0.02};
        ks[2] = {5, 8, ops2, par2, 0.1};
    }
};

static void worker(ngp_ctx *ctx, int id, int rounds) {
    Ensemble e;
    const int n = 130 + 7 * id, d = 2, D = 3, m = 4, P = 3;
    std::vector<double> t(n), y(n), t_add(d), y_add(D * d), t_new(m);
    for (int i = 0; i < n; ++i) { t[i] = (double)i / (n - 1); y[i] = std::sin(9.0 * t[i]); }
    for (int a = 0; a < d; ++a) t_add[a] = 1.0 + (double)(a + 1) / (n - 1);
    for (int i = 0; i < D * d; ++i) y_add[i] = 0.1 * i;
    for (int i = 0; i < m; ++i) t_new[i] = t_add[d - 1] + (double)(i + 1) / (n - 1);
    std::vector<double> lb(P), lf(P * D), mu(P * D * m), sg(P * m * m), grad(P * 16), lm(P);
    std::vector<int32_t> info(P);
    for (int r = 0; r < rounds; ++r) {
        ngp_spec sp;
        ngp_default_spec(&sp);
        if ((r + id) % 3 == 0) { sp.precision = NGP_PREC_MIXED; }
        if ((r + id) % 4 == 1) { sp.se_form = 1; sp.jitter = 1e-6; }
        CHECK(ngp_set_spec(ctx, &sp) == NGP_OK, "set_spec");
        ngp_spec back;
        CHECK(ngp_get_spec(ctx, &back) == NGP_OK, "get_spec");
        CHECK(ngp_nowcast_batch(ctx, P, e.ks, n, t.data(), y.data(), d, t_add.data(), D, y_add.data(),
                                m, t_new.data(), 1, lb.data(), lf.data(), mu.data(), sg.data(),
                                info.data()) == NGP_OK, "nowcast_batch");
        CHECK(ngp_logml_batch(ctx, P, e.ks, n, t.data(), y.data(), 0, lm.data(), info.data()) == NGP_OK,
              "logml_batch");
        CHECK(ngp_logml_grad_batch(ctx, P, e.ks, n, t.data(), y.data(), 0, lm.data(), grad.data(),
                                   info.data()) == NGP_OK, "logml_grad_batch");
        // staged job: run twice, fetch, destroy (the allocator hands its blocks to the next caller)
        ngp_job *job = nullptr;
        CHECK(ngp_nowcast_stage(ctx, P, e.ks, n, t.data(), y.data(), d, t_add.data(), D, y_add.data(),
                                m, t_new.data(), 1, &job) == NGP_OK && job, "nowcast_stage");
        if (job) {
            CHECK(ngp_job_run(job) == NGP_OK, "job_run");
            CHECK(ngp_job_run(job) == NGP_OK, "job_run again");
            CHECK(ngp_job_fetch(job, lb.data(), lf.data(), mu.data(), sg.data(), info.data()) == NGP_OK,
                  "job_fetch");
            ngp_job_destroy(job);
        }
        // resident gradient job: run, new parameters, run again, destroy; and the storage option
        ngp_grad_job *gj = nullptr;
        CHECK(ngp_grad_stage(ctx, P, e.ks, n, t.data(), y.data(), 0, &gj) == NGP_OK && gj, "grad_stage");
        if (gj) {
            CHECK(ngp_grad_job_run(gj, lm.data(), grad.data(), info.data()) == NGP_OK, "grad_job_run");
            std::vector<double> flat, nz;
            for (int k = 0; k < P; ++k) {
                for (int q = 0; q < e.ks[k].n_params; ++q) flat.push_back(e.ks[k].params[q] * 1.01);
                nz.push_back(e.ks[k].noise * 0.9);
            }
            CHECK(ngp_grad_job_set_params(gj, flat.data(), nz.data()) == NGP_OK, "grad_job_set_params");
            CHECK(ngp_grad_job_run(gj, nullptr, grad.data(), nullptr) == NGP_OK, "grad_job_run again");
            CHECK(ngp_grad_job_set_params(gj, nullptr, nz.data()) != NGP_OK, "null parameters accepted");
            ngp_grad_job_destroy(gj);
        }
        CHECK(ngp_set_structured_storage(ctx, (r + id) & 1) == NGP_OK, "set_structured_storage");
        // resident factor
        ngp_factor *f = nullptr;
        CHECK(ngp_factor_create(ctx, P, e.ks, n, t.data(), y.data(), 0, &f) == NGP_OK && f, "factor_create");
        if (f) {
            CHECK(ngp_factor_logml(f, lm.data(), info.data()) == NGP_OK, "factor_logml");
            CHECK(ngp_factor_nowcast(f, d, t_add.data(), D, y_add.data(), m, t_new.data(), 1, lb.data(),
                                     lf.data(), mu.data(), sg.data(), info.data()) == NGP_OK,
                  "factor_nowcast");
            ngp_factor_destroy(f);
        }
        // error returns must not leave anything behind
        ngp_kernel bad = e.ks[0];
        bad.n_ops = 2;
        CHECK(ngp_logml_batch(ctx, 1, &bad, n, t.data(), y.data(), 0, lm.data(), info.data()) != NGP_OK,
              "malformed program accepted");
        CHECK(ngp_logml_batch(ctx, 0, e.ks, n, t.data(), y.data(), 0, lm.data(), info.data()) != NGP_OK,
              "empty batch accepted");
        ngp_profile pr;
        CHECK(ngp_profile_enable(ctx, r & 1) == NGP_OK, "profile_enable");
        CHECK(ngp_profile_get(ctx, &pr) == NGP_OK, "profile_get");
        double w[5] = {0.1, -0.3, 0.2, NAN, -1.0}, wn[5], ess, ln;
        CHECK(ngp_weights_normalize(5, w, wn, &ess, &ln) == NGP_OK, "weights_normalize");
    }
}

static void churn(int rounds) {   // contexts of its own, created and destroyed while others work
    Ensemble e;
    for (int r = 0; r < rounds; ++r) {
        ngp_ctx *c = nullptr;
        CHECK(ngp_ctx_create(0, &c) == NGP_OK && c, "ctx_create");
        if (!c) return;
        double t[70], y[70], lm[3];
        int32_t info[3];
        for (int i = 0; i < 70; ++i) { t[i] = i / 69.0; y[i] = std::cos(5.0 * t[i]); }
        CHECK(ngp_logml_batch(c, 3, e.ks, 70, t, y, 0, lm, info) == NGP_OK, "logml_batch (own ctx)");
        ngp_ctx_destroy(c);
    }
}

int main() {
    ngp_ctx *ctx = nullptr;
    if (ngp_ctx_create(0, &ctx) != NGP_OK) return 2;
    const int T = 4, rounds = 6;
    std::vector<std::thread> th;
    for (int i = 0; i < T; ++i) th.emplace_back(worker, ctx, i, rounds);
    th.emplace_back(churn, 3 * rounds);
    for (auto &x : th) x.join();
    ngp_ctx_destroy(ctx);
    CHECK(mock_hip_errors() == 0, "bad free / out-of-bounds copy seen by the mock runtime");
    CHECK(mock_hip_live_allocations() == 0, "device allocations left after every context was destroyed");
    std::printf("host_stress: %ld kernel launches issued, %d failures\n", mock_hip_launches(), fails.load());
    return fails.load() ? 1 : 0;
}
