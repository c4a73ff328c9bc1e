// Flat combining of concurrent callers (include/ngp.h "concurrent callers"; the reference's
// Threads.@spawn-per-scenario pattern, src/forecasting.jl:131-159) against the mock HIP runtime with a
// "busy device" (every synchronisation sleeps): T threads enter the same one-shot entry point at
// once on the same dates.  Checked: every caller returns (no lost wake-up, no deadlock — the test
// has a time limit), every caller's output arrays were written in full and nothing beyond them,
// the requests shared launch sequences (ngp_combine_stats), a combined burst issues about the
// kernel launches of ONE call instead of T, incompatible requests (other dates, other entry points)
// keep their own sequences, bad arguments never join a group, and with combining switched off
// every request runs alone.  Built with -fsanitize=thread and -fsanitize=address,undefined by
// tests/test_host_sanitizers.py.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ngp.h"

extern "C" long mock_hip_launches(void);
extern "C" long mock_hip_live_allocations(void);
extern "C" long mock_hip_errors(void);
extern "C" void mock_hip_set_sync_delay_us(long us);

static std::atomic<int> fails{0};
#define CHECK(c, what) do { if (!(c)) { ++fails; std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what); } } while (0)

struct Gate {   // all threads of a burst leave together
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0, parties;
    explicit Gate(int n) : parties(n) {}
    void arrive() {
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++waiting == parties) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != generation; });
    }
};

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

constexpr double POISON = -777.25;
constexpr int P = 3, NG = 6 + 7 + 8 + 3;   // gradient entries of the ensemble

// kind: 0 logml, 1 logml + gradient, 2 predict, 3 mixture sample, 4 a resident gradient job run
// three times (the leapfrog steps of one HMC move), 5 / 6 logml / predict of scenario clones (same
// trees, observations that differ in their last two points only: served from one shared
// factorisation per particle); `n` picks the dates
static void one_call(ngp_ctx *ctx, int kind_in, int n, int id) {
    int kind = kind_in;
    Ensemble e;
    const int m = 4, draws = 5;
    std::vector<double> t(n), y(n), t_new(m);
    const bool clones = kind >= 5;
    for (int i = 0; i < n; ++i) {
        t[i] = (double)i / (n - 1);
        y[i] = std::sin(9.0 * t[i] + (clones && i < n - 2 ? 0 : id));
    }
    if (clones) kind = kind == 5 ? 0 : 2;
    for (int i = 0; i < m; ++i) t_new[i] = 1.0 + (double)(i + 1) / (n - 1);
    // every output array carries one guard element behind its end
    std::vector<double> lm(P + 1, POISON), grad(NG + 1, POISON), mu(P * m + 1, POISON),
        sg(P * m * m + 1, POISON), out(draws * m + 1, POISON);
    std::vector<int32_t> info(P + 1, -99), comp(draws + 1, -99);
    auto written = [](const std::vector<double> &v) {
        for (size_t i = 0; i + 1 < v.size(); ++i) if (v[i] == POISON) return false;
        return v.back() == POISON;
    };
    ngp_status st = NGP_OK;
    if (kind == 0) {
        st = ngp_logml_batch(ctx, P, e.ks, n, t.data(), y.data(), 0, lm.data(), info.data());
        CHECK(st == NGP_OK && written(lm) && info[0] != -99 && info[P - 1] != -99 && info[P] == -99, "logml_batch");
    } else if (kind == 1) {
        st = ngp_logml_grad_batch(ctx, P, e.ks, n, t.data(), y.data(), 0, lm.data(), grad.data(), info.data());
        CHECK(st == NGP_OK && written(lm) && written(grad) && info[P - 1] != -99 && info[P] == -99,
              "logml_grad_batch");
    } else if (kind == 2) {
        st = ngp_predict_batch(ctx, P, e.ks, n, t.data(), y.data(), 0, m, t_new.data(), 1, mu.data(),
                               sg.data(), lm.data(), info.data());
        CHECK(st == NGP_OK && written(lm) && written(mu) && written(sg) && info[P] == -99, "predict_batch");
    } else if (kind == 4) {
        ngp_grad_job *gj = nullptr;
        st = ngp_grad_stage(ctx, P, e.ks, n, t.data(), y.data(), 0, &gj);
        CHECK(st == NGP_OK && gj, "grad_stage");
        std::vector<double> flat, nz;
        for (int k = 0; k < P; ++k) {
            for (int q = 0; q < e.ks[k].n_params; ++q) flat.push_back(e.ks[k].params[q]);
            nz.push_back(e.ks[k].noise);
        }
        for (int step = 0; gj && step < 3; ++step) {
            for (double &v : flat) v *= 1.01;
            CHECK(ngp_grad_job_set_params(gj, flat.data(), nz.data()) == NGP_OK, "grad_job_set_params");
            std::fill(lm.begin(), lm.end(), POISON);
            std::fill(grad.begin(), grad.end(), POISON);
            st = ngp_grad_job_run(gj, lm.data(), grad.data(), info.data());
            CHECK(st == NGP_OK && written(lm) && written(grad) && info[P] == -99, "grad_job_run");
        }
        ngp_grad_job_destroy(gj);
    } else {
        std::vector<double> w(P, 1.0 / P), mmu(P * m, 0.5), msg(P * m * m, 0.0);
        for (int k = 0; k < P; ++k) for (int i = 0; i < m; ++i) msg[(k * m + i) * m + i] = 1.0;
        st = ngp_mixture_sample(ctx, P, 1, m, w.data(), mmu.data(), msg.data(), draws, 1234 + id,
                                out.data(), comp.data(), info.data());
        CHECK(st == NGP_OK && written(out) && comp[draws - 1] != -99 && comp[draws] == -99 && info[P] == -99,
              "mixture_sample");
    }
}

static void burst(ngp_ctx *ctx, int T, const std::vector<int> &kinds, const std::vector<int> &ns, int rounds) {
    Gate gate(T);
    std::vector<std::thread> th;
    for (int i = 0; i < T; ++i)
        th.emplace_back([&, i] {
            for (int r = 0; r < rounds; ++r) {
                gate.arrive();
                one_call(ctx, kinds[(size_t)i % kinds.size()], ns[(size_t)i % ns.size()], i);
            }
        });
    for (auto &x : th) x.join();
}

int main() {
    ngp_ctx *ctx = nullptr;
    if (ngp_ctx_create(0, &ctx) != NGP_OK) return 2;
    int64_t st4[6];
    const int T = 8;

    // reference: launches of ONE gradient call alone
    mock_hip_set_sync_delay_us(0);
    long l0 = mock_hip_launches();
    one_call(ctx, 1, 200, 0);
    const long single = mock_hip_launches() - l0;
    CHECK(single > 0, "a call launches kernels");
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK && st4[0] == 1 && st4[1] == 1 && st4[3] == 0,
          "a call that arrives alone runs alone");

    // (1) T callers of one entry point on the same dates behind a busy device
    mock_hip_set_sync_delay_us(20000);
    l0 = mock_hip_launches();
    burst(ctx, T, {1}, {200}, 1);
    const long combined = mock_hip_launches() - l0;
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK, "combine_stats");
    std::printf("burst of %d gradient calls: %lld requests in %lld launch sequences, largest group %lld, "
                "%ld kernel launches (one call alone: %ld)\n", T, (long long)st4[0], (long long)st4[1],
                (long long)st4[2], combined, single);
    CHECK(st4[0] == T, "every request was counted");
    CHECK(st4[1] <= 3, "eight concurrent requests took more than three launch sequences");
    CHECK(st4[2] >= T - 2, "the callers that piled up were not served together");
    CHECK(combined <= 3 * single + 8, "a combined burst should cost about the launches of one call, not of T");

    // (1b) the same with resident jobs: T tasks, each three runs of its own job
    mock_hip_set_sync_delay_us(5000);
    burst(ctx, T, {4}, {200}, 2);
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK && st4[0] == T * 3 * 2, "requests of the job runs");
    std::printf("resident jobs: %lld runs in %lld launch sequences, largest group %lld\n", (long long)st4[0],
                (long long)st4[1], (long long)st4[2]);
    CHECK(st4[2] >= T / 2 && st4[1] < st4[0], "concurrent runs of resident jobs were not combined");

    // (1c) scenario clones: same trees, observations equal up to the last two points
    mock_hip_set_sync_delay_us(5000);
    l0 = mock_hip_launches();
    burst(ctx, T, {5, 6}, {200}, 3);
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK && st4[0] == T * 3, "requests of the clone rounds");
    std::printf("scenario clones: %lld requests, %lld sequences, %lld served from a shared factorisation\n",
                (long long)st4[0], (long long)st4[1], (long long)st4[4]);
    CHECK(st4[4] >= T, "clones of one model were not served from one factorisation per particle");

    // (2) many rounds, every combinable entry point, two different series at once: nobody is lost
    mock_hip_set_sync_delay_us(300);
    burst(ctx, T, {0, 1, 2, 3, 4}, {200, 200, 200, 200, 200, 200, 131, 131}, 20);
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK && st4[0] >= T * 20, "requests of the mixed rounds");
    std::printf("mixed rounds: %lld requests, %lld sequences, largest group %lld, %lld shared\n",
                (long long)st4[0], (long long)st4[1], (long long)st4[2], (long long)st4[3]);
    CHECK(st4[2] <= 4, "requests of different entry points or dates shared a sequence");

    // (3) bad arguments are answered by the direct path and never poison a group
    {
        Ensemble e;
        ngp_kernel bad = e.ks[0];
        bad.n_ops = 2;
        double t[70], y[70], lm[3], g[32];
        int32_t info[3];
        for (int i = 0; i < 70; ++i) { t[i] = i / 69.0; y[i] = std::cos(5.0 * t[i]); }
        std::thread good([&] { one_call(ctx, 1, 200, 1); });
        CHECK(ngp_logml_batch(ctx, 1, &bad, 70, t, y, 0, lm, info) == NGP_ERR_PROGRAM, "malformed program");
        CHECK(ngp_logml_grad_batch(ctx, 3, e.ks, 70, t, y, 0, lm, nullptr, info) == NGP_ERR_ARG, "null gradient");
        CHECK(ngp_logml_grad_batch(ctx, 0, e.ks, 70, t, y, 0, lm, g, info) == NGP_ERR_ARG, "empty batch");
        CHECK(ngp_predict_batch(ctx, 3, e.ks, 70, t, y, 0, 0, nullptr, 1, nullptr, nullptr, lm, info) != NGP_OK,
              "predict without dates");
        good.join();
    }

    // (4) switched off: every request alone
    CHECK(ngp_set_combining(ctx, 0) == NGP_OK, "set_combining");
    (void)ngp_combine_stats(ctx, st4, 1);
    mock_hip_set_sync_delay_us(2000);
    burst(ctx, T, {1}, {200}, 2);
    CHECK(ngp_combine_stats(ctx, st4, 1) == NGP_OK && st4[0] == 0, "combining off: nothing goes through the queue");
    CHECK(ngp_set_combining(ctx, 1) == NGP_OK, "set_combining");

    mock_hip_set_sync_delay_us(0);
    ngp_ctx_destroy(ctx);
    CHECK(mock_hip_errors() == 0, "bad free / out-of-bounds copy seen by the mock runtime");
    CHECK(mock_hip_live_allocations() == 0, "device allocations left after the context was destroyed");
    std::printf("combine_stress: %d failures\n", fails.load());
    return fails.load() ? 1 : 0;
}
