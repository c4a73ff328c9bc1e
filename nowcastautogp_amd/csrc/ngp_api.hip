// ngp_api.hip — host side of libngp: contexts, staged jobs, launch schedule, C-ABI (include/ngp.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ngp_internal.h"

using namespace ngp;

#define HIPCHK(expr)                                                     \
    do {                                                                 \
        hipError_t e__ = (expr);                                         \
        if (e__ != hipSuccess) return (ngp_status)(e__ > 0 ? e__ : 999); \
    } while (0)

// ---- page-locked host memory for everything that crosses the bus ---------------------------
// A copy to or from pageable memory is not asynchronous: the runtime waits for the stream, moves
// the bytes through a bounce buffer of its own and only then returns (kernel trace of a 24-particle
// call at n = 208: the result copy started 25 us after the last kernel had ended).  The staging
// vectors of the jobs therefore live in page-locked memory, from a process-wide pool: blocks are
// kept by size class when a vector lets go of them, because hipHostMalloc costs a hundred
// microseconds and a fit stages ten thousand jobs.  When page-locking fails (no device, limits)
// the block is ordinary memory and everything still works, only slower.
namespace {
class PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_;          // size class -> block
    std::map<void *, std::pair<size_t, bool>> live_;   // block -> (class, page-locked)
    size_t cached_ = 0;
    static constexpr size_t MAX_CACHED = (size_t)256 << 20, MAX_PINNED_BLOCK = (size_t)8 << 20;
    static size_t size_class(size_t n) {
        size_t c = 4096;
        while (c < n) c <<= 1;
        return c;
    }
public:
    void *take(size_t n) {
        const size_t c = size_class(n);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_.find(c);
            if (it != free_.end()) {
                void *p = it->second;
                free_.erase(it);
                cached_ -= c;
                return p;
            }
        }
        // (page-locking is for the latency of small jobs; a large job's staging copy is a one-off whose
        // time is its bytes, and locking hundreds of megabytes costs more than it saves)
        void *p = nullptr;
        bool pinned = c <= MAX_PINNED_BLOCK &&
                      hipHostMalloc(&p, c, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && p;
        if (!pinned) {
            (void)hipGetLastError();
            p = std::malloc(c);
            if (!p) throw std::bad_alloc();
        }
        std::lock_guard<std::mutex> lk(mu);
        live_[p] = {c, pinned};
        return p;
    }
    bool is_pinned(const void *p) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live_.find(const_cast<void *>(p));
        return it != live_.end() && it->second.second;
    }
    void give(void *p) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live_.find(p);
        if (it == live_.end()) return;
        const size_t c = it->second.first;
        // (small blocks are always kept: were the cache ever full of large ones, every 24-item call
        // would page-lock and unlock its staging buffers again — 450 us per call, seen once)
        if (c <= MAX_PINNED_BLOCK && (c <= ((size_t)256 << 10) || cached_ + c <= MAX_CACHED)) {
            free_.emplace(c, p);
            cached_ += c;
            return;
        }
        const bool pinned = it->second.second;
        live_.erase(it);
        if (pinned) (void)hipHostFree(p);
        else std::free(p);
    }
    ~PinnedPool() {   // process exit: the runtime may already be gone — ordinary blocks only
        for (auto &kv : free_) {
            auto it = live_.find(kv.second);
            if (it != live_.end() && !it->second.second) std::free(kv.second);
        }
    }
};
inline PinnedPool &pinned_pool() {
    static PinnedPool *pool = new PinnedPool();   // never destroyed: vectors of static lifetime may outlive main
    return *pool;
}
template <class T> struct PinnedAlloc {
    typedef T value_type;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(pinned_pool().take(n * sizeof(T))); }
    void deallocate(T *p, size_t) { pinned_pool().give(p); }
    template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <class T> using PinVec = std::vector<T, PinnedAlloc<T>>;
}  // namespace

namespace {
struct ScopedEvent {   // destroyed on every exit path of the microbenchmarks
    hipEvent_t e = nullptr;
    ScopedEvent() { (void)hipEventCreate(&e); }
    ~ScopedEvent() { if (e) (void)hipEventDestroy(e); }
    operator hipEvent_t() const { return e; }
};
}  // namespace

namespace {

constexpr int k_nparams[10] = {0, 1, 3, 2, 3, 3, 0, 0, 2, 2};

// ---------------------------------------------------------------------------------------
// program validation + flattening for the device
// ---------------------------------------------------------------------------------------
ngp_status check_program(const ngp_kernel *k) {
    if (!k || !k->ops || k->n_ops <= 0) return NGP_ERR_PROGRAM;
    if (k->n_ops > NGP_MAX_OPS) return NGP_ERR_TOO_LARGE;
    int depth = 0, np = 0;
    for (int i = 0; i < k->n_ops; ++i) {
        const int op = k->ops[i];
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
    if (np != k->n_params) return NGP_ERR_PROGRAM;
    if (np > NGP_MAX_PARAMS) return NGP_ERR_TOO_LARGE;
    if (np > 0 && !k->params) return NGP_ERR_PROGRAM;
    return NGP_OK;
}

struct TNode {
    int op, left, right, pfirst, need;
};

// Re-emit the postfix program so the deeper subtree of every operator is evaluated first
// (Sethi-Ullman): the device keeps its evaluation stack in DEV_STACK registers.  Plus/Times
// commute bit-exactly in IEEE arithmetic; ChangePoint gets a swapped-operand opcode.
// perm[device param index] = caller's param index.
ngp_status compile_program(const ngp_kernel *k, DevProgram *out, std::vector<int> *perm,
                           int *n_stat = nullptr, int *n_cp = nullptr, int *n_tab = nullptr) {
    ngp_status st = check_program(k);
    if (st) return st;
    std::vector<TNode> nodes;
    std::vector<int> stack;
    int pi = 0;
    for (int i = 0; i < k->n_ops; ++i) {
        const int op = k->ops[i];
        TNode nd{op, -1, -1, pi, 1};
        if (op >= NGP_OP_PLUS) {
            nd.right = stack.back(); stack.pop_back();
            nd.left = stack.back(); stack.pop_back();
            const int a = nodes[nd.left].need, b = nodes[nd.right].need;
            nd.need = (a == b) ? a + 1 : std::max(a, b);
        }
        pi += k_nparams[op];
        nodes.push_back(nd);
        stack.push_back((int)nodes.size() - 1);
    }
    if (nodes[stack.back()].need > DEV_STACK) return NGP_ERR_TOO_LARGE;
    std::memset(out, 0, sizeof(DevProgram));
    out->n_ops = k->n_ops;
    out->n_params = k->n_params;
    out->noise = k->noise;
    int no = 0, np = 0, nstat = 0, ncp = 0;
    if (perm) perm->clear();
    // iterative post-order with child reordering
    struct Frame { int node, stage; };
    std::vector<int> dev_index(nodes.size(), -1), dev_first(nodes.size(), -1);
    std::vector<Frame> fs{{stack.back(), 0}};
    while (!fs.empty()) {
        Frame &f = fs.back();
        const TNode &nd = nodes[f.node];
        if (dev_first[(size_t)f.node] < 0) dev_first[(size_t)f.node] = no;   // first op of its subtree
        const bool swap = nd.op >= NGP_OP_PLUS && nodes[nd.right].need > nodes[nd.left].need;
        if (nd.op < NGP_OP_PLUS || f.stage == 2) {
            int op = nd.op;
            if (op == NGP_OP_CHANGEPOINT && swap) op = OP_CP_SWAPPED;
            if (nd.op >= NGP_OP_SQEXP && nd.op <= NGP_OP_PERIODIC) out->slot[no] = (uint8_t)nstat++;
            if (nd.op == NGP_OP_CHANGEPOINT) out->slot[no] = (uint8_t)ncp++;
            if (nd.op >= NGP_OP_PLUS)
                out->first[no] = (uint8_t)dev_index[(size_t)(swap ? nd.right : nd.left)];
            out->poff[no] = (uint8_t)np;
            dev_index[(size_t)f.node] = no;
            out->ops[no++] = (uint8_t)op;
            for (int q = 0; q < k_nparams[nd.op]; ++q) {
                out->params[np++] = k->params[nd.pfirst + q];
                if (perm) perm->push_back(nd.pfirst + q);
            }
            fs.pop_back();
        } else if (f.stage == 0) {
            f.stage = 1;
            fs.push_back({swap ? nd.right : nd.left, 0});
        } else {
            f.stage = 2;
            fs.push_back({swap ? nd.left : nd.right, 0});
        }
    }
    if (n_stat) *n_stat = nstat;
    if (n_cp) *n_cp = ncp;
    // ---- reduced program: maximal stationary subtrees become table leaves ----
    std::vector<char> stat(nodes.size(), 0);
    for (size_t i = 0; i < nodes.size(); ++i) {   // postfix input order: children come first
        const TNode &nd = nodes[i];
        if (nd.op == NGP_OP_PLUS || nd.op == NGP_OP_TIMES)
            stat[i] = stat[(size_t)nd.left] && stat[(size_t)nd.right];
        else
            stat[i] = nd.op != NGP_OP_LINEAR && nd.op != NGP_OP_CHANGEPOINT;
    }
    int nr = 0, ntab = 0;
    struct RFrame { int node, stage; };
    auto new_table = [&](int node) {
        out->tb_first[ntab] = (uint8_t)dev_first[(size_t)node];
        out->tb_last[ntab] = (uint8_t)dev_index[(size_t)node];
        return ntab++;
    };
    std::vector<RFrame> rs{{stack.back(), 0}};
    while (!rs.empty()) {
        RFrame &f = rs.back();
        const TNode &nd = nodes[(size_t)f.node];
        const int di = dev_index[(size_t)f.node];
        if (stat[(size_t)f.node]) {
            out->rops[nr] = (uint8_t)OP_TABLE;
            out->rslot[nr] = (uint8_t)new_table(f.node);
            ++nr;
            rs.pop_back();
        } else if (nd.op < NGP_OP_PLUS) {   // Linear
            out->rops[nr] = (uint8_t)nd.op;
            out->rpoff[nr] = out->poff[di];
            ++nr;
            rs.pop_back();
        } else {
            const bool swap = nodes[(size_t)nd.right].need > nodes[(size_t)nd.left].need;
            const int second = swap ? nd.left : nd.right;
            const bool leaf2 = stat[(size_t)second] || nodes[(size_t)second].op == NGP_OP_LINEAR;
            if (f.stage == 0) {
                f.stage = 1;
                rs.push_back({swap ? nd.right : nd.left, 0});
            } else if (f.stage == 1 && !leaf2) {
                f.stage = 2;
                rs.push_back({second, 0});
            } else {
                int code = out->ops[di];            // Plus / Times / ChangePoint or its swapped form
                if (leaf2) {                        // the second operand rides along (never pushed)
                    if (stat[(size_t)second]) {
                        code |= RLEAF_TABLE << 4;
                        out->rleaf[nr] = (uint8_t)new_table(second);
                    } else {
                        code |= RLEAF_LINEAR << 4;
                        out->rleaf[nr] = out->poff[dev_index[(size_t)second]];
                    }
                }
                out->rops[nr] = (uint8_t)code;
                out->rslot[nr] = out->slot[di];
                out->rpoff[nr] = out->poff[di];
                ++nr;
                rs.pop_back();
            }
        }
    }
    out->n_rops = nr;
    out->n_tab = ntab;
    out->rchain = 1;
    for (int i = 1; i < nr; ++i)
        if ((out->rops[i] >> 4) == 0) out->rchain = 0;
    if (n_tab) *n_tab = ntab;
    return NGP_OK;
}

// Do all times sit on a lattice t = tmin + q h (true for integer-day dates after the [0,1]
// rescale)?  Floating-point Euclid over the gaps; accepted only if every point is reproduced to a
// few ulp, so a table value at distance k h equals the direct evaluation to rounding.
bool detect_lattice(const std::vector<double> &t, double *h_out, std::vector<int32_t> *q,
                    int *R_out) {
    const size_t n = t.size();
    if (n < 2) return false;
    double tmin = t[0], tmax = t[0];
    for (double v : t) { tmin = std::min(tmin, v); tmax = std::max(tmax, v); }
    if (!(tmax > tmin) || !std::isfinite(tmax) || !std::isfinite(tmin)) return false;
    const double span = tmax - tmin;
    const double tol = 1e-9 * span;
    double g = 0.0;
    for (double v : t) {
        double a = v - tmin;
        if (a <= tol) continue;
        if (g == 0.0) { g = a; continue; }
        double x = g, y = a;          // Euclid with snapping
        for (int it = 0; it < 64 && y > tol; ++it) {
            double r = std::fmod(x, y);
            if (y - r <= tol) r = 0.0;
            x = y;
            y = r;
        }
        g = x;
        if (g < span / (double)(1 << 20)) return false;
    }
    if (g <= 0.0) return false;
    const double qmaxd = std::round(span / g);
    if (qmaxd < 1.0 || qmaxd > (double)(1 << 20)) return false;
    // a lattice far sparser than the data (a few points on a very fine grid) would cost more in
    // tables (R entries per stationary subtree and item) than direct evaluation costs in the fill
    if (qmaxd > 16.0 * (double)n + 4096.0) return false;
    const double h = span / qmaxd;
    const double scale = std::max(std::max(std::fabs(tmin), std::fabs(tmax)), 1.0);
    q->resize(n);
    for (size_t i = 0; i < n; ++i) {
        const double qi = std::round((t[i] - tmin) / h);
        if (std::fabs((t[i] - tmin) - qi * h) > 16.0 * 2.220446049250313e-16 * scale) return false;
        (*q)[i] = (int32_t)qi;
    }
    *h_out = h;
    *R_out = (int)qmaxd + 1;
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
struct ngp_comb_req;   // one caller's request while it waits to be combined (ngp_combine below)

struct ngp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // diag-ahead tiles run beside the main schedule
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // extra lanes: small chunks of long series are swept as several sub-chunks side by side
    // (factor_chunk); made on first use
    static constexpr int MAX_LANES = 2;
    hipStream_t lane_main[MAX_LANES] = {}, lane_side[MAX_LANES] = {};
    hipEvent_t lane_fork[MAX_LANES] = {}, lane_join[MAX_LANES] = {}, lane_done[MAX_LANES] = {};
    hipEvent_t ev_lane_go = nullptr;
    int lanes_made = 1;   // lane 0 is (stream, side, ev_fork, ev_join)
    bool make_lanes(int n) {
        if (!ev_lane_go && hipEventCreateWithFlags(&ev_lane_go, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ev_lane_go = nullptr;
            return false;
        }
        for (; lanes_made < n; ++lanes_made) {
            const int i = lanes_made;
            if (hipStreamCreateWithFlags(&lane_main[i], hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&lane_side[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&lane_fork[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&lane_join[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&lane_done[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
        }
        return true;
    }
    ngp_spec spec{};
    std::mutex mu;
    // Flat combining of concurrent callers ("combining" section below): one-shot calls that arrive
    // while the device is busy wait in `pending`; the thread that finds nobody serving takes ALL of
    // them and runs every group of compatible requests as ONE launch sequence.  qmu guards these
    // fields only and is never held across a device call; mu stays the execution lock.
    std::mutex qmu;
    std::vector<ngp_comb_req *> pending;
    bool combining = false, combine_on = true, combine_linger = true;
    int64_t comb_stats[6] = {};   // requests | launch sequences | largest group | requests that shared one |
                                  // requests served from ONE factorisation per particle | (reserved)
    // company seen lately: the number of requests the last server took (decays when a wait for
    // it was in vain) and the condition a would-be server waits on for at most COMB_LINGER_US
    size_t comb_expect = 1;
    std::condition_variable comb_arrival;
    bool profiling = false;
    bool toeplitz = true;   // ngp_set_structured_storage
    bool invariant = false; // ngp_set_batch_invariant (JobGeom::invariant of the jobs staged after)
    bool short_series = true;   // ngp_set_short_series_path: n0 <= 256 factorised in one launch
    ngp_profile prof{};
    size_t mem_cap = 0;  // bytes the factor storage of one job may take
    // caching allocator: repeated jobs of the same shape (the SMC/MCMC loop, the steps of a
    // staged batch) reuse blocks.  A context that cannot allocate first gives its own cache back
    // to the device, then the caches of the process's other contexts (registry below).
    std::multimap<size_t, void *> free_blocks;
    std::map<void *, size_t> live;
    size_t cached_bytes = 0;

    void drop_cache() {
        for (auto &kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
        cached_bytes = 0;
    }
    // Workspace: ONE grow-only block for the working storage of a call (factor slabs, K^-1, tables —
    // everything a run takes at its start and gives back at its end; calls on a context are
    // serialised and end synchronised, so the next call may overwrite it).  Jobs of different
    // shapes alternate in the reference's flow — gradient calls (115 + 57 GB blocks at 12,800 items)
    // and a predictive call (221 GB) in forecast_with_nowcasts' refinement modes — and with
    // per-shape blocks every switch freed and re-allocated most of the device: 7.7 of the 8.5 s the
    // predictive call of the lockstep HMC leg took (scripts/hmc_leg_cprofile.py).  Things that
    // outlive a call (job arenas, resident factors) stay with alloc / release.
    void *ws_base = nullptr;
    size_t ws_cap = 0, ws_used = 0;
    void ws_drop() {
        if (ws_base) (void)hipFree(ws_base);
        ws_base = nullptr;
        ws_cap = ws_used = 0;
    }
    static size_t ws_round(size_t bytes) { return (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; }
    // room for `total` bytes of bump allocations (sum of ws_round of the pieces); the previous
    // call's contents are dead
    ngp_status ws_reserve(size_t total) {
        ws_used = 0;
        if (total <= ws_cap) return NGP_OK;
        ws_drop();
        hipError_t e = hipMalloc(&ws_base, total);
        for (int attempt = 0; e != hipSuccess && attempt < 2; ++attempt) {
            (void)hipGetLastError();   // handled here: must not surface as the job's last error
            if (attempt == 0) drop_cache();
            else drop_other_caches(this);
            e = hipMalloc(&ws_base, total);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ws_base = nullptr;
            return NGP_ERR_TOO_LARGE;
        }
        ws_cap = total;
        return NGP_OK;
    }
    void *ws_take(size_t bytes) {   // inside the reservation: cannot fail
        bytes = ws_round(bytes);
        if (ws_used + bytes > ws_cap) return nullptr;
        void *p = (char *)ws_base + ws_used;
        ws_used += bytes;
        return p;
    }
    static void drop_other_caches(ngp_ctx *self);
    ngp_status alloc(void **p, size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        if (bytes == 0) bytes = 256;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes * 2) {
            *p = it->second;
            live[*p] = it->first;
            cached_bytes -= it->first;
            free_blocks.erase(it);
            return NGP_OK;
        }
        hipError_t e = hipMalloc(p, bytes);
        for (int attempt = 0; e != hipSuccess && attempt < 2; ++attempt) {
            (void)hipGetLastError();   // handled here: must not surface as the job's last error
            if (attempt == 0) drop_cache();
            else drop_other_caches(this);
            e = hipMalloc(p, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            return NGP_ERR_TOO_LARGE;
        }
        live[*p] = bytes;
        return NGP_OK;
    }
    void release(void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.emplace(it->second, p);
        cached_bytes += it->second;
        live.erase(it);
    }
    // split-k fat steps of small chunks (ngp_col_kernels.h): grow-only, owned by the context
    double *splitk_part = nullptr;
    int splitk_items = 0;
    ngp_status splitk_reserve(int items) {
        if (items <= splitk_items) return NGP_OK;
        void *a = nullptr;
        if (alloc(&a, sizeof(double) * (size_t)items * SPLITK_SLOTS * 4 * 64 * 64))
            return NGP_ERR_TOO_LARGE;
        release(splitk_part);
        splitk_part = (double *)a;
        splitk_items = items;
        return NGP_OK;
    }
    // bytes a job's factor storage may take: 3/4 of what the device has free now plus what this
    // context's cache would give back (the figure of ngp_ctx_create goes stale as soon as another
    // context or a resident factor allocates)
    void refresh_mem_cap() {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess)
            mem_cap = (size_t)(0.75 * (double)(fr + cached_bytes + ws_cap));
    }
};

// every live context of the process (alloc falls back on the others' caches)
static std::mutex g_ctx_registry_mu;
static std::vector<ngp_ctx *> g_ctx_registry;
void ngp_ctx::drop_other_caches(ngp_ctx *self) {
    std::lock_guard<std::mutex> lk(g_ctx_registry_mu);
    for (ngp_ctx *o : g_ctx_registry) {
        if (o == self || o->device != self->device) continue;
        if (o->mu.try_lock()) {   // a context in the middle of a call keeps its cache
            o->drop_cache();
            o->ws_drop();
            o->mu.unlock();
        }
    }
}

// The working storage of a run is laid out twice through the same sequence of requests: once to
// add the sizes up (the workspace is then reserved in one piece), once to take the pieces.
struct WsPlan {
    ngp_ctx *c;
    bool measuring;
    size_t total = 0;
    ngp_status operator()(void **q, size_t bytes) {
        if (measuring) {
            total += ngp_ctx::ws_round(bytes);
            *q = nullptr;
            return NGP_OK;
        }
        *q = c->ws_take(bytes);
        return *q ? NGP_OK : NGP_ERR_TOO_LARGE;
    }
};

static DevSpec dev_spec(const ngp_spec &s) {
    return DevSpec{s.se_form, s.periodic_form, s.cp_form, s.precision, s.jitter, s.mixed_tau};
}

extern "C" void ngp_default_spec(ngp_spec *s) {
    if (!s) return;
    s->se_form = 0;
    s->periodic_form = 0;
    s->cp_form = 0;
    s->precision = NGP_PREC_F64;
    s->jitter = 1e-5;
    s->mixed_tau = 1e-6;
    s->refine_tol = 1e-9;
    s->refine_max = 3;
    s->reserved = 0;
}

extern "C" const char *ngp_version(void) { return "libngp 0.1.0 (gfx950)"; }

extern "C" const char *ngp_strerror(ngp_status st) {
    switch (st) {
    case NGP_OK: return "ok";
    case NGP_ERR_ARG: return "bad argument (null pointer or negative size)";
    case NGP_ERR_PROGRAM: return "malformed kernel program";
    case NGP_ERR_TOO_LARGE: return "problem exceeds a library limit or device memory";
    case NGP_ERR_NO_DEVICE: return "no usable HIP device";
    case NGP_ERR_STATE: return "job used out of order / entry point unavailable";
    case NGP_ERR_UNAVAILABLE: return "optional component (librccl) is not available";
    default: break;
    }
    if (st > 0) return hipGetErrorString((hipError_t)st);
    return "unknown error";
}

extern "C" ngp_status ngp_kernel_check(const ngp_kernel *k) {
    DevProgram tmp;
    return compile_program(k, &tmp, nullptr);
}

extern "C" ngp_status ngp_ctx_create(int32_t device, ngp_ctx **out) {
    if (!out) return NGP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NGP_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return NGP_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    ngp_ctx *c = new (std::nothrow) ngp_ctx();
    if (!c) return NGP_ERR_TOO_LARGE;
    c->device = device;
    ngp_default_spec(&c->spec);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return (ngp_status)e;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return NGP_ERR_NO_DEVICE;
    }
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) c->mem_cap = (size_t)(0.75 * (double)fr);
    else c->mem_cap = (size_t)8 << 30;
    {
        std::lock_guard<std::mutex> lk(g_ctx_registry_mu);
        g_ctx_registry.push_back(c);
    }
    *out = c;
    return NGP_OK;
}

extern "C" void ngp_ctx_destroy(ngp_ctx *c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_ctx_registry_mu);
        g_ctx_registry.erase(std::remove(g_ctx_registry.begin(), g_ctx_registry.end(), c),
                             g_ctx_registry.end());
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->free_blocks) (void)hipFree(kv.second);
    for (auto &kv : c->live) (void)hipFree(kv.first);
    c->ws_drop();
    (void)hipStreamDestroy(c->stream);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 1; i < ngp_ctx::MAX_LANES; ++i) {
        if (c->lane_main[i]) { (void)hipStreamSynchronize(c->lane_main[i]); (void)hipStreamDestroy(c->lane_main[i]); }
        if (c->lane_side[i]) { (void)hipStreamSynchronize(c->lane_side[i]); (void)hipStreamDestroy(c->lane_side[i]); }
        for (hipEvent_t e : {c->lane_fork[i], c->lane_join[i], c->lane_done[i]})
            if (e) (void)hipEventDestroy(e);
    }
    if (c->ev_lane_go) (void)hipEventDestroy(c->ev_lane_go);
    delete c;
}

extern "C" ngp_status ngp_set_spec(ngp_ctx *c, const ngp_spec *s) {
    if (!c || !s) return NGP_ERR_ARG;
    if (s->se_form < 0 || s->se_form > 1 || s->periodic_form < 0 || s->periodic_form > 1 ||
        s->cp_form < 0 || s->cp_form > 1 || !(s->jitter >= 0.0))
        return NGP_ERR_ARG;
    if (s->precision != NGP_PREC_F64 && s->precision != NGP_PREC_MIXED) return NGP_ERR_ARG;
    if (s->precision == NGP_PREC_MIXED &&
        (!(s->mixed_tau >= 0.0) || !(s->refine_tol > 0.0) || s->refine_max < 0 || s->refine_max > 16))
        return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->spec = *s;
    return NGP_OK;
}
extern "C" ngp_status ngp_get_spec(const ngp_ctx *c, ngp_spec *s) {
    if (!c || !s) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(const_cast<ngp_ctx *>(c)->mu);
    *s = c->spec;
    return NGP_OK;
}

extern "C" ngp_status ngp_set_structured_storage(ngp_ctx *c, int32_t on) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->toeplitz = on != 0;
    return NGP_OK;
}
extern "C" ngp_status ngp_set_batch_invariant(ngp_ctx *c, int32_t on) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->invariant = on != 0;
    return NGP_OK;
}
extern "C" ngp_status ngp_set_short_series_path(ngp_ctx *c, int32_t on) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->short_series = on != 0;
    return NGP_OK;
}
extern "C" ngp_status ngp_profile_enable(ngp_ctx *c, int32_t on) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profiling = on != 0;
    return NGP_OK;
}
extern "C" ngp_status ngp_profile_reset(ngp_ctx *c) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    std::memset(&c->prof, 0, sizeof(c->prof));
    return NGP_OK;
}
extern "C" ngp_status ngp_profile_get(ngp_ctx *c, ngp_profile *out) {
    if (!c || !out) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    *out = c->prof;
    return NGP_OK;
}

// ---------------------------------------------------------------------------------------
// jobs
// ---------------------------------------------------------------------------------------
struct ngp_job {
    ngp_ctx *ctx = nullptr;
    JobGeom g{};
    int n = 0;          // base training points (n0 + tail)
    bool ran = false;
    // device buffers
    DevProgram *progs = nullptr;
    double *t0 = nullptr, *taux = nullptr, *y0 = nullptr, *ya = nullptr;
    double *logdet = nullptr, *G = nullptr, *work = nullptr, *zbuf = nullptr;
    int32_t *info = nullptr, *qpts = nullptr;
    double *logml_base = nullptr, *logml_full = nullptr, *mu = nullptr, *sigma = nullptr;
    int64_t work_stride = 0;
    std::vector<void *> owned;
    // the staging copy of the inputs (kept while small: stage_general), the result region of the
    // arena ([info | logml_base | logml_full | mu | sigma], offsets from info) and whether logdet /
    // info still hold the zeros they were staged with
    PinVec<unsigned char> h_in, h_out;
    size_t out_off[5] = {}, out_bytes = 0;
    bool zeroed = false, copy_in_flight = false;
    bool out_fetched = false;   // h_out holds the result region of the last run
    // the spec the job was staged under: a later ngp_set_spec does not reach a staged job
    ngp_spec spec{};
    // lattice jobs: items whose reduced program is a chain of more than one instruction / the other
    // items (ascending job-wide indices; device copies fill_chain_d / fill_other_d)
    std::vector<int32_t> fill_chain, fill_other, fill_single;   // single: the whole tree is ONE table
    int32_t *fill_chain_d = nullptr, *fill_other_d = nullptr, *fill_single_d = nullptr;
    // NGP_PREC_MIXED: per item, filled by ngp_job_run
    std::vector<int32_t> refine_steps;
    std::vector<double> refine_delta, frac32;
};

namespace {

struct EventTimer {  // HIP events on the launch stream, resolved after the job's final sync
    struct Rec { int cls; hipEvent_t a, b; double flops, bytes; };
    std::vector<Rec> recs;
    bool on;
    hipStream_t s;
    EventTimer(bool on_, hipStream_t s_) : on(on_), s(s_) {}
    ~EventTimer() {   // an error path left before resolve()
        for (auto &r : recs) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
    }
    // `on_stream`: the stream f launches on when it is not the timer's own (diag-ahead tiles)
    template <class F> void run(int cls, double flops, double bytes, F &&f,
                                hipStream_t on_stream = nullptr) {
        if (!on) { f(); return; }
        hipStream_t es = on_stream ? on_stream : s;
        Rec r{cls, nullptr, nullptr, flops, bytes};
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, es);
        f();
        (void)hipEventRecord(r.b, es);
        recs.push_back(r);
    }
    void resolve(ngp_profile &p) {
        for (auto &r : recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                p.ms[r.cls] += ms;
                p.launches[r.cls] += 1;
                p.flops[r.cls] += r.flops;
                p.bytes[r.cls] += r.bytes;
            }
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        recs.clear();
    }
};


// The left-looking factorisation of one chunk (every item's block columns in lock step).
// Block columns in pairs: a FAT step finishes column jj and pre-accumulates column jj+1 (and,
// on a side stream, the diagonal tile of jj+2) from the same streamed rows; the THIN step that
// follows only adds k in [64 (jj-1), 64 jj).  See chol_col_lds_kernel.
struct Lane {
    hipStream_t main, side;
    hipEvent_t fork, join;
    ngp_ctx *ctx;
};
inline Lane lane_of(ngp_ctx *c) { return Lane{c->stream, c->side, c->ev_fork, c->ev_join, c}; }

// dinv_step != 0: block column jj writes / reads its M at p.dinv + jj * dinv_step (cached factor:
// every M_j is kept); 0: one buffer reused by every step.
// sp != null and p0.L32 set: mixed-precision job (fat steps on chol_col_glds_kernel<MIXED>, class 9).
// order_buf / order_prev ([bc] each, mixed jobs): every MIXED_REORDER block columns the items are
// re-ranked by the fp64 tile products they needed since the last ranking.
constexpr size_t MAX_CHUNK_ITEMS = 65535;   // gridDim.y
// The column kernels address an item's factor storage through a buffer descriptor with 32-bit
// byte offsets: it must stay below 2 GiB (n <= 16,256 for value jobs, 11,520 for gradient jobs).
inline bool item_too_large(int64_t item_stride) { return item_stride * 8 > (int64_t)0x7fffffff; }
constexpr int MIXED_REORDER = 16;
constexpr int AHEAD_EARLY_MAX_ITEMS = 512;
// Two lanes pay from n ~ 1500 and 64 items on (measured, 64 items: n = 2048 logml 6.78 -> 6.57 ms,
// logml + gradient 17.07 -> 16.41; n = 1024: 2.00 -> 2.20 and 4.13 -> 4.21, so not there); three and
// four lanes were slower everywhere (more streams than hardware queues).
constexpr int TWO_LANE_MIN_ITEMS = 64, TWO_LANE_MIN_NB = 24;
void factor_chunk(const Lane &ln, const JobGeom &g, const ChunkPtrs &p_in, int bc, EventTimer &tm,
                  size_t dinv_step = 0, const DevSpec *sp = nullptr, int32_t *order_buf = nullptr,
                  unsigned *order_prev = nullptr, bool half = false) {
    ChunkPtrs p0 = p_in;
    const bool mixed = sp != nullptr && p0.L32 != nullptr;
    hipStream_t s = ln.main;
    // Short series: the whole sweep of an item in one launch (ngp_small_kernels.h) — a rule of the
    // geometry alone, so an item's arithmetic does not depend on its batch.  Resident factors and the
    // Toeplitz gradient path keep every M_j (dinv_step) and stay on the column sweep; so do chunks
    // that fill the chip many times over, where the column sweep's matrix-core rate wins.
    SmallPlan spl;
    if (!mixed && dinv_step == 0 && small_job(g, bc, &spl)) {
        const double nn = 16.0 * spl.nbe;
        tm.run(13, bc * small_flops(g, spl), bc * 8.0 * (nn * nn * (spl.ident ? 1.5 : 1.0) + 2.0 * g.naux * nn),
               [&] { launch_chol_small(g, p0, bc, spl, s); });
        return;
    }
    // small chunks of long series: room for the split-k fat steps (chol_col_glds_kernel<.., SPLITK>);
    // the buffer stays with the context
    if (!half && bc <= AHEAD_EARLY_MAX_ITEMS && !mixed && !g.aux_identity && g.nb0 >= 8 && !g.invariant &&
        ln.ctx->splitk_reserve(bc) == NGP_OK)
        p0.splitk_part = ln.ctx->splitk_part;
    // Small chunks of long series (the 64-particle calls of a fit) are swept as two half-chunks side
    // by side, each on its own pair of streams (lanes).  A launch of such a chunk rarely fills the chip or
    // fills it one and a fraction times (64 items at n = 2048: 576 workgroups on 512 slots at
    // column 14 — two rounds for the price of 1.1), and chol_diag / the thin step leave it almost
    // empty: the other half's fat step runs in those gaps.  Every item's arithmetic is what it
    // would be in a chunk of its half's size.
    const int nl = ngp_ctx::MAX_LANES;
    if (!half && !mixed && nl >= 2 && bc >= TWO_LANE_MIN_ITEMS && bc <= AHEAD_EARLY_MAX_ITEMS &&
        g.nb0 >= TWO_LANE_MIN_NB && ln.main == ln.ctx->stream && ln.ctx->make_lanes(nl)) {
        ngp_ctx *c = ln.ctx;
        (void)hipEventRecord(c->ev_lane_go, s);
        std::vector<EventTimer> tms;
        tms.reserve((size_t)nl);
        int h0 = 0;
        for (int i = 0; i < nl; ++i) {
            const int h1 = (int)((long)bc * (i + 1) / nl);
            ChunkPtrs pi = p0;
            pi.L += (size_t)h0 * g.item_stride;
            pi.dinv += (size_t)h0 * NB * NB;
            pi.progs += h0;
            pi.logdet += h0;
            pi.info += h0;
            if (pi.tab) pi.tab += (size_t)h0 * g.maxstat * g.R;   // read by the column kernels (toep)
            pi.n_fill_single = (int32_t)((long)p0.n_fill_single * (h1 - h0) / bc);   // byte accounting only
            if (pi.splitk_part) pi.splitk_part += (size_t)h0 * SPLITK_SLOTS * 4 * 64 * 64;
            if (i == 0) {
                factor_chunk(ln, g, pi, h1 - h0, tm, dinv_step, sp, order_buf, order_prev, true);
            } else {
                const Lane li{c->lane_main[i], c->lane_side[i], c->lane_fork[i], c->lane_join[i], c};
                tms.emplace_back(tm.on, li.main);
                (void)hipStreamWaitEvent(li.main, c->ev_lane_go, 0);
                factor_chunk(li, g, pi, h1 - h0, tms.back(), dinv_step, sp, order_buf, order_prev, true);
                (void)hipEventRecord(c->lane_done[i], li.main);
            }
            h0 = h1;
        }
        for (int i = 1; i < nl; ++i) (void)hipStreamWaitEvent(s, c->lane_done[i], 0);
        for (auto &t : tms) {
            tm.recs.insert(tm.recs.end(), t.recs.begin(), t.recs.end());
            t.recs.clear();
        }
        return;
    }
    const double nrows_aux = (double)g.naux;
    const double lazy_frac =
        (g.toep && p0.n_fill_single > 0 && !mixed) ? std::min(1.0, (double)p0.n_fill_single / bc) : 0.0;
    bool ahead_pending = false;
    // An odd number of block columns: column 0 goes alone (a FULL step without a k-loop: only the
    // solve) and the pairs start at column 1.  Pairing from column 0 leaves the LAST column alone,
    // whose FULL step carries the longest k-loop of the sweep on the direct-load kernel (gradient
    // jobs at n = 2049, 33 block columns: 62 MB of reads per item and 3.7 % of the call).
    const int o = (g.nb0 >= 3 && (g.nb0 & 1)) ? 1 : 0;
    for (int jj = 0; jj < g.nb0; ++jj) {
        const bool fat = jj >= o && ((jj - o) % 2 == 0) && (jj + 1 < g.nb0);
        const bool thin = jj >= o && ((jj - o) % 2 == 1);
        const int mode = fat ? COL_FAT : (thin ? COL_THIN : COL_FULL);
        const int ahead = (fat && jj + 2 < g.nb0) ? 1 : 0;
        const int k0_col = thin ? (jj - 1) * NB : 0;
        // diag tile (jj,jj): second column of a pair: pre-accumulated over k < 64 (jj-1) by the fat
        // step jj-1; first column of the second pair on: over k < 64 (jj-2) by its diag-ahead tile
        const int k0_diag = thin ? (jj - 1) * NB : (jj - o >= 2 ? (jj - 2) * NB : 0);
        const double k = (double)jj * NB;
        const double kd = k - k0_diag;
        ChunkPtrs p = p0;
        p.dinv = p0.dinv + (size_t)jj * dinv_step;
        // the diag-ahead tile (jj, jj) was launched on the side stream at step jj-2, beside
        // diag(jj-1) / col(jj-1); chol_diag(jj) is its only consumer
        if (ahead_pending && ((jj - o) % 2 == 0)) {
            (void)hipStreamWaitEvent(s, ln.join, 0);
            ahead_pending = false;
        }
        // Small chunks (the 64-particle calls of a fit): the diag-ahead tile of this pair goes to
        // the side stream BEFORE chol_diag / the fat step — everything it reads (rows of block
        // jj + 2, columns < 64 jj) is final once column jj - 1 is.  Beside the fat step it has
        // several hundred microseconds to hide in; launched after it (the large-batch order below)
        // its single-wave k-loop outlasts chol_diag + the thin step from n ~ 1500 on and
        // chol_diag(jj + 2) waits for it.  The order of launches does not change any result.
        const bool ahead_early = bc <= AHEAD_EARLY_MAX_ITEMS;
        auto launch_ahead = [&] {
            (void)hipEventRecord(ln.fork, s);
            (void)hipStreamWaitEvent(ln.side, ln.fork, 0);
            // class 8: on the side stream
            tm.run(8, bc * (double)NB * NB * k, bc * 8.0 * NB * k,
                   [&] { launch_diag_ahead(g, p0, bc, jj, ln.side); }, ln.side);
            (void)hipEventRecord(ln.join, ln.side);
            ahead_pending = true;
        };
        if (ahead && jj > 0 && ahead_early) launch_ahead();
        tm.run(1, bc * ((double)NB * NB * kd + (double)NB * NB * NB / 3.0),
               bc * 8.0 * (NB * kd + 2.0 * NB * NB),
               [&] { launch_chol_diag(g, p, bc, jj, k0_diag, s); });
        // rows that take part and the k-products they carry.  Gradient jobs (aux rows [I ; y']):
        // identity tile a joins from block column a on and its k-loop starts at 64 a — the kernels
        // skip the rest, so it is not counted either
        double rows = (double)(g.n0 - (jj + 1) * NB) + nrows_aux;
        const double kc = k - k0_col;
        double rows_kc = rows * kc;
        if (g.aux_identity) {
            rows = (double)(g.n0 - (jj + 1) * NB) + (double)(g.naux - g.n0);   // main rows + y'
            rows_kc = rows * kc;
            for (int a = 0; a <= jj && a < g.nb0; ++a) {
                rows += NB;
                rows_kc += NB * std::max(0.0, k - std::max((double)k0_col, (double)a * NB));
            }
        }
        double fl = 2.0 * NB * rows_kc + rows * (double)NB * NB;
        double by = 8.0 * (rows_kc + NB * kc + 2.0 * rows * NB);
        if (fat) {  // + column jj+1 partial sums from the same rows
            fl += 2.0 * NB * rows_kc;
            by += 8.0 * (NB * k + 2.0 * rows * NB);
        }
        // Toeplitz jobs: the first step that touches a main tile of a single-table item reads 127
        // table entries instead of the stored tile (the sibling wave of the first row tile works
        // on the stored diagonal tile)
        if (lazy_frac > 0.0 && (fat || (mode == COL_FULL && jj == 0))) {
            const double rm = (double)(g.n0 - (jj + 1) * NB);
            by -= lazy_frac * 8.0 * NB * (rm + (fat ? std::max(0.0, rm - NB) : 0.0));
        }
        // class 0: the LDS-DMA kernel of the fat steps (the dominant kernel, the roofline figure);
        // class 12: its gradient-geometry instantiation (aux rows [I ; y'], <.., IDENT>: another
        // kernel with its own flops, bytes and rate); class 6: the direct-load kernel of the thin /
        // full steps
        tm.run(fat ? (mixed ? 9 : (g.aux_identity ? 12 : 0)) : 6, bc * fl, bc * by,
               [&] { launch_chol_col(g, p, bc, jj, mode, k0_col, s, sp); });
        if (mixed && fat && order_buf && jj >= 8 && jj % MIXED_REORDER == 8) {
            // p0.order stays null (dispatch order) unless the ranking was really launched:
            // order_buf comes from the caching allocator uninitialised
            if (launch_mixed_order(p0, order_prev, order_buf, bc, s)) p0.order = order_buf;
        }
        // large chunks: beside chol_diag(jj+1) / the thin step of jj+1 (beside the fat step it
        // cost more: profiles/r02/README.md)
        if (ahead && jj > 0 && !ahead_early) launch_ahead();
    }
    if (ahead_pending) {
        (void)hipStreamWaitEvent(s, ln.join, 0);
        ahead_pending = false;
    }
}

// stage_general keeps a staging buffer up to this size with the job instead of waiting for the copy;
// ngp_job_fetch brings a result region up to FETCH_PACKED_BYTES back in one copy
constexpr size_t STAGE_KEEP_BYTES = (size_t)4 << 20, FETCH_PACKED_BYTES = (size_t)1 << 20;

template <class T> ngp_status job_alloc(ngp_job *j, T **p, size_t count) {
    void *v = nullptr;
    ngp_status st = j->ctx->alloc(&v, count * sizeof(T));
    if (st) return st;
    j->owned.push_back(v);
    *p = (T *)v;
    return NGP_OK;
}

// Stage a job in its general form: P kernels; base data (t[n], y [P or 1][n]); d appended
// times; D scenarios y_add [(P or 1)][D][d]; m forecast times.
// yrows (combined calls, ngp_combine below): item b's observations are yrows[b][0..n) — the rows of
// several callers' arrays, never copied into one matrix on the host; y / ldy are then not read.
ngp_status stage_general(ngp_ctx *c, int P, const ngp_kernel *kernels, int n, const double *t,
                         const double *y, int64_t ldy, int d, const double *t_add, int D,
                         const double *y_add, int64_t ld_yadd_item, int m, const double *t_new,
                         int noise_on_new, ngp_job **out, const double *const *yrows = nullptr) {
    if (!c || !out || !kernels || P <= 0 || n < 0 || d < 0 || m < 0 || D <= 0) return NGP_ERR_ARG;
    if (n + d <= 0) return NGP_ERR_ARG;
    if ((n > 0 && (!t || (!y && !yrows))) || (d > 0 && (!t_add || !y_add)) || (m > 0 && !t_new))
        return NGP_ERR_ARG;
    *out = nullptr;
    auto yrow = [&](int b) { return yrows ? yrows[b] : y + (int64_t)b * ldy; };
    std::vector<DevProgram> hp((size_t)P);
    int maxstat = 0, maxcp = 0;
    for (int i = 0; i < P; ++i) {
        int ns = 0, nc = 0, nt = 0;
        ngp_status st = compile_program(&kernels[i], &hp[(size_t)i], nullptr, &ns, &nc, &nt);
        if (st) return st;
        maxstat = std::max(maxstat, nt);   // value job: one table per maximal stationary subtree
        maxcp = std::max(maxcp, nc);
    }
    JobGeom g{};
    g.B = P;
    g.n0 = (n / NB) * NB;
    g.nb0 = g.n0 / NB;
    g.tail = n - g.n0;
    g.da = g.tail + d;
    g.d = d;
    g.m = m;
    g.naux = g.da + m + 1;
    if (g.naux > NGP_MAX_AUX) return NGP_ERR_TOO_LARGE;
    g.naux_pad = (g.naux + NB - 1) / NB * NB;
    g.D = D;
    g.noise_on_new = noise_on_new ? 1 : 0;
    g.y_shared = (ldy == 0 && !yrows) ? 1 : 0;
    g.ld = g.n0;
    g.item_stride = (int64_t)(g.n0 + g.naux_pad) * g.n0;
    if (item_too_large(g.item_stride)) return NGP_ERR_TOO_LARGE;
    g.npts = g.n0 + g.da + m;
    g.n_real = g.n0;
    g.aux_identity = 0;
    g.maxstat = std::max(maxstat, 1);
    g.maxcp = std::max(maxcp, 1);
    std::vector<int32_t> h_q;
    int32_t toep_stride = 0;
    if (g.n0 > 0) {
        std::vector<double> allt((size_t)g.npts);
        for (int i = 0; i < n; ++i) allt[(size_t)i] = t[i];
        for (int a = 0; a < d; ++a) allt[(size_t)(n + a)] = t_add[a];
        for (int i = 0; i < m; ++i) allt[(size_t)(n + d + i)] = t_new[i];
        double hh = 0.0;
        int R = 0;
        if (detect_lattice(allt, &hh, &h_q, &R)) {
            g.lattice = 1;
            g.h = hh;
            g.R = R;
            // the main block at a constant lattice stride (a regular series — every one the
            // reference's tests and vignettes fit): K of a stationary tree is Toeplitz there and
            // its off-diagonal tiles are never stored (JobGeom::toep).  fp64 value jobs with at
            // least one tile below the diagonal; mixed precision keeps every tile (its shadow
            // copies and tile maxima come from the stored rows).
            if (g.nb0 >= 2) {
                const long st0 = (long)h_q[1] - (long)h_q[0];
                bool reg = st0 != 0;
                for (int i = 2; i < g.n0 && reg; ++i)
                    reg = (long)h_q[(size_t)i] - (long)h_q[(size_t)i - 1] == st0;
                if (reg) toep_stride = (int32_t)std::labs(st0);
            }
        }
    }

    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    g.invariant = c->invariant ? 1 : 0;
    g.short_series = c->short_series ? 1 : 0;
    // (short series are factorised from registers in one launch and store every tile)
    if (c->toeplitz && c->spec.precision != NGP_PREC_MIXED &&
        !(g.short_series && g.nb0 <= 4 && (P <= SM_MAX_ITEMS || g.invariant)))
        g.toep = toep_stride;
    ngp_job *j = new (std::nothrow) ngp_job();
    if (!j) return NGP_ERR_TOO_LARGE;
    j->ctx = c;
    j->g = g;
    j->n = n;
    j->spec = c->spec;
    ngp_status st = NGP_OK;
    auto fail = [&](ngp_status s) {
        for (void *p : j->owned) c->release(p);
        delete j;
        return s;
    };
    const int ny = g.y_shared ? 1 : P;
    // the fill kernel of every item: one table for the whole tree (in a Toeplitz job the
    // structured items, prog_structure) / chain programs / the rest
    if (g.lattice && g.n0 > 0)
        for (int i = 0; i < P; ++i)
            (prog_single_table(&hp[(size_t)i]) ? j->fill_single
                                                : hp[(size_t)i].rchain ? j->fill_chain : j->fill_other)
                .push_back(i);
    // ONE device arena for everything that crosses the bus, inputs first, outputs last:
    //   [programs | t0 | taux | y0 | ya | lattice indices | fill lists | logdet | info |
    //    logml_base | logml_full | mu | sigma]
    // The inputs go up in ONE copy (logdet and info arrive as the zeros the first run expects) and
    // a small job's results come back in one (ngp_job_fetch): the 24- and 64-particle calls of a
    // fit on a short series are chains of dependent launches a few microseconds long, and every
    // separate copy or memset was one more link (ten copies and two memsets before).
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(std::max<size_t>(bytes, 8)); return o; };
    const size_t n_t0 = (size_t)std::max(g.n0, 1), n_taux = (size_t)std::max(g.da + m, 1),
                 n_y0 = (size_t)std::max(ny * g.n0, 1),
                 n_ya = (size_t)std::max((int64_t)ny * D * g.da, (int64_t)1);
    const size_t o_progs = take(sizeof(DevProgram) * (size_t)P), o_t0 = take(8 * n_t0),
                 o_taux = take(8 * n_taux), o_y0 = take(8 * n_y0), o_ya = take(8 * n_ya),
                 o_q = take(g.lattice ? 4 * (size_t)g.npts : 0),
                 o_fs = take(4 * j->fill_single.size()), o_fc = take(4 * j->fill_chain.size()),
                 o_fo = take(4 * j->fill_other.size()), o_logdet = take(8 * (size_t)P),
                 o_info = take(4 * (size_t)P);
    const size_t in_bytes = off;
    const size_t o_lb = take(8 * (size_t)P), o_lf = take(8 * (size_t)P * D),
                 o_mu = take(m > 0 ? 8 * (size_t)P * D * m : 0),
                 o_sigma = take(m > 0 ? 8 * (size_t)P * m * m : 0);
    j->h_in.assign(in_bytes, 0);
    {
        unsigned char *h = j->h_in.data();
        std::memcpy(h + o_progs, hp.data(), sizeof(DevProgram) * (size_t)P);
        if (g.n0 > 0) std::memcpy(h + o_t0, t, 8 * (size_t)g.n0);
        double *h_taux = (double *)(h + o_taux), *h_y0 = (double *)(h + o_y0),
               *h_ya = (double *)(h + o_ya);
        for (int a = 0; a < g.tail; ++a) h_taux[a] = t[g.n0 + a];
        for (int a = 0; a < d; ++a) h_taux[g.tail + a] = t_add[a];
        for (int i = 0; i < m; ++i) h_taux[g.da + i] = t_new[i];
        for (int b = 0; b < ny; ++b)
            if (g.n0 > 0) std::memcpy(h_y0 + (size_t)b * g.n0, yrow(b), 8 * (size_t)g.n0);
        for (int b = 0; b < ny; ++b)
            for (int sc = 0; sc < D; ++sc) {
                double *dst = h_ya + ((size_t)b * D + sc) * g.da;
                for (int a = 0; a < g.tail; ++a) dst[a] = yrow(b)[g.n0 + a];
                for (int a = 0; a < d; ++a)
                    dst[g.tail + a] = y_add[(int64_t)b * ld_yadd_item + (int64_t)sc * d + a];
            }
        if (g.lattice) std::memcpy(h + o_q, h_q.data(), 4 * (size_t)g.npts);
        if (!j->fill_single.empty())
            std::memcpy(h + o_fs, j->fill_single.data(), 4 * j->fill_single.size());
        if (!j->fill_chain.empty())
            std::memcpy(h + o_fc, j->fill_chain.data(), 4 * j->fill_chain.size());
        if (!j->fill_other.empty())
            std::memcpy(h + o_fo, j->fill_other.data(), 4 * j->fill_other.size());
    }
    unsigned char *io = nullptr;
    if ((st = job_alloc(j, &io, off))) return fail(st);
    j->progs = (DevProgram *)(io + o_progs);
    j->t0 = (double *)(io + o_t0);
    j->taux = (double *)(io + o_taux);
    j->y0 = (double *)(io + o_y0);
    j->ya = (double *)(io + o_ya);
    if (g.lattice) j->qpts = (int32_t *)(io + o_q);
    if (g.lattice && g.n0 > 0) {
        j->fill_single_d = (int32_t *)(io + o_fs);
        j->fill_chain_d = (int32_t *)(io + o_fc);
        j->fill_other_d = (int32_t *)(io + o_fo);
    }
    j->logdet = (double *)(io + o_logdet);
    j->info = (int32_t *)(io + o_info);
    j->logml_base = (double *)(io + o_lb);
    j->logml_full = (double *)(io + o_lf);
    if (m > 0) {
        j->mu = (double *)(io + o_mu);
        j->sigma = (double *)(io + o_sigma);
    }
    j->out_off[0] = 0;                     // offsets inside the result region, which starts at info
    j->out_off[1] = o_lb - o_info;
    j->out_off[2] = o_lf - o_info;
    j->out_off[3] = o_mu - o_info;
    j->out_off[4] = o_sigma - o_info;
    j->out_bytes = off - o_info;
    if ((st = job_alloc(j, &j->G, (size_t)P * g.naux * g.naux))) return fail(st);
    j->work_stride = (int64_t)g.da * g.da + (int64_t)m * g.da + g.da + 8;
    if ((st = job_alloc(j, &j->work, (size_t)P * j->work_stride))) return fail(st);
    if ((st = job_alloc(j, &j->zbuf, (size_t)std::max((int64_t)P * D * g.da, (int64_t)1))))
        return fail(st);
    hipStream_t s = c->stream;
    if (hipMemcpyAsync(io, j->h_in.data(), in_bytes, hipMemcpyHostToDevice, s) != hipSuccess)
        return fail(NGP_ERR_STATE);
    j->zeroed = true;
    j->copy_in_flight = true;
    // a small staging buffer stays with the job (no wait here: the run is queued right behind the
    // copy); a large one is given back once the copy has left it
    if (in_bytes > STAGE_KEEP_BYTES) {
        if (hipStreamSynchronize(s) != hipSuccess) return fail(NGP_ERR_STATE);
        PinVec<unsigned char>().swap(j->h_in);
        j->copy_in_flight = false;
    }
    *out = j;
    return NGP_OK;
}

}  // namespace

namespace {

// Device buffers of the Gram refinement of one chunk (NGP_PREC_MIXED).
struct RefineBufs {
    double *X = nullptr, *A = nullptr, *R = nullptr;   // [Bc][naux_pad][n0]
    double *S = nullptr, *T = nullptr;                 // [Bc][naux^2]
    double *U = nullptr;                               // [Bc][naux]
    double *delta = nullptr;                           // [Bc][2]
    int32_t *items = nullptr;                          // [Bc] items a later step still works on
};

// G <- X K^-1 X' refined against the fp64 covariance (see kapply_kernel).  On entry the aux rows
// of the slab hold W = X L^-T and p.dinv every block inverse M_j ([nb0][mstep]).  steps / delta
// (host, [bc]) receive the per-item step count and last correction; returns the items that did
// not reach `tol` within `max_steps` in not_refined.
ngp_status refine_chunk(ngp_ctx *c, const JobGeom &g, const ChunkPtrs &p, const RefineBufs &rb,
                        double *Gchunk, int bc, size_t mstep, const ngp_spec &spec,
                        const DevSpec &sp, EventTimer &tm, int32_t *steps, double *delta_out,
                        std::vector<int> *not_refined) {
    hipStream_t s = c->stream;
    const double naux = (double)g.naux, n0 = (double)g.n0;
    ChunkPtrs pf = p;          // the sweeps are fp64: no shadow rows, no tile maxima
    pf.L32 = nullptr;
    pf.tmax = nullptr;
    pf.mixcnt = nullptr;
    pf.auxX = nullptr;
    int na = bc;               // items still being refined (pf.items: their indices; null = all)
    auto backward = [&](int accumulate) {
        for (int cc = g.nb0 - 1; cc >= 0; --cc)
            tm.run(10, na * naux * 2.0 * NB * (double)(cc + 1) * NB,
                   na * 8.0 * ((double)NB * NB * (cc + 1) + 2.0 * naux * NB * (cc + 1)), [&] {
                       launch_aux_back(g, pf, p.dinv, mstep, rb.A, accumulate, na, cc, s);
                   });
    };
    backward(0);               // A_0 = W L^-1
    std::vector<double> dprev((size_t)bc, 1.0), dboth(2 * (size_t)bc, 0.0);
    std::vector<int32_t> active((size_t)bc);
    for (int i = 0; i < bc; ++i) { steps[i] = 0; delta_out[i] = 0.0; active[(size_t)i] = i; }
    for (int it = 1;; ++it) {
        tm.run(10, na * 2.0 * naux * n0 * n0, na * 8.0 * 3.0 * naux * n0,
               [&] { launch_kapply(g, pf, rb.A, rb.X, rb.R, na, sp, s); });
        tm.run(10, na * 4.0 * naux * naux * n0, na * 8.0 * 3.0 * naux * n0, [&] {
            launch_refine_gram(g, rb.A, rb.X, rb.R, rb.S, rb.T, rb.U, Gchunk, rb.delta, na,
                               pf.items, s);
        });
        HIPCHK(hipMemcpyAsync(dboth.data(), rb.delta, sizeof(double) * 2 * (size_t)bc,
                              hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int32_t> still;
        for (int i : active) {
            const double d = dboth[2 * (size_t)i];
            // What is left after this step's correction d is about d * rho, rho the contraction
            // of the iteration: measured as d / d_previous from the second step on, and from the
            // residual itself (max_a |R_a| / |X_a|) at the first, whichever is larger.
            double rho = dboth[2 * (size_t)i + 1];
            if (it > 1) rho = std::max(rho, d / dprev[(size_t)i]);
            const double est = d * std::min(rho, 1.0);
            steps[i] = it;
            delta_out[i] = d;
            dprev[(size_t)i] = d;
            if (!(est <= spec.refine_tol)) still.push_back(i);   // NaN stays
        }
        active.swap(still);
        if (active.empty() || it >= spec.refine_max) break;
        // the next step only touches the items that are not there yet
        na = (int)active.size();
        HIPCHK(hipMemcpy(rb.items, active.data(), 4 * (size_t)na, hipMemcpyHostToDevice));
        pf.items = rb.items;
        // A += (R L^-T) L^-1: R into the aux rows, forward sweep (the resident-factor kernels),
        // backward sweep accumulating into A
        for (int i : active)
            HIPCHK(hipMemcpyAsync(p.L + (size_t)i * g.item_stride + (size_t)g.n0 * g.ld,
                                  rb.R + (size_t)i * g.naux_pad * g.n0,
                                  (size_t)g.naux_pad * g.n0 * 8, hipMemcpyDeviceToDevice, s));
        for (int jj = 0; jj < g.nb0; ++jj) {
            ChunkPtrs pj = pf;
            pj.dinv = p.dinv + (size_t)jj * mstep;
            tm.run(10, na * naux * (double)NB * NB, na * 8.0 * 3.0 * naux * NB,
                   [&] { launch_chol_col(g, pj, na, jj, COL_AUX, jj * NB, s); });
            tm.run(10, na * naux * 2.0 * NB * (double)(g.n0 - (jj + 1) * NB),
                   na * 8.0 * (g.n0 - (jj + 1) * NB) * (2.0 * naux + NB),
                   [&] { launch_aux_update(g, pj, na, jj, s); });
        }
        backward(1);
    }
    for (int i : active) not_refined->push_back(i);
    return NGP_OK;
}

}  // namespace

extern "C" ngp_status ngp_job_run(ngp_job *j) {
    if (!j) return NGP_ERR_ARG;
    ngp_ctx *c = j->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const JobGeom &g = j->g;
    hipStream_t s = c->stream;
    const DevSpec sp = dev_spec(j->spec);
    EventTimer tm(c->profiling, s);
    // a re-run: the first one finds the zeros the staging copy brought.  (Short jobs: chol_small_kernel
    // writes both outright; whatever the chunk size turns out to be, it is at most the batch.)
    if (!j->zeroed && !(g.n0 > 0 && small_job(g, g.B) && j->spec.precision != NGP_PREC_MIXED)) {
        HIPCHK(hipMemsetAsync(j->logdet, 0, sizeof(double) * (size_t)g.B, s));
        HIPCHK(hipMemsetAsync(j->info, 0, sizeof(int32_t) * (size_t)g.B, s));
    }
    j->zeroed = false;
    // mixed precision needs at least one fat step (two block columns); shorter series run fp64, and
    // so do series of more than 129 block columns (n > 8,319): a fat step classifies its k-tiles
    // in two 64-bit masks
    const bool mixed = j->spec.precision == NGP_PREC_MIXED && g.nb0 >= 2 && g.nb0 <= 129 &&
                       !g.aux_identity;
    const bool refine = mixed && j->spec.refine_max > 0;
    j->refine_steps.assign((size_t)g.B, 0);
    j->refine_delta.assign((size_t)g.B, 0.0);
    j->frac32.assign((size_t)g.B, 0.0);
    // the run's working storage comes from the context's workspace (ngp_ctx::ws_reserve): nothing
    // to give back on the exit paths, the next call overwrites it
    void *Lbuf = nullptr, *dinv = nullptr, *tab = nullptr, *sig = nullptr;
    void *L32 = nullptr, *tmx = nullptr, *cnt = nullptr, *order_buf = nullptr, *order_prev = nullptr;
    RefineBufs rb;
    bool single_chunk = false;
    if (g.n0 > 0) {
        const size_t tab_bytes = g.lattice ? sizeof(double) * (size_t)g.maxstat * g.R : 0;
        const size_t sig_bytes = g.lattice ? sizeof(double) * (size_t)g.maxcp * g.npts : 0;
        const size_t l_bytes = (size_t)g.item_stride * sizeof(double);
        const size_t nbt = (size_t)g.nb0 + (size_t)g.naux_pad / NB;
        const size_t aux_bytes = sizeof(double) * (size_t)g.naux_pad * g.n0;
        size_t item_bytes = l_bytes + tab_bytes + sig_bytes + sizeof(double) * NB * NB;
        if (mixed)
            item_bytes += l_bytes / 2 + 4 * nbt * g.nb0 + aux_bytes + 16 +
                          (refine ? sizeof(double) * NB * NB * (size_t)g.nb0 + 2 * aux_bytes +
                                        16 * (size_t)g.naux * g.naux + 8 * (size_t)g.naux + 20
                                  : 0);
        // as few chunks as the memory allows, of equal size (a short last chunk runs every launch
        // of the sweep again for a fraction of the items)
        // (several kernels index the items of a chunk with blockIdx.y: at most 65,535 of them)
        if ((size_t)g.B * item_bytes > c->mem_cap) c->refresh_mem_cap();   // large job: today's figure
        const size_t bc_max = std::min<size_t>(
            std::min<size_t>((size_t)g.B, MAX_CHUNK_ITEMS), std::max<size_t>(1, c->mem_cap / item_bytes));
        size_t nchunks = ((size_t)g.B + bc_max - 1) / bc_max;
        int Bc = 0;
        size_t mstep = 0;
        ngp_status st = NGP_OK;
        // the estimate above leaves the allocator's rounding and other live handles out: when an
        // allocation fails the job is cut into more chunks instead of being refused
        auto take_all = [&](WsPlan &dalloc) -> ngp_status {
            ngp_status st = dalloc(&Lbuf, l_bytes * (size_t)Bc);
            if (!st) st = dalloc(&dinv, sizeof(double) * (size_t)Bc * NB * NB * (refine ? g.nb0 : 1));
            if (!st && g.lattice) st = dalloc(&tab, tab_bytes * (size_t)Bc);
            if (!st && g.lattice) st = dalloc(&sig, sig_bytes * (size_t)Bc);
            if (!st && mixed) {
                st = dalloc(&L32, (l_bytes / 2) * (size_t)Bc);
                if (!st) st = dalloc(&tmx, 4 * nbt * g.nb0 * (size_t)Bc);
                if (!st) st = dalloc(&cnt, 8 * (size_t)Bc);
                if (!st) st = dalloc(&order_buf, 4 * (size_t)Bc);
                if (!st) st = dalloc(&order_prev, 4 * (size_t)Bc);
                if (!st) st = dalloc((void **)&rb.X, aux_bytes * (size_t)Bc);
            }
            if (!st && refine) {
                st = dalloc((void **)&rb.A, aux_bytes * (size_t)Bc);
                if (!st) st = dalloc((void **)&rb.R, aux_bytes * (size_t)Bc);
                if (!st) st = dalloc((void **)&rb.S, 8 * (size_t)g.naux * g.naux * Bc);
                if (!st) st = dalloc((void **)&rb.T, 8 * (size_t)g.naux * g.naux * Bc);
                if (!st) st = dalloc((void **)&rb.U, 8 * (size_t)g.naux * Bc);
                if (!st) st = dalloc((void **)&rb.delta, 16 * (size_t)Bc);
                if (!st) st = dalloc((void **)&rb.items, 4 * (size_t)Bc);
            }
            return st;
        };
        for (;; nchunks *= 2) {
            Bc = (int)(((size_t)g.B + nchunks - 1) / nchunks);
            // refinement sweeps need every block inverse M_j, not only the current one
            mstep = refine ? (size_t)Bc * NB * NB : 0;
            WsPlan measure{c, true}, take{c, false};
            (void)take_all(measure);
            st = c->ws_reserve(measure.total);
            if (!st) st = take_all(take);
            if (st != NGP_ERR_TOO_LARGE || Bc <= 1) break;
        }
        single_chunk = Bc >= g.B;
        if (st) return st;
        const Lane ln = lane_of(c);
        const double nrows_aux = (double)g.naux;
        for (int b0 = 0; b0 < g.B; b0 += Bc) {
            const int bc = std::min(Bc, g.B - b0);
            ChunkPtrs p{};
            p.L = (double *)Lbuf;
            p.dinv = (double *)dinv;
            p.progs = j->progs + b0;
            p.t0 = j->t0;
            p.taux = j->taux;
            p.y0 = j->y0 + (g.y_shared ? 0 : (int64_t)b0 * g.n0);
            p.logdet = j->logdet + b0;
            p.info = j->info + b0;
            p.tab = (double *)tab;
            p.sig = (double *)sig;
            p.qpts = j->qpts;
            if (j->fill_other_d) {   // the chunk's share of the two item lists
                auto range = [&](const std::vector<int32_t> &v, const int32_t *dev,
                                 const int32_t **ptr, int32_t *cnt) {
                    const auto lo = std::lower_bound(v.begin(), v.end(), b0);
                    const auto hi = std::lower_bound(v.begin(), v.end(), b0 + bc);
                    *ptr = dev + (lo - v.begin());
                    *cnt = (int32_t)(hi - lo);
                };
                range(j->fill_chain, j->fill_chain_d, &p.fill_chain, &p.n_fill_chain);
                range(j->fill_other, j->fill_other_d, &p.fill_other, &p.n_fill_other);
                range(j->fill_single, j->fill_single_d, &p.fill_single, &p.n_fill_single);
                p.fill_base = b0;
            }
            if (mixed) {
                p.L32 = (float *)L32;
                p.tmax = (float *)tmx;
                p.mixcnt = (unsigned *)cnt;
                p.auxX = rb.X;
                HIPCHK(hipMemsetAsync(cnt, 0, 8 * (size_t)bc, s));
            }
            if (g.lattice)
                tm.run(4, 0.0, 0.0, [&] { launch_tables(g, p, bc, sp, s); });
            // Toeplitz jobs: single-table items store their diagonal tiles and aux rows only
            const double fill_elems =
                (double)bc * ((double)g.n0 * (g.n0 + NB) / 2.0 + nrows_aux * g.n0) -
                (g.toep ? (double)p.n_fill_single * ((double)g.n0 * (g.n0 - NB) / 2.0) : 0.0);
            tm.run(4, 0.0, 8.0 * fill_elems, [&] { launch_fill(g, p, bc, sp, s); });
            if (mixed) HIPCHK(hipMemsetAsync(order_prev, 0, 4 * (size_t)bc, s));
            double *Gchunk = j->G + (int64_t)b0 * g.naux * g.naux;
            // short jobs: the one launch of the factorisation leaves G too
            const bool gram_inside = !mixed && mstep == 0 && small_job(g, bc);
            if (gram_inside) p.G = Gchunk;
            factor_chunk(ln, g, p, bc, tm, mstep, mixed ? &sp : nullptr, (int32_t *)order_buf,
                         (unsigned *)order_prev);
            if (!gram_inside)
                tm.run(2, bc * nrows_aux * nrows_aux * g.n0, bc * 8.0 * nrows_aux * g.n0,
                       [&] { launch_gram(g, (const double *)Lbuf, Gchunk, bc, s); });
            if (mixed) {
                std::vector<unsigned> hc(2 * (size_t)bc);
                HIPCHK(hipMemcpyAsync(hc.data(), cnt, 8 * (size_t)bc, hipMemcpyDeviceToHost, s));
                std::vector<int> bad;
                if (refine) {
                    ngp_status rs = refine_chunk(c, g, p, rb, Gchunk, bc, mstep, j->spec, sp, tm,
                                                 j->refine_steps.data() + b0,
                                                 j->refine_delta.data() + b0, &bad);
                    if (rs) { tm.resolve(c->prof); return rs; }
                } else {
                    HIPCHK(hipStreamSynchronize(s));
                }
                for (int i = 0; i < bc; ++i) {
                    const double a = hc[2 * (size_t)i], b = hc[2 * (size_t)i + 1];
                    j->frac32[(size_t)(b0 + i)] = (a + b) > 0.0 ? a / (a + b) : 0.0;
                }
                if (!bad.empty()) {   // keep a pivot failure if the factorisation reported one
                    std::vector<int32_t> hi((size_t)bc);
                    HIPCHK(hipMemcpy(hi.data(), p.info, 4 * (size_t)bc, hipMemcpyDeviceToHost));
                    for (int i : bad)
                        if (hi[(size_t)i] == 0) hi[(size_t)i] = NGP_INFO_NOT_REFINED;
                    HIPCHK(hipMemcpy(p.info, hi.data(), 4 * (size_t)bc, hipMemcpyHostToDevice));
                }
            }
        }
    }
    EpiPtrs e{};
    e.progs = j->progs;
    e.taux = j->taux;
    e.G = j->G;
    e.ya = j->ya;
    e.logdet = j->logdet;
    e.info = j->info;
    e.work = j->work;
    e.zbuf = j->zbuf;
    e.logml_base = j->logml_base;
    e.logml_full = j->logml_full;
    e.mu = j->mu;
    e.sigma = j->sigma;
    e.work_stride = j->work_stride;
    // (batch-invariant jobs evaluate the small Schur blocks directly, whatever the number of chunks)
    if (g.lattice && single_chunk && !g.invariant) {   // the tables of the only chunk are still in place
        e.tab = (const double *)tab;
        e.sig = (const double *)sig;
        e.qpts = j->qpts;
    }
    tm.run(3, 0.0, 0.0, [&] { launch_epilogue(g, e, sp, s); });
    // small results travel behind the last kernel, in the same queue: a fetch after the run's
    // synchronisation would be a second host round trip (25 us of a 24-item call: the copy started
    // that long after the epilogue had ended)
    j->out_fetched = false;
    if (j->out_bytes <= FETCH_PACKED_BYTES) {
        j->h_out.resize(j->out_bytes);
        if (hipMemcpyAsync(j->h_out.data(), j->info, j->out_bytes, hipMemcpyDeviceToHost, s) == hipSuccess)
            j->out_fetched = true;
    }
    hipError_t err = hipStreamSynchronize(s);
    if (err == hipSuccess) err = hipGetLastError();
    tm.resolve(c->prof);
    if (err != hipSuccess) return (ngp_status)err;
    j->copy_in_flight = false;
    j->ran = true;
    return NGP_OK;
}

extern "C" ngp_status ngp_job_mixed_stats(ngp_job *j, int32_t *refine_steps, double *refine_delta,
                                          double *frac_f32) {
    if (!j) return NGP_ERR_ARG;
    if (!j->ran) return NGP_ERR_STATE;
    for (int i = 0; i < j->g.B; ++i) {
        if (refine_steps) refine_steps[i] = j->refine_steps[(size_t)i];
        if (refine_delta) refine_delta[i] = j->refine_delta[(size_t)i];
        if (frac_f32) frac_f32[i] = j->frac32[(size_t)i];
    }
    return NGP_OK;
}

extern "C" ngp_status ngp_job_fetch(ngp_job *j, double *logml_base, double *logml_full, double *mu,
                                    double *sigma, int32_t *info) {
    if (!j) return NGP_ERR_ARG;
    if (!j->ran) return NGP_ERR_STATE;
    ngp_ctx *c = j->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const JobGeom &g = j->g;
    hipStream_t s = c->stream;
    if (j->out_bytes <= FETCH_PACKED_BYTES) {   // small results: one copy of the whole region
        if (!j->out_fetched) {                  // (normally ngp_job_run has brought it already)
            j->h_out.resize(j->out_bytes);
            HIPCHK(hipMemcpyAsync(j->h_out.data(), j->info, j->out_bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            j->copy_in_flight = false;
        }
        const unsigned char *h = j->h_out.data();
        if (info) std::memcpy(info, h + j->out_off[0], 4 * (size_t)g.B);
        if (logml_base) std::memcpy(logml_base, h + j->out_off[1], 8 * (size_t)g.B);
        if (logml_full) std::memcpy(logml_full, h + j->out_off[2], 8 * (size_t)g.B * g.D);
        if (mu && g.m > 0) std::memcpy(mu, h + j->out_off[3], 8 * (size_t)g.B * g.D * g.m);
        if (sigma && g.m > 0) std::memcpy(sigma, h + j->out_off[4], 8 * (size_t)g.B * g.m * g.m);
        return NGP_OK;
    }
    if (logml_base)
        HIPCHK(hipMemcpyAsync(logml_base, j->logml_base, sizeof(double) * (size_t)g.B,
                              hipMemcpyDeviceToHost, s));
    if (logml_full)
        HIPCHK(hipMemcpyAsync(logml_full, j->logml_full, sizeof(double) * (size_t)g.B * g.D,
                              hipMemcpyDeviceToHost, s));
    if (mu && g.m > 0)
        HIPCHK(hipMemcpyAsync(mu, j->mu, sizeof(double) * (size_t)g.B * g.D * g.m,
                              hipMemcpyDeviceToHost, s));
    if (sigma && g.m > 0)
        HIPCHK(hipMemcpyAsync(sigma, j->sigma, sizeof(double) * (size_t)g.B * g.m * g.m,
                              hipMemcpyDeviceToHost, s));
    if (info)
        HIPCHK(hipMemcpyAsync(info, j->info, sizeof(int32_t) * (size_t)g.B, hipMemcpyDeviceToHost,
                              s));
    HIPCHK(hipStreamSynchronize(s));
    j->copy_in_flight = false;
    return NGP_OK;
}

extern "C" void ngp_job_destroy(ngp_job *j) {
    if (!j) return;
    {
        std::lock_guard<std::mutex> lk(j->ctx->mu);
        // staged but never fetched: the staging copy may still be reading j->h_in
        if (j->copy_in_flight && hipSetDevice(j->ctx->device) == hipSuccess)
            (void)hipStreamSynchronize(j->ctx->stream);
        for (void *p : j->owned) j->ctx->release(p);
    }
    delete j;
}

// ---------------------------------------------------------------------------------------
// staged + one-shot entry points
// ---------------------------------------------------------------------------------------
extern "C" ngp_status ngp_logml_stage(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                                      const double *t, const double *y, int64_t ldy,
                                      ngp_job **job) {
    if (n <= 0) return NGP_ERR_ARG;
    static const double dummy = 0.0;
    return stage_general(c, B, k, n, t, y, ldy, 0, &dummy, 1, &dummy, 0, 0, nullptr, 0, job);
}

extern "C" ngp_status ngp_predict_stage(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                                        const double *t, const double *y, int64_t ldy, int32_t m,
                                        const double *t_new, int32_t noise_on_new, ngp_job **job) {
    if (n <= 0 || m <= 0) return NGP_ERR_ARG;
    static const double dummy = 0.0;
    return stage_general(c, B, k, n, t, y, ldy, 0, &dummy, 1, &dummy, 0, m, t_new, noise_on_new,
                         job);
}

extern "C" ngp_status ngp_nowcast_stage(ngp_ctx *c, int32_t P, const ngp_kernel *k, int32_t n,
                                        const double *t, const double *y, int32_t d,
                                        const double *t_add, int32_t D, const double *y_add,
                                        int32_t m, const double *t_new, int32_t noise_on_new,
                                        ngp_job **job) {
    if (n <= 0 || d < 0 || D <= 0) return NGP_ERR_ARG;
    static const double dummy = 0.0;
    if (d == 0) { t_add = &dummy; y_add = &dummy; }
    return stage_general(c, P, k, n, t, y, 0, d, t_add, D, y_add, 0, m, t_new, noise_on_new, job);
}

static ngp_status run_fetch_destroy(ngp_job *job, double *lb, double *lf, double *mu,
                                    double *sigma, int32_t *info) {
    ngp_status st = ngp_job_run(job);
    if (!st) st = ngp_job_fetch(job, lb, lf, mu, sigma, info);
    ngp_job_destroy(job);
    return st;
}

static ngp_status logml_direct(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                               const double *t, const double *y, int64_t ldy, double *logml,
                               int32_t *info) {
    ngp_job *job = nullptr;
    ngp_status st = ngp_logml_stage(c, B, k, n, t, y, ldy, &job);
    if (st) return st;
    return run_fetch_destroy(job, nullptr, logml, nullptr, nullptr, info);
}

static ngp_status predict_direct(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                                 const double *t, const double *y, int64_t ldy, int32_t m,
                                 const double *t_new, int32_t noise_on_new, double *mu,
                                 double *sigma, double *logml, int32_t *info) {
    ngp_job *job = nullptr;
    ngp_status st = ngp_predict_stage(c, B, k, n, t, y, ldy, m, t_new, noise_on_new, &job);
    if (st) return st;
    return run_fetch_destroy(job, nullptr, logml, mu, sigma, info);
}

// (ngp_logml_batch, ngp_predict_batch, ngp_logml_grad_batch and ngp_mixture_sample are defined in
// the "combining" section further down: they enter through combine_submit)

extern "C" ngp_status ngp_nowcast_batch(ngp_ctx *c, int32_t P, const ngp_kernel *k, int32_t n,
                                        const double *t, const double *y, int32_t d,
                                        const double *t_add, int32_t D, const double *y_add,
                                        int32_t m, const double *t_new, int32_t noise_on_new,
                                        double *logml_base, double *logml_full, double *mu,
                                        double *sigma, int32_t *info) {
    ngp_job *job = nullptr;
    ngp_status st = ngp_nowcast_stage(c, P, k, n, t, y, d, t_add, D, y_add, m, t_new, noise_on_new,
                                      &job);
    if (st) return st;
    return run_fetch_destroy(job, logml_base, logml_full, mu, sigma, info);
}

// ---------------------------------------------------------------------------------------
// cached factor
// ---------------------------------------------------------------------------------------
struct ngp_factor {
    ngp_ctx *ctx = nullptr;
    int P = 0, n = 0;
    int64_t ldy = 0;
    std::vector<std::vector<int32_t>> ops;
    std::vector<std::vector<double>> params;
    std::vector<ngp_kernel> kernels;      // point into ops / params
    std::vector<double> t, y;             // y: [P or 1][n] packed
    int n0 = 0, nb0 = 0;
    int64_t item_stride = 0;              // (n0 + NGP_MAX_AUX) x n0 doubles
    double *L = nullptr;                  // [P][item_stride]
    double *dinv = nullptr;               // [nb0][P][64 x 64]
    double *logdet = nullptr;             // [P] sum log diag L of the main block
    int32_t *info = nullptr;              // [P]
    std::vector<double> logml0;
    std::vector<int32_t> info0;
    ngp_spec spec{};                      // the spec the factor was created under (always fp64)
};

namespace {

// run the job of a cached-factor query: only the aux rows are filled and swept
ngp_status factor_run(ngp_factor *f, ngp_job *j, bool create) {
    ngp_ctx *c = f->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    JobGeom &g = j->g;
    g.item_stride = f->item_stride;
    hipStream_t s = c->stream;
    j->spec = f->spec;
    const DevSpec sp = dev_spec(f->spec);
    EventTimer tm(c->profiling, s);
    const int P = f->P;
    void *tab = nullptr, *sig = nullptr;
    struct Scratch {   // the tables go back to the allocator on every exit path
        ngp_ctx *c; void *&a, *&b;
        ~Scratch() { c->release(a); c->release(b); }
    } scratch{c, tab, sig};
    if (g.n0 > 0) {
        if (g.lattice) {
            ngp_status st = c->alloc(&tab, sizeof(double) * (size_t)g.maxstat * g.R * P);
            if (!st) st = c->alloc(&sig, sizeof(double) * (size_t)g.maxcp * g.npts * P);
            if (st) return st;
        }
        ChunkPtrs p{};
        p.L = f->L;
        p.dinv = f->dinv;
        p.progs = j->progs;
        p.t0 = j->t0;
        p.taux = j->taux;
        p.y0 = j->y0;
        p.logdet = j->logdet;
        p.info = j->info;
        p.tab = (double *)tab;
        p.sig = (double *)sig;
        p.qpts = j->qpts;
        const double nrows_aux = (double)g.naux;
        const size_t mstep = (size_t)P * NB * NB;
        if (g.lattice) tm.run(4, 0.0, 0.0, [&] { launch_tables(g, p, P, sp, s); });
        if (create) {
            HIPCHK(hipMemsetAsync(j->logdet, 0, sizeof(double) * (size_t)P, s));
            HIPCHK(hipMemsetAsync(j->info, 0, sizeof(int32_t) * (size_t)P, s));
            tm.run(4, 0.0, 8.0 * P * ((double)g.n0 * (g.n0 + NB) / 2.0 + nrows_aux * g.n0),
                   [&] { launch_fill(g, p, P, sp, s); });
            factor_chunk(lane_of(c), g, p, P, tm, mstep);
            HIPCHK(hipMemcpyAsync(f->logdet, j->logdet, sizeof(double) * (size_t)P,
                                  hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(f->info, j->info, sizeof(int32_t) * (size_t)P,
                                  hipMemcpyDeviceToDevice, s));
        } else {
            HIPCHK(hipMemcpyAsync(j->logdet, f->logdet, sizeof(double) * (size_t)P,
                                  hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(j->info, f->info, sizeof(int32_t) * (size_t)P,
                                  hipMemcpyDeviceToDevice, s));
            tm.run(4, 0.0, 8.0 * P * nrows_aux * g.n0,
                   [&] { launch_fill(g, p, P, sp, s, /*aux_only=*/true); });
            // right-looking sweep of the aux rows through the resident factor
            for (int jj = 0; jj < g.nb0; ++jj) {
                ChunkPtrs pj = p;
                pj.dinv = f->dinv + (size_t)jj * mstep;
                tm.run(6, P * nrows_aux * (double)NB * NB, P * 8.0 * 3.0 * nrows_aux * NB,
                       [&] { launch_chol_col(g, pj, P, jj, COL_AUX, jj * NB, s); });
                tm.run(7, P * nrows_aux * 2.0 * NB * (double)(g.n0 - (jj + 1) * NB),
                       P * 8.0 * (g.n0 - (jj + 1) * NB) * (2.0 * nrows_aux + NB),
                       [&] { launch_aux_update(g, pj, P, jj, s); });
            }
        }
        tm.run(2, P * nrows_aux * nrows_aux * g.n0, P * 8.0 * nrows_aux * g.n0,
               [&] { launch_gram(g, f->L, j->G, P, s); });
    } else {
        HIPCHK(hipMemsetAsync(j->logdet, 0, sizeof(double) * (size_t)P, s));
        HIPCHK(hipMemsetAsync(j->info, 0, sizeof(int32_t) * (size_t)P, s));
    }
    EpiPtrs e{};
    e.progs = j->progs;
    e.taux = j->taux;
    e.G = j->G;
    e.ya = j->ya;
    e.logdet = j->logdet;
    e.info = j->info;
    e.work = j->work;
    e.zbuf = j->zbuf;
    e.logml_base = j->logml_base;
    e.logml_full = j->logml_full;
    e.mu = j->mu;
    e.sigma = j->sigma;
    e.work_stride = j->work_stride;
    if (g.lattice && g.n0 > 0) {
        e.tab = (const double *)tab;
        e.sig = (const double *)sig;
        e.qpts = j->qpts;
    }
    tm.run(3, 0.0, 0.0, [&] { launch_epilogue(g, e, sp, s); });
    hipError_t err = hipStreamSynchronize(s);
    if (err == hipSuccess) err = hipGetLastError();
    tm.resolve(c->prof);
    if (err != hipSuccess) return (ngp_status)err;
    j->ran = true;
    return NGP_OK;
}

}  // namespace

extern "C" ngp_status ngp_factor_create(ngp_ctx *c, int32_t P, const ngp_kernel *kernels,
                                        int32_t n, const double *t, const double *y, int64_t ldy,
                                        ngp_factor **out) {
    if (!c || !out || !kernels || !t || !y || P <= 0 || n <= 0) return NGP_ERR_ARG;
    *out = nullptr;
    if ((size_t)P > MAX_CHUNK_ITEMS) return NGP_ERR_TOO_LARGE;   // a resident factor is one chunk
    for (int i = 0; i < P; ++i) {
        ngp_status st = check_program(&kernels[i]);
        if (st) return st;
    }
    ngp_factor *f = new (std::nothrow) ngp_factor();
    if (!f) return NGP_ERR_TOO_LARGE;
    f->ctx = c;
    f->P = P;
    f->n = n;
    (void)ngp_get_spec(c, &f->spec);
    f->spec.precision = NGP_PREC_F64;     // a resident factor is queried many times: keep it exact
    f->ldy = ldy ? n : 0;
    f->ops.resize((size_t)P);
    f->params.resize((size_t)P);
    f->kernels.resize((size_t)P);
    for (int i = 0; i < P; ++i) {
        f->ops[(size_t)i].assign(kernels[i].ops, kernels[i].ops + kernels[i].n_ops);
        f->params[(size_t)i].assign(kernels[i].params, kernels[i].params + kernels[i].n_params);
        f->kernels[(size_t)i] = kernels[i];
        f->kernels[(size_t)i].ops = f->ops[(size_t)i].data();
        f->kernels[(size_t)i].params = f->params[(size_t)i].data();
    }
    f->t.assign(t, t + n);
    const int ny = ldy ? P : 1;
    f->y.resize((size_t)ny * n);
    for (int b = 0; b < ny; ++b)
        for (int i = 0; i < n; ++i) f->y[(size_t)b * n + i] = y[(int64_t)b * ldy + i];
    f->n0 = (n / NB) * NB;
    f->nb0 = f->n0 / NB;
    f->item_stride = (int64_t)(f->n0 + NGP_MAX_AUX) * f->n0;
    if (item_too_large(f->item_stride)) {
        delete f;
        return NGP_ERR_TOO_LARGE;
    }
    ngp_status st = NGP_OK;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) st = NGP_ERR_NO_DEVICE;
        void *q = nullptr;
        if (!st && f->n0 > 0) {
            if (!(st = c->alloc(&q, sizeof(double) * (size_t)f->item_stride * P))) f->L = (double *)q;
            if (!st && !(st = c->alloc(&q, sizeof(double) * (size_t)f->nb0 * P * NB * NB)))
                f->dinv = (double *)q;
        }
        if (!st && !(st = c->alloc(&q, sizeof(double) * (size_t)P))) f->logdet = (double *)q;
        if (!st && !(st = c->alloc(&q, sizeof(int32_t) * (size_t)P))) f->info = (int32_t *)q;
    }
    ngp_job *job = nullptr;
    static const double dummy = 0.0;
    if (!st)
        st = stage_general(c, P, f->kernels.data(), n, f->t.data(), f->y.data(), f->ldy, 0, &dummy,
                           1, &dummy, 0, 0, nullptr, 0, &job);
    if (!st) st = factor_run(f, job, /*create=*/true);
    f->logml0.resize((size_t)P);
    f->info0.resize((size_t)P);
    if (!st) st = ngp_job_fetch(job, nullptr, f->logml0.data(), nullptr, nullptr, f->info0.data());
    ngp_job_destroy(job);
    if (st) {
        ngp_factor_destroy(f);
        return st;
    }
    *out = f;
    return NGP_OK;
}

extern "C" ngp_status ngp_factor_logml(const ngp_factor *f, double *logml, int32_t *info) {
    if (!f) return NGP_ERR_ARG;
    for (int i = 0; i < f->P; ++i) {
        if (logml) logml[i] = f->logml0[(size_t)i];
        if (info) info[i] = f->info0[(size_t)i];
    }
    return NGP_OK;
}

extern "C" ngp_status ngp_factor_nowcast(ngp_factor *f, int32_t d, const double *t_add, int32_t D,
                                         const double *y_add, int32_t m, const double *t_new,
                                         int32_t noise_on_new, double *logml_base,
                                         double *logml_full, double *mu, double *sigma,
                                         int32_t *info) {
    if (!f || d < 0 || D <= 0 || m < 0) return NGP_ERR_ARG;
    static const double dummy = 0.0;
    if (d == 0) { t_add = &dummy; y_add = &dummy; }
    ngp_job *job = nullptr;
    ngp_status st = stage_general(f->ctx, f->P, f->kernels.data(), f->n, f->t.data(), f->y.data(),
                                  f->ldy, d, t_add, D, y_add, 0, m, t_new, noise_on_new, &job);
    if (st) return st;
    st = factor_run(f, job, /*create=*/false);
    if (!st) st = ngp_job_fetch(job, logml_base, logml_full, mu, sigma, info);
    ngp_job_destroy(job);
    return st;
}

extern "C" void ngp_factor_destroy(ngp_factor *f) {
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(f->ctx->mu);
        f->ctx->release(f->L);
        f->ctx->release(f->dinv);
        f->ctx->release(f->logdet);
        f->ctx->release(f->info);
    }
    delete f;
}

extern "C" ngp_status ngp_cov_batch(ngp_ctx *c, int32_t B, const ngp_kernel *kernels, int32_t n1,
                                    const double *t1, int32_t n2, const double *t2,
                                    int32_t add_diag, double *out) {
    if (!c || !kernels || !t1 || !t2 || !out || B <= 0 || n1 <= 0 || n2 <= 0) return NGP_ERR_ARG;
    std::vector<DevProgram> hp((size_t)B);
    for (int i = 0; i < B; ++i) {
        ngp_status st = compile_program(&kernels[i], &hp[(size_t)i], nullptr);
        if (st) return st;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    void *dp = nullptr, *d1 = nullptr, *d2 = nullptr, *dout = nullptr;
    const size_t nout = (size_t)B * n1 * n2;
    ngp_status st;
    if ((st = c->alloc(&dp, sizeof(DevProgram) * (size_t)B)) ||
        (st = c->alloc(&d1, sizeof(double) * (size_t)n1)) ||
        (st = c->alloc(&d2, sizeof(double) * (size_t)n2)) ||
        (st = c->alloc(&dout, sizeof(double) * nout))) {
        c->release(dp); c->release(d1); c->release(d2); c->release(dout);
        return st;
    }
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(dp, hp.data(), sizeof(DevProgram) * (size_t)B,
                                  hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d1, t1, sizeof(double) * (size_t)n1, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, t2, sizeof(double) * (size_t)n2, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        EventTimer tm(c->profiling, s);
        tm.run(4, 0.0, 8.0 * (double)nout, [&] {
            for (int b0 = 0; b0 < B; b0 += (int)MAX_CHUNK_ITEMS)   // items are blockIdx.y
                launch_cov((const DevProgram *)dp + b0, std::min(B - b0, (int)MAX_CHUNK_ITEMS),
                           (const double *)d1, n1, (const double *)d2, n2, add_diag,
                           (double *)dout + (size_t)b0 * n1 * n2, dev_spec(c->spec), s);
        });
        e = hipMemcpyAsync(out, dout, sizeof(double) * nout, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        tm.resolve(c->prof);
    }
    c->release(dp); c->release(d1); c->release(d2); c->release(dout);
    return e == hipSuccess ? NGP_OK : (ngp_status)e;
}

// ---------------------------------------------------------------------------------------
// gradient jobs: the inputs of ngp_logml_grad_batch kept on the device across calls
// ---------------------------------------------------------------------------------------
struct GradLeaf {
    bool toep_path = false;                // the Toeplitz gradient path
    ngp_ctx *ctx = nullptr;
    JobGeom g{};
    int B = 0, n = 0;
    PinVec<DevProgram> hp;                 // compiled programs (device parameter order), page-locked: they go up on every run
    std::vector<std::vector<int>> perm;    // device parameter k of item i = the caller's perm[i][k]
    std::vector<int32_t> n_ops, n_params;
    // ONE device arena for everything that crosses the bus:
    //   [programs | t | y | lattice indices | info | logdet | gradient | logml]
    unsigned char *io = nullptr;
    size_t o_t = 0, o_y = 0, o_q = 0, o_info = 0, o_logdet = 0, o_grad = 0, o_logml = 0, io_bytes = 0;
    // lattice jobs: the items by the shape of their reduced program, as in a staged value job
    // (ascending leaf indices; the fill of the main tiles runs on the value jobs' kernels)
    std::vector<int32_t> fill_single, fill_chain, fill_other;
    size_t o_fs = 0, o_fc = 0, o_fo = 0;
    PinVec<unsigned char> h_in, h_out;        // staging copy of a small job's inputs; results
    bool fresh = false;                    // info / logdet still hold the zeros they were staged with
    bool progs_dirty = false;              // set_params since the last upload
    ngp_spec spec{};
    int last_chunk = 0;                    // items per memory-driven chunk of the last run (ngp_grad_job_info)
};

namespace {

// toep_path: the items are stationary trees on a regular series — aux rows [y' ; e_1'] instead of
// [I ; y'] (the caller, ngp_grad_stage, has checked both)
// yrows: item b's observations are yrows[b][0..n) (y / ldy are then not read) — a subset of a
// caller's batch, or the rows of several callers' arrays in a combined call
ngp_status grad_leaf_stage(ngp_ctx *c, int32_t B, const ngp_kernel *kernels, int32_t n,
                           const double *t, const double *y, int64_t ldy, bool toep_path,
                           GradLeaf **out, const double *const *yrows = nullptr) {
    if (!c || !out || !kernels || !t || (!y && !yrows) || B <= 0 || n <= 0) return NGP_ERR_ARG;
    *out = nullptr;
    GradLeaf *j = new (std::nothrow) GradLeaf();
    if (!j) return NGP_ERR_TOO_LARGE;
    std::unique_ptr<GradLeaf> guard(j);
    j->toep_path = toep_path;
    j->ctx = c;
    j->B = B;
    j->n = n;
    j->hp.resize((size_t)B);
    j->perm.resize((size_t)B);
    j->n_ops.resize((size_t)B);
    j->n_params.resize((size_t)B);
    int maxstat = 0, maxcp = 0, maxops = 0, maxtab = 0;
    for (int i = 0; i < B; ++i) {
        int ns = 0, nc = 0, nt = 0;
        ngp_status st = compile_program(&kernels[i], &j->hp[(size_t)i], &j->perm[(size_t)i], &ns, &nc, &nt);
        if (st) return st;
        maxstat = std::max(maxstat, ns);
        maxtab = std::max(maxtab, nt);
        maxcp = std::max(maxcp, nc);
        maxops = std::max(maxops, (int)kernels[i].n_ops);
        j->n_ops[(size_t)i] = kernels[i].n_ops;
        j->n_params[(size_t)i] = kernels[i].n_params;
    }
    // Geometry: the matrix is padded to a multiple of 64 with identity rows/cols (log 1 = 0, a
    // zero in y), and the aux block is [I ; y'] so that the factorisation leaves W = [L^-T ; z'].
    JobGeom &g = j->g;
    g.B = B;
    g.n0 = (n + NB - 1) / NB * NB;
    g.nb0 = g.n0 / NB;
    g.n_real = n;
    g.aux_identity = toep_path ? 0 : 1;
    g.aux_e1 = toep_path ? 1 : 0;
    g.maxops = maxops;
    g.naux = toep_path ? 2 : g.n0 + 1;
    g.naux_pad = toep_path ? NB : g.n0 + NB;
    g.D = 1;
    g.y_shared = (ldy == 0 && !yrows) ? 1 : 0;
    g.ld = g.n0;
    g.item_stride = (int64_t)(g.n0 + g.naux_pad) * g.n0;
    if (item_too_large(g.item_stride)) return NGP_ERR_TOO_LARGE;
    g.npts = g.n0;
    g.maxstat = std::max(maxstat, 1);
    g.maxcp = std::max(maxcp, 1);
    std::vector<int32_t> h_q;
    {
        std::vector<double> real(t, t + n);
        double hh = 0.0;
        int R = 0;
        if (detect_lattice(real, &hh, &h_q, &R)) {
            g.lattice = 1;
            g.h = hh;
            g.R = R;
            h_q.resize((size_t)g.n0, 0);
            // the subtree tables of the reduced programs behind the per-leaf tables (tables_kernel),
            // and the items sorted by the kernel that fills their main tiles (launch_fill)
            g.tab_sub = g.maxstat;
            g.maxstat += maxtab;
            for (int i = 0; i < B; ++i)
                (prog_single_table(&j->hp[(size_t)i]) ? j->fill_single
                                                       : j->hp[(size_t)i].rchain ? j->fill_chain : j->fill_other)
                    .push_back(i);
        }
    }
    const int ny = g.y_shared ? 1 : B;
    const int GP = NGP_MAX_PARAMS + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    j->o_t = al(sizeof(DevProgram) * (size_t)B);
    j->o_y = j->o_t + al(8 * (size_t)g.n0);
    j->o_q = j->o_y + al(8 * (size_t)ny * g.n0);
    j->o_fs = j->o_q + (g.lattice ? al(4 * (size_t)g.n0) : 0);
    j->o_fc = j->o_fs + al(4 * j->fill_single.size());
    j->o_fo = j->o_fc + al(4 * j->fill_chain.size());
    j->o_info = j->o_fo + al(4 * j->fill_other.size());
    j->o_logdet = j->o_info + al(4 * (size_t)B);
    j->o_grad = j->o_logdet + al(8 * (size_t)B);
    j->o_logml = j->o_grad + al(8 * (size_t)B * GP);
    j->io_bytes = j->o_logml + al(8 * (size_t)B);
    j->h_in.assign(j->o_grad, 0);
    j->h_out.resize(j->io_bytes - j->o_info);
    std::memcpy(j->h_in.data(), j->hp.data(), sizeof(DevProgram) * (size_t)B);
    {
        double *ht = (double *)(j->h_in.data() + j->o_t), *hy = (double *)(j->h_in.data() + j->o_y);
        for (int i = 0; i < g.n0; ++i) ht[i] = t[std::min(i, n - 1)];
        for (int b = 0; b < ny; ++b)
            std::memcpy(hy + (size_t)b * g.n0, yrows ? yrows[b] : y + (int64_t)b * ldy, 8 * (size_t)n);
        if (g.lattice) std::memcpy(j->h_in.data() + j->o_q, h_q.data(), 4 * (size_t)g.n0);
        if (!j->fill_single.empty())
            std::memcpy(j->h_in.data() + j->o_fs, j->fill_single.data(), 4 * j->fill_single.size());
        if (!j->fill_chain.empty())
            std::memcpy(j->h_in.data() + j->o_fc, j->fill_chain.data(), 4 * j->fill_chain.size());
        if (!j->fill_other.empty())
            std::memcpy(j->h_in.data() + j->o_fo, j->fill_other.data(), 4 * j->fill_other.size());
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    j->spec = c->spec;
    g.invariant = c->invariant ? 1 : 0;
    g.short_series = c->short_series ? 1 : 0;
    // (gradient jobs store every tile: their tables are per leaf, and on a regular series the
    // stationary trees — the ones structured storage could serve — are on the Toeplitz path)
    void *q = nullptr;
    ngp_status st = c->alloc(&q, j->io_bytes);
    if (st) return st;
    j->io = (unsigned char *)q;
    // the front part goes up in one copy (info and logdet arrive as the zeros the factorisation
    // expects): a 24-particle call of a fit on a short series is a chain of dependent launches a
    // few microseconds long, and each separate copy or memset was one more of them
    if (hipMemcpyAsync(j->io, j->h_in.data(), j->h_in.size(), hipMemcpyHostToDevice, c->stream) !=
        hipSuccess) {
        c->release(j->io);
        return NGP_ERR_STATE;
    }
    j->fresh = true;
    // a small staging buffer stays with the job (the run is queued right behind the copy); a
    // large one is given back once the copy has left it
    if (j->h_in.size() > STAGE_KEEP_BYTES) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            c->release(j->io);
            return NGP_ERR_STATE;
        }
        PinVec<unsigned char>().swap(j->h_in);
    }
    *out = guard.release();
    return NGP_OK;
}

ngp_status grad_leaf_set_params(GradLeaf *j, const double *params, const double *noise) {
    if (!j || !params || !noise) return NGP_ERR_ARG;
    size_t off = 0;
    for (int i = 0; i < j->B; ++i) {
        DevProgram &P = j->hp[(size_t)i];
        const int np = j->n_params[(size_t)i];
        // values are taken as ngp_logml_grad_batch takes them: a parameter the kernels cannot use
        // shows in the item's info, not here
        for (int q = 0; q < np; ++q) P.params[q] = params[off + (size_t)j->perm[(size_t)i][(size_t)q]];
        P.noise = noise[i];
        off += (size_t)np;
    }
    j->progs_dirty = true;
    return NGP_OK;
}

// One evaluation of a leaf in three steps, so that two leaves of a small batch can be in flight side
// by side (grad_pair_run): lay the working storage out in the context's workspace, enqueue
// everything on a lane's streams (upload of new parameters ... results on their way back), and,
// after the lane has been synchronised, unpack.
struct LeafRun {
    GradLeaf *j = nullptr;
    int Bc = 0;
    void *d_L = nullptr, *d_dinv = nullptr, *d_tab = nullptr, *d_sig = nullptr, *d_dtab = nullptr,
         *d_kinv = nullptr, *d_alpha = nullptr, *d_quad = nullptr, *d_part = nullptr, *d_items = nullptr;
    std::vector<int32_t> h_items;   // alive until the lane is synchronised
    double *splitk = nullptr;       // pair runs: the context's split-k buffer for this leaf's value kernels

    size_t item_bytes() const {
        const JobGeom &g = j->g;
        const int GP = NGP_MAX_PARAMS + 1, ntri = g.nb0 * (g.nb0 + 1) / 2, nd = (g.n_real + 255) / 256;
        const size_t l_bytes = (size_t)g.item_stride * 8;
        const size_t tab_bytes = g.lattice ? 8 * (size_t)g.maxstat * g.R : 0;
        const size_t sig_bytes = g.lattice ? 8 * (size_t)g.maxcp * g.npts : 0;
        const size_t aux_bytes = 8 * (size_t)g.naux_pad * g.n0;
        return j->toep_path ? l_bytes + 4 * tab_bytes + sig_bytes + 8 * (size_t)NB * NB * g.nb0 + aux_bytes +
                                  8 * (size_t)g.n0 + 8 * (size_t)nd * GP
                            : l_bytes + 4 * tab_bytes + sig_bytes + 8 * (size_t)g.n0 * g.n0 +
                                  8 * (size_t)g.n0 + 8 * (size_t)ntri * GP;
    }
    // the working buffers of a chunk of Bc items, requested from `dalloc` (measured, then taken)
    ngp_status layout(WsPlan &dalloc) {
        const JobGeom &g = j->g;
        const bool tp = j->toep_path;
        const int GP = NGP_MAX_PARAMS + 1, ntri = g.nb0 * (g.nb0 + 1) / 2, nd = (g.n_real + 255) / 256;
        const size_t l_bytes = (size_t)g.item_stride * 8;
        const size_t tab_bytes = g.lattice ? 8 * (size_t)g.maxstat * g.R : 0;
        const size_t sig_bytes = g.lattice ? 8 * (size_t)g.maxcp * g.npts : 0;
        const size_t aux_bytes = 8 * (size_t)g.naux_pad * g.n0;
        ngp_status r;
        (void)((r = dalloc(&d_L, l_bytes * (size_t)Bc)) ||
               // the Toeplitz path keeps every block inverse M_j for its backward sweep
               (r = dalloc(&d_dinv, 8 * (size_t)Bc * NB * NB * (tp ? g.nb0 : 1))) ||
               (g.lattice && ((r = dalloc(&d_tab, tab_bytes * (size_t)Bc)) ||
                              (r = dalloc(&d_sig, sig_bytes * (size_t)Bc)) ||
                              (r = dalloc(&d_dtab, 3 * tab_bytes * (size_t)Bc)))) ||
               // general path: K^-1 [n0 x n0]; Toeplitz path: A = [a' ; x'] (one aux tile)
               (r = dalloc(&d_kinv, tp ? aux_bytes * (size_t)Bc : 8 * (size_t)Bc * g.n0 * g.n0)) ||
               (r = dalloc(&d_alpha, 8 * (size_t)Bc * g.n0)) ||      // Toeplitz path: w per distance
               (r = dalloc(&d_quad, 8 * (size_t)Bc)) ||
               // a chunk that is cut finer (split 2 or 4) writes at most 4096 partial rows; a coarse one Bc ntri
               (r = dalloc(&d_part,
                           tp ? 8 * (size_t)Bc * nd * GP
                              : 8 * std::max<size_t>((size_t)Bc * ntri * grad_contract_split(ntri, Bc, g.invariant != 0),
                                                     4096) * GP)) ||
               (r = dalloc(&d_items, 4 * (size_t)j->B)));
        return r;
    }

    // everything of the evaluation on the lane's streams; the caller synchronises ln.main
    ngp_status launch(const Lane &ln, EventTimer &tm, bool no_lane_split) {
        ngp_ctx *c = j->ctx;
        const JobGeom &g = j->g;
        const int B = j->B;
        const bool tp = j->toep_path;
        const int GP = NGP_MAX_PARAMS + 1, ntri = g.nb0 * (g.nb0 + 1) / 2, nd = (g.n_real + 255) / 256;
        hipStream_t s = ln.main;
        const DevSpec sp = dev_spec(j->spec);
        h_items.assign((size_t)B, 0);
        unsigned char *const io = j->io;
        void *const d_prog = io, *const d_t = io + j->o_t, *const d_y = io + j->o_y,
                    *const d_q = g.lattice ? io + j->o_q : nullptr, *const d_info = io + j->o_info,
                    *const d_logdet = io + j->o_logdet, *const d_grad = io + j->o_grad,
                    *const d_logml = io + j->o_logml;
        hipError_t e = hipSuccess;
        // new parameters for the same trees: the programs go up again, nothing else.  On a lattice the
        // first kernel of the chain (tables_kernel) fetches them from the page-locked host copy itself
        // and leaves the device copy for the rest — a copy ahead of the chain cost 20 us of a 24-item
        // call's 110.
        const bool fetch_in_kernel = j->progs_dirty && g.lattice && g.n0 > 0 &&
                                     pinned_pool().is_pinned(j->hp.data());
        if (j->progs_dirty && !fetch_in_kernel)
            e = hipMemcpyAsync(d_prog, j->hp.data(), sizeof(DevProgram) * (size_t)B,
                               hipMemcpyHostToDevice, s);
        // a re-run: info | logdet are contiguous in the arena (short jobs: written outright)
        if (e == hipSuccess && !j->fresh && !small_job(g, Bc))
            e = hipMemsetAsync(d_info, 0, j->o_grad - j->o_info, s);
        if (e != hipSuccess) return (ngp_status)e;
        j->progs_dirty = false;
        j->fresh = false;
        for (int b0 = 0; b0 < B; b0 += Bc) {
            const int bc = std::min(Bc, B - b0);
            ChunkPtrs p{};
            p.L = (double *)d_L;
            p.dinv = (double *)d_dinv;
            p.progs = (const DevProgram *)d_prog + b0;
            p.progs_src = fetch_in_kernel ? j->hp.data() + b0 : nullptr;
            p.t0 = (const double *)d_t;
            p.taux = (const double *)d_t;
            p.y0 = (const double *)d_y + (g.y_shared ? 0 : (int64_t)b0 * g.n0);
            p.logdet = (double *)d_logdet + b0;
            p.info = (int32_t *)d_info + b0;
            p.tab = (double *)d_tab;
            p.sig = (double *)d_sig;
            p.qpts = (const int32_t *)d_q;
            p.dtab = (double *)d_dtab;
            p.splitk_part = splitk;
            if (g.lattice) {   // the chunk's share of the three fill lists
                auto range = [&](const std::vector<int32_t> &v, size_t off, const int32_t **ptr, int32_t *cnt) {
                    const auto lo = std::lower_bound(v.begin(), v.end(), b0);
                    const auto hi = std::lower_bound(v.begin(), v.end(), b0 + bc);
                    *ptr = (const int32_t *)(io + off) + (lo - v.begin());
                    *cnt = (int32_t)(hi - lo);
                };
                range(j->fill_single, j->o_fs, &p.fill_single, &p.n_fill_single);
                range(j->fill_chain, j->o_fc, &p.fill_chain, &p.n_fill_chain);
                range(j->fill_other, j->o_fo, &p.fill_other, &p.n_fill_other);
                p.fill_base = b0;
            }
            if (g.lattice) tm.run(4, 0.0, 0.0, [&] { launch_tables(g, p, bc, sp, s); });
            // K's lower blocks, the y' tile row and the zero blocks (a, a-1): the identity block of
            // the aux rows is synthesised by the column kernels, not written
            tm.run(4, 0.0,
                   8.0 * bc * ((double)g.n0 * (g.n0 + NB) / 2.0 + (tp ? 1.0 : 2.0) * NB * (double)g.n0),
                   [&] { launch_fill(g, p, bc, sp, s); });
            const size_t mstep = tp ? (size_t)bc * NB * NB : 0;
            factor_chunk(ln, g, p, bc, tm, mstep, nullptr, nullptr, nullptr, no_lane_split);
            const double n3 = (double)g.n0 * g.n0 * g.n0;
            if (tp) {
                // z'z, then A = X K^-1 by one backward sweep of the two aux rows (class 10 with the
                // other backward sweeps of the library)
                tm.run(10, 0.0, bc * 8.0 * g.n0,
                       [&] { launch_toep_quad(g, (const double *)d_L, (double *)d_quad, bc, s); });
                for (int cc = g.nb0 - 1; cc >= 0; --cc)
                    tm.run(10, bc * 2.0 * 2.0 * NB * (double)(cc + 1) * NB,
                           bc * 8.0 * ((double)NB * NB * (cc + 1) + 2.0 * 2.0 * NB * (cc + 1)), [&] {
                               launch_aux_back(g, p, p.dinv, mstep, (double *)d_kinv, 0, bc, cc, s);
                           });
            } else {
                // short jobs (factor_chunk's rule), and small chunks of series up to 448 points on the
                // column sweep: its W_I has the same entries where the 16 x 16-block kernel reads them
                const bool short_job = small_job(g, bc) || (g.nb0 < 8 && bc <= AHEAD_EARLY_MAX_ITEMS && !g.invariant);
                tm.run(5, bc * n3 / 3.0, bc * 8.0 * 1.5 * (double)g.n0 * g.n0, [&] {
                    if (short_job)
                        launch_grad_kinv_small(g, (const double *)d_L, (double *)d_kinv, (double *)d_alpha,
                                               (double *)d_quad, bc, s);
                    else
                        launch_grad_kinv(g, (const double *)d_L, (double *)d_kinv, (double *)d_alpha,
                                         (double *)d_quad, bc, s, ln.side, ln.fork, ln.join);
                });
            }
            // the chunk's items sorted by tree size: every size class runs on the contraction kernel
            // sized for it
            // (small launches — the 24- or 64-particle calls of a fit on short series — stay ONE launch
            // sized by the largest tree: up to five dependent launches of a few microseconds each cost
            // more there than the occupancy of the smaller instantiations gains)
            // (batch-invariant jobs: always — an item then runs on the instantiation of ITS tree size,
            // not on the one the largest tree of its batch picks)
            const bool by_size = g.lattice && (g.invariant || (tp ? (long)nd * bc > 512 : (long)ntri * bc > 4096));
            int32_t counts[GRAD_BUCKETS] = {};
            if (by_size) {
                for (int i = 0; i < bc; ++i) ++counts[grad_bucket(j->n_ops[(size_t)(b0 + i)])];
                int32_t pos[GRAD_BUCKETS], acc = 0;
                for (int k = 0; k < GRAD_BUCKETS; ++k) { pos[k] = acc; acc += counts[k]; }
                for (int i = 0; i < bc; ++i)
                    h_items[(size_t)b0 + (size_t)pos[grad_bucket(j->n_ops[(size_t)(b0 + i)])]++] = i;
                const hipError_t ce = hipMemcpyAsync((int32_t *)d_items + b0, h_items.data() + b0,
                                                     4 * (size_t)bc, hipMemcpyHostToDevice, s);
                if (ce != hipSuccess) return (ngp_status)ce;
            }
            if (tp) {
                tm.run(11, 0.0, bc * 8.0 * 3.0 * (double)g.n0, [&] {
                    launch_toep_grad(g, p, (const double *)d_kinv, (double *)d_alpha,
                                     (const double *)d_quad, (double *)d_part,
                                     (double *)d_grad + (int64_t)b0 * GP, (double *)d_logml + b0, bc, sp,
                                     s, by_size ? (const int32_t *)d_items + b0 : nullptr, counts);
                });
            } else {
                tm.run(11, 0.0, bc * 8.0 * 0.5 * (double)g.n0 * g.n0, [&] {
                    launch_grad_contract(g, p, (const double *)d_kinv, (const double *)d_alpha,
                                         (const double *)d_quad, (double *)d_part,
                                         (double *)d_grad + (int64_t)b0 * GP, (double *)d_logml + b0,
                                         bc, sp, s, by_size ? (const int32_t *)d_items + b0 : nullptr,
                                         counts, ln.side, ln.fork, ln.join);
                });
            }
        }
        e = hipMemcpyAsync(j->h_out.data(), d_info, j->h_out.size(), hipMemcpyDeviceToHost, s);
        return e == hipSuccess ? NGP_OK : (ngp_status)e;
    }

    // after the lane is synchronised: device parameter order -> caller's order; d/d noise last
    void unpack(double *logml, double *grad, int32_t *info) {
        const int GP = NGP_MAX_PARAMS + 1;
        if (!j->h_in.empty()) PinVec<unsigned char>().swap(j->h_in);   // the staging copy has left it
        const int32_t *h_info = (const int32_t *)j->h_out.data();
        const double *h_grad = (const double *)(j->h_out.data() + (j->o_grad - j->o_info)),
                     *h_lm = (const double *)(j->h_out.data() + (j->o_logml - j->o_info));
        size_t off = 0;
        for (int i = 0; i < j->B; ++i) {
            const int np = j->n_params[(size_t)i];
            for (int k = 0; k < np; ++k)
                grad[off + (size_t)j->perm[(size_t)i][(size_t)k]] = h_grad[(size_t)i * GP + (size_t)k];
            grad[off + (size_t)np] = h_grad[(size_t)i * GP + (size_t)np];
            off += (size_t)np + 1;
            if (logml) logml[i] = h_lm[(size_t)i];
            if (info) info[i] = h_info[(size_t)i];
        }
    }
};

ngp_status grad_leaf_run(GradLeaf *j, double *logml, double *grad, int32_t *info) {
    if (!j || !grad) return NGP_ERR_ARG;
    ngp_ctx *c = j->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    LeafRun r;
    r.j = j;
    const size_t item_bytes = r.item_bytes();
    if ((size_t)j->B * item_bytes > c->mem_cap) c->refresh_mem_cap();   // large job: today's figure
    r.Bc = (int)std::min<size_t>(std::min<size_t>((size_t)j->B, MAX_CHUNK_ITEMS),
                                 std::max<size_t>(1, c->mem_cap / item_bytes));
    // the chunk is halved when the device cannot hold it after all (other handles, rounding)
    for (;; r.Bc = (r.Bc + 1) / 2) {
        WsPlan measure{c, true}, take{c, false};
        (void)r.layout(measure);
        ngp_status st = c->ws_reserve(measure.total);
        if (!st) st = r.layout(take);
        if (!st) break;
        if (st != NGP_ERR_TOO_LARGE || r.Bc <= 1) return st;
    }
    j->last_chunk = r.Bc;
    EventTimer tm(c->profiling, c->stream);
    ngp_status st = r.launch(lane_of(c), tm, false);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    tm.resolve(c->prof);
    if (st) return st;
    if (e != hipSuccess) return (ngp_status)e;
    r.unpack(logml, grad, info);
    return NGP_OK;
}

// The two leaves of a SMALL mixed batch side by side, each on its own pair of streams (the lanes of
// the two-lane sweep): one after the other they are two chains of dependent launches where the
// unsplit batch is one (64 items at n = 2048: 16.3 ms against 15.3).  Both as single chunks in one
// workspace reservation; NGP_ERR_TOO_LARGE if that does not fit (the caller then runs them in turn).
ngp_status grad_pair_run(GradLeaf *a, double *lm_a, double *g_a, int32_t *info_a, GradLeaf *b,
                         double *lm_b, double *g_b, int32_t *info_b) {
    ngp_ctx *c = a->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (!c->make_lanes(2)) return NGP_ERR_TOO_LARGE;
    LeafRun ra, rb;
    ra.j = a;
    rb.j = b;
    ra.Bc = a->last_chunk = a->B;
    rb.Bc = b->last_chunk = b->B;
    WsPlan measure{c, true}, take{c, false};
    (void)ra.layout(measure);
    (void)rb.layout(measure);
    if (measure.total > c->mem_cap) c->refresh_mem_cap();
    if (measure.total > c->mem_cap) return NGP_ERR_TOO_LARGE;
    ngp_status st = c->ws_reserve(measure.total);
    if (st) return st;
    if ((st = ra.layout(take)) || (st = rb.layout(take))) return st;
    // the Toeplitz leaf runs on the value kernels: their split-k fat steps of late columns (small
    // chunks) need the context's buffer, which factor_chunk only reserves for a chunk it owns whole
    if (b->toep_path && b->g.nb0 >= 8 && b->B <= AHEAD_EARLY_MAX_ITEMS && !b->g.invariant &&
        c->splitk_reserve(b->B) == NGP_OK)
        rb.splitk = c->splitk_part;
    const Lane l0 = lane_of(c);
    const Lane l1{c->lane_main[1], c->lane_side[1], c->lane_fork[1], c->lane_join[1], c};
    // what the context's stream holds (the staging copies of both leaves) comes first on lane 1 too
    (void)hipEventRecord(c->ev_lane_go, c->stream);
    (void)hipStreamWaitEvent(l1.main, c->ev_lane_go, 0);
    EventTimer tma(c->profiling, l0.main), tmb(c->profiling, l1.main);
    ngp_status sa = ra.launch(l0, tma, true);
    ngp_status sb = rb.launch(l1, tmb, true);
    hipError_t e = hipStreamSynchronize(l0.main);
    const hipError_t e1 = hipStreamSynchronize(l1.main);
    if (e == hipSuccess) e = e1;
    if (e == hipSuccess) e = hipGetLastError();
    tma.resolve(c->prof);
    tmb.resolve(c->prof);
    if (sa) return sa;
    if (sb) return sb;
    if (e != hipSuccess) return (ngp_status)e;
    ra.unpack(lm_a, g_a, info_a);
    rb.unpack(lm_b, g_b, info_b);
    return NGP_OK;
}

void grad_leaf_destroy(GradLeaf *j) {
    if (!j) return;
    {
        std::lock_guard<std::mutex> lk(j->ctx->mu);
        (void)hipSetDevice(j->ctx->device);
        // a staging buffer that was kept must outlive its copy
        if (!j->h_in.empty()) (void)hipStreamSynchronize(j->ctx->stream);
        j->ctx->release(j->io);
    }
    delete j;
}

}  // namespace

// The public handle: the batch as the caller sees it, carried by up to two leaves — the items that
// are stationary trees, when the series is regular, on the Toeplitz path, the others on the general
// one — with the index maps that scatter results and parameters.
struct ngp_grad_job {
    ngp_ctx *ctx = nullptr;
    int B = 0;
    GradLeaf *gen = nullptr, *toep = nullptr;
    std::vector<int32_t> idx_gen, idx_toep;     // leaf item -> caller's item
    std::vector<size_t> poff, goff;             // [B + 1] offsets of an item's parameters / gradient
    std::vector<double> buf_p, buf_n, buf_lm, buf_g, buf_lm2, buf_g2;
    std::vector<int32_t> buf_info, buf_info2;
    bool side_by_side = false;                  // a small split batch: both leaves in flight together
    // Small jobs keep a host copy of what a ONE-SHOT call over their items would take (trees and
    // current parameters in the caller's order, dates, observation rows): when several tasks run
    // their jobs at the same moment — the leapfrog steps of the per-scenario HMC moves of
    // forecast_with_nowcasts, src/forecasting.jl:145-148 — the runs are combined into one call
    // over all their items ("combining" section); alone, a run uses the resident inputs as before.
    std::vector<int32_t> h_ops;
    std::vector<double> h_params, h_t, h_y;
    std::vector<ngp_kernel> h_k;
    int32_t n = 0;
    int64_t h_ldy = 0;
};
// host copies up to this size (64 particles x 2,049 points = 1 MB; a lockstep job of 12,800 items
// is 210 MB and has no use for company)
static constexpr size_t GRAD_JOB_HOST_COPY_BYTES = (size_t)16 << 20;

static ngp_status grad_stage_impl(ngp_ctx *c, int32_t B, const ngp_kernel *kernels, int32_t n,
                                  const double *t, const double *y, int64_t ldy,
                                  const double *const *yrows, ngp_grad_job **out,
                                  bool keep_host = false) {
    if (!c || !out || !kernels || !t || (!y && !yrows) || B <= 0 || n <= 0) return NGP_ERR_ARG;
    *out = nullptr;
    for (int i = 0; i < B; ++i) {
        ngp_status st = check_program(&kernels[i]);
        if (st) return st;
    }
    std::unique_ptr<ngp_grad_job> j(new (std::nothrow) ngp_grad_job());
    if (!j) return NGP_ERR_TOO_LARGE;
    j->ctx = c;
    j->B = B;
    j->poff.assign((size_t)B + 1, 0);
    j->goff.assign((size_t)B + 1, 0);
    for (int i = 0; i < B; ++i) {
        j->poff[(size_t)i + 1] = j->poff[(size_t)i] + (size_t)kernels[i].n_params;
        j->goff[(size_t)i + 1] = j->goff[(size_t)i] + (size_t)kernels[i].n_params + 1;
    }
    // the Toeplitz path: a regular series (constant lattice stride over all n points), at least two
    // blocks, short enough for the weights kernel's LDS image, and a tree without Linear or
    // ChangePoint nodes
    bool regular = false;
    bool on, invariant, short_series;
    double jitter;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        on = c->toeplitz;
        invariant = c->invariant;
        short_series = c->short_series;
        jitter = c->spec.jitter;
    }
    // (series of up to 256 points: the general leaf factorises them in one launch,
    // ngp_small_kernels.h — shorter than the Toeplitz leaf's chain of sweeps)
    // (batches beyond what the one-launch path takes, SM_MAX_ITEMS, keep the Toeplitz leaf: there it is
    // the faster one — 8,192 stationary items at n = 208: 14.8 against 21.3 ms; batch-invariant contexts
    // route by the series alone)
    if (on && n >= 2 * NB && n <= 8192 && !(short_series && n <= 4 * NB && (B <= SM_MAX_ITEMS || invariant))) {
        std::vector<double> real(t, t + n);
        std::vector<int32_t> q;
        double hh = 0.0;
        int R = 0;
        if (detect_lattice(real, &hh, &q, &R)) {
            const long st0 = (long)q[1] - (long)q[0];
            regular = st0 != 0;
            for (int i = 2; i < n && regular; ++i) regular = (long)q[(size_t)i] - (long)q[(size_t)i - 1] == st0;
        }
    }
    for (int i = 0; i < B; ++i) {
        // (trees of more than 16 leaves stay on the general path: the 1-D contraction runs on the
        // register-accumulator kernels, which end there)
        bool stationary = regular && kernels[i].n_ops <= 31;
        for (int k = 0; stationary && k < kernels[i].n_ops; ++k)
            stationary = kernels[i].ops[k] != NGP_OP_LINEAR && kernels[i].ops[k] != NGP_OP_CHANGEPOINT;
        if (stationary) {
            // The Gohberg-Semencul sums divide by x_0 = (K^-1)_11 and lose about eps cond(K) of their
            // digits; cond(K) <= n k(0) / (noise + jitter).  A matrix whose diagonal shift is below
            // 1e-9 of its diagonal k(0) (reachable only with a jitter far below the default 1e-5: HMC
            // clamps the noise at 1e-12) is not trusted to that formula and takes the general leaf.
            double st[NGP_MAX_STACK];
            int sp_ = 0, pi = 0;
            for (int k = 0; k < kernels[i].n_ops; ++k) {
                const int op = kernels[i].ops[k];
                if (op == NGP_OP_PLUS || op == NGP_OP_TIMES) {
                    const double b = st[--sp_], a = st[--sp_];
                    st[sp_++] = op == NGP_OP_PLUS ? a + b : a * b;
                } else {   // value at distance zero: the constant, or the leaf's amplitude (its last parameter)
                    const int np = k_nparams[op];
                    st[sp_++] = std::fabs(kernels[i].params[pi + np - 1]);
                    pi += np;
                }
            }
            const double k0 = st[0];
            if (!(kernels[i].noise + jitter >= 1e-9 * k0)) stationary = false;
        }
        (stationary ? j->idx_toep : j->idx_gen).push_back(i);
    }
    // A batch that is split runs its two leaves one after the other: two chains of dependent launches
    // instead of one.  That pays when the leaves are throughput-bound (12,800 items at n = 2049:
    // 2,615 -> 1,794 ms) and costs when they are latency-bound (24 items at n = 208: 615 -> 790 us; 64 at
    // n = 2048: 15.9 -> 16.3 ms), so a mixed batch is split only from SPLIT_MIN_ITEMS on; a batch
    // of stationary trees only is never split and always takes the Toeplitz path.
    // Just below that (PAIR_MIN_ITEMS .. SPLIT_MIN_ITEMS, long series) the two leaves run SIDE BY SIDE
    // on two stream pairs (grad_pair_run); smaller mixed batches are not split at all.  Measured on
    // the prior ensemble at n = 2048, general job -> split: 64 items 15.1 -> 17.2 ms side by side
    // (each leaf's chain is as long as the whole batch's, and they compete for the chip), 128 items
    // 26.8 -> 25.0 side by side, 256 items 50.2 -> 40.0 and 512 items 98.8 -> 72.0 in turn
    // (scripts/mixed_grad_probe.py).
    constexpr int SPLIT_MIN_ITEMS = 256, PAIR_MIN_ITEMS = 128, PAIR_MIN_N = 1024;
    // (batch-invariant contexts route by the item alone: a stationary tree on a regular series
    // always takes the Toeplitz leaf, whatever travels with it)
    if (!j->idx_gen.empty() && !j->idx_toep.empty() && B < SPLIT_MIN_ITEMS) {
        if (invariant || (n >= PAIR_MIN_N && B >= PAIR_MIN_ITEMS)) {
            j->side_by_side = true;
        } else {
            j->idx_toep.clear();
            j->idx_gen.resize((size_t)B);
            for (int i = 0; i < B; ++i) j->idx_gen[(size_t)i] = i;
        }
    }
    auto stage_leaf = [&](const std::vector<int32_t> &idx, bool toep_path, GradLeaf **leaf) -> ngp_status {
        if (idx.empty()) return NGP_OK;
        if ((int)idx.size() == B)   // the whole batch: the caller's arrays as they are
            return grad_leaf_stage(c, B, kernels, n, t, y, ldy, toep_path, leaf, yrows);
        std::vector<ngp_kernel> ks(idx.size());
        for (size_t a = 0; a < idx.size(); ++a) ks[a] = kernels[idx[a]];
        if (!yrows && ldy == 0)
            return grad_leaf_stage(c, (int32_t)idx.size(), ks.data(), n, t, y, 0, toep_path, leaf);
        std::vector<const double *> rows(idx.size());   // the leaf's rows where they lie
        for (size_t a = 0; a < idx.size(); ++a)
            rows[a] = yrows ? yrows[idx[a]] : y + (int64_t)idx[a] * ldy;
        return grad_leaf_stage(c, (int32_t)idx.size(), ks.data(), n, t, nullptr, 0, toep_path, leaf,
                               rows.data());
    };
    ngp_status st = stage_leaf(j->idx_gen, false, &j->gen);
    if (!st) st = stage_leaf(j->idx_toep, true, &j->toep);
    if (st) {
        grad_leaf_destroy(j->gen);
        grad_leaf_destroy(j->toep);
        return st;
    }
    j->n = n;
    const bool shared = !yrows && ldy == 0;
    if (keep_host && (size_t)(shared ? 1 : B) * (size_t)n * 8 <= GRAD_JOB_HOST_COPY_BYTES) {
        j->h_t.assign(t, t + n);
        j->h_ldy = shared ? 0 : n;
        j->h_y.resize((size_t)(shared ? 1 : B) * (size_t)n);
        for (int b = 0; b < (shared ? 1 : B); ++b)
            std::memcpy(j->h_y.data() + (size_t)b * n, yrows ? yrows[b] : y + (int64_t)b * ldy, 8 * (size_t)n);
        size_t nops = 0;
        for (int i = 0; i < B; ++i) nops += (size_t)kernels[i].n_ops;
        j->h_ops.reserve(nops);
        j->h_params.resize(std::max<size_t>(j->poff[(size_t)B], 1));
        j->h_k.resize((size_t)B);
        for (int i = 0; i < B; ++i) {
            j->h_ops.insert(j->h_ops.end(), kernels[i].ops, kernels[i].ops + kernels[i].n_ops);
            if (kernels[i].n_params > 0)
                std::memcpy(j->h_params.data() + j->poff[(size_t)i], kernels[i].params,
                            8 * (size_t)kernels[i].n_params);
        }
        size_t o = 0;
        for (int i = 0; i < B; ++i) {
            j->h_k[(size_t)i] = ngp_kernel{kernels[i].n_ops, kernels[i].n_params, j->h_ops.data() + o,
                                           j->h_params.data() + j->poff[(size_t)i], kernels[i].noise};
            o += (size_t)kernels[i].n_ops;
        }
    }
    *out = j.release();
    return NGP_OK;
}

extern "C" ngp_status ngp_grad_stage(ngp_ctx *c, int32_t B, const ngp_kernel *kernels, int32_t n,
                                     const double *t, const double *y, int64_t ldy,
                                     ngp_grad_job **out) {
    if (!y) return NGP_ERR_ARG;
    return grad_stage_impl(c, B, kernels, n, t, y, ldy, nullptr, out, true);
}

extern "C" ngp_status ngp_grad_job_set_params(ngp_grad_job *j, const double *params,
                                              const double *noise) {
    if (!j || !params || !noise) return NGP_ERR_ARG;
    auto set_leaf = [&](GradLeaf *leaf, const std::vector<int32_t> &idx) -> ngp_status {
        if (!leaf) return NGP_OK;
        if ((int)idx.size() == j->B) return grad_leaf_set_params(leaf, params, noise);
        j->buf_p.clear();
        j->buf_n.clear();
        for (int32_t i : idx) {
            j->buf_p.insert(j->buf_p.end(), params + j->poff[(size_t)i], params + j->poff[(size_t)i + 1]);
            j->buf_n.push_back(noise[i]);
        }
        if (j->buf_p.empty()) j->buf_p.push_back(0.0);
        return grad_leaf_set_params(leaf, j->buf_p.data(), j->buf_n.data());
    };
    ngp_status st = set_leaf(j->gen, j->idx_gen);
    if (!st) st = set_leaf(j->toep, j->idx_toep);
    if (!st && !j->h_k.empty()) {   // the one-shot form of the job follows
        if (j->poff[(size_t)j->B] > 0) std::memcpy(j->h_params.data(), params, 8 * j->poff[(size_t)j->B]);
        for (int i = 0; i < j->B; ++i) j->h_k[(size_t)i].noise = noise[i];
    }
    return st;
}

// the run on the job's resident inputs (ngp_grad_job_run, defined in the "combining" section, comes
// here when the job is alone)
static ngp_status grad_job_run_resident(ngp_grad_job *j, double *logml, double *grad, int32_t *info) {
    if (!j || !grad) return NGP_ERR_ARG;
    auto scatter = [&](const std::vector<int32_t> &idx, const std::vector<double> &lm,
                       const std::vector<double> &g, const std::vector<int32_t> &inf) {
        size_t off = 0;
        for (size_t a = 0; a < idx.size(); ++a) {
            const size_t i = (size_t)idx[a], len = j->goff[i + 1] - j->goff[i];
            std::memcpy(grad + j->goff[i], g.data() + off, 8 * len);
            off += len;
            if (logml) logml[i] = lm[a];
            if (info) info[i] = inf[a];
        }
    };
    if (j->side_by_side && j->gen && j->toep) {
        auto size_bufs = [&](const std::vector<int32_t> &idx, std::vector<double> &lm,
                             std::vector<double> &g, std::vector<int32_t> &inf) {
            size_t ng = 0;
            for (int32_t i : idx) ng += j->goff[(size_t)i + 1] - j->goff[(size_t)i];
            lm.resize(idx.size());
            inf.resize(idx.size());
            g.resize(ng);
        };
        size_bufs(j->idx_gen, j->buf_lm, j->buf_g, j->buf_info);
        size_bufs(j->idx_toep, j->buf_lm2, j->buf_g2, j->buf_info2);
        const ngp_status st = grad_pair_run(j->gen, j->buf_lm.data(), j->buf_g.data(), j->buf_info.data(),
                                            j->toep, j->buf_lm2.data(), j->buf_g2.data(),
                                            j->buf_info2.data());
        if (st == NGP_OK) {
            scatter(j->idx_gen, j->buf_lm, j->buf_g, j->buf_info);
            scatter(j->idx_toep, j->buf_lm2, j->buf_g2, j->buf_info2);
            return NGP_OK;
        }
        if (st != NGP_ERR_TOO_LARGE) return st;   // no room for both at once: one after the other
    }
    auto run_leaf = [&](GradLeaf *leaf, const std::vector<int32_t> &idx) -> ngp_status {
        if (!leaf) return NGP_OK;
        if ((int)idx.size() == j->B) return grad_leaf_run(leaf, logml, grad, info);
        size_t ng = 0;
        for (int32_t i : idx) ng += j->goff[(size_t)i + 1] - j->goff[(size_t)i];
        j->buf_lm.resize(idx.size());
        j->buf_info.resize(idx.size());
        j->buf_g.resize(ng);
        ngp_status st = grad_leaf_run(leaf, j->buf_lm.data(), j->buf_g.data(), j->buf_info.data());
        if (st) return st;
        size_t off = 0;
        for (size_t a = 0; a < idx.size(); ++a) {
            const size_t i = (size_t)idx[a], len = j->goff[i + 1] - j->goff[i];
            std::memcpy(grad + j->goff[i], j->buf_g.data() + off, 8 * len);
            off += len;
            if (logml) logml[i] = j->buf_lm[a];
            if (info) info[i] = j->buf_info[a];
        }
        return NGP_OK;
    };
    ngp_status st = run_leaf(j->gen, j->idx_gen);
    if (!st) st = run_leaf(j->toep, j->idx_toep);
    return st;
}

extern "C" ngp_status ngp_grad_job_info(const ngp_grad_job *j, int32_t *out5) {
    if (!j || !out5) return NGP_ERR_ARG;
    out5[0] = j->gen ? j->gen->B : 0;
    out5[1] = j->gen ? j->gen->last_chunk : 0;
    out5[2] = j->toep ? j->toep->B : 0;
    out5[3] = j->toep ? j->toep->last_chunk : 0;
    out5[4] = j->side_by_side ? 1 : 0;
    return NGP_OK;
}

extern "C" void ngp_grad_job_destroy(ngp_grad_job *j) {
    if (!j) return;
    grad_leaf_destroy(j->gen);
    grad_leaf_destroy(j->toep);
    delete j;
}

static ngp_status logml_grad_direct(ngp_ctx *c, int32_t B, const ngp_kernel *kernels, int32_t n,
                                    const double *t, const double *y, int64_t ldy,
                                    const double *const *yrows, double *logml, double *grad,
                                    int32_t *info) {
    if (!grad) return NGP_ERR_ARG;
    ngp_grad_job *j = nullptr;
    ngp_status st = grad_stage_impl(c, B, kernels, n, t, y, ldy, yrows, &j);
    if (st) return st;
    st = grad_job_run_resident(j, logml, grad, info);
    ngp_grad_job_destroy(j);
    return st;
}

// one column of a [P x ld]-strided log-weight matrix (ld = 1: a plain vector)
static void weights_normalize_strided(int P, const double *logw, long ld, double *w_norm, long ldw,
                                      double *ess, double *log_norm) {
    // a particle whose factorisation failed carries -inf (or NaN after -inf - -inf): weight 0, it
    // must not poison the others
    double mx = -INFINITY;
    for (int i = 0; i < P; ++i)
        if (std::isfinite(logw[i * ld]) && logw[i * ld] > mx) mx = logw[i * ld];
    if (!(mx > -INFINITY)) {
        if (ess) *ess = NAN;
        if (log_norm) *log_norm = -INFINITY;
        if (w_norm) for (int i = 0; i < P; ++i) w_norm[i * ldw] = NAN;
        return;
    }
    auto e = [&](int i) { return std::isfinite(logw[i * ld]) ? std::exp(logw[i * ld] - mx) : 0.0; };
    double sum = 0.0;
    for (int i = 0; i < P; ++i) sum += e(i);
    double sq = 0.0;
    for (int i = 0; i < P; ++i) {
        const double w = e(i) / sum;
        if (w_norm) w_norm[i * ldw] = w;
        sq += w * w;
    }
    if (ess) *ess = 1.0 / sq;
    if (log_norm) *log_norm = mx + std::log(sum);
}

extern "C" ngp_status ngp_weights_normalize(int32_t P, const double *logw, double *w_norm,
                                            double *ess, double *log_norm) {
    if (P <= 0 || !logw) return NGP_ERR_ARG;
    weights_normalize_strided(P, logw, 1, w_norm, 1, ess, log_norm);
    return NGP_OK;
}

extern "C" ngp_status ngp_weights_normalize_cols(int32_t P, int32_t D, const double *logw,
                                                 double *w_norm, double *ess, double *log_norm) {
    if (P <= 0 || D <= 0 || !logw) return NGP_ERR_ARG;
    for (int s = 0; s < D; ++s)
        weights_normalize_strided(P, logw + s, D, w_norm ? w_norm + s : nullptr, D,
                                  ess ? ess + s : nullptr, log_norm ? log_norm + s : nullptr);
    return NGP_OK;
}

// shared body of ngp_mixture_sample (seeds == nullptr: S mixtures over the SAME P components,
// mu [P x S x m], sigma [P x m x m], one key) and ngp_mixture_sample_indep (seeds [S]: every
// mixture has its own P components, mu [S x P x m], sigma [S x P x m x m], its own key)
static ngp_status mixture_sample_impl(ngp_ctx *c, int32_t P, int32_t S, int32_t m, const double *w,
                                      const double *mu, const double *sigma, int32_t draws,
                                      uint64_t seed, const uint64_t *seeds, double *out,
                                      int32_t *comp, int32_t *info) {
    if (!c || !w || !mu || !sigma || !out || P <= 0 || S <= 0 || m <= 0 || draws <= 0)
        return NGP_ERR_ARG;
    if (m > NGP_MAX_AUX) return NGP_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t ncomp_mats = seeds ? (size_t)S * P : (size_t)P;
    const size_t nw = (size_t)S * P, nmu = (size_t)P * S * m, nsg = ncomp_mats * m * m,
                 nout = (size_t)S * draws * m, ncomp = (size_t)S * draws;
    void *dw = nullptr, *dmu = nullptr, *dsg = nullptr, *dout = nullptr, *dcomp = nullptr,
         *dinfo = nullptr, *dseeds = nullptr;
    auto freeall = [&] {
        c->release(dw); c->release(dmu); c->release(dsg); c->release(dout); c->release(dcomp);
        c->release(dinfo); c->release(dseeds);
    };
    ngp_status st;
    if ((st = c->alloc(&dw, 8 * nw)) || (st = c->alloc(&dmu, 8 * nmu)) ||
        (st = c->alloc(&dsg, 8 * nsg)) || (st = c->alloc(&dout, 8 * nout)) ||
        (st = c->alloc(&dcomp, 4 * ncomp)) || (st = c->alloc(&dinfo, 4 * ncomp_mats)) ||
        (seeds && (st = c->alloc(&dseeds, 8 * (size_t)S)))) {
        freeall();
        return st;
    }
    hipError_t e = hipMemcpyAsync(dw, w, 8 * nw, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dmu, mu, 8 * nmu, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dsg, sigma, 8 * nsg, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && seeds)
        e = hipMemcpyAsync(dseeds, seeds, 8 * (size_t)S, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        launch_mixture_sample(P, S, m, (const double *)dw, (const double *)dmu, (double *)dsg, draws,
                              seed, (const uint64_t *)dseeds, (double *)dout, (int32_t *)dcomp,
                              (int32_t *)dinfo, s);
        e = hipMemcpyAsync(out, dout, 8 * nout, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess && comp) e = hipMemcpyAsync(comp, dcomp, 4 * ncomp, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && info)
        e = hipMemcpyAsync(info, dinfo, 4 * ncomp_mats, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    freeall();
    return e == hipSuccess ? NGP_OK : (ngp_status)e;
}

extern "C" ngp_status ngp_mixture_sample_indep(ngp_ctx *c, int32_t P, int32_t S, int32_t m,
                                               const double *w, const double *mu,
                                               const double *sigma, int32_t draws,
                                               const uint64_t *seeds, double *out, int32_t *comp,
                                               int32_t *info) {
    if (!seeds) return NGP_ERR_ARG;
    return mixture_sample_impl(c, P, S, m, w, mu, sigma, draws, 0, seeds, out, comp, info);
}

// ---------------------------------------------------------------------------------------
// Combining concurrent callers (flat combining)
//
// The reference enters the boundary from one task per nowcast scenario (Threads.@spawn,
// src/forecasting.jl:131-159): D tasks, each with the P particles of its own clone, all on the same
// dates.  Serialised on the context they are D calls of P items — the small-batch regime D times
// over.  Here a one-shot call (ngp_logml_batch, ngp_predict_batch, ngp_logml_grad_batch,
// ngp_mixture_sample with S = 1) is a REQUEST: it joins `pending`; whoever finds nobody serving takes
// everything that is pending, sorts it into groups of compatible requests — same entry point, same
// series length, bytewise identical dates (a hash, then memcmp), same forecast dates / flags — and
// runs every group as ONE launch sequence over the concatenated items with per-item observation
// rows (the `ldy = n` form every entry point already has; a group of one takes the unchanged
// one-shot path), scatters the results and wakes the callers.  No timer and no extra thread:
// requests pile up by themselves while the device is busy with the previous sequence.  A server
// that has finished its own request hands the role to the first waiter, so nobody serves forever.
// Requests that are not compatible run in arrival order, as before; staged jobs, resident factors
// and the collective do not combine (they take ctx->mu as they always did).
// A merged group that fails as a whole (device memory, a HIP error) is re-run request by request,
// so every caller gets the status its own call would have had.
// ---------------------------------------------------------------------------------------
enum { CK_LOGML = 0, CK_PREDICT = 1, CK_GRAD = 2, CK_MIXTURE = 3 };
struct ngp_comb_req {
    int kind = CK_LOGML;
    // CK_LOGML / CK_PREDICT / CK_GRAD
    int32_t B = 0, n = 0, m = 0, noise_on_new = 0;
    const ngp_kernel *k = nullptr;
    const double *t = nullptr, *y = nullptr, *t_new = nullptr;
    int64_t ldy = 0;
    double *logml = nullptr, *grad = nullptr, *mu = nullptr, *sigma = nullptr;
    int32_t *info = nullptr;
    ngp_grad_job *gj = nullptr;   // CK_GRAD from ngp_grad_job_run: alone it runs on its resident inputs
    // CK_MIXTURE (one mixture of P components: ngp_mixture_sample with S = 1)
    int32_t P = 0, draws = 0;
    const double *w = nullptr, *cmu = nullptr, *csigma = nullptr;
    uint64_t seed = 0;
    double *out = nullptr;
    int32_t *comp = nullptr;
    // what must be equal for two requests to share a launch sequence, hashed by the caller
    uint64_t key = 0;
    ngp_status st = NGP_OK;
    bool done = false;
    std::condition_variable cv;
};

namespace {

inline uint64_t fnv_words(uint64_t h, const void *p, size_t bytes) {
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

uint64_t comb_key(const ngp_comb_req &r) {
    uint64_t h = 0xcbf29ce484222325ull;
    const int32_t head[6] = {r.kind, r.n, r.m, r.noise_on_new, r.P, r.draws};
    h = fnv_words(h, head, sizeof(head));
    if (r.kind != CK_MIXTURE) {
        h = fnv_words(h, r.t, 8 * (size_t)r.n);
        if (r.m > 0) h = fnv_words(h, r.t_new, 8 * (size_t)r.m);
    }
    return h;
}

bool comb_same(const ngp_comb_req &a, const ngp_comb_req &b) {
    if (a.kind != b.kind || a.key != b.key || a.n != b.n || a.m != b.m ||
        a.noise_on_new != b.noise_on_new || a.P != b.P || a.draws != b.draws)
        return false;
    if (a.kind == CK_MIXTURE) return true;
    if (std::memcmp(a.t, b.t, 8 * (size_t)a.n) != 0) return false;
    return a.m == 0 || std::memcmp(a.t_new, b.t_new, 8 * (size_t)a.m) == 0;
}

// the request alone: the one-shot path as it always was
ngp_status comb_run_one(ngp_ctx *c, ngp_comb_req &r) {
    switch (r.kind) {
    case CK_LOGML: return logml_direct(c, r.B, r.k, r.n, r.t, r.y, r.ldy, r.logml, r.info);
    case CK_PREDICT:
        return predict_direct(c, r.B, r.k, r.n, r.t, r.y, r.ldy, r.m, r.t_new, r.noise_on_new, r.mu,
                              r.sigma, r.logml, r.info);
    case CK_GRAD:
        if (r.gj) return grad_job_run_resident(r.gj, r.logml, r.grad, r.info);
        return logml_grad_direct(c, r.B, r.k, r.n, r.t, r.y, r.ldy, nullptr, r.logml, r.grad, r.info);
    default:
        return mixture_sample_impl(c, r.P, 1, r.m, r.w, r.cmu, r.csigma, r.draws, r.seed, nullptr,
                                   r.out, r.comp, r.info);
    }
}

// a group of compatible requests as ONE call; the results go back to every request's own arrays
ngp_status comb_run_group(ngp_ctx *c, const std::vector<ngp_comb_req *> &grp, bool *shared_k) {
    const ngp_comb_req &r0 = *grp[0];
    *shared_k = false;
    if (r0.kind == CK_MIXTURE) {
        // K mixtures of P components each = ngp_mixture_sample_indep with one seed per request
        const size_t K = grp.size(), P = (size_t)r0.P, m = (size_t)r0.m, dr = (size_t)r0.draws;
        std::vector<double> w(K * P), mu(K * P * m), sg(K * P * m * m), out(K * dr * m);
        std::vector<uint64_t> seeds(K);
        std::vector<int32_t> comp(K * dr), info(K * P);
        for (size_t a = 0; a < K; ++a) {
            std::memcpy(w.data() + a * P, grp[a]->w, 8 * P);
            std::memcpy(mu.data() + a * P * m, grp[a]->cmu, 8 * P * m);
            std::memcpy(sg.data() + a * P * m * m, grp[a]->csigma, 8 * P * m * m);
            seeds[a] = grp[a]->seed;
        }
        const ngp_status st = mixture_sample_impl(c, r0.P, (int32_t)K, r0.m, w.data(), mu.data(), sg.data(),
                                                  r0.draws, 0, seeds.data(), out.data(), comp.data(),
                                                  info.data());
        if (st) return st;
        for (size_t a = 0; a < K; ++a) {
            std::memcpy(grp[a]->out, out.data() + a * dr * m, 8 * dr * m);
            if (grp[a]->comp) std::memcpy(grp[a]->comp, comp.data() + a * dr, 4 * dr);
            if (grp[a]->info) std::memcpy(grp[a]->info, info.data() + a * P, 4 * P);
        }
        return NGP_OK;
    }
    // ---- the scenario tasks of the reference's DEFAULT mode (n_mcmc = n_hmc = 0, src/forecasting.jl:120):
    // every task holds a clone of the same model — the same trees and parameters — on the same
    // dates, and its observations differ from the others' only in the appended nowcast points
    // (src/create_nowcast_data.jl:36-37).  K does not depend on y: such a group is ONE factorisation
    // per particle with the tasks' appended observations as scenarios — the shared-K form of
    // ngp_nowcast_batch (SURVEY.md section 8d "dedupe rule") — instead of one per (particle, task).
    // Recognised, not assumed: same kernels byte for byte, one y per request, a common prefix of the
    // observations up to a few last points.  Not under ngp_set_batch_invariant (another route to the
    // same numbers: equal to rounding, not bit for bit).
    bool invariant;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        invariant = c->invariant;
    }
    if ((r0.kind == CK_LOGML || r0.kind == CK_PREDICT) && !invariant) {
        constexpr int SHARED_K_MAX_TAIL = 16;
        bool same = true;
        for (size_t a = 1; a < grp.size() && same; ++a) {
            const ngp_comb_req &r = *grp[a];
            same = r.B == r0.B && r.ldy == 0 && r0.ldy == 0;
            for (int b = 0; same && b < r.B; ++b) {
                const ngp_kernel &ka = r0.k[b], &kb = r.k[b];
                same = ka.n_ops == kb.n_ops && ka.n_params == kb.n_params && ka.noise == kb.noise &&
                       std::memcmp(ka.ops, kb.ops, 4 * (size_t)ka.n_ops) == 0 &&
                       (ka.n_params == 0 || std::memcmp(ka.params, kb.params, 8 * (size_t)ka.n_params) == 0);
            }
        }
        int npre = r0.n;   // observations every request shares
        for (size_t a = 1; a < grp.size() && same; ++a) {
            int i = 0;
            while (i < npre && grp[a]->y[i] == r0.y[i]) ++i;
            npre = i;
        }
        int dt = std::max(r0.n - npre, 1);
        npre = r0.n - dt;
        if (same && grp[0]->ldy == 0 && dt <= SHARED_K_MAX_TAIL && npre >= 1 &&
            (npre % NB) + dt + r0.m + 1 <= NGP_MAX_AUX) {
            const size_t P = (size_t)r0.B, D = grp.size(), m = (size_t)r0.m;
            std::vector<double> yadd(D * (size_t)dt), lf(P * D), mu(P * D * m), sg(P * m * m);
            std::vector<int32_t> info(P);
            for (size_t a = 0; a < D; ++a) std::memcpy(yadd.data() + a * dt, grp[a]->y + npre, 8 * (size_t)dt);
            ngp_job *job = nullptr;
            ngp_status st = stage_general(c, (int)P, r0.k, npre, r0.t, r0.y, 0, dt, r0.t + npre, (int)D,
                                          yadd.data(), 0, r0.m, r0.t_new, r0.noise_on_new, &job);
            if (!st) {
                st = ngp_job_run(job);
                if (!st) st = ngp_job_fetch(job, nullptr, lf.data(), m ? mu.data() : nullptr,
                                            m ? sg.data() : nullptr, info.data());
                ngp_job_destroy(job);
            }
            if (st) return st;
            for (size_t a = 0; a < D; ++a) {
                ngp_comb_req *r = grp[a];
                for (size_t b = 0; b < P; ++b) {
                    if (r->logml) r->logml[b] = lf[b * D + a];
                    if (m && r->mu) std::memcpy(r->mu + b * m, mu.data() + (b * D + a) * m, 8 * m);
                }
                if (m && r->sigma) std::memcpy(r->sigma, sg.data(), 8 * P * m * m);
                if (r->info) std::memcpy(r->info, info.data(), 4 * P);
            }
            *shared_k = true;
            return NGP_OK;
        }
    }
    size_t Bt = 0;
    for (const ngp_comb_req *r : grp) Bt += (size_t)r->B;
    if (Bt > (size_t)0x7fffffff) return NGP_ERR_TOO_LARGE;
    std::vector<ngp_kernel> ks;
    std::vector<const double *> rows;
    ks.reserve(Bt);
    rows.reserve(Bt);
    for (const ngp_comb_req *r : grp)
        for (int b = 0; b < r->B; ++b) {
            ks.push_back(r->k[b]);
            rows.push_back(r->y + (int64_t)b * r->ldy);
        }
    std::vector<double> lm(Bt);
    std::vector<int32_t> info(Bt);
    if (r0.kind == CK_GRAD) {
        size_t ng = 0;
        for (const ngp_kernel &k : ks) ng += (size_t)k.n_params + 1;
        std::vector<double> grad(ng);
        const ngp_status st = logml_grad_direct(c, (int32_t)Bt, ks.data(), r0.n, r0.t, nullptr, 0,
                                                rows.data(), lm.data(), grad.data(), info.data());
        if (st) return st;
        size_t b0 = 0, g0 = 0;
        for (ngp_comb_req *r : grp) {
            size_t len = 0;
            for (int b = 0; b < r->B; ++b) len += (size_t)r->k[b].n_params + 1;
            std::memcpy(r->grad, grad.data() + g0, 8 * len);
            if (r->logml) std::memcpy(r->logml, lm.data() + b0, 8 * (size_t)r->B);
            if (r->info) std::memcpy(r->info, info.data() + b0, 4 * (size_t)r->B);
            b0 += (size_t)r->B;
            g0 += len;
        }
        return NGP_OK;
    }
    const size_t m = (size_t)r0.m;
    std::vector<double> mu(Bt * m), sg(Bt * m * m);
    static const double dummy = 0.0;
    ngp_job *job = nullptr;
    ngp_status st = stage_general(c, (int)Bt, ks.data(), r0.n, r0.t, nullptr, 0, 0, &dummy, 1, &dummy, 0,
                                  r0.m, r0.t_new, r0.noise_on_new, &job, rows.data());
    if (st) return st;
    st = ngp_job_run(job);
    if (!st) st = ngp_job_fetch(job, nullptr, lm.data(), m ? mu.data() : nullptr,
                                m ? sg.data() : nullptr, info.data());
    ngp_job_destroy(job);
    if (st) return st;
    size_t b0 = 0;
    for (ngp_comb_req *r : grp) {
        const size_t B = (size_t)r->B;
        if (r->logml) std::memcpy(r->logml, lm.data() + b0, 8 * B);
        if (r->info) std::memcpy(r->info, info.data() + b0, 4 * B);
        if (m && r->mu) std::memcpy(r->mu, mu.data() + b0 * m, 8 * B * m);
        if (m && r->sigma) std::memcpy(r->sigma, sg.data() + b0 * m * m, 8 * B * m * m);
        b0 += B;
    }
    return NGP_OK;
}

// everything one server took from `pending`: groups in order of their first member's arrival
void comb_execute(ngp_ctx *c, const std::vector<ngp_comb_req *> &batch, int64_t stats[6]) {
    std::vector<char> taken(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        std::vector<ngp_comb_req *> grp{batch[i]};
        taken[i] = 1;
        for (size_t j = i + 1; j < batch.size(); ++j)
            if (!taken[j] && comb_same(*batch[i], *batch[j])) {
                grp.push_back(batch[j]);
                taken[j] = 1;
            }
        stats[0] += (int64_t)grp.size();
        if (grp.size() > 1) {
            bool shared_k = false;
            const ngp_status st = comb_run_group(c, grp, &shared_k);
            if (st == NGP_OK) {
                for (ngp_comb_req *r : grp) r->st = NGP_OK;
                if (shared_k) stats[4] += (int64_t)grp.size();
                stats[1] += 1;
                stats[2] = std::max<int64_t>(stats[2], (int64_t)grp.size());
                stats[3] += (int64_t)grp.size();
                continue;
            }
        }
        for (ngp_comb_req *r : grp) {   // alone, or the merged call failed: each for itself
            r->st = comb_run_one(c, *r);
            stats[1] += 1;
            stats[2] = std::max<int64_t>(stats[2], 1);
        }
    }
}

// A caller that finds nobody serving, but whose predecessors came in company, gives that company
// this long to arrive before it serves (tasks that were woken together by the previous sequence do
// their host work and come back within microseconds of each other: without the wait the first one
// back runs alone and the convoy alternates 1, T-1, 1, T-1 ...).  Bounded, and never paid twice in
// vain: a wait that ends short of the expected company lowers the expectation to what did arrive,
// so a caller that is alone waits at most once after a concurrent phase and never otherwise.
constexpr int COMB_LINGER_US = 200;

ngp_status combine_submit(ngp_ctx *c, ngp_comb_req &r) {
    r.key = comb_key(r);   // outside the lock: 16 KB of dates at n = 2048
    std::unique_lock<std::mutex> q(c->qmu);
    if (!c->combine_on) {
        q.unlock();
        return comb_run_one(c, r);
    }
    c->pending.push_back(&r);
    c->comb_arrival.notify_all();
    // Invariant: a request that is not done is either in `pending` or in the batch of the one
    // serving thread (combining == true).  So a thread that finds nobody serving finds its own
    // request in `pending`.
    while (!r.done) {
        if (c->combining) {
            r.cv.wait(q);
            continue;
        }
        c->combining = true;
        if (c->combine_linger && c->pending.size() < c->comb_expect) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(COMB_LINGER_US);
            (void)c->comb_arrival.wait_until(q, deadline, [&] { return c->pending.size() >= c->comb_expect; });
        }
        std::vector<ngp_comb_req *> batch;
        batch.swap(c->pending);
        c->comb_expect = std::max<size_t>(batch.size(), 1);
        q.unlock();
        int64_t stats[6] = {0, 0, 0, 0, 0, 0};
        comb_execute(c, batch, stats);
        q.lock();
        c->comb_stats[0] += stats[0];
        c->comb_stats[1] += stats[1];
        c->comb_stats[2] = std::max(c->comb_stats[2], stats[2]);
        c->comb_stats[3] += stats[3];
        c->comb_stats[4] += stats[4];
        // (qmu is held from `done = true` to the notify: a woken caller cannot leave — and take its
        // request off its stack — before this loop is through with it)
        for (ngp_comb_req *x : batch) {
            x->done = true;
            if (x != &r) x->cv.notify_one();
        }
        c->combining = false;
        // what arrived meanwhile is served by its first waiter, not by this thread
        if (!c->pending.empty()) c->pending.front()->cv.notify_one();
    }
    return r.st;
}

}  // namespace

extern "C" ngp_status ngp_set_combining(ngp_ctx *c, int32_t on) {
    if (!c) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->qmu);
    c->combine_on = on != 0;
    c->combine_linger = on != 2;   // 2: combine what is pending, never wait for company
    return NGP_OK;
}

extern "C" ngp_status ngp_combine_stats(ngp_ctx *c, int64_t *out6, int32_t reset) {
    if (!c || !out6) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->qmu);
    for (int i = 0; i < 6; ++i) out6[i] = c->comb_stats[i];
    if (reset) for (int i = 0; i < 6; ++i) c->comb_stats[i] = 0;
    return NGP_OK;
}

// Arguments the merged form could not carry (null arrays, a malformed tree, empty sizes) never
// join a group: the direct path reports them exactly as before.
static bool comb_value_args_ok(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n, const double *t,
                               const double *y) {
    if (!c || !k || !t || !y || B <= 0 || n <= 0) return false;
    for (int i = 0; i < B; ++i)
        if (check_program(&k[i]) != NGP_OK) return false;
    return true;
}

extern "C" ngp_status ngp_logml_batch(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                                      const double *t, const double *y, int64_t ldy, double *logml,
                                      int32_t *info) {
    if (!comb_value_args_ok(c, B, k, n, t, y)) return logml_direct(c, B, k, n, t, y, ldy, logml, info);
    ngp_comb_req r;
    r.kind = CK_LOGML;
    r.B = B; r.k = k; r.n = n; r.t = t; r.y = y; r.ldy = ldy;
    r.logml = logml; r.info = info;
    return combine_submit(c, r);
}

extern "C" ngp_status ngp_predict_batch(ngp_ctx *c, int32_t B, const ngp_kernel *k, int32_t n,
                                        const double *t, const double *y, int64_t ldy, int32_t m,
                                        const double *t_new, int32_t noise_on_new, double *mu,
                                        double *sigma, double *logml, int32_t *info) {
    if (!comb_value_args_ok(c, B, k, n, t, y) || m <= 0 || !t_new)
        return predict_direct(c, B, k, n, t, y, ldy, m, t_new, noise_on_new, mu, sigma, logml, info);
    ngp_comb_req r;
    r.kind = CK_PREDICT;
    r.B = B; r.k = k; r.n = n; r.t = t; r.y = y; r.ldy = ldy;
    r.m = m; r.t_new = t_new; r.noise_on_new = noise_on_new ? 1 : 0;
    r.mu = mu; r.sigma = sigma; r.logml = logml; r.info = info;
    return combine_submit(c, r);
}

extern "C" ngp_status ngp_logml_grad_batch(ngp_ctx *c, int32_t B, const ngp_kernel *kernels,
                                           int32_t n, const double *t, const double *y,
                                           int64_t ldy, double *logml, double *grad,
                                           int32_t *info) {
    if (!grad || !comb_value_args_ok(c, B, kernels, n, t, y))
        return logml_grad_direct(c, B, kernels, n, t, y, ldy, nullptr, logml, grad, info);
    ngp_comb_req r;
    r.kind = CK_GRAD;
    r.B = B; r.k = kernels; r.n = n; r.t = t; r.y = y; r.ldy = ldy;
    r.logml = logml; r.grad = grad; r.info = info;
    return combine_submit(c, r);
}

// A resident job that still has the one-shot form of its inputs (small jobs) is a gradient request
// like any other: with company it shares one call over everybody's items, alone it runs on its
// resident inputs.
extern "C" ngp_status ngp_grad_job_run(ngp_grad_job *j, double *logml, double *grad, int32_t *info) {
    if (!j || !grad) return NGP_ERR_ARG;
    if (j->h_k.empty()) return grad_job_run_resident(j, logml, grad, info);
    ngp_comb_req r;
    r.kind = CK_GRAD;
    r.gj = j;
    r.B = j->B; r.k = j->h_k.data(); r.n = j->n; r.t = j->h_t.data();
    r.y = j->h_y.data(); r.ldy = j->h_ldy;
    r.logml = logml; r.grad = grad; r.info = info;
    return combine_submit(j->ctx, r);
}

extern "C" ngp_status ngp_mixture_sample(ngp_ctx *c, int32_t P, int32_t S, int32_t m,
                                         const double *w, const double *mu, const double *sigma,
                                         int32_t draws, uint64_t seed, double *out, int32_t *comp,
                                         int32_t *info) {
    // S mixtures over shared components stay one call of their own; ONE mixture (what
    // predict_mvn(...) |> rand of a scenario task asks for, src/forecasting.jl:46-47) can share a
    // launch with other tasks' mixtures of the same shape
    if (S != 1 || !c || !w || !mu || !sigma || !out || P <= 0 || m <= 0 || m > NGP_MAX_AUX || draws <= 0)
        return mixture_sample_impl(c, P, S, m, w, mu, sigma, draws, seed, nullptr, out, comp, info);
    ngp_comb_req r;
    r.kind = CK_MIXTURE;
    r.P = P; r.m = m; r.draws = draws; r.w = w;
    r.cmu = mu; r.csigma = sigma;
    r.seed = seed; r.out = out; r.comp = comp; r.info = info;
    return combine_submit(c, r);
}

// ---------------------------------------------------------------------------------------
// The one collective of the path (north_star: RCCL over xGMI only for the particle-weight
// normalisation / resample reduction), for hosts that bring no collective library of their own
// (the Julia shim; the Python mirror uses torch.distributed, whose "nccl" backend is this same
// RCCL).  librccl is opened at run time: the library has no link-time dependency on it and the
// entry points answer NGP_ERR_UNAVAILABLE when it is not installed.
// ---------------------------------------------------------------------------------------
namespace {
struct RcclUid { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUid *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    bool ok = false;
};
RcclApi load_rccl() {
    RcclApi r;
    // The RCCL that belongs to the HIP runtime THIS library is bound to, by path: a host process may
    // hold a second ROCm stack (PyTorch wheels bundle libamdhip64 / libhsa-runtime64 / librccl; when
    // torch is imported after libngp both stacks are mapped), and a bare dlopen("librccl.so.1")
    // then returns the copy already loaded — tied to the other, uninitialised runtime
    // ("no ROCm-capable device", scripts/rccl_probe.py).  The directory of the libamdhip64 that
    // resolved our own HIP calls decides.
    std::vector<std::string> names;
    Dl_info di{};
    if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            names.push_back(dir + "librccl.so.1");
            names.push_back(dir + "librccl.so");
        }
    }
    for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(nm);
    for (const std::string &name : names) {
        r.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    return r;
}
const RcclApi &rccl() {
    static const RcclApi api = load_rccl();
    return api;
}
constexpr int RCCL_FLOAT64 = 8;   // ncclFloat64 (rccl.h)
}  // namespace

struct ngp_comm {
    ngp_ctx *ctx = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1;
    // send / receive buffers of the all-gather, grow-only: after the first call of a shape no
    // rank allocates inside a collective (an allocation that fails on ONE rank would leave its
    // peers waiting in the all-gather)
    void *d_send = nullptr, *d_recv = nullptr;
    size_t cap = 0;   // doubles per rank
};

extern "C" ngp_status ngp_comm_unique_id(void *id128) {
    if (!id128) return NGP_ERR_ARG;
    if (!rccl().ok) return NGP_ERR_UNAVAILABLE;
    RcclUid uid{};
    if (rccl().GetUniqueId(&uid) != 0) return NGP_ERR_UNAVAILABLE;
    std::memcpy(id128, uid.internal, sizeof(uid.internal));
    return NGP_OK;
}

extern "C" ngp_status ngp_comm_create(ngp_ctx *c, const void *id128, int32_t rank, int32_t world,
                                      ngp_comm **out) {
    if (!c || !id128 || !out || world <= 0 || rank < 0 || rank >= world) return NGP_ERR_ARG;
    *out = nullptr;
    if (!rccl().ok) return NGP_ERR_UNAVAILABLE;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    RcclUid uid{};
    std::memcpy(uid.internal, id128, sizeof(uid.internal));
    ngp_comm *m = new (std::nothrow) ngp_comm();
    if (!m) return NGP_ERR_TOO_LARGE;
    m->ctx = c;
    m->rank = rank;
    m->world = world;
    // RCCL allocates its own device buffers, and after a large job most of the device can sit in
    // the caching allocators of this process: they are given back BEFORE the one attempt.  (The
    // initialisation is a collective — a rank that retried it alone after a failure would pair
    // with nobody, its peers having returned from the first attempt or still waiting inside it.)
    c->drop_cache();
    c->ws_drop();
    ngp_ctx::drop_other_caches(c);
    if (rccl().CommInitRank(&m->comm, world, uid, rank) != 0) {
        (void)hipGetLastError();
        delete m;
        return NGP_ERR_UNAVAILABLE;
    }
    *out = m;
    return NGP_OK;
}

extern "C" void ngp_comm_destroy(ngp_comm *m) {
    if (!m) return;
    if (m->comm && rccl().ok) {
        (void)hipSetDevice(m->ctx->device);
        (void)rccl().CommDestroy(m->comm);
    }
    {
        std::lock_guard<std::mutex> lk(m->ctx->mu);
        m->ctx->release(m->d_send);
        m->ctx->release(m->d_recv);
    }
    delete m;
}

extern "C" ngp_status ngp_shard(int32_t P_total, int32_t world, int32_t rank, int32_t *first,
                                int32_t *rows) {
    if (P_total < 0 || world <= 0 || rank < 0 || rank >= world) return NGP_ERR_ARG;
    const int base = P_total / world, rem = P_total % world;
    if (first) *first = rank * base + std::min(rank, rem);
    if (rows) *rows = base + (rank < rem ? 1 : 0);
    return NGP_OK;
}

// The host half of the exchange: `padded` holds what an all-gather of equally sized, zero-padded
// shards delivers — world blocks of pmax x D doubles, pmax = the largest shard (rank 0's), block r =
// rank r's rows then padding — and comes out as the P_total x D normalised weights.  Exported so that
// a host with a collective of its own (MPI.jl, Julia's Distributed, torch.distributed) pads,
// gathers with that, and normalises exactly as ngp_weights_allgather_normalize does.
extern "C" ngp_status ngp_weights_unpad_normalize(int32_t P_total, int32_t world, int32_t D,
                                                  const double *padded, double *w_all, double *ess,
                                                  double *log_norm) {
    if (!padded || P_total <= 0 || D <= 0 || world <= 0 || P_total < world) return NGP_ERR_ARG;
    auto rows_of = [&](int r) { int32_t v = 0; (void)ngp_shard(P_total, world, r, nullptr, &v); return (size_t)v; };
    auto first_of = [&](int r) { int32_t v = 0; (void)ngp_shard(P_total, world, r, &v, nullptr); return (size_t)v; };
    const size_t cnt = rows_of(0) * (size_t)D;
    std::vector<double> lw((size_t)P_total * D), wn((size_t)P_total * D);
    for (int r = 0; r < world; ++r)
        std::memcpy(lw.data() + first_of(r) * D, padded + (size_t)r * cnt, 8 * rows_of(r) * D);
    for (int sidx = 0; sidx < D; ++sidx)
        weights_normalize_strided(P_total, lw.data() + sidx, D, wn.data() + sidx, D,
                                  ess ? ess + sidx : nullptr, log_norm ? log_norm + sidx : nullptr);
    if (w_all) std::memcpy(w_all, wn.data(), 8 * wn.size());
    return NGP_OK;
}

extern "C" ngp_status ngp_weights_allgather_normalize(ngp_comm *m, int32_t P_total, int32_t D,
                                                      const double *logw_local, double *w_local,
                                                      double *w_all, double *ess,
                                                      double *log_norm) {
    if (!m || !logw_local || P_total <= 0 || D <= 0 || P_total < m->world) return NGP_ERR_ARG;
    // block partition of the particles over the ranks, remainder to the low ranks (ngp_shard: the
    // partition nowcastautogp_amd.distributed.shard and the Julia shim use)
    auto rows_of = [&](int r) { int32_t v = 0; (void)ngp_shard(P_total, m->world, r, nullptr, &v); return (int)v; };
    auto first_of = [&](int r) { int32_t v = 0; (void)ngp_shard(P_total, m->world, r, &v, nullptr); return (int)v; };
    const int mine = rows_of(m->rank), pmax = rows_of(0);
    ngp_ctx *c = m->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // ragged shards: every rank sends pmax rows (its own, then padding).  The buffers stay with the
    // communicator; every rank sees the same (P_total, D), so they grow on all ranks in the same
    // call — and a rank that cannot grow them has left the collective for good: any non-zero
    // return of this function is FATAL for the communicator (its peers may be inside the
    // all-gather), destroy it and start over.
    const size_t cnt = (size_t)pmax * D;
    if (cnt > m->cap) {
        void *a = nullptr, *b = nullptr;
        ngp_status st;
        if ((st = c->alloc(&a, 8 * cnt)) || (st = c->alloc(&b, 8 * cnt * (size_t)m->world))) {
            c->release(a);
            return st;
        }
        c->release(m->d_send);
        c->release(m->d_recv);
        m->d_send = a;
        m->d_recv = b;
        m->cap = cnt;
    }
    void *d_send = m->d_send, *d_recv = m->d_recv;
    std::vector<double> h_all(cnt * (size_t)m->world);
    hipError_t e = hipMemsetAsync(d_send, 0, 8 * cnt, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_send, logw_local, 8 * (size_t)mine * D, hipMemcpyHostToDevice, s);
    int rc = 0;
    if (e == hipSuccess) rc = rccl().AllGather(d_send, d_recv, cnt, RCCL_FLOAT64, m->comm, s);
    if (e == hipSuccess && rc == 0)
        e = hipMemcpyAsync(h_all.data(), d_recv, 8 * h_all.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(s);
    if (rc != 0) return NGP_ERR_UNAVAILABLE;
    if (e != hipSuccess) return (ngp_status)e;
    // compact the padded shards to [P_total x D], then the columns as ngp_weights_normalize_cols
    std::vector<double> wn((size_t)P_total * D);
    const ngp_status st = ngp_weights_unpad_normalize(P_total, m->world, D, h_all.data(), wn.data(), ess, log_norm);
    if (st) return st;
    if (w_all) std::memcpy(w_all, wn.data(), 8 * wn.size());
    if (w_local)
        std::memcpy(w_local, wn.data() + (size_t)first_of(m->rank) * D, 8 * (size_t)mine * D);
    return NGP_OK;
}

// ---------------------------------------------------------------------------------------
// microbenchmarks / self tests
// ---------------------------------------------------------------------------------------
extern "C" ngp_status ngp_microbench_mfma_f64(ngp_ctx *c, int32_t iters, double *tflops) {
    if (!c || !tflops || iters <= 0) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    iters = (iters + 3) / 4 * 4;
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU
    void *out = nullptr;
    ngp_status st = c->alloc(&out, sizeof(double) * (size_t)blocks * 256);
    if (st) return st;
    struct Rel { ngp_ctx *c; void *p; ~Rel() { c->release(p); } } rel{c, out};
    ScopedEvent a, b;
    launch_mfma_bench((double *)out, iters, blocks, c->stream);  // warm-up
    HIPCHK(hipEventRecord(a, c->stream));
    launch_mfma_bench((double *)out, iters, blocks, c->stream);
    HIPCHK(hipEventRecord(b, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *tflops = (double)blocks * 4.0 * (double)iters * 2048.0 / ((double)ms * 1e-3) * 1e-12;
    return NGP_OK;
}

// out[0] TFLOP/s (wall), out[1] median shader cycles per MFMA per wave, out[2] median effective
// shader clock in GHz (s_memtime / s_memrealtime), out[3] waves per SIMD used
extern "C" ngp_status ngp_microbench_mfma_f64_detail(ngp_ctx *c, int32_t iters,
                                                     int32_t blocks_per_cu, double *out) {
    if (!c || !out || iters <= 0 || blocks_per_cu <= 0 || blocks_per_cu > 8) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    iters = (iters + 3) / 4 * 4;
    const int blocks = 256 * blocks_per_cu;
    const int waves = blocks * 4;
    void *st = nullptr;
    ngp_status s0 = c->alloc(&st, sizeof(unsigned long long) * 2 * (size_t)waves);
    if (s0) return s0;
    struct Rel { ngp_ctx *c; void *p; ~Rel() { c->release(p); } } rel{c, st};
    ScopedEvent a, b;
    launch_mfma_bench_detail((unsigned long long *)st, iters, blocks, c->stream);
    HIPCHK(hipEventRecord(a, c->stream));
    launch_mfma_bench_detail((unsigned long long *)st, iters, blocks, c->stream);
    HIPCHK(hipEventRecord(b, c->stream));
    std::vector<unsigned long long> h(2 * (size_t)waves);
    HIPCHK(hipMemcpyAsync(h.data(), st, sizeof(unsigned long long) * h.size(),
                          hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> cyc((size_t)waves), ghz((size_t)waves);
    for (int w = 0; w < waves; ++w) {
        cyc[(size_t)w] = (double)h[2 * (size_t)w] / (double)iters;
        ghz[(size_t)w] = (double)h[2 * (size_t)w] / ((double)h[2 * (size_t)w + 1] * 10.0);
    }
    std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
    std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
    out[0] = (double)waves * (double)iters * 2048.0 / ((double)ms * 1e-3) * 1e-12;
    out[1] = cyc[(size_t)waves / 2];
    out[2] = ghz[(size_t)waves / 2];
    out[3] = (double)blocks_per_cu;
    return NGP_OK;
}

extern "C" ngp_status ngp_microbench_hbm(ngp_ctx *c, int64_t bytes, double *write_gbs,
                                         double *copy_gbs) {
    if (!c || bytes < 4096) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = bytes / 16 * 2;
    void *a = nullptr, *b = nullptr;
    ngp_status st = c->alloc(&a, (size_t)n * 8);
    if (st) return st;
    st = c->alloc(&b, (size_t)n * 8);
    if (st) { c->release(a); return st; }
    struct Rel { ngp_ctx *c; void *p, *q; ~Rel() { c->release(p); c->release(q); } } rel{c, a, b};
    ScopedEvent e0, e1, e2;
    launch_stream_write((double *)a, n, c->stream);
    launch_stream_copy((double *)b, (const double *)a, n, c->stream);
    HIPCHK(hipEventRecord(e0, c->stream));
    launch_stream_write((double *)a, n, c->stream);
    HIPCHK(hipEventRecord(e1, c->stream));
    launch_stream_copy((double *)b, (const double *)a, n, c->stream);
    HIPCHK(hipEventRecord(e2, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float w = 0.f, cp = 0.f;
    HIPCHK(hipEventElapsedTime(&w, e0, e1));
    HIPCHK(hipEventElapsedTime(&cp, e1, e2));
    if (write_gbs) *write_gbs = (double)n * 8.0 / ((double)w * 1e-3) * 1e-9;
    if (copy_gbs) *copy_gbs = 2.0 * (double)n * 8.0 / ((double)cp * 1e-3) * 1e-9;
    return NGP_OK;
}

// D[0:256]   = A(16x4) B(4x16) through one v_mfma_f64_16x16x4_f64 using the operand maps the
//              kernels assume
// D[256:512] = the same product through four DPP-rotated v_mfma_f64_4x4x4_4b_f64 + the gather back
//              to the 16x16x4 C/D layout (the fast path of the k-loops)
// The caller compares both with A @ B on asymmetric data — tests/test_gpu_parity.py.
extern "C" ngp_status ngp_selftest_mfma_layout(ngp_ctx *c, const double *A, const double *Bm,
                                               double *D) {
    if (!c || !A || !Bm || !D) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    void *da = nullptr, *db = nullptr, *dd = nullptr;
    ngp_status st;
    if ((st = c->alloc(&da, 64 * 8)) || (st = c->alloc(&db, 64 * 8)) ||
        (st = c->alloc(&dd, 512 * 8))) {
        c->release(da); c->release(db); c->release(dd);
        return st;
    }
    hipError_t e = hipMemcpyAsync(da, A, 64 * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db, Bm, 64 * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_mfma_layout_probe((const double *)da, (const double *)db, (double *)dd, c->stream);
        e = hipMemcpyAsync(D, dd, 512 * 8, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    c->release(da); c->release(db); c->release(dd);
    return e == hipSuccess ? NGP_OK : (ngp_status)e;
}

// D (32 x 32, row-major) = A (32 x 2) B (2 x 32) through one v_mfma_f32_32x32x2_f32 with the operand
// and result maps the mixed-precision k-loop assumes; the caller compares with A @ B.
extern "C" ngp_status ngp_selftest_mfma_f32_layout(ngp_ctx *c, const float *A, const float *Bm,
                                                   float *D) {
    if (!c || !A || !Bm || !D) return NGP_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    void *da = nullptr, *db = nullptr, *dd = nullptr;
    ngp_status st;
    if ((st = c->alloc(&da, 64 * 4)) || (st = c->alloc(&db, 64 * 4)) ||
        (st = c->alloc(&dd, 1024 * 4))) {
        c->release(da); c->release(db); c->release(dd);
        return st;
    }
    hipError_t e = hipMemcpyAsync(da, A, 64 * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db, Bm, 64 * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_mfma_f32_probe((const float *)da, (const float *)db, (float *)dd, c->stream);
        e = hipMemcpyAsync(D, dd, 1024 * 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    c->release(da); c->release(db); c->release(dd);
    return e == hipSuccess ? NGP_OK : (ngp_status)e;
}
