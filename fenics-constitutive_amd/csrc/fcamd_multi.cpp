// One process, several GPUs (include/fcamd.h, "one process, several GPUs"): the single assembler's host entry.
//
// north_star's single-process mode has ONE dolfinx process whose arrays are host memory
// (src/fenics_constitutive/solver/_lawonsubmesh.py:87-94: views of Function.x.array; the scatter after the law loop,
// solver/_solver.py:146-147) and several GPUs.  An all-gather over xGMI (fcamd_multigpu.cpp) leaves the results in
// HBM, one PCIe link away from the assembler.  Here every device works on ITS slice of the caller's host arrays
// itself: a fcamd_multi owns one context + model handle + worker thread per device; a call cuts [0, n) into
// contiguous, tile-aligned slices (the rule of fcamd_shard_bounds) and every worker runs the single-GPU host entry
// (fcamd_evaluate_host / fcamd_evaluate_resident: one zero-copy launch that reads and writes the page-locked caller
// arrays over PCIe) on its slice at the same time.  N PCIe links move data in parallel, nothing crosses xGMI, results
// land where the assembler reads them.
//
// Page locks: the coordinator (the calling thread) locks each caller array ONCE, whole, in the process-wide registry
// of fcamd_hostpath.cpp; the workers' host entries find their slices inside those locks (reference counts) and only
// ask for the address THEIR device sees the slice at.  Page-locked host memory is reachable from every device of the
// process.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include "fcamd_host.h"

using namespace fcamd;

namespace {

// One device slot: a thread that owns a context and a model handle on its device and runs jobs posted to it.
struct Worker {
    int device = 0;
    fcamd_context* ctx = nullptr;
    fcamd_model* model = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int status = FCAMD_OK;
    std::string error;

    void loop() {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (!has_job && quit) return;
                j = std::move(job);
                has_job = false;
            }
            int st;
            try {
                st = j();
            } catch (const std::exception& e) {  // e.g. std::bad_alloc of a host temporary: a status, not std::terminate
                st = fail(FCAMD_ERR_BAD_ARG, "device slot job failed: %s", e.what());
            }
            std::string msg = st == FCAMD_OK ? std::string() : std::string(fcamd_last_error());
            {
                std::lock_guard<std::mutex> lk(mu);
                status = st;
                error = std::move(msg);
                done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return status;
    }
};

constexpr int64_t kTile = 64;

int64_t slot_points(int64_t n, int world) {
    const int64_t per = (n + world - 1) / world;
    return ((per + kTile - 1) / kTile) * kTile;
}

void slice_of(int64_t n, int world, int k, int64_t* lo, int64_t* hi) {  // = fcamd_shard_bounds
    const int64_t per = slot_points(n, world);
    *lo = std::min<int64_t>((int64_t)k * per, n);
    *hi = std::min<int64_t>(*lo + per, n);
}

}  // namespace

struct fcamd_multi {
    std::vector<std::unique_ptr<Worker>> w;
    fcamd::LawInfo info{};
    fcamd::Dims dims{9, 6, 3};
    int law = 0;
    int last_mode = 0, last_used = 0;
    int64_t min_points = FCAMD_MULTI_MIN_POINTS;  // option "min_points": a call uses at most n / min_points devices
    std::mutex call_mu;  // one call at a time
    std::vector<fcamd_multi_state*> states;  // live resident states (destroyed with the handle if the caller did not)

    // number of devices a call over n points uses
    int used_for(int64_t n) const {
        const int64_t want = std::max<int64_t>(1, n / min_points);
        return (int)std::min<int64_t>((int64_t)w.size(), want);
    }

    // run job(k) on workers [0, used) concurrently; first failing status (its message becomes the caller's last error)
    int run(int used, const std::function<int(int)>& job) {
        for (int k = 0; k < used; ++k) w[k]->post([&job, k] { return job(k); });
        int first = FCAMD_OK;
        for (int k = 0; k < used; ++k) {
            const int st = w[k]->wait();
            if (st != FCAMD_OK && first == FCAMD_OK) first = fail(st, "%s [device slot %d, device %d]", w[k]->error.c_str(), k, w[k]->device);
        }
        return first;
    }
};

struct fcamd_multi_state {
    fcamd_multi* mg = nullptr;
    int64_t n = 0;
    int flags = 0;  // FCAMD_EVAL_SPLIT_HISTORY or 0
    int nh = 0;     // history arrays per device (the law's fields, or 2 when split)
    size_t hdim[FCAMD_MAX_HISTORY] = {0, 0};
    struct Slice {
        int64_t lo = 0, hi = 0;
        double* stress[2] = {nullptr, nullptr};
        double* hist[2][FCAMD_MAX_HISTORY] = {{nullptr, nullptr}, {nullptr, nullptr}};
        uint64_t* mask = nullptr;
    };
    std::vector<Slice> s;
    int committed = 0;      // index of the committed copy
    bool evaluated = false;  // an evaluate since the last commit / set
    int failed = FCAMD_OK;   // status of the last evaluate if it failed
};

namespace {

struct LockSet {  // call-scoped page locks held by the coordinator
    std::vector<char*> bases;
    void add(const void* p, size_t bytes) {
        if (!p || bytes == 0) return;
        char *base = nullptr, *dev = nullptr;
        // failure is not an error: the per-device entries then fall back to their scratch path
        if (temp_lock_acquire(static_cast<char*>(const_cast<void*>(p)), bytes, &base, &dev) && base) bases.push_back(base);
    }
    ~LockSet() {
        for (char* b : bases) temp_lock_release(b);
    }
};

bool inside_registered(const fcamd_context* c, const void* p, size_t bytes) {
    if (!p || c->registered.empty()) return false;
    char* q = static_cast<char*>(const_cast<void*>(p));
    auto it = c->registered.upper_bound(q);
    if (it == c->registered.begin()) return false;
    --it;
    return q + bytes <= it->first + it->second.bytes;
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return FCAMD_OK;
    (void)hipGetLastError();
    return fail(FCAMD_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
}

}  // namespace

extern "C" {

int fcamd_multi_create(const int* devices, int n_devices, int model_id, int constraint, const double* params,
                       int n_params, fcamd_multi** out) {
    if (!out) return fail(FCAMD_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > FCAMD_MULTI_MAX_DEVICES)
        return fail(FCAMD_ERR_BAD_ARG, "1 .. %d devices expected", FCAMD_MULTI_MAX_DEVICES);
    if (!params) return fail(FCAMD_ERR_BAD_ARG, "params is NULL");
    std::unique_ptr<fcamd_multi> mg(new (std::nothrow) fcamd_multi());
    if (!mg) return fail(FCAMD_ERR_BAD_ARG, "out of host memory");
    std::vector<double> p(params, params + std::max(n_params, 0));
    for (int k = 0; k < n_devices; ++k) {
        mg->w.emplace_back(new Worker());
        Worker* w = mg->w.back().get();
        w->device = devices[k];
        w->th = std::thread([w] { w->loop(); });
    }
    // every worker creates its context and model handle in its own thread (one context per thread)
    const int st = mg->run(n_devices, [&](int k) {
        Worker* w = mg->w[k].get();
        int s = fcamd_context_create(w->device, nullptr, &w->ctx);
        if (s != FCAMD_OK) return s;
        // the expansion threads of the host tangent (fcamd_hosttangent.cpp): the devices of one process share the process's CPUs
        if (w->ctx->opt.host_tangent_threads < 0) w->ctx->opt.host_tangent_threads = host_tangent_threads_shared(n_devices);
        return fcamd_model_create(w->ctx, model_id, constraint, p.data(), (int)p.size(), &w->model);
    });
    if (st != FCAMD_OK) {
        const std::string msg = fcamd_last_error();
        fcamd_multi_destroy(mg.release());
        return fail(st, "%s", msg.c_str());
    }
    const fcamd_model* m0 = mg->w[0]->model;
    mg->info = m0->info;
    mg->dims = m0->dims;
    mg->law = m0->law;
    *out = mg.release();
    return FCAMD_OK;
}

int fcamd_multi_destroy(fcamd_multi* mg) {
    if (!mg) return FCAMD_OK;
    while (!mg->states.empty()) fcamd_multi_state_destroy(mg->states.back());  // their arrays live in this handle's contexts
    for (auto& w : mg->w) {
        if (!w->th.joinable()) continue;
        Worker* wp = w.get();
        wp->post([wp] {
            if (wp->model) fcamd_model_destroy(wp->model);
            if (wp->ctx) fcamd_context_destroy(wp->ctx);
            wp->model = nullptr;
            wp->ctx = nullptr;
            return (int)FCAMD_OK;
        });
        wp->wait();
        {
            std::lock_guard<std::mutex> lk(wp->mu);
            wp->quit = true;
        }
        wp->cv.notify_all();
        wp->th.join();
    }
    delete mg;
    return FCAMD_OK;
}

int fcamd_multi_plan(const fcamd_multi* mg, int64_t n, int k, int* n_used, int64_t* lo, int64_t* hi) {
    if (!mg) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (n < 0) return fail(FCAMD_ERR_SIZE, "negative number of quadrature points");
    const int used = mg->used_for(n);
    if (n_used) *n_used = used;
    if (!lo && !hi) return FCAMD_OK;
    if (!lo || !hi) return fail(FCAMD_ERR_BAD_ARG, "lo and hi: both or none");
    if (k < 0 || k >= (int)mg->w.size()) return fail(FCAMD_ERR_BAD_ARG, "device slot %d out of range", k);
    if (k >= used) {
        *lo = *hi = n;
        return FCAMD_OK;
    }
    slice_of(n, used, k, lo, hi);
    return FCAMD_OK;
}

int fcamd_multi_evaluate_host(fcamd_multi* mg, double t, double del_t, int64_t n, const double* grad, double* stress,
                              double* tangent, double* const* hist, int n_hist, fcamd_stats* stats) {
    if (!mg) return fail(FCAMD_ERR_BAD_ARG, "multi handle is NULL");
    std::lock_guard<std::mutex> call(mg->call_mu);
    // the reference's argument checks once, for the whole call, with the single-GPU entry's statuses
    int st = validate_call(mg->w[0]->model, del_t, n, grad, stress, stress, reinterpret_cast<const void* const*>(hist),
                           reinterpret_cast<const void* const*>(hist), n_hist);
    if (st != FCAMD_OK) return st;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const size_t GD2 = (size_t)mg->dims.gd2, SD = (size_t)mg->dims.sd, TD = SD * SD;
    const int NH = mg->info.n_hist;
    const int used = mg->used_for(n);
    mg->last_used = used;
    mg->last_mode = 0;
    const size_t N = (size_t)n;
    size_t hist_doubles = 0;
    for (int k = 0; k < NH; ++k) hist_doubles += (size_t)mg->info.hist[k].dim;
    LockSet locks;
    if (used > 1) {
        const size_t total = N * (GD2 + SD + (tangent ? TD : 0) + hist_doubles) * sizeof(double);
        const fcamd_context* c0 = mg->w[0]->ctx;
        if (total > (size_t)c0->opt.bounce_max) {
            auto want = [&](const void* p, size_t bytes) {
                if (p && bytes && !inside_registered(c0, p, bytes)) locks.add(p, bytes);
            };
            want(grad, N * GD2 * sizeof(double));
            want(stress, N * SD * sizeof(double));
            // (a tangent the CPU rebuilds is neither locked nor mapped: fcamd_hosttangent.cpp)
            if (!host_tangent_applies(mg->w[0]->model, (int64_t)(N / (size_t)used), 0)) want(tangent, N * TD * sizeof(double));
            for (int k = 0; k < NH; ++k) want(hist[k], N * (size_t)mg->info.hist[k].dim * sizeof(double));
        }
    }
    std::vector<fcamd_stats> part((size_t)used);
    std::vector<int> modes((size_t)used, 0);
    st = mg->run(used, [&](int k) {
        int64_t lo, hi;
        slice_of(n, used, k, &lo, &hi);
        Worker* w = mg->w[k].get();
        double* h[FCAMD_MAX_HISTORY] = {nullptr, nullptr};
        for (int f = 0; f < NH; ++f) h[f] = hist[f] + (size_t)mg->info.hist[f].dim * (size_t)lo;
        const int s = fcamd_evaluate_host(w->model, t, del_t, hi - lo, grad + GD2 * (size_t)lo, stress + SD * (size_t)lo,
                                          tangent ? tangent + TD * (size_t)lo : nullptr, NH ? h : nullptr, NH, &part[(size_t)k]);
        modes[(size_t)k] = w->ctx->last_host_mode;
        return s;
    });
    for (int k = 0; k < used; ++k) {
        mg->last_mode |= modes[(size_t)k];
        if (stats) {
            stats->n_nonconverged += part[(size_t)k].n_nonconverged;
            stats->n_plastic += part[(size_t)k].n_plastic;
            stats->n_newton_iters += part[(size_t)k].n_newton_iters;
            stats->n_domain += part[(size_t)k].n_domain;
        }
    }
    return st;
}

static int multi_unregister_host_buffer(fcamd_multi* mg, void* ptr);

int fcamd_multi_register_host_buffer(fcamd_multi* mg, void* ptr, size_t bytes) {
    if (!mg || !ptr) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (bytes == 0) return multi_unregister_host_buffer(mg, ptr);
    std::lock_guard<std::mutex> call(mg->call_mu);
    // slot 0 takes the page lock, the other contexts enter the range with their own device's view of it
    int st = mg->run(1, [&](int) { return fcamd_register_host_buffer(mg->w[0]->ctx, ptr, bytes); });
    if (st != FCAMD_OK) return st;
    for (size_t k = 1; k < mg->w.size() && st == FCAMD_OK; ++k) {
        Worker* w = mg->w[k].get();
        w->post([w, ptr, bytes] { return adopt_registered_range(w->ctx, ptr, bytes); });
        st = w->wait();
        if (st != FCAMD_OK) st = fail(st, "%s", w->error.c_str());
    }
    if (st != FCAMD_OK) {
        const std::string msg = fcamd_last_error();
        for (size_t k = mg->w.size(); k-- > 0;) {
            Worker* w = mg->w[k].get();
            w->post([w, ptr] { return fcamd_unregister_host_buffer(w->ctx, ptr); });
            (void)w->wait();
        }
        return fail(st, "%s", msg.c_str());
    }
    return FCAMD_OK;
}

static int multi_unregister_host_buffer(fcamd_multi* mg, void* ptr) {
    std::lock_guard<std::mutex> call(mg->call_mu);
    int first = FCAMD_OK;
    for (size_t k = mg->w.size(); k-- > 0;) {  // the borrowers first, the owner of the lock (slot 0) last
        Worker* w = mg->w[k].get();
        w->post([w, ptr] { return fcamd_unregister_host_buffer(w->ctx, ptr); });
        const int st = w->wait();
        if (st != FCAMD_OK && first == FCAMD_OK) first = fail(st, "%s", w->error.c_str());
    }
    return first;
}

int fcamd_multi_get_option(const fcamd_multi* mg, const char* name, long long* value) {
    if (!mg || !name || !value) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    if (std::strcmp(name, "n_devices") == 0) *value = (long long)mg->w.size();
    else if (std::strcmp(name, "last_host_mode") == 0) *value = mg->last_mode;
    else if (std::strcmp(name, "last_n_used") == 0) *value = mg->last_used;
    else if (std::strcmp(name, "min_points") == 0) *value = mg->min_points;
    else if (!mg->w.empty()) return fcamd_context_get_option(mg->w[0]->ctx, name, value);  // a context option: what the first device's context holds
    else return fail(FCAMD_ERR_BAD_ARG, "unknown option '%s'", name);
    return FCAMD_OK;
}

int fcamd_multi_set_option(fcamd_multi* mg, const char* name, long long value) {
    if (!mg || !name) return fail(FCAMD_ERR_BAD_ARG, "NULL argument");
    std::lock_guard<std::mutex> call(mg->call_mu);
    if (std::strcmp(name, "min_points") == 0) {  // the handle's own option
        mg->min_points = std::max<long long>(1, value);
        return FCAMD_OK;
    }
    for (auto& w : mg->w) {
        const int st = fcamd_context_set_option(w->ctx, name, value);
        if (st != FCAMD_OK) return st;
    }
    return FCAMD_OK;
}

// ---- device-resident increment state over several GPUs -------------------------------------------------------

int fcamd_multi_state_create(fcamd_multi* mg, int64_t n, int flags, fcamd_multi_state** out) {
    if (!out) return fail(FCAMD_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!mg) return fail(FCAMD_ERR_BAD_ARG, "multi handle is NULL");
    if (n < 0) return fail(FCAMD_ERR_SIZE, "negative number of quadrature points");
    if (flags & ~FCAMD_EVAL_SPLIT_HISTORY) return fail(FCAMD_ERR_BAD_ARG, "flags: FCAMD_EVAL_SPLIT_HISTORY or 0");
    if ((flags & FCAMD_EVAL_SPLIT_HISTORY) && (!has_split_history(mg->law) || mg->dims.gdim != 3))
        return fail(FCAMD_ERR_UNSUPPORTED, "FCAMD_EVAL_SPLIT_HISTORY exists for the 3-D laws with one [scalar, eps_p(6)] history row per point");
    std::lock_guard<std::mutex> call(mg->call_mu);
    std::unique_ptr<fcamd_multi_state> st(new (std::nothrow) fcamd_multi_state());
    if (!st) return fail(FCAMD_ERR_BAD_ARG, "out of host memory");
    st->mg = mg;
    st->n = n;
    st->flags = flags;
    const bool split = (flags & FCAMD_EVAL_SPLIT_HISTORY) != 0;
    st->nh = split ? 2 : mg->info.n_hist;
    for (int f = 0; f < st->nh; ++f) st->hdim[f] = split ? (f == 0 ? 1 : 6) : (size_t)mg->info.hist[f].dim;
    const int world = (int)mg->w.size();
    st->s.resize((size_t)world);
    fcamd_multi_state* sp = st.get();
    const size_t SD = (size_t)mg->dims.sd;
    const bool masks = has_sparse_history(mg->law);
    const int rc = mg->run(world, [&, sp](int k) {
        auto& sl = sp->s[(size_t)k];
        slice_of(n, world, k, &sl.lo, &sl.hi);
        const size_t nk = (size_t)(sl.hi - sl.lo);
        if (nk == 0) return (int)FCAMD_OK;
        HIP_TRY(hipSetDevice(mg->w[k]->device));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sl.stress[i]), nk * SD * sizeof(double)));
            HIP_TRY(hipMemset(sl.stress[i], 0, nk * SD * sizeof(double)));
            for (int f = 0; f < sp->nh; ++f) {
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sl.hist[i][f]), nk * sp->hdim[f] * sizeof(double)));
                HIP_TRY(hipMemset(sl.hist[i][f], 0, nk * sp->hdim[f] * sizeof(double)));
            }
        }
        if (masks) {
            const size_t words = (nk + 63) / 64;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sl.mask), words * sizeof(uint64_t)));
            HIP_TRY(hipMemset(sl.mask, 0, words * sizeof(uint64_t)));
        }
        // the fills run on the null stream; everything later runs on the contexts' non-blocking streams, which do
        // not wait for it: a fill still in flight would land on top of fcamd_multi_state_set's upload
        HIP_TRY(hipDeviceSynchronize());
        return (int)FCAMD_OK;
    });
    if (rc != FCAMD_OK) {
        const std::string msg = fcamd_last_error();
        // release what was allocated (call_mu is held: fcamd_multi_state_destroy would take it again)
        (void)mg->run(world, [&, sp](int k) {
            auto& sl = sp->s[(size_t)k];
            (void)hipSetDevice(mg->w[k]->device);
            for (int i = 0; i < 2; ++i) {
                if (sl.stress[i]) (void)hipFree(sl.stress[i]);
                for (int f = 0; f < FCAMD_MAX_HISTORY; ++f)
                    if (sl.hist[i][f]) (void)hipFree(sl.hist[i][f]);
            }
            if (sl.mask) (void)hipFree(sl.mask);
            return (int)FCAMD_OK;
        });
        return fail(rc, "%s", msg.c_str());
    }
    mg->states.push_back(st.get());
    *out = st.release();
    return FCAMD_OK;
}

int fcamd_multi_state_destroy(fcamd_multi_state* st) {
    if (!st) return FCAMD_OK;
    fcamd_multi* mg = st->mg;
    {
        std::lock_guard<std::mutex> call(mg->call_mu);
        mg->states.erase(std::remove(mg->states.begin(), mg->states.end(), st), mg->states.end());
        (void)mg->run((int)mg->w.size(), [&](int k) {
            auto& sl = st->s[(size_t)k];
            (void)hipSetDevice(mg->w[k]->device);
            for (int i = 0; i < 2; ++i) {
                if (sl.stress[i]) (void)hipFree(sl.stress[i]);
                for (int f = 0; f < FCAMD_MAX_HISTORY; ++f)
                    if (sl.hist[i][f]) (void)hipFree(sl.hist[i][f]);
            }
            if (sl.mask) (void)hipFree(sl.mask);
            (void)hipGetLastError();
            return (int)FCAMD_OK;
        });
    }
    delete st;
    return FCAMD_OK;
}

}  // extern "C"

namespace {

// history of one slice between the interface layout (the law's fields; 7-double rows for the split laws) and the
// state's device arrays; `to_device`: host -> committed copy `c` (and the trial copy), else device copy `c` -> host
int move_history(fcamd_multi_state* st, int k, int c, double* const* host, bool to_device) {
    fcamd_multi* mg = st->mg;
    auto& sl = st->s[(size_t)k];
    fcamd_context* ctx = mg->w[k]->ctx;
    const size_t nk = (size_t)(sl.hi - sl.lo), lo = (size_t)sl.lo;
    const bool split = (st->flags & FCAMD_EVAL_SPLIT_HISTORY) != 0;
    if (!split) {
        for (int f = 0; f < st->nh; ++f) {
            double* h = host[f] + st->hdim[f] * lo;
            const size_t bytes = nk * st->hdim[f] * sizeof(double);
            const int rc = to_device ? fcamd_copy_to_device(ctx, sl.hist[c][f], h, bytes) : fcamd_copy_to_host(ctx, h, sl.hist[c][f], bytes);
            if (rc != FCAMD_OK) return rc;
        }
        return FCAMD_OK;
    }
    // split: [scalar | eps_p(6)] rows of 7 <-> scalars (nk) + rows (6 nk), through host temporaries
    std::vector<double> sc(nk), rows(6 * nk);
    double* h7 = host[0] + 7 * lo;
    if (to_device) {
        for (size_t i = 0; i < nk; ++i) {
            sc[i] = h7[7 * i];
            std::memcpy(&rows[6 * i], h7 + 7 * i + 1, 6 * sizeof(double));
        }
        int rc = fcamd_copy_to_device(ctx, sl.hist[c][0], sc.data(), nk * sizeof(double));
        if (rc == FCAMD_OK) rc = fcamd_copy_to_device(ctx, sl.hist[c][1], rows.data(), 6 * nk * sizeof(double));
        return rc;
    }
    int rc = fcamd_copy_to_host(ctx, sc.data(), sl.hist[c][0], nk * sizeof(double));
    if (rc == FCAMD_OK) rc = fcamd_copy_to_host(ctx, rows.data(), sl.hist[c][1], 6 * nk * sizeof(double));
    if (rc != FCAMD_OK) return rc;
    for (size_t i = 0; i < nk; ++i) {
        h7[7 * i] = sc[i];
        std::memcpy(h7 + 7 * i + 1, &rows[6 * i], 6 * sizeof(double));
    }
    return FCAMD_OK;
}

// the coordinator's call-scoped page locks on the caller's whole state arrays (several devices, arrays beyond the scratch path)
void lock_state_arrays(fcamd_multi_state* st, LockSet& locks, const double* stress_host, const double* const* history_host) {
    fcamd_multi* mg = st->mg;
    if (mg->w.size() < 2) return;
    const fcamd_context* c0 = mg->w[0]->ctx;
    const size_t N = (size_t)st->n;
    auto want = [&](const void* p, size_t bytes) {
        if (p && bytes > (size_t)c0->opt.bounce_max && !inside_registered(c0, p, bytes)) locks.add(p, bytes);
    };
    want(stress_host, N * (size_t)mg->dims.sd * sizeof(double));
    if (history_host)
        for (int f = 0; f < mg->info.n_hist; ++f) want(history_host[f], N * (size_t)mg->info.hist[f].dim * sizeof(double));
}

}  // namespace

extern "C" {

int fcamd_multi_state_set(fcamd_multi_state* st, const double* stress_host, const double* const* history_host, int n_hist) {
    if (!st) return fail(FCAMD_ERR_BAD_ARG, "state is NULL");
    fcamd_multi* mg = st->mg;
    if (history_host && n_hist != mg->info.n_hist)
        return fail(FCAMD_ERR_SIZE, "law expects %d history fields, got %d", mg->info.n_hist, n_hist);
    if (history_host)
        for (int f = 0; f < n_hist; ++f)
            if (st->n > 0 && !history_host[f]) return fail(FCAMD_ERR_NULL_HISTORY, "history must not be None");
    std::lock_guard<std::mutex> call(mg->call_mu);
    const size_t SD = (size_t)mg->dims.sd;
    const int c = st->committed;
    // As in _evaluate: the coordinator page-locks every WHOLE caller array once; the workers then find their slices
    // locked.  (Slices are tile-aligned, not page-aligned: per-worker locks would share boundary pages.)
    LockSet locks;
    lock_state_arrays(st, locks, stress_host, history_host);
    const int rc = mg->run((int)mg->w.size(), [&](int k) {
        auto& sl = st->s[(size_t)k];
        const size_t nk = (size_t)(sl.hi - sl.lo);
        if (nk == 0) return (int)FCAMD_OK;
        fcamd_context* ctx = mg->w[k]->ctx;
        HIP_TRY(hipSetDevice(mg->w[k]->device));
        int s = FCAMD_OK;
        if (stress_host)
            s = fcamd_copy_to_device(ctx, sl.stress[c], stress_host + SD * (size_t)sl.lo, nk * SD * sizeof(double));
        else
            s = hip_status(hipMemset(sl.stress[c], 0, nk * SD * sizeof(double)), "hipMemset");
        if (s != FCAMD_OK) return s;
        if (history_host) {
            s = move_history(st, k, c, const_cast<double* const*>(history_host), true);
            if (s != FCAMD_OK) return s;
        } else {
            for (int f = 0; f < st->nh; ++f) HIP_TRY(hipMemset(sl.hist[c][f], 0, nk * st->hdim[f] * sizeof(double)));
        }
        // sparse trial-history contract: trial == committed wherever the mask is clear
        for (int f = 0; f < st->nh; ++f)
            HIP_TRY(hipMemcpy(sl.hist[1 - c][f], sl.hist[c][f], nk * st->hdim[f] * sizeof(double), hipMemcpyDeviceToDevice));
        if (sl.mask) HIP_TRY(hipMemset(sl.mask, 0, ((nk + 63) / 64) * sizeof(uint64_t)));
        HIP_TRY(hipDeviceSynchronize());
        return (int)FCAMD_OK;
    });
    st->evaluated = false;
    st->failed = FCAMD_OK;
    return rc;
}

int fcamd_multi_state_get(fcamd_multi_state* st, int trial, double* stress_host, double* const* history_host, int n_hist) {
    if (!st) return fail(FCAMD_ERR_BAD_ARG, "state is NULL");
    fcamd_multi* mg = st->mg;
    if (history_host && n_hist != mg->info.n_hist)
        return fail(FCAMD_ERR_SIZE, "law expects %d history fields, got %d", mg->info.n_hist, n_hist);
    if (history_host)
        for (int f = 0; f < n_hist; ++f)
            if (st->n > 0 && !history_host[f]) return fail(FCAMD_ERR_NULL_HISTORY, "history must not be None");
    std::lock_guard<std::mutex> call(mg->call_mu);
    const size_t SD = (size_t)mg->dims.sd;
    // before the first evaluate of an increment the trial state IS the committed one
    const int c = (trial && st->evaluated) ? 1 - st->committed : st->committed;
    LockSet locks;
    lock_state_arrays(st, locks, stress_host, history_host);
    return mg->run((int)mg->w.size(), [&](int k) {
        auto& sl = st->s[(size_t)k];
        const size_t nk = (size_t)(sl.hi - sl.lo);
        if (nk == 0) return (int)FCAMD_OK;
        fcamd_context* ctx = mg->w[k]->ctx;
        if (stress_host) {
            const int s = fcamd_copy_to_host(ctx, stress_host + SD * (size_t)sl.lo, sl.stress[c], nk * SD * sizeof(double));
            if (s != FCAMD_OK) return s;
        }
        if (history_host) return move_history(st, k, c, history_host, false);
        return (int)FCAMD_OK;
    });
}

int fcamd_multi_state_evaluate(fcamd_multi_state* st, double t, double del_t, const double* grad, double* stress_host,
                               double* tangent_host, int flags, fcamd_stats* stats) {
    if (!st) return fail(FCAMD_ERR_BAD_ARG, "state is NULL");
    fcamd_multi* mg = st->mg;
    if (flags & ~FCAMD_EVAL_SPARSE_TANGENT) return fail(FCAMD_ERR_BAD_ARG, "flags: FCAMD_EVAL_SPARSE_TANGENT or 0");
    if (st->n > 0 && !grad) return fail(FCAMD_ERR_BAD_ARG, "grad_del_u pointer is NULL");
    if (mg->info.needs_del_t && !(del_t > 0.0)) return fail(FCAMD_ERR_DEL_T, "Time step must be defined and positive.");
    std::lock_guard<std::mutex> call(mg->call_mu);
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const size_t GD2 = (size_t)mg->dims.gd2, SD = (size_t)mg->dims.sd, TD = SD * SD;
    const int world = (int)mg->w.size();
    const size_t N = (size_t)st->n;
    LockSet locks;
    if (world > 1) {
        const fcamd_context* c0 = mg->w[0]->ctx;
        const size_t total = N * (GD2 + (stress_host ? SD : 0) + (tangent_host ? TD : 0)) * sizeof(double);
        if (total > (size_t)c0->opt.bounce_max) {
            auto want = [&](const void* p, size_t bytes) {
                if (p && bytes && !inside_registered(c0, p, bytes)) locks.add(p, bytes);
            };
            want(grad, N * GD2 * sizeof(double));
            want(stress_host, N * SD * sizeof(double));
            if (!host_tangent_applies(mg->w[0]->model, (int64_t)(N / (size_t)world), flags)) want(tangent_host, N * TD * sizeof(double));
        }
    }
    const int c = st->committed;
    const int eflags = (flags & FCAMD_EVAL_SPARSE_TANGENT) | st->flags;
    std::vector<fcamd_stats> part((size_t)world);
    std::vector<int> modes((size_t)world, 0);
    st->evaluated = true;  // the trial state is touched even if the call fails
    mg->last_used = world;
    mg->last_mode = 0;
    const int rc = mg->run(world, [&](int k) {
        auto& sl = st->s[(size_t)k];
        std::memset(&part[(size_t)k], 0, sizeof(fcamd_stats));
        const int64_t nk = sl.hi - sl.lo;
        if (nk == 0) return (int)FCAMD_OK;
        Worker* w = mg->w[k].get();
        const double* hp[FCAMD_MAX_HISTORY] = {sl.hist[c][0], sl.hist[c][1]};
        double* hc[FCAMD_MAX_HISTORY] = {sl.hist[1 - c][0], sl.hist[1 - c][1]};
        const size_t lo = (size_t)sl.lo;
        fcamd_eval_args x{};
        x.grad_del_u = grad + GD2 * lo;  // host array
        x.stress_prev = sl.stress[c];
        x.stress = sl.stress[1 - c];
        x.history_prev = st->nh ? hp : nullptr;
        x.history = st->nh ? hc : nullptr;
        x.n_hist = st->nh;
        x.history_mask = sl.mask;
        x.flags = eflags;
        const int s = fcamd_evaluate_resident(w->model, t, del_t, nk, &x, stress_host ? stress_host + SD * lo : nullptr,
                                              tangent_host ? tangent_host + TD * lo : nullptr, &part[(size_t)k]);
        modes[(size_t)k] = w->ctx->last_host_mode;
        return s;
    });
    for (int k = 0; k < world; ++k) {
        mg->last_mode |= modes[(size_t)k];
        if (stats) {
            stats->n_nonconverged += part[(size_t)k].n_nonconverged;
            stats->n_plastic += part[(size_t)k].n_plastic;
            stats->n_newton_iters += part[(size_t)k].n_newton_iters;
            stats->n_domain += part[(size_t)k].n_domain;
        }
    }
    st->failed = rc;
    return rc;
}

int fcamd_multi_state_commit(fcamd_multi_state* st) {
    if (!st) return fail(FCAMD_ERR_BAD_ARG, "state is NULL");
    std::lock_guard<std::mutex> call(st->mg->call_mu);
    if (!st->evaluated) return fail(FCAMD_ERR_BAD_ARG, "commit before any evaluate of this increment");
    if (st->failed != FCAMD_OK)
        return fail(st->failed, "the last evaluate failed (%s): nothing to commit", fcamd_status_string(st->failed));
    st->committed = 1 - st->committed;  // the masks keep marking where the new trial arrays are stale
    st->evaluated = false;
    return FCAMD_OK;
}

}  // extern "C"
