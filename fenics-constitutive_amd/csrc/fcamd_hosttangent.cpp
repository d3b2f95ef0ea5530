// The tangent of the host (ndarray) entries, rebuilt on the CPU instead of shipped over PCIe.
//
// 288 of the 336-392 bytes per point that law.evaluate(ndarrays) brings down the link are the tangent, and the reference itself
// builds it on the host: np.tile(D.flatten(), n) for the laws with a constant tangent (linear_elasticity_model.py:45,
// spring_maxwell_model.py:84-86, spring_kelvin_model.py:85-86) and, for VonMises3D, ka xioi + B xpp + C N (x) N from two scalars and
// the flow direction (mises_plasticity_isotropic_hardening.py:170-175).  So the host entries (fcamd_hostpath.cpp) launch the kernel
//   * with no tangent at all for the constant-tangent laws, while a small pool of threads fills the caller's array from the law's 6 x 6
//     host table (the very table the kernel streams from LDS: fill_constants, fcamd_capi.cpp);
//   * with kFlagTangentParams for the two Mises laws: the kernel stores the 8 doubles per PLASTIC point it publishes to LDS anyway (B, C,
//     N[6]; kernels/tangent_writers.h) and the plastic ballot of every tile into a ring of page-locked chunks, and the pool expands
//     chunk k while the GPU works on chunk k + 1, with the expression of tangent_mises_chunk -- same operands, same order, this file is
//     built with -ffp-contract=off like the device code -- so that the array holds bit for bit what the kernel would have written;
//     elastic points get the row the same expression yields for an elastic point's parameters (B = 2 mu, C = 0, N = 0).
// The caller's tangent array is written by the CPU alone: it is neither page-locked nor mapped, and needs no alignment.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

#include <sched.h>
#include <cstdio>
#include <cstdlib>

#include "fcamd_host.h"

namespace fcamd {

namespace {

typedef double v2 __attribute__((ext_vector_type(2)));

template <bool NT>
inline void put2(double* dst, v2 v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<v2*>(dst));  // movntpd: the rows are written once and not read here
    else
        std::memcpy(dst, &v, sizeof(v));
}

// rows [p0, p1) of the tangent of a Mises law from 8 doubles per point; COMFE: the third term as mises_plasticity.rs forms it
// (kernels/tangent_writers.h: tangent_mises_chunk -- the expressions below are that function's, entry pair by entry pair)
template <bool COMFE, bool NT>
inline void expand_row(const double* t, const double* ta, const double* tb, double* row) {
    {
        const double B = t[0], C = t[1];
        for (int i = 0; i < 6; ++i) {
            const double ni = t[2 + i];
            for (int jj = 0; jj < 3; ++jj) {
                const int e = 6 * i + 2 * jj;
                const double njx = t[2 + 2 * jj], njy = t[3 + 2 * jj];
                v2 v;
                if constexpr (COMFE) {
                    v.x = (ta[e] + B * tb[e]) + (C * njx) * ni;
                    v.y = (ta[e + 1] + B * tb[e + 1]) + (C * njy) * ni;
                } else {
                    v.x = (ta[e] + B * tb[e]) + C * (ni * njx);
                    v.y = (ta[e + 1] + B * tb[e + 1]) + C * (ni * njy);
                }
                put2<NT>(row + e, v);
            }
        }
    }
}

// rows [p0, p1) (p0 a multiple of 64) of the tangent: the plastic points of `mask` (one word per 64 points, first word = the tile of
// p0) from their 8 doubles at prm + 8 (p - p0), the others the law's elastic row
template <bool COMFE, bool NT>
void expand_mises(const double* prm, const unsigned long long* mask, const double* ta, const double* tb, const double* elastic,
                  double* tangent, int64_t p0, int64_t p1) {
    for (int64_t t0 = p0; t0 < p1; t0 += 64) {
        const unsigned long long word = mask[(t0 - p0) >> 6];
        const int cnt = (int)std::min<int64_t>(64, p1 - t0);
        for (int l = 0; l < cnt; ++l) {
            double* row = tangent + 36 * (t0 + l);
            if ((word >> l) & 1ull) {
                expand_row<COMFE, NT>(prm + 8 * (t0 + l - p0), ta, tb, row);
            } else {
                for (int e = 0; e < 36; e += 2) {
                    v2 v;
                    v.x = elastic[e];
                    v.y = elastic[e + 1];
                    put2<NT>(row + e, v);
                }
            }
        }
    }
}

// Drucker-Prager: rows [p0, p1) from 12 doubles per plastic point (kernels/law_drucker_prager.h: tangent_dp_chunk -- the expressions below
// are that function's); elastic points get the law's E table itself, as the kernel gives them (general.rs:131-135)
template <bool NT>
void expand_dp(const double* prm, const unsigned long long* mask, const double* t11tab, const double* pdtab, const double* etab,
               double* tangent, int64_t p0, int64_t p1) {
    for (int64_t t0 = p0; t0 < p1; t0 += 64) {
        const unsigned long long word = mask[(t0 - p0) >> 6];
        const int cnt = (int)std::min<int64_t>(64, p1 - t0);
        for (int l = 0; l < cnt; ++l) {
            double* row = tangent + 36 * (t0 + l);
            const double* t = prm + 12 * (t0 + l - p0);
            const bool plastic = ((word >> l) & 1ull) != 0ull && t[5] != 0.0;  // (the record's own flag: 1.0 for every point the kernel sends)
            for (int i = 0; i < 6; ++i) {
                const double oi = i < 3 ? 1.0 : 0.0;
                for (int jj = 0; jj < 3; ++jj) {
                    const int e = 6 * i + 2 * jj;
                    v2 v;
                    if (plastic) {
                        const double c0x = t[0], c0y = t[1], c1x = t[2], c1y = t[3], ts1 = t[4];
                        const double si = t[6 + i], sjx = t[6 + 2 * jj], sjy = t[7 + 2 * jj];
                        v.x = (c0x * t11tab[e] + c0y * pdtab[e]) + ((c1x * si) * sjx + (c1y * oi) * sjx + (ts1 * si) * (jj < 2 ? 1.0 : 0.0));
                        v.y = (c0x * t11tab[e + 1] + c0y * pdtab[e + 1]) + ((c1x * si) * sjy + (c1y * oi) * sjy + (ts1 * si) * (jj < 1 ? 1.0 : 0.0));
                    } else {
                        v.x = etab[e];
                        v.y = etab[e + 1];
                    }
                    put2<NT>(row + e, v);
                }
            }
        }
    }
}

// tangent[td * p + k] = table[k] for p0 <= p < p1 (np.tile(D.flatten(), n)); td = 36, 16 or 1
template <bool NT>
void fill_const(const double* table, int td, double* tangent, int64_t p0, int64_t p1) {
    // one period of the pattern that is a whole number of 16-byte pairs: td doubles (td even) or two points (td odd)
    double pat[72];
    const int period = (td % 2 == 0) ? td : 2 * td;
    for (int k = 0; k < period; ++k) pat[k] = table[k % td];
    double* dst = tangent + (int64_t)td * p0;
    const int64_t total = (int64_t)td * (p1 - p0);
    int64_t i = 0;
    for (; i + period <= total; i += period)
        for (int k = 0; k < period; k += 2) {
            v2 v;
            v.x = pat[k];
            v.y = pat[k + 1];
            put2<NT>(dst + i + k, v);
        }
    for (int k = 0; i < total; ++i, ++k) dst[i] = pat[k];
}

}  // namespace

class ExpandPool {
  public:
    // (keeping the threads on the NUMA node of the calling thread was measured and changes nothing: 300 / 297 Mpts/s for VonMises3D,
    // 421 / 418 for linear elasticity at 1e7 points on the two-socket host of an MI355X box)
    explicit ExpandPool(int threads) {
        for (int t = 0; t < threads; ++t) workers_.emplace_back([this] { run(); });
    }
    ~ExpandPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int threads() const { return (int)workers_.size(); }

    void begin(const HostTangentJob& job) {
        std::lock_guard<std::mutex> g(mu_);
        job_ = job;
        left_.clear();
        busy_ns_ = 0;
        nt_ = (reinterpret_cast<uintptr_t>(job.tangent) & 15u) == 0;
    }
    // points [p0, p0 + np) are ready (p0 a multiple of 64): `src` = their 8 doubles per point, `mask` = their tiles' plastic ballots
    // (Mises kinds; unused for the constant fill).  Returns the ticket of the chunk for wait().
    int post(int64_t p0, int64_t np, const double* src, const unsigned long long* mask) {
        // up to four tasks per thread and chunk (a thread that another tenant of the host slows down delays a sixty-fourth of a chunk,
        // not a sixteenth), none smaller than kMinPart points: a small call wakes few threads -- waking a sleeping thread costs 50-300 us
        // on this host, sixteen of them for 64 Ki points cost more than the rows they write (LE, 65 536 points: 0.65 ms with 16 threads
        // woken, 0.48 ms with 4, 0.52 ms with the kernel's own tangent stores)
        // (the chunks of the parameter pipeline are small by design and follow each other within a millisecond: every thread takes part)
        const int64_t min_part = job_.kind == HostTangentJob::CONST ? kMinPart : kMinPartPipeline;
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(4 * (int64_t)threads(), np / min_part));
        std::lock_guard<std::mutex> g(mu_);
        const int ticket = (int)left_.size();
        left_.push_back(parts);
        for (int k = 0; k < parts; ++k) {
            // cut on whole tiles: one mask word per 64 points (and 64 rows are 288 whole 64-byte lines)
            const int64_t a = (np * k / parts) & ~(int64_t)63, b = (k + 1 == parts) ? np : ((np * (k + 1) / parts) & ~(int64_t)63);
            tasks_.push_back({ticket, p0 + a, p0 + b, src ? src + (size_t)job_.prm * a : nullptr, mask ? mask + (a >> 6) : nullptr});
        }
        if (parts >= threads()) cv_work_.notify_all();
        else
            for (int k = 0; k < parts; ++k) cv_work_.notify_one();
        return ticket;
    }
    void wait(int ticket) {
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [&] { return left_[ticket] == 0; });
    }
    void finish() {
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [&] {
            for (int v : left_)
                if (v) return false;
            return true;
        });
    }
    double busy_seconds() {
        std::lock_guard<std::mutex> g(mu_);
        return (double)busy_ns_ * 1e-9;
    }

  private:
    static constexpr int64_t kMinPart = 16384;         // points per task at least (4.7 MB of rows): the one-shot fill of a constant tangent
    static constexpr int64_t kMinPartPipeline = 2048;  // ... of a chunk of the parameter pipeline (VonMises3D, 131 072 points: 200 Mpts/s; with 16 384: 110)
    struct Task {
        int ticket;
        int64_t p0, p1;
        const double* src;
        const unsigned long long* mask;
    };
    void work(const Task& t, const HostTangentJob& j, bool nt) {
        switch (j.kind) {
            case HostTangentJob::CONST:
                nt ? fill_const<true>(j.table_c, j.td, j.tangent, t.p0, t.p1) : fill_const<false>(j.table_c, j.td, j.tangent, t.p0, t.p1);
                break;
            case HostTangentJob::MISES:
                nt ? expand_mises<false, true>(t.src, t.mask, j.table_a, j.table_b, j.elastic_row, j.tangent, t.p0, t.p1)
                   : expand_mises<false, false>(t.src, t.mask, j.table_a, j.table_b, j.elastic_row, j.tangent, t.p0, t.p1);
                break;
            case HostTangentJob::MISES_COMFE:
                nt ? expand_mises<true, true>(t.src, t.mask, j.table_a, j.table_b, j.elastic_row, j.tangent, t.p0, t.p1)
                   : expand_mises<true, false>(t.src, t.mask, j.table_a, j.table_b, j.elastic_row, j.tangent, t.p0, t.p1);
                break;
            case HostTangentJob::DRUCKER_PRAGER:
                nt ? expand_dp<true>(t.src, t.mask, j.table_a, j.table_b, j.table_c, j.tangent, t.p0, t.p1)
                   : expand_dp<false>(t.src, t.mask, j.table_a, j.table_b, j.table_c, j.tangent, t.p0, t.p1);
                break;
        }
        __builtin_ia32_sfence();  // the non-temporal stores are globally visible before the task counts as done
    }
    void run() {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_work_.wait(g, [&] { return stop_ || !tasks_.empty(); });
            if (stop_) return;
            const Task t = tasks_.front();
            tasks_.pop_front();
            const HostTangentJob j = job_;
            const bool nt = nt_;
            g.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            work(t, j, nt);
            const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            g.lock();
            busy_ns_ += ns;
            if (--left_[t.ticket] == 0) cv_done_.notify_all();
        }
    }

    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Task> tasks_;
    std::vector<int> left_;  // per posted chunk: tasks not finished yet
    std::vector<std::thread> workers_;
    HostTangentJob job_{};
    bool nt_ = false, stop_ = false;
    long long busy_ns_ = 0;
};

// which laws the host can rebuild the tangent of (every law of the library; 0: none)
int host_tangent_kind(const fcamd_model* m) {
    switch (m->law) {
        case FCAMD_LINEAR_ELASTICITY:
        case FCAMD_SPRING_MAXWELL:
        case FCAMD_SPRING_KELVIN:
        case FCAMD_COMFE_LINEAR_ELASTICITY: return 1 + HostTangentJob::CONST;
        case FCAMD_VON_MISES_3D: return 1 + HostTangentJob::MISES;
        case FCAMD_COMFE_MISES_PLASTICITY: return 1 + HostTangentJob::MISES_COMFE;
        case FCAMD_COMFE_DRUCKER_PRAGER:
        case FCAMD_COMFE_DRUCKER_PRAGER_HYPERBOLIC: return 1 + HostTangentJob::DRUCKER_PRAGER;
        default: return 0;
    }
}

// CPUs' worth of run time the process's cgroup grants (cgroup v2 cpu.max = "<quota> <period>" or "max <period>"); 0: no limit known
static int cgroup_cpu_quota() {
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (!f) return 0;
    char q[32] = {0};
    long long period = 0;
    const int k = fscanf(f, "%31s %lld", q, &period);
    fclose(f);
    if (k != 2 || period <= 0 || q[0] == 'm') return 0;
    const long long quota = atoll(q);
    return quota > 0 ? (int)((quota + period - 1) / period) : 0;
}

// Threads of the expansion for a context: option "host_tangent_threads" (FCAMD_HOST_TANGENT_THREADS); -1 = automatic: the CPUs this
// process may keep busy -- its affinity mask less the calling thread, within the cgroup's CPU quota (more threads than that are throttled:
// 32 threads under a quota of 16 CPUs took 1.3 - 12 x the CPU time of 16) --, at most 16 (a GPU's share of the cores of an 8-GPU host;
// VonMises3D at 1e7 points: 4 / 8 / 16 threads 244 / 288 / 300 Mpts/s).
static int usable_cpus() {
    static const int usable = [] {
        cpu_set_t set;
        int cpus = (int)std::thread::hardware_concurrency();
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        int n = cpus - 1;
        const int quota = cgroup_cpu_quota();
        if (quota > 0) n = std::min(n, quota);
        return std::max(1, n);
    }();
    return usable;
}

// Fewer threads than this cannot keep up with the link: one thread expands 32-41 Mpts/s of VonMises3D rows (70-140 of constant rows),
// the kernel's own tangent stores over PCIe give the whole call 135 Mpts/s -- with four threads the two are even.  A process that may
// use fewer CPUs (an MPI rank bound to one or two cores) keeps the kernel's stores unless the option asks for threads explicitly.
constexpr int kMinAutoThreads = 6;

int host_tangent_threads(const fcamd_context* c) {
    const int opt = c->opt.host_tangent_threads;
    if (opt >= 0) return std::min(opt, 256);
    const int n = std::min(16, usable_cpus());
    return n >= kMinAutoThreads ? n : 0;
}

// one process driving n devices (fcamd_multi): every device's context expands its own slice at the same time
int host_tangent_threads_shared(int n_contexts) {
    const int n = std::min(16, usable_cpus() / std::max(1, n_contexts));
    return n >= kMinAutoThreads ? n : 0;
}

bool host_tangent_applies(const fcamd_model* m, int64_t n, int flags) {
    const fcamd_context* c = m->ctx;
    const int kind = host_tangent_kind(m);
    // the one-shot fill of a constant tangent pays off from half the size at which the parameter pipeline does (measured, 16 threads:
    // LE 2.1 x at 32 768 points; VonMises3D even at 32 768, 1.4 x at 65 536)
    const long long min_points = kind == 1 + HostTangentJob::CONST ? c->opt.host_tangent_min_points / 2 : c->opt.host_tangent_min_points;
    if (!c->opt.zero_copy || n < min_points || n < 64 || kind == 0) return false;
    if ((flags & FCAMD_EVAL_SPARSE_TANGENT) || (flags != 0 && m->dims.gdim != 3)) return false;
    return host_tangent_threads(c) > 0;
}

ExpandPool* host_tangent_pool(fcamd_context* c) {
    const int want = host_tangent_threads(c);
    if (want <= 0) return nullptr;
    if (c->pool && c->pool->threads() != want) {
        delete c->pool;
        c->pool = nullptr;
    }
    if (!c->pool) c->pool = new (std::nothrow) ExpandPool(want);
    return c->pool;
}

void host_tangent_release(fcamd_context* c) {
    delete c->pool;
    c->pool = nullptr;
    for (int i = 0; i < fcamd_context::kTangentSlots; ++i) {
        if (c->tp_event[i]) (void)hipEventDestroy(c->tp_event[i]);
        c->tp_event[i] = nullptr;
    }
    if (c->tparams) (void)hipHostFree(c->tparams);
    c->tparams = c->tparams_dev = nullptr;
    c->tparams_bytes = 0;
}

void pool_begin(ExpandPool* p, const HostTangentJob& job) { p->begin(job); }
int pool_post(ExpandPool* p, int64_t p0, int64_t np, const double* src, const unsigned long long* mask) { return p->post(p0, np, src, mask); }
void pool_wait(ExpandPool* p, int ticket) { p->wait(ticket); }
void pool_finish(ExpandPool* p) { p->finish(); }
double pool_busy_seconds(ExpandPool* p) { return p->busy_seconds(); }
int pool_threads(ExpandPool* p) { return p->threads(); }

HostTangentJob host_tangent_job(const fcamd_model* m, double* tangent) {
    HostTangentJob j{};
    j.kind = (HostTangentJob::Kind)(host_tangent_kind(m) - 1);
    j.td = m->dims.sd * m->dims.sd;
    j.prm = j.kind == HostTangentJob::DRUCKER_PRAGER ? 12 : 8;
    j.table_a = m->tb.a;
    j.table_b = m->tb.b;
    j.table_c = m->tb.c;
    j.tangent = tangent;
    if (j.kind == HostTangentJob::MISES || j.kind == HostTangentJob::MISES_COMFE) {
        // what an elastic point publishes (law_von_mises.h: VMReturn's zeros through vm_tangent_coefficients; law_comfe_mises.h: cm_point)
        // expanded by the very function that expands the plastic points' parameters
        double prm[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (j.kind == HostTangentJob::MISES) {
            const double two_mu = m->sc.s[2], four_mu2 = m->sc.s[9], xc1 = 0.0, xc2 = 0.0;
            prm[0] = two_mu * (1.0 - two_mu * xc2);
            prm[1] = four_mu2 * (xc2 - xc1);
            expand_row<false, false>(prm, j.table_a, j.table_b, j.elastic_row);
        } else {
            prm[0] = m->sc.s[5];
            expand_row<true, false>(prm, j.table_a, j.table_b, j.elastic_row);
        }
    }
    return j;
}

// the ring of page-locked parameter chunks: `slots` x (`chunk` points x `prm` doubles + chunk / 64 ballot words), and one event per slot
int host_tangent_ring(fcamd_context* c, int64_t chunk, int slots, int prm) {
    const size_t bytes = (size_t)slots * host_tangent_slot_doubles(chunk, prm) * sizeof(double);
    if (bytes > c->tparams_bytes) {
        if (c->tparams) HIP_TRY(hipHostFree(c->tparams));
        c->tparams = c->tparams_dev = nullptr;
        c->tparams_bytes = 0;
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            return fail(FCAMD_ERR_HIP, "the page-locked tangent-parameter ring is not mapped into the device's address space");
        }
        c->tparams = static_cast<char*>(h);
        c->tparams_dev = static_cast<char*>(d);
        c->tparams_bytes = bytes;
    }
    for (int i = 0; i < slots; ++i)
        // (blocking sync: the calling thread sleeps in hipEventSynchronize instead of spinning next to the expansion threads; events that
        // spin were measured: no difference at 1.3e5, 1e6, 1e7 points)
        if (!c->tp_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->tp_event[i], hipEventDisableTiming | hipEventBlockingSync));
    return FCAMD_OK;
}

}  // namespace fcamd
