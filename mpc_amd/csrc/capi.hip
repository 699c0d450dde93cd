// C ABI (include/mpc_amd.h): handle life-cycle, table upload, host-pointer wrappers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "setup.h"
#include "tables.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(MPCB_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

template <class T> int upload(const std::vector<T> &h, T **d) {
    *d = nullptr;
    if (h.empty()) return 0;
    HIP_TRY(hipMalloc((void **)d, h.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

} // namespace

// Scratch of mpcb_solve_coupled_device_noalloc: the hand-over counter and list of the first
// stage and the dense second stage's J / R slots, in device memory that exists from the moment
// the library is loaded on a GPU until the process ends -- the entry point therefore never
// allocates.  Every GPU holds its own copy at its own address; g_scratch[device] is that
// address (hipGetSymbolAddress, taken once per device in mpcb_create with the device current).
// Host code must use that pointer, never the array's name.
constexpr int NOALLOC_MAX_BATCH = 65536;   // instances per call
constexpr int NOALLOC_MAX_NVAR = 400;      // variables of one QP (horizon 200, both axes)
constexpr int NOALLOC_SLOTS = 8;           // J / R slots of the dense stage
constexpr size_t SCRATCH_LIST_OFF = 256;
constexpr size_t SCRATCH_WIDE_OFF = SCRATCH_LIST_OFF + (size_t)NOALLOC_MAX_BATCH * sizeof(int);
constexpr size_t SCRATCH_J_OFF = SCRATCH_WIDE_OFF + (size_t)NOALLOC_MAX_BATCH * sizeof(int);
constexpr size_t SCRATCH_SLOT_BYTES = (size_t)NOALLOC_MAX_NVAR * NOALLOC_MAX_NVAR * sizeof(double);
constexpr size_t SCRATCH_R_OFF = SCRATCH_J_OFF + NOALLOC_SLOTS * SCRATCH_SLOT_BYTES;
// hand-over records of the first stage for its wide edition, one per instance (kernels.h)
constexpr size_t SCRATCH_WARM_I_OFF = SCRATCH_R_OFF + NOALLOC_SLOTS * SCRATCH_SLOT_BYTES;
constexpr size_t SCRATCH_WARM_D_OFF = SCRATCH_WARM_I_OFF + (size_t)NOALLOC_MAX_BATCH * mpcb::WARM_STRIDE * sizeof(int);
// list of the instances the wide edition leaves to the block edition
constexpr size_t SCRATCH_BLOCK_OFF = SCRATCH_WARM_D_OFF + (size_t)NOALLOC_MAX_BATCH * mpcb::WARM_STRIDE * sizeof(double);
constexpr size_t SCRATCH_BYTES = SCRATCH_BLOCK_OFF + (size_t)NOALLOC_MAX_BATCH * sizeof(int); // 47.25 MiB
static_assert(SCRATCH_WARM_D_OFF % 256 == 0 && SCRATCH_WARM_I_OFF % 256 == 0 && SCRATCH_BLOCK_OFF % 256 == 0,
              "scratch sections stay aligned");
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
// Streams and events mpcb_solve_coupled_groups can spread its groups over besides the caller's
// stream (MPCB_GROUP_LANES=2 in the environment; default: every group on the caller's stream).
// Opt-in because of a property of the HIP runtime measured on this platform: a process' streams
// are multiplexed onto a few hardware queues per priority level (four by default), and once the
// calling stream shares a hardware queue with any other stream, a fork/join issued call after
// call without a host synchronisation in between -- so that the waits reach the device as
// barrier packets on another queue's signal -- stops the queues for good.  Reproduced with
// nothing but torch streams and sleep kernels (scripts/stream_forkjoin_torch_only.py); the same
// chain completes when the process has no more streams than hardware queues.  The two lanes are
// high-priority streams (another pool of hardware queues), created on first use and kept.
constexpr int GROUP_LANES = 2;
struct GroupLanes {
    hipStream_t stream[GROUP_LANES] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[GROUP_LANES] = {nullptr, nullptr};
    bool ready = false;
};
static GroupLanes g_lanes[MAX_DEVICES];
static std::mutex g_lanes_mutex;
static unsigned char *g_scratch[MAX_DEVICES] = {nullptr};
static std::mutex g_scratch_mutex;

struct mpcb_handle_s {
    mpcb_params params;
    mpcb::HostSetup hs;
    mpcb::DevSetup ds;
    mpcb::DenseWs ws;
    int list_capacity = 0;   // entries ws.list can hold
    int dense_slots = 0;     // slots of ws.J / ws.R (HBM-resident dense factors only)
    int slot_target = 16;    // slots wanted: raised to the largest hand-over count seen (<= 512)
    bool dense_in_lds = false; // the dense kernel keeps J and R in LDS: no HBM slots at all
    bool wide_fits = false;    // the wide edition of the first-stage kernel fits the LDS for this shape
    bool block_fits = false;   // ... and behind it the block edition, run by the dense kernel's workgroups
    // the counter block is known to be zero (every completed solve leaves it so): no memset needed
    bool counters_clean = false;
    // longest-first dispatch of the first stage (mpcb_set_dispatch_hint): the permutation and the
    // running-time records it is built from -- a ring of dispatch_hint records, so that a solve is
    // ordered by the record of the solve that many calls before it, which it then overwrites
    int dispatch_hint = MPCB_DISPATCH_IN_ORDER;
    int *order = nullptr;
    int *order_rec = nullptr;   // MPCB_DISPATCH_MAX_LAG records of ORDER_HEADER + order_capacity ints
    int order_capacity = 0;
    unsigned order_calls = 0;   // ordered solves so far: the ring position
    std::vector<void *> dev_allocs;   // tables: freed by mpcb_destroy only
    // hand-over count of the latest finished solve: one int of page-locked host memory that the
    // dense kernel's last workgroup stores to (DenseWs::host_count); read, without waiting, at the
    // start of a solve to size the dense workspace
    int *count_host = nullptr;
    hipStream_t own_stream = nullptr; // stream of the host-pointer entry points (created by the first of them)
    // COUPLED_REF handles of other CoM heights (same parameters otherwise), keyed by the bits of
    // the height: a host-pointer batch that brings per-instance heights is split over them
    std::map<uint64_t, mpcb_handle> by_height;
    std::map<uint64_t, uint64_t> height_used; // call number of the last use (for eviction)
    uint64_t height_calls = 0;
    // COUPLED_REF with a free (per-knot) ZMP reference: Sp, Wp, F2, F3 with one segment per knot
    const double *knot_Sp = nullptr, *knot_Wp = nullptr, *knot_F2 = nullptr, *knot_F3 = nullptr;
    // device arena of the host-pointer entry points: grown when a batch needs more, kept
    // between calls (so they neither allocate nor free in steady state)
    void *stage_dev = nullptr;
    size_t stage_bytes = 0;
    // host thread of this handle's shard in mpcb_solve_coupled_multi: started by the first such
    // call, parked on `cv` between calls, ended by mpcb_destroy
    struct Worker {
        std::thread thread;
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool has_job = false, done = false, quit = false;
    } *worker = nullptr;
};

namespace {

template <class T> int upload_to(mpcb_handle h, const std::vector<T> &v, const T **dst) {
    T *p = nullptr;
    int rc = upload(v, &p);
    if (rc) return rc;
    if (p) h->dev_allocs.push_back(p);
    *dst = p;
    return 0;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int ensure_own_stream(mpcb_handle h) {
    if (!h->own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    return 0;
}

// Scratch copies for the host-pointer entry points.
// Device copies of host arrays, carved out of the handle's arena.  The staging sequence of an
// entry point runs twice: a dry pass that only adds up the sizes, commit() (which grows the
// arena if needed), then the pass that hands out the slices and copies the inputs.
struct Staging {
    mpcb_handle h;
    bool dry = true;
    size_t need = 0, used = 0;
    explicit Staging(mpcb_handle handle) : h(handle) {}
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    int commit() {
        if (need > h->stage_bytes) {
            HIP_TRY(hipStreamSynchronize(h->own_stream));
            if (h->stage_dev) HIP_TRY(hipFree(h->stage_dev));
            h->stage_dev = nullptr;
            h->stage_bytes = 0;
            HIP_TRY(hipMalloc(&h->stage_dev, need));
            h->stage_bytes = need;
        }
        dry = false;
        used = 0;
        return 0;
    }
    template <class T> T *slice(size_t n) {
        T *p = reinterpret_cast<T *>(static_cast<char *>(h->stage_dev) + used);
        used += padded(n * sizeof(T));
        return p;
    }
    template <class T> int in(const T *host, size_t n, const T **dev) {
        *dev = nullptr;
        if (!host || !n) return 0;
        if (dry) {
            need += padded(n * sizeof(T));
            return 0;
        }
        T *p = slice<T>(n);
        HIP_TRY(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, h->own_stream));
        *dev = p;
        return 0;
    }
    template <class T> int out(T *host, size_t n, T **dev) {
        *dev = nullptr;
        if (!host || !n) return 0;
        if (dry) {
            need += padded(n * sizeof(T));
            return 0;
        }
        *dev = slice<T>(n);
        return 0;
    }
};

int stage_outputs(Staging &st, const mpcb_outputs &h, mpcb_outputs &d, size_t B, size_t N) {
    int rc = 0;
    d = h;
#define OUT_(field, n) \
    if ((rc = st.out(h.field, (n), &d.field))) return rc;
    OUT_(jerk_x, B * N) OUT_(jerk_y, B * N) OUT_(com_x, B * N) OUT_(com_y, B * N)
    OUT_(com_xd, B * N) OUT_(com_yd, B * N) OUT_(zmp_x, B * N) OUT_(zmp_y, B * N)
    OUT_(min_margin, B * N) OUT_(cost, B) OUT_(status, B) OUT_(iters, B) OUT_(n_active, B)
    OUT_(active_idx, B * (size_t)h.active_stride) OUT_(multipliers, B * (size_t)h.active_stride)
    OUT_(alg_flops, B) OUT_(slacks, B * 2 * N)
#undef OUT_
    return 0;
}

// queues the copies back behind the kernels on `stream` and waits for all of them
int fetch_outputs(const mpcb_outputs &h, const mpcb_outputs &d, size_t B, size_t N, hipStream_t stream) {
#define BACK_(field, n)                                                                       \
    if (h.field)                                                                              \
        HIP_TRY(hipMemcpyAsync(h.field, d.field, (n) * sizeof(*h.field), hipMemcpyDeviceToHost, stream));
    BACK_(jerk_x, B * N) BACK_(jerk_y, B * N) BACK_(com_x, B * N) BACK_(com_y, B * N)
    BACK_(com_xd, B * N) BACK_(com_yd, B * N) BACK_(zmp_x, B * N) BACK_(zmp_y, B * N)
    BACK_(min_margin, B * N) BACK_(cost, B) BACK_(status, B) BACK_(iters, B) BACK_(n_active, B)
    BACK_(active_idx, B * (size_t)h.active_stride) BACK_(multipliers, B * (size_t)h.active_stride)
    BACK_(alg_flops, B) BACK_(slacks, B * 2 * N)
#undef BACK_
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

// Hands `job` to the handle's parked host thread (started on first use).
void worker_submit(mpcb_handle h, std::function<void()> job) {
    if (!h->worker) {
        h->worker = new mpcb_handle_s::Worker();
        mpcb_handle_s::Worker *w = h->worker;
        w->thread = std::thread([w]() {
            std::unique_lock<std::mutex> lock(w->m);
            for (;;) {
                w->cv.wait(lock, [w]() { return w->has_job || w->quit; });
                if (w->quit) return;
                std::function<void()> job = std::move(w->job);
                w->has_job = false;
                lock.unlock();
                job();
                lock.lock();
                w->done = true;
                w->cv.notify_all();
            }
        });
    }
    mpcb_handle_s::Worker *w = h->worker;
    {
        std::lock_guard<std::mutex> lock(w->m);
        w->job = std::move(job);
        w->has_job = true;
        w->done = false;
    }
    w->cv.notify_all();
}

void worker_wait(mpcb_handle h) {
    mpcb_handle_s::Worker *w = h->worker;
    std::unique_lock<std::mutex> lock(w->m);
    w->cv.wait(lock, [w]() { return w->done; });
}

void worker_stop(mpcb_handle h) {
    mpcb_handle_s::Worker *w = h->worker;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lock(w->m);
        w->quit = true;
    }
    w->cv.notify_all();
    w->thread.join();
    delete w;
    h->worker = nullptr;
}

// Waits for the stream when the scope is left early: copies queued into (or out of) host
// vectors of the enclosing function must not outlive them on an error path.
struct DrainOnExit {
    hipStream_t stream;
    bool armed = true;
    explicit DrainOnExit(hipStream_t s) : stream(s) {}
    ~DrainOnExit() {
        if (armed) (void)hipStreamSynchronize(stream);
    }
};

int count_not_ok(const int32_t *status, int B) {
    int n = 0;
    if (status)
        for (int i = 0; i < B; i++) n += status[i] != MPCB_OK;
    return n;
}

} // namespace

extern "C" {

const char *mpcb_last_error(void) { return g_err.c_str(); }

// Host part of a handle: parameters checked, layout of one QP.
static int create_begin(const mpcb_params *params, mpcb_handle *out) {
    *out = nullptr;
    mpcb_handle h = new mpcb_handle_s();
    h->params = *params;
    const char *err = "";
    int rc = mpcb::build_layout(*params, h->hs, &err);
    if (rc == 0 && params->table_build != MPCB_TABLES_DEVICE && params->table_build != MPCB_TABLES_HOST) {
        rc = MPCB_E_ARG;
        err = "table_build must be MPCB_TABLES_DEVICE or MPCB_TABLES_HOST";
    }
    if (rc) {
        delete h;
        return fail(rc, err);
    }
    *out = h;
    return 0;
}

// Points the kernels' argument blocks at tables built on the device (the handle already owns
// the allocation they live in).
static void adopt_tables(mpcb_handle h, const mpcb::TableSet &t) {
    mpcb::DevSetup &ds = h->ds;
    ds.S = t.S; ds.Wt = t.Wt; ds.Sc = t.Sc; ds.Sv = t.Sv;
    ds.Kx = t.Kx; ds.Kz = t.Kz; ds.Kc = t.Kc; ds.Kv = t.Kv;
    ds.Sp = t.Sp; ds.Wp = t.Wp; ds.F2 = t.F2; ds.F3 = t.F3;
    ds.Spp = t.Spp; ds.Sps = t.Sps; ds.Saa = t.Saa; ds.Wtp = t.Wtp; ds.Wta = t.Wta;
    ds.Kzp = t.Kzp; ds.Kza = t.Kza;
    h->ws.J1 = t.J0;
    h->ws.zu_col = t.zu_col; h->ws.xp_col = t.xp_col; h->ws.xv_col = t.xv_col;
    h->ws.zx = t.zx; h->ws.xpx = t.xpx; h->ws.xvx = t.xvx;
    std::memcpy(h->hs.F, t.F, sizeof(h->hs.F));
    mpcb::finish_scalars(h->params, h->hs, t.traceG1, t.traceJ1);
}

// Everything after the tables: scalars of the argument blocks, bookkeeping arrays, counters,
// stream.  `tables` = the device-built tables, or NULL to build them on the host (the checker).
static int create_finish(mpcb_handle h, const mpcb::TableSet *tables) {
    const mpcb_params *params = &h->params;
    int rc = 0;
    if (params->device < MAX_DEVICES) {
        // address of this device's copy of the scratch array (see its declaration)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch[params->device]) {
            unsigned char *ptr = nullptr;
            hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr)
                return fail(MPCB_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
            g_scratch[params->device] = ptr;
        }
    }
    mpcb::HostSetup &hs = h->hs;
    mpcb::DevSetup &ds = h->ds;
    std::memset(&ds, 0, sizeof(ds));
    std::memset(&h->ws, 0, sizeof(h->ws));
    if (tables) {
        adopt_tables(h, *tables);
    } else {
        const char *err = "";
        if ((rc = mpcb::build_tables_host(*params, hs, &err))) return fail(rc, err);
        if ((rc = upload_to(h, hs.S, &ds.S)) || (rc = upload_to(h, hs.Wt, &ds.Wt)) ||
            (rc = upload_to(h, hs.Sc, &ds.Sc)) || (rc = upload_to(h, hs.Sv, &ds.Sv)) ||
            (rc = upload_to(h, hs.Kx, &ds.Kx)) || (rc = upload_to(h, hs.Kz, &ds.Kz)) ||
            (rc = upload_to(h, hs.Kc, &ds.Kc)) || (rc = upload_to(h, hs.Kv, &ds.Kv)) ||
            (rc = upload_to(h, hs.Sp, &ds.Sp)) || (rc = upload_to(h, hs.Wp, &ds.Wp)) ||
            (rc = upload_to(h, hs.F2, &ds.F2)) || (rc = upload_to(h, hs.F3, &ds.F3)) ||
            (rc = upload_to(h, hs.J0, &h->ws.J1)) || (rc = upload_to(h, hs.zu_col, &h->ws.zu_col)) ||
            (rc = upload_to(h, hs.xp_col, &h->ws.xp_col)) || (rc = upload_to(h, hs.xv_col, &h->ws.xv_col)) ||
            (rc = upload_to(h, hs.zx, &h->ws.zx)) || (rc = upload_to(h, hs.xpx, &h->ws.xpx)) ||
            (rc = upload_to(h, hs.xvx, &h->ws.xvx)))
            return rc;
    }
    ds.N = hs.N;
    ds.m = hs.m;
    ds.axes = hs.axes;
    ds.nvar = hs.nvar;
    ds.slack_ls = hs.slack_ls;
    ds.slack_g0 = hs.slack_g0;
    ds.form = params->formulation;
    ds.steps = hs.steps;
    ds.phase_knots = hs.phase_knots;
    ds.nphase = hs.nphase;
    ds.nref = hs.nphase;
    const int n = hs.nvar;
    ds.iter_cap = params->iter_cap > 0 ? params->iter_cap : 50 * (n + hs.m);
    std::memcpy(ds.F, hs.F, sizeof(ds.F));
    ds.psi_thr = (double)hs.m * 2.220446049250313e-16 * hs.c1 * hs.c2 * 100.0;
    ds.traceG1 = hs.traceG1;
    ds.Ts = params->Ts;
    ds.inv_gravity = 1.0 / params->gravity;
    std::memcpy(ds.hip, params->hip_offsets, sizeof(ds.hip));
    if ((rc = upload_to(h, hs.knot_phase, &ds.knot_phase)) || (rc = upload_to(h, hs.knot_row0, &ds.knot_row0)) ||
        (rc = upload_to(h, hs.row_knot, &h->ws.row_knot)) || (rc = upload_to(h, hs.row_slot, &h->ws.row_slot)))
        return rc;
    {
        int *cnt = nullptr;
        if (hipMalloc((void **)&cnt, mpcb::COUNTER_INTS * sizeof(int)) != hipSuccess) // the counter block (kernels.h)
            return fail(MPCB_E_HIP, "hipMalloc failed");
        h->dev_allocs.push_back(cnt);
        h->ws.count = cnt;
        h->ws.wide_count = cnt + mpcb::CNT_WIDE;
    }
    if (params->formulation != MPCB_UNCONSTRAINED) {
        // The kernels' LDS carve-ups must fit a workgroup: found out here, not at the first solve.
        // A COUPLED_REF handle may later be given a per-knot ZMP reference, whose segments are the
        // N knots instead of the footstep phases: sized for the larger of the two.
        mpcb::DevSetup sized = ds;
        if (params->formulation == MPCB_COUPLED_REF && sized.N > sized.nref) sized.nref = sized.N;
        h->wide_fits = params->formulation != MPCB_BOX_1D && params->formulation != MPCB_BOX_SLACK &&
                       mpcb::zspace_wide_lds_bytes(sized) <= 160 * 1024 - 256;
        h->block_fits = h->wide_fits && params->solver_mode != MPCB_MODE_DENSE && mpcb::block_edition_applies(sized);
        const size_t lds = mpcb::dense_lds_bytes_needed(sized, &h->dense_in_lds, h->block_fits);
        if (lds > 160 * 1024)
            return fail(MPCB_E_ARG, "horizon / number_of_steps too large for this formulation: the dense "
                                    "second-stage kernel would need " + std::to_string(lds) +
                                    " bytes of LDS per workgroup (limit 163840)");
    }
    if (hipHostMalloc((void **)&h->count_host, sizeof(int), hipHostMallocMapped) != hipSuccess)
        return fail(MPCB_E_HIP, "allocating the handle's pinned counter failed");
    *h->count_host = 0;
    if (hipHostGetDevicePointer((void **)&h->ws.host_count, h->count_host, 0) != hipSuccess)
        return fail(MPCB_E_HIP, "hipHostGetDevicePointer failed");
    // (diagnostic: take the handle's stream -- and with it a share of the process' hardware queues --
    // at creation, as earlier versions of this library did, instead of at the first host-pointer call)
    if (std::getenv("MPCB_DEBUG_EAGER_STREAM") && (rc = ensure_own_stream(h))) return rc;
    return 0;
}

int mpcb_create_groups(const mpcb_params *params, int32_t n, mpcb_handle *out) {
    if (!params || !out || n <= 0) return fail(MPCB_E_ARG, "null argument or no parameter sets");
    for (int q = 0; q < n; q++) out[q] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MPCB_E_NOGPU, "no HIP device available");
    const int dev = params[0].device;
    if (dev < 0 || dev >= ndev) return fail(MPCB_E_ARG, "device ordinal out of range");
    for (int q = 1; q < n; q++)
        if (params[q].device != dev) return fail(MPCB_E_ARG, "the parameter sets of one call must name one device");
    int rc = 0;
    auto undo = [&](int code) {
        const std::string msg = g_err;
        for (int q = 0; q < n; q++) {
            if (out[q]) mpcb_destroy(out[q]);
            out[q] = nullptr;
        }
        g_err = msg;
        return code;
    };
    for (int q = 0; q < n; q++)
        if ((rc = create_begin(&params[q], &out[q]))) {
            if (n > 1) g_err = "parameter set " + std::to_string(q) + ": " + g_err;
            return undo(rc);
        }
    DeviceGuard guard(dev);
    if (!guard.ok) {
        fail(MPCB_E_HIP, "hipSetDevice failed");
        return undo(MPCB_E_HIP);
    }
    // the tables of every set that asks for the device build: one chain of launches
    std::vector<int> which;
    std::vector<mpcb::TableSpec> specs;
    for (int q = 0; q < n; q++)
        if (params[q].table_build == MPCB_TABLES_DEVICE) {
            which.push_back(q);
            specs.emplace_back();
            mpcb::table_spec(params[q], out[q]->hs, specs.back());
        }
    std::vector<mpcb::TableSet> sets(specs.size());
    if (!specs.empty()) {
        std::string err;
        if ((rc = mpcb::build_tables_device(specs.data(), (int)specs.size(), sets.data(), nullptr, &err))) {
            fail(rc, err);
            return undo(rc);
        }
    }
    // every set's allocation belongs to its handle from here on (mpcb_destroy frees it)
    std::vector<const mpcb::TableSet *> mine(n, nullptr);
    for (size_t k = 0; k < which.size(); k++) {
        mine[which[k]] = &sets[k];
        out[which[k]]->dev_allocs.push_back(sets[k].block);
    }
    for (int q = 0; q < n; q++)
        if ((rc = create_finish(out[q], mine[q]))) return undo(rc);
    return 0;
}

int mpcb_create(const mpcb_params *params, mpcb_handle *out) {
    if (!params || !out) return fail(MPCB_E_ARG, "null argument");
    return mpcb_create_groups(params, 1, out);
}

namespace {

bool first_stage_runs(mpcb_handle h) {
    return h->params.solver_mode != MPCB_MODE_DENSE && h->params.formulation != MPCB_BOX_SLACK &&
           h->params.formulation != MPCB_UNCONSTRAINED;
}

void release_order_buffers(mpcb_handle h) {
    if (h->order) (void)hipFree(h->order);
    if (h->order_rec) (void)hipFree(h->order_rec);
    h->order_rec = nullptr;
    h->order = nullptr;
    h->order_capacity = 0;
}

hipError_t forget_order_records(mpcb_handle h, hipStream_t stream) {
    if (!h->order_rec) return hipSuccess;
    const size_t rec = mpcb::ORDER_HEADER + (size_t)h->order_capacity;
    for (int r = 0; r < MPCB_DISPATCH_MAX_LAG; r++) {
        hipError_t e = hipMemsetAsync(h->order_rec + r * rec, 0, mpcb::ORDER_HEADER * sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Sizes the hand-over list for a batch of B and, when the dense kernel keeps its factors in
// HBM, its J / R slots: a small number to begin with, raised to the largest hand-over count an
// earlier solve of this handle reported (the kernel loops over the list with as many workgroups
// as there are slots, so fewer slots than hand-overs is slower, never wrong).  Superseded
// buffers are freed here, after the stream has drained.
int ensure_workspace(mpcb_handle h, int B, hipStream_t stream) {
    {
        const int seen = *static_cast<volatile int *>(h->count_host);
        if (seen > h->slot_target) h->slot_target = seen < 512 ? seen : 512;
    }
    if (B > h->list_capacity) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (h->ws.list) HIP_TRY(hipFree(h->ws.list));
        if (h->ws.wide_list) HIP_TRY(hipFree(h->ws.wide_list));
        if (h->ws.block_list) HIP_TRY(hipFree(h->ws.block_list));
        if (h->ws.warm_i) HIP_TRY(hipFree(h->ws.warm_i));
        if (h->ws.warm_d) HIP_TRY(hipFree(h->ws.warm_d));
        h->ws.list = nullptr;
        h->ws.wide_list = nullptr;
        h->ws.block_list = nullptr;
        h->ws.warm_i = nullptr;
        h->ws.warm_d = nullptr;
        h->list_capacity = 0;
        int cap = 1024;
        while (cap < B) cap *= 2;
        HIP_TRY(hipMalloc((void **)&h->ws.list, (size_t)cap * sizeof(int)));
        if (h->params.formulation != MPCB_BOX_1D && h->params.formulation != MPCB_BOX_SLACK && h->wide_fits)
            HIP_TRY(hipMalloc((void **)&h->ws.wide_list, (size_t)cap * sizeof(int)));
        if (h->block_fits) {
            HIP_TRY(hipMalloc((void **)&h->ws.block_list, (size_t)cap * sizeof(int)));
            const char *e = std::getenv("MPCB_DEBUG_BLOCK_ROWS"); // (tests of the fall-through to the full method)
            h->ws.block_rows = e ? std::atoi(e) : 1 << 30;
        }
        const bool coupled_wide = h->params.formulation != MPCB_BOX_1D && h->params.formulation != MPCB_BOX_SLACK &&
                                  h->wide_fits;
        if ((mpcb::box_dense_applies(h->ds) || coupled_wide) && !std::getenv("MPCB_DEBUG_COLD_SECOND_STAGE")) { // hand-over records, one per instance of the batch
            HIP_TRY(hipMalloc((void **)&h->ws.warm_i, (size_t)cap * mpcb::WARM_STRIDE * sizeof(int)));
            HIP_TRY(hipMalloc((void **)&h->ws.warm_d, (size_t)cap * mpcb::WARM_STRIDE * sizeof(double)));
        }
        h->list_capacity = cap;
    }
    if (h->dispatch_hint != MPCB_DISPATCH_IN_ORDER && first_stage_runs(h) && h->order_capacity < h->list_capacity) {
        HIP_TRY(hipStreamSynchronize(stream));
        release_order_buffers(h);
        const size_t cap = (size_t)h->list_capacity;
        HIP_TRY(hipMalloc((void **)&h->order, cap * sizeof(int)));
        const size_t rec = mpcb::ORDER_HEADER + cap;
        HIP_TRY(hipMalloc((void **)&h->order_rec, MPCB_DISPATCH_MAX_LAG * rec * sizeof(int)));
        h->order_capacity = (int)cap;
        // batch size 0 in a header is "no record": solves run in index order until there is one
        HIP_TRY(forget_order_records(h, stream));
    }
    if (h->dense_in_lds) return 0;
    int want = h->params.solver_mode == MPCB_MODE_DENSE ? 512 : h->slot_target;
    if (want > B) want = B;
    if (want > h->dense_slots) {
        const size_t n = (size_t)h->hs.nvar;
        HIP_TRY(hipStreamSynchronize(stream));
        if (h->ws.J) HIP_TRY(hipFree(h->ws.J));
        if (h->ws.R) HIP_TRY(hipFree(h->ws.R));
        h->ws.J = h->ws.R = nullptr;
        h->dense_slots = 0;
        HIP_TRY(hipMalloc((void **)&h->ws.J, (size_t)want * n * n * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&h->ws.R, (size_t)want * n * n * sizeof(double)));
        h->dense_slots = want;
    }
    return 0;
}

// Builds and uploads the height-free response tables the first time a batch brings its own
// heights (one-off host work and allocations, like mpcb_create's).
int ensure_height_tables(mpcb_handle h) {
    if (h->ds.Spp) return 0; // (handles whose tables were built on the device have them from the start)
    const char *err = "";
    int rc = mpcb::build_height_tables(h->params, h->hs, &err);
    if (rc) return fail(rc, err);
    mpcb::DevSetup &ds = h->ds;
    const mpcb::HostSetup &hs = h->hs;
    if ((rc = upload_to(h, hs.Sps, &ds.Sps)) || (rc = upload_to(h, hs.Saa, &ds.Saa)) ||
        (rc = upload_to(h, hs.Wtp, &ds.Wtp)) || (rc = upload_to(h, hs.Wta, &ds.Wta)) ||
        (rc = upload_to(h, hs.Kzp, &ds.Kzp)) || (rc = upload_to(h, hs.Kza, &ds.Kza)) ||
        (rc = upload_to(h, hs.Spp, &ds.Spp)))
        return rc;
    return 0;
}

// COUPLED_REF tables for a reference that changes from knot to knot: the per-phase tables
// (setup.cpp) with every knot a phase of its own -- Sp = Qs S, Wp = Qs W^T,
// F2[a][k] = -Qs (Zu Kx)[k][a], F3 = -Qs^2 S / 2 (g0 gains -Qs Zu' zref, MPCPlanner.cpp:731-735).
// Built and uploaded the first time such a reference is passed.
int ensure_knot_reference_tables(mpcb_handle h) {
    if (h->knot_Sp) return 0;
    const size_t N = (size_t)h->hs.N;
    double *blk = nullptr;
    HIP_TRY(hipMalloc((void **)&blk, (3 * N * N + 4 * N) * sizeof(double)));
    h->dev_allocs.push_back(blk);
    double *Sp = blk, *Wp = blk + N * N, *F3 = blk + 2 * N * N, *F2 = blk + 3 * N * N;
    HIP_TRY(mpcb::launch_knot_reference_tables((int)N, h->params.weight_Qs, h->ds.S, h->ds.Wt, h->ds.Kz, h->ws.zx,
                                               Sp, Wp, F2, F3, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    h->knot_Wp = Wp;
    h->knot_F2 = F2;
    h->knot_F3 = F3;
    h->knot_Sp = Sp;
    return 0;
}

// workgroups of the wide edition of the first-stage kernel (four fit a compute unit)
constexpr int WIDE_GRID = 1024;

// first stage (wave per QP; for the coupled forms its wide edition over the instances whose
// working set outgrew the first), then the dense kernel over whatever is left
int enqueue_two_stage(mpcb_handle h, const mpcb::DevIn &di, const mpcb_outputs &out, int B,
                      const mpcb::DenseWs &ws, int dense_grid, hipStream_t stream);

// (`clean`: the counter block at ws.count is zero -- the dense kernel of every solve that was
// enqueued completely leaves it so; otherwise it is cleared here first)
int launch_two_stage(mpcb_handle h, const mpcb::DevIn &di, const mpcb_outputs &out, int B,
                     const mpcb::DenseWs &ws, int dense_grid, hipStream_t stream, bool *clean) {
    if (!*clean) HIP_TRY(hipMemsetAsync(ws.count, 0, mpcb::COUNTER_INTS * sizeof(int), stream));
    *clean = false;
    const int rc = enqueue_two_stage(h, di, out, B, ws, dense_grid, stream);
    // (a stream that is being captured records the launches without running them: then nothing
    // is known about the counters when the next call comes)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (rc == 0 && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone)
        *clean = true;
    return rc;
}

int enqueue_two_stage(mpcb_handle h, const mpcb::DevIn &di, const mpcb_outputs &out, int B,
                      const mpcb::DenseWs &ws, int dense_grid, hipStream_t stream) {
    if (di.zref_knot_mode) {
        // same kernels, the reference's segments are the knots
        mpcb::DevSetup ds = h->ds;
        ds.Sp = h->knot_Sp;
        ds.Wp = h->knot_Wp;
        ds.F2 = h->knot_F2;
        ds.F3 = h->knot_F3;
        ds.nref = ds.N;
        if (ws.order) HIP_TRY(mpcb::launch_dispatch_order(ws, B, stream));
        if (h->params.solver_mode == MPCB_MODE_DENSE) {
            HIP_TRY(mpcb::launch_list_all(ws, B, stream));
        } else {
            HIP_TRY(mpcb::launch_zspace(ds, di, out, B, ws, stream));
            HIP_TRY(mpcb::launch_zspace_wide(ds, di, out, B < WIDE_GRID ? B : WIDE_GRID, ws, stream));
        }
        HIP_TRY(mpcb::launch_dense(ds, di, out, ws, dense_grid, stream));
        return 0;
    }
    // the slack formulation keeps about 2N rows active: beyond the first stage's capacity
    if (h->params.solver_mode == MPCB_MODE_DENSE || h->params.formulation == MPCB_BOX_SLACK) {
        // no first stage: nothing to take over, the records (if any) are stale
        mpcb::DenseWs cold = ws;
        cold.warm_i = nullptr;
        cold.warm_d = nullptr;
        cold.block_list = nullptr;
        HIP_TRY(mpcb::launch_list_all(cold, B, stream));
        HIP_TRY(mpcb::launch_dense(h->ds, di, out, cold, dense_grid, stream));
        return 0;
    } else {
        if (ws.order) HIP_TRY(mpcb::launch_dispatch_order(ws, B, stream));
        HIP_TRY(mpcb::launch_zspace(h->ds, di, out, B, ws, stream));
        HIP_TRY(mpcb::launch_zspace_wide(h->ds, di, out, B < WIDE_GRID ? B : WIDE_GRID, ws, stream));
    }
    HIP_TRY(mpcb::launch_dense(h->ds, di, out, ws, dense_grid, stream));
    return 0;
}

int solve_two_stage(mpcb_handle h, const mpcb::DevIn &di, const mpcb_outputs &out, int B, hipStream_t stream) {
    if (B == 0) return 0;
    int rc = ensure_workspace(h, B, stream);
    if (rc) return rc;
    const int cap = h->dense_in_lds ? 512 : h->dense_slots;
    mpcb::DenseWs ws = h->ws;
    if (h->dispatch_hint != MPCB_DISPATCH_IN_ORDER && first_stage_runs(h)) {
        ws.order = h->order;
        ws.order_record = h->order_rec + (size_t)(h->order_calls++ % (unsigned)h->dispatch_hint) *
                                             (mpcb::ORDER_HEADER + (size_t)h->order_capacity);
    }
    if ((rc = launch_two_stage(h, di, out, B, ws, B < cap ? B : cap, stream, &h->counters_clean))) return rc;
    return 0;
}

} // namespace

void mpcb_destroy(mpcb_handle h) {
    if (!h) return;
    worker_stop(h);
    for (auto &kv : h->by_height) mpcb_destroy(kv.second);
    h->by_height.clear();
    {
        DeviceGuard guard(h->params.device);
        // solves still in flight on the caller's streams read the tables and write the pinned counter
        (void)hipDeviceSynchronize();
        if (h->own_stream) {
            (void)hipStreamSynchronize(h->own_stream);
            (void)hipStreamDestroy(h->own_stream);
        }
        for (void *p : h->dev_allocs) (void)hipFree(p);
        if (h->ws.list) (void)hipFree(h->ws.list);
        if (h->ws.wide_list) (void)hipFree(h->ws.wide_list);
        if (h->ws.block_list) (void)hipFree(h->ws.block_list);
        if (h->ws.warm_i) (void)hipFree(h->ws.warm_i);
        if (h->ws.warm_d) (void)hipFree(h->ws.warm_d);
        if (h->ws.J) (void)hipFree(h->ws.J);
        if (h->ws.R) (void)hipFree(h->ws.R);
        release_order_buffers(h);
        if (h->stage_dev) (void)hipFree(h->stage_dev);
        if (h->count_host) (void)hipHostFree(h->count_host);
    }
    delete h;
}

int mpcb_num_constraints(mpcb_handle h) { return h ? h->hs.m : MPCB_E_ARG; }
int mpcb_num_phases(mpcb_handle h) { return h ? h->hs.nphase : MPCB_E_ARG; }

// Diagnostic builds (-DMPCB_STAMPS): device buffer of 16 cycle counters per instance.
int mpcb_debug_set_stamp_buffer(mpcb_handle h, long long *dev) {
    if (!h) return MPCB_E_ARG;
    h->ds.dbg = dev;
    return 0;
}

int mpcb_set_dispatch_hint(mpcb_handle h, int32_t mode) {
    if (!h) return fail(MPCB_E_ARG, "null handle");
    if (mode < MPCB_DISPATCH_IN_ORDER || mode > MPCB_DISPATCH_MAX_LAG)
        return fail(MPCB_E_ARG, "dispatch hint out of range (0 .. MPCB_DISPATCH_MAX_LAG)");
    if (mode != h->dispatch_hint && h->order_rec) {
        // another lag: the ring's records no longer sit where the new stride looks for them
        DeviceGuard guard(h->params.device);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(forget_order_records(h, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        h->order_calls = 0;
    }
    h->dispatch_hint = mode;
    return 0;
}

int mpcb_last_wide_count(mpcb_handle h) {
    if (!h) return fail(MPCB_E_ARG, "null handle");
    if (!h->ws.wide_count || !h->ws.wide_list) return 0;
    DeviceGuard guard(h->params.device);
    int n = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, h->ws.count + mpcb::CNT_LAST_WIDE, sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int mpcb_last_block_count(mpcb_handle h) {
    if (!h) return fail(MPCB_E_ARG, "null handle");
    if (!h->ws.count || !h->ws.block_list) return 0;
    DeviceGuard guard(h->params.device);
    int n = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, h->ws.count + mpcb::CNT_LAST_BLOCK, sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int mpcb_last_handover_count(mpcb_handle h) {
    if (!h) return fail(MPCB_E_ARG, "null handle");
    if (!h->ws.count) return 0;
    DeviceGuard guard(h->params.device);
    int n = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, h->ws.count + mpcb::CNT_LAST_HANDOVER, sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int mpcb_get_matrix(mpcb_handle h, int32_t which, double *dst) {
    if (!h) return fail(MPCB_E_ARG, "null argument");
    const mpcb::DevSetup &ds = h->ds;
    const size_t N = (size_t)h->hs.N, P = (size_t)h->hs.nphase;
    const double *src = nullptr;
    size_t count = 0;
    switch (which) {
    case 0: src = ds.S; count = N * N; break;
    case 1: src = ds.Wt; count = N * N; break;
    case 2: src = ds.Sc; count = N * N; break;
    case 3: src = ds.Sv; count = N * N; break;
    case 4: src = ds.Kz; count = N * 4; break;
    case 5: src = ds.Kx; count = N * 4; break;
    case 6: src = h->ws.zu_col; count = N; break;
    case 7: src = h->ws.zx; count = N * 3; break;
    case 8: src = h->ws.xp_col; count = N; break;
    case 9: src = h->ws.xpx; count = N * 3; break;
    case 10: src = h->ws.xv_col; count = N; break;
    case 11: src = h->ws.xvx; count = N * 3; break;
    case 12: src = h->ws.J1; count = N * N; break;
    case 13:
        if (dst) std::memcpy(dst, ds.F, sizeof(ds.F));
        return 16;
    case 14: src = ds.Sp; count = P * N; break;
    case 15: src = ds.Wp; count = P * N; break;
    case 16: src = ds.F2; count = 4 * P; break;
    case 17: src = ds.F3; count = P * P; break;
    case 18: src = ds.Spp; count = N * N; break;
    case 19: src = ds.Sps; count = N * N; break;
    case 20: src = ds.Saa; count = N * N; break;
    case 21: src = ds.Wtp; count = N * N; break;
    case 22: src = ds.Wta; count = N * N; break;
    case 23: src = ds.Kzp; count = N * 4; break;
    case 24: src = ds.Kza; count = N * 4; break;
    default: return fail(MPCB_E_ARG, "unknown matrix id");
    }
    if (!src) return fail(MPCB_E_ARG, "this handle does not hold that matrix");
    if (dst) {
        DeviceGuard guard(h->params.device);
        HIP_TRY(hipMemcpy(dst, src, count * sizeof(double), hipMemcpyDeviceToHost));
    }
    return (int)count;
}

int mpcb_debug_host_table(const mpcb_params *params, int32_t which, double *dst) {
    if (!params) return fail(MPCB_E_ARG, "null argument");
    mpcb::HostSetup hs;
    const char *err = "";
    int rc = mpcb::build_layout(*params, hs, &err);
    if (!rc) rc = mpcb::build_tables_host(*params, hs, &err);
    if (rc) return fail(rc, err);
    const std::vector<double> *src = nullptr;
    std::vector<double> tmp;
    switch (which) {
    case 0: src = &hs.S; break;
    case 1: src = &hs.Wt; break;
    case 2: src = &hs.Sc; break;
    case 3: src = &hs.Sv; break;
    case 4: src = &hs.Kz; break;
    case 5: src = &hs.Kx; break;
    case 6: src = &hs.zu_col; break;
    case 7: src = &hs.zx; break;
    case 8: src = &hs.xp_col; break;
    case 9: src = &hs.xpx; break;
    case 10: src = &hs.xv_col; break;
    case 11: src = &hs.xvx; break;
    case 12: src = &hs.J0; break;
    case 13: tmp.assign(hs.F, hs.F + 16); src = &tmp; break;
    case 14: src = &hs.Sp; break;
    case 15: src = &hs.Wp; break;
    case 16: src = &hs.F2; break;
    case 17: src = &hs.F3; break;
    case 25: tmp = {hs.traceG1, hs.c2 / hs.axes}; src = &tmp; break;
    default: return fail(MPCB_E_ARG, "unknown matrix id");
    }
    if (dst) std::memcpy(dst, src->data(), src->size() * sizeof(double));
    return (int)src->size();
}

// argument checks shared by the coupled entry points; fills the kernels' input block
static int coupled_arguments(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in, const mpcb_outputs *out,
                             mpcb::DevIn &di) {
    if (!h || !in || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    const int form = h->params.formulation;
    if (form != MPCB_COUPLED_JERK && form != MPCB_COUPLED_VEL && form != MPCB_COUPLED_REF)
        return fail(MPCB_E_ARG, "handle was not created for a coupled formulation");
    if (B == 0) return 0;
    if (!in->x0 || !in->y0 || !in->speed) return fail(MPCB_E_ARG, "missing coupled input array");
    if (!in->rows_pqr && !in->phase_pqr && (!in->feet_x || !in->feet_y || !in->swing0))
        return fail(MPCB_E_ARG, "feet_x, feet_y and swing0 are required unless rows_pqr or phase_pqr is given");
    if (in->phase_pqr && in->rows_pqr) return fail(MPCB_E_ARG, "give rows_pqr or phase_pqr, not both");
    if (in->phase_pqr && h->hs.steps <= 0)
        return fail(MPCB_E_ARG, "phase_pqr needs a footstep schedule (number_of_steps > 0)");
    if ((out->active_idx || out->multipliers) && out->active_stride <= 0)
        return fail(MPCB_E_ARG, "active_stride must be positive");
    std::memset(&di, 0, sizeof(di));
    di.x0 = in->x0;
    di.y0 = in->y0;
    di.speed = in->speed;
    di.feet_x = in->feet_x;
    di.feet_y = in->feet_y;
    di.swing0 = in->swing0;
    di.com0 = in->com0;
    di.rows_pqr = in->rows_pqr;
    di.phase_pqr = in->phase_pqr;
    di.zref_phase = in->zref_phase;
    di.height = in->height;
    if (in->zref_knot) {
        if (form != MPCB_COUPLED_REF) return fail(MPCB_E_ARG, "zref_knot is an input of COUPLED_REF only");
        if (in->zref_phase) return fail(MPCB_E_ARG, "give zref_phase or zref_knot, not both");
        di.zref_phase = in->zref_knot; // the kernels index it by segment; here the segments are the knots
        di.zref_knot_mode = 1;
    }
    if ((in->rows_pqr || in->phase_pqr) && form == MPCB_COUPLED_REF && !in->zref_phase && !in->zref_knot)
        return fail(MPCB_E_ARG, "COUPLED_REF with rows_pqr or phase_pqr needs zref_phase or zref_knot");
    if (in->height && form == MPCB_COUPLED_REF)
        return fail(MPCB_E_ARG, "COUPLED_REF with per-instance heights: the Hessian depends on the height, so every "
                                "distinct height is a parameter group -- use mpcb_solve_coupled (host pointers, groups "
                                "the instances itself) or mpcb_create_groups + mpcb_solve_coupled_groups");
    return 0;
}

int mpcb_solve_coupled_device(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in,
                              const mpcb_outputs *out, void *stream) {
    mpcb::DevIn di;
    int rc = coupled_arguments(h, B, in, out, di);
    if (rc || B == 0) return rc;
    DeviceGuard guard(h->params.device);
    if (di.height && (rc = ensure_height_tables(h))) return rc;
    if (di.zref_knot_mode && (rc = ensure_knot_reference_tables(h))) return rc;
    return solve_two_stage(h, di, *out, B, (hipStream_t)stream);
}

int mpcb_solve_coupled_device_noalloc(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in,
                                      const mpcb_outputs *out, void *stream) {
    mpcb::DevIn di;
    int rc = coupled_arguments(h, B, in, out, di);
    if (rc || B == 0) return rc;
    if (B > NOALLOC_MAX_BATCH)
        return fail(MPCB_E_ARG, "mpcb_solve_coupled_device_noalloc: at most " + std::to_string(NOALLOC_MAX_BATCH) +
                                    " instances per call");
    if (h->hs.nvar > NOALLOC_MAX_NVAR)
        return fail(MPCB_E_ARG, "mpcb_solve_coupled_device_noalloc: at most " + std::to_string(NOALLOC_MAX_NVAR) +
                                    " variables per QP (horizon 200)");
    if (h->params.device >= MAX_DEVICES || !g_scratch[h->params.device])
        return fail(MPCB_E_ARG, "mpcb_solve_coupled_device_noalloc: no scratch on this device");
    if (di.zref_knot_mode && !h->knot_Sp)
        return fail(MPCB_E_ARG, "mpcb_solve_coupled_device_noalloc: the tables for a per-knot ZMP reference are built "
                                "(and allocated) by the first mpcb_solve_coupled[_device] call that passes one");
    if (di.height && !h->ds.Spp)
        return fail(MPCB_E_ARG, "mpcb_solve_coupled_device_noalloc: the tables for per-instance heights are built (and "
                                "allocated) by the first mpcb_solve_coupled[_device] call that passes heights");
    DeviceGuard guard(h->params.device);
    // the handle's tables with this device's static scratch in place of the allocated workspace
    unsigned char *ptr = g_scratch[h->params.device];
    mpcb::DenseWs ws = h->ws;
    ws.host_count = nullptr;
    ws.order = nullptr; // index order: the order buffers belong to the allocating entry points
    ws.order_record = nullptr;
    ws.count = reinterpret_cast<int *>(ptr);
    ws.wide_count = ws.count + mpcb::CNT_WIDE;
    ws.list = reinterpret_cast<int *>(ptr + SCRATCH_LIST_OFF);
    ws.wide_list = h->wide_fits ? reinterpret_cast<int *>(ptr + SCRATCH_WIDE_OFF) : nullptr;
    ws.J = reinterpret_cast<double *>(ptr + SCRATCH_J_OFF);
    ws.R = reinterpret_cast<double *>(ptr + SCRATCH_R_OFF);
    ws.warm_i = h->wide_fits ? reinterpret_cast<int *>(ptr + SCRATCH_WARM_I_OFF) : nullptr;
    ws.warm_d = h->wide_fits ? reinterpret_cast<double *>(ptr + SCRATCH_WARM_D_OFF) : nullptr;
    ws.block_list = h->block_fits ? reinterpret_cast<int *>(ptr + SCRATCH_BLOCK_OFF) : nullptr;
    ws.block_rows = 1 << 30;
    const int cap = h->dense_in_lds ? 512 : NOALLOC_SLOTS;
    // The scratch's counter block is cleared in front of every call (a 32-byte memset node, no
    // allocation): under stream capture a call is only recorded, so nothing can be remembered
    // about what an earlier call has left there, and the array is shared by every handle and thread.
    bool clean = false;
    return launch_two_stage(h, di, *out, B, ws, B < cap ? B : cap, (hipStream_t)stream, &clean);
}

int mpcb_solve_coupled_multi(const mpcb_handle *handles, int32_t n_handles, int32_t B,
                             const mpcb_coupled_inputs *in, const mpcb_outputs *out) {
    if (!handles || n_handles <= 0 || !in || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative size");
    for (int g = 0; g < n_handles; g++) {
        if (!handles[g]) return fail(MPCB_E_ARG, "null handle in the list");
        for (int g2 = 0; g2 < g; g2++)
            if (handles[g2] == handles[g]) return fail(MPCB_E_ARG, "the same handle is listed twice");
        if (handles[g]->hs.N != handles[0]->hs.N || handles[g]->hs.m != handles[0]->hs.m ||
            handles[g]->hs.nphase != handles[0]->hs.nphase ||
            handles[g]->params.formulation != handles[0]->params.formulation)
            return fail(MPCB_E_ARG, "the handles were created with different shapes");
    }
    const size_t N = (size_t)handles[0]->hs.N, m = (size_t)handles[0]->hs.m, P = (size_t)handles[0]->hs.nphase;
    const size_t as = out->active_stride > 0 ? (size_t)out->active_stride : 0;
    std::vector<int> rcs(n_handles, 0);
    std::vector<std::string> msgs(n_handles);
    auto work = [&](int g) {
        // contiguous shard: instance i goes to handle floor(i * n_handles / B)
        const size_t first = ((size_t)g * (size_t)B + n_handles - 1) / n_handles;
        const size_t last = ((size_t)(g + 1) * (size_t)B + n_handles - 1) / n_handles;
        if (last <= first) return;
        mpcb_coupled_inputs si = *in;
        mpcb_outputs so = *out;
#define IN_(field, w) if (si.field) si.field += first * (w);
        IN_(x0, 3) IN_(y0, 3) IN_(speed, 2) IN_(feet_x, 4) IN_(feet_y, 4) IN_(swing0, 1) IN_(com0, 2)
        IN_(rows_pqr, m * 3) IN_(zref_phase, P * 2) IN_(height, 1) IN_(zref_knot, N * 2) IN_(phase_pqr, P * 12)
#undef IN_
#define OUT_(field, w) if (so.field) so.field += first * (w);
        OUT_(jerk_x, N) OUT_(jerk_y, N) OUT_(com_x, N) OUT_(com_y, N) OUT_(com_xd, N) OUT_(com_yd, N)
        OUT_(zmp_x, N) OUT_(zmp_y, N) OUT_(min_margin, N) OUT_(cost, 1) OUT_(status, 1) OUT_(iters, 1)
        OUT_(n_active, 1) OUT_(active_idx, as) OUT_(multipliers, as) OUT_(alg_flops, 1) OUT_(slacks, 2 * N)
#undef OUT_
        rcs[g] = mpcb_solve_coupled(handles[g], (int32_t)(last - first), &si, &so);
        if (rcs[g] < 0) msgs[g] = mpcb_last_error(); // the message is per thread: carry it over
    };
    // shard 0 on the calling thread, the others on their handles' parked threads
    for (int g = 1; g < n_handles; g++) worker_submit(handles[g], [&work, g]() { work(g); });
    work(0);
    for (int g = 1; g < n_handles; g++) worker_wait(handles[g]);
    int bad = 0;
    for (int g = 0; g < n_handles; g++) {
        if (rcs[g] < 0) return fail(rcs[g], "device " + std::to_string(handles[g]->params.device) + ": " + msgs[g]);
        bad += rcs[g];
    }
    return bad;
}

int mpcb_solve_coupled_multi_device(const mpcb_handle *handles, int32_t n_handles, const int32_t *B,
                                    const mpcb_coupled_inputs *in, const mpcb_outputs *out,
                                    void *const *streams) {
    if (!handles || n_handles <= 0 || !B || !in || !out) return fail(MPCB_E_ARG, "null argument or no handles");
    for (int g = 0; g < n_handles; g++) {
        if (!handles[g]) return fail(MPCB_E_ARG, "null handle in the list");
        if (B[g] < 0) return fail(MPCB_E_ARG, "negative shard size");
        for (int g2 = 0; g2 < g; g2++)
            if (handles[g2] == handles[g]) return fail(MPCB_E_ARG, "the same handle is listed twice");
    }
    for (int g = 0; g < n_handles; g++) {
        const int rc = mpcb_solve_coupled_device(handles[g], B[g], &in[g], &out[g], streams ? streams[g] : nullptr);
        if (rc < 0) return fail(rc, "shard " + std::to_string(g) + " (device " +
                                        std::to_string(handles[g]->params.device) + "): " + g_err);
    }
    return 0;
}

int mpcb_solve_coupled_groups(const mpcb_handle *handles, int32_t n_groups, const int32_t *B,
                              const mpcb_coupled_inputs *in, const mpcb_outputs *out, void *stream) {
    if (!handles || n_groups <= 0 || !B || !in || !out) return fail(MPCB_E_ARG, "null argument or no groups");
    for (int g = 0; g < n_groups; g++) {
        if (!handles[g]) return fail(MPCB_E_ARG, "null handle in the list");
        if (B[g] < 0) return fail(MPCB_E_ARG, "negative group size");
        if (handles[g]->params.device != handles[0]->params.device)
            return fail(MPCB_E_ARG, "the groups of one call must live on one device");
        for (int g2 = 0; g2 < g; g2++)
            if (handles[g2] == handles[g]) return fail(MPCB_E_ARG, "the same handle is listed twice");
    }
    const int dev = handles[0]->params.device;
    if (dev >= MAX_DEVICES) return fail(MPCB_E_ARG, "device ordinal beyond the library's table");
    DeviceGuard guard(dev);
    int lanes = 1; // lane 0 is the caller's stream
    if (const char *e = std::getenv("MPCB_GROUP_LANES")) {
        const int want = std::atoi(e);
        lanes = 1 + (want < 0 ? 0 : want > GROUP_LANES ? GROUP_LANES : want);
    }
    if (lanes > n_groups) lanes = n_groups;
    GroupLanes &L = g_lanes[dev];
    if (lanes > 1) {
        std::lock_guard<std::mutex> lock(g_lanes_mutex);
        if (!L.ready) {
            // (what an earlier, failed attempt created is kept and used: nothing is made twice)
            int least = 0, greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            if (!L.fork) HIP_TRY(hipEventCreateWithFlags(&L.fork, hipEventDisableTiming));
            for (int l = 0; l < GROUP_LANES; l++) {
                if (!L.stream[l]) HIP_TRY(hipStreamCreateWithPriority(&L.stream[l], hipStreamNonBlocking, greatest));
                if (!L.join[l]) HIP_TRY(hipEventCreateWithFlags(&L.join[l], hipEventDisableTiming));
            }
            L.ready = true;
        }
    }
    hipStream_t s0 = (hipStream_t)stream;
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(L.fork, s0));
        for (int l = 1; l < lanes; l++) HIP_TRY(hipStreamWaitEvent(L.stream[l - 1], L.fork, 0));
    }
    int rc = 0;
    for (int g = 0; g < n_groups && rc >= 0; g++) {
        const int lane = g % lanes;
        rc = mpcb_solve_coupled_device(handles[g], B[g], &in[g], &out[g], lane == 0 ? s0 : L.stream[lane - 1]);
        if (rc < 0) fail(rc, "group " + std::to_string(g) + ": " + g_err);
    }
    // the join is enqueued even after a failure: what was started must not outlive the caller's order
    for (int l = 1; l < lanes; l++) {
        HIP_TRY(hipEventRecord(L.join[l - 1], L.stream[l - 1]));
        HIP_TRY(hipStreamWaitEvent(s0, L.join[l - 1], 0));
    }
    return rc < 0 ? rc : 0;
}

int mpcb_replan_sweep_device(mpcb_handle h, int32_t B, int32_t n_stages, int32_t use_com_correction,
                             int32_t disturbance_start, const mpcb_replan_state *state,
                             const mpcb_outputs *out, double *stage_cost, int32_t *stage_status,
                             int32_t *stage_nactive, void *stream) {
    if (!h || !state || !out || B < 0 || n_stages < 0) return fail(MPCB_E_ARG, "null argument or negative size");
    const int form = h->params.formulation;
    if (form != MPCB_COUPLED_JERK && form != MPCB_COUPLED_VEL)
        return fail(MPCB_E_ARG, "replanning needs a COUPLED_VEL or COUPLED_JERK handle");
    if (!state->x0 || !state->y0 || !state->speed || !state->feet_x || !state->feet_y || !state->swing0 ||
        !state->com0)
        return fail(MPCB_E_ARG, "missing replanning state array");
    if (!out->jerk_x || !out->jerk_y || !out->cost || !out->status)
        return fail(MPCB_E_ARG, "replanning needs the jerk_x, jerk_y, cost and status outputs");
    if (h->hs.steps < 1) return fail(MPCB_E_ARG, "replanning needs number_of_steps >= 1");
    const int window = h->hs.N / h->hs.steps; // replanningWindow (testPlannerXYconstraintMPC.cpp:95)
    if (window < 1) return fail(MPCB_E_ARG, "horizon shorter than number_of_steps");
    DeviceGuard guard(h->params.device);
    hipStream_t s = (hipStream_t)stream;
    mpcb::ReplanState rs;
    rs.x0 = state->x0;
    rs.y0 = state->y0;
    rs.speed = state->speed;
    rs.feet_x = state->feet_x;
    rs.feet_y = state->feet_y;
    rs.swing0 = state->swing0;
    rs.com0 = state->com0;
    rs.disturbance = state->disturbance;
    HIP_TRY(mpcb::launch_replan_init(rs, B, use_com_correction, s));
    mpcb::DevIn di;
    std::memset(&di, 0, sizeof(di));
    di.x0 = rs.x0;
    di.y0 = rs.y0;
    di.speed = rs.speed;
    di.feet_x = rs.feet_x;
    di.feet_y = rs.feet_y;
    di.swing0 = rs.swing0;
    di.com0 = rs.com0;
    mpcb::ReplanStep sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.window = window;
    sp.N = h->hs.N;
    sp.disturbance_start = disturbance_start;
    sp.use_com_correction = use_com_correction;
    sp.feet_move = h->hs.phase_knots > 0; // with empty phases knot sampleW still shows the old feet
    sp.Ts = h->params.Ts;
    std::memcpy(sp.hip, h->params.hip_offsets, sizeof(sp.hip));
    sp.jerk_x = out->jerk_x;
    sp.jerk_y = out->jerk_y;
    sp.cost = out->cost;
    sp.status = out->status;
    sp.n_active = out->n_active;
    sp.stage_cost = stage_cost;
    sp.stage_status = stage_status;
    sp.stage_nactive = stage_nactive;
    // a sweep is ordered by its own stages only: records of earlier calls are dropped
    if (h->dispatch_hint != MPCB_DISPATCH_IN_ORDER) {
        HIP_TRY(forget_order_records(h, s));
        h->order_calls = 0;
    }
    for (int stage = 0; stage < n_stages; stage++) {
        int rc = solve_two_stage(h, di, *out, B, s);
        if (rc) return rc;
        sp.stage = stage;
        HIP_TRY(mpcb::launch_replan_advance(rs, sp, B, s));
    }
    return 0;
}

int mpcb_solve_box_device(mpcb_handle h, int32_t B, const double *x0, const double *zmin,
                          const double *zmax, const double *height, const mpcb_outputs *out, void *stream) {
    if (!h || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    if (B == 0) return 0;
    if (!x0 || !zmin || !zmax) return fail(MPCB_E_ARG, "null input array");
    if (h->params.formulation != MPCB_BOX_1D && h->params.formulation != MPCB_BOX_SLACK)
        return fail(MPCB_E_ARG, "handle was not created for BOX_1D or BOX_SLACK");
    if ((out->active_idx || out->multipliers) && out->active_stride <= 0)
        return fail(MPCB_E_ARG, "active_stride must be positive");
    DeviceGuard guard(h->params.device);
    mpcb::DevIn di;
    std::memset(&di, 0, sizeof(di));
    di.x0 = x0;
    di.zmin = zmin;
    di.zmax = zmax;
    di.height = height;
    if (height && h->params.formulation == MPCB_BOX_1D) {
        int rc = ensure_height_tables(h);
        if (rc) return rc;
    }
    return solve_two_stage(h, di, *out, B, (hipStream_t)stream);
}

int mpcb_solve_unconstrained_device(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                                    const mpcb_outputs *out, void *stream) {
    if (!h || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    if (B == 0) return 0;
    if (!x0 || !zmp_ref) return fail(MPCB_E_ARG, "null input array");
    if (h->params.formulation != MPCB_UNCONSTRAINED)
        return fail(MPCB_E_ARG, "handle was not created for UNCONSTRAINED");
    DeviceGuard guard(h->params.device);
    HIP_TRY(mpcb::launch_unconstrained(h->ds, x0, zmp_ref, *out, B, (hipStream_t)stream));
    return 0;
}

int mpcb_solve_unconstrained(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                             const mpcb_outputs *out) {
    if (!h || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    DeviceGuard guard(h->params.device);
    if (int rc0 = ensure_own_stream(h)) return rc0;
    Staging st(h);
    const double *dx0, *dz;
    int rc;
    const size_t b = (size_t)B, N = (size_t)h->hs.N;
    mpcb_outputs dout;
    auto stage = [&]() -> int {
        if ((rc = st.in(x0, b * 3, &dx0)) || (rc = st.in(zmp_ref, b * N, &dz))) return rc;
        return stage_outputs(st, *out, dout, b, N);
    };
    if ((rc = stage()) || (rc = st.commit()) || (rc = stage())) return rc;
    if ((rc = mpcb_solve_unconstrained_device(h, B, dx0, dz, &dout, h->own_stream))) return rc;
    if ((rc = fetch_outputs(*out, dout, b, N, h->own_stream))) return rc;
    return 0;
}

constexpr size_t MAX_HEIGHT_HANDLES = 1024;

// COUPLED_REF with a CoM height per instance (host pointers).  The Hessian of this form contains
// Qs Zu' Zu and Zu depends on the height (MPCPlanner.cpp:658, 728-729), so every distinct height
// is a parameter group of its own: the instances are sorted by height, the handles of heights
// not seen before are created together (one chain of table-building launches), the groups are
// solved by one mpcb_solve_coupled_groups call and the results go back in the caller's order.
// Instances sorted by height (`perm`: position -> instance), the distinct heights with their
// counts, and a handle per height: the handle itself where the height is its own, else one of
// h->by_height -- those not seen before are created here, all in one mpcb_create_groups call.
struct HeightGroups {
    std::vector<int> perm;
    std::vector<double> gh;
    std::vector<int32_t> gB;
    std::vector<mpcb_handle> gH;
};

static int group_by_height(mpcb_handle h, int32_t B, const double *height, HeightGroups &G) {
    const size_t b = (size_t)B;
    for (size_t i = 0; i < b; i++)
        if (!(height[i] > 0)) return fail(MPCB_E_ARG, "height must be positive for every instance");
    G.perm.resize(b);
    std::iota(G.perm.begin(), G.perm.end(), 0);
    std::stable_sort(G.perm.begin(), G.perm.end(), [&](int a, int c) { return height[a] < height[c]; });
    for (size_t k = 0; k < b; k++) {
        const double v = height[G.perm[k]];
        if (G.gh.empty() || G.gh.back() != v) {
            G.gh.push_back(v);
            G.gB.push_back(0);
        }
        G.gB.back()++;
    }
    const std::vector<double> &gh = G.gh;
    if (gh.size() > MAX_HEIGHT_HANDLES)
        return fail(MPCB_E_ARG, "more than " + std::to_string(MAX_HEIGHT_HANDLES) + " distinct heights in one batch");
    auto key = [](double v) {
        uint64_t k;
        std::memcpy(&k, &v, sizeof(k));
        return k;
    };
    h->height_calls++;
    // handles of the heights this call needs and the handle does not have yet: built together
    std::vector<mpcb_params> fresh;
    for (double v : gh)
        if (v != h->params.height && !h->by_height.count(key(v))) {
            fresh.push_back(h->params);
            fresh.back().height = v;
        }
    if (h->by_height.size() + fresh.size() > MAX_HEIGHT_HANDLES) {
        // make room: drop the handles that were used longest ago (none of this call's)
        std::vector<std::pair<uint64_t, uint64_t>> age; // (last use, key)
        for (auto &kv : h->by_height) age.emplace_back(h->height_used[kv.first], kv.first);
        std::sort(age.begin(), age.end());
        HIP_TRY(hipDeviceSynchronize());
        for (auto &a : age) {
            if (h->by_height.size() + fresh.size() <= MAX_HEIGHT_HANDLES) break;
            bool needed = false;
            for (double v : gh) needed = needed || key(v) == a.second;
            if (needed) continue;
            mpcb_destroy(h->by_height[a.second]);
            h->by_height.erase(a.second);
            h->height_used.erase(a.second);
        }
    }
    if (!fresh.empty()) {
        std::vector<mpcb_handle> made(fresh.size(), nullptr);
        int rc = mpcb_create_groups(fresh.data(), (int32_t)fresh.size(), made.data());
        if (rc) return rc;
        for (size_t k = 0; k < fresh.size(); k++) h->by_height[key(fresh[k].height)] = made[k];
    }
    G.gH.resize(gh.size());
    for (size_t g = 0; g < gh.size(); g++) {
        G.gH[g] = gh[g] == h->params.height ? h : h->by_height[key(gh[g])];
        h->height_used[key(gh[g])] = h->height_calls;
    }
    return 0;
}

static int solve_coupled_by_height(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in, const mpcb_outputs *out) {
    const size_t b = (size_t)B, N = (size_t)h->hs.N, m = (size_t)h->hs.m, P = (size_t)h->hs.nphase;
    const size_t as = out->active_stride > 0 ? (size_t)out->active_stride : 0;
    HeightGroups G;
    if (int rc0 = group_by_height(h, B, in->height, G)) return rc0;
    const std::vector<int> &perm = G.perm;
    const std::vector<double> &gh = G.gh;
    std::vector<int32_t> &gB = G.gB;
    std::vector<mpcb_handle> &gH = G.gH;
    // inputs in group order
    auto gather = [&](const auto *src, size_t w, auto &dst) {
        if (!src) return;
        dst.resize(b * w);
        for (size_t k = 0; k < b; k++) std::memcpy(&dst[k * w], src + (size_t)perm[k] * w, w * sizeof(*src));
    };
    std::vector<double> x0, y0, speed, feet_x, feet_y, com0, rows_pqr, zref_phase, zref_knot, phase_pqr;
    std::vector<int32_t> swing0;
    gather(in->x0, 3, x0); gather(in->y0, 3, y0); gather(in->speed, 2, speed);
    gather(in->feet_x, 4, feet_x); gather(in->feet_y, 4, feet_y); gather(in->swing0, 1, swing0);
    gather(in->com0, 2, com0); gather(in->rows_pqr, m * 3, rows_pqr); gather(in->zref_phase, P * 2, zref_phase);
    gather(in->zref_knot, N * 2, zref_knot); gather(in->phase_pqr, P * 12, phase_pqr);
    // outputs in group order (host side), scattered back at the end
    std::vector<double> o_d[13];
    std::vector<int32_t> o_i[4];
    mpcb_outputs hp = *out; // host, permuted
    {
        int kd = 0, ki = 0;
#define OD_(field, w) if (out->field) { o_d[kd].resize(b * (w)); hp.field = o_d[kd].data(); } kd++;
#define OI_(field, w) if (out->field) { o_i[ki].resize(b * (w)); hp.field = o_i[ki].data(); } ki++;
        OD_(jerk_x, N) OD_(jerk_y, N) OD_(com_x, N) OD_(com_y, N) OD_(com_xd, N) OD_(com_yd, N) OD_(zmp_x, N)
        OD_(zmp_y, N) OD_(min_margin, N) OD_(cost, 1) OD_(multipliers, as) OD_(alg_flops, 1) OD_(slacks, 2 * N)
        OI_(status, 1) OI_(iters, 1) OI_(n_active, 1) OI_(active_idx, as)
#undef OD_
#undef OI_
    }
    Staging st(h);
    DrainOnExit drain(h->own_stream);
    mpcb_coupled_inputs di;
    std::memset(&di, 0, sizeof(di));
    mpcb_outputs dout;
    int rc;
    auto ptr = [](auto &v) { return v.empty() ? nullptr : v.data(); };
    auto stage = [&]() -> int {
        if ((rc = st.in((const double *)ptr(x0), b * 3, &di.x0)) || (rc = st.in((const double *)ptr(y0), b * 3, &di.y0)) ||
            (rc = st.in((const double *)ptr(speed), b * 2, &di.speed)) ||
            (rc = st.in((const double *)ptr(feet_x), b * 4, &di.feet_x)) ||
            (rc = st.in((const double *)ptr(feet_y), b * 4, &di.feet_y)) ||
            (rc = st.in((const int32_t *)ptr(swing0), b, &di.swing0)) ||
            (rc = st.in((const double *)ptr(com0), b * 2, &di.com0)) ||
            (rc = st.in((const double *)ptr(rows_pqr), b * m * 3, &di.rows_pqr)) ||
            (rc = st.in((const double *)ptr(zref_phase), b * P * 2, &di.zref_phase)) ||
            (rc = st.in((const double *)ptr(zref_knot), b * N * 2, &di.zref_knot)) ||
            (rc = st.in((const double *)ptr(phase_pqr), b * P * 12, &di.phase_pqr)))
            return rc;
        return stage_outputs(st, hp, dout, b, N);
    };
    if ((rc = stage()) || (rc = st.commit()) || (rc = stage())) return rc;
    // per-group views of the staged arrays
    std::vector<mpcb_coupled_inputs> gin(gh.size());
    std::vector<mpcb_outputs> gout(gh.size());
    size_t first = 0;
    for (size_t g = 0; g < gh.size(); g++) {
        mpcb_coupled_inputs si = di;
        mpcb_outputs so = dout;
#define IN_(field, w) if (si.field) si.field += first * (w);
        IN_(x0, 3) IN_(y0, 3) IN_(speed, 2) IN_(feet_x, 4) IN_(feet_y, 4) IN_(swing0, 1) IN_(com0, 2)
        IN_(rows_pqr, m * 3) IN_(zref_phase, P * 2) IN_(zref_knot, N * 2) IN_(phase_pqr, P * 12)
#undef IN_
#define OUT_(field, w) if (so.field) so.field += first * (w);
        OUT_(jerk_x, N) OUT_(jerk_y, N) OUT_(com_x, N) OUT_(com_y, N) OUT_(com_xd, N) OUT_(com_yd, N)
        OUT_(zmp_x, N) OUT_(zmp_y, N) OUT_(min_margin, N) OUT_(cost, 1) OUT_(status, 1) OUT_(iters, 1)
        OUT_(n_active, 1) OUT_(active_idx, as) OUT_(multipliers, as) OUT_(alg_flops, 1) OUT_(slacks, 2 * N)
#undef OUT_
        gin[g] = si;
        gout[g] = so;
        first += (size_t)gB[g];
    }
    if ((rc = mpcb_solve_coupled_groups(gH.data(), (int32_t)gH.size(), gB.data(), gin.data(), gout.data(),
                                        h->own_stream)))
        return rc;
    if ((rc = fetch_outputs(hp, dout, b, N, h->own_stream))) return rc;
    drain.armed = false; // (fetch_outputs has waited)
    auto scatter = [&](auto *dst, const auto *src, size_t w) {
        if (!dst) return;
        for (size_t k = 0; k < b; k++) std::memcpy(dst + (size_t)perm[k] * w, src + k * w, w * sizeof(*dst));
    };
    scatter(out->jerk_x, hp.jerk_x, N); scatter(out->jerk_y, hp.jerk_y, N); scatter(out->com_x, hp.com_x, N);
    scatter(out->com_y, hp.com_y, N); scatter(out->com_xd, hp.com_xd, N); scatter(out->com_yd, hp.com_yd, N);
    scatter(out->zmp_x, hp.zmp_x, N); scatter(out->zmp_y, hp.zmp_y, N); scatter(out->min_margin, hp.min_margin, N);
    scatter(out->cost, hp.cost, 1); scatter(out->status, hp.status, 1); scatter(out->iters, hp.iters, 1);
    scatter(out->n_active, hp.n_active, 1); scatter(out->active_idx, hp.active_idx, as);
    scatter(out->multipliers, hp.multipliers, as); scatter(out->alg_flops, hp.alg_flops, 1);
    scatter(out->slacks, hp.slacks, 2 * N);
    return count_not_ok(out->status, B);
}

int mpcb_solve_coupled(mpcb_handle h, int32_t B, const mpcb_coupled_inputs *in, const mpcb_outputs *out) {
    if (!h || !in || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    DeviceGuard guard(h->params.device);
    if (in->height && h->params.formulation == MPCB_COUPLED_REF && B > 0) {
        if (int rc0 = ensure_own_stream(h)) return rc0;
        if (!in->x0 || !in->y0 || !in->speed) return fail(MPCB_E_ARG, "missing coupled input array");
        return solve_coupled_by_height(h, B, in, out);
    }
    if (int rc0 = ensure_own_stream(h)) return rc0;
    Staging st(h);
    mpcb_coupled_inputs di;
    std::memset(&di, 0, sizeof(di));
    int rc;
    const size_t b = (size_t)B, N = (size_t)h->hs.N;
    mpcb_outputs dout;
    auto stage = [&]() -> int {
        if ((rc = st.in(in->x0, b * 3, &di.x0)) || (rc = st.in(in->y0, b * 3, &di.y0)) ||
            (rc = st.in(in->speed, b * 2, &di.speed)) || (rc = st.in(in->feet_x, b * 4, &di.feet_x)) ||
            (rc = st.in(in->feet_y, b * 4, &di.feet_y)) || (rc = st.in(in->swing0, b, &di.swing0)) ||
            (rc = st.in(in->com0, b * 2, &di.com0)) ||
            (rc = st.in(in->rows_pqr, b * (size_t)h->hs.m * 3, &di.rows_pqr)) ||
            (rc = st.in(in->zref_phase, b * (size_t)h->hs.nphase * 2, &di.zref_phase)) ||
            (rc = st.in(in->height, b, &di.height)) || (rc = st.in(in->zref_knot, b * N * 2, &di.zref_knot)) ||
            (rc = st.in(in->phase_pqr, b * (size_t)h->hs.nphase * 12, &di.phase_pqr)))
            return rc;
        return stage_outputs(st, *out, dout, b, N);
    };
    if ((rc = stage()) || (rc = st.commit()) || (rc = stage())) return rc;
    if ((rc = mpcb_solve_coupled_device(h, B, &di, &dout, h->own_stream))) return rc;
    if ((rc = fetch_outputs(*out, dout, b, N, h->own_stream))) return rc;
    return count_not_ok(out->status, B);
}

int mpcb_solve_box(mpcb_handle h, int32_t B, const double *x0, const double *zmin, const double *zmax,
                   const double *height, const mpcb_outputs *out) {
    if (!h || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    DeviceGuard guard(h->params.device);
    if (int rc0 = ensure_own_stream(h)) return rc0;
    Staging st(h);
    const double *dx0, *dmin, *dmax, *dheight;
    int rc;
    const size_t b = (size_t)B, N = (size_t)h->hs.N;
    mpcb_outputs dout;
    auto stage = [&]() -> int {
        if ((rc = st.in(x0, b * 3, &dx0)) || (rc = st.in(zmin, b * N, &dmin)) || (rc = st.in(zmax, b * N, &dmax)) ||
            (rc = st.in(height, b, &dheight)))
            return rc;
        return stage_outputs(st, *out, dout, b, N);
    };
    if ((rc = stage()) || (rc = st.commit()) || (rc = stage())) return rc;
    if ((rc = mpcb_solve_box_device(h, B, dx0, dmin, dmax, dheight, &dout, h->own_stream))) return rc;
    if ((rc = fetch_outputs(*out, dout, b, N, h->own_stream))) return rc;
    return count_not_ok(out->status, B);
}

int mpcb_solve_unconstrained_heights(mpcb_handle h, int32_t B, const double *x0, const double *zmp_ref,
                                     const double *height, const mpcb_outputs *out) {
    if (!h || !out || B < 0) return fail(MPCB_E_ARG, "null argument or negative batch");
    if (!height) return mpcb_solve_unconstrained(h, B, x0, zmp_ref, out);
    if (B == 0) return 0;
    if (!x0 || !zmp_ref) return fail(MPCB_E_ARG, "null input array");
    if (h->params.formulation != MPCB_UNCONSTRAINED)
        return fail(MPCB_E_ARG, "handle was not created for UNCONSTRAINED");
    DeviceGuard guard(h->params.device);
    if (int rc0 = ensure_own_stream(h)) return rc0;
    const size_t b = (size_t)B, N = (size_t)h->hs.N;
    HeightGroups G;
    if (int rc0 = group_by_height(h, B, height, G)) return rc0;
    // inputs in group order, outputs back in the caller's
    std::vector<double> gx0(b * 3), gz(b * N);
    for (size_t k = 0; k < b; k++) {
        std::memcpy(&gx0[k * 3], x0 + (size_t)G.perm[k] * 3, 3 * sizeof(double));
        std::memcpy(&gz[k * N], zmp_ref + (size_t)G.perm[k] * N, N * sizeof(double));
    }
    // (everything solveQP produces: the kernel writes these outputs and no others)
    std::vector<double> oj, oz, oc, ov;
    std::vector<int32_t> os;
    mpcb_outputs hp = {};
    if (out->jerk_x) { oj.resize(b * N); hp.jerk_x = oj.data(); }
    if (out->zmp_x) { oz.resize(b * N); hp.zmp_x = oz.data(); }
    if (out->com_x) { oc.resize(b * N); hp.com_x = oc.data(); }
    if (out->com_xd) { ov.resize(b * N); hp.com_xd = ov.data(); }
    if (out->status) { os.resize(b); hp.status = os.data(); }
    if (out->iters) std::fill(out->iters, out->iters + b, 0);       // (no iterations, empty working set:
    if (out->n_active) std::fill(out->n_active, out->n_active + b, 0); //  what the kernel reports for solveQP)
    Staging st(h);
    DrainOnExit drain(h->own_stream);
    const double *dx0 = nullptr, *dz = nullptr;
    mpcb_outputs dout;
    int rc;
    auto stage = [&]() -> int {
        if ((rc = st.in((const double *)gx0.data(), b * 3, &dx0)) || (rc = st.in((const double *)gz.data(), b * N, &dz)))
            return rc;
        return stage_outputs(st, hp, dout, b, N);
    };
    if ((rc = stage()) || (rc = st.commit()) || (rc = stage())) return rc;
    size_t first = 0;
    for (size_t g = 0; g < G.gh.size(); g++) {
        mpcb_outputs so = dout;
        if (so.jerk_x) so.jerk_x += first * N;
        if (so.zmp_x) so.zmp_x += first * N;
        if (so.com_x) so.com_x += first * N;
        if (so.com_xd) so.com_xd += first * N;
        if (so.status) so.status += first;
        if ((rc = mpcb_solve_unconstrained_device(G.gH[g], G.gB[g], dx0 + first * 3, dz + first * N, &so, h->own_stream)))
            return rc;
        first += (size_t)G.gB[g];
    }
    if ((rc = fetch_outputs(hp, dout, b, N, h->own_stream))) return rc;
    drain.armed = false; // (fetch_outputs has waited)
    for (size_t k = 0; k < b; k++) {
        const size_t i = (size_t)G.perm[k];
        if (out->jerk_x) std::memcpy(out->jerk_x + i * N, &oj[k * N], N * sizeof(double));
        if (out->zmp_x) std::memcpy(out->zmp_x + i * N, &oz[k * N], N * sizeof(double));
        if (out->com_x) std::memcpy(out->com_x + i * N, &oc[k * N], N * sizeof(double));
        if (out->com_xd) std::memcpy(out->com_xd + i * N, &ov[k * N], N * sizeof(double));
        if (out->status) out->status[i] = os[k];
    }
    return 0;
}

} // extern "C"
