// cfnmpc_multi.cpp -- one fleet across several GPUs of a node from ONE process (include/cfnmpc.h,
// cfnmpc_multi_*): the native counterpart of the per-rank sharding that bench.py does with one
// process per GPU.
//
// NMPC instances are independent (one vehicle per solver in the reference: acados_mpc.cpp:76-82), so
// a fleet splits into contiguous shards with NO data-path exchange between devices: shard i owns the
// vehicles [lo_i, hi_i) on device device_ids[i], with its own cfnmpc_solver and its own stream
// created on that device.  cfnmpc_multi_solve launches every shard's RTI step and returns without
// waiting, so all devices work concurrently; the getters wait for the shard they read.  Host arrays
// at this boundary cover the WHOLE fleet in the caller's order.  For device-resident I/O take the
// shard's solver (cfnmpc_multi_shard) and use the single-device API with that device's pointers.
// Device ids may repeat (several shards on one GPU: used by the tests on a one-GPU box).
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/cfnmpc.h"

namespace {
struct Shard {
    int device = 0, lo = 0, hi = 0;
    cfnmpc_solver* s = nullptr;
    hipStream_t st = nullptr;
};
struct Dev {   // current device for the duration of a call
    int prev = -1;
    explicit Dev(int d) { (void)hipGetDevice(&prev); if (prev != d) (void)hipSetDevice(d); else prev = -1; }
    ~Dev() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

struct cfnmpc_multi {
    int B = 0, N = 0;
    std::vector<Shard> sh;
};

#define RC_TRY(x) do { int rc_ = (x); if (rc_ != CFNMPC_OK) return rc_; } while (0)
// for calls that ENQUEUE transfers on the shards' streams (CFNMPC_ON_HOST_ASYNC): on the first failing shard the earlier
// shards' copies may still be reading / writing the caller's arrays -- wait for them before the error is returned
#define RC_TRY_SYNC(m, x) do { int rc_ = (x); if (rc_ != CFNMPC_OK) { (void)sync_all(m); return rc_; } } while (0)

extern "C" {

int cfnmpc_multi_create(cfnmpc_multi** out, int n_shards, const int* device_ids, int total_batch, const cfnmpc_opts* opts) {
    if (!out || n_shards < 1 || !device_ids || total_batch < n_shards) return CFNMPC_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CFNMPC_EHIP;
    for (int i = 0; i < n_shards; i++) if (device_ids[i] < 0 || device_ids[i] >= ndev) return CFNMPC_EINVAL;
    cfnmpc_opts o;
    if (opts) o = *opts; else cfnmpc_default_opts(&o);
    cfnmpc_multi* m = new cfnmpc_multi;
    m->B = total_batch;
    m->N = o.N;
    const int base = total_batch / n_shards, rem = total_batch % n_shards;
    int lo = 0, rc = CFNMPC_OK;
    for (int i = 0; i < n_shards && rc == CFNMPC_OK; i++) {
        Shard s;
        s.device = device_ids[i];
        s.lo = lo;
        s.hi = lo + base + (i < rem ? 1 : 0);
        lo = s.hi;
        Dev d(s.device);
        rc = cfnmpc_create(&s.s, s.hi - s.lo, &o);
        if (rc == CFNMPC_OK && hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) rc = CFNMPC_EHIP;
        m->sh.push_back(s);
    }
    if (rc != CFNMPC_OK) { cfnmpc_multi_free(m); return rc; }
    *out = m;
    return CFNMPC_OK;
}

int cfnmpc_multi_free(cfnmpc_multi* m) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) {
        Dev d(s.device);
        if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
        if (s.s) cfnmpc_free(s.s);
    }
    delete m;
    return CFNMPC_OK;
}

int cfnmpc_multi_batch(const cfnmpc_multi* m) { return m ? m->B : CFNMPC_EINVAL; }
int cfnmpc_multi_num_shards(const cfnmpc_multi* m) { return m ? (int)m->sh.size() : CFNMPC_EINVAL; }

int cfnmpc_multi_shard(const cfnmpc_multi* m, int shard, cfnmpc_solver** solver, int* lo, int* hi, int* device, void** stream) {
    if (!m || shard < 0 || shard >= (int)m->sh.size()) return CFNMPC_EINVAL;
    const Shard& s = m->sh[shard];
    if (solver) *solver = s.s;
    if (lo) *lo = s.lo;
    if (hi) *hi = s.hi;
    if (device) *device = s.device;
    if (stream) *stream = (void*)s.st;
    return CFNMPC_OK;
}

// Host-array I/O of the whole fleet: every shard's transfer (+ layout kernel) is ENQUEUED on its own stream first
// (CFNMPC_ON_HOST_ASYNC), then the shards are waited for -- the copies of different GPUs overlap instead of
// running one after the other.
static int sync_all(cfnmpc_multi* m) {
    for (Shard& s : m->sh) {
        Dev d(s.device);
        if (hipStreamSynchronize(s.st) != hipSuccess) return CFNMPC_EHIP;
    }
    return CFNMPC_OK;
}

int cfnmpc_multi_set_x0(cfnmpc_multi* m, const double* x0) {
    if (!m || !x0) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_set_x0(s.s, x0 + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_set_yref(cfnmpc_multi* m, const double* yref, const double* yref_e) {
    if (!m || !yref || !yref_e) return CFNMPC_EINVAL;
    // (two arrays through ONE staging buffer per shard: the second put is ordered behind the first on the shard's stream)
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_set_yref(s.s, yref + (size_t)s.lo * m->N * 17, yref_e + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_set_box(cfnmpc_multi* m, double u_min, double u_max) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(cfnmpc_set_box(s.s, u_min, u_max));
    return CFNMPC_OK;
}

int cfnmpc_multi_set_box_stages(cfnmpc_multi* m, const double* lb, const double* ub) {
    if (!m || ((lb == nullptr) != (ub == nullptr))) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) {
        const size_t off = (size_t)s.lo * m->N * 4;
        RC_TRY_SYNC(m, cfnmpc_set_box_stages(s.s, lb ? lb + off : nullptr, ub ? ub + off : nullptr, CFNMPC_ON_HOST_ASYNC, s.st));
    }
    return sync_all(m);
}

int cfnmpc_multi_set_weights(cfnmpc_multi* m, const double* W, const double* WN) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(cfnmpc_set_weights(s.s, W, WN));
    return CFNMPC_OK;
}

int cfnmpc_multi_init_iterate(cfnmpc_multi* m, int mode) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(cfnmpc_init_iterate(s.s, mode, s.st));
    return CFNMPC_OK;
}

int cfnmpc_multi_solve(cfnmpc_multi* m, int n_rti) {
    if (!m || n_rti < 1) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(cfnmpc_solve(s.s, n_rti, s.st));   // asynchronous: every device gets its work before anyone waits
    return CFNMPC_OK;
}

int cfnmpc_multi_sync(cfnmpc_multi* m) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) {
        Dev d(s.device);
        if (hipStreamSynchronize(s.st) != hipSuccess) return CFNMPC_EHIP;
    }
    return CFNMPC_OK;
}

int cfnmpc_multi_get_u(cfnmpc_multi* m, int stage, double* u) {
    if (!m || !u) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_get_u(s.s, stage, u + (size_t)s.lo * 4, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_x(cfnmpc_multi* m, int stage, double* x) {
    if (!m || !x) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_get_x(s.s, stage, x + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_cmd(cfnmpc_multi* m, double* cmd_vel, int* motvel) {
    if (!m || !cmd_vel) return CFNMPC_EINVAL;
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_get_cmd(s.s, cmd_vel + (size_t)s.lo * 4, motvel ? motvel + (size_t)s.lo * 4 : nullptr, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_stats(cfnmpc_multi* m, int* status, int* qp_iter, double* res) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_get_stats(s.s, status ? status + s.lo : nullptr, qp_iter ? qp_iter + s.lo : nullptr, res ? res + s.lo : nullptr,
                                CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

}  // extern "C"
