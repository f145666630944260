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
//
// Mixed horizons (BASELINE.json config C5; cfnmpc_multi_create_horizons): the vehicles are bucketed by horizon and dealt
// out over the shards so that sum N_i -- the cost model of a step -- is balanced (cfnmpc_shard_by_horizon below, SURVEY.md
// section 8e "Partitioning"); a shard is then a cfnmpc_fleet (one solver per horizon bucket) over a NON-contiguous index set
// and the host arrays of the whole fleet are gathered / scattered per shard.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "../../include/cfnmpc.h"

namespace {
struct Shard {
    int device = 0, lo = 0, hi = 0;
    cfnmpc_solver* s = nullptr;   // uniform horizon: vehicles [lo, hi)
    cfnmpc_fleet* f = nullptr;    // mixed horizons: vehicles idx[0..count) (ascending), lo = hi = -1
    std::vector<int> idx;
    int Nmax = 0;                 // longest horizon of this shard's fleet
    hipStream_t st = nullptr;
    std::vector<double> h;        // host staging in shard order (mixed)
    std::vector<int> hi_;
};
struct Dev {   // current device for the duration of a call
    int prev = -1;
    explicit Dev(int d) { (void)hipGetDevice(&prev); if (prev != d) (void)hipSetDevice(d); else prev = -1; }
    ~Dev() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// mixed fleets: rows of the whole fleet's host array <-> a shard's staging in shard order (`len` of `fstride` elements per row)
const double* to_shard(Shard& s, const double* src, size_t len, size_t fstride) {
    s.h.resize(s.idx.size() * len);
    for (size_t r = 0; r < s.idx.size(); r++) std::copy_n(src + (size_t)s.idx[r] * fstride, len, s.h.data() + r * len);
    return s.h.data();
}
template <typename T>
void from_shard(const Shard& s, const T* stage, T* dst, size_t len) {
    for (size_t r = 0; r < s.idx.size(); r++) std::copy_n(stage + r * len, len, dst + (size_t)s.idx[r] * len);
}
}  // namespace

struct cfnmpc_multi {
    int B = 0, N = 0;             // N: the common horizon, or the longest one of a mixed fleet (row stride of yref / boxes)
    int Nmin = 0;
    bool mixed = false;
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
    if (opts && opts->struct_size != (int)sizeof(cfnmpc_opts)) return CFNMPC_EINVAL;   // ABI guard (include/cfnmpc.h)
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

// The partitioner of mixed-horizon fleets (pure host code): vehicles in order of decreasing horizon (stable), each to the
// shard with the smallest sum N so far (ties: lowest shard) -- longest-processing-time-first; with a handful of distinct
// horizons every shard ends with its share of every bucket and sum N agrees to within one vehicle's horizon.
int cfnmpc_shard_by_horizon(int batch, const int* N_per_instance, int n_shards, int* shard_of) {
    if (batch < 0 || n_shards < 1 || (batch > 0 && (!N_per_instance || !shard_of))) return CFNMPC_EINVAL;
    for (int i = 0; i < batch; i++) if (N_per_instance[i] < 1) return CFNMPC_EINVAL;
    std::vector<int> order(batch);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return N_per_instance[a] > N_per_instance[b]; });
    std::vector<long long> load(n_shards, 0);
    for (int i : order) {
        const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        shard_of[i] = r;
        load[r] += N_per_instance[i];
    }
    return CFNMPC_OK;
}

int cfnmpc_multi_create_horizons(cfnmpc_multi** out, int n_shards, const int* device_ids, int total_batch, const int* N_per_instance,
                                 const cfnmpc_opts* opts) {
    if (!out || n_shards < 1 || !device_ids || total_batch < n_shards || !N_per_instance) return CFNMPC_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CFNMPC_EHIP;
    for (int i = 0; i < n_shards; i++) if (device_ids[i] < 0 || device_ids[i] >= ndev) return CFNMPC_EINVAL;
    std::vector<int> of(total_batch);
    RC_TRY(cfnmpc_shard_by_horizon(total_batch, N_per_instance, n_shards, of.data()));
    if (opts && opts->struct_size != (int)sizeof(cfnmpc_opts)) return CFNMPC_EINVAL;   // ABI guard (include/cfnmpc.h)
    cfnmpc_opts o;
    if (opts) o = *opts; else cfnmpc_default_opts(&o);
    cfnmpc_multi* m = new cfnmpc_multi;
    m->B = total_batch;
    m->mixed = true;
    m->N = *std::max_element(N_per_instance, N_per_instance + total_batch);
    m->Nmin = *std::min_element(N_per_instance, N_per_instance + total_batch);
    m->sh.resize(n_shards);
    for (int i = 0; i < total_batch; i++) m->sh[of[i]].idx.push_back(i);     // ascending within a shard
    int rc = CFNMPC_OK;
    std::vector<int> hz;
    for (int i = 0; i < n_shards && rc == CFNMPC_OK; i++) {
        Shard& s = m->sh[i];
        s.device = device_ids[i];
        s.lo = s.hi = -1;
        hz.resize(s.idx.size());
        for (size_t r = 0; r < s.idx.size(); r++) hz[r] = N_per_instance[s.idx[r]];
        s.Nmax = *std::max_element(hz.begin(), hz.end());
        Dev d(s.device);
        rc = cfnmpc_fleet_create(&s.f, (int)hz.size(), hz.data(), &o);
        if (rc == CFNMPC_OK && hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) rc = CFNMPC_EHIP;
    }
    if (rc != CFNMPC_OK) { cfnmpc_multi_free(m); return rc; }
    *out = m;
    return CFNMPC_OK;
}

int cfnmpc_multi_shard_fleet(const cfnmpc_multi* m, int shard, cfnmpc_fleet** fleet, int* count, int* index, int* device, void** stream) {
    if (!m || !m->mixed || shard < 0 || shard >= (int)m->sh.size()) return CFNMPC_EINVAL;
    const Shard& s = m->sh[shard];
    if (fleet) *fleet = s.f;
    if (count) *count = (int)s.idx.size();
    if (index) std::copy(s.idx.begin(), s.idx.end(), index);
    if (device) *device = s.device;
    if (stream) *stream = (void*)s.st;
    return CFNMPC_OK;
}

int cfnmpc_multi_free(cfnmpc_multi* m) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) {
        Dev d(s.device);
        if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
        if (s.s) cfnmpc_free(s.s);
        if (s.f) cfnmpc_fleet_free(s.f);
    }
    delete m;
    return CFNMPC_OK;
}

int cfnmpc_multi_batch(const cfnmpc_multi* m) { return m ? m->B : CFNMPC_EINVAL; }
int cfnmpc_multi_num_shards(const cfnmpc_multi* m) { return m ? (int)m->sh.size() : CFNMPC_EINVAL; }

int cfnmpc_multi_shard(const cfnmpc_multi* m, int shard, cfnmpc_solver** solver, int* lo, int* hi, int* device, void** stream) {
    if (!m || m->mixed || shard < 0 || shard >= (int)m->sh.size()) return CFNMPC_EINVAL;   // mixed fleets: cfnmpc_multi_shard_fleet
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
    if (m->mixed) {
        for (Shard& s : m->sh) RC_TRY(cfnmpc_fleet_set_x0(s.f, to_shard(s, x0, 13, 13), CFNMPC_ON_HOST, s.st));
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_set_x0(s.s, x0 + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_set_yref(cfnmpc_multi* m, const double* yref, const double* yref_e) {
    if (!m || !yref || !yref_e) return CFNMPC_EINVAL;
    if (m->mixed) {   // yref [B][Nmax][17] of the whole fleet; a shard's fleet reads the first Nmax_shard rows of each vehicle
        std::vector<double> ye;
        for (Shard& s : m->sh) {
            ye.resize(s.idx.size() * 13);
            for (size_t r = 0; r < s.idx.size(); r++) std::copy_n(yref_e + (size_t)s.idx[r] * 13, 13, ye.data() + r * 13);
            RC_TRY(cfnmpc_fleet_set_yref(s.f, to_shard(s, yref, (size_t)s.Nmax * 17, (size_t)m->N * 17), ye.data(), CFNMPC_ON_HOST, s.st));
        }
        return CFNMPC_OK;
    }
    // (two arrays through ONE staging buffer per shard: the second put is ordered behind the first on the shard's stream)
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_set_yref(s.s, yref + (size_t)s.lo * m->N * 17, yref_e + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_set_box(cfnmpc_multi* m, double u_min, double u_max) {
    if (!m) return CFNMPC_EINVAL;
    if (m->mixed) {
        for (Shard& s : m->sh) RC_TRY(cfnmpc_fleet_set_box(s.f, u_min, u_max));
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh) RC_TRY(cfnmpc_set_box(s.s, u_min, u_max));
    return CFNMPC_OK;
}

int cfnmpc_multi_set_box_stages(cfnmpc_multi* m, const double* lb, const double* ub) {
    if (!m || ((lb == nullptr) != (ub == nullptr))) return CFNMPC_EINVAL;
    if (m->mixed) {   // [B][Nmax][4] of the whole fleet
        std::vector<double> hu;
        for (Shard& s : m->sh) {
            if (!lb) { RC_TRY(cfnmpc_fleet_set_box_stages(s.f, nullptr, nullptr)); continue; }
            const size_t len = (size_t)s.Nmax * 4, fs = (size_t)m->N * 4;
            hu.resize(s.idx.size() * len);
            for (size_t r = 0; r < s.idx.size(); r++) std::copy_n(ub + (size_t)s.idx[r] * fs, len, hu.data() + r * len);
            RC_TRY(cfnmpc_fleet_set_box_stages(s.f, to_shard(s, lb, len, fs), hu.data()));
        }
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh) {
        const size_t off = (size_t)s.lo * m->N * 4;
        RC_TRY_SYNC(m, cfnmpc_set_box_stages(s.s, lb ? lb + off : nullptr, ub ? ub + off : nullptr, CFNMPC_ON_HOST_ASYNC, s.st));
    }
    return sync_all(m);
}

int cfnmpc_multi_set_weights(cfnmpc_multi* m, const double* W, const double* WN) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(m->mixed ? cfnmpc_fleet_set_weights(s.f, W, WN) : cfnmpc_set_weights(s.s, W, WN));
    return CFNMPC_OK;
}

int cfnmpc_multi_init_iterate(cfnmpc_multi* m, int mode) {
    if (!m) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(m->mixed ? cfnmpc_fleet_init_iterate(s.f, mode, s.st) : cfnmpc_init_iterate(s.s, mode, s.st));
    return CFNMPC_OK;
}

int cfnmpc_multi_solve(cfnmpc_multi* m, int n_rti) {
    if (!m || n_rti < 1) return CFNMPC_EINVAL;
    for (Shard& s : m->sh) RC_TRY(m->mixed ? cfnmpc_fleet_solve(s.f, n_rti, s.st) : cfnmpc_solve(s.s, n_rti, s.st));   // asynchronous: every device gets its work before anyone waits
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
    if (m->mixed) {   // (a shard's host getter waits for that shard only; the other shards keep working meanwhile)
        if (stage < 0 || stage >= m->Nmin) return CFNMPC_EINVAL;
        for (Shard& s : m->sh) {
            s.h.resize(s.idx.size() * 4);
            RC_TRY(cfnmpc_fleet_get_u(s.f, stage, s.h.data(), CFNMPC_ON_HOST, s.st));
            from_shard(s, s.h.data(), u, 4);
        }
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_get_u(s.s, stage, u + (size_t)s.lo * 4, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_x(cfnmpc_multi* m, int stage, double* x) {
    if (!m || !x) return CFNMPC_EINVAL;
    if (m->mixed) {
        if (stage < 0 || stage > m->Nmin) return CFNMPC_EINVAL;
        for (Shard& s : m->sh) {
            s.h.resize(s.idx.size() * 13);
            RC_TRY(cfnmpc_fleet_get_x(s.f, stage, s.h.data(), CFNMPC_ON_HOST, s.st));
            from_shard(s, s.h.data(), x, 13);
        }
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh) RC_TRY_SYNC(m, cfnmpc_get_x(s.s, stage, x + (size_t)s.lo * 13, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_cmd(cfnmpc_multi* m, double* cmd_vel, int* motvel) {
    if (!m || !cmd_vel) return CFNMPC_EINVAL;
    if (m->mixed) {
        for (Shard& s : m->sh) {
            s.h.resize(s.idx.size() * 4); s.hi_.resize(s.idx.size() * 4);
            RC_TRY(cfnmpc_fleet_get_cmd(s.f, s.h.data(), motvel ? s.hi_.data() : nullptr, CFNMPC_ON_HOST, s.st));
            from_shard(s, s.h.data(), cmd_vel, 4);
            if (motvel) from_shard(s, s.hi_.data(), motvel, 4);
        }
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_get_cmd(s.s, cmd_vel + (size_t)s.lo * 4, motvel ? motvel + (size_t)s.lo * 4 : nullptr, CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

int cfnmpc_multi_get_stats(cfnmpc_multi* m, int* status, int* qp_iter, double* res) {
    if (!m) return CFNMPC_EINVAL;
    if (m->mixed) {
        for (Shard& s : m->sh) {
            const size_t n = s.idx.size();
            s.h.resize(n); s.hi_.resize(2 * n);
            RC_TRY(cfnmpc_fleet_get_stats(s.f, s.hi_.data(), s.hi_.data() + n, s.h.data(), CFNMPC_ON_HOST, s.st));
            if (status) from_shard(s, s.hi_.data(), status, 1);
            if (qp_iter) from_shard(s, s.hi_.data() + n, qp_iter, 1);
            if (res) from_shard(s, s.h.data(), res, 1);
        }
        return CFNMPC_OK;
    }
    for (Shard& s : m->sh)
        RC_TRY_SYNC(m, cfnmpc_get_stats(s.s, status ? status + s.lo : nullptr, qp_iter ? qp_iter + s.lo : nullptr, res ? res + s.lo : nullptr,
                                CFNMPC_ON_HOST_ASYNC, s.st));
    return sync_all(m);
}

}  // extern "C"
