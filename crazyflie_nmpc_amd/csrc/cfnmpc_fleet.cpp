// Mixed-horizon fleets behind the C-ABI (include/cfnmpc.h, cfnmpc_fleet_*).
//
// The reference fixes its horizon when the solver is generated (generate_c_code.py:41-42) and
// owns one vehicle per process; BASELINE.json's config C5 runs N in {30, 50, 100} side by side.
// A cfnmpc_solver's workspace is blocked by stage for ONE horizon, so a fleet buckets its
// vehicles by N: one solver per distinct horizon, addressed through index lists.  Fleet-level
// arrays keep the caller's vehicle order; rows move between that order and the buckets with
// the row gather / scatter kernel below (device pointers) or on the host (host pointers).
// Buckets are solved concurrently, each on its own stream forked from the caller's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <vector>

#include "../../include/cfnmpc.h"

namespace {

// dst bucket row r  <-  src fleet row idx[r]   (GATHER)
// dst fleet row idx[r]  <-  src bucket row r   (!GATHER)
// `len` elements of a row are moved; fleet rows are `fstride` elements apart, bucket rows `len`.
template <typename T, bool GATHER>
__global__ void k_rows(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, int count, int len,
                       long fstride) {
    const int r = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count || e >= len) return;
    const long f = (long)idx[r] * fstride + e, b = (long)r * len + e;
    if (GATHER) dst[b] = src[f];
    else dst[f] = src[b];
}

template <typename T, bool GATHER>
void rows(const T* src, T* dst, const int* idx, int count, int len, long fstride, hipStream_t st) {
    const int bx = len >= 256 ? 256 : 64;
    // blockIdx.y is limited to 65535: walk the bucket in slabs
    for (int r0 = 0; r0 < count; r0 += 65535) {
        const int n = std::min(65535, count - r0);
        const T* s = GATHER ? src : src + (long)r0 * len;
        T* d = GATHER ? dst + (long)r0 * len : dst;
        hipLaunchKernelGGL((k_rows<T, GATHER>), dim3((len + bx - 1) / bx, n), dim3(bx), 0, st, s, d, idx + r0, n, len,
                           fstride);
    }
}

struct Bucket {
    int N = 0, count = 0;
    cfnmpc_solver* s = nullptr;
    std::vector<int> idx;       // fleet index of every bucket row
    int* d_idx = nullptr;
    double* d_rows = nullptr;   // [count][N*17] staging in bucket order (device-pointer calls)
    int* d_ints = nullptr;      // [2][count]
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
};

}  // namespace

struct cfnmpc_fleet {
    int B = 0, Nmin = 0, Nmax = 0, device = 0;
    std::vector<Bucket> bk;      // ascending N (the order of cfnmpc_fleet_bucket)
    std::vector<int> order;      // bucket indices by decreasing work N x vehicles: launch order, stream priorities
    hipEvent_t fork = nullptr;
    std::vector<double> h_rows;  // host staging in bucket order (host-pointer calls)
    std::vector<int> h_ints;
};

#define HIP_TRY(x) do { if ((x) != hipSuccess) return CFNMPC_EHIP; } while (0)

// the fleet lives on the device that was current at cfnmpc_fleet_create (as its solvers do)
struct FleetDevice {
    int prev = -1;
    bool switched = false;
    explicit FleetDevice(const cfnmpc_fleet* f) {
        if (f && hipGetDevice(&prev) == hipSuccess && prev != f->device) switched = hipSetDevice(f->device) == hipSuccess;
    }
    ~FleetDevice() { if (switched) (void)hipSetDevice(prev); }
    FleetDevice(const FleetDevice&) = delete;
    FleetDevice& operator=(const FleetDevice&) = delete;
};
#define RC_TRY(x) do { int rc_ = (x); if (rc_ != CFNMPC_OK) return rc_; } while (0)

namespace {

// run fn(bucket, stream) for every bucket on its own stream, forked from / joined to `user`
template <typename F>
int on_buckets(cfnmpc_fleet* f, hipStream_t user, F fn) {
    HIP_TRY(hipEventRecord(f->fork, user));
    for (int bi : f->order) {   // (heaviest bucket first)
        Bucket& b = f->bk[bi];
        HIP_TRY(hipStreamWaitEvent(b.st, f->fork, 0));
        RC_TRY(fn(b, b.st));
        HIP_TRY(hipEventRecord(b.done, b.st));
        HIP_TRY(hipStreamWaitEvent(user, b.done, 0));
    }
    return CFNMPC_OK;
}

// fleet rows (host) -> bucket order (host staging); returns the staging pointer
const double* host_gather(cfnmpc_fleet* f, const Bucket& b, const double* src, int len, long fstride) {
    f->h_rows.resize((size_t)b.count * len);
    for (int r = 0; r < b.count; r++) std::copy_n(src + (long)b.idx[r] * fstride, len, f->h_rows.data() + (size_t)r * len);
    return f->h_rows.data();
}

}  // namespace

extern "C" {

int cfnmpc_fleet_create(cfnmpc_fleet** out, int batch, const int* N_per_instance, const cfnmpc_opts* opts) {
    if (!out || batch < 1 || !N_per_instance) return CFNMPC_EINVAL;
    *out = nullptr;
    if (opts && opts->struct_size != (int)sizeof(cfnmpc_opts)) return CFNMPC_EINVAL;   // ABI guard (include/cfnmpc.h)
    cfnmpc_opts o;
    if (opts) o = *opts; else cfnmpc_default_opts(&o);
    const int cond_N2_req = o.cond_N2;
    if (cond_N2_req < 0) return CFNMPC_EINVAL;
    std::map<int, std::vector<int>> by_n;
    for (int i = 0; i < batch; i++) {
        if (N_per_instance[i] < 1) return CFNMPC_EINVAL;
        by_n[N_per_instance[i]].push_back(i);
    }
    cfnmpc_fleet* f = new cfnmpc_fleet;
    f->B = batch;
    f->Nmin = by_n.begin()->first;
    f->Nmax = by_n.rbegin()->first;
    int rc = CFNMPC_OK;
    if (hipGetDevice(&f->device) != hipSuccess || hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) != hipSuccess) rc = CFNMPC_EHIP;
    for (auto& kv : by_n) {
        if (rc != CFNMPC_OK) break;
        f->bk.emplace_back();
        Bucket& b = f->bk.back();
        b.N = kv.first;
        b.count = (int)kv.second.size();
        b.idx = std::move(kv.second);
        o.N = b.N;
        // partial condensing applies per bucket: a bucket with no more than cond_N2 stages has nothing to condense
        // (cond_N2 >= N means "none", as for a single solver) instead of failing the whole fleet; a bucket whose
        // blocks would exceed the supported length is still refused by cfnmpc_create
        o.cond_N2 = (cond_N2_req > 0 && cond_N2_req < b.N) ? cond_N2_req : 0;
        rc = cfnmpc_create(&b.s, b.count, &o);
        if (rc != CFNMPC_OK) break;
        if (hipMalloc((void**)&b.d_idx, sizeof(int) * b.count) != hipSuccess ||
            hipMalloc((void**)&b.d_ints, sizeof(int) * 2 * b.count) != hipSuccess) { rc = CFNMPC_ENOMEM; break; }
        if (hipMemcpy(b.d_idx, b.idx.data(), sizeof(int) * b.count, hipMemcpyHostToDevice) != hipSuccess ||
            hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) { rc = CFNMPC_EHIP; break; }
    }
    // Longest job first: the buckets run concurrently on their own streams, and the step lasts as long as the bucket with the
    // most work (N x vehicles: the N = 100 third of config C5 carries 55 % of the stage-steps) -- its launches go out first.
    // (Round 6 also gave the heavy bucket's stream the device's highest PRIORITY and the light one the lowest: +1.3 % in a fresh
    //  process, but -17 % inside bench.py's full run, where a dozen solvers' streams have been created and destroyed before --
    //  9.5 against 11.4 M RTI steps/s, A/B on one box, profiles/r06_notes.md section 7.  Plain streams.)
    if (rc == CFNMPC_OK) {
        f->order.resize(f->bk.size());
        for (size_t i = 0; i < f->bk.size(); i++) f->order[i] = (int)i;
        std::stable_sort(f->order.begin(), f->order.end(), [&](int a, int b) {
            return (long)f->bk[a].N * f->bk[a].count > (long)f->bk[b].N * f->bk[b].count;
        });
        for (Bucket& b : f->bk)
            if (rc == CFNMPC_OK && hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking) != hipSuccess) rc = CFNMPC_EHIP;
    }
    if (rc != CFNMPC_OK) { cfnmpc_fleet_free(f); return rc; }
    *out = f;
    return CFNMPC_OK;
}

int cfnmpc_fleet_free(cfnmpc_fleet* f) {
    if (!f) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    (void)hipDeviceSynchronize();
    for (Bucket& b : f->bk) {
        if (b.s) cfnmpc_free(b.s);
        if (b.d_idx) (void)hipFree(b.d_idx);
        if (b.d_rows) (void)hipFree(b.d_rows);
        if (b.d_ints) (void)hipFree(b.d_ints);
        if (b.st) (void)hipStreamDestroy(b.st);
        if (b.done) (void)hipEventDestroy(b.done);
    }
    if (f->fork) (void)hipEventDestroy(f->fork);
    delete f;
    return CFNMPC_OK;
}

int cfnmpc_fleet_batch(const cfnmpc_fleet* f) { return f ? f->B : CFNMPC_EINVAL; }
int cfnmpc_fleet_max_horizon(const cfnmpc_fleet* f) { return f ? f->Nmax : CFNMPC_EINVAL; }
int cfnmpc_fleet_min_horizon(const cfnmpc_fleet* f) { return f ? f->Nmin : CFNMPC_EINVAL; }
int cfnmpc_fleet_num_buckets(const cfnmpc_fleet* f) { return f ? (int)f->bk.size() : CFNMPC_EINVAL; }

int cfnmpc_fleet_bucket(const cfnmpc_fleet* f, int bucket, int* N, int* count, cfnmpc_solver** solver, int* index) {
    if (!f || bucket < 0 || bucket >= (int)f->bk.size()) return CFNMPC_EINVAL;
    const Bucket& b = f->bk[bucket];
    if (N) *N = b.N;
    if (count) *count = b.count;
    if (solver) *solver = b.s;
    if (index) std::copy(b.idx.begin(), b.idx.end(), index);
    return CFNMPC_OK;
}

unsigned long long cfnmpc_fleet_workspace_bytes(const cfnmpc_fleet* f) {
    unsigned long long t = 0;
    if (f) for (const Bucket& b : f->bk) t += cfnmpc_workspace_bytes(b.s);
    return t;
}

static int staging(Bucket& b) {
    if (b.d_rows) return CFNMPC_OK;
    return hipMalloc((void**)&b.d_rows, sizeof(double) * (size_t)b.count * (b.N * 17 + 13)) == hipSuccess ? CFNMPC_OK : CFNMPC_ENOMEM;
}

int cfnmpc_fleet_set_x0(cfnmpc_fleet* f, const double* x0, int on_device, void* stream) {
    if (!f || !x0) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    if (!on_device) {
        for (Bucket& b : f->bk) RC_TRY(cfnmpc_set_x0(b.s, host_gather(f, b, x0, 13, 13), 0, stream));
        return CFNMPC_OK;
    }
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) {
        RC_TRY(staging(b));
        rows<double, true>(x0, b.d_rows, b.d_idx, b.count, 13, 13, st);
        return cfnmpc_set_x0(b.s, b.d_rows, 1, st);
    });
}

int cfnmpc_fleet_set_yref(cfnmpc_fleet* f, const double* yref, const double* yref_e, int on_device, void* stream) {
    if (!f || !yref || !yref_e) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    const long fs = (long)f->Nmax * 17;
    if (!on_device) {
        std::vector<double> ye;
        for (Bucket& b : f->bk) {
            ye.resize((size_t)b.count * 13);
            for (int r = 0; r < b.count; r++) std::copy_n(yref_e + (long)b.idx[r] * 13, 13, ye.data() + (size_t)r * 13);
            RC_TRY(cfnmpc_set_yref(b.s, host_gather(f, b, yref, b.N * 17, fs), ye.data(), 0, stream));
        }
        return CFNMPC_OK;
    }
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) {
        RC_TRY(staging(b));
        double* e = b.d_rows + (size_t)b.count * b.N * 17;
        rows<double, true>(yref, b.d_rows, b.d_idx, b.count, b.N * 17, fs, st);
        rows<double, true>(yref_e, e, b.d_idx, b.count, 13, 13, st);
        return cfnmpc_set_yref(b.s, b.d_rows, e, 1, st);
    });
}

int cfnmpc_fleet_set_weights(cfnmpc_fleet* f, const double* W, const double* WN) {
    if (!f) return CFNMPC_EINVAL;
    for (Bucket& b : f->bk) RC_TRY(cfnmpc_set_weights(b.s, W, WN));
    return CFNMPC_OK;
}

int cfnmpc_fleet_set_box(cfnmpc_fleet* f, double u_min, double u_max) {
    if (!f || !(u_max > u_min)) return CFNMPC_EINVAL;
    for (Bucket& b : f->bk) RC_TRY(cfnmpc_set_box(b.s, u_min, u_max));
    return CFNMPC_OK;
}

// per-stage / per-input boxes for a fleet: host arrays [B][Nmax][4] in the fleet's vehicle order (rows behind a vehicle's
// own horizon are ignored); NULL, NULL returns every bucket to the scalar box
int cfnmpc_fleet_set_box_stages(cfnmpc_fleet* f, const double* lb, const double* ub) {
    if (!f || ((lb == nullptr) != (ub == nullptr))) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    std::vector<double> hl, hu;
    for (Bucket& b : f->bk) {
        if (!lb) { RC_TRY(cfnmpc_set_box_stages(b.s, nullptr, nullptr, 0, nullptr)); continue; }
        const size_t row = (size_t)b.N * 4, frow = (size_t)f->Nmax * 4;
        hl.resize((size_t)b.count * row); hu.resize((size_t)b.count * row);
        for (int r = 0; r < b.count; r++) {
            std::copy_n(lb + (size_t)b.idx[r] * frow, row, hl.data() + (size_t)r * row);
            std::copy_n(ub + (size_t)b.idx[r] * frow, row, hu.data() + (size_t)r * row);
        }
        RC_TRY(cfnmpc_set_box_stages(b.s, hl.data(), hu.data(), 0, nullptr));
    }
    return CFNMPC_OK;
}

int cfnmpc_fleet_init_iterate(cfnmpc_fleet* f, int mode, void* stream) {
    if (!f) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) { return cfnmpc_init_iterate(b.s, mode, st); });
}

int cfnmpc_fleet_solve(cfnmpc_fleet* f, int n_rti, void* stream) {
    if (!f || n_rti < 1) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) { return cfnmpc_solve(b.s, n_rti, st); });
}

static int fleet_get(cfnmpc_fleet* f, int stage, double* out, int width, int on_device, void* stream,
                     int (*get)(cfnmpc_solver*, int, double*, int, void*)) {
    if (!on_device) {
        for (Bucket& b : f->bk) {
            f->h_rows.resize((size_t)b.count * width);
            RC_TRY(get(b.s, stage, f->h_rows.data(), 0, stream));
            for (int r = 0; r < b.count; r++) std::copy_n(f->h_rows.data() + (size_t)r * width, width, out + (long)b.idx[r] * width);
        }
        return CFNMPC_OK;
    }
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) {
        RC_TRY(staging(b));
        RC_TRY(get(b.s, stage, b.d_rows, 1, st));
        rows<double, false>(b.d_rows, out, b.d_idx, b.count, width, width, st);
        return (int)CFNMPC_OK;
    });
}

int cfnmpc_fleet_get_u(cfnmpc_fleet* f, int stage, double* u, int on_device, void* stream) {
    if (!f || !u || stage < 0 || stage >= f->Nmin) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    return fleet_get(f, stage, u, 4, on_device, stream, cfnmpc_get_u);
}

int cfnmpc_fleet_get_x(cfnmpc_fleet* f, int stage, double* x, int on_device, void* stream) {
    if (!f || !x || stage < 0 || stage > f->Nmin) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    return fleet_get(f, stage, x, 13, on_device, stream, cfnmpc_get_x);
}

int cfnmpc_fleet_get_cmd(cfnmpc_fleet* f, double* cmd_vel, int* motvel, int on_device, void* stream) {
    if (!f || !cmd_vel || f->Nmin < 4) return CFNMPC_EINVAL;   // the output stage reads u1 and x4
    FleetDevice fd(f);
    if (!on_device) {
        for (Bucket& b : f->bk) {
            f->h_rows.resize((size_t)b.count * 4);
            f->h_ints.resize((size_t)b.count * 4);
            RC_TRY(cfnmpc_get_cmd(b.s, f->h_rows.data(), motvel ? f->h_ints.data() : nullptr, 0, stream));
            for (int r = 0; r < b.count; r++) {
                std::copy_n(f->h_rows.data() + (size_t)r * 4, 4, cmd_vel + (long)b.idx[r] * 4);
                if (motvel) std::copy_n(f->h_ints.data() + (size_t)r * 4, 4, motvel + (long)b.idx[r] * 4);
            }
        }
        return CFNMPC_OK;
    }
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) {
        RC_TRY(staging(b));
        int* mv = (int*)(b.d_rows + (size_t)b.count * 4);
        RC_TRY(cfnmpc_get_cmd(b.s, b.d_rows, motvel ? mv : nullptr, 1, st));
        rows<double, false>(b.d_rows, cmd_vel, b.d_idx, b.count, 4, 4, st);
        if (motvel) rows<int, false>(mv, motvel, b.d_idx, b.count, 4, 4, st);
        return (int)CFNMPC_OK;
    });
}

int cfnmpc_fleet_get_stats(cfnmpc_fleet* f, int* status, int* qp_iter, double* res, int on_device, void* stream) {
    if (!f) return CFNMPC_EINVAL;
    FleetDevice fd(f);
    if (!on_device) {
        for (Bucket& b : f->bk) {
            f->h_ints.resize((size_t)2 * b.count);
            f->h_rows.resize(b.count);
            int* hs = f->h_ints.data(), *hi = hs + b.count;
            RC_TRY(cfnmpc_get_stats(b.s, hs, hi, f->h_rows.data(), 0, stream));
            for (int r = 0; r < b.count; r++) {
                if (status) status[b.idx[r]] = hs[r];
                if (qp_iter) qp_iter[b.idx[r]] = hi[r];
                if (res) res[b.idx[r]] = f->h_rows[r];
            }
        }
        return CFNMPC_OK;
    }
    return on_buckets(f, (hipStream_t)stream, [&](Bucket& b, hipStream_t st) {
        RC_TRY(staging(b));
        RC_TRY(cfnmpc_get_stats(b.s, b.d_ints, b.d_ints + b.count, b.d_rows, 1, st));
        if (status) rows<int, false>(b.d_ints, status, b.d_idx, b.count, 1, 1, st);
        if (qp_iter) rows<int, false>(b.d_ints + b.count, qp_iter, b.d_idx, b.count, 1, 1, st);
        if (res) rows<double, false>(b.d_rows, res, b.d_idx, b.count, 1, 1, st);
        return (int)CFNMPC_OK;
    });
}

}  // extern "C"
