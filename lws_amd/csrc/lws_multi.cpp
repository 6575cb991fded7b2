// lws_multi.cpp -- one node, several GPUs, no torch: the multi-device entry points of include/lws_hip.h.
//
// Spectrograms are independent problems (no cross-spectrogram term anywhere in lwslib.cpp), so a batch is dealt in
// contiguous blocks to one plan per device, each block driven by its own host thread through the single-device entry
// points (each of which uses its own device, stream and scratch): SURVEY.md 8(e) "one host thread + stream per device".
// Nothing is exchanged between devices during the sweeps; the optional residual pair is summed on the host.  The
// Python layer uses one process per GPU instead (bench.py, lws_amd/dist.py); this is what a mex gateway or a C++
// caller uses.
#include "../../include/lws_hip.h"

#include <string>
#include <thread>
#include <vector>

#include "lws_common.h"

struct lws_multi_plan {
    std::vector<lws_plan *> plans;
    std::vector<int> devices;
    int F = 0;
};

void lws_plan_set_host_threads(lws_plan *p, int n);   // lws_capi.hip (internal)

namespace {

// contiguous block [lo, hi) of n items owned by shard i of k; blocks differ by at most one item (lws_amd/dist.py)
void shard_range(int n, int i, int k, int &lo, int &hi) {
    const int base = n / k, extra = n % k;
    lo = i * base + (i < extra ? i : extra);
    hi = lo + base + (i < extra ? 1 : 0);
}

// runs fn(shard, lo, hi) for every non-empty shard on its own thread; first failure (lowest shard) is reported
template <typename Fn> int for_each_shard(lws_multi_plan *mp, int B, Fn fn) {
    const int k = (int)mp->plans.size();
    std::vector<int> rc(k, LWS_OK);
    std::vector<std::string> msg(k);
    std::vector<std::thread> th;
    for (int i = 0; i < k; ++i) {
        int lo, hi;
        shard_range(B, i, k, lo, hi);
        if (hi <= lo) continue;
        th.emplace_back([&, i, lo, hi]() {
            rc[i] = fn(i, lo, hi);
            if (rc[i] != LWS_OK) msg[i] = lws_last_error();   // the error text is per thread: carry it over
        });
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < k; ++i)
        if (rc[i] != LWS_OK) return lws::set_error(rc[i], "shard %d (device %d): %s", i, mp->devices[i], msg[i].c_str());
    return LWS_OK;
}

}  // namespace

extern "C" {

int lws_multi_plan_create(lws_multi_plan **out, int ndev, const int *devices, int F, int L, int Q, int Qp, const double *W,
                          const double *W_ai, const double *W_af, unsigned flags) {
    if (!out) return lws::set_error(LWS_ERR_INVALID, "null plan pointer");
    *out = nullptr;
    const int visible = lws_device_count();
    if (ndev <= 0) { ndev = visible; devices = nullptr; }
    if (ndev < 1) return lws::set_error(LWS_ERR_HIP, "no GPU visible");
    lws_multi_plan *mp = new (std::nothrow) lws_multi_plan();
    if (!mp) return lws::set_error(LWS_ERR_NOMEM, "out of host memory");
    mp->F = F;
    for (int i = 0; i < ndev; ++i) {
        const int dev = devices ? devices[i] : i;
        lws_plan *p = nullptr;
        const int rc = lws_plan_create(&p, dev, F, L, Q, Qp, W, W_ai, W_af, flags);
        if (rc != LWS_OK) {   // (lws_last_error() already holds the reason)
            lws_multi_plan_destroy(mp);
            return rc;
        }
        mp->plans.push_back(p);
        mp->devices.push_back(dev);
    }
    // the shards run side by side, each with its own conversion threads (host-array entry points): an even share of the CPUs this
    // process can use per device, not 32 each
    for (lws_plan *p : mp->plans) lws_plan_set_host_threads(p, std::max(2, lws::usable_cpus() / ndev));
    *out = mp;
    return LWS_OK;
}

void lws_multi_plan_destroy(lws_multi_plan *mp) {
    if (!mp) return;
    for (lws_plan *p : mp->plans) lws_plan_destroy(p);
    delete mp;
}

int lws_multi_plan_shards(const lws_multi_plan *mp) { return mp ? (int)mp->plans.size() : 0; }

int lws_multi_batch_lws(lws_multi_plan *mp, int wsel, const double *S_in, double *S_out, int B, int T, const double *thresholds,
                        int iters) {
    if (!mp || !S_in || !S_out) return lws::set_error(LWS_ERR_INVALID, "null argument");
    if (B < 0 || T < 1) return lws::set_error(LWS_ERR_INVALID, "need B >= 0 and T >= 1 (got B=%d T=%d)", B, T);
    const size_t per = (size_t)T * mp->F * 2;   // doubles per spectrogram
    return for_each_shard(mp, B, [&](int i, int lo, int hi) {
        return lws_batch_lws(mp->plans[i], wsel, S_in + lo * per, S_out + lo * per, hi - lo, T, thresholds, iters);
    });
}

int lws_multi_run_lws(lws_multi_plan *mp, const double *S_in, double *S_out, int B, int T, const double *thr_nofuture,
                      int it_nofuture, const double *thr_online, int it_online, int LA, double qdiv, const double *thr_batch,
                      int it_batch) {
    if (!mp || !S_in || !S_out) return lws::set_error(LWS_ERR_INVALID, "null argument");
    if (B < 0 || T < 1) return lws::set_error(LWS_ERR_INVALID, "need B >= 0 and T >= 1 (got B=%d T=%d)", B, T);
    const size_t per = (size_t)T * mp->F * 2;
    return for_each_shard(mp, B, [&](int i, int lo, int hi) {
        return lws_run_lws(mp->plans[i], S_in + lo * per, S_out + lo * per, hi - lo, T, thr_nofuture, it_nofuture, thr_online,
                           it_online, LA, qdiv, thr_batch, it_batch);
    });
}

int lws_multi_residual(lws_multi_plan *mp, const double *S, int B, int T, double *out) {
    if (!mp || !S || !out) return lws::set_error(LWS_ERR_INVALID, "null argument");
    if (B < 1 || T < 1) return lws::set_error(LWS_ERR_INVALID, "need B >= 1 and T >= 1");
    const size_t per = (size_t)T * mp->F * 2;
    std::vector<double> pairs((size_t)B * 2, 0.0);
    const int rc = for_each_shard(mp, B, [&](int i, int lo, int hi) {
        return lws_residual(mp->plans[i], S + lo * per, hi - lo, T, pairs.data() + (size_t)lo * 2);
    });
    if (rc != LWS_OK) return rc;
    out[0] = out[1] = 0.0;
    for (int b = 0; b < B; ++b) { out[0] += pairs[2 * b]; out[1] += pairs[2 * b + 1]; }   // fixed order: deterministic
    return LWS_OK;
}

}  // extern "C"
