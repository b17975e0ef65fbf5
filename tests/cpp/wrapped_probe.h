// wrapped_probe.h — one loop, two subjects: a per-env wrapper stack (Monitor<TimeLimit<...>> etc.) around the single-env face
// gym::HipCartPoleEnv / gym::HipMountainCarEnv, stepping on the GPU; one line per step (reset on done, as sync_env.h:92-100 does)
// for tests/test_ref_pin.py to compare with the oracle.  Instantiated (a) with the REFERENCE's own wrapper templates by
// oracle/ref_driver/ref_dropin_tu.cpp and (b) with the stand-alone mirror gym/wrappers.h by tests/cpp/hip_env_demo.cpp.
#pragma once
#include <any>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

namespace wrapped_probe {

// The stack behind an ObservationWrapper2 (common/wrapper.h:231-275) that narrows the env's double state to float — what
// TensorAdapter::encode(vector<double>) does on the way into SyncVecEnv's batch (tensor_adapter.cpp:84-87).
template <class Inner> struct FloatObs : gym::ObservationWrapper2<Inner, std::vector<float>> {
    explicit FloatObs(std::shared_ptr<Inner> const& e) : gym::ObservationWrapper2<Inner, std::vector<float>>(e->observationSpace(), e) {}
    std::vector<float> observation(std::vector<double>&& x) const noexcept override { return std::vector<float>(x.begin(), x.end()); }
};

template <class Wrapped> int run_wrapped(std::unique_ptr<Wrapped> env, uint64_t seed, int steps, unsigned n_act) {
    env->seed(seed);
    auto s0 = env->reset();
    std::printf("wreset"); for (double v : s0) std::printf(" %.17g", v); std::printf("\n");
    for (int t = 0; t < steps; t++) {
        const int a = (int)(((unsigned)t * 2654435761u >> 7) % n_act);
        auto r = env->step(a);
        const int has_tr = (r.info.count("TimeLimit.truncated") != 0), tr = has_tr ? (int)std::any_cast<bool>(r.info["TimeLimit.truncated"]) : 0;
        const int has_ep = (r.info.count("episode") != 0);
        float ep_r = 0, ep_l = 0;
        if (has_ep) { auto e = std::any_cast<gym::Result>(r.info["episode"]); ep_r = e.r; ep_l = e.l; }
        std::printf("wstep %d %d %.9g %d %d %d %d %.9g %.9g", t, a, (double)r.reward, (int)r.done, has_tr, tr, has_ep, (double)ep_r, (double)ep_l);
        for (double v : r.observation) std::printf(" %.17g", v);
        std::printf("\n");
        if (r.done) { auto s = env->reset(); std::printf("wreset"); for (double v : s) std::printf(" %.17g", v); std::printf("\n"); }
    }
    return 0;
}

}  // namespace wrapped_probe
