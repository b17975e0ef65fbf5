// hip_env.h — the single-env face: gym::HipCartPoleEnv / gym::HipMountainCarEnv : gym::Env<std::vector<double>, int>, the
// types and call shapes of the reference's CartPoleEnv / MountainCarEnv (classic_control/cartpole.h:14-26, mountain_car.h;
// env.h:34-85), backed by a ONE-env gymhip engine in lockstep arithmetic.  It exists for API completeness — code written
// against gym::Env (the reference's own per-env wrappers, e.g. gym::TimeLimit<gym::HipCartPoleEnv>, compile over it inside the
// reference tree) — not for speed: every call synchronises.  Batches belong in gym::HipVecEnv.
//
//   reset()  -> the observation = the env's double state (cartpole.cpp:19-31: four uniform_real_distribution<float>(-0.05, 0.05)
//               draws of the env's std::mt19937; mountain_car.cpp:38-42: (-0.5, 0))
//   step(a)  -> StepResponse{observation (double state after the step, the TERMINAL state when done: no auto-reset at this
//               level, sync_env.h:92 does that), reward, done, info {}}
//   seed(s)  -> Env::seed (env.h:60-68): the generator restarts from s; never seeded = default-constructed mt19937
// Deviation: the reference's CartPoleEnv keeps integrating when step() is called after done (and warns, cartpole.cpp:77-82);
// here a finished env stays at its terminal state until reset() (reward 0, done true, the same warning on stderr).
// Plain HIP runtime + the C ABI; no libtorch.  Inside the reference tree define GYMHIP_USE_REFERENCE_HEADERS and include env.h first.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "spaces/box.h"
#include "spaces/discrete.h"
#else
#include "gym/vec_env.h"
#endif
#include "gymhip.h"

namespace gym {

template <int KIND>
class HipClassicEnv : public Env<std::vector<double>, int> {
public:
    static constexpr int F = KIND == GYMHIP_ENV_CARTPOLE ? 4 : 2;
    explicit HipClassicEnv(Kwargs const& kwargs = std::nullopt, int device = 0) : Env<std::vector<double>, int>(kwargs), device(device) {
        if constexpr (KIND == GYMHIP_ENV_CARTPOLE) {   // cartpole.cpp:10-16
            const float fm = std::numeric_limits<float>::max(), x2 = 2.4f * 2, th2 = float(12 * 2 * 3.14159265358979323846 / 360) * 2;
            m_ObservationSpace = std::make_shared<Box<float>>(std::vector<Range<float>>{{-x2, x2}, {-fm, fm}, {-th2, th2}, {-fm, fm}}, Shape{{4}, c10::kFloat});
            m_ActionSpace = std::make_shared<Discrete>(2);
        } else {                                       // mountain_car.cpp:11-12
            m_ObservationSpace = std::make_shared<Box<float>>(Shape{{2}, c10::kFloat});
            m_ActionSpace = std::make_shared<Discrete>(3);
        }
    }
    ~HipClassicEnv() override {
        if (h) gymhip_destroy(h);
        if (dev_action) (void)hipFree(dev_action);
        if (dev_state) (void)hipFree(dev_state);
    }
    HipClassicEnv(HipClassicEnv const&) = delete;
    HipClassicEnv& operator=(HipClassicEnv const&) = delete;

    void seed(std::optional<uint64_t> const& _seed) noexcept override {
        Env<std::vector<double>, int>::seed(_seed);   // keeps m_Device in step with the reference's member (env.h:60-68)
        if (!_seed) { std::fprintf(stderr, "HipClassicEnv::seed(nullopt): the engine needs a reproducible seed; keeping the current stream\n"); return; }
        seeded = true; seed_value = *_seed;
        if (h && KIND == GYMHIP_ENV_CARTPOLE) ok(gymhip_seed(h, seed_value), "gymhip_seed");   // re-seeds the env's device-resident std::mt19937 (env.h:60-63)
    }

    std::vector<double> reset() noexcept override {
        if (!ensure()) return {};
        if (!ok(did_reset ? gymhip_reset_index(h, 0, nullptr) : gymhip_reset(h, nullptr), "gymhip_reset")) return {};
        did_reset = true; finished = false;
        return state();
    }

    StepT step(int const& action) noexcept override {
        if (!did_reset) { std::fprintf(stderr, "HipClassicEnv::step before reset()\n"); return {}; }
        if (finished)   // cartpole.cpp:77-82
            std::fprintf(stderr, "You are calling 'step()' even though this environment has already returned done = True. "
                                 "You should always call 'reset()' once you receive 'done = True'\n");
        const int32_t a = action;
        if (hipMemcpy(dev_action, &a, 4, hipMemcpyHostToDevice) != hipSuccess) return {};
        if (!ok(gymhip_step(h, dev_action, GYMHIP_ACT_I32, nullptr), "gymhip_step")) return {};
        float r = 0.f, d = 0.f;
        if (hipMemcpy(&r, buf.reward, 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&d, buf.done, 4, hipMemcpyDeviceToHost) != hipSuccess) return {};
        finished = d != 0.f;
        return {state(), r, finished, {}};
    }

    std::string info() override { return KIND == GYMHIP_ENV_CARTPOLE ? "HipCartPoleEnv" : "HipMountainCarEnv"; }

private:
    bool ok(int rc, const char* what) const noexcept {
        if (rc != GYMHIP_OK) std::fprintf(stderr, "HipClassicEnv: %s failed (%d): %s\n", what, rc, gymhip_last_error());
        return rc == GYMHIP_OK;
    }
    std::vector<double> state() noexcept {
        std::vector<double> s(F, 0.0);
        if (!ok(gymhip_read_state_f64(h, dev_state, nullptr), "gymhip_read_state_f64")) return {};
        if (hipMemcpy(s.data(), dev_state, F * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return {};
        return s;
    }
    bool ensure() noexcept {
        if (h) return true;
        if (hipSetDevice(device) != hipSuccess) return false;
        gymhip_config c{};
        c.struct_size = sizeof c; c.env_kind = KIND; c.precision = GYMHIP_PRECISION_LOCKSTEP_F64;
        c.reset_mode = KIND == GYMHIP_ENV_CARTPOLE ? GYMHIP_RESET_MT19937 : GYMHIP_RESET_COUNTER_RNG;   // the env's own std::mt19937 (env.h:83), on the device
        c.wrappers = GYMHIP_WRAP_NO_AUTO_RESET;   // the bare env: no TimeLimit / Monitor / auto-reset
        c.n_envs = 1; c.seed = seeded ? seed_value : 0;
        c.clip_obs = 10.f; c.clip_reward = 10.f; c.gamma = 0.99f; c.epsilon = 1e-8f;
        if (!ok(gymhip_create(&c, device, nullptr, 0, &h), "gymhip_create")) { h = nullptr; return false; }
        auto giveUp = [this]() {
            gymhip_destroy(h); h = nullptr;
            if (dev_action) { (void)hipFree(dev_action); dev_action = nullptr; }
            if (dev_state) { (void)hipFree(dev_state); dev_state = nullptr; }
            return false;
        };
        if (KIND == GYMHIP_ENV_CARTPOLE && seeded && !ok(gymhip_seed(h, seed_value), "gymhip_seed")) return giveUp();   // never seeded = default-constructed generator
        if (!ok(gymhip_get_buffers(h, &buf), "gymhip_get_buffers")) return giveUp();
        if (hipMalloc((void**)&dev_action, 4) != hipSuccess || hipMalloc((void**)&dev_state, F * sizeof(double)) != hipSuccess) return giveUp();
        return true;
    }

    int device;
    gymhip_engine* h = nullptr;
    gymhip_buffers buf{};
    int32_t* dev_action = nullptr;
    double* dev_state = nullptr;
    bool seeded = false, did_reset = false, finished = false;
    uint64_t seed_value = 0;
};

using HipCartPoleEnv = HipClassicEnv<GYMHIP_ENV_CARTPOLE>;
using HipMountainCarEnv = HipClassicEnv<GYMHIP_ENV_MOUNTAINCAR>;

}  // namespace gym
