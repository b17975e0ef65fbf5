// wrappers_probe.h — one probe, two subjects (like spaces_probe.h).  A scripted env (reward / done from a tape) under
//     Monitor<ClipRewardEnv<TimeLimit<ScriptEnv>>>      -> per-step reward, done, info["TimeLimit.truncated"], info["episode"]
//     Monitor<TimeLimit<ScriptEnv>> with a file name    -> the CSV file Monitor's ResultWriter produces
//     WrapperUnroll<ScriptEnv, ClipRewardEnv, an ObservationWrapper stage> and an ObservationWrapper2 adapter (run_unroll)
// compiled (a) by oracle/Makefile `_ref` against the REFERENCE's own common/wrapper.h, wrappers/time_limit.h, clip_reward.h and
// monitor.h where they lie (oracle/ref_driver/ref_wrappers_driver.cpp), and (b) by tests/cpp/Makefile against the stand-alone
// mirror gym/wrappers.h (wrappers_mirror_demo.cpp); tests/test_ref_pin.py compares the two outputs byte for byte and against
// the oracle.  Uses only names both trees define; include the subject's headers first.
#pragma once
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <limits>
#include <memory>
#include <optional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace wrappers_probe {

// A scripted env: reward/done come from a tape; observation = steps taken.  Derives from the reference's gym::Env.
struct ScriptEnv : gym::Env<std::vector<double>, int> {
    std::vector<float> rew; std::vector<uint8_t> dn; size_t t = 0;
    ScriptEnv(std::vector<float> r, std::vector<uint8_t> d) : rew(std::move(r)), dn(std::move(d)) {}
    std::vector<double> reset() noexcept override { return {double(t)}; }
    StepT step(const int&) noexcept override {
        size_t i = t % rew.size(); t++;
        return {{double(t)}, rew[i], dn[i] != 0, {}};
    }
};

template <class T> inline bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

inline int run_wrappers(const char* in, const char* out) {
    FILE* f = fopen(in, "rb"); if (!f) return 2;
    int64_t L, max_steps, steps, flags;
    if (!rd(f, &L, 1) || !rd(f, &max_steps, 1) || !rd(f, &steps, 1) || !rd(f, &flags, 1)) return 2;
    std::vector<float> rew(L); std::vector<uint8_t> dn(L);
    if (!rd(f, rew.data(), L) || !rd(f, dn.data(), L)) return 2;
    fclose(f);
    using TL = gym::TimeLimit<ScriptEnv>;
    using CR = gym::ClipRewardEnv<TL>;
    using MO = gym::Monitor<CR, true>;
    auto env = std::make_unique<MO>(std::make_unique<CR>(std::make_unique<TL>(std::make_unique<ScriptEnv>(rew, dn), std::optional<size_t>((size_t)max_steps))));
    FILE* o = fopen(out, "wb"); if (!o) return 2;
    env->reset();
    for (int64_t s = 0; s < steps; s++) {
        auto r = env->step(0);
        float reward = r.reward; uint8_t done = r.done, has_tr = (r.info.count("TimeLimit.truncated") != 0), tr = 0, has_ep = (r.info.count("episode") != 0);
        float ep_r = 0, ep_l = 0;
        if (has_tr) tr = std::any_cast<bool>(r.info["TimeLimit.truncated"]);
        if (has_ep) { auto e = std::any_cast<gym::Result>(r.info["episode"]); ep_r = e.r; ep_l = e.l; }
        fwrite(&reward, 4, 1, o); fwrite(&done, 1, 1, o); fwrite(&has_tr, 1, 1, o); fwrite(&tr, 1, 1, o); fwrite(&has_ep, 1, 1, o);
        fwrite(&ep_r, 4, 1, o); fwrite(&ep_l, 4, 1, o);
        if (r.done) env->reset();   // what SyncVecEnv::stepPerWorker does (sync_env.h:92-100)
    }
    fclose(o);
    return 0;
}

// Same scripted env as run_wrappers, Monitor constructed with id "env7" and a file name: the CSV the reference writes.
inline int run_csv(const char* in, const char* out) {
    FILE* f = fopen(in, "rb"); if (!f) return 2;
    int64_t L, max_steps, steps, flags;
    if (!rd(f, &L, 1) || !rd(f, &max_steps, 1) || !rd(f, &steps, 1) || !rd(f, &flags, 1)) return 2;
    std::vector<float> rew(L); std::vector<uint8_t> dn(L);
    if (!rd(f, rew.data(), L) || !rd(f, dn.data(), L)) return 2;
    fclose(f);
    using TL = gym::TimeLimit<ScriptEnv>;
    using MO = gym::Monitor<TL, true>;
    {
        auto env = std::make_unique<MO>(std::make_unique<TL>(std::make_unique<ScriptEnv>(rew, dn), std::optional<size_t>((size_t)max_steps)),
                                        std::numeric_limits<double>::max(), std::string("env7"), std::filesystem::path(out));
        env->reset();
        for (int64_t s = 0; s < steps; s++) { auto r = env->step(0); if (r.done) env->reset(); }
    }   // the writer's ofstream closes here
    return 0;
}

// --- common/wrapper.h's remaining templates: WrapperUnroll (a by-value chain of transformation-only stages) and ObservationWrapper2 ---
using SpacePtr = decltype(std::declval<ScriptEnv const&>().observationSpace());

struct AffineObs : gym::ObservationWrapper<ScriptEnv> {   // a stage without an inner env: obs -> 2 obs + 1
    AffineObs() : gym::ObservationWrapper<ScriptEnv>(SpacePtr{}, SpacePtr{}) {}
    std::vector<double> observation(std::vector<double>&& x) const noexcept override { for (auto& v : x) v = 2 * v + 1; return std::move(x); }
};
struct ClipStage : gym::ClipRewardEnv<ScriptEnv> {
    ClipStage() : gym::ClipRewardEnv<ScriptEnv>(SpacePtr{}, SpacePtr{}) {}
};
struct HalfObs : gym::ObservationWrapper2<ScriptEnv, float> {   // changes the observation TYPE: vector<double> -> float
    explicit HalfObs(std::shared_ptr<ScriptEnv> e) : gym::ObservationWrapper2<ScriptEnv, float>(SpacePtr{}, std::move(e)) {}
    float observation(std::vector<double>&& x) const noexcept override { return float(0.5 * x[0] - 3.0); }
};

// out: rows of {f64 obs, f32 reward, u8 done, f32 obs2, f32 reward2, u8 done2}; row 0 is the reset; then int64 wrapped()->t, int64 has_spaces,
// then four int64 of the Updater probe: first subscriber's sum and calls, second subscriber's sum and calls
inline int run_unroll(const char* in, const char* out) {
    FILE* f = fopen(in, "rb"); if (!f) return 2;
    int64_t L, max_steps, steps, flags;
    if (!rd(f, &L, 1) || !rd(f, &max_steps, 1) || !rd(f, &steps, 1) || !rd(f, &flags, 1)) return 2;
    std::vector<float> rew(L); std::vector<uint8_t> dn(L);
    if (!rd(f, rew.data(), L) || !rd(f, dn.data(), L)) return 2;
    fclose(f);
    using Chain = gym::WrapperUnroll<ScriptEnv, ClipStage, AffineObs>;
    static_assert(std::is_same_v<gym::lastType<ScriptEnv, ClipStage, AffineObs>, AffineObs> && std::is_same_v<gym::NthType<1, ScriptEnv, ClipStage, AffineObs>, ClipStage>);
    static_assert(std::is_base_of_v<gym::Env<std::vector<double>, int>, Chain>);
    Chain chain(ScriptEnv(rew, dn), ClipStage(), AffineObs());
    auto inner = std::make_shared<ScriptEnv>(rew, dn);
    HalfObs half(inner);
    gym::Env<float, int>& as_env = half;                       // ObservationWrapper2 IS an Env<OutputT, ActionT>
    FILE* o = fopen(out, "wb"); if (!o) return 2;
    auto put = [&](double obs, float r, bool d, float obs2, float r2, bool d2) {
        uint8_t a = d, b = d2;
        fwrite(&obs, 8, 1, o); fwrite(&r, 4, 1, o); fwrite(&a, 1, 1, o); fwrite(&obs2, 4, 1, o); fwrite(&r2, 4, 1, o); fwrite(&b, 1, 1, o);
    };
    put(chain.reset()[0], 0.f, false, as_env.reset(), 0.f, false);
    for (int64_t s = 0; s < steps; s++) {
        auto r = chain.step(0);
        auto q = s % 2 ? as_env.step(0) : half.step(inner->step(0));   // the action form and the "convert a computed result" form
        put(r.observation[0], r.reward, r.done, q.observation, q.reward, q.done);
        if (r.done) chain.reset();
        if (q.done) half.reset(inner->reset());
    }
    int64_t t = (int64_t)half.wrapped()->t, has_spaces = (chain.observationSpace() ? 1 : 0) + (half.actionSpace() ? 2 : 0) + (half.info() == inner->info() ? 4 : 0);
    fwrite(&t, 8, 1, o); fwrite(&has_spaces, 8, 1, o);
    // Updater / Updateable (wrappers/vec_env/updateable.h:10-31): notifyOne tells only the FIRST subscriber
    struct Sum : gym::Updateable<int> { int64_t total = 0, calls = 0; void update(int&& v) override { total += v; calls++; } };
    Sum first, second;
    gym::Updater<int> updater;
    updater.notifyOne(100);                        // nobody subscribed yet: dropped
    updater.subscribe(&first); updater.subscribe(&second);
    for (int v : {5, 7, 11}) updater.notifyOne(std::move(v));
    const int64_t upd[4] = {first.total, first.calls, second.total, second.calls};
    fwrite(upd, 8, 4, o);
    fclose(o);
    return 0;
}

}  // namespace wrappers_probe
