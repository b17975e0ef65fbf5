// hip_vec_env.h — gym::HipVecEnv : gym::VecEnv<false>, the drop-in for
//   SyncVecEnv<Monitor<ClipRewardEnv<TimeLimit<CartPoleEnv|MountainCarEnv>>>, false>  [+ VecNormalize<false,...>]
// (wrappers/vec_env/sync_env.h, vec_normalize.h) on one MI355X.  Header-only; all compute is behind the C ABI in
// include/gymhip.h (libgymhip.so).  libtorch is used only to hand results back as torch::Tensor views of device memory.
//
// Inside the reference tree:   #define GYMHIP_USE_REFERENCE_HEADERS, then #include "wrappers/vec_env/base.h" (VecEnv<false>,
// spaces, AnyMap) and "wrappers/monitor.h" (gym::Result) BEFORE this header (the build container compiles
// exactly that configuration against the reference's headers where they lie).
// Elsewhere the stand-alone mirror in gym/vec_env.h supplies the same names.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "spaces/box.h"        // the reference's own Box<T> / Discrete (its include root must be on the include path)
#include "spaces/discrete.h"
#include "wrappers/vec_env/updateable.h"   // Updateable / Updater
#include "wrappers/vec_env/vec_env_wrapper.h"
#if __has_include("boost/algorithm/string.hpp")
#include "wrappers/vec_env/vec_normalize.h"   // VecNormalizeStateRewardPair (that header needs boost)
#else
namespace gym { template <bool dict> struct VecNormalizeStateRewardPair { infer_type<dict> state; torch::Tensor reward, discounts; }; }   // vec_normalize.h:14-18
#endif
#else
#include "gym/vec_env.h"
#endif
#include "gymhip.h"

namespace gym {

struct HipVecEnvOptions {
    int env_kind = GYMHIP_ENV_CARTPOLE;       // GYMHIP_ENV_CARTPOLE | GYMHIP_ENV_MOUNTAINCAR
    int64_t n_envs = 1;                       // envs on THIS GPU
    int64_t env_offset = 0;                   // global index of local env 0 (sharding: env i is seeded seed + env_offset + i)
    int max_episode_steps = 0;                // TimeLimit (time_limit.h:22-26); 0 = no TimeLimit wrapper
    bool clip_reward = false;                 // ClipRewardEnv
    bool monitor = true;                      // Monitor
    bool auto_reset = true;                   // SyncVecEnv(envs, auto_reset) (sync_env.h:28,44,92): false = a finished env keeps its terminal
                                              // observation and waits for reset(index)
    bool lockstep = true;                     // reference-exact arithmetic (double state); false = float-state throughput mode
    bool reference_resets = true;             // CartPole reset states from the reference's own per-env std::mt19937 streams (env.h:83,
                                              // cartpole.cpp:19-31); false = stateless counter RNG (same distribution, other stream)
    int64_t tape_episodes = 0;                // 0: the generators live ON THE DEVICE (GYMHIP_RESET_MT19937: no refill, no host work).
                                              // > 0: the streams are drawn on the HOST into a reset tape of this many rows per env to start
                                              // with (GYMHIP_RESET_TAPE), extended — same streams, twice the rows — before any env runs out
    bool norm_obs = false, norm_reward = false, training = true;   // VecNormalize<false, training, normReward, normObs>
    float clip_obs = 10.f, clip_rew = 10.f, gamma = 0.99f, epsilon = 1e-8f;   // vec_normalize.h:24-27
    int device = 0;
    void* stream = nullptr;                   // hipStream_t; nullptr = the default stream
    // Sharded runs with the normaliser on: sums `count` doubles at `moments` (device) over all ranks on `stream`
    // (e.g. ncclAllReduce(moments, moments, count, ncclDouble, ncclSum, comm, stream)).  Empty = single GPU.
    std::function<void(double* moments, size_t count, void* stream)> allreduce;
    bool exchange_rows = false;               // what `allreduce` is handed: false = 2F+6 reduced doubles (GYMHIP_EXCHANGE_MOMENTS), true = the step
                                              // kernel's partial rows themselves, 32-64 KB, one launch fewer per step (GYMHIP_EXCHANGE_ROWS)
};

class HipVecEnv : public VecEnv<false>, public Updateable<VecNormalizeStateRewardPair<false>> {
public:
    explicit HipVecEnv(HipVecEnvOptions o) : VecEnv<false>((int)o.n_envs, makeObsSpace(o.env_kind), makeActSpace(o.env_kind)), opt(std::move(o)) {
        t_start = std::chrono::high_resolution_clock::now();
    }
    ~HipVecEnv() override { if (h) gymhip_destroy(h); }
    HipVecEnv(HipVecEnv const&) = delete;
    HipVecEnv& operator=(HipVecEnv const&) = delete;

    static std::unique_ptr<VecEnv<false>> make(HipVecEnvOptions o) { return std::make_unique<HipVecEnv>(std::move(o)); }   // sync_env.h:62-64 idiom

    std::vector<size_t> seed(size_t _seed) noexcept override {   // sync_env.h:74-83
        seeded = true; seed_value = _seed;
        if (h) {   // mid-run re-seed: every env's generator restarts, exactly as Env::seed does (env.h:60-63)
            if (usesTape()) { if (ok(gymhip_seed_reset_tape(h, _seed, tape_rows), "gymhip_seed_reset_tape")) tape_budget = tapeStepsFor(tape_rows); }
            else ok(gymhip_seed(h, _seed), "gymhip_seed");   // device generators re-seeded / counter RNG re-keyed
        }
        std::vector<size_t> s(numEnvs);
        for (int i = 0; i < numEnvs; i++) s[i] = _seed + (size_t)opt.env_offset + (size_t)i;
        return s;
    }

    torch::Tensor reset() noexcept override {   // sync_env.h:66-72 / vec_normalize.h:145-157
        if (!ensure() || !tapeSpend(minEpisodeSteps())) return {};
        if (hasNorm() && opt.allreduce) {   // sharded: batch moments all-reduced between the two halves
            if (!ok(gymhip_reset_begin(h, opt.stream), "gymhip_reset_begin")) return {};
            opt.allreduce(buf.moments, (size_t)buf.moments_len, opt.stream);
            if (!ok(gymhip_reset_finish(h, opt.stream), "gymhip_reset_finish")) return {};
        } else if (!ok(gymhip_reset(h, opt.stream), "gymhip_reset")) return {};
        return opt.norm_obs ? obs_norm : obs;
    }

    void stepAsync(torch::Tensor const& actions) noexcept override {   // sync_env.h:85-87
        // TensorAdapter::decode narrows any tensor to int32 (tensor_adapter.cpp:22-23); int32/int64/uint8 device tensors go straight in
        pending = actions;
    }

    StepT stepWait() noexcept override {   // sync_env.h:114-123 / vec_normalize.h:159-177
        if (!ready || !pending.defined()) { std::fprintf(stderr, "HipVecEnv::stepWait before reset()/stepAsync()\n"); return {}; }
        torch::Tensor a = pending;
        int dt;
        if (a.is_cuda() && a.is_contiguous() && a.scalar_type() == torch::kInt32) dt = GYMHIP_ACT_I32;
        else if (a.is_cuda() && a.is_contiguous() && a.scalar_type() == torch::kInt64) dt = GYMHIP_ACT_I64;
        else if (a.is_cuda() && a.is_contiguous() && a.scalar_type() == torch::kUInt8) dt = GYMHIP_ACT_U8;
        else { a = a.to(torch::Device(torch::kCUDA, opt.device), torch::kInt32).contiguous(); dt = GYMHIP_ACT_I32; }
        if (a.numel() != numEnvs) { std::fprintf(stderr, "HipVecEnv: expected %d actions, got %lld\n", numEnvs, (long long)a.numel()); return {}; }
        if (!tapeSpend(1)) return {};
        if (hasNorm() && opt.allreduce) {
            if (!ok(gymhip_step_begin(h, a.data_ptr(), dt, opt.stream), "gymhip_step_begin")) return {};
            opt.allreduce(buf.moments, (size_t)buf.moments_len, opt.stream);
            if (!ok(gymhip_step_finish(h, opt.stream), "gymhip_step_finish")) return {};
        } else if (!ok(gymhip_step(h, a.data_ptr(), dt, opt.stream), "gymhip_step")) return {};
        pending = torch::Tensor();
        return {opt.norm_obs ? obs_norm : obs, reward.unsqueeze(-1), done.unsqueeze(-1), {}};   // sync_env.h:107-112 shapes
    }

    // info maps for the envs that just finished (monitor.h:86-97, time_limit.h:32), built on demand: N unordered_maps per
    // step are the reference's host bottleneck.  Entry i is empty unless done[i] != 0.  Synchronises the stream.
    AnyMap infoOf(int index) {   // one env's map (synchronises the stream)
        AnyMap m;
        if (!ready || done[index].item<float>() == 0.f) return m;
        auto o = torch::TensorOptions().device(torch::kCUDA, opt.device);
        auto len = torch::empty({numEnvs}, o.dtype(torch::kInt32)), tr = torch::empty({numEnvs}, o.dtype(torch::kUInt8)), ret = torch::empty({numEnvs}, o.dtype(torch::kFloat32));
        if (!ok(gymhip_decode_counters(h, nullptr, len.data_ptr<int32_t>(), tr.data_ptr<uint8_t>(), ret.data_ptr<float>(), nullptr, opt.stream), "gymhip_decode_counters")) return m;
        float t = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - t_start).count();
        if (opt.max_episode_steps > 0) m["TimeLimit.truncated"] = bool(tr[index].item<uint8_t>());
        if (opt.monitor) m["episode"] = Result{ret[index].item<float>(), float(len[index].item<int32_t>()), t};
        return m;
    }
    std::vector<AnyMap> infos() {
        std::vector<AnyMap> out(numEnvs);
        if (!ready) return out;
        auto o = torch::TensorOptions().device(torch::kCUDA, opt.device);
        auto len = torch::empty({numEnvs}, o.dtype(torch::kInt32)), tr = torch::empty({numEnvs}, o.dtype(torch::kUInt8)), ret = torch::empty({numEnvs}, o.dtype(torch::kFloat32));
        if (!ok(gymhip_decode_counters(h, nullptr, len.data_ptr<int32_t>(), tr.data_ptr<uint8_t>(), ret.data_ptr<float>(), nullptr, opt.stream), "gymhip_decode_counters")) return out;
        auto hl = len.cpu(), ht = tr.cpu(), hr = ret.cpu(), hd = done.cpu();
        float t = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - t_start).count();
        for (int i = 0; i < numEnvs; i++) {
            if (hd.data_ptr<float>()[i] == 0.f) continue;
            if (opt.max_episode_steps > 0) out[i]["TimeLimit.truncated"] = bool(ht.data_ptr<uint8_t>()[i]);
            if (opt.monitor) out[i]["episode"] = Result{hr.data_ptr<float>()[i], float(hl.data_ptr<int32_t>()[i]), t};
        }
        return out;
    }

    // Switches the fused normaliser on (what wrapping in VecNormalize<false, isTraining, normReward, normObs> does in the
    // reference); only before the engine exists, i.e. before the first reset().  Used by gym::HipVecNormalize below.
    void configureNormalizer(bool is_training, bool norm_reward, bool norm_obs, float clip_obs, float clip_reward, float gamma, float epsilon) {
        if (h) throw std::runtime_error("HipVecEnv: the normaliser must be configured before the first reset()");
        opt.training = is_training; opt.norm_reward = norm_reward; opt.norm_obs = norm_obs;
        opt.clip_obs = clip_obs; opt.clip_rew = clip_reward; opt.gamma = gamma; opt.epsilon = epsilon;
    }
    // copies the running statistics of another env (the (main, env, ...) constructor of VecNormalize, vec_normalize.h:60-73)
    void copyNormStatsFrom(HipVecEnv& other) {
        if (!hasNorm() || !other.hasNorm() || !ensure() || !other.ensure()) return;
        std::vector<float> s(16);
        if (!ok(gymhip_get_norm_stats(other.h, s.data(), other.opt.stream), "gymhip_get_norm_stats")) return;
        ok(gymhip_set_norm_stats(h, s.data(), opt.stream), "gymhip_set_norm_stats");
    }

    // VecNormalize::save / load (vec_normalize.h:79-108): `returns.pt` and `observation.pt` under `dir`, each the archive
    // RunningMeanStdImpl::save writes — three non-buffer tensors "mean", "m_Var", "m_Count" (running_mean_std.h:28-40) —
    // so files move freely between this class and the reference's VecNormalize.  Synchronise the stream.
    void save(std::string const& dir) {
        if (!hasNorm() || !ensure()) return;
        std::vector<float> s(16);
        if (!ok(gymhip_get_norm_stats(h, s.data(), opt.stream), "gymhip_get_norm_stats")) return;
        const int64_t F = buf.obs_dim;
        auto put = [&](const char* file, torch::Tensor mean, torch::Tensor var, float count) {
            torch::serialize::OutputArchive ar;
            ar.write("mean", mean); ar.write("m_Var", var); ar.write("m_Count", torch::tensor(count, torch::kFloat32));
            ar.save_to(dir + "/" + file);
        };
        if (opt.norm_reward) put("returns.pt", torch::tensor(s[2 * F + 1], torch::kFloat32), torch::tensor(s[2 * F + 2], torch::kFloat32), s[2 * F + 3]);
        if (opt.norm_obs) put("observation.pt", torch::from_blob(s.data(), {F}, torch::kFloat32).clone(), torch::from_blob(s.data() + F, {F}, torch::kFloat32).clone(), s[2 * F]);
    }
    void load(std::string const& dir) {
        if (!hasNorm() || !ensure()) return;
        std::vector<float> s(16);
        if (!ok(gymhip_get_norm_stats(h, s.data(), opt.stream), "gymhip_get_norm_stats")) return;
        const int64_t F = buf.obs_dim;
        auto get = [&](const char* file, float* mean, float* var, float* count, int64_t numel) {
            torch::serialize::InputArchive ar;
            ar.load_from(dir + "/" + file);
            torch::Tensor m, v, c;
            ar.read("mean", m); ar.read("m_Var", v); ar.read("m_Count", c);
            if (m.numel() != numel || v.numel() != numel) throw std::runtime_error(std::string(file) + ": statistics of a different feature shape");
            m = m.to(torch::kCPU, torch::kFloat32).contiguous().reshape({-1}); v = v.to(torch::kCPU, torch::kFloat32).contiguous().reshape({-1});
            std::copy_n(m.data_ptr<float>(), numel, mean); std::copy_n(v.data_ptr<float>(), numel, var);
            *count = c.item<float>();
        };
        if (opt.norm_reward) get("returns.pt", &s[2 * F + 1], &s[2 * F + 2], &s[2 * F + 3], 1);
        if (opt.norm_obs) get("observation.pt", &s[0], &s[F], &s[2 * F], F);
        ok(gymhip_set_norm_stats(h, s.data(), opt.stream), "gymhip_set_norm_stats");
    }

    // Monitor::getEpisodeRewards / getEpisodeLengths in aggregate (monitor.h:104-107): {episodes, sum of returns, sum of
    // lengths} finished so far on this shard (sum over ranks yourself when sharded).  Synchronises the stream.
    std::array<double, 3> episodeStatistics() {
        std::array<double, 3> out{0.0, 0.0, 0.0};
        if (h) ok(gymhip_episode_stats(h, out.data(), opt.stream), "gymhip_episode_stats");
        return out;
    }

    torch::Tensor oldObs() const { return obs; }                         // vec_normalize.h:94
    torch::Tensor normStats() const { return norm_stats; }               // [2F+4]: obs mean, var, count, ret mean, var, count
    gymhip_engine* handle() const { return h; }

    // sync_env.h:128-139 (+ vec_normalize.h:179-189 with the normaliser fused in): one env is reset; returns its row
    torch::Tensor reset(int index) override {
        if (!ready) throw std::runtime_error("HipVecEnv::reset(int index) before reset()");
        if (!tapeSpend(minEpisodeSteps())) throw std::runtime_error("HipVecEnv: reset tape exhausted");
        if (gymhip_reset_index(h, index, opt.stream) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        return (opt.norm_obs ? obs_norm : obs)[index];
    }
    // Updateable::update (vec_normalize.h:191-206): state [T][N][F] and reward, any device; rows are pushed into the statistics
    void update(torch::Tensor const& state, torch::Tensor const& reward) {
        if (!hasNorm() || !ensure()) return;
        auto dev = torch::Device(torch::kCUDA, opt.device);
        torch::Tensor s = state.defined() ? state.to(dev, torch::kFloat32).reshape({-1, (int64_t)buf.obs_dim}).contiguous() : torch::Tensor();
        torch::Tensor r = reward.defined() ? reward.to(dev, torch::kFloat32).reshape({-1}).contiguous() : torch::Tensor();
        ok(gymhip_update_norm_stats(h, s.defined() ? s.data_ptr<float>() : nullptr, s.defined() ? s.size(0) : 0,
                                    r.defined() ? r.data_ptr<float>() : nullptr, r.defined() ? r.size(0) : 0, opt.stream), "gymhip_update_norm_stats");
    }
    // the observer face a learner's Updater<VecNormalizeStateRewardPair<false>> calls (updateable.h:10-31, vec_normalize.h:191-206;
    // `discounts` is ignored there too)
    void update(VecNormalizeStateRewardPair<false>&& x) override { update(x.state, x.reward); }
    // The reference's per-index step is a stub that does not step: it returns the BUFFERED reward / done of env `index` as
    // [1] tensors and its buffered info map (sync_env.h:141-157; the observation line is commented out there — the buffered
    // row is returned here, as that code intended).
    StepT step(int index, torch::Tensor const&) noexcept override {
        if (!ready || index < 0 || index >= numEnvs) { std::fprintf(stderr, "HipVecEnv::step(index): no buffered step for env %d\n", index); return {}; }
        return {(opt.norm_obs ? obs_norm : obs)[index], reward[index].unsqueeze(-1), done[index].unsqueeze(-1), {infoOf(index)}};
    }
    void render() const override {}

private:
    static std::shared_ptr<Space> makeObsSpace(int kind) {
        if (kind == GYMHIP_ENV_CARTPOLE) {   // cartpole.cpp:10-16
            const float fm = std::numeric_limits<float>::max(), x2 = 2.4f * 2, th2 = float(12 * 2 * M_PI / 360) * 2;
            return std::make_shared<Box<float>>(std::vector<Range<float>>{{-x2, x2}, {-fm, fm}, {-th2, th2}, {-fm, fm}}, Shape{{4}, c10::kFloat});
        }
        return std::make_shared<Box<float>>(Shape{{2}, c10::kFloat});   // mountain_car.cpp:11
    }
    static std::shared_ptr<Space> makeActSpace(int kind) { return std::make_shared<Discrete>(kind == GYMHIP_ENV_CARTPOLE ? 2 : 3); }

    bool hasNorm() const { return opt.norm_obs || opt.norm_reward; }
    bool ok(int rc, const char* what) const noexcept {
        if (rc != GYMHIP_OK) std::fprintf(stderr, "HipVecEnv: %s failed (%d): %s\n", what, rc, gymhip_last_error());   // reference style: stderr, no throw (cartpole.cpp:77-82)
        return rc == GYMHIP_OK;
    }
    template <typename T> torch::Tensor view(T* p, std::vector<int64_t> shape, c10::ScalarType dt) const {
        if (!p) return {};
        return torch::from_blob((void*)p, shape, torch::TensorOptions().dtype(dt).device(torch::kCUDA, opt.device));
    }
    bool ensure() noexcept {
        if (ready) return true;
        gymhip_config c{};
        c.struct_size = sizeof c;
        c.env_kind = opt.env_kind;
        c.precision = opt.lockstep ? GYMHIP_PRECISION_LOCKSTEP_F64 : GYMHIP_PRECISION_F32;
        const bool tape = usesTape();
        c.reset_mode = tape ? GYMHIP_RESET_TAPE : (opt.reference_resets ? GYMHIP_RESET_MT19937 : GYMHIP_RESET_COUNTER_RNG);
        auto flag = [](bool on, uint32_t bit) { return on ? bit : 0u; };
        c.wrappers = flag(opt.max_episode_steps > 0, GYMHIP_WRAP_TIME_LIMIT) | flag(opt.clip_reward, GYMHIP_WRAP_CLIP_REWARD) |
                     flag(opt.monitor, GYMHIP_WRAP_MONITOR) | flag(opt.norm_obs, GYMHIP_WRAP_NORM_OBS) |
                     flag(opt.norm_reward, GYMHIP_WRAP_NORM_REWARD) | flag(opt.training, GYMHIP_WRAP_NORM_TRAINING) |
                     flag(!opt.auto_reset, GYMHIP_WRAP_NO_AUTO_RESET);
        c.max_episode_steps = opt.max_episode_steps;
        c.n_envs = opt.n_envs; c.env_offset = opt.env_offset; c.seed = seeded ? seed_value : 0;
        c.clip_obs = opt.clip_obs; c.clip_reward = opt.clip_rew; c.gamma = opt.gamma; c.epsilon = opt.epsilon;
        if (!ok(gymhip_create(&c, opt.device, nullptr, 0, &h), "gymhip_create")) { h = nullptr; return false; }
        // any failure after create leaves NO half-built engine behind: the handle is destroyed and the next call starts over
        auto giveUp = [this]() { gymhip_destroy(h); h = nullptr; return false; };
        if (tape) {
            if (tape_rows <= 0) tape_rows = opt.tape_episodes;
            if (!ok(seeded ? gymhip_seed_reset_tape(h, seed_value, tape_rows) : gymhip_unseeded_reset_tape(h, tape_rows), "reset tape")) return giveUp();
            tape_budget = tapeStepsFor(tape_rows);
        } else if (c.reset_mode == GYMHIP_RESET_MT19937 && seeded) {
            if (!ok(gymhip_seed(h, seed_value), "gymhip_seed")) return giveUp();   // never seeded = default-constructed generators, as in the reference
        }
        if (opt.exchange_rows && hasNorm() && opt.allreduce && !ok(gymhip_set_exchange_form(h, GYMHIP_EXCHANGE_ROWS), "gymhip_set_exchange_form")) return giveUp();
        if (!ok(gymhip_get_buffers(h, &buf), "gymhip_get_buffers")) return giveUp();   // after the form is set: moments / moments_len follow it
        const int64_t n = opt.n_envs, F = buf.obs_dim;
        obs = view(buf.obs, {n, F}, c10::kFloat);
        obs_norm = view(buf.obs_norm, {n, F}, c10::kFloat);
        reward = view(buf.reward, {n}, c10::kFloat);
        done = view(buf.done, {n}, c10::kFloat);
        norm_stats = view(buf.norm_stats, {2 * F + 4}, c10::kFloat);
        if (!obs.defined() || !reward.defined() || !done.defined() || (opt.norm_obs && !obs_norm.defined())) return giveUp();
        ready = true;
        return true;
    }

    // Reset-tape guard.  An env consumes one tape row per finished episode and per reset; no CartPole episode is shorter than
    // min(8, TimeLimit) steps (cartpole.cpp:19-68, enumerated), so with r rows left for the env that consumed most, 8 (r - 1)
    // further steps cannot exhaust the tape.  When that budget is spent: one 4-byte read-back of the largest cursor, and with
    // under a quarter of the rows left the tape is redrawn from the same std::mt19937 streams with twice the rows.
    bool usesTape() const { return opt.reference_resets && opt.tape_episodes > 0 && opt.env_kind == GYMHIP_ENV_CARTPOLE; }
    int64_t minEpisodeSteps() const { return opt.max_episode_steps > 0 ? std::max<int64_t>(1, std::min<int64_t>(8, opt.max_episode_steps)) : 8; }
    int64_t tapeStepsFor(int64_t rows_left) const { return std::max<int64_t>(0, minEpisodeSteps() * (rows_left - 1)); }
    bool tapeSpend(int64_t steps) noexcept {
        if (!usesTape() || !h) return true;
        tape_budget -= steps;
        if (tape_budget > 0) return true;
        int32_t err = 0, used = 0;
        if (!ok(gymhip_poll_error(h, &err, opt.stream), "gymhip_poll_error") || !ok(gymhip_tape_cursor_max(h, &used, opt.stream), "gymhip_tape_cursor_max")) return false;
        if (err & 1) { std::fprintf(stderr, "HipVecEnv: reset tape exhausted — an env replayed old reset states\n"); return false; }
        if (tape_rows - used < std::max<int64_t>(8, tape_rows / 4)) {
            const int64_t want = std::max<int64_t>(2 * tape_rows, (int64_t)used + 16);
            if (!ok(gymhip_extend_reset_tape(h, want), "gymhip_extend_reset_tape")) return false;
            tape_rows = want;
        }
        tape_budget = tapeStepsFor(tape_rows - used);
        return true;
    }

    HipVecEnvOptions opt;
    gymhip_engine* h = nullptr;
    bool ready = false;            // set only once the engine, its tape and the tensor views all exist
    int64_t tape_rows = 0, tape_budget = 0;
    gymhip_buffers buf{};
    torch::Tensor obs, obs_norm, reward, done, norm_stats, pending;
    bool seeded = false;
    size_t seed_value = 0;
    std::chrono::high_resolution_clock::time_point t_start;
};

// gym::VecNormalize<false, isTraining, normReward, normObs> (vec_normalize.h:20-232) as a wrapper CLASS, for source compatibility
// with code that composes  VecNormalize<false, ...>(std::move(venv), clip_obs, clip_reward, gamma, epsilon)  over a vec env:
//     auto venv = gym::HipVecEnv::make(o);                                             // was SyncVecEnv<...>::make(envs)
//     auto env  = std::make_unique<gym::HipTrainingVecNormRO>(std::move(venv));        // was TrainingVecNormRO<false>
// The wrapper computes nothing itself: it switches the fused normaliser of the HipVecEnv it is given on (so it must get the env
// before its first reset(), else std::runtime_error) and forwards; reset / stepWait / reset(index) / step(index, .) come from
// VecEnvWrapper.  Same public surface as the reference class: save / load / oldObs / update (Updateable) / the (main, env, ...)
// constructor that copies a training env's statistics.
template <bool isTraining, bool normReward, bool normObs>
class HipVecNormalize : public VecEnvWrapper<false>, public Updateable<VecNormalizeStateRewardPair<false>> {
public:
    explicit HipVecNormalize(std::unique_ptr<VecEnv<false>> env, float clip_obs = 10.f, float clip_reward = 10.f, float gamma = 0.99f, float epsilon = 1e-8f)
        : VecEnvWrapper<false>(std::move(env)) {
        hip = dynamic_cast<HipVecEnv*>(this->venv.get());
        if (!hip) throw std::invalid_argument("HipVecNormalize fuses into a gym::HipVecEnv; got another VecEnv<false>");
        hip->configureNormalizer(isTraining, normReward, normObs, clip_obs, clip_reward, gamma, epsilon);
    }
    HipVecNormalize(VecEnv<false>* main, std::unique_ptr<VecEnv<false>> env, float clip_obs = 10.f, float clip_reward = 10.f, float gamma = 0.99f,
                    float epsilon = 1e-8f)
        : HipVecNormalize(std::move(env), clip_obs, clip_reward, gamma, epsilon) {
        if (auto* trained = dynamic_cast<HipVecNormalize<true, normReward, normObs>*>(main)) hip->copyNormStatsFrom(*trained->engineEnv());
    }
    void save(std::string const& dir) { hip->save(dir); }                      // vec_normalize.h:79-92
    void load(std::string const& dir) { hip->load(dir); }                      // vec_normalize.h:96-108
    torch::Tensor oldObs() const { return hip->oldObs(); }                     // vec_normalize.h:94
    torch::Tensor get() const { return hip->normStats(); }                     // the statistics both RunningMeanStd modules hold (vec_normalize.h:75-77): [2F+4] floats
    void update(VecNormalizeStateRewardPair<false>&& x) override { hip->update(x.state, x.reward); }   // vec_normalize.h:191-206
    HipVecEnv* engineEnv() { return hip; }

private:
    HipVecEnv* hip = nullptr;
};
using HipTrainingVecNormRO = HipVecNormalize<true, true, true>;               // gym_.h:15-16

// gym_.h:15-16 — TrainingVecNormRO<false> = VecNormalize<false, true, true, true> over a sync vec env
inline std::unique_ptr<VecEnv<false>> makeTrainingVecNormRO(HipVecEnvOptions o) {
    o.norm_obs = o.norm_reward = o.training = true;
    return HipVecEnv::make(std::move(o));
}

}  // namespace gym
