// wrappers.h — stand-alone mirror of the reference's PER-ENV wrapper API, for code that composes single envs the way the
// reference does (e.g. gym::Monitor<gym::TimeLimit<gym::HipCartPoleEnv>>) outside the reference tree:
//     Wrapper / ActionWrapper / RewardWrapper / ObservationWrapper      common/wrapper.h:85-229
//     ObservationWrapper2, WrapperUnroll, lastType / NthType            common/wrapper.h:231-275, 26-81
//     TimeLimit                                                         wrappers/time_limit.h:13-45
//     ClipRewardEnv                                                     wrappers/clip_reward.h:13-26
//     Monitor, ResultWriter, MonitorWithEarlyReset                      wrappers/monitor.h:20-123, gym_.h:18-19
// Same names, template parameters, member signatures and observable behaviour (including what is written to stderr and to the
// CSV file); tests/test_ref_pin.py runs the same scripted env under these templates and under the reference's own (compiled from
// its headers) and compares every output.  Inside the reference tree (GYMHIP_USE_REFERENCE_HEADERS) this header only includes
// the reference's files.
// Batches do not go through these: gym::HipVecEnv fuses TimeLimit / ClipReward / Monitor into the step kernel.
#pragma once
#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "common/wrapper.h"
#include "wrappers/clip_reward.h"
#include "wrappers/monitor.h"
#include "wrappers/time_limit.h"
#else
#include <chrono>
#include <filesystem>
#include <iostream>
#include <limits>
#include <numeric>
#include <optional>
#include <sstream>
#include <tuple>

#include "gym/monitor_csv.h"
#include "gym/vec_env.h"

namespace gym {

template <typename EnvT>
class Wrapper : public Env<typename EnvT::ObservationT, typename EnvT::ActionT, typename EnvT::StepT> {   // wrapper.h:85-151
public:
    using Base = Env<typename EnvT::ObservationT, typename EnvT::ActionT, typename EnvT::StepT>;
    using typename Base::ActionT;
    using typename Base::ObservationT;
    using typename Base::StepT;

    explicit Wrapper(std::unique_ptr<EnvT> env) : m_Env(std::move(env)) {
        this->m_ObservationSpace = m_Env->observationSpace();
        this->m_ActionSpace = m_Env->actionSpace();
    }
    // a wrapper used for its transformation only (no inner env), as the reference's WrapperUnroll stages are
    Wrapper(std::shared_ptr<Space> obsSpace, std::shared_ptr<Space> actSpace) : m_Env(nullptr) {
        this->m_ObservationSpace = std::move(obsSpace);
        this->m_ActionSpace = std::move(actSpace);
    }

    void render() override { m_Env->render(); }
    std::string info() override { return m_Env->info(); }
    void seed(std::optional<uint64_t> const& seed) noexcept override { m_Env->seed(seed); }
    StepT step(ActionT const& action) noexcept override { return m_Env->step(action); }
    ObservationT reset() noexcept override { return m_Env->reset(); }
    // the "apply to an already computed result" forms (wrapper.h:128-135)
    virtual StepT step(StepT&& prevStep) noexcept { return prevStep; }
    virtual ObservationT reset(ObservationT&& x) noexcept { return x; }

    // the first env of type T down the chain, or nullptr (wrapper.h:137-149)
    template <class T> T* try_cast() const noexcept {
        if (auto* hit = dynamic_cast<T*>(m_Env.get())) return hit;
        auto* inner = dynamic_cast<Wrapper*>(m_Env.get());
        return inner ? inner->template try_cast<T>() : nullptr;
    }

protected:
    std::shared_ptr<EnvT> m_Env;
};

template <typename EnvT>
class ActionWrapper : public Wrapper<EnvT> {   // wrapper.h:153-170 (tensor-typed, as in the reference)
public:
    explicit ActionWrapper(std::unique_ptr<EnvT> env) : Wrapper<EnvT>(std::move(env)) {}
    virtual torch::Tensor action(torch::Tensor const& action) const noexcept = 0;
    virtual torch::Tensor reverseAction(torch::Tensor const& action) const noexcept = 0;
    typename Wrapper<EnvT>::StepT step(typename Wrapper<EnvT>::ActionT const& a) noexcept override { return this->m_Env->step(action(a)); }
    typename Wrapper<EnvT>::ObservationT reset() noexcept override { return this->m_Env->reset(); }
};

template <typename EnvT>
class RewardWrapper : public Wrapper<EnvT> {   // wrapper.h:172-194
public:
    explicit RewardWrapper(std::unique_ptr<EnvT> env) : Wrapper<EnvT>(std::move(env)) {}
    RewardWrapper(std::shared_ptr<Space> obsSpace, std::shared_ptr<Space> actSpace) : Wrapper<EnvT>(std::move(obsSpace), std::move(actSpace)) {}
    [[nodiscard]] virtual float reward(float const& reward) const noexcept = 0;
    typename Wrapper<EnvT>::StepT step(typename Wrapper<EnvT>::ActionT const& action) noexcept override {
        auto resp = this->m_Env->step(action);
        resp.reward = reward(resp.reward);
        return resp;
    }
    typename Wrapper<EnvT>::StepT step(typename Wrapper<EnvT>::StepT&& prevStep) noexcept override {
        auto resp = prevStep;
        resp.reward = reward(resp.reward);
        return resp;
    }
};

template <typename EnvT>
class ObservationWrapper : public Wrapper<EnvT> {   // wrapper.h:196-229
public:
    using typename Wrapper<EnvT>::ObservationT;
    using typename Wrapper<EnvT>::StepT;
    explicit ObservationWrapper(std::unique_ptr<EnvT> env) : Wrapper<EnvT>(std::move(env)) {}
    ObservationWrapper(std::shared_ptr<Space> obsSpace, std::shared_ptr<Space> actSpace) : Wrapper<EnvT>(std::move(obsSpace), std::move(actSpace)) {}
    virtual ObservationT observation(ObservationT&&) const noexcept = 0;
    StepT step(typename Wrapper<EnvT>::ActionT const& action) noexcept override {
        auto resp = this->m_Env->step(action);
        resp.observation = observation(std::move(resp.observation));
        return resp;
    }
    ObservationT reset() noexcept override { return observation(this->m_Env->reset()); }
    StepT step(StepT&& resp) noexcept override { return {observation(std::move(resp.observation)), resp.reward, resp.done, resp.info}; }
    ObservationT reset(ObservationT&& x) noexcept override { return observation(std::move(x)); }
};

// wrapper.h:231-275.  An observation wrapper whose OUTPUT type is not the inner env's observation type (FlatObs-style
// adapters: vector<double> -> tensor, image -> vector<float> ...).  The inner env is shared, not owned; wrapped() hands it out.
// The && forms convert a result that was computed elsewhere.
template <typename EnvT, typename OutputT>
class ObservationWrapper2 : public Env<OutputT, typename EnvT::ActionT> {
public:
    ObservationWrapper2(const std::shared_ptr<Space>& obsSpace, std::shared_ptr<EnvT> env) : m_Env(std::move(env)) {
        this->m_ObservationSpace = obsSpace;
        this->m_ActionSpace = m_Env->actionSpace();
    }
    virtual OutputT observation(typename EnvT::ObservationT&&) const noexcept = 0;

    StepResponse<OutputT> step(typename EnvT::ActionT const& action) noexcept override { return step(m_Env->step(action)); }
    StepResponse<OutputT> step(typename EnvT::StepT&& resp) noexcept { return {observation(std::move(resp.observation)), resp.reward, resp.done, resp.info}; }
    OutputT reset() noexcept override { return observation(m_Env->reset()); }
    OutputT reset(typename EnvT::ObservationT&& x) noexcept { return observation(std::move(x)); }

    void render() override { m_Env->render(); }
    std::string info() override { return m_Env->info(); }
    void seed(std::optional<uint64_t> const& seed) noexcept override { m_Env->seed(seed); }
    EnvT* wrapped() { return m_Env.get(); }

protected:
    std::shared_ptr<EnvT> m_Env;
};

template <class... Stages> using lastType = std::tuple_element_t<sizeof...(Stages) - 1, std::tuple<Stages...>>;   // wrapper.h:26-30
template <size_t I, class... Stages> using NthType = std::tuple_element_t<I, std::tuple<Stages...>>;

// wrapper.h:32-81.  A wrapper chain resolved at compile time and held BY VALUE in a tuple: stage 0 is an env (reset(), step(action)),
// every later stage is a wrapper built for its transformation only (the (obsSpace, actSpace) constructors above) whose
// reset(Obs&&) / step(StepT&&) forms are applied in order.  Types and spaces are the last stage's.  render<I>() calls stage I's
// _render(), as in the reference (none of the wrappers here has one; its minigrid envs do).
template <class... Stages>
struct WrapperUnroll : Env<typename lastType<Stages...>::ObservationT, typename lastType<Stages...>::ActionT, typename lastType<Stages...>::StepT> {
    using StepT = typename lastType<Stages...>::StepT;
    using ActionT = typename lastType<Stages...>::ActionT;
    using ObservationT = typename lastType<Stages...>::ObservationT;
    static constexpr size_t kStages = sizeof...(Stages);

    std::tuple<Stages...> wrappers;

    WrapperUnroll(Stages... stages) : wrappers(std::forward<Stages>(stages)...) {
        this->m_ObservationSpace = std::get<kStages - 1>(wrappers).observationSpace();
        this->m_ActionSpace = std::get<kStages - 1>(wrappers).actionSpace();
    }

    template <size_t N> ObservationT wrap(ObservationT&& x) {   // stages N, N+1, ... applied to a reset observation
        if constexpr (N < kStages) return wrap<N + 1>(std::get<N>(wrappers).reset(std::move(x)));
        else return std::move(x);
    }
    template <size_t N> StepT wrap(StepT&& x) {                 // ... and to a step result
        if constexpr (N < kStages) return wrap<N + 1>(std::get<N>(wrappers).step(std::move(x)));
        else return std::move(x);
    }
    template <size_t I> void render() { std::get<I>(wrappers)._render(); }

    ObservationT reset() noexcept override { return wrap<1>(std::get<0>(wrappers).reset()); }
    StepT step(ActionT const& action) noexcept override { return wrap<1>(std::get<0>(wrappers).step(action)); }
};

// time_limit.h:13-45.  The counter runs on the wrapper's own steps; at maxEpisodeSteps the episode ends and
// info["TimeLimit.truncated"] says whether the env itself had NOT finished.  No limit given = SIZE_MAX.
template <class EnvT>
class TimeLimit : public Wrapper<EnvT> {
public:
    static auto make(std::unique_ptr<EnvT> env, std::optional<size_t> const& max_episode_steps) {
        return std::make_unique<TimeLimit<EnvT>>(std::move(env), max_episode_steps);
    }
    TimeLimit(std::unique_ptr<EnvT> env, std::optional<size_t> const& max_episode_steps) : Wrapper<EnvT>(std::move(env)) {
        if (max_episode_steps) maxEpisodeSteps = *max_episode_steps;
    }
    typename EnvT::StepT step(typename EnvT::ActionT const& action) noexcept override {
        auto response = this->m_Env->step(action);
        elapsedSteps++;
        if ((size_t)elapsedSteps >= maxEpisodeSteps) {   // time_limit.h:31 compares int64 with size_t: the same conversion
            response.info["TimeLimit.truncated"] = !response.done;
            response.done = true;
        }
        return response;
    }
    typename EnvT::ObservationT reset() noexcept override {
        elapsedSteps = 0;
        return this->m_Env->reset();
    }

private:
    size_t maxEpisodeSteps{std::numeric_limits<size_t>::max()};
    int64_t elapsedSteps{};
};

template <class EnvT>
struct ClipRewardEnv : public RewardWrapper<EnvT> {   // clip_reward.h:13-26: sign of the reward; a zero (of either sign) passes through
    explicit ClipRewardEnv(std::unique_ptr<EnvT> env) : RewardWrapper<EnvT>(std::move(env)) {}
    ClipRewardEnv(std::shared_ptr<Space> obsSpace, std::shared_ptr<Space> actSpace) : RewardWrapper<EnvT>(std::move(obsSpace), std::move(actSpace)) {}
    float reward(float const& rew) const noexcept final { return rew == 0 ? rew : (rew > 0.f ? 1.f : -1.f); }
};

using ResultWriter = EpisodeCsvWriter;   // monitor.h:20-35 (gym/monitor_csv.h writes the same bytes)

// monitor.h:36-122.  Per-episode return (the float rewards summed in order from 0.f), length and wall time since
// construction -> info["episode"] = Result{r, l, t} on done, optionally appended to `<fileName>.csv`.
template <class EnvType, bool allowEarlyResets = true>
class Monitor : public Wrapper<EnvType> {
public:
    explicit Monitor(std::unique_ptr<EnvType> env, std::optional<double> invalid_returns = std::numeric_limits<double>::max(), std::string const& id = "",
                     std::filesystem::path const& fileName = "")
        : Wrapper<EnvType>(std::move(env)), invalid_returns(invalid_returns) {
        t_start = std::chrono::high_resolution_clock::now();
        if (!fileName.empty()) writer.emplace(fileName, (size_t)t_start.time_since_epoch().count(), id);
    }

    typename EnvType::ObservationT reset() noexcept override { return reset(this->m_Env->reset()); }
    typename EnvType::ObservationT reset(typename EnvType::ObservationT&& x) noexcept override {
        if constexpr (!allowEarlyResets) {
            if (needsReset)   // sic: the reference tests needsReset, not its negation (monitor.h:62-66)
                std::cerr << "Tried to reset an environment before done. If you want to allow early resets, "
                             "wrap your env with Monitor(env, path, allow_early_resets=True)\n";
        }
        reward.clear();
        _steps = 0;
        needsReset = false;
        return x;
    }

    typename EnvType::StepT step(typename EnvType::ActionT const& action) noexcept override { return step(this->m_Env->step(action)); }
    typename EnvType::StepT step(typename EnvType::StepT&& response) noexcept override {
        if (needsReset) std::cerr << "Tried to step environment that needs reset\n";
        reward.push_back(response.reward);
        _steps++;
        if (response.done) {
            needsReset = true;
            episodeReturns.push_back(std::accumulate(reward.begin(), reward.end(), 0.f));
            episodeLengths.push_back(_steps);
            episodeTimes.push_back((float)std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t_start).count());
            const Result epInfo{episodeReturns.back(), _steps, episodeTimes.back()};
            if (invalid_returns && epInfo.r >= *invalid_returns) {
                std::stringstream ss;
                ss << this->info() << "\nInvalid Episode Returns: " << epInfo.r << "\n";
                for (float v : reward) ss << v << " ";
                ss << " Steps: " << epInfo.l;
                std::cerr << ss.str() << "\n";
            }
            if (writer) *writer << epInfo;
            response.info["episode"] = epInfo;
        }
        totalSteps++;
        return response;
    }

    auto getTotalSteps() const noexcept { return totalSteps; }
    auto getEpisodeRewards() const noexcept { return episodeReturns; }
    auto getEpisodeLengths() const noexcept { return episodeLengths; }
    auto getEpisodeTimes() const noexcept { return episodeTimes; }

private:
    std::chrono::high_resolution_clock::time_point t_start;
    std::optional<ResultWriter> writer;
    std::vector<float> episodeReturns, episodeLengths, episodeTimes;
    size_t totalSteps{};
    std::vector<float> reward{};
    float _steps{};
    bool needsReset{true};
    std::optional<double> invalid_returns;
};

template <class EnvT> using MonitorWithEarlyReset = Monitor<EnvT, true>;   // gym_.h:18-19

}  // namespace gym
#endif  // GYMHIP_USE_REFERENCE_HEADERS
