// wrappers.h — the reference's PER-ENV wrapper interface for code that composes single envs outside the reference tree
// (e.g. gym::Monitor<gym::TimeLimit<gym::HipCartPoleEnv>>).  What is kept is the PUBLIC surface — class names, template
// parameters, constructor and member signatures, what a caller observes (results, info keys, stderr text, CSV bytes):
//     Wrapper / ActionWrapper / RewardWrapper / ObservationWrapper      common/wrapper.h:85-229
//     ObservationWrapper2, WrapperUnroll, lastType / NthType            common/wrapper.h:231-275, 26-81
//     TimeLimit                                                         wrappers/time_limit.h:13-45
//     ClipRewardEnv                                                     wrappers/clip_reward.h:13-26
//     Monitor, ResultWriter, MonitorWithEarlyReset                      wrappers/monitor.h:20-123, gym_.h:18-19
// The insides are this repo's own: a derived wrapper's action-taking step()/reset() funnels into its "refine a finished result"
// form (one place per wrapper where its rule lives), the step budget and the episode ledger are small non-template value
// types shared by all instantiations, Monitor keeps a running float sum instead of a per-step reward list, and WrapperUnroll
// folds its stages with a pack expansion.  tests/test_ref_pin.py runs one scripted env under these templates and under the
// reference's own (compiled from its headers where they lie) and compares every output byte.
// Inside the reference tree (GYMHIP_USE_REFERENCE_HEADERS) this header only includes the reference's files.
// Batches do not go through these: gym::HipVecEnv fuses TimeLimit / ClipReward / Monitor into the step kernel.
#pragma once
#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "common/wrapper.h"
#include "wrappers/clip_reward.h"
#include "wrappers/monitor.h"
#include "wrappers/time_limit.h"
#else
#include <cfloat>
#include <chrono>
#include <filesystem>
#include <iostream>
#include <limits>
#include <optional>
#include <sstream>
#include <tuple>
#include <utility>

#include "gym/monitor_csv.h"
#include "gym/vec_env.h"

namespace gym {

namespace detail {

// TimeLimit's rule as a value: `limit` steps may be spent per episode; spend() says whether the budget is now used up.
struct StepBudget {
    size_t limit = std::numeric_limits<size_t>::max();   // no limit given (time_limit.h:43)
    size_t spent = 0;
    void refill() noexcept { spent = 0; }
    bool spend() noexcept { return ++spent >= limit; }
};

// Monitor's bookkeeping as a value.  The episode return is the float rewards added in arrival order starting from 0.f —
// exactly what summing a stored list front to back yields (monitor.h:82) — so no list is kept, except when a finite
// `invalid_returns` threshold may have to print it.
class EpisodeLedger {
public:
    struct Closed { float ret, len, secs; };

    explicit EpisodeLedger(bool keep_trace) : keep_trace_(keep_trace), born_(clock::now()) {}

    size_t born_ticks() const { return (size_t)born_.time_since_epoch().count(); }
    bool awaiting_reset() const noexcept { return awaiting_reset_; }
    void open_episode() noexcept { run_ = 0.f; len_ = 0.f; trace_.clear(); awaiting_reset_ = false; }
    // one step's reward; returns the closed episode when `done`
    std::optional<Closed> book(float reward, bool done) {
        run_ += reward;
        len_ += 1.f;
        if (keep_trace_) trace_.push_back(reward);
        ++steps_seen_;
        if (!done) return std::nullopt;
        awaiting_reset_ = true;
        const Closed c{run_, len_, (float)std::chrono::duration<double>(clock::now() - born_).count()};
        closed_.push_back(c);
        return c;
    }
    size_t steps_seen() const noexcept { return steps_seen_; }
    std::vector<float> const& trace() const noexcept { return trace_; }
    template <float Closed::*Field> std::vector<float> column() const {
        std::vector<float> v;
        v.reserve(closed_.size());
        for (auto const& c : closed_) v.push_back(c.*Field);
        return v;
    }

private:
    using clock = std::chrono::high_resolution_clock;
    bool keep_trace_;
    clock::time_point born_;
    float run_ = 0.f, len_ = 0.f;
    bool awaiting_reset_ = true;      // a fresh Monitor has not been reset yet (monitor.h:118)
    size_t steps_seen_ = 0;
    std::vector<float> trace_;
    std::vector<Closed> closed_;
};

}  // namespace detail

template <typename EnvT>
class Wrapper : public Env<typename EnvT::ObservationT, typename EnvT::ActionT, typename EnvT::StepT> {   // wrapper.h:85-151
public:
    using Base = Env<typename EnvT::ObservationT, typename EnvT::ActionT, typename EnvT::StepT>;
    using typename Base::ActionT;
    using typename Base::ObservationT;
    using typename Base::StepT;

    explicit Wrapper(std::unique_ptr<EnvT> env) : m_Env(std::move(env)) { adopt_spaces(m_Env->observationSpace(), m_Env->actionSpace()); }
    // a stage used for its rule only (no inner env): what WrapperUnroll holds after its first element
    Wrapper(std::shared_ptr<Space> obsSpace, std::shared_ptr<Space> actSpace) { adopt_spaces(std::move(obsSpace), std::move(actSpace)); }

    // forwarding face
    void render() override { m_Env->render(); }
    std::string info() override { return m_Env->info(); }
    void seed(std::optional<uint64_t> const& seed) noexcept override { m_Env->seed(seed); }
    StepT step(ActionT const& action) noexcept override { return m_Env->step(action); }
    ObservationT reset() noexcept override { return m_Env->reset(); }
    // the refine forms (wrapper.h:128-135): where a derived wrapper's rule lives (its action-taking forms ask the inner env and
    // pass the result through these); identity here
    virtual StepT step(StepT&& done_step) noexcept { return std::move(done_step); }
    virtual ObservationT reset(ObservationT&& first_obs) noexcept { return std::move(first_obs); }

    // the nearest env of type T below this wrapper, or nullptr (wrapper.h:137-149)
    template <class T> T* try_cast() const noexcept {
        for (Wrapper const* at = this; at; at = dynamic_cast<Wrapper const*>(at->m_Env.get()))
            if (T* hit = dynamic_cast<T*>(at->m_Env.get())) return hit;
        return nullptr;
    }

protected:
    std::shared_ptr<EnvT> m_Env;   // shared, as in the reference: ObservationWrapper2-style adapters may hold the same env

private:
    void adopt_spaces(std::shared_ptr<Space> o, std::shared_ptr<Space> a) {
        this->m_ObservationSpace = std::move(o);
        this->m_ActionSpace = std::move(a);
    }
};

template <typename EnvT>
class ActionWrapper : public Wrapper<EnvT> {   // wrapper.h:153-170 (tensor-typed, as in the reference)
public:
    using Wrapper<EnvT>::Wrapper;
    using Wrapper<EnvT>::step;
    virtual torch::Tensor action(torch::Tensor const& action) const noexcept = 0;
    virtual torch::Tensor reverseAction(torch::Tensor const& action) const noexcept = 0;
    typename Wrapper<EnvT>::StepT step(typename Wrapper<EnvT>::ActionT const& raw) noexcept override { return this->m_Env->step(action(raw)); }
};

template <typename EnvT>
class RewardWrapper : public Wrapper<EnvT> {   // wrapper.h:172-194
public:
    using Wrapper<EnvT>::Wrapper;
    using Wrapper<EnvT>::step;
    using typename Wrapper<EnvT>::StepT;
    [[nodiscard]] virtual float reward(float const& reward) const noexcept = 0;
    StepT step(typename Wrapper<EnvT>::ActionT const& action) noexcept override { return this->step(this->m_Env->step(action)); }
    StepT step(StepT&& s) noexcept override {
        s.reward = reward(s.reward);
        return std::move(s);
    }
};

template <typename EnvT>
class ObservationWrapper : public Wrapper<EnvT> {   // wrapper.h:196-229
public:
    using Wrapper<EnvT>::Wrapper;
    using typename Wrapper<EnvT>::ObservationT;
    using typename Wrapper<EnvT>::StepT;
    virtual ObservationT observation(ObservationT&&) const noexcept = 0;
    StepT step(typename Wrapper<EnvT>::ActionT const& action) noexcept override { return this->step(this->m_Env->step(action)); }
    ObservationT reset() noexcept override { return observation(this->m_Env->reset()); }
    StepT step(StepT&& s) noexcept override {
        s.observation = observation(std::move(s.observation));
        return std::move(s);
    }
    ObservationT reset(ObservationT&& x) noexcept override { return observation(std::move(x)); }
};

// wrapper.h:231-275.  An observation adapter whose OUTPUT type differs from the inner env's observation type (vector<double> ->
// tensor, image -> vector<float> ...), so it cannot be a Wrapper<EnvT>: it is an Env<OutputT, ActionT> that shares the inner env
// and hands it out through wrapped().
template <typename EnvT, typename OutputT>
class ObservationWrapper2 : public Env<OutputT, typename EnvT::ActionT> {
    using InnerObs = typename EnvT::ObservationT;
    using InnerStep = typename EnvT::StepT;

public:
    ObservationWrapper2(const std::shared_ptr<Space>& obsSpace, std::shared_ptr<EnvT> env) : m_Env(std::move(env)) {
        this->m_ObservationSpace = obsSpace;
        this->m_ActionSpace = m_Env->actionSpace();
    }
    virtual OutputT observation(InnerObs&&) const noexcept = 0;

    // convert a result computed elsewhere ...
    StepResponse<OutputT> step(InnerStep&& s) noexcept { return StepResponse<OutputT>{observation(std::move(s.observation)), s.reward, s.done, std::move(s.info)}; }
    OutputT reset(InnerObs&& x) noexcept { return observation(std::move(x)); }
    // ... or drive the inner env
    StepResponse<OutputT> step(typename EnvT::ActionT const& action) noexcept override { return step(m_Env->step(action)); }
    OutputT reset() noexcept override { return reset(m_Env->reset()); }

    void render() override { m_Env->render(); }
    std::string info() override { return m_Env->info(); }
    void seed(std::optional<uint64_t> const& seed) noexcept override { m_Env->seed(seed); }
    EnvT* wrapped() { return m_Env.get(); }

protected:
    std::shared_ptr<EnvT> m_Env;
};

template <class... Stages> using lastType = std::tuple_element_t<sizeof...(Stages) - 1, std::tuple<Stages...>>;   // wrapper.h:26-30
template <size_t I, class... Stages> using NthType = std::tuple_element_t<I, std::tuple<Stages...>>;

// wrapper.h:32-81.  A wrapper chain fixed at compile time and held BY VALUE: element 0 is an env, every later element is a
// wrapper constructed for its rule only (the (obsSpace, actSpace) constructors), applied left to right to what element 0
// produced.  Types and spaces are the last element's.  render<I>() asks element I for its _render().
template <class... Stages>
struct WrapperUnroll : Env<typename lastType<Stages...>::ObservationT, typename lastType<Stages...>::ActionT, typename lastType<Stages...>::StepT> {
    using Tail = lastType<Stages...>;
    using StepT = typename Tail::StepT;
    using ActionT = typename Tail::ActionT;
    using ObservationT = typename Tail::ObservationT;

    std::tuple<Stages...> wrappers;

    WrapperUnroll(Stages... stages) : wrappers(std::move(stages)...) {
        auto& tail = std::get<sizeof...(Stages) - 1>(wrappers);
        this->m_ObservationSpace = tail.observationSpace();
        this->m_ActionSpace = tail.actionSpace();
    }

    ObservationT reset() noexcept override { return refine(std::get<0>(wrappers).reset(), rules()); }
    StepT step(ActionT const& action) noexcept override { return refine(std::get<0>(wrappers).step(action), rules()); }
    template <size_t I> void render() { std::get<I>(wrappers)._render(); }

private:
    static constexpr auto rules() { return std::make_index_sequence<sizeof...(Stages) - 1>{}; }   // elements 1 .. last
    template <class S> static ObservationT pass(S& stage, ObservationT&& v) { return stage.reset(std::move(v)); }
    template <class S> static StepT pass(S& stage, StepT&& v) { return stage.step(std::move(v)); }
    template <class V, size_t... K> V refine(V v, std::index_sequence<K...>) {
        ((v = pass(std::get<K + 1>(wrappers), std::move(v))), ...);
        return v;
    }
};

// time_limit.h:13-45.  The budget counts the wrapper's own steps; when it is used up the episode ends and
// info["TimeLimit.truncated"] says whether the env itself had NOT finished.
template <class EnvT>
class TimeLimit : public Wrapper<EnvT> {
public:
    using Wrapper<EnvT>::step;
    using Wrapper<EnvT>::reset;
    TimeLimit(std::unique_ptr<EnvT> env, std::optional<size_t> const& max_episode_steps) : Wrapper<EnvT>(std::move(env)) {
        if (max_episode_steps) budget.limit = *max_episode_steps;
    }
    static auto make(std::unique_ptr<EnvT> env, std::optional<size_t> const& max_episode_steps) {
        return std::make_unique<TimeLimit<EnvT>>(std::move(env), max_episode_steps);
    }
    typename EnvT::StepT step(typename EnvT::ActionT const& action) noexcept override {
        auto s = this->m_Env->step(action);
        if (budget.spend()) {
            s.info["TimeLimit.truncated"] = !s.done;
            s.done = true;
        }
        return s;
    }
    typename EnvT::ObservationT reset() noexcept override {
        budget.refill();
        return this->m_Env->reset();
    }

private:
    detail::StepBudget budget;
};

// clip_reward.h:13-26: the reward's sign; a zero of either sign (and only a zero) passes through unchanged.
template <class EnvT>
struct ClipRewardEnv : public RewardWrapper<EnvT> {
    using RewardWrapper<EnvT>::RewardWrapper;
    float reward(float const& rew) const noexcept final {
        if (rew == 0) return rew;
        return rew > 0.f ? 1.f : -1.f;
    }
};

using ResultWriter = EpisodeCsvWriter;   // monitor.h:20-35 (gym/monitor_csv.h writes the same bytes)

// monitor.h:36-122.  Episode return / length / wall time since construction -> info["episode"] = Result{r, l, t} on done,
// optionally appended to `<fileName>.csv`; the getters hand out the history column by column.
template <class EnvType, bool allowEarlyResets = true>
class Monitor : public Wrapper<EnvType> {
    using ObsT = typename EnvType::ObservationT;
    using StepT_ = typename EnvType::StepT;

public:
    explicit Monitor(std::unique_ptr<EnvType> env, std::optional<double> invalid_returns = std::numeric_limits<double>::max(), std::string const& id = "",
                     std::filesystem::path const& fileName = "")
        : Wrapper<EnvType>(std::move(env)), ledger(/*keep_trace=*/invalid_returns && *invalid_returns <= (double)FLT_MAX), suspicious(invalid_returns) {
        if (!fileName.empty()) csv.emplace(fileName, ledger.born_ticks(), id);
    }

    ObsT reset() noexcept override { return reset(this->m_Env->reset()); }
    ObsT reset(ObsT&& first_obs) noexcept override {
        if constexpr (!allowEarlyResets) {
            // the reference warns when the episode HAS ended (it tests the flag, not its negation, monitor.h:62-66): same here
            if (ledger.awaiting_reset())
                std::cerr << "Tried to reset an environment before done. If you want to allow early resets, "
                             "wrap your env with Monitor(env, path, allow_early_resets=True)\n";
        }
        ledger.open_episode();
        return std::move(first_obs);
    }

    StepT_ step(typename EnvType::ActionT const& action) noexcept override { return step(this->m_Env->step(action)); }
    StepT_ step(StepT_&& s) noexcept override {
        if (ledger.awaiting_reset()) std::cerr << "Tried to step environment that needs reset\n";
        if (auto ep = ledger.book(s.reward, s.done)) {
            const Result record{ep->ret, ep->len, ep->secs};
            if (suspicious && record.r >= *suspicious) complain(record);
            if (csv) *csv << record;
            s.info["episode"] = record;
        }
        return std::move(s);
    }

    auto getTotalSteps() const noexcept { return ledger.steps_seen(); }
    auto getEpisodeRewards() const noexcept { return ledger.template column<&detail::EpisodeLedger::Closed::ret>(); }
    auto getEpisodeLengths() const noexcept { return ledger.template column<&detail::EpisodeLedger::Closed::len>(); }
    auto getEpisodeTimes() const noexcept { return ledger.template column<&detail::EpisodeLedger::Closed::secs>(); }

private:
    void complain(Result const& record) {   // monitor.h:87-92: one stderr block with the episode's rewards
        std::ostringstream msg;
        msg << this->info() << "\nInvalid Episode Returns: " << record.r << "\n";
        for (float r : ledger.trace()) msg << r << " ";
        msg << " Steps: " << record.l;
        std::cerr << msg.str() << "\n";
    }

    detail::EpisodeLedger ledger;
    std::optional<double> suspicious;
    std::optional<ResultWriter> csv;
};

template <class EnvT> using MonitorWithEarlyReset = Monitor<EnvT, true>;   // gym_.h:18-19

}  // namespace gym
#endif  // GYMHIP_USE_REFERENCE_HEADERS
