// ref_dropin_tu.cpp — the in-reference-tree configuration of the drop-in, compiled for real: the REFERENCE's own
// wrappers/vec_env/base.h, vec_env_wrapper.h, wrappers/monitor.h, env.h and spaces (included where they lie, its
// spaces/{space,discrete,util}.cpp compiled in place) + our gym/hip_vec_env.h with GYMHIP_USE_REFERENCE_HEADERS, linked against
// libgymhip.so and the image's libtorch.  No stand-in headers.  What it proves: gym::HipVecEnv IS-A the reference's
// gym::VecEnv<false>, is held by the unique_ptr type the reference's trainers hold (test/classic_control.cpp:31-41) and is
// accepted by the reference's own gym::VecEnvWrapper<false>.  Built by `make -C oracle _ref`; run (no GPU needed: the engine
// is created lazily at the first reset) by tests/test_ref_pin.py, which checks the line it prints.
#define GYMHIP_USE_REFERENCE_HEADERS
#include "wrappers/vec_env/base.h"
#include "wrappers/vec_env/vec_env_wrapper.h"
#include "wrappers/monitor.h"
#include "wrappers/time_limit.h"
#include "wrappers/clip_reward.h"

#include "gym/hip_vec_env.h"
#include "gym/hip_env.h"
#include "gym/hip_vec_frame_stack.h"
#include "gym/hip_atari_vec_env.h"
#include "../../tests/cpp/fake_emulator.h"   // an emulator-shaped type to instantiate gym::HipAtariVecEnv<Emu> with

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../tests/cpp/frame_stack_probe.h"   // run: HipVecFrameStack over a scripted image VecEnv, here on the reference's VecEnvWrapper / spaces
#include "../../tests/cpp/wrapped_probe.h"   // run_wrapped: the same loop the stand-alone mirror's wrappers are driven by

static_assert(std::is_base_of_v<gym::VecEnv<false>, gym::HipVecEnv>, "HipVecEnv derives from the reference's VecEnv<false>");
static_assert(std::is_same_v<gym::VecEnv<false>::StepT, gym::VecEnvStepReturn<torch::Tensor>>, "reference step type");
static_assert(!std::is_abstract_v<gym::HipVecEnv>, "every pure virtual of base.h:52-71 is overridden");
static_assert(std::is_base_of_v<gym::Updateable<gym::VecNormalizeStateRewardPair<false>>, gym::HipVecEnv>,
              "subscribable to the reference's Updater, as VecNormalize is (updateable.h:10-31, vec_normalize.h:21)");

// the single-env face under the reference's own per-env wrapper templates (common/wrapper.h:85-151, time_limit.h:13-45)
static_assert(std::is_base_of_v<gym::Env<std::vector<double>, int>, gym::HipCartPoleEnv>, "same Env instantiation as CartPoleEnv (cartpole.h:14)");
using WrappedSingle = gym::Monitor<gym::TimeLimit<gym::HipCartPoleEnv>, true>;

// a wrapper written against the reference's VecEnvWrapper, as its own VecNormalize / VecFrameStack are (vec_env_wrapper.h:12-55)
static_assert(std::is_base_of_v<gym::VecEnvWrapper<false>, gym::HipTrainingVecNormRO>, "the normaliser wrapper class derives from the reference's VecEnvWrapper<false>, as VecNormalize does (vec_normalize.h:21)");
static_assert(std::is_base_of_v<gym::VecEnvWrapper<false>, gym::HipVecFrameStack> && !std::is_abstract_v<gym::HipVecFrameStack>,
              "the frame-stack wrapper derives from the reference's VecEnvWrapper<false>, as VecFrameStack does (vec_frame_stack.h:16)");
static_assert(std::is_base_of_v<gym::VecEnv<false>, gym::HipAtariVecEnv<FakeEmulator>> && !std::is_abstract_v<gym::HipAtariVecEnv<FakeEmulator>>,
              "the Atari stack is a VecEnv<false> of the reference's, as the trainer holds it (test/atari_env.cpp:48-60)");
template class gym::HipAtariVecEnv<FakeEmulator>;   // every member compiled against the reference's headers, not only the ones a caller here touches
struct CountingWrapper : gym::VecEnvWrapper<false> {
    using gym::VecEnvWrapper<false>::VecEnvWrapper;
    int steps = 0;
    StepT stepWait() noexcept override { steps++; return gym::VecEnvWrapper<false>::stepWait(); }
};

int main(int argc, char** argv) {
    if (argc == 10 && std::string(argv[1]) == "framestack") return frame_stack_probe::run(argv + 2);   // mode n H W k steps host|device out.bin
    gym::HipVecEnvOptions o;
    o.env_kind = GYMHIP_ENV_CARTPOLE; o.n_envs = 64; o.max_episode_steps = 500; o.monitor = true;
    std::unique_ptr<gym::VecEnv<false>> env = gym::makeTrainingVecNormRO(o);          // gym_.h:15-16 shape
    auto seeds = env->seed(7);
    CountingWrapper w(std::move(env));                                                 // the reference's forwarding base owns it now
    auto* box = w.observationSpace()->as<gym::Box<float>>();
    auto* act = w.actionSpace()->as<gym::Discrete>();
    std::printf("dropin nEnvs=%d obs_size=%lld obs_type=%d ranges=%zu act_n=%lld act_size=%lld seeds=%zu first_seed=%zu is_wrapper=%d\n", w.nEnvs(),
                (long long)w.observationSpace()->size()[0], (int)w.observationSpace()->type(), box ? box->getRange().size() : 0, act ? (long long)act->n : -1,
                (long long)w.actionSpace()->size()[0], seeds.size(), seeds.empty() ? 0 : seeds[0], (int)w.isWrapper());
    auto single = std::make_unique<WrappedSingle>(gym::TimeLimit<gym::HipCartPoleEnv>::make(std::make_unique<gym::HipCartPoleEnv>(), 500));
    std::printf("single obs_size=%lld act_n=%lld\n", (long long)single->observationSpace()->size()[0], (long long)single->actionSpace()->as<gym::Discrete>()->n);
    if (argc == 5 && std::string(argv[1]) == "wrapped")        // CartPole under the reference's Monitor<TimeLimit<>>
        return wrapped_probe::run_wrapped<WrappedSingle>(std::make_unique<WrappedSingle>(gym::TimeLimit<gym::HipCartPoleEnv>::make(std::make_unique<gym::HipCartPoleEnv>(), (size_t)std::atoll(argv[4]))),
                                          std::strtoull(argv[2], nullptr, 10), std::atoi(argv[3]), 2);
    if (argc == 5 && std::string(argv[1]) == "wrapped_narrowed") {   // the same stack behind the reference's ObservationWrapper2 (vector<double> -> vector<float>)
        using FO = wrapped_probe::FloatObs<WrappedSingle>;
        return wrapped_probe::run_wrapped<FO>(std::make_unique<FO>(std::make_shared<WrappedSingle>(gym::TimeLimit<gym::HipCartPoleEnv>::make(std::make_unique<gym::HipCartPoleEnv>(), (size_t)std::atoll(argv[4])))),
                                              std::strtoull(argv[2], nullptr, 10), std::atoi(argv[3]), 2);
    }
    if (argc == 5 && std::string(argv[1]) == "wrapped_mc") {   // configs[2]'s per-env stack: Monitor<ClipRewardEnv<TimeLimit<MountainCar>>>
        using TL = gym::TimeLimit<gym::HipMountainCarEnv>; using CR = gym::ClipRewardEnv<TL>; using MO = gym::Monitor<CR, true>;
        return wrapped_probe::run_wrapped<MO>(std::make_unique<MO>(std::make_unique<CR>(TL::make(std::make_unique<gym::HipMountainCarEnv>(), (size_t)std::atoll(argv[4])))),
                               std::strtoull(argv[2], nullptr, 10), std::atoi(argv[3]), 3);
    }
    if (argc > 1 && std::string(argv[1]) == "run") {   // only meaningful where a GPU is visible
        gym::VecEnv<false>& v = w;   // trainers hold the base type; VecEnvWrapper's step(int, Tensor) hides step(actions) by name (vec_env_wrapper.h:49)
        auto obs = v.reset();
        auto r = v.step(torch::zeros({64}, torch::kInt64));
        std::printf("run obs=[%lld,%lld] reward=[%lld,%lld] done=[%lld,%lld] steps=%d\n", (long long)obs.size(0), (long long)obs.size(1), (long long)r.reward.size(0),
                    (long long)r.reward.size(1), (long long)r.done.size(0), (long long)r.done.size(1), w.steps);
    }
    return 0;
}
