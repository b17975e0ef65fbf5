// A training-loop-shaped driver of gym::HipVecEnv, modelled on the reference's own smoke test
// (test/classic_control.cpp:69-99, p_test_run): build the vec env, reset, check shapes, step with a tensor of actions.
// Prints one line per step so tests/test_cpp_host.py can compare against the CPU oracle.
//   usage: hip_vec_env_demo <cartpole|mountaincar> <n_envs> <steps> <norm 0|1|2> [checkpoint_dir]     norm 2 = through the gym::HipTrainingVecNormRO wrapper class
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gym/hip_vec_env.h"
#include "gym/monitor_csv.h"

int main(int argc, char** argv) {
    if (argc != 5 && argc != 6) { std::fprintf(stderr, "usage: %s cartpole|mountaincar n_envs steps norm\n", argv[0]); return 2; }
    const bool cart = !std::strcmp(argv[1], "cartpole");
    const int64_t n = std::atoll(argv[2]);
    const int steps = std::atoi(argv[3]);
    const int norm_mode = std::atoi(argv[4]);
    const bool norm = norm_mode != 0;

    std::setvbuf(stdout, nullptr, _IOLBF, 0);
    gym::HipVecEnvOptions o;
    o.env_kind = cart ? GYMHIP_ENV_CARTPOLE : GYMHIP_ENV_MOUNTAINCAR;
    o.n_envs = n;
    o.max_episode_steps = cart ? 500 : 200;
    o.clip_reward = !cart;
    o.tape_episodes = 4;   // far too few on purpose: the mirror's tape guard must extend the tape (same mt19937 streams) before any env runs out
    // norm 2: the reference's composition — a vec env handed to TrainingVecNormRO's constructor (test/classic_control.cpp:105-107)
    std::unique_ptr<gym::VecEnv<false>> env = norm_mode == 2 ? std::unique_ptr<gym::VecEnv<false>>(std::make_unique<gym::HipTrainingVecNormRO>(gym::HipVecEnv::make(o)))
                                              : norm ? gym::makeTrainingVecNormRO(o) : gym::HipVecEnv::make(o);
    gym::HipVecEnv* hv = norm_mode == 2 ? static_cast<gym::HipTrainingVecNormRO*>(env.get())->engineEnv() : static_cast<gym::HipVecEnv*>(env.get());
    auto seeds = env->seed(7);
    if ((int64_t)seeds.size() != n || seeds[1] != 8) return 3;
    const int64_t n_act = env->actionSpace()->as<gym::Discrete>()->n;

    auto obs = env->reset();
    if (obs.size(0) != n || obs.size(1) != (cart ? 4 : 2) || obs.scalar_type() != torch::kFloat32) return 4;   // test/classic_control.cpp:85-86
    std::printf("reset %.9e\n", obs.to(torch::kFloat64).sum().item<double>());
    auto idx = torch::arange(n, torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA, 0));
    // Monitor CSV of envs 0 and 3 when a directory is given (checked against the oracle's episode records by the test)
    std::unique_ptr<gym::MonitorCsvLog> csv;
    if (argc == 6) csv = std::make_unique<gym::MonitorCsvLog>(std::string(argv[5]) + "/monitor", std::vector<int>{0, 3}, 12345u);
    for (int t = 0; t < steps; t++) {
        auto actions = (idx * 7 + t * 3) % n_act;                       // int64 device tensor, like torch::randint(n, N_env)
        auto r = env->step(actions);
        if (!r.reward.defined() || r.reward.size(0) != n || r.reward.size(1) != 1 || r.done.size(1) != 1) return 5;   // sync_env.h:107-112
        std::printf("step %d %lld %.9e %.9e\n", t, (long long)r.done.sum().item<float>(), r.observation.to(torch::kFloat64).sum().item<double>(),
                    r.reward.to(torch::kFloat64).sum().item<double>());
        if (csv) csv->log(hv->infos());
        if (t == steps / 2) {
            // the per-index step is a stub that returns BUFFERED values (sync_env.h:141-157): [1]-shaped reward / done of that env
            auto one = env->step(3, actions);
            if (!one.reward.defined() || one.reward.dim() != 1 || one.reward.size(0) != 1 || one.done.size(0) != 1 || one.info.size() != 1 ||
                one.reward[0].item<float>() != r.reward[3][0].item<float>() || one.done[0].item<float>() != r.done[3][0].item<float>() ||
                !torch::equal(one.observation, r.observation[3])) return 6;
        }
        if (t == steps - 1) {
            if (csv) csv->flush();
            auto infos = hv->infos();
            int with_ep = 0; double len_sum = 0;
            for (auto& m : infos) if (m.count("episode")) { with_ep++; len_sum += std::any_cast<gym::Result>(m["episode"]).l; }
            std::printf("infos %d %.1f\n", with_ep, len_sum);
            auto tot = hv->episodeStatistics();
            std::printf("episodes %.0f %.1f %.1f\n", tot[0], tot[1], tot[2]);
        }
    }
    if (argc == 6 && norm) {
        // VecNormalize::save, then ::load into a second, evaluation-mode env (vec_normalize.h:79-108)
        auto* trained = hv;
        trained->save(argv[5]);
        o.norm_obs = o.norm_reward = true; o.training = false;
        auto eval_env = std::make_unique<gym::HipVecEnv>(o);
        eval_env->load(argv[5]);
        auto a = trained->normStats().cpu(), b = eval_env->normStats().cpu();
        std::printf("ckpt %d\n", (int)torch::equal(a, b));
        // Updateable::update (vec_normalize.h:191-206) and reset(int index) (vec_normalize.h:179-189) through the mirror
        const int64_t F = cart ? 4 : 2;
        auto rows = torch::full({3, n, F}, 0.25f), rew = torch::full({3, n}, -1.0f);
        const float count_before = trained->normStats().cpu()[2 * F].item<float>();
        gym::Updater<gym::VecNormalizeStateRewardPair<false>> learner;   // the observer a learner holds (updateable.h:16-31)
        learner.subscribe(trained);
        learner.notifyOne({rows, rew, torch::Tensor()});
        auto st = trained->normStats().cpu();
        auto row = env->reset(5);
        std::printf("update %d %d\n", (int)(st[2 * F].item<float>() == count_before + 3.0f * n), (int)(row.dim() == 1 && row.size(0) == F));
    }
    return 0;
}
