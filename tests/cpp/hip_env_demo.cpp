// The single-env face (gym/hip_env.h) driven like the reference's own CartPole loop (test/classic_control.cpp:57-63): reset,
// step with an action stream, reset when done.  Prints every step's double state so tests/test_cpp_host.py can compare it bit for
// bit with the CPU oracle's single env.   usage: hip_env_demo <cartpole|mountaincar> <seed|-1> <steps> [max_episode_steps]
// With max_episode_steps: the stand-alone mirror's wrapper templates (gym/wrappers.h) around the env — Monitor<TimeLimit<CartPole>>,
// Monitor<ClipRewardEnv<TimeLimit<MountainCar>>> — through the loop the reference's own templates are driven by (wrapped_probe.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gym/hip_env.h"
#include "gym/wrappers.h"
#include "wrapped_probe.h"

template <class EnvT> int run(long long seed, int steps, int n_act) {
    EnvT env;
    gym::Env<std::vector<double>, int>& e = env;       // through the base interface (env.h:34-85)
    if (seed >= 0) e.seed((uint64_t)seed);
    if ((int)e.observationSpace()->size()[0] != EnvT::F || e.actionSpace()->template as<gym::Discrete>()->n != n_act) return 3;
    auto s = e.reset();
    if ((int)s.size() != EnvT::F) return 4;
    std::printf("reset"); for (double v : s) std::printf(" %.17g", v); std::printf("\n");
    int episodes = 0;
    for (int t = 0; t < steps; t++) {
        const int a = (int)(((unsigned)t * 2654435761u >> 7) % (unsigned)n_act);
        auto r = e.step(a);
        if ((int)r.observation.size() != EnvT::F) return 5;
        std::printf("step %d %d %.9g %d", t, a, (double)r.reward, (int)r.done);
        for (double v : r.observation) std::printf(" %.17g", v);
        std::printf("\n");
        if (r.done) {
            episodes++;
            if (episodes == 2) {   // misuse once: stepping a finished env keeps it at its terminal state, reward 0, done
                auto again = e.step(a);
                if (!again.done || again.reward != 0.f || again.observation != r.observation) return 6;
            }
            s = e.reset();
            std::printf("reset"); for (double v : s) std::printf(" %.17g", v); std::printf("\n");
        }
    }
    std::printf("episodes %d\n", episodes);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4 && argc != 5) { std::fprintf(stderr, "usage: %s cartpole|mountaincar|cartpole_narrowed seed steps [max_episode_steps]\n", argv[0]); return 2; }
    const long long seed = std::atoll(argv[2]); const int steps = std::atoi(argv[3]);
    if (argc == 5) {
        const size_t max_steps = (size_t)std::atoll(argv[4]);
        if (!std::strcmp(argv[1], "cartpole_narrowed")) {   // ... behind an ObservationWrapper2 that returns vector<float>
            using MO = gym::Monitor<gym::TimeLimit<gym::HipCartPoleEnv>, true>; using FO = wrapped_probe::FloatObs<MO>;
            return wrapped_probe::run_wrapped<FO>(std::make_unique<FO>(std::make_shared<MO>(gym::TimeLimit<gym::HipCartPoleEnv>::make(std::make_unique<gym::HipCartPoleEnv>(), max_steps))), (uint64_t)seed, steps, 2);
        }
        if (!std::strcmp(argv[1], "cartpole")) {
            using MO = gym::Monitor<gym::TimeLimit<gym::HipCartPoleEnv>, true>;
            return wrapped_probe::run_wrapped<MO>(std::make_unique<MO>(gym::TimeLimit<gym::HipCartPoleEnv>::make(std::make_unique<gym::HipCartPoleEnv>(), max_steps)), (uint64_t)seed, steps, 2);
        }
        using TL = gym::TimeLimit<gym::HipMountainCarEnv>; using CR = gym::ClipRewardEnv<TL>; using MO = gym::MonitorWithEarlyReset<CR>;
        return wrapped_probe::run_wrapped<MO>(std::make_unique<MO>(std::make_unique<CR>(TL::make(std::make_unique<gym::HipMountainCarEnv>(), max_steps))), (uint64_t)seed, steps, 3);
    }
    return !std::strcmp(argv[1], "cartpole") ? run<gym::HipCartPoleEnv>(seed, steps, 2) : run<gym::HipMountainCarEnv>(seed, steps, 3);
}
