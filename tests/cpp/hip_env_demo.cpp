// The single-env face (gym/hip_env.h) driven like the reference's own CartPole loop (test/classic_control.cpp:57-63): reset,
// step with an action stream, reset when done.  Prints every step's double state so tests/test_cpp_host.py can compare it bit for
// bit with the CPU oracle's single env.   usage: hip_env_demo <cartpole|mountaincar> <seed|-1> <steps>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gym/hip_env.h"

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
    if (argc != 4) { std::fprintf(stderr, "usage: %s cartpole|mountaincar seed steps\n", argv[0]); return 2; }
    const long long seed = std::atoll(argv[2]); const int steps = std::atoi(argv[3]);
    return !std::strcmp(argv[1], "cartpole") ? run<gym::HipCartPoleEnv>(seed, steps, 2) : run<gym::HipMountainCarEnv>(seed, steps, 3);
}
