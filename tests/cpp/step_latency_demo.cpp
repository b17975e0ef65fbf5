// step_latency_demo — what one VecEnv::step costs a C++ caller (the reference's consumers are C++ training loops): the C ABI
// driven from a plain HIP host program, CartPole lockstep, tape resets, one launch per step, actions already on the device.
//   usage: step_latency_demo <n_envs> <steps>      prints: us_per_step env_steps_per_s error_flag
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gymhip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GYM(x) do { if ((x) != GYMHIP_OK) { std::fprintf(stderr, "%s: %s\n", #x, gymhip_last_error()); return 4; } } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 65536;
    const int T = argc > 2 ? std::atoi(argv[2]) : 4000;
    CHECK_HIP(hipSetDevice(0));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    gymhip_config c;
    std::memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.env_kind = GYMHIP_ENV_CARTPOLE;
    c.precision = GYMHIP_PRECISION_LOCKSTEP_F64;
    c.reset_mode = GYMHIP_RESET_TAPE;
    c.wrappers = GYMHIP_WRAP_TIME_LIMIT | GYMHIP_WRAP_MONITOR;
    c.max_episode_steps = 500;
    c.n_envs = n;
    c.clip_obs = 10.f; c.clip_reward = 10.f; c.gamma = 0.99f; c.epsilon = 1e-8f;
    gymhip_engine* h = nullptr;
    CHECK_GYM(gymhip_create(&c, 0, nullptr, 0, &h));
    const int warm = 200;
    CHECK_GYM(gymhip_generate_reset_tape(h, (T + warm) / 8 + 4, stream));
    CHECK_GYM(gymhip_reset(h, stream));
    // four action rows, cycled
    std::vector<int32_t> host((size_t)n * 4);
    uint64_t z = 12345;
    for (auto& a : host) { z = z * 6364136223846793005ull + 1442695040888963407ull; a = (int32_t)((z >> 33) & 1); }
    int32_t* actions = nullptr;
    CHECK_HIP(hipMalloc((void**)&actions, host.size() * 4));
    CHECK_HIP(hipMemcpy(actions, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    for (int t = 0; t < warm; t++) CHECK_GYM(gymhip_step(h, actions + (size_t)(t & 3) * n, GYMHIP_ACT_I32, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; t++) CHECK_GYM(gymhip_step(h, actions + (size_t)(t & 3) * n, GYMHIP_ACT_I32, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int32_t err = 0;
    CHECK_GYM(gymhip_poll_error(h, &err, stream));
    std::printf("step_latency n %lld steps %d us_per_step %.3f env_steps_per_s %.4e error_flag %d\n", (long long)n, T, sec / T * 1e6, (double)n * T / sec, err);
    CHECK_HIP(hipFree(actions));
    gymhip_destroy(h);
    CHECK_HIP(hipStreamDestroy(stream));
    return err ? 1 : 0;
}
