// rccl_moments_demo — the sharded-normaliser call sequence of INTEGRATION.md on the real collective library, from C++ with
// no libtorch: gymhip_step_begin -> ncclAllReduce(moments) on the launch stream -> gymhip_step_finish, against the unsplit
// gymhip_step.  A 1-GPU box can host one rank only (RCCL refuses two ranks on one device), so the communicator has one rank
// and the results must be identical; what this checks is the binding: buffer, count, dtype, stream order, link line.
//   usage: rccl_moments_demo <n_envs> <steps>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gymhip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define CHECK_GYM(x) do { if ((x) != GYMHIP_OK) { std::fprintf(stderr, "%s: %s\n", #x, gymhip_last_error()); return 4; } } while (0)

static uint64_t splitmix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 20000;
    const int T = argc > 2 ? std::atoi(argv[2]) : 50;
    CHECK_HIP(hipSetDevice(0));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    ncclComm_t comm;
    int dev = 0;
    CHECK_NCCL(ncclCommInitAll(&comm, 1, &dev));
    // the communicator must report the size it was given: on an N-GPU run this line is what shows N ranks in the log
    int comm_ranks = -1, comm_rank = -1;
    CHECK_NCCL(ncclCommCount(comm, &comm_ranks));
    CHECK_NCCL(ncclCommUserRank(comm, &comm_rank));
    std::printf("rccl_comm ranks %d rank %d\n", comm_ranks, comm_rank);
    if (comm_ranks != 1 || comm_rank != 0) { std::fprintf(stderr, "RCCL reports %d ranks (rank %d), expected the 1 it was created with\n", comm_ranks, comm_rank); return 5; }

    gymhip_config c;
    std::memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.env_kind = GYMHIP_ENV_MOUNTAINCAR;
    c.precision = GYMHIP_PRECISION_LOCKSTEP_F64;
    c.reset_mode = GYMHIP_RESET_COUNTER_RNG;
    c.wrappers = GYMHIP_WRAP_TIME_LIMIT | GYMHIP_WRAP_CLIP_REWARD | GYMHIP_WRAP_MONITOR | GYMHIP_WRAP_NORM_OBS | GYMHIP_WRAP_NORM_REWARD | GYMHIP_WRAP_NORM_TRAINING;
    c.max_episode_steps = 200;
    c.n_envs = n;
    c.clip_obs = 10.f; c.clip_reward = 10.f; c.gamma = 0.99f; c.epsilon = 1e-8f;
    gymhip_engine *a = nullptr, *b = nullptr;
    CHECK_GYM(gymhip_create(&c, 0, nullptr, 0, &a));
    CHECK_GYM(gymhip_create(&c, 0, nullptr, 0, &b));
    gymhip_buffers ba, bb;
    CHECK_GYM(gymhip_get_buffers(a, &ba));
    CHECK_GYM(gymhip_get_buffers(b, &bb));

    auto allreduce = [&](gymhip_buffers& buf) { return ncclAllReduce(buf.moments, buf.moments, (size_t)buf.moments_len, ncclDouble, ncclSum, comm, stream); };

    CHECK_GYM(gymhip_reset(a, stream));
    CHECK_GYM(gymhip_reset_begin(b, stream));
    CHECK_NCCL(allreduce(bb));
    CHECK_GYM(gymhip_reset_finish(b, stream));

    int32_t* actions = nullptr;
    CHECK_HIP(hipMalloc((void**)&actions, (size_t)n * 4));
    std::vector<int32_t> host(n);
    std::vector<float> xa((size_t)n * 2), xb((size_t)n * 2), sa(16), sb(16);
    int identical = 1;
    for (int t = 0; t < T; t++) {
        for (int64_t i = 0; i < n; i++) host[i] = (int32_t)(splitmix((uint64_t)t * n + i) % 3);
        CHECK_HIP(hipMemcpyAsync(actions, host.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
        CHECK_GYM(gymhip_step(a, actions, GYMHIP_ACT_I32, stream));
        CHECK_GYM(gymhip_step_begin(b, actions, GYMHIP_ACT_I32, stream));
        CHECK_NCCL(allreduce(bb));
        CHECK_GYM(gymhip_step_finish(b, stream));
        CHECK_HIP(hipMemcpyAsync(xa.data(), ba.obs_norm, xa.size() * 4, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(xb.data(), bb.obs_norm, xb.size() * 4, hipMemcpyDeviceToHost, stream));
        CHECK_GYM(gymhip_get_norm_stats(a, sa.data(), stream));
        CHECK_GYM(gymhip_get_norm_stats(b, sb.data(), stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        if (std::memcmp(xa.data(), xb.data(), xa.size() * 4) || std::memcmp(sa.data(), sb.data(), 8 * 4)) identical = 0;
    }
    std::printf("rccl_identical %d steps %d obs_count %.1f\n", identical, T, (double)sb[4]);
    CHECK_HIP(hipFree(actions));
    gymhip_destroy(a);
    gymhip_destroy(b);
    ncclCommDestroy(comm);
    CHECK_HIP(hipStreamDestroy(stream));
    return identical ? 0 : 1;
}
