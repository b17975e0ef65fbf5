// rccl_moments_demo — the sharded-normaliser call sequence of INTEGRATION.md on the real collective library, from C++ with no
// libtorch, for N ranks: ONE process drives N GPUs (ncclCommInitAll), rank d owns envs [d n, (d + 1) n) of a global batch of N n:
//     gymhip_step_begin(rank d)  ->  ncclAllReduce(moments) on each rank's launch stream (one group call)  ->  gymhip_step_finish(rank d)
// checked every step against ONE engine that holds all N n envs and takes the unsplit gymhip_step: raw observations and done
// flags bit for bit, statistics to 1e-6, the replicated statistics bit-identical on every rank, the all-reduced batch count = N n.
// Both exchange forms (gymhip_set_exchange_form): "moments" (reduce launch, 2F+6 doubles) and "rows" (the partial rows themselves).
// Then the forms are timed at the per-GPU size: unsplit (no exchange), eager, and — one rank only — replayed from a HIP graph that
// holds the RCCL call.  With one visible GPU the communicator has one rank and the "moments" form must equal the unsplit step bit
// for bit; asking for more GPUs than are visible is an error (exit 6), so an N-GPU log cannot show fewer ranks than it claims.
//   usage: rccl_moments_demo <n_envs_per_gpu> <steps> [n_gpus = 1] [time_steps = 0]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
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

static gymhip_config config(int64_t n, int64_t offset) {
    gymhip_config c;
    std::memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.env_kind = GYMHIP_ENV_MOUNTAINCAR;
    c.precision = GYMHIP_PRECISION_LOCKSTEP_F64;
    c.reset_mode = GYMHIP_RESET_COUNTER_RNG;
    c.wrappers = GYMHIP_WRAP_TIME_LIMIT | GYMHIP_WRAP_CLIP_REWARD | GYMHIP_WRAP_MONITOR | GYMHIP_WRAP_NORM_OBS | GYMHIP_WRAP_NORM_REWARD | GYMHIP_WRAP_NORM_TRAINING;
    c.max_episode_steps = 200;
    c.n_envs = n;
    c.env_offset = offset;
    c.clip_obs = 10.f; c.clip_reward = 10.f; c.gamma = 0.99f; c.epsilon = 1e-8f;
    return c;
}

struct Rank {
    int dev = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    gymhip_engine* eng = nullptr;
    gymhip_buffers buf{};
    int32_t* actions = nullptr;
};

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 20000;
    const int T = argc > 2 ? std::atoi(argv[2]) : 50;
    const int want = argc > 3 ? std::atoi(argv[3]) : 1;
    const int time_steps = argc > 4 ? std::atoi(argv[4]) : 0;
    int visible = 0;
    CHECK_HIP(hipGetDeviceCount(&visible));
    if (want < 1 || want > visible) { std::fprintf(stderr, "asked for %d GPUs, %d visible\n", want, visible); return 6; }
    const int N = want;
    std::vector<Rank> ranks(N);
    std::vector<int> devs(N);
    std::vector<ncclComm_t> comms(N);
    for (int d = 0; d < N; d++) devs[d] = d;
    CHECK_NCCL(ncclCommInitAll(comms.data(), N, devs.data()));
    for (int d = 0; d < N; d++) {
        ranks[d].dev = d; ranks[d].comm = comms[d];
        CHECK_HIP(hipSetDevice(d));
        CHECK_HIP(hipStreamCreateWithFlags(&ranks[d].stream, hipStreamNonBlocking));
        CHECK_HIP(hipMalloc((void**)&ranks[d].actions, (size_t)n * 4));
        int cr = -1, cn = -1;
        CHECK_NCCL(ncclCommCount(comms[d], &cn));
        CHECK_NCCL(ncclCommUserRank(comms[d], &cr));
        if (cn != N || cr != d) { std::fprintf(stderr, "RCCL reports %d ranks (rank %d), expected rank %d of %d\n", cn, cr, d, N); return 5; }
        if (d == 0) std::printf("rccl_comm ranks %d rank %d\n", cn, cr);   // the communicator reports the size it was given
    }

    auto exchange = [&]() -> int {   // one group call: every rank's payload on its own launch stream
        CHECK_NCCL(ncclGroupStart());
        for (auto& r : ranks) CHECK_NCCL(ncclAllReduce(r.buf.moments, r.buf.moments, (size_t)r.buf.moments_len, ncclDouble, ncclSum, r.comm, r.stream));
        CHECK_NCCL(ncclGroupEnd());
        return 0;
    };
    auto sync_all = [&]() -> int { for (auto& r : ranks) { CHECK_HIP(hipSetDevice(r.dev)); CHECK_HIP(hipStreamSynchronize(r.stream)); } return 0; };

    int all_ok = 1;
    for (int form : {GYMHIP_EXCHANGE_MOMENTS, GYMHIP_EXCHANGE_ROWS}) {
        const char* fname = form == GYMHIP_EXCHANGE_MOMENTS ? "moments" : "rows";
        // the whole batch on GPU 0, unsplit
        CHECK_HIP(hipSetDevice(0));
        gymhip_engine* whole = nullptr;
        gymhip_config cw = config(n * N, 0);
        CHECK_GYM(gymhip_create(&cw, 0, nullptr, 0, &whole));
        gymhip_buffers bw;
        CHECK_GYM(gymhip_get_buffers(whole, &bw));
        int32_t* whole_actions = nullptr;
        CHECK_HIP(hipMalloc((void**)&whole_actions, (size_t)n * N * 4));
        for (int d = 0; d < N; d++) {
            CHECK_HIP(hipSetDevice(d));
            gymhip_config c = config(n, (int64_t)d * n);
            CHECK_GYM(gymhip_create(&c, d, nullptr, 0, &ranks[d].eng));
            CHECK_GYM(gymhip_set_exchange_form(ranks[d].eng, form));
            CHECK_GYM(gymhip_get_buffers(ranks[d].eng, &ranks[d].buf));
        }
        CHECK_GYM(gymhip_reset(whole, ranks[0].stream));
        for (auto& r : ranks) CHECK_GYM(gymhip_reset_begin(r.eng, r.stream));
        if (exchange()) return 3;
        for (auto& r : ranks) CHECK_GYM(gymhip_reset_finish(r.eng, r.stream));

        std::vector<int32_t> host((size_t)n * N);
        std::vector<float> raw_w((size_t)n * N * 2), raw_s((size_t)n * 2), done_w((size_t)n * N), done_s((size_t)n), sw(16), s0(16), sd(16);
        int raw_identical = 1, replicated = 1, bits_identical = 1;
        double worst_stat = 0.0, count = 0.0;
        for (int t = 0; t < T; t++) {
            for (size_t i = 0; i < host.size(); i++) host[i] = (int32_t)(splitmix((uint64_t)t * host.size() + i) % 3);
            CHECK_HIP(hipSetDevice(0));
            CHECK_HIP(hipMemcpyAsync(whole_actions, host.data(), host.size() * 4, hipMemcpyHostToDevice, ranks[0].stream));
            CHECK_GYM(gymhip_step(whole, whole_actions, GYMHIP_ACT_I32, ranks[0].stream));
            for (auto& r : ranks) {
                CHECK_HIP(hipSetDevice(r.dev));
                CHECK_HIP(hipMemcpyAsync(r.actions, host.data() + (size_t)r.dev * n, (size_t)n * 4, hipMemcpyHostToDevice, r.stream));
                CHECK_GYM(gymhip_step_begin(r.eng, r.actions, GYMHIP_ACT_I32, r.stream));
            }
            if (exchange()) return 3;
            for (auto& r : ranks) CHECK_GYM(gymhip_step_finish(r.eng, r.stream));
            CHECK_HIP(hipSetDevice(0));
            CHECK_HIP(hipMemcpyAsync(raw_w.data(), bw.obs, raw_w.size() * 4, hipMemcpyDeviceToHost, ranks[0].stream));
            CHECK_HIP(hipMemcpyAsync(done_w.data(), bw.done, done_w.size() * 4, hipMemcpyDeviceToHost, ranks[0].stream));
            CHECK_GYM(gymhip_get_norm_stats(whole, sw.data(), ranks[0].stream));
            for (auto& r : ranks) {
                CHECK_HIP(hipSetDevice(r.dev));
                CHECK_HIP(hipMemcpyAsync(raw_s.data(), r.buf.obs, raw_s.size() * 4, hipMemcpyDeviceToHost, r.stream));
                CHECK_HIP(hipMemcpyAsync(done_s.data(), r.buf.done, done_s.size() * 4, hipMemcpyDeviceToHost, r.stream));
                CHECK_GYM(gymhip_get_norm_stats(r.eng, sd.data(), r.stream));   // synchronises the stream
                if (std::memcmp(raw_s.data(), raw_w.data() + (size_t)r.dev * n * 2, raw_s.size() * 4) || std::memcmp(done_s.data(), done_w.data() + (size_t)r.dev * n, done_s.size() * 4)) raw_identical = 0;
                if (r.dev == 0) s0 = sd;
                else if (std::memcmp(sd.data(), s0.data(), 8 * 4)) replicated = 0;
                if (std::memcmp(sd.data(), sw.data(), 8 * 4)) bits_identical = 0;
                for (int k = 0; k < 8; k++) worst_stat = std::fmax(worst_stat, std::fabs((double)sd[k] - sw[k]) / std::fmax(std::fabs((double)sw[k]), 1e-6));
            }
            double m0 = 0.0;
            CHECK_HIP(hipSetDevice(0));
            CHECK_HIP(hipMemcpy(&m0, ranks[0].buf.moments + (form == GYMHIP_EXCHANGE_MOMENTS ? 0 : 15), 8, hipMemcpyDeviceToHost));
            if (m0 != (double)n * N) { std::fprintf(stderr, "step %d: all-reduced batch count %.1f, expected %.1f\n", t, m0, (double)n * N); all_ok = 0; }
            count = sd[4];
        }
        const int ok = raw_identical && replicated && worst_stat <= 1e-6 && (N > 1 || form != GYMHIP_EXCHANGE_MOMENTS || bits_identical);
        std::printf("rccl_form %s ranks %d payload_bytes %d raw_identical %d replicated %d stats_bit_identical %d worst_stat_rel %.3g\n", fname, N, ranks[0].buf.moments_len * 8,
                    raw_identical, replicated, bits_identical, worst_stat);
        if (form == GYMHIP_EXCHANGE_MOMENTS) std::printf("rccl_identical %d steps %d obs_count %.1f\n", ok, T, count);
        all_ok = all_ok && ok;

        if (time_steps > 0) {
            using clk = std::chrono::steady_clock;
            auto us_per_step = [&](auto&& body) -> double {
                for (int i = 0; i < 50; i++) body();
                if (sync_all()) return -1;
                const auto t0 = clk::now();
                for (int i = 0; i < time_steps; i++) body();
                if (sync_all()) return -1;
                return std::chrono::duration<double, std::micro>(clk::now() - t0).count() / time_steps;
            };
            const double eager = us_per_step([&]() {
                for (auto& r : ranks) gymhip_step_begin(r.eng, r.actions, GYMHIP_ACT_I32, r.stream);
                exchange();
                for (auto& r : ranks) gymhip_step_finish(r.eng, r.stream);
            });
            double graph_us = -1.0, unsplit = -1.0;
            if (N == 1) {
                Rank& r = ranks[0];
                gymhip_engine* solo = nullptr;   // the unsplit step at the per-GPU size (no exchange)
                gymhip_config c = config(n, 0);
                CHECK_GYM(gymhip_create(&c, 0, nullptr, 0, &solo));
                CHECK_GYM(gymhip_reset(solo, r.stream));
                unsplit = us_per_step([&]() { gymhip_step(solo, r.actions, GYMHIP_ACT_I32, r.stream); });
                gymhip_destroy(solo);
                hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                CHECK_HIP(hipStreamBeginCapture(r.stream, hipStreamCaptureModeThreadLocal));
                int rc1 = gymhip_step_begin(r.eng, r.actions, GYMHIP_ACT_I32, r.stream);
                ncclResult_t rc2 = ncclAllReduce(r.buf.moments, r.buf.moments, (size_t)r.buf.moments_len, ncclDouble, ncclSum, r.comm, r.stream);
                int rc3 = gymhip_step_finish(r.eng, r.stream);
                hipError_t ce = hipStreamEndCapture(r.stream, &graph);
                if (rc1 == GYMHIP_OK && rc2 == ncclSuccess && rc3 == GYMHIP_OK && ce == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    graph_us = us_per_step([&]() { (void)hipGraphLaunch(exec, r.stream); });
                    (void)hipGraphExecDestroy(exec);
                } else {
                    std::fprintf(stderr, "graph capture of the %s form failed (%d, %s, %d, %s)\n", fname, rc1, ncclGetErrorString(rc2), rc3, hipGetErrorString(ce));
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            std::printf("rccl_time form %s ranks %d envs_per_gpu %lld eager_us %.2f graph_us %.2f unsplit_us %.2f\n", fname, N, (long long)n, eager, graph_us, unsplit);
        }
        for (auto& r : ranks) { gymhip_destroy(r.eng); r.eng = nullptr; }
        gymhip_destroy(whole);
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipFree(whole_actions));
    }
    for (auto& r : ranks) {
        CHECK_HIP(hipSetDevice(r.dev));
        CHECK_HIP(hipFree(r.actions));
        ncclCommDestroy(r.comm);
        CHECK_HIP(hipStreamDestroy(r.stream));
    }
    return all_ok ? 0 : 1;
}
