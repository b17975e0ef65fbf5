// Drives gym::HipAtariVecEnv<FakeEmulator> THROUGH THE gym::VecEnv<false> INTERFACE, the way the reference's Atari smoke test
// holds its stack (test/atari_env.cpp:48-60: unique_ptr<VecEnv<false>> env = VecNormAndPermute(VecFrameStack(SyncVecEnv(...)))),
// and dumps the same records as atari_frontend_demo so tests/test_cpp_host.py checks them against the oracle's restatement of the
// reference chain.   usage: hip_atari_vec_env_demo <n_envs> <steps> <seed> <terminal_on_life_loss 0|1> <fire_reset 0|1> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "fake_emulator.h"
#include "gym/hip_atari_vec_env.h"

static void put(FILE* f, const void* p, size_t bytes) { if (fwrite(p, 1, bytes, f) != bytes) { std::perror("fwrite"); std::exit(9); } }

int main(int argc, char** argv) {
    if (argc != 7) { std::fprintf(stderr, "usage: %s n_envs steps seed life_loss fire out.bin\n", argv[0]); return 2; }
    const int64_t n = std::atoll(argv[1]), T = std::atoll(argv[2]);
    gym::AtariFrontEndOptions o;
    o.terminal_on_life_loss = std::atoi(argv[4]) != 0;
    o.fire_reset = std::atoi(argv[5]) != 0;
    o.threads = 4;
    uint32_t palette[256];
    for (int k = 0; k < 256; k++) palette[k] = (k & 1) ? 0u : (uint32_t)(FakeEmulator::mix(77 + (uint64_t)k) & 0xFFFFFFu);
    std::vector<std::unique_ptr<FakeEmulator>> emus;
    for (int64_t i = 0; i < n; i++) emus.push_back(std::make_unique<FakeEmulator>(1000 + (uint64_t)i));
    std::unique_ptr<gym::VecEnv<false>> env = gym::HipAtariVecEnv<FakeEmulator>::make(std::move(emus), palette, 6, o);
    auto seeds = env->seed(std::strtoull(argv[3], nullptr, 10));                       // VecEnv::seed, not a constructor option
    if ((int64_t)seeds.size() != n) return 3;
    auto sz = env->observationSpace()->size();
    if (sz != std::vector<int64_t>{4, 84, 84} || env->observationSpace()->type() != c10::kFloat || env->actionSpace()->as<gym::Discrete>()->n != 6) return 4;

    FILE* f = std::fopen(argv[6], "wb");
    if (!f) return 5;
    const int64_t hdr[2] = {n, T};
    put(f, hdr, sizeof hdr);
    const int64_t P = 84 * 84;
    auto newest_u8 = [&](torch::Tensor const& obs) {   // channel nStack-1 = the newest frame; x/255 -> the byte
        return (obs.select(1, 3) * 255.f).round().to(torch::kUInt8).to(torch::kCPU).contiguous();
    };
    auto obs = env->reset();
    if (!obs.defined() || !obs.is_cuda() || obs.sizes() != torch::IntArrayRef({n, 4, 84, 84}) || obs.scalar_type() != torch::kFloat32) return 6;
    put(f, newest_u8(obs).data_ptr<uint8_t>(), (size_t)n * P);
    auto idx = torch::arange(n, torch::kInt64);
    gym::VecEnvStepReturn<torch::Tensor> r;
    for (int64_t t = 0; t < T; t++) {
        r = env->step((idx * 7 + t * 3) % 6);                                          // int64 host tensor, like torch::randint
        if (!r.observation.defined() || r.reward.sizes() != torch::IntArrayRef({n, 1}) || r.done.sizes() != torch::IntArrayRef({n, 1}) ||
            r.done.scalar_type() != torch::kFloat32 || (int64_t)r.info.size() != n) return 7;
        put(f, newest_u8(r.observation).data_ptr<uint8_t>(), (size_t)n * P);
        put(f, r.reward.contiguous().data_ptr<float>(), n * 4);
        auto d8 = r.done.to(torch::kUInt8).contiguous();
        put(f, d8.data_ptr<uint8_t>(), n);
        std::vector<uint8_t> has(n); std::vector<float> er(n, 0.f), el(n, 0.f); std::vector<int32_t> lives(n);
        for (int64_t i = 0; i < n; i++) {
            lives[i] = std::any_cast<int>(r.info[i].at("ale.lives"));
            has[i] = r.info[i].count("episode") ? 1 : 0;
            if (has[i]) { auto e = std::any_cast<gym::Result>(r.info[i].at("episode")); er[i] = e.r; el[i] = e.l; }
        }
        put(f, has.data(), n); put(f, er.data(), n * 4); put(f, el.data(), n * 4); put(f, lives.data(), n * 4);
    }
    auto one = env->step(2, torch::Tensor());                                            // buffered stub (sync_env.h:141-157)
    if (!one.reward.defined() || one.reward.item<float>() != r.reward[2].item<float>() || one.info.size() != 1) return 8;
    auto all = (r.observation * 255.f).round().to(torch::kUInt8).to(torch::kCPU).contiguous();
    put(f, all.data_ptr<uint8_t>(), (size_t)n * 4 * P);
    std::fclose(f);
    // reset(int index) (sync_env.h:128-139 under the stack): env 1 alone re-runs its reset chain; its stack starts over, the other
    // envs keep theirs; then the batch steps on.  <out>.ri = env 1's returned observation [4][84][84] as bytes, then the whole batch.
    {
        FILE* g = std::fopen((std::string(argv[6]) + ".ri").c_str(), "wb");
        if (!g) return 5;
        auto ri = env->reset(1);
        if (!ri.defined() || ri.sizes() != torch::IntArrayRef({4, 84, 84})) return 10;
        auto b1 = (ri * 255.f).round().to(torch::kUInt8).to(torch::kCPU).contiguous();
        if (fwrite(b1.data_ptr<uint8_t>(), 1, (size_t)4 * P, g) != (size_t)4 * P) return 9;
        auto r2 = env->step((idx * 5 + 1) % 6);
        if (!r2.observation.defined()) return 7;
        auto b2 = (r2.observation * 255.f).round().to(torch::kUInt8).to(torch::kCPU).contiguous();
        if (fwrite(b2.data_ptr<uint8_t>(), 1, (size_t)n * 4 * P, g) != (size_t)n * 4 * P) return 9;
        std::fclose(g);
    }
    std::printf("ok\n");
    return 0;
}
