// Drives gym::AtariFrontEnd<FakeEmulator> (gym_amd/include/gym/atari_frontend.h) like a rollout worker: reset, then T steps
// with a fixed action stream, and dumps everything the wrapper chain returns so tests/test_cpp_host.py can compare it with
// the oracle's class-by-class restatement of the reference chain.
//   usage: atari_frontend_demo <n_envs> <steps> <seed> <terminal_on_life_loss 0|1> <fire_reset 0|1> <out.bin> [as_written 0|1]   (as_written: MaxAndSkip aliasing + the as-written stack view)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "fake_emulator.h"
#include "gym/atari_frontend.h"

static void put(FILE* f, const void* p, size_t bytes) { if (fwrite(p, 1, bytes, f) != bytes) { std::perror("fwrite"); std::exit(9); } }

int main(int argc, char** argv) {
    if (argc != 7 && argc != 8) { std::fprintf(stderr, "usage: %s n_envs steps seed life_loss fire out.bin\n", argv[0]); return 2; }
    const int64_t n = std::atoll(argv[1]), T = std::atoll(argv[2]);
    gym::AtariFrontEndOptions o;
    o.seed = std::strtoull(argv[3], nullptr, 10);
    o.terminal_on_life_loss = std::atoi(argv[4]) != 0;
    o.fire_reset = std::atoi(argv[5]) != 0;
    o.threads = 4;
    o.max_as_written = o.stack_as_written = argc == 8 && std::atoi(argv[7]) != 0;
    uint32_t palette[256];
    for (int k = 0; k < 256; k++) palette[k] = (k & 1) ? 0u : (uint32_t)(FakeEmulator::mix(77 + (uint64_t)k) & 0xFFFFFFu);   // NTSC-table shape: odd entries 0
    std::vector<std::unique_ptr<FakeEmulator>> emus;
    for (int64_t i = 0; i < n; i++) emus.push_back(std::make_unique<FakeEmulator>(1000 + (uint64_t)i));
    gym::AtariFrontEnd<FakeEmulator> fe(std::move(emus), palette, o);

    FILE* f = std::fopen(argv[6], "wb");
    if (!f) return 3;
    const int64_t hdr[2] = {n, T};
    put(f, hdr, sizeof hdr);
    const size_t P = 84 * 84, K = 4;
    std::vector<float> obs((size_t)n * K * P);
    std::vector<uint8_t> u8((size_t)n * K * P), newest((size_t)n * P);
    auto fetch = [&]() {
        if (hipMemcpy(obs.data(), fe.observation(), obs.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) std::exit(4);   // syncs the default stream
        for (size_t k = 0; k < obs.size(); k++) u8[k] = (uint8_t)std::lrintf(obs[k] * 255.f);   // as written the values are signed bytes / 255: the cast wraps them back
        for (int64_t e = 0; e < n; e++) std::memcpy(&newest[(size_t)e * P], &u8[((size_t)e * K + (K - 1)) * P], P);
    };
    fe.reset();
    fetch();
    put(f, newest.data(), newest.size());
    std::vector<int> actions(n);
    for (int64_t t = 0; t < T; t++) {
        for (int64_t i = 0; i < n; i++) actions[i] = (int)((i * 7 + t * 3) % 6);
        fe.step(actions.data());
        fetch();
        put(f, newest.data(), newest.size());
        put(f, fe.reward.data(), n * 4); put(f, fe.done.data(), n); put(f, fe.has_episode.data(), n);
        put(f, fe.episode_return.data(), n * 4); put(f, fe.episode_length.data(), n * 4); put(f, fe.lives.data(), n * 4);
    }
    put(f, u8.data(), u8.size());
    std::fclose(f);
    std::printf("ok\n");
    return 0;
}
