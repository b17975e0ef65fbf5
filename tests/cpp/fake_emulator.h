// A deterministic stand-in for an Arcade Learning Environment instance, for tests of the host Atari front-end
// (gym_amd/include/gym/atari_frontend.h) and of its CPU restatement in oracle/.  It is NOT an emulator: it only has the
// interface the front-end needs — the calls atari/atari_env.cpp makes on ale::ALEInterface (reset_game, act, game_over,
// lives, getScreen().getArray()) — and scripted behaviour that exercises every branch of the wrapper chain: rewards of
// both signs, life losses, game-overs at awkward moments (inside a frame-skip group, during the FIRE steps of a reset),
// screens that change every frame.
#pragma once
#include <cstdint>
#include <vector>

struct FakeEmulator {
    static constexpr int H = 210, W = 160;
    uint64_t game;            // per-env constant
    uint64_t frame = 0;       // frames since construction
    uint64_t in_episode = 0;  // frames since reset_game
    int lives_ = 3;
    bool over = false;
    std::vector<uint8_t> scr;

    explicit FakeEmulator(uint64_t game_id) : game(game_id), scr((size_t)H * W) { draw(); }

    static uint64_t mix(uint64_t z) {
        z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void draw() {   // blocky, frame-dependent palette indices (even, like the TIA's)
        for (int y = 0; y < H; y++) {
            const uint64_t row = mix(game * 1315423911ull + frame * 2654435761ull + (uint64_t)(y >> 2));
            for (int x = 0; x < W; x++) scr[(size_t)y * W + x] = (uint8_t)(((row >> ((x >> 4) * 6)) + (uint64_t)x / 3) & 0xFEu);
        }
        scr[(size_t)(frame % H) * W + (size_t)(frame % W)] = (uint8_t)((frame * 2) & 0xFE);   // a moving single pixel
    }
    void reset_game() { in_episode = 0; lives_ = 3; over = false; frame++; draw(); }
    int act(int action) {   // returns the frame's reward
        frame++; in_episode++;
        const uint64_t h = mix(game ^ (frame * 0xD1B54A32D192ED03ull) ^ (uint64_t)action);
        int reward = 0;
        if (h % 5 == 0) reward = (int)((h >> 8) % 4) - 1;                       // -1, 0, 1, 2
        if (!over && (h >> 20) % 23 == 0) { lives_--; if (lives_ <= 0) { lives_ = 0; over = true; } }
        if (in_episode >= 150 + game % 40) over = true;                         // time-out game-over with lives left
        draw();
        return reward;
    }
    bool game_over() const { return over; }
    int lives() const { return lives_; }
    const uint8_t* screen() const { return scr.data(); }
};
