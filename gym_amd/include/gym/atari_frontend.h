// atari_frontend.h — host front-end that feeds host-emulated Atari envs into the gymhip Atari kernels.
//
// The emulator (ALE) stays on the host.  Per VecEnv step this class runs, for every env and on a pool of host threads,
// the part of the reference's wrapper chain that is control flow around the emulator
//
//     AtariEnv -> NOOPResetEnv -> MaxAndSkipEnv -> Monitor -> EpisodicLifeEnv -> [FireResetEnv] -> (WarpFrame) -> ClipRewardEnv
//     (AtariWrapper::make, wrappers/atari_wrappers.cpp:68-121)  +  SyncVecEnv's auto-reset (sync_env.h:92-100)
//
// and hands the GPU only what is data-parallel: the two frames whose max is the observation go, as the emulator's
// palette-index screens (1 B/pixel), into PINNED staging buffers; the envs are split in two halves so the H2D copy of the
// first half runs under the emulation of the second (double buffering); then one gymhip_atari_step_indexed launch does
// colour table + max + gray + 84x84 area resize + frame-stack ring write, and gymhip_atari_materialize_nchw_f32 the
// VecNormAndPermute view.  PCIe is the bound of this path (2 x 33,600 B per env-step), not HBM.
//
// Reference semantics kept, with file:line:
//   * NOOPResetEnv::reset (noop_reset.h:43-58): reset, then uniform_int_distribution<int>(1, noopMax + 1) no-op steps of
//     the BASE env (one frame each), re-resetting on game over; the observation is the single last frame.
//   * MaxAndSkipEnv::step (atari_wrappers.cpp:10-33): frames skip-2 / skip-1 of the group go to buffer 0 / 1, rewards are
//     summed in float, the loop breaks on done and the buffers then keep OLDER frames; observation = max of the buffers.
//     (Intended semantics: the reference's AtariEnv::getObs hands out one aliased cv::Mat)
//   * Monitor (monitor.h:57-102): running float sum and step count over MaxAndSkip steps — including the steps the
//     reset wrappers take — episode record on done, cleared on (real) reset.
//   * EpisodicLifeEnv (atari_wrappers.cpp:35-53): done on life loss; reset = real reset only after a real game over,
//     otherwise one no-op MaxAndSkip step.
//   * FireResetEnv::reset (atari_wrappers.cpp:55-67): reset, step(1), step(2), re-resetting if either ends the episode;
//     returns the observation of step(2) even then.
//   * ClipRewardEnv (clip_reward.h:22-24), SyncVecEnv auto-reset: terminal obs replaced by the reset obs.
//
// `Emu` is any type with: void reset_game(); int act(int action) (one frame, returns its reward); bool game_over() const;
// int lives() const; const uint8_t* screen() const (src_h * src_w palette indices).  For ALE: reset_game / act /
// game_over / lives / getScreen().getArray() (atari/atari_env.cpp:20-23,101-113; atari/ale_c_wrapper.cpp:6-11).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gymhip.h"

namespace gym {

struct AtariFrontEndOptions {
    int frame_skip = 4;                 // MaxAndSkipEnv(skip), max_skip.h:24
    int noop_max = 30;                  // NOOPResetEnv(noopMax), noop_reset.h:24
    bool terminal_on_life_loss = true;  // EpisodicLifeEnv in the chain (atari_wrappers.cpp:103-109)
    bool fire_reset = false;            // the game's action set has FIRE (atari_wrappers.cpp:111-113)
    bool clip_reward = true;            // atari_wrappers.cpp:117-119
    bool max_as_written = false;        // false = max of the last two frames (the intended semantics, python_gym/atari_wrappers.py:139-150).
                                        // true  = the reference as it runs today: AtariEnv::getObs returns its one cv::Mat by value
                                        // (atari_env.cpp:11-17), a header copy, so both MaxAndSkip buffers alias the same pixels and
                                        // cv::max (atari_wrappers.cpp:30) yields the LAST emulated frame.
    bool stack_as_written = false;      // true = gymhip_atari_set_stack_view(GYMHIP_STACK_AS_WRITTEN): VecFrameStack + VecNormAndPermute as the
                                        // reference computes them today ([prev, new, new, new] channels, signed bytes / 255; gymhip.h)
    bool reset_index_as_written = false;   // reset(int index) as the reference's VecFrameStack has it: every stack shifts, the one frame is
                                        // broadcast over the batch (vec_frame_stack.cpp:34-37); default: only that env's stack starts over
    int n_stack = 4;                    // VecFrameStack
    int src_h = 210, src_w = 160, dst_h = 84, dst_w = 84;
    int device = 0;
    int threads = 8;
    uint64_t seed = 0;                  // env i's NOOP generator is std::mt19937(seed + i)
};

template <class Emu>
class AtariFrontEnd {
public:
    // host-side results of the last reset()/step(), one entry per env
    std::vector<float> reward;          // ClipReward(MaxAndSkip sum)
    std::vector<uint8_t> done;          // after the life-loss rule
    std::vector<uint8_t> has_episode;   // info["episode"] present (monitor.h:80-98)
    std::vector<float> episode_return, episode_length;
    std::vector<int32_t> lives;         // info["ale.lives"]

    AtariFrontEnd(std::vector<std::unique_ptr<Emu>> emulators, const uint32_t* zrgb256_palette, AtariFrontEndOptions o = {})
        : opt(o), n((int64_t)emulators.size()), hw((size_t)o.src_h * o.src_w) {
        if (n <= 0) throw std::invalid_argument("AtariFrontEnd: no emulators");
        if (opt.frame_skip < 2) throw std::invalid_argument("AtariFrontEnd: frame_skip must be >= 2 (two buffers are filled per group)");
        envs.resize(n);
        for (int64_t i = 0; i < n; i++) {
            envs[i].emu = std::move(emulators[i]);
            envs[i].rng.seed(opt.seed + (uint64_t)i);
            envs[i].buf0.assign(hw, 0); envs[i].buf1.assign(hw, 0);
        }
        reward.assign(n, 0.f); done.assign(n, 0); has_episode.assign(n, 0); episode_return.assign(n, 0.f); episode_length.assign(n, 0.f); lives.assign(n, 0);
        check(hipSetDevice(opt.device), "hipSetDevice");
        gymhip_atari_config c{};
        c.struct_size = sizeof c; c.n_envs = n; c.src_h = opt.src_h; c.src_w = opt.src_w; c.dst_h = opt.dst_h; c.dst_w = opt.dst_w;
        c.n_stack = opt.n_stack; c.gray_bits = 15; c.clear_on_done = 0;
        if (gymhip_atari_create(&c, opt.device, &pp) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        if (gymhip_atari_set_palette(pp, zrgb256_palette) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        palette.assign(zrgb256_palette, zrgb256_palette + 256);
        if (opt.stack_as_written && gymhip_atari_set_stack_view(pp, GYMHIP_STACK_AS_WRITTEN) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        const size_t bytes = (size_t)n * 2 * hw;
        check(hipHostMalloc((void**)&staging, bytes, hipHostMallocDefault), "hipHostMalloc");
        check(hipMalloc((void**)&dev_screens, bytes), "hipMalloc(screens)");
        check(hipMalloc((void**)&dev_obs, (size_t)n * opt.n_stack * opt.dst_h * opt.dst_w * sizeof(float)), "hipMalloc(obs)");
        check(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking), "hipStreamCreate");
        check(hipEventCreateWithFlags(&copied, hipEventDisableTiming), "hipEventCreate");
        check(hipEventCreateWithFlags(&consumed, hipEventDisableTiming), "hipEventCreate");
    }
    ~AtariFrontEnd() {
        (void)hipSetDevice(opt.device);
        (void)hipDeviceSynchronize();
        if (pp) gymhip_atari_destroy(pp);
        if (pp_one) gymhip_atari_destroy(pp_one);
        if (dev_plane_one) (void)hipFree(dev_plane_one);
        if (staging) (void)hipHostFree(staging);
        if (dev_screens) (void)hipFree(dev_screens);
        if (dev_obs) (void)hipFree(dev_obs);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (copied) (void)hipEventDestroy(copied);
        if (consumed) (void)hipEventDestroy(consumed);
    }
    AtariFrontEnd(const AtariFrontEnd&) = delete;
    AtariFrontEnd& operator=(const AtariFrontEnd&) = delete;

    int64_t nEnvs() const { return n; }
    // VecEnv::seed (sync_env.h:74-83 -> env.h:60-63): env i's NOOP-reset generator restarts from seed + i
    void seed(uint64_t s) { opt.seed = s; for (int64_t i = 0; i < n; i++) envs[i].rng.seed(s + (uint64_t)i); }
    // VecNormAndPermute(VecFrameStack(...)) observation of the last reset()/step(): device f32 [N][n_stack][dst_h][dst_w], ready on `stream`
    const float* observation() const { return dev_obs; }
    // the engine behind it, e.g. for gymhip_atari_materialize_nchw_16(preprocessor(), out, GYMHIP_OUT_BF16, stream) when the policy runs in half precision
    gymhip_atari* preprocessor() const { return pp; }

    // VecEnv::reset(): every env runs its outermost reset (sync_env.h:66-72 -> ... -> FireResetEnv/EpisodicLifeEnv::reset)
    void reset(hipStream_t stream = nullptr) {
        run_halves([this](int64_t i) { top_reset(i); reward[i] = 0.f; done[i] = 0; has_episode[i] = 0; lives[i] = envs[i].emu->lives(); }, stream);
        if (gymhip_atari_reset_indexed(pp, dev_screens, stream) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        check(hipEventRecord(consumed, stream), "hipEventRecord"); have_consumed = true;
        materialize(stream);
    }

    // VecEnv::reset(int index) (sync_env.h:128-139 under VecFrameStack / VecNormAndPermute): env i alone runs its outermost reset on
    // the host; its two screens go up, a one-env preprocessor (same palette, same kernels) turns them into the 84x84 plane, and that
    // plane enters the ring through gymhip_atari_push_index — by default into env i's stack only, which starts over.
    void reset_index(int64_t i, hipStream_t stream = nullptr) {
        if (i < 0 || i >= n) throw std::out_of_range("AtariFrontEnd::reset_index");
        check(hipSetDevice(opt.device), "hipSetDevice");
        check(hipStreamSynchronize(copy_stream), "hipStreamSynchronize");
        if (!pp_one) {
            gymhip_atari_config c{};
            c.struct_size = sizeof c; c.n_envs = 1; c.src_h = opt.src_h; c.src_w = opt.src_w; c.dst_h = opt.dst_h; c.dst_w = opt.dst_w;
            c.n_stack = 1; c.gray_bits = 15; c.clear_on_done = 0;
            if (gymhip_atari_create(&c, opt.device, &pp_one) != GYMHIP_OK || gymhip_atari_set_palette(pp_one, palette.data()) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
            check(hipMalloc((void**)&dev_plane_one, (size_t)opt.dst_h * opt.dst_w), "hipMalloc(plane)");
        }
        top_reset(i);
        reward[i] = 0.f; done[i] = 0; has_episode[i] = 0; lives[i] = envs[i].emu->lives();
        const size_t off = (size_t)i * 2 * hw;
        check(hipStreamSynchronize(stream), "hipStreamSynchronize");   // the last step's kernel has read dev_screens
        check(hipMemcpyAsync(dev_screens + off, staging + off, 2 * hw, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
        if (gymhip_atari_preprocess_indexed(pp_one, dev_screens + off, dev_plane_one, stream) != GYMHIP_OK ||
            gymhip_atari_push_index(pp, i, dev_plane_one, opt.reset_index_as_written ? 1 : 0, stream) != GYMHIP_OK)
            throw std::runtime_error(gymhip_last_error());
        materialize(stream);
        check(hipStreamSynchronize(stream), "hipStreamSynchronize");   // staging may be rewritten by the next step's emulation
    }

    // VecEnv::step(actions): host int actions, one per env
    void step(const int* actions, hipStream_t stream = nullptr) {
        run_halves([this, actions](int64_t i) { top_step(i, actions[i]); }, stream);
        if (gymhip_atari_step_indexed(pp, dev_screens, nullptr, stream) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
        check(hipEventRecord(consumed, stream), "hipEventRecord"); have_consumed = true;
        materialize(stream);
    }

private:
    struct StepOut { float reward; bool done; int lives; bool has_ep; float ep_r, ep_l; };
    struct EnvState {
        std::unique_ptr<Emu> emu;
        std::mt19937 rng;                     // NOOPResetEnv's m_Device (env.h:83)
        std::vector<uint8_t> buf0, buf1;      // MaxAndSkipEnv::m_ObsBuffer (max_skip.h:15)
        float mon_sum = 0.f; int mon_steps = 0;   // Monitor::reward (summed in order), _steps (monitor.h:116-117)
        int el_lives = 0; bool el_was_real_done = true;   // EpisodicLifeEnv (episodic_life.h:16-17)
    };

    static void check(hipError_t e, const char* what) {
        if (e != hipSuccess) throw std::runtime_error(std::string("AtariFrontEnd: ") + what + ": " + hipGetErrorString(e));
    }
    uint8_t* slot(int64_t i, int k) { return staging + ((size_t)i * 2 + k) * hw; }
    // the observation handed upwards is materialised at the moment the reference would compute it
    void emit_single(int64_t i) { std::memcpy(slot(i, 0), envs[i].emu->screen(), hw); std::memcpy(slot(i, 1), envs[i].emu->screen(), hw); }
    void emit_max(int64_t i) {
        if (opt.max_as_written) { emit_single(i); return; }   // every call site follows the group's last act(): the screen IS that frame
        emit_max_of_two(i);
    }
    void emit_max_of_two(int64_t i) { std::memcpy(slot(i, 0), envs[i].buf0.data(), hw); std::memcpy(slot(i, 1), envs[i].buf1.data(), hw); }

    // AtariEnv::step with frameSkip 1 (atari_env.cpp:101-113)
    StepOut base_step(EnvState& s, int action) {
        const int r = s.emu->act(action);
        return {(float)r, s.emu->game_over(), s.emu->lives(), false, 0.f, 0.f};
    }
    // NOOPResetEnv::reset (noop_reset.h:43-58); observation = the current screen
    void noop_reset(EnvState& s) {
        s.emu->reset_game();
        int noops = std::uniform_int_distribution<int>(1, opt.noop_max + 1)(s.rng);
        while (noops-- > 0) {
            StepOut r = base_step(s, 0);
            if (r.done) s.emu->reset_game();
        }
    }
    // MaxAndSkipEnv::step (atari_wrappers.cpp:10-33); observation = max(buf0, buf1)
    StepOut maxskip_step(EnvState& s, int action) {
        float total = 0.f;
        StepOut r{};
        int t = 0;
        while (t < opt.frame_skip) {
            r = base_step(s, action);
            if (t == opt.frame_skip - 2) std::memcpy(s.buf0.data(), s.emu->screen(), hw);
            else if (t == opt.frame_skip - 1) std::memcpy(s.buf1.data(), s.emu->screen(), hw);
            total += r.reward;
            if (r.done) break;
            t++;
        }
        r.reward = total;
        return r;
    }
    // Monitor::step (monitor.h:74-101)
    StepOut monitor_step(EnvState& s, StepOut r) {
        s.mon_sum += r.reward; s.mon_steps++;
        if (r.done) { r.has_ep = true; r.ep_r = s.mon_sum; r.ep_l = (float)s.mon_steps; }
        return r;
    }
    void monitor_reset(EnvState& s) { s.mon_sum = 0.f; s.mon_steps = 0; }   // monitor.h:57-68
    // EpisodicLifeEnv::step (atari_wrappers.cpp:35-47) over Monitor over MaxAndSkip
    StepOut life_step(EnvState& s, int action) {
        StepOut r = monitor_step(s, maxskip_step(s, action));
        if (!opt.terminal_on_life_loss) return r;
        s.el_was_real_done = r.done;
        if (0 < r.lives && r.lives < s.el_lives) r.done = true;
        s.el_lives = r.lives;
        return r;
    }
    // EpisodicLifeEnv::reset (atari_wrappers.cpp:49-53); returns true if the observation is a single frame
    bool life_reset(EnvState& s) {
        bool single;
        if (!opt.terminal_on_life_loss || s.el_was_real_done) { monitor_reset(s); noop_reset(s); single = true; }
        else { monitor_step(s, maxskip_step(s, 0)); single = false; }
        s.el_lives = s.emu->lives();
        return single;
    }
    // the outermost reset: FireResetEnv::reset (atari_wrappers.cpp:55-67) when the game has FIRE, else EpisodicLifeEnv::reset
    void top_reset(int64_t i) {
        EnvState& s = envs[i];
        const bool single = life_reset(s);
        if (!opt.fire_reset) { if (single) emit_single(i); else emit_max(i); return; }
        StepOut r = life_step(s, 1);
        if (r.done) life_reset(s);
        r = life_step(s, 2);
        emit_max(i);                       // resp.observation of step(2), computed before the reset below
        if (r.done) life_reset(s);
    }
    void top_step(int64_t i, int action) {
        EnvState& s = envs[i];
        StepOut r = life_step(s, action);
        reward[i] = opt.clip_reward ? (r.reward == 0.f ? r.reward : (r.reward > 0.f ? 1.f : -1.f)) : r.reward;   // clip_reward.h:22-24
        done[i] = r.done; has_episode[i] = r.has_ep; episode_return[i] = r.ep_r; episode_length[i] = r.ep_l; lives[i] = r.lives;
        if (r.done) top_reset(i);          // sync_env.h:92-100: the terminal observation is replaced
        else emit_max(i);
    }

    template <class Fn> void run_range(int64_t lo, int64_t hi, Fn fn) {
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(opt.threads, hi - lo));
        if (nt == 1) { for (int64_t i = lo; i < hi; i++) fn(i); return; }
        std::atomic<int64_t> next{lo};
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&]() { for (int64_t i = next.fetch_add(1); i < hi; i = next.fetch_add(1)) fn(i); });
        for (auto& th : pool) th.join();
    }
    // emulate half A, start its H2D copy, emulate half B under it, copy B, make `stream` wait for both copies
    template <class Fn> void run_halves(Fn fn, hipStream_t stream) {
        check(hipSetDevice(opt.device), "hipSetDevice");
        check(hipStreamSynchronize(copy_stream), "hipStreamSynchronize");      // the previous step's copies have left staging
        if (have_consumed) check(hipStreamWaitEvent(copy_stream, consumed, 0), "hipStreamWaitEvent");   // ... and its kernel has read dev_screens
        const int64_t mid = n / 2;
        const int64_t cuts[3] = {0, mid, n};
        for (int h = 0; h < 2; h++) {
            if (cuts[h] == cuts[h + 1]) continue;
            run_range(cuts[h], cuts[h + 1], fn);
            const size_t off = (size_t)cuts[h] * 2 * hw, bytes = (size_t)(cuts[h + 1] - cuts[h]) * 2 * hw;
            check(hipMemcpyAsync(dev_screens + off, staging + off, bytes, hipMemcpyHostToDevice, copy_stream), "hipMemcpyAsync");
        }
        check(hipEventRecord(copied, copy_stream), "hipEventRecord");
        check(hipStreamWaitEvent(stream, copied, 0), "hipStreamWaitEvent");
    }
    void materialize(hipStream_t stream) {
        if (gymhip_atari_materialize_nchw_f32(pp, dev_obs, stream) != GYMHIP_OK) throw std::runtime_error(gymhip_last_error());
    }

    AtariFrontEndOptions opt;
    int64_t n;
    size_t hw;
    std::vector<EnvState> envs;
    gymhip_atari* pp = nullptr;
    gymhip_atari* pp_one = nullptr;   // one-env preprocessor of reset_index (created on first use)
    uint8_t* dev_plane_one = nullptr;
    std::vector<uint32_t> palette;
    uint8_t* staging = nullptr;       // pinned u8 [N][2][H][W]
    uint8_t* dev_screens = nullptr;   // device copy
    float* dev_obs = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied = nullptr, consumed = nullptr;
    bool have_consumed = false;
};

}  // namespace gym
