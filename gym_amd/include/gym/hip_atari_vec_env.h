// hip_atari_vec_env.h — gym::HipAtariVecEnv<Emu> : gym::VecEnv<false>, the drop-in for the Atari stack a reference trainer
// composes and holds as std::unique_ptr<gym::VecEnv<false>> (test/atari_env.cpp:48-60):
//
//     VecNormAndPermute( VecFrameStack( SyncVecEnv( AtariWrapper::make(AtariEnv) x N ), nStack ) )
//     (wrappers/vec_env/vec_atari.cpp:70-81, vec_frame_stack.cpp:10-32, sync_env.h:85-123, atari_wrappers.cpp:68-121)
//
// It is a thin VecEnv<false> face over gym::AtariFrontEnd<Emu> (gym/atari_frontend.h): the emulators and the wrapper chain's
// control flow run on host threads, the pixels (palette colouring, max of two frames, gray, 84x84 area resize, frame-stack
// ring, permute + /255) on the GPU.  What the interface returns:
//     reset()/stepWait().observation   f32 [N, nStack, 84, 84] ON THE GPU (a view of the front-end's buffer, overwritten by the
//                                      next step) — VecNormAndPermute's tensor (vec_atari.cpp:74-76)
//     reward, done                     f32 [N, 1] host tensors, as the reference's (sync_env.h:107-112): they come from host emulators
//     info[i]                          "ale.lives" (atari_env.cpp:110) always; "episode" = Result{r, l, t} on real game-overs (monitor.h:86-97)
//     observationSpace()               Box<float>(0, 1, {nStack, 84, 84}) (vec_atari.cpp:49-55), actionSpace() Discrete(n_actions)
// Inside the reference tree: #define GYMHIP_USE_REFERENCE_HEADERS and include the reference's base.h / monitor.h / spaces first
// (see gym/hip_vec_env.h).
#pragma once
#include <chrono>
#include <cstdio>

#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "spaces/box.h"
#include "spaces/discrete.h"
#else
#include "gym/vec_env.h"
#endif
#include "gym/atari_frontend.h"

namespace gym {

template <class Emu>
class HipAtariVecEnv : public VecEnv<false> {
public:
    HipAtariVecEnv(std::vector<std::unique_ptr<Emu>> emulators, const uint32_t* zrgb256_palette, int n_actions, AtariFrontEndOptions o = {})
        : VecEnv<false>((int)emulators.size(),
                        std::make_shared<Box<float>>(Range<float>{0.f, 1.f}, Shape{{(int64_t)o.n_stack, (int64_t)o.dst_h, (int64_t)o.dst_w}, c10::kFloat}),
                        std::make_shared<Discrete>(n_actions)),
          opt(o), fe(std::move(emulators), zrgb256_palette, o) {
        t_start = std::chrono::high_resolution_clock::now();
    }

    static std::unique_ptr<VecEnv<false>> make(std::vector<std::unique_ptr<Emu>> emulators, const uint32_t* palette, int n_actions, AtariFrontEndOptions o = {}) {
        return std::make_unique<HipAtariVecEnv<Emu>>(std::move(emulators), palette, n_actions, o);   // vec_frame_stack.h:34-36 idiom
    }

    torch::Tensor reset() noexcept override {   // vec_atari.cpp:83-85 over vec_frame_stack.cpp:22-25 over sync_env.h:66-72
        try { fe.reset(stream); } catch (std::exception const& e) { std::fprintf(stderr, "HipAtariVecEnv::reset: %s\n", e.what()); return {}; }
        did_reset = true;
        return observation();
    }

    void stepAsync(torch::Tensor const& actions) noexcept override { pending = actions; }   // sync_env.h:85-87

    StepT stepWait() noexcept override {   // vec_atari.cpp:98-102 over vec_frame_stack.cpp:27-32 over sync_env.h:114-123
        if (!did_reset || !pending.defined()) { std::fprintf(stderr, "HipAtariVecEnv::stepWait before reset()/stepAsync()\n"); return {}; }
        // TensorAdapter::decode (tensor_adapter.cpp:22-23,35-40): any tensor is narrowed to int32 on the host
        torch::Tensor a = pending.to(torch::kCPU, torch::kInt32).contiguous().reshape({-1});
        pending = torch::Tensor();
        if (a.numel() != numEnvs) { std::fprintf(stderr, "HipAtariVecEnv: expected %d actions, got %lld\n", numEnvs, (long long)a.numel()); return {}; }
        try { fe.step(a.data_ptr<int>(), stream); } catch (std::exception const& e) { std::fprintf(stderr, "HipAtariVecEnv::stepWait: %s\n", e.what()); return {}; }
        const int64_t n = numEnvs;
        auto reward = torch::from_blob(fe.reward.data(), {n}, torch::kFloat32).clone().unsqueeze(-1);          // [N, 1] f32 (sync_env.h:109)
        auto done = torch::from_blob(fe.done.data(), {n}, torch::kUInt8).to(torch::kFloat32).unsqueeze(-1);      // [N, 1] f32 0/1 (sync_env.h:110)
        std::vector<AnyMap> info(n);
        const float t = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - t_start).count();
        for (int64_t i = 0; i < n; i++) {
            info[i]["ale.lives"] = (int)fe.lives[i];
            if (fe.has_episode[i]) info[i]["episode"] = Result{fe.episode_return[i], fe.episode_length[i], t};
        }
        last_reward = reward; last_done = done; last_info = info;
        return {observation(), reward, done, std::move(info)};
    }

    // The reference's per-index calls on this stack.  reset(int index): SyncVecEnv::reset(int) re-runs ONE env's reset chain
    // (sync_env.h:128-139) and VecFrameStack::reset(int) feeds the single frame to StackedObservation::update (vec_frame_stack.cpp:34-37),
    // which — as written — shifts every env's stack and broadcasts that frame over the batch.  Default here: env `index` alone starts
    // over (its stack zeroed, the fresh frame newest) and ITS observation [nStack, 84, 84] is returned; with
    // AtariFrontEndOptions::reset_index_as_written the reference's behaviour is reproduced and the whole batch is returned.
    // step(index, action) is a stub in the reference that returns the buffered values (sync_env.h:141-157); so does this.
    torch::Tensor reset(int index) override {
        if (!did_reset) throw std::runtime_error("HipAtariVecEnv::reset(int index) before reset()");
        if (index < 0 || index >= numEnvs) throw std::runtime_error("HipAtariVecEnv::reset(int index): index out of range");
        fe.reset_index(index, stream);
        return opt.reset_index_as_written ? observation() : observation()[index];
    }
    StepT step(int index, torch::Tensor const&) noexcept override {
        if (!last_reward.defined() || index < 0 || index >= numEnvs) return {};
        return {observation()[index], last_reward[index], last_done[index], {last_info[index]}};
    }
    void render() const override {}
    std::vector<size_t> seed(size_t _seed) noexcept override {   // sync_env.h:74-83: env i <- seed + i (its NOOP-reset generator, noop_reset.h:46)
        fe.seed((uint64_t)_seed);
        std::vector<size_t> s(numEnvs);
        for (int i = 0; i < numEnvs; i++) s[i] = _seed + (size_t)i;
        return s;
    }

    AtariFrontEnd<Emu>& frontEnd() { return fe; }
    void setStream(hipStream_t s) { stream = s; }

private:
    torch::Tensor observation() const {
        return torch::from_blob((void*)fe.observation(), {(int64_t)numEnvs, (int64_t)opt.n_stack, (int64_t)opt.dst_h, (int64_t)opt.dst_w},
                                torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, opt.device));
    }
    AtariFrontEndOptions opt;
    AtariFrontEnd<Emu> fe;
    hipStream_t stream = nullptr;
    bool did_reset = false;
    torch::Tensor pending, last_reward, last_done;
    std::vector<AnyMap> last_info;
    std::chrono::high_resolution_clock::time_point t_start;
};

}  // namespace gym
