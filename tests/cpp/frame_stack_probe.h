// frame_stack_probe.h — one probe, two subjects (like wrapped_probe.h): gym::HipVecFrameStack (gym/hip_vec_frame_stack.h) over a
// scripted host VecEnv<false> of single-channel images, driven through the base-class interface as a trainer would
// (std::unique_ptr<gym::VecEnv<false>>); every returned observation is dumped for tests/test_cpp_host.py to compare with the
// oracle's frame stack.  Compiled (a) against the stand-alone mirror (tests/cpp/hip_vec_frame_stack_demo.cpp) and (b) against
// the REFERENCE's own base.h / vec_env_wrapper.h / spaces (oracle/ref_driver/ref_dropin_tu.cpp, mode "framestack").
// args: <mode: plain|as_written|norm|norm_as_written|clear|reset_index_as_written> <n_envs> <H> <W> <n_stack> <steps> <host|device> <out.bin>
// out.bin: (steps + 1) observations back to back (row 0 = reset) in the mode's dtype, then per step: reward[N] f32, done[N] f32.
// out.bin.ri: the reset(int index) leg — what reset(1) returned (env 1's stacked observation; the whole batch in the as-written
// form), then the whole batch after one more step.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace frame_stack_probe {

// frame t of env e: pixel p = a byte hash of (e, t, p); reward = t + e; done when (t + e) % 5 == 0.  No auto-reset semantics needed.
struct ScriptedImageVecEnv : gym::VecEnv<false> {
    int H, W; bool on_device; int64_t t = 0;
    torch::Tensor pending;
    ScriptedImageVecEnv(int n, int H, int W, bool on_device)
        : gym::VecEnv<false>(n, std::make_shared<gym::Box<uint8_t>>(gym::Range<uint8_t>{0, 255}, gym::Shape{{(int64_t)H, (int64_t)W, 1}, c10::kChar}),
                             std::make_shared<gym::Discrete>(2)), H(H), W(W), on_device(on_device) {}
    torch::Tensor frame(int64_t first = 0, int64_t count = -1, int64_t at = -1) const {
        if (count < 0) count = numEnvs;
        if (at < 0) at = t;
        auto x = torch::empty({count, (int64_t)H, (int64_t)W, 1}, torch::kUInt8);
        uint8_t* p = x.data_ptr<uint8_t>();
        for (int64_t e = first; e < first + count; e++)
            for (int64_t q = 0; q < (int64_t)H * W; q++) {
                uint32_t v = (uint32_t)(e * 2654435761u) ^ (uint32_t)(at * 40503u) ^ (uint32_t)(q * 2246822519u);
                v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
                p[(e - first) * H * W + q] = (uint8_t)(v >> 8);
            }
        return on_device ? x.to(torch::kCUDA) : x;
    }
    torch::Tensor reset() noexcept override { t = 0; return frame(); }
    void stepAsync(torch::Tensor const& a) noexcept override { pending = a; }
    StepT stepWait() noexcept override {
        t++;
        auto reward = torch::empty({(int64_t)numEnvs, 1}), done = torch::empty({(int64_t)numEnvs, 1});
        for (int64_t e = 0; e < numEnvs; e++) { reward[e][0] = float(t + e); done[e][0] = ((t + e) % 5 == 0) ? 1.f : 0.f; }
        return {frame(), reward, done, std::vector<gym::AnyMap>(numEnvs)};
    }
    torch::Tensor reset(int index) override { return frame(index, 1, 1000 + index)[0]; }   // SyncVecEnv::reset(int): ONE env's [H, W, 1] observation (sync_env.h:128-139)
    StepT step(int, torch::Tensor const&) noexcept override { return {}; }
    void render() const override {}
};

inline int run(char** a) {   // a[0..7] = the eight arguments above
    const std::string mode = a[0];
    const int n = std::atoi(a[1]), H = std::atoi(a[2]), W = std::atoi(a[3]), k = std::atoi(a[4]), steps = std::atoi(a[5]);
    const bool on_device = !std::strcmp(a[6], "device");
    gym::HipVecFrameStackOptions o;
    o.asWritten = mode == "as_written" || mode == "norm_as_written";
    o.clearOnDone = mode == "clear";
    o.resetIndexAsWritten = mode == "reset_index_as_written";
    auto inner = std::make_unique<ScriptedImageVecEnv>(n, H, W, on_device);
    std::unique_ptr<gym::VecEnv<false>> env;
    try {
        env = (mode == "norm" || mode == "norm_as_written") ? gym::HipVecFrameStack::makeNormAndPermute(std::move(inner), k, o)
                                                            : gym::HipVecFrameStack::make(std::move(inner), k, std::nullopt, o);
    } catch (std::exception const& e) { std::fprintf(stderr, "ctor: %s\n", e.what()); return 3; }
    auto sz = env->observationSpace()->size();
    std::printf("space %lld %lld %lld type %d nEnvs %d\n", (long long)sz[0], (long long)sz[1], (long long)sz[2], (int)env->observationSpace()->type(), env->nEnvs());
    FILE* f = std::fopen(a[7], "wb"); if (!f) return 2;
    auto dump = [&](torch::Tensor const& x) { auto c = x.to(torch::kCPU).contiguous(); std::fwrite(c.data_ptr(), c.element_size(), c.numel(), f); };
    auto obs = env->reset();
    if (!obs.defined()) return 4;
    std::printf("obs dtype %s dims", c10::toString(obs.scalar_type())); for (auto d : obs.sizes()) std::printf(" %lld", (long long)d); std::printf(" cuda %d\n", (int)obs.is_cuda());
    dump(obs);
    std::vector<torch::Tensor> rd;
    for (int s = 0; s < steps; s++) {
        auto r = env->step(torch::zeros({n}, torch::kInt64));
        if (!r.observation.defined()) return 5;
        dump(r.observation); rd.push_back(r.reward.clone()); rd.push_back(r.done.clone());
    }
    for (auto& t : rd) dump(t);
    std::fclose(f);
    // reset(int index): env 1 alone starts over (or, as written, one frame is broadcast over every stack), then the batch steps on
    if (n > 1) {
        FILE* g = std::fopen((std::string(a[7]) + ".ri").c_str(), "wb"); if (!g) return 2;
        auto dump2 = [&](torch::Tensor const& x) { auto c = x.to(torch::kCPU).contiguous(); std::fwrite(c.data_ptr(), c.element_size(), c.numel(), g); };
        try {
            auto ri = env->reset(1);
            std::printf("reset_index dims"); for (auto d : ri.sizes()) std::printf(" %lld", (long long)d); std::printf("\n");
            dump2(ri);
            auto r = env->step(torch::zeros({n}, torch::kInt64));
            if (!r.observation.defined()) return 5;
            dump2(r.observation);
        } catch (std::exception const& e) { std::printf("reset_index throws %s\n", e.what()); }
        std::fclose(g);
    }
    return 0;
}

}  // namespace frame_stack_probe
