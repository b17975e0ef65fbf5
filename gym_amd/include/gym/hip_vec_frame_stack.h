// hip_vec_frame_stack.h — gym::HipVecFrameStack : gym::VecEnvWrapper<false>, the drop-in for the reference's VecFrameStack
// wrapper ON ITS OWN (wrappers/vec_env/vec_frame_stack.{h,cpp}:10-32 over common/stacked_observation.h:96-130,186-210), and —
// with `normAndPermute` — for the pair a trainer composes as
//
//     VecNormAndPermute::make( VecFrameStack::make( <any VecEnv<false> of single-channel uint8 images>, nStack ) )
//     (wrappers/vec_env/vec_atari.cpp:70-81,123-125; vec_frame_stack.cpp:50-54)
//
// The inner env is ANY VecEnv<false> whose observation space is a 3-D channels-last image space {H, W, 1} of bytes (WarpFrame's
// layout, warp_frame.h:28) — a host SyncVecEnv of image envs, a GPU env, anything; its batch [N, H, W, 1] (or [N, H, W]) may
// live on the host or on the GPU.  The stack is the device ring of gymhip_atari (one plane copy per step: gymhip_atari_push,
// instead of the reference's roll + index_put_ of the whole [N, H, W, k] buffer), and the returned observation is a view of a
// device buffer that the next step overwrites:
//     plain:            uint8 [N, H, W, nStack]  (as-written view: int8, the reference's kChar buffer)   stacked_observation.h:96-99
//     normAndPermute:   float [N, nStack, H, W] = x / 255                                                vec_atari.cpp:74-76
// observationSpace(): Box<uint8_t>(low, high, {H, W, nStack}) (stacked_observation.h:54-63), resp. Box<float>(0, 1, {nStack, H, W})
// (vec_atari.cpp:54-58).  reward / done / info pass through untouched (vec_frame_stack.cpp:27-32).
// Differences from the reference, all at its defects: channels-first spaces and a `channelOrder` argument are
// rejected with std::runtime_error (the reference corrupts the heap / always throws); reset(int index) resets THAT env's stack
// (zeros + its fresh frame, as reset() does for all) and returns its stacked observation — `resetIndexAsWritten` selects what the
// reference's code does instead: every env's stack shifts and the one frame is broadcast over the batch (vec_frame_stack.cpp:34-37);
// the default view is the intended shift — `asWritten` selects the reference's [prev, new, new, ...] int8 output instead.
// Streams: the inner env's batch is staged on torch's CURRENT stream and, unless setStream() names another one, the push and the
// materialise kernels run on that same stream; with a stream of your own the two are ordered by events in both directions.
// Inside the reference tree: #define GYMHIP_USE_REFERENCE_HEADERS and include the reference's base.h / vec_env_wrapper.h / spaces first.
#pragma once
#include <hip/hip_runtime_api.h>
#include <c10/hip/HIPStream.h>

#include <cstdio>
#include <optional>
#include <stdexcept>
#include <string>

#ifdef GYMHIP_USE_REFERENCE_HEADERS
#include "spaces/box.h"
#include "wrappers/vec_env/vec_env_wrapper.h"
#else
#include "gym/vec_env.h"
#endif
#include "gymhip.h"

namespace gym {

struct HipVecFrameStackOptions {
    bool normAndPermute = false;   // fuse VecNormAndPermute onto the stack
    bool asWritten = false;        // the reference's output bug for bug (gymhip_atari_set_stack_view)
    bool clearOnDone = false;      // SB3's behaviour; the reference never clears
    bool resetIndexAsWritten = false;   // reset(int index) as the reference's code has it: all stacks shift, the one frame is broadcast
    int device = 0;
};

class HipVecFrameStack : public VecEnvWrapper<false> {
public:
    HipVecFrameStack(std::unique_ptr<VecEnv<false>> inner, int nStack, std::optional<std::string> const& channelOrder = std::nullopt,
                     HipVecFrameStackOptions o = {})
        : VecEnvWrapper<false>(std::move(inner)), opt(o), k(nStack) {
        if (channelOrder && *channelOrder != "last")
            throw std::runtime_error("HipVecFrameStack: only channels-last image spaces are supported (the reference throws for every channelOrder, stacked_observation.h:196-199)");
        auto inner_space = this->venv->observationSpace();
        auto sz = inner_space ? inner_space->size() : std::vector<int64_t>{};
        if (sz.size() != 3 || sz[2] != 1 || sz[0] <= 0 || sz[1] <= 0)
            throw std::runtime_error("HipVecFrameStack: the inner observation space must be a channels-last single-channel image {H, W, 1}");
        if (nStack <= 0) throw std::runtime_error("HipVecFrameStack: nStack must be positive");
        H = (int)sz[0]; W = (int)sz[1];
        uint8_t lo = 0, hi = 255;
        if (auto* b = inner_space->as<Box<uint8_t>>()) { lo = b->getRange()[0].low; hi = b->getRange()[0].high; }
        if (opt.normAndPermute)
            this->m_ObservationSpace = std::make_shared<Box<float>>(Range<float>{0.f, 1.f}, Shape{{(int64_t)k, (int64_t)H, (int64_t)W}, c10::kFloat});
        else
            this->m_ObservationSpace = std::make_shared<Box<uint8_t>>(Range<uint8_t>{lo, hi}, Shape{{(int64_t)H, (int64_t)W, (int64_t)k}, c10::kChar});   // make_shape<uint8_t> reports kChar (spaces/util.h:60-63)
        if (hipSetDevice(opt.device) != hipSuccess) throw std::runtime_error("HipVecFrameStack: hipSetDevice failed");
        gymhip_atari_config c{};
        c.struct_size = sizeof c; c.n_envs = this->numEnvs; c.src_h = c.dst_h = H; c.src_w = c.dst_w = W;   // no resampling: planes are pushed as they are
        c.n_stack = k; c.gray_bits = 15; c.clear_on_done = opt.clearOnDone ? 1 : 0;
        if (gymhip_atari_create(&c, opt.device, &h) != GYMHIP_OK) throw std::runtime_error(std::string("HipVecFrameStack: ") + gymhip_last_error());
        if (opt.asWritten && gymhip_atari_set_stack_view(h, GYMHIP_STACK_AS_WRITTEN) != GYMHIP_OK) {
            std::string msg = gymhip_last_error(); gymhip_atari_destroy(h); h = nullptr;
            throw std::runtime_error("HipVecFrameStack: " + msg);
        }
        auto dev = torch::TensorOptions().device(torch::kCUDA, opt.device);
        if (opt.normAndPermute) out = torch::empty({(int64_t)this->numEnvs, (int64_t)k, (int64_t)H, (int64_t)W}, dev.dtype(torch::kFloat32));
        else out = torch::empty({(int64_t)this->numEnvs, (int64_t)H, (int64_t)W, (int64_t)k}, dev.dtype(opt.asWritten ? torch::kInt8 : torch::kUInt8));
    }
    ~HipVecFrameStack() override { if (h) gymhip_atari_destroy(h); if (staged_ev) (void)hipEventDestroy(staged_ev); if (pushed_ev) (void)hipEventDestroy(pushed_ev); }
    HipVecFrameStack(HipVecFrameStack const&) = delete;
    HipVecFrameStack& operator=(HipVecFrameStack const&) = delete;

    static std::unique_ptr<VecEnv<false>> make(std::unique_ptr<VecEnv<false>> x, int nStack = 4, std::optional<std::string> const& channelOrder = std::nullopt,
                                               HipVecFrameStackOptions o = {}) {   // vec_frame_stack.cpp:50-54
        return std::make_unique<HipVecFrameStack>(std::move(x), nStack, channelOrder, o);
    }
    // VecNormAndPermute::make(VecFrameStack::make(x, nStack)) as one object
    static std::unique_ptr<VecEnv<false>> makeNormAndPermute(std::unique_ptr<VecEnv<false>> x, int nStack = 4, HipVecFrameStackOptions o = {}) {
        o.normAndPermute = true;
        return std::make_unique<HipVecFrameStack>(std::move(x), nStack, std::nullopt, o);
    }

    torch::Tensor reset() noexcept override {   // vec_frame_stack.cpp:22-25
        try {
            torch::Tensor planes = stage(this->venv->reset(), this->numEnvs);
            if (!planes.defined() || gymhip_atari_push_reset(h, planes.data_ptr<uint8_t>(), launchStream()) != GYMHIP_OK) return complain("reset");
            did_reset = true;
            return materialize();
        } catch (std::exception const& e) { std::fprintf(stderr, "HipVecFrameStack::reset: %s\n", e.what()); return {}; }
    }

    StepT stepWait() noexcept override {   // vec_frame_stack.cpp:27-32
        auto stepData = this->venv->stepWait();
        try {
            if (!did_reset) { std::fprintf(stderr, "HipVecFrameStack::stepWait before reset()\n"); return {}; }
            torch::Tensor planes = stage(stepData.observation, this->numEnvs), done;
            const uint8_t* dp = nullptr;
            if (opt.clearOnDone && stepData.done.defined()) {
                done = stepData.done.reshape({-1}).to(torch::kCUDA, torch::kUInt8, /*non_blocking=*/false).contiguous();
                if (done.numel() == this->numEnvs) dp = done.data_ptr<uint8_t>();
            }
            if (!planes.defined() || gymhip_atari_push(h, planes.data_ptr<uint8_t>(), dp, launchStream()) != GYMHIP_OK) { complain("stepWait"); return {}; }
            stepData.observation = materialize();
        } catch (std::exception const& e) { std::fprintf(stderr, "HipVecFrameStack::stepWait: %s\n", e.what()); return {}; }
        return stepData;
    }

    // sync_env.h:128-139 under vec_frame_stack.cpp:34-37: the inner env resets ONE env and returns its [H, W, 1] image.  Default: that
    // env's stack starts over (zeros + the fresh frame) and its stacked observation is returned, the other envs are untouched.
    // resetIndexAsWritten: the reference as it stands — the whole batch's stacks shift, the frame lands in every env's newest slot,
    // and the whole stacked batch is returned.
    torch::Tensor reset(int index) override {
        if (!did_reset) throw std::runtime_error("HipVecFrameStack::reset(int index) before reset()");
        if (index < 0 || index >= this->numEnvs) throw std::runtime_error("HipVecFrameStack::reset(int index): index out of range");
        torch::Tensor plane = stage(this->venv->reset(index), 1);
        if (!plane.defined() || gymhip_atari_push_index(h, index, plane.data_ptr<uint8_t>(), opt.resetIndexAsWritten ? 1 : 0, launchStream()) != GYMHIP_OK)
            throw std::runtime_error(std::string("HipVecFrameStack::reset(int index): ") + gymhip_last_error());
        torch::Tensor all = materialize();
        if (!all.defined()) throw std::runtime_error(std::string("HipVecFrameStack::reset(int index): ") + gymhip_last_error());
        return opt.resetIndexAsWritten ? all : all[index];
    }
    StepT step(int index, torch::Tensor const& action) noexcept override {   // the reference updates the whole stack with ONE env's frame here; we return the inner env's buffered row over the current stack
        auto stepData = this->venv->step(index, action);
        if (out.defined() && index >= 0 && index < this->numEnvs) stepData.observation = out[index];
        return stepData;
    }

    void setStream(hipStream_t s) { stream = s; own_stream = true; }
    gymhip_atari* handle() { return h; }

private:
    // the inner env's batch -> contiguous device bytes [N][H*W] (kept alive until the push that reads it has been issued on `stream`)
    torch::Tensor stage(torch::Tensor const& obs, int64_t n_planes) {
        if (!obs.defined()) return {};
        const int64_t want = n_planes * H * W;
        if (obs.numel() != want || (obs.scalar_type() != torch::kUInt8 && obs.scalar_type() != torch::kInt8)) {
            std::fprintf(stderr, "HipVecFrameStack: expected a byte tensor of %lld elements ([N, %d, %d, 1]), got %lld of dtype %s\n", (long long)want, H, W,
                         (long long)obs.numel(), c10::toString(obs.scalar_type()));
            return {};
        }
        torch::Tensor t = obs.scalar_type() == torch::kInt8 ? obs.view(torch::kUInt8) : obs;
        // The copy runs on torch's current stream.  With a launch stream of our own: (1) the tensor the PREVIOUS push read is released only
        // after torch's stream has been made to wait for that push (pushed_ev, recorded in materialize()), so its memory cannot be
        // reused under it; (2) our stream waits for this copy before the next push reads it.
        hipStream_t ts = c10::hip::getCurrentHIPStream((c10::DeviceIndex)opt.device).stream();
        if (own_stream && stream != ts && pushed_ev) (void)hipStreamWaitEvent(ts, pushed_ev, 0);
        staged = t.to(torch::Device(torch::kCUDA, opt.device)).contiguous();
        if (own_stream && stream != ts) {
            if (!staged_ev) (void)hipEventCreateWithFlags(&staged_ev, hipEventDisableTiming);
            (void)hipEventRecord(staged_ev, ts);
            (void)hipStreamWaitEvent(stream, staged_ev, 0);
        }
        return staged;
    }
    // the stream the push / materialise kernels run on: the caller's (setStream) or torch's current one — the one `staged` was produced on
    hipStream_t launchStream() const { return own_stream ? stream : c10::hip::getCurrentHIPStream((c10::DeviceIndex)opt.device).stream(); }
    torch::Tensor materialize() {
        hipStream_t ls = launchStream();
        const int rc = opt.normAndPermute ? gymhip_atari_materialize_nchw_f32(h, out.data_ptr<float>(), ls)
                                          : gymhip_atari_materialize_nhwc(h, (uint8_t*)out.data_ptr(), ls);
        if (rc != GYMHIP_OK) return complain("materialize");
        if (own_stream) {
            if (!pushed_ev) (void)hipEventCreateWithFlags(&pushed_ev, hipEventDisableTiming);
            (void)hipEventRecord(pushed_ev, ls);
        }
        return out;
    }
    torch::Tensor complain(const char* where) {
        std::fprintf(stderr, "HipVecFrameStack::%s failed: %s\n", where, gymhip_last_error());
        return {};
    }

    HipVecFrameStackOptions opt;
    int k, H = 0, W = 0;
    gymhip_atari* h = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;       // setStream() was called: push / materialise run there, ordered against torch's stream by the two events
    hipEvent_t staged_ev = nullptr, pushed_ev = nullptr;
    bool did_reset = false;
    torch::Tensor out, staged;
};

}  // namespace gym
