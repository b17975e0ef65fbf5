// gym_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A plain C++ restatement of the algorithms on the ava6969/gym hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
// as the checker / reported CPU baseline.  The product (gym_amd + libgymhip.so) never links or calls it.
//
// PARITY PINNING.  Upstream holds no golden vectors or known-answer tests for this path,
// and its classic_control / vec_normalize / atari translation units cannot be compiled in this image
// without writing stand-ins for absent third-party headers (GLFW, box2d, OpenCV, Eigen, Boost), so
// rows A1-A3, A6, A7, A10-A13 are "parity unpinned": they are pinned only by (i) the reference-run
// values the survey recorded (seed-0 reset draw; A10: first two normalised rewards),
// (ii) libstdc++'s own std::mt19937 / uniform_real_distribution<float>, which this file calls directly
// exactly as the reference does, and (iii) hand-derived known answers in tests/.  Rows A4, A5, A8, A9
// (TimeLimit, ClipReward, Monitor, RunningMeanStd) ARE pinned against the reference's own headers
// compiled in place by oracle/Makefile (target _ref) when /root/reference is present; the same build pins the
// VecNormalize checkpoint format (torch::save/load of RunningMeanStd, both directions) and Monitor's CSV bytes
// (ResultWriter).  The Atari wrapper chain at the end of this file (row f4) is parity unpinned as well: the
// reference's chain needs ALE and OpenCV.
//
// Every function cites the reference file:line it restates.  Compile WITHOUT fast-math and WITHOUT FMA
// contraction (-ffp-contract=off, no -march=native): the reference is built by g++ for baseline x86-64.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace oracle {

// ---------------------------------------------------------------------------------------------
// A3  reset draws — classic_control/cartpole.cpp:19-31, env.h:60-68, cartpole.h:24-26,
//     wrappers/vec_env/sync_env.h:74-83 (env i of a vec env seeded s uses mt19937(s+i)).
// The reference calls libstdc++ directly; so do we (this is the same dependency, not a restatement).
// ---------------------------------------------------------------------------------------------
struct ResetRng {
    std::mt19937 dev;  // env.h:83 — default-constructed (seed 5489) until seed() is called
    void seed(uint64_t s) { dev.seed(s); }
    void draw4(double* s) {
        std::uniform_real_distribution<float> dist(-0.05f, 0.05f);  // cartpole.cpp:21
        // braced-init list => strictly left-to-right (cartpole.cpp:23-26); float widened to double state.
        s[0] = dist(dev);
        s[1] = dist(dev);
        s[2] = dist(dev);
        s[3] = dist(dev);
    }
};

struct StepOut {
    float reward;
    bool done;
    bool truncated;       // info["TimeLimit.truncated"], only meaningful when has_truncated
    bool has_truncated;
    bool has_episode;     // info["episode"] present (Monitor)
    float ep_r, ep_l;     // monitor.h:16-18 Result{r,l,(t)}
};

// Per-env interface — env.h:34-85 (reset/step), observation widened doubles of length obs_dim().
struct Env {
    virtual ~Env() = default;
    virtual int obs_dim() const = 0;
    virtual void seed(uint64_t) {}
    virtual void reset(double* obs) = 0;
    virtual StepOut step(int action, double* obs) = 0;
};

// ---------------------------------------------------------------------------------------------
// A1  CartPoleEnv — classic_control/cartpole.cpp:33-88, constants cartpole.h:30-39 (all float).
// ---------------------------------------------------------------------------------------------
struct CartPole final : Env {
    const float gravity = 9.8f, masscart = 1.0f, masspole = 0.1f;
    const float total_mass = masspole + masscart;         // float add (cartpole.h:33)
    const float length = 0.5f;
    const float polemass_length = masspole * length;      // float mul (cartpole.h:35)
    const float force_mag = 10.0f, tau = 0.02f;
    const float theta_thr = (float)(12 * 2 * M_PI / 360);  // double expr narrowed (cartpole.h:38)
    const float x_thr = 2.4f;
    double st[4] = {0, 0, 0, 0};  // x, x_dot, theta, theta_dot (vector<double>, cartpole.h:43)
    int time_step = 0;
    int beyond_done = -1;         // std::optional<int> m_StepsBeyondDone: -1 == nullopt
    ResetRng rng;
    // Optional external reset tape f32 [episodes][n][4] (replays a device-generated tape; not reference behaviour).
    const float* tape = nullptr; int64_t tape_n = 0, tape_eps = 0, tape_idx = 0, tape_ep = 0;

    int obs_dim() const override { return 4; }
    void seed(uint64_t s) override { rng.seed(s); }
    void reset(double* obs) override {  // cartpole.cpp:19-31
        if (tape) {
            const float* r = tape + ((tape_ep % tape_eps) * tape_n + tape_idx) * 4;
            for (int f = 0; f < 4; f++) st[f] = (double)r[f];
            tape_ep++;
        } else {
            rng.draw4(st);
        }
        beyond_done = -1;
        time_step = 0;
        std::memcpy(obs, st, sizeof st);
    }
    StepOut step(int action, double* obs) override {  // cartpole.cpp:33-88
        double x = st[0], x_dot = st[1], theta = st[2], theta_dot = st[3];
        float force = (action == 1) ? force_mag : -force_mag;                       // :37
        float cos_t = (float)std::cos(theta);                                       // :39
        float sin_t = (float)std::sin(theta);                                       // :40
        // :42-43  float + float*(double*double)*float  => evaluated in double, narrowed, float divide
        float temp = (float)((double)force + ((double)polemass_length * (theta_dot * theta_dot)) * (double)sin_t) / total_mass;
        // :45-46  numerator float; denominator: float term inside a double bracket, narrowed to float
        float num = gravity * sin_t - cos_t * temp;
        float inner = masspole * (cos_t * cos_t) / total_mass;
        float den = (float)((double)length * (4.0 / 3.0 - (double)inner));
        float theta_acc = num / den;
        float x_acc = temp - polemass_length * theta_acc * cos_t / total_mass;      // :48 all float, left-to-right
        // :50-54 euler branch (m_KinematicsIntegrator == "euler", cartpole.h:41)
        x = x + (double)tau * x_dot;                 // float*double -> double
        x_dot = x_dot + (double)(tau * x_acc);       // float*float -> float, widened for the add
        theta = theta + (double)tau * theta_dot;
        theta_dot = theta_dot + (double)(tau * theta_acc);
        st[0] = x; st[1] = x_dot; st[2] = theta; st[3] = theta_dot;
        bool done = time_step == 500 || x < (double)(-x_thr) || x > (double)x_thr ||
                    theta < (double)(-theta_thr) || theta > (double)theta_thr;    // :64-68
        float reward;
        if (!done) reward = 1.0f;
        else if (beyond_done < 0) { beyond_done = 0; reward = 1.0f; }
        else { beyond_done += 1; reward = 0.0f; }                                   // :70-85 (stderr warning omitted)
        time_step++;                                                                // :86
        std::memcpy(obs, st, sizeof st);
        return {reward, done, false, false, false, 0.f, 0.f};
    }
};

// ---------------------------------------------------------------------------------------------
// A2  MountainCarEnv — classic_control/mountain_car.cpp:17-42, constants mountain_car.h:30-37,50.
// ---------------------------------------------------------------------------------------------
struct MountainCar final : Env {
    const float MIN_POS = -1.2f, MAX_POS = 0.6f, MAX_VEL = 0.07f, GOAL_POS = 0.5f;  // float consts, widened on use
    double pos = -0.5, vel = 0.0;   // ctor calls reset() (mountain_car.cpp:14)
    int obs_dim() const override { return 2; }
    void reset(double* obs) override { pos = -0.5; vel = 0.0; obs[0] = pos; obs[1] = vel; }  // :38-42 (deterministic)
    StepOut step(int action, double* obs) override {
        vel += (action - 1) * 0.001 + std::cos(3 * pos) * -0.0025;   // :19 all double
        if (vel > MAX_VEL) vel = MAX_VEL;                            // :20-23
        if (vel < -MAX_VEL) vel = -MAX_VEL;
        pos += vel;                                                  // :26
        if (pos > MAX_POS) pos = MAX_POS;                            // :27-30
        if (pos < MIN_POS) pos = MIN_POS;
        if (pos == MIN_POS && vel < 0) vel = 0;                      // :32-33
        obs[0] = pos; obs[1] = vel;
        return {-1.f, pos >= GOAL_POS, false, false, false, 0.f, 0.f};  // :35, mountain_car.h:50
    }
};

// A scripted env used only to pin the wrapper restatements against oracle/_ref (reference wrappers compiled
// in place around the same scripted env): reward/done come from caller tapes.
struct Scripted final : Env {
    const float* rew; const uint8_t* dn; int64_t len; int64_t t = 0;
    Scripted(const float* r, const uint8_t* d, int64_t n) : rew(r), dn(d), len(n) {}
    int obs_dim() const override { return 1; }
    void reset(double* obs) override { obs[0] = (double)t; }
    StepOut step(int, double* obs) override {
        int64_t i = t % len; t++;
        obs[0] = (double)t;
        return {rew[i], dn[i] != 0, false, false, false, 0.f, 0.f};
    }
};

// ---------------------------------------------------------------------------------------------
// Wrappers — common/wrapper.h:85-151 (forwarding base).
// ---------------------------------------------------------------------------------------------
struct Wrapper : Env {
    std::unique_ptr<Env> inner;
    explicit Wrapper(std::unique_ptr<Env> e) : inner(std::move(e)) {}
    int obs_dim() const override { return inner->obs_dim(); }
    void seed(uint64_t s) override { inner->seed(s); }
    void reset(double* obs) override { inner->reset(obs); }
};

// A4  TimeLimit — wrappers/time_limit.h:28-41.
struct TimeLimit final : Wrapper {
    uint64_t max_steps; int64_t elapsed = 0;
    TimeLimit(std::unique_ptr<Env> e, uint64_t m) : Wrapper(std::move(e)), max_steps(m) {}
    void reset(double* obs) override { elapsed = 0; inner->reset(obs); }
    StepOut step(int a, double* obs) override {
        StepOut r = inner->step(a, obs);
        elapsed++;
        if ((uint64_t)elapsed >= max_steps) { r.has_truncated = true; r.truncated = !r.done; r.done = true; }
        return r;
    }
};

// A5  ClipRewardEnv — wrappers/clip_reward.h:22-24 via RewardWrapper::step common/wrapper.h:183-187.
struct ClipReward final : Wrapper {
    using Wrapper::Wrapper;
    static float clip(float r) { return r == 0 ? r : (r > 0.f ? 1.f : -1.f); }
    StepOut step(int a, double* obs) override { StepOut r = inner->step(a, obs); r.reward = clip(r.reward); return r; }
};

// A8  Monitor — wrappers/monitor.h:57-102.  The per-step reward vector is summed front-to-back in float
// (std::accumulate(..., 0.f), monitor.h:82); a running float sum in arrival order is the same arithmetic.
struct Monitor final : Wrapper {
    float ret = 0.f; float steps = 0.f; uint64_t total_steps = 0;
    using Wrapper::Wrapper;
    void reset(double* obs) override { inner->reset(obs); ret = 0.f; steps = 0.f; }
    StepOut step(int a, double* obs) override {
        StepOut r = inner->step(a, obs);
        ret = ret + r.reward; steps += 1.f;
        if (r.done) { r.has_episode = true; r.ep_r = ret; r.ep_l = steps; }
        total_steps++;
        return r;
    }
};

// A6  EpisodicLifeEnv — wrappers/atari_wrappers.cpp:35-53, episodic_life.h:13-36.  The emulator stays on the
// host; this restates only the life-loss => done rule over caller-provided (done, lives) lanes.
struct EpisodicLifeState { int lives = 0; bool was_real_done = true; };
inline bool episodic_life_step(EpisodicLifeState& s, bool done, int lives) {
    s.was_real_done = done;
    if (0 < lives && lives < s.lives) done = true;
    s.lives = lives;
    return done;
}

enum Kind { CARTPOLE = 0, MOUNTAINCAR = 1, SCRIPTED = 2 };
enum Flags { F_TIME_LIMIT = 1, F_CLIP_REWARD = 2, F_MONITOR = 4 };

// ---------------------------------------------------------------------------------------------
// A7  SyncVecEnv — wrappers/vec_env/sync_env.h:66-123: serial loop over per-env objects, auto-reset on done
// (terminal obs replaced by the reset obs, terminal reward/done/info kept), obs narrowed to float32
// (TensorAdapter::encode(vector<double>) -> torch default dtype, common/tensor_adapter.cpp:124-127).
// Wrapper order, innermost first: Env -> TimeLimit -> ClipReward -> Monitor.
// ---------------------------------------------------------------------------------------------
struct VecEnv {
    std::vector<std::unique_ptr<Env>> envs;
    int F = 0; int threads = 1;
    bool auto_reset = true;      // SyncVecEnv's m_AutoReset (sync_env.h:28,44,92)
    int64_t n() const { return (int64_t)envs.size(); }
    void seed(uint64_t s) { for (size_t i = 0; i < envs.size(); i++) envs[i]->seed(s + i); }  // sync_env.h:74-83
    void reset(float* obs, double* obs64) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < n(); i++) {
            double o[8]; envs[i]->reset(o);
            for (int f = 0; f < F; f++) { obs[i * F + f] = (float)o[f]; if (obs64) obs64[i * F + f] = o[f]; }
        }
    }
    void step(const int32_t* act, float* obs, double* obs64, float* rew, float* done, uint8_t* trunc,
              float* ep_r, float* ep_l) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < n(); i++) {
            double o[8];
            StepOut r = envs[i]->step(act[i], o);         // sync_env.h:91
            if (r.done && auto_reset) envs[i]->reset(o);  // :92-100 (m_AutoReset); otherwise the terminal observation is returned
            rew[i] = r.reward; done[i] = r.done ? 1.f : 0.f;          // :102-103
            if (trunc) trunc[i] = (r.has_truncated && r.truncated) ? 1 : 0;
            if (ep_r) ep_r[i] = r.has_episode ? r.ep_r : 0.f;
            if (ep_l) ep_l[i] = r.has_episode ? r.ep_l : 0.f;
            for (int f = 0; f < F; f++) { obs[i * F + f] = (float)o[f]; if (obs64) obs64[i * F + f] = o[f]; }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// A9  RunningMeanStd — common/normalizers/running_mean_std.h:16-20,42-66.  All state float32.
// Batch moments: torch x.mean(0) is a float32 sum (ATen cascade order) / n, x.var(0,false) is a Welford
// pass with double accumulators narrowed to float32.  Restated as exact (double) moments narrowed to
// float32: equal to ATen's var to rounding and within a few float ulp of ATen's mean
// (tests/test_oracle_normalizer.py measures both against torch's CPU kernels).
// ---------------------------------------------------------------------------------------------
struct RunningMeanStd {
    std::vector<float> mean, var; float count;
    explicit RunningMeanStd(int F, float eps = 1e-4f) : mean(F, 0.f), var(F, 1.f), count(eps) {}
    void update_from_moments(const float* bm, const float* bv, int64_t n) {  // :46-66, evaluation order as written
        float nf = (float)n;
        float tot = count + nf;                                   // :51
        for (size_t f = 0; f < mean.size(); f++) {
            float delta = bm[f] - mean[f];                        // :50
            float new_mean = mean[f] + delta * nf / tot;          // :53  (delta*n)/tot
            float ma = var[f] * count;                            // :54
            float mb = bv[f] * nf;                                // :55
            float m2 = ma + mb + delta * delta * count * nf / tot;  // :56  (ma+mb) + (((d*d)*count)*n)/tot
            var[f] = m2 / tot;                                    // :57
            mean[f] = new_mean;
        }
        count = tot;                                              // :65
    }
    static void batch_moments(const float* x, int64_t n, int F, float* bm, float* bv) {
        for (int f = 0; f < F; f++) {
            double s = 0; for (int64_t i = 0; i < n; i++) s += (double)x[i * F + f];
            double m = s / (double)n;
            double q = 0; for (int64_t i = 0; i < n; i++) { double d = (double)x[i * F + f] - m; q += d * d; }
            bm[f] = (float)m; bv[f] = (float)(q / (double)n);
        }
    }
    void update(const float* x, int64_t n) {                       // :42-44
        std::vector<float> bm(mean.size()), bv(mean.size());
        batch_moments(x, n, (int)mean.size(), bm.data(), bv.data());
        update_from_moments(bm.data(), bv.data(), n);
    }
};

// ---------------------------------------------------------------------------------------------
// A10  VecNormalize<false,true,true,true> (= TrainingVecNormRO, gym_.h:15-16) —
//      wrappers/vec_env/vec_normalize.h:23-59 (defaults), 118-134, 145-177.
// ---------------------------------------------------------------------------------------------
struct VecNormalize {
    VecEnv* venv; RunningMeanStd obs_rms, ret_rms; std::vector<float> returns, old_obs;
    float clip_obs, clip_rew, gamma, eps;
    bool training = true, norm_reward = true, norm_obs = true;   // template<bool dict, bool isTraining, bool normReward, bool normObs> (:20)
    VecNormalize(VecEnv* v, float co, float cr, float g, float e)
        : venv(v), obs_rms(v->F), ret_rms(1), returns(v->n(), 0.f), old_obs(v->n() * v->F), clip_obs(co), clip_rew(cr), gamma(g), eps(e) {}
    static float clipf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }  // torch::clip
    void normalize_obs(float* obs) {  // :118-124
        int F = venv->F; int64_t n = venv->n();
        std::memcpy(old_obs.data(), obs, sizeof(float) * n * F);
        for (int f = 0; f < F; f++) {
            float sd = std::sqrt(obs_rms.var[f] + eps);
            for (int64_t i = 0; i < n; i++) obs[i * F + f] = clipf((obs[i * F + f] - obs_rms.mean[f]) / sd, -clip_obs, clip_obs);
        }
    }
    void reset(float* obs) {  // :145-157
        venv->reset(obs, nullptr);
        std::fill(returns.begin(), returns.end(), 0.f);
        if (training && norm_obs) obs_rms.update(obs, venv->n());                // :149-151
        if (norm_obs) normalize_obs(obs);                                        // :153-155
        else std::memcpy(old_obs.data(), obs, sizeof(float) * venv->n() * venv->F);
    }
    void step(const int32_t* act, float* obs, float* rew, float* done, uint8_t* trunc, float* ep_r, float* ep_l) {  // :159-177
        int64_t n = venv->n();
        venv->step(act, obs, nullptr, rew, done, trunc, ep_r, ep_l);
        if (training && norm_obs) obs_rms.update(obs, n);                        // :162-164
        if (norm_obs) normalize_obs(obs);                                        // :166-167
        else std::memcpy(old_obs.data(), obs, sizeof(float) * n * venv->F);
        if (training && norm_reward) {                                           // :169-170
            for (int64_t i = 0; i < n; i++) returns[i] = returns[i] * gamma + rew[i];  // :131-132
            ret_rms.update(returns.data(), n);                                   // :133
        }
        if (norm_reward) {                                                       // :172-173
            float sd = std::sqrt(ret_rms.var[0] + eps);
            for (int64_t i = 0; i < n; i++) rew[i] = clipf(rew[i] / sd, -clip_rew, clip_rew);  // :126-129
        }
        for (int64_t i = 0; i < n; i++) if (done[i] != 0.f) returns[i] = 0.f;    // :175
    }
};

// ---------------------------------------------------------------------------------------------
// A11  MaxAndSkipEnv frame max — wrappers/atari_wrappers.cpp:30 (cv::max of the last two frames; intended
//      semantics per python_gym/atari_wrappers.py:139-150).
// ---------------------------------------------------------------------------------------------
inline void max2(const uint8_t* a, const uint8_t* b, uint8_t* out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = std::max(a[i], b[i]); }

// ---------------------------------------------------------------------------------------------
// A12  WarpFrameEnv::observation — wrappers/warp_frame.h:41-47: cv::cvtColor(RGB2GRAY) then
//      cv::resize(84x84, INTER_AREA).  The arithmetic lives in OpenCV 4.x (third-party, NOT vendored, version
//      unpinned: CMakeLists.txt:28-29) — restated from its published algorithm: 8-bit gray with 15-bit
//      fixed-point weights, and the generic area resampler (per-axis decimation tables, float accumulation,
//      round-half-even saturate).  PARITY UNPINNED against OpenCV itself; contract is +-1 LSB.
// ---------------------------------------------------------------------------------------------
inline uint8_t gray_u8(uint8_t r, uint8_t g, uint8_t b, int bits15) {
    if (bits15) return (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15);
    return (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
}
struct Tap { int si, di; float alpha; };
inline std::vector<Tap> area_table(int ssize, int dsize) {
    double scale = 1.0 / ((double)dsize / (double)ssize);   // cv::resize: inv_scale = dsize/ssize ; scale = 1/inv_scale
    std::vector<Tap> tab;
    for (int d = 0; d < dsize; d++) {
        double f1 = d * scale, f2 = f1 + scale;
        double cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) tab.push_back({s1 - 1, d, (float)((s1 - f1) / cell)});
        for (int s = s1; s < s2; s++) tab.push_back({s, d, (float)(1.0 / cell)});
        if (f2 - s2 > 1e-3) tab.push_back({s2, d, (float)(std::min(std::min(f2 - s2, 1.), cell) / cell)});
    }
    return tab;
}
inline uint8_t sat_u8(float v) {  // cv::saturate_cast<uchar>(float): round-half-even then clamp
    int iv = (int)std::nearbyintf(v);
    return (uint8_t)std::min(std::max(iv, 0), 255);
}
inline void area_resize_gray(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    std::vector<Tap> xt = area_table(sw, dw), yt = area_table(sh, dh);
    std::vector<float> buf(dw), sum(dw, 0.f);
    int prev = yt[0].di;
    for (size_t j = 0; j < yt.size(); j++) {
        float beta = yt[j].alpha; int dy = yt[j].di; const uint8_t* S = src + (size_t)yt[j].si * sw;
        std::fill(buf.begin(), buf.end(), 0.f);
        for (const Tap& t : xt) buf[t.di] += S[t.si] * t.alpha;
        if (dy != prev) {
            for (int x = 0; x < dw; x++) { dst[(size_t)prev * dw + x] = sat_u8(sum[x]); sum[x] = beta * buf[x]; }
            prev = dy;
        } else {
            for (int x = 0; x < dw; x++) sum[x] += beta * buf[x];
        }
    }
    for (int x = 0; x < dw; x++) dst[(size_t)prev * dw + x] = sat_u8(sum[x]);
}

}  // namespace oracle

// =============================================================================================
// C interface (ctypes)
// =============================================================================================
using namespace oracle;

struct OracleVec { VecEnv v; std::unique_ptr<VecNormalize> norm; };

extern "C" {

// reset tape [episodes][n_envs][4] float32: env i (global index env_offset+i) seeded seed+env_offset+i, episode k = draws 4k..4k+3.
void oracle_cartpole_reset_tape(uint64_t seed, int64_t env_offset, int64_t n_envs, int64_t episodes, float* out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < n_envs; i++) {
        ResetRng r; r.seed(seed + (uint64_t)(env_offset + i));
        for (int64_t k = 0; k < episodes; k++) {
            double s[4]; r.draw4(s);
            for (int f = 0; f < 4; f++) out[(k * n_envs + i) * 4 + f] = (float)s[f];
        }
    }
}

// First `count` raw outputs of std::mt19937(seed) (known-answer pin: default seed 5489 -> 10000th output 4123659995).
void oracle_mt19937_raw(uint64_t seed, int64_t count, uint32_t* out) {
    std::mt19937 g; g.seed(seed);
    for (int64_t i = 0; i < count; i++) out[i] = (uint32_t)g();
}

// cartpole.cpp:39-40 as the reference's x86-64 build evaluates it: glibc's double cos/sin, narrowed to float.
void oracle_glibc_sincos_f32(const double* theta, int64_t n, float* sin_out, float* cos_out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < n; i++) { sin_out[i] = (float)std::sin(theta[i]); cos_out[i] = (float)std::cos(theta[i]); }
}

// Host twin of the HIP kernel's pole_sincos (classic_control.hip in the product tree): the fdlibm k_sin / k_cos polynomials without
// argument reduction, the same operations in the same order (std::fma == the device's v_fma_f64, both IEEE exact), valid
// for |x| < 0.3.  It exists so that the ONE arithmetic difference between the device path and the reference — its sin/cos
// are not glibc's — can be quantified on the CPU over far more angles than a GPU test can carry: how often does
// float(pole_sincos(theta)) differ from float(glibc(theta))?  (tests/test_sincos_parity.py; a GPU test checks that the
// device function equals this twin bit for bit.)
static inline void pole_sincos_twin(double x, double* sn, double* cs) {
    const double z = x * x;
    const double rs = std::fma(z, std::fma(z, std::fma(z, std::fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                           -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    *sn = std::fma(z * x, std::fma(z, rs, -1.66666666666666324348e-01), x);
    const double rc = z * std::fma(z, std::fma(z, std::fma(z, std::fma(z, std::fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                                            2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    *cs = 1.0 - (0.5 * z - z * rc);
}
void oracle_pole_sincos(const double* theta, int64_t n, double* sin_out, double* cos_out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < n; i++) pole_sincos_twin(theta[i], &sin_out[i], &cos_out[i]);
}
// Counts, over theta_k = lo + (hi - lo) * (k + 0.5) / count for k in [k0, k0 + count_here) (a dense sweep generated in place, no
// array), how often the float-rounded twin differs from the float-rounded glibc value: out[0] sin mismatches, out[1] cos
// mismatches, out[2] largest |twin - glibc| in double ulps of the glibc value (sin), out[3] the same for cos.
void oracle_sincos_mismatch_sweep(double lo, double hi, int64_t count, int64_t k0, int64_t count_here, double* out, int threads) {
    int64_t ms = 0, mc = 0; double us = 0, uc = 0;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static) reduction(+ : ms, mc) reduction(max : us, uc)
    for (int64_t k = k0; k < k0 + count_here; k++) {
        const double th = lo + (hi - lo) * (((double)k + 0.5) / (double)count);
        double s, c; pole_sincos_twin(th, &s, &c);
        const double gs = std::sin(th), gc = std::cos(th);
        ms += (float)s != (float)gs; mc += (float)c != (float)gc;
        const double ulp_s = std::nextafter(std::fabs(gs), INFINITY) - std::fabs(gs), ulp_c = std::nextafter(std::fabs(gc), INFINITY) - std::fabs(gc);
        us = std::max(us, std::fabs(s - gs) / ulp_s); uc = std::max(uc, std::fabs(c - gc) / ulp_c);
    }
    out[0] = (double)ms; out[1] = (double)mc; out[2] = us; out[3] = uc;
}

void* oracle_vec_create(int kind, int64_t n_envs, int flags, int64_t max_episode_steps, int threads,
                        const float* script_rew, const uint8_t* script_done, int64_t script_len) {
    auto* h = new OracleVec();
    h->v.threads = threads > 0 ? threads : 1;
    for (int64_t i = 0; i < n_envs; i++) {
        std::unique_ptr<Env> e;
        if (kind == CARTPOLE) e = std::make_unique<CartPole>();
        else if (kind == MOUNTAINCAR) e = std::make_unique<MountainCar>();
        else e = std::make_unique<Scripted>(script_rew, script_done, script_len);
        if (flags & F_TIME_LIMIT) e = std::make_unique<TimeLimit>(std::move(e), (uint64_t)max_episode_steps);
        if (flags & F_CLIP_REWARD) e = std::make_unique<ClipReward>(std::move(e));
        if (flags & F_MONITOR) e = std::make_unique<Monitor>(std::move(e));
        h->v.envs.push_back(std::move(e));
    }
    h->v.F = h->v.envs.empty() ? 0 : h->v.envs[0]->obs_dim();
    return h;
}
void oracle_vec_destroy(void* p) { delete (OracleVec*)p; }
int oracle_vec_obs_dim(void* p) { return ((OracleVec*)p)->v.F; }
void oracle_vec_seed(void* p, uint64_t seed) { ((OracleVec*)p)->v.seed(seed); }
// Replay an external reset tape (CartPole only): env i's k-th reset reads tape[k][i][0..3].
void oracle_vec_set_reset_tape(void* p, const float* tape, int64_t episodes) {
    VecEnv& v = ((OracleVec*)p)->v;
    for (int64_t i = 0; i < v.n(); i++) {
        Env* e = v.envs[i].get();
        while (auto* w = dynamic_cast<Wrapper*>(e)) e = w->inner.get();
        if (auto* c = dynamic_cast<CartPole*>(e)) { c->tape = tape; c->tape_n = v.n(); c->tape_eps = episodes; c->tape_idx = i; c->tape_ep = 0; }
    }
}
void oracle_vec_reset(void* p, float* obs, double* obs64) { ((OracleVec*)p)->v.reset(obs, obs64); }
void oracle_vec_set_auto_reset(void* p, int on) { ((OracleVec*)p)->v.auto_reset = on != 0; }
// SyncVecEnv::reset(int index) (sync_env.h:128-139): env `index` alone is reset; its observation row is returned.
void oracle_vec_reset_index(void* p, int64_t index, float* obs_row, double* obs64_row) {
    VecEnv& v = ((OracleVec*)p)->v;
    double o[8]; v.envs[index]->reset(o);
    for (int f = 0; f < v.F; f++) { obs_row[f] = (float)o[f]; if (obs64_row) obs64_row[f] = o[f]; }
}
void oracle_vec_step(void* p, const int32_t* act, float* obs, double* obs64, float* rew, float* done, uint8_t* trunc,
                     float* ep_r, float* ep_l) {
    ((OracleVec*)p)->v.step(act, obs, obs64, rew, done, trunc, ep_r, ep_l);
}
// T steps from a u8 action tape [T][N]; records per-step streams (any pointer may be null).  Returns wall seconds.
double oracle_vec_rollout(void* p, const uint8_t* tape, int64_t T, float* obs /*[T][N][F]*/, float* rew, uint8_t* done,
                          uint8_t* trunc, float* final_obs, double* final_obs64) {
    VecEnv& v = ((OracleVec*)p)->v; int64_t n = v.n(); int F = v.F;
    std::vector<int32_t> act(n); std::vector<float> o(n * F), r(n), d(n); std::vector<uint8_t> tr(n);
    auto t0 = std::chrono::steady_clock::now();
    for (int64_t t = 0; t < T; t++) {
        for (int64_t i = 0; i < n; i++) act[i] = tape[t * n + i];
        v.step(act.data(), o.data(), (t == T - 1) ? final_obs64 : nullptr, r.data(), d.data(), tr.data(), nullptr, nullptr);
        if (obs) std::memcpy(obs + t * n * F, o.data(), sizeof(float) * n * F);
        if (rew) std::memcpy(rew + t * n, r.data(), sizeof(float) * n);
        if (done) for (int64_t i = 0; i < n; i++) done[t * n + i] = d[i] != 0.f;
        if (trunc) std::memcpy(trunc + t * n, tr.data(), n);
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (final_obs) std::memcpy(final_obs, o.data(), sizeof(float) * n * F);
    return dt;
}

// VecNormalize on top of the vec env.
void oracle_vecnorm_attach(void* p, float clip_obs, float clip_rew, float gamma, float eps) {
    auto* h = (OracleVec*)p; h->norm = std::make_unique<VecNormalize>(&h->v, clip_obs, clip_rew, gamma, eps);
}
// the class template's flags (vec_normalize.h:20) and a loaded checkpoint (:97-108)
void oracle_vecnorm_set_flags(void* p, int training, int norm_reward, int norm_obs) {
    auto* h = (OracleVec*)p; h->norm->training = training; h->norm->norm_reward = norm_reward; h->norm->norm_obs = norm_obs;
}
void oracle_vecnorm_set_stats(void* p, const float* in) {
    auto* h = (OracleVec*)p; int F = h->v.F; auto& n = *h->norm;
    for (int f = 0; f < F; f++) { n.obs_rms.mean[f] = in[f]; n.obs_rms.var[f] = in[F + f]; }
    n.obs_rms.count = in[2 * F]; n.ret_rms.mean[0] = in[2 * F + 1]; n.ret_rms.var[0] = in[2 * F + 2]; n.ret_rms.count = in[2 * F + 3];
}
void oracle_vecnorm_reset(void* p, float* obs, float* raw_obs) {
    auto* h = (OracleVec*)p; h->norm->reset(obs);
    if (raw_obs) std::memcpy(raw_obs, h->norm->old_obs.data(), sizeof(float) * h->norm->old_obs.size());
}
void oracle_vecnorm_step(void* p, const int32_t* act, float* obs, float* raw_obs, float* rew, float* done, uint8_t* trunc,
                         float* ep_r, float* ep_l) {
    auto* h = (OracleVec*)p; h->norm->step(act, obs, rew, done, trunc, ep_r, ep_l);
    if (raw_obs) std::memcpy(raw_obs, h->norm->old_obs.data(), sizeof(float) * h->norm->old_obs.size());
}
// stats out: obs mean[F], obs var[F], obs count, ret mean, ret var, ret count  (2F+4 floats)
void oracle_vecnorm_stats(void* p, float* out) {
    auto* h = (OracleVec*)p; int F = h->v.F; auto& n = *h->norm;
    for (int f = 0; f < F; f++) { out[f] = n.obs_rms.mean[f]; out[F + f] = n.obs_rms.var[f]; }
    out[2 * F] = n.obs_rms.count; out[2 * F + 1] = n.ret_rms.mean[0]; out[2 * F + 2] = n.ret_rms.var[0]; out[2 * F + 3] = n.ret_rms.count;
}
void oracle_vecnorm_returns(void* p, float* out) { auto* h = (OracleVec*)p; std::memcpy(out, h->norm->returns.data(), sizeof(float) * h->norm->returns.size()); }

// Stand-alone RunningMeanStd (A9).  state = mean[F], var[F], count (2F+1 floats), updated in place.
void oracle_rms_init(int F, float* state) { for (int f = 0; f < F; f++) { state[f] = 0.f; state[F + f] = 1.f; } state[2 * F] = 1e-4f; }
void oracle_rms_batch_moments(const float* x, int64_t n, int F, float* bm, float* bv) { RunningMeanStd::batch_moments(x, n, F, bm, bv); }
void oracle_rms_update_from_moments(int F, float* state, const float* bm, const float* bv, int64_t n) {
    RunningMeanStd r(F); std::copy(state, state + F, r.mean.begin()); std::copy(state + F, state + 2 * F, r.var.begin()); r.count = state[2 * F];
    r.update_from_moments(bm, bv, n);
    std::copy(r.mean.begin(), r.mean.end(), state); std::copy(r.var.begin(), r.var.end(), state + F); state[2 * F] = r.count;
}

// Host-emulated env post-processing (A5, A6, A8 + optional TimeLimit) over caller lanes; state arrays updated in place.
// Order innermost-first as AtariWrapper::make builds it (atari_wrappers.cpp:95-121): [TimeLimit] -> Monitor -> EpisodicLife -> ClipReward.
void oracle_host_env_post_step(int64_t n, const float* rew_in, const uint8_t* done_in, const int32_t* lives_in,
                               int32_t* st_lives, uint8_t* st_was_real_done, float* st_ret, float* st_len, int clip,
                               float* rew_out, uint8_t* done_out, uint8_t* has_ep, float* ep_r, float* ep_l) {
    for (int64_t i = 0; i < n; i++) {
        float r = rew_in[i]; bool d = done_in[i] != 0;
        st_ret[i] = st_ret[i] + r; st_len[i] += 1.f;                                // Monitor sees the raw summed reward
        has_ep[i] = d; ep_r[i] = d ? st_ret[i] : 0.f; ep_l[i] = d ? st_len[i] : 0.f;
        if (d) { st_ret[i] = 0.f; st_len[i] = 0.f; }                                // Monitor::reset on the real reset
        EpisodicLifeState s{st_lives[i], st_was_real_done[i] != 0};
        d = episodic_life_step(s, d, lives_in[i]);
        st_lives[i] = s.lives; st_was_real_done[i] = s.was_real_done;
        rew_out[i] = clip ? ClipReward::clip(r) : r; done_out[i] = d;
    }
}

// Atari preprocessing (A11, A12) for n frames pairs u8 [n][2][H][W][3] -> gray area-resized u8 [n][dh][dw].
void oracle_atari_preprocess(const uint8_t* frames, int64_t n, int H, int W, int dh, int dw, int gray15, uint8_t* out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t e = 0; e < n; e++) {
        const uint8_t* a = frames + (size_t)e * 2 * H * W * 3; const uint8_t* b = a + (size_t)H * W * 3;
        std::vector<uint8_t> mx((size_t)H * W * 3), g((size_t)H * W);
        max2(a, b, mx.data(), (int64_t)H * W * 3);
        for (int64_t p = 0; p < (int64_t)H * W; p++) g[p] = gray_u8(mx[3 * p], mx[3 * p + 1], mx[3 * p + 2], gray15);
        area_resize_gray(g.data(), H, W, out + (size_t)e * dh * dw, dh, dw);
    }
}
void oracle_gray(const uint8_t* rgb, int64_t npix, int gray15, uint8_t* out) { for (int64_t p = 0; p < npix; p++) out[p] = gray_u8(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], gray15); }
void oracle_area_resize(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) { area_resize_gray(src, sh, sw, dst, dh, dw); }
int oracle_area_table(int ssize, int dsize, int32_t* si, int32_t* di, float* alpha, int cap) {
    auto t = area_table(ssize, dsize);
    for (size_t k = 0; k < t.size() && (int)k < cap; k++) { si[k] = t[k].si; di[k] = t[k].di; alpha[k] = t[k].alpha; }
    return (int)t.size();
}

// A13  StackedObservation (intended semantics, python_gym/atari_wrappers.py:227-244 / SB3): channels-last
// stack u8 [n][H][W][k]: reset => zeros then newest in the last channel; update => shift left by one, newest last.
void oracle_framestack_reset(uint8_t* stack, const uint8_t* frame, int64_t n, int hw, int k) {
    std::memset(stack, 0, (size_t)n * hw * k);
    for (int64_t e = 0; e < n; e++) for (int p = 0; p < hw; p++) stack[((size_t)e * hw + p) * k + (k - 1)] = frame[(size_t)e * hw + p];
}
void oracle_framestack_update(uint8_t* stack, const uint8_t* frame, int64_t n, int hw, int k) {
    for (int64_t e = 0; e < n; e++) for (int p = 0; p < hw; p++) {
        uint8_t* s = stack + ((size_t)e * hw + p) * k;
        for (int c = 0; c + 1 < k; c++) s[c] = s[c + 1];
        s[k - 1] = frame[(size_t)e * hw + p];
    }
}
// The reference AS WRITTEN for a channels-last stack (stacked_observation.h:126-130 + 96-99): update() rolls the stack by -C
// along the last dim (:128), then calls stack(y, x, stackDim, ...) — passing stackDim = -1 where stack() expects a SIZE — so the
// write is y.index_put_({"...", Slice(-(-1))}, x), i.e. y[..., 1:] = x broadcast over the k-1 trailing channels.  Feeding
// frames 1,2,3,4 after a reset gives [3,4,4,4].  Restated literally: a left roll by one (channel 0 wraps to the end), then
// every channel >= 1 takes the new frame.  (k = 1: the slice is empty, the stack never changes after reset.)
void oracle_framestack_update_as_written(uint8_t* stack, const uint8_t* frame, int64_t n, int hw, int k) {
    for (int64_t e = 0; e < n; e++) for (int p = 0; p < hw; p++) {
        uint8_t* s = stack + ((size_t)e * hw + p) * k;
        const uint8_t first = s[0];
        for (int c = 0; c + 1 < k; c++) s[c] = s[c + 1];      // torch::roll(y, -1, -1)
        s[k - 1] = first;
        for (int c = 1; c < k; c++) s[c] = frame[(size_t)e * hw + p];   // y[..., 1:] = x
    }
}
// VecNormAndPermute::observation — wrappers/vec_env/vec_atari.cpp:70-81: NHWC -> NCHW, / 255 -> float32.
void oracle_norm_permute(const uint8_t* nhwc, int64_t n, int hw, int k, float* nchw) {
    for (int64_t e = 0; e < n; e++) for (int c = 0; c < k; c++) for (int p = 0; p < hw; p++)
        nchw[((size_t)e * k + c) * hw + p] = (float)nhwc[((size_t)e * hw + p) * k + c] / 255.f;
}

// The same call on the reference's actual buffer: the stacked tensor is int8 (kChar: spaces/util.h:60-63,
// stacked_observation.h:208), so `y / 255` (vec_atari.cpp:74-76) is a true division of SIGNED bytes in float32.
void oracle_norm_permute_as_written(const uint8_t* nhwc, int64_t n, int hw, int k, float* nchw) {
    for (int64_t e = 0; e < n; e++) for (int c = 0; c < k; c++) for (int p = 0; p < hw; p++)
        nchw[((size_t)e * k + c) * hw + p] = (float)(int8_t)nhwc[((size_t)e * hw + p) * k + c] / 255.f;
}

int oracle_num_threads() {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"

// =============================================================================================
// f4  The Atari wrapper chain around a host emulator, restated class by class the way the reference composes it
//     (AtariWrapper::make, wrappers/atari_wrappers.cpp:68-121):
//       AtariEnv -> NOOPResetEnv -> MaxAndSkipEnv -> Monitor -> EpisodicLifeEnv -> [FireResetEnv] -> WarpFrameEnv -> ClipRewardEnv
//     under SyncVecEnv's auto-reset (sync_env.h:92-100).  The emulator is tests/cpp/fake_emulator.h (a scripted stand-in
//     with ALE's call surface: no ROMs or ALE in this image).  Used to check the product's host Atari front-end.
//     PARITY UNPINNED: the reference's chain needs ALE + OpenCV to build; restated from source, with the INTENDED
//     MaxAndSkip semantics (two distinct frames).  Buffers that the reference leaves as empty cv::Mat
//     before the first full frame-skip group start as an all-index-0 screen here (the reference would fail in cv::max).
// =============================================================================================
#include "../tests/cpp/fake_emulator.h"

namespace oracle_atari {
using oracle::area_resize_gray;
using oracle::gray_u8;

using Frame = std::vector<uint8_t>;   // RGB bytes [H][W][3] below WarpFrame, gray [84][84] above
struct Resp { Frame obs; float reward = 0.f; bool done = false; int lives = 0; bool has_ep = false; float ep_r = 0.f, ep_l = 0.f; };
struct Env {
    virtual ~Env() = default;
    virtual Frame reset() = 0;
    virtual Resp step(int action) = 0;
};

// AtariEnv<true> (atari/atari_env.cpp:11-23,101-113) with getScreenRGB2 colouring (atari/ale_c_wrapper.cpp:28-41), frameSkip 1
struct BaseEnv : Env {
    FakeEmulator emu; const uint32_t* pal;
    BaseEnv(uint64_t game, const uint32_t* palette) : emu(game), pal(palette) {}
    Frame colour() const {
        Frame f((size_t)FakeEmulator::H * FakeEmulator::W * 3);
        const uint8_t* s = emu.screen();
        for (size_t p = 0; p < (size_t)FakeEmulator::H * FakeEmulator::W; p++) {
            const uint32_t z = pal[s[p]];
            f[3 * p] = (z >> 16) & 0xff; f[3 * p + 1] = (z >> 8) & 0xff; f[3 * p + 2] = z & 0xff;
        }
        return f;
    }
    Frame reset() override { emu.reset_game(); return colour(); }
    Resp step(int a) override { Resp r; r.reward = (float)emu.act(a); r.obs = colour(); r.done = emu.game_over(); r.lives = emu.lives(); return r; }
};
struct NoopResetEnv : Env {   // noop_reset.h:43-58
    Env* in; std::mt19937 dev; int noop_max;
    NoopResetEnv(Env* e, uint64_t seed, int nm) : in(e), dev((std::mt19937::result_type)seed), noop_max(nm) {}
    Frame reset() override {
        Resp resp; resp.obs = in->reset();
        int noops = std::uniform_int_distribution<int>(1, noop_max + 1)(dev);
        while (noops-- > 0) { resp = in->step(0); if (resp.done) resp.obs = in->reset(); }
        return resp.obs;
    }
    Resp step(int a) override { return in->step(a); }
};
struct MaxAndSkipEnv : Env {   // atari_wrappers.cpp:10-33
    Env* in; int skip; Frame buf[2];
    // AS WRITTEN (optional): AtariEnv::getObs returns `data` BY VALUE (atari_env.cpp:11-17) and a cv::Mat copy is a header
    // onto the same pixels, so m_ObsBuffer[0], m_ObsBuffer[1] and every later getObs result alias AtariEnv's one frame
    // buffer: when cv::max runs (atari_wrappers.cpp:30) both operands hold the pixels of the LAST getObs call.  The frames
    // here are value copies (the intended semantics); the aliased result is restated by re-reading the base env's screen.
    const BaseEnv* aliased = nullptr;
    MaxAndSkipEnv(Env* e, int s, const Frame& blank) : in(e), skip(s) { buf[0] = blank; buf[1] = blank; }
    Frame reset() override { return in->reset(); }
    Resp step(int a) override {
        float total = 0.f; Resp resp; int t = 0;
        while (t < skip) {
            resp = in->step(a);
            if (t == skip - 2) buf[0] = std::move(resp.obs); else if (t == skip - 1) buf[1] = std::move(resp.obs);
            total += resp.reward;
            if (resp.done) break;
            t++;
        }
        if (aliased) { resp.obs = aliased->colour(); resp.reward = total; return resp; }   // max(x, x) of the one shared buffer
        resp.obs.resize(buf[0].size());
        for (size_t k = 0; k < buf[0].size(); k++) resp.obs[k] = std::max(buf[0][k], buf[1][k]);   // cv::max
        resp.reward = total;
        return resp;
    }
};
struct MonitorEnv : Env {   // monitor.h:57-102
    Env* in; std::vector<float> rewards; int steps = 0;
    explicit MonitorEnv(Env* e) : in(e) {}
    Frame reset() override { Frame f = in->reset(); rewards.clear(); steps = 0; return f; }
    Resp step(int a) override {
        Resp r = in->step(a);
        rewards.push_back(r.reward); steps++;
        if (r.done) { r.has_ep = true; r.ep_r = std::accumulate(rewards.begin(), rewards.end(), 0.f); r.ep_l = (float)steps; }
        return r;
    }
};
struct EpisodicLifeEnv : Env {   // atari_wrappers.cpp:35-53, episodic_life.h:16-17
    Env* in; BaseEnv* parent; int lives = 0; bool was_real_done = true;
    EpisodicLifeEnv(Env* e, BaseEnv* p) : in(e), parent(p) {}
    Resp step(int a) override {
        Resp r = in->step(a);
        was_real_done = r.done;
        if (0 < r.lives && r.lives < lives) r.done = true;
        lives = r.lives;
        return r;
    }
    Frame reset() override {
        Frame out = was_real_done ? in->reset() : in->step(0).obs;
        lives = parent->emu.lives();
        return out;
    }
};
struct FireResetEnv : Env {   // atari_wrappers.cpp:55-67
    Env* in;
    explicit FireResetEnv(Env* e) : in(e) {}
    Frame reset() override {
        in->reset();
        Resp r = in->step(1);
        if (r.done) in->reset();
        r = in->step(2);
        if (r.done) in->reset();
        return r.obs;
    }
    Resp step(int a) override { return in->step(a); }
};
struct WarpFrameEnv : Env {   // warp_frame.h:41-47
    Env* in;
    explicit WarpFrameEnv(Env* e) : in(e) {}
    static Frame warp(const Frame& rgb) {
        const int H = FakeEmulator::H, W = FakeEmulator::W;
        std::vector<uint8_t> g((size_t)H * W);
        for (size_t p = 0; p < g.size(); p++) g[p] = gray_u8(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], 1);
        Frame out(84 * 84);
        area_resize_gray(g.data(), H, W, out.data(), 84, 84);
        return out;
    }
    Frame reset() override { return warp(in->reset()); }
    Resp step(int a) override { Resp r = in->step(a); r.obs = warp(r.obs); return r; }
};
struct ClipRewardEnv : Env {   // clip_reward.h:22-24
    Env* in;
    explicit ClipRewardEnv(Env* e) : in(e) {}
    Frame reset() override { return in->reset(); }
    Resp step(int a) override { Resp r = in->step(a); r.reward = r.reward == 0.f ? r.reward : (r.reward > 0.f ? 1.f : -1.f); return r; }
};

struct Chain {
    std::vector<std::unique_ptr<Env>> owned;
    Env* top = nullptr;
    BaseEnv* base = nullptr; MaxAndSkipEnv* maxskip = nullptr;
    template <class T, class... A> T* add(A&&... a) { auto p = std::make_unique<T>(std::forward<A>(a)...); T* raw = p.get(); owned.push_back(std::move(p)); return raw; }
};
struct VecChain { std::vector<Chain> envs; std::vector<uint32_t> palette; };
}  // namespace oracle_atari

extern "C" {

void* oracle_atari_chain_create(int64_t n, uint64_t seed, uint64_t game_base, int frame_skip, int noop_max, int life_loss, int fire, int clip,
                                const uint32_t* zrgb256) {
    using oracle_atari::BaseEnv; using oracle_atari::Chain; using oracle_atari::Frame; using oracle_atari::VecChain;
    auto* v = new VecChain();
    v->palette.assign(zrgb256, zrgb256 + 256);
    v->envs.resize(n);
    for (int64_t i = 0; i < n; i++) {
        Chain& c = v->envs[i];
        BaseEnv* base = c.add<BaseEnv>(game_base + (uint64_t)i, v->palette.data());
        Frame blank((size_t)FakeEmulator::H * FakeEmulator::W * 3);
        for (size_t p = 0; p < blank.size() / 3; p++) { const uint32_t z = v->palette[0]; blank[3 * p] = (z >> 16) & 0xff; blank[3 * p + 1] = (z >> 8) & 0xff; blank[3 * p + 2] = z & 0xff; }
        oracle_atari::Env* e = c.add<oracle_atari::NoopResetEnv>(base, seed + (uint64_t)i, noop_max);
        c.base = base;
        c.maxskip = c.add<oracle_atari::MaxAndSkipEnv>(e, frame_skip, blank);
        e = c.maxskip;
        e = c.add<oracle_atari::MonitorEnv>(e);
        if (life_loss) e = c.add<oracle_atari::EpisodicLifeEnv>(e, base);
        if (fire) e = c.add<oracle_atari::FireResetEnv>(e);
        e = c.add<oracle_atari::WarpFrameEnv>(e);
        if (clip) e = c.add<oracle_atari::ClipRewardEnv>(e);
        c.top = e;
    }
    return v;
}
void oracle_atari_chain_destroy(void* h) { delete (oracle_atari::VecChain*)h; }
// 1 = MaxAndSkipEnv as written (its two buffers alias AtariEnv's frame buffer: the observation is the last emulated frame)
void oracle_atari_chain_set_max_as_written(void* h, int on) { for (auto& c : ((oracle_atari::VecChain*)h)->envs) c.maxskip->aliased = on ? c.base : nullptr; }
// SyncVecEnv::reset (sync_env.h:66-72): frames84 u8 [n][84*84]
void oracle_atari_chain_reset(void* h, uint8_t* frames84) {
    auto* v = (oracle_atari::VecChain*)h;
    for (size_t i = 0; i < v->envs.size(); i++) { auto f = v->envs[i].top->reset(); std::memcpy(frames84 + i * 7056, f.data(), 7056); }
}
// SyncVecEnv::reset(int index) (sync_env.h:128-139): ONE env runs its outermost reset; frame84 u8 [84*84]
void oracle_atari_chain_reset_index(void* h, int64_t index, uint8_t* frame84) {
    auto* v = (oracle_atari::VecChain*)h;
    auto f = v->envs[(size_t)index].top->reset();
    std::memcpy(frame84, f.data(), 7056);
}
// SyncVecEnv::stepWait with auto-reset (sync_env.h:89-105)
void oracle_atari_chain_step(void* h, const int32_t* actions, uint8_t* frames84, float* reward, uint8_t* done, uint8_t* has_ep, float* ep_r, float* ep_l,
                             int32_t* lives) {
    auto* v = (oracle_atari::VecChain*)h;
    for (size_t i = 0; i < v->envs.size(); i++) {
        auto r = v->envs[i].top->step(actions[i]);
        if (r.done) r.obs = v->envs[i].top->reset();
        std::memcpy(frames84 + i * 7056, r.obs.data(), 7056);
        reward[i] = r.reward; done[i] = r.done; has_ep[i] = r.has_ep; ep_r[i] = r.ep_r; ep_l[i] = r.ep_l; lives[i] = r.lives;
    }
}

}  // extern "C"
