// gymhip_internal.h — kernel parameter blocks shared by the engine's translation units (not part of the C ABI).
#pragma once
#include <cstdint>
#include <string>

namespace gymhip {

extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);

// Per-env device arrays (N = n_envs, F = obs_dim).  See include/gymhip.h gymhip_buffers for meanings.
struct StepBuffers {
    void* state;             // LOCKSTEP: i32 [N][F] split-double residuals; F32: == obs
    float* obs;              // f32 [N][F]
    float* reward;           // f32 [N]
    float* done;             // f32 [N]
    uint32_t* counter;       // u32 [N] packed step counter / finished-episode record / tape cursor (gymhip.h)
    float* returns;          // f32 [N] or null
};

// Where reset states come from, as DEVICE state (one block per engine, in its workspace): kernels read it through a pointer, so a
// step recorded into a HIP graph keeps following the engine when the host re-seeds it or replaces / extends its reset tape.
struct ResetSource {
    const float* tape;       // TAPE: f32 [tape_episodes][N][4], host-drawn or device-generated; else null
    int64_t tape_episodes;
    uint64_t seed;           // COUNTER_RNG: keys the stateless draw together with the global env index and the env's episode index
};

struct StepParams {
    int64_t n;
    int64_t env_offset;
    const void* actions;
    const ResetSource* src;  // device
    uint32_t* mt_state;      // MT19937: u32 [N][624], env i's std::mt19937 state array (libstdc++ _M_x), twisted in place; else null
    const float* stats;      // running statistics, two ping-pong slots of 16 floats (+ the public copy), or null; the current slot is *tick_w & 1
    const int32_t* tick_w;   // device-side slot hand-off (see normalize_kernel): kernels that only READ the statistics take the slot from
    int32_t* tick_r;         // tick_w and copy it into tick_r; the kernel that PUBLISHES new statistics reads tick_r and bumps tick_w
    double* partials;        // [grid][16]
    double* ep_partials;     // [grid][4] cumulative episodes / sum return / sum length
    int32_t* error_flag;
    uint32_t wrappers;
    int32_t max_episode_steps;
    int32_t reset_mode;
    float gamma;
    int32_t pad_rows;        // ROWS exchange form: the all-reduce payload is partial rows [0, pad_rows); the step / reset kernel writes this
                             // rank's env count into row 0's last column and zeroes every row its own blocks do not own (0 = off)
    int32_t lanes;           // threads of each workgroup that own envs (a multiple of 64, <= the block size): small env counts are spread
                             // over more workgroups than n / block would give, so that every CU gets one (the rest of the block idles)
    StepBuffers buf;
};

struct NormParams {
    int64_t n;
    float* stats;            // slots 0 / 1 = ping-pong, slot 2 = the public copy (16 floats each)
    const int32_t* tick_r;   // the current slot is *tick_r & 1 (the preceding step / reset / batch-moments kernel copied it from tick_w)
    int32_t* tick_w;         // block 0 publishes into the other slot and then sets *tick_w = *tick_r + 1
    const double* moments;   // reduced batch moments (split step), read when `partials` is null
    const double* partials;  // unsplit step: the step kernel's per-block rows [rows][16], folded in the kernel's prologue
    int rows;
    double n_local;
    bool n_from_rows;        // ROWS exchange form: the batch count is the folded total of the rows' last column (summed over ranks)
    double* moments_out;     // unsplit step: block 0 publishes the folded moments here (the engine's `moments` buffer)
    double* ep_totals_global;
    const float* obs;
    float* obs_norm;
    float* reward;
    float clip_obs, clip_reward, epsilon;
    bool norm_obs, norm_reward, update_obs, update_ret;
};

}  // namespace gymhip
