/* gymhip.h — C ABI of the MI355X batched environment engine (libgymhip.so).
 *
 * The reference (ava6969/gym) has no FFI: its boundary is the in-process C++ class templates gym::VecEnv<dict>
 * and its wrappers.  This ABI is what a maintainer's gym::VecEnv<false> subclass binds instead of running the
 * serial SyncVecEnv loop (see INTEGRATION.md for that subclass; a ready one ships in
 * gym_amd/include/gym/hip_vec_env.h).  Every entry point cites the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; int status returns (0 == GYMHIP_OK), never throws across the ABI;
 * gymhip_last_error() gives the thread-local message of the last failing call.  `stream` is a hipStream_t passed as
 * void* (NULL = the device's default stream); all work is enqueued, nothing synchronises unless documented.
 * Device buffers are owned by the engine (or by the caller-provided workspace) and stay valid until
 * gymhip_destroy(); they are OVERWRITTEN by the next step — the reference returns fresh tensors every step
 * (sync_env.h:107-112), so callers that keep a result across steps must copy it.
 */
#ifndef GYMHIP_H
#define GYMHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gymhip_engine gymhip_engine;

enum { GYMHIP_OK = 0, GYMHIP_EINVAL = 1, GYMHIP_EHIP = 2, GYMHIP_ESTATE = 3, GYMHIP_ETAPE = 4 };

/* classic_control/cartpole.{h,cpp}, classic_control/mountain_car.{h,cpp} */
enum { GYMHIP_ENV_CARTPOLE = 0, GYMHIP_ENV_MOUNTAINCAR = 1 };
/* LOCKSTEP_F64: the reference's exact op sequence and precisions (double state, float where it narrows) — bit-exact
 * streams.  F32: float state == the observation buffer, one 16-B read-modify-write per env (throughput mode). */
enum { GYMHIP_PRECISION_LOCKSTEP_F64 = 0, GYMHIP_PRECISION_F32 = 1 };
/* Where CartPole's reset states come from (cartpole.cpp:19-31; MountainCar's reset is constant, mountain_car.cpp:38-42).
 * TAPE: the k-th reset of env i reads tape[k][i][0..3], rows drawn on the HOST with std::mt19937 and uploaded (or any rows the
 *   caller supplies).  Finite; what the lockstep parity tests pin everything else against.
 * MT19937: every env owns the state array of a std::mt19937 ON THE DEVICE (2,496 B per env, part of the workspace) and the step
 *   kernel draws from it as the reference's CartPoleEnv::reset does from m_Device (env.h:83): libstdc++'s seeding, twist,
 *   tempering and uniform_real_distribution<float>(-0.05f, 0.05f) arithmetic in integer / float32 on the GPU, bit-identical to the
 *   host-drawn tape for any number of episodes — no refill, no host work, no PCIe traffic.  Until gymhip_seed() is called the
 *   generators are default-constructed (all envs draw the same stream), exactly as the reference's envs are before VecEnv::seed.
 *   Why not draw on the host at scale: 16.7 M envs finish ~7.5e5 episodes per 0.25-ms step — 4 x 10^9 draws/s more than 16 host
 *   cores produce, and 48 GB/s of rows over PCIe.
 * COUNTER_RNG: stateless hash(seed, global env index, the env's episode index) with the same float distribution arithmetic — same
 *   distribution, NOT the reference's stream.  The episode index is a 20-bit field of the env's counter word: after 2^20 episodes
 *   of one env (>= 8.4e6 steps) its reset states repeat with that period; the first wrap sets bit 4 of the error word. */
enum { GYMHIP_RESET_TAPE = 0, GYMHIP_RESET_COUNTER_RNG = 1, GYMHIP_RESET_MT19937 = 2 };
/* wrappers/time_limit.h, wrappers/clip_reward.h, wrappers/monitor.h, wrappers/vec_env/vec_normalize.h */
enum {
    GYMHIP_WRAP_TIME_LIMIT = 1u,
    GYMHIP_WRAP_CLIP_REWARD = 2u,
    GYMHIP_WRAP_MONITOR = 4u,
    GYMHIP_WRAP_NORM_OBS = 8u,
    GYMHIP_WRAP_NORM_REWARD = 16u,
    GYMHIP_WRAP_NORM_TRAINING = 32u, /* isTraining: update running stats every reset/step */
    /* SyncVecEnv(envs, auto_reset = false) (sync_env.h:28,44,92): a finished env is NOT reset; its terminal observation
     * is returned and it waits for gymhip_reset_index().  Stepping it before that reset is the misuse the reference's
     * Monitor reports on stderr ("Tried to step environment that needs reset", monitor.h:75-77): here the env stays
     * frozen (observation kept, reward 0, done 1) and bit 3 of the error word is set. */
    GYMHIP_WRAP_NO_AUTO_RESET = 64u
};
enum { GYMHIP_ACT_I32 = 0, GYMHIP_ACT_I64 = 1, GYMHIP_ACT_U8 = 2 };
/* Width S of the step field in the packed counter word (gymhip_buffers.counter).  A CartPole episode never exceeds
 * 501 steps (cartpole.cpp:64); MountainCar has no reset tape, so its word carries no cursor. */
#define GYMHIP_COUNTER_STEP_BITS_CARTPOLE 10
#define GYMHIP_COUNTER_STEP_BITS_MOUNTAINCAR 30

typedef struct gymhip_config {
    int32_t struct_size;        /* = sizeof(gymhip_config) */
    int32_t env_kind;
    int32_t precision;
    int32_t reset_mode;
    uint32_t wrappers;          /* GYMHIP_WRAP_* bitmask */
    int32_t max_episode_steps;  /* TimeLimit(max_episode_steps), time_limit.h:22-26 */
    int64_t n_envs;             /* envs resident on this GPU */
    int64_t env_offset;         /* global index of local env 0: env i is seeded seed + env_offset + i (sync_env.h:74-83) */
    uint64_t seed;
    float clip_obs, clip_reward, gamma, epsilon; /* vec_normalize.h:24-27 defaults 10, 10, 0.99, 1e-8 */
} gymhip_config;

/* Device pointers (NULL when the configuration has no such buffer).  N = n_envs, F = obs_dim. */
typedef struct gymhip_buffers {
    int32_t obs_dim;
    int32_t moments_len;            /* doubles in `moments` */
    void* state;                    /* LOCKSTEP: i32 [N][F] residuals k of the split-double state — the reference's double
                                       m_State element (cartpole.h:41, mountain_car.h:52) is exactly the
                                       double whose IEEE-754 bit pattern is bits((double)obs) + k, |k| <= 2^29;
                                       gymhip_read_state_f64() joins the halves.  F32: same memory as obs */
    float* obs;                     /* f32 [N][F] raw observation after auto-reset (VecNormalize::oldObs).  It is also the
                                       high half of the state: read-only for callers */
    float* obs_norm;                /* f32 [N][F] normalised observation (NORM_OBS) */
    float* reward;                  /* f32 [N]  (normalised in place when NORM_REWARD) */
    float* done;                    /* f32 [N]  0/1 — the reference's done tensor is float32 (sync_env.h:102,110) */
    uint32_t* counter;              /* u32 [N] packed per-env bookkeeping word, rewritten by every step (so the episode
                                       records and the tape cursor cost no HBM traffic of their own):
                                         bit 31      the LAST step finished an episode and auto-reset the env
                                         bit 30      that episode's info["TimeLimit.truncated"] (time_limit.h:32)
                                         bits S..29  CartPole: reset rows the env has consumed (TAPE: tape rows; MT19937: generator
                                                     rows mod 1,048,320; COUNTER_RNG: episode index mod 2^20), else 0
                                         bits 0..S-1 bit 31 set: the finished episode's length (info["episode"].l,
                                                     monitor.h:86); clear: steps since the last reset
                                                     (TimeLimit::elapsedSteps == Monitor::_steps == CartPole timeStep)
                                       S = GYMHIP_COUNTER_STEP_BITS_<ENV>.  gymhip_decode_counters() expands it. */
    float* returns;                 /* f32 [N] discounted return lane (vec_normalize.h:131-132,175) */
    float* norm_stats;              /* f32 [2F+4]: obs mean[F], obs var[F], obs count, ret mean, ret var, ret count */
    double* moments;                /* f64 [moments_len] all-reduce payload of the step in flight */
    double* episode_totals;         /* f64 [3] cumulative episodes, sum of returns, sum of lengths (after gymhip_episode_stats) */
    uint32_t* mt_state;             /* MT19937 engines: u32 [N][624], env i's std::mt19937 state array (libstdc++'s _M_x, untempered);
                                       its read position is (cursor % 156) * 4 words, a block being regenerated when a reset finds
                                       cursor % 156 == 0.  Together with `counter` this is the generator: copy both to checkpoint it */
} gymhip_buffers;

const char* gymhip_last_error(void);
int gymhip_version(void);

/* Bytes of device workspace an engine with this configuration needs (0 on invalid config). */
size_t gymhip_workspace_bytes(const gymhip_config* cfg);

/* Replaces constructing N env objects + SyncVecEnv (sync_env.h:25-60) [+ VecNormalize (vec_normalize.h:23-59)].
 * `workspace` may be NULL (the engine hipMallocs) or a caller-owned device allocation of at least
 * gymhip_workspace_bytes(cfg) bytes, 256-B aligned (e.g. a torch uint8 tensor). */
int gymhip_create(const gymhip_config* cfg, int device, void* workspace, size_t workspace_bytes, gymhip_engine** out);
int gymhip_destroy(gymhip_engine* h);
int gymhip_get_buffers(gymhip_engine* h, gymhip_buffers* out);

/* VecEnv::seed (base.h:74, sync_env.h:74-83): env i <- seed + env_offset + i.  MT19937: re-seeds every env's device-resident
 * std::mt19937 (libstdc++'s seed(value): x[0] = value mod 2^32, ...) and restarts its stream, as Env::seed does (env.h:60-63).
 * COUNTER_RNG: re-keys the stateless draw.  TAPE: the tape itself carries the seed (gymhip_seed_reset_tape).
 * Synchronises the device.  The seed is device state (kernels read it through the engine's reset-source block), so a step recorded
 * into a HIP graph before the call draws with the NEW seed when replayed. */
int gymhip_seed(gymhip_engine* h, uint64_t seed);

/* Host-drawn reset tape f32 [episodes][n_envs][4] (CartPole; cartpole.cpp:19-31 + env.h:60-68 semantics), copied to the
 * device.  MountainCar resets are deterministic (mountain_car.cpp:38-42) and need no tape. */
int gymhip_upload_reset_tape(gymhip_engine* h, const float* host_tape, int64_t episodes);
/* VecEnv::seed(seed) for TAPE mode (sync_env.h:74-83): draws the tape ON THE HOST with std::mt19937(seed + env_offset + i)
 * and std::uniform_real_distribution<float>(-0.05f, 0.05f) — the reference's own generator, 4 draws per reset in order
 * (cartpole.cpp:21-26) — for `episodes` resets per env, and uploads it.  Every call that loads a tape (this one,
 * _upload_, _unseeded_, _generate_) rewinds every env's tape cursor to row 0, as re-seeding restarts an env's generator
 * (env.h:60-63); between loads the cursor runs on across auto-resets, gymhip_reset() and gymhip_reset_index(), as the
 * generator does.  May be called again at any time (VecEnv::seed mid-run); synchronises the device. */
int gymhip_seed_reset_tape(gymhip_engine* h, uint64_t seed, int64_t episodes);
/* The same for envs that were never seeded: each keeps a default-constructed std::mt19937 (env.h:83), so — exactly as
 * in the reference — every env replays the same stream. */
int gymhip_unseeded_reset_tape(gymhip_engine* h, int64_t episodes);
/* A host-drawn tape is finite; an env that needs row `tape rows` wraps to row 0 and sets bit 0 of the error word.  These two
 * let a host mirror keep that from ever happening (gym::HipVecEnv and gym_amd.vec_env.HipVecEnv do, by default):
 *   _tape_cursor_max: rows consumed so far by the env that consumed most (synchronises `stream`; 0 without a tape).  No
 *       CartPole episode is shorter than min(8, max_episode_steps) steps, so with r rows left an engine can take 8 r - 8 more
 *       steps before the next look.
 *   _extend_reset_tape: redraw the SAME std::mt19937 streams with `episodes` (> current) rows per env; cursors are kept, a
 *       running engine just finds more rows.  Only for tapes drawn by _seed_/_unseeded_reset_tape; synchronises the device. */
int gymhip_tape_cursor_max(gymhip_engine* h, int32_t* host_out, void* stream);
int gymhip_extend_reset_tape(gymhip_engine* h, int64_t episodes);
/* Fill a device-resident tape of `episodes` rows with the COUNTER_RNG generator (bench / throughput set-up). */
int gymhip_generate_reset_tape(gymhip_engine* h, int64_t episodes, void* stream);
/* Copy the device tape back (tests: lets the CPU oracle replay a device-generated tape). */
int gymhip_download_reset_tape(gymhip_engine* h, float* host_tape, int64_t episodes);

/* VecEnv::reset() (base.h:52; sync_env.h:66-72; vec_normalize.h:145-157). */
int gymhip_reset(gymhip_engine* h, void* stream);
/* VecEnv::step(actions) = stepAsync + stepWait (base.h:56-63; sync_env.h:85-123; vec_normalize.h:159-177).
 * `actions` is a device array of n_envs elements of `action_dtype` (the reference narrows any tensor to int32,
 * tensor_adapter.cpp:22-23).  gymhip_step == gymhip_step_begin + gymhip_step_finish.
 * Stream capture: a step neither allocates nor synchronises, and all per-step state (tape cursor / generator row / counter-RNG
 * episode index per env, the normaliser's current statistics slot) AND the reset source itself (tape pointer, tape rows, seed: a
 * device-side block the kernels read through a pointer) live on the device: gymhip_step / gymhip_reset may be recorded into a HIP
 * graph and replayed; the caller rewrites its static action buffer between replays.  A recorded step stays valid across
 * gymhip_seed, gymhip_seed_reset_tape / _extend_reset_tape / _upload_reset_tape / _generate_reset_tape (a replaced tape is freed
 * only after the block points at the new one; tests/test_gpu_graph_capture.py). */
int gymhip_step(gymhip_engine* h, const void* actions, int action_dtype, void* stream);
/* Split form for sharded runs with the normaliser on: after _begin the `moments` buffer holds this rank's batch
 * moments + episode statistics; the caller all-reduces (sum) it over ranks (RCCL) on `stream`, then calls _finish,
 * which applies RunningMeanStd::updateFromMoment (running_mean_std.h:46-66) and normalises.  The same split exists
 * for reset. */
int gymhip_step_begin(gymhip_engine* h, const void* actions, int action_dtype, void* stream);
/* What the ranks exchange between _begin and _finish (normaliser engines; gymhip_get_buffers reports `moments` / `moments_len`
 * for the form in force — call it again after switching):
 *   GYMHIP_EXCHANGE_MOMENTS (default): _begin ends with a one-block reduce launch; the payload is 2F+6 doubles (80 - 112 B).
 *   GYMHIP_EXCHANGE_ROWS: no reduce launch — the payload is the step kernel's per-block partial rows themselves, f64
 *     [256 or 512][16] (32 / 64 KB, still latency-bound on xGMI), row 0 carrying this rank's env count; _finish folds the
 *     summed rows in the normalise kernel's prologue, as the single-GPU step does.  One launch fewer per step.  Every rank must
 *     use the same form and report the same `moments_len` (they do when their env counts are within the same size class).
 * No reference counterpart (the reference has no collective). */
enum { GYMHIP_EXCHANGE_MOMENTS = 0, GYMHIP_EXCHANGE_ROWS = 1 };
int gymhip_set_exchange_form(gymhip_engine* h, int form);
int gymhip_step_finish(gymhip_engine* h, void* stream);
int gymhip_reset_begin(gymhip_engine* h, void* stream);
int gymhip_reset_finish(gymhip_engine* h, void* stream);

/* K steps in ONE launch from a device action tape u8 [K][n_envs]; state stays in registers between steps.
 * Optional per-step records (device, may be NULL): done u8 [K][N], reward f32 [K][N], truncated u8 [K][N],
 * obs f32 [K][N][F].  Not available with the normaliser (it needs a grid-wide reduction per step).
 * No reference counterpart: it is the harness form of calling VecEnv::step K times (test/classic_control.cpp:92-96). */
int gymhip_rollout(gymhip_engine* h, const uint8_t* action_tape, int64_t k_steps, uint8_t* rec_done, float* rec_reward,
                   uint8_t* rec_truncated, float* rec_obs, void* stream);

/* Expand the packed counter words into plain device lanes (any pointer may be NULL): elapsed_steps i32 [N]; and, valid
 * where done != 0 (0 elsewhere), episode_length i32 [N] + episode_return f32 [N] = info["episode"].{l,r}
 * (monitor.h:74-98) and truncated u8 [N] = info["TimeLimit.truncated"] (time_limit.h:32); tape_cursor i32 [N]. */
int gymhip_decode_counters(gymhip_engine* h, int32_t* elapsed_steps, int32_t* episode_length, uint8_t* truncated,
                           float* episode_return, int32_t* tape_cursor, void* stream);

/* The envs' state as the reference holds it: f64 [N][F] device rows, (x, x_dot, theta, theta_dot) = CartPoleEnv::m_State
 * (cartpole.h:41) or (m_CarPos, m_CarVel) (mountain_car.h:52).  LOCKSTEP: exact; F32: the float state widened. */
int gymhip_read_state_f64(gymhip_engine* h, double* state_out, void* stream);

/* Reduce the per-block episode counters into episode_totals (device) and, if `host_out` is non-NULL, copy the three
 * doubles back (synchronises `stream`).  Monitor getters monitor.h:104-107 in aggregate form. */
int gymhip_episode_stats(gymhip_engine* h, double* host_out, void* stream);
/* Sticky device error word: bit0 = reset tape exhausted (an env needed more episodes than were uploaded; rows wrap),
 * bit1 = an env's tape-cursor field overflowed (2^20 episodes), bit2 = a LOCKSTEP state value left the split-double
 * range (non-finite, or nonzero below 2^-126 — the dynamics never produce either), bit3 = with GYMHIP_WRAP_NO_AUTO_RESET a
 * finished env was stepped again before its gymhip_reset_index().  bit4 is information, not an error: a COUNTER_RNG env's
 * 20-bit episode index wrapped (its reset states repeat from here on with period 2^20; use MT19937 for a stream that never does). */
int gymhip_poll_error(gymhip_engine* h, int32_t* host_out, void* stream);
/* Diagnostics, no counterpart in the reference: the launch shape and cache policy gymhip_create chose for this engine
 *  — out[0] = workgroups of the step / reset kernels, out[1] = threads per workgroup that own envs,
 * out[2] = 1 when the engine streams non-temporally (its per-step footprint does not fit the Infinity Cache), out[3] = its
 * residency plan: the 4-byte lanes kept in that cache, bit 0 counter, 1 reward, 2 done, 3 returns.  Speed only, never results. */
int gymhip_launch_info(gymhip_engine* h, int32_t out[4]);

/* VecEnv::reset(int index) (base.h:54; sync_env.h:128-139) under VecNormalize::reset(int index) (vec_normalize.h:179-189):
 * env `index` alone is reset (it consumes its next reset-tape row / a fresh stateless draw), its return lane is zeroed,
 * and obs_norm[index] is normalised with the current statistics, which are NOT updated.  The env's unfinished episode is
 * dropped without an episode record (Monitor's allowEarlyResets, monitor.h:49).  The per-index step(index, action) of
 * the reference is a stub that returns buffered values (sync_env.h:141-157): read the buffers. */
int gymhip_reset_index(gymhip_engine* h, int64_t index, void* stream);

/* Updateable::update / VecNormalize::update (vec_normalize.h:191-206): a learner pushes rollout data into the running
 * statistics.  obs_rows: device f32 [n_obs_rows][F] (the reference flattens [T][N][F], :197) -> obsRMS.update;
 * reward_rows: device f32 [n_reward_rows] (reshape({-1,1}), :204) -> retRMS.update.  Either pointer may be NULL.  Every
 * rank must push the same rows (the statistics are replicated). */
int gymhip_update_norm_stats(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows, int64_t n_reward_rows,
                             void* stream);

/* gymhip_update_norm_stats for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernels and bit-identical
 * results; the per-block partial sums and the reduced moments go through a static 131,200-byte device array that every GPU
 * the library runs on holds until the process ends, instead of the engine's workspace, and the call performs no
 * hipMalloc / hipFree / allocating library routine.  Limits: at most GYMHIP_NOALLOC_MAX_ROWS rows per pointer per call
 * (GYMHIP_EINVAL beyond).  All engines on one GPU share that GPU's array: calls that overlap on one GPU (different streams
 * or host threads) race — serialise them. */
#define GYMHIP_NOALLOC_MAX_ROWS (1ll << 26)
int gymhip_update_norm_stats_noalloc(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows,
                                     int64_t n_reward_rows, void* stream);

/* Diagnostic for the parity tests: the sin and cos the CartPole step kernel computes for a pole angle (cartpole.cpp:39-40
 * calls glibc's double cos/sin and narrows to float; the kernel evaluates the fdlibm polynomials in double and narrows the
 * same way).  theta / sin_out / cos_out: device f64 [n], values BEFORE the float narrowing, on the current device.
 * tests/test_sincos_parity.py counts how often the narrowed values differ from glibc's. */
int gymhip_diag_pole_sincos(const double* theta, int64_t n, double* sin_out, double* cos_out, void* stream);

/* VecNormalize::save/load state (vec_normalize.h:79-108; running_mean_std.h:28-40) as 2F+4 floats, host side. */
int gymhip_get_norm_stats(gymhip_engine* h, float* host_out, void* stream);
int gymhip_set_norm_stats(gymhip_engine* h, const float* host_in, void* stream);

/* ======================================================================================================================
 * Atari preprocessing (emulation stays on the host; frames arrive in device memory)
 * ==================================================================================================================== */
typedef struct gymhip_atari gymhip_atari;

typedef struct gymhip_atari_config {
    int32_t struct_size;    /* = sizeof(gymhip_atari_config) */
    int64_t n_envs;
    int32_t src_h, src_w;   /* 210, 160: AtariEnv's cv::Mat(210,160,CV_8UC3) RGB frame (atari/atari_env.cpp:101-113) */
    int32_t dst_h, dst_w;   /* 84, 84: WarpFrameEnv(width, height) (warp_frame.h:21-30) */
    int32_t n_stack;        /* VecFrameStack nStack (vec_frame_stack.cpp:10-16) */
    int32_t gray_bits;      /* 15 (OpenCV 4.x RGB2GRAY fixed point) or 14 (older builds); they differ by <= 1 LSB */
    int32_t clear_on_done;  /* 0 = as the reference (the stack is never cleared on done); 1 = SB3 semantics */
} gymhip_atari_config;

/* Replaces constructing MaxAndSkipEnv + WarpFrameEnv per env and VecFrameStack over them (atari_wrappers.cpp:95-121,
 * vec_frame_stack.cpp:10-20).  Owns the frame-stack ring u8 [n_envs][n_stack][dst_h*dst_w] (zero-initialised). */
int gymhip_atari_create(const gymhip_atari_config* cfg, int device, gymhip_atari** out);
int gymhip_atari_destroy(gymhip_atari* h);
/* Zero-copy access to the ring: plane (e, s) holds frame  t - ((head - s) mod n_stack)  of env e; `head` is the newest. */
int gymhip_atari_stack_buffer(gymhip_atari* h, uint8_t** ring, int32_t* head);

/* `frames`: device u8 [n_envs][2][src_h][src_w][3] — MaxAndSkipEnv's two observation buffers (frames skip-2 and skip-1
 * of the group; the host front-end overwrites only the slot it has a new frame for, which reproduces "buffers keep
 * older frames when the loop breaks on done", atari_wrappers.cpp:16-28).  Each call computes
 * resize_area(gray(max(buf0, buf1))) per env: atari_wrappers.cpp:30 + warp_frame.h:41-47.
 *   _preprocess: stateless, writes u8 [n_envs][dst_h][dst_w] to `out`.
 *   _reset: VecFrameStack::reset (vec_frame_stack.cpp:22-25; stacked_observation.h:106-109): zero the stack, newest last.
 *   _step:  VecFrameStack::stepWait (vec_frame_stack.cpp:27-32): shift by one frame, write the newest (INTENDED
 *           semantics of stacked_observation.h:126-130 — the reference as written mis-indexes the write).
 *           `done` (device u8 [n_envs], may be NULL) only matters with clear_on_done.
 * None of these allocates or synchronises.  The ring's `head` is host state, advanced by _step / _push: under HIP graph capture record
 * n_stack consecutive steps (a whole revolution of the ring), not one — a single recorded step replays into the same slot. */
int gymhip_atari_preprocess(gymhip_atari* h, const uint8_t* frames, uint8_t* out, void* stream);
int gymhip_atari_reset(gymhip_atari* h, const uint8_t* frames, void* stream);
int gymhip_atari_step(gymhip_atari* h, const uint8_t* frames, const uint8_t* done, void* stream);
/* The emulator's native screens instead of RGB frames.  AtariEnv::getObs colours ale->getScreen().getArray() (one palette
 * index per pixel) on the host through a 256-entry 0x00RRGGBB table (getScreenRGB2, atari/ale_c_wrapper.cpp:28-41;
 * the table is atari/atari_ntsc_rgb_palette.h) and hands RGB to the wrappers.  Here the host uploads the index screens
 * — a third of the bytes over PCIe and out of HBM — and the kernel applies the same table before the max:
 *   _set_palette: `zrgb256` = host array of 256 words 0x00RRGGBB (pass the reference's rgb_palette); call once.
 *   _*_indexed:   as the three calls above with `screens` = device u8 [n_envs][2][src_h][src_w].
 * Results are identical to colouring on the host and calling the RGB entry points. */
int gymhip_atari_set_palette(gymhip_atari* h, const uint32_t* zrgb256);
int gymhip_atari_preprocess_indexed(gymhip_atari* h, const uint8_t* screens, uint8_t* out, void* stream);
int gymhip_atari_reset_indexed(gymhip_atari* h, const uint8_t* screens, void* stream);
int gymhip_atari_step_indexed(gymhip_atari* h, const uint8_t* screens, const uint8_t* done, void* stream);
/* VecFrameStack ON ITS OWN (vec_frame_stack.cpp:10-32 over stacked_observation.h:106-130), for an inner VecEnv whose
 * observations are already dst_h x dst_w single-channel uint8 images (its [N, H, W, 1] batch, on the device, 4-byte aligned):
 *   _push_reset: VecFrameStack::reset — zero the stack, the planes become the newest frame;
 *   _push:       VecFrameStack::stepWait — shift by one frame, the planes become the newest (`done` as in _step).
 * The ring, the views and the materialise calls are those of _reset / _step: only the source of the newest plane differs. */
int gymhip_atari_push_reset(gymhip_atari* h, const uint8_t* planes, void* stream);
int gymhip_atari_push(gymhip_atari* h, const uint8_t* planes, const uint8_t* done, void* stream);
/* VecEnv::reset(int index) under VecFrameStack (sync_env.h:128-139 + vec_frame_stack.cpp:34-37): `plane` = the ONE freshly reset env's
 * dst_h x dst_w image on the device.  as_written = 0 (what a trainer means): env `index` starts over — its stack is zeroed and the
 * plane becomes its newest frame, as VecFrameStack::reset does for all envs (stacked_observation.h:106-109); the other envs' stacks
 * are untouched.  as_written = 1: the reference's code as it stands — StackedObservation::update is handed the single [H, W, 1]
 * frame, so EVERY env's stack shifts by one and that frame is broadcast into every env's newest slot. */
int gymhip_atari_push_index(gymhip_atari* h, int64_t index, const uint8_t* plane, int as_written, void* stream);
/* The stacked observation as a contiguous tensor: channels-last u8 [n_envs][dst_h][dst_w][n_stack], oldest frame in
 * channel 0 (what VecFrameStack returns over WarpFrame's {84,84,1} space, stacked_observation.h:96-99,203-209). */
int gymhip_atari_materialize_nhwc(gymhip_atari* h, uint8_t* out, void* stream);
/* Which stacked observation the materialise calls below produce.
 *   GYMHIP_STACK_INTENDED (default): the shift semantics StackedObservation means to implement (SB3's; oldest frame in
 *     channel 0), uint8 pixels.
 *   GYMHIP_STACK_AS_WRITTEN: what the reference's code computes today over WarpFrame's channels-last space, bug for bug:
 *     (ii) StackedObservation::update rolls the stack left by one channel and then passes stackDim (-1) to stack() where a size
 *     is expected (stacked_observation.h:96-99,126-130), so the write is y[..., 1:] = x: channel 0 = the previous call's
 *     channel 1, channels 1.. = the newest frame (frames 1,2,3,4 after a reset give [3,4,4,4]; straight after reset() the stack
 *     is [0,..,0,f]); (i) the buffer's dtype is int8 (make_shape<uint8_t> -> kChar, spaces/util.h:60-63;
 *     stacked_observation.h:208), so VecNormAndPermute's y / 255 (vec_atari.cpp:74-76) divides SIGNED bytes: a pixel of 200
 *     comes out as -56 / 255.  _nhwc writes the same bytes either way (read them as int8); the _nchw calls apply the signed
 *     reading.  Needs n_stack >= 2 (with one channel the reference's stack never changes after reset) and clear_on_done = 0.
 * The ring itself is the same for both; the view can be switched at any time. */
enum { GYMHIP_STACK_INTENDED = 0, GYMHIP_STACK_AS_WRITTEN = 1 };
int gymhip_atari_set_stack_view(gymhip_atari* h, int view);
/* VecNormAndPermute::observation fused onto the ring (vec_atari.cpp:70-81): f32 [n_envs][n_stack][dst_h][dst_w] = x / 255. */
int gymhip_atari_materialize_nchw_f32(gymhip_atari* h, float* out, void* stream);
/* The same tensor for half-precision policies: x / 255 in float32 as vec_atari.cpp:74-76 computes it,
 * rounded once (to nearest even) to IEEE half or bfloat16; `out` is 16-bit [n_envs][n_stack][dst_h][dst_w].  Half the
 * bytes of the float32 view; the reference itself only produces float32. */
enum { GYMHIP_OUT_F16 = 0, GYMHIP_OUT_BF16 = 1 };
int gymhip_atari_materialize_nchw_16(gymhip_atari* h, void* out, int dtype, void* stream);

/* Monitor -> EpisodicLifeEnv -> ClipRewardEnv for host-emulated envs, in AtariWrapper::make's order
 * (atari_wrappers.cpp:95-121), over device lanes of n envs.  Inputs per env: the MaxAndSkip-summed reward, the
 * emulator's done, info["ale.lives"].  Per-env state (updated in place): m_Lives, m_WasRealDone
 * (episodic_life.h:16-17), Monitor's running return and length (monitor.h:116-117).  Outputs: clipped reward
 * (clip_reward.h:22-24), done after the life-loss rule (atari_wrappers.cpp:35-47), info["episode"] lanes
 * (monitor.h:80-98; has_episode marks REAL game-overs, where the host must do a real reset; on a life-loss done it
 * steps the emulator with action 0 instead, atari_wrappers.cpp:49-53). */
int gymhip_host_env_post_step(int64_t n, const float* reward_in, const uint8_t* done_in, const int32_t* lives_in, int32_t* st_lives,
                              uint8_t* st_was_real_done, float* st_return, float* st_length, int clip_reward, float* reward_out,
                              uint8_t* done_out, uint8_t* has_episode, float* ep_return, float* ep_length, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GYMHIP_H */
