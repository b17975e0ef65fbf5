// classic_control.hip — gfx950 kernels + C ABI for the batched classic-control engine.
//
// Hot path replaced: SyncVecEnv::stepWait's serial loop (wrappers/vec_env/sync_env.h:114-123) over
// Monitor(ClipReward(TimeLimit(CartPoleEnv|MountainCarEnv))) objects plus VecNormalize::stepWait
// (wrappers/vec_env/vec_normalize.h:159-177).  One env per lane; everything per-env is fused into ONE kernel
// (dynamics, TimeLimit, ClipReward, Monitor bookkeeping, auto-reset, discounted-return lane, batch-moment partials);
// the normaliser's grid-wide reduction splits the step into  step -> reduce -> [all-reduce] -> normalise.
//
// Build with -ffp-contract=off: the lockstep path must round exactly where a stock x86-64 g++ build of the reference
// rounds (no FMA).  HBM-bound integer/float streaming work: no MFMA, no LDS tiling — coalescing and enough bytes in
// flight are what matter.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gymhip.h"
#include "gymhip_internal.h"

namespace gymhip {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace gymhip

using namespace gymhip;

namespace {

#ifndef GYMHIP_BLOCK
#define GYMHIP_BLOCK 512
#endif
constexpr int kBlock = GYMHIP_BLOCK;   // 512: equal to 256 on the headline kernel, best on the normaliser pair (profiles/r02_sweep_block.txt)
constexpr int kMaxGrid = 16384; // upper bound of the grid-stride launch (sizes the per-block episode-counter rows)
constexpr int kDefaultGrid = 512;   // 256 CUs x 2 blocks of 512 threads: best for the float-state and MountainCar kernels and for normaliser engines (fewer partial rows to fold); GYMHIP_GRID overrides
constexpr int kPlainGrid = 2048;    // CartPole lockstep without the normaliser (the 84-B kernel): 1024 .. 4096 measure alike, 1.5 % above 512 (profiles/r02_sweeps.txt)
constexpr int kStreamGrid = 16384;  // the same kernel once it streams non-temporally (residency plan): 2 envs per thread, +2 % over 2048 at 8 .. 33 M envs; the other kernels lose there
constexpr size_t kNtFootprint = (size_t)256 << 20;   // bytes (the Infinity Cache); see gymhip_create
constexpr size_t kMallBudget = (size_t)240 << 20;    // bytes of the 256 MiB Infinity Cache a residency plan fills (plan_residency)
enum : int { kKeepCounter = 1, kKeepReward = 2, kKeepDone = 4, kKeepReturns = 8, kKeepNotDone = 11, kKeepAll = 15 };
constexpr int kPlans[] = {kKeepAll, kKeepNotDone, kKeepCounter, 0};   // the plans the step kernel is compiled for, most bytes kept first
constexpr int kMaxCols = 16;    // >= 2F+6 for F=4

enum : int { KIND_CARTPOLE = GYMHIP_ENV_CARTPOLE, KIND_MOUNTAINCAR = GYMHIP_ENV_MOUNTAINCAR };

// ------------------------------------------------------------------------------------------------------------
// Reset-state generator.  TAPE rows come from the host's std::mt19937; COUNTER_RNG hashes (seed, env, episode) and
// applies libstdc++'s uniform_real_distribution<float>(-0.05f, 0.05f) arithmetic to the 32-bit word:
// c = float(u) / 2^32 (>= 1 -> nextafter(1,0)); value = c * (b - a) + a   (SURVEY.md §8a A3).
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline float canonical_to_range(uint32_t u) {
    float c = (float)u * 2.3283064365386963e-10f;  // exact power-of-two scale == float(u) / 4294967296.0f
    if (c >= 1.0f) c = 0.99999994f;
    return c * 0.1f + (-0.05f);
}
__host__ __device__ inline void counter_reset_draw(uint64_t seed, int64_t global_env, int64_t episode, float (&s)[4]) {
    uint64_t k = splitmix64(seed + (uint64_t)global_env);
    uint64_t a = splitmix64(k ^ ((uint64_t)episode * 0xD1B54A32D192ED03ull + 1ull));
    uint64_t b = splitmix64(a);
    s[0] = canonical_to_range((uint32_t)a);
    s[1] = canonical_to_range((uint32_t)(a >> 32));
    s[2] = canonical_to_range((uint32_t)b);
    s[3] = canonical_to_range((uint32_t)(b >> 32));
}

// ------------------------------------------------------------------------------------------------------------
// MT19937 reset mode: every env owns the state array of a std::mt19937 ON THE DEVICE (u32 [N][624], 2,496 B per env) — the
// reference's own generator (env.h:83, cartpole.cpp:21-26), continued for ever, with no host refill and no PCIe traffic.  The k-th
// reset of an env takes outputs 4k .. 4k+3 of its stream: "row" k % 156 of the current 624-word block; when a reset needs row 0
// the block is regenerated first, in place, exactly as libstdc++'s mersenne_twister_engine::_M_gen_rand does when _M_p reaches
// 624 (a freshly seeded engine starts there, so the first reset twists too).  The array holds the UNTEMPERED words; a row is
// tempered when read.  The twist is done by the whole WAVE for one env at a time: its 64 lanes stream the env's 624 words through
// LDS with coalesced loads and stores, so the regeneration (one per 156 resets of an env) moves 2 x 2,496 B at full line efficiency
// and costs the other lanes about a microsecond, instead of one lane walking 624 words alone.
// Integer arithmetic throughout: the device stream is bit-identical to std::mt19937's (tests/test_gpu_mt19937.py pins it against
// libstdc++ on the host).
// ------------------------------------------------------------------------------------------------------------
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
constexpr int kMtWords = 624, kMtShift = 397, kMtRows = kMtWords / 4;
constexpr uint32_t kMtCursorPeriod = 6720u * (uint32_t)kMtRows;   // 1,048,320: the largest multiple of 156 the 20-bit cursor field holds
__host__ __device__ inline uint32_t mt_temper(uint32_t y) {        // mersenne_twister_engine::operator() (u = 11, s = 7, t = 15, l = 18)
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}
__host__ __device__ inline uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7FFFFFFFu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
__device__ inline void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Starts reading one env's generator block into the wave's LDS scratch WITHOUT holding registers (LDS-DMA: three 16-B-per-lane
// global_load_lds, non-temporal), for the first lane with `likely` — the caller's exact forecast, made before the dynamics, that the lane
// will need a twist at the end of this step — so that the block's HBM round trip runs under the step's arithmetic instead of stalling
// the wave afterwards.  Full waves only.  Returns that lane (wave-uniform) or -1; mt_draw_row(..., prefetched) consumes it.
using global_cptr = const __attribute__((address_space(1))) void*;
using lds_ptr = __attribute__((address_space(3))) void*;
__device__ inline int mt_prefetch_block(bool likely, const uint32_t* __restrict__ mt_state, int64_t i, uint32_t* lds) {
    const uint64_t m = __ballot(likely);
    if (!m || __popcll(__ballot(1)) != 64) return -1;
    const int src = __builtin_ctzll(m), lane = (int)__lane_id();
    const int64_t ei = ((int64_t)__shfl((int)(i >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)i, src);
    const uint32_t* st = mt_state + ei * kMtWords + 4 * lane;            // 2,496 B per env: 16-B aligned
    __builtin_amdgcn_global_load_lds((global_cptr)st, (lds_ptr)lds, 16, 0, 2);                     // LDS destination = base + lane * 16
    __builtin_amdgcn_global_load_lds((global_cptr)(st + 256), (lds_ptr)(lds + 256), 16, 0, 2);
    if (lane < (kMtWords - 512) / 4) __builtin_amdgcn_global_load_lds((global_cptr)(st + 512), (lds_ptr)(lds + 512), 16, 0, 2);
    return src;
}

// Collective over the ACTIVE lanes of a wave (call it from converged code).  A lane with `want` gets the four raw (tempered)
// outputs of its env's next reset row in `raw`; `cursor` = rows its env has consumed so far (mod kMtCursorPeriod).
// `lds` = this wave's private 624-word scratch.
template <bool SAME_LAUNCH_WRITES = false>   // true (rollout kernel): an env's block may have been rewritten earlier in this launch, by other lanes
__device__ inline void mt_draw_row(bool want, uint32_t cursor, uint32_t* __restrict__ mt_state, int64_t i, uint32_t* lds, uint32_t (&raw)[4],
                                   int prefetched = -1) {   // prefetched: the lane whose block mt_prefetch_block has on its way into `lds`
    const uint32_t pos = cursor % (uint32_t)kMtRows;
    const bool twist = want && pos == 0;
    const uint64_t active = __ballot(1);
    uint64_t todo = __ballot(twist);
    if (prefetched >= 0 && !todo) __builtin_amdgcn_s_waitcnt(0);  // a forecast that did not come true (it is exact, so: never) still lands before the scratch is reused
    if (todo) {                                                   // wave-uniform
        const int lane = (int)__lane_id();
        const int nact = __popcll(active);
        const int rank = __popcll(active & ((1ull << lane) - 1ull));
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t ei = ((int64_t)__shfl((int)(i >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)i, src);
            uint32_t* st = mt_state + ei * kMtWords;
            // Partial waves (a launch's last wave, single-env kernels): rounds of `nact` consecutive words in increasing order; within a round every lane reads its three operands before any
            // lane writes.  That reproduces the sequential recurrence: x[k+1] is still old when x[k] is computed, and the far operand
            // x[(k+397) % 624] is old for k < 227 and already new (written >= 227 words, i.e. >= 3 rounds, earlier) for k >= 227.
            auto round = [&](int k) {
                uint32_t v = 0;
                if (k < kMtWords) {
                    const int k1 = k + 1 == kMtWords ? 0 : k + 1;
                    const int kf = k + kMtShift >= kMtWords ? k + kMtShift - kMtWords : k + kMtShift;
                    v = mt_mix(lds[k], lds[k1], lds[kf]);
                }
                wave_lds_fence();
                if (k < kMtWords) lds[k] = v;
                wave_lds_fence();
            };
            if (nact == 64) {
                // the full wave (all but a launch's last wave): everything unrolled, non-temporal both ways — a block is touched once per
                // ~3,500 steps
                constexpr int R = (kMtWords + 63) / 64;
                if (prefetched >= 0) __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): a pending LDS-DMA has landed (nothing else orders LDS accesses behind it)
                if (src != prefetched) {
                    // not forecast: the ten 256-B loads of the block issued back to back (ONE round trip to HBM instead of ten)
                    uint32_t w[R];
#pragma unroll
                    for (int r = 0; r < R; r++) { const int k = lane + 64 * r; w[r] = k < kMtWords ? __builtin_nontemporal_load(st + k) : 0u; }
#pragma unroll
                    for (int r = 0; r < R; r++) { const int k = lane + 64 * r; if (k < kMtWords) lds[k] = w[r]; }
                }
                prefetched = -1;                                       // whatever was prefetched is consumed or overwritten now
                wave_lds_fence();
                // Three phases instead of ten rounds: within [0, 227), [227, 454) and [454, 624) no word depends on another word of the
                // same phase except through the OLD successor x[k+1] (read before anyone writes); the far operand x[(k+397) % 624] is old
                // in the first phase and comes from the phase before in the other two; x[623] takes the new x[0].
                auto phase = [&](int lo, int hi) {
                    uint32_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int k = lo + lane + 64 * j;
                        if (k < hi) {
                            const int k1 = k + 1 == kMtWords ? 0 : k + 1;
                            const int kf = k + kMtShift >= kMtWords ? k + kMtShift - kMtWords : k + kMtShift;
                            v[j] = mt_mix(lds[k], lds[k1], lds[kf]);
                        }
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int k = lo + lane + 64 * j; if (k < hi) lds[k] = v[j]; }
                    wave_lds_fence();
                };
                phase(0, kMtWords - kMtShift);
                phase(kMtWords - kMtShift, 2 * (kMtWords - kMtShift));
                phase(2 * (kMtWords - kMtShift), kMtWords);
#pragma unroll
                for (int r = 0; r < R; r++) { const int k = lane + 64 * r; if (k < kMtWords) __builtin_nontemporal_store(lds[k], st + k); }
            } else {
                for (int k = rank; k < kMtWords; k += nact) lds[k] = st[k];
                wave_lds_fence();
                for (int base = 0; base < kMtWords; base += nact) round(base + rank);
                for (int k = rank; k < kMtWords; k += nact) st[k] = lds[k];
            }
            if (lane == src) { raw[0] = mt_temper(lds[0]); raw[1] = mt_temper(lds[1]); raw[2] = mt_temper(lds[2]); raw[3] = mt_temper(lds[3]); }
            // rollout kernel: the block's later rows are read back in this launch (by the env's own lane, past L1): the stores must have
            // reached L2 by then — wait for them here (vmcnt = 0) rather than paying a device-scope fence
            if constexpr (SAME_LAUNCH_WRITES) __builtin_amdgcn_s_waitcnt(0);
            wave_lds_fence();
        }
    }
    if (want && !twist) {
        const uint32_t* row = mt_state + i * kMtWords + pos * 4u;
        if constexpr (SAME_LAUNCH_WRITES) {   // L1-bypassing loads: the wave's own earlier stores (made by other lanes) must be seen
#pragma unroll
            for (int k = 0; k < 4; k++) raw[k] = mt_temper(__hip_atomic_load(row + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } else {                              // one 16-B gather per reset, as from a tape; this launch never wrote the block
            const u32x4 v = *(const u32x4*)row;
            raw[0] = mt_temper(v.x); raw[1] = mt_temper(v.y); raw[2] = mt_temper(v.z); raw[3] = mt_temper(v.w);
        }
    }
}

// std::mt19937::seed(value) (libstdc++ bits/random.tcc): x[0] = value mod 2^32, x[k] = 1812433253 (x[k-1] ^ x[k-1] >> 30) + k; the
// engine then stands at _M_p = 624.  env i <- seed + env_offset + i (sync_env.h:74-83), or the default-constructed 5489 (env.h:83).
// One env per lane runs the serial recurrence; 64-word chunks go through an LDS tile so that the stores are 256-B coalesced.
__global__ __launch_bounds__(256) void mt_seed_kernel(uint32_t* __restrict__ mt_state, int64_t n, int seeded, uint64_t seed, int64_t env_offset) {
    __shared__ uint32_t tile[4][64][65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < n; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + lane;
        uint32_t x = seeded ? (uint32_t)(seed + (uint64_t)(env_offset + i)) : 5489u;
        for (int c = 0; c < kMtWords; c += 64) {
            const int cnt = kMtWords - c < 64 ? kMtWords - c : 64;
            for (int j = 0; j < cnt; j++) {
                const uint32_t k = (uint32_t)(c + j);
                if (k > 0) x = 1812433253u * (x ^ (x >> 30)) + k;
                tile[wave][lane][j] = x;
            }
            wave_lds_fence();
            for (int e = 0; e < 64 && base + e < n; e++)
                if (lane < cnt) mt_state[(base + e) * kMtWords + c + lane] = tile[wave][e][lane];
            wave_lds_fence();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Dynamics.  State held as T[F] in registers.
// ------------------------------------------------------------------------------------------------------------
struct CartPoleC {  // cartpole.h:30-39 — float constants
    static constexpr float gravity = 9.8f, masscart = 1.0f, masspole = 0.1f;
    static constexpr float total_mass = masspole + masscart;
    static constexpr float length = 0.5f;
    static constexpr float polemass_length = masspole * length;
    static constexpr float force_mag = 10.0f, tau = 0.02f;
    static constexpr float theta_thr = (float)(12 * 2 * 3.14159265358979323846 / 360);
    static constexpr float x_thr = 2.4f;
};

// GYMHIP_DIAG (tools/sweep_variants.sh only; never defined in the shipped library — results are WRONG when it is set):
//   1 = trivial dynamics, no env ever done: the pure cost of the kernel's HBM streams
//   2 = sin/cos replaced by two-term polynomials      3 = float divides replaced by reciprocal multiplies
//   4 = the device library's sincos instead of pole_sincos (results stay right; A/B timing only)
#ifndef GYMHIP_DIAG
#define GYMHIP_DIAG 0
#endif
__device__ inline float diag_div(float a, float b) {
#if GYMHIP_DIAG == 3
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}

// sin and cos of the pole angle in double.  At step entry |theta| <= theta_thr (0.2095 rad) always holds: a step that
// leaves the range ends the episode and the state is replaced by a reset draw.  Below 0.3 rad no argument reduction is
// needed and the classic fdlibm kernels (k_sin.c / k_cos.c minimax coefficients, error < 1 ulp — the class of accuracy
// glibc's and the device library's sin/cos have) apply directly: ~25 fp64 instructions instead of the ~170 of the
// device library's sincos with its Cody-Waite/Payne-Hanek reduction.  Only the float-rounded values are used
// (cartpole.cpp:39-40), so two < 1 ulp doubles agree after rounding except within ~2^-29 relative of a float tie.
// Anything else falls back to the device library.
__device__ __attribute__((noinline)) double2 library_sincos(double x) { double2 r; r.x = sin(x); r.y = cos(x); return r; }
__device__ inline void pole_sincos(double x, double* sn, double* cs) {
    if (__builtin_expect(__builtin_fabs(x) < 0.3, 1)) {
        const double z = x * x;
        const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03);
        *sn = fma(z * x, fma(z, rs, -1.66666666666666324348e-01), x);
        const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                                  2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
        *cs = 1.0 - (0.5 * z - z * rc);
    } else {
        const double2 r = library_sincos(x);   // out of line: keeps its registers and code off the hot path
        *sn = r.x; *cs = r.y;
    }
}

// cartpole.cpp:33-68, exact precisions (A1).  Returns the env's own `done`.
__device__ inline bool cartpole_step(double (&s)[4], int action, int time_step) {
    using C = CartPoleC;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
#if GYMHIP_DIAG == 1
    s[0] = x + 1e-9; s[1] = x_dot + 1e-9; s[2] = theta + 1e-9 * action; s[3] = theta_dot + 1e-9;
    return time_step < 0;
#endif
    float force = (action == 1) ? C::force_mag : -C::force_mag;
    double sd, cd;
#if GYMHIP_DIAG == 2
    sd = theta - theta * theta * theta * (1.0 / 6.0); cd = 1.0 - theta * theta * 0.5;
#elif GYMHIP_DIAG == 4
    sincos(theta, &sd, &cd);      // the device library's sincos, for A/B runs against pole_sincos
#else
    pole_sincos(theta, &sd, &cd);
#endif
    float cos_t = (float)cd, sin_t = (float)sd;
    float temp = diag_div((float)((double)force + ((double)C::polemass_length * (theta_dot * theta_dot)) * (double)sin_t), C::total_mass);
    float num = C::gravity * sin_t - cos_t * temp;
    float inner = diag_div(C::masspole * (cos_t * cos_t), C::total_mass);
    float den = (float)((double)C::length * (4.0 / 3.0 - (double)inner));
    float theta_acc = diag_div(num, den);
    float x_acc = temp - diag_div(C::polemass_length * theta_acc * cos_t, C::total_mass);
    x = x + (double)C::tau * x_dot;
    x_dot = x_dot + (double)(C::tau * x_acc);
    theta = theta + (double)C::tau * theta_dot;
    theta_dot = theta_dot + (double)(C::tau * theta_acc);
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    return time_step == 500 || x < (double)(-C::x_thr) || x > (double)C::x_thr || theta < (double)(-C::theta_thr) ||
           theta > (double)C::theta_thr;
}
// Same op sequence with float state (throughput mode).
__device__ inline bool cartpole_step(float (&s)[4], int action, int time_step) {
    using C = CartPoleC;
    float x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
#if GYMHIP_DIAG == 1
    s[0] = x + 1e-6f; s[1] = x_dot + 1e-6f; s[2] = theta + 1e-6f * action; s[3] = theta_dot + 1e-6f;
    return time_step < 0;
#endif
    float force = (action == 1) ? C::force_mag : -C::force_mag;
    float sin_t, cos_t;
#if GYMHIP_DIAG == 2
    sin_t = theta - theta * theta * theta * (1.f / 6.f); cos_t = 1.f - theta * theta * 0.5f;
#else
    sincosf(theta, &sin_t, &cos_t);
#endif
    float temp = diag_div(force + C::polemass_length * (theta_dot * theta_dot) * sin_t, C::total_mass);
    float num = C::gravity * sin_t - cos_t * temp;
    float inner = diag_div(C::masspole * (cos_t * cos_t), C::total_mass);
    float den = C::length * (1.33333337f - inner);
    float theta_acc = diag_div(num, den);
    float x_acc = temp - diag_div(C::polemass_length * theta_acc * cos_t, C::total_mass);
    x = x + C::tau * x_dot;
    x_dot = x_dot + C::tau * x_acc;
    theta = theta + C::tau * theta_dot;
    theta_dot = theta_dot + C::tau * theta_acc;
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    return time_step == 500 || x < -C::x_thr || x > C::x_thr || theta < -C::theta_thr || theta > C::theta_thr;
}
// Whether the step about to be taken ends the episode, from the state BEFORE it: x and theta advance by tau times the OLD velocities
// (cartpole.cpp:50,52), so the forecast repeats those two operations and the test exactly — it needs neither the action nor sin / cos.
__device__ inline bool cartpole_done_ahead(const double (&s)[4], int time_step) {
    using C = CartPoleC;
    const double x = s[0] + (double)C::tau * s[1], theta = s[2] + (double)C::tau * s[3];
    return time_step == 500 || x < (double)(-C::x_thr) || x > (double)C::x_thr || theta < (double)(-C::theta_thr) || theta > (double)C::theta_thr;
}
__device__ inline bool cartpole_done_ahead(const float (&s)[4], int time_step) {
    using C = CartPoleC;
    const float x = s[0] + C::tau * s[1], theta = s[2] + C::tau * s[3];
    return time_step == 500 || x < -C::x_thr || x > C::x_thr || theta < -C::theta_thr || theta > C::theta_thr;
}
// mountain_car.cpp:17-36 (A2): float constants widened on use.
__device__ inline bool mountaincar_step(double (&s)[2], int action) {
    const double MIN_POS = (double)-1.2f, MAX_POS = (double)0.6f, MAX_VEL = (double)0.07f, GOAL = (double)0.5f;
    double pos = s[0], vel = s[1];
    vel += (action - 1) * 0.001 + cos(3 * pos) * -0.0025;
    if (vel > MAX_VEL) vel = MAX_VEL;
    if (vel < -MAX_VEL) vel = -MAX_VEL;
    pos += vel;
    if (pos > MAX_POS) pos = MAX_POS;
    if (pos < MIN_POS) pos = MIN_POS;
    if (pos == MIN_POS && vel < 0) vel = 0;
    s[0] = pos; s[1] = vel;
    return pos >= GOAL;
}
__device__ inline bool mountaincar_step(float (&s)[2], int action) {
    float pos = s[0], vel = s[1];
    vel += (float)(action - 1) * 0.001f + cosf(3.f * pos) * -0.0025f;
    if (vel > 0.07f) vel = 0.07f;
    if (vel < -0.07f) vel = -0.07f;
    pos += vel;
    if (pos > 0.6f) pos = 0.6f;
    if (pos < -1.2f) pos = -1.2f;
    if (pos == -1.2f && vel < 0) vel = 0;
    s[0] = pos; s[1] = vel;
    return pos >= 0.5f;
}

// Packed per-env counter word (u32, one per env, rewritten every step — see gymhip.h `counter`):
//   bit 31      the LAST step finished an episode (the env was auto-reset)
//   bit 30      that episode was truncated by TimeLimit
//   bits S..29  reset-tape cursor = tape rows consumed so far (TAPE mode, CartPole only), else 0
//   bits 0..S-1 steps: the finished episode's length if bit 31 is set, else steps since the last reset
// S = 10 for CartPole (an episode never exceeds 501 steps, cartpole.cpp:64), 30 for MountainCar (no tape, no cursor).
template <int KIND> struct CounterWord {
    static constexpr int S = KIND == GYMHIP_ENV_CARTPOLE ? GYMHIP_COUNTER_STEP_BITS_CARTPOLE : GYMHIP_COUNTER_STEP_BITS_MOUNTAINCAR;
    static constexpr uint32_t kStepMask = (1u << S) - 1u;
    static constexpr uint32_t kCursorMask = (1u << (30 - S)) - 1u;
    __host__ __device__ static int steps_field(uint32_t w) { return (int)(w & kStepMask); }
    __host__ __device__ static int steps_now(uint32_t w) { return (w >> 31) ? 0 : (int)(w & kStepMask); }
    __host__ __device__ static uint32_t cursor(uint32_t w) { return (w >> S) & kCursorMask; }
    __device__ static uint32_t next_cursor(uint32_t c, int32_t* error_flag) {
        if (c >= kCursorMask) { atomicOr(error_flag, 2); return 0; }   // cursor field exhausted (2^20 episodes per env)
        return c + 1;
    }
    // COUNTER_RNG: the field is the env's episode index and wraps after 2^20 episodes — from there on the env's reset states repeat
    // with that period (documented in gymhip.h); the wrap is reported once, as information, in bit 4 of the error word
    __device__ static uint32_t next_cursor_wrapping(uint32_t c, int32_t* error_flag) {
        const uint32_t nx = (c + 1u) & kCursorMask;
        if (nx == 0u) atomicOr(error_flag, 16);
        return nx;
    }
    // MT19937: rows consumed mod kMtCursorPeriod (a multiple of the 156 rows of a generator block): the stream itself never ends
    __host__ __device__ static uint32_t next_cursor_mt(uint32_t c) { return c + 1u == kMtCursorPeriod ? 0u : c + 1u; }
    __host__ __device__ static uint32_t pack(int t, uint32_t cur, bool done, bool trunc) {
        return (done ? 0x80000000u : 0u) | (trunc ? 0x40000000u : 0u) | (cur << S) | (uint32_t)t;
    }
};

// The statistics slot a kernel that only reads the statistics uses, and the hand-off that lets the publishing kernel find it
// without any host-side state (so a recorded step replays correctly from a HIP graph): see normalize_kernel.
__device__ inline const float* stats_reader_slot(const StepParams& p) {
    const int32_t t = *p.tick_w;
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.tick_r = t;      // nobody reads tick_r in this kernel
    return p.stats + 16 * (t & 1);
}
// The same, for the hot kernels: the values are read from BOTH slots and picked afterwards, so that the loads do not wait for the
// tick (one dependent round trip less at the start of a launch-latency-bound kernel).  `idx[k]` are offsets within a slot.
template <int K> __device__ inline void stats_reader_values(const StepParams& p, const int (&idx)[K], float (&out)[K]) {
    const int32_t t = *p.tick_w;
    float a[K], b[K];
#pragma unroll
    for (int k = 0; k < K; k++) { a[k] = p.stats[idx[k]]; b[k] = p.stats[16 + idx[k]]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.tick_r = t;
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = (t & 1) ? b[k] : a[k];
}

template <int KIND> struct EnvTraits;
template <> struct EnvTraits<KIND_CARTPOLE> { static constexpr int F = 4; static constexpr float step_reward = 1.0f; };
template <> struct EnvTraits<KIND_MOUNTAINCAR> { static constexpr int F = 2; static constexpr float step_reward = -1.0f; };

// ------------------------------------------------------------------------------------------------------------
// State I/O.  F32: the f32 obs rows [N][F] ARE the state (float4 / float2 per lane).  LOCKSTEP: the same obs rows plus
// int32 residual rows [N][F] that complete each float to the exact double (split-double, below).
// ------------------------------------------------------------------------------------------------------------
// Compile-time tuning knobs (profiles/NOTES.md records what each is worth on MI355X):
//   GYMHIP_NT   1 = non-temporal loads/stores on the once-per-step streams (working set >> 256 MiB Infinity Cache)
#ifndef GYMHIP_NT
#define GYMHIP_NT 1
#endif
//   GYMHIP_PIPE 1 = software-pipelined step loop (iteration k+1's loads in flight under iteration k's gather + stores)
// (An earlier knob, several envs per thread per iteration, was worth < 1 % and is gone: profiles/r01_sweep.txt.)
#ifndef GYMHIP_PIPE
#define GYMHIP_PIPE 1
#endif
//   GYMHIP_TAPE_NT 1 = non-temporal load for the reset-tape gather
#ifndef GYMHIP_TAPE_NT
#define GYMHIP_TAPE_NT 0
#endif
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using i32x4 = int32_t __attribute__((ext_vector_type(4)));
using i32x2 = int32_t __attribute__((ext_vector_type(2)));
// NT is a template argument so that the step kernel can pick per launch: non-temporal pays when the engine's footprint is
// well beyond the Infinity Cache (16.7 M envs: +7 %), and costs when it (nearly) fits (4 M envs: -16 %) — see kNtFootprint.
constexpr bool kNtDefault = GYMHIP_NT != 0;
// The cache policy of an access is a COMPILE-TIME property here on purpose.  A run-time choice (`keep ? plain : non-temporal`)
// does not survive the optimiser: it sees the same load (store) in both arms, sinks them into one access behind a pointer phi
// and drops the non-temporal hint — measured: every plan then ran like an all-temporal kernel; keeping the arms apart with asm
// fences costs the float-state kernel 15 %.  So plans are template arguments (step_kernel<.., KEEP>).
template <bool NT = kNtDefault, typename V> __device__ inline V ld_stream(const V* base, int64_t i) {
    if constexpr (NT) return __builtin_nontemporal_load(base + i);
    else return base[i];
}
template <bool NT = kNtDefault, typename V> __device__ inline void st_stream(V* base, int64_t i, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, base + i);
    else base[i] = v;
}
// Row accessors.  Rows are 16-B (F = 4) or 8-B (F = 2) vectors: one coalesced access per lane.
template <bool NT = kNtDefault> __device__ inline void load_obs(const float* obs, int64_t i, float (&s)[4]) { f32x4 v = ld_stream<NT>((const f32x4*)obs, i); s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w; }
template <bool NT = kNtDefault> __device__ inline void load_obs(const float* obs, int64_t i, float (&s)[2]) { f32x2 v = ld_stream<NT>((const f32x2*)obs, i); s[0] = v.x; s[1] = v.y; }
template <bool NT = kNtDefault> __device__ inline void store_obs(float* obs, int64_t i, const float (&s)[4]) { f32x4 v = {s[0], s[1], s[2], s[3]}; st_stream<NT>((f32x4*)obs, i, v); }
template <bool NT = kNtDefault> __device__ inline void store_obs(float* obs, int64_t i, const float (&s)[2]) { f32x2 v = {s[0], s[1]}; st_stream<NT>((f32x2*)obs, i, v); }
template <bool NT = kNtDefault> __device__ inline void load_i32(const int32_t* a, int64_t i, int32_t (&k)[4]) { i32x4 v = ld_stream<NT>((const i32x4*)a, i); k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w; }
template <bool NT = kNtDefault> __device__ inline void load_i32(const int32_t* a, int64_t i, int32_t (&k)[2]) { i32x2 v = ld_stream<NT>((const i32x2*)a, i); k[0] = v.x; k[1] = v.y; }
template <bool NT = kNtDefault> __device__ inline void store_i32(int32_t* a, int64_t i, const int32_t (&k)[4]) { i32x4 v = {k[0], k[1], k[2], k[3]}; st_stream<NT>((i32x4*)a, i, v); }
template <bool NT = kNtDefault> __device__ inline void store_i32(int32_t* a, int64_t i, const int32_t (&k)[2]) { i32x2 v = {k[0], k[1]}; st_stream<NT>((i32x2*)a, i, v); }

// Split-double state (LOCKSTEP).  A double d is kept as the pair (f, k): f = (float)d — exactly the observation the API
// returns, so the obs row doubles as the high half of the state — and k = bits(d) - bits((double)f), the distance between the
// two IEEE-754 bit patterns as a signed integer (= the number of doubles between them: patterns of one sign are ordered like
// their magnitudes, across binades too).  |d - (double)f| <= ulp_f / 2 = 2^(e_f - 24) and the doubles there are at least
// 2^(e_f - 53) apart (d has exponent e_f or e_f - 1), so |k| <= 2^29: an int32, and d == from_bits(bits((double)f) + k) exactly.
// 8 B per state element become 4 B beyond the obs row that is written anyway.  Not representable: nonzero |d| < 2^-126
// and non-finite d; the dynamics cannot produce either (every value is a short sum of terms >= 2^-80 in magnitude),
// and bit 2 of the error flag reports it if it ever happens.
// Integer arithmetic on purpose: two conversions, a 32-bit subtract and a 64-bit add per element, where the first form of this
// format (k = (d - f) scaled by a power of two) spent a double multiply, a double add / subtract and two int <-> double
// conversions on each side — they run four times per env-step on the hot path, and fp64 issues at half rate.
__device__ inline int32_t split_residual(double d, float f, bool& out_of_range) {
    const uint32_t e = (__float_as_uint(f) >> 23) & 0xffu;
    const bool bad = e == 0 || e == 255;                     // f is zero/subnormal or non-finite: no residual
    out_of_range |= bad && d != 0.0;
    const int32_t k = (int32_t)(__double_as_longlong(d) - __double_as_longlong((double)f));
    return bad ? 0 : k;
}
__device__ inline double split_join(float f, int32_t k) {   // k == 0 (also every e = 0 / 255 case) returns (double)f itself, -0.0 included
    return __longlong_as_double(__double_as_longlong((double)f) + (long long)k);
}
template <int F> __device__ inline void join_state(const float (&o)[F], const int32_t (&k)[F], double (&s)[F]) {
#pragma unroll
    for (int f = 0; f < F; f++) s[f] = split_join(o[f], k[f]);
}
template <int F> __device__ inline void load_state(const StepBuffers& b, int64_t i, double (&s)[F]) {
    float o[F]; int32_t k[F];
    load_obs(b.obs, i, o);
    load_i32((const int32_t*)b.state, i, k);
    join_state<F>(o, k, s);
}
// Stores the residual half; the caller stores o (= the float-rounded state) through store_obs.
template <int F, bool NT = kNtDefault> __device__ inline void store_state(const StepBuffers& b, int64_t i, const double (&s)[F], const float (&o)[F], int32_t* error_flag) {
    int32_t k[F];
    bool out_of_range = false;
#pragma unroll
    for (int f = 0; f < F; f++) k[f] = split_residual(s[f], o[f], out_of_range);
    if (out_of_range) atomicOr(error_flag, 4);
    store_i32<NT>((int32_t*)b.state, i, k);
}

// Reset state of env i for episode `ep` (already range-checked).  `rs` is the engine's reset source, read from device memory by the
// calling kernel (so a recorded launch follows re-seeds and tape replacements).
template <int KIND, typename T>
__device__ inline void reset_state(const StepParams& p, const ResetSource& rs, int64_t i, int64_t ep, T (&s)[EnvTraits<KIND>::F]) {
    if constexpr (KIND == KIND_CARTPOLE) {
        float r[4];
        if (p.reset_mode == GYMHIP_RESET_TAPE) {
#if GYMHIP_TAPE_NT
            f32x4 v = __builtin_nontemporal_load((const f32x4*)rs.tape + (ep * p.n + i));   // each row is read once: do not keep its line
#else
            f32x4 v = ((const f32x4*)rs.tape)[ep * p.n + i];
#endif
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
            counter_reset_draw(rs.seed, p.env_offset + i, ep, r);
        }
#pragma unroll
        for (int f = 0; f < 4; f++) s[f] = (T)r[f];  // float widened into the double state (cartpole.cpp:23-26)
    } else {
        s[0] = (T)-0.5; s[1] = (T)0.0;  // mountain_car.cpp:38-42
    }
}
// MT19937 mode: the four raw generator outputs of the env's reset row (mt_draw_row) through libstdc++'s
// uniform_real_distribution<float>(-0.05f, 0.05f) arithmetic (cartpole.cpp:21-26).
template <typename T> __device__ inline void reset_state_mt(const uint32_t (&raw)[4], T (&s)[4]) {
#pragma unroll
    for (int f = 0; f < 4; f++) s[f] = (T)canonical_to_range(raw[f]);
}
// The env's new episode: picks the reset source's row for `cursor`, advances the cursor, writes the state.  TAPE / COUNTER_RNG /
// MountainCar only (MT engines: mt_draw_row + reset_state_mt).
template <int KIND, typename T>
__device__ inline void begin_episode(const StepParams& p, const ResetSource& rs, int64_t i, uint32_t& cursor, T (&s)[EnvTraits<KIND>::F]) {
    int64_t e = cursor;
    if (KIND == KIND_CARTPOLE && p.reset_mode == GYMHIP_RESET_TAPE) {
        if (e >= rs.tape_episodes) { atomicOr(p.error_flag, 1); e = e % rs.tape_episodes; }
        cursor = CounterWord<KIND>::next_cursor(cursor, p.error_flag);
    } else if (KIND == KIND_CARTPOLE) {
        // stateless: (seed, env, the env's episode index) keys the draw; no host-side step counter, so a recorded step replays
        // correctly from a HIP graph
        cursor = CounterWord<KIND>::next_cursor_wrapping(cursor, p.error_flag);
    }
    reset_state<KIND, T>(p, rs, i, e, s);
}

// Block reduction of per-thread double columns in a fixed order (deterministic), through LDS: every thread parks its COLS
// values column-major, 16 threads per column then sum kBlock/16 consecutive entries each, and thread c < COLS sums the
// 16 segment totals of column c.  ~COLS LDS stores + kBlock/16 + 16 LDS loads/adds per thread — the earlier shuffle tree
// (6 levels x COLS 64-bit cross-lane moves) cost about as much as two MountainCar env-steps per thread, which is what a
// block's epilogue weighed at the 131,072 .. 1 M envs per GPU that BASELINE configs[2] specifies.
// Returns the column total in threads tid < COLS (0 elsewhere).  Needs 16 * COLS <= kBlock.
constexpr int kSeg = kBlock / 16;          // entries per segment
constexpr int kSegPad = kSeg + 1;          // +1 double: the 16 segments of a column start in different LDS banks
template <int COLS>
__device__ inline double block_reduce_cols(const double (&v)[COLS]) {
    static_assert(16 * COLS <= kBlock, "one 16-thread group per column");
    __shared__ double red[COLS][16 * kSegPad];
    __shared__ double seg_tot[COLS][16];
    const int tid = threadIdx.x;
    {
        const int seg = tid / kSeg, k = tid - seg * kSeg;
#pragma unroll
        for (int c = 0; c < COLS; c++) red[c][seg * kSegPad + k] = v[c];
    }
    __syncthreads();
    {
        const int c = tid >> 4, seg = tid & 15;
        if (c < COLS) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < kSeg; k++) t += red[c][seg * kSegPad + k];
            seg_tot[c][seg] = t;
        }
    }
    __syncthreads();
    double tot = 0;
    if (tid < COLS) {
#pragma unroll
        for (int k = 0; k < 16; k++) tot += seg_tot[tid][k];
    }
    return tot;
}
template <int COLS>
__device__ inline void block_reduce_store(double (&v)[COLS], double* out_row, bool accumulate) {
    double t = block_reduce_cols<COLS>(v);
    if (threadIdx.x < COLS) {
        if (accumulate) t += out_row[threadIdx.x];
        out_row[threadIdx.x] = t;
    }
}

// partials[rows][kMaxCols] -> column totals, by exactly kFoldThreads (256) threads of a block in a fixed order: 16 row groups
// x 16 columns, four independent accumulators per thread so that the row loads overlap, then 16 -> 1.  Used by the stand-alone
// reduce kernel (split step: the totals are the all-reduce payload) AND by the normalise kernel's prologue (unsplit step: no
// separate launch) — one function, one summation order, so the split and the unsplit step agree bit for bit.
// Thread tid < cols returns the total of column tid.  `red` is LDS [16][kMaxCols].
constexpr int kFoldThreads = 256;
static_assert(kBlock >= kFoldThreads, "the normalise kernel's prologue needs 256 threads");
__device__ inline double fold_partial_rows(const double* __restrict__ partials, int rows, int cols, double (*red)[kMaxCols], int tid) {
    if (tid < kFoldThreads) {
        const int c = tid & 15, g = tid >> 4;
        // 16 independent loads in flight per thread: at the <= 256 rows of BASELINE configs[2]'s sizes the whole fold is ONE round
        // trip to L2 (with 4 in flight it was four, ~3 us of a 6 us kernel at 131,072 envs)
        constexpr int U = 16;
        double t[U];
#pragma unroll
        for (int k = 0; k < U; k++) t[k] = 0.0;
        if (c < cols) {
            int r = g;
            for (; r + (U - 1) * 16 < rows; r += U * 16) {
#pragma unroll
                for (int k = 0; k < U; k++) t[k] += partials[(int64_t)(r + k * 16) * kMaxCols + c];
            }
            for (; r < rows; r += 16) t[0] += partials[(int64_t)r * kMaxCols + c];
        }
#pragma unroll
        for (int w = U / 2; w > 0; w >>= 1) {
#pragma unroll
            for (int k = 0; k < w; k++) t[k] += t[k + w];
        }
        red[g][c] = t[0];
    }
    __syncthreads();
    double s = 0;
    if (tid < cols) {
#pragma unroll
        for (int k = 0; k < 16; k++) s += red[k][tid];
    }
    return s;
}

// Kernel F (split step only): partials -> moments[0] = n_local, moments[1 + c] = column totals; one block.
// A separate launch on purpose when an all-reduce sits between step and normalise.  (Folding it into the step kernel — "the
// block that finishes last reduces" — needs a device-scope release/acquire fence per block, and on MI355X each such fence
// writes back and invalidates an XCD's whole L2: measured at +80 us per step at 16.7 M envs.)
__global__ __launch_bounds__(kFoldThreads) void reduce_partials_kernel(const double* __restrict__ partials, int rows, int cols, double n_local, double* moments) {
    __shared__ double red[16][kMaxCols];
    const double s = fold_partial_rows(partials, rows, cols, red, threadIdx.x);
    if ((int)threadIdx.x < cols) moments[1 + threadIdx.x] = s;
    if (threadIdx.x == 0) moments[0] = n_local;
}

// ROWS exchange form (gymhip_set_exchange_form): the ranks all-reduce the partial ROWS themselves — [pad_rows][kMaxCols] doubles,
// tens of KB, still latency-bound — and the normalise kernel folds the summed rows in its prologue, as it does on a single GPU: no
// reduce launch between the step and the collective.  For that the payload must be self-contained: the last column of row 0 carries
// this rank's env count (the other rows' last column is 0), and rows [gridDim.x, pad_rows) — which no block of this launch owns but
// an earlier Updateable::update pass may have used — are zeroed.  Thread c < kMaxCols of every block.
__device__ inline void publish_row_padding(const StepParams& p, int c) {
    if (p.pad_rows <= 0 || c >= kMaxCols) return;
    if (c == kMaxCols - 1) p.partials[(int64_t)blockIdx.x * kMaxCols + c] = blockIdx.x == 0 ? (double)p.n : 0.0;
    for (int r = blockIdx.x + gridDim.x; r < p.pad_rows; r += gridDim.x) p.partials[(int64_t)r * kMaxCols + c] = 0.0;
}

// ------------------------------------------------------------------------------------------------------------
// Kernel A: one VecEnv::step for every env.
// ------------------------------------------------------------------------------------------------------------
template <int KIND, bool LOCKSTEP, typename ActT, bool NORM, bool NT, int KEEP, bool MT = false>   // KEEP: the 4-B lanes an NT kernel keeps in the Infinity Cache (one of kPlans); MT: per-env std::mt19937 resets
#ifndef GYMHIP_STEP_WAVES
#define GYMHIP_STEP_WAVES 1   // experiment knob: min waves per SIMD the step kernel is compiled for (caps its VGPRs)
#endif
__global__ __launch_bounds__(kBlock, MT ? 4 : GYMHIP_STEP_WAVES) void step_kernel(const StepParams p) {   // MT: held to 128 VGPRs (4 waves per SIMD)
    static_assert(!MT || KIND == KIND_CARTPOLE, "only CartPole draws its reset state");
    constexpr int F = EnvTraits<KIND>::F;
    __shared__ __align__(16) uint32_t mt_lds[MT ? kBlock / 64 : 1][MT ? kMtWords : 1];   // one generator block per wave (mt_draw_row; 16-B aligned rows for the LDS-DMA)
    ResetSource rs{};
    if constexpr (KIND == KIND_CARTPOLE && !MT) rs = *p.src;   // uniform, issued with the first loads; only the reset path uses it
    using T = typename std::conditional<LOCKSTEP, double, float>::type;
    constexpr int NCOL = NORM ? 2 * F + 2 : 1;
    double acc[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; c++) acc[c] = 0.0;
    double ep[3] = {0.0, 0.0, 0.0};
    float shift[F + 1];
    if constexpr (NORM) {
        int idx[F + 1];
#pragma unroll
        for (int f = 0; f < F; f++) idx[f] = f;
        idx[F] = 2 * F + 1;
        stats_reader_values<F + 1>(p, idx, shift);
    }
    const ActT* actions = (const ActT*)p.actions;
    const StepBuffers& b = p.buf;
    const bool time_limit = p.wrappers & GYMHIP_WRAP_TIME_LIMIT;
    const bool clip = p.wrappers & GYMHIP_WRAP_CLIP_REWARD;
    const bool ret_lane = (p.wrappers & GYMHIP_WRAP_NORM_REWARD) && (p.wrappers & GYMHIP_WRAP_NORM_TRAINING);
    const bool auto_reset = !(p.wrappers & GYMHIP_WRAP_NO_AUTO_RESET);   // sync_env.h:28,92

    // Everything one env-step reads, as raw registers: the loads are issued as a batch and nothing depends on them
    // until the next iteration's dynamics start.
    struct Inputs {
        float o[F];
        int32_t k[LOCKSTEP ? F : 1];
        int action;
        uint32_t w;       // packed step counter / episode record / tape cursor (gymhip.h)
        float ret;
    };
    // `i` must be a valid env index.  A lane with no next iteration issues no loads (an earlier form re-read the lane's own
    // inputs unconditionally: one wasted 40-B read per THREAD per launch, 5 % of the fetch traffic at 2048 x 512 threads).
    // Residency plan (NT kernels only; plan_residency): the 4-B lanes in KEEP are accessed with the default (temporal) policy, so
    // that they stay in the 256 MiB Infinity Cache from one step to the next, while everything else streams past it non-temporally.
    constexpr bool NT_COUNTER = NT && !(KEEP & kKeepCounter), NT_REWARD = NT && !(KEEP & kKeepReward), NT_DONE = NT && !(KEEP & kKeepDone),
                   NT_RETURNS = NT && !(KEEP & kKeepReturns);
    auto issue_loads = [&](int64_t i, Inputs& in) {
        load_obs<NT>(b.obs, i, in.o);
        if constexpr (LOCKSTEP) load_i32<NT>((const int32_t*)b.state, i, in.k);
        in.action = (int)ld_stream<NT>(actions, i);
        in.w = ld_stream<NT_COUNTER>(b.counter, i);
        if constexpr (NORM) in.ret = ret_lane ? ld_stream<NT_RETURNS>(b.returns, i) : 0.f;
    };

    // Software pipeline: iteration k+1's loads are issued before iteration k's arithmetic, so they are in flight under
    // the dynamics (~500 VALU instructions) and the stores.  Measured on MI355X (16.7 M envs, lockstep): COUNTER_RNG
    // resets 0.282 -> 0.270 ms; TAPE resets unchanged at 0.284 ms — there the reset-tape gather's over-fetch (a 16-B row
    // per done env drags in a whole memory burst) sets the time, not latency.  Issuing the gather BEFORE the next loads
    // (so that, with the in-order vmcnt counter, its wait would not cover them) was tried and is slower for both.
    const int64_t stride = (int64_t)gridDim.x * p.lanes;
    const int64_t first = (int)threadIdx.x < p.lanes ? (int64_t)blockIdx.x * p.lanes + threadIdx.x : p.n;   // threads beyond `lanes` own no env
    int64_t i = first;
    Inputs cur, nxt;
    if (i < p.n) issue_loads(i, cur);
    for (; i < p.n; i += stride) {
#if GYMHIP_PIPE
        if (i + stride < p.n) issue_loads(i + stride, nxt);
#else
        if (i != first) issue_loads(i, cur);
#endif
        T s[F];
        if constexpr (LOCKSTEP) {
            join_state<F>(cur.o, cur.k, s);
        } else {
#pragma unroll
            for (int f = 0; f < F; f++) s[f] = cur.o[f];
        }
        const int action = cur.action;
        const uint32_t w = cur.w;
        int t = CounterWord<KIND>::steps_now(w);
        uint32_t cursor = CounterWord<KIND>::cursor(w);

        // auto_reset off: an env that finished and has not been reset (gymhip_reset_index) is frozen — stepping it is the
        // misuse the reference reports on stderr (monitor.h:75-77); state and record stay, reward 0, done 1, error bit 3
        const bool frozen = !auto_reset && (w >> 31);
        // Forecast (MT19937 engines): whether this step ends the episode is known before the dynamics run, exactly — x and theta advance by
        // tau times the OLD velocities.  The reset row's gather is therefore issued HERE, in parallel with the arithmetic, instead of after
        // it on the iteration's critical path, and an env about to need row 0 starts its block's read instead (mt_prefetch_block).
        // (Tape engines: the same early gather measures +0.7 % in lockstep mode and costs the float kernels 9 % — not compiled in.)
        int mt_ahead = -1;
        bool early = false;                 // this lane's reset row is in `early_row`
        u32x4 early_row = {0u, 0u, 0u, 0u};
        if constexpr (MT) {
            const uint32_t pos = cursor % (uint32_t)kMtRows;
            const bool ahead = auto_reset && ((time_limit && t + 1 >= p.max_episode_steps) || cartpole_done_ahead(s, t));
            if (ahead && pos != 0u) { early_row = *(const u32x4*)(p.mt_state + i * kMtWords + pos * 4u); early = true; }
            mt_ahead = mt_prefetch_block(ahead && pos == 0u, p.mt_state, i, mt_lds[threadIdx.x >> 6]);
        }
        bool env_done = false;
        if (!frozen) {
            if constexpr (KIND == KIND_CARTPOLE) env_done = cartpole_step(s, action, t); else env_done = mountaincar_step(s, action);
            t += 1;
        }
        bool done = env_done, trunc = false;
        if (time_limit && t >= p.max_episode_steps) { trunc = !env_done; done = true; }   // time_limit.h:30-34
        float reward = EnvTraits<KIND>::step_reward;
        if (clip) reward = reward == 0.f ? reward : (reward > 0.f ? 1.f : -1.f);           // clip_reward.h:22-24
        if (frozen) { atomicOr(p.error_flag, 8); t = CounterWord<KIND>::steps_field(w); done = true; trunc = (w >> 30) & 1u; reward = 0.f; }

        uint32_t mt_raw[4];
        if constexpr (MT) {
            mt_draw_row(done && !frozen && auto_reset && !early, cursor, p.mt_state, i, mt_lds[threadIdx.x >> 6], mt_raw, mt_ahead);   // wave-collective
            if (early) { mt_raw[0] = mt_temper(early_row.x); mt_raw[1] = mt_temper(early_row.y); mt_raw[2] = mt_temper(early_row.z); mt_raw[3] = mt_temper(early_row.w); }
        }
        if (done && !frozen) {
            // Monitor (monitor.h:78-97): every step of these envs pays the same reward, so the ordered float sum
            // of t copies equals float(t)*reward exactly (t < 2^24).  The finished episode's (length, truncated)
            // and the reset-tape cursor live in the counter word, which is rewritten every step anyway: no extra
            // HBM traffic and no scattered read-modify-write on the done path.
            ep[0] += 1.0; ep[1] += (double)((float)t * reward); ep[2] += (double)t;
            if (auto_reset) {                  // sync_env.h:92-100: terminal obs replaced by the reset obs (else: kept, no draw consumed)
                if constexpr (MT) { reset_state_mt<T>(mt_raw, s); cursor = CounterWord<KIND>::next_cursor_mt(cursor); }
                else begin_episode<KIND, T>(p, rs, i, cursor, s);
            }
        }

        float o[F];
#pragma unroll
        for (int f = 0; f < F; f++) o[f] = (float)s[f];
        if constexpr (LOCKSTEP) store_state<F, NT>(b, i, s, o, p.error_flag);
        store_obs<NT>(b.obs, i, o);
        st_stream<NT_COUNTER>(b.counter, i, CounterWord<KIND>::pack(t, cursor, done, trunc));
        st_stream<NT_REWARD>(b.reward, i, reward);
        st_stream<NT_DONE>(b.done, i, done ? 1.f : 0.f);

        if constexpr (NORM) {
#pragma unroll
            for (int f = 0; f < F; f++) { double d = (double)o[f] - (double)shift[f]; acc[f] += d; acc[F + f] += d * d; }
            if (ret_lane) {
                float r = cur.ret * p.gamma + reward;              // vec_normalize.h:132
                double d = (double)r - (double)shift[F];
                acc[2 * F] += d; acc[2 * F + 1] += d * d;
                st_stream<NT_RETURNS>(b.returns, i, done ? 0.f : r);   // vec_normalize.h:175 (applied after the RMS update)
            }
        }
#if GYMHIP_PIPE
        cur = nxt;
#endif
    }

    // ONE block reduction for the batch-moment columns and the three episode counters; thread c < COLS owns column c.
    if constexpr (NORM) {
        double cols[2 * F + 5];
#pragma unroll
        for (int c = 0; c < 2 * F + 2; c++) cols[c] = acc[c];
        cols[2 * F + 2] = ep[0]; cols[2 * F + 3] = ep[1]; cols[2 * F + 4] = ep[2];
        const double t = block_reduce_cols<2 * F + 5>(cols);
        const int c = threadIdx.x;
        if (c < 2 * F + 5) {
            p.partials[(int64_t)blockIdx.x * kMaxCols + c] = t;
            // cumulative per-block episode counters (owner-block read-modify-write: no atomics, deterministic)
            if (c >= 2 * F + 2) p.ep_partials[(int64_t)blockIdx.x * 4 + (c - (2 * F + 2))] += t;
        }
        publish_row_padding(p, c);
    } else {
        block_reduce_store<3>(ep, p.ep_partials + (int64_t)blockIdx.x * 4, true);
    }
}

// VecEnv::reset(): every env starts a new episode.  In TAPE mode env i takes the tape row at its own cursor and advances it —
// a CartPoleEnv's generator keeps running across resets (cartpole.cpp:21-26), so a second reset() draws the NEXT four
// values, not the first four again.  A fresh workspace (or a freshly drawn tape) has every cursor at 0.
template <int KIND, bool LOCKSTEP, bool NORM, bool MT = false>
__global__ __launch_bounds__(kBlock) void reset_kernel(const StepParams p) {
    constexpr int F = EnvTraits<KIND>::F;
    using T = typename std::conditional<LOCKSTEP, double, float>::type;
    __shared__ uint32_t mt_lds[MT ? kBlock / 64 : 1][MT ? kMtWords : 1];
    ResetSource rs{};
    if constexpr (KIND == KIND_CARTPOLE && !MT) rs = *p.src;
    constexpr int NCOL = NORM ? 2 * F + 2 : 1;
    double acc[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; c++) acc[c] = 0.0;
    float shift[F];
    if constexpr (NORM) {
        const float* stats_cur = stats_reader_slot(p);
#pragma unroll
        for (int f = 0; f < F; f++) shift[f] = stats_cur[f];
    }
    const StepBuffers& b = p.buf;
    for (int64_t i = (int)threadIdx.x < p.lanes ? (int64_t)blockIdx.x * p.lanes + threadIdx.x : p.n; i < p.n; i += (int64_t)gridDim.x * p.lanes) {
        T s[F];
        uint32_t cursor = 0;
        if constexpr (KIND == KIND_CARTPOLE) cursor = CounterWord<KIND>::cursor(b.counter[i]);   // the env's episode index: the tape row, the generator row or the counter-RNG key
        if constexpr (MT) {
            uint32_t mt_raw[4];
            mt_draw_row(true, cursor, p.mt_state, i, mt_lds[threadIdx.x >> 6], mt_raw);   // wave-collective (the loop body is converged)
            reset_state_mt<T>(mt_raw, s);
            cursor = CounterWord<KIND>::next_cursor_mt(cursor);
        } else {
            begin_episode<KIND, T>(p, rs, i, cursor, s);
        }
        float o[F];
#pragma unroll
        for (int f = 0; f < F; f++) o[f] = (float)s[f];
        if constexpr (LOCKSTEP) store_state<F>(b, i, s, o, p.error_flag);
        store_obs(b.obs, i, o);
        b.counter[i] = CounterWord<KIND>::pack(0, cursor, false, false);
        b.reward[i] = 0.f;
        b.done[i] = 0.f;
        if (b.returns) b.returns[i] = 0.f;                        // vec_normalize.h:147
        if constexpr (NORM) {
#pragma unroll
            for (int f = 0; f < F; f++) { double d = (double)o[f] - (double)shift[f]; acc[f] += d; acc[F + f] += d * d; }
        }
    }
    if constexpr (NORM) {
        double cols[2 * F + 5];
#pragma unroll
        for (int c = 0; c < 2 * F + 2; c++) cols[c] = acc[c];
        cols[2 * F + 2] = 0; cols[2 * F + 3] = 0; cols[2 * F + 4] = 0;
        block_reduce_store<2 * F + 5>(cols, p.partials + (int64_t)blockIdx.x * kMaxCols, false);
        publish_row_padding(p, threadIdx.x);
    }
}


// RunningMeanStd::updateFromMoment in float32, evaluation order as written (running_mean_std.h:50-57, A9).
__device__ inline void rms_update(float& mean, float& var, float count, float bm, float bv, float nf, float tot) {
    float delta = bm - mean;
    float new_mean = mean + delta * nf / tot;
    float ma = var * count;
    float mb = bv * nf;
    float m2 = ma + mb + delta * delta * count * nf / tot;
    var = m2 / tot;
    mean = new_mean;
}

// Kernel B: apply the (all-reduced) batch moments to the running statistics, then normalise (vec_normalize.h:118-129).
// Every block derives the new statistics redundantly (identical instructions on identical inputs); block 0 publishes
// them into the other ping-pong slot, which only the NEXT launches read.
template <int F, bool NT, bool KEEP_REWARD = false>   // KEEP_REWARD: an NT kernel leaves the reward lane in the Infinity Cache (plan_residency)
__global__ __launch_bounds__(kBlock) void normalize_kernel(const NormParams q) {
    // A thread's unit of streaming work is a GROUP: F 16-B observation vectors (one from each of F equal chunks of the first
    // 4 * ng envs' rows) + one 16-B reward vector; ng = n / 4 groups cover exactly the first 4 * ng envs.  The first
    // group's loads are issued before anything else, so that they are in flight under the statistics prologue below (at the
    // 131,072 .. 1 M envs per GPU of BASELINE configs[2] the prologue is a third of this kernel's time).
    struct Group { float o[F][4]; float r[4]; };
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x, nthr = (int64_t)gridDim.x * kBlock;
    const int64_t ng = q.n / 4;
    auto load_group = [&](int64_t j, Group& g) {
        if (q.norm_obs) {
#pragma unroll
            for (int v = 0; v < F; v++) load_obs<NT>(q.obs, v * ng + j, g.o[v]);   // F chunks of ng vectors: every load instruction is lane-contiguous
        }
        if (q.norm_reward) {
            load_obs<NT && !KEEP_REWARD>(q.reward, j, g.r);
        }
    };
    Group cur;
    if (tid < ng) load_group(tid, cur);

    // Which of the two ping-pong slots is current is device state, handed over between kernels so that no launch carries a
    // host-side slot index (a recorded step then replays correctly from a HIP graph):  the kernel that ran before this one in the
    // stream (step / reset / batch-moments) read the slot from tick_w and copied it into tick_r; every block here reads tick_r,
    // block 0 publishes into the OTHER slot and then sets tick_w = tick_r + 1.  Within a kernel nobody reads what that kernel writes.
    const int32_t tick = *q.tick_r;
    float st[2 * F + 4];                                     // both slots are read, independently of the tick, and picked afterwards:
    {                                                        // one dependent round trip less in a kernel that is launch-latency-bound at BASELINE configs[2]'s sizes
        float s0[2 * F + 4], s1[2 * F + 4];
#pragma unroll
        for (int k = 0; k < 2 * F + 4; k++) { s0[k] = q.stats[k]; s1[k] = q.stats[16 + k]; }
#pragma unroll
        for (int k = 0; k < 2 * F + 4; k++) st[k] = (tick & 1) ? s1[k] : s0[k];
    }
    float mean[F], var[F];
#pragma unroll
    for (int f = 0; f < F; f++) { mean[f] = st[f]; var[f] = st[F + f]; }
    float ocount = st[2 * F], rmean = st[2 * F + 1], rvar = st[2 * F + 2], rcount = st[2 * F + 3];
    // The batch moments: either already reduced (and all-reduced over ranks) in q.moments — the split step — or, for the
    // unsplit single-GPU step, summed here from the step kernel's per-block partial rows by every block redundantly (the rows
    // sit in L2; same function and order as reduce_partials_kernel, so both forms give identical bits).  Saves one launch.
    __shared__ double fold_red[16][kMaxCols];
    __shared__ double mom[kMaxCols];
    if (q.partials) {
        const double tot = fold_partial_rows(q.partials, q.rows, q.n_from_rows ? kMaxCols : 2 * F + 5, fold_red, threadIdx.x);
        if ((int)threadIdx.x < 2 * F + 5) { mom[1 + threadIdx.x] = tot; if (blockIdx.x == 0) q.moments_out[1 + threadIdx.x] = tot; }
        if (q.n_from_rows) { if (threadIdx.x == kMaxCols - 1) { mom[0] = tot; if (blockIdx.x == 0) q.moments_out[0] = tot; } }   // all ranks' env counts, summed by the all-reduce
        else if (threadIdx.x == 0) { mom[0] = q.n_local; if (blockIdx.x == 0) q.moments_out[0] = q.n_local; }
    } else if ((int)threadIdx.x < 2 * F + 6) {
        mom[threadIdx.x] = q.moments[threadIdx.x];
    }
    __syncthreads();
    const double n = mom[0];
    const float nf = (float)n;
    if (q.update_obs) {
        const float tot = ocount + nf;
#pragma unroll
        for (int f = 0; f < F; f++) {
            double m1 = mom[1 + f] / n, m2 = mom[1 + F + f] / n;
            double v = m2 - m1 * m1;
            float bm = (float)((double)mean[f] + m1), bv = (float)(v > 0 ? v : 0.0);
            rms_update(mean[f], var[f], ocount, bm, bv, nf, tot);
        }
        ocount = tot;
    }
    if (q.update_ret) {
        const float tot = rcount + nf;
        double m1 = mom[1 + 2 * F] / n, m2 = mom[2 + 2 * F] / n;
        double v = m2 - m1 * m1;
        float bm = (float)((double)rmean + m1), bv = (float)(v > 0 ? v : 0.0);
        rms_update(rmean, rvar, rcount, bm, bv, nf, tot);
        rcount = tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float* outs[2] = {q.stats + 16 * ((tick & 1) ^ 1), q.stats + 32};
        *q.tick_w = tick + 1;
        for (int k = 0; k < 2; k++) {
            float* o = outs[k];
#pragma unroll
            for (int f = 0; f < F; f++) { o[f] = mean[f]; o[F + f] = var[f]; }
            o[2 * F] = ocount; o[2 * F + 1] = rmean; o[2 * F + 2] = rvar; o[2 * F + 3] = rcount;
        }
        if (q.ep_totals_global) {
            q.ep_totals_global[0] += mom[2 * F + 3]; q.ep_totals_global[1] += mom[2 * F + 4]; q.ep_totals_global[2] += mom[2 * F + 5];
        }
    }
    float sd[F];
#pragma unroll
    for (int f = 0; f < F; f++) sd[f] = sqrtf(var[f] + q.epsilon);   // divisor, not reciprocal: keep the reference's divide
    const float rsd = sqrtf(rvar + q.epsilon);
    auto norm_o = [&](float x, int f) { return fminf(fmaxf((x - mean[f]) / sd[f], -q.clip_obs), q.clip_obs); };       // vec_normalize.h:118-124
    auto norm_r = [&](float x) { return fminf(fmaxf(x / rsd, -q.clip_reward), q.clip_reward); };                      // vec_normalize.h:126-129
    // One streaming pass, software-pipelined one group ahead.  Element c of an observation vector is feature c % F (4 % F == 0).
    for (int64_t j = tid; j < ng; j += nthr) {
        Group nxt;
        if (j + nthr < ng) load_group(j + nthr, nxt);
        if (q.norm_obs) {
#pragma unroll
            for (int v = 0; v < F; v++) {
#pragma unroll
                for (int c = 0; c < 4; c++) cur.o[v][c] = norm_o(cur.o[v][c], c % F);
                store_obs<NT>(q.obs_norm, v * ng + j, cur.o[v]);
            }
        }
        if (q.norm_reward) {
#pragma unroll
            for (int c = 0; c < 4; c++) cur.r[c] = norm_r(cur.r[c]);
            store_obs<NT && !KEEP_REWARD>(q.reward, j, cur.r);
        }
        cur = nxt;
    }
    for (int64_t i = ng * 4 + tid; i < q.n; i += nthr) {               // the last n % 4 envs
        if (q.norm_obs) {
#pragma unroll
            for (int f = 0; f < F; f++) q.obs_norm[i * F + f] = norm_o(q.obs[i * F + f], f);
        }
        if (q.norm_reward) q.reward[i] = norm_r(q.reward[i]);
    }
}

// VecEnv::reset(int index) (sync_env.h:128-139) under VecNormalize::reset(int index) (vec_normalize.h:179-189): one env is
// reset, its return lane cleared, and its observation normalised with the CURRENT statistics (no RMS update).  Monitor
// allows early resets (monitor.h:49, allowEarlyResets): the unfinished episode is dropped without a record.
template <int KIND, bool LOCKSTEP, bool MT = false>
__global__ __launch_bounds__(64) void reset_index_kernel(const StepParams p, int64_t i, float* obs_norm, float clip_obs, float epsilon) {
    constexpr int F = EnvTraits<KIND>::F;
    using T = typename std::conditional<LOCKSTEP, double, float>::type;
    const StepBuffers& b = p.buf;
    __shared__ uint32_t mt_lds[MT ? kMtWords : 1];
    const bool owner = blockIdx.x == 0 && threadIdx.x == 0;
    uint32_t cursor = CounterWord<KIND>::cursor(b.counter[i]);
    T s[F];
    if constexpr (MT) {
        if (blockIdx.x != 0) return;
        uint32_t mt_raw[4];
        mt_draw_row(owner, cursor, p.mt_state, i, mt_lds, mt_raw);   // the whole wave helps when the env's generator block has to be regenerated
        if (!owner) return;
        reset_state_mt<T>(mt_raw, s);
        cursor = CounterWord<KIND>::next_cursor_mt(cursor);
    } else {
        if (!owner) return;
        ResetSource rs{};
        if constexpr (KIND == KIND_CARTPOLE) rs = *p.src;
        begin_episode<KIND, T>(p, rs, i, cursor, s);
    }
    float o[F];
#pragma unroll
    for (int f = 0; f < F; f++) o[f] = (float)s[f];
    if constexpr (LOCKSTEP) store_state<F>(b, i, s, o, p.error_flag);
    store_obs(b.obs, i, o);
    b.counter[i] = CounterWord<KIND>::pack(0, cursor, false, false);
    if (b.returns) b.returns[i] = 0.f;                                   // vec_normalize.h:182
    if (obs_norm) {                                                      // normalizeObs<true> (vec_normalize.h:118-123)
        const float* stats_cur = p.stats + 16 * (*p.tick_w & 1);         // a pure reader: no hand-off
#pragma unroll
        for (int f = 0; f < F; f++)
            obs_norm[i * F + f] = fminf(fmaxf((o[f] - stats_cur[f]) / sqrtf(stats_cur[F + f] + epsilon), -clip_obs), clip_obs);
    }
}

// Batch moments of caller rows x f32 [rows][width] for Updateable::update (vec_normalize.h:191-206): shifted sums into
// partial columns col0 .. col0+width-1 (S1) and col0_sq .. (S2); every other column of the row is zeroed.
__global__ __launch_bounds__(kBlock) void batch_moments_kernel(const float* __restrict__ x, int64_t rows, int width, const float* stats, const int32_t* tick_w,
                                                               int32_t* tick_r, int stat_off, int col0, int col0_sq, double* partials) {
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float sh[4] = {0, 0, 0, 0};
    const int32_t tick = *tick_w;                                   // a reader of the statistics: slot from tick_w, handed on through tick_r
    if (blockIdx.x == 0 && threadIdx.x == 0) *tick_r = tick;
    const float* shift = stats + 16 * (tick & 1) + stat_off;
    for (int f = 0; f < width; f++) sh[f] = shift[f];
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kBlock)
        for (int f = 0; f < width; f++) { const double d = (double)x[r * width + f] - (double)sh[f]; s1[f] += d; s2[f] += d * d; }
    double cols[kMaxCols];
#pragma unroll
    for (int c = 0; c < kMaxCols; c++) cols[c] = 0.0;
    for (int f = 0; f < width; f++) { cols[col0 + f] = s1[f]; cols[col0_sq + f] = s2[f]; }
    block_reduce_store<kMaxCols>(cols, partials + (int64_t)blockIdx.x * kMaxCols, false);
}

// K steps in one launch; state lives in registers between steps.
template <int KIND, bool LOCKSTEP, bool MT = false>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const StepParams p, const uint8_t* __restrict__ tape, int64_t K,
                                                          uint8_t* rec_done, float* rec_reward, uint8_t* rec_trunc, float* rec_obs) {
    constexpr int F = EnvTraits<KIND>::F;
    using T = typename std::conditional<LOCKSTEP, double, float>::type;
    const StepBuffers& b = p.buf;
    const bool time_limit = p.wrappers & GYMHIP_WRAP_TIME_LIMIT;
    const bool clip = p.wrappers & GYMHIP_WRAP_CLIP_REWARD;
    __shared__ uint32_t mt_lds[MT ? kBlock / 64 : 1][MT ? kMtWords : 1];
    ResetSource rs{};
    if constexpr (KIND == KIND_CARTPOLE && !MT) rs = *p.src;
    double ep[3] = {0.0, 0.0, 0.0};
    for (int64_t i = (int)threadIdx.x < p.lanes ? (int64_t)blockIdx.x * p.lanes + threadIdx.x : p.n; i < p.n; i += (int64_t)gridDim.x * p.lanes) {
        T s[F];
        if constexpr (LOCKSTEP) load_state<F>(b, i, s); else load_obs(b.obs, i, s);
        const uint32_t w = b.counter[i];
        int t = CounterWord<KIND>::steps_now(w);
        uint32_t cursor = CounterWord<KIND>::cursor(w);
        float reward = 0.f; bool done = false; uint32_t rec = w;
        // the action tape is read four steps ahead: at one wave per SIMD (N = 65,536) nothing else hides the load latency
        constexpr int AHEAD = 4;
        int next_action[AHEAD];
#pragma unroll
        for (int d = 0; d < AHEAD; d++) next_action[d] = d < K ? (int)tape[(int64_t)d * p.n + i] : 0;
        for (int64_t k = 0; k < K; k++) {
            const int action = next_action[0];
#pragma unroll
            for (int d = 0; d + 1 < AHEAD; d++) next_action[d] = next_action[d + 1];
            next_action[AHEAD - 1] = k + AHEAD < K ? (int)tape[(k + AHEAD) * p.n + i] : 0;
            bool env_done;
            if constexpr (KIND == KIND_CARTPOLE) env_done = cartpole_step(s, action, t); else env_done = mountaincar_step(s, action);
            t += 1;
            done = env_done; bool trunc = false;
            if (time_limit && t >= p.max_episode_steps) { trunc = !env_done; done = true; }
            reward = EnvTraits<KIND>::step_reward;
            if (clip) reward = reward == 0.f ? reward : (reward > 0.f ? 1.f : -1.f);
            uint32_t mt_raw[4];
            if constexpr (MT) mt_draw_row<true>(done, cursor, p.mt_state, i, mt_lds[threadIdx.x >> 6], mt_raw);   // wave-collective: every lane of the wave runs the same K steps
            if (done) {
                ep[0] += 1.0; ep[1] += (double)((float)t * reward); ep[2] += (double)t;
                if constexpr (MT) { reset_state_mt<T>(mt_raw, s); cursor = CounterWord<KIND>::next_cursor_mt(cursor); }
                else begin_episode<KIND, T>(p, rs, i, cursor, s);
            }
            rec = CounterWord<KIND>::pack(t, cursor, done, trunc);
            if (done) t = 0;
            if (rec_done) rec_done[k * p.n + i] = done;
            if (rec_reward) rec_reward[k * p.n + i] = reward;
            if (rec_trunc) rec_trunc[k * p.n + i] = trunc;
            if (rec_obs) {
                float o[F];
#pragma unroll
                for (int f = 0; f < F; f++) o[f] = (float)s[f];
                store_obs(rec_obs + k * p.n * F, i, o);
            }
        }
        float o[F];
#pragma unroll
        for (int f = 0; f < F; f++) o[f] = (float)s[f];
        if constexpr (LOCKSTEP) store_state<F>(b, i, s, o, p.error_flag);
        store_obs(b.obs, i, o);
        b.counter[i] = rec;
        b.reward[i] = reward;
        b.done[i] = done ? 1.f : 0.f;
    }
    block_reduce_store<3>(ep, p.ep_partials + (int64_t)blockIdx.x * 4, true);
}

__global__ __launch_bounds__(kBlock) void reduce_episode_kernel(const double* ep_partials, int rows, const double* extra, double* totals) {
    __shared__ double red[kBlock][3];
    double t[3] = {0, 0, 0};
    for (int r = threadIdx.x; r < rows; r += kBlock) { t[0] += ep_partials[r * 4]; t[1] += ep_partials[r * 4 + 1]; t[2] += ep_partials[r * 4 + 2]; }
    red[threadIdx.x][0] = t[0]; red[threadIdx.x][1] = t[1]; red[threadIdx.x][2] = t[2];
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int k = 0; k < kBlock; k++) s += red[k][threadIdx.x];
        totals[threadIdx.x] = s;
    }
    (void)extra;
}

// Materialise info["episode"] / info["TimeLimit.truncated"] lanes from the step-counter record words (valid where done).
template <int KIND>
__global__ __launch_bounds__(kBlock) void decode_counter_kernel(const uint32_t* counter, int64_t n, int32_t* elapsed, int32_t* len,
                                                                uint8_t* trunc, float* ret, int32_t* cursor) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint32_t w = counter[i];
        const bool fin = w >> 31;
        const int l = fin ? CounterWord<KIND>::steps_field(w) : 0;
        if (elapsed) elapsed[i] = CounterWord<KIND>::steps_now(w);
        if (len) len[i] = l;
        if (trunc) trunc[i] = fin && (w & 0x40000000u) ? 1 : 0;
        if (ret) ret[i] = (float)l * EnvTraits<KIND>::step_reward;   // ClipReward maps +-1 to itself (clip_reward.h:22-24)
        if (cursor) cursor[i] = (int32_t)CounterWord<KIND>::cursor(w);
    }
}

// Exact double state rows [N][F] (LOCKSTEP: obs + residual joined; F32: obs widened) for checkers and checkpoints.
template <int F, bool LOCKSTEP>
__global__ __launch_bounds__(kBlock) void read_state_kernel(const StepBuffers b, int64_t n, double* out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double s[F];
        if constexpr (LOCKSTEP) {
            load_state<F>(b, i, s);
        } else {
            float o[F];
            load_obs(b.obs, i, o);
#pragma unroll
            for (int f = 0; f < F; f++) s[f] = (double)o[f];
        }
#pragma unroll
        for (int f = 0; f < F; f++) out[i * F + f] = s[f];
    }
}

// A (re)drawn tape restarts every env's stream (Env::seed re-seeds the generator, env.h:60-63): cursors back to 0, the
// step counts and episode records of the counter words untouched.
__global__ __launch_bounds__(kBlock) void rewind_tape_cursor_kernel(uint32_t* counter, int64_t n) {
    using W = CounterWord<KIND_CARTPOLE>;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        counter[i] &= ~(W::kCursorMask << W::S);
}

// Largest reset-tape cursor over all envs (rows consumed by the env that consumed most) -> *out (zeroed by the caller).
__global__ __launch_bounds__(kBlock) void tape_cursor_max_kernel(const uint32_t* counter, int64_t n, int32_t* out) {
    using W = CounterWord<KIND_CARTPOLE>;
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) m = max(m, (int)W::cursor(counter[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

__global__ __launch_bounds__(kBlock) void fill_tape_kernel(float4* tape, int64_t n, int64_t episodes, uint64_t seed, int64_t env_offset) {
    const int64_t total = n * episodes;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
        const int64_t epi = j / n, i = j - epi * n;
        float r[4];
        counter_reset_draw(seed, env_offset + i, epi, r);
        tape[j] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// Diagnostic: the step kernel's own sin/cos of the pole angle, in double, before the float narrowing of cartpole.cpp:39-40.
__global__ __launch_bounds__(kBlock) void diag_pole_sincos_kernel(const double* __restrict__ theta, int64_t n, double* sn, double* cs) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) pole_sincos(theta[i], &sn[i], &cs[i]);
}

// ------------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------------
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t state, obs, obs_norm, reward, done, counter, returns;
    size_t stats /*3 slots*/, ticks /*tick_w, tick_r*/, moments, partials, ep_partials, ep_totals, ep_totals_global, error_flag, cursor_max, reset_source, mt_state, total;
};

int obs_dim_of(int kind) { return kind == KIND_CARTPOLE ? 4 : 2; }

bool valid_config(const gymhip_config* c) {
    if (!c || c->struct_size != (int32_t)sizeof(gymhip_config)) return false;
    if (c->env_kind != KIND_CARTPOLE && c->env_kind != KIND_MOUNTAINCAR) return false;
    if (c->precision != GYMHIP_PRECISION_LOCKSTEP_F64 && c->precision != GYMHIP_PRECISION_F32) return false;
    if (c->reset_mode != GYMHIP_RESET_TAPE && c->reset_mode != GYMHIP_RESET_COUNTER_RNG && c->reset_mode != GYMHIP_RESET_MT19937) return false;
    if (c->n_envs <= 0 || c->env_offset < 0) return false;
    if ((c->wrappers & GYMHIP_WRAP_TIME_LIMIT) && c->max_episode_steps <= 0) return false;
    return true;
}

// Launch shape of the step / reset / rollout kernels: `grid` workgroups of kBlock threads of which the first `lanes` own envs.
int grid_for(int64_t n, bool norm, bool cartpole_lockstep, bool streaming, int* lanes) {
    *lanes = kBlock;
    int64_t cap = (!norm && cartpole_lockstep) ? kPlainGrid : kDefaultGrid;
    if (const char* e = getenv("GYMHIP_GRID")) { int v = atoi(e); if (v > 0 && v <= kMaxGrid) return (int)std::min<int64_t>(v, (n + kBlock - 1) / kBlock); }
    // Fewer than 256 full workgroups (BASELINE configs[1]: 65,536 envs = 128 x 512): spread the envs over up to 256 workgroups with
    // fewer active waves each, so that every CU takes part — these sizes are latency-bound, and one wave per SIMD on 256 CUs beats
    // two per SIMD on 128 (measured with a 256-thread build: step 4.47 -> 4.22 us, 500-step rollout 0.867 -> 0.762 us per step).
    const int64_t one_per_thread = (n + kBlock - 1) / kBlock;
    // (Not with the normaliser: its second kernel folds one partial row per workgroup, and more rows cost more than the spreading
    // gains — config 3 at 65,536 envs: 9.4 us with 128 workgroups, 10.0 with 256.)
    if (one_per_thread < 256 && !norm) {
        int64_t per_block = (n + 255) / 256;
        per_block = std::min<int64_t>(kBlock, std::max<int64_t>(64, (per_block + 63) / 64 * 64));
        *lanes = (int)per_block;
        return (int)((n + per_block - 1) / per_block);
    }
    // One env per thread until every CU has a block (256); then up to 8 envs per thread before the grid grows to its cap: at
    // the 131,072 .. 1 M envs per GPU of BASELINE configs[2] fewer blocks mean fewer partial rows for the normalise kernel to fold.
    if (one_per_thread <= 256) return (int)one_per_thread;
    if (!norm && cartpole_lockstep && streaming) return (int)std::min<int64_t>(kStreamGrid, std::max<int64_t>(256, (n + (int64_t)kBlock * 2 - 1) / ((int64_t)kBlock * 2)));
    const int64_t g = (n + (int64_t)kBlock * 8 - 1) / ((int64_t)kBlock * 8);
    return (int)std::min<int64_t>(cap, std::max<int64_t>(256, g));
}

Layout make_layout(const gymhip_config& c) {
    Layout L{};
    const size_t n = (size_t)c.n_envs, F = (size_t)obs_dim_of(c.env_kind);
    const bool lock = c.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    const bool norm = c.wrappers & (GYMHIP_WRAP_NORM_OBS | GYMHIP_WRAP_NORM_REWARD);
    size_t off = 0;
#ifndef GYMHIP_SKEW
#define GYMHIP_SKEW 0
#endif
    // GYMHIP_SKEW (experiment): stagger the stream bases so that equal env indices of different streams do not land on
    // the same HBM channel when the stream sizes are powers of two.
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes) + (size_t)GYMHIP_SKEW; return o; };
    L.obs = take(n * F * 4);
    L.state = lock ? take(n * F * 4) : L.obs;   // LOCKSTEP: int32 residual rows (split-double), else the obs rows
    L.obs_norm = (c.wrappers & GYMHIP_WRAP_NORM_OBS) ? take(n * F * 4) : (size_t)-1;
    L.reward = take(n * 4);
    L.done = take(n * 4);
    L.counter = take(n * 4);
    L.returns = (c.wrappers & GYMHIP_WRAP_NORM_REWARD) ? take(n * 4) : (size_t)-1;
    L.stats = norm ? take(3 * 16 * 4) : (size_t)-1;
    L.ticks = norm ? take(2 * 4) : (size_t)-1;
    L.moments = take(kMaxCols * 8);
    L.partials = norm ? take((size_t)kMaxGrid * kMaxCols * 8) : (size_t)-1;
    L.ep_partials = take((size_t)kMaxGrid * 4 * 8);
    L.ep_totals = take(4 * 8);
    L.ep_totals_global = take(4 * 8);
    L.error_flag = take(4);
    L.cursor_max = take(4);
    L.reset_source = take(sizeof(ResetSource));
    // MT19937: one std::mt19937 state array per env, 2,496 B each — 156 MiB at 65,536 envs, 39 GiB at 16.7 M (of 288 GB of HBM)
    L.mt_state = (c.env_kind == KIND_CARTPOLE && c.reset_mode == GYMHIP_RESET_MT19937) ? take(n * (size_t)kMtWords * 4) : (size_t)-1;
    L.total = off;
    return L;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// Static scratch of the allocation-free entry point (gymhip_update_norm_stats_noalloc): the per-block partial sums and the
// reduced moments that gymhip_update_norm_stats keeps in the engine's workspace.  One copy per GPU, held until the process
// ends: kNoAllocGrid x kMaxCols partial doubles + kMaxCols moment doubles = 131,200 B.  The array's NAME is only a host-side
// stand-in; the device address is resolved once per device with hipGetSymbolAddress (resolve_scratch) and every use in
// host code goes through that pointer.
// ------------------------------------------------------------------------------------------------------------
constexpr int kNoAllocGrid = 1024;                               // partial rows the scratch holds (>= the batch-moments launch's blocks)
static_assert(kNoAllocGrid >= kDefaultGrid, "the batch-moments launch writes one partial row per block");
constexpr int64_t kNoAllocMaxRows = (int64_t)1 << 26;             // rows per call (obs rows or reward rows), stated limit
constexpr size_t kScratchDoubles = (size_t)kNoAllocGrid * kMaxCols + kMaxCols;
static __device__ __align__(256) double scratch[kScratchDoubles];

struct gymhip_engine {
    gymhip_config cfg;
    int device;
    int F;
    bool owns_ws;
    char* ws;
    size_t ws_bytes;
    Layout L;
    int grid;
    bool norm;
    float* tape;
    int64_t tape_episodes;
    int tape_source;         // 0 = none / uploaded / device-generated, 1 = drawn on the host with std::mt19937 (can be extended)
    bool tape_seeded;        // host-drawn tape: were the envs seeded (seed + index) or default-constructed
    uint64_t tape_seed;
    int lanes;               // threads per workgroup that own envs in the step / reset / rollout kernels (grid_for)
    bool did_reset;
    bool pending_finish;     // a *_begin with the normaliser on awaits its *_finish
    bool pending_is_reset;
    bool stream_nt;          // step kernel streams with non-temporal accesses (footprint beyond the Infinity Cache)
    int keep_small;          // residency plan of an NT engine (plan_residency): the 4-B lanes that stay in the Infinity Cache, one of kPlans
    double* scratch_dev;     // this device's copy of the static `scratch` array (resolve_scratch)
    int exchange_form;       // GYMHIP_EXCHANGE_MOMENTS (reduce launch, 2F+6 doubles) | GYMHIP_EXCHANGE_ROWS (the partial rows themselves)
    bool mt;                 // CartPole with per-env std::mt19937 states on the device (GYMHIP_RESET_MT19937)

    template <typename T> T* at(size_t off) const { return off == (size_t)-1 ? nullptr : (T*)(ws + off); }
    float* stats_slot_ptr(int s) const { return at<float>(L.stats) + 16 * s; }
};

namespace {

// The residency plan of an engine that streams non-temporally (its per-step footprint does not fit the 256 MiB Infinity Cache).
// Non-temporal accesses pass that cache by; what is accessed with the default policy stays in it from one step to the next.
// So the 4-B lanes (counter, reward, done, returns) are kept there when they fit kMallBudget and only the wide rows and the
// actions stream from HBM.  Measured at 16.7 M CartPole envs, lockstep (profiles/r02_sweeps.txt): counter alone +9 % for 64 MiB,
// reward + done +10 % for 128 MiB, all three 0.685 -> 0.82 of 8 TB/s; kept bytes beyond ~256 MiB lose everything again (a kept
// prefix of the wide rows on top of 192 MiB of lanes: +2 % at 256 MiB in total, -3 % at 272 MiB).  The step kernel is compiled
// for the plans in kPlans; the first one that fits is taken.  GYMHIP_MALL_MB / GYMHIP_KEEP_SMALL override for sweeps.
void plan_residency(gymhip_engine* h) {
    h->keep_small = 0;
    if (h->stream_nt) {
        size_t budget = kMallBudget;
        if (const char* e = getenv("GYMHIP_MALL_MB")) budget = (size_t)std::max(0, atoi(e)) << 20;
        const bool returns = h->L.returns != (size_t)-1;
        const bool tape = h->cfg.env_kind == KIND_CARTPOLE && (h->cfg.reset_mode == GYMHIP_RESET_TAPE || h->cfg.reset_mode == GYMHIP_RESET_MT19937);   // a per-reset gather either way
        for (int plan : kPlans) {
            if (tape && plan == kKeepAll) continue;   // measured: with the reset-tape gather in the mix the done lane is better streamed (0.707 vs 0.677 lockstep, 0.650 vs 0.635 float)
            int lanes = 0;
            for (int bit : {kKeepCounter, kKeepReward, kKeepDone, kKeepReturns}) lanes += (plan & bit) && (bit != kKeepReturns || returns);
            if ((size_t)h->cfg.n_envs * 4 * lanes <= budget) { h->keep_small = plan; break; }
        }
    }
    if (const char* e = getenv("GYMHIP_KEEP_SMALL")) {
        const int v = atoi(e);
        for (int plan : kPlans) if (plan == v) h->keep_small = v;
    }
}

// Partial rows of the ROWS exchange payload: the engine's grid rounded up to a size every rank with (nearly) the same env count
// agrees on — 256 (one round trip of the fold) or kDefaultGrid.
int exchange_rows(const gymhip_engine* h) { return h->grid <= 256 ? 256 : kDefaultGrid; }

StepParams make_params(gymhip_engine* h) {
    StepParams p{};
    p.n = h->cfg.n_envs;
    p.env_offset = h->cfg.env_offset;
    p.src = h->at<ResetSource>(h->L.reset_source);
    p.mt_state = h->at<uint32_t>(h->L.mt_state);
    p.wrappers = h->cfg.wrappers;
    p.max_episode_steps = h->cfg.max_episode_steps;
    p.reset_mode = h->cfg.reset_mode;
    p.gamma = h->cfg.gamma;
    p.lanes = h->lanes;
    p.pad_rows = (h->norm && h->exchange_form == GYMHIP_EXCHANGE_ROWS) ? exchange_rows(h) : 0;
    p.error_flag = h->at<int32_t>(h->L.error_flag);
    p.partials = h->at<double>(h->L.partials);
    p.ep_partials = h->at<double>(h->L.ep_partials);
    p.stats = h->norm ? h->stats_slot_ptr(0) : nullptr;
    p.tick_w = h->at<int32_t>(h->L.ticks);
    p.tick_r = h->norm ? h->at<int32_t>(h->L.ticks) + 1 : nullptr;
    StepBuffers& b = p.buf;
    b.state = h->at<void>(h->L.state);
    b.obs = h->at<float>(h->L.obs);
    b.reward = h->at<float>(h->L.reward);
    b.done = h->at<float>(h->L.done);
    b.counter = h->at<uint32_t>(h->L.counter);
    b.returns = h->at<float>(h->L.returns);
    return p;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(GYMHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int check_handle(gymhip_engine* h) {
    if (!h) return fail(GYMHIP_EINVAL, "null engine handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(GYMHIP_EHIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(e));
    return GYMHIP_OK;
}

// Publish the engine's reset source into its device block (ResetSource): kernels — also ones recorded into a HIP graph earlier —
// read it from there.  Blocking; callers that replace a tape synchronise the device first.
int push_reset_source(gymhip_engine* h) {
    ResetSource rs{};
    rs.tape = h->tape; rs.tape_episodes = h->tape_episodes; rs.seed = h->cfg.seed;
    HIP_TRY(hipMemcpy(h->at<ResetSource>(h->L.reset_source), &rs, sizeof rs, hipMemcpyHostToDevice));
    return GYMHIP_OK;
}

// (Re)seed every env's device-resident std::mt19937 (MT19937 engines): env i <- seed + env_offset + i, or the default-constructed
// engine when `seeded` is false.  Enqueued on the null stream; the caller synchronises.
int seed_generators(gymhip_engine* h, bool seeded, uint64_t seed) {
    const int64_t n = h->cfg.n_envs;
    const int g = (int)std::min<int64_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(mt_seed_kernel, dim3(g), dim3(256), 0, (hipStream_t) nullptr, h->at<uint32_t>(h->L.mt_state), n, seeded ? 1 : 0, seed, h->cfg.env_offset);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

// Device address of the static `scratch` array on `device` (which must be current).  Each GPU holds its own copy of the
// array at its own address, so the lookup is done once per device and kept; the array's name itself is never passed to a
// kernel or a copy.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double* g_scratch_dev[kMaxDevices] = {};
int resolve_scratch(int device, double** out) {
    if (device < 0 || device >= kMaxDevices) return fail(GYMHIP_EINVAL, "device %d beyond the %d this library keeps scratch pointers for", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_dev[device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(GYMHIP_EHIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
        g_scratch_dev[device] = (double*)ptr;
    }
    *out = g_scratch_dev[device];
    return GYMHIP_OK;
}

template <int KIND, bool LOCKSTEP, bool NORM, bool NT, int KEEP, bool MT>
void launch_step_act(gymhip_engine* h, const StepParams& p, int action_dtype, hipStream_t s) {
    switch (action_dtype) {
        case GYMHIP_ACT_I32: hipLaunchKernelGGL((step_kernel<KIND, LOCKSTEP, int32_t, NORM, NT, KEEP, MT>), dim3(h->grid), dim3(kBlock), 0, s, p); break;
        case GYMHIP_ACT_I64: hipLaunchKernelGGL((step_kernel<KIND, LOCKSTEP, int64_t, NORM, NT, KEEP, MT>), dim3(h->grid), dim3(kBlock), 0, s, p); break;
        default: hipLaunchKernelGGL((step_kernel<KIND, LOCKSTEP, uint8_t, NORM, NT, KEEP, MT>), dim3(h->grid), dim3(kBlock), 0, s, p); break;
    }
}
template <int KIND, bool LOCKSTEP, bool NORM, bool MT>
void launch_step_nt(gymhip_engine* h, const StepParams& p, int action_dtype, hipStream_t s) {
    if (!h->stream_nt) { launch_step_act<KIND, LOCKSTEP, NORM, false, kKeepAll, MT>(h, p, action_dtype, s); return; }
    switch (h->keep_small) {   // one of kPlans (plan_residency)
        case kKeepAll: launch_step_act<KIND, LOCKSTEP, NORM, true, kKeepAll, MT>(h, p, action_dtype, s); break;
        case kKeepNotDone: launch_step_act<KIND, LOCKSTEP, NORM, true, kKeepNotDone, MT>(h, p, action_dtype, s); break;
        case kKeepCounter: launch_step_act<KIND, LOCKSTEP, NORM, true, kKeepCounter, MT>(h, p, action_dtype, s); break;
        default: launch_step_act<KIND, LOCKSTEP, NORM, true, 0, MT>(h, p, action_dtype, s); break;
    }
}
template <int KIND, bool LOCKSTEP, bool MT>
void launch_step_norm(gymhip_engine* h, const StepParams& p, int action_dtype, hipStream_t s) {
    if (h->norm) launch_step_nt<KIND, LOCKSTEP, true, MT>(h, p, action_dtype, s);
    else launch_step_nt<KIND, LOCKSTEP, false, MT>(h, p, action_dtype, s);
}
template <int KIND>
void launch_step_kind(gymhip_engine* h, const StepParams& p, int action_dtype, hipStream_t s) {
    const bool lock = h->cfg.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    if constexpr (KIND == KIND_CARTPOLE) {
        if (h->mt) { if (lock) launch_step_norm<KIND, true, true>(h, p, action_dtype, s); else launch_step_norm<KIND, false, true>(h, p, action_dtype, s); return; }
    }
    if (lock) launch_step_norm<KIND, true, false>(h, p, action_dtype, s);
    else launch_step_norm<KIND, false, false>(h, p, action_dtype, s);
}

template <int KIND, bool LOCKSTEP, bool MT>
void launch_reset_norm(gymhip_engine* h, const StepParams& p, hipStream_t s) {
    if (h->norm) hipLaunchKernelGGL((reset_kernel<KIND, LOCKSTEP, true, MT>), dim3(h->grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((reset_kernel<KIND, LOCKSTEP, false, MT>), dim3(h->grid), dim3(kBlock), 0, s, p);
}
template <int KIND>
void launch_reset_kind(gymhip_engine* h, const StepParams& p, hipStream_t s) {
    const bool lock = h->cfg.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    if constexpr (KIND == KIND_CARTPOLE) {
        if (h->mt) { if (lock) launch_reset_norm<KIND, true, true>(h, p, s); else launch_reset_norm<KIND, false, true>(h, p, s); return; }
    }
    if (lock) launch_reset_norm<KIND, true, false>(h, p, s);
    else launch_reset_norm<KIND, false, false>(h, p, s);
}

void launch_reduce_partials(gymhip_engine* h, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kFoldThreads), 0, s, h->at<double>(h->L.partials), h->grid, 2 * h->F + 5,
                       (double)h->cfg.n_envs, h->at<double>(h->L.moments));
}

int launch_normalize(gymhip_engine* h, bool is_reset, hipStream_t s, bool fold, bool rows_form = false) {
    const uint32_t w = h->cfg.wrappers;
    const bool training = w & GYMHIP_WRAP_NORM_TRAINING;
    NormParams q{};
    q.n = h->cfg.n_envs;
    q.stats = h->stats_slot_ptr(0);
    q.tick_w = h->at<int32_t>(h->L.ticks); q.tick_r = h->at<int32_t>(h->L.ticks) + 1;
    q.moments = h->at<double>(h->L.moments);
    if (fold) { q.partials = h->at<double>(h->L.partials); q.rows = h->grid; q.n_local = (double)h->cfg.n_envs; q.moments_out = h->at<double>(h->L.moments); }
    if (rows_form) { q.rows = exchange_rows(h); q.n_from_rows = true; }   // the rows were summed over ranks by the caller's all-reduce
    q.ep_totals_global = is_reset ? nullptr : h->at<double>(h->L.ep_totals_global);
    q.obs = h->at<float>(h->L.obs);
    q.obs_norm = h->at<float>(h->L.obs_norm);
    q.reward = h->at<float>(h->L.reward);
    q.clip_obs = h->cfg.clip_obs; q.clip_reward = h->cfg.clip_reward; q.epsilon = h->cfg.epsilon;
    q.norm_obs = (w & GYMHIP_WRAP_NORM_OBS) != 0;
    q.norm_reward = !is_reset && (w & GYMHIP_WRAP_NORM_REWARD);
    q.update_obs = training && (w & GYMHIP_WRAP_NORM_OBS);                       // vec_normalize.h:149-151,162-164
    q.update_ret = !is_reset && training && (w & GYMHIP_WRAP_NORM_REWARD);       // vec_normalize.h:169-170
    const bool keep_reward = h->keep_small & kKeepReward;
    if (h->F == 4) {
        if (!h->stream_nt) hipLaunchKernelGGL((normalize_kernel<4, false>), dim3(h->grid), dim3(kBlock), 0, s, q);
        else if (keep_reward) hipLaunchKernelGGL((normalize_kernel<4, true, true>), dim3(h->grid), dim3(kBlock), 0, s, q);
        else hipLaunchKernelGGL((normalize_kernel<4, true>), dim3(h->grid), dim3(kBlock), 0, s, q);
    } else {
        if (!h->stream_nt) hipLaunchKernelGGL((normalize_kernel<2, false>), dim3(h->grid), dim3(kBlock), 0, s, q);
        else if (keep_reward) hipLaunchKernelGGL((normalize_kernel<2, true, true>), dim3(h->grid), dim3(kBlock), 0, s, q);
        else hipLaunchKernelGGL((normalize_kernel<2, true>), dim3(h->grid), dim3(kBlock), 0, s, q);
    }
    return GYMHIP_OK;   // the kernel always publishes the (possibly unchanged) statistics into the other slot and bumps the device-side tick
}

}  // namespace

extern "C" {

const char* gymhip_last_error(void) { return g_last_error.c_str(); }
int gymhip_version(void) { return 1; }

size_t gymhip_workspace_bytes(const gymhip_config* cfg) {
    if (!valid_config(cfg)) return 0;
    return make_layout(*cfg).total;
}

int gymhip_create(const gymhip_config* cfg, int device, void* workspace, size_t workspace_bytes, gymhip_engine** out) {
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    *out = nullptr;
    if (!valid_config(cfg)) return fail(GYMHIP_EINVAL, "invalid gymhip_config (struct_size/env_kind/precision/reset_mode/n_envs/max_episode_steps)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(GYMHIP_EINVAL, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    gymhip_engine* h = new (std::nothrow) gymhip_engine();
    if (!h) return fail(GYMHIP_EINVAL, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    h->F = obs_dim_of(cfg->env_kind);
    h->L = make_layout(*cfg);
    h->norm = cfg->wrappers & (GYMHIP_WRAP_NORM_OBS | GYMHIP_WRAP_NORM_REWARD);
    h->tape = nullptr; h->tape_episodes = 0; h->tape_source = 0; h->tape_seeded = false; h->tape_seed = 0;
    h->did_reset = false; h->pending_finish = false; h->pending_is_reset = false;
    h->mt = cfg->env_kind == KIND_CARTPOLE && cfg->reset_mode == GYMHIP_RESET_MT19937;
    h->exchange_form = GYMHIP_EXCHANGE_MOMENTS;
    // Plain (temporal) accesses win while everything a step touches stays in the 256 MiB Infinity Cache from one step to the next;
    // beyond that they thrash it and the engine streams non-temporally under a residency plan (plan_residency).  "Everything" is
    // the workspace plus the callers' action rows — the bench rotates four int32 rows, a trainer's policy has its own traffic — so
    // the footprint counts 16 B of actions per env.  Measured crossovers (tools/sweep_nt_threshold.sh): CartPole lockstep between
    // 4.2 M (plain 0.79 | streaming 0.71) and 5.2 M envs (0.64 | 0.79); MountainCar lockstep between 5.8 M (0.79 | 0.75) and 6.3 M (0.67 | 0.74).
    size_t nt_footprint = kNtFootprint;
    if (const char* e = getenv("GYMHIP_NT_MB")) nt_footprint = (size_t)std::max(0, atoi(e)) << 20;   // sweeps only
    const size_t step_footprint = h->L.total - (h->mt ? (size_t)cfg->n_envs * kMtWords * 4 : 0);   // the generator states are touched by resets only
    h->stream_nt = kNtDefault && step_footprint + (size_t)cfg->n_envs * 16 > nt_footprint;
    h->grid = grid_for(cfg->n_envs, h->norm, cfg->env_kind == KIND_CARTPOLE && cfg->precision == GYMHIP_PRECISION_LOCKSTEP_F64, h->stream_nt && cfg->reset_mode == GYMHIP_RESET_COUNTER_RNG /* engines that gather a reset row (tape, per-env generators) are best at 2048: MT19937 0.2505 ms against 0.2626 at 16384 */, &h->lanes);
    plan_residency(h);
    if (int rc = resolve_scratch(device, &h->scratch_dev)) { delete h; return rc; }
    if (workspace) {
        if (workspace_bytes < h->L.total || ((uintptr_t)workspace & 255)) { delete h; return fail(GYMHIP_EINVAL, "workspace too small (%zu < %zu) or not 256-B aligned", workspace_bytes, h->L.total); }
        h->ws = (char*)workspace; h->owns_ws = false;
    } else {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, h->L.total);
        if (e != hipSuccess) { delete h; return fail(GYMHIP_EHIP, "hipMalloc(%zu) failed: %s", h->L.total, hipGetErrorString(e)); }
        h->ws = (char*)p; h->owns_ws = true;
    }
    h->ws_bytes = h->L.total;
    hipError_t e = hipMemset(h->ws, 0, h->L.total);
    if (e == hipSuccess && h->norm) {
        // RunningMeanStd ctor (running_mean_std.h:16-20): mean 0, var 1, count 1e-4 — obs and return statistics
        float init[16] = {0};
        for (int f = 0; f < h->F; f++) { init[f] = 0.f; init[h->F + f] = 1.f; }
        init[2 * h->F] = 1e-4f; init[2 * h->F + 1] = 0.f; init[2 * h->F + 2] = 1.f; init[2 * h->F + 3] = 1e-4f;
        for (int s = 0; s < 3 && e == hipSuccess; s++) e = hipMemcpy(h->stats_slot_ptr(s), init, sizeof init, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { if (h->owns_ws) (void)hipFree(h->ws); delete h; return fail(GYMHIP_EHIP, "workspace init failed: %s", hipGetErrorString(e)); }
    int rc = push_reset_source(h);
    // MT19937: until VecEnv::seed is called every env holds a default-constructed std::mt19937 (env.h:83), as in the reference
    if (rc == GYMHIP_OK && h->mt) { rc = seed_generators(h, false, 0); if (rc == GYMHIP_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(GYMHIP_EHIP, "seeding the per-env generators failed"); }
    if (rc != GYMHIP_OK) { if (h->owns_ws) (void)hipFree(h->ws); delete h; return rc; }
    *out = h;
    return GYMHIP_OK;
}

int gymhip_destroy(gymhip_engine* h) {
    if (!h) return GYMHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->tape) (void)hipFree(h->tape);
    if (h->owns_ws) (void)hipFree(h->ws);
    delete h;
    return GYMHIP_OK;
}

int gymhip_get_buffers(gymhip_engine* h, gymhip_buffers* out) {
    if (!h || !out) return fail(GYMHIP_EINVAL, "null argument");
    std::memset(out, 0, sizeof *out);
    out->obs_dim = h->F;
    out->moments_len = 1 + 2 * h->F + 5;
    out->state = h->at<void>(h->L.state);
    out->obs = h->at<float>(h->L.obs);
    out->obs_norm = h->at<float>(h->L.obs_norm);
    out->reward = h->at<float>(h->L.reward);
    out->done = h->at<float>(h->L.done);
    out->counter = h->at<uint32_t>(h->L.counter);
    out->returns = h->at<float>(h->L.returns);
    out->norm_stats = h->norm ? h->stats_slot_ptr(2) : nullptr;
    out->moments = h->at<double>(h->L.moments);
    if (h->norm && h->exchange_form == GYMHIP_EXCHANGE_ROWS) { out->moments = h->at<double>(h->L.partials); out->moments_len = exchange_rows(h) * kMaxCols; }
    out->episode_totals = h->at<double>(h->L.ep_totals);
    out->mt_state = h->at<uint32_t>(h->L.mt_state);
    return GYMHIP_OK;
}

static int rewind_tape_cursors(gymhip_engine* h, hipStream_t s);
int gymhip_seed(gymhip_engine* h, uint64_t seed) {
    if (int rc = check_handle(h)) return rc;
    h->cfg.seed = seed;
    HIP_TRY(hipDeviceSynchronize());                 // no stream argument: nothing of this engine may still be running on the old seed
    if (h->mt) {                                     // Env::seed re-seeds the env's generator (env.h:60-63): every stream restarts
        if (int rc = seed_generators(h, true, seed)) return rc;
        if (int rc = rewind_tape_cursors(h, nullptr)) return rc;
        HIP_TRY(hipDeviceSynchronize());
    }
    return push_reset_source(h);                     // COUNTER_RNG kernels (recorded ones too) read the seed from the device block
}

// every freshly loaded tape is read from row 0 again (see rewind_tape_cursor_kernel)
static int rewind_tape_cursors(gymhip_engine* h, hipStream_t s) {
    hipLaunchKernelGGL(rewind_tape_cursor_kernel, dim3(h->grid), dim3(kBlock), 0, s, h->at<uint32_t>(h->L.counter), h->cfg.n_envs);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

// A device tape of exactly `episodes` rows.  A tape of another size is replaced: the device is synchronised, the new rows are
// allocated, the engine's device-side reset source is pointed at them, and only then is the old tape freed — so a step recorded into
// a HIP graph before the change gathers from the NEW tape when it is replayed (it reads pointer and row count from the device block).
static int ensure_tape(gymhip_engine* h, int64_t episodes) {
    if (h->cfg.env_kind != KIND_CARTPOLE) return fail(GYMHIP_EINVAL, "only CartPole draws its reset state (MountainCar resets are constant)");
    if (h->mt) return fail(GYMHIP_EINVAL, "an MT19937 engine draws its reset states from per-env generators on the device: seed it with gymhip_seed");
    if (episodes <= 0) return fail(GYMHIP_EINVAL, "episodes must be positive");
    if (h->tape && h->tape_episodes == episodes) return GYMHIP_OK;
    HIP_TRY(hipDeviceSynchronize());
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, (size_t)episodes * h->cfg.n_envs * 16));
    float* old = h->tape;
    h->tape = (float*)p; h->tape_episodes = episodes;
    if (int rc = push_reset_source(h)) return rc;
    if (old) HIP_TRY(hipFree(old));
    return GYMHIP_OK;
}

int gymhip_upload_reset_tape(gymhip_engine* h, const float* host_tape, int64_t episodes) {
    if (int rc = check_handle(h)) return rc;
    if (!host_tape) return fail(GYMHIP_EINVAL, "host_tape is null");
    if (int rc = ensure_tape(h, episodes)) return rc;
    HIP_TRY(hipDeviceSynchronize());   // no stream argument here: nothing of this engine may still be reading the old rows
    HIP_TRY(hipMemcpy(h->tape, host_tape, (size_t)episodes * h->cfg.n_envs * 16, hipMemcpyHostToDevice));
    h->tape_source = 0;
    if (int rc = rewind_tape_cursors(h, nullptr)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return GYMHIP_OK;
}

static int draw_reset_tape(gymhip_engine* h, bool seeded, uint64_t seed, int64_t episodes, bool rewind);
int gymhip_seed_reset_tape(gymhip_engine* h, uint64_t seed, int64_t episodes) { return draw_reset_tape(h, true, seed, episodes, true); }
int gymhip_unseeded_reset_tape(gymhip_engine* h, int64_t episodes) { return draw_reset_tape(h, false, 0, episodes, true); }
// More rows of the SAME streams: the tape is redrawn from the remembered seeds with `episodes` rows per env (a prefix of an
// env's stream never changes) and the cursors stay where they are, so a running engine simply finds more rows.
int gymhip_extend_reset_tape(gymhip_engine* h, int64_t episodes) {
    if (int rc = check_handle(h)) return rc;
    if (!h->tape || h->tape_source != 1) return fail(GYMHIP_ESTATE, "only a tape drawn by gymhip_seed_reset_tape / gymhip_unseeded_reset_tape can be extended");
    if (episodes <= h->tape_episodes) return fail(GYMHIP_EINVAL, "the tape already holds %lld rows per env", (long long)h->tape_episodes);
    return draw_reset_tape(h, h->tape_seeded, h->tape_seed, episodes, false);
}
static int draw_reset_tape(gymhip_engine* h, bool seeded, uint64_t seed, int64_t episodes, bool rewind) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = ensure_tape(h, episodes)) return rc;
    // The reference's own generator, called the way it calls it: env i owns std::mt19937 seeded `seed + i`
    // (sync_env.h:74-83 -> env.h:60-63 / cartpole.h:24-26) and every reset draws four
    // std::uniform_real_distribution<float>(-0.05f, 0.05f) values in order (cartpole.cpp:21-26).
    const int64_t n = h->cfg.n_envs;
    std::vector<float> tape;
    try { tape.resize((size_t)episodes * n * 4); } catch (...) { return fail(GYMHIP_EINVAL, "out of host memory for the reset tape"); }
    unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), 32, n / 1024 + 1}));
    std::vector<std::thread> pool;
    float* out = tape.data();
    const int64_t off = h->cfg.env_offset;
    for (int t = 0; t < n_thr; t++) {
        pool.emplace_back([=]() {
            for (int64_t i = t; i < n; i += n_thr) {
                std::mt19937 dev;   // default-constructed (5489) when the caller never seeded: all envs share one stream (env.h:83)
                if (seeded) dev.seed(seed + (uint64_t)(off + i));
                for (int64_t k = 0; k < episodes; k++) {
                    std::uniform_real_distribution<float> dist(-0.05f, 0.05f);
                    float* row = out + (k * n + i) * 4;
                    row[0] = dist(dev); row[1] = dist(dev); row[2] = dist(dev); row[3] = dist(dev);
                }
            }
        });
    }
    for (auto& th : pool) th.join();
    if (seeded) h->cfg.seed = seed;
    h->tape_source = 1; h->tape_seeded = seeded; h->tape_seed = seed;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = push_reset_source(h)) return rc;
    HIP_TRY(hipMemcpy(h->tape, tape.data(), tape.size() * sizeof(float), hipMemcpyHostToDevice));
    if (rewind) { if (int rc = rewind_tape_cursors(h, nullptr)) return rc; }
    HIP_TRY(hipDeviceSynchronize());
    return GYMHIP_OK;
}

int gymhip_tape_cursor_max(gymhip_engine* h, int32_t* host_out, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!host_out) return fail(GYMHIP_EINVAL, "host_out is null");
    hipStream_t s = (hipStream_t)stream;
    int32_t* slot = h->at<int32_t>(h->L.cursor_max);
    HIP_TRY(hipMemsetAsync(slot, 0, 4, s));
    if (h->cfg.env_kind == KIND_CARTPOLE && (h->cfg.reset_mode == GYMHIP_RESET_TAPE || h->mt))
        hipLaunchKernelGGL(tape_cursor_max_kernel, dim3(h->grid), dim3(kBlock), 0, s, h->at<uint32_t>(h->L.counter), h->cfg.n_envs, slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_out, slot, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GYMHIP_OK;
}

int gymhip_generate_reset_tape(gymhip_engine* h, int64_t episodes, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = ensure_tape(h, episodes)) return rc;
    hipLaunchKernelGGL(fill_tape_kernel, dim3(kMaxGrid), dim3(kBlock), 0, (hipStream_t)stream, (float4*)h->tape, h->cfg.n_envs, episodes, h->cfg.seed, h->cfg.env_offset);
    HIP_TRY(hipGetLastError());
    h->tape_source = 0;
    return rewind_tape_cursors(h, (hipStream_t)stream);
}

int gymhip_download_reset_tape(gymhip_engine* h, float* host_tape, int64_t episodes) {
    if (int rc = check_handle(h)) return rc;
    if (!host_tape || !h->tape || episodes > h->tape_episodes || episodes <= 0) return fail(GYMHIP_EINVAL, "no device tape of that size");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_tape, h->tape, (size_t)episodes * h->cfg.n_envs * 16, hipMemcpyDeviceToHost));
    return GYMHIP_OK;
}

// A step (or reset) issues no allocation and no synchronisation, and every per-step value its kernels need lives on the device —
// the reset-tape cursor and the counter-RNG episode index in each env's counter word, the normaliser's current slot in the
// tick pair — so a recorded step replays correctly from a HIP graph (tests/test_gpu_graph_capture.py).
static int reset_begin_impl(gymhip_engine* h, void* stream, bool split) {
    if (int rc = check_handle(h)) return rc;
    if (h->pending_finish) return fail(GYMHIP_ESTATE, "previous *_begin has no matching *_finish");
    if (h->cfg.env_kind == KIND_CARTPOLE && h->cfg.reset_mode == GYMHIP_RESET_TAPE && !h->tape)
        return fail(GYMHIP_ETAPE, "reset_mode TAPE but no reset tape uploaded (gymhip_upload_reset_tape)");
    hipStream_t s = (hipStream_t)stream;
    StepParams p = make_params(h);
    if (h->cfg.env_kind == KIND_CARTPOLE) launch_reset_kind<KIND_CARTPOLE>(h, p, s); else launch_reset_kind<KIND_MOUNTAINCAR>(h, p, s);
    if (h->norm) {
        if (!split) launch_normalize(h, true, s, true);                                   // unsplit: the normalise kernel folds the partial rows itself
        else if (h->exchange_form == GYMHIP_EXCHANGE_MOMENTS) launch_reduce_partials(h, s);   // the totals are the all-reduce payload (ROWS: the rows are)
    }
    HIP_TRY(hipGetLastError());
    h->did_reset = true;
    h->pending_finish = h->norm && split; h->pending_is_reset = true;
    return GYMHIP_OK;
}
int gymhip_reset_begin(gymhip_engine* h, void* stream) { return reset_begin_impl(h, stream, true); }

static int step_begin_impl(gymhip_engine* h, const void* actions, int action_dtype, void* stream, bool split) {
    if (int rc = check_handle(h)) return rc;
    if (!actions) return fail(GYMHIP_EINVAL, "actions is null");
    if (action_dtype < GYMHIP_ACT_I32 || action_dtype > GYMHIP_ACT_U8) return fail(GYMHIP_EINVAL, "unknown action_dtype %d", action_dtype);
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "step before reset");
    if (h->pending_finish) return fail(GYMHIP_ESTATE, "previous *_begin has no matching *_finish");
    hipStream_t s = (hipStream_t)stream;
    StepParams p = make_params(h);
    p.actions = actions;
    if (h->cfg.env_kind == KIND_CARTPOLE) launch_step_kind<KIND_CARTPOLE>(h, p, action_dtype, s); else launch_step_kind<KIND_MOUNTAINCAR>(h, p, action_dtype, s);
    if (h->norm) {
        if (!split) launch_normalize(h, false, s, true);
        else if (h->exchange_form == GYMHIP_EXCHANGE_MOMENTS) launch_reduce_partials(h, s);
    }
    HIP_TRY(hipGetLastError());
    h->pending_finish = h->norm && split; h->pending_is_reset = false;
    return GYMHIP_OK;
}
int gymhip_step_begin(gymhip_engine* h, const void* actions, int action_dtype, void* stream) { return step_begin_impl(h, actions, action_dtype, stream, true); }

static int finish_common(gymhip_engine* h, bool is_reset, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!h->norm) return GYMHIP_OK;
    if (!h->pending_finish || h->pending_is_reset != is_reset) return fail(GYMHIP_ESTATE, "*_finish without the matching *_begin");
    const bool rows_form = h->exchange_form == GYMHIP_EXCHANGE_ROWS;
    launch_normalize(h, is_reset, (hipStream_t)stream, rows_form, rows_form);
    HIP_TRY(hipGetLastError());
    h->pending_finish = false;
    return GYMHIP_OK;
}
int gymhip_step_finish(gymhip_engine* h, void* stream) { return finish_common(h, false, stream); }
int gymhip_reset_finish(gymhip_engine* h, void* stream) { return finish_common(h, true, stream); }

// The unsplit forms: two launches with the normaliser on (step/reset, then normalise with the partial-row fold in its
// prologue), one without; bit-identical to _begin + _finish.
int gymhip_reset(gymhip_engine* h, void* stream) { return reset_begin_impl(h, stream, false); }
int gymhip_step(gymhip_engine* h, const void* actions, int action_dtype, void* stream) { return step_begin_impl(h, actions, action_dtype, stream, false); }

int gymhip_rollout(gymhip_engine* h, const uint8_t* action_tape, int64_t k_steps, uint8_t* rec_done, float* rec_reward,
                   uint8_t* rec_truncated, float* rec_obs, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!action_tape || k_steps <= 0) return fail(GYMHIP_EINVAL, "action_tape null or k_steps <= 0");
    if (h->norm) return fail(GYMHIP_EINVAL, "rollout is not available with the normaliser (needs a grid-wide reduction per step)");
    if (h->cfg.wrappers & GYMHIP_WRAP_NO_AUTO_RESET) return fail(GYMHIP_EINVAL, "rollout always auto-resets: not available with GYMHIP_WRAP_NO_AUTO_RESET");
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "rollout before reset");
    hipStream_t s = (hipStream_t)stream;
    StepParams p = make_params(h);
    const bool lock = h->cfg.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    dim3 g(h->grid), bdim(kBlock);
    if (h->mt) {
        if (lock) hipLaunchKernelGGL((rollout_kernel<KIND_CARTPOLE, true, true>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
        else hipLaunchKernelGGL((rollout_kernel<KIND_CARTPOLE, false, true>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
    } else if (h->cfg.env_kind == KIND_CARTPOLE) {
        if (lock) hipLaunchKernelGGL((rollout_kernel<KIND_CARTPOLE, true>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
        else hipLaunchKernelGGL((rollout_kernel<KIND_CARTPOLE, false>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
    } else {
        if (lock) hipLaunchKernelGGL((rollout_kernel<KIND_MOUNTAINCAR, true>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
        else hipLaunchKernelGGL((rollout_kernel<KIND_MOUNTAINCAR, false>), g, bdim, 0, s, p, action_tape, k_steps, rec_done, rec_reward, rec_truncated, rec_obs);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_decode_counters(gymhip_engine* h, int32_t* elapsed_steps, int32_t* episode_length, uint8_t* truncated, float* episode_return,
                           int32_t* tape_cursor, void* stream) {
    if (int rc = check_handle(h)) return rc;
    const uint32_t* c = h->at<uint32_t>(h->L.counter);
    if (h->cfg.env_kind == KIND_CARTPOLE)
        hipLaunchKernelGGL(decode_counter_kernel<KIND_CARTPOLE>, dim3(h->grid), dim3(kBlock), 0, (hipStream_t)stream, c, h->cfg.n_envs, elapsed_steps, episode_length, truncated, episode_return, tape_cursor);
    else
        hipLaunchKernelGGL(decode_counter_kernel<KIND_MOUNTAINCAR>, dim3(h->grid), dim3(kBlock), 0, (hipStream_t)stream, c, h->cfg.n_envs, elapsed_steps, episode_length, truncated, episode_return, tape_cursor);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_read_state_f64(gymhip_engine* h, double* state_out, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!state_out) return fail(GYMHIP_EINVAL, "state_out is null");
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "state read before reset");
    const StepParams p = make_params(h);
    const bool lock = h->cfg.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    dim3 g(h->grid), bdim(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (h->F == 4) {
        if (lock) hipLaunchKernelGGL((read_state_kernel<4, true>), g, bdim, 0, s, p.buf, p.n, state_out);
        else hipLaunchKernelGGL((read_state_kernel<4, false>), g, bdim, 0, s, p.buf, p.n, state_out);
    } else {
        if (lock) hipLaunchKernelGGL((read_state_kernel<2, true>), g, bdim, 0, s, p.buf, p.n, state_out);
        else hipLaunchKernelGGL((read_state_kernel<2, false>), g, bdim, 0, s, p.buf, p.n, state_out);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_episode_stats(gymhip_engine* h, double* host_out, void* stream) {
    if (int rc = check_handle(h)) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(reduce_episode_kernel, dim3(1), dim3(kBlock), 0, s, h->at<double>(h->L.ep_partials), h->grid, (const double*)nullptr, h->at<double>(h->L.ep_totals));
    HIP_TRY(hipGetLastError());
    if (host_out) {
        HIP_TRY(hipMemcpyAsync(host_out, h->at<double>(h->L.ep_totals), 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return GYMHIP_OK;
}

int gymhip_poll_error(gymhip_engine* h, int32_t* host_out, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!host_out) return fail(GYMHIP_EINVAL, "host_out is null");
    HIP_TRY(hipMemcpyAsync(host_out, h->at<int32_t>(h->L.error_flag), 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GYMHIP_OK;
}

int gymhip_set_exchange_form(gymhip_engine* h, int form) {
    if (int rc = check_handle(h)) return rc;
    if (form != GYMHIP_EXCHANGE_MOMENTS && form != GYMHIP_EXCHANGE_ROWS) return fail(GYMHIP_EINVAL, "unknown exchange form %d", form);
    if (!h->norm) return fail(GYMHIP_EINVAL, "engine has no normaliser: nothing is exchanged");
    if (h->pending_finish) return fail(GYMHIP_ESTATE, "previous *_begin has no matching *_finish");
    h->exchange_form = form;
    return GYMHIP_OK;
}

int gymhip_launch_info(gymhip_engine* h, int32_t out[4]) {
    if (!h || !out) return fail(GYMHIP_EINVAL, "null argument");
    out[0] = h->grid; out[1] = h->lanes; out[2] = h->stream_nt ? 1 : 0; out[3] = h->stream_nt ? h->keep_small : 0;
    return GYMHIP_OK;
}

int gymhip_reset_index(gymhip_engine* h, int64_t index, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (index < 0 || index >= h->cfg.n_envs) return fail(GYMHIP_EINVAL, "env index %lld out of range [0, %lld)", (long long)index, (long long)h->cfg.n_envs);
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "reset(index) before reset()");
    if (h->pending_finish) return fail(GYMHIP_ESTATE, "previous *_begin has no matching *_finish");
    const StepParams p = make_params(h);
    float* on = (h->cfg.wrappers & GYMHIP_WRAP_NORM_OBS) ? h->at<float>(h->L.obs_norm) : nullptr;
    const bool lock = h->cfg.precision == GYMHIP_PRECISION_LOCKSTEP_F64;
    hipStream_t s = (hipStream_t)stream;
    if (h->mt) {
        if (lock) hipLaunchKernelGGL((reset_index_kernel<KIND_CARTPOLE, true, true>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
        else hipLaunchKernelGGL((reset_index_kernel<KIND_CARTPOLE, false, true>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
    } else if (h->cfg.env_kind == KIND_CARTPOLE) {
        if (lock) hipLaunchKernelGGL((reset_index_kernel<KIND_CARTPOLE, true>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
        else hipLaunchKernelGGL((reset_index_kernel<KIND_CARTPOLE, false>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
    } else {
        if (lock) hipLaunchKernelGGL((reset_index_kernel<KIND_MOUNTAINCAR, true>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
        else hipLaunchKernelGGL((reset_index_kernel<KIND_MOUNTAINCAR, false>), dim3(1), dim3(64), 0, s, p, index, on, h->cfg.clip_obs, h->cfg.epsilon);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

// Shared body of gymhip_update_norm_stats and its allocation-free variant: `partials` ([>= kDefaultGrid][kMaxCols]) and
// `moments` ([kMaxCols]) are the only device memory it touches beyond the caller's rows and the engine's statistics.
static int update_norm_stats_impl(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows, int64_t n_reward_rows,
                                  double* partials, double* moments, hipStream_t s) {
    const int F = h->F;
    // one pass per statistic: batch moments -> moments buffer -> RunningMeanStd::updateFromMoment (no normalisation: n = 0 rows)
    auto pass = [&](const float* x, int64_t rows, int width, int stat_off, int col0, int col0_sq, bool is_obs) -> int {
        int64_t g64 = (rows + kBlock - 1) / kBlock;
        const int g = (int)(g64 < kDefaultGrid ? g64 : kDefaultGrid);
        // every launch rewrites all kMaxCols columns of partial rows [0, g) and the reduction reads exactly those rows:
        // nothing in `partials` has to start at zero, and rows a larger earlier call left behind are never read
        int32_t* ticks = h->at<int32_t>(h->L.ticks);
        hipLaunchKernelGGL(batch_moments_kernel, dim3(g), dim3(kBlock), 0, s, x, rows, width, (const float*)h->stats_slot_ptr(0), (const int32_t*)ticks, ticks + 1, stat_off,
                           col0, col0_sq, partials);
        NormParams q{};
        q.n = 0;
        q.stats = h->stats_slot_ptr(0);
        q.tick_w = ticks; q.tick_r = ticks + 1;
        q.partials = partials; q.rows = g; q.n_local = (double)rows; q.moments_out = moments;   // the one-block normalise launch folds the rows itself
        q.clip_obs = h->cfg.clip_obs; q.clip_reward = h->cfg.clip_reward; q.epsilon = h->cfg.epsilon;
        q.update_obs = is_obs; q.update_ret = !is_obs;
        if (F == 4) hipLaunchKernelGGL((normalize_kernel<4, false>), dim3(1), dim3(kBlock), 0, s, q);
        else hipLaunchKernelGGL((normalize_kernel<2, false>), dim3(1), dim3(kBlock), 0, s, q);
        HIP_TRY(hipGetLastError());
        return GYMHIP_OK;
    };
    if (obs_rows && (h->cfg.wrappers & GYMHIP_WRAP_NORM_OBS))
        if (int rc = pass(obs_rows, n_obs_rows, F, 0, 0, F, true)) return rc;                       // updateObservation (vec_normalize.h:199-201)
    if (reward_rows && (h->cfg.wrappers & GYMHIP_WRAP_NORM_REWARD))
        if (int rc = pass(reward_rows, n_reward_rows, 1, 2 * F + 1, 2 * F, 2 * F + 1, false)) return rc;   // m_RetRMS->update (vec_normalize.h:203-205)
    return GYMHIP_OK;
}

static int update_norm_stats_checks(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows, int64_t n_reward_rows) {
    if (int rc = check_handle(h)) return rc;
    if (!h->norm) return fail(GYMHIP_EINVAL, "engine has no normaliser");
    if (h->pending_finish) return fail(GYMHIP_ESTATE, "previous *_begin has no matching *_finish");
    if ((obs_rows && n_obs_rows <= 0) || (reward_rows && n_reward_rows <= 0)) return fail(GYMHIP_EINVAL, "row counts must be positive");
    return GYMHIP_OK;
}

int gymhip_update_norm_stats(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows, int64_t n_reward_rows, void* stream) {
    if (int rc = update_norm_stats_checks(h, obs_rows, n_obs_rows, reward_rows, n_reward_rows)) return rc;
    return update_norm_stats_impl(h, obs_rows, n_obs_rows, reward_rows, n_reward_rows, h->at<double>(h->L.partials), h->at<double>(h->L.moments), (hipStream_t)stream);
}

// The allocation-free variant: same kernels, same order, same results; partial sums and moments live in this GPU's copy of
// the static `scratch` array instead of the engine's workspace.  No hipMalloc / hipFree / allocating routine on this path.
int gymhip_update_norm_stats_noalloc(gymhip_engine* h, const float* obs_rows, int64_t n_obs_rows, const float* reward_rows, int64_t n_reward_rows, void* stream) {
    if (int rc = update_norm_stats_checks(h, obs_rows, n_obs_rows, reward_rows, n_reward_rows)) return rc;
    if ((obs_rows && n_obs_rows > kNoAllocMaxRows) || (reward_rows && n_reward_rows > kNoAllocMaxRows))
        return fail(GYMHIP_EINVAL, "gymhip_update_norm_stats_noalloc takes at most %lld rows per call (got %lld obs rows, %lld reward rows)",
                    (long long)kNoAllocMaxRows, (long long)(obs_rows ? n_obs_rows : 0), (long long)(reward_rows ? n_reward_rows : 0));
    double* sc = h->scratch_dev;   // resolved for this engine's device in gymhip_create
    if (!sc) return fail(GYMHIP_ESTATE, "static scratch was not resolved for device %d", h->device);
    return update_norm_stats_impl(h, obs_rows, n_obs_rows, reward_rows, n_reward_rows, sc, sc + (size_t)kNoAllocGrid * kMaxCols, (hipStream_t)stream);
}

int gymhip_diag_pole_sincos(const double* theta, int64_t n, double* sin_out, double* cos_out, void* stream) {
    if (!theta || !sin_out || !cos_out || n <= 0) return fail(GYMHIP_EINVAL, "null pointer or n <= 0");
    const int64_t g = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(diag_pole_sincos_kernel, dim3((unsigned)(g < kDefaultGrid ? g : kDefaultGrid)), dim3(kBlock), 0, (hipStream_t)stream, theta, n, sin_out, cos_out);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_get_norm_stats(gymhip_engine* h, float* host_out, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!h->norm || !host_out) return fail(GYMHIP_EINVAL, "engine has no normaliser or host_out is null");
    // slot 2 is the public copy: every publishing kernel writes it together with the new current slot
    HIP_TRY(hipMemcpyAsync(host_out, h->stats_slot_ptr(2), (2 * h->F + 4) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GYMHIP_OK;
}

int gymhip_set_norm_stats(gymhip_engine* h, const float* host_in, void* stream) {
    if (int rc = check_handle(h)) return rc;
    if (!h->norm || !host_in) return fail(GYMHIP_EINVAL, "engine has no normaliser or host_in is null");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int s = 0; s < 3; s++) HIP_TRY(hipMemcpy(h->stats_slot_ptr(s), host_in, (2 * h->F + 4) * 4, hipMemcpyHostToDevice));
    return GYMHIP_OK;
}

}  // extern "C"
