// atari.hip — gfx950 kernels + C ABI for the Atari preprocessing wrappers.
//
// Replaces, per VecEnv step over N host-emulated envs (ALE stays on the host):
//   MaxAndSkipEnv's cv::max of the last two frames      (wrappers/atari_wrappers.cpp:30, max_skip.h:12-29)
//   WarpFrameEnv::observation: RGB2GRAY + INTER_AREA 84x84 (wrappers/warp_frame.h:41-47)
//   StackedObservation / VecFrameStack                   (common/stacked_observation.h:106-130, vec_frame_stack.cpp:22-32)
//   VecNormAndPermute::observation                       (wrappers/vec_env/vec_atari.cpp:70-81)
//   EpisodicLife / Monitor / ClipReward bookkeeping lanes (wrappers/atari_wrappers.cpp:35-47, monitor.h:74-102, clip_reward.h:22-24)
//
// One workgroup per env frame.  Phase 1 streams both 210x160x3 frames with 12-B-per-lane loads (4 whole pixels, fully
// coalesced), takes the byte-wise max, converts to 8-bit gray and parks the gray image (33.6 KB) in LDS.  Phase 2 runs
// OpenCV's area resampler out of LDS (per-axis decimation tables, float accumulation in OpenCV's order, round-half-even
// saturate) and writes the 84x84 result straight into the env's ring-buffer plane: HBM sees each input byte once and
// 7,056 output bytes per env — 208,656 B per env-step, the algorithmic minimum.  Two kernels share that arithmetic:
// atari_preprocess_kernel (any geometry, RGB24 frames or palette-index screens, one workgroup per frame) and
// atari_preprocess_fixed_kernel (RGB 210x160 -> 84x84 with compile-time extents, persistent workgroups that walk the frames).
//
// -ffp-contract=off (Makefile) keeps the resampler's mul/add separate, as OpenCV's baseline x86-64 build does.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "gymhip.h"
#include "gymhip_internal.h"

using namespace gymhip;

namespace {

constexpr int kBlock = 256;     // elementwise helper kernels
constexpr size_t kRingKeepBytes = (size_t)240 << 20;   // a ring up to this size is read with the default cache policy (see materialize_nchw_kernel)
constexpr int kFrameBlock = 512;  // one workgroup per frame: 8 waves; 4 workgroups (37 KB LDS each) fill a CU's 32 wave slots
#ifndef GYMHIP_ATARI_WAVES
#define GYMHIP_ATARI_WAVES 6   // min waves/SIMD: 80 VGPRs -> 3 frames (24 waves) per CU; 4 = 4.67 TB/s, 6 = 5.14, 8 spills = 3.5 (profiles/r01_sweep.txt)
#endif
constexpr int kMaxT = 4;        // taps per destination index in the padded tables (zero-weight padding)

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(GYMHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Per-axis decimation table, padded to kMaxT taps per destination index.  A padding tap repeats the last source index
// with weight 0: `acc + S*0.f` is an exact no-op, so the padded sum equals OpenCV's variable-length sum bit for bit.
struct AxisTable {
    std::vector<int32_t> si;    // [dsize][kMaxT]
    std::vector<float> alpha;   // [dsize][kMaxT]
    int max_taps = 0;
};

// OpenCV computeResizeAreaTab (imgproc/resize.cpp), generic (non-integer-scale) INTER_AREA path; scale = 1/(dsize/ssize).
AxisTable make_axis_table(int ssize, int dsize) {
    AxisTable t;
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++) {
        std::vector<int32_t> si; std::vector<float> al;
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) { si.push_back(s1 - 1); al.push_back((float)((s1 - f1) / cell)); }
        for (int s = s1; s < s2; s++) { si.push_back(s); al.push_back((float)(1.0 / cell)); }
        if (f2 - s2 > 1e-3) { si.push_back(s2); al.push_back((float)(std::min(std::min(f2 - s2, 1.), cell) / cell)); }
        t.max_taps = std::max(t.max_taps, (int)si.size());
        for (int k = 0; k < kMaxT; k++) {
            const bool real = k < (int)si.size();
            t.si.push_back(real ? si[k] : (si.empty() ? 0 : si.back()));
            t.alpha.push_back(real ? al[k] : 0.f);
        }
    }
    return t;
}

struct AtariParams {
    const uint8_t* frames;     // RGB24: u8 [N][2][H][W][3]; INDEX8: u8 [N][2][H][W] palette indices
    const uint32_t* lut;       // INDEX8: 256 colours packed R | G<<8 | B<<16 (device)
    uint8_t* out;              // plane of env e starts at out + e * env_stride + slot_offset
    int64_t env_stride;
    int64_t slot_offset;
    const uint8_t* done;       // u8 [N] or null
    uint8_t* ring;             // ring base (for clear-on-done), or null
    const int32_t* xsi; const float* xal;   // [dw][kMaxT]
    const int32_t* ysi; const float* yal;   // [dh][kMaxT]
    int H, W, dh, dw, n_stack, slot, clear_on_done;
    int64_t n_frames;          // frame pairs in this launch (the persistent kernel's grid is smaller: its workgroups stride over them)
};

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
#ifndef GYMHIP_ATARI_SDWA_MAX
#define GYMHIP_ATARI_SDWA_MAX 1
#endif
#ifndef GYMHIP_ATARI_P2_UNROLL
#define GYMHIP_ATARI_P2_UNROLL 1   // the resampler's row loop is NOT unrolled: two rows in flight cost 20 VGPRs, i.e. the fourth workgroup per CU
#endif
#ifndef GYMHIP_ATARI_J_UNROLL
#define GYMHIP_ATARI_J_UNROLL NT
#endif
#ifndef GYMHIP_ATARI_UNR
#define GYMHIP_ATARI_UNR 2   // items per thread per batch of the RGB load pipeline (two batches in flight)
#endif
__device__ inline uint32_t max_u8x4(uint32_t a, uint32_t b) {
#if GYMHIP_ATARI_SDWA_MAX
    // byte-wise unsigned max, one sub-dword-addressed v_max_u16 per byte: the byte selects zero-extend both sources and the result
    // byte lands in place (the other bytes of the destination are preserved) — 4 instructions per dword with no masks, shifts or
    // re-packing, where the v_pk_max_u16 form below needs 10.  The compiler does not form these by itself.
    uint32_t d;
    asm("v_max_u16_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b));
    asm("v_max_u16_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1" : "+v"(d) : "v"(a), "v"(b));
    asm("v_max_u16_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2" : "+v"(d) : "v"(a), "v"(b));
    asm("v_max_u16_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3" : "+v"(d) : "v"(a), "v"(b));
    return d;
#else
    // even and odd bytes widened in place to two 16-bit lanes each -> v_pk_max_u16
    const uint32_t m = 0x00FF00FFu;
    const u16x2 e = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a & m), __builtin_bit_cast(u16x2, b & m));
    const u16x2 o = __builtin_elementwise_max(__builtin_bit_cast(u16x2, (a >> 8) & m), __builtin_bit_cast(u16x2, (b >> 8) & m));
    return __builtin_bit_cast(uint32_t, e) | (__builtin_bit_cast(uint32_t, o) << 8);
#endif
}

template <int GRAY15>
__device__ inline uint32_t gray_px(uint32_t r, uint32_t g, uint32_t b) {
    // cv::cvtColor(COLOR_RGB2GRAY) for 8-bit: fixed-point BT.601 weights (15-bit in OpenCV >= 4.x builds, 14-bit before)
    if (GRAY15) return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
    return (r * 4899u + g * 9617u + b * 1868u + (1u << 13)) >> 14;
}

// The same gray value from a pixel held as bytes R, G, B in one dword (byte 3: anything): two v_dot4_u32_u8 with the weights split
// into their low and high bytes (w = 256 * hi + lo, so sum(c * w) = 256 * sum(c * hi) + sum(c * lo) exactly; the fourth weight byte is 0)
// and one shift-add, instead of three byte extractions and three multiplies.
template <int GRAY15>
__device__ inline uint32_t gray_rgbx(uint32_t px) {
    constexpr uint32_t wr = GRAY15 ? 9798u : 4899u, wg = GRAY15 ? 19235u : 9617u, wb = GRAY15 ? 3735u : 1868u;
    constexpr int sh = GRAY15 ? 15 : 14;
    constexpr uint32_t lo = (wr & 255u) | ((wg & 255u) << 8) | ((wb & 255u) << 16), hi = (wr >> 8) | ((wg >> 8) << 8) | ((wb >> 8) << 16);
    return (__builtin_amdgcn_udot4(px, lo, 1u << (sh - 1), false) + (__builtin_amdgcn_udot4(px, hi, 0u, false) << 8)) >> sh;
}

struct u32x3 { uint32_t x, y, z; };   // 12-B load unit: 4 whole RGB pixels (the three dword loads merge into one dwordx3)
__device__ inline u32x3 load_px4(const uint32_t* base, int j) {
    u32x3 v;
    v.x = __builtin_nontemporal_load(base + 3 * j);
    v.y = __builtin_nontemporal_load(base + 3 * j + 1);
    v.z = __builtin_nontemporal_load(base + 3 * j + 2);
    return v;
}

// SRC: 0 = RGB24 frames (AtariEnv::getObs output, atari_env.cpp:11-17); 1 = INDEX8 screens, the emulator's native palette
// indices (ale->getScreen().getArray()), coloured in-kernel by the 256-entry table getScreenRGB2 applies on the host
// (atari/ale_c_wrapper.cpp:28-41): a third of the bytes over PCIe and out of HBM, identical pixels.
template <int GRAY15, int NT, int SRC>
__global__ __launch_bounds__(kFrameBlock, GYMHIP_ATARI_WAVES) void atari_preprocess_kernel(const AtariParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int HW = p.H * p.W;
    uint8_t* gray = lds;                                           // [H*W] u8
    int32_t* ysi = (int32_t*)(lds + ((HW + 15) & ~15));            // [dh][kMaxT]
    float* yal = (float*)(ysi + p.dh * kMaxT);                     // [dh][kMaxT]
    uint32_t* lut = (uint32_t*)(yal + p.dh * kMaxT);               // [256] (INDEX8)
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;

    for (int k = tid; k < p.dh * kMaxT; k += kFrameBlock) { ysi[k] = p.ysi[k]; yal[k] = p.yal[k]; }

    // ---- phase 1: max of the two frames -> gray -> LDS ---------------------------------------------------------
    uint32_t* gray32 = (uint32_t*)gray;
    if constexpr (SRC == 1) {
        // One 4-B table entry per colour (bytes R, G, B): measured against an 8-B entry {R | G << 16, B*wB + round} that
        // turns max + gray into pk_max + max + v_dot2_u32_u16, the smaller entry wins (0.132 vs 0.218 ms at 4096 envs,
        // random indices): the table lookups, not the arithmetic, bound this phase, and an 8-B entry doubles them.
        if (tid < 256) lut[tid] = p.lut[tid];
        __syncthreads();
        using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
        const u32x4* fa = (const u32x4*)(p.frames + e * 2 * (int64_t)HW);
        const u32x4* fb = (const u32x4*)(p.frames + e * 2 * (int64_t)HW + (int64_t)HW);
        const int items = HW / 16;                                 // 16 pixels (16 index bytes) per item
        constexpr int UNR = 3;
        for (int it = tid; it < items; it += kFrameBlock * UNR) {
            u32x4 a[UNR], b[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int j = it + u * kFrameBlock;
                if (j < items) { a[u] = __builtin_nontemporal_load(fa + j); b[u] = __builtin_nontemporal_load(fb + j); }
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int j = it + u * kFrameBlock;
                if (j >= items) continue;
                u32x4 g;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t ia = a[u][w], ib = b[u][w];
                    uint32_t out = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t m = max_u8x4(lut[(ia >> (8 * k)) & 255u], lut[(ib >> (8 * k)) & 255u]);   // cv::max of the two coloured pixels
                        out |= gray_rgbx<GRAY15>(m) << (8 * k);
                    }
                    g[w] = out;
                }
                ((u32x4*)gray)[j] = g;
            }
        }
    } else {
    const uint32_t* fa = (const uint32_t*)(p.frames + e * 2 * (int64_t)HW * 3);
    const uint32_t* fb = (const uint32_t*)(p.frames + e * 2 * (int64_t)HW * 3 + (int64_t)HW * 3);
    const int items = HW / 4;                                      // 4 pixels (12 B) per item
    // One item: byte-wise max of the two frames' 12 bytes, gray of the 4 pixels, one 4-B LDS store.
    auto item = [&](int j, const u32x3& a, const u32x3& b) {
        const uint32_t w0 = max_u8x4(a.x, b.x), w1 = max_u8x4(a.y, b.y), w2 = max_u8x4(a.z, b.z);
        // bytes: w0 = R0 G0 B0 R1 | w1 = G1 B1 R2 G2 | w2 = B2 R3 G3 B3   (little endian)
        const uint32_t g0 = gray_rgbx<GRAY15>(w0);                                       // v_alignbyte_b32 lines each pixel's three bytes up
        const uint32_t g1 = gray_rgbx<GRAY15>(__builtin_amdgcn_alignbyte(w1, w0, 3));
        const uint32_t g2 = gray_rgbx<GRAY15>(__builtin_amdgcn_alignbyte(w2, w1, 2));
        const uint32_t g3 = gray_rgbx<GRAY15>(w2 >> 8);
        gray32[j] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    };
    // Software pipeline over batches of UNR items per thread, two register sets in ping-pong: the next batch's loads are issued before
    // the current batch's arithmetic, so a wave always has 2 x UNR 12-B loads in flight (the single-set form waited for a whole batch,
    // computed, and only then issued the next).  Full batches carry no bounds checks; the ragged end goes item by item.
    constexpr int UNR = GYMHIP_ATARI_UNR, STEP = kFrameBlock * UNR;
    const int full_end = items - (UNR - 1) * kFrameBlock;          // it < full_end: all UNR items of the batch exist
    auto load = [&](int it, u32x3 (&a)[UNR], u32x3 (&b)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) { a[u] = load_px4(fa, it + u * kFrameBlock); b[u] = load_px4(fb, it + u * kFrameBlock); }
    };
    auto work = [&](int it, const u32x3 (&a)[UNR], const u32x3 (&b)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) item(it + u * kFrameBlock, a[u], b[u]);
    };
    int it = tid;
    u32x3 a0[UNR], b0[UNR], a1[UNR], b1[UNR];
    if (it < full_end) load(it, a0, b0);
    while (it < full_end) {
        if (it + STEP < full_end) load(it + STEP, a1, b1);
        work(it, a0, b0);
        it += STEP;
        if (it >= full_end) break;
        if (it + STEP < full_end) load(it + STEP, a0, b0);
        work(it, a1, b1);
        it += STEP;
    }
    for (; it < items; it += kFrameBlock) item(it, load_px4(fa, it), load_px4(fb, it));
    }

    // ---- phase 2: INTER_AREA out of LDS.  A thread owns 4 horizontally adjacent output columns and walks down the rows; the 4 results
    // leave as one 4-B store.  VALU-lean on purpose (the kernel overlaps this phase with other workgroups' loads, and what it costs is
    // issue slots): an output column's x-taps are CONSECUTIVE source pixels (computeResizeAreaTab), so a column keeps one base index and
    // its taps are immediate offsets of the LDS reads (a padded tap reads the next pixel and multiplies it by its weight 0: adds +0);
    // columns are processed in pairs with packed fp32 multiplies and adds (v_pk_mul_f32 / v_pk_add_f32: per lane the same IEEE
    // operations in the same order as OpenCV's scalar loop); the horizontal sum starts from the first product rather than from 0 + product
    // (the products are >= +0, so 0 + p == p bit for bit); v_cvt_pk_u8_f32 is saturate_cast<uchar> (round half to even, clamp:
    // tools/cvt_probe.hip) and packs the byte in the same instruction. -----------------------------------------------------------------
    using f32x2 = float __attribute__((ext_vector_type(2)));
    const int dw4 = p.dw / 4, rows_per_pass = kFrameBlock / dw4;
    const int qx = tid % dw4, ry = tid / dw4;
    int x0[4]; f32x2 xa01[NT], xa23[NT];
#pragma unroll
    for (int q = 0; q < 4; q++) x0[q] = p.xsi[(qx * 4 + q) * kMaxT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        xa01[k] = f32x2{p.xal[(qx * 4 + 0) * kMaxT + k], p.xal[(qx * 4 + 1) * kMaxT + k]};
        xa23[k] = f32x2{p.xal[(qx * 4 + 2) * kMaxT + k], p.xal[(qx * 4 + 3) * kMaxT + k]};
    }
    __syncthreads();
    uint32_t* dst32 = (uint32_t*)(p.out + e * p.env_stride + p.slot_offset);
    if (ry < rows_per_pass) {
#pragma unroll GYMHIP_ATARI_P2_UNROLL
        for (int dy = ry; dy < p.dh; dy += rows_per_pass) {
            f32x2 sum01, sum23;
#pragma unroll GYMHIP_ATARI_J_UNROLL
            for (int j = 0; j < NT; j++) {
                const float beta = yal[dy * kMaxT + j];
                const uint8_t* S = gray + ysi[dy * kMaxT + j] * p.W;
                const uint8_t *c0 = S + x0[0], *c1 = S + x0[1], *c2 = S + x0[2], *c3 = S + x0[3];
                f32x2 b01 = f32x2{(float)c0[0], (float)c1[0]} * xa01[0];
                f32x2 b23 = f32x2{(float)c2[0], (float)c3[0]} * xa23[0];
#pragma unroll
                for (int k = 1; k < NT; k++) {
                    b01 = b01 + f32x2{(float)c0[k], (float)c1[k]} * xa01[k];
                    b23 = b23 + f32x2{(float)c2[k], (float)c3[k]} * xa23[k];
                }
                const f32x2 bb = f32x2{beta, beta};
                if (j == 0) { sum01 = bb * b01; sum23 = bb * b23; }
                else { sum01 = sum01 + bb * b01; sum23 = sum23 + bb * b23; }
            }
            uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(sum01.x, 0u, 0u);
            px = __builtin_amdgcn_cvt_pk_u8_f32(sum01.y, 1u, px);
            px = __builtin_amdgcn_cvt_pk_u8_f32(sum23.x, 2u, px);
            px = __builtin_amdgcn_cvt_pk_u8_f32(sum23.y, 3u, px);
            dst32[dy * dw4 + qx] = px;
        }
    }

    // ---- optional SB3-style clear of the older planes when the env finished an episode --------------------------
    if (p.clear_on_done && p.ring && p.done && p.done[e]) {
        const int plane4 = p.dh * p.dw / 4;
        for (int s = 0; s < p.n_stack; s++) {
            if (s == p.slot) continue;
            uint32_t* pl = (uint32_t*)(p.ring + (e * p.n_stack + s) * (int64_t)(p.dh * p.dw));
            for (int k = tid; k < plane4; k += kFrameBlock) pl[k] = 0u;
        }
    }
}

// Item tid + 512 m of a frame, addressed as (uniform base + 6144 m) + 12 tid with the uniform part pinned in SGPRs (the empty asm makes
// it opaque): left alone, the compiler folds each load's constant into a VGPR offset of its own and hoists them all out of the frame
// loop — one VGPR per load of the unrolled pipeline, 15 of the kernel's 80.  This way every load shares the one offset register.
__device__ inline u32x3 load_px4_uniform_base(const uint8_t* frame, int m, int tid) {
    const uint8_t* base = frame + (int64_t)m * (kFrameBlock * 12);
    asm("" : "+s"(base));
    return load_px4((const uint32_t*)base, tid);
}

// A workgroup barrier that orders LDS traffic only: every LDS operation of this wave has completed (lgkmcnt = 0) before it arrives.
// __syncthreads() would also wait for the wave's outstanding GLOBAL loads and stores (vmcnt = 0) — exactly what the persistent kernel
// below must not do: the next frame's first loads and the previous frame's output stores stay in flight across it.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The Atari geometry itself — 210 x 160 RGB screens to 84 x 84 planes (warp_frame.h:41-47 on AtariEnv's frames) — with every extent a
// compile-time constant, as a PERSISTENT kernel: the grid is what the chip holds at once and a workgroup walks frames
// e = blockIdx.x, += gridDim.x.  Same arithmetic as atari_preprocess_kernel, item for item; what differs is the schedule:
//   * phase 1 is straight-line code (8 batches of 2 items per thread + a ragged item), so every s_waitcnt is an exact count;
//   * the next frame's first batch of loads is issued BEFORE the resampler runs and lands while it computes;
//   * the resampler leaves the finished plane in LDS and each thread sends its own four dwords to memory from INSIDE the next
//     frame's load stream, where their acknowledgement is behind loads the wave waits for anyway (vmcnt counts in order): no
//     workgroup ever sits at its end waiting for stores with nothing else in flight.  Only a workgroup's last plane goes out alone.
// MI355X, 4096 envs: 0.163 ms (one workgroup per frame) -> 0.157.  Where the rest goes (FDIAG builds): loads + gray alone 0.127 ms,
// + resampler 0.142, + the planes' stores 0.157.  The 29 MB of stores cost as much as 150 MB of loads and the cost is memory-side:
// the same for 4-B and 16-B stores, early or late in the stream, planes on 128-B lines or not; nothing when every workgroup stores
// into the same plane; half when a workgroup stores all its planes in one burst at its end (profiles/NOTES.md, round 3).
struct AtariGeo {
    static constexpr int H = 210, W = 160, DH = 84, DW = 84, HW = H * W, NT = 3;
    static constexpr int DW4 = DW / 4;                              // 21 output dwords per row
    static constexpr int ROWS_PER_PASS = kFrameBlock / DW4;         // 24 rows per pass of the workgroup
    static constexpr int PASS_WORDS = ROWS_PER_PASS * DW4;          // 504 threads resample; a thread's dwords: tid, tid + 504, ..
    static constexpr int PLANE_WORDS = DH * DW4;                    // 1764
    static constexpr int PASSES = (PLANE_WORDS + PASS_WORDS - 1) / PASS_WORDS;   // 4 (the last one half full)
    static constexpr int XREC = 4 + 4 * NT;                         // per column quad: 4 base indices + NT x 4 weights
};

#ifndef GYMHIP_ATARI_FDIAG
// Timing experiments only (results WRONG; profiles/r03_atari_persistent_decomposition.txt): bit 0 = planes are not stored (but a workgroup's
// last), bit 1 = no resampler (the plane is a copy of the gray image's first bytes), bit 2 = every workgroup stores into the same plane,
// bit 5 = a workgroup stores all its planes in one burst when it ends.
#define GYMHIP_ATARI_FDIAG 0
#endif
template <int GRAY15>
__global__ __launch_bounds__(kFrameBlock, GYMHIP_ATARI_WAVES) void atari_preprocess_fixed_kernel(const AtariParams p) {
    using G = AtariGeo;
    using f32x2 = float __attribute__((ext_vector_type(2)));
    using f32x4 = float __attribute__((ext_vector_type(4)));
    using i32x4 = int __attribute__((ext_vector_type(4)));
    __shared__ __align__(16) uint32_t gray32[G::HW / 4];            // the gray image
    __shared__ __align__(16) uint32_t stage32[G::PLANE_WORDS];      // the finished plane, parked until the next frame's loads are under way
    __shared__ __align__(16) float xrec[G::DW4 * G::XREC];          // resampler constants per column quad
    __shared__ __align__(16) int32_t yoff[G::DH * G::NT];           // y-taps: byte offset of the source row
    __shared__ __align__(16) float ybeta[G::DH * G::NT];            // y-taps: weight
    const uint8_t* gray = (const uint8_t*)gray32;
    const int tid = threadIdx.x;

    for (int k = tid; k < G::DH * G::NT; k += kFrameBlock) {
        const int d = k / G::NT, j = k % G::NT;
        yoff[k] = p.ysi[d * kMaxT + j] * G::W; ybeta[k] = p.yal[d * kMaxT + j];
    }
    // The resampler's per-thread constants live in LDS and are read back at the start of every phase 2: held in registers across
    // phase 1 they push the kernel past 80 VGPRs, and a spilled register's reload waits for vmcnt(0) — for every load in flight.
    for (int k = tid; k < G::DW4 * G::XREC; k += kFrameBlock) {
        const int q = k / G::XREC, w = k % G::XREC;
        if (w < 4) ((int32_t*)xrec)[k] = p.xsi[(q * 4 + w) * kMaxT];                  // base source index of column 4q + w (its taps are consecutive pixels)
        else xrec[k] = p.xal[(q * 4 + ((w - 4) & 3)) * kMaxT + ((w - 4) >> 2)];     // tap (w-4)/4 of column 4q + (w-4)%4
    }
    __syncthreads();

    // Items of 4 pixels = 12 B per frame: 8400 = 16 x 512 + 208, i.e. 16 per thread and one more for 208 threads.
    constexpr int64_t kFrameBytes = G::HW * 3;
    constexpr int kItems = G::HW / 4, kFull = kItems / kFrameBlock, kRagged = kItems % kFrameBlock;
    auto load_item = [&](const uint8_t* frame, int m) { return load_px4_uniform_base(frame, m, tid); };   // item tid + 512 m
    auto work_item = [&](int m, const u32x3& a, const u32x3& b) {                   // max of the two frames -> gray -> LDS
        const uint32_t w0 = max_u8x4(a.x, b.x), w1 = max_u8x4(a.y, b.y), w2 = max_u8x4(a.z, b.z);
        // bytes: w0 = R0 G0 B0 R1 | w1 = G1 B1 R2 G2 | w2 = B2 R3 G3 B3   (little endian); v_alignbyte_b32 lines each pixel's three bytes up
        const uint32_t g0 = gray_rgbx<GRAY15>(w0);
        const uint32_t g1 = gray_rgbx<GRAY15>(__builtin_amdgcn_alignbyte(w1, w0, 3));
        const uint32_t g2 = gray_rgbx<GRAY15>(__builtin_amdgcn_alignbyte(w2, w1, 2));
        const uint32_t g3 = gray_rgbx<GRAY15>(w2 >> 8);
        gray32[tid + kFrameBlock * m] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    };
    // Two register sets of UNR items per frame in ping-pong: set 0 takes batches 0, 2, ..; set 1 takes batches 1, 3, ..
    constexpr int UNR = 2, kBatches = kFull / UNR;
    static_assert(kFull % UNR == 0 && kBatches % 2 == 0, "phase 1 is written for an even number of full batches");
    u32x3 a0[UNR], b0[UNR], a1[UNR], b1[UNR];
    auto load_batch = [&](const uint8_t* fa, int batch, u32x3 (&a)[UNR], u32x3 (&b)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) { a[u] = load_item(fa, batch * UNR + u); b[u] = load_item(fa + kFrameBytes, batch * UNR + u); }
    };
    auto work_batch = [&](int batch, const u32x3 (&a)[UNR], const u32x3 (&b)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; u++) work_item(batch * UNR + u, a[u], b[u]);
    };
    // The parked plane goes to memory: four dwords per thread (504 threads; the fourth for half of them).
    auto flush = [&](uint32_t* dst) {
        if (tid < G::PASS_WORDS) {
#pragma unroll
            for (int n = 0; n < G::PASSES; n++)
                if ((n + 1) * G::PASS_WORDS <= G::PLANE_WORDS || tid + n * G::PASS_WORDS < G::PLANE_WORDS) dst[tid + n * G::PASS_WORDS] = stage32[tid + n * G::PASS_WORDS];
        }
    };

    int64_t e = blockIdx.x;
    if (e >= p.n_frames) return;
    load_batch(p.frames + e * 2 * kFrameBytes, 0, a0, b0);
    uint32_t* pend = nullptr;                                      // where the parked plane belongs (null: nothing parked)
#if GYMHIP_ATARI_FDIAG & 32
    __shared__ uint32_t* pend_list[64];
    int n_pend = 0;
#endif
    for (;;) {
        const uint8_t* fa = p.frames + e * 2 * kFrameBytes;
        // the done flag is read up here, in front of the load stream: read after the resampler, its wait (vmcnt counts in order) would also be a wait for the next frame's first batch
        const bool clear = p.clear_on_done && p.ring && p.done && p.done[e];
        // ---- phase 1: max of the two frames -> gray -> LDS; set 0 holds batch 0 already ---------------------------------
#pragma unroll
        for (int b = 0; b < kBatches; b += 2) {
            load_batch(fa, b + 1, a1, b1);
#if !(GYMHIP_ATARI_FDIAG & (1 | 32))
            if (b == 0 && pend) flush(pend);                       // the previous frame's plane leaves between this frame's 2nd and 3rd batch
#elif defined(GYMHIP_ATARI_BURST)   // with bit 5: the planes leave in bursts of GYMHIP_ATARI_BURST instead of all at the end
            if (b == 0 && n_pend >= GYMHIP_ATARI_BURST) { for (int i = 0; i < n_pend; i++) flush(pend_list[i]); n_pend = 0; }
#endif
            work_batch(b, a0, b0);
            if (b + 2 < kBatches) load_batch(fa, b + 2, a0, b0);
            else if (kRagged && tid < kRagged) { a0[0] = load_item(fa, kFull); b0[0] = load_item(fa + kFrameBytes, kFull); }
            work_batch(b + 1, a1, b1);
        }
        if (kRagged && tid < kRagged) work_item(kFull, a0[0], b0[0]);
        lds_barrier();                                             // the gray image is complete

        const int64_t en = e + gridDim.x;
        const bool more = en < p.n_frames;
        if (more) load_batch(p.frames + en * 2 * kFrameBytes, 0, a0, b0);   // rides across the resampler

        // ---- phase 2: INTER_AREA out of LDS into the parked plane (see atari_preprocess_kernel for the arithmetic) -------
        if (tid < G::PASS_WORDS && !(GYMHIP_ATARI_FDIAG & 2)) {
            const int qx = tid % G::DW4, ry = tid / G::DW4;
            const i32x4 x0 = *(const i32x4*)(xrec + qx * G::XREC);
            f32x2 xa01[G::NT], xa23[G::NT];
#pragma unroll
            for (int k = 0; k < G::NT; k++) {
                const f32x4 t = *(const f32x4*)(xrec + qx * G::XREC + 4 + 4 * k);
                xa01[k] = f32x2{t.x, t.y}; xa23[k] = f32x2{t.z, t.w};
            }
#pragma unroll 1
            for (int dy = ry; dy < G::DH; dy += G::ROWS_PER_PASS) {
                f32x2 sum01, sum23;
#pragma unroll 1
                for (int j = 0; j < G::NT; j++) {
                    const float beta = ybeta[dy * G::NT + j];
                    const uint8_t* S = gray + yoff[dy * G::NT + j];
                    const uint8_t *c0 = S + x0[0], *c1 = S + x0[1], *c2 = S + x0[2], *c3 = S + x0[3];
                    f32x2 b01 = f32x2{(float)c0[0], (float)c1[0]} * xa01[0];
                    f32x2 b23 = f32x2{(float)c2[0], (float)c3[0]} * xa23[0];
#pragma unroll
                    for (int k = 1; k < G::NT; k++) {
                        b01 = b01 + f32x2{(float)c0[k], (float)c1[k]} * xa01[k];
                        b23 = b23 + f32x2{(float)c2[k], (float)c3[k]} * xa23[k];
                    }
                    const f32x2 bb = f32x2{beta, beta};
                    if (j == 0) { sum01 = bb * b01; sum23 = bb * b23; }
                    else { sum01 = sum01 + bb * b01; sum23 = sum23 + bb * b23; }
                }
                uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(sum01.x, 0u, 0u);
                px = __builtin_amdgcn_cvt_pk_u8_f32(sum01.y, 1u, px);
                px = __builtin_amdgcn_cvt_pk_u8_f32(sum23.x, 2u, px);
                px = __builtin_amdgcn_cvt_pk_u8_f32(sum23.y, 3u, px);
                stage32[dy * G::DW4 + qx] = px;                    // = stage32[tid + pass * PASS_WORDS]: read back by this thread only
            }
        }
#if GYMHIP_ATARI_FDIAG & 2
        for (int k = tid; k < G::PLANE_WORDS; k += kFrameBlock) stage32[k] = gray32[k];
#endif
        pend = (uint32_t*)(p.out + e * p.env_stride + p.slot_offset);
#if GYMHIP_ATARI_FDIAG & 4
        pend = (uint32_t*)(p.out + p.slot_offset);
#endif

        // ---- optional SB3-style clear of the older planes when the env finished an episode --------------------------
        if (clear) {
            for (int s = 0; s < p.n_stack; s++) {
                if (s == p.slot) continue;
                uint32_t* pl = (uint32_t*)(p.ring + (e * p.n_stack + s) * (int64_t)(G::DH * G::DW));
                for (int k = tid; k < G::PLANE_WORDS; k += kFrameBlock) pl[k] = 0u;
            }
        }
#if GYMHIP_ATARI_FDIAG & 32
        if (tid == 0 && n_pend < 64) pend_list[n_pend] = pend;
        n_pend++;
#endif
        if (!more) break;
        lds_barrier();                                             // every wave is done reading the gray image: the next frame may overwrite it
        e = en;
    }
#if GYMHIP_ATARI_FDIAG & 32
    __syncthreads();
    for (int i = 0; i < n_pend && i < 64; i++) flush(pend_list[i]);
    return;
#endif
    flush(pend);                                                   // the workgroup's last plane
}

// VecFrameStack on its own: planes that are ALREADY dst_h x dst_w gray images go into ring slot `slot` (one workgroup per env
// plane, 4-B units: dst_w % 4 == 0), with the same optional clear of a finished env's older planes as the fused kernel.
__global__ __launch_bounds__(kBlock) void push_planes_kernel(const uint32_t* __restrict__ planes, uint8_t* ring, int64_t n, int plane4, int k, int slot,
                                                             const uint8_t* done, int clear_on_done) {
    for (int64_t e = blockIdx.x; e < n; e += gridDim.x) {
        const uint32_t* src = planes + e * plane4;
        uint32_t* env0 = (uint32_t*)ring + e * k * (int64_t)plane4;
        uint32_t* dst = env0 + (int64_t)slot * plane4;
        for (int i = threadIdx.x; i < plane4; i += kBlock) dst[i] = __builtin_nontemporal_load(src + i);
        if (clear_on_done && done && done[e]) {
            for (int s = 0; s < k; s++) {
                if (s == slot) continue;
                uint32_t* pl = env0 + (int64_t)s * plane4;
                for (int i = threadIdx.x; i < plane4; i += kBlock) pl[i] = 0u;
            }
        }
    }
}

// One plane for one env, or for all of them (gymhip_atari_push_index).  `only_env` >= 0: that env starts over — its other slots are
// zeroed and the plane becomes its newest frame, nobody else is touched.  `only_env` < 0: the same plane goes into slot `slot` of
// EVERY env (the reference's VecFrameStack::reset(int) as written: a [H, W, 1] frame broadcast over the batch).
__global__ __launch_bounds__(kBlock) void push_one_plane_kernel(const uint32_t* __restrict__ plane, uint8_t* ring, int64_t n, int plane4, int k, int slot, int64_t only_env) {
    const int64_t lo = only_env >= 0 ? only_env : blockIdx.x, hi = only_env >= 0 ? only_env + 1 : n, step = only_env >= 0 ? 1 : gridDim.x;
    if (only_env >= 0 && blockIdx.x != 0) return;
    for (int64_t e = lo; e < hi; e += step) {
        uint32_t* env0 = (uint32_t*)ring + e * k * (int64_t)plane4;
        for (int i = threadIdx.x; i < plane4; i += kBlock) env0[(int64_t)slot * plane4 + i] = plane[i];
        if (only_env >= 0) {
            for (int s = 0; s < k; s++) {
                if (s == slot) continue;
                for (int i = threadIdx.x; i < plane4; i += kBlock) env0[(int64_t)s * plane4 + i] = 0u;
            }
        }
    }
}

// ring u8 [N][k][P] (slot `head` newest) -> NHWC u8 [N][P][k], channel c = (head + 1 + c) % k (oldest first).
// `packed`: out is 4-byte aligned, so the four channels of a pixel may go out as one dword (host-side check; a byte view at an odd
// offset takes the byte path).
__global__ __launch_bounds__(kBlock) void materialize_nhwc_kernel(const uint8_t* ring, uint8_t* out, int64_t n, int P, int k, int head, bool packed) {
    const int64_t total = n * (int64_t)P;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
        const int64_t e = j / P; const int px = (int)(j - e * P);
        if (k == 4 && packed) {
            uint32_t w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) w |= (uint32_t)ring[(e * 4 + ((head + 1 + c) & 3)) * (int64_t)P + px] << (8 * c);
            ((uint32_t*)out)[j] = w;
        } else {
            for (int c = 0; c < k; c++) out[j * k + c] = ring[(e * k + ((head + 1 + c) % k)) * (int64_t)P + px];
        }
    }
}

// The k = 4 case of both NHWC views, the one every Atari stack uses: a workgroup owns a piece of ONE env's planes (no division per
// element), a thread takes 4 pixels — one dword from each channel's plane — transposes the 4 x 4 bytes in registers
// (v_perm_b32) and stores the 16 output bytes as one vector.  `slot0` < 0: channel 0 is zeros (as-written view straight after the
// first update).  RING_NT / OUT_NT as in the NCHW kernel: ring and view of a 4096-env engine (116 MB each) fit the Infinity Cache
// and the consumer reads the view next, so both use the default policy there; larger engines stream.
template <bool STREAM>
__global__ __launch_bounds__(kBlock) void materialize_nhwc4_kernel(const uint8_t* __restrict__ ring, uint8_t* __restrict__ out, int64_t n, int P4, int split,
                                                                   int slot0, int slot1, int slot2, int slot3) {
    using u32x4v = uint32_t __attribute__((ext_vector_type(4)));
    const int per = (P4 + split - 1) / split;
    for (int64_t item = blockIdx.x; item < n * split; item += gridDim.x) {
        const int64_t e = item / split; const int piece = (int)(item - e * split);
        const uint32_t* env = (const uint32_t*)ring + e * 4 * (int64_t)P4;
        const uint32_t *p0 = env + (int64_t)(slot0 < 0 ? 0 : slot0) * P4, *p1 = env + (int64_t)slot1 * P4, *p2 = env + (int64_t)slot2 * P4, *p3 = env + (int64_t)slot3 * P4;
        u32x4v* dst = (u32x4v*)out + e * (int64_t)P4;
        const int hi = min(P4, (piece + 1) * per);
        for (int q = piece * per + (int)threadIdx.x; q < hi; q += kBlock) {
            uint32_t w0, w1, w2, w3;
            if constexpr (STREAM) { w0 = __builtin_nontemporal_load(p0 + q); w1 = __builtin_nontemporal_load(p1 + q); w2 = __builtin_nontemporal_load(p2 + q); w3 = __builtin_nontemporal_load(p3 + q); }
            else { w0 = p0[q]; w1 = p1[q]; w2 = p2[q]; w3 = p3[q]; }
            if (slot0 < 0) w0 = 0u;
            // __builtin_amdgcn_perm(a, b, sel): selector values 0..3 pick bytes of b, 4..7 bytes of a
            const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);   // [w0.b0 w1.b0 w0.b1 w1.b1], [w0.b2 w1.b2 w0.b3 w1.b3]
            const uint32_t u0 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), u1 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
            u32x4v o;
            o.x = __builtin_amdgcn_perm(u0, t0, 0x05040100u);   // pixel 0: channels 0..3
            o.y = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            o.z = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            o.w = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
            if constexpr (STREAM) __builtin_nontemporal_store(o, dst + q); else dst[q] = o;
        }
    }
}

// The reference's stack AS WRITTEN (gymhip.h): channel 0 = plane `first_slot` of the ring (or zeros when first_slot < 0),
// channels 1 .. k-1 = the newest plane.
__global__ __launch_bounds__(kBlock) void materialize_nhwc_as_written_kernel(const uint8_t* ring, uint8_t* out, int64_t n, int P, int k, int head, int first_slot, bool packed) {
    const int64_t total = n * (int64_t)P;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
        const int64_t e = j / P; const int px = (int)(j - e * P);
        const uint32_t newest = ring[(e * k + head) * (int64_t)P + px];
        const uint32_t first = first_slot < 0 ? 0u : ring[(e * k + first_slot) * (int64_t)P + px];
        if (k == 4 && packed) {
            ((uint32_t*)out)[j] = first | (newest << 8) | (newest << 16) | (newest << 24);
        } else {
            out[j * k] = (uint8_t)first;
            for (int c = 1; c < k; c++) out[j * k + c] = (uint8_t)newest;
        }
    }
}

// ring -> NCHW [N][k][P] = float(u8) / 255 (vec_atari.cpp:74-76).  A write-dominated stream (1 B in, 4 or 2 B out per pixel,
// 231-462 MB per 4096-env call, written once and far larger than the Infinity Cache), so: non-temporal loads and stores, a
// workgroup owns one contiguous piece of one (env, channel) plane — the ring-slot arithmetic is per workgroup and scalar, no
// per-element integer division — and the launch has tens of thousands of workgroups (profiles/r01_stream_ceiling.txt: a
// non-temporal write-only stream runs at 4.8 TB/s from 1024 workgroups, 6.8 TB/s from 65,536).
// PX pixels per thread-item: 4 (f32: 4 B in, 16 B out) or 8 / 4 (16-bit: 8 B in, 16 B out / 4 B in, 8 B out).
using u32x2v = uint32_t __attribute__((ext_vector_type(2)));
using u32x4v = uint32_t __attribute__((ext_vector_type(4)));
using f32x4v = float __attribute__((ext_vector_type(4)));
template <int BF16> __device__ inline uint32_t to16(float v) {
    if (BF16) { const uint32_t u = __float_as_uint(v); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }   // finite, non-negative input: round to nearest even
    return (uint32_t)__half_as_ushort(__float2half_rn(v));
}
// SGN reads the byte as the reference's int8 buffer does (GYMHIP_STACK_AS_WRITTEN: 200 -> -56); else as the uint8 pixel.  A template
// argument: as a run-time flag its compare + subtract per pixel cost the default view 6 % (f32) / 14 % (bf16) of its rate.
template <int SGN> __device__ inline float px(uint32_t w, int b) {
    if constexpr (SGN) return (float)(int)(int8_t)((w >> (8 * b)) & 255u) / 255.f;
    else return (float)((w >> (8 * b)) & 255u) / 255.f;
}
template <int BF16, int SGN> __device__ inline uint32_t pack16(uint32_t w, int b) { return to16<BF16>(px<SGN>(w, b)) | (to16<BF16>(px<SGN>(w, b + 1)) << 16); }

// MODE 0: f32 out, 4 px per item.  MODE 1: 16-bit out, 8 px per item.  MODE 2: 16-bit out, 4 px per item (P % 8 != 0).
constexpr int kViewIntended = -2;   // `first` argument of the materialise kernels: the intended shift view
#ifndef GYMHIP_MAT_BLOCK
#define GYMHIP_MAT_BLOCK 256
#endif
#ifndef GYMHIP_MAT_UNR
#define GYMHIP_MAT_UNR 4       // items whose loads are issued before the first of them is converted and stored
#endif
constexpr int kMatBlock = GYMHIP_MAT_BLOCK;
// RING_NT: how the ring is read.  A ring that fits the 256 MiB Infinity Cache (4096 envs: 116 MB) is read with the default policy:
// the planes then stay in that cache from one step to the next — three of the four planes a step reads were read by the step
// before, the fourth was just written by the preprocess kernel — while the (4 - 8 x larger, write-once) output streams past it
// non-temporally.  Measured at 4096 envs: f32 0.105 -> 0.095 ms (0.69 -> 0.76 of 8 TB/s), bf16 0.070 -> 0.065; inside the
// step pipeline (preprocess, then this) f32 0.114 -> 0.103.  Larger rings are streamed non-temporally as before.
template <int MODE, int BF16, int AW, bool RING_NT>
__global__ __launch_bounds__(kMatBlock) void materialize_nchw_kernel(const uint8_t* __restrict__ ring, void* __restrict__ out, int64_t planes, int P, int k, int head, int split,
                                                                    int first) {
    // AW = 0: the intended shift view (`first` unused).  AW = 1: the as-written view — channel 0 = ring slot `first` (zeros if < 0),
    // channels 1.. = the newest frame, bytes read as int8.  AW = 2: the intended channel mapping with bytes read as int8 (the
    // as-written view straight after reset(), when the two stacks hold the same planes).
    constexpr int PX = MODE == 1 ? 8 : 4;
    constexpr int UNR = GYMHIP_MAT_UNR;
    using InV = typename std::conditional<MODE == 1, u32x2v, uint32_t>::type;
    const int items = P / PX;                                        // per plane
    const int per_piece = (items + split - 1) / split;
    for (int64_t w = blockIdx.x; w < planes * split; w += gridDim.x) {
        const int64_t plane = w / split; const int piece = (int)(w - plane * split);
        const int64_t e = plane / k; const int c = (int)(plane - e * k);
        const bool zero = AW == 1 && c == 0 && first < 0;
        const int slot = AW != 1 ? (head + 1 + c) % k : (c == 0 && first >= 0 ? first : head);   // intended: channel c = the (c+1)-th oldest frame
        const InV* src = (const InV*)(ring + (e * k + slot) * (int64_t)P);
        const int lo = piece * per_piece, hi = min(items, lo + per_piece);
        for (int q0 = lo + (int)threadIdx.x; q0 < hi; q0 += kMatBlock * UNR) {
            InV v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) { const int q = q0 + u * kMatBlock; if (q < hi) v[u] = zero ? InV{} : (RING_NT ? __builtin_nontemporal_load(src + q) : src[q]); }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int q = q0 + u * kMatBlock;
                if (q >= hi) continue;
                if constexpr (MODE == 0) {
                    const f32x4v o = {px<(AW != 0)>(v[u], 0), px<(AW != 0)>(v[u], 1), px<(AW != 0)>(v[u], 2), px<(AW != 0)>(v[u], 3)};
                    __builtin_nontemporal_store(o, (f32x4v*)out + plane * items + q);
                } else if constexpr (MODE == 1) {
                    const u32x4v o = {pack16<BF16, (AW != 0)>(v[u].x, 0), pack16<BF16, (AW != 0)>(v[u].x, 2), pack16<BF16, (AW != 0)>(v[u].y, 0), pack16<BF16, (AW != 0)>(v[u].y, 2)};
                    __builtin_nontemporal_store(o, (u32x4v*)out + plane * items + q);
                } else {
                    const u32x2v o = {pack16<BF16, (AW != 0)>(v[u], 0), pack16<BF16, (AW != 0)>(v[u], 2)};
                    __builtin_nontemporal_store(o, (u32x2v*)out + plane * items + q);
                }
            }
        }
    }
}

// Monitor -> EpisodicLife -> ClipReward over caller lanes, order of AtariWrapper::make (atari_wrappers.cpp:95-121).
__global__ __launch_bounds__(kBlock) void host_env_post_step_kernel(int64_t n, const float* rew_in, const uint8_t* done_in, const int32_t* lives_in,
                                                                    int32_t* st_lives, uint8_t* st_wrd, float* st_ret, float* st_len, int clip,
                                                                    float* rew_out, uint8_t* done_out, uint8_t* has_ep, float* ep_r, float* ep_l) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float r = rew_in[i]; bool d = done_in[i] != 0;
        float ret = st_ret[i] + r, len = st_len[i] + 1.f;               // monitor.h:78-79 (ordered float sum)
        has_ep[i] = d; ep_r[i] = d ? ret : 0.f; ep_l[i] = d ? len : 0.f;   // monitor.h:80-98
        if (d) { ret = 0.f; len = 0.f; }                                // Monitor::reset on the real reset (monitor.h:64-66)
        st_ret[i] = ret; st_len[i] = len;
        const int lives = lives_in[i];
        st_wrd[i] = d;                                                  // atari_wrappers.cpp:38
        if (0 < lives && lives < st_lives[i]) d = true;                 // :42-44
        st_lives[i] = lives;                                            // :45
        rew_out[i] = clip ? (r == 0.f ? r : (r > 0.f ? 1.f : -1.f)) : r;  // clip_reward.h:22-24
        done_out[i] = d;
    }
}

}  // namespace

struct gymhip_atari {
    gymhip_atari_config cfg;
    int device;
    uint8_t* ring;        // u8 [N][k][dh*dw]
    int head;             // slot of the newest frame
    int32_t *xsi, *ysi;   // padded tap tables [dst][kMaxT]
    float *xal, *yal;
    int max_taps;
    size_t lds_bytes;
    bool did_reset;
    int64_t updates_since_reset;   // VecFrameStack::stepWait calls since the last reset (the as-written view needs 0 / 1 / more)
    int view;                      // GYMHIP_STACK_INTENDED | GYMHIP_STACK_AS_WRITTEN
    uint32_t* lut;        // INDEX8 colour table (device, 256 entries R | G<<8 | B<<16) or null until gymhip_atari_set_palette
    bool fixed_geo;       // 210 x 160 -> 84 x 84 with <= 3 taps: RGB24 frames go through atari_preprocess_fixed_kernel (persistent)
    int persist_grid;     // workgroups of that kernel the chip holds at once: its grid
};

namespace {

int check_atari(gymhip_atari* h) {
    if (!h) return fail(GYMHIP_EINVAL, "null atari handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(GYMHIP_EHIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(e));
    return GYMHIP_OK;
}

template <typename T> int upload(const std::vector<T>& v, T** out) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)p;
    return GYMHIP_OK;
}

const void* fixed_kernel_of(const gymhip_atari* h) {
    return h->cfg.gray_bits == 15 ? (const void*)atari_preprocess_fixed_kernel<1> : (const void*)atari_preprocess_fixed_kernel<0>;
}

// GYMHIP_ATARI_GENERIC=1 sends the Atari geometry through the generic kernel too; GYMHIP_ATARI_GRID overrides the persistent grid
// (sweeps and tests; both are read at every launch).
template <int SRC>
void launch_variant(gymhip_atari* h, AtariParams p, hipStream_t s) {
    dim3 grid((unsigned)h->cfg.n_envs), block(kFrameBlock);
    const bool g15 = h->cfg.gray_bits == 15;
    const char* ge = getenv("GYMHIP_ATARI_GENERIC");
    const bool force_generic = ge && atoi(ge) != 0;
    if (SRC == 0 && h->fixed_geo && !force_generic) {
        p.n_frames = h->cfg.n_envs;
        int64_t g = h->persist_grid;
        if (const char* e = getenv("GYMHIP_ATARI_GRID")) { const long v = atol(e); if (v > 0) g = v; }
        g = std::max<int64_t>(1, std::min<int64_t>(g, p.n_frames));
        void* args[] = {(void*)&p};
        (void)hipLaunchKernel(fixed_kernel_of(h), dim3((unsigned)g), block, args, 0, s);   // the caller checks hipGetLastError
        return;
    }
    if (h->max_taps <= 3) {
        if (g15) hipLaunchKernelGGL((atari_preprocess_kernel<1, 3, SRC>), grid, block, h->lds_bytes, s, p);
        else hipLaunchKernelGGL((atari_preprocess_kernel<0, 3, SRC>), grid, block, h->lds_bytes, s, p);
    } else {
        if (g15) hipLaunchKernelGGL((atari_preprocess_kernel<1, 4, SRC>), grid, block, h->lds_bytes, s, p);
        else hipLaunchKernelGGL((atari_preprocess_kernel<0, 4, SRC>), grid, block, h->lds_bytes, s, p);
    }
}

int launch_preprocess(gymhip_atari* h, bool indexed, const uint8_t* frames, uint8_t* out, int64_t env_stride, int64_t slot_offset,
                      const uint8_t* done, bool ring_dst, hipStream_t s) {
    if (indexed && !h->lut) return fail(GYMHIP_ESTATE, "indexed frames need a colour table first (gymhip_atari_set_palette)");
    if (indexed && ((int64_t)h->cfg.src_h * h->cfg.src_w) % 16) return fail(GYMHIP_EINVAL, "indexed screens must be a multiple of 16 bytes (210x160 is)");
    AtariParams p{};
    p.lut = h->lut;
    p.frames = frames; p.out = out; p.env_stride = env_stride; p.slot_offset = slot_offset;
    p.done = done; p.ring = ring_dst ? h->ring : nullptr;
    p.xsi = h->xsi; p.xal = h->xal; p.ysi = h->ysi; p.yal = h->yal;
    p.H = h->cfg.src_h; p.W = h->cfg.src_w; p.dh = h->cfg.dst_h; p.dw = h->cfg.dst_w;
    p.n_stack = h->cfg.n_stack; p.slot = h->head; p.clear_on_done = h->cfg.clear_on_done;
    if (indexed) launch_variant<1>(h, p, s); else launch_variant<0>(h, p, s);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

}  // namespace

namespace {
// `first` for the materialise kernels.  As written: straight after reset the stack is [0, .., 0, f], the same as the intended
// view; afterwards channel 0 = the previous call's channel 1 — the previous frame, except on the first update after a reset
// with k >= 3, where the reset stack's channel 1 is still zero.
int view_first(const gymhip_atari* h) {
    if (h->view == GYMHIP_STACK_INTENDED || h->updates_since_reset == 0) return kViewIntended;
    const int k = h->cfg.n_stack;
    return (h->updates_since_reset == 1 && k >= 3) ? -1 : (h->head + k - 1) % k;
}

// Launch shape of the NCHW materialisation: `split` workgroups per (env, channel) plane, at most kMatMaxGrid workgroups (a
// grid-stride loop covers the rest).  GYMHIP_MAT_SPLIT / GYMHIP_MAT_GRID override for sweeps (tools/sweep_materialize.sh).
constexpr int kMatMaxGrid = 1 << 20;
void materialize_shape(int64_t planes, int items, bool ring_kept, int* split, unsigned* grid) {
    int sp = (int)((2 * (int64_t)items + 7 * kMatBlock / 2) / (7 * kMatBlock));   // default: ~3.5 items per thread (14 KB out per workgroup), best of tools/sweep_materialize.sh
    if (ring_kept) sp = 1;   // with the ring served by the Infinity Cache one workgroup per plane wins for both views (f32: 0.099 -> 0.084 ms, profiles/r02_sweeps.txt)
    if (const char* e = getenv("GYMHIP_MAT_SPLIT")) { const int v = atoi(e); if (v > 0) sp = v; }
    sp = std::max(1, std::min(sp, items));
    int64_t g = planes * sp;
    int64_t cap = kMatMaxGrid;
    if (const char* e = getenv("GYMHIP_MAT_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }
    *split = sp; *grid = (unsigned)std::min<int64_t>(g, cap);
}


// One launch of the NCHW materialisation: MODE / BF16 pick the pixel format, the handle's view picks AW.  As written, the stack
// straight after reset() equals the intended one in channel CONTENT, but its bytes are still read as int8.
template <int MODE, int BF16, bool RING_NT> int launch_nchw_policy(gymhip_atari* h, void* out, int px_per_item, hipStream_t s) {
    const int P = h->cfg.dst_h * h->cfg.dst_w, k = h->cfg.n_stack;
    const int64_t planes = h->cfg.n_envs * k;
    int split; unsigned grid;
    materialize_shape(planes, P / px_per_item, !RING_NT, &split, &grid);
    if (h->view == GYMHIP_STACK_AS_WRITTEN) {
        // straight after reset() the stack is [0, .., 0, f] in both views: the intended mapping, with the signed reading (AW = 2)
        const int first = view_first(h);
        if (first == kViewIntended) hipLaunchKernelGGL((materialize_nchw_kernel<MODE, BF16, 2, RING_NT>), dim3(grid), dim3(kMatBlock), 0, s, h->ring, out, planes, P, k, h->head, split, 0);
        else hipLaunchKernelGGL((materialize_nchw_kernel<MODE, BF16, 1, RING_NT>), dim3(grid), dim3(kMatBlock), 0, s, h->ring, out, planes, P, k, h->head, split, first);
    } else {
        hipLaunchKernelGGL((materialize_nchw_kernel<MODE, BF16, 0, RING_NT>), dim3(grid), dim3(kMatBlock), 0, s, h->ring, out, planes, P, k, h->head, split, 0);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}
template <int MODE, int BF16> int launch_nchw(gymhip_atari* h, void* out, int px_per_item, hipStream_t s) {
    const size_t ring_bytes = (size_t)h->cfg.n_envs * h->cfg.n_stack * h->cfg.dst_h * h->cfg.dst_w;
    return ring_bytes <= kRingKeepBytes ? launch_nchw_policy<MODE, BF16, false>(h, out, px_per_item, s) : launch_nchw_policy<MODE, BF16, true>(h, out, px_per_item, s);
}
}  // namespace

extern "C" {

int gymhip_atari_create(const gymhip_atari_config* cfg, int device, gymhip_atari** out) {
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    *out = nullptr;
    if (!cfg || cfg->struct_size != (int32_t)sizeof(gymhip_atari_config)) return fail(GYMHIP_EINVAL, "bad gymhip_atari_config.struct_size");
    if (cfg->n_envs <= 0 || cfg->n_envs > 0x7fffffff) return fail(GYMHIP_EINVAL, "n_envs out of range");
    if (cfg->src_h <= 0 || cfg->src_w <= 0 || cfg->dst_h <= 0 || cfg->dst_w <= 0 || cfg->n_stack <= 0) return fail(GYMHIP_EINVAL, "non-positive geometry");
    if (cfg->src_w % 4 || cfg->dst_w % 4 || ((int64_t)cfg->src_h * cfg->src_w * 3) % 16)
        return fail(GYMHIP_EINVAL, "src_w and dst_w must be multiples of 4 and a frame a multiple of 16 bytes (210x160x3 -> 84x84 is)");
    if (cfg->dst_h > cfg->src_h || cfg->dst_w > cfg->src_w) return fail(GYMHIP_EINVAL, "INTER_AREA path implemented for down-scaling only (warp_frame.h:43-45)");
    if (cfg->gray_bits != 14 && cfg->gray_bits != 15) return fail(GYMHIP_EINVAL, "gray_bits must be 14 or 15");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(GYMHIP_EINVAL, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    AxisTable xt = make_axis_table(cfg->src_w, cfg->dst_w), yt = make_axis_table(cfg->src_h, cfg->dst_h);
    if (xt.max_taps > kMaxT || yt.max_taps > kMaxT) return fail(GYMHIP_EINVAL, "scale factors above 3 need more than %d taps per output: not supported", kMaxT);
    if (cfg->dst_w / 4 > kFrameBlock) return fail(GYMHIP_EINVAL, "dst_w too large for one workgroup per frame");
    gymhip_atari* h = new (std::nothrow) gymhip_atari();
    if (!h) return fail(GYMHIP_EINVAL, "out of host memory");
    std::memset(h, 0, sizeof *h);
    h->cfg = *cfg; h->device = device; h->head = 0; h->did_reset = false; h->updates_since_reset = 0; h->view = GYMHIP_STACK_INTENDED;
    h->max_taps = std::max(xt.max_taps, yt.max_taps);
    const size_t hw = (size_t)cfg->src_h * cfg->src_w;
    h->lds_bytes = ((hw + 15) & ~(size_t)15) + (sizeof(int32_t) + sizeof(float)) * cfg->dst_h * kMaxT + 256 * sizeof(uint32_t);
    int rc = GYMHIP_OK;
    if (h->lds_bytes > 160 * 1024) rc = fail(GYMHIP_EINVAL, "gray frame + tables (%zu B) exceed the 160 KiB LDS of a CU", h->lds_bytes);
    if (!rc) rc = upload(xt.si, &h->xsi);
    if (!rc) rc = upload(xt.alpha, &h->xal);
    if (!rc) rc = upload(yt.si, &h->ysi);
    if (!rc) rc = upload(yt.alpha, &h->yal);
    if (!rc) {
        void* p = nullptr;
        const size_t bytes = (size_t)cfg->n_envs * cfg->n_stack * cfg->dst_h * cfg->dst_w;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) rc = fail(GYMHIP_EHIP, "ring allocation (%zu B) failed: %s", bytes, hipGetErrorString(e));
        h->ring = (uint8_t*)p;
    }
    if (!rc && h->lds_bytes > 64 * 1024) {
        const void* fns[] = {(const void*)atari_preprocess_kernel<1, 3, 0>, (const void*)atari_preprocess_kernel<0, 3, 0>,
                             (const void*)atari_preprocess_kernel<1, 4, 0>, (const void*)atari_preprocess_kernel<0, 4, 0>,
                             (const void*)atari_preprocess_kernel<1, 3, 1>, (const void*)atari_preprocess_kernel<0, 3, 1>,
                             (const void*)atari_preprocess_kernel<1, 4, 1>, (const void*)atari_preprocess_kernel<0, 4, 1>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess)
                rc = fail(GYMHIP_EHIP, "cannot raise the dynamic LDS limit to %zu B", h->lds_bytes);
    }
    h->fixed_geo = cfg->src_h == AtariGeo::H && cfg->src_w == AtariGeo::W && cfg->dst_h == AtariGeo::DH && cfg->dst_w == AtariGeo::DW && h->max_taps <= AtariGeo::NT;
    if (!rc && h->fixed_geo) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        int per_cu = 0;
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fixed_kernel_of(h), kFrameBlock, 0);
        h->persist_grid = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
        if (e != hipSuccess) rc = fail(GYMHIP_EHIP, "occupancy query for the preprocess kernel failed: %s", hipGetErrorString(e));
    }
    if (rc) { gymhip_atari_destroy(h); return rc; }
    *out = h;
    return GYMHIP_OK;
}

int gymhip_atari_destroy(gymhip_atari* h) {
    if (!h) return GYMHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {h->ring, h->xsi, h->xal, h->ysi, h->yal, h->lut};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return GYMHIP_OK;
}

int gymhip_atari_stack_buffer(gymhip_atari* h, uint8_t** ring, int32_t* head) {
    if (!h) return fail(GYMHIP_EINVAL, "null atari handle");
    if (ring) *ring = h->ring;
    if (head) *head = h->head;
    return GYMHIP_OK;
}

int gymhip_atari_set_palette(gymhip_atari* h, const uint32_t* zrgb256) {
    if (int rc = check_atari(h)) return rc;
    if (!zrgb256) return fail(GYMHIP_EINVAL, "palette is null");
    uint32_t packed[256];
    for (int k = 0; k < 256; k++) {   // 0x00RRGGBB (ale_c_wrapper.cpp:35-38) -> bytes R, G, B in memory order
        const uint32_t z = zrgb256[k];
        packed[k] = ((z >> 16) & 0xffu) | (((z >> 8) & 0xffu) << 8) | ((z & 0xffu) << 16);
    }
    if (!h->lut) { void* p = nullptr; HIP_TRY(hipMalloc(&p, sizeof packed)); h->lut = (uint32_t*)p; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->lut, packed, sizeof packed, hipMemcpyHostToDevice));
    return GYMHIP_OK;
}

static int preprocess_any(gymhip_atari* h, bool indexed, const uint8_t* frames, uint8_t* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames || !out) return fail(GYMHIP_EINVAL, "frames/out is null");
    return launch_preprocess(h, indexed, frames, out, (int64_t)h->cfg.dst_h * h->cfg.dst_w, 0, nullptr, false, (hipStream_t)stream);
}

static int reset_any(gymhip_atari* h, bool indexed, const uint8_t* frames, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames) return fail(GYMHIP_EINVAL, "frames is null");
    if (indexed && !h->lut) return fail(GYMHIP_ESTATE, "indexed frames need a colour table first (gymhip_atari_set_palette)");
    const int64_t plane = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    HIP_TRY(hipMemsetAsync(h->ring, 0, (size_t)h->cfg.n_envs * h->cfg.n_stack * plane, (hipStream_t)stream));   // stacked_observation.h:107
    h->head = h->cfg.n_stack - 1;
    h->did_reset = true;
    h->updates_since_reset = 0;
    return launch_preprocess(h, indexed, frames, h->ring, plane * h->cfg.n_stack, plane * h->head, nullptr, false, (hipStream_t)stream);
}

static int step_any(gymhip_atari* h, bool indexed, const uint8_t* frames, const uint8_t* done, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames) return fail(GYMHIP_EINVAL, "frames is null");
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "atari step before reset");
    if (indexed && !h->lut) return fail(GYMHIP_ESTATE, "indexed frames need a colour table first (gymhip_atari_set_palette)");
    const int64_t plane = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    h->updates_since_reset++;
    h->head = (h->head + 1) % h->cfg.n_stack;   // ring advance == the reference's roll(-1) + write-last (stacked_observation.h:126-130, intended semantics)
    return launch_preprocess(h, indexed, frames, h->ring, plane * h->cfg.n_stack, plane * h->head, done, true, (hipStream_t)stream);
}

int gymhip_atari_preprocess(gymhip_atari* h, const uint8_t* frames, uint8_t* out, void* stream) { return preprocess_any(h, false, frames, out, stream); }
int gymhip_atari_reset(gymhip_atari* h, const uint8_t* frames, void* stream) { return reset_any(h, false, frames, stream); }
int gymhip_atari_step(gymhip_atari* h, const uint8_t* frames, const uint8_t* done, void* stream) { return step_any(h, false, frames, done, stream); }
int gymhip_atari_preprocess_indexed(gymhip_atari* h, const uint8_t* screens, uint8_t* out, void* stream) { return preprocess_any(h, true, screens, out, stream); }
int gymhip_atari_reset_indexed(gymhip_atari* h, const uint8_t* screens, void* stream) { return reset_any(h, true, screens, stream); }
int gymhip_atari_step_indexed(gymhip_atari* h, const uint8_t* screens, const uint8_t* done, void* stream) { return step_any(h, true, screens, done, stream); }

static int push_any(gymhip_atari* h, const uint8_t* planes, const uint8_t* done, bool reset, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!planes) return fail(GYMHIP_EINVAL, "planes is null");
    if ((uintptr_t)planes % 4) return fail(GYMHIP_EINVAL, "planes must be 4-byte aligned");
    if (!reset && !h->did_reset) return fail(GYMHIP_ESTATE, "atari push before reset");
    const int64_t plane = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    hipStream_t s = (hipStream_t)stream;
    if (reset) {
        HIP_TRY(hipMemsetAsync(h->ring, 0, (size_t)h->cfg.n_envs * h->cfg.n_stack * plane, s));   // stacked_observation.h:107
        h->head = h->cfg.n_stack - 1;
        h->did_reset = true;
        h->updates_since_reset = 0;
    } else {
        h->updates_since_reset++;
        h->head = (h->head + 1) % h->cfg.n_stack;
    }
    const unsigned grid = (unsigned)std::min<int64_t>(h->cfg.n_envs, 1 << 16);
    hipLaunchKernelGGL(push_planes_kernel, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)planes, h->ring, (int64_t)h->cfg.n_envs, (int)(plane / 4),
                       h->cfg.n_stack, h->head, reset ? nullptr : done, h->cfg.clear_on_done);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}
int gymhip_atari_push_reset(gymhip_atari* h, const uint8_t* planes, void* stream) { return push_any(h, planes, nullptr, true, stream); }
int gymhip_atari_push(gymhip_atari* h, const uint8_t* planes, const uint8_t* done, void* stream) { return push_any(h, planes, done, false, stream); }

int gymhip_atari_push_index(gymhip_atari* h, int64_t index, const uint8_t* plane, int as_written, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!plane || ((uintptr_t)plane % 4)) return fail(GYMHIP_EINVAL, "plane is null or not 4-byte aligned");
    if (index < 0 || index >= h->cfg.n_envs) return fail(GYMHIP_EINVAL, "env index %lld out of range [0, %lld)", (long long)index, (long long)h->cfg.n_envs);
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "atari push_index before reset");
    const int64_t plane_bytes = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    hipStream_t s = (hipStream_t)stream;
    if (as_written) {   // StackedObservation::update over ONE env's frame: every stack shifts, the frame is broadcast (vec_frame_stack.cpp:34-37)
        h->updates_since_reset++;
        h->head = (h->head + 1) % h->cfg.n_stack;
        const unsigned grid = (unsigned)std::min<int64_t>(h->cfg.n_envs, 1 << 16);
        hipLaunchKernelGGL(push_one_plane_kernel, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)plane, h->ring, (int64_t)h->cfg.n_envs, (int)(plane_bytes / 4), h->cfg.n_stack, h->head, (int64_t)-1);
    } else {
        hipLaunchKernelGGL(push_one_plane_kernel, dim3(1), dim3(kBlock), 0, s, (const uint32_t*)plane, h->ring, (int64_t)h->cfg.n_envs, (int)(plane_bytes / 4), h->cfg.n_stack, h->head, index);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_atari_set_stack_view(gymhip_atari* h, int view) {
    if (int rc = check_atari(h)) return rc;
    if (view != GYMHIP_STACK_INTENDED && view != GYMHIP_STACK_AS_WRITTEN) return fail(GYMHIP_EINVAL, "view must be GYMHIP_STACK_INTENDED or GYMHIP_STACK_AS_WRITTEN");
    if (view == GYMHIP_STACK_AS_WRITTEN) {
        if (h->cfg.n_stack < 2) return fail(GYMHIP_EINVAL, "the as-written view needs n_stack >= 2 (with one channel the reference's stack never changes after reset)");
        if (h->cfg.clear_on_done) return fail(GYMHIP_EINVAL, "the as-written view needs clear_on_done = 0 (the reference never clears the stack)");
    }
    h->view = view;
    return GYMHIP_OK;
}

int gymhip_atari_materialize_nhwc(gymhip_atari* h, uint8_t* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    const int P = h->cfg.dst_h * h->cfg.dst_w, k = h->cfg.n_stack;
    const int first = view_first(h);
    hipStream_t s = (hipStream_t)stream;
    if (k == 4 && ((uintptr_t)out & 15) == 0) {   // dst_w % 4 == 0 (gymhip_atari_create) => P % 4 == 0
        const int hd = h->head, P4 = P / 4;
        const int split = std::max(1, std::min(P4, (P4 + 2 * kBlock - 1) / (2 * kBlock)));          // ~2 items per thread
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)h->cfg.n_envs * split, 1 << 20);
        const int s0 = first == kViewIntended ? (hd + 1) & 3 : first, s1 = first == kViewIntended ? (hd + 2) & 3 : hd, s2 = first == kViewIntended ? (hd + 3) & 3 : hd;
        const size_t ring_bytes = (size_t)h->cfg.n_envs * 4 * P;
        if (ring_bytes <= kRingKeepBytes / 2) hipLaunchKernelGGL(materialize_nhwc4_kernel<false>, dim3(grid), dim3(kBlock), 0, s, h->ring, out, (int64_t)h->cfg.n_envs, P4, split, s0, s1, s2, hd);
        else hipLaunchKernelGGL(materialize_nhwc4_kernel<true>, dim3(grid), dim3(kBlock), 0, s, h->ring, out, (int64_t)h->cfg.n_envs, P4, split, s0, s1, s2, hd);
    } else if (first == kViewIntended) {
        hipLaunchKernelGGL(materialize_nhwc_kernel, dim3(2048), dim3(kBlock), 0, s, h->ring, out, h->cfg.n_envs, P, k, h->head, ((uintptr_t)out & 3) == 0);
    } else {
        hipLaunchKernelGGL(materialize_nhwc_as_written_kernel, dim3(2048), dim3(kBlock), 0, s, h->ring, out, h->cfg.n_envs, P, k, h->head, first, ((uintptr_t)out & 3) == 0);
    }
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_atari_materialize_nchw_f32(gymhip_atari* h, float* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    return launch_nchw<0, 0>(h, out, 4, (hipStream_t)stream);       // dst_w % 4 == 0 (gymhip_atari_create) => P % 4 == 0
}
int gymhip_atari_materialize_nchw_16(gymhip_atari* h, void* out, int dtype, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    if (dtype != GYMHIP_OUT_F16 && dtype != GYMHIP_OUT_BF16) return fail(GYMHIP_EINVAL, "dtype must be GYMHIP_OUT_F16 or GYMHIP_OUT_BF16");
    const int P = h->cfg.dst_h * h->cfg.dst_w;
    hipStream_t s = (hipStream_t)stream;
    if (P % 8 == 0) return dtype == GYMHIP_OUT_BF16 ? launch_nchw<1, 1>(h, out, 8, s) : launch_nchw<1, 0>(h, out, 8, s);
    return dtype == GYMHIP_OUT_BF16 ? launch_nchw<2, 1>(h, out, 4, s) : launch_nchw<2, 0>(h, out, 4, s);
}

int gymhip_host_env_post_step(int64_t n, const float* reward_in, const uint8_t* done_in, const int32_t* lives_in, int32_t* st_lives,
                              uint8_t* st_was_real_done, float* st_return, float* st_length, int clip_reward, float* reward_out,
                              uint8_t* done_out, uint8_t* has_episode, float* ep_return, float* ep_length, void* stream) {
    if (n <= 0) return fail(GYMHIP_EINVAL, "n must be positive");
    if (!reward_in || !done_in || !lives_in || !st_lives || !st_was_real_done || !st_return || !st_length || !reward_out || !done_out || !has_episode ||
        !ep_return || !ep_length)
        return fail(GYMHIP_EINVAL, "null lane pointer");
    int64_t g = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(host_env_post_step_kernel, dim3((unsigned)(g < 2048 ? g : 2048)), dim3(kBlock), 0, (hipStream_t)stream, n, reward_in, done_in,
                       lives_in, st_lives, st_was_real_done, st_return, st_length, clip_reward, reward_out, done_out, has_episode, ep_return, ep_length);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

}  // extern "C"
