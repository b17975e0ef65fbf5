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
// 7,056 output bytes per env — 208,656 B per env-step, the algorithmic minimum.
//
// -ffp-contract=off (Makefile) keeps the resampler's mul/add separate, as OpenCV's baseline x86-64 build does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "gymhip.h"
#include "gymhip_internal.h"

using namespace gymhip;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxTaps = 1024;  // per axis; 210->84 needs 252, 160->84 needs 244

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(GYMHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Per-axis decimation table in per-destination form: taps of dst index d are [ofs[d], ofs[d+1]).
struct AxisTable {
    std::vector<int32_t> ofs;   // dsize + 1
    std::vector<int32_t> si;    // source index per tap
    std::vector<float> alpha;   // weight per tap
};

// OpenCV computeResizeAreaTab (imgproc/resize.cpp), generic (non-integer-scale) INTER_AREA path; scale = 1/(dsize/ssize).
AxisTable make_axis_table(int ssize, int dsize) {
    AxisTable t;
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    t.ofs.push_back(0);
    for (int d = 0; d < dsize; d++) {
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) { t.si.push_back(s1 - 1); t.alpha.push_back((float)((s1 - f1) / cell)); }
        for (int s = s1; s < s2; s++) { t.si.push_back(s); t.alpha.push_back((float)(1.0 / cell)); }
        if (f2 - s2 > 1e-3) { t.si.push_back(s2); t.alpha.push_back((float)(std::min(std::min(f2 - s2, 1.), cell) / cell)); }
        t.ofs.push_back((int32_t)t.si.size());
    }
    return t;
}

struct AtariParams {
    const uint8_t* frames;     // u8 [N][2][H][W][3]
    uint8_t* out;              // plane base: plane of env e starts at out + e * env_stride + slot_offset
    int64_t env_stride;        // bytes between consecutive envs' destination planes
    int64_t slot_offset;       // byte offset of the destination plane inside an env's block
    const uint8_t* done;       // u8 [N] or null
    uint8_t* ring;             // ring base (for clear-on-done), or null
    const int32_t* xofs; const int32_t* xsi; const float* xalpha;
    const int32_t* yofs; const int32_t* ysi; const float* yalpha;
    int nxt, nyt;              // tap counts
    int H, W, dh, dw, n_stack, slot, gray_bits, clear_on_done;
};

__device__ inline uint32_t max_u8x4(uint32_t a, uint32_t b) {
    // byte-wise unsigned max without widening each byte separately: compare even/odd bytes as 16-bit lanes
    const uint32_t m = 0x00FF00FFu;
    uint32_t ae = a & m, be = b & m, ao = (a >> 8) & m, bo = (b >> 8) & m;
    uint32_t e = ((ae & 0xFFFFu) > (be & 0xFFFFu) ? (ae & 0xFFFFu) : (be & 0xFFFFu)) | ((ae >> 16) > (be >> 16) ? (ae & 0xFFFF0000u) : (be & 0xFFFF0000u));
    uint32_t o = ((ao & 0xFFFFu) > (bo & 0xFFFFu) ? (ao & 0xFFFFu) : (bo & 0xFFFFu)) | ((ao >> 16) > (bo >> 16) ? (ao & 0xFFFF0000u) : (bo & 0xFFFF0000u));
    return e | (o << 8);
}

template <int GRAY15>
__device__ inline uint32_t gray_px(uint32_t r, uint32_t g, uint32_t b) {
    // cv::cvtColor(COLOR_RGB2GRAY) for 8-bit: fixed-point BT.601 weights (15-bit in OpenCV >= 4.x builds, 14-bit before)
    if (GRAY15) return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
    return (r * 4899u + g * 9617u + b * 1868u + (1u << 13)) >> 14;
}

__device__ inline uint32_t sat_u8(float v) {   // cv::saturate_cast<uchar>(float): round half to even, clamp
    int iv = (int)rintf(v);
    iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
    return (uint32_t)iv;
}

struct u32x3 { uint32_t x, y, z; };   // 12-B load unit: 4 whole RGB pixels
__device__ inline u32x3 load_px4(const uint32_t* base, int j) {
    u32x3 v;
    v.x = __builtin_nontemporal_load(base + 3 * j);
    v.y = __builtin_nontemporal_load(base + 3 * j + 1);
    v.z = __builtin_nontemporal_load(base + 3 * j + 2);
    return v;
}

template <int GRAY15>
__global__ __launch_bounds__(kBlock) void atari_preprocess_kernel(const AtariParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int HW = p.H * p.W;
    uint8_t* gray = lds;                                           // [H*W] u8
    int32_t* t_i = (int32_t*)(lds + ((HW + 15) & ~15));            // xofs[dw+1] yofs[dh+1] xsi[nxt] ysi[nyt]
    int32_t* xofs = t_i; int32_t* yofs = xofs + (p.dw + 1); int32_t* xsi = yofs + (p.dh + 1); int32_t* ysi = xsi + p.nxt;
    float* xalpha = (float*)(ysi + p.nyt); float* yalpha = xalpha + p.nxt;
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;

    for (int k = tid; k <= p.dw; k += kBlock) xofs[k] = p.xofs[k];
    for (int k = tid; k <= p.dh; k += kBlock) yofs[k] = p.yofs[k];
    for (int k = tid; k < p.nxt; k += kBlock) { xsi[k] = p.xsi[k]; xalpha[k] = p.xalpha[k]; }
    for (int k = tid; k < p.nyt; k += kBlock) { ysi[k] = p.ysi[k]; yalpha[k] = p.yalpha[k]; }

    // ---- phase 1: max of the two frames -> gray -> LDS ---------------------------------------------------------
    const uint32_t* fa = (const uint32_t*)(p.frames + e * 2 * (int64_t)HW * 3);
    const uint32_t* fb = (const uint32_t*)(p.frames + e * 2 * (int64_t)HW * 3 + (int64_t)HW * 3);
    const int items = HW / 4;                                      // 4 pixels (12 B) per item
    uint32_t* gray32 = (uint32_t*)gray;
    constexpr int UNR = 4;
    for (int it = tid; it < items; it += kBlock * UNR) {
        u32x3 a[UNR], b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int j = it + u * kBlock;
            if (j < items) { a[u] = load_px4(fa, j); b[u] = load_px4(fb, j); }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int j = it + u * kBlock;
            if (j >= items) continue;
            const uint32_t w0 = max_u8x4(a[u].x, b[u].x), w1 = max_u8x4(a[u].y, b[u].y), w2 = max_u8x4(a[u].z, b[u].z);
            // bytes: w0 = R0 G0 B0 R1 | w1 = G1 B1 R2 G2 | w2 = B2 R3 G3 B3   (little endian)
            const uint32_t g0 = gray_px<GRAY15>(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
            const uint32_t g1 = gray_px<GRAY15>(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
            const uint32_t g2 = gray_px<GRAY15>((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
            const uint32_t g3 = gray_px<GRAY15>((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
            gray32[j] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        }
    }
    __syncthreads();

    // ---- phase 2: INTER_AREA out of LDS, 4 horizontally adjacent outputs per thread -> one 4-B store -------------
    uint8_t* dst = p.out + e * p.env_stride + p.slot_offset;
    const int dw4 = p.dw / 4, outs = p.dh * dw4;
    for (int o = tid; o < outs; o += kBlock) {
        const int dy = o / dw4, dx0 = (o - dy * dw4) * 4;
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        const int y0 = yofs[dy], y1 = yofs[dy + 1];
        for (int j = y0; j < y1; j++) {
            const float beta = yalpha[j];
            const uint8_t* S = gray + ysi[j] * p.W;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float buf = 0.f;
                for (int k = xofs[dx0 + q]; k < xofs[dx0 + q + 1]; k++) buf += (float)S[xsi[k]] * xalpha[k];
                sum[q] = (j == y0) ? beta * buf : sum[q] + beta * buf;
            }
        }
        const uint32_t packed = sat_u8(sum[0]) | (sat_u8(sum[1]) << 8) | (sat_u8(sum[2]) << 16) | (sat_u8(sum[3]) << 24);
        ((uint32_t*)dst)[o] = packed;
    }

    // ---- optional SB3-style clear of the older planes when the env finished an episode --------------------------
    if (p.clear_on_done && p.ring && p.done && p.done[e]) {
        const int plane4 = p.dh * p.dw / 4;
        for (int s = 0; s < p.n_stack; s++) {
            if (s == p.slot) continue;
            uint32_t* pl = (uint32_t*)(p.ring + (e * p.n_stack + s) * (int64_t)(p.dh * p.dw));
            for (int k = tid; k < plane4; k += kBlock) pl[k] = 0u;
        }
    }
}

// ring u8 [N][k][P] (slot `head` newest) -> NHWC u8 [N][P][k], channel c = (head + 1 + c) % k (oldest first).
__global__ __launch_bounds__(kBlock) void materialize_nhwc_kernel(const uint8_t* ring, uint8_t* out, int64_t n, int P, int k, int head) {
    const int64_t total = n * (int64_t)P;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
        const int64_t e = j / P; const int px = (int)(j - e * P);
        if (k == 4) {
            uint32_t w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) w |= (uint32_t)ring[(e * 4 + ((head + 1 + c) & 3)) * (int64_t)P + px] << (8 * c);
            ((uint32_t*)out)[j] = w;
        } else {
            for (int c = 0; c < k; c++) out[j * k + c] = ring[(e * k + ((head + 1 + c) % k)) * (int64_t)P + px];
        }
    }
}

// ring -> NCHW f32 [N][k][P] = float(u8) / 255 (vec_atari.cpp:74-76), 4 pixels per thread.
__global__ __launch_bounds__(kBlock) void materialize_nchw_f32_kernel(const uint8_t* ring, float* out, int64_t n, int P, int k, int head) {
    const int P4 = P / 4;
    const int64_t total = n * (int64_t)k * P4;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
        const int64_t ec = j / P4; const int q = (int)(j - ec * P4);
        const int64_t e = ec / k; const int c = (int)(ec - e * k);
        const uint32_t w = ((const uint32_t*)(ring + (e * k + ((head + 1 + c) % k)) * (int64_t)P))[q];
        float4 v = make_float4((float)(w & 255u) / 255.f, (float)((w >> 8) & 255u) / 255.f, (float)((w >> 16) & 255u) / 255.f, (float)(w >> 24) / 255.f);
        ((float4*)out)[j] = v;
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
    int32_t *xofs, *xsi, *yofs, *ysi;
    float *xalpha, *yalpha;
    int nxt, nyt;
    size_t lds_bytes;
    bool did_reset;
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

int launch_preprocess(gymhip_atari* h, const uint8_t* frames, uint8_t* out, int64_t env_stride, int64_t slot_offset,
                      const uint8_t* done, bool ring_dst, hipStream_t s) {
    AtariParams p{};
    p.frames = frames; p.out = out; p.env_stride = env_stride; p.slot_offset = slot_offset;
    p.done = done; p.ring = ring_dst ? h->ring : nullptr;
    p.xofs = h->xofs; p.xsi = h->xsi; p.xalpha = h->xalpha; p.yofs = h->yofs; p.ysi = h->ysi; p.yalpha = h->yalpha;
    p.nxt = h->nxt; p.nyt = h->nyt;
    p.H = h->cfg.src_h; p.W = h->cfg.src_w; p.dh = h->cfg.dst_h; p.dw = h->cfg.dst_w;
    p.n_stack = h->cfg.n_stack; p.slot = h->head; p.gray_bits = h->cfg.gray_bits; p.clear_on_done = h->cfg.clear_on_done;
    dim3 grid((unsigned)h->cfg.n_envs), block(kBlock);
    if (h->cfg.gray_bits == 15) hipLaunchKernelGGL(atari_preprocess_kernel<1>, grid, block, h->lds_bytes, s, p);
    else hipLaunchKernelGGL(atari_preprocess_kernel<0>, grid, block, h->lds_bytes, s, p);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
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
    if ((int)xt.si.size() > kMaxTaps || (int)yt.si.size() > kMaxTaps) return fail(GYMHIP_EINVAL, "decimation table too large");
    gymhip_atari* h = new (std::nothrow) gymhip_atari();
    if (!h) return fail(GYMHIP_EINVAL, "out of host memory");
    std::memset(h, 0, sizeof *h);
    h->cfg = *cfg; h->device = device; h->head = 0; h->did_reset = false;
    h->nxt = (int)xt.si.size(); h->nyt = (int)yt.si.size();
    const size_t hw = (size_t)cfg->src_h * cfg->src_w;
    h->lds_bytes = ((hw + 15) & ~(size_t)15) + sizeof(int32_t) * (cfg->dst_w + 1 + cfg->dst_h + 1 + h->nxt + h->nyt) + sizeof(float) * (h->nxt + h->nyt);
    int rc = GYMHIP_OK;
    if (h->lds_bytes > 160 * 1024) rc = fail(GYMHIP_EINVAL, "gray frame + tables (%zu B) exceed the 160 KiB LDS of a CU", h->lds_bytes);
    if (!rc) rc = upload(xt.ofs, &h->xofs);
    if (!rc) rc = upload(xt.si, &h->xsi);
    if (!rc) rc = upload(xt.alpha, &h->xalpha);
    if (!rc) rc = upload(yt.ofs, &h->yofs);
    if (!rc) rc = upload(yt.si, &h->ysi);
    if (!rc) rc = upload(yt.alpha, &h->yalpha);
    if (!rc) {
        void* p = nullptr;
        const size_t bytes = (size_t)cfg->n_envs * cfg->n_stack * cfg->dst_h * cfg->dst_w;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) rc = fail(GYMHIP_EHIP, "ring allocation (%zu B) failed: %s", bytes, hipGetErrorString(e));
        h->ring = (uint8_t*)p;
    }
    if (!rc && h->lds_bytes > 64 * 1024) {
        hipError_t e1 = hipFuncSetAttribute((const void*)atari_preprocess_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        hipError_t e0 = hipFuncSetAttribute((const void*)atari_preprocess_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e1 != hipSuccess || e0 != hipSuccess) rc = fail(GYMHIP_EHIP, "cannot raise the dynamic LDS limit to %zu B", h->lds_bytes);
    }
    if (rc) { gymhip_atari_destroy(h); return rc; }
    *out = h;
    return GYMHIP_OK;
}

int gymhip_atari_destroy(gymhip_atari* h) {
    if (!h) return GYMHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {h->ring, h->xofs, h->xsi, h->xalpha, h->yofs, h->ysi, h->yalpha};
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

int gymhip_atari_preprocess(gymhip_atari* h, const uint8_t* frames, uint8_t* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames || !out) return fail(GYMHIP_EINVAL, "frames/out is null");
    return launch_preprocess(h, frames, out, (int64_t)h->cfg.dst_h * h->cfg.dst_w, 0, nullptr, false, (hipStream_t)stream);
}

int gymhip_atari_reset(gymhip_atari* h, const uint8_t* frames, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames) return fail(GYMHIP_EINVAL, "frames is null");
    const int64_t plane = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    HIP_TRY(hipMemsetAsync(h->ring, 0, (size_t)h->cfg.n_envs * h->cfg.n_stack * plane, (hipStream_t)stream));   // stacked_observation.h:107
    h->head = h->cfg.n_stack - 1;
    h->did_reset = true;
    return launch_preprocess(h, frames, h->ring, plane * h->cfg.n_stack, plane * h->head, nullptr, false, (hipStream_t)stream);
}

int gymhip_atari_step(gymhip_atari* h, const uint8_t* frames, const uint8_t* done, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!frames) return fail(GYMHIP_EINVAL, "frames is null");
    if (!h->did_reset) return fail(GYMHIP_ESTATE, "atari step before reset");
    const int64_t plane = (int64_t)h->cfg.dst_h * h->cfg.dst_w;
    h->head = (h->head + 1) % h->cfg.n_stack;   // ring advance == the reference's roll(-1) + write-last (stacked_observation.h:126-130, intended semantics)
    return launch_preprocess(h, frames, h->ring, plane * h->cfg.n_stack, plane * h->head, done, true, (hipStream_t)stream);
}

int gymhip_atari_materialize_nhwc(gymhip_atari* h, uint8_t* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    const int P = h->cfg.dst_h * h->cfg.dst_w;
    hipLaunchKernelGGL(materialize_nhwc_kernel, dim3(2048), dim3(kBlock), 0, (hipStream_t)stream, h->ring, out, h->cfg.n_envs, P, h->cfg.n_stack, h->head);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
}

int gymhip_atari_materialize_nchw_f32(gymhip_atari* h, float* out, void* stream) {
    if (int rc = check_atari(h)) return rc;
    if (!out) return fail(GYMHIP_EINVAL, "out is null");
    const int P = h->cfg.dst_h * h->cfg.dst_w;
    hipLaunchKernelGGL(materialize_nchw_f32_kernel, dim3(2048), dim3(kBlock), 0, (hipStream_t)stream, h->ring, out, h->cfg.n_envs, P, h->cfg.n_stack, h->head);
    HIP_TRY(hipGetLastError());
    return GYMHIP_OK;
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
