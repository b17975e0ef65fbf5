// bw_probe.hip — what HBM rate does the step kernel's ACCESS PATTERN allow, with the arithmetic taken out?
// Stand-alone (hipcc, no torch).  Streams of the CartPole lockstep step, per env: obs f32x4 read+write in place, residual
// i32x4 read+write in place, counter u32 read+write in place, action i32 read, reward f32 write, done f32 write = 84 B.
// main() sweeps the STAGGER between the stream bases (one slab, array k at its cumulative offset + k x skew) for the shipped
// shape — one env per lane per iteration, non-temporal, next iteration's loads issued before this one's stores.  The copy
// kernel and the V = 2 / 4 envs-per-lane forms are kept for ad-hoc runs (tools/stream_ceiling.hip holds the round-1 study
// of those).  Result (profiles/r02_sweeps.txt): 0.67-0.69 of 8 TB/s at every skew, 0.70 at 4 KiB: the pattern's ceiling.
// Build: hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bin/bw_probe      Run: tools/bin/bw_probe [n_envs]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

using f32x4 = float __attribute__((ext_vector_type(4)));
using i32x4 = int32_t __attribute__((ext_vector_type(4)));
template <int V> struct Vec;
template <> struct Vec<1> { using f = float; using i = int32_t; using u = uint32_t; };
template <> struct Vec<2> { using f = float __attribute__((ext_vector_type(2))); using i = int32_t __attribute__((ext_vector_type(2))); using u = uint32_t __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { using f = f32x4; using i = i32x4; using u = uint32_t __attribute__((ext_vector_type(4))); };

template <bool NT, typename T> __device__ inline T ld(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT, typename T> __device__ inline void st(T* p, T v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT> __global__ __launch_bounds__(512) void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 512;
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += stride) st<NT>(out + i, ld<NT>(in + i) + 1.0f);
}

// PIPE: the next iteration's loads are issued before this iteration's stores (as the shipped kernel does)
template <int V, bool NT, bool PIPE> __global__ __launch_bounds__(512) void mix_kernel(f32x4* obs, i32x4* res, const int32_t* __restrict__ act, uint32_t* ctr,
                                                                                      float* rew, float* done, int64_t n) {
    using FV = typename Vec<V>::f; using IV = typename Vec<V>::i; using UV = typename Vec<V>::u;
    const int64_t groups = n / V, stride = (int64_t)gridDim.x * 512;
    struct In { f32x4 o[V]; i32x4 k[V]; IV a; UV w; };
    auto load = [&](int64_t g, In& in) {
#pragma unroll
        for (int j = 0; j < V; j++) { in.o[j] = ld<NT>(obs + g * V + j); in.k[j] = ld<NT>(res + g * V + j); }
        in.a = ld<NT>((const IV*)act + g); in.w = ld<NT>((const UV*)ctr + g);
    };
    int64_t g = (int64_t)blockIdx.x * 512 + threadIdx.x;
    In cur, nxt;
    if (g < groups) load(g, cur);
    for (; g < groups; g += stride) {
        if constexpr (PIPE) { if (g + stride < groups) load(g + stride, nxt); }
        else if (g != (int64_t)blockIdx.x * 512 + threadIdx.x) load(g, cur);
#pragma unroll
        for (int j = 0; j < V; j++) { st<NT>(obs + g * V + j, cur.o[j] * 1.0001f); st<NT>(res + g * V + j, cur.k[j] + 1); }
        UV w = cur.w + 1u;
        FV r, d;
        if constexpr (V == 1) { r = (float)cur.a; d = (float)(w & 1u); }
        else {
#pragma unroll
            for (int j = 0; j < V; j++) { r[j] = (float)cur.a[j]; d[j] = (float)(w[j] & 1u); }
        }
        st<NT>((UV*)ctr + g, w); st<NT>((FV*)rew + g, r); st<NT>((FV*)done + g, d);
        if constexpr (PIPE) cur = nxt;
    }
}


// Which streams cost what: MASK bit 0 = obs rw, 1 = residual rw, 2 = counter rw, 3 = action r, 4 = reward w, 5 = done w.
// MERGED: residual and counter live in ONE array of 64-env tiles [64 x 16 B residual | 64 x 4 B counter] (1280 B per tile).
template <int MASK, bool MERGED, bool KEEP_LANES = false> __global__ __launch_bounds__(512) void subset_kernel(f32x4* obs, i32x4* res, const int32_t* __restrict__ act, uint32_t* ctr,
                                                                                        float* rew, float* done, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 512;
    struct In { f32x4 o; i32x4 k; int32_t a; uint32_t w; };
    auto res_ptr = [&](int64_t i) { return MERGED ? (i32x4*)((char*)res + (i >> 6) * 1280 + (i & 63) * 16) : res + i; };
    auto ctr_ptr = [&](int64_t i) { return MERGED ? (uint32_t*)((char*)res + (i >> 6) * 1280 + 1024 + (i & 63) * 4) : ctr + i; };
    auto load = [&](int64_t i, In& in) {
        if (MASK & 1) in.o = ld<true>(obs + i);
        if (MASK & 2) in.k = ld<true>(res_ptr(i));
        if (MASK & 8) in.a = ld<true>(act + i);
        if (MASK & 4) in.w = ld<!KEEP_LANES>(ctr_ptr(i));
    };
    int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x;
    In cur{}, nxt{};
    if (i < n) load(i, cur);
    for (; i < n; i += stride) {
        if (i + stride < n) load(i + stride, nxt);
        if (MASK & 1) st<true>(obs + i, cur.o * 1.0001f);
        if (MASK & 2) st<true>(res_ptr(i), cur.k + 1);
        const uint32_t w = cur.w + 1u;
        if (MASK & 4) st<!KEEP_LANES>(ctr_ptr(i), w);
        if (MASK & 16) st<!KEEP_LANES>(rew + i, (float)cur.a + cur.o.x);
        if (MASK & 32) st<!KEEP_LANES>(done + i, (float)(w & 1u));
        cur = nxt;
    }
}

// The wide rows out of place: read obs / residual from one pair of arrays, write the other pair (a ping-pong state), + the action read.
__global__ __launch_bounds__(512) void pingpong_kernel(const f32x4* __restrict__ obs_in, const i32x4* __restrict__ res_in, const int32_t* __restrict__ act,
                                                       f32x4* __restrict__ obs_out, i32x4* __restrict__ res_out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 512;
    int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x;
    f32x4 o{}, on{}; i32x4 k{}, kn{}; int32_t a = 0, an = 0;
    if (i < n) { o = ld<true>(obs_in + i); k = ld<true>(res_in + i); a = ld<true>(act + i); }
    for (; i < n; i += stride) {
        if (i + stride < n) { on = ld<true>(obs_in + i + stride); kn = ld<true>(res_in + i + stride); an = ld<true>(act + i + stride); }
        st<true>(obs_out + i, o * 1.0001f + (float)a); st<true>(res_out + i, k + 1);
        o = on; k = kn; a = an;
    }
}

template <typename F> static double time_ms(F&& launch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : (int64_t)1 << 24;
    if (n <= 0 || n % 2048) { std::fprintf(stderr, "n_envs must be a positive multiple of 2048\n"); return 2; }
    // ONE slab; array k starts at its cumulative offset + k * skew bytes.  With skew 0 and a power-of-two env count every
    // array base is a multiple of a large power of two, so at any moment all streams address the same DRAM channel / bank
    // bits; a skew staggers them.
    const int64_t sizes[6] = {n * 16, n * 16, n * 4, n * 4, n * 4, n * 4};   // obs, residual, action, counter, reward, done
    const int64_t max_skew = 1 << 22;
    int64_t total = 6 * max_skew;
    for (int64_t sz : sizes) total += sz;
    char* slab; CK(hipMalloc(&slab, total)); CK(hipMemset(slab, 0, total));
    const double bytes = 84.0 * n;
    const int reps = 40;
    std::printf("n_envs %lld, slab %p\n", (long long)n, (void*)slab);
    const bool full_sweep = argc > 2;
    std::vector<int64_t> skews = {0, 4096};
    if (full_sweep) skews = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 65536 + 4096, 131072, 262144, 262144 + 4096 + 256, 1048576,
                             1048576 + 65536 + 4096 + 256, 2097152, 2097152 + 4096};
    for (int64_t skew : skews) {
        char* p[6]; int64_t off = 0;
        for (int k = 0; k < 6; k++) { p[k] = slab + off + k * skew; off += sizes[k]; }
        for (int grid : {2048, 4096}) {
            const double ms = time_ms([&] { mix_kernel<1, true, true><<<grid, 512>>>((f32x4*)p[0], (i32x4*)p[1], (const int32_t*)p[2], (uint32_t*)p[3], (float*)p[4], (float*)p[5], n); }, reps);
            std::printf("mix V=1 nt pipe  skew %8lld  grid %5d  %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", (long long)skew, grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
            std::fflush(stdout);
        }
    }

    {
        char* p[6]; int64_t off = 0;
        for (int k = 0; k < 6; k++) { p[k] = slab + off; off += sizes[k]; }
        char* merged = slab + off;   // n * 20 B fits in the 6 x 4 MiB of skew head-room only for small n: take obs+residual space of a second slab instead
        char* slab2; CK(hipMalloc(&slab2, n * 20 + 4096)); CK(hipMemset(slab2, 0, n * 20 + 4096));
        merged = slab2;
        auto run = [&](const char* name, double bpe, auto kernel, char* resbase) {
            for (int grid : {2048, 4096}) {
                const double ms = time_ms([&] { kernel<<<grid, 512>>>((f32x4*)p[0], (i32x4*)resbase, (const int32_t*)p[2], (uint32_t*)p[3], (float*)p[4], (float*)p[5], n); }, reps);
                std::printf("%-44s grid %5d  %.4f ms  %5.0f B/env  %.0f GB/s  %.3f of 8 TB/s\n", name, grid, ms, bpe, bpe * n / ms / 1e6, bpe * n / ms / 1e6 / 8000.0);
                std::fflush(stdout);
            }
        };
        run("all six streams (84 B)", 84, subset_kernel<63, false>, p[1]);
        run("residual + counter merged in 64-env tiles", 84, subset_kernel<63, true>, merged);
        run("no reward / done writes", 76, subset_kernel<15, false>, p[1]);
        run("no reward / done / action", 72, subset_kernel<7, false>, p[1]);
        run("obs + residual only (16-B streams)", 64, subset_kernel<3, false>, p[1]);
        run("the 4-B streams only", 20, subset_kernel<60, false>, p[1]);
        run("obs only", 32, subset_kernel<1, false>, p[1]);
        run("obs + residual in place + action read (68 B): what still streams under the residency plan", 68, subset_kernel<11, false>, p[1]);
        {
            char* alt; CK(hipMalloc(&alt, n * 32)); CK(hipMemset(alt, 0, n * 32));
            for (int grid : {1024, 2048, 4096, 16384}) {
                bool flip = false;
                const double ms = time_ms([&] {
                    if (flip) pingpong_kernel<<<grid, 512>>>((const f32x4*)alt, (const i32x4*)(alt + n * 16), (const int32_t*)p[2], (f32x4*)p[0], (i32x4*)p[1], n);
                    else pingpong_kernel<<<grid, 512>>>((const f32x4*)p[0], (const i32x4*)p[1], (const int32_t*)p[2], (f32x4*)alt, (i32x4*)(alt + n * 16), n);
                    flip = !flip; }, reps);
                std::printf("%-44s grid %5d  %.4f ms  %5.0f B/env  %.0f GB/s  %.3f of 8 TB/s\n", "obs + residual OUT OF PLACE (ping-pong) + action", grid, ms, 68.0, 68.0 * n / ms / 1e6, 68.0 * n / ms / 1e6 / 8000.0);
            }
            for (int grid : {2048, 4096, 8192, 16384}) {   // the residency plan without arithmetic: all six streams, the 4-B lanes with the default cache policy
                const double ms = time_ms([&] { subset_kernel<63, false, true><<<grid, 512>>>((f32x4*)p[0], (i32x4*)p[1], (const int32_t*)p[2], (uint32_t*)p[3], (float*)p[4], (float*)p[5], n); }, reps);
                std::printf("%-44s grid %5d  %.4f ms  %5.0f B/env  %.0f GB/s  %.3f of 8 TB/s\n", "all six streams, lanes kept in the cache", grid, ms, 84.0, 84.0 * n / ms / 1e6, 84.0 * n / ms / 1e6 / 8000.0);
            }
            for (int grid : {1024, 16384}) {
                const double ms = time_ms([&] { subset_kernel<11, false><<<grid, 512>>>((f32x4*)p[0], (i32x4*)p[1], (const int32_t*)p[2], (uint32_t*)p[3], (float*)p[4], (float*)p[5], n); }, reps);
                std::printf("%-44s grid %5d  %.4f ms  %5.0f B/env  %.0f GB/s  %.3f of 8 TB/s\n", "obs + residual in place + action", grid, ms, 68.0, 68.0 * n / ms / 1e6, 68.0 * n / ms / 1e6 / 8000.0);
            }
            CK(hipFree(alt));
        }
        CK(hipFree(slab2));
    }
    CK(hipDeviceSynchronize());
    return 0;
}
