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
    for (int64_t skew : {(int64_t)0, (int64_t)256, (int64_t)512, (int64_t)1024, (int64_t)2048, (int64_t)4096, (int64_t)8192, (int64_t)16384, (int64_t)32768,
                         (int64_t)65536, (int64_t)(65536 + 4096), (int64_t)131072, (int64_t)262144, (int64_t)(262144 + 4096 + 256), (int64_t)1048576,
                         (int64_t)(1048576 + 65536 + 4096 + 256), (int64_t)2097152, (int64_t)(2097152 + 4096)}) {
        char* p[6]; int64_t off = 0;
        for (int k = 0; k < 6; k++) { p[k] = slab + off + k * skew; off += sizes[k]; }
        for (int grid : {2048, 4096}) {
            const double ms = time_ms([&] { mix_kernel<1, true, true><<<grid, 512>>>((f32x4*)p[0], (i32x4*)p[1], (const int32_t*)p[2], (uint32_t*)p[3], (float*)p[4], (float*)p[5], n); }, reps);
            std::printf("mix V=1 nt pipe  skew %8lld  grid %5d  %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", (long long)skew, grid, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
            std::fflush(stdout);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
