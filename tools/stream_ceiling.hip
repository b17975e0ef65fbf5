// stream_ceiling — what HBM bandwidth does a plain grid-stride streaming kernel reach on this GPU for a given
// read:write mix?  The step kernels are priced against the 8 TB/s spec peak; this tool measures the practical ceiling
// the same access style (16 B per lane, non-temporal, 256-thread blocks, capped grid) gets with no arithmetic at all.
//   hipcc -O3 --offload-arch=gfx950 -o stream_ceiling stream_ceiling.hip && ./stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// R read streams and W write streams of 16 B per lane per iteration; NT selects non-temporal access.
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t n, u32x4* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        u32x4 v = {1, 2, 3, 4};
#pragma unroll
        for (int r = 0; r < R; r++) {
            u32x4 t = NT ? __builtin_nontemporal_load(in + (int64_t)r * n + i) : in[(int64_t)r * n + i];
            v += t;
        }
#pragma unroll
        for (int w = 0; w < W; w++) {
            u32x4 t = v + (uint32_t)w;
            if (NT) __builtin_nontemporal_store(t, out + (int64_t)w * n + i); else out[(int64_t)w * n + i] = t;
        }
        if (W == 0) acc += v;
    }
    if (W == 0 && acc.x == 0x12345678u) *sink = acc;   // keeps the loads alive
}

// The lockstep CartPole step kernel's exact stream set with no arithmetic: per env 2 x 16 B in (obs, residuals) + 2 x 4 B in
// (action, counter); 2 x 16 B out + 3 x 4 B out (counter, reward, done).  WIDE = 0: every lane moves its own 4-B words
// (256 B per wave instruction).  WIDE = 1: a thread owns 4 envs per iteration and moves the 4-B streams as one 16-B
// vector for 4 consecutive envs (the memory side of an LDS-transposed layout; the lane shuffle itself is not modelled).
template <int WIDE>
__global__ __launch_bounds__(256) void pattern_kernel(const u32x4* __restrict__ big_in, u32x4* __restrict__ big_out,
                                                      const uint32_t* __restrict__ small_in, uint32_t* __restrict__ small_out, int64_t n) {
    if (WIDE == 0) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            u32x4 a = __builtin_nontemporal_load(big_in + i), b = __builtin_nontemporal_load(big_in + n + i);
            uint32_t c = __builtin_nontemporal_load(small_in + i), d = __builtin_nontemporal_load(small_in + n + i);
            a += c; b += d;
            __builtin_nontemporal_store(a, big_out + i);
            __builtin_nontemporal_store(b, big_out + n + i);
            __builtin_nontemporal_store(a.x, small_out + i);
            __builtin_nontemporal_store(b.y, small_out + n + i);
            __builtin_nontemporal_store(a.z ^ b.w, small_out + 2 * n + i);
        }
    } else if (WIDE == 2) {
        // block-contiguous: block b owns envs [b * chunk, (b + 1) * chunk) and walks them tile by tile
        const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            u32x4 a = __builtin_nontemporal_load(big_in + i), b = __builtin_nontemporal_load(big_in + n + i);
            uint32_t c = __builtin_nontemporal_load(small_in + i), d = __builtin_nontemporal_load(small_in + n + i);
            a += c; b += d;
            __builtin_nontemporal_store(a, big_out + i);
            __builtin_nontemporal_store(b, big_out + n + i);
            __builtin_nontemporal_store(a.x, small_out + i);
            __builtin_nontemporal_store(b.y, small_out + n + i);
            __builtin_nontemporal_store(a.z ^ b.w, small_out + 2 * n + i);
        }
    } else {
        for (int64_t base = (int64_t)blockIdx.x * 1024; base < n; base += (int64_t)gridDim.x * 1024) {
            u32x4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = __builtin_nontemporal_load(big_in + base + u * 256 + threadIdx.x);
                b[u] = __builtin_nontemporal_load(big_in + n + base + u * 256 + threadIdx.x);
            }
            const int64_t q = (base >> 2) + threadIdx.x;   // 16-B group of 4 consecutive envs
            u32x4 c = __builtin_nontemporal_load((const u32x4*)small_in + q), d = __builtin_nontemporal_load((const u32x4*)(small_in + n) + q);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] += c; b[u] += d;
                __builtin_nontemporal_store(a[u], big_out + base + u * 256 + threadIdx.x);
                __builtin_nontemporal_store(b[u], big_out + n + base + u * 256 + threadIdx.x);
            }
            __builtin_nontemporal_store(a[0] + a[1], (u32x4*)small_out + q);
            __builtin_nontemporal_store(b[2] + b[3], (u32x4*)(small_out + n) + q);
            __builtin_nontemporal_store(a[2] ^ b[1], (u32x4*)(small_out + 2 * n) + q);
        }
    }
}

template <int WIDE>
static void run_pattern(const char* label, u32x4* in, u32x4* out, int64_t n, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t* small_in = (const uint32_t*)(in + 2 * n);
    uint32_t* small_out = (uint32_t*)(out + 2 * n);
    for (int k = 0; k < 5; k++) pattern_kernel<WIDE><<<grid, 256>>>(in, out, small_in, small_out, n);
    const int reps = 40;
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) pattern_kernel<WIDE><<<grid, 256>>>(in, out, small_in, small_out, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-34s grid %5d  %8.4f ms  %7.1f GB/s\n", label, grid, ms, 84.0 * (double)n / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

template <int R, int W, bool NT>
static void run(const char* label, u32x4* in, u32x4* out, u32x4* sink, int64_t n, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 5; k++) stream_kernel<R, W, NT><<<grid, 256>>>(in, out, n, sink);
    const int reps = 40;
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) stream_kernel<R, W, NT><<<grid, 256>>>(in, out, n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double bytes = (double)(R + W) * 16.0 * (double)n;
    printf("%-34s grid %5d  %8.4f ms  %7.1f GB/s\n", label, grid, ms, bytes / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t n = (int64_t)1 << 24;            // 16,777,216 lanes x 16 B = 256 MiB per stream (the bench's env count)
    u32x4 *in, *out, *sink;
    CK(hipMalloc(&in, (size_t)n * 16 * 4));
    CK(hipMalloc(&out, (size_t)n * 16 * 6));
    CK(hipMalloc(&sink, 16));
    CK(hipMemset(in, 1, (size_t)n * 16 * 4));
    CK(hipMemset(out, 0, (size_t)n * 16 * 6));
    int grids[16] = {1024, 2048, 8192, 65536};
    int n_grids = 4;
    if (argc > 1) { n_grids = 0; for (int a = 1; a < argc && n_grids < 16; a++) grids[n_grids++] = atoi(argv[a]); }
    for (int gi = 0; gi < n_grids; gi++) {
        const int grid = grids[gi];
        run<1, 0, true>("read-only  nt", in, out, sink, n, grid);
        run<0, 1, true>("write-only nt", in, out, sink, n, grid);
        run<1, 1, true>("copy 1R:1W nt", in, out, sink, n, grid);
        run<1, 1, false>("copy 1R:1W plain", in, out, sink, n, grid);
        run<2, 3, true>("2R:3W nt (lockstep 40:60 mix)", in, out, sink, n, grid);
        run<2, 3, false>("2R:3W plain", in, out, sink, n, grid);
        run<4, 6, true>("4R:6W nt (10 streams)", in, out, sink, n, grid);
        run<2, 2, true>("2R:2W nt", in, out, sink, n, grid);
        run_pattern<0>("step pattern 84 B, 4-B lanes", in, out, n, grid);
        run_pattern<1>("step pattern 84 B, 16-B groups", in, out, n, grid);
        run_pattern<2>("step pattern 84 B, block-contiguous", in, out, n, grid);
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(sink));
    return 0;
}
