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
    const int grids[] = {1024, 2048, 8192, 65536};
    for (int grid : grids) {
        run<1, 0, true>("read-only  nt", in, out, sink, n, grid);
        run<0, 1, true>("write-only nt", in, out, sink, n, grid);
        run<1, 1, true>("copy 1R:1W nt", in, out, sink, n, grid);
        run<1, 1, false>("copy 1R:1W plain", in, out, sink, n, grid);
        run<2, 3, true>("2R:3W nt (lockstep 40:60 mix)", in, out, sink, n, grid);
        run<2, 3, false>("2R:3W plain", in, out, sink, n, grid);
        run<4, 6, true>("4R:6W nt (10 streams)", in, out, sink, n, grid);
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(sink));
    return 0;
}
