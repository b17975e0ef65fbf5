// gather_probe.hip — what does ONE scattered 16-B reset-row gather cost at the L2's memory side, and how does FETCH_SIZE count it?
// The guide's gfx950 correction (double FETCH_SIZE) is calibrated on wide coalesced streaming reads only; the reset gather of the
// tape / MT19937 engines is a 16-B load by ~4.5 % of the lanes, each in a line of its own.  This probe issues exactly that, with a
// known row count, in the three layouts in question, so that `rocprofv3 --pmc FETCH_SIZE` gives bytes-per-row for each:
//   dense   every lane loads its 16-B row, coalesced                      (calibration: FETCH_SIZE x 2 == bytes)
//   tape    lane i of the 1-in-22 subset loads row [e][i], e random       (the [E][N][4] reset tape)
//   mt      the same subset loads 16 B at a random row of env i's 2,496-B block ([N][624] generator states)
//   wave    the same subset, rows of one wave adjacent per slot: [N/64][E][64][4]   (the layout VERDICT r2 item 3 suggests)
// Stand-alone (hipcc, no torch).  Build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/bin/gather_probe
// Run: tools/bin/gather_probe <dense|tape|mt|wave> [n_envs] [reps]      (prints rows gathered per launch)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// mode 0 dense, 1 tape [E][N], 2 mt [N][156 rows], 3 wave-blocked [N/64][E][64]
__global__ __launch_bounds__(512) void gather_kernel(const u32x4* __restrict__ table, int64_t n, int mode, int E, uint64_t salt, uint32_t* out, unsigned long long* rows) {
    uint32_t acc = 0;
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 512) {
        const uint64_t h = mix(salt + (uint64_t)i);
        const bool take = mode == 0 || (h % 22u) == 0;         // mean CartPole episode under a random policy: ~22 steps
        if (!take) continue;
        const uint32_t e = (uint32_t)((h >> 32) % (uint32_t)E);
        int64_t idx;
        if (mode == 0) idx = i;
        else if (mode == 1) idx = (int64_t)e * n + i;
        else if (mode == 2) idx = i * 156 + (e % 156u);
        else idx = ((i >> 6) * E + e) * 64 + (i & 63);
        const u32x4 v = table[idx];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        mine++;
    }
    if (acc == 0x12345u) out[0] = acc;                          // keeps the loads alive
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(rows, mine);
}

int main(int argc, char** argv) {
    const char* m = argc > 1 ? argv[1] : "tape";
    const int64_t n = argc > 2 ? atoll(argv[2]) : (int64_t)1 << 24;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const int mode = !strcmp(m, "dense") ? 0 : !strcmp(m, "tape") ? 1 : !strcmp(m, "mt") ? 2 : 3;
    const int E = mode == 2 ? 156 : 32;                         // tape: 32 rows x 256 MiB = 8 GiB at 16.7 M envs; mt: 39 GiB
    const size_t rows_total = mode == 0 ? (size_t)n : (size_t)n * E;
    u32x4* table; uint32_t* out; unsigned long long* rows;
    CK(hipMalloc(&table, rows_total * 16)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&rows, 8));
    CK(hipMemset(table, 1, rows_total * 16));
    for (int r = 0; r < reps; r++) {
        CK(hipMemset(rows, 0, 8));
        hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(512), 0, 0, table, n, mode, E, (uint64_t)r * 7919u, out, rows);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h = 0;
    CK(hipMemcpy(&h, rows, 8, hipMemcpyDeviceToHost));
    std::printf("mode %s n %lld table_bytes %zu rows_per_launch %llu bytes_per_launch %llu\n", m, (long long)n, rows_total * 16, h, h * 16ull);
    return 0;
}
