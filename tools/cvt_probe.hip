// Probe (timing-free): does v_cvt_pk_u8_f32 equal cv::saturate_cast<uchar>(float) — round half to even, clamp to [0, 255] — for every
// input the resampler can produce?  Sweeps all multiples of 1/64 in [-8, 300], the floats next to every half-integer, and specials.
// build: hipcc -O2 --offload-arch=gfx950 tools/cvt_probe.hip -o tools/bin/cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, int n, uint32_t* hw, uint32_t* sw) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hw[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, 0u);
    int iv = (int)rintf(x[i]);
    iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
    sw[i] = (uint32_t)iv;
}
int main() {
    std::vector<float> x;
    for (int m = -8 * 64; m <= 300 * 64; m++) x.push_back(m / 64.0f);
    for (int h = 0; h < 260; h++) { float c = h + 0.5f; x.push_back(std::nextafterf(c, 0.f)); x.push_back(c); x.push_back(std::nextafterf(c, 1e9f)); }
    for (float s : {0.f, -0.f, 1e-30f, -1e-30f, 254.99998f, 255.49998f, 255.5f, 256.f, 1e9f, -1e9f, 3.4e38f}) x.push_back(s);
    int n = (int)x.size();
    float* dx; uint32_t *dh, *ds;
    hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, n, dh, ds);
    std::vector<uint32_t> h(n), s(n);
    hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (h[i] != s[i]) { if (bad < 12) std::printf("x=%.9g hw=%u sw=%u\n", x[i], h[i], s[i]); bad++; }
    std::printf("cvt_pk_u8_f32 vs rint+clamp: %d inputs, %d differ\n", n, bad);
    return bad != 0;
}
