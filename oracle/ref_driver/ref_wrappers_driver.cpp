// ref_wrappers_driver.cpp — OUR driver around the REFERENCE's own headers, compiled in place from /root/reference by
// oracle/Makefile (target _ref).  Nothing of the reference is copied: the #includes below resolve to its files where
// they lie.  Used only in the build container to pin oracle rows A4 (TimeLimit), A5 (ClipRewardEnv), A8 (Monitor) and
// A9 (RunningMeanStd); outputs are compared with oracle/gym_oracle.cpp by tests/test_ref_pin.py and frozen as
// tests/golden/ref_*.npz so the pin also travels to the GPU box.
//
// usage: ref_wrappers_driver wrappers <in> <out>   |   ref_wrappers_driver rms <in> <out>
//        ref_wrappers_driver csv <in> <out.csv>            the reference's Monitor with a ResultWriter file (monitor.h:20-35,44-50,93-94)
//        ref_wrappers_driver unroll <in> <out>             the reference's WrapperUnroll and ObservationWrapper2 (common/wrapper.h:32-81,231-275)
//        ref_wrappers_driver ckpt_save <in.bin> <out.pt>   the reference's torch::save(RunningMeanStd) after updates (vec_normalize.h:79-92)
//        ref_wrappers_driver ckpt_load <in.pt> <out.bin>   the reference's torch::load(RunningMeanStd) (vec_normalize.h:96-108)
#include "wrappers/time_limit.h"
#include "wrappers/clip_reward.h"
#include "wrappers/monitor.h"
#include "common/normalizers/running_mean_std.h"
#include <cassert>
#include "wrappers/vec_env/updateable.h"

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../tests/cpp/wrappers_probe.h"   // ScriptEnv, run_wrappers, run_csv: the same probe the stand-alone mirror is put through

namespace {
using wrappers_probe::rd; using wrappers_probe::run_wrappers; using wrappers_probe::run_csv; using wrappers_probe::run_unroll;

int run_rms(const char* in, const char* out) {
    FILE* f = fopen(in, "rb"); if (!f) return 2;
    int64_t nb, F;
    if (!rd(f, &nb, 1) || !rd(f, &F, 1)) return 2;
    RunningMeanStd rms(std::vector<int64_t>{F});
    FILE* o = fopen(out, "wb"); if (!o) return 2;
    for (int64_t b = 0; b < nb; b++) {
        int64_t n; if (!rd(f, &n, 1)) return 2;
        std::vector<float> x(n * F); if (!rd(f, x.data(), n * F)) return 2;
        auto t = torch::from_blob(x.data(), {n, F}, torch::kFloat32).clone();
        // batch moments exactly as RunningMeanStdImpl::update computes them, also written out so the merge can be checked alone
        auto bm = t.mean(0), bv = t.var(0, false);
        rms->update(t);
        auto mean = rms->mean<false>().contiguous(), var = rms->var<false>().contiguous(); float cnt = rms->count().item<float>();
        fwrite(bm.contiguous().data_ptr<float>(), 4, F, o); fwrite(bv.contiguous().data_ptr<float>(), 4, F, o);
        fwrite(mean.data_ptr<float>(), 4, F, o); fwrite(var.data_ptr<float>(), 4, F, o); fwrite(&cnt, 4, 1, o);
    }
    fclose(f); fclose(o);
    return 0;
}


// in: int64 F (0 = scalar shape {}, as the return statistics, vec_normalize.h:57), int64 n, float x[n][max(F,1)].
// out.pt: torch::save of the reference module after one update(x); <out.pt>.bin: mean[F'], var[F'], count as floats.
int run_ckpt_save(const char* in, const char* out) {
    FILE* f = fopen(in, "rb"); if (!f) return 2;
    int64_t F, n;
    if (!rd(f, &F, 1) || !rd(f, &n, 1)) return 2;
    const int64_t W = F > 0 ? F : 1;
    std::vector<float> x(n * W); if (!rd(f, x.data(), n * W)) return 2;
    fclose(f);
    RunningMeanStd rms(F > 0 ? std::vector<int64_t>{F} : std::vector<int64_t>{});
    auto t = torch::from_blob(x.data(), F > 0 ? std::vector<int64_t>{n, F} : std::vector<int64_t>{n}, torch::kFloat32).clone();
    rms->update(t);
    torch::save(rms, std::string(out));
    FILE* o = fopen((std::string(out) + ".bin").c_str(), "wb"); if (!o) return 2;
    auto mean = rms->mean<false>().contiguous().reshape({-1}), var = rms->var<false>().contiguous().reshape({-1}); float cnt = rms->count().item<float>();
    fwrite(mean.data_ptr<float>(), 4, W, o); fwrite(var.data_ptr<float>(), 4, W, o); fwrite(&cnt, 4, 1, o);
    fclose(o);
    return 0;
}

// out.bin: int64 numel, then mean[numel], var[numel], count as floats, read back through the reference's load().
int run_ckpt_load(const char* in, const char* out) {
    RunningMeanStd rms;                       // default-constructed: load() assigns the three tensors (running_mean_std.h:35-40)
    torch::load(rms, std::string(in));
    auto mean = rms->mean<false>().contiguous().reshape({-1}), var = rms->var<false>().contiguous().reshape({-1}); float cnt = rms->count().item<float>();
    if (mean.scalar_type() != torch::kFloat32 || var.scalar_type() != torch::kFloat32) return 3;
    FILE* o = fopen(out, "wb"); if (!o) return 2;
    int64_t numel = mean.numel();
    fwrite(&numel, 8, 1, o);
    fwrite(mean.data_ptr<float>(), 4, numel, o); fwrite(var.data_ptr<float>(), 4, numel, o); fwrite(&cnt, 4, 1, o);
    fclose(o);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s wrappers|rms|csv|unroll|ckpt_save|ckpt_load <in> <out>\n", argv[0]); return 1; }
    if (!strcmp(argv[1], "wrappers")) return run_wrappers(argv[2], argv[3]);
    if (!strcmp(argv[1], "rms")) return run_rms(argv[2], argv[3]);
    if (!strcmp(argv[1], "csv")) return run_csv(argv[2], argv[3]);
    if (!strcmp(argv[1], "unroll")) return run_unroll(argv[2], argv[3]);
    if (!strcmp(argv[1], "ckpt_save")) return run_ckpt_save(argv[2], argv[3]);
    if (!strcmp(argv[1], "ckpt_load")) return run_ckpt_load(argv[2], argv[3]);
    return 1;
}
