// The stand-alone per-env wrapper mirror (gym/wrappers.h) put through the probe the reference's own templates are put through
// (wrappers_probe.h; oracle/ref_driver/ref_wrappers_driver.cpp).   usage: wrappers_mirror_demo wrappers|csv|unroll <in> <out>
#include <cstring>

#include "gym/wrappers.h"
#include "wrappers_probe.h"

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: %s wrappers|csv|unroll <in> <out>\n", argv[0]); return 1; }
    if (!std::strcmp(argv[1], "wrappers")) return wrappers_probe::run_wrappers(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "csv")) return wrappers_probe::run_csv(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "unroll")) return wrappers_probe::run_unroll(argv[2], argv[3]);
    return 1;
}
