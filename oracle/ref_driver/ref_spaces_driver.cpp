// ref_spaces_driver.cpp — tests/cpp/spaces_probe.h run against the REFERENCE's own space types and Env::seed: its
// spaces/{space,box,discrete,util}.h and env.h included where they lie, spaces/{space,discrete,util}.cpp compiled in place by
// oracle/Makefile (target _ref).  No stand-in headers: these files need only the C++ standard library and c10/core/ScalarType.h
// from the image's own libtorch.  Output frozen as tests/golden/ref_spaces.json (tests/test_ref_pin.py).
#include "env.h"
#include "spaces/box.h"
#include "spaces/discrete.h"

#include "../../tests/cpp/spaces_probe.h"

int main(int argc, char** argv) {
    const std::string s = spaces_probe::run();
    if (argc > 1) { FILE* f = fopen(argv[1], "w"); if (!f) return 2; fputs(s.c_str(), f); fclose(f); }
    else fputs(s.c_str(), stdout);
    return 0;
}
