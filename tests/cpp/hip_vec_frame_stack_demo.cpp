// gym::HipVecFrameStack over a scripted image VecEnv, stand-alone mirror build: see frame_stack_probe.h.
// usage: hip_vec_frame_stack_demo <mode: plain|as_written|norm|norm_as_written|clear> <n_envs> <H> <W> <n_stack> <steps> <host|device> <out.bin>
#include "gym/hip_vec_frame_stack.h"
#include "frame_stack_probe.h"

int main(int argc, char** argv) {
    if (argc != 9) { std::fprintf(stderr, "usage: %s plain|as_written|norm|norm_as_written|clear n_envs H W n_stack steps host|device out.bin\n", argv[0]); return 2; }
    return frame_stack_probe::run(argv + 1);
}
