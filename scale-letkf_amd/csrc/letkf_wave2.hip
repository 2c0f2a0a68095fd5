// letkf_wave2.hip -- the two-wave instantiations (63 <= k <= 100) of the wave kernel as a compilation unit of their own:
// same source (letkf_wave.hip), LETKF_WAVE_UNIT2 selects launch_wave_kernel_two and leaves the host helpers to unit 1.
// The Makefile compiles this unit with -mllvm -amdgpu-sched-strategy=max-memory-clause (see letkf_wave.hip, dispatch).
#define LETKF_WAVE_UNIT2 1
#include "letkf_wave.hip"
