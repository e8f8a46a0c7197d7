// Experiment knobs read from the environment exist only in a library built with -DRSCM_EXPERIMENTS (`make experiments` ->
// librscm_gpu_experiments.so, what the sweep scripts under scripts/ load through RSCM_GPU_LIB): the shipped librscm_gpu.so's launch
// plans never depend on an undocumented environment variable.  In the shipped build every knob is its default, folded at compile
// time.  The environment variables the shipped library DOES read are documented in include/rscm_gpu.h ("Environment"):
// RSCM_SPLIT_RUNS and RSCM_POISON_ALLOC.
//
//   RSCM_SPLIT_CHUNK / RSCM_SPLIT_CHUNK2   model steps per launch of the two member blocks of a cut run   (rscm_gpu.cpp)
//   RSCM_SPLIT_FIRST                       members of the first block                                        (rscm_gpu.cpp)
//   RSCM_LOCKSTEP_SPLIT=0                  lock-step groups never split over two streams                      (lockstep.cpp)
//   RSCM_UDEB_VARIANT=0|2|3                one ClimateUDEB kernel shape for the whole process                 (udeb.hip)
#pragma once
#include <cstdlib>

namespace rscm {
#ifdef RSCM_EXPERIMENTS
inline long long experiment_env(const char* name, long long dflt)
{
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}
#else
constexpr long long experiment_env(const char*, long long dflt) { return dflt; }
#endif
}  // namespace rscm
