// h2_tune.hpp -- knobs of the tuning sweeps (tools/sweep_*.sh; results in DESIGN.md section 4).  They exist only in a
// build made with -DH2_TUNING (H2_BUILD_TUNING=1 python -m halo2_prover_amd.build --force): the product library never
// reads its configuration from the environment.
#pragma once
#include <cstdlib>

namespace h2 {
inline int tune_int(const char* name, int fallback) {
#ifdef H2_TUNING
  if (const char* v = getenv(name)) return atoi(v);
#else
  (void)name;
#endif
  return fallback;
}
}  // namespace h2
