// knobs.h -- the only place of the library that looks at the environment.
// Product switches are read ONCE per context, at creation (api.hip: read_knobs).  The A/B switches that exist so that a measured
// claim of DESIGN.md can be repeated are compiled in only with -DMCA_MEASURE (make MEASURE=1); the default build never sees them.
#pragma once
#include <cstdlib>

namespace mca {

inline const char *env_str(const char *name) { return std::getenv(name); }
#ifdef MCA_MEASURE
inline const char *measure_env(const char *name) { return env_str(name); }
#else
inline const char *measure_env(const char *) { return nullptr; }
#endif

}  // namespace mca
