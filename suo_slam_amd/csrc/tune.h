// Runtime switches of libsuo_hip.so.
//   env_switch("SUO_X", default): a SUPPORTED switch, read from the environment in every build and listed in include/suo_hip.h -- it selects between forms whose
//     results the test suite holds against each other (matrix pipe, fused / separate launches, side streams).
//   SUO_TUNE("SUO_X", default): a TUNING knob -- a threshold or an A/B switch behind a measurement logged in DESIGN.md / profiles/REJECTED.md.  The product library
//     compiles it to its default and reads nothing; tools/build_variant.sh builds with -DSUO_TUNING, where the environment overrides the default.
#pragma once
#include <stdlib.h>

namespace suo {
inline long env_switch(const char* name, long dflt) {
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
}
inline bool env_set(const char* name) { return getenv(name) != nullptr; }
}  // namespace suo

#ifdef SUO_TUNING
#define SUO_TUNE(name, dflt) suo::env_switch(name, (long)(dflt))
#else
#define SUO_TUNE(name, dflt) ((long)(dflt))
#endif
