// build_id.hip -- the identity of the sources this libmjhip.so was built from (one translation unit of libmjhip.so, see host.hpp).
// mujoco_warp_amd/_abi.py hashes csrc/*, include/mjhip.h and the compiler flags into MJH_BUILD_ID when it builds, and compares the hash of
// the sources on disk with the one baked in here before it loads the library: a stale library is rebuilt or refused, never run.
#include "../../include/mjhip.h"

#ifndef MJH_BUILD_ID
#define MJH_BUILD_ID "unknown"
#endif
// (the marker in front lets the loader read the id out of the file without mapping it)
static const char kBuildId[] = "MJH_BUILD_ID=" MJH_BUILD_ID;
extern "C" __attribute__((visibility("default"))) const char* mjh_build_id(void) { return kBuildId + 13; }
