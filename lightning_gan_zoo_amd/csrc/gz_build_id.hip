// Identity of the sources this library was built from (lightning_gan_zoo_amd/build.py passes the digest).
// _lib.load() compares it with the digest of the tree it runs from: a stale libgz_hip.so is refused, not run.
#include <hip/hip_runtime.h>
#include "gz_ops.h"

#ifndef GZ_SOURCE_DIGEST
#define GZ_SOURCE_DIGEST unknown
#endif
#define GZ_STR2(x) #x
#define GZ_STR(x) GZ_STR2(x)

extern "C" const char* gz_source_digest(void) { return GZ_STR(GZ_SOURCE_DIGEST); }
