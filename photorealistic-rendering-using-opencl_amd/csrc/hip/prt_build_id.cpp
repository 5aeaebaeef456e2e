// prt_build_id.cpp -- what this libprt.so was built from (include/prt.h prt_build_id).  build.py hashes the content of every source,
// header and compiler flag of the library into PRT_BUILD_ID and recompiles this file whenever that hash changes; tools/build_variant.sh
// prefixes the variant's name.  bench.py prints it and refuses to time a library whose id is not the working tree's; the tests assert
// the same: a library left behind by an experiment (round 3, twice) can no longer pass for the product.
#include "prt.h"

#ifndef PRT_BUILD_ID
#define PRT_BUILD_ID "unstamped"
#endif

extern "C" const char* prt_build_id(void) { return PRT_BUILD_ID; }
