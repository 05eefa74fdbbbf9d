// TEST INFRASTRUCTURE ONLY: stands in for <hip/hip_runtime.h> when the device sources are
// compiled with g++ for the CPU emulator (tests/emu/build.py puts this directory first on
// the include path).  Never on the product's include path.
#ifndef PQP_EMU_FAKE_HIP_RUNTIME_H
#define PQP_EMU_FAKE_HIP_RUNTIME_H
// address-space qualifiers and the inlining policy are meaningless on the host
#define PQP_LDS
#define PQP_GLOBAL
#define PQP_CALL inline
// scalarisation is the identity on the host
#define __builtin_amdgcn_readfirstlane(x) (x)
#include "../../hip_emu.hpp"
#endif
