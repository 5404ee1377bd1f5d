// TEST INFRASTRUCTURE: when the kernel sources are compiled for the HOST emulator
// (tests/hipemu/build_emu.py puts this directory first on the include path) this header
// replaces <hip/hip_runtime.h>.
#pragma once
#include "../hipemu.h"
