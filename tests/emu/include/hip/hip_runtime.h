// tests/emu: stand-in for <hip/hip_runtime.h> used ONLY by the CPU emulation build
// (tests/emu/build_emu.py).  See hip_emu.hpp.
#include "../../hip_emu.hpp"
