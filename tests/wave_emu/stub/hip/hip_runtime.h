// test-only stand-in for <hip/hip_runtime.h>: the lane-level emulator (tests/wave_emu/emu_hip.h)
#pragma once
#include "../../emu_hip.h"
