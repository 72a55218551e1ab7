// tests/emu/dev — TEST INFRASTRUCTURE: see hip_runtime.h beside this file
#pragma once
#include "hip_runtime.h"
