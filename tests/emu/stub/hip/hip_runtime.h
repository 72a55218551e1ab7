/* tests/emu: the device headers include <hip/hip_runtime.h>; the lane harness compiles them for the HOST (tests/emu/README.md) */
