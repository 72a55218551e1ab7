/* Test-only circuits (libzkgl_testcircuits.so, built by build.sh beside libzkgl.so; NOT part of the product ABI in include/). */
#ifndef ZKGL_TESTCIRCUITS_H
#define ZKGL_TESTCIRCUITS_H
#include "../../../include/zkgl.h"
#ifdef __cplusplus
extern "C" {
#endif
/* main_vm-shaped synthetic cycle (SURVEY.md §8d C2 of round 1; geometry src/main_vm/cycle.rs:959-966) */
int zk_test_circuit_vm_shaped_configure(zk_cs *cs);
int zk_test_circuit_vm_shaped(zk_cs *cs, uint32_t limit);
#ifdef __cplusplus
}
#endif
#endif
