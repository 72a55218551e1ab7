// libzkgl_testcircuits.so — circuits that exist for the TESTS only (the main_vm-shaped synthetic cycle of round 1: fixtures of the copy-
// permutation / lookup-argument / device-program tests).  Not part of the product library: built beside it, linked against it, loaded by
// tests through zkgl.testlib().
#include <string>
#include "zkgl_testcircuits.h"
#include "../cs.hpp"

namespace zkgl {
CS* cs_of(zk_cs* h);
void set_last_error(const std::string& m);
void vm_shaped_configure(CS& cs);
void vm_shaped_entry_point(CS& cs, uint32_t limit);
}  // namespace zkgl

namespace {
template <class F>
int guarded(zk_cs* cs, F&& f) {
    if (!cs) { zkgl::set_last_error("null argument: cs"); return (int)ZK_ERR_INVALID; }
    try { f(*zkgl::cs_of(cs)); return ZK_OK; }
    catch (const zkgl::ZkError& e) { zkgl::set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { zkgl::set_last_error(e.what()); return (int)ZK_ERR_INVALID; }
}
}  // namespace

extern "C" int zk_test_circuit_vm_shaped_configure(zk_cs* cs) { return guarded(cs, [](zkgl::CS& c) { zkgl::vm_shaped_configure(c); }); }
extern "C" int zk_test_circuit_vm_shaped(zk_cs* cs, uint32_t limit) { return guarded(cs, [&](zkgl::CS& c) { zkgl::vm_shaped_entry_point(c, limit); }); }
