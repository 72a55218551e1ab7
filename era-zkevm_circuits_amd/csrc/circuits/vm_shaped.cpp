// circuits/vm_shaped.cpp — placeholder until the main_vm-shaped cycle is recorded (see below).
#include "../gadgets.hpp"
namespace zkgl {
void vm_shaped_configure(CS&) { throw ZkError(ZK_ERR_INVALID, "vm_shaped: not built yet"); }
void vm_shaped_entry_point(CS&, uint32_t) { throw ZkError(ZK_ERR_INVALID, "vm_shaped: not built yet"); }
}
