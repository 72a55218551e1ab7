// tests/emu/dev/rccl/rccl.h — TEST INFRASTRUCTURE (the emulated device): a one-rank communicator; the all-gather of a world of one is a copy.
// More than one rank is refused: the RCCL path is covered by tests/test_multi.py (gloo on CPU, RCCL on a multi-GPU box).
#pragma once
#include <hip/hip_runtime.h>
typedef struct emu_nccl_comm { int rank, world; }* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
static inline const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "emulated device: a world of one rank only"; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof *id); return ncclSuccess; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId, int rank) {
    if (world != 1 || rank != 0) return ncclInvalidArgument;
    *c = new emu_nccl_comm{0, 1};
    return ncclSuccess;
}
static inline ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t, hipStream_t) {
    if (send != recv) memmove(recv, send, count * 8);
    return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
