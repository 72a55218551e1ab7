// tests/emu/dev/rccl/rccl.h — TEST INFRASTRUCTURE (the emulated device): the three RCCL calls csrc/comm.cpp makes.
// A world of one is a copy.  A world of N PROCESSES (one per rank, as the product runs) meets in a shared-memory file named by the unique id:
// every rank writes its contribution into its slot, a generation barrier, every rank reads all slots, a second barrier before the slots are reused.
// It shows that comm.cpp and the host above it hand the collective the right buffers, counts and ranks for world > 1 (tests/test_multi.py on the emulated
// device); it says nothing about RCCL, xGMI or stream ordering — the real collective is covered on a multi-GPU box only.
#pragma once
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>

struct emu_nccl_shared {
    volatile unsigned arrived;      // ranks that reached the current barrier
    volatile unsigned generation;   // barriers completed
    char pad[56];
    // then `world` slots of EMU_NCCL_SLOT bytes
};
enum { EMU_NCCL_SLOT = 1 << 20 };
typedef struct emu_nccl_comm { int rank, world; emu_nccl_shared* sh; size_t bytes; char name[64]; }* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
static inline const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "success" : r == ncclSystemError ? "emulated device: shared-memory rendezvous failed" : "emulated device: invalid argument (contribution larger than 1 MB?)";
}
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/emu_nccl_%ld_%ld_%ld", (long)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return ncclSuccess;
}
static inline bool emu_nccl_barrier(emu_nccl_comm* c) {
    emu_nccl_shared* s = c->sh;
    const unsigned gen = __atomic_load_n(&s->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&s->arrived, 1, __ATOMIC_ACQ_REL) == (unsigned)c->world) {
        __atomic_store_n(&s->arrived, 0, __ATOMIC_RELEASE);
        __atomic_add_fetch(&s->generation, 1, __ATOMIC_ACQ_REL);
        return true;
    }
    for (long spins = 0; __atomic_load_n(&s->generation, __ATOMIC_ACQUIRE) == gen; ++spins) {
        if (spins > 600000) return false;   // ~ a minute: a rank that never arrives must not hang the test
        if (spins < 1000) sched_yield(); else usleep(100);
    }
    return true;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    emu_nccl_comm* c = new emu_nccl_comm{rank, world, nullptr, 0, {0}};
    if (world > 1) {
        id.internal[sizeof id.internal - 1] = 0;
        if (id.internal[0] != '/' || strlen(id.internal) >= sizeof c->name) { delete c; return ncclInvalidArgument; }
        strcpy(c->name, id.internal);
        c->bytes = sizeof(emu_nccl_shared) + (size_t)world * EMU_NCCL_SLOT;
        const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);   // a fresh object is zero-filled: counters start at 0 whoever comes first
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
        void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { delete c; return ncclSystemError; }
        c->sh = (emu_nccl_shared*)p;
        if (!emu_nccl_barrier(c)) { munmap(p, c->bytes); delete c; return ncclSystemError; }   // like RCCL: returns when every rank has joined
    }
    *out = c;
    return ncclSuccess;
}
static inline ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t) {
    const size_t bytes = count * 8;
    if (c->world == 1) { if (send != recv) memmove(recv, send, bytes); return ncclSuccess; }
    if (bytes > EMU_NCCL_SLOT) return ncclInvalidArgument;
    char* slots = (char*)c->sh + sizeof(emu_nccl_shared);
    memcpy(slots + (size_t)c->rank * EMU_NCCL_SLOT, send, bytes);
    if (!emu_nccl_barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) memcpy((char*)recv + (size_t)r * bytes, slots + (size_t)r * EMU_NCCL_SLOT, bytes);
    if (!emu_nccl_barrier(c)) return ncclSystemError;   // nobody overwrites a slot another rank is still reading
    return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (c->sh) { munmap((void*)c->sh, c->bytes); if (c->rank == 0) shm_unlink(c->name); }
    delete c;
    return ncclSuccess;
}
