// comm.cpp — the path's only collective, behind the C ABI: one all-gather of the 4-element input commitments of every instance
// (`input_commitment`, /root/reference/src/ram_permutation/mod.rs:203-209; main_vm: src/main_vm/mod.rs closed-form commitment)
// over RCCL / xGMI.  Circuit instances are sharded across GPUs with no data-path exchange; the payload is
// 32 B per instance, so the collective is latency-bound: one ncclAllGather of batch * n_public u64 per step.
// One process per GPU: rank 0 draws the unique id (zk_comm_unique_id), the host's launcher hands it to the other ranks (a file, an
// environment variable, its own control plane — not this library's business), every rank calls zk_comm_create.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include "../../include/zkgl.h"
#include "cs.hpp"

struct zk_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint64_t* d_local = nullptr;  // packed public inputs of this rank's batch
    size_t local_words = 0;
};

namespace zkgl { void set_last_error(const std::string& m); zkgl::CS* cs_of(zk_cs* h); int initialized_device(); }

static int fail(const std::string& m, int code = ZK_ERR_HIP) { zkgl::set_last_error(m); return code; }
// the communicator binds to the CURRENT HIP device: it must be the one zk_init bound this process to
static int check_device() {
    const int want = zkgl::initialized_device();
    if (want < 0) return fail("zk_init() has not succeeded: no GPU context (there is no CPU fallback)");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail("hipGetDevice");
    if (cur != want) return fail("the calling thread's current device is not the one zk_init bound this process to", ZK_ERR_INVALID);
    return ZK_OK;
}

extern "C" {

int zk_comm_unique_id(uint8_t id[ZK_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= ZK_COMM_ID_BYTES, "ZK_COMM_ID_BYTES too small for ncclUniqueId");
    if (!id) return ZK_ERR_INVALID;
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memset(id, 0, ZK_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return ZK_OK;
}

int zk_comm_create(zk_comm** out, const uint8_t id[ZK_COMM_ID_BYTES], int rank, int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail("zk_comm_create: bad arguments", ZK_ERR_INVALID);
    if (int rc = check_device()) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    zk_comm* c = new zk_comm;
    c->rank = rank; c->world = world;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);  // binds to the current HIP device (zk_init)
    if (r != ncclSuccess) { delete c; return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    *out = c;
    return ZK_OK;
}

int zk_comm_destroy(zk_comm* c) {
    if (!c) return ZK_OK;
    if (c->d_local) hipFree(c->d_local);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return ZK_OK;
}

int zk_cs_gather_commitments(zk_cs* h, zk_comm* c, uint64_t* dev_out, uint32_t* n_public, void* stream) {
    if (!h || !c || !dev_out) return fail("zk_cs_gather_commitments: null argument", ZK_ERR_INVALID);
    if (int rc = check_device()) return rc;
    try {
        zkgl::CS* cs = zkgl::cs_of(h);
        if (cs->batch() == 0) return fail("zk_cs_gather_commitments before set_batch", ZK_ERR_INVALID);
        const size_t n_pub = cs->public_cells().size();
        if (n_pub == 0) return fail("zk_cs_gather_commitments: the circuit has no public inputs", ZK_ERR_INVALID);
        if (n_pub > 8) return fail("more than 8 public inputs per instance", ZK_ERR_CAPACITY);   // before anything is written: the staging buffer holds 8
        const size_t words = (size_t)cs->batch() * 8;  // room for up to 8 public inputs per instance
        if (c->local_words < words) {
            if (c->d_local) hipFree(c->d_local);
            if (hipMalloc((void**)&c->d_local, std::max<size_t>(words * 8, 8)) != hipSuccess) return fail("hipMalloc commitments");
            c->local_words = words;
        }
        const uint32_t n = cs->pack_public_inputs(c->d_local, stream);
        if (n_public) *n_public = n;
        const size_t count = (size_t)cs->batch() * n;
        // every rank holds the same batch size (instances are dealt round-robin and padded by the host): dev_out[rank][instance][k]
        ncclResult_t r = ncclAllGather(c->d_local, dev_out, count, ncclUint64, c->comm, (hipStream_t)stream);
        if (r != ncclSuccess) return fail(std::string("ncclAllGather: ") + ncclGetErrorString(r));
        return ZK_OK;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

}  // extern "C"
