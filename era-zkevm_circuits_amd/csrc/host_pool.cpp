// host_pool.cpp — the host side of a step is per-instance work with no shared state (every zk_pack_* entry writes one instance's words of
// the batch staging arrays): the reference resolves its witness closures on a worker pool (/root/reference/src/ram_permutation/mod.rs:553-556,
// closures `Send + Sync`, src/base_structures/memory_query/mod.rs:236).  zk_parallel_for is that pool behind the C ABI: a caller (Rust, C,
// ctypes) hands it a plain function that packs instance `job`; zk_pack_main_vm_witness_batch is the array form for the headline circuit.
#include <atomic>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/zkgl.h"
#include "../../include/zkgl_vm.h"
#include "../../include/zkgl_witness.h"

namespace zkgl { void set_last_error(const std::string& m); }

namespace {
template <class F>
int run_jobs(uint32_t n_jobs, uint32_t n_threads, uint32_t* first_failed, F&& job) {
    if (first_failed) *first_failed = UINT32_MAX;
    if (n_jobs == 0) return ZK_OK;
    uint32_t hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (n_threads == 0) n_threads = hw;
    n_threads = std::min(n_threads, n_jobs);
    std::atomic<uint32_t> next{0};
    std::mutex mu;
    int rc_first = ZK_OK;
    uint32_t job_first = UINT32_MAX;
    std::string err_first;
    auto worker = [&]() noexcept {
        for (;;) {
            const uint32_t j = next.fetch_add(1, std::memory_order_relaxed);
            if (j >= n_jobs) return;
            int rc;
            try { rc = job(j); }   // nothing may leave a worker thread (std::terminate) or, on the caller's thread, the extern "C" entry
            catch (const std::exception& e) { zkgl::set_last_error(e.what()); rc = (int)ZK_ERR_INVALID; }
            catch (...) { zkgl::set_last_error("job threw"); rc = (int)ZK_ERR_INVALID; }
            if (rc != ZK_OK) {   // the lowest failing job wins, whatever the interleaving; the other jobs still run (they are independent)
                std::lock_guard<std::mutex> g(mu);
                if (j < job_first) { job_first = j; rc_first = rc; err_first = zk_last_error(); }   // zk_last_error is per thread: take it here
            }
        }
    };
    if (n_threads == 1) worker();
    else {
        std::vector<std::thread> ts;
        ts.reserve(n_threads - 1);
        // thread creation can fail (std::system_error: thread limit, cgroup pids): carry on with the threads that did start — the caller's
        // thread works too, every job still runs — and join them; nothing crosses the C ABI as an exception
        try {
            for (uint32_t t = 1; t < n_threads; ++t) ts.emplace_back(worker);
        } catch (...) {}
        worker();
        for (auto& t : ts) t.join();
    }
    if (rc_first != ZK_OK) {
        zkgl::set_last_error("job " + std::to_string(job_first) + ": " + err_first);
        if (first_failed) *first_failed = job_first;
    }
    return rc_first;
}
}  // namespace

extern "C" int zk_parallel_for(uint32_t n_jobs, uint32_t n_threads, zk_job_fn fn, void* ctx, uint32_t* first_failed_job) {
    if (!fn) { zkgl::set_last_error("zk_parallel_for: null job function"); return (int)ZK_ERR_INVALID; }
    return run_jobs(n_jobs, n_threads, first_failed_job, [&](uint32_t j) { return fn(ctx, j); });
}

extern "C" int zk_host_threads(void) {
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? (int)hw : 1;
}

extern "C" int zk_pack_main_vm_witness_batch(zk_cs* cs, uint32_t n_instances, const zk_vm_closed_form_input* inputs, const zk_vm_witness_oracle* oracles,
                                             zk_vm_queue_states* states, uint32_t first_instance, uint32_t batch, uint64_t* outer_words, uint64_t* loop_words,
                                             uint32_t flags, zk_vm_pack_report* reports, uint32_t n_threads) {
    if (!cs || !inputs || !oracles || !outer_words || !loop_words || !reports) { zkgl::set_last_error("zk_pack_main_vm_witness_batch: null argument"); return (int)ZK_ERR_INVALID; }
    if ((uint64_t)first_instance + n_instances > batch) { zkgl::set_last_error("zk_pack_main_vm_witness_batch: first_instance + n_instances > batch"); return (int)ZK_ERR_INVALID; }
    return run_jobs(n_instances, n_threads, nullptr, [&](uint32_t j) {
        return zk_pack_main_vm_witness_states(cs, inputs + j, oracles + j, states ? states + j : nullptr, first_instance + j, batch, outer_words, loop_words, flags, reports + j);
    });
}
