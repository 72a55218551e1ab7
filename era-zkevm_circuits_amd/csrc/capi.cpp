// capi.cpp — the extern "C" boundary declared in include/zkgl.h.  No torch types, no
// exceptions across the boundary, no abort: every failure becomes a negative zk_status plus a
// thread-local message (the reference panics instead: SURVEY.md §5 "Failure detection").
#include <hip/hip_runtime_api.h>
#include <cstring>
#include <string>
#include "../../include/zkgl.h"
#include "../../include/zkgl_vm.h"
#include "cs.hpp"
#include "device_api.hpp"
#include "gadgets.hpp"
#include "poseidon_consts.hpp"

namespace zkgl {
void ram_permutation_configure(CS& cs);
void ram_permutation_entry_point(CS& cs, uint32_t limit);
void main_vm_configure(CS& cs, const zk_opcode_defs& defs, uint32_t flags);
void main_vm_entry_point(CS& cs, uint32_t limit);
void keccak_configure(CS& cs);
void sha256_configure(CS& cs);
void sha256_configure_reference_tables(CS& cs);
void linear_hasher_configure(CS& cs);
void linear_hasher_entry_point(CS& cs, uint32_t limit);
void code_unpacker_configure(CS& cs);
void unpack_code_into_memory_entry_point(CS& cs, uint32_t limit);
void sort_decommits_configure(CS& cs);
void sort_and_deduplicate_code_decommittments_entry_point(CS& cs, uint32_t limit);
void demux_log_queue_configure(CS& cs);
void demultiplex_storage_logs_entry_point(CS& cs, uint32_t limit);
void eip_4844_configure(CS& cs);
void eip_4844_entry_point(CS& cs, uint32_t n_chunks);
void sha256_blocks_entry_point(CS& cs, uint32_t n_blocks);
void sha256_round_function_entry_point(CS& cs, uint32_t limit);
void keccak256_blocks_entry_point(CS& cs, uint32_t n_blocks);
void keccak_f1600_gadget(CS& cs, zk_var* state);
void sha256_compress_gadget(CS& cs, zk_var* state, const zk_var* block);
void keccak256_round_function_entry_point(CS& cs, uint32_t limit);
void log_sorter_configure(CS& cs);
void sort_and_deduplicate_events_entry_point(CS& cs, uint32_t limit);
void storage_validity_configure(CS& cs);
void sort_and_deduplicate_storage_access_entry_point(CS& cs, uint32_t limit, bool enforce_permutation);
}  // namespace zkgl

struct zk_cs {
    zkgl::CS* cs;
};

namespace {
thread_local std::string g_err;
bool g_inited = false;
int g_device = -1;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) { return fail(ZK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
int dev_rc(int rc) { return rc == 0 ? ZK_OK : fail(rc == -1 ? ZK_ERR_INVALID : ZK_ERR_HIP, zkdev::last_hip_error()); }

template <class F>
int guard(F&& f) {
    try {
        f();
        return ZK_OK;
    } catch (const zkgl::ZkError& e) {
        return fail(e.code, e.what());
    } catch (const std::exception& e) {
        return fail(ZK_ERR_INVALID, e.what());
    } catch (...) {
        return fail(ZK_ERR_INVALID, "unknown exception");
    }
}
int need_init() { return g_inited ? ZK_OK : fail(ZK_ERR_HIP, "zk_init() has not succeeded: no GPU context (there is no CPU fallback)"); }
#define NEED_INIT() do { int rc__ = need_init(); if (rc__) return rc__; } while (0)
#define NEED(p) do { if (!(p)) return fail(ZK_ERR_INVALID, "null argument: " #p); } while (0)
}  // namespace

namespace zkgl {  // for comm.cpp
void set_last_error(const std::string& m) { g_err = m; }
CS* cs_of(zk_cs* h) { return h->cs; }
int initialized_device() { return g_inited ? g_device : -1; }
static int g_cu_count = 256;
int device_cu_count() { return g_cu_count; }   // compute units of the device zk_init bound (256 on a whole MI355X; fewer under CPX / partitioned modes)
}  // namespace zkgl

extern "C" {

const char* zk_last_error(void) { return g_err.c_str(); }

uint32_t zk_build_features(void) {
    return ZK_BUILD_BYTEBUF_KERNEL | ZK_BUILD_SHA4_KERNEL;   // one build: every device path of the tree is in it
}

int zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zk_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(ZK_ERR_HIP, "no HIP device visible: libzkgl has no CPU fallback");
    if (device < 0 || device >= n) return fail(ZK_ERR_INVALID, "device index out of range");
    // one device per process (one process per GPU, as the multi-GPU path runs): the Poseidon2 constants, the NTT twiddle tables and the
    // kernels' LDS opt-ins are process-wide state bound to the first device
    if (g_inited && device != g_device) return fail(ZK_ERR_INVALID, "zk_init: this process is already bound to another device (one process per GPU)");
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = zkdev::upload_round_constants(zkgl::poseidon_round_constants());
    if (rc) return fail(ZK_ERR_HIP, zkdev::last_hip_error());
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) zkgl::g_cu_count = cus;
    }
    g_inited = true;
    g_device = device;
    return ZK_OK;
}

int zk_poseidon_round_constants(uint64_t out[360]) {
    NEED(out);
    std::memcpy(out, zkgl::poseidon_round_constants(), 360 * sizeof(uint64_t));
    return ZK_OK;
}

int zk_malloc(void** dptr, size_t bytes) {
    NEED(dptr); NEED_INIT();
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipMalloc");
}
int zk_free(void* dptr) {
    hipError_t e = hipFree(dptr);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipFree");
}
int zk_memset(void* dptr, int value, size_t bytes, void* stream) {
    NEED_INIT();
    hipError_t e = hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipMemsetAsync");
}
int zk_h2d(void* dptr, const void* hptr, size_t bytes, void* stream) {
    NEED_INIT();
    hipError_t e = hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipMemcpy H2D");
}
int zk_d2h(void* hptr, const void* dptr, size_t bytes, void* stream) {
    NEED_INIT();
    hipError_t e = hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipMemcpy D2H");
}
int zk_sync(void* stream) {
    NEED_INIT();
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? ZK_OK : hip_fail(e, "hipStreamSynchronize");
}

// ---- K1 ----
int zk_gl_fma_cols(uint64_t* dst, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t q, uint64_t l,
                   size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(0, dst, a, b, c, q, l, n, stream));
}
int zk_gl_add_cols(uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(1, dst, a, b, a, 0, 0, n, stream));
}
int zk_gl_sub_cols(uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(2, dst, a, b, a, 0, 0, n, stream));
}
int zk_gl_mul_cols(uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(3, dst, a, b, a, 0, 0, n, stream));
}
int zk_gl_select_cols(uint64_t* dst, const uint64_t* s, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(4, dst, s, a, b, 0, 0, n, stream));
}
int zk_gl_inv_cols(uint64_t* dst, const uint64_t* a, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_col(5, dst, a, a, a, 0, 0, n, stream));
}

// ---- K2 / K3 / a9 / K4 ----
int zk_poseidon2_permute_soa(uint64_t* states, size_t n, size_t stride, void* stream) {
    NEED_INIT();
    if (stride < n) return fail(ZK_ERR_INVALID, "stride < n");
    return dev_rc(zkdev::launch_poseidon2_soa(states, n, stride, stream));
}
int zk_poseidon2_permute_aos(uint64_t* states, size_t n, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_poseidon2_aos(states, n, stream));
}
int zk_commit_encoding_batch(const uint64_t* input, size_t len, size_t n, uint64_t* out, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_commit_encoding(input, len, n, out, stream));
}
int zk_queue_full_push_chain(const uint64_t* enc, size_t nq, size_t items, uint64_t* tail_io, uint64_t* states_out,
                             void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_queue_full_chain(enc, nq, items, tail_io, states_out, stream));
}
int zk_memory_query_encode(const uint64_t* q, size_t n, uint64_t* enc, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_memory_query_encode(q, n, enc, stream));
}
int zk_execution_context_encode(const uint64_t* rec, size_t n, uint64_t* enc, void* stream) {
    NEED_INIT();
    return dev_rc(zkdev::launch_execution_context_encode(rec, n, enc, stream));
}
int zk_grand_product(const uint64_t* enc, const uint64_t* flags, const uint64_t* challenges, size_t enc_len, size_t n,
                     uint64_t init, uint64_t* acc_out, uint64_t* scratch, void* stream) {
    NEED_INIT();
    if (enc_len == 0) return fail(ZK_ERR_INVALID, "ENCODING_LENGTH must be > 0");  // src/utils.rs:96
    if (init >= 0xFFFFFFFF00000001ull) return fail(ZK_ERR_INVALID, "non-canonical init");
    return dev_rc(zkdev::launch_grand_product(enc, flags, challenges, enc_len, n, init, acc_out, scratch, stream));
}

// ---- constraint system ----
int zk_cs_create(const zk_geometry* geometry, uint64_t max_trace_len, uint64_t max_variables, zk_cs** out) {
    NEED(geometry); NEED(out);
    return guard([&] {
        auto* h = new zk_cs;
        try { h->cs = new zkgl::CS(*geometry, max_trace_len, max_variables); } catch (...) { delete h; throw; }
        *out = h;
    });
}
int zk_cs_destroy(zk_cs* cs) {
    if (!cs) return ZK_OK;
    delete cs->cs;
    delete cs;
    return ZK_OK;
}
int zk_cs_allow_lookup(zk_cs* cs, uint32_t width, uint32_t reps, int share) {
    NEED(cs);
    return guard([&] { cs->cs->allow_lookup(width, reps, share != 0); });
}
int zk_cs_allow_gate(zk_cs* cs, uint32_t kind) {
    NEED(cs);
    return guard([&] { cs->cs->allow_gate(kind); });
}
int zk_cs_gate_is_allowed(zk_cs* cs, uint32_t kind) { return cs && cs->cs->gate_is_allowed(kind) ? 1 : 0; }
int zk_cs_add_table(zk_cs* cs, uint32_t marker, uint32_t n_keys, uint32_t n_vals, const uint64_t* rows, uint32_t n_rows,
                    uint32_t* id) {
    NEED(cs); NEED(rows); NEED(id);
    return guard([&] { *id = cs->cs->add_table(marker, n_keys, n_vals, rows, n_rows); });
}
int zk_cs_table_id(zk_cs* cs, uint32_t marker, uint32_t* id) {
    NEED(cs); NEED(id);
    return guard([&] { *id = cs->cs->table_id(marker); });
}
int zk_cs_alloc_vars(zk_cs* cs, uint32_t n, zk_var* first) {
    NEED(cs); NEED(first);
    return guard([&] { *first = cs->cs->alloc_vars(n); });
}
int zk_cs_alloc_constant(zk_cs* cs, uint64_t value, zk_var* out) {
    NEED(cs); NEED(out);
    return guard([&] { *out = cs->cs->alloc_constant(value); });
}
int zk_cs_input(zk_cs* cs, uint32_t word, zk_var* out) {
    NEED(cs); NEED(out);
    return guard([&] { *out = cs->cs->input(word); });
}
int zk_cs_place_gate(zk_cs* cs, uint32_t kind, const zk_var* vars, uint32_t n_vars, const uint64_t* consts,
                     uint32_t n_consts) {
    NEED(cs);
    return guard([&] { cs->cs->place_gate(kind, vars, n_vars, consts, n_consts); });
}
int zk_cs_emit_op(zk_cs* cs, uint32_t opcode, uint32_t a, uint32_t b, const zk_var* ins, uint32_t n_in,
                  const zk_var* outs, uint32_t n_out, const uint64_t* imm, uint32_t n_imm) {
    NEED(cs);
    return guard([&] { cs->cs->emit_op(opcode, a, b, ins, n_in, outs, n_out, imm, n_imm); });
}
int zk_cs_lookup(zk_cs* cs, uint32_t table_id, const zk_var* keys, uint32_t n_keys, zk_var* vals, uint32_t n_vals) {
    NEED(cs); NEED(keys); NEED(vals);
    return guard([&] { cs->cs->lookup(table_id, keys, n_keys, vals, n_vals); });
}
int zk_cs_side_begin(zk_cs* cs) {
    NEED(cs);
    return guard([&] { cs->cs->side_begin(); });
}
int zk_cs_loop_begin(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { cs->cs->loop_begin(limit); });
}
int zk_cs_loop_end(zk_cs* cs) {
    NEED(cs);
    return guard([&] { cs->cs->loop_end(); });
}
int zk_cs_link(zk_cs* cs, uint32_t kind, zk_var loop_var, zk_var other) {
    NEED(cs);
    return guard([&] { cs->cs->link(kind, loop_var, other); });
}
int zk_cs_stream_link(zk_cs* cs, const zk_var* a_vars, uint32_t period_a, const zk_var* b_vars, uint32_t period_b, uint32_t n_total) {
    NEED(cs); NEED(a_vars); NEED(b_vars);
    return guard([&] { cs->cs->stream_link(a_vars, period_a, b_vars, period_b, n_total); });
}
int zk_cs_seed_hint(zk_cs* cs, uint32_t opcode, const zk_var* ins, uint32_t n_in, const zk_var* outs, uint32_t n_out) {
    NEED(cs); NEED(ins); NEED(outs);
    return guard([&] { cs->cs->seed_hint(opcode, ins, n_in, outs, n_out); });
}
int zk_cs_loop_last(zk_cs* cs, zk_var loop_var, zk_var* outer_out) {
    NEED(cs); NEED(outer_out);
    return guard([&] { *outer_out = cs->cs->loop_last(loop_var); });
}
int zk_cs_loop_import(zk_cs* cs, zk_var outer_var, zk_var* loop_out) {
    NEED(cs); NEED(loop_out);
    return guard([&] { *loop_out = cs->cs->loop_import(outer_var); });
}
int zk_cs_next_available_row(zk_cs* cs, uint64_t* row) {
    NEED(cs); NEED(row);
    return guard([&] { *row = cs->cs->next_available_row(); });
}
int zk_cs_finalize(zk_cs* cs) {
    NEED(cs);
    return guard([&] { cs->cs->finalize(); });
}
int zk_cs_set_batch(zk_cs* cs, uint32_t n) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->set_batch(n); });
}
int zk_cs_bind_inputs(zk_cs* cs, int loop_scope, const uint64_t* dev_words, uint32_t n_words) {
    NEED(cs);
    return guard([&] { cs->cs->bind_inputs(loop_scope != 0, dev_words, n_words); });
}
int zk_cs_bind_inputs_window(zk_cs* cs, int loop_scope, const uint64_t* dev_words, uint32_t n_words, uint64_t lane_stride) {
    NEED(cs);
    return guard([&] { cs->cs->bind_inputs(loop_scope != 0, dev_words, n_words, lane_stride); });
}
int zk_cs_seed_stream(zk_cs* cs, uint32_t n_instances, const uint64_t* dev_outer_inputs, uint64_t* dev_loop_inputs_rw, void* stream) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->seed_stream(n_instances, dev_outer_inputs, dev_loop_inputs_rw, stream); });
}
int zk_cs_resolve(zk_cs* cs, void* stream) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->resolve(stream); });
}
int zk_cs_seed_window_async(zk_cs* cs, uint32_t n_instances, const uint64_t* dev_outer_window, uint64_t outer_lane_stride, uint64_t* dev_loop_window_rw,
                            uint64_t loop_lane_stride, void* stream) {
    NEED(cs); NEED_INIT();
    if (n_instances && (!dev_outer_window || !dev_loop_window_rw)) return fail(ZK_ERR_INVALID, "zk_cs_seed_window_async: null window");
    return guard([&] { cs->cs->seed_stream(n_instances, dev_outer_window, dev_loop_window_rw, stream, false, outer_lane_stride, loop_lane_stride); });
}
int zk_cs_carried_words(zk_cs* cs, uint32_t* words, uint32_t max_words, uint32_t* n_words) {
    NEED(cs); NEED(n_words);
    return guard([&] {
        const std::vector<uint32_t> w = cs->cs->carried_words();
        *n_words = (uint32_t)w.size();
        if (!words) return;
        if (max_words < w.size()) throw zkgl::ZkError(ZK_ERR_CAPACITY, "zk_cs_carried_words: buffer too small");
        for (size_t i = 0; i < w.size(); ++i) words[i] = w[i];
    });
}
int zk_cs_set_seed_given(zk_cs* cs, const uint32_t* loop_words, uint32_t n_words) {
    NEED(cs);
    if (n_words && !loop_words) return fail(ZK_ERR_INVALID, "zk_cs_set_seed_given: null words");
    return guard([&] { cs->cs->set_seed_given(loop_words, n_words); });
}
int zk_cs_seed_carried_inputs(zk_cs* cs, uint64_t* dev_loop_inputs_rw, void* stream) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->seed_carried_inputs(dev_loop_inputs_rw, stream); });
}
int zk_cs_set_check_mode(zk_cs* cs, uint32_t mode) {
    NEED(cs);
    if (mode > ZK_CHECK_FUSED_DEFER_P2) return fail(ZK_ERR_INVALID, "zk_cs_set_check_mode: unknown mode");
    return guard([&] { cs->cs->set_check_mode(mode); });
}
int zk_cs_complete_store(zk_cs* cs, void* stream) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->ensure_p2_filled(stream); });
}
int zk_cs_narrow_byte_input_words(zk_cs* cs, uint32_t* buf, size_t max_words, size_t* n_words) {
    NEED(cs);
    if (!n_words) return fail(ZK_ERR_INVALID, "zk_cs_narrow_byte_input_words: null count");
    return guard([&] {
        const std::vector<uint32_t> w = cs->cs->narrow_byte_input_words();
        *n_words = w.size();
        if (buf) { if (max_words < w.size()) throw zkgl::ZkError(ZK_ERR_CAPACITY, "zk_cs_narrow_byte_input_words: buffer too small"); std::copy(w.begin(), w.end(), buf); }
    });
}
int zk_cs_check_satisfied(zk_cs* cs, void* stream, zk_failure* first) {
    NEED(cs); NEED_INIT();
    int result = ZK_OK;
    int rc = guard([&] { result = cs->cs->check_satisfied(stream, first); });
    if (rc) return rc;
    if (result == ZK_ERR_UNSATISFIED) return fail(ZK_ERR_UNSATISFIED, "constraint system is not satisfied");
    return result;
}
int zk_cs_resolve_and_check(zk_cs* cs, void* stream, zk_failure* first) {
    NEED(cs); NEED_INIT();
    int result = ZK_OK;
    int rc = guard([&] { result = cs->cs->resolve_and_check(stream, first); });
    if (rc) return rc;
    if (result == ZK_ERR_UNSATISFIED) return fail(ZK_ERR_UNSATISFIED, "constraint system is not satisfied");
    return result;
}
int zk_cs_read_var(zk_cs* cs, zk_var var, uint32_t instance, uint32_t iteration, uint64_t* out) {
    NEED(cs); NEED(out);
    return guard([&] { *out = cs->cs->read_var(var, instance, iteration); });
}
int zk_cs_hook_compare_witness(zk_cs* cs, const zk_var* vars, uint32_t n_vars, const uint64_t* dev_expected, void* stream, zk_failure* first) {
    NEED(cs); NEED_INIT();
    int result = ZK_OK;
    int rc = guard([&] { result = cs->cs->hook_compare_witness(vars, n_vars, dev_expected, stream, first); });
    if (rc) return rc;
    return result == ZK_OK ? ZK_OK : fail(ZK_ERR_UNSATISFIED, "circuit values differ from the expected closed-form input");
}
int zk_cs_debug_poke_store(zk_cs* cs, int loop_scope, uint32_t slot, uint32_t lane, uint64_t value) {
    NEED(cs); NEED_INIT();
    return guard([&] { cs->cs->debug_poke_store(loop_scope != 0, slot, lane, value); });
}
int zk_cs_store_slots(zk_cs* cs, int loop_scope, uint32_t* n) {
    NEED(cs); NEED(n);
    *n = cs->cs->store_slots(loop_scope != 0);
    return ZK_OK;
}
int zk_cs_write_cell(zk_cs* cs, int loop_scope, uint32_t cell, uint32_t lane, uint64_t value) {
    NEED(cs);
    return guard([&] { cs->cs->write_cell(loop_scope != 0, cell, lane, value); });
}
int zk_cs_public_inputs(zk_cs* cs, uint32_t instance, uint64_t* out, uint32_t max, uint32_t* n) {
    NEED(cs); NEED(n);
    return guard([&] {
        auto v = cs->cs->public_inputs(instance);
        *n = (uint32_t)v.size();
        if (out) for (uint32_t i = 0; i < v.size() && i < max; ++i) out[i] = v[i];
    });
}
int zk_cs_var_cell(zk_cs* cs, zk_var var, uint32_t* cell) {
    NEED(cs); NEED(cell);
    return guard([&] { *cell = cs->cs->var_cell(var); });
}
int zk_cs_public_cells(zk_cs* cs, uint32_t* cells, uint32_t max, uint32_t* n) {
    NEED(cs); NEED(n);
    return guard([&] {
        auto v = cs->cs->public_cells();
        *n = (uint32_t)v.size();
        if (cells) for (uint32_t i = 0; i < v.size() && i < max; ++i) cells[i] = v[i];
    });
}
int zk_cs_multiplicities(zk_cs* cs, uint32_t instance, uint32_t* out, uint32_t max, uint32_t* n) {
    NEED(cs); NEED(n);
    return guard([&] {
        auto v = cs->cs->multiplicities(instance);
        *n = (uint32_t)v.size();
        if (out) for (uint32_t i = 0; i < v.size() && i < max; ++i) out[i] = v[i];
    });
}
int zk_cs_lookup_argument(zk_cs* cs, const uint64_t beta[2], const uint64_t gamma[2], void* stream, uint64_t* out, uint32_t max_instances,
                          uint32_t* n_mismatch) {
    NEED(cs); NEED(beta); NEED(gamma); NEED(n_mismatch); NEED_INIT();
    return guard([&] {
        std::vector<uint64_t> v;
        *n_mismatch = cs->cs->lookup_argument(beta, gamma, stream, v);
        if (out) for (size_t i = 0; i < v.size() && i < 4 * (size_t)max_instances; ++i) out[i] = v[i];
    });
}
int zk_cs_copy_permutation(zk_cs* cs, const uint64_t beta[2], const uint64_t gamma[2], void* stream, uint64_t* dev_z, uint64_t* out,
                           uint32_t max_instances, uint32_t* n_mismatch) {
    NEED(cs); NEED(beta); NEED(gamma); NEED(n_mismatch); NEED_INIT();
    return guard([&] {
        std::vector<uint64_t> v;
        *n_mismatch = cs->cs->copy_permutation(beta, gamma, stream, dev_z, v);
        if (out) for (size_t i = 0; i < v.size() && i < 4 * (size_t)max_instances; ++i) out[i] = v[i];
    });
}
int zk_cs_sigma(zk_cs* cs, int loop_scope, uint32_t iteration, uint64_t* buf, size_t max_words, size_t* n_words) {
    NEED(cs); NEED(n_words);
    return guard([&] {
        std::vector<uint64_t> v = cs->cs->sigma_labels(loop_scope != 0, iteration);
        *n_words = v.size();
        if (buf) for (size_t i = 0; i < v.size() && i < max_words; ++i) buf[i] = v[i];
    });
}
int zk_two_adic_root(uint32_t log_n, uint64_t* out) {
    NEED(out);
    return guard([&] { *out = zkgl::two_adic_root(log_n); });
}
int zk_ntt(uint64_t* dev_data, uint32_t log_n, uint32_t n_polys, uint64_t stride, uint32_t mode, uint64_t coset_shift, void* stream) {
    NEED(dev_data); NEED_INIT();
    return guard([&] { zkgl::ntt(dev_data, log_n, n_polys, stride, mode, coset_shift, stream); });
}
int zk_lde(const uint64_t* dev_coeffs, uint64_t src_stride, uint64_t* dev_out, uint32_t log_n, uint32_t log_blowup, uint32_t n_polys,
           uint32_t mode, uint64_t coset_shift, void* stream) {
    NEED(dev_coeffs); NEED(dev_out); NEED_INIT();
    return guard([&] { zkgl::lde(dev_coeffs, src_stride, dev_out, log_n, log_blowup, n_polys, mode, coset_shift, stream); });
}
int zk_cs_stats(zk_cs* cs, zk_stats* out) {
    NEED(cs); NEED(out);
    return guard([&] { cs->cs->stats(out); });
}
int zk_cs_last_ms(zk_cs* cs, int which, float* ms) {
    NEED(cs); NEED(ms);
    *ms = cs->cs->last_ms(which);
    return *ms < 0 ? fail(ZK_ERR_INVALID, "bad timer index") : ZK_OK;
}
int zk_cs_export(zk_cs* cs, int loop_scope, uint32_t* buf, size_t max_words, size_t* n_words) {
    NEED(cs); NEED(n_words);
    return guard([&] {
        if (!cs->cs->finalized()) throw zkgl::ZkError(ZK_ERR_INVALID, "export before finalize");
        auto v = cs->cs->export_scope(loop_scope != 0);
        *n_words = v.size();
        if (buf) {
            if (max_words < v.size()) throw zkgl::ZkError(ZK_ERR_INVALID, "export buffer too small");
            std::memcpy(buf, v.data(), v.size() * 4);
        }
    });
}
int zk_cs_trace_columns(zk_cs* cs, uint32_t instance, uint64_t* dev_out, uint32_t log_n, uint64_t stride, void* stream) {
    NEED(cs); NEED(dev_out); NEED_INIT();
    return guard([&] { cs->cs->trace_columns(instance, dev_out, log_n, stride, stream); });
}
int zk_cs_trace_columns_batch(zk_cs* cs, uint32_t first_instance, uint32_t n_instances, uint64_t* dev_out, uint32_t log_n, uint64_t stride, uint64_t instance_stride,
                              void* stream) {
    NEED(cs); NEED(dev_out); NEED_INIT();
    return guard([&] { cs->cs->trace_columns(first_instance, dev_out, log_n, stride, stream, n_instances, instance_stride); });
}
int zk_cs_trace_ptr(zk_cs* cs, int loop_scope, uint64_t** dev_cells, uint64_t* n_cells, uint64_t* stride) {
    NEED(cs); NEED(dev_cells); NEED(n_cells); NEED(stride);
    return guard([&] { cs->cs->trace_ptr(loop_scope != 0, dev_cells, n_cells, stride); });
}

// ---- circuits ----
int zk_circuit_ram_permutation_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::ram_permutation_configure(*cs->cs); });
}
int zk_circuit_ram_permutation(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::ram_permutation_entry_point(*cs->cs, limit); });
}
int zk_circuit_hook_vars(zk_cs* cs, const char* name, zk_var* vars, uint32_t max, uint32_t* n) {
    NEED(cs); NEED(name); NEED(n);
    auto it = cs->cs->hooks.find(name);
    if (it == cs->cs->hooks.end()) return fail(ZK_ERR_INVALID, std::string("the recorded circuit publishes no hook group named ") + name);
    *n = (uint32_t)it->second.size();
    if (vars) for (uint32_t i = 0; i < std::min<uint32_t>(max, *n); ++i) vars[i] = it->second[i];
    return ZK_OK;
}
int zk_circuit_input_words(zk_cs* cs, uint32_t* outer_words, uint32_t* loop_words) {
    NEED(cs); NEED(outer_words); NEED(loop_words);
    *outer_words = cs->cs->outer_input_words();
    *loop_words = cs->cs->loop_input_words();
    return ZK_OK;
}
int zk_circuit_storage_validity_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::storage_validity_configure(*cs->cs); });
}
int zk_circuit_storage_validity(zk_cs* cs, uint32_t limit, int enforce_permutation) {
    NEED(cs);
    return guard([&] { zkgl::sort_and_deduplicate_storage_access_entry_point(*cs->cs, limit, enforce_permutation != 0); });
}
int zk_circuit_log_sorter_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::log_sorter_configure(*cs->cs); });
}
int zk_circuit_log_sorter(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::sort_and_deduplicate_events_entry_point(*cs->cs, limit); });
}
int zk_circuit_keccak_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::keccak_configure(*cs->cs); });
}
int zk_circuit_keccak256_blocks(zk_cs* cs, uint32_t n_blocks) {
    NEED(cs);
    return guard([&] { zkgl::keccak256_blocks_entry_point(*cs->cs, n_blocks); });
}
int zk_gadget_keccak_f1600(zk_cs* cs, zk_var* state_io) {
    NEED(cs); NEED(state_io);
    return guard([&] { zkgl::keccak_f1600_gadget(*cs->cs, state_io); });
}
int zk_gadget_sha256_compress(zk_cs* cs, zk_var* state_io, const zk_var* block) {
    NEED(cs); NEED(state_io); NEED(block);
    return guard([&] { zkgl::sha256_compress_gadget(*cs->cs, state_io, block); });
}
int zk_circuit_keccak256_round_function(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::keccak256_round_function_entry_point(*cs->cs, limit); });
}
int zk_circuit_eip_4844_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::eip_4844_configure(*cs->cs); });
}
int zk_circuit_eip_4844(zk_cs* cs, uint32_t n_chunks) {
    NEED(cs);
    return guard([&] { zkgl::eip_4844_entry_point(*cs->cs, n_chunks); });
}
int zk_circuit_demux_log_queue_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::demux_log_queue_configure(*cs->cs); });
}
int zk_circuit_demux_log_queue(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::demultiplex_storage_logs_entry_point(*cs->cs, limit); });
}
int zk_circuit_sort_decommits_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::sort_decommits_configure(*cs->cs); });
}
int zk_circuit_sort_decommits(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::sort_and_deduplicate_code_decommittments_entry_point(*cs->cs, limit); });
}
int zk_circuit_code_unpacker_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::code_unpacker_configure(*cs->cs); });
}
int zk_circuit_code_unpacker(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::unpack_code_into_memory_entry_point(*cs->cs, limit); });
}
int zk_circuit_linear_hasher_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::linear_hasher_configure(*cs->cs); });
}
int zk_circuit_linear_hasher(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::linear_hasher_entry_point(*cs->cs, limit); });
}
int zk_circuit_sha256_configure_reference_tables(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::sha256_configure_reference_tables(*cs->cs); });
}
int zk_circuit_sha256_configure(zk_cs* cs) {
    NEED(cs);
    return guard([&] { zkgl::sha256_configure(*cs->cs); });
}
int zk_circuit_sha256_blocks(zk_cs* cs, uint32_t n_blocks) {
    NEED(cs);
    return guard([&] { zkgl::sha256_blocks_entry_point(*cs->cs, n_blocks); });
}
int zk_circuit_sha256_round_function(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::sha256_round_function_entry_point(*cs->cs, limit); });
}
int zk_circuit_main_vm_configure_flags(zk_cs* cs, const zk_opcode_defs* defs, uint32_t flags) {
    NEED(cs);
    if (!defs) return fail(ZK_ERR_INVALID, "zk_circuit_main_vm_configure_flags: null opcode-defs blob");
    if (flags & ~(uint32_t)ZK_VM_CFG_U32_FMA_ROLE) return fail(ZK_ERR_INVALID, "zk_circuit_main_vm_configure_flags: unknown flag");
    return guard([&] { zkgl::main_vm_configure(*cs->cs, *defs, flags); });
}
int zk_circuit_main_vm_configure(zk_cs* cs, const zk_opcode_defs* defs) {
    NEED(cs);
    if (!defs) return fail(ZK_ERR_INVALID, "zk_circuit_main_vm_configure: null opcode-defs blob");
    return guard([&] { zkgl::main_vm_configure(*cs->cs, *defs, 0); });
}
int zk_circuit_main_vm(zk_cs* cs, uint32_t limit) {
    NEED(cs);
    return guard([&] { zkgl::main_vm_entry_point(*cs->cs, limit); });
}
int zk_circuit_main_vm_layout(zk_cs* cs, char* buf, size_t max_bytes, size_t* n_bytes) {
    NEED(cs);
    const std::string& t = cs->cs->input_layout;
    if (n_bytes) *n_bytes = t.size();
    if (!buf) return ZK_OK;
    if (max_bytes < t.size()) return fail(ZK_ERR_CAPACITY, "zk_circuit_main_vm_layout: buffer too small");
    std::memcpy(buf, t.data(), t.size());
    return ZK_OK;
}

}  // extern "C"
