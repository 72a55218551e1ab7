// cs_perm.cpp — host side of K12 (kernels_perm.hpp): the permutation sigma over the trace-cell labels and the grand product.
//
// Copy classes of this engine = the cells of one variable inside a scope and iteration (Scope::var_cells), joined by the
// links that stand for the reference's cross-chunk / cross-cycle copies (hidden_fsm chain,
// /root/reference/src/ram_permutation/mod.rs:119-143,178-196) and by the stream links (include/zkgl_ir.h).  sigma starts as
// one cycle per variable and scope-iteration; every link whose two ends still sit in different cycles swaps the images of
// its ends, which splices the two cycles into one (a union-find over the link endpoints keeps a second link between
// already joined classes from splitting them again).  Only link endpoints ever change, so sigma is stored as a per-cell
// table relative to the iteration plus a dense [endpoint][iteration] table of absolute labels.
#include <hip/hip_runtime.h>
#include <unordered_map>
#include "cs.hpp"
#include "device_api.hpp"

namespace zkgl {
namespace {
constexpr uint64_t P = 0xFFFFFFFF00000001ull;
constexpr uint32_t NONE = 0xffffffffu;
void hipc(hipError_t e, const char* what) {
    if (e != hipSuccess) throw ZkError(ZK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
void devc(int rc) { if (rc) throw ZkError(ZK_ERR_HIP, zkdev::last_hip_error()); }
template <typename T> T* up(const std::vector<T>& v) {
    T* d = nullptr;
    hipc(hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * sizeof(T)), "hipMalloc sigma");
    if (!v.empty()) hipc(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy sigma");
    return d;
}
}  // namespace

void CS::build_sigma() {
    if (sigma_built_) return;
    if (!finalized_) throw ZkError(ZK_ERR_INVALID, "copy_permutation before finalize");
    const uint64_t NTo = outer_.n_trace_cells, NTl = limit_ ? loop_.n_trace_cells : 0;
    const uint32_t L = std::max<uint32_t>(limit_, 1);
    Scope* sc[2] = {&outer_, &loop_};
    for (int s = 0; s < 2; ++s) {
        const uint32_t nt = s && !limit_ ? 0 : sc[s]->n_trace_cells;
        sig_rel_[s].resize(nt);
        for (uint32_t c = 0; c < nt; ++c) sig_rel_[s][c] = c;
        ep_index_[s].assign(nt, NONE);
        if (nt == 0) continue;
        for (auto& cells : sc[s]->var_cells) {  // one cycle per variable over its trace cells
            if (cells.size() < 2) continue;
            for (size_t i = 0; i < cells.size(); ++i) sig_rel_[s][cells[i]] = cells[(i + 1) % cells.size()];
        }
    }
    // link endpoints (trace cells only: a variable no gate references takes no part in the argument)
    std::vector<uint32_t> ep_cells[2];
    auto endpoint = [&](int s, uint32_t cell) -> bool {
        if (cell >= sig_rel_[s].size()) return false;
        if (ep_index_[s][cell] == NONE) { ep_index_[s][cell] = (uint32_t)ep_cells[s].size(); ep_cells[s].push_back(cell); }
        return true;
    };
    struct Join { uint64_t a, b; };  // labels
    auto label = [&](int s, uint32_t k, uint32_t cell) -> uint64_t { return s ? NTo + (uint64_t)k * NTl + cell : cell; };
    std::vector<Join> joins;
    if (limit_) {
        for (auto& l : links_) {
            const bool carry = l.kind == ZK_LINK_CARRY;
            if (!endpoint(1, l.loop_cell) || !endpoint(carry ? 1 : 0, l.other_cell)) continue;
            if (carry) for (uint32_t k = 1; k < limit_; ++k) joins.push_back({label(1, k, l.loop_cell), label(1, k - 1, l.other_cell)});
            else if (l.kind == ZK_LINK_FIRST) joins.push_back({label(1, 0, l.loop_cell), label(0, 0, l.other_cell)});
            else if (l.kind == ZK_LINK_LAST) joins.push_back({label(1, limit_ - 1, l.loop_cell), label(0, 0, l.other_cell)});
            else for (uint32_t k = 0; k < limit_; ++k) joins.push_back({label(1, k, l.loop_cell), label(0, 0, l.other_cell)});
        }
        for (auto& sr : streams_) {
            bool ok = true;
            for (uint32_t c : sr.a) ok = endpoint(1, c) && ok;
            for (uint32_t c : sr.b) ok = endpoint(1, c) && ok;
            if (!ok) continue;
            const uint32_t pa = (uint32_t)sr.a.size(), pb = (uint32_t)sr.b.size();
            for (uint32_t k = 0; k < sr.n_total; ++k) joins.push_back({label(1, k / pa, sr.a[k % pa]), label(1, k / pb, sr.b[k % pb])});
        }
    }
    n_ep_[0] = (uint32_t)ep_cells[0].size(); n_ep_[1] = (uint32_t)ep_cells[1].size();
    ovr_[0].resize(n_ep_[0]);
    for (uint32_t e = 0; e < n_ep_[0]; ++e) ovr_[0][e] = sig_rel_[0][ep_cells[0][e]];
    ovr_[1].resize((size_t)n_ep_[1] * L);
    for (uint32_t e = 0; e < n_ep_[1]; ++e)
        for (uint32_t k = 0; k < L; ++k) ovr_[1][(size_t)e * L + k] = label(1, k, sig_rel_[1][ep_cells[1][e]]);
    auto image = [&](uint64_t lab) -> uint64_t& {  // sigma(label) of an endpoint
        if (lab < NTo) return ovr_[0][ep_index_[0][lab]];
        const uint64_t r = lab - NTo;
        return ovr_[1][(size_t)ep_index_[1][r % NTl] * L + r / NTl];
    };
    std::unordered_map<uint64_t, uint64_t> parent;
    auto find = [&](uint64_t x) {
        uint64_t r = x;
        for (auto it = parent.find(r); it != parent.end() && it->second != r; it = parent.find(r)) r = it->second;
        for (uint64_t y = x; y != r;) { auto it = parent.find(y); uint64_t nx = it->second; it->second = r; y = nx; }
        return r;
    };
    for (auto& j : joins) {
        parent.emplace(j.a, j.a); parent.emplace(j.b, j.b);
        const uint64_t ra = find(j.a), rb = find(j.b);
        if (ra == rb) continue;          // already one class: a swap would cut the cycle in two
        std::swap(image(j.a), image(j.b));
        parent[ra] = rb;
    }
    sigma_built_ = true;  // host tables only: zk_cs_sigma needs no GPU; copy_permutation uploads them on first use
}

std::vector<uint64_t> CS::sigma_labels(bool loop_scope, uint32_t iteration) {
    build_sigma();
    const int s = loop_scope ? 1 : 0;
    if (loop_scope && iteration >= limit_) throw ZkError(ZK_ERR_INVALID, "sigma_labels: iteration out of range");
    const uint64_t NTo = outer_.n_trace_cells, NTl = loop_.n_trace_cells;
    const uint32_t L = std::max<uint32_t>(limit_, 1), k = loop_scope ? iteration : 0;
    std::vector<uint64_t> out(sig_rel_[s].size());
    for (uint32_t c = 0; c < out.size(); ++c) {
        const uint32_t e = ep_index_[s][c];
        if (e != NONE) out[c] = ovr_[s][(size_t)e * (s ? L : 1) + k];
        else out[c] = s ? NTo + (uint64_t)k * NTl + sig_rel_[s][c] : sig_rel_[s][c];
    }
    return out;
}

uint32_t CS::copy_permutation(const uint64_t beta[2], const uint64_t gamma[2], void* stream, uint64_t* d_z, std::vector<uint64_t>& out) {
    if (batch_ == 0 || !uploaded_) throw ZkError(ZK_ERR_INVALID, "copy_permutation before set_batch / resolve");
    for (int i = 0; i < 2; ++i)
        if (beta[i] >= P || gamma[i] >= P) throw ZkError(ZK_ERR_INVALID, "copy_permutation: non-canonical challenge");
    // the argument runs over every populated trace cell; a compact batch is read through the trace view (cell -> store slot): the 4x
    // larger materialised trace of the whole batch is not built for this (round 3: its materialisation was most of K12's 3.3 ms / instance)
    if (compact_) ensure_trace_view();
    ensure_p2_filled(stream);
    build_sigma();
    if (!d_sig_rel_[0]) {
        for (int s = 0; s < 2; ++s) { d_sig_rel_[s] = up(sig_rel_[s]); d_ep_index_[s] = up(ep_index_[s]); d_ovr_[s] = up(ovr_[s]); }
    }
    hipStream_t st = (hipStream_t)stream;
    const uint64_t NTo = outer_.n_trace_cells, NTl = limit_ ? loop_.n_trace_cells : 0;
    const uint32_t n_cols = geo_.num_columns_under_copy_permutation + lookup_width_ * lookup_reps_;
    const uint32_t loop_lanes = limit_ ? loop_.n_lanes : 0;
    // chunks of rows per lane: the loop scope has one lane per (instance, iteration) and needs none, the outer scope has one
    // lane per instance and is cut into ~2048 / instances pieces
    const uint32_t n_slots[2] = {outer_.n_slots, limit_ ? loop_.n_slots : 0};
    uint32_t chunks[2], spc[2];
    for (int s = 0; s < 2; ++s) {
        const uint32_t lanes = s ? loop_lanes : outer_.n_lanes;
        // ~1 M threads per launch (16 wavefronts per SIMD): a lane's rows are walked with several scalar loads per cell, so the kernel lives
        // on resident wavefronts (64 instances x 2 352 cycles = 2.3 wavefronts per SIMD ran at a third of its multiplication rate)
        uint32_t want = std::max<uint32_t>(1, (1u << 20) / std::max<uint32_t>(lanes, 1));
        want = std::min(want, std::max<uint32_t>(n_slots[s], 1));
        spc[s] = (std::max<uint32_t>(n_slots[s], 1) + want - 1) / want;
        chunks[s] = (std::max<uint32_t>(n_slots[s], 1) + spc[s] - 1) / spc[s];
    }
    uint64_t *d_tb[2] = {nullptr, nullptr}, *d_part[2] = {nullptr, nullptr}, *d_excl[2] = {nullptr, nullptr}, *d_pre[2] = {nullptr, nullptr};
    uint64_t *d_total_l = nullptr, *d_inst = nullptr;
    struct Temps {  // device temporaries of this call: released on every exit path, exceptions included
        std::vector<void*> ptrs;
        ~Temps() { for (void* p : ptrs) hipFree(p); }
    } temps;
    auto alloc = [&](uint64_t** p, size_t words) {
        hipc(hipMalloc((void**)p, std::max<size_t>(words, 1) * 8), "hipMalloc copy_permutation");
        temps.ptrs.push_back(*p);
    };
    alloc(&d_tb[0], 4 * (size_t)NTo);
    alloc(&d_tb[1], 4 * (size_t)NTl);
    alloc(&d_part[0], 4 * (size_t)outer_.n_lanes * chunks[0]);
    alloc(&d_part[1], 4 * (size_t)loop_lanes * chunks[1]);
    alloc(&d_total_l, 4 * (size_t)batch_);
    alloc(&d_inst, 4 * (size_t)batch_);
    if (d_z) {
        alloc(&d_excl[0], 4 * (size_t)outer_.n_lanes * chunks[0]);
        alloc(&d_excl[1], 4 * (size_t)loop_lanes * chunks[1]);
        alloc(&d_pre[0], 4 * (size_t)n_slots[0] * outer_.n_lanes);
        alloc(&d_pre[1], 4 * (size_t)n_slots[1] * loop_lanes);
    }
    devc(zkdev::launch_perm_tb(beta, d_sig_rel_[0], d_tb[0], (uint32_t)NTo, st));
    devc(zkdev::launch_perm_tb(beta, d_sig_rel_[1], d_tb[1], (uint32_t)NTl, st));
    auto side = [&](int s) {
        const Scope& sc = s ? loop_ : outer_;
        zkdev::PermArgs a;
        if (compact_) { a.cells = sc.d_store; a.n_cells = sc.store_geom(); a.slot1 = sc.d_slot1; }
        else { a.cells = sc.d_cells; a.n_cells = sc.n_cells; a.slot1 = nullptr; }
        a.n_cols = n_cols; a.n_lanes = sc.n_lanes; a.n_slots = sc.n_slots;
        a.n_copy_cols = geo_.num_columns_under_copy_permutation; a.lookup_width = lookup_width_; a.rows = sc.d_rows; a.lrows = sc.d_lrows;
        a.sigma_rel = d_sig_rel_[s]; a.ep_index = d_ep_index_[s]; a.ovr = d_ovr_[s]; a.lanes_per_instance = s ? limit_ : 1;
        a.label_base = s ? NTo : 0; a.label_step = s ? NTl : 0; a.tb = d_tb[s];
        a.beta[0] = beta[0]; a.beta[1] = beta[1]; a.gamma[0] = gamma[0]; a.gamma[1] = gamma[1];
        a.slots_per_chunk = spc[s]; a.n_chunks = chunks[s]; a.lane_out = d_part[s]; a.prefix = d_pre[s];
        devc(zkdev::launch_perm_lane(a, st));
    };
    side(0);
    if (limit_) side(1);
    // rows of an instance: all loop rows, then the outer rows
    devc(zkdev::launch_perm_scan(d_part[1], limit_ * chunks[1], nullptr, d_excl[1], d_total_l, batch_, st));
    devc(zkdev::launch_perm_scan(d_part[0], chunks[0], d_total_l, d_excl[0], d_inst, batch_, st));
    if (d_z) {
        const uint64_t loop_rows = (uint64_t)n_slots[1] * limit_, rows = loop_rows + outer_.n_slots;
        if (limit_) devc(zkdev::launch_perm_z(d_excl[1], d_pre[1], loop_lanes, n_slots[1], spc[1], chunks[1], limit_, 0, rows, d_z, st));
        devc(zkdev::launch_perm_z(d_excl[0], d_pre[0], outer_.n_lanes, n_slots[0], spc[0], chunks[0], 1, loop_rows, rows, d_z, st));
    }
    out.resize(4 * (size_t)batch_);
    hipc(hipMemcpyAsync(out.data(), d_inst, out.size() * 8, hipMemcpyDeviceToHost, st), "memcpy copy_permutation");
    hipc(hipStreamSynchronize(st), "copy_permutation sync");
    uint32_t bad = 0;
    for (uint32_t i = 0; i < batch_; ++i) bad += (out[4 * i] != out[4 * i + 2] || out[4 * i + 1] != out[4 * i + 3]);
    return bad;
}

}  // namespace zkgl
