// tests/emu/emu_harness.cpp — the LANE HARNESS: the product's witness interpreter (csrc/kernels_engine2.hpp run_tile2, the source the GPU runs,
// cut out unchanged by tests/emu/gen.py) compiled for the host and run ONE LANE AT A TIME on the device programs a recorded circuit carries.
//
// TEST INFRASTRUCTURE (like oracle/): it exists so that a change to the interpreter or to a macro-op's device backend can be exercised when no
// GPU is reachable.  It is NOT a fallback: libzkgl has no path to it, it lives under tests/, it is built by tests/emu/build.sh only, and it
// proves nothing about wavefront-level behaviour (coalescing, LDS ordering between lanes, occupancy, timing): the -m gpu tests stay the
// parity gate.  What it does prove: the scalar semantics of every op as the kernel source spells them, the program words the host emits
// as the kernel decodes them, store addressing, the fused-mode failure flags, and the strand form's level / barrier structure.
#include <barrier>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "emu_shim.hpp"
#include "../../era-zkevm_circuits_amd/csrc/device_api.hpp"
#include "engine2_host.hpp"   // generated: the interpreter
// the recorder's internals (host programs, store slots, tables): read-only access for the harness
#define private public
#define protected public
#include "../../era-zkevm_circuits_amd/csrc/cs.hpp"
#undef private
#undef protected
#include "../../era-zkevm_circuits_amd/csrc/poseidon_consts.hpp"

namespace zkgl { CS* cs_of(zk_cs* h); void set_last_error(const std::string& m); }

namespace emu {
thread_local Dim3 tid{0, 0, 0}, bid{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
static std::barrier<>* g_barrier = nullptr;
void block_barrier() { if (g_barrier) g_barrier->arrive_and_wait(); }
}  // namespace emu

namespace {
using zkgl::CS;
using zkgl::Scope;

void init_device_globals() {
    static bool done = false;
    if (done) return;
    const uint64_t* rc = zkgl::poseidon_round_constants();
    for (int i = 0; i < 360; ++i) p2::RC[i] = rc[i];
    p2::INV_SMALL[0] = 0;
    for (uint32_t k = 1; k < p2::INV_SMALL_N; ++k) p2::INV_SMALL[k] = gl::inv(k);
    done = true;
}

std::vector<uint32_t> padded(const std::vector<uint32_t>& p) {   // as cs.cpp upload_scope pads the programs for the 16-word scalar fetches
    std::vector<uint32_t> r(p);
    r.resize(((r.size() + 63) / 64) * 64 + 192, 0);
    return r;
}

struct Run {
    CS& cs;
    uint32_t batch, limit;
    std::vector<uint64_t> store[2];   // 0 outer, 1 loop
    uint64_t geom[2];
    std::vector<uint32_t> mult;
    unsigned long long fail[16];
    unsigned long long p2_stats[2] = {0, 0};
    const uint64_t* in[2];
    uint64_t in_stride[2];
    std::vector<uint32_t> prog[2], sprog[2];
    Run(CS& c, uint32_t b) : cs(c), batch(b), limit(c.limit_) {
        for (auto& f : fail) f = ~0ull;
        for (int sc = 0; sc < 2; ++sc) {
            const Scope& s = sc ? cs.loop_ : cs.outer_;
            const uint64_t lanes = sc ? (uint64_t)batch * limit : batch;
            geom[sc] = zkgeom::pack(s.n_store, zkgeom::WAVE_TILE_LOG2);
            store[sc].assign((size_t)std::max<uint64_t>(1, (uint64_t)s.n_store * zkgeom::padded_lanes(geom[sc], std::max<uint64_t>(lanes, 1))), 0);
            prog[sc] = padded(s.prog2);
            sprog[sc] = padded(s.sprog);
        }
        mult.assign((size_t)batch * std::max<uint32_t>(cs.total_table_rows_, 1), 0);
    }
    zke::ScopeDev dev(int sc, bool fused) {
        const Scope& s = sc ? cs.loop_ : cs.outer_;
        zke::ScopeDev d;
        std::memset(&d, 0, sizeof d);
        d.prog = prog[sc].data(); d.n_words = (uint32_t)s.prog2.size();
        d.n_lanes = sc ? batch * limit : batch;
        d.consts = s.const_pool.data();
        d.cells = store[sc].data(); d.n_cells = geom[sc];
        d.inputs = in[sc]; d.in_stride = in_stride[sc];
        d.outer_cells = store[0].data(); d.outer_n_cells = geom[0];
        d.limit = sc ? limit : 1; d.is_loop = sc ? 1 : 0;
        d.tables = cs.tdesc_host_.data(); d.table_words = cs.table_words_host_.data();
        d.mult = cs.inline_multiplicities() ? mult.data() : nullptr; d.total_table_rows = cs.total_table_rows_;
        d.loop_cells = store[1].data(); d.loop_n_cells = geom[1]; d.loop_limit = limit;
        d.fail = fused ? fail + (sc ? 3 : 0) : nullptr;
        d.defer_p2 = 0; d.p2_stats = p2_stats; d.clock_probe = nullptr;
        return d;
    }
    template <bool BIG, int X = 0>
    void plain(const zke::ScopeDev& d, uint32_t w0, uint32_t w1, uint32_t slot0) {
        emu::tid = {0, 0, 0}; emu::bid = {0, 0, 0}; emu::bdim = {(unsigned)zke::TPB, 1, 1};
        for (uint32_t lane = 0; lane < d.n_lanes; ++lane)
            zke::run_tile2<BIG, false, zke::TPB, false, false, X>(d, lane, d.is_loop ? lane / d.limit : lane, true, w0, w1, slot0);
    }
    template <bool BIG, int X = 0>
    void strands(zke::ScopeDev d, const Scope& s, int phase) {
        constexpr uint32_t NS = zke::STRANDS_PER_TILE;
        d.prog = sprog[s.is_loop ? 1 : 0].data(); d.n_words = (uint32_t)s.sprog.size();
        for (uint32_t lane = 0; lane < d.n_lanes; ++lane) {   // one lane = one tile of NS one-lane wavefronts
            std::barrier<> bar(NS);
            emu::g_barrier = &bar;
            std::vector<std::thread> ts;
            for (uint32_t w = 0; w < NS; ++w)
                ts.emplace_back([&, w] {
                    emu::tid = {64 * w, 0, 0}; emu::bid = {lane / 64, 0, 0}; emu::bdim = {64 * NS, 1, 1};
                    zke::run_tile2<BIG, false, 64 * NS, true, false, X>(d, lane, d.is_loop ? lane / d.limit : lane, true, s.s_begin[phase][w], s.s_end[phase][w], 0);
                });
            for (auto& t : ts) t.join();
            emu::g_barrier = nullptr;
        }
    }
    void phase(int sc, int ph, bool use_strands, bool fused) {
        const Scope& s = sc ? cs.loop_ : cs.outer_;
        zke::ScopeDev d = dev(sc, fused);
        if (d.n_lanes == 0) return;
        // the kernels that carry a macro-op backend beyond the basic set (kernels_engine2.hpp X_SHA4 / X_BYTEBUF; CS::launch_phase -> zkdev::launch_witness*)
        const int x = (cs.uses_sha4_macro_ ? zke::X_SHA4 : 0) | (cs.uses_bytebuf_macro_ ? zke::X_BYTEBUF : 0);
        if (use_strands && !s.sprog.empty()) {
            if (x == zke::X_SHA4) strands<true, zke::X_SHA4>(d, s, ph);
            else if (x == zke::X_BYTEBUF) strands<true, zke::X_BYTEBUF>(d, s, ph);
            else if (s.uses_bigint) strands<true>(d, s, ph); else strands<false>(d, s, ph);
            return;
        }
        const uint32_t end = (uint32_t)s.prog2.size();
        uint32_t w0 = 0, w1 = end, slot0 = 0;
        if (!s.is_loop) {
            if (ph == 0) w1 = s.pre_words2;
            else if (ph == 1) { w0 = s.pre_words2; w1 = s.side_words2; slot0 = s.pre_slots; }
            else { w0 = s.side_words2; slot0 = s.side_slots; }
        }
        if (w0 >= w1) return;
        if (x == zke::X_SHA4) plain<true, zke::X_SHA4>(d, w0, w1, slot0);
        else if (x == zke::X_BYTEBUF) plain<true, zke::X_BYTEBUF>(d, w0, w1, slot0);
        else if (s.uses_bigint) plain<true>(d, w0, w1, slot0); else plain<false>(d, w0, w1, slot0);
    }
    void trace(int sc, uint64_t* out, uint64_t stride) {   // k_materialize: trace cell <- store slot, the populated cells only
        const Scope& s = sc ? cs.loop_ : cs.outer_;
        const uint64_t lanes = sc ? (uint64_t)batch * limit : batch;
        for (auto& pr : s.mat_pairs)
            for (uint64_t l = 0; l < lanes; ++l) out[(uint64_t)pr.cell * stride + l] = store[sc][zkgeom::offset(geom[sc], pr.home, l)];
    }
};
}  // namespace

static std::unique_ptr<Run> g_last;

// Resolve `batch` instances of the recorded circuit on the lane harness.  Inputs: host arrays in the C ABI's stream layout (outer
// [word][batch], loop [word][batch * limit], loop-carried words already seeded).  strands: 0 plain kernels, 1 strand form where a scope
// has one.  Outputs (any may be NULL): the materialised traces [n_cells][stride] (stride = lanes rounded up to 64, as zk_cs_trace_ptr
// describes them), the public inputs [batch][n_public], the fused-mode failure words (16: outer 0..2, loop 3..5 as the device keeps
// them; ~0 = none), the inline multiplicities [batch][total_table_rows].
extern "C" int zk_emu_resolve(zk_cs* h, const uint64_t* outer_in, const uint64_t* loop_in, uint32_t batch, int strands, uint64_t* outer_trace,
                              uint64_t* loop_trace, uint64_t* public_out, unsigned long long* fail_out, uint32_t* mult_out) {
    try {
        CS& cs = *zkgl::cs_of(h);
        if (!cs.finalized_) throw std::runtime_error("zk_emu_resolve before finalize");
        if (cs.outer_.prog2.empty() && cs.loop_.prog2.empty()) throw std::runtime_error("no device programs");
        init_device_globals();
        g_last = std::make_unique<Run>(cs, batch);
        Run& r = *g_last;
        r.in[0] = outer_in; r.in_stride[0] = batch;
        r.in[1] = loop_in; r.in_stride[1] = (uint64_t)batch * std::max<uint32_t>(cs.limit_, 1);
        const bool st = strands != 0;
        r.phase(0, 0, st, true);                 // outer PRE
        if (cs.limit_) r.phase(1, 0, st, true);  // the loop
        r.phase(0, 1, st, true);                 // outer SIDE (independent of the loop; the device overlaps it)
        r.phase(0, 2, st, true);                 // outer POST
        const uint64_t so = ((uint64_t)batch + 63) / 64 * 64, sl = ((uint64_t)batch * cs.limit_ + 63) / 64 * 64;
        if (outer_trace) r.trace(0, outer_trace, so);
        if (loop_trace && cs.limit_) r.trace(1, loop_trace, sl);
        if (public_out)
            for (uint32_t i = 0; i < batch; ++i)
                for (size_t k = 0; k < cs.public_vars_.size(); ++k)
                    public_out[(size_t)i * cs.public_vars_.size() + k] = r.store[0][zkgeom::offset(r.geom[0], cs.outer_.var_slot[cs.public_vars_[k]], i)];
        if (fail_out) for (int i = 0; i < 16; ++i) fail_out[i] = r.fail[i];
        if (mult_out) std::memcpy(mult_out, r.mult.data(), r.mult.size() * sizeof(uint32_t));
        return 0;
    } catch (const std::exception& e) {
        zkgl::set_last_error(std::string("zk_emu_resolve: ") + e.what());
        return 1;
    }
}

// The checkers of the step on the stored values of the last zk_emu_resolve: the product's check kernels (k_check_prog, k_check_p2, k_check_links,
// k_check_stream, k_check_inputs — the same cut-out source) run lane by lane.  mode 0: the FUSED step's check program (what is left to the
// store: cprog_fused), mode 1: every relation (cprog + the Poseidon2 macro packets) = zk_cs_set_check_mode(ZK_CHECK_STORED).  fail_out: the
// device's six failure words (outer 0..2, loop 3..5; ~0 = none), merged with the witness kernels' fused flags in mode 0.  Returns 1 when the
// batch is rejected, 0 when it is accepted, -1 on error.
extern "C" int zk_emu_check(zk_cs* h, int mode, unsigned long long* fail_out) {
    try {
        if (!g_last) throw std::runtime_error("zk_emu_check before zk_emu_resolve");
        Run& r = *g_last;
        CS& cs = r.cs;
        unsigned long long f[16];
        for (auto& x : f) x = ~0ull;
        if (mode == 0) for (int i = 0; i < 6; ++i) f[i] = r.fail[i];
        for (int sc = 0; sc < 2; ++sc) {
            const Scope& s = sc ? cs.loop_ : cs.outer_;
            if (sc && !cs.limit_) continue;
            const uint32_t lanes = sc ? r.batch * r.limit : r.batch;
            const std::vector<uint32_t>& prog = mode == 0 ? s.cprog_fused : s.cprog;
            const std::vector<uint32_t>& chunks = mode == 0 ? s.cchunks_fused : s.cchunks;
            if (prog.empty() || chunks.size() < 2) throw std::runtime_error("this scope has no check program (lookup tables wider than the packets): not covered by the harness");
            std::vector<uint32_t> pp = padded(prog);
            zke::CheckProgDev cd;
            cd.cells = r.store[sc].data(); cd.n_cells = r.geom[sc]; cd.n_lanes = lanes;
            cd.prog = pp.data(); cd.chunk_tab = chunks.data(); cd.n_chunks = (uint32_t)chunks.size() - 1; cd.chunks_per_block = cd.n_chunks;
            cd.rowconsts = s.rowconsts.data(); cd.tables = cs.tdesc_host_.data(); cd.table_words = cs.table_words_host_.data();
            cd.fail = f + 3 * sc;
            emu::bdim = {(unsigned)zke::TPB, 1, 1};
            for (uint32_t lane = 0; lane < lanes; ++lane) {
                emu::bid = {lane / zke::TPB, 0, 0}; emu::tid = {lane % zke::TPB, 0, 0};
                if (chunks.back() > chunks.front()) zke::k_check_prog_t<false>(cd);
            }
            if (mode == 1 && !s.cmacros.empty()) {
                std::vector<uint32_t> mm(s.cmacros);
                mm.resize(mm.size() + 64, 0);
                zke::CheckP2Dev md;
                md.cells = r.store[sc].data(); md.n_cells = r.geom[sc]; md.n_lanes = lanes; md.macros = mm.data(); md.n_macros = (uint32_t)(s.cmacros.size() / 14);
                md.per_block = md.n_macros; md.fail = f + 3 * sc;
                for (uint32_t lane = 0; lane < lanes; ++lane) {
                    emu::bid = {lane / zke::TPB, 0, 0}; emu::tid = {lane % zke::TPB, 0, 0};
                    zke::k_check_p2(md);
                }
            }
            if (r.in[sc] && s.n_input_words)
                for (uint32_t lane = 0; lane < lanes; ++lane) {
                    emu::bid = {lane / 256, 0, 0}; emu::tid = {lane % 256, 0, 0}; emu::bdim = {256, 1, 1}; emu::gdim = {1, 1, 1};
                    zke::k_check_inputs(r.in[sc], s.n_input_words, lanes, r.in_stride[sc], f + 3 * sc);
                }
            emu::bdim = {(unsigned)zke::TPB, 1, 1};
        }
        if (cs.limit_) {
            const uint32_t lanes = r.batch * r.limit;
            emu::bdim = {(unsigned)zke::TPB, 1, 1};
            for (uint32_t lane = 0; lane < lanes; ++lane) {
                emu::bid = {lane / zke::TPB, 0, 0}; emu::tid = {lane % zke::TPB, 0, 0};
                zke::k_check_links_t<false>(r.store[1].data(), r.geom[1], lanes, r.limit, r.store[0].data(), r.geom[0], cs.links_store_.data(), (uint32_t)cs.links_store_.size(), f + 3);
            }
            for (size_t i = 0; i < cs.streams_store_.size(); ++i) {
                const auto& sr = cs.streams_store_[i];
                const uint64_t n = (uint64_t)r.batch * sr.n_total;
                for (uint64_t t = 0; t < n; ++t) {
                    emu::bid = {(unsigned)(t / zke::TPB), 0, 0}; emu::tid = {(unsigned)(t % zke::TPB), 0, 0};
                    zke::k_check_stream_t<false>(r.store[1].data(), r.geom[1], r.batch, r.limit, sr.a.data(), (uint32_t)sr.a.size(), sr.b.data(), (uint32_t)sr.b.size(), sr.n_total, (uint32_t)i, f + 3);
                }
            }
        }
        bool rejected = false;
        for (int i = 0; i < 6; ++i) rejected |= f[i] != ~0ull;
        if (fail_out) for (int i = 0; i < 6; ++i) fail_out[i] = f[i];
        return rejected ? 1 : 0;
    } catch (const std::exception& e) {
        zkgl::set_last_error(std::string("zk_emu_check: ") + e.what());
        return -1;
    }
}

// debugging aid: after zk_emu_resolve (the run is kept), the first op of a scope, in program order, one of whose outputs differs from the expected
// trace (same layout as the traces above) in lane `lane`: prints the op, its operand values and the outputs got / wanted to stderr; returns its index or -1
extern "C" int zk_emu_first_bad_op(zk_cs* h, int loop_scope, const uint64_t* want_trace, uint64_t stride, uint32_t lane) {
    if (!g_last) return -2;
    Run& r = *g_last;
    const Scope& s = loop_scope ? r.cs.loop_ : r.cs.outer_;
    const int sc = loop_scope ? 1 : 0;
    auto val = [&](uint32_t v) { return r.store[sc][zkgeom::offset(r.geom[sc], s.var_slot[v], lane)]; };
    for (size_t oi = 0; oi < s.ops.size(); ++oi) {
        const zkgl::OpRec& op = s.ops[oi];
        if (op.seed_only) continue;
        for (uint32_t ov : op.outs) {
            const uint64_t want = want_trace[(uint64_t)s.var_cells[ov][0] * stride + lane];
            if (val(ov) == want) continue;
            fprintf(stderr, "[emu] first bad op %zu of %zu: opcode %u a %u b %u, %zu ins, %zu outs\n", oi, s.ops.size(), op.opcode, op.a, op.b, op.ins.size(), op.outs.size());
            for (size_t k = 0; k < op.ins.size() && k < 16; ++k)
                if (op.ins[k].kind == zkgl::Operand::VAR) fprintf(stderr, "   in %zu: var %u slot %u = %llu (trace wants %llu)\n", k, op.ins[k].idx, s.var_slot[op.ins[k].idx], (unsigned long long)val(op.ins[k].idx),
                                                                   (unsigned long long)want_trace[(uint64_t)s.var_cells[op.ins[k].idx][0] * stride + lane]);
                else fprintf(stderr, "   in %zu: kind %d idx %u\n", k, (int)op.ins[k].kind, op.ins[k].idx);
            for (size_t k = 0; k < op.outs.size() && k < 8; ++k)
                fprintf(stderr, "   out %zu: var %u slot %u got %llu want %llu\n", k, op.outs[k], s.var_slot[op.outs[k]], (unsigned long long)val(op.outs[k]),
                        (unsigned long long)want_trace[(uint64_t)s.var_cells[op.outs[k]][0] * stride + lane]);
            return (int)oi;
        }
    }
    return -1;
}

// statistics aid (DESIGN: what the batched inversions can gain): over the loop scope of the last zk_emu_resolve, per 64-lane wavefront, how many of the
// ZK_OP_ISZERO ops leave the small-inverse table (some lane holds |x| >= INV_SMALL_N: the whole wavefront runs the 72-multiplication chain).
// out[0] = zero-check ops per lane, out[1] = wavefronts, out[2] = sum over wavefronts of the ops that take the chain, out[3] = the same per LANE (a lane
// alone: what a one-lane wavefront would do)
extern "C" int zk_emu_iszero_stats(zk_cs* h, uint64_t out[4]) {
    if (!g_last) return -1;
    Run& r = *g_last;
    const Scope& s = r.cs.loop_;
    const uint64_t lanes = (uint64_t)r.batch * r.limit;
    out[0] = out[1] = out[2] = out[3] = 0;
    out[1] = (lanes + 63) / 64;
    for (auto& op : s.ops) {
        if (op.seed_only || op.opcode != ZK_OP_ISZERO || op.ins[0].kind != zkgl::Operand::VAR) continue;
        ++out[0];
        for (uint64_t w0 = 0; w0 < lanes; w0 += 64) {
            bool slow = false;
            for (uint64_t l = w0; l < std::min(lanes, w0 + 64); ++l) {
                const uint64_t x = r.store[1][zkgeom::offset(r.geom[1], s.var_slot[op.ins[0].idx], l)];
                const bool big = x >= p2::INV_SMALL_N && gl::P - x >= p2::INV_SMALL_N;
                slow |= big; out[3] += big;
            }
            out[2] += slow;
        }
    }
    return 0;
}

// the same for the execute-gated witness-only permutations (ZK_OP_POSEIDON2 a = 1: a wavefront runs the permutation as soon as ONE of its lanes has
// the flag on).  out[0] = gated ops per lane, out[1] = wavefronts, out[2] = sum over wavefronts of the ops it runs, out[3] = sum over lanes of the ops
// whose flag is on (what the lanes actually need)
extern "C" int zk_emu_gated_p2_stats(zk_cs* h, uint64_t out[4]) {
    if (!g_last) return -1;
    Run& r = *g_last;
    const Scope& s = r.cs.loop_;
    const uint64_t lanes = (uint64_t)r.batch * r.limit;
    out[0] = out[2] = out[3] = 0;
    out[1] = (lanes + 63) / 64;
    for (auto& op : s.ops) {
        if (op.seed_only || op.opcode != ZK_OP_POSEIDON2 || op.a != 1 || op.ins.size() != 13) continue;
        ++out[0];
        for (uint64_t w0 = 0; w0 < lanes; w0 += 64) {
            bool any = false;
            for (uint64_t l = w0; l < std::min(lanes, w0 + 64); ++l) {
                const bool on = r.store[1][zkgeom::offset(r.geom[1], s.var_slot[op.ins[12].idx], l)] != 0;
                any |= on; out[3] += on;
            }
            out[2] += any;
        }
    }
    return 0;
}

// ... and what merging the gated permutations of a dependency level would leave: per wavefront and level, the rounds = the largest number of members
// any ONE lane has on.  out[0] = levels, out[1] = wavefronts, out[2] = sum over wavefronts of the rounds (the permutations a merged kernel runs)
extern "C" int zk_emu_gated_p2_merged_rounds(zk_cs* h, uint64_t out[3]) {
    if (!g_last) return -1;
    Run& r = *g_last;
    const Scope& s = r.cs.loop_;
    const uint64_t lanes = (uint64_t)r.batch * r.limit;
    std::vector<int> vlvl(s.n_vars, 0);
    std::map<int, std::vector<uint32_t>> flags_of_level;
    for (auto& op : s.ops) {
        if (op.seed_only) continue;
        int in = 0;
        for (auto& x : op.ins) if (x.kind == zkgl::Operand::VAR) in = std::max(in, vlvl[x.idx]);
        const bool gated = op.opcode == ZK_OP_POSEIDON2 && op.a == 1;
        for (uint32_t ov : op.outs) vlvl[ov] = gated ? in + 1 : in;
        if (gated) flags_of_level[in + 1].push_back(op.ins[12].idx);
    }
    out[0] = flags_of_level.size(); out[1] = (lanes + 63) / 64; out[2] = 0;
    for (auto& kv : flags_of_level)
        for (uint64_t w0 = 0; w0 < lanes; w0 += 64) {
            uint32_t rounds = 0;
            for (uint64_t l = w0; l < std::min(lanes, w0 + 64); ++l) {
                uint32_t on = 0;
                for (uint32_t fv : kv.second) on += r.store[1][zkgeom::offset(r.geom[1], s.var_slot[fv], l)] != 0;
                rounds = std::max(rounds, on);
            }
            out[2] += rounds;
        }
    return 0;
}

// analysis aid: the dependency levels of the gated witness-only permutations of the loop scope (level = the longest chain of gated permutations a site
// depends on, through any path of ops).  Prints one line per site to stderr; returns the number of levels.
extern "C" int zk_emu_gated_p2_levels(zk_cs* h) {
    CS& cs = *zkgl::cs_of(h);
    const Scope& s = cs.loop_;
    const auto& ops = cs.loop_ops_recorded_.empty() ? s.ops : cs.loop_ops_recorded_;   // recording order
    std::vector<int> lvl(s.n_vars, 0);   // per variable: the number of gated permutations on the longest path that produces it
    int n_levels = 0, site = 0;
    for (size_t oi = 0; oi < ops.size(); ++oi) {
        const zkgl::OpRec& op = ops[oi];
        if (op.seed_only) continue;
        int in = 0;
        for (auto& x : op.ins) if (x.kind == zkgl::Operand::VAR) in = std::max(in, lvl[x.idx]);
        const bool gated = op.opcode == ZK_OP_POSEIDON2 && op.a == 1;
        const int out = gated ? in + 1 : in;
        for (uint32_t ov : op.outs) lvl[ov] = out;
        if (gated) { fprintf(stderr, "[emu] gated permutation %2d (op %zu of %zu): level %d, flag var %u\n", site++, oi, ops.size(), out, op.ins[12].idx); n_levels = std::max(n_levels, out); }
    }
    return n_levels;
}

// analysis aid (round 6's narrow store): per opcode of the loop scope, how many outputs the census bounds by 2^8 / 2^32 (CS::bound_values) and
// what packing WITHIN one op would save (k byte outputs of an op -> ceil(k / 8) slots; k narrow outputs -> ceil(k / 2)): an op's outputs are
// the next consecutive slots and a packed slot has to be complete when the op ends (the next op may read it).
extern "C" int zk_emu_narrow_stats(zk_cs* h, int outer) {
    CS& cs = *zkgl::cs_of(h);
    const Scope& s = outer ? cs.outer_ : cs.loop_;
    if (s.value_class.size() != s.n_vars) return -1;
    struct Row { uint64_t ops = 0, outs = 0, b8 = 0, b32 = 0, save8 = 0, save32 = 0, reads8 = 0, reads = 0; } rows[256], tot;
    for (auto& op : s.ops) {
        if (op.seed_only) continue;
        Row& r = rows[op.opcode];
        uint32_t k8 = 0, k32 = 0;
        for (uint32_t v : op.outs) { k8 += s.value_class[v] == 2; k32 += s.value_class[v] >= 1; }
        r.ops++; r.outs += op.outs.size(); r.b8 += k8; r.b32 += k32;
        if (k8 >= 2) r.save8 += k8 - (k8 + 7) / 8;
        if (k32 >= 2) r.save32 += (k32 - (k32 + 1) / 2);
        for (auto& x : op.ins) if (x.kind == zkgl::Operand::VAR) { r.reads++; r.reads8 += s.value_class[x.idx] == 2; }
    }
    fprintf(stderr, "[emu] %s scope: %u variables, %u slots\n[emu] %-6s %8s %8s %8s %8s %10s %10s %8s %8s\n", outer ? "outer" : "loop", s.n_vars, s.n_slots,
            "opcode", "ops", "outs", "<2^8", "<2^32", "save(8in1)", "save(2in1)", "reads", "reads<2^8");
    for (int o = 0; o < 256; ++o) {
        const Row& r = rows[o];
        if (!r.ops) continue;
        fprintf(stderr, "[emu] %-6d %8llu %8llu %8llu %8llu %10llu %10llu %8llu %8llu\n", o, (unsigned long long)r.ops, (unsigned long long)r.outs, (unsigned long long)r.b8,
                (unsigned long long)r.b32, (unsigned long long)r.save8, (unsigned long long)r.save32, (unsigned long long)r.reads, (unsigned long long)r.reads8);
        tot.ops += r.ops; tot.outs += r.outs; tot.b8 += r.b8; tot.b32 += r.b32; tot.save8 += r.save8; tot.save32 += r.save32; tot.reads += r.reads; tot.reads8 += r.reads8;
    }
    fprintf(stderr, "[emu] %-6s %8llu %8llu %8llu %8llu %10llu %10llu %8llu %8llu\n", "all", (unsigned long long)tot.ops, (unsigned long long)tot.outs, (unsigned long long)tot.b8,
            (unsigned long long)tot.b32, (unsigned long long)tot.save8, (unsigned long long)tot.save32, (unsigned long long)tot.reads, (unsigned long long)tot.reads8);
    return 0;
}

extern "C" void zk_emu_sizes(zk_cs* h, uint32_t batch, uint64_t out[6]) {   // outer n_cells, outer stride, loop n_cells, loop stride, n_public, total_table_rows
    CS& cs = *zkgl::cs_of(h);
    out[0] = cs.outer_.n_cells; out[1] = ((uint64_t)batch + 63) / 64 * 64;
    out[2] = cs.loop_.n_cells; out[3] = ((uint64_t)batch * cs.limit_ + 63) / 64 * 64;
    out[4] = cs.public_vars_.size(); out[5] = cs.total_table_rows_;
}
