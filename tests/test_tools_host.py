"""Host tests of the evidence tooling (no GPU): the scripts a GPU call runs once must not fail on their own parsing.
  tools/isa_diff.py  — kernel-by-kernel comparison of two device-only assemblies (labels renumbered, comments dropped)
  tools/pmc_json.py  — the `traffic` / VALU-issue record from the PMC passes, kernel selected by exact name, the measured FETCH_SIZE factor of tools/rprobe.hip"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ASM_A = """
\t.text
\t.globl\t_Z1kPm
_Z1kPm:                                  ; @_Z1kPm
; %bb.0:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\ts_and_b32 s2, s3, 0xffffff
\ts_cbranch_scc1 .LBB0_2
.LBB0_1:
\tv_mov_b32_e32 v0, 0                     ; a comment
.LBB0_2:
\ts_endpgm
.Lfunc_end0:
_Z1jPm:                                  ; @_Z1jPm
\ts_endpgm
.Lfunc_end1:
"""


def test_isa_diff_ignores_labels_and_comments_and_sees_an_instruction(tmp_path):
    a, b, c = tmp_path / "a.s", tmp_path / "b.s", tmp_path / "c.s"
    a.write_text(ASM_A)
    b.write_text(ASM_A.replace(".LBB0_", ".LBB7_").replace("; a comment", "; another"))      # other label numbers, other comments: the same code
    c.write_text(ASM_A.replace("0xffffff", "0x7fffff"))
    run = lambda x, y: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_diff.py"), str(x), str(y)], capture_output=True, text=True, check=True).stdout
    out = run(a, b)
    assert out.count("identical") == 2, out
    out = run(a, c)
    assert "_Z1kPm: 2 differing lines" in out and "_Z1jPm: identical" in out, out


def test_pmc_json_selects_the_kernel_by_exact_name_and_reads_the_calibration(tmp_path):
    g = tmp_path / "gpurun_out"
    g.mkdir()
    line = "{:40s} {:28s} n={:3d} mean={:.4g} max={:.4g}\n".format
    (g / "pmc_rX_fetch.txt").write_text(line("zke::k_witness_loop_narrow", "FETCH_SIZE", 4, 2.0e7, 2.1e7) + line("zke::k_witness_loop", "FETCH_SIZE", 4, 4.0e7, 4.1e7))
    (g / "pmc_rX_write.txt").write_text(line("zke::k_witness_loop", "WRITE_SIZE", 4, 1.2e8, 1.2e8) + line("zke::k_witness_loop_narrow", "WRITE_SIZE", 4, 9.0e7, 9.0e7))
    (g / "pmc_rX_valu.txt").write_text(line("zke::k_witness_loop", "SQ_INSTS_VALU", 32, 4.4e8, 4.5e8) + line("zke::k_witness_loop", "SQ_ACTIVE_INST_VALU", 32, 4.36e8, 4.4e8)
                                       + line("zke::k_witness_loop", "GRBM_GUI_ACTIVE", 32, 8.18e7, 8.2e7))
    (g / "pmc_rX_salu.txt").write_text(line("zke::k_witness_loop", "SQ_WAVES", 32, 447, 447))
    bench = {"roofline": {"algorithmic_bytes_per_launch": 130489712640, "avg_launch_ms": 38.0, "achieved": 3400.0}}
    (g / "rX_bench.json").write_text(json.dumps(bench) + "\n")
    (g / "rX_bench_under_rocprof.json").write_text(json.dumps(bench) + "\n")
    (g / "rX_kernel_trace.md").write_text("| zke::k_witness_loop_narrow | 6 | 1 | 30.1 |\n| zke::k_witness_loop | 6 | 233.683 | 38.9471 |\n")
    (g / "rprobe.json").write_text("".join(json.dumps({"kernel": f"k_read<{m}>", "pattern": "p", "known_bytes_per_launch": 8589934592.0, "ms": 2.0, "GBps": 4000}) + "\n" for m in range(4)))
    (g / "pmc_rprobe.txt").write_text(line("void k_read<0>", "FETCH_SIZE", 16, 8388608.0, 8.4e6) + line("void k_read<1>", "FETCH_SIZE", 16, 4194304.0, 4.2e6))
    run = lambda *a: json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_json.py"), *a], cwd=tmp_path, capture_output=True, text=True, check=True).stdout)
    d = run("rX")
    assert d["kernel"] == "zke::k_witness_loop" and d["fetch_bytes_x2"] == 2 * 1024 * 4.0e7 and d["write_bytes_reported"] == 1024 * 1.2e8
    assert "k_witness_loop |" in d["kernel_trace_line"] and "narrow" not in d["kernel_trace_line"]
    assert abs(d["valu_issue"]["valu_busy_frac"] - 4.36e8 * 4 / 32 / 8.18e7) < 1e-9
    cal = d["fetch_correction_measured"]
    assert abs(cal["factor_8B_per_lane"] - 1.0) < 1e-3 and abs(cal["by_pattern"]["k_read<1>"]["factor_known_over_counted"] - 2.0) < 2e-3      # 8 B/lane counted in full, 16 B/lane by half
    assert abs(cal["traffic_over_algorithmic_with_measured_factor"] - (1024 * 4.0e7 + 1024 * 1.2e8) / 130489712640) < 1e-3
    n = run("rX", "zke::k_witness_loop_narrow")
    assert n["kernel"].endswith("_narrow") and n["fetch_bytes_x2"] == 2 * 1024 * 2.0e7 and n["write_bytes_reported"] == 1024 * 9.0e7 and "narrow" in n["kernel_trace_line"]
