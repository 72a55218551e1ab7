"""repo root, after tools/profile_r4.sh: gpurun_out/pmc_r4_{fetch,write}.txt + r4_bench*.json + r4_kernel_trace.md -> the `traffic` record
(profiles/pmc_r4*.json).  Counters are KiB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 counts half), WRITE_SIZE as reported.
usage: python tools/pmc_json.py <tag> [kernel] > profiles/pmc_<tag>.json        (tag: suffix for the note, e.g. r4_final; kernel: zke::k_witness_loop (default) or
zke::k_witness_loop_narrow for a run made with bench.py --narrow-store).  When gpurun_out/rprobe.json + gpurun_out/pmc_rprobe.txt exist (tools/rprobe.hip under
FETCH_SIZE: a known byte count read in this kernel's access pattern), the measured FETCH_SIZE factor of 8 B/lane loads is reported beside the guide's x2."""
import json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "zke::k_witness_loop"
G = "gpurun_out/"
RT = tag.split("_")[0]   # file prefix written by tools/profile_tag.sh (TAG=r5 -> r5_*)


def mean(path, kernel, counter):
    for line in open(path):
        f = line.split()
        if f and f[0] == kernel and counter in f[1:2]:    # exact kernel name (k_witness_loop is a prefix of k_witness_loop_narrow)
            return float(re.search(r"mean=([0-9.e+]+)", line).group(1))
    raise SystemExit(f"{kernel} {counter} not in {path}")


fetch = 2 * 1024 * mean(G + "pmc_" + RT + "_fetch.txt", KERNEL, "FETCH_SIZE")
write = 1024 * mean(G + "pmc_" + RT + "_write.txt", KERNEL, "WRITE_SIZE")
bench = json.loads(open(G + RT + "_bench.json").read().strip().splitlines()[-1])
under = json.loads(open(G + RT + "_bench_under_rocprof.json").read().strip().splitlines()[-1])
roof = bench["roofline"]
alg = roof["algorithmic_bytes_per_launch"] if "algorithmic_bytes_per_launch" in roof else roof["achieved"] * 1e9 * roof["avg_launch_ms"] * 1e-3
kt = None
for line in open(G + RT + "_kernel_trace.md"):
    if re.search(r"(^|[^A-Za-z0-9_])" + re.escape(KERNEL.split("::")[-1]) + r"([^A-Za-z0-9_]|$)", line):
        nums = re.findall(r"[0-9]+\.[0-9]+", line)
        kt = line.strip()
        break
def opt(path, kernel, counter):
    try:
        return mean(path, kernel, counter)
    except (SystemExit, OSError):
        return None


# VALU-issue roofline of the same kernel (profiles/r3_loop_probe.md §2 method): SQ counters are per shader-engine slice (n = 32 x launches),
# SQ_ACTIVE_INST_VALU counts quad-cycles per slice: busy = ACTIVE_INST_VALU * 4 / (SIMDs per slice = 32) / GRBM_GUI_ACTIVE
K = KERNEL
valu = {"SQ_INSTS_VALU": opt(G + "pmc_" + RT + "_valu.txt", K, "SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": opt(G + "pmc_" + RT + "_valu.txt", K, "SQ_ACTIVE_INST_VALU"),
        "SQ_BUSY_CYCLES": opt(G + "pmc_" + RT + "_valu.txt", K, "SQ_BUSY_CYCLES"), "GRBM_GUI_ACTIVE": opt(G + "pmc_" + RT + "_valu.txt", K, "GRBM_GUI_ACTIVE"),
        "SQ_INSTS_SALU": opt(G + "pmc_" + RT + "_salu.txt", K, "SQ_INSTS_SALU"), "SQ_WAVES": opt(G + "pmc_" + RT + "_salu.txt", K, "SQ_WAVES"),
        "SQ_INSTS_VMEM_RD": opt(G + "pmc_" + RT + "_salu.txt", K, "SQ_INSTS_VMEM_RD"), "SQ_INSTS_VMEM_WR": opt(G + "pmc_" + RT + "_salu.txt", K, "SQ_INSTS_VMEM_WR")}
valu_busy = None
if valu["SQ_ACTIVE_INST_VALU"] and valu["GRBM_GUI_ACTIVE"]:
    valu_busy = valu["SQ_ACTIVE_INST_VALU"] * 4 / 32 / valu["GRBM_GUI_ACTIVE"]
per_wave = None
if valu["SQ_INSTS_VALU"] and valu["SQ_WAVES"]:
    per_wave = {"valu": valu["SQ_INSTS_VALU"] / valu["SQ_WAVES"], "salu": (valu["SQ_INSTS_SALU"] or 0) / valu["SQ_WAVES"],
                "vmem_rd": (valu["SQ_INSTS_VMEM_RD"] or 0) / valu["SQ_WAVES"], "vmem_wr": (valu["SQ_INSTS_VMEM_WR"] or 0) / valu["SQ_WAVES"]}
# measured FETCH_SIZE factor for this kernel's access pattern (tools/rprobe.hip: known bytes / counter), when the probe ran in the same call
fetch_cal = None
try:
    probe = {}
    for line in open(G + "rprobe.json"):
        d = json.loads(line)
        probe[d["kernel"]] = d
    cal = {}
    for line in open(G + "pmc_rprobe.txt"):
        f = line.split()
        if "FETCH_SIZE" in f:
            k = "".join(f[:f.index("FETCH_SIZE")]).replace("void", "")   # ("void k_read<0>" for a templated kernel)
            m = re.search(r"mean=([0-9.e+]+)", line)
            for name, d in probe.items():
                if name.replace(" ", "") in k.replace(" ", ""):
                    cal[name] = {"pattern": d["pattern"], "known_bytes": d["known_bytes_per_launch"], "FETCH_SIZE_KiB_mean": float(m.group(1)),
                                 "factor_known_over_counted": d["known_bytes_per_launch"] / (1024 * float(m.group(1)))}
    if cal:
        f8 = cal.get("k_read<0>", {}).get("factor_known_over_counted")
        fetch_cal = {"by_pattern": cal, "factor_8B_per_lane": f8,
                     "fetch_bytes_with_measured_factor": None if f8 is None else f8 * fetch / 2,
                     "traffic_over_algorithmic_with_measured_factor": None if f8 is None else (f8 * fetch / 2 + write) / alg}
except OSError:
    pass
out = {"fetch_correction_measured": fetch_cal, "valu_issue": {"counters_per_slice_mean": valu, "valu_busy_frac": valu_busy, "instructions_per_wavefront": per_wave,
                      "how": "VALUBusy = SQ_ACTIVE_INST_VALU x 4 / 32 SIMDs per slice / GRBM_GUI_ACTIVE (profiles/r3_loop_probe.md §2); the peak of this roofline is "
                             "VALUBusy = 1: one wave-instruction issued per SIMD every cycle it can take one"},
       "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/profile_r4.sh -> tools/pmc_pass.sh), bench.py --headline-only at batch 384, {tag} code (SELECT flags "
               "as bit planes); counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count); WRITE_SIZE as reported",
       "batch": 384, "kernel": KERNEL, "algorithmic_bytes_per_launch": alg, "fetch_bytes_x2": fetch, "write_bytes_reported": write,
       "hbm_traffic_bytes_per_launch": fetch + write, "traffic_over_algorithmic": (fetch + write) / alg,
       "k_witness_loop_avg_ms_bench_same_box_no_profiler": roof["avg_launch_ms"],
       "k_witness_loop_avg_ms_bench_under_the_same_rocprof_run": under["roofline"]["avg_launch_ms"],
       "kernel_trace_line": kt,
       "hbm_busy_TBps_same_box": (fetch + write) / (roof["avg_launch_ms"] * 1e-3) / 1e12}
print(json.dumps(out, indent=1))
