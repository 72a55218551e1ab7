"""repo root, after tools/profile_r4.sh: gpurun_out/pmc_r4_{fetch,write}.txt + r4_bench*.json + r4_kernel_trace.md -> the `traffic` record
(profiles/pmc_r4*.json).  Counters are KiB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 counts half), WRITE_SIZE as reported.
usage: python tools/pmc_json.py <tag> > profiles/pmc_<tag>.json        (tag: suffix for the note, e.g. r4_final)"""
import json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
G = "gpurun_out/"
RT = tag.split("_")[0]   # file prefix written by tools/profile_tag.sh (TAG=r5 -> r5_*)


def mean(path, kernel, counter):
    for line in open(path):
        if line.startswith(kernel) and counter in line:
            return float(re.search(r"mean=([0-9.e+]+)", line).group(1))
    raise SystemExit(f"{kernel} {counter} not in {path}")


fetch = 2 * 1024 * mean(G + "pmc_" + RT + "_fetch.txt", "zke::k_witness_loop", "FETCH_SIZE")
write = 1024 * mean(G + "pmc_" + RT + "_write.txt", "zke::k_witness_loop", "WRITE_SIZE")
bench = json.loads(open(G + RT + "_bench.json").read().strip().splitlines()[-1])
under = json.loads(open(G + RT + "_bench_under_rocprof.json").read().strip().splitlines()[-1])
roof = bench["roofline"]
alg = roof["algorithmic_bytes_per_launch"] if "algorithmic_bytes_per_launch" in roof else roof["achieved"] * 1e9 * roof["avg_launch_ms"] * 1e-3
kt = None
for line in open(G + RT + "_kernel_trace.md"):
    if "k_witness_loop" in line:
        nums = re.findall(r"[0-9]+\.[0-9]+", line)
        kt = line.strip()
        break
out = {"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/profile_r4.sh -> tools/pmc_pass.sh), bench.py --headline-only at batch 384, {tag} code (SELECT flags "
               "as bit planes); counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count); WRITE_SIZE as reported",
       "batch": 384, "kernel": "zke::k_witness_loop", "algorithmic_bytes_per_launch": alg, "fetch_bytes_x2": fetch, "write_bytes_reported": write,
       "hbm_traffic_bytes_per_launch": fetch + write, "traffic_over_algorithmic": (fetch + write) / alg,
       "k_witness_loop_avg_ms_bench_same_box_no_profiler": roof["avg_launch_ms"],
       "k_witness_loop_avg_ms_bench_under_the_same_rocprof_run": under["roofline"]["avg_launch_ms"],
       "kernel_trace_line": kt,
       "hbm_busy_TBps_same_box": (fetch + write) / (roof["avg_launch_ms"] * 1e-3) / 1e12}
print(json.dumps(out, indent=1))
