"""Kernel resource usage from `hipcc -Rpass-analysis=kernel-resource-usage` remark dumps: a table of VGPRs / SGPRs / scratch / spills / LDS / occupancy
per kernel, for one dump or for two side by side (which kernels' register allocation a change touched — no GPU needed).
usage: python tools/resource_usage.py <remarks.txt> [<other_remarks.txt>]
make a dump:  (cd era-zkevm_circuits_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c zkgl_device.hip -o /tmp/x.o 2> remarks.txt)"""
import re, subprocess, sys


def parse(path):
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            out[cur][m.group(1).strip()] = m.group(2)
    return out


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        d = d[5:] if d.startswith("void ") else d
        return d.split("(")[0]
    except Exception:
        return n


KEYS = ["VGPRs", "TotalSGPRs", "ScratchSize", "VGPRs Spill", "SGPRs Spill", "LDS Size", "Occupancy"]
a = parse(sys.argv[1])
b = parse(sys.argv[2]) if len(sys.argv) > 2 else None
print("| kernel | " + " | ".join(KEYS) + (" | changed vs the other dump |" if b else " |"))
print("|---|" + "---|" * (len(KEYS) + (1 if b else 0)))
for k in sorted(a, key=demangle):
    row = [a[k].get(x, "") for x in KEYS]
    tail = ""
    if b is not None:
        if k not in b:
            tail = " new |"
        else:
            diff = [f"{x}: {b[k].get(x, '')} -> {a[k].get(x, '')}" for x in KEYS if b[k].get(x, "") != a[k].get(x, "")]
            tail = " " + ("; ".join(diff) if diff else "same") + " |"
    print("| `" + demangle(k) + "` | " + " | ".join(row) + " |" + tail)
