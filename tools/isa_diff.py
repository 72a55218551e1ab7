"""tools/isa_diff.py <old.s> <new.s> [kernel-substring ...] — are the kernels of two device-only assemblies (hipcc --cuda-device-only -S) the same code?
Per kernel: the instruction sequence (mnemonic + operands; labels renumbered in order of appearance, comments and directives dropped) of both files is compared;
prints `identical`, or the number of differing lines with the first few.  No GPU needed.  Round 6 used it to show that the kernels a device has measured did not
change under the narrow store / the variant clean-up (profiles/r6_resource_usage.md)."""
import re, sys, difflib


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"\n(_Z[\w]+):[^\n]*\n", txt):
        name = m.group(1)
        j = txt.find(".Lfunc_end", m.end())
        if j < 0:
            continue
        body, labels = [], {}
        for l in txt[m.end():j].split("\n"):
            l = l.split(";")[0].split("//")[0].strip()
            if not l or l.startswith("."):
                if l.endswith(":") and l.startswith(".L"):
                    labels.setdefault(l[:-1], f"L{len(labels)}")
                    body.append(labels[l[:-1]] + ":")
                continue
            body.append(l)
        # renumber label references in order of first appearance (definitions above; references may precede them)
        def ren(mm):
            return labels.setdefault(mm.group(0), f"L{len(labels)}")
        out[name] = [re.sub(r"\.L[\w$]+", ren, l) for l in body]
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    want = sys.argv[3:]
    for name in sorted(a):
        if want and not any(w in name for w in want):
            continue
        if name not in b:
            print(f"{name}: only in {sys.argv[1]}")
            continue
        if a[name] == b[name]:
            print(f"{name}: identical ({len(a[name])} lines)")
        else:
            d = [l for l in difflib.unified_diff(a[name], b[name], lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
            print(f"{name}: {len(d)} differing lines of {len(a[name])} / {len(b[name])}; first: {d[:6]}")
    for name in sorted(set(b) - set(a)):
        if not want or any(w in name for w in want):
            print(f"{name}: only in {sys.argv[2]}")


if __name__ == "__main__":
    main()
