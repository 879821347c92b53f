#!/usr/bin/env python3
"""Compare two tools/isa_dump.sh listings kernel by kernel (bodies only: a kernel's address does not matter).

    python3 tools/isa_compare.py before.s after.s     ->  which kernels are identical / differ / exist on one side only; exit 1 on a difference
"""
import re
import sys


def kernels(path):
    out, name = {}, None
    for line in open(path):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            out[name].append(re.sub(r"^\s*[0-9a-f]+:\s*", "", line.strip()))      # (no address column in these dumps, but be safe)
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = [k for k in a if k in b and a[k] == b[k]]
diff = [k for k in a if k in b and a[k] != b[k]]
print(f"{len(same)} kernels identical, {len(diff)} differ, {len(set(a) - set(b))} only in {sys.argv[1]}, {len(set(b) - set(a))} only in {sys.argv[2]}")
for k in diff:
    n = next((i for i, (x, y) in enumerate(zip(a[k], b[k])) if x != y), min(len(a[k]), len(b[k])))
    print(f"  DIFFERS {k}: {len(a[k])} vs {len(b[k])} instructions, first difference at {n}")
for k in sorted(set(a) - set(b)):
    print("  only before:", k)
for k in sorted(set(b) - set(a)):
    print("  only after: ", k)
sys.exit(1 if diff else 0)
