#!/usr/bin/env python3
"""The row-tile epilogue of the fused bf16 kernels, instruction by instruction class ("put the 80-instruction row-tile epilogue
on the table").  Disassembles nca_kernels_bf16.hip for gfx950, takes one kernel, splits its instruction stream at every 16th MFMA (one row
tile = 16 x v_mfma_f32_32x32x16_bf16: 8 k-steps x 2 column tiles) and prints, per row tile, the count of every class of instruction between
that row tile's first MFMA and the next row tile's -- the MFMA block and the epilogue of the 32 x 64 output tile it produced.

    python3 tools/isa_epilogue_table.py ['nca_fused_bf16<128, 2, true, true>'] > profiles/r06_epilogue_isa_table.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [  # (class, regex on the mnemonic, what it is there for)
    ("mfma", r"v_mfma", "the contraction: 8 k-steps x 2 column tiles"),
    ("cvt_pk_bf16", r"v_cvt_pk_bf16_f32", "f32 accumulators -> packed bf16 pairs = the next layer's B operand (2 values per instruction: 32 values x 2 column tiles)"),
    ("pk_max (ReLU)", r"v_pk_max_i16", "ReLU on the packed pair (one integer max per pair)"),
    ("pk_min (mask)", r"v_pk_min_u16", "0 / 1 per half: the ReLU mask bits of the pair (the backward from the store recomputes nothing: it reads them)"),
    ("lshl_or (mask)", r"v_lshl_or_b32", "gathers the 16 mask bits of a (row tile, column tile) into one field: 7 per 8 words"),
    ("cvt e4m3 / e5m2", r"v_cvt_scalef32_pk_(fp8|bf8)", "the staged copy for the weight-gradient kernel: packed bf16 -> 8-bit, 2 values per instruction"),
    ("pk_mul (keep)", r"v_pk_mul_lo_u16", "backward: zeroes the masked halves of a packed pair (x 0 / 1)"),
    ("shift / and (flags)", r"v_(lshrrev_b32|and_b32|and_or_b32|bfe_u32)", "backward: the pair's two mask bits out of the field (bits k and 16 + k)"),
    ("v_or / v_mov / other VALU", r"v_", "field bookkeeping (mw |= field << 8), address arithmetic, register copies"),
    ("ds_read", r"ds_read", "A fragments of the next row tile (8 x 1 KiB per wave) + the bias row of the accumulators"),
    ("global_store", r"global_store|buffer_store", "the two 1 KiB stores of the staged block (one per column tile)"),
    ("global_load", r"global_load|buffer_load", ""),
    ("s_nop", r"s_nop", "MFMA -> VALU / VALU -> MFMA wait states the hazard recogniser inserts"),
    ("s_waitcnt", r"s_waitcnt", "LDS reads of the A ring"),
    ("scalar other", r"s_", "loop / address bookkeeping on the scalar ALU"),
]


def classify(op):
    for name, rx, _ in CLASSES:
        if re.match(rx, op):
            return name
    return "other"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "nca_fused_bf16<128, 2, true, true>"
    src = os.path.join(ROOT, "nerf-ca_amd", "csrc", "nca_kernels_bf16.hip")
    with tempfile.TemporaryDirectory() as t:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                        "--cuda-device-only", "-c", src, "-o", f"{t}/x.co"], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/x.co", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={t}/x.elf"], check=True)
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", f"{t}/x.elf"], check=True, capture_output=True, text=True).stdout
    ops, on = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            on = want in m.group(1)
            continue
        if on:
            m = re.match(r"^\s+(\S+)", line)
            if m and not line.strip().startswith("//"):
                ops.append(m.group(1))
    if not ops:
        sys.exit(f"no kernel matching {want!r}")
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    groups = [mf[i:i + 16] for i in range(0, len(mf), 16)]
    print(f"{want}: {len(ops)} instructions, {len(mf)} MFMAs = {len(groups)} row tiles of 16 (static code: both the e4m3-staging layers and the last layer of a net appear once per")
    print("unrolled copy).  Per row tile: instructions from its first MFMA to the next row tile's first MFMA.\n")
    rows = []
    for gi, g in enumerate(groups):
        a = g[0]
        b = groups[gi + 1][0] if gi + 1 < len(groups) else len(ops)
        rows.append(collections.Counter(classify(o) for o in ops[a:b]))
        rows[-1]["total"] = b - a
    names = [c[0] for c in CLASSES] + ["other", "total"]
    # the regular row tiles: the ones whose length is within 25 % of the median (the others carry a layer boundary, the tile prologue or the last layer)
    med = sorted(r["total"] for r in rows)[len(rows) // 2]
    kinds = collections.OrderedDict()
    for r in rows:
        key = ("hidden layer, 8-bit staged output" if r["cvt e4m3 / e5m2"] and r["pk_max (ReLU)"] and r["total"] < 1.3 * med else
               "last layer (f32 ReLU + output dot product, mask only)" if not r["cvt e4m3 / e5m2"] and r["pk_min (mask)"] and r["total"] < 1.6 * med else
               "backward row tile (mask, keep, e5m2)" if r["pk_mul (keep)"] and r["total"] < 1.6 * med else "with a layer boundary / tile prologue")
        kinds.setdefault(key, []).append(r)
    for key, rs in kinds.items():
        print(f"== {key}: {len(rs)} row tiles")
        print(f"   {'class':28s} {'min':>5s} {'median':>7s} {'max':>6s}   purpose")
        for n in names:
            vals = sorted(r[n] for r in rs)
            if vals[-1] == 0:
                continue
            why = next((c[2] for c in CLASSES if c[0] == n), "")
            print(f"   {n:28s} {vals[0]:5d} {vals[len(vals) // 2]:7d} {vals[-1]:6d}   {why}")
        valu = sorted(sum(r[n] for n in ("cvt_pk_bf16", "pk_max (ReLU)", "pk_min (mask)", "lshl_or (mask)", "cvt e4m3 / e5m2", "pk_mul (keep)", "shift / and (flags)",
                                        "v_or / v_mov / other VALU")) for r in rs)
        print(f"   vector-ALU instructions besides the 16 MFMAs: median {valu[len(valu) // 2]} (min {valu[0]}, max {valu[-1]})\n")


if __name__ == "__main__":
    main()
