#!/usr/bin/env python3
"""Where do a kernel's spilled registers land?  (SGPR spills are v_writelane / v_readlane, i.e. vector issue slots,
in kernels that are bound by their MFMA + vector issue.)

    python3 tools/isa_spills.py nerf-ca_amd/csrc/nca_kernels_bf16.hip 'nca_fused_bf16<128, 2, true, true>' 'nca_fused_bf16<128, 5, true, true>' ...

Compiles the translation unit for gfx950 (device only), disassembles it WITH addresses, and for every kernel named (demangled
prefix match) reports

  * its loops: every backward branch (target = address + 4 + 4 * simm16) as [head, tail], nested by span -- the outermost one of a
    fused kernel is the persistent TILE loop, the ones inside it are the k-step / row-tile loops the compiler kept as loops;
  * v_readlane / v_writelane (scalar spill traffic), scratch_load / scratch_store (vector spill traffic), s_load / s_buffer_load
    (argument-block reads) and v_mfma counts: in the whole kernel, outside the tile loop (prologue / epilogue: once per launch),
    inside the tile loop (once per 64-sample tile), and inside the INNER loops and the MFMA-dense stretches (the layer chains: from the
    first to the last MFMA of every run of MFMAs less than 64 instructions apart);
  * per MFMA-dense stretch: instructions, MFMAs, lane moves, scratch operations.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(src):
    t = tempfile.mkdtemp()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                    "--cuda-device-only", "-c", src, "-o", f"{t}/x.co"], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/x.co", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={t}/x.elf"], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", f"{t}/x.elf"], check=True, capture_output=True, text=True).stdout


def kernels(text):
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            name = m.group(1)
            out[name] = []
            continue
        m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-F]+):", line)
        if m and name is not None:
            out[name].append((int(m.group(2), 16), m.group(1)))
    return out


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "sload"
    if op.startswith("v_mfma"):
        return "mfma"
    return None


def analyse(name, ins):
    addr_index = {a: i for i, (a, _) in enumerate(ins)}
    loops = []
    for i, (a, s) in enumerate(ins):
        m = re.match(r"^s_c?branch\S*\s+(\d+)$", s)
        if m:
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
        # a branch beyond +-128 KiB (these kernels are up to 140 KiB of straight-line code per tile): s_getpc_b64 / s_add_u32 imm /
        # s_addc_u32 / s_setpc_b64 -- target = address of the s_getpc + 4 + imm
        if s.startswith("s_getpc_b64") and i + 3 < len(ins) and ins[i + 3][1].startswith("s_setpc_b64"):
            m2 = re.match(r"^s_add_u32\s+\S+\s+\S+\s+(0x[0-9a-f]+|-?\d+)$", ins[i + 1][1])
            if m2:
                imm = int(m2.group(1), 0)
                if imm >= 1 << 31:
                    imm -= 1 << 32
                tgt = a + 4 + imm
                if tgt <= a and tgt in addr_index:
                    loops.append((addr_index[tgt], i + 3))
    loops.sort(key=lambda l: l[0] - l[1])            # widest first
    kinds = [classify(s) for _, s in ins]
    total = {k: sum(1 for x in kinds if x == k) for k in ("lane", "scratch", "sload", "mfma")}
    print(f"== {name}: {len(ins)} instructions, {total['mfma']} MFMAs, {total['lane']} v_readlane/v_writelane, {total['scratch']} scratch operations, {total['sload']} scalar loads")
    if not loops:
        print("   no loop")
        return
    # the tile loop: the NARROWEST loop that holds every MFMA of the kernel (a wider one may start in the launch prologue: the
    # compiler's loop rotation and the long-branch trampolines behind s_endpgm)
    allm = [i for i, k in enumerate(kinds) if k == "mfma"]
    holding = [l for l in loops if allm and l[0] <= allm[0] and l[1] >= allm[-1]]
    outer = max(holding, key=lambda l: l[0] - l[1]) if holding else loops[0]
    loops = [outer] + [l for l in loops if l != outer]

    def count(lo, hi, inner_mask=None):
        c = {k: 0 for k in ("lane", "scratch", "sload", "mfma", "all")}
        for i in range(lo, hi + 1):
            if inner_mask is not None and not inner_mask[i]:
                continue
            c["all"] += 1
            if kinds[i]:
                c[kinds[i]] += 1
        return c

    inside = count(*outer)
    print(f"   tile loop = instructions {outer[0]} .. {outer[1]} ({inside['all']} instructions, once per 64-sample tile): "
          f"{inside['mfma']} MFMAs, {inside['lane']} lane moves, {inside['scratch']} scratch operations, {inside['sload']} scalar loads")
    print(f"   outside it (once per launch): {total['lane'] - inside['lane']} lane moves, {total['scratch'] - inside['scratch']} scratch operations, "
          f"{total['sload'] - inside['sload']} scalar loads")
    inner = [l for l in loops[1:] if l[0] >= outer[0] and l[1] <= outer[1]]
    mask = [False] * len(ins)
    for lo, hi in inner:
        for i in range(lo, hi + 1):
            mask[i] = True
    ci = count(outer[0], outer[1], mask)
    print(f"   inner loops (kept as loops inside the tile loop): {len(inner)} covering {ci['all']} instructions: {ci['mfma']} MFMAs, {ci['lane']} lane moves, {ci['scratch']} scratch operations")
    # MFMA-dense stretches of the tile loop
    mf = [i for i in range(outer[0], outer[1] + 1) if kinds[i] == "mfma"]
    runs, start, prev = [], None, None
    for i in mf:
        if start is None:
            start = prev = i
        elif i - prev > 64:
            runs.append((start, prev))
            start = prev = i
        else:
            prev = i
    if start is not None:
        runs.append((start, prev))
    tl = ts = tm = ta = 0
    rows = []
    for lo, hi in runs:
        c = count(lo, hi)
        tl += c["lane"]; ts += c["scratch"]; tm += c["mfma"]; ta += c["all"]
        rows.append((lo, hi, c))
    print(f"   MFMA-dense stretches (MFMAs < 64 instructions apart; the layer chains): {len(runs)} stretches, {ta} instructions, {tm} MFMAs, "
          f"{tl} lane moves, {ts} scratch operations  -> {inside['lane'] - tl} lane moves and {inside['scratch'] - ts} scratch operations of the tile loop sit BETWEEN the chains")
    for lo, hi, c in rows:
        if c["lane"] or c["scratch"] or c["all"] > 400:
            print(f"      [{lo:6d} .. {hi:6d}] {c['all']:5d} instructions, {c['mfma']:4d} MFMAs, {c['lane']:3d} lane moves, {c['scratch']:3d} scratch operations")
    # the stretches BETWEEN the chains that hold lane moves: what else is there (the kind of code the spilled scalars serve)
    fam = (("sincos / encoding", ("v_sin", "v_cos", "v_fract", "v_rndne", "v_mul_f64", "v_fma_f64", "v_trig")), ("global / scratch memory", ("global_", "scratch_", "buffer_")),
           ("LDS", ("ds_",)), ("conversions", ("v_cvt",)), ("packed 16-bit", ("v_pk_",)), ("permute / swap", ("v_permlane", "ds_bpermute", "v_mov_b32_dpp")),
           ("scalar ALU", ("s_add", "s_mul", "s_lshl", "s_lshr", "s_and", "s_or", "s_cmp", "s_mov", "s_cselect", "s_sub", "s_ashr", "s_bfe", "s_min", "s_max")))
    gaps, prev_hi = [], outer[0] - 1
    for lo, hi in runs + [(outer[1] + 1, outer[1] + 1)]:
        if lo - 1 > prev_hi:
            gaps.append((prev_hi + 1, lo - 1))
        prev_hi = hi
    print("   stretches between the chains with lane moves:")
    for lo, hi in gaps:
        c = count(lo, hi)
        if not c["lane"]:
            continue
        ops = [ins[i][1].split()[0] for i in range(lo, hi + 1)]
        mix = ", ".join(f"{n} {sum(1 for o in ops if o.startswith(pre))}" for n, pre in fam if sum(1 for o in ops if o.startswith(pre)))
        print(f"      [{lo:6d} .. {hi:6d}] {c['all']:5d} instructions, {c['lane']:3d} lane moves ({sum(1 for o in ops if o.startswith('v_readlane'))} reads), {c['sload']:2d} scalar loads;  {mix}")


def main():
    src = sys.argv[1]
    want = sys.argv[2:]
    ks = kernels(disassemble(src) if src.endswith(".hip") else open(src).read())
    for w in want:
        hits = [k for k in ks if k.startswith(w)]
        if not hits:
            print(f"== {w}: no such kernel; have e.g. {sorted(ks)[:3]}")
        for k in hits:
            analyse(k, ks[k])


if __name__ == "__main__":
    main()
