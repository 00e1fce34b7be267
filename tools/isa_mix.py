"""Static instruction mix of one kernel from hipcc's gfx950 assembly, by category.

  python tools/isa_mix.py <source.hip> <mangled-name-substring> [extra hipcc flags]

The persistent 32 x 32 kernel is fully unrolled (32 steps, one basic block chain per round), so the static count of the round body
IS the dynamic count per pair, up to the rare branches (near-tie resolution, norm recomputation, the exact redo), which are listed
separately: blocks whose label is reached only through a conditional branch that `__builtin_expect` marked cold end up after the
main chain, and everything from the kernel's first `s_endpgm`-free return to its end counts as 'cold'.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CATS = [
    ("fp64 fma (dpp row_newbcast)", re.compile(r"^v_(fmac|fma)_f64.*(_dpp|row_newbcast)")),
    ("fp64 fma (plain)", re.compile(r"^v_(fmac|fma)_f64")),
    ("fp64 mul/add", re.compile(r"^v_(mul|add|pk_fma|pk_mul|pk_add)_f64")),
    ("fp64 rcp/rsq/sqrt/cmp/other", re.compile(r"^v_(rcp|rsq|sqrt|cmp|cmpx|div|ldexp|frexp|trig|max|min|fract|cvt).*_f64")),
    ("mfma", re.compile(r"^v_mfma")),
    ("v_mov / dpp moves", re.compile(r"^v_(mov|accvgpr)")),
    ("v_readlane/readfirstlane/writelane", re.compile(r"^v_(readlane|readfirstlane|writelane)")),
    ("v_permlane / v_perm / bpermute-free swaps", re.compile(r"^v_(permlane|perm_)")),
    ("v_cndmask / v_cmp (int, f32)", re.compile(r"^v_(cndmask|cmp|cmpx)")),
    ("other VALU (int, logic, shifts, address)", re.compile(r"^v_")),
    ("LDS read", re.compile(r"^ds_(read|load)")),
    ("LDS write", re.compile(r"^ds_(write|store)")),
    ("LDS bpermute/permute/other", re.compile(r"^ds_")),
    ("global/flat/buffer load", re.compile(r"^(global|flat|buffer|scratch)_load")),
    ("global/flat/buffer store/atomic", re.compile(r"^(global|flat|buffer|scratch)_(store|atomic)")),
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("s_nop / s_sleep", re.compile(r"^s_(nop|sleep)")),
    ("s_barrier / sched", re.compile(r"^s_(barrier|setprio|sethalt)")),
    ("branches", re.compile(r"^s_(cbranch|branch|call|setpc|swappc|getpc|endpgm)")),
    ("s_load / s_buffer_load", re.compile(r"^s_(load|buffer_load|store)")),
    ("other SALU", re.compile(r"^s_")),
]


def categorise(op):
    for name, rx in CATS:
        if rx.match(op):
            return name
    return "other"


def main():
    src, needle = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "qrkit_amd", "csrc"), "-S", "--cuda-device-only"] + flags + [src, "-o", out])
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_]\w*:", l) and needle in l)
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    meta = [l.strip() for l in lines[end:end + 80] if re.search(r"NumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte|LDSByteSize", l)]
    counts = collections.Counter()
    dpp_fma = 0
    total = 0
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        full = l
        cat = categorise(op if "row_newbcast" not in full else (op + "_dpp" if op.startswith(("v_fmac_f64", "v_fma_f64")) else op))
        if op.startswith("v_mov") and ("row_" in full or "quad_perm" in full or "_dpp" in op):
            cat = "v_mov / dpp moves"
        counts[cat] += 1
        total += 1
    print(f"kernel {name}")
    for m in meta:
        print("  " + m.lstrip("; "))
    print(f"  static instructions: {total}")
    groups = collections.OrderedDict()
    for cat, _ in CATS + [("other", None)]:
        if counts.get(cat):
            print(f"  {cat:48s} {counts[cat]:7d}  {100.0 * counts[cat] / total:5.1f} %")
    fma = counts["fp64 fma (dpp row_newbcast)"] + counts["fp64 fma (plain)"]
    valu = sum(v for k, v in counts.items() if k.startswith(("fp64", "v_", "other VALU", "mfma")))
    print(f"  -> FP64 FMAs {fma} = {100.0 * fma / total:.1f} % of all, {100.0 * fma / max(valu, 1):.1f} % of the {valu} VALU instructions")


if __name__ == "__main__":
    main()
