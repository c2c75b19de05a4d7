#!/usr/bin/env python3
"""Static opcode-class table of one kernel from hipcc's assembly (where the non-fp64 VALU instructions of
rollout_kernel<3,1024,3,true,false> sit).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPMPC_DP=3 -S --cuda-device-only -o /tmp/rollout_dp3.s csrc/rollout_dp.hip
  python tools/opcode_table.py /tmp/rollout_dp3.s _ZN9gpmpc_hip14rollout_kernelILi3ELi1024ELi3ELb1ELb0EEEvNS_11RolloutArgsE

Per basic block: instruction counts by class (fp64 arithmetic | other VALU split into mov / cndmask / compare / integer + logic /
DPP + lane exchange / conversions + other | LDS | VMEM | SALU | waits), the loop depth LLVM prints, and whether the block
branches back to itself (an innermost loop).  Printed: the kernel totals, the innermost loops ranked by their fp64 content (the
pairwise item loops), and everything else aggregated by loop depth -- the per-item and per-step code."""
import re
import sys
from collections import Counter, defaultdict

path, kname = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kname + ":"))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))


def classify(ins):
    op = ins.split()[0]
    if re.match(r"v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|div_fixup|div_fmas|div_scale|ldexp|frexp_mant|trunc|rndne|floor)_f64", op):
        return "fp64"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "dpp" in ins or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_bpermute")):
            return "v_dpp_lane"
        if op.startswith(("v_mov", "v_pk_mov", "v_accvgpr", "v_swap")):
            return "v_mov"
        if op.startswith("v_cndmask"):
            return "v_cndmask"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "v_cmp"
        if op.startswith("v_cvt") or op.startswith("v_frexp") or op.startswith("v_ldexp"):
            return "v_cvt"
        return "v_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, cur = [], {"name": "entry", "ins": [], "depth": 0}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
    if m:
        blocks.append(cur)
        d = re.search(r"Depth=(\d+)", l)
        cur = {"name": m.group(1), "ins": [], "depth": int(d.group(1)) if d else 0}
        continue
    d = re.search(r"Depth=(\d+)", l)
    if d and not cur["ins"]:
        cur["depth"] = max(cur["depth"], int(d.group(1)))
    t = l.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t)
blocks.append(cur)

CL = ["fp64", "v_mov", "v_cndmask", "v_cmp", "v_int", "v_dpp_lane", "v_cvt", "lds", "vmem", "salu", "wait", "barrier"]


def row(label, c, extra=""):
    valu = sum(c[k] for k in ("fp64", "v_mov", "v_cndmask", "v_cmp", "v_int", "v_dpp_lane", "v_cvt"))
    share = f"{100.0 * c['fp64'] / valu:5.1f}%" if valu else "   - "
    print(f"{label:26s}" + "".join(f"{c[k]:8d}" for k in CL) + f"  fp64/VALU {share} {extra}")


print(f"# {kname}\n# static instruction counts by class (one count per instruction in the code object, not per execution)")
print(f"{'':26s}" + "".join(f"{k:>8s}" for k in CL))
total = Counter()
for b in blocks:
    b["c"] = Counter(classify(i) for i in b["ins"])
    b["self"] = any(("s_cbranch" in i and b["name"] in i) for i in b["ins"])
    total.update(b["c"])
row("kernel total", total)
inner = sorted([b for b in blocks if b["self"] and b["c"]["fp64"] >= 12], key=lambda b: -b["c"]["fp64"])
print("# innermost loops of the pairwise items (per trip: 2 rows x 2 columns per lane in the two-column forms), by Taylor degree")
for b in inner:
    row(f"  loop {b['name']} d{b['depth']}", b["c"])
agg = defaultdict(Counter)
for b in blocks:
    if b in inner:
        continue
    agg[b["depth"]].update(b["c"])
print("# everything outside those loops, by loop depth (0: once per launch, 1: per horizon step, 2: per pair group / work item, 3+: item bodies)")
for d in sorted(agg):
    row(f"  depth {d}", agg[d])
