#!/usr/bin/env python3
"""tools/isa_census.py <file.s> <kernel symbol substring>  --  static instruction census of one gfx950 kernel.

Counts the VALU instructions of the kernel by class, for the whole kernel and per basic block (label), so that the blocks of the
hot loop can be read off (the largest blocks with a backward branch).  Classes follow what tools/valu_issue_probe.hip measured on MI355X:
  full rate (one wave64 instruction per ~2.4 cycles per SIMD): v_fma / v_mul / v_add / v_sub / v_mac / v_mov / v_max / v_min / v_cvt / v_and ...
  half rate (~4.2 cycles): three-source VOP3 selects and medians (v_cndmask_b32_e64, v_med3, v_min3 / v_max3, v_bfe, v_perm ...)
  quarter rate (~8.2 cycles): v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos
"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w.$]*:", l) and want in l.split(":")[0] and not l.startswith("."))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def klass(op):
    if op.startswith(TRANS):
        return "transcendental"
    if op.startswith("v_cndmask"):
        return "v_cndmask e64" if op.endswith("_e64") else "v_cndmask e32 (vcc)"
    if op.startswith(("v_med3", "v_min3", "v_max3")):
        return "v_med3/min3/max3"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_readlane"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith(("v_fma", "v_mac", "v_fmac", "v_mad")):
        return "v_fma"
    if op.startswith(("v_mul_f", "v_add_f", "v_sub_f", "v_subrev_f")):
        return "v_mul/add f32"
    if op.startswith(("v_max_f", "v_min_f")):
        return "v_max/min f32"
    if op.startswith("v_cvt"):
        return "v_cvt"
    if op.startswith("v_pk_"):
        return "v_pk"
    return "other valu"


total = collections.Counter()
blocks = []
cur, cur_c, cur_other = lines[start].split(":")[0], collections.Counter(), collections.Counter()
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((cur, cur_c, cur_other)); cur, cur_c, cur_other = t.rstrip(":").split(":")[0], collections.Counter(), collections.Counter()
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        total[klass(op)] += 1; cur_c[klass(op)] += 1
    else:
        fam = "global/buffer load" if op.startswith(("global_load", "buffer_load", "flat_load")) else "global store" if op.startswith(("global_store", "buffer_store")) else \
              "lds" if op.startswith("ds_") else "s_load" if op.startswith("s_load") else "s_waitcnt" if op.startswith("s_waitcnt") else "branch" if op.startswith("s_cbranch") or op.startswith("s_branch") else "salu"
        total["| " + fam] += 1; cur_other[fam] += 1
blocks.append((cur, cur_c, cur_other))
valu = sum(v for k, v in total.items() if not k.startswith("|"))
print(f"kernel {lines[start].split(':')[0]}: {end - start} lines, {valu} VALU instructions (static)")
for k, v in sorted(total.items(), key=lambda kv: (kv[0].startswith("|"), -kv[1])):
    print(f"  {k:24s} {v:6d}" + (f"  {100.0 * v / valu:5.1f} % of VALU" if not k.startswith("|") else ""))
print("largest basic blocks (VALU count: classes):")
for name, c, o in sorted(blocks, key=lambda b: -sum(b[1].values()))[:12]:
    n = sum(c.values())
    print(f"  {name:12s} {n:5d}: " + ", ".join(f"{k} {v}" for k, v in c.most_common(8)) + " | " + ", ".join(f"{k} {v}" for k, v in o.most_common(4)))
