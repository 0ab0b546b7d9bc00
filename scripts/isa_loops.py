"""Innermost loops of one kernel in a gfx950 assembly listing: for every backward branch the layout-contiguous span
[target label .. branch] with its instruction mix (v_mov copies listed apart: loop-carried values the register allocator
reconciles on every trip).

    python scripts/isa_loops.py /tmp/fm.s og_k_<hash>_00 [min instructions]
"""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
ins, label_at = [], {}
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        label_at[m.group(1)] = len(ins)
        continue
    if not t or t[0] in ";.":
        continue
    ins.append(t.split(";")[0].strip())
spans = []
for i, t in enumerate(ins):
    op = t.split()[0]
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = t.split()[1]
        if tgt in label_at and label_at[tgt] <= i:
            spans.append((label_at[tgt], i, tgt))
# innermost: spans that contain no other span
inner = [s for s in spans if not any(o is not s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]
for a, b, tgt in sorted(inner):
    if b - a + 1 < min_n:
        continue
    c = collections.Counter()
    for t in ins[a:b + 1]:
        op = t.split()[0]
        if op in ("v_mov_b32_e32", "v_mov_b64_e32"): c["v_mov"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch": c["branch"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")): c["vmem"] += 1
        else: c["other"] += 1
    print("%-12s %5d instr: valu %4d v_mov %3d salu %3d branch %2d lds %2d vmem %2d wait %2d" % (
        tgt, b - a + 1, c["valu"], c["v_mov"], c["salu"], c["branch"], c["lds"], c["vmem"], c["wait"]))
