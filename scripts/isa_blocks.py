"""Basic blocks of one kernel in a gfx950 assembly listing, largest first, with their instruction mix.

    hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Ioscen_amd/csrc -S --cuda-device-only \
          -o /tmp/fm.s oscen_amd/csrc/gen/fm_voice.hip
    python scripts/isa_blocks.py /tmp/fm.s og_k4_<hash>_00 [min instructions]

The unrolled chunk bodies of the generated kernels are the large straight-line blocks: VALU per block / frames per chunk is
what a quiet frame costs the wave (a static count; rocprofv3's SQ_INSTS_VALU is the dynamic one)."""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].strip().startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur))
            name, cur = t.split(":")[0], []
        continue
    op = t.split()[0]
    cur.append((op, t))
    if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm" or op == "s_setpc_b64":
        blocks.append((name, cur))
        name, cur = name + "+", []
blocks.append((name, cur))


def kind(op):
    if op.startswith("v_pk_"): return "vpk"
    if op in ("v_sin_f32_e32", "v_cos_f32_e32", "v_rcp_f32_e32", "v_exp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32",
              "v_sin_f32_e64", "v_rcp_f32_e64", "v_exp_f32_e64"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


tot = collections.Counter()
for n, b in blocks:
    for op, _ in b: tot[kind(op)] += 1
print(kern, "total", dict(tot), "blocks", len(blocks))
big = sorted(((len(b), n, b) for n, b in blocks if len(b) >= min_n), reverse=True)
for ln, n, b in big:
    c = collections.Counter(kind(op) for op, _ in b)
    ops = collections.Counter(op for op, _ in b if op.startswith("v_"))
    print("%-14s %5d instr  valu %4d trans %3d vpk %3d salu %3d lds %3d vmem %3d wait %3d | top: %s" % (
        n, ln, c["valu"], c["trans"], c["vpk"], c["salu"], c["lds"], c["vmem"], c["wait"],
        " ".join("%s:%d" % (k.replace("_e32", "").replace("_e64", "").replace("v_", ""), v) for k, v in ops.most_common(14))))
