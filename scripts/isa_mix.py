"""Instruction mix of a generated kernel's loops, from the compiler's own assembly listing.

    python scripts/isa_mix.py <graph> [kernel-name-substring] [ENV=VAL ...]

Regenerates the graph's source with the graph compiler (oscen_amd/_build/ogc), compiles it for gfx950 with the
library's flags and -save-temps, and prints per kernel: registers, scratch, code size, and -- per loop nest depth,
using the asm printer's "in Loop: Header=... Depth=N" block annotations -- how many instructions of each class the
blocks of that depth hold.  The innermost depth of the voice kernel is the per-frame body (what a voice pays per
sample); scratch_* instructions there would mean spills on the hot path.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oscen_amd import build as b  # noqa: E402


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32") or op.startswith("v_rcp") or op.startswith("v_rsq"):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    return "other"


def main():
    graph = sys.argv[1]
    want = None
    env = dict(os.environ)
    for a in sys.argv[2:]:
        if "=" in a:
            k, v = a.split("=", 1)
            env[k] = v
        else:
            want = a
    b.generate()
    ogc = os.path.join(b.BUILD, "ogc")
    src = subprocess.run([ogc, graph], stdout=subprocess.PIPE, text=True, check=True, env=env).stdout
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, graph + ".hip")
        open(path, "w").write(src)
        subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-x", "hip", "-c", path, "-o", os.path.join(d, "x.o"), "-save-temps"] + b.COMMON,
                       check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(d, graph + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        out = os.path.join(d, graph + "-hip-amdgcn-amd-amdhsa-gfx950.out")
        sizes = {}
        for ln in subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", out], stdout=subprocess.PIPE, text=True).stdout.splitlines():
            f = ln.split()
            if len(f) >= 8 and f[3] == "FUNC":
                sizes[f[7]] = int(f[2])
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - |\Z)", asm, flags=re.S):
        kv = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
        meta[m.group(1)] = kv
    # function bodies
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", asm, flags=re.S | re.M):
        name = m.group(1)
        body = m.group(2).split(".section\t.rodata")[0]
        if want and want not in name:
            continue
        depth = 0
        per = collections.defaultdict(collections.Counter)
        pending = False  # a block label was seen: its loop annotations follow on comment lines
        for ln in body.splitlines():
            s = ln.strip()
            is_label = bool(re.match(r"^\.LBB\d+_\d+:", s)) or s.startswith("; %bb.")
            if is_label:
                pending, depth = True, 0
            if is_label or (pending and s.startswith(";")):
                dm = re.search(r"(?:in Loop: Header=\S+ Depth=|This (?:Inner )?Loop Header: Depth=)(\d+)", ln)
                if dm:
                    depth = int(dm.group(1))
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            pending = False
            per[depth][classify(s.split()[0])] += 1
        mk = meta.get(name, {})
        print("== %s  vgpr %s sgpr %s scratch(private_segment) %s B  vgpr_spills %s  code %s B" % (
            name, mk.get("vgpr_count"), mk.get("sgpr_count"), mk.get("private_segment_fixed_size"), mk.get("vgpr_spill_count"), sizes.get(name)))
        for d in sorted(per):
            c = per[d]
            tot = sum(c.values())
            print("   loop depth %d: %5d instr  " % (d, tot) + "  ".join("%s %d" % (k, c[k]) for k in sorted(c)))


if __name__ == "__main__":
    main()
