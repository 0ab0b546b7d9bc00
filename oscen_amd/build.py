"""Build liboscen_gpu.so (HIP kernels for gfx950 + host runtime + C ABI).

    python -m oscen_amd.build [--force]

1. builds the graph compiler front end `ogc` with g++ and regenerates the
   ahead-of-time kernels csrc/gen/<graph>.hip for every built-in graph;
2. compiles everything with hipcc --offload-arch=gfx950 into
   oscen_amd/liboscen_gpu.so (in-tree: it travels to the GPU box as is).

hipcc cross-compiles without a GPU.  -ffp-contract=off everywhere: the
reference's Rust never fuses or re-associates f32 arithmetic; the only fused
operations are the explicit fmaf() calls in og_math.h.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
GEN = os.path.join(CSRC, "gen")
LIB = os.path.join(PKG, "liboscen_gpu.so")
BUILD = os.path.join(PKG, "_build")
ARCH = "gfx950"

HOST_SRCS = ["og_engine.cpp", "og_graph.cpp", "og_builtin.cpp", "og_dsl.cpp", "og_midi.cpp", "og_wav.cpp", "og_jit.cpp"]
HEADERS = ["og_math.h", "og_nodes.hip.h", "og_kernel_rt.hip.h", "og_graph.h", "og_registry.h", "og_jit.h", "og_cluster.inl",
           os.path.join("..", "..", "include", "oscen_gpu.h")]
# -fno-slp-vectorize: left to itself clang pairs adjacent scalar f32 ops of the tick into v_pk_*_f32; on gfx950
# a packed f32 instruction costs more issue time than the two scalar ones it replaces (measured: fm_voice
# 0.0741 -> 0.0708 ms at 65 536 voices, 0.194 -> 0.185 ms at 262 144; see also scripts/pk_probe)
COMMON = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-I" + CSRC]


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build step failed: " + cmd[0])
    return r.stdout


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def generate(force=False):
    """Regenerate csrc/gen/*.hip with the graph compiler; returns the file list."""
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(GEN, exist_ok=True)
    ogc = os.path.join(BUILD, "ogc")
    srcs = [os.path.join(CSRC, f) for f in ("ogc_main.cpp", "og_graph.cpp", "og_builtin.cpp")]
    if force or _newer(ogc, srcs + [os.path.join(CSRC, "og_graph.h")]):
        _run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I" + CSRC, "-o", ogc] + srcs)
    names = _run([ogc, "--list"]).split()
    files = []
    for n in names:
        text = _run([ogc, n])
        path = os.path.join(GEN, n + ".hip")
        old = open(path).read() if os.path.exists(path) else None
        if old != text:
            with open(path, "w") as f:
                f.write(text)
        files.append(path)
    return files


def build(force=False, verbose=False):
    gen_files = generate(force)
    deps = [os.path.join(CSRC, f) for f in HOST_SRCS + HEADERS] + gen_files
    if not force and not _newer(LIB, deps):
        return LIB
    cc = hipcc()
    objs = []
    for src in [os.path.join(CSRC, f) for f in HOST_SRCS] + gen_files:
        obj = os.path.join(BUILD, os.path.basename(src) + ".o")
        if force or _newer(obj, [src] + [os.path.join(CSRC, h) for h in HEADERS]):
            out = _run([cc, "--offload-arch=" + ARCH, "-x", "hip", "-c", src, "-o", obj] + COMMON)
            if verbose and out.strip():
                print(out)
        objs.append(obj)
    _run([cc, "--offload-arch=" + ARCH, "-shared", "-o", LIB] + objs + ["-lhiprtc", "-ldl"])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
