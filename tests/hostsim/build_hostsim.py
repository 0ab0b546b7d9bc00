"""TEST INFRASTRUCTURE: build tests/hostsim/_build/liboscen_gpu_hostsim.so -- the engine's host code and the generated voice
kernels compiled for x86 against tests/hostsim/hip/hip_runtime.h (see there).  Used by tests/test_hostsim_cpu.py only.

    python tests/hostsim/build_hostsim.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oscen_amd import build as b  # noqa: E402  (source lists and the generator; nothing of the product is changed)

# OG_HOSTSIM_ASAN=1: an AddressSanitizer build in its own directory (device memory is host memory here, so an out-of-bounds
# access of a KERNEL is a heap overflow ASan sees).  Run the tests with LD_PRELOAD=<libclang_rt.asan-x86_64.so> and
# ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 (the fibres' stacks are not ASan's).
ASAN = bool(os.environ.get("OG_HOSTSIM_ASAN"))
OUT = os.path.join(HERE, "_build_asan" if ASAN else "_build")
LIB = os.path.join(OUT, "liboscen_gpu_hostsim.so")


def cxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "clang++", "g++"):
        if os.path.exists(c) or "/" not in c:
            return c


FLAGS = ["-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mfma", "-mavx2", "-DOG_HOSTSIM=1", "-Wno-unknown-attributes", "-Wno-unused-value",
         "-Wno-pass-failed", "-Wno-unknown-pragmas", "-I" + HERE, "-I" + b.CSRC,
         # og_jit_hostsim.cpp: where graphs without an ahead-of-time kernel are compiled at run time
         '-DOG_HOSTSIM_DIR="%s"' % HERE, '-DOG_CSRC_DIR="%s"' % b.CSRC, '-DOG_HOSTSIM_CXX="%s"' % cxx()]


def build(force=False):
    if ASAN and "-fsanitize=address" not in FLAGS:
        FLAGS.extend(["-fsanitize=address", "-fno-omit-frame-pointer", "-g1", '-DOG_HOSTSIM_JIT_FLAGS=" -fsanitize=address -fno-omit-frame-pointer -g1"',
                      '-DOG_HOSTSIM_BUILD="_build_asan"'])
    os.makedirs(OUT, exist_ok=True)
    gens = b.generate()
    srcs = [os.path.join(b.CSRC, f) for f in b.HOST_SRCS if f != "og_jit.cpp"]
    srcs += [os.path.join(HERE, "og_jit_hostsim.cpp"), os.path.join(HERE, "simt.cpp")] + list(gens)
    deps = [os.path.join(b.CSRC, h) for h in b.HEADERS] + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        if force or b._newer(o, [s] + deps):
            jobs.append((s, subprocess.Popen([cxx(), "-x", "c++", "-c", s, "-o", o] + FLAGS, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    bad = False
    for s, p in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write("== %s\n%s\n" % (s, out[-6000:]))
            bad = True
    if bad:
        raise RuntimeError("hostsim build failed")
    if force or jobs or not os.path.exists(LIB):
        import shutil

        shutil.rmtree(os.path.join(OUT, "jit"), ignore_errors=True)  # (kernels compiled at run time against the old headers)
        b._run([cxx(), "-shared", "-o", LIB] + objs + (["-fsanitize=address", "-shared-libasan"] if ASAN else []) + ["-Wl,--allow-multiple-definition", "-ldl", "-lpthread"])  # (a noinline __device__ function of og_nodes.hip.h is emitted by every generated unit)
    # the stand-in for librccl.so.1 (clusters of several simulated devices): found through LD_LIBRARY_PATH by the test process
    fake_dir = os.path.join(OUT, "fake_rccl")
    os.makedirs(fake_dir, exist_ok=True)
    fake = os.path.join(fake_dir, "librccl.so.1")
    fsrc = os.path.join(HERE, "fake_rccl.cpp")
    if force or b._newer(fake, [fsrc]):
        b._run([cxx(), "-shared", "-fPIC", "-O1", "-std=c++17", "-o", fake, fsrc])
    return LIB


if __name__ == "__main__":
    print(build("--force" in sys.argv))
