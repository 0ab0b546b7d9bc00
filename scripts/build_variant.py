"""Build an A/B variant of liboscen_gpu.so: python scripts/build_variant.py <tag> [ENV=VAL ...] [-- extra hipcc flags]
Output: oscen_amd/_build/liboscen_gpu_<tag>.so (select it with OSCEN_GPU_LIB=...).
The hand-off barrier experiment is a compile-time flag: python scripts/build_variant.py nosync -- -DOG_EXPERIMENT_NOSYNC (results are wrong, timing only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oscen_amd import build as b

tag = sys.argv[1]
env = dict(os.environ)
env["OSCEN_GPU_EXPERIMENTAL"] = "1"  # the OGC_* knobs are ignored without it (og_abi.h)
flags = []
rest = sys.argv[2:]
if "--" in rest:
    i = rest.index("--"); flags = rest[i + 1:]; rest = rest[:i]
for kv in rest:
    k, v = kv.split("=", 1); env[k] = v
with open(os.path.join(b.BUILD, "liboscen_gpu_%s.env" % tag), "w") as f:  # read back by scripts/ab_bench.sh
    for kv in rest:
        f.write(kv + "\n")
vdir = os.path.join(b.BUILD, "variant_" + tag)
os.makedirs(vdir, exist_ok=True)
b.generate()
ogc = os.path.join(b.BUILD, "ogc")
names = subprocess.run([ogc, "--list"], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
objs = []
for n in names:
    src = os.path.join(vdir, n + ".hip")
    with open(src, "w") as f:
        f.write(subprocess.run([ogc, n], stdout=subprocess.PIPE, text=True, check=True, env=env).stdout)
    obj = src + ".o"
    subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-x", "hip", "-c", src, "-o", obj] + b.COMMON + flags, check=True)
    objs.append(obj)
for f in b.HOST_SRCS:
    objs.append(os.path.join(b.BUILD, f + ".o"))
out = os.path.join(b.BUILD, "liboscen_gpu_%s.so" % tag)
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-o", out] + objs + ["-lhiprtc", "-ldl"], check=True)
print(out)
