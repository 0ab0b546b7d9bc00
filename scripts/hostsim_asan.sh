#!/bin/bash
# The GPU tests the host simulator can run, under AddressSanitizer (device memory is host memory there: an out-of-bounds
# access of a KERNEL -- state planes, event lists, partial rows, LDS tiles -- is a heap / global overflow ASan reports).
#   scripts/hostsim_asan.sh            the subset of tests/hostsim/subset.txt (about a minute)
#   scripts/hostsim_asan.sh all        every *_gpu.py test (minutes; the full-size banks time out)
set -e
cd "$(dirname "$0")/.."
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$($CXX -print-file-name=libclang_rt.asan-x86_64.so)
OG_HOSTSIM_ASAN=1 python tests/hostsim/build_hostsim.py
python tests/hostsim/build_hostsim.py > /dev/null   # (the stand-in librccl lives in the plain build directory)
if [ "${1:-}" = "all" ]; then IDS=$(ls tests/*_gpu.py tests/test_golden_waveforms.py); else IDS=$(grep -v "^#" tests/hostsim/subset.txt); fi
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
LD_LIBRARY_PATH=$PWD/tests/hostsim/_build/fake_rccl OSCEN_GPU_LIB=$PWD/tests/hostsim/_build_asan/liboscen_gpu_hostsim.so \
python -m pytest -m gpu -q -n ${HOSTSIM_WORKERS:-8} --timeout 300 -p no:cacheprovider $IDS
