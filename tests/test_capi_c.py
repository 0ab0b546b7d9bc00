"""The Rust shim's call sequence as a C program (tests/capi/shim_sequence.c) against include/oscen_gpu.h, with guard
words around its buffers: a 4-channel graph (`out_block` = OG_MAX_BUS_CHANNELS x 512 floats) and a graph with a
Frame<2> stream input (`stream_in_block` = frames x og_stream_input_channels floats).  The crate itself cannot be
compiled here (no rustc); this holds the buffer contract it relies on.  Reference surface:
oscen-graph-compiler/src/codegen/mod.rs:1196,1220,1306; oscen-lib/src/graph/offline.rs:19-113.

CPU: the program compiles with -Wall -Wextra -Werror -pedantic against the header (so the header is valid C99), links
against liboscen_gpu.so and passes its argument checks.  GPU: it runs."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi", "shim_sequence.c")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import oscen_amd

    lib = oscen_amd.load_library()._name
    out = str(tmp_path_factory.mktemp("capi") / "shim_sequence")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out,
           lib, "-Wl,-rpath," + os.path.dirname(lib), "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out


def test_c_program_builds_against_the_header_and_rejects_bad_arguments(exe):
    r = subprocess.run([exe, "--no-device"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "argument checks ok" in r.stdout


def test_rust_shim_buffers_match_the_c_program():
    """the sizes the C program guards are the sizes the shim declares"""
    shim = open(os.path.join(ROOT, "bindings", "rust", "oscen-gpu", "src", "lib.rs")).read()
    code = "\n".join(ln for ln in shim.splitlines() if not ln.lstrip().startswith("//"))
    assert re.search(r"pub const MAX_BUS_CHANNELS: usize = 4;", code)
    assert re.search(r"out_block: \[f32; MAX_BUS_CHANNELS \* MAX_BLOCK_SIZE\]", code)
    assert re.search(r"out: \[f32; MAX_BUS_CHANNELS\]", code)
    assert re.search(r"stream_in_blocks: \[\[f32; MAX_BUS_CHANNELS \* MAX_BLOCK_SIZE\]; IN\]", code)
    assert "og_stream_input_channels" in code and "device_id" in code
    hdr = open(os.path.join(ROOT, "oscen_amd", "csrc", "og_kernel_rt.hip.h")).read()
    assert re.search(r"#define OG_MAX_BUS_CHANNELS 4\b", hdr)
    # init and the setters hand their error codes on
    for fn in ("init", "set", "set_with_ramp", "set_immediate", "set_voice", "process_block"):
        assert re.search(r"pub fn %s\(&mut self[^)]*\) -> Result<" % fn, code), fn


@pytest.mark.gpu
def test_shim_sequence_runs_on_the_device_with_guard_words_intact(exe):
    r = subprocess.run([exe, "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "shim_sequence: ok" in r.stdout
