"""ABI hygiene (CPU): no C++ exception may cross the C ABI (include/oscen_gpu.h:13 "never throws across the ABI").
Every entry point the header declares is either run under ogabi::guard / guard_value (og_abi.h: exceptions carry their
OG_E_* code, std::bad_alloc -> OG_E_NOMEM) or is a plain getter whose body cannot allocate, compile or touch the
device.  The check reads the sources: it finds each definition and looks at its body."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oscen_amd", "csrc")
SOURCES = ["og_engine.cpp", "og_cluster.inl", "og_midi.cpp", "og_wav.cpp", "og_dsl.cpp", "og_graph.cpp", "og_builtin.cpp", "og_jit.cpp"]
# what makes a body able to throw: allocation, containers, strings, the compiler, explicit throws
RISKY = ["new ", "std::", "push_back", "resize", "emplace", ".assign(", "throw", "compile(", "make_unique", "HIPCK"]


def _body(text, name):
    m = re.search(r"^[A-Za-z_][A-Za-z0-9_ \*]*\b%s\s*\([^;{]*\)\s*\{" % re.escape(name), text, flags=re.M)
    if not m:
        return None
    i, depth = m.end(), 1
    while depth:
        depth += (text[i] == "{") - (text[i] == "}")
        i += 1
    return text[m.end():i]


def test_every_entry_point_is_guarded_or_cannot_throw():
    hdr = open(os.path.join(ROOT, "include", "oscen_gpu.h")).read()
    names = sorted(set(re.findall(r"\b(og_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 100
    text = {f: open(os.path.join(CSRC, f)).read() for f in SOURCES}
    undefined, unguarded = [], []
    for n in names:
        bodies = [b for b in (_body(t, n) for t in text.values()) if b is not None]
        if not bodies:
            undefined.append(n)
            continue
        for b in bodies:
            guarded = "guard(" in b or "guard_value<" in b
            # an entry point that only forwards to other (guarded) entry points is fine
            if not guarded and any(r in b for r in RISKY):
                unguarded.append(n)
    assert not undefined, undefined
    assert not unguarded, unguarded


def test_guard_maps_exception_types_to_codes_without_reading_messages():
    abi = open(os.path.join(CSRC, "og_abi.h")).read()
    guard = abi[abi.index("int guard(F&& f) noexcept"):abi.index("template <class T, class F>")]
    assert "e.code" in guard and "std::bad_alloc" in guard and "catch (...)" in guard
    assert ".find(" not in guard  # (round 3 classified errors by substrings of the message)
    eng = open(os.path.join(CSRC, "og_engine.cpp")).read()
    assert 'm.find("not supported")' not in eng


def test_environment_knobs_are_not_read_on_the_block_path():
    eng = open(os.path.join(CSRC, "og_engine.cpp")).read()
    for fn in ("og_process_block", "og_process_block_async", "og_process_blocks_async", "og_midi_process_block"):
        b = _body(eng, fn) or _body(open(os.path.join(CSRC, "og_midi.cpp")).read(), fn)
        assert b is not None and "getenv" not in b, fn


def test_only_listed_environment_variables_are_read_and_experiment_knobs_are_gated():
    """og_abi.h: SETTINGS are read as they are; every other variable is an experiment knob, read through
    ogabi::experiment_knob() (nullptr unless OSCEN_GPU_EXPERIMENTAL=1).  No source reads a variable that og_version() does not
    list, no experiment knob is read with a bare getenv, and the one knob whose effect was WRONG RESULTS (the hand-off barrier
    experiment) is not a run-time knob at all: it is a compile-time flag of scripts/build_variant.py."""
    import re

    abi = open(os.path.join(CSRC, "og_abi.h")).read()
    settings = set(re.search(r'#define OG_SETTINGS "([^"]+)"', abi).group(1).split())
    knobs = set(" ".join(re.findall(r'"([A-Z0-9_ ]+)"', abi[abi.index("#define OG_EXPERIMENT_KNOBS"):abi.index("inline bool experiments_enabled")])).split())
    assert settings and knobs and not (settings & knobs)
    raw, gated = set(), set()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".cpp", ".h", ".inl")):
            continue
        t = open(os.path.join(CSRC, f)).read()
        raw |= set(re.findall(r'(?<![\w:])getenv\("([A-Z0-9_]+)"\)', t))
        gated |= set(re.findall(r'experiment_knob\("([A-Z0-9_]+)"\)', t))
    assert raw <= settings, sorted(raw - settings)
    assert gated <= knobs, sorted(gated - knobs)
    assert "OGC_NOSYNC" not in knobs | settings | raw | gated
    assert "OG_EXPERIMENT_NOSYNC" in open(os.path.join(CSRC, "og_kernel_rt.hip.h")).read()
    eng = open(os.path.join(CSRC, "og_engine.cpp")).read()
    assert "OG_SETTINGS" in _body(eng, "og_version") and "OG_EXPERIMENT_KNOBS" in _body(eng, "og_version")
