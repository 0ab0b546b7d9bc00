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
