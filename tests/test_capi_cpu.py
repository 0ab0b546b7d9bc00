"""CPU-only checks of the product's host side: the C-ABI library loads and
exports every symbol include/oscen_gpu.h declares, the graph compiler lowers
descriptions the way the reference's graph! macro does, and the synthetic input
generator agrees with the oracle's.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oscen_amd
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from oscen_amd import build

    build.build()
    return oscen_amd.load_library()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "oscen_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(og_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 40
    for n in sorted(names):
        assert hasattr(lib, n), "missing export " + n
    assert b"gfx950" in lib.og_version()


def test_generated_sources_are_committed_and_current(lib):
    # csrc/gen/*.hip must be exactly what the graph compiler emits today
    for name in ("fm_voice", "sub_voice", "sat4x_voice", "sat1x_voice"):
        src = oscen_amd.Graph(builtin=name).kernel_source()
        path = os.path.join(ROOT, "oscen_amd", "csrc", "gen", name + ".hip")
        assert open(path).read() == src


def test_fm_voice_lowering(lib):
    src = oscen_amd.Graph(builtin="fm_voice").kernel_source()
    order = re.search(r"// Node order: (.*)", src).group(1).split()
    assert len(order) == 13
    pos = {n: i for i, n in enumerate(order)}
    # topological constraints of fm_voice.rs:81-155
    for a, b in [("env3", "op3_osc"), ("op3_osc", "op3_route"), ("op3_route", "op2_osc"),
                 ("op2_osc", "op1_mod_mixer"), ("op1_mod_mixer", "op1_osc"), ("op1_osc", "filter"),
                 ("env_filter", "filter_env_gain"), ("filter_env_gain", "cutoff_mod"), ("cutoff_mod", "filter"),
                 ("filter", "output_gain")]:
        assert pos[a] < pos[b]
    assert "30 state words/voice" in src and "8 ramped inputs" in src


def _simple(extra=None):
    g = oscen_amd.Graph("t")
    g.input_value("frequency", 440.0, per_voice=True)
    g.input_value("cutoff", 1200.0)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("filter", "TptFilter::new", 1200.0, 0.707)
    g.connect("frequency", "osc.frequency").connect("cutoff", "filter.cutoff")
    g.connect("osc.output", "filter.input")
    if extra:
        extra(g)
    return g


def test_dead_node_removal_and_fanin_sum(lib):
    def extra(g):
        g.node("unused", "PolyBlepOscillator::sine", 5.0, 0.2)   # reaches no output -> pruned
        g.node("osc2", "PolyBlepOscillator::square", 220.0, 0.25)
        g.connect("osc2.output", "filter.input")                 # second source into one input = sum
        g.connect("filter.output * 0.5 + osc2.output", "out")    # compound source
    src = _simple(extra).kernel_source()
    assert "unused" not in re.search(r"// Node order: (.*)", src).group(1)
    v = r"(?:x\d+_)?n\d+_output"  # a value that crosses the two-wave pipeline cut carries an x<k>_ alias
    assert re.search(r"tpt_tick\(\(%s \+ %s\)" % (v, v), src)
    assert re.search(r"go0 = \(\(%s \* 0x1p-1f\) \+ %s\)" % (v, v), src) and "g_out = go0;" in src  # (every stream output is a named value; the bus takes them in order)


def test_compile_errors_are_reported(lib):
    g = _simple()
    g.connect("nosuch.output", "out")
    with pytest.raises(oscen_amd.OscenError) as e:
        g.kernel_source()
    assert "unknown node" in str(e.value)
    g = oscen_amd.Graph("cyc")
    g.output_stream("out")
    g.node("a", "Gain::new", 1.0)
    g.node("b", "Gain::new", 1.0)
    g.connect("a.output", "b.input").connect("b.output", "a.input").connect("b.output", "out")
    with pytest.raises(oscen_amd.OscenError) as e:
        g.kernel_source()
    assert "cycle" in str(e.value)
    g = oscen_amd.Graph("bad")
    g.node("x", "NoSuchNode::new")
    with pytest.raises(oscen_amd.OscenError):
        g.kernel_source()


def test_engine_needs_a_gpu_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Engine("fm_voice", 64)
    assert e.value.code == oscen_amd.OG_E_DEVICE


def test_note_plans_match_oracle_generator():
    n = 300
    plans = oscen_amd.note_plans(n)
    for v in list(range(40)) + [n - 1]:
        p = ol.note_plan(oscen_amd.SYNTH_SEED, v)
        assert (p.note, p.velocity, p.on_frame, p.off_frame, p.retrig_frame) == oscen_amd.note_plan(v)
        assert plans["note"][v] == p.note and plans["velocity"][v] == p.velocity
        assert plans["on_frame"][v] == p.on_frame and plans["off_frame"][v] == p.off_frame
        assert plans["retrig_frame"][v] == p.retrig_frame
        assert abs(plans["frequency"][v] - p.frequency) <= 2e-6 * p.frequency
    # midi contract known answers (oscen-lib/src/midi.rs:237-250)
    assert oscen_amd.midi_note_to_freq(69) == 440.0
    assert abs(oscen_amd.midi_note_to_freq(60) - 261.626) < 0.01
    assert abs(oscen_amd.midi_velocity_to_gate(100) - 100 / 127) < 1e-7


def test_unregistered_graph_compiles_with_hiprtc_for_gfx950(lib):
    # the og_create() path of a graph that has no ahead-of-time kernel; compile only, no device
    def extra(g):
        g.node("osc2", "Oscillator::sine", 3.0, 0.1)
        g.connect("osc2.output", "osc.frequency_mod")
        g.connect("filter.output", "out")
    size = _simple(extra).jit_check("gfx950")
    assert size > 4096


def test_multirate_lowering(lib):
    src = oscen_amd.Graph(builtin="sat4x_voice").kernel_source()
    assert "for (int j = 0; j < 4; ++j)" in src and "og::sinc_down<4>" in src
    assert "polyblep_tick<1u" in src.split("for (int j")[1].split("sinc_down")[0]  # osc runs inside the x4 loop
    # an oversampled node may not depend on an outer node that is downstream of the oversampled region
    g = oscen_amd.Graph("bad_rates")
    g.output_stream("out")
    g.node("a", "PolyBlepOscillator::saw", 100.0, 0.5, rate=4)
    g.node("b", "Gain::new", 1.0)
    g.node("c", "HardClip::new", rate=4)
    g.connect("a.output", "b.input").connect("b.output", "c.input").connect("c.output", "out")
    with pytest.raises(oscen_amd.OscenError) as e:
        g.kernel_source()
    assert "depends on the oversampled region" in str(e.value)
    g = oscen_amd.Graph("two_factors")
    g.output_stream("out")
    g.node("a", "PolyBlepOscillator::saw", 100.0, 0.5, rate=4)
    g.node("c", "HardClip::new", rate=2)
    g.connect("a.output", "c.input").connect("c.output", "out")
    with pytest.raises(oscen_amd.OscenError):
        g.kernel_source()


def test_rust_sys_crate_binds_the_whole_header_with_matching_arity():
    """bindings/rust/oscen-gpu-sys (shipped as source: no rustc here) must bind what include/oscen_gpu.h
    declares -- every function, same names, same number of parameters -- and every struct passed by pointer."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "oscen_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(og_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rs = open(os.path.join(root, "bindings", "rust", "oscen-gpu-sys", "src", "lib.rs")).read()
    bound = {}
    for m in re.finditer(r"pub fn (og_\w+)\s*\(([^)]*)\)", rs, flags=re.S):
        args = m.group(2).strip()
        bound[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    assert len(bound) >= 40
    for name, n in bound.items():
        assert name in decl, name + " is not declared in include/oscen_gpu.h"
        assert decl[name] == n, (name, decl[name], n)
    assert sorted(set(decl) - set(bound)) == []  # nothing of the C ABI is left unbound
    for ty in set(re.findall(r"\}\s*(og_\w+)\s*;", hdr)):  # `typedef struct { .. } og_x;`
        assert re.search(r"pub struct %s\b" % ty, rs), ty + " has no #[repr(C)] twin"


def test_kernel_structure_chunk_variants_and_pipelines():
    """What the graph compiler emits for the built-ins: pipelined 2- and 4-wave kernels with
    per-node stages, chunk bodies without stage-end checks / release arithmetic, block-constant work hoisted
    into derive(), delay-line staging with the wave-uniform fast path."""
    import re
    fm = oscen_amd.Graph(builtin="fm_voice").kernel_source()
    assert "voice_block_p2" in fm and "voice_block_p4" in fm
    # round 4: depth-first schedule order (pure dataflow graph: any topological order gives the same bits) -- every
    # envelope sits right in front of its consumer, the two-wave cut falls between the operator chains
    assert re.search(r"// Node order: env3 env2 env1 env_filter op3_osc", fm)  # the reference's Kahn order, for the record
    assert re.search(r"// Schedule \([^)]*\): env3 op3_osc op3_route env2 op2_osc op1_mod_mixer env1 op1_osc env_filter filter_env_gain cutoff_mod filter output_gain", fm)
    assert re.search(r"//   wave 0: env3 op3_osc op3_route env2 op2_osc\b", fm) and re.search(r"//   wave 1: [a-z0-9_ ]*op1_osc env_filter", fm)
    # wave priorities of the four-wave workgroup: behind the hand-off barrier (og_k4_*) first wave 2, middle waves 1, the wave
    # that closes the chunk 0; with the flag hand-off (og_k4w_*, round 6) the other way round, 0 1 2 3
    p4 = fm[fm.index("voice_block_p4"):]
    assert [m for m in re.findall(r"constexpr int BASE_PRIO = FD_T != 0 \? (\d) : (\d);", p4)] == [("0", "2"), ("1", "1"), ("2", "1"), ("3", "0")]
    # ... and a wave runs a chunk that holds an event or a stage end (the checked body) at priority 3, then drops back
    # (three of the four waves hold an envelope; the last one -- filter and bus -- has no checked body)
    assert p4.count("__builtin_amdgcn_s_setprio(3); // the wave on the slow path") == 3
    assert len(re.findall(r"\n {8,}og::set_prio<BASE_PRIO>\(\);", p4)) == 3
    # sticky chunks: every quiet variant of every wave is a loop of its own around its body (hand-off barrier inside), left
    # only when its conditions fail on the next chunk -- three envelope waves x {release-free, release} + the filter wave
    assert p4.count("for (;;) { // sticky: this variant again while its conditions hold") == 7
    assert p4.count("const uint32_t ch1 = ch + 1u, base1 = base + XCH;") == 7 and p4.count("OG_HANDOFF_BARRIER();") == 4 + 7
    # round 6: every hand-off is `flags or barrier` by instantiation; the flag form publishes the stage's progress word and waits
    # for its producers' words and its consumers' release words only
    assert p4.count("if constexpr (FD_T != 0) { og::handoff_publish(&prog[") == 4 + 7 and "og::handoff_wait(&cons[1], seen_c1, ch + 1u - FD_T);" in p4
    sub_src = oscen_amd.Graph(builtin="sub_voice").kernel_source()
    assert "// Node order:" in sub_src
    for k in ("og_k_", "og_k2_", "og_k4_", "og_k4w_"):  # (og_k4w_: the four-wave pipeline with 16-frame hand-offs, round 5)
        assert len(re.findall(r"__global__[^\n]*\b%s[0-9a-f]{16}_(00|10|01|11)\b" % k, fm)) == 4
    assert "voice_block_p4<false, false, 16, 2>" in fm and "constexpr uint32_t XCH = XCH_T ? XCH_T : 8u;" in fm
    # chunk variants: (stage-end checks, release arithmetic[, hand-off prefetch, node steady states])
    assert "og::BoolC<false, false>{}" in fm and "og::BoolC<false, true>{}" in fm and "og::BoolC<true>{}" in fm
    # (round 5: every wave of the pipelines prefetches something at the top of a chunk -- hand-off values or ramp-table rows --
    #  so the third flag is set in all their unrolled bodies)
    assert "og::BoolC<false, false, true, false>{}" in fm and "og::BoolC<true, true, true, false>{}" in fm
    assert "og::row_fetch<XCH>(rv_0, A.ramp_table + (size_t)0 * A.ramp_stride + base);" in fm and "RVP(1, 2)" in fm
    assert "og::adsr_tick<decltype(chk)::release, true>" in fm and ".fc = (float)" in fm  # float countdown kept per chunk
    assert fm.count("if constexpr (decltype(chk)::value)") >= 3
    # the cutoff of FMVoice moves with an envelope: per-tick parameter check; sub_voice's is block-constant
    tick = fm[fm.index("auto tick"):fm.index("auto events")]
    # (round 5: the per-tick check watches the raw input -- one integer compare -- and runs the reference's test only on
    #  frames whose input differs from the previous frame's; q is watched in the kernel variants that read a ramp table)
    # (round 6: 1 / q is formed once per launch -- `inv_q` in derive() -- where q cannot change inside it)
    assert "og::tpt_params_nomod_lazy<RAMPS, true>(" in tick and "og::tpt_params_nomod(" not in tick
    assert "inv_q = og::uniform_f(1.0f / og::clampf(SF(" in fm[fm.index("auto derive"):fm.index("auto tick")]  # (kept in a scalar register)
    sub = oscen_amd.Graph(builtin="sub_voice").kernel_source()
    derive = sub[sub.index("auto derive"):sub.index("auto tick")]
    tick = sub[sub.index("auto tick"):sub.index("auto events")]
    assert "og::tpt_params_nomod(" in derive and "tpt_params" not in tick
    assert "og::polyblep_increment(" in derive and "polyblep_increment" not in tick
    echo = oscen_amd.Graph(builtin="echo_voice").kernel_source()
    assert "og::ring_chunk_begin(A.rings[0]" in echo and "ring_lds[1]" in echo
    assert "voice_block_p2" not in echo  # feedback edge / delay line: ordinary kernel only
    ep = oscen_amd.Graph(builtin="epiano_voice").kernel_source()
    assert "og::ep_bank_tick<TAPS>(" in ep and "og::bus_put<TAPS, !TAPS>" in ep
    assert "og::ep_bank_update(" in ep[ep.index("auto derive"):ep.index("auto tick")]


def test_folded_note_plans_match_the_oracle_generator():
    """oscen_amd.note_plans(span=...) (numpy) == oo_note_plan_scaled (C): the GPU bank and the CPU oracle
    must see the same synthetic note streams at every window length."""
    import ctypes as C

    import oscen_amd
    from tests import oracle_lib as ol

    lib = ol.load()
    for span in (0, 512, 1024, 6400, 47999, 48000, 50176):
        plans = oscen_amd.note_plans(300, first_voice=65500, span=span)
        for i in range(300):
            p = ol.NotePlan()
            lib.oo_note_plan_scaled(oscen_amd.SYNTH_SEED, 65500 + i, span, C.byref(p))
            assert (p.on_frame, p.off_frame, p.retrig_frame) == (
                int(plans["on_frame"][i]), int(plans["off_frame"][i]), int(plans["retrig_frame"][i]))
            assert p.on_frame < p.off_frame < p.retrig_frame
            assert p.frequency == float(plans["frequency"][i])  # bit for bit: both sides go through libm powf


def test_note_event_streams_match_the_oracle_generator_in_both_fold_modes():
    """the frame-sorted event stream per voice ("events" of note_plans) == oo_note_events_for_voice, scale and slice"""
    import ctypes as C

    import oscen_amd
    from tests import oracle_lib as ol

    lib = ol.load()
    for span, fold in ((6400, "slice"), (512, "slice"), (1024, "scale"), (0, "scale"), (50176, "slice")):
        p = oscen_amd.note_plans(500, first_voice=70000, span=span, fold=fold)
        ev_v, ev_f, ev_x = p["events"]
        k = 0
        for i in range(500):
            e = ol.NoteEvents()
            lib.oo_note_events_for_voice(oscen_amd.SYNTH_SEED, 70000 + i, span, ol.FOLD[fold], C.byref(e))
            for j in range(e.n):
                assert (ev_v[k], ev_f[k], ev_x[k]) == (i, e.frame[j], np.float32(e.value[j]))
                k += 1
            assert e.frequency == p["frequency"][i]
        assert k == len(ev_v)
        if fold == "slice" and 0 < span < 48000:  # all three kinds of event inside a short window
            win = ev_f < span
            assert (ev_x[win] > 0).sum() > 0 and (ev_x[win] == 0).sum() > 0


def test_cyclic_fold_is_the_slice_fold_repeated():
    """fold="cyclic" (bench.py, regions shorter than the 1 s score): below 48 000 frames it IS the slice fold; beyond, the
    rotated plan repeats with the period of the score and only the first period carries the held voices' initial note-on"""
    import oscen_amd

    n = 400
    a = oscen_amd.note_plans(n, first_voice=123, span=6400, fold="slice")["events"]
    b = oscen_amd.note_plans(n, first_voice=123, span=6400, fold="cyclic")["events"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    v, f, x = oscen_amd.note_plans(n, first_voice=123, span=48000 * 2 + 5000, fold="cyclic")["events"]
    assert np.all(np.diff(v) >= 0) and f.max() < 3 * 48000
    per = [np.sort(f[(f >= 48000 * k) & (f < 48000 * (k + 1))] - 48000 * k) for k in range(3)]
    assert len(per[1]) == 3 * n and np.array_equal(per[1], per[2])       # 3 events per voice per second, every second alike
    assert len(per[0]) > 3 * n                                            # + the note-on of the voices sounding at the cut
    for voice in range(0, n, 37):                                         # per voice: frames ascending, on / off alternate
        m = v == voice
        assert np.all(np.diff(f[m]) >= 0)
        g = x[m] > 0
        assert not np.any(~g[1:] & ~g[:-1])                               # never two note-offs in a row


def test_rust_shim_block_render_constants_and_calls():
    """bindings/rust/oscen-gpu (source only: no rustc here): BlockRender::NUM_STREAM_INPUTS must be a real constant --
    the trait's default render() asserts `inputs.len() == NUM_STREAM_INPUTS` and loops over it
    (oscen-lib/src/graph/offline.rs:46-75; round 2 had usize::MAX there, which panics) -- and every `sys::og_*` the
    shim calls must be declared by the sys crate and by include/oscen_gpu.h."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "bindings", "rust", "oscen-gpu", "src", "lib.rs")).read()
    code = "\n".join(ln for ln in shim.splitlines() if not ln.lstrip().startswith("//"))
    assert "usize::MAX" not in code
    assert re.search(r"impl<const IN: usize> oscen::BlockRender<f32> for GpuGraph<IN>", code)
    assert re.search(r"const NUM_STREAM_INPUTS: usize = IN;", code)
    assert re.search(r"stream_in_blocks: \[\[f32; MAX_BUS_CHANNELS \* MAX_BLOCK_SIZE\]; IN\]", code)  # room for a Frame<4> input
    assert "og_num_stream_inputs(e) } as usize" in code and "have != IN" in code  # checked against the engine at construction
    sys_rs = open(os.path.join(root, "bindings", "rust", "oscen-gpu-sys", "src", "lib.rs")).read()
    bound = set(re.findall(r"pub fn (og_\w+)\s*\(", sys_rs))
    hdr = open(os.path.join(root, "include", "oscen_gpu.h")).read()
    called = set(re.findall(r"sys::(og_\w+)\s*\(", code))
    assert len(called) >= 20
    for name in called:
        assert name in bound, name + " is called by the shim but not bound by oscen-gpu-sys"
        assert re.search(r"\b%s\s*\(" % name, hdr), name + " is not declared in include/oscen_gpu.h"
    for ty in re.findall(r"sys::(og_\w+)\b(?!\s*\()", code):
        assert re.search(r"pub struct %s\b" % ty, sys_rs), ty
