"""MIDI bytes -> allocator -> voice bank on the GPU, against the oracle fed with the same routed events."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_midi_driven_bank_matches_oracle():
    n, frames, sr = 8, 256, 48000.0
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=sr)
    eng.set_voice_taps(list(range(n)))
    midi = oscen_amd.Midi(eng)
    twin = oscen_amd.Midi(n_voices=n)          # same allocator, detached: tells us what each voice receives
    bank = ol.Bank(ol.BANK_FM, n, sr)
    rng = np.random.default_rng(5)
    held = []
    worst = 0.0
    for b in range(20):
        for _ in range(rng.integers(0, 4)):
            fo = int(rng.integers(0, frames))
            if held and rng.random() < 0.45:
                note = held.pop(int(rng.integers(0, len(held))))
                msg = [0x80, note, 0]
            else:
                note = int(rng.integers(40, 90))
                held.append(note)
                msg = [0x90, note, int(rng.integers(30, 128))]
            midi.send(msg, fo)
            twin.send(msg, fo)
        twin.flush()
        for voice, fo, hz, gate in twin.pop_outputs():
            if hz is not None:
                bank.push_event(voice, fo, ol.EV_FREQ, hz)
            bank.push_event(voice, fo, ol.EV_GATE, gate)
        bus = midi.process_block(frames)
        taps = eng.read_voice_taps(frames)
        ref_bus, ref = bank.process_block(frames, taps=list(range(n)))
        worst = max(worst, float(np.max(np.abs(taps - ref) / np.maximum(1.0, np.abs(ref)))))
    assert np.max(np.abs(ref)) > 0.01
    observed.note(worst)
    assert worst <= 1e-5, worst


def test_midi_into_a_cluster_matches_midi_into_one_engine():
    """og_midi_create_cluster: one allocator over the global voice ids of a 3-shard bank; the same MIDI stream into a
    single engine of the same size gives the same voices (taps bit for bit) and the same bus up to the association of
    the cross-shard sum"""
    n, frames, sr = 96, 256, 48000.0
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=sr)
    cl = oscen_amd.Cluster("fm_voice", n, [0, 0, 0], sample_rate=sr)
    m_eng, m_cl = oscen_amd.Midi(eng), oscen_amd.Midi(cl)
    for m in (m_eng, m_cl):
        m.set_queue_capacity(256)
    shards = [cl.shard(s) for s in range(3)]
    for sh in shards:
        sh.set_voice_taps(np.arange(sh.n_voices, dtype=np.uint32))
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    rng = np.random.default_rng(11)
    held = []
    for b in range(24):
        for _ in range(int(rng.integers(0, 40))):    # enough notes to run the allocator through all three shards and into stealing
            fo = int(rng.integers(0, frames))
            if held and rng.random() < 0.4:
                msg = [0x80, held.pop(int(rng.integers(0, len(held)))), 0]
            else:
                note = int(rng.integers(36, 97))
                held.append(note)
                msg = [0x90, note, int(rng.integers(30, 128))]
            m_eng.send(msg, fo)
            m_cl.send(msg, fo)
        a = m_eng.process_block(frames)
        t_eng = eng.read_voice_taps(frames)
        c = m_cl.process_block(frames)[:, 0]
        t_cl = np.concatenate([sh.read_voice_taps(frames) for sh in shards], axis=0)
        assert np.array_equal(t_eng, t_cl)
        assert np.max(np.abs(a[:, 0] - c)) <= 1e-5 * max(1.0, float(np.abs(a).max()))
    assert np.abs(a).max() > 1e-2
    for v in (0, 40, 95):
        assert m_eng.voice_state(v) == m_cl.voice_state(v)
    assert cl.lib.og_midi_process_block_async(m_cl.h, frames, None) == oscen_amd.OG_E_UNSUPPORTED


def poly_wrapper_text(builtin, voice_type, voice_out, n, post=None):
    """A poly wrapper in the reference's DSL, GENERATED from the inputs of a built-in voice-bank graph: MidiParser ->
    VoiceAllocator<n> -> [MidiVoiceHandler; n] -> [voice_type; n] -> sum [-> post-mix node] (the shape of
    examples/fm-synth/src/lib.rs:22-131; the reference's own text is exercised on the CPU side where the checkout
    exists, tests/test_dsl_corpus_cpu.py)"""
    import re

    decl = oscen_amd.Graph(builtin=builtin).to_dsl()
    inputs = [ln.split("//")[0].strip() for ln in decl.splitlines() if ln.startswith("input ")]
    inputs = [ln for ln in inputs if not re.match(r"input (frequency|gate)\b", ln)]
    names = [re.match(r"input (\w+)", ln).group(1) for ln in inputs]
    post_inputs = [] if post is None else list(post["params"].values())
    t = ["name: GeneratedPoly;", "input midi_in: event;"] + inputs
    t += ["output out: stream%s;" % (": Frame<2>" if post else "")]
    t += ["nodes {", "  midi_parser = MidiParser::new();", "  voice_allocator = VoiceAllocator::<%d>::new();" % n,
          "  voice_handlers = [MidiVoiceHandler::new(); %d];" % n, "  voices = [%s::new(); %d];" % (voice_type, n)]
    if post:
        t += ["  %s = %s();" % (post["name"], post["ctor"])]
    t += ["}", "connections {", "  midi_in -> midi_parser.midi_in;", "  midi_parser.note_on -> voice_allocator.note_on;",
          "  midi_parser.note_off -> voice_allocator.note_off;", "  voice_allocator.voices -> voice_handlers.note_on;",
          "  voice_allocator.voices -> voice_handlers.note_off;", "  voice_handlers.frequency -> voices.frequency;",
          "  voice_handlers.gate -> voices.gate;"]
    t += ["  %s -> voices.%s;" % (nm, nm) for nm in names if nm not in post_inputs]
    if post:
        t += ["  voices.%s -> %s.input;" % (voice_out, post["name"])]
        t += ["  %s -> %s.%s;" % (src, post["name"], port) for port, src in post["params"].items()]
        t += ["  %s.output -> out;" % post["name"]]
    else:
        t += ["  voices.%s -> out;" % voice_out]
    t += ["}"]
    return "\n".join(t)


def test_midi_into_a_parsed_poly_wrapper_matches_the_oracle():
    """the fm-synth poly wrapper as DSL TEXT through og_graph_parse: it lowers to the ahead-of-time fm_voice kernel (no
    JIT), og_graph_poly_info names the inputs the MIDI front end drives, and MIDI played into it matches the oracle bank
    fed the same routed events -- with a ramped wrapper parameter moving on the way"""
    n, frames, sr = 8, 256, 48000.0
    text = poly_wrapper_text("fm_voice", "FMVoice", "audio_out", n)
    g = oscen_amd.Graph(dsl=text)
    info = g.poly_info()
    assert info == {"declared_voices": n, "frequency_input": "frequency", "gate_input": "gate"}
    bank_voices = 24  # the bank size is og_create's, not the N of the text
    eng = oscen_amd.Engine(g, bank_voices, sample_rate=sr)
    assert eng.lib.og_kernel_is_jit(eng.h) == 0 and eng.kernel_hash == oscen_amd.Engine("fm_voice", 8, sample_rate=sr).kernel_hash
    eng.set_voice_taps(list(range(bank_voices)))
    midi = oscen_amd.Midi(eng, frequency_input=info["frequency_input"], gate_input=info["gate_input"])
    twin = oscen_amd.Midi(n_voices=bank_voices)
    bank = ol.Bank(ol.BANK_FM, bank_voices, sr)
    rng = np.random.default_rng(17)
    held, worst = [], 0.0
    for b in range(24):
        if b == 5:  # set_filter_cutoff: the wrapper declares [ramp: 2205]
            eng.set_value("filter_cutoff", 5000.0)
            bank.set_value(ol.FM_PARAMS.index("filter_cutoff"), 5000.0)
        for _ in range(int(rng.integers(0, 5))):
            fo = int(rng.integers(0, frames))
            if held and rng.random() < 0.4:
                msg = [0x80, held.pop(int(rng.integers(0, len(held)))), 0]
            else:
                note = int(rng.integers(40, 90))
                held.append(note)
                msg = [0x90, note, int(rng.integers(30, 128))]
            midi.send(msg, fo)
            twin.send(msg, fo)
        twin.flush()
        for voice, fo, hz, gate in twin.pop_outputs():
            if hz is not None:
                bank.push_event(voice, fo, ol.EV_FREQ, hz)
            bank.push_event(voice, fo, ol.EV_GATE, gate)
        midi.process_block(frames)
        taps = eng.read_voice_taps(frames)
        _, ref = bank.process_block(frames, taps=list(range(bank_voices)))
        worst = max(worst, float(np.max(np.abs(taps - ref) / np.maximum(1.0, np.abs(ref)))))
    assert np.max(np.abs(ref)) > 0.01 and worst <= 1e-5, worst


def test_parsed_epiano_wrapper_equals_the_builtin_bank():
    """the electric piano's wrapper shape (voices -> sum -> Tremolo -> Frame<2>) as text: same kernel as the built-in
    epiano_voice bank, same stereo bus bit for bit"""
    n, sr = 40, 48000.0
    text = poly_wrapper_text("epiano_voice", "ElectricPianoVoiceNode", "output", 16,
                             post={"name": "tremolo", "ctor": "Tremolo::new", "params": {"depth": "vibrato_intensity", "rate": "vibrato_speed"}})
    g = oscen_amd.Graph(dsl=text)
    assert g.poly_info()["declared_voices"] == 16
    a = oscen_amd.Engine(g, n, sample_rate=sr)
    b = oscen_amd.Engine("epiano_voice", n, sample_rate=sr)
    assert a.kernel_hash == b.kernel_hash and a.channels == 2
    plans = oscen_amd.note_plans(n, span=2048)
    for e in (a, b):
        oscen_amd.schedule_note_plans(e, plans, total_frames=2048)
        e.set_value("vibrato_speed", 6.0)
    ra, rb = a.render(2048, 256), b.render(2048, 256)
    assert np.array_equal(ra, rb) and np.abs(ra).max() > 1e-3
