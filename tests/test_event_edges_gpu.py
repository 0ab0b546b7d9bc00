"""Node-to-node event edges inside a voice ("events = clear + copy, last write wins";
oscen-lib/src/graph/static_context.rs:84-155, oscen-macros/src/lib.rs:237-285, event_passthrough.rs): a user node
with an `#[output(event)]` field clocks an AdsrEnvelope, a handler forwards events, EventPassthrough is routing only,
and on fan-in only the last connected source delivers.  Checked against per-sample models over the oracle's nodes."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol
from tests.graph_interp import _make

pytestmark = pytest.mark.gpu
SR = 48000.0
f32 = np.float32

CLOCK = dict(
    inputs=[("period", "value", 100.0, 0), ("velocity", "value", 0.8, -1)], outputs=[], n_ctor_args=1,
    state=[("count", "u32", 0, -1), ("high", "u32", 0, -1)], event_outputs=["trig"],
    process="""
    count += 1u;
    if ((float)count >= period) { count = 0u; high ^= 1u; trig.push(high ? velocity : 0.0f); }
""")
# forwards gate events at half velocity (a handler that pushes), and counts them in a stream output
HALVER = dict(
    inputs=[("input", "event", 0.0, -1)], outputs=["seen"], state=[("n", "f32", 0.0, -1)], event_outputs=["output"],
    process="    seen = n;\n", handlers={"input": "    n += 1.0f; output.push(value * 0.5f);\n"})


def clocked_voice(extra=None):
    oscen_amd.register_node("Clock::new", CLOCK["inputs"], CLOCK["outputs"], CLOCK["process"], state=CLOCK["state"],
                            n_ctor_args=1, event_outputs=CLOCK["event_outputs"])
    g = oscen_amd.Graph("seq")
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("period", 100.0, per_voice=True)
    g.output_stream("out")
    g.node("clk", "Clock::new", 100.0)
    g.node("env", "AdsrEnvelope::new", 0.002, 0.01, 0.6, 0.01)
    g.node("osc", "PolyBlepOscillator::saw", 220.0, 1.0)
    g.connect("period", "clk.period")
    g.connect("frequency", "osc.frequency")
    if extra:
        extra(g)
    else:
        g.connect("clk.trig", "env.gate")
    g.connect("osc.output * env.output", "out")
    return g


def model(freqs, periods, frames, gate_scale=1.0):
    lib = ol.load()
    ref = np.zeros((len(freqs), frames), dtype=np.float32)
    for v in range(len(freqs)):
        env = _make(lib, f32(SR), "AdsrEnvelope::new", [0.002, 0.01, 0.6, 0.01])
        osc = _make(lib, f32(SR), "PolyBlepOscillator::saw", [220.0, 1.0])
        osc.set("frequency", freqs[v])
        count, high = 0, 0
        for f in range(frames):
            count += 1
            trig = None
            if f32(count) >= f32(periods[v]):
                count, high = 0, high ^ 1
                trig = f32(0.8) if high else f32(0.0)
            osc.process()
            if trig is not None:
                env.gate(f32(trig * f32(gate_scale)))
            env.process()
            ref[v, f] = f32(osc.get("output") * env.get("output"))
    return ref


def render(g, freqs, periods, frames, blocks=(256, 300, 212)):
    n = len(freqs)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_values("period", periods)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    got = []
    for b in blocks:
        eng.process_block(b)
        got.append(eng.read_voice_taps(b))
    assert eng.pipeline_depth == 1  # the ordinary kernel: node-to-node events do not cross pipeline waves
    return np.concatenate(got, axis=1)


def close(got, ref):
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5 and float(np.abs(ref).max()) > 0.1, err


def test_a_clock_node_gates_the_envelope_of_its_voice():
    n, frames = 96, 768
    freqs = np.linspace(80.0, 2500.0, n).astype(np.float32)
    periods = (37 + 5 * np.arange(n)).astype(np.float32)  # every voice its own tempo: stage ends and gates on any frame
    got = render(clocked_voice(), freqs, periods, frames)
    close(got, model(freqs, periods, frames))


def test_handler_that_pushes_and_passthrough_chain():
    """clk.trig -> half.input (on_input pushes value/2) -> pass1 -> pass2 -> env.gate"""
    oscen_amd.register_node("Halver::new", HALVER["inputs"], HALVER["outputs"], HALVER["process"], state=HALVER["state"],
                            handlers=HALVER["handlers"], event_outputs=HALVER["event_outputs"])

    def wire(g):
        g.node("half", "Halver::new")
        g.node("pass1", "EventPassthrough::new")
        g.node("pass2", "EventPassthrough::new")
        g.connect("clk.trig", "half.input")
        g.connect("half.output", "pass1.input")
        g.connect("pass1.output", "pass2.input")
        g.connect("pass2.output", "env.gate")

    n, frames = 70, 768
    freqs = np.linspace(100.0, 1500.0, n).astype(np.float32)
    periods = (50 + 3 * np.arange(n)).astype(np.float32)
    g = clocked_voice(wire)
    assert "EventPassthrough" not in oscen_amd.Graph(dsl=g.to_dsl(), per_voice=("frequency", "period")).kernel_source()
    close(render(g, freqs, periods, frames), model(freqs, periods, frames, gate_scale=0.5))


def test_event_fan_in_is_last_write_wins():
    """two sources into env.gate: only the last connected one delivers (clear + copy) -- here the graph's own gate input
    is connected first and the clock second, so scheduled gate events never reach the envelope"""
    def wire(g):
        g.input_event("gate")
        g.connect("gate", "env.gate")
        g.connect("clk.trig", "env.gate")

    n, frames = 64, 768
    freqs = np.linspace(100.0, 1500.0, n).astype(np.float32)
    periods = (41 + 7 * np.arange(n)).astype(np.float32)
    g = clocked_voice(wire)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_values("period", periods)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    for v in range(n):
        eng.schedule_voice_event("gate", v, 10 + v, 1.0)
    got = []
    for b in (256, 256, 256):
        eng.process_block(b)
        got.append(eng.read_voice_taps(b))
    close(np.concatenate(got, axis=1), model(freqs, periods, frames))


def test_passthrough_between_a_graph_event_input_and_a_node_is_free():
    """gate -> pass.input; pass.output -> env.gate is the kernel of gate -> env.gate (same hash, pipelines still on)"""
    def build(with_pass):
        g = oscen_amd.Graph("pt")
        g.input_value("frequency", 220.0, per_voice=True)
        g.input_event("gate")
        g.output_stream("out")
        g.node("env", "AdsrEnvelope::new", 0.002, 0.01, 0.6, 0.01)
        g.node("osc", "PolyBlepOscillator::saw", 220.0, 1.0)
        g.connect("frequency", "osc.frequency")
        if with_pass:
            g.node("pass", "EventPassthrough::new")
            g.connect("gate", "pass.input")
            g.connect("pass.output", "env.gate")
        else:
            g.connect("gate", "env.gate")
        g.connect("osc.output * env.output", "out")
        return g

    a, b = oscen_amd.Engine(build(True), 64, sample_rate=SR), oscen_amd.Engine(build(False), 64, sample_rate=SR)
    assert a.kernel_hash == b.kernel_hash
    for e in (a, b):
        for v in range(64):
            e.schedule_voice_event("gate", v, 3 * v, 0.9)
    assert np.array_equal(a.process_block(256), b.process_block(256))


def test_graph_event_outputs_reach_the_host_in_frame_voice_push_order():
    """`output ticks: event;` fed by a node's #[output(event)] field (EventOutput, oscen-lib/src/graph/types.rs:137-241;
    the reference's caller iterates `graph.ticks` after process_block): the events of every voice leave through the
    device log with their exact frame; a third push on one frame is dropped and counted (the in-voice queue holds
    OG_NODE_EVENTS_PER_FRAME = 2)."""
    oscen_amd.register_node(
        "Burst::new", inputs=[("period", "value", 100.0, 0), ("pushes", "value", 1.0, 1)], outputs=["level"], n_ctor_args=2,
        state=[("count", "u32", 0, -1), ("fired", "f32", 0.0, -1)], event_outputs=["tick"],
        process="""
    count += 1u;
    if ((float)count >= period) {
        count = 0u;
        fired += 1.0f;
        for (int k = 0; k < (int)pushes; ++k) tick.push(fired * 10.0f + (float)k);
    }
    level = fired;
""")
    try:
        g = oscen_amd.Graph("bursts")
        g.input_value("period", 100.0, per_voice=True)
        g.input_value("pushes", 1.0, per_voice=True)
        g.output_stream("out")
        g.output_event("ticks")
        g.node("b", "Burst::new", 100.0, 1.0)
        g.connect("period", "b.period")
        g.connect("pushes", "b.pushes")
        g.connect("b.level", "out")
        g.connect("b.tick", "ticks")
        n = 200
        periods = (37 + (np.arange(n) * 7) % 90).astype(np.float32)
        pushes = (1 + np.arange(n) % 3).astype(np.float32)  # 1, 2 or 3 pushes per firing: the third is dropped
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        assert eng.lib.og_num_event_outputs(eng.h) == 1 and eng.event_output_index("ticks") == 0
        eng.set_voice_values("period", periods)
        eng.set_voice_values("pushes", pushes)
        got = []
        blocks = (256, 100, 412, 256)
        for i, b in enumerate(blocks):
            eng.process_block(b)
            if i % 2 == 1:  # drained every other block: events of two blocks arrive together, still ordered
                ev, over = eng.read_output_events()
                assert over == 0
                got.append(ev)
        ev, over = eng.read_output_events()
        assert over == 0 and len(ev) == 0
        got = np.concatenate(got)
        # model
        want, lost = [], 0
        total = sum(blocks)
        for v in range(n):
            count, fired = 0, 0.0
            for f in range(total):
                count += 1
                if np.float32(count) >= periods[v]:
                    count = 0
                    fired += 1.0
                    for k in range(int(pushes[v])):
                        if k < 2:
                            want.append((f, v, np.float32(np.float32(fired * 10.0) + np.float32(k))))
                        else:
                            lost += 1
        want.sort(key=lambda t: (t[0], t[1]))  # (stable: a voice's two pushes of one frame keep their order)
        assert len(got) == len(want) and len(want) > 1000
        assert np.array_equal(got["frame"], np.array([w[0] for w in want], dtype=np.uint64))
        assert np.array_equal(got["voice"], np.array([w[1] for w in want], dtype=np.uint32))
        assert np.array_equal(got["value"], np.array([w[2] for w in want], dtype=np.float32))
        assert np.all(got["output"] == 0)
        assert eng.events_dropped == lost and lost > 50
        # a log too small for what arrives between two reads: the excess is counted, not lost silently
        import os
        os.environ["OSCEN_GPU_EXPERIMENTAL"] = "1"
        os.environ["OSCEN_GPU_OUT_EVENTS"] = "64"
        try:
            small = oscen_amd.Engine(g, n, sample_rate=SR)
        finally:
            del os.environ["OSCEN_GPU_OUT_EVENTS"]
            del os.environ["OSCEN_GPU_EXPERIMENTAL"]
        small.set_voice_values("period", periods)
        small.set_voice_values("pushes", np.ones(n, dtype=np.float32))
        small.process_block(512)
        ev, over = small.read_output_events()
        assert len(ev) == 64 and over > 0
    finally:
        oscen_amd.unregister_node("Burst::new")


@pytest.mark.parametrize("capacity", [5, 32])
def test_event_queue_capacity_up_to_the_reference_32(capacity):
    """`event_queue_capacity` of a node type (the reference's EventOutput is an ArrayVec<EventInstance, 32>,
    graph/types.rs:18): a node that pushes k events on one frame keeps min(k, capacity) of them, in push order, both for
    the graph's event output (host log) and for a node-to-node event edge (the consumer's handler runs once per event);
    only what exceeds the capacity is dropped, and counted"""
    oscen_amd.register_node(
        "QBurst::new", inputs=[("period", "value", 100.0, 0), ("pushes", "value", 1.0, 1)], outputs=["level"], n_ctor_args=2,
        state=[("count", "u32", 0, -1), ("fired", "f32", 0.0, -1)], event_outputs=["tick"], event_capacity=capacity,
        process="""
    count += 1u;
    if ((float)count >= period) {
        count = 0u;
        fired += 1.0f;
        for (int k = 0; k < (int)pushes; ++k) tick.push(fired * 100.0f + (float)k);
    }
    level = fired;
""")
    oscen_amd.register_node(
        "QSum::new", inputs=[("trig", "event", 0.0, -1)], outputs=["total"], state=[("acc", "f32", 0.0, -1), ("calls", "f32", 0.0, -1)],
        handlers={"trig": "    acc += value;\n    calls += 1.0f;\n"}, process="    total = acc;\n")
    try:
        g = oscen_amd.Graph("qbursts")
        g.input_value("period", 100.0, per_voice=True)
        g.input_value("pushes", 1.0, per_voice=True)
        g.output_stream("out")
        g.output_event("ticks")
        g.node("b", "QBurst::new", 100.0, 1.0)
        g.node("s", "QSum::new")
        g.connect("period", "b.period")
        g.connect("pushes", "b.pushes")
        g.connect("b.tick", "s.trig")
        g.connect("b.tick", "ticks")
        g.connect("s.total", "out")
        n = 130
        periods = (23 + (np.arange(n) * 5) % 60).astype(np.float32)
        pushes = (np.arange(n) % (capacity + 3)).astype(np.float32)  # 0 .. capacity + 2 pushes per firing
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("period", periods)
        eng.set_voice_values("pushes", pushes)
        eng.set_voice_taps(list(range(n)))
        frames = 300
        eng.process_block(frames)
        taps = eng.read_voice_taps(frames)
        ev, over = eng.read_output_events()
        assert over == 0
        want, lost = [], 0
        acc = np.zeros(n, dtype=np.float32)
        calls = np.zeros(n)
        for v in range(n):
            count, fired = 0, 0.0
            for f in range(frames):
                count += 1
                if np.float32(count) >= periods[v]:
                    count = 0
                    fired += 1.0
                    for k in range(int(pushes[v])):
                        if k < capacity:
                            x = np.float32(np.float32(fired * 100.0) + np.float32(k))
                            want.append((f, v, x))
                            acc[v] = np.float32(acc[v] + x)
                            calls[v] += 1
                        else:
                            lost += 1
        want.sort(key=lambda t: (t[0], t[1]))
        assert len(ev) == len(want) and len(want) > 300
        assert np.array_equal(ev["frame"], np.array([w[0] for w in want], dtype=np.uint64))
        assert np.array_equal(ev["voice"], np.array([w[1] for w in want], dtype=np.uint32))
        assert np.array_equal(ev["value"], np.array([w[2] for w in want], dtype=np.float32))
        assert np.array_equal(taps[:, -1], acc)                                   # the consumer saw every kept event, in order
        assert np.array_equal(eng.read_state_field("s.calls"), calls.astype(np.float32))
        assert eng.events_dropped == lost and lost > 0
    finally:
        oscen_amd.unregister_node("QBurst::new")
        oscen_amd.unregister_node("QSum::new")


def test_event_output_of_an_oversampled_node():
    """a `* 2` / `* 4` node pushing into a graph event output: its queue is cleared after every inner tick, so its events
    are logged there -- every inner tick of outer frame f under frame f, in tick order (the inner -> outer rescale of a
    cross-rate event edge, ir/lower.rs:846-852)"""
    oscen_amd.register_node(
        "OsBurst::new", inputs=[("period", "value", 10.0, 0)], outputs=["level"], n_ctor_args=1,
        state=[("count", "u32", 0, -1), ("fired", "f32", 0.0, -1)], event_outputs=["tick"],
        process="    count += 1u;\n    if ((float)count >= period) { count = 0u; fired += 1.0f; tick.push(fired); }\n    level = fired;\n")
    try:
        for factor in (2, 4):
            g = oscen_amd.Graph(dsl=f"""name: OsEv; input period: value = 10.0; output out: stream; output ticks: event;
                nodes {{ b = OsBurst::new(10.0) * {factor}; }}
                connections {{ period -> b.period; [latch] b.level -> out; b.tick -> ticks; }}""", per_voice=["period"])
            n, frames = 40, 300
            periods = (1 + (np.arange(n) * 3) % 17).astype(np.float32)  # period 1: an event on EVERY inner tick
            eng = oscen_amd.Engine(g, n, sample_rate=SR)
            eng.set_voice_values("period", periods)
            eng.process_block(frames)
            ev, over = eng.read_output_events()
            assert over == 0
            want = []
            for v in range(n):
                count, fired = 0, 0.0
                for f in range(frames):
                    for _ in range(factor):
                        count += 1
                        if np.float32(count) >= periods[v]:
                            count = 0
                            fired += 1.0
                            want.append((f, v, np.float32(fired)))
            want.sort(key=lambda t: (t[0], t[1]))
            assert len(ev) == len(want) > 2000
            assert np.array_equal(ev["frame"], np.array([w[0] for w in want], dtype=np.uint64))
            assert np.array_equal(ev["voice"], np.array([w[1] for w in want], dtype=np.uint32))
            assert np.array_equal(ev["value"], np.array([w[2] for w in want], dtype=np.float32))
            assert eng.events_dropped == 0
    finally:
        oscen_amd.unregister_node("OsBurst::new")


def test_event_edge_between_two_oversampled_nodes_is_delivered_once_per_outer_tick():
    """Both ends of an event edge inside the oversampled region (round 4).  The reference copies the producer's queue into
    the consumer's on every inner tick but runs the consumer's handlers -- process_event_inputs() -- once per OUTER tick,
    in front of the inner loop (step 6a of the multirate body, oscen-graph-compiler/src/codegen/emit_frame.rs:150-160), and
    the producer's queue is cleared there too (oscen-macros/src/lib.rs:266-285): what an inner node pushes during outer
    frame f reaches an inner consumer at the START of frame f + 1, all of it, in push order.  (Rounds 2-3 delivered on
    the same inner tick.)  The queue is part of the voice's state: a block boundary between push and delivery changes
    nothing."""
    oscen_amd.register_node(
        "R4Clock::new", inputs=[("every", "value", 3.0, 0)], outputs=[], n_ctor_args=1, state=[("t", "u32", 0, -1)], event_outputs=["tick"],
        process="    t += 1u;\n    if ((float)(t % max((uint32_t)every, 1u)) == 0.0f) tick.push((float)t);\n")  # (max: lanes beyond the bank hold every = 0)
    oscen_amd.register_node(
        "R4Counter::new", inputs=[("trig", "event", 0.0, -1)], outputs=["out"],
        state=[("received", "u32", 0, -1), ("sum", "f32", 0.0, -1), ("seen_at", "u32", 0, -1), ("ticks", "u32", 0, -1)],
        handlers={"trig": "    received += 1u;\n    sum += value;\n    seen_at = ticks;\n"},
        process="    ticks += 1u;\n    out = sum;\n")
    try:
        for factor in (2, 4):
            g = oscen_amd.Graph(dsl=f"""name: InnerEv; input every: value = 3.0; output out: stream;
                nodes {{ clk = R4Clock::new(3.0) * {factor}; cnt = R4Counter::new() * {factor}; }}
                connections {{ every -> clk.every; clk.tick -> cnt.trig; [latch] cnt.out -> out; }}""", per_voice=["every"])
            n, frames = 12, 64
            every = (1 + np.arange(n) % 5).astype(np.float32)  # 1: a push on every inner tick (factor pushes per outer frame)
            # model: per outer frame -- 6a: deliver last frame's pushes, clear; then `factor` inner ticks
            want_out = np.zeros((n, frames), dtype=np.float32)
            want = []
            for v in range(n):
                t, ticks, received, ssum, seen_at, queue = 0, 0, 0, np.float32(0.0), 0, []
                for f in range(frames):
                    for x in queue:
                        received += 1
                        ssum = np.float32(ssum + np.float32(x))
                        seen_at = ticks
                    queue = []
                    for _ in range(factor):
                        t += 1
                        if t % int(every[v]) == 0:
                            queue.append(t)
                        ticks += 1
                    want_out[v, f] = ssum
                want.append((received, ssum, seen_at, ticks))

            def run(blocks):
                eng = oscen_amd.Engine(g, n, sample_rate=SR)
                eng.set_voice_values("every", every)
                eng.set_voice_taps(list(range(n)))
                got = []
                for b in blocks:
                    eng.process_block(b)
                    got.append(eng.read_voice_taps(b))
                return eng, np.concatenate(got, axis=1)

            eng, got = run([frames])
            assert np.array_equal(got, want_out), factor
            assert np.array_equal(eng.read_state_field("cnt.received", dtype=np.uint32), np.array([w[0] for w in want], dtype=np.uint32))
            assert np.array_equal(eng.read_state_field("cnt.seen_at", dtype=np.uint32), np.array([w[2] for w in want], dtype=np.uint32))
            assert np.array_equal(eng.read_state_field("cnt.ticks", dtype=np.uint32), np.full(n, factor * frames, dtype=np.uint32))
            assert eng.events_dropped == 0
            # delivery happens a whole outer frame after the push: with a push on every tick the first frame still reads 0
            assert want_out[0, 0] == 0.0 and want_out[0, 1] > 0.0
            # block boundaries between push and delivery (the queue travels in the state planes)
            _, got2 = run([1, 7, 1, 23, 32])
            assert np.array_equal(got2, got), factor
        # ACROSS the rate boundary (the reference's event drains, emit_frame.rs:341-374): an outer node's events reach an
        # inner node at step 6a of the SAME outer frame (outer nodes run first); an inner node's events -- all N ticks' --
        # reach an outer node behind the region when it runs, in the same frame
        factor, n, frames = 4, 12, 48
        every = (1 + np.arange(n) % 5).astype(np.float32)
        for shape in ("outer_to_inner", "inner_to_outer"):
            ck, ct = ("", " * 4") if shape == "outer_to_inner" else (" * 4", "")
            pol = "[latch] " if shape == "outer_to_inner" else ""
            g = oscen_amd.Graph(dsl=f"""name: CrossEv; input every: value = 3.0; output out: stream;
                nodes {{ clk = R4Clock::new(3.0){ck}; cnt = R4Counter::new(){ct}; }}
                connections {{ every -> clk.every; clk.tick -> cnt.trig; {pol}cnt.out -> out; }}""", per_voice=["every"])
            want_out = np.zeros((n, frames), dtype=np.float32)
            want_recv, want_seen = [], []
            for v in range(n):
                t, ticks, received, ssum, seen_at = 0, 0, 0, np.float32(0.0), 0
                for f in range(frames):
                    queue = []
                    for _ in range(1 if shape == "outer_to_inner" else factor):  # the clock's ticks of this outer frame
                        t += 1
                        if t % int(every[v]) == 0:
                            queue.append(t)
                    for x in queue:  # delivered in the same outer frame, before the counter's own ticks
                        received += 1
                        ssum = np.float32(ssum + np.float32(x))
                        seen_at = ticks
                    ticks += factor if shape == "outer_to_inner" else 1
                    want_out[v, f] = ssum
                want_recv.append(received)
                want_seen.append(seen_at)
            eng = oscen_amd.Engine(g, n, sample_rate=SR)
            eng.set_voice_values("every", every)
            eng.set_voice_taps(list(range(n)))
            got = []
            for b in (5, 11, 32):
                eng.process_block(b)
                got.append(eng.read_voice_taps(b))
            assert np.array_equal(np.concatenate(got, axis=1), want_out), shape
            assert np.array_equal(eng.read_state_field("cnt.received", dtype=np.uint32), np.array(want_recv, dtype=np.uint32)), shape
            assert np.array_equal(eng.read_state_field("cnt.seen_at", dtype=np.uint32), np.array(want_seen, dtype=np.uint32)), shape
            assert eng.events_dropped == 0
    finally:
        oscen_amd.unregister_node("R4Clock::new")
        oscen_amd.unregister_node("R4Counter::new")


def test_a_node_to_node_event_carries_the_frame_offset_its_producer_gave_it():
    """EventInstance::frame_offset (oscen-lib/src/graph/types.rs:129-132) on IN-VOICE events: the producer's value travels
    with the event -- not the frame the handler happens to run on -- and a rate boundary rescales it like the reference's
    drains (oscen-graph-compiler/src/codegen/emit_edge.rs:86-99: outer -> inner `saturating_mul(N)`, inner -> outer `/ N`).
    `out.push_at(offset, value)` = try_push(EventInstance { frame_offset, Scalar(value) }); `push(value)` leaves it 0."""
    oscen_amd.register_node(
        "OffSrc::new", inputs=[("period", "value", 10.0, 0), ("base", "value", 0.0, -1)], outputs=[], n_ctor_args=1,
        state=[("count", "u32", 0, -1), ("fired", "u32", 0, -1)], event_outputs=["trig"],
        process="""
    count += 1u;
    if ((float)count >= period) { count = 0u; fired += 1u; trig.push_at((uint32_t)base + 3u * fired, 0.25f * (float)fired); }
""")
    oscen_amd.register_node(
        "OffCap::new", inputs=[("input", "event", 0.0, -1)], outputs=["seen"],
        state=[("cap", "f32", -1.0, -1), ("val", "f32", 0.0, -1), ("hits", "f32", 0.0, -1)],
        handlers={"input": "    cap = (float)frame_offset;\n    val = value;\n    hits += 1.0f;\n"}, process="    seen = cap;\n")
    # a forwarding handler: what it pushes with push() starts again at offset 0, with push_at(frame_offset, ..) it passes on
    oscen_amd.register_node(
        "OffFwd::new", inputs=[("input", "event", 0.0, -1)], outputs=[], event_outputs=["same", "zero"],
        handlers={"input": "    same.push_at(frame_offset, value);\n    zero.push(value);\n"}, process="")
    try:
        n, frames = 67, 96
        periods = (7 + np.arange(n) % 13).astype(np.float32)
        bases = (100 * (np.arange(n) % 5)).astype(np.float32)
        fired = frames // periods.astype(np.int64)          # pushes per voice in `frames` frames
        last_off = bases.astype(np.int64) + 3 * fired        # the producer's offset of its LAST push
        for src_rate, dst_rate, scale in (("", "", lambda o: o), ("", " * 4", lambda o: np.minimum(o * 4, 0xFFFFFFFF)), (" * 4", "", lambda o: o // 4)):
            inner_src = bool(src_rate)
            g = oscen_amd.Graph(dsl=f"""name: OffEdge; input period: value = 10.0; input base: value = 0.0; output out: stream;
                nodes {{ src = OffSrc::new(10.0){src_rate}; fwd = OffFwd::new(){src_rate}; a = OffCap::new(){dst_rate}; b = OffCap::new(){dst_rate}; z = OffCap::new(){dst_rate}; }}
                connections {{ period -> src.period; base -> src.base; src.trig -> a.input; src.trig -> fwd.input; fwd.same -> b.input; fwd.zero -> z.input;
                               {'[latch] ' if dst_rate else ''}a.seen + b.seen + z.seen -> out; }}""",
                                per_voice=["period", "base"])
            eng = oscen_amd.Engine(g, n, sample_rate=SR)
            # an oversampled producer ticks N times per frame: its counter reaches `period` N times as often
            eng.set_voice_values("period", periods * (4 if inner_src else 1))
            eng.set_voice_values("base", bases)
            for b in (64, 32):
                eng.process_block(b)
            hits = eng.read_state_field("a.hits")
            assert np.array_equal(hits, fired.astype(np.float32)), (src_rate, dst_rate)
            want = scale(last_off).astype(np.float32)
            want[fired == 0] = -1.0
            assert np.array_equal(eng.read_state_field("a.cap"), want), (src_rate, dst_rate)   # the straight edge
            # ... and through the forwarding handler.  With BOTH src and fwd in the oversampled region the handler runs at
            # step 6a of the NEXT outer frame (codegen/emit_frame.rs:150-160), so b has not seen a push of the last frame yet
            fired_b = fired
            if inner_src:
                p4 = (periods * 4).astype(np.int64)
                fired_b = np.array([sum(1 for k in range(1, int(fired[v]) + 1) if (k * p4[v] - 1) // 4 <= frames - 2) for v in range(n)])
            want_b = scale(bases.astype(np.int64) + 3 * fired_b).astype(np.float32)
            want_b[fired_b == 0] = -1.0
            assert np.array_equal(eng.read_state_field("b.cap"), want_b), (src_rate, dst_rate, "through the forwarder")
            zero = np.where(fired_b > 0, 0.0, -1.0).astype(np.float32)
            assert np.array_equal(eng.read_state_field("z.cap"), zero)  # push(): frame_offset 0
            assert np.array_equal(eng.read_state_field("a.val"), (0.25 * fired).astype(np.float32))
    finally:
        for t in ("OffSrc::new", "OffCap::new", "OffFwd::new"):
            oscen_amd.unregister_node(t)
