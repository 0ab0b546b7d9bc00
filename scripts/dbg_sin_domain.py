import sys, numpy as np
sys.path.insert(0, ".")
import oscen_amd
SR = 48000.0
oscen_amd.register_node("SineProbe::new", inputs=[("frequency", "value", 1.0, -1)], outputs=["narrow", "wide"], state=[("k", "f32", 0.0, -1)],
    process="""
    const float t = frequency * (k + 1.0f) * 0.001f + 0.125f;
    narrow = og_sin_turns(t);
    wide = og_sin_turns_wide(t);
    k += 1.0f;
""")
n, frames = 64, 256
scale = np.geomspace(1.0, 4.0e6, n).astype(np.float32)
outs = {}
for port in ("narrow", "wide"):
    g = oscen_amd.Graph("sine_probe_" + port)
    g.input_value("frequency", 1.0, per_voice=True); g.output_stream("out"); g.node("p", "SineProbe::new")
    g.connect("frequency", "p.frequency"); g.connect("p." + port, "out")
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", scale); eng.set_voice_taps(np.arange(n, dtype=np.uint32)); eng.process_block(frames)
    outs[port] = eng.read_voice_taps(frames)
k = np.arange(1, frames + 1, dtype=np.float32)[None, :]
t = ((scale[:, None] * k).astype(np.float32) * np.float32(0.001)).astype(np.float32)
t = (t + np.float32(0.125)).astype(np.float32)
t64 = t.astype(np.float64); ref = np.sin(2 * np.pi * (t64 - np.floor(t64)))
for lo, hi in ((0, 1), (1, 16), (16, 256), (256, 257), (257, 1024), (1024, 1e5), (1e5, 1e9)):
    m = (np.abs(t) > lo) & (np.abs(t) <= hi)
    if m.any():
        print("|t| in (%g, %g]: n %d  narrow err %.3g  wide err %.3g  narrow==0: %d" % (lo, hi, m.sum(), np.max(np.abs(outs["narrow"][m] - ref[m])), np.max(np.abs(outs["wide"][m] - ref[m])), int(np.sum(outs["narrow"][m] == 0))))
