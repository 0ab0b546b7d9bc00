"""The reference's own custom nodes (examples/fm-synth/src/nodes/*.rs, examples/pivot/src/vca.rs,
examples/oversampled-saturator/src/main.rs:32-62) written against the plug-in API (og_register_node) the way a user
of the reference writes `#[derive(Node)]` structs: ports, private fields, process() body.  Test helper."""
import oscen_amd

# fm_operator.rs:12-76
FM_OPERATOR = dict(
    inputs=[("base_freq", "value", 440.0, -1), ("ratio", "value", 1.0, -1), ("phase_mod", "stream", 0.0, -1),
            ("feedback", "value", 0.0, -1), ("envelope", "stream", 1.0, -1), ("level", "value", 1.0, -1)],
    outputs=["output"],
    state=[("phase", "f32", 0.0, -1), ("prev_output", "f32", 0.0, -1)],
    process="""
    const float feedback_mod = prev_output * feedback;
    const float total_phase_mod = phase_mod + feedback_mod;
    // `((phase + mod) * TAU).sin()`: the device library's sine of an argument in TURNS (og_math.h, og_sin_turns) -- what
    // the built-in operator calls; og_sinf((phase + mod) * 6.2831855f) is the radian form, 12 instructions more
    output = og_sin_turns(phase + total_phase_mod) * envelope * level;
    prev_output = output;
    const float p = phase + (base_freq * ratio) / sample_rate;
    phase = p - truncf(p); // fract()
""")

# crossfade.rs:37-44
CROSSFADE = dict(
    inputs=[("input", "stream", 0.0, -1), ("mix", "value", 0.0, -1)],
    outputs=["output_a", "output_b"],
    process="""
    const float m = og::clamp01(mix);
    output_a = input * (1.0f - m);
    output_b = input * m;
""")

MIXER = dict(inputs=[("input_a", "stream", 0.0, -1), ("input_b", "stream", 0.0, -1)], outputs=["output"],
             process="    output = input_a + input_b;\n")                      # mixer.rs:30-34
ADD_VALUE = dict(inputs=[("input", "stream", 0.0, -1), ("value_in", "value", 0.0, 0)], outputs=["output"], n_ctor_args=1,
                 process="    output = input + value_in;\n")                     # add_value.rs:31-35 (`value` is reserved here)
HARD_CLIP = dict(inputs=[("input", "stream", 0.0, -1)], outputs=["output"],
                 process="    output = og::clampf(input * 1.5f, -0.7f, 0.7f);\n")  # oversampled-saturator main.rs:54-61
VCA = dict(inputs=[("input", "stream", 0.0, -1), ("control", "stream", 1.0, -1)], outputs=["output"],
           process="    output = input * control;\n")                           # pivot vca.rs:31-35

ALL = {"UFmOperator::new": FM_OPERATOR, "UCrossfade::new": CROSSFADE, "UMixer::new": MIXER, "UAddValue::new": ADD_VALUE,
       "UHardClip::new": HARD_CLIP, "UVca::new": VCA}


def register_all():
    for name, d in ALL.items():
        oscen_amd.register_node(name, d["inputs"], d["outputs"], d["process"], state=d.get("state", ()),
                                handlers=d.get("handlers"), n_ctor_args=d.get("n_ctor_args", 0))


def user_fm_voice():
    """the built-in fm_voice description with every example-crate node swapped for its plug-in twin"""
    register_all()
    dsl = oscen_amd.Graph(builtin="fm_voice").to_dsl()
    for t in ("FmOperator", "Crossfade", "Mixer", "AddValue"):
        dsl = dsl.replace(t + "::new", "U" + t + "::new")
    dsl = dsl.replace("cutoff_mod.value", "cutoff_mod.value_in")
    return oscen_amd.Graph(dsl=dsl, per_voice=("frequency",))
