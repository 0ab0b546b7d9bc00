// og_builtin.cpp -- builder-form descriptions of the reference graphs in scope.
// These are the graphs whose kernels are compiled ahead of time into the
// library; any other description goes through the same compiler at og_create().
#include <map>
#include <stdexcept>

#include "og_graph.h"

namespace ogc {
namespace {

GInput value_in(const char* name, float def, uint32_t ramp = 0)
{
    GInput i;
    i.name = name;
    i.kind = Kind::Value;
    i.def = def;
    i.ramp_frames = ramp;
    return i;
}
GInput voice_in(const char* name, float def)
{
    GInput i = value_in(name, def);
    i.per_voice = true;
    return i;
}
GInput event_in(const char* name)
{
    GInput i;
    i.name = name;
    i.kind = Kind::Event;
    return i;
}

// FMVoice (examples/fm-synth/src/fm_voice.rs:6-156) as driven by the FMGraph
// wrapper (examples/fm-synth/src/lib.rs:22-131): `frequency`/`gate` arrive per
// voice, every other input is broadcast, and the wrapper's [ramp: 2205]
// inputs are ramped.
GraphDesc fm_voice()
{
    GraphDesc g;
    g.name = "fm_voice";
    const uint32_t R = 2205;
    g.inputs.push_back(voice_in("frequency", 440.0f));
    g.inputs.push_back(event_in("gate"));
    struct Op {
        const char* n;
        float ratio, level, a, d, s, r;
        bool has_level;
    } ops[3] = {{"op3", 3.0f, 0.5f, 0.01f, 0.1f, 0.7f, 0.3f, true},
                {"op2", 2.0f, 0.5f, 0.01f, 0.1f, 0.7f, 0.3f, true},
                {"op1", 1.0f, 0.0f, 0.01f, 0.2f, 0.8f, 0.5f, false}};
    for (const Op& o : ops) {
        std::string p = o.n;
        g.inputs.push_back(value_in((p + "_ratio").c_str(), o.ratio));
        if (o.has_level) {
            g.inputs.push_back(value_in((p + "_level").c_str(), o.level, R));
            g.inputs.push_back(value_in((p + "_feedback").c_str(), 0.0f, R));
        }
        g.inputs.push_back(value_in((p + "_attack").c_str(), o.a));
        g.inputs.push_back(value_in((p + "_decay").c_str(), o.d));
        g.inputs.push_back(value_in((p + "_sustain").c_str(), o.s));
        g.inputs.push_back(value_in((p + "_release").c_str(), o.r));
    }
    g.inputs.push_back(value_in("route", 0.0f, R));
    g.inputs.push_back(value_in("filter_cutoff", 2000.0f, R));
    g.inputs.push_back(value_in("filter_resonance", 0.707f, R));
    g.inputs.push_back(value_in("filter_attack", 0.01f));
    g.inputs.push_back(value_in("filter_decay", 0.2f));
    g.inputs.push_back(value_in("filter_sustain", 0.5f));
    g.inputs.push_back(value_in("filter_release", 0.3f));
    g.inputs.push_back(value_in("filter_env_amount", 0.0f, R));
    g.outputs.push_back({"audio_out", Kind::Stream});

    g.nodes.push_back({"env3", "AdsrEnvelope::new", {0.01f, 0.1f, 0.7f, 0.3f}, 1});
    g.nodes.push_back({"env2", "AdsrEnvelope::new", {0.01f, 0.1f, 0.7f, 0.3f}, 1});
    g.nodes.push_back({"env1", "AdsrEnvelope::new", {0.01f, 0.2f, 0.8f, 0.5f}, 1});
    g.nodes.push_back({"env_filter", "AdsrEnvelope::new", {0.01f, 0.2f, 0.5f, 0.3f}, 1});
    g.nodes.push_back({"filter_env_gain", "Gain::new", {0.0f}, 1});
    g.nodes.push_back({"cutoff_mod", "AddValue::new", {2000.0f}, 1});
    g.nodes.push_back({"op3_osc", "FmOperator::new", {}, 1});
    g.nodes.push_back({"op2_osc", "FmOperator::new", {}, 1});
    g.nodes.push_back({"op1_osc", "FmOperator::new", {}, 1});
    g.nodes.push_back({"op3_route", "Crossfade::new", {}, 1});
    g.nodes.push_back({"op1_mod_mixer", "Mixer::new", {}, 1});
    g.nodes.push_back({"filter", "TptFilter::new", {2000.0f, 0.707f}, 1});
    g.nodes.push_back({"output_gain", "Gain::new", {0.3f}, 1});

    auto c = [&](const std::string& s, const std::string& d) { g.edges.push_back({s, d, ""}); };
    const char* envs[4][2] = {{"env3", "op3"}, {"env2", "op2"}, {"env1", "op1"}, {"env_filter", "filter"}};
    for (auto& e : envs) c("gate", std::string(e[0]) + ".gate");
    for (auto& e : envs)
        for (const char* prm : {"attack", "decay", "sustain", "release"})
            c(std::string(e[1]) + "_" + prm, std::string(e[0]) + "." + prm);
    c("env_filter.output", "filter_env_gain.input");
    c("filter_env_amount", "filter_env_gain.gain");
    c("filter_env_gain.output", "cutoff_mod.input");
    c("filter_cutoff", "cutoff_mod.value");
    c("cutoff_mod.output", "filter.cutoff");
    for (const char* op : {"op3", "op2", "op1"}) {
        std::string o = op;
        c("frequency", o + "_osc.base_freq");
        c(o + "_ratio", o + "_osc.ratio");
        if (o != "op1") {
            c(o + "_feedback", o + "_osc.feedback");
            c(o + "_level", o + "_osc.level");
        }
        c("env" + o.substr(2) + ".output", o + "_osc.envelope");
    }
    c("op3_osc.output", "op3_route.input");
    c("route", "op3_route.mix");
    c("op3_route.output_a", "op2_osc.phase_mod");
    c("op2_osc.output", "op1_mod_mixer.input_a");
    c("op3_route.output_b", "op1_mod_mixer.input_b");
    c("op1_mod_mixer.output", "op1_osc.phase_mod");
    c("op1_osc.output", "filter.input");
    c("filter_resonance", "filter.q");
    c("filter.output", "output_gain.input");
    c("output_gain.output", "audio_out");
    return g;
}

// "osc+env+TptFilter" voice (oscen-lib/perf/profile_graph.rs:12-37)
GraphDesc sub_voice()
{
    GraphDesc g;
    g.name = "sub_voice";
    g.inputs.push_back(voice_in("frequency", 440.0f));
    g.inputs.push_back(event_in("gate"));
    g.inputs.push_back(value_in("cutoff", 3000.0f));
    g.inputs.push_back(value_in("q", 0.707f));
    g.outputs.push_back({"audio", Kind::Stream});
    g.nodes.push_back({"osc", "PolyBlepOscillator::saw", {440.0f, 0.6f}, 1});
    g.nodes.push_back({"filter", "TptFilter::new", {3000.0f, 0.707f}, 1});
    g.nodes.push_back({"envelope", "AdsrEnvelope::new", {0.01f, 0.1f, 0.7f, 0.2f}, 1});
    g.edges.push_back({"frequency", "osc.frequency", ""});
    g.edges.push_back({"gate", "envelope.gate", ""});
    g.edges.push_back({"cutoff", "filter.cutoff", ""});
    g.edges.push_back({"q", "filter.q", ""});
    g.edges.push_back({"osc.output", "filter.input", ""});
    g.edges.push_back({"filter.output * envelope.output", "audio", ""});
    return g;
}

// Echo voice: the sub-style voice into the feedback echo of examples/simple-echo/src/lib.rs:35-64
// (Delay::new(11025, 0) -> TptFilter::new(4000, 0.7), filter output fed back into the delay input,
// dry/wet mix), written as a graph with a `-> [delay] ->` feedback edge.  The delay line of every
// voice lives in HBM: this is the configuration whose traffic is dominated by HBM rather than state.
GraphDesc echo_voice()
{
    GraphDesc g;
    g.name = "echo_voice";
    g.inputs.push_back(voice_in("frequency", 440.0f));
    g.inputs.push_back(event_in("gate"));
    g.inputs.push_back(value_in("delay_samples", 11025.0f));
    g.inputs.push_back(value_in("feedback", 0.4f));
    g.inputs.push_back(value_in("cutoff", 4000.0f));
    g.inputs.push_back(value_in("dry", 0.65f));
    g.inputs.push_back(value_in("wet", 0.35f));
    g.outputs.push_back({"audio", Kind::Stream});
    g.nodes.push_back({"osc", "PolyBlepOscillator::saw", {440.0f, 0.6f}, 1});
    g.nodes.push_back({"envelope", "AdsrEnvelope::new", {0.01f, 0.1f, 0.7f, 0.2f}, 1});
    g.nodes.push_back({"sum", "Mixer::new", {}, 1});
    g.nodes.push_back({"echo", "Delay::new", {11025.0f, 0.0f}, 1});
    g.nodes.push_back({"filter", "TptFilter::new", {4000.0f, 0.7f}, 1});
    g.edges.push_back({"frequency", "osc.frequency", ""});
    g.edges.push_back({"gate", "envelope.gate", ""});
    g.edges.push_back({"delay_samples", "echo.delay_samples", ""});
    g.edges.push_back({"cutoff", "filter.cutoff", ""});
    g.edges.push_back({"osc.output * envelope.output", "sum.input_a", ""});
    g.edges.push_back({"filter.output * feedback", "sum.input_b", ""});
    g.edges.push_back({"sum.output", "echo.input", ""});
    GEdge fb{"echo.output", "filter.input", ""};
    fb.feedback = true;
    g.edges.push_back(fb);
    g.edges.push_back({"osc.output * envelope.output * dry + filter.output * wet", "audio", ""});
    return g;
}

// SatGraph_{1,4}x (examples/oversampled-saturator/src/main.rs:64-80): saw -> HardClip, both `* N`,
// `[sinc] clip.output -> audio_out`.  The oscillator frequency is a per-voice input here (the
// reference fixes 2000 Hz) so that a bank of voices is not N copies of one signal.
GraphDesc sat_voice(uint32_t factor)
{
    GraphDesc g;
    g.name = factor > 1 ? "sat4x_voice" : "sat1x_voice";
    g.inputs.push_back(voice_in("frequency", 2000.0f));
    g.outputs.push_back({"audio_out", Kind::Stream});
    g.nodes.push_back({"osc", "PolyBlepOscillator::saw", {2000.0f, 0.6f}, factor});
    g.nodes.push_back({"clip", "HardClip::new", {}, factor});
    g.edges.push_back({"frequency", "osc.frequency", ""});
    g.edges.push_back({"osc.output", "clip.input", ""});
    g.edges.push_back({"clip.output", "audio_out", factor > 1 ? "sinc" : ""});
    return g;
}

// ElectricPianoVoiceNode (examples/electric-piano/src/electric_piano_voice.rs:362-402) inside the
// ElectricPianoGraph wrapper (examples/electric-piano/src/main.rs:33-97): per-voice frequency/gate,
// six broadcast voice parameters, `voices.output -> tremolo.input` (sum), Tremolo -> Frame<2> out.
GraphDesc epiano_voice()
{
    GraphDesc g;
    g.name = "epiano_voice";
    g.inputs.push_back(voice_in("frequency", 440.0f));
    g.inputs.push_back(event_in("gate"));
    const char* prm[6] = {"brightness", "velocity_scaling", "decay_rate", "harmonic_decay", "key_scaling", "release_rate"};
    const float def[6] = {30.0f, 50.0f, 90.0f, 70.0f, 50.0f, 40.0f};
    for (int i = 0; i < 6; ++i) g.inputs.push_back(value_in(prm[i], def[i]));
    g.inputs.push_back(value_in("vibrato_intensity", 0.3f));
    g.inputs.push_back(value_in("vibrato_speed", 5.0f));
    g.outputs.push_back({"output", Kind::Stream});
    g.outputs.push_back({"out", Kind::Stream});
    g.nodes.push_back({"amplitude_source", "AmplitudeSource::new", {}, 1});
    g.nodes.push_back({"oscillator_bank", "OscillatorBank::new", {}, 1});
    GNode trem{"tremolo", "Tremolo::new", {}, 1};
    trem.bus = true;
    g.nodes.push_back(trem);
    auto c = [&](const std::string& s, const std::string& d) { g.edges.push_back({s, d, ""}); };
    c("frequency", "amplitude_source.frequency");
    c("gate", "amplitude_source.gate");
    for (int i = 0; i < 6; ++i) c(prm[i], std::string("amplitude_source.") + prm[i]);
    c("frequency", "oscillator_bank.frequency");
    c("gate", "oscillator_bank.gate");
    c("amplitude_source.amplitudes", "oscillator_bank.amplitudes");
    c("oscillator_bank.output", "output");
    c("output", "tremolo.input");
    c("vibrato_intensity", "tremolo.depth");
    c("vibrato_speed", "tremolo.rate");
    c("tremolo.output", "out");
    return g;
}

} // namespace

// The voice graphs by themselves, as the example crates declare them (no wrapper: `frequency` is an ordinary value
// input, nothing is ramped, no post-mix stage) -- what `voices = [FMVoice::new(); 8]` names.
std::map<std::string, GraphDesc> builtin_voice_graph_types()
{
    std::map<std::string, GraphDesc> R;
    {
        GraphDesc v = fm_voice();
        v.name = "FMVoice";
        for (GInput& in : v.inputs) {
            in.per_voice = false;
            in.ramp_frames = 0;
        }
        R["FMVoice"] = v;
    }
    {
        GraphDesc w = epiano_voice(), v;
        v.name = "ElectricPianoVoiceNode";
        for (GInput in : w.inputs) {
            if (in.name == "vibrato_intensity" || in.name == "vibrato_speed") continue; // (the wrapper's Tremolo parameters)
            in.per_voice = false;
            v.inputs.push_back(in);
        }
        v.outputs.push_back({"output", Kind::Stream});
        for (const GNode& n : w.nodes)
            if (!n.bus) v.nodes.push_back(n);
        for (const GEdge& e : w.edges)
            if (e.src.find("tremolo") == std::string::npos && e.dst.find("tremolo") == std::string::npos && e.src != "output" &&
                e.src != "vibrato_intensity" && e.src != "vibrato_speed")
                v.edges.push_back(e);
        R["ElectricPianoVoiceNode"] = v;
    }
    return R;
}

std::vector<std::string> builtin_graph_names() { return {"fm_voice", "sub_voice", "sat4x_voice", "sat1x_voice", "epiano_voice", "echo_voice"}; }

GraphDesc builtin_graph(const std::string& name)
{
    if (name == "fm_voice") return fm_voice();
    if (name == "sub_voice") return sub_voice();
    if (name == "sat4x_voice") return sat_voice(4);
    if (name == "sat1x_voice") return sat_voice(1);
    if (name == "epiano_voice") return epiano_voice();
    if (name == "echo_voice") return echo_voice();
    throw std::runtime_error("unknown builtin graph '" + name + "'");
}

} // namespace ogc
