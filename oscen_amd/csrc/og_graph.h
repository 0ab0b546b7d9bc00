// og_graph.h -- host-side graph description, lowering and HIP code generation.
//
// This is the MI355X engine's counterpart of the reference's compile-time
// layer (oscen-macros `graph!` -> oscen-graph-compiler: parse -> IR -> passes
// -> codegen).  The reference decides the schedule when rustc expands the
// macro; here the same decisions are taken when a graph description is
// flattened on the host:
//   * Kahn topological order over non-feedback edges   (ir/lower.rs:1015-1085)
//   * dead-node removal by reverse BFS from the outputs (ir/passes/dead_nodes.rs:11-62)
//   * >= 2 stream sources into one input = sum in edge order
//                                                       (codegen/emit_node.rs:35-111,153-174)
//   * compound sources (`a.x * b.y -> dst`) evaluated as f32 expressions
//                                                       (codegen/emit_node.rs:241-286)
//   * event inputs dispatched to the node handlers on their exact frame
//                                                       (codegen/mod.rs:755-873)
// and the result is one fused voice kernel (HIP source) whose per-voice state
// is laid out as [word][voice] planes.
#pragma once
#include "og_abi.h"
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace ogc {

enum class Kind { Value = 0, Event = 1, Stream = 2 };

// How fast a value changes, which decides where it is computed.
enum class Rate {
    Const = 0,   // literal
    UBlock = 1,  // voice-uniform, constant over the block (plain broadcast value input)
    UFrame = 2,  // voice-uniform, per-frame (ramped broadcast value input)
    VBlock = 3,  // per-voice, changes only through set-value events
    Vary = 4,    // per-voice per-sample
};

struct UEnv { // host-side evaluation environment for block-uniform expressions
    float sample_rate;
    const float* input_values; // by graph-input index (ramped inputs: `.current`)
    // the blocks one launch renders: first frame of block k inside the launch (block 0 starts at 0), for handlers that
    // read EventInstance::frame_offset (the offset inside the process_block call)
    const uint32_t* block_starts = nullptr;
    uint32_t n_blocks = 0;
};
using HostFn = std::function<float(const UEnv&)>;

struct GInput {
    std::string name;
    Kind kind = Kind::Value;
    float def = 0.0f;
    uint32_t ramp_frames = 0; // [ramp: N] (value inputs only)
    bool per_voice = false;   // value input fed per voice (MidiVoiceHandler.frequency)
    int channels = 1;         // stream inputs: `input stream dry: Frame<2>;` (oscen-lib/tests/stereo_render.rs:46-47) = N rows
};
struct GOutput {
    std::string name;
    Kind kind = Kind::Stream;
    int channels = 0; // `output out: stream: Frame<2>;`: the declared frame width (0 = not declared: taken from what feeds it)
};
struct GNode {
    std::string name;
    std::string type; // "AdsrEnvelope::new", "PolyBlepOscillator::saw", ...
    std::vector<float> args;
    uint32_t rate_factor = 1; // `* N`
    bool bus = false;         // post-mix node of the wrapper graph (runs once on the summed bus, e.g. Tremolo)
    uint32_t array_len = 0;   // `name = [Type::ctor(..); N]` (parse.rs:447-520): a node array of N elements; 0 = a single node
    // constructor arguments that are not numbers (`Voice::new(sample_rate)`, `Convolver::with_ir(reverb_ir())`): raw text,
    // parallel to `args` ("" where the argument is a number; empty vector = all numeric).  A node type that needs the
    // value of such an argument is diagnosed at lowering; nested graph types ignore their constructor arguments.
    std::vector<std::string> raw_args;
    bool inline_bare = false; // nested graph inlined WITHOUT the `<name>_` prefix (the voice graph of a lowered poly wrapper)
};
struct GEdge {
    std::string src; // endpoint or compound expression: "env.output", "a.x * b.y", "gate"
    std::string dst; // "node.port" or graph output name
    std::string policy; // "", "sinc", "sinc_iir", "linear", "latch"
    // outgoing leg of `src -> [via] -> dst` (ir/lower.rs:342-347): imposes no ordering; a consumer
    // scheduled before `via` reads the output `via` produced on the previous frame
    bool feedback = false;
};

struct GraphDesc {
    std::string name;
    std::vector<GInput> inputs;
    std::vector<GOutput> outputs;
    std::vector<GNode> nodes;
    std::vector<GEdge> edges;
};

// What lowering a poly WRAPPER graph (MidiParser -> VoiceAllocator<N> -> [MidiVoiceHandler; N] -> [Voice; N] -> sum
// [-> post-mix node], examples/fm-synth/src/lib.rs:22-131) found: the control-rate MIDI nodes run on the host
// (og_midi_*), the voice array is the bank.  `declared_voices` is the reference's N (the bank size is og_create's).
struct PolyInfo {
    bool is_wrapper = false;
    uint32_t declared_voices = 0;
    std::string voice_type;      // "FMVoice"
    std::string frequency_input; // per-voice value input of the lowered graph fed by MidiVoiceHandler.frequency
    std::string gate_input;      // event input fed by MidiVoiceHandler.gate
    std::string midi_input;      // the wrapper's raw-MIDI event input (`midi_in`), served by og_midi_send
    std::vector<std::string> dropped_event_outputs; // `midi_parser.note_on -> note_on_out`: host-side events, not lowered
};
// The voice-bank graph a poly wrapper lowers to (identity for any other description).
GraphDesc lower_poly_wrapper(const GraphDesc& g, PolyInfo* info = nullptr);
// the reference's example voice graphs as graph TYPES (usable as nested nodes and as the voice of a poly wrapper):
// "FMVoice" (examples/fm-synth/src/fm_voice.rs:6-156), "ElectricPianoVoiceNode" (electric_piano_voice.rs:362-402)
std::map<std::string, GraphDesc> builtin_voice_graph_types();

// ---- compiled form ---------------------------------------------------------
struct InputInfo {
    GInput decl;
    int slot = -1;       // broadcast value inputs: block-uniform slot with the current value
    int ramp_row = -1;   // ramped inputs: row in the per-frame ramp table
    int state_word = -1; // per-voice value inputs: state plane
    int event_index = -1;
    int stream_row = -1; // stream inputs: row of the per-frame table (after the ramp rows)
};
struct StateWord {
    std::string name;
    bool is_float = true;
    std::function<uint32_t(const UEnv&)> init; // initial bits
    bool read_mostly = false; // written back only in blocks that changed it (tables derived from events)
};
struct UniformProg {
    int dst;
    std::function<uint32_t(const UEnv&)> fn;
};
struct RingSpec { // a per-voice delay line in HBM (Delay's RingBuffer, delay/mod.rs:59-69)
    std::string name;
    float rate_factor = 1.0f;
    // next_power_of_two(min((2.0 * sr) as usize, 88200))
    uint32_t capacity(float graph_sr) const;
};
struct CompiledGraph {
    std::string name;
    std::string source; // complete HIP translation unit for this graph
    uint64_t hash = 0;  // FNV-1a of the kernel body: AOT registry / JIT cache key
    std::vector<InputInfo> inputs;
    std::vector<StateWord> state;      // per-voice words
    std::vector<StateWord> lane_state; // per-(voice, lane) words of LPV > 1 graphs
    std::vector<RingSpec> rings;       // delay lines (at most OG_MAX_RINGS)
    int lpv = 1;                       // lanes per voice (8 for the electric-piano voice)
    int lane_width = 1;                // words a lane owns of every lane_state array (OG_HPL = 4 when lpv > 1)
    bool can_split = false;            // a two-wave pipeline variant of the kernel exists (og_k2_*)
    int max_pipeline = 1;              // deepest pipeline variant generated: 1, 2 (og_k2_*) or 4 (og_k4_*)
    bool wide4 = false;                // the four-wave pipeline also exists with 16-frame hand-offs (og_k4w_*): fewer barriers, twice
                                       // the LDS rings -- for banks whose workgroups all fit a CU at once
    int valu_estimate = 0;             // estimated VALU instructions per frame of one wave of the ordinary kernel (node weights)
    // post-mix stage (electric-piano/src/main.rs:88-96): Tremolo on the summed bus -> Frame<2>
    bool bus_tremolo = false;
    HostFn tremolo_rate, tremolo_depth;
    std::vector<UniformProg> uprogs;
    int n_slots = 0;
    int n_ramps = 0;
    int n_streams = 0; // rows n_ramps.. of the per-frame table taken by graph-level stream inputs (`<stream_in>_block`;
                       // a Frame<N> input takes N consecutive rows, one per channel)
    int n_stream_inputs = 0; // the stream inputs themselves (BlockRender::NUM_STREAM_INPUTS)
    int n_event_inputs = 0;
    std::vector<std::string> event_outputs; // the graph's event outputs that a node feeds, in declaration order (og_read_output_events)
    bool has_node_event_outputs = false;    // some live node has an #[output(event)] field (drops are counted on the device)
    uint32_t channels = 1;
    uint32_t voice_channels = 1; // channels of the voice sum: one per stream output of the voice graph, a Frame<2> output two
    struct OutChan {
        std::string name;
        int offset, width;
    };
    std::vector<OutChan> output_channels; // where every stream output sits in a frame of the bus / of a tap
    uint32_t latency_samples = 0;
    std::vector<std::string> node_order;     // the reference's topological order (Kahn, ir/lower.rs:1015-1085)
    std::vector<std::string> schedule_order; // the order the nodes are emitted in (== node_order unless the graph is a pure
                                             // per-frame dataflow, which is scheduled depth first from its sinks)
    int find_input(const std::string& n) const;
};

// Throws std::runtime_error with a diagnostic on malformed/unsupported graphs.
std::unique_ptr<CompiledGraph> compile(const GraphDesc& g);
// The connections the reference's macro refuses (kind mismatch between typed ends, ir/lower.rs:459-490; a summed fan-in with a
// compound / cross-rate / array source, codegen/emit_node.rs:35-125), checked on the description as written.  compile() and
// register_graph_type() call it; throws with the reference's wording.
void check_reference_rules(const GraphDesc& g);

// ---- user node types: the `#[derive(Node)]` plug-in surface (oscen-macros/src/lib.rs:7-327) -----------------
// A node type = its endpoints (`#[input(stream|value|event)]`, `#[output(stream)]` fields), its private fields
// (per-voice state words), the body of `SignalProcessor::process()` and of the `on_<event>()` handlers, given as
// device source.  Inside the bodies every input is a `const float <name>`, every state field a `float& <name>` /
// `uint32_t& <name>`, every output a `float& <name>`, plus `const float sample_rate`; an event handler also sees
// `const float value` (the scalar payload).  The bodies are compiled into the fused voice kernel (hiprtc).
struct UserPort {
    std::string name;
    Kind kind = Kind::Stream;
    float def = 0.0f;
    int arg = -1; // constructor argument that initialises the field, or -1
    int channels = 1; // stream inputs: 1 = f32, N = Frame<N>
};
struct UserState {
    std::string name;
    bool is_uint = false;
    float init_f = 0.0f;
    uint32_t init_u = 0;
    int arg = -1; // constructor argument that initialises it (f32 fields), or -1
};
struct UserNodeType {
    std::string type; // "FmOperator::new"
    size_t nargs = 0;
    std::vector<UserPort> inputs;
    std::vector<std::string> outputs;
    std::vector<int> out_channels;       // per output: 1 = f32, N = Frame<N> (empty: all f32)
    std::vector<std::string> ev_outputs; // `#[output(event)]` fields
    std::vector<UserState> state;
    std::string process_src;
    std::map<std::string, std::string> handlers; // event input -> body of on_<input>()
    int weight = 0; // estimated VALU cost per tick (0 = estimate from the source)
    int event_capacity = 0; // pushes per frame an event output of this type holds (0 = 2; the reference: 32)
};
void register_user_node(const UserNodeType& t); // throws on malformed descriptions or a clash with a built-in type
bool unregister_user_node(const std::string& type);

// named pure functions applied on a connection (`half(a.output) -> out`, `dsp::decode_ms(s.output) -> out`;
// ast.rs:126-128, oscen-lib/tests/connection_expr_functions.rs): device source, like a user node's process()
struct UserFunction {
    std::string name;                   // "half", or a path "dsp::decode_ms"
    std::vector<std::string> arg_names; // parameter names inside `source`
    std::vector<int> arg_channels;      // 1 = f32, N = Frame<N> (og::Frame<N>)
    int result_channels = 1;
    std::string source;                 // the function body: `return x * 0.5f;`
};
void register_user_function(const UserFunction& f); // throws on a malformed description
bool unregister_user_function(const std::string& name);

// graph types usable as nodes of other graphs (nested graphs: `sub = SubGraph::new()`), expanded inline
void register_graph_type(const std::string& name, const GraphDesc& g);
bool unregister_graph_type(const std::string& name);
// node arrays and nested graphs are desugared before lowering; exposed for tests / to_dsl of the expansion
GraphDesc expand(const GraphDesc& g);
// `TptFilter::<Frame<2>>::new` / `::<Stereo>` -> "TptFilter<2>::new"; `::<f32>` -> "TptFilter::new"
std::string normalize_type(const std::string& type);

uint64_t fnv1a(const std::string& s);

// `graph! { ... }` body text -> description (og_dsl.cpp).  `per_voice` names the value inputs the
// poly wrapper feeds per voice (e.g. "frequency").  to_dsl() prints a description back as DSL text.
GraphDesc parse_dsl(const std::string& text, const std::vector<std::string>& per_voice);
std::string to_dsl(const GraphDesc& g);

// Built-in graph descriptions (builder form of the reference graphs in scope).
GraphDesc builtin_graph(const std::string& name);
std::vector<std::string> builtin_graph_names();

} // namespace ogc
