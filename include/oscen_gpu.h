/*
 * oscen_gpu.h -- C ABI of the MI355X voice-parallel synthesis engine.
 *
 * The reference (reedrosenbluth/oscen) has no FFI boundary on this path: node
 * bodies are inlined into the struct that the `graph!` proc-macro generates.
 * The boundary this library sits behind is therefore the PUBLIC SURFACE OF A
 * GENERATED GRAPH STRUCT (oscen-graph-compiler/src/codegen/mod.rs:1292-1392);
 * every entry point below cites the generated item it replaces.  A Rust shim
 * implementing the same-named inherent methods over this ABI is shown in
 * INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative OG_E_* code and never throws across the ABI;
 * og_last_error() gives the message for the calling thread.  An engine is not
 * thread-safe (the reference's graph is `&mut self`, one audio thread).
 * There is no CPU fallback: creating an engine without a usable HIP device
 * fails with OG_E_DEVICE.
 */
#ifndef OSCEN_GPU_H
#define OSCEN_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_OK 0
#define OG_E_INVALID (-1)     /* bad argument / unknown name            */
#define OG_E_UNSUPPORTED (-2) /* graph uses a feature this build lacks  */
#define OG_E_DEVICE (-3)      /* HIP error or no device                 */
#define OG_E_STATE (-4)       /* call order (e.g. process before init)  */
#define OG_E_OVERFLOW (-5)    /* event queue capacity (event dropped)   */
#define OG_E_NOMEM (-6)       /* host allocation failed                 */

#define OG_MAX_BLOCK_SIZE 512u /* MAX_BLOCK_SIZE, oscen-lib/src/graph/types.rs:12 */
#define OG_MAX_EVENTS_PER_BLOCK 32u /* per voice per endpoint per block, types.rs:18 */

/* endpoint kinds of `input name: kind` (oscen-graph-compiler/src/parse.rs:387-443) */
#define OG_KIND_VALUE 0
#define OG_KIND_EVENT 1
#define OG_KIND_STREAM 2

/* input flags */
#define OG_IN_PER_VOICE 1u /* value fed per voice (MidiVoiceHandler.frequency, oscen-lib/src/midi.rs:49) */
#define OG_IN_CHANNELS(n) ((uint32_t)(n) << 8) /* stream inputs: `input stream dry: Frame<n>;` (n = 2..4;
                                                * oscen-lib/tests/stereo_render.rs:46-47) */

typedef struct og_graph_desc og_graph_desc; /* a `graph! { ... }` body, builder form */
typedef struct og_engine og_engine;         /* N voices of one voice graph + the mix bus */

/* ---- graph description: the `graph!` DSL surface ----------------------------
 * name: X;                                  -> og_graph_new("X")
 * input n: value = d [ramp: N];             -> og_graph_add_input(g,"n",OG_KIND_VALUE,d,N,flags)
 * input n: event;                           -> og_graph_add_input(g,"n",OG_KIND_EVENT,0,0,0)
 * output n: stream;                         -> og_graph_add_output(g,"n",OG_KIND_STREAM)
 * nodes { n = Type::ctor(a, b) [* N]; }     -> og_graph_add_node(g,"n","Type::ctor",args,nargs,N)
 * connections { [policy] src_expr -> dst; } -> og_graph_connect(g,"src_expr","dst","policy")
 * connections { src -> [via] -> dst; }      -> og_graph_connect_via(g,"src","via","dst")
 * (oscen-graph-compiler/src/parse.rs:195-979)                                            */
int og_graph_new(const char* name, og_graph_desc** out);
int og_graph_builtin(const char* name, og_graph_desc** out); /* "fm_voice", "sub_voice", ... */
int og_graph_add_input(og_graph_desc* g, const char* name, int kind, float default_value,
                       uint32_t ramp_frames, uint32_t flags);
int og_graph_add_output(og_graph_desc* g, const char* name, int kind);
int og_graph_add_node(og_graph_desc* g, const char* name, const char* type_ctor, const float* args,
                      uint32_t n_args, uint32_t rate_factor);
/* `name = [Type::ctor(..); length] [* N]` (parse.rs:447-520): an array of nodes inside the voice graph.  Edges follow
 * the reference's fan-out rules (ir/lower.rs:784-786, codegen/emit_edge.rs:30-84): array -> array of the same length
 * pairs the elements, scalar -> array broadcasts, `array.port -> scalar` is the sum in index order; `name[i].port`
 * addresses one element. */
int og_graph_add_node_array(og_graph_desc* g, const char* name, const char* type_ctor, const float* args,
                            uint32_t n_args, uint32_t rate_factor, uint32_t length);

/* ---- custom node types: the `#[derive(Node)]` plug-in surface (oscen-macros/src/lib.rs:7-327) ------------------
 * What the reference's FmOperator / Crossfade / Vca are to oscen-lib: a struct with `#[input(stream|value|event)]`
 * and `#[output(stream)]` fields, private fields, `impl SignalProcessor { fn process(&mut self) }` and
 * `fn on_<event_input>(&mut self, &EventInstance)` handlers.  Here the two bodies are DEVICE SOURCE (C++): inside
 * them every stream/value input is a `const float <name>`, every private field a `float& <name>` (or `uint32_t&`),
 * every output a `float& <name>` to assign, plus `const float sample_rate`; an event handler also sees
 * `const float value` (the scalar payload) -- and, if its source names it, `const uint32_t frame_offset`
 * (EventInstance::frame_offset, oscen-lib/src/graph/types.rs:129-132: for an event of a GRAPH input the offset inside the
 * process_block call, times N for a `* N` node; for an event pushed by another NODE the offset its producer gave it --
 * `<event_output>.push_at(frame_offset, value)`, plain `push(value)` = offset 0 -- multiplied (saturating) by N on an
 * outer -> inner edge and divided by N on an inner -> outer edge, codegen/emit_edge.rs:86-99) -- but only the VALUE inputs
 * (streams do not exist yet when an event fires).  og_math.h / og_nodes.hip.h helpers (og_sin_turns, og_sinf, og::clampf,
 * ...) are in scope; og_sin_turns -- sin of an argument in TURNS -- is the hardware sine, documented for |t| <= 256 turns
 * (gfx950 reduces larger arguments itself, measured; og_sin_turns_wide takes the fractional part first and does not rely on
 * that).  The bodies are compiled into
 * the fused voice kernel by hiprtc when an engine is created for a graph that uses the type.  Process-wide registry. */
typedef struct {
    const char* name;
    int kind;            /* OG_KIND_STREAM / OG_KIND_VALUE / OG_KIND_EVENT */
    float default_value; /* field value while the input is unconnected */
    int ctor_arg;        /* index of the constructor argument that sets that value, or -1 */
    uint32_t channels;   /* stream inputs: 0 or 1 = f32, N (2..4) = Frame<N> (oscen-lib/src/frame.rs): the source sees a
                          * `const og::Frame<N>` with `.v[i]`, `+`, `-`, `* float` and unary minus */
} og_node_port;
typedef struct {
    const char* name;
    int is_uint;        /* 0: f32 field, 1: u32 field */
    float init;         /* initial value of an f32 field ... */
    uint32_t init_uint; /* ... of a u32 field */
    int ctor_arg;       /* f32 fields: constructor argument that initialises it, or -1 */
} og_node_field;
typedef struct {
    const char* type_ctor; /* "FmOperator::new" */
    uint32_t n_ctor_args;
    const og_node_port* inputs;
    uint32_t n_inputs;
    const char* const* outputs;
    uint32_t n_outputs;
    const og_node_field* state;
    uint32_t n_state;
    const char* process_src;               /* body of process() */
    const char* const* event_handler_src;  /* n_inputs entries (or NULL): body of on_<input>() for event inputs */
    uint32_t cost_hint;                    /* estimated VALU instructions per tick; 0 = estimate from the source */
    /* `#[output(event)]` fields (oscen-macros/src/lib.rs:127-133).  process() and the handlers see each as an
     * object with `push(float scalar)` (= EventOutput::try_push of a scalar payload at the current frame; at most
     * event_queue_capacity -- default OG_NODE_EVENTS_PER_FRAME = 2, at most 32 like the reference's queue -- per frame
     * and output, further pushes are dropped like try_push on a full queue and counted in og_events_dropped).  `a.trig -> b.gate` runs b's on_gate for every event a pushed on that frame,
     * before b.process(); the outputs are cleared once per frame (clear_event_outputs, lib.rs:237-256). */
    const char* const* event_outputs;
    uint32_t n_event_outputs;
    const uint32_t* output_channels;       /* n_outputs entries (or NULL = all f32): N > 1 makes the output a Frame<N>,
                                            * `og::Frame<N>&` in the source.  Frame edges: copy, element-wise fan-in sum,
                                            * `frame + frame`, `frame - frame`, `frame * f32`, `-frame` in compound sources;
                                            * the built-in frame node is TptFilter::<Frame<N>>::new (N = 2, 4). */
    uint32_t event_queue_capacity;         /* pushes per FRAME each event output of this type holds: 0 = the default
                                            * OG_NODE_EVENTS_PER_FRAME = 2, up to 32 = the reference's
                                            * ArrayVec<EventInstance, 32> (graph/types.rs:18).  The queue lives in
                                            * registers: a graph's kernel is built for the largest capacity among its
                                            * node types (that many VGPRs per event output). */
} og_node_type;
#define OG_NODE_EVENTS_PER_FRAME 2
#define OG_NODE_EVENTS_MAX 32
int og_register_node(const og_node_type* t);
int og_unregister_node(const char* type_ctor);

/* ---- named functions on a connection (ast.rs:126-128 `Call`; oscen-lib/tests/connection_expr_functions.rs:14-27,
 * connection_expr_function_paths.rs:15-26) -------------------------------------------------------------------------
 * `half(a.output) -> out`, `dsp::decode_ms(s.output) -> out`, `merge2(a.output, b.output) -> out`: the reference
 * passes the call through to a pure Rust function in scope.  Here that function is registered once, its body as
 * DEVICE SOURCE like a node's process(): every f32 parameter is a `const float <name>`, every Frame<N> parameter a
 * `const og::Frame<N> <name>` (`.v[i]`, `+`, `-`, `* float`), the body returns a `float` or an `og::Frame<N>`.
 * A call matches a registration by the path as written, by its last segment (what `use dsp::decode_ms;` gives), or a
 * registration under a longer path ending in it.  Calls whose path ends in `Frame` (`Frame::<2>(a, b)`,
 * `oscen::frame::Frame(a, b)`) are the frame constructor and need no registration; `x.tanh()`, `x.clamp(lo, hi)` and the
 * other f32 methods are built in.  The result of a function is a per-frame value whatever its arguments are. */
typedef struct {
    const char* name;               /* "half" or "dsp::decode_ms" */
    uint32_t n_args;                /* 1..8 */
    const char* const* arg_names;
    const uint32_t* arg_channels;   /* per argument: 0 or 1 = f32, N (2..4) = Frame<N>; NULL = all f32 */
    uint32_t result_channels;       /* 0 or 1 = f32, N = Frame<N> */
    const char* source;             /* the function body, e.g. "return x * 0.5f;" */
} og_function_type;
int og_register_function(const og_function_type* f);
int og_unregister_function(const char* name);
/* A graph description usable as a node of other graphs (`inner = InnerGraph;`, nested graphs:
 * examples/src/bin/nested_static_graph_test.rs): expanded inline, its inputs/outputs become the node's ports. */
int og_register_graph_type(const char* type_name, const og_graph_desc* g);
int og_unregister_graph_type(const char* type_name);

/* A node of the poly WRAPPER graph that runs once on the summed voices (e.g.
 * `tremolo = Tremolo::new()` with `voices.output -> tremolo.input; tremolo.output -> out`,
 * examples/electric-piano/src/main.rs:56,88-96).  Wire it with og_graph_connect:
 * "<voice output name>" -> "node.input", value inputs -> "node.rate"/"node.depth",
 * "node.output" -> "<second graph output>" (Frame<2>: the engine then has 2 channels). */
int og_graph_add_bus_node(og_graph_desc* g, const char* name, const char* type_ctor, const float* args,
                          uint32_t n_args);
int og_graph_connect(og_graph_desc* g, const char* src_expr, const char* dst, const char* policy);
/* `src -> [via] -> dst` (oscen-graph-compiler/src/ir/lower.rs:342-347): route through a declared
 * Delay node, or, when `via` is a sample count ("64"), through an anonymous Delay::new(N, 0.0).
 * Adds `src -> via.input` and the feedback edge `via.output -> dst`, which imposes no ordering:
 * a consumer scheduled before the delay reads the sample the delay produced one frame earlier. */
int og_graph_connect_via(og_graph_desc* g, const char* src_expr, const char* via, const char* dst);
/* The same description from the TEXT of a `graph! { ... }` body (the reference DSL,
 * oscen-graph-compiler/src/parse.rs:195-979): name / input / output / nodes{} /
 * connections{} with [policy] prefixes, `* N` rates and compound sources.
 * per_voice_inputs: comma-separated value inputs the poly wrapper feeds per voice
 * (e.g. "frequency"); may be NULL. */
int og_graph_parse(const char* dsl_text, const char* per_voice_inputs, og_graph_desc** out);
/* POLY WRAPPER graphs (examples/fm-synth/src/lib.rs:22-131, examples/electric-piano/src/main.rs:33-97) go through
 * og_graph_parse / og_create as written: `midi_parser = MidiParser::new(); voice_allocator = VoiceAllocator::<N>::new();
 * voice_handlers = [MidiVoiceHandler::new(); N]; voices = [Voice::new(); N];` plus an optional node fed by the voice
 * sum.  The three MIDI node kinds are this library's host-side front end (og_midi_*), `voices` is the bank (its size
 * is og_create's n_voices, not the N of the text), `voices.out -> out` the mix bus, the node after the sum the
 * post-mix stage.  The engine's inputs are: the voice input MidiVoiceHandler.frequency feeds (per voice), the one its
 * gate feeds (event), then the wrapper's value inputs with their defaults and [ramp: N] specs; the raw-MIDI event
 * input is served by og_midi_send.  The voice type must be a graph type: registered with og_register_graph_type, or
 * one of the built-in FMVoice / ElectricPianoVoiceNode.  Event outputs fed by the MIDI nodes (`midi_parser.note_on ->
 * note_on_out`) are host-side events and are not lowered.
 * og_graph_poly_info: 1 if `g` is such a wrapper (0 if not, < 0 on a malformed one), the N it declares, and the names
 * of the per-voice frequency and gate inputs to hand to og_midi_create. */
int og_graph_poly_info(const og_graph_desc* g, uint32_t* declared_voices, char* frequency_input, char* gate_input, size_t cap);
/* Print a description back as DSL text; returns the length, copies at most cap-1 bytes. */
int64_t og_graph_to_dsl(const og_graph_desc* g, char* buf, size_t cap);
void og_graph_free(og_graph_desc* g);
/* The HIP source of the fused voice kernel this description lowers to (for
 * inspection / ahead-of-time builds).  Returns the length; copies at most cap-1 bytes. */
int64_t og_graph_kernel_source(const og_graph_desc* g, char* buf, size_t cap);
/* Compile that source with hiprtc for `arch` (e.g. "gfx950") without touching a
 * device -- the path og_create() takes for graphs that were not compiled ahead
 * of time.  Returns the code-object size in bytes, or a negative OG_E_* code. */
int64_t og_graph_jit_check(const og_graph_desc* g, const char* arch);

/* ---- engine ------------------------------------------------------------------- */
/* Graph::new()  codegen/mod.rs:1309-1328: n_voices copies of the voice graph,
 * sample rate 44100 until og_init. */
int og_create(const og_graph_desc* g, uint32_t n_voices, int device_id, og_engine** out);
void og_destroy(og_engine* e);
/* init(sample_rate) = set_sample_rate + prepare  codegen/mod.rs:1335-1350, 1374-1382 */
int og_init(og_engine* e, float sample_rate);

/* name -> index of a graph input (value or event), or OG_E_INVALID */
int og_input_index(const og_engine* e, const char* name);
uint32_t og_num_inputs(const og_engine* e);

/* Generated value setters  codegen/mod.rs:917-976.  For an input declared with
 * [ramp: N]: og_set_value = set_<n>(v) (default ramp, no-op if v == target),
 * og_set_value_ramp = set_<n>_with_ramp(v, frames), og_set_value_immediate =
 * set_<n>_immediate(v).  For a plain value input all three store the value. */
int og_set_value(og_engine* e, uint32_t input, float v);
int og_set_value_ramp(og_engine* e, uint32_t input, float v, uint32_t frames);
int og_set_value_immediate(og_engine* e, uint32_t input, float v);
int og_get_value(const og_engine* e, uint32_t input, float* out);
/* The public fields of a `[ramp: N]` input -- `graph.<input>` is a ValueRampState (oscen-lib/src/graph/types.rs:300-373:
 * current, target, is_ramping() = frames_remaining > 0) -- and `graph.active_ramps`, the counter tick_ramps() keeps
 * (oscen-graph-compiler/src/codegen/mod.rs:878-914).  A plain value input reports current == target, 0 frames. */
int og_ramp_state(const og_engine* e, uint32_t input, float* current, float* target, uint32_t* frames_remaining);
uint32_t og_active_ramps(const og_engine* e);

/* Per-voice value input (the `voice_handlers.frequency -> voices.frequency`
 * edge, examples/fm-synth/src/lib.rs:88): takes effect at the next block. */
int og_set_voice_value(og_engine* e, uint32_t input, uint32_t voice, float v);
int og_set_voice_values(og_engine* e, uint32_t input, uint32_t first_voice, uint32_t count, const float* v);

/* <event_input>.try_push(EventInstance{frame_offset, Scalar(v)}) for one voice
 * (oscen-lib/src/graph/types.rs:137-241).  frame_offset is relative to the
 * next og_process_block; events at frame_offset >= frames of that block are
 * dropped exactly like the reference does (codegen/mod.rs:782-871).
 * OG_E_OVERFLOW (event dropped) past 32 events per voice per input per block. */
int og_push_voice_event(og_engine* e, uint32_t input, uint32_t voice, uint32_t frame_offset, float scalar);
/* The note-on frame also changes MidiVoiceHandler.frequency (midi.rs:91-105):
 * set a per-voice value input exactly on frame_offset of the next block. */
int og_push_voice_value(og_engine* e, uint32_t input, uint32_t voice, uint32_t frame_offset, float v);
/* Extension for bulk rendering: schedule on the absolute timeline (frames since
 * og_init); consumed by whichever block contains that frame. */
int og_schedule_voice_event(og_engine* e, uint32_t input, uint32_t voice, uint64_t abs_frame, float scalar);
int og_schedule_voice_value(og_engine* e, uint32_t input, uint32_t voice, uint64_t abs_frame, float v);
/* Bulk form of the two calls above (a whole score at once): `input` is an event input (values = scalar
 * payloads) or a per-voice value input (values = new values).  Events wait on the host until the
 * blocks they belong to are launched (with og_set_bus_batching several blocks share a launch and
 * one timeline update).  How they get to the device: a large batch rebuilds the timeline (O(V)
 * upload, one stream sync); a small one (live playing: og_push_voice_event, og_midi_*) appends
 * per-voice segments that the update kernel reads from pinned host memory -- no copy call, no
 * synchronisation. */
int og_schedule_voice_events(og_engine* e, uint32_t input, uint32_t n, const uint32_t* voices,
                             const uint64_t* abs_frames, const float* values);

/* process_block(frames)  codegen/mod.rs:755-873, frames <= 512, then copies
 * <out>_block[..frames] (the sum over voices, `voices.out -> out`) to host
 * memory: out_bus[frames * channels].  Blocking (the device writes the block's bus into pinned host
 * memory; the call spins on a completion word instead of synchronising the stream).  frames == 0
 * runs no frame and discards the events pushed for the block (their frame_offset >= frames), like
 * the generated loop. */
int og_process_block(og_engine* e, uint32_t frames, float* out_bus);
/* Same, but the bus stays in device memory (d_out_bus[frames*channels], may be
 * NULL to keep it in the engine's own buffer) and the call only enqueues work
 * on the engine's stream. */
int og_process_block_async(og_engine* e, uint32_t frames, float* d_out_bus);
/* n_blocks consecutive og_process_block_async(frames) calls in one -- the reference's render loops call process_block
 * back to back with nothing in between (BlockRender::render, oscen-lib/src/graph/offline.rs:46-90; the criterion loops,
 * oscen-lib/benches/static_vs_runtime.rs:96-116) --: block b's bus goes to d_out_bus + b * out_stride_bytes (NULL keeps
 * the buses in the engine's own buffer, where only the last one stays readable).  frames in 1..512.  Results are those
 * of n_blocks separate calls, bit for bit; what it saves is n_blocks - 1 crossings of the boundary. */
int og_process_blocks_async(og_engine* e, uint32_t frames, uint32_t n_blocks, float* d_out_bus, size_t out_stride_bytes);
int og_synchronize(og_engine* e);
/* Throughput option for streaming callers of og_process_block_async: up to `blocks` (1..32) consecutive async blocks
 * that nothing but event pushes separates (no value change, no taps) are rendered by ONE launch of the voice kernel over
 * their frames back to back -- per-voice state loaded and stored once, one inter-kernel gap, one bus reduce -- instead
 * of a launch each.  The bus of an async block is then complete after the last block of its batch, after og_flush()
 * (launches what is queued, does not wait) or og_synchronize(); every call that reads or changes engine state launches
 * the queue first, and og_process_block / og_render* always deliver complete buses.  Results are those of block-by-block
 * processing, bit for bit.  Default 1 (a launch per block); blocks = 0 picks 8..32 from the bank size (what og_render*
 * and cluster shards use: as many blocks as keep one launch's partial-sum rows within 32 MB).  Measured, fm_voice at
 * 65 536 voices: 1 -> 8 blocks +30 %, 8 -> 32 another +3.5 %. */
int og_set_bus_batching(og_engine* e, uint32_t blocks);
int og_flush(og_engine* e);
int og_set_stream(og_engine* e, void* hip_stream);
/* BlockRender::render  oscen-lib/src/graph/offline.rs:46-90: total_frames in
 * chunks of `block` (<= 512); out_bus[total_frames*channels] (host). */
int og_render(og_engine* e, uint64_t total_frames, uint32_t block, float* out_bus);

/* Stream inputs of the graph (`pub <stream_in>_block: [f32; 512]`, codegen/mod.rs:1196): the caller fills the block
 * before process_block; every voice of the bank reads the same samples.  The buffer keeps its contents between blocks,
 * like the generated field.  A Frame<N> input (`[Frame<N>; 512]`) takes n FRAMES of N interleaved samples. */
int og_set_stream_block(og_engine* e, uint32_t input, const float* samples, uint32_t n);
uint32_t og_num_stream_inputs(const og_engine* e); /* BlockRender::NUM_STREAM_INPUTS */
uint32_t og_stream_input_channels(const og_engine* e, uint32_t input); /* 1 = f32, N = Frame<N>; 0: not a stream input */
/* BlockRender::render(inputs, tail)  oscen-lib/src/graph/offline.rs:46-90: one buffer per stream input (declaration
 * order; lengths in FRAMES, a Frame<N> input holds N interleaved samples per frame -- `render_mono(&[Frame<2>], tail)`,
 * oscen-lib/tests/stereo_render.rs:96-99), total = max input length + tail frames, shorter inputs padded with silence, chunks of 512 frames.
 * out_bus[total * channels] (host); *frames_rendered = total (call with out_bus == NULL and total known = 0 is a no-op). */
int og_render_inputs(og_engine* e, const float* const* inputs, const uint64_t* input_lens, uint32_t n_inputs,
                     uint64_t tail, float* out_bus, uint64_t* frames_rendered);

/* `graph.<node>.<field>`: the fields of the generated struct's nodes are public in the reference and its fixtures read
 * them after process() (`graph.sinks[i].last`, oscen-lib/tests/connection_expr_functions.rs:300-330;
 * `graph.inner.dummy.val`, examples/src/bin/nested_static_graph_test.rs:71).  A node's persistent fields (the `state`
 * of a registered node type, the built-in nodes' filter / envelope / phase words) are planes of the state image:
 * path = "node.field", "array[i].field" or "nested.node.field"; out[n] receives the field of voices first_voice.. as
 * 4-byte words (f32 fields as floats, u32 fields as integers).  Flushes queued blocks first.  The index call returns
 * the plane number or -1. */
int og_state_field_index(const og_engine* e, const char* path);
int og_read_state_field(og_engine* e, const char* path, uint32_t first_voice, uint32_t n, void* out);

/* Introspection of `graph.voices[i].<out>` (tests poke node fields in the
 * reference): record the per-voice output of the listed voices during the
 * following blocks.  og_read_voice_taps copies the last block: out[n*frames]. */
int og_set_voice_taps(og_engine* e, const uint32_t* voices, uint32_t n);
int og_read_voice_taps(og_engine* e, float* out, uint32_t n, uint32_t frames);

uint32_t og_channels(const og_engine* e);
/* 1: every voice contributes an f32 sample per frame; 2: the graph's stream output is fed a Frame<2> (e.g. a per-voice
 * pan): both channels are summed over the voices, the bus is interleaved L R and og_read_voice_taps delivers
 * [tap][frame][2]; up to 4 with several stream outputs (below).  (og_channels is also 2 for a mono voice sum with a
 * stereo post-mix node.) */
uint32_t og_voice_channels(const og_engine* e);
/* A voice graph with SEVERAL stream outputs (`output out_a: stream; output out_b: stream; output out: stream;`,
 * oscen-lib/tests/multirate_graph.rs:444-458 -- the reference exposes `graph.out_a`, `graph.out_b`, `graph.out`) has one
 * bus channel per output (a Frame<2> output: two), in declaration order, at most 4: out_bus[frames][og_channels()], taps
 * [tap][frame][og_voice_channels()].  og_output_channel tells where an output sits in a frame. */
int og_output_channel(const og_engine* e, const char* name, uint32_t* offset, uint32_t* width);
uint32_t og_num_voices(const og_engine* e);
uint32_t og_latency_samples(const og_engine* e); /* emit_struct.rs:534-570 */
/* The post-mix node of a wrapper graph (`voices.output -> tremolo.input`, electric-piano/src/main.rs:88-96): 0 = none,
 * 1 = a node that is LINEAR in its input (Tremolo: out = in * pan, tremolo.rs:40-62), 2 = any other.  A host that sums the
 * buses of several engines itself (oscen_amd/distributed.py) may sum AFTER a linear post-mix stage; for kind 2 it must sum
 * the voice sums first and run the node once, as og_cluster_* does. */
int og_post_mix_kind(const og_engine* e);
uint64_t og_frames_processed(const og_engine* e);
/* layout facts used by the roofline accounting */
uint32_t og_state_words_per_voice(const og_engine* e);
/* words a steady-state block writes back (read-mostly tables -- e-piano decay/release/rotation multipliers -- are
 * only stored in blocks whose events rewrote them) */
uint32_t og_state_words_written_per_voice(const og_engine* e);
uint32_t og_lanes_per_voice(const og_engine* e); /* 1, or 4 for graphs with per-harmonic arrays (8 harmonics per lane) */
int og_uses_split_kernel(const og_engine* e); /* pipeline depth of the launched kernel: 0 = one wave per 64 voices,
                                                 2 or 4 = that many waves per 64 voices (small banks) */
uint32_t og_voices_per_wave(const og_engine* e); /* 64, or 32/16 when that puts two waves on every SIMD */
/* events that were never delivered: try_push overflows and late block-local events (host side) + pushes the in-voice
 * event queues of user nodes could not hold (OG_NODE_EVENTS_PER_FRAME per frame and output; counted on the device).
 * A pure getter (safe to poll from a monitoring thread): the device-side count is what og_sync_event_counters(),
 * og_synchronize() or og_read_output_events() -- called by the rendering thread -- last folded in. */
uint64_t og_events_dropped(const og_engine* e);
/* launches the queued blocks, synchronises the stream and folds the device-side "pushes lost" counter into
 * og_events_dropped(); a no-op for graphs without in-voice event queues.  Errors are reported, not swallowed. */
int og_sync_event_counters(og_engine* e);

/* ---- event OUTPUTS of the graph (`output x: event;` fed by a node's #[output(event)] field: EventOutput,
 * oscen-lib/src/graph/types.rs:137-241; in the reference the caller iterates `graph.x` after process_block).
 * With N voices the events of all voices go to one device log; og_read_output_events moves what has arrived since the
 * last call to the host, ordered by (frame, voice, push order).  `frame` is absolute (frames since og_init):
 * frame_offset within the last block = frame - (og_frames_processed() - frames of that block).  The log holds
 * max(65 536, 4 x voices) events between two reads; what does not fit THE LOG is counted in *n_overflowed.  Events that
 * were drained from the log but did not fit `buf` stay queued on the host and come first in the next call. */
typedef struct {
    uint32_t voice;
    uint32_t output; /* og_event_output_index */
    uint64_t frame;
    float value;     /* scalar payload */
    uint32_t reserved;
} og_out_event;
uint32_t og_num_event_outputs(const og_engine* e);
int og_event_output_index(const og_engine* e, const char* name);
int og_read_output_events(og_engine* e, og_out_event* buf, uint32_t cap, uint32_t* n, uint64_t* n_overflowed);
/* which voice kernel runs: FNV-1a hash of the generated kernel body (the `<hash>` in the kernel names
 * og_k_<hash>_*, og_k2_<hash>_*, ... that rocprofv3 reports), and whether it came from hiprtc */
uint64_t og_kernel_hash(const og_engine* e);
int og_kernel_is_jit(const og_engine* e);
const char* og_kernel_name(const og_engine* e); /* "og_k_<hash>" / "og_k2_<hash>" / "og_k4_<hash>": the launched variant family */
uint32_t og_partial_rows(const og_engine* e);   /* partial bus rows one launch writes (one per workgroup) */
/* og_bus_reduce launches of the last block: 1, or 1 + the levels of the multi-pass tree (> 1024 partial rows) */
uint32_t og_bus_reduce_passes(const og_engine* e);
/* event-path counters: full timeline rebuilds, incremental (per-voice segment) updates, events resident */
int og_event_stats(const og_engine* e, uint64_t* full_rebuilds, uint64_t* incremental_updates, uint64_t* resident_events);
/* times the live path's append pointer wrapped around the device event buffer (a ring: the space of consumed /
 * superseded segments is reused, so steady live playing never needs the O(V) timeline rebuild) */
uint64_t og_event_ring_wraps(const og_engine* e);
/* events the incremental path has written to the ring so far: the pushes themselves plus whatever it carried over of the
 * voice's waiting events.  A live message on a voice with a long resident score ahead of it carries over only what is due
 * up to the end of the launch being prepared -- the rest of the score stays in place as the voice's continuation, and the
 * engine points the voice at it before the launch in which its first event is due (the reference merges staged events into
 * the block by frame_offset per block, oscen-graph-compiler/src/codegen/mod.rs:782-871; here the merge is by absolute
 * frame and a message costs its own records whatever the score's length) */
uint64_t og_events_copied(const og_engine* e);
/* room, in events, the device timeline keeps BEHIND a bulk score for live pushes (default: half the score + 2 M).
 * A live push re-writes the touched voice's remaining events as a fresh segment at the tail; a voice that still has a
 * long resident score ahead of it costs that many events per push, and once the tail meets the (still live) score the
 * timeline is rebuilt -- O(score), a missed audio deadline.  A host that plays live on top of a long resident score
 * sizes the room once, before the score goes to the device.  No counterpart in the reference (its queues are the
 * fixed 32-deep ArrayVecs of graph/types.rs:18; the score there is the caller's own loop). */
int og_reserve_events(og_engine* e, uint64_t n_events);
/* Voice grouping for resident scores (no counterpart in the reference, whose `voices[i]` are one array in one loop:
 * fm-synth/src/lib.rs:22-131).  A wave renders 64 consecutive voice SLOTS and takes, chunk by chunk, the cheapest body all
 * of its lanes allow; policy 1 re-orders the slots so that voices whose notes end at about the same time share waves (by
 * the frame of the voice's first scheduled note-off -- an event with a value <= 0 --, then by its first event), policy 2
 * additionally deals those groups of 64 slots out so that the workgroups a CU is handed carry about the same number of
 * events (banks of >= 32 768 voices; experimental), policy 0 restores the identity.  Voice NUMBERS do not change: every entry point keeps taking and handing out the caller's
 * numbers (events, per-voice values, taps, state fields, event outputs, MIDI voices); per-voice samples are bit for bit
 * those of the ungrouped bank, the bus differs by the association of the sum.  Call after og_init and after scheduling
 * the score, before the first block and before og_set_voice_taps (OG_E_STATE otherwise); og_init restores the identity;
 * og_save_state blobs of a grouped engine carry the order and og_load_state adopts it. */
int og_group_voices(og_engine* e, uint32_t policy);
int og_voice_slot(const og_engine* e, uint32_t voice, uint32_t* slot); /* the physical slot of a voice (== voice when not grouped) */
/* the blocking entry (og_process_block / og_midi_process_block): calls that waited on the completion word, and how
 * many of those waits ended because the stream was found finished (hipStreamQuery, asked every 128 us from 256 us on)
 * before the completion word was seen -- 0 in the ordinary case */
int og_blocking_stats(const og_engine* e, uint64_t* calls, uint64_t* marker_timeouts);
/* average device time of the voice kernel over the launches since the last
 * call (HIP events on the engine's stream); returns <0 if timing is off */
int og_enable_kernel_timing(og_engine* e, int on);
double og_kernel_time_ms(og_engine* e, uint32_t* n_launches);
uint64_t og_kernel_blocks_timed(const og_engine* e); /* blocks those launches covered (> launches with og_set_bus_batching) */
/* the shader clock the launches averaged by the LAST og_kernel_time_ms call ran at, measured by the voice kernel itself: its
 * first workgroup reads the shader-cycle counter and the constant 100 MHz counter at its first and last instruction (0 when
 * timing was off or the device has no such counters) */
double og_kernel_clock_ghz(const og_engine* e);
/* the shader clock the device runs at NOW: a one-wave probe on the engine's stream counts shader cycles (s_memtime) over
 * 20 us of the constant 100 MHz counter (s_memrealtime); blocks until it has run.  A measurement aid: the clock governor
 * takes tens of milliseconds of load to settle, so short bursts of blocks run slower than a steady stream of them. */
int og_shader_clock_ghz(og_engine* e, double* ghz);

/* State snapshot: the DSP state (SoA planes, per-harmonic arrays, post-mix phase, delay lines) followed by a control
 * block -- frame counter, value / ValueRampState of every input, every event that has not fired yet -- so that loading
 * it into an engine of the same graph, voice count and sample rate continues the render sample for sample, also from
 * the middle of a ramp or with notes still scheduled.  og_state_bytes() is the size needed NOW (it grows with the
 * number of pending events). */
size_t og_state_bytes(const og_engine* e);
int og_save_state(og_engine* e, void* dst, size_t cap);
int og_load_state(og_engine* e, const void* src, size_t len);

/* ---- multi-GPU: voice banks sharded over the GPUs of one node ----------------------------
 * The poly wrapper's `voices.out -> out` is a plain sum (codegen/emit_node.rs:463-466), so a bank shards
 * by contiguous GLOBAL voice ranges: shard s = voices [s*V/n, (s+1)*V/n) on device_ids[s] (a device may be
 * listed more than once).  Broadcast setters go to every shard, per-voice calls are routed by global voice
 * id.  The only exchange on the data path is the mix bus: the mono buses of a batch of blocks are summed
 * with ONE ncclReduce (RCCL over xGMI, root = device_ids[0]) per batch -- og_cluster_render renders up to
 * 256 blocks per reduce and overlaps the reduce of a batch with the kernels of the next.  A post-mix node
 * (Tremolo -> Frame<2>) runs once, on the root, after the reduce, as in the reference
 * (examples/electric-piano/src/main.rs:88-96); fm-synth's mono bus is duplicated to L/R by the caller after
 * the sum (examples/fm-synth/src/lib.rs:269-274).  Same conventions as og_engine: not thread-safe per
 * cluster, no CPU fallback. */
typedef struct og_cluster og_cluster;
int og_cluster_create(const og_graph_desc* g, uint64_t n_voices_total, const int* device_ids, uint32_t n_shards,
                      og_cluster** out);
void og_cluster_destroy(og_cluster* c);
int og_cluster_init(og_cluster* c, float sample_rate);
int og_cluster_input_index(const og_cluster* c, const char* name);
int og_cluster_set_value(og_cluster* c, uint32_t input, float v);
int og_cluster_set_value_ramp(og_cluster* c, uint32_t input, float v, uint32_t frames);
int og_cluster_set_value_immediate(og_cluster* c, uint32_t input, float v);
int og_cluster_set_voice_values(og_cluster* c, uint32_t input, uint64_t first_voice, uint64_t count, const float* v);
int og_cluster_push_voice_event(og_cluster* c, uint32_t input, uint64_t voice, uint32_t frame_offset, float scalar);
int og_cluster_push_voice_value(og_cluster* c, uint32_t input, uint64_t voice, uint32_t frame_offset, float v);
int og_cluster_schedule_voice_events(og_cluster* c, uint32_t input, uint64_t n, const uint64_t* voices,
                                     const uint64_t* abs_frames, const float* values);
/* process_block(frames) on every shard + the bus reduce; out_bus[frames * channels] (host).  Blocking. */
int og_cluster_process_block(og_cluster* c, uint32_t frames, float* out_bus);
/* BlockRender::render over the whole cluster: total_frames in blocks of `block`, one reduce per 256 blocks. */
int og_cluster_render(og_cluster* c, uint64_t total_frames, uint32_t block, float* out_bus);
uint32_t og_cluster_num_shards(const og_cluster* c);
uint32_t og_cluster_num_devices(const og_cluster* c); /* distinct GPUs = ranks of the RCCL communicator */
uint64_t og_cluster_num_voices(const og_cluster* c);
uint32_t og_cluster_channels(const og_cluster* c);
uint64_t og_cluster_rccl_reduces(const og_cluster* c); /* ncclReduce batches issued so far (0 on a one-device cluster) */
/* device time of those reduces on the root's stream (HIP events around each batched ncclReduce, includes waiting for the
 * slowest device): enable, render, then read the sum and count since enabling (reading clears them) */
int og_cluster_enable_reduce_timing(og_cluster* c, int on);
int og_cluster_reduce_time_ms(og_cluster* c, double* total_ms, uint64_t* n_reduces);
/* event outputs of the graph (og_read_output_events) over all shards: merged into (frame, GLOBAL voice, push order) */
int og_cluster_read_output_events(og_cluster* c, og_out_event* buf, uint32_t cap, uint32_t* n, uint64_t* n_overflowed);
uint64_t og_cluster_events_dropped(og_cluster* c); /* og_events_dropped summed over the shards */
/* og_sync_event_counters on every shard; returns the first error (og_cluster_events_dropped folds the counters the same way
 * but can only leave an error in og_last_error) */
int og_cluster_sync_event_counters(og_cluster* c);
int og_cluster_group_voices(og_cluster* c, uint32_t policy); /* og_group_voices on every shard (voices never change shard) */
/* the engine of shard s (owned by the cluster) and its first global voice: taps, state snapshots, statistics */
og_engine* og_cluster_shard(og_cluster* c, uint32_t s, uint64_t* first_voice);

/* ---- MIDI front end (host, control rate): MidiParser -> VoiceAllocator<N> -> MidiVoiceHandler
 * (oscen-lib/src/midi.rs:40-225, voice_allocator.rs:46-136) with N = the engine's voice count.
 * og_midi_send = `midi_in.try_push(raw_midi_event(bytes))` with frame_offset; messages are applied
 * in frame order by og_midi_process_block (= flush + og_process_block).  With engine == NULL the
 * object runs detached over n_voices and logs what the voices would receive (og_midi_pop_output). */
typedef struct og_midi og_midi;
int og_midi_create(og_engine* e, uint32_t n_voices, const char* frequency_input, const char* gate_input, og_midi** out);
/* The same front end in front of a multi-GPU bank: one allocator over the cluster's GLOBAL voice ids (N = the whole
 * bank, the decisions of voice_allocator.rs:57-136 for a single bank of that size), every per-voice message routed
 * to the shard that owns the voice; og_midi_process_block = flush + og_cluster_process_block. */
int og_midi_create_cluster(og_cluster* c, const char* frequency_input, const char* gate_input, og_midi** out);
void og_midi_destroy(og_midi* m);
/* MidiVoiceHandler::midi_note_to_freq (midi.rs:69-72): 440 * 2^((note - 69) / 12) in f32 through the platform powf */
float og_midi_note_to_freq(uint8_t note);
int og_midi_send(og_midi* m, const uint8_t* bytes, uint32_t len, uint32_t frame_offset);
/* n three-byte messages at once: bytes3[3*n], frame_offsets[n] */
int og_midi_send_batch(og_midi* m, const uint8_t* bytes3, const uint32_t* frame_offsets, uint32_t n);
/* `midi_in` is an ArrayVec<EventInstance, 32> in the reference (graph/types.rs:18): the 33rd message queued for a
 * block is dropped (OG_E_OVERFLOW).  A bank-sized instrument raises the capacity. */
int og_midi_set_queue_capacity(og_midi* m, uint32_t capacity);
uint64_t og_midi_dropped(const og_midi* m); /* queue overflows + messages whose frame_offset >= the block's frames */
int og_midi_flush(og_midi* m);
int og_midi_process_block(og_midi* m, uint32_t frames, float* out_bus);
/* the same over og_process_block_async: the bus stays on the device (d_out_bus or the engine's own buffer), so the
 * host can parse the next block's messages while this block renders */
int og_midi_process_block_async(og_midi* m, uint32_t frames, float* d_out_bus);
int og_midi_voice_state(const og_midi* m, uint32_t voice, int* active, int* released, int* note, uint32_t* age);
int og_midi_pop_output(og_midi* m, uint32_t* voice, uint32_t* frame, float* frequency, int* has_frequency, float* gate);

/* Output step: interleaved bus -> RIFF/WAVE file, 16-bit PCM or 32-bit float (what the reference
 * examples do with the `hound` crate after BlockRender::render). */
int og_write_wav(const char* path, const float* interleaved, uint64_t frames, uint32_t channels,
                 uint32_t sample_rate, uint32_t bits_per_sample);

const char* og_last_error(void);
const char* og_version(void);

#ifdef __cplusplus
}
#endif
#endif
