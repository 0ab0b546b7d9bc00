// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no rustc/cargo): shipped as source for the
// maintainer of the reference.  Kept in sync with INTEGRATION.md (tests/test_capi_cpu.py checks it).
// oscen-gpu-sys/src/lib.rs
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_float, c_int, c_void};

#[repr(C)] pub struct og_graph_desc { _p: [u8; 0] }
#[repr(C)] pub struct og_engine { _p: [u8; 0] }

pub const OG_KIND_VALUE: c_int = 0;
pub const OG_KIND_EVENT: c_int = 1;
pub const OG_KIND_STREAM: c_int = 2;
pub const OG_IN_PER_VOICE: u32 = 1;
pub const OG_E_OVERFLOW: c_int = -5;

#[link(name = "oscen_gpu")]
extern "C" {
    pub fn og_graph_new(name: *const c_char, out: *mut *mut og_graph_desc) -> c_int;
    pub fn og_graph_builtin(name: *const c_char, out: *mut *mut og_graph_desc) -> c_int;
    pub fn og_graph_add_input(g: *mut og_graph_desc, name: *const c_char, kind: c_int,
                              default_value: c_float, ramp_frames: u32, flags: u32) -> c_int;
    pub fn og_graph_add_output(g: *mut og_graph_desc, name: *const c_char, kind: c_int) -> c_int;
    pub fn og_graph_add_node(g: *mut og_graph_desc, name: *const c_char, type_ctor: *const c_char,
                             args: *const c_float, n_args: u32, rate_factor: u32) -> c_int;
    pub fn og_graph_connect(g: *mut og_graph_desc, src: *const c_char, dst: *const c_char,
                            policy: *const c_char) -> c_int;
    pub fn og_graph_free(g: *mut og_graph_desc);
    pub fn og_create(g: *const og_graph_desc, n_voices: u32, device: c_int,
                     out: *mut *mut og_engine) -> c_int;
    pub fn og_destroy(e: *mut og_engine);
    pub fn og_init(e: *mut og_engine, sample_rate: c_float) -> c_int;
    pub fn og_input_index(e: *const og_engine, name: *const c_char) -> c_int;
    pub fn og_set_value(e: *mut og_engine, input: u32, v: c_float) -> c_int;
    pub fn og_set_value_ramp(e: *mut og_engine, input: u32, v: c_float, frames: u32) -> c_int;
    pub fn og_set_value_immediate(e: *mut og_engine, input: u32, v: c_float) -> c_int;
    pub fn og_set_voice_value(e: *mut og_engine, input: u32, voice: u32, v: c_float) -> c_int;
    pub fn og_push_voice_event(e: *mut og_engine, input: u32, voice: u32, frame_offset: u32,
                               scalar: c_float) -> c_int;
    pub fn og_push_voice_value(e: *mut og_engine, input: u32, voice: u32, frame_offset: u32,
                               v: c_float) -> c_int;
    pub fn og_process_block(e: *mut og_engine, frames: u32, out_bus: *mut c_float) -> c_int;
    pub fn og_process_block_async(e: *mut og_engine, frames: u32, d_out_bus: *mut c_void) -> c_int;
    pub fn og_process_blocks_async(e: *mut og_engine, frames: u32, n_blocks: u32, d_out_bus: *mut c_void, out_stride_bytes: usize) -> c_int;
    pub fn og_synchronize(e: *mut og_engine) -> c_int;
    pub fn og_render(e: *mut og_engine, total_frames: u64, block: u32, out: *mut c_float) -> c_int;
    pub fn og_latency_samples(e: *const og_engine) -> u32;
    pub fn og_post_mix_kind(e: *const og_engine) -> c_int;
    pub fn og_last_error() -> *const c_char;
}

// ---- the rest of include/oscen_gpu.h (DSL text front end, feedback edges, bus nodes, MIDI, WAV, state) ----
#[repr(C)] pub struct og_midi { _p: [u8; 0] }
extern "C" {
    pub fn og_graph_add_bus_node(g: *mut og_graph_desc, name: *const c_char, type_ctor: *const c_char,
                                 args: *const c_float, n_args: u32) -> c_int;
    pub fn og_graph_connect_via(g: *mut og_graph_desc, src: *const c_char, via: *const c_char,
                                dst: *const c_char) -> c_int;
    pub fn og_graph_parse(dsl_text: *const c_char, per_voice_inputs: *const c_char,
                          out: *mut *mut og_graph_desc) -> c_int;
    pub fn og_graph_to_dsl(g: *const og_graph_desc, buf: *mut c_char, cap: usize) -> i64;
    pub fn og_set_voice_values(e: *mut og_engine, input: u32, first: u32, count: u32, v: *const c_float) -> c_int;
    pub fn og_schedule_voice_event(e: *mut og_engine, input: u32, voice: u32, abs_frame: u64, scalar: c_float) -> c_int;
    pub fn og_schedule_voice_value(e: *mut og_engine, input: u32, voice: u32, abs_frame: u64, v: c_float) -> c_int;
    pub fn og_set_stream(e: *mut og_engine, hip_stream: *mut c_void) -> c_int;
    pub fn og_channels(e: *const og_engine) -> u32;
    pub fn og_num_voices(e: *const og_engine) -> u32;
    pub fn og_state_bytes(e: *const og_engine) -> usize;
    pub fn og_save_state(e: *mut og_engine, dst: *mut c_void, cap: usize) -> c_int;
    pub fn og_load_state(e: *mut og_engine, src: *const c_void, len: usize) -> c_int;
    pub fn og_midi_create(e: *mut og_engine, n_voices: u32, frequency_input: *const c_char,
                          gate_input: *const c_char, out: *mut *mut og_midi) -> c_int;
    pub fn og_midi_destroy(m: *mut og_midi);
    pub fn og_midi_send(m: *mut og_midi, bytes: *const u8, len: u32, frame_offset: u32) -> c_int;
    pub fn og_midi_flush(m: *mut og_midi) -> c_int;
    pub fn og_midi_process_block(m: *mut og_midi, frames: u32, out_bus: *mut c_float) -> c_int;
    pub fn og_write_wav(path: *const c_char, interleaved: *const c_float, frames: u64, channels: u32,
                        sample_rate: u32, bits: u32) -> c_int;
}

// ---- round 2: bulk scheduling, stream inputs / render(inputs, tail), node arrays, custom nodes, nested graphs,
// ---- multi-GPU clusters, async MIDI ----
#[repr(C)] pub struct og_cluster { _p: [u8; 0] }
#[repr(C)] pub struct og_node_port { pub name: *const c_char, pub kind: c_int, pub default_value: c_float, pub ctor_arg: c_int, pub channels: u32 }
#[repr(C)] pub struct og_node_field { pub name: *const c_char, pub is_uint: c_int, pub init: c_float, pub init_uint: u32, pub ctor_arg: c_int }
#[repr(C)] pub struct og_node_type {
    pub type_ctor: *const c_char, pub n_ctor_args: u32,
    pub inputs: *const og_node_port, pub n_inputs: u32,
    pub outputs: *const *const c_char, pub n_outputs: u32,
    pub state: *const og_node_field, pub n_state: u32,
    pub process_src: *const c_char, pub event_handler_src: *const *const c_char, pub cost_hint: u32,
    pub event_outputs: *const *const c_char, pub n_event_outputs: u32,
    pub output_channels: *const u32,
    pub event_queue_capacity: u32,
}
extern "C" {
    pub fn og_graph_add_node_array(g: *mut og_graph_desc, name: *const c_char, type_ctor: *const c_char,
                                   args: *const c_float, n_args: u32, rate_factor: u32, length: u32) -> c_int;
    pub fn og_register_node(t: *const og_node_type) -> c_int;
    pub fn og_unregister_node(type_ctor: *const c_char) -> c_int;
    pub fn og_register_graph_type(type_name: *const c_char, g: *const og_graph_desc) -> c_int;
    pub fn og_unregister_graph_type(type_name: *const c_char) -> c_int;
    pub fn og_schedule_voice_events(e: *mut og_engine, input: u32, n: u32, voices: *const u32,
                                    abs_frames: *const u64, values: *const c_float) -> c_int;
    pub fn og_set_stream_block(e: *mut og_engine, input: u32, samples: *const c_float, n: u32) -> c_int;
    pub fn og_num_stream_inputs(e: *const og_engine) -> u32;
    pub fn og_render_inputs(e: *mut og_engine, inputs: *const *const c_float, input_lens: *const u64, n_inputs: u32,
                            tail: u64, out_bus: *mut c_float, frames_rendered: *mut u64) -> c_int;
    pub fn og_kernel_hash(e: *const og_engine) -> u64;
    pub fn og_voice_channels(e: *const og_engine) -> u32;
    pub fn og_midi_send_batch(m: *mut og_midi, bytes3: *const u8, frame_offsets: *const u32, n: u32) -> c_int;
    pub fn og_midi_set_queue_capacity(m: *mut og_midi, capacity: u32) -> c_int;
    pub fn og_midi_process_block_async(m: *mut og_midi, frames: u32, d_out_bus: *mut c_float) -> c_int;
    pub fn og_midi_note_to_freq(note: u8) -> c_float;
    pub fn og_cluster_create(g: *const og_graph_desc, n_voices_total: u64, device_ids: *const c_int, n_shards: u32,
                             out: *mut *mut og_cluster) -> c_int;
    pub fn og_cluster_destroy(c: *mut og_cluster);
    pub fn og_cluster_init(c: *mut og_cluster, sample_rate: c_float) -> c_int;
    pub fn og_cluster_input_index(c: *const og_cluster, name: *const c_char) -> c_int;
    pub fn og_cluster_set_value(c: *mut og_cluster, input: u32, v: c_float) -> c_int;
    pub fn og_cluster_set_value_ramp(c: *mut og_cluster, input: u32, v: c_float, frames: u32) -> c_int;
    pub fn og_cluster_set_voice_values(c: *mut og_cluster, input: u32, first_voice: u64, count: u64, v: *const c_float) -> c_int;
    pub fn og_cluster_push_voice_event(c: *mut og_cluster, input: u32, voice: u64, frame_offset: u32, scalar: c_float) -> c_int;
    pub fn og_cluster_push_voice_value(c: *mut og_cluster, input: u32, voice: u64, frame_offset: u32, v: c_float) -> c_int;
    pub fn og_cluster_process_block(c: *mut og_cluster, frames: u32, out_bus: *mut c_float) -> c_int;
    pub fn og_cluster_render(c: *mut og_cluster, total_frames: u64, block: u32, out_bus: *mut c_float) -> c_int;
    pub fn og_cluster_channels(c: *const og_cluster) -> u32;
    pub fn og_graph_poly_info(g: *const og_graph_desc, declared_voices: *mut u32, frequency_input: *mut c_char,
                              gate_input: *mut c_char, cap: usize) -> c_int;
    pub fn og_midi_create_cluster(c: *mut og_cluster, frequency_input: *const c_char, gate_input: *const c_char,
                                  out: *mut *mut og_midi) -> c_int;
    pub fn og_blocking_stats(e: *const og_engine, calls: *mut u64, marker_timeouts: *mut u64) -> c_int;
    pub fn og_event_ring_wraps(e: *const og_engine) -> u64;
    pub fn og_events_copied(e: *const og_engine) -> u64;
    pub fn og_reserve_events(e: *mut og_engine, n_events: u64) -> c_int;
    pub fn og_group_voices(e: *mut og_engine, policy: u32) -> c_int;
    pub fn og_voice_slot(e: *const og_engine, voice: u32, slot: *mut u32) -> c_int;
    pub fn og_sync_event_counters(e: *mut og_engine) -> c_int;
    pub fn og_cluster_enable_reduce_timing(c: *mut og_cluster, on: c_int) -> c_int;
    pub fn og_cluster_reduce_time_ms(c: *mut og_cluster, total_ms: *mut f64, n_reduces: *mut u64) -> c_int;
    pub fn og_ramp_state(e: *const og_engine, input: u32, current: *mut c_float, target: *mut c_float, frames_remaining: *mut u32) -> c_int;
    pub fn og_active_ramps(e: *const og_engine) -> u32;
}

// ---- round 3 (late): named functions on a connection, Frame<N> stream inputs ----
#[repr(C)] pub struct og_function_type {
    pub name: *const c_char, pub n_args: u32, pub arg_names: *const *const c_char, pub arg_channels: *const u32,
    pub result_channels: u32, pub source: *const c_char,
}
extern "C" {
    pub fn og_register_function(f: *const og_function_type) -> c_int;
    pub fn og_unregister_function(name: *const c_char) -> c_int;
    pub fn og_stream_input_channels(e: *const og_engine, input: u32) -> u32;
}
extern "C" {
    pub fn og_state_field_index(e: *const og_engine, path: *const c_char) -> c_int;
    pub fn og_read_state_field(e: *mut og_engine, path: *const c_char, first_voice: u32, n: u32, out: *mut c_void) -> c_int;
}
extern "C" {
    pub fn og_cluster_read_output_events(c: *mut og_cluster, buf: *mut og_out_event, cap: u32, n: *mut u32, n_overflowed: *mut u64) -> c_int;
    pub fn og_cluster_events_dropped(c: *mut og_cluster) -> u64;
    pub fn og_cluster_sync_event_counters(c: *mut og_cluster) -> c_int;
    pub fn og_cluster_group_voices(c: *mut og_cluster, policy: u32) -> c_int;
}

// ---- the rest of include/oscen_gpu.h: event outputs of the graph, taps, introspection, statistics ----
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct og_out_event { pub voice: u32, pub output: u32, pub frame: u64, pub value: c_float, pub reserved: u32 }
extern "C" {
    pub fn og_version() -> *const c_char;
    pub fn og_num_inputs(e: *const og_engine) -> u32;
    pub fn og_get_value(e: *const og_engine, input: u32, out: *mut c_float) -> c_int;
    pub fn og_flush(e: *mut og_engine) -> c_int;
    pub fn og_set_bus_batching(e: *mut og_engine, blocks: u32) -> c_int;
    pub fn og_frames_processed(e: *const og_engine) -> u64;
    pub fn og_num_event_outputs(e: *const og_engine) -> u32;
    pub fn og_event_output_index(e: *const og_engine, name: *const c_char) -> c_int;
    pub fn og_read_output_events(e: *mut og_engine, buf: *mut og_out_event, cap: u32, n: *mut u32, n_overflowed: *mut u64) -> c_int;
    pub fn og_output_channel(e: *const og_engine, name: *const c_char, offset: *mut u32, width: *mut u32) -> c_int;
    pub fn og_events_dropped(e: *const og_engine) -> u64;
    pub fn og_event_stats(e: *const og_engine, full_rebuilds: *mut u64, incremental_updates: *mut u64, resident_events: *mut u64) -> c_int;
    pub fn og_set_voice_taps(e: *mut og_engine, voices: *const u32, n: u32) -> c_int;
    pub fn og_read_voice_taps(e: *mut og_engine, out: *mut c_float, n: u32, frames: u32) -> c_int;
    pub fn og_enable_kernel_timing(e: *mut og_engine, on: c_int) -> c_int;
    pub fn og_kernel_time_ms(e: *mut og_engine, n_launches: *mut u32) -> f64;
    pub fn og_shader_clock_ghz(e: *mut og_engine, ghz: *mut f64) -> c_int;
    pub fn og_kernel_clock_ghz(e: *const og_engine) -> f64;
    pub fn og_kernel_blocks_timed(e: *const og_engine) -> u64;
    pub fn og_kernel_is_jit(e: *const og_engine) -> c_int;
    pub fn og_kernel_name(e: *const og_engine) -> *const c_char;
    pub fn og_uses_split_kernel(e: *const og_engine) -> c_int;
    pub fn og_lanes_per_voice(e: *const og_engine) -> u32;
    pub fn og_voices_per_wave(e: *const og_engine) -> u32;
    pub fn og_partial_rows(e: *const og_engine) -> u32;
    pub fn og_bus_reduce_passes(e: *const og_engine) -> u32;
    pub fn og_state_words_per_voice(e: *const og_engine) -> u32;
    pub fn og_state_words_written_per_voice(e: *const og_engine) -> u32;
    pub fn og_graph_kernel_source(g: *const og_graph_desc, buf: *mut c_char, cap: usize) -> i64;
    pub fn og_graph_jit_check(g: *const og_graph_desc, arch: *const c_char) -> i64;
    pub fn og_midi_dropped(m: *const og_midi) -> u64;
    pub fn og_midi_pop_output(m: *mut og_midi, voice: *mut u32, frame: *mut u32, frequency: *mut c_float,
                              has_frequency: *mut c_int, gate: *mut c_float) -> c_int;
    pub fn og_midi_voice_state(m: *const og_midi, voice: u32, active: *mut c_int, released: *mut c_int, note: *mut c_int,
                               age: *mut u32) -> c_int;
    pub fn og_cluster_num_devices(c: *const og_cluster) -> u32;
    pub fn og_cluster_num_shards(c: *const og_cluster) -> u32;
    pub fn og_cluster_num_voices(c: *const og_cluster) -> u64;
    pub fn og_cluster_rccl_reduces(c: *const og_cluster) -> u64;
    pub fn og_cluster_shard(c: *mut og_cluster, s: u32, first_voice: *mut u64) -> *mut og_engine;
    pub fn og_cluster_set_value_immediate(c: *mut og_cluster, input: u32, v: c_float) -> c_int;
    pub fn og_cluster_schedule_voice_events(c: *mut og_cluster, input: u32, n: u64, voices: *const u64,
                                            abs_frames: *const u64, values: *const c_float) -> c_int;
}
