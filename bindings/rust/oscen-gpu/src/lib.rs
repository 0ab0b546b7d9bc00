// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no rustc/cargo): shipped as source for the
// maintainer of the reference.  Kept in sync with INTEGRATION.md (tests/test_capi_cpu.py checks the sys crate).
//
// oscen-gpu: the public surface of a `graph!`-generated struct (oscen-graph-compiler/src/codegen/mod.rs:1292-1392)
// over the MI355X engine, for ANY graph: the `graph! { ... }` body text goes to the engine's DSL front end.
//
//   let mut g = GpuGraph::<0>::from_dsl(include_str!("fm_voice.graph"), &["frequency"], [], 65_536)?;   // Graph::new()
//   (the const parameter is the number of STREAM INPUTS of the graph -- BlockRender::NUM_STREAM_INPUTS, a compile-time
//    constant in the reference too -- and the array names them in declaration order; a poly wrapper such as
//    examples/fm-synth/src/lib.rs `FMGraph` goes in as written: GpuGraph::<0>::from_dsl(FM_GRAPH_BODY, &[], [], 65_536))
//   g.init(48_000.0);                                                                          // init(sr)
//   let cutoff = g.value("filter_cutoff")?;               // input handles are resolved ONCE (no per-call lookup)
//   g.set(cutoff, 3_000.0); g.set_with_ramp(cutoff, 6_000.0, 2_205); g.set_immediate(cutoff, 1_000.0);
//   g.try_push(gate, voice, EventInstance { frame_offset: 17, payload: 0.8 })?;
//   g.process_block(256);  let bus = &g.out_block[..256 * g.channels()];
//   let rendered: Vec<Vec<f32>> = BlockRender::render(&mut g1, &[&input[..]], tail);  // g1: GpuGraph<1>, offline.rs:46-90
//
// `gpu_graph!` below generates a struct with the reference's own method names (set_<name>, set_<name>_with_ramp,
// set_<name>_immediate) for a fixed list of inputs, so existing call sites compile unchanged.
use oscen_gpu_sys as sys;
use std::ffi::{CStr, CString};

pub const MAX_BLOCK_SIZE: usize = 512; // oscen-lib/src/graph/types.rs:12

#[derive(Debug)]
pub struct GpuError(pub i32, pub String);
fn ck(rc: i32) -> Result<i32, GpuError> {
    if rc >= 0 { Ok(rc) } else {
        let msg = unsafe { CStr::from_ptr(sys::og_last_error()) }.to_string_lossy().into_owned();
        Err(GpuError(rc, msg))
    }
}

/// index of a graph input, resolved once by `value()` / `event()` / `stream()`
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct InputId(pub u32);

/// EventInstance { frame_offset, payload: Scalar(f32) }  (oscen-lib/src/graph/types.rs:22,87-90)
#[derive(Clone, Copy, Debug)]
pub struct EventInstance { pub frame_offset: u32, pub payload: f32 }

/// `IN` = the graph's number of stream inputs (`BlockRender::NUM_STREAM_INPUTS`, offline.rs:23): a compile-time
/// constant of a generated graph, so a const parameter here; `from_dsl` / `builtin` check it against the engine.
pub struct GpuGraph<const IN: usize = 0> {
    e: *mut sys::og_engine,
    channels: usize,
    stream_inputs: [InputId; IN],
    pub stream_in_blocks: [[f32; MAX_BLOCK_SIZE]; IN],     // `pub <stream_in>_block: [f32; 512]`
    /// `<out>_block`: interleaved when the graph ends in a Frame<2> post-mix node
    pub out_block: [f32; 2 * MAX_BLOCK_SIZE],
    /// `pub <out>: F`: the last frame of the last block
    pub out: [f32; 2],
}
unsafe impl<const IN: usize> Send for GpuGraph<IN> {} // SignalProcessor: Send (traits.rs:27); one `&mut self` caller at a time

impl<const IN: usize> GpuGraph<IN> {
    /// Graph::new(): `voices = [Voice::new(); N]` with N = n_voices, 44.1 kHz until init().  `per_voice` names the value
    /// inputs the poly wrapper feeds per voice (MidiVoiceHandler.frequency; empty for a poly-wrapper body, which names
    /// them itself); `stream_inputs` names the graph's stream inputs in declaration order.  Fails loudly without a GPU.
    pub fn from_dsl(graph_body: &str, per_voice: &[&str], stream_inputs: [&str; IN], n_voices: u32) -> Result<Self, GpuError> {
        let text = CString::new(graph_body).unwrap();
        let pv = CString::new(per_voice.join(",")).unwrap();
        let mut g = std::ptr::null_mut();
        ck(unsafe { sys::og_graph_parse(text.as_ptr(), pv.as_ptr(), &mut g) })?;
        Self::from_desc(g, stream_inputs, n_voices)
    }
    pub fn builtin(name: &str, stream_inputs: [&str; IN], n_voices: u32) -> Result<Self, GpuError> {
        let n = CString::new(name).unwrap();
        let mut g = std::ptr::null_mut();
        ck(unsafe { sys::og_graph_builtin(n.as_ptr(), &mut g) })?;
        Self::from_desc(g, stream_inputs, n_voices)
    }
    fn from_desc(g: *mut sys::og_graph_desc, stream_inputs: [&str; IN], n_voices: u32) -> Result<Self, GpuError> {
        let mut e = std::ptr::null_mut();
        let rc = unsafe { sys::og_create(g, n_voices, 0, &mut e) };
        unsafe { sys::og_graph_free(g) };
        ck(rc)?;
        let channels = unsafe { sys::og_channels(e) } as usize;
        let mut me = Self { e, channels, stream_inputs: [InputId(0); IN], stream_in_blocks: [[0.0; MAX_BLOCK_SIZE]; IN],
                            out_block: [0.0; 2 * MAX_BLOCK_SIZE], out: [0.0; 2] };
        // NUM_STREAM_INPUTS is a type-level constant: it must be the engine's count (assert_eq! in the default render)
        let have = unsafe { sys::og_num_stream_inputs(e) } as usize;
        if have != IN {
            return Err(GpuError(-1, format!("graph has {} stream inputs, GpuGraph::<{}> was asked for", have, IN)));
        }
        for k in 0..IN { me.stream_inputs[k] = me.id(stream_inputs[k])?; }
        Ok(me)
    }
    pub fn init(&mut self, sample_rate: f32) { unsafe { sys::og_init(self.e, sample_rate); } }
    pub fn set_sample_rate(&mut self, sample_rate: f32) { self.init(sample_rate) }
    pub fn channels(&self) -> usize { self.channels }
    pub fn latency_samples(&self) -> u32 { unsafe { sys::og_latency_samples(self.e) } }

    fn id(&self, name: &str) -> Result<InputId, GpuError> {
        let c = CString::new(name).unwrap();
        Ok(InputId(ck(unsafe { sys::og_input_index(self.e, c.as_ptr()) })? as u32))
    }
    pub fn value(&self, name: &str) -> Result<InputId, GpuError> { self.id(name) }
    pub fn event(&self, name: &str) -> Result<InputId, GpuError> { self.id(name) }
    /// handle of stream input k (declaration order = BlockRender order)
    pub fn stream(&self, k: usize) -> InputId { self.stream_inputs[k] }

    // generated setters  codegen/mod.rs:917-976
    pub fn set(&mut self, i: InputId, v: f32) { unsafe { sys::og_set_value(self.e, i.0, v); } }
    pub fn set_with_ramp(&mut self, i: InputId, v: f32, frames: u32) { unsafe { sys::og_set_value_ramp(self.e, i.0, v, frames); } }
    pub fn set_immediate(&mut self, i: InputId, v: f32) { unsafe { sys::og_set_value_immediate(self.e, i.0, v); } }
    /// `voice_handlers.frequency -> voices.frequency` for one voice, effective at the next block
    pub fn set_voice(&mut self, i: InputId, voice: u32, v: f32) { unsafe { sys::og_set_voice_value(self.e, i.0, voice, v); } }
    /// `<event_input>.try_push(ev)` for one voice: Err on overflow (33rd event of the block), like ArrayVec::try_push
    pub fn try_push(&mut self, i: InputId, voice: u32, ev: EventInstance) -> Result<(), GpuError> {
        ck(unsafe { sys::og_push_voice_event(self.e, i.0, voice, ev.frame_offset, ev.payload) }).map(|_| ())
    }
    /// MidiVoiceHandler::on_note_on (midi.rs:91-105): frequency and gate change on the same frame
    pub fn note_on(&mut self, frequency: InputId, gate: InputId, voice: u32, note: u8, velocity: f32, frame_offset: u32) {
        let hz = unsafe { sys::og_midi_note_to_freq(note) };
        unsafe {
            let _ = sys::og_push_voice_value(self.e, frequency.0, voice, frame_offset, hz);
            let _ = sys::og_push_voice_event(self.e, gate.0, voice, frame_offset, velocity); // `let _ = try_push(..)` as in the reference
        }
    }

    /// process_block(frames): reads `stream_in_blocks`, fills out_block[..frames * channels]
    pub fn process_block(&mut self, frames: usize) {
        debug_assert!(frames <= MAX_BLOCK_SIZE);
        for (k, id) in self.stream_inputs.iter().enumerate() {
            unsafe { sys::og_set_stream_block(self.e, id.0, self.stream_in_blocks[k].as_ptr(), frames as u32); }
        }
        unsafe { sys::og_process_block(self.e, frames as u32, self.out_block.as_mut_ptr()); }
        if frames > 0 {
            for c in 0..self.channels { self.out[c] = self.out_block[(frames - 1) * self.channels + c]; }
        }
    }
    pub fn process(&mut self) { self.process_block(1) }
    pub fn get_stream_output(&self, i: usize) -> Option<f32> { if i < self.channels { Some(self.out[i]) } else { None } }
}
impl<const IN: usize> Drop for GpuGraph<IN> { fn drop(&mut self) { unsafe { sys::og_destroy(self.e) } } }

/// oscen::graph::offline::BlockRender<f32> (offline.rs:19-113) for a graph with a mono bus.  NUM_STREAM_INPUTS is the
/// const parameter, so the trait's default `render` (`assert_eq!(inputs.len(), Self::NUM_STREAM_INPUTS)`, the loop
/// `for i in 0..Self::NUM_STREAM_INPUTS`, offline.rs:46-75) and `render_mono` (`assert_eq!(.., 1)`, :96-101) work
/// as they do for a generated graph: chunks of 512, silence padding, `tail`.
impl<const IN: usize> oscen::BlockRender<f32> for GpuGraph<IN> {
    const NUM_STREAM_INPUTS: usize = IN;
    const NUM_STREAM_OUTPUTS: usize = 1;
    fn run_block(&mut self, frames: usize) { debug_assert!(self.channels == 1); self.process_block(frames) }
    fn stream_input_block_mut(&mut self, index: usize) -> &mut [f32] { &mut self.stream_in_blocks[index][..] }
    fn stream_output_block(&self, _index: usize) -> &[f32] { &self.out_block[..MAX_BLOCK_SIZE] }
}
impl<const IN: usize> GpuGraph<IN> {
    pub fn num_stream_inputs(&self) -> usize { unsafe { sys::og_num_stream_inputs(self.e) as usize } }
    /// render(inputs, tail) in one call into the library (the device keeps the whole output until the end); interleaved
    /// when the bus is a Frame<2>
    pub fn render_all(&mut self, inputs: &[&[f32]; IN], tail: usize) -> Vec<f32> {
        let ptrs: Vec<*const f32> = inputs.iter().map(|s| s.as_ptr()).collect();
        let lens: Vec<u64> = inputs.iter().map(|s| s.len() as u64).collect();
        let total = lens.iter().copied().max().unwrap_or(0) as usize + tail;
        let mut out = vec![0.0f32; total * self.channels];
        let mut got = 0u64;
        unsafe { sys::og_render_inputs(self.e, ptrs.as_ptr(), lens.as_ptr(), inputs.len() as u32, tail as u64,
                                       out.as_mut_ptr(), &mut got); }
        out
    }
}

/// A pure function applied on a connection (`half(a.output) -> out`, `dsp::decode_ms(s.output) -> out`): what the
/// reference resolves to a Rust function in scope is registered here once, its body as device source.
/// `args`: (parameter name, channels) with channels 1 = f32, N = Frame<N> (`og::Frame<N>`, `.v[i]`).
pub fn register_function(name: &str, args: &[(&str, u32)], result_channels: u32, body: &str) -> Result<(), GpuError> {
    let c_name = CString::new(name).unwrap();
    let c_body = CString::new(body).unwrap();
    let names: Vec<CString> = args.iter().map(|a| CString::new(a.0).unwrap()).collect();
    let name_ptrs: Vec<*const std::os::raw::c_char> = names.iter().map(|n| n.as_ptr()).collect();
    let widths: Vec<u32> = args.iter().map(|a| a.1).collect();
    let f = sys::og_function_type {
        name: c_name.as_ptr(), n_args: args.len() as u32, arg_names: name_ptrs.as_ptr(), arg_channels: widths.as_ptr(),
        result_channels, source: c_body.as_ptr(),
    };
    ck(unsafe { sys::og_register_function(&f) }).map(|_| ())
}

impl<const IN: usize> GpuGraph<IN> {
    /// `graph.<node>.<field>` of every voice in `first_voice .. first_voice + n` (the generated struct's node fields are
    /// public in the reference; here a node's persistent fields are planes of the state image): "node.field",
    /// "array[i].field", "nested.node.field"
    pub fn read_state_field(&mut self, path: &str, first_voice: u32, n: u32) -> Result<Vec<f32>, GpuError> {
        let c_path = CString::new(path).unwrap();
        let mut out = vec![0.0f32; n as usize];
        ck(unsafe { sys::og_read_state_field(self.e, c_path.as_ptr(), first_voice, n, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
    /// the events the voices pushed into the graph's event outputs since the last call, ordered by (frame, voice, push
    /// order) -- what iterating `graph.<event_output>` gives after process_block -- and how many the log could not hold
    pub fn read_output_events(&mut self, cap: usize) -> Result<(Vec<sys::og_out_event>, u64), GpuError> {
        let mut buf = vec![sys::og_out_event::default(); cap];
        let (mut n, mut over) = (0u32, 0u64);
        ck(unsafe { sys::og_read_output_events(self.e, buf.as_mut_ptr(), cap as u32, &mut n, &mut over) })?;
        buf.truncate(n as usize);
        Ok((buf, over))
    }
}

/// A struct with the generated graph's own method names for a fixed input list:
///   gpu_graph! { FMGraphGpu, dsl = include_str!("fm_voice.graph"), per_voice = [frequency],
///                values = [op3_ratio, op3_level, filter_cutoff], events = [gate] }
///   g.set_filter_cutoff(3000.0); g.set_filter_cutoff_with_ramp(6000.0, 2205); g.gate_try_push(voice, ev)?;
#[macro_export]
macro_rules! gpu_graph {
    ($name:ident, dsl = $dsl:expr, per_voice = [$($pv:ident),*], values = [$($v:ident),*], events = [$($ev:ident),*]) => {
        paste::paste! {
            pub struct $name { pub g: $crate::GpuGraph<0>, $($v: $crate::InputId,)* $($ev: $crate::InputId,)* }
            impl $name {
                pub fn new(n_voices: u32) -> Result<Self, $crate::GpuError> {
                    let g = $crate::GpuGraph::<0>::from_dsl($dsl, &[$(stringify!($pv)),*], [], n_voices)?;
                    Ok(Self { $($v: g.value(stringify!($v))?,)* $($ev: g.event(stringify!($ev))?,)* g })
                }
                pub fn init(&mut self, sr: f32) { self.g.init(sr) }
                pub fn process_block(&mut self, frames: usize) { self.g.process_block(frames) }
                $(
                    pub fn [<set_ $v>](&mut self, v: f32) { self.g.set(self.$v, v) }
                    pub fn [<set_ $v _with_ramp>](&mut self, v: f32, frames: u32) { self.g.set_with_ramp(self.$v, v, frames) }
                    pub fn [<set_ $v _immediate>](&mut self, v: f32) { self.g.set_immediate(self.$v, v) }
                )*
                $(
                    pub fn [<$ev _try_push>](&mut self, voice: u32, ev: $crate::EventInstance) -> Result<(), $crate::GpuError> {
                        self.g.try_push(self.$ev, voice, ev)
                    }
                )*
            }
        }
    };
}
