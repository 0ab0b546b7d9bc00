// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no rustc/cargo): shipped as source for the
// maintainer of the reference.  Kept in sync with INTEGRATION.md (tests/test_capi_cpu.py checks the sys crate).
//
// oscen-gpu: the public surface of a `graph!`-generated struct (oscen-graph-compiler/src/codegen/mod.rs:1292-1392)
// over the MI355X engine, for ANY graph: the `graph! { ... }` body text goes to the engine's DSL front end.
//
//   let mut g = GpuGraph::<0>::from_dsl(include_str!("fm_voice.graph"), &["frequency"], [], 65_536, 0)?;   // Graph::new() on device 0
//   (the const parameter is the number of STREAM INPUTS of the graph -- BlockRender::NUM_STREAM_INPUTS, a compile-time
//    constant in the reference too -- and the array names them in declaration order; a poly wrapper such as
//    examples/fm-synth/src/lib.rs `FMGraph` goes in as written: GpuGraph::<0>::from_dsl(FM_GRAPH_BODY, &[], [], 65_536, 0))
//   g.init(48_000.0)?;                                                                         // init(sr)
//   let cutoff = g.value("filter_cutoff")?;               // input handles are resolved ONCE (no per-call lookup)
//   g.set(cutoff, 3_000.0)?; g.set_with_ramp(cutoff, 6_000.0, 2_205)?; g.set_immediate(cutoff, 1_000.0)?;
//   g.try_push(gate, voice, EventInstance { frame_offset: 17, payload: 0.8 })?;
//   g.process_block(256)?;  let bus = &g.out_block[..256 * g.channels()];
//   let rendered: Vec<Vec<f32>> = BlockRender::render(&mut g1, &[&input[..]], tail);  // g1: GpuGraph<1>, offline.rs:46-90
//
// Every call that reaches the library returns Result: the generated Rust methods cannot fail, the device can (round 3
// discarded the codes).  Buffers are sized for the engine's limits -- up to MAX_BUS_CHANNELS = 4 interleaved bus
// channels, Frame<N> stream inputs of up to 4 channels -- and tests/capi/shim_sequence.c performs this file's call
// sequence in C with guard words around exactly these buffers (tests/test_capi_c.py).
//
// `gpu_graph!` below generates a struct with the reference's own method names (set_<name>, set_<name>_with_ramp,
// set_<name>_immediate) for a fixed list of inputs, so existing call sites compile unchanged.
use oscen_gpu_sys as sys;
use std::ffi::{CStr, CString};

pub const MAX_BLOCK_SIZE: usize = 512; // oscen-lib/src/graph/types.rs:12
pub const MAX_BUS_CHANNELS: usize = 4; // OG_MAX_BUS_CHANNELS (og_kernel_rt.hip.h): stream outputs / Frame<N> channels of the bus

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
    /// channels of every stream input (og_stream_input_channels: 1 = f32, N = Frame<N>)
    stream_in_channels: [usize; IN],
    /// `pub <stream_in>_block: [F; 512]`: frames x channels interleaved samples, sized for a Frame<4>
    pub stream_in_blocks: [[f32; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE]; IN],
    /// `<out>_block`: frames x channels() interleaved samples (several stream outputs / a Frame<N> bus: up to 4 channels)
    pub out_block: [f32; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE],
    /// `pub <out>: F`: the last frame of the last block
    pub out: [f32; MAX_BUS_CHANNELS],
}
unsafe impl<const IN: usize> Send for GpuGraph<IN> {} // SignalProcessor: Send (traits.rs:27); one `&mut self` caller at a time

impl<const IN: usize> GpuGraph<IN> {
    /// Graph::new(): `voices = [Voice::new(); N]` with N = n_voices, 44.1 kHz until init().  `per_voice` names the value
    /// inputs the poly wrapper feeds per voice (MidiVoiceHandler.frequency; empty for a poly-wrapper body, which names
    /// them itself); `stream_inputs` names the graph's stream inputs in declaration order.  Fails loudly without a GPU.
    pub fn from_dsl(graph_body: &str, per_voice: &[&str], stream_inputs: [&str; IN], n_voices: u32, device_id: i32) -> Result<Self, GpuError> {
        let text = CString::new(graph_body).map_err(|_| GpuError(-1, "graph text holds a NUL byte".into()))?;
        let pv = CString::new(per_voice.join(",")).map_err(|_| GpuError(-1, "input name holds a NUL byte".into()))?;
        let mut g = std::ptr::null_mut();
        ck(unsafe { sys::og_graph_parse(text.as_ptr(), pv.as_ptr(), &mut g) })?;
        Self::from_desc(g, stream_inputs, n_voices, device_id)
    }
    pub fn builtin(name: &str, stream_inputs: [&str; IN], n_voices: u32, device_id: i32) -> Result<Self, GpuError> {
        let n = CString::new(name).map_err(|_| GpuError(-1, "graph name holds a NUL byte".into()))?;
        let mut g = std::ptr::null_mut();
        ck(unsafe { sys::og_graph_builtin(n.as_ptr(), &mut g) })?;
        Self::from_desc(g, stream_inputs, n_voices, device_id)
    }
    fn from_desc(g: *mut sys::og_graph_desc, stream_inputs: [&str; IN], n_voices: u32, device_id: i32) -> Result<Self, GpuError> {
        let mut e = std::ptr::null_mut();
        let rc = unsafe { sys::og_create(g, n_voices, device_id, &mut e) };
        unsafe { sys::og_graph_free(g) };
        ck(rc)?;
        let channels = unsafe { sys::og_channels(e) } as usize;
        let mut me = Self { e, channels, stream_inputs: [InputId(0); IN], stream_in_channels: [1; IN],
                            stream_in_blocks: [[0.0; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE]; IN],
                            out_block: [0.0; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE], out: [0.0; MAX_BUS_CHANNELS] };
        // (from here on `me` owns the engine: an early return drops it)
        if channels == 0 || channels > MAX_BUS_CHANNELS {
            return Err(GpuError(-2, format!("the bus has {} channels, this shim holds up to {}", channels, MAX_BUS_CHANNELS)));
        }
        // NUM_STREAM_INPUTS is a type-level constant: it must be the engine's count (assert_eq! in the default render)
        let have = unsafe { sys::og_num_stream_inputs(e) } as usize;
        if have != IN {
            return Err(GpuError(-1, format!("graph has {} stream inputs, GpuGraph::<{}> was asked for", have, IN)));
        }
        for k in 0..IN {
            me.stream_inputs[k] = me.id(stream_inputs[k])?;
            let ch = unsafe { sys::og_stream_input_channels(me.e, me.stream_inputs[k].0) } as usize;
            if ch == 0 || ch > MAX_BUS_CHANNELS {
                return Err(GpuError(-1, format!("'{}' is not a stream input of 1..{} channels", stream_inputs[k], MAX_BUS_CHANNELS)));
            }
            me.stream_in_channels[k] = ch;
        }
        Ok(me)
    }
    pub fn init(&mut self, sample_rate: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_init(self.e, sample_rate) }).map(|_| ()) }
    pub fn set_sample_rate(&mut self, sample_rate: f32) -> Result<(), GpuError> { self.init(sample_rate) }
    /// channels of stream input k: stream_in_blocks[k][..frames * stream_in_channels(k)] is what process_block reads
    pub fn stream_in_channels(&self, k: usize) -> usize { self.stream_in_channels[k] }
    pub fn channels(&self) -> usize { self.channels }
    pub fn latency_samples(&self) -> u32 { unsafe { sys::og_latency_samples(self.e) } }

    fn id(&self, name: &str) -> Result<InputId, GpuError> {
        let c = CString::new(name).map_err(|_| GpuError(-1, "input name holds a NUL byte".into()))?;
        Ok(InputId(ck(unsafe { sys::og_input_index(self.e, c.as_ptr()) })? as u32))
    }
    pub fn value(&self, name: &str) -> Result<InputId, GpuError> { self.id(name) }
    pub fn event(&self, name: &str) -> Result<InputId, GpuError> { self.id(name) }
    /// handle of stream input k (declaration order = BlockRender order)
    pub fn stream(&self, k: usize) -> InputId { self.stream_inputs[k] }

    // generated setters  codegen/mod.rs:917-976
    pub fn set(&mut self, i: InputId, v: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_set_value(self.e, i.0, v) }).map(|_| ()) }
    pub fn set_with_ramp(&mut self, i: InputId, v: f32, frames: u32) -> Result<(), GpuError> { ck(unsafe { sys::og_set_value_ramp(self.e, i.0, v, frames) }).map(|_| ()) }
    pub fn set_immediate(&mut self, i: InputId, v: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_set_value_immediate(self.e, i.0, v) }).map(|_| ()) }
    /// `voice_handlers.frequency -> voices.frequency` for one voice, effective at the next block
    pub fn set_voice(&mut self, i: InputId, voice: u32, v: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_set_voice_value(self.e, i.0, voice, v) }).map(|_| ()) }
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

    /// process_block(frames): reads stream_in_blocks[k][..frames * channels of input k], fills out_block[..frames * channels].
    /// The library reads `frames x N` floats of a Frame<N> input and writes `frames x channels()` floats: both fit the
    /// arrays above for every frames <= 512 (checked here, not assumed).
    pub fn process_block(&mut self, frames: usize) -> Result<(), GpuError> {
        if frames > MAX_BLOCK_SIZE { return Err(GpuError(-1, format!("process_block({}): at most {} frames", frames, MAX_BLOCK_SIZE))); }
        for (k, id) in self.stream_inputs.iter().enumerate() {
            debug_assert!(frames * self.stream_in_channels[k] <= self.stream_in_blocks[k].len());
            ck(unsafe { sys::og_set_stream_block(self.e, id.0, self.stream_in_blocks[k].as_ptr(), frames as u32) })?;
        }
        debug_assert!(frames * self.channels <= self.out_block.len());
        ck(unsafe { sys::og_process_block(self.e, frames as u32, self.out_block.as_mut_ptr()) })?;
        if frames > 0 {
            for c in 0..self.channels { self.out[c] = self.out_block[(frames - 1) * self.channels + c]; }
        }
        Ok(())
    }
    pub fn process(&mut self) -> Result<(), GpuError> { self.process_block(1) }
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
    // (the trait's methods cannot fail: a device error here is a panic, like an audio callback that lost its device)
    fn run_block(&mut self, frames: usize) { assert!(self.channels == 1); self.process_block(frames).expect("oscen-gpu: process_block") }
    fn stream_input_block_mut(&mut self, index: usize) -> &mut [f32] {
        assert!(self.stream_in_channels[index] == 1, "BlockRender<f32>: stream input {} is a Frame<{}>", index, self.stream_in_channels[index]);
        &mut self.stream_in_blocks[index][..MAX_BLOCK_SIZE]
    }
    fn stream_output_block(&self, _index: usize) -> &[f32] { &self.out_block[..MAX_BLOCK_SIZE] }
}
impl<const IN: usize> GpuGraph<IN> {
    pub fn num_stream_inputs(&self) -> usize { unsafe { sys::og_num_stream_inputs(self.e) as usize } }
    /// render(inputs, tail) in one call into the library (the device keeps the whole output until the end); interleaved
    /// when the bus is a Frame<2>
    /// `inputs[k]` holds interleaved frames of stream input k (len = frames x stream_in_channels(k))
    pub fn render_all(&mut self, inputs: &[&[f32]; IN], tail: usize) -> Result<Vec<f32>, GpuError> {
        let ptrs: Vec<*const f32> = inputs.iter().map(|s| s.as_ptr()).collect();
        // og_render_inputs takes lengths in FRAMES: a Frame<N> input's slice holds N floats per frame
        let mut lens: Vec<u64> = Vec::with_capacity(IN);
        for (k, s) in inputs.iter().enumerate() {
            if s.len() % self.stream_in_channels[k] != 0 {
                return Err(GpuError(-1, format!("input {}: {} floats is not a whole number of Frame<{}>", k, s.len(), self.stream_in_channels[k])));
            }
            lens.push((s.len() / self.stream_in_channels[k]) as u64);
        }
        let total = lens.iter().copied().max().unwrap_or(0) as usize + tail;
        let mut out = vec![0.0f32; total * self.channels];
        let mut got = 0u64;
        ck(unsafe { sys::og_render_inputs(self.e, ptrs.as_ptr(), lens.as_ptr(), inputs.len() as u32, tail as u64,
                                          out.as_mut_ptr(), &mut got) })?;
        out.truncate(got as usize * self.channels);
        Ok(out)
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
    /// `og_group_voices`: after the score has been scheduled and before the first block, order the voice SLOTS so that voices
    /// whose notes end together share waves (policy 1; 0 = identity).  Voice numbers -- the `voices[i]` of the generated
    /// struct -- do not change anywhere on this surface.
    pub fn group_voices(&mut self, policy: u32) -> Result<(), GpuError> { ck(unsafe { sys::og_group_voices(self.e, policy) }).map(|_| ()) }
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

/// A voice bank sharded over the GPUs of one node (og_cluster_*): contiguous global voice ranges, one engine per entry of
/// `device_ids`, the per-shard buses summed onto device_ids[0] with ONE RCCL reduce per batch of blocks -- the
/// reference's `voices.out -> out` sum (oscen-graph-compiler/src/codegen/emit_node.rs:463-466) over 8 x the voices.
pub struct GpuCluster {
    c: *mut sys::og_cluster,
    channels: usize,
    /// frames x channels() interleaved samples of the last process_block
    pub out_block: [f32; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE],
}
unsafe impl Send for GpuCluster {}
impl GpuCluster {
    pub fn from_dsl(graph_body: &str, per_voice: &[&str], n_voices_total: u64, device_ids: &[i32]) -> Result<Self, GpuError> {
        let text = CString::new(graph_body).map_err(|_| GpuError(-1, "graph text holds a NUL byte".into()))?;
        let pv = CString::new(per_voice.join(",")).map_err(|_| GpuError(-1, "input name holds a NUL byte".into()))?;
        let mut g = std::ptr::null_mut();
        ck(unsafe { sys::og_graph_parse(text.as_ptr(), pv.as_ptr(), &mut g) })?;
        let mut c = std::ptr::null_mut();
        let rc = unsafe { sys::og_cluster_create(g, n_voices_total, device_ids.as_ptr(), device_ids.len() as u32, &mut c) };
        unsafe { sys::og_graph_free(g) };
        ck(rc)?;
        let me = Self { c, channels: unsafe { sys::og_cluster_channels(c) } as usize, out_block: [0.0; MAX_BUS_CHANNELS * MAX_BLOCK_SIZE] };
        if me.channels == 0 || me.channels > MAX_BUS_CHANNELS {
            return Err(GpuError(-2, format!("the bus has {} channels, this shim holds up to {}", me.channels, MAX_BUS_CHANNELS)));
        }
        Ok(me)
    }
    pub fn init(&mut self, sample_rate: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_cluster_init(self.c, sample_rate) }).map(|_| ()) }
    pub fn channels(&self) -> usize { self.channels }
    pub fn num_voices(&self) -> u64 { unsafe { sys::og_cluster_num_voices(self.c) } }
    pub fn num_devices(&self) -> u32 { unsafe { sys::og_cluster_num_devices(self.c) } }
    /// RCCL reduces issued so far (0 on a one-device cluster: the shards are added on the device)
    pub fn rccl_reduces(&self) -> u64 { unsafe { sys::og_cluster_rccl_reduces(self.c) } }
    pub fn group_voices(&mut self, policy: u32) -> Result<(), GpuError> { ck(unsafe { sys::og_cluster_group_voices(self.c, policy) }).map(|_| ()) }
    pub fn input(&self, name: &str) -> Result<InputId, GpuError> {
        let n = CString::new(name).map_err(|_| GpuError(-1, "input name holds a NUL byte".into()))?;
        Ok(InputId(ck(unsafe { sys::og_cluster_input_index(self.c, n.as_ptr()) })? as u32))
    }
    pub fn set(&mut self, i: InputId, v: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_cluster_set_value(self.c, i.0, v) }).map(|_| ()) }
    pub fn set_with_ramp(&mut self, i: InputId, v: f32, frames: u32) -> Result<(), GpuError> { ck(unsafe { sys::og_cluster_set_value_ramp(self.c, i.0, v, frames) }).map(|_| ()) }
    pub fn set_immediate(&mut self, i: InputId, v: f32) -> Result<(), GpuError> { ck(unsafe { sys::og_cluster_set_value_immediate(self.c, i.0, v) }).map(|_| ()) }
    /// per-voice values of the GLOBAL voices first_voice .. first_voice + v.len(), routed to the shards that own them
    pub fn set_voices(&mut self, i: InputId, first_voice: u64, v: &[f32]) -> Result<(), GpuError> {
        ck(unsafe { sys::og_cluster_set_voice_values(self.c, i.0, first_voice, v.len() as u64, v.as_ptr()) }).map(|_| ())
    }
    pub fn try_push(&mut self, i: InputId, voice: u64, ev: EventInstance) -> Result<(), GpuError> {
        ck(unsafe { sys::og_cluster_push_voice_event(self.c, i.0, voice, ev.frame_offset, ev.payload) }).map(|_| ())
    }
    /// the real-time entry: every shard renders the block, the buses are summed, out_block holds frames x channels()
    pub fn process_block(&mut self, frames: usize) -> Result<(), GpuError> {
        if frames > MAX_BLOCK_SIZE { return Err(GpuError(-1, format!("process_block({}): at most {} frames", frames, MAX_BLOCK_SIZE))); }
        ck(unsafe { sys::og_cluster_process_block(self.c, frames as u32, self.out_block.as_mut_ptr()) }).map(|_| ())
    }
    /// offline: total_frames in blocks of `block`, batched launches, one reduce per batch
    pub fn render(&mut self, total_frames: u64, block: u32) -> Result<Vec<f32>, GpuError> {
        let mut out = vec![0.0f32; total_frames as usize * self.channels];
        ck(unsafe { sys::og_cluster_render(self.c, total_frames, block, out.as_mut_ptr()) })?;
        Ok(out)
    }
}
impl Drop for GpuCluster { fn drop(&mut self) { unsafe { sys::og_cluster_destroy(self.c) } } }

/// MidiParser -> VoiceAllocator -> [MidiVoiceHandler; N] of the reference's poly wrappers (oscen-lib/src/midi.rs:40-122,
/// voice_allocator.rs:57-136) as the host-side front end of a bank: raw MIDI in, per-voice frequency / gate events out,
/// on the exact frame.  Borrows the graph (or cluster) it drives for its whole life.
pub struct GpuMidi<'a> {
    m: *mut sys::og_midi,
    _bank: std::marker::PhantomData<&'a mut ()>,
}
impl<'a> GpuMidi<'a> {
    pub fn new<const IN: usize>(g: &'a mut GpuGraph<IN>, frequency_input: &str, gate_input: &str) -> Result<Self, GpuError> {
        let (f, ga) = (CString::new(frequency_input).map_err(|_| GpuError(-1, "NUL in name".into()))?,
                       CString::new(gate_input).map_err(|_| GpuError(-1, "NUL in name".into()))?);
        let mut m = std::ptr::null_mut();
        ck(unsafe { sys::og_midi_create(g.e, 0, f.as_ptr(), ga.as_ptr(), &mut m) })?;
        Ok(Self { m, _bank: std::marker::PhantomData })
    }
    pub fn over_cluster(c: &'a mut GpuCluster, frequency_input: &str, gate_input: &str) -> Result<Self, GpuError> {
        let (f, ga) = (CString::new(frequency_input).map_err(|_| GpuError(-1, "NUL in name".into()))?,
                       CString::new(gate_input).map_err(|_| GpuError(-1, "NUL in name".into()))?);
        let mut m = std::ptr::null_mut();
        ck(unsafe { sys::og_midi_create_cluster(c.c, f.as_ptr(), ga.as_ptr(), &mut m) })?;
        Ok(Self { m, _bank: std::marker::PhantomData })
    }
    /// `midi_in.try_push(RawMidiMessage::new(bytes), frame_offset)`: Err(OG_E_OVERFLOW) when the block's queue is full
    pub fn send(&mut self, bytes: &[u8], frame_offset: u32) -> Result<(), GpuError> {
        ck(unsafe { sys::og_midi_send(self.m, bytes.as_ptr(), bytes.len() as u32, frame_offset) }).map(|_| ())
    }
    pub fn set_queue_capacity(&mut self, capacity: u32) -> Result<(), GpuError> { ck(unsafe { sys::og_midi_set_queue_capacity(self.m, capacity) }).map(|_| ()) }
    pub fn dropped(&self) -> u64 { unsafe { sys::og_midi_dropped(self.m) } }
    /// the audio callback: apply the queued messages, render `frames`, bus -> out[..frames * channels]
    pub fn process_block(&mut self, frames: usize, out: &mut [f32], channels: usize) -> Result<(), GpuError> {
        if frames > MAX_BLOCK_SIZE || out.len() < frames * channels {
            return Err(GpuError(-1, format!("process_block({}): `out` holds {} floats, {} needed", frames, out.len(), frames * channels)));
        }
        ck(unsafe { sys::og_midi_process_block(self.m, frames as u32, out.as_mut_ptr()) }).map(|_| ())
    }
}
impl<'a> Drop for GpuMidi<'a> { fn drop(&mut self) { unsafe { sys::og_midi_destroy(self.m) } } }

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
                pub fn new(n_voices: u32, device_id: i32) -> Result<Self, $crate::GpuError> {
                    let g = $crate::GpuGraph::<0>::from_dsl($dsl, &[$(stringify!($pv)),*], [], n_voices, device_id)?;
                    Ok(Self { $($v: g.value(stringify!($v))?,)* $($ev: g.event(stringify!($ev))?,)* g })
                }
                pub fn init(&mut self, sr: f32) -> Result<(), $crate::GpuError> { self.g.init(sr) }
                pub fn process_block(&mut self, frames: usize) -> Result<(), $crate::GpuError> { self.g.process_block(frames) }
                $(
                    pub fn [<set_ $v>](&mut self, v: f32) -> Result<(), $crate::GpuError> { self.g.set(self.$v, v) }
                    pub fn [<set_ $v _with_ramp>](&mut self, v: f32, frames: u32) -> Result<(), $crate::GpuError> { self.g.set_with_ramp(self.$v, v, frames) }
                    pub fn [<set_ $v _immediate>](&mut self, v: f32) -> Result<(), $crate::GpuError> { self.g.set_immediate(self.$v, v) }
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
