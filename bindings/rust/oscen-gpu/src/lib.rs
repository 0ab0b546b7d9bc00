// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no rustc/cargo): shipped as source for the
// maintainer of the reference.  Kept in sync with INTEGRATION.md (tests/test_capi_cpu.py checks it).
// oscen-gpu/src/lib.rs  (not compiled in this repo's image: no rustc)
use oscen_gpu_sys as sys;
use std::ffi::CString;

pub const MAX_BLOCK_SIZE: usize = 512;               // oscen-lib/src/graph/types.rs:12

pub struct FMGraphGpu {
    e: *mut sys::og_engine,
    gate: u32, frequency: u32,
    pub audio_out_block: [f32; MAX_BLOCK_SIZE],       // same field callers read today
    pub audio_out: f32,
}
unsafe impl Send for FMGraphGpu {}                    // SignalProcessor: Send (traits.rs:27)

fn idx(e: *mut sys::og_engine, n: &str) -> u32 {
    let c = CString::new(n).unwrap();
    unsafe { sys::og_input_index(e, c.as_ptr()) as u32 }
}

impl FMGraphGpu {
    /// Graph::new()  (44.1 kHz until init), `voices = [FMVoice::new(); N]` with N lifted to n_voices
    pub fn new(n_voices: u32) -> Self {
        let mut g = std::ptr::null_mut();
        let mut e = std::ptr::null_mut();
        let name = CString::new("fm_voice").unwrap();
        unsafe {
            assert_eq!(sys::og_graph_builtin(name.as_ptr(), &mut g), 0);
            assert_eq!(sys::og_create(g, n_voices, 0, &mut e), 0);   // fails loudly without a GPU
            sys::og_graph_free(g);
        }
        Self { e, gate: idx(e, "gate"), frequency: idx(e, "frequency"),
               audio_out_block: [0.0; MAX_BLOCK_SIZE], audio_out: 0.0 }
    }
    pub fn init(&mut self, sample_rate: f32) { unsafe { sys::og_init(self.e, sample_rate); } }

    // generated setters: set_<name>, set_<name>_with_ramp, set_<name>_immediate
    pub fn set(&mut self, name: &str, v: f32) { unsafe { sys::og_set_value(self.e, idx(self.e, name), v); } }
    pub fn set_with_ramp(&mut self, name: &str, v: f32, frames: u32) {
        unsafe { sys::og_set_value_ramp(self.e, idx(self.e, name), v, frames); } }
    pub fn set_immediate(&mut self, name: &str, v: f32) {
        unsafe { sys::og_set_value_immediate(self.e, idx(self.e, name), v); } }

    /// MidiVoiceHandler::on_note_on (midi.rs:91-105): frequency changes and the gate fires on the
    /// event's frame.  Errors are dropped like the reference's `let _ = try_push(..)`.
    pub fn note_on(&mut self, voice: u32, note: u8, velocity: f32, frame_offset: u32) {
        let hz = 440.0_f32 * 2f32.powf((note as f32 - 69.0) / 12.0);
        unsafe {
            let _ = sys::og_push_voice_value(self.e, self.frequency, voice, frame_offset, hz);
            let _ = sys::og_push_voice_event(self.e, self.gate, voice, frame_offset, velocity);
        }
    }
    pub fn note_off(&mut self, voice: u32, frame_offset: u32) {
        unsafe { let _ = sys::og_push_voice_event(self.e, self.gate, voice, frame_offset, 0.0); }
    }

    /// process_block(frames): fills audio_out_block[..frames]
    pub fn process_block(&mut self, frames: usize) {
        debug_assert!(frames <= MAX_BLOCK_SIZE);
        unsafe { sys::og_process_block(self.e, frames as u32, self.audio_out_block.as_mut_ptr()); }
        if frames > 0 { self.audio_out = self.audio_out_block[frames - 1]; }
    }
    pub fn process(&mut self) { self.process_block(1); }
}
impl Drop for FMGraphGpu { fn drop(&mut self) { unsafe { sys::og_destroy(self.e) } } }
