"""ctypes bindings for the CPU oracle (oracle/liboscen_oracle.so).

TEST INFRASTRUCTURE ONLY: nothing under oscen_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboscen_oracle.so")

OO_MAX_EVENTS = 32
OO_NUM_HARMONICS = 32


class Event(C.Structure):
    _fields_ = [("frame_offset", C.c_uint32), ("scalar", C.c_float), ("is_object", C.c_int32)]


class Queue(C.Structure):
    _fields_ = [("ev", Event * OO_MAX_EVENTS), ("len", C.c_uint32)]


class Ramp(C.Structure):
    _fields_ = [("current", C.c_float), ("target", C.c_float), ("increment", C.c_float),
                ("frames_remaining", C.c_uint32)]


class RampedInput(C.Structure):
    _fields_ = [("r", Ramp), ("default_frames", C.c_uint32)]


class Oscillator(C.Structure):
    _fields_ = [("phase", C.c_float), ("frequency", C.c_float), ("frequency_mod", C.c_float),
                ("amplitude", C.c_float), ("output", C.c_float), ("waveform", C.c_int32),
                ("sample_rate", C.c_float)]


class PolyBlep(C.Structure):
    _fields_ = [("phase", C.c_float), ("phase_mod", C.c_float), ("frequency", C.c_float),
                ("frequency_mod", C.c_float), ("amplitude", C.c_float), ("pulse_width", C.c_float),
                ("output", C.c_float), ("waveform", C.c_int32), ("sample_rate", C.c_float)]


class Tpt(C.Structure):
    _fields_ = [("input", C.c_float * 2), ("cutoff", C.c_float), ("q", C.c_float), ("f_mod", C.c_float),
                ("output", C.c_float * 2), ("current_cutoff", C.c_float), ("current_q", C.c_float),
                ("z", (C.c_float * 2) * 2), ("h", C.c_float), ("g", C.c_float), ("r", C.c_float),
                ("k", C.c_float), ("sample_rate", C.c_float), ("channels", C.c_int32)]


class Adsr(C.Structure):
    _fields_ = [("gate", Queue), ("attack", C.c_float), ("decay", C.c_float), ("sustain", C.c_float),
                ("release", C.c_float), ("output", C.c_float), ("stage", C.c_int32),
                ("attack_samples", C.c_uint32), ("decay_samples", C.c_uint32),
                ("release_samples", C.c_uint32), ("samples_remaining", C.c_uint32),
                ("attack_coeff", C.c_float), ("decay_coeff", C.c_float), ("release_increment", C.c_float),
                ("level", C.c_float), ("target_level", C.c_float), ("sustain_level", C.c_float),
                ("velocity", C.c_float), ("sample_rate", C.c_float)]


class IirLowpass(C.Structure):
    _fields_ = [("input", C.c_float), ("cutoff", C.c_float), ("q", C.c_float), ("output", C.c_float),
                ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("a1", C.c_float), ("a2", C.c_float),
                ("v1", C.c_float), ("v2", C.c_float), ("sample_rate", C.c_float),
                ("frame_counter", C.c_uint32), ("frames_per_update", C.c_uint32)]


class Ring(C.Structure):
    _fields_ = [("buffer", C.POINTER(C.c_float)), ("write_pos", C.c_size_t), ("capacity", C.c_size_t),
                ("mask", C.c_size_t), ("mode", C.c_int32)]


class Delay(C.Structure):
    _fields_ = [("input", C.c_float), ("delay_samples", C.c_float), ("feedback", C.c_float), ("output", C.c_float),
                ("buffer", Ring), ("sample_rate", C.c_float), ("frames_per_update", C.c_size_t),
                ("frame_counter", C.c_size_t)]


class Lp18(C.Structure):
    _fields_ = [("input", C.c_float), ("cutoff", C.c_float), ("fmod", C.c_float), ("resonance", C.c_float),
                ("output", C.c_float), ("z", C.c_float * 3), ("g", C.c_float), ("h", C.c_float),
                ("last_cutoff", C.c_float), ("last_fmod", C.c_float), ("last_resonance", C.c_float),
                ("sample_rate", C.c_float)]


class Gain(C.Structure):
    _fields_ = [("input", C.c_float), ("gain", C.c_float), ("output", C.c_float)]


class FmOperator(C.Structure):
    _fields_ = [("phase", C.c_float), ("prev_output", C.c_float), ("sample_rate", C.c_float),
                ("base_freq", C.c_float), ("ratio", C.c_float), ("phase_mod", C.c_float),
                ("feedback", C.c_float), ("envelope", C.c_float), ("level", C.c_float),
                ("output", C.c_float)]


class HbDownStage(C.Structure):
    _fields_ = [("history", C.c_float * 24), ("head", C.c_uint32)]


class HbUpStage(C.Structure):
    _fields_ = [("history", C.c_float * 12), ("head", C.c_uint32)]


class SincDown(C.Structure):
    _fields_ = [("st", HbDownStage * 3), ("n_stages", C.c_uint32), ("factor", C.c_uint32)]


class SincUp(C.Structure):
    _fields_ = [("st", HbUpStage * 3), ("n_stages", C.c_uint32), ("factor", C.c_uint32)]


class Allpass1(C.Structure):
    _fields_ = [("a", C.c_float), ("x_prev", C.c_float), ("y_prev", C.c_float)]


class IirHb2x(C.Structure):
    _fields_ = [("a", Allpass1 * 2), ("b", Allpass1 * 2), ("prev_odd_in", C.c_float)]


class IirResampler(C.Structure):
    _fields_ = [("st", IirHb2x * 3), ("n_stages", C.c_uint32), ("factor", C.c_uint32)]


class LinearUp(C.Structure):
    _fields_ = [("prev", C.c_float), ("factor", C.c_uint32)]


class Tremolo(C.Structure):
    _fields_ = [("input", C.c_float), ("rate", C.c_float), ("depth", C.c_float), ("output", C.c_float * 2),
                ("phase", C.c_float), ("sample_rate", C.c_float)]


class StaticSimple(C.Structure):
    _fields_ = [("osc", Oscillator), ("filter", Tpt), ("gain", Gain)]


class StaticComplex(C.Structure):
    _fields_ = [("osc1", PolyBlep), ("osc2", PolyBlep), ("osc3", PolyBlep),
                ("mix1", Gain), ("mix2", Gain), ("mix3", Gain), ("mixer", Gain), ("env_amount", Gain),
                ("vca", Gain), ("filter_env", Adsr), ("amp_env", Adsr), ("filter", Tpt)]


class NotePlan(C.Structure):
    _fields_ = [("note", C.c_uint8), ("velocity", C.c_uint8), ("on_frame", C.c_uint32),
                ("off_frame", C.c_uint32), ("retrig_frame", C.c_uint32), ("frequency", C.c_float)]


class NoteEvents(C.Structure):
    _fields_ = [("n", C.c_uint32), ("frame", C.c_uint32 * 4), ("value", C.c_float * 4), ("frequency", C.c_float)]


BANK_FM, BANK_SUB, BANK_EPIANO, BANK_SAT4X, BANK_SAT1X = range(5)
EV_GATE, EV_FREQ = 0, 1
PB_SINE, PB_SAW, PB_SQUARE, PB_TRIANGLE = range(4)
WAVE_SINE, WAVE_SQUARE, WAVE_SAW = range(3)

FM_PARAMS = [
    "op3_ratio", "op3_level", "op3_feedback", "op3_attack", "op3_decay", "op3_sustain", "op3_release",
    "op2_ratio", "op2_level", "op2_feedback", "op2_attack", "op2_decay", "op2_sustain", "op2_release",
    "op1_ratio", "op1_attack", "op1_decay", "op1_sustain", "op1_release",
    "route",
    "filter_cutoff", "filter_resonance", "filter_attack", "filter_decay", "filter_sustain",
    "filter_release", "filter_env_amount",
]
SUB_PARAMS = ["cutoff", "q"]
EPIANO_PARAMS = ["brightness", "velocity_scaling", "decay_rate", "harmonic_decay", "key_scaling",
                 "release_rate", "vibrato_intensity", "vibrato_speed"]


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oscen_oracle.c", "oracle_bench.c", "oscen_oracle.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        build()
    lib = C.CDLL(LIB_PATH)
    f32p = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    lib.oo_bank_create.restype = C.c_void_p
    lib.oo_bank_create.argtypes = [C.c_int, C.c_uint32]
    lib.oo_bank_destroy.argtypes = [C.c_void_p]
    lib.oo_bank_init.argtypes = [C.c_void_p, C.c_float]
    lib.oo_bank_num_params.argtypes = [C.c_void_p]
    lib.oo_bank_num_params.restype = C.c_uint32
    lib.oo_bank_channels.argtypes = [C.c_void_p]
    lib.oo_bank_channels.restype = C.c_uint32
    lib.oo_bank_set_value.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    lib.oo_bank_set_value_with_ramp.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_uint32]
    lib.oo_bank_set_value_immediate.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    lib.oo_bank_set_voice_frequency.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    lib.oo_bank_push_event.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_float]
    lib.oo_bank_process_block.argtypes = [C.c_void_p, C.c_uint32, f32p, u32p, C.c_uint32, f32p]
    lib.oo_bank_process_per_sample.argtypes = [C.c_void_p, C.c_uint32, f32p, u32p, C.c_uint32, f32p]
    lib.oo_bank_last_bus_f64.argtypes = [C.c_void_p]
    lib.oo_bank_last_bus_f64.restype = C.POINTER(C.c_double)
    lib.oo_bank_bench.restype = C.c_double
    lib.oo_bank_bench.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.POINTER(C.c_double)]
    lib.oo_bank_bench_grouped.restype = C.c_double
    lib.oo_bank_bench_grouped.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    lib.oo_bank_render_mt.restype = C.c_double
    lib.oo_bank_render_mt.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oo_bank_last_abs_f64.argtypes = [C.c_void_p]
    lib.oo_bank_last_abs_f64.restype = C.POINTER(C.c_double)
    lib.oo_note_plan_scaled.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(NotePlan)]
    lib.oo_note_events_for_voice.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(NoteEvents)]
    lib.oo_bench_set_fold.argtypes = [C.c_int]
    lib.oo_bench_scenario_set.argtypes = [C.c_uint32, C.c_float]
    lib.oo_bench_scenario_ramp.argtypes = [C.c_uint32, C.c_float, C.c_uint32]
    lib.oo_tremolo_new.argtypes = [C.c_void_p]
    lib.oo_tremolo_process.argtypes = [C.c_void_p]
    lib.oo_midi_note_to_freq.restype = C.c_float
    lib.oo_midi_note_to_freq.argtypes = [C.c_uint8]
    lib.oo_midi_velocity_to_gate.restype = C.c_float
    lib.oo_midi_velocity_to_gate.argtypes = [C.c_uint8]
    lib.oo_sinc_down_process.restype = C.c_float
    lib.oo_iir_down_process.restype = C.c_float
    lib.oo_linear_down_process.restype = C.c_float
    lib.oo_linear_down_process.argtypes = [C.c_uint32, f32p]
    lib.oo_latch_down_process.restype = C.c_float
    lib.oo_latch_down_process.argtypes = [C.c_uint32, f32p]
    lib.oo_latch_up_process.argtypes = [C.c_uint32, C.c_float, f32p]
    lib.oo_sinc_down_latency.restype = C.c_uint32
    lib.oo_sinc_up_latency.restype = C.c_uint32
    lib.oo_iir_latency.restype = C.c_uint32
    lib.oo_polyblep_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    lib.oo_oscillator_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    lib.oo_tpt_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    lib.oo_iir_lowpass_new.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.oo_iir_lowpass_process_sample.argtypes = [C.c_void_p, C.c_float]
    lib.oo_iir_lowpass_process_sample.restype = C.c_float
    lib.oo_ring_new.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lib.oo_ring_push.argtypes = [C.c_void_p, C.c_float]
    for fn in (lib.oo_ring_get, lib.oo_ring_get_linear, lib.oo_ring_get_cubic):
        fn.argtypes = [C.c_void_p, C.c_float]
        fn.restype = C.c_float
    lib.oo_delay_new.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.oo_lp18_new.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.oo_adsr_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.oo_ramped_set.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    lib.oo_ramped_set_with_ramp.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]
    lib.oo_ramped_set_immediate.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    lib.oo_ramp_new.argtypes = [C.c_void_p, C.c_float]
    lib.oo_ramp_set_immediate.argtypes = [C.c_void_p, C.c_float]
    lib.oo_ramp_set_with_ramp.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    lib.oo_sinc_up_process.argtypes = [C.c_void_p, C.c_float, f32p]
    lib.oo_iir_up_process.argtypes = [C.c_void_p, C.c_float, f32p]
    lib.oo_linear_up_process.argtypes = [C.c_void_p, C.c_float, f32p]
    lib.oo_static_simple_init.argtypes = [C.c_void_p, C.c_float]
    lib.oo_static_complex_init.argtypes = [C.c_void_p, C.c_float]
    lib.oo_fm_compute_waveform.argtypes = [C.c_float] * 8 + [C.c_uint32, f32p]
    lib.oo_note_plan_for_voice.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(NotePlan)]
    _LIB = lib
    return lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def uptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Bank:
    """Thin OO wrapper over oo_bank_* (the poly-wrapper graph of the reference)."""

    def __init__(self, kind, n_voices, sample_rate=48000.0):
        self.lib = load()
        self.kind = kind
        self.n = n_voices
        self.h = self.lib.oo_bank_create(kind, n_voices)
        self.lib.oo_bank_init(self.h, sample_rate)
        self.channels = self.lib.oo_bank_channels(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oo_bank_destroy(self.h)
            self.h = None

    def set_value(self, p, v):
        assert self.lib.oo_bank_set_value(self.h, p, v) == 0

    def set_value_with_ramp(self, p, v, frames):
        assert self.lib.oo_bank_set_value_with_ramp(self.h, p, v, frames) == 0

    def set_value_immediate(self, p, v):
        assert self.lib.oo_bank_set_value_immediate(self.h, p, v) == 0

    def set_voice_frequency(self, voice, hz):
        self.lib.oo_bank_set_voice_frequency(self.h, voice, hz)

    def push_event(self, voice, frame_offset, kind, value):
        assert self.lib.oo_bank_push_event(self.h, voice, frame_offset, kind, value) == 0

    def process_block(self, frames, taps=None, per_sample=False):
        out = np.zeros(frames * self.channels, dtype=np.float32)
        fn = self.lib.oo_bank_process_per_sample if per_sample else self.lib.oo_bank_process_block
        if taps is None or len(taps) == 0:
            fn(self.h, frames, fptr(out), None, 0, None)
            return out.reshape(frames, self.channels), None
        tv = np.asarray(taps, dtype=np.uint32)
        tb = np.zeros((len(tv), frames), dtype=np.float32)
        fn(self.h, frames, fptr(out), uptr(tv), len(tv), fptr(tb))
        return out.reshape(frames, self.channels), tb

    def last_bus_f64(self, frames):
        p = self.lib.oo_bank_last_bus_f64(self.h)
        return np.array([p[i] for i in range(frames)], dtype=np.float64)


FOLD = {"scale": 0, "slice": 1}


def render_mt(kind, first_voice, n_voices, frames_total, block=256, threads=None, group=8, seed=0x05CE2026, span=0,
              fold="scale", sets=(), ramps=()):
    """Multi-threaded oracle render of a voice range over its (scaled) note plans:
    (mono f64 sum [frames], sum of |voice output| [frames], seconds).
    sets = [(param index, value)]: set_<n>_immediate before frame 0; ramps = [(param index, value, at_frame)]: set_<n>(value)
    right before frame `at_frame` (the render cuts its block there)."""
    lib = load()
    lib.oo_bench_scenario_clear()
    for p, v in sets:
        assert lib.oo_bench_scenario_set(p, v) == 0
    for p, v, f in ramps:
        assert lib.oo_bench_scenario_ramp(p, v, f) == 0
    threads = threads or (os.cpu_count() or 1)
    mono = np.zeros(frames_total, dtype=np.float64)
    ab = np.zeros(frames_total, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    lib.oo_bench_set_fold(FOLD[fold])
    t = lib.oo_bank_render_mt(kind, first_voice, n_voices, frames_total, block, threads, group, seed, span,
                              mono.ctypes.data_as(dp), ab.ctypes.data_as(dp))
    lib.oo_bench_scenario_clear()
    return mono, ab, t


def tremolo_pan(frames, rate, depth, sample_rate=48000.0):
    """[frames, 2] = the Tremolo's (pan, 1 - pan) sequence from a fresh node (input 1.0: 1*pan is exact)."""
    lib = load()
    t = Tremolo()
    lib.oo_tremolo_new(C.byref(t))
    t.sample_rate = sample_rate
    out = np.zeros((frames, 2), dtype=np.float32)
    for f in range(frames):
        t.input, t.rate, t.depth = 1.0, rate, depth
        lib.oo_tremolo_process(C.byref(t))
        out[f] = (t.output[0], t.output[1])
    return out


def note_plan(seed, voice):
    p = NotePlan()
    load().oo_note_plan_for_voice(seed, voice, C.byref(p))
    return p
