"""A SECOND, independent restatement of the unpinned numerics (TEST INFRASTRUCTURE).

The reference cannot be built here (nightly Rust), so FmOperator waveforms, the ADSR curve and the electric-piano voice
have no reference-held vectors: the C oracle (oracle/oscen_oracle.c) is a transliteration.  This module is a second
transliteration, written from the RUST sources (not from the C), in numpy float32 scalar arithmetic -- every `+ - * /` on
np.float32 rounds to f32 exactly like Rust's f32 ops -- with the transcendental functions taken from the platform libm
through ctypes (sinf / cosf / tanf / expf: what Rust's f32::sin etc. bind to on Linux).  tests/test_second_witness.py
demands BIT equality with the C oracle: two hand translations agreeing to the last bit on tens of thousands of samples is
the substitute for an oracle/_ref build.

Sources followed:
  AdsrEnvelope     oscen-lib/src/envelope/adsr.rs:57-306
  FmOperator       examples/fm-synth/src/nodes/fm_operator.rs:34-76
  Crossfade/Mixer/AddValue  examples/fm-synth/src/nodes/{crossfade.rs:37-44,mixer.rs:30-34,add_value.rs:31-35}
  Gain             oscen-lib/src/gain/mod.rs:30-34
  TptFilter<f32>   oscen-lib/src/filters/tpt/mod.rs:47-123
  FMVoice          examples/fm-synth/src/fm_voice.rs:6-156 (schedule: any topological order, the graph is feed-forward)
  AmplitudeSource / OscillatorBank / ElectricPianoVoiceNode  examples/electric-piano/src/electric_piano_voice.rs:50-402
  PolyBlepOscillator, Halfband2x{Up,Down}Stage, Sinc{Up,Down}Fir, the saturator voice   (round 3)
  Oscillator       oscen-lib/src/oscillators/mod.rs:7-77
  IirLowpass       oscen-lib/src/filters/iir_lowpass/mod.rs:15-163
  RingBuffer / Delay  oscen-lib/src/ring_buffer/mod.rs, oscen-lib/src/delay/mod.rs:5-83
  LP18Filter       examples/nih-twin-peaks/src/lp18_filter.rs:7-104
  Tremolo          examples/electric-piano/src/tremolo.rs:8-67
  IirHalfband{Up,Down}, Linear{Up,Down}  oscen-lib/src/resample/{halfband_iir.rs,linear.rs}   (round 3, late)
With these every node the C oracle restates has a second, independent restatement that agrees with it bit for bit.
"""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "cosf", "tanf", "expf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]


def sinf(x):
    return f32(_libm.sinf(float(x)))


def cosf(x):
    return f32(_libm.cosf(float(x)))


def tanf(x):
    return f32(_libm.tanf(float(x)))


def expf(x):
    return f32(_libm.expf(float(x)))


def clamp(x, lo, hi):  # f32::clamp
    x, lo, hi = f32(x), f32(lo), f32(hi)
    if x < lo:
        return lo
    if x > hi:
        return hi
    return x


def as_u32(x):  # Rust `f32 as u32`: saturating, NaN -> 0
    x = float(x)
    if not x > 0.0:
        return 0
    if x >= 4294967296.0:
        return 4294967295
    return int(x)


EPS = f32(1.1920929e-07)
PI = f32(3.14159274101257324)
TAU = f32(6.28318548202514648)

IDLE, ATTACK, DECAY, SUSTAIN, RELEASE = range(5)
MIN_TIME_SECONDS = f32(1.0e-5)
CURVE_TIME_CONSTANT = f32(4.605_170_2)


class Adsr:
    """envelope/adsr.rs"""

    def __init__(self, attack, decay, sustain, release):  # new() :57-82
        self.attack, self.decay, self.sustain, self.release = f32(attack), f32(decay), f32(sustain), f32(release)
        self.output = f32(0)
        self.stage = IDLE
        self.attack_samples = self.decay_samples = self.release_samples = 0
        self.samples_remaining = 0
        self.attack_coeff = self.decay_coeff = self.release_increment = f32(0)
        self.level = self.target_level = f32(0)
        self.sustain_level = clamp(self.sustain, 0.0, 1.0)
        self.velocity = f32(1)
        self.sample_rate = f32(44100.0)
        self.update_sustain_level()

    def apply_parameters(self):  # :84-90
        self.attack = max(self.attack, f32(0))
        self.decay = max(self.decay, f32(0))
        self.sustain = clamp(self.sustain, 0.0, 1.0)
        self.release = max(self.release, f32(0))
        self.update_sustain_level()

    def update_sustain_level(self):  # :92-115
        self.sustain_level = clamp(f32(self.sustain * self.velocity), 0.0, 1.0)
        self.recalculate_cached_steps()
        if self.samples_remaining > 0:
            if self.stage == ATTACK:
                self.samples_remaining = max(min(self.samples_remaining, self.attack_samples), 1)
            elif self.stage == DECAY:
                self.samples_remaining = max(min(self.samples_remaining, self.decay_samples), 1)
            elif self.stage == RELEASE:
                self.samples_remaining = max(min(self.samples_remaining, self.release_samples), 1)
        if self.stage in (DECAY, SUSTAIN):
            self.target_level = self.sustain_level
        elif self.stage == RELEASE:
            self.target_level = f32(0)
        if self.stage == RELEASE:
            self.update_release_increment()

    def recalculate_cached_steps(self):  # :117-134
        sr = max(self.sample_rate, f32(1))
        self.attack_samples = max(as_u32(f32(max(self.attack, MIN_TIME_SECONDS) * sr)), 1)
        self.decay_samples = max(as_u32(f32(max(self.decay, MIN_TIME_SECONDS) * sr)), 1)
        self.release_samples = max(as_u32(f32(max(self.release, MIN_TIME_SECONDS) * sr)), 1)
        self.attack_coeff = f32(f32(1) - expf(f32(-CURVE_TIME_CONSTANT / f32(self.attack_samples))))
        self.decay_coeff = f32(f32(1) - expf(f32(-CURVE_TIME_CONSTANT / f32(self.decay_samples))))

    def set_stage(self, stage, target_level):  # :136-159
        self.stage = stage
        self.target_level = clamp(target_level, 0.0, 1.0)
        samples = {ATTACK: self.attack_samples, DECAY: self.decay_samples, RELEASE: self.release_samples}.get(stage, 0)
        if samples == 0:
            self.samples_remaining = 0
            self.release_increment = f32(0)
            self.level = self.target_level
            if stage not in (SUSTAIN, IDLE):
                self.complete_stage()
        else:
            self.samples_remaining = samples
            self.update_release_increment()

    def update_release_increment(self):  # :161-173
        if self.samples_remaining == 0 or self.stage != RELEASE:
            self.release_increment = f32(0)
            return
        current = clamp(self.level, 0.0, 1.0)
        self.release_increment = f32(0) if current <= 0 else f32(-current / f32(self.samples_remaining))

    def complete_stage(self):  # :175-204
        if self.stage == ATTACK:
            self.level = f32(1)
            self.set_stage(DECAY, self.sustain_level)
        elif self.stage == DECAY:
            self.level = self.sustain_level
            self.stage = SUSTAIN
            self.samples_remaining = 0
            self.release_increment = f32(0)
        elif self.stage == RELEASE:
            self.level = f32(0)
            self.stage = IDLE
            self.samples_remaining = 0
            self.release_increment = f32(0)
        elif self.stage == SUSTAIN:
            self.level = self.sustain_level
            self.samples_remaining = 0
            self.release_increment = f32(0)
        else:
            self.level = f32(0)
            self.samples_remaining = 0
            self.release_increment = f32(0)

    def process_stage(self):  # :206-248
        if self.stage == ATTACK:
            if self.samples_remaining > 0:
                self.level = f32(self.level + f32(f32(f32(1) - self.level) * self.attack_coeff))
                self.samples_remaining -= 1
                self.level = clamp(self.level, 0.0, 1.0)
            if self.samples_remaining == 0:
                self.level = f32(1)
                self.complete_stage()
        elif self.stage == DECAY:
            if self.samples_remaining > 0:
                self.level = f32(self.level + f32(f32(self.sustain_level - self.level) * self.decay_coeff))
                self.samples_remaining -= 1
                self.level = clamp(self.level, 0.0, 1.0)
            if self.samples_remaining == 0:
                self.level = self.sustain_level
                self.complete_stage()
        elif self.stage == RELEASE:
            if self.samples_remaining > 0:
                self.level = f32(self.level + self.release_increment)
                self.samples_remaining -= 1
                self.level = clamp(self.level, 0.0, 1.0)
            if self.samples_remaining == 0:
                self.level = f32(0)
                self.complete_stage()
        elif self.stage == SUSTAIN:
            self.level = self.sustain_level
        else:
            self.level = f32(0)

    def on_gate(self, velocity):  # handle_gate_event :250-273 (scalar payload)
        velocity = f32(velocity)
        if velocity > 0:
            self.velocity = clamp(velocity, 0.0, 1.0)
            self.update_sustain_level()
            if self.attack <= MIN_TIME_SECONDS:
                self.level = f32(1)
                self.set_stage(DECAY, self.sustain_level)
            else:
                self.set_stage(ATTACK, f32(1))
        elif self.release <= MIN_TIME_SECONDS:
            self.stage = IDLE
            self.level = f32(0)
            self.samples_remaining = 0
            self.release_increment = f32(0)
        else:
            self.set_stage(RELEASE, f32(0))

    def prepare(self, sr):
        self.sample_rate = f32(sr)
        self.update_sustain_level()

    def process(self):  # :282-292
        self.apply_parameters()
        self.process_stage()
        self.output = self.level


class FmOperator:
    """nodes/fm_operator.rs"""

    def __init__(self):
        self.phase = self.prev_output = f32(0)
        self.sample_rate = f32(44100.0)
        self.base_freq, self.ratio = f32(440), f32(1)
        self.phase_mod = self.feedback = f32(0)
        self.envelope = self.level = f32(1)
        self.output = f32(0)

    def process(self):  # :58-76
        frequency = f32(self.base_freq * self.ratio)
        feedback_mod = f32(self.prev_output * self.feedback)
        total_phase_mod = f32(self.phase_mod + feedback_mod)
        phase_rad = f32(f32(self.phase + total_phase_mod) * TAU)
        output = f32(f32(sinf(phase_rad) * self.envelope) * self.level)
        self.output = output
        self.prev_output = output
        phase_inc = f32(frequency / self.sample_rate)
        self.phase = f32(self.phase + phase_inc)
        self.phase = f32(self.phase - np.trunc(self.phase))  # fract()


class Tpt:
    """filters/tpt/mod.rs, F = f32"""

    def __init__(self, cutoff, q):  # new() :47-66
        self.input = f32(0)
        self.cutoff, self.q, self.f_mod = f32(cutoff), f32(q), f32(0)
        self.output = f32(0)
        self.current_cutoff, self.current_q = f32(cutoff), f32(q)
        self.z = [f32(0), f32(0)]
        self.h = self.g = self.r = self.k = f32(0)
        self.sample_rate = f32(44100.0)
        self.update_coefficients(f32(44100.0), self.cutoff, self.q)

    def update_coefficients(self, sample_rate, cutoff, q):  # :69-82
        nyquist = f32(f32(sample_rate * f32(0.5)) - EPS)
        freq = clamp(cutoff, 20.0, nyquist)
        period = f32(f32(0.5) / sample_rate)
        f = f32(f32(f32(f32(2) * sample_rate) * tanf(f32(f32(f32(f32(2) * PI) * freq) * period))) * period)
        inv_q = f32(f32(1) / q)
        self.h = f32(f32(1) / f32(f32(f32(1) + f32(inv_q * f)) + f32(f * f)))
        self.g = f
        self.r = inv_q
        self.k = f32(self.g + self.r)
        self.current_cutoff = f32(cutoff)
        self.current_q = f32(q)

    def prepare(self, sr):  # :129-131
        self.sample_rate = f32(sr)
        self.update_coefficients(self.sample_rate, self.cutoff, self.q)

    def process(self):  # :85-122
        sr = self.sample_rate
        nyquist = f32(f32(sr * f32(0.5)) - EPS)
        max_cutoff = min(nyquist, f32(20000))
        cutoff_base = clamp(self.cutoff, 20.0, max_cutoff)
        q = clamp(self.q, 0.1, 10.0)
        modulation = clamp(self.f_mod, -1.0, 1.0)
        min_factor = f32(f32(20) / cutoff_base)
        max_factor = f32(max_cutoff / cutoff_base)
        factor = clamp(f32(f32(1) + modulation), min_factor, max_factor)
        cutoff = clamp(f32(cutoff_base * factor), 20.0, max_cutoff)
        if abs(f32(cutoff - self.current_cutoff)) > EPS or abs(f32(q - self.current_q)) > EPS:
            self.update_coefficients(sr, cutoff, q)
        high = f32(f32(f32(self.input - f32(self.z[0] * self.k)) - self.z[1]) * self.h)
        band = f32(f32(high * self.g) + self.z[0])
        low = f32(f32(band * self.g) + self.z[1])
        self.z[0] = f32(f32(high * self.g) + band)
        self.z[1] = f32(f32(band * self.g) + low)
        self.output = low


FM_DEFAULTS = dict(  # fm_voice.rs:10-48
    op3_ratio=3.0, op3_level=0.5, op3_feedback=0.0, op3_attack=0.01, op3_decay=0.1, op3_sustain=0.7, op3_release=0.3,
    op2_ratio=2.0, op2_level=0.5, op2_feedback=0.0, op2_attack=0.01, op2_decay=0.1, op2_sustain=0.7, op2_release=0.3,
    op1_ratio=1.0, op1_attack=0.01, op1_decay=0.2, op1_sustain=0.8, op1_release=0.5, route=0.0, filter_cutoff=2000.0,
    filter_resonance=0.707, filter_attack=0.01, filter_decay=0.2, filter_sustain=0.5, filter_release=0.3,
    filter_env_amount=0.0)


class FMVoice:
    """fm_voice.rs: nodes :51-80, connections :82-155"""

    def __init__(self, sr):
        self.p = {k: f32(v) for k, v in FM_DEFAULTS.items()}
        self.frequency = f32(440)
        self.env3, self.env2 = Adsr(0.01, 0.1, 0.7, 0.3), Adsr(0.01, 0.1, 0.7, 0.3)
        self.env1, self.env_filter = Adsr(0.01, 0.2, 0.8, 0.5), Adsr(0.01, 0.2, 0.5, 0.3)
        self.op3, self.op2, self.op1 = FmOperator(), FmOperator(), FmOperator()
        self.filter = Tpt(2000.0, 0.707)
        for e in (self.env3, self.env2, self.env1, self.env_filter):
            e.prepare(sr)
        for o in (self.op3, self.op2, self.op1):
            o.sample_rate = f32(sr)
        self.filter.prepare(sr)
        self.audio_out = f32(0)

    def frame(self, gate=None):
        p = self.p
        # envelopes: value edges, then the gate event, then process() (emit_node.rs order)
        for env, pre in ((self.env3, "op3"), (self.env2, "op2"), (self.env1, "op1"), (self.env_filter, "filter")):
            env.attack, env.decay = p[pre + "_attack"], p[pre + "_decay"]
            env.sustain, env.release = p[pre + "_sustain"], p[pre + "_release"]
            if gate is not None:
                env.on_gate(gate)
            env.process()
        # filter envelope -> Gain(amount) -> AddValue(cutoff)      (gain/mod.rs:30-34, add_value.rs:31-35)
        filter_env_gain = f32(self.env_filter.output * p["filter_env_amount"])
        cutoff_mod = f32(filter_env_gain + p["filter_cutoff"])
        o3 = self.op3
        o3.base_freq, o3.ratio, o3.feedback = self.frequency, p["op3_ratio"], p["op3_feedback"]
        o3.envelope, o3.level = self.env3.output, p["op3_level"]
        o3.process()
        mix = clamp(p["route"], 0.0, 1.0)  # crossfade.rs:37-44
        route_a = f32(o3.output * f32(f32(1) - mix))
        route_b = f32(o3.output * mix)
        o2 = self.op2
        o2.phase_mod = route_a
        o2.base_freq, o2.ratio, o2.feedback = self.frequency, p["op2_ratio"], p["op2_feedback"]
        o2.envelope, o2.level = self.env2.output, p["op2_level"]
        o2.process()
        o1 = self.op1
        o1.phase_mod = f32(o2.output + route_b)  # mixer.rs:30-34
        o1.base_freq, o1.ratio, o1.envelope = self.frequency, p["op1_ratio"], self.env1.output
        o1.process()
        fl = self.filter
        fl.input, fl.cutoff, fl.q = o1.output, cutoff_mod, p["filter_resonance"]
        fl.process()
        self.audio_out = f32(fl.output * f32(0.3))  # output_gain = Gain::new(0.3)
        return self.audio_out


# ---------------------------------------------------------------------------------------------------------------
NUM_HARMONICS = 32
INTERPOLATION_STEPS = 64
VELOCITY_0_SPECTRUM = np.zeros(32, dtype=f32)
VELOCITY_0_SPECTRUM[:2] = [0.02, 0.05]
VELOCITY_127_SPECTRUM = np.array(
    [0.150869, 0.385766, 0.215543, 0.117811, 0.100411, 0.0128637, 0.0288844, 0.00243388, 0.00963092, 0.0035634, 0.00256945,
     0.00184799, 0.000399878, 0.000660576, 3.00995e-05, 0.00021866, 9.33705e-05, 0.000177973, 0.0002545, 0.000323602,
     0.000779045, 0.000116569, 0.000772873, 0.000364486, 0.000248027, 0.00018236, 3.27292e-05, 6.64988e-05, 0.0, 0.0, 0.0,
     0.0], dtype=f32)


class EPianoVoice:
    """electric_piano_voice.rs: AmplitudeSource :174-356 -> OscillatorBank :79-170 (numpy float32 arrays: element-wise
    ops round per element like the Rust loops)"""

    def __init__(self, sr):
        self.sr = f32(sr)
        self.frequency = f32(440)
        self.brightness, self.velocity_scaling, self.decay_rate = f32(30), f32(50), f32(90)
        self.harmonic_decay, self.key_scaling, self.release_rate = f32(70), f32(50), f32(40)
        self.current_value = np.zeros(32, dtype=f32)
        self.target_value = np.zeros(32, dtype=f32)
        self.decay = np.zeros(32, dtype=f32)
        self.release = np.zeros(32, dtype=f32)
        self.released = False
        self.note_pitch = f32(60)
        self.step = INTERPOLATION_STEPS
        self.osc_re = np.ones(32, dtype=f32)
        self.osc_im = np.zeros(32, dtype=f32)
        self.mul_re = np.ones(32, dtype=f32)
        self.mul_im = np.zeros(32, dtype=f32)
        self.last_frequency = f32(0)
        self.output = f32(0)

    def get_decay(self, note):  # :244-268
        base_decay_rate = f32(f32(f32(100) - self.decay_rate) / f32(40000))
        harmonic_scaling = f32(f32(1) - f32(f32(f32(100) - self.harmonic_decay) / f32(200000)))
        scaling_multiplier = f32(f32(f32(48) - note) / f32(12))
        key_scaling_factor = f32(scaling_multiplier * f32(self.key_scaling * f32(0.02)))
        if key_scaling_factor > 0:
            adjusted = f32(f32(1) - f32(base_decay_rate / f32(f32(1) + key_scaling_factor)))
        else:
            adjusted = f32(f32(1) - f32(base_decay_rate * f32(f32(1) - key_scaling_factor)))
        decay = np.zeros(32, dtype=f32)
        scaling = f32(1)
        for i in range(32):
            decay[i] = f32(adjusted * scaling)
            scaling = f32(scaling * harmonic_scaling)
        return decay

    def on_gate(self, velocity):  # AmplitudeSource::on_gate :308-318, OscillatorBank::on_gate :115-122
        velocity = f32(velocity)
        if velocity > 0:
            self.velocity = velocity  # trigger_note :292-299
            self.decay = self.get_decay(self.note_pitch)
            rel = f32(f32(0.999) - f32(f32(f32(100) - self.release_rate) / f32(1000)))
            self.release = np.full(32, rel, dtype=f32)
            amps = (VELOCITY_127_SPECTRUM * velocity + VELOCITY_0_SPECTRUM * f32(f32(1) - velocity)).astype(f32)
            bs = f32(f32(-0.2) + f32(f32(0.8) * f32(self.brightness * f32(0.01))))
            bs = f32(bs + f32(f32(f32(velocity * self.velocity_scaling) * f32(0.01)) * f32(0.5)))
            idx = np.arange(32, dtype=f32)
            amps = (amps * (f32(1) + bs * idx).astype(f32)).astype(f32)
            self.current_value = amps
            self.released = False
            self.step = 0
            self.osc_re = np.ones(32, dtype=f32)
            self.osc_im = np.zeros(32, dtype=f32)
        else:
            self.released = True  # release_note :301-304
            self.step = 0

    def frame(self, gate=None):
        if gate is not None:
            self.on_gate(gate)
        # AmplitudeSource::process :321-351
        if self.step == 0:
            mult = self.release if self.released else self.decay
            self.target_value = (self.current_value * mult).astype(f32)
        if self.step < INTERPOLATION_STEPS:
            t = f32(f32(self.step + 1) / f32(INTERPOLATION_STEPS))
            self.current_value = ((self.current_value * f32(f32(1) - t)).astype(f32) + (self.target_value * t).astype(f32)).astype(f32)
            self.step += 1
        else:
            self.current_value = self.target_value.copy()
            self.step = 0
        amplitudes = self.current_value
        # OscillatorBank::process :154-169
        if self.frequency > 0 and not (abs(f32(self.last_frequency - self.frequency)) < f32(0.01)):  # update_multipliers :126-150
            self.last_frequency = self.frequency
            nyquist = f32(self.sr * f32(0.5))
            for i in range(32):
                hf = f32(self.frequency * f32(i + 1))
                if hf < nyquist:
                    angle = f32(f32(f32(f32(2) * PI) * hf) / self.sr)
                    self.mul_re[i], self.mul_im[i] = cosf(angle), sinf(angle)
                else:
                    self.mul_re[i], self.mul_im[i] = f32(1), f32(0)
            self.osc_re = np.ones(32, dtype=f32)
            self.osc_im = np.zeros(32, dtype=f32)
        new_re = ((self.osc_re * self.mul_re).astype(f32) - (self.osc_im * self.mul_im).astype(f32)).astype(f32)  # Complex::mul :66-72
        new_im = ((self.osc_re * self.mul_im).astype(f32) + (self.osc_im * self.mul_re).astype(f32)).astype(f32)
        self.osc_re, self.osc_im = new_re, new_im
        prod = (self.osc_im * amplitudes).astype(f32)
        s = f32(0)
        for i in range(32):  # sequential f32 fold
            s = f32(s + prod[i])
        self.output = f32(s * f32(3))
        return self.output


# ---------------------------------------------------------------------------------------------------------------
# Round 3: the multirate path (BASELINE config 5) -- PolyBlepOscillator, HardClip, the 23-tap half-band sinc FIR
# resamplers and the SatGraph body -- transliterated from the RUST sources:
#   PolyBlepOscillator   oscen-lib/src/oscillators/mod.rs:88-233  (poly_blep :139-153, poly_blamp :155-169, process :176-232)
#   Halfband2xUpStage / Halfband2xDownStage / SincUpFir / SincDownFir   oscen-lib/src/resample/sinc_fir.rs:33-266
#   taps                 oscen-lib/src/resample/coeffs.rs:17-27
#   HardClip, SatGraph   examples/oversampled-saturator/src/main.rs:32-80; multirate frame body
#                        oscen-graph-compiler/src/codegen/emit_frame.rs:114-176 (inner nodes see sr * N, emit_struct.rs:575-587)
# ---------------------------------------------------------------------------------------------------------------
F32_EPSILON = f32(1.1920929e-07)
TAU = f32(6.2831855)  # std::f32::consts::TAU


def rem_euclid1(x):  # f32::rem_euclid(1.0): r = x % 1.0 (C fmod); if r < 0 { r + 1.0 }
    r = f32(np.fmod(f32(x), f32(1.0)))
    return f32(r + f32(1.0)) if r < f32(0.0) else r


class PolyBlep:
    SINE, SAW, SQUARE, TRIANGLE = range(4)

    def __init__(self, frequency, amplitude, waveform, sample_rate):
        self.phase = f32(0.0)
        self.phase_mod = f32(0.0)
        self.frequency = f32(frequency)
        self.frequency_mod = f32(0.0)
        self.amplitude = f32(amplitude)
        self.pulse_width = f32(0.5)
        self.output = f32(0.0)
        self.waveform = waveform
        self.sample_rate = f32(sample_rate)

    @staticmethod
    def poly_blep(t, dt):
        if dt <= F32_EPSILON:
            return f32(0.0)
        if t < dt:
            x = f32(t / dt)
            return f32(f32(f32(x + x) - f32(x * x)) - f32(1.0))
        if t > f32(f32(1.0) - dt):
            x = f32(f32(t - f32(1.0)) / dt)
            return f32(f32(f32(f32(x * x) + x) + x) + f32(1.0))
        return f32(0.0)

    @staticmethod
    def poly_blamp(t, dt):
        if dt <= F32_EPSILON:
            return f32(0.0)
        if t < dt:
            x = f32(f32(t / dt) - f32(1.0))
            return f32(f32(-f32(f32(x * x) * x)) / f32(3.0))
        if t > f32(f32(1.0) - dt):
            x = f32(f32(f32(t - f32(1.0)) / dt) + f32(1.0))
            return f32(f32(f32(x * x) * x) / f32(3.0))
        return f32(0.0)

    def process(self):
        frequency = max(f32(self.frequency * f32(f32(1.0) + self.frequency_mod)), f32(0.0))
        amplitude = self.amplitude
        pulse_width = clamp(self.pulse_width, 0.0001, 0.9999)
        phase = rem_euclid1(f32(self.phase + self.phase_mod))
        freq_per_sample = f32(frequency / max(self.sample_rate, F32_EPSILON))
        dt = min(freq_per_sample, f32(1.0))
        if pulse_width <= f32(0.0):
            pulse_width = f32(0.0001)
        if frequency >= f32(self.sample_rate * f32(0.25)):
            value = sinf(f32(phase * TAU))
        elif self.waveform == self.SINE:
            value = sinf(f32(phase * TAU))
        elif self.waveform == self.SAW:
            y = f32(f32(f32(2.0) * phase) - f32(1.0))
            value = f32(y - self.poly_blep(phase, dt))
        elif self.waveform == self.SQUARE:
            y = f32(1.0) if phase < pulse_width else f32(-1.0)
            y = f32(y + self.poly_blep(phase, dt))
            t = rem_euclid1(f32(f32(phase + f32(1.0)) - pulse_width))
            value = f32(y - self.poly_blep(t, dt))
        else:
            y = f32(f32(4.0) * phase)
            if y >= f32(3.0):
                y = f32(y - f32(4.0))
            elif y > f32(1.0):
                y = f32(f32(2.0) - y)
            t1 = rem_euclid1(f32(phase + f32(0.25)))
            t2 = rem_euclid1(f32(phase + f32(0.75)))
            value = f32(y + f32(f32(f32(4.0) * dt) * f32(self.poly_blamp(t1, dt) - self.poly_blamp(t2, dt))))
        value = f32(value * amplitude)
        self.output = value
        self.phase = rem_euclid1(f32(self.phase + freq_per_sample))
        return value


HALFBAND_23_HALF = [f32(-3.8558514e-5), f32(1.2218465e-3), f32(-7.2854808e-3), f32(2.6409210e-2), f32(-7.8128843e-2),
                    f32(3.0782697e-1)]
HALFBAND_23_CENTER = f32(0.4999897)


class HalfbandUp:  # Halfband2xUpStage  sinc_fir.rs:33-82
    def __init__(self):
        self.history = [f32(0.0)] * 12
        self.head = 0

    def step(self, x):
        cap = 12
        self.head = (self.head + 1) % cap
        self.history[self.head] = f32(x)
        at = lambda d: self.history[(self.head + cap - d) % cap]
        out1 = f32(at(5) * f32(f32(2.0) * HALFBAND_23_CENTER))
        acc = f32(0.0)
        for k, tap in enumerate(HALFBAND_23_HALF):
            acc = f32(acc + f32(f32(at(k) + at(11 - k)) * tap))
        return f32(acc * f32(2.0)), out1


class HalfbandDown:  # Halfband2xDownStage  sinc_fir.rs:96-144
    def __init__(self):
        self.history = [f32(0.0)] * 24
        self.head = 0

    def step(self, x0, x1):
        cap = 24
        self.head = (self.head + 1) % cap
        self.history[self.head] = f32(x0)
        self.head = (self.head + 1) % cap
        self.history[self.head] = f32(x1)
        at = lambda d: self.history[(self.head + cap - 1 - d) % cap]
        acc = f32(at(11) * HALFBAND_23_CENTER)
        for k, tap in enumerate(HALFBAND_23_HALF):
            acc = f32(acc + f32(f32(at(2 * k) + at(22 - 2 * k)) * tap))
        return acc


class SincUp:  # SincUpFir<N>::upsample  sinc_fir.rs:166-184
    def __init__(self, n):
        self.n = n
        self.stages = [HalfbandUp() for _ in range(n.bit_length() - 1)]

    def upsample(self, x):
        buf = [f32(x)]
        for st in self.stages:
            nxt = []
            for v in buf:
                a, b = st.step(v)
                nxt += [a, b]
            buf = nxt
        return buf


class SincDown:  # SincDownFir<N>::downsample  sinc_fir.rs:232-248
    def __init__(self, n):
        self.n = n
        self.stages = [HalfbandDown() for _ in range(n.bit_length() - 1)]

    def downsample(self, xs):
        buf = [f32(v) for v in xs]
        for st in self.stages:
            buf = [st.step(buf[2 * i], buf[2 * i + 1]) for i in range(len(buf) // 2)]
        return buf[0]


class SatVoice:
    """SatGraph_<N>x (examples/oversampled-saturator/src/main.rs:64-80): osc = PolyBlepOscillator::saw(f, 0.6) * N,
    clip = HardClip::new() * N, osc.output -> clip.input, [sinc] clip.output -> audio_out; N = 1: no resampler"""

    def __init__(self, sr, n, frequency):
        self.n = n
        self.osc = PolyBlep(frequency, 0.6, PolyBlep.SAW, f32(f32(sr) * f32(n)))
        self.down = SincDown(n) if n > 1 else None

    def frame(self):
        xs = []
        for _ in range(self.n):
            driven = f32(self.osc.process() * f32(1.5))
            xs.append(clamp(driven, -0.7, 0.7))
        return self.down.downsample(xs) if self.down else xs[0]


# ---------------------------------------------------------------------------------------------------------------------
# Round 3 (late): the nodes that still had one restatement only.  Written from the Rust sources named on each class.
# ---------------------------------------------------------------------------------------------------------------------
for _n in ("tanhf",):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.fmodf.restype = ctypes.c_float
_libm.fmodf.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.fmaf.restype = ctypes.c_float
_libm.fmaf.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float]

PI = f32(np.pi)  # std::f32::consts::PI


def tanhf(x):
    return f32(_libm.tanhf(float(x)))


def fmodf(x, y):  # Rust's `%` on f32
    return f32(_libm.fmodf(float(x), float(y)))


def fract(x):  # f32::fract = x - x.trunc()
    x = f32(x)
    return f32(x - f32(np.trunc(x)))


class Oscillator:  # oscen-lib/src/oscillators/mod.rs:7-77
    SINE, SQUARE, SAW = range(3)

    def __init__(self, frequency, amplitude, waveform, sr):
        self.phase = f32(0.0)
        self.frequency = f32(frequency)
        self.frequency_mod = f32(0.0)
        self.amplitude = f32(amplitude)
        self.waveform = waveform
        self.sr = f32(sr)
        self.output = f32(0.0)

    def _wave(self, p):
        if self.waveform == self.SINE:  # |p| (p * 2.0 * PI).sin()
            return sinf(f32(f32(p * f32(2.0)) * PI))
        if self.waveform == self.SQUARE:
            return f32(1.0) if p < f32(0.5) else f32(-1.0)
        transition_width = f32(0.1)
        raw_saw = f32(f32(f32(2.0) * p) - f32(1.0))
        edge = f32(f32(1.0) - f32(transition_width / f32(2.0)))
        if p > edge:
            t = f32(f32(p - edge) / f32(transition_width / f32(2.0)))
            return f32(f32(-1.0) + f32(f32(f32(1.0) - f32(t * t)) * f32(raw_saw + f32(1.0))))
        return raw_saw

    def process(self):
        frequency = f32(self.frequency * f32(f32(1.0) + self.frequency_mod))
        modulated_phase = fmodf(self.phase, 1.0)
        self.output = f32(self._wave(modulated_phase) * self.amplitude)
        self.phase = f32(self.phase + f32(frequency / self.sr))
        self.phase = fmodf(self.phase, 1.0)
        return self.output


class IirLowpass:  # oscen-lib/src/filters/iir_lowpass/mod.rs:15-163
    DENORMAL = f32(1e-15)

    def __init__(self, cutoff, q, sr):
        self.cutoff, self.q, self.sr = f32(cutoff), f32(q), f32(sr)
        self.b0, self.b1, self.b2, self.a1, self.a2 = f32(1.0), f32(0.0), f32(0.0), f32(0.0), f32(0.0)
        self.v1 = self.v2 = f32(0.0)
        self.frame_counter, self.frames_per_update = 0, 32
        self.update_coefficients()  # prepare()

    def update_coefficients(self):
        nyquist = f32(f32(self.sr * f32(0.5)) - f32(np.finfo(np.float32).eps))
        freq = clamp(self.cutoff, 20.0, nyquist)
        q = self.q if self.q > f32(0.01) else f32(0.01)  # .max(0.01)
        n = f32(f32(1.0) / tanf(f32(f32(PI * freq) / self.sr)))
        n_squared = f32(n * n)
        c1 = f32(f32(1.0) / f32(f32(f32(1.0) + f32(f32(f32(1.0) / q) * n)) + n_squared))
        self.b0 = c1
        self.b1 = f32(c1 * f32(2.0))
        self.b2 = c1
        self.a1 = f32(f32(c1 * f32(2.0)) * f32(f32(1.0) - n_squared))
        self.a2 = f32(c1 * f32(f32(f32(1.0) - f32(f32(f32(1.0) / q) * n)) + n_squared))

    def process(self, x):
        if self.frame_counter == 0:
            self.update_coefficients()
        self.frame_counter = (self.frame_counter + 1) % self.frames_per_update
        x = f32(x)
        if abs(x) < self.DENORMAL:
            x = f32(0.0)
        out = f32(f32(self.b0 * x) + self.v1)
        self.v1 = f32(f32(f32(self.b1 * x) - f32(self.a1 * out)) + self.v2)
        self.v2 = f32(f32(self.b2 * x) - f32(self.a2 * out))
        if abs(self.v1) < self.DENORMAL:
            self.v1 = f32(0.0)
        if abs(self.v2) < self.DENORMAL:
            self.v2 = f32(0.0)
        return out


class RingBuffer:  # oscen-lib/src/ring_buffer/mod.rs (PowerOfTwo mode, the default)
    def __init__(self, size):
        cap = 1
        while cap < max(size, 1):
            cap *= 2
        self.buf = np.zeros(cap, dtype=np.float32)
        self.capacity, self.mask, self.write_pos = cap, cap - 1, 0

    def push(self, v):
        self.buf[self.write_pos] = f32(v)
        self.write_pos = (self.write_pos + 1) & self.mask

    def read_pos(self, offset):
        n = f32(self.capacity)
        rp = f32(f32(f32(self.write_pos) - f32(offset)) - f32(1.0))
        return fmodf(f32(fmodf(rp, n) + n), n)

    def get_cubic(self, offset):
        rp = self.read_pos(offset)
        i = int(rp)
        f = fract(rp)
        v0, v1 = self.buf[(i - 1) & self.mask], self.buf[i]
        v2, v3 = self.buf[(i + 1) & self.mask], self.buf[(i + 2) & self.mask]
        c0 = v1
        c1 = f32(f32(0.5) * f32(v2 - v0))
        c2 = f32(f32(f32(v0 - f32(f32(2.5) * v1)) + f32(f32(2.0) * v2)) - f32(f32(0.5) * v3))
        c3 = f32(f32(f32(0.5) * f32(v3 - v0)) + f32(f32(1.5) * f32(v1 - v2)))
        return f32(c0 + f32(f * f32(c1 + f32(f * f32(c2 + f32(f * c3))))))

    def get_linear(self, offset):
        rp = self.read_pos(offset)
        i = int(rp)
        f = fract(rp)
        a, b = self.buf[i], self.buf[(i + 1) & self.mask]
        return f32(_libm.fmaf(float(a), float(f32(f32(1.0) - f)), float(f32(b * f))))  # a.mul_add(1.0 - f, b * f)

    def get(self, offset):
        off = f32(offset)
        if not off > f32(0.0):  # offset.max(0.0)
            off = f32(0.0)
        fr = fract(off)
        if fr < f32(1e-6) or f32(f32(1.0) - fr) < f32(1e-6):
            r = float(off)  # f32::round: half away from zero
            samples = int(np.floor(r + 0.5)) if r >= 0 else int(np.ceil(r - 0.5))
            idx = ((self.write_pos + self.capacity) - (samples % self.capacity) - 1) % self.capacity
            return f32(self.buf[idx])
        return self.get_cubic(off) if self.capacity >= 4 else self.get_linear(off)


class Delay:  # oscen-lib/src/delay/mod.rs:5-83
    def __init__(self, delay_samples, feedback, sr):
        self.delay_samples, self.feedback = f32(delay_samples), f32(feedback)
        size = min(int(f32(f32(2.0) * f32(sr))), 88200)  # prepare(): (target_seconds * sr) as usize, capped
        self.buffer = RingBuffer(size)
        self.frame_counter, self.frames_per_update = 0, 32

    def process(self, x):
        if self.frame_counter == 0:
            max_delay = f32(f32(self.buffer.capacity) - f32(1.0))
            self.delay_samples = clamp(self.delay_samples, 0.0, max_delay)
            self.feedback = clamp(self.feedback, 0.0, 0.99)
        self.frame_counter = (self.frame_counter + 1) % self.frames_per_update
        delayed = self.buffer.get(self.delay_samples)
        self.buffer.push(f32(f32(x) + f32(delayed * self.feedback)))
        return delayed


class Lp18:  # examples/nih-twin-peaks/src/lp18_filter.rs:7-104
    def __init__(self, cutoff, resonance, sr):
        self.cutoff, self.fmod, self.sr = f32(cutoff), f32(0.0), f32(sr)
        self.resonance = clamp(resonance, 0.0, 0.99)
        self.z = [f32(0.0)] * 3
        self.last_cutoff, self.last_fmod, self.last_resonance = f32(cutoff), f32(0.0), f32(resonance)
        self.update_cutoff()  # prepare()
        self.h = f32(f32(2.0) * self.resonance)

    def update_cutoff(self):
        fc = clamp(f32(f32(self.cutoff + self.fmod) / self.sr), 0.001, 0.33)
        self.g = tanf(f32(PI * fc))

    def process(self, x):
        if self.cutoff != self.last_cutoff or self.fmod != self.last_fmod:
            self.last_cutoff, self.last_fmod = self.cutoff, self.fmod
            self.update_cutoff()
        if self.resonance != self.last_resonance:
            self.last_resonance = self.resonance
            self.resonance = clamp(self.resonance, 0.0, 0.99)
            self.h = f32(f32(2.0) * self.resonance)
        g, z = self.g, self.z
        hp = f32(f32(f32(f32(f32(x) - f32(self.h * z[0])) - z[1]) - z[2]) / f32(f32(1.0) + g))
        bp1 = f32(f32(g * hp) + z[0])
        z[0] = tanhf(bp1)
        bp2 = f32(f32(g * bp1) + z[1])
        z[1] = bp2
        lp = f32(f32(g * bp2) + z[2])
        z[2] = lp
        return lp


class Tremolo:  # examples/electric-piano/src/tremolo.rs:8-67
    def __init__(self, sr):
        self.rate, self.depth, self.phase, self.sr = f32(5.0), f32(0.5), f32(0.0), f32(sr)

    def process(self, x):
        x = f32(x)
        lfo = sinf(f32(f32(self.phase * f32(2.0)) * PI))
        scaled_depth = f32(self.depth / f32(3.0))
        pan = f32(f32(0.5) + f32(lfo * scaled_depth))
        out = (f32(x * pan), f32(x * f32(f32(1.0) - pan)))
        self.phase = fract(f32(self.phase + f32(self.rate / self.sr)))
        return out


class _Allpass1:  # resample/halfband_iir.rs Allpass1::step
    TH = f32(1e-15)

    def __init__(self, a):
        self.a, self.x_prev, self.y_prev = f32(a), f32(0.0), f32(0.0)

    def step(self, x):
        x = f32(x)
        y = f32(f32(f32(x - self.y_prev) * self.a) + self.x_prev)
        self.x_prev = f32(0.0) if abs(x) < self.TH else x
        self.y_prev = f32(0.0) if abs(y) < self.TH else y
        return y


class _IirHalfband2x:  # IirHalfband2x::{step_up, step_down}
    A = (0.135_574_1, 0.697_584_9)  # resample/coeffs.rs BRANCH_A_BETAS / BRANCH_B_BETAS
    B = (0.425_380_4, 0.905_560_1)

    def __init__(self):
        self.a = [_Allpass1(c) for c in self.A]
        self.b = [_Allpass1(c) for c in self.B]
        self.prev_odd_in = f32(0.0)

    def step_up(self, x):
        a = b = f32(x)
        for s in self.a:
            a = s.step(a)
        for s in self.b:
            b = s.step(b)
        return a, b

    def step_down(self, x0, x1):
        a, b = f32(x0), self.prev_odd_in
        for s in self.a:
            a = s.step(a)
        for s in self.b:
            b = s.step(b)
        self.prev_odd_in = f32(x1)
        return f32(f32(a + b) * f32(0.5))


class IirUp:  # IirHalfbandUp<N>::upsample
    def __init__(self, n):
        self.n = n
        self.stages = [_IirHalfband2x() for _ in range(n.bit_length() - 1)]

    def upsample(self, x):
        buf = [f32(x)]
        for st in self.stages:
            nxt = []
            for v in buf:
                nxt.extend(st.step_up(v))
            buf = nxt
        return buf


class IirDown:  # IirHalfbandDown<N>::downsample
    def __init__(self, n):
        self.n = n
        self.stages = [_IirHalfband2x() for _ in range(n.bit_length() - 1)]

    def downsample(self, xs):
        buf = [f32(v) for v in xs]
        for st in self.stages:
            buf = [st.step_down(buf[2 * i], buf[2 * i + 1]) for i in range(len(buf) // 2)]
        return buf[0]


class LinearUp:  # resample/linear.rs:8-35
    def __init__(self, n):
        self.n, self.prev = n, f32(0.0)

    def upsample(self, x):
        x = f32(x)
        n_inv = f32(f32(1.0) / f32(self.n))
        delta = f32(x - self.prev)
        out = [f32(self.prev + f32(delta * f32(f32(i) * n_inv))) for i in range(self.n)]
        self.prev = x
        return out


def linear_down(xs):  # resample/linear.rs:48-66
    acc = f32(0.0)
    for x in xs:
        acc = f32(acc + f32(x))
    return f32(acc * f32(f32(1.0) / f32(len(xs))))
