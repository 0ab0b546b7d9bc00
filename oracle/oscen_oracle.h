/*
 * oscen_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A scalar, single-threaded, array-of-structs restatement in plain C of the
 * reference's (reedrosenbluth/oscen) per-sample graph-evaluation hot path.
 * One struct per reference node with the reference's field set, and the
 * reference's f32 operation order (build with -ffp-contract=off, no fast-math).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include/link/execute anything in this directory.  The product
 * (oscen_amd/) never does.
 *
 * PINNING STATUS.  The reference is nightly Rust and cannot be compiled or
 * imported in this image (no rustc/cargo), so there is no oracle/_ref build.
 * The oracle is pinned against every known answer the reference's own tests
 * hold for this path (tests/test_oracle_golden.py):
 *   - TptFilter 8-sample impulse response + coefficient identities + stereo
 *     independence  (oscen-lib/src/filters/tpt/mod.rs:152-264)
 *   - ValueRampState exact arithmetic (oscen-lib/src/graph/types.rs:379-503)
 *   - Latch/Linear resampler exact vectors, sinc/IIR DC-gain / pass-band /
 *     stop-band bounds (oscen-lib/tests/resample_kernels.rs)
 *   - PolyBLEP bounds (oscillators/mod.rs:239-303), ADSR windows
 *     (envelope/adsr.rs:313-386), MIDI note->Hz (midi.rs:237-250)
 *   - process_block(N) == N x process() bit-exactness law
 *     (oscen-lib/tests/block_processing_test.rs)
 *   - multirate properties (oscen-lib/tests/multirate_graph.rs)
 *   - IirLowpass coefficient formula / DC gain / stability / denormal snap
 *     (oscen-lib/src/filters/iir_lowpass/mod.rs:166-330)
 *   - RingBuffer exact / linear / Catmull-Rom reads and wrap-around vectors
 *     (oscen-lib/src/ring_buffer/tests.rs), on which Delay rests
 * The data vectors are committed as tests/golden/reference_vectors.json.
 * Everything else (FmOperator waveform, ADSR curve shape, OscillatorBank,
 * SincDown sample values, full-voice waveforms) is PARITY UNPINNED by any
 * reference vector: for those the oracle is a line-by-line transliteration
 * with file:line citations and nothing more.
 */
#ifndef OSCEN_ORACLE_H
#define OSCEN_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OO_MAX_EVENTS 32      /* oscen-lib/src/graph/types.rs:18  */
#define OO_MAX_BLOCK 512      /* oscen-lib/src/graph/types.rs:12  */
#define OO_DEFAULT_SR 44100.0f /* oscen-lib/src/graph/types.rs:258-263 */

/* ---- events: EventInstance{frame_offset, payload} types.rs:22,87-90,129-132 */
typedef struct {
    uint32_t frame_offset;
    float scalar;      /* EventPayload::Scalar(f32)                        */
    int32_t is_object; /* EventPayload::Object(..) -> handlers see no scalar */
} oo_event;

typedef struct {
    oo_event ev[OO_MAX_EVENTS];
    uint32_t len;
} oo_queue; /* StaticEventQueue = ArrayVec<EventInstance, 32> */

void oo_queue_clear(oo_queue *q);
int oo_queue_try_push(oo_queue *q, oo_event e); /* 0 ok, -1 overflow (dropped) */
void oo_queue_connect(const oo_queue *src, oo_queue *dst); /* clear + copy */

/* ---- ValueRampState  types.rs:300-373 ---------------------------------- */
typedef struct {
    float current, target, increment;
    uint32_t frames_remaining;
} oo_ramp;
void oo_ramp_new(oo_ramp *r, float initial);
void oo_ramp_set_immediate(oo_ramp *r, float v);
void oo_ramp_set_with_ramp(oo_ramp *r, float target, uint32_t frames);
int oo_ramp_tick(oo_ramp *r);       /* 1 when the ramp just completed */
int oo_ramp_is_ramping(const oo_ramp *r);

/* graph-level ramped input + generated setters  codegen/mod.rs:917-976 */
typedef struct {
    oo_ramp r;
    uint32_t default_frames;
} oo_ramped_input;
void oo_ramped_set(oo_ramped_input *in, uint32_t *active_ramps, float v);
void oo_ramped_set_with_ramp(oo_ramped_input *in, uint32_t *active_ramps, float v, uint32_t frames);
void oo_ramped_set_immediate(oo_ramped_input *in, uint32_t *active_ramps, float v);

/* ---- Oscillator  oscillators/mod.rs:7-76 ------------------------------- */
enum { OO_WAVE_SINE = 0, OO_WAVE_SQUARE = 1, OO_WAVE_SAW = 2 };
typedef struct {
    float phase, frequency, frequency_mod, amplitude, output;
    int32_t waveform;
    float sample_rate;
} oo_oscillator;
void oo_oscillator_new(oo_oscillator *o, float frequency, float amplitude, int waveform);
void oo_oscillator_process(oo_oscillator *o);

/* ---- PolyBlepOscillator  oscillators/mod.rs:79-233 --------------------- */
enum { OO_PB_SINE = 0, OO_PB_SAW = 1, OO_PB_SQUARE = 2, OO_PB_TRIANGLE = 3 };
typedef struct {
    float phase, phase_mod, frequency, frequency_mod, amplitude, pulse_width, output;
    int32_t waveform;
    float sample_rate;
} oo_polyblep;
void oo_polyblep_new(oo_polyblep *o, float frequency, float amplitude, int waveform);
void oo_polyblep_process(oo_polyblep *o);

/* ---- TptFilter<F>  filters/tpt/mod.rs:12-138 (F = f32 or Frame<2>) ----- */
typedef struct {
    float input[2], cutoff, q, f_mod, output[2];
    float current_cutoff, current_q;
    float z[2][2]; /* z[i][channel] */
    float h, g, r, k;
    float sample_rate;
    int32_t channels; /* 1 = f32, 2 = Frame<2> */
} oo_tpt;
void oo_tpt_new(oo_tpt *f, float cutoff, float q, int channels);
void oo_tpt_prepare(oo_tpt *f);
void oo_tpt_process(oo_tpt *f);

/* ---- AdsrEnvelope  envelope/adsr.rs:20-306 ----------------------------- */
enum { OO_ST_IDLE = 0, OO_ST_ATTACK, OO_ST_DECAY, OO_ST_SUSTAIN, OO_ST_RELEASE };
typedef struct {
    oo_queue gate;
    float attack, decay, sustain, release, output;
    int32_t stage;
    uint32_t attack_samples, decay_samples, release_samples, samples_remaining;
    float attack_coeff, decay_coeff, release_increment;
    float level, target_level, sustain_level, velocity;
    float sample_rate;
} oo_adsr;
void oo_adsr_new(oo_adsr *e, float a, float d, float s, float r);
void oo_adsr_prepare(oo_adsr *e);
void oo_adsr_handle_gate_event(oo_adsr *e, const oo_event *ev);
void oo_adsr_process_event_inputs(oo_adsr *e);
void oo_adsr_process(oo_adsr *e);

/* ---- small nodes ------------------------------------------------------- */
typedef struct { float input, gain, output; } oo_gain;             /* gain/mod.rs:5-35 */
typedef struct { float input, value, output; } oo_add_value;       /* fm-synth nodes/add_value.rs */
typedef struct { float input, mix, output_a, output_b; } oo_crossfade; /* nodes/crossfade.rs */
typedef struct { float input_a, input_b, output; } oo_mixer;       /* nodes/mixer.rs */
typedef struct { float input, output; } oo_hardclip;               /* oversampled-saturator/src/main.rs:32-62 */
void oo_gain_process(oo_gain *g);
void oo_add_value_process(oo_add_value *n);
void oo_crossfade_process(oo_crossfade *n);
void oo_mixer_process(oo_mixer *n);
void oo_hardclip_process(oo_hardclip *n);

/* ---- IirLowpass  oscen-lib/src/filters/iir_lowpass/mod.rs:15-164 -------- */
typedef struct {
    float input, cutoff, q, output;
    float b0, b1, b2, a1, a2, v1, v2;
    float sample_rate;
    uint32_t frame_counter, frames_per_update;
} oo_iir_lowpass;
void oo_iir_lowpass_new(oo_iir_lowpass *f, float cutoff, float q);
void oo_iir_lowpass_prepare(oo_iir_lowpass *f);
void oo_iir_lowpass_process(oo_iir_lowpass *f);
float oo_iir_lowpass_process_sample(oo_iir_lowpass *f, float input);

/* ---- RingBuffer  oscen-lib/src/ring_buffer/mod.rs:14-208 ---------------- */
enum { OO_RING_POW2 = 0, OO_RING_EXACT = 1 };
typedef struct {
    float *buffer; /* heap, `capacity` samples */
    size_t write_pos, capacity, mask;
    int32_t mode;
} oo_ring;
void oo_ring_new(oo_ring *r, size_t size, int mode); /* with_mode :35-54 */
void oo_ring_free(oo_ring *r);
void oo_ring_push(oo_ring *r, float v);
float oo_ring_get(const oo_ring *r, float offset);
float oo_ring_get_linear(const oo_ring *r, float offset);
float oo_ring_get_cubic(const oo_ring *r, float offset);

/* ---- Delay  oscen-lib/src/delay/mod.rs:5-86 ------------------------------ */
typedef struct {
    float input, delay_samples, feedback, output;
    oo_ring buffer;
    float sample_rate;
    size_t frames_per_update, frame_counter;
} oo_delay;
void oo_delay_new(oo_delay *d, float delay_samples, float feedback);
void oo_delay_prepare(oo_delay *d);
void oo_delay_process(oo_delay *d);
void oo_delay_free(oo_delay *d);

/* ---- LP18Filter  examples/nih-twin-peaks/src/lp18_filter.rs:1-108 ------- */
typedef struct {
    float input, cutoff, fmod, resonance, output;
    float z[3], g, h, last_cutoff, last_fmod, last_resonance, sample_rate;
} oo_lp18;
void oo_lp18_new(oo_lp18 *f, float cutoff, float resonance);
void oo_lp18_prepare(oo_lp18 *f);
void oo_lp18_process(oo_lp18 *f);

/* ---- FmOperator  examples/fm-synth/src/nodes/fm_operator.rs:12-76 ------ */
typedef struct {
    float phase, prev_output, sample_rate;
    float base_freq, ratio, phase_mod, feedback, envelope, level, output;
} oo_fm_operator;
void oo_fm_operator_new(oo_fm_operator *o);
void oo_fm_operator_process(oo_fm_operator *o);

/* ---- resamplers  oscen-lib/src/resample/ -------------------------------- */
typedef struct { float history[24]; uint32_t head; } oo_hb_down_stage; /* sinc_fir.rs:96-144 */
typedef struct { float history[12]; uint32_t head; } oo_hb_up_stage;   /* sinc_fir.rs:33-82  */
typedef struct { oo_hb_down_stage st[3]; uint32_t n_stages, factor; } oo_sinc_down;
typedef struct { oo_hb_up_stage st[3]; uint32_t n_stages, factor; } oo_sinc_up;
void oo_sinc_down_new(oo_sinc_down *d, uint32_t factor);
float oo_sinc_down_process(oo_sinc_down *d, const float *xs);
uint32_t oo_sinc_down_latency(const oo_sinc_down *d);
void oo_sinc_up_new(oo_sinc_up *u, uint32_t factor);
void oo_sinc_up_process(oo_sinc_up *u, float x, float *out);
uint32_t oo_sinc_up_latency(const oo_sinc_up *u);

typedef struct { float a, x_prev, y_prev; } oo_allpass1;               /* halfband_iir.rs:30-63 */
typedef struct { oo_allpass1 a[2], b[2]; float prev_odd_in; } oo_iir_hb2x; /* :74-149 */
typedef struct { oo_iir_hb2x st[3]; uint32_t n_stages, factor; } oo_iir_resampler;
void oo_iir_resampler_new(oo_iir_resampler *r, uint32_t factor);
void oo_iir_up_process(oo_iir_resampler *r, float x, float *out);
float oo_iir_down_process(oo_iir_resampler *r, const float *xs);
uint32_t oo_iir_latency(const oo_iir_resampler *r);

typedef struct { float prev; uint32_t factor; } oo_linear_up;          /* linear.rs:12-46 */
void oo_linear_up_new(oo_linear_up *u, uint32_t factor);
void oo_linear_up_process(oo_linear_up *u, float x, float *out);
float oo_linear_down_process(uint32_t factor, const float *xs);        /* linear.rs:48-74 */
void oo_latch_up_process(uint32_t factor, float x, float *out);        /* latch.rs:8-31   */
float oo_latch_down_process(uint32_t factor, const float *xs);         /* latch.rs:33-54  */

/* ---- electric piano  examples/electric-piano/src/ ---------------------- */
#define OO_NUM_HARMONICS 32
typedef struct {
    float frequency;
    oo_queue gate;
    float brightness, velocity_scaling, decay_rate, harmonic_decay, key_scaling, release_rate;
    float amplitudes[OO_NUM_HARMONICS];
    float current_value[OO_NUM_HARMONICS], target_value[OO_NUM_HARMONICS];
    float decay[OO_NUM_HARMONICS], release[OO_NUM_HARMONICS];
    int32_t released;
    float note_pitch, velocity;
    uint32_t interpolation_step;
} oo_amplitude_source;
typedef struct {
    float frequency;
    oo_queue gate;
    float amplitudes[OO_NUM_HARMONICS];
    float output;
    float osc_re[OO_NUM_HARMONICS], osc_im[OO_NUM_HARMONICS];
    float mul_re[OO_NUM_HARMONICS], mul_im[OO_NUM_HARMONICS];
    float last_frequency, sample_rate;
} oo_oscillator_bank;
typedef struct { float input, rate, depth, output[2], phase, sample_rate; } oo_tremolo;
void oo_amplitude_source_new(oo_amplitude_source *a);
void oo_amplitude_source_process_event_inputs(oo_amplitude_source *a);
void oo_amplitude_source_process(oo_amplitude_source *a);
void oo_oscillator_bank_new(oo_oscillator_bank *b);
void oo_oscillator_bank_process_event_inputs(oo_oscillator_bank *b);
void oo_oscillator_bank_process(oo_oscillator_bank *b);
void oo_tremolo_new(oo_tremolo *t);
void oo_tremolo_process(oo_tremolo *t);

/* ---- MIDI contract feeding the path  midi.rs:69-72, 99-116, 147-171 ---- */
float oo_midi_note_to_freq(uint8_t note);
float oo_midi_velocity_to_gate(uint8_t velocity);

/* ======================================================================== */
/* Generated-graph restatements (hand-expanded graph! output, per           */
/* codegen/mod.rs:499-573, emit_frame.rs:29-176, emit_node.rs, emit_edge.rs) */
/* ======================================================================== */

/* FMVoice  examples/fm-synth/src/fm_voice.rs:6-156 (nested-graph node) */
typedef struct {
    /* graph inputs */
    float frequency;
    oo_queue gate;
    float op3_ratio, op3_level, op3_feedback, op3_attack, op3_decay, op3_sustain, op3_release;
    float op2_ratio, op2_level, op2_feedback, op2_attack, op2_decay, op2_sustain, op2_release;
    float op1_ratio, op1_attack, op1_decay, op1_sustain, op1_release;
    float route;
    float filter_cutoff, filter_resonance, filter_attack, filter_decay, filter_sustain,
        filter_release, filter_env_amount;
    /* output */
    float audio_out;
    /* nodes */
    oo_adsr env3, env2, env1, env_filter;
    oo_gain filter_env_gain;
    oo_add_value cutoff_mod;
    oo_fm_operator op3_osc, op2_osc, op1_osc;
    oo_crossfade op3_route;
    oo_mixer op1_mod_mixer;
    oo_tpt filter;
    oo_gain output_gain;
    float sample_rate;
} oo_fm_voice;
void oo_fm_voice_new(oo_fm_voice *v);
void oo_fm_voice_init(oo_fm_voice *v, float sample_rate);
void oo_fm_voice_process(oo_fm_voice *v); /* inherent process(): one frame */

/* Index of broadcast value inputs of the FM bank (order = fm_voice.rs:13-48) */
enum {
    OO_FM_OP3_RATIO = 0, OO_FM_OP3_LEVEL, OO_FM_OP3_FEEDBACK, OO_FM_OP3_ATTACK, OO_FM_OP3_DECAY,
    OO_FM_OP3_SUSTAIN, OO_FM_OP3_RELEASE,
    OO_FM_OP2_RATIO, OO_FM_OP2_LEVEL, OO_FM_OP2_FEEDBACK, OO_FM_OP2_ATTACK, OO_FM_OP2_DECAY,
    OO_FM_OP2_SUSTAIN, OO_FM_OP2_RELEASE,
    OO_FM_OP1_RATIO, OO_FM_OP1_ATTACK, OO_FM_OP1_DECAY, OO_FM_OP1_SUSTAIN, OO_FM_OP1_RELEASE,
    OO_FM_ROUTE,
    OO_FM_FILTER_CUTOFF, OO_FM_FILTER_RESONANCE, OO_FM_FILTER_ATTACK, OO_FM_FILTER_DECAY,
    OO_FM_FILTER_SUSTAIN, OO_FM_FILTER_RELEASE, OO_FM_FILTER_ENV_AMOUNT,
    OO_FM_NUM_PARAMS
};

/* Generic voice bank = the poly wrapper graph of examples/fm-synth/src/lib.rs:22-131
 * (and electric-piano/src/main.rs:33-97) from `voice_handlers.{frequency,gate} ->
 * voices.{frequency,gate}` downwards: N voices, broadcast params (ramped where
 * the wrapper declares [ramp: N]), `voices.out -> out` = sequential f32 sum. */
enum { OO_BANK_FM = 0, OO_BANK_SUB = 1, OO_BANK_EPIANO = 2, OO_BANK_SAT4X = 3, OO_BANK_SAT1X = 4 };
enum { OO_EV_GATE = 0, OO_EV_FREQ = 1 };

typedef struct oo_bank oo_bank;
oo_bank *oo_bank_create(int kind, uint32_t n_voices);
void oo_bank_destroy(oo_bank *b);
void oo_bank_init(oo_bank *b, float sample_rate);
uint32_t oo_bank_num_params(const oo_bank *b);
uint32_t oo_bank_channels(const oo_bank *b); /* 1, or 2 for e-piano (post-tremolo Frame<2>) */
/* value inputs: generated setters; ramped inputs follow codegen/mod.rs:917-976 */
int oo_bank_set_value(oo_bank *b, uint32_t param, float v);
int oo_bank_set_value_with_ramp(oo_bank *b, uint32_t param, float v, uint32_t frames);
int oo_bank_set_value_immediate(oo_bank *b, uint32_t param, float v);
void oo_bank_set_voice_frequency(oo_bank *b, uint32_t voice, float hz);
/* stage an event for the NEXT process_block: kind GATE (scalar velocity) or
 * FREQ (the MidiVoiceHandler frequency output changing on that frame). */
int oo_bank_push_event(oo_bank *b, uint32_t voice, uint32_t frame_offset, int kind, float value);
/* process_block(frames<=512): out_bus[frames*channels]; taps (optional):
 * per-voice mono output for voices tap_voices[0..n_taps), laid out [tap][frame]. */
void oo_bank_process_block(oo_bank *b, uint32_t frames, float *out_bus,
                           const uint32_t *tap_voices, uint32_t n_taps, float *taps);
/* same thing via `frames` calls of the single-frame process() with events
 * pushed right before their frame (block_processing_test.rs law) */
void oo_bank_process_per_sample(oo_bank *b, uint32_t frames, float *out_bus,
                                const uint32_t *tap_voices, uint32_t n_taps, float *taps);
/* f64 sum of the per-voice outputs of the last block, [frames] (bus-parity aid) */
const double *oo_bank_last_bus_f64(const oo_bank *b);
/* f64 sum of |per-voice output| of the last block, [frames] (scale of the bus-sum tolerance) */
const double *oo_bank_last_abs_f64(const oo_bank *b);
/* multi-threaded render for the CPU baseline: voices partitioned statically */
double oo_bank_bench(int kind, uint32_t n_voices, uint32_t frames_total, uint32_t block,
                     uint32_t n_threads, uint64_t seed, double *checksum);
/* the same with each thread's voices run as sub-banks of `group` voices (8 = the reference's
 * `[FMVoice; 8]` graph, cache-resident) and the note plans folded into `span` frames */
double oo_bank_bench_grouped(int kind, uint32_t n_voices, uint32_t frames_total, uint32_t block,
                             uint32_t n_threads, uint32_t group, uint64_t seed, uint32_t span, double *checksum);

/* bench graphs  oscen-lib/benches/static_vs_runtime.rs:5-66 */
typedef struct { oo_oscillator osc; oo_tpt filter; oo_gain gain; } oo_static_simple;
void oo_static_simple_new(oo_static_simple *g);
void oo_static_simple_init(oo_static_simple *g, float sr);
void oo_static_simple_process(oo_static_simple *g);
typedef struct {
    oo_polyblep osc1, osc2, osc3;
    oo_gain mix1, mix2, mix3, mixer, env_amount, vca;
    oo_adsr filter_env, amp_env;
    oo_tpt filter;
} oo_static_complex;
void oo_static_complex_new(oo_static_complex *g);
void oo_static_complex_init(oo_static_complex *g, float sr);
void oo_static_complex_process(oo_static_complex *g);

/* FM core cross-check  examples/fm-synth/src/waveform.rs:24-52 */
void oo_fm_compute_waveform(float op3_ratio, float op3_level, float op3_feedback,
                            float op2_ratio, float op2_level, float op2_feedback,
                            float op1_ratio, float route, uint32_t n, float *out);

/* deterministic synthetic note generator shared by tests & bench:
 * splitmix64, seed ^ voice */
typedef struct {
    uint8_t note, velocity;
    uint32_t on_frame, off_frame, retrig_frame;
    float frequency;
} oo_note_plan;
void oo_note_plan_for_voice(uint64_t seed, uint32_t voice, oo_note_plan *p);
/* the same plan with every frame scaled by span / 48000 (span == 0 or >= 48000: unchanged) */
void oo_note_plan_scaled(uint64_t seed, uint32_t voice, uint32_t span, oo_note_plan *p);
/* the note stream of one voice as a frame-sorted list of gate events; fold 0 = scaled plan, 1 = a slice of the
 * cyclic 1 s plan at its real event density (see oscen_oracle.c) */
typedef struct {
    uint32_t n;
    uint32_t frame[4];
    float value[4];
    float frequency;
} oo_note_events;
void oo_note_events_for_voice(uint64_t seed, uint32_t voice, uint32_t span, int fold, oo_note_events *out);
/* Multi-threaded render of voices [first_voice, first_voice + n_voices) over their (scaled) note plans.
 * Each thread walks its contiguous voice range in sub-banks of `group` voices (0 = one sub-bank per
 * thread), every sub-bank rendered block by block over the whole timeline before the next one starts
 * (group = 8 is the reference's own shape: `[FMVoice; 8]` per graph, examples/fm-synth/src/lib.rs:68-73).
 * mono64 / abs64 (optional, [frames_total]): f64 sums over all voices of the per-voice output and of its
 * magnitude (pre-Tremolo for the e-piano).  Returns the wall time in seconds. */
double oo_bank_render_mt(int kind, uint32_t first_voice, uint32_t n_voices, uint32_t frames_total, uint32_t block,
                         uint32_t n_threads, uint32_t group, uint64_t seed, uint32_t span, double *mono64,
                         double *abs64);
/* note-plan fold mode of the three multi-threaded entry points (process-wide; 0 = scale, 1 = slice) */
void oo_bench_set_fold(int fold);
/* scenario of the multi-threaded renders: immediate parameter values before frame 0, setter calls before a frame */
void oo_bench_scenario_clear(void);
int oo_bench_scenario_set(uint32_t param, float value);
int oo_bench_scenario_ramp(uint32_t param, float value, uint32_t at_frame);

#ifdef __cplusplus
}
#endif
#endif
