/*
 * oscen_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See oscen_oracle.h for scope and pinning status.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  Build: see oracle/Makefile (-O2 -ffp-contract=off,
 * no fast-math: Rust never contracts or reassociates f32 arithmetic).
 */
#include "oscen_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Rust f32 semantics helpers ---------------------------------------- */

#define F32_EPSILON 1.1920929e-7f /* f32::EPSILON */
#define F32_PI 3.14159274101257324f /* std::f32::consts::PI  */
#define F32_TAU 6.28318548202514648f /* std::f32::consts::TAU */

/* f32::clamp: `if x < min {min} else if x > max {max} else {x}` (NaN passes) */
static inline float rs_clamp(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* f32::max / f32::min ignore NaN (IEEE maxNum/minNum) */
static inline float rs_max(float a, float b) { return fmaxf(a, b); }
static inline float rs_min(float a, float b) { return fminf(a, b); }
/* f32::rem_euclid(rhs): r = self % rhs; if r < 0 { r + |rhs| } else { r } */
static inline float rs_rem_euclid(float x, float rhs)
{
    float r = fmodf(x, rhs);
    return (r < 0.0f) ? r + fabsf(rhs) : r;
}
/* f32::fract(): self - self.trunc() */
static inline float rs_fract(float x) { return x - truncf(x); }
/* `x as u32`: saturating, NaN -> 0 */
static inline uint32_t rs_as_u32(float x)
{
    if (!(x > 0.0f)) return 0u; /* negatives, -0, NaN */
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
static inline uint32_t u32_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t u32_max(uint32_t a, uint32_t b) { return a > b ? a : b; }

/* ---- event queues  graph/types.rs:137-241, static_context.rs:80-155 ---- */

void oo_queue_clear(oo_queue *q) { q->len = 0; }

int oo_queue_try_push(oo_queue *q, oo_event e)
{
    if (q->len >= OO_MAX_EVENTS) return -1; /* Err, dropped by every caller */
    q->ev[q->len++] = e;
    return 0;
}

/* ConnectEndpoints for event queues: clear destination, copy all (static_context.rs:80-104) */
void oo_queue_connect(const oo_queue *src, oo_queue *dst)
{
    dst->len = 0;
    for (uint32_t i = 0; i < src->len; ++i) (void)oo_queue_try_push(dst, src->ev[i]);
}

/* ---- ValueRampState  graph/types.rs:300-373 ---------------------------- */

void oo_ramp_new(oo_ramp *r, float initial)
{
    r->current = initial;
    r->target = initial;
    r->increment = 0.0f;
    r->frames_remaining = 0;
}
void oo_ramp_set_immediate(oo_ramp *r, float v)
{
    r->current = v;
    r->target = v;
    r->increment = 0.0f;
    r->frames_remaining = 0;
}
void oo_ramp_set_with_ramp(oo_ramp *r, float target, uint32_t frames)
{
    if (frames == 0) {
        oo_ramp_set_immediate(r, target);
    } else {
        r->target = target;
        r->increment = (target - r->current) / (float)frames;
        r->frames_remaining = frames;
    }
}
int oo_ramp_tick(oo_ramp *r)
{
    if (r->frames_remaining > 0) {
        r->frames_remaining -= 1;
        if (r->frames_remaining == 0) {
            r->current = r->target;
            r->increment = 0.0f;
            return 1;
        } else {
            r->current += r->increment;
        }
    }
    return 0;
}
int oo_ramp_is_ramping(const oo_ramp *r) { return r->frames_remaining > 0; }

/* generated setters  oscen-graph-compiler/src/codegen/mod.rs:917-976 */
void oo_ramped_set(oo_ramped_input *in, uint32_t *active, float v)
{
    if (v != in->r.target) {
        if (!oo_ramp_is_ramping(&in->r)) *active += 1;
        oo_ramp_set_with_ramp(&in->r, v, in->default_frames);
    }
}
void oo_ramped_set_with_ramp(oo_ramped_input *in, uint32_t *active, float v, uint32_t frames)
{
    if (v != in->r.target) {
        if (frames > 0 && !oo_ramp_is_ramping(&in->r)) *active += 1;
        oo_ramp_set_with_ramp(&in->r, v, frames);
    }
}
void oo_ramped_set_immediate(oo_ramped_input *in, uint32_t *active, float v)
{
    if (oo_ramp_is_ramping(&in->r)) *active -= 1;
    oo_ramp_set_immediate(&in->r, v);
}

/* ---- Oscillator  oscillators/mod.rs:7-76 ------------------------------- */

void oo_oscillator_new(oo_oscillator *o, float frequency, float amplitude, int waveform)
{
    o->phase = 0.0f;
    o->frequency = frequency;
    o->frequency_mod = 0.0f;
    o->amplitude = amplitude;
    o->output = 0.0f;
    o->waveform = waveform;
    o->sample_rate = OO_DEFAULT_SR;
}

static float osc_waveform(int w, float p)
{
    switch (w) {
    case OO_WAVE_SINE: /* mod.rs:37-39: (p * 2.0 * PI).sin() */
        return sinf(p * 2.0f * F32_PI);
    case OO_WAVE_SQUARE: /* mod.rs:41-43 */
        return (p < 0.5f) ? 1.0f : -1.0f;
    default: { /* saw, mod.rs:46-62 */
        const float transition_width = 0.1f;
        float raw_saw = 2.0f * p - 1.0f;
        if (p > (1.0f - transition_width / 2.0f)) {
            float t = (p - (1.0f - transition_width / 2.0f)) / (transition_width / 2.0f);
            return -1.0f + (1.0f - t * t) * (raw_saw + 1.0f);
        }
        return raw_saw;
    }
    }
}

void oo_oscillator_process(oo_oscillator *o) /* mod.rs:65-76 */
{
    float frequency = o->frequency * (1.0f + o->frequency_mod);
    float amplitude = o->amplitude;
    float modulated_phase = fmodf(o->phase, 1.0f);
    o->output = osc_waveform(o->waveform, modulated_phase) * amplitude;
    o->phase += frequency / o->sample_rate;
    o->phase = fmodf(o->phase, 1.0f);
}

/* ---- PolyBlepOscillator  oscillators/mod.rs:88-233 --------------------- */

void oo_polyblep_new(oo_polyblep *o, float frequency, float amplitude, int waveform)
{
    o->phase = 0.0f;
    o->phase_mod = 0.0f;
    o->frequency = frequency;
    o->frequency_mod = 0.0f;
    o->amplitude = amplitude;
    o->pulse_width = 0.5f;
    o->output = 0.0f;
    o->waveform = waveform;
    o->sample_rate = OO_DEFAULT_SR;
}

static float poly_blep(float t, float dt) /* mod.rs:139-153 */
{
    if (dt <= F32_EPSILON) return 0.0f;
    if (t < dt) {
        float x = t / dt;
        return x + x - x * x - 1.0f;
    } else if (t > 1.0f - dt) {
        float x = (t - 1.0f) / dt;
        return x * x + x + x + 1.0f;
    }
    return 0.0f;
}

static float poly_blamp(float t, float dt) /* mod.rs:155-169 */
{
    if (dt <= F32_EPSILON) return 0.0f;
    if (t < dt) {
        float x = t / dt - 1.0f;
        return -(x * x * x) / 3.0f;
    } else if (t > 1.0f - dt) {
        float x = (t - 1.0f) / dt + 1.0f;
        return (x * x * x) / 3.0f;
    }
    return 0.0f;
}

static inline float wrap_phase(float p) { return rs_rem_euclid(p, 1.0f); } /* mod.rs:171-173 */

void oo_polyblep_process(oo_polyblep *o) /* mod.rs:176-233 */
{
    float frequency = rs_max(o->frequency * (1.0f + o->frequency_mod), 0.0f);
    float amplitude = o->amplitude;
    float pulse_width = rs_clamp(o->pulse_width, 0.0001f, 0.9999f);

    float phase = wrap_phase(o->phase + o->phase_mod);
    float freq_per_sample = frequency / rs_max(o->sample_rate, F32_EPSILON);
    float dt = rs_min(freq_per_sample, 1.0f);

    if (pulse_width <= 0.0f) pulse_width = 0.0001f;

    float value;
    if (frequency >= o->sample_rate * 0.25f) {
        value = sinf(phase * F32_TAU);
    } else {
        switch (o->waveform) {
        case OO_PB_SINE:
            value = sinf(phase * F32_TAU);
            break;
        case OO_PB_SAW: {
            float y = 2.0f * phase - 1.0f;
            y -= poly_blep(phase, dt);
            value = y;
            break;
        }
        case OO_PB_SQUARE: {
            float y = (phase < pulse_width) ? 1.0f : -1.0f;
            y += poly_blep(phase, dt);
            float t = wrap_phase(phase + 1.0f - pulse_width);
            y -= poly_blep(t, dt);
            value = y;
            break;
        }
        default: { /* Triangle */
            float y = 4.0f * phase;
            if (y >= 3.0f) {
                y -= 4.0f;
            } else if (y > 1.0f) {
                y = 2.0f - y;
            }
            float t1 = wrap_phase(phase + 0.25f);
            float t2 = wrap_phase(phase + 0.75f);
            value = y + 4.0f * dt * (poly_blamp(t1, dt) - poly_blamp(t2, dt));
            break;
        }
        }
    }

    value = value * amplitude;
    o->output = value;

    phase = wrap_phase(o->phase + freq_per_sample);
    o->phase = phase;
}

/* ---- TptFilter  filters/tpt/mod.rs:47-138 ------------------------------ */

static void tpt_update_coefficients(oo_tpt *f, float sample_rate, float cutoff, float q) /* :69-82 */
{
    float nyquist = sample_rate * 0.5f - F32_EPSILON;
    float freq = rs_clamp(cutoff, 20.0f, nyquist);
    float period = 0.5f / sample_rate;
    float ff = (2.0f * sample_rate) * tanf(2.0f * F32_PI * freq * period) * period;
    float inv_q = 1.0f / q;

    f->h = 1.0f / (1.0f + inv_q * ff + ff * ff);
    f->g = ff;
    f->r = inv_q;
    f->k = f->g + f->r;
    f->current_cutoff = cutoff;
    f->current_q = q;
}

void oo_tpt_new(oo_tpt *f, float cutoff, float q, int channels) /* :48-67 */
{
    memset(f, 0, sizeof *f);
    f->cutoff = cutoff;
    f->q = q;
    f->f_mod = 0.0f;
    f->current_cutoff = cutoff;
    f->current_q = q;
    f->sample_rate = OO_DEFAULT_SR;
    f->channels = channels;
    tpt_update_coefficients(f, 44100.0f, cutoff, q);
}

void oo_tpt_prepare(oo_tpt *f) /* :129-131 */
{
    tpt_update_coefficients(f, f->sample_rate, f->cutoff, f->q);
}

void oo_tpt_process(oo_tpt *f) /* :85-123 */
{
    float sample_rate = f->sample_rate;
    /* apply_parameter_updates :85-102 */
    float nyquist = sample_rate * 0.5f - F32_EPSILON;
    float max_cutoff = rs_min(nyquist, 20000.0f);
    float cutoff_base = rs_clamp(f->cutoff, 20.0f, max_cutoff);
    float q = rs_clamp(f->q, 0.1f, 10.0f);

    float modulation = rs_clamp(f->f_mod, -1.0f, 1.0f);
    float min_factor = 20.0f / cutoff_base;
    float max_factor = max_cutoff / cutoff_base;
    float factor = rs_clamp(1.0f + modulation, min_factor, max_factor);
    float cutoff = rs_clamp(cutoff_base * factor, 20.0f, max_cutoff);

    if (fabsf(cutoff - f->current_cutoff) > F32_EPSILON || fabsf(q - f->current_q) > F32_EPSILON) {
        tpt_update_coefficients(f, sample_rate, cutoff, q);
    }

    /* state-variable filter :114-122, element-wise per channel */
    for (int c = 0; c < f->channels; ++c) {
        float high = (f->input[c] - f->z[0][c] * f->k - f->z[1][c]) * f->h;
        float band = high * f->g + f->z[0][c];
        float low = band * f->g + f->z[1][c];
        f->z[0][c] = high * f->g + band;
        f->z[1][c] = band * f->g + low;
        f->output[c] = low;
    }
}

/* ---- AdsrEnvelope  envelope/adsr.rs:57-306 ----------------------------- */

#define ADSR_MIN_TIME_SECONDS 1.0e-5f
#define ADSR_CURVE_TIME_CONSTANT 4.6051702f

static void adsr_update_release_increment(oo_adsr *e) /* :162-173 */
{
    if (e->samples_remaining == 0 || e->stage != OO_ST_RELEASE) {
        e->release_increment = 0.0f;
        return;
    }
    float current = rs_clamp(e->level, 0.0f, 1.0f);
    e->release_increment = (current <= 0.0f) ? 0.0f : -current / (float)e->samples_remaining;
}

static void adsr_recalculate_cached_steps(oo_adsr *e) /* :117-134 */
{
    float sample_rate = rs_max(e->sample_rate, 1.0f);

    e->attack_samples = rs_as_u32(rs_max(e->attack, ADSR_MIN_TIME_SECONDS) * sample_rate);
    e->attack_samples = u32_max(e->attack_samples, 1);

    e->decay_samples = rs_as_u32(rs_max(e->decay, ADSR_MIN_TIME_SECONDS) * sample_rate);
    e->decay_samples = u32_max(e->decay_samples, 1);

    e->release_samples = rs_as_u32(rs_max(e->release, ADSR_MIN_TIME_SECONDS) * sample_rate);
    e->release_samples = u32_max(e->release_samples, 1);

    e->attack_coeff = 1.0f - expf(-ADSR_CURVE_TIME_CONSTANT / (float)e->attack_samples);
    e->decay_coeff = 1.0f - expf(-ADSR_CURVE_TIME_CONSTANT / (float)e->decay_samples);
}

static void adsr_update_sustain_level(oo_adsr *e) /* :92-115 */
{
    e->sustain_level = rs_clamp(e->sustain * e->velocity, 0.0f, 1.0f);
    adsr_recalculate_cached_steps(e);
    switch (e->stage) {
    case OO_ST_ATTACK:
        if (e->samples_remaining > 0)
            e->samples_remaining = u32_max(u32_min(e->samples_remaining, e->attack_samples), 1);
        break;
    case OO_ST_DECAY:
        if (e->samples_remaining > 0)
            e->samples_remaining = u32_max(u32_min(e->samples_remaining, e->decay_samples), 1);
        break;
    case OO_ST_RELEASE:
        if (e->samples_remaining > 0)
            e->samples_remaining = u32_max(u32_min(e->samples_remaining, e->release_samples), 1);
        break;
    default:
        break;
    }
    switch (e->stage) {
    case OO_ST_DECAY:
    case OO_ST_SUSTAIN:
        e->target_level = e->sustain_level;
        break;
    case OO_ST_RELEASE:
        e->target_level = 0.0f;
        break;
    default:
        break;
    }
    if (e->stage == OO_ST_RELEASE) adsr_update_release_increment(e);
}

static void adsr_apply_parameters(oo_adsr *e) /* :84-90 */
{
    e->attack = rs_max(e->attack, 0.0f);
    e->decay = rs_max(e->decay, 0.0f);
    e->sustain = rs_clamp(e->sustain, 0.0f, 1.0f);
    e->release = rs_max(e->release, 0.0f);
    adsr_update_sustain_level(e);
}

static void adsr_complete_stage(oo_adsr *e);

static void adsr_set_stage(oo_adsr *e, int stage, float target_level) /* :136-160 */
{
    e->stage = stage;
    e->target_level = rs_clamp(target_level, 0.0f, 1.0f);

    uint32_t samples;
    switch (stage) {
    case OO_ST_ATTACK: samples = e->attack_samples; break;
    case OO_ST_DECAY: samples = e->decay_samples; break;
    case OO_ST_RELEASE: samples = e->release_samples; break;
    default: samples = 0; break;
    }

    if (samples == 0) {
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
        e->level = e->target_level;
        if (!(stage == OO_ST_SUSTAIN || stage == OO_ST_IDLE)) adsr_complete_stage(e);
    } else {
        e->samples_remaining = samples;
        adsr_update_release_increment(e);
    }
}

static void adsr_complete_stage(oo_adsr *e) /* :175-204 */
{
    switch (e->stage) {
    case OO_ST_ATTACK:
        e->level = 1.0f;
        adsr_set_stage(e, OO_ST_DECAY, e->sustain_level);
        break;
    case OO_ST_DECAY:
        e->level = e->sustain_level;
        e->stage = OO_ST_SUSTAIN;
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
        break;
    case OO_ST_RELEASE:
        e->level = 0.0f;
        e->stage = OO_ST_IDLE;
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
        break;
    case OO_ST_SUSTAIN:
        e->level = e->sustain_level;
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
        break;
    default: /* Idle */
        e->level = 0.0f;
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
        break;
    }
}

static void adsr_process_stage(oo_adsr *e) /* :206-248 */
{
    switch (e->stage) {
    case OO_ST_ATTACK:
        if (e->samples_remaining > 0) {
            e->level += (1.0f - e->level) * e->attack_coeff;
            e->samples_remaining -= 1;
            e->level = rs_clamp(e->level, 0.0f, 1.0f);
        }
        if (e->samples_remaining == 0) {
            e->level = 1.0f;
            adsr_complete_stage(e);
        }
        break;
    case OO_ST_DECAY:
        if (e->samples_remaining > 0) {
            e->level += (e->sustain_level - e->level) * e->decay_coeff;
            e->samples_remaining -= 1;
            e->level = rs_clamp(e->level, 0.0f, 1.0f);
        }
        if (e->samples_remaining == 0) {
            e->level = e->sustain_level;
            adsr_complete_stage(e);
        }
        break;
    case OO_ST_RELEASE:
        if (e->samples_remaining > 0) {
            e->level += e->release_increment;
            e->samples_remaining -= 1;
            e->level = rs_clamp(e->level, 0.0f, 1.0f);
        }
        if (e->samples_remaining == 0) {
            e->level = 0.0f;
            adsr_complete_stage(e);
        }
        break;
    case OO_ST_SUSTAIN:
        e->level = e->sustain_level;
        break;
    default:
        e->level = 0.0f;
        break;
    }
}

void oo_adsr_new(oo_adsr *e, float a, float d, float s, float r) /* :57-82 */
{
    memset(e, 0, sizeof *e);
    e->attack = a;
    e->decay = d;
    e->sustain = s;
    e->release = r;
    e->stage = OO_ST_IDLE;
    e->sustain_level = rs_clamp(s, 0.0f, 1.0f);
    e->velocity = 1.0f;
    e->sample_rate = OO_DEFAULT_SR;
    adsr_update_sustain_level(e);
}

void oo_adsr_prepare(oo_adsr *e) { adsr_update_sustain_level(e); } /* :276-279 */

void oo_adsr_handle_gate_event(oo_adsr *e, const oo_event *ev) /* :250-273 */
{
    float velocity = ev->is_object ? 1.0f : ev->scalar;

    if (velocity > 0.0f) {
        e->velocity = rs_clamp(velocity, 0.0f, 1.0f);
        adsr_update_sustain_level(e);
        if (e->attack <= ADSR_MIN_TIME_SECONDS) {
            e->level = 1.0f;
            adsr_set_stage(e, OO_ST_DECAY, e->sustain_level);
        } else {
            adsr_set_stage(e, OO_ST_ATTACK, 1.0f);
        }
    } else if (e->release <= ADSR_MIN_TIME_SECONDS) {
        e->stage = OO_ST_IDLE;
        e->level = 0.0f;
        e->samples_remaining = 0;
        e->release_increment = 0.0f;
    } else {
        adsr_set_stage(e, OO_ST_RELEASE, 0.0f);
    }
}

/* derive(Node) process_event_inputs: oscen-macros/src/lib.rs:266-295 */
void oo_adsr_process_event_inputs(oo_adsr *e)
{
    oo_queue tmp = e->gate; /* collect into a temp ArrayVec, then dispatch */
    for (uint32_t i = 0; i < tmp.len; ++i) oo_adsr_handle_gate_event(e, &tmp.ev[i]);
}

void oo_adsr_process(oo_adsr *e) /* :281-291 */
{
    adsr_apply_parameters(e);
    adsr_process_stage(e);
    e->output = e->level;
}

/* ---- IirLowpass  filters/iir_lowpass/mod.rs ----------------------------- */

static void iir_lowpass_update_coefficients(oo_iir_lowpass *f, float sample_rate) /* :84-101 */
{
    float nyquist = sample_rate * 0.5f - F32_EPSILON;
    float freq = rs_clamp(f->cutoff, 20.0f, nyquist);
    float q = rs_max(f->q, 0.01f);
    float n = 1.0f / tanf(F32_PI * freq / sample_rate);
    float n_squared = n * n;
    float c1 = 1.0f / (1.0f + 1.0f / q * n + n_squared);
    f->b0 = c1;
    f->b1 = c1 * 2.0f;
    f->b2 = c1;
    f->a1 = c1 * 2.0f * (1.0f - n_squared);
    f->a2 = c1 * (1.0f - 1.0f / q * n + n_squared);
}

void oo_iir_lowpass_new(oo_iir_lowpass *f, float cutoff, float q) /* :43-76 */
{
    memset(f, 0, sizeof *f);
    f->cutoff = cutoff;
    f->q = q;
    f->b0 = 1.0f;
    f->frames_per_update = 32;
    f->sample_rate = OO_DEFAULT_SR;
}

void oo_iir_lowpass_prepare(oo_iir_lowpass *f) { iir_lowpass_update_coefficients(f, f->sample_rate); } /* :149-151 */

float oo_iir_lowpass_process_sample(oo_iir_lowpass *f, float input) /* :109-134 */
{
    const float DENORMAL_THRESHOLD = 1e-15f;
    if (fabsf(input) < DENORMAL_THRESHOLD) input = 0.0f;
    float output = f->b0 * input + f->v1;
    f->v1 = f->b1 * input - f->a1 * output + f->v2;
    f->v2 = f->b2 * input - f->a2 * output;
    if (fabsf(f->v1) < DENORMAL_THRESHOLD) f->v1 = 0.0f;
    if (fabsf(f->v2) < DENORMAL_THRESHOLD) f->v2 = 0.0f;
    return output;
}

void oo_iir_lowpass_process(oo_iir_lowpass *f) /* :137-143, 153-162 */
{
    if (f->frame_counter == 0) iir_lowpass_update_coefficients(f, f->sample_rate);
    f->frame_counter = (f->frame_counter + 1) % f->frames_per_update;
    f->output = oo_iir_lowpass_process_sample(f, f->input);
}

/* ---- RingBuffer  ring_buffer/mod.rs --------------------------------------- */

void oo_ring_new(oo_ring *r, size_t size, int mode) /* :35-54 */
{
    size_t capacity = size < 1 ? 1 : size;
    if (mode == OO_RING_POW2) {
        size_t p = 1;
        while (p < capacity) p <<= 1;
        capacity = p;
    }
    r->buffer = (float *)calloc(capacity, sizeof(float));
    r->write_pos = 0;
    r->capacity = capacity;
    r->mask = capacity - 1;
    r->mode = mode;
}

void oo_ring_free(oo_ring *r)
{
    free(r->buffer);
    r->buffer = NULL;
}

static size_t ring_wrap(const oo_ring *r, size_t i) /* & mask  |  % capacity */
{
    return r->mode == OO_RING_POW2 ? (i & r->mask) : (i % r->capacity);
}

void oo_ring_push(oo_ring *r, float v) /* :57-77 */
{
    r->buffer[r->write_pos] = v;
    r->write_pos = ring_wrap(r, r->write_pos + 1);
}

static float ring_read_pos(const oo_ring *r, float offset) /* :81-92 */
{
    float n = (float)r->capacity;
    float rp = (float)r->write_pos - offset - 1.0f;
    return fmodf(fmodf(rp, n) + n, n);
}

float oo_ring_get_linear(const oo_ring *r, float offset) /* :95-118 */
{
    float rp = ring_read_pos(r, offset);
    size_t i = (size_t)rp;
    float f = rp - truncf(rp);
    float a = r->buffer[i];
    float b = r->buffer[ring_wrap(r, i + 1)];
    return fmaf(a, 1.0f - f, b * f); /* a.mul_add(1.0 - f, b * f) */
}

float oo_ring_get_cubic(const oo_ring *r, float offset) /* :121-164 */
{
    if (r->capacity < 4) return oo_ring_get_linear(r, offset);
    float rp = ring_read_pos(r, offset);
    size_t i = (size_t)rp;
    float f = rp - truncf(rp);
    size_t im1 = r->mode == OO_RING_POW2 ? ((i - 1) & r->mask) : ((i + r->capacity - 1) % r->capacity);
    float v0 = r->buffer[im1];
    float v1 = r->buffer[i];
    float v2 = r->buffer[ring_wrap(r, i + 1)];
    float v3 = r->buffer[ring_wrap(r, i + 2)];
    float c0 = v1;
    float c1 = 0.5f * (v2 - v0);
    float c2 = v0 - 2.5f * v1 + 2.0f * v2 - 0.5f * v3;
    float c3 = 0.5f * (v3 - v0) + 1.5f * (v1 - v2);
    return c0 + f * (c1 + f * (c2 + f * c3));
}

static size_t f32_as_usize(float x) /* Rust `as usize`: saturating, NaN -> 0 */
{
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)x;
}

float oo_ring_get(const oo_ring *r, float offset) /* :167-203 */
{
    float o = rs_max(offset, 0.0f);
    float fr = o - truncf(o);
    if (fr < 1e-6f || (1.0f - fr) < 1e-6f) {
        size_t offset_samples = f32_as_usize(roundf(o));
        size_t read_idx = ((r->write_pos + r->capacity) - (offset_samples % r->capacity) - 1) % r->capacity;
        return r->buffer[read_idx];
    }
    if (r->capacity >= 4) return oo_ring_get_cubic(r, o);
    return oo_ring_get_linear(r, o);
}

/* ---- Delay  delay/mod.rs --------------------------------------------------- */

void oo_delay_new(oo_delay *d, float delay_samples, float feedback) /* :25-39 */
{
    memset(d, 0, sizeof *d);
    d->delay_samples = delay_samples;
    d->feedback = feedback;
    oo_ring_new(&d->buffer, 1024, OO_RING_POW2);
    d->sample_rate = OO_DEFAULT_SR;
    d->frames_per_update = 32;
}

void oo_delay_prepare(oo_delay *d) /* :59-69 */
{
    size_t buffer_size = f32_as_usize(2.0f * d->sample_rate);
    if (buffer_size > 88200) buffer_size = 88200;
    oo_ring_free(&d->buffer);
    oo_ring_new(&d->buffer, buffer_size, OO_RING_POW2);
}

void oo_delay_process(oo_delay *d) /* :47-56, 72-83 */
{
    if (d->frame_counter == 0) {
        float max_delay = (float)d->buffer.capacity - 1.0f;
        d->delay_samples = rs_clamp(d->delay_samples, 0.0f, max_delay);
        d->feedback = rs_clamp(d->feedback, 0.0f, 0.99f);
    }
    d->frame_counter = (d->frame_counter + 1) % d->frames_per_update;
    float delayed = oo_ring_get(&d->buffer, d->delay_samples);
    oo_ring_push(&d->buffer, d->input + delayed * d->feedback);
    d->output = delayed;
}

void oo_delay_free(oo_delay *d) { oo_ring_free(&d->buffer); }

/* ---- LP18Filter  examples/nih-twin-peaks/src/lp18_filter.rs -------------- */

static void lp18_update_cutoff(oo_lp18 *f) /* :63-67 */
{
    float modulated_cutoff = f->cutoff + f->fmod;
    float fc = rs_clamp(modulated_cutoff / f->sample_rate, 0.001f, 0.33f);
    f->g = tanf(F32_PI * fc);
}

void oo_lp18_new(oo_lp18 *f, float cutoff, float resonance) /* :45-61 */
{
    memset(f, 0, sizeof *f);
    f->cutoff = cutoff;
    f->resonance = rs_clamp(resonance, 0.0f, 0.99f);
    f->last_cutoff = cutoff;
    f->last_resonance = resonance;
    f->sample_rate = OO_DEFAULT_SR;
}

void oo_lp18_prepare(oo_lp18 *f) /* :75-78 */
{
    lp18_update_cutoff(f);
    f->h = 2.0f * f->resonance;
}

void oo_lp18_process(oo_lp18 *f) /* :80-107 */
{
    if (f->cutoff != f->last_cutoff || f->fmod != f->last_fmod) {
        f->last_cutoff = f->cutoff;
        f->last_fmod = f->fmod;
        lp18_update_cutoff(f);
    }
    if (f->resonance != f->last_resonance) {
        f->last_resonance = f->resonance;
        f->resonance = rs_clamp(f->resonance, 0.0f, 0.99f);
        f->h = 2.0f * f->resonance;
    }
    float hp = (f->input - f->h * f->z[0] - f->z[1] - f->z[2]) / (1.0f + f->g);
    float bp1 = f->g * hp + f->z[0];
    f->z[0] = tanhf(bp1);
    float bp2 = f->g * bp1 + f->z[1];
    f->z[1] = bp2;
    float lp = f->g * bp2 + f->z[2];
    f->z[2] = lp;
    f->output = lp;
}

/* ---- small nodes ------------------------------------------------------- */

void oo_gain_process(oo_gain *g) { g->output = g->input * g->gain; } /* gain/mod.rs:30-34 */
void oo_add_value_process(oo_add_value *n) { n->output = n->input + n->value; } /* add_value.rs:31-35 */
void oo_crossfade_process(oo_crossfade *n) /* crossfade.rs:37-44 */
{
    float mix = rs_clamp(n->mix, 0.0f, 1.0f);
    float input = n->input;
    n->output_a = input * (1.0f - mix);
    n->output_b = input * mix;
}
void oo_mixer_process(oo_mixer *n) { n->output = n->input_a + n->input_b; } /* mixer.rs:30-34 */
void oo_hardclip_process(oo_hardclip *n) /* oversampled-saturator/src/main.rs:54-61 */
{
    float driven = n->input * 1.5f;
    n->output = rs_clamp(driven, -0.7f, 0.7f);
}

/* ---- FmOperator  fm-synth/src/nodes/fm_operator.rs:33-76 --------------- */

void oo_fm_operator_new(oo_fm_operator *o)
{
    o->phase = 0.0f;
    o->prev_output = 0.0f;
    o->sample_rate = OO_DEFAULT_SR;
    o->base_freq = 440.0f;
    o->ratio = 1.0f;
    o->phase_mod = 0.0f;
    o->feedback = 0.0f;
    o->envelope = 1.0f;
    o->level = 1.0f;
    o->output = 0.0f;
}

void oo_fm_operator_process(oo_fm_operator *o) /* :58-76 */
{
    float frequency = o->base_freq * o->ratio;
    float feedback_mod = o->prev_output * o->feedback;
    float total_phase_mod = o->phase_mod + feedback_mod;
    float phase_rad = (o->phase + total_phase_mod) * F32_TAU;
    float output = sinf(phase_rad) * o->envelope * o->level;
    o->output = output;
    o->prev_output = output;
    float phase_inc = frequency / o->sample_rate;
    o->phase += phase_inc;
    o->phase = rs_fract(o->phase);
}

/* ---- resamplers  oscen-lib/src/resample/ ------------------------------- */

/* coeffs.rs:17-27 */
static const float HALFBAND_23_HALF[6] = {
    -3.8558514e-5f, 1.2218465e-3f, -7.2854808e-3f, 2.6409210e-2f, -7.8128843e-2f, 3.0782697e-1f,
};
static const float HALFBAND_23_CENTER = 0.4999897f;
#define HALFBAND_23_GROUP_DELAY 11u
/* coeffs.rs:48-54 */
static const float BRANCH_A_BETAS[2] = {0.1355741f, 0.6975849f};
static const float BRANCH_B_BETAS[2] = {0.4253804f, 0.9055601f};
#define IIR_HALFBAND_GROUP_DELAY 2u

static uint32_t n_stages_for(uint32_t factor)
{ /* (N as u32).trailing_zeros() for N in {1,2,4,8} */
    uint32_t n = 0;
    while ((1u << n) < factor) ++n;
    return n;
}

/* Halfband2xDownStage::step  sinc_fir.rs:115-138 */
static float hb_down_step(oo_hb_down_stage *s, float x0, float x1)
{
    const uint32_t cap = 24;
    s->head = (s->head + 1) % cap;
    s->history[s->head] = x0;
    s->head = (s->head + 1) % cap;
    s->history[s->head] = x1;
#define HB_AT(d) (s->history[(s->head + cap - 1 - (d)) % cap])
    float acc = HB_AT(11) * HALFBAND_23_CENTER;
    for (uint32_t k = 0; k < 6; ++k) {
        float left = HB_AT(2 * k);
        float right = HB_AT(22 - 2 * k);
        acc = acc + (left + right) * HALFBAND_23_HALF[k];
    }
#undef HB_AT
    return acc;
}

void oo_sinc_down_new(oo_sinc_down *d, uint32_t factor)
{
    memset(d, 0, sizeof *d);
    d->factor = factor;
    d->n_stages = n_stages_for(factor);
}

float oo_sinc_down_process(oo_sinc_down *d, const float *xs) /* sinc_fir.rs:232-247 */
{
    float buf[8] = {0};
    for (uint32_t i = 0; i < d->factor; ++i) buf[i] = xs[i];
    uint32_t len = d->factor;
    for (uint32_t s = 0; s < d->n_stages; ++s) {
        float next[8] = {0};
        uint32_t half = len / 2;
        for (uint32_t i = 0; i < half; ++i) next[i] = hb_down_step(&d->st[s], buf[2 * i], buf[2 * i + 1]);
        len = half;
        for (uint32_t i = 0; i < len; ++i) buf[i] = next[i];
    }
    return buf[0];
}

uint32_t oo_sinc_down_latency(const oo_sinc_down *d) /* :248-259 */
{
    return d->n_stages == 0 ? 0 : HALFBAND_23_GROUP_DELAY * ((1u << d->n_stages) - 1);
}

/* Halfband2xUpStage::step  sinc_fir.rs:50-76 */
static void hb_up_step(oo_hb_up_stage *s, float x, float out[2])
{
    const uint32_t cap = 12;
    s->head = (s->head + 1) % cap;
    s->history[s->head] = x;
#define UP_AT(d) (s->history[(s->head + cap - (d)) % cap])
    out[1] = UP_AT(5) * (2.0f * HALFBAND_23_CENTER);
    float acc = 0.0f;
    for (uint32_t k = 0; k < 6; ++k) {
        float left = UP_AT(k);
        float right = UP_AT(11 - k);
        acc = acc + (left + right) * HALFBAND_23_HALF[k];
    }
#undef UP_AT
    out[0] = acc * 2.0f;
}

void oo_sinc_up_new(oo_sinc_up *u, uint32_t factor)
{
    memset(u, 0, sizeof *u);
    u->factor = factor;
    u->n_stages = n_stages_for(factor);
}

void oo_sinc_up_process(oo_sinc_up *u, float x, float *out) /* sinc_fir.rs:170-190 */
{
    float buf[8] = {0}, next[8] = {0};
    uint32_t len = 1;
    buf[0] = x;
    for (uint32_t s = 0; s < u->n_stages; ++s) {
        for (uint32_t i = 0; i < len; ++i) {
            float pair[2] = {0, 0};
            hb_up_step(&u->st[s], buf[i], pair);
            next[2 * i] = pair[0];
            next[2 * i + 1] = pair[1];
        }
        len *= 2;
        for (uint32_t i = 0; i < len; ++i) buf[i] = next[i];
    }
    for (uint32_t i = 0; i < u->factor; ++i) out[i] = buf[i];
}

uint32_t oo_sinc_up_latency(const oo_sinc_up *u) /* :191-200 */
{
    return u->n_stages == 0 ? 0 : HALFBAND_23_GROUP_DELAY * ((1u << u->n_stages) - 1);
}

/* Allpass1::step  halfband_iir.rs:46-57;  f32::flush_denormal frame.rs */
static inline float flush_denormal(float x, float thr) { return (fabsf(x) < thr) ? 0.0f : x; }
static float allpass1_step(oo_allpass1 *s, float x)
{
    float y = (x - s->y_prev) * s->a + s->x_prev;
    s->x_prev = x;
    s->y_prev = y;
    s->x_prev = flush_denormal(s->x_prev, 1e-15f);
    s->y_prev = flush_denormal(s->y_prev, 1e-15f);
    return y;
}

void oo_iir_resampler_new(oo_iir_resampler *r, uint32_t factor) /* halfband_iir.rs:83-96 */
{
    memset(r, 0, sizeof *r);
    r->factor = factor;
    r->n_stages = n_stages_for(factor);
    for (uint32_t s = 0; s < 3; ++s) {
        for (int i = 0; i < 2; ++i) {
            r->st[s].a[i].a = BRANCH_A_BETAS[i];
            r->st[s].b[i].a = BRANCH_B_BETAS[i];
        }
    }
}

static void iir_step_up(oo_iir_hb2x *h, float x, float out[2]) /* :104-116 */
{
    float a = x;
    for (int i = 0; i < 2; ++i) a = allpass1_step(&h->a[i], a);
    float b = x;
    for (int i = 0; i < 2; ++i) b = allpass1_step(&h->b[i], b);
    out[0] = a;
    out[1] = b;
}

static float iir_step_down(oo_iir_hb2x *h, float x0, float x1) /* :124-136 */
{
    float a = x0;
    for (int i = 0; i < 2; ++i) a = allpass1_step(&h->a[i], a);
    float b = h->prev_odd_in;
    for (int i = 0; i < 2; ++i) b = allpass1_step(&h->b[i], b);
    h->prev_odd_in = x1;
    return (a + b) * 0.5f;
}

void oo_iir_up_process(oo_iir_resampler *r, float x, float *out) /* :176-194 */
{
    float buf[8] = {0}, next[8] = {0};
    uint32_t len = 1;
    buf[0] = x;
    for (uint32_t s = 0; s < r->n_stages; ++s) {
        for (uint32_t i = 0; i < len; ++i) {
            float pair[2] = {0, 0};
            iir_step_up(&r->st[s], buf[i], pair);
            next[2 * i] = pair[0];
            next[2 * i + 1] = pair[1];
        }
        len *= 2;
        for (uint32_t i = 0; i < len; ++i) buf[i] = next[i];
    }
    for (uint32_t i = 0; i < r->factor; ++i) out[i] = buf[i];
}

float oo_iir_down_process(oo_iir_resampler *r, const float *xs) /* :242-258 */
{
    float buf[8] = {0};
    for (uint32_t i = 0; i < r->factor; ++i) buf[i] = xs[i];
    uint32_t len = r->factor;
    for (uint32_t s = 0; s < r->n_stages; ++s) {
        float next[8] = {0};
        uint32_t half = len / 2;
        for (uint32_t i = 0; i < half; ++i) next[i] = iir_step_down(&r->st[s], buf[2 * i], buf[2 * i + 1]);
        len = half;
        for (uint32_t i = 0; i < len; ++i) buf[i] = next[i];
    }
    return buf[0];
}

uint32_t oo_iir_latency(const oo_iir_resampler *r)
{
    return r->n_stages == 0 ? 0 : IIR_HALFBAND_GROUP_DELAY * ((1u << r->n_stages) - 1);
}

void oo_linear_up_new(oo_linear_up *u, uint32_t factor)
{
    u->prev = 0.0f;
    u->factor = factor;
}
void oo_linear_up_process(oo_linear_up *u, float x, float *out) /* linear.rs:26-35 */
{
    float n_inv = 1.0f / (float)u->factor;
    float delta = x - u->prev;
    for (uint32_t i = 0; i < u->factor; ++i) out[i] = u->prev + delta * ((float)i * n_inv);
    u->prev = x;
}
float oo_linear_down_process(uint32_t factor, const float *xs) /* linear.rs:62-69 */
{
    float acc = 0.0f;
    for (uint32_t i = 0; i < factor; ++i) acc = acc + xs[i];
    return acc * (1.0f / (float)factor);
}
void oo_latch_up_process(uint32_t factor, float x, float *out) /* latch.rs:17-23 */
{
    for (uint32_t i = 0; i < factor; ++i) out[i] = x;
}
float oo_latch_down_process(uint32_t factor, const float *xs) /* latch.rs:44-47 */
{
    (void)factor;
    return xs[0];
}

/* ---- electric piano  examples/electric-piano/src/electric_piano_voice.rs */

#define EP_INTERPOLATION_STEPS 64u

static const float VELOCITY_0_SPECTRUM[OO_NUM_HARMONICS] = {0.02f, 0.05f}; /* :10-13, rest 0 */
static const float VELOCITY_127_SPECTRUM[OO_NUM_HARMONICS] = {
    /* :15-48 */
    0.150869f, 0.385766f, 0.215543f, 0.117811f, 0.100411f, 0.0128637f, 0.0288844f, 0.00243388f,
    0.00963092f, 0.0035634f, 0.00256945f, 0.00184799f, 0.000399878f, 0.000660576f, 3.00995e-05f,
    0.00021866f, 9.33705e-05f, 0.000177973f, 0.0002545f, 0.000323602f, 0.000779045f, 0.000116569f,
    0.000772873f, 0.000364486f, 0.000248027f, 0.00018236f, 3.27292e-05f, 6.64988e-05f, 0.0f, 0.0f,
    0.0f, 0.0f,
};

void oo_amplitude_source_new(oo_amplitude_source *a) /* :222-242 */
{
    memset(a, 0, sizeof *a);
    a->frequency = 440.0f;
    a->brightness = 30.0f;
    a->velocity_scaling = 50.0f;
    a->decay_rate = 90.0f;
    a->harmonic_decay = 70.0f;
    a->key_scaling = 50.0f;
    a->release_rate = 40.0f;
    a->released = 0;
    a->note_pitch = 60.0f;
    a->velocity = 0.0f;
    a->interpolation_step = EP_INTERPOLATION_STEPS;
}

static void ep_get_decay(const oo_amplitude_source *a, float note, float *decay) /* :244-268 */
{
    float base_decay_rate = (100.0f - a->decay_rate) / 40000.0f;
    float harmonic_scaling = 1.0f - ((100.0f - a->harmonic_decay) / 200000.0f);
    float scaling_multiplier = (48.0f - note) / 12.0f;
    float key_scaling_factor = scaling_multiplier * (a->key_scaling * 0.02f);
    float adjusted_decay;
    if (key_scaling_factor > 0.0f)
        adjusted_decay = 1.0f - (base_decay_rate / (1.0f + key_scaling_factor));
    else
        adjusted_decay = 1.0f - (base_decay_rate * (1.0f - key_scaling_factor));
    float scaling = 1.0f;
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) {
        decay[i] = adjusted_decay * scaling;
        scaling *= harmonic_scaling;
    }
}

static void ep_trigger_note(oo_amplitude_source *a, float velocity) /* :292-299, :270-290 */
{
    a->velocity = velocity;
    ep_get_decay(a, a->note_pitch, a->decay);
    /* get_release :270-274 */
    float release_value = 0.999f - ((100.0f - a->release_rate) / 1000.0f);
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) a->release[i] = release_value;
    /* get_initial_amplitudes :276-290 */
    float amplitudes[OO_NUM_HARMONICS];
    for (int i = 0; i < OO_NUM_HARMONICS; ++i)
        amplitudes[i] = (VELOCITY_127_SPECTRUM[i] * velocity) + (VELOCITY_0_SPECTRUM[i] * (1.0f - velocity));
    float brightness_scaling = -0.2f + (0.8f * (a->brightness * 0.01f));
    brightness_scaling += velocity * a->velocity_scaling * 0.01f * 0.5f;
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) amplitudes[i] *= 1.0f + brightness_scaling * (float)i;
    memcpy(a->current_value, amplitudes, sizeof amplitudes);
    a->released = 0;
    a->interpolation_step = 0;
}

void oo_amplitude_source_process_event_inputs(oo_amplitude_source *a) /* on_gate :308-318 */
{
    oo_queue tmp = a->gate;
    for (uint32_t i = 0; i < tmp.len; ++i) {
        const oo_event *ev = &tmp.ev[i];
        if (!ev->is_object && ev->scalar > 0.0f) {
            ep_trigger_note(a, ev->scalar);
        } else { /* release_note :301-304 */
            a->released = 1;
            a->interpolation_step = 0;
        }
    }
}

void oo_amplitude_source_process(oo_amplitude_source *a) /* :321-351 */
{
    if (a->interpolation_step == 0) {
        const float *multiplier = a->released ? a->release : a->decay;
        for (int i = 0; i < OO_NUM_HARMONICS; ++i) a->target_value[i] = a->current_value[i] * multiplier[i];
    }
    if (a->interpolation_step < EP_INTERPOLATION_STEPS) {
        float t = (float)(a->interpolation_step + 1) / (float)EP_INTERPOLATION_STEPS;
        for (int i = 0; i < OO_NUM_HARMONICS; ++i)
            a->current_value[i] = a->current_value[i] * (1.0f - t) + a->target_value[i] * t;
        a->interpolation_step += 1;
    } else {
        memcpy(a->current_value, a->target_value, sizeof a->current_value);
        a->interpolation_step = 0;
    }
    memcpy(a->amplitudes, a->current_value, sizeof a->amplitudes);
}

void oo_oscillator_bank_new(oo_oscillator_bank *b) /* :102-113 */
{
    memset(b, 0, sizeof *b);
    b->frequency = 440.0f;
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) {
        b->osc_re[i] = 1.0f;
        b->osc_im[i] = 0.0f;
        b->mul_re[i] = 1.0f;
        b->mul_im[i] = 0.0f;
    }
    b->last_frequency = 0.0f;
    b->sample_rate = OO_DEFAULT_SR;
}

void oo_oscillator_bank_process_event_inputs(oo_oscillator_bank *b) /* on_gate :115-122 */
{
    oo_queue tmp = b->gate;
    for (uint32_t i = 0; i < tmp.len; ++i) {
        const oo_event *ev = &tmp.ev[i];
        if (!ev->is_object && ev->scalar > 0.0f) {
            for (int h = 0; h < OO_NUM_HARMONICS; ++h) {
                b->osc_re[h] = 1.0f;
                b->osc_im[h] = 0.0f;
            }
        }
    }
}

static void ep_update_multipliers(oo_oscillator_bank *b, float note_frequency) /* :126-150 */
{
    if (fabsf(b->last_frequency - note_frequency) < 0.01f) return;
    b->last_frequency = note_frequency;
    float nyquist = b->sample_rate * 0.5f;
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) {
        float harmonic_num = (float)(i + 1);
        float harmonic_freq = note_frequency * harmonic_num;
        if (harmonic_freq < nyquist) {
            float angle = 2.0f * F32_PI * harmonic_freq / b->sample_rate;
            b->mul_re[i] = cosf(angle);
            b->mul_im[i] = sinf(angle);
        } else {
            b->mul_re[i] = 1.0f;
            b->mul_im[i] = 0.0f;
        }
    }
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) {
        b->osc_re[i] = 1.0f;
        b->osc_im[i] = 0.0f;
    }
}

void oo_oscillator_bank_process(oo_oscillator_bank *b) /* :154-169 */
{
    if (b->frequency > 0.0f) ep_update_multipliers(b, b->frequency);
    float sum = 0.0f;
    for (int i = 0; i < OO_NUM_HARMONICS; ++i) {
        /* Complex::mul :66-72 */
        float new_real = b->osc_re[i] * b->mul_re[i] - b->osc_im[i] * b->mul_im[i];
        float new_imag = b->osc_re[i] * b->mul_im[i] + b->osc_im[i] * b->mul_re[i];
        b->osc_re[i] = new_real;
        b->osc_im[i] = new_imag;
        sum += b->osc_im[i] * b->amplitudes[i];
    }
    b->output = sum * 3.0f;
}

void oo_tremolo_new(oo_tremolo *t) /* tremolo.rs:27-37 */
{
    memset(t, 0, sizeof *t);
    t->rate = 5.0f;
    t->depth = 0.5f;
    t->sample_rate = OO_DEFAULT_SR;
}

void oo_tremolo_process(oo_tremolo *t) /* tremolo.rs:40-62 */
{
    float input = t->input;
    float rate = t->rate;
    float depth = t->depth;
    float lfo = sinf(t->phase * 2.0f * F32_PI);
    float scaled_depth = depth / 3.0f;
    float pan = 0.5f + lfo * scaled_depth;
    t->output[0] = input * pan;
    t->output[1] = input * (1.0f - pan);
    float phase_increment = rate / t->sample_rate;
    t->phase = rs_fract(t->phase + phase_increment);
}

/* ---- MIDI contract  midi.rs:69-72, 147-171 ----------------------------- */

float oo_midi_note_to_freq(uint8_t note)
{
    float semitone_offset = (float)note - 69.0f;
    return 440.0f * powf(2.0f, semitone_offset / 12.0f);
}
float oo_midi_velocity_to_gate(uint8_t velocity) { return rs_clamp((float)velocity / 127.0f, 0.0f, 1.0f); }

/* ======================================================================== */
/* FMVoice  examples/fm-synth/src/fm_voice.rs:6-156                          */
/* ======================================================================== */

void oo_fm_voice_new(oo_fm_voice *v) /* Graph::new codegen/mod.rs:1309-1328 */
{
    memset(v, 0, sizeof *v);
    v->frequency = 440.0f; /* defaults fm_voice.rs:10-48 */
    v->op3_ratio = 3.0f; v->op3_level = 0.5f; v->op3_feedback = 0.0f;
    v->op3_attack = 0.01f; v->op3_decay = 0.1f; v->op3_sustain = 0.7f; v->op3_release = 0.3f;
    v->op2_ratio = 2.0f; v->op2_level = 0.5f; v->op2_feedback = 0.0f;
    v->op2_attack = 0.01f; v->op2_decay = 0.1f; v->op2_sustain = 0.7f; v->op2_release = 0.3f;
    v->op1_ratio = 1.0f;
    v->op1_attack = 0.01f; v->op1_decay = 0.2f; v->op1_sustain = 0.8f; v->op1_release = 0.5f;
    v->route = 0.0f;
    v->filter_cutoff = 2000.0f; v->filter_resonance = 0.707f;
    v->filter_attack = 0.01f; v->filter_decay = 0.2f; v->filter_sustain = 0.5f; v->filter_release = 0.3f;
    v->filter_env_amount = 0.0f;
    /* nodes fm_voice.rs:52-79 */
    oo_adsr_new(&v->env3, 0.01f, 0.1f, 0.7f, 0.3f);
    oo_adsr_new(&v->env2, 0.01f, 0.1f, 0.7f, 0.3f);
    oo_adsr_new(&v->env1, 0.01f, 0.2f, 0.8f, 0.5f);
    oo_adsr_new(&v->env_filter, 0.01f, 0.2f, 0.5f, 0.3f);
    v->filter_env_gain.gain = 0.0f;
    v->cutoff_mod.value = 2000.0f;
    oo_fm_operator_new(&v->op3_osc);
    oo_fm_operator_new(&v->op2_osc);
    oo_fm_operator_new(&v->op1_osc);
    oo_tpt_new(&v->filter, 2000.0f, 0.707f, 1);
    v->output_gain.gain = 0.3f;
    v->sample_rate = OO_DEFAULT_SR;
}

void oo_fm_voice_init(oo_fm_voice *v, float sr) /* init = set_sample_rate + prepare, codegen/mod.rs:1335-1382 */
{
    v->sample_rate = sr;
    v->env3.sample_rate = v->env2.sample_rate = v->env1.sample_rate = v->env_filter.sample_rate = sr;
    v->op3_osc.sample_rate = v->op2_osc.sample_rate = v->op1_osc.sample_rate = sr;
    v->filter.sample_rate = sr;
    oo_adsr_prepare(&v->env3);
    oo_adsr_prepare(&v->env2);
    oo_adsr_prepare(&v->env1);
    oo_adsr_prepare(&v->env_filter);
    oo_tpt_prepare(&v->filter);
}

/* inherent process(): codegen/mod.rs:539-573; per node: incoming edge copies
 * (emit_node.rs:191-362), process_event_inputs, process (emit_node.rs:365-379);
 * then output assignments; then clear graph-level event queues. */
void oo_fm_voice_process(oo_fm_voice *v)
{
    /* env3 */
    oo_queue_connect(&v->gate, &v->env3.gate);
    v->env3.attack = v->op3_attack; v->env3.decay = v->op3_decay;
    v->env3.sustain = v->op3_sustain; v->env3.release = v->op3_release;
    oo_adsr_process_event_inputs(&v->env3);
    oo_adsr_process(&v->env3);
    /* env2 */
    oo_queue_connect(&v->gate, &v->env2.gate);
    v->env2.attack = v->op2_attack; v->env2.decay = v->op2_decay;
    v->env2.sustain = v->op2_sustain; v->env2.release = v->op2_release;
    oo_adsr_process_event_inputs(&v->env2);
    oo_adsr_process(&v->env2);
    /* env1 */
    oo_queue_connect(&v->gate, &v->env1.gate);
    v->env1.attack = v->op1_attack; v->env1.decay = v->op1_decay;
    v->env1.sustain = v->op1_sustain; v->env1.release = v->op1_release;
    oo_adsr_process_event_inputs(&v->env1);
    oo_adsr_process(&v->env1);
    /* env_filter */
    oo_queue_connect(&v->gate, &v->env_filter.gate);
    v->env_filter.attack = v->filter_attack; v->env_filter.decay = v->filter_decay;
    v->env_filter.sustain = v->filter_sustain; v->env_filter.release = v->filter_release;
    oo_adsr_process_event_inputs(&v->env_filter);
    oo_adsr_process(&v->env_filter);
    /* filter_env_gain */
    v->filter_env_gain.input = v->env_filter.output;
    v->filter_env_gain.gain = v->filter_env_amount;
    oo_gain_process(&v->filter_env_gain);
    /* cutoff_mod */
    v->cutoff_mod.input = v->filter_env_gain.output;
    v->cutoff_mod.value = v->filter_cutoff;
    oo_add_value_process(&v->cutoff_mod);
    /* op3_osc */
    v->op3_osc.base_freq = v->frequency;
    v->op3_osc.ratio = v->op3_ratio;
    v->op3_osc.feedback = v->op3_feedback;
    v->op3_osc.envelope = v->env3.output;
    v->op3_osc.level = v->op3_level;
    oo_fm_operator_process(&v->op3_osc);
    /* op3_route */
    v->op3_route.input = v->op3_osc.output;
    v->op3_route.mix = v->route;
    oo_crossfade_process(&v->op3_route);
    /* op2_osc */
    v->op2_osc.phase_mod = v->op3_route.output_a;
    v->op2_osc.base_freq = v->frequency;
    v->op2_osc.ratio = v->op2_ratio;
    v->op2_osc.feedback = v->op2_feedback;
    v->op2_osc.envelope = v->env2.output;
    v->op2_osc.level = v->op2_level;
    oo_fm_operator_process(&v->op2_osc);
    /* op1_mod_mixer */
    v->op1_mod_mixer.input_a = v->op2_osc.output;
    v->op1_mod_mixer.input_b = v->op3_route.output_b;
    oo_mixer_process(&v->op1_mod_mixer);
    /* op1_osc (level, feedback unconnected: stay 1.0 / 0.0) */
    v->op1_osc.phase_mod = v->op1_mod_mixer.output;
    v->op1_osc.base_freq = v->frequency;
    v->op1_osc.ratio = v->op1_ratio;
    v->op1_osc.envelope = v->env1.output;
    oo_fm_operator_process(&v->op1_osc);
    /* filter */
    v->filter.cutoff = v->cutoff_mod.output;
    v->filter.input[0] = v->op1_osc.output;
    v->filter.q = v->filter_resonance;
    oo_tpt_process(&v->filter);
    /* output_gain (gain unconnected: stays 0.3) */
    v->output_gain.input = v->filter.output[0];
    oo_gain_process(&v->output_gain);
    /* graph outputs */
    v->audio_out = v->output_gain.output;
    /* clear graph-level event queues */
    oo_queue_clear(&v->gate);
}

/* ======================================================================== */
/* "osc+env+TptFilter" voice  oscen-lib/perf/profile_graph.rs:12-37          */
/* ======================================================================== */
typedef struct {
    float frequency;
    oo_queue gate;
    float cutoff, q;
    float audio;
    oo_polyblep osc;
    oo_tpt filter;
    oo_adsr envelope;
} oo_sub_voice;

static void sub_voice_new(oo_sub_voice *v)
{
    memset(v, 0, sizeof *v);
    v->frequency = 440.0f;
    v->cutoff = 3000.0f;
    v->q = 0.707f;
    oo_polyblep_new(&v->osc, 440.0f, 0.6f, OO_PB_SAW);
    oo_tpt_new(&v->filter, 3000.0f, 0.707f, 1);
    oo_adsr_new(&v->envelope, 0.01f, 0.1f, 0.7f, 0.2f);
}
static void sub_voice_init(oo_sub_voice *v, float sr)
{
    v->osc.sample_rate = sr;
    v->filter.sample_rate = sr;
    v->envelope.sample_rate = sr;
    oo_tpt_prepare(&v->filter);
    oo_adsr_prepare(&v->envelope);
}
static void sub_voice_process(oo_sub_voice *v)
{
    v->osc.frequency = v->frequency;
    oo_polyblep_process(&v->osc);
    v->filter.cutoff = v->cutoff;
    v->filter.q = v->q;
    v->filter.input[0] = v->osc.output;
    oo_tpt_process(&v->filter);
    oo_queue_connect(&v->gate, &v->envelope.gate);
    oo_adsr_process_event_inputs(&v->envelope);
    oo_adsr_process(&v->envelope);
    /* compound source `filter.output * envelope.output -> audio` (emit_node.rs:463-478) */
    v->audio = v->filter.output[0] * v->envelope.output;
    oo_queue_clear(&v->gate);
}

/* ======================================================================== */
/* ElectricPianoVoiceNode  electric_piano_voice.rs:362-402                   */
/* ======================================================================== */
typedef struct {
    float frequency;
    oo_queue gate;
    float brightness, velocity_scaling, decay_rate, harmonic_decay, key_scaling, release_rate;
    float output;
    oo_amplitude_source amplitude_source;
    oo_oscillator_bank oscillator_bank;
} oo_epiano_voice;

static void epiano_voice_new(oo_epiano_voice *v)
{
    memset(v, 0, sizeof *v);
    v->frequency = 440.0f;
    v->brightness = 30.0f;
    v->velocity_scaling = 50.0f;
    v->decay_rate = 90.0f;
    v->harmonic_decay = 70.0f;
    v->key_scaling = 50.0f;
    v->release_rate = 40.0f;
    oo_amplitude_source_new(&v->amplitude_source);
    oo_oscillator_bank_new(&v->oscillator_bank);
}
static void epiano_voice_init(oo_epiano_voice *v, float sr) { v->oscillator_bank.sample_rate = sr; }
static void epiano_voice_process(oo_epiano_voice *v)
{
    oo_amplitude_source *a = &v->amplitude_source;
    a->frequency = v->frequency;
    oo_queue_connect(&v->gate, &a->gate);
    a->brightness = v->brightness;
    a->velocity_scaling = v->velocity_scaling;
    a->decay_rate = v->decay_rate;
    a->harmonic_decay = v->harmonic_decay;
    a->key_scaling = v->key_scaling;
    a->release_rate = v->release_rate;
    oo_amplitude_source_process_event_inputs(a);
    oo_amplitude_source_process(a);
    oo_oscillator_bank *b = &v->oscillator_bank;
    b->frequency = v->frequency;
    oo_queue_connect(&v->gate, &b->gate);
    memcpy(b->amplitudes, a->amplitudes, sizeof b->amplitudes);
    oo_oscillator_bank_process_event_inputs(b);
    oo_oscillator_bank_process(b);
    v->output = b->output;
    oo_queue_clear(&v->gate);
}

/* ======================================================================== */
/* SatGraph_{1,4}x  oversampled-saturator/src/main.rs:64-80                  */
/* multirate body: emit_frame.rs:114-176                                     */
/* ======================================================================== */
typedef struct {
    float frequency; /* per-voice osc frequency (the reference fixes 2000 Hz) */
    float audio_out;
    oo_polyblep osc;
    oo_hardclip clip;
    oo_sinc_down down; /* [sinc] clip.output -> audio_out : Down{4,Sinc} */
    uint32_t factor;
} oo_sat_voice;

static void sat_voice_new(oo_sat_voice *v, uint32_t factor)
{
    memset(v, 0, sizeof *v);
    v->frequency = 2000.0f;
    oo_polyblep_new(&v->osc, 2000.0f, 0.6f, OO_PB_SAW);
    v->factor = factor;
    oo_sinc_down_new(&v->down, factor);
}
static void sat_voice_init(oo_sat_voice *v, float sr)
{
    /* nodes declared `* N` get sample_rate * N (emit_struct.rs:575-587) */
    v->osc.sample_rate = sr * (float)v->factor;
    oo_sinc_down_new(&v->down, v->factor); /* prepare() resets every resampler (emit_struct.rs:500-531) */
}
static void sat_voice_process(oo_sat_voice *v)
{
    if (v->factor <= 1) { /* same-rate body */
        v->osc.frequency = v->frequency;
        oo_polyblep_process(&v->osc);
        v->clip.input = v->osc.output;
        oo_hardclip_process(&v->clip);
        v->audio_out = v->clip.output;
        return;
    }
    float buf[8] = {0};
    for (uint32_t inner = 0; inner < v->factor; ++inner) {
        v->osc.frequency = v->frequency;
        oo_polyblep_process(&v->osc);
        v->clip.input = v->osc.output;
        oo_hardclip_process(&v->clip);
        buf[inner] = v->clip.output;
    }
    v->audio_out = oo_sinc_down_process(&v->down, buf);
}

/* ======================================================================== */
/* Voice bank (poly wrapper)                                                 */
/* ======================================================================== */

#define OO_MAX_BANK_PARAMS 32

typedef struct {
    uint32_t voice, frame_offset;
    int32_t kind;
    float value;
} oo_bank_event;

struct oo_bank {
    int kind;
    uint32_t n;
    float sample_rate;
    /* broadcast value inputs; `ramped[i]` when the wrapper declares [ramp: N] */
    uint32_t n_params;
    int ramped[OO_MAX_BANK_PARAMS];
    float plain[OO_MAX_BANK_PARAMS];
    oo_ramped_input ramp[OO_MAX_BANK_PARAMS];
    uint32_t active_ramps;
    /* per-voice handler outputs */
    float *voice_freq;
    oo_queue *voice_gate; /* MidiVoiceHandler.gate EventOutput per voice */
    /* voices */
    oo_fm_voice *fm;
    oo_sub_voice *sub;
    oo_epiano_voice *ep;
    oo_sat_voice *sat;
    /* post-mix */
    oo_tremolo tremolo;
    float vibrato_intensity, vibrato_speed;
    /* staged events for the next block */
    oo_bank_event *staged;
    uint32_t n_staged, cap_staged;
    /* output */
    float out[2];
    double last_bus_f64[OO_MAX_BLOCK];
    double last_abs_f64[OO_MAX_BLOCK]; /* sum of |voice output|: the scale of the bus-sum tolerance */
};

static const float FM_DEFAULTS[OO_FM_NUM_PARAMS] = {
    3.0f, 0.5f, 0.0f, 0.01f, 0.1f, 0.7f, 0.3f, /* op3 */
    2.0f, 0.5f, 0.0f, 0.01f, 0.1f, 0.7f, 0.3f, /* op2 */
    1.0f, 0.01f, 0.2f, 0.8f, 0.5f,             /* op1 */
    0.0f,                                      /* route */
    2000.0f, 0.707f, 0.01f, 0.2f, 0.5f, 0.3f, 0.0f,
};
/* [ramp: 2205] inputs of FMGraph  examples/fm-synth/src/lib.rs:31-64 */
static const int FM_RAMPED[OO_FM_NUM_PARAMS] = {
    0, 1, 1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 1,
};

oo_bank *oo_bank_create(int kind, uint32_t n)
{
    oo_bank *b = (oo_bank *)calloc(1, sizeof *b);
    if (!b) return NULL;
    b->kind = kind;
    b->n = n;
    b->sample_rate = OO_DEFAULT_SR;
    b->voice_freq = (float *)calloc(n, sizeof(float));
    b->voice_gate = (oo_queue *)calloc(n, sizeof(oo_queue));
    for (uint32_t i = 0; i < n; ++i) b->voice_freq[i] = 440.0f; /* MidiVoiceHandler::new midi.rs:58-67 */
    switch (kind) {
    case OO_BANK_FM:
        b->n_params = OO_FM_NUM_PARAMS;
        for (uint32_t p = 0; p < b->n_params; ++p) {
            b->ramped[p] = FM_RAMPED[p];
            b->plain[p] = FM_DEFAULTS[p];
            oo_ramp_new(&b->ramp[p].r, FM_DEFAULTS[p]);
            b->ramp[p].default_frames = 2205;
        }
        b->fm = (oo_fm_voice *)calloc(n, sizeof(oo_fm_voice));
        for (uint32_t i = 0; i < n; ++i) oo_fm_voice_new(&b->fm[i]);
        break;
    case OO_BANK_SUB:
        b->n_params = 2; /* cutoff, q  (PolySynth profile_graph.rs:40-66) */
        b->plain[0] = 3000.0f;
        b->plain[1] = 0.707f;
        b->sub = (oo_sub_voice *)calloc(n, sizeof(oo_sub_voice));
        for (uint32_t i = 0; i < n; ++i) sub_voice_new(&b->sub[i]);
        break;
    case OO_BANK_EPIANO: {
        /* electric-piano/src/main.rs:36-45: 6 voice params + vibrato_intensity, vibrato_speed */
        static const float d[8] = {30.0f, 50.0f, 90.0f, 70.0f, 50.0f, 40.0f, 0.3f, 5.0f};
        b->n_params = 8;
        for (int p = 0; p < 8; ++p) b->plain[p] = d[p];
        b->ep = (oo_epiano_voice *)calloc(n, sizeof(oo_epiano_voice));
        for (uint32_t i = 0; i < n; ++i) epiano_voice_new(&b->ep[i]);
        oo_tremolo_new(&b->tremolo);
        break;
    }
    case OO_BANK_SAT4X:
    case OO_BANK_SAT1X:
        b->n_params = 0;
        b->sat = (oo_sat_voice *)calloc(n, sizeof(oo_sat_voice));
        for (uint32_t i = 0; i < n; ++i) {
            sat_voice_new(&b->sat[i], kind == OO_BANK_SAT4X ? 4 : 1);
            b->voice_freq[i] = 2000.0f;
        }
        break;
    default:
        break;
    }
    return b;
}

void oo_bank_destroy(oo_bank *b)
{
    if (!b) return;
    free(b->voice_freq);
    free(b->voice_gate);
    free(b->fm);
    free(b->sub);
    free(b->ep);
    free(b->sat);
    free(b->staged);
    free(b);
}

void oo_bank_init(oo_bank *b, float sr)
{
    b->sample_rate = sr;
    for (uint32_t i = 0; i < b->n; ++i) {
        if (b->fm) oo_fm_voice_init(&b->fm[i], sr);
        if (b->sub) sub_voice_init(&b->sub[i], sr);
        if (b->ep) epiano_voice_init(&b->ep[i], sr);
        if (b->sat) sat_voice_init(&b->sat[i], sr);
    }
    b->tremolo.sample_rate = sr;
}

uint32_t oo_bank_num_params(const oo_bank *b) { return b->n_params; }
uint32_t oo_bank_channels(const oo_bank *b) { return b->kind == OO_BANK_EPIANO ? 2u : 1u; }

int oo_bank_set_value(oo_bank *b, uint32_t p, float v)
{
    if (p >= b->n_params) return -1;
    if (b->ramped[p]) oo_ramped_set(&b->ramp[p], &b->active_ramps, v);
    else b->plain[p] = v;
    return 0;
}
int oo_bank_set_value_with_ramp(oo_bank *b, uint32_t p, float v, uint32_t frames)
{
    if (p >= b->n_params) return -1;
    if (b->ramped[p]) oo_ramped_set_with_ramp(&b->ramp[p], &b->active_ramps, v, frames);
    else b->plain[p] = v;
    return 0;
}
int oo_bank_set_value_immediate(oo_bank *b, uint32_t p, float v)
{
    if (p >= b->n_params) return -1;
    if (b->ramped[p]) oo_ramped_set_immediate(&b->ramp[p], &b->active_ramps, v);
    else b->plain[p] = v;
    return 0;
}
void oo_bank_set_voice_frequency(oo_bank *b, uint32_t voice, float hz)
{
    if (voice < b->n) b->voice_freq[voice] = hz;
}

int oo_bank_push_event(oo_bank *b, uint32_t voice, uint32_t frame_offset, int kind, float value)
{
    if (voice >= b->n) return -1;
    if (b->n_staged == b->cap_staged) {
        uint32_t nc = b->cap_staged ? b->cap_staged * 2 : 1024;
        oo_bank_event *ne = (oo_bank_event *)realloc(b->staged, nc * sizeof *ne);
        if (!ne) return -1;
        b->staged = ne;
        b->cap_staged = nc;
    }
    oo_bank_event e = {voice, frame_offset, kind, value};
    b->staged[b->n_staged++] = e;
    return 0;
}

static inline float bank_param(const oo_bank *b, uint32_t p)
{ /* ramped inputs are read through `.current` (emit_node.rs:317-321) */
    return b->ramped[p] ? b->ramp[p].r.current : b->plain[p];
}

/* tick_ramps  codegen/mod.rs:878-914 */
static void bank_tick_ramps(oo_bank *b)
{
    if (b->active_ramps > 0) {
        for (uint32_t p = 0; p < b->n_params; ++p)
            if (b->ramped[p] && oo_ramp_tick(&b->ramp[p].r)) b->active_ramps -= 1;
    }
}

/* one frame of the wrapper graph: emit_frame.rs:29-69; voices iterated
 * voice-minor (emit_node.rs:365-379); output = sequential f32 sum in voice
 * order (emit_node.rs:463-466 / emit_edge.rs:68-84). */
static void bank_advance_one_frame(oo_bank *b, uint32_t frame, const uint32_t *tap_voices,
                                   uint32_t n_taps, float *taps, uint32_t tap_stride)
{
    bank_tick_ramps(b);
    float sum = 0.0f;
    double sum64 = 0.0, abs64 = 0.0;
    for (uint32_t i = 0; i < b->n; ++i) {
        float y = 0.0f;
        switch (b->kind) {
        case OO_BANK_FM: {
            oo_fm_voice *v = &b->fm[i];
            v->frequency = b->voice_freq[i];
            oo_queue_connect(&b->voice_gate[i], &v->gate);
            v->op3_ratio = bank_param(b, OO_FM_OP3_RATIO);
            v->op3_level = bank_param(b, OO_FM_OP3_LEVEL);
            v->op3_feedback = bank_param(b, OO_FM_OP3_FEEDBACK);
            v->op3_attack = bank_param(b, OO_FM_OP3_ATTACK);
            v->op3_decay = bank_param(b, OO_FM_OP3_DECAY);
            v->op3_sustain = bank_param(b, OO_FM_OP3_SUSTAIN);
            v->op3_release = bank_param(b, OO_FM_OP3_RELEASE);
            v->op2_ratio = bank_param(b, OO_FM_OP2_RATIO);
            v->op2_level = bank_param(b, OO_FM_OP2_LEVEL);
            v->op2_feedback = bank_param(b, OO_FM_OP2_FEEDBACK);
            v->op2_attack = bank_param(b, OO_FM_OP2_ATTACK);
            v->op2_decay = bank_param(b, OO_FM_OP2_DECAY);
            v->op2_sustain = bank_param(b, OO_FM_OP2_SUSTAIN);
            v->op2_release = bank_param(b, OO_FM_OP2_RELEASE);
            v->op1_ratio = bank_param(b, OO_FM_OP1_RATIO);
            v->op1_attack = bank_param(b, OO_FM_OP1_ATTACK);
            v->op1_decay = bank_param(b, OO_FM_OP1_DECAY);
            v->op1_sustain = bank_param(b, OO_FM_OP1_SUSTAIN);
            v->op1_release = bank_param(b, OO_FM_OP1_RELEASE);
            v->route = bank_param(b, OO_FM_ROUTE);
            v->filter_cutoff = bank_param(b, OO_FM_FILTER_CUTOFF);
            v->filter_resonance = bank_param(b, OO_FM_FILTER_RESONANCE);
            v->filter_attack = bank_param(b, OO_FM_FILTER_ATTACK);
            v->filter_decay = bank_param(b, OO_FM_FILTER_DECAY);
            v->filter_sustain = bank_param(b, OO_FM_FILTER_SUSTAIN);
            v->filter_release = bank_param(b, OO_FM_FILTER_RELEASE);
            v->filter_env_amount = bank_param(b, OO_FM_FILTER_ENV_AMOUNT);
            oo_fm_voice_process(v);
            y = v->audio_out;
            break;
        }
        case OO_BANK_SUB: {
            oo_sub_voice *v = &b->sub[i];
            v->frequency = b->voice_freq[i];
            oo_queue_connect(&b->voice_gate[i], &v->gate);
            v->cutoff = bank_param(b, 0);
            v->q = bank_param(b, 1);
            sub_voice_process(v);
            y = v->audio;
            break;
        }
        case OO_BANK_EPIANO: {
            oo_epiano_voice *v = &b->ep[i];
            v->frequency = b->voice_freq[i];
            oo_queue_connect(&b->voice_gate[i], &v->gate);
            v->brightness = bank_param(b, 0);
            v->velocity_scaling = bank_param(b, 1);
            v->decay_rate = bank_param(b, 2);
            v->harmonic_decay = bank_param(b, 3);
            v->key_scaling = bank_param(b, 4);
            v->release_rate = bank_param(b, 5);
            epiano_voice_process(v);
            y = v->output;
            break;
        }
        default: {
            oo_sat_voice *v = &b->sat[i];
            v->frequency = b->voice_freq[i];
            sat_voice_process(v);
            y = v->audio_out;
            break;
        }
        }
        sum += y;
        sum64 += (double)y;
        abs64 += fabs((double)y);
        for (uint32_t t = 0; t < n_taps; ++t)
            if (tap_voices[t] == i) taps[(size_t)t * tap_stride + frame] = y;
    }
    b->last_bus_f64[frame % OO_MAX_BLOCK] = sum64;
    b->last_abs_f64[frame % OO_MAX_BLOCK] = abs64;
    if (b->kind == OO_BANK_EPIANO) {
        b->tremolo.input = sum;
        b->tremolo.depth = bank_param(b, 6);
        b->tremolo.rate = bank_param(b, 7);
        oo_tremolo_process(&b->tremolo);
        b->out[0] = b->tremolo.output[0];
        b->out[1] = b->tremolo.output[1];
    } else {
        b->out[0] = sum;
    }
    /* handler EventOutputs are cleared by their next process_event_inputs
     * (clear_event_outputs, oscen-macros/src/lib.rs:234-263) */
    for (uint32_t i = 0; i < b->n; ++i) oo_queue_clear(&b->voice_gate[i]);
}

/* deliver the staged events whose frame_offset == frame: the handler's
 * on_note_on sets current_frequency + pushes the gate (midi.rs:91-121) and
 * its process() publishes frequency on that same frame (midi.rs:80-88). */
static void bank_deliver(oo_bank *b, const oo_bank_event *ev, uint32_t n_ev, uint32_t *cursor, uint32_t frame)
{
    while (*cursor < n_ev && ev[*cursor].frame_offset == frame) {
        const oo_bank_event *e = &ev[*cursor];
        if (e->kind == OO_EV_FREQ) {
            b->voice_freq[e->voice] = e->value;
        } else {
            oo_event ge = {frame, e->value, 0};
            (void)oo_queue_try_push(&b->voice_gate[e->voice], ge);
        }
        *cursor += 1;
    }
}

/* stable insertion sort by frame_offset (the reference uses sort_unstable_by_key,
 * codegen/mod.rs:795; for <= 20 elements Rust's implementation is an insertion
 * sort, so equal-offset events keep push order) */
static void sort_events(oo_bank_event *ev, uint32_t n)
{
    for (uint32_t i = 1; i < n; ++i) {
        oo_bank_event k = ev[i];
        uint32_t j = i;
        while (j > 0 && ev[j - 1].frame_offset > k.frame_offset) {
            ev[j] = ev[j - 1];
            --j;
        }
        ev[j] = k;
    }
}

static void write_out(const oo_bank *b, float *out_bus, uint32_t frame)
{
    uint32_t ch = oo_bank_channels(b);
    for (uint32_t c = 0; c < ch; ++c) out_bus[(size_t)frame * ch + c] = b->out[c];
}

/* process_block with sub-block splitting  codegen/mod.rs:755-873 */
void oo_bank_process_block(oo_bank *b, uint32_t frames, float *out_bus, const uint32_t *tap_voices,
                           uint32_t n_taps, float *taps)
{
    uint32_t n_ev = b->n_staged;
    oo_bank_event *ev = b->staged;
    sort_events(ev, n_ev);
    b->n_staged = 0;
    uint32_t cursor = 0;
    uint32_t frame = 0;
    while (frame < frames) {
        uint32_t next_event = frames;
        if (cursor < n_ev) {
            uint32_t fo = ev[cursor].frame_offset;
            uint32_t cand = fo > frame ? fo : frame;
            if (cand < next_event) next_event = cand;
        }
        while (frame < next_event) {
            bank_advance_one_frame(b, frame, tap_voices, n_taps, taps, frames);
            write_out(b, out_bus, frame);
            frame += 1;
        }
        if (frame >= frames) break;
        bank_deliver(b, ev, n_ev, &cursor, frame);
        bank_advance_one_frame(b, frame, tap_voices, n_taps, taps, frames);
        write_out(b, out_bus, frame);
        frame += 1;
    }
}

/* N x process() with events pushed right before their frame */
void oo_bank_process_per_sample(oo_bank *b, uint32_t frames, float *out_bus, const uint32_t *tap_voices,
                                uint32_t n_taps, float *taps)
{
    uint32_t n_ev = b->n_staged;
    oo_bank_event *ev = b->staged;
    sort_events(ev, n_ev);
    b->n_staged = 0;
    uint32_t cursor = 0;
    for (uint32_t frame = 0; frame < frames; ++frame) {
        bank_deliver(b, ev, n_ev, &cursor, frame);
        bank_advance_one_frame(b, frame, tap_voices, n_taps, taps, frames);
        write_out(b, out_bus, frame);
    }
}

const double *oo_bank_last_bus_f64(const oo_bank *b) { return b->last_bus_f64; }
const double *oo_bank_last_abs_f64(const oo_bank *b) { return b->last_abs_f64; }

/* ======================================================================== */
/* bench graphs  oscen-lib/benches/static_vs_runtime.rs:5-66                 */
/* Neither graph declares outputs => dead-node removal is skipped            */
/* (ir/passes/dead_nodes.rs:17-19) and every node runs each process().       */
/* ======================================================================== */

void oo_static_simple_new(oo_static_simple *g)
{
    memset(g, 0, sizeof *g);
    oo_oscillator_new(&g->osc, 440.0f, 1.0f, OO_WAVE_SINE);
    oo_tpt_new(&g->filter, 1000.0f, 0.7f, 1);
    g->gain.gain = 0.5f;
}
void oo_static_simple_init(oo_static_simple *g, float sr)
{
    g->osc.sample_rate = sr;
    g->filter.sample_rate = sr;
    oo_tpt_prepare(&g->filter);
}
void oo_static_simple_process(oo_static_simple *g)
{
    oo_oscillator_process(&g->osc);
    g->filter.input[0] = g->osc.output;
    oo_tpt_process(&g->filter);
    g->gain.input = g->filter.output[0];
    oo_gain_process(&g->gain);
}

void oo_static_complex_new(oo_static_complex *g)
{
    memset(g, 0, sizeof *g);
    oo_polyblep_new(&g->osc1, 440.0f, 0.33f, OO_PB_SAW);
    oo_polyblep_new(&g->osc2, 442.0f, 0.33f, OO_PB_SAW);
    oo_polyblep_new(&g->osc3, 438.0f, 0.33f, OO_PB_SAW);
    g->mix1.gain = g->mix2.gain = g->mix3.gain = g->mixer.gain = 1.0f;
    oo_adsr_new(&g->filter_env, 0.01f, 0.3f, 0.5f, 0.2f);
    g->env_amount.gain = 2000.0f;
    oo_tpt_new(&g->filter, 800.0f, 0.7f, 1);
    oo_adsr_new(&g->amp_env, 0.01f, 0.2f, 0.7f, 0.3f);
    g->vca.gain = 1.0f;
}
void oo_static_complex_init(oo_static_complex *g, float sr)
{
    g->osc1.sample_rate = g->osc2.sample_rate = g->osc3.sample_rate = sr;
    g->filter_env.sample_rate = g->amp_env.sample_rate = sr;
    g->filter.sample_rate = sr;
    oo_adsr_prepare(&g->filter_env);
    oo_adsr_prepare(&g->amp_env);
    oo_tpt_prepare(&g->filter);
}
void oo_static_complex_process(oo_static_complex *g)
{
    oo_polyblep_process(&g->osc1);
    oo_polyblep_process(&g->osc2);
    oo_polyblep_process(&g->osc3);
    g->mix1.input = g->osc1.output; oo_gain_process(&g->mix1);
    g->mix2.input = g->osc2.output; oo_gain_process(&g->mix2);
    g->mix3.input = g->osc3.output; oo_gain_process(&g->mix3);
    g->mixer.input = g->mix1.output; oo_gain_process(&g->mixer);
    oo_adsr_process_event_inputs(&g->filter_env);
    oo_adsr_process(&g->filter_env);
    g->env_amount.input = g->filter_env.output; oo_gain_process(&g->env_amount);
    g->filter.input[0] = g->mixer.output;
    g->filter.f_mod = g->env_amount.output;
    oo_tpt_process(&g->filter);
    oo_adsr_process_event_inputs(&g->amp_env);
    oo_adsr_process(&g->amp_env);
    g->vca.input = g->filter.output[0];
    g->vca.gain = g->amp_env.output;
    oo_gain_process(&g->vca);
}

/* FM core cross-check  examples/fm-synth/src/waveform.rs:24-52 */
void oo_fm_compute_waveform(float op3_ratio, float op3_level, float op3_feedback, float op2_ratio,
                            float op2_level, float op2_feedback, float op1_ratio_unused, float route,
                            uint32_t n, float *out)
{
    (void)op1_ratio_unused;
    const uint32_t NUM_SAMPLES = 512, WARMUP = 2;
    uint32_t total = NUM_SAMPLES * (WARMUP + 1);
    float op3_prev = 0.0f, op2_prev = 0.0f;
    uint32_t w = 0;
    for (uint32_t i = 0; i < total; ++i) {
        float phase = (float)(i % NUM_SAMPLES) / (float)NUM_SAMPLES;
        float op3_total_phase = op3_ratio * phase + op3_feedback * op3_prev;
        float op3_out = sinf(op3_total_phase * F32_TAU) * op3_level;
        op3_prev = op3_out;
        float op3_to_2 = op3_out * (1.0f - route);
        float op3_to_1 = op3_out * route;
        float op2_total_phase = op2_ratio * phase + op3_to_2 + op2_feedback * op2_prev;
        float op2_out = sinf(op2_total_phase * F32_TAU) * op2_level;
        op2_prev = op2_out;
        float op1_mod = op2_out + op3_to_1;
        float op1_out = sinf((phase + op1_mod) * F32_TAU);
        if (i >= NUM_SAMPLES * WARMUP && w < n) out[w++] = op1_out;
    }
}

/* ======================================================================== */
/* synthetic note plan: splitmix64(seed ^ voice)              */
/* ======================================================================== */
static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void oo_note_plan_for_voice(uint64_t seed, uint32_t voice, oo_note_plan *p)
{
    uint64_t s = seed ^ (uint64_t)voice;
    p->note = (uint8_t)(36 + splitmix64(&s) % 61);        /* U{36..96}  */
    p->velocity = (uint8_t)(32 + splitmix64(&s) % 96);    /* U{32..127} */
    p->on_frame = (uint32_t)(splitmix64(&s) % 256);       /* U[0,255]   */
    p->off_frame = 12000u + (uint32_t)(splitmix64(&s) % 24001);   /* U[12000,36000] */
    p->retrig_frame = 36001u + (uint32_t)(splitmix64(&s) % 8000); /* U[36001,44000] */
    p->frequency = oo_midi_note_to_freq(p->note);
}

/* The note stream of one voice as a frame-sorted event list (at most 4 gate events).
 * fold 0 ("scale"): the plan of oo_note_plan_scaled -- on / off / retrigger compressed into `span` frames.
 * fold 1 ("slice"): the window [0, span) shows the slice [rot, rot + span) of the voice's cyclic 1 s plan, rot = a
 *   sixth splitmix64 draw % 48000: every event keeps the plan's REAL density (3 events per voice per 48000 frames)
 *   and all three kinds fall inside a short run, spread over different voices.  A voice whose note is sounding at the
 *   cut (on < rot <= off, or rot > retrig) also gets its note-on at the plan's own on-frame (0..255). */
void oo_note_events_for_voice(uint64_t seed, uint32_t voice, uint32_t span, int fold, oo_note_events *out)
{
    oo_note_plan p;
    out->n = 0;
    if (fold == 0 || span == 0 || span >= 48000u) {
        oo_note_plan_scaled(seed, voice, span, &p);
        const float vel = oo_midi_velocity_to_gate(p.velocity);
        out->frequency = p.frequency;
        out->frame[0] = p.on_frame, out->value[0] = vel;
        out->frame[1] = p.off_frame, out->value[1] = 0.0f;
        out->frame[2] = p.retrig_frame, out->value[2] = vel;
        out->n = 3;
        return;
    }
    oo_note_plan_for_voice(seed, voice, &p);
    uint64_t s = seed ^ (uint64_t)voice;
    for (int i = 0; i < 5; ++i) (void)splitmix64(&s);
    const uint32_t rot = (uint32_t)(splitmix64(&s) % 48000u);
    const float vel = oo_midi_velocity_to_gate(p.velocity);
    out->frequency = p.frequency;
    uint32_t fr[4];
    float va[4];
    uint32_t n = 0;
    const int held = (p.on_frame < rot && rot <= p.off_frame) || rot > p.retrig_frame;
    if (held) fr[n] = p.on_frame, va[n] = vel, ++n;
    const uint32_t t[3] = {p.on_frame, p.off_frame, p.retrig_frame};
    const float v[3] = {vel, 0.0f, vel};
    for (int i = 0; i < 3; ++i) fr[n] = (t[i] + 48000u - rot) % 48000u, va[n] = v[i], ++n;
    for (uint32_t i = 1; i < n; ++i) { /* stable insertion sort by frame */
        const uint32_t kf = fr[i];
        const float kv = va[i];
        uint32_t j = i;
        while (j > 0 && fr[j - 1] > kf) fr[j] = fr[j - 1], va[j] = va[j - 1], --j;
        fr[j] = kf, va[j] = kv;
    }
    for (uint32_t i = 0; i < n; ++i) out->frame[i] = fr[i], out->value[i] = va[i];
    out->n = n;
}

/* The same plan folded into a shorter window: every frame scaled by span / 48000 (integer
 * arithmetic), so that a run of `span` < 48000 frames still sees note-off and retrigger.
 * span == 0 or >= 48000: the plan as is. */
void oo_note_plan_scaled(uint64_t seed, uint32_t voice, uint32_t span, oo_note_plan *p)
{
    oo_note_plan_for_voice(seed, voice, p);
    if (span == 0 || span >= 48000u) return;
    p->on_frame = (uint32_t)((uint64_t)p->on_frame * span / 48000u);
    p->off_frame = (uint32_t)((uint64_t)p->off_frame * span / 48000u);
    p->retrig_frame = (uint32_t)((uint64_t)p->retrig_frame * span / 48000u);
}
