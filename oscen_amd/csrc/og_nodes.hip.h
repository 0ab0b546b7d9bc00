// og_nodes.hip.h -- hand-written gfx950 device bodies of the Oscen DSP nodes.
//
// One lane = one voice.  Every function works on scalar references so that a
// generated voice kernel keeps all per-voice state in VGPRs across the block.
// Arithmetic follows the reference's f32 operation order (the translation
// unit is compiled with -ffp-contract=off; the only fused ops are the explicit
// fmaf() inside og_math.h).  Parameter-derived coefficients that depend only
// on block-uniform values (ADSR step counts / one-pole coefficients, filter
// limits) are computed once per block by the host library and arrive as
// kernel-argument constants; what remains here is the per-sample, per-voice
// recurrence.  Each function cites the reference lines it implements.
#pragma once
#ifndef __HIPCC_RTC__ // hiprtc pre-includes the HIP runtime and the fixed-width types
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "og_kernel_rt.hip.h"
#include "og_math.h"

namespace og {

#define OG_DEV __device__ __forceinline__

constexpr float F32_EPSILON = 1.1920929e-7f;
constexpr float F32_TAU = 6.28318548202514648f;

// f32::clamp(lo, hi) with lo <= hi as ONE v_med3_f32 (the compare/select form
// costs two VALU ops plus VCC wait states).  Identical for every non-NaN x;
// the signal path never carries NaN (a NaN input clamps to lo here instead of
// propagating as Rust's clamp would).
OG_DEV float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
OG_DEV float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// f32::fract(x) / `x % 1.0`: x - trunc(x) is that value exactly; sign(x) * (|x| - floor(|x|)) is the same number in two
// instructions (v_fract_f32 with an |x| input modifier, v_bfi_b32) instead of three -- the subtraction inside v_fract is
// exact for a non-negative argument, the hardware's clamp below 1.0 never engages there.  (Differs from x - trunc(x) only
// in the sign of a zero result for negative whole x.)
OG_DEV float fract_keep_sign(float x)
{
#ifndef OG_HOSTSIM
    return __builtin_copysignf(__builtin_amdgcn_fractf(__builtin_fabsf(x)), x);
#else
    return x - truncf(x);
#endif
}
// f32::rem_euclid(1.0): r = x % 1.0; if r < 0 { r + 1.0 } else { r } -- ONE v_fract_f32 (x - floor(x), rounded once: r is
// exact, so r + 1.0 rounds to the same number).  The single difference: where r + 1.0 rounds UP to 1.0 (-6e-8 < x < 0) the
// reference returns 1.0 and v_fract its clamp 0.99999994.
OG_DEV float fract_floor(float x)
{
#ifndef OG_HOSTSIM
    return __builtin_amdgcn_fractf(x);
#else
    const float r = x - truncf(x);
    const float w = (r < 0.0f) ? r + 1.0f : r;
    return w >= 1.0f ? 0.99999994f : w;
#endif
}

// The phase accumulator of an FM operator: `self.phase = (self.phase + inc).fract()` (fm_operator.rs:68-70).  The phase only
// ever enters sin(2 pi (phase + mod)), which has period 1, so the representative in [0, 1) -- ONE v_fract_f32 -- gives the
// same output as the reference's sign-keeping fract; for inc >= 0 (every graph in scope) the two are the same number, bit
// for bit; for a negative increment the state word reads phase + 1 where the reference holds a negative phase.
// OG_STRICT: x - trunc(x).
OG_DEV float fract_phase(float p)
{
#ifdef OG_STRICT
    return p - truncf(p);
#else
    return fract_floor(p);
#endif
}

// a / b for finite a and b >= 1: v_rcp_f32 plus one Newton step on the
// quotient (q' = q + (a - q*b) * rcp(b)).  Correctly rounded except for rare
// 1-ulp misses; 5 VALU ops instead of the 12 of the IEEE expansion.  Only
// used where the reference's quotient feeds a contracting recurrence (the
// ADSR release slope, the TPT coefficient h), never for phase increments.
OG_DEV float div_near(float a, float b)
{
    const float rb = __builtin_amdgcn_rcpf(b);
    const float q = a * rb;
    const float r = fmaf(-q, b, a);
    return fmaf(r, rb, q);
}

// ---------------------------------------------------------------------------
// AdsrEnvelope  oscen-lib/src/envelope/adsr.rs
// ---------------------------------------------------------------------------
// Stage codes chosen so that the successor at a stage end is (stage + 1) & 7:
// Attack -> Decay -> Sustain, Release -> Idle.
enum : uint32_t { ST_IDLE = 0, ST_ATTACK = 1, ST_DECAY = 2, ST_SUSTAIN = 3, ST_RELEASE = 7 };

// Block-uniform, host-derived (adsr.rs:84-90, 117-134), eight consecutive
// uniform slots per envelope: stage lengths in samples, one-pole coefficients
// 1 - exp(-4.6051702/n), sustain clamped to [0,1], and the two
// "time <= MIN_TIME_SECONDS" flags (adsr.rs:259, 265).  Only d_n/a_c/d_c are
// needed every sample; the rest is read from the kernel arguments where it is
// used (block start, gate events) so it does not pin SGPRs across the loop.
enum : int { ADSR_A_N = 0, ADSR_D_N, ADSR_R_N, ADSR_A_C, ADSR_D_C, ADSR_SUSTAIN, ADSR_A_INST, ADSR_R_INST };

// Register form of one envelope.  (stage, cnt, lv) mirror the reference's
// (stage, samples_remaining, level); (tgt, cf) are the one-pole target and
// coefficient of the current stage, refreshed only when the stage changes, so
// the per-sample update is `lv += (tgt - lv) * cf` for every stage: holding
// stages and Release run it with cf = 0, which leaves lv bit-unchanged.  tgt
// is also the level a stage ends on (1 after Attack, sustain_level after
// Decay, 0 after Release).  vel/sus are only touched at block start and by
// gate events.  Holding stages keep cnt at ADSR_HOLD so the per-sample
// countdown never reaches the stage-end test for them.
//
// The voice kernels are bound by VALU issue / dependent-instruction latency
// (3-6 cycles per wave-instruction per SIMD, measured), so every instruction
// here is paid in full:
//  * the reference's per-sample clamps of `level` are provable no-ops for
//    lv, tgt in [0,1] and cf in [0,1] (round-to-nearest cannot carry
//    lv + (tgt-lv)*cf past tgt's side of [0,1], nor lv - lv/n below 0) and are
//    not re-executed;
//  * `if current <= 0 {0} else {-current/n}` is -current/n for current >= 0
//    (-0/n = -0, and x + -0 = x);
//  * a stage end (countdown reaching 0) is rare: the tick does not test for it.
//    One check per run of consecutive envelopes (adsr_complete, emitted right
//    after the run and before anything reads their outputs) puts the level and
//    this frame's output on the stage's target and does complete_stage();
//  * Release adds rs * (-lv / n) with rs = 1 in Release and 0 elsewhere, one
//    fma instead of a compare and a select (1*q + lv rounds like lv + q; n >= 1
//    while a stage is ticking, so q is finite and 0*q + lv == lv).
constexpr uint32_t ADSR_HOLD = 0xFFFFFFFFu;

struct Adsr {
    uint32_t stage, cnt;
    float lv, tgt, cf, vel, sus, rs;
    // (float)cnt, kept by the release arithmetic of a chunk so that it costs one subtraction per sample instead of an
    // integer step and a conversion: adsr_enter() sets it, adsr_tick<.., true> counts it down, and the generated chunk
    // loops re-derive it from cnt at their top (the release-free chunk variant steps cnt only).  Exact -- identical to
    // converting cnt every sample -- for stages shorter than 2^24 samples (5.8 minutes at 48 kHz); beyond that the
    // reciprocal it feeds differs in its last bits, in an increment that is itself below half an ulp of the level.
    float fc;
};

OG_DEV void adsr_enter(Adsr& e, uint32_t stage, uint32_t n, float a_c, float d_c)
{
    const bool att = stage == ST_ATTACK, dec = stage == ST_DECAY;
    e.stage = stage;
    e.cnt = (stage == ST_SUSTAIN || stage == ST_IDLE) ? ADSR_HOLD : n;
    e.tgt = att ? 1.0f : (dec ? e.sus : 0.0f);
    e.cf = att ? a_c : (dec ? d_c : 0.0f);
    e.rs = (stage == ST_RELEASE) ? 1.0f : 0.0f;
    e.fc = (float)e.cnt;
}

// Once per block: load + the part of apply_parameters()/update_sustain_level()
// that only changes when a parameter or the velocity changed -- sustain_level
// (adsr.rs:93), the samples_remaining clamp against the re-derived stage
// length (adsr.rs:95-106), and Sustain/Idle pinning the level (adsr.rs:241-246).
OG_DEV void adsr_block_begin(Adsr& e, uint32_t stage, uint32_t rem, float level, float vel, const OgBlockArgs& A, int k)
{
    // scalar_u(): keep the kernel-argument reads scalar; a lane-varying select
    // between them must select values, not addresses
    const uint32_t a_n = scalar_u(A, k + ADSR_A_N), d_n = scalar_u(A, k + ADSR_D_N), r_n = scalar_u(A, k + ADSR_R_N);
    const float a_c = __uint_as_float(scalar_u(A, k + ADSR_A_C)), d_c = __uint_as_float(scalar_u(A, k + ADSR_D_C));
    e.vel = vel;
    e.sus = clamp01(__uint_as_float(scalar_u(A, k + ADSR_SUSTAIN)) * vel);
    const uint32_t lim = (stage == ST_ATTACK) ? a_n : (stage == ST_DECAY) ? d_n : r_n;
    const bool moving = (stage == ST_ATTACK) | (stage == ST_DECAY) | (stage == ST_RELEASE);
    if (moving && rem > 0) rem = max(min(rem, lim), 1u);
    e.lv = (stage == ST_SUSTAIN) ? e.sus : ((stage == ST_IDLE) ? 0.0f : clamp01(level));
    adsr_enter(e, stage, rem, a_c, d_c);
}

// what the state planes hold at block end
OG_DEV uint32_t adsr_rem(const Adsr& e) { return (e.stage == ST_SUSTAIN || e.stage == ST_IDLE) ? 0u : e.cnt; }

// handle_gate_event  adsr.rs:250-273 (scalar payload)
OG_DEV void adsr_gate(Adsr& e, float v, const OgBlockArgs& A, int k)
{
    const uint32_t a_n = scalar_u(A, k + ADSR_A_N), d_n = scalar_u(A, k + ADSR_D_N), r_n = scalar_u(A, k + ADSR_R_N);
    const uint32_t a_inst = scalar_u(A, k + ADSR_A_INST), r_inst = scalar_u(A, k + ADSR_R_INST);
    const float a_c = __uint_as_float(scalar_u(A, k + ADSR_A_C)), d_c = __uint_as_float(scalar_u(A, k + ADSR_D_C));
    const float sustain = __uint_as_float(scalar_u(A, k + ADSR_SUSTAIN));
    if (v > 0.0f) {
        e.vel = clamp01(v);
        e.sus = clamp01(sustain * e.vel); // update_sustain_level :93
        if (a_inst) {                     // attack <= MIN_TIME_SECONDS
            e.lv = 1.0f;
            adsr_enter(e, ST_DECAY, d_n, a_c, d_c); // set_stage(Decay, sustain_level): d_n >= 1
        } else {
            adsr_enter(e, ST_ATTACK, a_n, a_c, d_c); // set_stage(Attack, 1.0): a_n >= 1
        }
    } else if (r_inst) {
        e.lv = 0.0f;
        adsr_enter(e, ST_IDLE, 0, a_c, d_c);
    } else {
        adsr_enter(e, ST_RELEASE, r_n, a_c, d_c); // release_increment is re-derived every sample (adsr.rs:112-114)
    }
}

// process_stage  adsr.rs:206-248, the per-sample part
//
// Tolerance mode (default; og_math.h): every moving stage is ONE fused one-pole step towards the stage's
// target.  Attack / Decay: lv += (tgt - lv) * cf as one subtraction and one fma.  Release -- the reference's
// `level += -level / samples_remaining` with the increment re-derived every sample (adsr.rs:112-114, 162-173) -- is
// the same step with target 0 and the time-varying coefficient 1 / samples_remaining: tgt = 0 and cf = 0 in
// Release (adsr_enter), so the coefficient is cf + rs * rcp(cnt) for every stage and the release arithmetic is a
// conversion, v_rcp_f32 and one fma in front of the common step (6 VALU per envelope and sample; the
// operation-for-operation form with its Newton-refined quotient took 11).  Against the reference's correctly
// rounded quotient the increment differs by <= 2^-23 relative: the step lands on a neighbouring f32 about 2/n of
// the time, and the recurrence contracts (lv shrinks by 1 - 1/n), so the level stays within a few ulp over a whole
// release (observed: DESIGN.md section 5).  Nothing here feeds a phase accumulator.
// FC: the caller keeps e.fc == (float)e.cnt at the top of the chunk (see Adsr::fc); otherwise cnt is converted here.
template <bool RELEASE = true, bool FC = false>
OG_DEV float adsr_tick(Adsr& e)
{
#ifdef OG_STRICT
    // Attack: lv += (1 - lv) * attack_coeff; Decay: lv += (sustain_level - lv) * decay_coeff; else cf == 0
    float lv = e.lv + (e.tgt - e.lv) * e.cf;
    // Release: lv += -lv / samples_remaining, increment re-derived every sample (adsr.rs:162-173)
    // (computed unconditionally: a wave-uniform "any lane releasing" test costs more issue slots
    //  than it saves -- the compiler if-converts it into the same arithmetic plus scalar selects)
    // (RELEASE = false: the caller has established rs == 0 in every lane for the whole chunk)
    if (RELEASE) lv = fmaf(e.rs, div_near(-lv, (float)e.cnt), lv);
#else
    float cf = e.cf;
    // (cnt >= 1 while a stage is ticking and ADSR_HOLD otherwise: the reciprocal is finite, rs * rcp is 0 or rcp)
    if (RELEASE) {
        cf = fmaf(e.rs, __builtin_amdgcn_rcpf(FC ? e.fc : (float)e.cnt), cf);
        if (FC) e.fc -= 1.0f;
    }
    const float lv = fmaf(e.tgt - e.lv, cf, e.lv);
#endif
    e.cnt -= 1u; // samples_remaining -= 1; reaching 0 is handled by adsr_complete()
    e.lv = lv;
    return lv;
}

// complete_stage  adsr.rs:175-204 for an envelope whose countdown hit 0 this frame
OG_DEV void adsr_complete(Adsr& e, float a_c, float d_c, uint32_t d_n, float& out)
{
    if (e.cnt == 0u) {
        e.lv = e.tgt; // the stage ends on its target level, which is also this frame's output
        out = e.lv;
        const uint32_t next = (e.stage + 1u) & 7u; // Attack -> Decay -> Sustain, Release -> Idle
        adsr_enter(e, next, d_n, a_c, d_c);
    }
}

// ---------------------------------------------------------------------------
// FmOperator  examples/fm-synth/src/nodes/fm_operator.rs:58-76
// `inc` = (base_freq * ratio) / sample_rate, hoisted by the caller when both
// factors are constant over the block (bit-identical: same IEEE ops).
// ---------------------------------------------------------------------------
OG_DEV float fm_operator_tick(float& phase, float& prev_output, float inc, float phase_mod, float feedback,
                              float envelope, float level)
{
    const float total_phase_mod = OG_FMA(prev_output, feedback, phase_mod); // phase_mod + prev_output * feedback
    const float output = og_sin_turns(phase + total_phase_mod) * envelope * level; // ((phase + mod) * TAU).sin(): og_math.h, OG_SIN_TURNS
    prev_output = output;
    const float p = phase + inc; // the phase accumulator keeps the reference's exact operations
    phase = fract_phase(p); // f32::fract
    return output;
}

// feedback input left at its constructor value 0.0 (FMVoice op1): prev*0 + pm == pm
// for finite prev (and a -0/+0 difference cannot reach the output: it is added to phase >= 0)
OG_DEV float fm_operator_tick_nofb(float& phase, float& prev_output, float inc, float phase_mod, float envelope,
                                   float level)
{
    const float output = og_sin_turns(phase + phase_mod) * envelope * level;
    prev_output = output;
    const float p = phase + inc;
    phase = fract_phase(p);
    return output;
}

// ---------------------------------------------------------------------------
// TptFilter<f32>  oscen-lib/src/filters/tpt/mod.rs
// ---------------------------------------------------------------------------
// update_coefficients :69-82.  two_sr = 2*sr, period = 0.5/sr, nyquist =
// sr*0.5 - EPSILON are block-uniform host slots.
OG_DEV void tpt_update_coefficients(float cutoff, float q, float two_sr, float period, float nyquist, float& cur_c,
                                    float& cur_q, float& h, float& g, float& k)
{
    const float freq = clampf(cutoff, 20.0f, nyquist);
    const float f = two_sr * og_tanf_q1(F32_TAU * freq * period) * period;
    const float inv_q = 1.0f / q;
    // (rcp + one Newton step: within an ulp of the IEEE quotient, a third of its instructions; the
    //  coefficient already carries og_tanf_q1's few-ulp error)
    h = div_near(1.0f, OG_FMA(f, f, OG_FMA(inv_q, f, 1.0f))); // 1 / (1 + inv_q * f + f * f)
    g = f;
    k = f + inv_q;
    cur_c = cutoff;
    cur_q = q;
}

// apply_parameter_updates :85-102, general form (f_mod connected)
OG_DEV void tpt_params_mod(float cutoff_in, float q_in, float f_mod, float max_cutoff, float two_sr, float period,
                           float nyquist, float& cur_c, float& cur_q, float& h, float& g, float& k)
{
    const float cutoff_base = clampf(cutoff_in, 20.0f, max_cutoff);
    const float q = clampf(q_in, 0.1f, 10.0f);
    const float modulation = clampf(f_mod, -1.0f, 1.0f);
    const float min_factor = 20.0f / cutoff_base;
    const float max_factor = max_cutoff / cutoff_base;
    const float factor = clampf(1.0f + modulation, min_factor, max_factor);
    const float cutoff = clampf(cutoff_base * factor, 20.0f, max_cutoff);
    if (fabsf(cutoff - cur_c) > F32_EPSILON || fabsf(q - cur_q) > F32_EPSILON)
        tpt_update_coefficients(cutoff, q, two_sr, period, nyquist, cur_c, cur_q, h, g, k);
}

// Same with f_mod structurally 0.0 (input left unconnected, e.g. FMVoice):
// factor = clamp(1.0, 20/cb, max/cb) == 1.0 exactly because correctly rounded
// 20/cb <= 1 <= max/cb for cb in [20, max]; cb*1.0 == cb; the outer clamp is
// then the identity.  Result-identical, two divides cheaper.
// Used where cutoff and q cannot change between events (block-uniform or per-voice
// values): the caller runs it once per block and after per-voice value events.
OG_DEV void tpt_params_nomod(float cutoff_in, float q_in, float max_cutoff, float two_sr, float period,
                             float nyquist, float& cur_c, float& cur_q, float& h, float& g, float& k)
{
    const float cutoff = clampf(cutoff_in, 20.0f, max_cutoff);
    const float q = clampf(q_in, 0.1f, 10.0f);
    if (fabsf(cutoff - cur_c) > F32_EPSILON || fabsf(q - cur_q) > F32_EPSILON)
        tpt_update_coefficients(cutoff, q, two_sr, period, nyquist, cur_c, cur_q, h, g, k);
}

// apply_parameter_updates() for a cutoff that is computed every frame but mostly does NOT move (FMVoice: `env * amount +
// cutoff` with amount 0, or an envelope that sits in Sustain): the reference runs its test on every frame; if this frame's
// inputs are bit for bit the previous frame's, that test cannot fire -- the previous frame either updated to these very
// values or found them within EPSILON of the current ones -- so one integer compare of the raw input against the last one
// seen replaces clamp, two subtractions and two float compares.  `last_in` / `last_q` are per-lane registers, not state:
// they start from a sentinel at every launch (and after a per-voice value event), so the first frame always runs the full
// test.  QCHK: q can change inside a launch (a ramped or per-frame q); otherwise only the cutoff is watched.
// update_coefficients for the lazy form below: the same operations as tpt_update_coefficients, arranged for a wave that
// updates on (nearly) every frame -- `inv_q` = 1.0f / q comes from the caller when q cannot change inside the launch (an IEEE
// division, ten instructions, formed once per launch instead of once per frame: the compiler cannot hoist it itself, the
// slot it derives from is pinned to a scalar register by a convergent readfirstlane), and og_tanf_q1's `x > pi/4` arm
// (cutoff above a quarter of the sample rate) sits behind ONE wave-uniform test instead of a divergent region per frame.
OG_DEV void tpt_update_coefficients_iq(float cutoff, float q, float inv_q, float two_sr, float period, float nyquist, float& cur_c,
                                       float& cur_q, float& h, float& g, float& k)
{
    const float freq = clampf(cutoff, 20.0f, nyquist);
    const float x = F32_TAU * freq * period;
    float t;
    if (__any((int)(x > 0x1.921fb6p-1f))) t = og_tanf_q1(x);
    else t = og_tan_poly(x);
    const float f = two_sr * t * period;
    h = div_near(1.0f, OG_FMA(f, f, OG_FMA(inv_q, f, 1.0f))); // 1 / (1 + inv_q * f + f * f)
    g = f;
    k = f + inv_q;
    cur_c = cutoff;
    cur_q = q;
}

// IQ: `inv_q_in` = 1.0f / clamp(q) was formed by the caller (q is constant over the launch)
template <bool QCHK, bool IQ>
OG_DEV void tpt_params_nomod_lazy(float cutoff_in, float q_in, float inv_q_in, float& last_in, float& last_q, float max_cutoff, float two_sr,
                                  float period, float nyquist, float& cur_c, float& cur_q, float& h, float& g, float& k)
{
#ifdef OG_STRICT // the reference's test on every frame (the lazy form picks up a q step below EPSILON between launches)
    constexpr bool strict = true;
#else
    constexpr bool strict = false;
#endif
    if (QCHK || strict) {
        // q can change inside this launch (the kernel variants that read the ramp table): the reference's own per-frame test.
        // Watching two inputs and keeping the EPSILON test for q -- below 1.0 two distinct q can be closer than EPSILON, and
        // the reference ignores such a step -- nests two divergent regions and measured 9 % slower on those launches
        // (profiles/r05g_session6.log, scripts/r5_session8.sh) than the plain test.
        (void)last_q;
        tpt_params_nomod(cutoff_in, q_in, max_cutoff, two_sr, period, nyquist, cur_c, cur_q, h, g, k);
        return;
    }
    // Wave-uniform: if any lane's cutoff moved, every lane updates -- a lane whose input did not move re-derives the
    // coefficients it already has from the very same input (the first frame of a launch moves every lane off the
    // sentinel), so the result is the per-lane test's; the usual frame leaves on one scalar branch (v_cmp + s_cbranch_vccz)
    // instead of entering and leaving an empty divergent region (two more SALU and a register copy per frame), and the
    // frame that does update runs without an exec mask.
    const bool moved = __float_as_uint(cutoff_in) != __float_as_uint(last_in);
    last_in = cutoff_in; // (unconditional: where nothing moved it is the value it had -- no register to reconcile at the join)
    if (__any((int)moved)) {
        const float cutoff = clampf(cutoff_in, 20.0f, max_cutoff);
        const float q = clampf(q_in, 0.1f, 10.0f);
        // ONE divergent region, no second test inside it: a clamped cutoff (>= 20 Hz, ulp 1.9e-6) that differs from the
        // current one at all differs by more than EPSILON, so the reference updates; one that does not differ re-derives the
        // coefficients it already has.  (With the reference's test nested inside, the bank whose cutoff moves every frame
        // lost 11 %.)  q is constant over the launch here; a change of q between launches smaller than EPSILON, which the
        // reference would ignore, is picked up by the first frame's update -- part of the tolerance mode.
#ifdef OG_TPT_OLD_UPDATE // (A/B: round 5's update path)
        tpt_update_coefficients(cutoff, q, two_sr, period, nyquist, cur_c, cur_q, h, g, k);
#else
        tpt_update_coefficients_iq(cutoff, q, IQ ? inv_q_in : 1.0f / q, two_sr, period, nyquist, cur_c, cur_q, h, g, k);
#endif
    }
}
constexpr uint32_t TPT_LAZY_SENTINEL = 0x7fc0a5a5u; // a NaN payload no computation produces

// The same update for a cutoff that moves every sample (an envelope on the cutoff, FMVoice): written
// without a branch -- the new coefficients are computed on every tick and selected in.  A branch per
// frame costs more than it saves here: it is taken on most frames anyway, and it cuts the unrolled
// frames into separate scheduling regions, so the (serial) sine and filter chains of consecutive
// frames cannot be overlapped.
OG_DEV void tpt_params_nomod_flat(float cutoff_in, float q_in, float inv_q_in, float max_cutoff, float two_sr, float period,
                                  float nyquist, float& cur_c, float& cur_q, float& h, float& g, float& k)
{
    const float cutoff = clampf(cutoff_in, 20.0f, max_cutoff);
    const float q = clampf(q_in, 0.1f, 10.0f); // inv_q_in = 1.0f / q, formed by the caller (hoisted when q is block-constant)
    const bool upd = fabsf(cutoff - cur_c) > F32_EPSILON || fabsf(q - cur_q) > F32_EPSILON;
    const float freq = clampf(cutoff, 20.0f, nyquist);
    // og_tanf_q1 flattened: x <= pi/4 -> poly(x), else 1 / poly(pi/2 - x)
    const float x = F32_TAU * freq * period;
    const bool big = x > 0x1.921fb6p-1f;
    const float y = big ? ((0x1.921fb6p+0f - x) + -0x1.777a5cp-25f) : x;
    const float t0 = og_tan_poly(y);
    const float t = big ? div_near(1.0f, t0) : t0;
    const float f = two_sr * t * period;
    const float nh = div_near(1.0f, OG_FMA(f, f, OG_FMA(inv_q_in, f, 1.0f)));
    h = upd ? nh : h;
    g = upd ? f : g;
    k = upd ? f + inv_q_in : k;
    cur_c = upd ? cutoff : cur_c;
    cur_q = upd ? q : cur_q;
}

// state-variable core :114-122.  The integrators are a stable (contracting) recurrence, not an accumulator:
// tolerance mode fuses every product into the sum it feeds -- 7 VALU instead of 9.
OG_DEV float tpt_tick(float in, float& z0, float& z1, float h, float g, float k)
{
#ifdef OG_STRICT
    const float high = (in - z0 * k - z1) * h;
    const float hg = high * g;
    const float band = hg + z0;
    const float bg = band * g;
    const float low = bg + z1;
    z0 = hg + band;
    z1 = bg + low;
#else
    const float high = (fmaf(-z0, k, in) - z1) * h;
    const float band = fmaf(high, g, z0);
    const float low = fmaf(band, g, z1);
    z0 = fmaf(high, g, band);
    z1 = fmaf(band, g, low);
#endif
    return low;
}

// ---------------------------------------------------------------------------
// PolyBlepOscillator  oscen-lib/src/oscillators/mod.rs:88-233
// ---------------------------------------------------------------------------
enum : uint32_t { PB_SINE = 0, PB_SAW = 1, PB_SQUARE = 2, PB_TRIANGLE = 3 };

// x % 1.0 (Rust `%` = C fmod): x - trunc(x) is that value exactly (the subtraction is exact: both
// operands share the exponent range of x and the result has fewer significant bits).  Two VALU ops
// instead of the general fmod loop.  (Differs from fmod only in the sign of a zero result for -0.0.)
// -DOG_STRICT keeps the reference's own operations in both: x - trunc(x), and for rem_euclid the `r + 1.0` that
// can round up to 1.0 where v_fract clamps to 0.99999994.
#ifdef OG_STRICT
OG_DEV float fmod1(float x) { return x - truncf(x); }
OG_DEV float wrap_phase(float p) // rem_euclid(1.0) :171-173
{
    const float r = p - truncf(p);
    return (r < 0.0f) ? r + 1.0f : r;
}
#else
OG_DEV float fmod1(float x) { return fract_keep_sign(x); }

OG_DEV float wrap_phase(float p) { return fract_floor(p); } // rem_euclid(1.0) :171-173
#endif

// poly_blep :139-153 and poly_blamp :155-169, written without branches: in a bank some lane is next to
// a discontinuity on almost every sample, so both sides are evaluated and selected.  rdt = rcp(dt) is
// shared by the two quotients, each refined by one Newton step (see div_near).
OG_DEV float div_rcp(float a, float b, float rb)
{
    const float q = a * rb;
    const float r = fmaf(-q, b, a);
    return fmaf(r, rb, q);
}

OG_DEV float poly_blep(float t, float dt, float rdt)
{
    const float x1 = div_rcp(t, dt, rdt);
    const float r1 = OG_FMA(-x1, x1, x1 + x1) - 1.0f; // x1 + x1 - x1 * x1 - 1
    const float x2 = div_rcp(t - 1.0f, dt, rdt);
    const float r2 = OG_FMA(x2, x2, x2) + x2 + 1.0f;
    const float res = (t < dt) ? r1 : ((t > 1.0f - dt) ? r2 : 0.0f);
    return (dt > F32_EPSILON) ? res : 0.0f;
}

OG_DEV float poly_blamp(float t, float dt, float rdt)
{
    const float x1 = div_rcp(t, dt, rdt) - 1.0f;
    const float r1 = -(x1 * x1 * x1) / 3.0f;
    const float x2 = div_rcp(t - 1.0f, dt, rdt) + 1.0f;
    const float r2 = (x2 * x2 * x2) / 3.0f;
    const float res = (t < dt) ? r1 : ((t > 1.0f - dt) ? r2 : 0.0f);
    return (dt > F32_EPSILON) ? res : 0.0f;
}

// frequency after modulation and its per-sample increment (:176-180).  Constant over the block for a
// voice whose frequency inputs are; the caller then forms them in derive().
OG_DEV float polyblep_frequency(float frequency_in, float frequency_mod) { return fmaxf(frequency_in * (1.0f + frequency_mod), 0.0f); }
OG_DEV float polyblep_increment(float frequency, float sr) { return frequency / fmaxf(sr, F32_EPSILON); }

// BELOW_QUARTER: the caller has established frequency < sr / 4 for every lane (a per-voice block constant),
// so the sine fallback of :186 and its divergent branch are compiled out.
template <uint32_t WAVE, bool BELOW_QUARTER = false>
OG_DEV float polyblep_tick(float& phase_state, float frequency, float freq_per_sample, float phase_mod,
                           float amplitude, float pulse_width_in, float sr)
{
    float pulse_width = clampf(pulse_width_in, 0.0001f, 0.9999f);
    float phase = wrap_phase(phase_state + phase_mod);
    const float dt = fminf(freq_per_sample, 1.0f);
    const float rdt = __builtin_amdgcn_rcpf(dt);
    if (pulse_width <= 0.0f) pulse_width = 0.0001f;
    float value;
    if ((!BELOW_QUARTER && frequency >= sr * 0.25f) || WAVE == PB_SINE) {
        value = og_sinf(phase * F32_TAU);
    } else if (WAVE == PB_SAW) {
        float y = OG_FMA(2.0f, phase, -1.0f);
        y -= poly_blep(phase, dt, rdt);
        value = y;
    } else if (WAVE == PB_SQUARE) {
        float y = (phase < pulse_width) ? 1.0f : -1.0f;
        y += poly_blep(phase, dt, rdt);
        const float t = wrap_phase(phase + 1.0f - pulse_width);
        y -= poly_blep(t, dt, rdt);
        value = y;
    } else {
        float y = 4.0f * phase;
        if (y >= 3.0f) {
            y -= 4.0f;
        } else if (y > 1.0f) {
            y = 2.0f - y;
        }
        const float t1 = wrap_phase(phase + 0.25f);
        const float t2 = wrap_phase(phase + 0.75f);
        value = OG_FMA(4.0f * dt, poly_blamp(t1, dt, rdt) - poly_blamp(t2, dt, rdt), y);
    }
    value = value * amplitude;
    phase_state = wrap_phase(phase_state + freq_per_sample);
    return value;
}

// ---------------------------------------------------------------------------
// Oscillator  oscen-lib/src/oscillators/mod.rs:7-76
// ---------------------------------------------------------------------------
enum : uint32_t { OSC_SINE = 0, OSC_SQUARE = 1, OSC_SAW = 2 };

template <uint32_t WAVE>
OG_DEV float oscillator_tick(float& phase, float frequency_in, float frequency_mod, float amplitude, float sr)
{
    const float frequency = frequency_in * (1.0f + frequency_mod);
    const float p = fmod1(phase);
    float w;
    if (WAVE == OSC_SINE) {
        w = og_sinf(p * 2.0f * 3.14159274101257324f); // (p * 2.0 * PI).sin()
    } else if (WAVE == OSC_SQUARE) {
        w = (p < 0.5f) ? 1.0f : -1.0f;
    } else {
        const float transition_width = 0.1f;
        const float raw_saw = 2.0f * p - 1.0f;
        if (p > (1.0f - transition_width / 2.0f)) {
            const float t = (p - (1.0f - transition_width / 2.0f)) / (transition_width / 2.0f);
            w = -1.0f + (1.0f - t * t) * (raw_saw + 1.0f);
        } else {
            w = raw_saw;
        }
    }
    const float out = w * amplitude;
    phase += frequency / sr;
    phase = fmod1(phase);
    return out;
}

// ---------------------------------------------------------------------------
// Small nodes: gain/mod.rs:30-34, fm-synth nodes/{add_value,crossfade,mixer}.rs,
// pivot vca.rs:31-35, oversampled-saturator main.rs:54-61
// ---------------------------------------------------------------------------
OG_DEV float hardclip(float in) { return clampf(in * 1.5f, -0.7f, 0.7f); }

// ---------------------------------------------------------------------------
// IirLowpass  oscen-lib/src/filters/iir_lowpass/mod.rs:84-164
// ---------------------------------------------------------------------------
constexpr float F32_PI = 3.14159274101257324f;

// update_coefficients :84-101 (runs when frame_counter == 0, i.e. every 32nd tick)
OG_DEV void iir_lowpass_coeffs(float cutoff, float q_in, float sr, float nyquist, float& b0, float& b1, float& b2,
                               float& a1, float& a2)
{
    const float freq = clampf(cutoff, 20.0f, nyquist);
    const float q = fmaxf(q_in, 0.01f);
    const float n = 1.0f / og_tanf_q1(F32_PI * freq / sr);
    const float n2 = n * n;
    const float iq = 1.0f / q;
    const float c1 = 1.0f / (1.0f + iq * n + n2);
    b0 = c1;
    b1 = c1 * 2.0f;
    b2 = c1;
    a1 = c1 * 2.0f * (1.0f - n2);
    a2 = c1 * (1.0f - iq * n + n2);
}

// process_sample :109-134 (transposed direct form II with the 1e-15 denormal snaps)
OG_DEV float iir_lowpass_tick(float in, float& v1, float& v2, float b0, float b1, float b2, float a1, float a2)
{
    constexpr float DENORMAL_THRESHOLD = 1e-15f;
    in = (fabsf(in) < DENORMAL_THRESHOLD) ? 0.0f : in;
    // (NOT contracted, also in tolerance mode: at low cutoffs a1 -> -2, a2 -> 1 and the two state updates cancel almost
    //  completely -- fusing them moved the output by 2.6e-5 against the oracle, tests/test_n3_gpu.py::test_iir_lowpass)
    const float out = b0 * in + v1;
    v1 = b1 * in - a1 * out + v2;
    v2 = b2 * in - a2 * out;
    v1 = (fabsf(v1) < DENORMAL_THRESHOLD) ? 0.0f : v1;
    v2 = (fabsf(v2) < DENORMAL_THRESHOLD) ? 0.0f : v2;
    return out;
}

// ---------------------------------------------------------------------------
// LP18Filter  examples/nih-twin-peaks/src/lp18_filter.rs:63-107
// ---------------------------------------------------------------------------
OG_DEV void lp18_params(float cutoff, float fmod, float resonance, float sr, float& g, float& h, float& last_cutoff,
                        float& last_fmod, float& last_resonance)
{
    if (cutoff != last_cutoff || fmod != last_fmod) { // :82-86, update_cutoff_coefficient :63-67
        last_cutoff = cutoff;
        last_fmod = fmod;
        g = og_tanf_q1(F32_PI * clampf((cutoff + fmod) / sr, 0.001f, 0.33f));
    }
    if (resonance != last_resonance) { // :88-92
        last_resonance = resonance;
        h = 2.0f * clampf(resonance, 0.0f, 0.99f);
    }
}

OG_DEV float lp18_tick(float in, float& z0, float& z1, float& z2, float g, float h) // :94-106
{
    const float hp = (in - h * z0 - z1 - z2) / (1.0f + g);
    const float bp1 = OG_FMA(g, hp, z0);
    z0 = tanhf(bp1);
    const float bp2 = OG_FMA(g, bp1, z1);
    z1 = bp2;
    const float lp = OG_FMA(g, bp2, z2);
    z2 = lp;
    return lp;
}

// ---------------------------------------------------------------------------
// Delay  oscen-lib/src/delay/mod.rs:47-83 over RingBuffer (PowerOfTwo mode)
// oscen-lib/src/ring_buffer/mod.rs:56-203.  The line of voice v lives in HBM at
// ring[slot * n_voices + v]: the 64 voices of a wave read/write one 256-byte
// row per access.  write_pos / frame_counter / the two parameter fields are
// ordinary per-voice state words.
// ---------------------------------------------------------------------------
OG_DEV float ring_get(const float* ring, uint32_t nv, uint32_t v, uint32_t cap, uint32_t wp, float offset) // :167-203
{
    const uint32_t mask = cap - 1u;
    const float o = fmaxf(offset, 0.0f);
    const float fr = o - truncf(o);
    if (fr < 1e-6f || (1.0f - fr) < 1e-6f) { // (almost) whole samples: the exact sample :178-191
        const unsigned long long os = (unsigned long long)roundf(o);
        const uint32_t idx = ((wp + cap) - ((uint32_t)os & mask) - 1u) & mask;
        return ring[(size_t)idx * nv + v];
    } // (delay_tick takes this path through ring_whole_offset; kept here so that ring_get is the whole of get())
    // get_cubic :120-164 (capacity >= 4 always holds for a prepared Delay); read_pos :80-92 in f32
    const float n = (float)cap;
    float rp = (float)wp - o - 1.0f;
    rp = fmodf(fmodf(rp, n) + n, n);
    const uint32_t i = (uint32_t)rp;
    const float f = rp - truncf(rp);
    const float v0 = ring[(size_t)((i - 1u) & mask) * nv + v];
    const float v1 = ring[(size_t)(i & mask) * nv + v];
    const float v2 = ring[(size_t)((i + 1u) & mask) * nv + v];
    const float v3 = ring[(size_t)((i + 2u) & mask) * nv + v];
    const float c0 = v1;
    const float c1 = 0.5f * (v2 - v0);
    const float c2 = v0 - 2.5f * v1 + 2.0f * v2 - 0.5f * v3;
    const float c3 = 0.5f * (v3 - v0) + 1.5f * (v1 - v2);
    return c0 + f * (c1 + f * (c2 + f * c3));
}

// Whole-sample reads are staged one 16-frame chunk ahead: at the top of a chunk the 16 samples the
// NEXT chunk will read are requested from HBM into registers (one exposed latency per chunk at most,
// none when the delay is >= 32 samples) and the ones requested a chunk ago move to an LDS column that
// the ticks read.  The staged samples are only a prediction: every tick recomputes the reference's
// index and falls back to a direct load when the offset it finds is not the predicted one.
constexpr uint32_t RING_NONE = 0xffffffffu;
struct RingPre {
    uint32_t pred;      // (offset_samples & mask) the LDS column was loaded for, or RING_NONE
    uint32_t pred_next; // same for next[]
    bool fast;          // wave-uniform: every tick of this chunk reads its staged sample (see ring_chunk_begin)
    float next[OG_BUS_CHUNK];
};

// offset_samples % capacity of the exact-sample path of RingBuffer::get (:178-191), RING_NONE on the cubic path.
// On this path fract < 1e-6 or > 1 - 1e-6, so round() is trunc or trunc + 1; `as usize` is exact below 2^32
// and goes through the 64-bit conversion above (floats there are whole multiples of 512).
OG_DEV uint32_t ring_whole_offset(float offset, uint32_t cap)
{
    const float o = fmaxf(offset, 0.0f);
    const float t = truncf(o);
    const float fr = o - t;
    const bool lo = fr < 1e-6f, hi = (1.0f - fr) < 1e-6f;
    const float r = lo ? t : t + 1.0f;
    uint32_t om = (uint32_t)r;
    if (r >= 4294967296.0f) om = (uint32_t)((unsigned long long)r);
    return (lo || hi) ? (om & (cap - 1u)) : RING_NONE;
}

// fixed_params: delay_samples and feedback cannot change inside the chunk (block-constant inputs or
// unconnected fields); when, in addition, every lane's offset is the staged one and the every-32nd-tick
// clamps are no-ops, the chunk's ticks skip the index arithmetic and its branches (P.fast).
OG_DEV void ring_chunk_begin(const float* ring, uint32_t cap, uint32_t nv, uint32_t v, bool valid, float offset_hint,
                             float feedback_hint, bool fixed_params, uint32_t wp, RingPre& P, float (*lds)[OG_WAVE],
                             uint32_t lane, bool more)
{
    const uint32_t mask = cap - 1u;
    const uint32_t om = valid ? ring_whole_offset(offset_hint, cap) : RING_NONE;
    if (P.pred_next != RING_NONE) { // requested a chunk ago for exactly this write position
#pragma unroll
        for (uint32_t j = 0; j < OG_BUS_CHUNK; ++j) lds[j][lane] = P.next[j];
        P.pred = P.pred_next;
    } else if (om != RING_NONE && om >= OG_BUS_CHUNK) { // first chunk of a launch / short delays
#pragma unroll
        for (uint32_t j = 0; j < OG_BUS_CHUNK; ++j) lds[j][lane] = ring[(size_t)((wp + j - om - 1u) & mask) * nv + v];
        P.pred = om;
    } else {
        P.pred = RING_NONE;
    }
    if (more && om != RING_NONE && om >= 2u * OG_BUS_CHUNK) {
#pragma unroll
        for (uint32_t j = 0; j < OG_BUS_CHUNK; ++j)
            P.next[j] = ring[(size_t)((wp + OG_BUS_CHUNK + j - om - 1u) & mask) * nv + v];
        P.pred_next = om;
    } else {
        P.pred_next = RING_NONE;
    }
    const bool lane_fast = om != RING_NONE && om == P.pred && offset_hint == clampf(offset_hint, 0.0f, (float)cap - 1.0f) &&
                           feedback_hint == clampf(feedback_hint, 0.0f, 0.99f);
    P.fast = fixed_params && __all((int)(!valid || lane_fast));
}

OG_DEV float delay_tick(float* ring, uint32_t cap, uint32_t nv, uint32_t v, bool valid, float in, float& delay_samples,
                        float& feedback, uint32_t& wp, uint32_t& fc, const RingPre& P, const float (*lds)[OG_WAVE],
                        uint32_t lane, uint32_t j)
{
    if (P.fast) { // established for the whole chunk by ring_chunk_begin: clamps are no-ops, the sample is staged
        fc = (fc + 1u) & 31u;
        float delayed = 0.0f;
        if (valid) {
            delayed = lds[j][lane];
            ring[(size_t)wp * nv + v] = in + delayed * feedback;
        }
        wp = (wp + 1u) & (cap - 1u);
        return delayed;
    }
    // apply_parameter_updates :47-56 (selects, not a branch: the counter is the same in every lane)
    const bool upd = fc == 0u;
    const float ds_c = clampf(delay_samples, 0.0f, (float)cap - 1.0f), fb_c = clampf(feedback, 0.0f, 0.99f);
    delay_samples = upd ? ds_c : delay_samples;
    feedback = upd ? fb_c : feedback;
    fc = (fc + 1u) & 31u;
    float delayed = 0.0f;
    if (valid) {
        const uint32_t om = ring_whole_offset(delay_samples, cap);
        if (om != RING_NONE && om == P.pred) {
            delayed = lds[j][lane]; // staged a chunk ago
        } else {
            if (om == RING_NONE)
                delayed = ring_get(ring, nv, v, cap, wp, delay_samples);
            else
                delayed = ring[(size_t)((wp + cap - om - 1u) & (cap - 1u)) * nv + v];
            // finish the load inside this (rare) branch: otherwise the wait for it is placed after the join
            // as vmcnt(0), where the common path would sit out every store and staging load in flight
            asm volatile("" : "+v"(delayed));
        }
        ring[(size_t)wp * nv + v] = in + delayed * feedback; // push :57-77
    }
    wp = (wp + 1u) & (cap - 1u);
    return delayed;
}

// ---------------------------------------------------------------------------
// Cross-rate kernels  oscen-lib/src/resample/{sinc_fir,halfband_iir,linear,latch}.rs
// Histories are kept UNROTATED in registers: index 0 is always the newest
// sample, a push shifts the array (the reference's ring + head gives the same
// taps: its at(d) is h[d] for the up stage and h[d+1] for the down stage).
// Same taps, same summation order, bit-identical results.
// ---------------------------------------------------------------------------
template <int N>
struct Log2 {
    static constexpr int v = (N >= 8) ? 3 : (N >= 4) ? 2 : (N >= 2) ? 1 : 0;
};

// Halfband2xDownStage::step  sinc_fir.rs:115-138 (taps coeffs.rs:17-27)
OG_DEV float hb_down_step(float (&h)[24], float x0, float x1)
{
    const float HALF[6] = {-3.8558514e-5f, 1.2218465e-3f, -7.2854808e-3f,
                           2.6409210e-2f,  -7.8128843e-2f, 3.0782697e-1f};
    const float CENTER = 0.4999897f;
#pragma unroll
    for (int i = 23; i >= 2; --i) h[i] = h[i - 2];
    h[1] = x0;
    h[0] = x1;
    float acc = h[11 + 1] * CENTER;
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
        const float left = h[2 * kk + 1];
        const float right = h[22 - 2 * kk + 1];
        acc = OG_FMA(left + right, HALF[kk], acc); // (taps in the reference's order; tolerance mode drops the product's rounding)
    }
    return acc;
}

// SincDownFir<N>::downsample  sinc_fir.rs:232-247
template <int N>
OG_DEV float sinc_down(float (&h)[Log2<N>::v][24], const float (&xs)[N])
{
    float buf[8];
#pragma unroll
    for (int i = 0; i < N; ++i) buf[i] = xs[i];
    int len = N;
#pragma unroll
    for (int s = 0; s < Log2<N>::v; ++s) {
        const int half = len / 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < half) buf[i] = hb_down_step(h[s], buf[2 * i], buf[2 * i + 1]);
        len = half;
    }
    return buf[0];
}

// Halfband2xUpStage::step  sinc_fir.rs:50-76
OG_DEV void hb_up_step(float (&h)[12], float x, float& y0, float& y1)
{
    const float HALF[6] = {-3.8558514e-5f, 1.2218465e-3f, -7.2854808e-3f,
                           2.6409210e-2f,  -7.8128843e-2f, 3.0782697e-1f};
    const float CENTER2 = 2.0f * 0.4999897f;
#pragma unroll
    for (int i = 11; i >= 1; --i) h[i] = h[i - 1];
    h[0] = x;
    y1 = h[5] * CENTER2;
    float acc = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) acc = OG_FMA(h[kk] + h[11 - kk], HALF[kk], acc);
    y0 = acc * 2.0f;
}

// SincUpFir<N>::upsample  sinc_fir.rs:170-190
template <int N>
OG_DEV void sinc_up(float (&h)[Log2<N>::v][12], float x, float (&out)[N])
{
    float buf[8], next[8];
    buf[0] = x;
    int len = 1;
#pragma unroll
    for (int s = 0; s < Log2<N>::v; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < len) hb_up_step(h[s], buf[i], next[2 * i], next[2 * i + 1]);
        len *= 2;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < len) buf[i] = next[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = buf[i];
}

// IirHalfband2x  halfband_iir.rs:30-136; state per stage: a0{x,y} a1{x,y} b0{x,y} b1{x,y} prev_odd_in
OG_DEV float flush_denormal(float x) { return (fabsf(x) < 1e-15f) ? 0.0f : x; }
OG_DEV float allpass1_step(float a, float& xp, float& yp, float x)
{
    const float y = OG_FMA(x - yp, a, xp);
    xp = flush_denormal(x);
    yp = flush_denormal(y);
    return y;
}
OG_DEV void iir_step_up(float (&st)[9], float x, float& y0, float& y1)
{
    float a = allpass1_step(0.1355741f, st[0], st[1], x);
    a = allpass1_step(0.6975849f, st[2], st[3], a);
    float b = allpass1_step(0.4253804f, st[4], st[5], x);
    b = allpass1_step(0.9055601f, st[6], st[7], b);
    y0 = a;
    y1 = b;
}
OG_DEV float iir_step_down(float (&st)[9], float x0, float x1)
{
    float a = allpass1_step(0.1355741f, st[0], st[1], x0);
    a = allpass1_step(0.6975849f, st[2], st[3], a);
    float b = allpass1_step(0.4253804f, st[4], st[5], st[8]);
    b = allpass1_step(0.9055601f, st[6], st[7], b);
    st[8] = x1;
    return (a + b) * 0.5f;
}
template <int N>
OG_DEV void iir_up(float (&st)[Log2<N>::v][9], float x, float (&out)[N])
{
    float buf[8], next[8];
    buf[0] = x;
    int len = 1;
#pragma unroll
    for (int s = 0; s < Log2<N>::v; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < len) iir_step_up(st[s], buf[i], next[2 * i], next[2 * i + 1]);
        len *= 2;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < len) buf[i] = next[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = buf[i];
}
template <int N>
OG_DEV float iir_down(float (&st)[Log2<N>::v][9], const float (&xs)[N])
{
    float buf[8];
#pragma unroll
    for (int i = 0; i < N; ++i) buf[i] = xs[i];
    int len = N;
#pragma unroll
    for (int s = 0; s < Log2<N>::v; ++s) {
        const int half = len / 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < half) buf[i] = iir_step_down(st[s], buf[2 * i], buf[2 * i + 1]);
        len = half;
    }
    return buf[0];
}

// LinearUp / LinearDown  linear.rs:26-35, 62-69;  LatchUp / LatchDown  latch.rs
template <int N>
OG_DEV void linear_up(float& prev, float x, float (&out)[N])
{
    const float n_inv = 1.0f / (float)N;
    const float delta = x - prev;
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = OG_FMA(delta, (float)i * n_inv, prev);
    prev = x;
}
template <int N>
OG_DEV float linear_down(const float (&xs)[N])
{
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) acc = acc + xs[i];
    return acc * (1.0f / (float)N);
}
template <int N>
OG_DEV void latch_up(float x, float (&out)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = x;
}

// ---------------------------------------------------------------------------
// Electric piano voice  examples/electric-piano/src/electric_piano_voice.rs
// One voice spans LPV = 8 lanes: lane l owns harmonics 4l .. 4l+3 (OG_HPL per lane) of AmplitudeSource
// (current/target/decay/release) and of OscillatorBank (complex phasor + rotation multiplier); per-voice
// scalars are replicated on the voice's lanes.  (Round 1 used one harmonic per lane: every lane then
// repeated the per-voice step/interpolation bookkeeping, and a wave carried only two voices.)
// ---------------------------------------------------------------------------
constexpr int EP_HARMONICS = 32;
constexpr int EP_LPV = EP_HARMONICS / OG_HPL;
constexpr uint32_t EP_INTERP_STEPS = 64;

// reference spectra :10-48
__device__ const float EP_VEL0[EP_HARMONICS] = {0.02f, 0.05f};
__device__ const float EP_VEL127[EP_HARMONICS] = {
    0.150869f,   0.385766f,   0.215543f,   0.117811f,   0.100411f,    0.0128637f,  0.0288844f,  0.00243388f,
    0.00963092f, 0.0035634f,  0.00256945f, 0.00184799f, 0.000399878f, 0.000660576f, 3.00995e-05f, 0.00021866f,
    9.33705e-05f, 0.000177973f, 0.0002545f, 0.000323602f, 0.000779045f, 0.000116569f, 0.000772873f, 0.000364486f,
    0.000248027f, 0.00018236f, 3.27292e-05f, 6.64988e-05f, 0.0f, 0.0f, 0.0f, 0.0f};

// Round 5: the read-mostly tables of AmplitudeSource are not register state any more.  `decay` and `release` (32 + 32 words
// per voice) are read only by the gate handler and `mult` (= released ? release : decay) only when a new ramp target is
// formed, once per 65 frames of a voice; held in registers they were what the three-waves-per-SIMD budget spilled around
// every chunk (76 registers, 144 B of scratch per lane).  Now decay / release stay in their state planes (the gate
// handler writes them there when a note starts and reads `release` back when it ends) and `mult` lives in an LDS column
// per lane: the frame loop keeps cur / tgt and the oscillator bank's four arrays, nothing else of OG_HPL words.
struct EpAmp {
    HarmV cur, tgt;          // this lane's harmonics
    uint32_t released, step; // per voice
    float velocity;
};

// the wave's LDS copy of (released ? release : decay), one column per lane (one-wave workgroups: column = lane).  Reached
// through this accessor, not through a pointer in EpAmp: the compiler then knows the address space (ds_read_b128 /
// ds_write_b128 with a 32-bit address) and no 64-bit generic pointer lives in registers across the frame loop.
OG_DEV float4 (&ep_mult_lds())[OG_HPL >= 4 ? OG_HPL / 4 : 1][OG_WAVE]
{
    __shared__ float4 ep_mult[OG_HPL >= 4 ? OG_HPL / 4 : 1][OG_WAVE];
    return ep_mult;
}

OG_DEV void ep_mult_put(const HarmV& m)
{
    const uint32_t lane = threadIdx.x % OG_WAVE;
#if OG_HPL >= 4
#pragma unroll
    for (int q = 0; q < OG_HPL / 4; ++q) ep_mult_lds()[q][lane] = make_float4(m.p[2 * q].x, m.p[2 * q].y, m.p[2 * q + 1].x, m.p[2 * q + 1].y);
#else
    ep_mult_lds()[0][lane] = make_float4(m.p[0].x, m.p[0].y, 0.0f, 0.0f);
#endif
}
OG_DEV HarmV ep_mult_get()
{
    const uint32_t lane = threadIdx.x % OG_WAVE;
    HarmV m;
#if OG_HPL >= 4
#pragma unroll
    for (int q = 0; q < OG_HPL / 4; ++q) {
        const float4 v = ep_mult_lds()[q][lane];
        m.p[2 * q] = og_f2{v.x, v.y};
        m.p[2 * q + 1] = og_f2{v.z, v.w};
    }
#else
    const float4 v = ep_mult_lds()[0][lane];
    m.p[0] = og_f2{v.x, v.y};
#endif
    return m;
}
OG_DEV HarmV ep_plane_get(const float* p)
{
    HarmV r;
#if OG_HPL >= 4
#pragma unroll
    for (int i = 0; i < OG_HPL / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        r.p[2 * i] = og_f2{q.x, q.y};
        r.p[2 * i + 1] = og_f2{q.z, q.w};
    }
#else
    const float2 q = *reinterpret_cast<const float2*>(p);
    r.p[0] = og_f2{q.x, q.y};
#endif
    return r;
}
OG_DEV void ep_plane_put(float* p, const HarmV& x)
{
#if OG_HPL >= 4
#pragma unroll
    for (int i = 0; i < OG_HPL / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(x.p[2 * i].x, x.p[2 * i].y, x.p[2 * i + 1].x, x.p[2 * i + 1].y);
#else
    *reinterpret_cast<float2*>(p) = make_float2(x.p[0].x, x.p[0].y);
#endif
}

// block start: bind the lane's planes and its LDS column, bring the table the next target will be formed with
OG_DEV void ep_amp_begin(EpAmp& a, const float* decay_plane, const float* release_plane, bool valid)
{
    HarmV m = harm_splat(0.0f);
    if (valid) m = ep_plane_get(a.released != 0u ? release_plane : decay_plane);
    ep_mult_put(m);
}

// on_gate :308-318 -> trigger_note :292-299 (get_decay :244-268, get_release :270-274,
// get_initial_amplitudes :276-290; note_pitch stays 60.0) or release_note :301-304.   h0 = first harmonic of the lane
// `decay` / `release`: this lane's OG_HPL words of the two state planes, formed by the caller where the event fires (no
// pointer lives in registers across the frame loop).  A lane beyond the bank's last voice gets its slot of the dump area
// instead (og::lane_plane_or_dump): a node-to-node event can reach its handler -- every lane runs the tick -- and it must
// not write through planes that are not its own.
OG_DEV void ep_amp_gate(EpAmp& a, float* decay_plane, float* release_plane, uint32_t h0, float v, float brightness, float velocity_scaling,
                        float decay_rate, float harmonic_decay, float key_scaling, float release_rate)
{
    if (v > 0.0f) {
        a.velocity = v;
        const float note = 60.0f;
        const float base_decay_rate = (100.0f - decay_rate) / 40000.0f;
        const float harmonic_scaling = 1.0f - ((100.0f - harmonic_decay) / 200000.0f);
        const float scaling_multiplier = (48.0f - note) / 12.0f;
        const float key_scaling_factor = scaling_multiplier * (key_scaling * 0.02f);
        const float adjusted_decay = (key_scaling_factor > 0.0f) ? 1.0f - (base_decay_rate / (1.0f + key_scaling_factor))
                                                                  : 1.0f - (base_decay_rate * (1.0f - key_scaling_factor));
        float scaling = 1.0f; // scaling_h = ((1 * hs) * hs) ... h sequentially rounded products
        for (uint32_t i = 0; i < h0; ++i) scaling *= harmonic_scaling;
        const float rel = 0.999f - ((100.0f - release_rate) / 1000.0f);
        float brightness_scaling = -0.2f + (0.8f * (brightness * 0.01f));
        brightness_scaling += v * velocity_scaling * 0.01f * 0.5f;
        float dec[OG_HPL], cur[OG_HPL];
#pragma unroll
        for (int j = 0; j < OG_HPL; ++j) {
            const uint32_t h = h0 + (uint32_t)j;
            dec[j] = adjusted_decay * scaling;
            scaling *= harmonic_scaling;
            float amp = (EP_VEL127[h] * v) + (EP_VEL0[h] * (1.0f - v));
            amp *= 1.0f + brightness_scaling * (float)h;
            cur[j] = amp;
        }
        const HarmV decay = harm_make(dec);
        ep_plane_put(decay_plane, decay); // self.decay = get_decay(..), self.release = get_release(..): the state planes
        ep_plane_put(release_plane, harm_splat(rel));
        ep_mult_put(decay);
        a.cur = harm_make(cur);
        a.released = 0u;
        a.step = 0u;
    } else {
        a.released = 1u;
        a.step = 0u;
        ep_mult_put(ep_plane_get(release_plane)); // (this lane's own earlier stores, if any, are visible to its loads)
    }
}

// AmplitudeSource::process :321-351 for the lane's harmonics, written without branches: with eight voices per wave
// some voice is next to a ramp boundary on most frames, so a divergent `if` costs both of its sides.
//   step == 0            -> new target = current * (released ? release : decay)       (select)
//   step < 64            -> current = current * (1 - t) + target * t, t = (step + 1) / 64; step += 1
//   step == 64           -> current = target; step = 0: the same lerp with t = 1 (x * 0 + y * 1 == y for the finite,
//                           non-negative amplitudes here)
// The table the next target uses (release or decay) only changes in the gate handler; it is read from LDS.
OG_DEV HarmV ep_amp_tick(EpAmp& a)
{
    const bool fresh = a.step == 0u;
    const bool ramping = a.step < EP_INTERP_STEPS;
    const float t = ramping ? (float)(a.step + 1u) / (float)EP_INTERP_STEPS : 1.0f;
    const float u = 1.0f - t;
#ifndef OG_EP_FRESH_BRANCH
#define OG_EP_FRESH_BRANCH 1
#endif
    // A new target is formed once per 65 frames of a voice.  With 16 voices in a wave no lane needs one on ~78 % of the
    // frames: one wave-uniform test skips the LDS reads, the products and the selects there (a per-lane `if` would cost
    // both sides).
    if (!OG_EP_FRESH_BRANCH || __any((int)fresh)) {
        const HarmV mult = ep_mult_get();
#pragma unroll
        for (int i = 0; i < OG_HPAIRS; ++i) {
            const og_f2 tn = f2_mul(a.cur.p[i], mult.p[i]);
            a.tgt.p[i].x = fresh ? tn.x : a.tgt.p[i].x;
            a.tgt.p[i].y = fresh ? tn.y : a.tgt.p[i].y;
        }
    }
#pragma unroll
    for (int i = 0; i < OG_HPAIRS; ++i) a.cur.p[i] = f2_fma(a.cur.p[i], u, f2_mul(a.tgt.p[i], t)); // current * (1 - t) + target * t
    a.step = ramping ? a.step + 1u : 0u;
    return a.cur;
}

struct EpBank {
    HarmV re, im, mre, mim; // this lane's harmonics
    float last_frequency;   // per voice
    bool mul_dirty;         // the rotation multipliers were rewritten in this block (read-mostly state)
};

OG_DEV void ep_bank_gate(EpBank& b, float v) // on_gate :115-122
{
    if (v > 0.0f) {
        b.re = harm_splat(1.0f);
        b.im = harm_splat(0.0f);
    }
}

// The rotation multipliers of the lane's harmonics (update_multipliers :130-146): cos/sin of 2*pi*f_h/sr through the
// bit-exact libm restatement (the rotation is applied every sample, an ulp here is a drift there).  Out of line on
// purpose: this is a few hundred instructions that run only when a note changes the frequency; inlined into the voice
// kernel they cost the hot loop its registers (the kernel spilled).  out = mre[OG_HPL] then mim[OG_HPL].
// Results go through LDS (one column per lane), not through a stack array: an array whose address escapes into an
// out-of-line function lives in scratch memory, and a kernel that touches scratch at all pays the scratch set-up at
// every wave launch.
#ifdef OG_EP_TABLES_INLINE // experiment switch (profiles/r03_epiano_lanes.md)
__device__ __forceinline__ void ep_bank_tables(
#else
__device__ __attribute__((noinline)) void ep_bank_tables(
#endif
    float frequency, float sr, uint32_t h0, float* __restrict__ out, uint32_t stride)
{
    const float nyquist = sr * 0.5f;
#pragma unroll 1
    for (int j = 0; j < OG_HPL; ++j) {
        const float harmonic_freq = frequency * (float)(h0 + (uint32_t)j + 1u);
        float c = 1.0f, s = 0.0f;
        if (harmonic_freq < nyquist) {
            const float angle = 2.0f * 3.14159274101257324f * harmonic_freq / sr;
            c = og_cosf_exact(angle);
            s = og_sinf_exact(angle);
        }
        out[(uint32_t)j * stride] = c;
        out[(uint32_t)(OG_HPL + j) * stride] = s;
    }
}

// update_multipliers :126-150 behind the frequency-change test of process() :155-158.  The caller runs
// it every tick, or -- when the frequency can only change through per-voice value events -- in
// derive() (block start and after such an event), which is when the reference's test can fire.
OG_DEV void ep_bank_update(EpBank& b, uint32_t h0, float frequency, float sr)
{
    if (frequency > 0.0f && !(fabsf(b.last_frequency - frequency) < 0.01f)) {
        b.last_frequency = frequency;
        __shared__ float ep_tab[2 * OG_HPL][OG_WAVE]; // (one-wave workgroups: column = lane)
        float* tab = &ep_tab[0][threadIdx.x % OG_WAVE];
        ep_bank_tables(frequency, sr, h0, tab, OG_WAVE);
#pragma unroll
        for (int i = 0; i < OG_HPAIRS; ++i) {
            b.mre.p[i] = og_f2{tab[(2 * i) * OG_WAVE], tab[(2 * i + 1) * OG_WAVE]};
            b.mim.p[i] = og_f2{tab[(OG_HPL + 2 * i) * OG_WAVE], tab[(OG_HPL + 2 * i + 1) * OG_WAVE]};
        }
        b.re = harm_splat(1.0f);
        b.im = harm_splat(0.0f);
        b.mul_dirty = true;
    }
}

// OscillatorBank::process :159-169.  The lane folds its own harmonics in index order.  VOICE_SUM: also fold the
// voice's lanes (reference: one sequential f32 fold over the 32 harmonics; here a butterfly inside the 8-lane
// group -- re-association only) and return the voice output on every lane.  Otherwise return this lane's share:
// used when the output feeds nothing but the mix bus, whose reduction adds the lanes anyway.
template <bool VOICE_SUM>
OG_DEV float ep_bank_tick(EpBank& b, const HarmV& amp)
{
    // Complex::mul :66-72 on two harmonics per instruction; the lane folds its pairs (h0 + h2 + .., h1 + h3 + ..), then
    // the two halves
    og_f2 acc = og_f2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < OG_HPAIRS; ++i) {
        const og_f2 re = f2_sub(f2_mul(b.re.p[i], b.mre.p[i]), f2_mul(b.im.p[i], b.mim.p[i]));
        const og_f2 im = f2_add(f2_mul(b.re.p[i], b.mim.p[i]), f2_mul(b.im.p[i], b.mre.p[i]));
        b.re.p[i] = re;
        b.im.p[i] = im;
        acc = (i == 0) ? f2_mul(im, amp.p[i]) : f2_fma(im, amp.p[i], acc); // output += im * amplitude (the sum, not the rotation)
    }
    float s = acc.x + acc.y;
    if (VOICE_SUM) {
#pragma unroll
        for (int m = 1; m < EP_LPV; m <<= 1) s += __shfl_xor(s, m);
    }
    return s * 3.0f;
}

} // namespace og
