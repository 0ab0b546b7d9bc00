/*
 * oracle_bench.c -- CPU ORACLE timing / full-size reference leg (TEST/BENCH INFRASTRUCTURE ONLY).
 *
 * Runs the scalar restatement of the reference CPU path on the host cores:
 * same per-sample, voice-minor tick order as the reference's generated
 * process_block (oscen-graph-compiler/src/codegen/mod.rs:755-873), AoS voices.
 * Multi-threading = static partition of the voices (the reference itself is
 * single-threaded per graph).  Inside its range a thread runs
 * sub-banks of `group` voices: group = 8 is the reference's own shape
 * (`voices = [FMVoice; 8]` per graph, examples/fm-synth/src/lib.rs:68-73), whose
 * working set stays in L1/L2; group = 0 makes one sub-bank of the whole range
 * (every voice touched once per sample: cache-hostile, kept for comparison).
 * Results do not depend on the grouping: voices are independent and the
 * per-frame sums are accumulated in f64.
 */
#define _POSIX_C_SOURCE 200809L
#include "oscen_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int kind;
    uint32_t lo, hi, frames_total, block, group, span;
    uint64_t seed;
    double checksum;
    double *mono64, *abs64; /* [frames_total] or NULL */
} worker_arg;

static int g_fold = 0;
void oo_bench_set_fold(int fold) { g_fold = fold; }

/* Scenario of the multi-threaded renders (full-size parity tests of SURVEY 8(d) config 2's VARIANT): broadcast
 * parameters set immediately before the first frame, and setter calls `set_<n>(v)` (ramped inputs ramp over their
 * declared length, examples/fm-synth/src/fm_voice.rs `[ramp: 2205]`) made right before frame `at_frame` -- the render
 * cuts its block there, as a host calling the setter between two process_block calls does. */
#define OO_SCN_MAX 16
static struct { uint32_t param; float value; } g_scn_set[OO_SCN_MAX];
static struct { uint32_t param; float value; uint32_t at_frame; } g_scn_ramp[OO_SCN_MAX];
static uint32_t g_scn_nset = 0, g_scn_nramp = 0;
void oo_bench_scenario_clear(void) { g_scn_nset = g_scn_nramp = 0; }
int oo_bench_scenario_set(uint32_t param, float value)
{
    if (g_scn_nset >= OO_SCN_MAX) return -1;
    g_scn_set[g_scn_nset].param = param, g_scn_set[g_scn_nset].value = value, ++g_scn_nset;
    return 0;
}
int oo_bench_scenario_ramp(uint32_t param, float value, uint32_t at_frame)
{
    if (g_scn_nramp >= OO_SCN_MAX) return -1;
    g_scn_ramp[g_scn_nramp].param = param, g_scn_ramp[g_scn_nramp].value = value, g_scn_ramp[g_scn_nramp].at_frame = at_frame;
    ++g_scn_nramp;
    return 0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void render_group(worker_arg *a, uint32_t lo, uint32_t hi, oo_note_events *plans, double *cs)
{
    const uint32_t n = hi - lo;
    oo_bank *b = oo_bank_create(a->kind, n);
    oo_bank_init(b, 48000.0f);
    for (uint32_t i = 0; i < n; ++i) {
        oo_note_events_for_voice(a->seed, lo + i, a->span, g_fold, &plans[i]);
        oo_bank_set_voice_frequency(b, i, plans[i].frequency);
    }
    const int gated = a->kind != OO_BANK_SAT4X && a->kind != OO_BANK_SAT1X;
    float out[OO_MAX_BLOCK * 2];
    const uint32_t ch = oo_bank_channels(b);
    for (uint32_t k = 0; k < g_scn_nset; ++k) oo_bank_set_value_immediate(b, g_scn_set[k].param, g_scn_set[k].value);
    for (uint32_t f0 = 0, frames = 0; f0 < a->frames_total; f0 += frames) {
        frames = a->frames_total - f0 < a->block ? a->frames_total - f0 : a->block;
        for (uint32_t k = 0; k < g_scn_nramp; ++k) {
            if (g_scn_ramp[k].at_frame == f0) oo_bank_set_value(b, g_scn_ramp[k].param, g_scn_ramp[k].value);
            else if (g_scn_ramp[k].at_frame > f0 && g_scn_ramp[k].at_frame < f0 + frames) frames = g_scn_ramp[k].at_frame - f0;
        }
        for (uint32_t i = 0; gated && i < n; ++i) {
            const oo_note_events *pl = &plans[i];
            for (uint32_t k = 0; k < pl->n; ++k)
                if (pl->frame[k] >= f0 && pl->frame[k] < f0 + frames)
                    oo_bank_push_event(b, i, pl->frame[k] - f0, OO_EV_GATE, pl->value[k]);
        }
        oo_bank_process_block(b, frames, out, NULL, 0, NULL);
        for (uint32_t k = 0; k < frames * ch; ++k) *cs += (double)out[k];
        if (a->mono64) {
            const double *m = oo_bank_last_bus_f64(b), *ab = oo_bank_last_abs_f64(b);
            for (uint32_t k = 0; k < frames; ++k) {
                a->mono64[f0 + k] += m[k];
                a->abs64[f0 + k] += ab[k];
            }
        }
    }
    oo_bank_destroy(b);
}

static void *worker(void *p)
{
    worker_arg *a = (worker_arg *)p;
    const uint32_t n = a->hi - a->lo;
    const uint32_t g = (a->group == 0 || a->group > n) ? n : a->group;
    oo_note_events *plans = (oo_note_events *)malloc(sizeof(oo_note_events) * (g ? g : 1));
    double cs = 0.0;
    for (uint32_t lo = a->lo; lo < a->hi; lo += g) render_group(a, lo, lo + g < a->hi ? lo + g : a->hi, plans, &cs);
    a->checksum = cs;
    free(plans);
    return NULL;
}

static double run_mt(int kind, uint32_t first_voice, uint32_t n_voices, uint32_t frames_total, uint32_t block,
                     uint32_t n_threads, uint32_t group, uint64_t seed, uint32_t span, double *mono64, double *abs64,
                     double *checksum)
{
    if (n_threads == 0) n_threads = 1;
    if (n_threads > n_voices) n_threads = n_voices;
    if (block == 0 || block > OO_MAX_BLOCK) block = 256;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    worker_arg *args = (worker_arg *)calloc(n_threads, sizeof(worker_arg));
    const double t0 = now_s();
    for (uint32_t t = 0; t < n_threads; ++t) {
        args[t].kind = kind;
        args[t].lo = first_voice + (uint32_t)((uint64_t)n_voices * t / n_threads);
        args[t].hi = first_voice + (uint32_t)((uint64_t)n_voices * (t + 1) / n_threads);
        args[t].frames_total = frames_total;
        args[t].block = block;
        args[t].group = group;
        args[t].span = span;
        args[t].seed = seed;
        if (mono64) {
            args[t].mono64 = (double *)calloc(frames_total, sizeof(double));
            args[t].abs64 = (double *)calloc(frames_total, sizeof(double));
        }
        pthread_create(&th[t], NULL, worker, &args[t]);
    }
    double cs = 0.0;
    if (mono64) {
        memset(mono64, 0, sizeof(double) * frames_total);
        if (abs64) memset(abs64, 0, sizeof(double) * frames_total);
    }
    for (uint32_t t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        cs += args[t].checksum;
        if (mono64) {
            for (uint32_t k = 0; k < frames_total; ++k) {
                mono64[k] += args[t].mono64[k];
                if (abs64) abs64[k] += args[t].abs64[k];
            }
            free(args[t].mono64);
            free(args[t].abs64);
        }
    }
    const double t1 = now_s();
    if (checksum) *checksum = cs;
    free(th);
    free(args);
    return t1 - t0;
}

double oo_bank_bench(int kind, uint32_t n_voices, uint32_t frames_total, uint32_t block, uint32_t n_threads,
                     uint64_t seed, double *checksum)
{
    return run_mt(kind, 0, n_voices, frames_total, block, n_threads, 0, seed, 0, NULL, NULL, checksum);
}

double oo_bank_bench_grouped(int kind, uint32_t n_voices, uint32_t frames_total, uint32_t block, uint32_t n_threads,
                             uint32_t group, uint64_t seed, uint32_t span, double *checksum)
{
    return run_mt(kind, 0, n_voices, frames_total, block, n_threads, group, seed, span, NULL, NULL, checksum);
}

double oo_bank_render_mt(int kind, uint32_t first_voice, uint32_t n_voices, uint32_t frames_total, uint32_t block,
                         uint32_t n_threads, uint32_t group, uint64_t seed, uint32_t span, double *mono64,
                         double *abs64)
{
    return run_mt(kind, first_voice, n_voices, frames_total, block, n_threads, group, seed, span, mono64, abs64, NULL);
}

/* ---- the reference's own criterion shapes (oscen-lib/benches/static_vs_runtime.rs:68-116), single thread ----
 * out[0]: ns per StaticSimpleGraph::process()        ("simple_graph/static",   :68-82,  init(44100))
 * out[1]: ns per StaticComplexGraph::process()       ("complex_graph/static",  :84-98)
 * out[2]: ns per 512 x StaticSimpleGraph::process()  ("batch_processing/static_graph_512", :100-116)
 * out[3]: seconds for BASELINE config 1: one FMVoice bank of 1 voice, 48 000 frames at 48 kHz in blocks of 256,
 *         note-on at frame 0 (velocity 100/127, 440 Hz), note-off at frame 24 000
 * Each shape loops for about `budget_s` seconds; a volatile sink stands in for criterion's black_box. */
static volatile float g_sink;
void oo_criterion_shapes(double budget_s, double *out)
{
    if (!(budget_s > 0.0)) budget_s = 0.2;
    {
        oo_static_simple g;
        oo_static_simple_new(&g);
        oo_static_simple_init(&g, 44100.0f);
        uint64_t it = 0;
        const double t0 = now_s();
        double t1 = t0;
        while (t1 - t0 < budget_s) {
            for (int k = 0; k < 4096; ++k) {
                oo_static_simple_process(&g);
                g_sink = g.gain.output;
            }
            it += 4096;
            t1 = now_s();
        }
        out[0] = (t1 - t0) * 1e9 / (double)it;
    }
    {
        oo_static_complex g;
        oo_static_complex_new(&g);
        oo_static_complex_init(&g, 44100.0f);
        uint64_t it = 0;
        const double t0 = now_s();
        double t1 = t0;
        while (t1 - t0 < budget_s) {
            for (int k = 0; k < 4096; ++k) {
                oo_static_complex_process(&g);
                g_sink = g.vca.output;
            }
            it += 4096;
            t1 = now_s();
        }
        out[1] = (t1 - t0) * 1e9 / (double)it;
    }
    {
        oo_static_simple g;
        oo_static_simple_new(&g);
        oo_static_simple_init(&g, 44100.0f);
        uint64_t it = 0;
        const double t0 = now_s();
        double t1 = t0;
        while (t1 - t0 < budget_s) {
            for (int r = 0; r < 16; ++r) {
                for (int k = 0; k < 512; ++k) {
                    oo_static_simple_process(&g);
                    g_sink = g.gain.output;
                }
                it += 1;
            }
            t1 = now_s();
        }
        out[2] = (t1 - t0) * 1e9 / (double)it;
    }
    {
        double best = 1e30;
        const double t_begin = now_s();
        do {
            oo_bank *b = oo_bank_create(OO_BANK_FM, 1);
            oo_bank_init(b, 48000.0f);
            oo_bank_set_voice_frequency(b, 0, 440.0f);
            float bus[OO_MAX_BLOCK * 2];
            const double t0 = now_s();
            for (uint32_t f0 = 0; f0 < 48000; f0 += 256) {
                const uint32_t frames = 48000 - f0 < 256 ? 48000 - f0 : 256;
                if (f0 == 0) oo_bank_push_event(b, 0, 0, OO_EV_GATE, 100.0f / 127.0f);
                if (24000 >= f0 && 24000 < f0 + frames) oo_bank_push_event(b, 0, 24000 - f0, OO_EV_GATE, 0.0f);
                oo_bank_process_block(b, frames, bus, NULL, 0, NULL);
                g_sink = bus[frames - 1];
            }
            const double dt = now_s() - t0;
            if (dt < best) best = dt;
            oo_bank_destroy(b);
        } while (now_s() - t_begin < budget_s);
        out[3] = best;
    }
}
