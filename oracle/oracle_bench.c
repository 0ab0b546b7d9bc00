/*
 * oracle_bench.c -- CPU ORACLE timing leg (TEST/BENCH INFRASTRUCTURE ONLY).
 *
 * Times the scalar restatement of the reference CPU path on the host cores:
 * same per-sample, voice-minor tick order as the reference's generated
 * process_block (oscen-graph-compiler/src/codegen/mod.rs:755-873), AoS voices.
 * Multi-threading = static partition of the voices, one private sub-bank per
 * thread (the reference itself is single-threaded per graph).
 */
#define _POSIX_C_SOURCE 200809L
#include "oscen_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int kind;
    uint32_t lo, hi, frames_total, block;
    uint64_t seed;
    double checksum;
} worker_arg;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *p)
{
    worker_arg *a = (worker_arg *)p;
    uint32_t n = a->hi - a->lo;
    oo_bank *b = oo_bank_create(a->kind, n);
    oo_bank_init(b, 48000.0f);
    oo_note_plan *plans = (oo_note_plan *)malloc(sizeof(oo_note_plan) * n);
    for (uint32_t i = 0; i < n; ++i) {
        oo_note_plan_for_voice(a->seed, a->lo + i, &plans[i]);
        oo_bank_set_voice_frequency(b, i, plans[i].frequency);
    }
    float out[OO_MAX_BLOCK * 2];
    double cs = 0.0;
    uint32_t ch = oo_bank_channels(b);
    for (uint32_t f0 = 0; f0 < a->frames_total; f0 += a->block) {
        uint32_t frames = a->frames_total - f0 < a->block ? a->frames_total - f0 : a->block;
        for (uint32_t i = 0; i < n; ++i) {
            const oo_note_plan *pl = &plans[i];
            float vel = oo_midi_velocity_to_gate(pl->velocity);
            if (pl->on_frame >= f0 && pl->on_frame < f0 + frames)
                oo_bank_push_event(b, i, pl->on_frame - f0, OO_EV_GATE, vel);
            if (pl->off_frame >= f0 && pl->off_frame < f0 + frames)
                oo_bank_push_event(b, i, pl->off_frame - f0, OO_EV_GATE, 0.0f);
            if (pl->retrig_frame >= f0 && pl->retrig_frame < f0 + frames)
                oo_bank_push_event(b, i, pl->retrig_frame - f0, OO_EV_GATE, vel);
        }
        oo_bank_process_block(b, frames, out, NULL, 0, NULL);
        for (uint32_t k = 0; k < frames * ch; ++k) cs += (double)out[k];
    }
    a->checksum = cs;
    free(plans);
    oo_bank_destroy(b);
    return NULL;
}

double oo_bank_bench(int kind, uint32_t n_voices, uint32_t frames_total, uint32_t block, uint32_t n_threads,
                     uint64_t seed, double *checksum)
{
    if (n_threads == 0) n_threads = 1;
    if (n_threads > n_voices) n_threads = n_voices;
    if (block == 0 || block > OO_MAX_BLOCK) block = 256;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    worker_arg *args = (worker_arg *)calloc(n_threads, sizeof(worker_arg));
    double t0 = now_s();
    for (uint32_t t = 0; t < n_threads; ++t) {
        args[t].kind = kind;
        args[t].lo = (uint32_t)((uint64_t)n_voices * t / n_threads);
        args[t].hi = (uint32_t)((uint64_t)n_voices * (t + 1) / n_threads);
        args[t].frames_total = frames_total;
        args[t].block = block;
        args[t].seed = seed;
        pthread_create(&th[t], NULL, worker, &args[t]);
    }
    double cs = 0.0;
    for (uint32_t t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        cs += args[t].checksum;
    }
    double t1 = now_s();
    if (checksum) *checksum = cs;
    free(th);
    free(args);
    return t1 - t0;
}
