/* The call sequence of the Rust shim (bindings/rust/oscen-gpu/src/lib.rs: from_dsl -> init -> process_block -> drop)
 * written in C against include/oscen_gpu.h, with guard words around every buffer the library reads or writes.
 *
 * The Rust crate cannot be compiled in this image (no rustc); this program is what holds its buffer contract to the
 * library: `out_block` must take frames x og_channels() floats (up to OG_MAX_BUS_CHANNELS = 4 channels -- round 3's
 * shim had 2 x 512 and a 4-channel graph wrote 1 024 floats past it), a Frame<N> stream input hands
 * og_set_stream_block n x N floats (round 3: a 512-float block for a Frame<2> input -> out-of-bounds read).
 *
 * Generated-struct items this stands for: `pub <out>_block: [F; 512]`, `pub <stream_in>_block: [F; 512]`,
 * process_block(frames)  (oscen-graph-compiler/src/codegen/mod.rs:1196,1220,1306); BlockRender::render
 * (oscen-lib/src/graph/offline.rs:19-113).
 *
 * Built by tests/test_capi_c.py with gcc -std=c99 -Wall -Wextra -Werror -pedantic; the CPU test compiles and links
 * it (and runs it with --no-device: argument checks only), the -m gpu test runs it on the device. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oscen_gpu.h"

#define MAX_BLOCK 512
#define MAX_CH 4 /* OG_MAX_BUS_CHANNELS of the engine; the shim's out_block is [f32; 4 * 512] */
#define GUARD 64
#define GUARD_WORD 0x7fc0dead /* a NaN payload nothing on the path produces */

#define CHECK(call)                                                                                    \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ < 0) {                                                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, og_last_error());                            \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

typedef struct {
    uint32_t* base; /* GUARD words | n floats | GUARD words */
    size_t n;
} guarded;

static int guarded_new(guarded* g, size_t n)
{
    size_t i;
    g->base = (uint32_t*)malloc((n + 2 * GUARD) * sizeof(uint32_t));
    g->n = n;
    if (!g->base) return -1;
    for (i = 0; i < n + 2 * GUARD; ++i) g->base[i] = GUARD_WORD;
    return 0;
}
static float* guarded_data(guarded* g) { return (float*)(void*)(g->base + GUARD); }
static int guarded_intact(const guarded* g)
{
    size_t i;
    for (i = 0; i < GUARD; ++i)
        if (g->base[i] != GUARD_WORD || g->base[GUARD + g->n + i] != GUARD_WORD) return 0;
    return 1;
}
/* how many of the first `n` payload words were written (are no longer the guard pattern) */
static size_t guarded_written(const guarded* g, size_t n)
{
    size_t i, k = 0;
    for (i = 0; i < n; ++i) k += g->base[GUARD + i] != GUARD_WORD;
    return k;
}

static const char* FOUR_OUT =
    "name: FourOut;\n"
    "input frequency: value = 220.0;\n"
    "output a: stream;\noutput b: stream;\noutput c: stream;\noutput d: stream;\n"
    "nodes { o = Oscillator::sine(220.0, 0.5); p = Oscillator::saw(110.0, 0.25); }\n"
    "connections { frequency -> o.frequency; o.output -> a; p.output -> b; o.output * 0.5 -> c; o.output + p.output -> d; }\n";

static const char* STEREO_IN =
    "name: StereoThrough;\n"
    "input stream dry: Frame<2>;\n"
    "output stream wet: Frame<2>;\n"
    "nodes { f = TptFilter::<Frame<2>>::new(18000.0, 0.7); }\n"
    "connections { dry -> f.input; f.output -> wet; }\n";

/* from_dsl(): og_graph_parse -> og_create(device) -> og_graph_free */
static int open_engine(const char* text, const char* per_voice, uint32_t voices, int device, og_engine** e)
{
    og_graph_desc* g = NULL;
    int rc;
    CHECK(og_graph_parse(text, per_voice, &g));
    rc = og_create(g, voices, device, e);
    og_graph_free(g);
    if (rc < 0) {
        fprintf(stderr, "og_create -> %d: %s\n", rc, og_last_error());
        return rc;
    }
    return 0;
}

static int four_channel_graph(int device)
{
    og_engine* e = NULL;
    guarded out;
    uint32_t ch, frames = MAX_BLOCK, f;
    float* o;
    if (open_engine(FOUR_OUT, "frequency", 100, device, &e)) return 1;
    ch = og_channels(e);
    if (ch != 4 || og_num_stream_inputs(e) != 0) {
        fprintf(stderr, "FourOut: %u channels, %u stream inputs\n", ch, og_num_stream_inputs(e));
        return 1;
    }
    CHECK(og_init(e, 48000.0f));
    if (guarded_new(&out, (size_t)MAX_CH * MAX_BLOCK)) return 1; /* the shim's out_block */
    CHECK(og_process_block(e, frames, guarded_data(&out)));
    if (!guarded_intact(&out)) {
        fprintf(stderr, "FourOut: og_process_block wrote outside frames x channels floats\n");
        return 1;
    }
    if (guarded_written(&out, (size_t)frames * ch) != (size_t)frames * ch) {
        fprintf(stderr, "FourOut: not every sample of the block was written\n");
        return 1;
    }
    /* channel c = 0.5 * channel a, channel d = a + b summed per voice: hold the interleaving to the declaration order */
    o = guarded_data(&out);
    for (f = 0; f < frames; ++f) {
        const float a = o[f * 4 + 0], b = o[f * 4 + 1], c = o[f * 4 + 2], d = o[f * 4 + 3];
        if (fabsf(c - 0.5f * a) > 1e-3f || fabsf(d - (a + b)) > 1e-3f) {
            fprintf(stderr, "FourOut: frame %u: a %g b %g c %g d %g\n", f, (double)a, (double)b, (double)c, (double)d);
            return 1;
        }
    }
    /* a short block writes frames x channels floats and nothing else */
    {
        size_t i;
        for (i = 0; i < out.n; ++i) out.base[GUARD + i] = GUARD_WORD;
        CHECK(og_process_block(e, 7, guarded_data(&out)));
        if (!guarded_intact(&out) || guarded_written(&out, out.n) != 7u * ch) {
            fprintf(stderr, "FourOut: a 7-frame block wrote %lu words\n", (unsigned long)guarded_written(&out, out.n));
            return 1;
        }
    }
    og_destroy(e);
    free(out.base);
    return 0;
}

static int stereo_stream_input_graph(int device)
{
    og_engine* e = NULL;
    guarded in, out;
    int dry;
    uint32_t nin, frames = MAX_BLOCK, f, voices = 3;
    float *x, *o;
    if (open_engine(STEREO_IN, NULL, voices, device, &e)) return 1;
    dry = og_input_index(e, "dry");
    if (dry < 0) return 1;
    nin = og_stream_input_channels(e, (uint32_t)dry);
    if (og_num_stream_inputs(e) != 1 || nin != 2 || og_channels(e) != 2) {
        fprintf(stderr, "StereoThrough: %u stream inputs, %u channels in, %u out\n", og_num_stream_inputs(e), nin, og_channels(e));
        return 1;
    }
    CHECK(og_init(e, 48000.0f));
    /* the shim's stream_in_blocks[k] is sized frames x og_stream_input_channels floats: EXACTLY that much, guarded, so a
     * library that read more than n x N floats would run into memory this program does not own (ASan / valgrind see
     * it; the guard words make sure at least that nothing was written there) */
    if (guarded_new(&in, (size_t)frames * nin) || guarded_new(&out, (size_t)MAX_CH * MAX_BLOCK)) return 1;
    x = guarded_data(&in);
    for (f = 0; f < frames; ++f) {
        x[2 * f] = 0.25f;      /* DC per channel: a lowpass passes it */
        x[2 * f + 1] = -0.125f;
    }
    CHECK(og_set_stream_block(e, (uint32_t)dry, x, frames));
    CHECK(og_process_block(e, frames, guarded_data(&out)));
    if (!guarded_intact(&in) || !guarded_intact(&out)) {
        fprintf(stderr, "StereoThrough: guard words damaged\n");
        return 1;
    }
    o = guarded_data(&out);
    /* the filter has settled by the end of the block: left = voices * 0.25, right = voices * -0.125 */
    if (fabsf(o[2 * (frames - 1)] - 0.25f * (float)voices) > 1e-2f || fabsf(o[2 * (frames - 1) + 1] + 0.125f * (float)voices) > 1e-2f) {
        fprintf(stderr, "StereoThrough: last frame %g %g\n", (double)o[2 * (frames - 1)], (double)o[2 * (frames - 1) + 1]);
        return 1;
    }
    /* more frames than a block holds is an error, not an overrun */
    if (og_set_stream_block(e, (uint32_t)dry, x, MAX_BLOCK + 1) >= 0 || og_process_block(e, MAX_BLOCK + 1, guarded_data(&out)) >= 0) {
        fprintf(stderr, "StereoThrough: an oversized block was accepted\n");
        return 1;
    }
    og_destroy(e);
    free(in.base);
    free(out.base);
    return 0;
}

/* without a device: every entry point of the sequence rejects bad arguments with a code (no crash, no exception) */
static int argument_checks(void)
{
    og_graph_desc* g = NULL;
    og_engine* e = NULL;
    float buf[8];
    if (og_graph_parse(NULL, NULL, &g) >= 0 || og_graph_parse("nodes {", NULL, &g) >= 0) return 1;
    if (og_create(NULL, 1, 0, &e) >= 0) return 1;
    if (og_init(NULL, 48000.0f) >= 0 || og_process_block(NULL, 1, buf) >= 0 || og_set_stream_block(NULL, 0, buf, 1) >= 0) return 1;
    if (og_channels(NULL) != 0 || og_num_stream_inputs(NULL) != 0) return 1;
    if (og_midi_send(NULL, (const uint8_t*)"\x90\x40\x7f", 3, 0) >= 0) return 1;
    CHECK(og_graph_parse(FOUR_OUT, "frequency", &g));
    og_graph_free(g);
    CHECK(og_graph_parse(STEREO_IN, NULL, &g));
    og_graph_free(g);
    og_destroy(NULL);
    return 0;
}

int main(int argc, char** argv)
{
    int device = 0;
    if (argc > 1 && strcmp(argv[1], "--no-device") == 0) {
        if (argument_checks()) return 1;
        puts("shim_sequence: argument checks ok");
        return 0;
    }
    if (argc > 1) device = atoi(argv[1]); /* the shim takes the device id as a parameter (round 3 hard-coded 0) */
    if (argument_checks() || four_channel_graph(device) || stereo_stream_input_graph(device)) return 1;
    puts("shim_sequence: ok");
    return 0;
}
