// og_midi.cpp -- host-side MIDI front end feeding the voice bank: the control-rate nodes that sit
// immediately upstream of the hot path in the reference's poly wrapper
// (examples/fm-synth/src/lib.rs:68-89):
//   MidiParser        oscen-lib/src/midi.rs:126-225   raw bytes -> NoteOn{note, velocity/127} / NoteOff
//   VoiceAllocator<N> oscen-lib/src/voice_allocator.rs:46-136   LRU, released voices stolen first
//   MidiVoiceHandler  oscen-lib/src/midi.rs:40-122    -> per-voice `frequency` value + `gate` event
// with N lifted from the reference's MAX_VOICES = 24 to the engine's voice count.  Messages are
// queued with their frame offset and applied in frame order when the next block starts, like the
// generated process_block sorts its staged events (codegen/mod.rs:782-799).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <functional>
#include <string>
#include <vector>

#include "../../include/oscen_gpu.h"
#include "og_abi.h"

struct og_midi {
    og_engine* engine = nullptr;
    og_cluster* cluster = nullptr; // or: a multi-GPU bank (voices = GLOBAL voice ids, routed to their shard by the cluster)
    uint32_t n = 0;
    int freq_input = -1, gate_input = -1;
    struct Voice {
        bool active = false, released = false;
        int note = -1; // Option<u8>
        uint32_t age = 0;
        int handler_note = -1; // MidiVoiceHandler.current_note
    };
    std::vector<Voice> voices;
    uint32_t current_age = 0;
    // The reference scans all voices per note (24 of them).  With N = the bank size the same decisions come from
    // three indexes of FIXED size (one slot per voice each; nothing is allocated, rebuilt or compacted while playing,
    // so a message costs O(log N) at worst whatever came before it -- the earlier lazy-deletion heaps grew with every
    // note and paid for it with multi-millisecond reallocations at large N):
    //  * voices become active in index order and never turn inactive again (release keeps `active`,
    //    voice_allocator.rs:101-108), so "first inactive voice" is a counter;
    //  * stealing = min over (released ? 0 : 1, age).  Ages are handed out in increasing order at allocation, so the
    //    HELD voices in age order are a FIFO: a doubly linked list through the voices (append on allocation, unlink on
    //    release, the head is the oldest).  The RELEASED voices enter in any age order and leave only by being
    //    allocated, oldest first: a binary min-heap by age in which every entry is live;
    //  * find_voice_for_note = lowest index among the held voices playing that note: a min-heap of voice indices per
    //    note with the position of every voice kept beside it (a held voice is in exactly one of them), so that a voice
    //    leaves its note's heap the moment it is released or stolen.
    uint32_t n_fresh = 0; // voices [0, n_fresh) are active
    static constexpr uint32_t NIL = 0xFFFFFFFFu;
    std::vector<uint32_t> held_prev, held_next; // age-ordered list of the held voices
    uint32_t held_head = NIL, held_tail = NIL;
    void held_append(uint32_t i)
    {
        held_prev[i] = held_tail;
        held_next[i] = NIL;
        if (held_tail != NIL) held_next[held_tail] = i;
        else held_head = i;
        held_tail = i;
    }
    void held_unlink(uint32_t i)
    {
        const uint32_t p = held_prev[i], q = held_next[i];
        if (p != NIL) held_next[p] = q;
        else held_head = q;
        if (q != NIL) held_prev[q] = p;
        else held_tail = p;
    }
    struct AgeEntry {
        uint32_t age, voice;
    };
    std::vector<AgeEntry> released; // min-heap by age, at most one entry per voice (reserved for N at creation)
    void released_push(AgeEntry e)
    {
        size_t k = released.size();
        released.push_back(e);
        while (k > 0) {
            const size_t up = (k - 1) / 2;
            if (released[up].age <= e.age) break;
            released[k] = released[up];
            k = up;
        }
        released[k] = e;
    }
    uint32_t released_pop()
    {
        const uint32_t voice = released.front().voice;
        const AgeEntry e = released.back();
        released.pop_back();
        const size_t m = released.size();
        if (m) {
            size_t k = 0;
            for (;;) {
                size_t c = 2 * k + 1;
                if (c >= m) break;
                if (c + 1 < m && released[c + 1].age < released[c].age) c += 1;
                if (released[c].age >= e.age) break;
                released[k] = released[c];
                k = c;
            }
            released[k] = e;
        }
        return voice;
    }
    std::vector<uint32_t> by_note[256]; // min-heaps of voice indices
    std::vector<uint32_t> note_pos;     // where voice i sits in by_note[its note]
    void note_place(std::vector<uint32_t>& h, size_t k, uint32_t i)
    {
        h[k] = i;
        note_pos[i] = (uint32_t)k;
    }
    void note_up(std::vector<uint32_t>& h, size_t k, uint32_t i)
    {
        while (k > 0) {
            const size_t up = (k - 1) / 2;
            if (h[up] <= i) break;
            note_place(h, k, h[up]);
            k = up;
        }
        note_place(h, k, i);
    }
    void note_down(std::vector<uint32_t>& h, size_t k, uint32_t i)
    {
        const size_t m = h.size();
        for (;;) {
            size_t c = 2 * k + 1;
            if (c >= m) break;
            if (c + 1 < m && h[c + 1] < h[c]) c += 1;
            if (h[c] >= i) break;
            note_place(h, k, h[c]);
            k = c;
        }
        note_place(h, k, i);
    }
    void note_insert(uint8_t note, uint32_t i)
    {
        auto& h = by_note[note];
        h.push_back(i);
        note_up(h, h.size() - 1, i);
    }
    void note_erase(uint8_t note, uint32_t i)
    {
        auto& h = by_note[note];
        const size_t k = note_pos[i];
        const uint32_t last = h.back();
        h.pop_back();
        if (k == h.size()) return;
        if (k > 0 && h[(k - 1) / 2] > last) note_up(h, k, last);
        else note_down(h, k, last);
    }
    void size_indexes()
    {
        voices.resize(n);
        held_prev.assign(n, NIL);
        held_next.assign(n, NIL);
        note_pos.assign(n, 0);
        released.reserve(n);
        // a note's heap holds what is held on that note; reserve four times the even share of the 128 MIDI notes (address
        // space only until touched; a bank that piles more than that on one note grows that heap, once)
        const size_t share = std::min<size_t>(n, std::max<size_t>(1024, (size_t)n / 32));
        for (size_t k = 0; k < 128; ++k) by_note[k].reserve(share);
    }
    // `midi_in` is an ArrayVec<EventInstance, 32> (graph/types.rs:18) in front of MAX_VOICES = 24 voices; the capacity
    // is lifted with N in the same proportion (32 per 24 voices); og_midi_set_queue_capacity overrides it
    uint32_t queue_cap = 32;
    struct Msg {
        uint8_t bytes[3];
        uint32_t len, frame;
        uint64_t seq;
    };
    std::vector<Msg> queue;
    bool queue_sorted = true; // messages arrived in frame order so far (the usual case): flush() skips the sort
    uint64_t seq = 0;
    uint64_t dropped = 0;
    struct Out { // what reached the voices (log for detached use / tests)
        uint32_t voice, frame;
        float frequency, gate;
        int has_frequency;
    };
    std::deque<Out> log;

    // allocate_voice  voice_allocator.rs:57-89
    uint32_t allocate(uint8_t note)
    {
        uint32_t i;
        if (n_fresh < n) {
            i = n_fresh++;
        } else if (!released.empty()) {
            i = released_pop(); // released voices first, the oldest of them
        } else {
            i = held_head; // all held: the oldest (never empty here)
            held_unlink(i);
            note_erase((uint8_t)voices[i].note, i);
        }
        Voice& v = voices[i];
        v.active = true;
        v.released = false;
        v.note = note;
        v.age = current_age++;
        held_append(i);
        note_insert(note, i);
        return i;
    }
    int find(uint8_t note) // find_voice_for_note :92-98
    {
        const auto& h = by_note[note];
        return h.empty() ? -1 : (int)h.front();
    }
    void release(uint32_t i) // release_voice :101-108 (only ever called on a held voice: find() returns nothing else)
    {
        held_unlink(i);
        note_erase((uint8_t)voices[i].note, i);
        voices[i].released = true;
        voices[i].note = -1;
        released_push(AgeEntry{voices[i].age, i});
    }
    static float note_to_freq(uint8_t note) // midi.rs:69-72
    {
        const float semitone_offset = (float)note - 69.0f;
        return 440.0f * powf(2.0f, semitone_offset / 12.0f);
    }
    int last_rc = OG_OK; // first engine error of the current flush (e.g. OG_E_OVERFLOW: 33rd gate of a voice in one block)
    void emit(uint32_t voice, uint32_t frame, bool has_f, float f, float gate)
    {
        if (engine) {
            int rc = OG_OK;
            if (has_f) rc = og_push_voice_value(engine, (uint32_t)freq_input, voice, frame, f);
            const int rc2 = og_push_voice_event(engine, (uint32_t)gate_input, voice, frame, gate);
            if (rc == OG_OK) rc = rc2;
            if (rc != OG_OK && last_rc == OG_OK) last_rc = rc;
        } else if (cluster) {
            int rc = OG_OK;
            if (has_f) rc = og_cluster_push_voice_value(cluster, (uint32_t)freq_input, voice, frame, f);
            const int rc2 = og_cluster_push_voice_event(cluster, (uint32_t)gate_input, voice, frame, gate);
            if (rc == OG_OK) rc = rc2;
            if (rc != OG_OK && last_rc == OG_OK) last_rc = rc;
        } else {
            log.push_back(Out{voice, frame, f, gate, has_f ? 1 : 0});
            if (log.size() > 65536) log.pop_front();
        }
    }
    void apply(const Msg& m)
    {
        if (m.len < 3) return; // parse_bytes :147-171
        const uint8_t status = m.bytes[0] & 0xF0, note = m.bytes[1], vel = m.bytes[2];
        const bool on = status == 0x90 && vel != 0;
        const bool off = status == 0x80 || (status == 0x90 && vel == 0);
        if (on) {
            float v = (float)vel / 127.0f;
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            const uint32_t i = allocate(note);            // on_note_on :112-122
            voices[i].handler_note = note;                // MidiVoiceHandler::on_note_on midi.rs:91-105
            emit(i, m.frame, true, note_to_freq(note), v);
        } else if (off) {
            const int i = find(note);                     // on_note_off :124-136
            if (i >= 0) {
                if (voices[i].handler_note == (int)note) { // midi.rs:107-121
                    emit((uint32_t)i, m.frame, false, 0.0f, 0.0f);
                    voices[i].handler_note = -1;
                }
                release((uint32_t)i);
            }
        }
    }
    // frames: length of the block the messages are for.  A `midi_in` event whose frame_offset >= frames never
    // reaches the parser (the generated loop only visits frames < frames and the queue is cleared with the
    // block, codegen/mod.rs:782-871), so it must not touch the allocator either.
    // OSCEN_GPU_HOST_PROF=1: time spent parsing / allocating / pushing (printed by og_midi_destroy)
    bool prof_on = ogabi::experiment_knob("OSCEN_GPU_HOST_PROF") != nullptr;
    double prof_t = 0.0;
    uint64_t prof_msgs = 0, prof_calls = 0;
    void flush(uint32_t frames = 0xFFFFFFFFu)
    {
        std::chrono::steady_clock::time_point t0;
        if (prof_on) {
            t0 = std::chrono::steady_clock::now();
            prof_msgs += queue.size();
            prof_calls += 1;
        }
        flush_impl(frames);
        if (prof_on) prof_t += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void flush_impl(uint32_t frames)
    {
        if (!queue_sorted) std::stable_sort(queue.begin(), queue.end(), [](const Msg& a, const Msg& b) { return a.frame < b.frame; });
        queue_sorted = true;
        for (const Msg& m : queue) {
            if (m.frame >= frames) {
                dropped += 1;
                continue;
            }
            apply(m);
        }
        queue.clear();
    }
};

extern "C" {

int og_midi_create(og_engine* e, uint32_t n_voices, const char* frequency_input, const char* gate_input, og_midi** out)
{
    return ogabi::guard([&]() -> int {
    if (!out) return OG_E_INVALID;
    og_midi* m = new og_midi;
    m->engine = e;
    if (e) {
        m->n = og_num_voices(e);
        m->freq_input = og_input_index(e, frequency_input ? frequency_input : "frequency");
        m->gate_input = og_input_index(e, gate_input ? gate_input : "gate");
        if (m->freq_input < 0 || m->gate_input < 0) {
            delete m;
            return OG_E_INVALID;
        }
    } else {
        m->n = n_voices;
    }
    if (m->n == 0) {
        delete m;
        return OG_E_INVALID;
    }
    m->size_indexes();
    m->queue_cap = 32u * ((m->n + 23u) / 24u);
    *out = m;
    return OG_OK;
    });
}

// The same front end over a multi-GPU bank: ONE allocator over the cluster's global voice ids (the LRU / stealing
// decisions are those of a single bank of that size), every voice message routed to the shard that owns the voice.
int og_midi_create_cluster(og_cluster* c, const char* frequency_input, const char* gate_input, og_midi** out)
{
    return ogabi::guard([&]() -> int {
    if (!out || !c) return OG_E_INVALID;
    if (og_cluster_num_voices(c) > 0xFFFFFFFFull) return OG_E_UNSUPPORTED; // (voice ids of the allocator are 32-bit)
    og_midi* m = new og_midi;
    m->cluster = c;
    m->n = (uint32_t)og_cluster_num_voices(c);
    m->freq_input = og_cluster_input_index(c, frequency_input ? frequency_input : "frequency");
    m->gate_input = og_cluster_input_index(c, gate_input ? gate_input : "gate");
    if (m->freq_input < 0 || m->gate_input < 0 || m->n == 0) {
        delete m;
        return OG_E_INVALID;
    }
    m->size_indexes();
    m->queue_cap = 32u * ((m->n + 23u) / 24u);
    *out = m;
    return OG_OK;
    });
}

void og_midi_destroy(og_midi* m)
{
    if (m && m->prof_on && m->prof_calls)
        fprintf(stderr, "[oscen_gpu host prof] midi flush (parse+allocate+push) %llu calls %llu msgs %.1f us total %.2f us/call %.1f ns/msg\n",
                (unsigned long long)m->prof_calls, (unsigned long long)m->prof_msgs, m->prof_t * 1e6, m->prof_t * 1e6 / (double)m->prof_calls,
                m->prof_msgs ? m->prof_t * 1e9 / (double)m->prof_msgs : 0.0);
    delete m;
}

float og_midi_note_to_freq(uint8_t note) { return og_midi::note_to_freq(note); }

int og_midi_send(og_midi* m, const uint8_t* bytes, uint32_t len, uint32_t frame_offset)
{
    return ogabi::guard([&]() -> int {
    if (!m || !bytes) return OG_E_INVALID;
    og_midi::Msg msg;
    memset(&msg, 0, sizeof msg);
    msg.len = len < 3 ? len : 3; // RawMidiMessage::new midi.rs:15-22
    memcpy(msg.bytes, bytes, msg.len);
    msg.frame = frame_offset;
    msg.seq = m->seq++;
    if (m->queue.size() >= m->queue_cap) { // try_push on a full ArrayVec: Err, the event is dropped
        m->dropped += 1;
        return OG_E_OVERFLOW;
    }
    if (!m->queue.empty() && m->queue.back().frame > msg.frame) m->queue_sorted = false;
    m->queue.push_back(msg);
    return OG_OK;
    });
}

int og_midi_send_batch(og_midi* m, const uint8_t* bytes3, const uint32_t* frame_offsets, uint32_t n)
{
    return ogabi::guard([&]() -> int {
    if (!m || (n && (!bytes3 || !frame_offsets))) return OG_E_INVALID;
    int rc = OG_OK;
    for (uint32_t i = 0; i < n; ++i) {
        const int r = og_midi_send(m, bytes3 + 3 * (size_t)i, 3, frame_offsets[i]);
        if (r != OG_OK && rc == OG_OK) rc = r;
    }
    return rc;
    });
}

int og_midi_set_queue_capacity(og_midi* m, uint32_t capacity)
{
    return ogabi::guard([&]() -> int {
    if (!m || capacity == 0) return OG_E_INVALID;
    m->queue_cap = capacity;
    return OG_OK;
    });
}

uint64_t og_midi_dropped(const og_midi* m) { return m ? m->dropped : 0; }

int og_midi_flush(og_midi* m)
{
    return ogabi::guard([&]() -> int {
    if (!m) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush();
    return m->last_rc;
    });
}

int og_midi_process_block(og_midi* m, uint32_t frames, float* out_bus)
{
    return ogabi::guard([&]() -> int {
    if (!m || (!m->engine && !m->cluster)) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush(frames);
    const int rc = m->engine ? og_process_block(m->engine, frames, out_bus) : og_cluster_process_block(m->cluster, frames, out_bus);
    return rc != OG_OK ? rc : m->last_rc; // the block was rendered; a non-zero code reports a dropped event
    });
}

int og_midi_process_block_async(og_midi* m, uint32_t frames, float* d_out_bus)
{
    return ogabi::guard([&]() -> int {
    if (m && m->cluster) return OG_E_UNSUPPORTED; // (a cluster's bus is complete only after the cross-device reduce)
    if (!m || !m->engine) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush(frames);
    const int rc = og_process_block_async(m->engine, frames, d_out_bus);
    return rc != OG_OK ? rc : m->last_rc;
    });
}

int og_midi_voice_state(const og_midi* m, uint32_t voice, int* active, int* released, int* note, uint32_t* age)
{
    return ogabi::guard([&]() -> int {
    if (!m || voice >= m->n) return OG_E_INVALID;
    const auto& v = m->voices[voice];
    if (active) *active = v.active;
    if (released) *released = v.released;
    if (note) *note = v.note;
    if (age) *age = v.age;
    return OG_OK;
    });
}

int og_midi_pop_output(og_midi* m, uint32_t* voice, uint32_t* frame, float* frequency, int* has_frequency, float* gate)
{
    return ogabi::guard([&]() -> int {
    if (!m) return OG_E_INVALID;
    if (m->log.empty()) return 0;
    const auto o = m->log.front();
    m->log.pop_front();
    if (voice) *voice = o.voice;
    if (frame) *frame = o.frame;
    if (frequency) *frequency = o.frequency;
    if (has_frequency) *has_frequency = o.has_frequency;
    if (gate) *gate = o.gate;
    return 1;
    });
}

} // extern "C"
