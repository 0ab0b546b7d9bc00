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
    // three indexes (binary heaps with lazy deletion: ~tens of ns per note, no allocation in steady state):
    //  * voices become active in index order and never turn inactive again (release keeps `active`,
    //    voice_allocator.rs:101-108), so "first inactive voice" is a counter;
    //  * stealing = min over (released ? 0 : 1, age): ages are unique, so one age-ordered heap per release state;
    //  * find_voice_for_note = lowest index among the held voices playing that note: a min-heap of indices per note.
    // An entry is stale when the voice has since changed state (its age / note / release flag no longer match).
    uint32_t n_fresh = 0; // voices [0, n_fresh) are active
    struct AgeEntry {
        uint32_t age, voice;
        bool operator>(const AgeEntry& o) const { return age > o.age; }
    };
    template <class T>
    struct MinHeap {
        std::vector<T> v;
        void push(const T& x)
        {
            v.push_back(x);
            std::push_heap(v.begin(), v.end(), std::greater<T>());
        }
        void pop()
        {
            std::pop_heap(v.begin(), v.end(), std::greater<T>());
            v.pop_back();
        }
        const T& top() const { return v.front(); }
        bool empty() const { return v.empty(); }
    };
    MinHeap<AgeEntry> released_by_age, held_by_age;
    MinHeap<uint32_t> held_by_note[256];
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

    bool held_ok(const AgeEntry& e) const { return voices[e.voice].active && !voices[e.voice].released && voices[e.voice].age == e.age; }
    bool released_ok(const AgeEntry& e) const { return voices[e.voice].active && voices[e.voice].released && voices[e.voice].age == e.age; }
    // allocate_voice  voice_allocator.rs:57-89
    uint32_t allocate(uint8_t note)
    {
        uint32_t i;
        if (n_fresh < n) {
            i = n_fresh++;
        } else {
            while (!released_by_age.empty() && !released_ok(released_by_age.top())) released_by_age.pop();
            if (!released_by_age.empty()) {
                i = released_by_age.top().voice; // released voices first, the oldest of them
                released_by_age.pop();
            } else {
                while (!held_ok(held_by_age.top())) held_by_age.pop(); // all held: the oldest (never empty here)
                i = held_by_age.top().voice;
                held_by_age.pop();
            }
        }
        Voice& v = voices[i];
        v.active = true;
        v.released = false;
        v.note = note;
        v.age = current_age++;
        held_by_age.push(AgeEntry{v.age, i});
        held_by_note[note].push(i);
        // stale entries never outnumber the live ones by much: rebuild a heap that has grown past 4 N
        if (held_by_age.v.size() > 4u * (size_t)n + 64u) rebuild();
        return i;
    }
    void rebuild()
    {
        held_by_age.v.clear();
        released_by_age.v.clear();
        for (auto& h : held_by_note) h.v.clear();
        for (uint32_t i = 0; i < n_fresh; ++i) {
            const Voice& v = voices[i];
            if (v.released) {
                released_by_age.v.push_back(AgeEntry{v.age, i});
            } else {
                held_by_age.v.push_back(AgeEntry{v.age, i});
                if (v.note >= 0) held_by_note[v.note & 255].v.push_back(i);
            }
        }
        std::make_heap(held_by_age.v.begin(), held_by_age.v.end(), std::greater<AgeEntry>());
        std::make_heap(released_by_age.v.begin(), released_by_age.v.end(), std::greater<AgeEntry>());
        for (auto& h : held_by_note) std::make_heap(h.v.begin(), h.v.end(), std::greater<uint32_t>());
    }
    int find(uint8_t note) // find_voice_for_note :92-98
    {
        auto& h = held_by_note[note];
        while (!h.empty()) {
            const Voice& v = voices[h.top()];
            if (v.active && !v.released && v.note == (int)note) return (int)h.top();
            h.pop();
        }
        return -1;
    }
    void release(uint32_t i) // release_voice :101-108
    {
        voices[i].released = true;
        voices[i].note = -1;
        released_by_age.push(AgeEntry{voices[i].age, i});
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
    bool prof_on = getenv("OSCEN_GPU_HOST_PROF") != nullptr;
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
    m->voices.resize(m->n);
    m->queue_cap = 32u * ((m->n + 23u) / 24u);
    *out = m;
    return OG_OK;
}

// The same front end over a multi-GPU bank: ONE allocator over the cluster's global voice ids (the LRU / stealing
// decisions are those of a single bank of that size), every voice message routed to the shard that owns the voice.
int og_midi_create_cluster(og_cluster* c, const char* frequency_input, const char* gate_input, og_midi** out)
{
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
    m->voices.resize(m->n);
    m->queue_cap = 32u * ((m->n + 23u) / 24u);
    *out = m;
    return OG_OK;
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
}

int og_midi_send_batch(og_midi* m, const uint8_t* bytes3, const uint32_t* frame_offsets, uint32_t n)
{
    if (!m || (n && (!bytes3 || !frame_offsets))) return OG_E_INVALID;
    int rc = OG_OK;
    for (uint32_t i = 0; i < n; ++i) {
        const int r = og_midi_send(m, bytes3 + 3 * (size_t)i, 3, frame_offsets[i]);
        if (r != OG_OK && rc == OG_OK) rc = r;
    }
    return rc;
}

int og_midi_set_queue_capacity(og_midi* m, uint32_t capacity)
{
    if (!m || capacity == 0) return OG_E_INVALID;
    m->queue_cap = capacity;
    return OG_OK;
}

uint64_t og_midi_dropped(const og_midi* m) { return m ? m->dropped : 0; }

int og_midi_flush(og_midi* m)
{
    if (!m) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush();
    return m->last_rc;
}

int og_midi_process_block(og_midi* m, uint32_t frames, float* out_bus)
{
    if (!m || (!m->engine && !m->cluster)) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush(frames);
    const int rc = m->engine ? og_process_block(m->engine, frames, out_bus) : og_cluster_process_block(m->cluster, frames, out_bus);
    return rc != OG_OK ? rc : m->last_rc; // the block was rendered; a non-zero code reports a dropped event
}

int og_midi_process_block_async(og_midi* m, uint32_t frames, float* d_out_bus)
{
    if (m && m->cluster) return OG_E_UNSUPPORTED; // (a cluster's bus is complete only after the cross-device reduce)
    if (!m || !m->engine) return OG_E_INVALID;
    m->last_rc = OG_OK;
    m->flush(frames);
    const int rc = og_process_block_async(m->engine, frames, d_out_bus);
    return rc != OG_OK ? rc : m->last_rc;
}

int og_midi_voice_state(const og_midi* m, uint32_t voice, int* active, int* released, int* note, uint32_t* age)
{
    if (!m || voice >= m->n) return OG_E_INVALID;
    const auto& v = m->voices[voice];
    if (active) *active = v.active;
    if (released) *released = v.released;
    if (note) *note = v.note;
    if (age) *age = v.age;
    return OG_OK;
}

int og_midi_pop_output(og_midi* m, uint32_t* voice, uint32_t* frame, float* frequency, int* has_frequency, float* gate)
{
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
}

} // extern "C"
