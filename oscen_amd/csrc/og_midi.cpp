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
#include <cmath>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/oscen_gpu.h"

struct og_midi {
    og_engine* engine = nullptr;
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
    struct Msg {
        uint8_t bytes[3];
        uint32_t len, frame;
        uint64_t seq;
    };
    std::vector<Msg> queue;
    uint64_t seq = 0;
    struct Out { // what reached the voices (log for detached use / tests)
        uint32_t voice, frame;
        float frequency, gate;
        int has_frequency;
    };
    std::deque<Out> log;

    // allocate_voice  voice_allocator.rs:57-89
    uint32_t allocate(uint8_t note)
    {
        for (uint32_t i = 0; i < n; ++i)
            if (!voices[i].active) return take(i, note);
        uint32_t best = 0;
        for (uint32_t i = 1; i < n; ++i) { // min_by_key((released ? 0 : 1, age)), first minimum wins
            const int pa = voices[i].released ? 0 : 1, pb = voices[best].released ? 0 : 1;
            if (pa < pb || (pa == pb && voices[i].age < voices[best].age)) best = i;
        }
        return take(best, note);
    }
    uint32_t take(uint32_t i, uint8_t note)
    {
        voices[i].active = true;
        voices[i].released = false;
        voices[i].note = note;
        voices[i].age = current_age++;
        return i;
    }
    int find(uint8_t note) const // find_voice_for_note :92-98
    {
        for (uint32_t i = 0; i < n; ++i)
            if (voices[i].active && !voices[i].released && voices[i].note == (int)note) return (int)i;
        return -1;
    }
    static float note_to_freq(uint8_t note) // midi.rs:69-72
    {
        const float semitone_offset = (float)note - 69.0f;
        return 440.0f * powf(2.0f, semitone_offset / 12.0f);
    }
    void emit(uint32_t voice, uint32_t frame, bool has_f, float f, float gate)
    {
        if (engine) {
            if (has_f) og_push_voice_value(engine, (uint32_t)freq_input, voice, frame, f);
            og_push_voice_event(engine, (uint32_t)gate_input, voice, frame, gate);
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
                voices[i].released = true; // release_voice :101-108
                voices[i].note = -1;
            }
        }
    }
    void flush()
    {
        std::stable_sort(queue.begin(), queue.end(), [](const Msg& a, const Msg& b) { return a.frame < b.frame; });
        for (const Msg& m : queue) apply(m);
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
    *out = m;
    return OG_OK;
}

void og_midi_destroy(og_midi* m) { delete m; }

int og_midi_send(og_midi* m, const uint8_t* bytes, uint32_t len, uint32_t frame_offset)
{
    if (!m || !bytes) return OG_E_INVALID;
    og_midi::Msg msg;
    memset(&msg, 0, sizeof msg);
    msg.len = len < 3 ? len : 3; // RawMidiMessage::new midi.rs:15-22
    memcpy(msg.bytes, bytes, msg.len);
    msg.frame = frame_offset;
    msg.seq = m->seq++;
    m->queue.push_back(msg);
    return OG_OK;
}

int og_midi_flush(og_midi* m)
{
    if (!m) return OG_E_INVALID;
    m->flush();
    return OG_OK;
}

int og_midi_process_block(og_midi* m, uint32_t frames, float* out_bus)
{
    if (!m || !m->engine) return OG_E_INVALID;
    m->flush();
    return og_process_block(m->engine, frames, out_bus);
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
