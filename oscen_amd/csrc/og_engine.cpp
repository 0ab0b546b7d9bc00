// og_engine.cpp -- host runtime of the voice-bank engine + the C ABI of
// include/oscen_gpu.h.  Compiled by hipcc (host side only; the kernels live in
// csrc/gen/*.hip and in og_kernel_rt.hip.h).
//
// An engine is the MI355X form of the reference's poly wrapper graph
// (examples/fm-synth/src/lib.rs:22-131): `voices = [Voice; N]`, every broadcast
// value input fanned out to all voices (ramped where declared `[ramp: N]`),
// per-voice `frequency`/`gate`, and `voices.out -> out` summed onto the bus.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/oscen_gpu.h"
#include "og_abi.h"
#include "og_graph.h"
#include "og_jit.h"
#include "og_math.h"
#include "og_registry.h"

// Sum the per-workgroup partial rows (fixed association, no atomics).
// One workgroup per 16 frames, 64 row-slices x 16 frames = 1024 threads: thread (slice s, frame f) adds
// rows s, s+64, s+128, ... of its group of <= 1024 rows with eight independent accumulators -- its
// <= 16 loads are all in flight together, so the pass costs about two memory latencies (the rows
// were written by other XCDs, they come from HBM/MALL) -- then the 64 slice sums are folded 4 -> 1
// and 16 -> 1 through LDS.  History: 4 waves walking all rows 40 us; 16 slices x 64 frames 6.5 us.
#define OG_RED_SLICES 64
#define OG_RED_FRAMES 16
static_assert(OG_RED_FRAMES == OG_BUS_CHUNK, "og_bus_reduce reads the chunks og::bus_chunk_reduce writes");
#define OG_RED_GROUP 1024 // rows per workgroup; larger banks take a second pass over the group sums
// out_stride / out_off: the last pass of a stereo bus writes channel out_off of interleaved Frame<2> samples
__device__ __forceinline__ void og_bus_reduce_body(const float* __restrict__ partials, uint32_t n_rows, uint32_t frames,
                                                   float* __restrict__ out, uint32_t out_stride, uint32_t out_off)
{
    __shared__ float part[OG_RED_SLICES][OG_RED_FRAMES];
    __shared__ float quad[OG_RED_SLICES / 4][OG_RED_FRAMES];
    const uint32_t fx = threadIdx.x % OG_RED_FRAMES;
    const uint32_t slice = threadIdx.x / OG_RED_FRAMES;
    const uint32_t f = blockIdx.x * OG_RED_FRAMES + fx;
    const uint32_t row0 = blockIdx.y * OG_RED_GROUP;
    const uint32_t row1 = min(n_rows, row0 + OG_RED_GROUP);
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // rows are chunk-major, [chunk][row][OG_RED_FRAMES] (og::bus_chunk_reduce): this workgroup's chunk is contiguous
    const float* __restrict__ chunk = partials + (size_t)blockIdx.x * n_rows * OG_RED_FRAMES;
    if (f < frames) {
        uint32_t r = row0 + slice;
        for (; r + 7 * OG_RED_SLICES < row1; r += 8 * OG_RED_SLICES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += chunk[(size_t)(r + i * OG_RED_SLICES) * OG_RED_FRAMES + fx];
        }
        for (int i = 0; r < row1; r += OG_RED_SLICES, ++i) acc[i] += chunk[(size_t)r * OG_RED_FRAMES + fx];
    }
    part[slice][fx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (slice < OG_RED_SLICES / 4)
        quad[slice][fx] = (part[4 * slice][fx] + part[4 * slice + 1][fx]) + (part[4 * slice + 2][fx] + part[4 * slice + 3][fx]);
    __syncthreads();
    if (slice == 0 && f < frames) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < OG_RED_SLICES / 4; ++i) s[i & 3] += quad[i][fx];
        // group sums in the same layout, [chunk][group][OG_RED_FRAMES]; with one group that is the bus itself
        out[(((size_t)blockIdx.x * gridDim.y + blockIdx.y) * OG_RED_FRAMES + fx) * out_stride + out_off] = (s[0] + s[1]) + (s[2] + s[3]);
    }
}

// og_shader_clock_ghz: one wave spins for `ticks` of the constant 100 MHz counter (s_memrealtime) and counts shader cycles
// (s_memtime: one tick per shader clock, MI355X_MICROARCH.md) over the same span
__global__ __launch_bounds__(64) void og_clock_probe(unsigned long long* out, unsigned ticks)
{
#ifndef OG_HOSTSIM
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r1;
    do {
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < ticks);
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
#else
    if (threadIdx.x == 0) out[0] = out[1] = 0ull; // (the host simulator has no shader clock)
    (void)ticks;
#endif
}

__global__ __launch_bounds__(1024) void og_bus_reduce(const float* __restrict__ partials, uint32_t n_rows,
                                                      uint32_t frames, float* __restrict__ out, uint32_t out_stride, uint32_t out_off)
{
    og_bus_reduce_body(partials, n_rows, frames, out, out_stride, out_off);
}

// Post-mix Tremolo (examples/electric-piano/src/tremolo.rs:40-62) on the summed bus -> Frame<2>.
// The LFO phase recurrence `phase = fract(phase + rate/sr)` is serial but two ops per frame: lane 0
// runs it into LDS, then every frame computes its sine (bit-exact libm restatement) in parallel.
__global__ __launch_bounds__(512) void og_bus_tremolo(const float* __restrict__ mono, uint32_t frames, float rate,
                                                      float depth, float sr, float* __restrict__ phase_state,
                                                      float* __restrict__ out)
{
    __shared__ float ph[OG_MAX_BLOCK];
    if (threadIdx.x == 0) {
        float p = *phase_state;
        const float phase_increment = rate / sr;
        for (uint32_t f = 0; f < frames; ++f) {
            ph[f] = p;
            const float q = p + phase_increment;
            p = q - truncf(q);
        }
        *phase_state = p;
    }
    __syncthreads();
    for (uint32_t f = threadIdx.x; f < frames; f += blockDim.x) {
        const float input = mono[f];
        const float lfo = og_sinf_exact(ph[f] * 2.0f * 3.14159274101257324f);
        const float scaled_depth = depth / 3.0f;
        const float pan = 0.5f + lfo * scaled_depth;
        out[2 * f] = input * pan;
        out[2 * f + 1] = input * (1.0f - pan);
    }
}

// Incremental event path: append the staged segments to the timeline and point the listed voices at them.
// `staged` and `upd` are PINNED HOST buffers read by the kernel itself: a kernel launch never waits for the stream,
// whereas hipMemcpyAsync of a small pinned buffer was measured to block until the stream had drained (~290 us with a
// batch of blocks in flight), which serialised the host's event preparation with the GPU.
// upd = n x {voice, cursor, end}
constexpr size_t EV_UPD_WORDS = 3;
__global__ void og_apply_event_updates(const uint4* __restrict__ staged, uint32_t n_ev, uint4* __restrict__ timeline_tail,
                                       const uint32_t* __restrict__ upd, uint32_t n, uint32_t* __restrict__ cursor,
                                       uint32_t* __restrict__ end)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ev) timeline_tail[i] = staged[i]; // (OgEvent = 16 bytes)
    if (i < n) {
        const uint32_t v = upd[EV_UPD_WORDS * i];
        cursor[v] = upd[EV_UPD_WORDS * i + 1];
        end[v] = upd[EV_UPD_WORDS * i + 2];
    }
}

// Stream progress marker: launched after every batch of blocks, writes the batch number into a word of pinned host
// memory.  The host checks it before it reuses a staging buffer (event segments, ramp tables): hipEventSynchronize on an
// event recorded many launches earlier was measured to block until the whole stream had drained (~350 us per call with
// a batch in flight), which serialised the host's preparation of the next batch with the GPU.
__global__ void og_stream_mark(volatile uint64_t* host_word, uint64_t seq)
{
    __threadfence_system();
    __hip_atomic_store(const_cast<uint64_t*>(host_word), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- registry ---------------------------------------------------------------
OgKernelEntry*& og_kernel_registry_head()
{
    static OgKernelEntry* head = nullptr;
    return head;
}
OgLaunchFn og_find_kernel(uint64_t hash)
{
    for (OgKernelEntry* e = og_kernel_registry_head(); e; e = e->next)
        if (e->hash == hash) return e->launch;
    return nullptr;
}
OgOccupancyFn og_find_occupancy(uint64_t hash)
{
    for (OgKernelEntry* e = og_kernel_registry_head(); e; e = e->next)
        if (e->hash == hash) return e->occupancy;
    return nullptr;
}

namespace {
thread_local std::string g_err;
}
namespace ogabi {
int set_error(int code, const std::string& m)
{
    try {
        g_err = m;
    } catch (...) { // (out of memory while recording the message: keep the code)
        g_err.clear();
    }
    return code;
}
// the form guard()'s catch handlers use: the std::string is built INSIDE the try (guard is noexcept: a bad_alloc while
// copying e.what() must not become std::terminate)
int set_error(int code, const char* m) noexcept
{
    try {
        g_err.assign(m ? m : "");
    } catch (...) {
        g_err.clear();
    }
    return code;
}
} // namespace ogabi

namespace {

int set_err(int code, const std::string& m) { return ogabi::set_error(code, m); }

using HipError = ogabi::DeviceError;
#define HIPCK(expr)                                                                                    \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));                         \
    } while (0)

// Host memory this library does not own (a caller's array, a std::vector) never goes to hipMemcpyAsync directly.  For
// a transfer above a size threshold the runtime pins the pages where they lie (a userptr mapping) and keeps the
// pinning cached; whenever the kernel later migrates, compacts or unmaps those pages the driver evicts and restores
// EVERY queue of the process.  Measured through the blocking entry at 4 M / 8 M voices (bench.py's real-time record:
// the 32 MB frequency array of og_set_voice_values and the event-timeline vectors of the first rebuild were such
// mappings): one ~23 ms stall of the stream -- several missed audio deadlines -- every few thousand blocks.  So every
// transfer larger than a staging copy goes through two pinned bounce buffers of this engine's own.
struct Bounce {
    static constexpr size_t CHUNK = (size_t)4 << 20, DIRECT = 16384; // (below DIRECT the runtime stages the bytes itself)
    void* h[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int k = 0;
    void ensure()
    {
        if (h[0]) return;
        for (int i = 0; i < 2; ++i) {
            HIPCK(hipHostMalloc(&h[i], CHUNK, hipHostMallocDefault));
            HIPCK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
    // host -> device, asynchronous like hipMemcpyAsync from pinned memory: `src` may be reused when the call returns
    void h2d(void* dst, const void* src, size_t n, hipStream_t s)
    {
        if (n <= DIRECT) {
            if (n) HIPCK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
            return;
        }
        ensure();
        for (size_t off = 0; off < n; off += CHUNK) {
            const size_t len = std::min(CHUNK, n - off);
            if (busy[k]) HIPCK(hipEventSynchronize(ev[k]));
            memcpy(h[k], (const char*)src + off, len);
            HIPCK(hipMemcpyAsync((char*)dst + off, h[k], len, hipMemcpyHostToDevice, s));
            HIPCK(hipEventRecord(ev[k], s));
            busy[k] = true;
            k ^= 1;
        }
    }
    // device -> host; the bytes are in `dst` when the call returns (the stream is drained up to the copy)
    void d2h(void* dst, const void* src, size_t n, hipStream_t s)
    {
        if (n <= DIRECT) {
            if (n) HIPCK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            return;
        }
        ensure();
        size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
        auto land = [&](int i) {
            if (!pend_len[i]) return;
            HIPCK(hipEventSynchronize(ev[i]));
            memcpy((char*)dst + pend_off[i], h[i], pend_len[i]);
            pend_len[i] = 0;
            busy[i] = false;
        };
        for (int i = 0; i < 2; ++i)
            if (busy[i]) { // (an upload still reading the buffer)
                HIPCK(hipEventSynchronize(ev[i]));
                busy[i] = false;
            }
        for (size_t off = 0; off < n; off += CHUNK) {
            const size_t len = std::min(CHUNK, n - off);
            land(k);
            HIPCK(hipMemcpyAsync(h[k], (const char*)src + off, len, hipMemcpyDeviceToHost, s));
            HIPCK(hipEventRecord(ev[k], s));
            pend_off[k] = off;
            pend_len[k] = len;
            k ^= 1;
        }
        land(k);
        land(k ^ 1);
    }
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (h[i]) (void)hipHostFree(h[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            h[i] = nullptr;
            ev[i] = nullptr;
        }
    }
};

struct Ramp { // ValueRampState  oscen-lib/src/graph/types.rs:300-373
    float current = 0, target = 0, increment = 0;
    uint32_t frames_remaining = 0;
    uint32_t default_frames = 0;
    bool ramping() const { return frames_remaining > 0; }
    void set_immediate(float v)
    {
        current = target = v;
        increment = 0.0f;
        frames_remaining = 0;
    }
    void set_with_ramp(float t, uint32_t frames)
    {
        if (frames == 0) {
            set_immediate(t);
        } else {
            target = t;
            increment = (t - current) / (float)frames;
            frames_remaining = frames;
        }
    }
    bool tick()
    {
        if (frames_remaining > 0) {
            frames_remaining -= 1;
            if (frames_remaining == 0) {
                current = target;
                increment = 0.0f;
                return true;
            }
            current += increment;
        }
        return false;
    }
};

struct HostEvent { // a push that has not reached the device timeline yet
    uint32_t voice;
    uint64_t frame;
    uint32_t target;
    float value;
    uint64_t seq;
    bool block_local; // pushed relative to the next block (reference try_push semantics)
};

constexpr int RAMP_RING = 8;
constexpr int EV_RING = 8;              // pinned staging buffers of the incremental event path
constexpr size_t EV_STAGE_EVENTS = 131072; // events (and voice updates) one staging buffer holds (8 x 2 MB + 8 x 1.5 MB pinned)

} // namespace

struct og_graph_desc {
    ogc::GraphDesc g;
};

// OSCEN_GPU_HOST_PROF=1: wall time of the host-side phases of the live path, printed when the engine is destroyed
struct HostProf {
    enum { SYNC_EVENTS, INCREMENTAL, REBUILD, LAUNCH, RAMPS, EV_WAIT, EV_COMMIT, N };
    double t[N] = {};
    uint64_t n[N] = {};
    bool on = ogabi::experiment_knob("OSCEN_GPU_HOST_PROF") != nullptr;
    struct Scope {
        HostProf& p;
        int k;
        std::chrono::steady_clock::time_point t0;
        Scope(HostProf& p_, int k_) : p(p_), k(k_) { if (p.on) t0 = std::chrono::steady_clock::now(); }
        ~Scope()
        {
            if (!p.on) return;
            p.t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            p.n[k] += 1;
        }
    };
    void report() const
    {
        if (!on) return;
        static const char* names[N] = {"sync_events", "incremental_update", "full_rebuild", "flush_bus (launches)", "ramp table",
                                        "  staging-slot wait", "  commit + launch"};
        for (int k = 0; k < N; ++k)
            if (n[k]) fprintf(stderr, "[oscen_gpu host prof] %-22s %9llu calls %10.1f us total %8.2f us/call\n", names[k],
                              (unsigned long long)n[k], t[k] * 1e6, t[k] * 1e6 / (double)n[k]);
    }
};

struct og_engine {
    HostProf prof;
    std::unique_ptr<ogc::CompiledGraph> cg;
    OgLaunchFn launch = nullptr;
    std::unique_ptr<OgJitKernel> jit;
    uint32_t V = 0;
    int device = 0;
    float sr = 44100.0f;
    bool inited = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    std::vector<float> values; // per input: plain value, or mirror of ramp.current
    std::vector<std::vector<float>> stream_blocks; // per input: `<stream_in>_block` (stream inputs only), OG_MAX_BLOCK samples
    std::vector<Ramp> ramps;   // per input (only meaningful when ramp_row >= 0)
    uint32_t active_ramps = 0;

    uint32_t n_wg = 0;
    uint32_t lanes = OG_WAVE;
    uint32_t split = 0; // pipeline depth of the launched kernel variant: 0 (ordinary), 2 or 4 waves per 64 voices
    bool wide = false;  // split == 4: the 16-frame hand-off form (og_k4w_*)
    uint32_t* d_state = nullptr;
    Bounce bounce; // pinned staging for transfers from / to memory that is not ours
    uint32_t* d_lane_state = nullptr;
    float* d_ring[OG_MAX_RINGS] = {nullptr, nullptr, nullptr, nullptr}; // delay lines [capacity][V]
    uint32_t ring_cap[OG_MAX_RINGS] = {0, 0, 0, 0};
    float* d_mono = nullptr;      // summed voices before the post-mix stage
    float* d_bus_phase = nullptr; // Tremolo.phase
    OgEvent* d_events = nullptr;
    size_t ev_cap = 0;
    size_t ev_headroom_env = 0; // OSCEN_GPU_EV_HEADROOM at og_create (0 = unset)
    size_t ev_reserve = 0; // og_reserve_events: room kept behind a bulk score for live segments
    uint32_t* d_ev_end = nullptr;
    uint32_t* d_ev_cursor = nullptr;
    float* d_partials = nullptr;
    float* d_partials2 = nullptr; // group sums of the multi-pass bus reduce
    // Block queue (og_set_bus_batching): up to `bus_batch` consecutive async blocks that nothing separates (no value
    // change, no event push, no taps) are rendered by ONE launch of the voice kernel over their frames back to back --
    // state loaded and stored once, one inter-kernel gap, one bus reduce per tree level -- instead of one launch each.
    // A queued block has had its ramps ticked and its stream samples captured; anything that touches engine state
    // launches the queue first.  Results are those of block-by-block processing, bit for bit.
    std::vector<uint32_t> tap_voices; // og_set_voice_taps: the tapped voices by the caller's numbers (slots are resolved from them)
    uint32_t bus_batch = 1; // queue limit (blocks per launch)
    uint32_t batch_cap = 1; // what the buffers are sized for
    struct QueuedBlock {
        float* dst; // where the block's bus goes
        uint32_t frames;
        float trem_rate, trem_depth;
    };
    std::vector<QueuedBlock> queue;
    uint64_t q_frame0 = 0;   // absolute frame of the first queued block
    uint32_t q_frames = 0;   // frames queued
    bool q_ramps = false;    // some queued block ticked a ramp (or the graph has stream inputs): table-reading variant
    int q_ramp_slot = -1;    // staging buffer of the per-frame table being filled
    float* d_stage_bus = nullptr; // bus of a launch whose blocks' destinations are not contiguous
    float* d_bus = nullptr;
    float* d_ramp[RAMP_RING] = {};
    float* h_ramp[RAMP_RING] = {};
    uint64_t ramp_seq[RAMP_RING] = {}; // batch whose launch copied ramp table i to the device (0 = never)
    int ramp_head = 0;
    // events leaving the voices (graph event outputs) and pushes the in-voice queues dropped: device log / counter
    OgOutEvent* d_out_ev = nullptr;
    uint32_t* d_out_ev_count = nullptr; // [0] = events appended, [1] = in-voice pushes lost
    uint32_t out_ev_cap = 0;
    uint64_t out_ev_overflow = 0;       // events that did not fit the log (reported by og_read_output_events)
    uint64_t ev_lost_total = 0;         // in-voice pushes lost, read back so far
    std::vector<OgOutEvent> out_ev_carry; // og_read_output_events: drained from the device log, not yet handed out
    float* d_taps = nullptr;
    int32_t* d_tap_slot = nullptr;
    uint32_t n_taps = 0;
    uint32_t last_frames = 0;

    // ---- event timeline ---------------------------------------------------------------------------
    // Device: d_events holds every voice's unconsumed events as one segment [cursor, end), sorted by
    // (frame, push order).  A full rebuild lays the segments out in voice order (CSR) and uploads
    // O(V) words; the incremental path (live pushes: og_push_voice_event / MIDI) appends a new segment
    // for each voice that received events -- its unconsumed old events merged with the new ones -- at the
    // tail of d_events and repoints that voice's (cursor, end) with a tiny kernel: O(#pushes) host work,
    // one async copy from a pinned staging ring, no stream synchronisation.
    std::vector<HostEvent> pending;      // pushes not yet on the device timeline (PHYSICAL voice slots)
    // og_group_voices: logical voice (what every entry point takes and hands out) -> physical slot (what the device arrays
    // and everything below the entry points index).  Empty = identity.
    std::vector<uint32_t> phys_of, logical_of;
    uint32_t phys(uint32_t v) const { return phys_of.empty() ? v : phys_of[v]; }
    uint32_t logical(uint32_t p) const { return logical_of.empty() ? p : logical_of[p]; }
    std::vector<OgEvent> h_events;       // host mirror of d_events (the whole ring)
    std::vector<uint32_t> seg_begin, seg_end; // per voice: its current segment (empty vectors = all segments empty)
    std::vector<uint64_t> seg_last;           // per voice: frame of the segment's last event (< frame_now: all consumed)
    // Continuation (round 6): on the HOST a voice's timeline is its segment [seg_begin, seg_end) followed by
    // [cont_begin, cont_end) -- the tail of an EARLIER segment left where it lies in the ring.  A live push onto a voice with
    // a long score ahead of it writes a new segment {what is due up to the end of the launch being prepared, merged with the
    // pushes} and remembers the rest as the continuation; every frame in the segment is < every frame in the continuation.
    // The device knows nothing of this (one segment per voice, as ever -- a first form that taught the kernels to hop cost
    // the four-wave kernel 4 % and the ordinary one 30 spills): before the launch in which a continuation's first event is
    // due the host points the voice at it -- a 12-byte cursor update when the segment in front has been played, which is
    // the usual case (incremental_update / merge_voice).  Cost of a message: its own records, whatever the score's length.
    // Vectors are empty until a continuation exists; `cont_due` orders the voices by their continuation's first frame.
    std::vector<uint32_t> cont_begin, cont_end;
    std::vector<uint64_t> cont_last;
    uint64_t n_events_copied = 0; // events written to the ring by incremental updates (old ones carried over + new ones)
    std::vector<std::pair<uint64_t, uint32_t>> cont_due; // min-heap of {first frame of the continuation, voice}; stale entries are skipped
    void cont_due_push(uint32_t v)
    {
        cont_due.emplace_back(h_events[cont_begin[v]].frame, v);
        std::push_heap(cont_due.begin(), cont_due.end(), std::greater<std::pair<uint64_t, uint32_t>>());
    }
    // end of the launch that is being prepared: everything before it must be in the voices' segments
    uint64_t launch_end() const { return queue.empty() ? frame_now : q_frame0 + q_frames; }
    bool continuation_due() const { return !cont_due.empty() && cont_due.front().first < launch_end(); }
    // A rest shorter than this is carried over with the merge: copying a kilobyte costs less than the bookkeeping of a
    // continuation (a heap entry, a second cursor update, two more looks into the cold host mirror) -- measured on the loaded
    // real-time banks, whose 1 s scores leave ~14 events per voice: with continuations for those p99 rose from 2.29 to
    // 2.69 ms at 4 194 304 voices (gpurun r06q).
    static constexpr uint32_t CONT_MIN = 64;
    size_t n_conts = 0; // voices that have a continuation right now (0: nobody looks at the cont_* arrays -- cold memory)
    bool has_cont(uint32_t v) const { return n_conts != 0 && cont_begin[v] != cont_end[v]; }
    void set_cont(uint32_t v, uint32_t b, uint32_t e)
    {
        const bool had = cont_begin[v] != cont_end[v];
        cont_begin[v] = b;
        cont_end[v] = e;
        if (b != e) cont_last[v] = h_events[e - 1].frame;
        n_conts += (b != e ? 1 : 0);
        n_conts -= (had ? 1 : 0);
    }
    void ensure_cont() // (sized with the segment arrays, outside the real-time path: 16 bytes per voice of zero-fill)
    {
        if (cont_begin.empty()) {
            cont_begin.assign(V, 0);
            cont_end.assign(V, 0);
            cont_last.assign(V, 0);
        }
    }
    // events of one segment are sorted by frame: first index in [b, e) whose frame is >= fr / > fr
    uint32_t lower_frame(uint32_t b, uint32_t e, uint64_t fr) const
    {
        while (b < e) {
            const uint32_t m = b + (e - b) / 2;
            if (h_events[m].frame < fr) b = m + 1;
            else e = m;
        }
        return b;
    }
    uint32_t upper_frame(uint32_t b, uint32_t e, uint64_t fr) const
    {
        while (b < e) {
            const uint32_t m = b + (e - b) / 2;
            if (h_events[m].frame <= fr) b = m + 1;
            else e = m;
        }
        return b;
    }
    // the unconsumed part of the voice's segment / of its continuation, as index ranges of h_events
    void unconsumed_ranges(uint32_t v, uint64_t hz, uint32_t& hb, uint32_t& he, uint32_t& cb, uint32_t& ce) const
    {
        hb = he = cb = ce = 0;
        if (seg_begin.empty()) return;
        if (seg_begin[v] != seg_end[v] && seg_last[v] >= hz) {
            hb = lower_frame(seg_begin[v], seg_end[v], hz);
            he = seg_end[v];
        }
        if (has_cont(v) && cont_last[v] >= hz) {
            cb = lower_frame(cont_begin[v], cont_end[v], hz);
            ce = cont_end[v];
        }
    }
    std::vector<uint32_t> grp_head, grp_tail, grp_next, grp_voices; // incremental path: pending events chained per voice
    std::vector<uint8_t> local_cnt;      // [voice * n_event_inputs + event input]: try_push'ed events queued for the next block
    std::vector<uint32_t> local_touched; // entries of local_cnt to clear when the block starts
    size_t ev_tail = 0;                  // next free slot of d_events
    // d_events is a RING for the live path: segments are appended at ev_tail; a segment is dead once its voice has been
    // given a newer one or its last event lies before the consumed horizon, and the space of the dead segments at the
    // front is reused when the tail reaches the end of the buffer -- a steady stream of live pushes (MIDI playing)
    // never triggers the O(V) rebuild + stream synchronise that a bump pointer needs for compaction (measured: a
    // 7 ms stall every ~1 400 blocks at 1 M voices, a missed audio deadline).  Only a segment that stays alive at the
    // front for a whole lap (an event scheduled far ahead) still forces a rebuild.
    struct RingSeg {
        uint32_t voice, begin, end;
    };
    std::deque<RingSeg> ring_live; // live segments in append order (oldest first)
    uint64_t n_ring_wraps = 0;
    // where `n` events can be appended, or SIZE_MAX when the ring is full
    size_t ring_alloc(size_t n)
    {
        size_t at = ring_alloc_bounded(n, 4096);
        if (at == SIZE_MAX) at = ring_alloc_bounded(n, SIZE_MAX); // (no room behind a bounded sweep: finish it, then decide)
        return at;
    }
    // `sweep` bounds how many dead segments one call retires.  When a long resident score runs out, the segments of ALL
    // voices die within a few blocks: retiring millions of them in one call was a 10-15 ms stall on the real-time path
    // (round 4, the loaded bank: one block near the end of every run at 4-8 M voices); the room they occupy is not needed
    // at once, so the sweep is spread over the following calls.
    size_t ring_alloc_bounded(size_t n, size_t sweep)
    {
        const uint64_t hz = consumed_horizon();
        while (!ring_live.empty() && sweep-- > 0) {
            const RingSeg& f = ring_live.front();
            // alive: the voice's current segment, or the segment its continuation lies in, with something left to play
            // (a voice that was pointed at its continuation plays a sub-range of the older segment the continuation lay in)
            // (seg_begin first: a superseded segment -- the usual one at the front -- is settled by that one cold word, as ever)
            const uint32_t sb = seg_begin[f.voice];
            const bool is_head = sb >= f.begin && sb < f.end && seg_end[f.voice] <= f.end && sb != seg_end[f.voice] && seg_last[f.voice] >= hz;
            const bool holds_cont = has_cont(f.voice) && cont_begin[f.voice] >= f.begin && cont_end[f.voice] <= f.end && cont_last[f.voice] >= hz;
            const bool dead = !is_head && !holds_cont;
            if (!dead) break;
            ring_live.pop_front();
        }
        if (ring_live.empty()) {
            ev_tail = 0;
            return n <= ev_cap ? 0 : SIZE_MAX;
        }
        const size_t head = ring_live.front().begin;
        if (ev_tail >= head) { // live data is [head, tail): room behind the tail, or -- wrapping -- in front of the head
            if (ev_tail + n <= ev_cap) return ev_tail;
            if (n < head) {
                n_ring_wraps += 1;
                return 0;
            }
            return SIZE_MAX;
        }
        return ev_tail + n < head ? ev_tail : SIZE_MAX; // wrapped: the tail runs up to the head
    }
    bool ev_rebuild = false;             // next block must rebuild the whole timeline
    void sync_lost_counter(); // device "lost pushes" counter -> ev_lost_total (synchronises the stream)
    OgEvent* h_stage_ev[EV_RING] = {};   // pinned
    uint32_t* h_stage_upd[EV_RING] = {}; // pinned, n x {voice, cursor, end}
    uint32_t* d_stage_upd[EV_RING] = {};
    uint64_t stage_seq[EV_RING] = {};         // batch (flush_seq) whose launch read staging slot i (0 = never used)
    uint64_t flush_seq = 0;                    // batches launched so far
    bool batch_staged = false;                 // the batch being assembled reads a host staging buffer
    volatile uint64_t* h_progress = nullptr;  // pinned: number of the last batch the stream has finished (og_stream_mark)
    float* h_bus_pinned = nullptr; // pinned + device-visible: destination of a blocking block's bus (og_process_block)
    bool blocking_memcpy = false; // OSCEN_GPU_BLOCKING_MEMCPY (A/B knob), read once at og_create
    uint64_t blocking_waits = 0, blocking_timeouts = 0; // og_process_block calls / calls whose marker wait timed out
    bool wait_progress(uint64_t seq) // false: the stream was found finished before the marker was seen
    {
        // the batch is tens of microseconds long: spin on the marker word (a runtime wait costs more than the block).
        // A marker that does not show is rare (twice in 44 000 blocks of 4 M voices, both a 20-30 ms block under the
        // earlier "give it 20 ms, then hipStreamSynchronize" rule): from 256 us on the stream itself is asked every
        // 128 us (hipStreamQuery does not block), so a late marker costs a fraction of a block, not several deadlines.
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        auto next_query = t0 + std::chrono::microseconds(256);
        for (uint32_t spins = 0; !h_progress || *h_progress < seq; ++spins) {
            if ((spins & 255u) == 255u) {
                const auto now = clk::now();
                if (now >= next_query) {
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) return h_progress && *h_progress >= seq;
                    if (q != hipErrorNotReady) HIPCK(q);
                    next_query = now + std::chrono::microseconds(128);
                }
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        return true;
    }
    bool batch_done(uint64_t seq)
    {
        if (seq == 0 || (h_progress && *h_progress >= seq)) return true;
        if (seq > flush_seq) return true; // (staged for a batch that was never launched: nothing read it)
        HIPCK(hipStreamSynchronize(stream)); // (a ring slot that is still in flight: never seen in practice)
        return true;
    }
    int stage_head = 0;
    uint64_t n_full_rebuilds = 0, n_incremental = 0;
    size_t n_block_local = 0; // events pushed with try_push semantics for the next block
    size_t local_from = 0;    // index in `pending` of the first push since the previous block was queued
    uint64_t seq = 0;
    uint32_t bus_passes = 0; // og_bus_reduce launches of the last block (1 + levels of the multi-pass tree)
    uint64_t frame_now = 0;
    uint64_t dropped = 0;

    bool bus_stage = true; // run the post-mix node (Tremolo) here; a cluster shard hands over the mono sum instead

    bool timing = false;
    std::vector<hipEvent_t> t_start, t_stop;
    unsigned long long* h_clock = nullptr; // pinned, [T_CLOCK][4]: {cycles, ticks} at the start and at the end of timed launch i
    static constexpr size_t T_CLOCK = 8192;
    size_t t_used = 0;
    size_t t_blocks = 0; // blocks the timed launches covered
    double last_clock_ghz = 0.0;

    ~og_engine()
    {
        prof.report();
        (void)hipSetDevice(device);
        // a borrowed stream (og_set_stream) may already be gone: wait for the device instead of touching it
        if (own_stream && stream) (void)hipStreamSynchronize(stream);
        else (void)hipDeviceSynchronize();
        bounce.release();
        (void)hipFree(d_state);
        (void)hipFree(d_lane_state);
        for (int k = 0; k < OG_MAX_RINGS; ++k) (void)hipFree(d_ring[k]);
        (void)hipFree(d_mono);
        (void)hipFree(d_bus_phase);
        (void)hipFree(d_events);
        (void)hipFree(d_ev_end);
        (void)hipFree(d_ev_cursor);
        (void)hipFree(d_partials);
        (void)hipFree(d_partials2);
        (void)hipFree(d_stage_bus);
        (void)hipFree(d_bus);
        (void)hipFree(d_taps);
        (void)hipFree(d_tap_slot);
        (void)hipFree(d_out_ev);
        (void)hipFree(d_out_ev_count);
        for (int i = 0; i < RAMP_RING; ++i) {
            (void)hipFree(d_ramp[i]);
            if (h_ramp[i]) (void)hipHostFree(h_ramp[i]);
        }
        for (int i = 0; i < EV_RING; ++i) {
            if (h_stage_ev[i]) (void)hipHostFree(h_stage_ev[i]);
            if (h_stage_upd[i]) (void)hipHostFree(h_stage_upd[i]);
            (void)hipFree(d_stage_upd[i]);
        }
        if (h_progress) (void)hipHostFree((void*)h_progress);
        if (h_bus_pinned) (void)hipHostFree(h_bus_pinned);
        if (h_clock) (void)hipHostFree(h_clock);
        for (auto ev : t_start) (void)hipEventDestroy(ev);
        for (auto ev : t_stop) (void)hipEventDestroy(ev);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    ogc::UEnv env() const { return ogc::UEnv{sr, values.data()}; }

    void upload_initial_state()
    {
        const size_t nw = cg->state.size();
        std::vector<uint32_t> img(nw * (size_t)V);
        ogc::UEnv e = env();
        for (size_t w = 0; w < nw; ++w) {
            const uint32_t bits = cg->state[w].init(e);
            std::fill(img.begin() + w * V, img.begin() + (w + 1) * V, bits);
        }
        bounce.h2d(d_state, img.data(), img.size() * 4, stream);
        std::vector<uint32_t> limg;
        if (!cg->lane_state.empty()) {
            const size_t per = (size_t)V * cg->lpv * cg->lane_width;
            limg.resize(cg->lane_state.size() * per);
            for (size_t k = 0; k < cg->lane_state.size(); ++k)
                std::fill(limg.begin() + k * per, limg.begin() + (k + 1) * per, cg->lane_state[k].init(e));
            bounce.h2d(d_lane_state, limg.data(), limg.size() * 4, stream);
        }
        if (d_bus_phase) HIPCK(hipMemsetAsync(d_bus_phase, 0, 4, stream));
        // prepare(): every Delay gets a fresh zeroed ring sized from the sample rate (delay/mod.rs:59-69)
        for (size_t k = 0; k < cg->rings.size(); ++k) {
            const uint32_t cap = cg->rings[k].capacity(sr);
            if (cap != ring_cap[k]) {
                if (d_ring[k]) HIPCK(hipFree(d_ring[k]));
                d_ring[k] = nullptr;
                ring_cap[k] = 0;
                HIPCK(hipMalloc(&d_ring[k], (size_t)cap * V * 4));
                ring_cap[k] = cap;
            }
            HIPCK(hipMemsetAsync(d_ring[k], 0, (size_t)cap * V * 4, stream));
        }
        HIPCK(hipStreamSynchronize(stream));
    }
    size_t ring_bytes() const
    {
        size_t n = 0;
        for (size_t k = 0; k < cg->rings.size(); ++k) n += (size_t)ring_cap[k] * V * 4;
        return n;
    }

    void reset_timeline()
    {
        pending.clear();
        local_from = 0;
        h_events.clear();
        seg_begin.clear();
        seg_end.clear();
        seg_last.clear();
        cont_begin.clear();
        cont_end.clear();
        cont_last.clear();
        cont_due.clear();
        n_conts = 0;
        ev_tail = 0;
        ring_live.clear();
        ev_rebuild = false;
        n_block_local = 0;
        clear_local_counts();
        HIPCK(hipMemsetAsync(d_ev_cursor, 0, (size_t)V * 4, stream));
        HIPCK(hipMemsetAsync(d_ev_end, 0, (size_t)V * 4, stream));
    }

    void clear_local_counts()
    {
        for (uint32_t k : local_touched) local_cnt[k] = 0;
        local_touched.clear();
    }

    // Everything before this frame has been consumed on the device by the time an update issued now takes effect:
    // launched blocks run before it in stream order; blocks still in the queue have not seen their events yet.
    // Blocks per launch for throughput callers (og_render*, cluster shards, og_set_bus_batching(e, 0)): as many as the
    // launch overhead still pays for while the launch's partial-sum rows stay modest.  Measured: fm_voice at 65 536 voices
    // gains 30 % from 1 -> 8 blocks and 3.5 % more from 8 -> 32 (32 MB of rows); sat4x_voice at 131 072 voices and sub_voice
    // at 262 144 LOSE 10-25 % once the rows of one launch pass ~64 MB, with either row layout (not understood further).
    uint32_t auto_batch() const
    {
        const size_t per_block = (size_t)std::max<uint32_t>(n_wg, 1u) * 256u * 4u; // rows of one 256-frame block
        const size_t b = ((size_t)32 << 20) / per_block;
        return (uint32_t)std::min<size_t>(OG_MAX_LAUNCH_BLOCKS, std::max<size_t>(8, b));
    }
    uint64_t consumed_horizon() const { return queue.empty() ? frame_now : q_frame0; }
    // unconsumed events of voice v on the device timeline (a block consumes everything before its end)
    bool has_old_events(uint32_t v) const
    {
        if (seg_begin.empty()) return false;
        const uint64_t hz = consumed_horizon();
        return (seg_begin[v] != seg_end[v] && seg_last[v] >= hz) || (has_cont(v) && cont_last[v] >= hz);
    }
    void old_events(uint32_t v, std::vector<OgEvent>& out, uint64_t hz) const
    {
        uint32_t hb, he, cb, ce;
        unconsumed_ranges(v, hz, hb, he, cb, ce); // (binary search for the horizon: no walk over what has been played)
        out.insert(out.end(), h_events.begin() + hb, h_events.begin() + he);
        out.insert(out.end(), h_events.begin() + cb, h_events.begin() + ce);
    }
    void old_events(uint32_t v, std::vector<OgEvent>& out) const
    {
        if (!has_old_events(v)) return; // (no look at h_events: cold memory)
        old_events(v, out, consumed_horizon());
    }

    static bool push_order(const HostEvent& a, const HostEvent& b)
    {
        if (a.voice != b.voice) return a.voice < b.voice;
        if (a.frame != b.frame) return a.frame < b.frame;
        return a.seq < b.seq;
    }

    // old (already on the device) before new on equal frames: the old ones were pushed earlier
    static void merge_by_frame(const std::vector<OgEvent>& old, const HostEvent* nb, const HostEvent* ne,
                               std::vector<OgEvent>& out)
    {
        size_t i = 0;
        while (i < old.size() || nb != ne) {
            if (nb == ne || (i < old.size() && old[i].frame <= nb->frame)) {
                out.push_back(old[i++]);
            } else {
                out.push_back(OgEvent{nb->frame, nb->target, nb->value});
                ++nb;
            }
        }
    }

    void full_rebuild()
    {
        HostProf::Scope ps(prof, HostProf::REBUILD);
        std::stable_sort(pending.begin(), pending.end(), push_order);
        std::vector<OgEvent> evs;
        std::vector<uint32_t> cursor(V), end(V);
        std::vector<uint64_t> last(V, 0);
        std::vector<OgEvent> old;
        size_t p = 0;
        const size_t np = pending.size();
        const bool had = !seg_begin.empty();
        evs.reserve(np + (had ? ev_tail : 0));
        for (uint32_t v = 0; v < V; ++v) {
            cursor[v] = (uint32_t)evs.size();
            size_t q = p;
            while (q < np && pending[q].voice == v) ++q;
            if (had && (seg_begin[v] != seg_end[v] || has_cont(v))) {
                old.clear();
                old_events(v, old);
                merge_by_frame(old, pending.data() + p, pending.data() + q, evs);
            } else {
                for (size_t i = p; i < q; ++i) evs.push_back(OgEvent{pending[i].frame, pending[i].target, pending[i].value});
            }
            p = q;
            end[v] = (uint32_t)evs.size();
            if (end[v] != cursor[v]) last[v] = evs.back().frame;
        }
        const size_t n = evs.size();
        if (n > 0xFFFFFFF0ull) throw std::runtime_error("event timeline too long");
        if (n + std::max<size_t>(std::min<size_t>(EV_STAGE_EVENTS, 64), ev_reserve) > ev_cap || !d_events) {
            if (d_events) HIPCK(hipFree(d_events));
            d_events = nullptr;
            size_t headroom = (size_t)2 << 20; // room for appended segments (2 M events, 32 MB) before the ring has to wrap
            if (ev_headroom_env) headroom = ev_headroom_env; // OSCEN_GPU_EV_HEADROOM (tests: force compactions), read at og_create
            ev_cap = std::max<size_t>(n + std::max<size_t>(n / 2, ev_reserve), 1024) + headroom;
            // ring positions (seg_begin / seg_end, the device cursor and end words) are 32-bit: the ring never grows past
            // what they address -- the slack shrinks first, and a score that does not fit with its reserve is refused
            const size_t EV_CAP_MAX = 0xFFFFFFF0ull;
            if (ev_cap > EV_CAP_MAX) ev_cap = EV_CAP_MAX;
            if (n + ev_reserve > ev_cap) throw ogabi::Error(OG_E_NOMEM, "event timeline + og_reserve_events exceed the 32-bit event ring");
            HIPCK(hipMalloc(&d_events, ev_cap * sizeof(OgEvent)));
        }
        if (n) bounce.h2d(d_events, evs.data(), n * sizeof(OgEvent), stream);
        bounce.h2d(d_ev_cursor, cursor.data(), (size_t)V * 4, stream);
        bounce.h2d(d_ev_end, end.data(), (size_t)V * 4, stream);
        cont_due.clear();
        HIPCK(hipStreamSynchronize(stream)); // the staging vectors die here
        h_events.swap(evs);
        // mirror of the whole ring, reserved up front so that the live path never reallocates while playing (grown on demand:
        // no zero-fill of the room here) -- up to 2^28 events (4 GB of host memory); a ring larger than that mirrors what is
        // resident plus the promised reserve and grows on demand beyond it
        h_events.reserve(ev_cap <= ((size_t)1 << 28) ? ev_cap : std::min(ev_cap, h_events.size() + std::max<size_t>(ev_reserve, (size_t)1 << 28)));
        // ... and the first half gigabyte of that room is touched HERE, while the score is being laid out: the live path appends
        // ~150 KB of segments per block at 8 M voices, i.e. it enters a fresh 2 MB region of this mapping every dozen blocks,
        // and a first touch that has to wait for the kernel to find (compact) a huge page is a stall of milliseconds on the
        // real-time path (one 12.5 ms block in five driver-command runs of round 6, one 6.4 ms block in round 5's -- neither
        // reproduced on demand).  Half a gigabyte covers ~3 000 blocks of live playing at that size; costs ~0.1 s here.
        {
            const size_t room = (h_events.capacity() - h_events.size()) * sizeof(OgEvent);
            if (room) memset(reinterpret_cast<char*>(h_events.data()) + h_events.size() * sizeof(OgEvent), 0, std::min<size_t>(room, (size_t)512 << 20));
        }
        seg_begin.swap(cursor);
        seg_end.swap(end);
        seg_last.swap(last);
        n_conts = 0;
        cont_begin.assign(V, 0); // (everything is in the segments now; the arrays are sized here, not on the live path)
        cont_end.assign(V, 0);
        cont_last.assign(V, 0);
        ev_tail = n;
        ring_live.clear();
        for (uint32_t v = 0; v < V; ++v)
            if (seg_begin[v] != seg_end[v]) ring_live.push_back(RingSeg{v, seg_begin[v], seg_end[v]});
        pending.clear();
        local_from = 0;
        ev_rebuild = false;
        n_full_rebuilds += 1;
    }

    // One voice of an incremental batch: its pushes `mine` (sorted; may be empty: a voice whose continuation falls due) merged
    // with what it still has to play UP TO `bound` = the later of the last pushed frame and the end of the launch being
    // prepared; what lies behind stays where it is and becomes (or remains) the continuation -- the rest of a long score is
    // never copied.  A short rest is carried over instead.  A voice without pushes whose segment has been played is simply
    // pointed at its continuation: no record is written at all.  false: the batch does not fit the staging buffer.
    bool merge_voice(uint32_t v, const std::vector<HostEvent>& mine, uint64_t lend, OgEvent* sev, uint32_t* upd, size_t& n_ev, size_t& n_upd,
                     std::vector<uint32_t>& kept, std::vector<OgEvent>& old, std::vector<OgEvent>& merged)
    {
        old.clear();
        merged.clear();
        uint32_t keep_b = 0, keep_e = 0;
        if (!mine.empty() && !has_cont(v) && seg_end[v] - seg_begin[v] < CONT_MIN) {
            // the usual loaded voice: a handful of waiting events, no continuation, nothing worth leaving behind -- one pass
            // over the segment, as before round 6 (the real-time banks: ~1000 of these per block)
            const uint64_t hz = consumed_horizon();
            if (seg_last[v] >= hz)
                for (uint32_t i = seg_begin[v]; i < seg_end[v]; ++i)
                    if (h_events[i].frame >= hz) old.push_back(h_events[i]);
            merge_by_frame(old, mine.data(), mine.data() + mine.size(), merged);
            if (n_ev + merged.size() > EV_STAGE_EVENTS || n_upd >= EV_STAGE_EVENTS) return false;
            memcpy(sev + n_ev, merged.data(), merged.size() * sizeof(OgEvent));
            upd[EV_UPD_WORDS * n_upd] = v;
            upd[EV_UPD_WORDS * n_upd + 1] = (uint32_t)n_ev;
            upd[EV_UPD_WORDS * n_upd + 2] = (uint32_t)(n_ev + merged.size());
            n_ev += merged.size();
            n_upd += 1;
            return true;
        }
        const uint64_t bound = std::max<uint64_t>(mine.empty() ? 0 : mine.back().frame, lend ? lend - 1 : 0);
        uint32_t hb, he, cb, ce;
        unconsumed_ranges(v, consumed_horizon(), hb, he, cb, ce);
        if (mine.empty()) {
            if (cb == ce) return true; // (nothing left behind after all)
            if (hb == he) { // the segment in front has been played: the continuation IS the voice's segment from here on
                if (n_upd >= EV_STAGE_EVENTS) return false;
                upd[EV_UPD_WORDS * n_upd] = v;
                upd[EV_UPD_WORDS * n_upd + 1] = cb; // (absolute ring positions)
                upd[EV_UPD_WORDS * n_upd + 2] = ce;
                kept.insert(kept.end(), {(uint32_t)n_upd, 0u, 0u, 1u});
                n_upd += 1;
                return true;
            }
        }
        if (cb != ce) { // (every frame of the segment is <= every frame of the continuation)
            const uint32_t split = upper_frame(cb, ce, bound);
            old.insert(old.end(), h_events.begin() + hb, h_events.begin() + he);
            old.insert(old.end(), h_events.begin() + cb, h_events.begin() + split);
            if (ce - split >= CONT_MIN) {
                keep_b = split;
                keep_e = ce;
            } else {
                old.insert(old.end(), h_events.begin() + split, h_events.begin() + ce);
            }
        } else {
            const uint32_t split = upper_frame(hb, he, bound);
            if (he - split >= CONT_MIN) {
                old.insert(old.end(), h_events.begin() + hb, h_events.begin() + split);
                keep_b = split;
                keep_e = he;
            } else {
                old.insert(old.end(), h_events.begin() + hb, h_events.begin() + he);
            }
        }
        merge_by_frame(old, mine.data(), mine.data() + mine.size(), merged);
        if (merged.empty()) { // (only a rest behind the launch: point the voice at it, as above)
            if (keep_b == keep_e) return true;
            if (n_upd >= EV_STAGE_EVENTS) return false;
            upd[EV_UPD_WORDS * n_upd] = v;
            upd[EV_UPD_WORDS * n_upd + 1] = keep_b;
            upd[EV_UPD_WORDS * n_upd + 2] = keep_e;
            kept.insert(kept.end(), {(uint32_t)n_upd, 0u, 0u, 1u});
            n_upd += 1;
            return true;
        }
        if (n_ev + merged.size() > EV_STAGE_EVENTS || n_upd >= EV_STAGE_EVENTS) return false;
        memcpy(sev + n_ev, merged.data(), merged.size() * sizeof(OgEvent));
        upd[EV_UPD_WORDS * n_upd] = v;
        upd[EV_UPD_WORDS * n_upd + 1] = (uint32_t)n_ev; // (relative to the batch: its place in the ring is chosen at the commit)
        upd[EV_UPD_WORDS * n_upd + 2] = (uint32_t)(n_ev + merged.size());
        if (keep_b != keep_e) kept.insert(kept.end(), {(uint32_t)n_upd, keep_b, keep_e, 0u});
        n_ev += merged.size();
        n_upd += 1;
        return true;
    }

    // live pushes: O(#pushes) host work, asynchronous upload.  Returns false when the batch does not fit
    // (staging buffer, tail of d_events): the caller falls back to full_rebuild().
    bool incremental_update()
    {
        HostProf::Scope ps(prof, HostProf::INCREMENTAL);
        if (pending.size() > EV_STAGE_EVENTS) return false;
        if (!h_stage_ev[0]) {
            for (int i = 0; i < EV_RING; ++i) {
                HIPCK(hipHostMalloc((void**)&h_stage_ev[i], EV_STAGE_EVENTS * sizeof(OgEvent), hipHostMallocDefault));
                HIPCK(hipHostMalloc((void**)&h_stage_upd[i], EV_STAGE_EVENTS * EV_UPD_WORDS * 4, hipHostMallocDefault));
            }
        }
        if (seg_begin.empty()) {
            seg_begin.assign(V, 0);
            seg_end.assign(V, 0);
            seg_last.assign(V, 0);
            ensure_cont();
        }
        // group the pending pushes per voice without sorting the batch: chain them in arrival order (O(n)), then put
        // every (short) chain into (frame, push order)
        const size_t np = pending.size();
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        if (grp_head.empty()) {
            grp_head.assign(V, NONE);
            grp_tail.assign(V, NONE);
        }
        grp_next.assign(np, NONE);
        grp_voices.clear();
        for (size_t i = 0; i < np; ++i) {
            const uint32_t v = pending[i].voice;
            if (grp_head[v] == NONE) {
                grp_head[v] = (uint32_t)i;
                grp_voices.push_back(v);
            } else {
                grp_next[grp_tail[v]] = (uint32_t)i;
            }
            grp_tail[v] = (uint32_t)i;
        }
        // voices whose continuation has its first event inside the launch being prepared (and no push in this batch: a
        // pushed voice is brought up to the end of the launch anyway)
        const uint64_t lend = launch_end();
        std::vector<uint32_t> due;
        while (!cont_due.empty() && cont_due.front().first < lend) {
            const std::pair<uint64_t, uint32_t> top = cont_due.front();
            std::pop_heap(cont_due.begin(), cont_due.end(), std::greater<std::pair<uint64_t, uint32_t>>());
            cont_due.pop_back();
            const uint32_t v = top.second;
            if (!has_cont(v) || h_events[cont_begin[v]].frame != top.first) continue; // (stale: the voice was merged or re-pointed since)
            if (grp_head[v] != NONE) continue;
            due.push_back(v);
        }
        std::sort(due.begin(), due.end());
        due.erase(std::unique(due.begin(), due.end()), due.end());
        const int r = stage_head;
        {
            HostProf::Scope pw(prof, HostProf::EV_WAIT);
            batch_done(stage_seq[r]); // (EV_RING batches ago: long done)
        }
        OgEvent* sev = h_stage_ev[r];
        uint32_t* upd = h_stage_upd[r];
        std::vector<OgEvent> old, merged;
        std::vector<HostEvent> mine;
        // the updates that are not plain "new segment, no continuation" (few): {update index, begin and end of the continuation
        // the voice keeps, 1 = the update's positions are absolute ring positions (the voice is pointed at records in place)}
        std::vector<uint32_t> kept;
        size_t n_ev = 0, n_upd = 0;
        bool fits = true;
        for (const uint32_t v : grp_voices) {
            // the usual live case: nothing of this voice is waiting on the device and its pushes arrived in frame order
            // (a note-on is a frequency value and a gate on one frame) -- straight into the staging buffer
            if (fits && !has_old_events(v)) {
                size_t k = 0;
                uint64_t last = 0;
                bool ordered = true;
                for (uint32_t i = grp_head[v]; i != NONE; i = grp_next[i], ++k) {
                    const HostEvent& h = pending[i];
                    ordered = ordered && h.frame >= last;
                    last = h.frame;
                    if (n_ev + k < EV_STAGE_EVENTS) sev[n_ev + k] = OgEvent{h.frame, h.target, h.value};
                }
                if (ordered && n_ev + k <= EV_STAGE_EVENTS) {
                    grp_head[v] = NONE;
                    upd[EV_UPD_WORDS * n_upd] = v; // (cursor, end) relative to the batch: its place in the ring is chosen below
                    upd[EV_UPD_WORDS * n_upd + 1] = (uint32_t)n_ev;
                    upd[EV_UPD_WORDS * n_upd + 2] = (uint32_t)(n_ev + k);
                    n_ev += k;
                    n_upd += 1;
                    continue;
                }
            }
            mine.clear();
            for (uint32_t i = grp_head[v]; i != NONE; i = grp_next[i]) mine.push_back(pending[i]);
            grp_head[v] = NONE; // (left clean for the next batch, also on the early exit below)
            if (!fits) continue;
            for (size_t i = 1; i < mine.size(); ++i) { // insertion sort by frame; arrival order breaks ties
                const HostEvent k = mine[i];
                size_t j = i;
                while (j > 0 && mine[j - 1].frame > k.frame) {
                    mine[j] = mine[j - 1];
                    --j;
                }
                mine[j] = k;
            }
            if (!merge_voice(v, mine, lend, sev, upd, n_ev, n_upd, kept, old, merged)) fits = false;
        }
        for (const uint32_t v : due) { // (after the pushed voices: `due` holds none of them)
            if (!fits) { // (the batch goes to full_rebuild, which merges every continuation)
                break;
            }
            mine.clear();
            if (!merge_voice(v, mine, lend, sev, upd, n_ev, n_upd, kept, old, merged)) fits = false;
        }
        if (!fits) return false;
        if (n_upd == 0) { // (every due entry was stale, nothing was pushed: no kernel to launch)
            pending.clear();
            local_from = 0;
            return true;
        }
        // a place in the ring for the whole batch (the segments touched above count as superseded only after the commit,
        // so the batch never lands on the old events it was merged from)
        const size_t base = ring_alloc(n_ev);
        if (base == SIZE_MAX) return false;
        // commit: host mirror, then the device
        HostProf::Scope pc(prof, HostProf::EV_COMMIT);
        if (h_events.size() < base + n_ev) h_events.resize(base + n_ev); // (capacity ev_cap is reserved: no reallocation, no fill of the unused room)
        memcpy(h_events.data() + base, sev, n_ev * sizeof(OgEvent));
        size_t kq = 0; // (next entry of `kept`: they are in update order)
        for (size_t i = 0; i < n_upd; ++i) {
            uint32_t* u = upd + EV_UPD_WORDS * i;
            const uint32_t v = u[0];
            uint32_t kb = 0, ke = 0, in_place = 0;
            if (kq < kept.size() && kept[kq] == (uint32_t)i) {
                kb = kept[kq + 1];
                ke = kept[kq + 2];
                in_place = kept[kq + 3];
                kq += 4;
            }
            if (in_place) { // pointed at records that are in the ring already (the segment they lie in stays alive: ring_alloc)
                seg_begin[v] = u[1];
                seg_end[v] = u[2];
                seg_last[v] = h_events[u[2] - 1].frame;
                if (n_conts) set_cont(v, 0u, 0u);
                continue;
            }
            seg_last[v] = sev[u[2] - 1].frame;
            u[1] += (uint32_t)base;
            u[2] += (uint32_t)base;
            seg_begin[v] = u[1];
            seg_end[v] = u[2];
            if (kb != ke) {
                ensure_cont();
                set_cont(v, kb, ke);
                cont_due_push(v);
            } else if (n_conts) {
                set_cont(v, 0u, 0u);
            }
            ring_live.push_back(RingSeg{v, u[1], u[2]});
        }
        n_events_copied += n_ev;
        ev_tail = base;
        static_assert(sizeof(OgEvent) == sizeof(uint4), "og_apply_event_updates copies events as 16-byte words");
        const uint32_t n_wg_upd = (uint32_t)((std::max(n_upd, n_ev) + 255) / 256);
        hipLaunchKernelGGL(og_apply_event_updates, dim3(n_wg_upd), dim3(256), 0, stream, (const uint4*)sev, (uint32_t)n_ev,
                           (uint4*)(d_events + ev_tail), (const uint32_t*)upd, (uint32_t)n_upd, d_ev_cursor, d_ev_end);
        HIPCK(hipGetLastError());
        stage_seq[r] = flush_seq + 1; // read in stream order before the batch that is about to be launched
        batch_staged = true;
        stage_head = (stage_head + 1) % EV_RING;
        ev_tail += n_ev;
        pending.clear();
        local_from = 0;
        n_incremental += 1;
        return true;
    }

    // a block of `frames` frames is about to be queued: a try_push'ed event whose frame_offset >= frames is never
    // delivered (the reference clears the queues at the end of the block)
    void drop_late_local(uint32_t frames)
    {
        const size_t from = std::min(local_from, pending.size()); // pushes since the previous block sit behind this index
        local_from = pending.size();
        if (!n_block_local) return;
        const uint64_t lim = frame_now + frames;
        const size_t before = pending.size();
        pending.erase(std::remove_if(pending.begin() + (long)from, pending.end(),
                                     [&](const HostEvent& h) { return h.block_local && h.frame >= lim; }),
                      pending.end());
        dropped += before - pending.size();
        for (size_t i = from; i < pending.size(); ++i) pending[i].block_local = false; // (delivered: part of the timeline from here on)
        local_from = pending.size();
        n_block_local = 0;
        clear_local_counts();
    }
    // bring the device timeline up to date: right before the queued blocks are launched (their events may have arrived
    // over several blocks: one staging copy and one cursor update for all of them)
    void upload_events()
    {
        if (pending.empty() && !ev_rebuild && !continuation_due()) return;
        // try_push'ed events of the block that is still being assembled (pushed since the last block was queued) stay
        // on the host: drop_late_local() has not judged them against that block's length yet.  A flush in between --
        // og_process_block launches earlier async blocks, the setters launch the queue -- must not turn an event whose
        // frame_offset >= frames into one that fires in a later block.
        std::vector<HostEvent> held;
        if (n_block_local > 0) {
            const size_t from = std::min(local_from, pending.size());
            auto mid = std::stable_partition(pending.begin() + (long)from, pending.end(), [](const HostEvent& h) { return !h.block_local; });
            held.assign(mid, pending.end());
            pending.erase(mid, pending.end());
        }
        struct PutBack {
            og_engine* e;
            std::vector<HostEvent>& held;
            ~PutBack()
            {
                if (held.empty()) return;
                e->local_from = std::min(e->local_from, e->pending.size());
                e->pending.insert(e->pending.end(), held.begin(), held.end());
            }
        } put_back{this, held};
        if (pending.empty() && !ev_rebuild && !continuation_due()) return;
        HostProf::Scope ps(prof, HostProf::SYNC_EVENTS);
        // many voices touched at once (bulk scheduling): one compact CSR rebuild beats per-voice segments
        const bool bulk = ev_rebuild || pending.size() > EV_STAGE_EVENTS || pending.size() > (size_t)V / 2 + 64 || !d_events;
        if (bulk || !incremental_update()) full_rebuild();
    }

    void alloc_bus_buffers(uint32_t batch)
    {
        HIPCK(hipStreamSynchronize(stream));
        for (float** p : {&d_partials, &d_partials2, &d_mono, &d_stage_bus})
            if (*p) {
                HIPCK(hipFree(*p));
                *p = nullptr;
            }
        batch_cap = batch;
        const size_t max_frames = (size_t)OG_MAX_BLOCK * batch;
        const size_t vc = cg->voice_channels; // a Frame<2> voice output: two planes of partial rows
        HIPCK(hipMalloc(&d_partials, (size_t)n_wg * max_frames * 4 * vc));
        HIPCK(hipMemset(d_partials, 0, (size_t)n_wg * max_frames * 4 * vc));
        HIPCK(hipMalloc(&d_partials2, ((size_t)n_wg / OG_RED_GROUP + 2 + 64) * max_frames * 4));
        HIPCK(hipMalloc(&d_stage_bus, max_frames * OG_MAX_BUS_CHANNELS * 4));
        if (cg->bus_tremolo) HIPCK(hipMalloc(&d_mono, max_frames * 4));
        const size_t rows = (size_t)(cg->n_ramps + cg->n_streams);
        for (int i = 0; i < RAMP_RING && rows; ++i) {
            if (d_ramp[i]) HIPCK(hipFree(d_ramp[i]));
            if (h_ramp[i]) HIPCK(hipHostFree(h_ramp[i]));
            d_ramp[i] = h_ramp[i] = nullptr;
            HIPCK(hipMalloc(&d_ramp[i], rows * max_frames * 4));
            HIPCK(hipHostMalloc((void**)&h_ramp[i], rows * max_frames * 4, hipHostMallocDefault));
            ramp_seq[i] = 0;
        }
    }

    // process_block(frames), asynchronous: the block joins the queue; the queue is launched when it is full or when
    // something needs its results or is about to change what it would see
    void process_async(uint32_t frames, float* d_out)
    {
        HIPCK(hipSetDevice(device));
        drop_late_local(frames);
        // events wait on the host until the queue is launched: launch it before they outgrow the staging buffers (or the
        // size up to which per-voice segments beat a rebuild of the whole timeline)
        if (!queue.empty() && pending.size() > std::min<size_t>(EV_STAGE_EVENTS / 2, (size_t)V / 4 + 32)) flush_bus();
        if (queue.empty()) {
            q_frame0 = frame_now;
            q_frames = 0;
            q_ramps = cg->n_streams > 0;
            q_ramp_slot = -1;
        }
        // tick_ramps (codegen/mod.rs:878-914): the value seen by frame f is the one after f+1 ticks.  The same
        // per-frame table carries the graph's stream inputs (`<stream_in>_block`, one row each, broadcast to every
        // voice); a graph with stream inputs always runs the table-reading kernel variant.
        const size_t stride = (size_t)OG_MAX_BLOCK * batch_cap;
        const bool ramping = active_ramps > 0 && cg->n_ramps > 0;
        if ((ramping || cg->n_streams > 0 || q_ramps) && cg->n_ramps + cg->n_streams > 0) {
            if (q_ramp_slot < 0) { // first block of the queue that needs the table: earlier blocks get constant rows
                q_ramp_slot = ramp_head;
                ramp_head = (ramp_head + 1) % RAMP_RING;
                batch_done(ramp_seq[q_ramp_slot]);
                float* tab0 = h_ramp[q_ramp_slot];
                for (size_t i = 0; i < cg->inputs.size(); ++i) {
                    const int row = cg->inputs[i].ramp_row;
                    if (row >= 0) std::fill(tab0 + (size_t)row * stride, tab0 + (size_t)row * stride + q_frames, ramps[i].current);
                }
            }
            q_ramps = true;
            float* tab = h_ramp[q_ramp_slot];
            for (uint32_t f = 0; f < frames; ++f) {
                for (size_t i = 0; i < cg->inputs.size(); ++i) {
                    const int row = cg->inputs[i].ramp_row;
                    if (row < 0) continue;
                    if (active_ramps > 0 && ramps[i].tick()) active_ramps -= 1;
                    tab[(size_t)row * stride + q_frames + f] = ramps[i].current;
                }
            }
            for (size_t i = 0; i < cg->inputs.size(); ++i) {
                if (cg->inputs[i].ramp_row >= 0) values[i] = ramps[i].current;
                const int srow = cg->inputs[i].stream_row;
                for (int k = 0; srow >= 0 && k < std::max(1, cg->inputs[i].decl.channels); ++k) // (planar: one row per channel)
                    memcpy(tab + (size_t)(srow + k) * stride + q_frames, stream_blocks[i].data() + (size_t)k * OG_MAX_BLOCK, (size_t)frames * 4);
            }
        }
        QueuedBlock qb;
        qb.dst = d_out ? d_out : d_bus;
        qb.frames = frames;
        qb.trem_rate = qb.trem_depth = 0.0f;
        if (cg->bus_tremolo && bus_stage) { // voices.output -> tremolo.input; tremolo.output -> out (Frame<2>)
            ogc::UEnv ev = env();
            qb.trem_rate = cg->tremolo_rate(ev);
            qb.trem_depth = cg->tremolo_depth(ev);
        }
        queue.push_back(qb);
        q_frames += frames;
        frame_now += frames;
        last_frames = frames;
        // The last ramp ended inside this block: launch what is queued, so that the blocks that follow -- nothing moves in
        // them -- start a queue of their own on the kernel variant that does not read the table.  (Round 6, the moving-cutoff
        // variant of the bench: the launch that held the 2 205-frame cutoff ramp also held the ~20 quiet blocks queued behind
        // it and ran them 35 % slower -- per-frame parameter tests, table reads -- than the `_00` variant; a launch costs 25 us.)
        const bool ramps_done = ramping && active_ramps == 0 && cg->n_streams == 0;
        if (queue.size() >= bus_batch || n_taps > 0 || ramps_done) flush_bus();
    }

    // launch the queued blocks: voice kernel over their frames, bus reduce (fixed-association tree: groups of 1024
    // rows, then, for > 1024 waves, the group sums), post-mix stage block by block
    void flush_bus()
    {
        if (queue.empty()) return;
        upload_events(); // (before the launch arguments are formed: a rebuild may move the timeline)
        HostProf::Scope ps(prof, HostProf::LAUNCH);
        OgBlockArgs A;
        memset(&A, 0, sizeof A);
        A.n_voices = V;
        A.frames = q_frames;
        A.ramp_stride = (uint32_t)((size_t)OG_MAX_BLOCK * batch_cap);
        A.lanes = lanes;
        A.split = split;
        A.wide = wide ? 1u : 0u;
        A.frame0 = q_frame0;
        A.state = d_state;
        A.lane_state = d_lane_state;
        A.lane_dump = d_lane_state ? d_lane_state + cg->lane_state.size() * (size_t)V * cg->lpv * cg->lane_width : nullptr;
        A.events = d_events;
        A.ev_end = d_ev_end;
        A.ev_cursor = d_ev_cursor;
        A.partials = d_partials;
        const uint32_t n_chunks16 = (q_frames + OG_RED_FRAMES - 1) / OG_RED_FRAMES;
        A.partial_plane = (uint32_t)((size_t)n_chunks16 * n_wg * OG_RED_FRAMES);
        A.taps = d_taps;
        A.tap_slot = d_tap_slot;
        A.out_ev = d_out_ev;
        A.out_ev_cap = out_ev_cap;
        A.out_ev_count = d_out_ev ? d_out_ev_count : nullptr;
        A.ev_lost = d_out_ev_count ? d_out_ev_count + 1 : nullptr;
        for (size_t k = 0; k < cg->rings.size(); ++k) {
            A.rings[k] = d_ring[k];
            A.ring_cap[k] = ring_cap[k];
        }
        // block-uniform slots: the values the queued blocks were queued under (a setter launches the queue before it
        // changes one; ramped inputs are read from the table whenever a ramp moved inside the queue)
        {
            ogc::UEnv e = env();
            uint32_t starts[OG_MAX_LAUNCH_BLOCKS + 1];
            uint32_t acc = 0, nb = 0;
            for (const QueuedBlock& qb : queue) {
                if (nb < OG_MAX_LAUNCH_BLOCKS) starts[nb++] = acc;
                acc += qb.frames;
            }
            e.block_starts = starts;
            e.n_blocks = nb;
            for (const auto& up : cg->uprogs) A.slots[up.dst] = up.fn(e);
        }
        const bool ramps_on = q_ramps && q_ramp_slot >= 0;
        if (ramps_on) {
            const int r = q_ramp_slot;
            const size_t rows = (size_t)(cg->n_ramps + cg->n_streams);
            HIPCK(hipMemcpyAsync(d_ramp[r], h_ramp[r], rows * A.ramp_stride * 4, hipMemcpyHostToDevice, stream));
            ramp_seq[r] = flush_seq + 1;
            batch_staged = true;
            A.ramp_table = d_ramp[r];
        }
        const bool taps_on = n_taps > 0;
        const bool timed = timing && t_used < 8192; // (a host that never collects the timings stops adding events)
        if (timed) {
            if (t_used == t_start.size()) {
                hipEvent_t a, b;
                HIPCK(hipEventCreate(&a));
                HIPCK(hipEventCreate(&b));
                t_start.push_back(a);
                t_stop.push_back(b);
            }
            HIPCK(hipEventRecord(t_start[t_used], stream));
            if (!h_clock) {
                HIPCK(hipHostMalloc((void**)&h_clock, T_CLOCK * 4 * sizeof(unsigned long long), hipHostMallocDefault));
                memset(h_clock, 0, T_CLOCK * 4 * sizeof(unsigned long long));
            }
            A.clock_out = h_clock + 4 * t_used; // (pinned host memory is device-visible: four 8-byte stores per launch)
        }
        if (launch)
            launch(A, ramps_on, taps_on, stream);
        else
            jit->launch(A, ramps_on, taps_on, stream);
        if (timed) {
            HIPCK(hipEventRecord(t_stop[t_used], stream));
            ++t_used;
            t_blocks += queue.size();
        }
        HIPCK(hipGetLastError());
        // ---- bus: sum the partial rows ----------------------------------------------------------------
        const bool post_mix = cg->bus_tremolo && bus_stage;
        const uint32_t ch = cg->voice_channels; // summed voices: mono, or Frame<2> voices (a post-mix node writes its Frame<2> bus itself)
        bool contiguous = !post_mix;
        for (size_t k = 1; k < queue.size() && contiguous; ++k)
            contiguous = queue[k].dst == queue[k - 1].dst + (size_t)queue[k - 1].frames * ch;
        float* sum_dst = post_mix ? d_mono : (contiguous ? queue[0].dst : d_stage_bus);
        for (uint32_t c = 0; c < ch; ++c) { // one tree per channel plane; the last pass interleaves Frame<2> samples
            const float* src = d_partials + (size_t)c * A.partial_plane;
            uint32_t rows = n_wg;
            float* tmp = d_partials2;
            bus_passes = 1;
            while (rows > OG_RED_GROUP) {
                bus_passes += 1;
                const uint32_t groups = (rows + OG_RED_GROUP - 1) / OG_RED_GROUP;
                hipLaunchKernelGGL(og_bus_reduce, dim3(n_chunks16, groups), dim3(1024), 0, stream, src, rows, q_frames, tmp, 1u, 0u);
                src = tmp;
                rows = groups;
                tmp = tmp + (size_t)groups * n_chunks16 * OG_RED_FRAMES; // next level writes behind this one
            }
            hipLaunchKernelGGL(og_bus_reduce, dim3(n_chunks16, 1), dim3(1024), 0, stream, src, rows, q_frames, sum_dst, ch, c);
        }
        HIPCK(hipGetLastError());
        size_t off = 0;
        for (const QueuedBlock& qb : queue) {
            if (post_mix) {
                hipLaunchKernelGGL(og_bus_tremolo, dim3(1), dim3(512), 0, stream, d_mono + off, qb.frames, qb.trem_rate, qb.trem_depth, sr,
                                   d_bus_phase, qb.dst);
            } else if (!contiguous) {
                HIPCK(hipMemcpyAsync(qb.dst, d_stage_bus + off * ch, (size_t)qb.frames * ch * 4, hipMemcpyDeviceToDevice, stream));
            }
            off += qb.frames;
        }
        HIPCK(hipGetLastError());
        flush_seq += 1;
        if (batch_staged) { // this batch read a host staging buffer: tell the host when the stream is past it
            if (!h_progress) {
                HIPCK(hipHostMalloc((void**)&h_progress, 64, hipHostMallocCoherent)); // (fine-grained: a device store is visible to the host while the stream runs)
                *h_progress = 0;
            }
            hipLaunchKernelGGL(og_stream_mark, dim3(1), dim3(1), 0, stream, h_progress, flush_seq);
            HIPCK(hipGetLastError());
            batch_staged = false;
        }
        queue.clear();
        q_frames = 0;
    }
};

namespace {

using ogabi::guard; // Error (carries its OG_E_* code) / bad_alloc / std::exception -> code + og_last_error(); og_abi.h

int check_value_input(const og_engine* e, uint32_t input, bool per_voice)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (input >= e->cg->inputs.size()) return set_err(OG_E_INVALID, "input index out of range");
    const auto& in = e->cg->inputs[input];
    if (in.decl.kind != ogc::Kind::Value) return set_err(OG_E_INVALID, "'" + in.decl.name + "' is not a value input");
    if (in.decl.per_voice != per_voice)
        return set_err(OG_E_INVALID, "'" + in.decl.name + (per_voice ? "' is a broadcast input" : "' is a per-voice input"));
    return OG_OK;
}

int push_event(og_engine* e, uint32_t input, uint32_t voice, uint64_t frame, float value, bool local, bool setvalue)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (input >= e->cg->inputs.size()) return set_err(OG_E_INVALID, "input index out of range");
    if (voice >= e->V) return set_err(OG_E_INVALID, "voice index out of range");
    voice = e->phys(voice);
    const auto& in = e->cg->inputs[input];
    uint32_t target;
    if (setvalue) {
        if (in.decl.kind != ogc::Kind::Value || !in.decl.per_voice)
            return set_err(OG_E_INVALID, "'" + in.decl.name + "' is not a per-voice value input");
        target = OG_EV_SETVALUE | input;
    } else {
        if (in.decl.kind != ogc::Kind::Event) return set_err(OG_E_INVALID, "'" + in.decl.name + "' is not an event input");
        target = (uint32_t)in.event_index;
    }
    if (local && !setvalue) { // ArrayVec<EventInstance, 32> capacity per endpoint per block
        const size_t ne = (size_t)std::max(1, e->cg->n_event_inputs);
        if (e->local_cnt.empty()) e->local_cnt.assign((size_t)e->V * ne, 0);
        const size_t k = (size_t)voice * ne + target;
        uint8_t& cnt = e->local_cnt[k];
        if (cnt < OG_MAX_EVENTS_PER_BLOCK) {
            if (cnt == 0) e->local_touched.push_back((uint32_t)k);
            cnt += 1;
        } else {
            e->dropped += 1;
            return set_err(OG_E_OVERFLOW, "event queue full (32 per voice per input per block): event dropped");
        }
    }
    // (an event scheduled in the past fires on the first frame of the next block, like a late event on the device)
    e->pending.push_back(HostEvent{voice, std::max(frame, e->frame_now), target, value, e->seq++, local});
    if (local) e->n_block_local += 1;
    return OG_OK;
}

} // namespace

extern "C" {

const char* og_last_error(void) { return g_err.c_str(); }
const char* og_version(void)
{
    return "oscen_amd 0.1 (gfx950); settings: " OG_SETTINGS "; experiment knobs (read only when OSCEN_GPU_EXPERIMENTAL=1): " OG_EXPERIMENT_KNOBS;
}

int og_graph_new(const char* name, og_graph_desc** out)
{
    return ogabi::guard([&]() -> int {
    if (!name || !out) return set_err(OG_E_INVALID, "null argument");
    auto* g = new og_graph_desc;
    g->g.name = name;
    *out = g;
    return OG_OK;
    });
}

int og_graph_builtin(const char* name, og_graph_desc** out)
{
    if (!name || !out) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        auto* g = new og_graph_desc;
        try {
            g->g = ogc::builtin_graph(name);
        } catch (...) {
            delete g;
            throw;
        }
        *out = g;
        return OG_OK;
    });
}

int og_graph_add_input(og_graph_desc* g, const char* name, int kind, float def, uint32_t ramp_frames, uint32_t flags)
{
    return ogabi::guard([&]() -> int {
    if (!g || !name) return set_err(OG_E_INVALID, "null argument");
    if (kind < 0 || kind > 2) return set_err(OG_E_INVALID, "bad endpoint kind");
    ogc::GInput in;
    in.name = name;
    in.kind = (ogc::Kind)kind;
    in.def = def;
    in.ramp_frames = ramp_frames;
    in.per_voice = (flags & OG_IN_PER_VOICE) != 0;
    const uint32_t ch = (flags >> 8) & 0xFu; // OG_IN_CHANNELS(n): a Frame<n> stream input
    if (ch > 4 || (ch > 1 && kind != OG_KIND_STREAM)) return set_err(OG_E_INVALID, "OG_IN_CHANNELS: a stream input is an f32 or a Frame<2..4>");
    in.channels = ch > 1 ? (int)ch : 1;
    g->g.inputs.push_back(in);
    return (int)g->g.inputs.size() - 1;
    });
}

int og_graph_add_output(og_graph_desc* g, const char* name, int kind)
{
    return ogabi::guard([&]() -> int {
    if (!g || !name) return set_err(OG_E_INVALID, "null argument");
    if (kind < 0 || kind > 2) return set_err(OG_E_INVALID, "bad endpoint kind");
    g->g.outputs.push_back({name, (ogc::Kind)kind});
    return (int)g->g.outputs.size() - 1;
    });
}

int og_graph_add_node(og_graph_desc* g, const char* name, const char* type_ctor, const float* args, uint32_t n_args,
                      uint32_t rate_factor)
{
    return ogabi::guard([&]() -> int {
    if (!g || !name || !type_ctor || (n_args && !args)) return set_err(OG_E_INVALID, "null argument");
    ogc::GNode n;
    n.name = name;
    n.type = type_ctor;
    n.args.assign(args, args + n_args);
    n.rate_factor = rate_factor ? rate_factor : 1;
    g->g.nodes.push_back(n);
    return (int)g->g.nodes.size() - 1;
    });
}

int og_graph_add_node_array(og_graph_desc* g, const char* name, const char* type_ctor, const float* args, uint32_t n_args,
                            uint32_t rate_factor, uint32_t length)
{
    if (length == 0) return set_err(OG_E_INVALID, "node array length must be > 0");
    int rc = og_graph_add_node(g, name, type_ctor, args, n_args, rate_factor);
    if (rc >= 0) g->g.nodes.back().array_len = length;
    return rc;
}

int og_register_node(const og_node_type* t)
{
    if (!t || !t->type_ctor || !t->process_src || (t->n_inputs && !t->inputs) || (t->n_outputs && !t->outputs) ||
        (t->n_state && !t->state))
        return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        ogc::UserNodeType u;
        u.type = t->type_ctor;
        u.nargs = t->n_ctor_args;
        for (uint32_t i = 0; i < t->n_inputs; ++i) {
            const og_node_port& p = t->inputs[i];
            if (!p.name || p.kind < 0 || p.kind > 2) throw std::runtime_error("bad input port description");
            u.inputs.push_back({p.name, (ogc::Kind)p.kind, p.default_value, p.ctor_arg, p.channels > 1 ? (int)p.channels : 1});
            if (p.kind == OG_KIND_EVENT && t->event_handler_src && t->event_handler_src[i]) u.handlers[p.name] = t->event_handler_src[i];
        }
        for (uint32_t i = 0; i < t->n_outputs; ++i) {
            if (!t->outputs[i]) throw std::runtime_error("bad output name");
            u.outputs.push_back(t->outputs[i]);
        }
        for (uint32_t i = 0; i < t->n_state; ++i) {
            const og_node_field& f = t->state[i];
            if (!f.name) throw std::runtime_error("bad state field description");
            ogc::UserState st;
            st.name = f.name;
            st.is_uint = f.is_uint != 0;
            st.init_f = f.init;
            st.init_u = f.init_uint;
            st.arg = f.ctor_arg;
            u.state.push_back(st);
        }
        for (uint32_t i = 0; t->output_channels && i < t->n_outputs; ++i) u.out_channels.push_back((int)t->output_channels[i]);
        for (uint32_t i = 0; i < t->n_event_outputs; ++i) {
            if (!t->event_outputs || !t->event_outputs[i]) throw std::runtime_error("bad event output name");
            u.ev_outputs.push_back(t->event_outputs[i]);
        }
        u.process_src = t->process_src;
        u.weight = (int)t->cost_hint;
        if (t->event_queue_capacity > 32) throw std::runtime_error("event_queue_capacity: at most 32 (the reference's ArrayVec<EventInstance, 32>)");
        u.event_capacity = (int)t->event_queue_capacity;
        ogc::register_user_node(u);
        return OG_OK;
    });
}

int og_unregister_node(const char* type_ctor)
{
    return ogabi::guard([&]() -> int {
    if (!type_ctor) return set_err(OG_E_INVALID, "null argument");
    return ogc::unregister_user_node(type_ctor) ? OG_OK : set_err(OG_E_INVALID, std::string("no user node type '") + type_ctor + "'");
    });
}

int og_register_function(const og_function_type* f)
{
    if (!f || !f->name || !f->source || !f->n_args || !f->arg_names) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        ogc::UserFunction u;
        u.name = f->name;
        for (uint32_t k = 0; k < f->n_args; ++k) {
            if (!f->arg_names[k]) throw std::runtime_error("null argument name");
            u.arg_names.push_back(f->arg_names[k]);
            u.arg_channels.push_back(f->arg_channels && f->arg_channels[k] > 1 ? (int)f->arg_channels[k] : 1);
        }
        u.result_channels = f->result_channels > 1 ? (int)f->result_channels : 1;
        u.source = f->source;
        ogc::register_user_function(u);
        return OG_OK;
    });
}

int og_unregister_function(const char* name)
{
    return ogabi::guard([&]() -> int {
    if (!name) return set_err(OG_E_INVALID, "null argument");
    return ogc::unregister_user_function(name) ? OG_OK : set_err(OG_E_INVALID, std::string("no function '") + name + "'");
    });
}

int og_register_graph_type(const char* type_name, const og_graph_desc* g)
{
    if (!type_name || !g) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        ogc::register_graph_type(type_name, g->g);
        return OG_OK;
    });
}

int og_unregister_graph_type(const char* type_name)
{
    return ogabi::guard([&]() -> int {
    if (!type_name) return set_err(OG_E_INVALID, "null argument");
    return ogc::unregister_graph_type(type_name) ? OG_OK : set_err(OG_E_INVALID, std::string("no graph type '") + type_name + "'");
    });
}

int og_graph_add_bus_node(og_graph_desc* g, const char* name, const char* type_ctor, const float* args, uint32_t n_args)
{
    int rc = og_graph_add_node(g, name, type_ctor, args, n_args, 1);
    if (rc >= 0) g->g.nodes.back().bus = true;
    return rc;
}

int og_graph_connect(og_graph_desc* g, const char* src, const char* dst, const char* policy)
{
    return ogabi::guard([&]() -> int {
    if (!g || !src || !dst) return set_err(OG_E_INVALID, "null argument");
    g->g.edges.push_back({src, dst, policy ? policy : ""});
    return OG_OK;
    });
}

int og_graph_connect_via(og_graph_desc* g, const char* src, const char* via, const char* dst)
{
    return ogabi::guard([&]() -> int {
    if (!g || !src || !via || !dst) return set_err(OG_E_INVALID, "null argument");
    std::string v = via;
    bool numeric = !v.empty();
    for (char ch : v) numeric = numeric && isdigit((unsigned char)ch);
    if (numeric) { // `-> [N] ->`: an anonymous Delay::new(N, 0.0)  (ir/lower.rs:575-650)
        int k = 0;
        for (const auto& nd : g->g.nodes)
            if (nd.name.rfind("__inline_delay_", 0) == 0) ++k;
        ogc::GNode d;
        d.name = "__inline_delay_" + std::to_string(k);
        d.type = "Delay::new";
        d.args = {(float)atof(via), 0.0f};
        g->g.nodes.push_back(d);
        v = d.name;
    }
    ogc::GEdge in_leg{src, v + ".input", ""}, out_leg{v + ".output", dst, ""};
    out_leg.feedback = true;
    g->g.edges.push_back(in_leg);
    g->g.edges.push_back(out_leg);
    return OG_OK;
    });
}

int og_graph_parse(const char* dsl_text, const char* per_voice_inputs, og_graph_desc** out)
{
    if (!dsl_text || !out) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        std::vector<std::string> pv;
        if (per_voice_inputs) {
            std::string cur;
            for (const char* p = per_voice_inputs;; ++p) {
                if (*p == ',' || *p == 0) {
                    while (!cur.empty() && isspace((unsigned char)cur.back())) cur.pop_back();
                    while (!cur.empty() && isspace((unsigned char)cur.front())) cur.erase(cur.begin());
                    if (!cur.empty()) pv.push_back(cur);
                    cur.clear();
                    if (*p == 0) break;
                } else {
                    cur.push_back(*p);
                }
            }
        }
        std::unique_ptr<og_graph_desc> g(new og_graph_desc);
        g->g = ogc::parse_dsl(dsl_text, pv);
        *out = g.release();
        return OG_OK;
    });
}

int og_graph_poly_info(const og_graph_desc* g, uint32_t* declared_voices, char* frequency_input, char* gate_input, size_t cap)
{
    if (!g) return set_err(OG_E_INVALID, "null graph");
    int is_wrapper = 0;
    int rc = guard([&] {
        ogc::PolyInfo pi;
        (void)ogc::lower_poly_wrapper(g->g, &pi);
        is_wrapper = pi.is_wrapper ? 1 : 0;
        if (declared_voices) *declared_voices = pi.declared_voices;
        auto put = [&](char* dst, const std::string& s) {
            if (!dst || !cap) return;
            const size_t n = std::min(cap - 1, s.size());
            memcpy(dst, s.data(), n);
            dst[n] = 0;
        };
        put(frequency_input, pi.frequency_input);
        put(gate_input, pi.gate_input);
        return OG_OK;
    });
    return rc == OG_OK ? is_wrapper : rc;
}

int64_t og_graph_to_dsl(const og_graph_desc* g, char* buf, size_t cap)
{
    return ogabi::guard_value<int64_t>((int64_t)OG_E_NOMEM, [&]() -> int64_t {
    if (!g) return set_err(OG_E_INVALID, "null graph");
    const std::string t = ogc::to_dsl(g->g);
    if (buf && cap) {
        size_t n = std::min(cap - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int64_t)t.size();
    });
}

void og_graph_free(og_graph_desc* g) { delete g; }

int64_t og_graph_kernel_source(const og_graph_desc* g, char* buf, size_t cap)
{
    if (!g) return set_err(OG_E_INVALID, "null graph");
    int64_t len = -1;
    int rc = guard([&] {
        auto cg = ogc::compile(g->g);
        len = (int64_t)cg->source.size();
        if (buf && cap) {
            size_t n = std::min(cap - 1, cg->source.size());
            memcpy(buf, cg->source.data(), n);
            buf[n] = 0;
        }
        return OG_OK;
    });
    return rc == OG_OK ? len : rc;
}

int64_t og_graph_jit_check(const og_graph_desc* g, const char* arch)
{
    if (!g || !arch) return set_err(OG_E_INVALID, "null argument");
    int64_t len = -1;
    int rc = guard([&] {
        auto cg = ogc::compile(g->g);
        len = (int64_t)og_jit_compile_only(*cg, arch);
        return OG_OK;
    });
    return rc == OG_OK ? len : rc;
}

int og_create(const og_graph_desc* g, uint32_t n_voices, int device_id, og_engine** out)
{
    if (!g || !out) return set_err(OG_E_INVALID, "null argument");
    if (n_voices == 0) return set_err(OG_E_INVALID, "n_voices must be > 0");
    return guard([&] {
        std::unique_ptr<og_engine> e(new og_engine);
        e->cg = ogc::compile(g->g);
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
            throw HipError("no HIP device available (this engine has no CPU fallback)");
        if (device_id < 0 || device_id >= n_dev) throw HipError("device id out of range");
        e->device = device_id;
        HIPCK(hipSetDevice(device_id));
        e->launch = og_find_kernel(e->cg->hash);
        if (!e->launch) e->jit = og_jit_compile(*e->cg); // throws if hiprtc is unavailable or fails
        e->V = n_voices;
        // voices per wave: narrow the wave until every SIMD holds two (og_kernel_rt.hip.h)
        {
            hipDeviceProp_t prop;
            HIPCK(hipGetDeviceProperties(&prop, device_id));
            const uint32_t simds = 4u * (uint32_t)prop.multiProcessorCount;
            // Measured on MI355X (65 536 fm voices): 64 lanes 0.129 ms, 32 lanes 0.185 ms, 16 lanes
            // 0.325 ms per block -- a wave-instruction costs the same issue time however many of its
            // lanes are active, so narrowing only multiplies instructions.  Kept as an experiment knob.
            e->blocking_memcpy = ogabi::experiment_knob("OSCEN_GPU_BLOCKING_MEMCPY") != nullptr; // (environment knobs are read HERE, once)
            if (const char* hv = ogabi::experiment_knob("OSCEN_GPU_EV_HEADROOM")) e->ev_headroom_env = std::max<size_t>(64, (size_t)atoll(hv));
            uint32_t lanes = OG_WAVE;
            if (const char* ev = ogabi::experiment_knob("OSCEN_GPU_LANES")) {
                const int l = atoi(ev);
                if (l == 16 || l == 32 || l == 64) lanes = (uint32_t)l;
            }
            e->lanes = lanes;
            // Pipelined variants for banks too small to put enough ordinary waves on every SIMD.  Which depth (waves per 64
            // voices: 1, 2 or 4) runs is decided by a small model of time against waves per SIMD.  (Rounds 1-3 derived it
            // from two constants -- 4.8 cycles per instruction for a lone wave, 3.05 per SIMD otherwise -- which put the
            // four-wave pipeline behind the two-wave one at 65 536 voices; with the round-4 kernels it is 12 % ahead.)
            // Round 4: calibrated on the round-4 kernels instead of derived (profiles/r04_depth_sweep.md: fm_voice, 40
            // blocks of 256 frames per region, ms per block by bank size and depth).  With r = waves per SIMD
            // (ceil(workgroups x depth / SIMDs)) every depth is a straight line in r above a latency floor:
            //   one wave per 64 voices   max(0.074, 0.0365 r + 0.030)   r = 1..4   (16 384 .. 262 144 voices)
            //   two waves                max(0.058, 0.0255 r + 0.006)   r = 1..8
            //   four waves               max(0.034, 0.0113 r + 0.0085)  r = 1..16
            // i.e. a lone wave is bound by its own dependent-issue latency (the floors), and at saturation the pipelines
            // cost MORE SIMD time per voice than the ordinary kernel (2 x 0.0255 and 4 x 0.0113 against 0.0365: hand-offs,
            // barriers, per-wave loop and event bookkeeping) -- they win where they raise occupancy: four waves up to
            // ~131 072 voices (measured 65 536: 0.0535 against 0.0605 / 0.0753; 98 304: 0.076 against 0.097 / 0.101), the
            // ordinary kernel from ~196 608 on.  Other graphs scale the lines by their estimated cost.
            // Round 5: re-calibrated on the round-5 kernels (profiles/r05e_depth_sweep.md) and made aware of RESIDENCY.  A CU
            // holds only so many workgroups of a shape at once (registers, LDS: 16 / 8 / 6 for fm_voice's three shapes --
            // asked of the runtime per kernel, hipOccupancyMaxActiveBlocksPerMultiprocessor); a bank of more workgroups per CU
            // runs in rounds, and a short last round costs almost a full one: at 131 072 voices the four-wave shape (8
            // workgroups per CU, 6 resident) took 0.070 ms per block against 0.057 for two waves, at 196 608 (12 = two full
            // rounds) 0.086 against 0.096 and 0.094 -- the straight lines of round 4 could not see that.  Per round, time
            // against w = workgroups per CU (ms per 256-frame block, fm_voice; other graphs scale by their estimated cost,
            // which does not change the choice):
            //   one wave per 64 voices   w <= 4: 0.058; 8: 0.072; 12: 0.094; 16: 0.116
            //   two waves                w <= 4: 0.043; 6..8: 0.057
            //   four waves               1: 0.0318; 2: 0.0329; 3: 0.0345; 4: 0.0370; 6: 0.0438
            const uint32_t waves1 = (n_voices + OG_WAVE - 1) / OG_WAVE;
            const double cus = std::max(1, prop.multiProcessorCount);
            const double scale = (double)std::max(8, e->cg->valu_estimate) / 76.0; // fm_voice
            auto interp = [](const double* xs, const double* ys, int n, double x) {
                if (x <= xs[0]) return ys[0];
                for (int k = 1; k < n; ++k)
                    if (x <= xs[k]) return ys[k - 1] + (ys[k] - ys[k - 1]) * (x - xs[k - 1]) / (xs[k] - xs[k - 1]);
                return ys[n - 1] * x / xs[n - 1];
            };
            auto round_ms = [&](int d, double w) {
                static const double x1[] = {4, 8, 12, 16}, y1[] = {0.058, 0.072, 0.094, 0.116};
                static const double x2[] = {4, 6, 8}, y2[] = {0.043, 0.057, 0.057};
                static const double x4[] = {1, 2, 3, 4, 6}, y4[] = {0.0318, 0.0329, 0.0345, 0.0370, 0.0438};
                return d == 1 ? interp(x1, y1, 4, w) : (d == 2 ? interp(x2, y2, 3, w) : interp(x4, y4, 5, w));
            };
            auto resident = [&](int d) { // workgroups of this shape a CU holds at once
                int n = 0;
                if (OgOccupancyFn occ = og_find_occupancy(e->cg->hash)) n = occ(d);
                else if (e->jit) n = e->jit->occupancy(d);
                if (n <= 0) n = d == 4 ? 6 : (d == 2 ? 8 : 16);
                return (double)n;
            };
            auto cycles = [&](int d) {
                const double w = std::ceil((double)waves1 / cus); // workgroups per CU (every shape: one workgroup per 64 voices)
                const double cap = resident(d);
                const double full = std::floor(w / cap), rest = w - full * cap;
                return scale * (full * round_ms(d, cap) + (rest > 0.0 ? round_ms(d, rest) : 0.0));
            };
            uint32_t depth = 0;
            double best = cycles(1);
            if (e->cg->max_pipeline >= 2 && cycles(2) < 0.97 * best) { // (a deeper pipeline has to pay for itself)
                best = cycles(2);
                depth = 2;
            }
            if (e->cg->max_pipeline >= 4 && cycles(4) < 0.97 * best) depth = 4;
            if (const char* ev = getenv("OSCEN_GPU_SPLIT")) {
                const int want = atoi(ev);
                depth = (want >= 4 && e->cg->max_pipeline >= 4) ? 4 : ((want >= 2 && e->cg->max_pipeline >= 2) ? 2 : 0);
                if (want == 1 && e->cg->max_pipeline >= 2) depth = 2; // (old boolean meaning)
            }
            if (e->lanes != OG_WAVE) depth = 0; // the pipelined variants always run 64 voices per workgroup
            e->split = depth;
            // the wide four-wave form (16-frame hand-offs: half the barriers, twice the LDS rings) where one round holds the
            // whole bank: +2.4 .. +3.3 % at 65 536 voices (profiles/r05l_session11.log).  OSCEN_GPU_WIDE=0|1 pins it.
            e->wide = false;
            if (depth == 4 && e->cg->wide4) {
                int cap = 0;
                if (OgOccupancyFn occ = og_find_occupancy(e->cg->hash)) cap = occ(5);
                else if (e->jit) cap = e->jit->occupancy(5);
                e->wide = cap > 0 && std::ceil((double)waves1 / cus) <= (double)cap;
                if (const char* ev = getenv("OSCEN_GPU_WIDE")) e->wide = atoi(ev) != 0 && cap > 0;
            }
        }
        e->n_wg = (uint32_t)(((size_t)n_voices * e->cg->lpv + e->lanes - 1) / e->lanes);
        HIPCK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
        const auto& cg = *e->cg;
        e->values.resize(cg.inputs.size());
        e->ramps.resize(cg.inputs.size());
        for (size_t i = 0; i < cg.inputs.size(); ++i) {
            e->values[i] = cg.inputs[i].decl.def;
            e->ramps[i].set_immediate(cg.inputs[i].decl.def); // ValueRampState::new(default)
            e->ramps[i].default_frames = cg.inputs[i].decl.ramp_frames;
        }
        HIPCK(hipMalloc(&e->d_state, std::max<size_t>(1, cg.state.size()) * (size_t)n_voices * 4));
        if (!cg.lane_state.empty())
            // (+ one wave's worth of lane words behind the last plane: OgBlockArgs::lane_dump, where a lane beyond the last
            //  voice writes what an event handler of an array-valued node writes through its state planes)
            HIPCK(hipMalloc(&e->d_lane_state, cg.lane_state.size() * (size_t)n_voices * cg.lpv * cg.lane_width * 4 + (size_t)OG_WAVE * OG_LANE_DUMP_WORDS * 4));
        if (cg.bus_tremolo) HIPCK(hipMalloc(&e->d_bus_phase, 4));
        HIPCK(hipMalloc(&e->d_ev_end, (size_t)n_voices * 4));
        HIPCK(hipMalloc(&e->d_ev_cursor, (size_t)n_voices * 4));
        HIPCK(hipMemset(e->d_ev_end, 0, (size_t)n_voices * 4));
        HIPCK(hipMemset(e->d_ev_cursor, 0, (size_t)n_voices * 4));
        e->alloc_bus_buffers(1);
        HIPCK(hipMalloc(&e->d_bus, (size_t)OG_MAX_BLOCK * OG_MAX_BUS_CHANNELS * 4));
        HIPCK(hipMalloc(&e->d_tap_slot, (size_t)n_voices * 4));
        HIPCK(hipMemset(e->d_tap_slot, 0xFF, (size_t)n_voices * 4));
        if (!cg.event_outputs.empty() || cg.has_node_event_outputs) {
            HIPCK(hipMalloc(&e->d_out_ev_count, 2 * sizeof(uint32_t)));
            HIPCK(hipMemset(e->d_out_ev_count, 0, 2 * sizeof(uint32_t)));
        }
        if (!cg.event_outputs.empty()) { // room for four events per voice between two reads (OSCEN_GPU_OUT_EVENTS overrides)
            size_t cap = std::max<size_t>(65536, (size_t)n_voices * 4);
            if (const char* ev = ogabi::experiment_knob("OSCEN_GPU_OUT_EVENTS")) cap = std::max<size_t>(16, (size_t)atoll(ev));
            e->out_ev_cap = (uint32_t)std::min<size_t>(cap, 0x7FFFFFFFu);
            HIPCK(hipMalloc(&e->d_out_ev, (size_t)e->out_ev_cap * sizeof(OgOutEvent)));
        }
        e->stream_blocks.resize(cg.inputs.size());
        for (size_t i = 0; i < cg.inputs.size(); ++i)
            if (cg.inputs[i].stream_row >= 0) e->stream_blocks[i].assign((size_t)OG_MAX_BLOCK * std::max(1, cg.inputs[i].decl.channels), 0.0f);
        e->upload_initial_state(); // Graph::new(): 44.1 kHz until init()
        *out = e.release();
        return OG_OK;
    });
}

void og_destroy(og_engine* e) { delete e; }

int og_init(og_engine* e, float sample_rate)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (!(sample_rate > 0.0f)) return set_err(OG_E_INVALID, "sample rate must be positive");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->sr = sample_rate;
        e->queue.clear();
        e->q_frames = 0;
        e->upload_initial_state();
        e->reset_timeline();
        if (e->d_out_ev_count) HIPCK(hipMemsetAsync(e->d_out_ev_count, 0, 2 * sizeof(uint32_t), e->stream));
        e->out_ev_overflow = 0;
        e->out_ev_carry.clear();
        e->ev_lost_total = 0;
        e->phys_of.clear(); // (og_group_voices: a fresh state has no voice order to keep)
        e->logical_of.clear();
        e->frame_now = 0;
        e->inited = true;
        return OG_OK;
    });
}

int og_input_index(const og_engine* e, const char* name)
{
    return ogabi::guard([&]() -> int {
    if (!e || !name) return set_err(OG_E_INVALID, "null argument");
    int i = e->cg->find_input(name);
    return i >= 0 ? i : set_err(OG_E_INVALID, std::string("no input named '") + name + "'");
    });
}
uint32_t og_num_inputs(const og_engine* e) { return e ? (uint32_t)e->cg->inputs.size() : 0; }

int og_set_value(og_engine* e, uint32_t input, float v)
{
    int rc = check_value_input(e, input, false);
    if (rc) return rc;
    if (!e->queue.empty() && (rc = og_flush(e)) != OG_OK) return rc; // queued blocks keep the old value
    if (e->cg->inputs[input].ramp_row < 0) {
        e->values[input] = v;
        return OG_OK;
    }
    Ramp& r = e->ramps[input]; // set_<name>  codegen/mod.rs:931-940
    if (v != r.target) {
        if (!r.ramping()) e->active_ramps += 1;
        r.set_with_ramp(v, r.default_frames);
        e->values[input] = r.current;
    }
    return OG_OK;
}

int og_set_value_ramp(og_engine* e, uint32_t input, float v, uint32_t frames)
{
    int rc = check_value_input(e, input, false);
    if (rc) return rc;
    if (!e->queue.empty() && (rc = og_flush(e)) != OG_OK) return rc; // queued blocks keep the old value
    if (e->cg->inputs[input].ramp_row < 0) {
        e->values[input] = v;
        return OG_OK;
    }
    Ramp& r = e->ramps[input]; // set_<name>_with_ramp  codegen/mod.rs:944-953
    if (v != r.target) {
        if (frames > 0 && !r.ramping()) e->active_ramps += 1;
        // (the reference leaves active_ramps untouched when a running ramp is cut by
        //  frames == 0; the counter only gates ticking, so results are unaffected)
        if (frames == 0 && r.ramping()) e->active_ramps -= 1;
        r.set_with_ramp(v, frames);
        e->values[input] = r.current;
    }
    return OG_OK;
}

int og_set_value_immediate(og_engine* e, uint32_t input, float v)
{
    int rc = check_value_input(e, input, false);
    if (rc) return rc;
    if (!e->queue.empty() && (rc = og_flush(e)) != OG_OK) return rc; // queued blocks keep the old value
    if (e->cg->inputs[input].ramp_row >= 0) { // set_<name>_immediate  codegen/mod.rs:957-963
        Ramp& r = e->ramps[input];
        if (r.ramping()) e->active_ramps -= 1;
        r.set_immediate(v);
    }
    e->values[input] = v;
    return OG_OK;
}

int og_get_value(const og_engine* e, uint32_t input, float* out)
{
    int rc = check_value_input(e, input, false);
    if (rc) return rc;
    if (!out) return set_err(OG_E_INVALID, "null argument");
    *out = e->values[input];
    return OG_OK;
}

int og_ramp_state(const og_engine* e, uint32_t input, float* current, float* target, uint32_t* frames_remaining)
{
    int rc = check_value_input(e, input, false);
    if (rc) return rc;
    const Ramp& r = e->ramps[input];
    const bool ramped = e->cg->inputs[input].decl.ramp_frames > 0 || r.ramping();
    if (current) *current = ramped ? r.current : e->values[input];
    if (target) *target = ramped ? r.target : e->values[input];
    if (frames_remaining) *frames_remaining = r.frames_remaining;
    return OG_OK;
}
uint32_t og_active_ramps(const og_engine* e) { return e ? e->active_ramps : 0; }

int og_set_voice_values(og_engine* e, uint32_t input, uint32_t first, uint32_t count, const float* v)
{
    int rc = check_value_input(e, input, true);
    if (rc) return rc;
    if (!v || (uint64_t)first + count > e->V) return set_err(OG_E_INVALID, "voice range out of bounds");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        const size_t w = (size_t)e->cg->inputs[input].state_word;
        if (e->phys_of.empty()) {
            e->bounce.h2d(e->d_state + w * e->V + first, v, (size_t)count * 4, e->stream);
        } else if (count <= 64u) { // grouped voices, a few values: one word each
            for (uint32_t i = 0; i < count; ++i) e->bounce.h2d(e->d_state + w * e->V + e->phys_of[first + i], v + i, 4, e->stream);
        } else { // grouped voices: the range is scattered over the plane
            std::vector<float> plane(e->V);
            if (count < e->V) {
                e->bounce.d2h(plane.data(), e->d_state + w * e->V, (size_t)e->V * 4, e->stream);
                HIPCK(hipStreamSynchronize(e->stream));
            }
            for (uint32_t i = 0; i < count; ++i) plane[e->phys_of[first + i]] = v[i];
            e->bounce.h2d(e->d_state + w * e->V, plane.data(), (size_t)e->V * 4, e->stream);
        }
        HIPCK(hipStreamSynchronize(e->stream));
        return OG_OK;
    });
}

int og_set_voice_value(og_engine* e, uint32_t input, uint32_t voice, float v)
{
    return og_set_voice_values(e, input, voice, 1, &v);
}

int og_push_voice_event(og_engine* e, uint32_t input, uint32_t voice, uint32_t frame_offset, float scalar)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return push_event(e, input, voice, e->frame_now + frame_offset, scalar, true, false);
}
int og_push_voice_value(og_engine* e, uint32_t input, uint32_t voice, uint32_t frame_offset, float v)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return push_event(e, input, voice, e->frame_now + frame_offset, v, true, true);
}
int og_schedule_voice_event(og_engine* e, uint32_t input, uint32_t voice, uint64_t abs_frame, float scalar)
{
    return push_event(e, input, voice, abs_frame, scalar, false, false);
}
int og_schedule_voice_value(og_engine* e, uint32_t input, uint32_t voice, uint64_t abs_frame, float v)
{
    return push_event(e, input, voice, abs_frame, v, false, true);
}

int og_schedule_voice_events(og_engine* e, uint32_t input, uint32_t n, const uint32_t* voices, const uint64_t* abs_frames,
                             const float* values)
{
    return ogabi::guard([&]() -> int {
    if (!e || (n && (!voices || !abs_frames || !values))) return set_err(OG_E_INVALID, "null argument");
    if (input >= e->cg->inputs.size()) return set_err(OG_E_INVALID, "input index out of range");
    const auto& in = e->cg->inputs[input];
    uint32_t target;
    if (in.decl.kind == ogc::Kind::Event) target = (uint32_t)in.event_index;
    else if (in.decl.kind == ogc::Kind::Value && in.decl.per_voice) target = OG_EV_SETVALUE | input;
    else return set_err(OG_E_INVALID, "'" + in.decl.name + "' is neither an event input nor a per-voice value input");
    for (uint32_t i = 0; i < n; ++i)
        if (voices[i] >= e->V) return set_err(OG_E_INVALID, "voice index out of range");
    e->pending.reserve(e->pending.size() + n);
    for (uint32_t i = 0; i < n; ++i)
        e->pending.push_back(HostEvent{e->phys(voices[i]), std::max(abs_frames[i], e->frame_now), target, values[i], e->seq++, false});
    return OG_OK;
    });
}

int og_process_block_async(og_engine* e, uint32_t frames, float* d_out_bus)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before processing");
    if (frames > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "frames must be in 0..512");
    return guard([&] {
        if (frames == 0) { // process_block(0): no frame runs; events queued for the block are discarded with it
            HIPCK(hipSetDevice(e->device));
            e->flush_bus();
            e->drop_late_local(0);
            e->last_frames = 0;
            return OG_OK;
        }
        e->process_async(frames, d_out_bus);
        return OG_OK;
    });
}

// n consecutive og_process_block_async calls in one: a render loop that has nothing to say between its blocks (offline
// rendering, a streaming caller that queues a buffer's worth) crosses the boundary once.  Same queue, same results.
int og_process_blocks_async(og_engine* e, uint32_t frames, uint32_t n_blocks, float* d_out_bus, size_t out_stride_bytes)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before processing");
    if (frames == 0 || frames > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "frames must be in 1..512");
    if (d_out_bus && out_stride_bytes % sizeof(float)) return set_err(OG_E_INVALID, "out_stride_bytes must be a multiple of 4");
    return guard([&] {
        for (uint32_t b = 0; b < n_blocks; ++b)
            e->process_async(frames, d_out_bus ? d_out_bus + (size_t)b * (out_stride_bytes / sizeof(float)) : nullptr);
        return OG_OK;
    });
}

int og_synchronize(og_engine* e)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        HIPCK(hipStreamSynchronize(e->stream));
        e->sync_lost_counter(); // (graphs with in-voice event queues only)
        return OG_OK;
    });
}

int og_set_bus_batching(og_engine* e, uint32_t blocks)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (blocks > OG_MAX_LAUNCH_BLOCKS) return set_err(OG_E_INVALID, "bus batching: 0 (automatic) or 1..32 blocks");
    if (blocks == 0) blocks = e->auto_batch();
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        if (blocks > e->batch_cap) e->alloc_bus_buffers(blocks);
        e->bus_batch = blocks;
        return OG_OK;
    });
}

int og_flush(og_engine* e)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        return OG_OK;
    });
}

// The blocking drop-in entry (the reference's process_block + reading the output field): the bus of this one block is
// written by the device straight into pinned host memory, and completion is a word of pinned memory the stream marker
// writes -- no hipMemcpy, no hipStreamSynchronize (both cost tens of microseconds of runtime latency per block).
int og_process_block(og_engine* e, uint32_t frames, float* out_bus)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before processing");
    if (frames > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "frames must be in 0..512");
    if (frames == 0 || !out_bus || e->n_taps > 0 || e->blocking_memcpy) { // (taps are read with a stream sync anyway)
        int rc = og_process_block_async(e, frames, nullptr);
        if (rc) return rc;
        return guard([&] {
            e->flush_bus();
            if (out_bus && frames)
                e->bounce.d2h(out_bus, e->d_bus, (size_t)frames * e->cg->channels * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
            return OG_OK;
        });
    }
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus(); // earlier async blocks keep their own destinations
        if (!e->h_bus_pinned) HIPCK(hipHostMalloc((void**)&e->h_bus_pinned, (size_t)OG_MAX_BLOCK * OG_MAX_BUS_CHANNELS * 4, hipHostMallocDefault));
        // ask for the stream marker behind this block's launch BEFORE the block is queued: with one block per launch
        // (the default) process_async launches it itself, and a marker requested afterwards would never be written
        // (the wait then ran into its 20 ms timeout on every call -- ADVICE r2)
        e->batch_staged = true;
        const uint64_t seq = e->flush_seq + 1;
        e->process_async(frames, e->h_bus_pinned);
        e->flush_bus();
        e->blocking_waits += 1;
        if (!e->wait_progress(seq)) e->blocking_timeouts += 1;
        memcpy(out_bus, e->h_bus_pinned, (size_t)frames * e->cg->channels * 4);
        return OG_OK;
    });
}

int og_set_stream(og_engine* e, void* s)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        HIPCK(hipStreamSynchronize(e->stream));
        if (e->own_stream) HIPCK(hipStreamDestroy(e->stream));
        e->own_stream = false;
        e->stream = (hipStream_t)s;
        return OG_OK;
    });
}

int og_render(og_engine* e, uint64_t total_frames, uint32_t block, float* out_bus)
{
    if (!e || !out_bus) return set_err(OG_E_INVALID, "null argument");
    if (block == 0 || block > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "block must be in 1..512");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before processing");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        const uint32_t ch = e->cg->channels;
        float* d_all = nullptr;
        HIPCK(hipMalloc(&d_all, (size_t)total_frames * ch * 4));
        // an offline render has nothing between its blocks: several per launch (the caller's setting is restored)
        e->flush_bus();
        const uint32_t user_batch = e->bus_batch;
        if (e->batch_cap < e->auto_batch()) e->alloc_bus_buffers(e->auto_batch());
        e->bus_batch = e->auto_batch();
        struct Restore {
            og_engine* e;
            uint32_t b;
            ~Restore() { e->bus_batch = b; }
        } restore{e, user_batch};
        try {
            for (uint64_t f0 = 0; f0 < total_frames; f0 += block) {
                const uint32_t frames = (uint32_t)std::min<uint64_t>(block, total_frames - f0);
                e->process_async(frames, d_all + f0 * ch);
            }
            e->flush_bus();
            e->bounce.d2h(out_bus, d_all, (size_t)total_frames * ch * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
        } catch (...) {
            (void)hipFree(d_all);
            throw;
        }
        HIPCK(hipFree(d_all));
        return OG_OK;
    });
}

int og_set_stream_block(og_engine* e, uint32_t input, const float* samples, uint32_t n)
{
    return ogabi::guard([&]() -> int {
    if (!e || (n && !samples)) return set_err(OG_E_INVALID, "null argument");
    if (input >= e->cg->inputs.size() || e->cg->inputs[input].stream_row < 0) return set_err(OG_E_INVALID, "not a stream input");
    if (n > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "a stream block holds at most 512 samples");
    const uint32_t w = (uint32_t)std::max(1, e->cg->inputs[input].decl.channels); // Frame<N>: `samples` holds n frames of N, interleaved
    float* blk = e->stream_blocks[input].data();
    for (uint32_t k = 0; k < w; ++k)
        for (uint32_t j = 0; j < n; ++j) blk[(size_t)k * OG_MAX_BLOCK + j] = samples[(size_t)j * w + k];
    return OG_OK;
    });
}

uint32_t og_num_stream_inputs(const og_engine* e) { return e ? (uint32_t)e->cg->n_stream_inputs : 0; }

uint32_t og_stream_input_channels(const og_engine* e, uint32_t input)
{
    return ogabi::guard_value<uint32_t>((uint32_t)0, [&]() -> uint32_t {
    if (!e || input >= e->cg->inputs.size() || e->cg->inputs[input].stream_row < 0) return 0;
    return (uint32_t)std::max(1, e->cg->inputs[input].decl.channels);
    });
}

int og_render_inputs(og_engine* e, const float* const* inputs, const uint64_t* input_lens, uint32_t n_inputs, uint64_t tail,
                     float* out_bus, uint64_t* frames_rendered)
{
    if (!e || (n_inputs && (!inputs || !input_lens))) return set_err(OG_E_INVALID, "null argument");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before processing");
    if (n_inputs != (uint32_t)e->cg->n_stream_inputs) // render(): assert_eq!(inputs.len(), NUM_STREAM_INPUTS)
        return set_err(OG_E_INVALID, "render: expected " + std::to_string(e->cg->n_stream_inputs) + " input streams, got " + std::to_string(n_inputs));
    uint64_t in_len = 0;
    for (uint32_t i = 0; i < n_inputs; ++i) in_len = std::max(in_len, input_lens[i]);
    const uint64_t total = in_len + tail;
    if (frames_rendered) *frames_rendered = total;
    if (total == 0) return OG_OK;
    if (!out_bus) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        std::vector<int> sidx; // stream inputs in declaration order
        for (size_t i = 0; i < e->cg->inputs.size(); ++i)
            if (e->cg->inputs[i].stream_row >= 0) sidx.push_back((int)i);
        const uint32_t ch = e->cg->channels;
        float* d_all = nullptr;
        HIPCK(hipMalloc(&d_all, (size_t)total * ch * 4));
        e->flush_bus();
        const uint32_t user_batch = e->bus_batch;
        if (e->batch_cap < e->auto_batch()) e->alloc_bus_buffers(e->auto_batch());
        e->bus_batch = e->auto_batch();
        struct Restore {
            og_engine* e;
            uint32_t b;
            ~Restore() { e->bus_batch = b; }
        } restore{e, user_batch};
        try {
            for (uint64_t pos = 0; pos < total; pos += OG_MAX_BLOCK) { // chunks of DEFAULT_MAX_BLOCK_SIZE, offline.rs:71-90
                const uint32_t n = (uint32_t)std::min<uint64_t>(OG_MAX_BLOCK, total - pos);
                for (uint32_t k = 0; k < n_inputs; ++k) {
                    float* blk = e->stream_blocks[sidx[k]].data();
                    const uint32_t w = (uint32_t)std::max(1, e->cg->inputs[sidx[k]].decl.channels); // (lengths count FRAMES)
                    for (uint32_t c = 0; c < w; ++c)
                        for (uint32_t j = 0; j < n; ++j) // silence past the end
                            blk[(size_t)c * OG_MAX_BLOCK + j] = (pos + j < input_lens[k]) ? inputs[k][(pos + j) * w + c] : 0.0f;
                }
                e->process_async(n, d_all + pos * ch);
            }
            e->flush_bus();
            e->bounce.d2h(out_bus, d_all, (size_t)total * ch * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
        } catch (...) {
            (void)hipFree(d_all);
            throw;
        }
        HIPCK(hipFree(d_all));
        return OG_OK;
    });
}

// `graph.<node>.<field>`: the generated struct's node fields are public in the reference and its tests read them
// (`graph.sinks[i].last`, `graph.inner.dummy.val`).  Here a node's persistent fields are planes of the state image.
static int find_state_word(const og_engine* e, const char* path)
{
    // the reference's spelling -> the names of the flattened graph: `sinks[1].last` -> `sinks__1.last`,
    // `inner.dummy.val` -> `inner_dummy.val` (the last dot separates node and field)
    std::string p;
    for (const char* c = path; *c; ++c) {
        if (*c == '[') p += "__";
        else if (*c != ']' && !isspace((unsigned char)*c)) p.push_back(*c);
    }
    const size_t last = p.rfind('.');
    for (size_t k = 0; k < p.size(); ++k)
        if (p[k] == '.' && k != last) p[k] = '_';
    for (size_t w = 0; w < e->cg->state.size(); ++w)
        if (e->cg->state[w].name == p) return (int)w;
    return -1;
}

int og_state_field_index(const og_engine* e, const char* path)
{
    if (!e || !path) return -1;
    return find_state_word(e, path);
}

int og_read_state_field(og_engine* e, const char* path, uint32_t first_voice, uint32_t n, void* out)
{
    if (!e || !path || (n && !out)) return set_err(OG_E_INVALID, "null argument");
    const int w = find_state_word(e, path);
    if (w < 0) return set_err(OG_E_INVALID, std::string("no state field '") + path + "'");
    if ((uint64_t)first_voice + n > e->V) return set_err(OG_E_INVALID, "voice range out of bounds");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        if (n && e->phys_of.empty()) {
            e->bounce.d2h(out, e->d_state + (size_t)w * e->V + first_voice, (size_t)n * 4, e->stream);
        } else if (n) { // grouped voices: gather
            std::vector<uint32_t> plane(e->V);
            e->bounce.d2h(plane.data(), e->d_state + (size_t)w * e->V, (size_t)e->V * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
            for (uint32_t i = 0; i < n; ++i) ((uint32_t*)out)[i] = plane[e->phys_of[first_voice + i]];
        }
        HIPCK(hipStreamSynchronize(e->stream));
        return OG_OK;
    });
}

int og_set_voice_taps(og_engine* e, const uint32_t* voices, uint32_t n)
{
    if (!e || (n && !voices)) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        HIPCK(hipStreamSynchronize(e->stream));
        std::vector<int32_t> slot(e->V, -1);
        for (uint32_t i = 0; i < n; ++i) {
            if (voices[i] >= e->V) throw std::runtime_error("tap voice out of range");
            slot[e->phys(voices[i])] = (int32_t)i;
        }
        e->bounce.h2d(e->d_tap_slot, slot.data(), (size_t)e->V * 4, e->stream);
        HIPCK(hipStreamSynchronize(e->stream));
        if (e->d_taps) HIPCK(hipFree(e->d_taps));
        e->d_taps = nullptr;
        if (n) {
            // (a Frame<2> voice output: [tap][frame][2])
            HIPCK(hipMalloc(&e->d_taps, (size_t)n * OG_MAX_BLOCK * 4 * e->cg->voice_channels));
            HIPCK(hipMemset(e->d_taps, 0, (size_t)n * OG_MAX_BLOCK * 4 * e->cg->voice_channels));
        }
        e->n_taps = n;
        e->tap_voices.assign(voices, voices + n); // the caller's voice numbers: og_load_state re-resolves them when it adopts another slot order
        return OG_OK;
    });
}

int og_read_voice_taps(og_engine* e, float* out, uint32_t n, uint32_t frames)
{
    if (!e || !out) return set_err(OG_E_INVALID, "null argument");
    if (n > e->n_taps || frames != e->last_frames) return set_err(OG_E_INVALID, "taps: n/frames do not match the last block");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        e->bounce.d2h(out, e->d_taps, (size_t)n * frames * 4 * e->cg->voice_channels, e->stream);
        HIPCK(hipStreamSynchronize(e->stream));
        return OG_OK;
    });
}

uint32_t og_channels(const og_engine* e) { return e ? e->cg->channels : 0; }
uint32_t og_voice_channels(const og_engine* e) { return e ? e->cg->voice_channels : 0; }
int og_output_channel(const og_engine* e, const char* name, uint32_t* offset, uint32_t* width)
{
    return ogabi::guard([&]() -> int {
    if (!e || !name) return set_err(OG_E_INVALID, "null argument");
    for (const auto& oc : e->cg->output_channels)
        if (oc.name == name) {
            if (offset) *offset = (uint32_t)oc.offset;
            if (width) *width = (uint32_t)oc.width;
            return OG_OK;
        }
    return set_err(OG_E_INVALID, std::string("no stream output named '") + name + "' on the bus");
    });
}
uint32_t og_num_voices(const og_engine* e) { return e ? e->V : 0; }
uint32_t og_latency_samples(const og_engine* e) { return e ? e->cg->latency_samples : 0; }
// (Tremolo is the only post-mix node type the compiler takes -- og_graph.cpp, bus nodes -- and it is linear in its input)
int og_post_mix_kind(const og_engine* e) { return (e && e->cg->bus_tremolo) ? 1 : 0; }
uint64_t og_frames_processed(const og_engine* e) { return e ? e->frame_now : 0; }
uint32_t og_state_words_per_voice(const og_engine* e)
{
    return e ? (uint32_t)(e->cg->state.size() + e->cg->lane_state.size() * e->cg->lpv * e->cg->lane_width) : 0;
}
uint32_t og_state_words_written_per_voice(const og_engine* e)
{
    if (!e) return 0;
    uint32_t n = 0;
    for (const auto& w : e->cg->state) n += w.read_mostly ? 0u : 1u;
    for (const auto& w : e->cg->lane_state) n += w.read_mostly ? 0u : (uint32_t)(e->cg->lpv * e->cg->lane_width);
    return n;
}
uint32_t og_lanes_per_voice(const og_engine* e) { return e ? (uint32_t)e->cg->lpv : 0; }
int og_uses_split_kernel(const og_engine* e) { return e ? (int)e->split : 0; }
uint32_t og_voices_per_wave(const og_engine* e) { return e ? e->lanes / (uint32_t)e->cg->lpv : 0; }
// Side-effect free (a monitoring thread may poll it while the audio thread renders): host-side drops plus what
// og_sync_event_counters() last folded in from the device.
uint64_t og_events_dropped(const og_engine* e) { return e ? e->dropped + e->ev_lost_total : 0; }

// in-voice event queues (#[output(event)] fields of user nodes) hold OG_NODE_EVENTS_PER_FRAME events per frame and
// output; what they could not hold is counted on the device.  This launches the queued blocks, reads that counter back
// (one small copy + a stream synchronise; graphs with such nodes only) and folds it into og_events_dropped().  Called
// by the thread that renders; og_synchronize() and og_read_output_events() do it as part of their own synchronise.
int og_sync_event_counters(og_engine* e)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    return guard([&] {
        e->sync_lost_counter();
        return OG_OK;
    });
}

uint32_t og_num_event_outputs(const og_engine* e) { return e ? (uint32_t)e->cg->event_outputs.size() : 0; }
int og_event_output_index(const og_engine* e, const char* name)
{
    return ogabi::guard([&]() -> int {
    if (!e || !name) return set_err(OG_E_INVALID, "null argument");
    for (size_t i = 0; i < e->cg->event_outputs.size(); ++i)
        if (e->cg->event_outputs[i] == name) return (int)i;
    return set_err(OG_E_INVALID, std::string("no event output named '") + name + "'");
    });
}

void og_engine::sync_lost_counter()
{
    if (!(d_out_ev_count && cg->has_node_event_outputs && inited)) return;
    HIPCK(hipSetDevice(device));
    flush_bus();
    uint32_t lost = 0;
    HIPCK(hipMemcpyAsync(&lost, d_out_ev_count + 1, sizeof lost, hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemsetAsync(d_out_ev_count + 1, 0, sizeof(uint32_t), stream));
    HIPCK(hipStreamSynchronize(stream));
    ev_lost_total += lost;
}

// Drains the device log into a host-side queue and hands out the oldest `cap` events; what does not fit the caller's
// buffer STAYS queued for the next call (round 3 threw it away).  `n_overflowed` counts only what the device log itself
// could not hold.
int og_read_output_events(og_engine* e, og_out_event* buf, uint32_t cap, uint32_t* n_out, uint64_t* n_overflowed)
{
    if (!e || (cap && !buf) || !n_out) return set_err(OG_E_INVALID, "null argument");
    *n_out = 0;
    if (n_overflowed) *n_overflowed = 0;
    if (!e->d_out_ev) return OG_OK; // (a graph without event outputs: nothing ever arrives)
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        uint32_t counters[2] = {0u, 0u}; // {events logged, in-voice pushes lost}
        HIPCK(hipMemcpyAsync(counters, e->d_out_ev_count, sizeof counters, hipMemcpyDeviceToHost, e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
        const uint32_t count = counters[0];
        const uint32_t have = std::min(count, e->out_ev_cap);
        const size_t old = e->out_ev_carry.size();
        e->out_ev_carry.resize(old + have);
        try {
            if (have) e->bounce.d2h(e->out_ev_carry.data() + old, e->d_out_ev, (size_t)have * sizeof(OgOutEvent), e->stream);
            HIPCK(hipMemsetAsync(e->d_out_ev_count, 0, 2 * sizeof(uint32_t), e->stream));
            HIPCK(hipStreamSynchronize(e->stream));
        } catch (...) {
            e->out_ev_carry.resize(old); // (a failed copy leaves no half-filled records in the queue)
            throw;
        }
        e->out_ev_overflow += count - have;
        e->ev_lost_total += counters[1];
        if (!e->logical_of.empty()) // grouped voices: the log holds physical slots
            for (size_t i = old; i < e->out_ev_carry.size(); ++i) e->out_ev_carry[i].voice = e->logical_of[e->out_ev_carry[i].voice];
        // frame order; within a frame voice order; a voice's events of one frame in push order (the append index of
        // one lane grows in program order).  Events carried over from an earlier call lie on earlier frames.
        std::sort(e->out_ev_carry.begin() + (ptrdiff_t)old, e->out_ev_carry.end(), [](const OgOutEvent& a, const OgOutEvent& b) {
            if (a.frame != b.frame) return a.frame < b.frame;
            if (a.voice != b.voice) return a.voice < b.voice;
            return a.seq < b.seq;
        });
        const uint32_t give = (uint32_t)std::min<size_t>(e->out_ev_carry.size(), cap);
        for (uint32_t i = 0; i < give; ++i) {
            const OgOutEvent& x = e->out_ev_carry[i];
            buf[i] = og_out_event{x.voice, x.output, x.frame, x.value, 0u};
        }
        e->out_ev_carry.erase(e->out_ev_carry.begin(), e->out_ev_carry.begin() + give);
        *n_out = give;
        if (n_overflowed) {
            *n_overflowed = e->out_ev_overflow;
            e->out_ev_overflow = 0;
        }
        return OG_OK;
    });
}
uint64_t og_kernel_hash(const og_engine* e) { return e ? e->cg->hash : 0; }
int og_kernel_is_jit(const og_engine* e) { return e ? (e->launch ? 0 : 1) : 0; }
uint32_t og_bus_reduce_passes(const og_engine* e) { return e ? e->bus_passes : 0; }
uint32_t og_partial_rows(const og_engine* e) { return e ? e->n_wg : 0; }
const char* og_kernel_name(const og_engine* e)
{
    static thread_local char buf[64];
    if (!e) return "";
    snprintf(buf, sizeof buf, "og_k%s_%016llx", e->split == 4 ? (e->wide ? "4w" : "4") : (e->split == 2 ? "2" : ""), (unsigned long long)e->cg->hash);
    return buf;
}
int og_event_stats(const og_engine* e, uint64_t* full_rebuilds, uint64_t* incremental_updates, uint64_t* resident_events)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (full_rebuilds) *full_rebuilds = e->n_full_rebuilds;
    if (incremental_updates) *incremental_updates = e->n_incremental;
    if (resident_events) *resident_events = (uint64_t)e->ev_tail;
    return OG_OK;
}

uint64_t og_event_ring_wraps(const og_engine* e) { return e ? e->n_ring_wraps : 0; }
uint64_t og_events_copied(const og_engine* e) { return e ? e->n_events_copied : 0; }

int og_reserve_events(og_engine* e, uint64_t n_events)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (n_events > 0xF0000000ull) return set_err(OG_E_INVALID, "og_reserve_events: at most 2^32 - 2^28 events");
    e->ev_reserve = (size_t)n_events;
    e->ev_rebuild = e->ev_rebuild || (e->d_events && e->ev_tail + e->ev_reserve > e->ev_cap); // takes effect with the next rebuild
    return OG_OK;
}

// Voice grouping for resident scores.  A wave renders 64 consecutive voice slots and takes, chunk by chunk, the cheapest body
// ALL of its lanes allow: one releasing lane puts the whole wave on the release arithmetic, one event or envelope stage end
// on the checked body.  With the voices of a bank in arbitrary order nearly every wave holds a releasing lane nearly all
// the time.  og_group_voices re-orders the SLOTS so that voices whose notes end at about the same time share waves --
// policy 1: by the frame of the voice's first scheduled note-off (an event-input event with a value <= 0), then by its
// first event of any kind -- and keeps a logical -> physical table: every entry point still takes and hands out the
// caller's voice numbers, per-voice samples are bit for bit those of the ungrouped bank, the summed bus differs by the
// association of the sum only.  Policy 0 restores the identity.  To be called after og_init and after the score has been
// scheduled, before the first block (the only per-voice state then are the per-voice value inputs, which move along).
int og_group_voices(og_engine* e, uint32_t policy)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (policy > 2u) return set_err(OG_E_INVALID, "og_group_voices: policy 0 (identity), 1 (by first note-off) or 2 (1 + waves dealt out by weight)");
    if (!e->inited) return set_err(OG_E_STATE, "og_init must be called before og_group_voices");
    if (e->frame_now != 0 || !e->queue.empty() || !e->h_events.empty() || e->n_block_local != 0)
        return set_err(OG_E_STATE, "og_group_voices: only before the first block (and before block-local pushes)");
    if (e->n_taps != 0) return set_err(OG_E_STATE, "og_group_voices: set the voice taps after grouping");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        const uint32_t V = e->V;
        std::vector<uint32_t> new_phys(V);
        if (policy == 0u) {
            for (uint32_t v = 0; v < V; ++v) new_phys[v] = v;
        } else {
            const uint64_t NONE = ~(uint64_t)0;
            std::vector<uint64_t> first_off(V, NONE), first_any(V, NONE);
            for (const HostEvent& h : e->pending) {
                const uint32_t v = e->logical(h.voice);
                first_any[v] = std::min(first_any[v], h.frame);
                if (!(h.target & OG_EV_SETVALUE) && h.value <= 0.0f) first_off[v] = std::min(first_off[v], h.frame);
            }
            std::vector<uint32_t> order(V);
            for (uint32_t v = 0; v < V; ++v) order[v] = v;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                if (first_off[a] != first_off[b]) return first_off[a] < first_off[b];
                return first_any[a] < first_any[b];
            });
            if (policy == 2u && V >= 2u * 256u * OG_WAVE) {
                // Policy 2 (experiment, not measured): the same waves, DEALT OUT.  With every workgroup of a launch resident
                // at once (65 536 voices = 1 024 workgroups = 4 per CU) a launch ends with the busiest CU; policy 1 leaves
                // the waves that hold the score's events next to each other in dispatch order.  Here the groups of 64 slots
                // are ranked by the number of events they hold and placed boustrophedon over rows of 256 groups -- row 0
                // left to right, row 1 right to left, ... -- so that groups i, i + 256, i + 512, ... (which share a CU if
                // workgroups are handed to the 256 CUs round robin: an assumption) add up to about the same weight.
                const uint32_t W = (V + OG_WAVE - 1) / OG_WAVE, ROW = 256u;
                std::vector<uint32_t> weight(W, 0u), rank(W);
                std::vector<uint32_t> slot_of(V);
                for (uint32_t r = 0; r < V; ++r) slot_of[order[r]] = r;
                for (const HostEvent& h : e->pending) weight[slot_of[e->logical(h.voice)] / OG_WAVE] += 1u;
                for (uint32_t w = 0; w < W; ++w) rank[w] = w;
                std::stable_sort(rank.begin(), rank.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
                std::vector<uint32_t> place(W); // place[k] = position of the k-th heaviest group
                for (uint32_t k = 0; k < W; ++k) {
                    const uint32_t row = k / ROW, col = k % ROW;
                    const uint32_t row_len = std::min(ROW, W - row * ROW);
                    place[k] = row * ROW + ((row & 1u) ? (row_len - 1u - std::min(col, row_len - 1u)) : col);
                }
                std::vector<uint32_t> group_pos(W);
                for (uint32_t k = 0; k < W; ++k) group_pos[rank[k]] = place[k];
                // (a last, partial group stays last: it is the lightest or is forced there)
                const bool partial = (V % OG_WAVE) != 0u;
                if (partial && group_pos[W - 1] != W - 1) {
                    for (uint32_t w = 0; w < W; ++w)
                        if (group_pos[w] == W - 1) {
                            group_pos[w] = group_pos[W - 1];
                            break;
                        }
                    group_pos[W - 1] = W - 1;
                }
                std::vector<uint32_t> dealt(V);
                for (uint32_t r = 0; r < V; ++r) dealt[r] = group_pos[r / OG_WAVE] * OG_WAVE + r % OG_WAVE;
                for (uint32_t r = 0; r < V; ++r) new_phys[order[r]] = dealt[r];
            } else {
                for (uint32_t r = 0; r < V; ++r) new_phys[order[r]] = r;
            }
        }
        // the per-voice value inputs move with their voices
        std::vector<float> plane(V), moved(V);
        for (size_t i = 0; i < e->cg->inputs.size(); ++i) {
            const auto& in = e->cg->inputs[i];
            if (!(in.decl.kind == ogc::Kind::Value && in.decl.per_voice)) continue;
            float* d = reinterpret_cast<float*>(e->d_state) + (size_t)in.state_word * V;
            e->bounce.d2h(plane.data(), d, (size_t)V * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
            for (uint32_t v = 0; v < V; ++v) moved[new_phys[v]] = plane[e->phys(v)];
            e->bounce.h2d(d, moved.data(), (size_t)V * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
        }
        for (HostEvent& h : e->pending) h.voice = new_phys[e->logical(h.voice)];
        bool identity = true;
        for (uint32_t v = 0; v < V && identity; ++v) identity = new_phys[v] == v;
        if (identity) {
            e->phys_of.clear();
            e->logical_of.clear();
        } else {
            e->logical_of.assign(V, 0u);
            for (uint32_t v = 0; v < V; ++v) e->logical_of[new_phys[v]] = v;
            e->phys_of.swap(new_phys);
        }
        return OG_OK;
    });
}
int og_voice_slot(const og_engine* e, uint32_t voice, uint32_t* slot)
{
    if (!e || !slot) return set_err(OG_E_INVALID, "null argument");
    if (voice >= e->V) return set_err(OG_E_INVALID, "voice index out of range");
    *slot = e->phys(voice);
    return OG_OK;
}

int og_blocking_stats(const og_engine* e, uint64_t* calls, uint64_t* marker_timeouts)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    if (calls) *calls = e->blocking_waits;
    if (marker_timeouts) *marker_timeouts = e->blocking_timeouts;
    return OG_OK;
}

int og_enable_kernel_timing(og_engine* e, int on)
{
    if (!e) return set_err(OG_E_INVALID, "null engine");
    e->timing = on != 0;
    e->t_used = 0;
    e->t_blocks = 0;
    return OG_OK;
}

uint64_t og_kernel_blocks_timed(const og_engine* e) { return e ? (uint64_t)e->t_blocks : 0; }

double og_kernel_time_ms(og_engine* e, uint32_t* n_launches)
{
    if (!e || !e->timing) return -1.0;
    double total = 0.0;
    (void)hipSetDevice(e->device);
    (void)og_flush(e);
    (void)hipStreamSynchronize(e->stream);
    for (size_t i = 0; i < e->t_used; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e->t_start[i], e->t_stop[i]) == hipSuccess) total += ms;
    }
    if (n_launches) *n_launches = (uint32_t)e->t_used;
    const double avg = e->t_used ? total / (double)e->t_used : 0.0;
    // the shader clock of those launches: total cycles / total 100 MHz ticks between the marks of workgroup 0
    double cyc = 0.0, ticks = 0.0;
    for (size_t i = 0; e->h_clock && i < e->t_used; ++i) {
        const unsigned long long* q = e->h_clock + 4 * i;
        if (q[3] > q[1] && q[2] > q[0]) {
            cyc += (double)(q[2] - q[0]);
            ticks += (double)(q[3] - q[1]);
        }
    }
    e->last_clock_ghz = ticks > 0.0 ? cyc / ticks * 0.1 : 0.0;
    e->t_used = 0;
    return avg;
}

/* the shader clock the launches of the LAST og_kernel_time_ms call ran at, measured by the voice kernel itself (0 where the
 * device has no such counters) */
double og_kernel_clock_ghz(const og_engine* e) { return e ? e->last_clock_ghz : 0.0; }

int og_shader_clock_ghz(og_engine* e, double* ghz)
{
    if (!e || !ghz) return set_err(OG_E_INVALID, "null argument");
    return guard([&]() -> int {
        HIPCK(hipSetDevice(e->device));
        unsigned long long* h = nullptr;
        HIPCK(hipHostMalloc((void**)&h, 2 * sizeof(unsigned long long), hipHostMallocDefault));
        h[0] = h[1] = 0ull;
        hipLaunchKernelGGL(og_clock_probe, dim3(1), dim3(64), 0, e->stream, h, 2000u); // 20 us
        const hipError_t rc = hipStreamSynchronize(e->stream);
        const unsigned long long cyc = h[0], ref = h[1];
        (void)hipHostFree(h);
        if (rc != hipSuccess) throw HipError(std::string("og_clock_probe: ") + hipGetErrorString(rc));
        if (ref == 0ull) return set_err(OG_E_UNSUPPORTED, "no shader clock counter on this device");
        *ghz = (double)cyc / (double)ref * 0.1; // cycles per 10 ns tick
        return OG_OK;
    });
}

} // extern "C"

// ---- state snapshot ---------------------------------------------------------------------------------------------
// blob = DSP state (state planes, lane arrays, post-mix phase, delay lines) + control block: frame counter, the value
// and ValueRampState of every input, the ramp counter and every event that has not fired yet (resident timeline and
// queued pushes).  Loading it into an engine of the same graph / voice count / sample rate continues the render
// sample for sample.
namespace {
struct SnapHeader {
    uint32_t magic, version;
    uint64_t frame_now;
    uint32_t n_inputs, active_ramps;
    uint64_t n_events;
};
struct SnapRamp {
    float current, target, increment;
    uint32_t frames_remaining;
};
struct SnapEvent {
    uint32_t voice, target;
    uint64_t frame;
    float value;
    uint32_t block_local;
};
constexpr uint32_t SNAP_MAGIC = 0x3253474Fu; // "OGS2"

size_t dsp_bytes(const og_engine* e)
{
    return (e->cg->state.size() + e->cg->lane_state.size() * e->cg->lpv * e->cg->lane_width) * (size_t)e->V * 4 +
           (e->cg->bus_tremolo ? 4 : 0) + e->ring_bytes();
}
// Every event that has not fired by frame_now.  Blocks that are still queued (og_set_bus_batching) consume the events
// in [q_frame0, frame_now) when they are launched, and the snapshot header stores the post-queue frame_now: those
// events must not be saved (they would fire a second time after a load -- ADVICE r2), so the horizon is frame_now,
// not consumed_horizon().  og_save_state launches the queue first; og_state_bytes only counts.
void collect_unconsumed(const og_engine* e, std::vector<SnapEvent>& out)
{
    const uint64_t hz = e->frame_now;
    if (!e->seg_begin.empty()) {
        std::vector<OgEvent> old;
        for (uint32_t v = 0; v < e->V; ++v) {
            old.clear();
            e->old_events(v, old, hz); // the voice's segment, then its continuation
            for (const OgEvent& ev : old) out.push_back(SnapEvent{v, ev.target, ev.frame, ev.value, 0u});
        }
    }
    for (const HostEvent& h : e->pending)
        if (h.frame >= hz) out.push_back(SnapEvent{h.voice, h.target, h.frame, h.value, h.block_local ? 1u : 0u});
}
// (a blob of an engine with grouped voices -- og_group_voices -- is version 3: the state planes and the events are in
//  PHYSICAL order, and the logical -> physical table follows the events; loading it adopts the table)
size_t control_bytes(const og_engine* e, size_t n_events, bool grouped)
{
    return sizeof(SnapHeader) + e->cg->inputs.size() * (sizeof(float) + sizeof(SnapRamp)) + n_events * sizeof(SnapEvent) +
           (grouped ? (size_t)e->V * sizeof(uint32_t) : 0);
}
size_t control_bytes(const og_engine* e, size_t n_events) { return control_bytes(e, n_events, !e->phys_of.empty()); }
} // namespace

extern "C" {

size_t og_state_bytes(const og_engine* e)
{
    return ogabi::guard_value<size_t>((size_t)0, [&]() -> size_t {
    // (delay lines are part of the state: their size is known once og_init has sized them)
    if (!e) return 0;
    std::vector<SnapEvent> evs;
    collect_unconsumed(e, evs);
    return dsp_bytes(e) + control_bytes(e, evs.size());
    });
}

int og_save_state(og_engine* e, void* dst, size_t cap)
{
    if (!e || !dst) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus(); // queued blocks consume their events first: what is collected below is what frame_now has not reached
        std::vector<SnapEvent> evs;
        collect_unconsumed(e, evs);
        if (cap < dsp_bytes(e) + control_bytes(e, evs.size())) throw std::runtime_error("buffer too small");
        const size_t a = e->cg->state.size() * (size_t)e->V * 4, b = e->cg->lane_state.size() * (size_t)e->V * e->cg->lpv * e->cg->lane_width * 4;
        e->bounce.d2h(dst, e->d_state, a, e->stream);
        if (b) e->bounce.d2h((char*)dst + a, e->d_lane_state, b, e->stream);
        if (e->d_bus_phase) HIPCK(hipMemcpyAsync((char*)dst + a + b, e->d_bus_phase, 4, hipMemcpyDeviceToHost, e->stream));
        size_t off = a + b + (e->d_bus_phase ? 4 : 0);
        for (size_t k = 0; k < e->cg->rings.size(); ++k) {
            const size_t n = (size_t)e->ring_cap[k] * e->V * 4;
            if (n) e->bounce.d2h((char*)dst + off, e->d_ring[k], n, e->stream);
            off += n;
        }
        HIPCK(hipStreamSynchronize(e->stream));
        char* p = (char*)dst + off;
        const SnapHeader h{SNAP_MAGIC, e->phys_of.empty() ? 2u : 3u, e->frame_now, (uint32_t)e->cg->inputs.size(), e->active_ramps, (uint64_t)evs.size()};
        memcpy(p, &h, sizeof h);
        p += sizeof h;
        memcpy(p, e->values.data(), e->values.size() * sizeof(float));
        p += e->values.size() * sizeof(float);
        for (const Ramp& r : e->ramps) {
            const SnapRamp sr{r.current, r.target, r.increment, r.frames_remaining};
            memcpy(p, &sr, sizeof sr);
            p += sizeof sr;
        }
        if (!evs.empty()) memcpy(p, evs.data(), evs.size() * sizeof(SnapEvent));
        p += evs.size() * sizeof(SnapEvent);
        if (!e->phys_of.empty()) memcpy(p, e->phys_of.data(), (size_t)e->V * sizeof(uint32_t));
        return OG_OK;
    });
}

int og_load_state(og_engine* e, const void* src, size_t len)
{
    if (!e || !src) return set_err(OG_E_INVALID, "null argument");
    const size_t dsp = dsp_bytes(e);
    if (len < dsp + sizeof(SnapHeader)) return set_err(OG_E_INVALID, "state blob size mismatch");
    SnapHeader h;
    memcpy(&h, (const char*)src + dsp, sizeof h);
    // (n_events is checked against the bytes that are there BEFORE it enters any size arithmetic: a crafted count must
    //  not wrap control_bytes() around to a matching length)
    const bool grouped = h.version == 3u;
    const size_t fixed = control_bytes(e, 0, grouped);
    if (h.magic != SNAP_MAGIC || (h.version != 2u && h.version != 3u) || h.n_inputs != e->cg->inputs.size() || len < dsp + fixed ||
        h.n_events > (uint64_t)((len - dsp - fixed) / sizeof(SnapEvent)) || len != dsp + control_bytes(e, (size_t)h.n_events, grouped))
        return set_err(OG_E_INVALID, "state blob does not belong to this graph / voice count (or is from another version)");
    std::vector<uint32_t> new_phys, new_logical;
    if (grouped) { // the voice order must be a permutation before anything is changed
        new_phys.resize(e->V);
        new_logical.assign(e->V, 0xFFFFFFFFu);
        memcpy(new_phys.data(), (const char*)src + len - (size_t)e->V * sizeof(uint32_t), (size_t)e->V * sizeof(uint32_t));
        for (uint32_t v = 0; v < e->V; ++v) {
            if (new_phys[v] >= e->V || new_logical[new_phys[v]] != 0xFFFFFFFFu) return set_err(OG_E_INVALID, "state blob: the voice order is not a permutation");
            new_logical[new_phys[v]] = v;
        }
    }
    { // validate the events before anything is changed: the kernel indexes handlers / per-voice inputs by `target`
        const char* q = (const char*)src + dsp + control_bytes(e, 0, false);
        for (uint64_t i = 0; i < h.n_events; ++i, q += sizeof(SnapEvent)) {
            SnapEvent ev;
            memcpy(&ev, q, sizeof ev);
            bool ok = ev.voice < e->V;
            if (ev.target & OG_EV_SETVALUE) {
                const uint32_t in = ev.target & ~OG_EV_SETVALUE;
                ok = ok && in < e->cg->inputs.size() && e->cg->inputs[in].decl.kind == ogc::Kind::Value && e->cg->inputs[in].decl.per_voice;
            } else {
                ok = ok && (int)ev.target < e->cg->n_event_inputs;
            }
            if (!ok) return set_err(OG_E_INVALID, "state blob: event " + std::to_string(i) + " addresses a voice or input this graph does not have");
        }
    }
    return guard([&] {
        HIPCK(hipSetDevice(e->device));
        e->flush_bus();
        const size_t a = e->cg->state.size() * (size_t)e->V * 4, b = e->cg->lane_state.size() * (size_t)e->V * e->cg->lpv * e->cg->lane_width * 4;
        e->bounce.h2d(e->d_state, src, a, e->stream);
        if (b) e->bounce.h2d(e->d_lane_state, (const char*)src + a, b, e->stream);
        if (e->d_bus_phase) HIPCK(hipMemcpyAsync(e->d_bus_phase, (const char*)src + a + b, 4, hipMemcpyHostToDevice, e->stream));
        size_t off = a + b + (e->d_bus_phase ? 4 : 0);
        for (size_t k = 0; k < e->cg->rings.size(); ++k) {
            const size_t n = (size_t)e->ring_cap[k] * e->V * 4;
            if (n) e->bounce.h2d(e->d_ring[k], (const char*)src + off, n, e->stream);
            off += n;
        }
        e->reset_timeline();
        HIPCK(hipStreamSynchronize(e->stream));
        const bool order_changed = e->phys_of != new_phys;
        e->phys_of.swap(new_phys); // (version 2: identity)
        e->logical_of.swap(new_logical);
        if (order_changed && e->n_taps) { // taps were resolved to physical slots under the old order: resolve the SAME voices again
            std::vector<int32_t> slot(e->V, -1);
            for (uint32_t i = 0; i < e->n_taps; ++i) slot[e->phys(e->tap_voices[i])] = (int32_t)i;
            e->bounce.h2d(e->d_tap_slot, slot.data(), (size_t)e->V * 4, e->stream);
            HIPCK(hipStreamSynchronize(e->stream));
        }
        const char* p = (const char*)src + off + sizeof h;
        memcpy(e->values.data(), p, e->values.size() * sizeof(float));
        p += e->values.size() * sizeof(float);
        for (Ramp& r : e->ramps) {
            SnapRamp sr;
            memcpy(&sr, p, sizeof sr);
            p += sizeof sr;
            r.current = sr.current;
            r.target = sr.target;
            r.increment = sr.increment;
            r.frames_remaining = sr.frames_remaining;
        }
        e->active_ramps = h.active_ramps;
        e->frame_now = h.frame_now;
        for (uint64_t i = 0; i < h.n_events; ++i) {
            SnapEvent ev;
            memcpy(&ev, p, sizeof ev);
            p += sizeof ev;
            const bool local = ev.block_local != 0u;
            e->pending.push_back(HostEvent{ev.voice, std::max(ev.frame, h.frame_now), ev.target, ev.value, e->seq++, local});
            if (local) {
                e->n_block_local += 1;
                if (!(ev.target & OG_EV_SETVALUE)) { // the try_push capacity count of the block being assembled
                    const size_t ne = (size_t)std::max(1, e->cg->n_event_inputs);
                    if (e->local_cnt.empty()) e->local_cnt.assign((size_t)e->V * ne, 0);
                    const size_t k = (size_t)ev.voice * ne + ev.target;
                    if (e->local_cnt[k] == 0) e->local_touched.push_back((uint32_t)k);
                    if (e->local_cnt[k] < 255) e->local_cnt[k] += 1;
                }
            }
        }
        return OG_OK;
    });
}

} // extern "C"

#include "og_cluster.inl"
