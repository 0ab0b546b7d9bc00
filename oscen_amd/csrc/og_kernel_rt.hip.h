// og_kernel_rt.hip.h -- hand-written device runtime shared by every generated
// voice kernel: the block-argument ABI between the host engine and a kernel,
// SoA state access, per-voice event cursors, and the mix-bus reduction.
//
// Execution model (gfx950): one workgroup = one wave64, one lane = one voice.
// A launch covers ceil(n_voices/64) workgroups (1024 for 65 536 voices, i.e.
// one wave per SIMD on 256 CUs; 4 per SIMD at 262 144).  Per-voice state is
// loaded once from the [word][voice] planes (coalesced: lane i reads word w at
// address w*n_voices + v, 256 B per wave-instruction), lives in VGPRs for the
// whole block of <= 512 frames, and is stored once.
//
// Voices per wave (OgBlockArgs::lanes) is 64.  Narrower waves (32/16 voices, to
// put two waves on every SIMD at 65 536 voices) were measured and are slower:
// the path is bound by VALU issue and dependent-instruction latency (measured
// 3-6 cycles per wave-instruction per SIMD depending on how many waves a SIMD
// interleaves; a half-empty wave costs the same), so what counts is the
// number of wave-instructions.
//
// Mix bus (reference: `voices.audio_out -> audio_out` = sequential f32 sum in
// voice order, oscen-graph-compiler/src/codegen/emit_node.rs:463-466): each
// frame every lane drops its sample into an LDS tile [16 frames][64+1 lanes];
// every 16 frames the wave transposes: lane (q*16+j) adds voices q*16..q*16+15
// of frame j in voice order, two cross-lane adds combine the quarters, and 16
// lanes append the wave's partial for those frames.  The same 16-frame chunk
// is the unit of the frame loop: a chunk in which no lane of the wave has an
// event runs as one straight-line, unrolled body (no per-frame branches, so
// the scheduler can overlap frame f+1's envelopes with frame f's operator
// chain -- with one wave per SIMD at 65 536 voices there is no other wave to
// hide VALU latency behind).  One partial row per
// workgroup goes to HBM; og_bus_reduce sums the rows in workgroup order.  No
// atomics: the result is deterministic and differs from the reference's left
// fold only by this fixed re-association.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else // hiprtc pre-includes the HIP runtime; its fixed-width types live in __hip_internal
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

#define OG_WAVE 64
#define OG_MAX_BLOCK 512
#define OG_MAX_LAUNCH_BLOCKS 32                            // blocks one launch may cover (og_set_bus_batching)
#define OG_MAX_LAUNCH_FRAMES (OG_MAX_BLOCK * OG_MAX_LAUNCH_BLOCKS)
#define OG_MAX_SLOTS 160
#define OG_BUS_CHUNK 16
#define OG_LANE_DUMP_WORDS 16 // words per lane of OgBlockArgs::lane_dump (og_create allocates OG_WAVE of them; lane_plane_or_dump)
// the hand-off of the pipelined shapes: one workgroup barrier per chunk.  -DOG_EXPERIMENT_NOSYNC (scripts/build_variant.py
// <tag> -- -DOG_EXPERIMENT_NOSYNC; never a product build) removes it to time an upper bound: the results are then WRONG.
#ifdef OG_EXPERIMENT_NOSYNC
#define OG_HANDOFF_BARRIER() ((void)0)
#else
#define OG_HANDOFF_BARRIER() __syncthreads()
#endif

// ---- flag hand-off (pipelined shapes instantiated with FD_T >= 2) ---------------------------------------------------------
// Instead of one workgroup barrier per chunk -- every wave waits for the slowest of the four at every step -- each stage
// publishes its progress in an LDS word (`prog[stage]` = chunks completed) and waits only for what it needs: its producers
// to have completed the chunk it is about to read, its consumers to have left the ring slot it is about to overwrite
// (FD_T slots per crossing value: a producer may run FD_T chunks ahead).  LDS operations of one wave complete in issue
// order, so data written before the progress word is visible to whoever reads the word first and the data after it; the
// fences keep the COMPILER from moving accesses across (workgroup scope, LDS only: `s_waitcnt lgkmcnt(0)`, no vmcnt).
// `seen` caches the last value read (wave-uniform, an SGPR): a stage that is already known to be far enough ahead costs a
// scalar compare and no LDS read.  All four waves of a workgroup are resident together, so polling cannot starve a producer.
namespace og {
// returns whether the wave had to poll (the value was not there yet)
__device__ __forceinline__ bool handoff_wait(uint32_t* word, uint32_t& seen, const uint32_t need)
{
    if ((int)(seen - need) >= 0) return false;
    bool polled = false;
    for (;;) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(seen - need) >= 0) break;
        polled = true;
#if defined(OG_HOSTSIM) || (defined(OG_HANDOFF_SLEEP) && OG_HANDOFF_SLEEP > 0)
        __builtin_amdgcn_s_sleep(1); // (measured: a tight poll beats a sleeping one by 1-2 %; the host simulator yields here)
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return polled;
}
template <int P>
__device__ __forceinline__ void set_prio()
{
    __builtin_amdgcn_s_setprio(P);
}
__device__ __forceinline__ void handoff_publish(uint32_t* word, const uint32_t done)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(word, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
} // namespace og
#define OG_NO_EVENT 0xFFFFFFFFu
#define OG_EV_SETVALUE 0x80000000u

struct OgEvent {
    uint64_t frame;  // absolute frame since og_init
    uint32_t target; // event-input index, or OG_EV_SETVALUE | per-voice value index
    float value;     // scalar payload (gate velocity) or the new value
};

#define OG_MAX_RINGS 4

struct OgOutEvent { // one event of a graph event output (og_read_output_events)
    uint32_t voice;
    uint32_t output; // index among the graph's event outputs
    uint64_t frame;  // absolute frame since og_init
    float value;     // scalar payload
    uint32_t seq;    // append order (ties of one voice on one frame keep their push order)
};

struct OgBlockArgs {
    uint32_t n_voices;
    uint32_t frames;           // frames this launch renders: one block (<= 512), or several queued blocks back to back
    uint32_t ramp_stride;      // row stride of ramp_table in frames
    uint32_t lanes;            // active lanes per wave (64; experiment knob)
    uint32_t partial_plane;    // floats between the channel planes of `partials` (stereo voice output)
    uint32_t split;            // pipeline depth to launch: 0 = ordinary kernel, 2 / 4 = waves per 64 voices (og_graph.cpp)
    uint64_t frame0;
    uint32_t* state;           // [n_state_words][n_voices], raw 32-bit words
    uint32_t* lane_state;      // [n_lane_words][n_voices][LPV]: words of voices that span LPV lanes
    const OgEvent* events;     // sorted by (voice, frame, push order)
    const uint32_t* ev_end;    // [n_voices] one past the last event of the voice's current segment
    uint32_t* ev_cursor;       // [n_voices] next unconsumed event
    float* partials;           // [ceil(frames / 16)][n_workgroups][16]: per-workgroup partial sums of the mix bus
                               // (a Frame<2> voice output: two such planes, `partial_plane` floats apart)
    const float* ramp_table;   // [n_ramps + n_streams][ramp_stride] per-frame values of ramped / stream inputs
    float* taps;               // [n_taps][frames] per-voice output taps (or null)
    const int32_t* tap_slot;   // [n_voices] tap row or -1 (or null)
    // events that leave the voice (`output x: event;` fed by a node's #[output(event)] field): appended to a device log
    uint32_t* out_ev_count;    // events logged so far (may run past out_ev_cap: the excess is counted as dropped)
    struct OgOutEvent* out_ev; // [out_ev_cap]
    uint32_t out_ev_cap;
    uint32_t wide;             // split == 4: launch the 16-frame hand-off form (og_k4w_*) -- every workgroup of the bank is resident at once
    uint32_t* ev_lost;         // pushes an in-voice event queue could not hold (OG_NODE_EVENTS_PER_FRAME per frame and output)
    // timed launches only (og_enable_kernel_timing), else null: workgroup 0 writes {shader cycles, 100 MHz ticks} at its first
    // and last instruction -- the shader clock the launch ran at UNDER ITS OWN LOAD (og_kernel_clock_ghz)
    unsigned long long* clock_out; // [4]
    uint32_t* lane_dump;       // [OG_WAVE][OG_LANE_DUMP_WORDS] words behind the last lane-state plane: the write target of lanes beyond the last voice
    float* rings[OG_MAX_RINGS];        // delay lines: [capacity][n_voices] each (slot-major, voices contiguous)
    uint32_t ring_cap[OG_MAX_RINGS];   // capacity in samples (a power of two, ring_buffer/mod.rs:35-41)
    uint32_t slots[OG_MAX_SLOTS]; // block-uniform values (f32 or u32 bits)
};

namespace og {

__device__ __forceinline__ float slot_f(const OgBlockArgs& a, int i) { return __uint_as_float(a.slots[i]); }
__device__ __forceinline__ uint32_t slot_u(const OgBlockArgs& a, int i) { return a.slots[i]; }
// a slot pinned to an SGPR value (opaque to the optimizer, see og_nodes.hip.h)
__device__ __forceinline__ uint32_t scalar_u(const OgBlockArgs& a, int i)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)a.slots[i]);
}

// a value every lane computes alike (it derives from kernel-argument slots only), pinned to a scalar register: it then
// costs no VGPR for the length of the launch (the ordinary kernels sit at the 128-VGPR cap of four waves per SIMD)
__device__ __forceinline__ float uniform_f(const float x)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x)));
}

// compile-time flags passed to the generated tick lambdas: `value` = envelope stage-end checks on,
// `release` = envelope release arithmetic on (off in chunks where no lane of the wave is releasing),
// `pre` = the chunk's hand-off values were read from LDS into registers at the top of the chunk,
// `steady` = node-specific steady-state conditions hold for the whole chunk (Sect::fast_conds)
template <bool B, bool R = true, bool P = false, bool S = false>
struct BoolC {
    static constexpr bool value = B;
    static constexpr bool release = R;
    static constexpr bool pre = P;
    static constexpr bool steady = S;
};

// Frame<N> stream payload (oscen-lib/src/frame.rs): N f32 channels of one sample instant, with the arithmetic the
// reference's AudioFrame has.  Only indexed with constants by the generated code, so it lives in registers.
template <int N>
struct Frame {
    float v[N];
    __device__ __forceinline__ Frame operator+(const Frame& o) const
    {
        Frame r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i];
        return r;
    }
    __device__ __forceinline__ Frame operator-(const Frame& o) const
    {
        Frame r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i];
        return r;
    }
    __device__ __forceinline__ Frame operator*(float s) const
    {
        Frame r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = v[i] * s;
        return r;
    }
    __device__ __forceinline__ Frame operator-() const
    {
        Frame r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = -v[i];
        return r;
    }
    static __device__ __forceinline__ Frame splat(float s) // From<f32>: one sample on every channel
    {
        Frame r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = s;
        return r;
    }
};

// An event output of a node (`#[output(event)]`, EventOutput): the scalar events the node pushed on the current
// frame.  Lives in registers -- every access is a compile-time index or a select chain, never a dynamic one -- and is
// cleared at the end of every frame.  Capacity: 2 unless a node type of the graph asks for more (og_node_type::
// event_queue_capacity, up to the 32 of the reference's ArrayVec<EventInstance, 32>, graph/types.rs:18): the generator
// then defines OG_NODE_EVENTS_PER_FRAME in front of this header.
#ifndef OG_NODE_EVENTS_PER_FRAME
#define OG_NODE_EVENTS_PER_FRAME 2
#endif
struct EvOut {
    uint32_t n = 0u;
    uint32_t lost = 0u; // pushes past the capacity on this frame (reported through OgBlockArgs::ev_lost, og_events_dropped)
    float v[OG_NODE_EVENTS_PER_FRAME] = {};
    // EventInstance::frame_offset (graph/types.rs:129-132) as the producer set it: carried along the edge, multiplied / divided
    // by N across a rate boundary (codegen/emit_edge.rs:86-99), handed to a handler that names `frame_offset`.  Nobody
    // reading it, the compiler drops the array.
    uint32_t o[OG_NODE_EVENTS_PER_FRAME] = {};
    __device__ __forceinline__ void push_at(uint32_t frame_offset, float x) // try_push(EventInstance { frame_offset, Scalar(x) }): dropped when the frame's queue is full
    {
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)OG_NODE_EVENTS_PER_FRAME; ++k) {
            v[k] = (n == k) ? x : v[k];
            o[k] = (n == k) ? frame_offset : o[k];
        }
        lost += (n >= (uint32_t)OG_NODE_EVENTS_PER_FRAME) ? 1u : 0u;
        n = min(n + 1u, (uint32_t)OG_NODE_EVENTS_PER_FRAME);
    }
    __device__ __forceinline__ void push(float x) { push_at(0u, x); } // frame_offset 0: "now", what the reference's per-frame producers push
    __device__ __forceinline__ uint32_t off(uint32_t k) const
    {
        uint32_t r = o[0];
#pragma unroll
        for (uint32_t j = 1; j < (uint32_t)OG_NODE_EVENTS_PER_FRAME; ++j) r = (k == j) ? o[j] : r;
        return r;
    }
    __device__ __forceinline__ float get(uint32_t k) const
    {
        float r = v[0];
#pragma unroll
        for (uint32_t j = 1; j < (uint32_t)OG_NODE_EVENTS_PER_FRAME; ++j) r = (k == j) ? v[j] : r;
        return r;
    }
    __device__ __forceinline__ void clear() { n = 0u; lost = 0u; }
};

// s_memtime ticks once per shader cycle, s_memrealtime at a constant 100 MHz (MI355X_MICROARCH.md): the two, read at the first
// and the last instruction of workgroup 0, give the clock the launch really ran at.  One scalar branch per call.
__device__ __forceinline__ void clock_mark(const OgBlockArgs& a, int which)
{
#ifndef OG_HOSTSIM
    if (a.clock_out && blockIdx.x == 0 && threadIdx.x == 0) {
        a.clock_out[2 * which] = __builtin_readcyclecounter();
        a.clock_out[2 * which + 1] = __builtin_amdgcn_s_memrealtime();
    }
#else
    (void)a;
    (void)which;
#endif
}

// Ramp-table rows in the pipelined kernels: N consecutive words of a row from a UNIFORM address (one or two wide scalar
// loads per row and chunk instead of a scalar load behind a 64-bit address computation per input and frame) ...
template <uint32_t N>
__device__ __forceinline__ void row_fetch(float (&dst)[N], const float* __restrict__ src)
{
#ifndef OG_HOSTSIM
    // the table is written by the host before the launch and read-only for its whole duration: the CONSTANT address space,
    // from which a uniform load is a scalar load whatever stores the kernel has issued (through a plain global pointer the
    // compiler falls back to one vector load per lane once a store may alias)
    typedef __attribute__((address_space(4))) const float cfloat;
    const cfloat* s = (const cfloat*)(unsigned long long)src;
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) dst[j] = s[j];
#else
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) dst[j] = src[j];
#endif
}
// ... picked by the frame's position in the chunk where the chunk body is the unrolled one; read directly elsewhere
template <bool PRE, uint32_t N>
__device__ __forceinline__ float row_pick(const float (&pre)[N], uint32_t j, const OgBlockArgs& a, int row, uint32_t f)
{
    if constexpr (PRE) return pre[j];
    else return a.ramp_table[(size_t)row * a.ramp_stride + f];
}

// u32::saturating_mul (the reference's outer -> inner rescale of an event's frame_offset)
__device__ __forceinline__ uint32_t sat_mul_u32(uint32_t a, uint32_t b)
{
    const unsigned long long p = (unsigned long long)a * (unsigned long long)b;
    return p > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)p;
}

struct VoiceCtx {
    uint32_t v;     // voice index
    bool valid;     // v < n_voices
    bool lead;      // first lane of the voice (the only one that reports / stores per-voice things)
    uint32_t h;     // lane within the voice: 0 for ordinary graphs; LPV > 1: owns array elements h * OG_HPL .. + OG_HPL - 1
    uint32_t lane;
    // events
    uint32_t ev_cur, ev_cur0, ev_end;
    uint32_t next_ev; // frame offset inside this block of the next event (record ev_cur), or OG_NO_EVENT
    // Round 4: the next record's payload and the frame of the record after it are PREFETCHED.  Firing an event used to
    // cost two dependent global loads in the middle of a chunk (the record, then the next record's frame for the
    // "another one on this frame?" test) -- ~3 us per event and workgroup with only two to four waves per SIMD to hide
    // them, far more than the handler itself.  Now the handler reads registers, ev_advance() takes next_ev from
    // `next2` at once and issues the loads for the record after next without waiting for them.
    uint32_t nx_target; // target of record ev_cur
    float nx_value;     // value of record ev_cur
    uint64_t n2_frame;  // absolute frame of record ev_cur + 1 as loaded (~0 = none): turned into a frame offset only when
                        // that record becomes the next one, so that nothing waits for the load here
    // taps
    int32_t tap;
};

__device__ __forceinline__ uint32_t ev_rel_of(const OgBlockArgs& a, uint64_t fr)
{
    const uint64_t rel = (fr > a.frame0) ? (fr - a.frame0) : 0ull; // late events fire on frame 0
    return (rel < (uint64_t)a.frames) ? (uint32_t)rel : OG_NO_EVENT;
}
__device__ __forceinline__ uint32_t ev_rel_frame(const OgBlockArgs& a, uint32_t idx) { return ev_rel_of(a, a.events[idx].frame); }

// load record ev_cur (frame, target, value) and the frame of the record behind it.
// PRE = false (the ordinary one-wave kernel): only the frame is read here and the record itself when it fires -- that
// kernel runs with 4 to 16 waves per SIMD, which hide the two loads, and sits at the 128-VGPR cap of four waves per SIMD:
// the four registers of the prefetch cost it 19 spills around its chunk loop (profiles/r04h_fm262144_summary.md).
template <bool PRE = true>
__device__ __forceinline__ void ev_arm(const OgBlockArgs& a, VoiceCtx& c)
{
    c.next_ev = OG_NO_EVENT;
    c.n2_frame = ~0ull;
    c.nx_target = 0u;
    c.nx_value = 0.0f;
    if (c.ev_cur < c.ev_end) {
        if (PRE) {
            const OgEvent ev = a.events[c.ev_cur];
            c.next_ev = ev_rel_of(a, ev.frame);
            c.nx_target = ev.target;
            c.nx_value = ev.value;
            if (c.ev_cur + 1u < c.ev_end) c.n2_frame = a.events[c.ev_cur + 1u].frame;
        } else {
            c.next_ev = ev_rel_frame(a, c.ev_cur);
        }
    }
}
// the record that fires now: {target, value}
struct EvRec {
    uint32_t target;
    float value;
};
template <bool PRE = true>
__device__ __forceinline__ EvRec ev_record(const OgBlockArgs& a, const VoiceCtx& c)
{
    if (PRE) return EvRec{c.nx_target, c.nx_value};
    const OgEvent ev = a.events[c.ev_cur];
    return EvRec{ev.target, ev.value};
}

// LPV = lanes per voice: 1 for ordinary graphs (64 voices per wave).  Graphs whose nodes carry
// per-harmonic arrays (the electric-piano voice: 32 partials, ~260 state words) spread one voice
// over LPV = 8 lanes instead of 260 VGPRs: lane h owns OG_HPL = 4 harmonics, per-voice scalars are
// replicated on the voice's lanes (so the per-voice bookkeeping is paid once per four harmonics).
template <bool TAPS, int LPV = 1>
__device__ __forceinline__ void voice_begin(const OgBlockArgs& a, VoiceCtx& c)
{
    c.lane = threadIdx.x;
    const uint32_t gl = blockIdx.x * a.lanes + threadIdx.x;
    c.v = gl / LPV;
    c.h = gl % LPV;
    c.lead = c.h == 0;
    c.valid = (threadIdx.x < a.lanes) && (c.v < a.n_voices);
    c.ev_cur = c.ev_end = 0;
    c.tap = -1;
    if (c.valid) {
        c.ev_cur = a.ev_cursor[c.v];
        c.ev_end = a.ev_end[c.v];
        if (TAPS) c.tap = a.tap_slot[c.v];
    }
    ev_arm<false>(a, c);
    c.ev_cur0 = c.ev_cur;
    clock_mark(a, 0);
}

// two-wave pipeline kernel: both waves of the workgroup see the same 64 voices
template <bool TAPS>
__device__ __forceinline__ void voice_begin_split(const OgBlockArgs& a, VoiceCtx& c)
{
    c.lane = threadIdx.x % OG_WAVE;
    c.v = blockIdx.x * OG_WAVE + c.lane;
    c.h = 0;
    c.lead = true;
    c.valid = c.v < a.n_voices;
    c.ev_cur = c.ev_end = 0;
    c.tap = -1;
    if (c.valid) {
        c.ev_cur = a.ev_cursor[c.v];
        c.ev_end = a.ev_end[c.v];
        if (TAPS) c.tap = a.tap_slot[c.v];
    }
    ev_arm(a, c);
    c.ev_cur0 = c.ev_cur;
    clock_mark(a, 0);
}

__device__ __forceinline__ void voice_end(const OgBlockArgs& a, const VoiceCtx& c)
{
    if (c.valid && c.lead && c.ev_cur != c.ev_cur0) a.ev_cursor[c.v] = c.ev_cur;
    clock_mark(a, 1);
}

// end of a frame, before the queues are cleared: count the pushes an event output had to drop ...
__device__ __forceinline__ void ev_report_lost(const OgBlockArgs& a, const VoiceCtx& c, const EvOut& q)
{
    if (c.valid && c.lead && q.lost != 0u && a.ev_lost) atomicAdd(a.ev_lost, q.lost);
}
// ... and hand the events of a GRAPH event output (`node.trig -> x` with `output x: event`) to the host log
__device__ __forceinline__ void ev_out_log(const OgBlockArgs& a, const VoiceCtx& c, uint32_t output, uint32_t f, const EvOut& q)
{
    if (!(c.valid && c.lead) || !a.out_ev_count) return;
#if OG_NODE_EVENTS_PER_FRAME <= 4
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)OG_NODE_EVENTS_PER_FRAME; ++k)
        if (k < q.n) {
#else
    for (uint32_t k = 0; k < q.n; ++k) {
        {
#endif
            const uint32_t idx = atomicAdd(a.out_ev_count, 1u);
            if (idx < a.out_ev_cap) a.out_ev[idx] = OgOutEvent{c.v, output, a.frame0 + (uint64_t)f, q.get(k), idx};
        }
#if OG_NODE_EVENTS_PER_FRAME > 4
    }
#endif
}

// pop the current event: the next one's frame is already known (next2); its payload and the frame of the one behind it
// are requested here and only waited for when they are used
template <bool PRE = true>
__device__ __forceinline__ void ev_advance(const OgBlockArgs& a, VoiceCtx& c)
{
    c.ev_cur += 1;
    if (!PRE) {
        c.next_ev = (c.ev_cur < c.ev_end) ? ev_rel_frame(a, c.ev_cur) : OG_NO_EVENT;
        return;
    }
    c.next_ev = ev_rel_of(a, c.n2_frame); // (~0: far beyond the launch -> OG_NO_EVENT)
    c.n2_frame = ~0ull;
    if (c.ev_cur < c.ev_end) {
        const OgEvent ev = a.events[c.ev_cur];
        c.nx_target = ev.target;
        c.nx_value = ev.value;
        if (c.ev_cur + 1u < c.ev_end) c.n2_frame = a.events[c.ev_cur + 1u].frame;
    }
}

__device__ __forceinline__ float ld_f(const OgBlockArgs& a, const VoiceCtx& c, int w)
{
    return __uint_as_float(a.state[(size_t)w * a.n_voices + c.v]);
}
__device__ __forceinline__ uint32_t ld_u(const OgBlockArgs& a, const VoiceCtx& c, int w)
{
    return a.state[(size_t)w * a.n_voices + c.v];
}
__device__ __forceinline__ void st_f(const OgBlockArgs& a, const VoiceCtx& c, int w, float x)
{
    a.state[(size_t)w * a.n_voices + c.v] = __float_as_uint(x);
}
__device__ __forceinline__ void st_u(const OgBlockArgs& a, const VoiceCtx& c, int w, uint32_t x)
{
    a.state[(size_t)w * a.n_voices + c.v] = x;
}

// Array-valued state of an LPV > 1 voice (`[f32; 32]` fields: the electric piano's per-harmonic arrays): a lane
// owns OG_HPL consecutive elements, h = c.h * OG_HPL + j, kept in registers as one HarmV and moved with 16-byte
// accesses -- the 64 lanes of a wave read consecutive bytes per array.
// The elements sit in 2-vectors so that the per-element arithmetic (identical and independent across elements) runs
// on packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32: two IEEE f32 operations per issue, each rounded exactly
// like its scalar form).
#ifndef OG_HPL
#define OG_HPL 8 // harmonics per lane (2, 4 or 8: 16, 8 or 4 lanes per 32-harmonic voice); a generated source pre-defines it
#endif
#define OG_HPAIRS (OG_HPL / 2)
typedef float og_f2 __attribute__((ext_vector_type(2)));
struct HarmV {
    og_f2 p[OG_HPAIRS]; // elements (0,1), (2,3), ...
};
// (a value select: `c ? x : y` on two HarmV lvalues would select ADDRESSES and pin both to scratch memory)
// (by VALUE, element by element through scalars: a select between two loads is turned into a load from a selected
//  address, which takes the addresses of both operands and moves whole node structs into scratch -- seen again in
//  round 3 when the elements became an array: 96-byte EpAmp stores/loads around every chunk)
__device__ __forceinline__ og_f2 f2_select(bool c, og_f2 x, og_f2 y)
{
    const float x0 = x.x, x1 = x.y, y0 = y.x, y1 = y.y;
    return og_f2{c ? x0 : y0, c ? x1 : y1};
}
__device__ __forceinline__ HarmV harm_select(bool c, HarmV x, HarmV y)
{
    HarmV r;
#if OG_HPAIRS >= 1
    r.p[0] = f2_select(c, x.p[0], y.p[0]);
#endif
#if OG_HPAIRS >= 2
    r.p[1] = f2_select(c, x.p[1], y.p[1]);
#endif
#if OG_HPAIRS >= 4
    r.p[2] = f2_select(c, x.p[2], y.p[2]);
    r.p[3] = f2_select(c, x.p[3], y.p[3]);
#endif
    return r;
}
// element-wise helpers; OG_EP_SCALAR (experiment switch) forces one scalar instruction per element
#ifdef OG_EP_SCALAR
__device__ __forceinline__ float og_opaque(float x)
{
    asm volatile("" : "+v"(x)); // keeps the two halves of a pair apart so that the back end cannot re-pack them
    return x;
}
__device__ __forceinline__ og_f2 f2_mul(og_f2 x, og_f2 y) { return og_f2{og_opaque(x.x * y.x), og_opaque(x.y * y.y)}; }
__device__ __forceinline__ og_f2 f2_add(og_f2 x, og_f2 y) { return og_f2{og_opaque(x.x + y.x), og_opaque(x.y + y.y)}; }
__device__ __forceinline__ og_f2 f2_sub(og_f2 x, og_f2 y) { return og_f2{og_opaque(x.x - y.x), og_opaque(x.y - y.y)}; }
#else
__device__ __forceinline__ og_f2 f2_mul(og_f2 x, og_f2 y) { return x * y; }
__device__ __forceinline__ og_f2 f2_add(og_f2 x, og_f2 y) { return x + y; }
__device__ __forceinline__ og_f2 f2_sub(og_f2 x, og_f2 y) { return x - y; }
#endif
__device__ __forceinline__ og_f2 f2_mul(og_f2 x, float y) { return f2_mul(x, og_f2{y, y}); }
// x * y + z: tolerance mode fuses (v_pk_fma_f32: one issue for two harmonics, one rounding each); OG_STRICT keeps the
// reference's two roundings.  Only used where the result does not feed a recurrence that preserves magnitude (the
// amplitude interpolation, which is re-anchored on its target every 65 frames, and the output sum -- never the rotation).
__device__ __forceinline__ og_f2 f2_fma(og_f2 x, og_f2 y, og_f2 z)
{
#if defined(OG_STRICT) || defined(OG_EP_SCALAR)
    return f2_add(f2_mul(x, y), z);
#else
    return __builtin_elementwise_fma(x, y, z);
#endif
}
__device__ __forceinline__ og_f2 f2_fma(og_f2 x, float y, og_f2 z) { return f2_fma(x, og_f2{y, y}, z); }
__device__ __forceinline__ HarmV harm_splat(float x)
{
    HarmV r;
#pragma unroll
    for (int i = 0; i < OG_HPAIRS; ++i) r.p[i] = og_f2{x, x};
    return r;
}
__device__ __forceinline__ HarmV harm_make(const float (&t)[OG_HPL])
{
    HarmV r;
#pragma unroll
    for (int i = 0; i < OG_HPAIRS; ++i) r.p[i] = og_f2{t[2 * i], t[2 * i + 1]};
    return r;
}
template <int LPV>
__device__ __forceinline__ HarmV ldl_h(const OgBlockArgs& a, const VoiceCtx& c, int k)
{
    const float* src = reinterpret_cast<const float*>(a.lane_state) + (((size_t)k * a.n_voices + c.v) * LPV + c.h) * OG_HPL;
    HarmV r;
#if OG_HPL >= 4
#pragma unroll
    for (int i = 0; i < OG_HPL / 4; ++i) { // 16-byte accesses: the lanes of a wave read consecutive bytes
        const float4 q = reinterpret_cast<const float4*>(src)[i];
        r.p[2 * i] = og_f2{q.x, q.y};
        r.p[2 * i + 1] = og_f2{q.z, q.w};
    }
#else
    const float2 q = *reinterpret_cast<const float2*>(src);
    r.p[0] = og_f2{q.x, q.y};
#endif
    return r;
}
// this lane's OG_HPL words of lane-state plane k (arrays that stay in memory: read and written by event handlers only)
template <int LPV>
__device__ __forceinline__ float* lane_plane(const OgBlockArgs& a, const VoiceCtx& c, int k)
{
    return reinterpret_cast<float*>(a.lane_state) + (((size_t)k * a.n_voices + c.v) * LPV + c.h) * OG_HPL;
}
// ... or, for a lane beyond the bank's last voice, its slot of the dump area: every lane runs the tick, a node-to-node event
// can reach the handler of such a lane, and what it writes must not land in another voice's planes.  A select, not a
// branch: a guard around the handler's stores cost the e-piano kernel its scratch-free register allocation.
template <int LPV>
__device__ __forceinline__ float* lane_plane_or_dump(const OgBlockArgs& a, const VoiceCtx& c, int k)
{
    float* p = lane_plane<LPV>(a, c, k);
    static_assert(OG_HPL <= OG_LANE_DUMP_WORDS, "a dump slot holds one lane's HarmV");
    float* d = reinterpret_cast<float*>(a.lane_dump) + c.lane * OG_LANE_DUMP_WORDS; // (c.lane < OG_WAVE: one slot per lane)
    return c.valid ? p : d;
}
template <int LPV>
__device__ __forceinline__ void stl_h(const OgBlockArgs& a, const VoiceCtx& c, int k, const HarmV& x)
{
    float* dst = reinterpret_cast<float*>(a.lane_state) + (((size_t)k * a.n_voices + c.v) * LPV + c.h) * OG_HPL;
#if OG_HPL >= 4
#pragma unroll
    for (int i = 0; i < OG_HPL / 4; ++i)
        reinterpret_cast<float4*>(dst)[i] = make_float4(x.p[2 * i].x, x.p[2 * i].y, x.p[2 * i + 1].x, x.p[2 * i + 1].y);
#else
    *reinterpret_cast<float2*>(dst) = make_float2(x.p[0].x, x.p[0].y);
#endif
}

// All LDS traffic of the mix bus stays inside one wave (DS operations of a wave execute in order),
// so a wavefront-scope fence is the whole synchronisation; a workgroup barrier would also stall on
// the other wave of the two-wave pipeline kernel.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- mix bus ---------------------------------------------------------------
struct BusLds {
    float tile[OG_BUS_CHUNK][OG_WAVE + 1]; // +1 pad: conflict-free transposed read
};

// A lane that contributes nothing to the bus (beyond the last voice; not the lead lane of a multi-lane voice) is not masked
// to 0.0 on every frame (one v_cndmask per frame): it writes into the pad column, which the transposed read never visits,
// and its own column holds the zeros bus_init() put there.  The column is loop-invariant, so the store of a frame is one
// ds_write_b32 with an immediate offset.
// ALL_LANES: every lane of a multi-lane voice holds a share of the voice's output (see og::ep_bank_tick)
template <bool ALL_LANES>
__device__ __forceinline__ uint32_t bus_col(const VoiceCtx& c)
{
    return (c.valid && (ALL_LANES || c.lead)) ? c.lane : (uint32_t)OG_WAVE;
}
// once per launch, by the wave that owns the tile, before its first bus_put
__device__ __forceinline__ void bus_init(const VoiceCtx& c, BusLds& lds)
{
#pragma unroll
    for (int j = 0; j < OG_BUS_CHUNK; ++j) lds.tile[j][c.lane] = 0.0f;
}

// one sample of one voice into the wave's transpose tile (row j = frame within the chunk)
template <bool TAPS, bool ALL_LANES = false>
__device__ __forceinline__ void bus_put(const OgBlockArgs& a, const VoiceCtx& c, BusLds& lds, uint32_t f, uint32_t j,
                                        float out)
{
    lds.tile[j][bus_col<ALL_LANES>(c)] = out;
    if (TAPS) {
        if (c.tap >= 0 && c.lead) a.taps[(size_t)c.tap * a.frames + f] = out;
    }
}

// after n <= OG_BUS_CHUNK frames starting at `base`: transpose-sum the tile; 16 lanes append the wave's partial sums
// for those frames to its row in HBM (64 bytes per chunk; the row is complete when the launch ends, whatever its length)
__device__ __forceinline__ void bus_chunk_reduce(const OgBlockArgs& a, const VoiceCtx& c, BusLds& lds, uint32_t base,
                                                 uint32_t n)
{
    wave_sync(); // orders the wave's LDS writes before its transposed reads
    const uint32_t j = c.lane & (OG_BUS_CHUNK - 1);
    const uint32_t q = c.lane / OG_BUS_CHUNK;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < OG_WAVE / 4; ++i) s += lds.tile[j][q * (OG_WAVE / 4) + i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    // chunk-major rows: [chunk][workgroup][OG_BUS_CHUNK].  All waves of the bank are at about the same chunk, so what is
    // written (and what one workgroup of og_bus_reduce reads) at any moment is one compact region instead of one cache
    // line in each of thousands of rows a whole launch apart (row-major rows of 32 blocks: -25 % on sat4x_voice).
    if (c.lane < OG_BUS_CHUNK && j < n)
        a.partials[((size_t)(base / OG_BUS_CHUNK) * gridDim.x + blockIdx.x) * OG_BUS_CHUNK + j] = s;
    wave_sync();
}

__device__ __forceinline__ void bus_flush(const OgBlockArgs&, const VoiceCtx&, BusLds&) {} // (rows are written as they are formed)

// ---- several bus channels: a Frame<2> voice output (a per-voice pan) and / or several stream outputs of the voice
// graph (`output out_a: stream; output out_b: stream; ...`: every one of them is summed over the voices like the
// reference sums `voices.<out>`).  N tiles, N planes of partial rows, taps [tap][frame][N] -- same calls, picked by
// overload.
#define OG_MAX_BUS_CHANNELS 4
template <int N>
struct OutN {
    float v[N];
};
template <int N>
struct BusLdsN {
    BusLds ch[N];
};
using Out2 = OutN<2>;
using BusLds2 = BusLdsN<2>;
template <bool TAPS, bool ALL_LANES = false, int N = 2>
__device__ __forceinline__ void bus_put(const OgBlockArgs& a, const VoiceCtx& c, BusLdsN<N>& lds, uint32_t f, uint32_t j, OutN<N> out)
{
    const uint32_t col = bus_col<ALL_LANES>(c);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        lds.ch[k].tile[j][col] = out.v[k];
        if (TAPS) {
            if (c.tap >= 0 && c.lead) a.taps[((size_t)c.tap * a.frames + f) * N + k] = out.v[k];
        }
    }
}
template <int N>
__device__ __forceinline__ void bus_init(const VoiceCtx& c, BusLdsN<N>& lds)
{
#pragma unroll
    for (int k = 0; k < N; ++k) bus_init(c, lds.ch[k]);
}
template <int N>
__device__ __forceinline__ void bus_chunk_reduce(const OgBlockArgs& a, const VoiceCtx& c, BusLdsN<N>& lds, uint32_t base, uint32_t n)
{
    wave_sync();
    const uint32_t j = c.lane & (OG_BUS_CHUNK - 1);
    const uint32_t q = c.lane / OG_BUS_CHUNK;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < OG_WAVE / 4; ++i) s += lds.ch[k].tile[j][q * (OG_WAVE / 4) + i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (c.lane < OG_BUS_CHUNK && j < n)
            a.partials[(size_t)k * a.partial_plane + ((size_t)(base / OG_BUS_CHUNK) * gridDim.x + blockIdx.x) * OG_BUS_CHUNK + j] = s;
    }
    wave_sync();
}
template <int N>
__device__ __forceinline__ void bus_flush(const OgBlockArgs&, const VoiceCtx&, BusLdsN<N>&) {}

} // namespace og

