// og_cluster.inl -- multi-GPU voice banks behind the C ABI (textually included at the end of og_engine.cpp:
// it works on og_engine's internals).
//
// SURVEY 8(e) / examples/fm-synth/src/lib.rs:269-274: voices are independent, the only cross-voice operation is
// the sum onto the mix bus.  A cluster owns one engine ("shard") per entry of device_ids; shard s holds the
// contiguous global voices [s*V/n, (s+1)*V/n) (note streams and every per-voice call are keyed by GLOBAL voice
// id), broadcast values are replicated, and the data path has exactly one exchange step: the per-device mono
// buses of a whole BATCH of blocks are summed with one ncclReduce(sum, f32, root = first device) over xGMI (a
// per-block 1 KB reduce would be latency-bound).  Shards that share a device are added on that device first,
// so the communicator has one rank per distinct GPU.  A post-mix node (the e-piano Tremolo, a17) runs once, on
// the root, after the reduce -- like the reference runs it after the voice sum.  One host thread per shard
// issues that shard's launches (a single thread issuing 8 x 2 launches per block would be the bottleneck: a block
// is ~66 us of device time); the reduce of batch k runs on a side stream while batch k+1's kernels are issued.
//
// RCCL is bound at run time (dlopen) the first time a cluster spans more than one device: a single-GPU user of
// liboscen_gpu.so does not need it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

__global__ void og_bus_accumulate(float* __restrict__ dst, const float* __restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    void load()
    {
        if (lib) return;
        // OSCEN_GPU_RCCL_LIB = path of the RCCL build to bind (an explicit path is loaded even when another library of
        // the same soname is already in the process -- e.g. the one a host application links)
        if (const char* forced = getenv("OSCEN_GPU_RCCL_LIB")) {
            lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!lib) throw HipError(std::string("cannot load OSCEN_GPU_RCCL_LIB=") + forced + ": " + dlerror());
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib) throw HipError(std::string("cannot load RCCL (librccl.so.1): ") + dlerror());
        auto sym = [&](const char* s) {
            void* p = dlsym(lib, s);
            if (!p) throw HipError(std::string("RCCL symbol missing: ") + s);
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
    void ck(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) throw HipError(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
    }
};
Rccl& rccl()
{
    static Rccl r;
    return r;
}

// one persistent host thread per shard
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
    std::string err;
    bool device_err = false;
    Worker()
    {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return quit || busy; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                lk.unlock();
                std::string e;
                bool dev = false;
                try {
                    j();
                } catch (const HipError& ex) {
                    e = ex.what();
                    dev = true;
                } catch (const std::exception& ex) {
                    e = ex.what();
                }
                lk.lock();
                err = e;
                device_err = dev;
                busy = false;
                cv.notify_all();
            }
        });
    }
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
    void run(std::function<void()> j)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return !busy; }); // (never overwrite a job that is still running: its error would be lost)
        job = std::move(j);
        busy = true;
        err.clear();
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return !busy; });
        if (!err.empty()) {
            if (device_err) throw HipError(err);
            throw std::runtime_error(err);
        }
    }
};

constexpr uint32_t CL_BATCH_BLOCKS = 256; // blocks per reduce

} // namespace

struct og_cluster {
    std::vector<og_engine*> shard;
    bool time_reduces = false; // og_cluster_enable_reduce_timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> reduce_events; // recorded, not yet read (bounded: fold_reduce_events)
    double reduce_ms_folded = 0.0; // device time of the pairs already read and destroyed
    uint64_t reduce_n_folded = 0;
    // reads and destroys the event pairs (all of them, or only the completed ones at the front)
    void fold_reduce_events(bool wait)
    {
        size_t k = 0;
        for (; k < reduce_events.size(); ++k) {
            auto& pr = reduce_events[k];
            if (wait) HIPCK(hipEventSynchronize(pr.second));
            else if (hipEventQuery(pr.second) != hipSuccess) break;
            float ms = 0.0f;
            HIPCK(hipEventElapsedTime(&ms, pr.first, pr.second));
            reduce_ms_folded += ms;
            reduce_n_folded += 1;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        reduce_events.erase(reduce_events.begin(), reduce_events.begin() + (ptrdiff_t)k);
    }
    void drop_reduce_events()
    {
        for (auto& pr : reduce_events) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        reduce_events.clear();
        reduce_ms_folded = 0.0;
        reduce_n_folded = 0;
    }
    std::vector<og_out_event> out_ev_carry, out_ev_scratch; // og_cluster_read_output_events: queue (global voice ids) / per-shard scratch
    std::vector<uint64_t> lo; // first global voice of every shard; lo[n] = total
    std::vector<int> devs;    // distinct devices, devs[0] = root
    std::vector<int> dev_of_shard;
    std::vector<std::unique_ptr<Worker>> workers;
    uint64_t total = 0;
    uint32_t channels = 1;
    uint32_t vch = 1; // channels of the voice sum the shards hand over (og_voice_channels)
    bool tremolo = false;
    bool inited = false;
    // per shard: two batch buffers of mono partial buses + "buffer filled" events
    size_t cap_frames = 0;
    std::vector<float*> sh_buf[2];
    std::vector<hipEvent_t> sh_done[2];
    // per device: side stream + "reduce finished with buffer b" events
    std::vector<hipStream_t> dev_stream;
    std::vector<hipEvent_t> dev_free[2];
    std::vector<float*> dev_acc[2]; // the buffer of the device's first shard (accumulated in place)
    std::vector<ncclComm_t> comms;
    bool use_rccl = false;
    uint64_t n_reduces = 0;
    // root: post-mix stage and host hand-over
    float* d_out = nullptr; // [cap_frames * channels]
    float* d_phase = nullptr;
    float* h_pin[2] = {nullptr, nullptr}; // pinned staging of a batch
    hipEvent_t host_ready[2] = {nullptr, nullptr};

    ~og_cluster()
    {
        workers.clear();
        for (og_engine* e : shard)
            if (e) {
                (void)hipSetDevice(e->device);
                (void)hipStreamSynchronize(e->stream);
            }
        drop_reduce_events();
        for (size_t d = 0; d < devs.size(); ++d) {
            (void)hipSetDevice(devs[d]);
            if (d < dev_stream.size() && dev_stream[d]) {
                (void)hipStreamSynchronize(dev_stream[d]);
                (void)hipStreamDestroy(dev_stream[d]);
            }
            for (int b = 0; b < 2; ++b)
                if (d < dev_free[b].size() && dev_free[b][d]) (void)hipEventDestroy(dev_free[b][d]);
        }
        for (ncclComm_t c : comms)
            if (c) (void)rccl().CommDestroy(c);
        for (size_t s = 0; s < shard.size(); ++s) {
            if (shard[s]) (void)hipSetDevice(shard[s]->device);
            for (int b = 0; b < 2; ++b) {
                if (s < sh_buf[b].size()) (void)hipFree(sh_buf[b][s]);
                if (s < sh_done[b].size() && sh_done[b][s]) (void)hipEventDestroy(sh_done[b][s]);
            }
        }
        if (!devs.empty()) (void)hipSetDevice(devs[0]);
        (void)hipFree(d_out);
        (void)hipFree(d_phase);
        for (int b = 0; b < 2; ++b) {
            if (h_pin[b]) (void)hipHostFree(h_pin[b]);
            if (host_ready[b]) (void)hipEventDestroy(host_ready[b]);
        }
        for (og_engine* e : shard) delete e;
    }

    uint32_t shard_of(uint64_t voice) const
    {
        uint32_t s = (uint32_t)(std::upper_bound(lo.begin(), lo.end(), voice) - lo.begin()) - 1u;
        return s;
    }

    void ensure_buffers(size_t frames)
    {
        if (frames <= cap_frames) return;
        sync_all();
        for (size_t s = 0; s < shard.size(); ++s) {
            HIPCK(hipSetDevice(shard[s]->device));
            for (int b = 0; b < 2; ++b) {
                if (sh_buf[b][s]) HIPCK(hipFree(sh_buf[b][s]));
                sh_buf[b][s] = nullptr;
                HIPCK(hipMalloc(&sh_buf[b][s], frames * vch * sizeof(float)));
            }
        }
        HIPCK(hipSetDevice(devs[0]));
        if (d_out) HIPCK(hipFree(d_out));
        d_out = nullptr;
        HIPCK(hipMalloc(&d_out, frames * channels * sizeof(float)));
        for (int b = 0; b < 2; ++b) {
            if (h_pin[b]) HIPCK(hipHostFree(h_pin[b]));
            h_pin[b] = nullptr;
            HIPCK(hipHostMalloc((void**)&h_pin[b], frames * channels * sizeof(float), hipHostMallocDefault));
            if (!host_ready[b]) HIPCK(hipEventCreateWithFlags(&host_ready[b], hipEventDisableTiming));
        }
        cap_frames = frames;
    }

    // every shard thread has finished its job; the FIRST error is rethrown only after all of them are idle (a thread
    // that is still issuing launches holds raw engine pointers and events)
    void wait_all()
    {
        std::string first;
        bool dev = false;
        for (auto& w : workers) {
            try {
                w->wait();
            } catch (const HipError& ex) {
                if (first.empty()) { first = ex.what(); dev = true; }
            } catch (const std::exception& ex) {
                if (first.empty()) first = ex.what();
            }
        }
        if (!first.empty()) {
            if (dev) throw HipError(first);
            throw std::runtime_error(first);
        }
    }

    void sync_all()
    {
        for (og_engine* e : shard) {
            HIPCK(hipSetDevice(e->device));
            HIPCK(hipStreamSynchronize(e->stream));
        }
        for (size_t d = 0; d < devs.size(); ++d) {
            HIPCK(hipSetDevice(devs[d]));
            HIPCK(hipStreamSynchronize(dev_stream[d]));
        }
    }

    // steps (2)-(4) of a batch whose shard launches have been issued into buffer b: per-device accumulation, ONE reduce over
    // xGMI, post-mix stage on the root, hand-over to the pinned staging buffer (host_ready[b] fires when it is there)
    void reduce_and_hand_over(int b, size_t nf)
    {
        const size_t n_sh = shard.size();
        const size_t vc = vch;
        // (2) per device: add the buffers of the other shards on that device into the first one's
        for (size_t d = 0; d < devs.size(); ++d) {
            HIPCK(hipSetDevice(devs[d]));
            float* acc = nullptr;
            for (size_t s = 0; s < n_sh; ++s) {
                if (dev_of_shard[s] != (int)d) continue;
                HIPCK(hipStreamWaitEvent(dev_stream[d], sh_done[b][s], 0));
                if (!acc) {
                    acc = sh_buf[b][s];
                } else {
                    hipLaunchKernelGGL(og_bus_accumulate, dim3((uint32_t)((nf * vc + 255) / 256)), dim3(256), 0, dev_stream[d], acc,
                                       sh_buf[b][s], nf * vc);
                }
            }
            dev_acc[b][d] = acc;
        }
        // (3) ONE reduce of the whole batch over xGMI (root = devs[0])
        if (use_rccl) {
            Rccl& R = rccl();
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if (time_reduces) { // (og_cluster_enable_reduce_timing: device time of the reduce on the ROOT's stream)
                HIPCK(hipSetDevice(devs[0]));
                if (reduce_events.size() >= 256) fold_reduce_events(false); // (a caller that never asks must not pile up events)
                HIPCK(hipEventCreate(&t0));
                HIPCK(hipEventCreate(&t1));
                reduce_events.push_back({t0, t1}); // (owned by the list from here on: an error below does not leak them)
                HIPCK(hipEventRecord(t0, dev_stream[0]));
            }
            R.ck(R.GroupStart(), "ncclGroupStart");
            for (size_t d = 0; d < devs.size(); ++d)
                R.ck(R.Reduce(dev_acc[b][d], dev_acc[b][d], nf * vc, ncclFloat, ncclSum, 0, comms[d], dev_stream[d]), "ncclReduce");
            R.ck(R.GroupEnd(), "ncclGroupEnd");
            if (time_reduces) {
                HIPCK(hipSetDevice(devs[0]));
                HIPCK(hipEventRecord(t1, dev_stream[0]));
            }
            n_reduces += 1;
        }
        // (4) root: post-mix stage, then hand the batch to the host through a pinned staging buffer
        HIPCK(hipSetDevice(devs[0]));
        const float* mono = dev_acc[b][0];
        if (tremolo) {
            og_engine* e0 = shard[0];
            ogc::UEnv env = e0->env();
            const float rate = e0->cg->tremolo_rate(env), depth = e0->cg->tremolo_depth(env);
            for (size_t g = 0; g < nf; g += OG_MAX_BLOCK) {
                const uint32_t frames = (uint32_t)std::min<size_t>(OG_MAX_BLOCK, nf - g);
                hipLaunchKernelGGL(og_bus_tremolo, dim3(1), dim3(512), 0, dev_stream[0], mono + g, frames, rate, depth, e0->sr,
                                   d_phase, d_out + 2 * g);
            }
            HIPCK(hipMemcpyAsync(h_pin[b], d_out, nf * 2 * sizeof(float), hipMemcpyDeviceToHost, dev_stream[0]));
        } else {
            HIPCK(hipMemcpyAsync(h_pin[b], mono, nf * vc * sizeof(float), hipMemcpyDeviceToHost, dev_stream[0])); // (Frame<2> voices: interleaved L R)
        }
        HIPCK(hipEventRecord(host_ready[b], dev_stream[0]));
        // buffer b may be rewritten once every reader on its device is done
        for (size_t d = 0; d < devs.size(); ++d) {
            HIPCK(hipSetDevice(devs[d]));
            HIPCK(hipEventRecord(dev_free[b][d], dev_stream[d]));
        }
    }

    // ONE block, for the real-time entry (og_cluster_process_block): the CALLING thread issues every shard's two launches
    // itself (a block is one voice-kernel launch and one bus reduce per shard: ~10 us of host time each) instead of
    // waking a thread per shard, and waits for the hand-over by polling the event -- a condition-variable wake-up or a
    // blocking event wait costs a scheduler round trip, which on a loaded host is where a 5.3 ms audio deadline is
    // lost (measured on a shared box: 7 ms outliers through the threaded path).  The shard buffers alternate between
    // calls; the events that guard them are the same as in render().
    int rt_buf = 0;
    void process_one_block(uint32_t frames, float* out)
    {
        ensure_buffers(frames);
        const int b = rt_buf;
        rt_buf ^= 1;
        const size_t vc = vch;
        for (size_t s = 0; s < shard.size(); ++s) {
            og_engine* e = shard[s];
            HIPCK(hipSetDevice(e->device));
            HIPCK(hipStreamWaitEvent(e->stream, dev_free[b][dev_of_shard[s]], 0));
            e->process_async(frames, sh_buf[b][s]);
            e->flush_bus();
            HIPCK(hipEventRecord(sh_done[b][s], e->stream));
        }
        (void)vc;
        reduce_and_hand_over(b, frames);
        HIPCK(hipSetDevice(devs[0]));
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(host_ready[b]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIPCK(q);
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
                HIPCK(hipEventSynchronize(host_ready[b])); // (something far slower than a block is in front of it)
                break;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        memcpy(out, h_pin[b], (size_t)frames * channels * sizeof(float));
    }

    // Render `total_frames` in blocks of `block`; the summed (and post-mixed) bus ends up in out (host).
    void render(uint64_t total_frames, uint32_t block, float* out)
    {
        const size_t batch_frames = (size_t)std::min<uint64_t>(total_frames, (uint64_t)CL_BATCH_BLOCKS * block);
        ensure_buffers(batch_frames);
        const size_t n_sh = shard.size();
        const size_t vc = vch; // floats per frame of a shard's voice sum: 1, or 2 for Frame<2> voice outputs (interleaved)
        int b = 0;
        bool have_prev = false;
        uint64_t prev_f0 = 0;
        size_t prev_nf = 0;
        for (uint64_t f0 = 0; f0 < total_frames; f0 += batch_frames, b ^= 1) {
            const size_t nf = (size_t)std::min<uint64_t>(batch_frames, total_frames - f0);
            // (1) every shard renders the batch into its buffer b, on its own stream, issued by its own thread
            for (size_t s = 0; s < n_sh; ++s) {
                og_engine* e = shard[s];
                float* buf = sh_buf[b][s];
                hipEvent_t done = sh_done[b][s];
                hipEvent_t free_ev = dev_free[b][dev_of_shard[s]];
                workers[s]->run([=] {
                    HIPCK(hipSetDevice(e->device));
                    HIPCK(hipStreamWaitEvent(e->stream, free_ev, 0)); // the reduce that last read this buffer is over
                    for (size_t g = 0; g < nf; g += block) {
                        const uint32_t frames = (uint32_t)std::min<size_t>(block, nf - g);
                        e->process_async(frames, buf + g * vc);
                    }
                    e->flush_bus(); // (shards queue 8..32 blocks per launch: og_cluster_create)
                    HIPCK(hipEventRecord(done, e->stream));
                });
            }
            wait_all();
            reduce_and_hand_over(b, nf);
            // drain the PREVIOUS batch while this one runs
            if (have_prev) {
                HIPCK(hipEventSynchronize(host_ready[b ^ 1]));
                memcpy(out + prev_f0 * channels, h_pin[b ^ 1], prev_nf * channels * sizeof(float));
            }
            have_prev = true;
            prev_f0 = f0;
            prev_nf = nf;
        }
        if (have_prev) {
            HIPCK(hipSetDevice(devs[0]));
            HIPCK(hipEventSynchronize(host_ready[b ^ 1]));
            memcpy(out + prev_f0 * channels, h_pin[b ^ 1], prev_nf * channels * sizeof(float));
        }
        sync_all();
    }
};

extern "C" {

int og_cluster_create(const og_graph_desc* g, uint64_t n_voices_total, const int* device_ids, uint32_t n_shards, og_cluster** out)
{
    if (!g || !device_ids || !out) return set_err(OG_E_INVALID, "null argument");
    if (n_shards == 0 || n_voices_total < n_shards) return set_err(OG_E_INVALID, "need at least one voice per shard");
    return guard([&] {
        std::unique_ptr<og_cluster> c(new og_cluster);
        c->total = n_voices_total;
        for (uint32_t s = 0; s <= n_shards; ++s) c->lo.push_back(n_voices_total * s / n_shards);
        for (uint32_t s = 0; s < n_shards; ++s) {
            const uint64_t nv = c->lo[s + 1] - c->lo[s];
            if (nv > 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 voices in one shard");
            og_engine* raw = nullptr;
            const int rc = og_create(g, (uint32_t)nv, device_ids[s], &raw);
            if (rc != OG_OK) {
                if (rc == OG_E_DEVICE) throw HipError(g_err);
                throw std::runtime_error(g_err);
            }
            std::unique_ptr<og_engine> e(raw); // (owned here until the cluster has it: alloc_bus_buffers may throw)
            e->bus_stage = false; // shards hand over the voice sum; the post-mix node runs once, on the root
            {
                HIPCK(hipSetDevice(e->device));
                e->alloc_bus_buffers(e->auto_batch()); // shards render 8..32 blocks per launch
                e->bus_batch = e->auto_batch();
            }
            c->shard.push_back(e.get());
            e.release();
            int di = -1;
            for (size_t d = 0; d < c->devs.size(); ++d)
                if (c->devs[d] == device_ids[s]) di = (int)d;
            if (di < 0) {
                c->devs.push_back(device_ids[s]);
                di = (int)c->devs.size() - 1;
            }
            c->dev_of_shard.push_back(di);
            c->workers.emplace_back(new Worker);
        }
        c->channels = c->shard[0]->cg->channels;
        c->vch = c->shard[0]->cg->voice_channels;
        c->tremolo = c->shard[0]->cg->bus_tremolo;
        const size_t nd = c->devs.size();
        c->dev_stream.assign(nd, nullptr);
        for (int b = 0; b < 2; ++b) {
            c->sh_buf[b].assign(n_shards, nullptr);
            c->sh_done[b].assign(n_shards, nullptr);
            c->dev_free[b].assign(nd, nullptr);
            c->dev_acc[b].assign(nd, nullptr);
        }
        for (size_t d = 0; d < nd; ++d) {
            HIPCK(hipSetDevice(c->devs[d]));
            HIPCK(hipStreamCreateWithFlags(&c->dev_stream[d], hipStreamNonBlocking));
            for (int b = 0; b < 2; ++b) {
                HIPCK(hipEventCreateWithFlags(&c->dev_free[b][d], hipEventDisableTiming));
                HIPCK(hipEventRecord(c->dev_free[b][d], c->dev_stream[d]));
            }
        }
        for (uint32_t s = 0; s < n_shards; ++s) {
            HIPCK(hipSetDevice(c->shard[s]->device));
            for (int b = 0; b < 2; ++b) HIPCK(hipEventCreateWithFlags(&c->sh_done[b][s], hipEventDisableTiming));
        }
        HIPCK(hipSetDevice(c->devs[0]));
        if (c->tremolo) {
            HIPCK(hipMalloc(&c->d_phase, 4));
            HIPCK(hipMemset(c->d_phase, 0, 4));
        }
        const char* force = ogabi::experiment_knob("OSCEN_GPU_FORCE_RCCL"); // exercise the RCCL leg on a one-device cluster (tests)
        if (nd > 1 || (force && atoi(force) != 0)) {
            Rccl& R = rccl();
            R.load();
            c->comms.assign(nd, nullptr);
            R.ck(R.CommInitAll(c->comms.data(), (int)nd, c->devs.data()), "ncclCommInitAll");
            c->use_rccl = true;
        }
        *out = c.release();
        return OG_OK;
    });
}

void og_cluster_destroy(og_cluster* c) { delete c; }

int og_cluster_init(og_cluster* c, float sample_rate)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    for (og_engine* e : c->shard) {
        const int rc = og_init(e, sample_rate);
        if (rc != OG_OK) return rc;
    }
    return guard([&] {
        if (c->d_phase) {
            HIPCK(hipSetDevice(c->devs[0]));
            HIPCK(hipMemset(c->d_phase, 0, 4));
        }
        c->inited = true;
        return OG_OK;
    });
}

uint32_t og_cluster_num_shards(const og_cluster* c) { return c ? (uint32_t)c->shard.size() : 0; }
uint32_t og_cluster_num_devices(const og_cluster* c) { return c ? (uint32_t)c->devs.size() : 0; }
uint64_t og_cluster_num_voices(const og_cluster* c) { return c ? c->total : 0; }
uint32_t og_cluster_channels(const og_cluster* c) { return c ? c->channels : 0; }
uint64_t og_cluster_rccl_reduces(const og_cluster* c) { return c ? c->n_reduces : 0; }
og_engine* og_cluster_shard(og_cluster* c, uint32_t s, uint64_t* first_voice)
{
    if (!c || s >= c->shard.size()) return nullptr;
    if (first_voice) *first_voice = c->lo[s];
    return c->shard[s];
}

// Event outputs of the graph over a cluster: every shard keeps its own log (a voice lives in one shard); the logs are
// merged into (frame, GLOBAL voice, push order) -- what one engine of that size would hand over.
int og_cluster_read_output_events(og_cluster* c, og_out_event* buf, uint32_t cap, uint32_t* n_out, uint64_t* n_overflowed)
{
    return ogabi::guard([&]() -> int {
    if (!c || (cap && !buf) || !n_out) return set_err(OG_E_INVALID, "null argument");
    *n_out = 0;
    if (n_overflowed) *n_overflowed = 0;
    if (c->total > 0xFFFFFFFFull) return set_err(OG_E_UNSUPPORTED, "event outputs carry 32-bit voice ids");
    // drain every shard's queue through one persistent scratch buffer into the cluster's own queue (global voice ids);
    // what the caller's buffer cannot take stays queued for the next call
    std::vector<og_out_event>& all = c->out_ev_carry;
    const size_t old = all.size();
    uint64_t over = 0;
    if (c->out_ev_scratch.empty()) c->out_ev_scratch.resize(65536);
    for (size_t s = 0; s < c->shard.size(); ++s) {
        og_engine* e = c->shard[s];
        if (!e->d_out_ev) continue;
        uint32_t n = 0;
        do {
            uint64_t o = 0;
            const int rc = og_read_output_events(e, c->out_ev_scratch.data(), (uint32_t)c->out_ev_scratch.size(), &n, &o);
            if (rc != OG_OK) return rc;
            over += o;
            for (uint32_t i = 0; i < n; ++i) {
                og_out_event x = c->out_ev_scratch[i];
                x.voice += (uint32_t)c->lo[s];
                all.push_back(x);
            }
        } while (n == c->out_ev_scratch.size());
    }
    std::stable_sort(all.begin() + (ptrdiff_t)old, all.end(), [](const og_out_event& a, const og_out_event& b) {
        if (a.frame != b.frame) return a.frame < b.frame;
        return a.voice < b.voice; // (a voice's events of one frame stay in push order: each shard's list is sorted)
    });
    const size_t give = std::min<size_t>(all.size(), cap);
    for (size_t i = 0; i < give; ++i) buf[i] = all[i];
    all.erase(all.begin(), all.begin() + (ptrdiff_t)give);
    *n_out = (uint32_t)give;
    if (n_overflowed) *n_overflowed = over; // (only what a shard's device log could not hold)
    return OG_OK;
    });
}

// Device time of the batched ncclReduce on the root's stream (HIP events around it): lets a caller -- bench.py --cluster --
// say how much of a multi-GPU run is the xGMI reduce.  The time between the two events includes waiting for the slowest
// device's contribution.  Query after og_cluster_render / og_cluster_synchronize returned (the events are complete then).
int og_cluster_enable_reduce_timing(og_cluster* c, int on)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    return guard([&] {
        c->drop_reduce_events();
        c->time_reduces = on != 0;
        return OG_OK;
    });
}
int og_cluster_reduce_time_ms(og_cluster* c, double* total_ms, uint64_t* n_reduces)
{
    if (!c || !total_ms || !n_reduces) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        if (!c->devs.empty()) HIPCK(hipSetDevice(c->devs[0]));
        c->fold_reduce_events(true);
        *total_ms = c->reduce_ms_folded;
        *n_reduces = c->reduce_n_folded;
        c->reduce_ms_folded = 0.0;
        c->reduce_n_folded = 0;
        return OG_OK;
    });
}

int og_cluster_group_voices(og_cluster* c, uint32_t policy)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    return guard([&]() -> int {
        for (size_t s = 0; s < c->shard.size(); ++s) {
            const int rc = og_group_voices(c->shard[s], policy);
            if (rc != OG_OK) { // all shards or none: the ones already re-ordered go back to the identity (og_last_error keeps the cause)
                const std::string why = og_last_error();
                for (size_t k = 0; k < s; ++k) (void)og_group_voices(c->shard[k], 0u);
                return set_err(rc, why);
            }
        }
        return OG_OK;
    });
}

// fold the device-side drop counts of every shard into the counters og_cluster_events_dropped reads; the first error is
// reported (and the remaining shards are still synchronised)
int og_cluster_sync_event_counters(og_cluster* c)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    return guard([&]() -> int {
        int first = OG_OK;
        std::string why;
        for (og_engine* e : c->shard) {
            const int rc = og_sync_event_counters(e);
            if (rc != OG_OK && first == OG_OK) {
                first = rc;
                why = og_last_error();
            }
        }
        return first == OG_OK ? OG_OK : set_err(first, why);
    });
}

uint64_t og_cluster_events_dropped(og_cluster* c)
{
    uint64_t d = 0;
    // (the caller is the rendering thread: og_cluster* is not const here.  A device error while the counters are folded
    //  stays in og_last_error(); og_cluster_sync_event_counters is the call that RETURNS it)
    if (c) (void)og_cluster_sync_event_counters(c);
    for (size_t s = 0; c && s < c->shard.size(); ++s) d += og_events_dropped(c->shard[s]);
    return d;
}

int og_cluster_input_index(const og_cluster* c, const char* name) { return c ? og_input_index(c->shard[0], name) : set_err(OG_E_INVALID, "null cluster"); }

#define OG_CLUSTER_BROADCAST(call)                   \
    if (!c) return set_err(OG_E_INVALID, "null cluster"); \
    for (og_engine* e : c->shard) {                  \
        const int rc = call;                         \
        if (rc != OG_OK) return rc;                  \
    }                                                \
    return OG_OK;

int og_cluster_set_value(og_cluster* c, uint32_t input, float v) { OG_CLUSTER_BROADCAST(og_set_value(e, input, v)) }
int og_cluster_set_value_ramp(og_cluster* c, uint32_t input, float v, uint32_t frames) { OG_CLUSTER_BROADCAST(og_set_value_ramp(e, input, v, frames)) }
int og_cluster_set_value_immediate(og_cluster* c, uint32_t input, float v) { OG_CLUSTER_BROADCAST(og_set_value_immediate(e, input, v)) }

int og_cluster_set_voice_values(og_cluster* c, uint32_t input, uint64_t first_voice, uint64_t count, const float* v)
{
    return ogabi::guard([&]() -> int {
    if (!c || !v) return set_err(OG_E_INVALID, "null argument");
    if (first_voice + count > c->total) return set_err(OG_E_INVALID, "voice range out of bounds");
    for (size_t s = 0; s < c->shard.size(); ++s) {
        const uint64_t a = std::max(first_voice, c->lo[s]), b = std::min(first_voice + count, c->lo[s + 1]);
        if (a >= b) continue;
        const int rc = og_set_voice_values(c->shard[s], input, (uint32_t)(a - c->lo[s]), (uint32_t)(b - a), v + (a - first_voice));
        if (rc != OG_OK) return rc;
    }
    return OG_OK;
    });
}

int og_cluster_push_voice_event(og_cluster* c, uint32_t input, uint64_t voice, uint32_t frame_offset, float scalar)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    if (voice >= c->total) return set_err(OG_E_INVALID, "voice index out of range");
    const uint32_t s = c->shard_of(voice);
    return og_push_voice_event(c->shard[s], input, (uint32_t)(voice - c->lo[s]), frame_offset, scalar);
}

int og_cluster_push_voice_value(og_cluster* c, uint32_t input, uint64_t voice, uint32_t frame_offset, float v)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    if (voice >= c->total) return set_err(OG_E_INVALID, "voice index out of range");
    const uint32_t s = c->shard_of(voice);
    return og_push_voice_value(c->shard[s], input, (uint32_t)(voice - c->lo[s]), frame_offset, v);
}

int og_cluster_schedule_voice_events(og_cluster* c, uint32_t input, uint64_t n, const uint64_t* voices, const uint64_t* abs_frames,
                                     const float* values)
{
    return ogabi::guard([&]() -> int {
    if (!c || (n && (!voices || !abs_frames || !values))) return set_err(OG_E_INVALID, "null argument");
    std::vector<std::vector<uint32_t>> lv(c->shard.size());
    std::vector<std::vector<uint64_t>> lf(c->shard.size());
    std::vector<std::vector<float>> lx(c->shard.size());
    for (uint64_t i = 0; i < n; ++i) {
        if (voices[i] >= c->total) return set_err(OG_E_INVALID, "voice index out of range");
        const uint32_t s = c->shard_of(voices[i]);
        lv[s].push_back((uint32_t)(voices[i] - c->lo[s]));
        lf[s].push_back(abs_frames[i]);
        lx[s].push_back(values[i]);
    }
    for (size_t s = 0; s < c->shard.size(); ++s) {
        if (lv[s].empty()) continue;
        const int rc = og_schedule_voice_events(c->shard[s], input, (uint32_t)lv[s].size(), lv[s].data(), lf[s].data(), lx[s].data());
        if (rc != OG_OK) return rc;
    }
    return OG_OK;
    });
}

int og_cluster_render(og_cluster* c, uint64_t total_frames, uint32_t block, float* out_bus)
{
    if (!c || !out_bus) return set_err(OG_E_INVALID, "null argument");
    if (block == 0 || block > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "block must be in 1..512");
    if (!c->inited) return set_err(OG_E_STATE, "og_cluster_init must be called before processing");
    if (total_frames == 0) return OG_OK;
    return guard([&] {
        c->render(total_frames, block, out_bus);
        return OG_OK;
    });
}

int og_cluster_process_block(og_cluster* c, uint32_t frames, float* out_bus)
{
    if (!c) return set_err(OG_E_INVALID, "null cluster");
    if (frames > OG_MAX_BLOCK_SIZE) return set_err(OG_E_INVALID, "frames must be in 0..512");
    if (!c->inited) return set_err(OG_E_STATE, "og_cluster_init must be called before processing");
    if (frames == 0) {
        for (og_engine* e : c->shard) {
            const int rc = og_process_block_async(e, 0, nullptr);
            if (rc != OG_OK) return rc;
        }
        return OG_OK;
    }
    if (!out_bus) return set_err(OG_E_INVALID, "null argument");
    return guard([&] {
        c->process_one_block(frames, out_bus);
        return OG_OK;
    });
}

} // extern "C"
