#!/usr/bin/env python
"""Benchmark of the hot path: voices*samples/sec of the fm-synth voice bank.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one 256-frame block of the fm_voice graph over this rank's voice shard (BASELINE.json
configs[1]: 65 536 voices per GPU, 48 kHz, f32, synthetic note streams already resident in HBM;
`--voices-per-gpu 262144 --gpus 8` is configs[3], the 2 097 152-voice node).  With --gpus N > 1 and
no torch.distributed environment the script launches itself under torch.distributed.run (one rank
per GPU, RCCL); the driver's own `python -m torch.distributed.run ... bench.py --gpus N` works too.
Rank 0 prints ONE JSON line.

Other modes (not the headline): --midi-live (events arrive through og_midi_send + the blocking
og_process_block every block instead of a resident timeline), --graph <built-in> for the other
BASELINE configurations.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def pmc_profile(voices, block, graph, kernel_hash, blocks_per_launch=None):
    """The committed rocprofv3 PMC summary (profiles/*_summary.json, scripts/prof_summary.py) of THIS kernel:
    same graph, bank size, block and kernel hash.  FETCH_SIZE / WRITE_SIZE / SQ_* come from separate --pmc
    runs, so they cannot be measured in-process; a summary of another kernel build is never mixed in --
    it is reported as stale instead."""
    match, stale = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("voices") != voices or d.get("frames") != block or d.get("graph", "fm_voice") != graph:
            continue
        if d.get("kernel_hash") == kernel_hash and "hbm_traffic" in d:
            # several summaries of this kernel: the one whose command queued the same number of blocks per launch
            def off(x):
                return abs(x.get("blocks_per_launch", 1.0) - (blocks_per_launch or x.get("blocks_per_launch", 1.0)))
            if match is None or off(d) <= off(match[1]):
                match = (path, d)
        else:
            stale = path
    if match:
        d = match[1]
        valu = d.get("pmc_voice_kernel", {}).get("SQ_INSTS_VALU", {}).get("avg_per_dispatch")
        return {"bytes": d["hbm_traffic"]["total_bytes_corrected"], "valu": valu, "blocks_per_launch": d.get("blocks_per_launch", 1.0),
                "source": os.path.relpath(match[0], ROOT), "stale": None}
    return {"bytes": None, "valu": None, "blocks_per_launch": None, "source": None,
            "stale": os.path.relpath(stale, ROOT) if stale else None}


def cpu_baseline(block, seed, frames, span):
    """Time the CPU oracle (scalar C port of the reference path) on the host cores, bounded sample of the
    same note streams.  Two shapes of the same arithmetic: banks of 8 voices per graph, each rendered block
    by block (the reference's own `voices = [FMVoice; 8]`, examples/fm-synth/src/lib.rs:68-73, whose state
    stays in cache) -- the headline CPU figure -- and one big bank per thread walked voice-minor every sample
    (cache-hostile; what round 1 reported)."""
    import ctypes as C

    from tests import oracle_lib as ol

    lib = ol.load()
    lib.oo_bench_set_fold(ol.FOLD["slice"])
    cores = os.cpu_count() or 1
    cs = C.c_double()
    frames = int(max(block, min(frames, 48000)))

    def timed(voices, nframes, threads, group):
        t = lib.oo_bank_bench_grouped(ol.BANK_FM, voices, nframes, block, threads, group, seed, span, C.byref(cs))
        return voices * nframes / max(t, 1e-9)

    # How many threads does this box really run?  os.cpu_count() reports the host's CPUs, a container may be
    # throttled far below that (the round-1 figure of 1.6e5 voices*samples/s per thread against 2.6e6 on one
    # unconstrained core was this, not cache misses).  A short scaling probe picks the thread count with the best
    # aggregate rate; the single-thread rate is reported next to it.
    pf = min(frames, 1024)
    single = max(timed(64, pf, 1, 8), timed(64, pf, 1, 8))
    best_t, best_rate = 1, single
    t = 2
    cand = []
    while t < cores:
        cand.append(t)
        t *= 2
    cand.append(cores)
    for t in cand:
        r = max(timed(64 * t, pf, t, 8), timed(64 * t, pf, t, 8))
        if r > best_rate:
            best_t, best_rate = t, r
    threads = best_t

    def run(group, budget_s, rate_guess):
        for _ in range(3):  # size the sample for ~budget_s seconds of wall time, re-sizing once or twice if the guess was off
            voices = int(max(8 * threads, min(1 << 21, rate_guess * budget_s / frames)))
            voices = max(8 * threads, (voices // (8 * threads)) * 8 * threads)
            rate = timed(voices, frames, threads, group)
            secs = voices * frames / rate
            if secs >= 0.4 * budget_s or voices >= (1 << 21):
                break
            rate_guess = rate
        return voices, secs, 0.0

    v8, t8, _ = run(8, 4.0, best_rate)  # (a throttled container sustains less than the probe burst: lands at 10-20 s)
    vw, tw, _ = run(0, 1.5, best_rate * 0.5)
    return {
        "value": v8 * frames / t8,
        "unit": "voices*samples/s",
        "cores": threads,
        "cpu_count": cores,
        "kind": "port",
        "per_thread": v8 * frames / t8 / threads,
        "single_thread": single,
        "sample": "%d voices x %d frames (block %d) of the same synthetic fm-synth note streams as banks of 8 voices "
                  "(the reference's [FMVoice; 8] graph) rendered block by block, C oracle, %d threads (best of a scaling "
                  "probe over 1..%d; one thread alone: %.3g), %.1f s" % (v8, frames, block, threads, cores, single, t8),
        "whole_bank_per_thread": {
            "value": vw * frames / tw,
            "per_thread": vw * frames / tw / threads,
            "sample": "%d voices x %d frames, one bank per thread walked voice-minor per sample (cache-hostile), "
                      "%d threads, %.1f s" % (vw, frames, threads, tw),
        },
    }


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed environment: one rank per GPU over RCCL."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=188)   # 188 x 256 frames = 1 s of audio
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--voices-per-gpu", type=int, default=65536,
                    help="65536 = BASELINE configs[1]; 262144 with --gpus 8 = configs[3] (2 097 152 voices)")
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--graph", default="fm_voice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bus-batch", type=int, default=0,
                    help="blocks the engine may render per kernel launch (og_set_bus_batching: 0 = the engine's choice, 8..32 "
                         "by bank size; 1 = a launch and a bus reduce per block)")
    ap.add_argument("--sparse-events", action="store_true",
                    help="keep the 1 s note plan as is even when the run is shorter (default: every voice plays a slice "
                         "of its cyclic plan, so that note-on, note-off and retrigger all fall inside the timed region at "
                         "the plan's real density)")
    ap.add_argument("--midi-live", type=int, default=0, metavar="N",
                    help="live path: N MIDI messages per block through og_midi_send_batch; the block is enqueued with "
                         "og_midi_process_block_async (the host parses block k+1 while block k renders)")
    ap.add_argument("--midi-blocking", action="store_true",
                    help="with --midi-live: the blocking drop-in entry og_midi_process_block (= process_block + bus to host)")
    # plumbing checks of the multi-rank path on a 1-GPU box (not a benchmark configuration):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--single-device", action="store_true", help="map every rank onto GPU 0")
    ap.add_argument("--dist-single", action="store_true",
                    help="run the RCCL leg (process group, reduce, barrier) with a one-rank communicator: RCCL refuses two "
                         "ranks on one GPU ('Duplicate GPU detected'), so this is how a one-GPU box exercises it")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    import oscen_amd
    from oscen_amd import distributed as ogd

    rank, local_rank, world_size = ogd.world()
    if world_size != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world_size))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_ranks = None
    if world_size > 1 or args.dist_single:
        import torch.distributed as dist

        if world_size == 1:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            s.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, rank=rank, world_size=world_size)

    V = args.voices_per_gpu
    total_voices = V * world_size
    lo, hi = ogd.shard_range(rank, world_size, total_voices)
    block, K, W = args.block, args.steps, args.warmup
    total_frames = (K + W) * block
    span = 0 if args.sparse_events else min(total_frames, 48000)

    eng = oscen_amd.Engine(args.graph, hi - lo, device=local_rank, sample_rate=48000.0)
    # global voice ids keep their note streams; a run shorter than the 1 s score sees a slice of it at its real density
    plans = oscen_amd.note_plans(hi - lo, first_voice=lo, span=span, fold="slice")
    midi = None
    if args.midi_live:
        eng.set_voice_values("frequency", plans["frequency"])
        midi = oscen_amd.Midi(eng)
        midi.set_queue_capacity(max(32, args.midi_live))
        n_events_timed = args.midi_live * K
    else:
        oscen_amd.schedule_note_plans(eng, plans, total_frames=total_frames)
        ev_f = plans["events"][1]
        n_events_timed = int(np.count_nonzero((ev_f >= W * block) & (ev_f < total_frames))) if "gate" in eng.input_names else 0
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    if args.bus_batch != 1:  # queued blocks share a launch; every bus is complete before the timed region closes (flush below)
        eng.set_bus_batching(args.bus_batch)
    ch = eng.channels
    bus = torch.zeros((K + W, block * ch), dtype=torch.float32, device="cuda")
    base = bus.data_ptr()
    host_bus = np.zeros((K + W, block * ch), dtype=np.float32)
    rng = np.random.default_rng(0x05CE2026 + rank)
    live = []
    if midi is not None:  # messages built once: the timed loop pays for the engine's live path, not for numpy
        live_notes = rng.integers(36, 97, size=(K + W, args.midi_live)).astype(np.uint8)
        live_frames = np.sort(rng.integers(0, block, size=(K + W, args.midi_live)), axis=1).astype(np.uint32)
        for i in range(K + W):  # even blocks play notes, odd blocks release the notes of the block before
            live.append(midi.pack_messages(live_notes[i - (i % 2)], live_frames[i], on=(i % 2 == 0)))

    def step(i):
        if midi is None:
            eng.process_block_async(block, base + i * block * ch * 4)
        else:
            midi.send_packed(live[i])
            if args.midi_blocking:
                host_bus[i] = midi.process_block(block).reshape(-1)
            else:
                midi.process_block_async(block, base + i * block * ch * 4)

    def reduce_bus(t):
        if args.backend == "gloo":  # CPU collective (plumbing check only)
            h = t.cpu()
            ogd.reduce_bus(h)
            t.copy_(h)
        else:
            ogd.reduce_bus(t)

    def barrier():
        if args.backend == "gloo":
            dist.barrier()
        else:
            dist.barrier(device_ids=[local_rank])

    # Warm-up: W - 1 blocks, the communicator set-up, the barrier -- and then the LAST warm-up block, so that the voice
    # kernel is the most recent thing every CU ran when the clock starts.  Measured on MI355X: the first launch of the
    # voice kernel after OTHER kernels (RCCL's, or any torch kernel) is ~35 % slower than in a steady stream of blocks
    # (same instruction count, +57 % instruction-fetch wait: its code has to come back from HBM), and with one launch per
    # 20-32 blocks that launch is the whole timed region -- a measurement artefact of the barrier, not of the path.
    for i in range(max(0, W - 1)):
        step(i)
    eng.flush()
    if dist is not None:  # communicator set-up (lazy in RCCL) must not land in the timed region
        reduce_bus(bus[:max(1, W - 1)] if W > 1 else torch.zeros((1, block * ch), dtype=torch.float32, device="cuda"))
        ones = torch.ones(1, dtype=torch.float32, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(ones)  # every rank of the communicator took part
        rccl_ranks = int(round(float(ones.item())))
    torch.cuda.synchronize()
    if dist is not None:
        barrier()
    torch.cuda.synchronize()
    if W > 0:
        step(W - 1)
        eng.flush()
        torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    eng.flush()  # the reduces of the last (partial) batch of blocks: inside the timed region
    if dist is not None:
        reduce_bus(bus[W:])  # ONE RCCL reduce of the [K, block] mix bus over xGMI
    torch.cuda.synchronize()
    if dist is not None:
        barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms, n_launch = eng.kernel_time_ms()       # average duration of a voice-kernel LAUNCH (HIP events on the engine's stream)
    n_blocks_timed = eng.kernel_blocks_timed         # blocks those launches rendered (up to --bus-batch per launch)
    eng.enable_kernel_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        mix = host_bus[W:] if (midi is not None and args.midi_blocking) else bus[W:].float().cpu().numpy()
        assert np.isfinite(mix).all() and np.abs(mix).max() > 0.0, "bus is silent or non-finite"
        value = total_voices * K * block / elapsed
        words = eng.state_words_per_voice
        lanes = eng.voices_per_wave * eng.lanes_per_voice
        # algorithmic HBM bytes of one launch: state planes read once + written once,
        # the two event-cursor words read per voice, one partial-bus row written per workgroup
        n_wg = eng.partial_rows
        # SURVEY 8(d)'s per-unit figure (state read + written once per 256-frame block, event cursors, one partial row per
        # workgroup) x the units one launch processes: a launch that renders several queued blocks is charged that many
        # blocks' worth, although it touches the state planes only once
        bytes_per_block = V * (4 * (words + eng.state_words_written_per_voice) + 8) + n_wg * block * 4
        # graphs with a Delay: every voice-sample reads one and writes one 4-byte slot of its HBM ring
        bytes_per_block += V * block * 8 * {"echo_voice": 1}.get(args.graph, 0)
        blocks_per_launch = n_blocks_timed / float(max(1, n_launch))
        bytes_per_launch = bytes_per_block * blocks_per_launch
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        prof = pmc_profile(V, block, args.graph, eng.kernel_hash, blocks_per_launch)
        pmc_bytes, pmc_valu, pmc_src = prof["bytes"], prof["valu"], prof["source"]
        pmc_note = None
        if pmc_src and abs(prof["blocks_per_launch"] - blocks_per_launch) > 0.02 * blocks_per_launch:
            # the profiled command queued a different number of blocks per launch: instruction counts scale with the frames
            # rendered; HBM traffic does not (the state planes are touched once per launch) and is not extrapolated
            pmc_note = ("profile has %.2f blocks per launch, this run %.2f: VALU count scaled by the ratio, traffic omitted"
                        % (prof["blocks_per_launch"], blocks_per_launch))
            pmc_valu = pmc_valu * blocks_per_launch / prof["blocks_per_launch"] if pmc_valu else None
            pmc_bytes = None
        traffic = pmc_bytes / (kern_ms * 1e-3) / 1e9 if (pmc_bytes and kern_ms > 0) else None
        line = {
            "metric": "voices*samples/sec (fm-synth graph, 48 kHz)",
            "value": value,
            "unit": "voices*samples/s",
            "n_gpus": world_size,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "fm-synth voice bank (FMVoice graph), %d voices/GPU, block=%d frames, 48 kHz, f32; "
                            "synthetic note streams splitmix64(0x05CE2026 ^ voice) resident in HBM%s; "
                            "%d note events (on / off / retrigger) fall inside the timed region; "
                            "mix bus reduced once per run over RCCL"
                            % (V, block, "" if not span or span >= 48000 else
                               ", the %d-frame run shows a per-voice slice of the cyclic 1 s note plan at its real event density" % span,
                               n_events_timed),
                "graph": args.graph,
                "voices_per_gpu": V,
                "total_voices": total_voices,
                "block": block,
                "sample_rate": 48000,
                "parallelism": "voice-shard x%d" % world_size,
                "events_in_timed_region": n_events_timed,
                "note_plan_span_frames": span if span else 48000,
                "blocks_per_launch_limit": args.bus_batch if args.bus_batch else "engine's choice (8..32 by bank size)",
                "event_path": ("midi-live (og_midi_send_batch + %s per block)" %
                               ("og_midi_process_block, blocking" if args.midi_blocking else "og_midi_process_block_async"))
                              if midi is not None
                              else "resident timeline (og_schedule_voice_events)",
            },
            "rccl_ranks": rccl_ranks,
            "realtime_voices_at_48k": value / 48000.0,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_bytes_per_launch": pmc_bytes,
                "traffic_source": pmc_src,
                "stale_profile": prof["stale"],  # newest summary of this configuration taken on ANOTHER kernel build
                "profile_note": pmc_note,
                "kernel_hash": eng.kernel_hash,
                "kernel_variant": eng.kernel_variant,
                "kernel_ms_avg": kern_ms,
                "kernel_launches": n_launch,
                "blocks_per_launch": blocks_per_launch,
                "kernel_ms_per_block": kern_ms / blocks_per_launch if blocks_per_launch else None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "voices_per_wave": lanes,
                "pipeline_waves_per_64_voices": eng.pipeline_depth,
                "bytes_per_voice_sample": bytes_per_block / float(V * block),
                "note": "path is VALU/transcendental-bound (SURVEY F8): HBM is touched once per block; "
                        "see valu_issue for the bound that applies",
                # The limiter: instruction issue / dependent-instruction latency.  CDNA4 SIMDs are
                # 32 wide: a wave64 VALU instruction issues in 2 cycles (1024 SIMDs x 2.4 GHz / 2 = the 157 TF
                # vector peak); the measured ceiling for scalar f32 streams is ~3.05 cycles (103 TF,
                # MI355X_MICROARCH.md; scripts/pk_probe: 3.2 with 8 waves per SIMD).  achieved = SQ_INSTS_VALU per
                # launch (committed PMC pass of the SAME kernel hash) / the kernel duration measured in this run.
                "valu_issue": None if not (pmc_valu and kern_ms > 0) else {
                    "achieved": pmc_valu / (kern_ms * 1e-3) / 1e9,
                    "peak": 1024 * 2.4 / 2.0,
                    "measured_ceiling": 1024 * 2.4 / 3.05,
                    "unit": "G wave-instructions/s",
                    "frac": pmc_valu / (kern_ms * 1e-3) / 1e9 / (1024 * 2.4 / 2.0),
                    "frac_of_measured_ceiling": pmc_valu / (kern_ms * 1e-3) / 1e9 / (1024 * 2.4 / 3.05),
                    "valu_wave_inst_per_64_voices_per_frame": pmc_valu / (V / 64.0 * block * blocks_per_launch),
                    "source": pmc_src,
                },
            },
        }
        if midi is not None:
            line["config"]["midi_messages_per_block"] = args.midi_live
            line["event_stats"] = eng.event_stats
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(block, oscen_amd.SYNTH_SEED, K * block, span)
        else:
            line["cpu_baseline"] = None
    else:
        line = None
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes a version banner through C stdio: push it out first, so that the JSON is the LAST line of stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
