#!/usr/bin/env python
"""Benchmark of the hot path: voices*samples/sec of the fm-synth voice bank.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one 256-frame block of the fm_voice graph over this rank's voice
shard (BASELINE.json configs[1]: 65 536 voices per GPU, 48 kHz, f32, synthetic
note streams already resident in HBM).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def pmc_traffic(voices, block, graph):
    """HBM bytes per launch of the voice kernel from the committed rocprofv3 PMC passes
    (profiles/*_summary.json, written by scripts/prof_summary.py): FETCH_SIZE and WRITE_SIZE are
    collected in separate --pmc runs of this same command, so they cannot be measured in-process."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if (d.get("voices") == voices and d.get("frames") == block and d.get("graph", "fm_voice") == graph
                and "hbm_traffic" in d):
            best = (path, d)
    if not best:
        return None, None, None
    valu = best[1].get("pmc_voice_kernel", {}).get("SQ_INSTS_VALU", {}).get("avg_per_dispatch")
    return best[1]["hbm_traffic"]["total_bytes_corrected"], os.path.relpath(best[0], ROOT), valu


def cpu_baseline(block, seed):
    """Time the CPU oracle (scalar C port of the reference path) on the host cores, bounded sample."""
    import ctypes as C

    from tests import oracle_lib as ol

    lib = ol.load()
    cores = os.cpu_count() or 1
    cs = C.c_double()
    # calibrate on a tiny bank, then size the sample for ~12 s of CPU work on all cores
    probe_v, probe_f = 64 * min(cores, 8), 2048
    t = lib.oo_bank_bench(ol.BANK_FM, probe_v, probe_f, block, cores, seed, C.byref(cs))
    rate = probe_v * probe_f / max(t, 1e-6)
    frames = 12000  # first 0.25 s of the note streams (attack/decay + first note-offs)
    voices = int(max(cores, min(65536, rate * 12.0 / frames)))
    voices = max(cores, (voices // cores) * cores)
    t = lib.oo_bank_bench(ol.BANK_FM, voices, frames, block, cores, seed, C.byref(cs))
    return {
        "value": voices * frames / t,
        "unit": "voices*samples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d voices x %d frames (block %d) of the same synthetic fm-synth note streams, "
                  "C oracle, %d threads, %.1f s" % (voices, frames, block, cores, t),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=188)   # 188 x 256 frames = 1 s of audio
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--voices-per-gpu", type=int, default=65536)
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--graph", default="fm_voice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # plumbing checks of the multi-rank path on a 1-GPU box (not a benchmark configuration):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--single-device", action="store_true", help="map every rank onto GPU 0")
    args = ap.parse_args()

    import numpy as np
    import torch

    import oscen_amd
    from oscen_amd import distributed as ogd

    rank, local_rank, world_size = ogd.world()
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, rank=rank, world_size=world_size)

    V = args.voices_per_gpu
    total_voices = V * world_size
    lo, hi = ogd.shard_range(rank, world_size, total_voices)
    block, K, W = args.block, args.steps, args.warmup
    total_frames = (K + W) * block

    eng = oscen_amd.Engine(args.graph, hi - lo, device=local_rank, sample_rate=48000.0)
    plans = oscen_amd.note_plans(hi - lo, first_voice=lo)  # global voice ids keep their note streams
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total_frames)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    ch = eng.channels
    bus = torch.zeros((K + W, block * ch), dtype=torch.float32, device="cuda")
    base = bus.data_ptr()

    def step(i):
        eng.process_block_async(block, base + i * block * ch * 4)

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

    for i in range(W):
        step(i)
    if dist is not None:  # communicator set-up (lazy in RCCL) must not land in the timed region
        reduce_bus(bus[:W] if W else torch.zeros((1, block * ch), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    if dist is not None:
        barrier()
    torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    if dist is not None:
        reduce_bus(bus[W:])  # ONE RCCL reduce of the [K, block] mix bus over xGMI
    torch.cuda.synchronize()
    if dist is not None:
        barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms, n_launch = eng.kernel_time_ms()
    eng.enable_kernel_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        mix = bus[W:].float().cpu().numpy()
        assert np.isfinite(mix).all() and np.abs(mix).max() > 0.0, "bus is silent or non-finite"
        value = total_voices * K * block / elapsed
        words = eng.state_words_per_voice
        lanes = eng.voices_per_wave * eng.lanes_per_voice
        # algorithmic HBM bytes of one launch: state planes read once + written once,
        # the two event-cursor words read per voice, one partial-bus row written per workgroup
        n_wg = (V * eng.lanes_per_voice + lanes - 1) // lanes
        bytes_per_launch = V * (2 * 4 * words + 8) + n_wg * block * 4
        # graphs with a Delay: every voice-sample reads one and writes one 4-byte slot of its HBM ring
        bytes_per_launch += V * block * 8 * {"echo_voice": 1}.get(args.graph, 0)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        pmc_bytes, pmc_src, pmc_valu = pmc_traffic(V, block, args.graph)
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
                            "synthetic note streams splitmix64(0x05CE2026 ^ voice) resident in HBM; "
                            "mix bus reduced once per run over RCCL" % (V, block),
                "graph": args.graph,
                "voices_per_gpu": V,
                "total_voices": total_voices,
                "block": block,
                "sample_rate": 48000,
                "parallelism": "voice-shard x%d" % world_size,
            },
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
                "kernel_ms_avg": kern_ms,
                "kernel_launches": n_launch,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "voices_per_wave": lanes,
                "pipeline_waves_per_64_voices": eng.pipeline_depth,
                "bytes_per_voice_sample": bytes_per_launch / float(V * block),
                "note": "path is VALU/transcendental-bound (SURVEY F8): HBM is touched once per block; "
                        "see valu_issue for the bound that applies",
                # The limiter: instruction issue / dependent-instruction latency.  CDNA4 SIMDs are
                # 32 wide: a wave64 VALU instruction issues in 2 cycles (1024 SIMDs x 2.4 GHz / 2 = the 157 TF
                # vector peak); the measured ceiling for scalar f32 streams is ~3.05 cycles (103 TF,
                # MI355X_MICROARCH.md; scripts/pk_probe: 3.2 with 8 waves per SIMD).  achieved = SQ_INSTS_VALU per
                # launch (committed PMC pass) / the kernel duration measured in this run.
                "valu_issue": None if not (pmc_valu and kern_ms > 0) else {
                    "achieved": pmc_valu / (kern_ms * 1e-3) / 1e9,
                    "peak": 1024 * 2.4 / 2.0,
                    "measured_ceiling": 1024 * 2.4 / 3.05,
                    "unit": "G wave-instructions/s",
                    "frac": pmc_valu / (kern_ms * 1e-3) / 1e9 / (1024 * 2.4 / 2.0),
                    "frac_of_measured_ceiling": pmc_valu / (kern_ms * 1e-3) / 1e9 / (1024 * 2.4 / 3.05),
                    "valu_wave_inst_per_64_voices_per_frame": pmc_valu / (V / 64.0 * block),
                    "source": pmc_src,
                },
            },
        }
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(block, oscen_amd.SYNTH_SEED)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
