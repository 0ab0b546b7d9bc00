"""The N>1 path on CPU: two gloo ranks shard a voice bank by contiguous global voice ranges,
evaluate their shards independently (no data-path collective) and combine the per-rank mix buses
with ONE reduce -- exactly what bench.py does with RCCL.  The per-rank "engine" here is the CPU
oracle (test infrastructure): this test is about the sharding / reduce plumbing of
oscen_amd.distributed, not about the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, total_voices, blocks, frames, port, out_path):
    sys.path.insert(0, ROOT)
    import oscen_amd
    from oscen_amd import distributed as ogd
    from tests import oracle_lib as ol

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ogd.shard_range(rank, world, total_voices)
    plans = oscen_amd.note_plans(hi - lo, first_voice=lo)      # global voice ids keep their note streams
    bank = ol.Bank(ol.BANK_FM, hi - lo, 48000.0)
    for i in range(hi - lo):
        bank.set_voice_frequency(i, float(plans["frequency"][i]))
    bus = torch.zeros((blocks, frames), dtype=torch.float32)
    for b in range(blocks):
        for i in range(hi - lo):
            on = int(plans["on_frame"][i])
            if b * frames <= on < (b + 1) * frames:
                bank.push_event(i, on - b * frames, ol.EV_GATE, float(plans["gate"][i]))
        out, _ = bank.process_block(frames)
        bus[b] = torch.from_numpy(out[:, 0].copy())
    ogd.reduce_bus(bus, dst=0)
    if rank == 0:
        np.save(out_path, bus.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_bus_refuses_a_post_mix_node_that_is_not_linear():
    """the torch-level path sums AFTER each rank's post-mix stage: right for Tremolo (out = in * pan), wrong for anything
    else -- refused (og_cluster_* reduces the voice sums and runs the node once on the root)"""
    import types

    import pytest
    import torch

    from oscen_amd import distributed as ogd

    bus = torch.zeros((2, 8))
    for kind in (0, 1):
        assert ogd.reduce_bus(bus, engine=types.SimpleNamespace(post_mix_kind=kind)) is bus
    with pytest.raises(ValueError, match="not linear"):
        ogd.reduce_bus(bus, engine=types.SimpleNamespace(post_mix_kind=2))


def test_shard_ranges_partition_the_bank():
    from oscen_amd import distributed as ogd

    for total in (1, 7, 64, 65536, 2097152):
        for world in (1, 2, 3, 8):
            r = [ogd.shard_range(k, world, total) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))


def test_two_rank_voice_shards_reduce_to_the_single_process_bus(tmp_path):
    from tests import oracle_lib as ol
    import oscen_amd

    total, blocks, frames = 24, 3, 128
    out_path = str(tmp_path / "bus.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, total, blocks, frames, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    # single-process reference over all voices
    plans = oscen_amd.note_plans(total)
    bank = ol.Bank(ol.BANK_FM, total, 48000.0)
    for i in range(total):
        bank.set_voice_frequency(i, float(plans["frequency"][i]))
    ref = np.zeros((blocks, frames), dtype=np.float64)
    for b in range(blocks):
        for i in range(total):
            on = int(plans["on_frame"][i])
            if b * frames <= on < (b + 1) * frames:
                bank.push_event(i, on - b * frames, ol.EV_GATE, float(plans["gate"][i]))
        bank.process_block(frames)
        ref[b] = bank.last_bus_f64(frames)
    assert np.max(np.abs(ref)) > 0.01
    assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, float(np.max(np.abs(ref))))
