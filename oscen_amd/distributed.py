"""Multi-GPU sharding of a voice bank (one process per GPU, torch.distributed).

Voices are independent; the only cross-voice operation on the path is the sum
onto the mix bus (`voices.audio_out -> audio_out`,
oscen-graph-compiler/src/codegen/emit_node.rs:463-466).  So the bank is cut
into contiguous voice ranges, one per rank, no data-path collective runs while
the voices are evaluated, and the per-rank partial buses of a whole batch of
blocks are combined by ONE reduce (RCCL over xGMI when the backend is "nccl";
a [blocks x frames] f32 message is latency-bound, so it is batched rather than
issued per 1 KB block).  The fm-synth bus is mono and the hosts duplicate it
to L/R after the sum (examples/fm-synth/src/lib.rs:269-274).  In this torch-level path an engine with a post-mix
node (the e-piano's stereo Tremolo) applies it per rank before the reduce -- the node is linear in its input, so the
reduced bus is the same; the C-ABI cluster (og_cluster_*) reduces the mono sums and runs it once on the root.
"""
import os


def world():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(rank, world_size, total_voices):
    """Contiguous voice range [lo, hi) owned by `rank` (global voice ids keep their note streams)."""
    lo = (total_voices * rank) // world_size
    hi = (total_voices * (rank + 1)) // world_size
    return lo, hi


def reduce_bus(bus, dst=0, group=None):
    """Sum the per-rank partial mix buses onto `dst` with a single collective.

    `bus` is a torch tensor [blocks, frames(, channels)] on the rank's device
    (CUDA tensor -> RCCL, CPU tensor -> gloo).  In place; only `dst` holds the
    full mix afterwards.
    """
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return bus
