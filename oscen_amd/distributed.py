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
node applies it per rank BEFORE the reduce, which is only the same bus when the node is linear in its input (the e-piano's
stereo Tremolo: out = in * pan): reduce_bus() checks that (`engine=`, og_post_mix_kind) and refuses any other node -- the
C-ABI cluster (og_cluster_*) is the general path: it reduces the voice sums and runs the node once on the root.
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


def reduce_bus(bus, dst=0, group=None, engine=None):
    """Sum the per-rank partial mix buses onto `dst` with a single collective.

    `bus` is a torch tensor [blocks, frames(, channels)] on the rank's device
    (CUDA tensor -> RCCL, CPU tensor -> gloo).  In place; only `dst` holds the
    full mix afterwards.  `engine` (the Engine that rendered `bus`): its post-mix
    node, if any, has already run on this rank's partial bus -- summing after it is
    only right for a node that is linear in its input; anything else is refused.
    """
    import torch.distributed as dist

    if engine is not None and engine.post_mix_kind not in (0, 1):
        raise ValueError("reduce_bus: the engine's post-mix node is not linear in its input: sum the voice sums first and run the "
                         "node once on the destination (og_cluster_* does that)")

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return bus
