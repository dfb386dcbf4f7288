"""Multi-GPU bookkeeping of the throughput mode (one process per GPU, independent stereo pairs).

The path has no cross-pair coupling, so pairs are dealt round-robin to the
ranks and no data-path collective exists; the only communication is the barrier and the
max-over-ranks of the elapsed time that bench.py's contract asks for.
"""


def pairs_of_rank(n_pairs, world, rank):
    """Indices of the pairs rank `rank` processes (round-robin, every pair exactly once)."""
    return list(range(rank, n_pairs, world))


def job_rate(units_per_rank, elapsed_max_s):
    """Whole-job rate: all units processed by all ranks over the slowest rank's time."""
    return sum(units_per_rank) / elapsed_max_s


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (RCCL on GPUs, gloo in the CPU tests)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
