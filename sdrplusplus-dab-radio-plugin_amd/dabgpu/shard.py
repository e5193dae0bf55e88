"""Multi-GPU plan: independent DAB ensembles are sharded across ranks; there is no
data-path collective.  The reference is one Radio_Block per plugin instance
(/root/reference/src/dab_module.h:61) with no shared state between instances, so the only
cross-rank traffic is the run barrier and two tiny reductions of the report (RCCL on GPUs,
gloo in the CPU tests)."""


def ensembles_of_rank(n_total, world, rank):
    """Ensemble ids owned by `rank`: id % world == rank."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_total, world))


def reduce_report(dist, device, elapsed_s, frames_done, flags_ok):
    """max elapsed, sum frames, min of the correctness flags across ranks.
    `dist` is torch.distributed (or None for a single process)."""
    import torch
    if dist is None or not dist.is_initialized():
        return float(elapsed_s), int(frames_done), [bool(f) for f in flags_ok]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([frames_done], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    f = torch.tensor([int(bool(x)) for x in flags_ok], dtype=torch.int64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return float(t.item()), int(n.item()), [bool(x) for x in f.tolist()]


def gather_per_rank(dist, device, values):
    """Every rank's row of floats, in rank order, on every rank (one all_gather of len(values) doubles)."""
    import torch
    row = [float(v) for v in values]
    if dist is None or not dist.is_initialized():
        return [row]
    t = torch.tensor(row, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]
