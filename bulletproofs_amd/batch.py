"""Multi-GPU batch layer (SURVEY.md §8e mode (i)): independent proofs are sharded across ranks,
one process per GPU; generators are replicated; the only exchange is gathering the verdict bytes.

No data-path collective: every rank verifies its own contiguous shard.  `gather_verdicts` uses
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""


def shard_range(nproofs, rank, world):
    """Contiguous, balanced shard [lo, hi) of proof indices owned by `rank`."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(nproofs, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_verdicts(local_verdicts, nproofs, device=None):
    """All-gather the per-rank verdict bytes into the full verdict list (same on every rank)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_verdicts)
    world, rank = dist.get_world_size(), dist.get_rank()
    width = (nproofs + world - 1) // world
    buf = torch.full((width,), 254, dtype=torch.uint8)
    lo, hi = shard_range(nproofs, rank, world)
    if hi - lo != len(local_verdicts):
        raise ValueError("local verdict count does not match this rank's shard")
    if hi > lo:
        buf[:hi - lo] = torch.tensor(list(local_verdicts), dtype=torch.uint8)
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        l, h = shard_range(nproofs, r, world)
        out.extend(parts[r][:h - l].cpu().tolist())
    return out


def verify_batch_sharded(verify_fn, proofs, proof_len, coms, com_len, nproofs, device=None):
    """Run `verify_fn(proof_bytes, com_bytes) -> [verdict codes]` on this rank's shard and gather.

    On the GPU box verify_fn is `lambda p, c: ctx.rangeproof_verify_batch(gens, n, m, p, c)`."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(nproofs, rank, world)
    local = verify_fn(proofs[lo * proof_len:hi * proof_len], coms[lo * com_len:hi * com_len]) if hi > lo else []
    return gather_verdicts(local, nproofs, device)


def msm_sharded(n, partial_fn, combine_fn, device=None):
    """One large MSM across the ranks (SURVEY.md §8e mode (ii)): points are sharded N/G per rank,
    every rank runs the bucket phase of the Pippenger pipeline on its shard (all windows) and produces
    its per-window partial sums; the partials (W x 144 bytes per rank) are all-gathered - EC addition
    is not a reduction operator, so gather + local combine, no all-reduce - and every rank adds the G
    partials per window and runs ONE Horner combine.
      partial_fn(lo, hi)      -> bytes     rank-local partial over terms [lo, hi)
                                           (GPU: ctx.msm_partial_device(d_points, d_scalars, hi - lo, c))
      combine_fn(parts, G)    -> 64 bytes  (GPU: ctx.msm_combine(parts, G, c))
    Every rank must return partials of the same length (same window width c: take
    c = bp_msm_auto_window(largest shard)).  Returns the same 64 bytes on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n, rank, world)
    part = partial_fn(lo, hi)
    if world == 1:
        return combine_fn(part, 1)
    buf = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allp = b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts)
    return combine_fn(allp, world)


def largest_shard(n, world):
    return (n + world - 1) // world
