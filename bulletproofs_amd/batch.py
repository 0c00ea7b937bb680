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


def msm_sharded_tensors(n, psize, partial_into, combine_from, device=None, after_partial=None, before_combine=None):
    """One large MSM across the ranks (SURVEY.md 8e mode (ii)): points are sharded N/G per rank, every rank runs the
    bucket phase of the Pippenger pipeline on its shard (all windows) into ITS per-window partial sums (psize =
    bp_msm_partial_bytes(c) bytes, the same window width c on every rank: c = bp_msm_auto_window(largest shard)); the
    partials are all-gathered - EC addition is not a reduction operator, so gather + local combine, no all-reduce - and
    every rank adds the G partials per window and runs ONE Horner combine.  The buffers are torch uint8 tensors on
    `device` and never leave it: on the GPU box they are device memory and the gather is RCCL over xGMI,
        partial_into(lo, hi, t)   ctx.msm_partial_device_out(d_points, d_scalars, hi - lo, c, t.data_ptr())   (asynchronous)
        after_partial()           ctx.record_event(ev) ; torch.cuda.current_stream().wait_event(ev)   - the collective is
                                  queued right behind the bucket-reduction kernels, no host synchronisation
        before_combine()          ev2.record() ; ctx.wait_event(ev2)
        combine_from(t, G)        ctx.msm_combine_device(t.data_ptr(), G, c)   -> 64 bytes
    in the CPU tests (gloo) they are host tensors filled / read by stand-ins.  Returns the same 64 bytes on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n, rank, world)
    mine = torch.empty(psize, dtype=torch.uint8, device=device)
    partial_into(lo, hi, mine)
    if world == 1:
        if after_partial:
            after_partial()
        if before_combine:
            before_combine()
        return combine_from(mine, 1)
    allp = torch.empty(world * psize, dtype=torch.uint8, device=device)
    if after_partial:
        after_partial()
    try:
        dist.all_gather_into_tensor(allp, mine)
    except (RuntimeError, NotImplementedError):  # a backend without the flat form
        dist.all_gather(list(allp.view(world, psize).unbind(0)), mine)
    if before_combine:
        before_combine()
    return combine_from(allp, world)


def msm_sharded(n, partial_fn, combine_fn, device=None):
    """The same exchange with byte strings at the boundary (host-pointer entry points bp_msm_partial / bp_msm_combine):
      partial_fn(lo, hi)      -> bytes     rank-local partial over terms [lo, hi)
      combine_fn(parts, G)    -> 64 bytes
    A wrapper over msm_sharded_tensors - one code path for both forms."""
    import torch

    def combine_from(t, g):
        return combine_fn(bytes(t.cpu().numpy().tobytes()), g)

    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n, rank, world)
    # the partial's size is only known once it exists: produce it first, then run the tensor form with a cached copy
    part = partial_fn(lo, hi)
    return msm_sharded_tensors(n, len(part), lambda lo_, hi_, t: t.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8)), combine_from, device)


def largest_shard(n, world):
    return (n + world - 1) // world
