"""GPU parity: bp_msm (HIP Pippenger, through the C ABI) == oracle fold idiom, bit for bit."""
import hashlib
import random

import pytest

from oracle import ec

pytestmark = pytest.mark.gpu


def _rand_points(rng, n, oracle_c):
    ks = [rng.randrange(1, ec.Q) for _ in range(n)]
    return oracle_c.mul_batch([ec.G] * n, ks)


def _pack(points, scalars):
    return b"".join(ec.enc64(p) for p in points), b"".join((k % 2**256).to_bytes(32, "big") for k in scalars)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 63, 64, 65, 127, 128, 129, 146, 257, 700])  # 64|65: one-workgroup path ends; 127|128: sorted path begins
def test_msm_matches_oracle(ctx, oracle_c, n):
    rng = random.Random(1000 + n)
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    pb, sb = _pack(pts, ks)
    init = oracle_c.mul(ec.G, 777)
    want = oracle_c.msm_naive(pts, ks, init)
    got = ctx.msm(pb, sb, ec.enc64(init))
    assert got == ec.enc64(want)
    got0 = ctx.msm(pb, sb, None)
    assert got0 == ec.enc64(oracle_c.msm_naive(pts, ks, None))


def test_msm_edge_scalars_and_points(ctx, oracle_c):
    """zero scalars, q-1, unreduced scalars >= q, infinity points, duplicate and opposite points
    (the reference recipe makes h_vec[i] == g_vec[i+n]: range_proof.rs:837-841)."""
    rng = random.Random(5)
    base = _rand_points(rng, 8, oracle_c)
    pts = base + [base[0], ec.neg(base[1]), None, base[2], base[2]]
    ks = [0, 1, ec.Q - 1, ec.Q, ec.Q + 5, 2**256 - 1, 2**255, 12345, 5, 7, 99, 2**200 + 1, ec.Q - (2**200 + 1)]
    pb, sb = _pack(pts, ks)
    want = oracle_c.msm_naive(pts, [k % ec.Q for k in ks], None)
    assert ctx.msm(pb, sb, None) == ec.enc64(want)
    # same point with equal scalars lands in the same bucket -> doubling inside a bucket
    pts2 = [base[3]] * 6 + [ec.neg(base[3])] * 6
    ks2 = [1234567] * 12
    assert ctx.msm(*_pack(pts2, ks2), None) == b"\0" * 64
    pts3 = [base[3]] * 5
    assert ctx.msm(*_pack(pts3, [3] * 5), None) == ec.enc64(oracle_c.mul(base[3], 15))


@pytest.mark.parametrize("c", [2, 3, 5, 8, 9, 10, 11])
def test_msm_all_window_widths(ctx, oracle_c, c):
    rng = random.Random(c)
    n = 300
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    want = ec.enc64(oracle_c.msm_naive(pts, ks, None))
    ctx.set_window(c)
    try:
        assert ctx.msm(*_pack(pts, ks), None) == want
    finally:
        ctx.set_window(0)


@pytest.mark.parametrize("c,n", [(2, 5), (3, 40), (6, 300), (9, 1000), (12, 3000), (15, 3000), (0, 1)])
def test_msm_sorted_path_matches_oracle(ctx, oracle_c, c, n):
    """The large-N path (global counting sort + balanced segment accumulation, msm_sorted.hip) forced at
    sizes the oracle finishes: every window width class, ragged last segments, single point."""
    rng = random.Random(3000 + 16 * c + n)
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    init = oracle_c.mul(ec.G, 31337)
    want = ec.enc64(oracle_c.msm_naive(pts, ks, init))
    ctx.set_path(2)
    ctx.set_window(c)
    try:
        assert ctx.msm(*_pack(pts, ks), ec.enc64(init)) == want
    finally:
        ctx.set_window(0)
        ctx.set_path(0)


@pytest.mark.parametrize("c,n", [(2, 5), (4, 40), (7, 300), (11, 1500), (12, 3000), (15, 3000), (0, 1), (0, 2600)])
def test_msm_two_pass_sort_matches_oracle(ctx, oracle_c, c, n):
    """The two-pass bucket sort of the sorted path (partition pass + in-LDS pass, msm_sorted.hip k_srt2_*; automatic from
    2^14 terms) forced at sizes the oracle finishes: one partition / many, one bin per partition / 1024, empty partitions."""
    rng = random.Random(5000 + 16 * c + n)
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    if n > 100:  # small and repeated scalars: empty windows, partitions larger than the LDS image
        for i in range(0, n, 3):
            ks[i] = rng.choice([1, 2, ec.Q - 1, 0x1234567, rng.randrange(1 << 30)])
    init = oracle_c.mul(ec.G, 4711)
    want = ec.enc64(oracle_c.msm_naive(pts, ks, init))
    ctx.set_path(3)
    ctx.set_window(c)
    try:
        assert ctx.msm(*_pack(pts, ks), ec.enc64(init)) == want
    finally:
        ctx.set_window(0)
        ctx.set_path(0)


def test_msm_sorted_path_skewed_buckets(ctx, oracle_c):
    """Bucket sizes far from uniform: the +-1 / 0 / small scalars of the prover's A commitment
    (range_proof.rs:93-107), repeated and opposite points, infinity, and buckets spanning many segments."""
    rng = random.Random(77)
    base = _rand_points(rng, 16, oracle_c)
    n = 2500
    pts = [base[rng.randrange(16)] for _ in range(n)]
    pts[7] = None
    pts[8] = ec.neg(pts[9])
    ks = [rng.choice([0, 1, ec.Q - 1, 2, 3, 1 << 128, ec.Q - (1 << 128)]) for _ in range(n)]
    ks[9] = ks[8] = 5
    want = ec.enc64(oracle_c.msm_naive(pts, ks, None))
    for path in (1, 2, 3):
        ctx.set_path(path)
        try:
            assert ctx.msm(*_pack(pts, ks), None) == want, path
        finally:
            ctx.set_path(0)


def test_msm_rejects_off_curve_point(ctx):
    import bulletproofs_amd as bpa
    bad = (1).to_bytes(32, "big") + (1).to_bytes(32, "big")
    with pytest.raises(bpa.BpError) as e:
        ctx.msm(bad, (3).to_bytes(32, "big"), None)
    assert e.value.code == -2


def test_msm_linearity_large(ctx, oracle_c):
    """Size-independent property at a size the oracle cannot finish quickly:
    MSM(P, a) + MSM(P, b) == MSM(P, a+b), and sum k_i (r_i G) == (sum k_i r_i) G."""
    rng = random.Random(99)
    n = 20000
    rs = [rng.randrange(1, ec.Q) for _ in range(64)]
    base = oracle_c.mul_batch([ec.G] * 64, rs)
    idx = [rng.randrange(64) for _ in range(n)]
    pts = [base[i] for i in idx]
    a = [rng.randrange(ec.Q) for _ in range(n)]
    pb, sb = _pack(pts, a)
    want_scalar = sum(a[i] * rs[idx[i]] for i in range(n)) % ec.Q
    want = ec.enc64(oracle_c.mul(ec.G, want_scalar))
    assert ctx.msm(pb, sb, None) == want          # automatic path (sorted at this size)
    for path in (1, 2):
        ctx.set_path(path)
        try:
            assert ctx.msm(pb, sb, None) == want, path
        finally:
            ctx.set_path(0)


def test_msm_grouped_pipeline_skewed_scalars(ctx, oracle_c):
    """The grouped, multi-stream pipeline of the sorted path (automatic from 2^19 points, forced here: window groups top-down, bucket reduction
    and Horner chain beside the next group's accumulation, per-window bucket grid) on scalar distributions that
    leave windows empty, fill single buckets, or fill only the lowest buckets of a window:
    sum k_i (r_j G) == (sum k_i r_j) G."""
    rng = random.Random(4242)
    n = 70000
    rs = [rng.randrange(1, ec.Q) for _ in range(64)]
    base = oracle_c.mul_batch([ec.G] * 64, rs)
    idx = [rng.randrange(64) for _ in range(n)]
    pts = [base[i] for i in idx]
    mixes = {
        "pm1": lambda: rng.choice([0, 1, ec.Q - 1]),
        "small": lambda: rng.randrange(1 << 20),
        "u64": lambda: rng.randrange(1 << 64),
        "mixed": lambda: rng.choice([rng.randrange(ec.Q), rng.randrange(1 << 40), 1, ec.Q - 2, 0, 1 << 200]),
        "same": lambda: 0x1234567,
    }
    for name, draw in mixes.items():
        ks = [draw() for _ in range(n)]
        pb, sb = _pack(pts, ks)
        want_scalar = sum(ks[i] * rs[idx[i]] for i in range(n)) % ec.Q
        want = ec.enc64(oracle_c.mul(ec.G, want_scalar)) if want_scalar else bytes(64)
        assert ctx.msm(pb, sb, None) == want, name  # automatic: one group at this size
        ctx.set_msm_pipeline(2)  # the window groups (automatic from 2^19 points)
        assert ctx.msm(pb, sb, None) == want, name
        if name in ("pm1", "mixed"):  # with an init point: the final kernel adds S_0, the doubled upper combination AND init
            init = oracle_c.mul(ec.G, 0xC0FFEE)
            want_i = (want_scalar + 0xC0FFEE) % ec.Q
            assert ctx.msm(pb, sb, ec.enc64(init)) == (ec.enc64(oracle_c.mul(ec.G, want_i)) if want_i else bytes(64)), name
        ctx.set_msm_pipeline(1)  # one window group, no side streams (bp_msm_set_pipeline): same bytes
        try:
            assert ctx.msm(pb, sb, None) == want, name
        finally:
            ctx.set_msm_pipeline(0)


@pytest.mark.timeout(150)
def test_msm_grouped_pipeline_with_other_streams_in_the_process(ctx, oracle_c):
    """The multi-stream pipeline next to other users of the process's four hardware queues: a second context that has
    run work, and the batch prover's lane (which runs on the context's side stream).  Regression test: with a
    high-priority reduction stream the cross-stream waits of the pipeline never completed once any further stream
    existed, and the call hung (scripts/experiments/prove_then_msm.py)."""
    import bulletproofs_amd as bpa
    from oracle import tape
    rng = random.Random(777)
    n = 70000
    rs = [rng.randrange(1, ec.Q) for _ in range(64)]
    base = oracle_c.mul_batch([ec.G] * 64, rs)
    idx = [rng.randrange(64) for _ in range(n)]
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    pb, sb = _pack([base[i] for i in idx], ks)
    want = ec.enc64(oracle_c.mul(ec.G, sum(ks[i] * rs[idx[i]] for i in range(n)) % ec.Q))
    other = bpa.Context(0)
    g, h, G, H = bpa.generate_bases(0, bpa.KZEN_SEED, 8, 1)
    gens = bpa.Gens(ctx, g, h, G, H, 4)
    try:
        assert len(other.fixed_base_mul_batch((7).to_bytes(32, "big") * 64)) == 64 * 64
        count, ntape = 1100, 2 + 2 * 8 + 2  # more than one chunk: both prover lanes run
        tb = [tape.make_tape_bytes("msm-lanes-%d" % i, 1 + ntape) for i in range(count)]
        proofs = ctx.rangeproof_prove_batch(gens, 8, 1, b"".join((i % 256).to_bytes(32, "big") for i in range(count)),
                                            b"".join(t[:32] for t in tb), b"".join(t[32:] for t in tb), 32 * ntape)
        assert len(proofs) == count
        ctx.set_msm_pipeline(2)  # the window groups at this size
        other.set_msm_pipeline(2)
        for _ in range(2):
            assert ctx.msm(pb, sb, None) == want
        assert other.msm(pb, sb, None) == want  # and the other context's own pipeline (a seventh stream)
        assert ctx.msm(pb, sb, None) == want
    finally:
        ctx.set_msm_pipeline(0)
        gens.close()
        other.close()


def test_msm_full_size_2pow20(ctx, oracle_c):
    """BASELINE.json configs[4] size (2^20 terms): sum k_i (r_i G) == (sum k_i r_i) G with 2^20 distinct
    points r_i G made on the GPU (fixed-base batch) and independent 256-bit scalars."""
    import numpy as np
    n = 1 << 20
    rng = np.random.default_rng(2020)
    rb = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    kb = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    rb[:, 0] &= 0x7F  # r_i < 2^255 < q: no reduction needed on the checking side
    pts = b"".join(ctx.fixed_base_mul_batch(rb[i:i + 65536].tobytes()) for i in range(0, n, 65536))
    got = ctx.msm(pts, kb.tobytes(), None)
    acc = 0
    rbb, kbb = rb.tobytes(), kb.tobytes()
    for i in range(n):
        acc += int.from_bytes(rbb[32 * i:32 * i + 32], "big") * int.from_bytes(kbb[32 * i:32 * i + 32], "big")
    assert got == ec.enc64(oracle_c.mul(ec.G, acc % ec.Q))


def test_msm_2pow21_two_group_plan(ctx, oracle_c):
    """From 2^21 terms the window groups are W - 1 + 1 (capi.cpp plan_window_groups): the same size-independent check as
    above on 2^21 terms made of 2^16 distinct points (each used 32 times with independent scalars)."""
    import numpy as np
    n, nd = 1 << 21, 1 << 16
    rng = np.random.default_rng(2121)
    rb = rng.integers(0, 256, size=(nd, 32), dtype=np.uint8)
    kb = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    rb[:, 0] &= 0x7F
    base = ctx.fixed_base_mul_batch(rb.tobytes())
    got = ctx.msm(base * (n // nd), kb.tobytes(), None)
    rs = [int.from_bytes(rb[i].tobytes(), "big") for i in range(nd)]
    kbb = kb.tobytes()
    acc = 0
    for i in range(n):
        acc += rs[i % nd] * int.from_bytes(kbb[32 * i:32 * i + 32], "big")
    assert got == ec.enc64(oracle_c.mul(ec.G, acc % ec.Q))


def test_msm_batch_shared_generators(ctx, oracle_c):
    """bp_msm_batch: jobs sharing the generator prefix (fixed-base table) plus job-specific points.
    Checked term by term against the oracle fold idiom; includes a job without generator scalars, a job
    without extras, zero / unreduced scalars, an infinity extra point and an empty job."""
    import bulletproofs_amd as bpa
    from oracle import curv
    rng = random.Random(4242)
    g_vec, h_vec, G, H = curv.bp_generators(8, 1)
    gens_pts = g_vec + h_vec + [G, H]
    gens = bpa.Gens(ctx, b"".join(ec.enc64(p) for p in g_vec), b"".join(ec.enc64(p) for p in h_vec), ec.enc64(G), ec.enc64(H), 6)
    extra_pool = _rand_points(rng, 12, oracle_c)
    jobs, want = [], []
    for j in range(21):
        gs = [rng.randrange(ec.Q) for _ in gens_pts]
        if j == 3:
            gs[0], gs[5] = 0, 2**256 - 1
        ne = [0, 1, 5, 12][j % 4]
        xp = [extra_pool[rng.randrange(12)] for _ in range(ne)]
        xs = [rng.randrange(2**256) for _ in range(ne)]
        if j == 6 and ne:
            xp[0] = None
        use_g = j != 5
        if j == 8:
            use_g, xp, xs = False, [], []
        acc = oracle_c.msm_naive(gens_pts, [s % ec.Q for s in gs], None) if use_g else None
        if xp:
            acc = oracle_c.msm_naive(xp, [s % ec.Q for s in xs], acc)
        want.append(ec.enc64(acc))
        pb, sb = _pack(xp, xs)
        jobs.append((b"".join((s % 2**256).to_bytes(32, "big") for s in gs) if use_g else None, pb, sb))
    try:
        got = ctx.msm_batch(gens, jobs)
        assert [got[64 * i:64 * i + 64] for i in range(len(jobs))] == want
        assert ctx.msm_batch(gens, []) == b""
    finally:
        gens.close()


@pytest.mark.parametrize("shards", [[700], [300, 0, 399, 1], [1, 1, 1]])
def test_msm_window_partials_combine(ctx, oracle_c, shards):
    """Multi-GPU exchange step on one GPU: per-shard window partials (bp_msm_partial) combined by
    bp_msm_combine equal the single MSM over the concatenated inputs, bit for bit (uneven and empty shards)."""
    import bulletproofs_amd as bpa
    n = sum(shards)
    rng = random.Random(600 + n)
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(2**256) for _ in range(n)]
    init = oracle_c.mul(ec.G, 99)
    want = ec.enc64(oracle_c.msm_naive(pts, [k % ec.Q for k in ks], init))
    c = bpa.msm_auto_window(max(shards))
    parts, lo = b"", 0
    for sz in shards:
        parts += ctx.msm_partial(*_pack(pts[lo:lo + sz], ks[lo:lo + sz]), c)
        lo += sz
    assert ctx.msm_combine(parts, len(shards), c, ec.enc64(init)) == want


@pytest.mark.parametrize("shards", [[700], [300, 0, 399, 1]])
def test_msm_device_resident_exchange(ctx, oracle_c, shards):
    """bp_msm_partial_device_out / bp_msm_combine_device with the partials in device memory (torch tensors stand in for the
    all-gather's buffers) and the stream bridging of bulletproofs_amd.batch.msm_sharded_tensors (bp_ctx_record_event /
    bp_ctx_wait_event on torch events): equal to the single MSM, bit for bit; the host-pointer pair reads the same bytes."""
    import torch
    import bulletproofs_amd as bpa
    from bulletproofs_amd import batch
    n = sum(shards)
    rng = random.Random(900 + n)
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(2**256) for _ in range(n)]
    init = oracle_c.mul(ec.G, 4242)
    want = ec.enc64(oracle_c.msm_naive(pts, [k % ec.Q for k in ks], init))
    c = bpa.msm_auto_window(max(shards))
    psize = bpa.lib().bp_msm_partial_bytes(c)
    pb, sb = _pack(pts, ks)
    d_p = torch.frombuffer(bytearray(pb), dtype=torch.uint8).cuda()
    d_s = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    allp = torch.zeros(len(shards) * psize, dtype=torch.uint8, device="cuda")
    ev = torch.cuda.Event()
    ev.record()  # creates the handle
    torch.cuda.synchronize()
    lo = 0
    for g, sz in enumerate(shards):
        ctx.msm_partial_device_out(d_p.data_ptr() + 64 * lo, d_s.data_ptr() + 32 * lo, sz, c, allp.data_ptr() + g * psize)
        lo += sz
    ctx.record_event(ev.cuda_event)
    torch.cuda.current_stream().wait_event(ev)
    host_copy = bytes(allp.cpu().numpy().tobytes())  # on torch's stream, ordered behind the library's kernels by the event
    ev2 = torch.cuda.Event()
    ev2.record()
    ctx.wait_event(ev2.cuda_event)
    assert ctx.msm_combine_device(allp.data_ptr(), len(shards), c, ec.enc64(init)) == want
    assert ctx.msm_combine(host_copy, len(shards), c, ec.enc64(init)) == want
    # the batch layer's tensor form on one rank (no process group): the same entry points behind its hooks
    got = batch.msm_sharded_tensors(
        n, psize, lambda lo_, hi_, t: ctx.msm_partial_device_out(d_p.data_ptr() + 64 * lo_, d_s.data_ptr() + 32 * lo_, hi_ - lo_, c, t.data_ptr()),
        lambda t, g: ctx.msm_combine_device(t.data_ptr(), g, c, ec.enc64(init)), torch.device("cuda"),
        after_partial=lambda: (ctx.record_event(ev.cuda_event), torch.cuda.current_stream().wait_event(ev)),
        before_combine=lambda: (ev2.record(), ctx.wait_event(ev2.cuda_event)))
    assert got == want


def test_msm_partial_device_out_reports_off_curve_point_at_sync(ctx):
    import torch
    import bulletproofs_amd as bpa
    bad = (1).to_bytes(32, "big") + (1).to_bytes(32, "big")
    d_p = torch.frombuffer(bytearray(bad * 300), dtype=torch.uint8).cuda()
    d_s = torch.frombuffer(bytearray((3).to_bytes(32, "big") * 300), dtype=torch.uint8).cuda()
    c = bpa.msm_auto_window(300)
    out = torch.zeros(bpa.lib().bp_msm_partial_bytes(c), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.msm_partial_device_out(d_p.data_ptr(), d_s.data_ptr(), 300, c, out.data_ptr())
    with pytest.raises(bpa.BpError) as e:
        ctx.sync()
    assert e.value.code == -2
    ctx.sync()  # reported once


def test_msm_partial_byte_format(ctx, oracle_c):
    """The partial buffers that cross the process boundary (bench.py --workload msm, tests/test_dist_gloo.py): tests/_partial.py
    reads real bp_msm_partial output (Horner-combined by the oracle == bp_msm) and bp_msm_combine reads partials written by
    tests/_partial.py (== the oracle's MSM), so the gloo test's stand-in buffers are the product's format."""
    import bulletproofs_amd as bpa
    import _partial
    rng = random.Random(4242)
    n, c = 150, 7
    pts = _rand_points(rng, n, oracle_c)
    ks = [rng.randrange(ec.Q) for _ in range(n)]
    assert bpa.lib().bp_msm_partial_bytes(c) == _partial.windows(c) * _partial.XYZZ_BYTES
    want = oracle_c.msm_naive(pts, ks, None)
    parts = ctx.msm_partial(*_pack(pts[:90], ks[:90]), c) + ctx.msm_partial(*_pack(pts[90:], ks[90:]), c) + ctx.msm_partial(b"", b"", c)
    assert _partial.combine(parts, 3, c) == want
    assert ctx.msm_combine(parts, 3, c) == ec.enc64(want)
    small = [k >> 130 for k in ks]
    enc = _partial.encode_partial(_partial.unsigned_window_sums(pts[:70], small[:70], c), c) + \
        _partial.encode_partial(_partial.unsigned_window_sums(pts[70:], small[70:], c), c)
    assert ctx.msm_combine(enc, 2, c) == ec.enc64(oracle_c.msm_naive(pts, small, None))
