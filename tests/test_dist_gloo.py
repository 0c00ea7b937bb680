"""N>1 path on CPU: world_size-2 gloo run of the proof-sharding batch layer
(bulletproofs_amd/batch.py).  The per-shard verifier is the oracle here (no GPU in this container);
on the GPU box the same layer wraps ctx.rangeproof_verify_batch."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bulletproofs_amd import batch
    from oracle import curv, ec, encoding
    from _golden import bp_cases, VERDICT_CODE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = bp_cases()[0]  # n=8 m=1
    n, m = case["n"], case["m"]
    g_vec, h_vec, G, H = curv.bp_generators(n, m)
    items = [(case["proof"], case["commitments"], "Ok")] + [(t["proof"], t["commitments"], t["verdict"]) for t in case["tampered"]]
    items = [it for it in items if it[2] != "Malformed"][:5]
    plen = len(bytes.fromhex(items[0][0]))
    proofs = b"".join(bytes.fromhex(p) for p, _, _ in items)
    coms = b"".join(bytes.fromhex(c[0]) for _, c, _ in items)
    want = [VERDICT_CODE[v] for _, _, v in items]

    def verify_fn(pb, cb):
        out = []
        for i in range(len(pb) // plen):
            p = encoding.decode_bp(pb[i * plen:(i + 1) * plen])
            out.append(VERDICT_CODE[p.aggregated_verify(g_vec, h_vec, G, H, [ec.dec64(cb[i * 64:(i + 1) * 64])], n)])
        return out

    got = batch.verify_batch_sharded(verify_fn, proofs, plen, coms, 64, len(items))
    lo, hi = batch.shard_range(len(items), rank, world)
    # point-sharded MSM with an all-gather of the per-rank partial sums
    import random
    rng = random.Random(5)
    pts = [ec.mul(ec.G, rng.randrange(1, ec.Q)) for _ in range(11)]
    ks = [rng.randrange(ec.Q) for _ in range(11)]

    # the product's byte format across the processes: every rank's partial is a bp_msm_partial_bytes(c)-sized buffer of W XYZZ
    # window sums (tests/_partial.py; tied to the library by tests/test_gpu_msm.py), all-gathered by msm_sharded exactly as
    # bench.py --workload msm does with ctx.msm_partial_device / ctx.msm_combine; the oracle stands in for the two GPU calls
    import bulletproofs_amd as bpa
    import _partial
    c = 7
    ks = [k >> 128 for k in ks]  # < 2^128: unsigned c-bit digits fit the W = (129 + c) // c windows of the format
    psize = bpa.lib().bp_msm_partial_bytes(c)
    assert psize == _partial.windows(c) * _partial.XYZZ_BYTES

    def partial_fn(lo_, hi_):
        part = _partial.encode_partial(_partial.unsigned_window_sums(pts[lo_:hi_], ks[lo_:hi_], c), c)
        assert len(part) == psize
        return part

    def combine_fn(parts, g):
        assert len(parts) == g * psize
        return ec.enc64(_partial.combine(parts, g, c))

    msm_got = batch.msm_sharded(len(pts), partial_fn, combine_fn)
    msm_ok = msm_got == ec.enc64(ec.msm_naive(pts, ks, None))
    q.put((rank, got == want and msm_ok, hi - lo))
    dist.destroy_process_group()


def test_shard_range():
    from bulletproofs_amd import batch
    for n in [0, 1, 5, 8, 4096, 4099]:
        for w in [1, 2, 3, 8]:
            cover = []
            for r in range(w):
                lo, hi = batch.shard_range(n, r, w)
                cover.extend(range(lo, hi))
                assert hi - lo in (n // w, n // w + 1)
            assert cover == list(range(n))


@pytest.mark.timeout(300)
def test_sharded_verify_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res)
    assert sum(r[2] for r in res) == 5
