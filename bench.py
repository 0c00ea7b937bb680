#!/usr/bin/env python3
"""bench.py — verified range proofs/s (secp256k1, n=64, m=1) on N MI355X GPUs.

A "step" is one pass of the hot path over one batch of synthetic proofs: the device-side
`RangeProof::aggregated_verify` (reference range_proof.rs:537-782) of `--proofs-per-gpu` independent
n=64, m=1 proofs on every rank (BASELINE.json configs[2], the configuration the metric
"verified range proofs/sec" is quoted on; weak scaling: per-GPU work fixed, no data-path collective,
proofs are sharded).  Inputs (proof bytes, commitments) are resident in HBM before the timed region.

Contract (one JSON line on rank 0): metric/value/unit/n_gpus/steps/warmup/ms_per_step/... plus
  "roofline"     — dominant kernel (k_bpv_msm) against the measured VALU INT32-MAD peak,
  "cpu_baseline" — the oracle's aggregated_verify ("port") timed on this host, rank 0, N=1 only.

Synthetic data: every proof of the batch is DISTINCT (the reference's benches prove fresh proofs,
benches/range_proof.rs:36-103): made at start-up on the GPU by bp_rangeproof_prove_batch (pinned to the
oracle's bytes by tests/test_gpu_proto.py) from SHAKE-256-derived secrets / blindings / tapes; every
16th proof of the batch is tampered (tx bit flipped) so both verdicts are exercised and checked.  A
batch of identical gather patterns would keep the 52.8 GiB generator table's gathers inside the caches;
`tiled_256` in the JSON is that easier case, measured once for the record.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself
(python -m torch.distributed.run ... bench.py --gpus N ...), before anything touches a GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic MAD32 (32x32->64 multiply-accumulates) per n=64 m=1 verification,
# fixed by the Pippenger model (146-point MSM, c=5) independent of this implementation's choices.
MAD32_PER_PROOF_MSM = 6.56e6        # the 146-term MSM  (dominant kernel k_bpv_msm)
MAD32_PER_PROOF_TOTAL = 6.60e6      # + the two fixed-base multiplications y*G, x_u*G
# Peak: measured with bench_micro/peak_mad32 on MI355X (profiles/r01_peak_mad32.jsonl):
# v_mad_u64_u32 sustains 31.2e12 lane-results/s at 8 waves/SIMD (SURVEY's assumed quarter-rate
# figure was 9.83e12; the measured one is the denominator used here).
PEAK_MAD32_PER_S = 31.2e12
MAD32_PER_POINT_MSM = 11833.0       # SURVEY.md 8(d), C5 model (2^20 points, c = 16)
SURVEY_MAD32_PER_FIELD_OP = 64      # a field multiplication or squaring = 64 MAD32 (8x8 schoolbook)


class Evidence:
    """The executed-work counts and PMC traffic figures of the roofline block, READ from the committed evidence files of the
    newest round that has them (profiles/rNN_opcount.json by scripts/opcount.py with the instrumented build; profiles/rNN_pmc_*.csv,
    rNN_msm_pmc_*.csv by scripts/profiles.sh + profiles/summarize.py) - never retyped.  Two work units per figure:
      executed      v_mad_u64_u32 lane-results: counted fp_mul x 102 + fp_sqr x 65 (the MADs of their ISA incl. the reduction)
      survey unit   counted field operations x 64 MAD32 (SURVEY 8(d)'s definition of a field multiplication)"""

    def __init__(self):
        import csv
        import glob
        import re
        self.files = {}
        prof = os.path.join(ROOT, "profiles")

        def newest(pattern, exclude=None):
            c = [f for f in glob.glob(os.path.join(prof, pattern)) if not (exclude and exclude in os.path.basename(f))]
            c.sort(key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
            return c[-1] if c else None

        def rows(f):
            out = {}
            if f:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        out[(r["kernel"], r["counter"])] = float(r["avg_per_launch"])
            return out

        f = newest("r*_opcount.json")
        self.files["opcount"] = os.path.basename(f) if f else None
        self.oc = json.load(open(f)) if f else {}
        for key, pat, excl in (("pmc_read", "r*_pmc_read.csv", "_msm_"), ("pmc_write", "r*_pmc_write.csv", "_msm_"),
                               ("msm_pmc_read", "r*_msm_pmc_read.csv", None), ("msm_pmc_write", "r*_msm_pmc_write.csv", None)):
            f = newest(pat, excl)
            self.files[key] = os.path.basename(f) if f else None
            setattr(self, key, rows(f))

    def _verify_entry(self):
        e = self.oc.get("k_bpv_msm_tableauto_lanes2")
        return e if e and "executed_mad32_per_proof" in e else None

    def verify_msm(self):
        """(executed MAD32, survey-unit MAD32, counted at how many proofs) per n=64 m=1 proof of the launch pair k_bpv_fix + k_bpv_msm"""
        e = self._verify_entry()
        if not e:
            return None, None, None
        return (e["executed_mad32_per_proof"], (e["msm_only_fp_mul_per_proof"] + e["msm_only_fp_sqr_per_proof"]) * SURVEY_MAD32_PER_FIELD_OP,
                e["proofs"])

    def decompress(self):
        e = self.oc.get("k_bpv_decompress")
        if not e:
            return None, None
        return ((e["fp_mul"] * self.oc["mad_per_fp_mul"] + e["fp_sqr"] * self.oc["mad_per_fp_sqr"]) / e["proofs"],
                (e["fp_mul"] + e["fp_sqr"]) * SURVEY_MAD32_PER_FIELD_OP / e["proofs"])

    def msm_point(self, log2n=20):
        e = self.oc.get("msm_2p%d" % log2n)
        if not e:
            return None, None
        return e["executed_mad32_per_point"], (e["fp_mul"] + e["fp_sqr"]) * SURVEY_MAD32_PER_FIELD_OP / e["points"]

    @staticmethod
    def _bytes(rd, wr, kernel):
        """HBM-side bytes per launch of the kernels whose name contains `kernel`: read requests x 64 B (32-byte ones x 32) +
        write requests (64-byte ones x 64, the others x 32).  64-byte gathers, not the 16-B/lane streaming pattern the guide's
        x2 correction was calibrated on: RDREQ x 64 B already matches the algorithmic gather bytes, so no doubling."""
        tot, found = 0.0, False
        for (k, c), v in rd.items():
            if kernel in k and c == "TCC_EA0_RDREQ_sum":
                tot += (v - rd.get((k, "TCC_EA0_RDREQ_32B_sum"), 0.0)) * 64 + rd.get((k, "TCC_EA0_RDREQ_32B_sum"), 0.0) * 32
                found = True
        for (k, c), v in wr.items():
            if kernel in k and c == "TCC_EA0_WRREQ_sum":
                w64 = wr.get((k, "TCC_EA0_WRREQ_64B_sum"), 0.0)
                tot += w64 * 64 + (v - w64) * 32
                found = True
        return tot if found else None

    def verify_traffic(self):
        # the instantiations of the default configuration (65536 proofs per launch: 4 lanes per proof in the table kernel, 2 x 8 points in
        # the ladder kernel); the batch sweep's smaller launches use others
        a, b = self._bytes(self.pmc_read, self.pmc_write, "k_bpv_fix<4>"), self._bytes(self.pmc_read, self.pmc_write, "k_bpv_msm<2, 8>")
        return (a + b) if (a is not None and b is not None) else None

    def msm_acc_traffic(self):
        return self._bytes(self.msm_pmc_read, self.msm_pmc_write, "k_srt_acc")


# Rehearsal of the multi-rank path on a ONE-GPU box (BENCH_REHEARSAL=1 under torch.distributed.run): every rank uses
# GPU 0 and the collectives go over gloo with host tensors.  Never used by the driver's runs (one rank per GPU, RCCL).
REHEARSAL = os.environ.get("BENCH_REHEARSAL") == "1"
REHEARSAL_DRY = os.environ.get("BENCH_REHEARSAL") == "dry"


def dist_setup(world, local_rank):
    import torch
    import torch.distributed as dist
    if REHEARSAL:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    return local_rank, torch.device("cuda", local_rank), (torch.device("cpu") if REHEARSAL else torch.device("cuda", local_rank))



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--proofs-per-gpu", type=int, default=65536)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per proof in k_bpv_msm (16/32/64)")
    ap.add_argument("--window-bits", type=int, default=0, help="fixed-base table window (0 = auto)")
    ap.add_argument("--workload", default="verify", choices=["verify", "msm", "prove"],
                    help="verify = headline metric (default); msm = raw 2^20-point Pippenger MSM, points sharded over the ranks; "
                         "prove = single-proof prove / verify latencies of BASELINE.json configs[0], [1], [3] (one GPU)")
    ap.add_argument("--msm-log2", type=int, default=20)
    ap.add_argument("--msm-per-gpu", action="store_true", help="--workload msm: weak scaling, 2^msm-log2 points on EVERY rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rlc-single", action="store_true", help="RLC side measurement with one context instead of two")
    ap.add_argument("--no-rlc", action="store_true", help="skip the random-linear-combination side measurement")
    ap.add_argument("--no-bpplus", action="store_true", help="skip the Bulletproofs+ batch side measurement")
    ap.add_argument("--no-c2", action="store_true", help="skip the n=64 m=16 batch side measurement")
    ap.add_argument("--no-c4", action="store_true", help="skip the Bulletproofs+ n=64 m=64 batch side measurement (configs[3])")
    ap.add_argument("--no-prove", action="store_true", help="skip the prover side measurements (configs[0], [1], [3])")
    ap.add_argument("--no-msm", action="store_true", help="skip the raw 2^k-point MSM side measurement")
    ap.add_argument("--no-overlap", action="store_true", help="bp_verify_set_overlap(ctx, 0): successive device calls one after the other (A/B)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep")
    ap.add_argument("--no-tiled", action="store_true", help="skip the tiled-batch (round 1 data shape) side measurement")
    ap.add_argument("--prove-threads", type=int, default=8, help="--workload prove: contexts in flight for the throughput figure")
    ap.add_argument("--skip-check", action="store_true", help="timing experiments only: do not compare verdicts")
    ap.add_argument("--cpu-sample", type=int, default=1200, help="proofs verified by the CPU baseline")
    ap.add_argument("--cpu-worker", type=float, default=0, help=argparse.SUPPRESS)  # one process of cpu_baseline's all-cores leg
    return ap.parse_args()


def load_batch():
    with open(os.path.join(ROOT, "tests", "golden", "bp_c3_batch_n64_m1.json")) as f:
        b = json.load(f)
    return b


def _cpu_verify_loop(batch, sample, seconds):
    """(proofs verified, seconds) of the oracle's aggregated_verify over the golden C3 batch on this core."""
    import subprocess
    from oracle import ec, encoding
    if not ec.have_c():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if not ec.have_c():
        return None
    g = bytes.fromhex(batch["g_vec"])
    h = bytes.fromhex(batch["h_vec"])
    g_vec = [ec.dec64(g[i:i + 64]) for i in range(0, len(g), 64)]
    h_vec = [ec.dec64(h[i:i + 64]) for i in range(0, len(h), 64)]
    G, H = ec.dec64(bytes.fromhex(batch["G"])), ec.dec64(bytes.fromhex(batch["H"]))
    proofs = bytes.fromhex(batch["proofs"])
    coms = bytes.fromhex(batch["commitments"])
    plen = 292 + 66 * 6
    n_ok = 0
    t0 = time.perf_counter()
    done = 0
    for j in range(sample):
        i = j % batch["count"]
        p = encoding.decode_bp(proofs[i * plen:(i + 1) * plen])
        com = [ec.dec64(coms[i * 64:(i + 1) * 64])]
        n_ok += p.aggregated_verify(g_vec, h_vec, G, H, com, 64) == "Ok"
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    assert n_ok == done
    return done, dt


def _openssl_mul_rate(seconds=2.0):
    """EC_POINT_mul(secp256k1, variable point) per second on one core through libcrypto, or None when it is absent."""
    import ctypes
    import ctypes.util
    name = ctypes.util.find_library("crypto")
    if not name:
        return None
    try:
        L = ctypes.CDLL(name)
        vp = ctypes.c_void_p
        for f, a in (("EC_GROUP_new_by_curve_name", [ctypes.c_int]), ("EC_POINT_new", [vp]), ("BN_new", []), ("BN_CTX_new", []),
                     ("EC_GROUP_get0_generator", [vp])):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = a
        L.EC_POINT_mul.argtypes = [vp, vp, vp, vp, vp, vp]
        L.BN_bin2bn.argtypes = [ctypes.c_char_p, ctypes.c_int, vp]
        L.BN_bin2bn.restype = vp
        grp = L.EC_GROUP_new_by_curve_name(714)  # NID_secp256k1
        bctx = L.BN_CTX_new()
        P, R = L.EC_POINT_new(grp), L.EC_POINT_new(grp)
        k = L.BN_bin2bn(shake(b"bpmsm-openssl-k", 32), 32, None)
        L.EC_POINT_mul(grp, P, k, None, None, bctx)  # P = k G: a variable point
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(200):
                if L.EC_POINT_mul(grp, R, None, P, k, bctx) != 1:
                    return None
            n += 200
        return n / (time.perf_counter() - t0)
    except Exception:
        return None


def cpu_baseline(batch, sample):
    """Oracle aggregated_verify (Python scalar derivation + C term-by-term MSM): one core (`value`, the reference is
    single-threaded), the same loop on every core of this box's share as separate processes (`all_cores`), and
    libcrypto's EC_POINT_mul rate turned into proofs/s by the reference's 147 multiplications per verification."""
    import subprocess
    r = _cpu_verify_loop(batch, sample, 20)
    if r is None:
        return None
    done, dt = r
    out = {"value": done / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
           "sample": "%d n=64 m=1 proofs, oracle aggregated_verify (term-by-term MSM in C, as range_proof.rs:772-774)" % done}
    ncores = min(16, os.cpu_count() or 1)  # the GPU box's share is 16 cores whatever os.cpu_count() says
    procs = []
    try:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "8"], stdout=subprocess.PIPE, text=True)
                 for _ in range(ncores)]
        tot, tmax = 0, 0.0
        for p in procs:
            o, _ = p.communicate(timeout=120)
            d, t = o.strip().split()[-2:]
            tot += int(d)
            tmax = max(tmax, float(t))
        out["all_cores"] = {"value": tot / tmax, "unit": "proofs/s", "cores": ncores,
                            "sample": "%d proofs over %d processes of the same loop, %.1f s" % (tot, ncores, tmax)}
    except Exception as e:  # a baseline, never a reason to lose the bench line
        out["all_cores"] = {"error": repr(e)[:200]}
    finally:  # no worker outlives this function (a timed-out communicate() leaves the others running)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
    rate = _openssl_mul_rate()
    if rate:
        out["openssl_ec_point_mul"] = {"mul_per_s": rate, "proofs_per_s_equiv": rate / 147.0, "cores": 1,
                                       "what": "libcrypto EC_POINT_mul on a variable secp256k1 point, one core; a verification is 147 "
                                               "such multiplications + 146 additions in the reference (range_proof.rs:772-774)"}
    return out


def msm_roofline(ev, mpts_per_gpu, log2n):
    """roofline object of the raw MSM: whole bp_msm_device pipeline against the measured MAD peak, in three work units."""
    ex, su = ev.msm_point(log2n)
    rate = mpts_per_gpu * 1e6
    return {"bound": "valu_int32_mad", "kernel": "whole bp_msm_device pipeline (dominant kernel k_srt_acc)",
            "achieved": rate * (ex or MAD32_PER_POINT_MSM) / 1e12, "peak": PEAK_MAD32_PER_S / 1e12, "unit": "TMAD32/s",
            "frac": rate * (ex or MAD32_PER_POINT_MSM) / PEAK_MAD32_PER_S,
            "frac_executed": rate * ex / PEAK_MAD32_PER_S if ex else None,
            "frac_survey_unit": rate * su / PEAK_MAD32_PER_S if su else None,
            "frac_model": rate * MAD32_PER_POINT_MSM / PEAK_MAD32_PER_S,
            "executed_mad32_per_point": ex, "survey_unit_mad32_per_point": su, "model_mad32_per_point": MAD32_PER_POINT_MSM,
            "what": "frac = frac_executed (counted fp_mul x 102 + fp_sqr x 65 per point, %s) when a count exists for this size, else "
                    "frac_model; frac_survey_unit = counted field operations x 64 MAD32 (SURVEY 8(d)'s unit); frac_model = SURVEY 8(d)'s "
                    "Pippenger model (c = 16)" % ev.files.get("opcount"),
            "traffic": ev.msm_acc_traffic() if log2n == 20 else None,
            "traffic_unit": "bytes per k_srt_acc launch (PMC, %s / %s); algorithmic per MSM: 18.9 M entries x (8 B entry + 64 B point) "
                            "= 1.36 GB over the launches of one MSM" % (ev.files.get("msm_pmc_read"), ev.files.get("msm_pmc_write")),
            "evidence_files": ev.files}


def bench_msm(args):
    """BASELINE.json configs[4]: raw Pippenger MSM of 2^k secp256k1 points; points sharded N/G per rank, per-rank per-window
    partial sums all-gathered over RCCL in DEVICE memory right behind the bucket-reduction kernels (no host hop:
    bp_msm_partial_device_out -> bp_ctx_record_event -> all_gather_into_tensor -> bp_ctx_wait_event -> bp_msm_combine_device;
    bulletproofs_amd/batch.py: msm_sharded_tensors), added, one Horner combine.  Inputs resident in HBM; the 64-byte result
    returns to the host.  Default: STRONG scaling (2^--msm-log2 points in total).  --msm-per-gpu: WEAK scaling (2^--msm-log2
    points on every rank, N * 2^k in total, the same exchange)."""
    import hashlib
    import torch
    import torch.distributed as dist
    import bulletproofs_amd as bpa
    from bulletproofs_amd import batch as bt
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_rank, dev, cdev = dist_setup(world, local_rank)
    N = (1 << args.msm_log2) * (world if args.msm_per_gpu else 1)
    lo, hi = bt.shard_range(N, rank, world)
    n = hi - lo
    ctx = bpa.Context(local_rank)
    # synthetic points P_i = r_i * G (tape scalars) made on the GPU; scalars from a SHA-256 counter stream
    def stream(tag, count, start):
        out = bytearray()
        for i in range(start, start + count):
            out += hashlib.sha256(b"bpmsm-c5-%s-%d" % (tag, i)).digest()
        return bytes(out)
    pts = bytearray()
    CH = 1 << 16
    for s0 in range(lo, hi, CH):
        c = min(CH, hi - s0)
        pts += ctx.fixed_base_mul_batch(stream(b"r", c, s0))
    scal = stream(b"k", n, lo)
    d_pts = torch.frombuffer(pts, dtype=torch.uint8).to(dev)
    d_sc = torch.frombuffer(bytearray(scal), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    cwin = bpa.msm_auto_window(bt.largest_shard(N, world))
    psize = bpa.lib().bp_msm_partial_bytes(cwin)
    ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
    ev_a.record()
    ev_b.record()  # (creates the handles)
    torch.cuda.synchronize()
    on_gpu = cdev.type == "cuda"  # the one-GPU rehearsal moves host tensors over gloo

    def partial_into(lo_, hi_, t):
        if on_gpu:
            ctx.msm_partial_device_out(d_pts.data_ptr(), d_sc.data_ptr(), hi_ - lo_, cwin, t.data_ptr())
        else:
            t.copy_(torch.frombuffer(bytearray(ctx.msm_partial_device(d_pts.data_ptr(), d_sc.data_ptr(), hi_ - lo_, cwin)), dtype=torch.uint8))

    def after_partial():  # the collective's stream waits for the library's kernels, the host does not
        if on_gpu:
            ctx.record_event(ev_a.cuda_event)
            torch.cuda.current_stream().wait_event(ev_a)

    def before_combine():
        if on_gpu:
            ev_b.record()
            ctx.wait_event(ev_b.cuda_event)

    def combine_from(t, g):
        if on_gpu:
            return ctx.msm_combine_device(t.data_ptr(), g, cwin)
        return ctx.msm_combine(bytes(t.numpy().tobytes()), g, cwin)

    def step():
        if world == 1:
            return ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
        return bt.msm_sharded_tensors(N, psize, partial_into, combine_from, cdev, after_partial, before_combine)

    for _ in range(args.warmup):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # Throughput with two contexts in flight (one GPU only): two host threads, one bp_ctx each, every call still one
    # complete 2^k MSM; the latency-bound tail of one (bucket reduction, Horner chain, inversion) overlaps the
    # other's accumulation.  Reported beside `value`, which stays the one-call-at-a-time rate.
    two = None
    if world == 1:
        import threading
        ctx_b = bpa.Context(local_rank)
        ctx.set_msm_pipeline(1)  # several contexts in flight: one window group each (bp_msm_set_pipeline)
        ctx_b.set_msm_pipeline(1)
        ctx_b.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
        outs = [None, None]

        def worker(i, c, cnt):
            for _ in range(cnt):
                outs[i] = c.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), n)

        thr = [threading.Thread(target=worker, args=(0, ctx, args.steps)), threading.Thread(target=worker, args=(1, ctx_b, args.steps))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        assert outs[0] == res and outs[1] == res
        two = {"value": 2 * N * args.steps / dt2 / 1e6, "unit": "Mpts/s", "contexts_in_flight": 2, "ms_per_msm": dt2 / (2 * args.steps) * 1e3}
        ctx_b.close()
    if rank == 0:
        mpts = N * args.steps / dt / 1e6
        ev = Evidence()
        shard_log2 = (bt.largest_shard(N, world) - 1).bit_length()
        print(json.dumps({
            "metric": "MSM Mpts/s (raw Pippenger MSM, secp256k1)", "value": mpts, "unit": "Mpts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.msm_per_gpu else "strong", "vs_baseline": None, "dtype": "u32 (256-bit modular integer arithmetic)",
            "data": "synthetic (P_i = r_i*G, SHA-256 counter-stream scalars)",
            "config": {"workload": "raw MSM of %d points (%s), points sharded over %d GPU(s), per-window partial sums all-gathered in device "
                                   "memory" % (N, "2^%d per GPU" % args.msm_log2 if args.msm_per_gpu else "2^%d in total" % args.msm_log2, world),
                       "points_per_gpu": n, "window_bits": cwin, "exchange_bytes_per_rank": psize, "result": res.hex()},
            "two_in_flight": two,
            "roofline": msm_roofline(ev, mpts / world, shard_log2)}), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def bench_prove(args):
    """Single-proof latencies through the C ABI (protocol layer in C++ over the GPU primitives):
    BASELINE.json configs[0] (BP n=64 m=1), configs[1] (BP n=64 m=16), configs[3] (BP+ n=64 m=64).
    Inputs are the golden fixtures (tests/golden); every produced proof is compared with the fixture bytes."""
    import bulletproofs_amd as bpa
    ctx = bpa.Context(int(os.environ.get("LOCAL_RANK", "0")))

    def be(xs):
        return b"".join(int(x, 16).to_bytes(32, "big") for x in xs)

    def load(name, idx=None):
        with open(os.path.join(ROOT, "tests", "golden", name)) as f:
            d = json.load(f)
        return d if idx is None else d[idx]

    cases = [("C1 BP n=64 m=1", 0, load("bp_cases.json", 3)), ("C2 BP n=64 m=16", 0, load("bp_c2_n64_m16.json")),
             ("C4 BP+ n=64 m=64", 1, load("bpplus_c4_n64_m64.json"))]
    rows = []
    for name, kind, case in cases:
        n, m = case["n"], case["m"]
        g, h, G, H = bpa.generate_bases(kind, bpa.KZEN_SEED, n, m)
        gens = bpa.Gens(ctx, g, h, G, H, 0)
        sec, bl, tape = be(case["secrets"]), be(case["blindings"]), bytes.fromhex(case["tape"])
        coms = b"".join(bytes.fromhex(c) for c in case["commitments"])
        prove = ctx.wip_rangeproof_prove if kind else ctx.rangeproof_prove
        verify = ctx.wip_rangeproof_verify if kind else ctx.rangeproof_verify

        def timeit(fn, reps):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            return (time.perf_counter() - t0) / reps * 1e3, r

        reps = max(1, args.steps // 4)
        t_prove, proof = timeit(lambda: prove(gens, n, sec, bl, tape), reps)
        assert proof.hex() == case["proof"], "proof bytes differ from the golden fixture"
        row = {"config": name, "prove_ms": t_prove, "proof_bytes": len(proof)}
        for mode, label in ((2, "aggregated_verify_ms"), (1, "fast_verify_ms"), (0, "verify_ms")):
            if kind and mode == 1:
                continue
            t, v = timeit(lambda: verify(gens, n, m, proof, coms, mode), reps)
            assert v == 0
            row[label] = t
        # throughput with several contexts in flight (the library's threading model: one bp_ctx per host thread, bp_gens
        # shared): the per-round host work and round trips of one proof overlap the GPU work of the others
        import threading
        nthr = args.prove_threads
        if nthr > 1:
            ctxs = [bpa.Context(int(os.environ.get("LOCAL_RANK", "0"))) for _ in range(nthr)]
            per = max(2, reps)
            bad = []

            def worker(c):
                for _ in range(per):
                    if prove(gens, n, sec, bl, tape) != proof if c is ctx else (c.wip_rangeproof_prove if kind else c.rangeproof_prove)(
                            gens, n, sec, bl, tape) != proof:
                        bad.append(1)

            for c in ctxs:  # warm-up: workspace allocation
                (c.wip_rangeproof_prove if kind else c.rangeproof_prove)(gens, n, sec, bl, tape)
            thr = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
            t0 = time.perf_counter()
            for t in thr:
                t.start()
            for t in thr:
                t.join()
            dtp = time.perf_counter() - t0
            assert not bad, "a concurrent prove produced different bytes"
            row["prove_per_s_%d_contexts" % nthr] = nthr * per / dtp
            for c in ctxs:
                c.close()
        # batch prover: N proofs per call, one GPU launch per protocol step for all of them, scalar work on host threads
        nb = {1: 512, 16: 128, 64: 32}.get(m, 64)
        pb = ctx.wip_rangeproof_prove_batch if kind else ctx.rangeproof_prove_batch
        pb(gens, n, m, sec * 8, bl * 8, tape * 8, len(tape))
        t0 = time.perf_counter()
        got = pb(gens, n, m, sec * nb, bl * nb, tape * nb, len(tape))
        dtb = time.perf_counter() - t0
        assert got[0] == proof and got[-1] == proof and len(got) == nb
        row["prove_batch_per_s"] = nb / dtb
        row["prove_batch_size"] = nb
        rows.append(row)
        gens.close()
    out = {"metric": "single-proof latency (ms), secp256k1 range proofs through the C ABI", "unit": "ms", "n_gpus": 1,
           "higher_is_better": False, "data": "golden fixtures (tests/golden), proof bytes checked", "rows": rows}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_prove_baseline(cases[0][2])
    print(json.dumps(out), flush=True)
    ctx.close()


def cpu_prove_baseline(case):
    """Oracle RangeProof.prove / aggregated_verify of the C1 fixture on one host core ("port": Python scalars +
    the C engine's term-by-term MSMs, the reference's cost structure; not curv/libsecp256k1)."""
    import subprocess
    from oracle import curv, ec, encoding, tape
    from oracle.range_proof import RangeProof
    if not ec.have_c():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    n, m = case["n"], case["m"]
    g_vec, h_vec, G, H = curv.bp_generators(n, m)
    v = [int(x, 16) for x in case["secrets"]]
    r = [int(x, 16) for x in case["blindings"]]
    com = [ec.dec64(bytes.fromhex(c)) for c in case["commitments"]]
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = RangeProof.prove(g_vec, h_vec, G, H, v, r, n, tape.Tape(bytes.fromhex(case["tape"])))
    t_prove = (time.perf_counter() - t0) / reps * 1e3
    assert encoding.encode_bp(proof).hex() == case["proof"]
    t0 = time.perf_counter()
    for _ in range(reps):
        ok = proof.aggregated_verify(g_vec, h_vec, G, H, com, n)
    t_ver = (time.perf_counter() - t0) / reps * 1e3
    assert ok == "Ok"
    return {"config": "C1 BP n=64 m=1", "prove_ms": t_prove, "aggregated_verify_ms": t_ver, "unit": "ms", "cores": 1, "kind": "port",
            "sample": "%d proves + %d verifies of the C1 fixture by the oracle (C engine: %s)" % (reps, reps, ec.have_c())}


def shake(tag, nbytes):
    """Deterministic synthetic bytes (SHAKE-256 of a tag): one call per array instead of one SHA-256 per scalar."""
    import hashlib
    return hashlib.shake_256(tag).digest(nbytes)


def scalars_in_range(tag, count):
    """count 32-byte big-endian scalars in [1, 2^255) (all < q): top bit cleared, low bit set."""
    import numpy as np
    a = np.frombuffer(shake(tag, 32 * count), dtype=np.uint8).reshape(count, 32).copy()
    a[:, 0] &= 0x7F
    a[:, 31] |= 1
    return a.tobytes()


def make_distinct_batch(ctx, gens, G, H, kind, m, count, tag):
    """count distinct valid n=64 proofs (+ commitments V_j = v_j G + r_j H) made on the GPU by the batch provers.
    kind 0 = Bulletproofs, 1 = Bulletproofs+.  Returns (proofs bytes, commitments bytes, proof_len)."""
    import numpy as np
    nm = 64 * m
    k = nm.bit_length() - 1
    ntape = (1 + 2 * k + 4) if kind else (2 + 2 * nm + 2)
    v8 = np.frombuffer(shake(b"v-" + tag, 8 * count * m), dtype=np.uint8).reshape(count * m, 8)
    secs = np.zeros((count * m, 32), dtype=np.uint8)
    secs[:, 24:] = v8
    secs = secs.tobytes()
    blinds = scalars_in_range(b"r-" + tag, count * m)
    prove = ctx.wip_rangeproof_prove_batch if kind else ctx.rangeproof_prove_batch
    plen = (195 if kind else 292) + 66 * k
    out = []
    CH = 8192 if m == 1 else 512
    for lo in range(0, count, CH):
        c = min(CH, count - lo)
        tapes = scalars_in_range(b"t-%d-" % lo + tag, c * ntape)
        out += prove(gens, 64, m, secs[32 * m * lo:32 * m * (lo + c)], blinds[32 * m * lo:32 * m * (lo + c)], tapes, 32 * ntape)
    # commitments: v*G (fixed base), r*H, added by a fold with unit scalars
    one = (1).to_bytes(32, "big")
    coms = bytearray()
    CC = 1 << 15
    for lo in range(0, count * m, CC):
        c = min(CC, count * m - lo)
        vG = ctx.scalar_mul_batch(G * c, secs[32 * lo:32 * (lo + c)])
        rH = ctx.scalar_mul_batch(H * c, blinds[32 * lo:32 * (lo + c)])
        coms += ctx.fold_bases(vG, rH, one, one)
    return b"".join(out), bytes(coms), plen


def tamper_batch(pb, plen, count, at):
    """flip one bit of every 16th proof (byte `at`: tx / delta'); returns the expected verdict bytes"""
    want = bytearray(count)
    for j in range(5, count, 16):
        pb[j * plen + at] ^= 1
        want[j] = 3
    return want


def spawn_ranks(args):
    """`bench.py --gpus N` started by hand (no WORLD_SIZE): start N ranks as a child torch.distributed.run and pass its
    exit code on.  This parent has not touched a GPU and never execs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def dry_run(args):
    """BENCH_REHEARSAL=dry: the launcher / rendezvous / barrier / max-over-ranks / JSON plumbing with NO library call
    (a box without a GPU, tests/test_bench_launcher.py); prints "dry_run": true and value 0 - never a measurement."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "verified range proofs/sec (secp256k1, n=64) at 1/2/4/8 GPUs; MSM Mpts/s", "value": 0.0, "unit": "proofs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
                          "dry_run": True, "config": {"workload": args.workload}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def msm_side(ctx, dev, log2n, steps):
    """BASELINE.json configs[4] on this one GPU inside the default run: raw 2^k-point MSM, inputs resident in HBM,
    one call at a time; result cross-checked between two calls.  (--workload msm is the multi-GPU form.)"""
    import torch
    N = 1 << log2n
    pts = bytearray()
    CH = 1 << 16
    for s0 in range(0, N, CH):
        pts += ctx.fixed_base_mul_batch(scalars_in_range(b"bpmsm-c5-r-%d" % s0, CH))
    scal = shake(b"bpmsm-c5-k", 32 * N)
    d_pts = torch.frombuffer(pts, dtype=torch.uint8).to(dev)
    d_sc = torch.frombuffer(bytearray(scal), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    res = ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)
    ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r2 = ctx.msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert r2 == res
    mpts = N / dt / 1e6
    # The same MSM from two host threads with a context each (the library's threading model; bp_msm_device returns the result,
    # so ONE context cannot have two MSMs in flight): one's sort / reduction / Horner phases run beside the other's
    # accumulation.  Reported beside `value`, never as it.
    import threading
    import bulletproofs_amd as bpa
    two = []
    extra = []
    try:
        ctx.set_msm_pipeline(1)  # one window group, no side streams: less total GPU work with several contexts in flight
        for nctx in (2, 3):
            while len(extra) < nctx - 1:
                extra.append(bpa.Context(dev.index or 0))
                extra[-1].set_msm_pipeline(1)
            ctxs = [ctx] + extra
            outs = [None] * nctx

            def worker(i, reps):
                for _ in range(reps):
                    outs[i] = ctxs[i].msm_device(d_pts.data_ptr(), d_sc.data_ptr(), N)

            for i in range(nctx):
                worker(i, 2)
            torch.cuda.synchronize()
            reps = max(steps // 2, 4)
            thr = [threading.Thread(target=worker, args=(i, reps)) for i in range(nctx)]
            t0 = time.perf_counter()
            for t in thr:
                t.start()
            for t in thr:
                t.join()
            dt2 = (time.perf_counter() - t0) / (nctx * reps)
            assert all(o == res for o in outs)
            two.append({"value": N / dt2 / 1e6, "unit": "Mpts/s", "ms_per_msm": dt2 * 1e3, "contexts_in_flight": nctx,
                        "what": "%d host threads, one context each, bp_msm_set_pipeline(ctx, 1); same points and result" % nctx})
    finally:
        ctx.set_msm_pipeline(0)
        for c in extra:
            c.close()
    return {"value": mpts, "unit": "Mpts/s", "log2_points": log2n, "ms_per_msm": dt * 1e3, "result": res.hex(), "contexts_in_flight": two,
            "what": "bp_msm_device, one call at a time, inputs resident in HBM, 64-byte result returned to the host",
            "roofline": msm_roofline(Evidence(), mpts, log2n)}


def main():
    args = parse()
    if args.cpu_worker > 0:  # CPU only: one process of cpu_baseline's all-cores leg
        done, dt = _cpu_verify_loop(load_batch(), 1 << 30, args.cpu_worker)
        print(done, "%.3f" % dt)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if REHEARSAL_DRY:
        return dry_run(args)
    if args.workload == "msm":
        return bench_msm(args)
    if args.workload == "prove":
        return bench_prove(args)
    import torch
    import torch.distributed as dist
    import bulletproofs_amd as bpa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libbpmsm has no CPU path)")
    local_rank, dev, cdev = dist_setup(world, local_rank)

    golden = load_batch()
    P = args.proofs_per_gpu
    ctx = bpa.Context(local_rank)
    g_xy, h_xy, G_xy, H_xy = (bytes.fromhex(golden[x]) for x in ("g_vec", "h_vec", "G", "H"))
    gens = bpa.Gens(ctx, g_xy, h_xy, G_xy, H_xy, args.window_bits)
    if args.lanes:
        ctx.set_verify_lanes(args.lanes)
    TX_AT = 4 * 33 + 64 + 31  # last byte of tx

    # The proof batch is made ONCE per node - rank 0's GPU batch prover - and broadcast: the batch prover's per-proof scalar
    # work runs on host threads, and eight ranks generating at once fight for the same cores before the first timed step
    # (10.8 s alone, 12.5 s with two ranks in round 2's rehearsal).  Every rank then verifies the same 65536 distinct proofs.
    t_gen = time.perf_counter()
    plen = 292 + 66 * 6
    if rank == 0:
        proofs, coms, plen = make_distinct_batch(ctx, gens, G_xy, H_xy, 0, 1, P, b"bpmsm-bench-c3")
        pb = bytearray(proofs)
        want = tamper_batch(pb, plen, P, TX_AT)
        d_proofs = torch.frombuffer(pb, dtype=torch.uint8).to(dev)
        d_coms = torch.frombuffer(bytearray(coms), dtype=torch.uint8).to(dev)
    else:
        proofs = None
        want = tamper_batch(bytearray(P * plen), plen, P, TX_AT)
        d_proofs = torch.empty(P * plen, dtype=torch.uint8, device=dev)
        d_coms = torch.empty(P * 64, dtype=torch.uint8, device=dev)
    if world > 1:
        if REHEARSAL:  # gloo: host tensors
            hp, hc = d_proofs.cpu(), d_coms.cpu()
            dist.broadcast(hp, 0)
            dist.broadcast(hc, 0)
            d_proofs, d_coms = hp.to(dev), hc.to(dev)
        else:
            dist.broadcast(d_proofs, 0)
            dist.broadcast(d_coms, 0)
    t_gen = time.perf_counter() - t_gen
    # Successive device calls of one context overlap (bp_verify_set_overlap: two batches in flight), so every call in
    # flight gets a verdict buffer of its own.
    d_verd = [torch.full((P,), 77, dtype=torch.uint8, device=dev) for _ in range(2)]
    calls = [0]
    if args.no_overlap:
        ctx.set_verify_overlap(0)
    torch.cuda.synchronize()

    def run(count, dp=d_proofs, dc=d_coms):
        dv = d_verd[calls[0] & 1]
        calls[0] += 1
        ctx.rangeproof_verify_batch_device(gens, 64, 1, dp.data_ptr(), plen, dc.data_ptr(), count, dv.data_ptr())

    def check(count, what, dp=d_proofs, dc=d_coms):
        for dv in d_verd:
            dv.fill_(77)
        torch.cuda.synchronize()
        run(count, dp, dc)
        run(count, dp, dc)
        ctx.sync()
        for dv in d_verd:
            if bytes(dv[:count].cpu().numpy().tobytes()) != bytes(want[:count]) and not args.skip_check:
                raise SystemExit("%s: verdict mismatch on rank %d" % (what, rank))

    def step():
        run(P)

    for _ in range(args.warmup):
        step()
    check(P, "headline batch")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # timed region: exactly K steps, barrier + synchronize on both sides
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)

    # per-kernel device time of the dominant kernel, HIP events on the library's own stream
    # (separate pass so event recording does not perturb the timed region)
    def kernel_ms(fn, reps):
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(reps):
            fn()
            ctx.sync()
        out = {}
        for name, kid in (("bpv_decompress", 0), ("bpv_scalars", 1), ("bpv_msm", 2)):
            ms, n = ctx.profile_get(kid)
            out[name] = ms / max(n, 1)
        ctx.profile_enable(False)
        return out

    kms = kernel_ms(step, max(3, min(args.steps, 10)))

    # batch-size sweep on prefixes of the same distinct batch (BASELINE.json configs[2] names 4096 proofs over 8 GPUs =
    # 512 per GPU: the latency-bound regime), verdicts of every prefix checked
    sweep = []
    if not args.no_sweep:
        for count in (512, 4096, 16384, 65536):
            if count > P:
                continue
            check(count, "sweep at %d proofs" % count)
            reps = max(args.steps, 20 if count <= 4096 else args.steps)
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                run(count)
            ctx.sync()
            barrier()
            dts = max_over_ranks(time.perf_counter() - t0)
            # the same launches one at a time (a sync after each): the latency of one batch, no overlap between calls
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                run(count)
                ctx.sync()
            barrier()
            dtl = max_over_ranks(time.perf_counter() - t0)
            sweep.append({"proofs_per_launch_per_gpu": count, "ms_per_launch": dts / reps * 1e3, "proofs_per_s": world * count * reps / dts,
                          "ms_one_call_at_a_time": dtl / reps * 1e3})

    # the round-1 data shape, once, for the record: 256 distinct golden proofs tiled to the batch
    tiled = None
    if world == 1 and not args.no_tiled:
        cnt = golden["count"]
        gp, gc = bytes.fromhex(golden["proofs"]), bytes.fromhex(golden["commitments"])
        reps_t = (P + cnt - 1) // cnt
        tb = bytearray((gp * reps_t)[:P * plen])
        tamper_batch(tb, plen, P, TX_AT)
        d_tp = torch.frombuffer(tb, dtype=torch.uint8).to(dev)
        d_tc = torch.frombuffer(bytearray((gc * reps_t)[:P * 64]), dtype=torch.uint8).to(dev)
        check(P, "tiled batch", d_tp, d_tc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run(P, d_tp, d_tc)
        ctx.sync()
        torch.cuda.synchronize()
        dtt = time.perf_counter() - t0
        kt = kernel_ms(lambda: run(P, d_tp, d_tc), 3)
        tiled = {"value": P * args.steps / dtt, "unit": "proofs/s", "ms_per_step": dtt / args.steps * 1e3, "bpv_msm_ms": kt["bpv_msm"],
                 "what": "the same launch on 256 distinct proofs tiled to the batch (round 1's data): every gather pattern repeats "
                         "%d times, so the table gathers stay in the caches - an easier memory pattern, not the metric" % (P // cnt)}
        del d_tp, d_tc

    # Random-linear-combination mode on the all-valid version of the same batch (reported beside the headline
    # value, never as it: `value` is N independent aggregated_verify checks, the reference's semantics).
    rlc = None
    if not args.no_rlc:
        d_valid = d_proofs.clone()  # the all-valid version: flip the tampered bits back
        idx = torch.arange(5, P, 16, device=dev, dtype=torch.int64) * plen + TX_AT
        d_valid[idx] ^= 1
        d_v2 = torch.full((P,), 77, dtype=torch.uint8, device=dev)
        seed = bytes((7 * i + rank) & 255 for i in range(32))
        fb = ctx.rangeproof_verify_batch_rlc_device(gens, 64, 1, d_valid.data_ptr(), plen, d_coms.data_ptr(), P, seed, d_v2.data_ptr())
        assert not fb and not bool(d_v2.any().item()), "rlc: all-valid batch must pass the combined check"
        # two contexts on two host threads (the library's threading model: one bp_ctx per thread, bp_gens shared):
        # the latency-bound tail of one batch's MSM overlaps the other batch's decompression / accumulation
        import threading
        ctx_b = bpa.Context(local_rank)
        if not args.rlc_single:  # several contexts in flight: one window group per MSM (bp_msm_set_pipeline)
            ctx.set_msm_pipeline(1)
            ctx_b.set_msm_pipeline(1)
        d_v3 = torch.full((P,), 77, dtype=torch.uint8, device=dev)
        ctx_b.rangeproof_verify_batch_rlc_device(gens, 64, 1, d_valid.data_ptr(), plen, d_coms.data_ptr(), P, seed, d_v3.data_ptr())
        nthr = 1 if args.rlc_single else 2

        def worker(c, dv, cnt):
            for _ in range(cnt):
                c.rangeproof_verify_batch_rlc_device(gens, 64, 1, d_valid.data_ptr(), plen, d_coms.data_ptr(), P, seed, dv.data_ptr())

        share = [args.steps - args.steps // nthr * (nthr - 1)] + [args.steps // nthr] * (nthr - 1)
        thr = [threading.Thread(target=worker, args=(c, dv, cnt)) for c, dv, cnt in zip((ctx, ctx_b), (d_v2, d_v3), share)]
        barrier()
        t0 = time.perf_counter()
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        barrier()
        dt_rlc = max_over_ranks(time.perf_counter() - t0)
        assert not bool(d_v2.any().item()) and not bool(d_v3.any().item())
        ctx_b.close()
        ctx.set_msm_pipeline(0)
        rlc = {"value": world * P * args.steps / dt_rlc, "unit": "proofs/s", "ms_per_step": dt_rlc / args.steps * 1e3, "contexts_in_flight": nthr,
               "what": "bp_rangeproof_verify_batch_rlc_device on the same batch with no tampered proof: one fixed-base MSM over the "
                       "generators + one %d-point Pippenger MSM per step; a failing batch falls back to the per-proof path" % (17 * P)}
        del d_valid

    # Side measurements on one GPU (reported beside `value`, never as it), DISTINCT proofs made here by the GPU batch
    # provers (pinned to the oracle's bytes by tests/test_gpu_proto.py), 1/16 tampered, verdicts checked:
    #   bpplus_batch  RangeProofWIP::aggregated_verify (Bulletproofs+), n=64 m=1, as many proofs per step as the headline
    #   c2_batch      RangeProof::aggregated_verify at BASELINE.json configs[1]'s shape, n=64 m=16 (2089-term MSM per proof)
    def be32(xs):
        return b"".join(int(x, 16).to_bytes(32, "big") for x in xs)

    def prover_numbers(gens_x, kind, fixture, nb):
        """Single-proof prove / aggregated_verify latency and batch-prove rate on a golden fixture (tests/golden), the proof
        bytes compared with the fixture's (RangeProof::prove range_proof.rs:45-263 / RangeProofWIP::prove range_proof_wip.rs:101-274)."""
        with open(os.path.join(ROOT, "tests", "golden", fixture[0])) as f:
            case = json.load(f)
        if fixture[1] is not None:
            case = case[fixture[1]]
        n_, m_ = case["n"], case["m"]
        sec, bl, tape = be32(case["secrets"]), be32(case["blindings"]), bytes.fromhex(case["tape"])
        coms_ = b"".join(bytes.fromhex(c) for c in case["commitments"])
        prove = ctx.wip_rangeproof_prove if kind else ctx.rangeproof_prove
        verify = ctx.wip_rangeproof_verify if kind else ctx.rangeproof_verify
        pbatch = ctx.wip_rangeproof_prove_batch if kind else ctx.rangeproof_prove_batch
        proof = prove(gens_x, n_, sec, bl, tape)
        if proof.hex() != case["proof"]:
            raise SystemExit("prover: proof bytes differ from the golden fixture %s" % fixture[0])
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            prove(gens_x, n_, sec, bl, tape)
        t_prove = (time.perf_counter() - t0) / reps * 1e3
        assert verify(gens_x, n_, m_, proof, coms_, 2) == 0
        t0 = time.perf_counter()
        for _ in range(reps):
            verify(gens_x, n_, m_, proof, coms_, 2)
        t_ver = (time.perf_counter() - t0) / reps * 1e3
        t_fast = None
        if not kind:  # RangeProof::fast_verify (range_proof.rs:400-535): the call most of the reference's criterion benches time
            assert verify(gens_x, n_, m_, proof, coms_, 1) == 0
            t0 = time.perf_counter()
            for _ in range(reps):
                verify(gens_x, n_, m_, proof, coms_, 1)
            t_fast = (time.perf_counter() - t0) / reps * 1e3
        pbatch(gens_x, n_, m_, sec * 8, bl * 8, tape * 8, len(tape))
        t0 = time.perf_counter()
        got = pbatch(gens_x, n_, m_, sec * nb, bl * nb, tape * nb, len(tape))
        dtb = time.perf_counter() - t0
        if not (len(got) == nb and got[0] == proof and got[-1] == proof):
            raise SystemExit("batch prover: proof bytes differ from the golden fixture %s" % fixture[0])
        return {"fixture": fixture[0], "n": n_, "m": m_, "proof_bytes": len(proof), "prove_ms": t_prove, "aggregated_verify_ms": t_ver,
                "fast_verify_ms": t_fast, "prove_batch_per_s": nb / dtb, "prove_batch_size": nb,
                "what": "single-proof latency through the C ABI and batch-prover rate (bp_%srangeproof_prove(_batch)), every proof "
                        "byte-identical to the oracle-made fixture" % ("wip_" if kind else "")}

    def side_batch(kind, m, P_side, what, fixture=None, nb=64):
        nm = 64 * m
        k = nm.bit_length() - 1
        g2, h2, G2, H2 = ctx.generate_bases(kind, bpa.KZEN_SEED, 64, m)
        gens2 = bpa.Gens(ctx, g2, h2, G2, H2, args.window_bits)
        pf2, cm2, plen2 = make_distinct_batch(ctx, gens2, G2, H2, kind, m, P_side, b"bpmsm-bench-side-%d-%d" % (kind, m))
        tamper_at = (33 * (2 * k + 3) + 64 + 31) if kind else TX_AT  # delta' / tx, last byte
        pb2 = bytearray(pf2)
        want2 = tamper_batch(pb2, plen2, P_side, tamper_at)
        d_p2 = torch.frombuffer(pb2, dtype=torch.uint8).to(dev)
        d_c2 = torch.frombuffer(bytearray(cm2), dtype=torch.uint8).to(dev)
        d_v4 = torch.full((P_side,), 77, dtype=torch.uint8, device=dev)
        d_v5 = torch.full((P_side,), 77, dtype=torch.uint8, device=dev)
        fn = ctx.wip_rangeproof_verify_batch_device if kind else ctx.rangeproof_verify_batch_device
        calls2 = [0]

        def step2():
            dv = (d_v4, d_v5)[calls2[0] & 1]  # one verdict buffer per call in flight
            calls2[0] += 1
            fn(gens2, 64, m, d_p2.data_ptr(), plen2, d_c2.data_ptr(), P_side, dv.data_ptr())

        for _ in range(2 * max(1, args.warmup)):
            step2()
        ctx.sync()
        for dv in (d_v4, d_v5):
            if bytes(dv.cpu().numpy().tobytes()) != bytes(want2) and not args.skip_check:
                raise SystemExit("side measurement: verdict mismatch (%s)" % what)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        ctx.sync()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        rlc2 = None
        if kind:  # the same batch without the tampered proofs through the random-linear-combination check, one context
            d_pv = torch.frombuffer(bytearray(pf2), dtype=torch.uint8).to(dev)
            seed2 = bytes((11 * i + 3) & 255 for i in range(32))
            fbk = ctx.wip_rangeproof_verify_batch_rlc_device(gens2, 64, m, d_pv.data_ptr(), plen2, d_c2.data_ptr(), P_side, seed2, d_v4.data_ptr())
            assert not fbk and not bool(d_v4.any().item()), "BP+ rlc: all-valid batch must pass the combined check"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.wip_rangeproof_verify_batch_rlc_device(gens2, 64, m, d_pv.data_ptr(), plen2, d_c2.data_ptr(), P_side, seed2, d_v4.data_ptr())
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            rlc2 = {"value": P_side * args.steps / dt3, "unit": "proofs/s", "ms_per_step": dt3 / args.steps * 1e3, "contexts_in_flight": 1,
                    "what": "bp_wip_rangeproof_verify_batch_rlc_device on the all-valid version of the batch"}
        nterms = 2 * nm + 2 * k + m + 5 if not kind else 2 * nm + 2 * k + m + 4
        res = {"value": P_side * args.steps / dt2, "unit": "proofs/s", "ms_per_step": dt2 / args.steps * 1e3, "proofs_per_step": P_side,
               "msm_terms_per_proof": nterms, "verifier_terms_per_s": P_side * args.steps / dt2 * nterms, "table_bytes": gens2.table_bytes,
               "what": what + " (%d distinct GPU-made proofs, 1/16 tampered, verdicts checked)" % P_side}
        if rlc2:
            res["rlc"] = rlc2
        if fixture and not args.no_prove:
            res["prover"] = prover_numbers(gens2, kind, fixture, nb)
        del d_p2, d_c2, d_v4, d_v5
        gens2.close()
        return res

    plus = c2 = c4 = msm = prove_c1 = None
    if world == 1 and not args.no_prove:
        prove_c1 = prover_numbers(gens, 0, ("bp_cases.json", 3), 4096)  # BASELINE.json configs[0]: n=64 m=1 prove (+ verify)
    if world == 1 and not args.no_bpplus:
        plus = side_batch(1, 1, P, "bp_wip_rangeproof_verify_batch_device: RangeProofWIP::aggregated_verify (Bulletproofs+), n=64 m=1")
    if world == 1 and not args.no_c2:
        c2 = side_batch(0, 16, max(256, P // 16), "bp_rangeproof_verify_batch_device at BASELINE.json configs[1]'s shape: "
                        "RangeProof::aggregated_verify, n=64 m=16", fixture=("bp_c2_n64_m16.json", None), nb=512)
    if world == 1 and not args.no_c4:
        c4 = side_batch(1, 64, max(64, P // 64), "bp_wip_rangeproof_verify_batch_device at BASELINE.json configs[3]'s shape: "
                        "RangeProofWIP::aggregated_verify (range_proof_wip.rs:388-609), n=64 m=64", fixture=("bpplus_c4_n64_m64.json", None), nb=256)
    if world == 1 and not args.no_msm:
        msm = msm_side(ctx, dev, args.msm_log2, max(args.steps, 10))

    if rank == 0:
        total_proofs = world * P * args.steps
        value = total_proofs / dt
        msm_ms = kms["bpv_msm"]
        ev = Evidence()
        ex_msm, su_msm, ex_at = ev.verify_msm()
        ex_dec, su_dec = ev.decompress()
        model = MAD32_PER_PROOF_MSM * P / (msm_ms * 1e-3) if msm_ms > 0 else 0.0
        # the counted figure belongs to the default configuration (>= 40960 proofs per launch: 2 lanes per proof; automatic table window)
        counted = P >= 40960 and not args.window_bits and not args.lanes and msm_ms > 0 and ex_msm is not None
        executed = ex_msm * P / (msm_ms * 1e-3) if counted else model
        survey = su_msm * P / (msm_ms * 1e-3) if counted else None
        by_count = {r["proofs_per_launch_per_gpu"]: r for r in sweep}
        out = {
            "metric": "verified range proofs/sec (secp256k1, n=64) at 1/2/4/8 GPUs; MSM Mpts/s",
            "value": value,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 (256-bit modular integer arithmetic, 9x29-bit limbs on v_mad_u64_u32)",
            "data": "synthetic: %d distinct GPU-made proofs per GPU (SHAKE-256 secrets / blindings / tapes), 1/16 tampered, verdicts checked" % P,
            "config": {"workload": "batch aggregated_verify of %d distinct independent n=64 m=1 Bulletproofs range proofs per GPU per step "
                                   "(BASELINE.json configs[2]), proofs sharded, no collective; the K steps are K asynchronous calls, of "
                                   "which the library keeps %d in flight per GPU (bp_verify_set_overlap)" % (P, 1 if args.no_overlap else 2),
                       "calls_in_flight": 1 if args.no_overlap else 2,
                       "proofs_per_gpu": P, "msm_terms_per_proof": 147, "table_bytes": gens.table_bytes,
                       "verifier_terms_per_s": value * 147, "batch_generation_s": t_gen},
            "kernel_ms": kms,
            "batch_sweep": sweep,
            "configs2_literal": {"proofs_per_s_at_512_per_gpu": by_count.get(512, {}).get("proofs_per_s"),
                                 "proofs_per_s_at_4096_per_gpu": by_count.get(4096, {}).get("proofs_per_s"),
                                 "what": "BASELINE.json configs[2] names 4096 proofs over 8 GPUs = 512 per GPU per launch; 4096 per GPU is "
                                         "the whole batch on one GPU"},
            "msm_mpts_per_s": msm["value"] if msm else None,
            "msm": msm,
            "tiled_256": tiled,
            "rlc_batch": rlc,
            "bpplus_batch": plus,
            "c2_batch": c2,
            "c4_batch": c4,
            "prove_c1": prove_c1,
            "provers": {"configs[0] n=64 m=1": prove_c1, "configs[1] n=64 m=16": (c2 or {}).get("prover"),
                        "configs[3] BP+ n=64 m=64": (c4 or {}).get("prover")},
            "roofline": {
                "bound": "valu_int32_mad",
                "kernel": "k_bpv_fix + k_bpv_msm (the verification MSM: fixed-generator half and proof-specific half, launched back to back)",
                "achieved": executed / 1e12,
                "peak": PEAK_MAD32_PER_S / 1e12,
                "unit": "TMAD32/s",
                "frac": executed / PEAK_MAD32_PER_S,
                "frac_executed": executed / PEAK_MAD32_PER_S if counted else None,
                "frac_survey_unit": survey / PEAK_MAD32_PER_S if counted else None,
                "frac_model": model / PEAK_MAD32_PER_S,
                "traffic": ev.verify_traffic() if (P == 65536 and not args.window_bits and not args.lanes) else None,
                "traffic_unit": "bytes per launch pair (PMC passes %s / %s: TCC_EA0 read requests x 64 B + write requests x 64 / 32 B)"
                                % (ev.files.get("pmc_read"), ev.files.get("pmc_write")),
                "executed_mad32_per_proof": ex_msm if counted else None,
                "survey_unit_mad32_per_proof": su_msm if counted else None,
                "model_mad32_per_proof": MAD32_PER_PROOF_MSM,
                "evidence_files": ev.files,
                "what": "frac = frac_executed: v_mad_u64_u32 lane-results the kernels execute per proof - field multiplications x 102 + squarings "
                        "x 65 (the MADs of the 9x29-limb ISA incl. the reduction), COUNTED by the instrumented build on a %s-proof launch and "
                        "read from profiles/%s - x proofs / launch time / measured peak; frac_survey_unit: the same counted field operations x "
                        "64 MAD32, SURVEY 8(d)'s definition of a field multiplication (8x8 schoolbook), i.e. without this implementation's "
                        "extra limb products; frac_model: SURVEY 8(d)'s Pippenger model (146 points, c=5), which charges doublings and "
                        "buckets a table-based verifier does not execute, so it is not a bound for this kernel" % (ex_at, ev.files.get("opcount")),
                "proofs_per_launch": P,
                "avg_launch_ms": msm_ms,
                "avg_launch_ms_what": "k_bpv_fix + k_bpv_msm of one call with the GPU to themselves (HIP events on the library's stream, calls "
                                      "one at a time: per-stage timing switches the overlap of successive calls off); agrees with "
                                      "the newest profiles/rNN_bench_kernel_stats_serial.csv (rocprofv3 of bench.py --no-overlap).  In the timed region two "
                                      "calls are in flight and each kernel's wall time stretches while it shares the GPU "
                                      "(profiles/rNN_bench_kernel_stats.csv), so the rate of the whole pipelined step is given beside it",
                "frac_whole_step": ((ex_msm + (ex_dec or 0.0)) if counted else MAD32_PER_PROOF_MSM) * value / world / PEAK_MAD32_PER_S,
                "frac_whole_step_survey_unit": ((su_msm + (su_dec or 0.0)) * value / world / PEAK_MAD32_PER_S) if counted else None,
                "frac_whole_step_what": "counted MADs per proof of the MSM pair AND the decompression (the scalar kernels' mod-q multiplies are "
                                        "not counted) x proofs/s per GPU / peak: the whole pipelined step as the timed region runs it; equals "
                                        "value x (executed_mad32_per_proof + decompression) / peak, so it can be checked against the driver's clock",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(golden, args.cpu_sample)
        print(json.dumps(out), flush=True)
    gens.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
