"""Per-kernel time and idle gaps of the LAST batch in a rocprofv3 rocpd database of scripts/experiments/prove_batch_sweep.py."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels order by start"))
i0 = [i for i, r in enumerate(rows) if "k_pv_init" in r[0] or "k_pw_init" in r[0]][-1]
t0 = rows[i0][1]
agg = collections.OrderedDict()
for n, s, e in rows[i0:]:
    n = n.split("(")[0].replace("bp::", "").replace("void ", "")
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = 0
for n, (k, t) in agg.items():
    print("%-28s %3d launches %9.1f us" % (n[:28], k, t / 1e3))
    tot += t
print("gpu busy %.1f us of %.1f us" % (tot / 1e3, (rows[-1][2] - t0) / 1e3))
