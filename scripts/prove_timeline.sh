#!/bin/bash
# kernel timeline of the last single-proof prove (or verify: $1 = verify) of prove_once.py; $2 = kernels to print; $3 = c1 | c2 | c4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlp
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tlp -o tl --output-format csv -- python3 $R/scripts/experiments/prove_once.py ${1:-prove} ${3:-c1} > /tmp/op.log 2>&1 || { tail -20 /tmp/op.log; exit 1; }
grep "^prove" /tmp/op.log
python3 $R/scripts/experiments/timeline2.py /tmp/tlp ${2:-40}
