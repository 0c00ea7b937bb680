"""The driver's round-end entry: __graft_entry__.smoke() must keep passing as fixtures and verdict codes grow
(it broke silently once when the Bulletproofs+ fixtures gained Malformed variants)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.smoke()


def test_smoke_knows_every_verdict_name_in_the_fixtures():
    """CPU side of the same guard: the verdict names smoke() maps cover the ones the golden files use."""
    import json
    import re
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    known = set(re.findall(r'"(\w+)": \d+', src[src.index("codes = {"):src.index("}", src.index("codes = {"))]))
    used = set()
    for name in ("bp_cases.json", "bpplus_cases.json"):
        for case in json.load(open(os.path.join(ROOT, "tests", "golden", name))):
            used.add(case["verdict"])
            used.update(t["verdict"] for t in case["tampered"])
    assert used <= known, used - known
