"""rust/patches/*.patch: the patch set that routes the reference crate's folds through libbpmsm.so.
No Rust toolchain exists here, so what CAN be checked is checked: every hunk's context and removed lines still match
/root/reference (skipped where the reference is absent, e.g. on the GPU box), the generator reproduces the committed
files, and every accel:: function a patch calls exists in the shim with the wrapper-crate calls it needs."""
import glob, os, re, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "rust", "patches", "00*.patch")))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not PATCHES, reason="rust/ is not shipped to the GPU box (.gpurunignore)")


def hunks(text):
    """[(old_path, old_start, [lines])] per hunk of a unified diff"""
    path, cur, out = None, None, []
    for line in text.splitlines():
        if line.startswith("--- "):
            path = line[4:].split("\t")[0]
            continue
        if line.startswith("+++ "):
            continue
        m = re.match(r"@@ -(\d+)(?:,(\d+))? \+(\d+)(?:,(\d+))? @@", line)
        if m:
            cur = (path, int(m.group(1)), [])
            out.append(cur)
        elif cur is not None:
            cur[2].append(line)
    return out


def test_patch_set_present():
    assert len(PATCHES) == 5
    for p in PATCHES:
        assert open(p).read().count("\n@@") + open(p).read().startswith("@@") >= 1


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_every_hunk_applies_to_the_reference():
    n = 0
    for p in PATCHES:
        for path, start, body in hunks(open(p).read()):
            if path == "/dev/null":
                continue
            src = open(os.path.join(REF, path[2:])).read().split("\n")
            k = start - 1
            for line in body:
                if line.startswith("+") or line.startswith("\\"):
                    continue
                assert line[:1] in (" ", "-"), (p, line)
                assert src[k] == line[1:], "%s: %s line %d: %r != %r" % (os.path.basename(p), path, k + 1, src[k], line[1:])
                k += 1
            n += 1
    assert n >= 30


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_generator_reproduces_committed_patches(tmp_path):
    before = {p: open(p).read() for p in PATCHES}
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_rust_patches.py")], check=True, capture_output=True)
    for p in PATCHES:
        assert open(p).read() == before[p], p


def test_patches_only_call_what_the_shim_and_wrapper_define():
    shim = open(os.path.join(ROOT, "rust", "patches", "accel.rs")).read()
    wrapper = open(os.path.join(ROOT, "rust", "bpmsm", "src", "lib.rs")).read()
    used = set()
    for p in PATCHES:
        for line in open(p).read().splitlines():
            if line.startswith("+") and not line.startswith("+++") and "src/accel.rs" not in line:
                used |= set(re.findall(r"crate::accel::(\w+)", line))
    assert used == {"msm", "msm2", "fold_bases", "random_scalar", "signed_unit_sum", "scalar_mul_batch", "generator_mul", "fixed_base_mul"}
    for f in used:
        assert re.search(r"pub fn %s\b" % f, shim), f
    for f in re.findall(r"\bc\.(\w+)\(", shim):
        assert re.search(r"pub fn %s\b" % f, wrapper), f
    # no protocol-file call site of Scalar::random() is left outside the tests (lines the patches do not touch are checked
    # against the reference when it is present)
    if os.path.isdir(REF):
        for f, last in (("range_proof.rs", 263), ("range_proof_wip.rs", 274), ("weighted_inner_product.rs", 268), ("inner_product.rs", 169)):
            src = open(os.path.join(REF, "src", "proofs", f)).read().split("\n")[:last]
            sites = [i + 1 for i, l in enumerate(src) if "Scalar::<Secp256k1>::random()" in l]
            patched = set()
            for p in PATCHES:
                for path, start, body in hunks(open(p).read()):
                    if path.endswith("/" + f):
                        k = start
                        for line in body:
                            if line.startswith("-"):
                                patched.add(k)
                            if not line.startswith("+"):
                                k += 1
            assert set(sites) <= patched, (f, sorted(set(sites) - patched))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_every_patch_point_of_integration_md_has_a_hunk():
    """Every `file:line(-line)` range of INTEGRATION.md section 2's table lies inside the removed lines of some hunk
    (the patch set covered 6 of the 11 rows)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("| file:line | today | becomes |"):doc.index("Two integration depths.")]
    removed = {}
    for p in PATCHES:
        for path, start, body in hunks(open(p).read()):
            if path == "/dev/null":
                continue
            k = start
            for line in body:
                if line.startswith("-"):
                    removed.setdefault(os.path.basename(path), set()).add(k)
                if not line.startswith("+"):
                    k += 1
    rows = 0
    for row in table.splitlines()[2:]:
        if not row.startswith("|"):
            continue
        cell = row.split("|")[1]
        fname = None
        for tok in re.findall(r"`([^`]+)`", cell):
            for part in tok.split(","):
                part = part.strip()
                m = re.match(r"(?:src/proofs/)?(\w+\.rs):(.*)", part)
                if m:
                    fname, part = m.group(1), m.group(2)
                m = re.match(r"(\d+)(?:-(\d+))?$", part.strip())
                assert m and fname, (row, part)
                lo, hi = int(m.group(1)), int(m.group(2) or m.group(1))
                hit = [n for n in range(lo, hi + 1) if n in removed.get(fname, set())]
                assert hit, "no hunk removes a line of %s:%d-%d" % (fname, lo, hi)
        rows += 1
    assert rows >= 11
