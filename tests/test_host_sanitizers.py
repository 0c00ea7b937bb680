"""Host C++ under sanitizers: `make san` builds
  - the host-compiled arithmetic headers' self-test with -fsanitize=address,undefined (host_selftest_san.cpp),
  - the batch provers' thread rendezvous (fbbatcher.h, used by proto.cpp's FbBatcher) with -fsanitize=thread and with
    -fsanitize=address,undefined (host_batcher_san.cpp),
  - the single-proof calls' host worker pool (hostpool.h) with -fsanitize=thread and with -fsanitize=address,undefined
    (host_pool_san.cpp: passes of many sizes back to back and after pauses, several caller threads at once),
  - the serde wire-format codec, which parses untrusted JSON / bincode, with -fsanitize=address,undefined
    (serde_codec_san.cpp: round trips, every truncation and 20 000 single-byte corruptions of valid encodings),
and every binary must exit 0 with no sanitizer report.  CPU only (g++); the GPU build is not sanitized (no GPU ASan on
this pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bulletproofs_amd", "csrc")


@pytest.fixture(scope="module")
def san_dir():
    subprocess.check_call(["make", "-C", CSRC, "san"], stdout=subprocess.DEVNULL)
    return os.path.join(CSRC, "_san")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("binary,marker", [("host_selftest_asan", "host_selftest_san OK"), ("host_batcher_tsan", "host_batcher_san OK"),
                                           ("host_batcher_asan", "host_batcher_san OK"), ("serde_codec_asan", "serde_codec_san OK"),
                                           ("host_pool_tsan", "host_pool_san OK"), ("host_pool_asan", "host_pool_san OK")])
def test_host_code_is_clean_under_sanitizers(san_dir, binary, marker):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    # (the sanitizer runtimes are linked statically - Makefile SANFLAGS - so the environment's library order does not matter)
    out = subprocess.run([os.path.join(san_dir, binary)], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert marker in out.stdout
    assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
