"""The oracle's own memory safety: oracle/*.c built with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`) runs the oracle-vs-reference
suites (tests/test_oracle.py: fixtures of all corpora, the edge segment, trees, masks; tests/test_fastpfor.py) in a child interpreter with the sanitizer
runtimes preloaded.  A checker that reads out of bounds or relies on undefined arithmetic would pin nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_runtime("libasan.so") is None or _runtime("libubsan.so") is None, reason="gcc's sanitizer runtimes not found")
def test_oracle_suites_pass_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    env = dict(os.environ)
    env.update(LD_PRELOAD=f"{_runtime('libasan.so')} {_runtime('libubsan.so')}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               TRINITY_ORACLE_LIB=os.path.join(ROOT, "oracle", "_ref", "liboracle_asan.so"))  # fmt: skip
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_fastpfor.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)  # fmt: skip
    tail = r.stdout[-2500:] + r.stderr[-2500:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error:" not in r.stdout + r.stderr, tail
    assert " passed" in r.stdout and "liboracle_asan.so" in env["TRINITY_ORACLE_LIB"]


@pytest.mark.skipif(_runtime("libasan.so") is None or _runtime("libubsan.so") is None, reason="gcc's sanitizer runtimes not found")
def test_host_tools_pass_under_asan_and_ubsan(tmp_path):
    """libtrinity_host.so's sources — the host planner (the code tri_batch_create runs, csrc/planner.hpp, fragments recycled or fresh), the byte-exact
    encoders of both codecs, the Lucene encoder's units (the device kernels' bodies, lucene_enc_units.hpp / pfor128_group.hpp), the FastPFor restatement
    and the transcoder, the corpus generator — built with ASan + UBSan and run through their CPU suites."""
    from trinity_amd.build import HOST_SRCS

    lib = str(tmp_path / "libtrinity_host_asan.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                    "-o", lib] + HOST_SRCS, check=True)  # fmt: skip
    env = dict(os.environ)
    env.update(LD_PRELOAD=f"{_runtime('libasan.so')} {_runtime('libubsan.so')}", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", TRINITY_HOST_LIB=lib)
    suites = [os.path.join(ROOT, "tests", f) for f in ("test_planner.py", "test_fastpfor.py", "test_golden_merge.py")]
    r = subprocess.run([sys.executable, "-m", "pytest"] + suites + ["-x", "-q", "-s", "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = r.stdout[-2500:] + r.stderr[-2500:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error:" not in r.stdout + r.stderr, tail
