"""The native FLAC decoder (csrc/flac_host.cpp) is the one parser of this repository that reads
untrusted bytes.  tests/native/flac_fuzz.cpp builds it under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build — the
task's rule for sanitizers) and feeds it the reference's own FLAC fixture and thousands of seeded mutants of it (truncations,
forged STREAMINFO and frame headers, noise bursts, splices, repeated streams, too-small output buffers): any status is fine,
a sanitizer report is not.  The host Silero VAD network (csrc/vad_host.cpp) runs under the same sanitizers on exact-size
buffers (tests/native/vad_asan.cpp).  (Round 6 found one this way: a corrupted LPC / fixed predictor recursion diverged into signed
64-bit overflow; the predictors now run in wrapping arithmetic and the frame CRC rejects the frame.)  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "native", "flac_fuzz.cpp"), os.path.join(ROOT, "faster_whisper_amd", "csrc", "flac_host.cpp")]
FIXTURE = os.path.join(ROOT, "tests", "golden", "flac_jfk_head.flac")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("flac_fuzz") / "flac_fuzz")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", *SRC, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitize" in r.stderr or "asan" in r.stderr.lower():
            pytest.skip("this g++ has no sanitizer runtimes: " + r.stderr[-200:])
        raise AssertionError(r.stderr[-2000:])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mutated_streams_under_asan_ubsan(harness, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([harness, FIXTURE, "1500", str(seed)], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout.strip())
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "no sanitizer report" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    # the mutants exercise both outcomes
    words = r.stdout.split()
    decoded, refused = int(words[words.index("mutants:") + 1]), int(words[words.index("refused;") - 1])
    assert decoded > 50 and refused > 50


def test_host_vad_network_under_asan_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "vad_asan")
    src = [os.path.join(ROOT, "tests", "native", "vad_asan.cpp"), os.path.join(ROOT, "faster_whisper_amd", "csrc", "vad_host.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        *src, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitize" in r.stderr or "asan" in r.stderr.lower():
            pytest.skip("this g++ has no sanitizer runtimes: " + r.stderr[-200:])
        raise AssertionError(r.stderr[-2000:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout.strip())
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
