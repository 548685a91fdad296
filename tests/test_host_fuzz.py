"""Differential fuzzing of the host logic against the REFERENCE's own code (oracle/fuzz_host.py): random recordings
and random transcribe() argument combinations through both implementations on the scripted backend — segments,
words, info, backend call logs and raised error types must be equal.  Needs the reference checkout (build
container); a subprocess keeps the stub modules out of this session.  `python oracle/fuzz_host.py --seeds 300` was
clean at the end of round 1 (1514 segments, 12761 words, 1151 generate calls); at the end of round 3 a fresh range,
`--seeds 300 --start 5000 --units 2000 --vad 1000 --logmel 40`, was clean as well (1597 segments, 12372 words, 1134
generate and 437 align calls, 10 error cases raising the same type on both sides).  No GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/faster_whisper"), reason="reference checkout not on this box")
def test_fuzz_against_reference_host_code():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "fuzz_host.py"), "--seeds", "40", "--start", "1000", "--units", "600", "--vad", "400", "--logmel", "24"],
                       capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1]
    stats = json.loads(last)
    assert r.returncode == 0 and stats["mismatches"] == 0 and stats["unit_mismatches"] == 0 \
        and stats["vad_mismatches"] == 0 and stats["logmel_mismatches"] == 0, r.stdout[-3000:]
    assert stats["segments"] > 100 and stats["generate"] > 100
