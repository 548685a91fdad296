"""CPU oracle for the Whisper hot path — TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package.  The product (`faster_whisper_amd`) never does: without the
HIP library it fails loudly instead of falling back to this code.

Pinning status
  * log-mel (`oracle.logmel`, `oracle/logmel_ref.c`): PINNED against the reference's own
    `faster_whisper/feature_extractor.py`, imported in the build container by
    `oracle/gen_golden.py`; fixtures committed under `tests/golden/`.
  * everything behind `ctranslate2.models.Whisper` (`oracle.whisper`): PARITY UNPINNED.
    CTranslate2 is a third-party dependency (`ctranslate2>=4.0,<5`, requirements.txt:1)
    that is neither in /root/reference nor installed, and no Whisper checkpoint is on
    disk.  The restatement follows openai-whisper / CTranslate2 4.x published behaviour
     and is anchored on the reference's call sites
    (transcribe.py:222-236, :1433-1459, :1709-1715, :1823) and on an architecture
    cross-check against the installed `transformers` Whisper implementation
    (tests/test_oracle_arch.py).
"""
