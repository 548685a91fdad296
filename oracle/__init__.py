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
    (transcribe.py:222-236, :1433-1459, :1709-1715, :1823) and on cross-checks against the
    installed `transformers` Whisper implementation: architecture and a whole greedy decode with
    timestamps — tokens, score, no-speech probability — against a loop made of transformers' forward
    and transformers' own logits processors, and a whole BEAM-5 decode — all five hypotheses, their order
    and scores, finishing in different steps — against transformers' own beam search with early stopping
    (tests/test_oracle_arch.py),
    timestamp / suppress logits rules, median filter and DTW (tests/test_oracle_vs_hf_rules.py:
    identical on random inputs).  Beam-search bookkeeping and int8 conventions stay [CT2-ext].
  * host logic of the callers (prompts, seek loop, temperature fallback, word timestamps, VAD
    state machine, chunk merging; `oracle.scripted_backend`, `oracle.micro_tokenizer`,
    `oracle.host_scenarios`): PINNED — `oracle/gen_golden_host.py` runs the reference's own
    transcribe.py / tokenizer.py / vad.py in the build container; fixtures under `tests/golden/`.
  * Silero VAD network (`oracle.silero`): PARITY vs onnxruntime UNPINNED (absent); restates the ONNX
    graph of the reference's asset, plausibility-checked on the reference's speech fixture and pinned
    to a GENERIC execution of that graph by `oracle.onnx_exec` (every node by its ONNX operator
    definition on torch's kernels; tests/test_oracle_silero_graph.py: 5.8e-7 with the real weights).
"""
