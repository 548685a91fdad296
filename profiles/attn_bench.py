"""Encoder self-attention micro-benchmark (fw_bench_attention): large-v3 geometry, 16 chunks x 20 heads x 1500 positions.
    python profiles/attn_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_whisper_amd import Whisper, _lib, get_config, synthetic_weights  # noqa: E402


def main():
    cfg = get_config("micro")
    m = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": synthetic_weights(cfg, seed=1)},
                max_batch_size=1, max_beam_size=1)
    h = m._replicas[0].handle
    ms = C.c_float()
    B, H, T = 16, 20, 1500
    fl = 4.0 * B * H * T * T * 64
    only = int(sys.argv[1]) if len(sys.argv) > 1 else -1      # one variant only (for a counter pass per variant)
    for rnd in range(3):          # interleaved A/B in one process (variant: attn_enc.hip workgroup mapping)
        for variant, what in ((0, "XCD-aware grid"), (1, "round-3 grid")):
            if only >= 0 and variant != only:
                continue
            _lib.check(m._lib.fw_bench_attention(h, B, H, T, variant, 50, C.byref(ms)))
            print(f"round {rnd} {what}: {ms.value * 1e3:.1f} us per launch, {fl / ms.value / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
