"""Encoder-GEMM micro-benchmark through the C ABI measurement hook (fw_bench_gemm): the large-v3 encoder shapes at 16
chunks per batch, TFLOP/s per shape.  Operand leading dimensions can be padded (stride experiments).

    python profiles/gemm_bench.py [--int8] [--pad 0,64]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (name, M per chunk, N, K, trans)
    ("conv1", 3000, 1280, 384, 0), ("conv2", 1500, 1280, 3840, 0), ("qk", 1500, 2560, 1280, 0),
    ("v^T", 1500, 1280, 1280, 1), ("out", 1500, 1280, 1280, 0), ("ffn1", 1500, 5120, 1280, 0),
    ("ffn2", 1500, 1280, 5120, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--int8", action="store_true")
    ap.add_argument("--pad", default="0")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from faster_whisper_amd import Whisper, _lib, get_config, synthetic_weights
    cfg = get_config("micro")
    model = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": synthetic_weights(cfg, 3)},
                    compute_type="int8_float16" if args.int8 else "float16", max_batch_size=1, max_beam_size=1)
    lib = _lib.load()
    h = model._replicas[0].handle
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("FWAMD_")}, "int8": args.int8}
    for pad in [int(x) for x in args.pad.split(",")]:
        tot_ms = tot_fl = 0.0
        weights = {"conv1": 1, "conv2": 1, "qk": 32, "v^T": 32, "out": 32, "ffn1": 32, "ffn2": 32}
        for name, M, N, K, tr in SHAPES:
            ms = C.c_float()
            _lib.check(lib.fw_bench_gemm(h, M, N, K, args.batch, pad, pad, tr, args.iters, C.byref(ms)))
            fl = 2.0 * args.batch * M * N * K
            out[f"{name} pad={pad}"] = {"ms": round(ms.value, 4), "TFLOP/s": round(fl / ms.value / 1e9, 1)}
            tot_ms += ms.value * weights[name]
            tot_fl += fl * weights[name]
        out[f"encoder-weighted pad={pad}"] = {"ms": round(tot_ms, 2), "TFLOP/s": round(tot_fl / tot_ms / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
