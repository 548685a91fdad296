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
    ap.add_argument("--order", type=int, default=-1, help="tile order of every launch (1 blocked, 0 n fastest); default: A/B")
    ap.add_argument("--square", default="",
                    help="comma-separated sizes n: time the kernel at n x n x n (batch 1) — the guide quotes its 256^2 "
                         "8-phase template at 4096^3 / 8192^3 on random operands (cdna_hip_programming.md), where a launch "
                         "is exactly one or four full rounds of 256 workgroups and K is 64-128 tiles deep; the encoder "
                         "shapes are 1.9-7.5 rounds and 6-80 K tiles")
    args = ap.parse_args()
    from faster_whisper_amd import Whisper, _lib, get_config, synthetic_weights
    cfg = get_config("micro")
    model = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": synthetic_weights(cfg, 3)},
                    compute_type="int8_float16" if args.int8 else "float16", max_batch_size=1, max_beam_size=1)
    lib = _lib.load()
    h = model._replicas[0].handle
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("FWAMD_")}, "int8": args.int8}
    if args.square:
        for n in [int(x) for x in args.square.split(",")]:
            ms = C.c_float()
            _lib.check(lib.fw_bench_gemm(h, n, n, n, 1, 0, 0, 0, args.iters, C.byref(ms)))
            out[f"square {n}"] = {"ms": round(ms.value, 4), "TFLOP/s": round(2.0 * n * n * n / ms.value / 1e9, 1)}
        print(json.dumps(out, indent=1))
        return
    weights = {"conv1": 1, "conv2": 1, "qk": 32, "v^T": 32, "out": 32, "ffn1": 32, "ffn2": 32}
    for pad in [int(x) for x in args.pad.split(",")]:
        # tile order A/B, interleaved shape by shape in one process (fw_test_knob 1: 1 = blocked, 0 = n fastest)
        tot = {1: [0.0, 0.0], 0: [0.0, 0.0]}
        for name, M, N, K, tr in SHAPES:
            for order in ((1, 0, 1, 0) if args.order < 0 else (args.order,)):
                _lib.check(lib.fw_test_knob(1, order))
                ms = C.c_float()
                _lib.check(lib.fw_bench_gemm(h, M, N, K, args.batch, pad, pad, tr, args.iters, C.byref(ms)))
                fl = 2.0 * args.batch * M * N * K
                key = f"{name} pad={pad} order={order}"
                prev = out.get(key)
                if prev is None or ms.value < prev["ms"]:
                    out[key] = {"ms": round(ms.value, 4), "TFLOP/s": round(fl / ms.value / 1e9, 1)}
            for order in ((1, 0) if args.order < 0 else (args.order,)):
                tot[order][0] += out[f"{name} pad={pad} order={order}"]["ms"] * weights[name]
                tot[order][1] += 2.0 * args.batch * M * N * K * weights[name]
        for order in ((1, 0) if args.order < 0 else (args.order,)):
            out[f"encoder-weighted pad={pad} order={order}"] = {"ms": round(tot[order][0], 2),
                                                               "TFLOP/s": round(tot[order][1] / tot[order][0] / 1e9, 1)}
    _lib.check(lib.fw_test_knob(1, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
