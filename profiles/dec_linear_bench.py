"""Tile-shape micro-benchmark of the decoder linear kernel (fw_bench_dec_linear): microseconds per launch for the
large-v3 decode-step shapes at the row counts of solo (80) and merged (320, 640) decode runs.
    python profiles/dec_linear_bench.py > gpurun_out/dec_linear_bench.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_whisper_amd import Whisper, _lib, get_config, synthetic_weights  # noqa: E402

VARIANTS = {0: "4w 2x2 ch5 (product K<2560)", 1: "8w 2x2 ch5 (product K>=2560)", 2: "8w 4x2 ch3", 3: "8w 4x2 ch4",
            4: "8w 4x4 ch2", 5: "8w 4x4 ch3", 6: "4w 4x2 ch3", 7: "8w 2x4 ch3", 8: "8w 8x2 ch2", 9: "4w 4x4 ch2",
            10: "big 4x2 waves 256x128 KC2x3", 11: "big 2x2 waves 128x64 KC1x5", 12: "big 2x2 waves 128x128 KC1x4",
            13: "sp 4 slices x 2x2 waves 128x64 x3",
            21: "4w 4x4 ch5", 22: "8w 4x4 ch5", 23: "4w 1x1 ch10", 24: "8w 1x1 ch10", 25: "4w 1x2 ch6", 26: "8w 1x2 ch6"}
if os.environ.get("DLB_VARIANTS"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["DLB_VARIANTS"].split(",")}
SHAPES = [("qkv", 3840, 1280, 1), ("dxd", 1280, 1280, 0), ("ffn1", 5120, 1280, 1), ("ffn2", 1280, 5120, 0)]
if os.environ.get("DLB_LNF") is not None:      # force the LayerNorm-folded form on / off for every shape (what the statistics cost)
    SHAPES = [(n, N, K, int(os.environ["DLB_LNF"])) for n, N, K, _ in SHAPES]


def main():
    cfg = get_config("micro")
    m = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": synthetic_weights(cfg, seed=1)},
                max_batch_size=1, max_beam_size=1)
    h = m._replicas[0].handle
    us = C.c_float()
    rows = [int(a) for a in sys.argv[1:]] or [80, 320, 640]
    for R in rows:
        print(f"R = {R}")
        print("  %-38s" % "variant" + "".join("%10s" % s[0] for s in SHAPES) + "   layer (qkv + 3 dxd + ffn1 + ffn2)   TFLOP/s")
        for v, name in VARIANTS.items():
            t = []
            for _, N, K, lnf in SHAPES:
                rc = m._lib.fw_bench_dec_linear(h, R, N, K, lnf, v, 400, C.byref(us))
                t.append(us.value if rc == 0 else float("nan"))     # (a form that does not take this shape)
            layer = t[0] + 3 * t[1] + t[2] + t[3]
            flop = 2.0 * R * sum(n * k * (3 if nm == "dxd" else 1) for nm, n, k, _ in SHAPES)
            print("  %-38s" % name + "".join("%10.2f" % x for x in t) + "   %8.1f   %8.0f" % (layer, flop / layer / 1e6), flush=True)


if __name__ == "__main__":
    main()
