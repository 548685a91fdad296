"""A/B of the plain transposed GEMM epilogue (the encoder's V^T projection: 16 chunks x 1500 x 1280 x 1280, Ct row stride
1536) in one process: fw_test_knob 5 = 1 (staged through LDS, whole row segments) against 0 (direct 8-byte stores).

    python profiles/vt_epilogue_ab.py [--rounds 3] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from faster_whisper_amd import Whisper, _lib, get_config, synthetic_weights
    cfg = get_config("micro")
    model = Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": synthetic_weights(cfg, 3)},
                    compute_type="float16", max_batch_size=1, max_beam_size=1)
    lib = _lib.load()
    h = model._replicas[0].handle
    M, N, K, B = 1500, 1280, 1280, 16
    fl = 2.0 * B * M * N * K
    for rnd in range(a.rounds):
        for knob in (0, 1):
            _lib.check(lib.fw_test_knob(5, knob))
            rec = {"round": rnd, "vt_stage": knob}
            for name, tr in (("v^T (transposed)", 1), ("out (row-major, same shape)", 0)):
                ms = C.c_float()
                _lib.check(lib.fw_bench_gemm(h, M, N, K, B, 0, 0, tr, a.iters, C.byref(ms)))
                rec[name] = {"us": round(1e3 * ms.value, 1), "TFLOP/s": round(fl / ms.value / 1e9, 1)}
            print(json.dumps(rec), flush=True)
    _lib.check(lib.fw_test_knob(5, 1))


if __name__ == "__main__":
    main()
