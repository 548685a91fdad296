"""Round 6, verdict item 1a: do complementary CU masks make the two roofs overlap?

ONE process, ONE box, one packed weight blob; for every setting a fresh model is built with the encoder streams and the
decode-lane streams created under the setting's CU masks (engine.hip: create_stream, FWAMD_ENC_CUS / FWAMD_DEC_CUS), then

  * steady state: `--steps` batches over 32 workers, two decode lanes (the bench's own loop)  -> x real time
  * one profiled round (HIP events per kernel family, ONE lane)                               -> ms per batch per family,
    i.e. what dec_cross_attn (HBM-bound) and the GEMM families (MFMA-bound) do on the CUs the mask leaves them
  * one 16-chunk batch at a time (solo decode runs)                                           -> ms per batch

    python profiles/ab_r06_overlap.py [--steps 96] [--only base,dec96_enc-96] >> profiles/r06_ab_overlap.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402

SETTINGS = [
    ("base", {}),
    ("dec128_enc-128", {"FWAMD_DEC_CUS": "128", "FWAMD_ENC_CUS": "-128"}),
    ("dec96_enc-96", {"FWAMD_DEC_CUS": "96", "FWAMD_ENC_CUS": "-96"}),
    ("dec64_enc-64", {"FWAMD_DEC_CUS": "64", "FWAMD_ENC_CUS": "-64"}),
    ("dec160_enc-160", {"FWAMD_DEC_CUS": "160", "FWAMD_ENC_CUS": "-160"}),
    ("dec128", {"FWAMD_DEC_CUS": "128"}),
    ("dec96", {"FWAMD_DEC_CUS": "96"}),
    ("dec64", {"FWAMD_DEC_CUS": "64"}),
    ("enc160", {"FWAMD_ENC_CUS": "-96"}),
    ("enc128", {"FWAMD_ENC_CUS": "128"}),
    ("base_again", {}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from faster_whisper_amd import Whisper, get_config, pack_blob, synthetic_weights
    args = bench.parse_args(["--workers", str(a.workers), "--steps", str(a.steps)])
    cfg = get_config(args.model)
    blob = pack_blob(cfg, synthetic_weights(cfg, seed=1234), 0)
    dev_blob = torch.from_numpy(blob).cuda()
    del blob
    chunks = bench.synth_chunks(args.batch, seed=1000)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    L = args.new_tokens
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=args.beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + L, return_scores=True,
              return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup, min_new_tokens=L)
    only = [s for s in a.only.split(",") if s]
    ref_ids = None
    for name, env in SETTINGS:
        if only and name not in only:
            continue
        for k in ("FWAMD_DEC_CUS", "FWAMD_ENC_CUS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        model = Whisper(f"synthetic:{args.model}", device="cuda", blob_dev=(dev_blob.data_ptr(), dev_blob.numel()),
                        max_batch_size=args.batch, max_beam_size=args.beam, inter_threads=a.workers)
        staged = model.stage_pcm(chunks)
        pool = ThreadPoolExecutor(max_workers=a.workers)

        def step():
            return model.generate(model.encode_pcm_staged(staged), [prompt] * args.batch, **kw)

        def merged(n):
            model.synchronize()
            t0 = time.perf_counter()
            outs = [f.result() for f in [pool.submit(step) for _ in range(n)]]
            model.synchronize()
            return time.perf_counter() - t0, outs[-1]

        merged(a.workers)                         # warm: graphs, pools
        dt, res = merged(a.steps)
        ids = [r.sequences_ids for r in res]
        if ref_ids is None:
            ref_ids = ids
        rec = {"setting": name, "env": env, "steady_x": round(30.0 * args.batch * a.steps / dt, 1),
               "ms_per_step": round(1e3 * dt / a.steps, 2), "same_ids_as_first_setting": ids == ref_ids}
        # one profiled round, one lane
        model.set_decode_lanes(1)
        model.profile(True, replica=None)
        merged(a.workers)
        rep = model.profile_report(replica=None)
        model.profile(False, replica=None)
        model.set_decode_lanes(2)
        fam = {k: round(v["ms"] / a.workers, 3) for k, v in rep.items() if v["ms"] > 0}
        rec["families_ms_per_step"] = fam
        rec["families_sum_ms"] = round(sum(fam.values()), 2)
        ca = rep.get("dec_cross_attn")
        if ca and ca["ms"] > 0:
            rec["dec_cross_attn_GBps"] = round(ca["bytes"] / (ca["ms"] * 1e-3) / 1e9, 1)
        eg = rep.get("enc_gemm")
        if eg and eg["ms"] > 0:
            rec["enc_gemm_TFLOPs"] = round(eg["flops"] / (eg["ms"] * 1e-3) / 1e12, 1)
        # solo: one batch at a time
        step()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        rec["one_batch_ms"] = round(1e3 * (time.perf_counter() - t0) / 3, 1)
        print(json.dumps(rec), flush=True)
        pool.shutdown()
        model.free_staged(staged)
        model.unload_model()
        del model


if __name__ == "__main__":
    main()
