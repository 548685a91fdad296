"""Is 1e-3 on the beam score attainable for compute_type int8_float16 at large-v3 depth?  CPU only, ~20 minutes.

    python tests/numerics_int8_order_noise.py > tests/golden/numerics_int8_order_noise.txt

BASELINE.json's north star asks for log-probs within 1e-3 of the reference path; the int8_float16 rows of the EXCEPTIONS
table in tests/test_gpu_full_size.py list 1e-2 on the 48-step beam score (3.7e-3 measured on the box).  The cause claimed
there: engine and oracle quantise activations that differ by fp16 rounding, and a flipped int8 code is a step of 1/127 of
the row's absmax.  This script measures that claim WITHOUT the engine: it evaluates the oracle's int8_float16 restatement
twice, the second time with the fp32 SUMMATION ORDER of every reduction that feeds an fp16 rounding reversed (LayerNorm
statistics, QK^T, the softmax denominator, PV — the feature / key axis is flipped before the reduction and the result
flipped back; the integer GEMMs are exact in any order).  Both are valid evaluations of the same int8_float16 model: they
differ by ~1e-7 relative before rounding, i.e. only in which side of an fp16 rounding boundary — and then of an int8
quantisation boundary — a value falls.  The same experiment on the float16 restatement is the control.

What the output shows is the noise floor of the quantity the north star constrains, for ANY two implementations that do
not share their summation order (the HIP engine and CTranslate2's CPU kernels do not): if the two orders of the SAME
restatement already differ by more than 1e-3, no engine can meet 1e-3 against the reference on that quantity, and the
listed exception is permanent; if they do not, the exception is the engine's to fix.
Not a test (pytest does not collect it): it measures, it asserts nothing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import bench_audio  # noqa: E402
from faster_whisper_amd import get_config, synthetic_weights  # noqa: E402
from oracle import logmel as olm  # noqa: E402
from oracle.whisper import OracleWhisper  # noqa: E402


class ReversedOrder(OracleWhisper):
    """the same model, every fp32 reduction that feeds an fp16 rounding summed over the flipped axis"""

    def _ln(self, x, p):
        if self.fold_ln and not self.int8 and p.startswith("dec."):
            return super()._ln(x, p)
        y = torch.nn.functional.layer_norm(x.flip(-1), (self.d,), self.w[p + ".g"].flip(-1), self.w[p + ".b"].flip(-1), 1e-5)
        return self._r(y.flip(-1))

    def _attn(self, q, k, v, mask=None, return_probs=False):
        if q.shape[0] > 1 and k.shape[0] == q.shape[0] and k.stride(0) == 0 and v.stride(0) == 0 and mask is None:
            return super()._attn(q, k, v, mask, return_probs)     # (folds the rows, then comes back here)
        s = torch.matmul(q.flip(-1) * 0.125, k.flip(-1).transpose(-1, -2))          # head dimension reversed
        if mask is not None:
            s = s + mask
        p = torch.softmax(s.flip(-1), dim=-1).flip(-1)                              # key axis reversed
        o = torch.matmul(self._r(p).flip(-1), v.flip(-2))
        B, H, Tq, _ = o.shape
        o = self._r(o.transpose(1, 2).reshape(B, Tq, H * 64))
        return (o, p) if return_probs else o

    def _dense(self, x, key, bias_key):
        if self.int8:
            return super()._dense(x, key, bias_key)             # exact integers: order-free
        y = torch.matmul(x.flip(-1), self.w[key].flip(-1).t())
        return y + self.w[bias_key] if bias_key else y


def run(int8, n, steps, out):
    cfg = get_config("large-v3")
    w = synthetic_weights(cfg, seed=1234)
    a = OracleWhisper(cfg, w, emulate_fp16=True, int8=int8)
    b = ReversedOrder(cfg, w, emulate_fp16=True, int8=int8)
    b.w, b.q = a.w, a.q                                          # one copy of the weights
    name = "int8_float16" if int8 else "float16"
    t0 = time.time()
    feats = olm.log_mel_chunks([bench_audio(480000, seed=100 + 13 * i) for i in range(n)], cfg.n_mels)
    ea, eb = a.encode(feats), b.encode(feats)
    rel = float(np.abs(ea - eb).max() / np.abs(ea).max())
    print(f"# {name}: {n} chunks encoded twice in {time.time() - t0:.0f} s; encoder outputs of the two orders differ by "
          f"{rel:.2e} (max, relative)", file=out, flush=True)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]
    kw = dict(beam_size=1, max_length=len(prompt) + steps, length_penalty=1.0, suppress_tokens=sup)
    ra = a.generate(ea, [prompt] * n, **kw)                       # order A's own greedy ids ...
    ids = [r.sequences_ids[0] for r in ra]
    rb = b.generate(eb, [prompt] * n, force_tokens=ids, **kw)     # ... scored by order B end to end (its own encoder output)
    la, lb = [dict(r) for r in a.detect_language(ea)], [dict(r) for r in b.detect_language(eb)]
    for j in range(n):
        sa, sb = ra[j].scores[0], rb[j].scores[0]                 # cum / len (length_penalty 1): the beam-score quantity
        dl = max((abs(la[j][k] - lb[j][k]) for k in la[j]), default=0.0)
        gaps = np.array(rb[j].forced_gaps)
        print(f"{name} chunk {j}: {steps}-step score {sa:.5f} vs {sb:.5f}: rel {abs(sa - sb) / max(1.0, abs(sb)):.2e} "
              f"(per token {abs(sa - sb):.2e}); language prob {dl:.2e}; no_speech {abs(ra[j].no_speech_prob - rb[j].no_speech_prob):.1e}; "
              f"{int((gaps > 0).sum())} of {len(gaps)} greedy ids differ from order B's arg-max (largest gap {gaps.max():.3f})",
              file=out, flush=True)


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out = sys.stdout
    print("# two valid evaluation orders of ONE restatement (oracle/whisper.py), large-v3 synthetic weights, end to end from "
          "the PCM;\n# quantity = score of the same token ids under both orders, cum log-prob / length", file=out)
    run(False, 3, 48, out)
    run(True, 3, 48, out)


if __name__ == "__main__":
    main()
