import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from faster_whisper_amd import _lib
        return _lib.load().fw_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the driver records
    # silent fallbacks as "native code not loaded".
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


def make_model(name="micro", seed=7, max_batch=4, max_beam=5, compute_type="float16", cfg=None, weights=None):
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    cfg = cfg or get_config(name)
    weights = weights if weights is not None else synthetic_weights(cfg, seed=seed)
    model = Whisper(f"synthetic:{name}", device="cuda", files={"config": cfg, "weights": weights},
                    compute_type=compute_type, max_batch_size=max_batch, max_beam_size=max_beam)
    return cfg, weights, model


def bench_audio(n_samples=480000, seed=0):
    """SURVEY.md section 8d synthetic audio: 0.1*N(0,1) + three partials (220/440/880 Hz, amp 0.05)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / 16000.0
    x = 0.1 * rng.standard_normal(n_samples)
    for f in (220.0, 440.0, 880.0):
        x += 0.05 * np.sin(2 * np.pi * f * t)
    return x.astype(np.float32)


def frag_perm(w, ke=32):
    """MFMA-fragment-major storage of a [N][K] operand (N % 16 == 0): element (n, k) at
    ((n/16 * K/ke + k/ke) * 64 + 16*((k/(ke/4))%4) + n%16) * (ke/4) + k%(ke/4); ke = 32 (fp16) or 64 (int8)."""
    n_rows, k_cols = w.shape
    octet = ke // 4
    n, k = np.meshgrid(np.arange(n_rows), np.arange(k_cols), indexing="ij")
    off = (((n >> 4) * (k_cols // ke) + k // ke) * 64 + ((k // octet) & 3) * 16 + (n & 15)) * octet + (k % octet)
    out = np.empty(n_rows * k_cols, dtype=w.dtype)
    out[off.reshape(-1)] = w.reshape(-1)
    return out.reshape(n_rows, k_cols)


def frag_unperm(t, ke=32):
    """inverse of frag_perm"""
    n_rows, k_cols = t.shape
    octet = ke // 4
    n, k = np.meshgrid(np.arange(n_rows), np.arange(k_cols), indexing="ij")
    off = (((n >> 4) * (k_cols // ke) + k // ke) * 64 + ((k // octet) & 3) * 16 + (n & 15)) * octet + (k % octet)
    return t.reshape(-1)[off.reshape(-1)].reshape(n_rows, k_cols)


def forced_result(oracle, enc_np_b, prompt, ids, kw):
    """the oracle teacher-forced along a GIVEN token sequence under the same decoding rules (its greedy path): the
    GenResult (score = cum log-prob / len^length_penalty exactly as a finished hypothesis is scored, no-speech
    probability, margins).  A sequence shorter than the budget ended with <eot>."""
    from oracle.whisper import max_new_tokens
    budget = max_new_tokens(kw.get("max_length", 448), len(prompt))
    forced = list(ids) + ([oracle.cfg.eot] if len(ids) < budget else [])
    k2 = {k: v for k, v in kw.items() if k not in ("beam_size", "patience", "num_hypotheses")}
    r = oracle.generate(enc_np_b[None] if enc_np_b.ndim == 2 else enc_np_b, [list(prompt)], beam_size=1,
                        force_tokens=[forced], **k2)[0]
    assert r.sequences_ids[0] == list(ids), (r.sequences_ids[0], ids)
    return r


def forced_score(oracle, enc_np_b, prompt, ids, kw):
    """the oracle's score of a GIVEN token sequence (forced_result above)"""
    r = forced_result(oracle, enc_np_b, prompt, ids, kw)
    forced_score.rule_margins = [m for m in (r.rule_margins or []) if m == m]     # (of the last call; nan = rule not applicable)
    return r.scores[0]


def greedy_gaps(oracle, enc_np_b, prompt, ids, kw):
    """the oracle teacher-forced along a GIVEN greedy sequence: per step, log-prob of the oracle's best token minus the
    log-prob of the sequence's token (0 where they agree).  A greedy engine whose every choice is within the numerical
    noise of the oracle's arg-max has all gaps below that noise — a statement about EVERY step of the engine's output,
    which a comparison of id prefixes (it ends at the first numerically tied step) cannot make."""
    from oracle.whisper import max_new_tokens
    budget = max_new_tokens(kw.get("max_length", 448), len(prompt))
    forced = list(ids) + ([oracle.cfg.eot] if len(ids) < budget else [])
    k2 = {k: v for k, v in kw.items() if k not in ("beam_size", "patience", "num_hypotheses")}
    r = oracle.generate(enc_np_b[None] if enc_np_b.ndim == 2 else enc_np_b, [list(prompt)], beam_size=1,
                        force_tokens=[forced], **k2)[0]
    return r.forced_gaps


def check_hypothesis(oracle, enc_np_b, prompt, got, ref, kw, tol=1e-3, gap=2e-2, what="", search=True, boundary=0.0,
                     rule_tie=None):
    """Parity criterion for one chunk of a beam-search (or greedy) result — never skipped, never "most of the time":
      1. the engine's reported score equals the ORACLE's score of the engine's own token sequence within `tol`
         (relative to max(1, |score|): the north-star 1e-3 on log-probs), whatever the search path was;
      2. the engine's sequence IS the oracle's, or it is not worse than the oracle's best by more than `gap` under the
         oracle's own scoring (beam search is a heuristic: a different path through a numerically tied step may end
         somewhere else, even somewhere better; a WORSE hypothesis is a bug).
    A rule condition can be numerically tied too (timestamp mass vs best text token, SURVEY.md A.3 rule (e)): then
    the oracle forbids a token the engine was allowed, its score of the engine's ids is -inf and check 1 cannot be
    made; the engine's own score then has to satisfy check 2.
    search=False (greedy): after a tied step greedy decoding just follows another path, better or worse, so check 2
    does not apply — the caller checks the margin-safe prefix of the ids instead.
    boundary: the same idea for beam search.  The oracle reports, per step, the gap between candidates K and K + 1
    (the pruning boundary, GenResult.margins).  If the engine's scores carry noise of size `boundary` (int8
    activations: a few 1e-2) and some step's gap is smaller, the two searches may keep different beam SETS from there
    on and end arbitrarily far apart — that is the heuristic, not a defect — so check 2 is made only when every
    boundary gap exceeds `boundary` (check 1 is made always).
    rule_tie: the finite form of the tied rule.  Timestamp rule (e) forbids text when log(timestamp mass) exceeds the best
    text log-prob; the oracle reports that difference per step of the forced path (GenResult.rule_margins).  Where it is
    within the numerical noise (default: max(boundary, 2e-2)) engine and oracle may decide the rule differently, and the
    log-prob of the SAME token then differs by log(timestamp mass) — a renormalisation, not an arithmetic error — so
    check 1 is not made for that hypothesis (it is printed); everything else is.
    Returns True when the ids are identical."""
    ids = got.sequences_ids[0]
    s_forced = forced_score(oracle, enc_np_b, prompt, ids, kw)
    s_got, s_ref = got.scores[0], ref.scores[0]
    same = ids == ref.sequences_ids[0]
    print(f"{what}: ids equal={same} engine score {s_got:.5f}, oracle score of the engine's ids {s_forced:.5f}, "
          f"oracle's best {s_ref:.5f}")
    rule_tie = max(boundary, 2e-2) if rule_tie is None else rule_tie
    tied_rules = [m for m in getattr(forced_score, "rule_margins", []) if abs(m) < rule_tie]
    if tied_rules and np.isfinite(s_forced) and not abs(s_got - s_forced) < tol * max(1.0, abs(s_forced)):
        print(f"{what}: timestamp rule (e) tied at {len(tied_rules)} step(s) of the engine's path (|log ts-mass - best text "
              f"log-prob| = {min(abs(m) for m in tied_rules):.4f} < {rule_tie:g}): the two sides may renormalise differently, "
              "scores not compared")
    elif np.isfinite(s_forced):
        assert abs(s_got - s_forced) < tol * max(1.0, abs(s_forced)), (what, s_got, s_forced)
    else:
        assert not same, (what, "the oracle scores its own sequence -inf")
        s_forced = s_got
    if not same and search:
        tight = [g for g in (ref.margins or []) if g < boundary]
        if tight:
            print(f"{what}: {len(tight)} pruning-boundary gap(s) below {boundary:g} (smallest {min(tight):.4f}): "
                  "the beam sets may differ, hypotheses not compared")
        else:
            assert s_forced > s_ref - gap * max(1.0, abs(s_ref)), (what, ids, ref.sequences_ids[0], s_forced, s_ref)
    return same
