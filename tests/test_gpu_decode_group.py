"""Decode groups (include/fwamd.h: fw_model_set_decode_batch / fw_model_join_decoder): the worker replicas of a
device share one decode workspace and concurrent generate() calls are merged into one decode run.  A merged run
must return, for every caller, EXACTLY what that caller gets alone — every kernel of a step works per row or per
chunk, so the comparison is bit-exact (ids and scores)."""
import threading

import numpy as np
import pytest

from conftest import bench_audio

pytestmark = pytest.mark.gpu


def _model(name, workers, compute_type="float16", max_batch=3):
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    cfg = get_config(name)
    w = synthetic_weights(cfg, seed=21)
    m = Whisper(f"synthetic:{name}", device="cuda", files={"config": cfg, "weights": w}, compute_type=compute_type,
                max_batch_size=max_batch, max_beam_size=5, inter_threads=workers)
    return cfg, m


def _batches(n, per):
    return [[bench_audio(480000 if (i + j) % 3 else 250000, seed=50 + 10 * i + j) for j in range(per)] for i in range(n)]


@pytest.mark.parametrize("name,compute_type", [("micro", "float16"), ("tiny.en", "float16"), ("micro", "int8_float16")])
def test_merged_runs_equal_solo_runs(name, compute_type):
    W = 4
    cfg, model = _model(name, W, compute_type)
    st = model.decode_stats()
    assert st["decode_batch"] == W * 3
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    sup = [cfg.sot, cfg.no_speech, 1, 2]
    kw = dict(beam_size=5, patience=1.0, length_penalty=1.0, max_length=len(prompt) + 10, suppress_tokens=sup,
              return_scores=True, return_no_speech_prob=True)
    batches = _batches(W, 3)
    batches[2] = batches[2][:2]                       # a smaller batch rides along
    # solo: one call at a time from this thread
    solo = [model.generate(model.encode_pcm(b), [prompt] * len(b), **kw) for b in batches]
    runs0 = model.decode_stats()["runs"]
    # concurrent: W host threads (one worker replica each) encode, then meet at a barrier and call generate together.
    # One more thread keeps an encode in flight meanwhile, so the leader of the run waits for the others to arrive.
    out = [None] * W
    encs = [None] * W
    bar = threading.Barrier(W)
    errs = []

    def work(i):
        try:
            encs[i] = model.encode_pcm(batches[i])
            bar.wait()
            out[i] = model.generate(encs[i], [prompt] * len(batches[i]), **kw)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    stop = threading.Event()

    def keep_encoding():
        while not stop.is_set():
            model.encode_pcm(batches[0][:1])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(W)]
    bg = threading.Thread(target=keep_encoding)
    for t in ts:
        t.start()
    bg.start()
    for t in ts:
        t.join()
    stop.set()
    bg.join()
    assert not errs, errs
    st = model.decode_stats()
    print(f"[{name} {compute_type}] {W} concurrent calls -> {st['runs'] - runs0} decode runs, largest run "
          f"{st['max_run_chunks']} chunks")
    assert st["max_run_chunks"] > 3                   # at least two calls shared a run
    for i in range(W):
        for a, b in zip(out[i], solo[i]):
            assert a.sequences_ids == b.sequences_ids
            assert a.scores == b.scores
            assert a.no_speech_prob == b.no_speech_prob


def test_mixed_options_are_not_merged_and_sampling_runs_alone():
    cfg, model = _model("micro", 3)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    prompt_ts = list(cfg.sot_sequence)
    batches = _batches(3, 2)
    kws = [dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True),
           dict(beam_size=2, max_length=len(prompt) + 8, return_scores=True),
           dict(beam_size=1, num_hypotheses=3, sampling_topk=0, sampling_temperature=0.7, seed=5,
                max_length=len(prompt) + 8, return_scores=True)]
    prompts = [prompt, prompt_ts, prompt]
    solo = [model.generate(model.encode_pcm(b), [p] * len(b), **kw) for b, p, kw in zip(batches, prompts, kws)]
    out = [None] * 3
    bar = threading.Barrier(3)

    def work(i):
        e = model.encode_pcm(batches[i])
        bar.wait()
        out[i] = model.generate(e, [prompts[i]] * len(batches[i]), **kws[i])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(3):
        for a, b in zip(out[i], solo[i]):
            assert a.sequences_ids == b.sequences_ids and a.scores == b.scores


def test_detect_language_and_align_through_a_worker():
    """decode-side calls of a worker replica run on the group's workspace"""
    from faster_whisper_amd.backend import StorageView
    cfg, model = _model("micro", 2)
    cfg1, single = _model("micro", 1)
    chunks = _batches(1, 3)[0]
    res = []

    def work():
        e = model.encode_pcm(chunks)                   # second thread -> second replica
        res.append((model.detect_language(e), model.align(e, cfg.sot_sequence, [[11, 12, 13]] * 3, [3000] * 3)))

    t = threading.Thread(target=work)
    model.encode_pcm(chunks[:1])                       # pins replica 0 to this thread
    t.start()
    t.join()
    e1 = single.encode_pcm(chunks)
    lang1 = single.detect_language(e1)
    al1 = single.align(e1, cfg.sot_sequence, [[11, 12, 13]] * 3, [3000] * 3)
    lang, al = res[0]
    assert lang == lang1
    for a, b in zip(al, al1):
        assert a.alignments == b.alignments and a.text_token_probs == b.text_token_probs


def test_merge_wait_zero_never_waits_and_knob_is_validated():
    """fw_model_set_merge_wait(0, ...): the leader of a run takes what is queued at once — a call that arrives alone runs
    alone even while other workers encode — and the result is the same bits either way"""
    from faster_whisper_amd import _lib
    cfg, model = _model("micro", 3)
    lib = model._lib
    h = model._replicas[0].handle
    assert lib.fw_model_set_merge_wait(h, -2, 50) != 0 and lib.fw_model_set_merge_wait(h, 10, 0) != 0   # rejected
    _lib.check(lib.fw_model_set_merge_wait(model._replicas[1].handle, 0, 90))    # through a worker: reaches the group
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True)
    batch = _batches(1, 3)[0]
    ref = model.generate(model.encode_pcm(batch), [prompt] * 3, **kw)
    stop = threading.Event()

    def keep_encoding():
        while not stop.is_set():
            model.encode_pcm(batch[:1])

    bg = threading.Thread(target=keep_encoding)
    bg.start()
    try:
        runs0 = model.decode_stats()["runs"]
        got = model.generate(model.encode_pcm(batch), [prompt] * 3, **kw)
        assert model.decode_stats()["runs"] == runs0 + 1
    finally:
        stop.set()
        bg.join()
    for a, b in zip(got, ref):
        assert a.sequences_ids == b.sequences_ids and a.scores == b.scores


def test_free_order_does_not_matter():
    """fw_model_free of the primary (weights + decode workspace) before its workers is deferred until the last worker is
    gone: the workers keep working, nothing dangles"""
    cfg, model = _model("micro", 3)
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    batch = _batches(1, 2)[0]
    kw = dict(beam_size=2, max_length=len(prompt) + 6, return_scores=True)
    ref = model.generate(model.encode_pcm(batch), [prompt] * 2, **kw)
    reps = list(model._replicas)
    reps[0].close()                                   # the primary first
    model._replicas = reps[1:]                        # the workers still encode and decode (on the primary's workspace)
    model._tls = threading.local()
    got = model.generate(model.encode_pcm(batch), [prompt] * 2, **kw)
    for a, b in zip(got, ref):
        assert a.sequences_ids == b.sequences_ids and a.scores == b.scores
    for r in reps[1:]:
        r.close()                                     # the last one takes the primary with it
    model._replicas = []
    cfg2, fresh = _model("micro", 1)
    assert len(fresh.generate(fresh.encode_pcm(batch), [prompt] * 2, **kw)) == 2


def test_row_capacity_is_checked_at_creation():
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    cfg = get_config("micro")
    w = synthetic_weights(cfg, seed=21)
    with pytest.raises(ValueError):
        Whisper("synthetic:micro", device="cuda", files={"config": cfg, "weights": w}, max_batch_size=200, max_beam_size=16)


def test_decode_lanes_switch():
    """fw_model_set_decode_lanes: one run at a time or two — same kernels, same weights, same results"""
    cfg, model = _model("micro", 4)                   # 12 chunks >= 4 encoder batches: the group has two lanes
    lib = model._lib
    assert lib.fw_model_set_decode_lanes(model._replicas[0].handle, 3) != 0       # rejected
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True)
    batches = _batches(4, 3)
    ref = [model.generate(model.encode_pcm(b), [prompt] * 3, **kw) for b in batches]
    for lanes in (1, 2):
        model.set_decode_lanes(lanes)
        out = [None] * 4

        def work(i):
            out[i] = model.generate(model.encode_pcm(batches[i]), [prompt] * 3, **kw)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i in range(4):
            for a, b in zip(out[i], ref[i]):
                assert a.sequences_ids == b.sequences_ids and a.scores == b.scores, (lanes, i)


def test_four_decode_lanes_knob(monkeypatch):
    """FWAMD_DECODE_LANES=4 (measurement knob, profiles/ab_r06_lanes.py): the group builds four lanes, fw_model_set_decode_lanes
    takes 1 .. 4 (5 refused), eight concurrent callers get what each gets alone whatever the number of runs in flight"""
    monkeypatch.setenv("FWAMD_DECODE_LANES", "4")
    cfg, model = _model("micro", 8)                   # 24 chunks >= 4 encoder batches: the lanes are built
    lib = model._lib
    assert lib.fw_model_set_decode_lanes(model._replicas[0].handle, 5) != 0
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True)
    batches = _batches(8, 3)
    ref = [model.generate(model.encode_pcm(b), [prompt] * 3, **kw) for b in batches]
    for lanes in (4, 3, 1, 4):
        model.set_decode_lanes(lanes)
        out = [None] * 8

        def work(i):
            out[i] = model.generate(model.encode_pcm(batches[i]), [prompt] * 3, **kw)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i in range(8):
            for a, b in zip(out[i], ref[i]):
                assert a.sequences_ids == b.sequences_ids and a.scores == b.scores, (lanes, i)


def test_decode_batch_cannot_change_under_a_run():
    """fw_model_set_decode_batch rebuilds the decode workspaces (and the second lane): refused with FW_EINVAL while the
    group has a run in flight, accepted again once it is idle — and results after a resize are what they were"""
    cfg, model = _model("micro", 4)
    lib, h = model._lib, model._replicas[0].handle
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    batch = _batches(1, 3)[0]
    enc = model.encode_pcm(batch)
    long_kw = dict(beam_size=5, max_length=len(prompt) + 400, min_new_tokens=400, return_scores=True)
    short_kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True)
    ref = model.generate(enc, [prompt] * 3, **short_kw)
    runs0 = model.decode_stats()["runs"]
    t = threading.Thread(target=lambda: model.generate(enc, [prompt] * 3, **long_kw))
    t.start()
    import time
    t_end = time.time() + 30
    while model.decode_stats()["runs"] == runs0 and time.time() < t_end:
        time.sleep(0.0005)                            # the long run has started
    rc = lib.fw_model_set_decode_batch(h, 6)
    still_running = t.is_alive()
    t.join()
    if still_running:                                 # (400 steps of the micro model: tens of milliseconds)
        assert rc != 0 and "in flight" in lib.fw_last_error().decode()
    assert lib.fw_model_set_decode_batch(h, 6) == 0 and model.decode_stats()["decode_batch"] == 6
    out = model.generate(model.encode_pcm(batch), [prompt] * 3, **short_kw)
    for a, b in zip(out, ref):
        assert a.sequences_ids == b.sequences_ids and a.scores == b.scores
    assert lib.fw_model_set_decode_batch(h, 12) == 0 and model.decode_stats()["decode_batch"] == 12


def test_cross_kv_pool_blocks_are_found_again_and_recycled():
    """The cross-attention K / V^T of a decode group live in ONE pool of per-encoder-output blocks (decoder.hip:
    CrossPool), shared by the lanes.  detect_language, generate and align on the same encoder output find its block
    again; encoder outputs beyond the pool's block count recycle the least recently used block; results never depend on
    which block (or which lane) served a call."""
    cfg, model = _model("micro", 4)                       # 4 blocks of 3 chunk slots, two lanes
    st = model.decode_stats()
    assert st["decode_batch"] == 12 and st["run_capacity"] == 12
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=5, max_length=len(prompt) + 8, return_scores=True, return_no_speech_prob=True)
    batches = _batches(7, 3)                              # more encoder outputs than blocks
    encs = [model.encode_pcm(b) for b in batches]
    first = []
    for e in encs:                                        # every call takes (and recycles) a block
        first.append((model.detect_language(e), model.generate(e, [prompt] * 3, **kw),
                      model.align(e, cfg.sot_sequence, [[11, 12, 13]] * 3, [3000] * 3)))
    for e, (dl, gen, al) in zip(reversed(encs), reversed(first)):   # again, in another order: blocks long recycled
        dl2, gen2 = model.detect_language(e), model.generate(e, [prompt] * 3, **kw)
        al2 = model.align(e, cfg.sot_sequence, [[11, 12, 13]] * 3, [3000] * 3)
        assert dl2 == dl
        for a, b in zip(gen2, gen):
            assert a.sequences_ids == b.sequences_ids and a.scores == b.scores and a.no_speech_prob == b.no_speech_prob
        for a, b in zip(al2, al):
            assert a.alignments == b.alignments and a.text_token_probs == b.text_token_probs


def test_a_call_that_cannot_fit_a_run_is_refused_at_entry():
    """random sampling runs every hypothesis as a row of its own: batch x num_hypotheses x positions beyond what ONE decode
    run holds is a ValueError for that caller alone, raised before the call is queued (it must not take down the calls
    it would have been merged with), and the model keeps working"""
    cfg, model = _model("micro", 2)                       # 6 chunks x beam 5 = 30 rows per run
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    e = model.encode_pcm(_batches(1, 3)[0])
    with pytest.raises(ValueError, match="decoder rows"):
        model.generate(e, [prompt] * 3, beam_size=1, num_hypotheses=11, sampling_topk=0, sampling_temperature=0.9,
                       max_length=len(prompt) + 8)
    ok = model.generate(e, [prompt] * 3, beam_size=1, num_hypotheses=10, sampling_topk=0, sampling_temperature=0.9,
                        max_length=len(prompt) + 8, seed=3, return_scores=True)
    assert len(ok) == 3 and all(len(r.sequences_ids) == 10 for r in ok)
    ref = model.generate(e, [prompt] * 3, beam_size=5, max_length=len(prompt) + 8, return_scores=True)
    assert all(len(r.sequences_ids[0]) >= 1 for r in ref)


def test_set_decode_batch_while_callers_arrive():
    """fw_model_set_decode_batch rebuilds the pool, the run workspaces and the second lane: refused while runs are queued or
    in flight, and a generate() that arrives DURING a rebuild waits for it instead of reading half-built state (ADVICE
    round 3: the rebuild used to drop the group lock before it touched lane1)"""
    from faster_whisper_amd import _lib
    cfg, model = _model("micro", 4)
    lib = _lib.load()
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    kw = dict(beam_size=5, max_length=len(prompt) + 6, return_scores=True)
    batch = _batches(1, 3)[0]
    ref = model.generate(model.encode_pcm(batch), [prompt] * 3, **kw)
    primary = model._replicas[0].handle
    stop = threading.Event()
    errs, n_ok = [], [0]

    def caller():
        try:
            while not stop.is_set():
                r = model.generate(model.encode_pcm(batch), [prompt] * 3, **kw)
                assert [x.sequences_ids for x in r] == [x.sequences_ids for x in ref]
                n_ok[0] += 1
        except Exception as ex:   # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=caller) for _ in range(3)]
    for t in ts:
        t.start()
    n_done = n_refused = 0
    for i in range(40):                                   # resize back and forth between one lane and two
        rc = lib.fw_model_set_decode_batch(primary, 12 if i % 2 else 6)
        n_done += rc == 0
        n_refused += rc != 0
    stop.set()
    for t in ts:
        t.join()
    assert not errs, errs
    print(f"resizes applied {n_done}, refused while busy {n_refused}, generate calls served {n_ok[0]}")
    assert n_ok[0] > 0 and n_done + n_refused == 40
