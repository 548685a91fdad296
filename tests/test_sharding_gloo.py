"""N > 1 path on CPU: two processes, `gloo` backend.  Covers the two exchanges of the sharded
batched path — weight-blob broadcast and per-round result gather — and
the sharded segment generator: rank 0 must yield exactly what a serial run yields, in order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_generate(chunks):
    """deterministic stand-in for encode+generate: ids and scores are pure functions of the PCM"""
    outs = []
    for c in chunks:
        h = int(abs(float(c.sum())) * 1000) % 997
        n = 3 + h % 5
        outs.append(dict(tokens=[(h + 7 * i) % 300 + 10 for i in range(n)], avg_logprob=-0.001 * h,
                         no_speech_prob=(h % 10) / 10.0))
    return outs


def _make_pipeline():
    from faster_whisper_amd import get_config
    from faster_whisper_amd.transcribe import BatchedInferencePipeline, WhisperModel
    cfg = get_config("micro")
    m = WhisperModel.__new__(WhisperModel)
    m.max_length, m.time_precision, m.input_stride = 448, 0.02, 2
    m.frames_per_second, m.tokens_per_second = 100, 50
    m.hf_tokenizer = None

    class _B:
        config = cfg
        is_multilingual = True
    m.model = _B()
    pipe = BatchedInferencePipeline(m)
    pipe.generate_segment_batched = lambda feats, tok, opt, audio_chunks=None: (None, _fake_generate(audio_chunks))
    return pipe, m


def _run_pipeline(pipe, m, shard):
    from faster_whisper_amd.transcribe import TranscriptionOptions
    rng = np.random.default_rng(5)
    chunks = [rng.standard_normal(1600 + 37 * i).astype(np.float32) for i in range(11)]
    meta = [{"offset": 30.0 * i, "duration": len(c) / 16000.0, "segments": []} for i, c in enumerate(chunks)]
    tok = m.make_tokenizer(task="transcribe", language="en")
    opt = TranscriptionOptions(5, 5, 1, 1, 1, 0, -1.0, 0.6, 2.4, False, 0.5, [0.0], None, None, True, (), True, 0.0,
                               False, "", "", False, None, [], None, None)
    return list(pipe._batched_segments_generator(chunks, tok, meta, 3, opt, False, shard, True))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from faster_whisper_amd.sharding import broadcast_blob, gather_results
    # 1) weight-blob broadcast
    blob = (np.arange(100003) % 251).astype(np.uint8) if rank == 0 else None
    t = broadcast_blob(blob, rank, rank)
    assert t.numel() == 100003 and int(t[100002]) == 100002 % 251 and int(t.sum()) == int(((np.arange(100003) % 251)).sum())

    # 2) ragged per-round gather
    class R:
        def __init__(self, ids, s, n):
            self.sequences_ids, self.scores, self.no_speech_prob = [ids], [s], n
    mine = [R([rank * 10 + i, 5], -float(rank + i), 0.25 * rank) for i in range(2 - rank)]  # rank0: 2, rank1: 1
    got = gather_results(mine, 8, rank, world, rank, counts=[2, 1])
    if rank == 0:
        assert [g[0] for g in got] == [[0, 5], [1, 5], [10, 5]]
        assert got[2][1] == -1.0 and got[2][2] == 0.25
    else:
        assert got is None
    # 3) sharded segment generator == serial
    pipe, m = _make_pipeline()
    segs = _run_pipeline(pipe, m, shard=True)
    if rank == 0:
        np.save(os.path.join(tmpdir, "sharded.npy"),
                np.array([[s.id, s.seek, len(s.tokens), s.tokens[0], round(s.avg_logprob * 1e6)] for s in segs]))
    else:
        assert segs == []
    # 4) the whole batched pipeline with WORD TIMESTAMPS, sharded: chunk-local alignment on the rank that holds
    #    the encoder output, gather_object of the word lists, sequential pause heuristics on rank 0
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_golden import _plain, make_model
    from faster_whisper_amd import get_config
    from faster_whisper_amd.transcribe import BatchedInferencePipeline
    from oracle import host_scenarios as hs
    from oracle import micro_tokenizer
    sc = hs.SCENARIOS["bat_clips_words"]
    model = make_model(get_config("micro"), micro_tokenizer.build())
    segments, info = BatchedInferencePipeline(model).transcribe(hs.synth_audio(*sc["audio"]), shard=True,
                                                                **json.loads(json.dumps(sc["kwargs"])))
    segments = [_plain(x) for x in segments]
    if rank == 0:
        with open(os.path.join(tmpdir, "sharded_words.json"), "w") as f:
            json.dump(segments, f)
    else:
        assert segments == []
        # this rank decoded and aligned only its half of the chunks
        assert [c[0] for c in model.model.calls].count("align") >= 1
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sharded = np.load(tmp_path / "sharded.npy")
    sys.path.insert(0, ROOT)
    pipe, m = _make_pipeline()
    serial = _run_pipeline(pipe, m, shard=False)
    ref = np.array([[s.id, s.seek, len(s.tokens), s.tokens[0], round(s.avg_logprob * 1e6)] for s in serial])
    assert sharded.shape == ref.shape == (11, 5)
    assert np.array_equal(sharded, ref)
    # the sharded word-timestamp run reproduces the REFERENCE's serial result (tests/golden/host_scenarios.json)
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_golden import _close
    with open(os.path.join(ROOT, "tests", "golden", "host_scenarios.json")) as f:
        want = json.load(f)["bat_clips_words"]["segments"]
    with open(tmp_path / "sharded_words.json") as f:
        got = json.load(f)
    assert len(got) == len(want) and sum(len(s["words"]) for s in got) > 50
    for i, (g, w) in enumerate(zip(got, want)):
        _close(g, w, f"segments[{i}]")
