"""SURVEY.md section 8f-1 on the GPU: a CTranslate2 model DIRECTORY (model.bin + config.json, the file set of
utils.py:91-97) written by ct2_format's writer, loaded from disk through the same constructor path the reference uses
(`WhisperModel(path)` -> `Whisper(path)` -> load_model_dir -> load_ct2_model_dir -> fw_model_create), must be the same
model as the one built from the in-memory weights: encoder output, greedy and beam results, language probabilities
and alignments bit for bit.  ([CT2-ext] the layout is restated from the published converter; what this pins is that
nothing between the file and the kernels — name mapping, aliases, dtypes, config.json fields — changes a bit.)"""
import numpy as np
import pytest

from conftest import bench_audio

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("base,compute_type", [("tiny.en", "float16"), ("tiny", "int8_float16")])
def test_model_directory_equals_in_memory_model(tmp_path, base, compute_type):
    from faster_whisper_amd import Whisper, get_config, synthetic_weights
    from faster_whisper_amd.ct2_format import save_ct2_model_dir
    from faster_whisper_amd.transcribe import WhisperModel
    cfg = get_config(base)
    cfg.n_enc_layers = cfg.n_dec_layers = 2            # a Whisper vocabulary (the loader keys on it), a short stack
    cfg.alignment_heads = [(1, 0), (1, 3)]
    w = synthetic_weights(cfg, seed=31)
    w16 = {k: v.astype(np.float16) for k, v in w.items()}          # what a converted checkpoint stores
    d = str(tmp_path / "ct2dir")
    save_ct2_model_dir(d, cfg, w16)
    mem = Whisper(f"synthetic:{base}", device="cuda", files={"config": cfg, "weights": w16}, compute_type=compute_type,
                  max_batch_size=3, max_beam_size=5)
    wm = WhisperModel(d, device="cuda", compute_type=compute_type, max_batch_size=3, max_beam_size=5)
    disk = wm.model
    c2 = disk.config
    assert (c2.d_model, c2.n_heads, c2.n_mels, c2.n_enc_layers, c2.n_dec_layers, c2.n_vocab) == (
        cfg.d_model, cfg.n_heads, cfg.n_mels, 2, 2, cfg.n_vocab)
    assert list(c2.alignment_heads) == [(1, 0), (1, 3)] and disk.is_multilingual == mem.is_multilingual
    chunks = [bench_audio(480000, seed=1), bench_audio(200000, seed=2), bench_audio(480000, seed=3)[::-1].copy()]
    ea, eb = mem.encode_pcm(chunks), disk.encode_pcm(chunks)
    assert np.array_equal(ea.to_numpy(), eb.to_numpy())
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    for kw in (dict(beam_size=1, max_length=len(prompt) + 12), dict(beam_size=5, max_length=len(prompt) + 12)):
        ga = mem.generate(ea, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
        gb = disk.generate(eb, [prompt] * 3, return_scores=True, return_no_speech_prob=True, **kw)
        for x, y in zip(ga, gb):
            assert x.sequences_ids == y.sequences_ids and x.scores == y.scores and x.no_speech_prob == y.no_speech_prob
    if cfg.is_multilingual:
        assert mem.detect_language(ea) == disk.detect_language(eb)
    text = [[11, 12, 13, 14]] * 3
    for x, y in zip(mem.align(ea, cfg.sot_sequence, text, [3000, 1250, 3000]),
                    disk.align(eb, cfg.sot_sequence, text, [3000, 1250, 3000])):
        assert x.alignments == y.alignments and x.text_token_probs == y.text_token_probs
