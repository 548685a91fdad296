#!/usr/bin/env python
"""Benchmark of the hot path named by BASELINE.json: large-v3, fp16, beam 5, batch 16.

A "step" = one pass of the hot path over one batch of 16 synthetic 30 s chunks whose PCM is
already resident in HBM: log-mel -> encoder -> cross-K/V projection -> beam-5 decode of
`--new-tokens` tokens per chunk (fixed length: synthetic weights never emit <|endoftext|>
on their own) -> ids/scores back on the host (+ gather to rank 0).
metric = audio seconds per wall second (whole job, all ranks).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (no rank environment: launches its own N ranks, see relaunch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--workers` host threads (CTranslate2's inter_threads / faster-whisper's num_workers) each submit batches of 16;
the workers of a GPU form a decode group: their encoders run side by side, their generate() calls are merged into
shared decode runs (include/fwamd.h).  N > 1: one process per GPU; rank 0 packs the weight blob and broadcasts it
over RCCL/xGMI, every rank processes its own batches (weak scaling, no data-path collective), per-step results
are gathered to rank 0.

Besides the metric the line carries (rank 0): `roofline` (dominant kernel family, HIP events on the engine's
streams during a profiled round of the same workload), `cpu_baseline` (oracle port on the host cores, N = 1),
and secondary measurements of the other SURVEY.md section 8 configurations: `pipeline` (section 8d wall-time
definition: ndarray in host memory -> last Segment of a 1 h recording), `cap_case` (224 new tokens per chunk),
`single_utterance` (C2: one 30 s chunk), and at N > 1 `sharded_recording` (C4: the 1 h recording sharded over
the ranks — strong scaling).
"""
import argparse
import datetime
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # HBM3E spec, MI355X_MICROARCH.md
MFMA_FAMILIES = ("enc_gemm", "enc_attn", "cross_kv_gemm")


def synth_chunks(n, seed):
    """SURVEY.md 8d audio: 0.1*N(0,1) + 220/440/880 Hz partials (amp 0.05), 30 s @ 16 kHz per chunk."""
    rng = np.random.default_rng(seed)
    t = np.arange(480000) / 16000.0
    tone = sum(0.05 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 880.0))
    return [(0.1 * rng.standard_normal(480000) + tone).astype(np.float32) for _ in range(n)]


def host_model(backend, cfg):
    """the host-side WhisperModel shell around an already loaded backend (no tokenizer file exists offline)"""
    import logging
    from faster_whisper_amd.transcribe import FeatureExtractor, WhisperModel
    wm = WhisperModel.__new__(WhisperModel)
    wm.logger = logging.getLogger("bench")
    wm.model = backend
    wm.hf_tokenizer = None
    wm.feature_extractor = FeatureExtractor(feature_size=cfg.n_mels, backend=backend)
    wm.input_stride, wm.time_precision, wm.max_length = 2, 0.02, 448
    wm.num_samples_per_token = wm.feature_extractor.hop_length * wm.input_stride
    wm.frames_per_second, wm.tokens_per_second = 100, 50
    return wm


def synthetic_vad(device_index=0, device=None):
    """A Silero-VAD-v6-shaped network (the initializer names / shapes of the reference's ONNX asset, faster_whisper/vad.py:
    288-351) with random weights, on the DEVICE path (csrc/vad.hip).  A random network answers ~ a constant; its output
    layer (a 1 x 1 convolution before the sigmoid) is rescaled so that digital silence and the bench's noise land on
    opposite sides of the threshold: logit' = gain * (logit - mid), mid / gain read off a calibration clip."""
    from faster_whisper_amd import vad as fvad
    device = device or os.environ.get("FWAMD_BENCH_VAD_DEVICE", "cuda")      # ("cpu": the CPU test of this path)
    rng = np.random.default_rng(3)
    f = lambda *sh, scale=0.08: (rng.standard_normal(sh) * scale).astype(np.float32)   # noqa: E731
    w = {"encoder.feature_extractor.forward_basis_buffer": f(258, 1, 256, scale=0.05),
         "decoder.conv1d.weight": f(1, 128, 1, scale=0.3), "decoder.conv1d.bias": f(1, scale=0.1),
         "onnx::LSTM_w": f(1, 512, 128), "onnx::LSTM_r": f(1, 512, 128), "onnx::LSTM_b": f(1, 1024, scale=0.2)}
    for i, (co, ci) in enumerate([(128, 129), (64, 128), (64, 64), (128, 64)]):
        w[f"encoder.conv_layers.{i}.weight"] = f(co, ci, 3)
        w[f"encoder.conv_layers.{i}.bias"] = f(co, scale=0.05)
    clip = np.concatenate([np.zeros(512 * 96, np.float32), synth_chunks(1, seed=5)[0][:512 * 96]])
    p = np.asarray(fvad.SileroVADModel(weights=w, device=device, device_index=device_index)(clip), dtype=np.float64)
    z = np.log(p / (1.0 - p))
    z_sil, z_noise = float(np.median(z[8:88])), float(np.percentile(z[104:184], 2))
    mid, gain = 0.5 * (z_sil + z_noise), 8.0 / max(1e-4, abs(z_noise - z_sil))
    if z_noise < z_sil:
        gain = -gain
    w["decoder.conv1d.weight"] = (w["decoder.conv1d.weight"] * gain).astype(np.float32)
    w["decoder.conv1d.bias"] = (w["decoder.conv1d.bias"] * gain - gain * mid).astype(np.float32)
    return fvad.SileroVADModel(weights=w, device=device, device_index=device_index)


def pipeline_rtf(backend, cfg, n_chunks, batch, beam, new_tokens, seed=0, shard=False, sync=None,
                 word_timestamps=False, vad=False):
    """End-to-end number (SURVEY.md section 8d wall-time definition): BatchedInferencePipeline.transcribe on one
    synthetic recording of n_chunks x 30 s that starts as an ndarray in HOST memory, timed until the last Segment is
    yielded — includes the host->device copy of the PCM, prompt / suppress-set construction, timestamp splitting and
    text rendering.  Decode length is fixed by suppressing <|endoftext|> up to max_new_tokens.
    shard=True: every rank of the torch.distributed job calls this; the chunk list is block-partitioned over the
    ranks and rank 0 yields all segments (strong scaling); `sync` = barrier + device sync used to bracket the timing.
    -> dict for the bench line (never raises: a failure is reported as {"error": ...})."""
    try:
        from faster_whisper_amd.transcribe import BatchedInferencePipeline
        wm = host_model(backend, cfg)
        chunks = synth_chunks(min(n_chunks, 8), seed=2000 + seed)
        audio = np.concatenate([chunks[i % len(chunks)] for i in range(n_chunks)])
        clips = [{"start": 30.0 * i, "end": 30.0 * (i + 1)} for i in range(n_chunks)]
        kw = dict(language="en", beam_size=beam, batch_size=batch, clip_timestamps=clips, max_new_tokens=new_tokens,
                  suppress_tokens=[cfg.eot], without_timestamps=True, word_timestamps=word_timestamps)
        if vad:
            # config C5: no clips — every 30 s slot ends in 2.5 s of digital silence, the native VAD (device kernels,
            # csrc/vad.hip) finds the 27.5 s bursts inside the timed wall, collect_chunks makes one chunk of each
            # (two padded spans exceed 30 s) and restore_speech_timestamps maps the times back
            from faster_whisper_amd import vad as fvad
            fvad._VAD_MODEL = synthetic_vad(getattr(backend, "device_index", [0])[0])
            audio = audio.copy()
            for i in range(n_chunks):
                audio[480000 * i + 440000:480000 * (i + 1)] = 0.0
            kw.pop("clip_timestamps")
            kw["vad_filter"] = True
        if shard:
            kw["shard"] = True
        pipe = BatchedInferencePipeline(wm)
        n_warm = min(n_chunks, batch * max(1, int(getattr(backend, "inter_threads", 1))))
        list(pipe.transcribe(audio[:480000 * n_warm], **(kw if vad else dict(kw, clip_timestamps=clips[:n_warm])))[0])
        if sync:
            sync()
        t0 = time.perf_counter()
        segments, _info = pipe.transcribe(audio, **kw)
        n_seg = n_tok = n_words = 0
        digest = 0
        for s in segments:
            n_seg += 1
            n_tok += len(s.tokens)
            n_words += len(s.words or ())
            # order-sensitive checksum of what was yielded (ids, avg_logprob, no_speech_prob): a sharded run and a serial
            # run of the same recording must agree on it
            digest = zlib.crc32(np.asarray(s.tokens, dtype=np.int32).tobytes()
                                + np.asarray([s.avg_logprob, s.no_speech_prob], dtype=np.float64).tobytes(), digest)
        if sync:
            sync()
        dt = time.perf_counter() - t0
        out = {"value": round(30.0 * n_chunks / dt, 2), "unit": "audio-seconds per wall-second",
               "audio_s": 30.0 * n_chunks, "wall_s": round(dt, 3), "segments": n_seg, "tokens": n_tok, "digest": digest,
               "what": "BatchedInferencePipeline.transcribe, ndarray in host memory -> last Segment"}
        if word_timestamps:
            out["words"] = n_words
        if vad:
            out["vad"] = "native device VAD inside the wall (fw_vad_forward_dev), vad_filter=True"
            out["what"] = "BatchedInferencePipeline.transcribe(vad_filter=True), ndarray in host memory -> last Segment"
            out["duration_after_vad_s"] = round(float(_info.duration_after_vad), 1)
        return out
    except Exception as e:   # a secondary number must never take the bench line down
        return {"error": f"{type(e).__name__}: {e}"}


def _multi(world):
    """the N > 1 control flow; FWAMD_DIST_AT_WORLD_1=1 takes it with ONE rank (RCCL communicator, blob broadcast, result
    gather, MAX over ranks, sharded recording) — the way a 1-GPU box executes the nccl branch
    (tests/test_gpu_rccl_world1.py)"""
    return world > 1 or os.environ.get("FWAMD_DIST_AT_WORLD_1") == "1"


TIMES = {}     # start-up seconds of this rank that build_backend measures itself (blob_broadcast_s)


def build_backend(args, cfg, rank, world, local_rank):
    """-> (backend, weights or None).  N > 1: rank 0 packs, RCCL broadcast, every rank builds from its HBM copy."""
    from faster_whisper_amd import Whisper, pack_blob, synthetic_weights
    ct = 1 if args.compute_type == "int8_float16" else 0
    common = dict(device="cuda", device_index=local_rank, max_batch_size=args.batch, max_beam_size=args.beam,
                  inter_threads=args.workers, compute_type=args.compute_type)
    if getattr(args, "merge_fill", None) is not None:
        common["merge_fill_percent"] = args.merge_fill
    if getattr(args, "merge_wait_ms", None) is not None:
        common["merge_wait_ms"] = args.merge_wait_ms
    weights = None
    if _multi(world):
        # rank 0 builds its model from the weights (the packed blob is then in ITS HBM); the other ranks receive that
        # allocation over RCCL / xGMI straight from rank 0's device memory — no 3 GB host image, no second upload
        from faster_whisper_amd.sharding import broadcast_blob_dev
        model = None
        if rank == 0:
            weights = synthetic_weights(cfg, seed=1234)
            model = Whisper(f"synthetic:{args.model}", files={"config": cfg, "weights": weights}, **common)
        t_b = time.time()
        dev_blob, nbytes = broadcast_blob_dev(model.blob() if rank == 0 else None, rank, local_rank)
        import torch
        torch.cuda.synchronize(local_rank)
        TIMES["blob_broadcast_s"] = round(time.time() - t_b, 3)
        TIMES["blob_bytes"] = nbytes
        if rank != 0:
            model = Whisper(f"synthetic:{args.model}", blob_dev=(dev_blob.data_ptr(), nbytes), **common)
            model._blob_keepalive = dev_blob
    elif os.environ.get("FWAMD_BLOB_CACHE"):
        # profiling convenience: repeated invocations (rocprofv3 passes) reuse one packed weight blob
        import torch
        # (the plain LayerNorm / weight forms travel in the blob only when it is PACKED with FWAMD_LN_UNFOLD / FWAMD_PACK_PLAIN
        #  set, engine.hip: pack_blob — a cache file made without them must not be handed to a process that asks for them)
        plain = ".plain" if (os.environ.get("FWAMD_LN_UNFOLD", "0") not in ("", "0") or os.environ.get("FWAMD_PACK_PLAIN")) else ""
        cache = os.environ["FWAMD_BLOB_CACHE"] + f".{args.model}.{args.compute_type}{plain}.npy"
        if os.path.exists(cache):
            blob = np.load(cache, mmap_mode="r")
        else:
            weights = synthetic_weights(cfg, seed=1234)
            blob = pack_blob(cfg, weights, ct)
            np.save(cache, blob)
        dev_blob = torch.from_numpy(np.array(blob, copy=True)).cuda()
        model = Whisper(f"synthetic:{args.model}", blob_dev=(dev_blob.data_ptr(), dev_blob.numel()), **common)
        model._blob_keepalive = dev_blob
    else:
        weights = synthetic_weights(cfg, seed=1234)
        model = Whisper(f"synthetic:{args.model}", files={"config": cfg, "weights": weights}, **common)
    return model, weights


def relaunch(argv, n):
    """`python bench.py --gpus N` without a rank environment: start N ranks of this script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1, a free port) and hand their exit status back.  Rank 0 of the child
    job prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def load_seam():
    """FWAMD_BENCH_SEAM=<file.py>: a test seam for fresh `python bench.py` processes (tests/test_bench_dist_gloo.py) —
    the file defines `factory(args, cfg, rank, world, local_rank) -> (backend, weights)` and `DIST_BACKEND`.  The
    benchmark itself never sets it."""
    path = os.environ.get("FWAMD_BENCH_SEAM")
    if not path:
        return None, None
    import runpy
    ns = runpy.run_path(path)
    return ns["factory"], ns.get("DIST_BACKEND")


def oracle_rev():
    """content hash of the CPU oracle the cpu_baseline leg times (the repository's .git does not travel to the GPU
    box): two bench lines are comparable on `cpu_baseline` only when this agrees"""
    import hashlib
    h = hashlib.sha256()
    for name in ("whisper.py", "logmel.py"):
        with open(os.path.join(ROOT, "oracle", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--new-tokens", type=int, default=100)
    ap.add_argument("--workers", type=int, default=32,
                    help="host threads submitting batches per GPU (worker replicas: own encoder stream, shared "
                         "weights, shared decode group).  The merged decode run grows with the batches in flight "
                         "and every decoder weight is streamed once per run: measured 6: 2 221x, 8: 2 405x, 12: 2 523x, "
                         "16: 2 595x, 20: 2 740x, 24: 2 765x, 32: 2 834x (decode workspace clamped to 336 chunks by HBM)")
    ap.add_argument("--decode-lanes", type=int, default=2, choices=[1, 2, 3, 4],
                    help="decode runs in flight per GPU (2: the product's default; 1: one run at a time, used for the kernel "
                         "traces in profiles/ so that kernel durations are not stretched by the other lane's kernels)")
    ap.add_argument("--merge-fill", type=int, default=None,
                    help="share (percent) of a decode run's chunk capacity the leader of a run waits for (backend default 90)")
    ap.add_argument("--merge-wait-ms", type=int, default=None,
                    help="how long the leader of a decode run waits after the last arrival for workers that are still encoding "
                         "(backend default: 2.5 measured encoder passes, at most 250 ms)")
    ap.add_argument("--compute-type", default="float16", choices=["float16", "int8_float16"],
                    help="float16 is the metric's configuration; int8_float16 times SURVEY section 8 config C3")
    ap.add_argument("--word-timestamps", action="store_true",
                    help="the pipeline measurement runs with word_timestamps=True (config C5: align pass timed)")
    ap.add_argument("--vad", action="store_true",
                    help="the pipeline measurement runs with vad_filter=True on the native device VAD (config C5: the recording "
                         "has digital silences, the Silero network — synthetic weights, calibrated to separate them — runs "
                         "INSIDE the timed wall, its spans are collected into chunks and the times restored)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (pipeline, cap_case, single_utterance, sharded_recording)")
    ap.add_argument("--pipeline-chunks", type=int, default=960,
                    help="30 s chunks of the end-to-end recording (960 = 8 h: >= 10 s of wall, a steady state)")
    ap.add_argument("--dry-dist", action="store_true",
                    help="N > 1 only: initialise the process group, broadcast the weight blob, build the model, decode ONE "
                         "batch per rank, gather once, and print the seconds every phase took on every rank — so that a "
                         "failing multi-GPU lease says WHERE (rendezvous, RCCL, blob, HBM pools, first launch)")
    ap.add_argument("--sharded-chunks", type=int, default=120,
                    help="30 s chunks of the recording sharded over the ranks at N > 1 (BASELINE config C4: 1 h)")
    return ap.parse_args(argv)


def main(argv=None, backend_factory=None, dist_backend=None):
    """backend_factory / dist_backend: test seams (tests/test_bench_dist_gloo.py runs the N > 1 control flow on CPU
    with a scripted backend over gloo); the benchmark itself never passes them."""
    args = parse_args(argv)
    if backend_factory is None and dist_backend is None:
        backend_factory, dist_backend = load_seam()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be the launcher (one rank per GPU), then leave with the job's status
        rc = relaunch(sys.argv[1:] if argv is None else list(argv), args.gpus)
        if rc:
            raise SystemExit(rc)
        return None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): "
                         "refusing to report a line for a different job size")
    dist = None
    on_gpu = dist_backend in (None, "nccl")
    multi = _multi(world)
    phases = {}          # seconds per start-up phase of THIS rank (--dry-dist prints them; stderr gets them as they pass)
    t_ph = time.time()

    def phase(name):
        nonlocal t_ph
        now = time.time()
        phases[name] = round(now - t_ph, 3)
        t_ph = now
        if multi:
            print(f"[bench rank {rank}/{world}] {name}: {phases[name]:.2f} s", file=sys.stderr, flush=True)

    if multi:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if on_gpu:
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        phase("import_torch")
        # a rank that fails before a collective must not leave the others waiting for ever
        dist.init_process_group(dist_backend or "nccl", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=300), **kw)
        phase("init_process_group")
        dist.barrier()                 # the first collective creates the communicator (RCCL: rings over xGMI)
        phase("first_barrier")

    from faster_whisper_amd import get_config
    from faster_whisper_amd.sharding import gather_results

    cfg = get_config(args.model)
    t0 = time.time()
    model, weights = (backend_factory or build_backend)(args, cfg, rank, world, local_rank)
    load_s = time.time() - t0
    phase("blob_broadcast_and_model")
    if "blob_broadcast_s" in TIMES:          # (N > 1: the collective alone, measured inside build_backend)
        phases["blob_broadcast"] = TIMES["blob_broadcast_s"]
        if multi:
            print(f"[bench rank {rank}/{world}] blob_broadcast (inside the phase above, {TIMES['blob_bytes'] / 1e9:.2f} GB "
                  f"device -> device): {TIMES['blob_broadcast_s']:.2f} s", file=sys.stderr, flush=True)

    lanes = getattr(args, "decode_lanes", 2)
    if lanes != 2 and hasattr(model, "set_decode_lanes"):
        model.set_decode_lanes(lanes)
    chunks = synth_chunks(args.batch, seed=1000 + rank)
    staged = model.stage_pcm(chunks)
    # every worker slot of the pool its OWN 16 chunks of PCM (resident in HBM like `staged`): the batches in flight are
    # different audio, as the batches of a recording are (step i takes set i mod W; set 0 = `staged`)
    staged_sets = [staged]
    if hasattr(model, "stage_pcm") and not getattr(args, "dry_dist", False):
        for w in range(1, max(1, args.workers)):
            staged_sets.append(model.stage_pcm(synth_chunks(args.batch, seed=1000 + rank + 7919 * w)))
    prompt = list(cfg.sot_sequence) + [cfg.no_timestamps]
    L = args.new_tokens
    sup = [cfg.sot, cfg.sot_prev, cfg.sot_lm, cfg.no_speech, cfg.translate, cfg.transcribe]

    def gen_kw(n):
        return dict(beam_size=args.beam, patience=1.0, length_penalty=1.0, max_length=len(prompt) + n,
                    return_scores=True, return_no_speech_prob=True, suppress_blank=True, suppress_tokens=sup,
                    min_new_tokens=n)

    def step(n=L, which=0):
        enc = model.encode_pcm_staged(staged_sets[which % len(staged_sets)])
        return model.generate(enc, [prompt] * args.batch, **gen_kw(n))

    def barrier():
        if multi:
            dist.barrier()
        model.synchronize()

    # W host threads, one worker replica (encoder stream + workspaces, shared weights) each: the batches of a long
    # recording are independent, so several are kept in flight on the GPU at once.
    from concurrent.futures import ThreadPoolExecutor
    import threading
    W = max(1, args.workers)
    pool = ThreadPoolExecutor(max_workers=W)
    sync = threading.Barrier(W)

    def warm(_):
        sync.wait()                      # every pool thread takes exactly one of these tasks
        for _ in range(max(1, args.warmup)):
            r = step()
        return r

    def run_steps(n, n_tok=L, gather=True):
        futs = [pool.submit(step, n_tok, i) for i in range(n)]
        outs = [f.result() for f in futs]          # results come back in submission order = chunk order
        if multi and gather:
            # ONE gather per timed region (the path has no collective inside it: DESIGN.md section 7): the fixed-size
            # records of every step of this rank travel to rank 0 together, in rank order
            gather_results([r for o in outs for r in o], n_tok, rank, world, local_rank)
        return outs

    last = {}

    def timed(n, n_tok=L):
        barrier()
        t1 = time.perf_counter()
        res = run_steps(n, n_tok)[-1]
        barrier()
        dt = time.perf_counter() - t1
        last["res"] = res
        last["which"] = n - 1            # the PCM set of the last step (run_steps: step i takes set i mod W)
        if multi:
            import torch
            tt = torch.tensor([dt], device=f"cuda:{local_rank}" if on_gpu else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert all(len(r.sequences_ids[0]) == n_tok for r in res), "decode length is not the requested fixed length"
        return dt

    if getattr(args, "dry_dist", False):
        if not multi:
            raise SystemExit("bench.py: --dry-dist is the N > 1 start-up check (use --gpus N)")
        phase("stage_pcm")
        res = step()
        model.synchronize()
        phase("first_batch")
        gather_results(res, L, rank, world, local_rank)
        phase("gather")
        allp = [None] * world
        dist.all_gather_object(allp, phases)
        if rank == 0:
            print(json.dumps({"dry_dist": True, "n_gpus": world, "phases_s_by_rank": allp,
                              "slowest_phase": max(((k, max(p.get(k, 0.0) for p in allp)) for k in phases),
                                                   key=lambda kv: kv[1])}), flush=True)
        model.free_staged(staged)
        pool.shutdown()
        dist.barrier()
        dist.destroy_process_group()
        return {"dry_dist": True}
    if args.warmup > 0:
        list(pool.map(warm, range(W)))
    phase("warmup")
    stats0 = model.decode_stats()
    elapsed = timed(args.steps)
    stats1 = model.decode_stats()
    # ---- the timed configuration checks itself: the last batch of the timed region (decoded inside a merged run of
    #      many batches) against the same batch decoded ALONE afterwards — ids, scores and no-speech probabilities
    #      must be identical bit for bit (every kernel works per row / per chunk; tests/test_gpu_full_size.py checks
    #      the same geometry against the oracle) ----
    merged_res = last["res"]
    solo_res = step(which=last["which"])
    verified = all(a.sequences_ids == b.sequences_ids and a.scores == b.scores and a.no_speech_prob == b.no_speech_prob
                   for a, b in zip(merged_res, solo_res)) and len(merged_res) == len(solo_res) == args.batch

    audio_s = 30.0 * args.batch * args.steps * world
    value = audio_s / elapsed
    runs = max(1, stats1["runs"] - stats0["runs"])
    out = {
        "metric": "audio-sec/sec (RTF) large-v3 fp16 beam=5 batch=16", "value": round(value, 2),
        "unit": "audio-seconds per wall-second", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / max(1, args.steps), 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.compute_type == "float16" else "i8/f16",
        "data": "synthetic",
        "verified": bool(verified),
        "verified_what": "last batch of the timed region (decoded in a merged run) == the same batch decoded alone "
                         "afterwards: ids, scores, no_speech_prob bit-identical",
        "value_definition": "step = one 16-chunk batch, PCM resident in HBM -> ids / scores on the host (the task's bench "
                            "contract); `pipeline` is the SURVEY section 8d wall (ndarray in host memory -> last Segment)",
        "config": {"workload": f"{args.model} {args.compute_type} BatchedInferencePipeline hot path: {args.batch} x 30 s "
                               f"chunks/step, beam_size={args.beam}, {L} new tokens/chunk (fixed), PCM resident in HBM",
                   "global_batch": args.batch * world, "new_tokens": L, "workers_per_gpu": W, "decode_lanes": lanes,
                   "decode_group": {"capacity_chunks": stats1["decode_batch"],
                                    "run_capacity_chunks": stats1.get("run_capacity", stats1["decode_batch"]),
                                    "decode_runs": runs,
                                    "chunks_per_run": round((stats1["chunks"] - stats0["chunks"]) / runs, 1),
                                    "largest_run_chunks": stats1["max_run_chunks"]},
                   "steady_state": args.steps >= 2 * W,
                   "steady_state_note": ("the timed region is at least two rounds of the worker pool" if args.steps >= 2 * W else
                                         f"{args.steps} steps < 2 x {W} workers: the timed region is ONE burst of the pool; "
                                         "`steady` holds the same measurement over 4 rounds"),
                   "model_load_s": round(load_s, 1),
                   **({"blob_broadcast_s": TIMES["blob_broadcast_s"],
                       "blob_broadcast_what": "RCCL broadcast of the packed weight blob from rank 0's device memory (no host "
                                              "image); part of model_load_s"} if "blob_broadcast_s" in TIMES else {})},
    }

    # the combined roofline ceiling of this configuration (encoder MFMA + decode HBM, L = 100) is
    # 4 525x per GPU; L = 224: 2 211x
    ceiling = {100: 4525.0, 224: 2211.0}
    if args.model == "large-v3" and args.compute_type == "float16" and L in ceiling:
        out["roofline_combined"] = {"value_over_ceiling": round(value / (ceiling[L] * world), 4),
                                    "ceiling_per_gpu": ceiling[L], "source": "SURVEY.md section 8d",
                                    "attainable_note": "SURVEY's ceiling charges the 1.47 GB of decoder weights to every "
                                                       "16-chunk batch and step; merged decode runs stream them once per RUN "
                                                       "(~16 batches).  With that, per batch: cross-attention 393 GB at the "
                                                       "6.3 TB/s a copy achieves = 64.3 ms, weights 1.5 ms, encoder + cross-K/V "
                                                       "+ decoder linears 53 TFLOP at the 2.5 PFLOP/s MFMA peak = 21.2 ms: "
                                                       "5 500x with the two roofs ADDED (SURVEY's method), 7 450x if they "
                                                       "overlapped perfectly — the fraction over 4 525x flatters the gap"}
        out["roofline_combined"]["value_over_merged_run_ceiling"] = round(value / (5500.0 * world), 4)
    secondary = not args.no_secondary
    # ---- secondary, all ranks take part: steady state, the cap case and (N > 1) the sharded recording ----
    if secondary and args.steps < 2 * W:
        ns = 4 * W
        dts = timed(ns)
        out["steady"] = {"value": round(30.0 * args.batch * ns * world / dts, 2), "unit": "audio-seconds per wall-second",
                         "steps": ns, "ms_per_step": round(1000.0 * dts / ns, 3)}
        if "roofline_combined" in out:
            out["roofline_combined"]["steady_over_ceiling"] = round(out["steady"]["value"] / (ceiling[L] * world), 4)
    if secondary:
        n2 = max(W, min(args.steps, 2 * W))
        run_steps(W, 224, gather=False)        # graph capture etc. for the other decode length
        dt2 = timed(n2, 224)
        out["cap_case"] = {"new_tokens": 224, "value": round(30.0 * args.batch * n2 * world / dt2, 2),
                           "unit": "audio-seconds per wall-second", "steps": n2}
        if multi:
            out["sharded_recording"] = dict(
                pipeline_rtf(model, cfg, args.sharded_chunks, args.batch, args.beam, L, shard=True, sync=barrier),
                scaling="strong", n_gpus=world)

    if rank == 0:
        # ---- roofline of the dominant kernel family: one profiled round of the same workload (HIP events on the
        #      engine's streams; the decode step runs eagerly instead of as a graph replay while profiling) ----
        if not args.no_profile_pass:
            # ONE decode run at a time while profiling: with two lanes a kernel's event-to-event time would include the
            # other lane's kernels running beside it; the roofline is a statement about the kernel
            if hasattr(model, "set_decode_lanes"):
                model.set_decode_lanes(1)
            model.profile(True, replica=None)
            ps0 = model.decode_stats()
            run_steps(W, gather=False)
            model.synchronize()
            ps1 = model.decode_stats()
            rep = model.profile_report(replica=None)
            model.profile(False, replica=None)
            if hasattr(model, "set_decode_lanes"):
                model.set_decode_lanes(lanes)
            for v in rep.values():             # per batch-step
                for f in ("ms", "bytes", "flops"):
                    v[f] /= W
            tot = sum(v["ms"] for v in rep.values())
            # Dominant KERNEL = the kernel symbol with the largest share of the step.  The decoder linears are three
            # instantiations of dec_gemm_frag_kernel (LayerNorm-folded / plain / long-K), timed here as four role
            # families (qkv, d x d, ffn1, ffn2): together they are listed under roofline_others as "dec_gemm".
            # (the decode runs of the PROFILED round: one lane, W batches — not those of the timed region)
            rows_per_run = args.beam * (ps1["chunks"] - ps0["chunks"]) / max(1, ps1["runs"] - ps0["runs"])
            # runs of at least DEC_BIG_MIN_ROWS rows take the GEMM-shaped kernel (dec_gemm_big_kernel) and are priced against
            # the MFMA roof; below it the linears are weight-streaming launches priced against HBM
            # (every linear has its own measured crossover: include/fwamd_test.h fw_dec_big_min_rows_of; a family is priced
            #  against the MFMA roof only when ITS kernel of the profiled round was the GEMM-shaped one)
            roles = {"dec_gemm_qkv": 0, "dec_gemm_dxd": 1, "dec_gemm_ffn1": 2, "dec_gemm_ffn2": 3}
            if hasattr(model, "dec_big_min_rows_of"):
                big_of = {k: model.dec_big_min_rows_of(r) for k, r in roles.items()}
            else:
                big_of = {k: (model.dec_big_min_rows() if hasattr(model, "dec_big_min_rows") else 1024) for k in roles}
            shaped = {k: rows_per_run >= v for k, v in big_of.items()}
            gemm_shaped = all(shaped.values())          # "dec_gemm" as a whole: every linear took the GEMM-shaped kernel
            big_rows = max(big_of.values())

            def roof_of(name, v):
                if name in MFMA_FAMILIES or (name == "dec_gemm" and gemm_shaped) or shaped.get(name, False):
                    ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
                    return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None}
                ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
            name, dom = max(rep.items(), key=lambda kv: kv[1]["ms"])
            roof = roof_of(name, dom)
            roof["kernel"] = name
            # `traffic` = HBM read bytes per launch from the PMC pass (a number, or null); the pass profiles one
            # 16-chunk batch per launch, so `traffic_pass` carries the algorithmic bytes of a launch IN THAT PASS
            tp = pmc_traffic(name)
            roof["traffic"] = tp["hbm_read_bytes_per_launch"] if tp else None
            roof["traffic_pass"] = tp
            if tp and tp.get("algorithmic_bytes_per_launch_in_that_pass"):
                roof["traffic_over_algorithmic"] = round(
                    tp["hbm_read_bytes_per_launch"] / tp["algorithmic_bytes_per_launch_in_that_pass"], 4)
            roof["kernel_ms_per_step"] = round(dom["ms"], 3)
            roof["launch_groups_in_round"] = dom["launches"]
            roof["timing"] = (f"HIP events around every launch on the engine's streams, one round of {W} batches with "
                              "all workers active, divided by the batches; the profiled round decodes ONE run at a time "
                              f"(the timed region: {lanes} decode lane(s) — with two, kernels of two runs share the chip "
                              "and a kernel's wall time is no longer its own)")
            parts = [v for k, v in rep.items() if k.startswith("dec_gemm_")]
            others = {}
            mixed = any(shaped.values()) and not gemm_shaped
            if parts and not mixed:
                others["dec_gemm"] = {f: sum(v[f] for v in parts) for f in ("ms", "bytes", "flops", "launches")}
            elif parts:                     # some linears on either kernel: one entry per role, each with its own bound
                for k in roles:
                    if k in rep:
                        others[k] = rep[k]
            for k in ("enc_gemm", "dec_cross_attn", "dec_self_attn", "enc_attn"):
                if k in rep and k != name:
                    others[k] = rep[k]
            out["roofline_others"] = {
                k: dict(roof_of(k, v), kernel_ms_per_step=round(v["ms"], 3),
                        traffic=(pmc_traffic(k, gemm_shaped or shaped.get(k, False)) or {}).get("hbm_read_bytes_per_launch"),
                        traffic_kernel=(pmc_traffic(k, gemm_shaped or shaped.get(k, False)) or {}).get("kernel"))
                for k, v in others.items()
                if v["ms"] > 0 and (v["flops"] > 0 if (k in MFMA_FAMILIES or (k == "dec_gemm" and gemm_shaped)
                                                     or shaped.get(k, False)) else v["bytes"] > 0)}
            if mixed:
                out["roofline_others_note"] = (
                    f"decode runs of {rows_per_run:.0f} rows on average: the linears switch to dec_gemm_big_kernel at "
                    f"{big_of} rows, so the four role families are listed one by one, each against the roof of the kernel "
                    "that served it")
            if "dec_gemm" in out["roofline_others"]:
                out["roofline_others"]["dec_gemm"]["note"] = (
                    f"decode runs of {rows_per_run:.0f} rows on average (>= {big_rows}: dec_gemm_big_kernel): the six per-layer "
                    "linears are GEMM-shaped and priced against the MFMA roof" if gemm_shaped else
                    f"decode runs of {rows_per_run:.0f} rows on average (< {big_rows}: the register-streaming kernel): "
                    "weight-streaming launches, priced against the HBM roof")
            out["roofline"] = roof
            out["families_ms_per_step"] = {k: round(v["ms"], 3) for k, v in rep.items()}
            out["families_sum_ms"] = round(tot, 3)
            fam = {}
            for k, v in rep.items():
                if v["ms"] <= 0:
                    continue
                if k in MFMA_FAMILIES:
                    fam[k] = {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                elif v["bytes"] > 0:
                    fam[k] = {"GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
            out["families_rate"] = fam
        if secondary and not multi:
            out["pipeline"] = pipeline_rtf(model, cfg, args.pipeline_chunks, args.batch, args.beam, L,
                                           word_timestamps=args.word_timestamps, vad=args.vad)
            out["single_utterance"] = single_utterance(model, cfg, chunks[0], prompt, gen_kw(L), L)
            out["one_batch_at_a_time"] = one_batch(model, staged, chunks, prompt, gen_kw(L), L, args.batch)
        # ---- CPU baseline: the oracle (torch fp32 port) on a bounded sample of the same workload ----
        # (reported at N = 1 only: at N > 1 the other ranks would idle at the final barrier while it runs)
        if not args.no_cpu_baseline and not multi:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, chunks, prompt, args.beam, L, gen_kw(L))
        print(json.dumps(out), flush=True)
    for st_ in staged_sets:
        model.free_staged(st_)
    pool.shutdown()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        # a merged run that does not reproduce the solo run bit for bit is a wrong result, not a slow one
        raise SystemExit("bench.py: the last batch of the timed region differs from the same batch decoded alone "
                         "(`verified`: false) — the line above must not be used")
    return out


def single_utterance(model, cfg, chunk, prompt, kw, L, reps=3):
    """config C2: ONE 30 s utterance, beam 5 (latency-bound: 5 decoder rows) — PCM from host memory, log-mel,
    encoder, decode, results on the host"""
    try:
        best = None
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            enc = model.encode_pcm([chunk])
            r = model.generate(enc, [prompt], **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)     # first iteration warms the graph for this shape
        assert len(r[0].sequences_ids[0]) == L
        return {"value": round(30.0 / best, 2), "unit": "audio-seconds per wall-second", "latency_ms": round(1e3 * best, 1),
                "what": f"one 30 s chunk, beam 5, {L} new tokens, host PCM -> ids on the host (best of {reps})"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def one_batch(model, staged, chunks, prompt, kw, L, batch, reps=4):
    """no batches in flight besides one (`--workers 1` semantics inside the same process): solo decode runs of 16 chunks,
    and the latency of ONE 15-chunk batch — what each rank of BASELINE config C4 (1 h = 120 chunks over 8 GPUs) gets —
    from which the per-rank wall and the real-time factor of that configuration at N = 8 follow."""
    try:
        model.generate(model.encode_pcm_staged(staged), [prompt] * batch, **kw)      # warm (graph of the solo shape)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = model.generate(model.encode_pcm_staged(staged), [prompt] * batch, **kw)
        dt = (time.perf_counter() - t0) / reps
        assert all(len(x.sequences_ids[0]) == L for x in r)
        c15 = chunks[:15] if len(chunks) >= 15 else chunks
        model.generate(model.encode_pcm(c15), [prompt] * len(c15), **kw)             # warm
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            model.generate(model.encode_pcm(c15), [prompt] * len(c15), **kw)        # PCM from host memory
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        one_call = best
        # the same 15 chunks as TWO sub-batches (8 + 7) on two host threads = what the batched driver does with a single
        # batch and >= 2 workers (transcribe.py: _batched_segments_generator): the second half's encoder pass runs under
        # the first half's decode run, the two runs decode side by side on the two lanes
        halves = None
        if getattr(model, "inter_threads", 1) >= 2 and len(c15) >= 4:
            import threading
            h = (len(c15) + 1) // 2
            parts = [c15[:h], c15[h:]]

            def half(i, out):
                out[i] = model.generate(model.encode_pcm(parts[i]), [prompt] * len(parts[i]), **kw)

            for rep in range(4):            # (first: warm — graphs of the two shapes)
                out = [None, None]
                ths = [threading.Thread(target=half, args=(i, out)) for i in range(2)]
                t0 = time.perf_counter()
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                d = time.perf_counter() - t0
                if rep:
                    halves = d if halves is None else min(halves, d)
            best = min(best, halves)
        return {"value": round(30.0 * batch / dt, 2), "unit": "audio-seconds per wall-second",
                "latency_ms_per_batch": round(1e3 * dt, 1),
                "c4_rank_batch": {"chunks": len(c15), "latency_ms": round(1e3 * best, 1),
                                  "one_call_ms": round(1e3 * one_call, 1),
                                  "two_halves_ms": None if halves is None else round(1e3 * halves, 1),
                                  "predicted_rtf_at_8_gpus_1h": round(3600.0 / best, 1),
                                  "what": "1 h = 120 chunks over 8 ranks = one 15-chunk batch per rank: wall = this latency "
                                          "(+ one gather); nothing merges at that size"},
                "what": f"one {batch}-chunk batch at a time (solo decode runs), beam 5, {L} new tokens"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


PMC_TAGS = ("r06", "r05", "r04")                 # the round's counter passes (profiles/collect.sh <tag>), newest first:
#   <tag>_pmc_fetch_w32.json   the TIMED configuration (32 workers, merged decode runs, one lane)
#   <tag>_pmc_fetch.json       one 16-chunk batch per decode run
_PMC_KERNEL = {"dec_cross_attn": "dec_cross_attn_kernel", "enc_gemm": "gemm_f16", "enc_attn": "attn_enc_kernel",
               "dec_gemm": "dec_gemm_frag_kernel", "dec_self_attn": "dec_self_attn", "dec_logits": "dec_gemm_wave_kernel"}


def _pmc_file(suffix):
    for tag in PMC_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{suffix}.json")
        if os.path.exists(path):
            return path
    return None


def pmc_traffic(family, gemm_shaped=False):
    """HBM read bytes per launch of the family's kernel from the newest committed rocprofv3 --pmc FETCH_SIZE pass of the
    TIMED configuration (merged decode runs; the one-batch-per-run pass when that kernel is not in it), x2 gfx950
    correction already applied by profiles/parse_pmc.py; null if the kernel was not measured.  gemm_shaped: the decoder
    linears of the profiled round took dec_gemm_big_kernel (runs >= fw_dec_big_min_rows rows) — its figure, not the
    register-streaming kernel's."""
    fam0 = "dec_gemm" if family.startswith("dec_gemm") else family
    if fam0 not in _PMC_KERNEL:
        return None
    kernel = "dec_gemm_big_kernel" if (fam0 == "dec_gemm" and gemm_shaped) else _PMC_KERNEL[fam0]
    for suffix, what in (("fetch_w32", "the counter pass of the timed configuration (profiles/collect.sh: the bench command "
                                       "with 32 workers, one decode lane, eager decode step: merged decode runs)"),
                         ("fetch", "the counter pass with --workers 1 (one 16-chunk batch per decode run, eager decode step)")):
        path = _pmc_file(suffix)
        if not path:
            continue
        with open(path) as f:
            j = json.load(f)
        tot = n = alg = chunks = 0.0
        for k, v in j.items():            # a templated kernel appears once per instantiation: dispatch-weighted mean
            b = v.get("hbm_read_bytes_per_launch_corrected")
            if kernel in k and b is not None:
                d = v.get("FETCH_SIZE", {}).get("dispatches", 1)
                tot += b * d
                n += d
                alg += v.get("algorithmic_bytes_per_launch", 0.0) * d
                chunks += v.get("chunks_per_launch_mean", 0.0) * d
        if n == 0:
            continue
        out = {"hbm_read_bytes_per_launch": round(tot / n), "kernel": kernel, "source": os.path.relpath(path, ROOT),
               "note": "mean over the launches of that kernel in " + what}
        if family == "dec_cross_attn":
            # one launch = one decoder layer for the chunks of the run: K and V^T of that layer, 1500 x 1280 fp16 each;
            # merged runs: the launch's grid gives the chunks it streamed (profiles/parse_pmc.py)
            out["algorithmic_bytes_per_launch_in_that_pass"] = round(alg / n) if alg else 16 * 2 * 1500 * 1280 * 2
            if chunks:
                out["chunks_per_launch_in_that_pass"] = round(chunks / n, 1)
        return out
    return None


def cpu_baseline(cfg, weights, chunks, prompt, beam, L, gen_kw):
    """oracle/ (CPU restatement, torch fp32) on a BOUNDED sample of the same workload — per 30 s chunk: log-mel, the whole
    encoder, cross-K/V + prompt (all measured) and 8 beam steps, extrapolated linearly to L steps.

    Two figures (round 6, verdict item 6: "timed on the host cores of the same box"):
      * `value`: ALL host cores — host_cores // 32 concurrent streams of 32 threads each (the oracle decodes one chunk at a
        time, and a 32-thread team is where its matmuls stop scaling: the way CTranslate2 would be run on this box is
        inter_threads x intra_threads), each on its own chunk of the batch, sharing one copy of the weights; the streams run
        the sample side by side (so they contend for memory bandwidth as a real run would) and the figure is the SUM of
        their rates.  `cores` = the threads actually used.
      * `one_stream`: one chunk on 32 threads alone (what rounds 1-5 reported as the baseline)."""
    import threading
    import torch
    from oracle import logmel as olm
    from oracle.whisper import OracleWhisper
    if weights is None:
        from faster_whisper_amd import synthetic_weights
        weights = synthetic_weights(cfg, seed=1234)
    host_cores = os.cpu_count() or 1
    team = min(host_cores, 32)    # more threads per matmul only add synchronisation cost at these sizes
    if os.environ.get("FWAMD_CPU_BASELINE_TEAM"):          # (test seam: several streams on a small host)
        team = max(1, min(host_cores, int(os.environ["FWAMD_CPU_BASELINE_TEAM"])))
    torch.set_num_threads(team)
    oracle = OracleWhisper(cfg, weights, emulate_fp16=False)
    n_meas = 8

    def timed(fn):
        t0 = time.perf_counter()
        r = fn()
        return r, time.perf_counter() - t0

    def sample(chunk):
        feats, t_mel = timed(lambda: olm.log_mel_chunks([chunk], cfg.n_mels))
        enc, t_enc = timed(lambda: oracle.encode(feats))             # the whole encoder of one chunk, measured
        kw = dict(gen_kw)
        kw.pop("return_scores", None)
        kw.pop("return_no_speech_prob", None)

        def gen(n):
            kw["max_length"] = len(prompt) + n
            kw["min_new_tokens"] = n
            return timed(lambda: oracle.generate(enc, [prompt], **kw))[1]

        t1, tn = gen(1), gen(1 + n_meas)
        per_step = max(1e-6, (tn - t1) / n_meas)
        fixed = max(0.0, t1 - per_step)              # cross-K/V projection + prompt forward
        return {"t_mel": t_mel, "t_enc": t_enc, "fixed": fixed, "per_step": per_step,
                "total": t_mel + t_enc + fixed + per_step * L}

    one = sample(chunks[0])

    def describe(r):
        return (f"numpy log-mel {r['t_mel']:.2f}s; the whole encoder measured {r['t_enc']:.1f}s; cross-KV+prompt "
                f"{r['fixed']:.2f}s; {n_meas} beam-{beam} steps measured ({r['per_step'] * 1e3:.0f} ms/step) -> {L} steps")

    out = {"value": round(30.0 / one["total"], 4), "unit": "audio-seconds per wall-second", "cores": team,
           "host_cores": host_cores, "kind": "port", "oracle_sha16": oracle_rev(),
           "one_stream": {"value": round(30.0 / one["total"], 4), "cores": team,
                          "sample": "1 chunk (30 s) alone on the box: " + describe(one)}}
    streams = host_cores // team
    if streams > 1:
        res = [None] * streams
        go = threading.Barrier(streams)

        def work(i):
            torch.set_num_threads(team)          # (per calling thread: every stream runs its own team)
            go.wait()
            res[i] = sample(chunks[i % len(chunks)])

        ths = [threading.Thread(target=work, args=(i,)) for i in range(streams)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        wall = time.perf_counter() - t0
        if all(r is not None for r in res):
            out["value"] = round(sum(30.0 / r["total"] for r in res), 4)
            out["cores"] = streams * team
            slow = max(res, key=lambda r: r["total"])
            out["all_cores"] = {"streams": streams, "threads_per_stream": team, "sample_wall_s": round(wall, 1),
                                "per_stream_value": [round(30.0 / r["total"], 4) for r in res],
                                "slowest_stream": describe(slow)}
    out["sample"] = (f"{out['cores'] // team} concurrent stream(s) x {team} threads, one 30 s chunk of the batch each (the oracle "
                     f"decodes chunk by chunk), sum of the streams' rates; per stream: log-mel + the whole encoder + cross-KV + "
                     f"prompt measured, {n_meas} beam-{beam} steps measured and extrapolated to {L} (a step's cost does not "
                     f"depend on its index: KV-cached); one stream alone: {describe(one)}; torch fp32 restatement (oracle/) on "
                     f"{out['cores']} of the box's {host_cores} hardware threads, not CTranslate2 (absent offline: BASELINE.md "
                     "section 3)")
    return out


if __name__ == "__main__":
    main()
