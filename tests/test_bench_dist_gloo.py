"""Dry run of bench.py's N > 1 control flow on CPU: two processes over `gloo`, a scripted backend instead of the
HIP engine.  Everything between the collectives is the real code path of the 8-GPU run the driver launches — rank /
world from the environment, weight-blob pack on rank 0 + broadcast + the arguments fw_model_create_from_blob_dev
would receive, the worker pool, per-step result gather, the max-over-ranks timing, the cap case, the sharded
1 h recording (BatchedInferencePipeline.transcribe(shard=True)), one JSON line from rank 0 — so the first real
multi-GPU run cannot die on plumbing.  (The engine itself is covered by the -m gpu tests.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SEAM = r'''
import json, os, struct, sys, threading
import numpy as np

DIST_BACKEND = "gloo"


class Result:
    def __init__(self, ids, score, nsp):
        self.sequences_ids, self.scores, self.no_speech_prob = [ids], [score], nsp


class FakeEnc:
    def __init__(self, keys):
        self.keys = keys


class FakeBackend:
    """what bench.py and the batched pipeline touch of backend.Whisper; ids are a pure function of the PCM"""
    def __init__(self, cfg, workers, blob_dev):
        self.config, self.is_multilingual, self.n_mels = cfg, cfg.is_multilingual, cfg.n_mels
        self.device, self.device_index, self.inter_threads = "cpu", [0], workers
        self.blob_dev = blob_dev
        self.lock = threading.Lock()
        self.calls = {"encode": 0, "generate": 0, "chunks": 0}

    def _keys(self, chunks):
        return [int(abs(float(np.asarray(c[:64]).sum())) * 1e4) % 9973 for c in chunks]

    def stage_pcm(self, chunks, replica=0):
        return {"chunks": chunks}

    def free_staged(self, staged):
        pass

    def encode_pcm_staged(self, staged):
        return self.encode_pcm(staged["chunks"])

    def encode_pcm(self, chunks):
        with self.lock:
            self.calls["encode"] += 1
        return FakeEnc(self._keys(chunks))

    def generate(self, enc, prompts, *, max_length=448, **kw):
        with self.lock:
            self.calls["generate"] += 1
            self.calls["chunks"] += len(enc.keys)
        out = []
        for k, p in zip(enc.keys, prompts):
            n = max_length - len(p)
            out.append(Result([(k + 3 * i) % 300 + 10 for i in range(n)], -0.001 * (k % 500) - 0.01, (k % 10) / 20.0))
        return out

    def decode_stats(self):
        c = self.calls
        return {"runs": c["generate"], "requests": c["generate"], "chunks": c["chunks"], "max_run_chunks": 4,
                "decode_batch": 4 * self.inter_threads}

    def synchronize(self):
        pass

    def profile(self, enable=True, replica=0):
        pass

    def profile_report(self, replica=0):
        return {"dec_cross_attn": dict(ms=2.0, launches=4, flops=0.0, bytes=8e9),
                "enc_gemm": dict(ms=1.0, launches=2, flops=1e12, bytes=0.0)}


def factory(args, cfg, rank, world, local_rank):
    """bench.build_backend with the engine replaced: the blob is really packed (host-side packer of libfwamd.so) and
    really broadcast; every rank checks what it would hand to fw_model_create_from_blob_dev"""
    from faster_whisper_amd import pack_blob, synthetic_weights
    from faster_whisper_amd.sharding import broadcast_blob
    blob = None
    if rank == 0:
        blob = pack_blob(cfg, synthetic_weights(cfg, seed=1234), 0)
    t = broadcast_blob(blob, rank, local_rank)
    raw = t.numpy()
    assert raw[:8].tobytes() == b"FWAMDBL1", raw[:8].tobytes()
    total = struct.unpack_from("<q", raw.tobytes()[:32], 16)[0]
    assert total == raw.shape[0], (total, raw.shape)
    assert t.data_ptr() != 0 and t.numel() == total
    return FakeBackend(cfg, args.workers, (t.data_ptr(), t.numel())), None
'''

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FW_ROOT"])
import bench
''' + SEAM + r'''
n_shard, batch = os.environ.get("FW_SHARDED_CHUNKS", "11"), os.environ.get("FW_BATCH", "4")
out = bench.main(["--gpus", os.environ["WORLD_SIZE"], "--model", "micro", "--batch", batch, "--beam", "5",
                  "--steps", os.environ.get("FW_STEPS", "6"),
                  "--warmup", "1", "--workers", "2", "--new-tokens", "12", "--pipeline-chunks", "11", "--sharded-chunks", n_shard,
                  "--no-cpu-baseline"] + os.environ.get("FW_EXTRA_ARGS", "").split(), backend_factory=factory, dist_backend="gloo")
if int(os.environ["RANK"]) != 0:
    print("RANK_DONE", flush=True)
if os.environ.get("FW_SERIAL_DIGEST"):
    # the same recording through the unsharded pipeline on a fresh scripted backend (no process group any more)
    from faster_whisper_amd import get_config
    cfg = get_config("micro")
    serial = bench.pipeline_rtf(FakeBackend(cfg, 2, None), cfg, int(n_shard), int(batch), 5, 12, shard=False)
    print("SERIAL " + json.dumps(serial), flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_bench_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FW_ROOT=ROOT, OMP_NUM_THREADS="1")
        env.pop("FWAMD_DIST_AT_WORLD_1", None)
        if rank == 0:
            env["FW_SERIAL_DIGEST"] = "1"
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert "RANK_DONE" in outs[1][0]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]          # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_batch"] == 8 and j["config"]["workers_per_gpu"] == 2
    assert j["cap_case"]["new_tokens"] == 224 and j["cap_case"]["value"] > 0
    sh = j["sharded_recording"]
    assert "error" not in sh, sh
    # rank 0 yielded every chunk of the recording, in order: one segment per 30 s chunk
    assert sh["segments"] == 11 and sh["scaling"] == "strong" and sh["n_gpus"] == 2 and sh["tokens"] == 11 * 12
    # ... and they are the unsharded pipeline's segments: same digest (ids, avg_logprob, no_speech_prob, in order)
    serial = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("SERIAL ")][0][7:])
    assert "error" not in serial and sh["digest"] == serial["digest"], (sh, serial)
    assert j["roofline"]["kernel"] == "dec_cross_attn" and j["roofline"]["bound"] == "hbm"
    assert "cpu_baseline" not in j and "pipeline" not in j      # N = 1 only


@pytest.mark.timeout(300)
def test_bench_one_rank_takes_the_multi_rank_flow(tmp_path):
    """FWAMD_DIST_AT_WORLD_1=1: ONE rank runs the N > 1 control flow (what tests/test_gpu_rccl_world1.py does over RCCL on a
    GPU box) — here over gloo; the sharded recording, assembled on rank 0 from the gathered records, must carry the digest
    (ids, avg_logprob, no_speech_prob of every segment, in order) of the unsharded pipeline"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               FW_ROOT=ROOT, OMP_NUM_THREADS="1", FWAMD_DIST_AT_WORLD_1="1", FW_SERIAL_DIGEST="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and "pipeline" not in j and "cpu_baseline" not in j      # the N > 1 flow was taken
    sh = j["sharded_recording"]
    serial = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("SERIAL ")][0][7:])
    assert "error" not in sh and "error" not in serial, (sh, serial)
    assert sh["segments"] == serial["segments"] == 11 and sh["tokens"] == serial["tokens"] == 11 * 12
    assert sh["digest"] == serial["digest"]


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 --steps 4` with NO rank environment (the form the driver uses for N = 1): bench.py itself
    starts two ranks under torch.distributed.run (bench.relaunch) and rank 0 prints ONE line with n_gpus: 2.  The ranks
    are fresh `python bench.py` processes; FWAMD_BENCH_SEAM hands them the scripted backend and the gloo backend."""
    seam = tmp_path / "seam.py"
    seam.write_text(SEAM)
    env = dict(os.environ, FWAMD_BENCH_SEAM=str(seam), OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FWAMD_DIST_AT_WORLD_1"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--model", "micro", "--batch", "4", "--beam", "5", "--workers", "2", "--new-tokens", "12",
                        "--pipeline-chunks", "11", "--sharded-chunks", "11", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["config"]["global_batch"] == 8 and j["value"] > 0
    assert j["sharded_recording"]["segments"] == 11


def test_bench_refuses_a_world_that_is_not_gpus(tmp_path):
    """--gpus must be the job size: a launcher that started a different number of ranks is an error, not a line"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "refusing" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def _run_ranks(tmp_path, world, extra_env, timeout=280):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FW_ROOT=ROOT, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", **extra_env)
        env.pop("FWAMD_DIST_AT_WORLD_1", None)
        if rank == 0 and "FW_EXTRA_ARGS" not in extra_env:
            env["FW_SERIAL_DIGEST"] = "1"
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    return outs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_chunks", [120, 121])
def test_bench_eight_ranks_one_hour_recording(tmp_path, n_chunks):
    """BASELINE config C4 as the driver's 8-GPU run shapes it: EIGHT ranks, a 1 h recording = 120 chunks = one 15-chunk
    batch per rank (and 121: an uneven block partition, rank 0 one chunk more), batches of 16.  Eight real processes over
    gloo with the scripted backend: rendezvous, blob broadcast to 8 ranks, one gather of 8 x 2 steps, the MAX over ranks,
    the sharded recording assembled on rank 0 — whose digest must be the unsharded pipeline's."""
    outs = _run_ranks(tmp_path, 8, {"FW_SHARDED_CHUNKS": str(n_chunks), "FW_BATCH": "16", "FW_STEPS": "2"}, timeout=560)
    for r in range(1, 8):
        assert "RANK_DONE" in outs[r][0]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["config"]["global_batch"] == 128 and j["value"] > 0
    sh = j["sharded_recording"]
    assert "error" not in sh, sh
    assert sh["segments"] == n_chunks and sh["n_gpus"] == 8 and sh["tokens"] == n_chunks * 12
    serial = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("SERIAL ")][0][7:])
    assert "error" not in serial and sh["digest"] == serial["digest"], (sh, serial)
    # every rank reported its start-up phases on stderr (what a failing lease is read from)
    for r in range(8):
        assert f"[bench rank {r}/8] init_process_group" in outs[r][1] and "blob_broadcast_and_model" in outs[r][1]


@pytest.mark.timeout(300)
def test_bench_dry_dist(tmp_path):
    """`bench.py --gpus N --dry-dist`: rendezvous -> blob broadcast -> model -> one batch per rank -> one gather -> exit;
    rank 0 prints the seconds of every phase on every rank (the first thing to run on a fresh multi-GPU lease)"""
    outs = _run_ranks(tmp_path, 2, {"FW_EXTRA_ARGS": "--dry-dist"})
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    j = json.loads(lines[0])
    assert j["dry_dist"] is True and j["n_gpus"] == 2 and len(j["phases_s_by_rank"]) == 2
    for ph in j["phases_s_by_rank"]:
        for k in ("init_process_group", "first_barrier", "blob_broadcast_and_model", "first_batch", "gather"):
            assert k in ph, (k, ph)
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
