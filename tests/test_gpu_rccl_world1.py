"""The `nccl` (= RCCL) branch of the N > 1 path, executed on hardware with the ONE GPU a test box has: a
torch.distributed job of one rank runs bench.py's whole multi-rank control flow — RCCL communicator on cuda:0, weight
blob broadcast into HBM and the model built from that copy (`fw_model_create_from_blob_dev`), per-step result gather,
MAX-over-ranks timing, the sharded recording (`BatchedInferencePipeline.transcribe(shard=True)`: block partition, one
gather of the result records) — and the sharded recording must yield exactly what the same recording yields unsharded.
(World size 2 of the same control flow runs on gloo: tests/test_bench_dist_gloo.py, tests/test_sharding_gloo.py; the
driver's 8-GPU bench is the first run with real peers.)  SURVEY.md section 8e."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["FW_ROOT"])
import numpy as np
import bench
out = bench.main(["--gpus", "1", "--model", "micro", "--batch", "4", "--beam", "5", "--steps", "6", "--warmup", "1",
                  "--workers", "2", "--new-tokens", "12", "--sharded-chunks", "11", "--no-profile-pass"])
import torch.distributed as dist
assert not dist.is_initialized()          # bench tore its process group down
# the same recording, same model geometry and weights, no process group: the serial result
os.environ.pop("FWAMD_DIST_AT_WORLD_1")
from faster_whisper_amd import get_config
args = bench.parse_args(["--model", "micro", "--batch", "4", "--beam", "5", "--workers", "2"])
cfg = get_config("micro")
model, _ = bench.build_backend(args, cfg, 0, 1, 0)
serial = bench.pipeline_rtf(model, cfg, 11, 4, 5, 12, shard=False)
print("SERIAL " + json.dumps(serial), flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_one_rank_over_rccl(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), FW_ROOT=ROOT, FWAMD_DIST_AT_WORLD_1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["value"] > 0 and j["verified"] is True
    assert "pipeline" not in j and "cpu_baseline" not in j          # the N > 1 control flow was taken
    sh = j["sharded_recording"]
    assert "error" not in sh, sh
    assert sh["segments"] >= 11 and sh["tokens"] > 0 and sh["scaling"] == "strong"      # 11 chunks, at least a segment each
    serial = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("SERIAL ")][0][7:])
    assert "error" not in serial, serial
    # what rank 0 assembled from the RCCL gather == what the unsharded pipeline yields
    assert sh["segments"] == serial["segments"] and sh["tokens"] == serial["tokens"]
    assert sh["digest"] == serial["digest"], (sh, serial)
    # round 6: the weight blob travels device -> device (sharding.broadcast_blob_dev wraps fw_model_blob's allocation, no
    # host image); the line says how long the collective alone took, apart from model_load_s
    # (model_load_s is rounded to 0.1 s in the line; micro loads in milliseconds)
    assert 0.0 <= j["config"]["blob_broadcast_s"] <= j["config"]["model_load_s"] + 0.1
    assert "blob_broadcast (inside the phase above" in p.stderr
