"""RCCL itself under the driver's GPU tests (VERDICT r4 "next" #7): a one-rank 'nccl' process group on the GPU box.
(a) parallel.FlatGradAllReduce and stock DistributedDataParallel on the real backbone give the plain run's gradients bit for bit and
    the process shuts down cleanly (tools/train.py:141, common_utils.py:141-154 in the reference);
(b) `bench.py --gpus 1` launched the way the driver launches N > 1 -- under torch.distributed.run -- with the collective forced on
    prints exactly ONE JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["VIRCONV_FORCE_DDP"] = "1"
    return env


def test_flat_all_reduce_and_stock_ddp_over_rccl_world_1_match_the_plain_run_and_shut_down_cleanly():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), str(_free_port())], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    assert "RCCL_WORLD1_OK" in r.stdout and "flat:" in r.stdout and "ddp:" in r.stdout and "syncbn:" in r.stdout
    assert "watchdog" not in r.stderr.lower() and "core dumped" not in r.stderr.lower(), r.stderr[-3000:]


def test_bench_under_torch_distributed_run_with_the_collective_on_prints_exactly_one_json_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--family-steps", "0", "--exact-steps", "0"]
    env = _env()
    env["VIRCONV_SETTLE_SEC"] = "0.2"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["steps"] == 4 and res["value"] > 0 and res["unit"] == "frames/s"
    assert "watchdog" not in r.stderr.lower(), r.stderr[-3000:]
