"""bench.main() itself, end to end, as a world_size-2 `gloo` run on the CPU: what an 8-GPU run exercises for the
first time besides the kernels -- frame sharding over the ranks, the flat gradient all-reduce in every step, the barrier +
max-over-ranks timing, the process-group shutdown and ONE JSON line from rank 0 only -- must not be able to fail on plumbing.
Operators = the CPU oracle (tests only; `bench.main(..., plumbing=True)` takes whatever backend is installed and runs on the CPU);
nothing here is a measurement.  Reference launch: tools/scripts/dist_train.sh:3 (one process per GPU, torch.distributed)."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VIRCONV_NUMA_BIND="0")
    torch.set_num_threads(2)
    import bench
    from oracle.backend import OracleBackend
    from virconv_amd import ops, parallel
    ops.set_backend(OracleBackend())
    res = bench.main(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--batch-size", "1", "--no-cpu-baseline",
                      "--family-steps", "0"], plumbing=True)
    parallel.shutdown()
    import torch.distributed as dist
    assert not dist.is_initialized()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


@pytest.mark.timeout(900)
def test_bench_main_runs_end_to_end_on_two_gloo_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r1 is None, "only rank 0 reports"
    assert r0["n_gpus"] == 2 and r0["steps"] == 1 and r0["warmup"] == 0 and r0["scaling"] == "weak"
    assert r0["config"]["frames_per_gpu"] == 1 and r0["config"]["global_batch"] == 2 and r0["config"]["parallelism"] == "dp2"
    assert r0["metric"].startswith("KITTI frames/sec") and r0["unit"] == "frames/s" and r0["higher_is_better"] is True
    # whole-job aggregate: 2 frames over the slower rank's time
    assert abs(r0["value"] - 2 / (r0["ms_per_step"] * 1e-3)) <= 1e-2 * r0["value"]
    assert r0["cpu_baseline"] is None and r0["vs_baseline"] is None
    json.dumps(r0)   # one serialisable line
