"""N>1 path on CPU: world_size-2 `gloo` run of the data-parallel wrapper (frames shard across ranks, the only exchange is
the DDP gradient all-reduce).  Operators = the CPU oracle (tests only); what is under test is virconv_amd.parallel +
the autograd/DDP wiring of the backbone."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode="ddp"):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from helpers import GRID, MODEL_CFG, fill_parameters, golden_batch, load_golden
    from oracle.backend import OracleBackend
    from virconv_amd import ops, parallel
    from virconv_amd.backbone import VirConvL8x

    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    ops.set_backend(OracleBackend())
    g = load_golden()
    full = golden_batch(g)
    mine = parallel.shard_frames([0, 1], rank, world)  # the fixture holds 2 frames: one per rank
    assert mine == [rank]
    sel = full["voxel_coords"][:, 0] == mine[0]
    coords = full["voxel_coords"][sel].clone()
    coords[:, 0] = 0
    batch = {"batch_size": 1, "voxel_features": full["voxel_features"][sel].clone(), "voxel_coords": coords,
             "calib": [full["calib"][mine[0]]], "aug_param": full["aug_param"][mine[0]:mine[0] + 1]}
    model = VirConvL8x(dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop"), 8, GRID).train()
    fill_parameters(model, 7 + rank)          # deliberately different: DDP must broadcast rank 0's parameters
    if mode == "ddp":
        ddp, sync = parallel.wrap_ddp(model, "cpu"), None
    else:  # the exchange bench.py uses: one flat all-reduce of the packed gradients
        ddp, sync = model, parallel.FlatGradAllReduce(model)
    out = ddp(batch)
    partial = mode == "flat_partial" or (mode == "flat_partial_one" and rank == 1)
    if partial:   # a loss that leaves vir_conv3 / vir_conv4 / conv_out WITHOUT a gradient (p.grad is None)
        loss = out["multi_scale_3d_features"]["x_conv2"].features.mean()
    else:
        loss = out["encoded_spconv_tensor"].features.square().mean() + out["multi_scale_3d_features"]["x_conv2"].features.mean()
    loss.backward()
    if partial:
        assert model.conv_out[0].weight.grad is None and model.vir_conv2.d3_conv1[0].weight.grad is not None
    if sync is not None:
        sync()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    w0 = model.vir_conv1.d3_conv1[0].weight.detach().clone()
    torch.save({"grads": grads, "w0": w0, "loss": float(loss)}, os.path.join(out_dir, f"rank{rank}.pt"))
    t = parallel.max_over_ranks(float(rank + 1), "cpu")
    assert t == float(world)
    parallel.barrier()
    parallel.shutdown()        # what bench.py calls before it prints its line: the group is gone, a second call is a no-op
    assert not dist.is_initialized()
    parallel.shutdown()


def _single(frame):
    from helpers import GRID, MODEL_CFG, fill_parameters, golden_batch, load_golden
    from oracle.backend import OracleBackend
    from virconv_amd import ops
    from virconv_amd.backbone import VirConvL8x
    with ops.use_backend(OracleBackend()):
        g = load_golden()
        full = golden_batch(g)
        sel = full["voxel_coords"][:, 0] == frame
        coords = full["voxel_coords"][sel].clone()
        coords[:, 0] = 0
        batch = {"batch_size": 1, "voxel_features": full["voxel_features"][sel].clone(), "voxel_coords": coords,
                 "calib": [full["calib"][frame]], "aug_param": full["aug_param"][frame:frame + 1]}
        model = VirConvL8x(dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop"), 8, GRID).train()
        fill_parameters(model, 7)
        out = model(batch)
        loss = out["encoded_spconv_tensor"].features.square().mean() + out["multi_scale_3d_features"]["x_conv2"].features.mean()
        loss.backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_ddp_world2_gloo_grad_allreduce(tmp_path, mode):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["w0"], r1["w0"])  # parameters were broadcast from rank 0
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), f"{k}: ranks disagree after all-reduce"
    g0, g1 = _single(0), _single(1)
    worst = 0.0
    for k in g0:
        exp = (g0[k] + g1[k]) / 2  # DDP averages
        worst = max(worst, float((r0["grads"][k] - exp).abs().max() / max(1e-6, float(exp.abs().max()))))
    assert worst < 1e-4, worst


def test_shard_frames():
    from virconv_amd import parallel
    ids = list(range(8))
    shards = [parallel.shard_frames(ids, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == ids and all(len(s) == 2 for s in shards)
    assert parallel.shard_frames(ids, 0, 1) == ids


def test_numa_binding_helpers_are_safe_without_a_gpu(monkeypatch):
    """bind_to_gpu_numa is best effort: it must never raise and must honour VIRCONV_NUMA_BIND=0; the cpulist parser handles
    the kernel's 'a-b,c' syntax."""
    from virconv_amd import parallel
    assert parallel._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parallel._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    msg = parallel.bind_to_gpu_numa(0)
    assert isinstance(msg, str) and os.sched_getaffinity(0) <= before
    os.sched_setaffinity(0, before)
    monkeypatch.setenv("VIRCONV_NUMA_BIND", "0")
    assert parallel.bind_to_gpu_numa(0) == "off"


@pytest.mark.timeout(600)
def test_flat_allreduce_with_parameters_that_received_no_gradient(tmp_path):
    """a parameter whose .grad is None must not change the size of the flat buffer on one rank only.  ADVICE r2: a
    parameter that is None on EVERY rank goes back to None after the exchange (the optimizer then skips it, as under
    DistributedDataParallel or on one GPU); one that is None on SOME ranks is averaged with zeros from those ranks."""
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "flat_partial"), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in r0["grads"]:
        assert (r0["grads"][k] is None) == (r1["grads"][k] is None), k
        if r0["grads"][k] is not None:
            assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    assert r0["grads"]["conv_out.0.weight"] is None
    assert float(r0["grads"]["vir_conv2.d3_conv1.0.weight"].abs().max()) > 0.0


@pytest.mark.timeout(600)
def test_flat_allreduce_when_only_one_rank_lacks_a_gradient(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "flat_partial_one"), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in r0["grads"]:
        assert r0["grads"][k] is not None and torch.equal(r0["grads"][k], r1["grads"][k]), k
    assert float(r0["grads"]["conv_out.0.weight"].abs().max()) > 0.0      # rank 0's gradient / 2 (rank 1 contributed zeros)
