"""Worker of tests/test_two_ranks_one_gpu.py: one of TWO ranks that share the one GPU of the box (launched under torch.distributed.run,
VIRCONV_DIST_BACKEND = gloo unless RCCL accepts two ranks on a device).  Different frame shards per rank; the gradients behind
parallel.FlatGradAllReduce are the MEAN of the two ranks' local gradients, bit for bit (tools/train.py:141, common_utils.py:141-154)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from virconv_amd import parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

rank, local_rank, world = parallel.init_distributed()
assert world == 2
dev = torch.device("cuda", local_rank % torch.cuda.device_count())
torch.cuda.set_device(dev)
seeds = parallel.shard_frames(list(range(4)), rank, world)       # 2 frames per rank, different ones
batch = bench.make_batch(seeds, dev, training=True)
torch.manual_seed(0)
model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
sync = parallel.FlatGradAllReduce(model)                          # broadcasts rank 0's parameters
lw = bench.make_loss_weights(dev)
torch.manual_seed(100 + rank)
bd = dict(batch)
bd["voxel_features"] = batch["voxel_features"].clone()
loss = bench.synthetic_loss(model(bd), lw)
loss.backward()
params = [p for p in model.parameters() if p.requires_grad]
local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
sync()
got = torch.cat([p.grad.reshape(-1) for p in params])
both = [torch.empty_like(local) for _ in range(2)]
dist.all_gather(both, local)
want = (both[0] + both[1]) / 2
assert not torch.equal(both[0], both[1]), "the two ranks computed the same gradient: shards are not different"
assert torch.equal(got, want), f"rank {rank}: synchronised gradient != mean of the local gradients ({int((got != want).sum())} elements)"
w0 = torch.cat([p.detach().reshape(-1) for p in params]).clone()
both_w = [torch.empty_like(w0) for _ in range(2)]
dist.all_gather(both_w, w0)
assert torch.equal(both_w[0], both_w[1]), "parameters differ between the ranks after the constructor's broadcast"
n_vox = [None, None]
dist.all_gather_object(n_vox, int(batch["voxel_features"].shape[0]))
torch.cuda.synchronize()
print(f"rank {rank}: frames {seeds} voxels {n_vox[rank]} loss {float(loss):.6f} backend {dist.get_backend()}", flush=True)
if rank == 0:
    assert n_vox[0] != n_vox[1], n_vox
    print("TWO_RANK_OK voxels", n_vox, flush=True)
parallel.shutdown()
