"""Is a plan finished under a running backward pass (split products on) the same as one built in place?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from virconv_amd import backbone as bb, ops, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

dev = torch.device("cuda", 0)
be = ops.get_backend()
lw = bench.make_loss_weights(dev)
assert be.lib.vc_debug_set(b"bw_split", 0) == 0 and be.lib.vc_debug_set(b"f32_split", int(os.environ.get("FS", "1"))) == 0
batch = bench.make_batch([0, 1], dev, training=True)
torch.manual_seed(3)
probe = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
p0 = probe.build_plan(batch["voxel_coords"], 2, batch["calib"], batch["aug_param"], batch)
bb.join_plan(p0)
batch["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].clone() for bi in range(3)}
torch.cuda.synchronize()


def tensors(plan):
    out = {"in": plan["in_indices"]}
    for si, st in enumerate(plan["stages"]):
        for k in ("out_indices", "uv", "keep", "kept_indices"):
            if st.get(k) is not None:
                out[f"s{si}.{k}"] = st[k]
        for grp in ("rb3d", "rb2d"):
            for key, rb in st[grp].items():
                for name in ("pair_fwd", "pair_bwd", "rep", "grp_plan", "order_fwd", "order_bwd", "out_indices"):
                    t = getattr(rb, name, None)
                    if t is not None:
                        out[f"s{si}.{key}.{name}"] = t
    for key, rb in plan["conv_out"].items():
        for name in ("pair_fwd", "pair_bwd", "order_bwd", "out_indices"):
            t = getattr(rb, name, None)
            if t is not None:
                out[f"out.{name}"] = t
    return out


for rep in range(8):
    torch.manual_seed(0)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
    bench.train_step(model, opt, batch, lw, next_batch=batch)       # step 0: its backward runs beside the finish of the next plan
    ahead = model._take_ahead(batch["voxel_coords"], "")
    assert ahead is not None
    bb.join_plan(ahead)
    torch.cuda.synchronize()
    fresh = model.build_plan(batch["voxel_coords"], 2, batch["calib"], batch["aug_param"], batch)
    bb.join_plan(fresh)
    torch.cuda.synchronize()
    ta, tf = tensors(ahead), tensors(fresh)
    assert ta.keys() == tf.keys()
    bad = [(k, int((ta[k] != tf[k]).sum())) for k in ta if ta[k].shape != tf[k].shape or not torch.equal(ta[k], tf[k])]
    print(f"rep {rep}: {len(ta)} structures, differing: {bad}")
    pa, pf = ahead["_arenas"][0][64:64 + 64].view(torch.float32), fresh["_arenas"][0][64:64 + 64].view(torch.float32)
    print("   params equal:", torch.equal(pa, pf), (pa - pf).abs().max().item())
    for k, _ in bad:
        if k.endswith(".uv"):
            rows = (ta[k] != tf[k]).any(1).nonzero().squeeze(1)
            si = int(k[1])
            third = ops.project_uv(ta[f"s{si}.out_indices"], batch["calib"], batch["aug_param"], 2, 2 ** si)
            torch.cuda.synchronize()
            print("    stand-alone projection of the same coordinates equals: ahead", bool(torch.equal(third, ta[k])), "fresh", bool(torch.equal(third, tf[k])))
            print("   ", k, "rows", rows[:8].tolist(), "... span", int(rows.min()), int(rows.max()), "ahead", ta[k][rows[:4]].tolist(), "fresh", tf[k][rows[:4]].tolist())
