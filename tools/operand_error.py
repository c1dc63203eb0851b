"""Error of the 16-bit-MFMA-operand modes (BASELINE configs[4]) against the exact fp32 train step: loss, features, worst gradients."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from virconv_amd import ops, synth
from virconv_amd.backbone import VirConvL8x
for seed in (0, 1):
    b1 = bench.make_batch([seed], torch.device("cuda", 0), training=True)
    torch.manual_seed(3)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    lw = bench.make_loss_weights("cuda")
    state = {k: v.clone() for k, v in model.state_dict().items()}
    def run(mode):
        model.load_state_dict(state); model.zero_grad(set_to_none=True)
        bd = dict(b1); bd["voxel_features"] = b1["voxel_features"].clone()
        torch.manual_seed(5)
        ops.MFMA_OPERAND = mode
        out = model(bd)
        loss = (out["encoded_spconv_tensor"].dense() * lw["dense"]).sum()
        for name, t in out["multi_scale_3d_features"].items():
            loss = loss + (t.features * lw[name]).sum()
        loss.backward()
        ops.MFMA_OPERAND = "f32"
        return float(loss), {n: t.features.detach().clone() for n, t in out["multi_scale_3d_features"].items()}, {k: p.grad.clone() for k, p in model.named_parameters()}
    l32, f32_, g32 = run("f32")
    for mode in ("f16", "bf16"):
        l, f, g = run(mode)
        fe = max(float((f[n]-f32_[n]).abs().max())/max(1.0,float(f32_[n].abs().max())) for n in f)
        ge = sorted(((float((g[k]-g32[k]).abs().max())/max(1e-6,float(g32[k].abs().max())), k) for k in g), reverse=True)
        a = torch.cat([g[k].reshape(-1) for k in g32]).double()
        b = torch.cat([g32[k].reshape(-1) for k in g32]).double()
        cos = float((a @ b) / (a.norm() * b.norm()))
        print(seed, mode, "loss", l32, l, "feat", fe, "grad cosine", cos, "rel l2", float((a - b).norm() / b.norm()),
              "worst max-norm", ge[:3])
