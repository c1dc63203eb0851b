"""The dx shift of the gather-GEMM (conv_kernels.hip, DXS) with split operands vs the product kernel: bits and time, 3-D SubM layers of
VirConv-L (experiments build: VIRCONV_LIB=.../libvirconv_hip_exp.so).  Round 6: the shift halves the gathered rows; with fp32 MFMA products it
lost to its register and DPP cost (rounds 2-3) -- does it still, now that the L1's miss handling is what the kernels wait for?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from virconv_amd import ops, synth
from tools.kbench import timeit
dev = torch.device("cuda", 0)
be = ops.get_backend()
assert be.lib.vc_debug_set(b"conv_autopack", 1) == 0
batch = bench.make_batch([0, 1, 2, 3], dev, training=True)
idx = batch["voxel_coords"].int()
shape = [int(v) for v in (np.asarray(synth.GRID_SIZE)[::-1] + [1, 0, 0])]
cur, cs = idx, shape
g = torch.Generator(device="cpu").manual_seed(0)
tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
for stage, (cin, cout) in enumerate([(16, 32), (32, 64), (64, 64)], start=2):
    pad = (0, 1, 1) if stage == 4 else (1, 1, 1)
    rb = ops.build_sparse_rulebook(cur, cs, 4, (3, 3, 3), (2, 2, 2), pad, 1)
    cur, cs = rb.out_indices, list(rb.out_shape)
    rb3 = ops.build_subm_rulebook(cur, cs, (3, 3, 3), 1, False)
    for ci, co in ((cout, cout // 2), (cout // 2, cout // 2)):
        x = torch.randn((rb3.n_in, ci), generator=g).to(dev)
        w = (torch.randn((co, 27, ci), generator=g) / np.sqrt(27 * ci)).to(dev).reshape((co, 3, 3, 3, ci))
        dy = torch.randn((rb3.n_out, co), generator=g).to(dev)
        res = {}
        for d in (0, 1):
            assert be.lib.vc_debug_set(b"conv_dxs", d) == 0
            y = be.conv_forward(x, w, rb3.pair_fwd, order=None)
            t = timeit(lambda: be.conv_forward(x, w, rb3.pair_fwd, order=None), 20)
            dx = be.conv_backward_input(dy, w, rb3.pair_fwd, rb3.n_in, True, rb3.centre, None, order=None)
            tb = timeit(lambda: be.conv_backward_input(dy, w, rb3.pair_fwd, rb3.n_in, True, rb3.centre, None, order=None), 20)
            res[d] = (y, t, dx, tb)
            tot[d][0] += t; tot[d][1] += tb
        (y0, t0, d0, b0), (y1, t1, d1, b1) = res[0], res[1]
        print(f"s{stage} {ci}->{co} rows {rb3.n_in}: product fwd {t0:.1f} bwd {b0:.1f} us | dx shift fwd {t1:.1f} ({t0 / t1:.2f}x, bits "
              f"{'==' if torch.equal(y0, y1) else '!='}) bwd {b1:.1f} ({b0 / b1:.2f}x, bits {'==' if torch.equal(d0, d1) else '!='})", flush=True)
print(f"TOTAL product fwd {tot[0][0]:.0f} bwd {tot[0][1]:.0f} us | dx shift fwd {tot[1][0]:.0f} bwd {tot[1][1]:.0f} us")
