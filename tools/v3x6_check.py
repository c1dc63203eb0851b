"""v3 (LDS row windows) with split operands vs v2 with split operands: bits and time, SubM layers of VirConv-L (experiments build)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from virconv_amd import ops, synth
from tools.kbench import timeit
dev = torch.device("cuda", 0)
be = ops.get_backend()
ops.WINDOW_GATHER = True
assert be.lib.vc_debug_set(b"conv_autopack", 1) == 0
batch = bench.make_batch([0, 1, 2, 3], dev, training=True)
idx = batch["voxel_coords"].int()
shape = [int(v) for v in (np.asarray(synth.GRID_SIZE)[::-1] + [1, 0, 0])]
cur, cs = idx, shape
g = torch.Generator(device="cpu").manual_seed(0)
for stage, (cin, cout) in enumerate([(16, 32), (32, 64), (64, 64)], start=2):
    pad = (0, 1, 1) if stage == 4 else (1, 1, 1)
    rb = ops.build_sparse_rulebook(cur, cs, 4, (3, 3, 3), (2, 2, 2), pad, 1)
    cur, cs = rb.out_indices, list(rb.out_shape)
    rb3 = ops.build_subm_rulebook(cur, cs, (3, 3, 3), 1, False)
    for ci, co in ((cout, cout // 2), (cout // 2, cout // 2)):
        if ci % 32:
            continue
        x = torch.randn((rb3.n_in, ci), generator=g).to(dev)
        w = (torch.randn((co, 27, ci), generator=g) / np.sqrt(27 * ci)).to(dev).reshape((co, 3, 3, 3, ci))
        dy = torch.randn((rb3.n_out, co), generator=g).to(dev)
        res = {}
        for name, srt in (("v2", False), ("v3", True)):
            for wr in ((32, 24) if srt else (32,)):
                assert be.lib.vc_debug_set(b"conv_winrows", wr) == 0
                y = be.conv_forward(x, w, rb3.pair_fwd, order=None, sorted_rows=srt)
                t = timeit(lambda: be.conv_forward(x, w, rb3.pair_fwd, order=None, sorted_rows=srt), 20)
                dxv = be.conv_backward_input(dy, w, rb3.pair_fwd, rb3.n_in, True, rb3.centre, None, order=None, sorted_rows=srt)
                tb = timeit(lambda: be.conv_backward_input(dy, w, rb3.pair_fwd, rb3.n_in, True, rb3.centre, None, order=None, sorted_rows=srt), 20)
                res[(name, wr)] = (y, t, dxv, tb)
        y2, t2, d2, tb2 = res[("v2", 32)]
        line = f"s{stage} {ci}->{co} rows {rb3.n_in}: v2 fwd {t2:.1f} bwd {tb2:.1f} us"
        for wr in (32, 24):
            y3, t3, d3, tb3 = res[("v3", wr)]
            line += f" | v3[{wr}] fwd {t3:.1f} ({t2 / t3:.2f}x, bits {'==' if torch.equal(y2, y3) else '!='}) bwd {tb3:.1f} ({tb2 / tb3:.2f}x, bits {'==' if torch.equal(d2, d3) else '!='})"
        print(line, flush=True)
