"""Analysis helper (not a test): what a PAIR-COMPACTED gather-GEMM (VERDICT r2 #1b/c) would issue on the VirConv-L rulebooks of
one synthetic frame, against the shipped output-stationary kernel.  Oracle rulebooks, CPU.  Run:

    python tests/analysis_pair_compaction.py          (numbers quoted in DESIGN.md 4.10)

Per layer, in 16-row MFMA tile-offset units per 64 output rows (one unit = CK/4 * CN/16 MFMAs):
  useful        pairs / 16                                    (no padding at all)
  v2            sum over the four 16-row tiles of |union of active offsets|   (what gather_gemm_v2 issues; natural row order)
  v2 sorted     the same after the 2048-row windowed mask sort (vc_row_order)
  pc32 / pc64 / pc256  per offset ceil(pairs of the 32- / 64- / 256-row group / 16): compaction over one wave's rows (accumulators can stay
                wave-private: no barrier, deterministic) / over a 4-wave block (needs a per-offset barrier or LDS-resident
                accumulators shared by the waves); pc256 is reported per 64 rows
and the number of 64-row wave units the layer has at bs 4 (x4 the single frame) against the chip's 1024 SIMDs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import geometry as geo  # noqa: E402
from oracle import sparse_ref as sr  # noqa: E402
from virconv_amd import data, synth  # noqa: E402

fr = synth.make_frame(0)
pts = data.prepare_frame(fr["points_lidar"], fr["points_virtual"], training=True, rng=np.random.default_rng(10000))
vox, coords, num = geo.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
idx = np.concatenate([np.zeros((len(coords), 1), np.int32), coords.astype(np.int32)], 1)
shape = [81, 1600, 1408]


def units(act, g):
    kv, n = act.shape
    m = n // g * g
    return act[:, :m].reshape(kv, m // g, g)


def stats(name, pair):
    kv, n = pair.shape
    act = pair >= 0
    useful = act.sum() / 16 / (n / 64)
    v2 = units(act, 16).any(2).sum() / (n // 16 * 16 / 64)
    key = np.zeros(n, np.int64)
    for k in range(kv):
        key |= act[k].astype(np.int64) << k
    o = np.concatenate([s + np.argsort(key[s:s + 2048], kind="stable") for s in range(0, n, 2048)])
    v2s = units(act[:, o], 16).any(2).sum() / (n // 16 * 16 / 64)
    pc32 = np.ceil(units(act, 32).sum(2) / 16).sum() / (n // 32) * 2
    pc64 = np.ceil(units(act, 64).sum(2) / 16).sum() / (n // 64)
    pc256 = np.ceil(units(act, 256).sum(2) / 16).sum() / (n // 256) / 4
    print(f"{name:10s} N {n:6d} pairs/row {act.sum() / n:5.2f} | per 64 rows: useful {useful:5.1f}  v2 {v2:5.1f}  v2 sorted {v2s:5.1f}  "
          f"pc32 {pc32:5.1f}  pc64 {pc64:5.1f}  pc256 {pc256:5.1f} | 64-row waves at bs 4: {4 * n // 64:5d} ({4 * n / 64 / 1024:.1f} per SIMD)")


cur, cs = idx, shape
stats("s1 subm", sr.subm_rulebook(cur, cs, (3, 3, 3)))
for st, pad in ((2, (1, 1, 1)), (3, (1, 1, 1)), (4, (0, 1, 1))):
    out = sr.sparse_rulebook(cur, cs, 1, (3, 3, 3), (2, 2, 2), pad)
    stats(f"s{st} down", out[2])
    cur, cs = out[0], list(out[1])
    stats(f"s{st} subm", sr.subm_rulebook(cur, cs, (3, 3, 3)))
