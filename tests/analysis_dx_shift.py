"""Analysis helper (not a test): how often is the dx = -1 / +1 fragment of a 16-row MFMA tile the centre (dx = 0) fragment of the
same (dz, dy) group shifted by one row?  Rows of a strided conv's output are in ascending (b, z, y, x) order, so for x-adjacent
output rows pair[3g][i] == pair[3g+1][i-1] and pair[3g+2][i] == pair[3g+1][i+1]: those lanes could take their rows from the centre
fragment with a DPP row shift instead of a gather.  Prints, per SubM table of a synthetic KITTI frame, the fraction of lanes that
match and the source rows a fix-up gather would still touch per (tile, offset).  Uses the CPU oracle.
Run: python tests/analysis_dx_shift.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import geometry as geo, sparse_ref as sr
from virconv_amd import data, synth

fr = synth.make_frame(0)
pts = data.prepare_frame(fr["points_lidar"], fr["points_virtual"], training=True, rng=np.random.default_rng(10000))
vox, coords, num = geo.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
idx = np.concatenate([np.zeros((len(coords), 1), np.int32), coords.astype(np.int32)], 1)


def stats(name, pair, tm=16):
    kv, n = pair.shape
    nb = n // tm
    p = pair[:, :nb * tm].reshape(kv, nb, tm)
    tot = match = fix_rows = full_rows = tiles_active = tiles_clean = 0
    for g in range(kv // 3):
        m, c, pl = p[3 * g], p[3 * g + 1], p[3 * g + 2]
        for side, shifted in ((m, np.concatenate([np.full((nb, 1), -2), c[:, :-1]], 1)), (pl, np.concatenate([c[:, 1:], np.full((nb, 1), -2)], 1))):
            act = (side >= 0).any(1)                       # (tile, offset) pairs that are issued at all
            eq = side == shifted                           # lanes served by the shifted centre fragment (incl. matching -1)
            need = (~eq) & (side >= 0)                     # lanes a fix-up gather still has to fetch
            tot += (side[act] >= 0).sum()
            match += ((side >= 0) & eq)[act].sum()
            fix_rows += need[act].sum()
            full_rows += (side[act] >= 0).sum()
            tiles_active += act.sum()
            tiles_clean += (act & ~need.any(1)).sum()
    print(f"{name:12s} N={n:6d}: dx=+-1 rows taken from the shifted centre fragment {100 * match / max(tot, 1):5.1f} %, "
          f"fix-up rows per issued (tile, offset) {fix_rows / max(tiles_active, 1):4.2f} (a full gather: {full_rows / max(tiles_active, 1):4.2f}), "
          f"(tile, offset) pairs needing no gather at all {100 * tiles_clean / max(tiles_active, 1):5.1f} %")


cur, cs = idx, [81, 1600, 1408]
stats("s1 subm", sr.subm_rulebook(cur, cs, (3, 3, 3)))
for st, pad in ((2, (1, 1, 1)), (3, (1, 1, 1)), (4, (0, 1, 1))):
    out = sr.sparse_rulebook(cur, cs, 1, (3, 3, 3), (2, 2, 2), pad)
    cur, cs = out[0], list(out[1])
    stats(f"s{st} subm", sr.subm_rulebook(cur, cs, (3, 3, 3)))
