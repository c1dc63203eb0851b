"""Analysis helper (not a test): for every 16-row MFMA tile and every group of 3 consecutive kernel offsets (the dx triple of one
(dz, dy)), the span of the source rows the group gathers -- the size of the LDS row window a wave would have to stage -- and
the gather redundancy inside the window.  Input of the gather-GEMM v3 design.  Uses the CPU oracle.
Run: python tests/analysis_tile_window.py"""
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


def stats(name, pair, tm=16, gsz=3):
    kv, n = pair.shape
    nb = n // tm
    p = pair[:, :nb * tm].reshape(kv, nb, tm)
    spans, valid_cnt, act = [], 0, 0
    for g in range(kv // gsz):
        grp = p[gsz * g:gsz * g + gsz]
        valid = grp >= 0
        big = np.where(valid, grp, -1).max(axis=(0, 2))
        small = np.where(valid, grp, 1 << 30).min(axis=(0, 2))
        has = valid.any(axis=(0, 2))
        spans.append(np.where(has, big - small + 1, 0))
        valid_cnt += valid.sum()
        act += valid.any(axis=2).sum()   # (offset, tile) pairs that issue MFMAs
    span = np.stack(spans)
    nz = span[span > 0]
    q = lambda w: 100 * (nz <= w).mean()
    print(f"{name:16s} N={n:6d} tile-groups {nz.size:6d} span med {np.median(nz):4.0f} p90 {np.percentile(nz, 90):5.0f} | <=24 {q(24):5.1f}% "
          f"<=32 {q(32):5.1f}% <=48 {q(48):5.1f}% <=64 {q(64):5.1f}% | gathered/window-rows {valid_cnt / max(nz[nz <= 32].sum(), 1):.2f} "
          f"| active tile-offsets/tile {act / nb:.1f} of {kv}, pairs/row {valid_cnt / n:.1f}")


cur, cs = idx, [81, 1600, 1408]
p1 = sr.subm_rulebook(cur, cs, (3, 3, 3))
stats("s1 subm (touch)", p1)
for st, pad in ((2, (1, 1, 1)), (3, (1, 1, 1)), (4, (0, 1, 1))):
    out = sr.sparse_rulebook(cur, cs, 1, (3, 3, 3), (2, 2, 2), pad)
    stats(f"s{st} down fwd", out[2])
    stats(f"s{st} down bwd", out[3])
    cur, cs = out[0], list(out[1])
    stats(f"s{st} subm", sr.subm_rulebook(cur, cs, (3, 3, 3)))
