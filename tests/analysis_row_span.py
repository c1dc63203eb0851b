"""Analysis helper (not a test): for every 64-row block and every (dz, dy) group of kernel offsets, the span of source rows the
three dx-offsets gather, and how often a row is gathered more than once.  Basis of the round-2 plan in DESIGN.md §4.2 (staging
one contiguous row window per group in LDS instead of three gathers).  Uses the CPU oracle, hence lives under tests/.
Run: python tests/analysis_row_span.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virconv_amd import synth, data
from oracle import sparse_ref as sr, geometry as geo
fr = synth.make_frame(0)
pts = data.prepare_frame(fr["points_lidar"], fr["points_virtual"], training=True, rng=np.random.default_rng(10000))
vox, coords, num = geo.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
idx = np.concatenate([np.zeros((len(coords),1),np.int32), coords.astype(np.int32)],1)
# sort ascending (stage-1 rows are in first-touch order, not sorted!)
def stats(name, pair, tm=64):
    kv,n=pair.shape
    nb=n//tm
    p=pair[:,:nb*tm].reshape(kv,nb,tm)
    res=[]
    tot_rows=0; tot_valid=0; uniq=0
    for g in range(9):
        grp=p[3*g:3*g+3]                      # 3 dx offsets of one (dz,dy)
        valid=grp>=0
        big=np.where(valid,grp,-1).max(axis=(0,2))
        small=np.where(valid,grp,1<<30).min(axis=(0,2))
        has=valid.any(axis=(0,2))
        span=np.where(has,big-small+1,0)
        res.append(span)
        tot_valid+=valid.sum()
        # unique rows per block-group
        for b in range(0,nb,max(1,nb//200)):
            v=grp[:,b,:][valid[:,b,:]]
            uniq+=len(np.unique(v)); tot_rows+=len(v)
    span=np.stack(res)
    nz=span[span>0]
    print(f"{name}: N={n} groups {nz.size}  span median {np.median(nz):.0f} p90 {np.percentile(nz,90):.0f} p99 {np.percentile(nz,99):.0f}  <=96: {100*(nz<=96).mean():.1f}%  <=128: {100*(nz<=128).mean():.1f}% ; gathered rows/unique rows (sampled) {tot_rows/max(uniq,1):.2f}")
cur=idx; cs=[81,1600,1408]
for st,pad in ((2,(1,1,1)),(3,(1,1,1)),(4,(0,1,1))):
    out=sr.sparse_rulebook(cur,cs,1,(3,3,3),(2,2,2),pad)
    cur,cs=out[0],list(out[1])
    stats(f"s{st} down fwd",out[2]); stats(f"s{st} down bwd",out[3])
    p=sr.subm_rulebook(cur,cs,(3,3,3)); stats(f"s{st} subm",p)
