"""Analysis helper (not a test): per-row vs per-tile active-offset counts of the VirConv-L rulebooks on one synthetic frame,
in natural row order and after the windowed mask sort of vc_row_order.  Uses the CPU oracle for the rulebooks, hence
lives under tests/.  Run: python tests/analysis_tile_union.py   (numbers quoted in DESIGN.md §4.3)"""
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
shape=[81,1600,1408]
def stats(name, pair):
    kv,n=pair.shape
    act=(pair>=0)
    per_row=act.sum(0).mean()
    def union(g):
        m=n//g*g
        a=act[:,:m].reshape(kv,m//g,g).any(2)
        return a.sum(0).mean()
    print(f"{name}: N={n} per-row {per_row:.2f} union16 {union(16):.2f} union64 {union(64):.2f}")
    # sorted by mask
    key=np.zeros(n,np.int64)
    for k in range(kv): key|= act[k].astype(np.int64)<<k
    order=np.argsort(key,kind='stable')
    a2=act[:,order]
    def union2(g):
        m=n//g*g
        a=a2[:,:m].reshape(kv,m//g,g).any(2)
        return a.sum(0).mean()
    print(f"   mask-sorted: union16 {union2(16):.2f} union64 {union2(64):.2f}  distinct masks {len(np.unique(key))}")
    # sort within windows of 4096 rows (locality preserved)
    for W in (1024,8192):
        o=np.concatenate([s+np.argsort(key[s:s+W],kind='stable') for s in range(0,n,W)])
        a3=act[:,o]; m=n//16*16
        u16=a3[:,:m].reshape(kv,m//16,16).any(2).sum(0).mean()
        m=n//64*64
        u64=a3[:,:m].reshape(kv,m//64,64).any(2).sum(0).mean()
        print(f"   window {W}: union16 {u16:.2f} union64 {u64:.2f}")
cur=idx; cs=shape
p=sr.subm_rulebook(cur,cs,(3,3,3)); stats("s1 subm",p)
for st,pad in ((2,(1,1,1)),(3,(1,1,1)),(4,(0,1,1))):
    out=sr.sparse_rulebook(cur,cs,1,(3,3,3),(2,2,2),pad)
    oi,os_,pf=out[0],out[1],out[2]
    stats(f"s{st} down",pf)
    cur,cs=oi,list(os_)
    p=sr.subm_rulebook(cur,cs,(3,3,3)); stats(f"s{st} subm",p)
