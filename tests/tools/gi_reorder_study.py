#!/usr/bin/env python3
"""Offline study (CPU, oracle only; round 4): what could re-ordering the four GI rays a lane owns in a whole-spp work unit buy?
A lane's samples may run in any order (the sums are integers, the RNG is keyed by the sample), so stage k could take every
lane's k-th longest GI ray -- IF the lengths were known before walking.  This prices the upper bounds on the oracle's traces of
the C3 frame (rounds per tile-sample; the shipped order: GI 20.0, GI shadow 24.8 of 110 in all):
    sorted by the GI ray's true length            GI 16.5  GI shadow 24.3     (-3.6 % of all rounds, with a PERFECT predictor)
    both stages sorted independently              GI 16.5  GI shadow 21.8     (-5.9 %; needs the four GI hits parked)
    the four rays chained in the lane, no stages  GI 15.0  GI shadow 20.5     (-8.4 % gross; needs refill: DESIGN.md section 9)
A realisable predictor (the direction's components) gets a fraction of the first line.  ANALYSIS INFRASTRUCTURE: oracle/ only."""
import sys, os, ctypes as C, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tests','tools'))
import tile_shape as T
import cpuvoxelraycaster_amd as vrc, oracle_lib as O
depth,W,Hh,spp=9,1920,1080,4
nodes=vrc.build_terrain_lsvo(O.load_terrain_heights(),depth); tex=O.load_textures(); L=O.lib()
L.orc_render_frame_optrace_geo.argtypes=[C.c_void_p,C.c_uint32,C.c_void_p,C.c_void_p,C.POINTER(O.Camera),C.POINTER(O.FrameParams),C.c_int,C.c_void_p,C.c_void_p,C.c_uint32,C.c_void_p]
tr,op,sk=T.trace_pose(L,O,vrc,nodes,tex,depth,W,Hh,-0.5,0.0,32,8,spp)
ln=np.maximum(tr-np.minimum(sk,tr),0)          # (spp, nb, 8, W, 4 stages) executed iterations
S,nb,_,_,_=ln.shape
# 8x8 tiles: lanes
def tiles(a):  # (spp, nb, 8, W) -> (spp, ntiles, 64)
    a=a.reshape(S,nb,8,W//8,8); a=np.moveaxis(a,3,2)  # spp, nb, W/8, 8, 8
    return a.reshape(S,-1,64)
gi=tiles(ln[...,2]); gs=tiles(ln[...,3])
base_gi=gi.max(axis=2).sum(); base_gs=gs.max(axis=2).sum()
print("baseline rounds per tile-sample: GI %.2f  GIshadow %.2f"%(base_gi/gi.shape[1]/S, base_gs/gs.shape[1]/S))
# perfect within-lane reordering of the GI ray (and its shadow ray follows): sort each lane's 4 samples by GI length
order=np.argsort(-gi,axis=0)
gi_s=np.take_along_axis(gi,order,axis=0); gs_s=np.take_along_axis(gs,order,axis=0)
print("sorted by GI length:           GI %.2f  GIshadow %.2f"%(gi_s.max(axis=2).sum()/gi.shape[1]/S, gs_s.max(axis=2).sum()/gs.shape[1]/S))
# sort by pair total
order=np.argsort(-(gi+gs),axis=0)
gi_p=np.take_along_axis(gi,order,axis=0); gs_p=np.take_along_axis(gs,order,axis=0)
print("sorted by GI+GIshadow:         GI %.2f  GIshadow %.2f"%(gi_p.max(axis=2).sum()/gi.shape[1]/S, gs_p.max(axis=2).sum()/gs.shape[1]/S))
# independent sorting of both stages (needs decoupling shadow from its GI in time: all 4 GI first, then 4 shadows)
gs_i=-np.sort(-gs,axis=0)
print("GI sorted, shadows sorted independently: GI %.2f GIshadow %.2f"%(gi_s.max(axis=2).sum()/gi.shape[1]/S, gs_i.max(axis=2).sum()/gs.shape[1]/S))
# chaining all 4 GI rays back to back in a lane (no stage sync between samples): max over lanes of the SUM
print("chained in a lane (sum then max): GI %.2f  GIshadow %.2f (per sample)"%(gi.sum(axis=0).max(axis=1).sum()/gi.shape[1]/S, gs.sum(axis=0).max(axis=1).sum()/gs.shape[1]/S))
print("mean lane length: GI %.2f GIshadow %.2f; zero-length frac GI %.3f"%(gi.mean(), gs.mean(), (gi==0).mean()))
import collections
print("GI length percentiles", np.percentile(gi,[50,75,90,95,99,100]))
