#!/usr/bin/env python3
"""Wave-scheduling lab (CPU, no GPU time): replays the per-ray step sequences of the wide-BVH traversal (node step = N, triangle
test = T, from the host twin phip_debug_host_trace_wide) on simulated 64-lane waves with refill, under different loop
policies, and counts how often the node block and the triangle block execute (each execution costs its vector-memory
instructions whatever the number of active lanes: the ray kernels are bound by the CU's texture-data path).

    python tools/wave_sim.py [atrium|glass_room] [n_rays]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba_amd import _ffi, _abi as A, scene as S


def sequences(scene, n, seed=7):
    phip = _ffi.lib(); gauss = _ffi.gaussian_filter(0.5)
    desc = getattr(S, scene)(64, 36, gauss).desc()
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()
    rng = np.random.default_rng(seed)
    lo, hi = P.min(axis=0), P.max(axis=0)
    # incoherent rays: origins on surfaces would be better; uniform origins inside the room with uniform directions are close enough
    o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32); rays[:, :3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
    hits = np.zeros((n, 4), np.float32); info = A.phip_accel_info()
    stride = 256
    seq = np.zeros((n, stride), np.uint8)
    rc = phip.phip_debug_host_trace_wide(P.ctypes.data_as(C.POINTER(C.c_float)), len(P), T.ctypes.data_as(C.POINTER(C.c_uint32)), len(T),
                                         rays.ctypes.data_as(C.POINTER(A.phip_ray)), n, hits.ctypes.data_as(C.POINTER(A.phip_hit)), 1, C.byref(info),
                                         seq.ctypes.data_as(C.POINTER(C.c_uint8)), stride)
    assert rc == 0
    return [s[:np.argmax(s == 0)] if (s == 0).any() else s for s in seq], info


def simulate(seqs, policy, refill=16, lanes=64, cost_n=5, cost_t=4):
    """policy(cn, ct, idle) -> (run_node_block, run_tri_block).  A lane advances one step per executed block that matches its
    next step (node block first, so N followed by T can both happen in one iteration, as in the kernel)."""
    it = iter(seqs)
    cur = [None] * lanes; pos = [0] * lanes
    nexec = texec = nlane = tlane = iters = 0
    more = True
    while True:
        idle = [i for i in range(lanes) if cur[i] is None]
        if more and (len(idle) >= refill or len(idle) == lanes):
            for i in idle:
                s = next(it, None)
                if s is None: more = False; break
                cur[i] = s; pos[i] = 0
        active = [i for i in range(lanes) if cur[i] is not None]
        if not active:
            if not more: break
            continue
        cn = sum(1 for i in active if pos[i] < len(cur[i]) and cur[i][pos[i]] == 1)
        ct = sum(1 for i in active if pos[i] < len(cur[i]) and cur[i][pos[i]] == 2)
        rn, rt = policy(cn, ct, lanes - len(active))
        iters += 1
        if rn and cn:
            nexec += 1; nlane += cn
            for i in active:
                if pos[i] < len(cur[i]) and cur[i][pos[i]] == 1: pos[i] += 1
        if rt:
            ct2 = sum(1 for i in active if pos[i] < len(cur[i]) and cur[i][pos[i]] == 2)
            if ct2:
                texec += 1; tlane += ct2
                for i in active:
                    if pos[i] < len(cur[i]) and cur[i][pos[i]] == 2: pos[i] += 1
        for i in active:
            if pos[i] >= len(cur[i]): cur[i] = None
    mem = nexec * cost_n + texec * cost_t
    return {"iters": iters, "node_blocks": nexec, "tri_blocks": texec, "node_lanes/blk": round(nlane / max(nexec, 1), 1), "tri_lanes/blk": round(tlane / max(texec, 1), 1),
            "mem_instr_per_ray": round(mem / len(seqs), 2)}


# ---- round 4: the triangle tests of an iteration DEALT over the wave (k_wide.h: WIDE_DEAL).  A lane's run of consecutive triangle steps (the records of the
#      leaves its last node step hit) is decided in one round; a round costs ceil(pairs / 64) executions of the triangle block.  simulate_dealt_thr holds a
#      round back until `thr` pairs are pending or no lane has a node step to take. ----
def simulate_dealt(seqs, refill=16, lanes=64, cost_n=5, cost_t=3):
    it = iter(seqs); cur=[None]*lanes; pos=[0]*lanes
    nexec=texec=nlane=tlane=iters=0; more=True
    while True:
        idle=[i for i in range(lanes) if cur[i] is None]
        if more and (len(idle)>=refill or len(idle)==lanes):
            for i in idle:
                s=next(it,None)
                if s is None: more=False; break
                cur[i]=s; pos[i]=0
        active=[i for i in range(lanes) if cur[i] is not None]
        if not active:
            if not more: break
            continue
        iters+=1
        cn=[i for i in active if pos[i]<len(cur[i]) and cur[i][pos[i]]==1]
        if cn:
            nexec+=1; nlane+=len(cn)
            for i in cn: pos[i]+=1
        tot=0
        for i in active:
            r=0
            while pos[i]+r<len(cur[i]) and cur[i][pos[i]+r]==2: r+=1
            pos[i]+=r; tot+=r
        if tot:
            b=-(-tot//lanes); texec+=b; tlane+=tot
        for i in active:
            if pos[i]>=len(cur[i]): cur[i]=None
    mem=nexec*cost_n+texec*cost_t
    return {"iters":iters,"node_blocks":nexec,"tri_blocks":texec,"node_lanes/blk":round(nlane/max(nexec,1),1),"tri_lanes/blk":round(tlane/max(texec,1),1),"mem_instr_per_ray":round(mem/len(seqs),2)}


def simulate_dealt_thr(seqs, thr, refill=16, lanes=64, cost_n=5, cost_t=3):
    it = iter(seqs); cur=[None]*lanes; pos=[0]*lanes
    nexec=texec=nlane=tlane=iters=0; more=True
    while True:
        idle=[i for i in range(lanes) if cur[i] is None]
        if more and (len(idle)>=refill or len(idle)==lanes):
            for i in idle:
                s=next(it,None)
                if s is None: more=False; break
                cur[i]=s; pos[i]=0
        active=[i for i in range(lanes) if cur[i] is not None]
        if not active:
            if not more: break
            continue
        iters+=1
        cn=[i for i in active if pos[i]<len(cur[i]) and cur[i][pos[i]]==1]
        if cn:
            nexec+=1; nlane+=len(cn)
            for i in cn: pos[i]+=1
        runs={}
        tot=0
        for i in active:
            r=0
            while pos[i]+r<len(cur[i]) and cur[i][pos[i]+r]==2: r+=1
            if r: runs[i]=r; tot+=r
        nodework=sum(1 for i in active if i not in runs and pos[i]<len(cur[i]))
        if tot and (tot>=thr or nodework==0):
            for i,r in runs.items(): pos[i]+=r
            b=-(-tot//lanes); texec+=b; tlane+=tot
        for i in active:
            if pos[i]>=len(cur[i]): cur[i]=None
    mem=nexec*cost_n+texec*cost_t
    return {"iters":iters,"node_blocks":nexec,"tri_blocks":texec,"node_lanes/blk":round(nlane/max(nexec,1),1),"tri_lanes/blk":round(tlane/max(texec,1),1),"mem_instr_per_ray":round(mem/len(seqs),2)}

if __name__ == "__main__":
    scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seqs, info = sequences(scene, n)
    L = np.array([len(s) for s in seqs]); N = np.array([(s == 1).sum() for s in seqs]); T = np.array([(s == 2).sum() for s in seqs])
    print(scene, info.as_dict()); print("steps/ray %.1f (N %.1f, T %.1f)" % (L.mean(), N.mean(), T.mean()))
    print("ideal (64 lanes/block): %.2f mem instr per ray" % ((N.mean() * 5 + T.mean() * 4) / 64))
    pols = {
        "both every iteration (shipping)": lambda cn, ct, idle: (True, True),
        "tri block only if >= 16 lanes want it (or no node work)": lambda cn, ct, idle: (True, ct >= 16 or cn == 0),
        "tri block only if >= 24 lanes": lambda cn, ct, idle: (True, ct >= 24 or cn == 0),
        "tri block only if >= 32 lanes": lambda cn, ct, idle: (True, ct >= 32 or cn == 0),
        "majority vote (one block per iteration)": lambda cn, ct, idle: (cn >= ct, ct > cn),
        "node if >= 24 else tri; tri if >= 24": lambda cn, ct, idle: (cn >= 24 or ct == 0, ct >= 24 or cn < 24),
    }
    for name, pol in pols.items():
        for refill in (16, 32):
            print("%-58s refill %2d: %s" % (name, refill, simulate(seqs, pol, refill=refill)))
    print("-- triangle tests dealt over the wave (round 4)")
    print("dealt, a round whenever pairs are pending: %s" % simulate_dealt(seqs))
    for thr in (16, 32, 48, 64):
        print("dealt, rounds held back until %2d pairs: %s" % (thr, simulate_dealt_thr(seqs, thr)))
    print("-- refill threshold sweep (both blocks every iteration)")
    for refill in (1, 4, 8, 16, 24):
        print("refill %2d: %s" % (refill, simulate(seqs, pols["both every iteration (shipping)"], refill=refill)))
