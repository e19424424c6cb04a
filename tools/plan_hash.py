"""A refactoring net for the host planner (no GPU): SHA-1 of the plan block (every array the device reads) for the cfg2 - cfg5 parts over a 2 M-document
segment in both codecs, planned on 1 and on 8 host threads.  Run it before and after a change that must not alter a plan and diff the outputs:
a byte-identical block means byte-identical launches.    usage: python tools/plan_hash.py > before.txt; ...; python tools/plan_hash.py | diff before.txt -"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, hashlib
import trinity_amd as T
from trinity_amd import hostplan as HP, workloads as W
D,V=2_000_000,200_000
out=[]
for codec in (1,2):
    seg=T.Segment(D,V,10,42,codec=codec); hi=HP.HostIndex.from_segment(seg)
    for wl in ("cfg2","cfg3","cfg4","cfg5"):
        parts,_=W.build_parts(wl,D,V,10,42,4096)
        for pt in parts:
            if pt.codec!=codec: continue
            for thr in (1,8):
                p=HP.HostPlan(hi,pt.programs,pt.flags,pt.topk,threads=thr)
                out.append((wl,pt.name[:12],codec,thr,hashlib.sha1(bytes(p.block)).hexdigest()[:16], len(p.block)))
                p.close()
for o in out: print(*o)
