"""Development aid: start time stamps of the covariance-form kernels in launch order, WITHOUT a profiler (rocprofv3
serialises every launch to >= 4.6 us and overstates the small ones).  Needs a library built with -DBESSX_KTRACE:
  make -C bess_amd/csrc ktrace
  BESSX_LIB_PATH=bess_amd/csrc/build_kt/libbessx.so python tools/ktrace.py
Prints start-to-start time per kernel for one 200-candidate path of configs[1] and a sample of the sequence."""
import sys, ctypes, collections, numpy as np
sys.path.insert(0,'.')
from bess_amd import capi, synth
X,y,_,_=synth.make_lm()
L=capi.lib()
cap=1<<16
buf=(ctypes.c_ulonglong*cap)()
names={1:'topk',2:'cgr',3:'cg',4:'cov_d',5:'panel',6:'reduce',7:'compact',8:'continue',9:'publish',10:'fill_list',11:'resume',12:'need',13:'begin',14:'sel_cgr',15:'after sel_cgr (boundary / idle)'}
import time
with capi.Session(X,y) as s:
    s.sequential_path(np.arange(1,201), ic_type=3)
    L.bessx_debug_ktrace(buf,cap,1)
    ph=(ctypes.c_ulonglong*32)(); L.bessx_debug_phase(ph,1)
    t0=time.time(); s.sequential_path(np.arange(1,201), ic_type=3); wall=time.time()-t0
    n=L.bessx_debug_ktrace(buf,cap,1)
    ph=(ctypes.c_ulonglong*32)(); L.bessx_debug_phase(ph,1)
    print("counters", s.counters())
    if ph[30]: print("hinv phases (us per solve, %d solves, %d refinements):"%(ph[30],ph[29]), {k:round(ph[i]*0.01/ph[30],2) for i,k in zip(range(8,15),("load","diff","update","x=Hq","resid","finish","commit"))})
    print("phases (us per call, %d calls):"%ph[31], [round(ph[i]*0.01/max(ph[31],1),2) for i in range(8)])
    for cls,nm in ((0,"repeated set"),(1,"other (solve / fell through)")):
        if ph[14+cls] and not ph[30]: print("k_sel_cgr block 0, %s (%d launches): selection, solve body, tail us:"%(nm,ph[14+cls]), [round(ph[8+3*cls+i]*0.01/ph[14+cls],2) for i in range(3)])
    if ph[28]: print("repeated-set launches (us per call, %d calls): entry, block extremes, decision, commit, snapshot:"%ph[28], [round(ph[i]*0.01/ph[28],2) for i in range(16,21)])
ev=[(buf[i]&255, (buf[i]>>8)*0.01) for i in range(n)]  # us
ev.sort(key=lambda e:e[1])
print("events",n,"wall ms",wall*1e3,"span ms",(ev[-1][1]-ev[0][1])/1e3)
tot=collections.defaultdict(float); cnt=collections.Counter()
for (k,t),(k2,t2) in zip(ev,ev[1:]):
    tot[k]+=t2-t; cnt[k]+=1
for k,v in sorted(tot.items(),key=lambda x:-x[1]): print("%-10s %5d launches  %8.3f ms  avg %6.2f us"%(names.get(k,k),cnt[k],v/1e3,v/cnt[k]))
# time between the end of a k_sel_cgr block 0 and the next kernel's start, by what that launch was
gaps=collections.defaultdict(list)
for i in range(len(ev)-2):
    if ev[i][0]==14 and ev[i+1][0]==15:
        body=ev[i+1][1]-ev[i][1]
        what="fell through" if body<1.5 else ("repeated set" if body<8 else "solve")
        gaps[(what,names.get(ev[i+2][0],ev[i+2][0]))].append(ev[i+2][1]-ev[i+1][1])
for k,v in sorted(gaps.items()): print("after sel_cgr %-14s -> %-8s %4d times  median %5.2f us  mean %5.2f us  sum %.3f ms"%(k[0],k[1],len(v),np.median(v),np.mean(v),np.sum(v)/1e3))
# the selection(+solve) launches by what they turned out to do, told by their duration
for kid in (14, 1, 2):
    d=np.array([t2-t for (k,t),(k2,t2) in zip(ev,ev[1:]) if k==kid])
    if d.size:
        for lo,hi,what in ((0,3,"fell through"),(3,12,"shortcut / commit / snapshot"),(12,1e9,"solve")):
            m=(d>=lo)&(d<hi)
            print("%-8s %-30s %4d launches %7.3f ms  median %6.2f us"%(names.get(kid,kid),what,int(m.sum()),d[m].sum()/1e3,np.median(d[m]) if m.any() else 0))
mid=n//2
for (k,t),(k2,t2) in list(zip(ev,ev[1:]))[mid:mid+24]: print("%-10s %7.2f us"%(names.get(k,k),t2-t))
# durations of the launch that follows a repeated-set selection (the publishing solve kernel): 8-9 us = publish only,
# more = the device waited for the host to queue the next fit
seq=list(zip(ev,ev[1:]))
pub=[]
for i in range(1,len(seq)):
    (k,t),(k2,t2)=seq[i]
    (kp,tp),(_,tpe)=seq[i-1]
    if k==2 and kp==1 and (tpe-tp)<3.0: pub.append(t2-t)
import numpy as np
pub=np.array(pub); print("publishing launches",len(pub),"median %.1f us"%np.median(pub),"sum %.2f ms"%(pub.sum()/1e3), "over 12us:",int((pub>12).sum()), "excess ms %.2f"%((pub[pub>12]-9).sum()/1e3))
