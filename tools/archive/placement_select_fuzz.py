#!/usr/bin/env python3
"""Fuzz of the chunk selection of dabgpu_alloc_frame_buffers(DABGPU_PLACE_DOMAINS) -- a Python restatement of the selection
in csrc/dabgpu_placement.hip (samples: the domain that can carry them alone, else the one with the most bytes first, 1 GiB
chunks before 256 MiB ones, filling up with small chunks before overshooting; soft bits piece by piece: the chunk whose
domain the samples read beside that piece are least in) -- over random chunk maps of the bench's request (24 GiB of samples,
3.5 GiB of soft bits, 28 + 53 chunks probed).  A failure = the selection cannot complete inside the pair's address range
and the call falls back to a plain pair.  usage: tools/placement_select_fuzz.py   (no GPU needed)"""
import random
CH=1<<30; CS=1<<28
def select(doms, n_big, iq_bytes, soft_bytes, frame_stride=196608):
    n_total=len(doms); sizes=[CH if i<n_big else CS for i in range(n_total)]
    need_iq=(iq_bytes+CH-1)//CH; need_soft=(soft_bytes+CS-1)//CS
    pair_bytes=need_iq*CH+need_soft*CS+2*CH
    bytes_in=[0,0,0]
    for i in range(n_total): bytes_in[doms[i]]+=sizes[i]
    allb=sum(bytes_in); iq_domain=-1
    for d in range(3):
        if bytes_in[d]>=iq_bytes and allb-bytes_in[d]>=soft_bytes and (iq_domain<0 or bytes_in[d]>bytes_in[iq_domain]): iq_domain=d
    order=sorted(range(3), key=lambda x:(0 if x==iq_domain else 1, -bytes_in[x], x))
    used=[0]*n_total; iq_sel=[]; iq_mapped=0
    for k in range(3):
        if iq_mapped>=iq_bytes: break
        for pas in range(2):
            if iq_mapped>=iq_bytes: break
            for i in range(n_total):
                if iq_mapped>=iq_bytes: break
                if (i<n_big)!=(pas==0) or used[i] or doms[i]!=order[k]: continue
                if pas==0 and iq_bytes-iq_mapped<CH:
                    small_left=sum(CS for j in range(n_big,n_total) if not used[j])
                    if small_left>=iq_bytes-iq_mapped: break
                iq_sel.append(i); used[i]=1; iq_mapped+=sizes[i]
    iq_end=[]; e=0
    for i in iq_sel: e+=sizes[i]; iq_end.append(e)
    def by_dom(lo,hi):
        w=[0.0,0.0,0.0]; begin=0
        for k,i in enumerate(iq_sel):
            a0=max(lo,begin); a1=min(hi,iq_end[k])
            if a1>a0: w[doms[i]]+=a1-a0
            begin=iq_end[k]
        return w
    iq_per_soft=frame_stride*8/230400.0; slack=1.5*CH
    soft_mapped=0; soft_sel=[]; shared=0.0
    while soft_mapped<soft_bytes:
        best=-1;best_cost=0
        for i in range(n_total):
            if used[i]: continue
            sz=sizes[i]
            if iq_mapped+soft_mapped+sz>pair_bytes: continue
            w=by_dom(soft_mapped*iq_per_soft-slack, min(soft_bytes,soft_mapped+sz)*iq_per_soft+slack)
            tot=sum(w); cost=w[doms[i]]/tot if tot>0 else 0
            if best<0 or cost<best_cost-1e-9 or (cost<best_cost+1e-9 and sz<sizes[best]): best=i;best_cost=cost
        if best<0: break
        soft_sel.append(best); used[best]=1
        shared+=best_cost*min(sizes[best], soft_bytes-soft_mapped); soft_mapped+=sizes[best]
    ok = not (iq_mapped<iq_bytes or soft_mapped<soft_bytes or iq_mapped+soft_mapped>pair_bytes)
    return ok, iq_mapped, soft_mapped, pair_bytes, shared/soft_bytes
iq=24*CH; soft=16384*230400
random.seed(1); fails=0
worst=0
for trial in range(20000):
    n_big=28; n_small=53
    # contiguous-ish domain runs like the driver hands out
    doms=[]
    d=random.randrange(3)
    while len(doms)<n_big+n_small:
        run=random.randrange(1,40); doms+= [d]*run; d=random.randrange(3)
    doms=doms[:n_big+n_small]
    ok,a,b,p,sh=select(doms,n_big,iq,soft)
    worst=max(worst,a+b)
    if not ok:
        fails+=1
        if fails<6: print("FAIL", "".join("ABC"[x] for x in doms[:n_big])+"".join("abc"[x] for x in doms[n_big:]), a/CH, b/CH, p/CH)
print("fails", fails, "largest mapped pair GiB", worst/CH)
