"""CPU model of the culled kNN kernel's pass 1 (csrc/knn.hip) on one 4096-point cloud: per wave of 64 consecutive queries,
32-point candidate tiles with bounding boxes, per-lane sorted top-20 lists, a tile is scanned when any lane's box bound
beats its threshold, a network pass is counted whenever any lane improves on a candidate.  It reproduces the kernel's
KNN_STATS counters (26.6 tiles scanned, ~50 hit batches, ~244 passes per wave on a Hilbert-ordered uniform cloud; 37.8 /
265 on the Z-order) and was used to decide, before touching the kernel, which ideas were worth building: Hilbert vs
Z-order (built), tile visiting orders, a per-lane index-neighbourhood first phase, a wave-level scalar pre-test.
Run: python scripts/sim_knn_order.py   (numpy only, ~2 minutes)."""
import numpy as np
def spread(v):
    v=v.astype(np.uint64)
    v=(v|(v<<16))&0x030000ff; v=(v|(v<<8))&0x0300f00f; v=(v|(v<<4))&0x030c30c3; v=(v|(v<<2))&0x09249249; return v
def morton(q): return spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
def hilbert(q, bits=10):
    X=[q[:,0].astype(np.uint32).copy(), q[:,1].astype(np.uint32).copy(), q[:,2].astype(np.uint32).copy()]
    M=1<<(bits-1)
    Q=M
    while Q>1:
        P=Q-1
        for i in range(3):
            m=(X[i]&Q)!=0
            X[0]=np.where(m, X[0]^P, X[0])
            t=np.where(~m, (X[0]^X[i])&P, 0).astype(np.uint32)
            X[0]^=t; X[i]^=t
        Q>>=1
    for i in range(1,3): X[i]^=X[i-1]
    t=np.zeros_like(X[0]); Q=M
    while Q>1:
        t=np.where((X[2]&Q)!=0, t^(Q-1), t); Q>>=1
    for i in range(3): X[i]^=t
    # interleave: X[0] most significant
    return (spread(X[0])<<2)|(spread(X[1])<<1)|spread(X[2])

import numpy as np, sys
rng=np.random.default_rng(0)
p=rng.uniform(-1,1,(4096,3)).astype(np.float32)
lo=p.min(0); hi=p.max(0); q=np.clip((p-lo)/(hi-lo)*1023,0,1023).astype(np.uint32)
p=p[np.argsort(hilbert(q),kind="stable")]
T=32; nt=128
tlo=p.reshape(nt,T,3).min(1); thi=p.reshape(nt,T,3).max(1)
def run(order_fn, waves=range(0,64,3)):
    tot_scan=tot_pass=tot_hitb=0
    for w in waves:
        Q=p[64*w:64*w+64]
        lists=np.full((64,20),np.inf)
        t0=2*w
        order=order_fn(w,t0,Q)
        scanned=passes=hitb=0
        for c in order:
            thr=lists[:,19]
            dx=np.maximum(np.maximum(tlo[c]-Q, Q-thi[c]),0); lb=(dx**2).sum(1)
            if not (lb<=thr).any(): continue
            scanned+=1
            C=p[32*c:32*c+32]
            D=((Q[:,None,:]-C[None,:,:])**2).sum(-1)   # (64,32)
            for b0 in range(0,32,8):
                if not (D[:,b0:b0+8]<=lists[:,19:20]).any(): continue
                hitb+=1
                for u in range(8):
                    d=D[:,b0+u]; better=d<lists[:,19]
                    if better.any():
                        passes+=1
                        idx=np.where(better)[0]
                        for i in idx:
                            l=lists[i]; l[19]=d[i]; l.sort()
        tot_scan+=scanned; tot_pass+=passes; tot_hitb+=hitb
    n=len(list(waves))
    return tot_scan/n, tot_hitb/n, tot_pass/n
def outward(w,t0,Q):
    o=[t0,t0+1]
    for d in range(1,nt):
        if t0-d>=0: o.append(t0-d)
        if t0+1+d<nt: o.append(t0+1+d)
    return o
def boxdist(w,t0,Q):
    ql=Q.min(0); qh=Q.max(0)
    dx=np.maximum(np.maximum(tlo-qh, ql-thi),0); lb=(dx**2).sum(1)
    key=lb.copy(); key[t0]=-2; key[t0+1]=-1
    return list(np.argsort(key,kind="stable"))
def centerdist(w,t0,Q):
    qc=Q.mean(0); tc=(tlo+thi)/2
    key=((tc-qc)**2).sum(1); key[t0]=-2; key[t0+1]=-1
    return list(np.argsort(key,kind="stable"))
for name,fn in (("index outward",outward),("box distance",boxdist),("centre distance",centerdist)):
    print("%-16s tiles scanned %.1f, hit batches %.1f, network passes %.1f"%((name,)+run(fn)))

def run2(waves=range(0,64,3)):
    res={"slot":0,"batch_max":0,"tile_max":0,"lane_ins":0,"batch_sorted":0}
    for w in waves:
        Q=p[64*w:64*w+64]; lists=np.full((64,20),np.inf); t0=2*w
        for c in outward(w,t0,Q):
            thr=lists[:,19]
            dx=np.maximum(np.maximum(tlo[c]-Q, Q-thi[c]),0); lb=(dx**2).sum(1)
            if not (lb<=thr).any(): continue
            C=p[32*c:32*c+32]; D=((Q[:,None,:]-C[None,:,:])**2).sum(-1)
            tile_hits=np.zeros(64,int)
            for b0 in range(0,32,8):
                if not (D[:,b0:b0+8]<=lists[:,19:20]).any(): continue
                bh=np.zeros(64,int)
                for u in range(8):
                    d=D[:,b0+u]; better=d<lists[:,19]
                    if better.any():
                        res["slot"]+=1
                        for i in np.where(better)[0]:
                            l=lists[i]; l[19]=d[i]; l.sort(); bh[i]+=1; res["lane_ins"]+=1
                res["batch_max"]+=bh.max(); tile_hits+=bh
            res["tile_max"]+=tile_hits.max()
    n=len(list(waves))
    return {k:v/n for k,v in res.items()}
print(run2())

def run3(W, waves=range(0,64,3)):
    tot_pass=tot_scan=tot_ins=0; passesA=0
    n=len(p)
    for w in waves:
        base=64*w; Q=p[base:base+64]; lists=np.full((64,20),np.inf); t0=2*w
        ii=np.arange(base,base+64)
        offs=[0]
        for s in range(1,W+1): offs+= [s,-s]
        for off in offs:
            j=ii+off; ok=(j>=0)&(j<n)
            d=np.where(ok, ((Q-p[np.clip(j,0,n-1)])**2).sum(1), np.inf)
            better=d<lists[:,19]
            if better.any():
                tot_pass+=1; passesA+=1
                for i in np.where(better)[0]:
                    l=lists[i]; l[19]=d[i]; l.sort(); tot_ins+=1
        for c in outward(w,t0,Q):
            thr=lists[:,19]
            dx=np.maximum(np.maximum(tlo[c]-Q, Q-thi[c]),0); lb=(dx**2).sum(1)
            if not (lb<=thr).any(): continue
            C=p[32*c:32*c+32]; D=((Q[:,None,:]-C[None,:,:])**2).sum(-1)
            jj=np.arange(32*c,32*c+32)
            seen=np.abs(jj[None,:]-ii[:,None])<=W
            D=np.where(seen,np.inf,D)
            if not (D<=lists[:,19:20]).any(): continue
            tot_scan+=1
            for b0 in range(0,32,8):
                if not (D[:,b0:b0+8]<=lists[:,19:20]).any(): continue
                for u in range(8):
                    d=D[:,b0+u]; better=d<lists[:,19]
                    if better.any():
                        tot_pass+=1
                        for i in np.where(better)[0]:
                            l=lists[i]; l[19]=d[i]; l.sort(); tot_ins+=1
    n_=len(list(waves))
    return tot_scan/n_, tot_pass/n_, passesA/n_, tot_ins/n_/64
for W in (16,32,48,64):
    print("window +-%d: tiles scanned after %.1f, passes %.1f (window phase %.1f), insertions per lane %.1f"%((W,)+run3(W)))

def run4(waves=range(0,64,3)):
    tests_lane=tests_skipped=0
    for w in waves:
        base=64*w; Q=p[base:base+64]; lists=np.full((64,20),np.inf); t0=2*w
        ql=Q.min(0); qh=Q.max(0)
        dxw=np.maximum(np.maximum(tlo-qh, ql-thi),0); wlb=(dxw**2).sum(1)
        for c in outward(w,t0,Q):
            thr=lists[:,19]
            if wlb[c]>thr.max():
                tests_skipped+=1; continue
            tests_lane+=1
            dx=np.maximum(np.maximum(tlo[c]-Q, Q-thi[c]),0); lb=(dx**2).sum(1)
            if not (lb<=thr).any(): continue
            C=p[32*c:32*c+32]; D=((Q[:,None,:]-C[None,:,:])**2).sum(-1)
            for b0 in range(0,32,8):
                if not (D[:,b0:b0+8]<=lists[:,19:20]).any(): continue
                for u in range(8):
                    d=D[:,b0+u]; better=d<lists[:,19]
                    for i in np.where(better)[0]:
                        l=lists[i]; l[19]=d[i]; l.sort()
    n_=len(list(waves))
    return tests_lane/n_, tests_skipped/n_
print("per-lane box tests %.1f, skipped by the wave-box scalar test %.1f"%run4())
