"""Offline experiment behind cfnmpc_opts.as_warm's default (profiles/r05_as_warm.md): on the closed-loop QP sequence of the CPU restatement
(staggered kicks as in bench.py), how many active-set solves does each constrained QP of a vehicle in its 2nd+ constrained step need
under different STARTING classifications?  cold = today's violations of the unconstrained minimiser; union = ... plus the previous
step's final set; near10 / near2 = ... plus previous-set members whose unconstrained value lies within 10 % / 2 % of the box width
of that bound; prevmult = the previous set alone.  Test infrastructure (imports oracle/); usage: python tools/as_warm_rules.py [B] [kick scale]."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np, cfnmpc_oracle as o, cref
N=50; B=int(sys.argv[1]) if len(sys.argv)>1 else 160; STEPS=40; KICK=20
scale=float(sys.argv[2]) if len(sys.argv)>2 else 1.0
rng=np.random.default_rng(5)
opts=cref.default_opts(N=N, active_set=1)
yr,ye=o.regulation_yref(N,(0,0,0.4)); yref=np.repeat(yr[None],B,0).copy(); yref_e=np.repeat(ye[None],B,0).copy()
x=o.sample_hover_x0(rng,B,scale=scale)
xit=np.repeat(x[:,None,:],N+1,1).copy(); uit=np.full((B,N,4),o.HOV_W)
cohort=(B+KICK-1)//KICK
kicks=o.sample_hover_x0(rng,cohort*KICK,scale=scale).reshape(KICK,cohort,13)

def pdas(H,h,lb,ub,lo,up,max_solves=12):
    solves=0
    while solves<max_solves:
        solves+=1
        act=lo|up; free=~act
        v=np.where(lo,lb,np.where(up,ub,0.0))
        if free.any():
            v[free]=np.linalg.solve(H[np.ix_(free,free)], -h[free]-H[np.ix_(free,act)]@v[act])
        grad=H@v+h
        lo2=(free&(v<lb))|(lo&(grad>0)); up2=(free&(v>ub))|(up&(grad<0))
        if np.array_equal(lo2,lo) and np.array_equal(up2,up): return solves,lo,up,v
        lo,up=lo2,up2
    return 99,lo,up,v

rules=["cold","union","near10","near2","prevmult"]
hist={r:[] for r in rules}
prev=[None]*B   # (lo,up,consec)
t0=time.time()
for t in range(STEPS):
    c0=(t%KICK)*cohort; c1=min(c0+cohort,B)
    if c1>c0: x[c0:c1]=kicks[t%KICK,:c1-c0]
    xb,ub_=xit.copy(),uit.copy()
    st,it,_,_=cref.rti_step(opts,xit,uit,x.copy(),yref,yref_e,nthreads=0)
    for i in np.nonzero(it>0)[0]:
        A,Bm,b,q,r=cref.linearise(opts,xb[i],ub_[i],x[i].copy(),yref[i],yref_e[i])
        qp=o.StageQP(N); qp.A,qp.B,qp.b,qp.q,qp.r=A,Bm,b,q,r
        qp.lb=opts.u_min-ub_[i]; qp.ub=opts.u_max-ub_[i]; qp.dx0=x[i]-xb[i,0]
        H,h,_,_=o.condense(qp)
        lb=qp.lb.reshape(-1); ub=qp.ub.reshape(-1)
        v0=np.linalg.solve(H,-h)
        lo0=v0<lb; up0=v0>ub
        w=ub-lb
        res={}
        s,lo,up,v=pdas(H,h,lb,ub,lo0.copy(),up0.copy()); res["cold"]=s
        final=(lo,up)
        if prev[i] is not None:
            plo,pup=prev[i]
            ins=~(lo0|up0)
            res["union"]=pdas(H,h,lb,ub,lo0|(ins&plo),up0|(ins&pup))[0]
            for nm,fr in (("near10",0.10),("near2",0.02)):
                res[nm]=pdas(H,h,lb,ub,lo0|(ins&plo&(v0-lb<fr*w)),up0|(ins&pup&(ub-v0<fr*w)))[0]
            # prevmult: previous set exactly (no today's violations)
            res["prevmult"]=pdas(H,h,lb,ub,plo.copy(),pup.copy())[0]
            for r_ in rules: hist[r_].append(res[r_])
        prev[i]=final
    for i in np.nonzero(it==0)[0]: prev[i]=None
    x=np.stack([o.rk4(x[i],uit[i,0]) for i in range(B)]) if False else cref.sim(x,uit[:,0].copy(),0.015,1)
print("B",B,"scale",scale,"warm-eligible constrained QPs:",len(hist["cold"]),"time %.0fs"%(time.time()-t0))
for r_ in rules:
    a=np.array(hist[r_]); print(f"{r_:9s} mean {a.mean():.3f}  hist",np.bincount(np.minimum(a,13))[1:])
