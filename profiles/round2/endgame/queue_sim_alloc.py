import heapq, numpy as np, sys
sys.path.insert(0,'/tmp/sim')
from sim import length, rays, report, hl, cls_n, N
def simulate(order, want_fn, tau_old=90.4, tau_young=105.6, tau_alone=62.0, fixed=0.25, n_wg=512, jitter=0.06, seed=0, max_rays=128):
    """order: list of (L, hint).  want_fn(hint, done) -> samples this ray asks per round.  Refill while sum(want) < 128 and rays < max_rays."""
    rng=np.random.default_rng(seed)
    q=list(order); qh=0; nq=len(q)
    pools=[[] for _ in range(n_wg)]
    done=[False]*n_wg
    end=np.zeros(n_wg); rounds=np.zeros(n_wg,int); samples=np.zeros(n_wg,int)
    sp=1+jitter*rng.standard_normal(n_wg)
    heap=[(1e-6*i,i) for i in range(n_wg)]
    heapq.heapify(heap)
    tq=None
    while heap:
        t,i=heapq.heappop(heap)
        p=pools[i]
        tot=sum(want_fn(e[3],e[1]) for e in p)
        while qh<nq and len(p)<max_rays:
            l,h=q[qh]; w=want_fn(h,0)
            if tot+w>128: break
            p.append([l,0,l,h]); tot+=w; qh+=1
        if qh>=nq and tq is None: tq=t
        if not p:
            done[i]=True; end[i]=t; continue
        req=[want_fn(e[3],e[1]) for e in p]
        left=128-sum(req)
        # spare slots: one more for everyone (round-robin from the oldest) up to 8 per ray
        if left>0:
            order_=sorted(range(len(p)), key=lambda r:-p[r][1])
            while left>0:
                prog=False
                for r in order_:
                    if left<=0: break
                    if req[r]<8: req[r]+=1; left-=1; prog=True
                if not prog: break
        Mv=0; newp=[]
        for e,mine in zip(p,req):
            rem=e[0]
            if rem<=mine:
                Mv+=mine if e[2]<16 else rem
            else:
                Mv+=mine; e[0]=rem-mine; e[1]+=mine; newp.append(e)
        pools[i]=newp
        nt=(min(Mv,128)+31)//32
        partner=(i+n_wg//2)%n_wg
        tau=tau_alone if done[partner] else (tau_old if i<n_wg//2 else tau_young)
        dur=tau*(fixed+(1-fixed)*nt/4)*sp[i]
        rounds[i]+=1; samples[i]+=Mv
        heapq.heappush(heap,(t+dur,i))
    return (end,rounds,samples),tq
if __name__=="__main__":
    rng=np.random.default_rng(1)
    rr=sorted(rays.tolist())
    base=[(length[r],length[r]) for r in rr]
    lpt4=[(length[r],length[r]) for r in np.concatenate([hl[k*N:k*N+cls_n[k]] for k in range(4)]).tolist()]
    perm=[base[i] for i in rng.permutation(len(base))]
    W=lambda thr,b: (lambda h,d: b if h>=thr else 1)
    for name,order in (("raster",base),("lpt4",lpt4),("perm",perm)):
        for thr,b in ((99,1),(13,2),(9,2),(9,3),(5,2),(13,4),(9,4)):
            res,tq=simulate(order, W(thr,b))
            report("%s thr=%d boost=%d tq=%.0f"%(name,thr,b,tq), res)
