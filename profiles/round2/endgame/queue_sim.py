import heapq, numpy as np, sys
hint=np.load('/root/repo/gpurun_out/r2y/spans_lpt_hint.npy')
ctrl=np.load('/root/repo/gpurun_out/r2y/spans_lpt_ctrl.npy').view(np.uint32)
hl=np.load('/root/repo/gpurun_out/r2y/spans_lpt_hitlist.npy')
N=262144
cls_n=ctrl[11:15]
print("class counts", cls_n, "nhit", ctrl[1])
rays=np.concatenate([hl[k*N:k*N+cls_n[k]] for k in range(4)])
L=hint[rays].astype(int)      # per-ray phase-0 samples
print("rays", len(rays), "mean len %.2f"%L.mean(), "hist", np.bincount(L, minlength=17))
length=dict(zip(rays.tolist(), L.tolist()))

def simulate(order, tau_old=90.4, tau_young=105.6, tau_alone=62.0, pool_cap=128, fixed=0.25, seed=0, n_wg=512, max_n=8, speed_jitter=0.0, verbose=False):
    """order: list of ray lengths in queue order.  Returns per-WG end times, rounds, samples."""
    rng=np.random.default_rng(seed)
    q=list(order); qh=0; nq=len(q)
    pools=[[] for _ in range(n_wg)]    # remaining lengths
    done=[False]*n_wg
    end=np.zeros(n_wg); rounds=np.zeros(n_wg,int); samples=np.zeros(n_wg,int)
    jit=1+speed_jitter*rng.standard_normal(n_wg)
    heap=[(0.0+1e-6*i,i) for i in range(n_wg)]
    heapq.heapify(heap)
    while heap:
        t,i=heapq.heappop(heap)
        p=pools[i]
        # refill
        want=pool_cap-len(p)
        if want>0 and qh<nq:
            take=min(want,nq-qh)
            p.extend(q[qh:qh+take]); qh+=take
        if not p:
            done[i]=True; end[i]=t; continue
        npool=len(p)
        n=min(max_n,128//npool); extra=128-n*npool if n<max_n else 0
        Mv=0; newp=[]
        for r,rem in enumerate(p):
            mine=n+(1 if r<extra else 0)
            # the ray has `rem` samples left before it dies; evaluated this round = min(mine, ...) : T-death wastes the rest of its request
            if rem<=mine:
                Mv+=mine if rem<16 else rem   # pessimistic: samples after the death are still evaluated (unless budget end)
            else:
                Mv+=mine; newp.append(rem-mine)
        Mv=min(Mv,128)
        pools[i]=newp
        nt=(Mv+31)//32
        partner=(i+n_wg//2)%n_wg
        tau=tau_alone if done[partner] else (tau_old if i<n_wg//2 else tau_young)
        dur=tau*(fixed+(1-fixed)*nt/4)*jit[i]
        rounds[i]+=1; samples[i]+=Mv
        heapq.heappush(heap,(t+dur,i))
    return end,rounds,samples

def report(tag,res):
    end,rounds,samples=res
    print("%-34s phase %.0f us | end mean %.0f p10 %.0f p50 %.0f p90 %.0f | rounds %d..%d mean %.1f | samples tot %d | balanced %.1f%%"%(tag,end.max(),end.mean(),*np.percentile(end,[10,50,90]),rounds.min(),rounds.max(),rounds.mean(),samples.sum(),100*end.mean()/end.max()))

if __name__=="__main__":
    rng=np.random.default_rng(1)
    # base order: raster order in chunks of 64-pixel waves -> sort rays by index
    base=[length[r] for r in sorted(rays.tolist())]
    report("base (raster)",simulate(base))
    perm=list(rng.permutation(base))
    report("random permutation",simulate(perm))
    lpt=sorted(base,reverse=True)
    report("LPT exact",simulate(lpt))
    lpt4=[length[r] for r in rays.tolist()]
    report("LPT 4 classes (as run)",simulate(lpt4))
    report("base, equal speeds",simulate(base,tau_old=98,tau_young=98))
    report("perm, equal speeds",simulate(perm,tau_old=98,tau_young=98))
