"""Estimate of what og_group_voices buys on the synthetic note plans: for waves of 64 consecutive voice slots, the share of
8-frame chunks each envelope wave of the four-wave fm kernel spends in the release-free / release / checked body, voices in
voice order against voices ordered by first note-off; cost weights = static VALU counts of the bodies (scripts/isa_blocks.py).
A model of instruction counts on a sample of 8 192 voices, not a measurement.  python scripts/group_model.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import oscen_amd
V=65536; SR=48000
def model(span_frames, fold, order_key=None):
    plans=oscen_amd.note_plans(V, span=span_frames if span_frames<48000 else 0, fold=fold)
    ev_v,ev_f,ev_x=plans["events"]
    T=span_frames
    keep=ev_f<T; ev_v,ev_f,ev_x=ev_v[keep],ev_f[keep],ev_x[keep]
    nch=T//8
    # per voice, per env: stage timeline -> arrays per chunk: releasing[v,ch] (bool), boundary[v,ch] (stage end or event)
    envs={'w0':[(480,4800,14400)],'w1':[(480,4800,14400)],'w2':[(480,9600,24000),(480,9600,14400)]}
    order=np.arange(V)
    if order_key is not None:
        order=order_key(plans,ev_v,ev_f,ev_x,T)
    res={}
    # events per voice sorted
    idx=np.argsort(ev_v,kind='stable')
    ev_v,ev_f,ev_x=ev_v[idx],ev_f[idx],ev_x[idx]
    starts=np.searchsorted(ev_v,np.arange(V)); ends=np.searchsorted(ev_v,np.arange(V),side='right')
    for w,es in envs.items():
        rel=np.zeros((V,nch),dtype=bool); chk=np.zeros((V,nch),dtype=bool)
        for v in range(V):
            fs=ev_f[starts[v]:ends[v]]; xs=ev_x[starts[v]:ends[v]]
            for (A,D,R) in es:
                # walk events
                for k in range(len(fs)):
                    f=int(fs[k]); nxt=int(fs[k+1]) if k+1<len(fs) else T
                    chk[v,min(f//8,nch-1)]=True
                    if xs[k]>0:
                        for b in (f+A, f+A+D):
                            if b<nxt and b<T: chk[v,b//8]=True
                    else:
                        e=min(f+R,nxt,T)
                        rel[v,f//8:(e+7)//8]=True
                        if f+R<nxt and f+R<T: chk[v,(f+R)//8]=True
        rel=rel[order]; chk=chk[order]
        W=V//64
        relw=rel.reshape(W,64,nch).any(1); chkw=chk.reshape(W,64,nch).any(1)
        n=W*nch
        c=chkw.sum(); r=(relw&~chkw).sum(); q=n-c-r
        res[w]=(q/n,r/n,c/n)
    return res
def cost(res):
    base={'w0':201,'w1':187,'w2':194}; relc={'w0':245,'w1':245,'w2':232}
    tot=0
    for w,(q,r,c) in res.items(): tot+=q*base[w]+r*relc[w]+c*2.0*base[w]
    return tot+ (8*15)  # wave 3
def by_off(plans,ev_v,ev_f,ev_x,T):
    key=np.full(V,1<<40,dtype=np.int64)
    off=ev_x<=0
    np.minimum.at(key,ev_v[off],ev_f[off])
    first=np.full(V,1<<40,dtype=np.int64); np.minimum.at(first,ev_v,ev_f)
    return np.lexsort((first,key))
V=8192  # sample
for name,span,fold in (("default 1s",48128,"scale"),("driver 20 blocks",5120,"slice")):
    a=model(span,fold); b=model(span,fold,by_off)
    print(name,"identity",{k:tuple(round(x,3) for x in v) for k,v in a.items()},"cost/8fr",round(cost(a),1))
    print(name,"grouped ",{k:tuple(round(x,3) for x in v) for k,v in b.items()},"cost/8fr",round(cost(b),1), "gain %.1f%%"%(100*(cost(a)/cost(b)-1)))
