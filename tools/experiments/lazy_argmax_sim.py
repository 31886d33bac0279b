import numpy as np, math, sys
rng=np.random.default_rng(0)
Q=np.array([48.9,67.6,-45.8,67.2,-46.3,66.6,60.3,66.9,60.6,-35.,60.7])
A=11; L=64; T=int(sys.argv[1]) if len(sys.argv)>1 else 20000
alpha=0.05; scale=150.; cap=100.; thres=10; rule=0
hoeff=scale*math.sqrt(math.log(1/alpha)/2)
mode=sys.argv[2] if len(sys.argv)>2 else 'sim1'
if mode=='rand': Qs=rng.uniform(-50,100,(L,A))
else: Qs=np.tile(Q,(L,1))
n=np.zeros((L,A)); s=np.zeros((L,A)); q=np.zeros((L,A))
V=np.full((L,A),-50.); V[:,rule]=100.
best=V.max(1); lead=V.argmax(1)
U=np.full(L,-50.)
lanes=np.arange(L)
resc=0; resc_lanes=0; hist=[]
for t in range(T):
    a=rng.integers(0,A,L)
    x=Qs[lanes,a]+50*rng.standard_normal(L)
    n[lanes,a]+=1; s[lanes,a]+=x; q[lanes,a]+=x*x
    nn=n[lanes,a]; m=s[lanes,a]/nn; var=np.maximum(q[lanes,a]/nn-m*m,0); sd=np.sqrt(var)
    up=np.minimum(cap,m+hoeff/np.sqrt(nn)); lo=m-hoeff/np.sqrt(nn); ci=m-4*sd/(nn+1)-hoeff/np.sqrt(nn+1)
    k=np.where(a==rule,up,np.minimum(lo,ci))
    valid=nn>thres
    oldk=V[lanes,a].copy()
    V[lanes,a]=np.where(valid,k,oldk)
    isL=(a==lead)&valid
    k2=np.where(valid,k,-np.inf)
    gt=(k2>best)&~isL
    # case2
    U=np.where(isL,U,np.maximum(U,np.where(gt,best,k2)))
    nb=np.where(gt|isL,np.where(valid,k,best),best)
    need=isL&(k<=U)
    best=nb; lead=np.where(gt,a,lead)
    if need.any():
        resc+=1; resc_lanes+=need.sum()
        srt=np.sort(V,1)
        best=srt[:,-1]; U=srt[:,-2]; lead=V.argmax(1)
    assert np.allclose(best,V.max(1)), t
    if (t+1)%2000==0: hist.append(resc)
print(mode,'T',T,'wave rescans',resc,'= %.1f %% of records'%(100*resc/T),'lane-needs',resc_lanes,'cum',hist)
