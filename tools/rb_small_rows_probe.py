import os, sys
sys.path.insert(0, "/root/repo")
import torch
from s2t_amd import kernels as K
DEV="cuda"; M=3904; D=256; NB=6
g=torch.Generator().manual_seed(0)
xs=[torch.randn(M,D,generator=g).bfloat16().to(DEV) for _ in range(NB)]
gam=torch.ones(D,device=DEV); bet=torch.zeros(D,device=DEV)
xl=torch.empty(M,D,dtype=torch.bfloat16,device=DEV); mean=torch.empty(M,device=DEV); rstd=torch.empty(M,device=DEV)
def timeit(fn,rounds=20):
    for i in range(NB): fn(i)
    torch.cuda.synchronize(); best=1e9
    for _ in range(3):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            for i in range(NB): fn(i)
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/(rounds*NB)*1e3)
    return best
for N in (64,128,192,256,768,2560,10000):
    Np=(N+7)//8*8
    ws=[(torch.randn(N,D,generator=g)*D**-0.5).bfloat16().to(DEV) for _ in range(NB)]
    ys=[torch.empty(M,Np,dtype=torch.bfloat16,device=DEV) for _ in range(NB)]
    bias=torch.zeros(N,device=DEV)
    try:
        t0=timeit(lambda i:K.rowblock_gemm(xs[i],ws[i],ys[i],N=N,ldc=Np,bias=bias))
        t1=timeit(lambda i:K.rowblock_gemm(xs[i],ws[i],ys[i],N=N,ldc=Np,bias=bias,ln=(gam,bet),x_ln=xl,ln_stats=(mean,rstd)))
        print("rowblock M %d N %5d: %6.2f us plain, %6.2f us with LN" % (M,N,t0,t1),flush=True)
    except Exception as e:
        print("rowblock N",N,"failed:",str(e)[:100])
    t2=timeit(lambda i:K.gemm(xs[i],ws[i],ys[i],M=M,N=N,K=D,lda=D,ldb=D,ldc=Np,bias=bias))
    print("s2t_gemm M %d N %5d: %6.2f us" % (M,N,t2),flush=True)
