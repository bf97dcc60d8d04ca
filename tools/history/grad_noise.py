"""How much does the bf16 gradient error against an fp32 run of the same HIP model move between EQUALLY VALID kernel variants?
(fused feed-forward kernels: 64-row / 128-row, one or two workgroups per row block — different fp32 summation orders only.)
Per seed and variant: median / 90th percentile / worst relative L2 error over the parameter tensors.  Used to set test bounds."""
import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from s2t_amd import criterions as C, functional as Fn, kernels as K, s2t_transformer as M
DEV="cuda"; V=10000
def grads(mask, split1, dtype, seed, layers=4):
    K.ffn_configure(pc_mask=mask, split=int(split1))
    torch.manual_seed(seed)
    args = M.recipe_args(conformer=True, vocab_size=V, encoder_layers=layers, decoder_layers=2)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    with torch.no_grad():
        for p in model.parameters(): p.copy_(p.bfloat16().float())
    model.prepare(dtype, DEV)
    model.train()
    g=torch.Generator().manual_seed(seed+1)
    B,T=16,1000
    lens=torch.tensor(sorted([T]+[int(torch.randint(600,T+1,(1,),generator=g)) for _ in range(B-1)],reverse=True))
    src=torch.randn(B,T,80,generator=g).bfloat16().float()
    for bb,l in enumerate(lens): src[bb,l:]=0
    tgt=torch.randint(4,V,(B,21),generator=g); tgt[:,-1]=2; prev=torch.roll(tgt,1,1); prev[:,0]=2
    crit=C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V),label_smoothing=0.1,ctc_weight=0.3)
    sample={"net_input":{"src_tokens":src.to(DEV),"src_lengths":lens.to(DEV),"prev_output_tokens":prev.to(DEV)},"target":tgt.to(DEV),"ntokens":B*21}
    Fn._FFN_FUSED_MIN_ROWS=1024
    model.flat.zero_grad()
    loss=crit(model,sample)[0]; loss.backward(); torch.cuda.synchronize()
    return {k:p.grad.detach().float().clone() for k,p in model.named_parameters()}
for seed in (51, 77, 123):
    ref=grads(0,"0",torch.float32,seed)
    for mask,sp in ((5,"0"),(7,"0"),(7,"1"),(0,"0")):
        g=grads(mask,sp,torch.bfloat16,seed)
        errs=[float((g[k]-ref[k]).norm()/ref[k].norm().clamp_min(1e-12)) for k in ref if not k.endswith(("k_proj.bias","linear_k.bias"))]
        print("seed",seed,"mask",mask,"split1",sp,"median %.4f  p90 %.4f  max %.4f"%(np.median(errs),np.percentile(errs,90),max(errs)))
