import torch, sys
sys.path.insert(0,'.')
from lightning_gan_zoo_amd import functional as F
for (M,N,K) in [(8,4096,32),(8,256,32),(8,4096,16),(4,2048,16),(512,16384,100)]:
    a=torch.randn(M,K).cuda(); b=torch.randn(K,N).cuda()
    try:
        c=F.gemm(a,b); torch.cuda.synchronize(); print((M,N,K),'ok',float((c.cpu()-a.cpu()@b.cpu()).abs().max()))
    except Exception as e: print((M,N,K),'ERR',e)
    try:
        c=F.gemm(a.t().contiguous(),b,trans_a=True); torch.cuda.synchronize(); print((M,N,K),'TN ok')
    except Exception as e: print((M,N,K),'TN ERR',e)
