import torch, time, sys
sys.path.insert(0, '/root/repo')
import torch_semantic_segmentation_amd as tssa
from oracle.recipe import ohem
dev='cuda:0'
B,C,H,W=8,19,1024,2048
logits=(torch.randn(B,C,H,W,device=dev)).bfloat16()
target=torch.randint(0,C,(B,H,W),device=dev)
target[torch.rand(B,H,W,device=dev)<0.1]=255
loss=tssa.OHEMLoss(ignore_index=255)
def run(fn, x):
    x=x.clone().requires_grad_(True)
    for _ in range(2):
        l=fn(x,target); l.backward(); x.grad=None
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5):
        l=fn(x,target); l.backward(); x.grad=None
    torch.cuda.synchronize(); return (time.perf_counter()-t)/5*1e3, l.item()
print('hip  ohem fwd+bwd ms, loss:', run(loss, logits))
print('torch ohem (sort) fwd+bwd ms, loss:', run(lambda a,t: ohem(a.float(), t, ignore_index=255), logits))
