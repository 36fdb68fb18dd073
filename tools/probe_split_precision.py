"""CPU experiment for DESIGN.md 10: a float32 GEMM of conv3_1's reduction length (473 x 9) emulated with bf16
MFMA products (operands split into 2-3 bf16 pieces, float32 accumulation) against float64.  No GPU needed."""
import torch
torch.manual_seed(0)
M,K,N=256,4257,256   # conv3_1: 473*9
a=torch.randn(M,K); b=torch.randn(K,N)*0.02
ref=(a.double()@b.double())
def split(x,n):
    parts=[]; r=x.clone()
    for _ in range(n):
        p=r.bfloat16().float(); parts.append(p); r=r-p
    return parts
def emu(n,terms):
    A=split(a,n); B=split(b,n); acc=torch.zeros(M,N)
    for i,j in terms: acc+= (A[i]@B[j])   # fp32 accumulate of exact bf16 products
    return acc
def err(x): return float(((x.double()-ref).abs().max()/ref.abs().max()))
print("fp32        ", err(a@b))
print("bf16x1      ", err(emu(1,[(0,0)])))
print("bf16x3 (3p) ", err(emu(2,[(0,0),(0,1),(1,0)])))
print("bf16x3 (6p) ", err(emu(3,[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)])))
print("bf16x3 (4p) ", err(emu(2,[(0,0),(0,1),(1,0),(1,1)])))
