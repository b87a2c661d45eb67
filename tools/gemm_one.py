import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import kernels as K
dev = torch.device('cuda:0')
M, N, Kd = 32768, 256, 768
x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(6):
    K.linear_fwd(x, W, b, act=0, out=y)
torch.cuda.synchronize()
