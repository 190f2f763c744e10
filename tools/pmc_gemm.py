import os, sys, torch
sys.path.insert(0, os.getcwd())
from flid_amd import ops
dev = torch.device("cuda:0")
f = lambda *s: torch.randn(*s, device=dev)
R = 13622
a, b, c = f(R, 888), f(272, 888), torch.empty(R, 272, device=dev)
for _ in range(5): ops.gemm(a, b, c, tb=True)
dres, agg, dV, br = f(R, 272), f(R, 888), torch.zeros(272, 888, device=dev), torch.zeros(272, device=dev)
for _ in range(5): ops.wgrad_group([(dres, agg, dV, br)])
torch.cuda.synchronize()
