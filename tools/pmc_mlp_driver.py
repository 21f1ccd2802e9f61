import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from parseq_amd import _native as nat
lib = nat.lib()
M, E, F = 65536, 384, 1536
x = torch.randn(M * E + 1024, device='cuda')
gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
W1 = (torch.randn(F, E, device='cuda') / E ** 0.5).bfloat16(); W2 = (torch.randn(E, F, device='cuda') / F ** 0.5).bfloat16()
b1, b2 = torch.randn(F, device='cuda') * 0.1, torch.randn(E, device='cuda') * 0.1
for _ in range(3):
    x.normal_()
    nat.check(lib.parseq_op_mlp(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W1), nat.ptr(b1), nat.ptr(W2), nat.ptr(b2), M, nat.stream_ptr()))
torch.cuda.synchronize()
