"""bf16x3 decoder race hunt: LayerNorm-fused GEMM (decoder form) at small and big M, f32 vs split, accuracy and run-to-run determinism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parseq_amd import _native as nat
lib = nat.lib()
DEV = 'cuda'
E = 384
torch.manual_seed(0)
for M in (208, 4096, 13312):
    for N in (384, 1536, 95):
        x = torch.randn(M, E, device=DEV) * 1.3 + 0.2
        gamma, beta = torch.rand(E, device=DEV) + 0.5, torch.randn(E, device=DEV) * 0.1
        W = torch.randn(N, E, device=DEV) / E ** 0.5
        b = torch.randn(N, device=DEV) * 0.1
        want = (torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()).float()
        Wp = torch.empty(N * E, dtype=torch.float32, device=DEV)
        nat.check(lib.parseq_op_split_pack(nat.ptr(W), nat.ptr(Wp), N * E, nat.stream_ptr()))
        for code, wt, nm in ((nat.PARSEQ_F32, W, 'f32'), (nat.PARSEQ_BF16X3, Wp, 'bf16x3')):
            outs = []
            for rep in range(4):
                out = torch.full((M, N), float('nan'), device=DEV)
                nat.check(lib.parseq_op_ln_linear(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(wt), nat.ptr(b), nat.ptr(out), code, M, N, 1e-5, nat.stream_ptr()))
                torch.cuda.synchronize()
                outs.append(out)
            err = (outs[0] - want).abs()
            rep_d = max(float((outs[0] - o).abs().max()) for o in outs[1:])
            badrows = (err.amax(1) > 1e-3).nonzero().flatten().tolist()
            print(f'ln_linear {nm:7s} M={M} N={N}: max|d| {float(err.max()):.3e}  run-to-run {rep_d:.3e}  bad rows {len(badrows)} {badrows[:12]}')
            if badrows:
                r = badrows[0]
                # what would the row be if its A operand were all zeros (bias only), or LayerNorm with the neighbour row's statistics?
                print('   row', r, 'got', [round(float(v), 4) for v in outs[0][r, :5]], 'want', [round(float(v), 4) for v in want[r, :5]], 'bias', [round(float(v), 4) for v in b[:5]])
                xn = torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-5)
                hi = xn.float().bfloat16().double()
                for nm2, cand in (('hi-plane only', hi[r] @ W.double().T + b.double()), ('lo-plane only', (xn[r] - hi[r]) @ W.double().T + b.double()),
                                  ('other 32-row group same lane', xn[(r + 32) % M] @ W.double().T + b.double())):
                    print(f'      |got - {nm2}| max {float((outs[0][r].double() - cand).abs().max()):.3e}')
