"""Where does the bf16x3 mode lose accuracy?  Per-image comparison of bf16x3 against the fp32 mode of the library on distinct crops:
encoder memory and a teacher-forced decoder pass, at batch 8 (small GEMM tiles) and batch 512 (big tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import DEV, make_model
from oracle.synth import CONFIGS, synth_images

name = 'parseq'
cfg = CONFIGS[name]
images = synth_images(512, cfg, seed=20250924 + 512).to(DEV)
m32, mx3 = make_model(name, 'fp32'), make_model(name, 'bf16x3')
causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
with torch.inference_mode():
    for B in (8, 512):
        x = images[:B]
        a, b = m32.model.encode(x).clone(), mx3.model.encode(x).clone()
        d = (a - b).abs().flatten(1).max(1).values
        print(f'[batch {B}] encoder memory bf16x3 vs fp32: per-image max|d| median {d.median():.3e} max {d.max():.3e} worst images {d.topk(min(5, B)).indices.tolist()}')
        dr = (a - b).abs().amax(-1)          # [B, tokens]
        w = int(d.argmax())
        print(f'   worst image {w}: per-token max|d| top {dr[w].topk(5).values.tolist()} tokens {dr[w].topk(5).indices.tolist()}')
        full = m32(x, 25)
        toks = torch.cat([torch.full((B, 1), 95, device=DEV), full[:, :-1].argmax(-1)], 1)
        m32.model.encode(x); la = m32.model.decode_logits(toks, 0, 26, None, causal).clone()
        mx3.model.encode(x); lb = mx3.model.decode_logits(toks, 0, 26, None, causal).clone()
        d2 = (la - lb).abs().flatten(1).max(1).values
        print(f'[batch {B}] teacher-forced AR logits bf16x3 vs fp32: per-image max|d| median {d2.median():.3e} max {d2.max():.3e} worst {d2.topk(min(5, B)).indices.tolist()}')
        # same memory for both decoders: isolates the decoder
        mem = m32.model.encode(x)
        ha = m32.model.decode(toks, mem, tgt_query_mask=causal.to(DEV)).clone()
        hb = mx3.model.decode(toks, mem, tgt_query_mask=causal.to(DEV)).clone()
        d3 = (ha - hb).abs().flatten(1).max(1).values
        print(f'[batch {B}] decoder only (same fp32 memory): per-image max|d| median {d3.median():.3e} max {d3.max():.3e}')

# ---- error map of the decoder-only difference at batch 512
with torch.inference_mode():
    x = images[:512]
    full = m32(x, 25)
    toks = torch.cat([torch.full((512, 1), 95, device=DEV), full[:, :-1].argmax(-1)], 1)
    mem = m32.model.encode(x)
    ha = m32.model.decode(toks, mem, tgt_query_mask=causal.to(DEV)).clone()
    hb = mx3.model.decode(toks, mem, tgt_query_mask=causal.to(DEV)).clone()
    d = (ha - hb).abs()                       # [512, 26, E]
    rows = d.amax(-1).flatten()               # [512 * 26]
    bad = (rows > 1e-3).nonzero().flatten().tolist()
    print(f'rows (of {rows.numel()}) with max|d| > 1e-3: {len(bad)}')
    for mrow in bad[:40]:
        e = d.view(-1, d.shape[-1])[mrow]
        print(f'  m={mrow} image {mrow // 26} pos {mrow % 26} tile {mrow // 128} row-in-tile {mrow % 128}: max {e.max():.3e}, elements > 1e-3: {int((e > 1e-3).sum())}/{e.numel()}')
    # repeatability
    hb2 = mx3.model.decode(toks, mem, tgt_query_mask=causal.to(DEV)).clone()
    print('bf16x3 decode repeat: max diff between two runs', float((hb - hb2).abs().max()))
    # op-level big-M linear in split mode (ARowMajor + EpiStore / EpiGelu)
    from parseq_amd import _native as nat
    lib = nat.lib()
    for (M, N, K) in ((13312, 384, 384), (13312, 1536, 384), (13312, 384, 1536), (13312, 95, 384)):
        A = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV) * 0.1
        Wp = torch.empty(N * K, dtype=torch.float32, device=DEV)
        nat.check(lib.parseq_op_split_pack(nat.ptr(W), nat.ptr(Wp), N * K, nat.stream_ptr()))
        out = torch.empty(M, N, device=DEV)
        nat.check(lib.parseq_op_linear(nat.ptr(A), nat.ptr(Wp), nat.ptr(b), nat.ptr(out), nat.PARSEQ_BF16X3, 0, M, N, K, nat.stream_ptr()))
        want = (A.double() @ W.double().T + b.double()).float()
        print(f'op_linear bf16x3 {M}x{N}x{K}: max|d| {float((out - want).abs().max()):.3e}')
