"""GPU parity tests, one per kernel family, each through the C ABI (parseq_op_*), against fp64 CPU references of the
same operator on the same seeded inputs.  Tolerances are stated next to each assertion."""
import ctypes as C

import pytest
import torch

from gpu_util import DEV, native, report

pytestmark = pytest.mark.gpu


def _gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('E', [192, 384, 768])
@pytest.mark.parametrize('rows', [1, 5, 1024])
def test_layernorm(E, rows):
    nat, lib = native()
    x = _gen(rows, E, seed=1, scale=3.0) + 0.5
    w, b = 1 + 0.1 * _gen(E, seed=2), 0.1 * _gen(E, seed=3)
    want = torch.nn.functional.layer_norm(x.double(), (E,), w.double(), b.double(), 1e-6).float()
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    for dt, code, tol in ((torch.float32, nat.PARSEQ_F32, 2e-5), (torch.bfloat16, nat.PARSEQ_BF16, 4e-2)):
        y = torch.empty(rows, E, dtype=dt, device=DEV)
        nat.check(lib.parseq_op_layernorm(nat.ptr(xd), nat.ptr(wd), nat.ptr(bd), nat.ptr(y), code, rows, E, 1e-6, nat.stream_ptr()))
        torch.cuda.synchronize()
        err, msg = report(f'layernorm E={E} rows={rows} {dt}', y, want)
        assert err <= tol, msg      # f32: rounding of the two reductions; bf16: one bf16 ulp at |y| <= ~5 is 2^-6
        if dt == torch.bfloat16:    # and it must be the correctly rounded value almost everywhere
            assert (y.float().cpu() == want.bfloat16().float()).float().mean() > 0.98


# (M, N, K): encoder shapes (big tiles), decoder shapes (small tiles), ragged M/N (bounds), K tail (patch embed K=96)
LINEAR_SHAPES = [(4096, 1152, 384), (4096, 384, 1536), (8192, 384, 384), (512, 384, 384), (26, 384, 384), (77, 95, 384),
                 (512, 1536, 384), (300, 768, 192), (4096, 384, 96), (130, 200, 96), (1, 95, 192)]


@pytest.mark.parametrize('M,N,K', [s for s in LINEAR_SHAPES if s[2] % 32 == 0])
def test_linear_bf16x3(M, N, K):
    """gemm.h SPLIT: f32 operands carried as bf16 hi / lo pairs, three MFMAs per product.  Against fp64 on UNROUNDED f32
    operands: the error is the dropped lo*lo term and the second rounding, ~2^-17 relative per product."""
    nat, lib = native()
    A = _gen(M, K, seed=4)
    W = _gen(N, K, seed=5) / K ** 0.5
    bias = 0.1 * _gen(N, seed=6)
    want = A.double() @ W.double().T + bias.double()
    Ad, Wd, bd = A.to(DEV), W.to(DEV).contiguous(), bias.to(DEV)
    Wp = torch.empty(N * K, dtype=torch.float32, device=DEV)              # same bytes, block-planar hi / lo
    nat.check(lib.parseq_op_split_pack(nat.ptr(Wd), nat.ptr(Wp), N * K, nat.stream_ptr()))
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_linear(nat.ptr(Ad), nat.ptr(Wp), nat.ptr(bd), nat.ptr(out), nat.PARSEQ_BF16X3, 0, M, N, K, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'linear bf16x3 {M}x{N}x{K}', out, want.float())
    assert err <= 5e-5, msg        # |sum| <= ~4, K <= 1536 random-sign terms of relative error 2^-17: ~1e-5 expected
    # and it must really be better than one bf16 product (guards against a path that silently drops the lo terms)
    one = (A.bfloat16().double() @ W.bfloat16().double().T + bias.double()).float()
    assert (one - want.float()).abs().max() > 20 * err


@pytest.mark.parametrize('M,N,K', LINEAR_SHAPES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_linear(M, N, K, dtype):
    nat, lib = native()
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    code = nat.PARSEQ_F32 if dtype == 'f32' else nat.PARSEQ_BF16
    A = _gen(M, K, seed=4).to(tdt)
    W = (_gen(N, K, seed=5) / K ** 0.5).to(tdt)            # asymmetric operands: catches transposed / swapped layouts
    bias = 0.1 * _gen(N, seed=6)
    want = A.double() @ W.double().T + bias.double()
    Ad, Wd, bd = A.to(DEV), W.to(DEV), bias.to(DEV)
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_linear(nat.ptr(Ad), nat.ptr(Wd), nat.ptr(bd), nat.ptr(out), code, 0, M, N, K, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'linear {dtype} {M}x{N}x{K}', out, want.float())
    assert err <= 2e-4, msg        # operands are exactly representable in `dtype`; only fp32 accumulation order differs
    if N % 4 == 0:                 # fused exact-erf GELU epilogue, output in storage dtype
        out2 = torch.full((M, N), float('nan'), dtype=tdt, device=DEV)
        nat.check(lib.parseq_op_linear(nat.ptr(Ad), nat.ptr(Wd), nat.ptr(bd), nat.ptr(out2), code, 1, M, N, K, nat.stream_ptr()))
        torch.cuda.synchronize()
        want2 = torch.nn.functional.gelu(want)
        err2, msg2 = report(f'linear+gelu {dtype} {M}x{N}x{K}', out2, want2.float())
        assert err2 <= (2e-4 if dtype == 'f32' else 2e-2), msg2   # bf16: output rounding, |y| <= ~4 -> ulp 2^-6


@pytest.mark.parametrize('heads,images', [(6, 3), (3, 2)])
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'bf16x3'])
def test_encoder_attention(heads, images, dtype):
    nat, lib = native()
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    code = {'f32': nat.PARSEQ_F32, 'bf16': nat.PARSEQ_BF16, 'bf16x3': nat.PARSEQ_BF16X3}[dtype]
    bh = heads * images
    q = _gen(bh, 128, 64, seed=7, scale=1.5).to(tdt)
    k = _gen(bh, 128, 64, seed=8, scale=1.5).to(tdt)
    v = _gen(bh, 128, 64, seed=9).to(tdt)
    want = torch.softmax(q.double() @ k.double().transpose(1, 2) * 64 ** -0.5, -1) @ v.double()     # [bh, 128, 64]
    want = want.view(images, heads, 128, 64).permute(0, 2, 1, 3).reshape(images * 128, heads * 64)
    qd, kd, vtd = q.to(DEV), k.to(DEV), v.transpose(1, 2).contiguous().to(DEV)     # keep the device tensors alive
    out = torch.full((images * 128, heads * 64), float('nan'), dtype=tdt, device=DEV)
    nat.check(lib.parseq_op_encoder_attention(nat.ptr(qd), nat.ptr(kd), nat.ptr(vtd), nat.ptr(out), code, bh, heads, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'enc attention {dtype} heads={heads}', out, want.float())
    # f32: exp/accumulation rounding.  bf16: probabilities and the output are rounded to bf16 (2^-9 relative each)
    # bf16x3: f32 tensors, operands as bf16 pairs -> ~2^-17 relative per product; scores here reach |s| ~ 20 (inputs scaled by 1.5),
    # so ~1e-4 absolute on a score and the same relative on its probability
    assert err <= {'f32': 2e-5, 'bf16': 1.5e-2, 'bf16x3': 1e-4}[dtype], msg


@pytest.mark.parametrize('variant', [0, 10])     # 0: load - LN - ... - reload - add - store; 10: x resident in the fc2 accumulators
@pytest.mark.parametrize('M', [128, 1000, 4096])
def test_fused_mlp(M, variant):
    """encoder_mlp.h: x += fc2(gelu(fc1(LN(x)))) in one kernel, against an fp64 reference with the same bf16 rounding points
    (LayerNorm output, weights, GELU output)."""
    nat, lib = native()
    E, F = 384, 1536
    x = _gen(M, E, seed=11, scale=1.5) + 0.2
    gamma, beta = 1 + 0.1 * _gen(E, seed=12), 0.1 * _gen(E, seed=13)
    W1 = (_gen(F, E, seed=14) / E ** 0.5).bfloat16()
    W2 = (_gen(E, F, seed=15) / F ** 0.5).bfloat16()
    b1, b2 = 0.1 * _gen(F, seed=16), 0.1 * _gen(E, seed=17)
    ln = torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-6).float().bfloat16().double()
    hidden = torch.nn.functional.gelu(ln @ W1.double().T + b1.double()).float().bfloat16().double()
    want = (x.double() + hidden @ W2.double().T + b2.double()).float()
    xd = x.to(DEV).clone()
    dev = [t.to(DEV) for t in (gamma, beta, W1, b1, W2, b2)]
    nat.check(lib.parseq_op_mlp_variant(nat.ptr(xd), nat.ptr(dev[0]), nat.ptr(dev[1]), nat.ptr(dev[2]), nat.ptr(dev[3]), nat.ptr(dev[4]),
                                        nat.ptr(dev[5]), M, variant, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'fused mlp M={M} variant={variant}', xd, want)
    # residual: bf16 re-rounding of LN / GELU values that land within fp32 noise of a rounding boundary (2^-9 relative on
    # a few of 1536 terms of magnitude <= 0.1) plus fp32 accumulation order
    assert err <= 5e-3, msg


@pytest.mark.parametrize('images', [1, 3, 40])
def test_fused_attention_branch(images):
    """encoder_attn_fused.h: x += proj(softmax(q k^T / 8) v), [q|k|v] = LN(x) Wqkv^T + b, one kernel, against an fp64 reference with
    the same bf16 rounding points (LayerNorm output, weights, q, k, v, un-normalised probabilities, attention output)."""
    nat, lib = native()
    E, H, N = 384, 6, 128
    M = images * N
    x = _gen(M, E, seed=21, scale=1.5) + 0.2
    gamma, beta = 1 + 0.1 * _gen(E, seed=22), 0.1 * _gen(E, seed=23)
    Wqkv = (_gen(3 * E, E, seed=24) * 1.5 / E ** 0.5).bfloat16()         # scores of a few units: a soft-max that is not flat
    Wproj = (_gen(E, E, seed=25) / E ** 0.5).bfloat16()
    bqkv, bproj = 0.1 * _gen(3 * E, seed=26), 0.1 * _gen(E, seed=27)
    r = lambda t: t.float().bfloat16().double()                          # noqa: E731  round to bf16, continue in fp64
    ln = r(torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-6))
    qkv = r(ln @ Wqkv.double().T + bqkv.double()).view(images, N, 3, H, 64).permute(2, 0, 3, 1, 4)       # [3, B, H, N, 64]
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.exp(s - s.amax(-1, keepdim=True))
    o = r((r(p) @ v) / p.sum(-1, keepdim=True))                          # probabilities rounded un-normalised, row sum exact
    ao = o.permute(0, 2, 1, 3).reshape(M, E)
    want = (x.double() + ao @ Wproj.double().T + bproj.double()).float()
    xd = x.to(DEV).clone()
    dev = [t.to(DEV) for t in (gamma, beta, Wqkv, bqkv, Wproj, bproj)]
    nat.check(lib.parseq_op_attn_fused(nat.ptr(xd), nat.ptr(dev[0]), nat.ptr(dev[1]), nat.ptr(dev[2]), nat.ptr(dev[3]), nat.ptr(dev[4]),
                                       nat.ptr(dev[5]), M, 0, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'fused attention branch images={images}', xd, want)
    # bf16 re-rounding of q / k / v / p / o values that land within fp32 noise of a rounding boundary (a 2^-9 relative step on one of
    # 64..384 terms), propagated through the soft-max, plus fp32 accumulation order
    assert err <= 1e-2, msg
    d = (xd.cpu() - want).abs()
    assert d.mean() <= 2e-4, f'mean |d| {d.mean():.3e}'                   # a wrong layout / permutation gives O(1) errors everywhere
