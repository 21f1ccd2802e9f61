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


def _unpack_pairs(buf, rows, cols):
    """Block-planar hi | lo bf16 pairs (32 logical elements -> 64 B of hi + 64 B of lo) back to f32 hi + lo."""
    raw = buf.view(torch.bfloat16).reshape(rows, cols // 32, 2, 32).float()
    return (raw[:, :, 0] + raw[:, :, 1]).reshape(rows, cols)


@pytest.mark.parametrize('M,N', [(4096, 384), (8192, 1536), (1000, 1152), (130, 96)])
def test_ln_linear_pairs_bf16x3(M, N):
    """The bf16x3 encoder's big-M form: layernorm_split_kernel writes the A operand as hi | lo pairs, gemm_kernel<PAIRS> reads both
    operands as pairs through the direct-to-LDS loop.  Against fp64 on unrounded operands, and (act) the pair-layout GELU epilogue."""
    nat, lib = native()
    E = 384
    x = _gen(M, E, seed=14) * 1.3 + 0.2
    gamma, beta = torch.rand(E, generator=torch.Generator().manual_seed(15)) + 0.5, 0.1 * _gen(E, seed=16)
    W = _gen(N, E, seed=17) / E ** 0.5
    bias = 0.1 * _gen(N, seed=18)
    xn = torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-6)
    want = xn @ W.double().T + bias.double()
    xd, gd, bd, Wd, biasd = x.to(DEV), gamma.to(DEV), beta.to(DEV), W.to(DEV).contiguous(), bias.to(DEV)
    Wp = torch.empty(N * E, dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_split_pack(nat.ptr(Wd), nat.ptr(Wp), N * E, nat.stream_ptr()))
    ws = torch.empty(M * E, dtype=torch.float32, device=DEV)
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_ln_linear_pairs(nat.ptr(xd), nat.ptr(gd), nat.ptr(bd), nat.ptr(Wp), nat.ptr(biasd), nat.ptr(out), nat.ptr(ws), 0, M, N, 1e-6, nat.stream_ptr()))
    torch.cuda.synchronize()
    # the LayerNorm output itself, unpacked from the pair layout: hi + lo carries ~16 mantissa bits
    err_ln, msg_ln = report(f'layernorm_split {M}x{E}', _unpack_pairs(ws.cpu(), M, E), xn.float())
    assert err_ln <= 1e-4, msg_ln
    err, msg = report(f'ln + linear pairs {M}x{N}', out, want.float())
    assert err <= 1e-4, msg
    one = (xn.float().bfloat16().double() @ W.bfloat16().double().T + bias.double()).float()
    assert (one - want.float()).abs().max() > 20 * err          # really better than one bf16 product
    if N % 32 == 0:
        out2 = torch.zeros(M * N, dtype=torch.float32, device=DEV)
        nat.check(lib.parseq_op_ln_linear_pairs(nat.ptr(xd), nat.ptr(gd), nat.ptr(bd), nat.ptr(Wp), nat.ptr(biasd), nat.ptr(out2), nat.ptr(ws), 1, M, N, 1e-6, nat.stream_ptr()))
        torch.cuda.synchronize()
        err2, msg2 = report(f'ln + linear + gelu, pair-layout output {M}x{N}', _unpack_pairs(out2.cpu(), M, N), torch.nn.functional.gelu(want).float())
        assert err2 <= 1e-4, msg2


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


@pytest.mark.parametrize('variant', [0, 10, 11])     # 0: load - LN - ... - reload - add - store; 10: x resident in the fc2 accumulators;
#                                                    11: the phase function the one-launch encoder uses (encoder_blocks.h)
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


@pytest.mark.parametrize('variant', [0, 1])          # 0: encoder_attn_fused.h; 1: the phase function the one-launch encoder uses
@pytest.mark.parametrize('images', [1, 3, 40])
def test_fused_attention_branch(images, variant):
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
                                       nat.ptr(dev[5]), M, variant, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'fused attention branch images={images}', xd, want)
    # bf16 re-rounding of q / k / v / p / o values that land within fp32 noise of a rounding boundary (a 2^-9 relative step on one of
    # 64..384 terms), propagated through the soft-max, plus fp32 accumulation order
    assert err <= 1e-2, msg
    d = (xd.cpu() - want).abs()
    assert d.mean() <= 2e-4, f'mean |d| {d.mean():.3e}'                   # a wrong layout / permutation gives O(1) errors everywhere


def _ref_block(x, P, r):
    """One timm Block in fp64 with the bf16 rounding points of the fused kernels (r = round-to-bf16-and-continue-in-fp64)."""
    M, E = x.shape
    images = M // 128
    ln = r(torch.nn.functional.layer_norm(x, (E,), P['g1'].double(), P['b1n'].double(), 1e-6))
    qkv = r(ln @ P['Wqkv'].double().T + P['bqkv'].double()).view(images, 128, 3, 6, 64).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.exp(s - s.amax(-1, keepdim=True))
    o = r((r(p) @ v) / p.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(M, E)
    x = (x + o @ P['Wproj'].double().T + P['bproj'].double()).float().double()          # the residual stream is fp32
    ln2 = r(torch.nn.functional.layer_norm(x, (E,), P['g2'].double(), P['b2n'].double(), 1e-6))
    hid = r(torch.nn.functional.gelu(ln2 @ P['W1'].double().T + P['b1'].double()))
    return (x + hid @ P['W2'].double().T + P['b2'].double()).float().double()


@pytest.mark.parametrize('images,depth', [(2, 1), (3, 3), (40, 2)])
def test_encoder_blocks_one_launch(images, depth):
    """encoder_blocks.h: `depth` blocks (attention branch + MLP branch each) in one launch with x resident in registers, against the
    fp64 reference with the same rounding points; distinct parameters per block (a kernel that re-used block 0's would fail)."""
    nat, lib = native()
    E, F = 384, 1536
    M = images * 128
    x = _gen(M, E, seed=31, scale=1.0) + 0.1
    blocks = []
    for l in range(depth):
        sd = 100 * l
        P = {'g1': 1 + 0.1 * _gen(E, seed=sd + 1), 'b1n': 0.1 * _gen(E, seed=sd + 2),
             'Wqkv': (_gen(3 * E, E, seed=sd + 3) * 1.5 / E ** 0.5).bfloat16(), 'bqkv': 0.1 * _gen(3 * E, seed=sd + 4),
             'Wproj': (_gen(E, E, seed=sd + 5) / E ** 0.5).bfloat16(), 'bproj': 0.1 * _gen(E, seed=sd + 6),
             'g2': 1 + 0.1 * _gen(E, seed=sd + 7), 'b2n': 0.1 * _gen(E, seed=sd + 8),
             'W1': (_gen(F, E, seed=sd + 9) / E ** 0.5).bfloat16(), 'b1': 0.1 * _gen(F, seed=sd + 10),
             'W2': (_gen(E, F, seed=sd + 11) / F ** 0.5).bfloat16(), 'b2': 0.1 * _gen(E, seed=sd + 12)}
        blocks.append(P)
    # the kernel addresses all matrices through one 32-bit buffer descriptor and all vectors relative to one base: keep each kind in
    # ONE allocation (separate torch tensors can sit in different allocator pools, more than 4 GiB apart) and hand out views
    order = ('g1', 'b1n', 'Wqkv', 'bqkv', 'Wproj', 'bproj', 'g2', 'b2n', 'W1', 'b1', 'W2', 'b2')
    wbuf = torch.empty(sum(P[k].numel() for P in blocks for k in order if P[k].dtype == torch.bfloat16), dtype=torch.bfloat16, device=DEV)
    vbuf = torch.empty(sum((P[k].numel() + 31) // 32 * 32 for P in blocks for k in order if P[k].dtype == torch.float32), dtype=torch.float32, device=DEV)
    keep, ptrs, wo, vo = [], [], 0, 0
    for P in blocks:
        for k in order:
            t = P[k]
            if t.dtype == torch.bfloat16:
                v = wbuf[wo:wo + t.numel()].view(t.shape); wo += t.numel()
            else:
                v = vbuf[vo:vo + t.numel()].view(t.shape); vo += (t.numel() + 31) // 32 * 32
            v.copy_(t)
            keep.append(v)
            ptrs.append(v.data_ptr())
    xd = x.to(DEV).clone()
    table = torch.empty(depth * 48, dtype=torch.uint8, device=DEV)
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    nat.check(lib.parseq_op_enc_blocks(nat.ptr(xd), arr, depth, M, nat.ptr(table), nat.stream_ptr()))
    torch.cuda.synchronize()
    # (1) bit for bit the chain of the stand-alone branch kernels built from the same phase functions (each individually checked
    #     against fp64 above): the only thing the one-launch form changes is that x stays in registers between them
    xc = x.to(DEV).clone()
    for l in range(depth):
        d_ = keep[12 * l:12 * l + 12]
        nat.check(lib.parseq_op_attn_fused(nat.ptr(xc), nat.ptr(d_[0]), nat.ptr(d_[1]), nat.ptr(d_[2]), nat.ptr(d_[3]), nat.ptr(d_[4]), nat.ptr(d_[5]), M, 1, nat.stream_ptr()))
        nat.check(lib.parseq_op_mlp_variant(nat.ptr(xc), nat.ptr(d_[6]), nat.ptr(d_[7]), nat.ptr(d_[8]), nat.ptr(d_[9]), nat.ptr(d_[10]), nat.ptr(d_[11]), M, 11, nat.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(xd, xc), f'one-launch encoder differs from the chained branch kernels by {float((xd - xc).abs().max()):.3e}'
    # (2) the fp64 reference with the same rounding points.  bf16 re-rounding makes the blocks chaotic on these random weights: a
    #     1e-4 difference after one block (a few bf16 roundings of LN / q / k / v / p / o / GELU values that flipped) grows ~30x per
    #     further block (measured with the reference itself under a 1e-4 perturbation: mean 2.6e-3 / max 1.7e-2 after the second
    #     block, 5e-3 / 4e-2 after the third), so only the first block is a sharp check and the rest a sanity bound
    r = lambda t: t.float().bfloat16().double()                          # noqa: E731
    want = x.double()
    for P in blocks:
        want = _ref_block(want, P, r)
    want = want.float()
    err, msg = report(f'encoder blocks images={images} depth={depth}', xd, want)
    d = (xd.cpu() - want).abs()
    assert err <= 1.5e-2 * 4 ** (depth - 1), msg
    assert d.mean() <= 3e-4 * 30 ** (depth - 1), f'mean |d| {d.mean():.3e}'


def _ref_block_exact(x, P):
    """One timm Block in fp64, no rounding anywhere (the bf16x3 arithmetic is held to the exact result)."""
    M, E = x.shape
    images = M // 128
    ln = torch.nn.functional.layer_norm(x, (E,), P['g1'].double(), P['b1n'].double(), 1e-6)
    qkv = (ln @ P['Wqkv'].double().T + P['bqkv'].double()).view(images, 128, 3, 6, 64).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    p = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(M, E)
    x = x + o @ P['Wproj'].double().T + P['bproj'].double()
    ln2 = torch.nn.functional.layer_norm(x, (E,), P['g2'].double(), P['b2n'].double(), 1e-6)
    hid = torch.nn.functional.gelu(ln2 @ P['W1'].double().T + P['b1'].double())
    return x + hid @ P['W2'].double().T + P['b2'].double()


@pytest.mark.parametrize('images,depth,tail', [(1, 1, False), (3, 2, False), (40, 3, True), (2, 12, True)])
def test_encoder_blocks_x3_one_launch(images, depth, tail):
    """encoder_blocks_x3.h: `depth` blocks (+ the final LayerNorm and the decoder's K | V projection as the tail) in one launch in the
    bf16x3 arithmetic, against an UNROUNDED fp64 reference; distinct parameters per block; repeated launches bit-identical."""
    nat, lib = native()
    E, F = 384, 1536
    M = images * 128
    x = _gen(M, E, seed=41, scale=1.0) + 0.1
    order = ('g1', 'b1n', 'Wqkv', 'bqkv', 'Wproj', 'bproj', 'g2', 'b2n', 'W1', 'b1', 'W2', 'b2')
    blocks = []
    for l in range(depth):
        sd = 1000 + 100 * l
        blocks.append({'g1': 1 + 0.1 * _gen(E, seed=sd + 1), 'b1n': 0.1 * _gen(E, seed=sd + 2),
                       'Wqkv': _gen(3 * E, E, seed=sd + 3) * 1.5 / E ** 0.5, 'bqkv': 0.1 * _gen(3 * E, seed=sd + 4),
                       'Wproj': _gen(E, E, seed=sd + 5) / E ** 0.5, 'bproj': 0.1 * _gen(E, seed=sd + 6),
                       'g2': 1 + 0.1 * _gen(E, seed=sd + 7), 'b2n': 0.1 * _gen(E, seed=sd + 8),
                       'W1': _gen(F, E, seed=sd + 9) / E ** 0.5, 'b1': 0.1 * _gen(F, seed=sd + 10),
                       'W2': _gen(E, F, seed=sd + 11) / F ** 0.5, 'b2': 0.1 * _gen(E, seed=sd + 12)})
    T = {'gn': 1 + 0.1 * _gen(E, seed=91), 'bn': 0.1 * _gen(E, seed=92), 'Wkv': _gen(2 * E, E, seed=93) / E ** 0.5, 'bkv': 0.1 * _gen(2 * E, seed=94)}
    tensors = [P[k] for P in blocks for k in order] + [T[k] for k in ('gn', 'bn', 'Wkv', 'bkv')]
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() + 31) // 32 * 32
    master = torch.zeros(total, dtype=torch.float32)
    for t, o in zip(tensors, offs):
        master[o:o + t.numel()] = t.reshape(-1)
    md = master.to(DEV)
    pack = torch.empty(total, dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_split_pack(nat.ptr(md), nat.ptr(pack), total, nat.stream_ptr()))
    o32 = (C.c_uint32 * (12 * depth))(*offs[:12 * depth])
    t32 = (C.c_uint32 * 4)(*offs[12 * depth:])
    table = torch.empty(depth * 48, dtype=torch.uint8, device=DEV)
    scratch = torch.empty(images * 393216 // 4, dtype=torch.float32, device=DEV)
    outs = []
    for rep in range(2):
        xd = x.to(DEV).clone()
        kmem = torch.full((images, 12, 128, 32), float('nan'), dtype=torch.float32, device=DEV)
        vmem = torch.full((images, 12, 128, 32), float('nan'), dtype=torch.float32, device=DEV)
        nat.check(lib.parseq_op_enc_blocks_x3(nat.ptr(xd), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32,
                                              nat.ptr(kmem) if tail else None, nat.ptr(vmem) if tail else None, nat.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((xd, kmem, vmem))
    want = x.double()
    for P in blocks:
        want = _ref_block_exact(want, P)
    scale = float(want.abs().max())
    tol = 2e-5 * depth * max(scale, 1.0)        # ~2^-16 relative per product, a handful of products per block
    if tail:
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), 'repeated launches differ (K / V)'
        kv = torch.nn.functional.layer_norm(want, (E,), T['gn'].double(), T['bn'].double(), 1e-6) @ T['Wkv'].double().T + T['bkv'].double()
        kw = kv[:, :E].view(images, 128, 12, 32).permute(0, 2, 1, 3)
        vw = kv[:, E:].view(images, 128, 12, 32).permute(0, 2, 1, 3)
        ek, msgk = report(f'x3 blocks tail K images={images} depth={depth}', outs[0][1], kw.float())
        ev, msgv = report(f'x3 blocks tail V images={images} depth={depth}', outs[0][2], vw.float())
        assert ek <= tol and ev <= tol, msgk + ' | ' + msgv
    else:
        assert torch.equal(outs[0][0], outs[1][0]), 'repeated launches differ'
        err, msg = report(f'x3 blocks images={images} depth={depth}', outs[0][0], want.float())
        assert err <= tol, msg
        # and it must really be better than single bf16 products (a path that dropped the lo terms would sit at ~1e-2)
        assert err <= 1e-3


@pytest.mark.parametrize('images,depth,tail', [(1, 1, False), (3, 2, True), (5, 12, True), (512, 1, True)])
def test_encoder_blocks_x3_eight_waves_bit_identical(images, depth, tail):
    """encoder_blocks_x3w.h (eight waves of 16 rows, two per SIMD: what parseq_forward launches) against encoder_blocks_x3.h (four waves of 32 rows) on the same
    inputs: every accumulator receives the same products in the same order, so x / the K | V rows must agree BIT FOR BIT (the four-wave kernel is the one the
    fp64 reference test above holds); 512 images = two rounds of workgroups per compute unit."""
    nat, lib = native()
    E, F = 384, 1536
    M = images * 128
    g = torch.Generator().manual_seed(77)
    shapes = [(E,), (E,), (3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,)]
    tens = []
    for l in range(depth):
        for i, sh in enumerate(shapes):
            t = torch.randn(*sh, generator=g)
            tens.append(t / sh[1] ** 0.5 if len(sh) == 2 else (1 + 0.1 * t if i in (0, 6) else 0.1 * t))
    tens += [1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g), torch.randn(2 * E, E, generator=g) / E ** 0.5, 0.1 * torch.randn(2 * E, generator=g)]
    offs, total = [], 0
    for t in tens:
        offs.append(total)
        total += (t.numel() + 31) // 32 * 32
    master = torch.zeros(total)
    for t, o in zip(tens, offs):
        master[o:o + t.numel()] = t.reshape(-1)
    md = master.to(DEV)
    pack = torch.empty(total, dtype=torch.float32, device=DEV)
    nat.check(lib.parseq_op_split_pack(nat.ptr(md), nat.ptr(pack), total, nat.stream_ptr()))
    o32 = (C.c_uint32 * (12 * depth))(*offs[:12 * depth])
    t32 = (C.c_uint32 * 4)(*offs[12 * depth:])
    table = torch.empty(depth * 48, dtype=torch.uint8, device=DEV)
    scratch = torch.empty(images * 393216 // 4, dtype=torch.float32, device=DEV)
    x = torch.randn(M, E, generator=g)
    outs = []
    for fn in (lib.parseq_op_enc_blocks_x3, lib.parseq_op_enc_blocks_x3w):
        xd = x.to(DEV).clone()
        kmem = torch.full((images, 12, 128, 32), float('nan'), dtype=torch.float32, device=DEV)
        vmem = torch.full((images, 12, 128, 32), float('nan'), dtype=torch.float32, device=DEV)
        nat.check(fn(nat.ptr(xd), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32,
                     nat.ptr(kmem) if tail else None, nat.ptr(vmem) if tail else None, nat.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((xd, kmem, vmem))
    if tail:
        assert bool(torch.isfinite(outs[1][1]).all()) and bool(torch.isfinite(outs[1][2]).all())
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), \
            f'K / V rows differ: max |dK| {float((outs[0][1] - outs[1][1]).abs().max()):.3e}, max |dV| {float((outs[0][2] - outs[1][2]).abs().max()):.3e}'
    else:
        assert bool(torch.isfinite(outs[1][0]).all())
        assert torch.equal(outs[0][0], outs[1][0]), f'x differs: max |dx| {float((outs[0][0] - outs[1][0]).abs().max()):.3e}'


def _patches(img):
    """[B, 3, 32, 128] -> [B * 128, 96]: token = 16 gy + gx, k = 32 c + 8 ky + kx (timm PatchEmbed's Conv2d(3, E, (4, 8), stride (4, 8))
    weight flattened)."""
    B = img.shape[0]
    return img.view(B, 3, 8, 4, 16, 8).permute(0, 2, 4, 1, 3, 5).reshape(B * 128, 96)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'u8'])
@pytest.mark.parametrize('images', [1, 5, 300])
def test_patch_head(images, dtype):
    """encoder_blocks.h patch_head alone (the head of the one-launch encoder, no blocks, x stored): x = patches W^T + (pos_embed + bias)
    against fp64 on the SAME bf16 operands — f32 pixels are rounded to bf16 by the kernel, u8 pixels get ToTensor + Normalize(0.5, 0.5)
    first (strhub/data/module.py:78-81).  Only the fp32 accumulation order differs: <= 2e-4 like test_linear."""
    nat, lib = native()
    E = 384
    g = torch.Generator().manual_seed(50 + images)
    if dtype == 'u8':
        raw = torch.randint(0, 256, (images, 3, 32, 128), generator=g, dtype=torch.uint8)
        pix = ((raw.float() / 255 - 0.5) / 0.5)
        dev_img, code = raw.to(DEV), nat.PARSEQ_U8
    else:
        pix = torch.rand(images, 3, 32, 128, generator=g) * 2 - 1
        if dtype == 'bf16':
            dev_img, code = pix.bfloat16().to(DEV), nat.PARSEQ_BF16
        else:
            dev_img, code = pix.to(DEV), nat.PARSEQ_F32
    W = (_gen(E, 96, seed=51) / 96 ** 0.5).bfloat16()
    posb = _gen(128, E, seed=52, scale=0.5)
    want = (_patches(pix.bfloat16().double()) @ W.double().T + posb.double().repeat(images, 1)).float()
    Wd, pd = W.to(DEV), posb.to(DEV)
    x = torch.full((images * 128, E), float('nan'), device=DEV)
    nat.check(lib.parseq_op_enc_head_tail(nat.ptr(x), nat.ptr(dev_img), code, nat.ptr(Wd), nat.ptr(pd), None, None, None, None, None, None,
                                          images * 128, nat.stream_ptr()))
    torch.cuda.synchronize()
    err, msg = report(f'patch_head images={images} {dtype}', x, want)
    assert err <= 2e-4, msg


@pytest.mark.parametrize('images', [1, 5, 300])
def test_kv_tail(images):
    """encoder_blocks.h kv_phase alone (the tail of the one-launch encoder, no blocks): K | V = LayerNorm(x) Wkv^T + bkv as bf16
    head-split rows, against fp64 with the same rounding points (LayerNorm output and weights bf16, result rounded to bf16).  A value
    that sits on a bf16 rounding boundary may land on the neighbouring bf16 number: at most one ulp, and rarely."""
    nat, lib = native()
    E, M = 384, images * 128
    x = _gen(M, E, seed=61, scale=1.5) + 0.2
    gn, bn = 1 + 0.1 * _gen(E, seed=62), 0.1 * _gen(E, seed=63)
    Wkv = (_gen(2 * E, E, seed=64) / E ** 0.5).bfloat16()
    bkv = 0.1 * _gen(2 * E, seed=65)
    ln = torch.nn.functional.layer_norm(x.double(), (E,), gn.double(), bn.double(), 1e-6).float().bfloat16().double()
    kv = ln @ Wkv.double().T + bkv.double()
    kw = kv[:, :E].reshape(images, 128, 12, 32).permute(0, 2, 1, 3).contiguous()
    vw = kv[:, E:].reshape(images, 128, 12, 32).permute(0, 2, 1, 3).contiguous()
    vec = torch.cat([gn, bn, bkv]).to(DEV)                       # one allocation: the kernel addresses the vectors relative to one base
    xd, Wd = x.to(DEV), Wkv.to(DEV)
    km = torch.zeros(images, 12, 128, 32, dtype=torch.bfloat16, device=DEV)
    vm = torch.zeros_like(km)
    nat.check(lib.parseq_op_enc_head_tail(nat.ptr(xd), None, 0, None, None, nat.ptr(vec), nat.ptr(vec[E:]), nat.ptr(Wd), nat.ptr(vec[2 * E:]),
                                          nat.ptr(km), nat.ptr(vm), M, nat.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x), 'the tail must not write x'
    for tag, got, want in (('K', km, kw), ('V', vm, vw)):
        got = got.float().cpu().double()
        d = (got - want).abs()
        ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -7      # bf16 spacing at |want|
        print(f'[kv_tail {tag} images={images}] max|d| {d.max():.3e}, max |d| / ulp {float((d / ulp).max()):.3f}, '
              f'exactly the rounded reference {(got == want.float().bfloat16().double()).float().mean():.4f}')
        # Sharp bound (one bf16 ulp of the result + 2e-5 for the fp32 accumulation order) on nearly every TOKEN; a token whose LayerNorm
        # output had one value land on the neighbouring bf16 number (fp32 vs fp64 statistics; a bf16 step of an O(4) value times a
        # weight of O(0.1)) moves all of its outputs by up to ~3e-3: allowed for a few tokens, never more than that
        sharp = (d <= 1.001 * ulp + 2e-5).all(-1)                         # [images, 12, 128]: per (head, token)
        tok_ok = sharp.all(1)                                             # per token
        print(f'[kv_tail {tag} images={images}] tokens inside the sharp bound {int(tok_ok.sum())}/{tok_ok.numel()}')
        assert tok_ok.float().mean() >= 0.95, f'{tag}: too many tokens outside one bf16 ulp'
        assert bool((d <= 1.001 * ulp + 4e-3).all()), f'{tag}: more than one bf16 ulp + 4e-3 from the fp64 reference'
        assert (got == want.float().bfloat16().double()).float().mean() >= 0.97
