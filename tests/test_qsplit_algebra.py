"""The algebra behind the AR step's split mid kernel (parseq_amd/csrc/decoder_step.h DS_QS, decoder_attn.h QAsm / q_finish), restated
in numpy and held against the plain chain  q = LayerNorm(x) @ Wq^T + bq,  x = pos_query + sa @ Wo^T + bo  (modules.py:71-85 as the decoder's
query stream applies it: self-attention out_proj + residual, norm1, cross-attention q-projection).  What the kernels do:

  * every workgroup knows the row mean BEFORE x exists:  mean(x) = c0 + sa . wbar,  wbar = column means of Wo,  c0 = mean(pos_query + bo);
  * workgroup s owns a column slice of x, centres it with that estimate m, scales it by ln_w, and multiplies by the same K-slice of Wq:
    qp[s] = ((x - m) * ln_w)[:, cols_s] @ Wq[:, cols_s]^T,  leaving  S1[s] = sum (x - m),  S2[s] = sum (x - m)^2  over its columns;
  * the consumer finishes LayerNorm behind the product:  d = sum_s S1 / E,  var = sum_s S2 / E - d^2,
    q = rsqrt(var + eps) * (sum_s qp[s] - d * cq) + bq2,   cq = Wq @ ln_w,   bq2 = Wq @ ln_b + bq.

The GPU test of the kernels themselves is tests/test_hip_parity.py::test_ar_step_split_over_workgroups_vs_one_workgroup."""
import numpy as np
import pytest


def chain(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq, eps):
    x = posq + sa @ Wo.T + bo
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * ln_w + ln_b) @ Wq.T + bq, x


def split_form(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq, eps, nsplit, mean_error=0.0):
    E = Wo.shape[0]
    wbar, c0 = Wo.mean(0), (posq + bo).mean()
    cq, bq2 = Wq @ ln_w, Wq @ ln_b + bq                     # dec_qfold_kernel
    m = (c0 + sa @ wbar)[:, None] + mean_error              # the estimate every workgroup computes for itself
    qp, s1, s2 = 0.0, 0.0, 0.0
    for s in range(nsplit):
        cols = slice(s * E // nsplit, (s + 1) * E // nsplit)
        xs = posq[cols] + sa @ Wo[cols].T + bo[cols]        # this workgroup's columns of x: a row slice of Wo
        d = xs - m
        qp = qp + (d * ln_w[cols]) @ Wq[:, cols].T          # K-slice of the q-projection
        s1, s2 = s1 + d.sum(-1, keepdims=True), s2 + (d * d).sum(-1, keepdims=True)
    dd = s1 / E                                             # q_finish
    rstd = 1.0 / np.sqrt(s2 / E - dd * dd + eps)
    return rstd * (qp - dd * cq) + bq2


@pytest.mark.parametrize('E,nsplit', [(384, 3), (192, 3)])
@pytest.mark.parametrize('offset', [0.0, 3.0])              # a residual stream whose mean is far from zero must not cost accuracy
def test_split_chain_equals_layernorm_then_projection(E, nsplit, offset):
    rng = np.random.default_rng(E + int(offset))
    rows = 16
    sa = rng.standard_normal((rows, E))
    Wo, Wq = rng.standard_normal((E, E)) / np.sqrt(E), rng.standard_normal((E, E)) / np.sqrt(E)
    bo, bq, posq = 0.1 * rng.standard_normal(E), 0.1 * rng.standard_normal(E), rng.standard_normal(E) + offset
    ln_w, ln_b, eps = 1 + 0.1 * rng.standard_normal(E), 0.1 * rng.standard_normal(E), 1e-5
    want, x = chain(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq, eps)
    got = split_form(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq, eps, nsplit)
    assert np.abs(got - want).max() <= 1e-11
    # the mean is known before x is
    assert np.abs((posq + bo).mean() + sa @ Wo.mean(0) - x.mean(-1)).max() <= 1e-12
    # an estimate that is off (rounded operands in the real kernels: ~1e-6; here grossly) is corrected exactly by the column sums
    off = split_form(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq, eps, nsplit, mean_error=0.25)
    assert np.abs(off - want).max() <= 1e-10
    # and in float32 the form is as accurate as the plain chain is
    f = lambda *a: [np.asarray(v, dtype=np.float32) for v in a]      # noqa: E731
    a32 = f(sa, Wo, bo, posq, ln_w, ln_b, Wq, bq)
    got32 = split_form(*a32, np.float32(eps), nsplit)
    plain32, _ = chain(*a32, np.float32(eps))
    scale = np.abs(want).max()
    assert np.abs(got32 - want).max() <= 2.0 * max(np.abs(plain32 - want).max(), 1e-6 * scale)
