"""The fused conv -> GELU -> conv node of the pixel-column engine (csrc/pce_mlp.hip, `mk_pce_mlp`) and the activated-operand
weight gradient (`mk_conv1x1_wgrad_act`) against float64 references with the same bf16 rounding points as the reference's
autocast path (layers.py:136-216: every convolution output and the GELU output are bf16 tensors).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def _bf(x):
    return x.to(torch.bfloat16).double()


def _rel(a, b):
    return (torch.linalg.norm(a.double() - b.double()) / max(torch.linalg.norm(b.double()).item(), 1e-30)).item()


# (K1, Hd, M, P, B): the production nodes (block MLP, encoder, decoder) on few pixels; all kernel instances (K1 <= 80 / 192 /
# 384 x M <= 96 / 384); ragged pixel tiles; hidden sizes that are not multiples of 32; more tiles than workgroups (the next
# tile's prefetch, the phase parked in registers) with two batch items (row sums change owner inside a workgroup's sequence)
SHAPES = [(384, 768, 384, 128 * 3 + 40, 1), (73, 384, 384, 1000, 2), (384, 384, 73, 520, 1), (4, 16, 8, 264, 2),
          (16, 64, 32, 72, 1), (100, 200, 130, 392, 1), (160, 96, 64, 136, 3), (384, 40, 384, 256, 1), (200, 8, 30, 1024, 1),
          (384, 768, 384, 128 * 300 + 8, 1), (73, 384, 384, 128 * 200 + 8, 2), (384, 384, 73, 128 * 150, 2),
          (150, 100, 150, 128 * 280, 1)]


def _make(K1, Hd, M, P, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, K1, P, generator=g).to(torch.bfloat16)
    w1 = torch.randn(Hd, K1, generator=g) * math.sqrt(2.0 / K1)
    w2 = torch.randn(M, Hd, generator=g) * math.sqrt(1.0 / Hd)
    b1 = torch.randn(Hd, generator=g) * 0.5
    b2 = torch.randn(M, generator=g) * 0.5
    return x, w1, w2, b1, b2


@pytest.mark.parametrize("K1,Hd,M,P,B", SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_fused_forward(dev, K1, Hd, M, P, B, with_bias):
    from makani_amd import ops
    if (P > 20000) and not with_bias:
        pytest.skip("large cases once")
    x, w1, w2, b1, b2 = _make(K1, Hd, M, P, B, 1)
    packed = ops.pce_mlp_pack(w1.to(dev), False, w2.to(dev), False)
    out = ops.pce_mlp(x.to(dev), packed, 0, b1=b1.to(dev) if with_bias else None, b2=b2.to(dev) if with_bias else None,
                      want_row_sums=True)
    y, pre, sums = (t.cpu() for t in out)
    pre_ref = torch.matmul(_bf(w1), x.double()) + (b1.double().view(1, -1, 1) if with_bias else 0.0)
    assert _rel(pre, pre_ref) < 3e-3
    # downstream of the kernel's own rounded pre (a 1-ulp difference in pre would otherwise dominate)
    h = _bf(_gelu(pre.double()))
    y_ref = torch.matmul(_bf(w2), h) + (b2.double().view(1, -1, 1) if with_bias else 0.0)
    assert _rel(y, y_ref) < 3e-3
    assert (y.double() - y_ref).abs().max() <= 2.0 ** -7 * y_ref.abs().max() + 1e-6
    s = sums.view(B, M, 2)
    yd = y.double()
    assert _rel(s[..., 0], yd.sum(-1)) < 1e-5 and _rel(s[..., 1], (yd * yd).sum(-1)) < 1e-5


@pytest.mark.parametrize("K1,Hd,M,P,B", SHAPES)
def test_fused_backward(dev, K1, Hd, M, P, B):
    """mode 1 with the roles of the backward pass: x = gy [B, M_fwd, P], A1 = W2^T, A2 = W1^T."""
    from makani_amd import ops
    x, w1, w2, b1, b2 = _make(K1, Hd, M, P, B, 2)
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(B, M, P, generator=g).to(torch.bfloat16)
    pre = (torch.randn(B, Hd, P, generator=g) * 1.5).to(torch.bfloat16)
    packed = ops.pce_mlp_pack(w2.to(dev), True, w1.to(dev), True)       # A1 = W2^T [Hd, M], A2 = W1^T [K1, Hd]
    assert packed[1:] == (K1, Hd, M)
    gx, gpre, gsum = (t.cpu() for t in ops.pce_mlp(gy.to(dev), packed, 1, pre=pre.to(dev), want_mid_sums=True))
    gh = torch.matmul(_bf(w2).t(), gy.double())
    gpre_ref = gh * _gelu_grad(pre.double())
    assert _rel(gpre, gpre_ref) < 3e-3
    gx_ref = torch.matmul(_bf(w1).t(), gpre.double())
    assert _rel(gx, gx_ref) < 3e-3
    assert _rel(gsum.view(B, Hd), gpre.double().sum(-1)) < 1e-5


@pytest.mark.parametrize("B,O,I,P", [(1, 384, 768, 4096 + 8), (2, 73, 384, 33 * 64), (1, 200, 96, 5000), (1, 384, 384, 240 * 480),
                                      (3, 5, 7, 24), (1, 384, 768, 400008)])
def test_wgrad_activated_operand(dev, B, O, I, P):
    from makani_amd import ops
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(B, O, P, generator=g).to(torch.bfloat16)
    pre = (torch.randn(B, I, P, generator=g) * 1.5).to(torch.bfloat16)
    gw = ops.conv1x1_wgrad_raw(gy.to(dev), pre.to(dev), x_gelu=True).cpu()
    h = _bf(_gelu(pre.double()))
    want = torch.einsum("bop,bip->oi", gy.double(), h)
    # the kernel's GELU (1.5e-7 absolute) rounds a few values per thousand to the neighbouring bf16 number
    assert _rel(gw, want) < 1e-3
    plain = ops.conv1x1_wgrad_raw(gy.to(dev), h.to(torch.bfloat16).to(dev)).cpu()
    assert _rel(gw, plain) < 1e-3
