"""Pixel-column engine (csrc/pce.hip) against a float64 GEMM on the same bf16 inputs, through the C ABI."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def _rel(a, b):
    return (torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double())).item()


# (M, K, P, batch): production channel counts on small pixel counts; ragged pixel tiles (P % 128 != 0), partial row
# tiles (73), two passes (768 rows), two K phases (768 -> 384), tiny test-net sizes; the last two give a two-pass layer at
# least one tile per workgroup, which is when the passes of a tile go to workgroup PAIRS (MK_PCE_SPLIT): whole and ragged
# second half, two batch items
SHAPES = [(384, 384, 1024, 1), (384, 384, 1000, 2), (768, 384, 520, 1), (384, 768, 776, 1), (384, 73, 640, 1),
          (200, 130, 72, 2), (384, 256, 64 * 300 + 24, 2), (768, 100, 64 * 520, 1),
          (73, 384, 264, 3), (73, 73, 1048, 1), (16, 8, 2048, 2), (3, 16, 72, 1), (96, 40, 392, 1), (128, 200, 136, 1),
          (200, 500, 256, 1), (768, 384, 128 * 140 + 40, 2), (500, 100, 128 * 260 + 8, 1)]


@pytest.mark.parametrize("M,K,P,B", SHAPES)
def test_pce_plain_gemm(dev, M, K, P, B):
    from makani_amd import ops
    torch.manual_seed(M * 7 + K)
    w = (torch.randn(M, K) / math.sqrt(K)).to(dev)
    x = torch.randn(B, K, P, device=dev).bfloat16()
    ref = torch.matmul(w.bfloat16().double(), x.double())
    y = ops.pce_gemm(x, ops.pce_pack(w), M)
    assert y.shape == (B, M, P) and y.dtype == torch.bfloat16
    assert _rel(y, ref) < 3e-3                      # one bf16 rounding of the output
    # bf16 weights and the transposed image (data gradient: A = W^T)
    yt = ops.pce_gemm(x, ops.pce_pack(w.t().contiguous().bfloat16(), transpose=True), M)
    assert torch.equal(yt, y)


@pytest.mark.parametrize("M,K,P,B", [(384, 384, 1000, 2), (768, 384, 520, 1), (73, 384, 264, 3), (16, 8, 2048, 2),
                                     (384, 73, 64 * 280 + 8, 2), (300, 200, 64 * 600, 1), (768, 384, 128 * 140 + 40, 2),
                                     (500, 384, 128 * 130 + 8, 2)])
def test_pce_epilogues(dev, M, K, P, B):
    from makani_amd import ops
    torch.manual_seed(11)
    w = (torch.randn(M, K) / math.sqrt(K)).to(dev)
    bias = torch.randn(M, device=dev)
    x = torch.randn(B, K, P, device=dev).bfloat16()
    add = torch.randn(B, M, P, device=dev).bfloat16()
    aux = torch.randn(B, M, P, device=dev).bfloat16()
    img = ops.pce_pack(w)
    acc = torch.matmul(w.bfloat16().double(), x.double()) + bias.double().view(1, -1, 1)
    # bias + GELU, pre-activation kept
    y, pre = ops.pce_gemm(x, img, M, bias=bias, want_pre=True, gelu=True)
    assert _rel(pre, acc) < 3e-3
    assert _rel(y, _gelu(acc)) < 3e-3
    # the GELU the kernel applies is the exact-erf one to within bf16 rounding of its own pre-activation
    assert _rel(y, _gelu(pre.double())) < 3e-3
    # addend
    y2 = ops.pce_gemm(x, img, M, addend=add)
    assert _rel(y2, acc - bias.double().view(1, -1, 1) + add.double()) < 3e-3
    # backward of the activation: y = acc * gelu'(aux)
    y3 = ops.pce_gemm(x, img, M, aux_in=aux)
    assert _rel(y3, (acc - bias.double().view(1, -1, 1)) * _gelu_grad(aux.double())) < 3e-3


def test_pce_gelu_accuracy(dev):
    """GELU / GELU' of the epilogue against the erf form over the whole range, via an identity GEMM."""
    from makani_amd import ops
    M = 32
    w = torch.eye(M, device=dev)
    xs = torch.linspace(-9.0, 9.0, 32 * 4096, device=dev).view(1, M, 4096).bfloat16()
    img = ops.pce_pack(w)
    y = ops.pce_gemm(xs, img, M, gelu=True).double()
    ref = _gelu(xs.double())
    assert (y - ref).abs().max().item() < 2e-2 and _rel(y, ref) < 2e-3
    ones = torch.ones_like(xs)
    g = ops.pce_gemm(ones, img, M, aux_in=xs).double()
    assert _rel(g, _gelu_grad(xs.double())) < 3e-3


@pytest.mark.parametrize("M,K,P,B", [(384, 768, 1000, 2), (768, 384, 520, 3), (73, 384, 264, 3), (16, 8, 2048, 2),
                                     (384, 384, 128 * 300 + 40, 2), (768, 384, 64 * 300 + 16, 2)])
def test_pce_row_sums(dev, M, K, P, B):
    """The epilogue's by-product: per-row (sum, sum of squares) of the STORED output, with every epilogue variant; the
    last shape gives each workgroup several tiles and a batch boundary inside its tile sequence."""
    from makani_amd import ops
    torch.manual_seed(5)
    w = (torch.randn(M, K) / math.sqrt(K)).to(dev)
    bias = torch.randn(M, device=dev)
    x = torch.randn(B, K, P, device=dev).bfloat16()
    aux = torch.randn(B, M, P, device=dev).bfloat16()
    img = ops.pce_pack(w)
    for kw in (dict(), dict(bias=bias, gelu=True), dict(aux_in=aux), dict(addend=aux)):
        y, sums = ops.pce_gemm(x, img, M, want_row_sums=True, **kw)
        assert torch.equal(y, ops.pce_gemm(x, img, M, **kw))
        assert sums.shape == (B * M, 2) and sums.dtype == torch.float64
        yd = y.double().view(B * M, P)
        np.testing.assert_allclose(sums[:, 0].cpu().numpy(), yd.sum(1).cpu().numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(P))
        np.testing.assert_allclose(sums[:, 1].cpu().numpy(), (yd * yd).sum(1).cpu().numpy(), rtol=1e-4)


def test_instance_norm_with_given_sums(dev):
    from makani_amd import ops
    torch.manual_seed(2)
    B, C, H, W = 2, 24, 20, 40
    w = torch.randn(C, 16, device=dev)
    x = torch.randn(B, 16, H * W, device=dev).bfloat16()
    y, sums = ops.pce_gemm(x, ops.pce_pack(w), C, want_row_sums=True)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    a = ops.instance_norm(y.view(B, C, H, W), g, b, 1e-5)
    c = ops.instance_norm(y.view(B, C, H, W), g, b, 1e-5, row_sums=sums)
    assert _rel(c, a) < 1e-5


@pytest.mark.parametrize("M,K,P,B", [(384, 384, 1000, 2), (73, 40, 264, 3), (768, 384, 520, 1), (768, 384, 128 * 140 + 40, 2)])
def test_pce_addend_affine(dev, M, K, P, B):
    """The addend enters as a[row] * addend + b[row]: the apply pass of an instance norm folded into the epilogue."""
    from makani_amd import ops
    torch.manual_seed(3)
    w = (torch.randn(M, K) / math.sqrt(K)).to(dev)
    x = torch.randn(B, K, P, device=dev).bfloat16()
    add = torch.randn(B, M, P, device=dev).bfloat16()
    aff = torch.randn(B * M, 2, device=dev)
    y = ops.pce_gemm(x, ops.pce_pack(w), M, addend=add, addend_affine=aff)
    ref = (torch.matmul(w.bfloat16().double(), x.double())
           + aff[:, 0].double().view(B, M, 1) * add.double() + aff[:, 1].double().view(B, M, 1))
    assert _rel(y, ref) < 3e-3


def test_conv_plus_instance_norm_node(dev):
    """conv(x) + instance_norm(z) as one launch: values and all five gradients against the torch composition (fp64)."""
    from makani_amd import ops
    from makani_amd.layers import Conv1x1, InstanceNorm2d, conv_plus_instance_norm
    torch.manual_seed(21)
    B, C, H, W = 2, 48, 20, 40
    conv = Conv1x1(C, C, bias=False).to(dev)
    norm = InstanceNorm2d(C, affine=True).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(C) + 0.5)
        norm.bias.copy_(torch.randn(C))
    x = torch.randn(B, C, H, W, device=dev).bfloat16().requires_grad_(True)
    src = torch.randn(B, 16, H * W, device=dev).bfloat16()
    z3, sums = ops.pce_gemm(src, ops.pce_pack(torch.randn(C, 16, device=dev) / 4), C, want_row_sums=True)
    z = z3.view(B, C, H, W).requires_grad_(True)
    y = conv_plus_instance_norm(conv, x, z, sums, norm)
    assert y is not None and y.dtype == torch.bfloat16
    gy = torch.randn_like(y)
    y.backward(gy)
    xd, zd = x.detach().double().requires_grad_(True), z.detach().double().requires_grad_(True)
    wd = conv.weight.detach().bfloat16().double().requires_grad_(True)
    nwd, nbd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd) + torch.nn.functional.instance_norm(zd, weight=nwd, bias=nbd, eps=norm.eps)
    yd.backward(gy.double())
    assert _rel(y, yd) < 4e-3
    assert _rel(x.grad, xd.grad) < 4e-3 and _rel(z.grad, zd.grad) < 1e-2
    assert _rel(conv.weight.grad, wd.grad) < 1e-3
    assert _rel(norm.weight.grad, nwd.grad) < 1e-3 and _rel(norm.bias.grad, nbd.grad) < 1e-3
