import os, sys, torch
sys.path.insert(0, '.')
from makani_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, M, K, P) in ((1, 384, 73, 181 * 1440), (1, 384, 384, 181 * 1440), (1, 73, 384, 178 * 1440), (1, 768, 384, 60 * 480), (1, 384, 73, 721 * 1440)):
    w = torch.randn(M, K, device=dev) / K ** 0.5
    x = torch.randn(B, K, P, device=dev)
    y = ops.conv1x1_x3(w, x)
    want = torch.matmul(w.double(), x.double())
    e = ((y.double() - want).norm() / want.norm()).item()
    bad = (~torch.isfinite(y)).sum().item()
    # where is the error
    d = (y.double() - want).abs().amax(dim=(0, 1))
    worst = torch.topk(d, 3).indices.tolist()
    print(B, M, K, P, f"rel {e:.2e} nonfinite {bad} worst px {worst} of {P}")
