"""Forward-only time of the full configuration (bf16 autocast, batch 1) with and without autograd bookkeeping."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SphericalFourierNeuralOperatorNet(**bench.CONFIG).to(dev)
x = torch.randn(1, 73, 721, 1440, device=dev)
def run(n=8):
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, y
with torch.no_grad():
    t0, y0 = run()
t1, y1 = run()
print(f"forward only: no_grad {t0:.3f} ms, with autograd bookkeeping {t1:.3f} ms; outputs identical: {torch.equal(y0, y1.detach())}")


def run32(n=4):
    for _ in range(2):
        y = net(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        y = net(x)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.no_grad():
    t32 = run32()
os.environ["MK_CONV_FP32"] = "torch"
with torch.no_grad():
    t32v = run32()
print(f"forward only, fp32 (no autocast): {t32:.3f} ms on the bf16x3 engine; {t32v:.3f} ms with the vendor GEMM for the 1x1 convolutions (MK_CONV_FP32=torch)")
