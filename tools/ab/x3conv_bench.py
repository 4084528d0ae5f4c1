"""fp32 1x1 convolution on the bf16x3 engine vs torch (hipBLASLt) at the production shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from makani_amd import ops
dev = torch.device("cuda:0")
FULL, LOW = 721 * 1440, 240 * 480
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, M, K, P in [("fc1 384->768 full", 768, 384, FULL), ("384->384 full", 384, 384, FULL), ("enc 73->384 full", 384, 73, FULL),
                      ("dec 384->73 full", 73, 384, FULL), ("fc1 384->768 low", 768, 384, LOW), ("fc2 768->384 low", 384, 768, LOW),
                      ("384->384 low", 384, 384, LOW)]:
    w = torch.randn(M, K, device=dev) / K ** 0.5
    x = torch.randn(1, K, P, device=dev)
    a = t(lambda: ops.conv1x1_x3(w, x))
    b = t(lambda: torch.mm(w, x[0]))
    gy = torch.randn(1, M, P, device=dev)
    c = t(lambda: ops.conv1x1_x3_wgrad(gy, x))
    d = t(lambda: torch.mm(gy[0], x[0].t()))
    fl = 2.0 * M * K * P
    print(f"{name:20s} x3 fwd {a:8.3f} ms ({fl / a / 1e9:6.1f} TF fp32-equiv)  torch.mm {b:8.3f} ms | x3 wgrad {c:8.3f} ms  torch {d:8.3f} ms", flush=True)
