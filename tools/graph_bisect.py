"""One-off diagnosis of test_hip_graph_capture_replay's NaN (same body as the test, more prints)."""
import gc, os, sys
sys.path.insert(0, os.getcwd())
import torch
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
dev = torch.device("cuda:0")
torch.manual_seed(7)
kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
static_inp = torch.zeros(2, 4, 32, 64, device=dev)
static_tar = torch.zeros(2, 3, 32, 64, device=dev)
x, tar = torch.randn(2, 4, 32, 64, device=dev), torch.randn(2, 3, 32, 64, device=dev)
static_inp.copy_(x); static_tar.copy_(tar)
variant = sys.argv[1] if len(sys.argv) > 1 else "test"
from makani_amd import layers, ops
STASH = []
_orig_rs = ops.row_sums
def _rs(t3):
    out = _orig_rs(t3)
    STASH.append(("row_sums in", t3)); STASH.append(("row_sums out", out))
    return out
if variant == 'stash': ops.row_sums = _rs
_orig_gemm = ops.pce_gemm
def _gemm(x3, wimg, m, **kw):
    out = _orig_gemm(x3, wimg, m, **kw)
    STASH.append((f"pce_gemm in  m={m} k={x3.shape[1]} P={x3.shape[2]} {sorted(k for k, v in kw.items() if v is not None and v is not False)}", x3))
    for i, o in enumerate(out if isinstance(out, tuple) else (out,)):
        STASH.append((f"pce_gemm out{i} m={m}", o))
    return out
if variant == 'stash': ops.pce_gemm = _gemm
cs = torch.cuda.Stream(); cs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cs):
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            static_pred = net(static_inp); static_loss = ((static_pred.float() - static_tar) ** 2).mean()
        static_loss.backward()
    cs.synchronize()
    ref_loss = static_loss.item()
    if variant != "nogradclone":
        ref_grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    gc.collect(); torch.cuda.empty_cache()
    graph = torch.cuda.CUDAGraph(); net.zero_grad(set_to_none=True)
    STASH.clear()
    graph.capture_begin()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        static_pred = net(static_inp); static_loss = ((static_pred.float() - static_tar) ** 2).mean()
    static_loss.backward()
    graph.capture_end()
torch.cuda.current_stream().wait_stream(cs)
def nan_report(tag):
    torch.cuda.synchronize()
    for name, t in STASH:
        if not torch.isfinite(t.float()).all():
            print("   non-finite:", name, tuple(t.shape), int((~torch.isfinite(t.float())).sum()))
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(tag, "loss", float(static_loss.detach()), "pred nonfinite", int((~torch.isfinite(static_pred)).sum()), "bad grads", bad[:6], flush=True)
for i in range(2):
    static_inp.copy_(x); static_tar.copy_(tar); graph.replay(); nan_report(f"replay x #{i}")
x2 = torch.randn_like(x)
static_inp.copy_(x2); graph.replay(); nan_report("replay x2")
static_inp.copy_(x2); graph.replay(); nan_report("replay x2 again")
static_inp.copy_(x); graph.replay(); nan_report("replay x")
with torch.autocast("cuda", dtype=torch.bfloat16):
    eager = ((net(x2).float() - tar) ** 2).mean()
print("eager x2", float(eager), "ref", ref_loss)
static_inp.copy_(x2); graph.replay(); nan_report("replay x2 after eager")
