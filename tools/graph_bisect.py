import os, sys, gc
sys.path.insert(0, os.getcwd())
import torch
from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
dev = torch.device("cuda:0")
torch.manual_seed(7)
kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
static_inp = torch.zeros(2, 4, 32, 64, device=dev); static_tar = torch.zeros(2, 3, 32, 64, device=dev)
x, tar = torch.randn(2, 4, 32, 64, device=dev), torch.randn(2, 3, 32, 64, device=dev)
static_inp.copy_(x); static_tar.copy_(tar)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = net(static_inp); loss = ((pred.float() - static_tar) ** 2).mean()
        loss.backward()
    s.synchronize(); gc.collect(); torch.cuda.empty_cache()
    g = torch.cuda.CUDAGraph(); net.zero_grad(set_to_none=True)
    g.capture_begin()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = net(static_inp); loss = ((pred.float() - static_tar) ** 2).mean()
    loss.backward()
    g.capture_end()
torch.cuda.current_stream().wait_stream(s)
for i in range(4):
    xi = torch.randn_like(x) if i else x
    static_inp.copy_(xi); g.replay(); torch.cuda.synchronize()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        e = ((net(xi).float() - tar) ** 2).mean()
    print("replay", i, float(loss), "eager", float(e), "pred nan", int(torch.isnan(pred).sum()), flush=True)
