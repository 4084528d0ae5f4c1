#!/usr/bin/env python
"""Diagnosis of the round-1 `CUDAGraph.capture_end` segfault (gpurun_out/tg.log).

Each scenario runs ONCE in its own child process (a crash must not take the others down) and prints one
line `SCENARIO <name> rc=<returncode> ...`.  Scenarios:

  torch_default_alive   pure torch (nn.Linear, no makani_amd op): one eager fwd+bwd on the DEFAULT stream whose loss
                        (and with it the autograd graph: its AccumulateGrad nodes) stays alive, then capture on the
                        side stream torch.cuda.graph() picks.  This is the pattern of the round-1 test.
  torch_side_alive      the same, but the eager step runs on the side stream that is then used for the capture
                        (the reference's sequence, makani/utils/trainer.py:109-148).
  sfno_default_alive    the round-1 test's pattern with the SFNO net (HIP kernels).
  sfno_reference_seq    trainer.py:109-148 literally with the SFNO net: warm-ups on the capture stream, static_loss
                        alive, gc.collect + empty_cache, capture_begin .. capture_end on that stream, replay.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net_and_data(kind):
    import torch
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    if kind == "torch":
        net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.GELU(), torch.nn.Linear(64, 8)).to(dev)
        x, tar = torch.randn(32, 64, device=dev), torch.randn(32, 8, device=dev)
    else:
        sys.path.insert(0, ROOT)
        from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
        kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
        net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
        x, tar = torch.randn(2, 4, 32, 64, device=dev), torch.randn(2, 3, 32, 64, device=dev)
    return net, x, tar


def scenario(name):
    import gc
    import torch
    kind, pattern = name.split("_", 1)
    net, x, tar = _net_and_data(kind)

    def fb():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        loss = ((y.float() - tar) ** 2).mean()
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    if pattern == "default_alive":
        with torch.cuda.stream(s):
            for _ in range(3):
                net.zero_grad(set_to_none=True)
                fb()
        torch.cuda.current_stream().wait_stream(s)
        net.zero_grad(set_to_none=True)
        keep = fb()                 # eager step on the default stream; graph (AccumulateGrad nodes) stays alive
        net.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = fb()
        g.replay()
        torch.cuda.synchronize()
        print("ok", float(loss), float(keep))
    else:
        with torch.cuda.stream(s):
            for _ in range(3):
                net.zero_grad(set_to_none=True)
                static_loss = fb()
            s.synchronize()
            gc.collect()
            torch.cuda.empty_cache()
            g = torch.cuda.CUDAGraph()
            net.zero_grad(set_to_none=True)
            g.capture_begin()
            static_loss = fb()      # the previous static_loss is released inside the capture
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        print("ok", float(static_loss))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
        sys.exit(0)
    for name in ("torch_default_alive", "torch_side_alive", "sfno_default_alive", "sfno_reference_seq"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "fault" in ln.lower() or "hip" in ln.lower()][:3]
        print(f"SCENARIO {name} rc={r.returncode} out={tail!r} err={err}", flush=True)
