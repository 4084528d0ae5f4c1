"""Diagnostic: configs[3] forward (h = 4 on one GPU, gloo), several runs with per-module output checksums; prints the first
module whose checksum moves between runs (fp32 forward is deterministic up to the fp64 atomics of the norm statistics)."""
import os
import sys
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
RUNS = int(os.environ.get("RUNS", "6"))


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from makani_amd import comm
    solo = os.environ.get("SOLO", "0") == "1"           # ranks share the card but not the model: no collective in the forward
    comm.init(model_parallel_sizes=[1 if solo else world, 1, 1, 1], backend="gloo")
    import bench
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from makani_amd.distributed import split_tensor_along_dim
    dev = torch.device("cuda:0")
    torch.manual_seed(333)
    net = SphericalFourierNeuralOperatorNet(**bench.CONFIG).to(dev)
    xg = torch.randn(1, 73, 721, 1440)
    xl = (xg if solo else split_tensor_along_dim(xg, 2, world)[rank].contiguous()).to(dev)
    names, diffs, ref, where = [], [], {}, []
    state = {"run": 0}
    keep = os.environ.get("KEEP", "encoder,blocks.0,blocks.1").split(",")

    def hook(name):
        def f(mod, inp, out):
            if not any(name == k or name.startswith(k + ".") for k in keep):
                return
            outs = list(out) if isinstance(out, (tuple, list)) else [out]
            for j, o in enumerate(outs):
                if not torch.is_tensor(o):
                    continue
                o = torch.view_as_real(o) if o.is_complex() else o
                key = f"{name}[{j}]#" + str(sum(1 for n in names if n.split("#")[0] == f"{name}[{j}]"))
                names.append(key)
                if state["run"] == 0:
                    ref[key] = o.clone()
                    continue
                r = ref[key]
                mask = torch.isfinite(r) & torch.isfinite(o)          # unwritten spectrum rows may hold anything
                d = torch.where(mask, o - r, torch.zeros_like(o)).double()
                diffs.append(torch.stack([d.norm() / torch.where(mask, r, torch.zeros_like(r)).double().norm(),
                                          d.abs().max(), (d != 0).sum().double()]))
                if o.dim() == 4 and len(where) < 4:
                    nz = (d != 0)
                    if bool(nz.any()):
                        idx = nz.nonzero()
                        ch = idx[:, 1].unique()
                        px = (idx[:, 2] * o.shape[3] + idx[:, 3]).unique()
                        where.append(f"rank {rank} run {state['run']} {key} shape {tuple(o.shape)}: {ch.numel()} channels "
                                     f"[{ch.min().item()}..{ch.max().item()}], {px.numel()} pixels [{px.min().item()}..{px.max().item()}] "
                                     f"rows {idx[:, 2].unique().tolist()[:12]} cols {(idx[:, 3].unique()).tolist()[:40]}")
                        if len(where) == 1:
                            c0, r0 = int(idx[0, 1]), int(idx[0, 2])
                            cols = idx[(idx[:, 1] == c0) & (idx[:, 2] == r0)][:, 3]
                            lo = max(int(cols.min()) - 4, 0)
                            hi = min(int(cols.max()) + 5, o.shape[3])
                            where.append(f"   ref  [{c0},{r0},{lo}:{hi}] " + " ".join(f"{v:+.4f}" for v in r[0, c0, r0, lo:hi].tolist()))
                            where.append(f"   new  [{c0},{r0},{lo}:{hi}] " + " ".join(f"{v:+.4f}" for v in o[0, c0, r0, lo:hi].tolist()))
        return f
    for n, m in net.named_modules():
        if n:
            m.register_forward_hook(hook(n))
    msgs = []
    for it in range(RUNS):
        names.clear()
        diffs.clear()
        state["run"] = it
        with torch.no_grad():
            y = net(xl)
        if it:
            d = torch.stack(diffs).cpu()
            bad = (d[:, 0] > 0).nonzero().flatten().tolist()
            if bad:
                msgs.append(f"rank {rank} run {it}: " + "; ".join(f"{names[i]} rel {d[i,0]:.1e} max {d[i,1]:.1e} n {int(d[i,2])}"
                                                                   for i in bad[:4]))
            else:
                msgs.append(f"rank {rank} run {it}: bit-identical over {len(names)} outputs")
        dist.barrier()
    q.put((rank, msgs + where))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = int(os.environ.get("WORLD", "4"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for _ in procs:
        r, msgs = q.get(timeout=900)
        print("\n".join(msgs), flush=True)
    for p in procs:
        p.join(timeout=60)
