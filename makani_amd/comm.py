"""Process-group tree for spatial model parallelism over torch.distributed (RCCL on MI355X).

Mirrors the accessors of ``makani/utils/comm.py:32-153`` (``init``, ``get_size``,
``get_rank``, ``get_group``, ``get_names`` ...) without the modulus
``DistributedManager``: the same orthogonal tree

    world -> { model -> { spatial -> {h, w}, matmul -> {fin, fout} }, data }

is built directly from ``torch.distributed.new_group``.  Rank layout: row-major over
(data, fout, fin, w, h) with ``h`` fastest, so one model-parallel instance occupies
consecutive ranks (= the GPUs of one xGMI island) and data parallelism goes across.
One process per GPU; backend "nccl" (RCCL) on GPUs, "gloo" on CPU (tests).
"""
import math
import os

import torch
import torch.distributed as dist

_LEAVES = ["h", "w", "fin", "fout"]
_state = None  # dict: sizes, ranks, groups
_lane = 0      # which copy of the group tree the collectives of the calling code use (see add_lane)


def _require():
    return _state is not None and _state["world_size"] > 1


def get_size(name):
    return _state["sizes"][name] if _require() else 1


def get_rank(name):
    return _state["ranks"][name] if _require() else 0


def get_group(name, lane=None):
    if _state is None:
        return None
    lane = _lane if lane is None else lane
    if lane == 0:
        return _state["groups"].get(name)
    return _state["lanes"][lane - 1].get(name)


# -- lanes: independent copies of the group tree (separate RCCL communicators, hence separate collective queues).
# Two micro-batches running on two HIP streams use one lane each, so the all-to-alls of one overlap the kernels of
# the other instead of queueing behind the other micro-batch's collectives (bench.py, N > 1).
def add_lane():
    """Create one more copy of every process group (collective: every rank calls it, in the same order)."""
    if _state is None:
        return 0
    groups = {}
    for gname, buckets in _state["members"].items():
        for rl in buckets:
            grp = dist.new_group(rl) if len(rl) > 1 else None
            if _state["world_rank"] in rl:
                groups[gname] = grp
    _state["lanes"].append(groups)
    return len(_state["lanes"])


def num_lanes():
    return 1 + (len(_state["lanes"]) if _state is not None else 0)


def get_lane():
    return _lane


def set_lane(i):
    global _lane
    if i < 0 or i >= num_lanes():
        raise ValueError(f"lane {i} does not exist ({num_lanes()} lane(s))")
    _lane = int(i)


class lane:
    """``with comm.lane(i):`` -- collectives issued inside use lane i (autograd nodes re-enter their forward lane)."""

    def __init__(self, i):
        self.i, self.prev = i, 0

    def __enter__(self):
        self.prev = get_lane()
        set_lane(self.i)

    def __exit__(self, *exc):
        set_lane(self.prev)


def get_root(name):
    return _state["roots"].get(name, 0) if _require() else 0


def get_world_size():
    return _state["world_size"] if _state is not None else 1


def get_world_rank():
    return _state["world_rank"] if _state is not None else 0


def get_local_rank():
    return _state["local_rank"] if _state is not None else 0


def get_names():
    return list(_state["groups"].keys()) if _state is not None else []


def is_distributed(name):
    return _state is not None and name in _state["groups"]


def _coords(rank, sizes):
    """rank -> {name: coordinate}; h fastest, data slowest."""
    out = {}
    for name in ["h", "w", "fin", "fout", "data"]:
        out[name] = rank % sizes[name]
        rank //= sizes[name]
    return out


def init(model_parallel_sizes=(1, 1, 1, 1), model_parallel_names=("h", "w", "fin", "fout"), verbose=False,
         backend=None):
    """Initialise torch.distributed from the torchrun environment and build the group tree.

    Returns the model-parallel size, like the reference's ``comm.init`` (comm.py:97-153).
    """
    global _state
    if not dist.is_initialized():
        world_size = int(os.environ.get("WORLD_SIZE", "1"))
        if world_size == 1 and "RANK" not in os.environ:
            _state = None
            return 1
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # A collective that does not complete within MK_COLLECTIVE_TIMEOUT seconds (default 300) makes the watchdog
        # abort the process with a non-zero exit code instead of hanging it (a rank that died, a mismatched sequence).
        import datetime
        timeout = datetime.timedelta(seconds=int(os.environ.get("MK_COLLECTIVE_TIMEOUT", "300")))
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")     # tear the process down on a timeout
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank), timeout=timeout)
        else:
            dist.init_process_group(backend=backend, timeout=timeout)
    world_size, rank = dist.get_world_size(), dist.get_rank()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    sizes = {n: 1 for n in _LEAVES}
    for n, s in zip(model_parallel_names, model_parallel_sizes):
        if n not in sizes:
            raise ValueError(f"unknown model-parallel group {n}")
        sizes[n] = int(s)
    model_size = math.prod(sizes.values())
    if world_size % model_size:
        raise ValueError(f"world size {world_size} not divisible by model-parallel size {model_size}")
    sizes["data"] = world_size // model_size
    sizes["spatial"] = sizes["h"] * sizes["w"]
    sizes["matmul"] = sizes["fin"] * sizes["fout"]
    sizes["model"] = model_size

    members = {"h": ["h"], "w": ["w"], "fin": ["fin"], "fout": ["fout"], "data": ["data"],
               "spatial": ["h", "w"], "matmul": ["fin", "fout"], "model": ["h", "w", "fin", "fout"]}
    all_coords = [_coords(r, sizes) for r in range(world_size)]
    groups, ranks, roots, member_lists = {}, {}, {}, {}
    for gname, axes in members.items():
        # ranks sharing every coordinate NOT in `axes` form one group; every rank creates all of them
        buckets = {}
        for r, c in enumerate(all_coords):
            key = tuple(c[a] for a in ["h", "w", "fin", "fout", "data"] if a not in axes)
            buckets.setdefault(key, []).append(r)
        member_lists[gname] = [buckets[key] for key in sorted(buckets)]
        for key in sorted(buckets):
            rl = buckets[key]
            grp = dist.new_group(rl) if len(rl) > 1 else None
            if rank in rl:
                groups[gname] = grp
                ranks[gname] = rl.index(rank)
                roots[gname] = min(rl)
    global _lane
    _lane = 0
    _state = {"sizes": sizes, "ranks": ranks, "groups": groups, "roots": roots, "world_size": world_size,
              "world_rank": rank, "local_rank": local_rank, "members": member_lists, "lanes": []}
    if verbose and rank == 0:
        print(f"makani_amd.comm: world {world_size}, sizes {sizes}")
    return model_size


def cleanup():
    global _state, _lane
    _state = None
    _lane = 0
    if dist.is_initialized():
        dist.destroy_process_group()
