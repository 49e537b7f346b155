"""Request-sharded data parallelism across the GPUs of one node.

The reference has no multi-GPU support at all (README.md:28-30, 54). The MI355X plan:
one process per GPU, each a full replica with its OWN KV pool and block manager; a request lives on
one replica for its whole life, so there is NO data-path collective and no RCCL traffic. What the
ranks share is bookkeeping only — who takes which requests, a barrier around timed regions, and the
reduction of per-rank counters — done over a `gloo` group on the host.
"""
import datetime
import os
import socket
import subprocess
import sys
import time
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_units: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `num_units` requests owned by `rank`
    (the first num_units % world_size ranks take one extra)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(num_units, world_size)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard(items: Sequence, rank: int, world_size: int) -> List:
    b, e = shard_bounds(len(items), rank, world_size)
    return list(items[b:e])


def least_loaded(outstanding_tokens: Sequence[int]) -> int:
    """Online routing rule: a new request goes to the replica with the fewest outstanding tokens
    (ties -> lowest rank). Requests never migrate afterwards."""
    best = 0
    for i, v in enumerate(outstanding_tokens):
        if v < outstanding_tokens[best]:
            best = i
    return best


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torchrun exports them; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_control_group(timeout_s: int = 600) -> bool:
    """Join the host-side control group (gloo) when launched with WORLD_SIZE > 1."""
    _, _, world = env_rank_world()
    if world <= 1:
        return False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return True


def spawn_local_ranks(argv: Sequence[str], world_size: int, timeout_s: float = None, visible_devices=None) -> int:
    """Launch `world_size` copies of the command `argv` on this node, one per GPU, with the environment a launcher
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free MASTER_PORT) and
    rank i restricted to GPU i (HIP_VISIBLE_DEVICES=i, so every replica sees its device as "cuda:0" — the reference's
    default-device convention) unless the caller already restricted visibility. Rank 0 inherits stdout;
    the other ranks' stdout goes to stderr. Returns the worst exit code; when one rank fails the rest are terminated.
    What `bench.py --gpus N` uses when no torchrun environment is present."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    restricted = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if visible_devices is None and os.environ.get("SWIFTLLM_SPAWN_DEVICES"):
        # e.g. "0,0": smoke-test the N-rank launch path on a box with fewer GPUs (ranks share a device)
        visible_devices = [d.strip() for d in os.environ["SWIFTLLM_SPAWN_DEVICES"].split(",")]
        if len(visible_devices) != world_size:
            raise ValueError(f"SWIFTLLM_SPAWN_DEVICES names {len(visible_devices)} devices for {world_size} ranks")
    procs = []
    for r in range(world_size):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world_size),
                   LOCAL_WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if visible_devices is not None:
            env["HIP_VISIBLE_DEVICES"] = str(visible_devices[r])
        elif not restricted:
            env["HIP_VISIBLE_DEVICES"] = str(r)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else _stderr_fd()))
    worst = 0
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    kill_at = None                      # set once the survivors have been told to terminate
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    rc = p.wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if rc != 0 and kill_at is None:
                    worst = worst or rc
                    for q in pending:       # a dead rank leaves the others waiting in a barrier for ever
                        q.terminate()
                    kill_at = time.monotonic() + TERMINATE_GRACE_S
            now = time.monotonic()
            if deadline is not None and now >= deadline and pending and kill_at is None:
                for q in pending:
                    q.terminate()
                worst = worst or 124
                kill_at = now + TERMINATE_GRACE_S
            if kill_at is not None and now >= kill_at:
                for q in pending:           # stuck in a HIP call or ignoring SIGTERM
                    q.kill()
                kill_at = float("inf")
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


TERMINATE_GRACE_S = 10.0


def _stderr_fd():
    """Where the non-zero ranks' stdout goes: this process's stderr as a real file descriptor (sys.stderr may be a
    wrapper without one under pytest or a notebook)."""
    try:
        return sys.stderr.fileno()
    except (AttributeError, OSError, ValueError):
        return 2


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_numa_nodes() -> List[int]:
    """NUMA node of every AMD GPU of this host in device order (KFD topology: GPU nodes carry a non-zero
    simd_count and their PCI location, from which sysfs gives the node). [] when the host does not tell."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    out = []
    try:
        for node in sorted(os.listdir(base), key=int):
            props = {}
            with open(os.path.join(base, node, "properties"), encoding="ascii") as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) == 0:
                continue        # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node", encoding="ascii") as f:
                out.append(int(f.read().strip()))
    except (OSError, ValueError):
        return []
    return out


def affinity_plan(local_rank: int, local_world: int, allowed: Sequence[int] = None) -> Tuple[List[int], str]:
    """(cores, how): the host cores replica `local_rank` of `local_world` should run on — the cores of its GPU's NUMA node,
    split evenly among the replicas that share the node (8 Python processes each feeding one GPU — the host
    side is the only shared resource of the request-sharded path) — and, in words, which rule produced them. When the
    topology is not readable (containers often hide /sys/class/kfd or report numa_node -1) the allowed cores are split
    evenly and contiguously, so eight ranks never share the same cores."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    if local_world <= 1 or not allowed:
        return list(allowed), "single replica on this host: all allowed cores"
    nodes = _gpu_numa_nodes()
    why = "GPU NUMA topology unreadable (/sys/class/kfd)"
    if len(nodes) >= local_world and all(n >= 0 for n in nodes[:local_world]):
        mine = nodes[local_rank]
        peers = [r for r in range(local_world) if nodes[r] == mine]
        try:
            with open(f"/sys/devices/system/node/node{mine}/cpulist", encoding="ascii") as f:
                node_cpus = [c for c in _parse_cpulist(f.read()) if c in set(allowed)]
        except OSError:
            node_cpus = []
        if len(node_cpus) >= len(peers):
            b, e = shard_bounds(len(node_cpus), peers.index(local_rank), len(peers))
            return node_cpus[b:e], f"NUMA node {mine} of this rank's GPU, shared by {len(peers)} replica(s)"
        why = f"NUMA node {mine} has fewer allowed cores than replicas"
    elif nodes:
        why = f"GPU NUMA nodes reported as {nodes[:local_world]}"
    b, e = shard_bounds(len(allowed), local_rank, local_world)
    return (allowed[b:e] or list(allowed)), f"even split of the {len(allowed)} allowed cores ({why})"


def cpus_for_local_rank(local_rank: int, local_world: int, allowed: Sequence[int] = None) -> List[int]:
    return affinity_plan(local_rank, local_world, allowed)[0]


last_affinity = dict(cores=None, how="pin_to_local_cores() not called")


def pin_to_local_cores(local_rank: int, local_world: int) -> List[int]:
    """Apply affinity_plan to this process (and cap torch's intra-op threads accordingly). What was done, or why nothing
    was, is left in `last_affinity` for reports (bench.py: config.cpu_affinity)."""
    cpus, how = affinity_plan(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except (OSError, AttributeError) as e:
        last_affinity.update(cores=None, how=f"sched_setaffinity refused ({type(e).__name__}); wanted: {how}")
        return []
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
    last_affinity.update(cores=len(cpus), how=how)
    return cpus


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_job(local_units: float, local_seconds: float) -> Tuple[float, float]:
    """Whole-job totals: (sum of units over ranks, max of seconds over ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(local_units), float(local_seconds)
    units = torch.tensor([float(local_units)], dtype=torch.float64)
    secs = torch.tensor([float(local_seconds)], dtype=torch.float64)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dist.all_reduce(secs, op=dist.ReduceOp.MAX)
    return float(units), float(secs)


def gather_lists(local: list) -> List[list]:
    """Every rank's list, in rank order (e.g. per-replica token streams)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out
