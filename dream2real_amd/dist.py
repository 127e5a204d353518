"""Pose-shard data parallelism for the render-and-score path: one process per GPU, candidate
poses split in contiguous blocks, ONE all-gather of the fp32 logits (RCCL over xGMI on GPUs,
gloo in the CPU tests), after which ratio / smoothing / argmax run identically on every rank.

The reference is single-GPU (README.md:27); candidates are independent through render,
composite and CLIP (reference combined_rendering.py:118-155, clip_scoring.py:175-183), and
only spatially_smooth_heatmap (geometry_utils.py:252-269) needs neighbouring grid cells —
hence gather first, smooth after."""
from __future__ import annotations

import os
import socket

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def process_rank() -> int:
    """Rank of this process in the torch.distributed group (the launcher's RANK before the group exists; 0 outside one)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except ImportError:
        pass
    return int(os.environ.get("RANK", "0"))


def init_from_env(backend: str | None = None):
    """torch.distributed process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, local_rank); a no-op (0, 1, 0) outside a launcher."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("D2R_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            # "nccl" is RCCL on ROCm; gloo is for CPU tests and for several ranks sharing one GPU
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allgather_logits(local_logits, n_total: int, rank: int, world: int):
    """local_logits: torch tensor [n_local, C] (cuda for nccl, cpu for gloo) -> [n_total, C] on
    every rank.  Shards may be ragged by one row: rows are padded to the largest shard for the
    single all_gather_into_tensor and the padding dropped afterwards."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_logits
    C = local_logits.shape[1]
    n_max = -(-n_total // world)
    if dist.get_backend() == "gloo" and local_logits.is_cuda:
        local_logits = local_logits.cpu()          # gloo moves host memory
    pad = torch.zeros((n_max, C), dtype=local_logits.dtype, device=local_logits.device)
    pad[: local_logits.shape[0]] = local_logits
    out = torch.empty((world * n_max, C), dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r * n_max: r * n_max + (hi - lo)])
    return torch.cat(parts, 0)


def _machine_id() -> str:
    """Names the MACHINE (not the container): the kernel's boot id is shared by every container on a host and differs
    between hosts, whatever their hostnames are."""
    try:
        return open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        return socket.gethostname()


LAST_COMM_REPORT: dict = {}      # how the last init_comm decided: path taken, why, what every rank held


def describe_environment(device_index=None) -> dict:
    """What a failed N > 1 run needs in its log: who this rank is, which GPU it holds, the runtime versions and every
    HSA_* / NCCL_* / RCCL_* / HIP_* / ROCR_* / MASTER_* variable of its environment."""
    import torch
    out = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "world": int(os.environ.get("WORLD_SIZE", "1")), "host": socket.gethostname(), "machine": _machine_id(),
           "torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "device_count": torch.cuda.device_count()}
    try:
        out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        out["rccl_version"] = None
    if device_index is not None and torch.cuda.is_available():
        try:
            out["device_identity"] = _device_identity(device_index)
            out["device_name"] = torch.cuda.get_device_name(device_index)
        except Exception as e:
            out["device_identity"] = f"unavailable: {e}"
    out["env"] = {k: v for k, v in sorted(os.environ.items())
                  if k.startswith(("HSA_", "NCCL_", "RCCL_", "HIP_", "ROCR_", "MASTER_", "D2R_", "GPU_", "CUDA_VISIBLE"))}
    return out


def _device_identity(index: int) -> str:
    """Something that names the physical GPU behind torch device `index` whatever HIP_VISIBLE_DEVICES maps it to:
    every identifying property the runtime reports (UUID, PCI domain / bus / device), concatenated — a runtime that
    leaves one of them empty or equal on all devices is still told apart by the others; with none of them, the
    visible-devices string + index."""
    import torch
    props = torch.cuda.get_device_properties(index)
    parts = []
    for attr in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):
        v = getattr(props, attr, None)
        if v not in (None, ""):
            parts.append(f"{attr}={v}")
    if not parts:
        vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "all")))
        parts.append(f"visible={vis}:{index}")
    return ";".join(parts)


def init_comm(ctx, rank: int, world: int) -> bool:
    """Bring up the C-ABI communicator of `ctx` (d2r_comm_init: RCCL over xGMI): rank 0 draws the id
    blob, torch.distributed (whatever backend the process group has) hands it to the other ranks.
    Returns True when d2r_allgather_scores is usable; False when RCCL cannot be (several ranks
    sharing one GPU in a smoke run, no RCCL installed) — every rank takes the same branch, and the
    caller then gathers through torch.distributed instead."""
    import torch
    import torch.distributed as dist
    from . import _lib
    if world == 1:
        ctx.comm_init(None, 0, 1)
        return True
    # all ranks must agree before anyone enters ncclCommInitRank (it blocks on the others).  RCCL refuses two ranks
    # on one physical GPU; whether that is the case is decided from what the ranks actually hold — (host, device
    # identity) pairs gathered over the process group — not from device_count(), which is 1 on every rank when a
    # launcher gives each rank its own HIP_VISIBLE_DEVICES and says nothing about other nodes
    # (the machine is named by its boot id: containers of one host have different hostnames but share it, and two
    # hosts with equal hostnames do not)
    if dist.get_backend() == "nccl" and ctx is not None:
        torch.cuda.set_device(ctx.device)           # object collectives on the nccl backend stage through the current device
    mine = (_machine_id(), _device_identity(ctx.device) if ctx is not None and torch.cuda.is_available() else "no-gpu")
    held = [None] * world
    dist.all_gather_object(held, mine)
    shared = len(set(held)) < world
    report = {"world": world, "held": [list(h) for h in held], "shared_gpu": shared, "path": None, "reason": None, "error": None}
    LAST_COMM_REPORT.clear()
    LAST_COMM_REPORT.update(report)

    def decided(path, reason, error=None):
        LAST_COMM_REPORT.update(path=path, reason=reason, error=error)
        if rank == 0 or error:
            print(f"[d2r dist] rank {rank}/{world}: gather path = {path} ({reason})" + (f"; error: {error}" if error else ""),
                  file=__import__("sys").stderr, flush=True)
        return path == "rccl"

    blob, err = [None], None
    if rank == 0 and not shared:
        try:
            blob[0] = ctx.comm_unique_id()
        except _lib.D2RError as e:
            blob[0], err = None, str(e)
    dist.broadcast_object_list(blob, src=0)
    if blob[0] is None:
        return decided("torch-gather", "several ranks hold one physical GPU (RCCL refuses that)" if shared
                       else "rank 0 could not draw an RCCL id", err)
    ok, err = 1, None
    try:
        ctx.comm_init(blob[0], rank, world)
    except _lib.D2RError as e:
        ok, err = 0, str(e)
    t = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) == 0:
        ctx.comm_destroy()
        return decided("torch-gather", "ncclCommInitRank failed on at least one rank", err)
    return decided("rccl", "d2r_comm_init: ncclCommInitRank succeeded on every rank")


class ShardGather:
    """Device buffers of the one collective: every rank writes its logits into `local`
    ([n_max, C], zero padded when its shard is one short), `gather()` returns [n_total, C] on the
    host in shard order.  With a C-ABI communicator the exchange is d2r_allgather_scores on the
    context's stream (no torch op on the data path); otherwise torch.distributed moves it."""

    def __init__(self, ctx, n_total: int, n_caps: int, rank: int, world: int, device, use_c_abi: bool):
        import torch
        self.ctx, self.rank, self.world, self.n_total, self.C = ctx, rank, world, n_total, n_caps
        self.n_max = -(-n_total // world)
        self.lo, self.hi = shard_range(n_total, rank, world)
        self.local = torch.zeros((self.n_max, n_caps), dtype=torch.float32, device=device)
        self.full = torch.zeros((world, self.n_max, n_caps), dtype=torch.float32, device=device) if world > 1 else None
        self.use_c_abi = use_c_abi
        sizes = [shard_range(n_total, r, world) for r in range(world)]
        self._rows = np.concatenate([r * self.n_max + np.arange(b - a) for r, (a, b) in enumerate(sizes)]) if world > 1 else None

    def _sync(self):
        if self.ctx is not None:          # CPU tests drive the torch fallback without a context
            self.ctx.synchronize()

    def gather(self) -> np.ndarray:
        """Blocking.  The producer of `local` (d2r_render_score) and the all-gather run on the CONTEXT's stream, the
        device -> host copies below on torch's current stream: the context is synchronised first, so the result
        does not depend on the caller having made the two streams the same one."""
        import torch.distributed as dist
        if self.world == 1:
            self._sync()
            return self.local[: self.n_total].cpu().numpy()
        if self.use_c_abi:
            self.ctx.allgather_scores(self.local.data_ptr(), self.n_max * self.C, self.full.data_ptr())
            self._sync()
            out = self.full.cpu().numpy()
        else:
            self._sync()
            loc = self.local.cpu() if dist.get_backend() == "gloo" else self.local
            full = self.full.cpu() if dist.get_backend() == "gloo" else self.full
            dist.all_gather_into_tensor(full.view(-1, self.C), loc)
            out = full.cpu().numpy()
        return out.reshape(-1, self.C)[self._rows]


class PoseShard:
    """This process' share of the valid poses inside optimise_pose_grid: `range(K)` = its contiguous block, `gather`
    = the one collective (logits of every block, in pose order, on every rank), `barrier` = the process-group barrier.
    `from_env` returns None unless a multi-rank process group exists (or can be created from the launcher's
    environment); the C-ABI communicator (RCCL) of the context is brought up once and kept on it."""

    def __init__(self, ctx, rank: int, world: int, use_c_abi: bool, device=None):
        self.ctx, self.rank, self.world, self.use_c_abi, self.device = ctx, rank, world, use_c_abi, device

    @classmethod
    def from_env(cls, ctx):
        import torch
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not (dist.is_available() and dist.is_initialized()):
            return None
        rank, world, _ = init_from_env()
        if dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        if world <= 1:
            return None
        have_gpu = ctx is not None and torch.cuda.is_available()
        if have_gpu:
            torch.cuda.set_device(ctx.device)
        state = getattr(ctx, "_pose_shard_comm", None) if ctx is not None else None
        if state is None:
            state = init_comm(ctx, rank, world) if have_gpu else False
            if ctx is not None:
                ctx._pose_shard_comm = state
        return cls(ctx, rank, world, bool(state), torch.device("cuda", ctx.device) if have_gpu else torch.device("cpu"))

    def range(self, n_items: int):
        return shard_range(n_items, self.rank, self.world)

    def barrier(self):
        import torch.distributed as dist
        dist.barrier()

    def gather(self, local_logits, n_total: int) -> np.ndarray:
        import torch
        loc = np.ascontiguousarray(local_logits, np.float32)
        g = ShardGather(self.ctx, n_total, loc.shape[1], self.rank, self.world, self.device, self.use_c_abi)
        g.local[: loc.shape[0]] = torch.from_numpy(loc).to(self.device)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)       # the upload ran on torch's stream, the gather runs on the context's
        return g.gather()


def score_sharded(pose_batch: np.ndarray, score_fn, sample_res, has_norm: bool, n_goal: int = 1,
                  smoothing: bool = True, rank: int = 0, world: int = 1, is_valid=None):
    """The multi-GPU form of optimise_pose_grid's scoring half.

    score_fn(poses [k,16]) -> torch tensor [k, C] of logits for this rank's block.
    Returns (best_pose_idx, pose_scores [N]) — identical on every rank."""
    import torch
    from .clip_scoring import reduce_logits
    from .geometry_utils import spatially_smooth_heatmap
    N = pose_batch.shape[0]
    valid = np.ones(N, bool) if is_valid is None else np.asarray(is_valid, bool)
    valid_idxs = np.nonzero(valid)[0]
    K = len(valid_idxs)
    if K == 0:
        raise Exception
    lo, hi = shard_range(K, rank, world)
    local = score_fn(pose_batch[valid_idxs[lo:hi]])
    full = allgather_logits(local, K, rank, world)
    logits = reduce_logits(full.detach().cpu().numpy(), n_goal, has_norm)
    scores = np.zeros(N, np.float32)
    scores[valid_idxs] = logits
    if smoothing:
        scores = spatially_smooth_heatmap(scores, sample_res)
    return int(np.argmax(scores)), scores
