"""Column sharding of one proof phase's commitments across ranks (one process per GPU).

The reference commits the m polynomials of a proof phase one after another against the same bases
(create_proof, reached from /root/reference/circuits/src/utils.rs:83-91,105-120).
They are independent, so column j goes to rank j mod world; every rank runs its MSMs on its own GPU
(bases are replicated at registration, scalars never cross GPUs) and ONE all-gather of the m x 64-byte
affine commitments over RCCL/xGMI gives every rank the full vector in column order for the transcript.
There is no other collective on this path.
"""
import numpy as np

# tests set this to run the gather even in a one-rank group (the RCCL code path on a one-GPU box)
FORCE_GATHER = False


def shard_columns(m, rank, world):
    """indices of the columns rank `rank` commits to"""
    return list(range(rank, m, world))


def gather_columns(local, m, group=None):
    """`local`: this rank's results as a (len(shard_columns(m, rank, world)), w) int64 tensor, rows in shard order.
    Returns the (m, w) tensor in column order on every rank -- the one collective of a commit phase
    (`all_gather_into_tensor`; every rank sends ceil(m / world) rows, short shards padded with zeros).
    RCCL gathers device tensors in place; any other backend (gloo in the tests) goes through host memory."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    slots = (m + world - 1) // world
    w = local.shape[1]
    on_device = dist.get_backend(group) == "nccl"
    dev = local.device if on_device else torch.device("cpu")
    send = torch.zeros((slots, w), dtype=torch.int64, device=dev)
    send[:local.shape[0]] = local.to(dev)
    recv = torch.empty((world * slots, w), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    # rank r's row i is column r + i * world
    cols = torch.arange(m)
    order = (cols % world) * slots + cols // world
    return recv[order.to(dev)].to(local.device)


def commit_columns(bases, columns, group=None, msm_batch=None):
    """Commit to `columns` (list of (n, 4) uint64 arrays, identical on every rank) with the columns sharded
    over the ranks of `group`; returns the (m, 8) affine commitments in column order on every rank.

    `msm_batch` defaults to the GPU path `bases.msm_batch`; tests on CPU (gloo) inject the oracle here --
    the product itself has no CPU path.
    """
    import torch
    import torch.distributed as dist

    m = len(columns)
    if msm_batch is None:
        msm_batch = bases.msm_batch
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(msm_batch(columns), dtype=np.uint64).reshape(m, 8)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = shard_columns(m, rank, world)
    local = np.asarray(msm_batch([columns[j] for j in mine]), dtype=np.uint64).reshape(len(mine), 8)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    out = gather_columns(torch.from_numpy(local.view(np.int64)).to(dev), m, group)
    return out.cpu().numpy().view(np.uint64).reshape(m, 8)


def phase_mode(m, world):
    """how a commit phase of m columns is spread over `world` ranks: whole columns (column j -> rank j mod world) when
    that balances exactly, else every rank takes a contiguous point range of EVERY column (SURVEY.md section 8(e):
    real proofs have m = 1..5 per phase, fewer than the GPUs of a node)"""
    if world <= 1:
        return "single"
    return "columns" if m % world == 0 else "range"


def all_gather_rows(local, group=None):
    """(rows, w) int64 tensor -> (world, rows, w) on every rank.  RCCL gathers device tensors in place; any other
    backend (gloo in the tests and on boxes with fewer GPUs than ranks) goes through host memory."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rows, w = local.shape
    if dist.get_backend(group) == "nccl":
        recv = torch.empty((world * rows, w), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, local.contiguous(), group=group)
        return recv.view(world, rows, w)
    host = local.cpu().contiguous()
    recv = torch.empty((world * rows, w), dtype=host.dtype)
    dist.all_gather_into_tensor(recv, host, group=group)
    return recv.view(world, rows, w).to(local.device)


def _run(bases, d_cols, first, n, stride, cols, d_out, stream):
    """`cols` = the column indices (of the phase) in this launch; `bases` one Bases for all columns or a list per column"""
    if isinstance(bases, (list, tuple)):
        from .api import msm_device_multi
        msm_device_multi([bases[j] for j in cols], d_cols, first, n, stride, d_out, stream)
    else:
        bases.msm_device_range(d_cols, first, n, stride, len(cols), d_out, stream)


def msm_phase_device(bases, d_cols, n, m, stream=None, group=None, mode=None, device=None):
    """The m commitments of one proof phase on the GPUs of `group`: d_cols is a device pointer to m columns of n
    scalars (stride n), identical on every rank; returns an (m, 12) int64 CUDA tensor of Jacobian points, the same
    group elements on every rank, in column order.  `bases`: one Bases object, or a list of m (column j commits against
    bases[j]: commitments over different SRS vectors that do not wait for each other share the launch).

    "columns": rank r runs columns r, r + world, ... whole, one all-gather of the 96-byte results.
    "range":   rank r runs bases / rows [n r / world, n (r+1) / world) of every column in one batched launch
               (h2_msm_device_range), the world x m partial sums are all-gathered and added on the device
               (h2_points_sum_device) -- BASELINE config 4's split of a single MSM, applied to every column.
    Either way ONE collective per phase, m * 96 bytes per rank; scalars and bases never cross GPUs.
    `device` defaults to the current CUDA device (the CPU tests of the sharding logic pass "cpu" and a stand-in
    `bases` object).

    Stream contract: everything this function does -- its tensors, the library's launches, the all-gather -- is ordered
    on torch's CURRENT stream of that device.  `stream` is the raw handle of that stream (default: looked up); handing in
    any other stream is refused, because the zero-fill of the local buffer, the collective and the copy-out are torch
    operations on the current stream and would race with launches made elsewhere."""
    import torch
    import torch.distributed as dist
    if device is None or torch.device(device).type == "cuda":
        current = torch.cuda.current_stream(torch.device(device) if device is not None else None).cuda_stream
        if stream is None:
            stream = current
        elif int(stream) != int(current):
            raise ValueError("msm_phase_device: `stream` must be torch's current stream (run inside "
                             "`with torch.cuda.stream(s)` and pass s.cuda_stream, or pass nothing)")
    elif stream is None:
        stream = 0
    have_group = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if have_group else 1
    rank = dist.get_rank(group) if have_group else 0
    mode = mode or phase_mode(m, world)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if mode == "single" or not have_group:
        out = torch.empty((m, 12), dtype=torch.int64, device=dev)
        _run(bases, d_cols, 0, n, n, list(range(m)), out.data_ptr(), stream)
        return out
    if mode == "columns":
        mine = len(range(rank, m, world))
        slots = (m + world - 1) // world
        local = torch.zeros((slots, 12), dtype=torch.int64, device=dev)
        if mine:
            _run(bases, d_cols + rank * n * 32, 0, n, world * n, list(range(rank, m, world)), local.data_ptr(), stream)
        allr = all_gather_rows(local, group)                    # (world, slots, 12): column r + i * world at [r][i]
        return allr.transpose(0, 1).reshape(slots * world, 12)[:m].contiguous()
    lo, hi = split_msm_by_range(n, rank, world)
    local = torch.zeros((m, 12), dtype=torch.int64, device=dev)
    if hi > lo:
        _run(bases, d_cols + lo * 32, lo, hi - lo, n, list(range(m)), local.data_ptr(), stream)
    allr = all_gather_rows(local, group)                        # (world, m, 12)
    out = torch.empty((m, 12), dtype=torch.int64, device=dev)
    (bases[0] if isinstance(bases, (list, tuple)) else bases).points_sum_device(allr.data_ptr(), world, m, out.data_ptr(), stream)
    return out


def split_msm_by_range(n, rank, world):
    """contiguous point range of a single large MSM handled by `rank`:
    each rank returns one partial sum, the partials are all-gathered and added (the group is abelian)."""
    lo = n * rank // world
    hi = n * (rank + 1) // world
    return lo, hi
