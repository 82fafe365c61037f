"""Multi-GPU rendering: one process per GPU, image rows dealt to ranks in interleaved bands,
one RCCL collective at the end.

The path shards perfectly: a pixel's RNG stream depends only on (v_position, u_time)
(static/shader.frag:354-357), never on which tile, workgroup or GPU traced it, so any row
partition reproduces the single-GPU image bit for bit.  Rank r of n owns the rows y with
(y // band_rows) % n == r (PtParams.band_*): interleaving balances cheap sky rows against
expensive ground rows.  There is no data-path collective while rendering; the only exchange is
the gather of the per-rank fp32 radiance buffers once the frame has converged — an all_gather of
equal-sized padded buffers over RCCL/xGMI (backend "nccl"), de-interleaved on every rank.
"""
import numpy as np

from . import abi


def band_of(rank, world, band_rows=8):
    """(band_rows, band_index, band_count) for this rank."""
    return (int(band_rows), int(rank), int(world))


def max_local_rows(height, band_rows, world):
    return max(abi.local_rows(height, band_rows, r, world) for r in range(world))


def band_layout(height, band_rows, world):
    """(pad_rows, perm): every rank sends a buffer of pad_rows rows (its own rows first, zeros behind: the ranks' shares
    differ by a band when the bands do not divide evenly — 1080 rows in 4-row bands over 8 ranks are 34 bands for six ranks and 33 for two),
    the collective concatenates them in rank order, and image row y is row perm[y] of that concatenation."""
    pad_rows = max_local_rows(height, band_rows, world)
    perm = np.zeros(height, dtype=np.int64)
    for r in range(world):
        ys = np.asarray(abi.owned_rows(height, band_rows, r, world), dtype=np.int64)
        perm[ys] = r * pad_rows + np.arange(len(ys), dtype=np.int64)
    return pad_rows, perm


def assemble_rows(parts, height, band_rows):
    """The gather for ONE process that drives every rank's context itself (one pt_ctx per GPU, as examples/render_bands.c does
    from plain C; also how an N-rank run is rehearsed on one device): parts[r] is rank r's (local_rows_or_more, width, 4)
    tensor; they are laid out exactly as all_gather_into_tensor lays out the ranks' padded buffers (band_layout) and
    de-interleaved by the same permutation.  Returns the (height, width, 4) frame on parts[0]'s device."""
    import torch

    world = len(parts)
    pad_rows, perm = band_layout(height, band_rows, world)
    width = parts[0].shape[1]
    recv = torch.zeros((world * pad_rows, width, 4), dtype=parts[0].dtype, device=parts[0].device)
    for r, part in enumerate(parts):
        rows = abi.local_rows(height, band_rows, r, world)
        recv[r * pad_rows: r * pad_rows + rows].copy_(part[:rows])
    return recv.index_select(0, torch.from_numpy(perm).to(recv.device))


def gather_rows(local, height, band_rows, rank, world, group=None):
    """all_gather the per-rank row bands and reassemble the full image.

    local: torch tensor (local_rows_or_more, width, 4) float32 on the rank's device (cuda for
    nccl, cpu for gloo).  Returns a (height, width, 4) tensor on the same device, identical on
    every rank.

    Stream-ordered and NOT re-entrant: the padded send buffer and the receive buffer are cached per
    (image, partition, device, process group) and reused by the next call, so a second gather with
    the same key must not be in flight on another stream or thread while this one runs (bench.py and
    the tests call it from one thread on the current stream).  The returned tensor is a fresh copy."""
    import torch
    import torch.distributed as dist

    width = local.shape[1]
    rows_here = abi.local_rows(height, band_rows, rank, world)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local[:rows_here].clone()
    # buffers and the de-interleave permutation are built once per (image, partition, device):
    # inside a timed frame the gather is one copy, one collective and one index_select
    key = (int(height), int(width), int(band_rows), int(rank), int(world), str(local.device), local.dtype,
           id(group) if group is not None else 0)
    ent = _GATHER_CACHE.get(key)
    if ent is None:
        pad_rows, perm = band_layout(height, band_rows, world)
        ent = (
            pad_rows,
            torch.from_numpy(perm).to(local.device),
            torch.zeros((pad_rows, width, 4), dtype=local.dtype, device=local.device),  # rows beyond rows_here stay 0
            torch.empty((world * pad_rows, width, 4), dtype=local.dtype, device=local.device),
        )
        _GATHER_CACHE.clear()  # one partition at a time: do not hoard device memory
        _GATHER_CACHE[key] = ent
    pad_rows, perm, send, recv = ent
    send[:rows_here].copy_(local[:rows_here])
    # concatenated along dim 0: the layout every backend's all_gather_into_tensor accepts
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.index_select(0, perm)


_GATHER_CACHE = {}


def render_band(scene, rank, world, band_rows=8, render_fn=None, device=0):
    """Render this rank's rows of `scene` (all passes).  Returns (tensor, segments, tracer).

    render_fn(scene, params) -> (ndarray (local_rows, width, 4), segments) replaces the HIP path
    in CPU-only tests of the partition/gather logic (tests inject the oracle there); the product
    path (render_fn None) always goes through libptrace on `device` and fails without a GPU."""
    import torch

    p = scene.params.copy()
    p.band_rows, p.band_index, p.band_count = band_of(rank, world, band_rows)
    if render_fn is not None:
        acc, seg = render_fn(scene, p)
        return torch.from_numpy(np.ascontiguousarray(acc)), seg, None
    from .tracer import PathTracer

    pt = PathTracer(p.width, p.height, device=device, use_torch=True)
    pt.set_spheres(scene.spheres)
    pt.set_params(p)
    pt.reserve_passes(scene.n_passes)
    pt.render_passes(scene.n_passes)
    return pt.accum_tensor, None, pt
