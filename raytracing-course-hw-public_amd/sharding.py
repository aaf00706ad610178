"""Image sharding across GPUs: interleaved pixel blocks + one gather of the framebuffer (linear float3,
or rgb8 when the film ran on the device: rt_render_rgb8, a 4x smaller message).

The reference self-schedules disjoint 256-pixel spans over CPU threads (raytracer.h:640-659); pixels are independent
given the read-only scene. Here the scene is replicated per GPU and the image is split into blocks of
`shard_block` row-major pixels, block b belonging to rank b % world (interleaved, because cost per row is very
uneven). There is no collective on the data path; the only exchange is one gather of each rank's blocks to rank 0
after the render (RCCL over xGMI when the tensors live on GPUs, gloo in the CPU tests): 12 bytes per pixel,
12 MB for 1000x1000.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def n_blocks(n_pix: int, block: int) -> int:
    return (n_pix + block - 1) // block


def shard_blocks(n_pix: int, block: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_blocks(n_pix, block), world))


def shard_pixels(n_pix: int, block: int, rank: int, world: int) -> int:
    return sum(min((b + 1) * block, n_pix) - b * block for b in shard_blocks(n_pix, block, rank, world))


class FramebufferGather:
    """Reusable buffers + the gather itself. `fb` is the rank's full-size flat framebuffer (n_pix*3 floats) in which
    only this rank's blocks are valid (that is what rt_render writes with shard_count = world)."""

    def __init__(self, n_pix: int, block: int, rank: int, world: int, device: torch.device, dtype: torch.dtype = torch.float32, all_gather: bool = False):
        """`all_gather`: exchange with all_gather_into_tensor instead of gather (same bytes per link on a ring; every rank ends up with the slabs)."""
        self.n_pix, self.block, self.rank, self.world = n_pix, block, rank, world
        self.nb = n_blocks(n_pix, block)
        self.max_blocks = (self.nb + world - 1) // world
        self.padded = torch.zeros(self.nb * block * 3, dtype=dtype, device=device)
        self.slab = torch.zeros(self.max_blocks * block * 3, dtype=dtype, device=device)
        self.gathered = [torch.empty_like(self.slab) for _ in range(world)] if (world > 1 and rank == 0) else None
        self.full = torch.zeros(self.nb * block * 3, dtype=dtype, device=device) if rank == 0 else None
        self.use_all_gather = bool(all_gather)
        self.all_slabs = None

    def gather(self, fb: torch.Tensor) -> Optional[torch.Tensor]:
        """Returns the assembled framebuffer (n_pix*3) on rank 0, None elsewhere."""
        if self.world == 1:
            return fb
        self.padded[: self.n_pix * 3] = fb
        mine = self.padded.view(self.nb, self.block * 3)[self.rank :: self.world]
        self.slab[: mine.numel()] = mine.reshape(-1)
        if self.use_all_gather:
            if self.all_slabs is None:
                self.all_slabs = torch.empty(self.world * self.slab.numel(), dtype=self.slab.dtype, device=self.slab.device)
            dist.all_gather_into_tensor(self.all_slabs, self.slab)
            if self.rank != 0:
                return None
            parts = list(self.all_slabs.view(self.world, -1).unbind(0))
        else:
            dist.gather(self.slab, self.gathered, dst=0)
            if self.rank != 0:
                return None
            parts = self.gathered
        fv = self.full.view(self.nb, self.block * 3)
        for r in range(self.world):
            k = len(range(r, self.nb, self.world))
            fv[r :: self.world] = parts[r][: k * self.block * 3].view(k, self.block * 3)
        return self.full[: self.n_pix * 3]
