"""Static sharding of the bodies over ranks: rank g owns the contiguous targets
[g*shard_len, (g+1)*shard_len), shard_len = ceil(N/G); every rank keeps a full-length fp32 copy of
the sources laid out chunk-major  pos_all[G][dim][shard_pad]  so that one all-gather of each
rank's own chunk refreshes it.  Pure integer logic, mirrored by nbx_ctx_create in csrc/nbx_api.hip."""
from __future__ import annotations

from dataclasses import dataclass

PAD_QUANTUM = 4096  # csrc/nbx_internal.h kPadQuantum


@dataclass(frozen=True)
class ShardLayout:
    n_total: int
    n_shards: int
    shard: int
    dim: int

    def __post_init__(self):
        if self.dim not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        if self.n_shards < 1 or not (0 <= self.shard < self.n_shards):
            raise ValueError("bad shard / n_shards")
        if self.n_total < 0:
            raise ValueError("n_total < 0")

    @property
    def shard_len(self) -> int:
        return -(-self.n_total // self.n_shards)

    @property
    def shard_pad(self) -> int:
        return max(PAD_QUANTUM, -(-self.shard_len // PAD_QUANTUM) * PAD_QUANTUM)

    def bounds(self, shard: int | None = None):
        """[lo, hi) global body indices owned by `shard` (default: this rank's)."""
        g = self.shard if shard is None else shard
        lo = min(self.n_total, g * self.shard_len)
        return lo, min(self.n_total, lo + self.shard_len)

    @property
    def count(self) -> int:
        lo, hi = self.bounds()
        return hi - lo

    def pos_all_shape(self):
        return (self.n_shards, self.dim, self.shard_pad)

    def mass_all_shape(self):
        return (self.n_shards, self.shard_pad)

    def interactions_per_step(self) -> int:
        """Ordered pairs evaluated by this shard per force evaluation: own targets x all N sources."""
        return self.count * self.n_total
