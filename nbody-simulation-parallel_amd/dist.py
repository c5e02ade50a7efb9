"""One process per GPU: sharded all-pairs stepping over torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for the CPU rehearsal in tests/).

The reference is single-process (no MPI/NCCL anywhere); this layer is new.  Per step and
per rank:

    comm stream   : all-gather of every rank's freshly drifted fp32 position chunk  (12 B/body)
    compute stream: force kernel over the rank's OWN chunk of sources  -- needs no remote data,
                    so it runs while the collective is in flight --, then (after the collective)
                    the force kernel over the other ranks' chunks, accumulated on top, then the
                    fused fp64 kick+drift, which rewrites this rank's chunk for the next exchange.

Masses never change, so they are distributed once, at upload: every rank moves only ITS OWN shard over its host link
(nbx_ctx_upload_shard), one all-gather brings the other ranks' masses, the step's own exchange the positions, and the
preconditions of the fast force path (largest mass / coordinate over ALL bodies) are combined with one all-reduce
(nbx_ctx_upload_finish) -- the protocol of the single-process node layer (csrc/nbx_node.hip).
The compute back end is an object with the five methods of `HipShardBackend`; the product back end
calls the HIP library through its C ABI and has no CPU fallback.  tests/ substitutes a numpy double
to rehearse the orchestration over gloo.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .sharding import ShardLayout


class HipShardBackend:
    """Shard compute on one MI355X: kernels via libnbody_hip.so, buffers and streams via torch."""

    def __init__(self, bodies: np.ndarray, layout: ShardLayout, device_index: int, variant: int = -1,
                 source_splits: int = 0, refine_tol: Optional[float] = None):
        """Buffers, streams and the context; the bodies go to the device in upload() (make_hip_system calls it: a collective
        for n_shards > 1).  `bodies` is only looked at for its shape here."""
        if not torch.cuda.is_available():
            raise capi.NbxError(capi.NBX_ERR_NO_DEVICE, "HipShardBackend", "no GPU visible to torch; there is no CPU fallback")
        self.layout = layout
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(layout.n_total, layout.dim, device=device_index, n_shards=layout.n_shards, shard=layout.shard)
        assert self.ctx.shard_pad == layout.shard_pad and self.ctx.shard_len == layout.shard_len
        # exchange buffers live in torch memory so RCCL (through torch.distributed) and the kernels share them
        self.pos_all = torch.zeros(layout.pos_all_shape(), dtype=torch.float32, device=self.device)
        self.mass_all = torch.zeros(layout.mass_all_shape(), dtype=torch.float32, device=self.device)
        self.ctx.set_gather_buffers(self.pos_all.data_ptr(), self.mass_all.data_ptr())
        self.compute_stream = torch.cuda.Stream(device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device)
        self.ctx.set_stream(self.compute_stream.cuda_stream)
        self.ctx.set_tuning(source_splits, variant)
        if refine_tol is not None:   # None: the library default (mixed mode, 1e-5); 0: plain fp32
            self.ctx.set_refine(refine_tol)   # mixed mode: the suspects of every evaluation re-evaluated in fp64 after the REMOTE pass
        torch.cuda.synchronize(self.device)  # zero fills done before the library's stream writes
        self._timing = False
        self._marks = []       # per step: dict of torch events (see enable_timing)
        self._cur = None
        # RCCL all-gather in place (send buffer = this rank's chunk of the receive buffer: NCCL's in-place form).  Should a
        # torch / RCCL build reject or mishandle the aliasing, verify_exchange switches to a separate send buffer.
        self.inplace_gather = True
        self._send_buf = None

    # -- upload: own shard over the host link, the rest device to device --
    def upload(self, bodies: np.ndarray, group=None):
        """COLLECTIVE for n_shards > 1 (every rank calls it with the same full array, or at least with its own rows right):
        this rank copies rows [lo, hi) to its device, the masses and positions of the other shards arrive through one all-gather
        each, the maxima that decide the fast path's preconditions through one all-reduce.  upload_bytes: what crossed this
        rank's host link."""
        lo, hi = self.layout.bounds()
        if self.layout.n_shards == 1:
            self.ctx.upload(bodies)
            self.upload_bytes = int(bodies.shape[0] * bodies.shape[1] * 8)
            return
        mine = np.ascontiguousarray(bodies[lo:hi])
        m, x = self.ctx.upload_shard(mine)
        self.upload_bytes = int(mine.size * 8)
        self.ctx.synchronize()
        self._gather_chunks_blocking(self.mass_all, group)          # masses: once per upload
        self.finish_exchange(self.start_exchange(group))            # positions: the exchange every step uses
        # a NaN must not get lost in a max-reduction whose NaN handling is the transport's business: +inf fails the same tests
        t = torch.tensor([m if m == m else float("inf"), x if x == x else float("inf")], dtype=torch.float64)
        if dist.get_backend(group) == "nccl":
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self.compute_stream.synchronize()
        self.ctx.upload_finish(float(t[0].item()), float(t[1].item()))

    def _gather_chunks_blocking(self, buf: torch.Tensor, group):
        """All-gather of buf[shard] into buf (chunk-major, every rank's chunk the same size), finished when this returns."""
        flat = buf.view(self.layout.n_shards, -1)
        if dist.get_backend(group) != "nccl":      # rehearsal transport: through host memory
            mine = flat[self.layout.shard].to("cpu")
            allc = torch.empty((self.layout.n_shards, mine.numel()), dtype=buf.dtype)
            dist.all_gather_into_tensor(allc.view(-1), mine, group=group)
            for g in range(self.layout.n_shards):
                if g != self.layout.shard:
                    flat[g].copy_(allc[g].to(self.device))
            torch.cuda.synchronize(self.device)
            return
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_stream(self.compute_stream)
            send = flat[self.layout.shard].clone()  # a separate send buffer: nothing to gain from the in-place form once per upload
            dist.all_gather_into_tensor(flat.view(-1), send, group=group)
        self.comm_stream.synchronize()

    # -- pass timing (bench.py --gpus N): events on the streams the work really runs on --
    def enable_timing(self, on: bool = True):
        """Record, per force evaluation, events around the LOCAL pass and the REMOTE pass (compute stream) and
        around the position exchange (comm stream), so that a first run on a multi-GPU node describes itself:
        how long each pass took and whether the collective was hidden behind the LOCAL pass."""
        self._timing = on
        self._marks = []
        self._cur = None

    def _mark(self, key, stream):
        if self._timing and self._cur is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            self._cur[key] = e

    def pass_times(self):
        """Means over the recorded force evaluations, in ms; exchange_hidden = every exchange had finished
        before the LOCAL pass of the same step did."""
        self.synchronize()
        loc, rem, exc, hidden = [], [], [], []
        for m in self._marks:
            if "l0" in m and "l1" in m:
                loc.append(m["l0"].elapsed_time(m["l1"]))
            if "r0" in m and "r1" in m:
                rem.append(m["r0"].elapsed_time(m["r1"]))
            if "x0" in m and "x1" in m:
                exc.append(m["x0"].elapsed_time(m["x1"]))
                if "l0" in m and "l1" in m:
                    hidden.append(m["l0"].elapsed_time(m["x1"]) <= m["l0"].elapsed_time(m["l1"]))
        mean = lambda v: float(sum(v) / len(v)) if v else None
        return {"local_ms": mean(loc), "remote_ms": mean(rem), "exchange_ms": mean(exc), "evaluations": len(self._marks),
                "exchange_hidden": bool(hidden) and all(hidden), "exchange_hidden_count": int(sum(hidden))}

    # -- the five operations the orchestrator needs --
    def accel_local(self):
        self._mark("l0", self.compute_stream)
        self.ctx.compute_accel(capi.SRC_LOCAL)
        self._mark("l1", self.compute_stream)

    def accel_remote(self):
        self._mark("r0", self.compute_stream)
        self.ctx.compute_accel(capi.SRC_REMOTE)
        self._mark("r1", self.compute_stream)

    def kick_drift(self, G: float, dt: float):
        self.ctx.kick_drift(dt, G)

    def kick_drift2(self, G: float, dt_kick: float, dt_drift: float):
        self.ctx.kick_drift2(dt_kick, dt_drift, G)

    def start_exchange(self, group):
        """Launch the position all-gather on the comm stream, ordered after everything already
        queued on the compute stream (the previous drift), and return the work handle."""
        if self._timing:
            self._cur = {}
            self._marks.append(self._cur)
        if self.layout.n_shards == 1:
            return None
        self.comm_stream.wait_stream(self.compute_stream)
        self._mark("x0", self.comm_stream)
        if dist.get_backend(group) != "nccl":
            # Rehearsal transport (gloo has no device collectives): stage the own chunk through host
            # memory.  Same buffers, same ordering; only the wire differs.  Not used on a multi-GPU node.
            with torch.cuda.stream(self.comm_stream):
                mine = self.pos_all[self.layout.shard].to("cpu", non_blocking=False).reshape(-1)
            allc = torch.empty((self.layout.n_shards, mine.numel()), dtype=torch.float32)
            dist.all_gather_into_tensor(allc.view(-1), mine, group=group)
            with torch.cuda.stream(self.comm_stream):
                for g in range(self.layout.n_shards):
                    if g != self.layout.shard:
                        self.pos_all[g].view(-1).copy_(allc[g].to(self.device, non_blocking=False))
            return "staged"
        with torch.cuda.stream(self.comm_stream):
            if self.inplace_gather:
                src = self.pos_all[self.layout.shard].view(-1)
            else:
                if self._send_buf is None:
                    self._send_buf = torch.empty_like(self.pos_all[self.layout.shard]).view(-1)
                self._send_buf.copy_(self.pos_all[self.layout.shard].view(-1))
                src = self._send_buf
            return dist.all_gather_into_tensor(self.pos_all.view(-1), src, group=group, async_op=True)

    def finish_exchange(self, work):
        if work is None:
            return
        if work != "staged":
            with torch.cuda.stream(self.comm_stream):
                work.wait()
        self._mark("x1", self.comm_stream)
        self.compute_stream.wait_stream(self.comm_stream)

    # -- exchange self-check (first contact with a multi-GPU node) --
    def poison_remote_chunks(self):
        """Overwrite every chunk this rank does not own with NaN, so that only a working exchange can
        restore them.  Ordered on the compute stream (the exchange waits for it)."""
        with torch.cuda.stream(self.compute_stream):
            for g in range(self.layout.n_shards):
                if g != self.layout.shard:
                    self.pos_all[g].fill_(float("nan"))

    def remote_chunk_mismatches(self, bodies: np.ndarray) -> int:
        """Number of fp32 position values in the other ranks' chunks that differ from the fp32-rounded
        positions of `bodies` (host compare).  Only meaningful while the bodies have not moved."""
        self.synchronize()
        host = self.pos_all.cpu().numpy()
        return _count_chunk_mismatches(host, bodies, self.layout)

    # -- results --
    def forces(self, G: float) -> np.ndarray:
        return self.ctx.forces(G)

    def energy(self, G: float):
        return self.ctx.energy(G)

    def download_into(self, bodies: np.ndarray):
        self.ctx.download(bodies)

    def synchronize(self):
        self.ctx.synchronize()
        torch.cuda.synchronize(self.device)

    def kernel_time(self):
        return self.ctx.kernel_time()

    def close(self):
        self.ctx.close()


def _count_chunk_mismatches(pos_all_host: np.ndarray, bodies: np.ndarray, layout: ShardLayout) -> int:
    bad = 0
    for g in range(layout.n_shards):
        if g == layout.shard:
            continue
        lo, hi = layout.bounds(g)
        want = np.ascontiguousarray(bodies[lo:hi, :layout.dim].T.astype(np.float32))
        got = pos_all_host[g, :, : hi - lo]
        bad += int((got.view(np.uint32) != want.view(np.uint32)).sum())
    return bad


class ExchangeError(RuntimeError):
    """The position exchange raised on at least one rank: the communicator may be wedged, nothing is retried."""


class ShardedNBody:
    """The stepping loop of one rank.  `backend` supplies the shard compute (see module docstring).
    check_store: a torch.distributed key-value store (bench.py: the rendezvous TCPStore) for the self-check's bookkeeping --
    a transport independent of the collective under test, so that a rank whose collective raised still meets the others.
    Without one the bookkeeping is an all_reduce on `group` itself."""

    def __init__(self, backend, layout: ShardLayout, group=None, check_store=None):
        self.be = backend
        self.layout = layout
        self.group = group
        self.check_store = check_store
        self._check_round = 0
        if layout.n_shards > 1:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for n_shards > 1")
            ws = dist.get_world_size(group)
            if ws != layout.n_shards or dist.get_rank(group) != layout.shard:
                raise ValueError(f"layout ({layout.shard}/{layout.n_shards}) does not match the process group "
                                 f"({dist.get_rank(group)}/{ws})")

    def compute_forces(self):
        """One force evaluation for this rank's targets against all N sources (positions as they
        stand in the exchange buffer after the last drift)."""
        work = self.be.start_exchange(self.group)
        self.be.accel_local()
        self.be.finish_exchange(work)
        if self.layout.n_shards > 1:
            self.be.accel_remote()

    def verify_exchange(self, bodies: np.ndarray) -> int:
        """Self-check of the position exchange, to be called BEFORE the first step (while the device positions
        still equal the uploaded ones): every rank poisons the chunks it does not own, the exchange runs once,
        and each remote chunk is compared on the host with the fp32-rounded positions of `bodies` (identical on
        every rank).  Returns the number of mismatching values summed over all ranks: 0 = the collective
        delivered every chunk to every rank.  A transport that does nothing, or gathers into the wrong
        offsets, cannot pass."""
        if self.layout.n_shards == 1:
            return 0
        total = self._verify_once(bodies)
        if total and getattr(self.be, "inplace_gather", False):
            # the in-place all-gather ran without raising anywhere but did not deliver (every rank sees the same total, so
            # every rank takes this branch): one more try with a separate send buffer before giving up
            self.be.inplace_gather = False
            total = self._verify_once(bodies)
        return total

    def _verify_once(self, bodies: np.ndarray) -> int:
        """One poisoned-buffer exchange.  Every rank runs the same sequence of collectives whatever happens locally: only the
        exchange itself and the host compare sit inside the try; the outcome {mismatches, raised} is then summed over the ranks
        through check_store (or `group`).  If the exchange RAISED anywhere, ExchangeError on every rank -- no retry on a communicator that may be
        wedged (ranks still inside a collective that others abandoned can only be released by the transport's own timeout)."""
        self.be.poison_remote_chunks()
        bad, failed = 0, 0
        try:
            work = self.be.start_exchange(self.group)
            self.be.finish_exchange(work)
            bad = self.be.remote_chunk_mismatches(bodies)
        except RuntimeError as e:
            import sys
            sys.stderr.write(f"[dist] rank {self.layout.shard}: position exchange raised: {e}\n")
            failed = 1
        if self.check_store is not None:
            # every rank posts its outcome under a key of this round and reads everybody's (get blocks until the key exists)
            self._check_round += 1
            base = f"nbx_exchange_check/{self._check_round}/"
            self.check_store.set(base + str(self.layout.shard), f"{bad},{failed}")
            total_bad = total_failed = 0
            for g in range(self.layout.n_shards):
                b, f = self.check_store.get(base + str(g)).decode().split(",")
                total_bad += int(b)
                total_failed += int(f)
        else:
            t = torch.tensor([bad, failed], dtype=torch.int64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            total_bad, total_failed = int(t[0].item()), int(t[1].item())
        if total_failed:
            raise ExchangeError(f"the position exchange raised on {total_failed} of {self.layout.n_shards} ranks")
        return total_bad

    def step(self, dt: float, G: float = capi.REFERENCE_G, nsteps: int = 1):
        for _ in range(nsteps):
            self.compute_forces()
            self.be.kick_drift(G, dt)

    def step_kdk(self, dt: float, G: float = capi.REFERENCE_G, nsteps: int = 1):
        """Extension: synchronised kick-drift-kick leapfrog (second order), adjacent half-kicks merged:
        K(dt/2) D(dt) [F K(dt) D(dt)]^(n-1) F K(dt/2) -- n + 1 force evaluations (each with its exchange) for n steps."""
        if nsteps <= 0:
            return
        self.compute_forces()
        for s in range(nsteps):
            self.be.kick_drift2(G, 0.5 * dt if s == 0 else dt, dt)
            self.compute_forces()
        self.be.kick_drift2(G, 0.5 * dt, 0.0)

    def forces(self, G: float = capi.REFERENCE_G) -> np.ndarray:
        return self.be.forces(G)

    def energy(self, G: float = capi.REFERENCE_G):
        """(kinetic, potential) of the WHOLE system: refresh every rank's source copy, evaluate the
        shard's share on the device, sum over ranks."""
        work = self.be.start_exchange(self.group)
        self.be.finish_exchange(work)
        ke, pe = self.be.energy(G)
        if self.layout.n_shards > 1:
            t = torch.tensor([ke, pe], dtype=torch.float64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            ke, pe = float(t[0]), float(t[1])
        return ke, pe

    def gather_bodies(self, bodies: np.ndarray) -> np.ndarray:
        """Assemble the full Body<D> array on every rank from the ranks' shards (host side, fp64)."""
        out = np.ascontiguousarray(bodies).copy()
        self.be.download_into(out)
        if self.layout.n_shards == 1:
            return out
        lo, hi = self.layout.bounds()
        mine = torch.from_numpy(out[lo:hi].copy())
        width = out.shape[1]
        padded = torch.zeros((self.layout.shard_len, width), dtype=torch.float64)
        padded[: hi - lo] = mine
        use_cuda = dist.get_backend(self.group) == "nccl"
        if use_cuda:
            padded = padded.cuda()  # the current device was set by the back end
        allb = torch.zeros((self.layout.n_shards, self.layout.shard_len, width), dtype=torch.float64, device=padded.device)
        dist.all_gather_into_tensor(allb.view(-1), padded.view(-1), group=self.group)
        allb = allb.cpu().numpy()
        for g in range(self.layout.n_shards):
            glo, ghi = self.layout.bounds(g)
            out[glo:ghi] = allb[g, : ghi - glo]
        return out


def make_hip_system(bodies: np.ndarray, dim: int, rank: int = 0, world_size: int = 1, device_index: Optional[int] = None,
                    group=None, variant: int = -1, source_splits: int = 0, refine_tol: Optional[float] = None, check_store=None) -> ShardedNBody:
    layout = ShardLayout(n_total=bodies.shape[0], n_shards=world_size, shard=rank, dim=dim)
    be = HipShardBackend(bodies, layout, rank if device_index is None else device_index, variant, source_splits, refine_tol)
    system = ShardedNBody(be, layout, group, check_store)
    be.upload(bodies, group)     # collective for world_size > 1: own shard over the host link, the rest device to device
    return system
