"""Multi-GPU driver: groups are sharded over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm), and the ONLY data-path collective is one all-gather per migration event.

Groups never interact inside update!/p_update! (src/main.jl:135-167) -- the reference itself runs one task per
group -- so rank r owns groups [r*G, (r+1)*G) with their particles and history.  migration! (src/migration.jl:11-19)
is split around the exchange:
    pack   : every rank picks one candidate particle per local group   (select_particle, migration.jl:64-70)
    gather : ONE all_gather of [G][D+3] doubles per rank               (slot, theta[D], weight, id)
    apply  : every rank derives the same group subset and circular shift from the shared Philox STEP stream
             (select_groups + shift_particles!, migration.jl:31-35, 84-91) and overwrites its own selected slots.
All randomness is keyed by GLOBAL group / slot indices, so an N-rank run reproduces the 1-rank run bit for bit.
"""
import numpy as np


class ShardedDriver:
    """Drives one engine shard.  `engine` follows the engine interface (HipEngine, or the CPU oracle when a TEST
    injects it); `dist` is torch.distributed (or None for a single shard)."""

    def __init__(self, engine, dist=None, device=None, stream_ordered=False):
        import torch
        self.torch = torch
        self.eng = engine
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = device if device is not None else torch.device("cpu")
        G, D = engine.cfg.n_groups, engine.cfg.D
        Gt = engine.cfg.n_groups_total
        if G * self.world != Gt:
            raise ValueError(f"n_groups_total={Gt} must equal world_size*n_groups={self.world}*{G}")
        self.rows = torch.zeros((G, D + 3), dtype=torch.float64, device=self.device)
        self.all_rows = torch.zeros((Gt, D + 3), dtype=torch.float64, device=self.device)
        self.on_device = self.device.type == "cuda"
        # True: the engine's stream IS torch's current stream (engine.set_stream(torch.cuda.current_stream().cuda_stream))
        self.stream_ordered = bool(stream_ordered) and self.on_device
        self.n_exchanges = 0

    def _exchange(self, it):
        t = self.torch
        if self.on_device:
            # Stream-ordered, no host synchronisation: the engine enqueues on the stream the collective is issued from
            # (`stream_ordered`: the caller gave the engine torch's current stream with set_stream), so
            # pack -> all-gather -> apply -> the next update simply queue behind each other.  torch's process group runs
            # the collective on its own stream, fenced against the current stream by events on both sides.
            if self.stream_ordered:
                self.eng.migration_pack_enqueue(it, self.rows.data_ptr())
            else:
                self.eng.migration_pack_dev(it, self.rows.data_ptr())  # returns after the engine's own stream drained
            if self.dist:
                self.dist.all_gather_into_tensor(self.all_rows, self.rows)  # the one collective (RCCL over xGMI)
            else:
                self.all_rows.copy_(self.rows)
            if self.stream_ordered:
                self.eng.migration_apply_enqueue(it, self.all_rows.data_ptr())
            else:
                t.cuda.current_stream().synchronize()
                self.eng.migration_apply_dev(it, self.all_rows.data_ptr())
        else:
            self.rows.copy_(t.from_numpy(self.eng.migration_pack(it)))
            if self.dist:
                self.dist.all_gather_into_tensor(self.all_rows, self.rows)
            else:
                self.all_rows.copy_(self.rows)
            self.eng.migration_apply(it, self.all_rows.numpy())
        self.n_exchanges += 1

    def step(self, iter0, n_iters):
        """n_iters of step!/pstep! (main.jl:84-107); runs of iterations without a migration event go to the engine
        as one call."""
        it, end = iter0, iter0 + n_iters
        while it < end:
            if self.eng.migration_due(it):
                self._exchange(it)
            run = 1
            while it + run < end and not self.eng.migration_due(it + run):
                run += 1
            self.eng.update(it, run)
            it += run


def gather_history(driver, row0, row1):
    """bundle_samples needs every particle's history on rank 0 (main.jl:222-250): one gather at the end."""
    th, acc, lp, idh = driver.eng.get_history(row0, row1)
    if not driver.dist:
        return th, acc, lp, idh
    t = driver.torch
    outs = []
    for a in (th, acc, lp, idh):
        x = t.from_numpy(np.ascontiguousarray(a)).to(driver.device)
        buf = [t.empty_like(x) for _ in range(driver.world)]
        driver.dist.all_gather(buf, x)
        outs.append(t.cat(buf, dim=1).cpu().numpy())
    return tuple(outs)
