"""ctypes binding of libdemc_hip.so (include/demc.h).

This is the only compute backend of the package: if the shared library is missing or no HIP
device is visible, every entry point raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdemc_hip.so")

OK, EINVAL, EHIP, ENOMEM, ERCCL, EUNSUPPORTED = 0, 1, 2, 3, 4, 5
_CODE = {1: "DEMC_EINVAL", 2: "DEMC_EHIP", 3: "DEMC_ENOMEM", 4: "DEMC_ERCCL", 5: "DEMC_EUNSUPPORTED"}

EXPORTS = [
    "demc_version", "demc_create", "demc_destroy", "demc_last_error", "demc_set_stream", "demc_set_model",
    "demc_set_model_source", "demc_set_priors", "demc_set_bounds", "demc_set_blocks", "demc_set_state", "demc_get_state",
    "demc_set_history_rows", "demc_get_history", "demc_export_chains", "demc_step", "demc_update", "demc_migration_due",
    "demc_migration_pack", "demc_migration_apply", "demc_migration_pack_async", "demc_migration_apply_async", "demc_migration_groups",
    "demc_update_groups_async", "demc_apply_migration", "demc_get_weights", "demc_logpost",
    "demc_get_trace", "demc_set_replay", "demc_timing_enable", "demc_timing_read", "demc_timing_clock",
    "demc_step_async", "demc_synchronize", "demc_last_kernels", "demc_set_model_source_row",
    "demc_comm_unique_id", "demc_comm_init", "demc_comm_destroy", "demc_comm_set_overlap", "demc_migration_exchange",
    "demc_migration_exchange_async", "demc_comm_allreduce", "demc_comm_stats",
    "demc_create_multi", "demc_destroy_multi", "demc_multi_last_error", "demc_multi_size", "demc_multi_shard", "demc_multi_step",
]
COMM_ID_BYTES = 128
_NOT_STATUS = {"demc_last_error": C.c_char_p, "demc_multi_last_error": C.c_char_p, "demc_multi_shard": C.c_void_p}


class DemcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_CODE.get(code, code)}: {msg}")
        self.code = code


class DemcConfig(C.Structure):
    """POD mirror of demc_config (include/demc.h) = the DE keyword constructor (structs.jl:80-131)."""
    _fields_ = [
        ("n_groups", C.c_int32), ("Np", C.c_int32), ("D", C.c_int32), ("n_blocks", C.c_int32),
        ("burnin", C.c_int64), ("n_initial", C.c_int64), ("n_rows", C.c_int64),
        ("alpha", C.c_double), ("beta", C.c_double), ("eps", C.c_double), ("sigma", C.c_double),
        ("kappa", C.c_double), ("theta_snooker", C.c_double),
        ("proposal_kind", C.c_int32), ("partner_kind", C.c_int32), ("update_kind", C.c_int32),
        ("fitness_kind", C.c_int32), ("schedule", C.c_int32), ("store_history", C.c_int32),
        ("group_offset", C.c_int32), ("n_groups_total", C.c_int32),
        ("seed", C.c_uint64), ("device_id", C.c_int32), ("loglike_mode", C.c_int32),
        ("trace", C.c_int32), ("fuse", C.c_int32), ("geometry_groups", C.c_int32), ("reserved0", C.c_int32),
    ]


class DemcReplay(C.Structure):
    """demc_replay (include/demc.h): caller-supplied draws for the test mode of demc_set_replay"""
    _fields_ = [
        ("u_step", C.POINTER(C.c_double)), ("u_group", C.POINTER(C.c_double)), ("u_part", C.POINTER(C.c_double)),
        ("partner", C.POINTER(C.c_int64)), ("u_noise", C.POINTER(C.c_double)), ("z_noise", C.POINTER(C.c_double)),
        ("u_recomb", C.POINTER(C.c_double)), ("mig_groups", C.POINTER(C.c_int32)), ("n_mig_groups", C.c_int32),
        ("reserved", C.c_int32), ("mig_particle", C.POINTER(C.c_int64)),
    ]


def fill_replay(struct, P, D, G, u_step=None, u_group=None, u_part=None, partner=None, u_noise=None, z_noise=None,
                u_recomb=None, mig_groups=None, mig_particle=None):
    """Fill a replay struct (this module's DemcReplay, or a struct of the same field names) from arrays; returns the list
    of arrays that must stay alive for the call.  Shapes: u_part [P][5], partner [P][3], u_noise/z_noise/u_recomb [P][D],
    u_group / mig_particle [G], mig_groups [n]."""
    keep = []

    def arr(x, dtype, shape):
        if x is None:
            return None
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=dtype), shape))
        keep.append(a)
        return a

    def ptr(a, ct):
        return None if a is None else a.ctypes.data_as(C.POINTER(ct))

    struct.u_step = ptr(arr(u_step, np.float64, (1,)), C.c_double)
    struct.u_group = ptr(arr(u_group, np.float64, (G,)), C.c_double)
    struct.u_part = ptr(arr(u_part, np.float64, (P, 5)), C.c_double)
    struct.partner = ptr(arr(partner, np.int64, (P, 3)), C.c_int64)
    struct.u_noise = ptr(arr(u_noise, np.float64, (P, D)), C.c_double)
    struct.z_noise = ptr(arr(z_noise, np.float64, (P, D)), C.c_double)
    struct.u_recomb = ptr(arr(u_recomb, np.float64, (P, D)), C.c_double)
    mg = None if mig_groups is None else np.ascontiguousarray(mig_groups, dtype=np.int32)
    if mg is not None:
        keep.append(mg)
    struct.mig_groups = ptr(mg, C.c_int32)
    struct.n_mig_groups = 0 if mg is None else int(mg.size)
    struct.reserved = 0
    struct.mig_particle = ptr(arr(mig_particle, np.int64, (G,)), C.c_int64)
    return keep


CFG_KEYS = [f[0] for f in DemcConfig._fields_]
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)
_lib = None


def load():
    """Load libdemc_hip.so; raises if it has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAME as /opt/rocm's).  If this
    # library pulled in /opt/rocm's copy first and torch loaded its own afterwards, the process would hold two HIP
    # runtimes and the second would see no GPU.  Loading torch first makes the loader resolve our DT_NEEDED
    # libamdhip64.so.7 to the copy that is already mapped, so both share one runtime (device memory, streams).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.demc_version.restype = C.c_int32
    L.demc_create.argtypes = [C.POINTER(DemcConfig), C.POINTER(H)]
    L.demc_destroy.argtypes = [H]
    L.demc_last_error.argtypes = [H]
    L.demc_last_error.restype = C.c_char_p
    L.demc_set_stream.argtypes = [H, C.c_void_p]
    L.demc_set_model.argtypes = [H, C.c_int32, _dp, _lp, C.c_int32, _dp, C.c_int32]
    L.demc_set_model_source.argtypes = [H, C.c_char_p, _dp, _lp, C.c_int32, _dp, C.c_int32]
    L.demc_set_model_source_row.argtypes = [H, C.c_char_p, _dp, _lp, C.c_int32, _dp, C.c_int32, C.c_int32]
    L.demc_set_priors.argtypes = [H, _ip, _dp, _dp, _ip]
    L.demc_set_bounds.argtypes = [H, _dp, _dp]
    L.demc_set_blocks.argtypes = [H, _bp, C.c_int32]
    L.demc_set_state.argtypes = [H, _dp, _dp, _lp]
    L.demc_get_state.argtypes = [H, _dp, _dp, _lp]
    L.demc_set_history_rows.argtypes = [H, C.c_int64, C.c_int64, _dp]
    L.demc_get_history.argtypes = [H, C.c_int64, C.c_int64, _dp, _bp, _dp, _lp]
    L.demc_export_chains.argtypes = [H, C.c_int64, C.c_int64, C.c_int32, _dp]
    L.demc_step.argtypes = [H, C.c_int64, C.c_int32]
    L.demc_update.argtypes = [H, C.c_int64, C.c_int32]
    L.demc_migration_due.argtypes = [C.POINTER(DemcConfig), C.c_int64]
    L.demc_migration_pack.argtypes = [H, C.c_int64, C.c_void_p]
    L.demc_migration_apply.argtypes = [H, C.c_int64, C.c_void_p]
    L.demc_migration_pack_async.argtypes = [H, C.c_int64, C.c_void_p]
    L.demc_migration_apply_async.argtypes = [H, C.c_int64, C.c_void_p]
    L.demc_migration_groups.argtypes = [C.POINTER(DemcConfig), C.c_int64, _ip, _ip]
    L.demc_update_groups_async.argtypes = [H, C.c_int64, C.c_int32, _ip, C.c_int32]
    L.demc_logpost.argtypes = [H, _dp, C.c_int64, _dp]
    L.demc_get_trace.argtypes = [H, _dp, _dp, _dp, _ip, _bp]
    L.demc_set_replay.argtypes = [H, C.POINTER(DemcReplay)]
    L.demc_timing_enable.argtypes = [H, C.c_int32]
    L.demc_timing_read.argtypes = [H, _dp, C.c_int32]
    L.demc_timing_clock.argtypes = [H, _dp]
    L.demc_step_async.argtypes = [H, C.c_int64, C.c_int32]
    L.demc_synchronize.argtypes = [H]
    L.demc_last_kernels.argtypes = [H, C.c_char_p, C.c_int32]
    L.demc_comm_unique_id.argtypes = [C.c_void_p, C.c_int32]
    L.demc_comm_init.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32]
    L.demc_comm_destroy.argtypes = [H]
    L.demc_comm_set_overlap.argtypes = [H, C.c_int32]
    L.demc_migration_exchange.argtypes = [H, C.c_int64]
    L.demc_migration_exchange_async.argtypes = [H, C.c_int64]
    L.demc_comm_allreduce.argtypes = [H, _dp, C.c_int32, C.c_int32]
    L.demc_comm_stats.argtypes = [H, _lp]
    L.demc_create_multi.argtypes = [C.POINTER(DemcConfig), C.c_int32, _ip, C.POINTER(H)]
    L.demc_destroy_multi.argtypes = [H]
    L.demc_multi_last_error.argtypes = [H]
    L.demc_multi_size.argtypes = [H]
    L.demc_multi_shard.argtypes = [H, C.c_int32]
    L.demc_multi_step.argtypes = [H, C.c_int64, C.c_int32]
    for name in EXPORTS:  # every entry point returns an int32 status, except the error strings and the shard accessor
        getattr(L, name).restype = _NOT_STATUS.get(name, C.c_int32)
    _lib = L
    return L


def make_config(**kw):
    d = dict(n_groups=4, Np=4, D=1, n_blocks=0, burnin=1000, n_initial=0, n_rows=0, alpha=0.1, beta=0.1, eps=0.001,
             sigma=0.05, kappa=1.0, theta_snooker=0.0, proposal_kind=0, partner_kind=0, update_kind=0,
             fitness_kind=0, schedule=2, store_history=1, group_offset=0, n_groups_total=0, seed=1, device_id=0,
             loglike_mode=0, trace=0, fuse=0, geometry_groups=0, reserved0=0)
    for k, v in kw.items():
        if k in d:
            d[k] = v
    c = DemcConfig()
    for k, v in d.items():
        setattr(c, k, v)
    if c.n_groups_total == 0:
        c.n_groups_total = c.n_groups
    return c


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


class HipEngine:
    """One demc_handle (one GPU, one shard of groups).  Method names and array shapes are the engine interface
    the host driver (sampler.py / distributed.py) is written against."""

    def __init__(self, _shard_of=None, **cfg):
        self.L = load()
        self.cfg = make_config(**cfg)
        self.h = C.c_void_p()
        self._owned = _shard_of is None
        if _shard_of is not None:  # a shard of a MultiEngine: the set owns the handle
            self.h = C.c_void_p(_shard_of)
        else:
            rc = self.L.demc_create(C.byref(self.cfg), C.byref(self.h))
            if rc != OK:
                msg = self.L.demc_last_error(self.h).decode() if self.h else "demc_create failed"
                if self.h:
                    self.L.demc_destroy(self.h)
                    self.h = C.c_void_p()
                raise DemcError(rc, msg)
            note = self.L.demc_last_error(self.h)
            if note and note.decode().startswith("note:"):  # a documented deviation in force on this handle (e.g. shard-local DE-MC_Z)
                import warnings
                warnings.warn(note.decode()[6:], stacklevel=2)
        self.P = self.cfg.n_groups * self.cfg.Np
        self.D = self.cfg.D

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.L.demc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise DemcError(rc, self.L.demc_last_error(self.h).decode())

    def set_stream(self, hip_stream):
        self._ck(self.L.demc_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_model(self, family, data, dims, hyper=None):
        data = None if data is None else np.ascontiguousarray(np.asarray(data, dtype=np.float64).ravel())
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int64).ravel())
        hyper = None if hyper is None else np.ascontiguousarray(np.asarray(hyper, dtype=np.float64).ravel())
        self._ck(self.L.demc_set_model(self.h, family, _d(data), dims.ctypes.data_as(_lp), dims.size, _d(hyper),
                                       0 if hyper is None else hyper.size))

    def set_model_source(self, source, data, dims, hyper=None):
        """user log-density plug-in: HIP source defining demc_user_obs(...) (include/demc.h), JIT-compiled for gfx950"""
        data = np.ascontiguousarray(np.asarray(data, dtype=np.float64).ravel())
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int64).ravel())
        hyper = None if hyper is None else np.ascontiguousarray(np.asarray(hyper, dtype=np.float64).ravel())
        self._ck(self.L.demc_set_model_source(self.h, source.encode(), _d(data), dims.ctypes.data_as(_lp), dims.size,
                                              _d(hyper), 0 if hyper is None else hyper.size))

    def set_model_source_row(self, source, data, dims, hyper=None, has_prior=False):
        """whole-row plug-in: HIP source defining demc_user_loglike_row(...) [and demc_user_prior_row(...)] (include/demc.h)"""
        data = np.ascontiguousarray(np.asarray(data, dtype=np.float64).ravel())
        dims = np.ascontiguousarray(np.asarray(dims, dtype=np.int64).ravel())
        hyper = None if hyper is None else np.ascontiguousarray(np.asarray(hyper, dtype=np.float64).ravel())
        self._ck(self.L.demc_set_model_source_row(self.h, source.encode(), _d(data), dims.ctypes.data_as(_lp), dims.size,
                                                  _d(hyper), 0 if hyper is None else hyper.size, 1 if has_prior else 0))

    def set_priors(self, kind, a, b, ref=None):
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        ref = np.zeros(self.D, np.int32) if ref is None else np.ascontiguousarray(ref, dtype=np.int32)
        self._ck(self.L.demc_set_priors(self.h, kind.ctypes.data_as(_ip), _d(a), _d(b), ref.ctypes.data_as(_ip)))

    def set_bounds(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self._ck(self.L.demc_set_bounds(self.h, _d(lo), _d(hi)))

    def set_blocks(self, masks):
        masks = np.ascontiguousarray(masks, dtype=np.uint8).reshape(-1, self.D)  # zero rows: blocking off
        self._ck(self.L.demc_set_blocks(self.h, masks.ctypes.data_as(_bp), masks.shape[0]))

    def set_state(self, theta, weight=None, ids=None):
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(self.P, self.D)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
        i = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        self._ck(self.L.demc_set_state(self.h, _d(theta), _d(w), None if i is None else i.ctypes.data_as(_lp)))

    def get_state(self):
        th = np.empty((self.P, self.D))
        w = np.empty(self.P)
        i = np.empty(self.P, np.int64)
        self._ck(self.L.demc_get_state(self.h, _d(th), _d(w), i.ctypes.data_as(_lp)))
        return th, w, i

    def get_weights(self):
        w = np.empty(self.P)
        self._ck(self.L.demc_get_weights(self.h, _d(w)))
        return w

    def apply_migration(self, src_slot, dst_slot):
        """shift_particles! with a host-drawn plan (migration.jl:84-91): slot dst[k] receives what slot src[k] held"""
        src = np.ascontiguousarray(src_slot, dtype=np.int32)
        dst = np.ascontiguousarray(dst_slot, dtype=np.int32)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError("src_slot and dst_slot must be 1-D and of equal length")
        _li = C.POINTER(C.c_int32)
        self._ck(self.L.demc_apply_migration(self.h, src.ctypes.data_as(_li), dst.ctypes.data_as(_li), len(src)))

    def set_history_rows(self, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, self.P, self.D)
        self._ck(self.L.demc_set_history_rows(self.h, row0, rows.shape[0], _d(rows)))

    def get_history(self, row0, row1):
        n = row1 - row0
        th = np.empty((n, self.P, self.D))
        acc = np.empty((n, self.P), np.uint8)
        lp = np.empty((n, self.P))
        idh = np.empty((n, self.P), np.int64)
        self._ck(self.L.demc_get_history(self.h, row0, row1, _d(th), acc.ctypes.data_as(_bp), _d(lp),
                                         idh.ctypes.data_as(_lp)))
        return th, acc, lp, idh

    def export_chains(self, row0, row1):
        """history rows [row0,row1) re-keyed by particle id on the device, as the Chains value array [n][D+2][P]
        (parameters, acceptance, lp) -- bundle_samples' gather (main.jl:232-241) without a host-side scatter"""
        out = np.empty((row1 - row0, self.D + 2, self.P))
        self._ck(self.L.demc_export_chains(self.h, row0, row1, 1, _d(out)))
        return out

    def step(self, iter0, n_iters=1):
        self._ck(self.L.demc_step(self.h, iter0, n_iters))

    def update(self, iter0, n_iters=1):
        self._ck(self.L.demc_update(self.h, iter0, n_iters))

    def step_enqueue(self, iter0, n_iters=1):
        """demc_step without the drain (demc_step_async); synchronize() waits and reports"""
        self._ck(self.L.demc_step_async(self.h, iter0, n_iters))

    def synchronize(self):
        self._ck(self.L.demc_synchronize(self.h))

    # ---- the communicator behind the C-ABI (one process per GPU; include/demc.h "The one collective") ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId as 128 bytes (rank 0 calls it, the host carries the bytes to the other ranks)"""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = load().demc_comm_unique_id(buf, COMM_ID_BYTES)
        if rc != OK:
            raise DemcError(rc, "demc_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self._ck(self.L.demc_comm_init(self.h, C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), rank, world))

    def comm_destroy(self):
        self._ck(self.L.demc_comm_destroy(self.h))

    def comm_set_overlap(self, on=True):
        self._ck(self.L.demc_comm_set_overlap(self.h, 1 if on else 0))

    def migration_exchange(self, it):
        self._ck(self.L.demc_migration_exchange(self.h, it))

    def migration_exchange_enqueue(self, it):
        self._ck(self.L.demc_migration_exchange_async(self.h, it))

    def comm_allreduce(self, values, op="sum"):
        """host doubles reduced over the ranks of the handle's communicator (empty: a barrier)"""
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64))).copy()
        self._ck(self.L.demc_comm_allreduce(self.h, _d(v) if v.size else None, v.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return v

    def comm_stats(self):
        out = np.zeros(3, np.int64)
        self._ck(self.L.demc_comm_stats(self.h, out.ctypes.data_as(_lp)))
        return dict(world=int(out[0]), rank=int(out[1]), exchanges=int(out[2]))

    def logpost(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1, self.D)
        out = np.empty(theta.shape[0])
        self._ck(self.L.demc_logpost(self.h, _d(theta), theta.shape[0], _d(out)))
        return out

    def get_trace(self):
        prop = np.empty((self.P, self.D))
        w = np.empty(self.P)
        adj = np.empty(self.P)
        idx = np.empty((self.P, 4), np.int32)
        acc = np.empty(self.P, np.uint8)
        self._ck(self.L.demc_get_trace(self.h, _d(prop), _d(w), _d(adj), idx.ctypes.data_as(_ip),
                                       acc.ctypes.data_as(_bp)))
        return dict(proposal=prop, w_prop=w, log_adj=adj, idx=idx, accepted=acc)

    def last_kernels(self):
        """the kernel instances the last update launched (diagnostic, demc_last_kernels)"""
        buf = C.create_string_buffer(512)
        self._ck(self.L.demc_last_kernels(self.h, buf, 512))
        return buf.value.decode()

    def set_replay(self, **draws):
        """test mode: caller-supplied draws (see fill_replay / demc_replay); no arguments -> back to Philox"""
        if not draws:
            self._ck(self.L.demc_set_replay(self.h, None))
            return
        r = DemcReplay()
        keep = fill_replay(r, self.P, self.D, self.cfg.n_groups, **draws)
        self._ck(self.L.demc_set_replay(self.h, C.byref(r)))
        del keep

    def migration_due(self, it):
        return bool(self.L.demc_migration_due(C.byref(self.cfg), it))

    def migration_pack_dev(self, it, dev_ptr):
        """select_particle for every local group; rows [n_groups][D+3] written to the device address dev_ptr."""
        self._ck(self.L.demc_migration_pack(self.h, it, C.c_void_p(dev_ptr)))

    def migration_apply_dev(self, it, dev_ptr):
        """shift_particles! given the all-gathered rows [n_groups_total][D+3] at the device address dev_ptr."""
        self._ck(self.L.demc_migration_apply(self.h, it, C.c_void_p(dev_ptr)))

    def migration_pack_enqueue(self, it, dev_ptr):
        """as migration_pack_dev, but only enqueued on the handle's stream (no drain)"""
        self._ck(self.L.demc_migration_pack_async(self.h, it, C.c_void_p(dev_ptr)))

    def migration_apply_enqueue(self, it, dev_ptr):
        self._ck(self.L.demc_migration_apply_async(self.h, it, C.c_void_p(dev_ptr)))

    def migration_groups(self, it):
        """select_groups' ordered sub-group of iteration `it` (global group indices)"""
        sel = np.empty(self.cfg.n_groups_total, np.int32)
        n = C.c_int32()
        rc = self.L.demc_migration_groups(C.byref(self.cfg), it, sel.ctypes.data_as(_ip), C.byref(n))
        if rc != OK:
            raise DemcError(rc, "demc_migration_groups")
        return sel[: n.value].copy()

    def update_groups_enqueue(self, iter0, n_iters, groups):
        """update! + store_samples! for a subset of the local groups, enqueued on the handle's stream (no drain)"""
        g = np.ascontiguousarray(groups, dtype=np.int32)
        self._ck(self.L.demc_update_groups_async(self.h, iter0, n_iters, g.ctypes.data_as(_ip), g.size))

    def timing_enable(self, on=True):
        self._ck(self.L.demc_timing_enable(self.h, 1 if on else 0))

    def timing_read(self, reset=True):
        out = np.zeros(10)
        self._ck(self.L.demc_timing_read(self.h, _d(out), 1 if reset else 0))
        names = ("propose", "loglike_prep", "loglike", "accept_store", "migration")
        return {n: dict(ms=out[i], launches=int(out[5 + i])) for i, n in enumerate(names)}

    def timing_clock(self):
        """shader clock (MHz) the DIRECT MvNormal likelihood kernel (or the LBA wave kernel) held in its last launch with timing enabled: median / min /
        max over the CUs (per CU: s_memtime ticks over 100 MHz s_memrealtime ticks between the first and the last workgroup to
        finish there); None if no such launch ran or it was too short to difference"""
        out = np.zeros(4)
        self._ck(self.L.demc_timing_clock(self.h, _d(out)))
        if out[3] == 0:
            return None
        return dict(mhz_median=float(out[0]), mhz_min=float(out[1]), mhz_max=float(out[2]), cus=int(out[3]))


class MultiEngine:
    """demc_create_multi: ONE host thread drives several shards (GPUs).  `cfg` describes the whole population; shard r is
    an ordinary engine (self.shards[r]) for the per-shard calls, step() runs step! over the set."""

    def __init__(self, n_shards, device_ids=None, **cfg):
        self.L = load()
        self.cfg = make_config(**cfg)
        self.m = C.c_void_p()
        dev = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        rc = self.L.demc_create_multi(C.byref(self.cfg), n_shards, None if dev is None else dev.ctypes.data_as(_ip), C.byref(self.m))
        if rc != OK:
            msg = self.L.demc_multi_last_error(self.m).decode() if self.m else "demc_create_multi failed"
            if self.m:
                self.L.demc_destroy_multi(self.m)
                self.m = C.c_void_p()
            raise DemcError(rc, msg)
        G = self.cfg.n_groups // n_shards
        self.shards = []
        for r in range(n_shards):
            kw = {k: getattr(self.cfg, k) for k in CFG_KEYS}
            kw.update(n_groups=G, group_offset=r * G, n_groups_total=self.cfg.n_groups,
                      device_id=r if dev is None else int(dev[r]))
            self.shards.append(HipEngine(_shard_of=self.L.demc_multi_shard(self.m, r), **kw))
        self.P = self.cfg.n_groups * self.cfg.Np
        self.D = self.cfg.D

    def each(self, fn):
        """the per-shard set-up calls are the same on every shard: fn(shard_engine)"""
        for e in self.shards:
            fn(e)

    def set_state(self, theta, weight=None, ids=None):
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(self.P, self.D)
        for e in self.shards:
            lo = e.cfg.group_offset * e.cfg.Np
            e.set_state(theta[lo:lo + e.P], None if weight is None else np.asarray(weight)[lo:lo + e.P],
                        None if ids is None else np.asarray(ids)[lo:lo + e.P])

    def get_state(self):
        parts = [e.get_state() for e in self.shards]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

    def get_history(self, row0, row1):
        parts = [e.get_history(row0, row1) for e in self.shards]
        return tuple(np.concatenate([p[i] for p in parts], axis=1) for i in range(4))

    def step(self, iter0, n_iters=1):
        rc = self.L.demc_multi_step(self.m, iter0, n_iters)
        if rc != OK:
            raise DemcError(rc, self.L.demc_multi_last_error(self.m).decode())

    def close(self):
        if getattr(self, "m", None):
            for e in self.shards:
                e.close()
            self.L.demc_destroy_multi(self.m)
            self.m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
