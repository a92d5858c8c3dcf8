#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DE-MCMC hot path.

Metric (BASELINE.json): particle-updates/sec (proposal + loglike + accept) at D=32, N=1e5.
Workload at N=1 (BASELINE.json configs[2], "cfg3"): Multivariate Gaussian D=32 full-Sigma, n_groups=256, Np=256,
N=1e5 observations, sampler defaults (alpha=beta=0.1, eps=1e-3, sigma=0.05, kappa=1, no snooker, burnin=1000),
two_colour schedule, STREAMING likelihood (every proposal visits every observation, as the reference's loglike does).
One "step" = one DE-MCMC iteration over all P = n_groups*Np particles (migration when the alpha coin fires,
proposal, prior+loglike, Metropolis accept, history store).  N>1: weak scaling, every rank owns 256 groups.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (= vector) peak; BASELINE.md section 5
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E nominal; /opt/skills/guides/MI355X_MICROARCH.md


def measured_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r01/*_pmc.json: FETCH_SIZE and WRITE_SIZE in KB, separate --pmc passes).  On gfx950 FETCH_SIZE counts
    half the bytes of wide streaming reads (MI355X_MICROARCH.md, HBM section), hence the factor 2.  Only valid for
    the default workload; otherwise None."""
    if (a.n_groups, a.Np, a.nobs, a.dim, a.schedule) != (256, 256, 100000, 32, "two_colour"):
        return None
    path = os.path.join(ROOT, "profiles", "r01", f"bench_cfg3_{a.mode}_pmc.json")
    try:
        rec = json.load(open(path))[kernel]
        if "bytes" in rec:  # already reduced by tools/collect_profiles.py (e.g. per iteration for the resident K1)
            return float(rec["bytes"])
        return (2.0 * rec["FETCH_SIZE_KB_mean"] + rec["WRITE_SIZE_KB_mean"]) * 1024.0
    except (OSError, KeyError):
        return None


def make_cfg3(n_groups, Np, N, d, seed=20260002):
    """Sigma = A A'/d + 0.5 I, X rows ~ N(mu*, Sigma), prior mu_j ~ N(0,1)"""
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 1, (d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    mu = rng.normal(0, 1, d)
    L = np.linalg.cholesky(Sigma)
    X = mu + rng.normal(0, 1, (N, d)) @ L.T
    return dict(Sigma=Sigma, X=np.ascontiguousarray(X), mu=mu)


def init_theta(P, d, rank, seed=20260003):
    return np.random.default_rng(seed + rank).normal(0, 1, (P, d))


def configure(eng, prob, d):
    from demc_amd import families as F
    eng.set_model(F.FAM_MVN_FULL, prob["X"], [prob["X"].shape[0], d], prob["Sigma"])
    eng.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
    eng.set_bounds([-np.inf] * d, [np.inf] * d)


def cpu_baseline(prob, Np, N, d, seconds_target=20.0):
    """C restatement of the reference algorithm and schedule (sequential in-place sweep per group, groups across
    OpenMP threads like p_update!, src/main.jl:135-148) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from demc_amd import families as F
    O.use_native_build()  # -O3 -march=native -fopenmp build of the same C source, compiled on this host
    # threads actually used: the process's CPU share, at most 16 (a 1-GPU box's share of the host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    ng = max(2, cores)
    def run(n_groups, threads, seconds):
        o = O.Oracle(n_groups=n_groups, Np=Np, D=d, schedule=0, n_rows=0, store_history=0, seed=7, n_threads=threads)
        o.set_model(F.FAM_MVN_FULL, prob["X"], [N, d], prob["Sigma"])
        o.set_priors([F.PRIOR_NORMAL] * d, [0.0] * d, [1.0] * d)
        o.set_bounds([-np.inf] * d, [np.inf] * d)
        o.set_state(init_theta(n_groups * Np, d, 1234))
        t0 = time.time()
        o.step(1, 1)
        t1 = time.time() - t0
        iters = int(max(1, min(20, seconds / max(t1, 1e-3))))
        t0 = time.time()
        o.step(2, iters)
        dt = time.time() - t0
        o.close()
        return n_groups * Np * iters / dt, iters

    one_thread, _ = run(1, 1, 3.0)  # one group on one thread, ~3 s
    value, iters = run(ng, cores, seconds_target)
    dt = ng * Np * iters / value
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(value=value, unit="particle-updates/s", cores=cores, kind="port", value_single_thread=one_thread,
                cpu_model=cpu_model,
                sample=f"cfg3 shape (D={d}, N={N}, Np={Np}) on {ng} of the groups, {iters} iterations, reference "
                       f"schedule (sequential in-group sweep, one group per OpenMP thread), whitened O(N*D) "
                       f"likelihood per proposal; gcc -O3 -march=native -fopenmp")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--mode", default="streaming", choices=["streaming", "suffstat"])
    ap.add_argument("--schedule", default="two_colour", choices=["two_colour", "synchronous"])
    ap.add_argument("--n-groups", type=int, default=256)
    ap.add_argument("--np", type=int, default=256, dest="Np")
    ap.add_argument("--nobs", type=int, default=100000)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--burnin", type=int, default=1000, help="DE burnin (reference default 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    import torch
    import demc_amd
    from demc_amd.distributed import ShardedDriver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible and there is no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("DEMC_FORCE_DIST") == "1":  # the latter: exercise the RCCL path with one rank
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        dist = dist_

    d, N, G, Np = a.dim, a.nobs, a.n_groups, a.Np
    P = G * Np
    n_rows = a.warmup + a.steps
    prob = make_cfg3(G, Np, N, d)
    eng = demc_amd.HipEngine(n_groups=G, Np=Np, D=d, n_rows=n_rows, schedule=2 if a.schedule == "two_colour" else 1,
                             group_offset=rank * G, n_groups_total=G * world, seed=20260001, device_id=local, burnin=a.burnin,
                             loglike_mode=0 if a.mode == "streaming" else 1, trace=0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    configure(eng, prob, d)
    eng.set_state(init_theta(P, d, rank))
    drv = ShardedDriver(eng, dist, torch.device("cuda", local))

    def sync():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    drv.step(1, a.warmup)
    sync()
    t0 = time.perf_counter()
    drv.step(1 + a.warmup, a.steps)
    sync()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = P * world * a.steps / dt

    # accuracy half of the metric ("posterior-mean L1 vs ref"): mu | X is Gaussian in closed form for this model
    # (prior mu ~ N(0, I), known Sigma): precision N Sigma^-1 + I, mean (N Sigma^-1 + I)^-1 N Sigma^-1 xbar.
    acc_stats = None
    if rank == 0:
        k_last = min(10, a.steps)
        th_h, acc_h, _, _ = eng.get_history(n_rows - k_last, n_rows)
        Ainv = np.linalg.inv(prob["Sigma"])
        post_mean = np.linalg.solve(N * Ainv + np.eye(d), N * Ainv @ prob["X"].mean(0))
        post_sd = np.sqrt(np.diag(np.linalg.inv(N * Ainv + np.eye(d))))
        chain_mean = th_h.reshape(-1, d).mean(0)
        acc_stats = dict(posterior_mean_l1_rel=float(np.abs(chain_mean - post_mean).sum() / np.abs(post_mean).sum()),
                         max_abs_err_in_posterior_sd=float(np.max(np.abs(chain_mean - post_mean) / post_sd)),
                         # the timed iterations lie inside the reference's burn-in (burnin = 1000: the gamma_2 pull towards
                         # high-weight particles is active, crossover.jl:164), so the ensemble is still contracting
                         ensemble_sd_over_posterior_sd_in_burnin=float(np.median(th_h.reshape(-1, d).std(0) / post_sd)),
                         accept_rate=float(acc_h.mean()), rows_used=k_last,
                         reference="closed-form Gaussian posterior of cfg3 (conjugate)")

    roofline = None
    if not a.no_roofline:
        # dominant kernel, timed live with HIP events on the stream the kernels run on (same process, extra iterations
        # overwrite the last history rows)
        k = min(20, a.steps)
        eng.timing_enable(True)
        eng.update(n_rows - k + 1, k)
        tm = eng.timing_read()
        eng.timing_enable(False)
        phases = 2 if a.schedule == "two_colour" else 1
        units = P / phases  # particle-updates per launch of the likelihood kernel
        if a.mode == "streaming":
            t_launch = tm["loglike"]["ms"] / max(1, tm["loglike"]["launches"]) * 1e-3
            flops = (3.0 * N * d + 2.0 * d * d) * units  # algorithmic flops per particle-update
            ach = flops / t_launch / 1e12
            roofline = dict(bound="mfma", kernel="k_cross_mfma<8,4> (v_mfma_f64_16x16x4_f64)", achieved=ach,
                            peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP64_MFMA_TFLOPS,
                            traffic=measured_traffic("demc::k_cross_mfma<8, 4>", a),
                            launch_ms=t_launch * 1e3, executed_tflops=2.0 * N * d * units / t_launch / 1e12)
        else:
            # fused K1 does the whole update.  Resident form: ONE launch runs all k iterations (both colour phases each), so
            # the rate is taken over the iterations, not per launch; launch_ms and updates_per_launch say what one launch was.
            t_total = (tm["propose"]["ms"] + tm["loglike_prep"]["ms"] + tm["accept_store"]["ms"]) * 1e-3
            launches = max(1, tm["propose"]["launches"])
            byts = (24.0 * d + 17.0) * P * k  # algorithmic bytes per particle-update x updates in the k iterations
            ach = byts / t_total / 1e9
            resident = launches < phases * k
            tr = measured_traffic("k_propose_fused_per_iteration", a)
            roofline = dict(bound="hbm",
                            kernel="k_propose, fused prep/accept/store tail, " +
                                   ("resident form (one launch per run of iterations between migrations)" if resident
                                    else "one launch per colour phase"),
                            achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                            traffic=None if tr is None else tr * k / launches, launch_ms=t_total / launches * 1e3,
                            updates_per_launch=P * k / launches)
        roofline["per_kernel_ms_per_iter"] = {n: v["ms"] / k for n, v in tm.items()}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(prob, Np, N, d)

    eng.close()
    if rank == 0:
        out = {
            "metric": "particle-updates/sec (proposal+loglike+accept) at D=32, N=1e5",
            "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfg3: MvNormal full-Sigma D={d}, N={N} obs, n_groups={G}x{world}, Np={Np}, "
                                   f"sampler defaults, schedule={a.schedule}, loglike={a.mode}",
                       "particles_per_gpu": P, "parallelism": f"groups sharded x{world}, migration all-gather"},
            "particle_parameter_updates_per_s": value * d,
            "accuracy": acc_stats, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
