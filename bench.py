#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X DE-MCMC hot path.

Metric (BASELINE.json): particle-updates/sec (proposal + loglike + accept) at D=32, N=1e5.
Default workload (BASELINE.json configs[2], "cfg3", the configuration the metric is quoted on): Multivariate Gaussian D=32
full-Sigma, n_groups=256, Np=256, N=1e5 observations, sampler defaults (alpha=beta=0.1, eps=1e-3, sigma=0.05, kappa=1, no
snooker, burnin=1000), two_colour schedule, DIRECT likelihood: the residual form the reference's loglike writes
(test/multivariate_normal_tests.jl:31-33: sum_i logpdf(MvNormal(mu, Sigma), x_i)), every proposal visits every observation term
by term -- SURVEY 8(d)'s 3*N*D flop per particle-update on the FP64 vector pipe, the work that does not collapse.  (Rounds 1-5
quoted the STREAMING mode here: the expanded quadratic form on the matrix cores, whose [proposals x D].[D x N] product is
analytically zero after centring -- it stays as a labelled row.)  One "step" = one DE-MCMC iteration over all P = n_groups*Np
particles (migration when the alpha coin fires, proposal, prior+loglike, Metropolis accept, history store).

--config cfg2 | cfg4 | cfg5 run the other BASELINE configs (their own roofline definition, same JSON contract); they are
measured rows of SURVEY 8(d), not the headline.

--gpus N: weak scaling, every rank owns the config's groups (256 for cfg3); groups are sharded, the only collective is
one all-gather per migration.  Started by a launcher (RANK / WORLD_SIZE in the environment, e.g. torch.distributed.run)
the process is one rank; started bare with N > 1 it spawns its N ranks itself -- as fresh child processes, before this
process has touched the GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 matrix = vector peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (datasheet)
PEAK_HBM_GBS = 8000.0     # MI355X HBM3E nominal; /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r06"     # profiles/<round>/ holds the rocprofv3 PMC passes `traffic` is read from


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "mvn30"])
    ap.add_argument("--mode", default="direct", choices=["streaming", "suffstat", "direct"],
                    help="MvNormal likelihood: direct = the residual form sum_i |L^-1(x_i - mu)|^2 term by term on the FP64 vector pipe "
                         "(headline: SURVEY 8d's 3*N*D per update, nothing collapses); streaming = expanded quadratic form on the FP64 "
                         "matrix cores (= suffstat + a [proposals x D].[D x N] product that is zero after centring); suffstat = O(D^2) per proposal")
    ap.add_argument("--schedule", default="two_colour", choices=["two_colour", "synchronous"])
    ap.add_argument("--n-groups", type=int, default=None, help="groups per GPU (default: the config's)")
    ap.add_argument("--np", "--particles-per-group", type=int, default=None, dest="Np",
                    help="(under torch.distributed.run spell it --particles-per-group: the launcher's parser claims --np as an abbreviation)")
    ap.add_argument("--nobs", type=int, default=None, help="observations / subjects / trials (default: the config's)")
    ap.add_argument("--dim", type=int, default=None, help="data dimension of cfg2 / cfg3")
    ap.add_argument("--burnin", type=int, default=1000, help="DE burnin (reference default 1000)")
    ap.add_argument("--snooker", type=float, default=None, help="override theta_snooker (crossover.jl:31; the reference's multivariate "
                                                                 "and hierarchical examples use 0.1): runs the snooker instances of the kernels")
    ap.add_argument("--beta", type=float, default=None, help="override beta, the probability of a mutation sweep (main.jl:199-207; default 0.1)")
    ap.add_argument("--fuse", type=int, default=0, help="demc_config.fuse (0 auto, 1 never, 2 per phase)")
    ap.add_argument("--accuracy-iters", type=int, default=1500, help="length of the untimed accuracy leg (0: skip)")
    ap.add_argument("--async-migration", action="store_true",
                    help="groups an exchange does not select update while the all-gather is in flight (SURVEY 8f #3)")
    ap.add_argument("--collective", default="library", choices=["library", "torch"],
                    help="home of the one all-gather per migration: the engine's own RCCL communicator behind the C-ABI "
                         "(demc_comm_init; torch only carries the 128-byte id) or torch.distributed (backend nccl = RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--partners", default="current", choices=["current", "history"],
                    help="partner source: the current population (`sample`, crossover.jl:138-140) or the history of all particles "
                         "(`resample`, DE-MC_Z, crossover.jl:113-124; needs --n-initial > 0 and runs the synchronous schedule)")
    ap.add_argument("--n-initial", type=int, default=0, help="history rows pre-filled with prior draws (structs.jl n_initial)")
    ap.add_argument("--start", default="prior", choices=["prior", "posterior"],
                    help="posterior: start the particles tightly around the generating parameters (the converged regime a long run "
                         "spends its time in) instead of from prior draws")
    ap.add_argument("--rows", default=None,
                    help="the other SURVEY 8(d) rows, measured after the headline's timed region and appended as `rows` to the JSON line: "
                         "`all` (default for the plain `bench.py [--gpus 1]` headline command), `none`, or a comma list of row names")
    ap.add_argument("--deadline", type=float, default=540.0,
                    help="seconds after which a multi-rank run is ended with evidence (parent: all ranks; a rank: itself)")
    ap.add_argument("--init-timeout", type=float, default=150.0,
                    help="seconds a rank may spend between the start of the id exchange and the end of communicator creation")
    return ap.parse_args(argv)


EXIT_INIT = 17    # a rank gave up while the communicator was being created (id exchange, ncclCommInitRank)
EXIT_STUCK = 18   # a rank gave up in a later stage (warm-up, timed region, teardown)


def _tail(path, n=25):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode("utf-8", "replace")
    except OSError:
        return ""


def supervise(cmds, envs, deadline_s, poll_s=0.05, log_dir=None):
    """Runs one child per rank and watches them: the first rank that exits non-zero, or the deadline, ends all the others
    (SIGTERM, SIGKILL two seconds later).  A failed rank never joins the collectives behind it -- its peers would sit in
    ncclAllGather / ncclCommInitRank until somebody else's timeout, and the record would hold nothing.  The supervisor never
    touches the GPU.  Returns {rc, failed_rank, reason, seconds, stdout0, stderr_tails}: rc 0 only if every rank exited 0."""
    log_dir = log_dir or tempfile.mkdtemp(prefix="demc_bench_ranks_")
    n = len(cmds)
    errs = [os.path.join(log_dir, f"rank{r}.stderr") for r in range(n)]
    out0 = os.path.join(log_dir, "rank0.stdout")
    procs, files = [], []
    t0 = time.monotonic()
    for r in range(n):
        fe = open(errs[r], "wb")
        fo = open(out0, "wb") if r == 0 else subprocess.DEVNULL
        files += [fe] + ([fo] if r == 0 else [])
        procs.append(subprocess.Popen(cmds[r], env=envs[r], stdout=fo, stderr=fe, stdin=subprocess.DEVNULL))
    rc, failed, reason = 0, None, None
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc, failed, reason = code, r, f"rank {r} exited with status {code}"
        if rc == 0 and live and time.monotonic() - t0 > deadline_s:
            rc, failed, reason = EXIT_STUCK, min(live), f"deadline of {deadline_s:g} s passed with rank(s) {sorted(live)} still running"
        if rc != 0 and live:
            for r in live:
                procs[r].terminate()
            t_kill = time.monotonic() + 2.0
            while any(procs[r].poll() is None for r in live) and time.monotonic() < t_kill:
                time.sleep(poll_s)
            for r in live:
                if procs[r].poll() is None:
                    procs[r].kill()
                procs[r].wait()
            reason += f"; ended rank(s) {sorted(live)}"
            live = set()
        if live:
            time.sleep(poll_s)
    for f in files:
        f.close()
    return dict(rc=rc, failed_rank=failed, reason=reason, seconds=time.monotonic() - t0, stdout0=_tail(out0, 1 << 20),
                stderr_tails={r: _tail(errs[r]) for r in range(n)}, exit_codes=[p.returncode for p in procs])


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (one process per GPU), before this
    process has imported torch or made any HIP call, and SUPERVISE them (supervise()): whatever happens in the first run of
    more than one rank, the caller gets a non-zero status within seconds of the first failure together with every rank's
    stderr tail.  If the ranks gave up while creating the library's communicator (EXIT_INIT), fresh children are started
    ONCE with --collective torch, and the line they print carries config.collective_fallback."""
    import socket

    def launch(extra_args, extra_env):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmds, envs = [], []
        for r in range(a.gpus):
            envs.append(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                             MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                             **extra_env))
            cmds.append([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra_args)
        return supervise(cmds, envs, a.deadline)

    def report(res, what):
        print(f"bench.py: {what}: {res['reason']} after {res['seconds']:.1f} s; exit codes {res['exit_codes']}", file=sys.stderr)
        for r, t in res["stderr_tails"].items():
            print(f"---- rank {r} stderr (tail) ----\n{t}", file=sys.stderr)
        sys.stderr.flush()

    def clear_rank_files():
        for r in range(a.gpus):
            try:
                os.remove(rank_file(r))
            except OSError:
                pass

    clear_rank_files()
    res = launch([], {})
    if res["rc"] == EXIT_INIT and a.collective == "library":
        report(res, "library collective failed at communicator creation, starting fresh ranks with --collective torch")
        why = f"library collective: {res['reason']}"
        clear_rank_files()
        res = launch(["--collective", "torch"], {"DEMC_BENCH_COLLECTIVE_FALLBACK": why})
    if res["rc"] != 0:
        report(res, f"{a.gpus}-rank run FAILED")
        recs = []
        for r in range(a.gpus):
            try:
                recs.append(json.load(open(rank_file(r))))
            except (OSError, ValueError):
                pass
        if len(recs) == a.gpus:  # every rank got through its timed region: their numbers are not lost with the run
            line = merge_rank_files(recs, f"run failed after the timed region ({res['reason']}); merged from gpurun_out/rank<r>.json")
            if line:
                print(line, flush=True)
        raise SystemExit(res["rc"] if 0 < res["rc"] < 256 else 1)
    sys.stdout.write(res["stdout0"] + ("\n" if res["stdout0"] and not res["stdout0"].endswith("\n") else ""))
    sys.stdout.flush()
    for r, t in res["stderr_tails"].items():  # warnings of a successful run are passed on, not swallowed
        if t.strip():
            print(f"---- rank {r} stderr (tail) ----\n{t}", file=sys.stderr)
    raise SystemExit(0)


class StageWatchdog(threading.Thread):
    """In-rank deadline: the rank names the stage it is in and how long that may take; a stage that overruns ends THIS
    process (os._exit -- no re-exec, nothing is started from a process that holds the GPU) with a line on stderr that says
    where it was.  The launcher (supervise() above, or torch.distributed.run) then ends the other ranks.  The blocking calls
    that can hang -- TCPStore, ncclCommInitRank behind demc_comm_init, a collective -- are C calls made through ctypes /
    torch, which release the GIL, so this thread gets to run."""

    def __init__(self, rank):
        super().__init__(daemon=True)
        self.rank, self.lock = rank, threading.Lock()
        self.name_, self.t_end, self.code = "start", None, EXIT_STUCK
        self.done = False
        self.start()

    def enter(self, name, limit_s, code=EXIT_STUCK):
        with self.lock:
            self.name_, self.t_end, self.code = name, time.monotonic() + limit_s, code
            self.limit = limit_s

    def stop(self):
        with self.lock:
            self.done = True

    def run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                if self.done:
                    return
                late = self.t_end is not None and time.monotonic() > self.t_end
                name, code, limit = self.name_, self.code, getattr(self, "limit", 0.0)
            if late:
                sys.stderr.write(f"bench.py rank {self.rank}: stage '{name}' exceeded its {limit:g} s -- giving up (exit {code})\n")
                sys.stderr.flush()
                os._exit(code)


class SclkSampler(threading.Thread):
    """What the driver says the shader clock is while the timed region runs: the starred level of the device's sysfs
    `pp_dpm_sclk`, read every 20 ms by a host thread (no GPU call).  Context only -- under a dense loop the clock the kernel
    really holds reads up to 10 % below it (MI355X_MICROARCH.md, DVFS give-back): the in-kernel figure is
    roofline.shader_clock_mhz (demc_timing_clock).  None when the file is not there or not readable."""

    def __init__(self, device_index):
        super().__init__(daemon=True)
        self.path, self.samples, self.halt = None, [], threading.Event()
        try:
            import glob
            import torch
            bus = None
            try:
                bus = torch.cuda.get_device_properties(device_index).pci_bus_id  # (torch >= 2.5)
            except Exception:
                pass
            cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            if bus is not None:
                def bus_of(c):  # .../0000:05:00.0 -> 5
                    try:
                        return int(os.path.basename(os.path.realpath(os.path.dirname(c))).split(":")[1], 16)
                    except (IndexError, ValueError):
                        return None
                hit = [c for c in cands if bus_of(c) == int(bus)]
                cands = hit or cands
            if len(cands) == 1 or (cands and bus is not None):
                self.path = cands[0]
        except Exception:
            self.path = None

    def read_once(self):
        try:
            for ln in open(self.path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return float(ln.split(":")[1].strip().split("M")[0])
        except (OSError, ValueError, IndexError, TypeError):
            pass
        return None

    def run(self):
        while self.path and not self.halt.is_set():
            v = self.read_once()
            if v is not None:
                self.samples.append(v)
            self.halt.wait(0.02)

    def result(self):
        self.halt.set()
        if self.is_alive():
            self.join(1.0)
        if not self.samples:
            return None
        xs = sorted(self.samples)
        return dict(median=xs[len(xs) // 2], min=xs[0], max=xs[-1], samples=len(xs), source=self.path)


# ----------------------------------------------------------------------------------------------------------------
# workloads and their rooflines
# ----------------------------------------------------------------------------------------------------------------
def build_workload(a):
    from demc_amd import workloads as W
    kw = {}
    if a.n_groups is not None:
        kw["G"] = a.n_groups
    if a.Np is not None:
        kw["Np"] = a.Np
    if a.config in ("cfg2", "cfg3"):
        if a.nobs is not None:
            kw["N"] = a.nobs
        if a.dim is not None:
            kw["d"] = a.dim
    elif a.config == "cfg4":
        if a.nobs is not None:
            kw["S"] = a.nobs
    elif a.nobs is not None:
        kw["N"] = a.nobs
    w = W.BUILDERS[a.config](**kw)
    if a.snooker is not None:
        w["engine"] = dict(w["engine"], theta_snooker=a.snooker)
    if getattr(a, "beta", None) is not None:
        w["engine"] = dict(w["engine"], beta=a.beta)
    return w


def describe(a, w, world):
    return _describe(a, w, world) + (f", partners=history (DE-MC_Z `resample`, n_initial={a.n_initial}, synchronous schedule)"
                                     if a.partners == "history" else "") + \
        (", started from a converged population (generating parameters +- 1 %)" if a.start == "posterior" and "truth" in w else "") + \
        ("" if a.burnin == 1000 else f", burnin={a.burnin}")


def _describe(a, w, world):
    G, Np, D = w["G"], w["Np"], w["D"]
    if a.config == "cfg1":
        return (f"cfg1: Examples/Gaussian_Example.jl, 1-D Normal(mu, sigma), N={w['dims'][0]} obs, n_groups={G}x{world}, Np={Np}, sampler "
                f"defaults, schedule={a.schedule} (BASELINE's CPU plumbing config, here on the GPU: 40 particles cannot fill one CU)")
    if a.config in ("cfg2", "cfg3"):
        return (f"{a.config}: MvNormal full-Sigma D={D}, N={w['dims'][0]} obs, n_groups={G}x{world}, Np={Np}, sampler defaults, "
                f"schedule={a.schedule}, loglike={a.mode}" + ("" if a.snooker is None else f", theta_snooker={a.snooker:g}"))
    if a.config == "cfg4":
        return (f"cfg4: hierarchical Binomial (Hierarchical_Example.jl shape), S={w['dims'][0]} subjects, D={D}, blocks [hyper; subject], "
                f"n_groups={G}x{world} (BASELINE: 128 groups over 8 GPUs), Np={Np}, schedule={a.schedule}")
    if a.config == "mvn30":
        return (f"mvn30: test/multivariate_normal_tests.jl (MvNormal(mu, sigma^2 I), theta = (mu[30], sigma), N={w['dims'][0]} obs), D={D}, "
                f"n_groups={G}x{world} (the reference: 1 group of 3), Np={Np}, schedule={a.schedule}" +
                ("" if a.snooker is None else f", theta_snooker={a.snooker:g}"))
    return (f"cfg5: LBA 3 accumulators (Run_LBA.jl), D={D}, N={w['dims'][0]} trials simulated from nu=(3,2,1) A=.8 k=.2 tau=.3, "
            f"n_groups={G}x{world} (BASELINE: 512 groups over 8 GPUs), Np={Np}, snooker 0.1, schedule={a.schedule}")


def source_fingerprint():
    """sha256 over the kernel / runtime sources the library is built from: the committed PMC summaries carry the fingerprint
    of the sources they were collected on, and are not quoted for any other (they would go stale silently)"""
    import hashlib
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join(ROOT, "include", "demc.h")]:
        path = f if os.path.isabs(f) else os.path.join(csrc, f)
        if os.path.isfile(path) and path.endswith((".cpp", ".hpp", ".h")):
            hsh.update(open(path, "rb").read())
    return hsh.hexdigest()[:16]


def profile_is_current(rec_file):
    try:
        return json.load(open(rec_file)).get("source_sha16") == source_fingerprint()
    except (OSError, ValueError):
        return False


def profile_tag(a):
    """name under which tools/collect_profiles.py files the rocprofv3 summaries of this command: `headline` for the default
    command, the row's name for a row of ROWS (also when its flags are given by hand), else None (nothing was profiled)"""
    if getattr(a, "profile_tag", None):
        return a.profile_tag
    mine = {k: getattr(a, k) for k in ROW_DEFAULTS}
    if mine == ROW_DEFAULTS:
        return "headline"
    for name, over in ROWS:
        if mine == dict(ROW_DEFAULTS, **{k: v for k, v in over.items() if k in ROW_DEFAULTS}):
            return name
    return None


def measured_traffic(a, launches, k_iters):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/bench_<config>_<mode>_pmc.json, written by tools/collect_profiles.py: FETCH_SIZE and WRITE_SIZE in
    separate --pmc passes; FETCH_SIZE doubled -- on gfx950 it counts half the bytes of wide streaming reads,
    MI355X_MICROARCH.md, HBM section).  Resident kernels are recorded per iteration and scaled to this run's launches.
    Only for shapes that were profiled (the config's default, or an --n-groups variant with a committed summary); otherwise None.  NOT measured in the run that prints it (see traffic_source)."""
    tag = profile_tag(a)
    if tag is None:
        return None, None
    for rnd in (PROFILE_ROUND,):
        path = os.path.join(ROOT, "profiles", rnd, f"bench_{tag}_pmc.json")
        try:
            rec = json.load(open(path))["dominant"]
        except (OSError, KeyError, ValueError):
            continue
        if not profile_is_current(path):
            return None, f"profiles/{rnd}/{os.path.basename(path)} was collected on other kernel sources: not quoted"
        src = f"profiles/{rnd}/{os.path.basename(path)} (rocprofv3 --pmc passes of this command, not this run)"
        LAST_PROFILE_REC.clear()
        LAST_PROFILE_REC.update(rec)
        if "bytes_per_launch" in rec:
            return float(rec["bytes_per_launch"]), src
        return float(rec["bytes_per_iteration"]) * k_iters / max(1, launches), src
    return None, None


def longrow_shape_roof_ms():
    """(subject-sweep + hyper-parameter-sweep launch of the traffic probe) / 2 in ms, from the committed probe output; None if absent"""
    import re
    try:
        txt = open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "longrow_traffic_probe.txt")).read()
    except OSError:
        return None
    us = {}
    for kind in ("subject sweep", "hyper sweep"):
        m = re.search(re.escape(kind) + r"[^\n]*WG 256 x 2/CU[^\n]*work   0 ahead 1 \|\s+([0-9.]+) us/launch", txt)
        if not m:
            return None
        us[kind] = float(m.group(1))
    return (us["subject sweep"] + us["hyper sweep"]) / 2.0 * 1e-3


LAST_PROFILE_REC = {}  # the `dominant` record measured_traffic() read last (FETCH / WRITE split for the rows that quote it)


def roofline_of(a, w, tm, k_iters, P, dt_per_iter):
    """`roofline` object for the dominant kernel.  tm = per-class device time from HIP events recorded on the TIMED
    iterations (on the stream the kernels run on); k_iters = the timed iterations; dt_per_iter = wall seconds per step."""
    D = w["D"]
    phases = 2 if a.schedule == "two_colour" else 1
    sweeps = 1 if w["masks"] is None else len(w["masks"])
    per_kernel = {n: v["ms"] / k_iters for n, v in tm.items()}
    launches = {n: v["launches"] for n, v in tm.items()}
    fused_ms = tm["propose"]["ms"] + tm["loglike_prep"]["ms"] + tm["accept_store"]["ms"]
    if a.config in ("cfg2", "cfg3"):
        N, d = w["dims"]
        if a.mode == "streaming":
            streamed_in_k1 = tm["loglike"]["launches"] == 0  # the observation stream ran inside the resident proposal kernel
            t_s = (fused_ms if streamed_in_k1 else tm["loglike"]["ms"]) * 1e-3
            n_launch = max(1, tm["propose"]["launches"] if streamed_in_k1 else tm["loglike"]["launches"])
            executed = 2.0 * N * d * P * k_iters          # expanded quadratic form: 2ND flop per particle-update
            survey = (3.0 * N * d + 2.0 * d * d) * P * k_iters  # SURVEY 8d's whitened-form count, for comparison only
            ach = executed / t_s / 1e12
            kern = ("k_propose<...,RES> with the observation stream inside (v_mfma_f64_16x16x4_f64)" if streamed_in_k1
                    else f"k_cross_mfma<{max(1, 1 << max(0, (max(1, (d + 3) // 4) - 1).bit_length())) if d <= 64 else 16},4> (v_mfma_f64_16x16x4_f64)")
            traffic, src = measured_traffic(a, n_launch, k_iters)
            # operands of one launch: X once per colour phase + the proposals' y rows
            alg_bytes = (8.0 * N * d * phases + P * 8.0 * d) * k_iters / n_launch
            rf = dict(bound="mfma", kernel=kern, achieved=ach, peak=PEAK_FP64_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP64_TFLOPS,
                      # `frac` = frac_executed: the MFMA flop the kernel really issues.  frac_survey prices SURVEY 8d's unit (3ND + 2D^2
                      # per update) over the same time: ABOVE 1, because this mode does not do the survey's per-pair work -- its
                      # [proposals x D].[D x N] product is analytically zero after centring (what_is_streamed)
                      frac_executed=ach / PEAK_FP64_TFLOPS, frac_survey=survey / t_s / 1e12 / PEAK_FP64_TFLOPS,
                      label="SUFFSTAT + a null GEMM: not the reference's per-pair work (that is --mode direct, the headline)",
                      flop_counted="executed 2*N*D per particle-update (every proposal x observation pair on the matrix cores)",
                      what_is_streamed="the cross term sum_i y.x~_i of the EXPANDED quadratic form, y = Sigma^-1 (theta' - xbar), x~_i = x_i - xbar. "
                                       "After centring it equals y.(sum_i x~_i), i.e. rounding noise around zero: STREAMING = SUFFSTAT + this "
                                       "[proposals x D].[D x N] product; the data dependence of the result sits in the data-only constant "
                                       "sum_i x~_i' Sigma^-1 x~_i and in N mu~' Sigma^-1 mu~.  The per-pair work that does not collapse is "
                                       "`--mode direct` (3*N*D per particle-update on the FP64 vector pipe)",
                      survey_equivalent_tflops=survey / t_s / 1e12,
                      survey_note="SURVEY 8d counts the whitened form 3ND+2D^2; the expanded form executes 2/3 of it for the same result",
                      launch_ms=t_s / n_launch * 1e3, launches=n_launch, updates_per_launch=P * k_iters / n_launch,
                      traffic=traffic, traffic_source=src,
                      wasted_traffic_ratio=None if traffic is None else traffic / alg_bytes)
        elif a.mode == "direct":
            in_k1 = tm["loglike"]["launches"] == 0  # small populations: the residual loop runs inside the streaming-resident lean kernel
            t_s = (fused_ms if in_k1 else tm["loglike"]["ms"]) * 1e-3
            n_launch = max(1, tm["propose"]["launches"] if in_k1 else tm["loglike"]["launches"])
            dp = 8 if d <= 8 else 16 if d <= 16 else 32 if d <= 32 else 64
            # SURVEY 8(d)'s unit: 3*N*D per (proposal, observation, dimension) -- one subtraction, one fused multiply-add -- and 2*D^2
            # for m = L^-1 mu~ (K1's preparation on the matrix cores: it runs in k_propose, outside this kernel's time, and is 0.02 %
            # of the count).  `frac` prices the flop of THIS kernel over THIS kernel's device time: 3*N*d, d the data dimension
            # (the kernel also runs the zero-padded dimensions up to dp; they are not counted).
            survey = 3.0 * N * d * P * k_iters
            ach = survey / t_s / 1e12
            clk = getattr(a, "clock", None) or {}
            mhz = clk.get("mhz_median")
            traffic, src = measured_traffic(a, n_launch, k_iters)
            # what one launch must read and write: the whitened rows once (they are shared by every proposal: L2 / MALL serve the re-reads),
            # the proposals' m rows, the per-chunk partial sums
            alg_bytes = 8.0 * N * dp + (P / phases) * 8.0 * dp + (P / phases) * 8.0 * 48  # (48 chunks at cfg3: 12.6 MB of partial sums)
            rf = dict(bound="valu", kernel=(f"k_res_mvn<256,true,{dp},direct> (streaming-resident lean kernel: the residual loop out of an LDS copy of the "
                                            "workgroup's chunk of whitened rows, every iteration up to the next migration in one launch)" if in_k1 else
                                            f"k_direct_mvn<{dp}> (thread per proposal, m = L^-1 mu~ in registers, wave-uniform z rows)"),
                      achieved=ach, peak=PEAK_FP64_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP64_TFLOPS,
                      flop_counted="3*N*D per particle-update (SURVEY 8d's whitened residual form: v_add_f64 + v_fma_f64 per dimension; "
                                   "an add counts one flop, so 0.75 of the FMA peak is this form's ceiling at the nominal clock)",
                      survey_tflops_whole_step=(3.0 * N * d + 2.0 * d * d) * P * k_iters / (dt_per_iter * k_iters) / 1e12,
                      # the clock the vector pipe held under this kernel (s_memtime over s_memrealtime between the first and the last workgroup to
                      # finish on the same CU, median over the CUs, last timed launch: demc_timing_clock): the peak above is quoted at 2400 MHz, the chip holds less under a dense FP64
                      # loop and devices differ (MI355X_MICROARCH.md, DVFS give-back) -- this is what moves `frac` between boxes
                      shader_clock_mhz=mhz, shader_clock_mhz_min=clk.get("mhz_min"), shader_clock_mhz_max=clk.get("mhz_max"),
                      peak_at_clock=None if not mhz else PEAK_FP64_TFLOPS * mhz / 2400.0,
                      frac_at_clock=None if not mhz else ach / (PEAK_FP64_TFLOPS * mhz / 2400.0),
                      mix_ceiling_at_clock=None if not mhz else 0.75 * mhz / 2400.0,
                      frac_of_mix_ceiling_at_clock=None if not mhz else ach / (0.75 * PEAK_FP64_TFLOPS * mhz / 2400.0),
                      sclk_sysfs_mhz=getattr(a, "sclk_sysfs", None),
                      launch_ms=t_s / n_launch * 1e3, launches=n_launch, updates_per_launch=P * k_iters / n_launch,
                      traffic=traffic, traffic_source=src,
                      wasted_traffic_ratio=None if (traffic is None or in_k1) else traffic / alg_bytes,  # (the resident form: a launch spans iterations)
                      # what the counters' bytes are: every (block of 256 proposals, observation chunk) workgroup reads its proposals' m rows
                      # (64 KB) -- the m rows are read once per CHUNK, by the decomposition that fills the chip (cfg3: 48 chunks x 8.4 MB)
                      implied_m_rereads=None if (traffic is None or in_k1) else (traffic - 8.0 * N * dp - (P / phases) * 8.0 * 48) / ((P / phases) * 8.0 * dp),
                      traffic_note="a VALU-bound kernel (the vector pipe issues on 0.94 of its cycles): what the counters see is mostly the proposals' m "
                                   "rows, read once per observation chunk by every block of 256 proposals (cfg3: 48 chunks x 8.4 MB = 403 MB per launch, "
                                   "65 GB/s -- far from the 8 TB/s roof); the whitened rows themselves (25.6 MB) stay with one XCD's L2 per chunk since "
                                   "round 6's workgroup -> (block, chunk) mapping")
        else:
            t_s = fused_ms * 1e-3
            n_launch = max(1, tm["propose"]["launches"])
            byts = (24.0 * D + 17.0) * P * k_iters
            ach = byts / t_s / 1e9
            LAST_PROFILE_REC.clear()
            traffic, src = measured_traffic(a, n_launch, k_iters)
            rf = dict(bound="hbm", kernel="k_propose with the fused prep/accept/store tail" +
                      (", resident form (one launch per run of iterations between migrations)" if n_launch < phases * k_iters else ""),
                      achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                      bytes_counted="24*D+17 per particle-update (SURVEY 8d)", launch_ms=t_s / n_launch * 1e3, launches=n_launch,
                      updates_per_launch=P * k_iters / n_launch,
                      traffic=traffic, traffic_source=src,
                      wasted_traffic_ratio=None if traffic is None else traffic / (byts / n_launch),
                      # `frac` prices SURVEY 8d's formula, which counts bytes the resident kernel never moves (theta is read from
                      # HBM once per launch, not once per update); the fraction of the HBM peak by the COUNTERS' bytes beside it
                      counter_frac=None if traffic is None else traffic * n_launch / t_s / 1e9 / PEAK_HBM_GBS)
            if a.partners == "history":
                # resample (crossover.jl:113-124): two partner rows (three with snooker) GATHERED from the history of all
                # particles, next to the particle's own row -- 3 x 8D bytes read per update; the counters say what the gather costs
                rf["gather_bytes_per_update"] = 3 * 8.0 * D
                fb = LAST_PROFILE_REC.get("fetch_bytes_per_launch")
                rf["fetch_bytes_per_update"] = None if fb is None else fb / (P * k_iters / n_launch)
                one = not tm["accept_store"]["launches"]  # past burn-in no base particle is read: the whole update in ONE launch
                rf["kernel"] = ("k_res_mvn<...,HIST> where the default sampler runs on MvNormal-full in SUFFSTAT mode (the lean body, partner rows "
                                "= history cells [row][slot][D] read straight from HBM, one launch per iteration), else k_propose<256,false,...>"
                                if one else
                                "k_propose<256,false,...> (no tile: partner rows are history cells [row][slot][D], each D contiguous doubles) "
                                "+ k_accept_store (burn-in: the base particle is read from the current population)")
    elif a.config == "mvn30":
        # test/multivariate_normal_tests.jl:50-59 as the reference runs it (DE-MC_Z + snooker), scaled to fill the chip: the rows are
        # 31 doubles, the 100 x 30 observations sit in LDS -- per update the path moves 24*D+17 bytes plus, with history
        # partners, the gathered partner cells (two rows of D, three for a snooker update)
        t_s = (fused_ms + tm["loglike"]["ms"]) * 1e-3
        n_launch = max(1, tm["propose"]["launches"])
        gather = 0.0 if a.partners != "history" else (2.0 + (a.snooker or 0.0)) * 8.0 * D
        byts = (24.0 * D + 17.0 + gather) * P * k_iters
        ach = byts / t_s / 1e9
        LAST_PROFILE_REC.clear()
        traffic, src = measured_traffic(a, n_launch, k_iters)
        rf = dict(bound="hbm", kernel=("k_res_mvn<...,HIST,iso> (the lean DE-MC_Z body, isotropic form: one launch per iteration, partner rows = history cells)"
                                       if gather and a.mode == "suffstat" else "k_propose (general kernel) with the fused MvNormal tail"),
                  achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                  bytes_counted="24*D+17 per particle-update (SURVEY 8d)" + (" + the gathered history cells (2 rows of D, 3 for a snooker update)" if gather else ""),
                  launch_ms=t_s / n_launch * 1e3, launches=n_launch, updates_per_launch=P * k_iters / n_launch,
                  traffic=traffic, traffic_source=src, wasted_traffic_ratio=None if traffic is None else traffic / (byts / n_launch))
        fb = LAST_PROFILE_REC.get("fetch_bytes_per_launch")
        rf["fetch_bytes_per_update"] = None if fb is None else fb / (P * k_iters / n_launch)
        rf["gather_bytes_per_update"] = gather
    elif a.config == "cfg1":
        t_s = fused_ms * 1e-3
        n_launch = max(1, tm["propose"]["launches"])
        byts = (24.0 * D + 17.0) * P * k_iters
        ach = byts / t_s / 1e9
        rf = dict(bound="hbm", kernel="k_res_obs<256>: the whole update of a group by one workgroup, sixteen lanes per particle, resident "
                                      "over the iterations between two migrations",
                  achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                  bytes_counted="24*D+17 per particle-update (SURVEY 8d); four workgroups on a 256-CU chip: the number says how little "
                                "of the chip 40 particles can use, not how good the kernel is -- latency per iteration is the figure of merit",
                  us_per_iteration=t_s / k_iters * 1e6, launch_ms=t_s / n_launch * 1e3, launches=n_launch,
                  updates_per_launch=P * k_iters / n_launch, traffic=None, traffic_source=None, wasted_traffic_ratio=None)
    elif a.config == "cfg4":
        S = w["dims"][0]
        t_s = (fused_ms + tm["loglike"]["ms"]) * 1e-3
        n_launch = max(1, tm["propose"]["launches"])
        # SURVEY 8d's formula, per particle-update and block sweep: read theta (8D) + write theta' (8D) + history row (8D) + 17
        # + the subjects' data (16 S).  It overstates what the path needs: the reference stores the history row once per
        # ITERATION (store_samples!, utilities.jl:161-180), not once per sweep; the hyper-parameter sweep rewrites two scalars,
        # not a row; and the 16 S bytes of counts are shared by every particle (L2).  What an iteration must move through HBM
        # per particle: its row in (8D) for each sweep, the accepted row out (<= 8D, subject sweep only) and ONE history
        # row (8D) -- `necessary_bytes`.  Both are reported; `frac` (the contract's field) is the honest one: counter traffic.
        survey_bytes = sweeps * (24.0 * D + 17.0 + 16.0 * S) * P * k_iters
        necessary_bytes = (sweeps * 8.0 * D + 8.0 * D + 8.0 * D + 17.0 * sweeps) * P * k_iters
        # DE-MC_Z (resample, crossover.jl:113-124; Examples/Hierarchical_Example.jl:103-114): the partner rows are cells of the
        # HISTORY of all particles -- 80 KB rows nobody else reads at that moment, so they DO cross the HBM interface: two per update
        # of the subject sweep (three for a snooker update, which also needs them in the hyper-parameter sweep for adjust_loglike's
        # norms, crossover.jl:268-273); the hyper-parameter sweep of a crossover update reads two scalars of each
        gather_bytes = 0.0
        if a.partners == "history":
            snk = a.snooker if a.snooker is not None else (w["engine"].get("theta_snooker") or 0.0)
            gather_bytes = ((1.0 - snk) * 2.0 + snk * 3.0 * sweeps) * 8.0 * D * P * k_iters
            necessary_bytes += gather_bytes
        LAST_PROFILE_REC.clear()
        traffic, src = measured_traffic(a, n_launch, k_iters)
        # (launch_phase's rule, 256 CUs: enough moving particles for two workgroups per CU -> the row-streaming kernel, both instances)
        moving = P // 2 if a.partners == "current" else P
        inst = "k_longrow<512>, one particle per CU"
        if moving >= 512 and w["masks"] is not None:
            inst = ("k_frozen_sweep<256,big> for the subject sweep + k_frozen_sweep<256> for the hyper-parameter sweep (rows streamed, no LDS "
                    "row, three / four workgroups per CU)")
        ach_survey = survey_bytes / t_s / 1e9
        ach_traffic = None if traffic is None else traffic * n_launch / t_s / 1e9
        rf = dict(bound="hbm", kernel=inst + " (a workgroup per particle, one pass: proposal, prior, subject terms, accept, store)",
                  achieved=ach_traffic if ach_traffic is not None else necessary_bytes / t_s / 1e9,
                  peak=PEAK_HBM_GBS, unit="GB/s",
                  frac=(ach_traffic if ach_traffic is not None else necessary_bytes / t_s / 1e9) / PEAK_HBM_GBS,
                  bytes_counted=("HBM bytes from the FETCH_SIZE / WRITE_SIZE counters of the profiled run / device time of this run"
                                 if ach_traffic is not None else
                                 "necessary bytes: row in per sweep + accepted row out + ONE history row per iteration (no profiled traffic for this shape)"),
                  necessary_gbs=necessary_bytes / t_s / 1e9, necessary_frac=necessary_bytes / t_s / 1e9 / PEAK_HBM_GBS,
                  survey_formula_gbs=ach_survey, survey_formula_frac=ach_survey / PEAK_HBM_GBS,
                  survey_formula=f"{sweeps} block sweeps x (24*D+17 + 16*S) per particle-update (SURVEY 8d; counts the history row per sweep and L2-resident data)",
                  launch_ms=t_s / n_launch * 1e3, launches=n_launch, updates_per_launch=sweeps * P * k_iters / n_launch,
                  traffic=traffic, traffic_source=src,
                  wasted_traffic_ratio=None if traffic is None else traffic / (necessary_bytes / n_launch))
        # what a bare kernel of this traffic shape reaches on this chip (tools/longrow_traffic_probe.hip: the kernel's row traffic and
        # occupancy, no arithmetic; profiles/<round>/longrow_traffic_probe.txt, measured at the whole cfg4's size): the average of
        # a subject-sweep and a hyper-parameter-sweep launch is the roof of the SHAPE -- quoted next to the 8 TB/s fraction
        roof = longrow_shape_roof_ms() if (a.partners == "current" and w["G"] == 128 and w["Np"] == 32 and S == 10000 and sweeps == 2) else None
        if roof is not None:
            rf["shape_roof_launch_ms"] = roof
            rf["shape_frac"] = roof / (t_s / n_launch * 1e3)
            rf["shape_roof_source"] = f"profiles/{PROFILE_ROUND}/longrow_traffic_probe.txt (subject sweep + hyper-parameter sweep) / 2"
        if a.partners == "history":
            rf["gather_bytes_per_update"] = gather_bytes / (sweeps * P * k_iters)
            fb = LAST_PROFILE_REC.get("fetch_bytes_per_launch")
            rf["fetch_bytes_per_update"] = None if fb is None else fb / (sweeps * P * k_iters / n_launch)
            if fb is None and LAST_PROFILE_REC.get("fetch_bytes_per_iteration") is not None:  # (two kernels alternate: reduced per iteration)
                rf["fetch_bytes_per_update"] = LAST_PROFILE_REC["fetch_bytes_per_iteration"] / (sweeps * P)
    else:  # cfg5
        N, na = w["dims"]
        t_s = tm["loglike"]["ms"] * 1e-3
        n_launch = max(1, tm["loglike"]["launches"])
        evals = float(N) * P * k_iters
        # executed FP64 flop per (trial, proposal): counted from the compiler's assembly of the shipped loop (an FMA two flop,
        # add / mul / max / min / rcp one; tools/count_lba_flop.py -> profiles/<round>/lba_inner_loop.json), not by hand
        inner, inner_src = {}, None
        pj = os.path.join(ROOT, "profiles", PROFILE_ROUND, "lba_inner_loop.json")
        if profile_is_current(pj):
            inner = json.load(open(pj))
            inner_src = f"profiles/{PROFILE_ROUND}/lba_inner_loop.json (static count over the batch loop of the shipped kernel)"
        else:  # the committed count belongs to other sources: count again (hipcc -S of the kernel header, ~20 s)
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "count_lba_flop.py")], capture_output=True, text=True, timeout=300)
                inner = json.loads(out.stdout)
                inner_src = "tools/count_lba_flop.py run now (the committed count was taken on other kernel sources)"
            except (OSError, ValueError, subprocess.SubprocessError):
                pass
        flop_per_eval = inner.get("fp64_flop_per_eval")
        ach = None if flop_per_eval is None else evals * flop_per_eval / t_s / 1e12
        pipes, pipes_src = {}, None
        try:
            pj = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"bench_{profile_tag(a)}_pipe_pmc.json")
            if profile_is_current(pj):
                pipes = [v for k, v in json.load(open(pj)).items() if "k_lba_wave" in k or "k_lba_loglike" in k or "k_obs_loglike" in k][0]
                pipes_src = f"profiles/{PROFILE_ROUND}/{os.path.basename(pj)} (rocprofv3 --pmc passes of this command, not this run)"
            else:
                pipes_src = f"profiles/{PROFILE_ROUND}/{os.path.basename(pj)} was collected on other kernel sources: not quoted"
        except (OSError, KeyError, ValueError, IndexError):
            pass
        clk = getattr(a, "clock", None) or {}
        mhz = clk.get("mhz_median")
        rf = dict(bound="valu", shader_clock_mhz=mhz, shader_clock_mhz_min=clk.get("mhz_min"), shader_clock_mhz_max=clk.get("mhz_max"),
                  frac_at_clock=None if (not mhz or ach is None) else ach / (PEAK_FP64_TFLOPS * mhz / 2400.0),
                  kernel="k_lba_wave (LBA: a wave per proposal, lanes across trials sorted by (choice, decision time), batches of 8 "
                                       "trials per lane, one degree-7 Phi polynomial table on intervals of 1/32 in LDS whose derivative gives phi -- "
                                       "read as broadcasts --, one log per batch)",
                  achieved=ach, peak=PEAK_FP64_TFLOPS, unit="TFLOP/s", frac=None if ach is None else ach / PEAK_FP64_TFLOPS,
                  flop_counted=None if flop_per_eval is None else f"{flop_per_eval:.1f} executed FP64 flop per (trial, proposal) evaluation x N x proposals",
                  flop_source=inner_src, static_valu_insts_per_eval=inner.get("valu_insts_per_eval"),
                  lds_reads_per_eval=inner.get("lds_reads_per_eval"),
                  valu_busy_frac=pipes.get("valu_busy_frac"), lds_busy_frac=pipes.get("lds_busy_frac"),
                  measured_valu_insts_per_eval=(pipes["SQ_INSTS_VALU_mean"] * 64.0 / (float(N) * P * k_iters / n_launch)
                                                if "SQ_INSTS_VALU_mean" in pipes else None), pipe_source=pipes_src,
                  limiting_pipe="the FP64 vector pipe (~190 vector instructions per evaluation); until round 5 the LDS pipe: with lanes = proposals "
                                "a population that had not converged read another table row in every lane",
                  trial_proposal_evaluations_per_s=evals / t_s, launch_ms=t_s / n_launch * 1e3, launches=n_launch,
                  updates_per_launch=P * k_iters / n_launch, traffic=None, traffic_source=None, wasted_traffic_ratio=None)
    rf["timing"] = ("HIP events recorded on the timed iterations, on the stream the kernels run on" if getattr(a, "events_inline", not two_pass(a)) else
                    "HIP events on the stream the kernels run on, recorded over a REPEAT of the timed iterations (a fresh engine from the same "
                    "start: the same chain): `value` and `ms_per_step` come from the un-instrumented pass (an event pair per dispatch costs "
                    "~6 us per launch)")
    rf["per_kernel_ms_per_iter"] = per_kernel
    rf["launches_by_class"] = launches
    rf["device_ms_per_iter"] = sum(per_kernel.values())
    rf["ms_per_step"] = dt_per_iter * 1e3
    return rf


# ----------------------------------------------------------------------------------------------------------------
# accuracy half of the metric
# ----------------------------------------------------------------------------------------------------------------
def accuracy_leg(a, w, demc_amd, local, rng):
    """posterior-mean L1 against the closed-form conjugate posterior (MvNormal configs), from an UNTIMED run that does not
    depend on --steps: the same sampler configuration run past the reference's burn-in BY THE KERNELS THAT WERE TIMED
    (the leg runs in the likelihood mode of the timed region: 1500 DIRECT iterations of the headline are ~21 s; rounds 1-3 ran it
    in SUFFSTAT mode, which makes the same decisions at a fraction of the cost).  The ensemble of all particles is averaged over the
    last iterations."""
    if a.accuracy_iters <= 0 or "posterior_mean" not in w:
        return None
    from demc_amd import workloads as W
    G, Np, D = w["G"], w["Np"], w["D"]
    P = G * Np
    n_it = max(a.accuracy_iters, a.burnin + 200)
    leg_mode = a.mode
    eng = demc_amd.HipEngine(n_groups=G, Np=Np, D=D, n_rows=0, store_history=0, schedule=2 if a.schedule == "two_colour" else 1,
                             seed=20260001, device_id=local, burnin=a.burnin, loglike_mode=MODES[leg_mode], trace=0, **w["engine"])
    W.configure(eng, w)
    eng.set_state(w["init"](P, rng))
    t0 = time.perf_counter()
    n_snap = 10
    eng.step(1, n_it - n_snap * 10)
    kern = eng.last_kernels()
    snaps = []
    it = 1 + n_it - n_snap * 10
    for _ in range(n_snap):
        eng.step(it, 10)
        it += 10
        snaps.append(eng.get_state()[0])
    dt = time.perf_counter() - t0
    eng.close()
    import numpy as np
    th = np.concatenate(snaps)
    m, sd = w["posterior_mean"], w["posterior_sd"]
    return dict(posterior_mean_l1_rel=float(np.abs(th.mean(0) - m).sum() / np.abs(m).sum()),
                max_abs_err_in_posterior_sd=float(np.max(np.abs(th.mean(0) - m) / sd)),
                ensemble_sd_over_posterior_sd=float(np.median(th.std(0) / sd)),
                leg=f"untimed run of {n_it} iterations (burn-in {a.burnin}) of the same sampler on this GPU, {leg_mode.upper()} likelihood: " +
                    (f"THE KERNELS OF THE TIMED REGION ({kern})" if leg_mode == a.mode else
                     f"DIFFERENT KERNELS than the timed ones ({kern}), same proposals and accept decisions "
                     f"(tests/test_gpu_production.py::test_the_three_likelihood_modes_make_the_same_decisions)") +
                    f"; all {P} particles at {n_snap} snapshots 10 iterations apart",
                leg_mode=leg_mode,
                seconds=dt, reference="closed-form Gaussian posterior (conjugate: prior N(0,I), known Sigma)")


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline: the C restatement of the reference algorithm and schedule on the host cores
# ----------------------------------------------------------------------------------------------------------------
def usable_cores():
    """host cores this process may actually use: the scheduler affinity, cut down to the cgroup's CPU quota (a 1-GPU box
    shows all 256 logical CPUs of the host in its affinity mask but is given 16 CPUs' worth of time)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(a, w, seconds_target=14.0, seconds_single=3.0):
    """oracle source built -O3 -march=native -fopenmp on this host, reference schedule (sequential in-place sweep per group,
    one group per OpenMP thread like p_update!, src/main.jl:135-148) on a bounded sample of the same workload: timed on
    ONE thread and on all the cores this process may use."""
    import numpy as np
    from demc_amd import workloads as W
    from oracle import oracle as O
    O.use_native_build()
    cores = usable_cores()
    Np, D = w["Np"], w["D"]
    n_sweeps = 1 if w["masks"] is None else len(w["masks"])
    rng = np.random.default_rng(1234)

    def run(n_groups, threads, seconds):
        o = O.Oracle(n_groups=n_groups, Np=Np, D=D, schedule=0, n_rows=0, store_history=0, seed=7, n_threads=threads,
                     burnin=a.burnin, **w["engine"])
        W.configure(o, w)
        o.set_state(w["init"](n_groups * Np, rng))
        t0 = time.time()
        o.step(1, 1)
        t1 = time.time() - t0
        iters = int(max(1, min(50, seconds / max(t1, 1e-4))))
        t0 = time.time()
        o.step(2, iters)
        dt = time.time() - t0
        o.close()
        return n_groups * Np * n_sweeps * iters / dt, iters  # (like `value`: every block sweep updates every particle once)

    one_thread, it1 = run(1, 1, seconds_single)
    ng = max(2, min(cores, 4 * w["G"]))
    value, iters = run(ng, cores, seconds_target)
    cpu_model, host_cores = "unknown", os.cpu_count()
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass
    sweeps = 1 if w["masks"] is None else len(w["masks"])
    return dict(value=value, unit="particle-updates/s", cores=cores, kind="port", value_single_thread=one_thread,
                cpu_model=cpu_model, host_logical_cpus=host_cores,
                sample=f"{w['name']} shape (D={D}, data {list(w['dims'])}, Np={Np}, {sweeps} sweep(s) per iteration) on {ng} groups, "
                       f"{iters} iterations on {cores} threads (= every core this process may use: affinity cut to the cgroup CPU quota) and 1 group x {it1} iterations "
                       f"on one thread; reference schedule (sequential in-group sweep, one group per OpenMP thread), every "
                       f"proposal visits every observation; gcc -O3 -march=native -fopenmp")


def control_plane_store(rank, world, timeout_s=120.0):
    """Control plane of the library collective = the launcher's key-value store and nothing else: it carries the
    communicator id from rank 0 to the others (no process group is created).  Under torch.distributed.run the agent
    hosts the store at MASTER_ADDR:MASTER_PORT and every rank is a client; started bare (spawn_ranks) rank 0 hosts it."""
    from datetime import timedelta
    from torch.distributed import TCPStore
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    return TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, is_master=(rank == 0 and not agent),
                    timeout=timedelta(seconds=timeout_s))


def exchange_comm_id(store, rank, make_id, tag=""):
    """rank 0 draws the id (make_id() -> 128 bytes, demc_comm_unique_id) and publishes it; everybody reads it back
    (`tag`: one key per communicator -- the sharded rows create their own)"""
    key = "demc/comm_id/" + tag + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if rank == 0:
        store.set(key, make_id())
    return bytes(store.get(key))


def agree_on_init(store, rank, world, error, tag=""):
    """every rank posts how its communicator creation ended and reads everybody's: None if all succeeded, else the first
    failure's text -- so that a CLEAN failure (an error code, not a hang: those are the watchdog's) takes every rank down
    the same road"""
    key = "demc/init/" + tag + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") + "/"
    store.set(key + str(rank), b"ok" if error is None else ("rank %d: %s" % (rank, error)).encode()[:400])
    states = [bytes(store.get(key + str(r))).decode("utf-8", "replace") for r in range(world)]
    bad = [x for x in states if x != "ok"]
    return bad[0] if bad else None


# ----------------------------------------------------------------------------------------------------------------
# one measurement: warm-up, the timed iterations (HIP events around every launch), the roofline of the dominant kernel
# ----------------------------------------------------------------------------------------------------------------
MODES = {"streaming": 0, "suffstat": 1, "direct": 2}


def start_rows(a, w, P, rng):
    """starting rows: prior draws like sample_init (main.jl:263-271), or -- `--start posterior` -- a population that has
    already converged: the generating parameters with a relative spread of one per cent (the regime a long run spends
    its time in; for the LBA the lanes of a wave then read neighbouring rows of the Phi table)"""
    if a.start == "prior" or "truth" not in w:
        return w["init"](P, rng)
    import numpy as np
    t = np.asarray(w["truth"], dtype=np.float64)
    if "truth_sd" in w:  # (an absolute spread per scalar where the workload gives one: scalars near zero need it)
        th = t + np.asarray(w["truth_sd"], dtype=np.float64) * rng.standard_normal((P, t.size))
    else:
        th = t * (1.0 + 0.01 * rng.standard_normal((P, t.size)))
    return np.minimum(np.maximum(th, np.asarray(w["lo"]) + 1e-9), np.asarray(w["hi"]) - 1e-9)


def is_plain_headline(a):
    """the command the driver runs: BASELINE's headline workload with nothing overridden"""
    return (a.config == "cfg3" and a.mode == "direct" and a.partners == "current" and a.start == "prior" and
            a.n_groups is None and a.Np is None and a.nobs is None and a.dim is None and a.snooker is None and not a.fuse)


def two_pass(a):
    """Where the per-kernel HIP events are recorded.  The headline keeps them INSIDE its timed region (its three launches per colour
    phase are milliseconds long: the events cost 0.6 %).  Every other workload times its K steps un-instrumented and then repeats
    them with the events on: a start / stop event pair in every dispatch packet costs ~6 us per launch, which is 13 % of the cfg4
    share's wall time and 19 % of DE-MC_Z's (one 21 us launch per iteration) -- a cost of the measurement, not of the path.
    The repeat runs on a FRESH engine from the same start (same seed, same rows: the same chain, iteration for iteration), so the
    kernel durations belong to the iterations that were timed; a multi-rank run repeats on the same engine instead."""
    return not a.no_roofline and not is_plain_headline(a)


def make_engine(a, w, demc_amd, local, rank=0, world=1, seed=20260001):
    import numpy as np
    G, Np, D = w["G"], w["Np"], w["D"]
    hist = a.partners == "history"
    if hist and a.n_initial < 1:
        raise SystemExit("bench.py: --partners history needs --n-initial >= 1 (utilities.jl:35-39: row 1 must exist)")
    n_rows = a.n_initial + a.warmup + a.steps * (2 if (two_pass(a) and world > 1) else 1)
    cfg = dict(n_groups=G, Np=Np, D=D, n_rows=n_rows, n_initial=a.n_initial,
               schedule=1 if (hist or a.schedule == "synchronous") else 2, partner_kind=1 if hist else 0,
               group_offset=rank * G, n_groups_total=G * world, seed=seed, device_id=local, burnin=a.burnin,
               loglike_mode=MODES[a.mode] if a.config in ("cfg2", "cfg3", "mvn30") else 0, trace=0, fuse=a.fuse)
    cfg.update(w["engine"])
    eng = demc_amd.HipEngine(**cfg)
    from demc_amd import workloads as W
    W.configure(eng, w)
    rng = np.random.default_rng(20260003 + rank)
    P = G * Np
    if a.n_initial > 0:  # initialize_samples (utilities.jl:35-39): rows 1:n_initial are independent prior draws
        eng.set_history_rows(0, np.stack([w["init"](P, rng) for _ in range(a.n_initial)]))
    eng.set_state(start_rows(a, w, P, rng))
    return eng


def measure_row(name, a, w, demc_amd, local):
    """one extra SURVEY 8(d) row on this GPU: a fresh engine, warm-up, the timed iterations, closed again"""
    import numpy as np
    import torch
    t_wall = time.perf_counter()
    P = w["G"] * w["Np"]
    it0 = 1 + a.n_initial

    def run_once(instrument):
        e = make_engine(a, w, demc_amd, local)
        e.step_enqueue(it0, a.warmup)
        e.synchronize()
        torch.cuda.synchronize()
        if instrument:
            e.timing_enable(True)
        t0 = time.perf_counter()
        e.step_enqueue(it0 + a.warmup, a.steps)
        e.synchronize()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        tm_ = e.timing_read() if instrument else None
        if instrument:
            a.clock = e.timing_clock()  # (DIRECT rows: the clock the likelihood kernel held; None otherwise)
            e.timing_enable(False)
        return e, dt_, tm_

    if two_pass(a):  # K steps un-instrumented for `value`; the same K steps of the same chain again, on a fresh engine, with the events
        eng2, _, tm = run_once(True)
        eng2.close()
        eng, dt, _ = run_once(False)
    else:
        eng, dt, tm = run_once(True)
    kernels = eng.last_kernels()
    n_rows = a.n_initial + a.warmup + a.steps
    k_last = min(10, a.steps)
    _, acc_h, _, _ = eng.get_history(n_rows - k_last, n_rows)
    _, w_now, _ = eng.get_state()
    eng.close()
    sweeps = 1 if w["masks"] is None else len(w["masks"])
    rf = roofline_of(a, w, tm, a.steps, P, dt / a.steps)
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_executed", "frac_survey", "label", "shader_clock_mhz",
            "frac_at_clock", "launch_ms", "launches", "traffic", "traffic_source",
            "wasted_traffic_ratio", "flop_counted", "bytes_counted", "counter_frac", "necessary_frac", "survey_formula_frac",
            "valu_busy_frac", "lds_busy_frac", "device_ms_per_iter", "gather_bytes_per_update", "fetch_bytes_per_update", "shape_frac",
            "shape_roof_launch_ms", "shape_roof_source", "necessary_gbs")
    value = P * sweeps * a.steps / dt
    return dict(name=name, workload=describe(a, w, 1), value=value, unit="particle-updates/s",
                particle_parameter_updates_per_s=value * w["D"], ms_per_step=dt / a.steps * 1e3, steps=a.steps, warmup=a.warmup,
                burnin=a.burnin, start=a.start, kernels=kernels, accept_rate=float(acc_h.mean()),
                finite_weights=bool(np.isfinite(w_now).all()), roofline={k: rf[k] for k in keep if k in rf},
                seconds=time.perf_counter() - t_wall)


# The other rows of SURVEY 8(d), as flag sets of this same program (tools/collect_profiles.py profiles exactly these commands
# and writes profiles/<round>/bench_<name>_line.json from them).  Steps are chosen so that every row takes a second or two.
ROWS = [
    # the expanded quadratic form on the matrix cores: SUFFSTAT + a [proposals x D].[D x N] product that is zero after centring (the
    # headline of rounds 1-5; `frac_executed` prices the MFMA flop it executes, `frac_survey` SURVEY 8d's unit -- above 1, which is
    # what "the kernel is not doing the survey's work" looks like)
    ("cfg3_streaming", dict(config="cfg3", mode="streaming", steps=20, warmup=5)),
    ("cfg3_streaming_post_burnin", dict(config="cfg3", mode="streaming", burnin=0, steps=20, warmup=5)),
    ("cfg3_suffstat", dict(config="cfg3", mode="suffstat", steps=400, warmup=50)),
    ("cfg3_suffstat_post_burnin", dict(config="cfg3", mode="suffstat", burnin=0, steps=400, warmup=50)),
    ("cfg3_suffstat_history_partners", dict(config="cfg3", mode="suffstat", partners="history", n_initial=16, steps=200, warmup=20)),
    ("cfg3_suffstat_history_partners_post_burnin", dict(config="cfg3", mode="suffstat", partners="history", n_initial=16, burnin=0,
                                                        steps=200, warmup=20)),
    ("cfg3_suffstat_history_partners_snooker", dict(config="cfg3", mode="suffstat", partners="history", n_initial=16, snooker=0.1, steps=200,
                                                    warmup=20)),  # DE-MC_Z as the reference's own runs configure it (theta_snooker = 0.1)
    ("cfg3_streaming_history_partners", dict(config="cfg3", mode="streaming", partners="history", n_initial=16, steps=20, warmup=5)),
    # cfg2 in the headline's mode: 2 048 particles cannot fill the chip with a K1 -> k_direct_mvn -> K3 chain per colour phase (six
    # dependent launches for 4.9e8 flop): a latency row, kept so that every MvNormal config is also measured in the form that does not collapse
    ("cfg2_direct", dict(config="cfg2", mode="direct", steps=400, warmup=50)),
    ("cfg2_streaming", dict(config="cfg2", mode="streaming", steps=400, warmup=50)),
    ("cfg2_streaming_post_burnin", dict(config="cfg2", mode="streaming", burnin=0, steps=400, warmup=50)),
    ("cfg4_share", dict(config="cfg4", steps=40, warmup=10)),
    ("cfg4_whole", dict(config="cfg4", n_groups=128, steps=20, warmup=5)),
    # the reference's OWN hierarchical configuration (Examples/Hierarchical_Example.jl:103-114): sample = resample (DE-MC_Z), theta_snooker
    # = 0.1, block updates [hyper; subject] -- the 80 KB partner rows are gathered from the history
    ("cfg4_whole_history_partners_snooker_blocks", dict(config="cfg4", n_groups=128, partners="history", n_initial=4, snooker=0.1,
                                                        steps=10, warmup=3)),
    # ... past burn-in (no base particle from the current population: nothing a particle reads is written in the launch)
    ("cfg4_whole_history_partners_snooker_blocks_post_burnin", dict(config="cfg4", n_groups=128, partners="history", n_initial=4,
                                                                    snooker=0.1, burnin=0, steps=10, warmup=3)),
    # test/multivariate_normal_tests.jl:50-59 (DE-MC_Z + snooker on MvNormal(mu, sigma^2 I), 31 parameters), groups scaled to fill the chip
    ("mvn30_demcz_snooker", dict(config="mvn30", mode="suffstat", Np=256, partners="history", n_initial=124, snooker=0.1, steps=100, warmup=20)),
    ("mvn30_demcz_snooker_post_burnin", dict(config="mvn30", mode="suffstat", Np=256, partners="history", n_initial=124, snooker=0.1, burnin=0,
                                             steps=100, warmup=20)),
    # ... from a converged population (the regime a long run spends its time in: next to nothing is accepted in 10^4 dimensions)
    ("cfg4_whole_converged", dict(config="cfg4", n_groups=128, start="posterior", burnin=0, steps=20, warmup=5)),
    ("cfg5_share", dict(config="cfg5", steps=20, warmup=5)),
    ("cfg5_share_converged", dict(config="cfg5", start="posterior", steps=20, warmup=5)),
    ("cfg1", dict(config="cfg1", steps=400, warmup=50)),
]


CPU_ROWS = ("cfg4_share", "cfg5_share")  # rows that also carry a cpu_baseline leg (one GPU's share of BASELINE's 8-GPU configs)

# what identifies a workload (steps / warm-up do not)
ROW_DEFAULTS = dict(config="cfg3", mode="direct", schedule="two_colour", n_groups=None, Np=None, nobs=None, dim=None,
                    burnin=1000, snooker=None, fuse=0, partners="current", n_initial=0, start="prior")


def row_args(a, name, over):
    """the argument set of a row: this run's defaults with the row's flags on top"""
    b = copy.copy(a)
    for k, v in ROW_DEFAULTS.items():
        setattr(b, k, v)
    for k, v in over.items():
        setattr(b, k, v)
    b.profile_tag = name
    return b


def row_flags(over):
    """the same as command-line flags (what tools/collect_profiles.py runs under rocprofv3)"""
    out = []
    for k, v in over.items():
        out += ["--" + k.replace("_", "-").lower(), str(v)]  # (Np -> --np)
    return out


def dict_args(a, **over):
    """a copy of the argument namespace with some fields replaced"""
    import argparse
    return argparse.Namespace(**dict(vars(a), **over))


def run_rows(a, w_headline, demc_amd, local, budget_s=120.0):
    want = None if a.rows in (None, "all") else set(a.rows.split(","))
    rows, t0, cpu_todo = [], time.perf_counter(), []
    # the headline's workload is reused only when it IS the plain one: sampler flags (--snooker ...) ride in w["engine"], and a
    # row that inherited them would be labelled and profile-tagged as the plain row while running another kernel
    cache = {("cfg3", None): w_headline} if is_plain_headline(a) else {}
    for name, over in ROWS:
        if want is not None and name not in want:
            continue
        if time.perf_counter() - t0 > budget_s:
            rows.append(dict(name=name, skipped=f"the rows' time budget of {budget_s:g} s was used up"))
            continue
        b = row_args(a, name, over)
        try:
            key = (b.config, b.n_groups)
            if key not in cache:
                cache[key] = build_workload(dict_args(b, snooker=None))
            wl = cache[key]
            if b.snooker is not None:  # (the sampler's settings ride in the workload's engine dict: a copy with this row's)
                wl = dict(wl, engine=dict(wl["engine"], theta_snooker=b.snooker))
            elif b.config in ("cfg2", "cfg3"):  # (cfg5's own theta_snooker = 0.1 is its workload's; the MvNormal rows' default is 0)
                assert not wl["engine"].get("theta_snooker"), (name, wl["engine"])
            rows.append(measure_row(name, b, wl, demc_amd, local))
            if name in CPU_ROWS and not a.no_cpu_baseline:
                cpu_todo.append((rows[-1], b, wl))
        except Exception as e:  # a row that fails is reported as failed; the headline stands
            rows.append(dict(name=name, error=f"{type(e).__name__}: {e}"[:500]))
    # the CPU figure of BASELINE's other configs (the reference's CPU path timed beside the GPU's): the oracle in
    # the reference's schedule on this row's workload, time-boxed, after every GPU row has been measured
    for row, b, wl in cpu_todo:
        try:
            row["cpu_baseline"] = cpu_baseline(b, wl, seconds_target=5.0, seconds_single=2.0)
            row["gpu_over_cpu"] = row["value"] / row["cpu_baseline"]["value"]
        except Exception as e:
            row["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    return rows


# BASELINE.json's configs that are DEFINED as 8-GPU runs, measured in the N > 1 line as SHARDED rows: the config's whole
# population partitioned over the ranks (strong partition: 128 / 512 groups in total whatever N is; p_update!'s groups side by
# side, main.jl:135-148), migration! across the ranks as ONE all-gather per migration event (migration.jl:11-19) on the row
# engine's own communicator behind the C-ABI -- or torch.distributed's when the run fell back to it.
SHARDED_ROWS = [
    ("cfg4_sharded", dict(config="cfg4", total_groups=128, steps=40, warmup=10)),  # hierarchical Binomial, Np 32, blocks [hyper; subject]
    ("cfg5_sharded", dict(config="cfg5", total_groups=512, steps=40, warmup=10)),  # LBA, Np 128, theta_snooker 0.1
]


def run_sharded_rows(a, demc_amd, local, rank, world, store, dist, barrier, reduce, stage, fallback):
    """After the headline's timed region (its engine and communicator still alive: `barrier` / `reduce` run on them), every
    rank builds its share of each config, creates the row's communicator (id through the launcher's store, agreed on like the
    headline's), and times K steps between two barriers; the line takes the MAX over the ranks.  A row that cannot be built
    on some rank is skipped by ALL ranks (agreed before the first collective); a rank's numbers are on disk
    (gpurun_out/rank<r>.json, `rows`) before the reduction that follows its timed region."""
    import torch
    rows, mine = [], []
    for name, spec in SHARDED_ROWS:
        t_wall = time.perf_counter()
        stage("row " + name, 150.0)
        total = spec["total_groups"]
        if total % world:
            rows.append(dict(name=name, skipped=f"{total} groups do not divide over {world} ranks"))
            continue
        b = row_args(a, name, dict(config=spec["config"], n_groups=total // world, steps=spec["steps"], warmup=spec["warmup"]))
        b.rows, b.no_roofline = "none", True
        eng = drv = None
        err = None
        lib_row = store is not None and dist is None
        try:
            w = build_workload(b)
            eng = make_engine(b, w, demc_amd, local, rank, world)
            if lib_row:
                eng.comm_init(exchange_comm_id(store, rank, eng.comm_unique_id, tag=name + "/"), rank, world)
                eng.comm_set_overlap(a.async_migration)
                step = eng.step_enqueue
            else:
                from demc_amd.distributed import ShardedDriver
                drv = ShardedDriver(eng, dist, torch.device("cuda", local), stream_ordered=True, async_migration=a.async_migration)
                step = drv.step
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        if store is not None:
            bad = agree_on_init(store, rank, world, err, tag=name + "/")
        else:  # (--collective torch from the start: no store of ours; a sum of failure flags over the process group)
            n_bad = reduce([0.0 if err is None else 1.0], "sum")[0]
            bad = None if n_bad == 0.0 else (err or "another rank failed to build the row")
        if bad is not None:
            if eng is not None:
                eng.close()
            rows.append(dict(name=name, error=short(bad, 300)))
            continue
        P = w["G"] * w["Np"]
        sweeps = 1 if w["masks"] is None else len(w["masks"])
        it0 = 1 + b.n_initial
        step(it0, b.warmup)
        eng.synchronize()
        barrier()
        t0 = time.perf_counter()
        step(it0 + b.warmup, b.steps)
        eng.synchronize()
        barrier()
        dt_own = time.perf_counter() - t0
        cstats = eng.comm_stats() if lib_row else None
        n_gathers = cstats["exchanges"] if lib_row else drv.n_exchanges
        mine.append(dict(name=name, config=spec["config"], steps=b.steps, warmup=b.warmup, seconds_timed=dt_own, particles=P,
                         block_sweeps_per_step=sweeps, all_gathers=n_gathers, groups=w["G"],
                         rccl_nranks=None if cstats is None else cstats["world"]))
        try:  # this rank's numbers on disk BEFORE the next collective (merge_rank_files reads `rows` too)
            rec = json.load(open(rank_file(rank)))
            rec["rows"] = mine
            write_rank_file(rank, rec)
        except (OSError, ValueError):
            pass
        dt = reduce([dt_own], "max")[0]
        dt_min = reduce([dt_own], "min")[0]
        gathers = [int(x) for x in reduce([n_gathers if r == rank else 0 for r in range(world)], "sum")]
        finite = True
        if rank == 0:
            import numpy as np
            finite = bool(np.isfinite(eng.get_state()[1]).all())
        kernels = eng.last_kernels()
        if lib_row:
            eng.comm_destroy()
        eng.close()
        value = P * world * sweeps * b.steps / dt
        rows.append(dict(name=name, workload=describe(b, w, world), value=value, unit="particle-updates/s", scaling="strong",
                         particle_parameter_updates_per_s=value * w["D"], ms_per_step=dt / b.steps * 1e3,
                         ms_per_step_min_over_ranks=dt_min / b.steps * 1e3, steps=b.steps, warmup=b.warmup, n_gpus=world,
                         groups_total=total, groups_per_rank=w["G"], particles_per_gpu=P, block_sweeps_per_step=sweeps,
                         all_gathers_per_rank=gathers, rccl_nranks=None if cstats is None else cstats["world"],
                         collective="library" if lib_row else "torch", collective_fallback=fallback, kernels=kernels,
                         finite_weights=finite, seconds=time.perf_counter() - t_wall))
    return rows


def compact_sharded_row(r):
    if "value" not in r:
        return {k: (short(v, 80) if isinstance(v, str) else v) for k, v in r.items() if k in ("name", "skipped", "error")}
    return sig({k: r[k] for k in ("name", "value", "ms_per_step", "steps", "n_gpus", "groups_per_rank", "all_gathers_per_rank", "rccl_nranks")})


# ----------------------------------------------------------------------------------------------------------------
# what the run prints: ONE compact last line (what a harness keeps: it holds on to the last few KB of stdout), with
# everything verbose on an EARLIER line tagged {"detail": ...} and in gpurun_out/bench_detail.json
# ----------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000   # bytes of the last stdout line (round 4's 23 KB line was cut off by the driver's 8 KB tail)
DETAIL_FILE = os.path.join("gpurun_out", "bench_detail.json")
ROW_KEYS = ("frac", "launch_ms", "bound", "traffic")  # of a row's roofline, next to name / value / ms_per_step / steps


def sig(x, n=5):
    """numbers to n significant digits (bools, ints, None and strings untouched); containers recursively"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return sig(float(x), n)


def short(s, n):
    return s if s is None or len(s) <= n else s[:n - 3] + "..."


def compact_row(r):
    if "value" not in r:
        return {k: (short(v, 80) if isinstance(v, str) else v) for k, v in r.items() if k in ("name", "skipped", "error")}
    rf = r.get("roofline") or {}
    c = dict(name=r["name"], value=r["value"], ms_per_step=r["ms_per_step"], steps=r["steps"])
    c.update({k: rf.get(k) for k in ROW_KEYS})
    if rf.get("counter_frac") is not None:  # (HBM-bound rows that quote the formula in `frac`: the counters' fraction beside it)
        c["counter_frac"] = rf["counter_frac"]
    if rf.get("shape_frac") is not None:  # (cfg4: against what a bare kernel of this traffic shape reaches)
        c["shape_frac"] = rf["shape_frac"]
    if (r.get("cpu_baseline") or {}).get("value") is not None:  # (rows with a CPU leg: the oracle on all cores / on one thread)
        c["cpu"], c["cpu_1thread"] = r["cpu_baseline"]["value"], r["cpu_baseline"].get("value_single_thread")
    if rf.get("shader_clock_mhz") is not None:  # (VALU-bound rows: the in-kernel clock the fraction was measured at)
        c["clock_mhz"] = rf["shader_clock_mhz"]
    if rf.get("frac_survey") is not None:  # (STREAMING rows: the executed MFMA flop and SURVEY 8d's unit side by side -- the latter > 1)
        c["frac_executed"], c["frac_survey"] = rf.get("frac_executed"), rf["frac_survey"]
    return sig(c)


def compact_line(out):
    """the last stdout line: the contract's fields + roofline + cpu_baseline + headline_context + numeric-only rows, no prose"""
    cfg, rf, cpu, ctx, acc = out["config"], out.get("roofline"), out.get("cpu_baseline"), out.get("headline_context"), out.get("accuracy") or {}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"] = dict(workload=short(cfg["workload"], 200), particles_per_gpu=cfg["particles_per_gpu"],
                          block_sweeps_per_step=cfg["block_sweeps_per_step"], parallelism=short(cfg["parallelism"], 60),
                          collective=short(cfg.get("collective"), 60), collective_fallback=short(cfg.get("collective_fallback"), 80),
                          rccl_nranks=cfg.get("rccl_nranks"), all_gathers_per_rank=cfg.get("all_gathers_per_rank"),
                          ms_per_step_min_over_ranks=cfg.get("ms_per_step_min_over_ranks"),
                          ms_per_step_max_over_ranks=cfg.get("ms_per_step_max_over_ranks"))
    line["particle_parameter_updates_per_s"] = out["particle_parameter_updates_per_s"]
    line["accuracy"] = dict(posterior_mean_l1_rel=acc.get("posterior_mean_l1_rel"),
                            max_abs_err_in_posterior_sd=acc.get("max_abs_err_in_posterior_sd"),
                            accept_rate=(acc.get("timed_chain") or {}).get("accept_rate"))
    line["roofline"] = None if rf is None else dict(
        {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "launch_ms", "launches", "traffic", "wasted_traffic_ratio",
                                "device_ms_per_iter", "counter_frac", "shader_clock_mhz", "frac_at_clock", "frac_of_mix_ceiling_at_clock")},
        kernel=short(rf.get("kernel"), 60), sclk_sysfs_mhz=(rf.get("sclk_sysfs_mhz") or {}).get("median"),
        # (what a wasted_traffic_ratio far from 1 is made of, where the detail line explains it: one clause here)
        **({"traffic_is": "the proposals' m rows, read once per observation chunk: the decomposition that fills the chip; VALU-bound"}
           if rf.get("traffic_note") and rf.get("wasted_traffic_ratio") else {}))
    line["cpu_baseline"] = None if cpu is None else dict(
        {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "value_single_thread", "cpu_model")}, sample=short(cpu.get("sample"), 160))
    line["headline_context"] = None if ctx is None else {k: ctx.get(k) for k in ("streaming_value", "streaming_frac_executed", "streaming_frac_survey",
                                                                                 "suffstat_value", "cpu_baseline_like_for_like_ratio")}
    rows = out.get("rows")
    line["rows"] = None if rows is None else [compact_sharded_row(r) if (r.get("scaling") == "strong" or "n_gpus" in r) else compact_row(r)
                                              for r in rows]
    line["detail"] = DETAIL_FILE
    line = sig(line)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) > LINE_LIMIT and line["rows"]:  # (cannot happen with ROWS as it is -- tests/test_host.py -- but never print a long line)
        line["rows"] = line["rows"][:-1]
        line["rows_truncated"] = True
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        raise RuntimeError(f"bench.py: the final line is {len(text)} bytes (> {LINE_LIMIT})")
    return text


def rank_file(rank):
    return os.path.join(ROOT, "gpurun_out", f"rank{rank}.json")


def write_rank_file(rank, rec):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        tmp = rank_file(rank) + ".tmp"
        with open(tmp, "w") as f:
            json.dump(rec, f)
        os.replace(tmp, rank_file(rank))
    except OSError as e:
        sys.stderr.write(f"bench.py rank {rank}: could not write {rank_file(rank)}: {e}\n")


def merge_rank_files(recs, why):
    """a line from the per-rank files alone (every rank timed its own K steps between the two barriers): what the parent
    prints when the ranks finished the timed region but the run did not get as far as rank 0's own line.  Marked partial:
    no roofline / cpu_baseline (those are rank 0's after the teardown)."""
    recs = sorted(recs, key=lambda r: r["rank"])
    world = recs[0]["world"]
    if [r["rank"] for r in recs] != list(range(world)):
        return None
    dt = max(r["seconds_timed"] for r in recs)
    steps = recs[0]["steps"]
    units = sum(r["particles"] * r["block_sweeps_per_step"] for r in recs) * steps
    out = {"metric": "particle-updates/sec (proposal+loglike+accept) at D=32, N=1e5" if recs[0]["config"] == "cfg3" else
                     f"particle-updates/sec (proposal+loglike+accept), {recs[0]['config']}",
           "value": units / dt, "unit": "particle-updates/s", "n_gpus": world, "steps": steps, "warmup": recs[0]["warmup"],
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic", "partial": short(why, 200),
           "config": {"workload": short(recs[0]["workload"], 200), "particles_per_gpu": recs[0]["particles"],
                      "collective": recs[0]["collective"], "collective_fallback": short(recs[0]["collective_fallback"], 80),
                      "rccl_nranks": recs[0]["rccl_nranks"], "all_gathers_per_rank": [r["all_gathers"] for r in recs],
                      "ms_per_step_min_over_ranks": min(r["ms_per_step"] for r in recs),
                      "ms_per_step_max_over_ranks": max(r["ms_per_step"] for r in recs)},
           "roofline": None, "cpu_baseline": None, "rows": None}
    names = [r["name"] for r in recs[0].get("rows") or []]
    merged = []
    for i, nm in enumerate(names):  # the sharded rows every rank got through (each rank's own K-step time, MAX over the ranks)
        per = [(r.get("rows") or [])[i] if len(r.get("rows") or []) > i and r["rows"][i]["name"] == nm else None for r in recs]
        if any(x is None for x in per):
            break
        dt_r = max(x["seconds_timed"] for x in per)
        merged.append(compact_sharded_row(dict(name=nm, value=sum(x["particles"] * x["block_sweeps_per_step"] for x in per) * per[0]["steps"] / dt_r,
                                               ms_per_step=dt_r / per[0]["steps"] * 1e3, steps=per[0]["steps"], n_gpus=world,
                                               groups_per_rank=per[0]["groups"], all_gathers_per_rank=[x["all_gathers"] for x in per],
                                               rccl_nranks=per[0]["rccl_nranks"])))
    out["rows"] = merged or None
    text = json.dumps(sig(out), separators=(",", ":"))
    assert len(text) <= LINE_LIMIT
    return text


def emit(out):
    """detail first (a tagged stdout line + a file under gpurun_out/), the compact line LAST"""
    detail = json.dumps({"detail": out})
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
            f.write(json.dumps(out, indent=1))
    except OSError:
        pass
    print(detail, flush=True)
    print(compact_line(out), flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    multi = world > 1 or os.environ.get("DEMC_FORCE_DIST") == "1"  # the latter: exercise the collective path with one rank
    wd = StageWatchdog(rank) if multi else None

    def stage(name, limit, code=EXIT_STUCK):
        if wd:
            wd.enter(name, limit, code)

    stage("import", 300.0)  # (the first `import torch` on a fresh box pages the image in: a minute or two)
    import numpy as np
    import torch
    import demc_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible and there is no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    fallback = os.environ.get("DEMC_BENCH_COLLECTIVE_FALLBACK")
    library = multi and a.collective == "library"
    store = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        stage("store", a.init_timeout, EXIT_INIT)
        if library:
            store = control_plane_store(rank, world, timeout_s=a.init_timeout)

    stage("workload", 300.0)
    w = build_workload(a)
    G, Np, D = w["G"], w["Np"], w["D"]
    P = G * Np
    n_rows = a.n_initial + a.warmup + a.steps * (2 if (two_pass(a) and world > 1) else 1)
    eng = make_engine(a, w, demc_amd, local, rank, world)
    if library:
        # the whole sharded iteration behind the C-ABI: demc_step on a handle that owns its RCCL communicator
        stage("comm_init", a.init_timeout, EXIT_INIT)
        err = None
        try:
            eng.comm_init(exchange_comm_id(store, rank, eng.comm_unique_id), rank, world)
        except Exception as e:  # a clean failure (DEMC_ERCCL ...): agree with the others on what to do
            err = f"{type(e).__name__}: {e}"
        bad = agree_on_init(store, rank, world, err)
        if bad is not None:
            # every rank leaves the library road together and takes torch.distributed's (same RCCL, torch's bootstrap)
            if err is None:
                eng.comm_destroy()
            library, fallback = False, f"library collective: {bad}"
            sys.stderr.write(f"bench.py rank {rank}: demc_comm_init failed somewhere ({bad}); falling back to --collective torch\n")
    if multi and not library:
        stage("process_group", a.init_timeout, EXIT_INIT)
        import torch.distributed as dist_
        dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        dist = dist_
    if library:
        eng.comm_set_overlap(a.async_migration)
        step = eng.step_enqueue
    else:
        from demc_amd.distributed import ShardedDriver
        drv = ShardedDriver(eng, dist, torch.device("cuda", local), stream_ordered=True, async_migration=a.async_migration)
        step = drv.step

    def sync():  # barrier + device drained, on both sides of the timed region
        if library:
            eng.synchronize()
            eng.comm_allreduce([])  # an all-reduce of one word over the engine's communicator = the barrier
        elif dist:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(vals, op):
        if library:
            return [float(x) for x in eng.comm_allreduce(list(vals), op)]
        if dist:
            tt = torch.tensor(list(vals), dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
            return [float(x) for x in tt.tolist()]
        return [float(x) for x in vals]

    it0 = 1 + a.n_initial
    stage("warmup", max(120.0, a.deadline / 3))
    step(it0, a.warmup)
    sync()
    split = two_pass(a) and not (multi and world == 1)  # (DEMC_FORCE_DIST with one rank: events inside the timed region)
    a.events_inline = not split
    if not a.no_roofline and not split:
        eng.timing_enable(True)  # HIP events on the handle's stream around every launch of the timed iterations
    stage("timed", max(120.0, a.deadline / 3))
    sclk = SclkSampler(local)
    sclk.start()
    t0 = time.perf_counter()
    step(it0 + a.warmup, a.steps)
    sync()
    dt_own = time.perf_counter() - t0
    a.sclk_sysfs = sclk.result()
    a.clock = None
    tm = None
    if split and not multi:  # (two_pass: the same K steps of the same chain on a fresh engine, with the events on)
        eng_p = make_engine(a, w, demc_amd, local, rank, world)
        eng_p.step_enqueue(it0, a.warmup)
        eng_p.synchronize()
        eng_p.timing_enable(True)
        eng_p.step_enqueue(it0 + a.warmup, a.steps)
        eng_p.synchronize()
        torch.cuda.synchronize()
        tm = eng_p.timing_read()
        a.clock = eng_p.timing_clock()
        eng_p.close()
    elif split:  # (several ranks: the K steps again on the same engines; not part of the timed region)
        stage("profiled", max(120.0, a.deadline / 3))
        eng.timing_enable(True)
        step(it0 + a.warmup + a.steps, a.steps)
        sync()
        tm = eng.timing_read()
        a.clock = eng.timing_clock()
        eng.timing_enable(False)
    elif not a.no_roofline:
        tm = eng.timing_read()
        a.clock = eng.timing_clock()
        eng.timing_enable(False)
    cstats = eng.comm_stats() if library else None
    n_gathers = cstats["exchanges"] if library else (drv.n_exchanges if multi else 0)
    if multi:  # every rank's own numbers on disk BEFORE the next collective: a late hang still leaves them (merge_rank_files)
        write_rank_file(rank, dict(rank=rank, world=world, steps=a.steps, warmup=a.warmup, seconds_timed=dt_own,
                                   ms_per_step=dt_own / a.steps * 1e3, particles=P,
                                   block_sweeps_per_step=1 if w["masks"] is None else len(w["masks"]),
                                   all_gathers=n_gathers, rccl_nranks=None if cstats is None else cstats["world"],
                                   collective="library" if library else "torch", collective_fallback=fallback,
                                   workload=describe(a, w, world), config=a.config))
    stage("reduce", 120.0)
    dt = reduce([dt_own], "max")[0]  # MAX over ranks
    dt_min = reduce([dt_own], "min")[0]
    per_rank_gathers = [int(x) for x in reduce([n_gathers if r == rank else 0 for r in range(world)], "sum")]
    sweeps = 1 if w["masks"] is None else len(w["masks"])  # block_update!: every block sweep updates every particle once
    value = P * world * sweeps * a.steps / dt

    out = None
    if rank == 0:
        k_last = min(10, a.steps)
        th_h, acc_h, _, _ = eng.get_history(n_rows - k_last, n_rows)
        th_now, w_now, ids = eng.get_state()
        timed_chain = dict(accept_rate=float(acc_h.mean()), rows_used=k_last, finite_weights=bool(np.isfinite(w_now).all()),
                           note="the timed iterations lie inside the reference's burn-in (burnin = 1000) unless --burnin says otherwise; "
                                "acceptance and spread say what the sampler did, they are not a throughput claim (ESS/s is not particle-updates/s)")
        if "posterior_sd" in w:
            timed_chain["ensemble_sd_over_posterior_sd"] = float(np.median(th_h.reshape(-1, D).std(0) / w["posterior_sd"]))
        if "truth" in w and "posterior_mean" not in w and D <= 64:  # (not 10^4 numbers of a hierarchical row)
            timed_chain["ensemble_mean"] = th_now.mean(0).tolist()
            timed_chain["generating_parameters"] = np.asarray(w["truth"]).tolist()
    roofline = None
    if tm is not None:
        roofline = roofline_of(a, w, tm, a.steps, P, dt / a.steps)
    kernels_ran = eng.last_kernels() if rank == 0 else None  # (demc_last_kernels: the instances the last step launched)
    sharded = None
    if multi and a.rows != "none" and is_plain_headline(a):  # BASELINE's 8-GPU configs, partitioned over the ranks of this run
        sharded = run_sharded_rows(a, demc_amd, local, rank, world, store, dist, sync, reduce, stage, fallback)
    stage("teardown", 120.0)
    if library:
        eng.comm_allreduce([])  # nobody leaves (and rank 0 keeps the store up) before everybody has finished
        eng.comm_destroy()
    eng.close()
    if wd:
        wd.stop()  # what follows is rank 0's own CPU work (accuracy leg); the other ranks wait in the last barrier at most

    if rank == 0:
        acc = accuracy_leg(a, w, demc_amd, local, np.random.default_rng(20260003))
        accuracy = dict(timed_chain=timed_chain)
        if acc is not None:
            accuracy.update(acc)
        rows = sharded
        plain_headline = is_plain_headline(a)
        if world == 1 and not multi and a.rows != "none" and (a.rows is not None or plain_headline) and not a.no_roofline:
            rows = run_rows(a, w, demc_amd, local)
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a, w)
        context = None
        if rows:
            byname = {r["name"]: r for r in rows if "value" in r}
            st, su = byname.get("cfg3_streaming"), byname.get("cfg3_suffstat")
            if a.config == "cfg3" and (st is not None or su is not None):
                context = dict(streaming_value=None if st is None else st["value"],
                               streaming_frac_executed=None if st is None else st["roofline"].get("frac_executed"),
                               streaming_frac_survey=None if st is None else st["roofline"].get("frac_survey"),
                               suffstat_value=None if su is None else su["value"],
                               cpu_baseline_like_for_like_ratio=None if cpu is None else value / cpu["value"],
                               note="the headline is DIRECT: the residual form term by term, SURVEY 8d's 3ND per update -- and what cpu_baseline "
                                    "(the oracle's whitened residual form on the host) does, so value / cpu_baseline is like for like.  "
                                    "STREAMING (the headline of rounds 1-5) = SUFFSTAT + a [proposals x D].[D x N] matrix-core product that is "
                                    "analytically zero after centring: same accept decisions, frac_survey above 1 says it is not the survey's work")
        out = {
            "metric": "particle-updates/sec (proposal+loglike+accept) at D=32, N=1e5" if a.config == "cfg3" else
                      f"particle-updates/sec (proposal+loglike+accept), {a.config}",
            "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": describe(a, w, world), "particles_per_gpu": P, "block_sweeps_per_step": sweeps,
                       "parallelism": f"groups sharded x{world}, one all-gather per migration" +
                                      (" (asynchronous: unselected groups update during the gather)" if a.async_migration else ""),
                       "collective": None if not multi else
                                     ("ncclAllGather on the engine's own communicator (demc_comm_init, behind the C-ABI)" if library
                                      else "torch.distributed.all_gather_into_tensor (backend nccl = RCCL)"),
                       "collective_fallback": fallback,
                       "rccl_nranks": None if cstats is None else cstats["world"],
                       "all_gathers_rank0": n_gathers, "all_gathers_per_rank": per_rank_gathers,
                       "ms_per_step_min_over_ranks": dt_min / a.steps * 1e3, "ms_per_step_max_over_ranks": dt / a.steps * 1e3},
            "particle_parameter_updates_per_s": value * D,
            "particle_iterations_per_s": P * world * a.steps / dt,  # (value counts every block sweep as an update: cfg4 has two)
            "kernels": kernels_ran,
            "accuracy": accuracy, "roofline": roofline, "headline_context": context, "cpu_baseline": cpu, "rows": rows,
        }
        emit(out)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
