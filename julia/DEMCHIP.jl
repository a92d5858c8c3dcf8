# DEMCHIP.jl -- the binding a maintainer of DifferentialEvolutionMCMC.jl would add to route the per-iteration
# hot path (step!/pstep!, src/main.jl:84-107) through libdemc_hip.so.  Thin, mechanical @ccall wrappers over
# include/demc.h plus one new `sample` method selected by dispatch on a tag type, exactly like the existing
# MCMCThreads method (src/main.jl:62-71).
#
# STATUS: written against include/demc.h but NOT executed -- there is no Julia runtime in the build container
# or on the GPU box (SURVEY.md 8c).  The tested host is the Python mirror in differentialevolutionmcmc.jl_amd/.
module DEMCHIP

using DifferentialEvolutionMCMC
import DifferentialEvolutionMCMC: DE, DEModel, sample_init, bundle_samples, random_gamma, fixed_gamma,
    variable_gamma, resample, mh_update!, maximize!, minimize!, compute_posterior!, evaluate_fun!
import DifferentialEvolutionMCMC: sample

const LIB = get(ENV, "DEMC_HIP_LIB", "libdemc_hip.so")

# POD mirror of demc_config (include/demc.h); field order and types must not change
struct DemcConfig
    n_groups::Int32; Np::Int32; D::Int32; n_blocks::Int32
    burnin::Int64; n_initial::Int64; n_rows::Int64
    alpha::Float64; beta::Float64; eps::Float64; sigma::Float64; kappa::Float64; theta_snooker::Float64
    proposal_kind::Int32; partner_kind::Int32; update_kind::Int32; fitness_kind::Int32
    schedule::Int32; store_history::Int32; group_offset::Int32; n_groups_total::Int32
    seed::UInt64; device_id::Int32; loglike_mode::Int32; trace::Int32; fuse::Int32
    geometry_groups::Int32; reserved0::Int32
end

# POD mirror of demc_replay (include/demc.h): caller-supplied draws (test mode).  C_NULL members = draw as usual.
struct DemcReplay
    u_step::Ptr{Float64}; u_group::Ptr{Float64}; u_part::Ptr{Float64}; partner::Ptr{Int64}
    u_noise::Ptr{Float64}; z_noise::Ptr{Float64}; u_recomb::Ptr{Float64}
    mig_groups::Ptr{Int32}; n_mig_groups::Int32; reserved::Int32; mig_particle::Ptr{Int64}
end

"""
Registered model family: what replaces the closures of `DEModel` on the device (closures cannot run in a kernel).
`family` is a DEMC_FAM_* id, `data`/`dims`/`hyper` as documented in include/demc.h, and one prior entry
`(kind, a, b, ref)` per scalar parameter.
"""
struct ModelSpec
    family::Int32
    data::Vector{Float64}
    dims::Vector{Int64}
    hyper::Vector{Float64}
    prior_kind::Vector{Int32}
    prior_a::Vector{Float64}
    prior_b::Vector{Float64}
    prior_ref::Vector{Int32}
end

struct HIPBackend
    schedule::Symbol      # :two_colour (default), :synchronous, or :sequential (the reference's own sweep; slow, for replay runs)
    loglike_mode::Symbol  # :streaming, :suffstat or :direct (include/demc.h DEMC_LOGLIKE_*)
    device_id::Int
    seed::UInt64
end
HIPBackend(; schedule = :two_colour, loglike_mode = :streaming, device_id = 0, seed = rand(UInt64)) =
    HIPBackend(schedule, loglike_mode, device_id, seed)

function check(h, rc)
    rc == 0 && return nothing
    msg = unsafe_string(@ccall LIB.demc_last_error(h::Ptr{Cvoid})::Cstring)
    error("libdemc_hip: status $rc: $msg")   # non-zero status -> Julia exception (SURVEY 8b)
end

"a `note:` line left by a SUCCESSFUL call -- a documented deviation in force on the handle (include/demc.h, demc_last_error) -- is a warning"
function note(h)
    msg = unsafe_string(@ccall LIB.demc_last_error(h::Ptr{Cvoid})::Cstring)
    startswith(msg, "note:") && @warn "libdemc_hip: " * msg[7:end]
    nothing
end

"shader clock (MHz: median, min, max over the CUs; CUs that reported) the DIRECT likelihood kernel held in its last timed launch"
function timing_clock(h)
    out = zeros(Float64, 4)
    check(h, @ccall LIB.demc_timing_clock(h::Ptr{Cvoid}, out::Ptr{Float64})::Int32)
    out
end

# function-valued hooks of DE (src/structs.jl:71-74) -> enums; anything else has no device implementation
hook_code(f, table, what) = haskey(table, f) ? table[f] :
    error("DE.$what = $f cannot run on the device; use the CPU path (step!/pstep!)")
const PROPOSALS = IdDict(random_gamma => 0, fixed_gamma => 1, variable_gamma => 2)
const PARTNERS = IdDict(DifferentialEvolutionMCMC.sample => 0, resample => 1)
const UPDATES = IdDict(mh_update! => 0, maximize! => 1, minimize! => 2)
const FITNESS = IdDict(compute_posterior! => 0, evaluate_fun! => 1)

flatten(Θ) = collect(Float64, Iterators.flatten(Θ))

"top-level layout of a (possibly nested) Θ: for every parameter its shape (() for a scalar) and its offset in the flat row"
function layout(Θ)
    shapes = [θ isa AbstractArray ? size(θ) : () for θ in Θ]
    lens = [θ isa AbstractArray ? length(θ) : 1 for θ in Θ]
    offs = cumsum(vcat(0, lens))
    return shapes, lens, offs
end

"one top-level element of Θ back from its slice of a flat row (utilities.jl:161-180 stores top-level elements)"
unflatten(x::AbstractVector{Float64}, shape) = shape == () ? x[1] : reshape(collect(x), shape)

"de.blocking_on(de) is asked on EVERY iteration with de.iter set (src/main.jl:34,137,162): (first iteration, count, on?) runs"
function blocking_runs(de, n_iter)
    saved = de.iter
    flags = Bool[]
    for it = 1:n_iter
        de.iter = it + de.n_initial
        push!(flags, de.blocking_on(de))
    end
    de.iter = saved
    runs = Tuple{Int,Int,Bool}[]
    for (i, f) in enumerate(flags)
        if !isempty(runs) && runs[end][3] == f
            runs[end] = (runs[end][1], runs[end][2] + 1, f)
        else
            push!(runs, (i, 1, f))
        end
    end
    return runs
end

"flatten de.bounds (one tuple per top-level parameter, zip-truncated: utilities.jl:73-78) to per-scalar lo/hi"
function flat_bounds(de, Θ)
    lo = Float64[]; hi = Float64[]
    for (i, θ) in enumerate(Θ)
        b = i <= length(de.bounds) ? de.bounds[i] : (-Inf, Inf)
        append!(lo, fill(Float64(b[1]), length(θ))); append!(hi, fill(Float64(b[2]), length(θ)))
    end
    return lo, hi
end

"DE keyword constructor (src/structs.jl:80-131) -> demc_config for a handle that owns `n_groups` groups starting at `group_offset`"
function make_config(de::DE, D::Int, n_iter::Int, b; n_groups = de.n_groups, group_offset = 0, device_id = b.device_id)
    sched = b.schedule == :two_colour ? 2 : b.schedule == :synchronous ? 1 : 0
    return DemcConfig(n_groups, de.Np, D, 0, de.burnin, de.n_initial, n_iter + de.n_initial,
        de.α, de.β, de.ϵ, de.σ, de.κ, de.θsnooker,
        hook_code(de.generate_proposal, PROPOSALS, "generate_proposal"), hook_code(de.sample, PARTNERS, "sample"),
        hook_code(de.update_particle!, UPDATES, "update_particle!"), hook_code(de.evaluate_fitness!, FITNESS, "evaluate_fitness!"),
        sched, 1, group_offset, de.n_groups, b.seed, device_id, b.loglike_mode == :streaming ? 0 : b.loglike_mode == :suffstat ? 1 : 2, 0, 0, de.n_groups, 0)
end

"model, priors, bounds, the prior-draw history rows and the particles `ps` (one shard's, in slot order) onto handle `h`"
function load_handle!(h, m::ModelSpec, de::DE, ps)
    GC.@preserve m begin
        check(h, @ccall LIB.demc_set_model(h::Ptr{Cvoid}, m.family::Int32, m.data::Ptr{Float64}, m.dims::Ptr{Int64},
            Int32(length(m.dims))::Int32, m.hyper::Ptr{Float64}, Int32(length(m.hyper))::Int32)::Int32)
        check(h, @ccall LIB.demc_set_priors(h::Ptr{Cvoid}, m.prior_kind::Ptr{Int32}, m.prior_a::Ptr{Float64},
            m.prior_b::Ptr{Float64}, m.prior_ref::Ptr{Int32})::Int32)
    end
    lo, hi = flat_bounds(de, ps[1].Θ)
    check(h, @ccall LIB.demc_set_bounds(h::Ptr{Cvoid}, lo::Ptr{Float64}, hi::Ptr{Float64})::Int32)
    if de.n_initial > 0   # initialize_samples (utilities.jl:35-39): rows 1:n_initial, [row][particle][D]
        rows = Float64[x for i = 1:de.n_initial for p in ps for x in Iterators.flatten(de.samples[i, :, p.id])]
        check(h, @ccall LIB.demc_set_history_rows(h::Ptr{Cvoid}, 0::Int64, Int64(de.n_initial)::Int64, rows::Ptr{Float64})::Int32)
    end
    theta = Float64[x for p in ps for x in flatten(p.Θ)]              # D x P column-major == [P][D] row-major
    weight = Float64[p.weight for p in ps]                            # evaluated by sample_init on the host
    ids = Int64[p.id - 1 for p in ps]                                 # ids are 0-based across the ABI
    check(h, @ccall LIB.demc_set_state(h::Ptr{Cvoid}, theta::Ptr{Float64}, weight::Ptr{Float64}, ids::Ptr{Int64})::Int32)
    return nothing
end

"""
    run_segments(step, de, n_iter, handles)

`for iter in 1:n_iter: de.iter = iter + n_initial; groups = stepfun(model, de, groups)` (src/main.jl:33-38) as runs of
iterations.  Whether a step is a block update is asked per iteration (`de.blocking_on(de)`, main.jl:137,162): consecutive
iterations with the same answer go to the device as ONE call `step(first, count)`, with the block masks of `de.blocks`
(one nested Bool array per block, src/structs.jl:45, flattened like Θ -> nblocks x D bytes, row-major) switched on or
off on EVERY handle in between -- a single handle, or all shards of a multi-GPU set.
"""
function run_segments(step, de::DE, n_iter::Int, handles)
    has_blocks = !isempty(de.blocks) && !(de.blocks[1] isa Bool)               # default is the placeholder [false]
    masks = has_blocks ? UInt8[x for blk in de.blocks for x in Iterators.flatten(blk)] : UInt8[]
    for (first, count, on) in blocking_runs(de, n_iter)
        on && !has_blocks && error("blocking_on(de) is true but de.blocks holds no blocks")
        nb = on ? length(de.blocks) : 0
        for h in handles
            check(h, @ccall LIB.demc_set_blocks(h::Ptr{Cvoid}, masks::Ptr{UInt8}, Int32(nb)::Int32)::Int32)
        end
        step(first, count)
    end
    return nothing
end

"""
    sample(model::DEModel, de::DE, backend::HIPBackend, n_iter; model_spec, progress=false)

Same contract as `sample(model, de, MCMCThreads(), n_iter)` (src/main.jl:62-71): `sample_init` and
`bundle_samples` run unchanged on the host; the iteration loop (src/main.jl:33-38) becomes `demc_step`.
"""
function sample(model::DEModel, de::DE, b::HIPBackend, n_iter::Int; model_spec::ModelSpec, progress = false, kwargs...)
    groups = sample_init(model, de, n_iter)                      # src/main.jl:263-271 (allocates de.samples)
    particles = vcat(groups...)
    P = length(particles); D = length(flatten(particles[1].Θ))
    shapes, lens, offs = layout(particles[1].Θ)
    cfg = make_config(de, D, n_iter, b)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = @ccall LIB.demc_create(Ref(cfg)::Ptr{DemcConfig}, href::Ptr{Ptr{Cvoid}})::Int32
    h = href[]
    try
        check(h, rc)
        note(h)
        load_handle!(h, model_spec, de, particles)
        run_segments(de, n_iter, [h]) do first, count
            check(h, @ccall LIB.demc_step(h::Ptr{Cvoid}, Int64(first + de.n_initial)::Int64, Int32(count)::Int32)::Int32)
        end
        de.iter = n_iter + de.n_initial
        n_rows = n_iter + de.n_initial
        # bundle_samples' gather (src/main.jl:232-241) on the device: layout 0 IS Array{Float64,3}(n_rows, D+2, P) in
        # Julia's column-major order, already keyed by particle id (parameters, then "acceptance", then "lp")
        v = Array{Float64,3}(undef, n_rows, D + 2, P)
        check(h, @ccall LIB.demc_export_chains(h::Ptr{Cvoid}, 0::Int64, Int64(n_rows)::Int64, 0::Int32, v::Ptr{Float64})::Int32)
        # de.samples[row, k, id] holds the k-th TOP-LEVEL element of Θ (utilities.jl:161-180): a scalar, or the array rebuilt
        # from its slice of the flat row (test/multivariate_normal_tests.jl:19 uses Θ = [μ::Vector, σ])
        for id = 1:P
            for k in eachindex(shapes), row = 1:n_rows
                de.samples[row, k, id] = unflatten(view(v, row, offs[k]+1:offs[k]+lens[k], id), shapes[k])
            end
            particles[id].accept .= v[:, D + 1, id] .!= 0
            particles[id].lp .= v[:, D + 2, id]
        end
        pull_state!(h, particles, P, D, shapes, lens, offs)
    finally
        h != C_NULL && @ccall LIB.demc_destroy(h::Ptr{Cvoid})::Int32
    end
    # particles are now ordered by id, so Θ columns and accept/lp agree (fixes the pairing quirk of main.jl:232-239)
    return bundle_samples(model, de, [particles], n_iter)                   # src/main.jl:222-250, unchanged
end

"final state of the particle objects behind handle `h` (bundle_samples reads accept/lp from them; Θ for completeness)"
function pull_state!(h, particles, Pl, D, shapes, lens, offs)
    th = Vector{Float64}(undef, Pl * D); wt = Vector{Float64}(undef, Pl); idv = Vector{Int64}(undef, Pl)
    check(h, @ccall LIB.demc_get_state(h::Ptr{Cvoid}, th::Ptr{Float64}, wt::Ptr{Float64}, idv::Ptr{Int64})::Int32)
    for s = 1:Pl
        p = particles[idv[s] + 1]
        p.Θ = [unflatten(view(th, (s - 1) * D + offs[k] + 1:(s - 1) * D + offs[k] + lens[k]), shapes[k]) for k in eachindex(shapes)]
        p.weight = wt[s]
    end
    return nothing
end

# ----------------------------------------------------------------------------------------------------------------------
# Multi-GPU (SURVEY 8e).  Groups are sharded over the GPUs of the node; groups never interact inside update!/p_update!
# (src/main.jl:135-167), and migration! (src/migration.jl:11-19, called from src/main.jl:85,103) becomes ONE RCCL all-gather
# per migration event -- which lives BEHIND the C-ABI, so this host needs no collective library of its own.
# ----------------------------------------------------------------------------------------------------------------------
"One Julia task drives every GPU of the node (demc_create_multi): `devices` are HIP device ids, de.n_groups must divide by their number"
struct HIPMultiBackend
    devices::Vector{Int32}
    schedule::Symbol
    loglike_mode::Symbol
    seed::UInt64
    device_id::Int           # (unused: every shard takes its own device)
end
HIPMultiBackend(devices; schedule = :two_colour, loglike_mode = :streaming, seed = rand(UInt64)) =
    HIPMultiBackend(Int32.(collect(devices)), schedule, loglike_mode, seed, 0)

function mcheck(m, rc)
    rc == 0 && return nothing
    msg = unsafe_string(@ccall LIB.demc_multi_last_error(m::Ptr{Cvoid})::Cstring)
    error("libdemc_hip (multi): status $rc: $msg")
end

"""
    sample(model::DEModel, de::DE, backend::HIPMultiBackend, n_iter; model_spec)

The threaded method's contract (src/main.jl:62-71) on several GPUs from ONE process: shard r owns groups
`r*G+1:(r+1)*G` with their particles and history; `demc_multi_step` enqueues every shard's iterations and the one
all-gather per migration (grouped ncclAllGather over xGMI) before it waits for any of them.
"""
function sample(model::DEModel, de::DE, b::HIPMultiBackend, n_iter::Int; model_spec::ModelSpec, progress = false, kwargs...)
    groups = sample_init(model, de, n_iter)
    particles = vcat(groups...)
    P = length(particles); D = length(flatten(particles[1].Θ))
    shapes, lens, offs = layout(particles[1].Θ)
    R = length(b.devices)
    de.n_groups % R == 0 || error("n_groups = $(de.n_groups) must divide by the number of devices ($R)")
    Pl = P ÷ R
    cfg = make_config(de, D, n_iter, b)          # the configuration of the WHOLE population
    mref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = @ccall LIB.demc_create_multi(Ref(cfg)::Ptr{DemcConfig}, Int32(R)::Int32, b.devices::Ptr{Int32}, mref::Ptr{Ptr{Cvoid}})::Int32
    m = mref[]
    try
        mcheck(m, rc)
        hs = [@ccall LIB.demc_multi_shard(m::Ptr{Cvoid}, Int32(r - 1)::Int32)::Ptr{Cvoid} for r = 1:R]
        for r = 1:R
            load_handle!(hs[r], model_spec, de, particles[(r - 1) * Pl + 1:r * Pl])   # ids 1..P in group-major order (main.jl:265-268)
        end
        # block updates and their per-iteration switch exactly as on one GPU: masks on every shard, one multi_step per run
        run_segments(de, n_iter, hs) do first, count
            mcheck(m, @ccall LIB.demc_multi_step(m::Ptr{Cvoid}, Int64(first + de.n_initial)::Int64, Int32(count)::Int32)::Int32)
        end
        de.iter = n_iter + de.n_initial
        n_rows = n_iter + de.n_initial
        for r = 1:R
            # raw history of the shard, keyed by slot, + the (global, 0-based) id that sat in the slot: [row][slot][D] row-major
            th = Array{Float64,3}(undef, D, Pl, n_rows); acc = Array{UInt8,2}(undef, Pl, n_rows)
            lp = Array{Float64,2}(undef, Pl, n_rows); idh = Array{Int64,2}(undef, Pl, n_rows)
            check(hs[r], @ccall LIB.demc_get_history(hs[r]::Ptr{Cvoid}, 0::Int64, Int64(n_rows)::Int64, th::Ptr{Float64},
                acc::Ptr{UInt8}, lp::Ptr{Float64}, idh::Ptr{Int64})::Int32)
            for row = 1:n_rows, s = 1:Pl
                id = idh[s, row] + 1                                       # samples[iter, :, p.id] (utilities.jl:170-180)
                for k in eachindex(shapes)
                    de.samples[row, k, id] = unflatten(view(th, offs[k]+1:offs[k]+lens[k], s, row), shapes[k])
                end
                particles[id].accept[row] = acc[s, row] != 0
                particles[id].lp[row] = lp[s, row]
            end
            pull_state!(hs[r], particles, Pl, D, shapes, lens, offs)
        end
    finally
        m != C_NULL && @ccall LIB.demc_destroy_multi(m::Ptr{Cvoid})::Int32
    end
    return bundle_samples(model, de, [particles], n_iter)
end

# One PROCESS per GPU (Distributed.jl workers, MPI.jl ranks): every worker creates its shard with
# make_config(de, D, n_iter, b; n_groups = G, group_offset = rank*G, device_id = local_gpu), joins the communicator and calls
# demc_step -- the exchange happens inside.  The host only carries the 128-byte id from rank 0 to the others.
const COMM_ID_BYTES = 128
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    rc = @ccall LIB.demc_comm_unique_id(id::Ptr{Cvoid}, Int32(COMM_ID_BYTES)::Int32)::Int32
    rc == 0 || error("libdemc_hip: demc_comm_unique_id: status $rc")
    return id
end
"collective: every rank calls it with the same id; `overlap`: unselected groups update while the all-gather is in flight"
function comm_init!(h, id::Vector{UInt8}, rank::Int, world::Int; overlap = false)
    length(id) == COMM_ID_BYTES || error("communicator id must be $COMM_ID_BYTES bytes")
    check(h, @ccall LIB.demc_comm_init(h::Ptr{Cvoid}, id::Ptr{Cvoid}, Int32(rank)::Int32, Int32(world)::Int32)::Int32)
    check(h, @ccall LIB.demc_comm_set_overlap(h::Ptr{Cvoid}, Int32(overlap)::Int32)::Int32)
    return nothing
end
comm_destroy!(h) = check(h, @ccall LIB.demc_comm_destroy(h::Ptr{Cvoid})::Int32)
"reduce host doubles over the ranks in place (op: 0 sum, 1 max, 2 min); an empty vector is a barrier"
function comm_allreduce!(h, x::Vector{Float64}, op::Int = 0)
    check(h, @ccall LIB.demc_comm_allreduce(h::Ptr{Cvoid}, x::Ptr{Float64}, Int32(length(x))::Int32, Int32(op)::Int32)::Int32)
    return x
end
"(world, rank, all-gathers issued so far)"
function comm_stats(h)
    out = Vector{Int64}(undef, 3)
    check(h, @ccall LIB.demc_comm_stats(h::Ptr{Cvoid}, out::Ptr{Int64})::Int32)
    return (world = out[1], rank = out[2], exchanges = out[3])
end
"migration! of iteration `iter` alone (pack -> all-gather -> apply) for a host loop that calls demc_update itself"
migration_exchange!(h, iter::Int) = check(h, @ccall LIB.demc_migration_exchange(h::Ptr{Cvoid}, Int64(iter)::Int64)::Int32)
"enqueue only / wait: for a task that drives several handles (one per device) without demc_create_multi"
step_async!(h, iter0::Int, n::Int) = check(h, @ccall LIB.demc_step_async(h::Ptr{Cvoid}, Int64(iter0)::Int64, Int32(n)::Int32)::Int32)
synchronize!(h) = check(h, @ccall LIB.demc_synchronize(h::Ptr{Cvoid})::Int32)

"""
    host_migration!(h, de, P)

`migration!` (src/migration.jl:11-19) with the reference's OWN random choices kept on the host: the weights come back
through `demc_get_weights`, `select_groups` / `select_particles` run unchanged on them (they only look at
`Particle.weight`), and only `shift_particles!` (:84-91) happens on the device through `demc_apply_migration`.
For callers that want the package's task-local RNG stream to decide who migrates (e.g. to compare runs with the CPU
path); the default `sample` method above leaves the whole exchange on the device (`demc_step`).
"""
function host_migration!(h, de::DE, P::Int)  # P = de.n_groups * de.Np
    w = Vector{Float64}(undef, P)
    check(h, @ccall LIB.demc_get_weights(h::Ptr{Cvoid}, w::Ptr{Float64})::Int32)
    # stand-in particles carrying only what selection reads; groups are contiguous blocks of Np slots
    groups = [[DifferentialEvolutionMCMC.Particle(Θ = [0.0], weight = w[(g - 1) * de.Np + p]) for p = 1:de.Np] for g = 1:de.n_groups]
    sub_group = DifferentialEvolutionMCMC.select_groups(de, groups)                       # migration.jl:31-35
    p_idx, _ = DifferentialEvolutionMCMC.select_particles(sub_group)                       # migration.jl:46-54
    g_idx = [findfirst(x -> x === sg, groups) for sg in sub_group]
    slots = Int32[(g_idx[i] - 1) * de.Np + (p_idx[i] - 1) for i in eachindex(g_idx)]      # 0-based local slots
    src = circshift(slots, 1)                                                              # shift_particles! :84-91
    check(h, @ccall LIB.demc_apply_migration(h::Ptr{Cvoid}, src::Ptr{Int32}, slots::Ptr{Int32}, Int32(length(slots))::Int32)::Int32)
    return nothing
end
# driver loop for that mode: `rand() <= de.α && host_migration!(h, de, P); demc_update(h, iter, 1)` per iteration

"""
    set_replay!(h; u_step, u_group, u_part, partner, u_noise, z_noise, u_recomb, mig_groups, mig_particle)

Test mode (`demc_set_replay`, include/demc.h): feed draws taken from Julia's own RNG -- in the order of SURVEY.md Appendix A --
in place of the library's addressed Philox draws, for one step at a time, e.g. with `HIPBackend(schedule = :sequential)` to
follow `crossover!`'s in-place sweep exactly.  Arrays use the C layouts of demc.h (`u_part`: 5 x P column-major = [P][5]
row-major, `partner`: 3 x P, the per-scalar tables D x P); partner rows are 0-based positions inside the group.  Any argument
left `nothing` is drawn as usual; `set_replay!(h)` with no arguments switches the mode off.
"""
function set_replay!(h; u_step = nothing, u_group = nothing, u_part = nothing, partner = nothing, u_noise = nothing,
                     z_noise = nothing, u_recomb = nothing, mig_groups = nothing, mig_particle = nothing)
    args = (u_step, u_group, u_part, partner, u_noise, z_noise, u_recomb, mig_groups, mig_particle)
    if all(isnothing, args)
        check(h, @ccall LIB.demc_set_replay(h::Ptr{Cvoid}, C_NULL::Ptr{DemcReplay})::Int32)
        return nothing
    end
    ptr(x, T) = x === nothing ? Ptr{T}(C_NULL) : pointer(x)
    GC.@preserve u_step u_group u_part partner u_noise z_noise u_recomb mig_groups mig_particle begin
        r = DemcReplay(ptr(u_step, Float64), ptr(u_group, Float64), ptr(u_part, Float64), ptr(partner, Int64),
                       ptr(u_noise, Float64), ptr(z_noise, Float64), ptr(u_recomb, Float64), ptr(mig_groups, Int32),
                       mig_groups === nothing ? Int32(0) : Int32(length(mig_groups)), Int32(0), ptr(mig_particle, Int64))
        check(h, @ccall LIB.demc_set_replay(h::Ptr{Cvoid}, Ref(r)::Ptr{DemcReplay})::Int32)
    end
    return nothing
end

end # module
