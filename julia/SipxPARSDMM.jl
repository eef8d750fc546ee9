# SipxPARSDMM.jl -- PARSDMM on one MI355X through libsipx.so (include/sipx.h), with the EXACT positional signature of the
# reference's entry point (src/PARSDMM.jl:25-35):
#
#     PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options[, x, l, y]) -> (x, log_PARSDMM, l, y)
#
# so that `(x, log) = PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options)` in a caller's script
# (examples/projection_intersection_2D.jl:81) needs no edit: `include("SipxPARSDMM.jl"); using .SipxPARSDMM: PARSDMM` in place of
# `using SetIntersectionProjection: PARSDMM`.
#
# NOT EXECUTED anywhere in this repository's pipeline: neither the build container nor the GPU box has a Julia toolchain
# (SURVEY 8c).  The C side of every call below IS exercised -- tests/c_abi/phases.c drives the same sequence of entry points
# from plain C and tests/test_gpu_parity.py drives it through ctypes.
#
# How the opaque inputs are recovered:
#   * P_sub[i] is a closure built by get_projector (src/get_projector.jl:3-103); EVERY branch captures the `constraint`
#     it was built from, so `getfield(P_sub[i], :constraint)` returns the set_definitions (set_type, TD_OP, min, max,
#     app_mode, custom_TD_OP) -- the projector descriptor;
#   * TD_OP[i] is described by set_Prop.tag[i] = (set_type, TD_OP name, app_mode[1], app_mode[2]) (src/setup_constraints.jl:86)
#     and comp_grid.n / .d; a constraint.custom_TD_OP[1] sparse matrix travels as its CSC arrays;
#   * AtA[i] (CDS, N x d_i) and set_Prop.AtA_offsets[i] are passed as they are (src/PARSDMM_precompute_distribute.jl:52-59).
module SipxPARSDMM

using SparseArrays
using TimerOutputs
import SetIntersectionProjection: log_type_PARSDMM, convert_options!

export PARSDMM, release_contexts

const libsipx = get(ENV, "SIPX_LIBRARY", "libsipx.so")

# ---- mirrors of the C structs (include/sipx.h) ---------------------------------------------------------------------
struct SipxSetDesc                     # sipx_set_desc
    op::Int32; proj::Int32
    pmin::Float64; pmax::Float64
    lb::Ptr{Cvoid}; ub::Ptr{Cvoid}
    ncvx::Int32; reserved::Int32
    mode::Int32; dir::Int32
    basis::Ptr{Cvoid}; basis_rows::Int64; basis_cols::Int32; basis_orth::Int32
    component::Int32
    csc_colptr::Ptr{Int64}; csc_rowval::Ptr{Int64}; csc_nzval::Ptr{Cvoid}; csc_rows::Int64
    transform::Int32; pad_::Int32
end

struct SipxOptions                     # sipx_options
    maxit::Int32
    evol_rel_tol::Float64; feas_tol::Float64; obj_tol::Float64
    rho_update_frequency::Int32
    adjust_rho::Int32; adjust_gamma::Int32; adjust_feasibility_rho::Int32
end

mutable struct SipxLog                 # sipx_log: flat row-major arrays the caller allocates for maxit rows
    set_feasibility::Ptr{Float64}; r_dual::Ptr{Float64}; r_pri::Ptr{Float64}
    r_dual_total::Ptr{Float64}; r_pri_total::Ptr{Float64}; obj::Ptr{Float64}; evol_x::Ptr{Float64}
    rho::Ptr{Float64}; gamma::Ptr{Float64}; cg_it::Ptr{Int64}; cg_relres::Ptr{Float64}
    timing_ms::NTuple{7,Float64}
    n_iter::Int32; n_feas_rows::Int32; stopped_feasible::Int32
end

const OPS  = Dict("identity" => 0, "D_x" => 1, "D_y" => 2, "D_z" => 3, "TV" => 4, "D2D" => 4, "D3D" => 4)
const MODE = Dict("matrix" => 0, "tensor" => 0, "fiber" => 1, "slice" => 2)
const SPECIAL = ("DFT", "DCT", "wavelet", "curvelet")          # src/setup_constraints.jl:54
# the seven @timeit sections of src/PARSDMM.jl:40,100,105,113,152,163,229
const SECTIONS = ("initialization", "form rhs for linear system", "argmin x", "argmin y and l update",
                  "stopping conditions check", "adjust rho and gamma", "Q-update")

check(rc) = rc == 0 || error(unsafe_string(ccall((:sipx_last_error, libsipx), Cstring, ())))

"0-based array dimension of an application direction: x = 0, y = 1, z = last"
dir_of(d::AbstractString, ndim::Int) = d == "x" ? 0 : d == "y" ? 1 : d == "z" ? ndim - 1 : 0

"(proj kind, pmin, pmax, lb, ub, transform) of one constraint -- the branches of get_projector (src/get_projector.jl:3-103)"
function projector_fields(c, TF)
    st, op = c.set_type, c.TD_OP
    vecb = (c.min isa AbstractVector) && length(c.min) > 1
    nothing_ = (0.0, 0.0, nothing, nothing, Int32(0))
    if op == "DCT"                                   # x -> C' P(C x), orthonormal DCT-II folded into the projector
        st in ("l2", "annulus") && return (st == "l2" ? 3 : 4, st == "annulus" ? Float64(c.min) : 0.0, Float64(c.max), nothing, nothing, Int32(0))
        st == "l1" && return (2, 0.0, Float64(c.max), nothing, nothing, Int32(1))
        st == "cardinality" && return (5, 0.0, Float64(c.max), nothing, nothing, Int32(1))
        st == "bounds" && !vecb && return (0, Float64(c.min), Float64(c.max), nothing, nothing, Int32(1))
        st == "bounds" && return (1, 0.0, 0.0, convert(Vector{TF}, c.min), convert(Vector{TF}, c.max), Int32(1))
        error("set type $st behind the DCT is not built in libsipx")
    elseif op == "DFT"
        st == "l1" && return (7, 0.0, Float64(c.max), nothing, nothing, Int32(0))                 # SIPX_PROJ_L1_DFT
        st == "bounds" && vecb && return (12, 0.0, 0.0, nothing, convert(Vector{TF}, c.max), Int32(0))   # SIPX_PROJ_BOUNDS_DFT (mask)
        st in ("l2", "annulus") && return (st == "l2" ? 3 : 4, st == "annulus" ? Float64(c.min) : 0.0, Float64(c.max), nothing, nothing, Int32(0))
        error("of the DFT-domain sets libsipx builds the l1 ball, masking bounds, the l2 ball and the annulus")
    elseif op in SPECIAL
        error("operator $op is outside libsipx (JOLI wavelet / curvelet transforms)")
    end
    st == "bounds"      && !vecb && return (0, Float64(c.min), Float64(c.max), nothing, nothing, Int32(0))
    st == "bounds"      && return (1, 0.0, 0.0, convert(Vector{TF}, c.min), convert(Vector{TF}, c.max), Int32(0))
    st == "l1"          && return (2, 0.0, Float64(c.max), nothing_[3:5]...)
    st == "l2"          && return (3, 0.0, Float64(c.max), nothing_[3:5]...)
    st == "annulus"     && return (4, Float64(c.min), Float64(c.max), nothing_[3:5]...)
    st == "cardinality" && return (5, 0.0, Float64(convert(Integer, c.max)), nothing_[3:5]...)
    st == "prox_l1"     && return (6, 0.0, Float64(c.max), nothing_[3:5]...)
    st == "rank"        && return (8, 0.0, Float64(convert(Integer, c.max)), nothing_[3:5]...)
    st == "nuclear"     && return (9, 0.0, Float64(c.max), nothing_[3:5]...)
    st == "histogram"   && return (10, 0.0, 0.0, convert(Vector{TF}, c.min), convert(Vector{TF}, c.max), Int32(0))
    st == "subspace"    && return (11, 0.0, 0.0, nothing, nothing, Int32(0))
    error("set type $st is not part of libsipx")
end

"log.timing with the reference's seven section names, filled from the engine's accumulators (milliseconds)"
function timer_from_sections(ms::NTuple{7,Float64}, ncalls::NTuple{7,Int})
    to = TimerOutput()
    for (k, name) in enumerate(SECTIONS)
        # TimerOutputs 0.5.x (the reference's compat bound): a section is a child TimerOutput whose accumulated_data holds
        # (ncalls, time in ns, allocated bytes)
        @timeit to name nothing
        td = to.inner_timers[name].accumulated_data
        td.ncalls = ncalls[k]
        td.time   = round(Int64, ms[k] * 1e6)
        td.allocs = 0
    end
    return to
end

# ---- options.parallel = true: one process per GPU, RCCL inside the engine (include/sipx.h, "sharded solve") ----------------
# Launch (one node, 8 GPUs), every process running the SAME script:
#     for r in 0:7:  SIPX_RANK=r SIPX_WORLD=8 SIPX_DEVICE=r SIPX_ID_FILE=/dev/shm/sipx.$JOB SIPX_NONCE=$JOB julia --project script.jl &
# (or under mpiexec with SIPX_RANK / SIPX_WORLD taken from the MPI environment).  Rank 0 obtains the 128-byte ncclUniqueId
# from the engine and publishes it through SIPX_ID_FILE; the others wait for the file.  SIPX_DECOMP=slab (default where the
# set list allows it: bounds / l1 / l2 / annulus on the identity or D_x / D_y / D_z / TV) divides the WHOLE iteration by
# z-slab; SIPX_DECOMP=sets is the reference's own split by constraint set.  SIPX_DECOMP=slab may also be ASKED for with the
# other set lists (round 5: slice-wise rank / nuclear norm on z-slices, cardinality, the l1 ball behind the DFT run inside the slab
# iteration, any other projector through an owner rank; sipx.h, sipx_set_decomp).  Every rank returns the same x and log;
# with the set decomposition l[i], y[i] are filled on the rank that owns set i only.
localize(v) = (isdefined(Main, :DistributedArrays) && v isa Main.DistributedArrays.DArray) ? convert(Vector, v) : v

function slab_decomposable(set_Prop)
    for t in set_Prop.tag                       # (set_type, TD_OP, app_mode[1], app_mode[2])
        (t[1] in ("bounds", "l1", "l2", "annulus", "prox_l1") && t[2] in ("identity", "D_x", "D_y", "D_z", "TV", "D2D", "D3D") &&
         t[3] in ("matrix", "tensor")) || return false
    end
    return true
end

function attach_comm!(ctx, set_Prop)
    world = parse(Int, ENV["SIPX_WORLD"]); rank = parse(Int, ENV["SIPX_RANK"])
    idfile = ENV["SIPX_ID_FILE"]
    id = zeros(UInt8, 128)
    # The id file must belong to THIS launch: a rerun with the same path would otherwise hand the previous job's ncclUniqueId to
    # ranks that find the stale file before rank 0 has replaced it (ncclCommInitRank then hangs).  Every record starts with a
    # nonce the launcher gives all ranks (SIPX_NONCE; without one, a record older than two minutes is taken to be stale).
    nonce = rpad(get(ENV, "SIPX_NONCE", ""), 32)[1:32]
    if rank == 0
        rm(idfile; force=true)
        check(ccall((:sipx_rccl_unique_id, libsipx), Cint, (Ptr{UInt8},), id))
        write(idfile * ".tmp", vcat(Vector{UInt8}(nonce), id)); mv(idfile * ".tmp", idfile; force=true)   # published atomically
    else
        t_start = time()
        while true
            if isfile(idfile) && filesize(idfile) == 160
                rec = read(idfile)
                fresh = haskey(ENV, "SIPX_NONCE") ? true : (mtime(idfile) >= t_start - 120.0)
                if String(rec[1:32]) == nonce && fresh
                    id = rec[33:160]
                    break
                end
            end
            time() - t_start > 600.0 && error("sipx: no ncclUniqueId for this launch appeared in $idfile within 10 minutes")
            sleep(0.01)
        end
    end
    check(ccall((:sipx_set_comm_rccl, libsipx), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx[], id, world, rank))
    rank == 0 && rm(idfile; force=true)       # ncclCommInitRank is collective: every rank has read the record by now
    decomp = get(ENV, "SIPX_DECOMP", slab_decomposable(set_Prop) ? "slab" : "sets")
    decomp == "slab" && check(ccall((:sipx_set_decomp, libsipx), Cint, (Ptr{Cvoid}, Cint), ctx[], 1))
    # what RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclGetVersion): worth one line in the job log
    nr = Ref{Cint}(0); rk = Ref{Cint}(0); dc = Ref{Cint}(0); ver = zeros(UInt8, 64)
    check(ccall((:sipx_comm_info, libsipx), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ptr{UInt8}, Cint, Ref{Cint}), ctx[], nr, rk, ver, 64, dc))
    rank == 0 && @info "sipx: sharded solve" ranks=nr[] version=unsafe_string(pointer(ver)) decomposition=(dc[] == 1 ? "slab" : "sets")
    nr[] == world || error("RCCL sees $(nr[]) ranks, SIPX_WORLD says $world")
    return nothing
end

# ---- contexts kept between calls (round 5) ----
# The reference's callers wrap PARSDMM as a projector and call it again and again with the SAME AtA / TD_OP / set_Prop / P_sub
# objects (examples/constrained_freq_FWI_simple.jl:468, examples/Constraint_examples_2D.jl:222-223): a call whose arguments are
# those very objects (objectid), on the same grid and precision, finds the device context of the previous call and resets it
# (sipx_reset: no allocation, no plan, no handle) instead of building one.  SIPX_CONTEXT_CACHE=0 switches this off; at most
# SIPX_CONTEXT_CACHE (default 2) contexts are kept, the oldest goes first; release_contexts() frees them (also at exit).
# A caller that mutates AtA / P_sub IN PLACE between calls must call release_contexts() itself -- the arrays were copied to the
# device when the context was built.  Sharded solves (options.parallel) are not cached.
const CONTEXTS = Vector{Pair{Any,Ptr{Cvoid}}}()

function release_contexts()
    for (_, c) in CONTEXTS
        ccall((:sipx_destroy, libsipx), Cvoid, (Ptr{Cvoid},), c)
    end
    empty!(CONTEXTS)
    return nothing
end
atexit(release_contexts)

cache_limit() = something(tryparse(Int, get(ENV, "SIPX_CONTEXT_CACHE", "2")), 2)

function PARSDMM(m         ::Vector{TF},
                 AtA,
                 TD_OP,
                 set_Prop,
                 P_sub,
                 comp_grid,
                 options,
                 x=zeros(TF,length(m)) ::Vector{TF},
                 l=[],
                 y=[]
                 ) where {TF<:Real}
    t_init = time_ns()
    TF in (Float32, Float64) || error("libsipx computes in Float32 or Float64")
    convert_options!(options, TF)                                           # src/PARSDMM.jl:43
    if isreal(m) == false || isreal(x) == false || isreal(l) == false || isreal(y) == false
        error("input for PARSDMM is not real")                              # src/PARSDMM.jl:50-52
    end
    # options.parallel = true (src/PARSDMM.jl:114-131: DArray TD_OP / P_sub, one Julia worker per set): here the solve is
    # sharded over PROCESSES, one per GPU -- every process calls this same method with the same arguments (see
    # attach_comm! below and INTEGRATION.md, section 5); distributed inputs are gathered, every rank builds every set.
    if options.parallel
        TD_OP = localize(TD_OP); P_sub = localize(P_sub); AtA = localize(AtA)
        isempty(l) || (l = localize(l)); isempty(y) || (y = localize(y))
    end
    p  = length(TD_OP)
    pp = options.feasibility_only ? p : p - 1                               # src/PARSDMM.jl:55-56
    length(P_sub) == pp || error("P_sub must hold one projector per constraint set")
    n    = collect(Int64, comp_grid.n)
    h    = collect(Float64, comp_grid.d)
    ndim = length(n)
    N    = prod(n)

    device = parse(Int, get(ENV, "SIPX_DEVICE", get(ENV, "SIPX_RANK", "0")))
    key    = (TF, Tuple(n), Tuple(h), device, Bool(options.feasibility_only), objectid(AtA), objectid(TD_OP), objectid(set_Prop), objectid(P_sub),
              Tuple(sort!([k => v for (k, v) in ENV if startswith(k, "SIPX_")])))
    cached = (!options.parallel && cache_limit() > 0) ? findfirst(e -> isequal(first(e), key), CONTEXTS) : nothing
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    reused = cached !== nothing
    if reused
        ctx[] = last(CONTEXTS[cached])
        deleteat!(CONTEXTS, cached)                                         # (back in at the end of a call that succeeded)
    else
        check(ccall((:sipx_create, libsipx), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Ptr{Int64}, Ptr{Float64}, Cint),
                    ctx, TF == Float32 ? 0 : 1, ndim, n, h, device))
    end
    keep = Any[]                                                            # arrays the descriptors point at, until finalize
    done = false
    try
        for i in (reused ? (1:0) : (1:pp))
            c   = getfield(P_sub[i], :constraint)                            # captured by every branch of get_projector
            tag = set_Prop.tag[i]                                            # (set_type, TD_OP, app_mode[1], app_mode[2])
            (tag[1] == c.set_type && tag[2] == c.TD_OP) || error("set_Prop.tag[$i] does not describe P_sub[$i]")
            (kind, pmin, pmax, lb, ub, transform) = projector_fields(c, TF)
            custom = c.set_type != "subspace" && !(c.custom_TD_OP[1] == [])   # src/setup_constraints.jl:70-72
            opcode = custom ? 5 : (c.TD_OP in SPECIAL ? 0 : OPS[c.TD_OP])    # orthogonal transforms: TD_OP[i] = I (:76-80)
            basis  = c.set_type == "subspace" ? convert(Matrix{TF}, c.custom_TD_OP[1]) : nothing
            colptr = rowval = nzval = nothing
            rows   = 0
            if custom
                A      = c.custom_TD_OP[1]::SparseMatrixCSC
                colptr = convert(Vector{Int64}, A.colptr) .- 1               # 0-based copies
                rowval = convert(Vector{Int64}, A.rowval) .- 1
                nzval  = convert(Vector{TF}, A.nzval)
                rows   = size(A, 1)
            end
            push!(keep, (lb, ub, basis, colptr, rowval, nzval))
            d = SipxSetDesc(opcode, kind, pmin, pmax,
                            lb === nothing ? C_NULL : pointer(lb), ub === nothing ? C_NULL : pointer(ub),
                            set_Prop.ncvx[i] ? 1 : 0, 0,
                            MODE[c.app_mode[1]], dir_of(c.app_mode[2], ndim),
                            basis === nothing ? C_NULL : pointer(basis), basis === nothing ? 0 : size(basis, 1),
                            basis === nothing ? 0 : size(basis, 2), basis === nothing ? 0 : Int32(c.custom_TD_OP[2]),
                            0,                                                # Minkowski component (see INTEGRATION.md)
                            colptr === nothing ? C_NULL : pointer(colptr), rowval === nothing ? C_NULL : pointer(rowval),
                            nzval === nothing ? C_NULL : pointer(nzval), rows,
                            transform, 0)
            # AtA[i] in CDS (N x d_i) with its offsets; a non-banded Q (sparse / JOLI AtA) is outside libsipx
            AtA[i] isa Matrix{TF} || error("libsipx needs every AtA in CDS storage (all operators banded)")
            off = convert(Vector{Int64}, set_Prop.AtA_offsets[i])
            rc = GC.@preserve keep off ccall((:sipx_add_set, libsipx), Cint,
                        (Ptr{Cvoid}, Ref{SipxSetDesc}, Ptr{Cvoid}, Ptr{Int64}, Cint),
                        ctx[], d, AtA[i], off, size(AtA[i], 2))
            rc < 0 && check(1)                                               # sipx_add_set returns the set index, -1 on error
        end

        options.parallel && !reused && attach_comm!(ctx, set_Prop)           # before sipx_finalize: this context is one RANK

        # l, y as PARSDMM_initialize allocates them (src/PARSDMM_initialize.jl:120-127)
        if isempty(l); l = Vector{Vector{TF}}(undef, p); for i in 1:p; l[i] = zeros(TF, size(TD_OP[i], 1)); end; end
        if isempty(y); y = Vector{Vector{TF}}(undef, p); for i in 1:p; y[i] = zeros(TF, size(TD_OP[i], 1)); end; end
        rho_ini = convert(Vector{Float64}, options.rho_ini)
        feas0   = zeros(Float64, max(pp, 1))
        lp = Ptr{Cvoid}[pointer(v) for v in l]; yp = Ptr{Cvoid}[pointer(v) for v in y]
        if reused                                                            # the same sets once more: src/PARSDMM.jl:58-61 without its allocations
            GC.@preserve l y check(ccall((:sipx_reset, libsipx), Cint,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Float64, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Float64}),
                  ctx[], m, rho_ini, length(rho_ini), Float64(options.gamma_ini), options.zero_ini_guess, x, lp, yp, feas0))
        else
            GC.@preserve keep l y check(ccall((:sipx_finalize, libsipx), Cint,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Float64, Cint, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Float64}),
                  ctx[], m, rho_ini, length(rho_ini), Float64(options.gamma_ini), options.feasibility_only, options.zero_ini_guess,
                  x, lp, yp, feas0))
        end
        empty!(keep)
        ms_init = (time_ns() - t_init) * 1e-6

        # ---- the main loop, natively (src/PARSDMM.jl:97-257) ----
        maxit = Int(options.maxit)
        sf = zeros(Float64, pp, maxit); rd = zeros(Float64, p, maxit); rpm = zeros(Float64, p, maxit)     # row-major [maxit][.]
        rdt = zeros(Float64, maxit); rpt = zeros(Float64, maxit); obj = zeros(Float64, maxit); evo = zeros(Float64, maxit)
        rho = zeros(Float64, p, maxit); gam = zeros(Float64, p, maxit); cgi = zeros(Int64, maxit); cgr = zeros(Float64, maxit)
        o   = SipxOptions(maxit, Float64(options.evol_rel_tol), Float64(options.feas_tol), Float64(options.obj_tol),
                          Int32(options.rho_update_frequency), options.adjust_rho, options.adjust_gamma,
                          options.adjust_feasibility_rho)
        lg  = SipxLog(pointer(sf), pointer(rd), pointer(rpm), pointer(rdt), pointer(rpt), pointer(obj), pointer(evo),
                      pointer(rho), pointer(gam), pointer(cgi), pointer(cgr), ntuple(_ -> 0.0, 7), 0, 0, 0)
        GC.@preserve sf rd rpm rdt rpt obj evo rho gam cgi cgr check(ccall((:sipx_parsdmm, libsipx), Cint,
              (Ptr{Cvoid}, Ref{SipxOptions}, Ref{SipxLog}), ctx[], o, lg))

        it, nf = Int(lg.n_iter), Int(lg.n_feas_rows)
        if lg.stopped_feasible != 0                                          # feasible input: x = m, one log row (src/PARSDMM.jl:63-82)
            copy!(x, m)
        else
            GC.@preserve l y check(ccall((:sipx_download, libsipx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
                                        ctx[], x, lp, yp))
        end
        # log_type_PARSDMM (src/SetIntersectionProjection.jl:95-108), truncated like output_check_PARSDMM (src/PARSDMM.jl:261-278);
        # the C arrays are row-major [row][column] = column-major (column, row) here, hence the transposes
        ms = ntuple(k -> k == 1 ? ms_init : lg.timing_ms[k], 7)
        nc = ntuple(k -> k == 1 ? 1 : it, 7)
        log_PARSDMM = log_type_PARSDMM(permutedims(sf[:, 1:nf]), permutedims(rd[:, 1:it]), permutedims(rpm[:, 1:it]),
                                       rdt[1:it], rpt[1:it], obj[1:it], evo[1:it],
                                       permutedims(rho[:, 1:it]), permutedims(gam[:, 1:it]), cgi[1:it], cgr[1:it],
                                       timer_from_sections(ms, nc))
        done = true
        return x, log_PARSDMM, l, y
    finally
        if done && !options.parallel && cache_limit() > 0                    # keep the context for the next call with these objects
            while length(CONTEXTS) >= cache_limit()
                ccall((:sipx_destroy, libsipx), Cvoid, (Ptr{Cvoid},), last(popfirst!(CONTEXTS)))
            end
            push!(CONTEXTS, key => ctx[])
        else
            ccall((:sipx_destroy, libsipx), Cvoid, (Ptr{Cvoid},), ctx[])
        end
    end
end

# ---- alternative: keep the reference's own loop (its @timeit sections, stop_PARSDMM, logging) and replace each phase ----
# Inside `for i=1:maxit` of src/PARSDMM.jl, with ctx from sipx_create / sipx_add_set / sipx_finalize as above:
#   :101  rhs_compose(...)            ->  ccall((:sipx_rhs_compose, libsipx), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx[], rho64)
#   :106-107 copy!(x_old,x); argmin_x ->  ccall((:sipx_argmin_x, libsipx), Cint, (Ptr{Cvoid}, Cint, Ref{Float64}, Ref{Int64}, Ref{Float64}, Ref{Cint}),
#                                               ctx[], i, tol_ref, cg_it, relres, flag)
#   :133  update_y_l(...)             ->  ccall((:sipx_update_y_l, libsipx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
#                                               ctx[], i, flags, rho64, gamma64, r_pri, r_dual, feas)
#         flags = (mod(i,10)==0 ? 1 : 0) | (((adjust_rho || adjust_gamma) && mod(i,rho_update_frequency)==0) ? 2 : 0) | (i==1 ? 4 : 0)
#   :140,145 obj, evol_x              ->  ccall((:sipx_log_scalars, libsipx), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), ctx[], obj_i, evol_i)
#   :153  stop_PARSDMM(...)               unchanged Julia
#   :164-206 l_hat, snapshots, adapt  ->  (fused into sipx_update_y_l by flags 2 / 4) then
#                                         ccall((:sipx_adapt_rho_gamma, libsipx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}), ctx[], adjust_rho, adjust_gamma, rho64, gamma64)
#   :213-226 feasibility doubling, clamp  unchanged Julia
#   :230-243 Q_update! + prox rebind  ->  ccall((:sipx_q_update, libsipx), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx[], rho64, Float64.(log_PARSDMM.rho[i,:]))
# tests/c_abi/phases.c runs exactly this sequence from C and checks it against sipx_parsdmm bit for bit.

end # module
