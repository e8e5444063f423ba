# gpu_evaluator.jl -- Julia side of the drop-in boundary (include/dto.h) for DirectTrajectoryOptimization.jl.
#
# `include("gpu_evaluator.jl")` inside the package module (after src/solver.jl) adds:
#
#   GPUEvaluator            an MOI.AbstractNLPEvaluator whose nine methods (src/moi.jl:1-125) are `ccall`s into
#                           libdto_hip.so -- Ipopt keeps running on top of it unchanged (A/B against the CPU closures);
#   gpu_solver_data(...)    SolverData (src/data.jl:229-255) with the GPU evaluator registered instead of `nlp`;
#   solve!(solver; gpu)     the switch of src/solver.jl:45-47: gpu = true replaces MOI.optimize!(Ipopt) by dto_solve;
#   solve_batch(...)        B independent instances (MPC rollouts / seeds) in one call, dto_solve_batch;
#   resolve_warm!(...)      receding-horizon re-solve that keeps the interior-point state on the device.
#
# The compiled model plugin (gfx950 code object with the per-stage functions, INTEGRATION.md section 2) and the stage-kind
# vector come from the code generator; their paths are arguments here.
#
# NOTE: Julia is not installed in the build environment of this repository -- this file is reviewed, not executed, code.
# Every ccall signature below is the one tests/test_capi_symbols.py checks against the C compiler's struct layout and
# tests/c_abi/drive_solve.c exercises from plain C.

const libdto = get(ENV, "DTO_AMD_LIB", joinpath(@__DIR__, "..", "directtrajectoryoptimization.jl_amd", "libdto_hip.so"))
const DTO_ABI_VERSION = Cint(4)

struct DtoSpec                      # include/dto.h: dto_problem_spec
    abi_version::Cint
    model_library::Cstring
    horizon::Cint
    stage_kind::Ptr{Int32}
    variable_lower::Ptr{Float64}
    variable_upper::Ptr{Float64}
    parameters::Ptr{Float64}
    num_parameters::Int64
    evaluate_hessian::Cint
end

struct DtoOptions                   # include/dto.h: dto_options (src/options.jl:6-36 + interior-point constants)
    tol::Float64
    s_max::Float64
    max_iter::Cint
    dual_inf_tol::Float64
    constr_viol_tol::Float64
    compl_inf_tol::Float64
    mu_init::Float64
    delta_c::Float64
    delta_w_init::Float64
    check_every::Cint
    max_cpu_time::Float64
    acceptable_tol::Float64
    acceptable_iter::Cint
    acceptable_dual_inf_tol::Float64
    acceptable_constr_viol_tol::Float64
    acceptable_compl_inf_tol::Float64
    acceptable_obj_change_tol::Float64
    diverging_iterates_tol::Float64
    mu_target::Float64
    line_search::Cint               # DTO_LS_FILTER = 0, DTO_LS_PENALTY_FILTER = 1 (default)
    penalty_switch_theta::Float64
    hessian_approximation::Cint     # DTO_HESSIAN_EXACT = 0, DTO_HESSIAN_LBFGS = 1 (Ipopt's limited-memory mode)
    kkt_refinement::Cint            # ABI 4: passes of iterative refinement per KKT step (0)
end

struct DtoBatch                     # include/dto.h: dto_batch (DEVICE pointers)
    B::Int64
    x::Ptr{Float64}
    ldx::Int64
    params::Ptr{Float64}
    ldp::Int64
    stream::Ptr{Cvoid}
end

dto_check(rc) = rc == 0 || error(unsafe_string(ccall((:dto_last_error, libdto), Cstring, ())))

const DTO_LS_FILTER = Cint(0)          # Ipopt's filter line search from the first iteration: what the reference itself runs
const DTO_LS_PENALTY_FILTER = Cint(1)  # l1-penalty line search far from the constraint manifold, then the filter (the library's default)

# limited_memory: what the reference means by evaluate_hessian = false (src/solver.jl:7: Ipopt's hessian_approximation stays
# "limited-memory") -- the callers below pass !ev.hessian_lagrangian.  (A plugin emitted with evaluate_hessian = false carries
# no second derivatives at all; the library then runs its per-stage SR1 blocks and `hessian_mode(ev)` says so.)
# line_search / penalty_switch_theta / kkt_refinement: the library's own knobs (include/dto.h) -- line_search = DTO_LS_FILTER
# reproduces the reference's globalisation.
DtoOptions(o::Options; limited_memory::Bool = false, line_search::Integer = DTO_LS_PENALTY_FILTER,
           penalty_switch_theta::Real = 1.0, kkt_refinement::Integer = 0) =
    DtoOptions(o.tol, o.s_max, o.max_iter, o.dual_inf_tol, o.constr_viol_tol, o.compl_inf_tol,
               0.1, 1.0e-8, 1.0e-4, 10, o.max_cpu_time,
               o.acceptable_tol, o.acceptable_iter, o.acceptable_dual_inf_tol,
               o.acceptable_constr_viol_tol, o.acceptable_compl_inf_tol, o.acceptable_obj_change_tol,
               o.diverging_iterates_tol, o.mu_target, Cint(line_search), Float64(penalty_switch_theta),
               Cint(limited_memory ? 1 : 0), Cint(kkt_refinement))

mutable struct GPUEvaluator <: MOI.AbstractNLPEvaluator
    handle::Ptr{Cvoid}
    hessian_lagrangian::Bool
    jacobian_sparsity::Vector{Tuple{Int,Int}}
    hessian_lagrangian_sparsity::Vector{Tuple{Int,Int}}
    num_variables::Int
    num_constraint::Int
end

"""
    GPUEvaluator(nlp::NLPData, plugin::String, kinds::Vector{Int32})

Device-resident twin of `nlp` (src/data.jl:106-121): same variable order, constraint order, Jacobian / Hessian structures
(the library returns the identical lists; `nlp`'s own are reused here).
"""
function GPUEvaluator(nlp::NLPData, plugin::String, kinds::Vector{Int32})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    lo, hi = nlp.variable_bounds
    par = nlp.parameters                                        # flattened w_1..w_T (src/data.jl:218)
    GC.@preserve plugin kinds lo hi par begin
        spec = DtoSpec(DTO_ABI_VERSION, Base.unsafe_convert(Cstring, plugin), Cint(length(kinds)), pointer(kinds),
                       pointer(lo), pointer(hi), isempty(par) ? Ptr{Float64}(C_NULL) : pointer(par), length(par),
                       Cint(nlp.hessian_lagrangian))
        dto_check(ccall((:dto_problem_create, libdto), Cint, (Ref{DtoSpec}, Ref{Ptr{Cvoid}}), spec, h))
    end
    ev = GPUEvaluator(h[], nlp.hessian_lagrangian, nlp.jacobian_sparsity, nlp.hessian_lagrangian_sparsity,
                      nlp.num_variables, nlp.num_constraint)
    finalizer(e -> ccall((:dto_problem_destroy, libdto), Cint, (Ptr{Cvoid},), e.handle), ev)
    return ev
end

# --- the nine MOI methods of src/moi.jl: same signatures, same ownership (caller-owned buffers, fully overwritten)
function MOI.eval_objective(e::GPUEvaluator, x::Vector{Float64})                          # src/moi.jl:1-13
    f = Ref{Float64}(0.0)
    dto_check(ccall((:dto_eval_f, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), e.handle, x, f))
    return f[]
end
function MOI.eval_objective_gradient(e::GPUEvaluator, g, x)                              # src/moi.jl:15-30
    dto_check(ccall((:dto_eval_grad_f, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, x, g))
    return nothing
end
function MOI.eval_constraint(e::GPUEvaluator, c, x)                                      # src/moi.jl:32-50
    dto_check(ccall((:dto_eval_g, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, x, c))
    return nothing
end
function MOI.eval_constraint_jacobian(e::GPUEvaluator, J, x)                             # src/moi.jl:52-70
    dto_check(ccall((:dto_eval_jac_g, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, x, J))
    return nothing
end
function MOI.eval_hessian_lagrangian(e::GPUEvaluator, H, x, σ, μ)                        # src/moi.jl:72-120
    dto_check(ccall((:dto_eval_h, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                    e.handle, x, σ, μ, H))
    return nothing
end
MOI.features_available(e::GPUEvaluator) = e.hessian_lagrangian ? [:Grad, :Jac, :Hess] : [:Grad, :Jac]   # src/moi.jl:122
MOI.initialize(e::GPUEvaluator, features) = nothing                                                        # src/moi.jl:123
MOI.jacobian_structure(e::GPUEvaluator) = e.jacobian_sparsity                                              # src/moi.jl:124
MOI.hessian_lagrangian_structure(e::GPUEvaluator) = e.hessian_lagrangian_sparsity                          # src/moi.jl:125

"""
    gpu_solver_data(nlp, plugin, kinds; options = Options())

`SolverData(nlp; options)` (src/data.jl:229-255) with the GPU evaluator handed to Ipopt instead of `nlp`.
Returns `(data, evaluator)`.
"""
function gpu_solver_data(nlp::NLPData, plugin::String, kinds::Vector{Int32}; options = Options())
    ev = GPUEvaluator(nlp, plugin, kinds)
    nlp_bounds = MOI.NLPBoundsPair.(nlp.constraint_bounds...)
    block_data = MOI.NLPBlockData(nlp_bounds, ev, true)
    optimizer = Ipopt.Optimizer()
    for name in fieldnames(typeof(options))
        optimizer.options[String(name)] = getfield(options, name)
    end
    z = MOI.add_variables(optimizer, nlp.num_variables)
    for i = 1:nlp.num_variables
        MOI.add_constraint(optimizer, z[i], MOI.LessThan(nlp.variable_bounds[2][i]))
        MOI.add_constraint(optimizer, z[i], MOI.GreaterThan(nlp.variable_bounds[1][i]))
    end
    MOI.set(optimizer, MOI.NLPBlock(), block_data)
    MOI.set(optimizer, MOI.ObjectiveSense(), MOI.MIN_SENSE)
    return SolverData(nlp_bounds, block_data, optimizer, z), ev
end

"""
    solve!(solver, ev::GPUEvaluator; options = Options())

`solve!(solver)` (src/solver.jl:45-47) with the whole interior-point iteration on the GPU (`dto_solve`): the initial
guess is what `initialize_states!` / `initialize_controls!` stored (src/solver.jl:23-39); afterwards `get_trajectory`
(src/solver.jl:41-43) returns the solution.  Returns (status, iterations); status 1 = converged, 2 = iteration limit,
4 = acceptable level, see include/dto.h.
"""
function solve!(solver::Solver, ev::GPUEvaluator; options = Options(), line_search = DTO_LS_PENALTY_FILTER,
                penalty_switch_theta = 1.0, kkt_refinement = 0)
    opt = DtoOptions(options; limited_memory = !ev.hessian_lagrangian, line_search = line_search,
                     penalty_switch_theta = penalty_switch_theta, kkt_refinement = kkt_refinement)
    x0 = Float64[something(MOI.get(solver.data.optimizer, MOI.VariablePrimalStart(), v), 0.0) for v in solver.data.variables]
    x = similar(x0)
    μ = zeros(max(1, ev.num_constraint))
    status = Ref{Int32}(0)
    iters = Ref{Int32}(0)
    dto_check(ccall((:dto_solve, libdto), Cint,
                    (Ptr{Cvoid}, Ref{DtoOptions}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int32}, Ref{Int32}),
                    ev.handle, opt, x0, x, μ, status, iters))
    trajectory!(solver.nlp.trajopt.states, solver.nlp.trajopt.actions, x,                # src/data.jl:258-267
                solver.nlp.indices.states, solver.nlp.indices.actions)
    return Int(status[]), Int(iters[])
end

"""
    hessian_mode(ev)

What stood in for the Hessian of the Lagrangian in the last solve on `ev`: 0 exact second derivatives, 1 limited-memory BFGS,
2 per-stage SR1 blocks (a plugin emitted without Hessians), -1 nothing solved yet (`dto_solver_hessian_mode`).
"""
function hessian_mode(ev::GPUEvaluator)
    m = Ref{Cint}(-1)
    dto_check(ccall((:dto_solver_hessian_mode, libdto), Cint, (Ptr{Cvoid}, Ref{Cint}), ev.handle, m))
    return Int(m[])
end

# --- device buffers without a HIP binding in the host language
function dto_device_array(host::Array{Float64})
    p = Ref{Ptr{Cvoid}}(C_NULL)
    dto_check(ccall((:dto_device_alloc, libdto), Cint, (Ref{Ptr{Cvoid}}, Int64), p, sizeof(host)))
    dto_check(ccall((:dto_copy_to_device, libdto), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), p[], host, sizeof(host)))
    return Ptr{Float64}(p[])
end
dto_device_free(p) = ccall((:dto_device_free, libdto), Cint, (Ptr{Cvoid},), p)

"""
    solve_batch(ev, X0::Matrix{Float64}; options = Options(), parameters = nothing)

Solve the columns of `X0` (num_variables × B initial guesses) as B independent problems of the same structure in one
`dto_solve_batch` call; `parameters` (num_parameters × B) gives every instance its own flattened `parameters` vector
(src/solver.jl:10), e.g. the measured state of each MPC rollout.  Returns (X, status, iterations).
"""
function solve_batch(ev::GPUEvaluator, X0::Matrix{Float64}; options = Options(), parameters = nothing,
                     line_search = DTO_LS_PENALTY_FILTER, penalty_switch_theta = 1.0, kkt_refinement = 0)
    nz, B = size(X0)
    nz == ev.num_variables || error("X0 must have num_variables rows")
    opt = DtoOptions(options; limited_memory = !ev.hessian_lagrangian, line_search = line_search,
                     penalty_switch_theta = penalty_switch_theta, kkt_refinement = kkt_refinement)
    dx0 = dto_device_array(X0)                                   # column-major nz × B == instance-major [B][nz]
    dx = dto_device_array(zeros(nz, B))
    dpar = parameters === nothing ? Ptr{Float64}(C_NULL) : dto_device_array(Matrix{Float64}(parameters))
    ldp = parameters === nothing ? 0 : size(parameters, 1)
    status = zeros(Int32, B)
    iters = zeros(Int32, B)
    batch = DtoBatch(B, dx0, nz, dpar, ldp, C_NULL)
    dto_check(ccall((:dto_solve_batch, libdto), Cint,
                    (Ptr{Cvoid}, Ref{DtoOptions}, Ref{DtoBatch}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Int32}),
                    ev.handle, opt, batch, dx, nz, C_NULL, 0, status, iters))
    X = zeros(nz, B)
    dto_check(ccall((:dto_copy_to_host, libdto), Cint, (Ptr{Float64}, Ptr{Cvoid}, Int64), X, dx, sizeof(X)))
    dto_device_free(dx0); dto_device_free(dx)
    parameters === nothing || dto_device_free(dpar)
    return X, status, iters
end

"""
    resolve_warm!(ev, B, parameters; options = Options(), mu0 = 0.0)

Receding-horizon re-solve of the batch solved last on `ev` (same B): the multipliers, bound multipliers, slacks and the
barrier parameter stay on the device (`dto_solver_begin_warm`), only the per-instance `parameters` (num_parameters × B)
change.  Returns (X, status, iterations).
"""
function resolve_warm!(ev::GPUEvaluator, B::Int, parameters::Matrix{Float64}; options = Options(), mu0 = 0.0,
                       line_search = DTO_LS_PENALTY_FILTER, penalty_switch_theta = 1.0, kkt_refinement = 0)
    opt = DtoOptions(options; limited_memory = !ev.hessian_lagrangian, line_search = line_search,
                     penalty_switch_theta = penalty_switch_theta, kkt_refinement = kkt_refinement)
    nz = ev.num_variables
    dpar = dto_device_array(parameters)
    dx = dto_device_array(zeros(nz, B))
    batch = DtoBatch(B, Ptr{Float64}(C_NULL), nz, dpar, size(parameters, 1), C_NULL)
    dto_check(ccall((:dto_solver_begin_warm, libdto), Cint, (Ptr{Cvoid}, Ref{DtoOptions}, Ref{DtoBatch}, Float64),
                    ev.handle, opt, batch, mu0))
    status = zeros(Int32, B)
    iters = zeros(Int32, B)
    dto_check(ccall((:dto_solver_run, libdto), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Cvoid}),
                    ev.handle, dx, nz, C_NULL, 0, status, iters, C_NULL))
    X = zeros(nz, B)
    dto_check(ccall((:dto_copy_to_host, libdto), Cint, (Ptr{Float64}, Ptr{Cvoid}, Int64), X, dx, sizeof(X)))
    dto_device_free(dx); dto_device_free(dpar)
    return X, status, iters
end
