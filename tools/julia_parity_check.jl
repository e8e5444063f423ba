# Pin this repository's golden fixtures against the REAL reference (DirectTrajectoryOptimization.jl + Symbolics + MOI).
#
# Not executed in the build environment (no Julia there): this is the script a maintainer runs once Julia and the
# reference's dependencies are available, to close the one gap the parity report states everywhere ("Symbolics' structural
# rules on corner cases and Ipopt's iterates cannot be checked here", SURVEY.md 8c / DESIGN.md section 6).
#
#     julia --project=/path/to/DirectTrajectoryOptimization.jl tools/julia_parity_check.jl tests/golden
#     julia --project=/path/to/DirectTrajectoryOptimization.jl tools/julia_parity_check.jl tests/golden --solve   # + Ipopt
#
# For every fixture it rebuilds the same problem with the reference's own constructors and the reference's own model
# functions (parsed out of examples/*/*.jl under DTO_REFERENCE, default: the package directory), evaluates the five MOI methods of src/moi.jl at the fixture's point (z, mu, sigma) and compares
#   * totals and both structures bit for bit            (src/data.jl:61-220)
#   * f, grad f, c, J, H                                to 1e-8 relative (the tolerance of BASELINE.json's north star).
# The fixtures were produced by oracle/ (sympy, 30-digit mpmath) -- agreement here pins the oracle, and with it every
# parity test of the HIP path, to the reference itself.
using DirectTrajectoryOptimization
using LinearAlgebra, JSON
const DTO = DirectTrajectoryOptimization
const MOI = DTO.MOI

# ---------------------------------------------------------------- models: the reference's OWN definitions, read at run time
# Symbolics' sparsity patterns depend on how an expression is written (`length1 * length1` vs `length1^2`, `[f1; f2] .* v` vs a
# diagonal matrix product, `0` vs `0.0` in a matrix literal: SURVEY.md App. A.4), so the models are not restated here: the
# top-level `function` definitions (and the two obstacle constants) are taken verbatim out of the reference's example files,
# each example into its own module, WITHOUT running the rest of the example (no solve, no plotting).
const REF = get(ENV, "DTO_REFERENCE", pkgdir(DirectTrajectoryOptimization))

defname(ex) = begin
    (ex isa Expr && (ex.head == :function || ex.head == :(=))) || return nothing
    sig = ex.args[1]
    while sig isa Expr && (sig.head == :where || sig.head == :(::)); sig = sig.args[1]; end
    if ex.head == :(=) && sig isa Symbol; return sig; end                       # plain assignment  name = value
    (sig isa Expr && sig.head == :call && sig.args[1] isa Symbol) ? sig.args[1] : nothing
end

function load_example(relpath, names)
    m = Module(gensym(:reference_example))
    Core.eval(m, :(using LinearAlgebra))
    src = read(joinpath(REF, relpath), String)
    pos, found = 1, Symbol[]
    while pos <= ncodeunits(src)
        ex, pos = Meta.parse(src, pos; raise=false, greedy=true)
        n = defname(ex)
        if n !== nothing && n in names
            Core.eval(m, ex)
            push!(found, n)
        end
    end
    all(n -> n in found, names) || error("$relpath: definitions not found: $(setdiff(names, found))")
    m
end

const EX_PENDULUM = load_example("examples/pendulum/pendulum.jl", [:pendulum, :midpoint_implicit])
const EX_CARTPOLE = load_example("examples/cartpole/cartpole.jl", [:cartpole, :rk3_explicit, :rk3_implicit])
const EX_ACROBOT = load_example("examples/acrobot/acrobot.jl", [:acrobot, :midpoint_implicit])
const EX_CAR = load_example("examples/car/car.jl", [:car, :midpoint_implicit, :p_obs, :r_obs, :obs])

# ---------------------------------------------------------------- problems of the fixtures (oracle/sympy_models.py:build)
function build(model, T)
    eh = true
    if model == "pendulum"
        n, m = 2, 1; x1, xT = [0.0, 0.0], [pi, 0.0]
        d = Dynamics(EX_PENDULUM.midpoint_implicit, n, n, m, evaluate_hessian=eh)     # examples/pendulum/pendulum.jl:36-39
        ct = Cost((x, u, w) -> 0.1 * dot(x[1:2], x[1:2]) + 0.1 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.1 * dot(x[1:2], x[1:2]), n, 0, evaluate_hessian=eh)
        cons = [Constraint((x, u, w) -> x - x1, n, m, evaluate_hessian=eh), [Constraint() for t = 2:T-1]...,
                Constraint((x, u, w) -> x - xT, n, 0, evaluate_hessian=eh)]
        bnds = [[Bound(n, m) for t = 1:T-1]..., Bound(n, 0)]
    elseif model == "cartpole"
        n, m = 4, 1; x1, xT = zeros(4), [0.0, pi, 0.0, 0.0]
        d = Dynamics(EX_CARTPOLE.rk3_implicit, n, n, m, evaluate_hessian=eh)         # examples/cartpole/cartpole.jl:44-56
        ct = Cost((x, u, w) -> 0.5 * 1.0e-2 * dot(x - xT, x - xT) + 0.5 * 1.0e-1 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.5 * 1.0e2 * dot(x - xT, x - xT), n, 0, evaluate_hessian=eh)
        cons = [Constraint((x, u, w) -> x - x1, n, m, evaluate_hessian=eh), [Constraint() for t = 2:T-1]...,
                Constraint((x, u, w) -> x - xT, n, 0, evaluate_hessian=eh)]
        bnds = [[Bound(n, m, action_lower=[-3.0], action_upper=[3.0]) for t = 1:T-1]..., Bound(n, 0)]
    elseif model == "acrobot" || model == "acrobot_bounds"
        n, m = 4, 1; x1 = zeros(4)
        d = Dynamics(EX_ACROBOT.midpoint_implicit, n, n, m, evaluate_hessian=eh)      # examples/acrobot/acrobot.jl:88-91
        ct = Cost((x, u, w) -> 0.1 * dot(x[3:4], x[3:4]) + 0.1 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.1 * dot(x[3:4], x[3:4]), n, 0, evaluate_hessian=eh)
        if model == "acrobot"
            xT = [pi, 0.0, 0.0, 0.0]
            cons = [Constraint((x, u, w) -> x - x1, n, m, evaluate_hessian=eh), [Constraint() for t = 2:T-1]...,
                    Constraint((x, u, w) -> x - xT, n, 0, evaluate_hessian=eh)]
            bnds = [[Bound(n, m) for t = 1:T-1]..., Bound(n, 0)]
        else
            xT = [0.0, pi, 0.0, 0.0]
            cons = [Constraint() for t = 1:T]
            bnds = [Bound(n, m, state_lower=x1, state_upper=x1), [Bound(n, m) for t = 2:T-1]...,
                    Bound(n, 0, state_lower=xT, state_upper=xT)]
        end
    elseif model == "car"
        n, m = 3, 2; x1, xT = zeros(3), [1.0, 1.0, 0.0]
        d = Dynamics(EX_CAR.midpoint_implicit, n, n, m, evaluate_hessian=eh)          # examples/car/car.jl:23-26
        ct = Cost((x, u, w) -> 0.0 * dot(x - xT, x - xT) + 1.0 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.0 * dot(x - xT, x - xT), n, 0, evaluate_hessian=eh)
        obs = EX_CAR.obs                                                               # examples/car/car.jl:51-56
        cons = [[Constraint(obs, n, m, indices_inequality=collect(1:1), evaluate_hessian=eh) for t = 1:T-1]...,
                Constraint(obs, n, 0, indices_inequality=collect(1:1), evaluate_hessian=eh)]
        lo, hi = [-0.5, -0.5], [0.5, 0.5]
        bnds = [Bound(n, m, state_lower=x1, state_upper=x1, action_lower=lo, action_upper=hi),
                [Bound(n, m, action_lower=lo, action_upper=hi) for t = 2:T-1]...,
                Bound(n, 0, state_lower=xT, state_upper=xT)]
    else
        error("unknown model $model")
    end
    Solver([d for t = 1:T-1], [[ct for t = 1:T-1]..., cT], cons, bnds, evaluate_hessian=eh)
end

relerr(a, b) = maximum(abs.(a .- b) ./ max.(abs.(b), 1.0e-3 * maximum(abs.(b)) + 1.0e-300))

function check(path)
    g = JSON.parsefile(path)
    (g isa AbstractDict && haskey(g, "model")) || return true   # (the full-size structure file has another layout)
    solver = build(g["model"], g["T"])
    nlp = solver.nlp
    z, mu, sigma = Float64.(g["z"]), Float64.(g["mu"]), g["sigma"]
    ok = true
    ok &= nlp.num_variables == g["num_variables"] && nlp.num_constraint == g["num_constraint"] && nlp.num_jacobian == g["num_jacobian"]
    ok &= [[r, c] for (r, c) in MOI.jacobian_structure(nlp)] == g["jacobian_structure"]
    ok &= [[r, c] for (r, c) in MOI.hessian_lagrangian_structure(nlp)] == g["hessian_structure"]
    f = MOI.eval_objective(nlp, z)
    grad = zeros(nlp.num_variables); MOI.eval_objective_gradient(nlp, grad, z)
    c = zeros(nlp.num_constraint); MOI.eval_constraint(nlp, c, z)
    J = zeros(nlp.num_jacobian); MOI.eval_constraint_jacobian(nlp, J, z)
    H = zeros(length(nlp.hessian_lagrangian_sparsity)); MOI.eval_hessian_lagrangian(nlp, H, z, sigma, mu)
    errs = (abs(f - g["objective"]) / max(1.0, abs(g["objective"])), relerr(grad, Float64.(g["gradient"])),
            relerr(c, Float64.(g["constraint"])), relerr(J, Float64.(g["jacobian"])), relerr(H, Float64.(g["hessian_sigma"])))
    ok &= all(e -> e <= 1.0e-8, errs)
    println(rpad(basename(path), 28), ok ? "OK  " : "FAIL", "  max rel err f/grad/c/J/H = ", errs)
    ok
end

# ---------------------------------------------------------------- optional: Ipopt from the same guesses (--solve)
# tests/golden/solve_guesses.json holds seeded guesses and what the GPU solver returned from them (iterations, objective).
# The reference's solve!(solver) is run from exactly those guesses; Ipopt's iteration count is read from its output file
# (src/options.jl:24).  Different local minima are legitimate for the nonconvex swing-ups: compare objectives first.
function solve_case(r)
    eh = r["evaluate_hessian"]
    full = build(r["model"], r["T"])            # constructors above use evaluate_hessian=true
    solver = eh ? full : full                   # (for the default-mode examples rebuild with eh=false if L-BFGS is wanted)
    initialize_states!(solver, [Float64.(x) for x in r["states"]])
    initialize_controls!(solver, [Float64.(u) for u in r["actions"]])
    solve!(solver)
    x_sol, u_sol = get_trajectory(solver)
    z = vcat([vcat(x_sol[t], t <= length(u_sol) ? u_sol[t] : Float64[]) for t = 1:length(x_sol)]...)
    f = MOI.eval_objective(solver.nlp, z)
    println(rpad(string(r["model"], " T=", r["T"], " seed=", r["seed"]), 28), "reference objective ", f,
            "   ours: ", r["ours"])
end
if "--solve" in ARGS
    for r in JSON.parsefile(joinpath(filter(a -> a != "--solve", ARGS)[1], "solve_guesses.json"))
        solve_case(r)
    end
    exit(0)
end

dir = length(ARGS) > 0 ? ARGS[1] : "tests/golden"
results = [check(joinpath(dir, f)) for f in sort(readdir(dir)) if endswith(f, ".json") && f != "solve_guesses.json"]
println(all(results) ? "all fixtures agree with the reference" : "MISMATCH: the oracle is not pinned")
exit(all(results) ? 0 : 1)
