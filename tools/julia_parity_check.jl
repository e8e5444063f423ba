# Pin this repository's golden fixtures against the REAL reference (DirectTrajectoryOptimization.jl + Symbolics + MOI).
#
# Not executed in the build environment (no Julia there): this is the script a maintainer runs once Julia and the
# reference's dependencies are available, to close the one gap the parity report states everywhere ("Symbolics' structural
# rules on corner cases and Ipopt's iterates cannot be checked here", SURVEY.md 8c / DESIGN.md section 6).
#
#     julia --project=/path/to/DirectTrajectoryOptimization.jl tools/julia_parity_check.jl tests/golden
#     julia --project=/path/to/DirectTrajectoryOptimization.jl tools/julia_parity_check.jl tests/golden --solve   # + Ipopt
#
# For every fixture it rebuilds the same problem with the reference's own constructors (models as in examples/*.jl and
# test/*.jl), evaluates the five MOI methods of src/moi.jl at the fixture's point (z, mu, sigma) and compares
#   * totals and both structures bit for bit            (src/data.jl:61-220)
#   * f, grad f, c, J, H                                to 1e-8 relative (the tolerance of BASELINE.json's north star).
# The fixtures were produced by oracle/ (sympy, 30-digit mpmath) -- agreement here pins the oracle, and with it every
# parity test of the HIP path, to the reference itself.
using DirectTrajectoryOptimization
using LinearAlgebra, JSON
const DTO = DirectTrajectoryOptimization
const MOI = DTO.MOI

# ---------------------------------------------------------------- models (examples/*/*.jl, test/dynamics.jl)
function pendulum(x, u, w)
    mass, length_com, gravity, damping = 1.0, 0.5, 9.81, 0.1
    [x[2], (u[1] / ((mass * length_com * length_com)) - gravity * sin(x[1]) / length_com - damping * x[2] / (mass * length_com * length_com))]
end
function cartpole(x, u, w)
    mc, mp, l, g = 1.0, 0.2, 0.5, 9.81
    q, qd = x[1:2], x[3:4]
    s, c = sin(q[2]), cos(q[2])
    H = [mc+mp mp*l*c; mp*l*c mp*l^2]
    Cm = [0.0 -mp*qd[2]*l*s; 0.0 0.0]
    G = [0.0, mp * g * l * s]
    B = [1.0, 0.0]
    qdd = -H \ (Cm * qd + G - B * u[1])
    [qd; qdd]
end
function acrobot(x, u, w)
    mass1, inertia1, length1, lengthcom1 = 1.0, 0.33, 1.0, 0.5
    mass2, inertia2, length2, lengthcom2 = 1.0, 0.33, 1.0, 0.5
    gravity, friction1, friction2 = 9.81, 0.1, 0.1
    function M(x, w)
        a = inertia1 + inertia2 + mass2 * length1^2 + 2.0 * mass2 * length1 * lengthcom2 * cos(x[2])
        b = inertia2 + mass2 * length1 * lengthcom2 * cos(x[2])
        c = inertia2
        [a b; b c]
    end
    function Mass_inv(x, w)
        m = M(x, w)
        1.0 / (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * [m[2, 2] -m[1, 2]; -m[2, 1] m[1, 1]]
    end
    function tau(x, w)
        a = (-1.0 * mass1 * gravity * lengthcom1 * sin(x[1]) - mass2 * gravity * (length1 * sin(x[1]) + lengthcom2 * sin(x[1] + x[2])))
        b = -1.0 * mass2 * gravity * lengthcom2 * sin(x[1] + x[2])
        [a, b]
    end
    function Cor(x, w)
        a = -2.0 * mass2 * length1 * lengthcom2 * sin(x[2]) * x[4]
        b = -1.0 * mass2 * length1 * lengthcom2 * sin(x[2]) * x[4]
        c = mass2 * length1 * lengthcom2 * sin(x[2]) * x[3]
        [a b; c 0.0]
    end
    Bm(x, w) = [0.0, 1.0]
    q, v = view(x, 1:2), view(x, 3:4)
    qdd = Mass_inv(q, w) * (-1.0 * Cor(x, w) * v + tau(q, w) + Bm(q, w) * u[1] - [friction1 0.0; 0.0 friction2] * v)
    [x[3], x[4], qdd[1], qdd[2]]
end
car(x, u, w) = [u[1] * cos(x[3]), u[1] * sin(x[3]), u[2]]
midpoint(f, h) = (y, x, u, w) -> y - (x + h * f(0.5 * (x + y), u, w))
function rk3_implicit(f, h)
    function explicit(x, u, w)
        k1 = f(x, u, w) * h
        k2 = f(x + 0.5 * k1, u, w) * h
        k3 = f(x - k1 + 2.0 * k2, u, w) * h
        x + (k1 + 4.0 * k2 + k3) / 6.0
    end
    (y, x, u, w) -> y - explicit(x, u, w)
end

# ---------------------------------------------------------------- problems of the fixtures (oracle/sympy_models.py:build)
function build(model, T)
    eh = true
    if model == "pendulum"
        n, m = 2, 1; x1, xT = [0.0, 0.0], [pi, 0.0]
        d = Dynamics(midpoint(pendulum, 0.05), n, n, m, evaluate_hessian=eh)
        ct = Cost((x, u, w) -> 0.1 * dot(x[1:2], x[1:2]) + 0.1 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.1 * dot(x[1:2], x[1:2]), n, 0, evaluate_hessian=eh)
        cons = [Constraint((x, u, w) -> x - x1, n, m, evaluate_hessian=eh), [Constraint() for t = 2:T-1]...,
                Constraint((x, u, w) -> x - xT, n, 0, evaluate_hessian=eh)]
        bnds = [[Bound(n, m) for t = 1:T-1]..., Bound(n, 0)]
    elseif model == "cartpole"
        n, m = 4, 1; x1, xT = zeros(4), [0.0, pi, 0.0, 0.0]
        d = Dynamics(rk3_implicit(cartpole, 0.05), n, n, m, evaluate_hessian=eh)
        ct = Cost((x, u, w) -> 0.5 * 1.0e-2 * dot(x - xT, x - xT) + 0.5 * 1.0e-1 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.5 * 1.0e2 * dot(x - xT, x - xT), n, 0, evaluate_hessian=eh)
        cons = [Constraint((x, u, w) -> x - x1, n, m, evaluate_hessian=eh), [Constraint() for t = 2:T-1]...,
                Constraint((x, u, w) -> x - xT, n, 0, evaluate_hessian=eh)]
        bnds = [[Bound(n, m, action_lower=[-3.0], action_upper=[3.0]) for t = 1:T-1]..., Bound(n, 0)]
    elseif model == "acrobot" || model == "acrobot_bounds"
        n, m = 4, 1; x1 = zeros(4)
        d = Dynamics(midpoint(acrobot, 0.05), n, n, m, evaluate_hessian=eh)
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
        d = Dynamics(midpoint(car, 0.1), n, n, m, evaluate_hessian=eh)
        ct = Cost((x, u, w) -> 0.0 * dot(x - xT, x - xT) + 1.0 * dot(u, u), n, m, evaluate_hessian=eh)
        cT = Cost((x, u, w) -> 0.0 * dot(x - xT, x - xT), n, 0, evaluate_hessian=eh)
        obs = (x, u, w) -> [0.1^2 - dot(x[1:2] - [0.5, 0.5], x[1:2] - [0.5, 0.5])]
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
