# emit_plugin.jl -- Julia side of the model hand-over: writes the Symbolics expressions the reference constructors trace
# (src/dynamics.jl:23-36, src/costs.jl:18-28, src/constraints.jl:27-41) into a "dto-dag-v1" JSON file that
# `python -m dto_amd.dagjson model.json` turns into the gfx950 model plugin (format: directtrajectoryoptimization.jl_amd/dagjson.py).
#
#   include("emit_plugin.jl")
#   emit_model("acrobot.json", dynamics_f, stage_cost_f, terminal_cost_f, constraint_fs, bounds, T;
#              num_state = 4, num_action = 1, evaluate_hessian = true)
#   run(`python -m dto_amd.dagjson acrobot.json`)                  # prints the plugin path for DtoSpec.model_library
#
# The closures are the SAME ones the reference scripts hand to Dynamics / Cost / Constraint
# (examples/acrobot/acrobot.jl:88-118): they are traced here exactly as the reference traces them (`@variables`, call the
# closure on symbolic arrays), only the result is written out instead of being `eval`'d into CPU code.
#
# NOTE: Julia / Symbolics.jl are not installed in the build environment of this repository: this file is reviewed, not
# executed, code.  The consumer of its output IS executed and tested (tests/test_dagjson.py: a file of this format produces the
# bit-identical plugin source of the traced model; n-ary sums / products and rational powers as Symbolics prints them).
using Symbolics
using SymbolicUtils

const _DTO_FUNCS = Dict{Any,String}(sin => "sin", cos => "cos", tan => "tan", exp => "exp", log => "log", sqrt => "sqrt",
                                    tanh => "tanh", atan => "atan", asin => "asin", acos => "acos", sinh => "sinh",
                                    cosh => "cosh", abs => "abs")

mutable struct _DagWriter
    nodes::Vector{String}               # JSON text of every node, in topological order
    index::Dict{Any,Int}                # expression -> node id (0-based)
    varid::Dict{Any,Tuple{String,Int}}  # symbolic variable -> (name, 0-based index)
end

_num(v) = isfinite(v) ? repr(Float64(v)) : error("non-finite constant in a model expression")

function _push!(w::_DagWriter, key, json::String)
    push!(w.nodes, json)
    w.index[key] = length(w.nodes) - 1
    return w.index[key]
end

# The reference pins Symbolics 0.1.29 - 0.1.32 (Project.toml:22; SymbolicUtils 0.11 - 0.13 underneath).  In that series the
# accessor of the expression inside a `Num` is `Symbolics.value` (`unwrap` was added in a later series and is the name used
# from Symbolics 1.x on); the tree interface is `SymbolicUtils.istree / operation / arguments` there (`iscall` replaces
# `istree` from SymbolicUtils 2.x on).  Both spellings are resolved once, at load time, so the walker runs on either.
const _unwrap = isdefined(Symbolics, :unwrap) ? Symbolics.unwrap : Symbolics.value
const _istree = isdefined(SymbolicUtils, :istree) ? SymbolicUtils.istree : SymbolicUtils.iscall

function _visit!(w::_DagWriter, ex)
    ex = _unwrap(ex)
    haskey(w.index, ex) && return w.index[ex]
    if ex isa Number
        return _push!(w, ex, "{\"op\":\"const\",\"value\":$(_num(ex))}")
    end
    if haskey(w.varid, ex)
        name, i = w.varid[ex]
        return _push!(w, ex, "{\"op\":\"var\",\"name\":\"$name\",\"index\":$i}")
    end
    _istree(ex) || error("unsupported leaf in a model expression: $ex")
    f = SymbolicUtils.operation(ex)
    args = [_visit!(w, a) for a in SymbolicUtils.arguments(ex)]
    list = join(args, ",")
    if f === (+)
        return _push!(w, ex, "{\"op\":\"add\",\"args\":[$list]}")
    elseif f === (*)
        return _push!(w, ex, "{\"op\":\"mul\",\"args\":[$list]}")
    elseif f === (-)
        return _push!(w, ex, length(args) == 1 ? "{\"op\":\"neg\",\"args\":[$list]}" : "{\"op\":\"sub\",\"args\":[$list]}")
    elseif f === (/)
        return _push!(w, ex, "{\"op\":\"div\",\"args\":[$list]}")
    elseif f === (^)
        return _push!(w, ex, "{\"op\":\"pow\",\"args\":[$list]}")
    elseif f === ifelse || nameof(f) === :ifelse        # IfElse.ifelse(cond, a, b), cond = (lhs < rhs) or (lhs <= rhs)
        c = SymbolicUtils.arguments(ex)[1]
        cf = SymbolicUtils.operation(_unwrap(c))
        cmp = cf === (<) ? "lt" : (cf === (<=) ? "le" : error("ifelse condition must be < or <="))
        l, r = [_visit!(w, a) for a in SymbolicUtils.arguments(_unwrap(c))]
        return _push!(w, ex, "{\"op\":\"ifelse\",\"cmp\":\"$cmp\",\"args\":[$l,$r,$(args[2]),$(args[3])]}")
    elseif haskey(_DTO_FUNCS, f)
        return _push!(w, ex, "{\"op\":\"call\",\"fn\":\"$(_DTO_FUNCS[f])\",\"args\":[$list]}")
    end
    error("unsupported operation in a model expression: $f")
end

"JSON text of one class: dims, the node list and the output node ids"
function _class_json(dims::Vector{Pair{String,Int}}, vars::Dict{String,Vector{Num}}, outputs; extra = "")
    w = _DagWriter(String[], Dict{Any,Int}(), Dict{Any,Tuple{String,Int}}())
    for (name, vs) in vars, (i, v) in enumerate(vs)
        w.varid[_unwrap(v)] = (name, i - 1)
    end
    outs = [_visit!(w, o) for o in outputs]
    head = join(["\"$k\":$v" for (k, v) in dims], ",")
    return "{$head$extra,\"nodes\":[$(join(w.nodes, ","))],\"outputs\":[$(join(outs, ","))]}"
end

function dynamics_class_json(f::Function, ny::Int, nx::Int, nu::Int; nw::Int = 0)
    @variables y[1:ny] x[1:nx] u[1:nu] w[1:nw]
    Y, X, U, W = collect(y), collect(x), collect(u), collect(w)
    out = f(Y, X, U, W)                                   # src/dynamics.jl:23-24
    _class_json(["num_next_state" => ny, "num_state" => nx, "num_action" => nu, "num_parameter" => nw],
                Dict("y" => Y, "x" => X, "u" => U, "w" => W), out)
end

function cost_class_json(f::Function, nx::Int, nu::Int; nw::Int = 0)
    @variables x[1:nx] u[1:nu] w[1:nw]
    X, U, W = collect(x), collect(u), collect(w)
    out = [f(X, U, W)]                                    # src/costs.jl:18-22
    _class_json(["num_state" => nx, "num_action" => nu, "num_parameter" => nw], Dict("x" => X, "u" => U, "w" => W), out)
end

function constraint_class_json(f::Function, nx::Int, nu::Int; nw::Int = 0, indices_inequality = Int[])
    @variables x[1:nx] u[1:nu] w[1:nw]
    X, U, W = collect(x), collect(u), collect(w)
    out = f(X, U, W)                                      # src/constraints.jl:27-28
    _class_json(["num_state" => nx, "num_action" => nu, "num_parameter" => nw], Dict("x" => X, "u" => U, "w" => W), out;
                extra = ",\"indices_inequality\":[$(join(indices_inequality, ","))]")
end

_lims(v) = "[" * join([isfinite(a) ? repr(Float64(a)) : "null" for a in v], ",") * "]"

"""
    emit_model(path; name, T, evaluate_hessian, dynamics, objective, constraints, bounds = nothing, parameters = nothing)

`dynamics`, `objective`, `constraints`: `(classes, stages)` -- the JSON texts of the distinct classes (the functions above)
and, per stage, the 0-based class id (`-1` = `Constraint()`): the time-invariant problems of the reference examples have one
dynamics class, two cost classes (stage, terminal) and one or two constraint classes.  `bounds`: per stage
`(state_lower, state_upper, action_lower, action_upper)` as in `Bound` (src/bounds.jl:8-14).
"""
function emit_model(path::String; name::String, T::Int, evaluate_hessian::Bool, dynamics, objective, constraints,
                    bounds = nothing, parameters = nothing)
    sect(cs) = "{\"classes\":[$(join(cs[1], ","))],\"stages\":[$(join(cs[2], ","))]}"
    parts = ["\"format\":\"dto-dag-v1\"", "\"name\":\"$name\"", "\"T\":$T", "\"evaluate_hessian\":$(evaluate_hessian)",
             "\"dynamics\":" * sect(dynamics), "\"objective\":" * sect(objective), "\"constraints\":" * sect(constraints)]
    if bounds !== nothing
        bs = ["{\"state_lower\":$(_lims(b[1])),\"state_upper\":$(_lims(b[2])),\"action_lower\":$(_lims(b[3])),\"action_upper\":$(_lims(b[4]))}" for b in bounds]
        push!(parts, "\"bounds\":[$(join(bs, ","))]")
    end
    if parameters !== nothing
        push!(parts, "\"parameters\":[" * join(["[" * join([repr(Float64(v)) for v in p], ",") * "]" for p in parameters], ",") * "]")
    end
    open(path, "w") do io
        write(io, "{" * join(parts, ",") * "}")
    end
    return path
end

# Example: the reference's acrobot (examples/acrobot/acrobot.jl:88-118), with `dynamics_f(y, x, u, w) = y - midpoint(x, y, u)`:
#   dyn = dynamics_class_json(dynamics_f, 4, 4, 1)
#   ct  = cost_class_json((x, u, w) -> 0.1 * dot(x[3:4], x[3:4]) + 0.1 * dot(u, u), 4, 1)
#   cT  = cost_class_json((x, u, w) -> 0.1 * dot(x[3:4], x[3:4]), 4, 0)
#   c1  = constraint_class_json((x, u, w) -> x - x1, 4, 1)
#   cN  = constraint_class_json((x, u, w) -> x - xT, 4, 0)
#   emit_model("acrobot.json"; name = "acrobot", T = T, evaluate_hessian = true,
#              dynamics = ([dyn], fill(0, T - 1)), objective = ([ct, cT], [fill(0, T - 1); 1]),
#              constraints = ([c1, cN], [0; fill(-1, T - 2); 1]))
