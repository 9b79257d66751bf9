"""PyTorch log density -> HIP source for ``CTarget.from_source(form="elementwise")`` (``TorchModel(compile=True)``).

The provider north_star lists first is "PyTorch-ROCm autograd" (GradModel.log_density_gradient, bayes_kit/typing.py:25-27).
Autograd costs eight passes over the state per gradient; the engine's own compiled targets cost one.  For the densities
that are ELEMENTWISE PLUS A SUM OVER THE COORDINATES -- log p(theta) = sum_d e(theta_d, params_d) (+ const), possibly several
such sums combined linearly -- this module reads the user's function once with ``torch.fx``, differentiates the per-coordinate
expression symbolically (forward mode, one variable) and emits the ``bk_term`` source (value and derivative) that
``CTarget.from_source`` compiles with hipcc.  This is code generation for ONE backend (HIP for gfx950), not a dispatch layer:
anything outside the supported set raises ``Unsupported`` naming the node, and ``TorchModel`` then keeps autograd.

Supported: + - * / neg, ``**`` with a constant exponent or a constant base, exp, log, log1p, expm1, sigmoid, logsigmoid,
softplus (beta = 1), tanh, sinh, cosh, atan, erf, sqrt, rsqrt, reciprocal, square, abs, sin, cos; tensor constants that broadcast
along the coordinate axis (shape (D,) / (1, D) for the (C, D) layout, (D, 1) for the (D, C) layout) or are scalars; one ``sum`` /
``mean`` over the coordinate axis per additive part; scalar multiples and sums of such parts after the sum; ``.shape`` /
``.size()`` of the coordinate axis; ``torch.broadcast_tensors``.  That set covers ``torch.distributions`` log_prob bodies such as
Normal, StudentT, Laplace, Cauchy, Logistic with constant parameters (argument validation is data-dependent control flow, so it
is switched off while the function is being read).
"""
from __future__ import annotations

import math
import operator

import torch


class Unsupported(Exception):
    """The function is outside the traceable set; the message names the node."""


# ---- expression DAG (per coordinate): ('th',) | ('p', k) | ('c', value) | (op, *args) ---------------------------------
class _E:
    __slots__ = ("op", "args", "dep")

    def __init__(self, op, args=(), dep=False):
        self.op, self.args, self.dep = op, tuple(args), dep


def _const(v):
    return _E("c", (float(v),), False)


class _Chain:
    """A per-chain value: sum over d of `elem` + `const`."""

    def __init__(self, elem, const=0.0):
        self.elem, self.const = elem, float(const)


_UNARY = {"neg", "exp", "log", "log1p", "expm1", "sigmoid", "logsigmoid", "softplus", "tanh", "sqrt", "square", "abs", "sin",
          "cos", "sinh", "cosh", "atan", "erf"}
_HOST_UNARY = {"exp": math.exp, "log": math.log, "log1p": math.log1p, "expm1": math.expm1, "tanh": math.tanh, "sqrt": math.sqrt,
               "square": lambda t: t * t, "abs": abs, "sin": math.sin, "cos": math.cos, "sinh": math.sinh, "cosh": math.cosh,
               "atan": math.atan, "erf": math.erf, "neg": lambda t: -t}
TWO_OVER_SQRT_PI = 2.0 / math.sqrt(math.pi)
_BINARY = {"add", "sub", "mul", "div", "pow", "gt", "ge", "lt", "le", "maximum", "minimum"}
_COMPARE = {"gt": ">", "ge": ">=", "lt": "<", "le": "<="}  # values 1.0 / 0.0; only ever the condition of a `where`


def _function_table():
    F = torch.nn.functional
    t = {
        operator.add: "add", operator.sub: "sub", operator.mul: "mul", operator.truediv: "div", operator.neg: "neg",
        operator.pow: "pow", torch.add: "add", torch.sub: "sub", torch.subtract: "sub", torch.mul: "mul",
        torch.multiply: "mul", torch.div: "div", torch.true_divide: "div", torch.divide: "div", torch.neg: "neg",
        torch.negative: "neg", torch.exp: "exp", torch.log: "log", torch.log1p: "log1p", torch.expm1: "expm1",
        torch.sigmoid: "sigmoid", torch.special.expit: "sigmoid", F.sigmoid: "sigmoid", F.logsigmoid: "logsigmoid",
        F.softplus: "softplus", torch.tanh: "tanh", F.tanh: "tanh", torch.sqrt: "sqrt", torch.square: "square",
        torch.pow: "pow", torch.abs: "abs", torch.sin: "sin", torch.cos: "cos", torch.sum: "sum", torch.mean: "mean",
        torch.sinh: "sinh", torch.cosh: "cosh", torch.atan: "atan", torch.arctan: "atan", torch.erf: "erf",
        torch.special.erf: "erf", torch.reciprocal: "reciprocal", torch.rsqrt: "rsqrt", torch.special.log1p: "log1p",
        torch.special.expm1: "expm1", torch.broadcast_tensors: "broadcast_tensors", getattr: "getattr",
        operator.getitem: "getitem", torch.absolute: "abs", operator.gt: "gt", operator.ge: "ge", operator.lt: "lt",
        operator.le: "le", torch.gt: "gt", torch.ge: "ge", torch.lt: "lt", torch.le: "le", torch.where: "where",
        torch.maximum: "maximum", torch.minimum: "minimum", torch.relu: "relu", F.relu: "relu", torch.clamp: "clamp",
        torch.clip: "clamp",
    }
    return t


class no_distribution_validation:
    """torch.distributions validates its arguments with data-dependent control flow (``if not valid.all(): raise``), which no
    tracer can read; validation is off while the function is traced and restored afterwards."""

    def __enter__(self):
        from torch.distributions import Distribution

        self.old = Distribution._validate_args
        Distribution.set_default_validate_args(False)

    def __exit__(self, *exc):
        from torch.distributions import Distribution

        Distribution.set_default_validate_args(self.old)
        return False


_METHODS = {"add": "add", "sub": "sub", "mul": "mul", "div": "div", "true_divide": "div", "neg": "neg", "exp": "exp",
            "log": "log", "log1p": "log1p", "expm1": "expm1", "sigmoid": "sigmoid", "tanh": "tanh", "sqrt": "sqrt",
            "square": "square", "pow": "pow", "abs": "abs", "sin": "sin", "cos": "cos", "sum": "sum", "mean": "mean", "double": "id",
            "float": None, "contiguous": "id", "clone": "id", "sinh": "sinh", "cosh": "cosh", "atan": "atan", "arctan": "atan",
            "erf": "erf", "reciprocal": "reciprocal", "rsqrt": "rsqrt", "size": "size", "absolute": "abs", "gt": "gt", "ge": "ge",
            "lt": "lt", "le": "le", "where": "where_method", "maximum": "maximum", "minimum": "minimum", "relu": "relu",
            "clamp": "clamp", "clip": "clamp", "clamp_min": "clamp_min", "clamp_max": "clamp_max"}


def expand_rewrite(rw, apply):
    """Evaluate a (possibly nested) rewrite (operation, operands) bottom-up with apply(operation, operands)."""
    def ex(a):
        return apply(a[0], [ex(x) for x in a[1]]) if isinstance(a, tuple) else a
    return ex(rw)


def piecewise_rewrite(name, args, kwargs, where):
    """relu / clamp / x.where(...) as (operation, operands) over `where` and comparisons; None for any other name.

    PyTorch's conventions, which a rewrite to maximum / minimum does not have (their tie rule splits the derivative 0.5 /
    0.5): relu'(0) = 0 and clamp' = 1 AT a bound (autograd: `x > 0`, `lo <= x <= hi`), and NaN goes through both (every
    comparison with NaN is false, so the `where`s below hand x itself on)."""
    if name == "where_method":  # x.where(cond, other) == torch.where(cond, x, other)
        if len(args) != 3:
            raise Unsupported(f"{where}: where with {len(args)} operands")
        return "where", [args[1], args[0], args[2]]
    if name == "relu":
        return "where", [("le", [args[0], 0.0]), 0.0, args[0]]
    if name in ("clamp", "clamp_min", "clamp_max"):
        lo = kwargs.get("min", args[1] if len(args) > 1 and name != "clamp_max" else None)
        hi = kwargs.get("max", args[1] if len(args) > 1 and name == "clamp_max" else (args[2] if len(args) > 2 else None))
        for b in (lo, hi):
            if b is not None and not isinstance(b, (int, float)):
                raise Unsupported(f"{where}: clamp bounds must be Python numbers")
        if lo is None and hi is None:
            raise Unsupported(f"{where}: clamp without bounds")
        x = args[0]
        if lo is not None and hi is not None:   # (torch.clamp with lo > hi returns hi everywhere: the outer test decides)
            return "where", [("gt", [x, float(hi)]), float(hi), ("where", [("lt", [x, float(lo)]), float(lo), x])] \
                if float(lo) <= float(hi) else ("where", [("le", [x, x]), float(hi), x])
        return ("where", [("lt", [x, float(lo)]), float(lo), x]) if lo is not None else ("where", [("gt", [x, float(hi)]), float(hi), x])
    return None


class _Tracer:
    def __init__(self, D, layout):
        self.D, self.dc = int(D), layout == "dc"
        self.rows = []  # packed per-coordinate constants, each (D,) float64 on the host

    # -- classification of a traced value ------------------------------------------------------------------------------
    def constant(self, t, where):
        if not isinstance(t, torch.Tensor):
            raise Unsupported(f"{where}: attribute of type {type(t).__name__}")
        if t.numel() == 1:
            return _const(float(t.reshape(()).item()))
        ok = tuple(t.shape) in (((self.D, 1),) if self.dc else ((self.D,), (1, self.D)))
        if not ok:
            want = f"({self.D}, 1)" if self.dc else f"({self.D},) or (1, {self.D})"
            raise Unsupported(f"{where}: a tensor constant of shape {tuple(t.shape)} does not broadcast along the coordinate "
                              f"axis (expected a scalar or shape {want})")
        row = t.detach().reshape(self.D).to(dtype=torch.float64, device="cpu")
        for k, r in enumerate(self.rows):
            if torch.equal(r, row):
                return _E("p", (k,), False)
        self.rows.append(row)
        return _E("p", (len(self.rows) - 1,), False)

    def as_elem(self, v, where):
        if isinstance(v, _E):
            return v
        if isinstance(v, bool):
            raise Unsupported(f"{where}: boolean operand")
        if isinstance(v, (int, float)):
            return _const(v)
        if isinstance(v, torch.Tensor):
            return self.constant(v, where)
        if isinstance(v, _Chain):
            raise Unsupported(f"{where}: a per-chain value (the result of a sum) used inside an elementwise expression")
        raise Unsupported(f"{where}: operand of type {type(v).__name__}")

    # -- elementwise algebra with light constant folding -----------------------------------------------------------------
    def unary(self, op, a):
        if a.op == "c":
            try:
                f = _HOST_UNARY.get(op)
                if f is not None:
                    return _const(f(a.args[0]))
            except (ValueError, OverflowError):
                pass
        return _E(op, (a,), a.dep)

    def binary(self, op, a, b):
        if a.op == "c" and b.op == "c" and op in ("add", "sub", "mul", "div", "pow"):
            x, y = a.args[0], b.args[0]
            try:
                return _const({"add": x + y, "sub": x - y, "mul": x * y, "div": x / y if y != 0 else float("nan"),
                               "pow": x ** y}[op])
            except (OverflowError, ZeroDivisionError, ValueError):
                pass
        if op == "pow":
            if b.dep and a.dep:
                raise Unsupported("pow with both base and exponent depending on theta")
            if not b.dep and b.op == "c":
                if b.args[0] == 2.0:
                    return self.unary("square", a)
                if b.args[0] == 1.0:
                    return a
                if b.args[0] == 0.5:
                    return self.unary("sqrt", a)
        return _E(op, (a, b), a.dep or b.dep)

    # -- the graph walk ------------------------------------------------------------------------------------------------
    def sum_axis_ok(self, dim):
        if isinstance(dim, (list, tuple)):
            if len(dim) != 1:
                return False
            dim = dim[0]
        return dim in ((0, -2) if self.dc else (1, -1))

    def run(self, fn):
        import torch.fx as fx

        try:
            with no_distribution_validation():
                gm = fx.symbolic_trace(fn)
        except Exception as e:  # data-dependent control flow, in-place tricks, ...
            raise Unsupported(f"torch.fx could not trace the function: {type(e).__name__}: {e}") from e
        table = _function_table()
        env = {}
        out = None
        n_inputs = 0
        for node in gm.graph.nodes:
            where = f"node `{node.format_node()}`"

            def val(a):
                if isinstance(a, (tuple, list)):
                    return type(a)(val(x) for x in a)
                return env[a] if isinstance(a, fx.Node) else a

            if node.op == "placeholder":
                n_inputs += 1
                if n_inputs > 1:
                    raise Unsupported("the function takes more than one argument")
                env[node] = _E("th", (), True)
            elif node.op == "get_attr":
                obj = gm
                for part in node.target.split("."):
                    obj = getattr(obj, part)
                env[node] = obj  # (classified where it is used: the operation is what an error should name)
            elif node.op in ("call_function", "call_method"):
                name = table.get(node.target) if node.op == "call_function" else _METHODS.get(node.target)
                if name is None:
                    raise Unsupported(f"{where}: unsupported operation {getattr(node.target, '__name__', node.target)!r}")
                args = [val(a) for a in node.args]
                kwargs = {k: val(v) for k, v in node.kwargs.items()}
                env[node] = self.apply(name, args, kwargs, where)
            elif node.op == "output":
                out = val(node.args[0])
            else:
                raise Unsupported(f"{where}: unsupported node kind {node.op}")
        if not isinstance(out, _Chain):
            raise Unsupported("the function does not end in a per-chain value: it must sum its elementwise expression over "
                              f"the coordinate axis (dim={'0' if self.dc else '1'})")
        if not out.elem.dep:
            raise Unsupported("the log density does not depend on theta")
        return out

    def shape_of(self, v, where):
        """The shape of a traced value with the chain count unknown (None: using it in arithmetic is an error)."""
        if isinstance(v, _E) and v.dep:
            return (self.D, None) if self.dc else (None, self.D)
        if isinstance(v, _Chain):
            return (None,)
        if isinstance(v, torch.Tensor):
            return tuple(v.shape)
        raise Unsupported(f"{where}: shape of a value the tracer does not follow")

    def apply(self, name, args, kwargs, where):
        if name == "id":
            return args[0]
        if name == "broadcast_tensors":  # (torch.distributions' broadcast_all): the operands themselves; every operation
            return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list)) else tuple(args)  # broadcasts anyway
        if name == "getitem":
            src, idx = args
            if isinstance(src, (tuple, list)) and isinstance(idx, int):
                return src[idx]
            raise Unsupported(f"{where}: unsupported operation 'getitem' (indexing theta couples coordinates: not elementwise)")
        if name == "getattr":
            if args[1] == "shape":
                return self.shape_of(args[0], where)
            raise Unsupported(f"{where}: attribute {args[1]!r}")
        if name == "size":
            shp = self.shape_of(args[0], where)
            return shp if len(args) == 1 else shp[args[1]]
        if name in ("reciprocal", "rsqrt"):
            if len(args) != 1 or kwargs:
                raise Unsupported(f"{where}: {name} with extra arguments")
            a = self.as_elem(args[0], where)
            return self.binary("div", _const(1.0), a if name == "reciprocal" else self.unary("sqrt", a))
        if name == "mean":
            s = self.apply("sum", args, kwargs, where)
            return self.chain_op("div", [s, float(self.D)], where)
        if name == "sum":
            x = args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim", kwargs.get("axis"))
            if kwargs.get("keepdim", False) or (len(args) > 2 and args[2]):
                raise Unsupported(f"{where}: sum(keepdim=True)")
            if kwargs.get("dtype") not in (None, torch.float64):
                raise Unsupported(f"{where}: sum(dtype=...)")
            if not isinstance(x, _E) or not x.dep:
                raise Unsupported(f"{where}: sum of something that is not an elementwise expression of theta")
            if dim is None or not self.sum_axis_ok(dim):
                raise Unsupported(f"{where}: the sum must run over the coordinate axis only "
                                  f"(dim={'0' if self.dc else '1'}), got dim={dim!r}")
            return _Chain(x, 0.0)
        if name == "relu" and kwargs.get("inplace") is False:
            kwargs = {}
        if kwargs and not (name == "softplus" and set(kwargs) <= {"beta", "threshold"}) \
                and not (name.startswith("clamp") and set(kwargs) <= {"min", "max"}):
            raise Unsupported(f"{where}: keyword arguments {sorted(kwargs)}")
        if name == "softplus":
            beta = kwargs.get("beta", args[1] if len(args) > 1 else 1.0)
            thr = kwargs.get("threshold", args[2] if len(args) > 2 else 20.0)
            if float(beta) != 1.0 or float(thr) != 20.0:
                raise Unsupported(f"{where}: softplus with beta / threshold other than the defaults")
            args = args[:1]
        rw = piecewise_rewrite(name, args, kwargs, where)
        if rw is not None:
            return expand_rewrite(rw, lambda n, a: self.apply(n, a, {}, where))
        chains = [a for a in args if isinstance(a, _Chain)]
        if chains:
            return self.chain_op(name, args, where)
        if name == "where":
            if len(args) != 3:
                raise Unsupported(f"{where}: where with {len(args)} operands")
            c, a, b = (self.as_elem(v, where) for v in args)
            if c.op not in _COMPARE:
                raise Unsupported(f"{where}: the condition of `where` must be a comparison (> >= < <=)")
            return _E("where", (c, a, b), c.dep or a.dep or b.dep)
        if name in _UNARY:
            if len(args) != 1:
                raise Unsupported(f"{where}: {name} with {len(args)} operands")
            return self.unary(name, self.as_elem(args[0], where))
        if name in _BINARY:
            if len(args) != 2:
                raise Unsupported(f"{where}: {name} with {len(args)} operands (alpha= / rounding_mode= are not supported)")
            return self.binary(name, self.as_elem(args[0], where), self.as_elem(args[1], where))
        raise Unsupported(f"{where}: unsupported operation {name}")

    def scalar(self, v, where):
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            return float(v)
        if isinstance(v, torch.Tensor) and v.numel() == 1:
            return float(v.reshape(()).item())
        if isinstance(v, _E) and v.op == "c":
            return v.args[0]
        raise Unsupported(f"{where}: after the sum only scalar multiples and sums of per-chain values are supported")

    def chain_op(self, name, args, where):
        """Linear algebra on per-chain values: each stays `sum_d elem + const`."""
        if name == "neg":
            c = args[0]
            return _Chain(self.unary("neg", c.elem), -c.const)
        if name in ("add", "sub") and len(args) == 2:
            a, b = args
            sign = 1.0 if name == "add" else -1.0
            if isinstance(a, _Chain) and isinstance(b, _Chain):
                return _Chain(self.binary(name, a.elem, b.elem), a.const + sign * b.const)
            if isinstance(a, _Chain):
                return _Chain(a.elem, a.const + sign * self.scalar(b, where))
            s = self.scalar(a, where)
            return _Chain(b.elem if name == "add" else self.unary("neg", b.elem), s + sign * b.const)
        if name == "mul" and len(args) == 2:
            a, b = args
            if isinstance(a, _Chain) and isinstance(b, _Chain):
                raise Unsupported(f"{where}: product of two per-chain values")
            c, s = (a, self.scalar(b, where)) if isinstance(a, _Chain) else (b, self.scalar(a, where))
            return _Chain(self.binary("mul", _const(s), c.elem), s * c.const)
        if name == "div" and len(args) == 2 and isinstance(args[0], _Chain) and not isinstance(args[1], _Chain):
            s = self.scalar(args[1], where)
            return _Chain(self.binary("div", args[0].elem, _const(s)), args[0].const / s)
        raise Unsupported(f"{where}: `{name}` of a per-chain value (only scalar multiples and sums are supported after the sum)")


# ---- code generation: forward-mode value + derivative in th ----------------------------------------------------------------
def _lit(x):
    if math.isnan(x):
        return "(0.0 / 0.0)"
    if math.isinf(x):
        return "INFINITY" if x > 0 else "(-INFINITY)"
    r = repr(float(x))
    return f"({r})" if r.startswith("-") else r


class _Emit:
    def __init__(self, D):
        self.D = D
        self.lines = []
        self.memo = {}
        self.n = 0

    def tmp(self, expr):
        name = f"t{self.n}"
        self.n += 1
        self.lines.append(f"  const double {name} = {expr};")
        return name

    def gen(self, e):
        """(value expression, derivative expression or None) of node e; shared nodes are emitted once."""
        key = id(e)
        if key in self.memo:
            return self.memo[key]
        r = self._gen(e)
        self.memo[key] = r
        return r

    def _gen(self, e):
        op = e.op
        if op == "th":
            return "th", "1.0"
        if op == "c":
            return _lit(e.args[0]), None
        if op == "p":
            return self.tmp(f"P[{e.args[0] * self.D} + d]"), None
        if op in _UNARY:
            a, da = self.gen(e.args[0])
            return self.unary(op, a, da)
        if op == "where":
            (c, _), (a, da), (b, db) = (self.gen(x) for x in e.args)
            v = self.tmp(f"({c} != 0.0) ? {a} : {b}")
            if da is None and db is None:
                return v, None
            return v, self.tmp(f"({c} != 0.0) ? {da or '0.0'} : {db or '0.0'}")
        a, da = self.gen(e.args[0])
        b, db = self.gen(e.args[1])
        return self.binary(op, a, da, b, db, e)

    def mul(self, x, dx):  # x * dx with the trivial cases folded
        if dx is None:
            return None
        if dx == "1.0":
            return x
        return self.tmp(f"{x} * {dx}")

    def unary(self, op, a, da):
        T = self.tmp
        if op == "neg":
            return T(f"-{a}"), (None if da is None else T(f"-{da}"))
        if op == "exp":
            v = T(f"bk_exp({a})")   # (the library's exp: the same double as the built-in funnel's, on host and device)
            return v, self.mul(v, da)
        if op == "log":
            return T(f"log({a})"), (None if da is None else T(f"{da} / {a}"))
        if op == "log1p":
            return T(f"log1p({a})"), (None if da is None else T(f"{da} / (1.0 + {a})"))
        if op == "expm1":
            v = T(f"expm1({a})")
            return v, (None if da is None else self.mul(T(f"{v} + 1.0"), da))
        if op == "sigmoid":
            v = T(f"1.0 / (1.0 + exp(-{a}))")
            return v, (None if da is None else self.mul(T(f"{v} * (1.0 - {v})"), da))
        if op == "logsigmoid":  # min(a, 0) - log1p(exp(-|a|)); derivative sigmoid(-a)
            v = T(f"fmin({a}, 0.0) - log1p(exp(-fabs({a})))")
            return v, (None if da is None else self.mul(T(f"1.0 / (1.0 + exp({a}))"), da))
        if op == "softplus":  # torch: a > 20 ? a : log1p(exp(a)); derivative sigmoid(a)
            v = T(f"({a} > 20.0) ? {a} : log1p(exp({a}))")
            return v, (None if da is None else self.mul(T(f"1.0 / (1.0 + exp(-{a}))"), da))
        if op == "tanh":
            v = T(f"tanh({a})")
            return v, (None if da is None else self.mul(T(f"1.0 - {v} * {v}"), da))
        if op == "sqrt":
            v = T(f"sqrt({a})")
            return v, (None if da is None else T(f"{da} / (2.0 * {v})"))
        if op == "square":
            return T(f"{a} * {a}"), (None if da is None else self.mul(T(f"2.0 * {a}"), da))
        if op == "abs":
            return T(f"fabs({a})"), (None if da is None else self.mul(T(f"(double)(({a} > 0.0) - ({a} < 0.0))"), da))
        if op == "sin":
            return T(f"sin({a})"), (None if da is None else self.mul(T(f"cos({a})"), da))
        if op == "cos":
            return T(f"cos({a})"), (None if da is None else self.mul(T(f"-sin({a})"), da))
        if op == "sinh":
            return T(f"sinh({a})"), (None if da is None else self.mul(T(f"cosh({a})"), da))
        if op == "cosh":
            return T(f"cosh({a})"), (None if da is None else self.mul(T(f"sinh({a})"), da))
        if op == "atan":
            return T(f"atan({a})"), (None if da is None else T(f"{da} / (1.0 + {a} * {a})"))
        if op == "erf":
            return T(f"erf({a})"), (None if da is None else self.mul(T(f"{_lit(TWO_OVER_SQRT_PI)} * exp(-({a} * {a}))"), da))
        raise AssertionError(op)

    def binary(self, op, a, da, b, db, e):
        T = self.tmp
        if op in ("add", "sub"):
            s = "+" if op == "add" else "-"
            v = T(f"{a} {s} {b}")
            if da is None and db is None:
                return v, None
            if db is None:
                return v, da
            if da is None:
                return v, (db if op == "add" else T(f"-{db}"))
            return v, T(f"{da} {s} {db}")
        if op == "mul":
            v = T(f"{a} * {b}")
            if da is None and db is None:
                return v, None
            if db is None:
                return v, self.mul(b, da)
            if da is None:
                return v, self.mul(a, db)
            return v, T(f"{self.mul(b, da)} + {self.mul(a, db)}")
        if op == "div":
            v = T(f"{a} / {b}")
            if da is None and db is None:
                return v, None
            if db is None:
                return v, T(f"{da} / {b}")
            if da is None:
                return v, T(f"-({v} * {db}) / {b}")
            return v, T(f"({da} - {v} * {db}) / {b}")
        if op in _COMPARE:
            return T(f"(double)({a} {_COMPARE[op]} {b})"), None
        if op in ("maximum", "minimum"):  # torch: the derivative goes to the selected operand (ties: split evenly)
            cmp = ">" if op == "maximum" else "<"
            v = T(f"f{'max' if op == 'maximum' else 'min'}({a}, {b})")
            if da is None and db is None:
                return v, None
            da, db = da or "0.0", db or "0.0"
            return v, T(f"({a} {cmp} {b}) ? {da} : (({a} == {b}) ? 0.5 * ({da} + {db}) : {db})")
        if op == "pow":
            v = T(f"pow({a}, {b})")
            if da is None and db is None:
                return v, None
            if db is None:  # d/dth a^n = n a^(n-1) da
                return v, self.mul(T(f"{b} * pow({a}, {b} - 1.0)"), da)
            return v, self.mul(T(f"{v} * log({a})"), db)  # c^b: c^b log(c) db
        raise AssertionError(op)


def term_source(fn, dims: int, layout: str = "cd"):
    """(HIP source defining bk_term, packed params tensor [n_rows * D] on the host or None, description dict).
    Raises Unsupported."""
    D = int(dims)
    tr = _Tracer(D, layout)
    chain = tr.run(fn)
    em = _Emit(D)
    v, dv = em.gen(chain.elem)
    if dv is None:
        raise Unsupported("the log density does not depend on theta")
    body = "\n".join(em.lines)
    const = chain.const
    term = v if const == 0.0 else f"{v} + ((d == 0) ? {_lit(const)} : 0.0)"
    src = ("// generated by bayes_kit_amd.trace from a PyTorch log density (torch.fx graph, forward-mode derivative in theta_d)\n"
           "__device__ __forceinline__ void bk_term(double th, i64 d, const double* P, double& term, double& grad) {\n"
           f"{body}\n  term = {term};\n  grad = {dv};\n}}\n")
    params = torch.cat(tr.rows) if tr.rows else None
    return src, params, {"param_rows": len(tr.rows), "temporaries": em.n, "constant": const}
