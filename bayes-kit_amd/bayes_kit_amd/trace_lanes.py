"""PyTorch log density -> HIP source for ``CTarget.from_source(form="lanes")`` (``TorchModel(compile=True)``, second try).

``trace.py`` handles separable densities.  This module handles the next shape up -- HIERARCHICAL densities:

    log p(theta) = F( h_0 .. h_{H-1},  S_0 .. S_{K-1} ),      S_k = sum over rows d >= H of e_k(theta_d, h, params_d)

i.e. a few leading "head" coordinates taken by integer index (``Th[:, 0]``), the remaining rows taken as slices (``Th[:, H:]``, or
several groups ``Th[:, a:b]`` -- each sum then runs over its own group; ``Th`` itself when there are no heads), elementwise
expressions of the rows that may use head-derived per-chain values broadcast with ``[:, None]``, sums (means) of those over the
row axis, and any scalar expression of heads and sums at the end.  Neal's funnel, hierarchical normal / logistic models with shared
location and scale -- written with arithmetic or with ``torch.distributions`` log_prob calls -- are of this shape.

The function is read once with ``torch.fx`` into a hash-consed expression DAG, differentiated SYMBOLICALLY

    d log p / d theta_d = sum_k F_{S_k} * d e_k / d theta_d
    d log p / d h_i     = F_{h_i} + sum_k F_{S_k} * T_{k,i},      T_{k,i} = sum_d d e_k / d h_i     (extra sums, only where nonzero)

and emitted as the ``bk_lanes_density`` source of ``csrc/bk_lanes.hpp``'s lane context (``c.head``, ``c.sum``, ``c.grad_head``,
``c.grad``): per-chain subexpressions are hoisted out of the row lambdas.  Code generation for ONE backend; anything outside the
shape raises ``trace.Unsupported`` naming the node.  Only the reference's (C, D) layout.
"""
from __future__ import annotations

import torch

from .trace import (Unsupported, _function_table, _METHODS, _UNARY, _BINARY, _COMPARE, _HOST_UNARY, TWO_OVER_SQRT_PI, _lit,
                    no_distribution_validation, piecewise_rewrite, expand_rewrite)


# ---- hash-consed expression DAG ------------------------------------------------------------------------------------------
class _N:
    __slots__ = ("op", "args", "vars", "id")

    def __init__(self, op, args, vars_, id_):
        self.op, self.args, self.vars, self.id = op, args, vars_, id_


class _Dag:
    """Nodes: ('c', value) | ('x',) row value | ('p', k) row constant | ('h', i) head | ('S', k) sum | (op, *children) |
    ('sel', e, lo, hi): e on the rows lo <= d < hi, 0 elsewhere."""

    def __init__(self):
        self.table = {}

    def mk(self, op, *args):
        if op == "c":
            key = ("c", float(args[0]))
        else:
            key = (op,) + tuple(a.id if isinstance(a, _N) else a for a in args)
        n = self.table.get(key)
        if n is None:
            if op in ("c", "x", "p", "h", "S"):
                vs = {"c": frozenset(), "x": frozenset({"x"}), "p": frozenset({"row"}), "h": frozenset({("h", args[0])} if op == "h" else ()),
                      "S": frozenset({("S", args[0])} if op == "S" else ())}[op]
            else:
                vs = frozenset().union(*[a.vars for a in args if isinstance(a, _N)])
                if op == "sel":
                    vs = vs | {"row"}  # (lives in a row lambda even when e is a constant: it reads the row index)
            n = _N(op, args, vs, len(self.table))
            self.table[key] = n
        return n

    def const(self, v):
        return self.mk("c", float(v))

    # -- constructors with the simplifications symbolic derivatives need --------------------------------------------------
    def is_c(self, n, v=None):
        return n.op == "c" and (v is None or n.args[0] == v)

    def add(self, a, b):
        if self.is_c(a, 0.0):
            return b
        if self.is_c(b, 0.0):
            return a
        if a.op == "c" and b.op == "c":
            return self.const(a.args[0] + b.args[0])
        return self.mk("add", a, b)

    def sub(self, a, b):
        if self.is_c(b, 0.0):
            return a
        if self.is_c(a, 0.0):
            return self.neg(b)
        if a.op == "c" and b.op == "c":
            return self.const(a.args[0] - b.args[0])
        return self.mk("sub", a, b)

    def mul(self, a, b):
        if self.is_c(a, 0.0) or self.is_c(b, 0.0):
            return self.const(0.0)
        if self.is_c(a, 1.0):
            return b
        if self.is_c(b, 1.0):
            return a
        if a.op == "c" and b.op == "c":
            return self.const(a.args[0] * b.args[0])
        if a is b:
            return self.un("square", a)   # (and its derivative 2 a a' instead of a' a + a a')
        if self.is_c(a, -1.0):
            return self.neg(b)
        if self.is_c(b, -1.0):
            return self.neg(a)
        # constants gather: c1 * (c2 * x) -> (c1 c2) * x  (symbolic derivatives produce chains of them)
        for c, o in ((a, b), (b, a)):
            if c.op == "c" and o.op == "mul":
                for k in (0, 1):
                    if o.args[k].op == "c":
                        return self.mul(self.const(c.args[0] * o.args[k].args[0]), o.args[1 - k])
            if c.op == "c" and o.op == "neg":
                return self.mul(self.const(-c.args[0]), o.args[0])
        return self.mk("mul", a, b)

    def div(self, a, b):
        if self.is_c(a, 0.0):
            return a
        if self.is_c(b, 1.0):
            return a
        if a.op == "c" and b.op == "c" and b.args[0] != 0.0:
            return self.const(a.args[0] / b.args[0])
        return self.mk("div", a, b)

    def neg(self, a):
        if a.op == "c":
            return self.const(-a.args[0])
        if a.op == "neg":
            return a.args[0]
        return self.mk("neg", a)

    def un(self, op, a):
        if op == "neg":
            return self.neg(a)
        if a.op == "c":
            try:
                f = _HOST_UNARY.get(op)
                if f is not None:
                    return self.const(f(a.args[0]))
            except (ValueError, OverflowError):
                pass
        return self.mk(op, a)

    def sel(self, e, lo, hi):
        if self.is_c(e, 0.0):
            return e
        if e.op == "sel" and e.args[1:] == (lo, hi):
            return e
        return self.mk("sel", e, lo, hi)

    def bin(self, op, a, b):
        if op == "add":
            return self.add(a, b)
        if op == "sub":
            return self.sub(a, b)
        if op == "mul":
            return self.mul(a, b)
        if op == "div":
            return self.div(a, b)
        if op == "pow":
            if a.vars and b.vars:
                raise Unsupported("pow with both base and exponent depending on theta")
            if b.op == "c":
                if b.args[0] == 2.0:
                    return self.un("square", a)
                if b.args[0] == 1.0:
                    return a
                if b.args[0] == 0.5:
                    return self.un("sqrt", a)
                if a.op == "c":
                    return self.const(a.args[0] ** b.args[0])
            return self.mk("pow", a, b)
        if op in _COMPARE or op in ("maximum", "minimum"):
            return self.mk(op, a, b)
        raise AssertionError(op)

    def where(self, c, a, b):
        if a is b:
            return a
        return self.mk("where", c, a, b)

    # -- symbolic derivative with respect to one variable -------------------------------------------------------------------
    def diff(self, e, var, memo=None):
        memo = {} if memo is None else memo
        key = e.id
        if key in memo:
            return memo[key]
        r = self._diff(e, var, memo)
        memo[key] = r
        return r

    def _diff(self, e, var, memo):
        zero, one = self.const(0.0), self.const(1.0)
        if var not in e.vars:
            return zero
        op = e.op
        if op in ("x", "h", "S"):
            return one
        d = lambda a: self.diff(a, var, memo)  # noqa: E731
        if op == "sel":
            return self.sel(d(e.args[0]), e.args[1], e.args[2])
        if op == "where":
            return self.where(e.args[0], d(e.args[1]), d(e.args[2]))
        if op in ("maximum", "minimum"):  # torch: the derivative goes to the selected operand (ties: split evenly)
            a, b = e.args
            da, db = d(a), d(b)
            first = self.mk("gt" if op == "maximum" else "lt", a, b)
            tie = self.mul(self.const(0.5), self.add(da, db))
            return self.where(first, da, self.where(self.mk("ge" if op == "maximum" else "le", a, b), tie, db))
        if op in _COMPARE:
            return zero
        if op == "add":
            return self.add(d(e.args[0]), d(e.args[1]))
        if op == "sub":
            return self.sub(d(e.args[0]), d(e.args[1]))
        if op == "mul":
            a, b = e.args
            return self.add(self.mul(d(a), b), self.mul(a, d(b)))
        if op == "div":
            a, b = e.args
            da, db = d(a), d(b)
            if self.is_c(db, 0.0):
                return self.div(da, b)
            return self.div(self.sub(da, self.mul(e, db)), b)
        a = e.args[0]
        da = d(a)
        if op == "neg":
            return self.neg(da)
        if op == "exp":
            return self.mul(e, da)
        if op == "log":
            return self.div(da, a)
        if op == "log1p":
            return self.div(da, self.add(one, a))
        if op == "expm1":
            return self.mul(self.add(e, one), da)
        if op == "sigmoid":
            return self.mul(self.mul(e, self.sub(one, e)), da)
        if op == "logsigmoid":
            return self.mul(self.mk("sigmoid", self.neg(a)), da)
        if op == "softplus":
            return self.mul(self.mk("sigmoid", a), da)
        if op == "tanh":
            return self.mul(self.sub(one, self.mul(e, e)), da)
        if op == "sqrt":
            return self.div(da, self.mul(self.const(2.0), e))
        if op == "square":
            return self.mul(self.mul(self.const(2.0), a), da)
        if op == "abs":
            return self.mul(self.mk("sign", a), da)
        if op == "sin":
            return self.mul(self.mk("cos", a), da)
        if op == "cos":
            return self.neg(self.mul(self.mk("sin", a), da))
        if op == "sinh":
            return self.mul(self.mk("cosh", a), da)
        if op == "cosh":
            return self.mul(self.mk("sinh", a), da)
        if op == "atan":
            return self.div(da, self.add(one, self.mul(a, a)))
        if op == "erf":
            return self.mul(self.mul(self.const(TWO_OVER_SQRT_PI), self.mk("exp", self.neg(self.mul(a, a)))), da)
        if op == "pow":
            b = e.args[1]
            if not b.vars:  # a^c
                return self.mul(self.mul(b, self.bin("pow", a, self.sub(b, one))), da)
            return self.mul(self.mul(e, self.un("log", a)), d(b))  # c^b
        raise AssertionError(op)


_C_UNARY = {"exp": "bk_exp({a})", "log": "log({a})", "log1p": "log1p({a})", "expm1": "expm1({a})", "sigmoid": "1.0 / (1.0 + exp(-{a}))",
            "logsigmoid": "fmin({a}, 0.0) - log1p(exp(-fabs({a})))", "softplus": "({a} > 20.0) ? {a} : log1p(exp({a}))",
            "tanh": "tanh({a})", "sqrt": "sqrt({a})", "square": "{a} * {a}", "abs": "fabs({a})", "sin": "sin({a})", "cos": "cos({a})",
            "neg": "-{a}", "sign": "(double)(({a} > 0.0) - ({a} < 0.0))", "sinh": "sinh({a})", "cosh": "cosh({a})",
            "atan": "atan({a})", "erf": "erf({a})"}
_C_BINARY = {"add": "{a} + {b}", "sub": "{a} - {b}", "mul": "{a} * {b}", "div": "{a} / {b}", "pow": "pow({a}, {b})",
             "maximum": "fmax({a}, {b})", "minimum": "fmin({a}, {b})", "gt": "(double)({a} > {b})", "ge": "(double)({a} >= {b})",
             "lt": "(double)({a} < {b})", "le": "(double)({a} <= {b})"}


# ---- values while walking the fx graph -------------------------------------------------------------------------------------
class _Row:      # (C, hi - lo): an expression of the row value x, row constants and head-derived per-chain values on rows lo..hi
    def __init__(self, e, lo, hi):
        self.e, self.lo, self.hi = e, lo, hi


class _Per:      # (C,) or, broadcast, (C, 1): an expression of heads and sums
    def __init__(self, e, bcast=False):
        self.e, self.bcast = e, bcast


class _RowConst:  # a tensor constant along the row axis, placed once its partner in an operation says which rows it spans
    def __init__(self, t):
        self.t = t


class _LanesTracer:
    def __init__(self, D):
        self.D = int(D)
        self.g = _Dag()
        self.ranges = set()   # row slices [lo, hi) the function takes
        self.max_head = -1
        self.rows = []        # packed row constants, each (D,) on the host (entries outside their slice unused)
        self.sums = []        # (e_k, lo, hi)

    @property
    def H(self):
        return min(lo for lo, _ in self.ranges) if self.ranges else None

    def take_rows(self, lo, hi):
        self.ranges.add((lo, hi))
        return _Row(self.g.mk("x"), lo, hi)

    def row_const(self, t, lo, hi, where):
        n = hi - lo
        if tuple(t.shape) not in ((n,), (1, n)):
            raise Unsupported(f"{where}: a tensor constant of shape {tuple(t.shape)} does not broadcast along the {n} rows "
                              f"Th[:, {lo}:{hi}] (expected a scalar or shape ({n},) / (1, {n}))")
        row = torch.zeros(self.D, dtype=torch.float64)
        row[lo:hi] = t.detach().reshape(n).to(dtype=torch.float64, device="cpu")
        for k, r in enumerate(self.rows):
            if torch.equal(r, row):
                return self.g.mk("p", k)
        self.rows.append(row)
        return self.g.mk("p", len(self.rows) - 1)

    def run(self, fn):
        import torch.fx as fx

        try:
            with no_distribution_validation():
                gm = fx.symbolic_trace(fn)
        except Exception as e:
            raise Unsupported(f"torch.fx could not trace the function: {type(e).__name__}: {e}") from e
        table = dict(_function_table())
        table[torch.unsqueeze] = "unsqueeze"
        methods = dict(_METHODS, unsqueeze="unsqueeze")
        env, out, n_inputs = {}, None, 0
        TH = object()
        for node in gm.graph.nodes:
            where = f"node `{node.format_node()}`"

            def val(a):
                if isinstance(a, fx.Node):
                    return env[a]
                if isinstance(a, (tuple, list)):
                    return type(a)(val(x) for x in a)
                return a

            if node.op == "placeholder":
                n_inputs += 1
                if n_inputs > 1:
                    raise Unsupported("the function takes more than one argument")
                env[node] = TH
            elif node.op == "get_attr":
                obj = gm
                for part in node.target.split("."):
                    obj = getattr(obj, part)
                env[node] = obj
            elif node.op in ("call_function", "call_method"):
                name = table.get(node.target) if node.op == "call_function" else methods.get(node.target)
                if name is None:
                    raise Unsupported(f"{where}: unsupported operation {getattr(node.target, '__name__', node.target)!r}")
                args = [val(a) for a in node.args]
                kwargs = {k: val(v) for k, v in node.kwargs.items()}
                env[node] = self.apply(name, args, kwargs, where, TH)
            elif node.op == "output":
                out = val(node.args[0])
            else:
                raise Unsupported(f"{where}: unsupported node kind {node.op}")
        if not isinstance(out, _Per) or out.bcast:
            raise Unsupported("the function does not end in a per-chain value built from head coordinates Th[:, i] and sums over "
                              "the rows Th[:, H:]")
        if not self.ranges or not self.sums:
            raise Unsupported("no row slice Th[:, H:] summed over dim=1: not a head-plus-sums density")
        if self.max_head >= self.H:
            raise Unsupported(f"head index {self.max_head} lies inside the row slice Th[:, {self.H}:]")
        if self.H > 8:
            raise Unsupported(f"{self.H} head coordinates (the lane-spread kernels hold at most 8)")
        return out.e

    def operand(self, v, where, TH):
        """A traced value as an operand of an elementwise operation."""
        if isinstance(v, (_Row, _Per, _RowConst)):
            return v
        if v is TH:
            return self.take_rows(0, self.D)  # the whole state as rows: a density without head coordinates
        if isinstance(v, bool):
            raise Unsupported(f"{where}: boolean operand")
        if isinstance(v, (int, float)):
            return _Per(self.g.const(v), bcast=None)  # scalar: fits both shapes
        if isinstance(v, torch.Tensor):
            if v.numel() == 1:
                return _Per(self.g.const(float(v.reshape(()).item())), bcast=None)
            return _RowConst(v)
        raise Unsupported(f"{where}: operand of type {type(v).__name__}")

    def shape_of(self, v, where, TH):
        if v is TH:
            return (None, self.D)
        if isinstance(v, _Row):
            return (None, v.hi - v.lo)
        if isinstance(v, _Per):
            return (None, 1) if v.bcast else (None,)
        if isinstance(v, torch.Tensor):
            return tuple(v.shape)
        raise Unsupported(f"{where}: shape of a value the tracer does not follow")

    def apply(self, name, args, kwargs, where, TH):
        g = self.g
        if name == "id":
            return args[0]
        if name == "broadcast_tensors":  # (torch.distributions' broadcast_all): the operands themselves; every operation
            return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list)) else tuple(args)  # broadcasts anyway
        if name == "getattr":
            if args[1] == "shape":
                return self.shape_of(args[0], where, TH)
            raise Unsupported(f"{where}: attribute {args[1]!r}")
        if name == "size":
            shp = self.shape_of(args[0], where, TH)
            return shp if len(args) == 1 else shp[args[1]]
        if name == "getitem":
            src, idx = args
            if isinstance(src, (tuple, list)) and isinstance(idx, int):
                return src[idx]
            idx = idx if isinstance(idx, tuple) else (idx,)
            full = lambda s: s is Ellipsis or (isinstance(s, slice) and s == slice(None, None, None))  # noqa: E731
            if src is TH:
                if len(idx) == 2 and full(idx[0]) and isinstance(idx[1], int) and not isinstance(idx[1], bool) and idx[1] >= 0:
                    self.max_head = max(self.max_head, idx[1])
                    return _Per(g.mk("h", idx[1]), bcast=False)
                if len(idx) == 2 and full(idx[0]) and isinstance(idx[1], slice) and idx[1].step in (None, 1):
                    lo, hi = idx[1].start, idx[1].stop
                    lo = 0 if lo is None else lo
                    hi = self.D if hi is None else (hi + self.D if isinstance(hi, int) and hi < 0 else hi)
                    if isinstance(lo, int) and isinstance(hi, int) and 0 <= lo < hi <= self.D:
                        return self.take_rows(lo, hi)
                raise Unsupported(f"{where}: theta may be indexed as Th[:, i] (a head coordinate) or Th[:, a:b] (rows) only")
            if isinstance(src, _Per) and src.bcast is False and len(idx) == 2 and full(idx[0]) and idx[1] is None:
                return _Per(src.e, bcast=True)
            raise Unsupported(f"{where}: unsupported indexing")
        if name == "unsqueeze":
            src = args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim")
            if isinstance(src, _Per) and src.bcast is False and dim in (1, -1):
                return _Per(src.e, bcast=True)
            raise Unsupported(f"{where}: unsupported unsqueeze")
        if name in ("sum", "mean"):
            x = self.operand(args[0], where, TH) if args[0] is TH else args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim", kwargs.get("axis"))
            if isinstance(dim, (list, tuple)) and len(dim) == 1:
                dim = dim[0]
            if kwargs.get("keepdim", False) or (len(args) > 2 and args[2]) or kwargs.get("dtype") not in (None, torch.float64):
                raise Unsupported(f"{where}: {name}(keepdim=True) / {name}(dtype=...)")
            if not isinstance(x, _Row) or "x" not in x.e.vars and "row" not in x.e.vars:
                raise Unsupported(f"{where}: {name} of something that is not a row expression")
            if dim not in (1, -1):
                raise Unsupported(f"{where}: the {name} must run over the row axis (dim=1), got dim={dim!r}")
            self.sums.append((x.e, x.lo, x.hi))
            S = g.mk("S", len(self.sums) - 1)
            return _Per(S if name == "sum" else g.div(S, g.const(float(x.hi - x.lo))), bcast=False)
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
        if name in ("reciprocal", "rsqrt"):
            if len(args) != 1:
                raise Unsupported(f"{where}: {name} with {len(args)} operands")
            den = args[0] if name == "reciprocal" else self.apply("sqrt", [args[0]], {}, where, TH)
            return self.apply("div", [1.0, den], {}, where, TH)
        if name in _UNARY:
            if len(args) != 1:
                raise Unsupported(f"{where}: {name} with {len(args)} operands")
            a = self.operand(args[0], where, TH)
            if isinstance(a, _RowConst):
                raise Unsupported(f"{where}: {name} of a tensor constant inside the traced function (compute it outside)")
            e = g.un(name, a.e)
            return _Row(e, a.lo, a.hi) if isinstance(a, _Row) else _Per(e, a.bcast)
        rw = piecewise_rewrite(name, args, kwargs, where)
        if rw is not None:
            return expand_rewrite(rw, lambda n, a: self.apply(n, a, {}, where, TH))
        if name in _BINARY or name == "where":
            want = 3 if name == "where" else 2
            if len(args) != want:
                raise Unsupported(f"{where}: {name} with {len(args)} operands (alpha= / rounding_mode= are not supported)")
            ops = [self.operand(v, where, TH) for v in args]
            if name == "where" and not (isinstance(ops[0], (_Row, _Per)) and ops[0].e.op in _COMPARE):
                raise Unsupported(f"{where}: the condition of `where` must be a comparison (> >= < <=)")
            build = (lambda es: g.where(*es)) if name == "where" else (lambda es: g.bin(name, *es))  # noqa: E731
            rows = [o for o in ops if isinstance(o, _Row)]
            if rows:
                lo, hi = rows[0].lo, rows[0].hi
                if any((o.lo, o.hi) != (lo, hi) for o in rows):
                    raise Unsupported(f"{where}: rows of two different slices in one expression ("
                                      + " and ".join(sorted({f"Th[:, {o.lo}:{o.hi}]" for o in rows}))
                                      + "); each expression and its sum must stay within ONE slice")
                es = []
                for o in ops:
                    if isinstance(o, _RowConst):
                        es.append(self.row_const(o.t, lo, hi, where))
                        continue
                    if isinstance(o, _Per):
                        if o.bcast is False:
                            raise Unsupported(f"{where}: a per-chain value meets a row expression without [:, None]")
                        if any(isinstance(v, tuple) and v[0] == "S" for v in o.e.vars):
                            raise Unsupported(f"{where}: a sum over the rows is used inside another row expression "
                                              "(rows may depend on head coordinates only)")
                    es.append(o.e)
                return _Row(build(es), lo, hi)
            if any(isinstance(o, _RowConst) for o in ops):
                raise Unsupported(f"{where}: a tensor constant meets a per-chain value before any row slice Th[:, a:b]: which rows it "
                                  "spans is unknown (combine it with the rows first)")
            shapes = {o.bcast for o in ops if o.bcast is not None}
            if len(shapes) > 1:
                raise Unsupported(f"{where}: a (C,) value meets a (C, 1) value")
            return _Per(build([o.e for o in ops]), shapes.pop() if shapes else None)
        raise Unsupported(f"{where}: unsupported operation {name}")


# ---- code generation: chain scope + row lambdas ----------------------------------------------------------------------------
class _Gen:
    def __init__(self, D, H):
        self.D, self.H = D, H
        self.outer = []          # lines of the chain scope, in order
        self.names = {}          # node id -> name in the chain scope
        self.n = 0

    def tmp(self, lines, expr):
        name = f"t{self.n}"
        self.n += 1
        lines.append(f"  const double {name} = {expr};")
        return name

    @staticmethod
    def row_level(e):
        return "x" in e.vars or "row" in e.vars

    def chain(self, e):
        """Emit e (no row dependency) in the chain scope; returns its name / literal."""
        if e.id in self.names:
            return self.names[e.id]
        if e.op == "c":
            return _lit(e.args[0])
        if e.op == "h":
            r = f"h{e.args[0]}"
        elif e.op == "S":
            r = f"S{e.args[0]}"
        else:
            r = self.tmp(self.outer, self.expr(e, [self.chain(a) for a in e.args if isinstance(a, _N)]))
        self.names[e.id] = r
        return r

    def row(self, e, lines, memo):
        """Emit e inside a row lambda (lines); chain-level subtrees are hoisted into the chain scope and captured."""
        if not self.row_level(e):
            return self.chain(e)
        if e.id in memo:
            return memo[e.id]
        if e.op == "x":
            r = "x"
        elif e.op == "p":
            r = self.tmp(lines, f"P[{e.args[0] * self.D} + d]")
        else:
            r = self.tmp(lines, self.expr(e, [self.row(a, lines, memo) for a in e.args if isinstance(a, _N)]))
        memo[e.id] = r
        return r

    def expr(self, e, a):
        if e.op == "sel":
            lo, hi = e.args[1], e.args[2]
            if (lo, hi) == (self.H, self.D):
                return a[0]
            return f"(d >= {lo} && d < {hi}) ? {a[0]} : 0.0"
        if e.op == "where":
            return f"({a[0]} != 0.0) ? {a[1]} : {a[2]}"
        if e.op in _C_UNARY:
            return _C_UNARY[e.op].format(a=a[0])
        return _C_BINARY[e.op].format(a=a[0], b=a[1])

    def lam(self, e):
        """`[=](double x, i64 d) { ...; return r; }` computing the row expression e."""
        lines, memo = [], {}
        r = self.row(e, lines, memo)
        body = "".join("    " + ln.strip() + "\n" for ln in lines)
        return "[=](double x, i64 d) {\n" + body + f"    return {r};\n  }}"


def lanes_source(fn, dims: int):
    """(HIP source defining bk_lanes_density, head count H, packed params tensor [n_rows * D] on the host or None, description).
    Raises trace.Unsupported."""
    D = int(dims)
    tr = _LanesTracer(D)
    F = tr.run(fn)
    g, H, K = tr.g, tr.H, len(tr.sums)
    if D - H < 1:
        raise Unsupported("no rows beyond the head coordinates")
    sums = [g.sel(e, lo, hi) for e, lo, hi in tr.sums]  # each sum runs over ITS slice: zero on the other rows
    heads = sorted({v[1] for v in F.vars if isinstance(v, tuple) and v[0] == "h"}
                   | {v[1] for e in sums for v in e.vars if isinstance(v, tuple) and v[0] == "h"})
    gen = _Gen(D, H)
    out = []
    for i in range(H):
        gen.outer.append(f"  const double h{i} = c.head({i});")
    # the sums, then the extra sums T[k][i] = sum_d d e_k / d h_i
    for k, e in enumerate(sums):
        lam = gen.lam(e)
        gen.outer.append(f"  const double S{k} = c.sum({lam});")
    T = {}
    for k, e in enumerate(sums):
        for i in heads:
            de = g.diff(e, ("h", i))
            if g.is_c(de, 0.0):
                continue
            if de.op == "sel" and not _Gen.row_level(de.args[0]):
                # a row-independent derivative summed over the slice: (hi - lo) copies of it
                T[(k, i)] = g.mul(g.const(float(de.args[2] - de.args[1])), de.args[0])
                continue
            lam = gen.lam(de)
            gen.outer.append(f"  const double T{k}_{i} = c.sum({lam});")
            node = g.mk("S", K + len([t for t in T.values() if t.op == "S"]))
            gen.names[node.id] = f"T{k}_{i}"
            T[(k, i)] = node
    FS = [g.diff(F, ("S", k)) for k in range(K)]
    # d log p / d h_i
    for i in range(H):
        gh = g.diff(F, ("h", i))
        for k in range(K):
            if (k, i) in T:
                gh = g.add(gh, g.mul(FS[k], T[(k, i)]))
        gen.outer.append(f"  c.grad_head({i}, {gen.chain(gh)});")
    # d log p / d theta_d
    gx = g.const(0.0)
    for k, e in enumerate(sums):
        gx = g.add(gx, g.mul(FS[k], g.diff(e, "x")))
    if g.is_c(gx, 0.0):
        raise Unsupported("the log density does not depend on the rows")
    val = gen.chain(F)          # (emitted before the row gradient: grad() must be the context's last call)
    lam = gen.lam(gx)
    gen.outer.append(f"  c.grad({lam});")
    gen.outer.append(f"  return {val};")
    src = ("// generated by bayes_kit_amd.trace_lanes from a PyTorch log density (torch.fx graph, symbolic derivatives):\n"
           f"// {H} head coordinate(s), {K} sum(s) over the rows, {len([t for t in T.values() if t.op == 'S'])} derivative sum(s)\n"
           "template <class L>\n__device__ double bk_lanes_density(L& c, const double* P) {\n" + "\n".join(gen.outer) + "\n}\n")
    params = torch.cat(tr.rows) if tr.rows else None
    return src, H, params, {"head": H, "sums": K, "derivative_sums": len([t for t in T.values() if t.op == "S"]),
                            "param_rows": len(tr.rows), "temporaries": gen.n}
