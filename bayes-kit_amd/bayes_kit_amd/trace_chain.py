"""PyTorch log density -> HIP source for ``CTarget.from_source(form="chain")`` (``TorchModel(compile=True)``, third try).

``trace.py`` handles separable densities, ``trace_lanes.py`` head-plus-exchangeable-rows ones.  This module takes what is left of
the "vector expression" models -- densities whose coordinates are COUPLED THROUGH SHIFTED SLICES:

    log p(theta) = F( theta_k ...,  S_0 .. S_{K-1} ),      S_k = sum_{i < n_k} e_k( theta_{a+i}, theta_{b+i}, ..., i )

i.e. scalars taken by integer index (``Th[:, 0]``), vectors taken as slices (``Th[:, 1:]``, ``Th[:, :-1]``, slices of computed
vectors, ``torch.diff``), elementwise expressions that may combine slices of DIFFERENT offsets (a random walk's increments
``x[:, 1:] - phi[:, None] * x[:, :-1]``, a stochastic-volatility likelihood, a smoothness prior) and per-chain scalars broadcast with
``[:, None]``, sums / means of those over the coordinate axis, and any scalar expression at the end.  AR(1) and random-walk priors,
state-space and stochastic-volatility models are of this shape.

Every vector value is kept as an expression of an index ``i`` whose leaves ``theta_{a+i}`` carry their offsets (slicing a
computed vector re-bases its leaves; nothing is materialised).  The function is read once with ``torch.fx`` into trace_lanes.py's
hash-consed DAG and differentiated SYMBOLICALLY, one variable per leaf offset:

    d log p / d theta_j = sum_k F_{S_k} * sum_{a in offsets(e_k), 0 <= j - a < n_k}  d e_k / d theta_{a+i} (i = j - a)
                          + [j is taken as a scalar] ( F_{theta_j} + sum_k F_{S_k} * sum_i d e_k / d theta_j (i) )

and emitted as a ``bk_chain`` (one lane walks one chain; csrc/bk_source_api.hpp's accessors): loops with literal bounds, fully
unrolled for D <= 128 so that the staged coordinates stay in registers and every range test folds at compile time.  The samplers
then get the compiled gradient op, one launch per leapfrog step and one launch per trajectory.  Code generation for ONE backend;
anything outside the shape raises ``trace.Unsupported`` naming the node.  Only the reference's (C, D) layout.
"""
from __future__ import annotations

import torch

from .trace import Unsupported, _function_table, _METHODS, _UNARY, _BINARY, _COMPARE, _lit, no_distribution_validation, piecewise_rewrite, expand_rewrite
from .trace_lanes import _C_BINARY, _C_UNARY, _Dag, _N


class _ChainDag(_Dag):
    """trace_lanes.py's DAG with offset-carrying leaves: ('x', a): theta_{a+i} | ('p', k, off): constant k's entry off + i."""

    def mk(self, op, *args):
        if op == "x":
            key = ("x", int(args[0]))
            n = self.table.get(key)
            if n is None:
                n = _N("x", (int(args[0]),), frozenset({("x", int(args[0]))}), len(self.table))
                self.table[key] = n
            return n
        if op == "p":
            key = ("p", int(args[0]), int(args[1]))
            n = self.table.get(key)
            if n is None:
                n = _N("p", (int(args[0]), int(args[1])), frozenset({"row"}), len(self.table))
                self.table[key] = n
            return n
        return super().mk(op, *args)

    def rebuild(self, e, leaf, memo):
        """e with every index-dependent leaf replaced by leaf(node)."""
        if e.id in memo:
            return memo[e.id]
        if e.op in ("x", "p"):
            r = leaf(e)
        elif e.op in ("c", "h", "S"):
            r = e
        else:
            a = [self.rebuild(x, leaf, memo) if isinstance(x, _N) else x for x in e.args]
            if e.op in _UNARY or e.op == "sign":
                r = self.un(e.op, a[0]) if e.op in _UNARY else self.mk("sign", a[0])
            elif e.op in _BINARY:
                r = self.bin(e.op, a[0], a[1])
            elif e.op == "where":
                r = self.where(*a)
            else:
                raise AssertionError(e.op)
        memo[e.id] = r
        return r

    def rebase(self, e, delta):
        """The vector v[delta:]: every leaf's offset moved by delta."""
        if delta == 0:
            return e
        return self.rebuild(e, lambda n: self.mk("x", n.args[0] + delta) if n.op == "x" else self.mk("p", n.args[0], n.args[1] + delta), {})


def _indexed(e):
    return any(v == "row" or (isinstance(v, tuple) and v[0] == "x") for v in e.vars)


class _Vec:      # (C, n): an expression of the index i < n
    def __init__(self, e, n):
        self.e, self.n = e, int(n)


class _Per:      # (C,) or, broadcast, (C, 1): an expression of scalars theta_k and sums
    def __init__(self, e, bcast=False):
        self.e, self.bcast = e, bcast


class _VecConst:  # a tensor constant along the coordinate axis, placed when its partner in an operation says how long it is
    def __init__(self, t):
        self.t = t


class _ChainTracer:
    def __init__(self, D):
        self.D = int(D)
        self.g = _ChainDag()
        self.consts = []      # constant vectors (host, float64), packed one after the other into the params array
        self.sums = []        # (e_k, n_k)

    def const_base(self, k):
        return sum(int(c.numel()) for c in self.consts[:k])

    def vec_const(self, t, n, where):
        if tuple(t.shape) not in ((n,), (1, n)):
            raise Unsupported(f"{where}: a tensor constant of shape {tuple(t.shape)} does not broadcast along a vector of "
                              f"{n} coordinates (expected a scalar or shape ({n},) / (1, {n}))")
        v = t.detach().reshape(n).to(dtype=torch.float64, device="cpu")
        for k, c in enumerate(self.consts):
            if c.shape == v.shape and torch.equal(c, v):
                return self.g.mk("p", k, 0)
        self.consts.append(v)
        return self.g.mk("p", len(self.consts) - 1, 0)

    def at(self, e, i0):
        """The entry i0 of a vector expression, as a per-chain scalar."""
        g = self.g
        return g.rebuild(e, lambda n: g.mk("h", n.args[0] + i0) if n.op == "x"
                         else g.const(float(self.consts[n.args[0]][n.args[1] + i0])), {})

    def run(self, fn):
        import torch.fx as fx

        try:
            with no_distribution_validation():
                gm = fx.symbolic_trace(fn)
        except Exception as e:
            raise Unsupported(f"torch.fx could not trace the function: {type(e).__name__}: {e}") from e
        table = dict(_function_table())
        table[torch.unsqueeze] = "unsqueeze"
        table[torch.diff] = "diff"
        methods = dict(_METHODS, unsqueeze="unsqueeze", diff="diff")
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
            raise Unsupported("the function does not end in a per-chain value built from coordinates Th[:, k] and sums over slices of Th")
        if not out.e.vars:
            raise Unsupported("the log density does not depend on theta")
        return out.e

    def operand(self, v, where, TH):
        if isinstance(v, (_Vec, _Per, _VecConst)):
            return v
        if v is TH:
            return _Vec(self.g.mk("x", 0), self.D)
        if isinstance(v, bool):
            raise Unsupported(f"{where}: boolean operand")
        if isinstance(v, (int, float)):
            return _Per(self.g.const(v), bcast=None)
        if isinstance(v, torch.Tensor):
            if v.numel() == 1:
                return _Per(self.g.const(float(v.reshape(()).item())), bcast=None)
            return _VecConst(v)
        raise Unsupported(f"{where}: operand of type {type(v).__name__}")

    def shape_of(self, v, where, TH):
        if v is TH:
            return (None, self.D)
        if isinstance(v, _Vec):
            return (None, v.n)
        if isinstance(v, _Per):
            return (None, 1) if v.bcast else (None,)
        if isinstance(v, torch.Tensor):
            return tuple(v.shape)
        raise Unsupported(f"{where}: shape of a value the tracer does not follow")

    def take(self, src, idx, where, TH):
        """src[:, idx] for src = theta or a computed vector."""
        v = self.operand(src, where, TH)
        if isinstance(idx, bool) or not isinstance(idx, (int, slice)):
            raise Unsupported(f"{where}: a vector may be indexed as v[:, k] or v[:, a:b] only")
        if isinstance(idx, int):
            k = idx + v.n if idx < 0 else idx
            if not 0 <= k < v.n:
                raise Unsupported(f"{where}: index {idx} outside a vector of {v.n} coordinates")
            return _Per(self.at(v.e, k), bcast=False)
        if idx.step not in (None, 1):
            raise Unsupported(f"{where}: strided slices are not supported")
        lo, hi = idx.start, idx.stop
        lo = 0 if lo is None else (lo + v.n if lo < 0 else lo)
        hi = v.n if hi is None else (hi + v.n if hi < 0 else hi)
        if not (isinstance(lo, int) and isinstance(hi, int) and 0 <= lo < hi <= v.n):
            raise Unsupported(f"{where}: slice {idx.start}:{idx.stop} of a vector of {v.n} coordinates is empty or out of range")
        return _Vec(self.g.rebase(v.e, lo), hi - lo)

    def apply(self, name, args, kwargs, where, TH):
        g = self.g
        if name == "id":
            return args[0]
        if name == "broadcast_tensors":
            return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list)) else tuple(args)
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
            if (src is TH or isinstance(src, _Vec)) and len(idx) == 2 and full(idx[0]):
                return self.take(src, idx[1], where, TH)
            if isinstance(src, _Per) and src.bcast is False and len(idx) == 2 and full(idx[0]) and idx[1] is None:
                return _Per(src.e, bcast=True)
            raise Unsupported(f"{where}: unsupported indexing (theta / a vector: v[:, k], v[:, a:b]; a per-chain value: s[:, None])")
        if name == "unsqueeze":
            src = args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim")
            if isinstance(src, _Per) and src.bcast is False and dim in (1, -1):
                return _Per(src.e, bcast=True)
            raise Unsupported(f"{where}: unsupported unsqueeze")
        if name == "diff":
            if kwargs.get("n", args[1] if len(args) > 1 else 1) != 1 or kwargs.get("dim", args[2] if len(args) > 2 else -1) not in (1, -1) \
                    or kwargs.get("prepend") is not None or kwargs.get("append") is not None:
                raise Unsupported(f"{where}: torch.diff with n != 1, another dim, prepend= or append=")
            v = self.operand(args[0], where, TH)
            if not isinstance(v, _Vec) or v.n < 2:
                raise Unsupported(f"{where}: diff of something that is not a vector of at least two coordinates")
            return _Vec(g.sub(g.rebase(v.e, 1), v.e), v.n - 1)
        if name in ("sum", "mean"):
            x = self.operand(args[0], where, TH) if args[0] is TH else args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim", kwargs.get("axis"))
            if isinstance(dim, (list, tuple)) and len(dim) == 1:
                dim = dim[0]
            if kwargs.get("keepdim", False) or (len(args) > 2 and args[2]) or kwargs.get("dtype") not in (None, torch.float64):
                raise Unsupported(f"{where}: {name}(keepdim=True) / {name}(dtype=...)")
            if not isinstance(x, _Vec):
                raise Unsupported(f"{where}: {name} of something that is not a vector expression")
            if dim not in (1, -1):
                raise Unsupported(f"{where}: the {name} must run over the coordinate axis (dim=1), got dim={dim!r}")
            if not _indexed(x.e):   # a per-chain value broadcast along n coordinates
                S = g.mul(g.const(float(x.n)), x.e)
            else:
                self.sums.append((x.e, x.n))
                S = g.mk("S", len(self.sums) - 1)
            return _Per(S if name == "sum" else g.div(S, g.const(float(x.n))), bcast=False)
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
            if isinstance(a, _VecConst):
                raise Unsupported(f"{where}: {name} of a tensor constant inside the traced function (compute it outside)")
            e = g.un(name, a.e)
            return _Vec(e, a.n) if isinstance(a, _Vec) else _Per(e, a.bcast)
        rw = piecewise_rewrite(name, args, kwargs, where)
        if rw is not None:
            return expand_rewrite(rw, lambda n, a: self.apply(n, a, {}, where, TH))
        if name in _BINARY or name == "where":
            want = 3 if name == "where" else 2
            if len(args) != want:
                raise Unsupported(f"{where}: {name} with {len(args)} operands (alpha= / rounding_mode= are not supported)")
            ops = [self.operand(v, where, TH) for v in args]
            if name == "where" and not (isinstance(ops[0], (_Vec, _Per)) and ops[0].e.op in _COMPARE):
                raise Unsupported(f"{where}: the condition of `where` must be a comparison (> >= < <=)")
            build = (lambda es: g.where(*es)) if name == "where" else (lambda es: g.bin(name, *es))  # noqa: E731
            vecs = [o for o in ops if isinstance(o, _Vec)]
            if vecs:
                n = vecs[0].n
                if any(o.n != n for o in vecs):
                    raise Unsupported(f"{where}: vectors of different lengths (" + ", ".join(str(o.n) for o in vecs) + ") in one operation")
                es = []
                for o in ops:
                    if isinstance(o, _VecConst):
                        es.append(self.vec_const(o.t, n, where))
                        continue
                    if isinstance(o, _Per):
                        if o.bcast is False:
                            raise Unsupported(f"{where}: a per-chain value meets a vector without [:, None]")
                        if any(isinstance(v, tuple) and v[0] == "S" for v in o.e.vars):
                            raise Unsupported(f"{where}: a sum over coordinates is used inside another vector expression "
                                              "(vector expressions may depend on the coordinates Th[:, k] only)")
                    es.append(o.e)
                return _Vec(build(es), n)
            if any(isinstance(o, _VecConst) for o in ops):
                raise Unsupported(f"{where}: a tensor constant meets a per-chain value before any slice of theta: which coordinates "
                                  "it spans is unknown (combine it with the slice first)")
            shapes = {o.bcast for o in ops if o.bcast is not None}
            if len(shapes) > 1:
                raise Unsupported(f"{where}: a (C,) value meets a (C, 1) value")
            return _Per(build([o.e for o in ops]), shapes.pop() if shapes else None)
        raise Unsupported(f"{where}: unsupported operation {name}")


# ---- code generation: a bk_chain of loops with literal bounds -------------------------------------------------------------------
class _Gen:
    def __init__(self, tracer):
        self.tr = tracer
        self.outer = []      # lines of the chain scope, in order
        self.names = {}      # node id -> name in the chain scope
        self.heads = set()
        self.n = 0
        self.loops = 0
        self.heavy = False

    def tmp(self, lines, expr, indent="  "):
        name = f"t{self.n}"
        self.n += 1
        lines.append(f"{indent}const double {name} = {expr};")
        return name

    @staticmethod
    def expr(e, a):
        if e.op == "where":
            return f"({a[0]} != 0.0) ? {a[1]} : {a[2]}"
        if e.op in _C_UNARY:
            return _C_UNARY[e.op].format(a=a[0])
        return _C_BINARY[e.op].format(a=a[0], b=a[1])

    def chain(self, e):
        """Emit e (no index dependency) in the chain scope; returns its name / literal."""
        if e.id in self.names:
            return self.names[e.id]
        if e.op == "c":
            return _lit(e.args[0])
        if e.op == "h":
            r = f"h{e.args[0]}"
            if e.args[0] not in self.heads:
                self.heads.add(e.args[0])
                self.outer.append(f"  const double {r} = th[{e.args[0]}];")
        elif e.op == "S":
            r = f"S{e.args[0]}"
        else:
            r = self.tmp(self.outer, self.expr(e, [self.chain(a) for a in e.args if isinstance(a, _N)]))
        self.names[e.id] = r
        return r

    def at_index(self, e, lines, memo, indent):
        """Emit e inside a loop over i (lines); chain-level subtrees are hoisted into the chain scope."""
        if not _indexed(e):
            return self.chain(e)
        if e.id in memo:
            return memo[e.id]
        if e.op == "x":
            r = self.tmp(lines, f"th[i + {e.args[0]}]" if e.args[0] else "th[i]", indent)
        elif e.op == "p":
            r = self.tmp(lines, f"P[i + {self.tr.const_base(e.args[0]) + e.args[1]}]", indent)
        else:
            r = self.tmp(lines, self.expr(e, [self.at_index(a, lines, memo, indent) for a in e.args if isinstance(a, _N)]), indent)
        memo[e.id] = r
        return r

    def sum_loop(self, name, e, n, unroll):
        lines, memo = [], {}
        r = self.at_index(e, lines, memo, "    ")
        if unroll and self.heavy and self.loops:
            # (each pass over the coordinates recomputes what it needs: without this the compiler keeps the per-coordinate
            # subexpressions two unrolled loops share -- every residual, every exp -- alive from one to the other)
            self.outer.append("  th.fence(); BK_CHAIN_FORGET(P);")
        self.loops += 1
        self.outer.append(f"  double {name} = 0.0;")
        if unroll:
            self.outer.append("#pragma unroll")
        self.outer.append(f"  for (i64 i = 0; i < {n}; ++i) {{")
        self.outer.extend(lines)
        self.outer.append(f"    {name} = {name} + {r};")
        if unroll and self.heavy:
            self.outer.append("    BK_CHAIN_PACE(i);")
        self.outer.append("  }")


def chain_source(fn, dims: int):
    """(HIP source defining bk_chain, packed params tensor on the host or None, description).  Raises trace.Unsupported."""
    D = int(dims)
    tr = _ChainTracer(D)
    F = tr.run(fn)
    g, K = tr.g, len(tr.sums)
    unroll = D <= 128   # (the staged coordinates live in registers only under full unrolling: bk_source_kernels.hpp)
    # Passes of a HEAVY function (transcendentals, divisions, vector constants) are separated by fences and handed to the
    # scheduler four iterations at a time; a function of plain arithmetic is left alone (measured on the generated kernels'
    # scratch use at D = 101: fences cut a stochastic-volatility model's spills by a third and triple a random walk's)
    # (Staging a heavy function's coordinates in LDS instead -- from_source(stage="lds"), loops four / sixteen iterations at a time or
    # unrolled -- was measured on an AR(1) state-space model at D = 101: 22 / 13 / 19 us per in-kernel leapfrog step against 10.5
    # for the register-staged, fenced form generated here.)
    cheap = {"c", "x", "p", "h", "S", "add", "sub", "mul", "neg", "square", "abs", "sign", "where", "maximum", "minimum", "gt", "ge",
             "lt", "le"}
    gen_heavy = bool(tr.consts) or any(n.op not in cheap for n in g.table.values())
    heads = sorted({v[1] for v in F.vars if isinstance(v, tuple) and v[0] == "h"}
                   | {v[1] for e, _ in tr.sums for v in e.vars if isinstance(v, tuple) and v[0] == "h"})
    gen = _Gen(tr)
    gen.heavy = gen_heavy and unroll
    for k, (e, n) in enumerate(tr.sums):
        gen.sum_loop(f"S{k}", e, n, unroll)
    val = gen.chain(F)
    # everything below is needed for the gradient only
    pre_grad = len(gen.outer)
    T = {}
    n_extra = 0
    for k, (e, n) in enumerate(tr.sums):
        for i in heads:
            de = g.diff(e, ("h", i))
            if g.is_c(de, 0.0):
                continue
            if not _indexed(de):
                T[(k, i)] = g.mul(g.const(float(n)), de)
                continue
            node = g.mk("S", K + n_extra)
            gen.sum_loop(f"S{K + n_extra}", de, n, unroll)
            n_extra += 1
            T[(k, i)] = node
    FS = [g.diff(F, ("S", k)) for k in range(K)]
    gh = {}
    for i in heads:
        v = g.diff(F, ("h", i))
        for k in range(K):
            if (k, i) in T:
                v = g.add(v, g.mul(FS[k], T[(k, i)]))
        if not g.is_c(v, 0.0):
            gh[i] = gen.chain(v)
    # d log p / d theta_j through the vector leaves: one guarded block per (sum, leaf offset)
    blocks = []
    for k, (e, n) in enumerate(tr.sums):
        if g.is_c(FS[k], 0.0):
            continue
        for a in sorted(v[1] for v in e.vars if isinstance(v, tuple) and v[0] == "x"):
            de = g.diff(e, ("x", a))
            if g.is_c(de, 0.0):
                continue
            lines, memo = [], {}
            r = gen.at_index(g.mul(FS[k], de), lines, memo, "        ")   # (F_{S_k} is hoisted into the chain scope)
            blocks.append((a, n, lines, r))
    if not blocks and not gh:
        raise Unsupported("the log density does not depend on theta")
    body = list(gen.outer[:pre_grad])
    body.append("  if (g.wanted()) {")
    body.extend("  " + ln for ln in gen.outer[pre_grad:])
    if gen.heavy:
        body.append("    th.fence(); BK_CHAIN_FORGET(P);  // (the gradient pass recomputes / reloads what it needs: bk_source_api.hpp)")
    if unroll:
        body.append("#pragma unroll")
    body.append(f"    for (i64 j = 0; j < {D}; ++j) {{")
    body.append("      double gj = 0.0;")
    for a, n, lines, r in blocks:
        cond = f"j >= {a} && j < {a + n}" if a > 0 else f"j < {n}"
        body.append(f"      if ({cond}) {{")
        body.append(f"        const i64 i = j - {a};")
        body.extend(lines)
        body.append(f"        gj = gj + {r};")
        body.append("      }")
    for i, name in sorted(gh.items()):
        body.append(f"      if (j == {i}) gj = gj + {name};")
    body.append("      g.set(j, gj);")
    if gen.heavy:
        body.append("      BK_CHAIN_PACE(j);")
    body.append("    }")
    body.append("  }")
    body.append(f"  return {val};")
    pace = ("// (an unrolled loop is handed to the scheduler four iterations at a time: interleaving all of them -- every exp, every\n"
            "// constant's load -- needs more registers than a lane has beside its staged coordinates)\n"
            "#ifndef BK_CHAIN_PACE\n#define BK_CHAIN_PACE(i) do { if (((i) & 3) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)\n"
            "// (... and between two passes the constants' pointer becomes an opaque value: their entries are loaded again, not kept)\n"
            "#define BK_CHAIN_FORGET(p) asm volatile(\"\" : \"+s\"(p))\n#endif\n")
    src = (pace +
           "// generated by bayes_kit_amd.trace_chain from a PyTorch log density (torch.fx graph, symbolic derivatives):\n"
           f"// {K} sum(s) over slices of theta, {n_extra} derivative sum(s), {len(blocks)} gradient block(s), "
           f"{len(heads)} coordinate(s) taken as scalars\n"
           "__device__ __forceinline__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 /*D*/, const double* P_) {\n"
           "  const double* P = P_;\n" + "\n".join(body) + "\n}\n")
    params = torch.cat(tr.consts) if tr.consts else None
    return src, params, {"sums": K, "derivative_sums": n_extra, "gradient_blocks": len(blocks), "scalars": len(heads),
                         "constants": len(tr.consts), "temporaries": gen.n}
