"""User-defined systems: the reference's subclass contract on the GPU.

Reference: python/class_files/systems/system_base.py:255-275 -- a user subclasses
``System``, writes ``_f_cont_fcn(self, x, u)`` (plus the costs) with ``jax.numpy`` math, and
the base class traces it with JAX and differentiates it (``jacfwd``, :203-219).  Here the
same method is written with ``sympy`` math (``from sympy import sin, cos``; everything else --
arithmetic, indexing ``x[0]``, python floats for the constants -- is unchanged).  It is traced
once with symbols, differentiated symbolically (``f_c``, ``df_c/dx``, ``df_c/du``), printed as
a ``Dyn`` struct into csrc/plugin_template.hip.in and compiled with hipcc against the SAME
kernel templates the built-in systems use (integrators and their chain-rule Jacobians, the
n_x = 4 / n_u = 1 DPP backward sweep, the ring rollout, the solver).  The resulting plugin is
loaded through ``ilqr_create_custom`` (include/ilqr_hip.h).

Scope: n_x <= 6, n_u <= n_x.  The cost is either the quadratic form every reference system uses
(pendulum_sys.py:77-98, given as ``x_target, Q, R, Q_f``) or the user's own ``_l_fcn`` / ``_l_f_fcn``
(system_base.py:262-275), traced and differentiated twice like the dynamics.
There is no interpreter fallback: without hipcc the plugin cannot be built and construction of
a handle raises.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

import numpy as np

from .. import _lib
from .system_base import System

PLUGIN_ROOT = os.path.join(_lib.HERE, "_plugins")
TEMPLATE = os.path.join(_lib.CSRC, "plugin_template.hip.in")
_KERNEL_HEADERS = ("dynamics.hpp", "kernels.hpp", "backward_tile16.hpp", "backward_tile16m2.hpp", "tile16m2_step_gen.inc", "backward_fused16.hpp",
                   "persistent.hpp", "forward_mfma16.hpp", "kernels_wave.hpp", "backward_mfma16.hpp", "fwd_in_gen.inc",
                   "solver.hpp", "plugin_template.hip.in", "check_ring_kernels.py", "verify_ring_isa.py")


def _ring_check():
    """csrc/check_ring_kernels.py (shared with the library's Makefile), loaded by path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ilqr_check_ring_kernels", os.path.join(_lib.CSRC, "check_ring_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _printer(n_x, n_u):
    from sympy.printing.c import C99CodePrinter

    class DevicePrinter(C99CodePrinter):
        """C99 printer whose literals are typed ``T(...)`` (no silent promotion of fp32 arithmetic to
        double) and whose math functions resolve to the plugin's ``um::`` overloads."""
        _ns = "um::"

        def _print_Float(self, e):
            return f"T({float(e)!r})"

        def _print_Rational(self, e):
            return f"T({int(e.p)}.0 / {int(e.q)}.0)"

        def _print_Integer(self, e):
            return f"T({int(e)})"

        def _print_NumberSymbol(self, e):
            return f"T({float(e.evalf(17))!r})"

        _print_Pi = _print_Exp1 = _print_NumberSymbol

        def _print_Pow(self, e):
            base, ex = e.base, e.exp
            b = self.parenthesize(base, 1000)
            if ex.is_Integer and 1 <= abs(int(ex)) <= 4:
                prod = " * ".join([b] * abs(int(ex)))
                return f"({prod})" if int(ex) > 0 else f"um::rcp({prod})"
            if ex == sp_half():
                return f"um::sqrt({self._print(base)})"
            if ex == -sp_half():
                return f"(T(1) / um::sqrt({self._print(base)}))"
            return f"um::pow({self._print(base)}, {self._print(ex)})"

    return DevicePrinter()


def sp_half():
    import sympy as sp
    return sp.Rational(1, 2)


def _trace_symbols(n_x, n_u):
    import sympy as sp
    xs = sp.symbols(f"x_0:{n_x}", real=True)
    us = sp.symbols(f"u_0:{n_u}", real=True)
    names = {**{s: f"x[{i}]" for i, s in enumerate(xs)}, **{s: f"u[{i}]" for i, s in enumerate(us)}}
    pr = _printer(n_x, n_u)
    pr._print_Symbol = lambda s: names.get(s, s.name)  # cse temporaries keep their own names

    def block(targets):
        exprs = [e for _, e in targets]
        # sin / cos of a SUM go through the addition theorems first (sin(q1 + q2) costs two multiply-adds once
        # sin q1, cos q1, sin q2, cos q2 are there, against a third range reduction and polynomial pair)
        sums = {f: sp.expand_trig(f) for e in exprs for f in e.atoms(sp.sin, sp.cos) if f.args[0].is_Add}
        if sums:
            exprs = [e.xreplace(sums) for e in exprs]
        # sines and cosines whose argument depends on x, u only are hoisted and evaluated as sin/cos PAIRS, two
        # angles per call (um::sincos2: one range reduction per angle, packed FP32 arithmetic for the two angles --
        # what the hand-written systems do); anything more exotic stays a plain um::sin / um::cos call
        args = []
        for e in exprs:
            for f in e.atoms(sp.sin, sp.cos):
                a = f.args[0]
                if a.free_symbols <= set(xs) | set(us) and not a.atoms(sp.sin, sp.cos) and a not in args:
                    args.append(a)
        args.sort(key=sp.default_sort_key)
        trig = {}
        for k, a in enumerate(args):
            trig[sp.sin(a)] = sp.Symbol(f"sn_{k}", real=True)
            trig[sp.cos(a)] = sp.Symbol(f"cs_{k}", real=True)
        exprs = [e.xreplace(trig) for e in exprs]
        lines = []
        if args:
            lines.append("        T " + ", ".join(f"sn_{k}, cs_{k}" for k in range(len(args))) + ";")
            for k in range(0, len(args) - 1, 2):
                lines.append(f"        um::sincos2({pr.doprint(args[k])}, {pr.doprint(args[k + 1])}, &sn_{k}, &cs_{k}, "
                             f"&sn_{k + 1}, &cs_{k + 1});")
            if len(args) % 2:
                k = len(args) - 1
                lines.append(f"        um::sincos({pr.doprint(args[k])}, &sn_{k}, &cs_{k});")
        repl, red = sp.cse(exprs, symbols=sp.numbered_symbols("w_"), optimizations="basic")
        lines += [f"        const T {pr.doprint(s)} = {pr.doprint(e)};" for s, e in repl]
        lines += [f"        {name} = {pr.doprint(e)};" for (name, _), e in zip(targets, red)]
        return "\n".join(lines)

    def check_free(exprs, what):
        free = set().union(*[e.free_symbols for e in exprs]) - set(xs) - set(us)
        if free:
            raise ValueError(f"{what} has free symbols other than x, u: {sorted(map(str, free))}")

    return xs, us, block, check_free


def generate_dyn_bodies(f_cont, n_x, n_u):
    """Trace ``f_cont(x, u)`` with sympy symbols; returns (f_body, fjac_body) C++ statement blocks."""
    import sympy as sp
    xs, us, block, check_free = _trace_symbols(n_x, n_u)
    out = f_cont(list(xs), list(us))
    out = [sp.sympify(e) for e in (out.tolist() if hasattr(out, "tolist") else list(out))]
    out = [e[0] if isinstance(e, (list, tuple)) else e for e in out]
    if len(out) != n_x:
        raise ValueError(f"_f_cont_fcn returned {len(out)} components, expected n_x = {n_x}")
    check_free(out, "_f_cont_fcn")
    nq = n_x // 2
    generate_dyn_bodies.second_order = n_x % 2 == 0 and all(sp.simplify(out[i] - xs[nq + i]) == 0 for i in range(nq))
    f_targets = [(f"xd[{i}]", e) for i, e in enumerate(out)]
    jac = list(f_targets)
    jac += [(f"Jx[{i}][{j}]", sp.diff(out[i], xs[j])) for i in range(n_x) for j in range(n_x)]
    jac += [(f"Ju[{i}][{j}]", sp.diff(out[i], us[j])) for i in range(n_x) for j in range(n_u)]
    return block(f_targets), block(jac)


def generate_cost_bodies(l_fcn, l_f_fcn, n_x, n_u):
    """Trace the user's stage and terminal cost; returns the four bodies (l, l derivatives, l_f, l_f derivatives).
    Derivative layout follows the reference (system_base.py:212-219): l_ux = d/dx (dl/du), shape (n_u, n_x)."""
    import sympy as sp
    xs, us, block, check_free = _trace_symbols(n_x, n_u)
    l = sp.sympify(l_fcn(list(xs), list(us)))
    lf = sp.sympify(l_f_fcn(list(xs)))
    check_free([l], "_l_fcn")
    check_free([lf], "_l_f_fcn")
    if lf.free_symbols & set(us):
        raise ValueError("_l_f_fcn must depend on x only")
    d = [(f"lx[{i}]", sp.diff(l, xs[i])) for i in range(n_x)] + [(f"lu[{j}]", sp.diff(l, us[j])) for j in range(n_u)]
    d += [(f"lxx[{i}][{j}]", sp.diff(l, xs[i], xs[j])) for i in range(n_x) for j in range(n_x)]
    d += [(f"lux[{j}][{i}]", sp.diff(l, us[j], xs[i])) for j in range(n_u) for i in range(n_x)]
    d += [(f"luu[{i}][{j}]", sp.diff(l, us[i], us[j])) for i in range(n_u) for j in range(n_u)]
    df = [(f"g[{i}]", sp.diff(lf, xs[i])) for i in range(n_x)]
    df += [(f"H[{i}][{j}]", sp.diff(lf, xs[i], xs[j])) for i in range(n_x) for j in range(n_x)]
    return block([("out", l)]), block(d), block([("out", lf)]), block(df)


def render_plugin_source(f_cont, n_x, n_u, dtype, l_fcn=None, l_f_fcn=None):
    f_body, fjac_body = generate_dyn_bodies(f_cont, n_x, n_u)
    if l_fcn is not None:
        cost = generate_cost_bodies(l_fcn, l_f_fcn, n_x, n_u)
    else:
        cost = ("        out = T(0);", "", "        out = T(0);", "")
    src = open(TEMPLATE).read()
    for key, val in (("@NX@", str(n_x)), ("@NU@", str(n_u)), ("@F_BODY@", f_body), ("@FJAC_BODY@", fjac_body),
                     ("@SECOND_ORDER@", "true" if generate_dyn_bodies.second_order else "false"),
                     ("@CUSTOM_COST@", "true" if l_fcn is not None else "false"),
                     ("@L_BODY@", cost[0]), ("@L_DERIVS_BODY@", cost[1]), ("@LF_BODY@", cost[2]),
                     ("@LF_DERIVS_BODY@", cost[3]),
                     ("@DTYPE@", "float" if np.dtype(dtype) == np.float32 else "double")):
        src = src.replace(key, val)
    return src


def build_plugin(source, verbose=False):
    """Compile a generated plugin for gfx950 (in-tree, keyed by source + kernel headers); returns the .so path."""
    h = hashlib.sha256(source.encode())
    for name in _KERNEL_HEADERS:
        h.update(open(os.path.join(_lib.CSRC, name), "rb").read())
    h.update(open(os.path.join(_lib.HERE, "..", "include", "ilqr_hip.h"), "rb").read())
    tag = h.hexdigest()[:16]
    d = os.path.join(PLUGIN_ROOT, tag)
    so = os.path.join(d, "ilqr_system_plugin.so")
    if os.path.exists(so):
        return so
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: a user-defined system needs it to compile its device code "
                           "(there is no interpreter / CPU fallback)")
    os.makedirs(d, exist_ok=True)
    hip = os.path.join(d, "plugin.hip")
    with open(hip, "w") as fh:
        fh.write(source)
    # every build works in a directory of its own (threads or ranks building the same system at the same time share
    # `d`, and the compiler's kept temporaries have fixed names); the finished library is moved into place atomically
    import tempfile
    work = tempfile.mkdtemp(prefix="build.", dir=d)
    tmp = os.path.join(work, "ilqr_system_plugin.so")
    crk = _ring_check()
    base = [hipcc, "-shared", "-fPIC", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fvisibility=hidden",
            "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj", "-I", _lib.CSRC,
            "-I", os.path.join(_lib.HERE, "..", "include"), "-o", tmp, hip]
    asm = os.path.join(work, "plugin-hip-amdgcn-amd-amdhsa-gfx950.s")

    mask = 0x1F
    for attempt in range(2):
        r = subprocess.run(base + [f"-DILQR_RING_INTEG_MASK={mask}"], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-8000:])
            shutil.rmtree(work, ignore_errors=True)
            raise RuntimeError(f"compiling the system plugin failed ({hip})")
        # The ring kernels count their own memory operations: they must not spill (check_ring_kernels.py) and nothing
        # may touch a register whose asm load can still be in flight (verify_ring_isa.py, on the compiler's assembly).
        # Integrators whose rollout violates either are routed to the compiler-scheduled kernel; compile once more.
        bad = [k["name"] for k in crk.violations(crk.parse(r.stderr))]
        bad += [k for k in crk.isa_violations(asm)[0] if k not in bad]
        # and no function may write a wide buffer store's data registers with a packed instruction right behind the store
        # (a gfx950 hazard hipcc does not pad; the library's own 16-byte stores carry their wait state)
        haz = crk.store_hazards(asm)
        if haz:
            shutil.rmtree(work, ignore_errors=True)
            raise RuntimeError("wide store followed by a packed write of its data registers in this plugin build: "
                               + "; ".join(f"{h[0]}: {h[2]} / {h[3]}" for h in haz[:4]))
        if verbose:
            print(f"plugin {tag}: ring mask {mask:#x}, rejected ring kernels: {bad}")
        if not bad:
            break
        integ = [crk.forward_ring_integrator(k) for k in bad]
        if attempt == 1 or any(i is None for i in integ):
            shutil.rmtree(work, ignore_errors=True)
            raise RuntimeError("a self-counted ring kernel spills or touches an in-flight register in this plugin build: "
                               + ", ".join(bad))
        for i in integ:
            mask &= ~(1 << i)
    with open(os.path.join(d, "ring_mask.txt"), "w") as fh:
        fh.write(f"{mask:#x}\n")
    os.replace(tmp, so)  # atomic: concurrent ranks building the same system never load a partial file
    shutil.rmtree(work, ignore_errors=True)
    return so


class SymbolicSystem(System):
    """Base class for user-defined systems.

    Subclass it, call ``super().__init__(n_x, n_u, dt, x_target, Q, R, Q_f, ...)`` and implement
    ``_f_cont_fcn(self, x, u) -> sequence of n_x expressions`` with sympy math.  All twelve public
    callables (``f_fcn`` ... ``l_f_xx_fcn``), ``iLQR(system, ...)``, ``solve`` and the MPC loop then work
    as for the built-in systems.

    Cost: either the quadratic form every reference system uses (give ``x_target, Q, R, Q_f``), or -- the full
    reference contract (system_base.py:262-275) -- override BOTH ``_l_fcn(self, x, u)`` and ``_l_f_fcn(self, x)``
    with sympy expressions; they are differentiated twice symbolically and compiled like the dynamics
    (``_l_fcn`` returns the stage cost exactly as written: multiply by ``self.dt`` yourself if you want the
    reference systems' convention, pendulum_sys.py:86).
    """

    SYSTEM_ID = _lib.SYS_CUSTOM

    def __init__(self, n_x, n_u, dt, x_target=None, Q=None, R=None, Q_f=None, use_jit=True, integrator="rk4",
                 dtype=np.float64):
        super().__init__(n_x, n_u, dt, use_jit=use_jit, integrator=integrator, dtype=dtype)
        if not (1 <= self.n_x <= 6 and 1 <= self.n_u <= self.n_x):
            raise ValueError("user-defined systems support 1 <= n_u <= n_x <= 6")
        own_l = type(self)._l_fcn is not SymbolicSystem._l_fcn
        own_lf = type(self)._l_f_fcn is not SymbolicSystem._l_f_fcn
        if own_l != own_lf:
            raise ValueError("override both _l_fcn and _l_f_fcn, or neither (quadratic cost)")
        self.custom_cost = own_l
        if self.custom_cost:
            n, m = self.n_x, self.n_u   # the quadratic block is unused: neutral placeholders
            x_target = np.zeros(n) if x_target is None else x_target
            Q, R, Q_f = (np.zeros((k, k)) if a is None else a for a, k in ((Q, n), (R, m), (Q_f, n)))
        elif any(a is None for a in (x_target, Q, R, Q_f)):
            raise ValueError("give x_target, Q, R, Q_f (quadratic cost) or override _l_fcn and _l_f_fcn")
        self._set_cost(x_target, Q, R, Q_f)
        self._plugins = {}

    def _f_cont_fcn(self, x, u):
        raise NotImplementedError("subclasses implement the continuous dynamics x_dot = f_c(x, u) (system_base.py:255)")

    def _l_fcn(self, x, u):
        raise NotImplementedError

    def _l_f_fcn(self, x):
        raise NotImplementedError

    def _system_params(self):
        return []  # the constants are part of the generated code

    def _bodies(self):
        b = generate_dyn_bodies(self._f_cont_fcn, self.n_x, self.n_u)
        if self.custom_cost:
            b += generate_cost_bodies(self._l_fcn, self._l_f_fcn, self.n_x, self.n_u)
        return b

    def same_dynamics(self, other):
        return super().same_dynamics(other) and self._bodies() == other._bodies()

    def plugin_source(self, dtype=None):
        return render_plugin_source(self._f_cont_fcn, self.n_x, self.n_u, self.dtype if dtype is None else dtype,
                                    self._l_fcn if self.custom_cost else None,
                                    self._l_f_fcn if self.custom_cost else None)

    def plugin_path(self, dtype=None, verbose=False):
        dt = np.dtype(self.dtype if dtype is None else dtype)
        if dt not in self._plugins:
            self._plugins[dt] = build_plugin(self.plugin_source(dt), verbose=verbose)
        return self._plugins[dt]

    def make_handle(self, horizon, batch, dtype=None, **kw):
        dt = self.dtype if dtype is None else dtype
        return _lib.Handle(system=self.SYSTEM_ID, n_x=self.n_x, n_u=self.n_u, horizon=horizon, batch=batch,
                           params=self.param_block(), dt=self.dt, integrator=self.integrator, dtype=dt,
                           plugin=self.plugin_path(dt), **kw)
