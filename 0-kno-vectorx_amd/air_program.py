"""Writes constraint programs for vx_air_register (include/vx.h `vx_air_program`): the host-side half of the run-time AIR
descriptor.  An AIR is stated as expressions over the local / next row, periodic columns, public inputs and constants --
what a starky `Stark::eval_packed_generic` body says (starky v0.2.0 stark.rs; the reference reaches it through curta's
AirParser) -- and lowered to the library's straight-line register code:

    b = AirBuilder(cols=2, n_public=3)
    b.assert_first(b.loc(0) - b.pub(0))
    b.assert_transition(b.nxt(1) - b.loc(0) - b.loc(1))
    air_id = b.register()            # -> lib.air_register

Constraints are consumed in the order they are asserted (the order is protocol).  Inside one constraint a shared
sub-expression (the same Python object used twice) is evaluated once; nothing is shared between constraints."""
import numpy as np

P = 2**64 - 2**32 + 1
OP = dict(loc=1, nxt=2, per=3, pub=4, const=5, add=6, sub=7, mul=8, assert_zero=9, assert_transition=10, assert_first=11, assert_last=12, chal=13, apub=14)
MAX_REGS = 32


def insn(op, d=0, a=0, b=0):
    return op | (d << 8) | (a << 16) | (b << 32)


class Expr:
    __slots__ = ("op", "a", "b")

    def __init__(self, op, a=0, b=0):
        self.op, self.a, self.b = op, a, b

    @staticmethod
    def of(x):
        return x if isinstance(x, Expr) else Expr("const", int(x) % P)

    def __add__(self, o):
        return Expr("add", self, Expr.of(o))

    def __radd__(self, o):
        return Expr("add", Expr.of(o), self)

    def __sub__(self, o):
        return Expr("sub", self, Expr.of(o))

    def __rsub__(self, o):
        return Expr("sub", Expr.of(o), self)

    def __mul__(self, o):
        return Expr("mul", self, Expr.of(o))

    def __rmul__(self, o):
        return Expr("mul", Expr.of(o), self)


class X2:
    """An element a + b X of the quadratic extension F[X]/(X^2 - 7) as a pair of expressions: lookup arguments (logUp) work there,
    and the library's instruction set is base-field only, so the products are written out here (csrc/air.cuh X2<F> does the same
    for the compiled AIRs, in the same order of operations)."""

    def __init__(self, a, b=0):
        self.a, self.b = Expr.of(a), Expr.of(b)

    def __add__(self, o):
        if isinstance(o, X2):
            return X2(self.a + o.a, self.b + o.b)
        return X2(self.a + o, self.b)

    def __sub__(self, o):
        if isinstance(o, X2):
            return X2(self.a - o.a, self.b - o.b)
        return X2(self.a - o, self.b)

    def __mul__(self, o):
        if isinstance(o, X2):
            bb = self.b * o.b
            b2 = bb + bb
            b4 = b2 + b2
            return X2(self.a * o.a + (b4 + b4 - bb), self.a * o.b + self.b * o.a)  # 7 x = 8 x - x
        return X2(self.a * o, self.b * o)


class AirBuilder:
    def __init__(self, cols, n_public=0, periodic=(), aux_cols=0, n_challenges=0, n_aux_public=0):
        self.cols, self.n_public = cols, n_public
        self.aux_cols, self.n_challenges, self.n_aux_public = aux_cols, n_challenges, n_aux_public
        self.periodic = [[int(v) % P for v in col] for col in periodic]
        self.constraints = []

    def chal(self, i):
        return Expr("chal", i)

    def apub(self, i):
        return Expr("apub", i)

    def aux(self, j):
        """auxiliary column j of the local row (rows hold the main columns followed by the auxiliary ones)"""
        return Expr("loc", self.cols + j)

    def aux_nxt(self, j):
        return Expr("nxt", self.cols + j)

    def assert_zero_x2(self, e):
        self.assert_zero(e.a)
        self.assert_zero(e.b)

    def loc(self, col):
        return Expr("loc", col)

    def nxt(self, col):
        return Expr("nxt", col)

    def per(self, q):
        return Expr("per", q)

    def pub(self, i):
        return Expr("pub", i)

    def const(self, v):
        return Expr("const", int(v) % P)

    def assert_zero(self, e):
        self.constraints.append(("assert_zero", Expr.of(e)))

    def assert_transition(self, e):
        self.constraints.append(("assert_transition", Expr.of(e)))

    def assert_first(self, e):
        self.constraints.append(("assert_first", Expr.of(e)))

    def assert_last(self, e):
        self.constraints.append(("assert_last", Expr.of(e)))

    def assemble(self):
        """-> (code uint64[], consts uint64[], n_regs)"""
        code, consts, const_ix = [], [], {}
        n_regs = 0
        for kind, root in self.constraints:
            uses, order, seen = {}, [], set()

            def visit(e):
                uses[id(e)] = uses.get(id(e), 0) + 1
                if id(e) in seen:
                    return
                seen.add(id(e))
                if e.op in ("add", "sub", "mul"):
                    visit(e.a), visit(e.b)
                order.append(e)

            visit(root)  # uses[e] = reads of e: one per parent reference, and the assertion's read of the root
            free, reg = list(range(MAX_REGS - 1, -1, -1)), {}

            def release(e):
                uses[id(e)] -= 1
                if uses[id(e)] == 0:
                    free.append(reg[id(e)])

            for e in order:
                if e.op in ("add", "sub", "mul"):
                    ra, rb = reg[id(e.a)], reg[id(e.b)]
                    release(e.a), release(e.b)  # the destination may reuse an operand's register
                if not free:
                    raise ValueError("constraint needs more than %d registers: split it or share less" % MAX_REGS)
                d = free.pop()
                reg[id(e)] = d
                n_regs = max(n_regs, d + 1)
                if e.op == "const":
                    if e.a not in const_ix:
                        const_ix[e.a] = len(consts)
                        consts.append(e.a)
                    code.append(insn(OP["const"], d, const_ix[e.a]))
                elif e.op in ("loc", "nxt", "per", "pub", "chal", "apub"):
                    code.append(insn(OP[e.op], d, e.a))
                else:
                    code.append(insn(OP[e.op], d, ra, rb))
            code.append(insn(OP[kind], 0, reg[id(root)]))
        return np.array(code, dtype=np.uint64), np.array(consts, dtype=np.uint64), n_regs

    def register(self, gen_aux=None):
        from . import lib

        code, consts, n_regs = self.assemble()
        return lib.air_register(self.cols, self.n_public, code, consts, self.periodic, n_regs, self.aux_cols, self.n_challenges, self.n_aux_public, gen_aux)
