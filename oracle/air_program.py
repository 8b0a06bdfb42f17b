"""TEST INFRASTRUCTURE (oracle): an independent reading of the constraint-program format of include/vx.h (`vx_air_program`) as an
AIR object for oracle/stark_ref.py -- prove / verify / check_trace run a program exactly as they run the restated AIRs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

The format stands in for starky's `Stark::eval_packed_generic` / `eval_ext_circuit` (starky v0.2.0 stark.rs; reached in the
reference through curta's AirParser, circuits/builder/header.rs:14-19).  One instruction per uint64: opcode in bits 0..7,
destination register 8..15, operand a 16..31, operand b 32..47.  Parity is pinned two ways: a program that restates FibAir / MixAir
must give the proofs of those AIRs (tests/test_air_program.py), and the GPU / C++ host interpreters must agree with this one."""
P = 2**64 - 2**32 + 1
LOC, NXT, PER, PUB, CONST, ADD, SUB, MUL, ASSERT, ASSERT_TRANSITION, ASSERT_FIRST, ASSERT_LAST, CHAL, APUB = range(1, 15)


class ProgramAir:
    def __init__(self, air_id, cols, n_public, code, consts=(), periodic=(), aux_cols=0, n_challenges=0, n_aux_public=0, gen_aux=None):
        """gen_aux: the auxiliary-round witness generator, (trace, challenges[, public inputs]) -> (aux columns, published words),
        as stark_ref.prove calls it; rows seen by the program hold the main columns followed by the auxiliary ones."""
        self.ID, self.COLS, self.PUB = air_id, cols, n_public
        self.AUX, self.CHAL, self.AUXPUB = aux_cols, n_challenges, n_aux_public
        if gen_aux is not None:
            self.gen_aux = gen_aux
        self.code = [int(w) for w in code]
        self.consts = [int(c) for c in consts]
        self._periodic = [[int(v) for v in col] for col in periodic]
        self.PERIODIC = len(self._periodic)
        self.PERIOD_LOGS = [len(col).bit_length() - 1 for col in self._periodic]
        self.PERIOD_LOG = max(self.PERIOD_LOGS, default=0)

    def periodic_values(self):
        return self._periodic

    def eval(self, loc, nxt, per, pub, c, chal=None, aux_pub=None):
        r = {}
        for w in self.code:
            op, d, a, b = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFFFF, (w >> 32) & 0xFFFF
            if op == LOC:
                r[d] = loc[a]
            elif op == NXT:
                r[d] = nxt[a]
            elif op == PER:
                r[d] = per[a]
            elif op == PUB:
                r[d] = pub[a]
            elif op == CONST:
                r[d] = self.consts[a]
            elif op == CHAL:
                r[d] = chal[a]
            elif op == APUB:
                r[d] = aux_pub[a]
            elif op == ADD:
                r[d] = r[a] + r[b]
            elif op == SUB:
                r[d] = r[a] - r[b]
            elif op == MUL:
                r[d] = r[a] * r[b]
            elif op == ASSERT:
                c.constraint(r[a])
            elif op == ASSERT_TRANSITION:
                c.transition(r[a])
            elif op == ASSERT_FIRST:
                c.first_row(r[a])
            elif op == ASSERT_LAST:
                c.last_row(r[a])
            else:
                raise ValueError("unknown opcode %d" % op)
