#!/usr/bin/env python3
"""Mutation fuzzing of the HOST verifier under AddressSanitizer + UBSan (CPU only; the verifier takes untrusted proofs).

Builds vx_verify.hip host-only with -fsanitize=address,undefined, loads it in a child interpreter started with the sanitizer runtime
preloaded, and feeds vx_stark_verify / vx_header_range_verify / vx_rotate_verify mutated proofs: seeds are real proofs of the small AIRs made by the CPU
reference prover (oracle/stark_ref.py -- test infrastructure, used here to make inputs only).  Mutations: word flips, random words,
truncation, extension, the degree-bits / length fields set to every small value, header words of a range blob set to extremes.
Any sanitizer report aborts the child: the script fails.  Every mutated proof must also be REJECTED (a flipped word that is accepted
would be a soundness bug) unless it is byte-identical to its seed.

usage: fuzz_verify_asan.py [iterations per seed]      (default 3000; about a minute)"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = "/tmp/libvxverify_asan.so"
SRC = os.path.join(ROOT, "0-kno-vectorx_amd", "csrc", "vx_verify.hip")  # every host verifier, no GPU code


def build():
    if os.path.exists(SO) and os.path.getmtime(SO) > os.path.getmtime(SRC):
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-fPIC", "-fvisibility=hidden", "-shared", "-o", SO, SRC], cwd=os.path.dirname(SRC))


def child(iters):
    import numpy as np

    sys.path.insert(0, ROOT)
    from oracle import stark_ref as S

    L = C.CDLL(SO)

    import vx_import

    vx = vx_import.load()
    StarkConfig = vx.lib.StarkConfig
    cfg = StarkConfig()
    for k, v in dict(S.DEFAULT_CFG, num_queries=8).items():
        setattr(cfg, k, v)
    ocfg = dict(S.DEFAULT_CFG, num_queries=8)
    vp, sz = C.c_void_p, C.c_size_t
    L.vx_stark_verify.argtypes = [C.POINTER(StarkConfig), vp, sz, C.c_int, vp, sz, C.c_char_p, sz]
    L.vx_header_range_verify.restype = C.c_int32
    L.vx_stark_verify.restype = C.c_int32
    err = C.create_string_buffer(256)

    def verify(words, air):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        return L.vx_stark_verify(C.byref(cfg), w.ctypes.data_as(vp), w.size, air, None, 0, err, 256)

    rng = np.random.default_rng(2026)
    total = rejected = 0
    for air, log_n in ((S.FibAir, 5), (S.MixAir, 6), (S.LookupAir, 8), (S.FibAir, 9)):
        tr, pub = air.trace(log_n)
        seed = S.prove(air, tr, pub, ocfg)
        assert verify(seed, air.ID) == 0, err.value
        n = seed.size
        for it in range(iters):
            p = seed.copy()
            kind = it % 8
            if kind == 0:
                p[rng.integers(n)] ^= np.uint64(1) << np.uint64(rng.integers(64))
            elif kind == 1:
                p[rng.integers(n)] = np.uint64(rng.integers(0, 2**63)) * np.uint64(2) + np.uint64(rng.integers(2))
            elif kind == 2:
                p = p[: rng.integers(0, n)]
            elif kind == 3:
                p = np.concatenate([p, rng.integers(0, 2**63, size=rng.integers(1, 40), dtype=np.uint64)])
            elif kind == 4:  # the header words (air id, degree bits, counts ...) set to small and extreme values
                p[rng.integers(0, min(16, n))] = np.uint64([0, 1, 2, 3, 26, 27, 63, 64, 2**32 - 1, 2**32, 2**63, 2**64 - 1][rng.integers(12)])
            elif kind == 5:
                k = rng.integers(1, 6)
                for _ in range(k):
                    p[rng.integers(n)] = np.uint64(2**64 - 1)
            elif kind == 6:
                a, b = sorted(rng.integers(0, n, size=2))
                p[a:b] = 0
            else:
                p = rng.integers(0, 2**63, size=rng.integers(0, 2 * n), dtype=np.uint64)
            rc = verify(p, air.ID)
            total += 1
            if p.size == n and (p == seed).all():
                continue
            assert rc != 0, f"mutation kind {kind} of a {air.__name__} proof was ACCEPTED"
            rejected += 1
    # range blobs: garbage behind a valid magic, extreme lengths
    L.vx_header_range_verify.argtypes = [C.POINTER(StarkConfig), vp, sz, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_char_p, sz]
    magic = int(vx.lib.HR_MAGIC)
    out96 = (C.c_uint8 * 96)()
    h32 = (C.c_uint8 * 32)()
    for it in range(iters):
        m = int(rng.integers(0, 4000))
        b = rng.integers(0, 2**63, size=m, dtype=np.uint64)
        if m > 22:
            b[0] = np.uint64(magic)
            b[1], b[2], b[3] = np.uint64(256), np.uint64(100), np.uint64(356)
            for q in range(16, 21):
                b[q] = np.uint64([0, 1, m, m - 22, (m - 22) // 5, 2**64 - 1, 2**63][rng.integers(7)])
        rc = L.vx_header_range_verify(C.byref(cfg), b.ctypes.data_as(vp), b.size, 256, 100, h32, 0, h32, 356, out96, err, 256)
        assert rc != 0
        total += 1
    # a real blob (tests/golden/circuit_blobs.npz, made on the GPU): five proofs behind a 23-word header (22 fixed words + one segment length)
    z = np.load(os.path.join(ROOT, "tests", "golden", "circuit_blobs.npz"), allow_pickle=False)
    cfg2 = StarkConfig()
    for k, v in dict(S.DEFAULT_CFG, num_queries=2).items():
        setattr(cfg2, k, v)
    hr = np.ascontiguousarray(z["hr_blob"], dtype=np.uint64)
    o96, th, sh = (np.ascontiguousarray(z[k]) for k in ("hr_out96", "hr_trusted_hash", "hr_set_hash"))
    tb, tg, sid = int(z["hr_trusted_block"]), int(z["hr_target_block"]), int(z["hr_set_id"])

    def hr_verify(b):
        b = np.ascontiguousarray(b, dtype=np.uint64)
        return L.vx_header_range_verify(C.byref(cfg2), b.ctypes.data_as(vp), b.size, 16, tb, th.ctypes.data_as(vp), sid, sh.ctypes.data_as(vp), tg, o96.ctypes.data_as(vp), err, 256)

    assert hr_verify(hr) == 0, err.value
    n = hr.size
    for it in range(max(iters // 4, 50)):
        p = hr.copy()
        kind = it % 6
        if kind == 0:
            p[rng.integers(n)] ^= np.uint64(1) << np.uint64(rng.integers(64))
        elif kind == 1:
            p[rng.integers(0, 23)] = np.uint64([0, 1, 21, 22, n, n - 22, 2**32, 2**63, 2**64 - 1][rng.integers(9)])
        elif kind == 2:
            p = p[: rng.integers(0, n)]
        elif kind == 3:
            a = int(rng.integers(22, n))
            p[a: a + int(rng.integers(1, 64))] = np.uint64(2**64 - 1)
        elif kind == 4:  # move a word between two length fields: the sum stays, the cut points move
            q = 16 + int(rng.integers(0, 4))
            d = np.uint64(rng.integers(1, 50))
            p[q], p[q + 1] = p[q] - d, p[q + 1] + d
        else:
            a, b_ = sorted(int(x) for x in rng.integers(22, n, size=2))
            p[a:b_] = rng.integers(0, 2**63, size=b_ - a, dtype=np.uint64)
        if p.size == n and (p == hr).all():
            continue
        assert hr_verify(p) != 0, f"mutation kind {kind} of the header_range blob was ACCEPTED"
        total += 1
        rejected += 1
    # ... and the rotate blob (six proofs, two buses) through vx_rotate_verify
    L.vx_rotate_verify.argtypes = [C.POINTER(StarkConfig), vp, sz, C.c_uint64, vp, vp, C.c_char_p, sz]
    L.vx_rotate_verify.restype = C.c_int32
    rot = np.ascontiguousarray(z["rot_blob"], dtype=np.uint64)
    o32, rsh, rid = np.ascontiguousarray(z["rot_out32"]), np.ascontiguousarray(z["rot_set_hash"]), int(z["rot_set_id"])

    def rot_verify(b):
        b = np.ascontiguousarray(b, dtype=np.uint64)
        return L.vx_rotate_verify(C.byref(cfg2), b.ctypes.data_as(vp), b.size, rid, rsh.ctypes.data_as(vp), o32.ctypes.data_as(vp), err, 256)

    assert rot_verify(rot) == 0, err.value
    n = rot.size
    for it in range(max(iters // 4, 50)):
        p = rot.copy()
        kind = it % 5
        if kind == 0:
            p[rng.integers(n)] ^= np.uint64(1) << np.uint64(rng.integers(64))
        elif kind == 1:
            p[rng.integers(0, 28)] = np.uint64([0, 1, 27, 28, n, n - 28, 510, 511, 35840, 2**32, 2**63, 2**64 - 1][rng.integers(12)])
        elif kind == 2:
            p = p[: rng.integers(0, n)]
        elif kind == 3:
            q = [16, 17, 18, 19, 24, 27][int(rng.integers(0, 6))]
            q2 = [16, 17, 18, 19, 24, 27][int(rng.integers(0, 6))]
            d = np.uint64(rng.integers(1, 50))
            if q != q2:
                p[q], p[q2] = p[q] - d, p[q2] + d
        else:
            a, b_ = sorted(int(x) for x in rng.integers(28, n, size=2))
            p[a:b_] = rng.integers(0, 2**63, size=b_ - a, dtype=np.uint64)
        if p.size == n and (p == rot).all():
            continue
        assert rot_verify(p) != 0, f"mutation kind {kind} of the rotate blob was ACCEPTED"
        total += 1
        rejected += 1
    # run-time AIR descriptors (vx_air_register + the host interpreter): random and mutated instruction streams must be refused or
    # registered without a sanitizer report; whatever registers is then run by the verifier on a real proof of another program
    # (it must reject: a different statement) and on that proof's mutations
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import air_programs as AP
    from oracle.air_program import ProgramAir

    AirProgramStruct = vx.lib.AirProgramStruct
    L.vx_air_register.argtypes = [C.POINTER(AirProgramStruct), C.POINTER(C.c_int), C.c_char_p, sz]
    L.vx_air_register.restype = C.c_int32

    def register(cols, n_pub, code, consts, periodic, n_regs):
        code, consts = np.ascontiguousarray(code, dtype=np.uint64), np.ascontiguousarray(consts, dtype=np.uint64)
        plog = np.array([len(c).bit_length() - 1 for c in periodic], dtype=np.uint8)
        pv = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint64) for c in periodic]) if periodic else np.zeros(1, np.uint64))
        st = AirProgramStruct(cols, n_pub, len(periodic), n_regs, plog.ctypes.data_as(vp) if periodic else None, pv.ctypes.data_as(vp) if periodic else None,
                              consts.ctypes.data_as(vp) if consts.size else None, consts.size, code.ctypes.data_as(vp) if code.size else None, code.size)
        aid = C.c_int(0)
        return L.vx_air_register(C.byref(st), C.byref(aid), err, 256), aid.value

    b = AP.cube_builder(vx.air_program)
    code, consts, n_regs = b.assemble()
    rc, cube_id = register(b.cols, b.n_public, code, consts, b.periodic, n_regs)
    assert rc == 0, err.value
    air = ProgramAir(cube_id, b.cols, b.n_public, code, consts, b.periodic)
    S.register_air(air)
    tr, pub = AP.cube_trace(6)
    seed = S.prove(air, tr, pub, ocfg)
    assert verify(seed, cube_id) == 0, err.value
    n, registered = seed.size, 0
    for it in range(iters):
        kind = it % 4
        c2 = code.copy()
        if kind == 0:  # one field of one instruction changed
            k = int(rng.integers(c2.size))
            c2[k] ^= np.uint64(1) << np.uint64(rng.integers(48))
        elif kind == 1:  # random words with plausible opcodes
            c2 = (rng.integers(1, 16, size=c2.size).astype(np.uint64) | (rng.integers(0, 40, size=c2.size).astype(np.uint64) << np.uint64(8))
                  | (rng.integers(0, 40, size=c2.size).astype(np.uint64) << np.uint64(16)) | (rng.integers(0, 40, size=c2.size).astype(np.uint64) << np.uint64(32)))
        elif kind == 2:
            c2 = rng.integers(0, 2**63, size=int(rng.integers(0, 64)), dtype=np.uint64)
        else:  # two instructions swapped: often still well-formed, a different statement
            i, j = (int(x) for x in rng.integers(0, c2.size, size=2))
            c2[i], c2[j] = c2[j], c2[i]
        rc, aid = register(b.cols, b.n_public, c2, consts, b.periodic, int(rng.integers(1, 34)) if kind == 2 else n_regs)
        total += 1
        if rc != 0:
            continue
        registered += 1
        p = seed.copy()
        p[1] = np.uint64(aid)
        same = c2.size == code.size and (c2 == code).all()
        rcv = verify(p, aid)
        assert (rcv == 0) == same or rcv == 0 and kind in (0, 3), f"cube proof under mutated program (kind {kind}): rc {rcv}"
        p[rng.integers(10, n)] ^= np.uint64(1) << np.uint64(rng.integers(64))
        assert verify(p, aid) != 0
    print(f"fuzz: {total} inputs, {rejected} mutated proofs rejected, {registered} mutated programs registered and run, no sanitizer report")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        build()
        rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")[0]
        env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        iters = sys.argv[1] if len(sys.argv) > 1 else "3000"
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child", iters], env=env))
