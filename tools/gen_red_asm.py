#!/usr/bin/env python3
"""Generate zero-chain_amd/csrc/red_asm.h: level 1 of the bucket reduction of the G1 multiexp (msm.h, pass 6) as ONE
straight-line gfx950 assembly loop with a hand-made register allocation.

What it replaces: k_msm_suffix_buckets + the first k_msm_segsum - compiled XYZZ full additions
through the out-of-line product routines (~390 operand moves per addition, VALU issue 0.35-0.60), the suffix sums R_k
written to HBM by one kernel and read back by the next (7x the algorithmic traffic).  Here ONE thread walks the L
buckets of a node from the top down with BOTH accumulators in registers,

    run <- run + B_k            (R_k, the suffix sum)                      k = L-1 .. 0
    acc <- acc + run            (sum of R_k over k >= 1)                   k = L-1 .. 1

and hands back S = run = R_0 and A = acc; the level above forms W = 2 A + S (msm.h k_msm_level2_acc).  Nothing but
the bucket sums is read, nothing is written inside the loop.

Group law: EFD add-2008-s on extended Jacobian XYZZ coordinates (the reference's Jacobian law for the same group
elements: core/pairing/src/bls12_381/ec.rs:356-444), every product emitted in place by the generators of
tools/gen_madd_asm.py on the SIGNED lazily reduced radix-2^28 field (see there):

    U1 = X1 ZZ2    U2 = X2 ZZ1    S1 = Y1 ZZZ2    S2 = Y2 ZZZ1    P = U2 - U1    R = S2 - S1
    PP = P^2       PPP = P PP     Q = U1 PP       ZZ3 = ZZ1 ZZ2 PP               ZZZ3 = ZZZ1 ZZZ2 PPP
    X3 = R^2 - PPP - 2 Q          Y3 = R (Q - X3) + (-S1) PPP                    (12 M + 2 S, 13 reductions)

The first operand (the accumulator) is overwritten in place, the second one is only read; three scratch blocks (U2 /
P / PPP, S2 / R, Q / T) are the bucket's own registers for `run += B` and the - by then dead - bucket registers for
`acc += run`.  No carry pass is needed anywhere: differences feed products as they are.

The point at infinity never enters the formulas: three lane masks in SGPRs (run is infinity, acc is infinity, the
bucket is empty) select, per step, the lanes that COPY and the lanes that ADD through EXEC, and a wave-uniform branch
skips a part no lane needs.  Equal or opposite operands (P == 0 mod p) leave ZZ == 0 (mod p), which every later
product preserves: the C++ wrapper tests S and A once after the loop and recomputes a flagged node with the compiled
addition that knows every special case (as the accumulation loops do).  A value that was only ever copied is still in
the unsigned form it was loaded in: two more masks tell the wrapper which conversion a coordinate needs.

Everything emitted here is executed for one lane by tools/sim_red_asm.py against big-integer curve arithmetic
(tests/test_asm_routines.py).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_madd_asm as gm

ROOT = gm.ROOT
N = gm.N
v, vp, s = gm.v, gm.vp, gm.s

FLAG_RUN_INF, FLAG_ACC_INF, FLAG_RUN_RAW, FLAG_ACC_RAW = 1, 2, 4, 8
POINT_BYTES = 224          # XYZZ<Fq28>: x, y, zz, zzz, 14 limbs of 4 bytes each


class Regs:
    """240 VGPRs = two waves per SIMD: the two accumulators and the bucket sum (4 blocks of 16 registers each: 14 limbs
    + 2 pads that carry the loop state), one product temporary, the doubled limbs of a square, the quotient digits."""
    def __init__(self):
        blk = lambda b: list(range(b, b + 14))
        pt = lambda b: (blk(b), blk(b + 16), blk(b + 32), blk(b + 48))     # X, Y, ZZ, ZZZ
        self.RUN, self.ACCP, self.B = pt(0), pt(64), pt(128)
        self.T1 = blk(192)
        self.D = list(range(208, 221))
        self.M = list(range(224, 238))
        self.ACC = (238, 239)
        # per-lane loop state in the pad registers
        self.CNTP = (14, 15)        # &cnt[first bucket of the node]
        self.TOFFP = (30, 31)       # &toff[first bucket of the node]
        self.TSP = (46, 47)         # the job's first task partial
        self.NL = 62                # buckets per node (the same in every lane)
        self.FLAGS = 63             # out: FLAG_* bits
        self.ADDR = (78, 79)
        self.VCNT, self.VTOFF, self.TMP = 94, 95, 110
        self.n_vgpr = 240
        # SGPRs
        self.sP = list(range(36, 50))
        self.sINV, self.sMASK = 50, 51
        self.sDUMMY = "s[52:53]"
        self.sEXEC = "s[54:55]"
        self.sK, self.sSTRIDE = 56, 57
        self.sRUNINF, self.sACCINF = "s[60:61]", "s[62:63]"
        self.sT, self.sADD = "s[64:65]", "s[66:67]"
        self.sRUNRAW, self.sACCRAW = "s[68:69]", "s[70:71]"
        self.sBNZ = "s[72:73]"
        self.clob_s = list(range(36, 74))


def neg(e, a):
    for i in range(N):
        e.valu_op("v_sub_u32_e32 %s, 0, %s" % (v(a[i]), v(a[i])))


def copy_point(e, dst, src):
    for d, s_ in zip(dst, src):
        for i in range(N):
            e.valu_op("v_mov_b32_e32 %s, %s" % (v(d[i]), v(s_[i])))


def xadd(e, R, a, b, TX, TY, TQ):
    """a <- a + b (both XYZZ, neither infinity, a != +-b).  b is only read unless a scratch block IS one of its
    coordinates (run += B: TX, TY, TQ = B.X, B.Y, B.ZZ, each written when the coordinate is dead or in place)."""
    aX, aY, aZZ, aZZZ = a
    bX, bY, bZZ, bZZZ = b
    gm.mul(e, R, aX, bZZ, aX)           # U1 = X1 ZZ2      (in place over X1; X1 may carry limbs up to 3 * 2^28)
    gm.mul(e, R, bX, aZZ, TX)           # U2 = X2 ZZ1
    gm.mul(e, R, aY, bZZZ, aY)          # S1 = Y1 ZZZ2     (in place over Y1)
    gm.mul(e, R, bY, aZZZ, TY)          # S2 = Y2 ZZZ1
    gm.sub(e, TX, aX, TX)               # P = U2 - U1      |limb| < 2^28
    gm.sub(e, TY, aY, TY)               # R = S2 - S1
    gm.sqr(e, R, TX, R.T1)              # PP = P^2
    gm.mul(e, R, TX, R.T1, TX)          # PPP = P PP       (in place over P)
    gm.mul(e, R, aZZ, bZZ, aZZ)         # ZZ1 ZZ2          (in place)
    gm.mul(e, R, aZZ, R.T1, aZZ)        # ZZ3 = ZZ1 ZZ2 PP
    gm.mul(e, R, aX, R.T1, TQ)          # Q = U1 PP        (b.ZZ is dead where TQ is its block)
    gm.sqr(e, R, TY, aX)                # R^2 -> X registers (U1 is dead)
    gm.x3(e, aX, TX, TQ, aX, R.TMP)     # X3 = R^2 - PPP - 2 Q     limbs in (-3 * 2^28, 2^28), value in (-6 p, 2 p)
    gm.sub(e, TQ, aX, TQ)               # T = Q - X3               |limb| < 2^30
    neg(e, aY)                          # -S1
    gm.mac2(e, R, TY, TQ, aY, TX, aY)   # Y3 = R T + (-S1) PPP     (in place over S1)
    gm.mul(e, R, aZZZ, bZZZ, aZZZ)      # ZZZ1 ZZZ2        (in place)
    gm.mul(e, R, aZZZ, TX, aZZZ)        # ZZZ3 = ZZZ1 ZZZ2 PPP


def load_bucket(e, R):
    """the 224-byte partial sum ts[toff] -> the four B blocks (14 loads, asynchronous)"""
    a = vp(R.ADDR[0])
    e.valu_op("v_mad_u64_u32 %s, %s, %s, %s, %s" % (a, R.sDUMMY, v(R.VTOFF), s(R.sSTRIDE), vp(R.TSP[0])))
    for c, blk in enumerate(R.B):
        b, off = blk[0], 56 * c
        if off % 16 == 0:
            for q in range(3):
                e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (b + 4 * q, b + 4 * q + 3, a, off + 16 * q))
            e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:%d" % (b + 12, b + 13, a, off + 48))
        else:
            e.vmem_op("global_load_dwordx2 v[%d:%d], %s, off offset:%d" % (b, b + 1, a, off))
            for q in range(3):
                e.vmem_op("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (b + 2 + 4 * q, b + 5 + 4 * q, a, off + 8 + 16 * q))


def gen_loop():
    R = Regs()
    e = gm.Emitter()
    for j in range(N):
        e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sP[j]), gm.PL[j]))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sINV), gm.INV))
    e.salu_op("s_mov_b32 %s, 0x%08x" % (s(R.sMASK), gm.MASK))
    e.salu_op("s_movk_i32 %s, 0x%x" % (s(R.sSTRIDE), POINT_BYTES))
    e.salu_op("s_mov_b64 %s, exec" % R.sEXEC)
    e.salu_op("s_mov_b64 %s, exec" % R.sRUNINF)          # both accumulators start at infinity
    e.salu_op("s_mov_b64 %s, exec" % R.sACCINF)
    e.salu_op("s_mov_b64 %s, 0" % R.sRUNRAW)
    e.salu_op("s_mov_b64 %s, 0" % R.sACCRAW)
    e.valu_op("v_readfirstlane_b32 %s, %s" % (s(R.sK), v(R.NL)))      # uniform: every lane walks L buckets
    e.salu_op("s_sub_u32 %s, %s, 1" % (s(R.sK), s(R.sK)))
    e.label("1")
    # ---- bucket k of every lane's node: count and first partial, then the partial itself
    e.valu_op("v_mad_u64_u32 %s, %s, %s, 4, %s" % (vp(R.ADDR[0]), R.sDUMMY, s(R.sK), vp(R.CNTP[0])))
    e.vmem_op("global_load_dword %s, %s, off" % (v(R.VCNT), vp(R.ADDR[0])))
    e.valu_op("v_mad_u64_u32 %s, %s, %s, 4, %s" % (vp(R.ADDR[0]), R.sDUMMY, s(R.sK), vp(R.TOFFP[0])))
    e.vmem_op("global_load_dword %s, %s, off" % (v(R.VTOFF), vp(R.ADDR[0])))
    e.salu_op("s_waitcnt vmcnt(0)")
    load_bucket(e, R)
    e.valu_op("v_cmp_ne_u32_e64 %s, 0, %s" % (R.sBNZ, v(R.VCNT)), writes=[R.sBNZ])
    e.salu_op("s_and_b64 %s, %s, %s" % (R.sT, R.sBNZ, R.sRUNINF))         # lanes that take the bucket as their first point
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sADD, R.sBNZ, R.sRUNINF))     # lanes that add it
    e.salu_op("s_waitcnt vmcnt(0)")
    e.salu_op("s_mov_b64 exec, %s" % R.sT)
    e.salu_op("s_cbranch_execz 2f")
    copy_point(e, R.RUN, R.B)
    e.label("2")
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sRUNINF, R.sRUNINF, R.sT))
    e.salu_op("s_or_b64 %s, %s, %s" % (R.sRUNRAW, R.sRUNRAW, R.sT))
    e.salu_op("s_mov_b64 exec, %s" % R.sADD)
    e.salu_op("s_cbranch_execz 3f")
    xadd(e, R, R.RUN, R.B, R.B[0], R.B[1], R.B[2])                        # run += B
    e.label("3")
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sRUNRAW, R.sRUNRAW, R.sADD))
    # ---- acc += run for k >= 1
    e.salu_op("s_cmp_eq_u32 %s, 0" % s(R.sK))
    e.salu_op("s_cbranch_scc1 6f")
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sBNZ, R.sEXEC, R.sRUNINF))    # lanes whose run is a point by now
    e.salu_op("s_and_b64 %s, %s, %s" % (R.sT, R.sBNZ, R.sACCINF))         # ... and whose acc takes it as its first
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sADD, R.sBNZ, R.sACCINF))     # ... or adds it
    e.salu_op("s_mov_b64 exec, %s" % R.sT)
    e.salu_op("s_cbranch_execz 4f")
    copy_point(e, R.ACCP, R.RUN)
    e.label("4")
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sACCINF, R.sACCINF, R.sT))
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sACCRAW, R.sACCRAW, R.sT))    # acc is raw exactly where the copied run was
    e.salu_op("s_and_b64 %s, %s, %s" % (R.sT, R.sT, R.sRUNRAW))
    e.salu_op("s_or_b64 %s, %s, %s" % (R.sACCRAW, R.sACCRAW, R.sT))
    e.salu_op("s_mov_b64 exec, %s" % R.sADD)
    e.salu_op("s_cbranch_execz 5f")
    xadd(e, R, R.ACCP, R.RUN, R.B[0], R.B[1], R.B[2])                     # acc += run (the bucket's registers are scratch)
    e.label("5")
    e.salu_op("s_andn2_b64 %s, %s, %s" % (R.sACCRAW, R.sACCRAW, R.sADD))
    e.salu_op("s_mov_b64 exec, %s" % R.sEXEC)
    e.salu_op("s_sub_u32 %s, %s, 1" % (s(R.sK), s(R.sK)))
    e.salu_op("s_branch 1b")
    e.label("6")
    e.salu_op("s_mov_b64 exec, %s" % R.sEXEC)
    # ---- the lane's flags
    e.valu_op("v_cndmask_b32_e64 %s, 0, %d, %s" % (v(R.FLAGS), FLAG_RUN_INF, R.sRUNINF), reads=[R.sRUNINF])
    for flag, mask in ((FLAG_ACC_INF, R.sACCINF), (FLAG_RUN_RAW, R.sRUNRAW), (FLAG_ACC_RAW, R.sACCRAW)):
        e.valu_op("v_cndmask_b32_e64 %s, 0, %d, %s" % (v(R.TMP), flag, mask), reads=[mask])
        e.valu_op("v_or_b32_e32 %s, %s, %s" % (v(R.FLAGS), v(R.FLAGS), v(R.TMP)))
    return R, e


def body_counts(e):
    """instruction counts of one full step (both additions taken)"""
    i0 = e.lines.index("1:")
    i1 = e.lines.index("s_branch 1b")
    body = e.lines[i0 + 1:i1 + 1]
    valu = sum(1 for l in body if l.startswith("v_"))
    salu = sum(1 for l in body if l.startswith("s_"))
    vmem = sum(1 for l in body if l.startswith("global_"))
    return valu, salu, vmem


def render(R, e):
    valu, salu, vmem = body_counts(e)
    out = ["// Level 1 of the G1 bucket reduction: XYZZ full additions in the signed lazy radix-2^28 field, %d VGPRs." % R.n_vgpr,
           "// per step (both additions, no copy taken): %d VALU + %d SALU + %d VMEM instructions" % (valu - 2 * 4 * N, salu, vmem),
           "#define ZK_RED_G1_ASM \\"]
    for l in e.lines:
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    clob = ["v%d" % i for i in range(128, R.n_vgpr)] + ["s%d" % i for i in R.clob_s] + ["vcc", "scc", "memory"]
    out.append("#define ZK_RED_G1_ASM_CLOBBERS %s" % ", ".join('"%s"' % c for c in clob))
    used = gm.used_vgprs(e.lines)   # (see gen_madd_asm.render: the clobber list of the scratch-free form of the kernel)
    assert max(used) < R.n_vgpr
    clob_min = ["v%d" % i for i in range(128, R.n_vgpr) if i in used] + ["s%d" % i for i in R.clob_s] + ["vcc", "scc", "memory"]
    out.append("#define ZK_RED_G1_ASM_CLOBBERS_MIN %s" % ", ".join('"%s"' % c for c in clob_min))
    out.append("#define ZK_RED_G1_VGPRS %d" % R.n_vgpr)
    out.append("#define ZK_RED_FLAG_RUN_INF %d\n#define ZK_RED_FLAG_ACC_INF %d\n#define ZK_RED_FLAG_RUN_RAW %d\n#define ZK_RED_FLAG_ACC_RAW %d"
               % (FLAG_RUN_INF, FLAG_ACC_INF, FLAG_RUN_RAW, FLAG_ACC_RAW))
    return "\n".join(out) + "\n"


def main():
    R, e = gen_loop()
    path = os.path.join(ROOT, "zero-chain_amd", "csrc", "red_asm.h")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_red_asm.py - do not edit.\n#pragma once\n\n")
        f.write(render(R, e))
    valu, salu, vmem = body_counts(e)
    print("G1 reduction loop: %d lines, per step VALU %d (of which %d copies) SALU %d VMEM %d, hazard wait states %d"
          % (len(e.lines), valu, 2 * 4 * N, salu, vmem, e.nops))


if __name__ == "__main__":
    main()
