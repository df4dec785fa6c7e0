"""The hand-scheduled gfx950 product routines (zero-chain_amd/csrc/mul_asm.h) without a GPU: the committed header is
exactly what tools/gen_mul_asm.py generates, and every routine, interpreted instruction by instruction for one lane
by tools/sim_mul_asm.py (64-bit accumulators, SGPR constants, the register contract), gives the Montgomery products
of core/pairing/src/bls12_381/fr.rs:438-571 and fq.rs:915-1127 on random and extreme operands.  The GPU suite then
checks the same routines as the hardware executes them (field KATs, every parity test above them)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    import sys
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_header_is_the_generators_output(tmp_path, monkeypatch):
    g = _load("gen_mul_asm")
    monkeypatch.setattr(g, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "zero-chain_amd" / "csrc")
    g.main()
    fresh = open(tmp_path / "zero-chain_amd" / "csrc" / "mul_asm.h").read()
    assert fresh == open(os.path.join(ROOT, "zero-chain_amd", "csrc", "mul_asm.h")).read()


def test_every_routine_in_the_single_lane_interpreter(capsys):
    s = _load("sim_mul_asm")
    s.main()                  # FR, FQ (8 / 12 x 32-bit, fully reduced)
    s.main28()                # FQ28
    s.main28(dual=True)       # FQ28D
    s.main28_sqr()            # FQ28SQR
    s.main28_mac2()           # FQ28MAC2: x0 y0 + x1 y1, one reduction
    s.main28_fq2mul()         # FQ2MUL28: the fused Fq2 product
    s.main28_mul2()           # FQ28MUL2: two independent products, interleaved
    s.worst_case_limbs()      # column accumulators at the magnitude limits
    out = capsys.readouterr().out
    for name in ("FR ok", "FQ ok", "FQ28 ok", "FQ28D ok", "FQ28SQR ok", "FQ28MAC2 ok", "FQ2MUL28 ok", "FQ28MUL2 ok"):
        assert name in out


def test_madd_loop_header_is_the_generators_output(tmp_path, monkeypatch):
    g = _load("gen_madd_asm")
    monkeypatch.setattr(g, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "zero-chain_amd" / "csrc")
    g.main()
    fresh = open(tmp_path / "zero-chain_amd" / "csrc" / "madd_asm.h").read()
    assert fresh == open(os.path.join(ROOT, "zero-chain_amd", "csrc", "madd_asm.h")).read()


def test_madd_loop_in_the_single_lane_interpreter(capsys):
    """The generated G1 accumulation loop (madd_asm.h), run for one lane over whole tasks - loads retired only by
    s_waitcnt, EXEC masking, 64-bit column accumulators, limb-wise differences - against the affine group law
    (core/pairing/src/bls12_381/ec.rs:356-444 computes the same sums in Jacobian coordinates); equal and opposite
    points must come out flagged (ZZ == 0 mod p) for the second pass."""
    s = _load("sim_madd_asm")
    s.main(cases=10)
    s.main_g2(cases=5)        # the G2 loop: Fq2 products on two interleaved column streams, W and ZZZ parked in LDS
    out = capsys.readouterr().out
    assert "MADD_G1 ok" in out and "MADD_G2 ok" in out


def test_reduction_loop_header_is_the_generators_output(tmp_path, monkeypatch):
    g = _load("gen_red_asm")
    monkeypatch.setattr(g, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "zero-chain_amd" / "csrc")
    g.main()
    fresh = open(tmp_path / "zero-chain_amd" / "csrc" / "red_asm.h").read()
    assert fresh == open(os.path.join(ROOT, "zero-chain_amd", "csrc", "red_asm.h")).read()


def test_reduction_loop_in_the_single_lane_interpreter(capsys):
    """The generated level-1 loop of the G1 bucket reduction (red_asm.h): whole nodes for one lane - empty buckets in every
    position, the copy / add selection through EXEC, the flags - against the affine group law
    (core/pairing/src/bls12_381/ec.rs:356-444 computes the same sums in Jacobian coordinates); nodes that meet equal or
    opposite operands must come out flagged (ZZ == 0 mod p) for the compiled recomputation."""
    s = _load("sim_red_asm")
    s.main()
    assert "RED_G1 ok" in capsys.readouterr().out


def test_assembly_kernels_resources():
    """The kernels around the generated loops, read from the gfx950 code objects inside the built library
    (tools/kernel_resources.py; no GPU needed): the occupancy the loops were sized for (three waves per SIMD for G1: at
    most 168 registers; two for G2 and the reduction loop), no scratch memory in the G1 accumulation kernels, no more than
    the few words the compiler carries across the loops that own every VGPR in the FIRST form of the other two (G2
    accumulation 144 B, level 1 of the reduction 72 B per lane) - and NONE in their scratch-free second form (`_sf`), which
    zk_params_load selects on a device where the first form runs under a low scratch-wave limit."""
    so = os.path.join(ROOT, "zero-chain_amd", "libzkamd.so")
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    if not os.path.exists(so) or not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("library not built or llvm tools absent")
    res = _load("kernel_resources").kernel_resources(so)
    hot = {n: r for n, r in res.items() if any(k in n for k in ("k_msm_accumulate_g1asm", "k_msm_accumulate_g2asm", "k_msm_reduce1_g1asm"))}
    assert len(hot) == 8, sorted(hot)
    assert sum(1 for n in hot if "_sf" in n) == 3
    for name, r in hot.items():
        g1acc = "accumulate_g1asm" in name
        assert r["vgpr"] <= (168 if g1acc else 256), (name, r)
        assert r["scratch"] <= (0 if g1acc or "_sf" in name else 160), (name, r)
