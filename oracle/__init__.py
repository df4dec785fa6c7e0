"""CPU oracle for the Groth16 prover hot path (TEST INFRASTRUCTURE ONLY).

This package is a big-integer / plain-C restatement of the algorithms on the
hot path of LayerXcom/zero-chain's prover (bellman 0.1.0 + pairing 0.14.2 as
used by core/proofs).  It exists to *check* the HIP product under
``zero-chain_amd/``; it is never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import or execute anything in here.  The product path
(``zero_chain_amd``) never imports it and fails loudly when its HIP library
is missing.

Parity status:
  * field / curve / encoding arithmetic: pinned against the reference's own
    golden vectors (core/pairing/src/bls12_381/tests/*.dat, fr.rs / fq.rs
    literal KATs, RELIC pairing vector, core/primitives/src/proof.rs:89).
  * Groth16 prover algebra: pinned against the DummyEngine known-answer test
    core/bellman-verifier/src/verifier.rs:74-92.
  * exact Transfer-circuit proof bytes: parity unpinned by the reference (no
    golden proof for a known (pk, witness, r, s) exists in tree, and the
    proving keys are missing blobs); pinned here three ways instead -
    trapdoor evaluation, this oracle's MSM/FFT prover, and pairing check.
"""
