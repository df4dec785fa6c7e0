"""CPU suite: the kernel SOURCES of zero-chain_amd/csrc compiled for x86 (TEST-ONLY emulation,
csrc/gpu_rt.h) against the oracle.  Exercises the kernels' index math, the sort / accumulate /
reduce pipeline, the H pipeline and the C-ABI host logic at small sizes.  The product library
and the GPU itself are covered by the `-m gpu` suite."""
import os
import pytest
import numpy as np

import parity_cases as pc


def _host_alloc(lib):
    bufs = {}

    def alloc(nbytes):
        a = np.zeros(nbytes, dtype=np.uint8)
        bufs[a.ctypes.data] = a

        def upload(p, src):
            bufs[p][:] = src

        def download(p, n):
            return bufs[p][:n].tobytes()

        def free(p):
            bufs.pop(p, None)
        return a.ctypes.data, upload, download, free
    return alloc


def test_field_kats(emu_lib):
    pc.field_kats(emu_lib)


def test_ntt_all_four_transforms(emu_lib):
    pc.ntt_against_oracle(emu_lib, [0, 1, 2, 5, 9, 12])


def test_ntt_permutation_free_pair(emu_lib):
    pc.ntt_roundtrip_dev_orders(emu_lib, 10, _host_alloc(emu_lib))


def test_msm_g1_golden(emu_lib):
    pc.msm_golden_vectors(emu_lib, 1, 300, 5)
    pc.msm_golden_vectors(emu_lib, 1, 40, 2, seed=2)
    pc.msm_golden_vectors(emu_lib, 1, 3, 9, seed=3)


def test_msm_g2_golden(emu_lib):
    pc.msm_golden_vectors(emu_lib, 2, 60, 4)


def test_msm_edges(emu_lib):
    pc.msm_edge_cases(emu_lib)


def test_secret_workspaces_are_wiped_before_release(emu_lib):
    pc.memory_hygiene(emu_lib)


def test_msm_noncanonical_scalars(emu_lib):
    pc.msm_noncanonical_scalars(emu_lib, sizes=(20, 200), groups=(1,))
    pc.msm_noncanonical_scalars(emu_lib, sizes=(20,), groups=(2,))


def test_msm_decoder_refusals(emu_lib):
    pc.msm_decoder_refusals(emu_lib, oneshot_every=16)


def test_msm_oneshot_entries(emu_lib):
    pc.msm_oneshot(emu_lib, n1=150, n2=40)


def test_prover_device_pointers(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "5")
    pc.prover_device_pointers(emu_lib, _host_alloc(emu_lib))


def test_runtime_hooks(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "5")
    pc.runtime_hooks(emu_lib, on_gpu=False)


def test_prover_small_checked(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "5")
    pc.prover_small(emu_lib, 1, 3, 10, 12)


def test_prover_montgomery_inputs_two_pass_ntt(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "6")
    pc.prover_small(emu_lib, 7, 4, 300, 330, checked=False, montgomery=True)   # m = 512: two NTT passes


# A launch set takes the latency-optimised form up to 128 jobs (zkamd.cpp MSM_FEW_JOBS) - every batch the emulation can afford.
# ZKAMD_FEW_JOBS=1 sends the same batches through the throughput form (one workgroup per job in the LDS sort, light / heavy
# merges, the tree of reduction levels): what a 1024-proof chunk runs on the GPU.
LAUNCH_SET_FORMS = ("latency", "throughput")


@pytest.mark.parametrize("form", LAUNCH_SET_FORMS)
def test_prover_batch(emu_lib, monkeypatch, form):
    if form == "throughput":
        monkeypatch.setenv("ZKAMD_FEW_JOBS", "1")
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "5")
    monkeypatch.setenv("ZKAMD_BATCH_CHUNK", "2")
    pc.prover_batch(emu_lib, 4, 3, 12, 3)       # two device chunks: block 2 is staged beside block 1
    monkeypatch.setenv("ZKAMD_BATCH_CHUNK", "8")
    monkeypatch.setenv("ZKAMD_HOST_CHUNK", "1")
    pc.prover_batch(emu_lib, 5, 3, 12, 3)       # one device chunk staged in three blocks


@pytest.mark.parametrize("form", LAUNCH_SET_FORMS)
def test_prover_batch_split_g1_launch_sets(emu_lib, monkeypatch, form):
    """the A jobs and the C' jobs of a batch as two launch sets with their own recoding widths (zkamd.cpp prove_chunk)"""
    if form == "throughput":
        monkeypatch.setenv("ZKAMD_FEW_JOBS", "1")
    monkeypatch.setenv("ZKAMD_WINDOW_BITS_G1", "6")
    monkeypatch.setenv("ZKAMD_WINDOW_BITS_G1A", "4")
    monkeypatch.setenv("ZKAMD_WINDOW_BITS_G2", "5")
    monkeypatch.setenv("ZKAMD_SPLIT_MIN", "1")
    pc.prover_batch(emu_lib, 6, 3, 12, 3)
    monkeypatch.setenv("ZKAMD_G1A_STREAM", "main")
    pc.prover_batch(emu_lib, 8, 3, 12, 2)


def test_prover_errors(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "4")
    pc.prover_errors(emu_lib)


def test_empty_batches(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "4")
    pc.empty_batches(emu_lib)


def test_parsers_survive_mutations(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "4")
    pc.parsers_survive_mutations(emu_lib, rounds=16 if os.environ.get("ZKAMD_EMU_SANITIZED") else 64)


def test_params_subgroup_refusal(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "4")
    pc.params_subgroup_refusal(emu_lib)


def test_msm_recoding_all_widths(emu_lib):
    pc.msm_recoding_stress(emu_lib, windows=(2, 3, 5, 8, 12, 14))   # every width 2..22: GPU suite (the emulation is thread-per-GPU-thread)


def test_msm_two_level_sort_path(emu_lib, monkeypatch):
    """The two-level counting sort used when the bucket histogram of a job does not fit LDS: one
    coarse bin (degenerate), several bins, a ragged scalar count, both groups, a prover run."""
    monkeypatch.setenv("ZKAMD_NO_LDS_SORT", "1")
    pc.msm_golden_vectors(emu_lib, 1, 300, 5)                 # 8 buckets, one bin
    monkeypatch.setenv("ZKAMD_SORT_FINE_LOG", "3")
    pc.msm_golden_vectors(emu_lib, 1, 1300, 9, seed=8)        # 128 buckets in 16 bins, two workgroups of scalars
    pc.msm_golden_vectors(emu_lib, 2, 60, 6)                  # 16 buckets in 2 bins
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "7")
    pc.prover_small(emu_lib, 5, 3, 10, 12)


@pytest.mark.parametrize("form", LAUNCH_SET_FORMS)
def test_prover_from_witness(emu_lib, monkeypatch, form):
    if form == "throughput":
        monkeypatch.setenv("ZKAMD_FEW_JOBS", "1")
    pc.prover_from_witness(emu_lib, 3, 3, 14, 3)
    pc.prover_from_witness(emu_lib, 4, 2, 9, 2, montgomery=True)


def test_msm_long_tasks(emu_lib, monkeypatch):
    """Accumulation tasks of up to 256 points (the setting of large batches)."""
    monkeypatch.setenv("ZKAMD_MSM_SEG", "256")
    pc.msm_golden_vectors(emu_lib, 1, 1500, 6, seed=12)


def test_msm_sliced(emu_lib, monkeypatch):
    """A stand-alone multiexp cut into independent jobs over runs of the bases (the default from 2^14
    bases up), ragged last slice, a base at infinity (mapped positions), both groups."""
    monkeypatch.setenv("ZKAMD_MSM_SLICE", "128")
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "6")
    pc.msm_golden_vectors(emu_lib, 1, 300, 0, seed=14)
    pc.msm_golden_vectors(emu_lib, 2, 150, 0, seed=15)


def test_prover_blinding_edges(emu_lib, monkeypatch):
    monkeypatch.setenv("ZKAMD_WINDOW_BITS", "5")
    pc.prover_blinding_edges(emu_lib)


def test_verifier_pairing_relic(emu_lib):
    pc.verifier_pairing_relic(emu_lib)


def test_verifier_pvk_fixtures(emu_lib):
    pc.verifier_pvk_fixtures(emu_lib)


def test_verifier_small_circuit(emu_lib):
    pc.verifier_small_circuit(emu_lib)


def test_verifier_chunk_sizes(emu_lib):
    pc.verifier_chunk_sizes(emu_lib, sizes=(65,))


def test_fq_inverse_on_rows(emu_lib):
    pc.fq_inverse_on_rows(emu_lib)


def test_verifier_forms_agree(emu_lib):
    pc.verifier_forms_agree(emu_lib)


def test_verifier_rlc(emu_lib, monkeypatch, capfd):
    monkeypatch.setenv("ZKAMD_DEBUG_RLC", "1")
    pc.verifier_rlc(emu_lib, n=11, capfd=capfd)


def test_verifier_reference_vectors(emu_lib):
    pc.verifier_reference_vectors(emu_lib)


def test_verifier_golden_multiples(emu_lib):
    pc.verifier_golden_multiples(emu_lib)


def test_witness_gpu_matches_host(emu_lib):
    pc.witness_gpu_matches_host(emu_lib, n_extra=1)


def test_setup_matches_oracle(emu_lib):
    pc.setup_matches_oracle(emu_lib)


def test_msm_variable_base(emu_lib):
    pc.msm_variable_base(emu_lib, windows=(9,), n=60, g2_n=24, auto_n=40, g2_w=9, one_w=6)


def test_msm_variable_base_lane_merges(emu_lib, monkeypatch):
    """ZKAMD_COOP_L1_MAX=0: the launch sets of a variable-base multiexp merge their task partials with the lanes' kernels -
    eight lanes per listed bucket (k_msm_merge_medium), a workgroup (of rows) for the few with more than 64 partials, one lane
    for the unlisted - as the sets with many buckets do by default; narrow windows make every bucket heavy."""
    monkeypatch.setenv("ZKAMD_MERGE_SPLIT_MIN", "16")   # ... and the listed buckets with more than 16 partials by sixteen workgroups of rows
    pc.msm_variable_base(emu_lib, windows=(3,), n=700, g2_n=300, auto_n=40, g2_w=3, one_w=6)     # merge and level 1 on rows
    monkeypatch.setenv("ZKAMD_COOP_L1_MAX", "0")
    pc.msm_variable_base(emu_lib, windows=(3, 9), n=700, g2_n=300, auto_n=40, g2_w=3, one_w=6)
    monkeypatch.setenv("ZKAMD_COOP_TAIL", "0")
    pc.msm_variable_base(emu_lib, windows=(3,), n=700, g2_n=200, auto_n=40, g2_w=3, one_w=6)


def test_proof_reader_subgroup_tests(emu_lib):
    pc.proof_reader(emu_lib)


def test_verifier_one_thread_per_pair_kernels(emu_lib, monkeypatch):
    """ZKAMD_VERIFY_WIDE=0: the Miller loop with one thread per pair and the final exponentiation with one thread per
    proof (the kernels the key's e(alpha, beta) always takes) give the verdicts of the six-lanes-per-element kernels."""
    monkeypatch.setenv("ZKAMD_VERIFY_WIDE", "0")
    pc.verifier_golden_multiples(emu_lib)
    pc.verifier_small_circuit(emu_lib)
    pc.proof_reader(emu_lib)            # B's r-torsion test inside the decoder instead of at the end of the line preparation
    pc.verifier_skipped_pairs(emu_lib)


def test_verifier_skipped_pairs(emu_lib):
    pc.verifier_skipped_pairs(emu_lib)


def test_anonymous_witness_gpu_matches_host(emu_lib):
    pc.anonymous_witness_gpu_matches_host(emu_lib, n=2)
