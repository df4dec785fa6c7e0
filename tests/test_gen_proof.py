"""The host half of the gen_proof glue (zk_spending_key_from_seed, zk_transfer_derive: host code of libzkamd.so, no
GPU needed) against the oracle's restatement (oracle/gen_proof.py) and the one value the reference holds for the
derivation chain.  The GPU half (proof, ciphertexts, self-check, packing) is tests/test_gpu_parity.py."""
import pytest

from oracle import gen_proof as og
from oracle import jubjub as jj
from oracle import synth

ALICE_SEED = b"Alice" + b" " * 27                                               # modules/encrypted-balances/src/lib.rs:381
ALICE_ENC_KEY = "fd0c0c0183770c99559bf64df4fe23f77ced9b8b4d02826a282bcd125117dcc2"   # :443 pkd_addr_alice
BOB_ADDR = "45e66da531088b55dcb3b273ca825454d79d2d1d5c4fa2ba4a12c1fa1ccd6389"        # :383
G_EPOCH = "0953f47325251a2f479c25527df6d977925bebafde84423b20ae6c903411665a"         # :409


def _lib():
    import zero_chain_amd
    return zero_chain_amd.load_library()


def test_oracle_derivation_reproduces_the_reference_address():
    sk = og.spending_key_from_seed(ALICE_SEED)
    _, _, enc_key = og.derive(sk)
    assert jj.write_point(enc_key).hex() == ALICE_ENC_KEY


def reference_request(rng_seed=1):
    """The transfer of test_call_from_zface (lib.rs:372-420): Alice -> Bob, 100 -> 91, amount 8, fee 1, the balance
    encrypted with randomness Fs::one(), g_epoch of block height one."""
    sk = og.spending_key_from_seed(ALICE_SEED)
    _, _, enc_key = og.derive(sk)
    bal = og.encrypt(100, 1, enc_key)
    rng = synth.SplitMix64(rng_seed)
    return dict(amount=8, fee=1, remaining_balance=91, spending_key=sk, enc_key_recipient=bytes.fromhex(BOB_ADDR),
                enc_balance_left=jj.write_point(bal[0]), enc_balance_right=jj.write_point(bal[1]), g_epoch=bytes.fromhex(G_EPOCH),
                randomness=rng.field(jj.FS_MOD), alpha=rng.field(jj.FS_MOD)), bal


def test_product_derivation_matches_oracle_and_reference():
    import zero_chain_amd as zk
    lib = _lib()
    assert zk.spending_key_from_seed(ALICE_SEED, lib=lib) == og.spending_key_from_seed(ALICE_SEED)
    for seed in (b"", b"x", b"Bob" + b" " * 29, bytes(range(200))):
        assert zk.spending_key_from_seed(seed, lib=lib) == og.spending_key_from_seed(seed)
    items = []
    for k in range(3):
        rq, bal = reference_request(k)
        if k == 2:
            rq["spending_key"] = og.spending_key_from_seed(b"another key")
        items.append((rq, bal))
    sts, rsks = zk.transfer_derive(zk.transfer_requests([rq for rq, _ in items]), lib=lib)
    for (rq, bal), st, rsk in zip(items, sts, rsks):
        pgk, dec_key, enc_key = og.derive(rq["spending_key"])
        assert bytes(st.proof_generation_key) == jj.write_point(pgk)
        assert int.from_bytes(bytes(st.dec_key_sender), "little") == dec_key
        assert rsk == ((rq["spending_key"] + rq["alpha"]) % jj.FS_MOD).to_bytes(32, "little")
        assert (st.amount, st.fee, st.remaining_balance) == (rq["amount"], rq["fee"], rq["remaining_balance"])
        assert bytes(st.enc_key_recipient) == rq["enc_key_recipient"] and bytes(st.g_epoch) == rq["g_epoch"]
        assert bytes(st.enc_balance_left) == rq["enc_balance_left"] and bytes(st.enc_balance_right) == rq["enc_balance_right"]
    # Alice's statement carries the key whose encryption key is the reference's address
    _, dk, ek = og.derive(items[0][0]["spending_key"])
    assert jj.write_point(ek).hex() == ALICE_ENC_KEY and int.from_bytes(bytes(sts[0].dec_key_sender), "little") == dk
    bad = dict(items[0][0], spending_key=jj.FS_MOD)
    with pytest.raises(zk.ZkError) as e:
        zk.transfer_derive(zk.transfer_requests([items[1][0], bad]), lib=lib)
    assert e.value.variant == "InvalidArgument" and "request 1" in str(e.value) and "spending_key" in str(e.value)


# ---- the anonymous transfer (core/proofs/src/anonymous.rs:97-183)
def anonymous_request(seed, amount=10, balance=100):
    """A request shaped like the reference's test (anonymous.rs:439-490): twelve members, the sender's balance
    encrypted under its own key, random balances for everyone else."""
    rng = synth.SplitMix64(seed)
    g = jj.note_commitment_randomness_generator()
    fs = lambda: rng.field(jj.FS_MOD)
    sk = og.spending_key_from_seed(b"anonymous sender %d" % seed)
    _, _, enc_key_sender = og.derive(sk)
    s_index = rng.below(12)
    t_index = (s_index + 1 + rng.below(11)) % 12
    recipient = jj.mul(g, fs())
    decoys = [jj.mul(g, fs()) for _ in range(10)]
    rest = list(decoys)
    keys = [enc_key_sender if i == s_index else recipient if i == t_index else rest.pop(0) for i in range(12)]
    bals = []
    for i, y in enumerate(keys):
        value = balance if i == s_index else rng.below(1 << 32)
        rb = fs()
        bals.append((jj.add(jj.mul(g, value), jj.mul(y, rb)), jj.mul(g, rb)))
    w = jj.write_point
    rq = dict(amount=amount, remaining_balance=balance - amount, s_index=s_index, t_index=t_index, spending_key=sk,
              enc_key_recipient=w(recipient), enc_keys_decoy=[w(d) for d in decoys], enc_balances_left=[w(b[0]) for b in bals],
              enc_balances_right=[w(b[1]) for b in bals], g_epoch=w(jj.mul(g, fs())), randomness=fs(), alpha=fs())
    return rq, recipient, decoys, bals


def anonymous_expected(rq, recipient, decoys, bals):
    return og.gen_anonymous_xt_fields(rq["spending_key"], rq["amount"], rq["remaining_balance"], rq["s_index"], rq["t_index"], recipient,
                                      decoys, bals, jj.read_point(rq["g_epoch"]), rq["randomness"], rq["alpha"])


def test_anonymous_derivation_matches_oracle():
    """zk_anonymous_derive: the statement the product derives from a request (keys, the assembled set, the twelve left
    ciphertexts) against the oracle's restatement of anonymous.rs:113-150, and that statement satisfies the circuit."""
    import zero_chain_amd as zk
    from oracle import anonymous_circuit as ac
    lib = _lib()
    cases = [anonymous_request(1), anonymous_request(2, amount=0, balance=0), anonymous_request(3, amount=0xFFFFFFFE, balance=0xFFFFFFFE),
             anonymous_request(5, amount=3, balance=40)]
    assert cases[0][0]["s_index"] > cases[0][0]["t_index"] and cases[3][0]["s_index"] < cases[3][0]["t_index"]   # both insertion orders (anonymous.rs:119-125)
    sts, rsks = zk.anonymous_derive(zk.anonymous_requests([c[0] for c in cases]), lib=lib)
    for (rq, recipient, decoys, bals), st, rsk in zip(cases, sts, rsks):
        want, stmt = anonymous_expected(rq, recipient, decoys, bals)
        d = ac.statement_dict(stmt)
        assert (st.amount, st.remaining_balance, st.s_index, st.t_index) == (d["amount"], d["remaining_balance"], d["s_index"], d["t_index"])
        for name in ("randomness", "alpha", "dec_key"):
            assert int.from_bytes(bytes(getattr(st, name)), "little") == d[name], name
        for name in ("proof_generation_key", "g_epoch"):
            assert bytes(getattr(st, name)) == d[name], name
        for name in ("enc_keys", "left_ciphertexts", "enc_balances_left", "enc_balances_right"):
            assert [bytes(e) for e in getattr(st, name)] == d[name], name
        assert [bytes(e) for e in st.enc_keys] == want["enc_keys"] and [bytes(e) for e in st.left_ciphertexts] == want["left_ciphertexts"]
        assert rsk == want["rsk"]
    assert ac.synthesize(anonymous_expected(*cases[0])[1]).which_is_unsatisfied() is None
    rq = cases[0][0]
    for bad, what in ((dict(rq, t_index=rq["s_index"]), "s_index"), (dict(rq, s_index=12), "s_index"),
                      (dict(rq, spending_key=jj.FS_MOD), "spending_key"), (dict(rq, enc_key_recipient=bytes([0xff] * 32)), "enc_key_recipient")):
        with pytest.raises(zk.ZkError) as e:
            zk.anonymous_derive(zk.anonymous_requests([cases[1][0], bad]), lib=lib)
        assert e.value.variant == "InvalidArgument" and "request 1" in str(e.value) and what in str(e.value)


def test_jubjub_entries_match_oracle():
    """zk_jubjub_base_mul = scalar * NoteCommitmentRandomness generator (EncryptionKey::from_decryption_key, keys.rs:250-261),
    zk_elgamal_encrypt = Ciphertext::encrypt (no_std_aliases/elgamal.rs:46-63), against the oracle's big-integer Jubjub."""
    import zero_chain_amd as zk
    lib = _lib()
    g = jj.note_commitment_randomness_generator()
    rng = synth.SplitMix64(77)
    ks = [0, 1, 2, jj.FS_MOD - 1] + [rng.field(jj.FS_MOD) for _ in range(5)]
    assert zk.jubjub_base_mul(ks, lib=lib) == [jj.write_point(jj.mul(g, k)) for k in ks]
    vals = [0, 1, 100, 0xffffffff, 12345]
    rnd = [rng.field(jj.FS_MOD) for _ in vals]
    keys = [jj.mul(g, rng.field(jj.FS_MOD)) for _ in vals]
    left, right = zk.elgamal_encrypt(vals, rnd, [jj.write_point(k) for k in keys], lib=lib)
    for v, r, k, l, rr in zip(vals, rnd, keys, left, right):
        want = og.encrypt(v, r, k)
        assert (l, rr) == (jj.write_point(want[0]), jj.write_point(want[1]))
    with pytest.raises(zk.ZkError):
        zk.jubjub_base_mul([jj.FS_MOD], lib=lib)                      # not a canonical Fs scalar


def test_no_exception_crosses_the_c_abi(monkeypatch):
    """The host on the other side of include/zkamd.h is Rust or C: a C++ exception must come back as a status (every exported
    entry is a function-try-block, host_common.h ZK_ABI_CATCH), also when it is thrown on a worker thread (run_threads hands
    it to the joining thread).  ZKAMD_INJECT_THROW makes the last worker of zk_jubjub_base_mul fail an allocation."""
    import zero_chain_amd as zk
    from zero_chain_amd import _lib as loader
    lib = loader.ZkLib(loader.HOOKS_LIB_PATH)   # the injection is compiled into the hooks library only (host_common.h hook_env)
    monkeypatch.setenv("ZKAMD_INJECT_THROW", "1")
    for n in (1, 40):   # the calling thread itself / a spawned worker
        with pytest.raises(zk.ZkError) as e:
            zk.jubjub_base_mul(list(range(1, n + 1)), lib=lib)
        assert e.value.variant == "OutOfMemory" and "host allocation failed" in str(e.value)
    monkeypatch.delenv("ZKAMD_INJECT_THROW")
    g = jj.note_commitment_randomness_generator()
    assert zk.jubjub_base_mul([5], lib=lib) == [jj.write_point(jj.mul(g, 5))]   # and the library goes on working


def test_points_outside_the_prime_order_subgroup_are_refused():
    """The reference's typed inputs pass through as_prime_order (EncryptionKey::read keys.rs:269-276, Ciphertext::read
    elgamal.rs:117-133): P + (0, -1) = (-x, -y) decodes, lies on the curve and has order 2 s - the wallet-level entries
    refuse it, as the reference's readers do."""
    import zero_chain_amd as zk
    lib = _lib()
    g = jj.note_commitment_randomness_generator()
    x, y = jj.mul(g, 0x1234567)
    torsion = jj.write_point(((-x) % jj.R, (-y) % jj.R))
    good = jj.write_point((x, y))
    with pytest.raises(zk.ZkError) as e:
        zk.elgamal_encrypt([1], [5], [torsion], lib=lib)
    assert "prime-order" in str(e.value)
    zk.elgamal_encrypt([1], [5], [good], lib=lib)
    for field in ("enc_key_recipient", "enc_balance_left", "enc_balance_right", "g_epoch"):
        rq, _ = reference_request(3)
        zk.transfer_derive(zk.transfer_requests([rq]), lib=lib)       # the untouched request is fine
        rq[field] = torsion
        with pytest.raises(zk.ZkError) as e:
            zk.transfer_derive(zk.transfer_requests([rq]), lib=lib)
        assert e.value.variant == "InvalidArgument" and field in str(e.value) and "prime-order" in str(e.value)


def test_bench_statements_built_natively_equal_the_oracles():
    """bench.py builds its statements with the product's Jubjub entries; they are the oracle's statements."""
    import importlib.util, os
    import zero_chain_amd as zk
    from oracle import transfer_circuit as tc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("zk_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = bench.make_statements_native(zk, _lib(), 1021, 1027, check=2)
    for k, i in enumerate(range(1021, 1027)):
        seed, amount, fee, balance = bench.statement_params(i)
        assert rows[k] == tc.statement_dict(tc.make_witness(seed, amount=amount, fee=fee, balance=balance))
    zk.transfer_statements(rows)
