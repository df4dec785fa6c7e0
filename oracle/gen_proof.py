"""The wallet-level gen_proof glue - ORACLE, test infrastructure.

Restates core/proofs/src/confidential.rs:105-172, 282-361 (gen_proof, gen_xt) with the derivations it calls:
    SpendingKey::from_seed                      no_std_aliases/keys.rs:45-58  (Blake2b "zech_ExpandSeed_", to_uniform)
    ProofGenerationKey::from_spending_key       keys.rs:132-145
    into_decryption_key / into_encryption_key   keys.rs:166-198  (Blake2s "zech_bdk", five top bits dropped)
    PublicKey::randomize, SpendingKey::into_rsk keys.rs:60-66 (redjubjub: rvk = pgk + alpha G, rsk = sk + alpha)
    elgamal::Ciphertext::encrypt                elgamal.rs:46-63
    MultiCiphertexts::encrypt                   amount under both keys, fee under the sender's key, one randomness
Pinned on the one value the reference holds for this chain: the encryption key of the seed
b"Alice" + 27 spaces is fd0c0c01...dcc2 (modules/encrypted-balances/src/lib.rs:443, tests/test_gen_proof.py).
"""
import hashlib

from . import jubjub as jj


def spending_key_from_seed(seed):
    h = hashlib.blake2b(bytes(seed), digest_size=64, person=b"zech_ExpandSeed_").digest()
    return int.from_bytes(h, "little") % jj.FS_MOD


def derive(spending_key):
    """(pgk point, dec_key scalar, enc_key point)"""
    g = jj.note_commitment_randomness_generator()
    pgk = jj.mul(g, spending_key)
    d = bytearray(hashlib.blake2s(jj.write_point(pgk), digest_size=32, person=b"zech_bdk").digest())
    d[31] &= 0b00000111
    dec_key = int.from_bytes(d, "little")
    return pgk, dec_key, jj.mul(g, dec_key)


def encrypt(amount, randomness, enc_key):
    g = jj.note_commitment_randomness_generator()
    return jj.add(jj.mul(g, amount), jj.mul(enc_key, randomness)), jj.mul(g, randomness)


def gen_xt_fields(spending_key, amount, fee, remaining_balance, enc_key_recipient, enc_balance, g_epoch, randomness, alpha):
    """Everything of ConfidentialXt except the proof, plus the circuit statement (as transfer_circuit.TransferWitness)."""
    from .transfer_circuit import TransferWitness
    g = jj.note_commitment_randomness_generator()
    pgk, dec_key, enc_key_sender = derive(spending_key)
    rvk = jj.add(pgk, jj.mul(g, alpha))
    nonce = jj.mul(g_epoch, dec_key)
    left_s, right = encrypt(amount, randomness, enc_key_sender)
    left_r, _ = encrypt(amount, randomness, enc_key_recipient)
    left_fee, _ = encrypt(fee, randomness, enc_key_sender)
    w = jj.write_point
    fields = {"enc_key_sender": w(enc_key_sender), "enc_key_recipient": w(enc_key_recipient), "left_amount_sender": w(left_s),
              "left_amount_recipient": w(left_r), "left_fee": w(left_fee), "right_randomness": w(right),
              "rsk": ((spending_key + alpha) % jj.FS_MOD).to_bytes(32, "little"), "rvk": w(rvk),
              "enc_balance": w(enc_balance[0]) + w(enc_balance[1]), "nonce": w(nonce)}
    statement = TransferWitness(amount, remaining_balance, randomness, alpha, pgk, dec_key, enc_key_recipient, enc_balance, fee, g_epoch)
    return fields, statement


def gen_anonymous_xt_fields(spending_key, amount, remaining_balance, s_index, t_index, enc_key_recipient, enc_keys_decoy, enc_balances,
                            g_epoch, randomness, alpha):
    """core/proofs/src/anonymous.rs:97-183, 277-352: everything of AnonymousXt except the proof, plus the circuit
    statement (anonymous_circuit.AnonymousWitness).  The set: sender at s_index, recipient at t_index, decoys in order
    elsewhere (:117-126); MultiCiphertexts::<Anonymous>::encrypt (crypto_components.rs:168-220): -amount under the
    sender's key (neg_encrypt, elgamal.rs:66-82), +amount under the recipient's, zero under the decoys'."""
    from .anonymous_circuit import ANONIMITY_SIZE, AnonymousWitness
    g = jj.note_commitment_randomness_generator()
    pgk, dec_key, enc_key_sender = derive(spending_key)
    decoys = list(enc_keys_decoy)
    keys = [enc_key_sender if i == s_index else enc_key_recipient if i == t_index else decoys.pop(0) for i in range(ANONIMITY_SIZE)]
    neg = lambda p: ((-p[0]) % jj.R, p[1])
    amount_g = jj.mul(g, amount)
    left = []
    for i, y in enumerate(keys):
        ry = jj.mul(y, randomness)
        left.append(jj.add(neg(amount_g), ry) if i == s_index else jj.add(amount_g, ry) if i == t_index else ry)
    rvk = jj.add(pgk, jj.mul(g, alpha))
    w = jj.write_point
    fields = {"enc_keys": [w(k) for k in keys], "left_ciphertexts": [w(c) for c in left], "right_ciphertext": w(jj.mul(g, randomness)),
              "nonce": w(jj.mul(g_epoch, dec_key)), "rsk": ((spending_key + alpha) % jj.FS_MOD).to_bytes(32, "little"), "rvk": w(rvk)}
    statement = AnonymousWitness(amount, remaining_balance, s_index, t_index, randomness, alpha, pgk, dec_key, keys, left,
                                 list(enc_balances), g_epoch)
    return fields, statement
