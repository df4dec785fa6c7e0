/* zkamd.h - C ABI of the MI355X-native Groth16 prover hot path (libzkamd.so).
 *
 * Drop-in boundary for the ONE call LayerXcom/zero-chain makes into bellman on the proving
 * path:   create_random_proof(instance, &self.proving_key, rng)
 *           /root/reference/core/proofs/src/confidential.rs:149  (and anonymous.rs:165)
 * plus the Parameters load next to it (confidential.rs:95-103, Parameters::read(.., true)).
 *
 * The reference has no FFI today (the seam is a Rust generic call into the un-vendored bellman
 * 0.1.0 crate).  The entry points below are what a Rust `core/proofs` would bind with
 * `extern "C"` (see INTEGRATION.md for the stub): the Rust side keeps Circuit::synthesize and
 * its ProvingAssignment (host, witness generation) and hands the assignment to zk_prove*, which
 * replaces bellman's EvaluationDomain pipeline, its eight multiexp calls and the final fold.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every byte buffer; the library owns device
 *     memory behind the opaque handles; no exceptions or panics cross the ABI.
 *   - scalars (Fr) are 32 bytes little-endian.  By default they are PLAIN integers < r
 *     (FrRepr::write_le order, core/pairing/src/bls12_381/fr.rs:57-58); with ZK_FR_MONTGOMERY they
 *     are the raw in-memory Montgomery limbs of `Fr` (fr.rs:246-247), i.e. a Rust Vec<Fr> can be
 *     passed without conversion.
 *   - points use the reference's encodings: uncompressed G1 96 B / G2 192 B big-endian
 *     (core/pairing/src/bls12_381/ec.rs:666-753, :1303-1427) in Parameters, and a Proof is the
 *     192 bytes of Proof::write (core/bellman-verifier/src/lib.rs:55-65): A (G1 compressed 48 B)
 *     || B (G2 compressed 96 B) || C (G1 compressed 48 B).
 *   - status codes 1..8 map 1:1 onto bellman's SynthesisError variants
 *     (mirrored in-tree at core/bellman-verifier/src/lib.rs:359-383).
 *   - one handle may be used from one host thread at a time (the reference caller is
 *     single-threaded); different handles - on the same or on different GPUs - may be used from
 *     different threads: streams live in a per-device context, the current device is selected on
 *     every entry.
 *   - no C++ exception leaves the library: every entry that returns a zk_status is an exception barrier
 *     (a failed host allocation is ZK_ERR_OUT_OF_MEMORY, anything else ZK_ERR_DEVICE with the text in
 *     zk_last_error()), also for work the library does on its own threads.
 *
 * Index: entry point -> the reference interface it replaces (paths under /root/reference; "bellman" = the un-vendored
 * bellman 0.1.0 crate, Cargo.lock:210-212, whose surface as the reference uses it is SURVEY.md 8(b)); "-" = no
 * counterpart in the reference (why it exists is said at the declaration).
 *   zk_params_load, zk_params_free  Parameters::read(reader, checked)             core/proofs/src/confidential.rs:99
 *   zk_params_get_info, zk_params_get_windows   - (fields of Parameters: lengths of the queries; the recoding widths chosen here)
 *   zk_params_write_vk              Parameters.vk, the input of prepare_verifying_key   core/proofs/src/setup.rs:31, 62
 *   zk_prove / zk_prove_batch       create_random_proof -> create_proof           confidential.rs:149, anonymous.rs:165
 *   zk_prove_batch_dev              the same, assignments already in HBM          confidential.rs:149
 *   zk_r1cs_load, zk_r1cs_free      the matrices a Circuit::synthesize enforces   circuit/confidential_transfer.rs:61-305
 *   zk_prove_batch_witness          create_proof; bellman's ProvingAssignment row evaluation
 *   zk_transfer_r1cs_load           structure half of ConfidentialTransfer::synthesize   circuit/confidential_transfer.rs:61-305
 *   zk_transfer_r1cs_fingerprint    TestConstraintSystem::hash / the pinned values       circuit/test.rs:228-251, confidential_transfer.rs:383-386
 *   zk_transfer_witness, zk_transfer_witness_gpu   value half of ConfidentialTransfer::synthesize       circuit/confidential_transfer.rs:29-41, 61-305
 *   zk_transfer_prove_batch         synthesize + create_random_proof              confidential.rs:134-149
 *   zk_pipeline_create, zk_pipeline_submit, zk_pipeline_wait, zk_pipeline_lanes, zk_pipeline_free
 *                                   the same for a stream of batches; - (the reference proves one transaction per call)
 *   zk_spending_key_from_seed       SpendingKey::from_seed                        core/proofs/src/no_std_aliases/keys.rs:45-58
 *   zk_jubjub_base_mul              EncryptionKey::from_decryption_key            no_std_aliases/keys.rs:250-261
 *   zk_elgamal_encrypt              elgamal::Ciphertext::encrypt                  no_std_aliases/elgamal.rs:46-63
 *   zk_transfer_derive              the derivations at the head of gen_proof      confidential.rs:105-133
 *   zk_transfer_gen_proof_batch     ProofBuilder::gen_proof (Confidential)        confidential.rs:105-172, check_proof :208-278
 *   zk_anonymous_r1cs_load          structure half of AnonymousTransfer::synthesize      circuit/anonymous_transfer.rs:56-337
 *   zk_anonymous_r1cs_fingerprint   TestConstraintSystem::hash (no pin in the reference) circuit/test.rs:228-251, anonymous_transfer.rs:446-451
 *   zk_anonymous_witness, zk_anonymous_witness_gpu   value half of AnonymousTransfer::synthesize          circuit/anonymous_transfer.rs:40-54, 56-337
 *   zk_anonymous_prove_batch        synthesize + create_random_proof              anonymous.rs:147-165
 *   zk_anonymous_derive             the derivations at the head of gen_proof      anonymous.rs:97-146
 *   zk_anonymous_gen_proof_batch    ProofBuilder::gen_proof (Anonymous)           anonymous.rs:97-183, check_proof :213-264
 *   zk_generate_parameters          generate_random_parameters                    core/proofs/src/setup.rs:28-31, 59-62
 *   zk_vk_prepare                   prepare_verifying_key                         core/bellman-verifier/src/verifier.rs:15-30
 *   zk_vk_read, zk_vk_write, zk_vk_free   PreparedVerifyingKey::read / write            core/bellman-verifier/src/lib.rs:175-244
 *   zk_vk_num_inputs                pvk.ic.len() - 1                              verifier.rs:38-40
 *   zk_verify_proof / zk_verify_batch   verify_proof                              verifier.rs:32-63; callers confidential.rs:271, modules/zk-system/src/lib.rs:57-108
 *   zk_verify_batch_rlc             - (bellman's batch verifier): the same verdicts, one final exponentiation
 *   zk_proof_read_batch             Proof::read                                   core/bellman-verifier/src/lib.rs:67-110
 *   zk_msm_create, zk_msm_create_variable, zk_msm_run, zk_msm_run_dev, zk_msm_free, zk_msm_g1, zk_msm_g2, zk_msm_cache_release
 *                                   bellman multiexp(FullDensity); group law core/pairing/src/bls12_381/ec.rs:296-526
 *   zk_ntt_fr, zk_ntt_create, zk_ntt_run_dev, zk_ntt_free
 *                                   bellman EvaluationDomain {fft, ifft, coset_fft, icoset_fft}; field core/pairing/src/bls12_381/fr.rs:341-571
 *   zk_debug_field_mul              Fr / Fq mul_assign (for the literal KATs)     fr.rs:438-464, fq.rs:915-965
 *   zk_strerror / zk_last_error     the variants of SynthesisError and their texts   core/bellman-verifier/src/lib.rs:359-383
 *   zk_device_count, zk_set_host_threads, zk_bind_host_to_device, zk_stream, zk_synchronize, zk_kernel_forms, zk_memory_stats,
 *   zk_profile_begin, zk_profile_get, zk_profile_end   - (device selection, host threads, measurement; the reference runs on the CPU)
 */
#ifndef ZKAMD_H
#define ZKAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t zk_status;

enum {
    ZK_OK = 0,
    ZK_ERR_ASSIGNMENT_MISSING = 1,          /* SynthesisError::AssignmentMissing */
    ZK_ERR_DIVISION_BY_ZERO = 2,            /* SynthesisError::DivisionByZero */
    ZK_ERR_UNSATISFIABLE = 3,               /* SynthesisError::Unsatisfiable */
    ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE = 4, /* SynthesisError::PolynomialDegreeTooLarge */
    ZK_ERR_UNEXPECTED_IDENTITY = 5,         /* SynthesisError::UnexpectedIdentity */
    ZK_ERR_IO = 6,                          /* SynthesisError::IoError (malformed / short pk) */
    ZK_ERR_MALFORMED_VERIFYING_KEY = 7,     /* SynthesisError::MalformedVerifyingKey */
    ZK_ERR_UNCONSTRAINED_VARIABLE = 8,      /* SynthesisError::UnconstrainedVariable */
    ZK_ERR_INVALID_ARGUMENT = 16,
    ZK_ERR_DEVICE = 17,                     /* HIP runtime error; see zk_last_error() */
    ZK_ERR_NO_DEVICE = 18,
    ZK_ERR_OUT_OF_MEMORY = 19
};

#define ZK_FR_MONTGOMERY 1u /* flag: scalars are raw Montgomery-form Fr limbs */

const char* zk_strerror(zk_status st);
/* Human-readable detail of the last failure on this thread (HIP error string, offending index). */
const char* zk_last_error(void);
zk_status zk_device_count(int* count);
/* Host threads the library may use for its CPU-side legs (witness calculation of zk_transfer_*,
 * proof encoding).  0 = default: ZKAMD_HOST_THREADS, else the cores of the process's affinity mask.
 * One process per GPU on a multi-GPU node should pass cores / ranks. */
void zk_set_host_threads(int n);

/* One process per GPU: restrict the CALLING thread - and every thread it starts afterwards, the
 * library's workers included - to the CPUs of the NUMA node the GPU hangs off (hipDeviceGetPCIBusId ->
 * /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/node<N>/cpulist, intersected with the mask the
 * process was given).  Call it first, before zk_params_load / zk_pipeline_create.  *numa_node_out = the node, or -1
 * when the platform reports none or its CPUs are not ours (the mask is then left as it was); *n_cpus_out = CPUs the
 * thread may run on afterwards.  No reference counterpart: the reference proves on the CPU in one process
 * (core/proofs/src/confidential.rs:149). */
zk_status zk_bind_host_to_device(int device, int* numa_node_out, int* n_cpus_out);

/* ------------------------------------------------------------------------------------------
 * Parameters  (bellman groth16::Parameters<Bls12>)
 * replaces: Parameters::read(reader, checked)   reference call: confidential.rs:99
 * Byte format (bellman Parameters::write): vk = alpha_g1 | beta_g1 | beta_g2 |
 * gamma_g2 | delta_g1 | delta_g2 | u32be n_ic | ic[] ; then u32be len | points for h, l, a,
 * b_g1 (G1) and b_g2 (G2).  checked != 0 additionally verifies on-curve and subgroup membership
 * of every point (on the GPU); both modes reject the point at infinity, as bellman does.
 * The bases are uploaded once, expanded into the table of all their doublings 2^k * P (k = 0..254;
 * ~4.2 GB for the transfer key) and stay resident in HBM.
 * ------------------------------------------------------------------------------------------ */
typedef struct zk_params zk_params;

typedef struct {
    uint32_t n_ic, n_h, n_l, n_a, n_b_g1, n_b_g2;
    uint32_t log_domain;  /* m = 2^log_domain = n_h + 1 */
    uint32_t window_bits; /* width c of the NAF recoding used for the G1 jobs of this key */
    uint32_t n_windows;   /* table slices per base (255: every doubling 2^k * P) */
    uint32_t device;
    uint64_t device_bytes; /* HBM held by the handle (tables + twiddles) */
} zk_params_info;

zk_status zk_params_load(const uint8_t* pk_bytes, size_t len, int checked, int device, zk_params** out);
zk_status zk_params_get_info(const zk_params* p, zk_params_info* info);
/* The recoding widths in use: out[0] = the C' jobs (H + L + r B1) of a batch, out[1] = its A jobs, out[2] = both G1 jobs
 * of a few proofs made alone (one launch set), out[3] = the G2 job (B2).  zk_params_info.window_bits is out[0]. */
zk_status zk_params_get_windows(const zk_params* p, uint32_t out[4]);
void zk_params_free(zk_params* p);

/* ------------------------------------------------------------------------------------------
 * Proving  (bellman groth16::create_proof(circuit, params, r, s) after synthesis)
 * replaces: create_random_proof / create_proof   reference call: confidential.rs:149
 * The assignment is exactly bellman's ProvingAssignment after `circuit.synthesize` and the
 * per-input `Input(i) * 0 = 0` rows: the row evaluations a, b, c (n_rows each), the input and
 * aux assignments, and the three density trackers (one byte per variable, 0 / 1).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n_rows;   /* constraints + n_inputs */
    uint32_t n_inputs; /* including ONE */
    uint32_t n_aux;
    uint32_t flags;    /* ZK_FR_MONTGOMERY */
    const uint8_t* a;  /* n_rows x 32 */
    const uint8_t* b;
    const uint8_t* c;
    const uint8_t* inputs;          /* n_inputs x 32 */
    const uint8_t* aux;             /* n_aux x 32 */
    const uint8_t* a_aux_density;   /* n_aux bytes */
    const uint8_t* b_input_density; /* n_inputs bytes */
    const uint8_t* b_aux_density;   /* n_aux bytes */
} zk_assignment;

/* r, s: 32-byte little-endian PLAIN scalars (the two Fr::rand draws of create_random_proof). */
zk_status zk_prove(zk_params* p, const zk_assignment* asg, const uint8_t r[32], const uint8_t s[32],
                   uint8_t proof_out[192]);

/* n independent proofs of the same circuit (same shape and densities).  rs: n x 64 bytes
 * (r then s per proof).  proofs_out: n x 192 bytes. */
zk_status zk_prove_batch(zk_params* p, size_t n, const zk_assignment* asgs, const uint8_t* rs,
                         uint8_t* proofs_out);

/* Same, with the assignments already resident in HBM (device pointers, plain scalars unless
 * ZK_FR_MONTGOMERY): a, b, c are [n][n_rows][32]; wit is [n][n_inputs + n_aux][32];
 * rs is a HOST pointer.  Densities are host byte arrays shared by the whole batch. */
typedef struct {
    uint32_t n_rows, n_inputs, n_aux, flags;
    const void* d_a;
    const void* d_b;
    const void* d_c;
    const void* d_wit;
    const uint8_t* a_aux_density;
    const uint8_t* b_input_density;
    const uint8_t* b_aux_density;
} zk_batch_dev;
zk_status zk_prove_batch_dev(zk_params* p, size_t n, const zk_batch_dev* batch, const uint8_t* rs,
                             uint8_t* proofs_out);

/* ------------------------------------------------------------------------------------------
 * Proving from the variable assignment alone (row f-1 of the hot-path scope).
 * bellman's ProvingAssignment evaluates every constraint row on the host while the circuit is
 * synthesized (a_j = <A_j, z>, b_j, c_j).  The R1CS of a circuit is fixed,
 * so it can live on the GPU: zk_r1cs_load takes the three matrices in CSR form once, and
 * zk_prove_batch_witness takes only z = (inputs | aux) per proof - a quarter of the bytes - and
 * computes a, b, c with a sparse matrix-vector product on the device.  The density trackers and
 * the per-input rows `Input(i) * 0 = 0` are derived from the matrices exactly as bellman's prover
 * derives them.  Coefficients: 32 bytes little-endian plain integers < r.
 * ------------------------------------------------------------------------------------------ */
typedef struct zk_r1cs zk_r1cs;
typedef struct {
    const uint32_t* row_ptr; /* n_constraints + 1 */
    const uint32_t* col;     /* variable index: input i -> i, aux j -> n_inputs + j */
    const uint8_t* coeff;    /* nnz x 32 */
} zk_csr;
zk_status zk_r1cs_load(uint32_t n_inputs, uint32_t n_aux, uint32_t n_constraints, const zk_csr* a, const zk_csr* b,
                       const zk_csr* c, int device, zk_r1cs** out);
void zk_r1cs_free(zk_r1cs* r);
/* The constraint system of the reference's confidential-transfer circuit, emitted natively by the library (the
 * structure half of ConfidentialTransfer::synthesize, core/proofs/src/circuit/confidential_transfer.rs:61-305):
 * zk_transfer_r1cs_load = zk_r1cs_load of those matrices.  zk_transfer_r1cs_fingerprint returns the counts and
 * the blake2s hash of the normalised system as the reference's circuit test defines it
 * (core/proofs/src/circuit/test.rs:228-251); the reference pins 19 974 constraints, 23 inputs and
 * d23c92fb60ee547d45118e160679929cfa186957280673af62f09fa12d401784 (confidential_transfer.rs:383-386). */
zk_status zk_transfer_r1cs_load(int device, zk_r1cs** out);
zk_status zk_transfer_r1cs_fingerprint(uint8_t hash_out[32], uint32_t* n_inputs, uint32_t* n_aux, uint32_t* n_constraints);
/* witness: n x (n_inputs + n_aux) x 32 bytes on the HOST (plain, or Montgomery limbs with
 * ZK_FR_MONTGOMERY in flags); rs: n x 64; proofs_out: n x 192. */
zk_status zk_prove_batch_witness(zk_params* p, zk_r1cs* circuit, size_t n, const uint8_t* witness, uint32_t flags,
                                 const uint8_t* rs, uint8_t* proofs_out);

/* ------------------------------------------------------------------------------------------
 * Native witness calculator of the reference's confidential-transfer circuit (row a3 / f-1).
 * replaces, value-wise: ConfidentialTransfer::synthesize under bellman's ProvingAssignment
 *   core/proofs/src/circuit/confidential_transfer.rs:61-305 (+ range_check.rs, utils.rs and the
 *   sapling-crypto gadgets).  The statement is the ten private values of the circuit
 *   (confidential_transfer.rs:29-41): u32 amounts, Fs scalars as 32 bytes little-endian, Jubjub
 *   points in the reference's 32-byte encoding (core/jubjub/src/curve/edwards.rs:92-117, 190-206).
 * zk_transfer_witness writes z = (23 inputs | 19 955 aux) per statement, in the reference's
 * variable order (host, multithreaded); zk_transfer_prove_batch = witness + row evaluations on
 * the GPU + create_proof, for the R1CS loaded with zk_r1cs_load.
 * ------------------------------------------------------------------------------------------ */
#define ZK_TRANSFER_N_INPUTS 23u
#define ZK_TRANSFER_N_AUX 19955u
typedef struct {
    uint32_t amount, remaining_balance, fee, reserved;
    uint8_t randomness[32], alpha[32], dec_key_sender[32];
    uint8_t proof_generation_key[32], enc_key_recipient[32], enc_balance_left[32], enc_balance_right[32], g_epoch[32];
} zk_transfer_statement;
/* witness_out: n x (23 + 19955) x 32 bytes; plain little-endian, or Montgomery limbs with ZK_FR_MONTGOMERY */
zk_status zk_transfer_witness(const zk_transfer_statement* st, size_t n, uint32_t flags, uint8_t* witness_out);
zk_status zk_transfer_prove_batch(zk_params* p, zk_r1cs* circuit, size_t n, const zk_transfer_statement* st,
                                  const uint8_t* rs, uint8_t* proofs_out);
/* zk_transfer_prove_batch and zk_pipeline compute the witnesses ON THE GPU (one thread per statement and gadget,
 * the assignment never leaves HBM) - except for a handful of statements (n <= 8 x host threads: one transaction at a time,
 * the reference's call pattern), whose assignments the native host calculator computes faster than the kernels' 8.4 ms of
 * serial chains; ZKAMD_WITNESS = host | gpu forces an engine, the proof bytes do not depend on it.  This entry returns
 * what that generator produces - same format as zk_transfer_witness - so that the two can be compared. */
zk_status zk_transfer_witness_gpu(zk_r1cs* circuit, const zk_transfer_statement* st, size_t n, uint32_t flags,
                                  uint8_t* witness_out);

/* A stream of statement batches: submit() queues a batch and returns at once; worker threads (one per lane, two lanes
 * by default) enqueue the witness kernels of a chunk beside the proving of the chunk before it, so two chunks are in flight
 * (ZKAMD_WITNESS=host: a producer thread computes the witnesses on the host cores, zk_set_host_threads, instead);
 * wait() blocks until everything submitted so far is proved and returns the first failure, if any (the
 * statement index in its message is relative to the failing submit).  The caller's buffers (statements, rs,
 * proofs_out) must stay valid until wait() returns, and `p` / `circuit` must not be used by other calls
 * while batches are in flight.  Batches larger than a device chunk (1024) are cut into chunks. */
typedef struct zk_pipeline zk_pipeline;
zk_status zk_pipeline_create(zk_params* p, zk_r1cs* circuit, zk_pipeline** out);
zk_status zk_pipeline_submit(zk_pipeline* pl, size_t n, const zk_transfer_statement* st, const uint8_t* rs,
                             uint8_t* proofs_out);
zk_status zk_pipeline_wait(zk_pipeline* pl);
/* chunks proved concurrently (ZKAMD_PIPELINE_LANES, default 2; fewer when the device's free memory does not hold the
 * workspaces of that many lanes - about 36 MB per proof of a chunk and lane) */
int zk_pipeline_lanes(const zk_pipeline* pl);
void zk_pipeline_free(zk_pipeline* pl);

/* ------------------------------------------------------------------------------------------
 * gen_proof: the reference's wallet-level entry (ProofBuilder::gen_proof, core/proofs/src/confidential.rs:
 * 105-172) for a batch of transfers.  Per request: the proof generation key G * sk, the decryption key
 * Blake2s("zech_bdk", pgk) with its five top bits dropped and the sender's encryption key
 * (no_std_aliases/keys.rs:132-198), rvk = pgk + alpha G, nonce = dec_key * g_epoch, the proof, the ElGamal
 * ciphertexts of amount and fee under both keys (elgamal.rs:46-63), check_proof against the prepared verifying key
 * (confidential.rs:208-278; ZK_ERR_UNSATISFIABLE if a proof does not verify, as the reference) and the packing of
 * ConfidentialXt (gen_xt :282-354, the struct :358-370), rsk = sk + alpha.  randomness / alpha are the two Fs::rand draws of gen_proof,
 * rs (n x 64 bytes) the two Fr::rand draws of create_random_proof; scalars 32 bytes little-endian, points in the
 * 32-byte Jubjub encoding.
 * zk_transfer_derive is the host half alone (request -> statement and rsk); zk_spending_key_from_seed is
 * SpendingKey::from_seed (keys.rs:45-58).  The points of a REQUEST (recipient key, encrypted balance, g_epoch) pass
 * through as_prime_order as the reference's typed readers do (keys.rs:269-276, elgamal.rs:117-133): a point outside the
 * prime-order subgroup is ZK_ERR_INVALID_ARGUMENT.  The points of a STATEMENT (zk_transfer_statement, the circuit
 * instance itself, whose fields are Point<E, PrimeOrder> by type in the reference) are decoded and curve-checked only.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t amount, fee, remaining_balance, reserved;
    uint8_t spending_key[32];
    uint8_t enc_key_recipient[32], enc_balance_left[32], enc_balance_right[32], g_epoch[32];
    uint8_t randomness[32], alpha[32];
} zk_transfer_request;
typedef struct {   /* ConfidentialXt, confidential.rs:358-370 */
    uint8_t proof[192];
    uint8_t enc_key_sender[32], enc_key_recipient[32], left_amount_sender[32], left_amount_recipient[32], left_fee[32],
        right_randomness[32], rsk[32], rvk[32], enc_balance[64], nonce[32];
} zk_confidential_xt;
zk_status zk_spending_key_from_seed(const uint8_t* seed, size_t len, uint8_t spending_key_out[32]);
/* The two Jubjub operations every key and ciphertext a wallet hands to gen_proof is made of (so a batch of requests can
 * be built without the reference's host code):
 * zk_jubjub_base_mul: scalar * FixedGenerators::NoteCommitmentRandomness, the generator of the reference's keys and
 *   ciphertexts; with the decryption key this is EncryptionKey::from_decryption_key (no_std_aliases/keys.rs:250-261).
 *   scalars: n x 32 bytes little-endian canonical Fs; points_out: n x 32 bytes (edwards::Point::write).
 * zk_elgamal_encrypt: elgamal::Ciphertext::encrypt (no_std_aliases/elgamal.rs:46-63): left = value G + randomness
 *   enc_key, right = randomness G.  enc_keys pass through as_prime_order as EncryptionKey::read does (keys.rs:269-276). */
zk_status zk_jubjub_base_mul(const uint8_t* scalars, size_t n, uint8_t* points_out);
zk_status zk_elgamal_encrypt(const uint32_t* values, const uint8_t* randomness, const uint8_t* enc_keys, size_t n,
                             uint8_t* left_out, uint8_t* right_out);
zk_status zk_transfer_derive(const zk_transfer_request* req, size_t n, zk_transfer_statement* statements_out, uint8_t* rsk_out);
struct zk_vk;
zk_status zk_transfer_gen_proof_batch(zk_params* p, zk_r1cs* circuit, struct zk_vk* vk, size_t n, const zk_transfer_request* req,
                                      const uint8_t* rs, zk_confidential_xt* out);

/* The value half of AnonymousTransfer::synthesize (core/proofs/src/circuit/anonymous_transfer.rs:56-337,
 * anonimity_set.rs; ANONIMITY_SIZE = 12, constants.rs:1): the private values of the instance (:40-54) ->
 * the variable assignment bellman's ProvingAssignment would hold, [105 inputs | 50429 aux].  Scalars are
 * FsRepr::write_le bytes, points edwards::Point::write bytes; member i of the anonymity set owns
 * enc_keys[i], left_ciphertexts[i] and its encrypted balance (left, right).  Prove with
 * zk_prove_batch_witness over the circuit's matrices (zk_anonymous_r1cs_load). */
#define ZK_ANONYMOUS_SIZE 12
#define ZK_ANONYMOUS_N_INPUTS 105
#define ZK_ANONYMOUS_N_AUX 50429
typedef struct {
    uint32_t amount, remaining_balance, s_index, t_index;
    uint8_t randomness[32], alpha[32], dec_key[32];
    uint8_t proof_generation_key[32], g_epoch[32];
    uint8_t enc_keys[ZK_ANONYMOUS_SIZE][32], left_ciphertexts[ZK_ANONYMOUS_SIZE][32];
    uint8_t enc_balances_left[ZK_ANONYMOUS_SIZE][32], enc_balances_right[ZK_ANONYMOUS_SIZE][32];
} zk_anonymous_statement;
/* witness_out: n x (105 + 50429) x 32 bytes; plain little-endian, or Montgomery limbs with ZK_FR_MONTGOMERY */
zk_status zk_anonymous_witness(const zk_anonymous_statement* st, size_t n, uint32_t flags, uint8_t* witness_out);
/* The same vectors from the GPU witness generator (csrc/witness_anon_gpu.h; tests compare the two element by element) */
zk_status zk_anonymous_witness_gpu(zk_r1cs* circuit, const zk_anonymous_statement* st, size_t n, uint32_t flags, uint8_t* witness_out);
/* statement -> proof for that circuit (create_random_proof of AnonymousTransfer, core/proofs/src/anonymous.rs:165):
 * witness generation on the GPU (round 4; a handful of statements, or ZKAMD_WITNESS=host: the host calculator), the kernels of
 * chunk k + 1 beside row evaluations + create_proof of chunk k, over the matrices loaded with zk_anonymous_r1cs_load. */
zk_status zk_anonymous_prove_batch(zk_params* p, zk_r1cs* circuit, size_t n, const zk_anonymous_statement* st,
                                   const uint8_t* rs, uint8_t* proofs_out);
/* The constraint system of that circuit, emitted natively (the structure half of AnonymousTransfer::synthesize):
 * zk_anonymous_r1cs_load = zk_r1cs_load of its matrices; the fingerprint is defined as for the transfer circuit
 * (core/proofs/src/circuit/test.rs:228-251).  The reference holds NO pin for it: the figures next to its test
 * (anonymous_transfer.rs:446-451: 50 634 constraints, 625c4b5d...ea37) are commented out and stale against the
 * source beside them, which has 50 514 constraints. */
zk_status zk_anonymous_r1cs_load(int device, zk_r1cs** out);
zk_status zk_anonymous_r1cs_fingerprint(uint8_t hash_out[32], uint32_t* n_inputs, uint32_t* n_aux, uint32_t* n_constraints);
/* gen_proof of the anonymous transfer (ProofBuilder::gen_proof, core/proofs/src/anonymous.rs:97-183): the key
 * derivations of the confidential entry, the anonymity set assembled with the sender at s_index, the recipient at
 * t_index and the ten decoys in their order elsewhere (:117-126), MultiCiphertexts::<Anonymous>::encrypt
 * (crypto_components.rs:168-220: -amount under the sender's key, +amount under the recipient's, zero under every
 * decoy's, one randomness), the proof, check_proof over the 104 public coordinates (:213-264; ZK_ERR_UNSATISFIABLE if
 * it fails) and the packing of AnonymousXt (gen_xt :278-348, the struct :351-359).  enc_balances_* are indexed by set member.
 * zk_anonymous_derive is the host half alone (request -> statement and rsk). */
typedef struct {
    uint32_t amount, remaining_balance, s_index, t_index;
    uint8_t spending_key[32];
    uint8_t enc_key_recipient[32], enc_keys_decoy[ZK_ANONYMOUS_SIZE - 2][32];
    uint8_t enc_balances_left[ZK_ANONYMOUS_SIZE][32], enc_balances_right[ZK_ANONYMOUS_SIZE][32];
    uint8_t g_epoch[32];
    uint8_t randomness[32], alpha[32];
} zk_anonymous_request;
typedef struct {   /* AnonymousXt, anonymous.rs:351-359 */
    uint8_t proof[192];
    uint8_t enc_keys[ZK_ANONYMOUS_SIZE][32], left_ciphertexts[ZK_ANONYMOUS_SIZE][32];
    uint8_t right_ciphertext[32], nonce[32], rsk[32], rvk[32];
} zk_anonymous_xt;
zk_status zk_anonymous_derive(const zk_anonymous_request* req, size_t n, zk_anonymous_statement* statements_out, uint8_t* rsk_out);
zk_status zk_anonymous_gen_proof_batch(zk_params* p, zk_r1cs* circuit, struct zk_vk* vk, size_t n, const zk_anonymous_request* req,
                                       const uint8_t* rs, zk_anonymous_xt* out);

/* ------------------------------------------------------------------------------------------
 * Parameter generation  (bellman groth16::generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau))
 * replaces: generate_random_parameters   reference calls: core/proofs/src/setup.rs:28-31, 59-62
 * The circuit is the R1CS held by `circuit` (zk_r1cs_load / zk_transfer_r1cs_load); the per-input rows
 * Input(i) * 0 = 0 are appended as bellman's generator appends them.  g1 / g2: the generators (uncompressed; bellman
 * draws them at random, any generators do), the five trapdoor scalars: 32 bytes little-endian plain < r.
 * out receives Parameters::write bytes (zk_params_load reads them back); out may be NULL to query the length.
 * Lagrange coefficients by the prover's NTT, QAP evaluation and all fixed-base multiplications on the GPU.
 * ZK_ERR_UNCONSTRAINED_VARIABLE as bellman (an aux variable whose L query is the identity),
 * ZK_ERR_UNEXPECTED_IDENTITY for gamma = 0 or delta = 0.
 * ------------------------------------------------------------------------------------------ */
zk_status zk_generate_parameters(zk_r1cs* circuit, const uint8_t g1[96], const uint8_t g2[192], const uint8_t alpha[32],
                                 const uint8_t beta[32], const uint8_t gamma[32], const uint8_t delta[32],
                                 const uint8_t tau[32], uint8_t* out, size_t cap, size_t* len);

/* ------------------------------------------------------------------------------------------
 * Verification  (bellman-verifier: prepare_verifying_key / verify_proof, PreparedVerifyingKey IO)
 * replaces: prepare_verifying_key + verify_proof   core/bellman-verifier/src/verifier.rs:15-63
 *           (called by the wallet's self-check core/proofs/src/confidential.rs:208-278 and by the
 *           runtime, modules/zk-system/src/lib.rs:57-108)
 * A zk_vk is a PreparedVerifyingKey resident on the GPU: e(alpha, beta), the line coefficients of
 * -gamma and -delta, the doubling tables of ic.
 *   zk_params_write_vk  VerifyingKey::write of the key inside a loaded Parameters (the first bytes of the
 *                       parameter file: alpha_g1 | beta_g1 | beta_g2 | gamma_g2 | delta_g1 | delta_g2 | n_ic | ic)
 *   zk_vk_prepare       VerifyingKey bytes -> prepare_verifying_key (pairing and G2 preparation on the GPU)
 *   zk_vk_read / write  PreparedVerifyingKey::read / write (core/bellman-verifier/src/lib.rs:175-244; the
 *                       reference's zface/params/conf_vk.dat is such a file)
 *   zk_verify_batch     n independent verify_proof calls (eighteen lanes per Fq12 element, csrc/pairing.h): proofs n x 192 bytes
 *                       (Proof::read: compressed points, curve and subgroup checks), public inputs
 *                       n x n_inputs x 32 bytes (plain little-endian, WITHOUT the leading ONE).
 *                       ok_out[i] = 1 iff proof i verifies; a malformed proof or input is 0, not an error.
 *                       ZK_ERR_MALFORMED_VERIFYING_KEY if n_inputs + 1 != ic length (verifier.rs:38-40).
 *                       The entry picks the faster of the two forms per chunk itself (round 6): the per-proof check below 4096
 *                       proofs, the combined check of zk_verify_batch_rlc from there on (measured crossover, verify.cpp
 *                       VERIFY_RLC_AUTO_MIN) - the verdicts are the same either way.
 * ------------------------------------------------------------------------------------------ */
typedef struct zk_vk zk_vk;
zk_status zk_params_write_vk(const zk_params* p, uint8_t* out, size_t cap, size_t* len);
zk_status zk_vk_prepare(const uint8_t* vk_bytes, size_t len, int device, zk_vk** out);
zk_status zk_vk_read(const uint8_t* pvk_bytes, size_t len, int device, zk_vk** out);
/* out may be NULL to query the length */
zk_status zk_vk_write(const zk_vk* vk, uint8_t* out, size_t cap, size_t* len);
zk_status zk_vk_num_inputs(const zk_vk* vk, uint32_t* n_inputs);
void zk_vk_free(zk_vk* vk);
zk_status zk_verify_batch(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs,
                          uint8_t* ok_out);
/* The same verdicts through a random linear combination of the batch (bellman's batch verifier):
 *     prod_i e(rho_i A_i, B_i) * e(sum_i rho_i acc_i, -gamma) * e(sum_i rho_i C_i, -delta) == e(alpha, beta)^(sum_i rho_i)
 * - n + 2 Miller loops and ONE final exponentiation per chunk of up to 8192 proofs instead of 3 n and n.  rho_i = 128 bits of
 * Blake2s over the batch itself (proofs, inputs, index): no randomness source, a wrong accept has probability 2^-128 per
 * attempt.  A chunk that holds a malformed or invalid proof - or whose combined check fails - is handed to the per-proof
 * verifier, which names the culprits: ok_out is ALWAYS what zk_verify_batch would have written.  Pays on large batches
 * (throughput); for a few hundred proofs the per-proof entry is as fast (both are one chain of serial stages). */
zk_status zk_verify_batch_rlc(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs,
                              uint8_t* ok_out);
zk_status zk_verify_proof(zk_vk* vk, const uint8_t proof[192], const uint8_t* public_inputs, size_t n_inputs, int* ok);
/* Proof::read (core/bellman-verifier/src/lib.rs:67-110) for n proofs of 192 bytes, without the pairing: is every
 * point a well-formed compressed encoding (ec.rs:785-837, :1438-1518) of a curve point in the r-torsion subgroup that
 * is not the point at infinity?  status_out[i] = 0 when Proof::read would succeed, else
 * (which point: 1 = A, 2 = B, 3 = C) | (reason << 2) for the first point that fails.  vk supplies the device and
 * the workspaces; its key material is not used. */
enum { ZK_PROOF_BAD_ENCODING = 1, ZK_PROOF_NOT_ON_CURVE = 2, ZK_PROOF_NOT_IN_SUBGROUP = 3, ZK_PROOF_INFINITY = 4 };
zk_status zk_proof_read_batch(zk_vk* vk, size_t n, const uint8_t* proofs, uint8_t* status_out);

/* ------------------------------------------------------------------------------------------
 * Stand-alone kernels (micro-benchmark / test entries)
 * ------------------------------------------------------------------------------------------ */
/* multiexp over G1 / G2: sum_i scalars[i] * bases[i].   replaces bellman multiexp (FullDensity).
 * bases: n x 96 (G1) / n x 192 (G2) uncompressed; scalars: n x 32 plain LE; out: uncompressed. */
typedef struct zk_msm zk_msm;
zk_status zk_msm_create(int group /*1 = G1, 2 = G2*/, const uint8_t* bases, size_t n, int window_bits /*0 = auto*/,
                        int checked, int device, zk_msm** out);
/* The same handle WITHOUT the table of doublings (bases that are used once or a few times: only the n bases stay
 * resident): a scalar is recoded into odd signed digits of `window_bits` bits at fixed positions (0 = auto), one
 * bucket set per digit position, the position results folded with doublings - classic variable-base Pippenger. */
zk_status zk_msm_create_variable(int group, const uint8_t* bases, size_t n, int window_bits, int checked, int device,
                                 zk_msm** out);
zk_status zk_msm_run(zk_msm* m, const uint8_t* scalars, uint32_t flags, uint8_t* out);
/* scalars already in HBM (device pointer, n x 32 bytes); out is a host buffer */
zk_status zk_msm_run_dev(zk_msm* m, const void* d_scalars, uint32_t flags, uint8_t* out);
void zk_msm_free(zk_msm* m);
zk_status zk_msm_g1(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[96]);
zk_status zk_msm_g2(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[192]);
/* The one-shot entries keep a handle per (device, group) - decoded bases, sort and reduction workspaces, ~1.1 KB per base -
 * for the next call; a call over more than 2^21 bases releases it again by itself.  This releases all of them now. */
void zk_msm_cache_release(void);

/* EvaluationDomain transforms over Fr, n = 2^log_n.
 * zk_ntt_fr: in-place on a HOST buffer of plain LE scalars, natural order in and out, the four
 * bellman operations: inverse = 0, coset = 0 -> fft ; 1, 0 -> ifft ; 0, 1 -> coset_fft ;
 * 1, 1 -> icoset_fft. */
zk_status zk_ntt_fr(uint8_t* data, uint32_t log_n, int inverse, int coset);

typedef struct zk_ntt zk_ntt;
#define ZK_NTT_INVERSE 1u
#define ZK_NTT_COSET 2u
#define ZK_NTT_IN_BITREV 4u   /* input is in bit-reversed order (skips the permutation) */
#define ZK_NTT_OUT_BITREV 8u  /* leave the output in bit-reversed order */
zk_status zk_ntt_create(uint32_t log_n, int device, zk_ntt** out);
/* d_data: device pointer, batch x 2^log_n x 32 bytes, Montgomery-form Fr, transformed in place */
zk_status zk_ntt_run_dev(zk_ntt* t, void* d_data, uint32_t batch, uint32_t flags);
void zk_ntt_free(zk_ntt* t);

/* Raw Montgomery products on the device multiplier (field 0 = Fr: 32-byte limbs, 1 = Fq: 48-byte
 * limbs; little-endian limb arrays exactly as the reference stores Fr / Fq): out = a*b*R^-1.
 * Exists so the reference's literal field KATs can be run through the kernels' arithmetic. */
zk_status zk_debug_field_mul(int field, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (used by bench.py): HIP-event timing of named kernels on the library's
 * stream.  zk_profile_begin() arms it, zk_profile_get() synchronises and reports.
 * ------------------------------------------------------------------------------------------ */
void zk_profile_begin(void);
/* returns the number of launches of `kernel` recorded since begin; *total_ms their summed time */
int zk_profile_get(const char* kernel, double* total_ms);
void zk_profile_end(void);
/* Two of the generated assembly kernels (the G2 bucket accumulation, level 1 of the G1 bucket reduction) exist in two
 * forms: the first keeps a few values in scratch memory across the loop, the second is scratch-free (1-2 % slower on a
 * healthy device, 3x faster on a device whose runtime caps the waves of scratch-using dispatches).  zk_params_load times
 * both on the device when the first key is loaded there and keeps the choice for the process (ZKAMD_KERNEL_FORM = scratch |
 * free overrides).  forms_out[0 / 1] = the form in use for the G2 accumulation / the reduction (0 = first, 1 = scratch-
 * free); ms_out = the comparison [G2 first, G2 scratch-free, reduction first, reduction scratch-free], zeros before any key
 * was loaded on the device.  No reference counterpart. */
zk_status zk_kernel_forms(int device, uint32_t forms_out[2], float ms_out[4]);
/* What the library holds on the device and in page-locked host memory, and what it wiped: every buffer that can hold
 * key-derived data (assignments, scalar vectors, the witness kernels' scratch, the digits and bucket sums of a multiexp,
 * staging areas) is zeroed before it goes back to the runtime; only the tables of a key (CRS points, twiddles) are not.
 * out = [device bytes held, device bytes of such buffers released so far, device bytes zeroed before release,
 *        page-locked bytes held, page-locked bytes released, page-locked bytes zeroed before release].
 * out[1] == out[2] and out[4] == out[5] always.  No reference counterpart (the reference leaves its heap to the OS). */
void zk_memory_stats(uint64_t out[6]);
/* the HIP stream (hipStream_t) all work of this library is enqueued on */
void* zk_stream(void);
zk_status zk_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKAMD_H */
