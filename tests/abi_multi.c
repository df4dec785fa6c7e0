/* Several GPUs behind ONE host process, from C: the pattern a Rust `core/proofs` (single process, zface's call site:
 * zface/src/transaction/commands.rs:311-324 -> core/proofs/src/confidential.rs:99,149) uses to prove a batch on all the
 * devices of a node without a second process or a collective.  N host threads; thread t binds itself to the NUMA node of
 * its device (zk_bind_host_to_device), loads the key there (zk_params_load), and proves its CONTIGUOUS block of the batch
 * straight into its range of the caller's ONE output buffer - in-process there is nothing to gather.  Handles of
 * different threads are independent (include/zkamd.h: one handle, one thread at a time).
 *
 *   mode "transfer": statement -> proof through zk_pipeline_* (the bench's path), statements built with the library's
 *                    own Jubjub entries, key from zk_generate_parameters over the natively emitted circuit
 *   mode "small":    a three-constraint circuit through zk_r1cs_load + zk_prove_batch_witness (seconds under the x86
 *                    emulation build: the CPU suite runs this mode, the GPU suite both)
 * Afterwards the main thread verifies EVERY proof (zk_vk_prepare + zk_verify_batch) and compares the buffer with the
 * same batch proved by one thread in one call: byte-identical (a proof is a function of key, statement and (r, s)).
 *
 *   gcc -std=c99 -pedantic -Wall -Werror -Iinclude tests/abi_multi.c -ldl -lpthread -o abi_multi
 *   ./abi_multi <libzkamd.so> <transfer|small> <threads> <statements> [device,device,...]   (default: every thread on 0)
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkamd.h"

#define FN(ret, name, args) typedef ret(*name##_t) args; static name##_t p_##name
FN(const char*, zk_last_error, (void));
FN(const char*, zk_strerror, (zk_status));
FN(zk_status, zk_device_count, (int*));
FN(zk_status, zk_bind_host_to_device, (int, int*, int*));
FN(zk_status, zk_params_load, (const uint8_t*, size_t, int, int, zk_params**));
FN(void, zk_params_free, (zk_params*));
FN(zk_status, zk_params_write_vk, (const zk_params*, uint8_t*, size_t, size_t*));
FN(zk_status, zk_r1cs_load, (uint32_t, uint32_t, uint32_t, const zk_csr*, const zk_csr*, const zk_csr*, int, zk_r1cs**));
FN(zk_status, zk_transfer_r1cs_load, (int, zk_r1cs**));
FN(void, zk_r1cs_free, (zk_r1cs*));
FN(zk_status, zk_generate_parameters, (zk_r1cs*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*,
                                       const uint8_t*, const uint8_t*, uint8_t*, size_t, size_t*));
FN(zk_status, zk_prove_batch_witness, (zk_params*, zk_r1cs*, size_t, const uint8_t*, uint32_t, const uint8_t*, uint8_t*));
FN(zk_status, zk_transfer_prove_batch, (zk_params*, zk_r1cs*, size_t, const zk_transfer_statement*, const uint8_t*, uint8_t*));
FN(zk_status, zk_transfer_witness, (const zk_transfer_statement*, size_t, uint32_t, uint8_t*));
FN(zk_status, zk_pipeline_create, (zk_params*, zk_r1cs*, zk_pipeline**));
FN(zk_status, zk_pipeline_submit, (zk_pipeline*, size_t, const zk_transfer_statement*, const uint8_t*, uint8_t*));
FN(zk_status, zk_pipeline_wait, (zk_pipeline*));
FN(void, zk_pipeline_free, (zk_pipeline*));
FN(zk_status, zk_jubjub_base_mul, (const uint8_t*, size_t, uint8_t*));
FN(zk_status, zk_elgamal_encrypt, (const uint32_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint8_t*));
FN(zk_status, zk_vk_prepare, (const uint8_t*, size_t, int, zk_vk**));
FN(void, zk_vk_free, (zk_vk*));
FN(zk_status, zk_verify_batch, (zk_vk*, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*));

static void* lib;
static int resolve(void) {
    int missing = 0;
#define GET(name) do { *(void**)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); missing++; } } while (0)
    GET(zk_last_error); GET(zk_strerror); GET(zk_device_count); GET(zk_bind_host_to_device); GET(zk_params_load); GET(zk_params_free);
    GET(zk_params_write_vk); GET(zk_r1cs_load); GET(zk_transfer_r1cs_load); GET(zk_r1cs_free); GET(zk_generate_parameters);
    GET(zk_prove_batch_witness); GET(zk_transfer_prove_batch); GET(zk_transfer_witness); GET(zk_pipeline_create); GET(zk_pipeline_submit);
    GET(zk_pipeline_wait); GET(zk_pipeline_free); GET(zk_jubjub_base_mul); GET(zk_elgamal_encrypt); GET(zk_vk_prepare); GET(zk_vk_free);
    GET(zk_verify_batch);
    return missing;
}
#define CHECK(expr) do { zk_status s_ = (expr); if (s_ != ZK_OK) { fprintf(stderr, "%s -> %d (%s): %s\n", #expr, (int)s_, p_zk_strerror(s_), p_zk_last_error()); return 1; } } while (0)

/* BLS12-381 generators, uncompressed (core/pairing/src/bls12_381/README.md:45-57) */
static const char* G1_HEX =
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
static const char* G2_HEX =
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
static void unhex(const char* h, uint8_t* out, size_t n) {
    size_t i;
    for (i = 0; i < n; i++) {
        unsigned v;
        sscanf(h + 2 * i, "%2x", &v);
        out[i] = (uint8_t)v;
    }
}
static uint64_t sm_state = 0x243f6a8885a308d3ull;
static uint64_t splitmix(void) {
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
/* a scalar below 2^248: canonical in Fr and in Fs, 32 bytes little-endian */
static void small_scalar(uint8_t out[32]) {
    int i;
    for (i = 0; i < 4; i++) {
        uint64_t v = splitmix();
        memcpy(out + 8 * i, &v, 8);   /* (little-endian hosts only, as the library) */
    }
    out[31] = 0;
}

enum { MODE_SMALL = 0, MODE_TRANSFER = 1 };
static int mode, n_total;
static uint8_t *pk, *rs, *proofs, *small_z;
static size_t pk_len;
static zk_transfer_statement* sts;
/* the three-constraint circuit: inputs (ONE, out), aux (x, y, z):  x * x = y,  y * x = z,  (z + x + 5 ONE) * ONE = out */
enum { S_IN = 2, S_AUX = 3, S_CON = 3 };
static uint8_t coef_one[32], coef_five[32];
static zk_status small_circuit(int device, zk_r1cs** out) {
    static const uint32_t a_ptr[4] = {0, 1, 2, 5}, a_col[5] = {2, 3, 4, 2, 0};   /* x | y | z + x + 5 */
    static const uint32_t b_ptr[4] = {0, 1, 2, 3}, b_col[3] = {2, 2, 0};         /* x | x | ONE */
    static const uint32_t c_ptr[4] = {0, 1, 2, 3}, c_col[3] = {3, 4, 1};         /* y | z | out */
    static uint8_t a_co[5 * 32], b_co[3 * 32], c_co[3 * 32];
    zk_csr A, B, C;
    int i;
    for (i = 0; i < 5; i++) memcpy(a_co + 32 * i, i == 4 ? coef_five : coef_one, 32);
    for (i = 0; i < 3; i++) {
        memcpy(b_co + 32 * i, coef_one, 32);
        memcpy(c_co + 32 * i, coef_one, 32);
    }
    A.row_ptr = a_ptr; A.col = a_col; A.coeff = a_co;
    B.row_ptr = b_ptr; B.col = b_col; B.coeff = b_co;
    C.row_ptr = c_ptr; C.col = c_col; C.coeff = c_co;
    return p_zk_r1cs_load(S_IN, S_AUX, S_CON, &A, &B, &C, device, out);
}
static void put_u64(uint8_t* out, uint64_t v) {
    memset(out, 0, 32);
    memcpy(out, &v, 8);
}

typedef struct {
    int t, n_threads, device, rc, numa, cpus;
} job_t;

static int run_block(job_t* j) {
    zk_params* p = NULL;
    zk_r1cs* c = NULL;
    const int lo = (int)(((long)n_total * j->t) / j->n_threads), hi = (int)(((long)n_total * (j->t + 1)) / j->n_threads);
    CHECK(p_zk_bind_host_to_device(j->device, &j->numa, &j->cpus));
    if (mode == MODE_TRANSFER) CHECK(p_zk_transfer_r1cs_load(j->device, &c));
    else CHECK(small_circuit(j->device, &c));
    CHECK(p_zk_params_load(pk, pk_len, 0, j->device, &p));
    if (hi > lo) {
        if (mode == MODE_TRANSFER) {
            zk_pipeline* pl = NULL;
            CHECK(p_zk_pipeline_create(p, c, &pl));
            CHECK(p_zk_pipeline_submit(pl, (size_t)(hi - lo), sts + lo, rs + 64 * (size_t)lo, proofs + 192 * (size_t)lo));
            CHECK(p_zk_pipeline_wait(pl));
            p_zk_pipeline_free(pl);
        } else {
            CHECK(p_zk_prove_batch_witness(p, c, (size_t)(hi - lo), small_z + (size_t)lo * (S_IN + S_AUX) * 32, 0, rs + 64 * (size_t)lo,
                                           proofs + 192 * (size_t)lo));
        }
    }
    p_zk_params_free(p);
    p_zk_r1cs_free(c);
    return 0;
}
static void* thread_main(void* arg) {
    job_t* j = (job_t*)arg;
    j->rc = run_block(j);
    return NULL;
}

static int run(int argc, char** argv) {
    int n_threads, n_dev = 0, devs[64], i, t;
    uint8_t g1[96], g2[192], toxic[5][32], *inputs, *ok, *ref, *vk_bytes;
    size_t vk_len = 0, n_pub;
    zk_r1cs* c0 = NULL;
    zk_params* p0 = NULL;
    zk_vk* vk = NULL;
    pthread_t th[64];
    job_t jobs[64];
    mode = strcmp(argv[2], "transfer") == 0 ? MODE_TRANSFER : MODE_SMALL;
    n_threads = atoi(argv[3]);
    n_total = atoi(argv[4]);
    if (n_threads < 1 || n_threads > 64 || n_total < 0) return 2;
    CHECK(p_zk_device_count(&n_dev));
    for (t = 0; t < n_threads; t++) devs[t] = 0;
    if (argc > 5) {
        char* s = argv[5];
        for (t = 0; t < n_threads && *s; t++) {
            devs[t] = (int)strtol(s, &s, 10);
            if (*s == ',') s++;
        }
    }
    unhex(G1_HEX, g1, 96);
    unhex(G2_HEX, g2, 192);
    for (i = 0; i < 5; i++) small_scalar(toxic[i]);
    put_u64(coef_one, 1);
    put_u64(coef_five, 5);
    /* the key, once, as bytes every thread loads on its own device */
    if (mode == MODE_TRANSFER) CHECK(p_zk_transfer_r1cs_load(devs[0], &c0));
    else CHECK(small_circuit(devs[0], &c0));
    CHECK(p_zk_generate_parameters(c0, g1, g2, toxic[0], toxic[1], toxic[2], toxic[3], toxic[4], NULL, 0, &pk_len));
    pk = (uint8_t*)malloc(pk_len);
    CHECK(p_zk_generate_parameters(c0, g1, g2, toxic[0], toxic[1], toxic[2], toxic[3], toxic[4], pk, pk_len, &pk_len));
    rs = (uint8_t*)malloc(64 * (size_t)(n_total + 1));
    for (i = 0; i < 2 * n_total; i++) small_scalar(rs + 32 * (size_t)i);
    proofs = (uint8_t*)calloc((size_t)n_total + 1, 192);
    ref = (uint8_t*)calloc((size_t)n_total + 1, 192);
    ok = (uint8_t*)calloc((size_t)n_total + 1, 1);
    if (mode == MODE_TRANSFER) {
        /* statement i: keys and ciphertexts from the library's own Jubjub entries (as bench.py make_statements_native) */
        uint8_t* sc = (uint8_t*)malloc(4 * 32 * (size_t)(n_total + 1));
        uint8_t* pts = (uint8_t*)malloc(4 * 32 * (size_t)(n_total + 1));
        uint8_t* rb = (uint8_t*)malloc(32 * (size_t)(n_total + 1));
        uint8_t* keys = (uint8_t*)malloc(32 * (size_t)(n_total + 1));
        uint8_t* left = (uint8_t*)malloc(32 * (size_t)(n_total + 1));
        uint8_t* right = (uint8_t*)malloc(32 * (size_t)(n_total + 1));
        uint32_t* bal = (uint32_t*)malloc(4 * (size_t)(n_total + 1));
        sts = (zk_transfer_statement*)calloc((size_t)n_total + 1, sizeof(zk_transfer_statement));
        for (i = 0; i < n_total; i++) {
            int k;
            for (k = 0; k < 4; k++) small_scalar(sc + 32 * (size_t)(4 * i + k));   /* dec_key, recipient, pgk, epoch */
            sc[32 * (size_t)(4 * i) + 30] = 0;                                        /* dec_key: the top bits of a real one are dropped */
            small_scalar(rb + 32 * (size_t)i);
            bal[i] = 1000u + 17u * (uint32_t)i;
        }
        CHECK(p_zk_jubjub_base_mul(sc, 4 * (size_t)n_total, pts));
        for (i = 0; i < n_total; i++) memcpy(keys + 32 * (size_t)i, pts + 32 * (size_t)(4 * i), 32);
        CHECK(p_zk_elgamal_encrypt(bal, rb, keys, (size_t)n_total, left, right));
        for (i = 0; i < n_total; i++) {
            zk_transfer_statement* s = &sts[i];
            s->amount = 10u + (uint32_t)i;
            s->fee = 1u + (uint32_t)(i % 3);
            s->remaining_balance = bal[i] - s->amount - s->fee;
            small_scalar(s->randomness);
            small_scalar(s->alpha);
            memcpy(s->dec_key_sender, sc + 32 * (size_t)(4 * i), 32);
            memcpy(s->enc_key_recipient, pts + 32 * (size_t)(4 * i + 1), 32);
            memcpy(s->proof_generation_key, pts + 32 * (size_t)(4 * i + 2), 32);
            memcpy(s->g_epoch, pts + 32 * (size_t)(4 * i + 3), 32);
            memcpy(s->enc_balance_left, left + 32 * (size_t)i, 32);
            memcpy(s->enc_balance_right, right + 32 * (size_t)i, 32);
        }
        free(sc); free(pts); free(rb); free(keys); free(left); free(right); free(bal);
        n_pub = ZK_TRANSFER_N_INPUTS - 1;
    } else {
        small_z = (uint8_t*)calloc((size_t)n_total + 1, (S_IN + S_AUX) * 32);
        for (i = 0; i < n_total; i++) {
            const uint64_t x = 2 + (uint64_t)i, y = x * x, z = y * x;
            uint8_t* w = small_z + (size_t)i * (S_IN + S_AUX) * 32;
            put_u64(w, 1);
            put_u64(w + 32, z + x + 5);
            put_u64(w + 64, x);
            put_u64(w + 96, y);
            put_u64(w + 128, z);
        }
        n_pub = S_IN - 1;
    }
    /* the batch, cut into contiguous blocks: one thread, one device, one range of `proofs` each */
    for (t = 0; t < n_threads; t++) {
        jobs[t].t = t; jobs[t].n_threads = n_threads; jobs[t].device = devs[t]; jobs[t].rc = 0; jobs[t].numa = -1; jobs[t].cpus = 0;
        if (devs[t] < 0 || devs[t] >= n_dev) {
            fprintf(stderr, "device %d out of range (%d visible)\n", devs[t], n_dev);
            return 2;
        }
        if (pthread_create(&th[t], NULL, thread_main, &jobs[t]) != 0) return 1;
    }
    for (t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    for (t = 0; t < n_threads; t++)
        if (jobs[t].rc) return 1;
    /* every proof verifies, and the buffer is what ONE thread makes of the same batch in one call */
    CHECK(p_zk_params_load(pk, pk_len, 0, devs[0], &p0));
    CHECK(p_zk_params_write_vk(p0, NULL, 0, &vk_len));
    vk_bytes = (uint8_t*)malloc(vk_len);
    CHECK(p_zk_params_write_vk(p0, vk_bytes, vk_len, &vk_len));
    CHECK(p_zk_vk_prepare(vk_bytes, vk_len, devs[0], &vk));
    inputs = (uint8_t*)calloc((size_t)n_total + 1, n_pub * 32);
    if (mode == MODE_TRANSFER) {
        const size_t nv = ZK_TRANSFER_N_INPUTS + ZK_TRANSFER_N_AUX;
        uint8_t* w = (uint8_t*)malloc(nv * 32);
        for (i = 0; i < n_total; i++) {   /* the public inputs of a statement are the head of its assignment (behind ONE) */
            CHECK(p_zk_transfer_witness(&sts[i], 1, 0, w));
            memcpy(inputs + (size_t)i * n_pub * 32, w + 32, n_pub * 32);
        }
        free(w);
        if (n_total) CHECK(p_zk_transfer_prove_batch(p0, c0, (size_t)n_total, sts, rs, ref));
    } else {
        for (i = 0; i < n_total; i++) memcpy(inputs + (size_t)i * 32, small_z + (size_t)i * (S_IN + S_AUX) * 32 + 32, 32);
        if (n_total) CHECK(p_zk_prove_batch_witness(p0, c0, (size_t)n_total, small_z, 0, rs, ref));
    }
    CHECK(p_zk_verify_batch(vk, (size_t)n_total, proofs, inputs, n_pub, ok));
    for (i = 0; i < n_total; i++)
        if (!ok[i]) {
            fprintf(stderr, "proof %d does not verify\n", i);
            return 1;
        }
    if (memcmp(proofs, ref, 192 * (size_t)n_total) != 0) {
        fprintf(stderr, "the threads' proofs differ from the one-call result\n");
        return 1;
    }
    if (n_total) {   /* and a proof moved to another statement's slot is refused (the check above is not vacuous) */
        memcpy(ref, proofs, 192 * (size_t)n_total);
        if (n_total > 1) {
            memcpy(ref, proofs + 192, 192);
            CHECK(p_zk_verify_batch(vk, (size_t)n_total, ref, inputs, n_pub, ok));
            if (ok[0]) {
                fprintf(stderr, "a proof of statement 1 was accepted for statement 0\n");
                return 1;
            }
        }
    }
    p_zk_vk_free(vk);
    p_zk_params_free(p0);
    p_zk_r1cs_free(c0);
    printf("abi_multi ok: %d %s proofs from %d threads on devices [", n_total, mode == MODE_TRANSFER ? "transfer" : "small-circuit", n_threads);
    for (t = 0; t < n_threads; t++) printf("%s%d", t ? "," : "", devs[t]);
    printf("] (numa nodes [");
    for (t = 0; t < n_threads; t++) printf("%s%d", t ? "," : "", jobs[t].numa);
    printf("]), all verified, equal to the one-call result\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <libzkamd.so> <transfer|small> <threads> <statements> [device,device,...]\n", argv[0]);
        return 2;
    }
    lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    if (resolve()) return 1;
    return run(argc, argv);
}
