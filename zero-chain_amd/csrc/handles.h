// Handle types shared by the translation units of libzkamd.
#pragma once
#include <vector>
#include "host_common.h"
#include "host_math.h"

// ------------------------------------------------------------------------------------------
// zk_r1cs: the fixed constraint matrices of a circuit, resident on the GPU
// ------------------------------------------------------------------------------------------
struct zk_r1cs {
    int device = 0;
    uint32_t n_in = 0, n_aux = 0, n_con = 0;
    zkrt::DevBuf row_ptr[3], col[3], coeff[3];
    std::vector<uint8_t> a_aux_density, b_input_density, b_aux_density;
    // host copy of the matrices (the parameter generator transposes them): CSR, Montgomery coefficients
    std::vector<uint32_t> h_row_ptr[3], h_col[3];
    std::vector<zkhost::Fr> h_coeff[3];
    // per-chunk workspaces: Montgomery assignment (two: the witness kernels fill one while the prover reads the
    // other), row evaluations
    zkrt::DevBuf z[2], abc;
    // GPU witness generator of the transfer circuit (witness_gpu.h)
    zkrt::DevBuf wit_st[2], wit_bad[2], wit_pts, wit_table, wit_consts, wit_scratch;
    zkrt::PinBuf pin_st[2], pin_bad[2];
    hipEvent_t wit_done[2] = {nullptr, nullptr};
    size_t anon_index_bad[2] = {(size_t)-1, (size_t)-1};   // first statement of the slot whose member indices are out of range
    bool wit_ready = false;
    // pinned host buffer of zk_transfer_prove_batch in host-witness mode (witness vectors of one chunk)
    void* host_z = nullptr;
    size_t host_z_cap = 0;
    ~zk_r1cs() {
        if (host_z) {
            explicit_bzero(host_z, host_z_cap);   // witness vectors: the bits of the keys
            (void)hipHostFree(host_z);
        }
        for (int k = 0; k < 2; k++)
            if (wit_done[k]) (void)hipEventDestroy(wit_done[k]);
    }
    zk_status host_ensure(size_t bytes) {
        if (bytes <= host_z_cap) return ZK_OK;
        if (host_z) {
            explicit_bzero(host_z, host_z_cap);
            (void)hipHostFree(host_z);
        }
        host_z = nullptr;
        host_z_cap = 0;
        if (hipHostMalloc(&host_z, bytes) != hipSuccess) {
            host_z = nullptr;
            return zkrt::fail(ZK_ERR_OUT_OF_MEMORY, "hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
        }
        host_z_cap = bytes;
        return ZK_OK;
    }
};


// GPU witness generator of the transfer circuit (witness.cpp): np statements -> R->z[slot], enqueued on `stream`;
// finish() waits for it and reports a malformed statement with its absolute index
// typed_inputs: also run the reference's as_prime_order on the four points a wallet-level request brings in
zk_status witness_gpu_enqueue(zk_r1cs* R, const zk_transfer_statement* st, size_t np, int slot, hipStream_t stream,
                              bool typed_inputs = false);
zk_status witness_gpu_finish(zk_r1cs* R, size_t np, int slot, size_t index_base);
// ... and of the anonymous-transfer circuit (witness_anon_gpu.h)
zk_status witness_anon_gpu_enqueue(zk_r1cs* R, const zk_anonymous_statement* st, size_t np, int slot, hipStream_t stream,
                                   size_t index_base = 0);
zk_status witness_anon_gpu_finish(zk_r1cs* R, size_t np, int slot, size_t index_base);

// Groth16 verification of a batch (verify.cpp; zk_verify_batch is this with own_proofs = false).  own_proofs: the
// proofs are this library's own fresh results (gen_proof's self-check) - decoded without the r-torsion test.
namespace zkrt {
// what wallet.cpp (gen_proof, the Jubjub host side) needs of zkamd.cpp: one chunk of proofs from the assignment the witness
// kernels left in R->z[slot]; proofs per launch set; the key's device and evaluation domain
zk_status lib_prove_from_z(zk_params* P, zk_r1cs* R, size_t np, int slot, const uint8_t* rs, uint8_t* proofs_out);
size_t lib_batch_chunk();
bool lib_witness_on_host(size_t n);   // a handful of statements: the assignment on the host cores (zkamd.cpp witness_on_host)
int lib_params_device(const zk_params* P);
size_t lib_params_domain(const zk_params* P);
zk_status verify_batch(zk_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, size_t n_inputs, uint8_t* ok_out,
                       bool own_proofs, int form = VERIFY_AUTO, const uint8_t* own_affine = nullptr);
}
