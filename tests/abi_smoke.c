/* C (not Python) smoke test of the drop-in boundary: include/zkamd.h compiles as strict C99, the struct layouts are the
 * ones the Rust #[repr(C)] mirrors of INTEGRATION.md (and the ctypes mirrors of zero-chain_amd/_lib.py) assume, and a
 * C program can dlopen the library, resolve every entry point the header declares and call zk_strerror - what the
 * reference-side binding (a `#[link(name = "zkamd")] extern "C"` block) relies on.  No GPU needed.
 *
 *   gcc -std=c99 -pedantic -Wall -Werror -Iinclude tests/abi_smoke.c -ldl -o abi_smoke && ./abi_smoke <library> <symbol>...
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "zkamd.h"

#define SIZE_IS(type, n) typedef char size_of_##type##_is_##n[(sizeof(type) == (n)) ? 1 : -1]
#define OFFSET_IS(type, field, n) typedef char offset_of_##type##_##field##_is_##n[(offsetof(type, field) == (n)) ? 1 : -1]

/* sizes on the one ABI this library is built for (x86-64 Linux, LP64) */
SIZE_IS(zk_params_info, 48);
SIZE_IS(zk_assignment, 80);
SIZE_IS(zk_batch_dev, 72);
SIZE_IS(zk_csr, 24);
SIZE_IS(zk_transfer_statement, 272);
SIZE_IS(zk_transfer_request, 240);
SIZE_IS(zk_confidential_xt, 544);
SIZE_IS(zk_anonymous_statement, 1712);
SIZE_IS(zk_anonymous_request, 1264);
SIZE_IS(zk_anonymous_xt, 1088);
/* a few field offsets the Rust mirrors spell out */
OFFSET_IS(zk_assignment, a, 16);
OFFSET_IS(zk_assignment, b_aux_density, 72);
OFFSET_IS(zk_transfer_statement, randomness, 16);
OFFSET_IS(zk_transfer_statement, g_epoch, 240);
OFFSET_IS(zk_transfer_request, spending_key, 16);
OFFSET_IS(zk_confidential_xt, enc_key_sender, 192);
OFFSET_IS(zk_confidential_xt, nonce, 512);
OFFSET_IS(zk_anonymous_xt, enc_keys, 192);
OFFSET_IS(zk_params_info, device_bytes, 40);

/* status codes 1..8 are bellman's SynthesisError in declaration order (core/bellman-verifier/src/lib.rs:331-357) */
typedef char status_codes[(ZK_OK == 0 && ZK_ERR_ASSIGNMENT_MISSING == 1 && ZK_ERR_UNCONSTRAINED_VARIABLE == 8) ? 1 : -1];

int main(int argc, char** argv) {
    void* h;
    const char* (*strerror_fn)(zk_status);
    int i, missing = 0;
    if (argc < 2) {
        fprintf(stderr, "usage: %s <libzkamd.so> [symbol ...]\n", argv[0]);
        return 2;
    }
    h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    for (i = 2; i < argc; i++)
        if (!dlsym(h, argv[i])) {
            fprintf(stderr, "missing symbol: %s\n", argv[i]);
            missing++;
        }
    *(void**)(&strerror_fn) = dlsym(h, "zk_strerror");
    if (!strerror_fn || !strerror_fn(ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE) ||
        !strstr(strerror_fn(ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE), "polynomial degree")) {
        fprintf(stderr, "zk_strerror does not answer\n");
        return 1;
    }
    printf("abi ok: %d symbols resolved, zk_strerror(4) = \"%s\"\n", argc - 2 - missing, strerror_fn(ZK_ERR_POLYNOMIAL_DEGREE_TOO_LARGE));
    return missing ? 1 : 0;
}
