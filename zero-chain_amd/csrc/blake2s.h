// BLAKE2s (RFC 7693), host side: unkeyed, 32-byte digests, optional 8-byte personalisation.
//
// Used where the reference uses the blake2_rfc / blake2s crates on the proving path's host side:
//   * the constraint-system fingerprint of the circuit tests (core/proofs/src/circuit/test.rs:228-251), by which
//     the natively emitted R1CS is pinned (transfer_r1cs.h);
//   * the key derivations of the wallet-level gen_proof glue (core/proofs/src/no_std_aliases/keys.rs:25-27,
//     166-185: personalised Blake2s -> Fs) and the Jubjub group hash (core/jubjub/src/group_hash.rs:17-46).
#pragma once
#include <stdint.h>
#include <string.h>

namespace zkhash {

struct Blake2s {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buflen = 0;
    uint64_t total = 0;

    static constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                       0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

    explicit Blake2s(const uint8_t* personal8 = nullptr, uint8_t outlen = 32) {
        // parameter block: digest length, key length 0, fanout 1, depth 1, ..., personalisation in bytes 24 .. 31
        uint32_t p[8] = {0x01010000u | outlen, 0, 0, 0, 0, 0, 0, 0};
        if (personal8) {
            p[6] = (uint32_t)personal8[0] | ((uint32_t)personal8[1] << 8) | ((uint32_t)personal8[2] << 16) | ((uint32_t)personal8[3] << 24);
            p[7] = (uint32_t)personal8[4] | ((uint32_t)personal8[5] << 8) | ((uint32_t)personal8[6] << 16) | ((uint32_t)personal8[7] << 24);
        }
        for (int i = 0; i < 8; i++) h[i] = IV[i] ^ p[i];
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t SIGMA[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) |
                   ((uint32_t)block[4 * i + 3] << 24);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[8 + i] = IV[i];
        }
        v[12] ^= (uint32_t)total;
        v[13] ^= (uint32_t)(total >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x;
            v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y;
            v[d] = rotr(v[d] ^ v[a], 8);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t* s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        while (n) {
            if (buflen == 64) {   // a full buffer is only compressed once more input is known to follow
                total += 64;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 64 - buflen < n ? 64 - buflen : n;
            memcpy(buf + buflen, p, take);
            buflen += take;
            p += take;
            n -= take;
        }
    }
    void update_u64be(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (56 - 8 * i));
        update(b, 8);
    }
    void finish(uint8_t out[32]) {
        total += buflen;
        memset(buf + buflen, 0, 64 - buflen);
        compress(buf, true);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
    }
};

// BLAKE2b (RFC 7693), 64-byte digests with a 16-byte personalisation: SpendingKey::from_seed
// (core/proofs/src/no_std_aliases/keys.rs:29-58, "zech_ExpandSeed_").
struct Blake2b {
    uint64_t h[8];
    uint8_t buf[128];
    size_t buflen = 0;
    uint64_t total = 0;   // inputs here are far below 2^64 bytes
    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    explicit Blake2b(const uint8_t* personal16 = nullptr, uint8_t outlen = 64) {
        uint64_t p[8] = {0x01010000ull | outlen, 0, 0, 0, 0, 0, 0, 0};
        if (personal16)
            for (int k = 0; k < 2; k++)
                for (int j = 0; j < 8; j++) p[6 + k] |= (uint64_t)personal16[8 * k + j] << (8 * j);
        for (int i = 0; i < 8; i++) h[i] = IV[i] ^ p[i];
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t SIGMA[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) {
            m[i] = 0;
            for (int j = 0; j < 8; j++) m[i] |= (uint64_t)block[8 * i + j] << (8 * j);
        }
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[8 + i] = IV[i];
        }
        v[12] ^= total;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x;
            v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y;
            v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        while (n) {
            if (buflen == 128) {
                total += 128;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen < n ? 128 - buflen : n;
            memcpy(buf + buflen, p, take);
            buflen += take;
            p += take;
            n -= take;
        }
    }
    void finish(uint8_t out[64]) {
        total += buflen;
        memset(buf + buflen, 0, 128 - buflen);
        compress(buf, true);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (8 * j));
    }
};

}  // namespace zkhash
