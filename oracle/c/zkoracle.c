/* zkoracle.c - CPU restatement of the reference prover's algorithms (ORACLE: TEST
 * INFRASTRUCTURE, never shipped, never linked by the product; see oracle/__init__.py).
 *
 * Plain C, 64-bit limbs.  Used (a) by tests/ to check the HIP path at sizes Python big-ints
 * cannot reach, (b) by bench.py's `cpu_baseline` leg ("port": bellman's algorithm timed on the
 * GPU box's host cores), (c) to generate synthetic CRS / bases for those two.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   Fr / Fq Montgomery arithmetic   core/pairing/src/bls12_381/fr.rs:341-571, fq.rs:749-1127
 *                                   (mac_with_carry / adc / sbb of core/pairing/src/lib.rs:626-739)
 *   Fq2                             core/pairing/src/bls12_381/fq2.rs:90-182
 *   Jacobian G1/G2: double, add_assign, add_assign_mixed, into_affine, mul
 *                                   core/pairing/src/bls12_381/ec.rs:296-354, :356-444, :446-526,
 *                                   :586-618, :534-553
 *   encodings                       ec.rs:666-868 (G1), :1303-1548 (G2)
 *   multiexp (Pippenger)            bellman 0.1.0 multiexp.rs   [NOT IN TREE: LayerXcom/librustzcash
 *   EvaluationDomain / best_fft     bellman 0.1.0 domain.rs      rev 2c19687, Cargo.lock:210-212;
 *   create_proof                    bellman 0.1.0 groth16/prover.rs   restated per SURVEY.md A.1-A.3]
 * Parity pins: tests/test_oracle_c.py checks this file against the Python oracle, which is itself
 * pinned on the reference's golden vectors and the DummyEngine prover KAT.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic N-limb Montgomery */
typedef struct {
    int n;
    uint64_t p[6], r[6], r2[6], inv;
} field_t;

static const field_t FQ = {6,
    {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull},
    {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull},
    {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull, 0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull},
    0x89f3fffcfffcfffdull};
static const field_t FR = {4,
    {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull, 0, 0},
    {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full, 0, 0},
    {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull, 0, 0},
    0xfffffffeffffffffull};

typedef struct { uint64_t l[6]; } fq;
typedef struct { uint64_t l[4]; } fr;

static inline int geq(const uint64_t* a, const uint64_t* p, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > p[i]) return 1;
        if (a[i] < p[i]) return 0;
    }
    return 1;
}
static inline void subn(uint64_t* a, const uint64_t* p, int n) {
    u128 b = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - p[i] - b;
        a[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline void f_add(const field_t* F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
    u128 c = 0;
    for (int i = 0; i < F->n; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq(r, F->p, F->n)) subn(r, F->p, F->n);
}
static inline void f_sub(const field_t* F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
    u128 bo = 0;
    uint64_t t[6];
    for (int i = 0; i < F->n; i++) {
        u128 d = (u128)a[i] - b[i] - bo;
        t[i] = (uint64_t)d;
        bo = (d >> 64) & 1;
    }
    if (bo) {
        u128 c = 0;
        for (int i = 0; i < F->n; i++) {
            c += (u128)t[i] + F->p[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r, t, 8 * F->n);
}
/* mul_assign + mont_reduce (fq.rs:915-1127): schoolbook product then word-by-word reduction */
static inline void f_mul(const field_t* F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
    const int n = F->n;
    uint64_t t[13];
    memset(t, 0, sizeof t);
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)a[i] * b[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + n] = (uint64_t)c;
    }
    uint64_t carry2 = 0;
    for (int i = 0; i < n; i++) {
        uint64_t k = t[i] * F->inv;
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)k * F->p[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        c += (u128)t[i + n] + carry2;
        t[i + n] = (uint64_t)c;
        carry2 = (uint64_t)(c >> 64);
    }
    memcpy(r, t + n, 8 * n);
    if (carry2 || geq(r, F->p, n)) subn(r, F->p, n);
}
static inline int f_is_zero(const uint64_t* a, int n) {
    uint64_t o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
}
static void f_pow(const field_t* F, uint64_t* r, const uint64_t* a, const uint64_t* e, int en) {
    uint64_t acc[6], base[6];
    memcpy(acc, F->r, 8 * F->n);
    memcpy(base, a, 8 * F->n);
    for (int i = en - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            f_mul(F, acc, acc, acc);
            if ((e[i] >> b) & 1) f_mul(F, acc, acc, base);
        }
    memcpy(r, acc, 8 * F->n);
}
static void f_inv(const field_t* F, uint64_t* r, const uint64_t* a) {
    uint64_t e[6];
    memcpy(e, F->p, 8 * F->n);
    e[0] -= 2; /* p - 2 (p's low limb is > 2) */
    f_pow(F, r, a, e, F->n);
}
static void f_to_mont(const field_t* F, uint64_t* r, const uint64_t* a) { f_mul(F, r, a, F->r2); }
static void f_from_mont(const field_t* F, uint64_t* r, const uint64_t* a) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    f_mul(F, r, a, one);
}

/* ------------------------------------------------------------------ Fq / Fq2 wrappers */
#define Q (&FQ)
static inline fq fq_add(fq a, fq b) { fq r; f_add(Q, r.l, a.l, b.l); return r; }
static inline fq fq_sub(fq a, fq b) { fq r; f_sub(Q, r.l, a.l, b.l); return r; }
static inline fq fq_mul(fq a, fq b) { fq r; f_mul(Q, r.l, a.l, b.l); return r; }
static inline fq fq_sqr(fq a) { return fq_mul(a, a); }
static inline fq fq_dbl(fq a) { return fq_add(a, a); }
static inline fq fq_zero(void) { fq r; memset(&r, 0, sizeof r); return r; }
static inline fq fq_one(void) { fq r; memcpy(r.l, FQ.r, 48); return r; }
static inline fq fq_neg(fq a) { return fq_sub(fq_zero(), a); }
static inline int fq_is_zero(fq a) { return f_is_zero(a.l, 6); }
static inline int fq_eq(fq a, fq b) { return memcmp(a.l, b.l, 48) == 0; }
static inline fq fq_inv(fq a) { fq r; f_inv(Q, r.l, a.l); return r; }

typedef struct { fq c0, c1; } fq2;
static inline fq2 fq2_add(fq2 a, fq2 b) { fq2 r = {fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; return r; }
static inline fq2 fq2_sub(fq2 a, fq2 b) { fq2 r = {fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; return r; }
static inline fq2 fq2_dbl(fq2 a) { return fq2_add(a, a); }
static inline fq2 fq2_mul(fq2 a, fq2 b) {
    fq aa = fq_mul(a.c0, b.c0), bb = fq_mul(a.c1, b.c1);
    fq o = fq_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
    fq2 r = {fq_sub(aa, bb), fq_sub(fq_sub(o, aa), bb)};
    return r;
}
static inline fq2 fq2_sqr(fq2 a) {
    fq ab = fq_mul(a.c0, a.c1);
    fq2 r = {fq_mul(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1)), fq_dbl(ab)};
    return r;
}
static inline fq2 fq2_zero(void) { fq2 r = {fq_zero(), fq_zero()}; return r; }
static inline fq2 fq2_one(void) { fq2 r = {fq_one(), fq_zero()}; return r; }
static inline fq2 fq2_neg(fq2 a) { fq2 r = {fq_neg(a.c0), fq_neg(a.c1)}; return r; }
static inline int fq2_is_zero(fq2 a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
static inline int fq2_eq(fq2 a, fq2 b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
static inline fq2 fq2_inv(fq2 a) {
    fq t = fq_inv(fq_add(fq_sqr(a.c0), fq_sqr(a.c1)));
    fq2 r = {fq_mul(a.c0, t), fq_neg(fq_mul(a.c1, t))};
    return r;
}

/* ------------------------------------------------------------------ Jacobian groups, generated
 * twice (G1 over fq, G2 over fq2) from the same text, as ec.rs does with its curve_impl! macro. */
#define CURVE_IMPL(G, F)                                                                          \
    typedef struct { F x, y; int inf; } G##_affine;                                               \
    typedef struct { F x, y, z; } G##_t;                                                          \
    static inline G##_t G##_zero(void) { G##_t r = {F##_zero(), F##_one(), F##_zero()}; return r; } \
    static inline int G##_is_zero(const G##_t* p) { return F##_is_zero(p->z); }                   \
    /* dbl-2009-l, ec.rs:296-354 */                                                               \
    static void G##_double(G##_t* p) {                                                            \
        if (G##_is_zero(p)) return;                                                               \
        F a = F##_sqr(p->x), b = F##_sqr(p->y), c = F##_sqr(b);                                   \
        F d = F##_sub(F##_sub(F##_sqr(F##_add(p->x, b)), a), c);                                  \
        d = F##_dbl(d);                                                                           \
        F e = F##_add(F##_dbl(a), a), f = F##_sqr(e);                                             \
        F z3 = F##_dbl(F##_mul(p->z, p->y));                                                      \
        F x3 = F##_sub(F##_sub(f, d), d);                                                         \
        F c8 = F##_dbl(F##_dbl(F##_dbl(c)));                                                      \
        F y3 = F##_sub(F##_mul(F##_sub(d, x3), e), c8);                                           \
        p->x = x3; p->y = y3; p->z = z3;                                                          \
    }                                                                                             \
    /* add-2007-bl, ec.rs:356-444 */                                                              \
    static void G##_add(G##_t* p, const G##_t* o) {                                               \
        if (G##_is_zero(p)) { *p = *o; return; }                                                  \
        if (G##_is_zero(o)) return;                                                               \
        F z1z1 = F##_sqr(p->z), z2z2 = F##_sqr(o->z);                                             \
        F u1 = F##_mul(p->x, z2z2), u2 = F##_mul(o->x, z1z1);                                     \
        F s1 = F##_mul(F##_mul(p->y, o->z), z2z2), s2 = F##_mul(F##_mul(o->y, p->z), z1z1);       \
        if (F##_eq(u1, u2) && F##_eq(s1, s2)) { G##_double(p); return; }                          \
        F h = F##_sub(u2, u1), i = F##_sqr(F##_dbl(h)), j = F##_mul(h, i);                        \
        F r = F##_dbl(F##_sub(s2, s1)), v = F##_mul(u1, i);                                       \
        F x3 = F##_sub(F##_sub(F##_sub(F##_sqr(r), j), v), v);                                    \
        F y3 = F##_sub(F##_mul(F##_sub(v, x3), r), F##_dbl(F##_mul(s1, j)));                      \
        F z3 = F##_mul(F##_sub(F##_sub(F##_sqr(F##_add(p->z, o->z)), z1z1), z2z2), h);            \
        p->x = x3; p->y = y3; p->z = z3;                                                          \
    }                                                                                             \
    /* madd-2007-bl, ec.rs:446-526 */                                                             \
    static void G##_add_mixed(G##_t* p, const G##_affine* o) {                                    \
        if (o->inf) return;                                                                       \
        if (G##_is_zero(p)) { p->x = o->x; p->y = o->y; p->z = F##_one(); return; }               \
        F z1z1 = F##_sqr(p->z);                                                                   \
        F u2 = F##_mul(o->x, z1z1), s2 = F##_mul(F##_mul(o->y, p->z), z1z1);                      \
        if (F##_eq(p->x, u2) && F##_eq(p->y, s2)) { G##_double(p); return; }                      \
        F h = F##_sub(u2, p->x), hh = F##_sqr(h), i = F##_dbl(F##_dbl(hh)), j = F##_mul(h, i);    \
        F r = F##_dbl(F##_sub(s2, p->y)), v = F##_mul(p->x, i);                                   \
        F x3 = F##_sub(F##_sub(F##_sub(F##_sqr(r), j), v), v);                                    \
        F y3 = F##_sub(F##_mul(F##_sub(v, x3), r), F##_dbl(F##_mul(p->y, j)));                    \
        F z3 = F##_sub(F##_sub(F##_sqr(F##_add(p->z, h)), z1z1), hh);                             \
        p->x = x3; p->y = y3; p->z = z3;                                                          \
    }                                                                                             \
    /* into_affine, ec.rs:586-618 */                                                              \
    static G##_affine G##_to_affine(const G##_t* p) {                                             \
        G##_affine r;                                                                             \
        if (G##_is_zero(p)) { r.x = F##_zero(); r.y = F##_one(); r.inf = 1; return r; }           \
        F zi = F##_inv(p->z), zi2 = F##_sqr(zi);                                                  \
        r.x = F##_mul(p->x, zi2); r.y = F##_mul(p->y, F##_mul(zi2, zi)); r.inf = 0;               \
        return r;                                                                                 \
    }                                                                                             \
    /* mul_assign by a plain 256-bit integer, ec.rs:534-553 */                                    \
    static G##_t G##_mul(const G##_t* p, const uint64_t k[4]) {                                   \
        G##_t r = G##_zero();                                                                     \
        int found = 0;                                                                            \
        for (int i = 3; i >= 0; i--)                                                              \
            for (int b = 63; b >= 0; b--) {                                                       \
                if (found) G##_double(&r);                                                        \
                if ((k[i] >> b) & 1) { G##_add(&r, p); found = 1; }                               \
            }                                                                                     \
        return r;                                                                                 \
    }                                                                                             \
    static G##_t G##_from_affine(const G##_affine* a) {                                           \
        if (a->inf) return G##_zero();                                                            \
        G##_t r = {a->x, a->y, F##_one()};                                                        \
        return r;                                                                                 \
    }

CURVE_IMPL(g1, fq)
CURVE_IMPL(g2, fq2)

/* ------------------------------------------------------------------ encodings */
static int fq_read_be(const uint8_t* b, fq* out) {
    fq v;
    for (int i = 0; i < 6; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | b[(5 - i) * 8 + j];
        v.l[i] = w;
    }
    if (geq(v.l, FQ.p, 6)) return 0;
    f_to_mont(Q, out->l, v.l);
    return 1;
}
static void fq_write_be(fq a, uint8_t* b) {
    fq v;
    f_from_mont(Q, v.l, a.l);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(v.l[i] >> (56 - 8 * j));
}
static int fq_plain_gt(fq a, fq b) {
    fq x, y;
    f_from_mont(Q, x.l, a.l);
    f_from_mont(Q, y.l, b.l);
    for (int i = 5; i >= 0; i--) {
        if (x.l[i] > y.l[i]) return 1;
        if (x.l[i] < y.l[i]) return 0;
    }
    return 0;
}
static int g1_read_uncompressed(const uint8_t* b, g1_affine* p) {
    if (b[0] & 0x80) return 0;
    if (b[0] & 0x40) { p->x = fq_zero(); p->y = fq_one(); p->inf = 1; return 1; }
    p->inf = 0;
    return fq_read_be(b, &p->x) && fq_read_be(b + 48, &p->y);
}
static int g2_read_uncompressed(const uint8_t* b, g2_affine* p) {
    if (b[0] & 0x80) return 0;
    if (b[0] & 0x40) { p->x = fq2_zero(); p->y = fq2_one(); p->inf = 1; return 1; }
    p->inf = 0;
    return fq_read_be(b, &p->x.c1) && fq_read_be(b + 48, &p->x.c0) && fq_read_be(b + 96, &p->y.c1) &&
           fq_read_be(b + 144, &p->y.c0);
}
static void g1_write_uncompressed(const g1_affine* p, uint8_t* b) {
    memset(b, 0, 96);
    if (p->inf) { b[0] = 0x40; return; }
    fq_write_be(p->x, b);
    fq_write_be(p->y, b + 48);
}
static void g2_write_uncompressed(const g2_affine* p, uint8_t* b) {
    memset(b, 0, 192);
    if (p->inf) { b[0] = 0x40; return; }
    fq_write_be(p->x.c1, b);
    fq_write_be(p->x.c0, b + 48);
    fq_write_be(p->y.c1, b + 96);
    fq_write_be(p->y.c0, b + 144);
}
static void g1_write_compressed(const g1_affine* p, uint8_t* b) {
    memset(b, 0, 48);
    if (p->inf) { b[0] = 0xc0; return; }
    fq_write_be(p->x, b);
    if (fq_plain_gt(p->y, fq_neg(p->y))) b[0] |= 0x20;
    b[0] |= 0x80;
}
static void g2_write_compressed(const g2_affine* p, uint8_t* b) {
    memset(b, 0, 96);
    if (p->inf) { b[0] = 0xc0; return; }
    fq_write_be(p->x.c1, b);
    fq_write_be(p->x.c0, b + 48);
    fq2 ny = fq2_neg(p->y);
    int gt = fq_eq(p->y.c1, ny.c1) ? fq_plain_gt(p->y.c0, ny.c0) : fq_plain_gt(p->y.c1, ny.c1);
    if (gt) b[0] |= 0x20;
    b[0] |= 0x80;
}

static void scalar_read_le(const uint8_t* b, uint64_t k[4]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; j--) w = (w << 8) | b[i * 8 + j];
        k[i] = w;
    }
}
static void scalar_write_le(const uint64_t k[4], uint8_t* b) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[i * 8 + j] = (uint8_t)(k[i] >> (8 * j));
}

/* ------------------------------------------------------------------ worker pool stand-in:
 * bellman's Worker runs one task per multiexp window and chunked FFT tasks on a CpuPool of
 * num_cpus threads; here `threads` pthreads pull task indices from a shared counter. */
typedef void (*task_fn)(void* ctx, int idx);
typedef struct {
    task_fn fn;
    void* ctx;
    int n_tasks;
    int next;
    pthread_mutex_t mu;
} pool_t;
static void* pool_main(void* arg) {
    pool_t* p = (pool_t*)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        int i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n_tasks) break;
        p->fn(p->ctx, i);
    }
    return NULL;
}
static void run_tasks(task_fn fn, void* ctx, int n_tasks, int threads) {
    if (threads < 1) threads = 1;
    if (threads > n_tasks) threads = n_tasks;
    pool_t p = {fn, ctx, n_tasks, 0, PTHREAD_MUTEX_INITIALIZER};
    if (threads <= 1) {
        for (int i = 0; i < n_tasks; i++) fn(ctx, i);
        return;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, pool_main, &p);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
}

/* ------------------------------------------------------------------ multiexp (bellman multiexp.rs)
 * c = 3 if n < 32 else ceil(ln n); one task per window [skip, skip+c); zero exponents skip
 * their base, an exponent equal to one is added straight to the accumulator in the FIRST window
 * only (handle_trivial), buckets are summed by parts; windows are folded high to low with c
 * doublings each.  `density` (optional, one byte per exponent) selects which exponents consume
 * a base (DensityTracker source); bases are consumed in order from `first_base`. */
static int is_one4(const uint64_t* k) { return k[0] == 1 && !k[1] && !k[2] && !k[3]; }
static int is_zero4(const uint64_t* k) { return !(k[0] | k[1] | k[2] | k[3]); }
static uint32_t window_of(const uint64_t* k, uint32_t skip, uint32_t c) {
    /* exp.shr(skip); exp.as_ref()[0] % (1 << c) */
    uint32_t w = skip >> 6, sh = skip & 63;
    uint64_t v = k[w] >> sh;
    if (sh && w + 1 < 4) v |= k[w + 1] << (64 - sh);
    return (uint32_t)(v & ((1ull << c) - 1));
}
static uint32_t window_size(size_t n) {
    if (n < 32) return 3;
    return (uint32_t)ceil(log((double)n));
}

#define MULTIEXP_IMPL(G)                                                                           \
    typedef struct {                                                                               \
        const G##_affine* bases;                                                                   \
        const uint8_t* density;                                                                    \
        const uint64_t* exps;                                                                      \
        size_t n;                                                                                  \
        uint32_t c;                                                                                \
        G##_t* window_sums;                                                                        \
    } G##_mexp_ctx;                                                                                \
    static void G##_mexp_window(void* vctx, int widx) {                                            \
        G##_mexp_ctx* x = (G##_mexp_ctx*)vctx;                                                     \
        const uint32_t c = x->c, skip = (uint32_t)widx * c;                                        \
        const int handle_trivial = widx == 0;                                                      \
        size_t nb = ((size_t)1 << c) - 1;                                                          \
        G##_t* buckets = (G##_t*)malloc(sizeof(G##_t) * nb);                                       \
        for (size_t i = 0; i < nb; i++) buckets[i] = G##_zero();                                   \
        G##_t acc = G##_zero();                                                                    \
        size_t bi = 0;                                                                             \
        for (size_t i = 0; i < x->n; i++) {                                                        \
            if (x->density && !x->density[i]) continue;                                            \
            const uint64_t* e = x->exps + 4 * i;                                                   \
            const G##_affine* base = &x->bases[bi++];                                              \
            if (is_zero4(e)) continue;                                                             \
            if (is_one4(e)) {                                                                      \
                if (handle_trivial) G##_add_mixed(&acc, base);                                     \
                continue;                                                                          \
            }                                                                                      \
            uint32_t d = window_of(e, skip, c);                                                    \
            if (d) G##_add_mixed(&buckets[d - 1], base);                                           \
        }                                                                                          \
        G##_t running = G##_zero();                                                                \
        for (size_t i = nb; i-- > 0;) {                                                            \
            G##_add(&running, &buckets[i]);                                                        \
            G##_add(&acc, &running);                                                               \
        }                                                                                          \
        free(buckets);                                                                             \
        x->window_sums[widx] = acc;                                                                \
    }                                                                                              \
    static G##_t G##_multiexp(const G##_affine* bases, const uint8_t* density, const uint64_t* exps, \
                              size_t n, int threads) {                                             \
        uint32_t c = window_size(n);                                                               \
        int nw = (int)((255 + c - 1) / c); /* while skip < Fr::NUM_BITS (255) */                   \
        G##_t* sums = (G##_t*)malloc(sizeof(G##_t) * nw);                                          \
        G##_mexp_ctx ctx = {bases, density, exps, n, c, sums};                                     \
        run_tasks(G##_mexp_window, &ctx, nw, threads);                                             \
        G##_t acc = sums[nw - 1];                                                                  \
        for (int w = nw - 2; w >= 0; w--) {                                                        \
            for (uint32_t k = 0; k < c; k++) G##_double(&acc);                                     \
            G##_add(&acc, &sums[w]);                                                               \
        }                                                                                          \
        free(sums);                                                                                \
        return acc;                                                                                \
    }

MULTIEXP_IMPL(g1)
MULTIEXP_IMPL(g2)

/* ------------------------------------------------------------------ EvaluationDomain (domain.rs) */
static inline fr fr_add_(fr a, fr b) { fr r; f_add(&FR, r.l, a.l, b.l); return r; }
static inline fr fr_sub_(fr a, fr b) { fr r; f_sub(&FR, r.l, a.l, b.l); return r; }
static inline fr fr_mul_(fr a, fr b) { fr r; f_mul(&FR, r.l, a.l, b.l); return r; }
static inline fr fr_one_(void) { fr r; memcpy(r.l, FR.r, 32); return r; }
static fr fr_pow_u64(fr a, uint64_t e) {
    fr r = fr_one_();
    while (e) {
        if (e & 1) r = fr_mul_(r, a);
        a = fr_mul_(a, a);
        e >>= 1;
    }
    return r;
}
static fr fr_inv_(fr a) { fr r; f_inv(&FR, r.l, a.l); return r; }
static const uint64_t FR_ROOT[4] = {0xb9b58d8c5f0e466aull, 0x5b1b4c801819d7ecull, 0x0af53ae352a31e64ull, 0x5bf3adda19e9b27bull};
static const uint64_t FR_GEN[4] = {0x0000000efffffff1ull, 0x17e363d300189c0full, 0xff9c57876f8457b0ull, 0x351332208fc5a8c4ull};

static uint32_t bitreverse(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; i++) {
        r = (r << 1) | (n & 1);
        n >>= 1;
    }
    return r;
}
/* serial_fft */
static void serial_fft(fr* a, size_t n, fr omega, uint32_t log_n) {
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rk = bitreverse(k, log_n);
        if (k < rk) { fr t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    size_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        fr w_m = fr_pow_u64(omega, n / (2 * m));
        for (size_t k = 0; k < n; k += 2 * m) {
            fr w = fr_one_();
            for (size_t j = 0; j < m; j++) {
                fr t = fr_mul_(a[k + j + m], w);
                fr tmp = fr_sub_(a[k + j], t);
                a[k + j + m] = tmp;
                a[k + j] = fr_add_(a[k + j], t);
                w = fr_mul_(w, w_m);
            }
        }
        m *= 2;
    }
}
/* parallel_fft: 2^log_cpus interleaved sub-transforms, then recombination */
typedef struct {
    fr* a;
    fr* tmp;
    size_t n;
    fr omega;
    uint32_t log_n, log_cpus;
} pfft_ctx;
static void pfft_task(void* vctx, int j) {
    pfft_ctx* x = (pfft_ctx*)vctx;
    const uint32_t log_new_n = x->log_n - x->log_cpus;
    const size_t num_cpus = (size_t)1 << x->log_cpus, new_n = (size_t)1 << log_new_n;
    fr* t = x->tmp + (size_t)j * new_n;
    fr omega_j = fr_pow_u64(x->omega, (uint64_t)j);
    fr omega_step = fr_pow_u64(x->omega, (uint64_t)j << log_new_n);
    fr new_omega = fr_pow_u64(x->omega, num_cpus);
    fr elt = fr_one_();
    for (size_t i = 0; i < new_n; i++) {
        fr acc;
        memset(&acc, 0, sizeof acc);
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = (i + (s << log_new_n)) % x->n;
            fr v = fr_mul_(x->a[idx], elt);
            acc = fr_add_(acc, v);
            elt = fr_mul_(elt, omega_step);
        }
        elt = fr_mul_(elt, omega_j);
        t[i] = acc;
    }
    serial_fft(t, new_n, new_omega, log_new_n);
}
static void best_fft(fr* a, size_t n, fr omega, uint32_t log_n, int threads) {
    uint32_t log_cpus = 0;
    while ((1 << (log_cpus + 1)) <= threads) log_cpus++;
    if (log_n <= log_cpus || threads <= 1) {
        serial_fft(a, n, omega, log_n);
        return;
    }
    fr* tmp = (fr*)malloc(sizeof(fr) * n);
    pfft_ctx ctx = {a, tmp, n, omega, log_n, log_cpus};
    run_tasks(pfft_task, &ctx, 1 << log_cpus, threads);
    const size_t mask = ((size_t)1 << log_cpus) - 1;
    for (size_t idx = 0; idx < n; idx++) a[idx] = tmp[((idx & mask) << (log_n - log_cpus)) + (idx >> log_cpus)];
    free(tmp);
}
typedef struct { fr omega, omegainv, geninv, minv; uint32_t exp; size_t m; } domain_t;
static void domain_init(domain_t* d, size_t len) {
    d->m = 1;
    d->exp = 0;
    while (d->m < len) { d->m *= 2; d->exp++; }
    memcpy(d->omega.l, FR_ROOT, 32);
    for (uint32_t i = d->exp; i < 32; i++) d->omega = fr_mul_(d->omega, d->omega);
    d->omegainv = fr_inv_(d->omega);
    fr g;
    memcpy(g.l, FR_GEN, 32);
    d->geninv = fr_inv_(g);
    fr mm;
    uint64_t mv[4] = {d->m, 0, 0, 0};
    f_to_mont(&FR, mm.l, mv);
    d->minv = fr_inv_(mm);
}
static void distribute_powers(fr* a, size_t n, fr g) {
    fr u = fr_one_();
    for (size_t i = 0; i < n; i++) {
        a[i] = fr_mul_(a[i], u);
        u = fr_mul_(u, g);
    }
}
static void dom_fft(const domain_t* d, fr* a, int threads) { best_fft(a, d->m, d->omega, d->exp, threads); }
static void dom_ifft(const domain_t* d, fr* a, int threads) {
    best_fft(a, d->m, d->omegainv, d->exp, threads);
    for (size_t i = 0; i < d->m; i++) a[i] = fr_mul_(a[i], d->minv);
}
static void dom_coset_fft(const domain_t* d, fr* a, int threads) {
    fr g;
    memcpy(g.l, FR_GEN, 32);
    distribute_powers(a, d->m, g);
    dom_fft(d, a, threads);
}
static void dom_icoset_fft(const domain_t* d, fr* a, int threads) {
    dom_ifft(d, a, threads);
    distribute_powers(a, d->m, d->geninv);
}

/* ------------------------------------------------------------------ exported API (ctypes) */
static fr* load_fr_vec(const uint8_t* le, size_t n, size_t m) {
    fr* v = (fr*)calloc(m ? m : 1, sizeof(fr));
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4];
        scalar_read_le(le + 32 * i, k);
        f_to_mont(&FR, v[i].l, k);
    }
    return v;
}

/* fft / ifft / coset_fft / icoset_fft on n = 2^log_n plain LE scalars, in place */
int zo_fft(uint8_t* data, uint32_t log_n, int inverse, int coset, int threads) {
    size_t n = (size_t)1 << log_n;
    domain_t d;
    domain_init(&d, n);
    fr* a = load_fr_vec(data, n, n);
    if (!inverse && !coset) dom_fft(&d, a, threads);
    else if (inverse && !coset) dom_ifft(&d, a, threads);
    else if (!inverse) dom_coset_fft(&d, a, threads);
    else dom_icoset_fft(&d, a, threads);
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4];
        f_from_mont(&FR, k, a[i].l);
        scalar_write_le(k, data + 32 * i);
    }
    free(a);
    return 0;
}

static g1_affine* load_g1_vec(const uint8_t* b, size_t n, int* ok) {
    g1_affine* v = (g1_affine*)malloc(sizeof(g1_affine) * (n ? n : 1));
    for (size_t i = 0; i < n; i++)
        if (!g1_read_uncompressed(b + 96 * i, &v[i])) *ok = 0;
    return v;
}
static g2_affine* load_g2_vec(const uint8_t* b, size_t n, int* ok) {
    g2_affine* v = (g2_affine*)malloc(sizeof(g2_affine) * (n ? n : 1));
    for (size_t i = 0; i < n; i++)
        if (!g2_read_uncompressed(b + 192 * i, &v[i])) *ok = 0;
    return v;
}
static uint64_t* load_scalars(const uint8_t* le, size_t n) {
    uint64_t* v = (uint64_t*)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) scalar_read_le(le + 32 * i, v + 4 * i);
    return v;
}

/* Opaque pre-decoded inputs, so that a timed multiexp does not include byte decoding */
typedef struct { g1_affine* b1; g2_affine* b2; size_t n; } zo_bases;
zo_bases* zo_bases_load(int group, const uint8_t* bytes, size_t n) {
    int ok = 1;
    zo_bases* z = (zo_bases*)calloc(1, sizeof(zo_bases));
    z->n = n;
    if (group == 1) z->b1 = load_g1_vec(bytes, n, &ok);
    else z->b2 = load_g2_vec(bytes, n, &ok);
    if (!ok) { free(z->b1); free(z->b2); free(z); return NULL; }
    return z;
}
void zo_bases_free(zo_bases* z) {
    if (!z) return;
    free(z->b1);
    free(z->b2);
    free(z);
}
/* multiexp(FullDensity) over pre-decoded bases; scalars plain LE; out uncompressed */
int zo_multiexp(const zo_bases* z, const uint8_t* scalars, size_t n, int threads, uint8_t* out) {
    if (n > z->n) return 1;
    uint64_t* e = load_scalars(scalars, n);
    if (z->b1) {
        g1_t r = g1_multiexp(z->b1, NULL, e, n, threads);
        g1_affine a = g1_to_affine(&r);
        g1_write_uncompressed(&a, out);
    } else {
        g2_t r = g2_multiexp(z->b2, NULL, e, n, threads);
        g2_affine a = g2_to_affine(&r);
        g2_write_uncompressed(&a, out);
    }
    free(e);
    return 0;
}

/* Parsed Parameters (bellman Parameters::read, unchecked) */
typedef struct {
    g1_affine alpha_g1, beta_g1, delta_g1;
    g2_affine beta_g2, gamma_g2, delta_g2;
    uint32_t n_ic, n_h, n_l, n_a, n_b1, n_b2;
    g1_affine *ic, *h, *l, *a, *b1;
    g2_affine* b2;
} zo_params;

static int rd_u32(const uint8_t** p, size_t* left, uint32_t* v) {
    if (*left < 4) return 0;
    *v = ((uint32_t)(*p)[0] << 24) | ((uint32_t)(*p)[1] << 16) | ((uint32_t)(*p)[2] << 8) | (*p)[3];
    *p += 4;
    *left -= 4;
    return 1;
}
static int rd_g1(const uint8_t** p, size_t* left, g1_affine* o) {
    if (*left < 96 || !g1_read_uncompressed(*p, o) || o->inf) return 0;
    *p += 96;
    *left -= 96;
    return 1;
}
static int rd_g2(const uint8_t** p, size_t* left, g2_affine* o) {
    if (*left < 192 || !g2_read_uncompressed(*p, o) || o->inf) return 0;
    *p += 192;
    *left -= 192;
    return 1;
}
static int rd_g1v(const uint8_t** p, size_t* left, uint32_t* n, g1_affine** v) {
    if (!rd_u32(p, left, n)) return 0;
    if ((size_t)*n * 96 > *left) return 0;
    *v = (g1_affine*)malloc(sizeof(g1_affine) * (*n ? *n : 1));
    for (uint32_t i = 0; i < *n; i++)
        if (!rd_g1(p, left, &(*v)[i])) return 0;
    return 1;
}
void zo_params_free(zo_params* P) {
    if (!P) return;
    free(P->ic); free(P->h); free(P->l); free(P->a); free(P->b1); free(P->b2);
    free(P);
}
zo_params* zo_params_read(const uint8_t* pk, size_t len) {
    zo_params* P = (zo_params*)calloc(1, sizeof(zo_params));
    const uint8_t* p = pk;
    size_t left = len;
    int ok = rd_g1(&p, &left, &P->alpha_g1) && rd_g1(&p, &left, &P->beta_g1) && rd_g2(&p, &left, &P->beta_g2) &&
             rd_g2(&p, &left, &P->gamma_g2) && rd_g1(&p, &left, &P->delta_g1) && rd_g2(&p, &left, &P->delta_g2) &&
             rd_g1v(&p, &left, &P->n_ic, &P->ic) && rd_g1v(&p, &left, &P->n_h, &P->h) &&
             rd_g1v(&p, &left, &P->n_l, &P->l) && rd_g1v(&p, &left, &P->n_a, &P->a) && rd_g1v(&p, &left, &P->n_b1, &P->b1);
    if (ok) ok = rd_u32(&p, &left, &P->n_b2) && (size_t)P->n_b2 * 192 <= left;
    if (ok) {
        P->b2 = (g2_affine*)malloc(sizeof(g2_affine) * (P->n_b2 ? P->n_b2 : 1));
        for (uint32_t i = 0; ok && i < P->n_b2; i++) ok = rd_g2(&p, &left, &P->b2[i]);
    }
    if (!ok) { zo_params_free(P); return NULL; }
    return P;
}

/* create_proof (bellman prover.rs) from a finished ProvingAssignment.
 * a, b, c: n_rows plain LE scalars; inputs / aux plain LE; densities one byte per variable.
 * Returns 0, or the SynthesisError code (5 = UnexpectedIdentity, 6 = IoError, 4 = degree). */
int zo_create_proof(const zo_params* P, uint32_t n_rows, const uint8_t* a_le, const uint8_t* b_le,
                    const uint8_t* c_le, uint32_t n_in, const uint8_t* inputs_le, uint32_t n_aux,
                    const uint8_t* aux_le, const uint8_t* a_aux_d, const uint8_t* b_in_d, const uint8_t* b_aux_d,
                    const uint8_t* r_le, const uint8_t* s_le, int threads, uint8_t* proof_out) {
    domain_t d;
    domain_init(&d, n_rows);
    if (d.m - 1 > P->n_h) return 4;
    if (n_aux > P->n_l || n_in > P->n_a) return 6;
    /* step 3: h */
    fr* a = load_fr_vec(a_le, n_rows, d.m);
    fr* b = load_fr_vec(b_le, n_rows, d.m);
    fr* c = load_fr_vec(c_le, n_rows, d.m);
    dom_ifft(&d, a, threads); dom_coset_fft(&d, a, threads);
    dom_ifft(&d, b, threads); dom_coset_fft(&d, b, threads);
    dom_ifft(&d, c, threads); dom_coset_fft(&d, c, threads);
    fr g, zi;
    memcpy(g.l, FR_GEN, 32);
    zi = fr_inv_(fr_sub_(fr_pow_u64(g, d.m), fr_one_())); /* divide_by_z_on_coset */
    for (size_t i = 0; i < d.m; i++) a[i] = fr_mul_(fr_sub_(fr_mul_(a[i], b[i]), c[i]), zi);
    dom_icoset_fft(&d, a, threads);
    size_t hn = d.m - 1;
    uint64_t* hc = (uint64_t*)malloc(32 * hn);
    for (size_t i = 0; i < hn; i++) f_from_mont(&FR, hc + 4 * i, a[i].l);
    free(b); free(c); free(a);
    g1_t h = g1_multiexp(P->h, NULL, hc, hn, threads);
    free(hc);
    /* step 4 */
    uint64_t* in = load_scalars(inputs_le, n_in);
    uint64_t* aux = load_scalars(aux_le, n_aux);
    g1_t l = g1_multiexp(P->l, NULL, aux, n_aux, threads);
    uint32_t a_aux_total = 0, b_in_total = 0, b_aux_total = 0;
    for (uint32_t i = 0; i < n_aux; i++) { a_aux_total += a_aux_d[i] != 0; b_aux_total += b_aux_d[i] != 0; }
    for (uint32_t i = 0; i < n_in; i++) b_in_total += b_in_d[i] != 0;
    if (n_in + a_aux_total > P->n_a || b_in_total + b_aux_total > P->n_b1 || b_in_total + b_aux_total > P->n_b2) {
        free(in); free(aux);
        return 6;
    }
    g1_t a_inputs = g1_multiexp(P->a, NULL, in, n_in, threads);
    g1_t a_aux = g1_multiexp(P->a + n_in, a_aux_d, aux, n_aux, threads);
    g1_t b1_inputs = g1_multiexp(P->b1, b_in_d, in, n_in, threads);
    g1_t b1_aux = g1_multiexp(P->b1 + b_in_total, b_aux_d, aux, n_aux, threads);
    g2_t b2_inputs = g2_multiexp(P->b2, b_in_d, in, n_in, threads);
    g2_t b2_aux = g2_multiexp(P->b2 + b_in_total, b_aux_d, aux, n_aux, threads);
    free(in); free(aux);
    if (P->delta_g1.inf || P->delta_g2.inf) return 5;
    /* step 6 */
    uint64_t r[4], s[4], rs[4];
    scalar_read_le(r_le, r);
    scalar_read_le(s_le, s);
    fr rm, sm, rsm;
    f_to_mont(&FR, rm.l, r);
    f_to_mont(&FR, sm.l, s);
    rsm = fr_mul_(rm, sm);
    f_from_mont(&FR, rs, rsm.l);
    g1_t dg1 = g1_from_affine(&P->delta_g1), ag1 = g1_from_affine(&P->alpha_g1), bg1 = g1_from_affine(&P->beta_g1);
    g2_t dg2 = g2_from_affine(&P->delta_g2);
    g1_t g_a = g1_mul(&dg1, r);
    g1_add_mixed(&g_a, &P->alpha_g1);
    g2_t g_b = g2_mul(&dg2, s);
    g2_add_mixed(&g_b, &P->beta_g2);
    g1_t g_c = g1_mul(&dg1, rs);
    g1_t t = g1_mul(&ag1, s);
    g1_add(&g_c, &t);
    t = g1_mul(&bg1, r);
    g1_add(&g_c, &t);
    g1_t a_answer = a_inputs;
    g1_add(&a_answer, &a_aux);
    g1_add(&g_a, &a_answer);
    a_answer = g1_mul(&a_answer, s);
    g1_add(&g_c, &a_answer);
    g1_t b1_answer = b1_inputs;
    g1_add(&b1_answer, &b1_aux);
    g2_t b2_answer = b2_inputs;
    g2_add(&b2_answer, &b2_aux);
    g2_add(&g_b, &b2_answer);
    b1_answer = g1_mul(&b1_answer, r);
    g1_add(&g_c, &b1_answer);
    g1_add(&g_c, &h);
    g1_add(&g_c, &l);
    g1_affine pa = g1_to_affine(&g_a), pc = g1_to_affine(&g_c);
    g2_affine pb = g2_to_affine(&g_b);
    g1_write_compressed(&pa, proof_out);
    g2_write_compressed(&pb, proof_out + 48);
    g1_write_compressed(&pc, proof_out + 144);
    return 0;
}

/* Many independent proofs, one single-threaded create_proof per pool thread: the
 * throughput-optimal way to use a many-core host for a batch (each proof's multiexp parallelism
 * is capped by its window count, so a batch scales better across proofs than inside one). */
typedef struct {
    const zo_params* P;
    uint32_t n_rows, n_in, n_aux;
    const uint8_t *a, *b, *c, *in, *aux, *ad, *bid, *bad, *rs;
    uint8_t* out;
    int* rc;
} many_ctx;
static void many_task(void* vctx, int i) {
    many_ctx* x = (many_ctx*)vctx;
    x->rc[i] = zo_create_proof(x->P, x->n_rows, x->a, x->b, x->c, x->n_in, x->in, x->n_aux, x->aux, x->ad, x->bid,
                               x->bad, x->rs + 64 * (size_t)i, x->rs + 64 * (size_t)i + 32, 1, x->out + 192 * (size_t)i);
}
/* n proofs of the SAME assignment with per-proof (r, s) = rs[64 i .. 64 i + 64) */
int zo_create_proofs_parallel(const zo_params* P, int n, uint32_t n_rows, const uint8_t* a_le, const uint8_t* b_le,
                              const uint8_t* c_le, uint32_t n_in, const uint8_t* inputs_le, uint32_t n_aux,
                              const uint8_t* aux_le, const uint8_t* a_aux_d, const uint8_t* b_in_d, const uint8_t* b_aux_d,
                              const uint8_t* rs, int threads, uint8_t* proofs_out) {
    int* rc = (int*)calloc(n > 0 ? n : 1, sizeof(int));
    many_ctx ctx = {P, n_rows, n_in, n_aux, a_le, b_le, c_le, inputs_le, aux_le, a_aux_d, b_in_d, b_aux_d, rs, proofs_out, rc};
    run_tasks(many_task, &ctx, n, threads);
    int bad = 0;
    for (int i = 0; i < n; i++) bad |= rc[i];
    free(rc);
    return bad;
}

/* ------------------------------------------------------------------ fixture generation helpers
 * (not part of any restated algorithm): k*G for many k with a fixed-base 8-bit window table. */
static const uint64_t G1X[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
static const uint64_t G1Y[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
static const uint64_t G2XC0[6] = {0xf5f28fa202940a10ull, 0xb3f5fb2687b4961aull, 0xa1a893b53e2ae580ull, 0x9894999d1a3caee9ull, 0x6f67b7631863366bull, 0x058191924350bcd7ull};
static const uint64_t G2XC1[6] = {0xa5a9c0759e23f606ull, 0xaaa0c59dbccd60c3ull, 0x3bb17e18e2867806ull, 0x1b1ab6cc8541b367ull, 0xc2b6ed0ef2158547ull, 0x11922a097360edf3ull};
static const uint64_t G2YC0[6] = {0x4c730af860494c4aull, 0x597cfa1f5e369c5aull, 0xe7e6856caa0a635aull, 0xbbefb5e96e0d495full, 0x07d3a975f0ef25a2ull, 0x0083fd8e7e80dae5ull};
static const uint64_t G2YC1[6] = {0xadc0fc92df64b05dull, 0x18aa270a2b1461dcull, 0x86adac6a3be4eba0ull, 0x79495c4ec93da33aull, 0xe7175850a43ccaedull, 0x0b2bc2a163de1bf2ull};

typedef struct { g1_t* t1; g2_t* t2; } fixed_tab; /* [32][256] */
static fixed_tab g_tab;
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;
static void build_tab(void) {
    g_tab.t1 = (g1_t*)malloc(sizeof(g1_t) * 32 * 256);
    g_tab.t2 = (g2_t*)malloc(sizeof(g2_t) * 32 * 256);
    g1_t b1 = {{{0}}, {{0}}, fq_one()};
    memcpy(b1.x.l, G1X, 48);
    memcpy(b1.y.l, G1Y, 48);
    g2_t b2;
    memcpy(b2.x.c0.l, G2XC0, 48); memcpy(b2.x.c1.l, G2XC1, 48);
    memcpy(b2.y.c0.l, G2YC0, 48); memcpy(b2.y.c1.l, G2YC1, 48);
    b2.z = fq2_one();
    for (int w = 0; w < 32; w++) {
        g_tab.t1[w * 256] = g1_zero();
        g_tab.t2[w * 256] = g2_zero();
        for (int k = 1; k < 256; k++) {
            g_tab.t1[w * 256 + k] = g_tab.t1[w * 256 + k - 1];
            g1_add(&g_tab.t1[w * 256 + k], &b1);
            g_tab.t2[w * 256 + k] = g_tab.t2[w * 256 + k - 1];
            g2_add(&g_tab.t2[w * 256 + k], &b2);
        }
        for (int k = 0; k < 8; k++) { g1_double(&b1); g2_double(&b2); }
    }
}
static g1_t fixed_mul_g1(const uint64_t k[4]) {
    g1_t r = g1_zero();
    for (int w = 0; w < 32; w++) {
        unsigned byte = (unsigned)(k[w / 8] >> (8 * (w % 8))) & 0xff;
        if (byte) g1_add(&r, &g_tab.t1[w * 256 + byte]);
    }
    return r;
}
static g2_t fixed_mul_g2(const uint64_t k[4]) {
    g2_t r = g2_zero();
    for (int w = 0; w < 32; w++) {
        unsigned byte = (unsigned)(k[w / 8] >> (8 * (w % 8))) & 0xff;
        if (byte) g2_add(&r, &g_tab.t2[w * 256 + byte]);
    }
    return r;
}
typedef struct { int group; const uint8_t* k; uint8_t* out; size_t n; int chunks; } fm_ctx;
static void fm_task(void* vctx, int idx) {
    fm_ctx* x = (fm_ctx*)vctx;
    size_t lo = x->n * idx / x->chunks, hi = x->n * (idx + 1) / x->chunks;
    for (size_t i = lo; i < hi; i++) {
        uint64_t k[4];
        scalar_read_le(x->k + 32 * i, k);
        if (x->group == 1) {
            g1_t p = fixed_mul_g1(k);
            g1_affine a = g1_to_affine(&p);
            g1_write_uncompressed(&a, x->out + 96 * i);
        } else {
            g2_t p = fixed_mul_g2(k);
            g2_affine a = g2_to_affine(&p);
            g2_write_uncompressed(&a, x->out + 192 * i);
        }
    }
}
/* out[i] = scalars[i] * generator, uncompressed (group 1: 96 B each, group 2: 192 B each) */
int zo_fixed_base_mul(int group, const uint8_t* scalars, size_t n, int threads, uint8_t* out) {
    pthread_once(&g_tab_once, build_tab);
    int chunks = threads > 1 ? threads * 4 : 1;
    if ((size_t)chunks > n) chunks = n ? (int)n : 1;
    fm_ctx ctx = {group, scalars, out, n, chunks};
    run_tasks(fm_task, &ctx, chunks, threads);
    return 0;
}
