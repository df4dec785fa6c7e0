// Device-side prime-field arithmetic for gfx950 (CDNA4): Montgomery form, 32-bit limbs.
//
// Spec followed: the reference's vendored arithmetic crate
//   Fr: core/pairing/src/bls12_381/fr.rs:341-571  (4 x u64, R = 2^256)
//   Fq: core/pairing/src/bls12_381/fq.rs:749-1127 (6 x u64, R = 2^384)
//   Fq2: core/pairing/src/bls12_381/fq2.rs:90-182
// Same values, same Montgomery radix; a little-endian array of 2k u32 limbs is byte-identical
// to the reference's k u64 limbs, so device buffers can be compared word-for-word with the
// literal KATs in fr.rs / fq.rs.
//
// CDNA4 has no 64x64 multiplier in the VALU: the unit of work is v_mad_u64_u32
// (32x32+64 -> 64).  mont_mul is a CIOS loop arranged so that every limb product is ONE
// v_mad_u64_u32 whose 64-bit addend carries the accumulator limb, followed by ONE
// v_addc_co_u32 in a carry chain: 2N^2 + N multiplier ops and ~2N^2 adds per product.
#pragma once
#include <stdint.h>
#ifdef ZK_EMU
#include <stdio.h>
#include <stdlib.h>
#endif
#include "gpu_rt.h"
#include "consts.h"
#ifndef ZK_EMU
#include "mul_asm.h"
#endif

namespace zkdev {

#define ZK_DI __device__ __forceinline__
// The Montgomery product is ~1000 instructions (8 KB of code for Fq).  A point addition inlines
// 10-14 of them (x3 for Fq2), which overflows the 64 KB instruction cache two CUs share, so the
// product is a real function by default and the curve formulas call it.
#ifndef ZK_MUL_ATTR
#ifdef ZK_EMU
#define ZK_MUL_ATTR inline
#else
#define ZK_MUL_ATTR __device__ __attribute__((noinline))
#endif
#endif

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));

struct FrCfg {
    static constexpr int N = 8;
    typedef u32x8 vec;
    static constexpr uint32_t P[8] = ZK_FR_P_32;
    static constexpr uint32_t R[8] = ZK_FR_R_32;     // Montgomery one
    static constexpr uint32_t R2[8] = ZK_FR_R2_32;
    static constexpr uint32_t INV = ZK_FR_INV32;
};
struct FqCfg {
    static constexpr int N = 12;
    typedef u32x12 vec;
    static constexpr uint32_t P[12] = ZK_FQ_P_32;
    static constexpr uint32_t R[12] = ZK_FQ_R_32;
    static constexpr uint32_t R2[12] = ZK_FQ_R2_32;
    static constexpr uint32_t INV = ZK_FQ_INV32;
};

template <class C>
struct Fp {
    static constexpr int N = C::N;
    static constexpr int MO = 1;   // fully reduced
    static constexpr int WB = 64;  // wr() is the identity
    uint32_t l[N];
    ZK_DI bool is_zero_norm() const { return is_zero(); }

    ZK_DI static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    ZK_DI static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = C::R[i];
        return r;
    }
    ZK_DI static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = C::R2[i];
        return r;
    }
    ZK_DI bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    ZK_DI bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    ZK_DI bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = (a + b) mod p ; inputs < p (both moduli leave >= 1 spare bit, so a + b cannot carry out)
template <class C>
ZK_DI Fp<C> add(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    uint32_t t[N], s[N], cy = 0, co;
#pragma unroll
    for (int i = 0; i < N; i++) {
        t[i] = __builtin_addc(a.l[i], b.l[i], cy, &co);
        cy = co;
    }
    uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i] = __builtin_subc(t[i], C::P[i], bo, &co);
        bo = co;
    }
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = bo ? t[i] : s[i];
    return r;
}

template <class C>
ZK_DI Fp<C> sub(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    uint32_t t[N], bo = 0, co;
#pragma unroll
    for (int i = 0; i < N; i++) {
        t[i] = __builtin_subc(a.l[i], b.l[i], bo, &co);
        bo = co;
    }
    uint32_t mask = 0u - bo, cy = 0;
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) {
        r.l[i] = __builtin_addc(t[i], C::P[i] & mask, cy, &co);
        cy = co;
    }
    return r;
}

template <class C>
ZK_DI Fp<C> neg(const Fp<C>& a) {
    return sub(Fp<C>::zero(), a);
}

template <class C>
ZK_DI Fp<C> dbl(const Fp<C>& a) {
    return add(a, a);
}

// Montgomery product a*b*R^-1 mod p (fr.rs:438-464 / fq.rs:915-1016 compute the same value).
// Operands and result are passed as N-wide vector values so that the (non-inlined) call keeps
// them in VGPRs v0..v(2N-1); an aggregate argument would be passed through scratch memory.
template <class C>
ZK_DI typename C::vec mul_raw_inl(typename C::vec av, typename C::vec bv) {
    constexpr int N = C::N;
    struct { uint32_t l[C::N]; } a, b;
#pragma unroll
    for (int j = 0; j < N; j++) {
        a.l[j] = av[j];
        b.l[j] = bv[j];
    }
    uint32_t t[N + 1];
#pragma unroll
    for (int j = 0; j <= N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t pr[N];
        uint32_t cy, co;
        // t += a * b[i]   (t < 2p before, so the top word t[N] is zero on entry)
#pragma unroll
        for (int j = 0; j < N; j++) pr[j] = (uint64_t)a.l[j] * b.l[i] + t[j];
        t[0] = (uint32_t)pr[0];
        cy = 0;
#pragma unroll
        for (int j = 1; j < N; j++) {
            t[j] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), cy, &co);
            cy = co;
        }
        t[N] = (uint32_t)(pr[N - 1] >> 32) + cy;
        // t = (t + m*p) >> 32 with m = t[0] * (-p^-1)
        uint32_t m = t[0] * C::INV;
#pragma unroll
        for (int j = 0; j < N; j++) pr[j] = (uint64_t)m * C::P[j] + t[j];
        cy = 0;
#pragma unroll
        for (int j = 1; j < N; j++) {
            t[j - 1] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), cy, &co);
            cy = co;
        }
        t[N - 1] = __builtin_addc(t[N], (uint32_t)(pr[N - 1] >> 32), cy, &co);
    }
    uint32_t s[N], bo = 0, co;
#pragma unroll
    for (int j = 0; j < N; j++) {
        s[j] = __builtin_subc(t[j], C::P[j], bo, &co);
        bo = co;
    }
    typename C::vec r;
#pragma unroll
    for (int j = 0; j < N; j++) r[j] = bo ? t[j] : s[j];
    return r;
}

// acc (96 bits: 64-bit pair + overflow word) += a * b.  v_mad_u64_u32 accumulates in place and
// leaves its carry-out in VCC, v_addc_co_u32 folds it into the overflow word: 2 instructions per
// limb product and no register shuffling (hipcc's own code for `(u64)a * b + t` spends a v_mov
// per product on building the {t, 0} addend pair).
ZK_DI void mac96(uint64_t& acc, uint32_t& ovf, uint32_t a, uint32_t b) {
#ifndef ZK_EMU
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "v"(b)
        : "vcc");
#else
    unsigned __int128 t = (unsigned __int128)((uint64_t)a * b) + acc;
    acc = (uint64_t)t;
    ovf += (uint32_t)(t >> 64);
#endif
}
// same with a wave-uniform multiplier (a modulus limb) held in an SGPR
ZK_DI void mac96_k(uint64_t& acc, uint32_t& ovf, uint32_t a, uint32_t k) {
#ifndef ZK_EMU
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "s"(k)
        : "vcc");
#else
    mac96(acc, ovf, a, k);
#endif
}

// Montgomery product, finely integrated product scanning (FIPS): column k of a*b + m*p is summed
// into one 96-bit accumulator, m[k] is chosen to clear its low word, the accumulator shifts down
// one word.  2N^2 limb products, each one mac96.
template <class C>
ZK_DI typename C::vec mul_raw_fips(typename C::vec av, typename C::vec bv) {
    constexpr int N = C::N;
    uint32_t a[N], b[N], m[N], r[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        a[j] = av[j];
        b[j] = bv[j];
    }
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mac96(acc, ovf, a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mac96_k(acc, ovf, m[i], C::P[k - i]);
        m[k] = (uint32_t)acc * C::INV;
        mac96_k(acc, ovf, m[k], C::P[0]);
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac96(acc, ovf, a[i], b[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac96_k(acc, ovf, m[i], C::P[k - i]);
        r[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
    r[N - 1] = (uint32_t)acc;   // a*b + m*p < 2pR: the quotient fits N words
    uint32_t s[N], bo = 0, co;
#pragma unroll
    for (int j = 0; j < N; j++) {
        s[j] = __builtin_subc(r[j], C::P[j], bo, &co);
        bo = co;
    }
    typename C::vec o;
#pragma unroll
    for (int j = 0; j < N; j++) o[j] = bo ? r[j] : s[j];
    return o;
}

// The out-of-line instance the curve formulas call (see ZK_MUL_ATTR above).  On the GPU it is the
// hand-scheduled assembly of mul_asm.h (tools/gen_mul_asm.py): operands arrive in v[0:N-1] and
// v[N:2N-1] by the calling convention, the result leaves in v[0:N-1].  ZK_MUL_CXX selects the
// compiler-generated CIOS loop instead (A/B measurements, and the x86 emulation build).
template <class C>
ZK_MUL_ATTR typename C::vec mul_raw(typename C::vec av, typename C::vec bv);

#if defined(ZK_EMU) || defined(ZK_MUL_CXX)
template <>
ZK_MUL_ATTR u32x8 mul_raw<FrCfg>(u32x8 av, u32x8 bv) { return mul_raw_inl<FrCfg>(av, bv); }
template <>
ZK_MUL_ATTR u32x12 mul_raw<FqCfg>(u32x12 av, u32x12 bv) { return mul_raw_inl<FqCfg>(av, bv); }
#else
template <>
ZK_MUL_ATTR u32x8 mul_raw<FrCfg>(u32x8 av, u32x8 bv) {
    u32x8 r;
    // b's registers are reused for r - p: read-write operand
    asm(ZK_MUL_ASM_FR : "={v[0:7]}"(r), "+{v[8:15]}"(bv) : "{v[0:7]}"(av) : ZK_MUL_ASM_FR_CLOBBERS);
    return r;
}
template <>
ZK_MUL_ATTR u32x12 mul_raw<FqCfg>(u32x12 av, u32x12 bv) {
    u32x12 r;
    asm(ZK_MUL_ASM_FQ : "={v[0:11]}"(r), "+{v[12:23]}"(bv) : "{v[0:11]}"(av) : ZK_MUL_ASM_FQ_CLOBBERS);
    return r;
}
#endif

template <class C>
ZK_DI Fp<C> mul(const Fp<C>& a, const Fp<C>& b) {
    typename C::vec av, bv;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        av[j] = a.l[j];
        bv[j] = b.l[j];
    }
    typename C::vec rv = mul_raw<C>(av, bv);
    Fp<C> r;
#pragma unroll
    for (int j = 0; j < C::N; j++) r.l[j] = rv[j];
    return r;
}

template <class C>
ZK_DI Fp<C> sqr(const Fp<C>& a) {
    return mul(a, a);
}

template <class C>
ZK_DI Fp<C> to_mont(const Fp<C>& a) {
    return mul(a, Fp<C>::r2());
}

template <class C>
ZK_DI Fp<C> from_mont(const Fp<C>& a) {
    Fp<C> o = Fp<C>::zero();
    o.l[0] = 1;
    return mul(a, o);
}

// a^e for a (public) multi-limb exponent, MSB first; used only off the hot path (inversion).
template <class C, int EN>
ZK_DI Fp<C> pow_limbs(const Fp<C>& a, const uint32_t (&e)[EN]) {
    Fp<C> r = Fp<C>::one();
    bool started = false;
    for (int i = EN - 1; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            if (started) r = sqr(r);
            if ((e[i] >> b) & 1u) {
                r = started ? mul(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

typedef Fp<FrCfg> Fr;
typedef Fp<FqCfg> Fq32;   // Fq in saturated 32-bit limbs: host interchange format, debug entries

// ---------------------------------------------------------------------------------------------
// Fq for the MSM kernels: radix 2^28, 14 limbs, Montgomery radix 2^392, lazily reduced.
//
// The saturated 32-bit product above is bound by its carries: 288 v_mad_u64_u32 + 288 v_addc at
// 4 cycles each, whatever the schedule (three implementations measure the same 59 G products/s).
// With 28-bit limbs a whole column of the product (28 limb products < 2^56.01) fits one 64-bit
// accumulator, so there is no carry instruction anywhere: 392 v_mad_u64_u32 + 96 others per
// product (mul_asm.h FQ28), additions are 14 independent v_add + a 3-instruction-per-limb weak
// normalisation, and there is no carry chain for the gfx950 SGPR hazard to stall.
//
// Invariants of a stored value x:  limbs x_i <= 2^28 + 8 for i < 13 ("weakly normalised"; the top
// limb takes whatever is left), integer value < 64 p.  Products need |a| |b| < 2^11.3 p^2 (then
// a*b*2^-392 + m p / 2^392 < 2p); the curve formulas below carry the bound of every intermediate
// in their template arguments, and the x86 emulation build checks them at run time.
// mul outputs are EXACTLY normalised (limbs < 2^28) and < 2p but not reduced below p.
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

struct Fq28Consts {
    static constexpr uint32_t P[14] = ZK_FQ28_P;
    static constexpr uint32_t ONE[14] = ZK_FQ28_ONE;
    static constexpr uint32_t KIN[14] = ZK_FQ28_KIN;
    static constexpr uint32_t KOUT[14] = ZK_FQ28_KOUT;
    static constexpr uint32_t KINV[14] = ZK_FQ28_KINV;
    static constexpr uint32_t R2[14] = ZK_FQ28_R2;
    static constexpr uint32_t B[14] = ZK_FQ28_B;
    static constexpr uint32_t INV = ZK_FQ28_INV;
};
template <int M> struct Fq28Spread;
#define ZK_SPREAD_DEF(M) template <> struct Fq28Spread<M> { static constexpr uint32_t V[14] = ZK_FQ28_SPREAD_##M; };
ZK_SPREAD_DEF(2) ZK_SPREAD_DEF(3) ZK_SPREAD_DEF(4) ZK_SPREAD_DEF(5) ZK_SPREAD_DEF(6) ZK_SPREAD_DEF(7) ZK_SPREAD_DEF(8)
ZK_SPREAD_DEF(9) ZK_SPREAD_DEF(10) ZK_SPREAD_DEF(11) ZK_SPREAD_DEF(12) ZK_SPREAD_DEF(13) ZK_SPREAD_DEF(14)
ZK_SPREAD_DEF(15) ZK_SPREAD_DEF(16) ZK_SPREAD_DEF(17) ZK_SPREAD_DEF(18) ZK_SPREAD_DEF(19) ZK_SPREAD_DEF(20)
ZK_SPREAD_DEF(21) ZK_SPREAD_DEF(22) ZK_SPREAD_DEF(23) ZK_SPREAD_DEF(24) ZK_SPREAD_DEF(25) ZK_SPREAD_DEF(26)
ZK_SPREAD_DEF(27) ZK_SPREAD_DEF(28) ZK_SPREAD_DEF(29) ZK_SPREAD_DEF(30) ZK_SPREAD_DEF(31) ZK_SPREAD_DEF(32)
ZK_SPREAD_DEF(33) ZK_SPREAD_DEF(34) ZK_SPREAD_DEF(35) ZK_SPREAD_DEF(36) ZK_SPREAD_DEF(37) ZK_SPREAD_DEF(38)
ZK_SPREAD_DEF(39) ZK_SPREAD_DEF(40) ZK_SPREAD_DEF(41) ZK_SPREAD_DEF(42) ZK_SPREAD_DEF(43) ZK_SPREAD_DEF(44)
ZK_SPREAD_DEF(45) ZK_SPREAD_DEF(46) ZK_SPREAD_DEF(47) ZK_SPREAD_DEF(48) ZK_SPREAD_DEF(49) ZK_SPREAD_DEF(50)
ZK_SPREAD_DEF(51) ZK_SPREAD_DEF(52) ZK_SPREAD_DEF(53) ZK_SPREAD_DEF(54) ZK_SPREAD_DEF(55) ZK_SPREAD_DEF(56)
ZK_SPREAD_DEF(57) ZK_SPREAD_DEF(58) ZK_SPREAD_DEF(59) ZK_SPREAD_DEF(60) ZK_SPREAD_DEF(61) ZK_SPREAD_DEF(62)
ZK_SPREAD_DEF(63) ZK_SPREAD_DEF(64)
#undef ZK_SPREAD_DEF

constexpr uint32_t FQ28_MASK = (1u << 28) - 1;

// plain C++ product (the emulation build, and the checker of the assembly in tools/ubench)
ZK_DI u32x16 mul28_cxx(u32x16 av, u32x16 bv) {
    uint32_t a[14], b[14], m[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        a[j] = av[j];
        b[j] = bv[j];
    }
    u32x16 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
        for (int i = 0; i < 14; i++)
            if (k - i >= 0 && k - i < 14) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < 14; i++)
            if (k - i >= 0 && k - i < 14 && (k >= 14 || i < k)) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        if (k < 14) {
            m[k] = ((uint32_t)acc * Fq28Consts::INV) & FQ28_MASK;
            acc += (uint64_t)m[k] * Fq28Consts::P[0];
        } else {
            r[k - 14] = (uint32_t)acc & FQ28_MASK;
        }
        acc >>= 28;
    }
    r[13] = (uint32_t)acc;
    r[14] = 0;
    r[15] = 0;
    return r;
}

#if defined(ZK_EMU) || defined(ZK_MUL_CXX)
ZK_MUL_ATTR u32x16 mul28_raw(u32x16 av, u32x16 bv) { return mul28_cxx(av, bv); }
#else
ZK_MUL_ATTR u32x16 mul28_raw(u32x16 av, u32x16 bv) {
    u32x16 r;
#ifndef ZK_MUL28_DUAL   // (a two-accumulator schedule, FQ28D, measured slower: the routine is issue-bound, not latency-bound)
    asm(ZK_MUL_ASM_FQ28 : "={v[0:15]}"(r), "+{v[16:31]}"(bv) : "{v[0:15]}"(av) : ZK_MUL_ASM_FQ28_CLOBBERS);
#else
    asm(ZK_MUL_ASM_FQ28D : "={v[0:15]}"(r), "+{v[16:31]}"(bv) : "{v[0:15]}"(av) : ZK_MUL_ASM_FQ28D_CLOBBERS);
#endif
    return r;
}
#endif



#ifdef ZK_EMU
// run-time check of the magnitude bookkeeping (x86 emulation build only): value / p
static inline long double fq28_ratio(const uint32_t* l) {
    long double v = 0, p = 0;
    for (int i = 13; i >= 0; i--) {
        v = v * 268435456.0L + (long double)l[i];
        p = p * 268435456.0L + (long double)Fq28Consts::P[i];
    }
    return v / p;
}
#include <execinfo.h>
#define ZK_FQ28_CHECK(cond) do { if (!(cond)) { fprintf(stderr, "Fq28 bound violated: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
    void* bt_[32]; backtrace_symbols_fd(bt_, backtrace(bt_, 32), 2); abort(); } } while (0)
#else
#define ZK_FQ28_CHECK(cond) do { } while (0)
#endif

struct Fq28 {
    static constexpr int MO = 2;   // a product is < MO * p
    static constexpr int WB = 64;  // wr() is the identity for G1
    uint32_t l[14];

    ZK_DI static Fq28 zero() {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = 0;
        return r;
    }
    ZK_DI static Fq28 from_const(const uint32_t (&v)[14]) {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.l[i] = v[i];
        return r;
    }
    ZK_DI static Fq28 one() { return from_const(Fq28Consts::ONE); }
    // zero test of an EXACTLY normalised value < 2p (a product, or a constant): 0 or p
    ZK_DI bool is_zero_norm() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            o |= l[i];
            q |= l[i] ^ Fq28Consts::P[i];
        }
        return o == 0 || q == 0;
    }
};

// limbs <= 2^28 + 8 again after an addition / subtraction (inputs: any 32-bit limbs)
ZK_DI void fq28_wnorm(uint32_t (&t)[14]) {
    uint32_t c[13];
#pragma unroll
    for (int i = 0; i < 13; i++) c[i] = t[i] >> 28;
#pragma unroll
    for (int i = 1; i < 13; i++) t[i] = (t[i] & FQ28_MASK) + c[i - 1];
    t[0] &= FQ28_MASK;
    t[13] += c[12];
}

ZK_DI Fq28 add(const Fq28& a, const Fq28& b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    fq28_wnorm(r.l);
    ZK_FQ28_CHECK(fq28_ratio(r.l) < 64.0L);
    return r;
}
ZK_DI Fq28 dbl(const Fq28& a) { return add(a, a); }

// a - b + (B + 1) p  for b < B p  (limb-wise against the spread form of (B + 1) p: never negative)
template <int B>
ZK_DI Fq28 sub_b(const Fq28& a, const Fq28& b) {
    static_assert(B + 1 >= 2 && B + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(fq28_ratio(b.l) < (long double)B);
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + Fq28Spread<B + 1>::V[i] - b.l[i];
    fq28_wnorm(r.l);
    ZK_FQ28_CHECK(fq28_ratio(r.l) < 64.0L);
    return r;
}
template <int B>
ZK_DI Fq28 neg_b(const Fq28& a) {
    return sub_b<B>(Fq28::zero(), a);
}

// The same WITHOUT the weak normalisation (limbs < 2^30.4): allowed where the value is consumed exactly once as
// the FIRST operand of a product (mul: 14 x 2^58.4 + the reduction's 2^59.8 < 2^64) or as an operand of mul_sub2
// whose partner is normalised.  Saves the 41-instruction carry pass.
template <int B>
ZK_DI Fq28 sub_raw(const Fq28& a, const Fq28& b) {
    static_assert(B + 1 >= 2 && B + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(fq28_ratio(b.l) < (long double)B);
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + Fq28Spread<B + 1>::V[i] - b.l[i];
    return r;
}
template <int B>
ZK_DI Fq28 neg_raw(const Fq28& a) {
    return sub_raw<B>(Fq28::zero(), a);
}
// a - b - 2 c + (BB + 2 BC + 1) p in one pass with one normalisation (x3 = r^2 - ppp - 2 q of a point addition)
template <int BB, int BC>
ZK_DI Fq28 sub_sub2(const Fq28& a, const Fq28& b, const Fq28& c) {
    static_assert(BB + 2 * BC + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(fq28_ratio(b.l) < (long double)BB && fq28_ratio(c.l) < (long double)BC);
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + Fq28Spread<BB + 2 * BC + 1>::V[i] - b.l[i] - 2u * c.l[i];
    fq28_wnorm(r.l);
    ZK_FQ28_CHECK(fq28_ratio(r.l) < 64.0L);
    return r;
}

ZK_DI Fq28 mul(const Fq28& a, const Fq28& b) {
    ZK_FQ28_CHECK(fq28_ratio(a.l) * fq28_ratio(b.l) < 2500.0L);
    u32x16 av, bv;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        av[j] = a.l[j];
        bv[j] = b.l[j];
    }
    av[14] = av[15] = bv[14] = bv[15] = 0;
    u32x16 rv = mul28_raw(av, bv);
    Fq28 r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = rv[j];
    return r;
}
// dedicated square (mul_asm.h FQ28SQR: 410 instead of 488 instructions)
#if defined(ZK_EMU) || defined(ZK_MUL_CXX)
ZK_MUL_ATTR u32x16 sqr28_raw(u32x16 av) { return mul28_cxx(av, av); }
#else
ZK_MUL_ATTR u32x16 sqr28_raw(u32x16 av) {
    u32x16 r;
    asm(ZK_MUL_ASM_FQ28SQR : "={v[0:15]}"(r) : "{v[0:15]}"(av) : ZK_MUL_ASM_FQ28SQR_CLOBBERS);
    return r;
}
#endif
ZK_DI Fq28 sqr(const Fq28& a) {
    ZK_FQ28_CHECK(fq28_ratio(a.l) * fq28_ratio(a.l) < 2500.0L);
    u32x16 av;
#pragma unroll
    for (int j = 0; j < 14; j++) av[j] = a.l[j];
    av[14] = av[15] = 0;
    u32x16 rv = sqr28_raw(av);
    Fq28 r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = rv[j];
    return r;
}

// ---------------------------------------------------------------------------------------------
// Fused routines with a custom register contract (mul_asm.h FQ28MAC2 / FQ2MUL28).  Four 14-limb operands
// do not fit the 32 VGPRs the calling convention passes in registers, so these are `naked` functions
// reached with s_swappc_b64 from an inline-assembly statement that names the operand registers and the
// clobbers itself: v[0:15], v[16:31], v[32:47], v[48:63] in, v[0:15] (and v[16:31]) out.
//   mac2:    c = (x0 y0 + x1 y1) 2^-392        one Montgomery reduction for a sum of two products
//   fq2mul:  c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0   (x 2^-392): 4 limb-product groups, 2 reductions
//            (two accumulator chains; four, folded per column with a 64-bit add, measured slower even for a wave
//            alone on its SIMD: G2 accumulation 90.2 -> 93.1 ms - the routine is bound by issue, not by latency)
//   mul2:    c0 = a0 b0, c1 = a1 b1                    (x 2^-392): two independent products, interleaved
// ---------------------------------------------------------------------------------------------
// plain C++ of the same column schedule (the emulation build, ZK_MUL_CXX): bit-identical results
ZK_DI u32x16 mac2_cxx(u32x16 x0, u32x16 y0, u32x16 x1, u32x16 y1) {
    uint32_t m[14];
    u32x16 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
        for (int i = 0; i < 14; i++)
            if (k - i >= 0 && k - i < 14) {
                acc += (uint64_t)x0[i] * y0[k - i];
                acc += (uint64_t)x1[i] * y1[k - i];
            }
#pragma unroll
        for (int i = 0; i < 14; i++)
            if (k - i >= 0 && k - i < 14 && (k >= 14 || i < k)) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        if (k < 14) {
            m[k] = ((uint32_t)acc * Fq28Consts::INV) & FQ28_MASK;
            acc += (uint64_t)m[k] * Fq28Consts::P[0];
        } else {
            r[k - 14] = (uint32_t)acc & FQ28_MASK;
        }
        acc >>= 28;
    }
    r[13] = (uint32_t)acc;
    r[14] = 0;
    r[15] = 0;
    return r;
}
constexpr int FQ2_SPREAD_K = 16;   // tools/gen_mul_asm.py SPREAD_K: the fused Fq2 product negates a1 against 16 p

#if defined(ZK_EMU) || defined(ZK_MUL_CXX)
ZK_DI void mac2_raw(u32x16& x0, const u32x16& y0, const u32x16& x1, const u32x16& y1) { x0 = mac2_cxx(x0, y0, x1, y1); }
ZK_DI void fq2mul_raw(u32x16& a0, u32x16& a1, const u32x16& b0, const u32x16& b1) {
    u32x16 n1;
#pragma unroll
    for (int i = 0; i < 14; i++) n1[i] = Fq28Spread<FQ2_SPREAD_K>::V[i] - a1[i];
    n1[14] = n1[15] = 0;
    const u32x16 c0 = mac2_cxx(a0, b0, n1, b1), c1 = mac2_cxx(a0, b1, a1, b0);
    a0 = c0;
    a1 = c1;
}
ZK_DI void mul2_raw(u32x16& a0, u32x16& a1, const u32x16& b0, const u32x16& b1) {
    a0 = mul28_cxx(a0, b0);
    a1 = mul28_cxx(a1, b1);
}
#else
extern "C" __device__ __attribute__((naked, noinline, used)) void zk_fq28_mul2() {
    asm volatile(ZK_MUL_ASM_FQ28MUL2 "s_setpc_b64 s[30:31]");
}
extern "C" __device__ __attribute__((naked, noinline, used)) void zk_fq28_mac2() {
    asm volatile(ZK_MUL_ASM_FQ28MAC2 "s_setpc_b64 s[30:31]");
}
extern "C" __device__ __attribute__((naked, noinline, used)) void zk_fq2mul28() {
    asm volatile(ZK_MUL_ASM_FQ2MUL28 "s_setpc_b64 s[30:31]");
}
#define ZK_ASM_CALL(fn)                                   \
    "s_getpc_b64 s[18:19]\n\t"                            \
    "s_add_u32 s18, s18, " fn "@rel32@lo+4\n\t"           \
    "s_addc_u32 s19, s19, " fn "@rel32@hi+12\n\t"         \
    "s_swappc_b64 s[30:31], s[18:19]"
ZK_DI void mac2_raw(u32x16& x0, const u32x16& y0, const u32x16& x1, const u32x16& y1) {
    asm(ZK_ASM_CALL("zk_fq28_mac2")
        : "+{v[0:15]}"(x0)
        : "{v[16:31]}"(y0), "{v[32:47]}"(x1), "{v[48:63]}"(y1)
        : ZK_MUL_ASM_FQ28MAC2_CLOBBERS, "s18", "s19", "s30", "s31");
}
ZK_DI void fq2mul_raw(u32x16& a0, u32x16& a1, const u32x16& b0, const u32x16& b1) {
    asm(ZK_ASM_CALL("zk_fq2mul28")
        : "+{v[0:15]}"(a0), "+{v[16:31]}"(a1)
        : "{v[32:47]}"(b0), "{v[48:63]}"(b1)
        : ZK_MUL_ASM_FQ2MUL28_CLOBBERS, "s18", "s19", "s30", "s31");
}
// two independent products a0 b0, a1 b1 on two interleaved accumulator chains (mul_asm.h FQ28MUL2)
ZK_DI void mul2_raw(u32x16& a0, u32x16& a1, const u32x16& b0, const u32x16& b1) {
    asm(ZK_ASM_CALL("zk_fq28_mul2")
        : "+{v[0:15]}"(a0), "+{v[16:31]}"(a1)
        : "{v[32:47]}"(b0), "{v[48:63]}"(b1)
        : ZK_MUL_ASM_FQ28MUL2_CLOBBERS, "s18", "s19", "s30", "s31");
}
#endif

ZK_DI u32x16 fq28_vec(const Fq28& a) {
    u32x16 v;
#pragma unroll
    for (int j = 0; j < 14; j++) v[j] = a.l[j];
    v[14] = v[15] = 0;
    return v;
}
ZK_DI Fq28 fq28_unvec(const u32x16& v) {
    Fq28 r;
#pragma unroll
    for (int j = 0; j < 14; j++) r.l[j] = v[j];
    return r;
}
// x0 y0 - x1 y1 for x1 < B p, with ONE reduction: the subtrahend enters as the limb-wise (B + 1) p - x1, left
// un-normalised (limbs < 2^30, which the routine tolerates in that operand).  Result exactly normalised, < 2p
// as long as |x0||y0| + (B + 1)|y1| < 2^11.
template <int B>
ZK_DI Fq28 mul_sub2(const Fq28& x0, const Fq28& y0, const Fq28& x1, const Fq28& y1) {
    static_assert(B + 1 >= 2 && B + 1 <= 64, "no spread constant for this bound");
    ZK_FQ28_CHECK(fq28_ratio(x1.l) < (long double)B);
    ZK_FQ28_CHECK(fq28_ratio(x0.l) * fq28_ratio(y0.l) + (long double)(B + 1) * fq28_ratio(y1.l) < 2000.0L);
    u32x16 a = fq28_vec(x0), n;
#pragma unroll
    for (int j = 0; j < 14; j++) n[j] = Fq28Spread<B + 1>::V[j] - x1.l[j];
    n[14] = n[15] = 0;
    mac2_raw(a, fq28_vec(y0), n, fq28_vec(y1));
    return fq28_unvec(a);
}

// the unique representative in [0, p), exactly normalised (rare paths: export, equality)
ZK_DI Fq28 canon(const Fq28& a) {
    Fq28 t = mul(a, Fq28::one());   // < 2p, exact limbs
    int32_t d[14], bo = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int32_t v = (int32_t)t.l[i] - (int32_t)Fq28Consts::P[i] - bo;
        bo = v < 0 ? 1 : 0;
        d[i] = i < 13 ? (v & (int32_t)FQ28_MASK) : v;
    }
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = bo ? t.l[i] : (uint32_t)d[i];
    return r;
}
ZK_DI bool is_zero_full(const Fq28& a) { return mul(a, Fq28::one()).is_zero_norm(); }

// host interchange: 12 x u32 canonical Montgomery (radix 2^384) limbs  <->  Fq28
// 12 x 32-bit words (an integer < 2^384) -> 14 limbs of 28 bits, as they are
ZK_DI Fq28 fq28_unpack(const uint32_t* h) {
    Fq28 t;
    uint32_t w[13];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = h[i];
    w[12] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, q = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[q] | ((uint64_t)w[q + 1] << 32);
        t.l[i] = (uint32_t)(two >> sh) & FQ28_MASK;
    }
    return t;
}
ZK_DI Fq28 fq28_import(const uint32_t* h) { return mul(fq28_unpack(h), Fq28::from_const(Fq28Consts::KIN)); }
// t = a * KOUT (exact limbs, < 2p) -> the canonical 12 words: subtract p once if needed, repack 14 x 28 -> 12 x 32 bits
ZK_DI void fq28_export_tail(const Fq28& t, uint32_t* h) {
    int32_t d[14], bo = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int32_t v = (int32_t)t.l[i] - (int32_t)Fq28Consts::P[i] - bo;
        bo = v < 0 ? 1 : 0;
        d[i] = i < 13 ? (v & (int32_t)FQ28_MASK) : v;
    }
    uint32_t c[14];
#pragma unroll
    for (int i = 0; i < 14; i++) c[i] = bo ? t.l[i] : (uint32_t)d[i];
#pragma unroll
    for (int q = 0; q < 12; q++) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int lo = 28 * i - 32 * q;   // position of limb i inside word q
            if (lo > -28 && lo < 32) v |= lo >= 0 ? (c[i] << lo) : (c[i] >> (-lo));
        }
        h[q] = v;
    }
}
ZK_DI void fq28_export(const Fq28& a, uint32_t* h) { fq28_export_tail(mul(a, Fq28::from_const(Fq28Consts::KOUT)), h); }

// Weak reduction: any stored value (< 64 p) -> the same residue, EXACTLY normalised, < 3 p.
// q = floor(top limb / (p_top + 1)) never exceeds floor(x / p) and misses it by at most 2
// (x < 64 p, p_top ~ 2^16.7); x - q p is then formed with a signed borrow chain (no SGPR carries).
ZK_DI Fq28 fq28_wred(const Fq28& x) {
    constexpr uint32_t PT = Fq28Consts::P[13] + 1;
    // the float quotient is rounded towards zero and scaled down by 2^-18 so that it never overshoots
    uint32_t q = (uint32_t)((float)x.l[13] * ((1.0f / (float)PT) * (1.0f - 1.0f / 262144.0f)));
    uint32_t qp[14];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        acc += (uint64_t)q * Fq28Consts::P[i];
        qp[i] = i < 13 ? ((uint32_t)acc & FQ28_MASK) : (uint32_t)acc;
        acc >>= 28;
    }
    Fq28 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int32_t d = (int32_t)x.l[i] - (int32_t)qp[i] + c;
        if (i < 13) {
            c = d >> 28;   // arithmetic: -1, 0 or +1 (weakly normalised input limbs exceed 2^28 by at most 8)
            r.l[i] = (uint32_t)d & FQ28_MASK;
        } else {
            r.l[i] = (uint32_t)d;
        }
    }
    ZK_FQ28_CHECK((int32_t)r.l[13] >= 0 && fq28_ratio(r.l) < 3.0L);
    return r;
}
// zero test of any stored value: after the weak reduction the candidates are 0, p and 2p
ZK_DI bool fq28_is_zero_lazy(const Fq28& x) {
    Fq28 t = fq28_wred(x);
    uint32_t o0 = 0, o1 = 0, o2 = 0;
    constexpr uint32_t P2[14] = ZK_FQ28_2P;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        o0 |= t.l[i];
        o1 |= t.l[i] ^ Fq28Consts::P[i];
        o2 |= t.l[i] ^ P2[i];
    }
    return o0 == 0 || o1 == 0 || o2 == 0;
}
// identity hook: a G1 intermediate never needs a weak reduction (see the bounds in dev_curve.h)
ZK_DI Fq28 wr(const Fq28& a) { return a; }

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1) over the radix-2^28 representation: the G2 field of the MSM kernels.  fq2.rs:90-182.
// The product is the fused routine above (4 limb-product groups, 2 reductions): both components of a
// product are exactly normalised and < 2p, so an Fq2x obeys the same magnitude bookkeeping as a G1
// coordinate (MO = 2, no weak reductions anywhere: the curve formulas' largest operands are 13 p against
// the 15 p the routine's internal negation allows).  A square is two base-field products,
// (a0 + a1)(a0 - a1) and 2 a0 a1.
// (Round 1 ran G2 on the saturated 12 x 32-bit Fq2 below: Karatsuba over three reduced products, 659
// instructions each plus carry-chain additions, with scratch spills - 100 ms per 1024-proof launch.)
// ---------------------------------------------------------------------------------------------
struct Fq2x {
    static constexpr int MO = 2;
    static constexpr int WB = 64;   // wr() is the identity
    Fq28 c0, c1;
    ZK_DI static Fq2x zero() { return Fq2x{Fq28::zero(), Fq28::zero()}; }
    ZK_DI static Fq2x one() { return Fq2x{Fq28::one(), Fq28::zero()}; }
    // zero test of an EXACTLY normalised value (a product, a constant, a table entry)
    ZK_DI bool is_zero_norm() const { return c0.is_zero_norm() && c1.is_zero_norm(); }
};
ZK_DI Fq2x add(const Fq2x& a, const Fq2x& b) { return Fq2x{add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_DI Fq2x dbl(const Fq2x& a) { return add(a, a); }
template <int B>
ZK_DI Fq2x sub_b(const Fq2x& a, const Fq2x& b) { return Fq2x{sub_b<B>(a.c0, b.c0), sub_b<B>(a.c1, b.c1)}; }
template <int B>
ZK_DI Fq2x neg_b(const Fq2x& a) { return Fq2x{neg_b<B>(a.c0), neg_b<B>(a.c1)}; }
template <int BB, int BC>
ZK_DI Fq2x sub_sub2(const Fq2x& a, const Fq2x& b, const Fq2x& c) {
    return Fq2x{sub_sub2<BB, BC>(a.c0, b.c0, c.c0), sub_sub2<BB, BC>(a.c1, b.c1, c.c1)};
}
ZK_DI Fq2x wr(const Fq2x& a) { return a; }
ZK_DI bool is_zero_full(const Fq2x& a) { return is_zero_full(a.c0) && is_zero_full(a.c1); }
ZK_DI Fq2x mul(const Fq2x& a, const Fq2x& b) {
    ZK_FQ28_CHECK(fq28_ratio(a.c1.l) < (long double)(FQ2_SPREAD_K - 1));
    ZK_FQ28_CHECK(fq28_ratio(a.c0.l) * fq28_ratio(b.c0.l) + (long double)FQ2_SPREAD_K * fq28_ratio(b.c1.l) < 2000.0L);
    ZK_FQ28_CHECK(fq28_ratio(a.c0.l) * fq28_ratio(b.c1.l) + fq28_ratio(a.c1.l) * fq28_ratio(b.c0.l) < 2000.0L);
    u32x16 a0 = fq28_vec(a.c0), a1 = fq28_vec(a.c1);
    fq2mul_raw(a0, a1, fq28_vec(b.c0), fq28_vec(b.c1));
    return Fq2x{fq28_unvec(a0), fq28_unvec(a1)};
}
template <int A>   // A = bound of the operand's components (for the difference a0 - a1)
ZK_DI Fq2x sqr_b(const Fq2x& a) {
    static_assert(A <= 30, "operand of an Fq2 square out of range");
    // (a0 + a1)(a0 - a1) and (2 a0) a1 in ONE routine, their multiply-adds alternating: two dependency chains for
    // the lone wave per SIMD the G2 kernels run with
    const Fq28 s = add(a.c0, a.c1), d = sub_b<A>(a.c0, a.c1), t = dbl(a.c0);
    ZK_FQ28_CHECK(fq28_ratio(s.l) * fq28_ratio(d.l) < 2500.0L);
    ZK_FQ28_CHECK(fq28_ratio(t.l) * fq28_ratio(a.c1.l) < 2500.0L);
    u32x16 x0 = fq28_vec(s), x1 = fq28_vec(t);
    mul2_raw(x0, x1, fq28_vec(d), fq28_vec(a.c1));
    return Fq2x{fq28_unvec(x0), fq28_unvec(x1)};
}
// square of a product / table entry / imported value / negated product (components < 4 p)
ZK_DI Fq2x sqr(const Fq2x& a) { return sqr_b<4>(a); }

// The same interface on the saturated representation (G2 still runs on it): bounds are ignored,
// every value is fully reduced.
template <int B, class C>
ZK_DI Fp<C> sub_b(const Fp<C>& a, const Fp<C>& b) { return sub(a, b); }
template <int B, class C>
ZK_DI Fp<C> neg_b(const Fp<C>& a) { return neg(a); }
template <class C>
ZK_DI bool is_zero_full(const Fp<C>& a) { return a.is_zero(); }
template <class C>
ZK_DI Fp<C> wr(const Fp<C>& a) { return a; }
template <int A, class C>
ZK_DI Fp<C> sqr_b(const Fp<C>& a) { return sqr(a); }
template <int A>
ZK_DI Fq28 sqr_b(const Fq28& a) { return sqr(a); }

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1)  (fq2.rs:90-182)
// ---------------------------------------------------------------------------------------------
typedef Fq28 Fq;   // G1 base field as the MSM kernels see it
struct Fq2 {
    typedef Fq32 Fq;   // the saturated representation (kept for A/B measurements and the debug entries)
    static constexpr int MO = 1;
    static constexpr int WB = 64;
    Fq c0, c1;
    ZK_DI bool is_zero_norm() const { return is_zero(); }
    ZK_DI static Fq2 zero() { return Fq2{Fq32::zero(), Fq32::zero()}; }
    ZK_DI static Fq2 one() { return Fq2{Fq32::one(), Fq32::zero()}; }
    ZK_DI bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_DI bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
    ZK_DI bool operator!=(const Fq2& b) const { return !(*this == b); }
};
template <int B>
ZK_DI Fq2 sub_b(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
template <int B>
ZK_DI Fq2 neg_b(const Fq2& a) { return Fq2{neg(a.c0), neg(a.c1)}; }
ZK_DI bool is_zero_full(const Fq2& a) { return a.is_zero(); }
ZK_DI Fq2 wr(const Fq2& a) { return a; }
ZK_DI Fq2 add(const Fq2& a, const Fq2& b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_DI Fq2 sub(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZK_DI Fq2 neg(const Fq2& a) { return Fq2{neg(a.c0), neg(a.c1)}; }
ZK_DI Fq2 dbl(const Fq2& a) { return Fq2{dbl(a.c0), dbl(a.c1)}; }
ZK_DI Fq2 mul(const Fq2& a, const Fq2& b) {
    // Karatsuba, fq2.rs:133-158: 3 base-field products
    Fq32 aa = mul(a.c0, b.c0);
    Fq32 bb = mul(a.c1, b.c1);
    Fq32 o = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fq2{sub(aa, bb), sub(sub(o, aa), bb)};
}
ZK_DI Fq2 sqr(const Fq2& a);
template <int A>
ZK_DI Fq2 sqr_b(const Fq2& a) { return sqr(a); }
ZK_DI Fq2 sqr(const Fq2& a) {
    // complex squaring, fq2.rs:109-131: 2 base-field products
    Fq32 ab = mul(a.c0, a.c1);
    Fq32 s = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
    return Fq2{s, dbl(ab)};
}

}  // namespace zkdev
