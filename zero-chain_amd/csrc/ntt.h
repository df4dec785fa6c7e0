// Radix-2 NTT over the BLS12-381 scalar field Fr for gfx950, LDS-tiled.
//
// Replaces bellman 0.1.0's EvaluationDomain (domain.rs: fft / ifft / coset_fft / icoset_fft /
// mul_assign / sub_assign / divide_by_z_on_coset) as used by create_proof for the H query
// (call site: /root/reference/core/proofs/src/confidential.rs:149).  Field constants follow
// core/pairing/src/bls12_381/fr.rs:38-55 (GENERATOR = 7, S = 32, ROOT_OF_UNITY).
//
// bellman bit-reverses and then runs an in-place DIT.  Here a forward/inverse *pair* never
// permutes: a DIF pass chain maps natural order -> bit-reversed order, a DIT chain maps
// bit-reversed -> natural, and everything between them (coset scaling, 1/m, pointwise ops,
// Montgomery conversion) is fused into table multiplications at pass boundaries.
//
// One pass = up to NTT_MAX_G consecutive butterfly stages done inside LDS.  A workgroup owns a
// tile of CW "columns" x 2^g rows; element (row m, column col) lives at global index
//      hi * 2^(s+g) + m * 2^s + lo,     col = hi * 2^s + lo,
// so for s >= log2(CW) every row of the tile is one contiguous CW*32-byte segment (coalesced),
// and for s = 0 the whole tile is contiguous.  Twiddles are read from a table in HBM.
#pragma once
#include "dev_field.h"

namespace zkdev {

constexpr int NTT_MAX_G = 8;        // stages per pass
constexpr int NTT_TILE_LOG = 10;   // 2^10 elements = 32 KiB of LDS per workgroup: 4 workgroups = 4 waves per SIMD on a CU
constexpr int NTT_THREADS = 256;
// Large transforms (k >= NTT_BIG_LOG): 2^12-element tiles (128 KiB of LDS, one workgroup of 1024 threads = 4 waves
// per SIMD on a CU) hold 10 stages, so 2^20 takes 2 passes instead of 3 (0.327 -> 0.315 ms per pair).
// (ZKAMD_NTT_TILES = mid | big | small selects the tile form of these sizes: zkamd.cpp NttPlan::tile_form; round 6 made the
// 2^11-element "mid" tiles - two workgroups per CU - the default.)  Also measured and dropped: per-stage twiddle
// tables laid out so that lanes of consecutive columns read consecutive entries (twice the table memory; 0.327 ms
// and 21.1 ms per chunk, i.e. no change: the strided twiddle gathers are not what the kernel waits for).
constexpr int NTT_BIG_LOG = 17, NTT_BIG_MAX_G = 10, NTT_BIG_TILE_LOG = 12, NTT_BIG_THREADS = 1024;

struct NttPass {
    uint32_t log_n;    // transform size
    uint32_t t0;       // first global stage of this pass
    uint32_t g;        // stages in this pass
    uint32_t log_cw;   // log2(columns per tile)
    uint32_t dif;      // 1: decimation in frequency (natural -> bit-reversed), 0: DIT
    uint32_t stride;   // elements between consecutive polynomials of the batch
    uint32_t src_stride;  // when `src` is given: elements between polynomials in src ...
    uint32_t src_valid;   // ... and how many leading elements exist (the rest read as zero)
};

ZK_DI Fr ld_fr(const uint32_t* p) {
    Fr r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
ZK_DI void st_fr(uint32_t* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// x >= r ?  (the reference cannot represent such an Fr: FrRepr -> Fr fails, fr.rs:276-289; the same
// holds for raw Montgomery limbs).  Kernels that take scalars from the caller raise a flag.
ZK_DI bool fr_geq_r(const Fr& x) {
    uint32_t bo = 0, co;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        (void)__builtin_subc(x.l[i], FrCfg::P[i], bo, &co);
        bo = co;
    }
    return bo == 0;
}
constexpr uint32_t ZK_BAD_SCALAR = 1u, ZK_BAD_ONE = 2u;
ZK_DI void raise_flag(uint32_t* bad, uint32_t bit) {
#ifdef ZK_EMU
    __atomic_fetch_or(bad, bit, __ATOMIC_RELAXED);
#else
    atomicOr(bad, bit);
#endif
}

// LDS tile layout: two planes of 16 bytes per element (limbs 0..3 | limbs 4..7), so that a wave's
// ds_read_b128 of consecutive elements covers every bank once (an array of 32-byte elements read as
// two b128 halves leaves every other quad of banks idle: 2-way conflicts).
ZK_DI Fr ld_tile(const uint32_t* tile, uint32_t tile_elems, uint32_t e) {
    Fr r;
    const uint4 a = *reinterpret_cast<const uint4*>(tile + (size_t)e * 4);
    const uint4 b = *reinterpret_cast<const uint4*>(tile + ((size_t)tile_elems + e) * 4);
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
ZK_DI void st_tile(uint32_t* tile, uint32_t tile_elems, uint32_t e, const Fr& v) {
    *reinterpret_cast<uint4*>(tile + (size_t)e * 4) = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    *reinterpret_cast<uint4*>(tile + ((size_t)tile_elems + e) * 4) = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// One pass over a batch of polynomials (blockIdx.y = polynomial).  `tw` holds w^e (Montgomery)
// for e in [0, n/2).  `pre` / `post` (optional, n entries each) are multiplied into every
// element at load / store, indexed by the element's global position.  `src` (optional) replaces
// `data` as the load source for the first pass of a chain; `minus` (optional, same element order as the store, one
// array of `minus_stride` elements per polynomial) is subtracted after `post`.
//
// Occupancy is what this kernel is sensitive to: the Montgomery product is one long dependent
// multiply-add chain per lane, so a SIMD needs ~4 waves to keep issuing.  Measured on MI355X (7 x 1024
// transforms of 2^15, serial): 2^11-element tiles at 2 waves/SIMD 24.8 ms, 2^10-element tiles at 4
// waves/SIMD 21.3 ms.  Taking two stages per LDS round trip (radix-4 groups in registers: half the LDS
// traffic and barriers, the same products, since w^(n/4) is no cheaper than any other twiddle in a prime
// field) measured 22.1 ms at the same occupancy and 26.4 ms at 2 waves/SIMD: rejected.
static __global__ void __launch_bounds__(NTT_BIG_THREADS, 4)
k_ntt_pass(uint32_t* data, const uint32_t* __restrict__ src, const uint32_t* __restrict__ tw,
           const uint32_t* __restrict__ pre, const uint32_t* __restrict__ post, NttPass ps, uint32_t* bad = nullptr,
           const uint32_t* __restrict__ minus = nullptr, uint32_t minus_stride = 0) {
    ZK_DYN_SHARED(uint32_t, tile);   // 2 planes x [2^g][CW][4]
    const uint32_t k = ps.log_n, g = ps.g, lcw = ps.log_cw;
    const uint32_t rows = 1u << g, cw = 1u << lcw;
    // s = log2 of the smallest butterfly half-distance (in elements) handled by this pass
    const uint32_t s = ps.dif ? (k - ps.t0 - g) : ps.t0;
    uint32_t* base = data + (size_t)blockIdx.y * ps.stride * 8;
    const uint32_t col0 = blockIdx.x << lcw;
    const uint32_t tid = threadIdx.x;
    const uint32_t tile_elems = rows << lcw;
    const uint32_t smask = (1u << s) - 1;

    // ---- load tile (row-major over m, columns fastest => coalesced segments)
    for (uint32_t e = tid; e < tile_elems; e += blockDim.x) {
        uint32_t c = e & (cw - 1), m = e >> lcw;
        uint32_t col = col0 + c;
        uint32_t hi = col >> s, lo = col & smask;
        uint32_t idx = (hi << (s + g)) | (m << s) | lo;
        Fr v;
        if (src) {
            // first pass of a chain: read the caller's (unpadded) array, zero-extend to n
            v = idx < ps.src_valid ? ld_fr(src + ((size_t)blockIdx.y * ps.src_stride + idx) * 8) : Fr::zero();
            if (bad && fr_geq_r(v)) raise_flag(bad, ZK_BAD_SCALAR);
        } else {
            v = ld_fr(base + (size_t)idx * 8);
        }
        if (pre) v = mul(v, ld_fr(pre + (size_t)idx * 8));
        st_tile(tile, tile_elems, e, v);
    }
    __syncthreads();

    for (uint32_t j = 0; j < g; j++) {
        const uint32_t nbf = tile_elems >> 1;
        // position (within m) of the bit that separates the two butterfly inputs
        const uint32_t pos = ps.dif ? (g - 1 - j) : j;
        const uint32_t half = 1u << pos;
        // global stage index and the shift that turns (i mod d) into a twiddle exponent
        const uint32_t tsh = ps.dif ? (ps.t0 + j) : (k - 1 - ps.t0 - j);
        for (uint32_t b = tid; b < nbf; b += blockDim.x) {
            uint32_t c = b & (cw - 1), mm = b >> lcw;
            uint32_t m = ((mm >> pos) << (pos + 1)) | (mm & (half - 1));
            uint32_t lo = (col0 + c) & smask;
            // i mod d, d = half * 2^s
            uint32_t imod = ((m & (half - 1)) << s) | lo;
            uint32_t e = imod << tsh;
            const uint32_t ax = (m << lcw) | c, ay = ((m + half) << lcw) | c;
            Fr x = ld_tile(tile, tile_elems, ax), y = ld_tile(tile, tile_elems, ay);
            if (ps.dif) {
                Fr u = add(x, y);
                Fr v = sub(x, y);
                if (e) v = mul(v, ld_fr(tw + (size_t)e * 8));
                st_tile(tile, tile_elems, ax, u);
                st_tile(tile, tile_elems, ay, v);
            } else {
                if (e) y = mul(y, ld_fr(tw + (size_t)e * 8));
                st_tile(tile, tile_elems, ax, add(x, y));
                st_tile(tile, tile_elems, ay, sub(x, y));
            }
        }
        __syncthreads();
    }

    for (uint32_t e = tid; e < tile_elems; e += blockDim.x) {
        uint32_t c = e & (cw - 1), m = e >> lcw;
        uint32_t col = col0 + c;
        uint32_t hi = col >> s, lo = col & smask;
        uint32_t idx = (hi << (s + g)) | (m << s) | lo;
        Fr v = ld_tile(tile, tile_elems, e);
        if (post) v = mul(v, ld_fr(post + (size_t)idx * 8));
        // last pass of the H pipeline: the coefficients of c are subtracted on the way out (see prove_chunk)
        if (minus) v = sub(v, ld_fr(minus + ((size_t)blockIdx.y * minus_stride + idx) * 8));
        st_fr(base + (size_t)idx * 8, v);
    }
}

// h' = a * b * zinv, element-wise over `count` = batch * m elements (a, b Montgomery, on the coset), written to
// out[proof * out_stride + e]: the last inverse transform then runs in place inside the per-proof scalar vector of
// the merged C multiexp.  With `c` given: (a * b - c) * zinv, bellman's literal order
// (a.mul_assign(b); a.sub_assign(c); a.divide_by_z_on_coset()).
// The prover passes c = nullptr and never evaluates c on the coset: with ab = lo + x^m hi (both halves of degree < m),
// the coset interpolation of ab is lo + g^m hi and c itself is a polynomial of degree < m, so
//     icoset_fft((ab - c) / Z on the coset) = (lo + g^m hi - c) / (g^m - 1) = (icoset_fft(ab on the coset) - c) / (g^m - 1)
// coefficient by coefficient - for ANY a, b, c, satisfied constraints or not: the same vector as bellman's, one
// transform of size m less (6 instead of 7).  The coefficients of c come out of its inverse transform scaled by
// 1 / (m (g^m - 1)) (NttPlan sc_*) and are subtracted in the store of the last pass (k_ntt_pass `minus`).
static __global__ void __launch_bounds__(256)
k_h_pointwise(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
              const uint32_t* __restrict__ zinv, uint32_t* out, uint32_t m, uint32_t out_stride, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr z = ld_fr(zinv);
    Fr x = mul(ld_fr(a + i * 8), ld_fr(b + i * 8));
    if (c) x = sub(x, ld_fr(c + i * 8));
    size_t proof = i / m, e = i % m;
    st_fr(out + (proof * out_stride + e) * 8, mul(x, z));
}

// out[i] = in[i] * tab[i]  (optional table) ; used by the stand-alone zk_ntt_fr entry
static __global__ void __launch_bounds__(256)
k_fr_scale(uint32_t* data, const uint32_t* __restrict__ tab, size_t n, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    st_fr(data + i * 8, mul(ld_fr(data + i * 8), ld_fr(tab + (i % n) * 8)));
}

// out[bitrev(i)] = in[i] (out-of-place)
static __global__ void __launch_bounds__(256)
k_fr_bitrev(uint32_t* out, const uint32_t* __restrict__ in, uint32_t log_n, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t n_mask = (1u << log_n) - 1;
    uint32_t lo = (uint32_t)i & n_mask;
    size_t poly = i >> log_n;
    uint32_t r = log_n ? (__builtin_bitreverse32(lo) >> (32 - log_n)) : 0;
    st_fr(out + ((poly << log_n) + r) * 8, ld_fr(in + i * 8));
}

ZK_DI Fr fr_pow_u32(Fr base, uint32_t e) {
    Fr r = Fr::one();
    while (e) {
        if (e & 1u) r = mul(r, base);
        base = sqr(base);
        e >>= 1;
    }
    return r;
}

// Table generation.  mode 0: out[i] = base^i                     (twiddles, i < count)
//                    mode 1: out[pos] = base^bitrev(pos) * scale   (coset tables, bit-reversed)
//                    mode 2: out[i] = base^i * scale
// `base`, `scale` are Montgomery; if raw_out the Montgomery factor is stripped from the result
// (so that multiplying a Montgomery value by the table entry yields a PLAIN value).
static __global__ void __launch_bounds__(256)
k_fr_pow_table(uint32_t* out, const uint32_t* __restrict__ base, const uint32_t* __restrict__ scale,
               uint32_t log_n, uint32_t mode, uint32_t raw_out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t e = i;
    if (mode == 1) e = log_n ? (__builtin_bitreverse32(i) >> (32 - log_n)) : 0;
    Fr v = fr_pow_u32(ld_fr(base), e);
    if (mode) v = mul(v, ld_fr(scale));
    if (raw_out) v = from_mont(v);
    st_fr(out + (size_t)i * 8, v);
}

// plain <-> Montgomery conversion of a scalar array (mode 0: to Montgomery, 1: from)
static __global__ void __launch_bounds__(256)
k_fr_convert(uint32_t* out, const uint32_t* __restrict__ in, uint32_t from, size_t count, uint32_t* bad = nullptr) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr v = ld_fr(in + i * 8);
    if (bad && fr_geq_r(v)) raise_flag(bad, ZK_BAD_SCALAR);
    st_fr(out + i * 8, from ? from_mont(v) : to_mont(v));
}

// stat[0] = smallest index of a scalar that is not < r (0xffffffff: none) - the device form of the host loop a caller's
// plain scalars used to pass through before a multiexp (2^20 scalars: 10 ms on one core, 20 us here)
static __global__ void __launch_bounds__(256)
k_fr_first_noncanonical(const uint32_t* __restrict__ in, size_t count, uint32_t* stat) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (fr_geq_r(ld_fr(in + i * 8))) atomicMin(&stat[0], (uint32_t)i);
}

// out[i] = a[i] * b[i] as raw Montgomery limbs (a*b*R^-1): lets the tests run the reference's
// literal field KATs (fr.rs:1239-1340, fq.rs:2562-2672) through the device multiplier.
template <class C>
static __global__ void __launch_bounds__(64)
k_field_mul_raw(uint32_t* out, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<C> x, y;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        x.l[j] = a[(size_t)i * C::N + j];
        y.l[j] = b[(size_t)i * C::N + j];
    }
    Fp<C> z = mul(x, y);
#pragma unroll
    for (int j = 0; j < C::N; j++) out[(size_t)i * C::N + j] = z.l[j];
}

// Row evaluations of a fixed R1CS: out[mat][p][row] = sum_k coeff[k] * z[p][col[k]] (Montgomery in,
// Montgomery out), one thread per (row, matrix, proof).  Rows >= n_con are bellman's per-input rows
// Input(i) * 0 = 0: a = z_i, b = c = 0.  blockIdx.y = matrix (0..2), blockIdx.z = proof.
struct R1csMat {
    const uint32_t* row_ptr;
    const uint32_t* col;
    const uint32_t* coeff;   // Montgomery
};
static __global__ void __launch_bounds__(256)
k_r1cs_eval(R1csMat ma, R1csMat mb, R1csMat mc, const uint32_t* __restrict__ z, uint32_t* out, uint32_t n_con,
            uint32_t n_in, uint32_t nv, uint32_t n_rows, size_t out_mat_stride) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const uint32_t mat = blockIdx.y;
    const size_t p = blockIdx.z;
    const R1csMat m = mat == 0 ? ma : (mat == 1 ? mb : mc);
    const uint32_t* zp = z + p * (size_t)nv * 8;
    Fr acc = Fr::zero();
    if (row < n_con) {
        for (uint32_t k = m.row_ptr[row]; k < m.row_ptr[row + 1]; k++)
            acc = add(acc, mul(ld_fr(m.coeff + (size_t)k * 8), ld_fr(zp + (size_t)m.col[k] * 8)));
    } else if (mat == 0) {
        acc = ld_fr(zp + (size_t)(row - n_con) * 8);
    }
    st_fr(out + (size_t)mat * out_mat_stride * 8 + (p * n_rows + row) * 8, acc);
}

// Per-proof scalar vectors for the multiexps (plain form):
//   wit_out[p] = [ wit[p][0..nv) | 1 | r | s ]                          (A and B2 multiexps)
//   cvec[p]    = [ h (m, written later) | aux (n_aux) | r * z (nv) | r ]  (merged C multiexp:
//                C' = H + L + r * (B1 + beta_1), one bucket set instead of three)
//                fold != 0 appends [ s * z (nv) | s | r * s ] over the bases of the A query, alpha_1 and delta_1: the job is
//                then C = s * A + C' itself (A = alpha_1 + sum z_i A_i + r delta_1) - for a few proofs made alone, whose
//                final fold would otherwise be a 255-bit double-and-add chain on the critical path (zkamd.cpp prove_chunk)
// tail[p] = (1, r, s).  Witness scalars are converted out of Montgomery form when `mont` is set.
static __global__ void __launch_bounds__(256)
k_build_scalars(uint32_t* wit_out, uint32_t* cvec, const uint32_t* __restrict__ wit, const uint32_t* __restrict__ tail,
                uint32_t nv, uint32_t n_in, uint32_t m, uint32_t cstride, uint32_t mont, uint32_t* bad, uint32_t fold) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv + 3) return;
    const size_t p = blockIdx.y;
    uint32_t* cv = cvec + p * (size_t)cstride * 8;
    const uint32_t n_aux = nv - n_in;
    Fr r = ld_fr(tail + (p * 3 + 1) * 8);   // plain
    Fr sR = fold ? mul(ld_fr(tail + (p * 3 + 2) * 8), Fr::r2()) : Fr::zero();   // s in Montgomery form
    Fr v;
    if (i < nv) {
        Fr raw = ld_fr(wit + (p * nv + i) * 8);
        if (fr_geq_r(raw)) {
            raise_flag(bad, ZK_BAD_SCALAR);
            raw = Fr::zero();   // the call fails at its end; until then the multiexp recoding must see values < r
        }
        if (i == 0 && !(mont ? raw == Fr::one() : (raw.l[0] == 1u && (raw.l[1] | raw.l[2] | raw.l[3] | raw.l[4] | raw.l[5] | raw.l[6] | raw.l[7]) == 0u)))
            raise_flag(bad, ZK_BAD_ONE);   // bellman: alloc_input(ONE = 1) comes first
        Fr rz;
        if (mont) {
            v = from_mont(raw);
            rz = mul(raw, r);                     // (z R)(r) / R = z r
        } else {
            v = raw;
            rz = mul(mul(raw, r), Fr::r2());      // (z r / R)(R^2) / R = z r
        }
        if (i >= n_in) st_fr(cv + (size_t)(m + (i - n_in)) * 8, v);
        st_fr(cv + (size_t)(m + n_aux + i) * 8, rz);
        if (fold) st_fr(cv + (size_t)(m + n_aux + nv + 1 + i) * 8, mul(v, sR));   // (z)(s R) / R = z s
    } else {
        v = ld_fr(tail + (p * 3 + (i - nv)) * 8);
        if (i == nv) st_fr(cv + (size_t)(m + n_aux + nv) * 8, r);   // r * 1 for beta_1
        if (fold && i == nv + 1) st_fr(cv + (size_t)(m + n_aux + 2 * nv + 2) * 8, mul(r, sR));   // r s for delta_1
        if (fold && i == nv + 2) st_fr(cv + (size_t)(m + n_aux + 2 * nv + 1) * 8, v);            // s for alpha_1
    }
    st_fr(wit_out + (p * (nv + 3) + i) * 8, v);
}

}  // namespace zkdev
