// ctg_stem.hip -- two consecutive steps of a contraction stem in one launch (gfx950).
//
//   C1[r1, n1] = sum_k1  A[r1, k1] B1[k1, n1]
//   C2[r2, n2] = sum_k2 C1[r2, k2] B2[k2, n2]          r2 u k2 = r1 u n1
//
// A and C2 are the big tensors of a sliced Sycamore contraction (2^31-2^32
// elements); B1, B2 hold a few hundred to a few thousand.  Run as two steps
// (reference: two turns of the loop in cotengra/contract.py:788-832) the
// intermediate C1 is written to HBM and read back; here it lives in LDS.  The
// planner (cotengra_amd/stem.py) splits the binary index digits of A into tile
// bits -- all of k1, the digits of k2 that are on A, and enough of A's
// lowest-stride digits to make 256 (512) tile rows -- and grid bits; one
// workgroup of 8 waves takes one grid value at a time:
//
//   step 1   wave w gathers rows [32 w, 32 w + 32) x K1 of the tile from HBM, 16 k at
//            a time (a task), STRAIGHT INTO MATRIX-CORE FRAGMENTS -- lane (row l & 31,
//            k parity l >> 5) loads the 8 elements k = 2 j + (l >> 5), two tasks
//            ahead -- and multiplies by B1 (fragments in registers or LDS);
//   barrier  (every wave is done reading the previous tile's intermediate)
//   scatter  the 32 x N1 accumulators go to the shared intermediate tile at
//            mid_row[row] + mid_col[n] = row2 * (K2 + 4) + k2: the layout step 2
//            wants, whatever index permutation lies between the two steps;
//   barrier
//   step 2   work items (32-row tile, 32-column group) of the intermediate are
//            multiplied by B2 and stored: 8 bytes per lane, 256 B runs.
//
// Complex on the real matrix cores, second formulation (the first one is in
// ctg_pair_mfma.hip).  v_mfma_f32_32x32x2_f32: D(32x32) += A'(32x2) B'(2x32).  Here a
// pair of tiles X / Y holds the REAL and the IMAGINARY parts of 32 complex
// columns:   X: A' = (Re a, -Im a), B' rows (Re b, Im b)
//            Y: A' = (Re a,  Im a), B' rows (Im b, Re b)
// so B needs only its two planes in LDS (the interleaved formulation needs four),
// the sign lives in one XOR per A fragment register, and a lane ends up with Re
// and Im of the same element -- an 8-byte store without any lane exchange.  With
// 16 columns both halves share one tile (columns 16-31 = imaginary parts): A' =
// (Re a, Im a), B' rows (Re b | Im b) and (-Im b | Re b), a third plane of 16 x K.
// Step 2 pairs (Re a_k, Im a_k) in the two k-rows of one MFMA (its A' comes from LDS
// planes); step 1 pairs (a_k, a_k+1) of the SAME component -- one MFMA for the real
// parts of two k, one for the imaginary parts -- because that is the shape in which a
// lane's 8-byte gather of one complex element IS a fragment: no LDS transpose of A.
#include "ctg_common.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

// Round 6 -- the SECOND ARITHMETIC of the bf16-pipe kernels, compiled from this same source with -DCTG_STEM_H2
// into a second object (its externals and its kernel renamed: both objects live in one library):
// every fp32 operand as TWO ROUNDED fp16 limbs (22 bits) under a per-tensor power-of-two scale and THREE
// products (h1 h1', h1 h2', h2 h1') on v_mfma_f32_32x32x16_f16, where the bf16 x 3 arithmetic spends three
// limbs and six products.  The pairs are bound by their matrix + split work (profiles/r6_stem_half_products.txt:
// half the products = 222 -> 150 ms per headline slice), so this is where their time goes.  What fp16 lacks is
// RANGE (5 exponent bits): every operand is brought to [2^13, 2^14) by an exact power of two before it is
// split -- the small operands by their largest element (found in-kernel, as before), the big operand A by the
// largest element its PRODUCER recorded (StemArgs::amax: every stem kernel tracks max |re|, |im| of what it
// stores, one v_max3 per value and one atomic per wave; a big operand of any other origin gets a max-abs pass,
// ctg_runtime.hip), the intermediate tile by its own largest element (a wave reduction and eight LDS words per
// tile) -- and the powers go back in where the result is stored.  Elements more than 2^-14 below their tensor's
// largest lose low bits gradually (absolute error <= 2^-24 of the largest): the error is norm-wise, like that
// of any blocked floating-point format; tools/exp_product_levers.py measures it on the narrowed m20 trees.
#ifdef CTG_STEM_H2
#define CTG_STEM_KNAME "stem2h_kernel"
#define stem2_kernel stem2h_kernel
#define stem2_lds_bytes stem2h_lds_bytes
#define stem3_instantiated_c stem3h_instantiated_c
#define stem3_supported stem3h_supported
#define stem2_supported stem2h_supported
#define stem2_variant stem2h_variant
#define stem2_kernel_name stem2h_kernel_name
#define launch_stem2 launch_stem2h
#define stem2_uses_bf3 stem2h_uses_h2
#define ctg_debug_stem_timeline ctg_debug_stem_timeline_h2   // (experiment builds: the same hooks, this object's kernels)
#define ctg_debug_stem_oob ctg_debug_stem_oob_h2
#define ctg_stem_tl ctg_stem_tl_h2
#define ctg_stem_tl_on ctg_stem_tl_on_h2
#define ctg_stem_oob ctg_stem_oob_h2
// B1's fragments live in registers up to this many 16-deep chunks of the first contraction (24 registers per chunk with
// three limbs, 16 with two).  Four chunks under H2 -- the K1 = 64 pairs, whose fragments come from LDS for every task --
// were measured (same box, alternating): 199.5 against 198.3 ms per slice, 0.6 % SLOWER; two it stays.
#define CTG_STEM_BR1_MAX 2
#else
#define CTG_STEM_KNAME "stem2_kernel"
#define CTG_STEM_BR1_MAX 2
#endif

namespace ctg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef CTG_STEM_BOUNDS
// Bounds-checked experiment build (tools/build_variants.py bounds=-DCTG_STEM_BOUNDS, tools/
// check_stem_bounds.py): every gather of the big operand and every store of the result is
// tested against the tensor's extent; a violation is counted ([0] gathers, [1] stores) and
// the access skipped.  The host validates the TABLES (ctg_plan_create); this checks the
// addresses the kernel actually forms from them.
__device__ unsigned long long ctg_stem_oob[2];
#endif

#ifdef CTG_STEM_TIMELINE
// Timeline experiment build (tools/build_variants.py tl=-DCTG_STEM_TIMELINE, tools/exp_stem_timeline.py): the 8 waves
// of workgroup 0 stamp the shader clock at the phase boundaries of their first CTG_TL_TILES tiles -- [wave][tile][0..5]
// = tile start, step 1 issued, past barrier 1, scatter done, past barrier 2, step 2 issued.
#define CTG_TL_TILES 256
__device__ unsigned long long ctg_stem_tl[8][CTG_TL_TILES][6];
__device__ int ctg_stem_tl_on;   // set per launch by the host: CTG_TL_SHAPE="K1,N1,K2,N2" (and the first match only)
#endif

namespace {

constexpr int SW = 8;            // waves per workgroup

// Knock-out switches of experiment builds (tools/build_variants.py; results are wrong by
// construction): what does the kernel cost without its matrix instructions / gathers /
// stores / scatter?  Off in the product.
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
#ifdef CTG_STEM_KO_MFMA
    c[0] = fmaf(a, b, c[0]);   // keeps the data dependences at one VALU op per MFMA
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// row of accumulator register t within a 32-row tile, for the lanes with kk = 0
__device__ __forceinline__ constexpr int rowmap(int t) { return (t & 3) + 8 * (t >> 2); }

__device__ __forceinline__ float flip(float v, unsigned mask) {
#ifdef CTG_STEM_KO_XOR   // (knock-out: what do the sign XORs cost?)
    return v;
#else
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ mask);
#endif
}

// planes of a small operand in LDS: plane p, column n, k contiguous
//   [np][N][K + 4], np = 2 (Re, Im) or 3 (Re, Im, -Im) for 16 columns
// FRAG1 (the first step's operand): within a chunk of 16 k the values of k-row 0 come first,
// then those of k-row 1, each in slot order -- the 8 values a lane multiplies with are two
// 16-byte reads
// (k-row h = k & 1, slot (k & 15) >> 1) -- with 16-byte gathers (vec) the k-row is bit 1 of k
// and the slot ((k & 15) >> 2) * 2 + (k & 1), see cotengra_amd/stem.py: geometry
template <bool FRAG1>
__device__ __forceinline__ void load_b_planes(float* P, const c64* __restrict__ B, const int64_t* off, int K,
                                              int N, bool pack, int tid, bool vec = false) {
    const int LDB = K + 4;
    for (int e = tid; e < K * N; e += SW * 64) {
        const int k = e / N, n = e - k * N;
        const int h = vec ? (k >> 1) & 1 : k & 1;
        const int slot = vec ? (((k & 15) >> 2) << 1) | (k & 1) : (k & 15) >> 1;
        const int kp = FRAG1 ? (k & ~15) + h * 8 + slot : k;
        const c64 v = B[off[e]];
        P[n * LDB + kp] = v.re;
        P[(N + n) * LDB + kp] = v.im;
        if (pack) P[(2 * N + n) * LDB + kp] = -v.im;
    }
}


// ---- fp32 products on the bf16 matrix cores (template argument BF3) ------------------------
// An fp32 value splits EXACTLY into three bfloat16 values (rounded limbs since round 5 -- split3 below; truncated
// ones, 8 + 8 + 8 mantissa bits, before); products of bf16 values are exact in fp32, so a real multiply-add becomes
// the 6 cross terms above 2^-24 (the three smallest of the nine are dropped: below 2^-26 of the product with
// rounded limbs, tools/exp_bf16x3.py) accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 -- 16 k per instruction at 16x the fp32 MFMA rate, i.e. 2.7x the fp32
// matrix peak (measured with the splitting: 1.7x, tools/exp_bf16x3_rate.py).  A lane holds 8
// values of k per operand; which 8 is the same function of (lane half, position) for both
// operands, so the k order inside the instruction does not matter.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// the six products kept, (limb of a, limb of b): those that need only the FIRST limb of the operand
// being split come first -- it is a byte permute of the words as they arrive, the MFMAs can start while
// the other two limbs are still being subtracted out
// (experiment build -DCTG_STEM_KO_HALF: only the three products a TWO-limb split would keep -- t = 0, 1, 3 --, the third
// limbs dead code: what halving the product count is worth in time; the results lose their third limb)
#ifdef CTG_STEM_H2
#define CTG_STEM_LIMBS 2
#else
#define CTG_STEM_LIMBS 3
#endif
#if defined(CTG_STEM_KO_HALF) || defined(CTG_STEM_H2)   // (H2: limbs 0, 1 only -- products (0, 0), (0, 1), (1, 0))
#define CTG_STEM_T_STEP(t) ((t) == 1 ? 2 : ((t) == 3 ? 3 : 1))
#define CTG_STEM_T_COUNT 3                        // products per k-block, and which of them t is (deferred stores go
#define CTG_STEM_T_INDEX(t) ((t) == 3 ? 2 : (t))  // out in as many portions)
#else
#define CTG_STEM_T_STEP(t) 1
#define CTG_STEM_T_COUNT 6
#define CTG_STEM_T_INDEX(t) (t)
#endif
__device__ __forceinline__ constexpr int bf3_ta(int t) { return t < 3 ? 0 : (t == 5 ? 2 : 1); }
__device__ __forceinline__ constexpr int bf3_tb(int t) { return t == 1 || t == 4 ? 1 : (t == 2 ? 2 : 0); }

// Two values -> the packed pair of their bf16 roundings (round to nearest even; lo = a, hi = b).
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#ifdef CTG_STEM_H2
// H2: x * scale as two rounded fp16 limbs (v_cvt_pk_f16_f32 rounds to nearest even and packs two values; the
// residual x s - h1 is exact in fp32); o[2] is not used by any product.  4 vector instructions per value.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 (&o)[3], float scale = 1.f) {
    u32x4 p1, p2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i] * scale, b = x[2 * i + 1] * scale;
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        p1[i] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
        p2[i] = pack_f16(a - (float)ha, b - (float)hb);
    }
    o[0] = __builtin_bit_cast(bf16x8, p1);
    o[1] = __builtin_bit_cast(bf16x8, p2);
    o[2] = o[1];
}
#else
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 (&o)[3], float = 1.f) {
    // Round 5: ROUNDED limbs.  l1 = rn(x), l2 = rn(x - l1), l3 = x - l1 - l2: the remainder after two rounded limbs
    // has at most 7 significant bits, so x = l1 + l2 + l3 stays EXACT, and the three cross terms that are not
    // computed (l2 m3, l3 m2, l3 m3) are below 2^-26 of the product with either sign -- truncated limbs (round 3-4)
    // leave up to 2^-23 of ONE sign there, which is where the bf16 x 3 kernels' 1.1-1.2 x the fp32 kernel's error
    // came from.  Same instruction count as the truncating split (5.5 per value): v_cvt_pk_bf16_f32 rounds and packs two
    // values at once (no byte permute for the first two limbs), a shift / a mask turn the pair back into floats for
    // the subtractions, one permute packs the third limbs.  (|x| within 2^-9 of the largest float rounds to inf:
    // inputs that large lost a power of two at upload, ctg_kernels_valu.hip: prescale_inputs_kernel.)
    u32x4 p1, p2, p3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        p1[i] = cvt_pk_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, p1[i] << 16), rb = b - __builtin_bit_cast(float, p1[i] & 0xffff0000u);
        p2[i] = cvt_pk_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, p2[i] << 16), sb = rb - __builtin_bit_cast(float, p2[i] & 0xffff0000u);
        p3[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
    }
    o[0] = __builtin_bit_cast(bf16x8, p1);
    o[1] = __builtin_bit_cast(bf16x8, p2);
    o[2] = __builtin_bit_cast(bf16x8, p3);
}
#endif

__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef CTG_STEM_KO_MFMA
    c[0] = fmaf((float)a[0], (float)b[0], c[0]);
    return c;
#elif defined(CTG_STEM_H2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// bf16 x 3 planes of a small operand in LDS -- (Re, Im), plus -Im with 16 columns, as in fp32 --
// fragment-ready: a lane's 8 values of one split are 16 contiguous bytes.
//   step 1:  [plane][n][chunk of 16 k][k-row h][split][slot]     row = (K / 16) * 48 + 8 values
//   step 2:  [plane][n][block of 8 k][split][k & 7]               row = (K / 8) * 24 + 8 values
__device__ __forceinline__ int bf3_row(int K, bool step1) { return step1 ? (K >> 4) * 48 + 8 : (K >> 3) * 24 + 8; }
// (scale: a power of two that brings an operand from the bottom / top of the fp32 range to O(1)
// before it is split -- exact; see bf3_operand_exponent)
template <bool STEP1>
__device__ __forceinline__ void load_b_planes_bf3(unsigned short* Q, const c64* __restrict__ B, const int64_t* off, int K,
                                                  int N, int planes, int tid, bool vec, float scale) {
    const int ROW = bf3_row(K, STEP1);
    for (int e = tid; e < K * N; e += SW * 64) {
        const int k = e / N, n = e - k * N;
        int at;
        if (STEP1) {
            const int h = vec ? (k >> 1) & 1 : k & 1;
            const int slot = vec ? (((k & 15) >> 2) << 1) | (k & 1) : (k & 15) >> 1;
            at = (k >> 4) * 48 + h * 24 + slot;
        } else {
            at = (k >> 3) * 24 + (k & 7);
        }
        const c64 v = B[off[e]];
        const float vals[3] = {v.re * scale, v.im * scale, -v.im * scale};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (pl >= planes) break;
            const float x = vals[pl];   // (rounded limbs, as split3)
#ifdef CTG_STEM_H2
            {
                const _Float16 f1 = (_Float16)x, f2 = (_Float16)(x - (float)f1);
                unsigned short* dh = Q + (pl * N + n) * ROW + at;
                dh[0] = __builtin_bit_cast(unsigned short, f1);
                dh[8] = __builtin_bit_cast(unsigned short, f2);
                dh[16] = 0;
                continue;
            }
#endif
            const unsigned h1 = cvt_pk_bf16(x, 0.f) << 16;
            const float r1 = x - __builtin_bit_cast(float, h1);
            const unsigned h2 = cvt_pk_bf16(r1, 0.f) << 16;
            const float r2 = r1 - __builtin_bit_cast(float, h2);
            unsigned short* d = Q + (pl * N + n) * ROW + at;
            d[0] = (unsigned short)(h1 >> 16);
            d[8] = (unsigned short)(h2 >> 16);
            d[16] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
        }
    }
}

// The three-way split is exact as long as the third limb (2^-16 of the value) is a bf16 number,
// i.e. for |x| >= 2^-110 (measured: below that the limb is lost and the products carry a relative
// error of up to 2^-15, tests/test_gpu_round4.py).  A SMALL operand whose largest element lies
// outside [2^-64, 2^64) is therefore multiplied by a power of two that brings it to [1, 2) before
// the split (exact), and the power goes into the factor the stores apply (alpha).  Returns the
// exponent to REMOVE (0: leave the operand alone).  All threads of the workgroup call it.
__device__ __forceinline__ int bf3_operand_exponent(const c64* __restrict__ B, const int64_t* off, int n_el, int tid,
                                                    float* red) {
    float mx = 0.f;
    for (int e = tid; e < n_el; e += SW * 64) {
        const c64 v = B[off[e]];
        mx = fmaxf(mx, fmaxf(fabsf(v.re), fabsf(v.im)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < SW; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();   // (red is reused for the other operand)
    int ex = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;
#ifdef CTG_STEM_H2
    // fp16 limbs: the largest element always goes to [2^13, 2^14) (fp16 holds up to 2^16; the rounding of the first
    // limb and the sums of a few terms stay clear of it)
    if (mx == 0.f || ex == 128) return 0;
    ex -= 13;
    return ex < -126 ? -126 : (ex > 126 ? 126 : ex);
#endif
    if (mx == 0.f || ex == 128 || (ex >= -64 && ex < 64)) return 0;   // (zero, inf / nan, or fine as it is)
    return ex < -126 ? -126 : (ex > 126 ? 126 : ex);
}
// H2: the exponent to REMOVE from a tensor whose largest |component| is mx (its producer's record)
__device__ __forceinline__ int h2_exponent_of(float mx) {
    const int ex = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;
    if (mx == 0.f || ex == 128 || ex == -127) return 0;   // (zero / inf / nan / subnormal: left alone)
    const int e = ex - 13;
    return e < -126 ? -126 : (e > 126 ? 126 : e);
}
__device__ __forceinline__ float pow2f(int ex) {   // 2^ex, -126 <= ex <= 127
    return __builtin_bit_cast(float, (unsigned)(ex + 127) << 23);
}

}  // namespace

// "use" a value: the compiler has to wait here for the load that produced it, not at
// its first use inside the tile loop (where s_waitcnt vmcnt(0) would drain the gathers
// and stores in flight once per tile)
#ifdef CTG_STEM_KO_BARRIER
#define CTG_STEM_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define CTG_STEM_SYNC() __syncthreads()
#endif

#ifdef CTG_STEM_TIMELINE
#define CTG_TL_STAMP(k)                                                                                   \
    do {                                                                                                  \
        if (blockIdx.x == 0 && lane == 0 && tl_n < CTG_TL_TILES && ctg_stem_tl_on)                        \
            ctg_stem_tl[wave][tl_n][k] = __builtin_readcyclecounter();                                    \
        if ((k) == 5) ++tl_n;                                                                             \
    } while (0)
#else
#define CTG_TL_STAMP(k) do {} while (0)
#endif

template <typename T>
__device__ __forceinline__ void settle(T& v) {
    asm volatile("" : "+v"(v));
}

// PACK1 / PACK2: the first / second step has 16 output columns (one tile holds Re | Im).
// RT1: units of step 1 per wave and tile (1 or 2).  CS1: 32-column groups of step 1 -- a
// unit is (32-row tile, column group), so 64 / 128 columns mean 128 / 64 tile rows whose
// row tiles are each taken by 2 / 4 waves (each gathers the rows itself: the second
// fetch comes from the L1 / L2, the matrix cores stay evenly loaded).
// NCH, IT2: 16-deep chunks of the first contraction and work items of step 2 per wave,
// known at compile time -- s_waitcnt vmcnt is positional and counts stores too, so only
// a tile whose sequence of gathers and stores is fixed lets the compiler wait for a
// gather issued one tile ago WITHOUT waiting for the stores issued since (see the
// steady-state loop of ctg_pair_mfma.hip's streaming kernel).  NCH = 0: both counts are
// run-time values (any shape; every wait drains the queue).
// BR1: the B1 fragments of this wave's columns live in registers for the whole kernel
// (K1 floats: Re and Im of the K1 / 2 values of k of the lane's parity) -- K2Q > 0: likewise
// the B2 fragments of this wave's column group (2 K2 floats, K2 with 16 columns), K2 = 4 K2Q
// known at compile time.  The fragments are the same for every tile; re-reading them from LDS
// for every 8 MFMAs cost 8 % of a slice (knock-out CTG_STEM_KO_BFRAG, profiles/
// r3_stem_knockout.txt).  Chosen per shape by the register budget (stem2_shape: at most 96
// floats of B per lane).
// VEC: A's stride-1 digit is a contracted one -- a lane gathers two adjacent k in one 16-byte load.
// BF3: both steps multiply on the bf16 matrix cores (three-way split, see above); B1 fragments
// in registers if BR1 (24 registers per chunk), B2 fragments from LDS (K2Q then only says that
// K2 is known at compile time).
// RI2 (round 4; >= 32 columns in step 2, static shapes, fp32 products): step 2 in the
// ROW-INTERLEAVED form.  A 32 x 32 matrix-core tile holds 16 complex rows -- tile row 2 i is
// Re c_i, row 2 i + 1 is Im c_i -- times 32 complex columns:
//     A' row 2 i     = (Re a_ik, -Im a_ik)        B' = (Re b_kn ; Im b_kn)
//     A' row 2 i + 1 = (Im a_ik,  Re a_ik)
// one MFMA per complex k (no wasted flops, as before), but B' is ONE value per k and lane (its
// k-row's plane) instead of two, so the B2 fragments of a wave's column group fit the registers
// up to K2 = 64 (K2 floats; the X / Y form needs 2 K2), the sign lives in a third plane (-Im) of
// the intermediate written once by the scatter instead of one XOR per MFMA, and a lane's
// accumulator registers (t, t + 1) ARE (Re, Im) of one element: the 8-byte stores take them
// where they are -- no copies into a staging array (32 moves per item in the X / Y form), so
// the deferred stores only need the accumulators to stay untouched until they are issued (two
// accumulator sets alternate when a wave has several items per tile).  A work item is still 32
// complex rows x 32 columns: two accumulators (rows 0-15, 16-31), each with its own A' fragment
// (one ds_read_b128 per 4 k and accumulator).  Probe of the two forms in isolation
// (csrc/tools/ctg_probe_loop.hip, profiles/r4_loop_probe.txt): K2 = 64 0.839 -> 0.872 of the
// fp32 matrix peak, K2 = 32 0.728 -> 0.797.  A wave keeps ONE column group for all its items
// (item = (row tile, column group) with the column group = wave % ng2).
// ONE (round 4): the first half alone -- a large step no pair took (a chain of odd length leaves one
// over): gather -> MFMA as in step 1, then the 32 x 32 accumulators of a unit go straight to the result
// (8-byte stores, issued between the MFMAs of the wave's next unit like every deferred store here).  No
// intermediate, no barrier in the tile loop, LDS for B1 only.  >= 32 columns.
// ITM > 0 (round 4, opt-in: CTG_STEM_TRIPLES): a MIDDLE stage between the two -- a three-step tile.  The
// fields and code of "step 2" then are the LAST step's; the middle step reads the first intermediate
// ([rowsM][ldM], written by the scatter of step 1), multiplies by BM (ITM work items per wave, accumulators
// kept in registers), and after a barrier scatters its result over it as the second intermediate
// ([rows2][ld2], mid2_row[rowM] + mid2_col[nM]) which the last step reads.  PACKM: its 16 columns.  Static
// shapes, X / Y form; four barriers per tile instead of two.
// XM (round 5; bf16 x 3): no sign flips of limbs.  The real part of a product is kept as TWO accumulators --
// Xp += Re a Re b, Xm += Im a Im b, X = Xp - Xm where the tile is handed on (16 subtractions per 32 x 32
// tile instead of 12 XORs per 16 k of either step) -- and step 2 multiplies 16 k of ONE component per
// instruction (the lane halves take k-blocks 2 c and 2 c + 1 of the same plane) instead of pairing Re | Im.
// LM (round 5; needs XM): the intermediate lives in LDS as bf16 LIMBS, fragment-ready for step 2 --
// [plane][row2][block of 8 k][limb][k & 7], a lane's 8 values of one limb are 16 contiguous bytes -- split
// ONCE by the scatter (two ANDs, two subtractions per value, three 2-byte writes that take the high halves
// where they are) instead of once per work item of step 2 (5.5 instructions per value and per column group:
// profiles/r4_stem_sq_counters.txt counted 5.6 vector instructions per MFMA, most of them this).  Half again
// as much LDS as the fp32 intermediate: B1's planes share its memory when the fragments live in registers.
// WS (round 5; needs XM, a pair, static): the waves SPECIALISE.  Waves 0-3 (one per SIMD) are producers: they
// gather, run step 1 for ALL units of the tile (twice RT1_ each) and scatter; waves 4-7 are consumers: they run
// step 2 for all its work items (twice IT2_ each) and store.  The producers work one tile ahead -- step 1 of tile
// t + 1 (registers only: B1's fragments live there) overlaps step 2 of tile t -- so that the two waves of a SIMD
// are never in the same phase: the phase timeline of the symmetric kernel (profiles/r5_stem_timeline_knockout.txt)
// shows its matrix phases at 55-64 % of the issue rate -- both waves of a SIMD wait for the LDS or split operands
// at the same moments -- and 24 % of a tile outside them (barrier waits, scatter).  Same tile, same tables, same
// LDS; the only serial part left is the producers' scatter between the two barriers (the consumers drain
// their pending stores there).  A consumer has the registers for two fragment sets: the loads and splits of
// chunk c + 1 go out before the MFMAs of chunk c.
#ifndef CTG_STEM_WS_DEPTH
#define CTG_STEM_WS_DEPTH 2
#endif
#ifndef CTG_STEM_DEPTH
#define CTG_STEM_DEPTH 2
#endif
template <bool PACK1, bool PACK2, int RT1_, int CS1, int NCH, int IT2_, bool BR1 = false, int K2Q = 0, bool VEC = false,
          bool BF3 = false, bool RI2 = false, bool ONE = false, int ITM = 0, bool PACKM = false, bool XM = false,
          bool LM = false, bool WS = false>
__global__ __launch_bounds__(SW * 64, 1) void stem2_kernel(StemArgs p) {
    constexpr bool TRI = ITM > 0;
    static_assert(!WS || (XM && !ONE && IT2_ > 0 && IT2_ <= 2 && (PACK1 || RT1_ == 1)), "specialised waves: a static 16-bit pair");
    constexpr int PW = WS ? 4 : SW;               // waves that run step 1 (and, symmetric kernel, step 2)
    constexpr int RT1 = WS ? 2 * RT1_ : RT1_;     // units of step 1 per such wave
    constexpr int IT2 = WS ? 2 * IT2_ : IT2_;     // work items of step 2 per consumer (symmetric: per wave)
    static_assert(!XM || (BF3 && !TRI && NCH > 0), "two-accumulator real part: bf16 x 3, static, no three-step tile");
    static_assert(!LM || (XM && !ONE), "limb intermediate: the round-5 form of a pair");
    // step 1 keeps Xm (16 columns: the sign lives in B1's third plane; specialised waves: a producer holds the
    // accumulators of TWO units until the barrier -- a third one per unit does not fit next to B1's fragments, and the
    // 12 sign flips per task cost a producer nothing: it has half a tile of slack)
    constexpr bool XM1 = XM && !PACK1 && !WS;
    constexpr bool XM2 = XM && !PACK2;   // step 2 likewise
    static_assert(!TRI || (NCH > 0 && IT2 > 0 && !RI2 && !ONE && K2Q == 0), "three-step tile: static, X / Y form");
    static_assert(!ONE || (!PACK1 && !PACK2 && !RI2 && IT2 == 0 && K2Q == 0), "one step: >= 32 columns, nothing of step 2");
    static_assert(!PACK1 || CS1 == 1, "16 columns are one group");
    static_assert(!BR1 || NCH > 0, "B1 in registers needs the chunk count at compile time");
    static_assert(K2Q == 0 || PACK2 || IT2 == 1 || RI2, "B2 in registers: one column group per wave");
    static_assert(!RI2 || (!PACK2 && !BF3 && NCH > 0 && IT2 > 0), "row-interleaved step 2: fp32, >= 32 columns, static");
    constexpr int RTW = PW / CS1;   // row tiles the waves of step 1 cover at once
    constexpr bool STATIC = NCH > 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (K2 is a compile-time constant where B2's fragments live in registers: LDS offsets that are
    // multiples of the intermediate's row length then fold into the instructions' immediates)
    const int K1 = p.K1, N1 = p.N1, K2 = K2Q > 0 ? 4 * K2Q : p.K2, N2 = p.N2;
    const int LDB1 = K1 + 4, LDB2 = K2 + 4, LD2 = K2Q > 0 ? 4 * K2Q + 4 : p.ld2;
    const int PLANE = p.rows2 * LD2;                       // floats per plane of the intermediate
    // (three-step tile: the middle step's operand, and the FIRST intermediate [rowsM][LDM])
    const int KM = TRI ? p.KM : 0, NM = TRI ? p.NM : 0, LDM = TRI ? p.ldM : 0, LDBM = KM + 4;
    const int PLANEM = TRI ? p.rowsM * LDM : 0;
    const int PLANE1 = TRI ? PLANEM : PLANE;               // plane of the intermediate step 1 scatters into
    float* P1 = (float*)smem;                              // [2|3][N1][LDB1]
    float* PM = P1 + (PACK1 ? 3 : 2) * N1 * LDB1;          // [2|3][NM][LDBM] (three-step tile)
    float* P2 = PM + (TRI ? (PACKM ? 3 : 2) * NM * LDBM : 0);   // [2|3][N2][LDB2]
    float* mid = P2 + (PACK2 ? 3 : 2) * N2 * LDB2;         // [2][rows2][LD2]
    // RI2: three planes (Re, Im, -Im) INTERLEAVED PER ROW -- [rows2][3][LD2], row pitch RP = 3 LD2:
    // the three values of an element are LD2 floats apart, an immediate offset of the scatter's
    // ds_write (16 address registers instead of 48; separate planes with or without a bank shift
    // and other pitches measured the same to 1 %, profiles/r4_loop_probe.txt) -- and, with B2 in
    // registers, the staging planes of B2 share the intermediate's memory (they are dead once the
    // fragments are loaded, before the first scatter's barrier)
    const int RP = 3 * LD2;
    // an offset row2 * LD2 + k2 of the planner's tables in that layout: row2 * RP + k2
    auto ri_off = [&](int e) __attribute__((always_inline)) { return RI2 ? (e / LD2) * RP + e % LD2 : e; };
    int mid_floats = 2 * (PLANE > PLANEM ? PLANE : PLANEM);   // (the two intermediates of a three-step tile share it)
    if constexpr (RI2) {
        mid_floats = 3 * PLANE;
        if constexpr (K2Q > 0) {
            mid = P2;
            const int p2f = 2 * N2 * LDB2;
            mid_floats = mid_floats > p2f ? mid_floats : p2f;
        }
    }
    // (BF3: the small operands as bf16 x 3 planes instead)
    const int ROW1 = bf3_row(K1, true), ROW2 = bf3_row(K2, false), ROWM = TRI ? bf3_row(KM, false) : 0;
    unsigned short* Q1 = (unsigned short*)smem;            // [2|3][N1][ROW1]
    unsigned short* QM = Q1 + (PACK1 ? 3 : 2) * N1 * ROW1; // [2|3][NM][ROWM] (three-step tile)
    unsigned short* Q2 = QM + (TRI ? (PACKM ? 3 : 2) * NM * ROWM : 0);   // [2|3][N2][ROW2]   (LM + BR1: see below)
    if constexpr (BF3) mid = (float*)(Q2 + (PACK2 ? 3 : 2) * N2 * ROW2);
    // LM: the intermediate as limbs -- [row2][block of 8 k][Re | Im][limb][k & 7]: a value's Im part sits 48
    // bytes behind its Re part, its limbs 16 and 32 bytes behind the first (immediate offsets of the scatter's
    // writes, whatever K2 is: 16 address registers in all); row pitch RPS shorts (K2 / 8 blocks of 48 + 8 of
    // padding: 16-byte reads of 32 consecutive rows hit every bank once).  B1's planes, dead once its
    // fragments are in registers (before the first scatter's barrier), lie over it
    const int RPS = LM ? (K2 >> 3) * 48 + 8 : 0;
    constexpr int PLS = 24;   // Re -> Im, in shorts
    if constexpr (LM) {
        if constexpr (BR1) {
            Q2 = (unsigned short*)smem;
            mid = (float*)(Q2 + (PACK2 ? 3 : 2) * N2 * ROW2);
            Q1 = (unsigned short*)mid;
        }
        mid_floats = (p.rows2 * RPS) >> 1;   // (rows2 x RPS shorts)
        if constexpr (BR1) {
            const int q1f = ((PACK1 ? 3 : 2) * N1 * ROW1 + 1) >> 1;
            mid_floats = mid_floats > q1f ? mid_floats : q1f;
        }
    }
    unsigned short* const midq = (unsigned short*)mid;
    int64_t* oc_s = (int64_t*)(mid + ((mid_floats + 1) & ~1));   // [N2] column offsets of the result

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5;
    const int l31 = lane & 31;
    const bool producer = !WS || wave < PW;
    const int wave1 = WS ? (wave & (PW - 1)) : wave;   // index among the waves of its role
    const int wrt = wave1 / CS1;           // this wave's row tile (within a round of RTW)
    const int wcol = (wave1 % CS1) * 32;   // ... and first column of step 1

    const int64_t z = (int64_t)p.z0 + blockIdx.y;
    const c64* __restrict__ A = (const c64*)p.A + (sload64(p.soffA + z * p.zsA) + z * p.zA);
    const c64* __restrict__ B1 = (const c64*)p.B1 + (sload64(p.soffB1 + z * p.zsB1) + z * p.zB1);
    const c64* __restrict__ B2 = (const c64*)p.B2 + (sload64(p.soffB2 + z * p.zsB2) + z * p.zB2);
    float* __restrict__ C = (float*)((c64*)p.C + (sload64(p.soffC + z * p.zsC) + z * p.zC));
    const c64* __restrict__ BM = TRI ? (const c64*)p.BM + (sload64(p.soffBM + z * p.zsBM) + z * p.zBM) : nullptr;

    int bf3_ex = 0;   // BF3: power of two taken out of the small operands (goes back in through alpha)
    float* const bf3_red = (float*)(oc_s + (ONE ? N1 : N2));   // (64 bytes behind the column table: stem2_lds_bytes_bf3)
#ifdef CTG_STEM_H2
    static_assert(!BF3 || (!TRI && (!LM || WS)), "fp16 x 2: pairs and single steps; a limb intermediate only on specialised waves");
    // H2: the big operand's power of two, from the largest element its producer recorded
    float h2_sa = 1.f;
    int h2_exa = 0;
    if constexpr (BF3) {
        h2_exa = p.amax != nullptr ? h2_exponent_of(read_max(p.amax)) : 0;
        h2_exa = __builtin_amdgcn_readfirstlane(h2_exa);
        h2_sa = pow2f(-h2_exa);
    }
    float h2_st = 1.f;    // ... and the intermediate tile's (per tile)
#else
    constexpr float h2_sa = 1.f, h2_st = 1.f;
#endif
    // (both 16-bit arithmetics) largest |component| this lane has stored -> StemArgs::cmax: what a consumer in the
    // fp16 x 2 arithmetic scales its split of this result with
    float h2_vmax = 0.f;
    if constexpr (BF3) {
        const int ex1 = bf3_operand_exponent(B1, p.b1_off, K1 * N1, tid, bf3_red);
        const int ex2 = ONE ? 0 : bf3_operand_exponent(B2, p.b2_off, K2 * N2, tid, bf3_red);
        int exm = 0;
        if constexpr (TRI) {
            exm = bf3_operand_exponent(BM, p.bm_off, KM * NM, tid, bf3_red);
            load_b_planes_bf3<false>(QM, BM, p.bm_off, KM, NM, PACKM ? 3 : 2, tid, false, pow2f(-exm));
        }
        bf3_ex = ex1 + ex2 + exm;
        load_b_planes_bf3<true>(Q1, B1, p.b1_off, K1, N1, PACK1 ? 3 : 2, tid, VEC, pow2f(-ex1));
        if constexpr (!ONE) load_b_planes_bf3<false>(Q2, B2, p.b2_off, K2, N2, PACK2 ? 3 : 2, tid, false, pow2f(-ex2));
    } else {
        load_b_planes<true>(P1, B1, p.b1_off, K1, N1, PACK1, tid, VEC);
        if constexpr (!ONE) load_b_planes<false>(P2, B2, p.b2_off, K2, N2, PACK2, tid);
        if constexpr (TRI) load_b_planes<false>(PM, BM, p.bm_off, KM, NM, PACKM, tid);
    }
    const int NOUT = ONE ? N1 : N2;   // columns of the result
    for (int n = tid; n < NOUT; n += SW * 64) oc_s[n] = p.out_col[n];

    // ---- per-lane constants ---------------------------------------------------
    // gather: this lane is (row l31, k parity kk) of every task; slot j = element k = 2 j + kk
    // at  task base + kj[j] (uniform) + a_lane (bytes, 32 bits: the planner sees to it)
    unsigned a_lane = (unsigned)(p.lane_a[lane] * 8);
    settle(a_lane);
    int64_t kj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kj[j] = sload64(p.kj_a + j);
    // the X tile (real parts) takes Re a Re b - Im a Im b: the sign of Im a
    const unsigned sgn = 0x80000000u;
    // B fragments of step 1: the 8 values of a chunk this lane multiplies with, per plane.
    //   32 columns: X += re * b1p + (-im) * b1q,  Y += re * b1q + im * b1p   (b1p = Re b, b1q = Im b)
    //   16 columns: lanes 0-15 hold real parts (b1p = Re b, b1q = -Im b), lanes 16-31 imaginary
    //   parts (b1p = Im b, b1q = Re b):  X += re * b1p + im * b1q
    const float* b1p;
    const float* b1q;
    if (PACK1) {
        const int n = l31 & 15, h = l31 >> 4;
        b1p = P1 + ((h ? 1 : 0) * N1 + n) * LDB1 + kk * 8;
        b1q = P1 + ((h ? 0 : 2) * N1 + n) * LDB1 + kk * 8;
    } else {
        b1p = P1 + (wcol + l31) * LDB1 + kk * 8;
        b1q = P1 + (N1 + wcol + l31) * LDB1 + kk * 8;
    }
    const unsigned sgn2 = kk ? 0x80000000u : 0u;   // step 2: A' is (Re, +-Im) by lane half
    // BF3 fragment bases (planes 0 = Re, 1 = Im, 2 = -Im; see load_b_planes_bf3)
    const unsigned short* q1p;
    const unsigned short* q1q;
    const unsigned short* q2x;
    const unsigned short* q2y = nullptr;
    {
        const int h16 = l31 >> 4;
        if (PACK1) {
            q1p = Q1 + ((h16 ? 1 : 0) * N1 + (l31 & 15)) * ROW1 + kk * 24;
            q1q = Q1 + ((h16 ? 0 : 2) * N1 + (l31 & 15)) * ROW1 + kk * 24;
        } else {
            q1p = Q1 + (wcol + l31) * ROW1 + kk * 24;
            q1q = Q1 + (N1 + wcol + l31) * ROW1 + kk * 24;
        }
        if (PACK2) {
            const int plane = kk == 0 ? (h16 ? 1 : 0) : (h16 ? 0 : 2);
            q2x = Q2 + (plane * N2 + (l31 & 15)) * ROW2;
        } else {   // X: (Re a, -Im a) x (Re b, Im b);  Y: (Re a, Im a) x (Im b, Re b)
            q2x = Q2 + ((kk ? 1 : 0) * N2 + l31) * ROW2;
            q2y = Q2 + ((kk ? 0 : 1) * N2 + l31) * ROW2;
        }
    }
    const float* b2x;
    const float* b2y = nullptr;
    if (PACK2) {
        const int n = l31 & 15, h = l31 >> 4;
        const int plane = kk == 0 ? (h ? 1 : 0) : (h ? 0 : 2);
        b2x = P2 + (plane * N2 + n) * LDB2;
    } else {
        b2x = P2 + ((kk ? 1 : 0) * N2 + l31) * LDB2;
        b2y = P2 + ((kk ? 0 : 1) * N2 + l31) * LDB2;
    }
    // LM: an offset row2 * LD2 + k2 of the planner's tables in the limb layout (shorts): row2 * RPS + block * 48 + k2 % 8
    auto lm_off = [&](int e) __attribute__((always_inline)) {
        const int r2 = e / p.ld2, k2 = e - r2 * p.ld2;
        return r2 * RPS + (k2 >> 3) * 48 + (k2 & 7);
    };
    auto mid_off = [&](int e) __attribute__((always_inline)) { return LM ? lm_off(e) : ri_off(e); };
    // scatter of the step-1 accumulators: lane part of mid_row[row] + mid_col[n]
    // (RI2 with 16 columns: the lanes of columns 16-31 hold imaginary parts -> plane Im, and
    // once more negated -> plane -Im)
    int mid_lane = 0;
    if constexpr (!ONE) {
        if (PACK1) mid_lane = mid_off((int)p.mid_col[l31 & 15]) + (l31 >> 4) * (LM ? PLS : RI2 ? LD2 : PLANE1) + mid_off((int)p.mid_row[4 * kk]);
        else mid_lane = mid_off((int)p.mid_col[wcol + l31]) + mid_off((int)p.mid_row[4 * kk]);
    }
    settle(mid_lane);
    // (accumulator register t is row rowmap(t) = bits 0, 1, 3, 4 of t's four bits: the tables are
    // additive over binary digits, so four entries each and a few scalar adds where they are
    // used replace 16-entry arrays that did not fit the scalar registers)
    int mid_o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) mid_o[b] = ONE ? 0 : mid_off((int)sload64(p.mid_row + (b < 2 ? 1 << b : 2 << b)));
    // (the row tiles' parts, per unit of this wave: scalar, fixed for the whole kernel)
    int mid_rt[RT1];
    int64_t one_rt[RT1];   // ONE: the result's offset of this wave's row tile, per unit
#pragma unroll
    for (int m = 0; m < RT1; ++m) {
        mid_rt[m] = ONE ? 0 : __builtin_amdgcn_readfirstlane(mid_off((int)sload64(p.mid_row + 32 * (wrt + RTW * m))));
        one_rt[m] = ONE ? sload64(p.out_row + 32 * (wrt + RTW * m)) : 0;
    }
    // ONE: this lane's column of the result (its column of step 1)
    const int64_t one_col = ONE ? p.out_col[wcol + l31] : 0;
    auto mid_t = [&](int t) __attribute__((always_inline)) {
        return ((t & 1) ? mid_o[0] : 0) + ((t & 2) ? mid_o[1] : 0) + ((t & 4) ? mid_o[2] : 0) + ((t & 8) ? mid_o[3] : 0);
    };
    // store of the step-2 accumulators: lane part of out_row[row2] + out_col[n2]
    // (PACK2: the lanes of columns 16-31 take the odd rows of each row pair)
    // RI2: register pair p = t >> 1 of accumulator a is complex row (p & 1) + 2 kk + 4 (p >> 1)
    // + 16 a of the item: the lane's k-row is bit 1, the pair's bits are bits 0, 2, 3, the
    // accumulator bit 4 (out_o[3] = out_row[16])
    int64_t out_lane = p.out_row[RI2 ? 2 * kk : 4 * kk];
    if (PACK2) out_lane += (l31 >> 4) ? p.out_row[1] : 0;
    settle(out_lane);
    int64_t out_o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) out_o[b] = sload64(p.out_row + (RI2 ? (b == 0 ? 1 : 2 << b) : (b < 2 ? 1 << b : 2 << b)));
    auto out_t = [&](int t) __attribute__((always_inline)) {
        return ((t & 1) ? out_o[0] : 0) + ((t & 2) ? out_o[1] : 0) + ((t & 4) ? out_o[2] : 0) + ((t & 8) ? out_o[3] : 0);
    };
    // three-step tile, middle stage: this wave's items (item = wave + SW i -> column group item / n_rtM,
    // row tile item % n_rtM) are the same for every tile -- their B fragments' bases and the scatter
    // addresses of their results in the second intermediate are kernel constants
    const int n_rtM = TRI ? p.rowsM >> 5 : 1;
    const unsigned short* qmx[TRI ? ITM : 1];
    const unsigned short* qmy[TRI ? ITM : 1];
    const float* bmx[TRI ? ITM : 1];
    const float* bmy[TRI ? ITM : 1];
    int m2_lane[TRI ? ITM : 1], m2_rt[TRI ? ITM : 1], am_row[TRI ? ITM : 1];
    int m2_o[4] = {0, 0, 0, 0};
    if constexpr (TRI) {
        const int h16 = l31 >> 4;
#pragma unroll
        for (int i = 0; i < ITM; ++i) {
            const int item = wave + SW * i;
            const int cg = item / n_rtM, rtm = item - cg * n_rtM;
            am_row[i] = (rtm * 32 + l31) * LDM;
            if (PACKM) {
                const int plane = kk == 0 ? (h16 ? 1 : 0) : (h16 ? 0 : 2);
                qmx[i] = QM + (plane * NM + (l31 & 15)) * ROWM;
                qmy[i] = nullptr;
                bmx[i] = PM + (plane * NM + (l31 & 15)) * LDBM;
                bmy[i] = nullptr;
                m2_lane[i] = (int)p.mid2_col[l31 & 15] + h16 * PLANE + (int)p.mid2_row[4 * kk];
            } else {
                qmx[i] = QM + ((kk ? 1 : 0) * NM + cg * 32 + l31) * ROWM;
                qmy[i] = QM + ((kk ? 0 : 1) * NM + cg * 32 + l31) * ROWM;
                bmx[i] = PM + ((kk ? 1 : 0) * NM + cg * 32 + l31) * LDBM;
                bmy[i] = PM + ((kk ? 0 : 1) * NM + cg * 32 + l31) * LDBM;
                m2_lane[i] = (int)p.mid2_col[cg * 32 + l31] + (int)p.mid2_row[4 * kk];
            }
            settle(m2_lane[i]);
            m2_rt[i] = __builtin_amdgcn_readfirstlane((int)sload64(p.mid2_row + 32 * rtm));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) m2_o[b] = (int)sload64(p.mid2_row + (b < 2 ? 1 << b : 2 << b));
    }
    auto m2_t = [&](int t) __attribute__((always_inline)) {
        return ((t & 1) ? m2_o[0] : 0) + ((t & 2) ? m2_o[1] : 0) + ((t & 4) ? m2_o[2] : 0) + ((t & 8) ? m2_o[3] : 0);
    };
    // RI2, step 2: this lane's A' plane -- (row parity, k-row) -> Re, -Im, Im, Re -- and B' plane
    const int ri_par = l31 & 1;
    const int ri_plane = (ri_par == kk ? 0 : (ri_par ? 1 : 2)) * LD2;
    const int ri_cg = RI2 ? wave % p.ng2 : 0;            // this wave's column group, all its items
    const int ri_rt0 = RI2 ? wave / p.ng2 : 0;           // ... and its first row tile (then + SW / ng2)
    const int ri_rts = RI2 ? SW / p.ng2 : 0;
    const float* ri_b = P2 + (kk * N2 + ri_cg * 32 + l31) * LDB2;

    // The factor the stores apply, as TWO floats (alpha x alpha2): the powers of two taken out of the small operands
    // add up -- two operands near 2^-70 give 2^-140, not a float -- while the result itself may well be one
    // (the big operand compensates): applied one after the other, each inside the float range, the product of an
    // fp32 value with 2^bf3_ex is exact wherever the fp32 kernel's step-by-step product is (advisor, round 4).
    float alpha = 1.f, alpha2 = 1.f;
    {
        int e1 = bf3_ex < -126 ? -126 : (bf3_ex > 126 ? 126 : bf3_ex);
        int e2 = bf3_ex - e1;
        e2 = e2 < -126 ? -126 : (e2 > 126 ? 126 : e2);   // (beyond 2^+-252: the result is out of range anyway)
        if (p.facA != nullptr) {
            const double f = (*p.facA) * (*p.facB1) * (*p.facB2) * (TRI ? *p.facBM : 1.0);
            alpha = (f == 0.0 && p.check_zero) ? 0.f : (float)(1.0 / f * (BF3 ? exp2((double)e1) : 1.0));
        } else if (BF3 && bf3_ex != 0) {
            alpha = pow2f(e1);
        }
        if (BF3 && e2 != 0) alpha2 = pow2f(e2);
    }
#ifdef CTG_STEM_H2
    // H2: the stores always scale -- by 2^(exponents taken out of A, B1, B2 and, per tile, of the intermediate), as two
    // factors inside the float range (set per tile: h2_set_alpha).  strip_exponent runs do not take this arithmetic
    // (ctg_runtime.hip): alpha carries no other factor.
    const int h2_e0 = BF3 ? bf3_ex + h2_exa : 0;
    auto h2_set_alpha = [&](int et) __attribute__((always_inline)) {
        const int E = h2_e0 + et;
        const int e1 = E < -126 ? -126 : (E > 126 ? 126 : E);
        int e2 = E - e1;
        e2 = e2 < -126 ? -126 : (e2 > 126 ? 126 : e2);
        alpha = pow2f(e1);
        alpha2 = pow2f(e2);
    };
    if constexpr (BF3) h2_set_alpha(0);
    const bool scaled = BF3 ? true : __builtin_amdgcn_readfirstlane(alpha != 1.f || alpha2 != 1.f);
#else
    const bool scaled = __builtin_amdgcn_readfirstlane(alpha != 1.f || alpha2 != 1.f);   // (strip_exponent runs, rescaled operands)
#endif
    __syncthreads();

#ifdef CTG_STEM_KO_BFRAG
    f32x4 ko_b = *(const f32x4*)(P1 + 4 * (lane & 3));
    settle(ko_b);
#endif
    // register-resident B fragments.  Step 1: [half chunk (4 slots)][b1p | b1q]; step 2:
    // [quad][X | Y] -- there the X tile's sign (A' = (Re a, -Im a)) is folded into the register
    // copy: the lanes of the second k-row hold -Im b instead, and the loop feeds A' = (Re a,
    // Im a) to both tiles
    f32x4 b1r[BR1 && !BF3 ? NCH * 2 : 1][2];
    f32x4 b2r[K2Q > 0 && !BF3 ? K2Q : 1][PACK2 ? 1 : 2];
    bf16x8 b1r3[BR1 && BF3 ? NCH : 1][3][2];   // BF3: [chunk][split][b1p | b1q]
    if constexpr (BR1 && BF3) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                b1r3[c][q][0] = *(const bf16x8*)(q1p + c * 48 + q * 8);
                b1r3[c][q][1] = *(const bf16x8*)(q1q + c * 48 + q * 8);
            }
    }
    if constexpr (BR1 && !BF3) {
#pragma unroll
        for (int q = 0; q < NCH * 2; ++q) {
            b1r[q][0] = *(const f32x4*)(b1p + (q >> 1) * 16 + (q & 1) * 4);
            b1r[q][1] = *(const f32x4*)(b1q + (q >> 1) * 16 + (q & 1) * 4);
        }
    }
    if constexpr (K2Q > 0 && RI2) {
        // (one value per k: the lane's k-row's plane of its column)
#pragma unroll
        for (int q = 0; q < K2Q; ++q) b2r[q][0] = *(const f32x4*)(ri_b + 4 * q);
    }
    if constexpr (K2Q > 0 && !BF3 && !RI2) {
        // (one item per wave and tile, always the same: its column group is this wave's)
        const int cg0 = PACK2 ? 0 : wave / (p.rows2 >> 5);
#pragma unroll
        for (int q = 0; q < K2Q; ++q) {
            b2r[q][0] = *(const f32x4*)(b2x + cg0 * 32 * LDB2 + 4 * q);
            if (!PACK2) {
                b2r[q][1] = *(const f32x4*)(b2y + cg0 * 32 * LDB2 + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) b2r[q][0][t] = flip(b2r[q][0][t], sgn2);
            }
        }
    }
    const int nch = STATIC ? NCH : (K1 >> 4);        // 16-deep chunks of the first contraction
    const int n_rt2 = p.rows2 >> 5;
    const int n_items = n_rt2 * p.ng2;
    const int64_t n_tiles = p.n_tiles;
    const int64_t tile0 = blockIdx.x, tile_step = gridDim.x;
    const int64_t my_tiles = (n_tiles - tile0 + tile_step - 1) / tile_step;   // >= 1: grid <= n_tiles
    const int64_t last_tile = tile0 + (my_tiles - 1) * tile_step;

    // ---- gather pipeline: tasks (tile, unit m, chunk) in order, two in flight --------
    // (specialised waves: a producer keeps GD tasks in flight -- CTG_STEM_WS_DEPTH, a power of two)
    // (symmetric static kernels: CTG_STEM_DEPTH = 4 where a tile has an even number of tasks -- experiment builds)
    constexpr int GD = WS ? CTG_STEM_WS_DEPTH : ((NCH > 0 && CTG_STEM_DEPTH == 4 && ((RT1 * NCH) & 1) == 0) ? 4 : 2);
    c64 regs[GD][8];
    int64_t ig = tile0;   // cursor of the next task to issue
    int im = 0, ic = 0;
    // prep: address of the next task to gather -- scalar loads, issued early (behind the
    // last MFMAs of the task before) so that their latency is nobody's problem;
    // fire2: two of the task's eight elements.  always_tag: unconditional (past the last
    // tile the last one is fetched again: the steady state must not contain a conditional
    // memory instruction)
    int64_t pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0;   // (summed where they are used)
    bool pend_live = false;
    auto prep = [&](auto always_tag) __attribute__((always_inline)) {
        constexpr bool ALWAYS = decltype(always_tag)::value;
        pend_live = ALWAYS || ig < n_tiles;
        if (pend_live) {
            const int64_t g = ALWAYS ? (ig < last_tile ? ig : last_tile) : ig;
            const int64_t gh = g >> p.g_lo_shift, gl = g & (p.g_lo - 1);
            pend0 = sload64(p.gA_hi + uniform64(gh));
            pend1 = sload64(p.gA_lo + uniform64(gl));
            pend2 = sload64(p.rt_a + (wrt + RTW * im));
            pend3 = sload64(p.chunk_a + ic);
            if (++ic == nch) {
                ic = 0;
                if (++im == RT1) {
                    im = 0;
                    ig += tile_step;
                }
            }
        }
    };
    // slots 2 q, 2 q + 1 of a task: two 8-byte loads, or one 16-byte load when they are adjacent
    auto fire2 = [&](c64 (&r)[8], int q, int64_t base, auto always_tag) __attribute__((always_inline)) {
        if (decltype(always_tag)::value || pend_live) {
#ifdef CTG_STEM_KO_GATHER
            r[2 * q] = c64{(float)(base + kj[2 * q]), (float)a_lane};
            r[2 * q + 1] = c64{(float)(base + kj[2 * q + 1]), (float)a_lane};
#else
            if constexpr (XM) {
                // the task's base as a SCALAR byte address (kept from being folded into a per-lane 64-bit address
                // that costs a vector add per load): scalar base + 32-bit lane offset is the load's addressing mode
                typedef const __attribute__((address_space(1))) char* gptr;
                uint64_t u0 = (uint64_t)A + ((uint64_t)(base + kj[2 * q]) << 3);
                uint64_t u1 = (uint64_t)A + ((uint64_t)(base + kj[2 * q + 1]) << 3);
                asm volatile("" : "+s"(u0), "+s"(u1));
                settle(a_lane);   // (the zero-extension stays in this block: instruction selection is per block)
                const unsigned al = a_lane;
                if (VEC) {
                    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)((gptr)u0 + al);
                    r[2 * q] = c64{v[0], v[1]};
                    r[2 * q + 1] = c64{v[2], v[3]};
                } else {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 v0 = *(const __attribute__((address_space(1))) f32x2*)((gptr)u0 + al);
                    const f32x2 v1 = *(const __attribute__((address_space(1))) f32x2*)((gptr)u1 + al);
                    r[2 * q] = c64{v0[0], v0[1]};
                    r[2 * q + 1] = c64{v1[0], v1[1]};
                }
                return;
            }
            const char* sb = (const char*)(A + (base + kj[2 * q]));   // uniform: the load's scalar base
#ifdef CTG_STEM_BOUNDS
            {
                const uint64_t lim = (uint64_t)p.a_elems * 8;
                const uint64_t o0 = (uint64_t)(sb + a_lane - (const char*)A);
                const uint64_t o1 = VEC ? o0 + 8 : (uint64_t)((const char*)(A + (base + kj[2 * q + 1])) + a_lane - (const char*)A);
                if (o0 + 8 > lim || o1 + 8 > lim) {
                    atomicAdd(&ctg_stem_oob[0], 1ull);
                    r[2 * q] = r[2 * q + 1] = c64{0.f, 0.f};
                    return;
                }
            }
#endif
            if (VEC) {
                const f32x4 v = *(const f32x4*)(sb + a_lane);
                r[2 * q] = c64{v[0], v[1]};
                r[2 * q + 1] = c64{v[2], v[3]};
            } else {
                const char* sb1 = (const char*)(A + (base + kj[2 * q + 1]));
                const float2 v0 = *(const float2*)(sb + a_lane);
                const float2 v1 = *(const float2*)(sb1 + a_lane);
                r[2 * q] = c64{v0.x, v0.y};
                r[2 * q + 1] = c64{v1.x, v1.y};
            }
#endif
        }
    };
    auto issue = [&](c64 (&r)[8], auto always_tag) __attribute__((always_inline)) {
        prep(always_tag);
        const int64_t base = pend0 + pend1 + pend2 + pend3;
#pragma unroll
        for (int q = 0; q < 4; ++q) fire2(r, q, base, always_tag);
    };

#ifdef CTG_STEM_TIMELINE
    int tl_n = 0;
#endif
    f32x16 ax[RT1], ay[RT1];
    f32x16 axm[XM1 ? RT1 : 1];   // XM: the Im a Im b half of the real parts (X = ax - axm)
    // one task: 16 k of MFMAs on the gathered registers, each register refilled (two tasks
    // ahead) as soon as the MFMAs reading it have been issued
    // Deferred stores: the 16 (8) stores of a work item of step 2 are not issued behind its last
    // MFMA but one or two at a time between the MFMAs of whatever the wave does next (the next
    // item, else the first task of the next tile).  All 8 waves finish their items together, and
    // 128 store instructions of 512 B each in one burst keep the CU's memory pipeline busy for
    // >1000 cycles in which nobody issues an MFMA: knock-out of the stores alone gave 12 % of a
    // slice, of the gathers alone 6 % (profiles/r3_stem_knockout.txt).
    constexpr int NST = PACK2 ? 8 : 16;
    float2 pv[RI2 ? 1 : NST];
    float* pdst = C;
    // RI2: the accumulators of step 2, [set][complex rows 0-15 | 16-31]; the stores of an item read
    // them in place (registers 2 p, 2 p + 1 = Re, Im), so a set is left alone until its stores are
    // out: items alternate between two sets when a wave has more than one per tile
    constexpr int NSET = RI2 ? (IT2 > 1 ? 2 : 1) : 1;
    f32x16 cr[NSET][2];
    auto store2 = [&](float* q, float2 v) __attribute__((always_inline)) {
#ifdef CTG_STEM_BOUNDS
        if ((uint64_t)((char*)q - (char*)C) + 8 > (uint64_t)p.c_elems * 8) {
            atomicAdd(&ctg_stem_oob[1], 1ull);
            return;
        }
#endif
#ifdef CTG_STEM_KO_STORE
        if (v.x == 12345.678f)
#endif
        *(float2*)q = v;
    };
    // stores lo .. hi - 1 of the pending item.  set_tag: which accumulator set holds it (RI2);
    // scaled_tag: a strip_exponent run (RI2 scales at the store; the X / Y form scaled its copy)
    auto drain = [&](int lo, int hi, auto set_tag, auto scaled_tag) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value < NSET ? decltype(set_tag)::value : 0;
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            if constexpr (RI2) {
                // store i: accumulator i >> 3, register pair i & 7 (Re, Im adjacent)
                float2 v;
                v.x = cr[SET][i >> 3][2 * (i & 7)];
                v.y = cr[SET][i >> 3][2 * (i & 7) + 1];
                if constexpr (decltype(scaled_tag)::value) {
                    v.x = v.x * alpha * alpha2;
                    v.y = v.y * alpha * alpha2;
                }
                store2(pdst + 2 * out_t(i), v);
            } else {
                store2(pdst + 2 * out_t(PACK2 ? 2 * i : i), pv[i]);
            }
        }
    };
    auto consume = [&](c64 (&r)[8], int m, int ch, auto always_tag, auto drain_tag, auto scaled_tag)
                       __attribute__((always_inline)) {
        constexpr bool DRAIN = decltype(drain_tag)::value >= 0;   // (-1: nothing pending, else the set)
        const int64_t base = pend0 + pend1 + pend2 + pend3;   // of the task two ahead (prep of the task before)
        if constexpr (BF3) {
            // all 8 elements of the task at once: split, refill the registers, 6 cross terms
            float re[8], im[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                re[j] = r[j].re;
                im[j] = r[j].im;
            }
            bf16x8 r3[3], i3[3], n3[3], bp3[3], bq3[3];
            split3(re, r3, h2_sa);
            split3(im, i3, h2_sa);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) fire2(r, q, base, always_tag);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (!PACK1 && !XM1) n3[q] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, i3[q]) ^ 0x80008000u);
                if constexpr (BR1) {
                    bp3[q] = b1r3[ch][q][0];
                    bq3[q] = b1r3[ch][q][1];
                } else {
                    bp3[q] = *(const bf16x8*)(q1p + ch * 48 + q * 8);
                    bq3[q] = *(const bf16x8*)(q1q + ch * 48 + q * 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (XM: a unit's first task starts its accumulators from a zero C operand -- an inline constant of the
            // instruction -- instead of 48 register clears per tile; ch is a constant of the unrolled tile)
            const bool fresh = XM && STATIC && ch == 0;
            f32x16 zero16;
#pragma unroll
            for (int u = 0; u < 16; ++u) zero16[u] = 0.f;
#pragma unroll
            for (int t = 0; t < 6; t += CTG_STEM_T_STEP(t)) {
                const int ta = bf3_ta(t), tb = bf3_tb(t);
                if (PACK1) {
                    ax[m] = mfma_bf(r3[ta], bp3[tb], (fresh && t == 0) ? zero16 : ax[m]);
                    ax[m] = mfma_bf(i3[ta], bq3[tb], ax[m]);
                } else if constexpr (XM1) {
                    ax[m] = mfma_bf(r3[ta], bp3[tb], (fresh && t == 0) ? zero16 : ax[m]);
                    ay[m] = mfma_bf(r3[ta], bq3[tb], (fresh && t == 0) ? zero16 : ay[m]);
                    axm[m] = mfma_bf(i3[ta], bq3[tb], (fresh && t == 0) ? zero16 : axm[m]);
                    ay[m] = mfma_bf(i3[ta], bp3[tb], ay[m]);
                } else {
                    ax[m] = mfma_bf(r3[ta], bp3[tb], (fresh && t == 0) ? zero16 : ax[m]);
                    ay[m] = mfma_bf(r3[ta], bq3[tb], (fresh && t == 0) ? zero16 : ay[m]);
                    ax[m] = mfma_bf(n3[ta], bq3[tb], ax[m]);
                    ay[m] = mfma_bf(i3[ta], bp3[tb], ay[m]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DRAIN)
                    drain(CTG_STEM_T_INDEX(t) * NST / CTG_STEM_T_COUNT, (CTG_STEM_T_INDEX(t) + 1) * NST / CTG_STEM_T_COUNT, drain_tag, scaled_tag);
            }
            __builtin_amdgcn_sched_barrier(0);
            prep(always_tag);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        f32x4 bp[2], bq[2];
        if constexpr (BR1) {   // (ch is a compile-time constant here: static variants only)
            bp[0] = b1r[ch * 2][0];
            bp[1] = b1r[ch * 2 + 1][0];
            bq[0] = b1r[ch * 2][1];
            bq[1] = b1r[ch * 2 + 1][1];
        } else {
#ifdef CTG_STEM_KO_BFRAG   // (knock-out: B fragments from registers instead of LDS)
            bp[0] = bp[1] = bq[0] = bq[1] = ko_b;
#else
            bp[0] = *(const f32x4*)(b1p + ch * 16);
            bq[0] = *(const f32x4*)(b1q + ch * 16);
            bp[1] = *(const f32x4*)(b1p + ch * 16 + 4);
            bq[1] = *(const f32x4*)(b1q + ch * 16 + 4);
#endif
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float re = r[j].re, im = r[j].im;
            const float p_ = bp[j >> 2][j & 3], q_ = bq[j >> 2][j & 3];
            __builtin_amdgcn_sched_barrier(0);
            if (PACK1) {
                ax[m] = mfma(re, p_, ax[m]);
                ax[m] = mfma(im, q_, ax[m]);
            } else {
                ax[m] = mfma(re, p_, ax[m]);
                ay[m] = mfma(re, q_, ay[m]);
                ax[m] = mfma(flip(im, sgn), q_, ax[m]);
                ay[m] = mfma(im, p_, ay[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j & 1) fire2(r, j >> 1, base, always_tag);
            if constexpr (DRAIN) drain(j * NST / 8, (j + 1) * NST / 8, drain_tag, scaled_tag);
        }
        __builtin_amdgcn_sched_barrier(0);
        prep(always_tag);   // (behind the last MFMAs)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto zero_acc = [&](int m) __attribute__((always_inline)) {
        if constexpr (XM && STATIC) return;   // (the first task of a unit takes a zero C operand)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            ax[m][t] = 0.f;
            if (!PACK1) ay[m][t] = 0.f;
            if constexpr (XM1) axm[m][t] = 0.f;
        }
    };
    // the 32 x (32 | 16) accumulators of every unit -> the shared intermediate tile
    // LM: one value -> its three limbs at dst[0], dst[8], dst[16] (the high halves of x, x - limb 1, and of
    // what is left of that)
    auto put3 = [&](unsigned short* dst, float x) __attribute__((always_inline)) {
        const unsigned u = cvt_pk_bf16(x, 0.f) << 16;
        const float r1 = x - __builtin_bit_cast(float, u);
        const unsigned u1 = cvt_pk_bf16(r1, 0.f) << 16;
        const float r2 = r1 - __builtin_bit_cast(float, u1);
        dst[0] = (unsigned short)(u >> 16);
        dst[8] = (unsigned short)(u1 >> 16);
        dst[16] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    };
#ifdef CTG_STEM_H2
    // H2 + LM (specialised waves): x * (the tile's scale) -> its two rounded fp16 limbs at dst[0], dst[8]
    auto put2 = [&](unsigned short* dst, float x) __attribute__((always_inline)) {
        const float a = x * h2_st;
        const _Float16 h = (_Float16)a;
        const _Float16 l = (_Float16)(a - (float)h);
        dst[0] = __builtin_bit_cast(unsigned short, h);
        dst[8] = __builtin_bit_cast(unsigned short, l);
    };
#endif
    auto scatter = [&]() __attribute__((always_inline)) {
        if constexpr (LM) {
#pragma unroll
            for (int m = 0; m < RT1; ++m) {
                unsigned short* dst = midq + (mid_lane + mid_rt[m]);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
#ifdef CTG_STEM_H2
                    put2(dst + mid_t(t), XM1 ? ax[m][t] - axm[m][t] : ax[m][t]);
                    if (!PACK1) put2(dst + PLS + mid_t(t), ay[m][t]);
#else
                    put3(dst + mid_t(t), XM1 ? ax[m][t] - axm[m][t] : ax[m][t]);
                    if (!PACK1) put3(dst + PLS + mid_t(t), ay[m][t]);
#endif
                }
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < RT1; ++m) {
            float* dst = mid + (mid_lane + mid_rt[m]);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#ifdef CTG_STEM_KO_SCATTER
                if (ax[m][t] != 12345.678f) continue;
#endif
                dst[mid_t(t)] = XM1 ? ax[m][t] - axm[m][t] : ax[m][t];
                if constexpr (RI2) {
                    // third plane: -Im (16 columns: the lanes of columns 16-31 hold the imaginary parts)
                    if constexpr (PACK1) {
                        if (l31 >> 4) dst[LD2 + mid_t(t)] = -ax[m][t];
                    } else {
                        dst[LD2 + mid_t(t)] = ay[m][t];
                        dst[2 * LD2 + mid_t(t)] = -ay[m][t];
                    }
                } else {
                    if (!PACK1) dst[PLANE1 + mid_t(t)] = ay[m][t];
                }
            }
        }
    };
    // one work item of step 2: (32-row tile, 32-column group) of the intermediate x B2
    auto item_row = [&](int item, int64_t c_tile) __attribute__((always_inline)) -> int64_t {
        const int cg = item / n_rt2, rt2 = item - cg * n_rt2;
        int64_t c_row = c_tile + sload64(p.out_row + 32 * rt2);
        asm volatile("" : "+s"(c_row));   // waited for here, not inside the fragment pipeline
        return c_row;
    };
    // drain_tag: the item before this one left its stores pending; defer_tag: leave this one's
    // RI2: one work item = 32 complex rows (row tile rt2) x this wave's 32 columns into accumulator
    // set SET; the stores of the item before (set_prev >= 0) are issued between its MFMAs
    auto item2r = [&](int rt2, int64_t c_row, auto set_tag, auto prev_tag, auto scaled_tag) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool DRAIN = decltype(prev_tag)::value >= 0;
        const float* a0p = mid + ri_plane + (rt2 * 32 + (l31 >> 1)) * RP;
        const float* a1p = a0p + 16 * RP;
        const int64_t c_col = oc_s[ri_cg * 32 + l31];
        f32x4 a0[2], a1[2], bq[2];
        a0[0] = *(const f32x4*)(a0p);
        a1[0] = *(const f32x4*)(a1p);
        constexpr int NQ = K2Q > 0 ? K2Q : 1;
        if constexpr (K2Q > 0) {
            // K2 known, B2 in registers: fully unrolled, first MFMA of each accumulator takes C = 0
            static_for<0, NQ>([&](auto qi) __attribute__((always_inline)) {
                constexpr int q = decltype(qi)::value;
                if (q + 1 < NQ) {
                    a0[(q + 1) & 1] = *(const f32x4*)(a0p + (q + 1) * 4);
                    a1[(q + 1) & 1] = *(const f32x4*)(a1p + (q + 1) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (q == 0 && t == 0) {
                        f32x16 z;
#pragma unroll
                        for (int u = 0; u < 16; ++u) z[u] = 0.f;
                        cr[SET][0] = mfma(a0[0][0], b2r[0][0][0], z);
                        cr[SET][1] = mfma(a1[0][0], b2r[0][0][0], z);
                    } else {
                        cr[SET][0] = mfma(a0[q & 1][t], b2r[q][0][t], cr[SET][0]);
                        cr[SET][1] = mfma(a1[q & 1][t], b2r[q][0][t], cr[SET][1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DRAIN) drain(q * NST / NQ, (q + 1) * NST / NQ, prev_tag, scaled_tag);
            });
        } else {
            // B2 from LDS (K2 = 128, or no registers left): one 16-byte read per 4 k serves both accumulators
            if constexpr (DRAIN) drain(0, NST, prev_tag, scaled_tag);
#pragma unroll
            for (int u = 0; u < 16; ++u) cr[SET][0][u] = cr[SET][1][u] = 0.f;
            const int nq = K2 >> 2;   // >= 4, even
            bq[0] = *(const f32x4*)(ri_b);
            for (int kq = 0; kq < nq; kq += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nx = (kq + h + 1 < nq ? kq + h + 1 : nq - 1) * 4;
                    a0[(h + 1) & 1] = *(const f32x4*)(a0p + nx);
                    a1[(h + 1) & 1] = *(const f32x4*)(a1p + nx);
                    bq[(h + 1) & 1] = *(const f32x4*)(ri_b + nx);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        cr[SET][0] = mfma(a0[h][t], bq[h][t], cr[SET][0]);
                        cr[SET][1] = mfma(a1[h][t], bq[h][t], cr[SET][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        pdst = C + 2 * (c_row + out_lane + c_col);   // (this item's stores are left pending)
    };
    auto item2 = [&](int item, int64_t c_row, auto scaled_tag, auto drain_tag, auto defer_tag)
                     __attribute__((always_inline)) {
        constexpr bool DRAIN = decltype(drain_tag)::value;
        if constexpr (DRAIN && (K2Q == 0 || BF3)) drain(0, NST, std::integral_constant<int, 0>{}, scaled_tag);   // (run-time trip count below: no slots to put them in)
        const int cg = item / n_rt2, rt2 = item - cg * n_rt2;
        f32x16 cx, cy;
        if constexpr (!XM2) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                cx[t] = 0.f;
                if (!PACK2) cy[t] = 0.f;
            }
        }
        const float* a_base = mid + kk * PLANE + (rt2 * 32 + l31) * LD2;
        const float* bxp = b2x + cg * 32 * LDB2;
        const float* byp = PACK2 ? nullptr : b2y + cg * 32 * LDB2;
        // the result's addresses: scalar row base + lane part (LDS copy of the column table)
        const int64_t c_col = oc_s[PACK2 ? (l31 & 15) : cg * 32 + l31];
        const int nq = K2 >> 2;   // >= 4, even
        f32x4 af[2], bx[2], by[2];
        f32x16 cxm;   // XM2: the Im a Im b half of the real parts
        if constexpr (XM2) {
            // 16 k of one component per instruction: the lane halves take the blocks 2 c, 2 c + 1 of the Re
            // and of the Im plane; Xp += Re Re, Y += Re Im, Xm += Im Im, Y += Im Re -- no sign anywhere
            const unsigned short* bR = Q2 + (cg * 32 + l31) * ROW2 + kk * 24;
            const unsigned short* bI = bR + N2 * ROW2;
            const unsigned short* aRq = midq + (rt2 * 32 + l31) * RPS + kk * 48;   // LM: block 2 c + kk
            const float* aRf = mid + (rt2 * 32 + l31) * LD2 + kk * 8;              // fp32 intermediate
            // fragments of chunk c into set F: A' from the limb planes (LM) or split here, B' from its planes
            struct Frag { bf16x8 ar[3], ai[3], br[3], bi[3]; };
            auto load_frag = [&](Frag& F, int c) __attribute__((always_inline)) {
                if constexpr (LM) {
#pragma unroll
                    for (int q = 0; q < CTG_STEM_LIMBS; ++q) {
                        F.ar[q] = *(const bf16x8*)(aRq + c * 96 + q * 8);
                        F.ai[q] = *(const bf16x8*)(aRq + PLS + c * 96 + q * 8);
                    }
                } else {
                    const f32x4 r0 = *(const f32x4*)(aRf + 16 * c), r1 = *(const f32x4*)(aRf + 16 * c + 4);
                    const f32x4 i0 = *(const f32x4*)(aRf + PLANE + 16 * c), i1 = *(const f32x4*)(aRf + PLANE + 16 * c + 4);
                    const float re8[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                    const float im8[8] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3]};
                    split3(re8, F.ar, h2_st);
                    split3(im8, F.ai, h2_st);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    F.br[q] = *(const bf16x8*)(bR + c * 48 + q * 8);
                    F.bi[q] = *(const bf16x8*)(bI + c * 48 + q * 8);
                }
            };
            // 24 MFMAs of one chunk; first_tag: the item's first chunk starts from a zero C operand
            auto mul_frag = [&](const Frag& F, auto first_tag) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                f32x16 zero16;
#pragma unroll
                for (int u = 0; u < 16; ++u) zero16[u] = 0.f;
#pragma unroll
                for (int t = 0; t < 6; t += CTG_STEM_T_STEP(t)) {
                    const int ta = bf3_ta(t), tb = bf3_tb(t);
                    cx = mfma_bf(F.ar[ta], F.br[tb], (FIRST && t == 0) ? zero16 : cx);
                    cy = mfma_bf(F.ar[ta], F.bi[tb], (FIRST && t == 0) ? zero16 : cy);
                    cxm = mfma_bf(F.ai[ta], F.bi[tb], (FIRST && t == 0) ? zero16 : cxm);
                    cy = mfma_bf(F.ai[ta], F.br[tb], cy);
                }
            };
            const int nc = K2 >> 4;   // 1, 2, 4 or 8
            if constexpr (WS) {
                // a consumer has room for two fragment sets: the loads (and splits) of chunk c + 1 are issued
                // before the MFMAs of chunk c
                Frag F0, F1;
                load_frag(F0, 0);
                if (nc == 1) {
                    mul_frag(F0, std::true_type{});
                } else {
                    load_frag(F1, 1);
                    mul_frag(F0, std::true_type{});
                    for (int c = 2; c < nc; c += 2) {
                        load_frag(F0, c);
                        mul_frag(F1, std::false_type{});
                        load_frag(F1, c + 1);
                        mul_frag(F0, std::false_type{});
                    }
                    mul_frag(F1, std::false_type{});
                }
            } else {
                // (one fragment set: B1's fragments, the gathers in flight and three accumulators leave no room for a
                // second one -- the other wave of the SIMD covers the LDS latency)
                Frag F0;
                load_frag(F0, 0);
                mul_frag(F0, std::true_type{});
                for (int c = 1; c < nc; ++c) {
                    load_frag(F0, c);
                    mul_frag(F0, std::false_type{});
                }
            }
        } else if constexpr (BF3 && LM) {
            // 16 columns, limb intermediate: the lane's plane (Re | Im by k-row) of its row, 8 k per instruction
            const unsigned short* aq = midq + kk * PLS + (rt2 * 32 + l31) * RPS;
            for (int kb = 0; kb < (K2 >> 3); ++kb) {
                bf16x8 a3[3], bx3[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q < CTG_STEM_LIMBS) a3[q] = *(const bf16x8*)(aq + kb * 48 + q * 8);
                    bx3[q] = *(const bf16x8*)(q2x + kb * 24 + q * 8);
                }
#pragma unroll
                for (int t = 0; t < 6; t += CTG_STEM_T_STEP(t)) cx = mfma_bf(a3[bf3_ta(t)], bx3[bf3_tb(t)], cx);
            }
        } else if constexpr (BF3) {
            // 8 k per instruction: the row's 8 values of this lane's plane, split; B2 from its planes
            const unsigned short* bxq = q2x + (PACK2 ? 0 : cg * 32 * ROW2);
            const unsigned short* byq = PACK2 ? nullptr : q2y + cg * 32 * ROW2;
            for (int kb = 0; kb < (K2 >> 3); ++kb) {
                const f32x4 lo = *(const f32x4*)(a_base + 8 * kb), hi = *(const f32x4*)(a_base + 8 * kb + 4);
                const float a8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bf16x8 a3[3], ax3[3], bx3[3], by3[3];
                split3(a8, a3, h2_st);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    // (the X tile takes -Im a: the lanes of the second k-row flip the sign)
                    ax3[q] = PACK2 ? a3[q]
                                   : __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a3[q]) ^ (sgn2 | (sgn2 >> 16)));
                    bx3[q] = *(const bf16x8*)(bxq + kb * 24 + q * 8);
                    if (!PACK2) by3[q] = *(const bf16x8*)(byq + kb * 24 + q * 8);
                }
#pragma unroll
                for (int t = 0; t < 6; t += CTG_STEM_T_STEP(t)) {
                    cx = mfma_bf(ax3[bf3_ta(t)], bx3[bf3_tb(t)], cx);
                    if (!PACK2) cy = mfma_bf(a3[bf3_ta(t)], by3[bf3_tb(t)], cy);
                }
            }
        } else {
        af[0] = *(const f32x4*)(a_base);
        if constexpr (K2Q > 0) {
            // B2 fragments in registers, K2 known: the quads fully unrolled
#pragma unroll
            for (int kq = 0; kq < K2Q; ++kq) {
                if (kq + 1 < K2Q) af[(kq + 1) & 1] = *(const f32x4*)(a_base + (kq + 1) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (PACK2) {
                        cx = mfma(af[kq & 1][t], b2r[kq][0][t], cx);
                    } else {
                        cx = mfma(af[kq & 1][t], b2r[kq][0][t], cx);   // (sign in b2r)
                        cy = mfma(af[kq & 1][t], b2r[kq][PACK2 ? 0 : 1][t], cy);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DRAIN && !BF3) drain(kq * NST / K2Q, (kq + 1) * NST / K2Q, std::integral_constant<int, 0>{}, scaled_tag);
            }
        } else {
#ifdef CTG_STEM_KO_BFRAG
        bx[0] = bx[1] = ko_b;
        by[0] = by[1] = ko_b;
#else
        bx[0] = *(const f32x4*)(bxp);
        if (!PACK2) by[0] = *(const f32x4*)(byp);
#endif
        for (int kq = 0; kq < nq; kq += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // next quad (past the end: the last one again -- no branch in the loop)
                const int nx = (kq + h + 1 < nq ? kq + h + 1 : nq - 1) * 4;
                af[(h + 1) & 1] = *(const f32x4*)(a_base + nx);
#ifndef CTG_STEM_KO_BFRAG
                bx[(h + 1) & 1] = *(const f32x4*)(bxp + nx);
                if (!PACK2) by[(h + 1) & 1] = *(const f32x4*)(byp + nx);
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (PACK2) {
                        cx = mfma(af[h][t], bx[h][t], cx);
                    } else {
                        cx = mfma(flip(af[h][t], sgn2), bx[h][t], cx);
                        cy = mfma(af[h][t], by[h][t], cy);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        }
        {
            constexpr bool SC = decltype(scaled_tag)::value;
            pdst = C + 2 * (c_row + out_lane + c_col);
            if (PACK2) {
                // lane c < 16 holds Re of column c, lane c + 16 its Im: lanes below 16
                // store row t, the others row t + 1 of each pair
                const bool hi = (l31 >> 4) != 0;
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    const float mine = hi ? cx[t] : cx[t + 1];   // what the partner needs
                    const float got = __builtin_bit_cast(
                        float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mine), 0x401F));
                    float2 v;
                    v.x = hi ? got : cx[t];
                    v.y = hi ? cx[t + 1] : got;
                    if (SC) {
                        v.x = v.x * alpha * alpha2;
                        v.y = v.y * alpha * alpha2;
                    }
                    if constexpr (BF3) h2_vmax = fmaxf(h2_vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    pv[t >> 1] = v;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    float2 v;
                    const float xr = XM2 ? cx[t] - cxm[t] : cx[t];
                    v.x = SC ? xr * alpha * alpha2 : xr;
                    v.y = SC ? cy[t] * alpha * alpha2 : cy[t];
                    if constexpr (BF3) h2_vmax = fmaxf(h2_vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    pv[t] = v;
                }
            }
            if constexpr (!decltype(defer_tag)::value) drain(0, NST, std::integral_constant<int, 0>{}, scaled_tag);
        }
    };
    // three-step tile: work item i of the middle stage -- (32 rows of the first intermediate) x BM's
    // column group into mx / my (the loops of item2, with the middle step's operand) ...
    f32x16 mx[TRI ? ITM : 1], my[TRI ? ITM : 1];
    auto item_mid = [&](auto ii) __attribute__((always_inline)) {
        constexpr int I = decltype(ii)::value;
        f32x16 cx, cy;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            cx[t] = 0.f;
            cy[t] = 0.f;
        }
        const float* a_base = mid + kk * PLANEM + am_row[I];
        if constexpr (BF3) {
            for (int kb = 0; kb < (KM >> 3); ++kb) {
                const f32x4 lo = *(const f32x4*)(a_base + 8 * kb), hi = *(const f32x4*)(a_base + 8 * kb + 4);
                const float a8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bf16x8 a3[3], ax3[3], bx3[3], by3[3];
                split3(a8, a3);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    ax3[q] = PACKM ? a3[q]
                                   : __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a3[q]) ^ (sgn2 | (sgn2 >> 16)));
                    bx3[q] = *(const bf16x8*)(qmx[I] + kb * 24 + q * 8);
                    if (!PACKM) by3[q] = *(const bf16x8*)(qmy[I] + kb * 24 + q * 8);
                }
#pragma unroll
                for (int t = 0; t < 6; t += CTG_STEM_T_STEP(t)) {
                    cx = mfma_bf(ax3[bf3_ta(t)], bx3[bf3_tb(t)], cx);
                    if (!PACKM) cy = mfma_bf(a3[bf3_ta(t)], by3[bf3_tb(t)], cy);
                }
            }
        } else {
            const int nq = KM >> 2;   // >= 4, even
            f32x4 af[2], bx[2], by[2];
            af[0] = *(const f32x4*)(a_base);
            bx[0] = *(const f32x4*)(bmx[I]);
            if (!PACKM) by[0] = *(const f32x4*)(bmy[I]);
            for (int kq = 0; kq < nq; kq += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nx = (kq + h + 1 < nq ? kq + h + 1 : nq - 1) * 4;
                    af[(h + 1) & 1] = *(const f32x4*)(a_base + nx);
                    bx[(h + 1) & 1] = *(const f32x4*)(bmx[I] + nx);
                    if (!PACKM) by[(h + 1) & 1] = *(const f32x4*)(bmy[I] + nx);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (PACKM) {
                            cx = mfma(af[h][t], bx[h][t], cx);
                        } else {
                            cx = mfma(flip(af[h][t], sgn2), bx[h][t], cx);
                            cy = mfma(af[h][t], by[h][t], cy);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        mx[I] = cx;
        my[I] = cy;
    };
    // ... and its accumulators -> the second intermediate, laid out as the last step's operand
    auto scatter_mid = [&](auto ii) __attribute__((always_inline)) {
        constexpr int I = decltype(ii)::value;
        float* dst = mid + (m2_lane[I] + m2_rt[I]);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            dst[m2_t(t)] = mx[I][t];
            if (!PACKM) dst[PLANE + m2_t(t)] = my[I][t];
        }
    };
    // ONE: the accumulators of unit m become the pending stores (copied: the unit's registers
    // are zeroed for its next tile before the stores are out)
    auto emit_one = [&](int m, int64_t c_tile, auto scaled_tag) __attribute__((always_inline)) {
        constexpr bool SC = decltype(scaled_tag)::value;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float2 v;
            const float xr = XM1 ? ax[m][t] - axm[m][t] : ax[m][t];
            v.x = SC ? xr * alpha * alpha2 : xr;
            v.y = SC ? ay[m][t] * alpha * alpha2 : ay[m][t];
            if constexpr (BF3) h2_vmax = fmaxf(h2_vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
            pv[RI2 ? 0 : t] = v;
        }
        pdst = C + 2 * (c_tile + one_rt[m] + out_lane + one_col);
    };
#ifdef CTG_STEM_H2
    // H2: the intermediate tile's power of two.  publish (step 1 of the tile done, before the barrier): this wave's
    // largest |component| of its accumulators -> LDS (two sets of eight words, alternating by tile: a fast wave's next
    // tile never overwrites what a slow one still reads); consume (after the scatter's barrier): the tile's largest
    // -> the scale step 2 splits with, and the factors its stores apply.
    int h2_par = 0;
    auto h2_publish = [&]() __attribute__((always_inline)) {
        if constexpr (BF3 && !ONE) {
            float mx = 0.f;
#pragma unroll
            for (int m = 0; m < RT1; ++m)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float xr = XM1 ? ax[m][t] - axm[m][t] : ax[m][t];
                    mx = fmaxf(mx, fabsf(xr));
                    if (!PACK1) mx = fmaxf(mx, fabsf(ay[m][t]));
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) bf3_red[h2_par * 8 + wave1] = mx;
            if constexpr (WS && !LM) h2_par ^= 1;   // (specialised waves: only the producers publish, only the consumers consume)
        }
    };
    // (limb intermediate on specialised waves: the PRODUCERS split -- after the barrier that follows their publish they
    // read the tile's largest themselves)
    auto h2_producer_scale = [&]() __attribute__((always_inline)) {
        if constexpr (BF3 && !ONE && WS && LM) {
            float mx = bf3_red[h2_par * 8];
#pragma unroll
            for (int w = 1; w < PW; ++w) mx = fmaxf(mx, bf3_red[h2_par * 8 + w]);
            h2_st = pow2f(-__builtin_amdgcn_readfirstlane(h2_exponent_of(mx)));
            h2_par ^= 1;
        }
    };
    auto h2_consume = [&]() __attribute__((always_inline)) {
        if constexpr (BF3 && !ONE) {
            float mx = bf3_red[h2_par * 8];
#pragma unroll
            for (int w = 1; w < PW; ++w) mx = fmaxf(mx, bf3_red[h2_par * 8 + w]);
            const int et = __builtin_amdgcn_readfirstlane(h2_exponent_of(mx));
            h2_st = pow2f(-et);
            h2_set_alpha(et);
            h2_par ^= 1;
        }
    };
#else
    auto h2_publish = [&]() __attribute__((always_inline)) {};
    auto h2_consume = [&]() __attribute__((always_inline)) {};
    auto h2_producer_scale = [&]() __attribute__((always_inline)) {};
#endif
    auto tile_c = [&](int64_t g) __attribute__((always_inline)) -> int64_t {
        const int64_t gh = g >> p.g_lo_shift, gl = g & (p.g_lo - 1);
        return sload64(p.gC_hi + uniform64(gh)) + sload64(p.gC_lo + uniform64(gl));
    };

    // (the whole tile loop exists twice, with and without the scale factor of a
    // strip_exponent run: a branch around the stores inside the steady state gives the
    // compiler paths with fewer stores than there are, and it waits accordingly)
    auto run = [&](auto scaled_tag) __attribute__((always_inline)) {
    if constexpr (WS) {
        // ---- specialised waves: producers one tile ahead of the consumers ---------------------------------
        constexpr int NT = RT1 * NCH;            // tasks per tile and producer
        constexpr int U = (NT % GD == 0) ? 1 : ((2 * NT) % GD == 0 ? 2 : 4);   // tiles per pass: the gather register sets rotate
        static_assert((U * NT) % GD == 0, "a pass of U tiles returns to gather set 0");
        // step 1 of the producer's next tile: every unit, every chunk (no stores on this side: a wait for a
        // gather counts gathers only)
        auto step1 = [&](auto slot0_tag) __attribute__((always_inline)) {
            constexpr int SLOT0 = decltype(slot0_tag)::value;
            static_for<0, RT1>([&](auto mi) __attribute__((always_inline)) {
                constexpr int M = decltype(mi)::value;
                static_for<0, NCH>([&](auto ci) __attribute__((always_inline)) {
                    constexpr int CH = decltype(ci)::value;
                    consume(regs[(SLOT0 + M * NCH + CH) & (GD - 1)], M, CH, std::true_type{}, std::integral_constant<int, -1>{}, scaled_tag);
                });
            });
        };
        // step 2 of the consumer's tile: its items one after the other; the stores of an item go out at the head
        // of the next one, those of the last item after the next barrier (while the producers scatter)
        auto step2 = [&](int64_t gc) __attribute__((always_inline)) {
            const int64_t c_tile = tile_c(gc);
            int64_t c_rows[IT2];
            static_for<0, IT2>([&](auto ii) __attribute__((always_inline)) {
                c_rows[decltype(ii)::value] = item_row(wave1 + PW * decltype(ii)::value, c_tile);
            });
            static_for<0, IT2>([&](auto ii) __attribute__((always_inline)) {
                constexpr int I = decltype(ii)::value;
                item2(wave1 + PW * I, c_rows[I], scaled_tag, std::integral_constant<bool, (I > 0)>{}, std::true_type{});
            });
        };
        // (one loop per role: what a role keeps in registers across tiles -- the producers' accumulators, gather
        // registers and B1 fragments; the consumers' pending stores -- must not be live in the other's loop.  Both
        // loops pass the same two barriers per tile.)
        if (producer) {
            static_for<0, GD>([&](auto gi) __attribute__((always_inline)) { issue(regs[decltype(gi)::value], std::true_type{}); });
            prep(std::true_type{});
            step1(std::integral_constant<int, 0>{});
            // tiles t, t + 1 (U = 2: the gather register sets swap roles from one tile to the next)
            for (int64_t t = 0; t < my_tiles; t += U) {
                static_for<0, U>([&](auto ui) __attribute__((always_inline)) {
                    constexpr int UI = decltype(ui)::value;
                    if (t + UI < my_tiles) {
                        h2_publish();
                        CTG_STEM_SYNC();   // the consumers have read tile t - 1's intermediate; tile t's accumulators are complete
                        h2_producer_scale();
                        scatter();
                        CTG_STEM_SYNC();
                        if (t + UI + 1 < my_tiles) step1(std::integral_constant<int, ((UI + 1) * NT) & (GD - 1)>{});
                    }
                });
            }
        } else {
            for (int64_t t = 0; t < my_tiles; ++t) {
                CTG_STEM_SYNC();
                if (t > 0) drain(0, NST, std::integral_constant<int, 0>{}, scaled_tag);   // the last item's stores: while the producers scatter
                CTG_STEM_SYNC();
                h2_consume();
                step2(tile0 + t * tile_step);
            }
            drain(0, NST, std::integral_constant<int, 0>{}, scaled_tag);
        }
    } else if constexpr (STATIC) {
        constexpr int NT = RT1 * NCH;            // tasks per tile and wave
        constexpr int U = (NT % GD) ? 2 : 1;     // tiles per pass: the register sets rotate
        static_assert((U * NT) % GD == 0, "a pass of U tiles returns to gather set 0");
        constexpr int LASTSET = RI2 ? ((IT2 > 0 ? IT2 - 1 : 0) & (NSET - 1)) : 0;
        static_for<0, GD>([&](auto gi) __attribute__((always_inline)) { issue(regs[decltype(gi)::value], std::true_type{}); });
        prep(std::true_type{});
        int64_t g = tile0;
        auto tile = [&](auto slot0_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr int SLOT0 = decltype(slot0_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;   // (no item before this tile: nothing pending)
            if constexpr (ONE) {
                const int64_t c_tile = tile_c(g);
                static_for<0, RT1>([&](auto mi) __attribute__((always_inline)) {
                    constexpr int M = decltype(mi)::value;
                    zero_acc(M);
                    static_for<0, NCH>([&](auto ci) __attribute__((always_inline)) {
                        constexpr int CH = decltype(ci)::value;
                        // (the first task of a unit issues the stores of the unit before)
                        consume(regs[(SLOT0 + M * NCH + CH) & (GD - 1)], M, CH, std::true_type{},
                                std::integral_constant<int, (CH == 0 && !(FIRST && M == 0)) ? 0 : -1>{}, scaled_tag);
                    });
                    emit_one(M, c_tile, scaled_tag);
                });
                g += tile_step;
                return;
            }
            CTG_TL_STAMP(0);
            static_for<0, RT1>([&](auto mi) __attribute__((always_inline)) {
                constexpr int M = decltype(mi)::value;
                zero_acc(M);
                static_for<0, NCH>([&](auto ci) __attribute__((always_inline)) {
                    constexpr int CH = decltype(ci)::value;
                    // (the first task of a tile issues the stores the tile before left pending:
                    // those of its last item, accumulator set LASTSET)
                    consume(regs[(SLOT0 + M * NCH + CH) & (GD - 1)], M, CH, std::true_type{},
                            std::integral_constant<int, (!FIRST && M == 0 && CH == 0) ? LASTSET : -1>{}, scaled_tag);
                });
            });
            CTG_TL_STAMP(1);
            h2_publish();
            CTG_STEM_SYNC();   // all waves have finished step 2 of the previous tile
            CTG_TL_STAMP(2);
            scatter();
            CTG_TL_STAMP(3);
            const int64_t c_tile = tile_c(g);
            int64_t c_rows[IT2 > 0 ? IT2 : 1];
            static_for<0, IT2>([&](auto ii) __attribute__((always_inline)) {
                constexpr int I = decltype(ii)::value;
                if constexpr (RI2) {
                    int64_t c_row = c_tile + sload64(p.out_row + 32 * (ri_rt0 + ri_rts * I));
                    asm volatile("" : "+s"(c_row));   // waited for here, not inside the fragment pipeline
                    c_rows[I] = c_row;
                } else {
                    c_rows[I] = item_row(wave + SW * I, c_tile);
                }
            });
            CTG_STEM_SYNC();
            h2_consume();
            CTG_TL_STAMP(4);
            if constexpr (TRI) {
                static_for<0, ITM>([&](auto ii) __attribute__((always_inline)) { item_mid(ii); });
                CTG_STEM_SYNC();   // every wave has read the first intermediate: the second goes over it
                static_for<0, ITM>([&](auto ii) __attribute__((always_inline)) { scatter_mid(ii); });
                CTG_STEM_SYNC();
            }
            static_for<0, IT2>([&](auto ii) __attribute__((always_inline)) {
                constexpr int I = decltype(ii)::value;
                if constexpr (RI2)
                    item2r(ri_rt0 + ri_rts * I, c_rows[I], std::integral_constant<int, I & (NSET - 1)>{},
                           std::integral_constant<int, (I > 0) ? ((I - 1) & (NSET - 1)) : -1>{}, scaled_tag);
                else
                    item2(wave + SW * I, c_rows[I], scaled_tag, std::integral_constant<bool, (I > 0)>{}, std::true_type{});
            });
            CTG_TL_STAMP(5);
            g += tile_step;
        };
        auto pass = [&](auto peel_tag) __attribute__((always_inline)) {
            static_for<0, U>([&](auto ui) __attribute__((always_inline)) {
                tile(std::integral_constant<int, (decltype(ui)::value * NT) & (GD - 1)>{},
                     std::integral_constant<bool, decltype(peel_tag)::value && decltype(ui)::value == 0>{});
            });
        };
        int64_t t = 0;
        if (my_tiles >= U) {
            // (first pass peeled: the waits at the loop header must hold for the entry path
            // as well, where no store has been issued yet -- see the streaming kernel)
            pass(std::true_type{});
            for (t = U; t + U <= my_tiles; t += U) pass(std::false_type{});
        }
        if (t < my_tiles) {   // (U = 2, odd count; t is even)
            if (t == 0) tile(std::integral_constant<int, 0>{}, std::true_type{});
            else tile(std::integral_constant<int, 0>{}, std::false_type{});
        }
        drain(0, NST, std::integral_constant<int, LASTSET>{}, scaled_tag);   // the last item's
    } else {
        issue(regs[0], std::false_type{});
        issue(regs[1], std::false_type{});
        prep(std::false_type{});
        int slot = 0;
        for (int64_t g = tile0; g < n_tiles; g += tile_step) {
            const int64_t c_tile1 = ONE ? tile_c(g) : 0;
#pragma unroll
            for (int m = 0; m < RT1; ++m) {
                zero_acc(m);
                for (int ch = 0; ch < nch; ++ch) {
                    if (slot == 0) consume(regs[0], m, ch, std::false_type{}, std::integral_constant<int, -1>{}, scaled_tag);
                    else consume(regs[1], m, ch, std::false_type{}, std::integral_constant<int, -1>{}, scaled_tag);
                    slot ^= 1;
                }
                if constexpr (ONE) {
                    emit_one(m, c_tile1, scaled_tag);
                    drain(0, NST, std::integral_constant<int, 0>{}, scaled_tag);
                }
            }
            if constexpr (ONE) continue;
            h2_publish();
            CTG_STEM_SYNC();
            scatter();
            CTG_STEM_SYNC();
            h2_consume();
            const int64_t c_tile = tile_c(g);
            for (int item = wave; item < n_items; item += SW)
                item2(item, item_row(item, c_tile), scaled_tag, std::false_type{}, std::false_type{});
        }
    }
    };
    if (scaled) run(std::true_type{});
    else run(std::false_type{});
    // the largest |component| this launch stored: what a consumer of the result scales its split with (fp16 x 2)
    if constexpr (BF3) {
        if (p.cmax != nullptr) {
            float mx = h2_vmax;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0 && mx > 0.f && mx < __builtin_bit_cast(float, 0x7f800000u))
                record_max(p.cmax, mx);
        }
    }
}

#ifdef CTG_STEM_TIMELINE
}  // namespace ctg
// (experiment build only; not in include/ctg_hip.h) the stamps of the last launch: 8 x CTG_TL_TILES x 6 words
extern "C" int ctg_debug_stem_timeline(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ctg::ctg_stem_tl), sizeof(unsigned long long) * 8 * CTG_TL_TILES * 6) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[8 * CTG_TL_TILES * 6];
        if (hipMemcpyToSymbol(HIP_SYMBOL(ctg::ctg_stem_tl), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
namespace ctg {
#endif

#ifdef CTG_STEM_BOUNDS
}  // namespace ctg
// (experiment build only; not in include/ctg_hip.h) out-of-bounds counters: [0] gathers, [1] stores
extern "C" int ctg_debug_stem_oob(unsigned long long out[2], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ctg::ctg_stem_oob), 16) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ctg::ctg_stem_oob), z, 16) != hipSuccess) return -1;
    }
    return 0;
}
namespace ctg {
#endif

#ifdef CTG_STEM_DEV_ONE
// (kernel development: ONE instantiation, compiled in half a minute -- hipcc -DCTG_STEM_DEV_ONE="<template arguments>" -c
// ctg_stem.hip -save-temps; the object is not linkable into the library)
template __global__ void stem2_kernel<CTG_STEM_DEV_ONE>(StemArgs);
#else
size_t stem2_lds_bytes(const StemArgs& p) {
    const size_t b1 = (size_t)(p.N1 == 16 ? 3 : 2) * p.N1 * (p.K1 + 4);
    const size_t b2 = (size_t)(p.N2 == 16 ? 3 : 2) * p.N2 * (p.K2 + 4);
    const size_t mid = (size_t)2 * p.rows2 * p.ld2;
    return 4 * (b1 + b2 + mid) + 8 * (size_t)p.N2;
}

// the same with the small operands as bf16 x 3 planes (BF3)
static size_t stem2_lds_bytes_bf3(const StemArgs& p) {
    const size_t q1 = (size_t)(p.N1 == 16 ? 3 : 2) * p.N1 * ((p.K1 >> 4) * 48 + 8);
    const size_t q2 = (size_t)(p.N2 == 16 ? 3 : 2) * p.N2 * ((p.K2 >> 3) * 24 + 8);
    return 2 * (q1 + q2) + 4 * (size_t)2 * p.rows2 * p.ld2 + 8 * (size_t)p.N2 + 64;   // (+ the reduction scratch)
}

// ... and with the intermediate as bf16 limbs (LM): 6 bytes per value and plane + 16 of padding per row; B1's planes
// lie over it when its fragments live in registers (up to two chunks of K1)
static size_t stem2_lds_bytes_lm(const StemArgs& p) {
    const size_t q1 = 2 * (size_t)(p.N1 == 16 ? 3 : 2) * p.N1 * ((p.K1 >> 4) * 48 + 8);
    const size_t q2 = 2 * (size_t)(p.N2 == 16 ? 3 : 2) * p.N2 * ((p.K2 >> 3) * 24 + 8);
    const size_t mid = (size_t)p.rows2 * ((p.K2 >> 3) * 96 + 16);
    const bool br1 = p.K1 <= 32;
    return q2 + (br1 ? (mid > q1 ? mid : q1) : q1 + mid) + 8 * (size_t)p.N2 + 64 + 8;
}

// ... and of the row-interleaved step 2 (RI2): three planes of the intermediate; B2's staging
// planes share them when its fragments go to registers
static size_t stem2_lds_bytes_ri2(const StemArgs& p, bool b2_in_regs) {
    const size_t b1 = (size_t)(p.N1 == 16 ? 3 : 2) * p.N1 * (p.K1 + 4);
    const size_t b2 = (size_t)2 * p.N2 * (p.K2 + 4);
    const size_t mid = (size_t)3 * p.rows2 * p.ld2;
    return 4 * (b1 + (b2_in_regs ? (mid > b2 ? mid : b2) : b2 + mid)) + 8 * (size_t)p.N2;
}

template <bool PACK1, bool PACK2, int RT1, int CS1, int NCH, int IT2, bool BR1 = false, int K2Q = 0, bool VEC = false,
          bool BF3 = false, bool RI2 = false, bool XM = false, bool LM = false, bool WS = false>
static hipError_t launch_stem2_t(const StemArgs& p_, hipStream_t stream) {
    const StemArgs& p = p_;
    auto kern = stem2_kernel<PACK1, PACK2, RT1, CS1, NCH, IT2, BR1, K2Q, VEC, BF3, RI2, false, 0, false, XM, LM, WS>;
    static unsigned long long ready = 0;   // (bit per device)
    {
        const hipError_t e = lds_opt_in((const void*)kern, 160 * 1024, &ready);
        if (e != hipSuccess) return e;
    }
#ifdef CTG_STEM_TIMELINE
    {
        static int taken = 0;
        int on = 0, k1 = 0, n1 = 0, k2 = 0, n2 = 0;
        if (const char* v = getenv("CTG_TL_SHAPE"))
            if (sscanf(v, "%d,%d,%d,%d", &k1, &n1, &k2, &n2) == 4 && k1 == p.K1 && n1 == p.N1 && k2 == p.K2 && n2 == p.N2 && !taken)
                on = taken = 1;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(ctg::ctg_stem_tl_on), &on, sizeof(int), 0, hipMemcpyHostToDevice, stream);
    }
#endif
    const size_t smem = LM ? stem2_lds_bytes_lm(p)
                           : BF3 ? stem2_lds_bytes_bf3(p) : (RI2 ? stem2_lds_bytes_ri2(p, K2Q > 0) : stem2_lds_bytes(p));
    // persistent: one workgroup per CU (the tile owns most of the CU's LDS)
    int64_t blocks = p.n_tiles < 256 ? p.n_tiles : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1), dim3(SW * 64), smem, stream, p);
    return hipGetLastError();
}

// Form of the bf16 x 3 kernels the library is built with.  1 (the product): two-accumulator real parts, fp32 intermediate.
// The two other round-5 forms were measured next to it on the headline tree and are experiment builds
// (tools/build_variants.py lm=-DCTG_STEM_FORM=2,-DCTG_STEM_LM  ws=-DCTG_STEM_FORM=3,-DCTG_STEM_WS; profiles/r5_forms_*.txt):
// 2 the intermediate as bf16 limbs (40 % fewer vector instructions, 3-4 % SLOWER: the split moves into the scatter
// between the two barriers, where no wave has MFMAs to hide it), 3 specialised waves (same time as form 1 to 1 %).
// The fp16 x 2 object (-DCTG_STEM_H2) is built with form 3 (-DCTG_STEM_FORM=3 -DCTG_STEM_WS, __graft_entry__.py): with
// half the matrix work per tile, the scatter between the barriers is a larger share of it, and producers one tile ahead
// of the consumers hide it -- every pair shape of the headline tree 1-8 % faster, 199 -> 193 ms/slice
// (profiles/r6_forms_h2_xm_vs_ws.txt).
#ifndef CTG_STEM_FORM
#define CTG_STEM_FORM 1
#endif
// the first half alone (ONE): B1's planes and the column table
static size_t stem2_lds_bytes_one(const StemArgs& p, bool bf3) {
    if (bf3) return 2 * (size_t)2 * p.N1 * ((p.K1 >> 4) * 48 + 8) + 8 * (size_t)p.N1 + 64;
    return 4 * (size_t)2 * p.N1 * (p.K1 + 4) + 8 * (size_t)p.N1;
}

template <int RT1, int CS1, int NCH, bool BR1, bool VEC, bool BF3>
static hipError_t launch_stem1_t(const StemArgs& p, hipStream_t stream) {
    // (bf16 x 3, static: the two-accumulator form of the real parts, round 5 -- XM; CTG_STEM_FORM=0 builds keep round 4's)
    constexpr bool XM = BF3 && NCH > 0 && CTG_STEM_FORM >= 1;
    auto kern = stem2_kernel<false, false, RT1, CS1, NCH, 0, BR1, 0, VEC, BF3, false, true, 0, false, XM, false>;
    static unsigned long long ready = 0;   // (bit per device)
    {
        const hipError_t e = lds_opt_in((const void*)kern, 160 * 1024, &ready);
        if (e != hipSuccess) return e;
    }
    // persistent; no LDS to speak of, one workgroup of 8 waves per CU (the register budget is the pair kernel's)
    int64_t blocks = p.n_tiles < 256 ? p.n_tiles : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1), dim3(SW * 64), stem2_lds_bytes_one(p, BF3), stream, p);
    return hipGetLastError();
}

// three-step tile (round 4): the three small operands' planes, the two intermediates in one region
static size_t stem3_lds_bytes(const StemArgs& p, bool bf3) {
    const size_t m1 = (size_t)p.rowsM * p.ldM, m2 = (size_t)p.rows2 * p.ld2;
    const size_t mid = 8 * (m1 > m2 ? m1 : m2);
    auto planes = [](int n) { return (size_t)(n == 16 ? 3 : 2); };
    if (bf3) {
        const size_t q = planes(p.N1) * p.N1 * ((p.K1 >> 4) * 48 + 8) + planes(p.NM) * p.NM * ((p.KM >> 3) * 24 + 8) +
                         planes(p.N2) * p.N2 * ((p.K2 >> 3) * 24 + 8);
        return 2 * q + mid + 8 * (size_t)p.N2 + 64;
    }
    const size_t b = planes(p.N1) * p.N1 * (p.K1 + 4) + planes(p.NM) * p.NM * (p.KM + 4) + planes(p.N2) * p.N2 * (p.K2 + 4);
    return 4 * b + mid + 8 * (size_t)p.N2;
}

template <bool P1, bool PM, bool P2, int RT1, int CS1, int NCH, int ITM, int IT2, bool VEC, bool BF3>
static hipError_t launch_stem3_t(const StemArgs& p, hipStream_t stream) {
    auto kern = stem2_kernel<P1, P2, RT1, CS1, NCH, IT2, (NCH <= 2), 0, VEC, BF3, false, false, ITM, PM>;
    static unsigned long long ready = 0;   // (bit per device)
    {
        const hipError_t e = lds_opt_in((const void*)kern, 160 * 1024, &ready);
        if (e != hipSuccess) return e;
    }
    int64_t blocks = p.n_tiles < 256 ? p.n_tiles : 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1), dim3(SW * 64), stem3_lds_bytes(p, BF3), stream, p);
    return hipGetLastError();
}

// static instantiations of the three-step tiles (16 columns in step 1 / middle / last, units per wave,
// column groups of step 1, chunks of K1, items per wave of the middle and of the last step, 16-byte
// gathers): the shapes the time-to-solution trees (sycamore_m20_w32_r4 / w33_bf3, first seven) and the
// test stems (last five) take when every tile that fits is chosen; there is no run-time-count variant --
// the planner asks ctg_stem_triple_instantiated before it emits one
// ROUND 5: measured slower than pairs on every tree (DESIGN / HISTORY section 8), so the product library is built
// WITHOUT these kernels -- ctg_stem_triple_instantiated answers 0 for every shape and the planner never emits a
// middle stage; an experiment build has them (tools/build_variants.py triples=-DCTG_STEM_TRIPLES_BUILD).
#if !defined(CTG_STEM_TRIPLES_BUILD)
#define CTG_STEM_TRI(X)
#elif defined(CTG_STEM_TRI_DEV)
#define CTG_STEM_TRI(X) \
    X(true, true, true, 2, 1, 1, 2, 2, false) X(false, false, false, 1, 1, 2, 1, 1, false) \
    X(true, false, true, 2, 1, 1, 1, 1, false) X(true, false, false, 1, 1, 1, 1, 1, true)
#else
#define CTG_STEM_TRI(X) \
    X(true, true, false, 2, 1, 1, 2, 1, true) X(true, true, false, 2, 1, 1, 2, 1, false) \
    X(true, false, true, 2, 1, 1, 1, 2, false) X(false, true, true, 1, 1, 2, 2, 2, false) \
    X(false, true, false, 1, 2, 1, 2, 2, false) X(false, false, true, 1, 1, 2, 1, 2, false) \
    X(false, false, true, 1, 1, 4, 1, 1, true) \
    X(true, false, true, 2, 1, 1, 1, 1, false) X(true, false, true, 2, 1, 1, 1, 1, true) \
    X(false, true, true, 1, 4, 2, 2, 2, true) X(false, false, false, 1, 1, 2, 1, 4, false) \
    X(true, false, false, 1, 1, 1, 1, 1, true)
#endif

static bool stem3_instantiated(bool p1, bool pm, bool p2, int rt1, int cs1, int nch, int itm, int it2, bool vec) {
#define CTG_STEM_HAS3(A, M, B, R, CS, NC, IM, IT, V) \
    if (p1 == A && pm == M && p2 == B && rt1 == R && cs1 == CS && nch == NC && itm == IM && it2 == IT && vec == V) return true;
    CTG_STEM_TRI(CTG_STEM_HAS3)
#undef CTG_STEM_HAS3
    return false;
}

struct Stem3Shape { bool p1, pm, p2; int rt1, cs1, nch, itm, it2; bool vec; };
static Stem3Shape stem3_shape(const StemArgs& p) {
    Stem3Shape s;
    s.p1 = p.N1 == 16; s.pm = p.NM == 16; s.p2 = p.N2 == 16;
    s.cs1 = p.N1 >= 32 ? p.N1 / 32 : 1;
    s.rt1 = ((1 << (p.nr1 - 5)) * s.cs1) / SW;
    s.nch = p.K1 / 16;
    const int im = (p.rowsM / 32) * p.ngM, i2 = (p.rows2 / 32) * p.ng2;
    s.itm = im % SW == 0 ? im / SW : 0;
    s.it2 = i2 % SW == 0 ? i2 / SW : 0;
    s.vec = p.vec != 0;
    return s;
}

bool stem3_instantiated_c(bool p1, bool pm, bool p2, int rt1, int cs1, int nch, int itm, int it2, bool vec) {
    return stem3_instantiated(p1, pm, p2, rt1, cs1, nch, itm, it2, vec);
}

bool stem3_supported(const StemArgs& p) {
    auto k_ok = [](int k) { return k == 16 || k == 32 || k == 64 || k == 128; };
    auto n_ok = [](int n) { return n == 16 || n == 32 || n == 64 || n == 128; };
    if (!p.tri || p.one || !k_ok(p.K1) || !k_ok(p.KM) || !k_ok(p.K2) || !n_ok(p.N1) || !n_ok(p.NM) || !n_ok(p.N2)) return false;
    const int cs1 = p.N1 >= 32 ? p.N1 / 32 : 1;
    if (p.nr1 < 5 || p.nr1 > 9) return false;
    const int units = (1 << (p.nr1 - 5)) * cs1;
    if (units != 8 && units != 16) return false;
    if (p.rowsM < 32 || (p.rowsM & 31) || p.ldM != p.KM + 4 || p.rows2 < 32 || (p.rows2 & 31) || p.ld2 != p.K2 + 4) return false;
    if ((int64_t)(1 << p.nr1) * p.N1 != (int64_t)p.rowsM * p.KM || (int64_t)p.rowsM * p.NM != (int64_t)p.rows2 * p.K2) return false;
    if (p.ngM != (p.NM >= 32 ? p.NM / 32 : 1) || p.ng2 != (p.N2 >= 32 ? p.N2 / 32 : 1)) return false;
    const Stem3Shape s = stem3_shape(p);
    if (s.itm < 1 || s.itm > 2 || s.it2 < 1 || s.it2 > 4) return false;
    if (!stem3_instantiated(s.p1, s.pm, s.p2, s.rt1, s.cs1, s.nch, s.itm, s.it2, s.vec)) return false;
    return stem3_lds_bytes(p, true) <= 160 * 1024 && stem3_lds_bytes(p, false) <= 160 * 1024;
}

bool stem2_supported(const StemArgs& p) {
    auto k_ok = [](int k) { return k == 16 || k == 32 || k == 64 || k == 128; };
    if (p.tri) return stem3_supported(p);
    if (p.one) {
        if (!k_ok(p.K1) || (p.N1 != 32 && p.N1 != 64 && p.N1 != 128) || p.K2 != 0 || p.N2 != 0) return false;
        if (p.nr1 < 5 || p.nr1 > 9) return false;
        const int units = (1 << (p.nr1 - 5)) * (p.N1 / 32);
        return (units == 8 || units == 16) && stem2_lds_bytes_one(p, false) <= 160 * 1024;
    }
    if (!k_ok(p.K1) || !k_ok(p.K2)) return false;
    if (p.N1 != 16 && p.N1 != 32 && p.N1 != 64 && p.N1 != 128) return false;
    if (p.N2 != 16 && p.N2 != 32 && p.N2 != 64 && p.N2 != 128) return false;
    {   // units of step 1 = row tiles x column groups: 8 or 16
        const int cs1 = p.N1 >= 32 ? p.N1 / 32 : 1;
        if (p.nr1 < 5 || p.nr1 > 9) return false;
        const int units = (1 << (p.nr1 - 5)) * cs1;
        if (units != 8 && units != 16) return false;
    }
    if (p.rows2 < 32 || (p.rows2 & 31) || p.ld2 != p.K2 + 4) return false;
    if ((int64_t)(1 << p.nr1) * p.N1 != (int64_t)p.rows2 * p.K2) return false;
    if (p.ng2 != (p.N2 >= 32 ? p.N2 / 32 : 1)) return false;
    return stem2_lds_bytes(p) <= 160 * 1024;
}

// static instantiations of the pairs the Sycamore m20 trees are made of (tools/stem_shapes.py
// prints the lists from the tree fixtures, planned in both arithmetics, with the rules of stem2_shape below); anything else
// runs on the run-time-count variant.
//   CTG_STEM_INST: fp32 products -- (16 columns first, 16 columns last, units per wave, column
//   groups of step 1, chunks of K1, items per wave, B1 in registers, K2 / 4 if B2 is (else 0),
//   16-byte gathers, step 2 row-interleaved)
//   CTG_STEM_GEO: the geometries (16 columns first, last, units per wave, column groups, chunks,
//   items per wave, 16-byte gathers) -- the bf16 x 3 instantiations (B1 in registers up to two
//   chunks, B2 from LDS, X / Y form)
#ifdef CTG_STEM_DEV_MIN   // (development builds: one instantiation of each kind, a minute to compile)
#define CTG_STEM_INST(X) X(false, false, 1, 1, 2, 1, true, 8, false, true)
#define CTG_STEM_GEO(G) G(false, false, 1, 1, 2, 1, false)
#else
#define CTG_STEM_INST(X) \
    X(false, false, 1, 1, 2, 1, true, 8, false, true) X(false, true, 1, 1, 2, 2, true, 4, false, false) \
    X(false, false, 1, 2, 4, 1, true, 8, false, true) X(false, false, 1, 1, 2, 1, true, 16, false, true) \
    X(true, true, 2, 1, 1, 2, true, 4, false, false) X(false, true, 1, 1, 2, 2, true, 4, true, false) \
    X(true, false, 2, 1, 1, 1, true, 8, false, true) X(false, false, 1, 2, 2, 1, true, 8, false, true) \
    X(false, false, 1, 2, 4, 1, true, 8, true, true) X(false, false, 1, 1, 8, 1, false, 8, false, true) \
    X(true, true, 2, 1, 1, 2, true, 4, true, false) X(true, false, 2, 1, 1, 1, true, 8, true, true) \
    X(false, false, 1, 2, 4, 1, true, 0, false, false) X(false, false, 1, 1, 2, 2, true, 8, false, true) \
    X(false, true, 1, 1, 1, 2, true, 4, false, false) X(false, false, 1, 2, 2, 1, true, 16, false, true) \
    X(false, true, 1, 2, 2, 2, true, 4, false, false) X(true, false, 2, 1, 1, 2, true, 8, true, true) \
    X(false, true, 1, 2, 4, 2, true, 4, false, false) X(false, false, 1, 2, 4, 2, true, 0, false, false) \
    X(false, false, 1, 1, 4, 1, true, 8, true, true) X(false, false, 1, 2, 1, 1, true, 8, false, true) \
    X(false, false, 1, 1, 8, 2, false, 8, false, true) X(true, false, 2, 1, 1, 2, true, 8, false, true) \
    X(false, false, 1, 2, 2, 4, true, 4, true, true) X(false, false, 1, 1, 2, 1, true, 8, true, true) \
    X(false, false, 1, 1, 2, 2, true, 0, false, false) X(true, false, 2, 1, 1, 2, true, 16, false, true) \
    X(false, false, 2, 1, 1, 2, true, 0, false, false) X(false, false, 1, 1, 1, 2, true, 8, false, true) \
    X(false, false, 1, 1, 4, 1, true, 0, false, true) X(false, true, 1, 2, 2, 1, true, 8, false, false) \
    X(false, true, 1, 1, 2, 1, true, 8, false, false) X(false, true, 1, 4, 4, 2, true, 4, true, false) \
    X(false, true, 1, 1, 8, 2, false, 4, true, false) X(true, false, 2, 1, 2, 1, true, 8, false, true) \
    X(true, false, 2, 1, 1, 2, true, 4, false, true) X(true, false, 2, 1, 1, 1, true, 16, false, true) \
    X(false, true, 1, 1, 8, 2, false, 4, false, false) X(false, false, 1, 1, 2, 4, true, 4, false, true) \
    X(false, false, 1, 1, 2, 4, true, 8, false, true) X(false, false, 1, 1, 2, 2, true, 4, false, true) \
    X(true, true, 2, 1, 4, 1, true, 8, false, false) X(false, false, 1, 2, 2, 2, true, 8, true, true) \
    X(false, true, 1, 2, 1, 2, true, 4, false, false) X(true, false, 2, 1, 4, 1, true, 8, false, true) \
    X(false, false, 1, 2, 2, 4, true, 4, false, true)

#define CTG_STEM_GEO(G) \
    G(false, false, 1, 1, 1, 2, false) G(false, false, 1, 1, 2, 1, false) G(false, false, 1, 1, 2, 1, true) \
    G(false, false, 1, 1, 2, 2, false) G(false, false, 1, 1, 2, 4, false) G(false, false, 1, 1, 4, 1, false) \
    G(false, false, 1, 1, 4, 1, true) G(false, false, 1, 1, 8, 1, false) G(false, false, 1, 1, 8, 2, false) \
    G(false, false, 1, 2, 1, 1, false) G(false, false, 1, 2, 2, 1, false) G(false, false, 1, 2, 2, 4, true) \
    G(false, false, 1, 2, 4, 1, false) G(false, false, 1, 2, 4, 1, true) G(false, false, 1, 2, 4, 2, false) \
    G(false, false, 2, 1, 1, 2, false) G(false, true, 1, 1, 1, 2, false) G(false, true, 1, 1, 2, 1, false) \
    G(false, true, 1, 1, 2, 2, false) G(false, true, 1, 1, 2, 2, true) G(false, true, 1, 1, 8, 2, false) \
    G(false, true, 1, 1, 8, 2, true) G(false, true, 1, 2, 2, 1, false) G(false, true, 1, 2, 2, 2, false) \
    G(false, true, 1, 2, 4, 2, false) G(false, true, 1, 4, 4, 2, true) G(true, false, 2, 1, 1, 1, false) \
    G(true, false, 2, 1, 1, 1, true) G(true, false, 2, 1, 1, 2, false) G(true, false, 2, 1, 1, 2, true) \
    G(true, false, 2, 1, 2, 1, false) G(true, true, 2, 1, 1, 2, false) G(true, true, 2, 1, 1, 2, true) \
    G(true, true, 2, 1, 4, 1, false) G(false, false, 1, 2, 2, 2, true) G(false, true, 1, 2, 1, 2, false) \
    G(true, false, 2, 1, 4, 1, false) G(false, false, 1, 2, 2, 4, false)
#endif

namespace {
struct StemShape {
    bool p1, p2;
    int rt1, cs1, nch, it2;   // it2 = 0: the item count is not a multiple of the waves
    bool br1;
    int k2q;
    bool vec;
    bool ri2;                 // step 2 in the row-interleaved form
};
// Which small operand's fragments go to registers.  X / Y form of step 2: B1 needs K1 floats per
// lane, K1 <= 64; B2 2 K2 (K2 with 16 columns) and one column group per wave (always with 16
// columns, else one item per wave) and K2 <= 32 (64); together at most 96 -- B1 first.
// Row-interleaved form (>= 32 columns in step 2, static item count, column groups dividing the
// waves): B2 needs K2 floats, K2 <= 64, whatever the item count; the budget is what a wave's 256
// registers leave after the accumulators (step 1: 32 -- 16 with 16 columns -- per unit; step 2: 32,
// or 64 when the items alternate between two sets), the 32 gather registers and ~40 of addresses
// and fragments in flight -- B1 first; and the three planes of the intermediate must fit the LDS
// (else the X / Y form).
StemShape stem2_shape(const StemArgs& p, bool bf3 = false) {
    StemShape s;
    s.p1 = p.N1 == 16;
    s.p2 = p.N2 == 16;
    s.cs1 = p.N1 >= 32 ? p.N1 / 32 : 1;
    s.rt1 = ((1 << (p.nr1 - 5)) * s.cs1) / SW;
    s.nch = p.K1 / 16;
    const int items = (p.rows2 / 32) * p.ng2;
    s.it2 = items % SW == 0 ? items / SW : 0;
    s.vec = p.vec != 0;
    s.ri2 = !bf3 && !s.p2 && s.it2 > 0 && p.ng2 >= 1 && p.ng2 <= SW && SW % p.ng2 == 0 && !env_on("CTG_STEM_NO_RI2");
    if (s.ri2) {
        const int fixed = s.rt1 * (s.p1 ? 16 : 32) + (s.it2 > 1 ? 64 : 32) + 32 + 40;
        int r1 = p.K1 <= 64 ? p.K1 : 0;
        int r2 = p.K2 <= 64 ? p.K2 : 0;
        if (fixed + r1 + r2 > 256) r2 = 0;
        if (fixed + r1 > 256) r1 = 0;
        if (stem2_lds_bytes_ri2(p, r2 != 0) <= 160 * 1024) {
            s.br1 = r1 != 0;
            s.k2q = r2 ? p.K2 / 4 : 0;
            return s;
        }
        s.ri2 = false;
    }
    int r1 = p.K1 <= 64 ? p.K1 : 0;
    int r2 = ((s.p2 && p.K2 <= 64) || (!s.p2 && s.it2 == 1 && p.K2 <= 32)) ? (s.p2 ? p.K2 : 2 * p.K2) : 0;
    if (r1 && r2 && r1 + r2 > 96) r2 = 0;
    s.br1 = r1 != 0;
    s.k2q = r2 ? p.K2 / 4 : 0;
    return s;
}
}  // namespace

// single steps (ONE): (units per wave, column groups, chunks of K1, 16-byte gathers) of the m20 trees
// (B1 in registers up to K1 = 64; bf16 x 3: up to two chunks); anything else: run-time counts, fp32
#ifdef CTG_STEM_DEV_MIN
#define CTG_STEM_ONE(X) X(1, 1, 2, false)
#else
#define CTG_STEM_ONE(X) \
    X(1, 4, 8, false) X(1, 1, 2, false) X(1, 2, 4, false) X(1, 1, 8, false) \
    X(1, 1, 4, false) X(1, 4, 4, false) X(1, 1, 2, true) X(1, 2, 2, true) X(2, 1, 1, false) X(1, 2, 8, false)
#endif

static bool stem1_static(const StemShape& s) {
    if (env_on("CTG_STEM_GENERIC")) return false;
#define CTG_STEM_HAS1(R, CS, NC, V) \
    if (s.rt1 == R && s.cs1 == CS && s.nch == NC && s.vec == V) return true;
    CTG_STEM_ONE(CTG_STEM_HAS1)
#undef CTG_STEM_HAS1
    return false;
}

// 1: static (counts known at compile time, fragments in registers where they fit), 0: run-time counts
int stem2_variant(const StemArgs& p) {
    if (p.one) return stem1_static(stem2_shape(p, true)) ? 1 : 0;
    const StemShape s = stem2_shape(p);
    if (env_on("CTG_STEM_GENERIC") || s.it2 == 0) return 0;
#define CTG_STEM_HAS(P1, P2, R, CS, NC, IT, B1, KQ, V, RI)                                             \
    if (s.p1 == P1 && s.p2 == P2 && s.rt1 == R && s.cs1 == CS && s.nch == NC && s.it2 == IT && s.br1 == B1 && \
        s.k2q == KQ && s.vec == V && s.ri2 == RI)                                                      \
        return 1;
    CTG_STEM_INST(CTG_STEM_HAS)
#undef CTG_STEM_HAS
    return 0;
}

// does the geometry have a bf16 x 3 instantiation?
static bool stem2_has_geo(const StemShape& s) {
    if (env_on("CTG_STEM_GENERIC") || s.it2 == 0) return false;
#define CTG_STEM_HASG(P1, P2, R, CS, NC, IT, V)                                                         \
    if (s.p1 == P1 && s.p2 == P2 && s.rt1 == R && s.cs1 == CS && s.nch == NC && s.it2 == IT && s.vec == V) return true;
    CTG_STEM_GEO(CTG_STEM_HASG)
#undef CTG_STEM_HASG
    return false;
}

// Arithmetic of a fused pair.  Default (round 4): bf16 x 3 -- static shapes run both steps on the
// bf16 matrix cores with three-way split operands (stem2_kernel<..., BF3 = true>); the executor's
// option ctg_exec_set_stem_arithmetic(exec, 0) selects fp32 products on the fp32 matrix cores; the
// environment variable CTG_STEM_BF16X3, when SET, overrides both ("0" / "" = fp32, anything else =
// bf16 x 3) and is read at every launch (tests switch it within a process).
static bool stem3_bf3(const StemArgs& p) {   // (three-step tiles: every listed shape exists in both arithmetics)
    const char* v = getenv("CTG_STEM_BF16X3");
    return v != nullptr ? !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0')) : p.bf3 != 0;
}

static bool stem2_bf3(const StemArgs& p) {
    const char* v = getenv("CTG_STEM_BF16X3");
    const bool want = v != nullptr ? !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0')) : p.bf3 != 0;
    if (p.one) return want && stem1_static(stem2_shape(p, true)) && stem2_lds_bytes_one(p, true) <= 160 * 1024;
    return want && stem2_has_geo(stem2_shape(p, true)) && (p.K2 & 7) == 0 && stem2_lds_bytes_bf3(p) <= 160 * 1024;
}

// Form of a bf16 x 3 pair (round 5): 1 = two-accumulator real parts, fp32 intermediate split by step 2 (XM: the
// product); 0 = the round-4 form (sign flips on limbs: pairs with four items per wave keep it, all others only in
// experiment builds); 2 / 3 = limb intermediate / specialised waves (experiment builds).  CTG_STEM_FORM in the
// environment lowers the form a build offers.
static int stem2_bf3_form(const StemArgs& p) {
    int form = CTG_STEM_FORM;
    if (const char* v = getenv("CTG_STEM_FORM")) form = atoi(v) < form ? atoi(v) : form;
#if !(CTG_STEM_FORM == 0 || defined(CTG_STEM_FORM_ALL))
    if (form < 1) form = 1;   // (the round-4 form of these shapes exists in experiment builds only)
#endif
#if !defined(CTG_STEM_WSLM) || !defined(CTG_STEM_WS)
    if (form >= 4) form = 3;  // (limb intermediate on specialised waves: -DCTG_STEM_WS -DCTG_STEM_WSLM)
#endif
#ifndef CTG_STEM_LM
    if (form == 2) form = 1;  // (the limb intermediate: experiment builds, -DCTG_STEM_LM)
#endif
#ifndef CTG_STEM_WS
    if (form == 3) form = 1;  // (specialised waves: experiment builds, -DCTG_STEM_WS)
#endif
    // form 4: where step 2 has two or more column groups (every consumer of a row re-splits it otherwise), one item per
    // consumer wave pair, and the limb planes fit
    if (form == 4 && !(p.ng2 >= 2 && p.N2 >= 32 && stem2_lds_bytes_lm(p) <= 160 * 1024)) form = 3;
    if (form == 2 && stem2_lds_bytes_lm(p) > 160 * 1024) form = 1;
    const int items = (p.rows2 / 32) * p.ng2, units = (1 << (p.nr1 - 5)) * (p.N1 >= 32 ? p.N1 / 32 : 1);
    // specialised waves: a producer takes two of the symmetric kernel's shares of step 1, a consumer two of step 2 --
    // at most two items per wave there, and one unit per wave unless step 1 has 16 columns (one accumulator per unit)
    if (form == 4 && !(items <= SW && (p.N1 == 16 || units == SW))) form = 3;
    if (form == 3 && !(items <= 2 * SW && (p.N1 == 16 || units == SW))) form = 1;
    // (... and a consumer with four items of 32 columns keeps three accumulator pairs next to its pending stores: the
    // compiler spills 46-51 registers there -- the symmetric kernel)
    if (form == 3 && items == 2 * SW && p.N2 >= 32) form = 1;
    // four items of step 2 per wave and tile: the pending stores of one item, three accumulators and the fragments
    // of the next do not fit the registers next to B1's fragments (the compiler spills 12-46 of them): round-4 form
    if (items >= 4 * SW) form = 0;
    return form < 0 ? 0 : form;
}

// the instantiation a step runs on, spelled like its symbol in a kernel trace
void stem2_kernel_name(const StemArgs& p, char* buf, size_t n) {
    const StemShape s = stem2_shape(p);
    auto tf = [](bool b) { return b ? "true" : "false"; };
    if (p.tri) {
        const Stem3Shape t = stem3_shape(p);
        snprintf(buf, n, CTG_STEM_KNAME "<%s,%s,%d,%d,%d,%d,%s,0,%s,%s,false,false,%d,%s>", tf(t.p1), tf(t.p2), t.rt1, t.cs1,
                 t.nch, t.it2, tf(t.nch <= CTG_STEM_BR1_MAX), tf(t.vec), tf(stem3_bf3(p)), t.itm, tf(t.pm));
        return;
    }
    if (p.one) {
        const bool st = stem1_static(s), b3 = stem2_bf3(p);
        if (st && b3 && CTG_STEM_FORM >= 1)
            snprintf(buf, n, CTG_STEM_KNAME "<false,false,%d,%d,%d,0,%s,0,%s,true,false,true,0,false,true,false>", s.rt1, s.cs1,
                     s.nch, tf(s.nch <= CTG_STEM_BR1_MAX), tf(s.vec));
        else
            snprintf(buf, n, CTG_STEM_KNAME "<false,false,%d,%d,%d,0,%s,0,%s,%s,false,true>", s.rt1, s.cs1, st ? s.nch : 0,
                     tf(st && (b3 ? s.nch <= CTG_STEM_BR1_MAX : p.K1 <= 64)), tf(s.vec), tf(b3));
        return;
    }
    if (stem2_bf3(p)) {
        const int form = stem2_bf3_form(p);
        if (form == 0)
            snprintf(buf, n, CTG_STEM_KNAME "<%s,%s,%d,%d,%d,%d,%s,0,%s,true,false,false>", tf(s.p1), tf(s.p2), s.rt1, s.cs1,
                     s.nch, s.it2, tf(s.nch <= CTG_STEM_BR1_MAX), tf(s.vec));
        else
            snprintf(buf, n, CTG_STEM_KNAME "<%s,%s,%d,%d,%d,%d,%s,0,%s,true,false,false,0,false,true,%s,%s>", tf(s.p1), tf(s.p2),
                     s.rt1, s.cs1, s.nch, s.it2, tf(s.nch <= CTG_STEM_BR1_MAX), tf(s.vec), tf(form == 2 || form == 4), tf(form >= 3));
    }
    else if (stem2_variant(p))
        snprintf(buf, n, CTG_STEM_KNAME "<%s,%s,%d,%d,%d,%d,%s,%d,%s,false,%s,false>", tf(s.p1), tf(s.p2), s.rt1, s.cs1, s.nch,
                 s.it2, tf(s.br1), s.k2q, tf(s.vec), tf(s.ri2));
    else
        snprintf(buf, n, CTG_STEM_KNAME "<%s,%s,%d,%d,0,0,false,0,%s,false,false,false>", tf(s.p1), tf(s.p2), s.rt1, s.cs1,
                 tf(s.vec));
}

// does this launch run an instantiation of the 16-bit matrix cores (which records the largest element of its result,
// StemArgs::cmax)?  In the object built with -DCTG_STEM_H2: in the fp16 x 2 arithmetic.
bool stem2_uses_bf3(const StemArgs& p) { return !p.tri && stem2_supported(p) && stem2_bf3(p); }

hipError_t launch_stem2(const StemArgs& p, hipStream_t stream) {
    if (!stem2_supported(p)) return hipErrorInvalidValue;
    if (p.tri) {
        const Stem3Shape t = stem3_shape(p);
        const bool b3 = stem3_bf3(p);
#define CTG_STEM_GO3T(A, M, B, R, CS, NC, IM, IT, V)                                                                  \
    if (t.p1 == A && t.pm == M && t.p2 == B && t.rt1 == R && t.cs1 == CS && t.nch == NC && t.itm == IM && t.it2 == IT && \
        t.vec == V)                                                                                                   \
        return b3 ? launch_stem3_t<A, M, B, R, CS, NC, IM, IT, V, true>(p, stream)                                    \
                  : launch_stem3_t<A, M, B, R, CS, NC, IM, IT, V, false>(p, stream);
        CTG_STEM_TRI(CTG_STEM_GO3T)
#undef CTG_STEM_GO3T
        return hipErrorInvalidValue;
    }
    if (p.one) {
        const StemShape s = stem2_shape(p, true);
        const bool b3 = stem2_bf3(p);
        if (stem1_static(s)) {
#define CTG_STEM_GO1(R, CS, NC, V)                                                           \
    if (s.rt1 == R && s.cs1 == CS && s.nch == NC && s.vec == V)                              \
        return b3 ? launch_stem1_t<R, CS, NC, (NC <= CTG_STEM_BR1_MAX), V, true>(p, stream)                 \
                  : launch_stem1_t<R, CS, NC, (NC <= 4), V, false>(p, stream);
            CTG_STEM_ONE(CTG_STEM_GO1)
#undef CTG_STEM_GO1
        }
#define CTG_STEM_CASE1(R, CS)                                                               \
    if (s.rt1 == R && s.cs1 == CS)                                                          \
        return s.vec ? launch_stem1_t<R, CS, 0, false, true, false>(p, stream)              \
                     : launch_stem1_t<R, CS, 0, false, false, false>(p, stream);
        CTG_STEM_CASE1(1, 1) CTG_STEM_CASE1(2, 1) CTG_STEM_CASE1(1, 2) CTG_STEM_CASE1(2, 2)
        CTG_STEM_CASE1(1, 4) CTG_STEM_CASE1(2, 4)
#undef CTG_STEM_CASE1
        return hipErrorInvalidValue;
    }
    if (stem2_bf3(p)) {
        const StemShape s = stem2_shape(p, true);
        const int form = stem2_bf3_form(p);
#if CTG_STEM_FORM >= 3 && defined(CTG_STEM_WS)
#define CTG_STEM_GO3_WS(P1, P2, R, CS, NC, IT, V) \
        if constexpr (IT <= 2 && (P1 || R == 1) && (P2 || IT < 2)) { if (form == 3) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true, false, true, false, true>(p, stream); }
#else
#define CTG_STEM_GO3_WS(P1, P2, R, CS, NC, IT, V)
#endif
#if CTG_STEM_FORM >= 4 && defined(CTG_STEM_WS) && defined(CTG_STEM_WSLM)
#define CTG_STEM_GO3_WSLM(P1, P2, R, CS, NC, IT, V) \
        if constexpr (IT <= 2 && (P1 || R == 1) && !P2 && IT < 2) { if (form == 4) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true, false, true, true, true>(p, stream); }
#else
#define CTG_STEM_GO3_WSLM(P1, P2, R, CS, NC, IT, V)
#endif
#if CTG_STEM_FORM >= 2 && defined(CTG_STEM_LM)
#define CTG_STEM_GO3_LM(P1, P2, R, CS, NC, IT, V) \
        if constexpr (IT < 4) { if (form == 2) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true, false, true, true>(p, stream); }
#else
#define CTG_STEM_GO3_LM(P1, P2, R, CS, NC, IT, V)
#endif
#if CTG_STEM_FORM >= 1
#define CTG_STEM_GO3_XM(P1, P2, R, CS, NC, IT, V) \
        if constexpr (IT < 4) { if (form == 1) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true, false, true, false>(p, stream); }
#else
#define CTG_STEM_GO3_XM(P1, P2, R, CS, NC, IT, V)
#endif
#if CTG_STEM_FORM == 0 || defined(CTG_STEM_FORM_ALL)
#define CTG_STEM_GO3_R4(P1, P2, R, CS, NC, IT, V) \
        if (form == 0) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true>(p, stream);
#else
#define CTG_STEM_GO3_R4(P1, P2, R, CS, NC, IT, V) \
        if constexpr (IT >= 4) { if (form == 0) return launch_stem2_t<P1, P2, R, CS, NC, IT, (NC <= CTG_STEM_BR1_MAX), 0, V, true>(p, stream); }
#endif
#define CTG_STEM_GO3(P1, P2, R, CS, NC, IT, V)                                                          \
    if (s.p1 == P1 && s.p2 == P2 && s.rt1 == R && s.cs1 == CS && s.nch == NC && s.it2 == IT && s.vec == V) { \
        CTG_STEM_GO3_WSLM(P1, P2, R, CS, NC, IT, V)                                                     \
        CTG_STEM_GO3_WS(P1, P2, R, CS, NC, IT, V)                                                       \
        CTG_STEM_GO3_LM(P1, P2, R, CS, NC, IT, V)                                                       \
        CTG_STEM_GO3_XM(P1, P2, R, CS, NC, IT, V)                                                       \
        CTG_STEM_GO3_R4(P1, P2, R, CS, NC, IT, V)                                                       \
        return hipErrorInvalidValue;                                                                    \
    }
        CTG_STEM_GEO(CTG_STEM_GO3)
#undef CTG_STEM_GO3
    }
    const StemShape s = stem2_shape(p);
    if (stem2_variant(p)) {
#define CTG_STEM_GO(P1, P2, R, CS, NC, IT, B1, KQ, V, RI)                                              \
    if (s.p1 == P1 && s.p2 == P2 && s.rt1 == R && s.cs1 == CS && s.nch == NC && s.it2 == IT && s.br1 == B1 && \
        s.k2q == KQ && s.vec == V && s.ri2 == RI)                                                      \
        return launch_stem2_t<P1, P2, R, CS, NC, IT, B1, KQ, V, false, RI>(p, stream);
        CTG_STEM_INST(CTG_STEM_GO)
#undef CTG_STEM_GO
    }
#define CTG_STEM_CASE(P1, P2, R, CS)                                                                    \
    if (s.p1 == P1 && s.p2 == P2 && s.rt1 == R && s.cs1 == CS)                                          \
        return s.vec ? launch_stem2_t<P1, P2, R, CS, 0, 0, false, 0, true>(p, stream)                   \
                     : launch_stem2_t<P1, P2, R, CS, 0, 0, false, 0, false>(p, stream);
#define CTG_STEM_CASES(P2)             \
    CTG_STEM_CASE(true, P2, 1, 1)      \
    CTG_STEM_CASE(true, P2, 2, 1)      \
    CTG_STEM_CASE(false, P2, 1, 1)     \
    CTG_STEM_CASE(false, P2, 2, 1)     \
    CTG_STEM_CASE(false, P2, 1, 2)     \
    CTG_STEM_CASE(false, P2, 2, 2)     \
    CTG_STEM_CASE(false, P2, 1, 4)     \
    CTG_STEM_CASE(false, P2, 2, 4)
    CTG_STEM_CASES(false)
    CTG_STEM_CASES(true)
#undef CTG_STEM_CASES
#undef CTG_STEM_CASE
    return hipErrorInvalidValue;
}

#endif   // CTG_STEM_DEV_ONE
}  // namespace ctg

#if !defined(CTG_STEM_DEV_ONE) && !defined(CTG_STEM_H2)
// (include/ctg_hip.h) is there a three-step tile kernel for this shape?  A pure function of the shape.
extern "C" int ctg_stem_triple_instantiated(int p1, int pm, int p2, int rt1, int cs1, int nch, int itm, int it2, int vec) {
    return ctg::stem3_instantiated_c(p1 != 0, pm != 0, p2 != 0, rt1, cs1, nch, itm, it2, vec != 0) ? 1 : 0;
}
#endif   // CTG_STEM_DEV_ONE
