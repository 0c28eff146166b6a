// ctg_common.h -- structures shared by the host runtime and the gfx950 kernels.
//
// Addressing model (see cotengra_amd/plan.py): every operand of a step is
//   base + *slice_offset + sum over index groups of an offset-table entry
// where the tables are int64 element offsets living in one device blob.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include <stdint.h>

namespace ctg {

// An environment switch is ON when it is set to anything but "" or "0" (the convention of
// every CTG_* switch, on the Python side as well).
inline bool env_on(const char* name) {
    const char* v = getenv(name);
    return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is honoured per device: remember per
// (kernel instantiation, device) that it was done.  `done` is a per-instantiation bit mask
// of devices (an executor may be created on any of a node's 8 GPUs in one process).
inline hipError_t lds_opt_in(const void* kern, int bytes, unsigned long long* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // (do not leave it for the next launch's error check to find)
        return e;
    }
    __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return hipSuccess;
}


// word offsets of the serialised step record (must match plan.py: Plan.serialise)
enum StepWord {
    W_KIND = 0, W_KERNEL = 1,
    W_A_SPACE = 2, W_A_OFF = 3, W_A_LEAF = 4,
    W_B_SPACE = 5, W_B_OFF = 6, W_B_LEAF = 7,
    W_C_SPACE = 8, W_C_OFF = 9, W_C_LEAF = 10,
    W_R = 11, W_BT = 12, W_K = 13, W_N = 14,
    W_ROW_LO = 15, W_ROW_HI_LEN = 16,
    W_ROWA_HI = 17, W_ROWA_LO = 18, W_ROWB_HI = 19, W_ROWB_LO = 20,
    W_ROWC_HI = 21, W_ROWC_LO = 22,
    W_KA = 23, W_KB = 24, W_NB = 25, W_NC = 26,
    W_BA = 27, W_BB = 28, W_BC = 29,
    W_A_SIZE = 30, W_B_SIZE = 31, W_C_SIZE = 32,
    W_MACS = 33, W_ELEMS = 34, W_NODE = 35,
    W_K_LO = 36, W_KA_HI = 37, W_KB_HI = 38, W_K_HI_LEN = 39,
    W_A_PROD = 40, W_B_PROD = 41,  // step that produced the operand (-1: input / preprocessing)
    W_INVARIANT = 42,              // 1: does not depend on sliced inputs, run once per upload;
                                   // 2 (round 4): does not depend on the plan's GROUP indices -- run once per
                                   // group of slices (ctg_plan_desc.slice_group)
    W_STEM = 43,                   // KIND_STEM2: word offset of the descriptor in the table blob
    W_LDS_COMP = 44,               // (round 6) member of an LDS-resident subtree: component id + 1 (0: none)
    W_LDS_DESC = 45,               // ... word offset of the component's descriptor in the table blob
    STEP_WORDS = 48
};

// KIND_STEM2: two consecutive pair steps of a stem as one launch (plan: cotengra_amd/stem.py,
// kernel: ctg_stem.hip).  A / B / C of the record are the big operand, the first small
// operand and the result; everything else is in the descriptor.
enum Kind { KIND_SINGLE = 0, KIND_PAIR = 1, KIND_ACCUM = 2, KIND_STEM2 = 3 };

// header of a STEM2 descriptor (int64 words; stem.py: serialise_stem)
enum StemWord {
    SW_MAGIC = 0, SW_K1 = 1, SW_N1 = 2, SW_K2 = 3, SW_N2 = 4, SW_NR1 = 5, SW_ROWS2 = 6, SW_NG2 = 7,
    SW_NTILES = 8, SW_GLO = 9, SW_LD2 = 10, SW_LDS = 11,
    SW_B2_SPACE = 12, SW_B2_OFF = 13, SW_B2_LEAF = 14, SW_B2_SIZE = 15, SW_B2_PROD = 16,
    SW_VEC = 17,    // 1: slots 2 q, 2 q + 1 of a task are adjacent in memory (one 16-byte load)
    SW_ONE = 18,    // 1: the first half alone (ONE step: K2 = N2 = rows2 = 0, no B2; out_row over the
                    //    tile rows of step 1, out_col over its columns) -- stem.py: build_stem_one
    SW_TABS = 20,   // 14 table offsets: gA_hi gA_lo gC_hi gC_lo kj_a lane_a rt_a chunk_a
                    //                   b1_off b2_off mid_row mid_col out_row out_col
    // a MIDDLE stage (three-step tile, stem.py: build_stem_triple; round 4): the fields above that
    // speak of "step 2" then describe the LAST step, these the one between -- its shape, its small
    // operand, its tables (bm_off, mid2_row, mid2_col: its result -> the second intermediate)
    SW_KM = 34, SW_NM = 35, SW_ROWSM = 36, SW_NGM = 37, SW_LDM = 38, SW_TRI = 39,
    SW_BM_SPACE = 40, SW_BM_OFF = 41, SW_BM_LEAF = 42, SW_BM_SIZE = 43, SW_BM_PROD = 44,
    SW_TABS_M = 45,
    STEM_WORDS = 56
};
constexpr int64_t STEM_MAGIC = 0x53544D33;
enum StemTab {
    ST_GA_HI = 0, ST_GA_LO, ST_GC_HI, ST_GC_LO, ST_KJ_A, ST_LANE_A, ST_RT_A, ST_CHUNK_A,
    ST_B1_OFF, ST_B2_OFF, ST_MID_ROW, ST_MID_COL, ST_OUT_ROW, ST_OUT_COL, ST_COUNT
};
enum Kernel { KERNEL_VALU = 0, KERNEL_MFMA = 1 };
enum Space { SPACE_INPUTS = 0, SPACE_ARENA = 1, SPACE_RESULT = 2 };

// Two-level row table: off(i) = hi[i / lo_size] + lo[i % lo_size].
struct RowTab {
    const int64_t* hi;
    const int64_t* lo;
};

// Kernel arguments of every step kind.  Pointers are device pointers; `soff*`
// point at the per-slice base offset of the operand (or at a constant 0).
struct StepArgs {
    const void* A;
    const void* B;
    void* C;
    const int64_t* soffA;
    const int64_t* soffB;
    const int64_t* soffC;
    int64_t R;   // rows (VALU: batch*M, MFMA: M)
    int64_t Bt;  // batch (MFMA)
    int64_t K;
    int64_t N;
    int64_t row_lo;      // size of the fast level of the row tables
    int32_t row_lo_shift;  // log2(row_lo) if a power of two, else -1
    RowTab rowA, rowB, rowC;
    // contracted group, two-level as well: K = k_hi_len * k_lo
    int64_t k_lo;
    int64_t k_hi_len;
    int32_t k_lo_shift;  // log2(k_lo) if a power of two, else -1
    RowTab kA, kB;
    const int64_t* nB;
    const int64_t* nC;
    const int64_t* bA;
    const int64_t* bB;
    const int64_t* bC;
    // strip_exponent (reference contract.py:816-829): max|.| of the operands'
    // producers; the step stores alpha * (A . B) with alpha = 1 / (facA * facB),
    // i.e. the contraction of the normalised operands.  nullptr = feature off.
    const double* facA;
    const double* facB;
    int32_t check_zero;
    // Slice batching: a launch may carry `nz` consecutive slices of a run in
    // gridDim.y (kernels whose launcher cannot, loop over z0 instead).  Slice-in-
    // batch z = z0 + blockIdx.y reads its per-leaf base offsets at soffX[z * zsX]
    // and its replica of the per-slice arena at z * zX elements (both 0 for
    // operands that are the same in every slice: inputs without a sliced index,
    // slice-invariant intermediates, the result tensor).
    int32_t nz, z0;
    // exact division by a row_lo / k_lo that is not a power of two (extent-3 indices of
    // hyper networks): q = umul64hi(i, magic) with magic = floor(2^64 / d) + 1, valid for
    // i < 2^32; 0 = not available (fall back to the 64-bit division)
    uint64_t row_lo_magic, k_lo_magic;
    int64_t scratch_total;  // bytes of scratch actually allocated (>= the 64 MiB the split
                            // heuristics are computed with): how many slices fit one launch
    int64_t zA, zB, zC;     // arena replica strides (elements)
    int64_t zsA, zsB, zsC;  // strides of the soff arrays (entries)
    // slice groups in a batched launch (round 4): operand A / B was written by a step the zq slices of a
    // group share -- it lives with the group's first slice: slice-in-batch z reads it at z / zq * zq (0, 1: off)
    int32_t zqA, zqB;
    // this executor multiplies its long tiled steps (and its stem pairs) with bf16 x 3 products
    // (ctg_exec_set_stem_arithmetic; csrc/ctg_pair_mfma.hip: pair_mfma_bf3_kernel)
    int32_t bf3;
};

// Kernel arguments of a STEM2 step.
struct StemArgs {
    const void* A;
    const void* B1;
    const void* B2;
    void* C;
    const int64_t* soffA;
    const int64_t* soffB1;
    const int64_t* soffB2;
    const int64_t* soffC;
    int32_t K1, N1, K2, N2;
    int32_t nr1;        // log2 of the tile rows of step 1 (8 or 9)
    int32_t rows2;      // rows of the intermediate tile as step 2 sees it
    int32_t ng2;        // 32-column groups of step 2
    int32_t ld2;        // K2 + 4
    int64_t n_tiles, g_lo;
    int32_t g_lo_shift;
    int32_t check_zero;
    int32_t vec;        // 16-byte gathers (SW_VEC)
    int32_t bf3;        // this executor multiplies its pairs on the bf16 matrix cores (ctg_exec_set_stem_arithmetic)
    const int64_t* gA_hi;
    const int64_t* gA_lo;
    const int64_t* gC_hi;
    const int64_t* gC_lo;
    const int64_t* kj_a;     // [8]  slot j of a task: k = 2 j
    const int64_t* lane_a;   // [64] lane part of a task's addresses (row l & 31, k parity l >> 5)
    const int64_t* rt_a;
    const int64_t* chunk_a;
    const int64_t* b1_off;
    const int64_t* b2_off;
    const int64_t* mid_row;
    const int64_t* mid_col;
    const int64_t* out_row;
    const int64_t* out_col;
    const double* facA;   // strip_exponent: max|.| of the three operands' producers (or null)
    const double* facB1;
    const double* facB2;
    int32_t nz, z0;       // slice batching, as StepArgs
    int64_t zA, zB1, zB2, zC;
    int64_t zsA, zsB1, zsB2, zsC;
    int32_t one;                // the first half alone (SW_ONE)
    // three-step tile (SW_TRI): the middle step
    int32_t tri, KM, NM, rowsM, ngM, ldM;
    const void* BM;
    const int64_t* soffBM;
    const int64_t* bm_off;
    const int64_t* mid2_row;    // [rowsM] + mid2_col [NM]: the middle step's result in the second intermediate
    const int64_t* mid2_col;
    const double* facBM;
    int64_t zBM, zsBM;
    int64_t a_elems, c_elems;   // extents of the big operand and of the result (the bounds-checked
                                // experiment build -DCTG_STEM_BOUNDS tests every gather and store)
    // (round 6) the fp16 x 2 arithmetic (ctg_stem.hip built with -DCTG_STEM_H2: launch_stem2h): largest |component| of
    // the big operand as its producer recorded it, and where this launch records that of its result (device floats)
    const float* amax;
    float* cmax;
};

// element offset of an operand for the slice-in-batch this block works on
__device__ __forceinline__ int64_t zid(const StepArgs& p) { return (int64_t)p.z0 + blockIdx.y; }
// A record of a largest |component| is kMaxSub floats (one 64-byte line): a wave records into one of them, picked by its
// workgroup and wave -- 500 K waves of a launch on ONE address cost a 60 us kernel 20-60 us of atomics --, a reader takes
// the largest of the kMaxSub.  (lane 0 calls record_max; v > 0, finite)
constexpr int kMaxSub = 16;
__device__ __forceinline__ void record_max(float* slot, float v) {
    atomicMax((unsigned*)slot + ((blockIdx.x + (threadIdx.x >> 6)) & (kMaxSub - 1)), __builtin_bit_cast(unsigned, v));
}
__device__ __forceinline__ float read_max(const float* slot) {
    float m = slot[0];
#pragma unroll
    for (int i = 1; i < kMaxSub; ++i) m = fmaxf(m, slot[i]);
    return m;
}
__device__ __forceinline__ int64_t zgroup(int64_t z, int32_t q) { return q > 1 ? z / q * q : z; }
__device__ __forceinline__ int64_t zoffA(const StepArgs& p) { const int64_t z = zgroup(zid(p), p.zqA); return p.soffA[z * p.zsA] + z * p.zA; }
__device__ __forceinline__ int64_t zoffB(const StepArgs& p) { const int64_t z = zgroup(zid(p), p.zqB); return p.soffB[z * p.zsB] + z * p.zB; }
__device__ __forceinline__ int64_t zoffC(const StepArgs& p) { const int64_t z = zid(p); return p.soffC[z * p.zsC] + z * p.zC; }

__device__ __forceinline__ double step_alpha(const StepArgs& p) {
    if (p.facA == nullptr) return 1.0;
    const double f = (*p.facA) * (*p.facB);
    if (f == 0.0 && p.check_zero) return 0.0;
    return 1.0 / f;  // inf -> nan downstream, like the reference without check_zero
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Tell the compiler a 64-bit value is wave-uniform so that loads indexed by it
// become scalar loads (s_load: own counter, no vmcnt drain of the vector
// memory pipeline).  Only call with values that ARE uniform across the wave.
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffll));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// Epilogue of the complex-on-real MFMA tiles.  A 32x32 tile leaves Re of complex
// column n in lane 2n and Im in lane 2n+1, registers t / t+1 holding two
// adjacent output rows.  Swapping with the neighbour lane (DPP quad_perm
// [1,0,3,2]: a plain VALU move, no LDS round trip like ds_bpermute) gives the
// even lane the whole complex number of row t and the odd lane that of row
// t+1, so every lane stores 8 contiguous bytes.  Written without a select
// between two elements of the accumulator vector on purpose: the compiler
// turns that into a per-lane *indexed* extract (a 16-deep v_cndmask chain).
__device__ __forceinline__ float dpp_swap1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float2 pair_rows(float a_t, float a_t1, bool odd) {
    const float x = dpp_swap1(a_t), y = dpp_swap1(a_t1);
    float2 v;
    v.x = odd ? y : a_t;
    v.y = odd ? a_t1 : x;
    return v;
}
__device__ __forceinline__ float2 pair_rows(float a_t, float a_t1, bool odd, float alpha) {
    const float x = dpp_swap1(a_t), y = dpp_swap1(a_t1);
    float2 v;
    v.x = (odd ? y : a_t) * alpha;
    v.y = (odd ? a_t1 : x) * alpha;
    return v;
}

// Scalar load of a table entry at a wave-uniform address.  The offset tables
// are written before the kernel starts and never by it, but the compiler
// cannot prove that once a persistent kernel has stored to C -- the constant
// address space tells it, and keeps the lookups on the scalar unit (s_load).
__device__ __forceinline__ int64_t sload64(const int64_t* p) {
    typedef const int64_t __attribute__((address_space(4))) * cptr;
    return *(cptr)(uintptr_t)p;
}

// the same through scalar loads (z is uniform over the block)
__device__ __forceinline__ int64_t zoffA_s(const StepArgs& p) { const int64_t z = zgroup(zid(p), p.zqA); return sload64(p.soffA + z * p.zsA) + z * p.zA; }
__device__ __forceinline__ int64_t zoffB_s(const StepArgs& p) { const int64_t z = zgroup(zid(p), p.zqB); return sload64(p.soffB + z * p.zsB) + z * p.zB; }
__device__ __forceinline__ int64_t zoffC_s(const StepArgs& p) { const int64_t z = zid(p); return sload64(p.soffC + z * p.zsC) + z * p.zC; }

__device__ __forceinline__ void split_k(const StepArgs& p, int64_t k, int64_t& hi, int64_t& lo) {
    if (p.k_lo_shift >= 0) {
        hi = k >> p.k_lo_shift;
        lo = k & (p.k_lo - 1);
    } else if (p.k_lo_magic != 0) {
        hi = (int64_t)__umul64hi((uint64_t)k, p.k_lo_magic);
        lo = k - hi * p.k_lo;
    } else {
        hi = k / p.k_lo;
        lo = k - hi * p.k_lo;
    }
}

__device__ __forceinline__ void split_row(const StepArgs& p, int64_t i, int64_t& hi, int64_t& lo) {
    if (p.row_lo_shift >= 0) {
        hi = i >> p.row_lo_shift;
        lo = i & (p.row_lo - 1);
    } else if (p.row_lo_magic != 0) {
        hi = (int64_t)__umul64hi((uint64_t)i, p.row_lo_magic);
        lo = i - hi * p.row_lo;
    } else {
        hi = i / p.row_lo;
        lo = i - hi * p.row_lo;
    }
}

// ---- complex arithmetic on plain structs (no thrust / hip_complex) -------- //

template <typename F>
struct cplx {
    F re, im;
};
typedef cplx<float> c64;
typedef cplx<double> c128;

template <typename T> struct Acc;  // accumulate helper

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ c64 zero_of(c64) { return c64{0.f, 0.f}; }
__device__ __forceinline__ c128 zero_of(c128) { return c128{0.0, 0.0}; }

__device__ __forceinline__ void fma_acc(float& acc, float a, float b) { acc = fmaf(a, b, acc); }
__device__ __forceinline__ void fma_acc(double& acc, double a, double b) { acc = fma(a, b, acc); }
template <typename F>
__device__ __forceinline__ void fma_acc(cplx<F>& acc, cplx<F> a, cplx<F> b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
}

__device__ __forceinline__ float scale_of(float a, double s) { return a * (float)s; }
__device__ __forceinline__ double scale_of(double a, double s) { return a * s; }
__device__ __forceinline__ cplx<float> scale_of(cplx<float> a, double s) {
    return cplx<float>{a.re * (float)s, a.im * (float)s};
}
__device__ __forceinline__ cplx<double> scale_of(cplx<double> a, double s) {
    return cplx<double>{a.re * s, a.im * s};
}
__device__ __forceinline__ double abs_of(float a) { return fabs((double)a); }
__device__ __forceinline__ double abs_of(double a) { return fabs(a); }
__device__ __forceinline__ double abs_of(cplx<float> a) { return hypot((double)a.re, (double)a.im); }
__device__ __forceinline__ double abs_of(cplx<double> a) { return hypot(a.re, a.im); }

__device__ __forceinline__ float add_of(float a, float b) { return a + b; }
__device__ __forceinline__ double add_of(double a, double b) { return a + b; }
template <typename F>
__device__ __forceinline__ cplx<F> add_of(cplx<F> a, cplx<F> b) {
    return cplx<F>{a.re + b.re, a.im + b.im};
}

// wave-wide (64 lanes) sum via cross-lane shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <typename F>
__device__ __forceinline__ cplx<F> wave_sum(cplx<F> v) {
    return cplx<F>{wave_sum(v.re), wave_sum(v.im)};
}

// ---- MFMA kernel hints built on the host per step (ctg_runtime.hip) -------- //

constexpr int MFMA_BM = 128;  // rows of a block tile
constexpr int MFMA_BK = 16;   // complex k per step

// ordA: [256][BM*BK/256] packed (r << 4 | c): the tile elements a thread
// gathers, in ascending memory order across the lanes of each load
// instruction; with vecA entries (2j, 2j+1) are adjacent in memory (one
// 16-byte load).  ordB: [256][ceil(BK*BN/256)] packed (n << 4 | k)
// ([64][BK*BN/64] for the k-streaming kernel).
struct MfmaHints {
    const uint16_t* ordA;
    const uint16_t* ordB;
    int bn;    // column tile: 16, 32 or 64
    int vecA;  // 1: pairs of A elements are contiguous + aligned, tiles are full
    int stream;      // 1: tall-skinny streaming kernel (row tile 32, B resident in LDS);
                     // 2: k-streaming kernel (R, N <= 32, both operands stream along K)
                     // 3: skinny FMA kernel (K <= 16, N <= 4: far below one MFMA tile)
                     // 4: row-wise FMA kernel (K <= 32, N <= 32, any extents and layout)
    int additive32;  // 1: row offsets are tile-additive for 32-row groups
    int fast;        // 1: full tiles + tile-additive 32-bit offsets (tiled fast path)
    const void* lane;  // fast path: per-thread gather / staging constants, built once per
                       // executor (launch_fast_lane_consts); null: computed by every block
    int splitk;        // tiled kernels: number of k-splits (>= 1), fixed when the executor is
                       // built -- a function of the step alone, so that a result does not
                       // depend on how many slices share a launch or on the tile width
    int bf3;           // round 5: a LONG tiled step (K >= 64, N a multiple of 64, full tiles at bn = 64) --
                       // its products may run as six bf16 products (pair_mfma_bf3_kernel).  A function of
                       // the step alone: such a step keeps tiles of >= 64 columns in every launch and stays
                       // out of the wave-front groups, so that the arithmetic never depends on batching
    // round 6, set PER LAUNCH by the executor on its copy of a bf3 step's hints (ctg_runtime.hip: KIND_PAIR):
    int h2;             // 1: this launch multiplies with two fp16 limbs (pair_mfma_h2_kernel)
    const float* amax;  // h2: the largest |component| of the operands (device memory; recorded by their producers,
    const float* bmax;  //     or found by a max-abs pass): the power of two each operand is split under
    float* cmax;        // either 16-bit arithmetic: where the launch records its result's largest |component| (or null)
    // slice z of a launch (StepArgs::z0 + blockIdx.y) reads the records at amax + z * amax_zs * kMaxSub, bmax + ..., and
    // records into cmax + z * cmax_zs * kMaxSub: a record per slice of a batch (stride 1), one for all of them where the
    // tensor is slice-invariant (0)
    int amax_zs, bmax_zs, cmax_zs;
};

// k-splits of a tiled step: when the output alone cannot fill the chip but K is long
// (slabs of the partial tiles go through `scratch_bytes` of scratch memory)
// (zn: slices a launch of this plan nominally carries -- they fill the chip like tiles do)
inline int64_t mfma_split_count(int64_t R, int64_t N, int64_t K, int64_t Bt, int bn, int64_t scratch_bytes,
                                int64_t zn = 1) {
    const int64_t tiles_m = (R + MFMA_BM - 1) / MFMA_BM, tiles_n = (N + bn - 1) / bn;
    const int64_t tiles = tiles_m * tiles_n * Bt * (zn > 1 ? zn : 1);
    const int64_t nk_total = (K + MFMA_BK - 1) / MFMA_BK;
    int64_t S = 1;
    if ((tiles < 256 && nk_total >= 16) || (tiles < 512 && nk_total >= 64)) {
        S = (1024 + tiles - 1) / tiles;
        if (S > nk_total / 4) S = nk_total / 4;
        const int64_t slab_bytes = tiles_m * MFMA_BM * tiles_n * bn * 8 * Bt;
        if (S * slab_bytes > scratch_bytes) S = scratch_bytes / slab_bytes;
        if (S > 65535) S = 65535;
        if (S < 1) S = 1;
    }
    if (S > nk_total) S = nk_total;
    return S;
}

// steps the streaming kernel takes: short contraction, few columns, many rows
inline bool mfma_use_stream(int64_t R, int64_t Bt, int64_t K, int64_t N) {
    // thresholds measured on the m20 steps (wider column tiles cost registers, i.e.
    // waves in flight; longer K costs the resident B panel's LDS)
    // (measured: with 32 columns the tiled kernel wins from K = 128 on -- the resident
    // B panel then costs the streaming kernel a third of its waves)
#ifdef CTG_STREAM_WIDE   // experiment builds: wider steps on the streaming kernel
    return Bt == 1 && (N <= 32 ? K <= 128 : (N <= 64 && K <= 64)) && R >= 8192;
#endif
    return Bt == 1 && (N <= 16 ? K <= 128 : (N <= 32 ? K <= 64 : (N <= 64 && K <= 32))) && R >= 8192;
}

inline int mfma_pick_bn(int64_t N) { return N <= 16 ? 16 : (N <= 32 ? 32 : 64); }

// exponent state of a strip_exponent run (device memory)
struct StripState {
    double E;       // running exponent of the result tensor (-inf: empty)
    double e_slice; // exponent of the slice just computed
    double coefM;   // rescale of the existing result  = 10^(E - E')
    double coefm;   // scale of the incoming slice      = 10^(e_slice - E') / fac_root
    int32_t zero;   // a zero intermediate was met (check_zero)
    int32_t pad;
};

// A launcher whose kernel cannot carry the slices of a batch in gridDim.y (it
// owns scratch memory, or its grid is persistent) runs them one after another.
template <typename F>
inline hipError_t for_each_z(const StepArgs& p, F launch_one) {
    if (p.nz <= 1) return launch_one(p);
    for (int z = 0; z < p.nz; ++z) {
        StepArgs q = p;
        q.z0 = p.z0 + z;
        q.nz = 1;
        const hipError_t e = launch_one(q);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The same in chunks of at most `fit` slices (scratch memory holds that many at once).
template <typename F>
inline hipError_t for_each_z_chunk(const StepArgs& p, int64_t fit, F launch_chunk) {
    if (fit < 1) fit = 1;
    for (int64_t z = 0; z < p.nz; z += fit) {
        StepArgs q = p;
        q.z0 = p.z0 + (int32_t)z;
        q.nz = (int32_t)(p.nz - z < fit ? p.nz - z : fit);
        const hipError_t e = launch_chunk(q);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- launchers implemented in the kernel translation units ---------------- //

// dtype: 0 f32, 1 f64, 2 c64, 3 c128
hipError_t launch_pair_valu(int dtype, const StepArgs& p, void* scratch, int64_t scratch_bytes,
                            hipStream_t stream);
// Independent small thread-per-output steps sharing one launch: item i owns the
// workgroups [block_begin, block_begin + n_blocks) of the grid.
struct ValuGroupItem {
    StepArgs p;
    int64_t col_tiles, n_tiles;
    int32_t tn_shift;
    uint32_t block_begin, n_blocks;
};
constexpr int64_t kValuGroupMaxTiles = 4096;  // larger steps fill the chip on their own
bool valu_thread_per_output(const StepArgs& p);  // the route launch_pair_valu takes for this step
uint32_t valu_group_fill(const StepArgs& p, ValuGroupItem* it, uint32_t block_begin);  // -> n_blocks
hipError_t launch_pair_valu_group(int dtype, const ValuGroupItem* d_items, int n_items, uint32_t blocks,
                                  int nz, hipStream_t stream);
hipError_t launch_pair_mfma(int dtype, const StepArgs& p, const MfmaHints& h, void* scratch,
                            int64_t scratch_bytes, hipStream_t stream);
// Independent small steps of the tiled fast kernel sharing one launch (same tile
// shape and gather width: `key`); item i owns workgroups [block_begin, +n_blocks).
struct FastGroupItem {
    StepArgs p;
    MfmaHints h;
    int64_t tiles_m, tiles_n;
    uint32_t block_begin, n_blocks;
};
constexpr int64_t kFastGroupMaxTiles = 128;  // larger steps fill the chip on their own
int fast_group_key(const StepArgs& p, const MfmaHints& h);  // -1: the step launches alone
uint32_t fast_group_fill(const StepArgs& p, const MfmaHints& h, FastGroupItem* it, uint32_t block_begin);
hipError_t launch_pair_mfma_fast_group(int key, const FastGroupItem* d_items, int n_items, uint32_t blocks,
                                       int nz, hipStream_t stream);
bool rowwise_ok(const StepArgs& p);  // shapes the row-wise kernel takes (MfmaHints::stream == 4)
int64_t fast_lane_table_bytes();
hipError_t launch_fast_lane_consts(const StepArgs& p, const MfmaHints& h, void* out, hipStream_t stream);
// complex128 on the FP64 matrix cores (ctg_pair_mfma_f64.hip)
hipError_t launch_pair_mfma_c128(const StepArgs& p, int flags, hipStream_t stream);
// float32 / float64 on the 16x16x4 matrix-core instructions
hipError_t launch_pair_mfma_real(int dtype, const StepArgs& p, int flags, hipStream_t stream);
// fused stem pair (ctg_stem.hip)
bool stem2_supported(const StemArgs& p);
bool stem3_supported(const StemArgs& p);   // (a three-step tile: shape, instantiation, LDS)
size_t stem2_lds_bytes(const StemArgs& p);
hipError_t launch_stem2(const StemArgs& p, hipStream_t stream);
void stem2_kernel_name(const StemArgs& p, char* buf, size_t n);
// the same kernels in the fp16 x 2 arithmetic (round 6; StemArgs::amax / cmax)
bool stem2h_supported(const StemArgs& p);
bool stem2h_uses_h2(const StemArgs& p);   // (else the launch is the fp32 kernel of that object: no record of the result)
bool stem2_uses_bf3(const StemArgs& p);   // the same question for launch_stem2 (bf16 x 3: records the result's largest element too)
hipError_t launch_stem2h(const StemArgs& p, hipStream_t stream);
void stem2h_kernel_name(const StemArgs& p, char* buf, size_t n);

hipError_t launch_single(int dtype, const StepArgs& p, hipStream_t stream);
bool pair_bf16x3_on(const StepArgs& p);   // (ctg_pair_mfma.hip) do long tiled steps multiply with bf16 x 3 products right now?
hipError_t launch_accum(int dtype, const StepArgs& p, const StripState* st, void* wide, const double* inscale, hipStream_t stream);
// (single-precision trees) inputs far from 1 are brought to [1, 2) by an exact power of two at upload; inscale[0] =
// the product of the powers taken out, inscale[1] = its log10
hipError_t launch_prescale_inputs(int dtype, void* inputs, const int64_t* offs, const int64_t* sizes, int64_t n_inputs,
                                  int* shift_total, double* inscale, hipStream_t stream);
// (float / complex64 results: the double-precision running sum next to the result -- ctg_kernels_valu.hip)
hipError_t launch_narrow(int dtype, void* result, const void* wide, int64_t n, hipStream_t stream);
hipError_t launch_widen(int dtype, void* wide, const void* result, int64_t n, hipStream_t stream);

hipError_t launch_maxabs(int dtype, const void* x, int64_t n, double* fac, hipStream_t stream);
// e_slice = sum_s log10(fac[s]) over the listed steps; coefficients for the accumulate
hipError_t launch_strip_prepare(const double* fac, const int32_t* counted, int64_t n_steps,
                                int64_t root_step, int check_zero, StripState* st, const double* inscale, hipStream_t stream);
hipError_t launch_rescale(int dtype, void* result, int64_t n, const StripState* st, hipStream_t stream);

struct SliceMeta {
    int64_t n_leaves;  // n_inputs + 1 (last = result chunk offset)
    int64_t n_sliced;
    const int64_t* sizes;    // [n_sliced]
    const int64_t* fixed;    // [n_sliced]
    const int64_t* strides;  // [n_leaves * n_sliced]
    double* fac;             // strip_exponent: per-step max|.| scalars (or null)
    const int32_t* fac_zero; // [n_fac] 1: zero the scalar at slice start (per-slice pair steps)
    int64_t n_fac;
    // (round 6) fp16 x 2 stem kernels: the largest element each stem step recorded for its result; the slots of
    // per-slice steps start every slice at zero (smax_zero[i] = 1), what a group shares or is slice-invariant keeps its own
    float* smax;
    const int32_t* smax_zero;
    int64_t n_smax;
};
// soff[n_leaves] receives the per-leaf base offsets of slice `sid`; with sid < 0
// the id is taken from the device counter state[0], which is then advanced by
// state[1] (lets a captured graph walk over slices without host involvement)
// (nz > 1: the offsets of slices sid, sid + stride, ... in soff[z * n_leaves + leaf])
// (ids != nullptr: the nz slices are ids[0 .. nz) -- device memory -- instead of sid + z * stride)
hipError_t launch_prologue(const SliceMeta& m, int64_t* state, int64_t* soff, int64_t sid,
                           hipStream_t stream, int nz = 1, int64_t stride = 1, const int64_t* ids = nullptr);
// state[0] = next, state[1] = stride (device-side slice counter of a graph replay)
hipError_t launch_set_state(int64_t* state, int64_t next, int64_t stride, hipStream_t stream);

}  // namespace ctg
