// ctg_pair_mfma_f64.hip -- complex128 gather-GEMM on the gfx950 FP64 matrix cores
// (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s peak): the double-precision parity mode
// of the contraction path (the reference's tests run every case in complex128 /
// float64 as well, tests/test_compute.py:102-115).
//
//   C[bC(b) + rowC(m) + nC(n)] = sum_k A[bA(b) + rowA(m) + kA(k)] * B[bB(b) + kB(k) + nB(n)]
//
// Same formulation as the complex64 kernel: one MFMA consumes 4 real k = 2
// complex k with A' = (Re, Im) pairs along k and B' = [[Re, Im], [-Im, Re]]
// interleaved along the 16 real columns (8 complex columns), so no flop is
// wasted and D is interleaved complex.  Block tile 64 x 32 complex, 8 complex
// k per step, 4 waves as 2 x 2 (wave tile 32 x 16 = 2 x 2 MFMA tiles),
// register-staged double buffering, bounds masks everywhere (general shapes),
// row offsets and the k offsets of the next steps resolved into LDS.
#include "ctg_common.h"

#include <type_traits>

namespace ctg {

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BK = 8, LD = BK + 1;
}  // namespace

// TM x TN MFMA tiles (16 rows x 8 complex columns each) per wave, 2 x 2 waves per
// block: 64 x 32 (TM = TN = 2) for small steps, 128 x 32 (TM = 4) when there are
// enough rows -- twice the matrix work per k-step, barrier and gathered element.
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void pair_mfma_c128_kernel(StepArgs p, int flags,
                                                               int64_t tiles_m, int64_t tiles_n) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 8;
    // a ds_read_b64 is served 32 lanes at a time: 16 rows of the Re plane and the same
    // 16 rows of the Im plane.  Rows are 18 banks apart (LD = 9 doubles) and BM * LD is a
    // multiple of the 64-bank period, so un-shifted the two planes collide bank for
    // bank; 16 doubles (32 banks) of shift put the Im rows exactly on the 16 bank pairs
    // the Re rows leave free
    constexpr int A_PLANE = BM * LD + 16;
    static_assert((BM * LD * 2) % 64 == 0, "plane shift assumes BM * LD is a multiple of the bank period");
    constexpr int A_DBL = 2 * A_PLANE, B_DBL = 2 * BN * LD;
    constexpr int NA = BM * BK / 256, NB = (BN * BK + 255) / 256;
    static_assert(BN * BK % 256 == 0, "B tile must divide over the block");
    __shared__ double lds[2 * (A_DBL + B_DBL)];
    __shared__ int64_t rowA_s[BM];
    __shared__ int64_t rowC_s[BM];
    __shared__ int64_t kofs_s[3][2][BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware: consecutive row tiles on different XCDs, column tiles of a row
    // tile on the same one (they share the A panel through that XCD's L2)
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, q = bid >> 3;
    const int64_t tm = (q / tiles_n) * 8 + xcd;
    const int64_t tn = q % tiles_n;
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int64_t bz = blockIdx.z;

    const c128* __restrict__ A = (const c128*)p.A + zoffA(p) + p.bA[bz];
    const c128* __restrict__ B = (const c128*)p.B + zoffB(p) + p.bB[bz];
    double* __restrict__ C = (double*)((c128*)p.C + zoffC(p) + p.bC[bz]);
    const bool a_kfast = flags & 1, b_kfast = flags & 2;
    const int64_t nk = (p.K + BK - 1) / BK;

    if (tid < BM) {
        const int64_t m = m0 + tid;
        int64_t oa = -1, oc = -1;
        if (m < p.R) {
            int64_t hi, lo;
            split_row(p, m, hi, lo);
            oa = p.rowA.hi[hi] + p.rowA.lo[lo];
            oc = p.rowC.hi[hi] + p.rowC.lo[lo];
        }
        rowA_s[tid] = oa;
        rowC_s[tid] = oc;
    }
    int64_t kofs_val = -1;
    auto kofs_fetch = [&](int64_t step) {  // threads tid < 2*BK
        const int which = tid / BK, c = tid % BK;
        const int64_t k = step * BK + c;
        int64_t off = -1;
        if (k < p.K) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            off = which ? p.kB.hi[kh] + p.kB.lo[kl] : p.kA.hi[kh] + p.kA.lo[kl];
        }
        kofs_val = off;
    };
    auto kofs_commit = [&](int64_t step) { kofs_s[step % 3][tid / BK][tid % BK] = kofs_val; };
    if (tid < 2 * BK) {
        for (int s = 0; s < 3 && s < nk; ++s) {
            kofs_fetch(s);
            kofs_commit(s);
        }
    }

    // ---- gather coordinates: NA A elements + NB B elements per thread per step ----
    int a_r[NA], a_c[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = j * 256 + tid;
        if (a_kfast) {
            a_r[j] = e / BK;
            a_c[j] = e % BK;
        } else {
            a_r[j] = e % BM;
            a_c[j] = e / BM;
        }
    }
    int b_k[NB], b_n[NB];
    int64_t b_col[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int e = j * 256 + tid;
        if (b_kfast) {
            b_k[j] = e % BK;
            b_n[j] = e / BK;
        } else {
            b_k[j] = e / BN;
            b_n[j] = e % BN;
        }
        b_col[j] = (n0 + b_n[j] < p.N) ? p.nB[n0 + b_n[j]] : -1;
    }
    __syncthreads();
    int64_t a_row[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) a_row[j] = rowA_s[a_r[j]];

    c128 a_reg[NA], b_reg[NB];
    auto gather = [&](int64_t step) {
        const int64_t* ka = kofs_s[step % 3][0];
        const int64_t* kb = kofs_s[step % 3][1];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int64_t ko = ka[a_c[j]];
            c128 v{0.0, 0.0};
            if (a_row[j] >= 0 && ko >= 0) v = A[a_row[j] + ko];
            a_reg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int64_t ko = kb[b_k[j]];
            c128 v{0.0, 0.0};
            if (b_col[j] >= 0 && ko >= 0) v = B[b_col[j] + ko];
            b_reg[j] = v;
        }
    };
    auto stage = [&](int buf) {
        double* As = lds + buf * (A_DBL + B_DBL);
        double* Bs = As + A_DBL;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            As[a_r[j] * LD + a_c[j]] = a_reg[j].re;
            As[A_PLANE + a_r[j] * LD + a_c[j]] = a_reg[j].im;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            Bs[(2 * b_n[j]) * LD + b_k[j]] = b_reg[j].re;
            Bs[(2 * b_n[j] + 1) * LD + b_k[j]] = b_reg[j].im;
        }
    };

    f64x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0.0;

    // fragment coordinates of v_mfma_f64_16x16x4_f64: lane -> (i16, k4)
    const int i16 = lane & 15, k4 = lane >> 4;
    const int part = k4 & 1;       // 0: Re a / first B' row of the pair, 1: Im a / second
    const int kc_in = k4 >> 1;     // which of the MFMA's two complex k
    const int cc = i16 & 1;        // B'/D column parity: 0 = real part, 1 = imaginary part
    const bool negate = part == 1 && cc == 0;

    gather(0);
    stage(0);
    if (nk > 1) gather(1);
    __syncthreads();

    for (int64_t kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1);
        const bool kofs_mine = kt + 3 < nk && tid < 2 * BK;
        if (kofs_mine) kofs_fetch(kt + 3);
        if (kt + 2 < nk) gather(kt + 2);

        const double* As = lds + buf * (A_DBL + B_DBL);
        const double* Bs = As + A_DBL;
        const double* a_base = As + part * A_PLANE + (wm * TM * 16 + i16) * LD + kc_in;
        const double* b_base = Bs + (2 * (wn * TN * 8 + (i16 >> 1)) + (cc ^ part)) * LD + kc_in;
        // fragments double-buffered in registers: the reads of quad q+1 are issued
        // before the MFMAs of quad q, so the matrix pipe does not wait for LDS
        double af[2][TM], bf[2][TN];
        auto load_frag = [&](int kq, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = a_base[i * 16 * LD + 2 * kq];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = b_base[j * 16 * LD + 2 * kq];
        };
        load_frag(0, 0);
#pragma unroll
        for (int kq = 0; kq < BK / 2; ++kq) {
            if (kq + 1 < BK / 2) load_frag(kq + 1, (kq + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            double bs[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bs[j] = negate ? -bf[kq & 1][j] : bf[kq & 1][j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kq & 1][i], bs[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kofs_mine) kofs_commit(kt + 3);
        __syncthreads();
    }

    // D: col = lane & 15 (complex column = col >> 1, part = col & 1), row = (lane >> 4) + 4 * reg
    const double alpha = step_alpha(p);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn * TN * 8 + j * 8 + (i16 >> 1);
        if (n >= p.N) continue;
        const int64_t ncol = p.nC[n];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = wm * TM * 16 + i * 16 + k4 + 4 * t;
                const int64_t ro = rowC_s[row];
                if (ro >= 0) C[2 * (ro + ncol) + cc] = acc[i][j][t] * alpha;
            }
    }
}

template <int TM, int TN>
static hipError_t launch_c128_t(const StepArgs& p, int flags, hipStream_t stream) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 8;
    const int64_t tiles_m = (p.R + BM - 1) / BM;
    const int64_t tiles_n = (p.N + BN - 1) / BN;
    const int64_t gx = ((tiles_m + 7) / 8) * 8 * tiles_n;
    if (gx > 0x7fffffffll || p.Bt > 65535 || p.nz > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((pair_mfma_c128_kernel<TM, TN>), dim3((unsigned)gx, (unsigned)p.nz, (unsigned)p.Bt),
                       dim3(256), 0, stream, p, flags, tiles_m, tiles_n);
    return hipGetLastError();
}

// flags: bit0 = A's fastest-varying memory index is a contracted one, bit1 = same for B
hipError_t launch_pair_mfma_c128(const StepArgs& p, int flags, hipStream_t stream) {
    // the tall tile once its blocks alone fill the chip twice over
    const int64_t tall = ((p.R + 127) / 128) * ((p.N + 31) / 32) * p.Bt;
    if (tall >= 1024) return launch_c128_t<4, 2>(p, flags, stream);
    return launch_c128_t<2, 2>(p, flags, stream);
}

}  // namespace ctg

// ------------------------------------------------------------------------- //
// Real-valued steps (float32 / float64) on the matrix cores: both dtypes have
// a 16x16x4 MFMA with the same A/B fragment layout (A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15]); only the D layout differs (f32: row = 4*(l>>4)
// + reg, f64: row = (l>>4) + 4*reg).  Block tile 64 x 64, 16 k per step,
// 2 x 2 waves with 2 x 2 MFMA tiles each; the reference's `benchmark()`
// defaults to float64 (core.py:4094).
// ------------------------------------------------------------------------- //

namespace ctg {

typedef float f32x4r __attribute__((ext_vector_type(4)));

namespace {
constexpr int RBK = 16, RLD = RBK + 1;

__device__ __forceinline__ f32x4r mfma16(float a, float b, f32x4r c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4r type; };
template <> struct Vec4<double> { typedef f64x4 type; };
// row of accumulator register t inside a 16x16 tile
__device__ __forceinline__ int d_row(float, int k4, int t) { return 4 * k4 + t; }
__device__ __forceinline__ int d_row(double, int k4, int t) { return k4 + 4 * t; }
}  // namespace

// TM x TN MFMA tiles (16 x 16) per wave, 2 x 2 waves per block: 64 x 64 (TM = TN =
// 2) for small steps, 128 x 128 (TM = TN = 4) when the output alone fills the
// chip -- four times the matrix work per k-step and barrier, twice per gathered
// element and fragment read.
// VEC: both operands are gathered in 16-byte pieces (4 floats / 2 doubles) along their
// fastest-varying index -- the host has checked that every such piece is contiguous and
// aligned in memory (ctg_runtime.hip: real_vec_ok).  One load instruction then moves what four
// (two) did: the element-wise gather, not the matrix cores, is what held float32 at 0.47 of
// its peak (profiles/r2_f64_kernel_rates.txt).
template <typename T, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256, 2) void pair_mfma_real_kernel(StepArgs p, int flags, int64_t tiles_m,
                                                               int64_t tiles_n) {
    constexpr int RBM = 2 * TM * 16, RBN = 2 * TN * 16;
    constexpr int V = VEC ? 16 / (int)sizeof(T) : 1;   // elements per load
    constexpr int NA = RBM * RBK / 256 / V, NB = RBN * RBK / 256 / V;   // loads per thread and k-step
    typedef typename Vec4<T>::type V4;
    typedef T VT __attribute__((ext_vector_type(V > 1 ? V : 2)));
    __shared__ T lds[2 * (RBM + RBN) * RLD];
    __shared__ int64_t rowA_s[RBM];
    __shared__ int64_t rowC_s[RBM];
    __shared__ int64_t kofs_s[3][2][RBK];
    static_assert(RBM <= 256, "row offsets are resolved by one thread per row");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, q = bid >> 3;
    const int64_t tm = (q / tiles_n) * 8 + xcd;
    const int64_t tn = q % tiles_n;
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * RBM, n0 = tn * RBN;
    const int64_t bz = blockIdx.z;

    const T* __restrict__ A = (const T*)p.A + zoffA(p) + p.bA[bz];
    const T* __restrict__ B = (const T*)p.B + zoffB(p) + p.bB[bz];
    T* __restrict__ C = (T*)p.C + zoffC(p) + p.bC[bz];
    const bool a_kfast = flags & 1, b_kfast = flags & 2;
    const int64_t nk = (p.K + RBK - 1) / RBK;

    if (tid < RBM) {
        const int64_t m = m0 + tid;
        int64_t oa = -1, oc = -1;
        if (m < p.R) {
            int64_t hi, lo;
            split_row(p, m, hi, lo);
            oa = p.rowA.hi[hi] + p.rowA.lo[lo];
            oc = p.rowC.hi[hi] + p.rowC.lo[lo];
        }
        rowA_s[tid] = oa;
        rowC_s[tid] = oc;
    }
    int64_t kofs_val = -1;
    auto kofs_fetch = [&](int64_t step) {  // threads tid < 2*RBK
        const int which = tid / RBK, c = tid % RBK;
        const int64_t k = step * RBK + c;
        int64_t off = -1;
        if (k < p.K) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            off = which ? p.kB.hi[kh] + p.kB.lo[kl] : p.kA.hi[kh] + p.kA.lo[kl];
        }
        kofs_val = off;
    };
    auto kofs_commit = [&](int64_t step) { kofs_s[step % 3][tid / RBK][tid % RBK] = kofs_val; };
    if (tid < 2 * RBK) {
        for (int s = 0; s < 3 && s < nk; ++s) {
            kofs_fetch(s);
            kofs_commit(s);
        }
    }

    int a_r[NA], a_c[NA], b_k[NB], b_n[NB];
    int64_t b_col[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = (j * 256 + tid) * V;
        if (a_kfast) {
            a_r[j] = e / RBK;
            a_c[j] = e % RBK;
        } else {
            a_r[j] = e % RBM;
            a_c[j] = e / RBM;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int e = (j * 256 + tid) * V;
        if (b_kfast) {
            b_k[j] = e % RBK;
            b_n[j] = e / RBK;
        } else {
            b_k[j] = e / RBN;
            b_n[j] = e % RBN;
        }
        b_col[j] = (n0 + b_n[j] < p.N) ? p.nB[n0 + b_n[j]] : -1;
    }
    __syncthreads();
    int64_t a_row[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) a_row[j] = rowA_s[a_r[j]];

    // (VEC: piece j of a thread starts at tile element (a_r, a_c) and runs along k when k is
    // the operand's fast index, along the rows / columns otherwise; rows, columns and k's
    // are valid or not in whole pieces: the host requires R, N, K to be multiples of V)
    VT a_reg[NA], b_reg[NB];
    auto gather = [&](int64_t step) {
        const int64_t* ka = kofs_s[step % 3][0];
        const int64_t* kb = kofs_s[step % 3][1];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int64_t ko = ka[a_c[j]];
            VT v = 0;
#ifdef CTG_REAL_KO_GATHER   // (knock-out builds, tools/exp_f32_ko.sh: wrong results by construction)
            v[0] = (T)(a_row[j] + ko);
#else
            if (a_row[j] >= 0 && ko >= 0) {
                if constexpr (VEC) v = *(const VT*)(A + a_row[j] + ko);
                else v[0] = A[a_row[j] + ko];
            }
#endif
            a_reg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int64_t ko = kb[b_k[j]];
            VT v = 0;
#ifdef CTG_REAL_KO_GATHER
            v[0] = (T)(b_col[j] + ko);
#else
            if (b_col[j] >= 0 && ko >= 0) {
                if constexpr (VEC) v = *(const VT*)(B + b_col[j] + ko);
                else v[0] = B[b_col[j] + ko];
            }
#endif
            b_reg[j] = v;
        }
    };
    auto stage = [&](int buf) {
#ifdef CTG_REAL_KO_STAGE
        if (a_reg[0][0] != (T)12345.678f) return;
#endif
        T* As = lds + buf * (RBM + RBN) * RLD;
        T* Bs = As + RBM * RLD;
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i)
                As[(a_r[j] + (a_kfast ? 0 : i)) * RLD + a_c[j] + (a_kfast ? i : 0)] = a_reg[j][i];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i)
                Bs[(b_n[j] + (b_kfast ? 0 : i)) * RLD + b_k[j] + (b_kfast ? i : 0)] = b_reg[j][i];
    };

    // float32 with an even number of 16-row / 16-column tiles per wave runs on the 32x32x2
    // instruction: it issues at 155 TFLOP/s on this hardware, the 16x16x4 form at 139
    // (profiles/r2_mfma_issue_rates.txt); float64 only has the 16x16x4 form.
    constexpr bool WIDE = std::is_same<T, float>::value && TM % 2 == 0 && TN % 2 == 0;
    if constexpr (WIDE) {
        constexpr int WM = TM / 2, WN = TN / 2;   // 32x32 tiles per wave
        typedef float f32x16r __attribute__((ext_vector_type(16)));
        f32x16r acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
        const int l31 = lane & 31, kk = lane >> 5;   // A[row = l31][k = kk], B[k = kk][col = l31]

        gather(0);
        stage(0);
        if (nk > 1) gather(1);
        __syncthreads();

        for (int64_t kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(buf ^ 1);
            const bool kofs_mine = kt + 3 < nk && tid < 2 * RBK;
            if (kofs_mine) kofs_fetch(kt + 3);
            if (kt + 2 < nk) gather(kt + 2);

            const float* As = (const float*)lds + buf * (RBM + RBN) * RLD;
            const float* Bs = As + RBM * RLD;
            const float* a_base = As + (wm * TM * 16 + l31) * RLD + kk;
            const float* b_base = Bs + (wn * TN * 16 + l31) * RLD + kk;
            float af[2][WM], bf[2][WN];
            auto load_frag = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) af[slot][i] = a_base[i * 32 * RLD + 2 * ks];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[slot][j] = b_base[j * 32 * RLD + 2 * ks];
            };
            load_frag(0, 0);
#pragma unroll
            for (int ks = 0; ks < RBK / 2; ++ks) {
                if (ks + 1 < RBK / 2) load_frag(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kofs_mine) kofs_commit(kt + 3);
            __syncthreads();
        }

        // D of the 32x32 tile: column l31, rows (t & 3) + 8 * (t >> 2) + 4 * kk
        const float alpha = (float)step_alpha(p);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int64_t n = n0 + wn * TN * 16 + j * 32 + l31;
            if (n >= p.N) continue;
            const int64_t ncol = p.nC[n];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int64_t ro = rowC_s[wm * TM * 16 + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk];
                    if (ro >= 0) ((float*)C)[ro + ncol] = acc[i][j][t] * alpha;
                }
        }
        return;
    } else {
    V4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][j][t] = 0;

    const int i16 = lane & 15, k4 = lane >> 4;

    gather(0);
    stage(0);
    if (nk > 1) gather(1);
    __syncthreads();

    for (int64_t kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1);
        const bool kofs_mine = kt + 3 < nk && tid < 2 * RBK;
        if (kofs_mine) kofs_fetch(kt + 3);
        if (kt + 2 < nk) gather(kt + 2);

        const T* As = lds + buf * (RBM + RBN) * RLD;
        const T* Bs = As + RBM * RLD;
        const T* a_base = As + (wm * TM * 16 + i16) * RLD + k4;
        const T* b_base = Bs + (wn * TN * 16 + i16) * RLD + k4;
        T af[2][TM], bf[2][TN];
        auto load_frag = [&](int kq, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = a_base[i * 16 * RLD + 4 * kq];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = b_base[j * 16 * RLD + 4 * kq];
        };
        load_frag(0, 0);
#pragma unroll
        for (int kq = 0; kq < RBK / 4; ++kq) {
            if (kq + 1 < RBK / 4) load_frag(kq + 1, (kq + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(af[kq & 1][i], bf[kq & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kofs_mine) kofs_commit(kt + 3);
        __syncthreads();
    }

    const T alpha = (T)step_alpha(p);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn * TN * 16 + j * 16 + i16;
        if (n >= p.N) continue;
        const int64_t ncol = p.nC[n];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t ro = rowC_s[wm * TM * 16 + i * 16 + d_row(T(0), k4, t)];
                if (ro >= 0) C[ro + ncol] = acc[i][j][t] * alpha;
            }
    }
    }
}

template <typename T, int TM, int TN>
static hipError_t launch_real_t(const StepArgs& p, int flags, hipStream_t stream) {
    const bool vec = (flags & 4) && (flags & 8);   // both operands gather in 16-byte pieces
    constexpr int RBM = 2 * TM * 16, RBN = 2 * TN * 16;
    const int64_t tiles_m = (p.R + RBM - 1) / RBM;
    const int64_t tiles_n = (p.N + RBN - 1) / RBN;
    const int64_t gx = ((tiles_m + 7) / 8) * 8 * tiles_n;
    if (gx > 0x7fffffffll || p.Bt > 65535 || p.nz > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)gx, (unsigned)p.nz, (unsigned)p.Bt);
    if (vec)
        hipLaunchKernelGGL((pair_mfma_real_kernel<T, TM, TN, true>), grid, dim3(256), 0, stream, p, flags,
                           tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((pair_mfma_real_kernel<T, TM, TN, false>), grid, dim3(256), 0, stream, p, flags,
                           tiles_m, tiles_n);
    return hipGetLastError();
}

hipError_t launch_pair_mfma_real(int dtype, const StepArgs& p, int flags, hipStream_t stream) {
    if (dtype != 0 && dtype != 1) return hipErrorInvalidValue;
    // the large tile once its blocks alone fill the chip twice over
    const int64_t big = ((p.R + 127) / 128) * ((p.N + 127) / 128) * p.Bt;
    const bool wide = big >= 1024 && p.N >= 96;
    const int64_t tall = ((p.R + 127) / 128) * ((p.N + 63) / 64) * p.Bt;
    if (dtype == 0) {
        if (wide) return launch_real_t<float, 4, 4>(p, flags, stream);
        if (tall >= 1024) return launch_real_t<float, 4, 2>(p, flags, stream);
        return launch_real_t<float, 2, 2>(p, flags, stream);
    }
    // (128 x 128 in double precision would need more than 256 registers per lane)
    if (tall >= 1024) return launch_real_t<double, 4, 2>(p, flags, stream);
    return launch_real_t<double, 2, 2>(p, flags, stream);
}

}  // namespace ctg
