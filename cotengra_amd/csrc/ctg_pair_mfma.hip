// ctg_pair_mfma.hip -- complex64 gather-GEMM on the gfx950 fp32 matrix cores.
//
//   C[bC(b) + rowC(m) + nC(n)] = sum_k A[bA(b) + rowA(m) + kA(k)] * B[bB(b) + kB(k) + nB(n)]
//
// This one kernel replaces the reference's whole pairwise lowering
// `transpose -> reshape -> matmul -> reshape -> transpose`
// (cotengra/contract.py:364-411): the operand permutations are folded into
// offset tables and realised in the global->LDS gather, so no operand is ever
// copied just to be re-laid-out.
//
// Complex arithmetic on a real MFMA without wasted flops.  gfx950 has no
// complex MFMA; v_mfma_f32_32x32x2_f32 computes D(32x32) += A'(32x2) B'(2x32)
// in exact fp32.  For one complex k we feed
//     A'[i][0] = Re a_ik            A'[i][1] = Im a_ik
//     B'[0][2n] = Re b_kn  B'[0][2n+1] = Im b_kn
//     B'[1][2n] = -Im b_kn B'[1][2n+1] = Re b_kn
// so D[i][2n] / D[i][2n+1] accumulate Re / Im of sum_k a_ik b_kn: one
// instruction = 32 rows x 16 complex columns x 1 complex k = 4096 real flops,
// all of them useful (8 flops per complex multiply-add), and D is already in
// interleaved complex layout for the store.
//
// Tiling: 256 threads = 4 waves (64 lanes each), block tile BM x BN complex,
// BK complex per k-step, register-staged double buffering through LDS:
// global gathers of step t+1 are in flight while the MFMAs of step t run.
// LDS holds A as two planes (re, im) [BM][BK+4] and B transposed
// [2*BN][BK+4] so every fragment is one aligned ds_read_b128 (4 k's).
//
// Coalescing of the gather.  Which tile element a lane fetches is given by a
// per-step *order table* built on the host (ctg_runtime.hip): the BM x BK
// tile elements sorted by their memory offset.  Lane l of load instruction j
// takes sorted element j*256 + l, so a wave always walks memory in ascending
// address order whatever index permutation the step encodes -- the HBM-side
// equivalent of the LDS-staged transpose.  When adjacent sorted elements are
// adjacent in memory (always the case for the big operand of power-of-two
// networks) each lane moves two complex numbers with one 16-byte load.
// Offsets come from LDS, not from global tables: row offsets are resolved once
// per tile, k offsets by 2*BK threads one k-step ahead (3-slot ring).
//
// The block-id -> tile map (map_tile) hands 8x8 patches of tiles to one XCD at a
// time (block b runs on XCD b % 8) so A and B k-slices are shared through that
// XCD's L2 instead of being re-fetched per tile.
#include "ctg_common.h"

#include <type_traits>

namespace ctg {

#ifdef CTG_TIMING
// experiment build only (tools/exp_timing.py): wall-clock phases of the fast
// kernel, summed over blocks -- [0] constants, [1] first gather + stage, [2] k
// loop, [3] epilogue issue, [4] store drain, [5] number of blocks
__device__ unsigned long long ctg_timing[8];
#define CTG_STAMP(name) const unsigned long long name = wall_clock64()
#else
#define CTG_STAMP(name)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Knock-out switches of experiment builds (tools/exp_knockout.sh; results are
// wrong by construction): what does a kernel cost without its stores / matrix
// instructions / LDS transpose / table-driven prologue?  Off in the product.
#ifdef CTG_KO_STORE
#define CTG_STORE_GUARD(alpha) if ((alpha) == 12345.678f)
#else
#define CTG_STORE_GUARD(alpha)
#endif
#ifdef CTG_KO_MFMA
// keeps the data dependences (fragments are consumed) at one VALU op per MFMA
__device__ __forceinline__ f32x16 ko_mfma(float a, float b, f32x16 c) {
    c[0] = fmaf(a, b, c[0]);
    return c;
}
#define CTG_MFMA(a, b, c) ko_mfma(a, b, c)
#else
#define CTG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct MfmaCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int kThreads = 256;
    static constexpr int LD = BK + 4;    // padded k extent of an LDS row (floats)
    static constexpr int WTM = BM / WM;  // wave tile rows
    static constexpr int WTN = BN / WN;  // wave tile complex cols
    static constexpr int FM = WTM / 32;  // MFMA tiles per wave along m
    static constexpr int FN = WTN / 16;  // MFMA tiles per wave along n (16 complex = 32 real)
    static constexpr int A_PER_T = BM * BK / kThreads;
    static constexpr int B_PER_T = (BK * BN + kThreads - 1) / kThreads;
    static constexpr int A_FLOATS = 2 * BM * LD;
    static constexpr int B_FLOATS = 2 * BN * LD;
    // blocks per CU the fast kernel is compiled for (LDS: 2*(BM+BN)*2*BK*4 bytes)
    static constexpr int FAST_BLOCKS = BN >= 128 ? 2 : 3;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(WTM % 32 == 0 && WTN % 16 == 0, "wave tile must hold whole MFMA tiles");
    static_assert((BM * BK) % (2 * kThreads) == 0, "A tile must divide over the block in pairs");
    static_assert(BK == MFMA_BK && BM == MFMA_BM, "tile constants shared with the host");
    static_assert(BK % 4 == 0, "k-step is consumed 4 at a time");
};


// XCD-aware work assignment of the tiled kernels.  Block b runs on XCD b % 8
// (each XCD has its own 4 MB L2).  A unit is a (row tile, k-split) pair.  The
// (unit, column tile) grid is cut into patches of PM x PN = 64 tiles; a patch
// is handed to ONE XCD as 64 consecutive blocks, which run concurrently and
// step through K roughly together -- so each A k-slice is fetched once per PN
// column tiles and each B k-slice once per PM row tiles from that XCD's L2
// instead of once per tile from HBM / Infinity Cache.  Consecutive patches go
// to different XCDs (also spreads the k-splits of a single row tile).
struct TileMap {
    int64_t unit, tn;
    bool valid;
};
__device__ __forceinline__ void patch_shape(int64_t tiles_n, int& PM, int& PN) {
    PN = tiles_n >= 8 ? 8 : (tiles_n >= 4 ? 4 : (tiles_n >= 2 ? 2 : 1));
    PM = 64 / PN;
}
__device__ __forceinline__ TileMap map_tile(int64_t bid, int64_t units, int64_t tiles_n) {
    int PM, PN;
    patch_shape(tiles_n, PM, PN);
    const int64_t npn = (tiles_n + PN - 1) / PN;
    const int64_t xcd = bid & 7, q = bid >> 3;
    const int64_t pid = (q >> 6) * 8 + xcd;  // patch index
    const int within = (int)(q & 63);
    const int64_t pm = pid / npn, pn = pid - pm * npn;
    TileMap t;
    t.unit = pm * PM + within / PN;
    t.tn = pn * PN + within % PN;
    t.valid = t.unit < units && t.tn < tiles_n;
    return t;
}
static int64_t tile_grid_blocks(int64_t units, int64_t tiles_n) {
    const int PN = tiles_n >= 8 ? 8 : (tiles_n >= 4 ? 4 : (tiles_n >= 2 ? 2 : 1));
    const int PM = 64 / PN;
    const int64_t patches = ((units + PM - 1) / PM) * ((tiles_n + PN - 1) / PN);
    return ((patches + 7) / 8) * 8 * 64;
}

template <typename Cfg, bool VEC_A>
__global__ __launch_bounds__(256, 2) void pair_mfma_c64_kernel(StepArgs p, MfmaHints h,
                                                               int64_t tiles_m, int64_t tiles_n,
                                                               int64_t k_chunk,
                                                               float* __restrict__ partial) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, WN = Cfg::WN, LD = Cfg::LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * (Cfg::A_FLOATS + Cfg::B_FLOATS)];
    __shared__ int64_t rowA_s[BM];
    __shared__ int64_t rowC_s[BM];
    __shared__ int64_t kofs_s[3][2][BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int64_t S_split = k_chunk;  // number of k-splits (1 = none)
    // (slices of a batch rotate the XCD their patches go to: a small step whose tiles
    // make up a single patch would otherwise put every slice on XCD 0)
    const TileMap tmap = map_tile((blockIdx.x & ~7u) | ((blockIdx.x + blockIdx.y) & 7u), tiles_m * S_split, tiles_n);
    if (!tmap.valid) return;
    const int64_t tn = tmap.tn;
    const int64_t tm = tmap.unit % tiles_m;
    const int64_t ksplit = tmap.unit / tiles_m;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int64_t bz = blockIdx.z;

    const c64* __restrict__ A = (const c64*)p.A + zoffA(p) + p.bA[bz];
    const c64* __restrict__ B = (const c64*)p.B + zoffB(p) + p.bB[bz];
    float* __restrict__ C = (float*)((c64*)p.C + zoffC(p) + p.bC[bz]);

    // split-K: the k-steps are dealt out cyclically -- block y of S takes steps
    // y, y+S, y+2S, ... -- so that the blocks running at the same time sweep
    // one contiguous window of K together (streaming-like page / DRAM-row
    // locality; contiguous chunks per block thrash address translation when
    // the chunks are megabytes apart).  Partial tiles go to scratch.
    const int64_t nk_total = (p.K + BK - 1) / BK;
    const int64_t nk = (nk_total - ksplit + S_split - 1) / S_split;

    // --- offsets into LDS ---------------------------------------------------
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
    // k offsets of a future step: fetched from the global tables into a register
    // (kofs_fetch) early in an iteration and written to the LDS ring
    // (kofs_commit) only after that iteration's MFMAs, so the table-load latency
    // never sits on the critical path of the wave that does it.
    int64_t kofs_val = -1;
    auto kofs_fetch = [&](int64_t step) {  // executed by threads tid < 2*BK
        const int which = tid / BK, c = tid % BK;
        const int64_t k = (step * S_split + ksplit) * BK + c;
        int64_t off = -1;
        if (k < p.K) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            off = which ? p.kB.hi[kh] + p.kB.lo[kl] : p.kA.hi[kh] + p.kA.lo[kl];
        }
        kofs_val = off;
    };
    auto kofs_commit = [&](int64_t step) { kofs_s[step % 3][tid / BK][tid % BK] = kofs_val; };
    auto fill_kofs = [&](int64_t step) {
        kofs_fetch(step);
        kofs_commit(step);
    };
    if (tid < 2 * BK) {
        fill_kofs(0);
        if (nk > 1) fill_kofs(1);
        if (nk > 2) fill_kofs(2);
    }

    // --- per-thread gather coordinates (sorted-by-address order tables) ------
    int a_lds[Cfg::A_PER_T], a_c[Cfg::A_PER_T], a_r[Cfg::A_PER_T];
    {
        const uint16_t* oa = h.ordA + tid * Cfg::A_PER_T;
#pragma unroll
        for (int j = 0; j < Cfg::A_PER_T; ++j) {
            const int v = oa[j];
            a_r[j] = v >> 4;
            a_c[j] = v & 15;
            a_lds[j] = a_r[j] * LD + a_c[j];
        }
    }
    int b_k[Cfg::B_PER_T], b_lds[Cfg::B_PER_T];
    int64_t b_col[Cfg::B_PER_T];
    {
        const uint16_t* ob = h.ordB + tid * Cfg::B_PER_T;
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) {
            const int v = ob[j];
            const int nn = v >> 4;
            b_k[j] = v & 15;
            b_lds[j] = 2 * nn * LD + b_k[j];
            const int64_t n = n0 + nn;
            b_col[j] = (j * 256 + tid < BK * BN && n < p.N) ? p.nB[n] : -1;
        }
    }
    __syncthreads();
    int64_t a_row[Cfg::A_PER_T];
#pragma unroll
    for (int j = 0; j < Cfg::A_PER_T; ++j) a_row[j] = rowA_s[a_r[j]];

    c64 a_reg[Cfg::A_PER_T], b_reg[Cfg::B_PER_T];

    auto gather = [&](int64_t step) {
        const int64_t* ka = kofs_s[step % 3][0];
        const int64_t* kb = kofs_s[step % 3][1];
        if (VEC_A) {
            // sorted elements (2j, 2j+1) are adjacent in memory and the tile is full
#pragma unroll
            for (int j = 0; j < Cfg::A_PER_T; j += 2) {
                const f32x4 v = *(const f32x4*)(A + a_row[j] + ka[a_c[j]]);
                a_reg[j] = c64{v[0], v[1]};
                a_reg[j + 1] = c64{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < Cfg::A_PER_T; ++j) {
                const int64_t ko = ka[a_c[j]];
                c64 v{0.f, 0.f};
                if (a_row[j] >= 0 && ko >= 0) v = A[a_row[j] + ko];
                a_reg[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) {
            const int64_t ko = kb[b_k[j]];
            c64 v{0.f, 0.f};
            if (b_col[j] >= 0 && ko >= 0) v = B[b_col[j] + ko];
            b_reg[j] = v;
        }
    };
    auto stage = [&](int buf) {
        float* As = lds + buf * (Cfg::A_FLOATS + Cfg::B_FLOATS);
        float* Bs = As + Cfg::A_FLOATS;
#pragma unroll
        for (int j = 0; j < Cfg::A_PER_T; ++j) {
            As[a_lds[j]] = a_reg[j].re;
            As[BM * LD + a_lds[j]] = a_reg[j].im;
        }
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) {
            if (j * 256 + tid < BK * BN) {
                Bs[b_lds[j]] = b_reg[j].re;
                Bs[b_lds[j] + LD] = b_reg[j].im;
            }
        }
    };

    f32x16 acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

    const int kk = lane >> 5;  // 0: real part of a / first row of B', 1: imaginary part
    const int l31 = lane & 31;
    const bool negate = kk == 1 && (lane & 1) == 0;

    // software pipeline: at the top of iteration kt the registers already hold
    // tile kt+1 (gathered a full iteration ago, so no wait), it is written to
    // the other LDS buffer and the gather of tile kt+2 is issued -- both
    // asynchronous to the MFMAs of tile kt that follow.
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

        const float* As = lds + buf * (Cfg::A_FLOATS + Cfg::B_FLOATS);
        const float* Bs = As + Cfg::A_FLOATS;
        const float* a_base = As + kk * BM * LD + (wm * Cfg::WTM + l31) * LD;
        const float* b_base = Bs + (2 * wn * Cfg::WTN + (l31 ^ kk)) * LD;
#pragma unroll
        for (int kq = 0; kq < BK / 4; ++kq) {
            f32x4 af[Cfg::FM], bf[Cfg::FN];
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i) af[i] = *(const f32x4*)(a_base + i * 32 * LD + kq * 4);
#pragma unroll
            for (int j = 0; j < Cfg::FN; ++j) {
                f32x4 v = *(const f32x4*)(b_base + j * 32 * LD + kq * 4);
                bf[j] = negate ? -v : v;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t],
                                                                        acc[i][j], 0, 0, 0);
        }
        if (kofs_mine) kofs_commit(kt + 3);
        __syncthreads();
    }

    // --- epilogue: D[row][2n+c], row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ------
    if (partial != nullptr) {
        // dense fp32 slab [batch][split][tiles_m*BM][2*tiles_n*BN]
        const int64_t ldp = 2 * tiles_n * BN;
        // (slice-in-batch z owns its own set of slabs: gridDim.z * S_split of them)
        // (scratch is divided among the slices of THIS launch: blockIdx.y, not z0 + blockIdx.y)
        float* slab = partial + ((((int64_t)blockIdx.y * gridDim.z + bz) * S_split + ksplit) * (tiles_m * BM)) * ldp;
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j) {
            const int64_t col = 2 * (n0 + wn * Cfg::WTN + j * 16) + l31;
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int64_t row = m0 + wm * Cfg::WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk;
                    slab[row * ldp + col] = acc[i][j][t];
                }
        }
        return;
    }
    // pair up rows (t, t+1): even lanes end up with (re, im) of row t, odd lanes
    // with (re, im) of row t+1, so every lane stores one whole complex number
    const bool odd = lane & 1;
    const float alpha = (float)step_alpha(p);
#pragma unroll
    for (int j = 0; j < Cfg::FN; ++j) {
        const int64_t n = n0 + wn * Cfg::WTN + j * 16 + (l31 >> 1);
        const bool n_ok = n < p.N;
        const int64_t ncol = n_ok ? p.nC[n] : 0;
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i) {
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const float2 v = pair_rows(acc[i][j][t], acc[i][j][t + 1], odd, alpha);
                const int row = wm * Cfg::WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk + (odd ? 1 : 0);
                const int64_t ro = rowC_s[row];
                if (n_ok && ro >= 0) *(float2*)(C + 2 * (ro + ncol)) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------- //
// Fast path of the tiled kernel for the steps that carry the flops of a
// power-of-two network: full tiles and tile-additive offset tables (checked on
// the host, MfmaHints::fast).  Every address is then
//     [uniform 64-bit base in SGPRs] + [per-lane 32-bit constant]
// -- the per-lane constants are computed once per block, the bases once per
// tile / k-step with scalar loads -- so the inner loop carries no 64-bit
// vector arithmetic, no LDS offset lookups, no bounds checks and no branches:
// 4 x 16-byte A loads + B loads + LDS traffic + 64 MFMAs per wave per k-step.
// ------------------------------------------------------------------------- //

// Per-lane constants of the fast kernel (see its prologue): which tile elements a
// thread gathers (offsets relative to the tile bases) and where it stages them
// in LDS.  `compute` derives them from the order and offset tables; the builder
// kernel below runs it once per step and executor, the tiled kernel then reads
// the packed words back.
template <typename Cfg, bool VEC_A>
struct FastLane {
    static constexpr int LD = Cfg::BK;
    static constexpr int NA = VEC_A ? Cfg::A_PER_T / 2 : Cfg::A_PER_T;
    static constexpr int NAL = (Cfg::A_PER_T + 1) / 2, NBL = (Cfg::B_PER_T + 1) / 2;
    static constexpr int NW = NA + NAL + Cfg::B_PER_T + NBL;   // 32-bit words per thread
    static constexpr int NQ = (NW + 3) / 4;                    // 16-byte loads per thread
    __device__ static __forceinline__ int fsw(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }
    __device__ static __forceinline__ int swz(int row, int c) {
        return row * LD + (((c >> 1) ^ fsw(row)) << 1) + (c & 1);
    }
    __device__ static __forceinline__ void compute(const StepArgs& p, const MfmaHints& h, int tid,
                                                   unsigned (&a_off)[NA], unsigned (&a_lds)[NAL],
                                                   unsigned (&b_off)[Cfg::B_PER_T], unsigned (&b_lds)[NBL]) {
        const uint16_t* oa = h.ordA + tid * Cfg::A_PER_T;
#pragma unroll
        for (int j = 0; j < Cfg::A_PER_T; ++j) {
#ifdef CTG_KO_PRO
            const int v = ((tid * Cfg::A_PER_T + j) * 37) & 0x7ff;   // no table loads at all
#else
            const int v = oa[j];
#endif
            const int r = v >> 4, c = v & 15;
            if (j & 1) a_lds[j / 2] |= (unsigned)swz(r, c) << 16;
            else a_lds[j / 2] = (unsigned)swz(r, c);
            if (!VEC_A || (j & 1) == 0)
#ifdef CTG_KO_PRO
                a_off[VEC_A ? j / 2 : j] = (unsigned)((tid * NA + (VEC_A ? j / 2 : j)) * 2);
#else
                a_off[VEC_A ? j / 2 : j] = (unsigned)(p.rowA.lo[r] + p.kA.lo[c]);
#endif
        }
        const uint16_t* ob = h.ordB + tid * Cfg::B_PER_T;
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) {
#ifdef CTG_KO_PRO
            const int v = ((tid * Cfg::B_PER_T + j) * 29) & (16 * Cfg::BN - 1);
#else
            const int v = ob[j];
#endif
            const int nn = v >> 4, c = v & 15;
            if (j & 1) b_lds[j / 2] |= (unsigned)swz(2 * nn, c) << 16;
            else b_lds[j / 2] = (unsigned)swz(2 * nn, c);
#ifdef CTG_KO_PRO
            b_off[j] = (unsigned)(tid * Cfg::B_PER_T + j);
#else
            b_off[j] = (unsigned)(p.nB[nn] + p.kB.lo[c]);
#endif
        }
    }
    __device__ static __forceinline__ void pack_words(unsigned (&w)[NQ * 4], const unsigned (&a_off)[NA],
                                                      const unsigned (&a_lds)[NAL],
                                                      const unsigned (&b_off)[Cfg::B_PER_T],
                                                      const unsigned (&b_lds)[NBL]) {
        int i = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) w[i++] = a_off[j];
#pragma unroll
        for (int j = 0; j < NAL; ++j) w[i++] = a_lds[j];
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) w[i++] = b_off[j];
#pragma unroll
        for (int j = 0; j < NBL; ++j) w[i++] = b_lds[j];
#pragma unroll
        for (; i < NQ * 4; ++i) w[i] = 0u;
    }
    __device__ static __forceinline__ void unpack_words(const unsigned (&w)[NQ * 4], unsigned (&a_off)[NA],
                                                        unsigned (&a_lds)[NAL],
                                                        unsigned (&b_off)[Cfg::B_PER_T],
                                                        unsigned (&b_lds)[NBL]) {
        int i = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) a_off[j] = w[i++];
#pragma unroll
        for (int j = 0; j < NAL; ++j) a_lds[j] = w[i++];
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) b_off[j] = w[i++];
#pragma unroll
        for (int j = 0; j < NBL; ++j) b_lds[j] = w[i++];
    }
};

// one block of 256 threads: out[q * 256 + tid] = the q-th 16 bytes of thread tid
template <typename Cfg, bool VEC_A>
__global__ __launch_bounds__(256) void fast_lane_consts_kernel(StepArgs p, MfmaHints h, uint4* out) {
    typedef FastLane<Cfg, VEC_A> Lane;
    const int tid = threadIdx.x;
    unsigned a_off[Lane::NA], a_lds[Lane::NAL], b_off[Cfg::B_PER_T], b_lds[Lane::NBL];
    Lane::compute(p, h, tid, a_off, a_lds, b_off, b_lds);
    unsigned w[Lane::NQ * 4];
    Lane::pack_words(w, a_off, a_lds, b_off, b_lds);
#pragma unroll
    for (int q = 0; q < Lane::NQ; ++q) out[q * 256 + tid] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}

// GROUPED: one launch carries the tiles of several independent small steps (no
// k-split, no batch group); workgroups [block_begin, block_begin + n_blocks) of
// the grid belong to item i, whose step description replaces the kernel
// arguments.  The arithmetic of a tile is the same instruction sequence either
// way: grouping does not change a result bit.
template <typename Cfg, bool VEC_A, bool GROUPED = false>
__global__ __launch_bounds__(256, Cfg::FAST_BLOCKS) void pair_mfma_fast_kernel(StepArgs p_, MfmaHints h_,
                                                                int64_t tiles_m_, int64_t tiles_n_,
                                                                int64_t k_chunk,
                                                                float* __restrict__ partial,
                                                                const FastGroupItem* __restrict__ items,
                                                                int n_items) {
    StepArgs p;
    MfmaHints h;
    int64_t tiles_m, tiles_n;
    unsigned bx;
    if constexpr (GROUPED) {
        int gi = 0;
        while (gi + 1 < n_items && blockIdx.x >= items[gi + 1].block_begin) ++gi;
        gi = __builtin_amdgcn_readfirstlane(gi);
        p = items[gi].p;
        h = items[gi].h;
        tiles_m = items[gi].tiles_m;
        tiles_n = items[gi].tiles_n;
        bx = blockIdx.x - items[gi].block_begin;
    } else {
        p = p_;
        h = h_;
        tiles_m = tiles_m_;
        tiles_n = tiles_n_;
        bx = blockIdx.x;
    }
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, WN = Cfg::WN;
    constexpr int NA = VEC_A ? Cfg::A_PER_T / 2 : Cfg::A_PER_T;  // A load instructions per thread
    // Unpadded LDS rows of BK floats.  Fragments are read one k-pair (8 bytes)
    // at a time; the k-pair slot of a row is XORed with fsw(row), which makes
    // the 32 rows of a ds_read_b64 lane group (and the 16 rows of each half of
    // a ds_read2_b64) land on distinct 8-byte slots of the bank row
    // (conflict-free), and a block needs 48 KB (BN = 64) so three share a CU.
    constexpr int LD = BK;
    constexpr int AF = 2 * BM * LD, BF = 2 * BN * LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF)];
    auto fsw = [](int row) { return FastLane<Cfg, VEC_A>::fsw(row); };

    CTG_STAMP(T0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int64_t S_split = k_chunk;
    // (slices of a batch rotate the XCD their patches go to: a small step whose tiles
    // make up a single patch would otherwise put every slice on XCD 0)
    TileMap tmap;
    if constexpr (GROUPED) {
        // (a few tiles per step: plain row-major, consecutive workgroups go to different XCDs)
        tmap.unit = bx / (unsigned)tiles_n;
        tmap.tn = bx - (unsigned)tmap.unit * (unsigned)tiles_n;
        tmap.valid = true;
    } else {
        tmap = map_tile((bx & ~7u) | ((bx + blockIdx.y) & 7u), tiles_m * S_split, tiles_n);
    }
    if (!tmap.valid) return;
    const int64_t bz = blockIdx.z;
    // (64-bit divisions run on the vector ALU: results go back to scalar registers)
    const int64_t unit = uniform64(tmap.unit), tn = uniform64(tmap.tn);
    const int64_t ksplit = uniform64(unit / tiles_m);
    const int64_t tm = unit - ksplit * tiles_m;
    const int64_t m0 = tm * BM, n0 = tn * BN;

    // ---- uniform bases (scalar loads) --------------------------------------------
    int64_t rhi, rlo;
    split_row(p, m0, rhi, rlo);
    rhi = uniform64(rhi);
    rlo = uniform64(rlo);
    const c64* __restrict__ A = (const c64*)p.A + zoffA_s(p) + sload64(p.bA + bz) +
                                sload64(p.rowA.hi + rhi) + sload64(p.rowA.lo + rlo);
    const c64* __restrict__ B = (const c64*)p.B + zoffB_s(p) + sload64(p.bB + bz) + sload64(p.nB + n0);
    float* __restrict__ C = (float*)((c64*)p.C + zoffC_s(p) + sload64(p.bC + bz) +
                                     sload64(p.rowC.hi + rhi) + sload64(p.rowC.lo + rlo) + sload64(p.nC + n0));

    // split-K: cyclic distribution of the k-steps over the k-split units (see
    // the general kernel)
    const int64_t nk_total = p.K / BK;
    const int64_t nk = uniform64((nk_total - ksplit + S_split - 1) / S_split);

    // ---- per-lane constants -----------------------------------------------------
    // They depend on the thread and the step only, not on the tile: built once per
    // executor by fast_lane_consts_kernel into a table (h.lane) that every block
    // reads with NQ coalesced 16-byte loads -- instead of two levels of dependent
    // table lookups (order table -> offset tables) at the head of every tile.
    typedef FastLane<Cfg, VEC_A> Lane;
    unsigned a_off[NA];          // element offset of the load relative to the bases
    // LDS float offsets of the elements this lane stages, two 16-bit values per register
    constexpr int NAL = Lane::NAL, NBL = Lane::NBL;
    unsigned a_lds[NAL], b_lds[NBL];   // b: Re row 2n of B'; the Im row 2n+1 is LD floats further
    unsigned b_off[Cfg::B_PER_T];
#ifndef CTG_KO_PRO
    if (h.lane != nullptr) {
        unsigned w[Lane::NQ * 4];
        const uint4* L = (const uint4*)h.lane;
#pragma unroll
        for (int q = 0; q < Lane::NQ; ++q) {
            const uint4 v = L[q * 256 + tid];
            w[4 * q] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        Lane::unpack_words(w, a_off, a_lds, b_off, b_lds);
    } else
#endif
    {
        Lane::compute(p, h, tid, a_off, a_lds, b_off, b_lds);
    }
    auto unpack = [](const unsigned* pk, int j) { return (int)((j & 1) ? pk[j / 2] >> 16 : pk[j / 2] & 0xffffu); };

    CTG_STAMP(T1);
    c64 a_reg[Cfg::A_PER_T], b_reg[Cfg::B_PER_T];
    // uniform k offsets of the k-step being gathered: the four scalar loads are
    // issued one phase before their first use (the host only takes this path
    // when the k tables split at a power of two)
    int64_t kAh = 0, kAl = 0, kBh = 0, kBl = 0;
    auto k_bases = [&](int64_t step) {
        const int64_t k = uniform64((step * S_split + ksplit) * BK);
        const int64_t kh = k >> p.k_lo_shift, kl = k & (p.k_lo - 1);
        kAh = sload64(p.kA.hi + kh);
        kAl = sload64(p.kA.lo + kl);
        kBh = sload64(p.kB.hi + kh);
        kBl = sload64(p.kB.lo + kl);
    };
    auto gather_a = [&]() {
        const c64* Ak = A + kAh + kAl;
        if (VEC_A) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
#ifdef CTG_KO_GATHER
                const f32x4 v = {(float)a_off[j], 1.f, 2.f, (float)(uintptr_t)Ak};
#else
                const f32x4 v = *(const f32x4*)(Ak + a_off[j]);
#endif
                a_reg[2 * j] = c64{v[0], v[1]};
                a_reg[2 * j + 1] = c64{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) a_reg[j] = Ak[a_off[j]];
        }
    };
    auto gather_b = [&]() {
        const c64* Bk = B + kBh + kBl;
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j)
            if (Cfg::B_PER_T * 256 == BK * BN || j * 256 + tid < BK * BN) b_reg[j] = Bk[b_off[j]];
    };
    auto stage_a = [&](int buf) {
        float* As = lds + buf * (AF + BF);
#pragma unroll
        for (int j = 0; j < Cfg::A_PER_T; ++j) {
            const int o = unpack(a_lds, j);
            As[o] = a_reg[j].re;
            As[BM * LD + o] = a_reg[j].im;
        }
    };
    auto stage_b = [&](int buf) {
        float* Bs = lds + buf * (AF + BF) + AF;
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) {
            if (Cfg::B_PER_T * 256 == BK * BN || j * 256 + tid < BK * BN) {
                const int o = unpack(b_lds, j);
                Bs[o] = b_reg[j].re;
                Bs[o + LD] = b_reg[j].im;
            }
        }
    };

    f32x16 acc[Cfg::FM][Cfg::FN];

    const int kk = lane >> 5;
    const int l31 = lane & 31;
    const bool odd = lane & 1;
    // sign bit to flip on the B fragment: -Im b for lanes (row 1 of B', even column)
    const unsigned sign = (kk == 1 && !odd) ? 0x80000000u : 0u;
    const float alpha = (float)step_alpha(p);

    // Fragments: one k-pair (two MFMA k's) per ds_read_b64, double buffered in
    // registers -- the reads of phase ph+1 are issued before the MFMAs of phase
    // ph, so a wave never waits on LDS latency.  i*32 / j*32 rows do not change
    // fsw(row).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x2 fa[2][Cfg::FM], fb[2][Cfg::FN];
    const int a_row = wm * Cfg::WTM + l31;
    const int b_row = 2 * wn * Cfg::WTN + (l31 ^ kk);
    const int a_sw = fsw(a_row), b_sw = fsw(b_row);
    const float* a_frag = lds + kk * BM * LD + a_row * LD;
    const float* b_frag = lds + AF + b_row * LD;
    auto load_frag = [&](int buf, int ph, int slot) {
        const int boff = buf * (AF + BF);
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i)
            fa[slot][i] = *(const f32x2*)(a_frag + boff + i * 32 * LD + (((ph ^ a_sw) & 7) << 1));
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j)
            fb[slot][j] = *(const f32x2*)(b_frag + boff + j * 32 * LD + (((ph ^ b_sw) & 7) << 1));
    };

    // One k-step = 8 phases of FM*FN*2 MFMAs, each phase fenced for the
    // instruction scheduler.  Everything else a wave has to do for the pipeline
    // sits between the MFMAs of some phase: fragment reads of the next phase,
    // LDS writes of the next k-step (phases 0-1), scalar k bases and global
    // gathers of the k-step after that (phases 2-4).  The single barrier of the
    // step comes before the last phase: by then this wave has finished reading
    // `buf` and writing `buf ^ 1`, so after the barrier it prefetches the first
    // fragments of the next step while its last MFMAs of this step run.
    // has1 / has2: the tile has a k-step kt+1 / kt+2.
    auto k_step = [&](int64_t kt, auto has1, auto has2) {
        constexpr bool H1 = decltype(has1)::value, H2 = decltype(has2)::value;
        const int buf = (int)(kt & 1);
#pragma unroll
        for (int ph = 0; ph < BK / 2; ++ph) {
            if (ph + 1 < BK / 2) {
                load_frag(buf, ph + 1, (ph + 1) & 1);
            } else if (H1) {
                __syncthreads();
                load_frag(buf ^ 1, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // reads first: a full phase of MFMAs covers them
            if (H1 && ph == 0) stage_a(buf ^ 1);
            if (H1 && ph == 1) stage_b(buf ^ 1);
            if (H2 && ph == 2) k_bases(kt + 2);
            if (H2 && ph == 3) gather_a();
            if (H2 && ph == 4) gather_b();
            // the sign of -Im b is applied here, not at the load, so that the
            // read stays in flight across the previous phase
            f32x2 bs[Cfg::FN];
#pragma unroll
            for (int j = 0; j < Cfg::FN; ++j)
                bs[j] = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, fb[ph & 1][j]) ^ sign);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::FN; ++j)
                        acc[i][j] = CTG_MFMA(fa[ph & 1][i][t], bs[j][t], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    k_bases(0);
    gather_a();
    gather_b();
    stage_a(0);
    stage_b(0);
    if (nk > 1) {
        k_bases(1);
        gather_a();
        gather_b();
    }
    __syncthreads();
    CTG_STAMP(T2);
    load_frag(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    unsigned ro[Cfg::FM][8] = {}, co[Cfg::FN] = {};   // epilogue store offsets
    {
        int64_t kt = 0;
        for (; kt + 2 < nk; ++kt) k_step(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) {
            k_step(kt, std::true_type{}, std::false_type{});
            ++kt;
        }
        // store offsets of the epilogue: loaded under the last k-step (the gather
        // registers are dead by now), not after it
        if (partial == nullptr) {
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = 2 * u;
#ifdef CTG_KO_PRO
                    ro[i][u] = (unsigned)((wm * Cfg::WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk + (odd ? 1 : 0)) * BN);
#else
                    ro[i][u] = (unsigned)p.rowC.lo[wm * Cfg::WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk +
                                                   (odd ? 1 : 0)];
#endif
                }
#pragma unroll
#ifdef CTG_KO_PRO
            for (int j = 0; j < Cfg::FN; ++j) co[j] = (unsigned)(wn * Cfg::WTN + j * 16 + (l31 >> 1));
#else
            for (int j = 0; j < Cfg::FN; ++j) co[j] = (unsigned)p.nC[wn * Cfg::WTN + j * 16 + (l31 >> 1)];
#endif
        }
        k_step(kt, std::false_type{}, std::false_type{});
    }

    CTG_STAMP(T3);
    if (partial != nullptr) {
        const int64_t ldp = 2 * tiles_n * BN;
        // (slice-in-batch z owns its own set of slabs: gridDim.z * S_split of them)
        // (scratch is divided among the slices of THIS launch: blockIdx.y, not z0 + blockIdx.y)
        float* slab = partial + ((((int64_t)blockIdx.y * gridDim.z + bz) * S_split + ksplit) * (tiles_m * BM)) * ldp;
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j) {
            const int64_t col = 2 * (n0 + wn * Cfg::WTN + j * 16) + l31;
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int64_t row = m0 + wm * Cfg::WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk;
                    slab[row * ldp + col] = acc[i][j][t];
                }
        }
        return;
    }
    // (round 6: the largest |component| stored -> MfmaHints::cmax, where the executor asks for it: a consumer in the
    // fp16 x 2 arithmetic splits this result under it)
    float vmax = 0.f;
    auto store_tile = [&](auto scaled_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 2 * u;
#pragma unroll
                for (int j = 0; j < Cfg::FN; ++j) {
                    float2 v;
                    if constexpr (decltype(scaled_tag)::value) v = pair_rows(acc[i][j][t], acc[i][j][t + 1], odd, alpha);
                    else v = pair_rows(acc[i][j][t], acc[i][j][t + 1], odd);
                    CTG_STORE_GUARD(alpha)
                    if constexpr (!GROUPED) vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    *(float2*)(C + 2 * (size_t)(ro[i][u] + co[j])) = v;
                }
            }
        }
    };
    // (alpha != 1 only in strip_exponent runs: two multiplies per store otherwise saved)
    if (alpha != 1.f) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    if constexpr (!GROUPED) {
        if (h.cmax != nullptr) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if ((threadIdx.x & 63) == 0 && vmax > 0.f && vmax < __builtin_bit_cast(float, 0x7f800000u))
                record_max(h.cmax + ((int64_t)p.z0 + blockIdx.y) * h.cmax_zs * kMaxSub, vmax);
        }
    }
#ifdef CTG_TIMING
    CTG_STAMP(T4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // all stores acknowledged
    CTG_STAMP(T5);
    if (tid == 0) {
        atomicAdd(&ctg_timing[0], T1 - T0);
        atomicAdd(&ctg_timing[1], T2 - T1);
        atomicAdd(&ctg_timing[2], T3 - T2);
        atomicAdd(&ctg_timing[3], T4 - T3);
        atomicAdd(&ctg_timing[4], T5 - T4);
        atomicAdd(&ctg_timing[5], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------- //
// The tiled fast path with fp32 products as SIX bf16 products (round 5).
//
// Same tiles, same gather order, same tile map, same split-K slabs and the same per-thread constants
// (FastLane / MfmaHints::lane) as pair_mfma_fast_kernel -- what changes is the arithmetic, as in the
// fused stem kernel (ctg_stem.hip, DESIGN 4.2 / HISTORY 4b): every fp32 operand is split EXACTLY
// into three bfloat16 limbs (rounded to nearest: see put3x2), the six cross terms above 2^-24 are accumulated
// in fp32 by v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate for 6x the products).  The split happens ONCE per
// element, where a thread stages what it gathered into LDS (five vector instructions per value, three 2-byte
// writes); the LDS tile holds limb planes
//     [Re | Im][limb][k-block of 8][row][8 k]        (a lane's 8 values of one limb: 16 contiguous bytes)
// so a fragment is one ds_read_b128 and the k-step's 16 k are ONE instruction per (component pair, limb pair).
// Complex on real matrix cores: 32 complex columns per tile, X += Re a Re b + Im a (-Im b), Y += Re a Im b +
// Im a Re b -- the sign lives in a negated copy of Im b's limbs (12 XORs per k-step and column tile), a lane
// ends up with Re and Im of the same element: 8-byte stores, no lane exchange.  Steps with K >= 64 take it
// when the executor multiplies its stem pairs this way (ctg_exec_set_stem_arithmetic; CTG_PAIR_BF16X3=0
// keeps fp32 products here); the wave-front GROUPED launches of small steps stay on fp32 products.
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ constexpr int pbf3_ta(int t) { return t < 3 ? 0 : (t == 5 ? 2 : 1); }
__device__ __forceinline__ constexpr int pbf3_tb(int t) { return t == 1 || t == 4 ? 1 : (t == 2 ? 2 : 0); }

// H2 (round 6; pair_mfma_h2_kernel): the same kernel in the stem kernels' second arithmetic (ctg_stem.hip) --
// TWO rounded fp16 limbs per value under a power-of-two scale per OPERAND TENSOR (MfmaHints::amax / bmax: the largest
// |component|, recorded by the operand's producer or found by a max-abs pass; brought to [2^13, 2^14) before the split),
// THREE products on v_mfma_f32_32x32x16_f16, the two powers of two back in where the result is stored.  4 instead of 5
// vector instructions and 2 instead of 3 LDS writes per value staged, 2 / 3 of the LDS, half the matrix instructions.
// Both arithmetics record the largest |component| they store (MfmaHints::cmax) for a consumer that splits this way.
#ifndef CTG_PAIR16_KBPAD
#define CTG_PAIR16_KBPAD 32   // shorts
#endif
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int pair_h2_exponent_of(float mx) {
    const int ex = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;
    if (mx == 0.f || ex == 128 || ex == -127) return 0;   // (zero / inf / nan / subnormal: left alone)
    const int e = ex - 13;
    return e < -126 ? -126 : (e > 126 ? 126 : e);
}
__device__ __forceinline__ float pair_pow2f(int ex) { return __builtin_bit_cast(float, (unsigned)(ex + 127) << 23); }

template <typename Cfg, bool VEC_A, bool H2>
__device__ __forceinline__ void pair_mfma_16bit_body(const StepArgs& p, const MfmaHints& h, int64_t tiles_m, int64_t tiles_n,
                                                     int64_t k_chunk, float* __restrict__ partial) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    static_assert(BK == 16 && BM == 128 && (BN == 64 || BN == 128), "one bf16 MFMA k-step per tile step");   // (the product uses 64)
    constexpr int WTM = 64, WTN = BN / 2;        // 2 x 2 waves: 64 rows x (32 | 64) complex columns each
    constexpr int FM = 2, FN = WTN / 32;
    constexpr int NA = VEC_A ? Cfg::A_PER_T / 2 : Cfg::A_PER_T;
    constexpr int NL = H2 ? 2 : 3;               // limbs per value
    // limb planes, in shorts: [buf][comp 2][limb NL][kb 2][rows][8]
    // (KBP: the second k-block of a plane starts 64 bytes -- 16 banks -- behind a multiple of the bank period: the 2-byte
    // writes of the lanes that stage k and k + 8 of one row no longer meet in a bank.  SQ counters of the unpadded
    // layout, 65536 x 512 x 512: half of the LDS array's active cycles were bank conflicts, profiles/r6_pair_sq_counters.txt)
    constexpr int KBP = CTG_PAIR16_KBPAD;
    constexpr int AKB = BM * 8 + KBP, BKB = BN * 8 + KBP;      // one k-block of a plane, in shorts
    constexpr int APL = 2 * AKB, BPL = 2 * BKB;                // one (comp, limb) plane of A / B
    constexpr int ASZ = 2 * NL * APL, BSZ = 2 * NL * BPL;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_q[];
    typedef FastLane<Cfg, VEC_A> Lane;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kk = lane >> 5, l31 = lane & 31;
    const int64_t S_split = k_chunk;
    const TileMap tmap = map_tile((blockIdx.x & ~7u) | ((blockIdx.x + blockIdx.y) & 7u), tiles_m * S_split, tiles_n);
    if (!tmap.valid) return;
    const int64_t bz = blockIdx.z;
    const int64_t unit = uniform64(tmap.unit), tn = uniform64(tmap.tn);
    const int64_t ksplit = uniform64(unit / tiles_m);
    const int64_t tm = unit - ksplit * tiles_m;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    int64_t rhi, rlo;
    split_row(p, m0, rhi, rlo);
    rhi = uniform64(rhi);
    rlo = uniform64(rlo);
    const c64* __restrict__ A = (const c64*)p.A + zoffA_s(p) + sload64(p.bA + bz) + sload64(p.rowA.hi + rhi) + sload64(p.rowA.lo + rlo);
    const c64* __restrict__ B = (const c64*)p.B + zoffB_s(p) + sload64(p.bB + bz) + sload64(p.nB + n0);
    float* __restrict__ C = (float*)((c64*)p.C + zoffC_s(p) + sload64(p.bC + bz) + sload64(p.rowC.hi + rhi) +
                                     sload64(p.rowC.lo + rlo) + sload64(p.nC + n0));
    const int64_t nk_total = p.K / BK;
    const int64_t nk = uniform64((nk_total - ksplit + S_split - 1) / S_split);

    // per-thread constants of the gather (shared with the fp32 fast kernel); the LDS slots are decoded from
    // its swizzled float offsets: element (row, k) of the tile
    unsigned a_off[NA], a_ldsf[Lane::NAL], b_ldsf[Lane::NBL], b_off[Cfg::B_PER_T];
    if (h.lane != nullptr) {
        unsigned w[Lane::NQ * 4];
        const uint4* L = (const uint4*)h.lane;
#pragma unroll
        for (int q = 0; q < Lane::NQ; ++q) {
            const uint4 v = L[q * 256 + tid];
            w[4 * q] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        Lane::unpack_words(w, a_off, a_ldsf, b_off, b_ldsf);
    } else {
        Lane::compute(p, h, tid, a_off, a_ldsf, b_off, b_ldsf);
    }
    // short index of element (r, c) inside a (comp, limb) plane: [kb][row][8]
    auto slot_of = [](unsigned o, int rows, int rowscale) {
        const int r = (int)(o / Lane::LD), sl = (int)(o % Lane::LD);
        const int c = (((sl >> 1) ^ Lane::fsw(r)) << 1) | (sl & 1);
        return (unsigned)((c >> 3) * (rows * 8 + KBP) + (r / rowscale) * 8 + (c & 7));
    };
    unsigned short a_q[Cfg::A_PER_T], b_q[Cfg::B_PER_T];
#pragma unroll
    for (int j = 0; j < Cfg::A_PER_T; ++j)
        a_q[j] = (unsigned short)slot_of((j & 1) ? a_ldsf[j / 2] >> 16 : a_ldsf[j / 2] & 0xffffu, BM, 1);
#pragma unroll
    for (int j = 0; j < Cfg::B_PER_T; ++j)   // (the fp32 kernel stages column n at row 2 n of its B' tile)
        b_q[j] = (unsigned short)slot_of((j & 1) ? b_ldsf[j / 2] >> 16 : b_ldsf[j / 2] & 0xffffu, BN, 2);

    c64 a_reg[Cfg::A_PER_T], b_reg[Cfg::B_PER_T];
    int64_t kAh = 0, kAl = 0, kBh = 0, kBl = 0;
    auto k_bases = [&](int64_t step) {
        const int64_t k = uniform64((step * S_split + ksplit) * BK);
        const int64_t kh = k >> p.k_lo_shift, kl = k & (p.k_lo - 1);
        kAh = sload64(p.kA.hi + kh);
        kAl = sload64(p.kA.lo + kl);
        kBh = sload64(p.kB.hi + kh);
        kBl = sload64(p.kB.lo + kl);
    };
    auto gather = [&]() {
        const c64* Ak = A + kAh + kAl;
        if (VEC_A) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const f32x4 v = *(const f32x4*)(Ak + a_off[j]);
                a_reg[2 * j] = c64{v[0], v[1]};
                a_reg[2 * j + 1] = c64{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) a_reg[j] = Ak[a_off[j]];
        }
        const c64* Bk = B + kBh + kBl;
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) b_reg[j] = Bk[b_off[j]];
    };
    // two values -> their three limbs each, planes PL shorts apart.  ROUNDED limbs (v_cvt_pk_bf16_f32: round to
    // nearest even, two values per instruction): x = l1 + l2 + l3 stays exact -- the remainder after two rounded
    // limbs has at most 7 significant bits -- and the three cross terms that are not computed (l2 m3, l3 m2, l3 m3)
    // are below 2^-26 of the product with either sign, where truncated limbs leave up to 2^-23 of one sign: the
    // products carry the fp32 kernel's error, not 1.2 x it (the split is once per element here: 5 instead of 4
    // vector instructions per value are nothing next to 96 MFMAs)
    auto put3x2 = [&](unsigned short* d0, unsigned short* d1, int PL, float x0, float x1) __attribute__((always_inline)) {
        unsigned p1, p2;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(x0), "v"(x1));
        const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r0), "v"(r1));
        const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
        d0[0] = (unsigned short)p1;
        d1[0] = (unsigned short)(p1 >> 16);
        d0[PL] = (unsigned short)p2;
        d1[PL] = (unsigned short)(p2 >> 16);
        d0[2 * PL] = (unsigned short)(__builtin_bit_cast(unsigned, s0) >> 16);
        d1[2 * PL] = (unsigned short)(__builtin_bit_cast(unsigned, s1) >> 16);
    };
    // H2: x s -> two rounded fp16 limbs (the residual x s - h1 is exact in fp32)
    int h2_e = 0;
    float h2_sa = 1.f, h2_sb = 1.f;
    if constexpr (H2) {
        const int64_t zz = (int64_t)p.z0 + blockIdx.y;   // (the slice of the launch this workgroup belongs to)
        int ea = h.amax != nullptr ? pair_h2_exponent_of(read_max(h.amax + zz * h.amax_zs * kMaxSub)) : 0;
        int eb = h.bmax != nullptr ? pair_h2_exponent_of(read_max(h.bmax + zz * h.bmax_zs * kMaxSub)) : 0;
        ea = __builtin_amdgcn_readfirstlane(ea);
        eb = __builtin_amdgcn_readfirstlane(eb);
        h2_sa = pair_pow2f(-ea);
        h2_sb = pair_pow2f(-eb);
        h2_e = ea + eb;
    }
    auto put2x2 = [&](unsigned short* d0, unsigned short* d1, int PL, float x0, float x1) __attribute__((always_inline)) {
        const _Float16 g0 = (_Float16)x0, g1 = (_Float16)x1;
        const _Float16 r0 = (_Float16)(x0 - (float)g0), r1 = (_Float16)(x1 - (float)g1);
        d0[0] = __builtin_bit_cast(unsigned short, g0);
        d1[0] = __builtin_bit_cast(unsigned short, g1);
        d0[PL] = __builtin_bit_cast(unsigned short, r0);
        d1[PL] = __builtin_bit_cast(unsigned short, r1);
    };
    auto stage = [&](int buf) {
        unsigned short* As = lds_q + buf * (ASZ + BSZ);
        unsigned short* Bs = As + ASZ;
        if constexpr (H2) {
#pragma unroll
            for (int j = 0; j < Cfg::A_PER_T; ++j) put2x2(As + a_q[j], As + NL * APL + a_q[j], APL, a_reg[j].re * h2_sa, a_reg[j].im * h2_sa);
#pragma unroll
            for (int j = 0; j < Cfg::B_PER_T; ++j) put2x2(Bs + b_q[j], Bs + NL * BPL + b_q[j], BPL, b_reg[j].re * h2_sb, b_reg[j].im * h2_sb);
            return;
        }
#pragma unroll
        for (int j = 0; j < Cfg::A_PER_T; ++j) put3x2(As + a_q[j], As + 3 * APL + a_q[j], APL, a_reg[j].re, a_reg[j].im);
#pragma unroll
        for (int j = 0; j < Cfg::B_PER_T; ++j) put3x2(Bs + b_q[j], Bs + 3 * BPL + b_q[j], BPL, b_reg[j].re, b_reg[j].im);
    };

    f32x16 ax[FM][FN], ay[FM][FN];
    const float alpha = (float)step_alpha(p);
    // fragment bases: lane = (row | column l31, k-block kk)
    const int a_frag = kk * AKB + (wm * WTM + l31) * 8;
    const int b_frag = kk * BKB + (wn * WTN + l31) * 8;
    auto k_step = [&](int buf, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const unsigned short* As = lds_q + buf * (ASZ + BSZ);
        const unsigned short* Bs = As + ASZ;
        f32x16 zero16;
#pragma unroll
        for (int u = 0; u < 16; ++u) zero16[u] = 0.f;
        pbf16x8 aR[FM][NL], aI[FM][NL];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                aR[i][q] = *(const pbf16x8*)(As + q * APL + a_frag + i * 32 * 8);
                aI[i][q] = *(const pbf16x8*)(As + (NL + q) * APL + a_frag + i * 32 * 8);
            }
        // (the sign bit of a bf16 and of an fp16 value is the same bit: one mask negates either)
        auto mm = [](pbf16x8 a, pbf16x8 b, f32x16 c) __attribute__((always_inline)) -> f32x16 {
            if constexpr (H2)
                return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, a), __builtin_bit_cast(pf16x8, b), c, 0, 0, 0);
            else
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        };
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            pbf16x8 bR[NL], bI[NL], nI[NL];
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                bR[q] = *(const pbf16x8*)(Bs + q * BPL + b_frag + j * 32 * 8);
                bI[q] = *(const pbf16x8*)(Bs + (NL + q) * BPL + b_frag + j * 32 * 8);
                nI[q] = __builtin_bit_cast(pbf16x8, __builtin_bit_cast(pu32x4, bI[q]) ^ 0x80008000u);
            }
            // (H2: limbs 0, 1 only -- the products (0, 0), (0, 1), (1, 0): t = 0, 1, 3)
#pragma unroll
            for (int t = 0; t < 6; t += (H2 ? (t == 1 ? 2 : (t == 3 ? 3 : 1)) : 1)) {
                const int ta = pbf3_ta(t), tb = pbf3_tb(t);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    ax[i][j] = mm(aR[i][ta], bR[tb], (FIRST && t == 0) ? zero16 : ax[i][j]);
                    ay[i][j] = mm(aR[i][ta], bI[tb], (FIRST && t == 0) ? zero16 : ay[i][j]);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    ax[i][j] = mm(aI[i][ta], nI[tb], ax[i][j]);
                    ay[i][j] = mm(aI[i][ta], bR[tb], ay[i][j]);
                }
            }
        }
    };

    // pipeline: gather k-step t + 1 into registers while the MFMAs of k-step t run; split and stage it into the
    // other LDS buffer; one barrier per k-step
    k_bases(0);
    gather();
    stage(0);
    if (nk > 1) {
        k_bases(1);
        gather();
    }
    __syncthreads();
    k_step(0, std::true_type{});
    for (int64_t kt = 1; kt < nk; ++kt) {
        const int buf = (int)(kt & 1);
        stage(buf);                      // (k-step kt: gathered during k-step kt - 1)
        if (kt + 1 < nk) {
            k_bases(kt + 1);
            gather();
        }
        __syncthreads();
        k_step(buf, std::false_type{});
    }

    // epilogue: accumulator register t of a tile is row rowmap(t) + 4 kk, the lane's column is l31
    if (partial != nullptr) {
        const int64_t ldp = 2 * tiles_n * BN;
        float* slab = partial + ((((int64_t)blockIdx.y * gridDim.z + bz) * S_split + ksplit) * (tiles_m * BM)) * ldp;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int64_t col = 2 * (n0 + wn * WTN + j * 32 + l31);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int64_t row = m0 + wm * WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk;
                    float2 v;
                    v.x = ax[i][j][t];
                    v.y = ay[i][j][t];
                    *(float2*)(slab + row * ldp + col) = v;
                }
        }
        return;
    }
    unsigned co[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) co[j] = (unsigned)p.nC[wn * WTN + j * 32 + l31];
    // H2: the operands' powers of two back in, as two factors inside the float range (the executor does not take this
    // arithmetic under strip_exponent: alpha is 1 there)
    float f1 = alpha, f2 = 1.f;
    if constexpr (H2) {
        const int e1 = h2_e < -126 ? -126 : (h2_e > 126 ? 126 : h2_e);
        int e2 = h2_e - e1;
        e2 = e2 < -126 ? -126 : (e2 > 126 ? 126 : e2);
        f1 = alpha * pair_pow2f(e1);
        f2 = pair_pow2f(e2);
    }
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned ro = (unsigned)p.rowC.lo[wm * WTM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * kk];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                float2 v;
                v.x = H2 ? ax[i][j][t] * f1 * f2 : ax[i][j][t] * alpha;
                v.y = H2 ? ay[i][j][t] * f1 * f2 : ay[i][j][t] * alpha;
                vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                *(float2*)(C + 2 * (size_t)(ro + co[j])) = v;
            }
        }
    if (h.cmax != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax > 0.f && vmax < __builtin_bit_cast(float, 0x7f800000u))
            record_max(h.cmax + ((int64_t)p.z0 + blockIdx.y) * h.cmax_zs * kMaxSub, vmax);
    }
}

template <typename Cfg, bool VEC_A>
__global__ __launch_bounds__(256, 1) void pair_mfma_bf3_kernel(StepArgs p, MfmaHints h, int64_t tiles_m, int64_t tiles_n,
                                                              int64_t k_chunk, float* __restrict__ partial) {
    pair_mfma_16bit_body<Cfg, VEC_A, false>(p, h, tiles_m, tiles_n, k_chunk, partial);
}
template <typename Cfg, bool VEC_A>
__global__ __launch_bounds__(256, 1) void pair_mfma_h2_kernel(StepArgs p, MfmaHints h, int64_t tiles_m, int64_t tiles_n,
                                                             int64_t k_chunk, float* __restrict__ partial) {
    pair_mfma_16bit_body<Cfg, VEC_A, true>(p, h, tiles_m, tiles_n, k_chunk, partial);
}

// sum the split-K slabs in a fixed order and scatter into C.  A block reduces
// 32 outputs: 8 thread groups each add every 8th slab (loads unrolled so many
// are in flight), then the 8 partial sums are combined in a fixed order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(StepArgs p, int64_t S, int64_t Mpad,
                                                            int64_t ldp,
                                                            const float* __restrict__ partial) {
    __shared__ float2 part[8][32];
    const float alpha = (float)step_alpha(p);
    const int ox = threadIdx.x & 31, sy = threadIdx.x >> 5;
    const int64_t per_b = p.R * p.N;
    const int64_t total = per_b * p.Bt;
    for (int64_t o0 = (int64_t)blockIdx.x * 32; o0 < total; o0 += (int64_t)gridDim.x * 32) {
        const int64_t o = o0 + ox;
        const bool ok = o < total;
        int64_t b = 0, m = 0, n = 0;
        float re = 0.f, im = 0.f;
        if (ok) {
            b = o / per_b;
            const int64_t rem = o - b * per_b;
            m = rem / p.N;
            n = rem - m * p.N;
            const float* src = partial + (((int64_t)blockIdx.y * p.Bt + b) * S * Mpad + m) * ldp + 2 * n;
            const int64_t slab = Mpad * ldp;
#pragma unroll 8
            for (int64_t s = sy; s < S; s += 8) {
                const float2 v = *(const float2*)(src + s * slab);
                re += v.x;
                im += v.y;
            }
        }
        part[sy][ox] = make_float2(re, im);
        __syncthreads();
        if (sy == 0 && ok) {
            float r2 = 0.f, i2 = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                r2 += part[g][ox].x;
                i2 += part[g][ox].y;
            }
            int64_t hi, lo;
            split_row(p, m, hi, lo);
            c64* C = (c64*)p.C + zoffC(p) + p.bC[b];
            C[p.rowC.hi[hi] + p.rowC.lo[lo] + p.nC[n]] = c64{r2 * alpha, i2 * alpha};
        }
        __syncthreads();
    }
}

// Arithmetic of the long tiled steps: bf16 x 3 products when the executor multiplies its stem pairs that way
// (StepArgs::bf3: ctg_exec_set_stem_arithmetic, default on) unless CTG_PAIR_BF16X3 / CTG_STEM_BF16X3 in the
// environment say otherwise ("0" / "" = fp32 products; read at every launch, tests switch within a process).
bool pair_bf16x3_on(const StepArgs& p) {
    auto off = [](const char* v) { return v != nullptr && (v[0] == '\0' || (v[0] == '0' && v[1] == '\0')); };
    const char* v = getenv("CTG_PAIR_BF16X3");
    if (v != nullptr) return !off(v);
    v = getenv("CTG_STEM_BF16X3");
    if (v != nullptr) return !off(v);
    return p.bf3 != 0;
}

template <typename Cfg>
static hipError_t launch_cfg(const StepArgs& p, const MfmaHints& h, void* scratch,
                             int64_t scratch_bytes, hipStream_t stream) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    const int64_t tiles_m = (p.R + BM - 1) / BM;
    const int64_t tiles_n = (p.N + BN - 1) / BN;
    // split-K when the output alone cannot fill the chip but K is long: decided once per
    // step when the executor is built (MfmaHints::splitk)
    const int64_t nk_total = (p.K + BK - 1) / BK;
    int64_t S = h.splitk > 0 ? h.splitk : mfma_split_count(p.R, p.N, p.K, p.Bt, BN, scratch_bytes);
    if (S > nk_total) S = nk_total;
    if (S > 1 && S * (tiles_m * BM * tiles_n * BN * 8 * p.Bt) > scratch_bytes) return hipErrorInvalidValue;
    // (the split depends on the step alone, never on how many slices a launch
    // carries: a result must not depend on the batching of a run)
    if (S > 1 && p.nz > 1) {   // the slabs of the slices in one launch must fit the scratch buffer
        const int64_t room = p.scratch_total > scratch_bytes ? p.scratch_total : scratch_bytes;
        const int64_t fit = room / (S * (tiles_m * BM * tiles_n * BN * 8 * p.Bt));
        if (fit < p.nz)
            return for_each_z_chunk(p, fit, [&](const StepArgs& q) {
                return launch_cfg<Cfg>(q, h, scratch, scratch_bytes, stream);
            });
    }
    const int64_t k_chunk = S;  // the kernels' k_chunk argument carries the split count
    const int64_t gx = tile_grid_blocks(tiles_m * S, tiles_n);
    if (gx > 0x7fffffffll || p.nz > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)gx, (unsigned)p.nz, (unsigned)p.Bt);
    float* part = S > 1 ? (float*)scratch : (float*)nullptr;
    if constexpr (Cfg::BN == 64) {
        // fp32 products as six bf16 products (pair_mfma_bf3_kernel): long contractions on full 64-column tiles
        if (h.fast && h.bf3 && pair_bf16x3_on(p)) {
            // (h.h2: this launch in the fp16 x 2 arithmetic -- the executor's decision, ctg_runtime.hip; never with
            // k-splits: the slabs hold unscaled sums)
            if (h.h2 && S > 1) return hipErrorInvalidValue;
            const bool h2 = h.h2 != 0;
            const size_t smem = 2 * 2 * (h2 ? 4 : 6) * (size_t)(2 * (Cfg::BM * 8 + CTG_PAIR16_KBPAD) + 2 * (Cfg::BN * 8 + CTG_PAIR16_KBPAD));
            static unsigned long long ready[4] = {0, 0, 0, 0};   // (per-device bit masks, updated atomically: lds_opt_in)
            const void* fn = h2 ? (h.vecA ? (const void*)pair_mfma_h2_kernel<Cfg, true> : (const void*)pair_mfma_h2_kernel<Cfg, false>)
                                : (h.vecA ? (const void*)pair_mfma_bf3_kernel<Cfg, true> : (const void*)pair_mfma_bf3_kernel<Cfg, false>);
            {
                const hipError_t e = lds_opt_in(fn, (int)smem, &ready[(h2 ? 2 : 0) + (h.vecA ? 1 : 0)]);
                if (e != hipSuccess) return e;
            }
            if (h2 && h.vecA)
                hipLaunchKernelGGL((pair_mfma_h2_kernel<Cfg, true>), grid, dim3(256), smem, stream, p, h, tiles_m, tiles_n, k_chunk, part);
            else if (h2)
                hipLaunchKernelGGL((pair_mfma_h2_kernel<Cfg, false>), grid, dim3(256), smem, stream, p, h, tiles_m, tiles_n, k_chunk, part);
            else if (h.vecA)
                hipLaunchKernelGGL((pair_mfma_bf3_kernel<Cfg, true>), grid, dim3(256), smem, stream, p, h, tiles_m, tiles_n, k_chunk, part);
            else
                hipLaunchKernelGGL((pair_mfma_bf3_kernel<Cfg, false>), grid, dim3(256), smem, stream, p, h, tiles_m, tiles_n, k_chunk, part);
            if (S > 1) {
                int64_t blocks = (p.R * p.N * p.Bt + 31) / 32;
                if (blocks > 8192) blocks = 8192;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p, S,
                                   tiles_m * BM, 2 * tiles_n * BN, (const float*)scratch);
            }
            return hipGetLastError();
        }
    }
    if (h.fast) {
        if (h.vecA)
            hipLaunchKernelGGL((pair_mfma_fast_kernel<Cfg, true>), grid, dim3(256), 0, stream, p, h,
                               tiles_m, tiles_n, k_chunk, part, (const FastGroupItem*)nullptr, 0);
        else
            hipLaunchKernelGGL((pair_mfma_fast_kernel<Cfg, false>), grid, dim3(256), 0, stream, p, h,
                               tiles_m, tiles_n, k_chunk, part, (const FastGroupItem*)nullptr, 0);
    } else if constexpr (Cfg::BN >= 128) {
        return hipErrorInvalidValue;  // the 128-wide tile exists for the fast path only
    } else if (h.vecA)
        hipLaunchKernelGGL((pair_mfma_c64_kernel<Cfg, true>), grid, dim3(256), 0, stream, p, h,
                           tiles_m, tiles_n, k_chunk, part);
    else
        hipLaunchKernelGGL((pair_mfma_c64_kernel<Cfg, false>), grid, dim3(256), 0, stream, p, h,
                           tiles_m, tiles_n, k_chunk, part);
    if (S > 1) {
        int64_t blocks = (p.R * p.N * p.Bt + 31) / 32;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p, S,
                           tiles_m * BM, 2 * tiles_n * BN, (const float*)scratch);
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------------- //
// Streaming variant for tall-skinny steps (K <= 128, N <= 64, R huge): the
// HBM-bound majority of a sliced Sycamore contraction.
//
// B (K x N, a few KB) is gathered into LDS once per block in MFMA fragment
// layout.  Every wave then streams its own 32-row groups of A completely on
// its own: 16-byte gathers in ascending address order -> wave-private LDS
// transpose -> fragments -> MFMA -> 8-byte stores.  There is no block barrier
// in the loop, so the 8-12 waves of a CU overlap each other's memory latency,
// and row/k offsets never touch global tables inside the loop (tile-additive
// offsets: one scalar base per group + per-lane constants).
// ------------------------------------------------------------------------- //

// Occupancy the register allocator is held to: 16 / 12 / 8 waves per CU for
// 16 / 32 / 64 output columns.  The narrow variants are latency-bound (bytes
// in flight per CU), so the extra waves are worth more than the registers.
template <int FN, bool VEC, bool ADD, bool SHORTK, int NV>
#ifndef CTG_STREAM_OCC2
#define CTG_STREAM_OCC2 3
#endif
#ifndef CTG_STREAM_OCC4
#define CTG_STREAM_OCC4 2
#endif
#ifndef CTG_STREAM_DEPTH
#define CTG_STREAM_DEPTH 2
#endif
#ifndef CTG_STREAM_PIPE4  // 64-column steps: fragment reads double-buffered in registers (costs 20 VGPRs)
#define CTG_STREAM_PIPE4 1
#endif
#ifndef CTG_STREAM_BREG   // B panels of up to this many (chunks x column tiles) in registers
#define CTG_STREAM_BREG 0
#endif
__global__ __launch_bounds__(256, (FN == 1 ? 4 : FN == 2 ? CTG_STREAM_OCC2 : CTG_STREAM_OCC4)) void pair_mfma_stream_kernel(StepArgs p, MfmaHints h, int KP,
                                                               int64_t n_groups) {
    constexpr int LD = MFMA_BK + 4;
    constexpr int PER_T = 32 * MFMA_BK / 64;  // A elements per lane per chunk (8)
    // A short contraction (K <= 8 / K <= 4) fills only the first NV = 4 / 2 slots of a
    // lane's gather list; the register budget of two full tasks then holds DEPTH =
    // 4 / 8 tasks in flight, which is what keeps HBM busy when a task is 1-2 KB.
    // (with the steady-state loop below keeping exactly DEPTH tasks in flight, two
    // are enough for every width; three or four measured the same)
    constexpr int DEPTH = CTG_STREAM_DEPTH * (PER_T / NV);
    static_assert((DEPTH & (DEPTH - 1)) == 0, "register sets rotate with a power-of-two period");
    static_assert(NV == PER_T || SHORTK, "partial gather lists only for short contractions");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDB = KP + 4;
    // B' as the MFMA wants it, one plane per k-row of B' so that no lane has to
    // flip a sign or pick its row at run time: plane 0 (lanes 0-31, kk = 0) holds
    // rows (Re b_n, Im b_n), plane 1 (lanes 32-63) rows (-Im b_n, Re b_n)
    constexpr int BROWS = 2 * 16 * FN;
    float* Bs = (float*)smem;                                   // [2][BROWS][LDB]
    int64_t* kofs_s = (int64_t*)(Bs + 2 * BROWS * LDB);         // [KP]
    float* As_all = (float*)(kofs_s + KP);                      // [4 waves][2][32][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int kk = lane >> 5;
    const int l31 = lane & 31;
    const bool odd = lane & 1;

    const c64* __restrict__ A = (const c64*)p.A + zoffA(p);
    const c64* __restrict__ B = (const c64*)p.B + zoffB(p);
    float* __restrict__ C = (float*)((c64*)p.C + zoffC(p));

    // ---- B and the k offsets of A into LDS (once per block) -----------------
    for (int e = tid; e < KP * 16 * FN; e += 256) {
        const int n = e / KP, k = e - n * KP;
        c64 v{0.f, 0.f};
        if (n < p.N && k < p.K) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            v = B[p.nB[n] + p.kB.hi[kh] + p.kB.lo[kl]];
        }
        Bs[(2 * n) * LDB + k] = v.re;
        Bs[(2 * n + 1) * LDB + k] = v.im;
        Bs[(BROWS + 2 * n) * LDB + k] = -v.im;
        Bs[(BROWS + 2 * n + 1) * LDB + k] = v.re;
    }
    for (int k = tid; k < KP; k += 256) {
        int64_t off = -1;
        if (k < p.K) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            off = p.kA.hi[kh] + p.kA.lo[kl];
        }
        kofs_s[k] = off;
    }

    // ---- per-lane constants ---------------------------------------------------
    // a_pk[j] = (row << 16) | lds offset of element j; column = lds offset % LD
    int a_pk[NV];
    int a_delta[NV];  // ADD: row offset relative to the group base (host-checked < 2^31)
    {
        const uint16_t* oa = h.ordA + lane * PER_T;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = oa[j];
            if (SHORTK && (v & 0x8000)) {  // short-K padding: nothing to gather for this slot
                a_pk[j] = -1;
                if (ADD) a_delta[j] = 0;
                continue;
            }
            const int r = v >> 4, c = v & 15;
            a_pk[j] = (r << 16) | (r * LD + c);
            if (ADD) a_delta[j] = (int)p.rowA.lo[r];
        }
    }
    float rec_max = 0.f;   // (round 6) largest |component| stored by this lane
    // rows this lane stores in the epilogue: 8 (paired) rows per MFMA tile
    int c_delta[8];
    if (ADD) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = 2 * u;
            c_delta[u] = (int)p.rowC.lo[(t & 3) + 8 * (t >> 2) + 4 * kk + (odd ? 1 : 0)];
        }
    }
    int64_t ncol[FN];
    bool n_ok[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int64_t n = j * 16 + (l31 >> 1);
        n_ok[j] = n < p.N;
        ncol[j] = n_ok[j] ? p.nC[n] : 0;
    }
    __syncthreads();

    float* As = As_all + wave * (2 * 32 * LD);
    // columns never gathered (k >= K of a short contraction) must read as zero
    if (SHORTK)
        for (int i = lane; i < 2 * 32 * LD; i += 64) As[i] = 0.f;
    const float alpha = (float)step_alpha(p);
    const bool scaled = alpha != 1.f;   // (strip_exponent runs only; wave-uniform)
    const int n_chunks = KP / MFMA_BK;
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    // ---- task stream: (group, chunk) pairs, gathers two tasks ahead ----------
    // Two register sets alternate (R0/R1) so that 2 x 4 KB per wave are always
    // in flight while the matrix cores work on the task before.
    int64_t a_base_off = 0;  // ADD: scalar row base of the group being gathered

    // !ADD (rows of a group are not base + per-lane constant: extents that are not
    // powers of two): lane l holds the A offset of row l of the group being
    // gathered (-1 past the last row) and of the group this wave gathers next --
    // fetched one group ahead, so the two dependent table loads per row sit behind a
    // whole task's gathers instead of in front of every single element's load; the
    // element loads pick their row's offset from its lane (ds_bpermute).  Same for
    // the C offsets of the group whose tile is stored next.
    int64_t a_row_cur = -1, a_row_next = -1, c_row_cur = -1, c_row_next = -1;
    bool a_rows_primed = false, c_rows_primed = false;
    auto fetch_row = [&](const RowTab& tab, int64_t g) __attribute__((always_inline)) -> int64_t {
        const int64_t m = g * 32 + l31;
        if (g >= n_groups || m >= p.R) return -1;
        int64_t hi, lo;
        split_row(p, m, hi, lo);
        return tab.hi[hi] + tab.lo[lo];
    };
    auto lane_get = [&](int64_t v, int src) __attribute__((always_inline)) -> int64_t {
        const int lo = __shfl((int)(v & 0xffffffffll), src, 64);
        const int hi = __shfl((int)(v >> 32), src, 64);
        return ((int64_t)hi << 32) | (unsigned)lo;
    };

    auto resolve_rows_a = [&](int64_t g) __attribute__((always_inline)) {
        const int64_t m0 = g * 32;
        if (ADD) {
            int64_t hi, lo;
            split_row(p, uniform64(m0), hi, lo);
            // (sload64: the tables are never written on the device, but the kernel
            // stores to C, so only the constant address space gets a scalar load --
            // a vector load here would wait for every gather in flight, vmcnt(0))
            a_base_off = sload64(p.rowA.hi + uniform64(hi)) + sload64(p.rowA.lo + uniform64(lo));
        } else {
            a_row_cur = a_rows_primed ? a_row_next : fetch_row(p.rowA, g);
            a_rows_primed = true;
            a_row_next = fetch_row(p.rowA, g + n_waves);
        }
    };
    auto gather = [&](c64 (&a_reg)[NV], int64_t g, int chunk, auto all_live) __attribute__((always_inline)) {
        const int64_t* ka = kofs_s + chunk * MFMA_BK;
        const int64_t m0 = g * 32;
        if (VEC) {
#pragma unroll
            for (int j = 0; j < NV; j += 2) {
                // wave-uniform (depends on j only); all_live: the host-side order
                // table fills every slot, so the loads are unconditional
                const int r = a_pk[j] >> 16, c = (a_pk[j] & 0xffff) - r * LD;
                int64_t ro;
                if (ADD) {
                    ro = a_base_off + a_delta[j];
                } else {
                    ro = lane_get(a_row_cur, r & 31);   // (cross-lane: before any per-lane skip)
                }
                if (!decltype(all_live)::value && SHORTK && a_pk[j] < 0) continue;
#ifdef CTG_KO_GATHER
                const f32x4 v = {(float)ro, 1.f, 2.f, (float)ka[c]};
#else
                const f32x4 v = *(const f32x4*)(A + ro + ka[c]);
#endif
                a_reg[j] = c64{v[0], v[1]};
                a_reg[j + 1] = c64{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = a_pk[j] >> 16, c = (a_pk[j] & 0xffff) - r * LD;
                int64_t ro = -1;
                if (ADD) {
                    ro = a_base_off + a_delta[j];
                } else {
                    ro = lane_get(a_row_cur, r & 31);   // (cross-lane: before any per-lane skip)
                }
                if (SHORTK && a_pk[j] < 0) continue;
                const int64_t ko = ka[c];
                c64 v{0.f, 0.f};
                if (ro >= 0 && ko >= 0) v = A[ro + ko];
                a_reg[j] = v;
            }
        }
    };

    f32x16 acc[FN];
    const int64_t my_groups = wave_g < n_groups ? (n_groups - wave_g + n_waves - 1) / n_waves : 0;
    const int64_t n_tasks = my_groups * n_chunks;

    // cursor of the next task to ISSUE
    int64_t ig = wave_g;
    int ic = 0;
    int64_t issued = 0;
    auto issue = [&](c64 (&a_reg)[NV], auto steady) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady)::value;   // the task is known to exist
        if (STEADY || issued < n_tasks) {
            if (ic == 0) resolve_rows_a(ig);
            gather(a_reg, ig, ic, steady);
            ++issued;
            if (++ic == n_chunks) {
                ic = 0;
                ig += n_waves;
            }
        }
    };
    // cursor of the task being CONSUMED
    int64_t cg = wave_g;
    int cc = 0;
    // steady / first / last: compile-time knowledge of the steady-state loop
    // below (std::false_type everywhere = the general, fully dynamic task)
    // bsrc: nullptr = B fragments come from LDS; else the 4 * FN fragments of this
    // chunk held in registers (steady state of small B panels, see steady_loop)
    auto consume = [&](c64 (&a_reg)[NV], auto steady, auto first_tag, auto last_tag, auto bsrc) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady)::value;
        constexpr bool BREG = !std::is_same<decltype(bsrc), std::nullptr_t>::value;
        if (STEADY ? decltype(first_tag)::value : cc == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;
        }
        // registers -> wave-private LDS (transpose to fragment layout)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (SHORTK && a_pk[j] < 0) continue;
            const int o = a_pk[j] & 0xffff;
#ifdef CTG_KO_LDS
            if (a_reg[j].re == 12345.678f) As[o] = a_reg[j].im;   // (keeps the loads alive)
#else
            As[o] = a_reg[j].re;
            As[32 * LD + o] = a_reg[j].im;
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // refill this register set DEPTH tasks ahead
        issue(a_reg, steady);
        const bool last = STEADY ? decltype(last_tag)::value : cc == n_chunks - 1;
        // C row base of this group: scalar loads hidden behind the MFMAs below
        int64_t c_base = 0;
        if (ADD && last) {
            int64_t hi, lo;
            split_row(p, uniform64(cg * 32), hi, lo);
            c_base = sload64(p.rowC.hi + uniform64(hi)) + sload64(p.rowC.lo + uniform64(lo));
        }
        if (!ADD && last) {
            c_row_cur = c_rows_primed ? c_row_next : fetch_row(p.rowC, cg);
            c_rows_primed = true;
            c_row_next = fetch_row(p.rowC, cg + n_waves);
        }
        const float* a_base = As + kk * 32 * LD + l31 * LD;
        const float* b_base = Bs + (kk * BROWS + l31) * LDB + cc * MFMA_BK;
        const int k_left = (int)p.K - cc * MFMA_BK;
        const int nq = k_left >= MFMA_BK ? MFMA_BK / 4 : (k_left + 3) / 4;
        if constexpr (STEADY && !SHORTK && (FN < 4 || CTG_STREAM_PIPE4)) {
            // full chunk: the four k-quads are unrolled with the fragments of quad
            // q+1 read (ds_read_b128) before the MFMAs of quad q are issued, so the
            // matrix pipe never waits for this wave's LDS latency.  B fragments in
            // registers (BREG) are already signed: one LDS read per quad remains.
            f32x4 af[2], bfr[2][FN];
            af[0] = *(const f32x4*)(a_base);
            if constexpr (!BREG) {
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[0][j] = *(const f32x4*)(b_base + j * 32 * LDB);
            }
#pragma unroll
            for (int kq = 0; kq < MFMA_BK / 4; ++kq) {
                if (kq + 1 < MFMA_BK / 4) {
                    af[(kq + 1) & 1] = *(const f32x4*)(a_base + (kq + 1) * 4);
                    if constexpr (!BREG) {
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            bfr[(kq + 1) & 1][j] = *(const f32x4*)(b_base + j * 32 * LDB + (kq + 1) * 4);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 bs[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (BREG) bs[j] = bsrc[kq * FN + j];
                    else bs[j] = bfr[kq & 1][j];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[j] = CTG_MFMA(af[kq & 1][t], bs[j][t], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int kq = 0; kq < nq; ++kq) {
            const f32x4 af = *(const f32x4*)(a_base + kq * 4);
            f32x4 bf[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bf[j] = *(const f32x4*)(b_base + j * 32 * LDB + kq * 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[j] = CTG_MFMA(af[t], bf[j][t], acc[j]);
        }
        if (last) {
            // epilogue: pair rows (t, t+1) so each lane stores whole complex numbers
            auto store_group = [&](auto scaled_tag) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = 2 * u;
                    int64_t ro;
                    if (ADD) {
                        ro = c_base + c_delta[u];
                    } else {
                        ro = lane_get(c_row_cur, (t & 3) + 8 * (t >> 2) + 4 * kk + (odd ? 1 : 0));
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        float2 v;
                        if constexpr (decltype(scaled_tag)::value) v = pair_rows(acc[j][t], acc[j][t + 1], odd, alpha);
                        else v = pair_rows(acc[j][t], acc[j][t + 1], odd);
                        CTG_STORE_GUARD(alpha)
                        if (STEADY || (n_ok[j] && ro >= 0)) {
                            rec_max = fmaxf(rec_max, fmaxf(fabsf(v.x), fabsf(v.y)));
                            *(float2*)(C + 2 * (ro + ncol[j])) = v;
                        }
                    }
                }
            };
            // (alpha != 1 only in strip_exponent runs; wave-uniform branch)
            if (scaled) store_group(std::true_type{});
            else store_group(std::false_type{});
            cc = 0;
            cg += n_waves;
        } else {
            ++cc;
        }
    };

    c64 regs[DEPTH][NV];
    int64_t t = 0;
    // Steady state.  s_waitcnt vmcnt is positional (it counts the loads AND stores
    // issued after the one waited for), so the compiler can only keep DEPTH tasks
    // in flight if it knows exactly which memory instructions lie between a gather
    // and its use: every conditional gather or store on the way makes it fall back
    // to "wait for everything".  For full tiles (all columns and gather slots live)
    // the loop is therefore unrolled over max(chunks per group, DEPTH) tasks with
    // the first / last chunk of a group known at compile time, unconditional
    // refills and unconditional stores; the general loop below finishes the tail.
    bool primed = false;
    if constexpr (VEC && ADD) {
        // (short contractions, K < 16, stay on the general loop: measured 25 % slower
        // in this form at K = 8 -- tasks of 2 KB in, 4 KB out)
        const bool full = p.N == 16 * FN && !SHORTK;
        auto steady_loop = [&](auto nc_tag) __attribute__((always_inline)) {
            constexpr int NC = decltype(nc_tag)::value;
            constexpr int BODY = NC > DEPTH ? NC : DEPTH;
            if (n_tasks < BODY + DEPTH) return;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) issue(regs[d], std::true_type{});
            primed = true;
            // A small B panel (K = N = 16) lives in registers for the whole
            // kernel: NC chunks x 4 quads x FN fragments = 16 VGPRs; larger
            // panels would spill.  That takes the B reads out of the task loop --
            // every instruction issued between MFMAs costs matrix-pipe time
            // (DESIGN section 4).
            constexpr bool BREG = NC * FN <= CTG_STREAM_BREG;
            f32x4 breg[BREG ? NC : 1][4 * FN];
            if constexpr (BREG) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            breg[c][kq * FN + j] =
                                *(const f32x4*)(Bs + (kk * BROWS + l31 + j * 32) * LDB + c * MFMA_BK + kq * 4);
                        }
            }
            auto body = [&]() __attribute__((always_inline)) {
                static_for<0, BODY>([&](auto i) __attribute__((always_inline)) {
                    constexpr int I = decltype(i)::value;
                    if constexpr (BREG)
                        consume(regs[I % DEPTH], std::true_type{}, std::bool_constant<(I % NC) == 0>{},
                                std::bool_constant<(I % NC) == NC - 1>{}, (const f32x4*)breg[I % NC]);
                    else
                        consume(regs[I % DEPTH], std::true_type{}, std::bool_constant<(I % NC) == 0>{},
                                std::bool_constant<(I % NC) == NC - 1>{}, nullptr);
                });
            };
            // The first pass is peeled.  The compiler's s_waitcnt at the loop header
            // must hold for the entry path and for the back edge, and it takes the
            // smaller count of the two: entered straight from the priming gathers
            // (no stores issued yet) the header wait became vmcnt(7) -- i.e. every
            // pass began by waiting for the previous group's stores to be
            // acknowledged (microseconds) although only its own gather was needed.
            // With one pass in front both paths carry the same memory operations.
            body();
            t += BODY;
            for (; t + BODY + DEPTH <= n_tasks; t += BODY) body();
        };
        if (full) {
            switch (n_chunks) {
                case 1: steady_loop(std::integral_constant<int, 1>{}); break;
                case 2: steady_loop(std::integral_constant<int, 2>{}); break;
                case 4: steady_loop(std::integral_constant<int, 4>{}); break;
                case 8: steady_loop(std::integral_constant<int, 8>{}); break;
            }
        }
    }
    if (!primed) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(regs[d], std::false_type{});
    }
    for (; t < n_tasks; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (t + d < n_tasks) consume(regs[d], std::false_type{}, std::false_type{}, std::false_type{}, nullptr);
    }
    // (round 6) the largest |component| this wave stored -> MfmaHints::cmax, where the executor asks for it
    if (h.cmax != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rec_max = fmaxf(rec_max, __shfl_xor(rec_max, o));
        if ((threadIdx.x & 63) == 0 && rec_max > 0.f && rec_max < __builtin_bit_cast(float, 0x7f800000u))
            record_max(h.cmax + ((int64_t)p.z0 + blockIdx.y) * h.cmax_zs * kMaxSub, rec_max);
    }
}

template <int FN, bool VEC, bool ADD, bool SHORTK, int NV>
static hipError_t launch_stream_t(const StepArgs& p, const MfmaHints& h, int KP, size_t smem,
                                  hipStream_t stream) {
    auto kern = pair_mfma_stream_kernel<FN, VEC, ADD, SHORTK, NV>;
    static int blocks_per_cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // by KP / 16
    int& bpc = blocks_per_cu[KP / 16];
    if (smem > 48 * 1024) {
        // opt in to more than the default dynamic LDS (the CU has 160 KiB); per device
        static unsigned long long opted = 0;
        hipError_t e = lds_opt_in((const void*)kern, 120 * 1024, &opted);
        if (e != hipSuccess) return e;
    }
    if (bpc == 0) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, smem);
        if (e != hipSuccess) return e;
        bpc = n < 1 ? 1 : (n > 8 ? 8 : n);
    }
    const int64_t n_groups = (p.R + 31) / 32;
    int64_t blocks = (n_groups + 3) / 4;
    // persistent: exactly the resident blocks (CTG_STREAM_OVERSUB=n, experiments: n times as
    // many, each with 1/n of the row groups, the hardware scheduler balancing them)
    static const int64_t oversub = getenv("CTG_STREAM_OVERSUB") ? atoll(getenv("CTG_STREAM_OVERSUB")) : 1;
    const int64_t cap = 256ll * bpc * (oversub > 0 ? oversub : 1);
    if (blocks > cap) blocks = cap;
    // the slices of a batch divide the resident blocks among themselves (every block
    // strides over the row groups of its slice, whatever their number)
    if (p.nz > 1 && blocks * p.nz > cap) blocks = cap / p.nz > 0 ? cap / p.nz : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), smem, stream, p, h, KP, n_groups);
    return hipGetLastError();
}

template <int FN, bool SHORTK, int NV>
static hipError_t launch_stream_v(const StepArgs& p, const MfmaHints& h, int KP, size_t smem,
                                  hipStream_t stream) {
    if (h.vecA && h.additive32) return launch_stream_t<FN, true, true, SHORTK, NV>(p, h, KP, smem, stream);
    if (h.additive32) return launch_stream_t<FN, false, true, SHORTK, NV>(p, h, KP, smem, stream);
    return launch_stream_t<FN, false, false, SHORTK, NV>(p, h, KP, smem, stream);
}

template <int FN>
static hipError_t launch_stream(const StepArgs& p, const MfmaHints& h, hipStream_t stream) {
    const int KP = (int)((p.K + MFMA_BK - 1) / MFMA_BK) * MFMA_BK;
    const int LDB = KP + 4;
    const size_t smem = (size_t)2 * 2 * 16 * FN * LDB * 4 + (size_t)KP * 8 +
                        (size_t)4 * 2 * 32 * (MFMA_BK + 4) * 4;
    // short contraction: the order table holds only the real columns, 32 K of
    // them per task = the first K / 2 slots of every lane
    if (p.K <= 4) return launch_stream_v<FN, true, 2>(p, h, KP, smem, stream);
    if (p.K <= 8) return launch_stream_v<FN, true, 4>(p, h, KP, smem, stream);
    if (p.K < MFMA_BK) return launch_stream_v<FN, true, 8>(p, h, KP, smem, stream);
    return launch_stream_v<FN, false, 8>(p, h, KP, smem, stream);
}

// ------------------------------------------------------------------------- //
// k-streaming variant: two big tensors contracted into a tiny result (R <= 32,
// N <= 32, K in the millions) -- the last step of every amplitude tree.  Here
// BOTH operands stream.  A wave owns every n_waves-th 16-deep k-chunk: it
// gathers the 32 x 16 tile of A and the 16 x 16*FN tile of B in address order
// (two chunks ahead, in registers), transposes both through wave-private LDS
// into MFMA fragment layout and accumulates one 32 x 16*FN tile for its whole
// share of K.  The per-wave tiles go to the split-K scratch and are summed in
// a fixed order by splitk_reduce_kernel.  Host-checked (MfmaHints::stream ==
// 2): K % 16 == 0, k tables tile-additive per chunk with a power-of-two split,
// 32-bit lane offsets.
// ------------------------------------------------------------------------- //

template <int FN, bool VEC_A>
__global__ __launch_bounds__(256, 3) void pair_mfma_kstream_kernel(StepArgs p, MfmaHints h,
                                                                   float* __restrict__ partial) {
    constexpr int LD = MFMA_BK + 4;
    constexpr int PA = 8;            // A elements per lane per chunk (32 x 16 / 64)
    constexpr int PB = 4 * FN;       // B elements per lane per chunk (16 x 16 FN / 64)
    constexpr int A_FL = 2 * 32 * LD, B_FL = 2 * 16 * FN * LD;
    __shared__ __attribute__((aligned(16))) float lds[4 * (A_FL + B_FL)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5;
    const int l31 = lane & 31;
    const bool negate = kk == 1 && !(lane & 1);
    float* As = lds + wave * (A_FL + B_FL);
    float* Bs = As + A_FL;

    const c64* __restrict__ A = (const c64*)p.A + zoffA_s(p) + sload64(p.rowA.hi);
    const c64* __restrict__ B = (const c64*)p.B + zoffB_s(p);

    // per-lane constants: LDS slot and 32-bit offset of every element this lane moves
    int a_lds[PA], b_lds[PB];
    int a_off[PA], b_off[PB];        // -1: padding (row >= R / column >= N), never gathered
    {
        const int64_t ka0 = p.kA.lo[0], kb0 = p.kB.lo[0];
        const uint16_t* oa = h.ordA + lane * PA;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int v = oa[j] & 0x7fff;
            const int r = v >> 4, c = v & 15;
            a_lds[j] = r * LD + c;
            a_off[j] = r < p.R ? (int)(p.rowA.lo[r] + p.kA.lo[c] - ka0) : -1;
        }
        const uint16_t* ob = h.ordB + lane * PB;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int v = ob[j];
            const int n = v >> 4, c = v & 15;
            b_lds[j] = (2 * n) * LD + c;
            b_off[j] = n < p.N ? (int)(p.nB[n] + p.kB.lo[c] - kb0) : -1;
        }
    }
    for (int i = lane; i < A_FL + B_FL; i += 64) As[i] = 0.f;   // padding reads as zero

    const int64_t n_chunks = p.K / MFMA_BK;
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    auto gather = [&](c64 (&ar)[PA], c64 (&br)[PB], int64_t chunk) {
        const int64_t k = uniform64(chunk * MFMA_BK);
        const int64_t kh = k >> p.k_lo_shift, kl = k & (p.k_lo - 1);
        const c64* Ak = A + sload64(p.kA.hi + kh) + sload64(p.kA.lo + kl);
        const c64* Bk = B + sload64(p.kB.hi + kh) + sload64(p.kB.lo + kl);
        if (VEC_A) {
#pragma unroll
            for (int j = 0; j < PA; j += 2) {
                const f32x4 v = *(const f32x4*)(Ak + a_off[j]);
                ar[j] = c64{v[0], v[1]};
                ar[j + 1] = c64{v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int j = 0; j < PA; ++j)
                if (a_off[j] >= 0) ar[j] = Ak[a_off[j]];
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (b_off[j] >= 0) br[j] = Bk[b_off[j]];
    };

    f32x16 acc[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;

    auto consume = [&](c64 (&ar)[PA], c64 (&br)[PB], int64_t next_chunk) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < PA; ++j)
            if (VEC_A || a_off[j] >= 0) {
                As[a_lds[j]] = ar[j].re;
                As[32 * LD + a_lds[j]] = ar[j].im;
            }
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (b_off[j] >= 0) {
                Bs[b_lds[j]] = br[j].re;
                Bs[b_lds[j] + LD] = br[j].im;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (next_chunk < n_chunks) gather(ar, br, next_chunk);   // refill two chunks ahead
        const float* a_base = As + kk * 32 * LD + l31 * LD;
        const float* b_base = Bs + (l31 ^ kk) * LD;
#pragma unroll
        for (int kq = 0; kq < MFMA_BK / 4; ++kq) {
            const f32x4 af = *(const f32x4*)(a_base + kq * 4);
            f32x4 bf[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 v = *(const f32x4*)(b_base + j * 32 * LD + kq * 4);
                bf[j] = negate ? -v : v;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[j][t], acc[j], 0, 0, 0);
        }
    };

    c64 a0[PA], b0[PB], a1[PA], b1[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) a0[j] = a1[j] = c64{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PB; ++j) b0[j] = b1[j] = c64{0.f, 0.f};
    int64_t c = wave_g;
    if (c < n_chunks) gather(a0, b0, c);
    if (c + n_waves < n_chunks) gather(a1, b1, c + n_waves);
    for (; c < n_chunks; c += 2 * n_waves) {
        consume(a0, b0, c + 2 * n_waves);
        if (c + n_waves < n_chunks) consume(a1, b1, c + 3 * n_waves);
    }

    // this wave's tile -> slab wave_g of the split-K scratch ([S][32][2*16*FN] floats)
    float* slab = partial + wave_g * (32 * 2 * 16 * FN);
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = (t & 3) + 8 * (t >> 2) + 4 * kk;
            slab[row * (2 * 16 * FN) + j * 32 + l31] = acc[j][t];
        }
}

template <int FN>
static hipError_t launch_kstream(const StepArgs& p, const MfmaHints& h, void* scratch,
                                 int64_t scratch_bytes, hipStream_t stream) {
    if (p.nz > 1)  // (per-wave partial tiles live in the one scratch buffer)
        return for_each_z(p, [&](const StepArgs& q) { return launch_kstream<FN>(q, h, scratch, scratch_bytes, stream); });
    const int64_t n_chunks = p.K / MFMA_BK;
    int64_t blocks = 256 * 3;                       // resident: 3 blocks per CU
    // at least 8 chunks per wave: every wave costs a slab in the final reduction
    if (blocks * 32 > n_chunks) blocks = (n_chunks + 31) / 32;
    const int64_t slab_bytes = 32 * 2 * 16 * FN * 4;
    if (blocks * 4 * slab_bytes > scratch_bytes) blocks = scratch_bytes / slab_bytes / 4;
    if (blocks < 1) return hipErrorInvalidValue;
    if (h.vecA)
        hipLaunchKernelGGL((pair_mfma_kstream_kernel<FN, true>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           p, h, (float*)scratch);
    else
        hipLaunchKernelGGL((pair_mfma_kstream_kernel<FN, false>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           p, h, (float*)scratch);
    int64_t rblocks = (p.R * p.N + 31) / 32;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, stream, p, blocks * 4,
                       (int64_t)32, (int64_t)(2 * 16 * FN), (const float*)scratch);
    return hipGetLastError();
}


// ------------------------------------------------------------------------- //
// Skinny steps (K in {2, 4, 8, 16}, N in {1, 2, 4}, K*N <= 16, R huge): pure HBM streams
// whose output tile would use 1/8 or less of a matrix-core tile.  Plain FMAs:
// every thread owns two adjacent rows -- K 16-byte gathers in flight, B and
// the k offsets in scalar registers, 16-byte stores -- so the only thing the
// kernel waits for is HBM.  Host-checked (skinny_ok, ctg_runtime.hip): row
// pairs are contiguous and 16-byte aligned in A and C, n is contiguous in C.
// ------------------------------------------------------------------------- //
template <int KU, int NN>
__global__ __launch_bounds__(256) void pair_skinny_kernel(StepArgs p) {
    const c64* __restrict__ A = (const c64*)p.A + zoffA_s(p);
    const c64* __restrict__ B = (const c64*)p.B + zoffB_s(p);
    c64* __restrict__ C = (c64*)p.C + zoffC_s(p);
    // row offsets: two table lookups per thread (vector loads), issued first
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (row >= p.R) return;
    const int64_t hi = row >> p.row_lo_shift, lo = row & (p.row_lo - 1);
    const int64_t a_hi = p.rowA.hi[hi], a_lo = p.rowA.lo[lo];
    const int64_t c_hi = p.rowC.hi[hi], c_lo = p.rowC.lo[lo];
    // k offsets and B: wave-uniform, all on the scalar unit (no LDS, no barrier)
    const int64_t ka0 = sload64(p.kA.hi), kb0 = sload64(p.kB.hi);
    int64_t ko[KU];
    float br[KU][NN], bi[KU][NN];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
        ko[k] = ka0 + sload64(p.kA.lo + k);
        const int64_t kb = kb0 + sload64(p.kB.lo + k);
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            typedef const float __attribute__((address_space(4))) * cfptr;
            cfptr bp = (cfptr)(uintptr_t)(B + kb + sload64(p.nB + n));
            br[k][n] = bp[0];
            bi[k][n] = bp[1];
        }
    }
    const c64* a = A + a_hi + a_lo;
    f32x4 v[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) v[k] = __builtin_nontemporal_load((const f32x4*)(a + ko[k]));
    const float alpha = (float)step_alpha(p);
    float acc[2][NN][2];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[0][n][0] = acc[0][n][1] = acc[1][n][0] = acc[1][n][1] = 0.f;
#pragma unroll
    for (int k = 0; k < KU; ++k)
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float ar = v[k][2 * r], ai = v[k][2 * r + 1];
                acc[r][n][0] = fmaf(ar, br[k][n], acc[r][n][0]);
                acc[r][n][0] = fmaf(-ai, bi[k][n], acc[r][n][0]);
                acc[r][n][1] = fmaf(ar, bi[k][n], acc[r][n][1]);
                acc[r][n][1] = fmaf(ai, br[k][n], acc[r][n][1]);
            }
    // two rows x NN columns = 2*NN contiguous complex numbers
    float* out = (float*)(C + c_hi + c_lo);
    if (NN == 1) {
        *(f32x4*)out = f32x4{acc[0][0][0] * alpha, acc[0][0][1] * alpha, acc[1][0][0] * alpha, acc[1][0][1] * alpha};
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int n = 0; n < NN; n += 2)
                *(f32x4*)(out + 2 * (r * NN + n)) = f32x4{acc[r][n][0] * alpha, acc[r][n][1] * alpha,
                                                          acc[r][n + 1][0] * alpha, acc[r][n + 1][1] * alpha};
    }
}

template <int KU, int NN>
static hipError_t launch_skinny_t(const StepArgs& p, hipStream_t stream) {
    const int64_t blocks = (p.R / 2 + 255) / 256;
    if (blocks > 0x7fffffffll || p.nz > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((pair_skinny_kernel<KU, NN>), dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p);
    return hipGetLastError();
}

static hipError_t launch_skinny(const StepArgs& p, hipStream_t stream) {
    switch ((int)p.K * 8 + (int)p.N) {
        case 2 * 8 + 1: return launch_skinny_t<2, 1>(p, stream);
        case 4 * 8 + 1: return launch_skinny_t<4, 1>(p, stream);
        case 8 * 8 + 1: return launch_skinny_t<8, 1>(p, stream);
        case 16 * 8 + 1: return launch_skinny_t<16, 1>(p, stream);
        case 2 * 8 + 2: return launch_skinny_t<2, 2>(p, stream);
        case 4 * 8 + 2: return launch_skinny_t<4, 2>(p, stream);
        case 8 * 8 + 2: return launch_skinny_t<8, 2>(p, stream);
        case 2 * 8 + 4: return launch_skinny_t<2, 4>(p, stream);
        case 4 * 8 + 4: return launch_skinny_t<4, 4>(p, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------- //
// Row-wise kernel for tall steps with a handful of multiply-adds per row and
// no structure to exploit: K <= 32, N <= 32, any extents (3s of hyper networks),
// any layout.  One thread per row: its A and C row offsets through the two-level
// tables (once), K element loads, K * N complex FMAs (two packed v_pk_fma_f32
// each) against B broadcast from LDS, N element stores.  Neighbouring threads are neighbouring rows, so loads
// and stores coalesce whenever a kept index is the fastest in memory.  The
// matrix-core kernels pad such a step to 16 x 16 tiles and, when 32-row groups
// are not base + constant, fall back to per-group table lookups.
// ------------------------------------------------------------------------- //
// TS: the N output columns are the fastest index of C -- the 256 x N results of a
// block go through LDS so that consecutive lanes store consecutive columns
// (contiguous 8-byte stores) instead of one column of 64 different rows
// (measured on the 200-tensor hyper network: 2.4 -> 4.6 TB/s at N = 8).  The
// same for the loads when a contracted index is the fastest one of A measured
// slower than the plain per-thread loads (L1 serves the k-neighbours) and is
// not built.  blockIdx.z: batch index of the step (folding a batch index that is
// the fastest one in memory into the thread index was tried as well: no gain).
template <int NN, bool TS>
__global__ __launch_bounds__(256) void pair_rowwise_kernel(StepArgs p) {
    constexpr int KMAX = 32;
    typedef float v2f __attribute__((ext_vector_type(2)));
    // B as (Re, Im, -Im, Re): one complex multiply-add = two packed FMAs (v_pk_fma_f32)
    // on a (Re, Im) accumulator, a.re * (b.re, b.im) + a.im * (-b.im, b.re)
    __shared__ f32x4 Bs[KMAX * NN];
    __shared__ int64_t kofs[KMAX];
    __shared__ int64_t ncol[NN];
    __shared__ int64_t roff[TS ? 256 : 1];
    extern __shared__ float2 tile[];   // TS: 256 rows of N | 1 results (odd stride: two-way bank conflicts at most)
    const int64_t bz = blockIdx.z;
    const c64* __restrict__ A = (const c64*)p.A + zoffA(p) + p.bA[bz];
    const c64* __restrict__ B = (const c64*)p.B + zoffB(p) + p.bB[bz];
    c64* __restrict__ C = (c64*)p.C + zoffC(p) + p.bC[bz];
    const int tid = threadIdx.x;
    const int K = (int)p.K, N = (int)p.N;
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int64_t row = row0 + tid;
    const bool live = row < p.R;
    int64_t a_off = 0, c_off = 0;
    if (live) {   // (issued first: the longest dependent chain of the thread)
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        a_off = p.rowA.hi[hi] + p.rowA.lo[lo];
        c_off = p.rowC.hi[hi] + p.rowC.lo[lo];
    }
    for (int e = tid; e < K * NN; e += 256) {
        const int k = e / NN, n = e - k * NN;
        c64 v{0.f, 0.f};
        if (n < N) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            v = B[p.nB[n] + p.kB.hi[kh] + p.kB.lo[kl]];
        }
        Bs[e] = f32x4{v.re, v.im, -v.im, v.re};
    }
    if (tid < K) {
        int64_t kh, kl;
        split_k(p, tid, kh, kl);
        kofs[tid] = p.kA.hi[kh] + p.kA.lo[kl];
    }
    if (tid >= 64 && tid < 64 + NN) ncol[tid - 64] = tid - 64 < N ? p.nC[tid - 64] : 0;
    __syncthreads();
    const float alpha = (float)step_alpha(p);
    v2f acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = v2f{0.f, 0.f};
    if (live) {
        const c64* a = A + a_off;
        for (int k0 = 0; k0 < K; k0 += 8) {
            c64 av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = k0 + j < K ? a[kofs[k0 + j]] : c64{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j >= K) break;   // (wave-uniform)
                const f32x4* b = Bs + (k0 + j) * NN;
                const v2f ar = {av[j].re, av[j].re}, ai = {av[j].im, av[j].im};
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const f32x4 bv = b[n];
                    acc[n] = __builtin_elementwise_fma(ar, v2f{bv[0], bv[1]}, acc[n]);
                    acc[n] = __builtin_elementwise_fma(ai, v2f{bv[2], bv[3]}, acc[n]);
                }
            }
        }
    }
    if (TS) {
        const int rows_here = (int)(p.R - row0 < 256 ? p.R - row0 : 256);
        const int CS = N | 1;
        roff[tid] = c_off;
#pragma unroll
        for (int n = 0; n < NN; ++n)
            if (n < N) tile[tid * CS + n] = float2{acc[n][0] * alpha, acc[n][1] * alpha};
        __syncthreads();
        const unsigned inv = 0xffffffffu / (unsigned)N + 1u;   // exact e / N for e < 2^16 (N >= 2)
        const int total = rows_here * N;
        for (int e = tid; e < total; e += 256) {
            const int r = (int)__umulhi((unsigned)e, inv), n = e - r * N;
            const float2 v = tile[r * CS + n];
            C[roff[r] + ncol[n]] = c64{v.x, v.y};
        }
    } else if (live) {
        c64* c = C + c_off;
#pragma unroll
        for (int n = 0; n < NN; ++n)
            if (n < N) c[ncol[n]] = c64{acc[n][0] * alpha, acc[n][1] * alpha};
    }
}

bool rowwise_ok(const StepArgs& p) {
    return p.Bt >= 1 && p.Bt <= 65535 && p.K >= 1 && p.K <= 32 && p.N >= 1 && p.N <= 32;
}

template <int NN>
static hipError_t launch_rowwise_t(const StepArgs& p, bool ts, dim3 grid, hipStream_t stream) {
    const size_t lds = ts ? (size_t)256 * (size_t)(p.N | 1) * sizeof(float2) : 0;
    if (ts) {
        // 25-32 columns stage 50-66 KB of results next to 18.5 KB of static LDS: beyond
        // the default limit of a launch, the kernel has to opt in (the CU has 160 KiB)
        static unsigned long long opted = 0;   // (bit per device)
        if (lds > 40 * 1024) {
            const hipError_t e = lds_opt_in((const void*)pair_rowwise_kernel<NN, true>, 96 * 1024, &opted);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((pair_rowwise_kernel<NN, true>), grid, dim3(256), lds, stream, p);
    } else {
        hipLaunchKernelGGL((pair_rowwise_kernel<NN, false>), grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

// flags (MfmaHints::vecA of a row-wise step): bit 0 = the output columns are the
// fastest-varying memory index of C
static hipError_t launch_rowwise(const StepArgs& p, int flags, hipStream_t stream) {
    const int64_t blocks = (p.R + 255) / 256;
    if (blocks > 0x7fffffffll || p.nz > 65535 || !rowwise_ok(p)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks, (unsigned)p.nz, (unsigned)p.Bt);
    const bool ts = (flags & 1) && p.N >= 2;
    if (p.N <= 4) return launch_rowwise_t<4>(p, ts, grid, stream);
    if (p.N <= 8) return launch_rowwise_t<8>(p, ts, grid, stream);
    if (p.N <= 12) return launch_rowwise_t<12>(p, ts, grid, stream);
    if (p.N <= 16) return launch_rowwise_t<16>(p, ts, grid, stream);
    if (p.N <= 24) return launch_rowwise_t<24>(p, ts, grid, stream);
    return launch_rowwise_t<32>(p, ts, grid, stream);
}

template <typename Cfg>
static hipError_t build_lane_t(const StepArgs& p, const MfmaHints& h, void* out, hipStream_t stream) {
    if (h.vecA)
        hipLaunchKernelGGL((fast_lane_consts_kernel<Cfg, true>), dim3(1), dim3(256), 0, stream, p, h, (uint4*)out);
    else
        hipLaunchKernelGGL((fast_lane_consts_kernel<Cfg, false>), dim3(1), dim3(256), 0, stream, p, h, (uint4*)out);
    return hipGetLastError();
}

// Fill the lane-constant table of a fast tiled step (h.fast, h.bn); `out` holds
// fast_lane_table_bytes() bytes.  Called once per step when an executor is built.
int64_t fast_lane_table_bytes() { return 256 * 16 * 8; }   // up to 8 x 16 bytes per thread (128 x 128 tiles: 6)
hipError_t launch_fast_lane_consts(const StepArgs& p, const MfmaHints& h, void* out, hipStream_t stream) {
    switch (h.bn) {
        case 16: return build_lane_t<MfmaCfg<128, 16, 16, 4, 1>>(p, h, out, stream);
        case 32: return build_lane_t<MfmaCfg<128, 32, 16, 4, 1>>(p, h, out, stream);
        case 64: return build_lane_t<MfmaCfg<128, 64, 16, 2, 2>>(p, h, out, stream);
        case 128: return build_lane_t<MfmaCfg<128, 128, 16, 2, 2>>(p, h, out, stream);
    }
    return hipErrorInvalidValue;
}

// ---- several independent small tiled steps in one launch ------------------- //

int fast_group_key(const StepArgs& p, const MfmaHints& h) {
    if (h.stream != 0 || !h.fast || h.splitk > 1 || p.Bt != 1) return -1;
    if (h.bn != 16 && h.bn != 32 && h.bn != 64) return -1;
    const int64_t tiles = ((p.R + MFMA_BM - 1) / MFMA_BM) * ((p.N + h.bn - 1) / h.bn);
    if (tiles > kFastGroupMaxTiles || p.K < MFMA_BK) return -1;
    return h.bn * 2 + (h.vecA ? 1 : 0);
}

uint32_t fast_group_fill(const StepArgs& p, const MfmaHints& h, FastGroupItem* it, uint32_t block_begin) {
    it->p = p;
    it->h = h;
    it->tiles_m = (p.R + MFMA_BM - 1) / MFMA_BM;
    it->tiles_n = (p.N + h.bn - 1) / h.bn;
    it->block_begin = block_begin;
    it->n_blocks = (uint32_t)(it->tiles_m * it->tiles_n);
    return it->n_blocks;
}

template <typename Cfg>
static hipError_t launch_fast_group_t(bool vec, const FastGroupItem* d_items, int n_items, uint32_t blocks,
                                      int nz, hipStream_t stream) {
    const dim3 grid(blocks, (unsigned)nz, 1);
    const StepArgs none{};
    const MfmaHints noh{};
    if (vec)
        hipLaunchKernelGGL((pair_mfma_fast_kernel<Cfg, true, true>), grid, dim3(256), 0, stream, none, noh,
                           (int64_t)0, (int64_t)0, (int64_t)1, (float*)nullptr, d_items, n_items);
    else
        hipLaunchKernelGGL((pair_mfma_fast_kernel<Cfg, false, true>), grid, dim3(256), 0, stream, none, noh,
                           (int64_t)0, (int64_t)0, (int64_t)1, (float*)nullptr, d_items, n_items);
    return hipGetLastError();
}

hipError_t launch_pair_mfma_fast_group(int key, const FastGroupItem* d_items, int n_items, uint32_t blocks,
                                       int nz, hipStream_t stream) {
    const bool vec = key & 1;
    switch (key >> 1) {
        case 16: return launch_fast_group_t<MfmaCfg<128, 16, 16, 4, 1>>(vec, d_items, n_items, blocks, nz, stream);
        case 32: return launch_fast_group_t<MfmaCfg<128, 32, 16, 4, 1>>(vec, d_items, n_items, blocks, nz, stream);
        case 64: return launch_fast_group_t<MfmaCfg<128, 64, 16, 2, 2>>(vec, d_items, n_items, blocks, nz, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pair_mfma(int dtype, const StepArgs& p, const MfmaHints& h, void* scratch,
                            int64_t scratch_bytes, hipStream_t stream) {
    if (dtype != 2) return hipErrorInvalidValue;
    if (h.stream == 3) return launch_skinny(p, stream);
    if (h.stream == 4) return launch_rowwise(p, h.vecA, stream);
    if (h.stream == 2) {
        switch (h.bn) {
            case 16: return launch_kstream<1>(p, h, scratch, scratch_bytes, stream);
            case 32: return launch_kstream<2>(p, h, scratch, scratch_bytes, stream);
        }
        return hipErrorInvalidValue;
    }
    if (h.stream) {
        switch (h.bn) {
            case 16: return launch_stream<1>(p, h, stream);
            case 32: return launch_stream<2>(p, h, stream);
            case 64: return launch_stream<4>(p, h, stream);
        }
        return hipErrorInvalidValue;
    }
    switch (h.bn) {
        case 16: return launch_cfg<MfmaCfg<128, 16, 16, 4, 1>>(p, h, scratch, scratch_bytes, stream);
        case 32: return launch_cfg<MfmaCfg<128, 32, 16, 4, 1>>(p, h, scratch, scratch_bytes, stream);
        case 64: return launch_cfg<MfmaCfg<128, 64, 16, 2, 2>>(p, h, scratch, scratch_bytes, stream);
        case 128: return launch_cfg<MfmaCfg<128, 128, 16, 2, 2>>(p, h, scratch, scratch_bytes, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace ctg

#ifdef CTG_TIMING
extern "C" void ctg_debug_timing(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ctg::ctg_timing), 64);
    if (reset) {
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(ctg::ctg_timing), z, 64);
    }
}
#endif

