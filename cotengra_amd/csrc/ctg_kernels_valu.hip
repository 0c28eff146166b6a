// ctg_kernels_valu.hip -- the dtype-generic (f32/f64/c64/c128) gfx950 kernels.
//
//   pair_valu   : gather-GEMM, one thread per output element, k-loop in
//                 registers.  Serves tiny tree steps, outer products /
//                 Hadamards (reference contract.py:122-164, 398-400), skinny
//                 HBM-bound steps and the f64/c128 parity mode.
//   pair_kred   : same op when there are few outputs but a long contracted
//                 dimension (the last steps of an amplitude contraction):
//                 lanes run along k (coalesced) and a wavefront shuffle
//                 reduction + fixed-order partial sum produce each output.
//   single      : single-term einsum -- diagonals, traces, sums, transposes
//                 (reference contract.py:62-119, 332-361).
//   accum       : result[chunk] += slice (reference core.py:3842-3876).
//   prologue    : slice id -> per-leaf base offsets (reference
//                 core.py:3775-3819), advancing the on-device slice counter so
//                 that a whole slice is a static launch sequence.
//
// All loads go through offset tables (ctg_common.h); wave = 64 lanes.
#include "ctg_common.h"

namespace ctg {

// ------------------------------------------------------------------------- //
// pair: thread per output
// ------------------------------------------------------------------------- //

// one workgroup's share (tiles bx, bx + gx, ...) of a thread-per-output step
template <typename T>
__device__ __forceinline__ void pair_valu_body(const StepArgs& p, int tn_shift, int64_t col_tiles,
                                               int64_t n_tiles, int64_t bx, int64_t gx) {
    const T* __restrict__ A = (const T*)p.A + zoffA(p);
    const T* __restrict__ B = (const T*)p.B + zoffB(p);
    T* __restrict__ C = (T*)p.C + zoffC(p);
    const double alpha = step_alpha(p);
    const int TN = 1 << tn_shift;
    const int TR = 256 >> tn_shift;
    const int c = threadIdx.x & (TN - 1);
    const int r = threadIdx.x >> tn_shift;

    for (int64_t tile = bx; tile < n_tiles; tile += gx) {
        const int64_t rt = tile / col_tiles;
        const int64_t ct = tile - rt * col_tiles;
        const int64_t row = rt * TR + r;
        const int64_t n = ct * TN + c;
        if (row >= p.R || n >= p.N) continue;
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        const T* a = A + p.rowA.hi[hi] + p.rowA.lo[lo];
        const T* b = B + p.rowB.hi[hi] + p.rowB.lo[lo] + p.nB[n];
        T acc = zero_of(T{});
        for (int64_t kh = 0; kh < p.k_hi_len; ++kh) {
            const T* a2 = a + p.kA.hi[kh];
            const T* b2 = b + p.kB.hi[kh];
            for (int64_t kl = 0; kl < p.k_lo; ++kl) {
                fma_acc(acc, a2[p.kA.lo[kl]], b2[p.kB.lo[kl]]);
            }
        }
        C[p.rowC.hi[hi] + p.rowC.lo[lo] + p.nC[n]] = scale_of(acc, alpha);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pair_valu_kernel(StepArgs p, int tn_shift,
                                                        int64_t col_tiles, int64_t n_tiles) {
    pair_valu_body<T>(p, tn_shift, col_tiles, n_tiles, blockIdx.x, gridDim.x);
}

// Several INDEPENDENT small steps in one launch (the leaves-upward wave fronts of
// a small tree: dozens of steps of a few microseconds each, every one of which
// would otherwise wait for the previous launch to drain).  Workgroups
// [block_begin, next block_begin) belong to item i; each item is a complete
// step description, computed exactly as its own launch would (same threads,
// same order of operations: the results are bit-identical).
template <typename T>
__global__ __launch_bounds__(256) void pair_valu_group_kernel(const ValuGroupItem* __restrict__ items,
                                                              int n_items) {
    int i = 0;
    while (i + 1 < n_items && blockIdx.x >= items[i + 1].block_begin) ++i;
    i = __builtin_amdgcn_readfirstlane(i);
    const ValuGroupItem& it = items[i];
    pair_valu_body<T>(it.p, it.tn_shift, it.col_tiles, it.n_tiles, blockIdx.x - it.block_begin,
                      it.n_blocks);
}

// ------------------------------------------------------------------------- //
// pair: wavefront reduction along k
// ------------------------------------------------------------------------- //

// work item w = (output o, k-chunk g); one wave per item.
template <typename T>
__global__ __launch_bounds__(256) void pair_kred_kernel(StepArgs p, int64_t G, int64_t chunk,
                                                        T* __restrict__ partial) {
    const T* __restrict__ A = (const T*)p.A + zoffA(p);
    const T* __restrict__ B = (const T*)p.B + zoffB(p);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t items = p.R * p.N * G;

    for (int64_t w = wave; w < items; w += n_waves) {
        const int64_t o = w / G;
        const int64_t g = w - o * G;
        const int64_t row = o / p.N;
        const int64_t n = o - row * p.N;
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        const T* a = A + p.rowA.hi[hi] + p.rowA.lo[lo];
        const T* b = B + p.rowB.hi[hi] + p.rowB.lo[lo] + p.nB[n];
        const int64_t k0 = g * chunk;
        const int64_t k1 = (k0 + chunk < p.K) ? k0 + chunk : p.K;
        T acc = zero_of(T{});
        // four elements per lane and round: the table lookups of all four, then their
        // loads, then the multiply-adds in the order a lane always took them (k, k + 64,
        // ...) -- a quarter of the dependent round trips, the same bits
        constexpr int U = 4;
        for (int64_t k = k0 + lane; k < k1; k += 64 * U) {
            int64_t ia[U], ib[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t ku = k + 64 * u < k1 ? k + 64 * u : k;   // (clamped: a valid address)
                int64_t kh, kl;
                split_k(p, ku, kh, kl);
                ia[u] = p.kA.hi[kh] + p.kA.lo[kl];
                ib[u] = p.kB.hi[kh] + p.kB.lo[kl];
            }
            T av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                av[u] = a[ia[u]];
                bv[u] = b[ib[u]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 64 * u < k1) fma_acc(acc, av[u], bv[u]);
        }
        acc = wave_sum(acc);
        if (lane == 0) partial[(int64_t)blockIdx.y * items + w] = acc;   // (scratch per slice of this launch)
    }
}

// Two to four outputs (the closing dot products of a tree: R x N = 3 x 1 in the 200-tensor hyper
// network, K = 1.1e6): one wave per k-chunk computes ALL of them -- the k offsets of both operands
// are looked up once per k instead of once per k and output (two-level tables: four dependent loads),
// and an operand that does not depend on the output (a zero row stride) is loaded once.  Every output
// sees the same lanes take the same k in the same order as in pair_kred_kernel: the same bits.
template <typename T, int NO>
__global__ __launch_bounds__(256) void pair_kred_multi_kernel(StepArgs p, int64_t G, int64_t chunk,
                                                              T* __restrict__ partial) {
    const T* __restrict__ A = (const T*)p.A + zoffA(p);
    const T* __restrict__ B = (const T*)p.B + zoffB(p);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const T* a[NO];
    const T* b[NO];
    bool same_a = true, same_b = true;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int64_t row = o / p.N, n = o - row * p.N;
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        a[o] = A + p.rowA.hi[hi] + p.rowA.lo[lo];
        b[o] = B + p.rowB.hi[hi] + p.rowB.lo[lo] + p.nB[n];
        same_a = same_a && a[o] == a[0];
        same_b = same_b && b[o] == b[0];
    }
    for (int64_t g = wave; g < G; g += n_waves) {
        const int64_t k0 = g * chunk;
        const int64_t k1 = (k0 + chunk < p.K) ? k0 + chunk : p.K;
        T acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = zero_of(T{});
        constexpr int U = 4;
        for (int64_t k = k0 + lane; k < k1; k += 64 * U) {
            int64_t ia[U], ib[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t ku = k + 64 * u < k1 ? k + 64 * u : k;   // (clamped: a valid address)
                int64_t kh, kl;
                split_k(p, ku, kh, kl);
                ia[u] = p.kA.hi[kh] + p.kA.lo[kl];
                ib[u] = p.kB.hi[kh] + p.kB.lo[kl];
            }
            T av[NO][U], bv[NO][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    av[o][u] = (o > 0 && same_a) ? av[0][u] : a[o][ia[u]];
                    bv[o][u] = (o > 0 && same_b) ? bv[0][u] : b[o][ib[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + 64 * u < k1) {
#pragma unroll
                    for (int o = 0; o < NO; ++o) fma_acc(acc[o], av[o][u], bv[o][u]);
                }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const T r = wave_sum(acc[o]);
            if (lane == 0) partial[((int64_t)blockIdx.y * NO + o) * G + g] = r;   // (the layout pair_kred_finish_kernel reads)
        }
    }
}

// WAVE: few outputs, many partials -- one wavefront per output adds the partials
// (lane l takes l, l + 64, ... in order, then a butterfly: a fixed tree), instead
// of one thread walking up to 256 of them.
template <typename T, bool WAVE>
__global__ __launch_bounds__(256) void pair_kred_finish_kernel(StepArgs p, int64_t G,
                                                               const T* __restrict__ partial) {
    T* __restrict__ C = (T*)p.C + zoffC(p);
    const double alpha = step_alpha(p);
    const int64_t outs = p.R * p.N;
    const int lane = threadIdx.x & 63;
    const int64_t first = WAVE ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t step = WAVE ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x * 256;
    for (int64_t o = first; o < outs; o += step) {
        const int64_t row = o / p.N;
        const int64_t n = o - row * p.N;
        T acc = zero_of(T{});
        const T* src = partial + ((int64_t)blockIdx.y * outs + o) * G;
        if (WAVE) {
            for (int64_t g = lane; g < G; g += 64) acc = add_of(acc, src[g]);
            acc = wave_sum(acc);
            if (lane != 0) continue;
        } else {
            for (int64_t g = 0; g < G; ++g) acc = add_of(acc, src[g]);
        }
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        C[p.rowC.hi[hi] + p.rowC.lo[lo] + p.nC[n]] = scale_of(acc, alpha);
    }
}

template <typename T>
static hipError_t launch_pair_valu_t(const StepArgs& p, void* scratch, int64_t scratch_bytes,
                                     hipStream_t stream) {
    const int64_t outs = p.R * p.N;
    // few outputs, long contraction -> lanes along k
    if (p.K >= 256 && outs <= (1 << 15)) {

        // k per work item: long enough to keep the partial-sum traffic low, short
        // enough that a handful of outputs still spreads over the whole chip
        // (at most 256 partials per output: the finish pass adds them serially)
        int64_t per_item = outs <= 64 ? 512 : 2048;
        if (per_item * 256 < p.K) per_item = (p.K + 255) / 256;
        int64_t G = (p.K + per_item - 1) / per_item;
        const int64_t want = (1 << 14) / (outs > 0 ? outs : 1);  // ~16k waves fill the chip
        if (G > want) G = want;
        if (G < 1) G = 1;
        const int64_t cap = scratch_bytes / (int64_t)sizeof(T) / (outs > 0 ? outs : 1);
        if (G > cap) G = cap;
        if (G >= 1) {
            int64_t chunk = (p.K + G - 1) / G;
            chunk = (chunk + 63) / 64 * 64;
            G = (p.K + chunk - 1) / chunk;
            const int64_t items = outs * G;
            // (G is a function of the step alone; the partial sums of every slice of a
            // batch must fit the scratch buffer, else the slices go one by one)
            const int64_t room = p.scratch_total > scratch_bytes ? p.scratch_total : scratch_bytes;
            if (p.nz > 1 && items * (int64_t)sizeof(T) * p.nz > room)
                return for_each_z_chunk(p, room / (items * (int64_t)sizeof(T)), [&](const StepArgs& q) {
                    return launch_pair_valu_t<T>(q, scratch, scratch_bytes, stream);
                });
            int64_t blocks = (items + 3) / 4;
            if (blocks > 8192) blocks = 8192;
            if (outs >= 2 && outs <= 4) {
                // (a handful of outputs: every wave takes a k-chunk of all of them)
                int64_t mb = (G + 3) / 4;
                if (mb > 8192) mb = 8192;
                const dim3 mg((unsigned)mb, (unsigned)p.nz);
                if (outs == 2) hipLaunchKernelGGL((pair_kred_multi_kernel<T, 2>), mg, dim3(256), 0, stream, p, G, chunk, (T*)scratch);
                else if (outs == 3) hipLaunchKernelGGL((pair_kred_multi_kernel<T, 3>), mg, dim3(256), 0, stream, p, G, chunk, (T*)scratch);
                else hipLaunchKernelGGL((pair_kred_multi_kernel<T, 4>), mg, dim3(256), 0, stream, p, G, chunk, (T*)scratch);
            } else
            hipLaunchKernelGGL(pair_kred_kernel<T>, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p,
                               G, chunk, (T*)scratch);
            if (outs <= 4096 && G >= 16) {
                hipLaunchKernelGGL((pair_kred_finish_kernel<T, true>), dim3((unsigned)((outs + 3) / 4), (unsigned)p.nz),
                                   dim3(256), 0, stream, p, G, (const T*)scratch);
            } else {
                int64_t fblocks = (outs + 255) / 256;
                if (fblocks > 4096) fblocks = 4096;
                hipLaunchKernelGGL((pair_kred_finish_kernel<T, false>), dim3((unsigned)fblocks, (unsigned)p.nz),
                                   dim3(256), 0, stream, p, G, (const T*)scratch);
            }
            return hipGetLastError();
        }
    }
    ValuGroupItem it;
    valu_group_fill(p, &it, 0);
    int64_t blocks = it.n_tiles;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (blocks < 1) blocks = 1;
    if (blocks * p.nz > (1 << 20)) blocks = (1 << 20) / p.nz;
    hipLaunchKernelGGL(pair_valu_kernel<T>, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p,
                       it.tn_shift, it.col_tiles, it.n_tiles);
    return hipGetLastError();
}

bool valu_thread_per_output(const StepArgs& p) { return !(p.K >= 256 && p.R * p.N <= (1 << 15)); }

uint32_t valu_group_fill(const StepArgs& p, ValuGroupItem* it, uint32_t block_begin) {
    int tn_shift = 0;
    while ((1 << tn_shift) < p.N && tn_shift < 8) ++tn_shift;
    const int64_t TN = 1 << tn_shift, TR = 256 >> tn_shift;
    it->p = p;
    it->tn_shift = tn_shift;
    it->col_tiles = (p.N + TN - 1) / TN;
    it->n_tiles = it->col_tiles * ((p.R + TR - 1) / TR);
    it->block_begin = block_begin;
    it->n_blocks = (uint32_t)(it->n_tiles < 1 ? 1 : (it->n_tiles > kValuGroupMaxTiles ? kValuGroupMaxTiles : it->n_tiles));
    return it->n_blocks;
}

hipError_t launch_pair_valu_group(int dtype, const ValuGroupItem* d_items, int n_items, uint32_t blocks,
                                  int nz, hipStream_t stream) {
    const dim3 grid(blocks, (unsigned)nz);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(pair_valu_group_kernel<float>, grid, dim3(256), 0, stream, d_items, n_items); break;
        case 1: hipLaunchKernelGGL(pair_valu_group_kernel<double>, grid, dim3(256), 0, stream, d_items, n_items); break;
        case 2: hipLaunchKernelGGL(pair_valu_group_kernel<c64>, grid, dim3(256), 0, stream, d_items, n_items); break;
        case 3: hipLaunchKernelGGL(pair_valu_group_kernel<c128>, grid, dim3(256), 0, stream, d_items, n_items); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pair_valu(int dtype, const StepArgs& p, void* scratch, int64_t scratch_bytes,
                            hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_pair_valu_t<float>(p, scratch, scratch_bytes, stream);
        case 1: return launch_pair_valu_t<double>(p, scratch, scratch_bytes, stream);
        case 2: return launch_pair_valu_t<c64>(p, scratch, scratch_bytes, stream);
        case 3: return launch_pair_valu_t<c128>(p, scratch, scratch_bytes, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------- //
// single-term einsum
// ------------------------------------------------------------------------- //

template <typename T>
__global__ __launch_bounds__(256) void single_kernel(StepArgs p) {
    const T* __restrict__ A = (const T*)p.A + zoffA(p);
    T* __restrict__ C = (T*)p.C + zoffC(p);
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < p.R;
         row += (int64_t)gridDim.x * 256) {
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        const T* a = A + p.rowA.hi[hi] + p.rowA.lo[lo];
        T acc = zero_of(T{});
        for (int64_t kh = 0; kh < p.k_hi_len; ++kh) {
            const T* a2 = a + p.kA.hi[kh];
            for (int64_t kl = 0; kl < p.k_lo; ++kl) acc = add_of(acc, a2[p.kA.lo[kl]]);
        }
        C[p.rowC.hi[hi] + p.rowC.lo[lo]] = acc;
    }
}

// one wave per output when few outputs are reduced over many elements (full
// traces): lanes along the summed group + wavefront reduction
template <typename T>
__global__ __launch_bounds__(256) void single_wave_kernel(StepArgs p) {
    const T* __restrict__ A = (const T*)p.A + zoffA(p);
    T* __restrict__ C = (T*)p.C + zoffC(p);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t row = wave; row < p.R; row += (int64_t)gridDim.x * 4) {
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        const T* a = A + p.rowA.hi[hi] + p.rowA.lo[lo];
        T acc = zero_of(T{});
        for (int64_t k = lane; k < p.K; k += 64) {
            int64_t kh, kl;
            split_k(p, k, kh, kl);
            acc = add_of(acc, a[p.kA.hi[kh] + p.kA.lo[kl]]);
        }
        acc = wave_sum(acc);
        if (lane == 0) C[p.rowC.hi[hi] + p.rowC.lo[lo]] = acc;
    }
}

template <typename T>
static hipError_t launch_single_t(const StepArgs& p, hipStream_t stream) {
    if (p.K >= 256 && p.R <= (1 << 14)) {
        int64_t blocks = (p.R + 3) / 4;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(single_wave_kernel<T>, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    int64_t blocks = (p.R + 255) / 256;
    if (blocks * p.nz > (1 << 20)) blocks = (1 << 20) / p.nz;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(single_kernel<T>, dim3((unsigned)blocks, (unsigned)p.nz), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_single(int dtype, const StepArgs& p, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_single_t<float>(p, stream);
        case 1: return launch_single_t<double>(p, stream);
        case 2: return launch_single_t<c64>(p, stream);
        case 3: return launch_single_t<c128>(p, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------- //
// accumulate a slice into the result tensor
// ------------------------------------------------------------------------- //

// The slices of a batch are added one after another by the same thread, in slice
// order: the sum is formed exactly as by one launch per slice (the left fold of
// gather_slices, core.py:3842-3844), whichever way the slices were batched.
//
// W (round 5): single-precision results are summed in DOUBLE precision.  The running sum lives in a
// double-precision copy of the result tensor (``wide``, same element offsets) and the result itself
// is that sum rounded once: a left fold of 2^20 slice amplitudes in fp32 loses ~N eps / sqrt(2) of one
// term -- 4e-5 of the m20 amplitude, more than the whole 1e-5 budget -- where the reference has the
// same flaw (core.py:3842-3844 adds in the arrays' dtype).  Double-precision trees: W = T, no copy.
template <typename T> struct wide_of { typedef T type; };
template <> struct wide_of<float> { typedef double type; };
template <> struct wide_of<c64> { typedef c128 type; };
__device__ __forceinline__ double widen(float a) { return (double)a; }
__device__ __forceinline__ double widen(double a) { return a; }
__device__ __forceinline__ c128 widen(c64 a) { return c128{(double)a.re, (double)a.im}; }
__device__ __forceinline__ c128 widen(c128 a) { return a; }
__device__ __forceinline__ void narrow_to(float& d, double a) { d = (float)a; }
__device__ __forceinline__ void narrow_to(double& d, double a) { d = a; }
__device__ __forceinline__ void narrow_to(c64& d, c128 a) { d = c64{(float)a.re, (float)a.im}; }
__device__ __forceinline__ void narrow_to(c128& d, c128 a) { d = a; }

template <typename T>
__global__ __launch_bounds__(256) void accum_kernel(StepArgs p, const StripState* st, void* wide_, const double* inscale) {
    typedef typename wide_of<T>::type W;
    W* const wide = (W*)wide_;   // null: the result itself is the running sum
    // (inscale: the power of two taken out of the input tensors at upload, prescale_inputs_kernel; a
    // strip_exponent run carries it in the exponent instead: strip_prepare_kernel)
    const double coef = st ? st->coefm : (inscale ? inscale[0] : 1.0);
    const bool scaled = st != nullptr || coef != 1.0;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < p.R;
         row += (int64_t)gridDim.x * 256) {
        int64_t hi, lo;
        split_row(p, row, hi, lo);
        const int64_t ra = p.rowA.hi[hi] + p.rowA.lo[lo], rc = p.rowC.hi[hi] + p.rowC.lo[lo];
        // The slices of a batch are added in slice order, as one launch per slice would;
        // their values are fetched eight at a time (independent loads) and a result
        // element that consecutive slices share (inner-sliced indices: all of them) stays
        // in a register between them instead of going through memory 64 times.
        int64_t cur = -1;
        W sum = zero_of(W{});
        auto flush = [&]() {
            if (cur < 0) return;
            if (wide) wide[cur] = sum;
            narrow_to(((T*)p.C)[cur], sum);
        };
        const int64_t z_end = (int64_t)p.z0 + p.nz;
        for (int64_t z0 = p.z0; z0 < z_end; z0 += 8) {
            T av[8];
            int64_t cp[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t z = z0 + i < z_end ? z0 + i : z_end - 1;
                av[i] = ((const T*)p.A + p.soffA[z * p.zsA] + z * p.zA)[ra];
                cp[i] = p.soffC[z * p.zsC] + z * p.zC + rc;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (z0 + i >= z_end) break;
                if (cp[i] != cur) {
                    flush();
                    cur = cp[i];
                    sum = wide ? wide[cur] : widen(((const T*)p.C)[cur]);
                }
                sum = add_of(sum, scaled ? scale_of(widen(av[i]), coef) : widen(av[i]));
            }
        }
        flush();
    }
}

hipError_t launch_accum(int dtype, const StepArgs& p, const StripState* st, void* wide, const double* inscale, hipStream_t stream) {
    int64_t blocks = (p.R + 255) / 256;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    if (blocks < 1) blocks = 1;
    switch (dtype) {
        case 0: hipLaunchKernelGGL(accum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, p, st, wide, inscale); break;
        case 1: hipLaunchKernelGGL(accum_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, p, st, nullptr, inscale); break;
        case 2: hipLaunchKernelGGL(accum_kernel<c64>, dim3((unsigned)blocks), dim3(256), 0, stream, p, st, wide, inscale); break;
        case 3: hipLaunchKernelGGL(accum_kernel<c128>, dim3((unsigned)blocks), dim3(256), 0, stream, p, st, nullptr, inscale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// result <- the double-precision running sum, rounded (after the sum was changed from outside the accumulate
// kernel: the collective, set_state); wide <- result widened (a state given in the result's own precision)
template <typename T>
__global__ __launch_bounds__(256) void narrow_kernel(T* __restrict__ x, const typename wide_of<T>::type* __restrict__ w, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) narrow_to(x[i], w[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void widen_kernel(typename wide_of<T>::type* __restrict__ w, const T* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] = widen(x[i]);
}
hipError_t launch_narrow(int dtype, void* result, const void* wide, int64_t n, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    if (dtype == 0) hipLaunchKernelGGL(narrow_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)result, (const double*)wide, n);
    else if (dtype == 2) hipLaunchKernelGGL(narrow_kernel<c64>, dim3((unsigned)blocks), dim3(256), 0, stream, (c64*)result, (const c128*)wide, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_widen(int dtype, void* wide, const void* result, int64_t n, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    if (dtype == 0) hipLaunchKernelGGL(widen_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (double*)wide, (const float*)result, n);
    else if (dtype == 2) hipLaunchKernelGGL(widen_kernel<c64>, dim3((unsigned)blocks), dim3(256), 0, stream, (c128*)wide, (const c64*)result, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// input tensors far from 1: an exact power of two comes out at upload
// ------------------------------------------------------------------------- //

// Single-precision trees.  An input whose largest |element| lies outside [2^-32, 2^32) is multiplied by the
// power of two that brings it back to the edge of that window -- exact -- and the powers taken out are summed; the accumulate step
// multiplies them back in (double precision), a strip_exponent run adds them to its exponent.  Inputs in the
// range are not touched (bit-identical results).  What this buys: the reference normalises after EVERY step
// under strip_exponent (contract.py:816-829); here normalisation is lazy -- the consumer's epilogue scales --
// so two raw inputs below 2^-40 meeting in one fused pair would underflow the fp32 intermediate before any
// scale is applied, and the bf16 x 3 split loses its third limb on an operand below 2^-110.  After this pass
// every input's largest element is within 2^+-32 of 1.  One workgroup per input (they are KBs).
__device__ __forceinline__ double abs_max_part(float a) { return fabs((double)a); }
__device__ __forceinline__ double abs_max_part(c64 a) { return fmax(fabs((double)a.re), fabs((double)a.im)); }

template <typename T>
__global__ __launch_bounds__(256) void prescale_inputs_kernel(T* inputs, const int64_t* offs, const int64_t* sizes,
                                                              int* shift_total) {
    __shared__ double red[256];
    T* x = inputs + offs[blockIdx.x];
    const int64_t n = sizes[blockIdx.x];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double a = abs_max_part(x[i]);
        if (a > mx && a < __longlong_as_double(0x7ff0000000000000ll)) mx = a;   // (inf / nan: left alone)
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && red[threadIdx.x + o] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + o];
        __syncthreads();
    }
    mx = red[0];
    if (mx == 0.0) return;
    int ex;
    (void)frexp(mx, &ex);
    ex -= 1;   // mx = m 2^ex with m in [1, 2)
    if (ex >= -32 && ex < 32) return;
    // the SMALLEST shift that brings the largest element into the window -- to [2^31, 2^32) from above, to
    // [2^-32, 2^-31) from below: an input with a wide range of its own (rows 2^80 apart) keeps as many of its
    // small elements -- and of the products they enter -- inside the fp32 range as the window allows
    const int shift = ex >= 32 ? ex - 31 : ex + 32;
    const double f = ldexp(1.0, -shift);
    for (int64_t i = threadIdx.x; i < n; i += 256) x[i] = scale_of(x[i], f);
    if (threadIdx.x == 0) atomicAdd(shift_total, shift);
}

__global__ void prescale_finish_kernel(const int* shift_total, double* inscale) {
    inscale[0] = ldexp(1.0, *shift_total);                 // 2^S (inf / 0 beyond the double range, as the product would be)
    inscale[1] = (double)*shift_total * 0.30102999566398120;   // S log10(2)
}

hipError_t launch_prescale_inputs(int dtype, void* inputs, const int64_t* offs, const int64_t* sizes, int64_t n_inputs,
                                  int* shift_total, double* inscale, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(shift_total, 0, sizeof(int), stream);
    if (err != hipSuccess) return err;
    if (dtype == 0)
        hipLaunchKernelGGL(prescale_inputs_kernel<float>, dim3((unsigned)n_inputs), dim3(256), 0, stream, (float*)inputs, offs, sizes, shift_total);
    else if (dtype == 2)
        hipLaunchKernelGGL(prescale_inputs_kernel<c64>, dim3((unsigned)n_inputs), dim3(256), 0, stream, (c64*)inputs, offs, sizes, shift_total);
    else
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(prescale_finish_kernel, dim3(1), dim3(1), 0, stream, shift_total, inscale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// strip_exponent support (reference contract.py:816-829, core.py:125-172)
// ------------------------------------------------------------------------- //

// fac = max |x_i|: wavefront max + one atomicMax per wave on the bit pattern of
// the non-negative double (order preserving)
template <typename T>
__global__ __launch_bounds__(256) void maxabs_kernel(const T* __restrict__ x, int64_t n, double* fac) {
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double a = abs_of(x[i]);
        m = a > m || a != a ? a : m;  // propagate nan
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_down(m, o, 64);
        m = other > m || other != other ? other : m;
    }
    if ((threadIdx.x & 63) == 0)
        atomicMax((unsigned long long*)fac, (unsigned long long)__double_as_longlong(m));
}

hipError_t launch_maxabs(int dtype, const void* x, int64_t n, double* fac, hipStream_t stream) {
    int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    switch (dtype) {
        case 0: hipLaunchKernelGGL(maxabs_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)x, n, fac); break;
        case 1: hipLaunchKernelGGL(maxabs_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, (const double*)x, n, fac); break;
        case 2: hipLaunchKernelGGL(maxabs_kernel<c64>, dim3((unsigned)blocks), dim3(256), 0, stream, (const c64*)x, n, fac); break;
        case 3: hipLaunchKernelGGL(maxabs_kernel<c128>, dim3((unsigned)blocks), dim3(256), 0, stream, (const c128*)x, n, fac); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// largest |re|, |im| of n complex64 values as a FLOAT: the power-of-two scale of a big operand of the fp16 x 2 stem
// kernels that no producer recorded (CTG_STEM_H2_ALL=1: tests and diagnostics -- by default such a pair is multiplied in
// bf16 x 3 instead, ctg_runtime.hip); *out must be zero
// (gridDim.y slices from z on, one record of kMaxSub floats each: out + blockIdx.y * out_zs)
__global__ __launch_bounds__(256) void maxabs_f32_kernel(const c64* __restrict__ base, const int64_t* soff, int64_t z,
                                                         int64_t zs, int64_t zstride, int64_t n, float* out, int out_zs) {
    z += blockIdx.y;
    out += (int64_t)blockIdx.y * out_zs;
    const c64* __restrict__ x = base + ((soff != nullptr ? soff[z * zs] : 0) + z * zstride);   // (no table: the tensor at `base`)
    float m = 0.f;
    if ((((uintptr_t)x) & 15) == 0) {
        // 16-byte loads, four in flight per thread
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* __restrict__ x4 = (const f4*)x;
        const int64_t n4 = n >> 1, step = (int64_t)gridDim.x * 256;
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * step < n4; i += 4 * step) {
            const f4 a = x4[i], b = x4[i + step], c = x4[i + 2 * step], d = x4[i + 3 * step];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))));
        }
        for (; i < n4; i += step) {
            const f4 a = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) m = fmaxf(m, fmaxf(fabsf(x[n - 1].re), fabsf(x[n - 1].im)));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const c64 v = x[i];
            m = fmaxf(m, fmaxf(fabsf(v.re), fabsf(v.im)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f && m < __builtin_bit_cast(float, 0x7f800000u))
        record_max(out, m);
}

hipError_t launch_maxabs_f32(const void* base, const int64_t* soff, int64_t z, int64_t zs, int64_t zstride, int64_t n,
                             float* out, hipStream_t stream, int nz, int out_zs) {
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = nz > 1 ? std::max<int64_t>(8192 / nz, 16) : 8192;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (nz < 1 || nz > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(maxabs_f32_kernel, dim3((unsigned)blocks, (unsigned)nz), dim3(256), 0, stream, (const c64*)base, soff, z, zs,
                       zstride, n, out, out_zs);
    return hipGetLastError();
}

// slice exponent and the coefficients of the exponent-aware accumulate
// (AdderWithMaybeExponentStripped, core.py:125-172): E' = max(E, e),
// result = result * 10^(E-E') + slice * 10^(e-E') / fac_root
__global__ void strip_prepare_kernel(const double* fac, const int32_t* counted, int64_t n_steps,
                                     int64_t root_step, int check_zero, StripState* st, const double* inscale) {
    double e = inscale ? inscale[1] : 0.0;   // log10 of the power of two taken out of the inputs at upload
    bool zero = false;
    for (int64_t s = 0; s < n_steps; ++s) {
        if (!counted[s]) continue;
        const double f = fac[s];
        if (f == 0.0) zero = true;
        e += log10(f);
    }
    const double froot = root_step >= 0 ? fac[root_step] : 1.0;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    if (zero && check_zero) e = -inf;
    const double E = st->E;
    const double En = e > E ? e : E;  // nan e (not check_zero) compares false: keeps E, coefm nan below
    st->e_slice = e;
    st->zero = zero ? 1 : 0;
    if (En == -inf) {
        st->coefM = 1.0;
        st->coefm = 0.0;
    } else {
        st->coefM = E == -inf ? 0.0 : pow(10.0, E - En);
        st->coefm = (zero && check_zero) ? 0.0 : pow(10.0, e - En) / froot;
    }
    st->E = En;
}

hipError_t launch_strip_prepare(const double* fac, const int32_t* counted, int64_t n_steps,
                                int64_t root_step, int check_zero, StripState* st, const double* inscale, hipStream_t stream) {
    hipLaunchKernelGGL(strip_prepare_kernel, dim3(1), dim3(1), 0, stream, fac, counted, n_steps,
                       root_step, check_zero, st, inscale);
    return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void rescale_kernel(T* __restrict__ x, int64_t n, const StripState* st) {
    const double c = st->coefM;
    if (c == 1.0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = c == 0.0 ? zero_of(T{}) : scale_of(x[i], c);
}

hipError_t launch_rescale(int dtype, void* result, int64_t n, const StripState* st, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    switch (dtype) {
        case 0: hipLaunchKernelGGL(rescale_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)result, n, st); break;
        case 1: hipLaunchKernelGGL(rescale_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, stream, (double*)result, n, st); break;
        case 2: hipLaunchKernelGGL(rescale_kernel<c64>, dim3((unsigned)blocks), dim3(256), 0, stream, (c64*)result, n, st); break;
        case 3: hipLaunchKernelGGL(rescale_kernel<c128>, dim3((unsigned)blocks), dim3(256), 0, stream, (c128*)result, n, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// slice prologue
// ------------------------------------------------------------------------- //

__global__ __launch_bounds__(256) void prologue_kernel(SliceMeta m, int64_t* state, int64_t* soff,
                                                       int64_t sid_arg, int nz, int64_t stride, const int64_t* ids) {
    // slices sid, sid + stride, ... (nz of them): soff[z * n_leaves + leaf]
    // (several workgroups when the slice id comes from the host: 64 slices x hundreds of
    // leaves x dozens of sliced indices is 0.4 ms of dependent divisions for one of them)
    const int64_t sid0 = sid_arg >= 0 ? sid_arg : state[0];
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < m.n_leaves * nz;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t z = w / m.n_leaves, leaf = w - z * m.n_leaves;
        int64_t rem = ids ? ids[z] : sid0 + z * stride;
        int64_t off = 0;
        const int64_t* st = m.strides + leaf * m.n_sliced;
        for (int64_t j = m.n_sliced - 1; j >= 0; --j) {
            int64_t digit;
            if (m.fixed[j] >= 0) {
                digit = m.fixed[j];
            } else {
                const int64_t d = m.sizes[j];
                const int64_t q = rem / d;
                digit = rem - q * d;
                rem = q;
            }
            off += digit * st[j];
        }
        soff[w] = off;
    }
    if (m.fac && blockIdx.x == 0)
        for (int64_t i = threadIdx.x; i < m.n_fac; i += blockDim.x)
            if (m.fac_zero[i]) m.fac[i] = 0.0;
    if (m.smax)   // (every block its share: a batch of 64 slices has 3 x steps x 64 of them)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m.n_smax; i += (int64_t)gridDim.x * blockDim.x)
            if (m.smax_zero[i]) m.smax[i] = 0.f;
    __syncthreads();
    // (the device-side slice counter of a graph replay: always a single workgroup)
    if (threadIdx.x == 0 && sid_arg < 0) state[0] = sid0 + state[1];
}

__global__ void set_state_kernel(int64_t* state, int64_t next, int64_t stride) {
    state[0] = next;
    state[1] = stride;
}

hipError_t launch_set_state(int64_t* state, int64_t next, int64_t stride, hipStream_t stream) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, stream, state, next, stride);
    return hipGetLastError();
}

hipError_t launch_prologue(const SliceMeta& m, int64_t* state, int64_t* soff, int64_t sid,
                           hipStream_t stream, int nz, int64_t stride, const int64_t* ids) {
    int64_t blocks = sid < 0 ? 1 : (m.n_leaves * nz + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, m, state, soff, sid, nz,
                       stride, ids);
    return hipGetLastError();
}

}  // namespace ctg
