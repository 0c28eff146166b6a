// ctg_lds_run.hip -- LDS-resident subtrees: the small-tree execution model (round 6).
//
// The reference walks a contraction tree one pairwise step at a time (cotengra/contract.py:788-832); a
// step on tensors of a few hundred elements cost this executor what every launch costs -- a launch plus
// a chain of dependent global loads, 5-12 us -- whatever it computed.  Here ONE workgroup executes a
// whole subtree ("component", plan: cotengra_amd/ldsrun.py) whose tensors all fit the 160 KB of LDS of a
// gfx950 compute unit:
//
//   phase 0   the component's leaves (and what it reads of slice-invariant / group-shared results) are
//             gathered from global memory into LDS, each in the layout its consumer wants; a leaf's
//             own preprocessing (diagonal / trace / sum, contract.py:62-119) happens in the gather;
//   phase p   the pair steps whose operands are ready, LDS -> LDS; a workgroup barrier between phases;
//   root      the last step writes the subtree's result into the arena (in the ordinary step's layout).
//
// All components of a tree are workgroups of the same launch (blockIdx.x), the slices of a batch its
// blockIdx.y.  The component's step records and offset tables (16-bit LDS offsets) are copied into LDS
// with one coalesced pass before anything else, so that no step waits for a table in global memory.
//
// A pair step: lane = one row r (consecutive lanes = consecutive rows) x TN columns.  The rows operand
// is stored [contracted..., rows...] by its producer, so the 64 lanes of a wavefront read 64 consecutive
// LDS words at every k (conflict-free); the other operand is read as a broadcast.  Small steps of one
// phase are dealt to single wavefronts (they run side by side), large ones are shared by all eight.
// The arithmetic is pair_valu_kernel's: one accumulator per output, k ascending, fused multiply-adds --
// a component gives the same bits as its steps launched one by one on that kernel.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "ctg_lds.h"

namespace ctg {

namespace {

struct RowEnt { uint32_t ab, c; };

#ifdef CTG_LDS_PROBE   // (experiment build: shader-clock stamps inside the matrix-core steps of component 0, wave 0)
__device__ unsigned long long g_lds_probe[512];
__device__ int g_lds_probe_n;
#define LDS_PROBE(tag)                                                                              \
    do {                                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g_lds_probe_n < 510) {        \
            g_lds_probe[g_lds_probe_n++] = ((unsigned long long)(tag) << 48) | (clock64() & 0xffffffffffffull); \
        }                                                                                           \
    } while (0)
#else
#define LDS_PROBE(tag) do {} while (0)
#endif

__device__ __forceinline__ void split_row32(const LdsStepDev& st, uint32_t r, uint32_t& hi, uint32_t& lo) {
    if (st.row_shift >= 0) {
        hi = r >> st.row_shift;
        lo = r & (uint32_t)(st.row_lo - 1);
    } else {
        hi = __umulhi(r, st.row_magic);
        lo = r - hi * (uint32_t)st.row_lo;
    }
}

__device__ __forceinline__ int64_t global_base(const LdsStepDev& st, int64_t z) {
    const int64_t zq = st.gzq > 1 ? z / st.gzq * st.gzq : z;
    return st.gsoff[zq * st.gzs] + zq * st.gz;
}

// global -> LDS: out[rowC(o)] = sum_k src[rowA(o) + kA(k)]
template <typename T>
__device__ __forceinline__ void run_load(const LdsStepDev& st, const char* blob, T* data, int64_t z, int lane, int n_lanes) {
    const T* __restrict__ src = (const T*)st.gptr + global_base(st, z);
    const RowEnt* __restrict__ rhi = (const RowEnt*)(blob + st.t_row_hi);
    const RowEnt* __restrict__ rlo = (const RowEnt*)(blob + st.t_row_lo);
    const uint32_t* __restrict__ kt = (const uint32_t*)(blob + st.t_k);
    T* __restrict__ dst = data + st.c_off;
    if (st.K == 1) {
        // a plain gather: four independent loads in flight per lane
        const uint32_t k0 = kt[0];
        for (int o = lane; o < st.R; o += 4 * n_lanes) {
            T v[4];
            uint32_t c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ou = o + u * n_lanes < st.R ? o + u * n_lanes : o;
                uint32_t hi, lo;
                split_row32(st, (uint32_t)ou, hi, lo);
                const RowEnt h = rhi[hi], l = rlo[lo];
                v[u] = src[(int64_t)h.ab + l.ab + k0];
                c[u] = h.c + l.c;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (o + u * n_lanes < st.R) dst[c[u]] = v[u];
        }
        return;
    }
    for (int o = lane; o < st.R; o += n_lanes) {
        uint32_t hi, lo;
        split_row32(st, (uint32_t)o, hi, lo);
        const RowEnt h = rhi[hi], l = rlo[lo];
        const T* a = src + ((int64_t)h.ab + l.ab);
        T acc = zero_of(T{});
        for (int k = 0; k < st.K; ++k) acc = add_of(acc, a[kt[k]]);
        dst[h.c + l.c] = acc;
    }
}

// LDS x LDS -> LDS (or the arena): C[rowC(r) + nC(n)] = sum_k A[rowA(r) + kA(k)] * B[rowB(r) + kB(k) + nB(n)]
template <typename T, int TN>
__device__ __forceinline__ void run_pair(const LdsStepDev& st, const char* blob, T* data, int64_t z, int lane, int n_lanes) {
    const T* __restrict__ A = data + st.a_off;
    const T* __restrict__ B = data + st.b_off;
    T* __restrict__ C = st.c_off >= 0 ? data + st.c_off : (T*)st.gptr + global_base(st, z);
    const RowEnt* __restrict__ rhi = (const RowEnt*)(blob + st.t_row_hi);
    const RowEnt* __restrict__ rlo = (const RowEnt*)(blob + st.t_row_lo);
    const uint32_t* __restrict__ kt = (const uint32_t*)(blob + st.t_k);
    const uint32_t* __restrict__ nt = (const uint32_t*)(blob + st.t_n);
    const int R = st.R, K = st.K, N = st.N;
    const int NG = (N + TN - 1) / TN;
    const int items = R * NG;
    for (int it = lane; it < items; it += n_lanes) {
        const int g = it / R;
        const int r = it - g * R;
        uint32_t hi, lo;
        split_row32(st, (uint32_t)r, hi, lo);
        const RowEnt h = rhi[hi], l = rlo[lo];
        const uint32_t ab = h.ab + l.ab;
        const T* a = A + (ab & 0xffffu);
        const T* b = B + (ab >> 16);
        uint32_t nbc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) nbc[j] = nt[g * TN + j < N ? g * TN + j : N - 1];
        uint32_t nb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) nb[j] = nbc[j] & 0xffffu;
        T acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = zero_of(T{});
        // four k at a time: their table entries in one 16-byte read, then every operand load, then the
        // multiply-adds in ascending k (the order of pair_valu_kernel: the same bits) -- one LDS round trip
        // per four k instead of two per k
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            const uint4 kk4 = *(const uint4*)(kt + k);
            const uint32_t kk[4] = {kk4.x, kk4.y, kk4.z, kk4.w};
            T av[4], bv[4][TN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = a[kk[u] & 0xffffu];
                const T* bk = b + (kk[u] >> 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[u][j] = bk[nb[j]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < TN; ++j) fma_acc(acc[j], av[u], bv[u][j]);
        }
        for (; k < K; ++k) {
            const uint32_t kk = kt[k];
            const T av = a[kk & 0xffffu];
            const T* bk = b + (kk >> 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fma_acc(acc[j], av, bk[nb[j]]);
        }
        T* c = C + (h.c + l.c);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (g * TN + j < N) c[nbc[j] >> 16] = acc[j];
    }
}

// The same step on the matrix cores (complex64; a GEMM: B does not depend on the row).  A wave task = 32 rows x
// 16 complex columns on v_mfma_f32_32x32x2_f32, the complex product as a real one: row r of A is (Re a_k, Im a_k)
// along 2 K real k, column n of B the two real columns (Re b, Im b | -Im b, Re b) -- lanes 0-31 feed Re a /
// the (Re b, Im b) row, lanes 32-63 Im a / the (-Im b, Re b) row, one instruction per complex k.  The rows
// operand is stored rows-fastest and the other one columns-fastest (ldsrun.py: layout_in_lds), so both
// fragment loads of a k read 64 consecutive LDS words.  Every output sees its products added in ascending k,
// Re a Re b before Im a (-Im b) -- the order of the fused multiply-adds of run_pair.
typedef float lds_f32x16 __attribute__((ext_vector_type(16)));

// a = this lane's element of the rows operand (row l & 31, Re / Im by l >> 5), b = its element of the real form
// of the other operand (real column l & 31, row l >> 5).  Columns on the lanes: D[row][2 n + part] -- a lane
// holds one real column, its registers 16 rows.  Rows on the lanes (the operands swapped: D^T): a lane holds
// one ROW, its registers 16 real columns = 8 complex numbers (Re, Im in adjacent registers).
template <bool ROWS_ON_LANES>
__device__ __forceinline__ lds_f32x16 mfma_step(float a, float b, lds_f32x16 acc) {
    return ROWS_ON_LANES ? __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

template <bool ROWS_ON_LANES>
__device__ __forceinline__ void run_pair_mfma(const LdsStepDev& st, const char* blob, c64* data, int64_t z, int wave, int n_waves) {
    const float* __restrict__ F = (const float*)data;
    float* __restrict__ C = st.c_off >= 0 ? (float*)(data + st.c_off) : (float*)((c64*)st.gptr + global_base(st, z));
    const RowEnt* __restrict__ rhi = (const RowEnt*)(blob + st.t_row_hi);
    const RowEnt* __restrict__ rlo = (const RowEnt*)(blob + st.t_row_lo);
    const uint32_t* __restrict__ kt = (const uint32_t*)(blob + st.t_k);
    const uint32_t* __restrict__ nt = (const uint32_t*)(blob + st.t_n);
    const int R = st.R, K = st.K, N = st.N;
    const int tiles_r = (R + 31) >> 5, tiles_n = (N + 15) >> 4;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, kp = lane >> 5, part = lane & 1;
    // (the steps of one phase start at different waves: st.rot = tasks of the phase's earlier steps)
    for (int task = (wave + n_waves - st.rot % n_waves) % n_waves; task < tiles_r * tiles_n; task += n_waves) {
        LDS_PROBE(0);
        const int tn = task / tiles_r, tr = task - tn * tiles_r;
        // this lane's row of A, its column of B
        const int r = tr * 32 + l31 < R ? tr * 32 + l31 : R - 1;
        uint32_t hi, lo;
        split_row32(st, (uint32_t)r, hi, lo);
        const uint32_t a_off = (rhi[hi].ab + rlo[lo].ab) & 0xffffu;
        const int n = tn * 16 + (l31 >> 1) < N ? tn * 16 + (l31 >> 1) : N - 1;
        const uint32_t nbc = nt[n];
        const uint32_t fa = ((uint32_t)st.a_off + a_off) * 2u + (uint32_t)kp;
        const uint32_t fb = ((uint32_t)st.b_off + (nbc & 0xffffu)) * 2u + (uint32_t)(part ^ kp);
        const float sgn = (kp == 1 && part == 0) ? -1.f : 1.f;
        lds_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        LDS_PROBE(1);
        int k = 0;
        // sixteen k at a time: their table entries (four 16-byte reads), then the 32 fragment loads, then the
        // sixteen matrix instructions -- two LDS round trips per sixteen k on the critical path of the wave
        for (; k + 16 <= K; k += 16) {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *(const uint4*)(kt + k + 4 * u);
            float av[16], bv[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t kk[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    av[4 * u + v] = F[fa + 2u * (kk[v] & 0xffffu)];
                    bv[4 * u + v] = F[fb + 2u * (kk[v] >> 16)];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = mfma_step<ROWS_ON_LANES>(av[u], bv[u] * sgn, acc);
        }
        for (; k + 4 <= K; k += 4) {
            const uint4 kk4 = *(const uint4*)(kt + k);
            const uint32_t kk[4] = {kk4.x, kk4.y, kk4.z, kk4.w};
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = F[fa + 2u * (kk[u] & 0xffffu)];
                bv[u] = F[fb + 2u * (kk[u] >> 16)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mfma_step<ROWS_ON_LANES>(av[u], bv[u] * sgn, acc);
        }
        for (; k < K; ++k) {
            const uint32_t kk = kt[k];
            acc = mfma_step<ROWS_ON_LANES>(F[fa + 2u * (kk & 0xffffu)], F[fb + 2u * (kk >> 16)] * sgn, acc);
        }
        LDS_PROBE(2);
        if (ROWS_ON_LANES) {
            // D^T[2 n + part][row]: the lane holds row tr * 32 + l31; registers (t, t + 1), t even, are (Re, Im) of
            // column tn * 16 + ((t & 3) + 8 (t >> 2) + 4 kp) / 2 -- whole complex numbers, 8-byte stores, consecutive
            // lanes = consecutive rows (conflict-free when the result is stored rows-fastest)
            const bool row_ok = tr * 32 + l31 < R;
            const uint32_t rcl = (rhi[hi].c + rlo[lo].c);
            uint32_t ncs[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int nn = tn * 16 + (((2 * t) & 3) + 8 * ((2 * t) >> 2) + 4 * kp) / 2;
                ncs[t] = nt[nn < N ? nn : N - 1] >> 16;
            }
            c64* __restrict__ Cc = (c64*)C;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int nn = tn * 16 + (((2 * t) & 3) + 8 * ((2 * t) >> 2) + 4 * kp) / 2;
                if (row_ok && nn < N) Cc[rcl + ncs[t]] = c64{acc[2 * t], acc[2 * t + 1]};
            }
        } else {
        // D[row][2 n + part]: register i holds row (i & 3) + 8 (i >> 2) + 4 kp of the tile
        const bool col_ok = tn * 16 + (l31 >> 1) < N;
        const uint32_t nc = nbc >> 16;
        uint32_t rc[16];   // (every row offset first, then the stores: the look-ups do not wait for each other)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = tr * 32 + (i & 3) + 8 * (i >> 2) + 4 * kp;
            uint32_t h2, l2;
            split_row32(st, (uint32_t)(row < R ? row : R - 1), h2, l2);
            rc[i] = rhi[h2].c + rlo[l2].c;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = tr * 32 + (i & 3) + 8 * (i >> 2) + 4 * kp;
            if (row < R && col_ok) C[(rc[i] + nc) * 2u + (uint32_t)part] = acc[i];
        }
        }
        LDS_PROBE(3);
    }
}

template <typename T>
__device__ __forceinline__ void run_mfma_if(const LdsStepDev& st, const char* blob, T* data, int64_t z, int wave, int n_waves) {}
template <>
__device__ __forceinline__ void run_mfma_if<c64>(const LdsStepDev& st, const char* blob, c64* data, int64_t z, int wave, int n_waves) {
    if (st.mfma == 2) run_pair_mfma<true>(st, blob, data, z, wave, n_waves);
    else run_pair_mfma<false>(st, blob, data, z, wave, n_waves);
}

template <typename T>
__device__ __forceinline__ void run_step(const LdsStepDev& st, const char* blob, T* data, int64_t z) {
    if (st.mfma) {
        if (st.wave >= 0) {
            if ((int)(threadIdx.x >> 6) == st.wave) run_mfma_if<T>(st, blob, data, z, 0, 1);
        } else {
            run_mfma_if<T>(st, blob, data, z, (int)(threadIdx.x >> 6), LDS_RUN_THREADS / 64);
        }
        return;
    }
    int lane, n_lanes;
    if (st.wave >= 0) {
        if ((int)(threadIdx.x >> 6) != st.wave) return;
        lane = threadIdx.x & 63;
        n_lanes = 64;
    } else {
        lane = threadIdx.x;
        n_lanes = LDS_RUN_THREADS;
    }
    if (st.kind == 0) {
        run_load<T>(st, blob, data, z, lane, n_lanes);
    } else if (st.N >= 3) {
        run_pair<T, 4>(st, blob, data, z, lane, n_lanes);
    } else if (st.N == 2) {
        run_pair<T, 2>(st, blob, data, z, lane, n_lanes);
    } else {
        run_pair<T, 1>(st, blob, data, z, lane, n_lanes);
    }
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(LDS_RUN_THREADS) void lds_run_kernel(const LdsCompDev* __restrict__ comps, int z0, int max_phase, unsigned long long* dbg) {
    extern __shared__ uint4 lds_run_smem[];
    const LdsCompDev comp = comps[blockIdx.x];
    const int64_t z = (int64_t)z0 + blockIdx.y;
    // the component's records and tables: one coalesced pass
    {
        const uint4* __restrict__ src = (const uint4*)comp.blob;
        const int n16 = (int)(comp.blob_bytes >> 4);
        for (int i = threadIdx.x; i < n16; i += LDS_RUN_THREADS) lds_run_smem[i] = src[i];
    }
    // (development, CTG_LDS_TIMELINE=file: wave 0 of every component of slice 0 stamps the shader clock after the
    // blob copy and after each of its steps -- 64 stamps per component)
    unsigned long long* stamp = (dbg && blockIdx.y == 0 && threadIdx.x == 0) ? dbg + (size_t)blockIdx.x * 64 : nullptr;
    const unsigned long long cyc0 = clock64();
    if (stamp) stamp[0] = wall_clock64();
    __syncthreads();
    if (stamp) stamp[1] = wall_clock64();
    const char* blob = (const char*)lds_run_smem;
    T* data = (T*)((char*)lds_run_smem + comp.data_off);
    const LdsStepDev* steps = (const LdsStepDev*)blob;
    int phase = 0;
    for (uint32_t s = 0; s < comp.n_steps; ++s) {
        // (a private copy: the record lives in the same LDS the steps store into, a reference would be
        // re-read after every store)
        const int4 head = *(const int4*)&steps[s];   // kind, phase, wave, mfma
        if (head.y != phase) {
            if (head.y > max_phase) break;   // (timing experiments: CTG_LDS_MAX_PHASE)
            __syncthreads();
            phase = head.y;
        }
        if (head.z >= 0 && head.z != (int)(threadIdx.x >> 6)) continue;   // another wave's step
        const LdsStepDev st = steps[s];
        LDS_PROBE(8);
        run_step<T>(st, blob, data, z);
        LDS_PROBE(9);
        if (stamp && s + 2 < 62) stamp[s + 2] = wall_clock64();
    }
    if (stamp) {
        __syncthreads();
        stamp[63] = wall_clock64();
        stamp[62] = 1000000ull + (clock64() - cyc0);   // (shader cycles of the whole component, + 1e6 to tell it apart)
    }
}

hipError_t launch_lds_run(int dtype, const LdsCompDev* d_comps, int n_comps, int nz, int z0, int lds_bytes, hipStream_t stream) {
    static unsigned long long ready[4] = {0, 0, 0, 0};
    const void* kern = nullptr;
    switch (dtype) {
        case 0: kern = (const void*)lds_run_kernel<float>; break;
        case 1: kern = (const void*)lds_run_kernel<double>; break;
        case 2: kern = (const void*)lds_run_kernel<c64>; break;
        case 3: kern = (const void*)lds_run_kernel<c128>; break;
        default: return hipErrorInvalidValue;
    }
    if (lds_bytes > 64 * 1024) {
        const hipError_t e = lds_opt_in(kern, LDS_RUN_MAX_BYTES, &ready[dtype]);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)n_comps, (unsigned)nz);
    static const int max_phase = getenv("CTG_LDS_MAX_PHASE") ? atoi(getenv("CTG_LDS_MAX_PHASE")) : (1 << 30);
    unsigned long long* dbg = nullptr;
    static const char* tl = getenv("CTG_LDS_TIMELINE");
    if (tl) (void)hipMalloc((void**)&dbg, (size_t)n_comps * 64 * 8), (void)hipMemset(dbg, 0, (size_t)n_comps * 64 * 8);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(lds_run_kernel<float>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0, max_phase, dbg); break;
        case 1: hipLaunchKernelGGL(lds_run_kernel<double>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0, max_phase, dbg); break;
        case 2: hipLaunchKernelGGL(lds_run_kernel<c64>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0, max_phase, dbg); break;
        case 3: hipLaunchKernelGGL(lds_run_kernel<c128>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0, max_phase, dbg); break;
    }
    static const bool twice = getenv("CTG_LDS_TWICE") != nullptr;   // (experiment: the same launch again, code now cached)
    if (twice && dbg) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)n_comps * 64);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int c = 0; c < n_comps; ++c) { lo = std::min(lo, h[(size_t)c * 64]); hi = std::max(hi, h[(size_t)c * 64 + 63]); }
        if (FILE* f = fopen(tl, "a")) { fprintf(f, "first launch: %llu ticks\n", hi - lo); fclose(f); }
        (void)hipMemset(dbg, 0, (size_t)n_comps * 64 * 8);
        hipLaunchKernelGGL(lds_run_kernel<c64>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0, max_phase, dbg);
    }
#ifdef CTG_LDS_PROBE
    if (dbg) {
        (void)hipStreamSynchronize(stream);
        unsigned long long pr[512];
        int n = 0;
        (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_lds_probe), sizeof(pr));
        (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_lds_probe_n), sizeof(n));
        if (FILE* f = fopen(tl, "a")) {
            fprintf(f, "probe (tag:cycles since the first):");
            for (int i = 0; i < n; ++i) fprintf(f, " %llu:%llu", pr[i] >> 48, (pr[i] & 0xffffffffffffull) - (pr[0] & 0xffffffffffffull));
            fprintf(f, "\n");
            fclose(f);
        }
        n = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lds_probe_n), &n, sizeof(n));
    }
#endif
    if (dbg) {
        // (the stamps of this launch appended to the file: 100 MHz wall clock ticks relative to the first)
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)n_comps * 64);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        if (FILE* f = fopen(tl, "a")) {
            unsigned long long t0 = ~0ull;
            for (int c = 0; c < n_comps; ++c) if (h[(size_t)c * 64]) t0 = std::min(t0, h[(size_t)c * 64]);
            fprintf(f, "launch of %d components x %d slices\n", n_comps, nz);
            for (int c = 0; c < n_comps; ++c) {
                fprintf(f, " comp %2d:", c);
                for (int i = 0; i < 64; ++i) if (h[(size_t)c * 64 + i]) fprintf(f, " %llu", h[(size_t)c * 64 + i] - t0);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return hipGetLastError();
}

}  // namespace ctg
