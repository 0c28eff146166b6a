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
#include "ctg_lds.h"

namespace ctg {

namespace {

struct RowEnt { uint32_t ab, c; };

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
        T acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = zero_of(T{});
        for (int k = 0; k < K; ++k) {
            const uint32_t kk = kt[k];
            const T av = a[kk & 0xffffu];
            const T* bk = b + (kk >> 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fma_acc(acc[j], av, bk[nbc[j] & 0xffffu]);
        }
        T* c = C + (h.c + l.c);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (g * TN + j < N) c[nbc[j] >> 16] = acc[j];
    }
}

template <typename T>
__device__ __forceinline__ void run_step(const LdsStepDev& st, const char* blob, T* data, int64_t z) {
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
__global__ __launch_bounds__(LDS_RUN_THREADS) void lds_run_kernel(const LdsCompDev* __restrict__ comps, int z0) {
    extern __shared__ uint4 lds_run_smem[];
    const LdsCompDev comp = comps[blockIdx.x];
    const int64_t z = (int64_t)z0 + blockIdx.y;
    // the component's records and tables: one coalesced pass
    {
        const uint4* __restrict__ src = (const uint4*)comp.blob;
        const int n16 = (int)(comp.blob_bytes >> 4);
        for (int i = threadIdx.x; i < n16; i += LDS_RUN_THREADS) lds_run_smem[i] = src[i];
    }
    __syncthreads();
    const char* blob = (const char*)lds_run_smem;
    T* data = (T*)((char*)lds_run_smem + comp.data_off);
    const LdsStepDev* steps = (const LdsStepDev*)blob;
    int phase = 0;
    for (uint32_t s = 0; s < comp.n_steps; ++s) {
        const LdsStepDev& st = steps[s];
        if (st.phase != phase) {
            __syncthreads();
            phase = st.phase;
        }
        run_step<T>(st, blob, data, z);
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
    switch (dtype) {
        case 0: hipLaunchKernelGGL(lds_run_kernel<float>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0); break;
        case 1: hipLaunchKernelGGL(lds_run_kernel<double>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0); break;
        case 2: hipLaunchKernelGGL(lds_run_kernel<c64>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0); break;
        case 3: hipLaunchKernelGGL(lds_run_kernel<c128>, grid, dim3(LDS_RUN_THREADS), lds_bytes, stream, d_comps, z0); break;
    }
    return hipGetLastError();
}

}  // namespace ctg
