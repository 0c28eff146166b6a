// ctg_probe_loop.hip -- issue-rate probe of the fused stem kernel's inner loops (experiment
// tool, not part of libctg_hip.so): what does the matrix pipe sustain under exactly the
// instruction mixes of ctg_stem.hip's step 2, with nothing else in the kernel?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libctg_probe_loop.so ctg_probe_loop.hip
// One "item" = a 32 x 32 complex tile over K2 = 64: 16 quads of 8 MFMAs (cx, cy alternating).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float flipf(float v, unsigned m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ m);
}

// ALDS: A' fragments from LDS (ds_read_b128 per quad, double-buffered) else registers
// BLDS: B' fragments from LDS (two ds_read_b128 per quad) else registers
// XOR: one sign flip per cx MFMA
// CH: independent accumulator pairs interleaved (1: cx, cy; 2: two items at once)
// DEP: 0 = (cx, cy) alternate; 1 = one accumulator only, back-to-back dependent
// READOUT: copy the 32 accumulators out after every item (what the stores need)
template <bool ALDS, bool BLDS, bool XOR, int CH, int DEP, bool READOUT>
__global__ __launch_bounds__(512, 1) void loop_kernel(float* out, int items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int K2 = 64, LD = K2 + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 5, l31 = lane & 31;
    float* mid = lds;                       // [2][128][LD]
    float* P2 = lds + 2 * 128 * LD;         // [2][64][LD]
    for (int i = tid; i < 2 * 128 * LD + 2 * 64 * LD; i += 512) lds[i] = 1.0f + 1e-3f * (i & 1023);
    __syncthreads();
    const unsigned sgn2 = kk ? 0x80000000u : 0u;
    const float* a_base = mid + kk * 128 * LD + ((wave & 3) * 32 + l31) * LD;
    const float* bxp = P2 + ((kk ? 1 : 0) * 64 + l31) * LD;
    const float* byp = P2 + ((kk ? 0 : 1) * 64 + l31) * LD;
    f32x4 areg[16], bxr[ALDS && !BLDS ? 16 : 1], byr[ALDS && !BLDS ? 16 : 1];
    if (!ALDS) {
#pragma unroll
        for (int q = 0; q < 16; ++q) areg[q] = *(const f32x4*)(a_base + 4 * q);
    }
    if (ALDS && !BLDS) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            bxr[q] = *(const f32x4*)(bxp + 4 * q);
            byr[q] = *(const f32x4*)(byp + 4 * q);
        }
    }
    const f32x4 b0 = *(const f32x4*)(bxp), b1 = *(const f32x4*)(byp);
    float sink = 0.f;
    for (int it = 0; it < items; ++it) {
        f32x16 cx[CH], cy[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int t = 0; t < 16; ++t) cx[c][t] = cy[c][t] = 0.f;
        f32x4 af[2], bx[2], by[2];
        if (ALDS) af[0] = *(const f32x4*)(a_base);
        if (BLDS) {
            bx[0] = *(const f32x4*)(bxp);
            by[0] = *(const f32x4*)(byp);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int nx = (q + 1 < 16 ? q + 1 : 15) * 4;
            if (ALDS) af[(q + 1) & 1] = *(const f32x4*)(a_base + nx);
            if (BLDS) {
                bx[(q + 1) & 1] = *(const f32x4*)(bxp + nx);
                by[(q + 1) & 1] = *(const f32x4*)(byp + nx);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 a = ALDS ? af[q & 1] : areg[q];
            const f32x4 x = BLDS ? bx[q & 1] : (ALDS ? bxr[q] : b0);
            const f32x4 y = BLDS ? by[q & 1] : (ALDS ? byr[q] : b1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float ax = XOR ? flipf(a[t], sgn2) : a[t];
                    if (DEP == 1) {
                        cx[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, x[t], cx[c], 0, 0, 0);
                        cx[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], y[t], cx[c], 0, 0, 0);
                    } else {
                        cx[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, x[t], cx[c], 0, 0, 0);
                        cy[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], y[t], cy[c], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (READOUT) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    float2 v;
                    v.x = cx[c][t];
                    v.y = cy[c][t];
                    if (v.x == 12345.678f) *(float2*)(out + 2 * (tid + t)) = v;
                }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) sink += cx[c][0] + cy[c][15];
        }
    }
    if (sink == 12345.678f) out[tid] = sink;
}

// returns the real flops of the launch
extern "C" double ctg_probe_loop(int variant, int blocks, int items, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (2 * 128 + 2 * 64) * 68 * 4;   // 102 KB: one workgroup per CU
    int ch = 1;
#define GO(V, A, B, X, C, D, R)                                                                         \
    if (variant == V) {                                                                                 \
        auto k = loop_kernel<A, B, X, C, D, R>;                                                         \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), smem > 160 * 1024 ? 160 * 1024 : smem, s, out, items); \
        ch = C;                                                                                         \
    }
    GO(0, false, false, false, 1, 0, false)   // registers only, 2 chains
    GO(1, false, false, true, 1, 0, false)    // + one xor per MFMA pair
    GO(2, true, false, false, 1, 0, false)    // A' from LDS, B' in registers (the K2Q > 0 path)
    GO(3, true, true, true, 1, 0, false)      // A', B' from LDS, xor (the K2Q = 0 path)
    GO(4, true, true, true, 2, 0, false)      // ... two items interleaved (4 chains)
    GO(5, false, false, false, 1, 1, false)   // one chain, back-to-back dependent
    GO(6, false, false, false, 2, 0, false)   // registers only, 4 chains
    GO(7, true, false, false, 1, 0, true)     // variant 2 + accumulator read-out per item
    GO(8, true, true, false, 1, 0, false)     // A', B' from LDS, no xor
    GO(9, true, false, false, 2, 0, false)    // variant 2, two items interleaved
#undef GO
    if (hipGetLastError() != hipSuccess) return -1.0;
    return 4096.0 * 128.0 * ch * (double)items * 8.0 * blocks;
}

// ---- second probe: step 2 of the stem kernel as it is vs the row-interleaved form ----------
// FORM 0: the kernel's current step 2 (X / Y tiles: cx = Re, cy = Im of 32 rows x 32 columns;
//         one A' read per 4 k serves both; B' = (Re b | Im b), (Im b | Re b): two values per k)
//         BLDS: B' from LDS + sign xor (K2Q = 0) / B' in registers with the sign folded (K2Q > 0)
// FORM 1: row-interleaved: tile row 2 i = Re, 2 i + 1 = Im of complex row i; A' row 2 i =
//         (Re a, -Im a), row 2 i + 1 = (Im a, Re a) from three LDS planes; B' = (Re b | Im b): ONE
//         value per k, always in registers; two accumulators (complex rows 0-15, 16-31); a lane's
//         registers (t, t + 1) are (Re, Im) of one element: 8-byte stores without any copy
// Both store every item's 32 x 32 complex results (8 bytes per lane and store, 16 stores).
template <int FORM, bool BLDS, int K2, int LAY = 0>
__global__ __launch_bounds__(512, 1) void step2_kernel(float* out, int items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = K2 + 4, ROWS = 128, PL = ROWS * LD + 32, NQ = K2 / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = lane >> 5, l31 = lane & 31;
    float* mid = lds;                      // [3][PL]
    float* P2 = lds + 3 * (ROWS * (K2 + 9) + 32);   // [2][64][LD]
    for (int i = tid; i < 3 * (ROWS * (K2 + 9) + 32) + 2 * 64 * LD; i += 512) lds[i] = 1.0f + 1e-3f * (i & 1023);
    __syncthreads();
    float* dst = out + ((size_t)blockIdx.x * 8 + wave) * 2048 + 2 * lane;   // 8 KB per wave
    const int rt = wave & 3;
    if constexpr (FORM == 0) {
        const unsigned sgn2 = kk ? 0x80000000u : 0u;
        const float* a_base = mid + kk * PL + (rt * 32 + l31) * LD;
        const float* bxp = P2 + ((kk ? 1 : 0) * 64 + l31) * LD;
        const float* byp = P2 + ((kk ? 0 : 1) * 64 + l31) * LD;
        f32x4 bxr[BLDS ? 1 : NQ], byr[BLDS ? 1 : NQ];
        if (!BLDS) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                bxr[q] = *(const f32x4*)(bxp + 4 * q);
                byr[q] = *(const f32x4*)(byp + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) bxr[q][t] = flipf(bxr[q][t], sgn2);
            }
        }
        for (int it = 0; it < items; ++it) {
            f32x16 cx, cy;
#pragma unroll
            for (int t = 0; t < 16; ++t) cx[t] = cy[t] = 0.f;
            f32x4 af[2], bx[2], by[2];
            af[0] = *(const f32x4*)(a_base);
            if (BLDS) {
                bx[0] = *(const f32x4*)(bxp);
                by[0] = *(const f32x4*)(byp);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int nx = (q + 1 < NQ ? q + 1 : NQ - 1) * 4;
                af[(q + 1) & 1] = *(const f32x4*)(a_base + nx);
                if (BLDS) {
                    bx[(q + 1) & 1] = *(const f32x4*)(bxp + nx);
                    by[(q + 1) & 1] = *(const f32x4*)(byp + nx);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (BLDS) {
                        cx = __builtin_amdgcn_mfma_f32_32x32x2f32(flipf(af[q & 1][t], sgn2), bx[q & 1][t], cx, 0, 0, 0);
                        cy = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][t], by[q & 1][t], cy, 0, 0, 0);
                    } else {
                        cx = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][t], bxr[q][t], cx, 0, 0, 0);
                        cy = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][t], byr[q][t], cy, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float2 v;
                v.x = cx[t];
                v.y = cy[t];
                *(float2*)(dst + 128 * t) = v;
            }
        }
    } else {
        // plane of this lane's A' values: (row parity, k-row) -> Re, -Im, Im, Re
        const int par = l31 & 1;
        const int plane = par == kk ? 0 : (par ? 1 : 2);
        // LAY 0: three planes, 32 floats of bank shift between them; 4: no shift; 1-3: the planes
        // interleaved per row, [row][plane][LDI] with row pitch PITCH
        constexpr int LDI = LAY == 2 ? K2 + 8 : K2 + 4;
        constexpr int PITCH = LAY == 3 ? 3 * LDI + 4 : 3 * LDI;
        const float* a_base = (LAY == 0 || LAY == 4)
                                  ? mid + plane * (LAY == 0 ? PL : ROWS * LD) + (rt * 32 + (l31 >> 1)) * LD
                                  : mid + (rt * 32 + (l31 >> 1)) * PITCH + plane * LDI;
        constexpr int A1OFF = (LAY == 0 || LAY == 4) ? 16 * LD : 16 * PITCH;
        const float* bp = P2 + (kk * 64 + l31) * LD;
        f32x4 br[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) br[q] = *(const f32x4*)(bp + 4 * q);
        for (int it = 0; it < items; ++it) {
            f32x16 c0, c1;
#pragma unroll
            for (int t = 0; t < 16; ++t) c0[t] = c1[t] = 0.f;
            f32x4 a0[2], a1[2];
            a0[0] = *(const f32x4*)(a_base);
            a1[0] = *(const f32x4*)(a_base + A1OFF);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int nx = (q + 1 < NQ ? q + 1 : NQ - 1) * 4;
                a0[(q + 1) & 1] = *(const f32x4*)(a_base + nx);
                a1[(q + 1) & 1] = *(const f32x4*)(a_base + A1OFF + nx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q & 1][t], br[q][t], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q & 1][t], br[q][t], c1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                float2 v, w;
                v.x = c0[t];
                v.y = c0[t + 1];
                w.x = c1[t];
                w.y = c1[t + 1];
                *(float2*)(dst + 128 * t) = v;
                *(float2*)(dst + 128 * (t + 1)) = w;
            }
        }
    }
}

extern "C" double ctg_probe_step2(int variant, int blocks, int items, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int k2 = 64;
#define GO2(V, F, B, K, ...)                                                                           \
    if (variant == V) {                                                                                \
        auto k = step2_kernel<F, B, K, ##__VA_ARGS__>;                                                 \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        const size_t smem = (3 * (128 * (K + 9) + 32) + 2 * 64 * (K + 4)) * 4;                         \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), smem, s, out, items);                           \
        k2 = K;                                                                                        \
    }
    GO2(0, 0, true, 64)     // current, B' from LDS + xor (K2 = 64 does not fit the registers today)
    GO2(1, 1, false, 64)    // row-interleaved, B' in registers (64 floats)
    GO2(2, 0, false, 32)    // current, B' in registers (K2 = 32: 64 floats)
    GO2(3, 1, false, 32)    // row-interleaved (32 floats)
    GO2(4, 0, true, 32)     // current, B' from LDS (K2 = 32 with two items per wave today)
    GO2(5, 0, false, 64)    // current with 128 floats of B' in registers (what fits if nothing else is live)
    GO2(6, 1, false, 64, 1)   // row-interleaved, planes interleaved per row, LD = K2 + 4
    GO2(7, 1, false, 64, 2)   // ... LD = K2 + 8
    GO2(8, 1, false, 64, 3)   // ... LD = K2 + 4, row pitch + 4
    GO2(9, 1, false, 64, 4)   // three planes without the bank shift
    GO2(10, 1, false, 32, 1)  // K2 = 32, planes interleaved per row
    GO2(11, 1, false, 32, 3)
#undef GO2
    if (hipGetLastError() != hipSuccess) return -1.0;
    return 4096.0 * 2.0 * k2 * (double)items * 8.0 * blocks;
}
