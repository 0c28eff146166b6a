// ctg_probe.hip -- memory-path probe (experiment tool, not part of libctg_hip.so):
// a persistent copy kernel shaped like the streaming kernel's traffic (each wave
// owns 4 KB tasks, two in flight), with the store width / pattern as a parameter.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libctg_probe.so ctg_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode 0: 16-B loads, 16-B stores   1: 16-B loads, 8-B stores (same bytes, twice the store instructions)
// mode 2: 16-B loads, 8-B stores in the MFMA epilogue pattern (4 rows x 128 B per instruction)
// mode 3: 16-B loads, no stores     4: no loads, 16-B stores     5: no loads, 8-B stores
// tpw == 0: persistent, wave w takes tasks w, w + n_waves, ...; tpw > 0: short-lived
// waves, wave w takes the tpw consecutive tasks [w * tpw, (w + 1) * tpw)
template <int MODE>
__global__ __launch_bounds__(256, 8) void probe_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                       int64_t n_tasks, int tpw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tpw > 0) {
        // (no prefetch ring: all loads of a task, then its stores, like an elementwise kernel)
        for (int i = 0; i < tpw; ++i) {
            const int64_t t = wave * tpw + i;
            if (t >= n_tasks) return;
            f32x4 x[4];
            const f32x4* p = src + t * 256;
            f32x4* q = dst + t * 256;
            if (MODE < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = p[j * 64 + lane];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = f32x4{(float)t, 1.f, 2.f, (float)j};
            }
            if (MODE == 3) {
                if (x[0][0] == 12345.678f && x[1][1] == x[2][2] && x[3][3] == 1.f) dst[t] = x[0];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j * 64 + lane] = x[j];
            }
        }
        return;
    }
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    f32x4 r[2][4];
    auto issue = [&](f32x4 (&x)[4], int64_t t) {
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = f32x4{(float)t, 1.f, 2.f, (float)j};
            return;
        }
        const f32x4* p = src + t * 256;   // 4 KB = 256 x 16 B
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = p[j * 64 + lane];
    };
    auto retire = [&](f32x4 (&x)[4], int64_t t) {
        if (MODE == 3) {
            if (x[0][0] == 12345.678f && x[1][1] == x[2][2] && x[3][3] == 1.f) dst[t] = x[0];
            return;
        }
        f32x4* q = dst + t * 256;
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j * 64 + lane] = x[j];
        } else if (MODE == 1 || MODE == 5) {
            f32x2* q2 = (f32x2*)q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q2[(2 * j) * 64 + lane] = f32x2{x[j][0], x[j][1]};
                q2[(2 * j + 1) * 64 + lane] = f32x2{x[j][2], x[j][3]};
            }
        } else {
            // 8 store instructions of 4 rows x 128 B: rows 256 B apart (N = 32 complex columns)
            f32x2* q2 = (f32x2*)q;
            const int col = lane & 15, rsel = lane >> 4;   // 16 lanes x 8 B = 128 B per row piece
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (j >> 1) * 4 + rsel, half = j & 1;   // 16 rows x 2 halves = 4 KB
                q2[row * 32 + half * 16 + col] = f32x2{x[j >> 1][2 * (j & 1)], x[j >> 1][2 * (j & 1) + 1]};
            }
        }
    };
    const int64_t mine = wave < n_tasks ? (n_tasks - wave + n_waves - 1) / n_waves : 0;
    if (mine == 0) return;
    issue(r[0], wave);
    if (mine > 1) issue(r[1], wave + n_waves);
    // (peeled so that the compiler's waits at the loop header count the stores)
    int64_t i = 0;
    for (; i + 2 < mine; i += 2) {
        retire(r[0], wave + i * n_waves);
        issue(r[0], wave + (i + 2) * n_waves);
        retire(r[1], wave + (i + 1) * n_waves);
        if (i + 3 < mine) issue(r[1], wave + (i + 3) * n_waves);
    }
    for (; i < mine; ++i) retire(r[i & 1], wave + i * n_waves);
}

extern "C" int ctg_probe_copy(const void* src, void* dst, int64_t nbytes, int mode, int blocks, void* stream, int tpw) {
    const int64_t n_tasks = nbytes / 4096;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)blocks), b(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL(probe_kernel<0>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        case 1: hipLaunchKernelGGL(probe_kernel<1>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        case 2: hipLaunchKernelGGL(probe_kernel<2>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        case 3: hipLaunchKernelGGL(probe_kernel<3>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        case 4: hipLaunchKernelGGL(probe_kernel<4>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        case 5: hipLaunchKernelGGL(probe_kernel<5>, g, b, 0, s, (const f32x4*)src, (f32x4*)dst, n_tasks, tpw); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- matrix-core issue rate: independent MFMA chains, no memory traffic ---- //
typedef double f64x4p __attribute__((ext_vector_type(4)));
typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef float f32x16p __attribute__((ext_vector_type(16)));

template <int WHICH, int CHAINS>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    if (WHICH == 0) {          // v_mfma_f64_16x16x4_f64
        f64x4p acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f64x4p{0, 0, 0, 0};
        const double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        double s = 0;
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
        if (s == 12345.678) out[threadIdx.x] = (float)s;
    } else if (WHICH == 1) {   // v_mfma_f32_16x16x4_f32
        f32x4p acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4p{0, 0, 0, 0};
        const float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        float s = 0;
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {                   // v_mfma_f32_32x32x2_f32
        f32x16p acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
        const float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        float s = 0;
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

// returns flops issued by the launch (per MFMA: 2048 / 2048 / 4096)
extern "C" double ctg_probe_mfma(int which, int chains, int blocks, int iters, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)blocks), b(256);
#define LAUNCH(W, C) hipLaunchKernelGGL((mfma_rate_kernel<W, C>), g, b, 0, s, out, iters)
    if (which == 0 && chains == 4) LAUNCH(0, 4);
    else if (which == 0 && chains == 8) LAUNCH(0, 8);
    else if (which == 1 && chains == 4) LAUNCH(1, 4);
    else if (which == 1 && chains == 8) LAUNCH(1, 8);
    else if (which == 2 && chains == 4) LAUNCH(2, 4);
    else return -1.0;
#undef LAUNCH
    const double per = which == 2 ? 4096.0 : 2048.0;
    return per * chains * (double)iters * 4.0 * blocks;
}


// ---- fp32 products on the bf16 matrix cores --------------------- //
// WHICH 0: chains of independent v_mfma_f32_32x32x16_bf16 (the instruction's issue rate).
// WHICH 1: what one task of the stem kernel's first step would do: 16 fp32 values per lane (8
// complex elements) split exactly into 3 bf16 values each (two ANDs, two SUBs, packing), then the
// 6 significant cross terms x (X: re Re b, -im Im b; Y: re Im b, im Re b) = 24 MFMAs with B
// fragments in registers.  Counts 65536 real MACs x 2 per task: the fp32-equivalent rate.
typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8p (&o)[3]) {
    unsigned w[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]);
        const unsigned h1 = u & 0xffff0000u;
        const float r1 = x[i] - __builtin_bit_cast(float, h1);
        const unsigned h2 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const float r2 = r1 - __builtin_bit_cast(float, h2);
        w[0][i] = h1; w[1][i] = h2; w[2][i] = __builtin_bit_cast(unsigned, r2);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        u32x4p pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = (w[s][2 * i] >> 16) | (w[s][2 * i + 1] & 0xffff0000u);
        o[s] = __builtin_bit_cast(bf16x8p, pk);
    }
}

template <int WHICH>
__global__ __launch_bounds__(512, 1) void bf16x3_kernel(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16p ax, ay;
    for (int t = 0; t < 16; ++t) ax[t] = ay[t] = 0.f;
    bf16x8p bre[3], bim[3];
    {
        float t0[8];
        for (int i = 0; i < 8; ++i) t0[i] = 0.5f - lane * 1e-3f + i;
        split3(t0, bre);
        for (int i = 0; i < 8; ++i) t0[i] = 0.25f + lane * 1e-3f - i;
        split3(t0, bim);
    }
    if (WHICH == 0) {
        f32x16p acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = ax;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bre[c % 3], bim[c % 3], acc[c], 0, 0, 0);
        float s = 0;
        for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][15];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {
        float re[8], im[8];
        for (int i = 0; i < 8; ++i) {
            re[i] = in[(threadIdx.x * 16 + i) & 1023];
            im[i] = in[(threadIdx.x * 16 + 8 + i) & 1023];
        }
        for (int it = 0; it < iters; ++it) {
            bf16x8p r3[3], i3[3], n3[3];
            split3(re, r3);
            split3(im, i3);
#pragma unroll
            for (int s = 0; s < 3; ++s)
                n3[s] = __builtin_bit_cast(bf16x8p, __builtin_bit_cast(u32x4p, i3[s]) ^ 0x80008000u);
            constexpr int TA[6] = {0, 0, 1, 1, 0, 2}, TB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r3[TA[t]], bre[TB[t]], ax, 0, 0, 0);
                ay = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r3[TA[t]], bim[TB[t]], ay, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(n3[TA[t]], bim[TB[t]], ax, 0, 0, 0);
                ay = __builtin_amdgcn_mfma_f32_32x32x16_bf16(i3[TA[t]], bre[TB[t]], ay, 0, 0, 0);
            }
            // (the next task's values: something the compiler cannot fold)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                re[i] = re[i] * 1.0000001f + 1e-7f;
                im[i] = im[i] * 0.9999999f - 1e-7f;
            }
        }
        float s = ax[0] + ay[15];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

// returns real flops of the launch: WHICH 0 -> 32768 per MFMA; WHICH 1 -> 2 x 65536 fp32-equivalent per task
extern "C" double ctg_probe_bf16x3(int which, int blocks, int iters, float* out, const float* in, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL((bf16x3_kernel<0>), dim3(blocks), dim3(512), 0, s, out, in, iters);
    else hipLaunchKernelGGL((bf16x3_kernel<1>), dim3(blocks), dim3(512), 0, s, out, in, iters);
    return 2 * 65536.0 * (double)iters * 8.0 * blocks;   // (4 MFMAs of 32768 flops | one task)
}

// ---- vector ALU rate: independent FMA chains per lane ---------------------- //
template <typename T, int CHAINS>
__global__ __launch_bounds__(256) void valu_rate_kernel(T* out, int iters) {
    T acc[CHAINS];
    const T a = (T)1.000001 + (T)threadIdx.x * (T)1e-9, b = (T)0.999999;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (T)c;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    T s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    if (s == (T)12345.678) out[threadIdx.x] = s;
}

// which: 0 = f64 FMA, 1 = f32 FMA; returns the flops issued (2 per FMA per lane)
extern "C" double ctg_probe_valu(int which, int blocks, int iters, void* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL((valu_rate_kernel<double, 16>), dim3(blocks), dim3(256), 0, s, (double*)out, iters);
    else hipLaunchKernelGGL((valu_rate_kernel<float, 16>), dim3(blocks), dim3(256), 0, s, (float*)out, iters);
    return 2.0 * 16 * (double)iters * 256.0 * blocks;
}
