// ubench_launch.hip -- what does a dependent step cost on this box?  (round 6, small-tree execution model)
//
//   1. a chain of N tiny kernels replayed from a hipGraph: microseconds per kernel (the floor of the
//      launch-per-wave-front executor),
//   2. the same chain where every kernel walks table -> table -> data (three dependent loads),
//   3. ONE persistent launch whose workgroups meet at N grid barriers (monotonic device counter,
//      release / acquire at agent scope), with and without 4 KB written / read across the barrier,
//   4. ONE workgroup doing N __syncthreads-separated LDS steps (the LDS-resident subtree floor).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_launch tools/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void tiny_kernel(float* x) {
    if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f;
}

__global__ void chain_kernel(const long long* t1, const long long* t2, float* data, float* out) {
    const long long a = t1[threadIdx.x & 15];
    const long long b = t2[a];
    out[blockIdx.x * 256 + threadIdx.x] = data[b + threadIdx.x] * 2.f;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// relaxed spinning, one acquire fence at the end
__device__ __forceinline__ void grid_barrier2(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned* counter, int n_barriers, float* buf, float* out) {
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < n_barriers; ++i) {
        if (MODE >= 1) {
            // write 4 KB, read the 4 KB the next workgroup wrote in the previous phase
            float* mine = buf + ((size_t)(i & 1) * nb + blockIdx.x) * 1024;
            for (int j = threadIdx.x; j < 1024; j += 256) mine[j] = acc + j;
        }
        if (MODE == 2) grid_barrier2(counter, (unsigned)(i + 1) * nb);
        else grid_barrier(counter, (unsigned)(i + 1) * nb);
        if (MODE >= 1) {
            const float* other = buf + ((size_t)(i & 1) * nb + (blockIdx.x + 1) % nb) * 1024;
            for (int j = threadIdx.x; j < 1024; j += 256) acc += other[j];
        }
    }
    if (out) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void lds_steps_kernel(int n_steps, float* out) {
    __shared__ float buf[2][4096];
    for (int j = threadIdx.x; j < 4096; j += 1024) buf[0][j] = j;
    __syncthreads();
    for (int i = 0; i < n_steps; ++i) {
        const float* src = buf[i & 1];
        float* dst = buf[(i + 1) & 1];
        for (int j = threadIdx.x; j < 4096; j += 1024) dst[j] = src[(j * 17 + i) & 4095] * 1.0001f + src[(j + 1) & 4095];
        __syncthreads();
    }
    out[threadIdx.x] = buf[n_steps & 1][threadIdx.x];
}

static float time_graph(hipGraphExec_t g, hipStream_t s, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipGraphLaunch(g, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(g, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float *x, *data, *out, *buf;
    long long *t1, *t2;
    unsigned* counter;
    CK(hipMalloc(&x, 4096));
    CK(hipMalloc(&data, 1 << 20));
    CK(hipMalloc(&out, 1 << 24));
    CK(hipMalloc(&buf, (size_t)2 * 4096 * 4096));
    CK(hipMalloc(&t1, 4096));
    CK(hipMalloc(&t2, 4096));
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(t1, 0, 4096));
    CK(hipMemset(t2, 0, 4096));
    CK(hipMemset(data, 0, 1 << 20));

    for (int N : {1, 8, 32}) {
        for (int variant = 0; variant < 3; ++variant) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, x);
                else if (variant == 1) hipLaunchKernelGGL(chain_kernel, dim3(16), dim3(256), 0, s, t1, t2, data, out);
                else hipLaunchKernelGGL(chain_kernel, dim3(1024), dim3(256), 0, s, t1, t2, data, out);
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const float ms = time_graph(ge, s, 50);
            printf("graph of %2d %-28s: %8.2f us per graph, %6.2f us per kernel\n", N,
                   variant == 0 ? "tiny kernels" : (variant == 1 ? "3-dependent-load x16 wg" : "3-dependent-load x1024 wg"),
                   ms * 1e3, ms * 1e3 / N);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    // plain stream launches (no graph)
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, x);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, x);
        CK(hipEventRecord(b, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("stream of 1000 tiny kernels       : %6.2f us per kernel\n", ms);
    }
    for (int nb : {32, 64, 256, 512, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int N : {1, 65}) {
                hipEvent_t a, b;
                CK(hipEventCreate(&a));
                CK(hipEventCreate(&b));
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipMemsetAsync(counter, 0, 4, s));
                    CK(hipEventRecord(a, s));
                    if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nb), dim3(256), 0, s, counter, N, buf, out);
                    else if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(nb), dim3(256), 0, s, counter, N, buf, out);
                    else hipLaunchKernelGGL(persistent_kernel<2>, dim3(nb), dim3(256), 0, s, counter, N, buf, out);
                    CK(hipEventRecord(b, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("persistent %4d wg mode %d, %2d barriers: %8.2f us\n", nb, mode, N, best * 1e3);
            }
        }
    }
    for (int N : {1, 101}) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL(lds_steps_kernel, dim3(1), dim3(1024), 0, s, N, out);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("one workgroup, %3d LDS steps of 4096 elements: %8.2f us\n", N, best * 1e3);
    }
    return 0;
}
