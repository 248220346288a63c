// Micro-benchmark: sustained int8 MFMA rate of the two gfx950 shapes on random
// operands held in registers (no memory traffic), two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape.hip -o gpurun_out/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) mfma_loop(const int* __restrict__ in, int* out, int iters)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    v4i a[4], b[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) {
            a[k][j] = in[(t * 32 + k * 4 + j) & 0xFFFFF];
            b[k][j] = in[(t * 32 + 16 + k * 4 + j) & 0xFFFFF];
        }
    int sum = 0;
    if (SHAPE == 32) {
        v16i acc[4];
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) acc[q][i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[(k + q) & 3], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) sum += acc[q][i];
    } else {
        v4i acc[16];
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) acc[q][i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(k + q) & 3], b[(k + 2 * q) & 3], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) sum += acc[q][i];
    }
    out[t] = sum;
}

int main()
{
    const int n = 1 << 20;
    std::vector<int> h(n);
    srand(1);
    for (auto& v : h) v = rand() ^ (rand() << 16);
    int *din, *dout;
    hipMalloc(&din, n * 4);
    hipMalloc(&dout, 2048 * 256 * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int shape : {32, 16, 32, 16}) {
        // ops per wave per iteration: 16 MFMAs of 32x32x32 (65536 ops) or 32 of 16x16x64 (32768 ops): equal
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) {
                if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(1024), dim3(256), 0, 0, din, dout, iters);
                else hipLaunchKernelGGL(mfma_loop<16>, dim3(1024), dim3(256), 0, 0, din, dout, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops = 20.0 * 1024 * 4 * iters * 16 * 65536.0;
            printf("shape %dx%d: %.2f ms  %.1f TOP/s (%.1f%% of 5000)\n", shape, shape, ms, ops / (ms * 1e-3) / 1e12,
                   ops / (ms * 1e-3) / 5e15 * 100);
        }
    }
    return 0;
}
