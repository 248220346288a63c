// Micro-benchmark: sustained issue rate of the VALU instructions the RANSAC
// scoring loop is made of (packed f32 FMA with vector and with scalar operands,
// plain f32 FMA, compare-to-SGPR + conditional add), 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) valu_loop(const float* __restrict__ in, float* out, int iters)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    f2v acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f2v{in[(t + k) & 1023], in[(t + 2 * k + 1) & 1023]};
    // wave-uniform multipliers: in SGPRs for KIND 1
    f2v s0 = f2v{in[blockIdx.x & 1023], in[(blockIdx.x + 7) & 1023]};
    f2v v0 = f2v{in[(t * 3) & 1023], in[(t * 5) & 1023]};
    int cnt = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_elementwise_fma(acc[k], v0, v0);
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_elementwise_fma(acc[k], s0, v0);
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k].x = __builtin_fmaf(acc[k].x, v0.x, v0.y);
                acc[k].y = __builtin_fmaf(acc[k].y, v0.y, v0.x);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k] = __builtin_elementwise_fma(acc[k], v0, v0);
                cnt += (acc[k].x <= 0.f ? 1 : 0) + (acc[k].y <= 0.f ? 1 : 0);
            }
        }
    }
    float sum = (float)cnt;
    for (int k = 0; k < 16; ++k) sum += acc[k].x + acc[k].y;
    out[t] = sum;
}

template <int KIND>
static void run(const char* what, int instr_per_iter, const float* din, float* dout)
{
    const int iters = 4096;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int waves = 256 * 4 * wps;  // one launch fills every SIMD with wps waves
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        valu_loop<KIND><<<waves, 64>>>(din, dout, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        valu_loop<KIND><<<waves, 64>>>(din, dout, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double per_simd = (double)wps * iters * instr_per_iter;  // wave instructions one SIMD issued
        printf("%-34s %d waves/SIMD: %7.1f us, %6.1f wave-instructions/us/SIMD (4-cycle issue at 2.4 GHz = 600)\n",
               what, wps, ms * 1e3, per_simd / (ms * 1e3));
    }
}

int main()
{
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
    float *din, *dout;
    hipMalloc(&din, 4096);
    hipMalloc(&dout, 256 * 4 * 8 * 64 * 4);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("v_pk_fma_f32 (vector operands)", 16, din, dout);
    run<1>("v_pk_fma_f32 (scalar multiplier)", 16, din, dout);
    run<2>("v_fma_f32 x2", 32, din, dout);
    run<3>("v_pk_fma_f32 + 2 compares + count", 16 * 5, din, dout);
    return 0;
}
