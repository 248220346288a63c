// Micro-benchmark: how many workgroups of a given shape a compute unit holds at
// once.  Every workgroup waits a fixed time (s_sleep loop on the realtime
// counter), so a launch of k workgroups per CU takes ceil(k / capacity) waits:
// the step in the curve is the capacity.  Shapes: waves per workgroup 1 / 2 / 4,
// static LDS 0 / 5 KB per wave, about 70 or 120 VGPRs.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/wg_residency.hip -o scripts/micro/wg_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WAVES, int LDS_BYTES, int REGS>
__global__ void __launch_bounds__(WAVES * 64) hold_kernel(float* out, unsigned long long ticks)
{
    __shared__ unsigned char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    float r[REGS];
#pragma unroll
    for (int k = 0; k < REGS; ++k) r[k] = (float)(threadIdx.x + k);
    if (LDS_BYTES > 0) lds[threadIdx.x] = (unsigned char)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int k = 0; k < REGS; ++k) r[k] = r[k] * 1.0001f + 1.f;  // keeps the registers live
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < REGS; ++k) s += r[k];
    if (LDS_BYTES > 0) s += lds[(threadIdx.x * 7) % 64];
    if (s == 12345.f) out[0] = s;
}

template <int WAVES, int LDS_BYTES, int REGS>
void sweep(const char* label, float* d_out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned long long ticks = 2000;  // 100 MHz realtime counter: 20 us
    printf("%s:", label);
    for (int per_cu : {4, 8, 12, 16, 20, 24, 28, 32, 40}) {
        const int wgs = 256 * per_cu / WAVES;
        if (wgs < 256) continue;
        hold_kernel<WAVES, LDS_BYTES, REGS><<<wgs, WAVES * 64>>>(d_out, ticks);  // warm
        hipDeviceSynchronize();
        hipEventRecord(a);
        hold_kernel<WAVES, LDS_BYTES, REGS><<<wgs, WAVES * 64>>>(d_out, ticks);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("  %d waves/CU %.0f us", per_cu, ms * 1e3);
    }
    printf("\n");
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, 64);
    sweep<1, 0, 16>("1 wave/WG, no LDS, few regs   ", d_out);
    sweep<1, 5120, 16>("1 wave/WG, 5 KB LDS, few regs ", d_out);
    sweep<1, 5120, 60>("1 wave/WG, 5 KB LDS, ~70 regs ", d_out);
    sweep<2, 10240, 60>("2 waves/WG, 10 KB LDS, ~70 regs", d_out);
    sweep<4, 20480, 60>("4 waves/WG, 20 KB LDS, ~70 regs", d_out);
    sweep<1, 5120, 110>("1 wave/WG, 5 KB LDS, ~120 regs", d_out);
    return 0;
}
