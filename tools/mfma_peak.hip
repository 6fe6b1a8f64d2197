// how fast does v_mfma_f32_32x32x2_f32 actually run on this box under sustained load (random operands)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    for (int i = 0; i < iters; ++i) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 4, iters = 20000;
    float *in, *out;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 5.0 * blocks * 4 /*waves*/ * (double)iters * 4 * 4096.0;
        printf("rep %d: %.2f ms  %.1f TFLOP/s\n", rep, ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
