// Probe of `buffer_load_dwordx4 ... offen lds` (LDS-DMA) on gfx950: where do the lanes' 16 bytes land, and what does a lane whose
// offset is out of the buffer's range write -- zero, or nothing?  (standalone; hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* x, float* y, int n) {
    __shared__ __attribute__((aligned(16))) float buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = 7.f;
    __syncthreads();
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every lane loads the 16 bytes at x + 16 * (63 - lane) (reversed: shows the placement is by LANE, not by address); odd lanes of wave 1 out of range
    const int off = (wave == 1 && (lane & 1)) ? -1 : (63 - lane) * 16 + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(reinterpret_cast<unsigned char*>(buf) + wave * 1024), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 1024; i += 256) y[i] = buf[i];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, n);
    hipMemcpy(h.data(), y, n * 4, hipMemcpyDeviceToHost);
    printf("wave 0, lanes 0..3 (expect x[252..255], x[248..251], ...):");
    for (int i = 0; i < 16; ++i) printf(" %g", h[i]);
    printf("\nwave 1, lanes 0..3 (odd lanes out of range: 0 = zero-filled, 7 = untouched):");
    for (int i = 256; i < 272; ++i) printf(" %g", h[i]);
    printf("\n");
    return 0;
}
