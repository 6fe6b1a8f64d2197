// What can one CU's matrix pipes sustain under the instruction mixes of the x3 loop?  (standalone; gfx950)
//   mix 0: MFMAs only -- 24 x v_mfma_f32_32x32x16_bf16 per trip on four accumulators (the x3 loop's k-step), operands fixed
//   mix 1: + the k-step's 12 conflict-free ds_read_b128 fragment reads (their results feed the MFMAs)
//   mix 2: + 88 VALU instructions per trip (the split of 16 fp32 values into three bf16 planes), fragments from LDS raw fp32
//   mix 3: mix 1 + one s_barrier per trip (4 waves)
// for 1, 2, 3 workgroups of 256 threads per CU.  Prints TFLOP/s of bf16 MFMA work (dense peak 2500) chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

template <int MIX, int LDSKB>
__global__ __launch_bounds__(256) void probe(float* out, int trips) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSKB * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < LDSKB * 256; i += 256) reinterpret_cast<float*>(lds)[i] = 1.0f + 0.001f * (i & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 fa[2][3], fb[2][3];
    for (int t = 0; t < 2; ++t) for (int p = 0; p < 3; ++p) for (int e = 0; e < 8; ++e) { fa[t][p][e] = (__bf16)(0.5f + 0.01f * p); fb[t][p][e] = (__bf16)(0.25f + 0.01f * t); }
    const int wi = wave >> 1, wj = wave & 1;
    // conflict-free fragment addresses as in the x3 K-tile-16 images: bf16 rows of 32 bytes, chunk h ^ ((row >> 3) & 1)
    int af[2], bfr[2], araw0[2], araw1[2];
    for (int t = 0; t < 2; ++t) {
        const int ra = wi * 64 + t * 32 + l31, rb = wj * 64 + t * 32 + l31;
        af[t] = ra * 32 + ((h ^ ((ra >> 3) & 1)) * 16);
        bfr[t] = 8192 + rb * 32 + ((h ^ ((rb >> 3) & 1)) * 16);
        araw0[t] = ra * 64 + (((2 * h) ^ ((ra >> 2) & 3)) * 16);
        araw1[t] = ra * 64 + (((2 * h + 1) ^ ((ra >> 2) & 3)) * 16);
    }
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < trips; ++it) {
        const int tog = (it & 1) * 24576;       // the fragment addresses change from trip to trip: the reads stay in the loop
        if (MIX == 1 || MIX == 3) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[t][p] = *reinterpret_cast<const bf16x8*>(lds + tog + af[t] + p * 4096 - (p == 2 ? 4096 : 0));
                    fb[t][p] = *reinterpret_cast<const bf16x8*>(lds + tog + bfr[t] + p * 4096 - (p == 2 ? 4096 : 0));
                }
        }
        if (MIX == 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(lds + tog + araw0[t]);
                const u32x4 v1 = *reinterpret_cast<const u32x4*>(lds + tog + araw1[t]);
                unsigned hh[4], mm[4], ll[4];
                split2(__uint_as_float(v0[0]), __uint_as_float(v0[1]), hh[0], mm[0], ll[0]);
                split2(__uint_as_float(v0[2]), __uint_as_float(v0[3]), hh[1], mm[1], ll[1]);
                split2(__uint_as_float(v1[0]), __uint_as_float(v1[1]), hh[2], mm[2], ll[2]);
                split2(__uint_as_float(v1[2]), __uint_as_float(v1[3]), hh[3], mm[3], ll[3]);
                u32x4 ph, pm, pl;
                for (int q = 0; q < 4; ++q) { ph[q] = hh[q]; pm[q] = mm[q]; pl[q] = ll[q]; }
                fa[t][0] = __builtin_bit_cast(bf16x8, ph); fa[t][1] = __builtin_bit_cast(bf16x8, pm); fa[t][2] = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[t][p] = *reinterpret_cast<const bf16x8*>(lds + tog + bfr[t] + (p & 1) * 4096);
            }
        }
        if (MIX == 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][TA[t]], fb[b][TB[t]], acc[a][b], 0, 0, 0);
        if (MIX == 0) asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));      // keep the loop a loop
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MIX, int LDSKB>
static void run(const char* what, int wgs_per_cu, float* out) {
    const int trips = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MIX, LDSKB>), dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MIX, LDSKB>), dim3(grid), dim3(256), 0, 0, out, trips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * trips * 24 * 2.0 * 32 * 32 * 16;
    printf("%-58s %d workgroup(s)/CU: %7.1f TFLOP/s bf16 (%.3f of 2500)\n", what, wgs_per_cu, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
}
int main() {
    float* out; hipMalloc(&out, 256 * 3 * 256 * 4);
    for (int w = 1; w <= 3; ++w) run<0, 52>("MFMAs only (52 KB LDS per workgroup)", w, out);
    for (int w = 1; w <= 3; ++w) run<1, 52>("+ 12 conflict-free ds_read_b128 per 24 MFMAs", w, out);
    for (int w = 1; w <= 3; ++w) run<2, 52>("+ 88 split VALU per 24 MFMAs (raw fp32 fragments)", w, out);
    for (int w = 1; w <= 3; ++w) run<3, 52>("reads + one s_barrier per 24 MFMAs", w, out);
    return 0;
}
