// What does landing the x3 loop's operands in LDS cost, by itself and under the MFMAs?  (standalone; gfx950)
// A workgroup of 256 threads walks `trips` K stages of a 128 x 128 x3 tile: per stage 8 KB of fp32 activation rows (128 rows x
// 64 B, row pitch 512 B: a 128-channel channels-last tensor, one tile per workgroup, the 9 taps x 8 channel blocks of a 3x3 layer)
// and 12 KB of bf16 weight planes (shared by every workgroup).
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), ring of S stages, counted vmcnt + raw barrier per stage, NO math
//   mode 1: register staging (5 x buffer_load_dwordx4 -> 5 x ds_write_b128 per thread and stage), one stage ahead, two barriers, NO math
//   mode 2: mode 0 + the stage's 24 MFMAs per wave on bf16 fragments read from the landed stage (no fp32 split)
//   mode 3: mode 1 + the same MFMAs
// Prints microseconds per stage and workgroup-slot, GB/s per CU, and for modes 2 / 3 the bf16 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_void_p;
constexpr int STG = 20480;

template <int MODE_, int S>
__global__ __launch_bounds__(256) void probe(const float* A, unsigned a_bytes, const float* B, unsigned b_bytes, float* out, int trips, int tiles) {
    constexpr int MODE = MODE_ & 3;
    constexpr bool BMAJOR = MODE_ >= 4;      // weight planes stored stage-major: [tap][channel block][plane][row][16] -- whole lines
    __shared__ __attribute__((aligned(1024))) unsigned char ring[S * STG];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, h = lane >> 5;
    const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, b_bytes, 0x00020000);
    const int tile = blockIdx.x % tiles;
    int a_vo[2], b_vo[3];
    for (int i = 0; i < 2; ++i) { const int row = 32 * wave + 16 * i + (lane >> 2); a_vo[i] = ((tile * 128 + row) * 128 + 4 * (lane & 3)) * 4; }
    for (int i = 0; i < 3; ++i) {
        const int q = 3 * wave + i;
        b_vo[i] = BMAJOR ? (q >> 2) * 4096 + (q & 3) * 1024 + lane * 16
                         : (q >> 2) * (128 * 9 * 128 * 2) + ((32 * (q & 3) + (lane >> 1)) * 9 * 128 + 8 * (lane & 1)) * 2;
    }
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int wi = wave >> 1, wj = wave & 1;
    int af[2], bf[2];
    for (int t = 0; t < 2; ++t) { af[t] = (wi * 64 + t * 32 + l31) * 32 + h * 16; bf[t] = 8192 + (wj * 64 + t * 32 + l31) * 32 + h * 16; }
    auto stage_off = [&](int s, int& ao, int& bo) {       // tap-fastest walk: 9 taps (pixel shifts) inside each of 8 channel blocks
        const int t = s % 9, cb = (s / 9) % 8;
        ao = ((t / 3) * 15 + (t % 3)) * 512 + cb * 64;
        bo = BMAJOR ? (t * 8 + cb) * 12288 : (t * 128 + cb * 16) * 2;
    };
    auto mma = [&](const unsigned char* st) {
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[t][p] = *reinterpret_cast<const bf16x8*>(st + af[t] + (p & 1) * 4096);
                fb[t][p] = *reinterpret_cast<const bf16x8*>(st + bf[t] + p * 4096);
            }
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][TA[t]], fb[b][TB[t]], acc[a][b], 0, 0, 0);
    };
    if (MODE == 0 || MODE == 2) {
        auto issue = [&](int s) {
            unsigned char* st = ring + (s % S) * STG;
            int ao, bo; stage_off(s, ao, bo);
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_p)(st + (32 * wave + 16 * i) * 64), 16, a_vo[i] + ao, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) { const int q = 3 * wave + i; __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_p)(st + 8192 + (q >> 2) * 4096 + (q & 3) * 1024), 16, b_vo[i], bo, 0, 0); }
        };
        int issued = 0;
        for (; issued < S - 1 && issued < trips; ++issued) issue(issued);
        for (int it = 0; it < trips; ++it) {
            if (issued - it - 1 >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (S - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (issued < trips) { issue(issued); ++issued; }
            if (MODE == 2) mma(ring + (it % S) * STG);
        }
    } else {
        u32x4 ra[2], rb[3];
        auto load = [&](int s) {
            int ao, bo; stage_off(s, ao, bo);
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_vo[i] + ao, 0, 0));
#pragma unroll
            for (int i = 0; i < 3; ++i) rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, b_vo[i], bo, 0));
        };
        auto store = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(ring + (32 * wave + 16 * i) * 64 + lane * 16) = ra[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) { const int q = 3 * wave + i; *reinterpret_cast<u32x4*>(ring + 8192 + (q >> 2) * 4096 + (q & 3) * 1024 + lane * 16) = rb[i]; }
        };
        load(0); store(); __syncthreads();
        for (int it = 0; it < trips; ++it) {
            if (it + 1 < trips) load(it + 1);
            if (MODE == 3) mma(ring);
            __syncthreads();
            if (it + 1 < trips) { store(); __syncthreads(); }
        }
    }
    float s = reinterpret_cast<float*>(ring)[threadIdx.x];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int S>
static void run(const char* what, int wgs_per_cu, const float* A, unsigned ab, const float* B, unsigned bb, float* out) {
    const int trips = 720, tiles = 720, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, S>), dim3(grid), dim3(256), 0, 0, A, ab, B, bb, out, 72, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, S>), dim3(grid), dim3(256), 0, 0, A, ab, B, bb, out, trips, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_stage = ms * 1e3 / trips;
    const double gbs_cu = wgs_per_cu * 20480.0 / (us_stage * 1e-6) / 1e9;
    const double tf = (MODE >= 2) ? (double)grid * 4 * trips * 24 * 2.0 * 32 * 32 * 16 / ms / 1e9 : 0.0;
    printf("%-44s stages %d, %d workgroup(s)/CU: %6.3f us per stage | %6.1f GB/s per CU landed | %7.1f TFLOP/s bf16 (%.2f of 2500)\n", what, S, wgs_per_cu,
           us_stage, gbs_cu, tf, tf / 2500.0);
}
int main() {
    const unsigned ab = 720u * 128 * 128 * 4 + (2 * 15 + 2 + 1) * 512 + 4096, bb = 3u * 128 * 9 * 128 * 2;
    float *A, *B, *out;
    hipMalloc(&A, ab); hipMalloc(&B, bb); hipMalloc(&out, 256 * 3 * 256 * 4);
    hipMemset(A, 0, ab); hipMemset(B, 0, bb);
    run<4, 3>("LDS-DMA only, stage-major weights", 2, A, ab, B, bb, out);
    run<4, 6>("LDS-DMA only, stage-major weights", 1, A, ab, B, bb, out);
    run<5, 1>("register staging only, stage-major weights", 2, A, ab, B, bb, out);
    run<6, 3>("LDS-DMA + MFMAs, stage-major weights", 2, A, ab, B, bb, out);
    run<6, 6>("LDS-DMA + MFMAs, stage-major weights", 1, A, ab, B, bb, out);
    run<7, 1>("register staging + MFMAs, stage-major weights", 2, A, ab, B, bb, out);
    run<0, 3>("LDS-DMA only", 2, A, ab, B, bb, out);
    run<0, 6>("LDS-DMA only", 1, A, ab, B, bb, out);
    run<0, 2>("LDS-DMA only", 3, A, ab, B, bb, out);
    run<1, 1>("register staging only", 1, A, ab, B, bb, out);
    run<1, 1>("register staging only", 2, A, ab, B, bb, out);
    run<1, 1>("register staging only", 3, A, ab, B, bb, out);
    run<2, 3>("LDS-DMA + 24 MFMAs per wave and stage", 2, A, ab, B, bb, out);
    run<2, 6>("LDS-DMA + 24 MFMAs per wave and stage", 1, A, ab, B, bb, out);
    run<2, 2>("LDS-DMA + 24 MFMAs per wave and stage", 3, A, ab, B, bb, out);
    run<3, 1>("register staging + 24 MFMAs", 1, A, ab, B, bb, out);
    run<3, 1>("register staging + 24 MFMAs", 2, A, ab, B, bb, out);
    run<3, 1>("register staging + 24 MFMAs", 3, A, ab, B, bb, out);
    return 0;
}
