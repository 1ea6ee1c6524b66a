// Is packed FP32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) worth it on gfx950?  Same flops through packed and scalar
// VALU instructions, 1 / 2 / 4 waves per SIMD.  (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(2))) float V2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: v_fma_f32 x2, 1: v_pk_fma_f32, 2: v_add_f32 x2, 3: v_pk_add_f32, 4: v_mul x2, 5: v_pk_mul
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    V2 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = V2{(float)threadIdx.x + i, 1.0f + i};
    V2 A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(A), "v"(B));
                else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(A));
                else if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(A));
                else if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].y) : "v"(a), "v"(b)); }
                else if (MODE == 2) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i].x) : "v"(a)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i].y) : "v"(a)); }
                else { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i].x) : "v"(a)); asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i].y) : "v"(a)); }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* out, int wg_per_cu) {
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 16, 1.0001f, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // element-operations (one float result each) per CU per clock at 2.4 GHz
    const double elems = (double)256 * wg_per_cu * 256 * iters * 4 * 8 * 2;
    printf("%-14s %d waves/SIMD: %7.3f ms  %.1f float results / clk / CU (2.4 GHz)\n", name, wg_per_cu, ms, elems / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    float* out; CK(hipMalloc((void**)&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x2", out, w); run<1>("v_pk_fma_f32", out, w);
        run<2>("v_add_f32 x2", out, w); run<3>("v_pk_add_f32", out, w);
        run<4>("v_mul_f32 x2", out, w); run<5>("v_pk_mul_f32", out, w);
    }
    return 0;
}
