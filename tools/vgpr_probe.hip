// VERDICT r04 1(c), probe: a 256 KiB vector (N = 2^15 complex float / 2^16 real float / 2^14 complex double) as ONE HBM pass with the vector held
// in VGPRs - one 1024-thread workgroup per CU, 16 x 16 bytes per thread (64 data VGPRs: the CU's register file holds the vector), LDS (160 KiB) only
// as a sliced transpose buffer: every exchange between butterfly stages moves the vector through LDS in two halves of 128 KiB.
// This is the MEMORY + LDS SKELETON of such a kernel (no butterflies; FMA = dummy arithmetic per value and stage to stand in for them):
// what could the organisation reach at best, against 0.36 of the roofline for the two tile passes these sizes run today?
//   NX   exchanges through LDS per vector (a 16 x 16 x 16 x 8 decomposition needs 3, 32 x 32 x 32 with 8-byte accesses 2)
//   FMA  dependent FMAs per float and stage (a radix-16 butterfly costs ~10 flops per float)
//   PF   1: the next vector's loads are issued before the stores of the current one (needs the 64 registers twice: 128 VGPRs at 1024 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float V4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int T = 1024, NCH = 16, VB = T * NCH * 16;     // 256 KiB per vector

template <int FMA> __device__ __forceinline__ void work(V4 (&v)[NCH], float c) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int k = 0; k < FMA; ++k) {
            v[i].x = __builtin_fmaf(v[i].x, c, v[(i + 1) % NCH].y); v[i].y = __builtin_fmaf(v[i].y, c, v[(i + 3) % NCH].z);
            v[i].z = __builtin_fmaf(v[i].z, c, v[(i + 5) % NCH].w); v[i].w = __builtin_fmaf(v[i].w, c, v[(i + 7) % NCH].x);
        }
    }
}

template <int NX, int FMA, int PF>
__global__ void __launch_bounds__(T) probe(const V4* __restrict__ in, V4* __restrict__ out, unsigned nvec, unsigned* ctr, float c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    V4* img = reinterpret_cast<V4*>(smem);               // 8192 chunks = 128 KiB
    __shared__ unsigned s_next[2];
    const int t = threadIdx.x;
    unsigned pend = 0;
    unsigned g = blockIdx.x;
    pend = blockIdx.x + gridDim.x;                        // first two vectors static, the counter hands out the rest
    V4 cur[NCH], nxt[PF ? NCH : 1];
    if (PF && g < nvec) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) nxt[i] = __builtin_nontemporal_load(in + (size_t)g * (VB / 16) + t + T * i);
    }
    for (unsigned it = 0; g < nvec; ++it) {
        if (t == 0) { s_next[(it + 1) & 1] = pend; pend = 2u * gridDim.x + atomicAdd(ctr, 1u); }
        const V4* src = in + (size_t)g * (VB / 16);
        V4* dst = out + (size_t)g * (VB / 16);
        if (PF) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) cur[i] = nxt[i];
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) cur[i] = __builtin_nontemporal_load(src + t + T * i);
        }
        __syncthreads();
        const unsigned gn = s_next[(it + 1) & 1];
        if (PF && gn < nvec) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) nxt[i] = __builtin_nontemporal_load(in + (size_t)gn * (VB / 16) + t + T * i);
        }
        work<FMA>(cur, c);
#pragma unroll
        for (int x = 0; x < NX; ++x) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                 // the vector through LDS in two halves of 128 KiB
#pragma unroll
                for (int i = 0; i < 8; ++i) img[i * T + t] = cur[8 * h + i];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[8 * h + i] = img[((t * 8 + i) ^ ((t >> 7) & 7)) & 8191];   // a transposing read
                __syncthreads();
            }
            work<FMA>(cur, c);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) __builtin_nontemporal_store(cur[i], dst + t + T * i);
        g = gn;
    }
}

template <int NX, int FMA, int PF>
void run(const V4* in, V4* out, size_t bytes) {
    static unsigned* ctr = nullptr; if (!ctr) CK(hipMalloc((void**)&ctr, 64));
    const unsigned nvec = (unsigned)(bytes / VB);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = probe<NX, FMA, PF>;
    const size_t lds = 128 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k)));
    for (int r = 0; r < 3; ++r) { CK(hipMemsetAsync(ctr, 0, 64)); k<<<256, T, lds>>>(in, out, nvec, ctr, 0.999f); }
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) { CK(hipMemsetAsync(ctr, 0, 64)); k<<<256, T, lds>>>(in, out, nvec, ctr, 0.999f); }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("256 KiB per 1024-thread workgroup, %d LDS exchanges, %2d FMA per float and stage, prefetch %d: %.3f of 8 TB/s  (%d VGPRs, %zu B scratch)\n",
           NX, FMA, PF, 2.0 * nvec * VB * 10 / (ms * 1e-3) / 8e12, fa.numRegs, (size_t)fa.localSizeBytes);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    V4 *in, *out; CK(hipMalloc((void**)&in, bytes)); CK(hipMalloc((void**)&out, bytes));
    CK(hipMemset(in, 1, bytes));
    run<0, 0, 0>(in, out, bytes);
    run<0, 0, 1>(in, out, bytes);
    run<2, 0, 0>(in, out, bytes);
    run<3, 0, 0>(in, out, bytes);
    run<3, 0, 1>(in, out, bytes);
    run<2, 10, 0>(in, out, bytes);
    run<3, 8, 0>(in, out, bytes);
    run<3, 8, 1>(in, out, bytes);
    run<3, 16, 0>(in, out, bytes);
    return 0;
}
