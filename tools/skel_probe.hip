// Memory skeleton of the persistent one-vector-per-workgroup transforms (fft_tiled.h, n = 8192 float: 64 KiB per vector, 256
// threads, two workgroups per CU): pure copies, no arithmetic, no LDS.  What does the LOAD -> STORE organisation itself reach?
//   VB      bytes per vector (one workgroup iteration)      WPC   workgroups per CU      T   threads per workgroup
//   MODE 0  load the whole vector, then store it (the kernel's shape)
//   MODE 1  the same with the next vector's loads issued before the stores of the current one (register prefetch)
//   MODE 2  in SUB sequential sub-blocks: load sub-block s + 1, store sub-block s (short pipeline, few registers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float V4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int VB, int T, int MODE, int SUB>
__global__ void __launch_bounds__(T) skel(const V4* __restrict__ in, V4* __restrict__ out, unsigned nvec, unsigned* ctr) {
    constexpr int NCH = VB / 16 / T;          // 16-byte chunks per thread and vector
    __shared__ unsigned s_next[2];
    const int t = threadIdx.x;
    unsigned pend = 0, g;
    if (t == 0) { s_next[0] = atomicAdd(ctr, 1u); pend = atomicAdd(ctr, 1u); }
    __syncthreads();
    g = s_next[0];
    V4 cur[NCH], nxt[MODE == 1 ? NCH : 1];
    if (MODE == 1 && g < nvec) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) nxt[i] = __builtin_nontemporal_load(in + (size_t)g * (VB / 16) + t + T * i);
    }
    for (unsigned it = 0; g < nvec; ++it) {
        if (t == 0) { s_next[(it + 1) & 1] = pend; pend = atomicAdd(ctr, 1u); }
        const V4* src = in + (size_t)g * (VB / 16);
        V4* dst = out + (size_t)g * (VB / 16);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) cur[i] = __builtin_nontemporal_load(src + t + T * i);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NCH; ++i) __builtin_nontemporal_store(cur[i], dst + t + T * i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) cur[i] = nxt[i];
            __syncthreads();
            const unsigned gn = s_next[(it + 1) & 1];
            if (gn < nvec) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) nxt[i] = __builtin_nontemporal_load(in + (size_t)gn * (VB / 16) + t + T * i);
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) __builtin_nontemporal_store(cur[i], dst + t + T * i);
        } else {
            constexpr int PER = NCH / SUB;
#pragma unroll
            for (int s = 0; s < SUB; ++s) {
#pragma unroll
                for (int i = 0; i < PER; ++i) cur[s * PER + i] = __builtin_nontemporal_load(src + t + T * (s * PER + i));
                if (s > 0) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) __builtin_nontemporal_store(cur[(s - 1) * PER + i], dst + t + T * ((s - 1) * PER + i));
                }
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) __builtin_nontemporal_store(cur[(SUB - 1) * PER + i], dst + t + T * ((SUB - 1) * PER + i));
            __syncthreads();
        }
        if (MODE != 1) __syncthreads();
        g = s_next[(it + 1) & 1];
        __syncthreads();
    }
}

template <int VB, int T, int MODE, int SUB = 1>
void run(const char* name, const V4* in, V4* out, size_t bytes, int wpc) {
    static unsigned* ctr = nullptr; if (!ctr) CK(hipMalloc((void**)&ctr, 64));
    const unsigned nvec = (unsigned)(bytes / VB);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = skel<VB, T, MODE, SUB>;
    for (int r = 0; r < 3; ++r) { CK(hipMemsetAsync(ctr, 0, 64)); k<<<256 * wpc, T>>>(in, out, nvec, ctr); }
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) { CK(hipMemsetAsync(ctr, 0, 64)); k<<<256 * wpc, T>>>(in, out, nvec, ctr); }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s VB %6d T %4d WG/CU %d: %.3f of 8 TB/s\n", name, VB, T, wpc, 2.0 * nvec * VB * 10 / (ms * 1e-3) / 8e12);
}

// one wavefront per vector, static stride, `wpc` wavefronts per CU (the N = 1024 kernels: 8 / 16 KiB per wavefront): how many
// wavefronts does the skeleton need?
template <int VB>
__global__ void __launch_bounds__(64) skel_wave(const V4* __restrict__ in, V4* __restrict__ out, unsigned nvec) {
    constexpr int NCH = VB / 16 / 64;
    for (unsigned g = blockIdx.x; g < nvec; g += gridDim.x) {
        V4 cur[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) cur[i] = __builtin_nontemporal_load(in + (size_t)g * (VB / 16) + threadIdx.x + 64 * i);
#pragma unroll
        for (int i = 0; i < NCH; ++i) __builtin_nontemporal_store(cur[i], out + (size_t)g * (VB / 16) + threadIdx.x + 64 * i);
    }
}
template <int VB> void run_wave(const V4* in, V4* out, size_t bytes, int wpc) {
    const unsigned nvec = (unsigned)(bytes / VB);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) skel_wave<VB><<<256 * wpc, 64>>>(in, out, nvec);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) skel_wave<VB><<<256 * wpc, 64>>>(in, out, nvec);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one wavefront per %5d-byte vector, static stride, %2d wavefronts per CU: %.3f of 8 TB/s\n", VB, wpc, 2.0 * nvec * VB * 10 / (ms * 1e-3) / 8e12);
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    V4 *in, *out; CK(hipMalloc((void**)&in, bytes)); CK(hipMalloc((void**)&out, bytes));
    CK(hipMemset(in, 1, bytes));
    for (int w : {2, 3, 4, 8}) run<65536, 256, 0>("whole vector: load all, store all", in, out, bytes, w);
    for (int w : {2, 4}) run<65536, 512, 0>("whole vector, 512 threads", in, out, bytes, w);
    for (int w : {1, 2}) run<65536, 1024, 0>("whole vector, 1024 threads", in, out, bytes, w);
    for (int w : {2, 3}) run<65536, 256, 1>("register prefetch of the next vector", in, out, bytes, w);
    for (int w : {2, 3, 4}) run<65536, 256, 2, 2>("2 sub-blocks pipelined", in, out, bytes, w);
    for (int w : {2, 3, 4}) run<65536, 256, 2, 4>("4 sub-blocks pipelined", in, out, bytes, w);
    for (int w : {2, 4}) run<65536, 256, 2, 8>("8 sub-blocks pipelined", in, out, bytes, w);
    for (int w : {2, 4, 8}) run<16384, 256, 0>("16 KiB vectors", in, out, bytes, w);
    for (int w : {2, 4, 8}) run<32768, 256, 0>("32 KiB vectors", in, out, bytes, w);
    for (int w : {1, 2, 4}) run<131072, 512, 0>("128 KiB vectors, 512 threads", in, out, bytes, w);
    for (int w : {2, 4, 8, 16}) run<8192, 64, 0>("8 KiB vectors, one wavefront", in, out, bytes, w);
    for (int w : {4, 6, 8, 9, 12, 16, 24, 32}) run_wave<16384>(in, out, bytes, w);
    for (int w : {8, 12, 16, 24, 32}) run_wave<8192>(in, out, bytes, w);
    return 0;
}
