// HBM streaming micro-benchmarks for MI355X: what does a copy reach with the access pattern of the
// FFT kernels?  (development tool; build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float V4;
typedef __attribute__((ext_vector_type(2))) float V2;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// (a) grid-stride float4 copy
template <int NT>
__global__ void copy_gs(const V4* __restrict__ in, V4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

// (b) wave-chunk copy: each wave moves CH x 1 KiB contiguous (CH float4 per lane), waves persistent
template <int CH, int NT, int UNROLL2>
__global__ void copy_wave(const V4* in, V4* out, unsigned nchunks) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned nw = gridDim.x * (blockDim.x >> 6);
    for (unsigned c = wave; c < nchunks; c += nw) {
        const V4* s = in + (size_t)c * CH * 64 + lane;
        V4* d = out + (size_t)c * CH * 64 + lane;
        V4 v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
        for (int j = 0; j < CH; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + 64 * j); else d[64 * j] = v[j]; }
    }
}

// (c) like (b) but with a software prefetch of the next chunk before storing the current one
template <int CH, int NT>
__global__ void copy_wave_pf(const V4* in, V4* out, unsigned nchunks) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned nw = gridDim.x * (blockDim.x >> 6);
    V4 v[CH], w[CH];
    unsigned c = wave;
    if (c < nchunks) {
        const V4* s = in + (size_t)c * CH * 64 + lane;
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
    }
    for (; c < nchunks; c += nw) {
        unsigned cn = c + nw;
        if (cn < nchunks) {
            const V4* s = in + (size_t)cn * CH * 64 + lane;
#pragma unroll
            for (int j = 0; j < CH; ++j) w[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
        }
        V4* d = out + (size_t)c * CH * 64 + lane;
#pragma unroll
        for (int j = 0; j < CH; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + 64 * j); else d[64 * j] = v[j]; }
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = w[j];
    }
}

// (d) 8-byte-per-lane variant of (b)
template <int CH, int NT>
__global__ void copy_wave8(const V2* in, V2* out, unsigned nchunks) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned nw = gridDim.x * (blockDim.x >> 6);
    for (unsigned c = wave; c < nchunks; c += nw) {
        const V2* s = in + (size_t)c * CH * 64 + lane;
        V2* d = out + (size_t)c * CH * 64 + lane;
        V2 v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
        for (int j = 0; j < CH; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + 64 * j); else d[64 * j] = v[j]; }
    }
}

// (e) non-persistent wave-chunk: block b, wave w handles chunk b*wavesPerBlock + w and exits
template <int CH, int NT>
__global__ void copy_wave_np(const V4* in, V4* out, unsigned nchunks) {
    const int lane = threadIdx.x & 63;
    const unsigned c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const V4* s = in + (size_t)c * CH * 64 + lane;
    V4* d = out + (size_t)c * CH * 64 + lane;
    V4 v[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
    for (int j = 0; j < CH; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + 64 * j); else d[64 * j] = v[j]; }
}

// (e2) non-persistent wave-chunk + an 8 KiB LDS table fill per block (what a twiddle table costs)
template <int CH>
__global__ void copy_wave_np_tab(const V4* in, V4* out, unsigned nchunks, const V4* tab, float* sink) {
    __shared__ V4 lt[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const V4* s = in + (size_t)c * CH * 64 + lane;
    V4* d = out + (size_t)c * CH * 64 + lane;
    V4 v[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] = __builtin_nontemporal_load(s + 64 * j);
#pragma unroll
    for (int j = 0; j < CH; ++j) { v[j] += lt[(lane * 7 + j) & 511]; __builtin_nontemporal_store(v[j], d + 64 * j); }
}

// (f) persistent workgroups pulling blocks of chunks from an atomic counter (in-order work distribution)
template <int CH, int NT>
__global__ void copy_wave_dyn(const V4* in, V4* out, unsigned nchunks, unsigned* counter) {
    __shared__ unsigned s_next;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    for (;;) {
        if (threadIdx.x == 0) s_next = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned blk = s_next;
        __syncthreads();
        const unsigned c = blk * wpb + wv;
        if (blk * wpb >= nchunks) return;
        if (c < nchunks) {
            const V4* s = in + (size_t)c * CH * 64 + lane;
            V4* d = out + (size_t)c * CH * 64 + lane;
            V4 v[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = NT ? __builtin_nontemporal_load(s + 64 * j) : s[64 * j];
#pragma unroll
            for (int j = 0; j < CH; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + 64 * j); else d[64 * j] = v[j]; }
        }
    }
}

// read-only and write-only streams
__global__ void read_only(const V4* __restrict__ in, float* sink, size_t n) {
    V4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[0] = 1;
}
__global__ void write_only(V4* __restrict__ out, size_t n) {
    V4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

template <typename F>
static double time_ms(F&& launch, int reps = 8) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    size_t gib = argc > 1 ? atoi(argv[1]) : 4;
    size_t bytes = gib << 30;
    size_t n4 = bytes / 16;
    float *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz, buffer %zu GiB each\n", p.name, cus, p.clockRate / 1000, gib);
    auto rep = [&](const char* name, double ms, double factor = 2.0) {
        printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, factor * bytes / ms / 1e6);
    };
    rep("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }));
    for (int bpc : {4, 8, 16, 32}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride f4 plain, 256thr x %d/CU", bpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_gs<0>), dim3(cus * bpc), dim3(256), 0, 0, (const V4*)in, (V4*)out, n4); }));
        snprintf(nm, sizeof nm, "grid-stride f4 nontemporal, 256thr x %d/CU", bpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_gs<1>), dim3(cus * bpc), dim3(256), 0, 0, (const V4*)in, (V4*)out, n4); }));
    }
    rep("grid-stride f4 plain, one block per 256 elems", time_ms([&] { hipLaunchKernelGGL((copy_gs<0>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const V4*)in, (V4*)out, n4); }));
    rep("grid-stride f4 nt, one block per 256 elems", time_ms([&] { hipLaunchKernelGGL((copy_gs<1>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const V4*)in, (V4*)out, n4); }));
    unsigned nch8 = (unsigned)(n4 / (8 * 64));
    for (int wpc : {8, 16, 24, 32}) {
        char nm[96];
        int blocks = cus * wpc / 8;  // 512-thread blocks = 8 waves
        snprintf(nm, sizeof nm, "wave-chunk 8KiB plain, %d waves/CU", wpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave<8, 0, 0>), dim3(blocks), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8); }));
        snprintf(nm, sizeof nm, "wave-chunk 8KiB nt, %d waves/CU", wpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave<8, 1, 0>), dim3(blocks), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8); }));
        snprintf(nm, sizeof nm, "wave-chunk 8KiB nt + prefetch, %d waves/CU", wpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave_pf<8, 1>), dim3(blocks), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8); }));
        snprintf(nm, sizeof nm, "wave-chunk 8KiB plain + prefetch, %d waves/CU", wpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave_pf<8, 0>), dim3(blocks), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8); }));
    }
    {
        unsigned nch = (unsigned)(n4 / (16 * 64));
        rep("wave-chunk 16KiB nt, 16 waves/CU", time_ms([&] { hipLaunchKernelGGL((copy_wave<16, 1, 0>), dim3(cus * 2), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch); }));
        unsigned nch4 = (unsigned)(n4 / (4 * 64));
        rep("wave-chunk 4KiB nt, 16 waves/CU", time_ms([&] { hipLaunchKernelGGL((copy_wave<4, 1, 0>), dim3(cus * 2), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch4); }));
        unsigned nc8 = (unsigned)(bytes / 8 / (16 * 64));
        rep("wave-chunk 8B/lane x16 (8KiB) nt, 16 waves/CU", time_ms([&] { hipLaunchKernelGGL((copy_wave8<16, 1>), dim3(cus * 2), dim3(512), 0, 0, (const V2*)in, (V2*)out, nc8); }));
        rep("wave-chunk 8B/lane x16 (8KiB) plain, 16 w/CU", time_ms([&] { hipLaunchKernelGGL((copy_wave8<16, 0>), dim3(cus * 2), dim3(512), 0, 0, (const V2*)in, (V2*)out, nc8); }));
    }
    for (int bpc : {1, 2, 3, 5, 6}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride f4 nontemporal, 256thr x %d/CU", bpc);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_gs<1>), dim3(cus * bpc), dim3(256), 0, 0, (const V4*)in, (V4*)out, n4); }));
    }
    for (int wpb : {1, 4, 8, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "NON-persistent wave-chunk 8KiB nt, %d waves/block", wpb);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave_np<8, 1>), dim3((nch8 + wpb - 1) / wpb), dim3(64 * wpb), 0, 0, (const V4*)in, (V4*)out, nch8); }));
        snprintf(nm, sizeof nm, "NON-persistent wave-chunk 8KiB plain, %d waves/block", wpb);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_wave_np<8, 0>), dim3((nch8 + wpb - 1) / wpb), dim3(64 * wpb), 0, 0, (const V4*)in, (V4*)out, nch8); }));
    }
    {
        unsigned nch1 = (unsigned)(n4 / 64), nch2 = (unsigned)(n4 / 128), nch4 = (unsigned)(n4 / 256);
        rep("NON-persistent wave-chunk 1KiB nt, 4 waves/block", time_ms([&] { hipLaunchKernelGGL((copy_wave_np<1, 1>), dim3(nch1 / 4), dim3(256), 0, 0, (const V4*)in, (V4*)out, nch1); }));
        rep("NON-persistent wave-chunk 2KiB nt, 4 waves/block", time_ms([&] { hipLaunchKernelGGL((copy_wave_np<2, 1>), dim3(nch2 / 4), dim3(256), 0, 0, (const V4*)in, (V4*)out, nch2); }));
        rep("NON-persistent wave-chunk 4KiB nt, 4 waves/block", time_ms([&] { hipLaunchKernelGGL((copy_wave_np<4, 1>), dim3(nch4 / 4), dim3(256), 0, 0, (const V4*)in, (V4*)out, nch4); }));
        float* tab; CK(hipMalloc(&tab, 8192)); CK(hipMemset(tab, 0, 8192));
        rep("NON-persistent 8KiB nt + 8KiB LDS table/block, 4 w/blk", time_ms([&] { hipLaunchKernelGGL((copy_wave_np_tab<8>), dim3(nch8 / 4), dim3(256), 0, 0, (const V4*)in, (V4*)out, nch8, (const V4*)tab, out); }));
        rep("NON-persistent 8KiB nt + 8KiB LDS table/block, 8 w/blk", time_ms([&] { hipLaunchKernelGGL((copy_wave_np_tab<8>), dim3(nch8 / 8), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8, (const V4*)tab, out); }));
        unsigned* ctr; CK(hipMalloc(&ctr, 4));
        for (int wpc : {8, 16}) {
            char nm[96];
            snprintf(nm, sizeof nm, "dynamic (atomic) wave-chunk 8KiB nt, %d waves/CU, 8 w/blk", wpc);
            rep(nm, time_ms([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((copy_wave_dyn<8, 1>), dim3(cus * wpc / 8), dim3(512), 0, 0, (const V4*)in, (V4*)out, nch8, ctr); }));
            snprintf(nm, sizeof nm, "dynamic (atomic) wave-chunk 8KiB nt, %d waves/CU, 4 w/blk", wpc);
            rep(nm, time_ms([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((copy_wave_dyn<8, 1>), dim3(cus * wpc / 4), dim3(256), 0, 0, (const V4*)in, (V4*)out, nch8, ctr); }));
        }
    }
    rep("in-place wave-chunk 8KiB nt, 16 waves/CU", time_ms([&] { hipLaunchKernelGGL((copy_wave<8, 1, 0>), dim3(cus * 2), dim3(512), 0, 0, (const V4*)in, (V4*)in, nch8); }));
    rep("read-only f4, 16 blk/CU", time_ms([&] { hipLaunchKernelGGL(read_only, dim3(cus * 16), dim3(256), 0, 0, (const V4*)in, out, n4); }), 1.0);
    rep("write-only f4, 16 blk/CU", time_ms([&] { hipLaunchKernelGGL(write_only, dim3(cus * 16), dim3(256), 0, 0, (V4*)out, n4); }), 1.0);
    return 0;
}
