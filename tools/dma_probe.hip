// LDS-DMA probe for MI355X (development tool): is `global_load_lds_dwordx4` usable for a double-buffered streaming
// kernel with ONE 512-thread workgroup per CU (the shape the multi-wave FFT kernels need: 64 KiB per group, two landing
// buffers, the second one above 64 KiB of LDS)?  Checks correctness of the copy (M0 range, counted vmcnt) and measures
// the HBM rate of   DMA -> LDS -> ds_read_b128 -> [optional LDS exchange work] -> global store.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float V4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }

constexpr int WG = 512, WAVES = 8, GROUP = 65536, PIECES = GROUP / 1024, PPW = PIECES / WAVES;  // 8 pieces per wave
constexpr int BUF = 73808;   // the FFT kernel's padded image size: second buffer starts above 64 KiB

// MODE 0: counted vmcnt (stores younger than the DMA stay in flight)   MODE 1: vmcnt(0)
// XCH: number of extra LDS exchange rounds (ds_write_b64 x16 + ds_read_b64 x16 per thread, like an FFT stage exchange)
template <int MODE, int XCH, int DYN, int VALU>
__global__ void __launch_bounds__(WG, 1)
dma_copy(const char* __restrict__ in, char* __restrict__ out, unsigned ngroups, unsigned* ctr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* s_next = reinterpret_cast<unsigned*>(smem + 2 * BUF);
    unsigned pend = 0, g = blockIdx.x;
    if (DYN) {
        if (tid == 0) { s_next[0] = atomicAdd(ctr, 1u); pend = atomicAdd(ctr, 1u); }
        __syncthreads();
        g = s_next[0];
    }
    auto issue = [&](unsigned grp, int b) {
        const unsigned gg = grp < ngroups ? grp : ngroups - 1;
        const char* src = in + (size_t)gg * GROUP + lane * 16;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + WAVES * i;
            glds16(src + p * 1024, lds0 + b * BUF + p * 1024);
        }
    };
    issue(g, 0);
    int b = 0;
    for (unsigned it = 0; g < ngroups; ++it) {
        if (DYN && tid == 0) { s_next[(it + 1) & 1] = pend; pend = atomicAdd(ctr, 1u); }
        // landing(g) complete: the DMA pieces are older than the 8 stores of the previous iteration
        if (MODE == 0 && it > 0) wait_vm<PPW>(); else wait_vm<0>();
        wait_lgkm0();
        raw_barrier();
        const unsigned gn = DYN ? s_next[(it + 1) & 1] : g + gridDim.x;
        issue(gn, b ^ 1);
        const V4* L = reinterpret_cast<const V4*>(smem + b * BUF);
        V4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = L[tid + WG * i];
        if (XCH > 0) {
            // realistic stage exchange: conflict-free ds_write_b64 x16 in "column" order, ds_read_b64 x16 in "row" order
            // through a padded image (stride 17 float2 per 16), VALU work of a radix-8/16 stage in between
            float2* X = reinterpret_cast<float2*>(smem + b * BUF);
#pragma unroll
            for (int r = 0; r < XCH; ++r) {
                if (VALU) {
#pragma unroll
                    for (int k = 0; k < VALU; ++k)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const V4 t = v[i];
                            v[i].x = __builtin_fmaf(t.y, 0.70710678f, t.x); v[i].y = __builtin_fmaf(t.x, -0.70710678f, t.y);
                            v[i].z = __builtin_fmaf(t.w, 0.38268343f, t.z); v[i].w = __builtin_fmaf(t.z, -0.38268343f, t.w);
                        }
                }
                wait_lgkm0(); raw_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int p0 = tid + WG * (2 * i), p1 = tid + WG * (2 * i + 1);
                    X[p0 + (p0 >> 4)] = float2{v[i].x, v[i].y};
                    X[p1 + (p1 >> 4)] = float2{v[i].z, v[i].w};
                }
                wait_lgkm0(); raw_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q0 = 16 * tid + 2 * i, q1 = q0 + 1;
                    const float2 a = X[q0 + (q0 >> 4)], c = X[q1 + (q1 >> 4)];
                    v[i] = V4{a.x, a.y, c.x, c.y};
                }
            }
        }
        V4* d = reinterpret_cast<V4*>(out + (size_t)g * GROUP);
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v[i], d + tid + WG * i);
        g = gn;
        b ^= 1;
    }
    wait_vm<0>();
    if (DYN && tid == 0) {
        __threadfence();
        unsigned dn = atomicAdd(&ctr[1], 1u);
        if (dn == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }
    }
}

template <int MODE, int XCH, int DYN, int VALU = 0>
static void run(const char* name, const char* din, char* dout, size_t bytes, unsigned* ctr, const std::vector<float>& h_in) {
    const unsigned ngroups = (unsigned)(bytes / GROUP);
    auto k = dma_copy<MODE, XCH, DYN, VALU>;
    const size_t lds = 2 * BUF + 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dout, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(WG), lds, 0, din, dout, ngroups, ctr);
    CK(hipDeviceSynchronize());
    // verify (sampled + first / last group)
    std::vector<float> h((size_t)GROUP / 4);
    size_t bad = 0;
    if (XCH == 0) for (unsigned gi : {0u, 1u, 255u, 256u, ngroups / 2, ngroups - 2, ngroups - 1}) {
        CK(hipMemcpy(h.data(), dout + (size_t)gi * GROUP, GROUP, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) if (h[i] != h_in[(size_t)gi * (GROUP / 4) + i]) ++bad;
    }
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(WG), lds, 0, din, dout, ngroups, ctr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbps = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-44s %8.1f GB/s  %.3f of 8 TB/s   mismatches %zu\n", name, gbps, gbps / 8000.0, bad);
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    std::vector<float> h(bytes / 4);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8); }
    char *din, *dout; unsigned* ctr;
    CK(hipMalloc((void**)&din, bytes)); CK(hipMalloc((void**)&dout, bytes)); CK(hipMalloc((void**)&ctr, 64));
    CK(hipMemset(ctr, 0, 64));
    CK(hipMemcpy(din, h.data(), bytes, hipMemcpyHostToDevice));
    run<0, 0, 0>("static, counted vmcnt", din, dout, bytes, ctr, h);
    run<1, 0, 0>("static, vmcnt(0)", din, dout, bytes, ctr, h);
    run<0, 0, 1>("in-order, counted vmcnt", din, dout, bytes, ctr, h);
    run<1, 0, 1>("in-order, vmcnt(0)", din, dout, bytes, ctr, h);
    run<0, 1, 1>("in-order, +1 exchange", din, dout, bytes, ctr, h);
    run<0, 2, 1>("in-order, +2 exchanges", din, dout, bytes, ctr, h);
    run<0, 3, 1>("in-order, +3 exchanges", din, dout, bytes, ctr, h);
    run<0, 4, 1>("in-order, +4 exchanges", din, dout, bytes, ctr, h);
    run<0, 3, 1, 6>("in-order, +3 exchanges, 3x192 VALU/thread", din, dout, bytes, ctr, h);
    run<0, 3, 1, 12>("in-order, +3 exchanges, 3x384 VALU/thread", din, dout, bytes, ctr, h);
    run<0, 4, 1, 8>("in-order, +4 exchanges, 4x256 VALU/thread", din, dout, bytes, ctr, h);
    run<0, 4, 1, 16>("in-order, +4 exchanges, 4x512 VALU/thread", din, dout, bytes, ctr, h);
    run<0, 7, 1, 8>("in-order, +7 exchanges, 7x256 VALU (FIR-like)", din, dout, bytes, ctr, h);
    run<1, 3, 1, 6>("vmcnt(0), +3 exchanges, 3x192 VALU/thread", din, dout, bytes, ctr, h);
    return 0;
}
