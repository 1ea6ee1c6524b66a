// What does HBM give for the access pattern of the tile passes (fft_tile.h)?  Pure copies, no arithmetic:
//   A: tile = RUN bytes x L rows at row stride `pitch` in AND out (pass A)      B: contiguous rows in, runs out (pass B)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float V4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// RUN16 = 16-byte units per run (8 -> 128 B, 16 -> 256 B, 4 -> 64 B); L rows per tile; threads = L/8 * RUN16
template <int RUN16, int L, int MODE, int DYNK = 0>
__global__ void __launch_bounds__(L / 8 * RUN16)
tile_copy(const V4* __restrict__ in, V4* __restrict__ out, unsigned long long ntiles, unsigned tiles_per_vec, unsigned long long pitch16, unsigned long long vec16, unsigned* ctr = nullptr) {
    constexpr int TPT = L / 8, WG = TPT * RUN16;
    __shared__ V4 img[L * RUN16];
    __shared__ unsigned s_grab;
    const int tid = threadIdx.x, t = tid / RUN16, p = tid % RUN16;
    unsigned long long tile = blockIdx.x, chunk_end = 0;
    if (DYNK) tile = 0;
    for (;;) {
        if (DYNK) {
            if (tile >= chunk_end) {          // grab the next DYNK adjacent tiles
                __syncthreads();
                if (tid == 0) s_grab = atomicAdd(ctr, 1u);
                __syncthreads();
                tile = (unsigned long long)s_grab * DYNK; chunk_end = tile + DYNK;
            }
            if (tile >= ntiles) break;
        } else if (tile >= ntiles) break;
        const unsigned long long tile_next = DYNK ? tile + 1 : tile + gridDim.x;
        {
        const unsigned a = (unsigned)(tile % tiles_per_vec);
        const unsigned long long vec = tile / tiles_per_vec;
        V4 v[8];
        if (MODE == 0 || MODE == 2) {
            const V4* src = in + vec * vec16 + (unsigned long long)a * RUN16;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = __builtin_nontemporal_load(src + (unsigned long long)(t + TPT * m) * pitch16 + p);
        } else if (MODE == 3) {
            // tile-contiguous intermediate: per band of 2 RUN16 columns a chunk of (2 RUN16) rows x (2 RUN16) columns x 8 bytes
            constexpr int UPB = (2 * RUN16) * (2 * RUN16) / 2;
            const V4* src = in + vec * vec16 + (unsigned long long)a * UPB;
#pragma unroll
            for (int m = 0; m < 8; ++m) { const int g = tid + WG * m; v[m] = __builtin_nontemporal_load(src + (unsigned long long)(g / UPB) * (L * RUN16) + g % UPB); }
        } else {
            // rows: tile = 2*RUN16 rows (8-byte elements) x L points, contiguous rows of L*8 bytes = L/2 units
            const V4* src = in + vec * vec16 + (unsigned long long)a * (2 * RUN16) * (L / 2);
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = __builtin_nontemporal_load(src + tid + WG * m);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) img[((t + TPT * m) * RUN16 + (p ^ ((t + TPT * m) & (RUN16 - 1))))] = v[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) { const int g = tid + WG * m, pt = g / RUN16, pu = g % RUN16; v[m] = img[pt * RUN16 + (pu ^ (pt & (RUN16 - 1)))]; }
        __syncthreads();
        if (MODE == 2) {   // dense: the tile as one contiguous block
            V4* dst = out + vec * vec16 + (unsigned long long)a * (L * RUN16);
#pragma unroll
            for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(v[m], dst + tid + WG * m);
        } else {
        V4* dst = out + vec * vec16 + (unsigned long long)a * RUN16;
#pragma unroll
        for (int m = 0; m < 8; ++m) { const int g = tid + WG * m, pt = g / RUN16, pu = g % RUN16; __builtin_nontemporal_store(v[m], dst + (unsigned long long)pt * pitch16 + pu); }
        }
        }
        tile = tile_next;
    }
}

template <int RUN16, int L, int MODE, int DYNK = 0>
void run(const char* name, const V4* in, V4* out, size_t bytes, int wgs_per_cu) {
    static unsigned* ctr = nullptr; if (!ctr) CK(hipMalloc((void**)&ctr, 64));
    // vectors of L x L2 complex floats with L2 = L (square), pitch = L2 * 8 bytes
    const unsigned long long L2 = L, pitch16 = L2 * 8 / 16, vec16 = (unsigned long long)L * pitch16;
    const unsigned tiles_per_vec = (MODE == 0 || MODE == 2) ? (unsigned)(pitch16 / RUN16) : (unsigned)(L / (2 * RUN16));
    const unsigned long long nvec = bytes / (vec16 * 16), ntiles = nvec * tiles_per_vec;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = tile_copy<RUN16, L, MODE, DYNK>;
    CK(hipMemset(ctr, 0, 64));
    k<<<256 * wgs_per_cu, L / 8 * RUN16>>>(in, out, ntiles, tiles_per_vec, pitch16, vec16, ctr);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) { CK(hipMemsetAsync(ctr, 0, 64)); k<<<256 * wgs_per_cu, L / 8 * RUN16>>>(in, out, ntiles, tiles_per_vec, pitch16, vec16, ctr); }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %d WG/CU: %7.1f GB/s = %.3f of 8 TB/s\n", name, wgs_per_cu, 2.0 * nvec * vec16 * 16 * 5 / (ms * 1e-3) / 1e9, 2.0 * nvec * vec16 * 16 * 5 / (ms * 1e-3) / 8e12);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    V4 *in, *out; CK(hipMalloc((void**)&in, bytes)); CK(hipMalloc((void**)&out, bytes));
    CK(hipMemset(in, 1, bytes));
    for (int rep = 0; rep < 2; ++rep)
    for (int w : {2, 3}) {
        run<8, 256, 0, 4>("A : 128-B runs in and out, L=256, K=4", in, out, bytes, w);
        run<8, 256, 2, 4>("A': 128-B runs in, dense tile out, K=4", in, out, bytes, w);
        run<8, 256, 1, 4>("B : rows in, 128-B runs out, K=4", in, out, bytes, w);
        run<8, 256, 3, 4>("B': 2-KiB chunks in, 128-B runs out, K=4", in, out, bytes, w);
        run<8, 256, 0>("A : static", in, out, bytes, w);
        run<8, 256, 2>("A': static", in, out, bytes, w);
        run<8, 256, 1>("B : static", in, out, bytes, w);
        run<8, 256, 3>("B': static", in, out, bytes, w);
    }
    run<4, 1024, 0, 4>("A : 64-B runs, L=1024, K=4", in, out, bytes, 1);
    run<4, 1024, 2, 4>("A': 64-B runs in, dense out, L=1024", in, out, bytes, 1);
    run<4, 1024, 1, 4>("B : rows in, 64-B runs out, L=1024", in, out, bytes, 1);
    run<4, 1024, 3, 4>("B': 512-B chunks in, 64-B runs out, L=1024", in, out, bytes, 1);
    return 0;
}
