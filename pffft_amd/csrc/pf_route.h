// The planner's records: what runs for a (setup, direction, layout), decided ONCE at pffft_new_setup and stored in the setup
// (round 5: the decisions used to be re-derived per call from ~60 scattered tests of an integer selector and ~40 environment
// reads).  Three pieces:
//   Env    every environment switch of the product build, read once;
//   AbSel  the A/B selector of pffft_hip_set_variant() decoded into named flags - the alternatives that still exist because a test
//          holds two independent routes to one answer; the measured-and-lost ones are gone from the product build;
//   Route  kernel family + configuration + launch rule per (direction, layout); pffft_hip_describe() prints it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>

namespace pf {

// ------------------------------------------------------------------------------------------------ environment
// The product build reads exactly these names (once, at first use).  Development builds (-DPFFFT_HIP_VARIANTS) keep further tuning
// knobs behind dev_env(): in the product build that is a constant.
struct Env {
    bool abort_on_error;   // PFFFT_HIP_ABORT=1        legacy void entries abort() instead of failing soft
    bool zero_copy;        // PFFFT_HIP_NO_ZEROCOPY=1  host-pointer calls through device staging instead of pinned images
    int oneshot;           // PFFFT_HIP_ONESHOT=<k>    launches of up to k groups per resident workgroup run one group per workgroup in
                           //                          dispatch order instead of the persistent in-order loop (default 4, 0 = never)
    int c1024_rounds;      // PFFFT_HIP_C1024_ONCE=<r> N = 1024 float: up to r resident sets of wavefronts as one transform per wavefront (0 = never)
    int tile_plans;        // PFFFT_HIP_TILE_PLANS=0   no run-time mixed-radix tile plans (fft_tileg.h): those sizes take the streaming passes
    const char* tile_force;  // PFFFT_HIP_TILE_FORCE="L1[g],L2[g]"  the two tile lengths of the sizes they multiply to (tests)
    int fir_nfft;          // PFFASTCONV_HIP_NFFT=<n>  internal block length of the throughput regime (0 = the reference's, -1 = planner)
    int fir_xcd;           // PFFASTCONV_HIP_XCD=0     split FIR kernels without the XCD-contiguous block ranges
};
const Env& env();

#ifdef PFFFT_HIP_VARIANTS
inline int dev_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int dev_env(const char*, int dflt) { return dflt; }
#endif

// ------------------------------------------------------------------------------------------------ A/B selector
// pffft_hip_set_variant(v), thread-local.  0 = the planner's routes.  The values that remain select an ALTERNATIVE ROUTE to the same
// result that a parity test compares with the default one (tests/test_gpu_*.py), or a development-build twin.
enum AbValue : int {
    AB_DEFAULT = 0,
    AB_INORDER_SMALL = 42,      // Stockham / zreorder / zconvolve kernels: in-order chunks where the default is a static stride
    AB_STATIC_LARGE = 43,       // Stockham: static stride where the default pulls in order
    AB_STOCK_FOR_TILED = 50,    // the Stockham plan also for the sizes that have a register-tiled kernel (and N = 1024 float)
    AB_STOCK_WORKGROUP = 52,    // (development build) workgroup-phase Stockham kernel where the wave-local one applies
    AB_STOCK_RUNTIME = 53,      // (development build) run-time-plan Stockham kernels
    AB_STOCK_DF_ON = 54,        // (development build) direct-first-stage twin of every Stockham plan
    AB_STOCK_DF_OFF = 55,       // (development build) deposit twin of every plan
    AB_AUX_DIRECT = 60,         // zreorder / zconvolve: the direct grid-stride kernels; shift + FFT as two passes
    AB_AUX_NO_STREAM = 61,      // zreorder / zconvolve: no in-order streaming kernel
    AB_BIG_STRIDED = 80,        // beyond LDS: the balanced two-pass strided kernels
    AB_BIG_NO_TILES = 82,       // beyond LDS: never the tile passes (three / five streaming sweeps)
    AB_BIG_NO_MR_TILES = 83,    // beyond LDS: tile passes for power-of-two n only (the route the mixed-radix plans replaced)
    AB_BIG_SEPARATE_LAYOUT = 86,  // beyond LDS: the internal layout through its own sweep instead of fused into the first / last pass
    AB_BIG_SEPARATE_SWEEPS = 87,  // beyond LDS: zreorder kernel + in-place pair pass instead of the one-sweep block kernels
    AB_FIR_PARTITIONED = 88,    // (development build) uniformly partitioned FIR kernel
    AB_NO_TINY = 91,            // n = 16 / 32 on the Stockham plan instead of one thread per transform
    AB_FIR_LOCKSTEP = 97,       // (development build) 16384-sample FIR blocks on the lock-step LDS-DMA kernel
    AB_FIR_FEW_16PT = 114,      // (development build) few-block FIR: 16 points per thread
    AB_FIR_FEW_LOCKSTEP = 115,  // few-block FIR: the lock-step kernel on 512 / 256 threads instead of the split one
    AB_FIR_SPLIT_PLAIN = 116,   // (development build) split FIR kernel with plain barriers / pieces at once
    AB_FIR_FUSED32_PF1 = 117,   // (development build) 16384-sample FIR blocks, fft_fir32.h: the next block requested at once after the product
    AB_FIR_FUSED32_NOPF = 118,  // (development build) ... after the output stores
    AB_FIR_SPLIT = 119,         // 16384-sample FIR blocks on the split kernel (fft_split.h) that fft_fir32.h replaced in round 6
    AB_CONV_COMPOSED = 120,     // pffft_hip_convolve_batch as the three batched entries
    AB_RFFT_THREE = 121,        // real transforms beyond LDS: always complex core + pair sweep
    AB_RFFT_TWO = 122,          // real transforms beyond LDS: two sweeps wherever the length splits
    AB_NO_ONE_IMAGE = 123,      // the sizes of the single-image kernel (fft_one.h) on the tile / streaming passes they ran on before round 6
    AB_FAKE_DEVICE = 130,       // the calling thread counts as being on ANOTHER device than its current one (key + 64): exercises the per-device
                                // replicas of a shared setup on a box with one GPU (tests/test_gpu_round6.py)
};
struct AbSel {
    int raw = 0;
    bool is(AbValue v) const { return raw == (int)v; }
    bool any() const { return raw != 0; }
};
AbSel ab();   // the calling thread's selector

// ------------------------------------------------------------------------------------------------ routes
enum Family : uint8_t { FAM_NONE = 0, FAM_TINY, FAM_C1024, FAM_TILED, FAM_STOCK, FAM_BIG, FAM_ONE };
const char* family_name(Family f);

// launch rule of an LDS-resident kernel
enum LaunchRule : uint8_t {
    LR_DISPATCH,       // one group per workgroup (or wavefront), workgroups in hardware dispatch order
    LR_STATIC,         // persistent workgroups on a static stride
    LR_INORDER,        // persistent workgroups pull groups in order from a counter (first two groups static);
                       // launches of <= `oneshot` groups per resident workgroup run as LR_DISPATCH
};
const char* launch_rule_name(LaunchRule r);

struct TiledSel {            // a register-tiled configuration (fft_tiled.h), type-erased
    const void* fn = nullptr;
    size_t lds = 0;
    int wg = 0, t_per_wg = 0;
    const char* cfg = "";
};

struct StockSel {            // a Stockham plan on its compile-time kernel (fft_stock.h)
    const void* fn = nullptr;    // StockCtFn<T>; nullptr: run-time-plan kernel (development build) or none
    bool wl = false;             // wave-local organisation (vectors <= 4 KiB)
    bool df = false;             // direct first stage
    int threads = 0;
    size_t lds = 0;
    int flags = 0;
    int groups_per_wg = 0;       // measured table (stock_grid_gen.h); 0: the size rule (grid_mul x the resident set)
    int grid_mul = 1;
};

enum BigCore : uint8_t { BIG_TILES, BIG_RFFT2, BIG_STREAM, BIG_STRIDED };
struct BigPlan {             // n beyond LDS: the sweeps over HBM, in order
    BigCore core = BIG_STREAM;
    int tmode = 0;               // planner mode of the tile plans (tile_tu.hip): 0 complex, 1 deep, 2 core of a real transform
    int lens[3] = {0, 0, 0};     // tile lengths (BIG_TILES / BIG_RFFT2), {R, N2} (BIG_STREAM), {N1, N2} (BIG_STRIDED)
    int pre = -1;                // one-sweep block kernel BEFORE the core (fft_big.h big_block_kernel mode), -1 none:
                                 //   1 complex internal -> canonical, 3 real internal -> packed, 4 real canonical -> packed
    bool pre_separate = false;   // AB_BIG_SEPARATE_SWEEPS: zreorder kernel / copy + in-place pair pass instead
    bool fuse_in = false;        // the first tile pass reads the internal layout itself
    bool col_in = false;         // the column pass of the streaming route reads it
    bool fuse_out = false;       // the last pass (tile pass or transpose) stores the internal layout itself
    bool rfuse = false;          // real forward (round 6): the last tile pass carries the pair pass (fft_tile.h RMODE 3) - no pair sweep
    int post = -1;               // block kernel AFTER the core: 0 complex canonical -> internal, 2 real pair pass + internal, 5 permutation
    bool pair_after = false;     // real forward ordered: in-place pair pass on the result
    bool post_separate = false;  // AB_BIG_SEPARATE_SWEEPS: pair pass + zreorder kernel
    int sweeps = 0;              // passes over the whole vector set
};

struct Route {
    Family fam = FAM_NONE;
    LaunchRule rule = LR_DISPATCH;
    int oneshot = 0;             // LR_INORDER: bound of the dispatch-order launch, groups per resident workgroup
    TiledSel tiled;
    StockSel stock;
    BigPlan big;
};

}  // namespace pf
