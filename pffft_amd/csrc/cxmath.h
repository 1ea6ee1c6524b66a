// Complex arithmetic and small-radix DFT butterflies for the gfx950 FFT kernels.
//
// These are the MI355X counterparts of the reference's SIMD macro layer
// (src/simd/pf_sse1_float.h:45-77, VCPLXMUL/VCPLXMULCONJ in src/simd/pf_float.h:76-81)
// and of the arithmetic inside passf2/3/4/5_ps (src/pffft_priv_impl.h:122-321).  They are
// NOT a translation: a butterfly here is a per-lane register DFT of radix 2/3/4/5/8/16 with
// natural-order output, used by one-wavefront-per-transform kernels; the reference's
// butterflies operate on 4-lane vectors sweeping main memory.
#pragma once
#include <hip/hip_runtime.h>

namespace pf {

constexpr int FWD = 0;  // PFFFT_FORWARD  (include/pffft/pffft.h:112) : exp(-2*pi*i*nk/N)
constexpr int BWD = 1;  // PFFFT_BACKWARD                            : exp(+2*pi*i*nk/N), unscaled

// A complex number is a native 2-vector (re, im).  (A struct {T x, y} works arithmetically, but arrays of it
// held in registers went through type-punned struct copies that the optimiser could not always promote out
// of private memory — scratch traffic in the hot loops.  Vectors have built-in +, -, scalar * and are SSA values.)
template <typename T> struct vec2t;
template <> struct vec2t<float> { typedef __attribute__((ext_vector_type(2))) float type; };
template <> struct vec2t<double> { typedef __attribute__((ext_vector_type(2))) double type; };
template <typename T> using vec2 = typename vec2t<T>::type;
template <typename T> using cx = vec2<T>;
template <typename V> struct scalar_of;
template <> struct scalar_of<vec2<float>> { typedef float type; };
template <> struct scalar_of<vec2<double>> { typedef double type; };
template <typename V> using sc = typename scalar_of<V>::type;

template <typename T> __host__ __device__ __forceinline__ cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename V> __device__ __forceinline__ V conj(V a) { return mk<sc<V>>(a.x, -a.y); }

// The library is compiled with -ffp-contract=off and every fused multiply-add is written out, so
// that the arithmetic of a butterfly is fixed by the source: all template instantiations of a kernel
// (ordered / unordered, in place / out of place) then produce bit-identical spectra, which is what
// pffft guarantees between pffft_transform_ordered and pffft_transform + pffft_zreorder
// (same arithmetic, src/pffft_priv_impl.h:1497-1498) and what benchmarks/bench_pffft.c:343-349 asserts.
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- packed FP32 with operand swizzles --------------------------------------------------------------------------------
// gfx950's v_pk_{add,mul,fma}_f32 take, per source, which 32-bit half feeds the low / the high result (op_sel / op_sel_hi)
// and a sign per half (neg_lo / neg_hi).  A complex product is then TWO instructions and "a +- i b" ONE; left to itself the
// compiler packs the arithmetic but builds the swapped / negated operands with v_mov + v_xor first (n = 8192 real forward:
// 219 v_mov + 148 v_xor next to 1045 packed operations).  Each helper computes exactly the operations of its scalar twin
// below (same products, same fused multiply-adds, exact sign flips), so float results are bit-identical to the unfused form.
#ifndef PF_NO_PK_SWIZZLE
#define PF_PK(name, text)                                                                                   \
    __device__ __forceinline__ vec2<float> name(vec2<float> a, vec2<float> b) {                             \
        vec2<float> r;                                                                                      \
        asm(text : "=v"(r) : "v"(a), "v"(b));                                                              \
        return r;                                                                                           \
    }
PF_PK(pk_add_mib, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a - i b = (a.x + b.y, a.y - b.x)
PF_PK(pk_add_pib, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a + i b = (a.x - b.y, a.y + b.x)
PF_PK(pk_add_cj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                  // a + conj(b)
PF_PK(pk_sub_cj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                  // a - conj(b)
PF_PK(pk_mul_yy_yx_n, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]")   // (-a.y b.y, a.y b.x)
PF_PK(pk_mul_yx_yy_n, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]")   // (a.y b.y, -a.x b.y)
#undef PF_PK
__device__ __forceinline__ vec2<float> pk_fma_xx_xy(vec2<float> a, vec2<float> b, vec2<float> c) {   // (a.x b.x + c.x, a.x b.y + c.y)
    vec2<float> r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ vec2<float> pk_fma_xy_xx(vec2<float> a, vec2<float> b, vec2<float> c) {   // (a.x b.x + c.x, a.y b.x + c.y)
    vec2<float> r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif

// a * w
template <typename V> __device__ __forceinline__ V cmul(V a, V w) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return pk_fma_xx_xy(a, w, pk_mul_yy_yx_n(a, w));
    else
#endif
    return mk<sc<V>>(fma_(a.x, w.x, -(a.y * w.y)), fma_(a.x, w.y, a.y * w.x));
}
// a * conj(w)
template <typename V> __device__ __forceinline__ V cmulc(V a, V w) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return pk_fma_xy_xx(a, w, pk_mul_yx_yy_n(a, w));
    else
#endif
    return mk<sc<V>>(fma_(a.x, w.x, a.y * w.y), fma_(a.y, w.x, -(a.x * w.y)));
}
// a * w for the forward transform, a * conj(w) for the backward one (table holds exp(-i*theta))
template <int DIR, typename V> __device__ __forceinline__ V twmul(V a, V w) {
    return DIR == FWD ? cmul(a, w) : cmulc(a, w);
}
// the same for a COMPILE-TIME constant w (the fixed twiddles inside radix 16 / 32 / 9 / 25 / 27): the scalar form lets the
// compiler keep the constants in scalar registers / literals; an asm operand would pin each one in a VGPR pair.  Same
// operations as cmul / cmulc.
template <int DIR, typename V> __device__ __forceinline__ V twmul_c(V a, V w) {
    return DIR == FWD ? mk<sc<V>>(fma_(a.x, w.x, -(a.y * w.y)), fma_(a.x, w.y, a.y * w.x))
                      : mk<sc<V>>(fma_(a.x, w.x, a.y * w.y), fma_(a.y, w.x, -(a.x * w.y)));
}
// multiply by -i (forward) / +i (backward): the radix-4 "quarter turn"
template <int DIR, typename V> __device__ __forceinline__ V rot(V a) {
    return DIR == FWD ? mk<sc<V>>(a.y, -a.x) : mk<sc<V>>(-a.y, a.x);
}

// a + rot<DIR>(b) and a - rot<DIR>(b) (forward: rot = -i): one packed add each in float
template <int DIR, typename V> __device__ __forceinline__ V add_rot(V a, V b) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return DIR == FWD ? pk_add_mib(a, b) : pk_add_pib(a, b);
    else
#endif
    return a + rot<DIR>(b);
}
template <int DIR, typename V> __device__ __forceinline__ V sub_rot(V a, V b) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return DIR == FWD ? pk_add_pib(a, b) : pk_add_mib(a, b);
    else
#endif
    return a - rot<DIR>(b);
}

// a + conj(b), a - conj(b)
template <typename V> __device__ __forceinline__ V add_conj(V a, V b) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return pk_add_cj(a, b);
    else
#endif
    return mk<sc<V>>(a.x + b.x, a.y - b.y);
}
template <typename V> __device__ __forceinline__ V sub_conj(V a, V b) {
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(sc<V>) == 4) return pk_sub_cj(a, b);
    else
#endif
    return mk<sc<V>>(a.x - b.x, a.y + b.y);
}

template <int DIR, typename V> __device__ __forceinline__ void dft2(V& a0, V& a1) {
    V t = a0 - a1; a0 = a0 + a1; a1 = t;
}

template <int DIR, typename V> __device__ __forceinline__ void dft3(V& a0, V& a1, V& a2) {
    typedef sc<V> T;
    const T s3 = (T)0.86602540378443864676372317075294L;  // sin(2*pi/3)
    cx<T> t1 = a1 + a2;
    cx<T> m = mk<T>(fma_((T)-0.5, t1.x, a0.x), fma_((T)-0.5, t1.y, a0.y));
    cx<T> e = (a1 - a2) * s3;            // d = (-/+ i) e
    a0 = a0 + t1; a1 = add_rot<DIR>(m, e); a2 = sub_rot<DIR>(m, e);
}

template <int DIR, typename V>
__device__ __forceinline__ void dft4(V& a0, V& a1, V& a2, V& a3) {
    typedef sc<V> T;
    cx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d3 = a1 - a3;
    a0 = t0 + t2; a1 = add_rot<DIR>(t1, d3); a2 = t0 - t2; a3 = sub_rot<DIR>(t1, d3);
}
// the same with the third input given BEFORE its quarter turn: transforms (a0, a1, rot(d2), a3)
template <int DIR, typename V>
__device__ __forceinline__ void dft4_r2(V& a0, V& a1, V& d2, V& a3) {
    typedef sc<V> T;
    cx<T> t0 = add_rot<DIR>(a0, d2), t1 = sub_rot<DIR>(a0, d2), t2 = a1 + a3, d3 = a1 - a3;
    a0 = t0 + t2; a1 = add_rot<DIR>(t1, d3); d2 = t0 - t2; a3 = sub_rot<DIR>(t1, d3);
}

template <int DIR, typename V>
__device__ __forceinline__ void dft5(V& a0, V& a1, V& a2, V& a3, V& a4) {
    typedef sc<V> T;
    const T c1 = (T)0.30901699437494742410229341718282L;   // cos(2*pi/5)
    const T c2 = (T)-0.80901699437494742410229341718282L;  // cos(4*pi/5)
    const T s1 = (T)0.95105651629515357211643933337938L;   // sin(2*pi/5)
    const T s2 = (T)0.58778525229247312916870595463907L;   // sin(4*pi/5)
    cx<T> p1 = a1 + a4, m1 = a1 - a4, p2 = a2 + a3, m2 = a2 - a3;
    cx<T> u1 = mk<T>(fma_(c2, p2.x, fma_(c1, p1.x, a0.x)), fma_(c2, p2.y, fma_(c1, p1.y, a0.y)));
    cx<T> u2 = mk<T>(fma_(c1, p2.x, fma_(c2, p1.x, a0.x)), fma_(c1, p2.y, fma_(c2, p1.y, a0.y)));
    cx<T> e1 = mk<T>(fma_(s2, m2.x, s1 * m1.x), fma_(s2, m2.y, s1 * m1.y));      // v1 = (-/+ i) e1
    cx<T> e2 = mk<T>(fma_(-s1, m2.x, s2 * m1.x), fma_(-s1, m2.y, s2 * m1.y));     // v2 = (-/+ i) e2
    a0 = a0 + p1 + p2; a1 = add_rot<DIR>(u1, e1); a4 = sub_rot<DIR>(u1, e1); a2 = add_rot<DIR>(u2, e2); a3 = sub_rot<DIR>(u2, e2);
}

// a * exp(-/+ i*pi/4) and a * exp(-/+ 3i*pi/4)
template <int DIR, typename V> __device__ __forceinline__ V mulw8_1(V a) {
    typedef sc<V> T;
    const T h = (T)0.70710678118654752440084436210485L;
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(T) == 4) return (DIR == FWD ? pk_add_mib(a, a) : pk_add_pib(a, a)) * h;   // (a -+ i a) h
    else
#endif
    return DIR == FWD ? mk<T>((a.x + a.y) * h, (a.y - a.x) * h) : mk<T>((a.x - a.y) * h, (a.x + a.y) * h);
}
template <int DIR, typename V> __device__ __forceinline__ V mulw8_3(V a) {
    typedef sc<V> T;
    const T h = (T)0.70710678118654752440084436210485L;
#ifndef PF_NO_PK_SWIZZLE
    if constexpr (sizeof(T) == 4) return (DIR == FWD ? pk_add_pib(a, a) : pk_add_mib(a, a)) * -h;  // -(a +- i a) h, signs exact
    else
#endif
    return DIR == FWD ? mk<T>((a.y - a.x) * h, -(a.x + a.y) * h) : mk<T>(-(a.x + a.y) * h, (a.x - a.y) * h);
}

// in-place radix-8, natural-order output (a[d] = sum_q a[q] W8^(q d))
// R4: a[4] is given BEFORE its quarter turn (radix 16 hands its c[4] over that way)
template <int DIR, bool R4 = false, typename V> __device__ __forceinline__ void dft8(V (&a)[8]) {
    typedef sc<V> T;
    cx<T> b0, c0;
    if constexpr (R4) { b0 = add_rot<DIR>(a[0], a[4]); c0 = sub_rot<DIR>(a[0], a[4]); }
    else { b0 = a[0] + a[4]; c0 = a[0] - a[4]; }
    cx<T> b1 = a[1] + a[5], c1 = mulw8_1<DIR>(a[1] - a[5]);
    cx<T> b2 = a[2] + a[6], c2 = a[2] - a[6];          // (its quarter turn is folded into dft4_r2)
    cx<T> b3 = a[3] + a[7], c3 = mulw8_3<DIR>(a[3] - a[7]);
    dft4<DIR>(b0, b1, b2, b3);
    dft4_r2<DIR>(c0, c1, c2, c3);
    a[0] = b0; a[2] = b1; a[4] = b2; a[6] = b3;
    a[1] = c0; a[3] = c1; a[5] = c2; a[7] = c3;
}

// in-place radix-16, natural-order output
template <int DIR, typename V> __device__ __forceinline__ void dft16(V (&a)[16]) {
    typedef sc<V> T;
    const T c1 = (T)0.92387953251128675612818318939679L;  // cos(pi/8)
    const T s1 = (T)0.38268343236508977172845998403040L;  // sin(pi/8)
    cx<T> b[8], c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { b[q] = a[q] + a[q + 8]; c[q] = a[q] - a[q + 8]; }
    // c[q] *= W16^q (forward: exp(-i*pi*q/8))
    const cx<T> w1 = mk<T>(c1, -s1), w3 = mk<T>(s1, -c1);
    c[1] = twmul_c<DIR>(c[1], w1);
    c[2] = mulw8_1<DIR>(c[2]);
    c[3] = twmul_c<DIR>(c[3], w3);
    // c[4]'s quarter turn is folded into the first butterflies of dft8<DIR, true>
    c[5] = twmul_c<DIR>(c[5], mk<T>(-s1, -c1));
    c[6] = mulw8_3<DIR>(c[6]);
    c[7] = twmul_c<DIR>(c[7], mk<T>(-c1, -s1));
    dft8<DIR>(b);
    dft8<DIR, true>(c);
#pragma unroll
    for (int d = 0; d < 8; ++d) { a[2 * d] = b[d]; a[2 * d + 1] = c[d]; }
}

template <int R, int DIR, typename V> __device__ __forceinline__ void dftR(V (&a)[R]);

// in-place radix-32, natural-order output: four interleaved radix-8 transforms, constant twiddles W32^(q0 k1),
// eight radix-4 transforms across them
template <int DIR, typename V> __device__ __forceinline__ void dft32(V (&a)[32]) {
    typedef sc<V> T;
    // cos / sin of 2 pi m / 32, m = 0 .. 8 (one octant + 1; the other angles by symmetry)
    constexpr long double C32[9] = {1.0L, 0.98078528040323044912618223613424L, 0.92387953251128675612818318939679L,
                                    0.83146961230254523707878837761791L, 0.70710678118654752440084436210485L,
                                    0.55557023301960222474283081394853L, 0.38268343236508977172845998403040L,
                                    0.19509032201612826784828486847702L, 0.0L};
    V b[4][8];
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        V t[8];
#pragma unroll
        for (int q1 = 0; q1 < 8; ++q1) t[q1] = a[4 * q1 + q0];
        dft8<DIR>(t);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            const int m = q0 * k1;   // 0 .. 21: W32^m = exp(-2 pi i m / 32)
            if (m == 0) { b[q0][k1] = t[k1]; continue; }
            // cos(2 pi m/32) and sin(2 pi m/32) from the first-octant table
            const int mm = m % 32;
            const int r = mm % 8, oct = mm / 8;   // angle = oct * 90deg/... (8 steps = 90 degrees)
            const T c0 = (T)C32[r], s0 = (T)C32[8 - r];   // cos, sin of r * (2 pi / 32)
            T cs, sn;
            if (oct == 0) { cs = c0; sn = s0; }
            else if (oct == 1) { cs = -s0; sn = c0; }
            else if (oct == 2) { cs = -c0; sn = -s0; }
            else { cs = s0; sn = -c0; }
            b[q0][k1] = twmul_c<DIR>(t[k1], mk<T>(cs, -sn));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        dft4<DIR>(b[0][k1], b[1][k1], b[2][k1], b[3][k1]);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) a[k1 + 8 * k2] = b[k2][k1];
    }
}

// Radix P*Q with coprime P, Q as a P x Q two-dimensional DFT (Good-Thomas index maps): no twiddles between
// the two passes, the input / output permutations are compile-time register renamings.
//   input  i(a, b)   = (Q a + P b) mod R        output k(ka, kb) = (ka Q (Q^-1 mod P) + kb P (P^-1 mod Q)) mod R
__host__ __device__ constexpr int modinv_(int a, int m) {
    for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x;
    return 1;
}
template <int P, int Q, int DIR, typename V> __device__ __forceinline__ void dft_pfa(V (&a)[P * Q]) {
    constexpr int R = P * Q, Qi = modinv_(Q % P, P), Pi = modinv_(P % Q, Q);
    V y[P][Q];
#pragma unroll
    for (int b = 0; b < Q; ++b) {
        V t[P];
#pragma unroll
        for (int i = 0; i < P; ++i) t[i] = a[(Q * i + P * b) % R];
        dftR<P, DIR>(t);
#pragma unroll
        for (int i = 0; i < P; ++i) y[i][b] = t[i];
    }
#pragma unroll
    for (int ka = 0; ka < P; ++ka) {
        V u[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) u[b] = y[ka][b];
        dftR<Q, DIR>(u);
#pragma unroll
        for (int kb = 0; kb < Q; ++kb) a[(ka * Q * Qi + kb * P * Pi) % R] = u[kb];
    }
}

// Radix P*Q with a common factor (9 = 3 x 3, 25 = 5 x 5, 27 = 3 x 9) as Cooley-Tukey: P-point transforms, constant twiddles
// W_R^(b k1) from a compile-time table, Q-point transforms across them.  Used by the column kernels
// of fft_big.h, where a register-sized factor must leave a row length that is a multiple of 16.
template <int R> struct CtTab;
template <> struct CtTab<9> {
    static constexpr long double C[9] = {1L, 0.766044443118978013452L, 0.173648177666930358942L, -0.5L, -0.939692620785908427905L, -0.939692620785908427905L, -0.5L, 0.173648177666930358942L, 0.766044443118978013452L};
    static constexpr long double S[9] = {0L, 0.642787609686539362919L, 0.984807753012208020316L, 0.866025403784438596588L, 0.342020143325668712908L, -0.342020143325668712908L, -0.866025403784438596588L, -0.984807753012208020316L, -0.642787609686539362919L};
};
template <> struct CtTab<25> {
    static constexpr long double C[25] = {1L, 0.968583161128631076053L, 0.87630668004386358394L, 0.728968627421411552447L, 0.53582679497899665666L, 0.309016994374947451263L, 0.062790519529313373881L, -0.18738131458572462873L, -0.425779291565072659509L, -0.637423989748689745483L, -0.809016994374947451263L, -0.929776485888251458256L, -0.992114701314477875904L, -0.992114701314477875904L, -0.929776485888251458256L, -0.809016994374947451263L, -0.637423989748689745483L, -0.425779291565072659509L, -0.18738131458572462873L, 0.062790519529313373881L, 0.309016994374947451263L, 0.53582679497899665666L, 0.728968627421411552447L, 0.87630668004386358394L, 0.968583161128631076053L};
    static constexpr long double S[25] = {0L, 0.248689887164854794843L, 0.481753674101715267941L, 0.684547105928688726095L, 0.84432792550201507531L, 0.951056516295153531182L, 0.998026728428271558968L, 0.982287250728688721146L, 0.904827052466019576826L, 0.770513242775789253258L, 0.587785252292473137103L, 0.368124552684677974757L, 0.125333233564304258323L, -0.125333233564304258323L, -0.368124552684677974757L, -0.587785252292473137103L, -0.770513242775789253258L, -0.904827052466019576826L, -0.982287250728688721146L, -0.998026728428271558968L, -0.951056516295153531182L, -0.84432792550201507531L, -0.684547105928688726095L, -0.481753674101715267941L, -0.248689887164854794843L};
};
template <> struct CtTab<27> {
    static constexpr long double C[27] = {1L, 0.973044870579823806267L, 0.893632640323412275052L, 0.766044443118978013452L, 0.597158591702786178956L, 0.396079766039156844215L, 0.173648177666930358942L, -0.0581448289104758292423L, -0.286803232711090261287L, -0.5L, -0.686241637868733600492L, -0.835487811412936376421L, -0.939692620785908427905L, -0.993238357741943023171L, -0.993238357741943023171L, -0.939692620785908427905L, -0.835487811412936376421L, -0.686241637868733600492L, -0.5L, -0.286803232711090261287L, -0.0581448289104758292423L, 0.173648177666930358942L, 0.396079766039156844215L, 0.597158591702786178956L, 0.766044443118978013452L, 0.893632640323412275052L, 0.973044870579823806267L};
    static constexpr long double S[27] = {0L, 0.230615870742440165486L, 0.448799180200462166646L, 0.642787609686539362919L, 0.802123192755043734614L, 0.918216106880273996715L, 0.984807753012208020316L, 0.998308158271268175632L, 0.957989512315488900285L, 0.866025403784438596588L, 0.727373641573048734799L, 0.549508978070806008986L, 0.342020143325668712908L, 0.116092914125230234346L, -0.116092914125230234346L, -0.342020143325668712908L, -0.549508978070806008986L, -0.727373641573048734799L, -0.866025403784438596588L, -0.957989512315488900285L, -0.998308158271268175632L, -0.984807753012208020316L, -0.918216106880273996715L, -0.802123192755043734614L, -0.642787609686539362919L, -0.448799180200462166646L, -0.230615870742440165486L};
};
// input n = Q a + b (a < P, b < Q): P-point transforms over a for every b -> t_b[k1]; times W_R^(b k1); Q-point transforms
// over b for every k1 -> X[k1 + P k2]
template <int P, int Q, int DIR, typename V> __device__ __forceinline__ void dft_ct(V (&a)[P * Q]) {
    typedef sc<V> T;
    constexpr int R = P * Q;
    V y[P][Q];
#pragma unroll
    for (int b = 0; b < Q; ++b) {
        V t[P];
#pragma unroll
        for (int i = 0; i < P; ++i) t[i] = a[Q * i + b];
        dftR<P, DIR>(t);
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            const int m = (b * k1) % R;
            y[k1][b] = m == 0 ? t[k1] : twmul_c<DIR>(t[k1], mk<T>((T)CtTab<R>::C[m], (T)-CtTab<R>::S[m]));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        V u[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) u[b] = y[k1][b];
        dftR<Q, DIR>(u);
#pragma unroll
        for (int k2 = 0; k2 < Q; ++k2) a[k1 + P * k2] = u[k2];
    }
}

template <int R, int DIR, typename V> __device__ __forceinline__ void dftR(V (&a)[R]) {
    if constexpr (R == 2) dft2<DIR>(a[0], a[1]);
    else if constexpr (R == 3) dft3<DIR>(a[0], a[1], a[2]);
    else if constexpr (R == 4) dft4<DIR>(a[0], a[1], a[2], a[3]);
    else if constexpr (R == 5) dft5<DIR>(a[0], a[1], a[2], a[3], a[4]);
    else if constexpr (R == 6) dft_pfa<2, 3, DIR>(a);
    else if constexpr (R == 8) dft8<DIR>(a);
    else if constexpr (R == 9) dft_ct<3, 3, DIR>(a);
    else if constexpr (R == 10) dft_pfa<2, 5, DIR>(a);
    else if constexpr (R == 12) dft_pfa<4, 3, DIR>(a);
    else if constexpr (R == 15) dft_pfa<3, 5, DIR>(a);
    else if constexpr (R == 16) dft16<DIR>(a);
    else if constexpr (R == 20) dft_pfa<4, 5, DIR>(a);
    else if constexpr (R == 24) dft_pfa<8, 3, DIR>(a);
    else if constexpr (R == 25) dft_ct<5, 5, DIR>(a);
    else if constexpr (R == 27) dft_ct<3, 9, DIR>(a);
    else if constexpr (R == 30) dft_pfa<6, 5, DIR>(a);
    else if constexpr (R == 32) dft32<DIR>(a);
}

// value of the same register in lane ^ 1 / lane ^ 2 (DPP quad_perm [1, 0, 3, 2] / [2, 3, 0, 1]); every lane of the quad must be active
template <int CTRL> __device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_quad(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ T dpp_xor1(T v) { return dpp_quad<0xB1>(v); }
template <typename T> __device__ __forceinline__ T dpp_xor2(T v) { return dpp_quad<0x4E>(v); }

// 16-byte (float) / 32-byte (double) global access unit: 4 scalars
template <typename T> struct vec4t;
template <> struct vec4t<float> { typedef __attribute__((ext_vector_type(4))) float type; };
template <> struct vec4t<double> { typedef __attribute__((ext_vector_type(4))) double type; };
template <typename T> using vec4 = typename vec4t<T>::type;

}  // namespace pf
