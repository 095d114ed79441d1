// deform_mlp16.hip -- fused PE + Deformable_Field MLP with 16-bit operands (fp16 or bf16) and fp32
// accumulation on the gfx950 matrix cores (v_mfma_f32_32x32x16_{f16,bf16}: 16x the fp32-MFMA rate).
// BASELINE config 5 ("deformation MLP on MFMA fp16"); opt-in via Deformable_Field(precision=...),
// the default stays exact fp32 (deform_kernels.hip) because the 1e-4 RGB parity bar covers the chain.
//
// Orientation (differs from the fp32 kernel on purpose): D[row][feature] = sum_k X[row][k] W[feature][k].
//   A operand = activations from LDS: act[row][k] row-major, 16-byte XOR swizzle ((row & 15) << 4 bytes),
//               one ds_read_b128 = the lane's 8 consecutive k
//   B operand = nn.Linear weights [out][in] row-major (16-bit copy), one 16-byte global load per lane
// Both operands use the same lane -> k-slice assumption, so the contraction is correct for any
// hardware k-permutation.  A workgroup = 4 waves = 64 rows; wave w owns output features [64w, 64w+64).
// Saved for backward (training): H_l TRANSPOSED ([256][rows], so the weight-gradient GEMM reads 8
// consecutive rows per lane as one 16-byte load) and a ReLU sign bitmask (32 B per row per layer).
// fp16 backward runs the whole dZ chain scaled by a power of two picked on the device from max|dL_dout|.
//
// SPLIT mode (GP_DTYPE_F16_SPLIT, precision="fp32s"): fp32-grade results at the 16-bit matrix-core rate.  Every fp32
// operand x is carried as TWO fp16 numbers, hi = fp16(x) and lo' = fp16((x - hi) * 2^11) (22 significant bits; lo' has
// hi's magnitude, so neither half lives in the fp16 subnormal range unless |x| < 2^-14, where hi := 0 and lo' carries x),
// and a product sum becomes three MFMA chains with fp32 accumulation:
//     sum a b  =  sum ah bh  +  2^-11 (sum ah bl' + sum al' bh)          (the al' bl' term, 2^-22 relative, is dropped)
// Tiles hold [hi | lo'] side by side (LDS row = 512 halves; saved tensors = 2 nf "features" per row block), the weight
// arrays are [hi copy][lo' copy].  Positional encoding uses sincosf (as the fp32 kernels), not the hardware sin/cos.
#include "gp_common.h"
#include "deform_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) h4 lds_h4;      // (an LDS location addressed by its 32-bit byte address)
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
template <typename T> struct Vec4;
template <> struct Vec4<_Float16> { typedef h4 type; };
template <> struct Vec4<__bf16> { typedef b4 type; };
template <typename T> struct Vec8;
template <> struct Vec8<_Float16> { typedef h8 type; };
template <> struct Vec8<__bf16> { typedef b8 type; };
// fp16 backward: the dZ chain is scaled by a power of two chosen on the device from max|dL_dout| so that
// the largest entry lands near 2^12 (no overflow at 65504, small gradients stay normal); bf16 needs none.
template <typename T> struct UsesScale;
template <> struct UsesScale<_Float16> { static constexpr bool v = true; };
template <> struct UsesScale<__bf16> { static constexpr bool v = false; };
__device__ __forceinline__ float grad_scale_from(const uint32_t* absmax_bits) {
    const float mx = __uint_as_float(absmax_bits[0]);
    if (!(mx > 0.f)) return 1.f;
    float e = floorf(12.f - log2f(mx));
    e = fminf(fmaxf(e, -60.f), 60.f);
    return exp2f(e);
}
__global__ __launch_bounds__(256) void gp_absmax_kernel(const float* __restrict__ x, size_t n, uint32_t* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bits
}

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(b8 a, b8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

#define M16_ROWS 64
// Saved / intermediate 16-bit tensors are stored BLOCKED: [row block of 16][feature][16 rows], rows padded to a multiple of
// 64 with zeros.  A 32-feature x 16-row MFMA operand fragment of the weight-gradient GEMM (lane = feature, 8 consecutive
// rows per lane) is then ONE contiguous 1 KB wave load; a forward workgroup (64 rows) still writes one contiguous
// (features x 128 B) chunk.  The plain feature-major layout [feature][rows] put every lane of a fragment load in a
// different page (2 MB stride at 1M rows: TLB-bound); 64-row blocks made each load touch 64 separate cache lines.
#define T16_BLK 16
__device__ __host__ __forceinline__ size_t t16_idx(int f, long row, int nf) { return ((size_t)(row >> 4) * nf + f) * T16_BLK + (row & 15); }
// (a tensor's extent is padded to 128 rows, the tile of the 128-row workgroups: their carried stores -- one 1 KB store per k-step inside
// the product's asm statement -- are unguarded, so the last workgroup's rows beyond the 64-row padding need a place inside the same tensor;
// readers never look past the zero padding to 64 rows)
__device__ __host__ __forceinline__ size_t t16_elems(int nf, long rows) { return (size_t)((rows + 127) / 128) * nf * 128; }
#define M16_THREADS 256
#define GP_MLP16_BIG_ROWS 65536   // from here on forward / data-backward use 128-row workgroups
#define M16_W 256

// element index of (row, feature) in the swizzled [64][256] tile (16-byte granules XORed by row & 15)
template <int WS = M16_W>
__device__ __forceinline__ int a16_idx(int row, int f) { return row * WS + (f ^ ((row & 15) << 3)); }
__device__ __forceinline__ int cd_row16(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Ablation bits of the forward body (gp_debug_option(9, bits); tools/probe/mlp16_ablate.sh): 1 no saved-tensor stores, 2 no ReLU-mask
// stores, 4 no matrix products, 8 no input staging (zeros), 16 no epilogue.  Timing only: the results are garbage.
__device__ int g_m16_ablate;
static int m16_set_ablate(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_m16_ablate), &v, sizeof(int)) == hipSuccess ? 0 : 1; }

struct Mlp16Dev {
    long rows;
    int in_dim, in_pad, out_dim;       // in_pad = in_dim rounded up to 16
    int feature_dim, xyz_freq, time_freq;
    const void* w[5];                  // 16-bit copies of [256][in_pad], 3 x [256][256], [32][256] (rows >= out_dim zero), FRAGMENT-PACKED:
                                       // element (f, k) of a [F][K] matrix at (((k / 16) * (F / 32) + f / 32) * 64 + ((k / 8) & 1) * 32 + f % 32) * 8 + k % 8
    const void* wlo[5];                // split mode: the lo' copies (same shapes), else null
    const float* b[5];                 // fp32 biases
    const float* feature;
    const float* xyz;
    const float* t;
    uint32_t* range_flag;              // split forward: OR-ed with 1 when a hidden activation reached 2^15 (gp_mlp16_params.range_flag)
};

// hardware sin/cos (v_sin_f32 / v_cos_f32 take revolutions and reduce the range themselves, |x| < 256 rev):
// ~1e-5 absolute error, far below 16-bit operand precision; 4 instructions instead of ~150 for ocml sincosf
__device__ __forceinline__ void fast_sincos(float a, float* s, float* c) {
    const float rev = a * 0.15915494309189535f;
    *s = __builtin_amdgcn_sinf(rev);
    *c = __builtin_amdgcn_cosf(rev);
}

// sin and cos of a positional-encoding argument (|a| up to a few thousand: 2^9 times a coordinate) for the split-mode kernels: the
// quadrant from rint(a 2/pi), a three-constant Cody-Waite reduction (pi/2 = C1 + C2 + C3 in fp32; the first step is exact for |a| < 2^13,
// the fma keeps every product unrounded), Cephes' minimax polynomials on [-pi/4, pi/4].  |error| <= 9.2e-8 (1.5 ulp at 1; ocml's sincosf: 0.5
// ulp) in ~25 instructions instead of ~110 with branches -- the encoding was a fifth of the forward's vector instructions.
__device__ __forceinline__ void sincos_pe(float a, float* s, float* c) {
    const float n = rintf(a * 0.6366197466850281f);
    float r = fmaf(-n, 1.5707963705062866f, a);
    r = fmaf(-n, -4.371138828673793e-08f, r);
    r = fmaf(-n, -1.7151245100058819e-15f, r);
    const float z = r * r;
    const float sr = fmaf(r * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), r);
    const float cr = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), fmaf(-0.5f, z, 1.f));
    const int q = (int)n;
    const float ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = __uint_as_float(__float_as_uint(ss) ^ ((uint32_t)(q & 2) << 30));
    *c = __uint_as_float(__float_as_uint(cc) ^ ((uint32_t)((q + 1) & 2) << 30));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// four floats -> four 16-bit values with the packed converts (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: one instruction per pair)
__device__ __forceinline__ h4 pack4(float a, float b, float c, float d, _Float16) {
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const hh2 lo = __builtin_convertvector((f32x2){a, b}, hh2), hi = __builtin_convertvector((f32x2){c, d}, hh2);
    return (h4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ b4 pack4(float a, float b, float c, float d, __bf16) {
    typedef __bf16 bb2 __attribute__((ext_vector_type(2)));
    const bb2 lo = __builtin_convertvector((f32x2){a, b}, bb2), hi = __builtin_convertvector((f32x2){c, d}, bb2);
    return (b4){lo[0], lo[1], hi[0], hi[1]};
}

// ReLU sign masks (forward -> data backward).  The two kernels give a lane the same values -- row = lane & 31 (+ 32 rt), half = lane >> 5,
// 16 features per 32-feature tile: 8 g + 4 half + e -- so a lane keeps ITS 32 signs of a row (two tiles) in one private dword:
// masks[layer][wave][row][half], bit 31 - (16 nt + 4 g + e) (the values are pushed in loop order, newest at bit 0).
__device__ __forceinline__ size_t relu_mask_idx(int l, int wave, long row, int half, long rows) { return (((size_t)l * 4 + wave) * rows + row) * 2 + half; }
__device__ __forceinline__ constexpr int relu_mask_pos(int nt, int g, int e) { return 31 - (16 * nt + 4 * g + e); }
// max(z, 0) and its sign pushed into `m`: the INTEGER maximum maps every non-positive float (-0 included) to +0 exactly, after
// which "v > 0" is "the bits are not zero" = the sign of (0 - bits): three instructions per value (v_max_i32, v_sub_u32, v_alignbit_b32)
__device__ __forceinline__ float relu_push(float z, uint32_t& m) {
    const int b = max(__float_as_int(z), 0);
    m = __builtin_amdgcn_alignbit(m, 0u - (uint32_t)b, 31);
    return __int_as_float(b);
}

template <typename T> struct Vec2;
template <> struct Vec2<_Float16> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<__bf16> { typedef __bf16 type __attribute__((ext_vector_type(2))); };

// Layer-0 input tile [ROWS][in_pad] = [feature | PE(xyz) | PE(t) | 0] in 16 bits.  Fast path (feature_dim % 4 == 0, the
// reference's 32): float4 feature loads -> 8-byte LDS stores, one thread per (row, coordinate) computing all its
// frequencies with (sin, cos) pairs stored as one dword, no run-time divisions (this stage was ~15 % of the forward).
template <typename T, int ROWS>
__device__ __forceinline__ void build_input16(T* buf, const Mlp16Dev& p, long row0, int tid) {
    typedef typename Vec4<T>::type V4;
    typedef typename Vec2<T>::type V2;
    const int fd = p.feature_dim, xf = p.xyz_freq, tf = p.time_freq;
    const float tv = tf > 0 ? p.t[0] : 0.f;
    if ((fd & 3) == 0 && fd <= 64 && ((uintptr_t)p.feature & 15) == 0) {
        // every global load of the tile before the first LDS store (as rolled loops: ROWS / 32 + 2 dependent trips to memory per workgroup)
        const int q = fd >> 2, nq = ROWS * q;
        constexpr int FU = ROWS * 64 / 4 / M16_THREADS, XU = (3 * ROWS + M16_THREADS - 1) / M16_THREADS;
        float4 fv[FU];
        float xv[XU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int e = tid + M16_THREADS * u;
            const int jj = e < nq ? e / q : 0, f4 = e < nq ? e - jj * q : 0;
            const long row = row0 + jj;
            const bool ok = e < nq && row < p.rows;
            fv[u] = *(const float4*)(p.feature + (ok ? row * fd + 4 * f4 : 0));
            if (!ok) fv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + M16_THREADS * u;
            const long row = row0 + e / 3;
            const bool ok = e < 3 * ROWS && row < p.rows;
            xv[u] = p.xyz[ok ? row * 3 + e % 3 : 0];
            if (!ok) xv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int e = tid + M16_THREADS * u;
            if (e < nq) {
                const int jj = e / q, f4 = e - jj * q;
                V4 pk;
                pk[0] = (T)fv[u].x; pk[1] = (T)fv[u].y; pk[2] = (T)fv[u].z; pk[3] = (T)fv[u].w;
                *(V4*)&buf[a16_idx(jj, 4 * f4)] = pk;
            }
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + M16_THREADS * u;
            if (e < 3 * ROWS) {
                const int jj = e / 3, c = e - 3 * jj;
                const bool ok = row0 + jj < p.rows;
                for (int fr = 0; fr < xf; ++fr) {
                    float sv, cv;
                    fast_sincos(xv[u] * (float)(1u << fr), &sv, &cv);
                    V2 pk;
                    pk[0] = (T)(ok ? sv : 0.f); pk[1] = (T)(ok ? cv : 0.f);
                    *(V2*)&buf[a16_idx(jj, fd + 2 * (c * xf + fr))] = pk;
                }
            }
        }
        for (int e = tid; e < tf * ROWS; e += M16_THREADS) {
            const int jj = e % ROWS, fr = e / ROWS;
            float sv, cv;
            fast_sincos(tv * (float)(1u << fr), &sv, &cv);
            const bool ok = row0 + jj < p.rows;
            V2 pk;
            pk[0] = (T)(ok ? sv : 0.f); pk[1] = (T)(ok ? cv : 0.f);
            *(V2*)&buf[a16_idx(jj, fd + 6 * xf + 2 * fr)] = pk;
        }
    } else {
        for (int e = tid; e < fd * ROWS; e += M16_THREADS) {
            const int jj = e / fd, f = e - jj * fd;
            const long row = row0 + jj;
            buf[a16_idx(jj, f)] = (T)(row < p.rows ? p.feature[row * fd + f] : 0.f);
        }
        for (int e = tid; e < 3 * xf * ROWS; e += M16_THREADS) {
            const int jj = e % ROWS, cf = e / ROWS;
            const int c = cf / xf, fr = cf - c * xf;
            const long row = row0 + jj;
            float sv = 0.f, cv = 0.f;
            if (row < p.rows) fast_sincos(p.xyz[row * 3 + c] * (float)(1u << fr), &sv, &cv);
            const int f = fd + 2 * cf;
            buf[a16_idx(jj, f)] = (T)sv;
            buf[a16_idx(jj, f + 1)] = (T)cv;
        }
        for (int e = tid; e < tf * ROWS; e += M16_THREADS) {
            const int jj = e % ROWS, fr = e / ROWS;
            float sv, cv;
            fast_sincos(tv * (float)(1u << fr), &sv, &cv);
            const bool ok = row0 + jj < p.rows;
            const int f = fd + 6 * xf + 2 * fr;
            buf[a16_idx(jj, f)] = (T)(ok ? sv : 0.f);
            buf[a16_idx(jj, f + 1)] = (T)(ok ? cv : 0.f);
        }
    }
    for (int e = tid; e < (p.in_pad - p.in_dim) * ROWS; e += M16_THREADS) {
        const int jj = e % ROWS, f = p.in_dim + e / ROWS;
        buf[a16_idx(jj, f)] = (T)0.f;
    }
}
// ---- split mode helpers --------------------------------------------------------------------------------------------
#define SP_W 512                 // halves per LDS row: hi at column f, lo' at column 256 + f
#define SP_LO 2048.f
#define SP_LO_INV 4.8828125e-4f
// x -> (hi, lo'): saturated to the fp16 range, hi flushed to zero below the fp16 normal range (lo' then carries x * 2^11)
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = (_Float16)(fabsf(c) < 6.103515625e-05f ? 0.f : c);
    lo = (_Float16)((c - (float)hi) * SP_LO);
}
__device__ __forceinline__ void split4(const float (&v)[4], h4& hi, h4& lo) {
    float c[4], r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
        r[e] = fabsf(c[e]) < 6.103515625e-05f ? 0.f : c[e];
    }
    hi = pack4(r[0], r[1], r[2], r[3], _Float16());
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (c[e] - (float)hi[e]) * SP_LO;
    lo = pack4(r[0], r[1], r[2], r[3], _Float16());
}
// Layer-0 input tile in split form, [ROWS][SP_W]; the encoding is the fp32 kernels' (sincosf on x * 2^fr)
template <int ROWS>
__device__ __forceinline__ void build_input16s(_Float16* buf, const Mlp16Dev& p, long row0, int tid) {
    const int fd = p.feature_dim, xf = p.xyz_freq, tf = p.time_freq;
    const float tv = tf > 0 ? p.t[0] : 0.f;
    auto put = [&](int jj, int f, float v) { split1(v, buf[a16_idx<SP_W>(jj, f)], buf[a16_idx<SP_W>(jj, 256 + f)]); };
    constexpr int FU = ROWS * 64 / 4 / M16_THREADS;         // float4 feature loads per thread at feature_dim = 64 (4 for 64 rows)
    if ((fd & 3) == 0 && fd <= 64 && ((uintptr_t)p.feature & 15) == 0) {
        // EVERY global load of the tile is issued before the first LDS store: as `for (e = tid; ...) put(.., p.feature[..])` and
        // `sincos_pe(p.xyz[..] ..)` loops the tile cost 8 + 8 dependent trips to memory per workgroup (a rolled loop with a conditional
        // load waits for each one) -- a quarter of a 64-row workgroup's lifetime at two workgroups per CU.  Same values, same
        // expressions per element.
        const int q = fd >> 2, nq = ROWS * q;
        float4 fv[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int e = tid + M16_THREADS * u;
            const int jj = e < nq ? e / q : 0, f4 = e < nq ? e - jj * q : 0;
            const long row = row0 + jj;
            const bool ok = e < nq && row < p.rows;
            fv[u] = *(const float4*)(p.feature + (ok ? row * fd + 4 * f4 : 0));
            if (!ok) fv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        static_assert(3 * ROWS <= 2 * M16_THREADS, "two coordinates per thread at most");
        float xv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + M16_THREADS * u;
            const long row = row0 + e / 3;
            const bool ok = e < 3 * ROWS && row < p.rows;
            xv[u] = p.xyz[ok ? row * 3 + e % 3 : 0];
            if (!ok) xv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int e = tid + M16_THREADS * u;
            if (e < nq) {
                const int jj = e / q, f4 = e - jj * q;
                const float v[4] = {fv[u].x, fv[u].y, fv[u].z, fv[u].w};
                h4 hi, lo;
                split4(v, hi, lo);
                *(h4*)&buf[a16_idx<SP_W>(jj, 4 * f4)] = hi;
                *(h4*)&buf[a16_idx<SP_W>(jj, 256 + 4 * f4)] = lo;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + M16_THREADS * u;
            if (e < 3 * ROWS) {
                const int jj = e / 3, c = e - 3 * jj;
                const bool ok = row0 + jj < p.rows;
                for (int fr = 0; fr < xf; ++fr) {
                    float sv = 0.f, cv = 0.f;
                    if (ok) sincos_pe(xv[u] * (float)(1u << fr), &sv, &cv);
                    put(jj, fd + 2 * (c * xf + fr), sv);
                    put(jj, fd + 2 * (c * xf + fr) + 1, cv);
                }
            }
        }
    } else {
        for (int e = tid; e < fd * ROWS; e += M16_THREADS) {
            const int jj = e / fd, f = e - jj * fd;
            const long row = row0 + jj;
            put(jj, f, row < p.rows ? p.feature[row * fd + f] : 0.f);
        }
        for (int e = tid; e < 3 * xf * ROWS; e += M16_THREADS) {
            const int jj = e % ROWS, cf = e / ROWS;
            const int c = cf / xf, fr = cf - c * xf;
            const long row = row0 + jj;
            float sv = 0.f, cv = 0.f;
            if (row < p.rows) sincos_pe(p.xyz[row * 3 + c] * (float)(1u << fr), &sv, &cv);
            put(jj, fd + 2 * cf, sv);
            put(jj, fd + 2 * cf + 1, cv);
        }
    }
    for (int e = tid; e < tf * ROWS; e += M16_THREADS) {
        const int jj = e % ROWS, fr = e / ROWS;
        float sv, cv;
        sincos_pe(tv * (float)(1u << fr), &sv, &cv);
        const bool ok = row0 + jj < p.rows;
        put(jj, fd + 6 * xf + 2 * fr, ok ? sv : 0.f);
        put(jj, fd + 6 * xf + 2 * fr + 1, ok ? cv : 0.f);
    }
    for (int e = tid; e < (p.in_pad - p.in_dim) * ROWS; e += M16_THREADS) put(e % ROWS, p.in_dim + e / ROWS, 0.f);
}

// Copy a [ROWS][nf] activation tile (row-major, swizzled, in LDS) to the blocked saved layout [16-row block][nf][16 rows].
// A lane owns 8 rows x 8 features: eight 16-byte LDS reads (one row each, conflict-free: 32 lanes cover one 512-B row),
// an 8x8 transpose of 16-bit elements in registers (32 v_perm_b32), eight 16-byte global stores (8 rows of one feature).
// The element-wise version (one 2-byte LDS read + one 2-byte store per element) was 55 % of the forward kernel.
// `nfs` = features per row block in the destination (= nf, or 2 nf in split mode where `buf` / `blk` are pre-offset to the
// half being stored); WS = LDS row length.
template <typename T, int ROWS, int WS = M16_W>
__device__ __forceinline__ void store_tile_T(const T* buf, T* __restrict__ blk, int nf, int nfs, long row0, long rows, long rows_pad,
                                             int wave, int lane) {
    typedef typename Vec8<T>::type V8;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int fg = lane & 31, h8 = lane >> 5;
#pragma unroll
    for (int sub = wave; sub < ROWS / T16_BLK; sub += M16_THREADS / 64) {
        if (row0 + sub * T16_BLK >= rows_pad) break;            // uniform per wave
        if (fg * 8 < nf) {
            const int r0 = sub * T16_BLK + 8 * h8;
            u4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(u4, *(const V8*)&buf[a16_idx<WS>(r0 + i, fg * 8)]);
            if (row0 + sub * T16_BLK + T16_BLK > rows) {        // zero padding rows (last block only)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (row0 + r0 + i >= rows) v[i] = u4{0u, 0u, 0u, 0u};
            }
            T* o = blk + ((size_t)sub * nfs + fg * 8) * T16_BLK + 8 * h8;
#pragma unroll
            for (int c = 0; c < 8; ++c) {                       // feature fg*8 + c: rows r0 .. r0+7
                u4 t;
#pragma unroll
                for (int d = 0; d < 4; ++d)                     // dword d = rows (2d, 2d+1), 16-bit column c of each
                    t[d] = __builtin_amdgcn_perm(v[2 * d + 1][c >> 1], v[2 * d][c >> 1], (c & 1) ? 0x07060302u : 0x05040100u);
                *(u4*)(o + c * T16_BLK) = t;
            }
        }
    }
}

// acc[rt][nt] += X[64 x K] . W[feature tile][K]^T for this wave's two feature tiles.
// The weight fragments (straight from L2) are software-pipelined in groups of four k-steps: the 8 loads
// of group g+1 are in flight while the 16 MFMAs (512 matrix-pipe cycles) of group g execute.
// SWAPPED = true swaps the MFMA operands (weights as A, activations as B): the accumulator then holds C[feature][row],
// i.e. lane = row, registers = 16 of the tile's 32 features in runs of FOUR CONSECUTIVE features -- an epilogue writes
// 8 bytes per run (4 stores per tile instead of 16 two-byte ones) and builds ReLU masks from its own registers.
template <typename T, bool SWAPPED = false, int RT = 2>
__device__ __forceinline__ void gemm16(const T* cur, const T* __restrict__ W, int ldk, int K, int n_feat, int wave, int lane,
                                       f32x16 (&acc)[RT][2]) {
    typedef typename Vec8<T>::type V8;
    const int half = lane >> 5, j = lane & 31;
    // fragment-packed weights [k-step][feature tile of 32][lane][8]: a wave's load is 1 KB contiguous (row-major weights made
    // every lane of a load touch its own cache line: the address unit, not the matrix pipe, set the pace)
    const int n_tiles = (n_feat + 31) / 32;
    const bool ok0 = 2 * wave < n_tiles, ok1 = 2 * wave + 1 < n_tiles;
    const size_t kst = (size_t)n_tiles * 512;
    const T* w0 = W + ((size_t)(2 * wave) * 64 + lane) * 8;
    const T* w1 = w0 + 512;
    const int nks = K / 16;
    V8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (T)0.f;
    // A-fragment base for row j; the swizzled column of k-step ks is ((ks*16 + 8*half) ^ ((j & 15) << 3)):
    // bits 3..6 only, so it is computed with one XOR on a per-lane constant
    const T* arow = cur + j * M16_W;
    const int swz = ((j & 15) << 3) ^ (8 * half);
    constexpr int G = RT > 2 ? 2 : 4;       // k-steps per prefetch group: 8 RT MFMAs (256 RT matrix-pipe cycles) cover one group's loads
    V8 bn[G][2];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        bn[u][0] = (ok0 && u < nks) ? *(const V8*)(w0 + u * kst) : zero;
        bn[u][1] = (ok1 && u < nks) ? *(const V8*)(w1 + u * kst) : zero;
    }
    for (int ks = 0; ks < nks; ks += G) {
        V8 bc[G][2];
#pragma unroll
        for (int u = 0; u < G; ++u) { bc[u][0] = bn[u][0]; bc[u][1] = bn[u][1]; }
#pragma unroll
        for (int u = 0; u < G; ++u) {   // prefetch the next group
            const int kn = ks + G + u;
            bn[u][0] = (ok0 && kn < nks) ? *(const V8*)(w0 + kn * kst) : zero;
            bn[u][1] = (ok1 && kn < nks) ? *(const V8*)(w1 + kn * kst) : zero;
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (ks + u < nks) {
                const int col = ((ks + u) * 16) ^ swz;
                V8 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = *(const V8*)(arow + rt * 32 * M16_W + col);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (SWAPPED) {
                        acc[rt][0] = mfma16(bc[u][0], a[rt], acc[rt][0]);
                        acc[rt][1] = mfma16(bc[u][1], a[rt], acc[rt][1]);
                    } else {
                        acc[rt][0] = mfma16(a[rt], bc[u][0], acc[rt][0]);
                        acc[rt][1] = mfma16(a[rt], bc[u][1], acc[rt][1]);
                    }
                }
            }
        }
    }
}

// Split-mode product: am += Xh . Wh^T,  ax += Xh . Wl'^T + Xl' . Wh^T  (result = am + 2^-11 ax).  One pass over K: every
// fragment (two LDS reads per row tile, four L2 loads) feeds three MFMAs per (row tile, feature tile).
template <bool SWAPPED, int RT>
__device__ __forceinline__ void gemm16s(const _Float16* cur, const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl, int ldk,
                                        int K, int n_feat, int wave, int lane, f32x16 (&am)[RT][2], f32x16 (&ax)[RT][2]) {
    const int half = lane >> 5, j = lane & 31;
    const int n_tiles = (n_feat + 31) / 32;                  // fragment-packed weights, see gemm16
    const bool ok0 = 2 * wave < n_tiles, ok1 = 2 * wave + 1 < n_tiles;
    const size_t kst = (size_t)n_tiles * 512;
    const size_t o0 = ((size_t)(2 * wave) * 64 + lane) * 8, o1 = o0 + 512;
    const int nks = K / 16;
    h8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.f;
    const _Float16* arow = cur + j * SP_W;
    const int swz = ((j & 15) << 3) ^ (8 * half);
    h8 bn[4];                            // next k-step: Wh tile 0, Wh tile 1, Wl' tile 0, Wl' tile 1
    bn[0] = (ok0 && nks > 0) ? *(const h8*)(Wh + o0) : zero;
    bn[1] = (ok1 && nks > 0) ? *(const h8*)(Wh + o1) : zero;
    bn[2] = (ok0 && nks > 0) ? *(const h8*)(Wl + o0) : zero;
    bn[3] = (ok1 && nks > 0) ? *(const h8*)(Wl + o1) : zero;
    for (int ks = 0; ks < nks; ++ks) {
        h8 bc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bc[u] = bn[u];
        const int kn = ks + 1;
        bn[0] = (ok0 && kn < nks) ? *(const h8*)(Wh + o0 + kn * kst) : zero;
        bn[1] = (ok1 && kn < nks) ? *(const h8*)(Wh + o1 + kn * kst) : zero;
        bn[2] = (ok0 && kn < nks) ? *(const h8*)(Wl + o0 + kn * kst) : zero;
        bn[3] = (ok1 && kn < nks) ? *(const h8*)(Wl + o1 + kn * kst) : zero;
        const int col = (ks * 16) ^ swz;
        h8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ah[rt] = *(const h8*)(arow + rt * 32 * SP_W + col);
            al[rt] = *(const h8*)(arow + rt * 32 * SP_W + 256 + col);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (SWAPPED) {
                am[rt][0] = mfma16(bc[0], ah[rt], am[rt][0]);
                am[rt][1] = mfma16(bc[1], ah[rt], am[rt][1]);
                ax[rt][0] = mfma16(bc[2], ah[rt], ax[rt][0]);
                ax[rt][1] = mfma16(bc[3], ah[rt], ax[rt][1]);
            } else {
                am[rt][0] = mfma16(ah[rt], bc[0], am[rt][0]);
                am[rt][1] = mfma16(ah[rt], bc[1], am[rt][1]);
                ax[rt][0] = mfma16(ah[rt], bc[2], ax[rt][0]);
                ax[rt][1] = mfma16(ah[rt], bc[3], ax[rt][1]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (SWAPPED) {
                ax[rt][0] = mfma16(bc[0], al[rt], ax[rt][0]);
                ax[rt][1] = mfma16(bc[1], al[rt], ax[rt][1]);
            } else {
                ax[rt][0] = mfma16(al[rt], bc[0], ax[rt][0]);
                ax[rt][1] = mfma16(al[rt], bc[1], ax[rt][1]);
            }
        }
    }
}

#include "deform_mlp16_kloop.inc"
// The split product of a 64-row workgroup as ONE hand-scheduled asm statement per layer (tools/gen_mlp16_kloop.py; same arithmetic and
// the same MFMA order per accumulator as gemm16s<true, 2>: bit-identical results): weight fragments three k-steps ahead, activation
// fragments one k-step ahead, counted waits, and -- in the training / backward forms -- the stores of the tile being read: the saved
// tensor of the previous layer, `st` = this workgroup's [4 row blocks][512 features][16 rows] block, one 1 KB store per k-step (wave w
// owns row block w; k-step g = features 32 g .. 32 g + 31 of [hi | lo']), so that the stores leave the CU at the rate of the product
// instead of as a 64 KB burst between two products.  `sel` picks the statement's second path: the layer with its own depth
// (forward: layer 0, K = 112; data backward: the output layer, K = 16), which carries nothing.
struct Kloop16sAddr {
    uint32_t abase, voff, sbt, vost;
};
__device__ __forceinline__ Kloop16sAddr kloop16s_addr(const _Float16* cur, int wave, int lane) {
    Kloop16sAddr a;
    const int half = lane >> 5, j = lane & 31;
    const uint32_t base = (uint32_t)(uintptr_t)cur;
    a.abase = base + j * (SP_W * 2) + 2 * ((((j & 15) << 3)) ^ (8 * half));
    a.voff = lane * 16;
    // the carried store (one 16-row block x 32 features per k-step g, wave w = row block w): 16-lane group G covers features
    // 32 g + 16 (G & 1) + 0..15 of rows 8 (G >> 1) + 0..7; as a SOURCE of the transpose read lane 4 jj + q of the group supplies the
    // address of features 4 q .. 4 q + 3 of row jj (+ 4 in the second pass); k-step and pass enter as one XOR with 64 (g ^ pass)
    // (+ 4096 for the pass), see tools/gen_mlp16_kloop.py
    const int G = lane >> 4, fb = G & 1, h = G >> 1, jj = (lane >> 2) & 3, q = lane & 3;
    a.sbt = base + (16 * wave + 8 * h + jj) * (SP_W * 2) + 2 * (((4 * q + 16 * fb) ^ (jj << 3)) + (h << 6));
    a.vost = (16 * fb + (lane & 15)) * 32 + 16 * h;
    return a;
}
template <typename P>
__device__ __forceinline__ P* uniform_ptr(P* p) {       // a wave-uniform pointer, in scalar registers for sure (the "s" operands below)
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (P*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
#define M16S_ACC_OUT(am, ax)                                                                                                   \
    [am10] "=&v"(am[1][0]), [am11] "=&v"(am[1][1]), [ax00] "=&v"(ax[0][0]), [ax01] "=&v"(ax[0][1]), [ax10] "=&v"(ax[1][0]),       \
        [ax11] "=&v"(ax[1][1])
// forward: am[0][*] hold the biases on entry (the statement never initialises an accumulator with moves), everything else is output only
template <bool TRAIN>
__device__ __forceinline__ void kloop16s_fwd(const Kloop16sAddr& a, const _Float16* wh, const _Float16* wl, _Float16* st, int first,
                                             f32x16 (&am)[2][2], f32x16 (&ax)[2][2]) {
    wh = uniform_ptr(wh); wl = uniform_ptr(wl);
    first = __builtin_amdgcn_readfirstlane(first);
    if constexpr (TRAIN) {
        st = uniform_ptr(st);
        asm volatile(M16S_FWD_TRAIN_ASM
                     : [am00] "+v"(am[0][0]), [am01] "+v"(am[0][1]), M16S_ACC_OUT(am, ax)
                     : [abase] "v"(a.abase), [voff] "v"(a.voff), [swh] "s"(wh), [swl] "s"(wl), [sel] "s"(first), [sst] "s"(st), [sbt] "v"(a.sbt),
                       [vost] "v"(a.vost)
                     : "memory", "scc", M16S_KLOOP_CLOBBERS);
    } else {
        asm volatile(M16S_FWD_INFER_ASM
                     : [am00] "+v"(am[0][0]), [am01] "+v"(am[0][1]), M16S_ACC_OUT(am, ax)
                     : [abase] "v"(a.abase), [voff] "v"(a.voff), [swh] "s"(wh), [swl] "s"(wl), [sel] "s"(first)
                     : "memory", "scc", M16S_KLOOP_CLOBBERS);
    }
}
// data backward: nothing is read on entry
__device__ __forceinline__ void kloop16s_bwd(const Kloop16sAddr& a, const _Float16* wh, const _Float16* wl, _Float16* st, int last,
                                             f32x16 (&am)[2][2], f32x16 (&ax)[2][2]) {
    wh = uniform_ptr(wh); wl = uniform_ptr(wl); st = uniform_ptr(st);
    last = __builtin_amdgcn_readfirstlane(last);
    asm volatile(M16S_BWD_DATA_ASM
                 : [am00] "=&v"(am[0][0]), [am01] "=&v"(am[0][1]), M16S_ACC_OUT(am, ax)
                 : [abase] "v"(a.abase), [voff] "v"(a.voff), [swh] "s"(wh), [swl] "s"(wl), [sel] "s"(last), [sst] "s"(st), [sbt] "v"(a.sbt),
                   [vost] "v"(a.vost)
                 : "memory", "scc", M16S_KLOOP_CLOBBERS);
}

// The plain 16-bit statements (fp16 / bf16 operands, 128-row workgroups: 4 row tiles x 2 feature tiles per wave; tools/gen_mlp16_kloop.py
// gen_plain): same structure -- two paths, counted waits, weight fragments five k-steps ahead, the carried store (wave w owns row blocks
// 2 w and 2 w + 1 of the tile, k-step g = features 32 (g & 7) .. + 31 of block 2 w + (g >> 3)).
__device__ __forceinline__ Kloop16sAddr kloop16p_addr(const void* cur, int wave, int lane) {
    Kloop16sAddr a;
    const int half = lane >> 5, j = lane & 31;
    const uint32_t base = (uint32_t)(uintptr_t)cur;
    a.abase = base + j * (M16_W * 2) + 2 * ((((j & 15) << 3)) ^ (8 * half));
    a.voff = lane * 16;
    const int G = lane >> 4, fb = G & 1, h = G >> 1, jj = (lane >> 2) & 3, q = lane & 3;
    a.sbt = base + (32 * wave + 8 * h + jj) * (M16_W * 2) + 2 * (((4 * q + 16 * fb) ^ (jj << 3)) + (h << 6));
    a.vost = (16 * fb + (lane & 15)) * 32 + 16 * h;
    return a;
}
#define M16P_ACC_OUT(c)                                                                                                              \
    [c10] "=&v"(c[1][0]), [c11] "=&v"(c[1][1]), [c20] "=&v"(c[2][0]), [c21] "=&v"(c[2][1]), [c30] "=&v"(c[3][0]), [c31] "=&v"(c[3][1])
#define M16P_IN(a) [abase] "v"(a.abase), [voff] "v"(a.voff), [swh] "s"(w), [sel] "s"(sel)
#define M16P_IN_CARRY(a) M16P_IN(a), [sst] "s"(st), [sbt] "v"(a.sbt), [vost] "v"(a.vost)
template <typename T, bool TRAIN>
__device__ __forceinline__ void kloop16p_fwd(const Kloop16sAddr& a, const T* w, T* st, int sel, f32x16 (&c)[4][2]) {
    w = uniform_ptr(w);
    sel = __builtin_amdgcn_readfirstlane(sel);
    constexpr bool F16 = sizeof(T) == 2 && UsesScale<T>::v;
    if constexpr (TRAIN) {
        st = uniform_ptr(st);
        if constexpr (F16) asm volatile(M16P_F16_FWD_TRAIN_ASM : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN_CARRY(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
        else asm volatile(M16P_BF16_FWD_TRAIN_ASM : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN_CARRY(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
    } else {
        if constexpr (F16) asm volatile(M16P_F16_FWD_INFER_ASM : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
        else asm volatile(M16P_BF16_FWD_INFER_ASM : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
    }
}
template <typename T>
__device__ __forceinline__ void kloop16p_bwd(const Kloop16sAddr& a, const T* w, T* st, int sel, f32x16 (&c)[4][2]) {
    w = uniform_ptr(w); st = uniform_ptr(st);
    sel = __builtin_amdgcn_readfirstlane(sel);
    constexpr bool F16 = sizeof(T) == 2 && UsesScale<T>::v;
    if constexpr (F16) asm volatile(M16P_F16_BWD_DATA_ASM : [c00] "=&v"(c[0][0]), [c01] "=&v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN_CARRY(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
    else asm volatile(M16P_BF16_BWD_DATA_ASM : [c00] "=&v"(c[0][0]), [c01] "=&v"(c[0][1]), M16P_ACC_OUT(c) : M16P_IN_CARRY(a) : "memory", "scc", M16P_KLOOP_CLOBBERS);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// The output layer of the forward: W4 padded to [32][256]; the 4 waves split K, reduce through LDS (fp32).  `cur` = the last hidden
// activations (LDS), `nxt` = scratch for the reduction (the same tile when the activations are updated in place).
template <typename T, int RT, bool SP>
__device__ __forceinline__ void mlp16_output_layer(const Mlp16Dev& p, float* __restrict__ out, const T* cur, T* nxt, long row0, int tid) {
    constexpr int ROWS = 32 * RT;
    constexpr int WS = SP ? SP_W : M16_W;
    constexpr bool INPLACE = RT > 2 || SP;
    typedef typename Vec8<T>::type V8;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    {   // output layer: W4 padded to [32][256]; the 4 waves split K, reduce through LDS (fp32)
        f32x16 acc[RT];
        f32x16 ax[SP ? RT : 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[rt][r] = 0.f; if (SP) ax[rt][r] = 0.f; }
        const T* w4 = (const T*)p.w[4] + lane * 8;              // fragment-packed, one feature tile: [k-step][lane][8]
        const T* w4l = SP ? (const T*)p.wlo[4] + lane * 8 : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = 4 * wave + u;
            const V8 b = *(const V8*)(w4 + ks * 512);
            if constexpr (SP) {
                const V8 bl = *(const V8*)(w4l + ks * 512);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const V8 ah = *(const V8*)(cur + a16_idx<WS>(32 * rt + j, ks * 16 + 8 * half));
                    const V8 al = *(const V8*)(cur + a16_idx<WS>(32 * rt + j, 256 + ks * 16 + 8 * half));
                    acc[rt] = mfma16(ah, b, acc[rt]);
                    ax[rt] = mfma16(ah, bl, ax[rt]);
                    ax[rt] = mfma16(al, b, ax[rt]);
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16(*(const V8*)(cur + a16_idx(32 * rt + j, ks * 16 + 8 * half)), b, acc[rt]);
            }
        }
        if (INPLACE) __syncthreads();
        float* red = (float*)nxt;   // [4 waves][ROWS][8]
        if (j < 8) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[rt][r];
                    if constexpr (SP) v = fmaf(ax[rt][r], SP_LO_INV, v);
                    red[(wave * ROWS + rt * 32 + cd_row16(r, half)) * 8 + j] = v;
                }
        }
        __syncthreads();
        for (int e = tid; e < ROWS * 8; e += M16_THREADS) {
            const int r = e / 8, f = e % 8;
            if (f < p.out_dim && row0 + r < p.rows) {
                float v = p.b[4][f];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[(w * ROWS + r) * 8 + f];
                out[(row0 + r) * p.out_dim + f] = v;
            }
        }
    }
}

// RT = 32-row tiles per wave.  RT = 2: 64 rows per workgroup, double-buffered activations.  RT = 4 (large row counts): 128
// rows per workgroup, so every weight fragment fetched from L2 feeds four MFMAs instead of two; the activations are
// updated IN PLACE (one 64 KB tile, two workgroups per CU) behind one extra barrier per layer.
template <typename T, int RT, bool SP = false>
__device__ __forceinline__ void mlp16_fwd_body(Mlp16Dev p, float* __restrict__ out, T* __restrict__ saved_xT /*[in_pad][rows]*/,
                                               T* __restrict__ saved_hT /*[4][256][rows]*/,
                                               uint32_t* __restrict__ masks /*[4][rows][8]*/) {
    constexpr int ROWS = 32 * RT;
    constexpr int WS = SP ? SP_W : M16_W;
    constexpr int NS = SP ? 2 : 1;                              // saved "features" per real feature
    constexpr bool INPLACE = RT > 2 || SP;
    __shared__ T smem[INPLACE ? 1 : 2][ROWS * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;                 // the saved tensors are allocated (and zero-padded) to 64 rows
    T* cur = smem[0];
    T* nxt = smem[INPLACE ? 0 : 1];
    const int ablate = g_m16_ablate;
    if (ablate & 8) { for (int e = tid; e < ROWS * WS; e += M16_THREADS) cur[e] = (T)0.f; }
    else if constexpr (SP) build_input16s<ROWS>(cur, p, row0, tid);
    else build_input16<T, ROWS>(cur, p, row0, tid);
    __syncthreads();
    auto store_T = [&](const T* buf, T* dst, int nf, int wave, int lane) {      // this workgroup's block: NS x nf x ROWS contiguous elements
        T* blk = dst + (size_t)blockIdx.x * NS * nf * ROWS;
        store_tile_T<T, ROWS, WS>(buf, blk, nf, NS * nf, row0, p.rows, rows_pad, wave, lane);
        if constexpr (SP) store_tile_T<T, ROWS, WS>(buf + 256, blk + (size_t)nf * T16_BLK, nf, NS * nf, row0, p.rows, rows_pad, wave, lane);
    };
    if (ablate & 1) { saved_xT = nullptr; saved_hT = nullptr; }
    if (ablate & 2) masks = nullptr;
    if (saved_xT) store_T(cur, saved_xT, p.in_pad, wave, lane);
    typedef typename Vec4<T>::type V4;
    float amax = 0.f;                   // split mode: the largest hidden activation this lane carried into (hi, lo') form
    for (int l = 0; l < 4; ++l) {
        const int K = l == 0 ? p.in_pad : M16_W;
        f32x16 acc[RT][2];
        f32x16 ax[SP ? RT : 1][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // this lane's features of tile nt: runs of four at (2 wave + nt) * 32 + 8 g + 4 half
                const float4 bv = *(const float4*)(p.b[l] + (2 * wave + nt) * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][nt][4 * g + 0] = bv.x; acc[rt][nt][4 * g + 1] = bv.y; acc[rt][nt][4 * g + 2] = bv.z; acc[rt][nt][4 * g + 3] = bv.w;
                }
            }
        if constexpr (SP) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ax[rt][nt][r] = 0.f;
            if (!(ablate & 4)) gemm16s<true, RT>(cur, (const _Float16*)p.w[l], (const _Float16*)p.wlo[l], K, K, M16_W, wave, lane, acc, ax);
        } else {
            if (!(ablate & 4)) gemm16<T, true, RT>(cur, (const T*)p.w[l], K, K, M16_W, wave, lane, acc);
        }
        if (INPLACE) __syncthreads();       // every wave has read the layer's input before anyone overwrites it
        if (!(ablate & 16))
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rt * 32 + j;
            const long grow = row0 + row;
            uint32_t mbits = 0;                 // ReLU signs of this lane's 32 values of the row (relu_mask_pos)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = (2 * wave + nt) * 32 + 8 * g + 4 * half;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float z = acc[rt][nt][4 * g + e];
                        if constexpr (SP) z = fmaf(ax[rt][nt][4 * g + e], SP_LO_INV, z);
                        v[e] = relu_push(z, mbits);
                    }
                    if constexpr (SP) {
                        h4 hi, lo;
                        split4(v, hi, lo);
                        amax = fmaxf(fmaxf(amax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));       // (post-ReLU: non-negative; two v_max3)
                        *(h4*)&nxt[a16_idx<WS>(row, f0)] = hi;
                        *(h4*)&nxt[a16_idx<WS>(row, 256 + f0)] = lo;
                    } else {
                        *(V4*)&nxt[a16_idx(row, f0)] = pack4(v[0], v[1], v[2], v[3], T());   // four consecutive features: one 8-byte store
                    }
                }
            }
            // one dword per lane and row: 32 lanes x 2 halves = 256 contiguous bytes per store (the [row][tile] words of rounds 1-5 --
            // half the lanes storing 4 bytes at a 32-byte stride after a cross-half shuffle -- cost 8 % of the kernel)
            if (masks && grow < p.rows) masks[relu_mask_idx(l, wave, grow, half, p.rows)] = mbits;
        }
        __syncthreads();
        if (saved_hT) store_T(nxt, saved_hT + (size_t)l * t16_elems(NS * M16_W, p.rows), M16_W, wave, lane);
        T* t = cur; cur = nxt; nxt = t;
    }
    if constexpr (SP) {
        // The range guard of "fp32s" (round-4 verdict): hi = fp16(x) saturates at 65504 -- silently.  An activation at or above 2^15
        // (a factor 2 of headroom) raises the caller's flag; the host then re-runs the pass on the exact-fp32 kernels or, checking
        // lazily, switches to them from the next pass on (deform_ops / deformable_field.py).  NaN compares false: not flagged here.
        if (p.range_flag && amax >= 32768.f) atomicOr(p.range_flag, 1u);
    }
    mlp16_output_layer<T, RT, SP>(p, out, cur, nxt, row0, tid);
}

__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_fwd_f16_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16_fwd_body<_Float16, 2>(p, out, (_Float16*)sx, (_Float16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_fwd_bf16_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16_fwd_body<__bf16, 2>(p, out, (__bf16*)sx, (__bf16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd_split_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16_fwd_body<_Float16, 2, true>(p, out, (_Float16*)sx, (_Float16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_f16_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16_fwd_body<_Float16, 4>(p, out, (_Float16*)sx, (_Float16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_bf16_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16_fwd_body<__bf16, 4>(p, out, (__bf16*)sx, (__bf16*)sh, masks);
}

// ------------------------------------------------------------------------------------------------
// Split mode, 64-row workgroups, input width 112 (the reference's 104 / 108 / 112 inputs): the forward on the hand-scheduled
// products.  One statement per layer (kloop16s_fwd) inside one rolled layer loop; TRAIN: the statement of layer l >= 1 also writes
// the saved tensor of layer l - 1 (the tile it reads), X and the last hidden layer leave as bursts (the latter has no product behind
// it to ride on).  Same arithmetic in the same order as mlp16_fwd_body<_Float16, 2, true>: bit-identical results
// (tools/probe/mlp16_kloop_ab.py, tests/test_gpu_deform.py).
// ------------------------------------------------------------------------------------------------
template <bool TRAIN>
__device__ __forceinline__ void mlp16s_fwd_hand_body(Mlp16Dev p, float* __restrict__ out, _Float16* __restrict__ saved_xT,
                                                     _Float16* __restrict__ saved_hT, uint32_t* __restrict__ masks) {
    constexpr int ROWS = 64, WS = SP_W, NS = 2;
    typedef _Float16 T;
    __shared__ __attribute__((aligned(1024))) T smem[ROWS * WS];       // one tile, updated in place
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: whatever is derived from it lives in scalar registers)
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;
    const int ablate = g_m16_ablate;
    if (ablate & 8) { for (int e = tid; e < ROWS * WS; e += M16_THREADS) smem[e] = (T)0.f; }
    else build_input16s<ROWS>(smem, p, row0, tid);
    __syncthreads();
    if (ablate & 2) masks = nullptr;
    auto burst = [&](T* dst, int nf, int lane) {            // this workgroup's block of a saved tensor: NS x nf x ROWS contiguous elements
        T* blk = dst + (size_t)blockIdx.x * NS * nf * ROWS;
        store_tile_T<T, ROWS, WS>(smem, blk, nf, NS * nf, row0, p.rows, rows_pad, wave, lane);
        store_tile_T<T, ROWS, WS>(smem + 256, blk + (size_t)nf * T16_BLK, nf, NS * nf, row0, p.rows, rows_pad, wave, lane);
    };
    if constexpr (TRAIN) burst(saved_xT, p.in_pad, lane_k);
    const size_t layer_elems = t16_elems(NS * M16_W, p.rows);
    const int bad_row = row0 + ROWS > p.rows ? (int)(p.rows - row0) : ROWS;        // the first row of the tile that does not exist
    int amax_bits = 0;                  // the largest hidden activation this lane carried into (hi, lo') form (its bit pattern)
    for (int l = 0; l < 4; ++l) {
        // Everything derived from the lane number is recomputed per layer (the empty statement hides the value's origin): hoisted out
        // of the layer loop, the epilogue's per-lane addresses were ~70 registers live ACROSS the product, which has none to spare --
        // its statement owns 104 fixed registers beside the 128 accumulators
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        f32x16 am[2][2], ax[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // this lane's features of tile nt: runs of four at (2 wave + nt) * 32 + 8 g + 4 half
                const float4 bv = *(const float4*)(p.b[l] + (2 * wave + nt) * 32 + 8 * g + 4 * half);
                am[0][nt][4 * g + 0] = bv.x; am[0][nt][4 * g + 1] = bv.y; am[0][nt][4 * g + 2] = bv.z; am[0][nt][4 * g + 3] = bv.w;
            }
        {   // (no "products off" switch here: a second definition of the accumulators costs ~130 moves per layer in THIS path)
            const Kloop16sAddr ka = kloop16s_addr(smem, wave, lane);
            _Float16* st = TRAIN ? saved_hT + (size_t)(l > 0 ? l - 1 : 0) * layer_elems + (size_t)blockIdx.x * NS * M16_W * ROWS + wave * (NS * M16_W * T16_BLK) : nullptr;
            kloop16s_fwd<TRAIN>(ka, (const _Float16*)p.w[l] + wave * 1024, (const _Float16*)p.wlo[l] + wave * 1024, st, l == 0, am, ax);
        }
        __syncthreads();                    // every wave has read the layer's input before anyone overwrites it
        // this lane's element (row j, feature 64 wave + 4 half) of the tile, as an LDS byte address; the feature bits of (nt, g) enter by XOR
        // (they lie inside the swizzled field), row tile and (hi | lo') half as instruction offsets: one v_xor per eight bytes stored
        const uint32_t ebase = (uint32_t)(uintptr_t)smem + 2 * a16_idx<WS>(j, 64 * wave + 4 * half);
        if (!(ablate & 16))
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long grow = row0 + rt * 32 + j;
            uint32_t mbits = 0;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_push(fmaf(ax[rt][nt][4 * g + e], SP_LO_INV, am[rt][nt][4 * g + e]), mbits);
                    h4 hi, lo;
                    split4(v, hi, lo);
                    // (non-negative floats order like their bit patterns: two v_max3_i32 per four values)
                    amax_bits = max(max(amax_bits, __float_as_int(v[0])), __float_as_int(v[1]));
                    amax_bits = max(max(amax_bits, __float_as_int(v[2])), __float_as_int(v[3]));
                    const uint32_t a = (ebase ^ (uint32_t)(64 * nt + 16 * g)) + rt * (32 * WS * 2);
                    *(lds_h4*)(uintptr_t)a = hi;
                    *(lds_h4*)(uintptr_t)(a + 512) = lo;
                }
            if (masks && grow < p.rows) masks[relu_mask_idx(l, wave, grow, half, p.rows)] = mbits;
        }
        __syncthreads();
        if constexpr (TRAIN) {
            if (bad_row < ROWS) {           // last workgroup only: rows beyond the input are stored as zeros (the weight gradient sums over them)
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                for (int e = tid; e < (ROWS - bad_row) * (WS * 2 / 16); e += M16_THREADS) ((u4*)(smem + (size_t)bad_row * WS))[e] = u4{0u, 0u, 0u, 0u};
                __syncthreads();
            }
            if (l == 3) burst(saved_hT + 3 * layer_elems, M16_W, lane);
        }
    }
    if (p.range_flag && __int_as_float(amax_bits) >= 32768.f) atomicOr(p.range_flag, 1u);         // the range guard of "fp32s": see mlp16_fwd_body
    mlp16_output_layer<T, 2, true>(p, out, smem, smem, row0, tid);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd_split_train_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16s_fwd_hand_body<true>(p, out, (_Float16*)sx, (_Float16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd_split_infer_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16s_fwd_hand_body<false>(p, out, nullptr, nullptr, nullptr);
}

// The same for the plain 16-bit kernels (fp16 / bf16 operands, 128-row workgroups, >= 65536 rows, input width 112): bit-identical
// to mlp16_fwd_body<T, 4>.
template <typename T, bool TRAIN>
__device__ __forceinline__ void mlp16p_fwd_hand_body(Mlp16Dev p, float* __restrict__ out, T* __restrict__ saved_xT, T* __restrict__ saved_hT,
                                                     uint32_t* __restrict__ masks) {
    constexpr int ROWS = 128, WS = M16_W;
    typedef typename Vec4<T>::type V4;
    typedef __attribute__((address_space(3))) V4 lds_v4;
    __shared__ __attribute__((aligned(1024))) T smem[ROWS * WS];       // one tile, updated in place
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;
    build_input16<T, ROWS>(smem, p, row0, tid);
    __syncthreads();
    if constexpr (TRAIN)
        store_tile_T<T, ROWS, WS>(smem, saved_xT + (size_t)blockIdx.x * p.in_pad * ROWS, p.in_pad, p.in_pad, row0, p.rows, rows_pad, wave, lane_k);
    const size_t layer_elems = t16_elems(M16_W, p.rows);
    const int bad_row = row0 + ROWS > p.rows ? (int)(p.rows - row0) : ROWS;
    for (int l = 0; l < 4; ++l) {
        int lane = lane_k;                  // (per-lane addresses recomputed per layer: see mlp16s_fwd_hand_body)
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        f32x16 c[4][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(p.b[l] + (2 * wave + nt) * 32 + 8 * g + 4 * half);
                c[0][nt][4 * g + 0] = bv.x; c[0][nt][4 * g + 1] = bv.y; c[0][nt][4 * g + 2] = bv.z; c[0][nt][4 * g + 3] = bv.w;
            }
        {
            const Kloop16sAddr ka = kloop16p_addr(smem, wave, lane);
            T* st = TRAIN ? saved_hT + (size_t)(l > 0 ? l - 1 : 0) * layer_elems + (size_t)blockIdx.x * M16_W * ROWS + wave * (2 * M16_W * T16_BLK) : nullptr;
            kloop16p_fwd<T, TRAIN>(ka, (const T*)p.w[l] + wave * 1024, st, l == 0, c);
        }
        __syncthreads();
        const uint32_t ebase = (uint32_t)(uintptr_t)smem + 2 * a16_idx<WS>(j, 64 * wave + 4 * half);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const long grow = row0 + rt * 32 + j;
            uint32_t mbits = 0;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_push(c[rt][nt][4 * g + e], mbits);
                    *(lds_v4*)(uintptr_t)((ebase ^ (uint32_t)(64 * nt + 16 * g)) + rt * (32 * WS * 2)) = pack4(v[0], v[1], v[2], v[3], T());
                }
            if (masks && grow < p.rows) masks[relu_mask_idx(l, wave, grow, half, p.rows)] = mbits;
        }
        __syncthreads();
        if constexpr (TRAIN) {
            if (bad_row < ROWS) {           // last workgroup only: rows beyond the input are stored as zeros
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                for (int e = tid; e < (ROWS - bad_row) * (WS * 2 / 16); e += M16_THREADS) ((u4*)(smem + (size_t)bad_row * WS))[e] = u4{0u, 0u, 0u, 0u};
                __syncthreads();
            }
            if (l == 3) store_tile_T<T, ROWS, WS>(smem, saved_hT + 3 * layer_elems + (size_t)blockIdx.x * M16_W * ROWS, M16_W, M16_W, row0, p.rows, rows_pad, wave, lane);
        }
    }
    mlp16_output_layer<T, 4, false>(p, out, smem, smem, row0, tid);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_f16_train_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16p_fwd_hand_body<_Float16, true>(p, out, (_Float16*)sx, (_Float16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_f16_infer_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16p_fwd_hand_body<_Float16, false>(p, out, nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_bf16_train_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16p_fwd_hand_body<__bf16, true>(p, out, (__bf16*)sx, (__bf16*)sh, masks);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_fwd4_bf16_infer_kernel(Mlp16Dev p, float* out, void* sx, void* sh, uint32_t* masks) {
    mlp16p_fwd_hand_body<__bf16, false>(p, out, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// backward, data chain.  wt[l] = TRANSPOSED 16-bit weights: wt[4]: [256][16] (k >= out_dim zero),
// wt[1..3]: [256][256] (= W_l^T), wt[0]: [in_pad][256] (= W_0^T, rows >= in_dim zero).
// Writes dZ_l TRANSPOSED ([4][256][rows], scaled) for the weight-gradient GEMM.
// ------------------------------------------------------------------------------------------------
// The tail of the data backward: dX[rows][in_pad] = dZ_1 . W_0 (feature tiles beyond in_pad are skipped; the result is kept in fp32 in
// LDS), then d feature and d xyz (through the encoding's derivative).  `cur` = dZ_1 (LDS), `nxt` = scratch, S = the loss scale.
template <typename T, int RT, bool SP>
__device__ __forceinline__ void mlp16_input_grad(const Mlp16Dev& p, const T* cur, T* nxt, float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                 float S, long row0, int tid) {
    constexpr int ROWS = 32 * RT;
    constexpr bool INPLACE = RT > 2 || SP;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    if (dfeature || dxyz) {
        // dX[64][in_pad] = dZ_1 . W_0 ; feature tiles beyond in_pad are skipped; result kept in fp32 in LDS
        f32x16 acc[RT][2];
        f32x16 ax[SP ? RT : 1][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[rt][nt][r] = 0.f; if (SP) ax[rt][nt][r] = 0.f; }
        if (wave * 64 < p.in_pad) {
            if constexpr (SP) gemm16s<false, RT>(cur, (const _Float16*)p.w[0], (const _Float16*)p.wlo[0], M16_W, M16_W, p.in_pad, wave, lane, acc, ax);
            else gemm16<T, false, RT>(cur, (const T*)p.w[0], M16_W, M16_W, p.in_pad, wave, lane, acc);
        }
        if (INPLACE) __syncthreads();
        float* dX = (float*)nxt;   // [ROWS][128] fp32
        const float inv = 1.f / S;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int f = (2 * wave + nt) * 32 + j;
                if (f < 128) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[rt][nt][r];
                        if constexpr (SP) v = fmaf(ax[rt][nt][r], SP_LO_INV, v);
                        dX[(rt * 32 + cd_row16(r, half)) * 128 + f] = v * inv;
                    }
                }
            }
        __syncthreads();
        if (dfeature) {
            const int fd = p.feature_dim;
            for (int e = tid; e < fd * ROWS; e += M16_THREADS) {
                const int r = e / fd, f = e - r * fd;
                if (row0 + r < p.rows) dfeature[(row0 + r) * fd + f] = dX[r * 128 + f];
            }
        }
        for (int e = tid; dxyz && e < 3 * ROWS; e += M16_THREADS) {
            const int r = e / 3, c = e % 3;
            const long row = row0 + r;
            if (row < p.rows) {
                const float x = p.xyz[row * 3 + c];
                float g = 0.f;
                for (int fr = 0; fr < p.xyz_freq; ++fr) {
                    const float sc = (float)(1u << fr);
                    float sv, cv;
                    if constexpr (SP) sincos_pe(x * sc, &sv, &cv);
                    else fast_sincos(x * sc, &sv, &cv);      // the same hardware sin/cos the forward encoded with
                    const int f = p.feature_dim + 2 * (c * p.xyz_freq + fr);
                    g += sc * (cv * dX[r * 128 + f] - sv * dX[r * 128 + f + 1]);
                }
                dxyz[row * 3 + c] = g;
            }
        }
    }
}

template <typename T, int RT, bool SP = false>
__device__ __forceinline__ void mlp16_bwd_data_body(Mlp16Dev p, const uint32_t* __restrict__ masks,
                                                    const float* __restrict__ dL_dout, T* __restrict__ dzT,
                                                    float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                    const uint32_t* __restrict__ absmax_bits) {
    constexpr int ROWS = 32 * RT;
    constexpr int WS = SP ? SP_W : M16_W;
    constexpr int NS = SP ? 2 : 1;
    constexpr bool INPLACE = RT > 2 || SP;
    __shared__ T smem[INPLACE ? 1 : 2][ROWS * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;
    const float S = UsesScale<T>::v ? grad_scale_from(absmax_bits) : 1.f;
    T* cur = smem[0];
    T* nxt = smem[INPLACE ? 0 : 1];
    {   // (all loads of the upstream gradient before the first LDS store: as a rolled loop they were dependent trips to memory)
        constexpr int DU = 16 * ROWS / M16_THREADS;
        float dv[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            const long row = row0 + r;
            const bool ok = f < p.out_dim && row < p.rows;
            dv[u] = dL_dout[ok ? row * p.out_dim + f : 0];
            if (!ok) dv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            const float v = dv[u] * S;
            if constexpr (SP) split1(v, cur[a16_idx<WS>(r, f)], cur[a16_idx<WS>(r, 256 + f)]);
            else cur[a16_idx(r, f)] = (T)v;
        }
    }
    __syncthreads();
    typedef typename Vec4<T>::type V4;
    for (int l = 4; l >= 1; --l) {
        const int K = l == 4 ? 16 : M16_W;
        f32x16 acc[RT][2];
        f32x16 ax[SP ? RT : 1][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[rt][nt][r] = 0.f; if (SP) ax[rt][nt][r] = 0.f; }
        // the ReLU sign words of this layer's epilogue, requested BEFORE the product (loaded where they are used, each of the
        // RT x 2 words was a dependent round trip to memory behind the matrix work: 16 per 64-row block and backward)
        constexpr bool PRE = RT == 2;          // (the 128-row variants have no registers to spare: they load in the epilogue)
        uint32_t mreg[RT];
        if (PRE) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const long grow = row0 + rt * 32 + j;
                mreg[rt] = masks[relu_mask_idx(l - 1, wave, grow < p.rows ? grow : p.rows - 1, half, p.rows)];
            }
        }
        if constexpr (SP) gemm16s<true, RT>(cur, (const _Float16*)p.w[l], (const _Float16*)p.wlo[l], K, K, M16_W, wave, lane, acc, ax);
        else gemm16<T, true, RT>(cur, (const T*)p.w[l], K, K, M16_W, wave, lane, acc);
        if (INPLACE) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rt * 32 + j;
            const long grow = row0 + row;
            const uint32_t m = grow < p.rows ? (PRE ? mreg[rt] : masks[relu_mask_idx(l - 1, wave, grow, half, p.rows)]) : 0u;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = (2 * wave + nt) * 32 + 8 * g + 4 * half;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // sign-extended mask bit (v_bfe_i32, constant position) AND value bits: two instructions per value
                        float z = acc[rt][nt][4 * g + e];
                        if constexpr (SP) z = fmaf(ax[rt][nt][4 * g + e], SP_LO_INV, z);
                        v[e] = __uint_as_float(__float_as_uint(z) & (uint32_t)__builtin_amdgcn_sbfe((int)m, relu_mask_pos(nt, g, e), 1));
                    }
                    if constexpr (SP) {
                        h4 hi, lo;
                        split4(v, hi, lo);
                        *(h4*)&nxt[a16_idx<WS>(row, f0)] = hi;
                        *(h4*)&nxt[a16_idx<WS>(row, 256 + f0)] = lo;
                    } else {
                        *(V4*)&nxt[a16_idx(row, f0)] = pack4(v[0], v[1], v[2], v[3], T());
                    }
                }
            }
        }
        __syncthreads();
        {   // dZ_l^T -> global, blocked
            T* blk = dzT + (size_t)(l - 1) * t16_elems(NS * M16_W, p.rows) + (size_t)blockIdx.x * NS * M16_W * ROWS;
            store_tile_T<T, ROWS, WS>(nxt, blk, M16_W, NS * M16_W, row0, p.rows, rows_pad, wave, lane);
            if constexpr (SP) store_tile_T<T, ROWS, WS>(nxt + 256, blk + (size_t)M16_W * T16_BLK, M16_W, NS * M16_W, row0, p.rows, rows_pad, wave, lane);
        }
        T* t = cur; cur = nxt; nxt = t;
    }
    mlp16_input_grad<T, RT, SP>(p, cur, nxt, dfeature, dxyz, S, row0, tid);
}
__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_bwd_data_f16_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                             void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16_bwd_data_body<_Float16, 2>(p, masks, dL_dout, (_Float16*)dzT, dfeature, dxyz, absmax_bits);
}
__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_bwd_data_bf16_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                              void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16_bwd_data_body<__bf16, 2>(p, masks, dL_dout, (__bf16*)dzT, dfeature, dxyz, absmax_bits);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data_split_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                  void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16_bwd_data_body<_Float16, 2, true>(p, masks, dL_dout, (_Float16*)dzT, dfeature, dxyz, absmax_bits);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data4_f16_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                 void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16_bwd_data_body<_Float16, 4>(p, masks, dL_dout, (_Float16*)dzT, dfeature, dxyz, absmax_bits);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data4_bf16_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                  void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16_bwd_data_body<__bf16, 4>(p, masks, dL_dout, (__bf16*)dzT, dfeature, dxyz, absmax_bits);
}

// Split mode, 64-row workgroups: the data backward on the hand-scheduled products (kloop16s_bwd): the statement of layer l <= 3 also
// writes dZ of layer index l (the tile it reads); index 0 leaves as a burst in front of the input-gradient product.  Rows beyond
// the input carry zeros by construction (their upstream gradient and their ReLU words are zero).  Bit-identical to
// mlp16_bwd_data_body<_Float16, 2, true>.
__device__ __forceinline__ void mlp16s_bwd_data_hand_body(Mlp16Dev p, const uint32_t* __restrict__ masks, const float* __restrict__ dL_dout,
                                                          _Float16* __restrict__ dzT, float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                          const uint32_t* __restrict__ absmax_bits) {
    constexpr int ROWS = 64, WS = SP_W, NS = 2;
    typedef _Float16 T;
    __shared__ __attribute__((aligned(1024))) T smem[ROWS * WS];
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;
    const float S = grad_scale_from(absmax_bits);
    {   // (all loads of the upstream gradient before the first LDS store)
        constexpr int DU = 16 * ROWS / M16_THREADS;
        float dv[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            const long row = row0 + r;
            const bool ok = f < p.out_dim && row < p.rows;
            dv[u] = dL_dout[ok ? row * p.out_dim + f : 0];
            if (!ok) dv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            split1(dv[u] * S, smem[a16_idx<WS>(r, f)], smem[a16_idx<WS>(r, 256 + f)]);
        }
    }
    __syncthreads();
    const size_t layer_elems = t16_elems(NS * M16_W, p.rows);
    for (int l = 4; l >= 1; --l) {
        int lane = lane_k;                  // (per-lane addresses recomputed per layer: see the forward)
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        uint32_t mreg[2];                   // the ReLU sign words of this layer's epilogue, requested BEFORE the product
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long grow = row0 + rt * 32 + j;
            mreg[rt] = masks[relu_mask_idx(l - 1, wave, grow < p.rows ? grow : p.rows - 1, half, p.rows)];
        }
        f32x16 am[2][2], ax[2][2];
        {
            const Kloop16sAddr ka = kloop16s_addr(smem, wave, lane);
            _Float16* st = dzT + (size_t)(l < 4 ? l : 0) * layer_elems + (size_t)blockIdx.x * NS * M16_W * ROWS + wave * (NS * M16_W * T16_BLK);
            kloop16s_bwd(ka, (const _Float16*)p.w[l] + wave * 1024, (const _Float16*)p.wlo[l] + wave * 1024, st, l == 4, am, ax);
        }
        __syncthreads();
        const uint32_t ebase = (uint32_t)(uintptr_t)smem + 2 * a16_idx<WS>(j, 64 * wave + 4 * half);      // (see the forward's epilogue)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const uint32_t m = row0 + rt * 32 + j < p.rows ? mreg[rt] : 0u;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // the sign-extended mask bit AND the value: two instructions (written as the instruction: from `z & sbfe(m, pos, 1)`
                        // hipcc makes v_and + v_cmp_ne + v_cndmask)
                        uint32_t keep;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"(relu_mask_pos(nt, g, e)));
                        v[e] = __uint_as_float(__float_as_uint(fmaf(ax[rt][nt][4 * g + e], SP_LO_INV, am[rt][nt][4 * g + e])) & keep);
                    }
                    h4 hi, lo;
                    split4(v, hi, lo);
                    const uint32_t a = (ebase ^ (uint32_t)(64 * nt + 16 * g)) + rt * (32 * WS * 2);
                    *(lds_h4*)(uintptr_t)a = hi;
                    *(lds_h4*)(uintptr_t)(a + 512) = lo;
                }
        }
        __syncthreads();
        if (l == 1) {   // dZ of layer index 0 -> global, blocked
            T* blk = dzT + (size_t)blockIdx.x * NS * M16_W * ROWS;
            store_tile_T<T, ROWS, WS>(smem, blk, M16_W, NS * M16_W, row0, p.rows, rows_pad, wave, lane);
            store_tile_T<T, ROWS, WS>(smem + 256, blk + (size_t)M16_W * T16_BLK, M16_W, NS * M16_W, row0, p.rows, rows_pad, wave, lane);
        }
    }
    mlp16_input_grad<T, 2, true>(p, smem, smem, dfeature, dxyz, S, row0, tid);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data_split_hand_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                       void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16s_bwd_data_hand_body(p, masks, dL_dout, (_Float16*)dzT, dfeature, dxyz, absmax_bits);
}

// ... and the plain 16-bit data backward (128-row workgroups): bit-identical to mlp16_bwd_data_body<T, 4>.
template <typename T>
__device__ __forceinline__ void mlp16p_bwd_data_hand_body(Mlp16Dev p, const uint32_t* __restrict__ masks, const float* __restrict__ dL_dout,
                                                          T* __restrict__ dzT, float* __restrict__ dfeature, float* __restrict__ dxyz,
                                                          const uint32_t* __restrict__ absmax_bits) {
    constexpr int ROWS = 128, WS = M16_W;
    typedef typename Vec4<T>::type V4;
    typedef __attribute__((address_space(3))) V4 lds_v4;
    __shared__ __attribute__((aligned(1024))) T smem[ROWS * WS];
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * ROWS;
    const long rows_pad = (p.rows + 63) & ~63L;
    const float S = UsesScale<T>::v ? grad_scale_from(absmax_bits) : 1.f;
    {
        constexpr int DU = 16 * ROWS / M16_THREADS;
        float dv[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            const long row = row0 + r;
            const bool ok = f < p.out_dim && row < p.rows;
            dv[u] = dL_dout[ok ? row * p.out_dim + f : 0];
            if (!ok) dv[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int e = tid + M16_THREADS * u, r = e / 16, f = e % 16;
            smem[a16_idx(r, f)] = (T)(dv[u] * S);
        }
    }
    __syncthreads();
    const size_t layer_elems = t16_elems(M16_W, p.rows);
    for (int l = 4; l >= 1; --l) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int half = lane >> 5, j = lane & 31;
        uint32_t mreg[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const long grow = row0 + rt * 32 + j;
            mreg[rt] = masks[relu_mask_idx(l - 1, wave, grow < p.rows ? grow : p.rows - 1, half, p.rows)];
        }
        f32x16 c[4][2];
        {
            const Kloop16sAddr ka = kloop16p_addr(smem, wave, lane);
            T* st = dzT + (size_t)(l < 4 ? l : 0) * layer_elems + (size_t)blockIdx.x * M16_W * ROWS + wave * (2 * M16_W * T16_BLK);
            kloop16p_bwd<T>(ka, (const T*)p.w[l] + wave * 1024, st, l == 4, c);
        }
        __syncthreads();
        const uint32_t ebase = (uint32_t)(uintptr_t)smem + 2 * a16_idx<WS>(j, 64 * wave + 4 * half);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const uint32_t m = row0 + rt * 32 + j < p.rows ? mreg[rt] : 0u;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t keep;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"(relu_mask_pos(nt, g, e)));
                        v[e] = __uint_as_float(__float_as_uint(c[rt][nt][4 * g + e]) & keep);
                    }
                    *(lds_v4*)(uintptr_t)((ebase ^ (uint32_t)(64 * nt + 16 * g)) + rt * (32 * WS * 2)) = pack4(v[0], v[1], v[2], v[3], T());
                }
        }
        __syncthreads();
        if (l == 1) store_tile_T<T, ROWS, WS>(smem, dzT + (size_t)blockIdx.x * M16_W * ROWS, M16_W, M16_W, row0, p.rows, rows_pad, wave, lane);
    }
    mlp16_input_grad<T, 4, false>(p, smem, smem, dfeature, dxyz, S, row0, tid);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data4_f16_hand_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                      void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16p_bwd_data_hand_body<_Float16>(p, masks, dL_dout, (_Float16*)dzT, dfeature, dxyz, absmax_bits);
}
__global__ __launch_bounds__(M16_THREADS, 2) void gp_mlp16_bwd_data4_bf16_hand_kernel(Mlp16Dev p, const uint32_t* masks, const float* dL_dout,
                                                                                       void* dzT, float* dfeature, float* dxyz, const uint32_t* absmax_bits) {
    mlp16p_bwd_data_hand_body<__bf16>(p, masks, dL_dout, (__bf16*)dzT, dfeature, dxyz, absmax_bits);
}

// ------------------------------------------------------------------------------------------------
// backward, weight grads: dW[o][i] += (1/S) sum_rows dZ^T[o][row] H^T[i][row]
// Both operands are stored feature-major, so a lane's 8 consecutive rows are one 16-byte load.
// grid = (row blocks, i-tile pairs, o-tile pairs); a wave owns a 64x64 output block (2x2 MFMA tiles),
// the 4 waves of a workgroup split the block's rows and are summed through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mlp16_bwd_weight_body(const T* __restrict__ dzT, int n_out, const T* __restrict__ hT, int n_in,
                                                      int nf_z, int nf_h, long rows, long rows_per_block, float* __restrict__ dW, int lddw,
                                                      float* __restrict__ db, const uint32_t* __restrict__ absmax_bits, float mul) {
    // nf_z / nf_h are the blocked layouts' features per row block (strides); split mode passes 2 nf with pre-offset pointers
    typedef typename Vec8<T>::type V8;
    const float inv_scale = (UsesScale<T>::v ? 1.f / grad_scale_from(absmax_bits) : 1.f) * mul;
    __shared__ float s_red[4][4][16][64];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    // grid = (tile pairs, row blocks): the 16 workgroups that share one slab of rows are dispatched back to back, so
    // the slab comes from HBM once and the other fifteen reads hit L2 / MALL (with row blocks as the fast index and few,
    // huge row blocks, every operand was streamed from HBM four times beyond ~256k rows)
    const int n_it = (n_in + 63) / 64;
    const int o0 = 64 * (blockIdx.x / n_it), i0 = 64 * (blockIdx.x % n_it);
    const long b_begin = (long)blockIdx.y * rows_per_block;
    long b_end = b_begin + rows_per_block;
    if (b_end > rows) b_end = rows;
    const long per_wave = ((b_end - b_begin + 3) / 4 + 15) & ~15L;
    const long r_begin = b_begin + wave * per_wave;
    long r_end = r_begin + per_wave;
    if (r_end > b_end) r_end = b_end;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    V8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (T)0.f;
    float bsum[2] = {0.f, 0.f};
    const bool full_rows = (rows % 8) == 0;
    for (long rb = r_begin; rb < r_end; rb += 16) {
        const long r8 = rb + 8 * half;
        V8 av[2], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int o = o0 + 32 * u + j, i = i0 + 32 * u + j;
            av[u] = zero; bv[u] = zero;
            if (full_rows && r8 + 8 <= r_end) {
                if (o < n_out) av[u] = *(const V8*)(dzT + t16_idx(o, r8, nf_z));
                if (i < n_in) bv[u] = *(const V8*)(hT + t16_idx(i, r8, nf_h));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (r8 + e < r_end) {
                        if (o < n_out) av[u][e] = dzT[t16_idx(o, r8 + e, nf_z)];
                        if (i < n_in) bv[u][e] = hT[t16_idx(i, r8 + e, nf_h)];
                    }
                }
            }
        }
        if (db && i0 == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[u] += (float)av[u][e];
        }
        acc[0][0] = mfma16(av[0], bv[0], acc[0][0]);
        acc[0][1] = mfma16(av[0], bv[1], acc[0][1]);
        acc[1][0] = mfma16(av[1], bv[0], acc[1][0]);
        acc[1][1] = mfma16(av[1], bv[1], acc[1][1]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[wave][2 * a + b][r][lane] = acc[a][b][r];
    __syncthreads();
    const bool single = gridDim.y == 1;
    for (int e = tid; e < 4 * 16 * 64; e += M16_THREADS) {
        const int tl = e >> 10, r = (e >> 6) & 15, l = e & 63;
        float v = s_red[0][tl][r][l] + s_red[1][tl][r][l] + s_red[2][tl][r][l] + s_red[3][tl][r][l];
        const int oo = o0 + 32 * (tl >> 1) + cd_row16(r, l >> 5), ii = i0 + 32 * (tl & 1) + (l & 31);
        if (oo < n_out && ii < n_in) {
            float* dst = &dW[(size_t)oo * lddw + ii];
            v *= inv_scale;
            if (single) *dst += v; else atomicAdd(dst, v);
        }
    }
    if (db && i0 == 0) {
        __syncthreads();
        float* sb = &s_red[0][0][0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v = bsum[u] + __shfl_xor(bsum[u], 32);
            if (half == 0) sb[(wave * 2 + u) * 32 + j] = v;
        }
        __syncthreads();
        if (tid < 64) {
            const int u = tid >> 5, jj = tid & 31;
            const float v = (sb[(0 * 2 + u) * 32 + jj] + sb[(1 * 2 + u) * 32 + jj] + sb[(2 * 2 + u) * 32 + jj] + sb[(3 * 2 + u) * 32 + jj]) * inv_scale;
            const int oo = o0 + 32 * u + jj;
            if (oo < n_out) { if (single) db[oo] += v; else atomicAdd(&db[oo], v); }
        }
    }
}
__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_bwd_weight_f16_kernel(const void* dzT, int n_out, const void* hT, int n_in, int nf_z, int nf_h, long rows,
                                                                               long rpb, float* dW, int lddw, float* db, const uint32_t* absmax_bits, float mul) {
    mlp16_bwd_weight_body<_Float16>((const _Float16*)dzT, n_out, (const _Float16*)hT, n_in, nf_z, nf_h, rows, rpb, dW, lddw, db, absmax_bits, mul);
}
__global__ __launch_bounds__(M16_THREADS) void gp_mlp16_bwd_weight_bf16_kernel(const void* dzT, int n_out, const void* hT, int n_in, int nf_z, int nf_h, long rows,
                                                                                long rpb, float* dW, int lddw, float* db, const uint32_t* absmax_bits, float mul) {
    mlp16_bwd_weight_body<__bf16>((const __bf16*)dzT, n_out, (const __bf16*)hT, n_in, nf_z, nf_h, rows, rpb, dW, lddw, db, absmax_bits, mul);
}

// The three 256x256 layers at large row counts: ONE workgroup owns the whole 256x256 gradient of (layer, row slab), so
// HBM sees each saved activation / dZ exactly once (the 64x64-per-wave kernel above re-read every slab through L2 sixteen
// times and ran at 165 TF/s).  The kernel is HBM-bound by design (1 KB of operands per row and layer), so it is built
// around bytes in flight: per k-step (16 rows) the 8 waves fetch the 16 unique 1 KB fragments (2 each) into registers
// W16_DEPTH k-steps ahead (64 KB of unique data in flight per CU), pass them through a double-buffered 32 KB LDS stage,
// and every wave reads the 6 fragments of its 128x64 block (4x2 MFMA tiles, 128 accumulator registers) back from LDS.
// The barrier waits only for LDS (lgkmcnt); the global prefetches stay in flight across it.
// The same kernel, shaped by <OT, IT, WI> (32x32 tiles per wave along o and i, waves along i), covers all five layers:
//   layers 1..3 (256 x 256):  <4, 2, 4>  one launch, blockIdx.y = layer
//   layer 0     (256 x in_pad <= 128): <2, 2, 2>
//   layer 4     (out_dim <= 16 x 256, dZ = the packed dL_dout): <1, 1, 8>
// Fragments beyond a tensor's feature count are loaded from a clamped address and zeroed, so every wave issues exactly
// two loads per k-step whatever the shape.
struct W16Job { const void* z; const void* h; float* dw; float* db; float mul; };   // dw += mul * dZ^T H (db may be null)
struct W16Jobs { W16Job j[9]; };
// nf_* = features a job reads, ks_* = features per row block of the tensor (= nf, or 2 nf in split mode with pre-offset
// pointers); shared = several jobs of one launch add into the same dw
struct W16Shape { int nf_z, nf_h, n_out, n_in, lddw, ks_z, ks_h, shared, xcd_terms, n_jobs, n_slabs; };
#define W16_THREADS 512
#ifndef W16_DEPTH
#define W16_DEPTH 4
#endif
#define W16_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <typename V8>
__device__ __forceinline__ V8 w16_keep(V8 v, bool ok) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 b = __builtin_bit_cast(u4, v);
    if (!ok) b = u4{0u, 0u, 0u, 0u};
    return __builtin_bit_cast(V8, b);
}

// One group of W16_DEPTH k-steps: write the staged fragments of step k to LDS, re-issue the stage's two loads for step
// k + W16_DEPTH (the tail re-loads its own step: every k-step issues exactly two loads, in a fixed order, so the
// compiler's vmcnt before an LDS write leaves the 2 (W16_DEPTH - 1) younger loads in flight), barrier, MFMA from LDS.
template <typename T, typename V8, int OT, int IT, int WI>
__device__ __forceinline__ void w16_group(V8 (&gq)[W16_DEPTH][2], f32x16 (&acc)[OT][IT], float (&bsum)[OT], T (*s_st)[2][2][8][512],
                                          const T* gz, const T* gh, bool okz, bool okh, long k0, long nk, size_t kstride_z,
                                          size_t kstride_h, int wave, int lane, int wo, int wi, int frag_off) {
    // TWO k-steps (32 rows) per barrier.  With 128 accumulator registers per wave one workgroup fills a CU, so nothing overlaps
    // the chain  LDS write -> barrier -> LDS read -> MFMA  of a step; at one k-step per barrier that chain (~1 us) times the 256
    // steps of a slab WAS the kernel's time (2.1 ms at 1 M rows for 8.5 GB of unique operands and 0.6 ms of matrix work).
    static_assert(W16_DEPTH % 4 == 0, "pairs of k-steps, two stage buffers");
#pragma unroll
    for (int d = 0; d < W16_DEPTH; d += 2) {
        const int buf = (d >> 1) & 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const long k = k0 + d + kk;
            *(V8*)&s_st[buf][kk][0][wave][lane * 8] = w16_keep(gq[d + kk][0], okz);
            *(V8*)&s_st[buf][kk][1][wave][lane * 8] = w16_keep(gq[d + kk][1], okh);
            const long kn = k + W16_DEPTH < nk ? k + W16_DEPTH : k;
            gq[d + kk][0] = *(const V8*)(gz + kn * kstride_z);
            gq[d + kk][1] = *(const V8*)(gh + kn * kstride_h);
        }
        W16_LDS_BARRIER();          // one barrier per PAIR of k-steps: a buffer is rewritten two pairs later, behind the next barrier
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            V8 a[OT], b[IT];
#pragma unroll
            for (int u = 0; u < OT; ++u) a[u] = *(const V8*)&s_st[buf][kk][0][wo * OT + u][frag_off];
#pragma unroll
            for (int u = 0; u < IT; ++u) b[u] = *(const V8*)&s_st[buf][kk][1][wi * IT + u][frag_off];
#pragma unroll
            for (int uo = 0; uo < OT; ++uo)
#pragma unroll
                for (int ui = 0; ui < IT; ++ui) acc[uo][ui] = mfma16(a[uo], b[ui], acc[uo][ui]);
            if (wi == 0) {
#pragma unroll
                for (int u = 0; u < OT; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[u] += (float)a[u][e];
            }
        }
    }
}

template <typename T, int OT, int IT, int WI>
__device__ __forceinline__ void mlp16_bwd_weight_lds_body(W16Jobs jobs, W16Shape sh, long n_kb, long kb_per_slab,
                                                          const uint32_t* __restrict__ absmax_bits) {
    typedef typename Vec8<T>::type V8;
    static_assert((8 / WI) * OT <= 8 && WI * IT <= 8, "eight fragment slots per operand");
    __shared__ T s_st[2][2][2][8][512];                        // [buffer][k-step of the pair][dZ | H][fragment of 32 features][16 rows x 32] = 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, j = lane & 31;
    const int wo = wave / WI, wi = wave % WI;
    // grid = (jobs, row slabs), jobs fastest: the workgroups that read the same slab are dispatched back to back.
    // Split mode (xcd_terms = 3): the three terms of a layer read dZ-hi and H-hi twice each.  Workgroups go round-robin over
    // the 8 XCDs (linear id & 7), each with its own L2, so "back to back" put the three terms on three L2s and every repeat
    // went out to the fabric.  1-D grid instead: id -> (xcd, q); the terms of unit u = 8 (q / 3) + xcd are q % 3 -- one XCD,
    // consecutive dispatch slots -- and units enumerate (slab, layer) with the layers of a slab adjacent.
    int job_idx = blockIdx.x;
    long slab = blockIdx.y;
    if (sh.xcd_terms > 0) {
        const long id = blockIdx.x, q = id >> 3;
        const int nlay = sh.n_jobs / sh.xcd_terms;
        const long u = (q / sh.xcd_terms) * 8 + (id & 7);
        slab = u / nlay;
        if (slab >= sh.n_slabs) return;
        job_idx = (int)(u % nlay) * sh.xcd_terms + (int)(q % sh.xcd_terms);
    }
    const W16Job job = jobs.j[job_idx];
    const float inv_scale = (UsesScale<T>::v ? 1.f / grad_scale_from(absmax_bits) : 1.f) * job.mul;
    const long kb0 = slab * kb_per_slab;
    long kb1 = kb0 + kb_per_slab;
    if (kb1 > n_kb) kb1 = n_kb;
    const long nk = kb1 - kb0;
    const size_t kstride_z = (size_t)sh.ks_z * T16_BLK, kstride_h = (size_t)sh.ks_h * T16_BLK;   // elements per 16-row block
    // this wave's share of a k-step: fragment `wave` of dZ and of H in lane-linear 16-byte chunks (chunk = feature, 8 rows)
    const int nfr_z = (sh.nf_z + 31) / 32, nfr_h = (sh.nf_h + 31) / 32;
    const int fz = wave < nfr_z ? wave : nfr_z - 1, fh = wave < nfr_h ? wave : nfr_h - 1;
    const bool okz = wave < nfr_z && fz * 32 + (lane >> 1) < sh.nf_z, okh = wave < nfr_h && fh * 32 + (lane >> 1) < sh.nf_h;
    const T* gz = (const T*)job.z + (size_t)kb0 * kstride_z + fz * 512 + (okz ? lane * 8 : 0);
    const T* gh = (const T*)job.h + (size_t)kb0 * kstride_h + fh * 512 + (okh ? lane * 8 : 0);
    const int frag_off = (2 * j + half) * 8;                    // (feature j, rows 8 half ..) inside a fragment
    f32x16 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[OT];
#pragma unroll
    for (int a = 0; a < OT; ++a) bsum[a] = 0.f;
    // nk is a multiple of W16_DEPTH (the host rounds slabs to 64 rows; the tensors are zero-padded to 64 rows).  The first
    // group is peeled so that the loop header sees the same load order from both of its predecessors.
    V8 gq[W16_DEPTH][2];
#pragma unroll
    for (int d = 0; d < W16_DEPTH; ++d) { gq[d][0] = *(const V8*)(gz + d * kstride_z); gq[d][1] = *(const V8*)(gh + d * kstride_h); }
    if (nk > 0) w16_group<T, V8, OT, IT, WI>(gq, acc, bsum, s_st, gz, gh, okz, okh, 0, nk, kstride_z, kstride_h, wave, lane, wo, wi, frag_off);
    for (long k0 = W16_DEPTH; k0 < nk; k0 += W16_DEPTH)
        w16_group<T, V8, OT, IT, WI>(gq, acc, bsum, s_st, gz, gh, okz, okh, k0, nk, kstride_z, kstride_h, wave, lane, wo, wi, frag_off);
    const bool single = gridDim.y == 1 && !sh.shared && sh.xcd_terms == 0;
#pragma unroll
    for (int uo = 0; uo < OT; ++uo)
#pragma unroll
        for (int ui = 0; ui < IT; ++ui) {
            __builtin_amdgcn_sched_barrier(0);      // one tile at a time
            const int i = (wi * IT + ui) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (wo * OT + uo) * 32 + cd_row16(r, half);
                if (o < sh.n_out && i < sh.n_in) {
                    const float v = acc[uo][ui][r] * inv_scale;
                    float* dst = job.dw + (size_t)o * sh.lddw + i;
                    if (single) *dst += v; else atomicAdd(dst, v);
                }
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    if (wi == 0) {
#pragma unroll
        for (int u = 0; u < OT; ++u) {
            const float v = (bsum[u] + __shfl_xor(bsum[u], 32)) * inv_scale;
            const int o = (wo * OT + u) * 32 + j;
            if (half == 0 && o < sh.n_out && job.db) { float* dst = &job.db[o]; if (single) *dst += v; else atomicAdd(dst, v); }
        }
    }
}
template <typename T, int OT, int IT, int WI>
__global__ __launch_bounds__(W16_THREADS) void gp_mlp16_bwd_weight_lds_kernel(W16Jobs jobs, W16Shape sh, long n_kb, long kb_per_slab,
                                                                             const uint32_t* absmax_bits) {
    mlp16_bwd_weight_lds_body<T, OT, IT, WI>(jobs, sh, n_kb, kb_per_slab, absmax_bits);
}

// dL_dout^T in 16 bits, scaled: [16][rows] (rows >= out_dim zero) -- A operand of the last layer's weight gradient
template <typename T>
__global__ __launch_bounds__(256) void gp_mlp16_pack_dout_kernel(const float* __restrict__ dL_dout, int out_dim, long rows,
                                                                T* __restrict__ dst, const uint32_t* __restrict__ absmax_bits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ((rows + 63) & ~63L)) return;                      // the padding rows (to 64) are written as zeros
    const float scale = UsesScale<T>::v ? grad_scale_from(absmax_bits) : 1.f;
    float v[16];
    const long ii = i < rows ? i : rows - 1;                    // (the row's values first, all in flight: clamped addresses)
#pragma unroll
    for (int f = 0; f < 16; ++f) v[f] = dL_dout[ii * out_dim + (f < out_dim ? f : out_dim - 1)];
#pragma unroll
    for (int f = 0; f < 16; ++f) dst[t16_idx(f, i, 16)] = (T)((f < out_dim && i < rows) ? v[f] * scale : 0.f);
}

// split mode: [16 hi | 16 lo'] "features"
__global__ __launch_bounds__(256) void gp_mlp16_pack_dout_split_kernel(const float* __restrict__ dL_dout, int out_dim, long rows,
                                                                      _Float16* __restrict__ dst, const uint32_t* __restrict__ absmax_bits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ((rows + 63) & ~63L)) return;
    const float scale = grad_scale_from(absmax_bits);
    float v[16];
    const long ii = i < rows ? i : rows - 1;
#pragma unroll
    for (int f = 0; f < 16; ++f) v[f] = dL_dout[ii * out_dim + (f < out_dim ? f : out_dim - 1)];
#pragma unroll
    for (int f = 0; f < 16; ++f)
        split1((f < out_dim && i < rows) ? v[f] * scale : 0.f, dst[t16_idx(f, i, 32)], dst[t16_idx(16 + f, i, 32)]);
}

// ------------------------------------------------------------------------------------------------
// gp_mlp16_pack: the five weight matrices -> their zero-padded, fragment-packed 16-bit copies (forward or transposed operand order;
// split mode: [hi copy][lo' copy]) in ONE launch.  The torch expression of the same (pad, clamp, compare, where, two casts, subtract,
// scale, reshape / permute / contiguous, stack: ~14 launches per matrix, ten matrices per training step) was 0.4 ms of launch-bound
// work per step beside kernels of 2 ms.
// ------------------------------------------------------------------------------------------------
struct Pack16Job {
    const float* w;     // nn.Linear weight [n_out][n_in], row-major
    void* out;
    int F, K;           // packed matrix [F][K]
    int n_out, n_in;
};
struct Pack16Jobs {
    Pack16Job j[5];
    long start[6];      // first packed position of every layer
};
__host__ __device__ __forceinline__ long mlp16_packed_elems(int l, int in_pad, bool transposed) {
    return l == 0 ? (transposed ? (long)((in_pad + 31) / 32 * 32) * 256 : 256L * in_pad) : l < 4 ? 65536L : (transposed ? 256L * 16 : 32L * 256);
}
template <int MODE /* 0 fp16, 1 bf16, 2 split fp16 */>
__global__ __launch_bounds__(256) void gp_mlp16_pack_kernel(Pack16Jobs jobs, int transposed) {
    const long pos = (long)blockIdx.x * 256 + threadIdx.x;
    if (pos >= jobs.start[5]) return;
    int l = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) l += pos >= jobs.start[i];
    const Pack16Job jb = jobs.j[l];
    const long q = pos - jobs.start[l];
    // position -> (f, k): (((k / 16) * (F / 32) + f / 32) * 64 + ((k / 8) & 1) * 32 + f % 32) * 8 + k % 8
    const int e = (int)(q & 7), lane64 = (int)((q >> 3) & 63);
    const long tile = q >> 9;
    const int ft = (int)(tile % (jb.F / 32)), ks = (int)(tile / (jb.F / 32));
    const int f = ft * 32 + (lane64 & 31), k = ks * 16 + (lane64 >> 5) * 8 + e;
    float v = 0.f;
    if (!transposed) { if (f < jb.n_out && k < jb.n_in) v = jb.w[(size_t)f * jb.n_in + k]; }
    else { if (k < jb.n_out && f < jb.n_in) v = jb.w[(size_t)k * jb.n_in + f]; }
    if constexpr (MODE == 0) ((_Float16*)jb.out)[q] = (_Float16)v;
    else if constexpr (MODE == 1) ((__bf16*)jb.out)[q] = (__bf16)v;
    else {
        _Float16 hi, lo;
        split1(v, hi, lo);
        ((_Float16*)jb.out)[q] = hi;
        ((_Float16*)jb.out)[(size_t)jb.F * jb.K + q] = lo;
    }
}
extern "C" int64_t gp_mlp16_packed_elems(int32_t layer, int32_t in_dim, int32_t transposed) {
    if (layer < 0 || layer > 4 || in_dim <= 0) return 0;
    return mlp16_packed_elems(layer, (in_dim + 15) / 16 * 16, transposed != 0);
}
extern "C" int gp_mlp16_pack(const gp_mlp_params* p, int32_t dtype, int32_t transposed, void* const* out, gp_stream_t stream_) {
    if (!p || !out) GP_FAIL("null mlp16 pack argument");
    if (dtype != GP_DTYPE_F16 && dtype != GP_DTYPE_BF16 && dtype != GP_DTYPE_F16_SPLIT) GP_FAIL("mlp16 pack: dtype must be GP_DTYPE_F16, GP_DTYPE_BF16 or GP_DTYPE_F16_SPLIT");
    if (p->width != 256 || p->depth != 4) GP_FAIL("Deformable_Field: only d=4, w=256 is implemented");
    if (p->in_dim <= 0 || p->in_dim > 128 || p->out_dim < 1 || p->out_dim > 16) GP_FAIL("mlp16 pack: in_dim must be 1..128, out_dim 1..16");
    const int in_pad = (p->in_dim + 15) / 16 * 16;
    Pack16Jobs jobs;
    long at = 0;
    for (int l = 0; l < 5; ++l) {
        if (!p->w[l] || !out[l]) GP_FAIL("null weight / output pointer (layer %d)", l);
        Pack16Job& jb = jobs.j[l];
        jb.w = p->w[l]; jb.out = out[l];
        jb.n_out = l < 4 ? 256 : p->out_dim; jb.n_in = l == 0 ? p->in_dim : 256;
        if (!transposed) { jb.F = l < 4 ? 256 : 32; jb.K = l == 0 ? in_pad : 256; }
        else { jb.F = l == 0 ? (in_pad + 31) / 32 * 32 : 256; jb.K = l < 4 ? 256 : 16; }
        jobs.start[l] = at;
        at += (long)jb.F * jb.K;
    }
    jobs.start[5] = at;
    hipStream_t s = (hipStream_t)stream_;
    const dim3 grid(gp_blocks((size_t)at, 256));
    if (dtype == GP_DTYPE_F16) hipLaunchKernelGGL(gp_mlp16_pack_kernel<0>, grid, dim3(256), 0, s, jobs, (int)(transposed != 0));
    else if (dtype == GP_DTYPE_BF16) hipLaunchKernelGGL(gp_mlp16_pack_kernel<1>, grid, dim3(256), 0, s, jobs, (int)(transposed != 0));
    else hipLaunchKernelGGL(gp_mlp16_pack_kernel<2>, grid, dim3(256), 0, s, jobs, (int)(transposed != 0));
    GP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int make16(const gp_mlp16_params* p, const gp_mlp_input* x, Mlp16Dev& m, bool transposed) {
    if (!p || !x) GP_FAIL("null mlp16 params/input");
    if (p->dtype != GP_DTYPE_F16 && p->dtype != GP_DTYPE_BF16 && p->dtype != GP_DTYPE_F16_SPLIT)
        GP_FAIL("mlp16: dtype must be GP_DTYPE_F16, GP_DTYPE_BF16 or GP_DTYPE_F16_SPLIT");
    if (p->width != 256 || p->depth != 4) GP_FAIL("Deformable_Field: only d=4, w=256 is implemented");
    if (p->out_dim < 7 || p->out_dim > 8) GP_FAIL("out_dim must be 7 or 8");
    const int in_dim = x->feature_dim + 6 * x->xyz_freq + 2 * x->time_freq;
    if (in_dim != p->in_dim || in_dim > 128 || in_dim <= 0) GP_FAIL("in_dim mismatch or > 128 (%d)", in_dim);
    if (x->rows < 0) GP_FAIL("negative rows");
    for (int l = 0; l < 5; ++l)
        if (!p->w16[l] || !p->b[l]) GP_FAIL("null weight pointer (layer %d)", l);
    if (x->rows > 0 && (!x->feature || (x->xyz_freq > 0 && !x->xyz) || (x->time_freq > 0 && !x->t))) GP_FAIL("null input pointer");
    m.rows = x->rows; m.in_dim = in_dim; m.in_pad = (in_dim + 15) / 16 * 16; m.out_dim = p->out_dim;
    m.feature_dim = x->feature_dim; m.xyz_freq = x->xyz_freq; m.time_freq = x->time_freq;
    for (int l = 0; l < 5; ++l) {
        m.w[l] = p->w16[l]; m.b[l] = p->b[l];
        // split mode: each array is [hi copy][lo' copy]
        const size_t elems = l == 0 ? (transposed ? (size_t)((m.in_pad + 31) / 32 * 32) * 256 : (size_t)256 * m.in_pad)
                                    : l < 4 ? (size_t)65536 : (transposed ? (size_t)256 * 16 : (size_t)32 * 256);
        m.wlo[l] = p->dtype == GP_DTYPE_F16_SPLIT ? (const void*)((const _Float16*)p->w16[l] + elems) : nullptr;
    }
    m.feature = x->feature; m.xyz = x->xyz; m.t = x->t;
    m.range_flag = p->range_flag;
    return 0;
}

extern "C" int gp_mlp16_forward(const gp_mlp16_params* p, const gp_mlp_input* x, float* out, void* saved_xT, void* saved_hT,
                                uint32_t* masks, gp_stream_t stream_) {
    Mlp16Dev m;
    if (make16(p, x, m, false)) return 1;
    if (m.rows == 0) return 0;
    if (!out) GP_FAIL("null output");
    hipStream_t s = (hipStream_t)stream_;
    {
        static thread_local int ablate_set = 0;
        if (gp_debug_get(9) != ablate_set) { ablate_set = gp_debug_get(9); if (m16_set_ablate(ablate_set)) GP_FAIL("mlp16 ablate flag"); }
    }
    GpProfScope _p("mlp16_fwd", s);
    if (p->dtype == GP_DTYPE_F16_SPLIT) {
        size_t dyn = 0;
        if (gp_debug_get(9) & 32) {       // (probe: 40 KB of unused dynamic LDS = ONE workgroup per CU instead of two)
            static thread_local bool set = false;
            if (!set) { GP_HIP_CHECK(hipFuncSetAttribute((const void*)gp_mlp16_fwd_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024)); set = true; }
            dyn = 40 * 1024;
        }
        // the hand-scheduled kernels (gp_debug_option(9, 64): the compiler's loops, for A/B; other input widths than 112 keep them too)
        const bool hand = m.in_pad == 112 && !(gp_debug_get(9) & (64 | 32 | 4));
        const bool train = saved_xT && saved_hT && masks && !(gp_debug_get(9) & 1);
        if (hand && train) hipLaunchKernelGGL(gp_mlp16_fwd_split_train_kernel, dim3(gp_blocks((size_t)m.rows, M16_ROWS)), dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else if (hand && ((!saved_xT && !saved_hT && !masks) || (gp_debug_get(9) & 1)))
            hipLaunchKernelGGL(gp_mlp16_fwd_split_infer_kernel, dim3(gp_blocks((size_t)m.rows, M16_ROWS)), dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else hipLaunchKernelGGL(gp_mlp16_fwd_split_kernel, dim3(gp_blocks((size_t)m.rows, M16_ROWS)), dim3(M16_THREADS), dyn, s, m, out, saved_xT, saved_hT, masks);
    } else if (m.rows >= GP_MLP16_BIG_ROWS) {        // 128 rows per workgroup
        const dim3 grid(gp_blocks((size_t)m.rows, 128));
        const bool f16 = p->dtype == GP_DTYPE_F16;
        const bool hand = m.in_pad == 112 && gp_debug_get(9) == 0;        // (any gp_debug_option(9, bits): the compiler's loops, for A/B and ablation)
        const bool train = saved_xT && saved_hT && masks;
        if (hand && train) hipLaunchKernelGGL(f16 ? gp_mlp16_fwd4_f16_train_kernel : gp_mlp16_fwd4_bf16_train_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else if (hand && !saved_xT && !saved_hT && !masks) hipLaunchKernelGGL(f16 ? gp_mlp16_fwd4_f16_infer_kernel : gp_mlp16_fwd4_bf16_infer_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else if (f16) hipLaunchKernelGGL(gp_mlp16_fwd4_f16_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else hipLaunchKernelGGL(gp_mlp16_fwd4_bf16_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
    } else {
        const dim3 grid(gp_blocks((size_t)m.rows, M16_ROWS));
        if (p->dtype == GP_DTYPE_F16) hipLaunchKernelGGL(gp_mlp16_fwd_f16_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
        else hipLaunchKernelGGL(gp_mlp16_fwd_bf16_kernel, grid, dim3(M16_THREADS), 0, s, m, out, saved_xT, saved_hT, masks);
    }
    GP_LAUNCH_CHECK();
    return 0;
}

extern "C" int gp_mlp16_backward(const gp_mlp16_params* p /* w16 = TRANSPOSED copies */, const gp_mlp_input* x, const void* saved_xT,
                                 const void* saved_hT, const uint32_t* masks, const float* dL_dout, gp_mlp_grads* g,
                                 float* dL_dfeature, float* dL_dxyz, gp_alloc_fn alloc, void* alloc_ctx, gp_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    Mlp16Dev m;
    if (make16(p, x, m, true)) return 1;
    if (m.rows == 0) return 0;
    if (!saved_xT || !saved_hT || !masks || !dL_dout || !g || !alloc) GP_FAIL("null argument");
    for (int l = 0; l < 5; ++l)
        if (!g->dw[l] || !g->db[l]) GP_FAIL("null weight-grad pointer (layer %d)", l);
    const bool split = p->dtype == GP_DTYPE_F16_SPLIT;
    const bool f16 = p->dtype == GP_DTYPE_F16 || split;
    const int NS = split ? 2 : 1;
    const size_t dz_elems = 4 * t16_elems(NS * 256, m.rows) + t16_elems(NS * 16, m.rows);
    char* dz = (char*)alloc(alloc_ctx, GP_BUF_TEMP, gp_align_up(dz_elems * 2, 256) + 256);
    if (!dz) GP_FAIL("allocator returned NULL for TEMP");
    char* dout16 = dz + 4 * t16_elems(NS * 256, m.rows) * 2;
    uint32_t* absmax = (uint32_t*)(dz + gp_align_up(dz_elems * 2, 256));
    GP_HIP_CHECK(hipMemsetAsync(absmax, 0, 4, s));
    if (f16) {
        hipLaunchKernelGGL(gp_absmax_kernel, dim3(256), dim3(256), 0, s, dL_dout, (size_t)m.rows * m.out_dim, absmax);
        GP_LAUNCH_CHECK();
    }
    {
        GpProfScope _p("mlp16_bwd_data", s);
        const bool big_rows = m.rows >= GP_MLP16_BIG_ROWS;
        const dim3 grid(gp_blocks((size_t)m.rows, big_rows ? 128 : M16_ROWS));
        if (split) {
            if (gp_debug_get(9) & 64) hipLaunchKernelGGL(gp_mlp16_bwd_data_split_kernel, dim3(gp_blocks((size_t)m.rows, M16_ROWS)), dim3(M16_THREADS), 0, s, m, masks, dL_dout, (void*)dz, dL_dfeature, dL_dxyz, absmax);
            else hipLaunchKernelGGL(gp_mlp16_bwd_data_split_hand_kernel, dim3(gp_blocks((size_t)m.rows, M16_ROWS)), dim3(M16_THREADS), 0, s, m, masks, dL_dout, (void*)dz, dL_dfeature, dL_dxyz, absmax);
            hipLaunchKernelGGL(gp_mlp16_pack_dout_split_kernel, dim3(gp_blocks((size_t)((m.rows + 63) & ~63L), 256)), dim3(256), 0, s, dL_dout, m.out_dim, m.rows, (_Float16*)dout16, absmax);
        } else if (f16) {
            const bool hand = big_rows && gp_debug_get(9) == 0;
            hipLaunchKernelGGL(hand ? gp_mlp16_bwd_data4_f16_hand_kernel : big_rows ? gp_mlp16_bwd_data4_f16_kernel : gp_mlp16_bwd_data_f16_kernel, grid, dim3(M16_THREADS), 0, s, m, masks, dL_dout, (void*)dz, dL_dfeature, dL_dxyz, absmax);
            hipLaunchKernelGGL((gp_mlp16_pack_dout_kernel<_Float16>), dim3(gp_blocks((size_t)((m.rows + 63) & ~63L), 256)), dim3(256), 0, s, dL_dout, m.out_dim, m.rows, (_Float16*)dout16, absmax);
        } else {
            const bool hand = big_rows && gp_debug_get(9) == 0;
            hipLaunchKernelGGL(hand ? gp_mlp16_bwd_data4_bf16_hand_kernel : big_rows ? gp_mlp16_bwd_data4_bf16_kernel : gp_mlp16_bwd_data_bf16_kernel, grid, dim3(M16_THREADS), 0, s, m, masks, dL_dout, (void*)dz, dL_dfeature, dL_dxyz, absmax);
            hipLaunchKernelGGL((gp_mlp16_pack_dout_kernel<__bf16>), dim3(gp_blocks((size_t)((m.rows + 63) & ~63L), 256)), dim3(256), 0, s, dL_dout, m.out_dim, m.rows, (__bf16*)dout16, absmax);
        }
        GP_LAUNCH_CHECK();
    }
    long nrb_l = (m.rows + 4095) / 4096;       // 4096-row slabs: 2 x 256 x 4096 x 2 B = 4 MB per layer, shared by 16 workgroups
    {   // small row counts: 256-row slabs keep the grid at a few hundred workgroups (two 4096-row slabs at 8k rows left
        // 32 workgroups walking 256 k-steps each: latency-bound, 5x the forward's time)
        long small = (m.rows + 255) / 256;
        if (small > 128) small = 128;
        if (nrb_l < small) nrb_l = small;
    }
    if (nrb_l < 1) nrb_l = 1;
    const long rpb = ((m.rows + nrb_l - 1) / nrb_l + 63) & ~63L;
    const unsigned nrb = (unsigned)((m.rows + rpb - 1) / rpb);
    GpProfScope _pw("mlp16_bwd_weight", s);
    // operand tensors of layer l: dZ_l (n_out x rows) and H_l (n_in x rows); nf = features read, ks = features per row block
    const size_t le = t16_elems(NS * 256, m.rows) * 2;               // bytes per 256-feature layer tensor
    struct Opnd { const char* z; const char* h; int nf_z, nf_h, n_out, n_in; };
    auto opnd = [&](int l) {
        Opnd o;
        o.z = l < 4 ? dz + (size_t)l * le : dout16;
        o.h = l == 0 ? (const char*)saved_xT : (const char*)saved_hT + (size_t)(l - 1) * le;
        o.nf_z = l < 4 ? 256 : 16; o.nf_h = l == 0 ? m.in_pad : 256;
        o.n_out = l < 4 ? 256 : m.out_dim; o.n_in = l == 0 ? m.in_dim : 256;
        return o;
    };
    // split mode: dW = Zh Hh^T + 2^-11 (Zh Hl'^T + Zl' Hh^T); the lo' half of a tensor starts nf features into each row block
    const int n_terms = split ? 3 : 1;
    auto term = [&](const Opnd& o, int t, float* dw, float* db) {
        W16Job jb;
        jb.z = o.z + (t == 2 ? (size_t)o.nf_z * T16_BLK * 2 : 0);
        jb.h = o.h + (t == 1 ? (size_t)o.nf_h * T16_BLK * 2 : 0);
        jb.dw = dw; jb.db = t == 1 ? nullptr : db; jb.mul = t == 0 ? 1.f : SP_LO_INV;
        return jb;
    };
    if (m.rows >= 4096) {       // LDS-staged kernel: grid = (row slabs, jobs), three launches cover the five layers
        const long n_kb = (m.rows + 63) / 64 * (64 / T16_BLK);   // 16-row blocks incl. the zero padding to 64 rows
        // 4096-row slabs: every slab ends in 64 k atomic adds per (layer, term), and below that size they show (round 5, tools/probe/
        // mlp16_wgrad_slabs.py at 200 k rows: 1024-row slabs 0.72 ms, 4096-row slabs 0.59; 1 M rows sat at the 256-slab cap = 4096 rows already)
        const int slab_kb = gp_debug_get(10) > 0 ? gp_debug_get(10) : 256;    // (gp_debug_option(10, n): n 16-row blocks per slab)
        long nslab = n_kb / slab_kb;
        const long few = n_kb / 16 < 32 ? n_kb / 16 : 32;        // ... but at least 32 slabs of >= 256 rows at small row counts
        if (nslab < few) nslab = few;
        if (nslab > 256) nslab = 256;
        if (nslab < 1) nslab = 1;
        const long kbs = ((n_kb + nslab - 1) / nslab + W16_DEPTH - 1) / W16_DEPTH * W16_DEPTH;
        const unsigned gx = (unsigned)((n_kb + kbs - 1) / kbs);
        W16Jobs mid, first, last;
        for (int t = 0; t < n_terms; ++t) {
            for (int l = 1; l <= 3; ++l) mid.j[(l - 1) * n_terms + t] = term(opnd(l), t, g->dw[l], g->db[l]);
            first.j[t] = term(opnd(0), t, g->dw[0], g->db[0]);
            last.j[t] = term(opnd(4), t, g->dw[4], g->db[4]);
        }
        const int xt = (split && gp_debug_get(3) == 0) ? n_terms : 0;      // (gp_debug_option(3, 1): the round-2 2-D grid, for A/B)
        const W16Shape sh_mid{256, 256, 256, 256, 256, NS * 256, NS * 256, split, xt, 3 * n_terms, (int)gx},
            sh_first{256, m.in_pad, 256, m.in_dim, m.in_dim, NS * 256, NS * m.in_pad, split, xt, n_terms, (int)gx},
            sh_last{16, 256, m.out_dim, 256, 256, NS * 16, NS * 256, split, xt, n_terms, (int)gx};
        // 1-D grids of the XCD-aware form: units = layers x slabs, rounded up to 8, x terms
        auto grid_of = [&](int nlay) { return xt ? dim3((unsigned)(((size_t)nlay * gx + 7) / 8 * 8 * n_terms)) : dim3(nlay * n_terms, gx); };
        if (f16) {
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<_Float16, 4, 2, 4>), grid_of(3), dim3(W16_THREADS), 0, s, mid, sh_mid, n_kb, kbs, absmax);
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<_Float16, 2, 2, 2>), grid_of(1), dim3(W16_THREADS), 0, s, first, sh_first, n_kb, kbs, absmax);
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<_Float16, 1, 1, 8>), grid_of(1), dim3(W16_THREADS), 0, s, last, sh_last, n_kb, kbs, absmax);
        } else {
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<__bf16, 4, 2, 4>), dim3(3, gx), dim3(W16_THREADS), 0, s, mid, sh_mid, n_kb, kbs, absmax);
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<__bf16, 2, 2, 2>), dim3(1, gx), dim3(W16_THREADS), 0, s, first, sh_first, n_kb, kbs, absmax);
            hipLaunchKernelGGL((gp_mlp16_bwd_weight_lds_kernel<__bf16, 1, 1, 8>), dim3(1, gx), dim3(W16_THREADS), 0, s, last, sh_last, n_kb, kbs, absmax);
        }
        GP_LAUNCH_CHECK();
        return 0;
    }
    for (int l = 0; l < 5; ++l) {
        const Opnd o = opnd(l);
        const dim3 grid((unsigned)((o.n_in + 63) / 64) * (unsigned)((o.n_out + 63) / 64), nrb);
        for (int t = 0; t < n_terms; ++t) {      // consecutive launches on one stream: the terms never add to dW concurrently
            const W16Job jb = term(o, t, g->dw[l], g->db[l]);
            if (f16) hipLaunchKernelGGL(gp_mlp16_bwd_weight_f16_kernel, grid, dim3(M16_THREADS), 0, s, jb.z, o.n_out, jb.h, o.n_in, NS * o.nf_z, NS * o.nf_h, m.rows, rpb, jb.dw, o.n_in, jb.db, absmax, jb.mul);
            else hipLaunchKernelGGL(gp_mlp16_bwd_weight_bf16_kernel, grid, dim3(M16_THREADS), 0, s, jb.z, o.n_out, jb.h, o.n_in, NS * o.nf_z, NS * o.nf_h, m.rows, rpb, jb.dw, o.n_in, jb.db, absmax, jb.mul);
        }
        GP_LAUNCH_CHECK();
    }
    return 0;
}
