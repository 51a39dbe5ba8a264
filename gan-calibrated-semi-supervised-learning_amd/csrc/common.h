// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the cGAN WGAN-GP step.
// Wave = 64 lanes everywhere in this tree; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GCSSL_OK 0
#define GCSSL_EBADSHAPE (-1)
#define GCSSL_EBADDTYPE (-2)
#define GCSSL_EALIGN (-3)
#define GCSSL_ENULL (-4)

#define GCSSL_F32 0
#define GCSSL_BF16 1
#define GCSSL_F16 2      // IEEE half operands (v_mfma_f32_32x32x16_f16: the bf16 rate, 3 more mantissa bits), fp32 accumulate
// Split-precision conv modes: tensors are fp32 in memory exactly as with GCSSL_F32 (every non-conv entry point is called with
// GCSSL_F32); only the conv contractions differ -- operands split into hi + lo 16-bit halves on their way into LDS, three
// 16-bit MFMAs per K step (hi*hi + lo*hi + hi*lo), fp32 accumulate.  Accepted by the gcssl_conv4x4s2_{fwd,dgrad,wgrad}[_splits]
// and gcssl_conv3x3_{fwd,wgrad} entry points only.
#define GCSSL_F32_F16X3 3    // fp16 halves: 22 mantissa bits per operand (operands must sit in fp16's range: static loss scale)
#define GCSSL_F32_BF16X3 4   // bf16 halves: 16 mantissa bits per operand, fp32's exponent range

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Run `...` with T bound to the element type of compute dtype `dt` (float / bf16_t / f16_t); the caller has validated dt.
#define GCSSL_DISPATCH(dt, ...)                                              \
    do {                                                                     \
        if ((dt) == GCSSL_F32) { typedef float T; __VA_ARGS__; }             \
        else if ((dt) == GCSSL_F16) { typedef f16_t T; __VA_ARGS__; }        \
        else { typedef bf16_t T; __VA_ARGS__; }                              \
    } while (0)
#define GCSSL_DISPATCH16(dt, ...)    /* 16-bit compute dtypes only (the caller has excluded GCSSL_F32) */ \
    do {                                                                     \
        if ((dt) == GCSSL_F16) { typedef f16_t T; __VA_ARGS__; }             \
        else { typedef bf16_t T; __VA_ARGS__; }                              \
    } while (0)
static inline bool gcssl_bad_dtype(int dt) { return dt != GCSSL_F32 && dt != GCSSL_BF16 && dt != GCSSL_F16; }
// conv entry points: + the split-precision modes.  T = storage element type, MM = MFMA mode (0 native, 1 fp16x3, 2 bf16x3)
#define GCSSL_DISPATCH_CONV(dt, ...)                                                            \
    do {                                                                                        \
        if ((dt) == GCSSL_F32) { typedef float T; constexpr int MM = 0; (void)MM; __VA_ARGS__; }            \
        else if ((dt) == GCSSL_F32_F16X3) { typedef float T; constexpr int MM = 1; (void)MM; __VA_ARGS__; } \
        else if ((dt) == GCSSL_F32_BF16X3) { typedef float T; constexpr int MM = 2; (void)MM; __VA_ARGS__; }\
        else if ((dt) == GCSSL_F16) { typedef f16_t T; constexpr int MM = 0; (void)MM; __VA_ARGS__; }       \
        else { typedef bf16_t T; constexpr int MM = 0; (void)MM; __VA_ARGS__; }                             \
    } while (0)
static inline bool gcssl_bad_conv_dtype(int dt) { return gcssl_bad_dtype(dt) && dt != GCSSL_F32_F16X3 && dt != GCSSL_F32_BF16X3; }
static inline bool gcssl_f32_storage(int dt) { return dt == GCSSL_F32 || dt == GCSSL_F32_F16X3 || dt == GCSSL_F32_BF16X3; }

static inline int gcssl_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GCSSL_OK : (int)e;
}
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- element traits: T is the storage/operand type of activations (float or bf16)
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int KV = 4;                       // elements per 16-byte vector
    __device__ static float ld(const float* p) { return *p; }
    __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int KV = 8;
    __device__ static float ld(const bf16_t* p) { return (float)*p; }
    __device__ static void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};
// fp16 stores SATURATE at +-65504 instead of overflowing to inf: the critic's loss-scaled gradient tensors have a heavy tail
// (a 2x2 InstanceNorm with a tiny variance multiplies by rstd up to 316; measured peaks reach a quarter of the ceiling once
// in a few thousand iterations), and one inf in a weight-gradient operand turns the whole update into NaN, while a clipped
// outlier perturbs a gradient that is norm-clipped to 1 anyway.  NaN is PRESERVED (fmaxf alone would turn it into -65504):
// a NaN gradient has to reach the optimiser's norm and the reference's NaN/Inf stop (cgan/cgan_train_enhanced.py:473-478),
// not become a large finite update.  Clipped gradient stores are COUNTED (sat_hits / sat_commit below): the engine keeps the
// count on the device, bench.py reports it and the tests assert it is zero.
__device__ __forceinline__ f16_t f32_to_f16_sat(float v) { return v != v ? (f16_t)v : (f16_t)fminf(fmaxf(v, -65504.f), 65504.f); }
template <> struct Elem<f16_t> {
    static constexpr int KV = 8;
    __device__ static float ld(const f16_t* p) { return (float)*p; }
    __device__ static void st(f16_t* p, float v) { *p = f32_to_f16_sat(v); }
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    bf16_t h = (bf16_t)f;                              // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
// 16-bit element <-> its bit pattern in the low half of a word, for either 16-bit operand type
template <typename T> struct Bits16;
template <> struct Bits16<bf16_t> {
    __device__ static __forceinline__ uint32_t enc(float f) { return f32_to_bf16_bits(f); }
    __device__ static __forceinline__ float dec(uint32_t b) { return bf16_bits_to_f32(b & 0xFFFFu); }
};
template <> struct Bits16<f16_t> {
    __device__ static __forceinline__ uint32_t enc(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, f32_to_f16_sat(f)); }
    __device__ static __forceinline__ float dec(uint32_t b) { return (float)__builtin_bit_cast(f16_t, (uint16_t)(b & 0xFFFFu)); }
};
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return Bits16<T>::enc(lo) | (Bits16<T>::enc(hi) << 16); }

// 16-byte vector of elements <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    float4 v;
    __device__ static Vec16 zero() { Vec16 r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
    __device__ static Vec16 load(const float* p) { Vec16 r; r.v = *reinterpret_cast<const float4*>(p); return r; }
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
    __device__ void set(int i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else if (i == 2) v.z = f; else v.w = f; }
};
#define GCSSL_VEC16_16BIT(TT)                                                                                         \
template <> struct Vec16<TT> {                                                                                        \
    uint4 v;                                                                                                          \
    __device__ static Vec16 zero() { Vec16 r; r.v = make_uint4(0, 0, 0, 0); return r; }                               \
    __device__ static Vec16 load(const TT* p) { Vec16 r; r.v = *reinterpret_cast<const uint4*>(p); return r; }        \
    __device__ void store(TT* p) const { *reinterpret_cast<uint4*>(p) = v; }                                          \
    __device__ uint32_t word(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }                \
    __device__ uint32_t bits(int i) const { uint32_t w = word(i >> 1); return (i & 1) ? (w >> 16) : (w & 0xFFFFu); }  \
    __device__ float get(int i) const { return Bits16<TT>::dec(bits(i)); }                                            \
    __device__ void setword(int i, uint32_t w) { if (i == 0) v.x = w; else if (i == 1) v.y = w; else if (i == 2) v.z = w; else v.w = w; } \
    __device__ void set2(int pair, float lo, float hi) { setword(pair, pack2<TT>(lo, hi)); }                          \
};
GCSSL_VEC16_16BIT(bf16_t)
GCSSL_VEC16_16BIT(f16_t)
#undef GCSSL_VEC16_16BIT

// ---- saturation count of 16-bit GRADIENT stores (fp16 only: bf16 and fp32 have fp32's exponent range).  A thread counts the
// values it is about to store whose magnitude exceeds fp16's largest finite number and adds a non-zero count to the
// caller's device counter (nullable): clipping is rare by construction (static loss scale), so the atomic almost never runs.
template <typename T> struct IsF16 { static constexpr bool v = false; };
template <> struct IsF16<f16_t> { static constexpr bool v = true; };
template <typename T> __device__ __forceinline__ int sat_hit(float v) { return (IsF16<T>::v && fabsf(v) > 65504.f) ? 1 : 0; }
template <typename T, int NV> __device__ __forceinline__ int sat_hits(const float (&v)[NV]) {
    int n = 0;
    if constexpr (IsF16<T>::v) {
#pragma unroll
        for (int i = 0; i < NV; ++i) n += fabsf(v[i]) > 65504.f ? 1 : 0;
    }
    return n;
}
__device__ __forceinline__ void sat_commit(unsigned* sat, int n) { if (sat && n) atomicAdd(sat, (unsigned)n); }

// ---- which kernel did the last conv dispatcher launch?  (bench.py's roofline names the rocprofv3 symbol of its dominant
// launch: the dispatchers pick a template instantiation per shape, so the host records the expression it launched.)
extern const char* g_gcssl_last_kernel;
extern long g_gcssl_last_grid;            // ... and its grid size in workgroups (tools/prof_labels.py joins a label to its rocprofv3 launches by both)
#define GCSSL_LAUNCH(kern, grid, ...) do { g_gcssl_last_kernel = #kern; const dim3 gcssl_g_ = (grid);          \
        g_gcssl_last_grid = (long)gcssl_g_.x * gcssl_g_.y * gcssl_g_.z; hipLaunchKernelGGL(kern, gcssl_g_, __VA_ARGS__); } while (0)

// ---- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over a block of NW waves; result valid in every thread. `sm` needs NW floats.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += sm[i];
    return r;
}

// ---- raw buffer access.  A raw buffer load whose offset lies outside the descriptor's range returns 0 in hardware and a
// store there is dropped, so validity is a v_cndmask on the OFFSET instead of an exec-mask branch around the access: every
// lane always issues, and the compiler can batch the loads of a whole tile / row set behind one s_waitcnt.  A descriptor of
// 0 bytes makes an absent optional operand read as zeros without touching memory.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float lrelu_f(float x) { return x > 0.f ? x : 0.2f * x; }

// ---- zero fills as KERNELS.  hipMemsetAsync / hipMemset2DAsync must not be used on a path that may be stream-captured: on
// ROCm 7.2 the memset node of the instantiated graph replays with a garbage fill pattern once the host memory the call's
// parameters lived in has been reused (measured: the InstanceNorm backward's scratch came back filled with 0x61 / 0x6f
// bytes = 2.6e20 / 7.4e28 on the second replay, depending on what Python had allocated in between; tools/archive/replay_diag9.py).
static __global__ __launch_bounds__(256) void gcssl_zero_kernel(float* __restrict__ p, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = 0.f;
}
// rows x cols floats with a row pitch of ld floats
static __global__ __launch_bounds__(256) void gcssl_zero2d_kernel(float* __restrict__ p, size_t ld, int cols, size_t rows) {
    const size_t total = rows * (size_t)cols, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
static inline void gcssl_zero_async(float* p, size_t n, hipStream_t st) {
    if ((((uintptr_t)p) & 15) != 0) {                                            // float4 stores need 16-byte alignment
        size_t b = (n + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1;
        hipLaunchKernelGGL(gcssl_zero2d_kernel, dim3((unsigned)b), dim3(256), 0, st, p, n, (int)(n < 0x7fffffff ? n : 0x7fffffff), (size_t)1);
        return;
    }
    size_t blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gcssl_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
}
static inline void gcssl_zero2d_async(float* p, size_t ld, int cols, size_t rows, hipStream_t st) {
    if (ld == (size_t)cols && (((uintptr_t)p) & 15) == 0) { gcssl_zero_async(p, rows * cols, st); return; }
    size_t blocks = (rows * cols + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gcssl_zero2d_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, ld, cols, rows);
}

// ---- spectral-norm batch arguments (misc.hip's chain; igemm.hip's weight re-pack launch can carry the chain's closing step)
struct SnLayer { const float* w; float* u; float* v; float* t; float* s; int rows, cols; };
struct SnBatch { SnLayer l[4]; int nl; float* sigma; float* isig; float* u_hist; float* v_hist; int hist_stride_u, hist_stride_v; int slot, nslots;
                 int par;            // which half of a layer's 2 x cols scratch `t` this iteration accumulates into (chain position & 1)
                 int fin_prev;       // sn_wtu: the PREVIOUS iteration of the chain has not been closed -- u is still s (= W v), see misc.hip
                 float* zero; long nzero; };   // sn_finish: an extra buffer to clear (the engine's scalar / replica block), nullable
// the closing step of a chain for layer `layer` (one 256-thread workgroup of `nblk` that share the extra fill):
// u = s / max(|s|, eps), sigma = u . s, t zeroed again for the next iteration
__device__ __forceinline__ void sn_finish_body(const SnBatch& b, int layer, int nblk) {
    const SnLayer L = b.l[layer];
    __shared__ float red[4];
    float q2 = 0.f;
    for (int r = threadIdx.x; r < L.rows; r += 256) { const float sv = L.s[r]; q2 += sv * sv; }
    const float s2 = block_sum<4>(q2, red);
    const float uinv = 1.f / fmaxf(sqrtf(s2), 1e-12f);
    float* uh = b.u_hist + ((size_t)layer * b.nslots + b.slot) * b.hist_stride_u;
    for (int r = threadIdx.x; r < L.rows; r += 256) { const float uu = L.s[r] * uinv; L.u[r] = uu; uh[r] = uu; }
    float* tcur = L.t + (b.par ? L.cols : 0);
    for (int c = threadIdx.x; c < L.cols; c += 256) tcur[c] = 0.f;            // (the other half was cleared by sn_wv_kernel)
    if (threadIdx.x == 0) {
        const float sg = s2 * uinv;
        b.sigma[layer * b.nslots + b.slot] = sg;
        b.isig[layer * b.nslots + b.slot] = 1.f / sg;
    }
    // an extra fill for the caller (the engine's scalar / striped-sum block, cleared once per critic step): the workgroups share it
    for (long i = (long)layer * 256 + threadIdx.x; i < b.nzero; i += (long)nblk * 256) b.zero[i] = 0.f;
}
extern "C" __attribute__((visibility("hidden"))) int gcssl_take_pending_sn(SnBatch* out);
// ---- the head conv's data gradient with per-group CONSTANT dout (misc.hip c5_dgrad_const_kernel) as a rider of the weight
// re-pack launch: dx[n, p, :] = g(n) * tab[p][:], tab[p][c] = sum of w[c][tap] over the taps through which pixel p reaches an
// output.  It reads the RAW weight [512][16] (the packed copy is being written by the same launch) and writes fp32.
struct C5DgradRider { float g[4]; int group_n; const float* w; float* dx; int lddx, N, Hi, Wi, per, nblk; };
__device__ __forceinline__ void c5_dgrad_rider_body(const C5DgradRider& r, int blk) {
    constexpr int C = 512;
    const int Ho = r.Hi - 1, Wo = r.Wi - 1, P = r.Hi * r.Wi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane * 8;
    const int nb = blk * r.per, ne = min(r.N, nb + r.per);
    for (int p = wave; p < P; p += 4) {
        const int iy = p / r.Wi, ix = p - iy * r.Wi;
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 4; ++ky) {
            if ((unsigned)(iy + 1 - ky) >= (unsigned)Ho) continue;
            for (int kx = 0; kx < 4; ++kx) {
                if ((unsigned)(ix + 1 - kx) >= (unsigned)Wo) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] += r.w[(size_t)(c + j) * 16 + ky * 4 + kx];
            }
        }
        for (int n = nb; n < ne; ++n) {
            const int g = n / r.group_n;
            const float gc = g == 0 ? r.g[0] : (g == 1 ? r.g[1] : (g == 2 ? r.g[2] : r.g[3]));
            float* o = r.dx + ((size_t)n * P + p) * r.lddx + c;
            reinterpret_cast<float4*>(o)[0] = make_float4(gc * t[0], gc * t[1], gc * t[2], gc * t[3]);
            reinterpret_cast<float4*>(o)[1] = make_float4(gc * t[4], gc * t[5], gc * t[6], gc * t[7]);
        }
    }
}
extern "C" __attribute__((visibility("hidden"))) int gcssl_take_pending_c5(C5DgradRider* out);   // misc.hip, as gcssl_take_pending_sn
   // misc.hip: 1 and *out = a closing step left pending by gcssl_sn_defer_finish, else 0
