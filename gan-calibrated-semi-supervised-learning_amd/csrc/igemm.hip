// Implicit-GEMM 4x4 stride-2 pad-1 convolution kernels for gfx950: forward, data-gradient
// (= transposed convolution, sub-pixel form: no zero insertion, no col2im atomics) and
// weight-gradient.  No im2col buffer is ever materialised: the loaders gather 16-byte channel
// vectors of the NHWC activations straight into LDS tiles, and the contraction runs on MFMA
// (v_mfma_f32_32x32x2_f32 in the exact-fp32 parity mode, v_mfma_f32_32x32x16_bf16 in bf16 mode).
//
// Replaces (reference has no custom op; these are the stock ops of its hot path):
//   nn.Conv2d(k4,s2,p1) forward/backward        cgan/models.py:57,236   (G.down*, D.c1-c4)
//   nn.ConvTranspose2d(k4,s2,p1) fwd/backward   cgan/models.py:72,113   (G.up*)
//   and the second-order forms needed by create_graph=True at cgan/losses.py:213-220.
//
// Layouts (SURVEY.md/DESIGN.md): activations NHWC with an explicit pixel stride `ld` (so concat
// buffers are written/read in place); weights packed Wf[Cout][16][Cin] (fwd) and Wt[Cin][16][Cout]
// (dgrad); weight gradients land in fp32 split-K slabs [split][Cout][16][Cin] that
// gcssl_wgrad_reduce sums into the PyTorch-layout gradient.
#include "common.h"
#include <cstring>
#include <type_traits>
#include <cstdlib>

namespace {

// K elements per LDS tile: chosen so that one gathered row is a full 128-byte line in either dtype (64 B half-line
// gathers measured at the per-CU L2 gather ceiling, MI355X_MICROARCH.md "Indexed rows: gather into LDS").
template <typename T> struct BKOf { static constexpr int v = 32; };       // fp32: 32 x 4 B
template <> struct BKOf<bf16_t> { static constexpr int v = 64; };          // bf16: 64 x 2 B
template <> struct BKOf<f16_t> { static constexpr int v = 64; };           // fp16: 64 x 2 B
template <typename T> struct Is16 { static constexpr bool v = !std::is_same<T, float>::value; };   // a 16-bit MFMA operand type
template <typename T> struct Op16 { typedef T type; };                     // (float -> bf16_t: lets never-taken 16-bit branches of
template <> struct Op16<float> { typedef bf16_t type; };                   //  the float instantiations compile)
template <typename T> struct Frag16;                                       // the 32x32x16 MFMA operand of a 16-bit type
template <> struct Frag16<bf16_t> { typedef bf16x8 type; };
template <> struct Frag16<f16_t> { typedef f16x8 type; };
constexpr int NT = 256;      // threads per workgroup: 4 waves as 2 (M) x 2 (N)

// ------------------------------------------------------------------------------------------
// LDS tiles.  KMajor: [row][k], filled by k-contiguous vectors (fwd, dgrad).
//             MMajor: [k][row], filled by row-contiguous vectors (wgrad; read transposed).
// frag(row0, ks, lane) returns the MFMA operand of lane `lane` for the 32 rows row0.. and
// k-step ks (2 k per step for f32 32x32x2, 16 per step for bf16 32x32x16).
// ------------------------------------------------------------------------------------------
template <typename T, int ROWS> struct KMajor;
template <int ROWS> struct KMajor<float, ROWS> {
    static constexpr int BK = BKOf<float>::v;
    static constexpr int STRIDE = BK + 1;                     // conflict-free b32 fragment reads
    static constexpr bool SPLIT = false;
    static constexpr int KSTEPS = BK / 2;
    typedef float Frag;
    float d[ROWS * STRIDE];
    __device__ void store_vec(int row, int chunk, const Vec16<float>& v) {
        float* p = d + row * STRIDE + chunk * 4;
        p[0] = v.v.x; p[1] = v.v.y; p[2] = v.v.z; p[3] = v.v.w;
    }
    __device__ Frag frag(int row0, int ks, int lane) const {
        return d[(row0 + (lane & 31)) * STRIDE + ks * 2 + (lane >> 5)];
    }
};
template <typename T, int ROWS> struct KMajor16 {
    static constexpr int BK = 64;
    static constexpr bool SPLIT = false;
    static constexpr int KSTEPS = BK / 16;
    typedef typename Frag16<T>::type Frag;
    uint4 d[ROWS * 8];                                        // 128-byte rows of 8 16-byte chunks, XOR-swizzled:
    // a ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31}) reads one logical chunk of 16 rows; with
    // chunk' = chunk ^ ((row>>1)&7) those land on 16 distinct 16-B slots of the 256-B bank row -> conflict-free.
    __device__ static int swz(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }
    __device__ void store_vec(int row, int chunk, const Vec16<T>& v) { d[swz(row, chunk)] = v.v; }
    __device__ Frag frag(int row0, int ks, int lane) const {
        return __builtin_bit_cast(Frag, d[swz(row0 + (lane & 31), ks * 2 + (lane >> 5))]);
    }
};
template <int ROWS> struct KMajor<bf16_t, ROWS> : KMajor16<bf16_t, ROWS> {};
template <int ROWS> struct KMajor<f16_t, ROWS> : KMajor16<f16_t, ROWS> {};

template <typename T, int ROWS> struct MMajor;
template <int ROWS> struct MMajor<float, ROWS> {
    static constexpr int BK = BKOf<float>::v;
    static constexpr bool SPLIT = false;
    static constexpr int KSTEPS = BK / 2;
    typedef float Frag;
    float d[BK * ROWS];
    __device__ void store_vec(int k, int chunk, const Vec16<float>& v) {
        *reinterpret_cast<float4*>(d + k * ROWS + chunk * 4) = v.v;
    }
    __device__ Frag frag(int row0, int ks, int lane) const {
        return d[(ks * 2 + (lane >> 5)) * ROWS + row0 + (lane & 31)];
    }
};
template <typename T, int ROWS> struct MMajor16 {
    static constexpr int BK = 64;
    static constexpr bool SPLIT = false;
    static constexpr int KSTEPS = BK / 16;
    static constexpr int STRIDE = ROWS * 2 + 64;              // bytes; +64 B keeps the 4 rows of a tr block on distinct banks
    typedef typename Frag16<T>::type Frag;
    __attribute__((aligned(16))) unsigned char d[BK * STRIDE];
    __device__ void store_vec(int k, int chunk, const Vec16<T>& v) {
        *reinterpret_cast<uint4*>(d + k * STRIDE + chunk * 16) = v.v;
    }
    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block is delivered column-major:
    // lane 4q+p supplies the address of k-row q, rows 4p..4p+3; lane i receives row i, k = q in element q.
    __device__ Frag frag(int row0, int ks, int lane) const {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_ptr;
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g >> 1;
        const int kb = ks * 16 + 8 * h + q;
        const unsigned char* a0 = d + kb * STRIDE + (row0 + 16 * (g & 1) + 4 * p) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * STRIDE));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(Frag, r);
    }
};
template <int ROWS> struct MMajor<bf16_t, ROWS> : MMajor16<bf16_t, ROWS> {};
template <int ROWS> struct MMajor<f16_t, ROWS> : MMajor16<f16_t, ROWS> {};

// ------------------------------------------------------------------------------------------
// SPLIT-PRECISION tiles (MM = 1: fp16 hi + lo, MM = 2: bf16 hi + lo; the "fp16x3" / "bf16x3" modes).  Tensors stay fp32 in
// HBM exactly as in the fp32 parity mode; a loaded 4-float vector is split ONCE, on its way into LDS, into
//   hi = rne16(x),  lo = rne16(x - hi)
// and the 32 real K elements of a tile row become one 128-byte row [hi k0..31 | lo k0..31] -- the 16-bit KMajor16 / MMajor16
// layouts with four physical 16-deep K steps (0, 1 = hi; 2, 3 = lo).  The contraction then runs on the 2.5-PFLOP/s 16-bit
// MFMA pipe as THREE products per real K step, A_hi B_hi + A_lo B_hi + A_hi B_lo (A_lo B_lo is below fp32's own rounding),
// accumulated in fp32: 22 (fp16) / 16 (bf16) mantissa bits per operand instead of 11 / 8, against v_mfma_f32_32x32x2_f32's
// 157 TFLOP/s for the exact form.  SURVEY.md 7 "hard part 2" names this ("bf16x3 split"); reference arithmetic: pure fp32,
// cgan/models.py:222-258, cgan/losses.py:185-233.
// ------------------------------------------------------------------------------------------
template <int MM> struct SplitH;
template <> struct SplitH<1> { typedef f16_t type; };
template <> struct SplitH<2> { typedef bf16_t type; };
// 4 floats -> 4 hi + 4 lo halves, 5 vector instructions per PAIR: v_cvt_pk_{f16,bf16}_f32 (RNE), two widenings, one
// v_pk_add_f32, one v_cvt_pk.  No clamp: an operand beyond fp16's range becomes inf, its residual -inf, the product NaN -- loud
// (the engine's finite checks), where a saturated value would be silently wrong; the static scales keep operands in range.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename H> struct Half2;
template <> struct Half2<f16_t> {
    typedef __attribute__((ext_vector_type(2))) _Float16 V;
    __device__ static __forceinline__ f32x2 widen(V h) { return __builtin_convertvector(h, f32x2); }
};
template <> struct Half2<bf16_t> {
    typedef __attribute__((ext_vector_type(2))) __bf16 V;
    __device__ static __forceinline__ f32x2 widen(V h) {
        const unsigned u = __builtin_bit_cast(unsigned, h);
        f32x2 r = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xFFFF0000u)};
        return r;
    }
};
template <typename H> __device__ __forceinline__ void split2(f32x2 x, unsigned& hi, unsigned& lo) {
    typedef typename Half2<H>::V V;
    const V h = __builtin_convertvector(x, V);
    const V l = __builtin_convertvector(x - Half2<H>::widen(h), V);
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}
template <typename H> __device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    split2<H>(a, hi.x, lo.x);
    split2<H>(b, hi.y, lo.y);
}
template <typename H, int ROWS> struct KMajorSplit {
    static constexpr int BK = 32;                             // REAL K elements per tile (fp32 in memory)
    static constexpr int KSTEPS = 4;                          // physical 16-deep steps: hi, hi, lo, lo
    static constexpr bool SPLIT = true;
    typedef typename Frag16<H>::type Frag;
    uint4 d[ROWS * 8];
    __device__ static int swz(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }
    // chunk: which 4-float vector of the row's 32 (0..7)
    __device__ void store_vec(int row, int chunk, const Vec16<float>& v) {
        uint2 hi, lo;
        split4<H>(v.v, hi, lo);
        uint2* b = reinterpret_cast<uint2*>(d);
        b[swz(row, chunk >> 1) * 2 + (chunk & 1)] = hi;
        b[swz(row, 4 + (chunk >> 1)) * 2 + (chunk & 1)] = lo;
    }
    __device__ Frag frag(int row0, int ks, int lane) const {
        return __builtin_bit_cast(Frag, d[swz(row0 + (lane & 31), ks * 2 + (lane >> 5))]);
    }
};
template <typename H, int ROWS> struct MMajorSplit {
    static constexpr int BK = 32;
    static constexpr int KSTEPS = 4;
    static constexpr bool SPLIT = true;
    static constexpr int STRIDE = ROWS * 2 + 64;              // bytes (as MMajor16)
    typedef typename Frag16<H>::type Frag;
    __attribute__((aligned(16))) unsigned char d[64 * STRIDE];   // k-rows 0..31: hi, 32..63: lo
    // chunk: which 4-row vector of k-row k (rows 4 chunk .. 4 chunk + 3)
    __device__ void store_vec(int k, int chunk, const Vec16<float>& v) {
        uint2 hi, lo;
        split4<H>(v.v, hi, lo);
        *reinterpret_cast<uint2*>(d + k * STRIDE + chunk * 8) = hi;
        *reinterpret_cast<uint2*>(d + (k + 32) * STRIDE + chunk * 8) = lo;
    }
    __device__ Frag frag(int row0, int ks, int lane) const {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_ptr;
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g >> 1;
        const int kb = ks * 16 + 8 * h + q;
        const unsigned char* a0 = d + kb * STRIDE + (row0 + 16 * (g & 1) + 4 * p) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * STRIDE));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(Frag, r);
    }
};
// tile types of the register-staged kernels by MFMA mode
template <typename T, int ROWS, int MM> struct KTile { typedef KMajor<T, ROWS> type; };
template <int ROWS> struct KTile<float, ROWS, 1> { typedef KMajorSplit<f16_t, ROWS> type; };
template <int ROWS> struct KTile<float, ROWS, 2> { typedef KMajorSplit<bf16_t, ROWS> type; };
template <typename T, int ROWS, int MM> struct MTile { typedef MMajor<T, ROWS> type; };
template <int ROWS> struct MTile<float, ROWS, 1> { typedef MMajorSplit<f16_t, ROWS> type; };
template <int ROWS> struct MTile<float, ROWS, 2> { typedef MMajorSplit<bf16_t, ROWS> type; };
// Branch-free gather loads.  hipcc turns `ok ? load(p) : 0` into an exec-mask branch per load (each with its own
// vmcnt drain), which serialises the whole tile fetch; a raw buffer load with an out-of-range offset returns 0 in
// hardware instead, so every lane always issues the load and validity is a single v_cndmask on the offset.
// (OOB / make_rsrc: common.h)
template <typename T> __device__ __forceinline__ Vec16<T> bload(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ Vec16<float> bload<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Vec16<float> v; v.v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); return v;
}
template <> __device__ __forceinline__ Vec16<bf16_t> bload<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Vec16<bf16_t> v; v.v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); return v;
}
template <> __device__ __forceinline__ Vec16<f16_t> bload<f16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Vec16<f16_t> v; v.v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); return v;
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// one BK slab: acc[i][j] += A(rows wm0+32i..) x B(rows wn0+32j..)^T
// split tiles (KMajorSplit / MMajorSplit): per real 16-deep K step s, acc += A_lo[s] B_hi[s] + A_hi[s] B_lo[s] + A_hi[s] B_hi[s]
// (physical steps 0, 1 = hi, 2, 3 = lo; every term is an fp32 accumulate, A_lo B_lo is dropped: below fp32's own rounding)
template <int TM, int TN, class TA, class TB>
__device__ __forceinline__ void mma_slab_split(const TA& As, const TB& Bs, int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
    typename TA::Frag a[4][TM];
    typename TB::Frag b[4][TN];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[ks][i] = As.frag(wm0 + 32 * i, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[ks][j] = Bs.frag(wn0 + 32 * j, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = mfma(a[ks + 2][i], b[ks][j], acc[i][j]);
                acc[i][j] = mfma(a[ks][i], b[ks + 2][j], acc[i][j]);
                acc[i][j] = mfma(a[ks][i], b[ks][j], acc[i][j]);
            }
}
template <int TM, int TN, class TA, class TB>
__device__ __forceinline__ void mma_slab(const TA& As, const TB& Bs, int wm0, int wn0, int lane,
                                         f32x16 (&acc)[TM][TN]) {
    if constexpr (TA::SPLIT) { mma_slab_split<TM, TN>(As, Bs, wm0, wn0, lane, acc); return; }
    if constexpr (sizeof(typename TA::Frag) == 16 && TA::KSTEPS * (TM + TN) <= 16) {
        // 16-bit operands: every fragment read of the slab goes out first, then the MFMAs run behind counted lgkmcnt waits
        // (left to itself hipcc emits reads, lgkmcnt(0), MFMAs per k-step: each group paid a full LDS round trip)
        typename TA::Frag a[TA::KSTEPS][TM];
        typename TB::Frag b[TA::KSTEPS][TN];
#pragma unroll
        for (int ks = 0; ks < TA::KSTEPS; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[ks][i] = As.frag(wm0 + 32 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[ks][j] = Bs.frag(wn0 + 32 * j, ks, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < TA::KSTEPS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[ks][i], b[ks][j], acc[i][j]);
        return;
    }
#pragma unroll
    for (int ks = 0; ks < TA::KSTEPS; ++ks) {
        typename TA::Frag a[TM];
        typename TB::Frag b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As.frag(wm0 + 32 * i, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs.frag(wn0 + 32 * j, ks, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
    }
}

// C-fragment coordinates of accumulator register r in a 32x32 tile (dtype independent on gfx950)
// validity of the 16 taps (bit ky * 4 + kx) of a 4x4 window whose corner is (iy0, ix0): separable, so 4 + 4 range tests
// and 4 selects instead of 16 x 2 tests per gathered row (the prologue of every forward launch)
__device__ __forceinline__ unsigned tap_mask16(int iy0, int ix0, int Hi, int Wi) {
    unsigned xm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) xm |= ((unsigned)(ix0 + k) < (unsigned)Wi ? 1u : 0u) << k;
    unsigned mk = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) mk |= ((unsigned)(iy0 + k) < (unsigned)Hi ? xm : 0u) << (4 * k);
    return mk;
}
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// filter geometry: KS = 4 -> 4x4 stride 2 pad 1 (the cGAN's convs); KS = 3 -> 3x3 stride 1 pad 1 (GeneratorSimpleRegressor,
// cgan/models.py:163-193).  Both use tap = ky*KS + kx inside a [tap][channel] K axis; packed weight rows are p.wk elements.
template <int KS> struct Geo;
template <> struct Geo<4> {
    static constexpr int ST = 2, TAPS = 16;
    __device__ static int ky(int t) { return t >> 2; }
    __device__ static int kx(int t) { return t & 3; }
};
template <> struct Geo<3> {
    static constexpr int ST = 1, TAPS = 9;
    __device__ static int ky(int t) { return (t * 11) >> 5; }            // t / 3 for 0 <= t < 16
    __device__ static int kx(int t) { return t - 3 * ((t * 11) >> 5); }
};

struct ConvParams {
    const void* x;      // fwd: input act; dgrad: dy; wgrad: x (high-res input of the conv)
    const void* w;      // fwd: Wf; dgrad: Wt; wgrad: dy (low-res)
    void* y;            // fwd: output; dgrad: dx; wgrad: slab (fp32)
    const float* bias;  // fwd only, nullable
    const float* gscale;  // per-group multiplier, nullable
    int group_n;        // samples per group
    float inv_group_n;  // 1 / group_n (sample -> group index in the epilogues)
    int ldx, ldw, ldy;  // pixel strides (elements); ldw = lddy for wgrad
    int wk;             // 3x3 forms: elements per packed weight row (9*Cin rounded up to 64) = length of the K loop
    int xcd_remap;      // wgrad: deal the workgroups of a K split to one XCD (A/B knob GCSSL_WGRAD_XCD=0 turns it off)
    int class_major;    // persistent dgrad form: 1 = walk the tiles class by class (A/B knob GCSSL_DGRAD_ORDER=1), 0 = class-interleaved
    int sib_remap;      // fwd / dgrad: deal all tiles that read the SAME input rows (the N tiles and, for dgrad, the four parity
                        // classes of one M tile) to one XCD back to back (A/B knob GCSSL_SIB_REMAP=0)
    int N, Hi, Wi, Cin, Cout;   // conv geometry: x is [N][Hi][Wi][Cin], y is [N][Hi/2][Wi/2][Cout]
    int lgWo, lgHoWo, lgCin, lgCout;
    int M;              // GEMM rows
    int act;            // fwd epilogue: 1 = LeakyReLU(0.2)
    int out_f32;        // fwd/dgrad: write fp32 regardless of T
    int ktiles_per_split;   // wgrad; fwd/dgrad when ksplit > 1
    int ksplit;             // fwd/dgrad: K is split over ksplit workgroups that atomically add into a zeroed fp32 output
    int* plan_out;          // dry run (gcssl_conv4x4s2_*_splits): the dispatcher stores the K split it would use and launches nothing
    long split_stride;      // ... or, when > 0, store their partial sums plainly into slab ks at y + ks*split_stride (fp32 elements);
                            // the consumer (gcssl_in_act_fwd nslab) adds the slabs: float atomics run at 1.3 TB/s chip-wide
    unsigned x_bytes, w_bytes;   // extents of the two operand buffers (buffer-load bounds; < 2^31)
    unsigned y_bytes;            // extent of the output (buffer-store bounds of the persistent kernel); 0 = unknown/too large
    int dbg;                     // timing experiments on the ring form (GCSSL_RING_DEBUG bits): 1 loaders issue nothing, 2 consumers
                                 // compute nothing, 4 no epilogue, 8 return at entry, 16 no K loop (results are garbage)
    int epi_lds;                 // conv_dma_kernel: result tile through LDS, 16-byte row stores (GCSSL_EPI_LDS=0: per-lane stores)
    // ---- FIN forms (gcssl_conv4x4s2_in_act_fwd): InstanceNorm + activation in the epilogue.  y is the 16-bit ACTIVATION
    // (pixel stride ldy); the whole H*W map of a sample lies inside one M tile, so the statistics need no second pass.
    float* in_mean; float* in_rstd;       // [N][Cout] fp32 outputs
    const uint8_t* in_mask;               // dropout keep mask [N][Ho*Wo][Cout] or null (kept values x 2)
    // ---- ACTB forms (gcssl_conv4x4s2_dgrad_act_bwd): the data gradient of a conv whose INPUT is the output of a norm-less
    // LeakyReLU layer (D.c1 / G.down1), with that layer's activation backward in the epilogue: y = dzs = lrelu'(a) * dx * gscale
    // in the compute dtype; the fp32 dx never goes to memory.
    const void* ab_a; int ab_lda; unsigned ab_bytes;   // the layer's stored activation lrelu(z) at this dgrad's OUTPUT pixels
    const float* ab_bias;                 // its conv bias (for the spectral-norm dot), nullable
    float* ab_dbias; float* ab_cdot;      // [Cin] += sum dz / [group] += sum dzs (z - bias), striped over ab_nrep replicas; nullable
    int ab_nrep, ab_rep_stride;
    unsigned* ab_sat;                     // += fp16 stores that clipped (nullable)
    const void* ab_dotx; int ab_lddot; float* ab_dot_out;   // conv_fwd_c8_kernel's ACTB form: dot_out += sum dotx * y (y = the conv
                                          // output BEFORE the activation backward: gcssl_dot_accum folded in), nullable
    int kcap;                             // timing experiment (GCSSL_KCAP, 3x3 persistent form only; results are garbage): walk only the
                                          // first kcap K steps of every tile -- the K volume a Winograd F(2x2,3x3) GEMM stage would have
    float mm_oscale;                      // split-precision forms (MM != 0): the packed weights carry a power-of-two pre-scale
                                          // (gcssl_prep_conv_weights with a split dtype); its inverse, applied in the epilogue
    float* fin_z; int ld_fin_z;           // conv_fwd_kernel's FIN form (split-precision modes): the fp32 pre-norm tensor is STILL stored (nullable)
    void* in_apre; int ld_apre, apre_n0;  // optional second output: the activation WITHOUT dropout for samples n >= apre_n0
                                          // ([N - apre_n0][Ho*Wo][ld_apre]): what the backward rebuilds xhat from
};

// ------------------------------------------------------------------------------------------
// forward: y[m][co] = act( gscale[g(m)] * sum_{tap,ci} x[n, 2oy-1+ky, 2ox-1+kx, ci] Wf[co][tap][ci] + bias[co] )
// GEMM M = N*Ho*Wo, N = Cout, K = 16*Cin
// ------------------------------------------------------------------------------------------
// WM x WN waves (default 2 x 2 = the 256 threads of NT; the split-precision forms also run 4 x 2: a K step of theirs is issue-bound
// -- MFMA phase, then split + LDS stores -- and with two waves per SIMD inside a workgroup one wave's VALU work runs under the other's MFMAs)
// FIN (split-precision forms, fp32 tensors): InstanceNorm + LeakyReLU in the epilogue, as in conv_dma_kernel's FIN forms -- the tile
// holds BM / (H*W) whole samples -- but with fp32 outputs and the pre-norm z STILL written (p.fin_z, nullable): the fp32 backward
// kernels read z; what the fusion removes is the separate norm launch and its read of z.
template <typename T, int BM, int BN, int KS = 4, int MM = 0, int WM = 2, int WN = 2, bool FIN = false>
__global__ __launch_bounds__(WM * WN * 64, MM ? (WM * WN >= 8 ? 4 : 2) : 1) void conv_fwd_kernel(ConvParams p) {
    typedef Geo<KS> G;
    constexpr int NT = WM * WN * 64;
    constexpr int BK = BKOf<T>::v;
    constexpr int KV = Elem<T>::KV, CH = BK / KV, RPT = NT / CH;   // rows covered per pass
    constexpr int NVA = BM / RPT, NVB = BN / RPT;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(NVA >= 1 && NVB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
    __shared__ typename KTile<T, BM, MM>::type As[2];
    __shared__ typename KTile<T, BN, MM>::type Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const T* x = static_cast<const T*>(p.x);
    const T* w = static_cast<const T*>(p.w);
    // split tiles: a 16-lane group of ds_write_b64 (banks mod 32 = a 128-byte window) must not hold two rows whose hi (or lo)
    // halves share a 64-byte half-row: rows R and R + 1 do (same swizzle), rows R and R + 8 do not (their swizzle differs by 4
    // chunks) -- so lanes 8-15 of a group take row R + 8 instead of R + 1 (SQ_LDS_BANK_CONFLICT: a third of the LDS cycles before)
    const int chunk = tid % CH;
    const int row_t = MM ? (((tid >> 4) & 7) + 8 * ((tid >> 3) & 1) + 16 * (tid >> 7)) : tid / CH;
    const int K = KS == 4 ? 16 * p.Cin : p.wk;
    const int Ho = p.Hi / G::ST, Wo = p.Wi / G::ST;

    constexpr int ES = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    int rowoff[NVA]; unsigned rowmask[NVA];     // byte offset of pixel (2oy-1, 2ox-1); bit t = tap t is inside the image
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + row_t + i * RPT;
        rowoff[i] = 0; rowmask[i] = 0;
        if (m < p.M) {
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            const int iy0 = G::ST * (rem >> p.lgWo) - 1, ix0 = G::ST * (rem & (Wo - 1)) - 1;
            rowoff[i] = ((n * p.Hi + iy0) * p.Wi + ix0) * p.ldx * ES;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < G::TAPS; ++t)
                if ((unsigned)(iy0 + G::ky(t)) < (unsigned)p.Hi && (unsigned)(ix0 + G::kx(t)) < (unsigned)p.Wi) mk |= 1u << t;
            rowmask[i] = mk;                      // (3x3, 8-channel first layer: the K axis is padded to 16 taps, bits 9-15 stay 0)
        }
    }
    (void)Ho;
    unsigned wrow[NVB];
#pragma unroll
    for (int j = 0; j < NVB; ++j) { const int co = n0 + row_t + j * RPT; wrow[j] = co < p.Cout ? (unsigned)(co * K * ES) : OOB; }
    Vec16<T> ra[MM ? 2 : 1][NVA], rb[MM ? 2 : 1][NVB];
    // live = false: a K tile past the end of this workgroup's range -- every offset out of range, the loads return zeros
    // without touching memory (the split-precision pipeline issues its loads unconditionally: see its loop)
    auto gload = [&](Vec16<T> (&qa)[NVA], Vec16<T> (&qb)[NVB], int k0, bool live = true) {
        const int k = k0 + chunk * KV;
        const int tap = k >> p.lgCin, ci = k & (p.Cin - 1);
        const int tapoff = ((G::ky(tap) * p.Wi + G::kx(tap)) * p.ldx + ci) * ES;
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            qa[i] = bload<T>(xr, (live && ((rowmask[i] >> (tap & 31)) & 1u)) ? (unsigned)(rowoff[i] + tapoff) : OOB);
#pragma unroll
        for (int j = 0; j < NVB; ++j) qb[j] = bload<T>(wr, live ? wrow[j] + (unsigned)(k * ES) : OOB);
    };
    auto lstore = [&](const Vec16<T> (&qa)[NVA], const Vec16<T> (&qb)[NVB], int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) As[buf].store_vec(row_t + i * RPT, chunk, qa[i]);
#pragma unroll
        for (int j = 0; j < NVB; ++j) Bs[buf].store_vec(row_t + j * RPT, chunk, qb[j]);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = K / BK;
    const int ks = p.ksplit > 1 ? (int)blockIdx.z : 0;
    const int t_beg = p.ksplit > 1 ? ks * p.ktiles_per_split : 0;
    const int t_end = p.ksplit > 1 ? min(nk_all, t_beg + p.ktiles_per_split) : nk_all;
    // Software pipeline: LDS double buffer + two register sets.  While the MFMAs consume tile t from LDS, tile t+1 is
    // landing in one register set (issued a whole iteration ago) and tile t+2 is being issued into the other.
    // (A two-tiles-ahead variant with a second register set was measured 1.5-1.9x SLOWER on the 128-wide tiles: it
    //  pushed them to 130-194 VGPRs and the lost occupancy cost more than the extra overlap bought.)
    if constexpr (MM != 0) {
        // split-precision forms: TWO K tiles of loads in flight (the second register set) -- these kernels are paced by their
        // fill rate = bytes in flight / latency (fp32 operands: twice the bytes of a 16-bit tile), and their MFMA phase is three
        // times as long, so a tile issued two steps ahead has landed when its turn to be split and stored comes
        // Every load / split / LDS store of the loop is UNCONDITIONAL (tiles past the end load zeros from out-of-range offsets):
        // a load behind a branch makes the compiler's s_waitcnt vmcnt assume it was not issued, so the split of the older tile
        // waited for the tile issued a moment ago as well -- one exposed memory latency per K step (round 4: 2 us per step).
        if (p.dbg) {                                               // timing experiments (GCSSL_X3_DEBUG; results are garbage):
            const bool nold = p.dbg & 1, nomma = p.dbg & 2, nost = p.dbg & 4;   // 1 no global loads, 2 no MFMA phase, 4 no split + LDS stores
            gload(ra[0], rb[0], t_beg * BK);
            gload(ra[1], rb[1], (t_beg + 1) * BK, t_beg + 1 < t_end);
            lstore(ra[0], rb[0], 0); lstore(ra[1], rb[1], 1); __syncthreads();
            for (int t = t_beg; t < t_end; t += 2) {
                if (!nold) gload(ra[0], rb[0], (t + 2) * BK, t + 2 < t_end);
                if (!nomma) mma_slab<TM, TN>(As[0], Bs[0], wm0, wn0, lane, acc);
                if (!nost) lstore(ra[1], rb[1], 1);
                __syncthreads();
                if (!nold) gload(ra[1], rb[1], (t + 3) * BK, t + 3 < t_end);
                if (!nomma) mma_slab<TM, TN>(As[1], Bs[1], wm0, wn0, lane, acc);
                if (!nost) lstore(ra[0], rb[0], 0);
                __syncthreads();
            }
        } else
        if (t_beg < t_end) {
            gload(ra[0], rb[0], t_beg * BK);
            gload(ra[1], rb[1], (t_beg + 1) * BK, t_beg + 1 < t_end);
            lstore(ra[0], rb[0], 0); __syncthreads();
            for (int t = t_beg; t < t_end; t += 2) {
                gload(ra[0], rb[0], (t + 2) * BK, t + 2 < t_end);
                mma_slab<TM, TN>(As[0], Bs[0], wm0, wn0, lane, acc);
                lstore(ra[1], rb[1], 1);
                __syncthreads();
                gload(ra[1], rb[1], (t + 3) * BK, t + 3 < t_end);
                if (t + 1 < t_end) mma_slab<TM, TN>(As[1], Bs[1], wm0, wn0, lane, acc);
                lstore(ra[0], rb[0], 0);
                __syncthreads();
            }
        }
    } else
    if (t_beg < t_end) {
        gload(ra[0], rb[0], t_beg * BK); lstore(ra[0], rb[0], 0); __syncthreads();
        for (int t = t_beg; t < t_end; ++t) {
            const int b = (t - t_beg) & 1;
            if (t + 1 < t_end) gload(ra[0], rb[0], (t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < t_end) lstore(ra[0], rb[0], b ^ 1);
            __syncthreads();
        }
    }
    // epilogue (out_f32: the pre-norm tensor z stays fp32 in bf16 mode, see norm.hip).  Everything the stores depend on is
    // loaded FIRST (per-row group scale, per-column bias): a load between two stores cannot be hoisted by the compiler (the output
    // may alias it), and one dependent load -> wait -> store round trip per row was most of a short-K launch (round 4: the
    // split-precision modes run this kernel's K loop 3x faster than the fp32 MFMA did, and 32 such round trips per wave showed)
    T* y = static_cast<T*>(p.y);
    float* y32 = static_cast<float*>(p.y);
    float sc[TM][16];
    const float osc = MM ? p.mm_oscale : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[i][r] = osc;
    if (p.gscale) {                                 // ONE uniform branch; inside it every lane loads (rows past M: the last sample's
#pragma unroll                                      // group), so the loads go out back to back behind a single wait
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(m0 + wm0 + 32 * i + crow(r, lane), p.M - 1);
                sc[i][r] = osc * p.gscale[(int)(((float)(m >> p.lgHoWo) + 0.5f) * p.inv_group_n)];
            }
    }
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn0 + 32 * j + (lane & 31);
        bcol[j] = (p.bias && co < p.Cout && (p.ksplit <= 1 || ks == 0)) ? p.bias[co] : 0.f;
    }
    if constexpr (FIN) {
        // (host: fp32 tensors, no K split, H*W <= 64 divides BM, Cout % 4 == 0, 16-byte aligned rows)
        constexpr int RS = BN * 4;                                   // LDS tile row stride (bytes): the operand tiles are done with
        static_assert(sizeof(As) + sizeof(Bs) >= (size_t)BM * RS, "result tile must fit the operand tiles");
        __shared__ float fin_stat[(BM / 4) * BN * 2];              // (mu, rstd) per (sample of the tile, column)
        unsigned char* lds = reinterpret_cast<unsigned char*>(&As[0]);
        const int lgHW = p.lgHoWo, HW = 1 << lgHW, nsamp = BM >> lgHW;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + 32 * i + crow(r, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float*>(lds + row * RS + (wn0 + 32 * j + (lane & 31)) * 4) = acc[i][j][r] * sc[i][r] + bcol[j];
            }
        __syncthreads();
        const int s0 = m0 >> lgHW;
        const float inv_hw = 1.f / (float)HW;
        for (int q = tid; q < nsamp * BN; q += NT) {                 // one (sample, column) per thread: exact two-pass statistics
            const int sidx = q / BN, c = q % BN;
            const unsigned char* colp = lds + (sidx << lgHW) * RS + c * 4;
            float sum = 0.f;
            for (int r = 0; r < HW; ++r) sum += *reinterpret_cast<const float*>(colp + r * RS);
            const float mu = sum * inv_hw;
            float m2 = 0.f;
            for (int r = 0; r < HW; ++r) { const float d = *reinterpret_cast<const float*>(colp + r * RS) - mu; m2 += d * d; }
            const float rs = 1.0f / sqrtf(m2 * inv_hw + 1e-5f);
            fin_stat[q * 2] = mu; fin_stat[q * 2 + 1] = rs;
            const int n = s0 + sidx, cg = n0 + c;
            if (n < p.N && cg < p.Cout) { p.in_mean[(size_t)n * p.Cout + cg] = mu; p.in_rstd[(size_t)n * p.Cout + cg] = rs; }
        }
        __syncthreads();
        float* az = static_cast<float*>(p.y);                        // the activation
        float* zz = p.fin_z;                                         // the pre-norm values (nullable)
        constexpr int CPR = BN / 4;                                  // 16-byte chunks per tile row
        for (int cix = tid; cix < BM * CPR; cix += NT) {
            const int row = cix / CPR, ch = cix % CPR, m = m0 + row, col0 = n0 + ch * 4;
            if (m >= p.M || col0 >= p.Cout) continue;
            const float4 v = *reinterpret_cast<const float4*>(lds + row * RS + ch * 16);
            const float4* st4 = reinterpret_cast<const float4*>(fin_stat + ((row >> lgHW) * BN + ch * 4) * 2);
            const float4 q0 = st4[0], q1 = st4[1];                   // (mu, rs) x 4 columns
            float4 o;
            o.x = lrelu_f((v.x - q0.x) * q0.y); o.y = lrelu_f((v.y - q0.z) * q0.w);
            o.z = lrelu_f((v.z - q1.x) * q1.y); o.w = lrelu_f((v.w - q1.z) * q1.w);
            if (zz) *reinterpret_cast<float4*>(zz + (size_t)m * p.ld_fin_z + col0) = v;
            *reinterpret_cast<float4*>(az + (size_t)m * p.ldy + col0) = o;
        }
        return;
    }
    float* yk = y32 + (p.ksplit > 1 ? (size_t)ks * p.split_stride : 0);
    const bool f32o = p.out_f32 || std::is_same<T, float>::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * i + crow(r, lane);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn0 + 32 * j + (lane & 31);
                if (co >= p.Cout) continue;
                float v = acc[i][j][r] * sc[i][r] + bcol[j];
                if (p.ksplit > 1) {                       // linear epilogue only: partial sums add up, bias once
                    if (p.split_stride) yk[(size_t)m * p.ldy + co] = v;
                    else atomicAdd(y32 + (size_t)m * p.ldy + co, v);
                    continue;
                }
                if (p.act == 1) v = lrelu_f(v);
                if (f32o) y32[(size_t)m * p.ldy + co] = v;
                else Elem<T>::st(y + (size_t)m * p.ldy + co, v);
            }
        }
}

// ------------------------------------------------------------------------------------------
// dgrad / transposed conv: dx[n,iy,ix,ci] = gscale * sum_{co, taps matching parity} dy[n,oy,ox,co] Wt[ci][tap][co]
// One launch z-slice per output parity class (py,px); per class M = N*Ho*Wo, N = Cin, K = 4*Cout.
//   ky = 1-py+2ty, oy = iy' + py - ty   (iy = 2 iy' + py), same in x.
// ------------------------------------------------------------------------------------------
// ACTB (split-precision modes only): the activation-backward epilogue of gcssl_conv4x4s2_dgrad_act_bwd.  A template parameter and
// not a run-time branch: with the branch in it the plain instantiation went from 168 to 176 registers -- from three waves per SIMD
// to two -- and EVERY data gradient of the split modes from 72 to 91 us.
template <typename T, int BM, int BN, int MM = 0, int WM = 2, int WN = 2, bool ACTB = false>
__global__ __launch_bounds__(WM * WN * 64, MM ? (WM * WN >= 8 ? 4 : (ACTB ? 3 : 2)) : 1) void conv_dgrad_kernel(ConvParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BK = BKOf<T>::v;
    constexpr int KV = Elem<T>::KV, CH = BK / KV, RPT = NT / CH;
    constexpr int NVA = BM / RPT, NVB = BN / RPT;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(NVA >= 1 && NVB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
    __shared__ typename KTile<T, BM, MM>::type As[2];
    __shared__ typename KTile<T, BN, MM>::type Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int cls = p.ksplit > 1 ? (int)blockIdx.z / p.ksplit : (int)blockIdx.z;
    const int ks = p.ksplit > 1 ? (int)blockIdx.z % p.ksplit : 0;
    const int py = cls >> 1, px = cls & 1;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const T* dy = static_cast<const T*>(p.x);
    const T* wt = static_cast<const T*>(p.w);
    // split tiles: a 16-lane group of ds_write_b64 (banks mod 32 = a 128-byte window) must not hold two rows whose hi (or lo)
    // halves share a 64-byte half-row: rows R and R + 1 do (same swizzle), rows R and R + 8 do not (their swizzle differs by 4
    // chunks) -- so lanes 8-15 of a group take row R + 8 instead of R + 1 (SQ_LDS_BANK_CONFLICT: a third of the LDS cycles before)
    const int chunk = tid % CH;
    const int row_t = MM ? (((tid >> 4) & 7) + 8 * ((tid >> 3) & 1) + 16 * (tid >> 7)) : tid / CH;
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1;
    const int K = 4 * p.Cout;

    constexpr int ES = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    int rowoff[NVA]; unsigned rowmask[NVA];     // byte offset of dy pixel (iy'+py, ix'+px); bit t=(ty,tx): (oy,ox)=(..-ty,..-tx) inside
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + row_t + i * RPT;
        rowoff[i] = 0; rowmask[i] = 0;
        if (m < p.M) {
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            const int yy = (rem >> p.lgWo) + py, xx = (rem & (Wo - 1)) + px;
            rowoff[i] = ((n * Ho + yy) * Wo + xx) * p.ldx * ES;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if ((unsigned)(yy - (t >> 1)) < (unsigned)Ho && (unsigned)(xx - (t & 1)) < (unsigned)Wo) mk |= 1u << t;
            rowmask[i] = mk;
        }
    }
    unsigned wrow[NVB];
#pragma unroll
    for (int j = 0; j < NVB; ++j) { const int ci = n0 + row_t + j * RPT; wrow[j] = ci < p.Cin ? (unsigned)(ci * 16 * p.Cout * ES) : OOB; }
    Vec16<T> ra[MM ? 2 : 1][NVA], rb[MM ? 2 : 1][NVB];
    auto gload = [&](Vec16<T> (&qa)[NVA], Vec16<T> (&qb)[NVB], int k0, bool live = true) {      // (live: see conv_fwd_kernel)
        const int k = k0 + chunk * KV;
        const int t = k >> p.lgCout, co = k & (p.Cout - 1);
        const int ty = (t >> 1) & 1, tx = t & 1;
        const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
        const int tapoff = (co - (ty * Wo + tx) * p.ldx) * ES;
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            qa[i] = bload<T>(xr, (live && ((rowmask[i] >> (t & 31)) & 1u)) ? (unsigned)(rowoff[i] + tapoff) : OOB);
#pragma unroll
        for (int j = 0; j < NVB; ++j) qb[j] = bload<T>(wr, live ? wrow[j] + (unsigned)((tap * p.Cout + co) * ES) : OOB);
    };
    auto lstore = [&](const Vec16<T> (&qa)[NVA], const Vec16<T> (&qb)[NVB], int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) As[buf].store_vec(row_t + i * RPT, chunk, qa[i]);
#pragma unroll
        for (int j = 0; j < NVB; ++j) Bs[buf].store_vec(row_t + j * RPT, chunk, qb[j]);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk_all = K / BK;
    const int t_beg = p.ksplit > 1 ? ks * p.ktiles_per_split : 0;
    const int t_end = p.ksplit > 1 ? min(nk_all, t_beg + p.ktiles_per_split) : nk_all;
    // Software pipeline: LDS double buffer + two register sets.  While the MFMAs consume tile t from LDS, tile t+1 is
    // landing in one register set (issued a whole iteration ago) and tile t+2 is being issued into the other.
    // (A two-tiles-ahead variant with a second register set was measured 1.5-1.9x SLOWER on the 128-wide tiles: it
    //  pushed them to 130-194 VGPRs and the lost occupancy cost more than the extra overlap bought.)
    if constexpr (MM != 0) {
        // split-precision forms: TWO K tiles of loads in flight (the second register set) -- these kernels are paced by their
        // fill rate = bytes in flight / latency (fp32 operands: twice the bytes of a 16-bit tile), and their MFMA phase is three
        // times as long, so a tile issued two steps ahead has landed when its turn to be split and stored comes
        // Every load / split / LDS store of the loop is UNCONDITIONAL (tiles past the end load zeros from out-of-range offsets):
        // a load behind a branch makes the compiler's s_waitcnt vmcnt assume it was not issued, so the split of the older tile
        // waited for the tile issued a moment ago as well -- one exposed memory latency per K step (round 4: 2 us per step).
        if (t_beg < t_end) {
            gload(ra[0], rb[0], t_beg * BK);
            gload(ra[1], rb[1], (t_beg + 1) * BK, t_beg + 1 < t_end);
            lstore(ra[0], rb[0], 0); __syncthreads();
            for (int t = t_beg; t < t_end; t += 2) {
                gload(ra[0], rb[0], (t + 2) * BK, t + 2 < t_end);
                mma_slab<TM, TN>(As[0], Bs[0], wm0, wn0, lane, acc);
                lstore(ra[1], rb[1], 1);
                __syncthreads();
                gload(ra[1], rb[1], (t + 3) * BK, t + 3 < t_end);
                if (t + 1 < t_end) mma_slab<TM, TN>(As[1], Bs[1], wm0, wn0, lane, acc);
                lstore(ra[0], rb[0], 0);
                __syncthreads();
            }
        }
    } else
    if (t_beg < t_end) {
        gload(ra[0], rb[0], t_beg * BK); lstore(ra[0], rb[0], 0); __syncthreads();
        for (int t = t_beg; t < t_end; ++t) {
            const int b = (t - t_beg) & 1;
            if (t + 1 < t_end) gload(ra[0], rb[0], (t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < t_end) lstore(ra[0], rb[0], b ^ 1);
            __syncthreads();
        }
    }
    // (epilogue: the per-row group scales are loaded before any store -- see conv_fwd_kernel)
    float sc[TM][16];
    const float osc = MM ? p.mm_oscale : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[i][r] = osc;
    if (p.gscale) {                                 // ONE uniform branch; inside it every lane loads (rows past M: the last sample's
#pragma unroll                                      // group), so the loads go out back to back behind a single wait
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(m0 + wm0 + 32 * i + crow(r, lane), p.M - 1);
                sc[i][r] = osc * p.gscale[(int)(((float)(m >> p.lgHoWo) + 0.5f) * p.inv_group_n)];
            }
    }
    float* y32 = static_cast<float*>(p.y);
    if constexpr (MM != 0 && ACTB) {
        {
            // split-precision modes, the data gradient into a NORM-LESS layer: that layer's LeakyReLU backward in this epilogue
            // (gcssl_conv4x4s2_dgrad_act_bwd on fp32 tensors; norm.hip act_bwd_kernel's arithmetic): d = the conv result,
            // dz = lrelu'(a) d, y = dz * gscale[group];  dbias[ci] += sum dz,  cdot[group] += sum y (z - bias[ci]), z = lrelu^-1(a).
            // A tile's rows belong to one group (host check); no K split (the epilogue is not linear).
            const float* ab = static_cast<const float*>(p.ab_a);
            const int mg = min(m0, p.M - 1);
            const int grp = (int)(((float)(mg >> p.lgHoWo) + 0.5f) * p.inv_group_n);
            const float gsv = p.gscale ? p.gscale[grp] : 1.f;
            float bj[TN], sbj[TN], sd = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = n0 + wn0 + 32 * j + (lane & 31);
                bj[j] = (p.ab_bias && ci < p.Cin) ? p.ab_bias[ci] : 0.f;
                sbj[j] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + 32 * i + crow(r, lane);
                    if (m >= p.M) continue;
                    const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                    const int iy = 2 * (rem >> p.lgWo) + py, ix = 2 * (rem & (Wo - 1)) + px;
                    const size_t pix = (size_t)(n * p.Hi + iy) * p.Wi + ix;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int ci = n0 + wn0 + 32 * j + (lane & 31);
                        if (ci >= p.Cin) continue;
                        const float d = acc[i][j][r] * osc;
                        const float av = ab[pix * p.ab_lda + ci];
                        const float dz = av > 0.f ? d : 0.2f * d;
                        const float zv = av > 0.f ? av : av * 5.0f;
                        const float o = dz * gsv;
                        sbj[j] += dz; sd += o * (zv - bj[j]);
                        y32[pix * p.ldy + ci] = o;
                    }
                }
            if (p.ab_dbias || p.ab_cdot) {
                float* red = reinterpret_cast<float*>(&As[0]);                     // [wave][TN * 32 + 1]
                constexpr int RW = TN * 32 + 1;
                __syncthreads();                                                   // (every wave is done with the operand tiles)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float t = sbj[j] + __shfl_xor(sbj[j], 32, 64);
                    if (lane < 32) red[wave * RW + j * 32 + lane] = t;
                }
                const float ts = wave_sum(sd);
                if (lane == 0) red[wave * RW + TN * 32] = ts;
                __syncthreads();
                const int rep = p.ab_nrep > 1 ? (int)(blockIdx.x % (unsigned)p.ab_nrep) * p.ab_rep_stride : 0;
                if (p.ab_dbias && tid < BN) {                                      // column tid of the tile
                    const int wc = tid / (BN / WN), cj = tid % (BN / WN);
                    float t = 0.f;
#pragma unroll
                    for (int wr = 0; wr < WM; ++wr) t += red[(wr * WN + wc) * RW + cj];
                    if (n0 + tid < p.Cin) atomicAdd(p.ab_dbias + rep + n0 + tid, t);
                }
                if (p.ab_cdot && tid == 0) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WM * WN; ++w) t += red[w * RW + TN * 32];
                    if (t != 0.f) atomicAdd(p.ab_cdot + rep + grp, t);
                }
            }
            return;
        }
    }
    float* yk = y32 + (p.ksplit > 1 ? (size_t)ks * p.split_stride : 0);
    const bool f32o = p.out_f32 || std::is_same<T, float>::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * i + crow(r, lane);
            if (m >= p.M) continue;
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            const int iy = 2 * (rem >> p.lgWo) + py, ix = 2 * (rem & (Wo - 1)) + px;
            const size_t pix = (size_t)(n * p.Hi + iy) * p.Wi + ix;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = n0 + wn0 + 32 * j + (lane & 31);
                if (ci >= p.Cin) continue;
                const float v = acc[i][j][r] * sc[i][r];
                if (p.ksplit > 1) {
                    if (p.split_stride) yk[pix * p.ldy + ci] = v;
                    else atomicAdd(y32 + pix * p.ldy + ci, v);
                }
                else if (f32o) y32[pix * p.ldy + ci] = v;
                else Elem<T>::st(static_cast<T*>(p.y) + pix * p.ldy + ci, v);
            }
        }
}

// ------------------------------------------------------------------------------------------
// bf16 forward / dgrad with LDS-DMA staging (MODE 0 = forward conv, 1 = dgrad / transposed conv).
// Same GEMM as conv_fwd_kernel / conv_dgrad_kernel, different pipeline: `buffer_load_dwordx4 ... lds` writes the
// gathered 16-byte channel vectors straight into a 3-slot LDS ring (no VGPR staging, no ds_write, no registers spent
// on prefetch), tiles t+1 and t+2 are in flight while the MFMAs consume tile t, and each K-step costs one raw
// s_barrier behind a COUNTED s_waitcnt vmcnt (never 0 inside the loop).
//   * The DMA destination is wave-uniform base + lane*16, i.e. lane-linear, so the XOR swizzle of KMajor<bf16> is
//     applied to the SOURCE: the lane that fills physical chunk pc of row r fetches logical chunk pc ^ ((r>>1)&7).
//   * Out-of-image taps / rows use an out-of-range buffer offset: the hardware returns 0 and the DMA writes zeros.
//   * All LDS lives in ONE __shared__ array (a second object makes hipcc drain vmcnt(0) before the ds_reads).
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_p;

// Linear workgroup / tile index -> (M tile, N tile, parity class).  Workgroups are dealt round-robin to the 8 XCDs, so indices
// L and L + 8 share an L2.  All SIB = tiles_n * ncls tiles of one M tile gather the same activation rows: blocks of 8 * SIB
// consecutive indices hold 8 M tiles, one per XCD, each with its siblings 8 apart -- the first sibling pulls the rows from
// HBM / Infinity Cache into the XCD's L2, the others hit there (the per-CU LDS-DMA rate is latency x bytes in flight: ~33 GB/s
// from beyond L2, ~70 GB/s from L2, MI355X_MICROARCH.md "Indexed rows: gather into LDS").  Placement only, never correctness.
__device__ __forceinline__ void tile_decode(int L, int tiles_m, int tiles_n, int ncls, int& mx, int& ny, int& cls) {
    const int sib = tiles_n * ncls, full = (tiles_m >> 3) * 8 * sib;
    int s;
    if (L < full) { const int w = L % (8 * sib); mx = (L / (8 * sib)) * 8 + (w & 7); s = w >> 3; }
    else { const int r = L - full; mx = (tiles_m & ~7) + r / sib; s = r % sib; }
    ny = s % tiles_n; cls = s / tiles_n;
}

// WM x WN waves per workgroup (4 or 8 waves); SMALLK: the K-tile spans several taps (first layer, Cin padded to 8),
// otherwise the tap of a K-tile is wave-uniform and its address arithmetic runs on the scalar unit.
// FIN: forward conv + InstanceNorm2d(eps 1e-5, biased variance) + LeakyReLU(0.2) [+ Dropout(0.5) mask] in ONE launch
// (cgan/models.py:57-63,236-242): for maps of H*W <= 64 pixels a BM-row tile holds BM / (H*W) whole samples, so the per-(n, c)
// statistics are complete inside the tile for its channel slice.  The fp32 accumulator tile goes to the idle ring, the
// statistics are the exact two-pass ones computed from LDS, and what leaves is the 16-bit activation + fp32 mean / rstd:
// the fp32 pre-norm tensor z (4 B written by the conv, 4 B read + 2 B written by a separate norm launch) never exists.
template <typename T, int BM, int BN, int MODE, int WM, int WN, bool SMALLK, int LW = 0, int NSLOT = 3, bool PIPE = true,
          bool FIN = false, bool ACTB = false>
__global__ __launch_bounds__((WM * WN + LW) * 64) void conv_dma_kernel(ConvParams p) {
    // The host pass only needs the launch stub; it silently marks this body invalid (device-only LDS-DMA builtin and
    // inline asm with template-dependent operands) and then emits NO stub, so the body is device-pass only.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    // LW > 0: LOADER / CONSUMER specialisation.  The WM x WN consumer waves only read fragments and run MFMAs; LW extra
    // waves own every LDS-DMA instruction and run NSLOT - 1 K tiles ahead in an NSLOT-slot ring; one s_barrier per K step
    // is the whole hand-off (tile t landed: the loaders' counted vmcnt in front of it; tile t-1 released: every consumer's
    // fragments are in registers before it arrives).  MI355X_MICROARCH.md prices an LDS-DMA piece at 100-185 cycles of issue
    // inside a phase that also carries fragment reads, 60 among bare instructions: in the LW = 0 form every wave pays
    // that between its reads and its MFMAs.
    constexpr int NCW = WM * WN;                                           // consumer waves
    constexpr int NTH = (LW ? LW : NCW) * 64;                              // threads that fetch
    constexpr int BK = 64, ES = 2, KV = 8, CH = 8, RPT = NTH / CH;        // tile rows per pass of the fetching threads
    constexpr int NVA = BM / RPT, NVB = BN / RPT, NL = NVA + NVB;          // DMA instructions per wave per K-tile
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for this many waves");
    static_assert(LW == 0 || (NSLOT >= 3 && NSLOT <= 5 && (NSLOT - 2) * NL <= 63), "ring depth");
    static_assert(LW > 0 || NSLOT == 3, "the LW = 0 form is a 3-slot ring");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = LW > 0 && wave >= NCW;
    if (LW > 0 && (p.dbg & 8)) return;                                  // (timing experiment: launch cost only)
    const int tid = LW ? (int)threadIdx.x - NCW * 64 : (int)threadIdx.x;   // index among the fetching threads (loaders: >= 0)
    const int fwave = LW ? wave - NCW : wave;
    int mxi = blockIdx.x, nyi = blockIdx.y;
    int cls = MODE == 1 ? (p.ksplit > 1 ? (int)blockIdx.z / p.ksplit : (int)blockIdx.z) : 0;
    const int ks = p.ksplit > 1 ? (MODE == 1 ? (int)blockIdx.z % p.ksplit : (int)blockIdx.z) : 0;
    if (p.sib_remap && p.ksplit <= 1)
        tile_decode((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)gridDim.x, (int)gridDim.y,
                    MODE == 1 ? 4 : 1, mxi, nyi, cls);
    const int m0 = mxi * BM, n0 = nyi * BN;
    const int py = cls >> 1, px = cls & 1;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int Ho = MODE == 2 ? p.Hi : p.Hi >> 1, Wo = MODE == 2 ? p.Wi : p.Wi >> 1;
    const int K = MODE == 0 ? 16 * p.Cin : MODE == 2 ? p.wk : 4 * p.Cout;
    const int row_t = tid / CH;
    const int lc = (tid % CH) ^ ((row_t >> 1) & 7);                        // logical chunk this lane fetches
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);

    int rowoff[NVA]; unsigned rowmask[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + row_t + i * RPT;
        rowoff[i] = 0; rowmask[i] = 0;
        if (m < p.M) {
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            unsigned mk = 0;
            if (MODE == 0) {
                const int iy0 = 2 * (rem >> p.lgWo) - 1, ix0 = 2 * (rem & (Wo - 1)) - 1;
                rowoff[i] = ((n * p.Hi + iy0) * p.Wi + ix0) * p.ldx * ES;
                mk = tap_mask16(iy0, ix0, p.Hi, p.Wi);
            } else if (MODE == 2) {                                        // 3x3 stride 1 pad 1 (Geo<3>): taps 0..8
                const int iy0 = (rem >> p.lgWo) - 1, ix0 = (rem & (Wo - 1)) - 1;
                rowoff[i] = ((n * p.Hi + iy0) * p.Wi + ix0) * p.ldx * ES;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if ((unsigned)(iy0 + Geo<3>::ky(t)) < (unsigned)p.Hi && (unsigned)(ix0 + Geo<3>::kx(t)) < (unsigned)p.Wi) mk |= 1u << t;
            } else {
                const int yy = (rem >> p.lgWo) + py, xx = (rem & (Wo - 1)) + px;
                rowoff[i] = ((n * Ho + yy) * Wo + xx) * p.ldx * ES;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if ((unsigned)(yy - (t >> 1)) < (unsigned)Ho && (unsigned)(xx - (t & 1)) < (unsigned)Wo) mk |= 1u << t;
            }
            rowmask[i] = mk;
            if (!SMALLK) rowoff[i] += lc * KV * ES;                        // the lane's chunk offset never changes
        }
    }
    unsigned wrow[NVB];
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
        const int r = n0 + row_t + j * RPT;
        if (MODE != 1) wrow[j] = r < p.Cout ? (unsigned)(r * K * ES) : OOB;
        else wrow[j] = r < p.Cin ? (unsigned)(r * 16 * p.Cout * ES) : OOB;
        if (!SMALLK && wrow[j] != OOB) wrow[j] += lc * KV * ES;
    }
    // (plain ints on purpose: with value-dependent operands hipcc's HOST pass silently rejects the LDS-DMA builtin call,
    //  marks the kernel invalid and emits no launch stub)
    int vstride = NTH * 16, a_bytes = A_BYTES, stage = STAGE;
    auto issue = [&](int t, int slot) {
        unsigned char* base = lds + slot * stage + fwave * 1024;
        int tapbit, tapoff; unsigned woff;
        const int kb = SMALLK ? t * BK + lc * KV : t * BK;                 // !SMALLK: wave-uniform -> SALU
        if (MODE == 0) {
            const int tap = kb >> p.lgCin, ci = kb & (p.Cin - 1);
            tapbit = tap; tapoff = (((tap >> 2) * p.Wi + (tap & 3)) * p.ldx + ci) * ES; woff = (unsigned)(kb * ES);
        } else if (MODE == 2) {
            const int tap = kb >> p.lgCin, ci = kb & (p.Cin - 1);
            tapbit = tap; tapoff = ((Geo<3>::ky(tap) * p.Wi + Geo<3>::kx(tap)) * p.ldx + ci) * ES; woff = (unsigned)(kb * ES);
        } else {
            const int t4 = kb >> p.lgCout, co = kb & (p.Cout - 1);
            const int ty = t4 >> 1, tx = t4 & 1;
            const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
            tapbit = t4; tapoff = (co - (ty * Wo + tx) * p.ldx) * ES; woff = (unsigned)((tap * p.Cout + co) * ES);
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            // (the offset expression stays inline: it keeps the call type-dependent, so hipcc's HOST pass defers checking
            //  this device-only builtin instead of silently invalidating the kernel and dropping its launch stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(base + i * vstride), 16,
                ((rowmask[i] >> tapbit) & 1u) ? (unsigned)(rowoff[i] + tapoff) : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_p)(base + a_bytes + j * vstride), 16, wrow[j] + woff, 0, 0, 0);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto frag = [&](const unsigned char* tile, int row0, int kstep) -> FragT {
        const int row = row0 + (lane & 31), c = kstep * 2 + (lane >> 5);
        return __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)));
    };
    const int nk_all = K / BK;
    const int t_beg = p.ksplit > 1 ? ks * p.ktiles_per_split : 0;
    const int t_end = p.ksplit > 1 ? min(nk_all, t_beg + p.ktiles_per_split) : nk_all;
    if (LW > 0 && (p.dbg & 16)) {
        if (loader) return;                                              // (timing experiment: no K loop)
    } else if (LW > 0) {
        constexpr int AHEAD = NSLOT - 1;                                   // K tiles issued ahead of the one being consumed
        if (loader) {
            for (int q = 0; q < AHEAD; ++q)
                if (t_beg + q < t_end && !(p.dbg & 1)) issue(t_beg + q, q);
            int slot = 0;
            for (int t = t_beg; t < t_end; ++t) {
                // tile t has landed once only the younger tiles' DMA instructions (NL each) are outstanding
                const int young = min(AHEAD - 1, t_end - 1 - t);
                if (young >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL <= 63 ? 3 * NL : 63) : "memory");
                else if (young == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL <= 63 ? 2 * NL : 63) : "memory");
                else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // tile t is published; tile t-1 is released (its slot is refilled now)
                if (t + AHEAD < t_end && !(p.dbg & 1)) issue(t + AHEAD, slot == 0 ? NSLOT - 1 : slot - 1);
                slot = slot == NSLOT - 1 ? 0 : slot + 1;
            }
            return;                                                        // (no barrier follows: the epilogue is the consumers')
        }
        // Consumers, software-pipelined over the barrier: after barrier j the fragment reads of tile j are ISSUED and the MFMAs
        // of tile j-1 (fragments already in registers) run while those reads are in flight.  With reads -> wait -> MFMAs inside
        // one barrier interval a step cost ~1550 cycles for 512 cycles of MFMA per SIMD: all eight waves read at once right
        // after the barrier, then all run their MFMAs, and nothing overlaps.  (tools/archive/probes/fill_probe.hip: the loaders alone
        // bring this layer's gather in at 80 GB/s per CU, twice what the unpipelined loop consumed.)
        if constexpr (!PIPE) {                                             // reads -> MFMAs inside one barrier interval (fewer registers)
            int slot = 0;
            for (int t = t_beg; t < t_end; ++t) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* At = lds + slot * STAGE;
                const unsigned char* Bt = At + A_BYTES;
                FragT a[BK / 16][TM], b[BK / 16][TN];
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[kk][i] = frag(At, wm0 + 32 * i, kk);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[kk][j] = frag(Bt, wn0 + 32 * j, kk);
                }
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[kk][i], b[kk][j], acc[i][j]);
                slot = slot == NSLOT - 1 ? 0 : slot + 1;
            }
        } else {
        FragT a0[BK / 16][TM], b0[BK / 16][TN], a1[BK / 16][TM], b1[BK / 16][TN];
        auto rd = [&](FragT (&a)[BK / 16][TM], FragT (&b)[BK / 16][TN], int slot) {
            if (p.dbg & 2) return;
            const unsigned char* At = lds + slot * STAGE;
            const unsigned char* Bt = At + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = frag(At, wm0 + 32 * i, kk);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = frag(Bt, wn0 + 32 * j, kk);
            }
        };
        auto mm = [&](const FragT (&a)[BK / 16][TM], const FragT (&b)[BK / 16][TN]) {
            if (p.dbg & 2) return;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[kk][i], b[kk][j], acc[i][j]);
        };
        auto next = [](int slot) { return slot == NSLOT - 1 ? 0 : slot + 1; };
        if (t_beg < t_end) {
            int slot = 0, t = t_beg;
            __builtin_amdgcn_s_barrier();                                  // barrier of tile t_beg
            rd(a0, b0, slot); slot = next(slot);
            // (lgkmcnt(0) in front of every barrier: the loaders refill the slot of the tile whose reads were issued in the
            //  interval that ends there -- those reads must have returned; they were issued a whole MFMA block earlier)
            for (; t + 2 < t_end; t += 2) {                                // tiles t (in a0/b0), t+1, t+2
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                rd(a1, b1, slot); slot = next(slot);
                mm(a0, b0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                rd(a0, b0, slot); slot = next(slot);
                mm(a1, b1);
            }
            if (t + 1 < t_end) {                                           // two tiles left: t in a0/b0 and t+1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                rd(a1, b1, slot);
                mm(a0, b0);
                mm(a1, b1);
            } else {
                mm(a0, b0);
            }
        }
        }
    } else if (t_beg < t_end) {
        issue(t_beg, 0);
        if (t_beg + 1 < t_end) issue(t_beg + 1, 1);
        int slot = 0;
        for (int t = t_beg; t < t_end; ++t) {
            // tile t has landed once at most the NL younger DMA instructions (tile t+1) are still outstanding
            if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone's part of tile t landed; everyone is done reading tile t-1
            __builtin_amdgcn_sched_barrier(0);
            // All fragment reads of the step go out FIRST, the DMA of tile t+2 is issued under their latency, then the MFMAs run
            // behind counted lgkmcnt waits.  (Left to itself hipcc emits read, read, lgkmcnt(0), MFMA four times over: every
            // MFMA of the step paid a full LDS round trip, ~4 x 160 cycles per wave and step for 4 x 32 cycles of matrix work.)
            const unsigned char* At = lds + slot * STAGE;
            const unsigned char* Bt = At + A_BYTES;
            FragT a[BK / 16][TM], b[BK / 16][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = frag(At, wm0 + 32 * i, kk);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = frag(Bt, wn0 + 32 * j, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < t_end) issue(t + 2, slot == 0 ? 2 : slot - 1);     // slot of tile t-1 == (slot+2)%3
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[kk][i], b[kk][j], acc[i][j]);
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    if (LW > 0 && (p.dbg & 4)) return;                                  // (timing experiment: no epilogue)
    // ---- epilogue.  Everything the stores depend on is loaded FIRST: a load that sits between two stores cannot be
    // hoisted by the compiler (the output may alias it), so the per-row group scale used to serialise the tail into
    // 16-32 dependent load -> store round trips per wave (measured with s_memtime: a third of the workgroup's life).
    float* y32 = static_cast<float*>(p.y);
    T* yt = static_cast<T*>(p.y);
    const int ncols = MODE != 1 ? p.Cout : p.Cin;
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + 32 * j + (lane & 31);
        bcol[j] = (MODE != 1 && p.bias && col < ncols && (p.ksplit <= 1 || ks == 0)) ? p.bias[col] : 0.f;
    }
    float sc[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * i + crow(r, lane);
            sc[i][r] = 1.f;
            if (p.gscale && m < p.M)          // sample -> group without an integer division (exact below 2^21 samples)
                sc[i][r] = p.gscale[(int)(((float)(m >> p.lgHoWo) + 0.5f) * p.inv_group_n)];
        }
    if constexpr (ACTB) {
        // ---- activation-backward epilogue of the dgrad form (MODE 1, no K split, elementwise part only: the striped bias /
        // spectral-norm sums are the persistent form's).  The fp32 tile goes through the idle ring; a thread then owns 8
        // consecutive channels of one output pixel: 16 bytes of the stored activation in, 16 bytes of dzs out.
        constexpr int RS = BN * 4 + 16;
        constexpr int CPR = BN / 8;
        static_assert(BM * RS <= NSLOT * STAGE && MODE == 1, "tile must fit the ring");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + 32 * i + crow(r, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float*>(lds + row * RS + (wn0 + 32 * j + (lane & 31)) * 4) = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* ab = static_cast<const unsigned char*>(p.ab_a);
        unsigned char* yb = static_cast<unsigned char*>(p.y);
        int nsat = 0;
        for (int cix = threadIdx.x; cix < BM * CPR; cix += NCW * 64) {
            const int row = cix / CPR, ch = cix % CPR, m = m0 + row, col0 = n0 + ch * 8;
            if (m >= p.M || col0 >= p.Cin) continue;
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            const size_t pix = (size_t)(n * p.Hi + 2 * (rem >> p.lgWo) + py) * p.Wi + 2 * (rem & (Wo - 1)) + px;
            const float gs = p.gscale ? p.gscale[(int)(((float)n + 0.5f) * p.inv_group_n)] : 1.f;
            const uint4 aw = *reinterpret_cast<const uint4*>(ab + (pix * p.ab_lda + col0) * 2);
            const float4 v0 = *reinterpret_cast<const float4*>(lds + row * RS + ch * 32);
            const float4 v1 = *reinterpret_cast<const float4*>(lds + row * RS + ch * 32 + 16);
            const float da[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const unsigned awv[4] = {aw.x, aw.y, aw.z, aw.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = Bits16<T>::dec(awv[e >> 1] >> (16 * (e & 1)));
                o[e] = (av > 0.f ? da[e] : 0.2f * da[e]) * gs;
            }
            nsat += sat_hits<T>(o);
            uint4 w; w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
            *reinterpret_cast<uint4*>(yb + (pix * p.ldy + col0) * 2) = w;
        }
        sat_commit(p.ab_sat, nsat);
        return;
    }
    if constexpr (FIN) {
        // ---- InstanceNorm + activation epilogue (MODE 0, no K split; the host guarantees H*W <= 64 divides BM, 16-byte
        // aligned activation rows and Cout % 8 == 0).  LDS: fp32 tile [BM][BN] (+16 B row pad), then mean | rstd per
        // (sample of the tile, column).
        constexpr int RS = BN * 4 + 16;
        constexpr int CPR = BN / 8;                                        // 8-column chunks per tile row
        const int lgHW = p.lgHoWo, HW = 1 << lgHW, nsamp = BM >> lgHW;
        float* stat = reinterpret_cast<float*>(lds + BM * RS);             // [nsamp][BN][2]
        static_assert(BM * RS + (BM / 4) * BN * 8 <= NSLOT * STAGE, "tile + statistics must fit the ring");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // every wave is done with the ring
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + 32 * i + crow(r, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float*>(lds + row * RS + (wn0 + 32 * j + (lane & 31)) * 4) = acc[i][j][r] * sc[i][r] + bcol[j];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int s0 = m0 >> lgHW;                                         // first sample of the tile
        const float inv_hw = 1.f / (float)HW;
        for (int q = threadIdx.x; q < nsamp * BN; q += NCW * 64) {          // one (sample, column) per thread: column reads
            const int s = q / BN, c = q % BN;                              // of consecutive lanes are consecutive words
            const unsigned char* colp = lds + (s << lgHW) * RS + c * 4;
            float sum = 0.f;
            for (int r = 0; r < HW; ++r) sum += *reinterpret_cast<const float*>(colp + r * RS);
            const float mu = sum * inv_hw;
            float m2 = 0.f;
            for (int r = 0; r < HW; ++r) { const float d = *reinterpret_cast<const float*>(colp + r * RS) - mu; m2 += d * d; }
            const float rs = 1.0f / sqrtf(m2 * inv_hw + 1e-5f);
            stat[q * 2] = mu; stat[q * 2 + 1] = rs;
            const int n = s0 + s, cg = n0 + c;
            if (n < p.N && cg < p.Cout) { p.in_mean[(size_t)n * p.Cout + cg] = mu; p.in_rstd[(size_t)n * p.Cout + cg] = rs; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned char* ab = static_cast<unsigned char*>(p.y);
        unsigned char* pb = static_cast<unsigned char*>(p.in_apre);
        for (int cix = threadIdx.x; cix < BM * CPR; cix += NCW * 64) {
            const int row = cix / CPR, ch = cix % CPR, m = m0 + row, col0 = n0 + ch * 8;
            if (m >= p.M || col0 >= p.Cout) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(lds + row * RS + ch * 32);
            const float4 v1 = *reinterpret_cast<const float4*>(lds + row * RS + ch * 32 + 16);
            const float4* st4p = reinterpret_cast<const float4*>(stat + ((row >> lgHW) * BN + ch * 8) * 2);
            const float4 q0 = st4p[0], q1 = st4p[1], q2 = st4p[2], q3 = st4p[3];   // (mu, rs) x 8 columns
            float o[8] = {lrelu_f((v0.x - q0.x) * q0.y), lrelu_f((v0.y - q0.z) * q0.w), lrelu_f((v0.z - q1.x) * q1.y),
                          lrelu_f((v0.w - q1.z) * q1.w), lrelu_f((v1.x - q2.x) * q2.y), lrelu_f((v1.y - q2.z) * q2.w),
                          lrelu_f((v1.z - q3.x) * q3.y), lrelu_f((v1.w - q3.z) * q3.w)};
            if (pb) {
                const int n = m >> lgHW;
                if (n >= p.apre_n0) {
                    uint4 w; w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
                    *reinterpret_cast<uint4*>(pb + ((size_t)(m - (p.apre_n0 << lgHW)) * p.ld_apre + col0) * 2) = w;
                }
            }
            if (p.in_mask) {
                const uint2 mk = *reinterpret_cast<const uint2*>(p.in_mask + (size_t)m * p.Cout + col0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] *= ((mk.x >> (8 * e)) & 0xFF) ? 2.f : 0.f;
                    o[4 + e] *= ((mk.y >> (8 * e)) & 0xFF) ? 2.f : 0.f;
                }
            }
            uint4 w; w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
            *reinterpret_cast<uint4*>(ab + ((size_t)m * p.ldy + col0) * 2) = w;
        }
        return;
    }
    // ---- result tile through LDS (default): the MFMA C layout gives a lane one 4-byte element per row, i.e. 32 store
    // instructions of 2 x 128 bytes per wave for a 32 x 64 wave tile; after a transpose through the (now idle) ring every
    // store instruction writes 64 x 16 bytes of whole rows.  tools/archive/tile_ab.sh with GCSSL_RING_DEBUG=3 (no DMA, no MFMA:
    // launch + prologue + barriers + epilogue) put D.c3.fwd's skeleton at 14.5 of its 24 us.
    {
        const int es = (p.out_f32 || p.ksplit > 1) ? 4 : 2;                // element size of what is stored
        const bool rows16 = p.epi_lds && !(p.ksplit > 1 && !p.split_stride) && (p.ldy * es) % 16 == 0 &&
                            (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && (p.split_stride * 4) % 16 == 0 && ncols % (16 / es) == 0;
        if (rows16) {
            const int RS = BN * es + 16;                                   // padded row stride of the LDS tile
            static_assert(BM * (BN * 4 + 16) <= NSLOT * STAGE, "result tile must fit the ring");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // every wave is done with the ring
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm0 + 32 * i + crow(r, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int colt = wn0 + 32 * j + (lane & 31);
                        float v = acc[i][j][r] * sc[i][r] + bcol[j];
                        if (MODE == 0 && p.act == 1 && p.ksplit <= 1) v = lrelu_f(v);
                        if (es == 4) *reinterpret_cast<float*>(lds + row * RS + colt * 4) = v;
                        else *reinterpret_cast<unsigned short*>(lds + row * RS + colt * 2) = (unsigned short)Bits16<T>::enc(v);
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int cpr = BN * es / 16;                                  // 16-byte chunks per tile row (a power of two)
            unsigned char* yb = static_cast<unsigned char*>(p.y) + (p.ksplit > 1 ? (size_t)ks * p.split_stride * 4 : 0);
            for (int c = threadIdx.x; c < BM * cpr; c += NCW * 64) {
                const int row = c / cpr, ch = c % cpr, m = m0 + row, col0 = n0 + ch * (16 / es);
                if (m >= p.M || col0 >= ncols) continue;
                size_t pix = (size_t)m;
                if (MODE == 1) {
                    const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                    pix = (size_t)(n * p.Hi + 2 * (rem >> p.lgWo) + py) * p.Wi + 2 * (rem & (Wo - 1)) + px;
                }
                *reinterpret_cast<uint4*>(yb + (pix * p.ldy + col0) * es) = *reinterpret_cast<const uint4*>(lds + row * RS + ch * 16);
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * i + crow(r, lane);
            if (m >= p.M) continue;
            size_t pix = (size_t)m;
            if (MODE == 1) {
                const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                const int iy = 2 * (rem >> p.lgWo) + py, ix = 2 * (rem & (Wo - 1)) + px;
                pix = (size_t)(n * p.Hi + iy) * p.Wi + ix;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 32 * j + (lane & 31);
                if (col >= ncols) continue;
                float v = acc[i][j][r] * sc[i][r] + bcol[j];
                if (p.ksplit > 1) {
                    if (p.split_stride) y32[(size_t)ks * p.split_stride + pix * p.ldy + col] = v;
                    else atomicAdd(y32 + pix * p.ldy + col, v);
                    continue;
                }
                if (MODE == 0 && p.act == 1) v = lrelu_f(v);
                if (p.out_f32) y32[pix * p.ldy + col] = v;
                else Elem<T>::st(yt + pix * p.ldy + col, v);
            }
        }
#endif
}

// ------------------------------------------------------------------------------------------
// wgrad: slab[split][co][tap][ci] = sum_{k in split} dy[k][co] * x[n, 2oy-1+ky, 2ox-1+kx, ci],  k = (n,oy,ox)
// GEMM M = Cout, N = (tap, ci), K = N*Ho*Wo split over blockIdx.z.  blockIdx.y = tap * (Cin/BN) + ci-tile.
// Both operands arrive row(k)-major with channels contiguous -> MMajor tiles, transposed LDS reads.
// ------------------------------------------------------------------------------------------
// SMALLC (first layers, Cin padded to 8): the N tile is all 16 taps x 8 channels (BN must be 128).
template <typename T, int BM, int BN, bool SMALLC, int KS = 4, int MM = 0, int WM = 2, int WN = 2>
__global__ __launch_bounds__(WM * WN * 64, MM ? (WM * WN >= 8 ? 4 : 2) : 1) void conv_wgrad_kernel(ConvParams p) {
    typedef Geo<KS> G;
    constexpr int NT = WM * WN * 64;                          // (4 x 2 waves: the split-precision forms, see conv_fwd_kernel)
    constexpr int BK = BKOf<T>::v;
    constexpr int KV = Elem<T>::KV;
    static_assert(!SMALLC || BN == 128, "SMALLC covers 16 taps x 8 channels");
    constexpr int CHA = BM / KV, CHB = BN / KV;               // vectors per k-row
    constexpr int NVA = BK * CHA / NT, NVB = BK * CHB / NT;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for this many threads");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile too small for this many waves");
    __shared__ typename MTile<T, BM, MM>::type As[2];
    __shared__ typename MTile<T, BN, MM>::type Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // All (co-tile, tap, ci-tile) workgroups of one K split read the same dy rows and overlapping x rows.  In dispatch order
    // (x, then y, then z) they are consecutive and therefore spread over the 8 XCDs, each of which fetches its own copy:
    // PMC FETCH_SIZE 197 MB per launch of D.c2.wgrad against 50 MB of operands, at 5 TB/s the launch was HBM-bound.
    // Re-deal the linear workgroup index so that a split's workgroups are consecutive on ONE XCD (index = 8*(per*q + t) + xcd
    // is split 8q + xcd, tile t of per), sharing its L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, per = gx * gridDim.y, ns = gridDim.z;      // per: workgroups of one K split
        const int L = bx + gx * (by + (int)gridDim.y * bz), full = (ns >> 3) * 8 * per;
        int t;
        if (L < full) { const int j = L >> 3; bz = (j / per) * 8 + (L & 7); t = j % per; }
        else { const int r = L - full; bz = (ns & ~7) + r / per; t = r % per; }
        bx = t % gx; by = t / gx;
    }
    const int co0 = bx * BM;
    const int ntile_ci = SMALLC ? 1 : p.Cin / BN;
    const int tap = SMALLC ? 0 : by / ntile_ci, ci0 = SMALLC ? 0 : (by % ntile_ci) * BN;
    const int ky = G::ky(tap), kx = G::kx(tap);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const T* x = static_cast<const T*>(p.x);
    const T* dy = static_cast<const T*>(p.w);
    const int Wo = p.Wi / G::ST;
    const int Ktot = p.M;                                      // N*Ho*Wo
    const int kt_beg = bz * p.ktiles_per_split;
    int kt_end = kt_beg + p.ktiles_per_split;
    const int nkt = (Ktot + BK - 1) / BK;
    if (kt_end > nkt) kt_end = nkt;

    constexpr int ES = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), dr = make_rsrc(p.w, p.w_bytes);
    Vec16<T> ra[MM ? 2 : 1][NVA], rb[MM ? 2 : 1][NVB];
    auto gload = [&](Vec16<T> (&qa)[NVA], Vec16<T> (&qb)[NVB], int k0, bool live = true) {      // (live: see conv_fwd_kernel)
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int v = tid + i * NT, kr = v / CHA, c = v % CHA;
            const int k = k0 + kr;                                 // rows k >= Ktot fall outside the dy buffer -> 0
            qa[i] = bload<T>(dr, (live && k < Ktot) ? (unsigned)((k * p.ldw + co0 + c * KV) * ES) : OOB);
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * NT, kr = v / CHB, c = v % CHB;
            const int k = k0 + kr;
            const int n = k >> p.lgHoWo, rem = k & ((1 << p.lgHoWo) - 1);
            int kyy = ky, kxx = kx, coff = ci0 + c * KV;
            bool tap_ok = true;
            if (SMALLC) { const int tp = (c * KV) >> 3; kyy = G::ky(tp); kxx = G::kx(tp); coff = (c * KV) & 7; tap_ok = tp < G::TAPS; }
            const int iy = G::ST * (rem >> p.lgWo) - 1 + kyy, ix = G::ST * (rem & (Wo - 1)) - 1 + kxx;
            const bool ok = live && tap_ok && k < Ktot && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            qb[i] = bload<T>(xr, ok ? (unsigned)((((n * p.Hi + iy) * p.Wi + ix) * p.ldx + coff) * ES) : OOB);
        }
    };
    auto lstore = [&](const Vec16<T> (&qa)[NVA], const Vec16<T> (&qb)[NVB], int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) { const int v = tid + i * NT; As[buf].store_vec(v / CHA, v % CHA, qa[i]); }
#pragma unroll
        for (int i = 0; i < NVB; ++i) { const int v = tid + i * NT; Bs[buf].store_vec(v / CHB, v % CHB, qb[i]); }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (MM != 0) {                                   // two K tiles of loads in flight (see conv_fwd_kernel)
        if (kt_beg < kt_end) {
            gload(ra[0], rb[0], kt_beg * BK);
            gload(ra[1], rb[1], (kt_beg + 1) * BK, kt_beg + 1 < kt_end);
            lstore(ra[0], rb[0], 0); __syncthreads();
            for (int t = kt_beg; t < kt_end; t += 2) {
                gload(ra[0], rb[0], (t + 2) * BK, t + 2 < kt_end);
                mma_slab<TM, TN>(As[0], Bs[0], wm0, wn0, lane, acc);
                lstore(ra[1], rb[1], 1);
                __syncthreads();
                gload(ra[1], rb[1], (t + 3) * BK, t + 3 < kt_end);
                if (t + 1 < kt_end) mma_slab<TM, TN>(As[1], Bs[1], wm0, wn0, lane, acc);
                lstore(ra[0], rb[0], 0);
                __syncthreads();
            }
        }
    } else
    if (kt_beg < kt_end) {
        gload(ra[0], rb[0], kt_beg * BK); lstore(ra[0], rb[0], 0); __syncthreads();
        for (int t = kt_beg; t < kt_end; ++t) {
            const int b = (t - kt_beg) & 1;
            if (t + 1 < kt_end) gload(ra[0], rb[0], (t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < kt_end) lstore(ra[0], rb[0], b ^ 1);
            __syncthreads();
        }
    }
    float* slab = static_cast<float*>(p.y) + (size_t)bz * p.Cout * 16 * p.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * i + crow(r, lane);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + wn0 + 32 * j + (lane & 31);   // SMALLC: ci is the packed (tap*8 + channel) column
                slab[((size_t)co * 16 + tap) * p.Cin + ci] = MM ? acc[i][j][r] * p.mm_oscale : acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------
// Persistent form of conv_dma_kernel for layers with more tiles than resident workgroups (G.up4, D.c1/c2: 768-2048
// tiles of 8-16 K steps).  There a workgroup's life is 1/4 gather setup + first-tile latency, 1/2 K loop, 1/4 store tail
// (s_memtime stamps), and every round of workgroups pays all three.  Here a workgroup walks tiles
// t, t+G, t+2G, ... and the DMA ring never drains: the last two K steps of a tile already fetch the first two of the
// next one (its gather offsets are computed while the current tile's loads are in flight), and the epilogue stores are
// not waited for.  vmcnt bookkeeping across the tile boundary: loads, stores and LDS-DMA retire in issue order, so
//   before the epilogue    vmcnt(NL)        -> step 0 of the next tile has landed (its step 1 may be in flight)
//   next tile, step 1      vmcnt(NL + NST)  -> skips this wave's NST epilogue stores and the step-2 loads behind it
//   next tile, step >= 2   vmcnt(NL)        -> as inside a tile (the stores are older and therefore complete)
// which needs a FIXED number of store instructions per tile: the epilogue is branch-free buffer stores whose
// out-of-range lanes carry an out-of-bounds offset (dropped by the hardware, still counted).  No split-K here.
// ------------------------------------------------------------------------------------------
// FIN (forward conv, 8x8 output maps, 32 x 32 wave tiles): InstanceNorm + LeakyReLU in the epilogue, statistics IN REGISTERS --
// the ring is busy with the next tile, so there is no LDS tile to stage through.  A 32-row C block lies inside one sample
// (64 rows), a lane holds 16 rows of one column: two-pass mean / M2 over the block (16 registers + one xor-32 shuffle), Chan's
// combination with the partner wave that holds the sample's other 32 rows (2 floats per column through 2 KB of LDS, one
// barrier), then the 16 stores of the 16-bit activation + mean / rstd (two more store instructions for every wave, out-of-range
// for the non-writers: the vmcnt bookkeeping needs a fixed count).
// ACTB (dgrad form, one N tile: Cin == BN): the activation backward of the norm-less layer that produced this conv's input in the
// epilogue -- per lane 16 two-byte loads of the stored activation at its output pixels, dzs = lrelu'(a) dx gscale stored in the
// compute dtype, and the bias-gradient / spectral-norm sums kept in registers across the workgroup's tiles (its columns never
// change: one N tile) and added to one replica of the striped sums when the workgroup retires.
template <typename T, int BM, int BN, int MODE, int WM, int WN, bool SMALLK = false, bool FIN = false, bool ACTB = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_dma_persist_kernel(ConvParams p, int tiles_m, int tiles_n, int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    constexpr int NTH = WM * WN * 64;
    constexpr int BK = 64, ES = 2, KV = 8, CH = 8, RPT = NTH / CH;
    constexpr int NVA = BM / RPT, NVB = BN / RPT, NL = NVA + NVB;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for this many waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(!FIN || (TM == 1 && TN == 1 && MODE == 0 && WM % 2 == 0), "FIN: one 32 x 32 C block per wave, wave rows pair up");
    constexpr int NST = TM * TN * 16 + (FIN ? 2 : 0);                    // epilogue store instructions per wave per tile
    static_assert(NL + NST <= 63, "vmcnt is a 6-bit counter");
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE + (FIN ? WM * WN * 32 * 8 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int Ho = MODE == 2 ? p.Hi : p.Hi >> 1, Wo = MODE == 2 ? p.Wi : p.Wi >> 1;
    const int K = MODE == 0 ? 16 * p.Cin : MODE == 2 ? p.wk : 4 * p.Cout;
    const int nk = (MODE == 2 && p.kcap >= 2 && p.kcap < K / BK) ? p.kcap : K / BK;
    const int ncols = MODE != 1 ? p.Cout : p.Cin;
    const int row_t = tid / CH;
    const int lc = (tid % CH) ^ ((row_t >> 1) & 7);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes), yr = make_rsrc(p.y, p.y_bytes);

    struct Addr { int rowoff[NVA]; unsigned rowmask[NVA]; unsigned wrow[NVB]; int m0, n0, py, px; };
    auto setup = [&](int tile, Addr& a) {
        int mx, ny, cls = 0;
        if (p.sib_remap) {
            tile_decode(tile, tiles_m, tiles_n, MODE == 1 ? 4 : 1, mx, ny, cls);
        } else if (MODE == 1 && p.class_major == 0) {
            // The four output-parity classes of a (m, n) tile read the SAME input rows.  Tile t runs on XCD t % 8 (workgroups
            // are dealt round-robin to the XCDs and the grid is a multiple of 8), so blocks of 32 consecutive tiles are
            // 8 groups x 4 classes with a group's classes 8 apart: back to back on one XCD, sharing its L2.  In class-major
            // order (all tiles of class 0, then class 1, ...) a 50 MB input was streamed from HBM once per class: PMC
            // FETCH_SIZE 270 MB per launch of G.up4.fwd at 768 samples against 50 MB of input.
            const int groups = tiles_m * tiles_n, full = (groups >> 3) << 5;
            int g;
            if (tile < full) { const int r = tile & 31; cls = r >> 3; g = ((tile >> 5) << 3) + (r & 7); }
            else { const int r = tile - full; g = (groups & ~7) + (r >> 2); cls = r & 3; }
            mx = g % tiles_m; ny = g / tiles_m;
        } else {
            mx = tile % tiles_m;
            const int rest = tile / tiles_m;
            ny = rest % tiles_n; cls = MODE == 1 ? rest / tiles_n : 0;
        }
        a.m0 = mx * BM; a.n0 = ny * BN; a.py = cls >> 1; a.px = cls & 1;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int m = a.m0 + row_t + i * RPT;
            a.rowoff[i] = 0; a.rowmask[i] = 0;
            if (m < p.M) {
                const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                unsigned mk = 0;
                if (MODE == 0) {
                    const int iy0 = 2 * (rem >> p.lgWo) - 1, ix0 = 2 * (rem & (Wo - 1)) - 1;
                    a.rowoff[i] = ((n * p.Hi + iy0) * p.Wi + ix0) * p.ldx * ES;
                    mk = tap_mask16(iy0, ix0, p.Hi, p.Wi);
                } else if (MODE == 2) {                                    // 3x3 stride 1 pad 1 (Geo<3>)
                    const int iy0 = (rem >> p.lgWo) - 1, ix0 = (rem & (Wo - 1)) - 1;
                    a.rowoff[i] = ((n * p.Hi + iy0) * p.Wi + ix0) * p.ldx * ES;
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        if ((unsigned)(iy0 + Geo<3>::ky(t)) < (unsigned)p.Hi && (unsigned)(ix0 + Geo<3>::kx(t)) < (unsigned)p.Wi) mk |= 1u << t;
                } else {
                    const int yy = (rem >> p.lgWo) + a.py, xx = (rem & (Wo - 1)) + a.px;
                    a.rowoff[i] = ((n * Ho + yy) * Wo + xx) * p.ldx * ES;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if ((unsigned)(yy - (t >> 1)) < (unsigned)Ho && (unsigned)(xx - (t & 1)) < (unsigned)Wo) mk |= 1u << t;
                }
                a.rowmask[i] = mk;
                if (!SMALLK) a.rowoff[i] += lc * KV * ES;
            }
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            const int r = a.n0 + row_t + j * RPT;
            if (MODE != 1) a.wrow[j] = r < p.Cout ? (unsigned)(r * K * ES) : OOB;
            else a.wrow[j] = r < p.Cin ? (unsigned)(r * 16 * p.Cout * ES) : OOB;
            if (!SMALLK && a.wrow[j] != OOB) a.wrow[j] += lc * KV * ES;
        }
    };
    int vstride = NTH * 16, a_bytes = A_BYTES, stage = STAGE;
    auto issue = [&](const Addr& a, int t, int slot) {
        unsigned char* base = lds + slot * stage + wave * 1024;
        int tapbit, tapoff; unsigned woff;
        const int kb = SMALLK ? t * BK + lc * KV : t * BK;             // !SMALLK: wave-uniform -> SALU
        if (MODE == 0) {
            const int tap = kb >> p.lgCin, ci = kb & (p.Cin - 1);
            tapbit = tap; tapoff = (((tap >> 2) * p.Wi + (tap & 3)) * p.ldx + ci) * ES; woff = (unsigned)(kb * ES);
        } else if (MODE == 2) {
            const int tap = kb >> p.lgCin, ci = kb & (p.Cin - 1);
            tapbit = tap; tapoff = ((Geo<3>::ky(tap) * p.Wi + Geo<3>::kx(tap)) * p.ldx + ci) * ES; woff = (unsigned)(kb * ES);
        } else {
            const int t4 = kb >> p.lgCout, co = kb & (p.Cout - 1);
            const int ty = t4 >> 1, tx = t4 & 1;
            const int tap = (1 - a.py + 2 * ty) * 4 + (1 - a.px + 2 * tx);
            tapbit = t4; tapoff = (co - (ty * Wo + tx) * p.ldx) * ES; woff = (unsigned)((tap * p.Cout + co) * ES);
        }
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(base + i * vstride), 16,
                ((a.rowmask[i] >> tapbit) & 1u) ? (unsigned)(a.rowoff[i] + tapoff) : OOB, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NVB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_p)(base + a_bytes + j * vstride), 16, a.wrow[j] + woff, 0, 0, 0);
    };
    auto frag = [&](const unsigned char* tile, int row0, int kstep) -> FragT {
        const int row = row0 + (lane & 31), c = kstep * 2 + (lane >> 5);
        return __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)));
    };

    int tile = blockIdx.x;
    Addr cur, nxt;
    setup(tile, cur);
    issue(cur, 0, 0);
    issue(cur, 1, 1);                                                    // nk >= 2 for every layer that gets here
    int slot = 0;
    bool first = true;
    float ab_sb[TN], ab_sd[4] = {0.f, 0.f, 0.f, 0.f};                    // ACTB: running column sums of dz, per-group sums of dzs (z - b)
    int ab_nsat = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) ab_sb[j] = 0.f;
    while (true) {
        const int next_tile = tile + (int)gridDim.x;
        const bool has_next = next_tile < total_tiles;
        if (has_next) setup(next_tile, nxt);                             // VALU work under the loads already in flight
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int t = 0; t < nk; ++t) {
            const bool younger = (t + 1 < nk) || has_next;               // is the following step's tile in flight?
            if (t == 0 && !first) { /* waited for before the previous epilogue */ }
            else if (t == 1 && !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + NST) : "memory");
            else if (younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (reads first, DMA issue under their latency, MFMAs behind counted lgkmcnt waits: see conv_dma_kernel)
            const unsigned char* At = lds + slot * STAGE;
            const unsigned char* Bt = At + A_BYTES;
            FragT a[BK / 16][TM], b[BK / 16][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = frag(At, wm0 + 32 * i, kk);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = frag(Bt, wn0 + 32 * j, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int s2 = slot == 0 ? 2 : slot - 1;                      // slot of step t+2 == slot of step t-1
            if (t + 2 < nk) issue(cur, t + 2, s2);
            else if (has_next) issue(nxt, t + 2 - nk, s2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma(a[kk][i], b[kk][j], acc[i][j]);
            slot = slot == 2 ? 0 : slot + 1;
        }
        // step 0 of the next tile must have landed before this wave's stores enter the queue behind it
        if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        // ---- epilogue: loads first (bias, 1/sigma of the row's sample group), then exactly NST buffer stores
        float bcol[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = cur.n0 + wn0 + 32 * j + (lane & 31);
            bcol[j] = (MODE != 1 && p.bias && col < ncols) ? p.bias[col] : 0.f;
        }
        float sc[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = cur.m0 + wm0 + 32 * i + crow(r, lane);
                sc[i][r] = 1.f;
                if (p.gscale && m < p.M) sc[i][r] = p.gscale[(int)(((float)(m >> p.lgHoWo) + 0.5f) * p.inv_group_n)];
            }
        if (p.gscale || (MODE != 1 && p.bias)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see note below
        if constexpr (ACTB) {
            static_assert(MODE == 1 && !FIN, "ACTB is an epilogue of the dgrad form");
            const __amdgpu_buffer_rsrc_t ar = make_rsrc(p.ab_a, p.ab_bytes);
            float bb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = cur.n0 + wn0 + 32 * j + (lane & 31);
                bb[j] = (p.ab_bias && col < ncols) ? p.ab_bias[col] : 0.f;
            }
            const int grp = (int)(((float)(cur.m0 >> p.lgHoWo) + 0.5f) * p.inv_group_n);   // (a tile never straddles sample groups)
            float sdt = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                unsigned short araw[16][TN];
                unsigned offs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {                               // all activation loads of the block first
                    const int m = cur.m0 + wm0 + 32 * i + crow(r, lane);
                    const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                    const int iy = 2 * (rem >> p.lgWo) + cur.py, ix = 2 * (rem & (Wo - 1)) + cur.px;
                    offs[r] = m < p.M ? (unsigned)((n * p.Hi + iy) * p.Wi + ix) : OOB;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = cur.n0 + wn0 + 32 * j + (lane & 31);
                        araw[r][j] = __builtin_amdgcn_raw_buffer_load_b16(ar, (offs[r] != OOB && col < ncols) ? (offs[r] * (unsigned)p.ab_lda + (unsigned)col) * 2u : OOB, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = cur.n0 + wn0 + 32 * j + (lane & 31);
                        const bool ok = offs[r] != OOB && col < ncols;
                        const float av = Bits16<T>::dec(araw[r][j]);
                        const float dx = acc[i][j][r];
                        const float dz = ok ? (av > 0.f ? dx : 0.2f * dx) : 0.f;
                        const float zv = av > 0.f ? av : 5.0f * av;              // invert LeakyReLU(0.2)
                        const float o = dz * sc[i][r];
                        ab_sb[j] += dz; sdt += o * (zv - bb[j]);
                        ab_nsat += sat_hit<T>(o);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Bits16<T>::enc(o), yr,
                                                              ok ? (offs[r] * (unsigned)p.ldy + (unsigned)col) * 2u : OOB, 0, 0);
                    }
            }
            if (grp == 0) ab_sd[0] += sdt; else if (grp == 1) ab_sd[1] += sdt; else if (grp == 2) ab_sd[2] += sdt; else ab_sd[3] += sdt;
        } else if constexpr (FIN) {
            float v[16], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[r] = acc[0][0][r] * sc[0][r] + bcol[0]; sum += v[r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float mw = sum * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = v[r] - mw; m2 += d * d; }
            m2 += __shfl_xor(m2, 32, 64);
            float* xs = reinterpret_cast<float*>(lds + 3 * STAGE);         // [wave][32 columns][mean, M2]
            if (lane < 32) { xs[(wave * 32 + lane) * 2] = mw; xs[(wave * 32 + lane) * 2 + 1] = m2; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (the next write to xs is a whole K loop of barriers away: no second barrier needed)
            const float pm = xs[(((wave ^ WN) * 32) + (lane & 31)) * 2], pm2 = xs[(((wave ^ WN) * 32) + (lane & 31)) * 2 + 1];
            const float mean = 0.5f * (mw + pm), dlt = mw - pm;
            const float rstd = 1.0f / sqrtf((m2 + pm2 + dlt * dlt * 16.f) * (1.f / 64.f) + 1e-5f);
            const int col = cur.n0 + wn0 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = cur.m0 + wm0 + crow(r, lane);
                const bool ok = m < p.M && col < ncols;
                const float o = lrelu_f((v[r] - mean) * rstd);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Bits16<T>::enc(o), yr,
                                                      ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)col) * 2u : OOB, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t mr = make_rsrc(p.in_mean, (unsigned)p.N * (unsigned)p.Cout * 4u),
                                         rr = make_rsrc(p.in_rstd, (unsigned)p.N * (unsigned)p.Cout * 4u);
            const int n = (cur.m0 + wm0) >> 6;
            const bool wr = ((wave / WN) & 1) == 0 && lane < 32 && n < p.N && col < ncols;
            const unsigned so = wr ? ((unsigned)n * (unsigned)p.Cout + (unsigned)col) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mean), mr, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd), rr, so, 0, 0);
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = cur.m0 + wm0 + 32 * i + crow(r, lane);
                unsigned pix = (unsigned)m;
                if (MODE == 1) {
                    const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
                    const int iy = 2 * (rem >> p.lgWo) + cur.py, ix = 2 * (rem & (Wo - 1)) + cur.px;
                    pix = (unsigned)((n * p.Hi + iy) * p.Wi + ix);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = cur.n0 + wn0 + 32 * j + (lane & 31);
                    const bool ok = m < p.M && col < ncols;
                    float v = acc[i][j][r] * sc[i][r] + bcol[j];
                    if (MODE == 0 && p.act == 1) v = lrelu_f(v);
                    const unsigned e = pix * (unsigned)p.ldy + (unsigned)col;
                    if (p.out_f32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, ok ? e * 4u : OOB, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Bits16<T>::enc(v), yr, ok ? e * 2u : OOB, 0, 0);
                }
            }
        }
        if (!has_next) break;
        cur = nxt; tile = next_tile; first = false;
    }
    if constexpr (ACTB) {
        // the workgroup's sums -> one replica of the striped bias-gradient / spectral-norm sums (norm.hip replica_offset)
        sat_commit(p.ab_sat, ab_nsat);
        if (p.ab_dbias || p.ab_cdot) {
            float* red = reinterpret_cast<float*>(lds);                      // [wave][TN * 32 + 4]
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
            constexpr int RW = TN * 32 + 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float t = ab_sb[j] + __shfl_xor(ab_sb[j], 32, 64);
                if (lane < 32) red[wave * RW + j * 32 + lane] = t;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float t = wave_sum(ab_sd[g]);
                if (lane == 0) red[wave * RW + TN * 32 + g] = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int rep = p.ab_nrep > 1 ? (int)(blockIdx.x % (unsigned)p.ab_nrep) * p.ab_rep_stride : 0;
            if (p.ab_dbias && tid < BN) {                                    // column tid of the tile: wave column tid / (BN / WN)
                const int wc = tid / (BN / WN), cj = tid % (BN / WN);
                float t = 0.f;
#pragma unroll
                for (int wr = 0; wr < WM; ++wr) t += red[(wr * WN + wc) * RW + cj];
                if (tid < ncols) atomicAdd(p.ab_dbias + rep + tid, t);
            }
            if (p.ab_cdot && tid < 4) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM * WN; ++w) t += red[w * RW + TN * 32 + tid];
                if (t != 0.f) atomicAdd(p.ab_cdot + rep + tid, t);
            }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// wgrad on the LDS-DMA ring (16-bit types, Cin a multiple of 64, Cout a multiple of 128): the same contraction as
// conv_wgrad_kernel, but a workgroup owns FOUR taps (one filter row ky, kx = 0..3) of a 128 (co) x 64 (ci) block and the dy
// tile is shared by them: 8 waves as tap x co-half, wave tile 64 x 64, BK = 64 output pixels per K step.
//
// Why: every conv kernel of this file runs at the rate its LDS fills allow -- bytes in flight per CU / latency (~96 KB /
// 2.3 us = 41 GB/s per CU under load, measured on every variant: register staging or LDS-DMA, 3- or 6-slot rings, L2-local
// tile orders made no difference).  One tap per workgroup moves 24 KB per 1.05 MFLOP; four taps move
// 16 KB (dy) + 4 x 8 KB (x) per 4.2 MFLOP: half the bytes per FLOP.  The four x gathers of a filter row are neighbouring
// pixels of the same input row.
//
// Both operands go global -> LDS by `buffer_load ... lds` into a 3-slot ring (tiles t+1, t+2 in flight behind a counted vmcnt
// + raw s_barrier); the [k][row] images are read transposed with ds_read_b64_tr_b16.  An LDS-DMA instruction writes 1 KB
// lane-linearly, so the images cannot be padded; bank conflicts of the transposed reads are removed by XOR swizzles applied
// to the SOURCE chunk each lane fetches and to the read address:
//   A = dy tile [64 k][128 co] (256-B rows, 16 chunks of 16 B):  chunk ^ (((k & 3) << 2) | ((k >> 2) & 3))
//       (cdna_hip_programming.md T10 image (b): conflict-free for row reads and transposed reads)
//   B = x tiles [4 taps][64 k][64 ci] (128-B rows, 8 chunks):    chunk ^ (((k >> 1) & 1) << 2)
//       (swaps the 64-B halves of k rows 2, 3 mod 4: the four k rows a 32-lane half reads then cover the four 64-B phases of
//        a 256-B bank row)
// ------------------------------------------------------------------------------------------
// (the body is a device function: conv_wgrad_dma_kernel runs it for one layer, conv_wgrad_dma_batch_kernel for up to three
// layers in ONE launch -- the weight gradients of a backward pass do not depend on each other)
template <typename T, int LW>
__device__ __forceinline__ void wgrad_dma_body(const ConvParams& p, int bx, int by, int bz, const int gdx, const int gdy, const int gdz) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    // LW = 8: the 8 waves of the tile only read fragments and run MFMAs, 8 more waves own the LDS-DMA (a piece costs ~105
    // cycles of issue: 6 of them per wave and step sat in front of each wave's 16 MFMAs)
    constexpr int BM = 128, BN = 64, BK = 64, NLA = 2, NLB = 4, NL = NLA + NLB;      // DMA instructions per fetching wave per K tile
    constexpr int A_BYTES = BK * BM * 2, B_TAP = BK * BN * 2, STAGE = A_BYTES + 4 * B_TAP;   // 16 KB + 4 x 8 KB
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= 8;
    const int fwave = LW ? wave - 8 : wave;                             // index among the fetching waves
    if (p.xcd_remap) {                                                  // a K split's workgroups on one XCD (see conv_wgrad_kernel)
        const int gx = gdx, per = gx * gdy, ns = gdz;
        const int L = bx + gx * (by + gdy * bz), full = (ns >> 3) * 8 * per;
        int t;
        if (L < full) { const int j = L >> 3; bz = (j / per) * 8 + (L & 7); t = j % per; }
        else { const int r = L - full; bz = (ns & ~7) + r / per; t = r % per; }
        bx = t % gx; by = t / gx;
    }
    const int co0 = bx * BM;
    const int ntile_ci = p.Cin / BN;
    const int ky = by / ntile_ci, ci0 = (by % ntile_ci) * BN;           // the workgroup's filter row and input-channel block
    const int wtap = wave >> 1, wm0 = (wave & 1) * 64;                  // the wave's tap kx and co half
    const int Wo = p.Wi >> 1, Ktot = p.M;
    const int kt_beg = bz * p.ktiles_per_split;
    int kt_end = kt_beg + p.ktiles_per_split;
    const int nkt = (Ktot + BK - 1) / BK;
    if (kt_end > nkt) kt_end = nkt;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), dr = make_rsrc(p.w, p.w_bytes);

    // ---- DMA.  A piece j = 2*wave + jj (0..15): k rows 4j + (lane>>4), physical chunk lane&15.
    //            B piece id = 4*wave + jj (0..31): tap kx = id>>3, k rows 8 (id&7) + (lane>>3), physical chunk lane&7.
    auto issue = [&](int t, int slot) {
        unsigned char* base = lds + slot * STAGE;
        const int k0 = t * BK;
#pragma unroll
        for (int jj = 0; jj < NLA; ++jj) {
            const int j = 2 * fwave + jj, kr = 4 * j + (lane >> 4), k = k0 + kr;
            const int lc = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_void_p)(base + j * 1024), 16,
                k < Ktot ? (unsigned)((k * p.ldw + co0 + lc * 8) * 2) : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < NLB; ++jj) {
            const int id = 4 * fwave + jj, kx = id >> 3, j = id & 7, kr = 8 * j + (lane >> 3), k = k0 + kr;
            const int lc = (lane & 7) ^ (((kr >> 1) & 1) << 2);
            const int n = k >> p.lgHoWo, rem = k & ((1 << p.lgHoWo) - 1);
            const int iy = 2 * (rem >> p.lgWo) - 1 + ky, ix = 2 * (rem & (Wo - 1)) - 1 + kx;
            const bool ok = k < Ktot && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(base + A_BYTES + id * 1024), 16,
                ok ? (unsigned)((((n * p.Hi + iy) * p.Wi + ix) * p.ldx + ci0 + lc * 8) * 2) : OOB, 0, 0, 0);
        }
    };
    // ---- transposed fragment reads (ds_read_b64_tr_b16: per 16-lane group a 4 (k) x 16 (row) block, column-major out).
    // Lane 4q+pp of group g supplies k row kb = 16 ks + 8 (g>>1) + q (and kb + 4 for the second half of the fragment),
    // rows r0 + 16 (g&1) + 4 pp .. +3.  The swizzle keys do not depend on ks (16 ks = 0 mod 16).
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3, hq = 8 * (g >> 1) + q;     // kb = 16 ks + hq
    const int keyA0 = ((hq & 3) << 2) | ((hq >> 2) & 3), keyA1 = (((hq + 4) & 3) << 2) | (((hq + 4) >> 2) & 3);
    const int keyB = ((hq >> 1) & 1) << 2;                              // (hq + 4 has the same bit 1)
    int aoff[2][2], boff[2][2];                                         // byte offsets inside a slot for ks = 0: [block][lo/hi]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cb = (wm0 + 32 * i + 16 * (g & 1) + 4 * pp) * 2;     // byte column of the lane's 4 co rows
        aoff[i][0] = 256 * hq + 16 * ((cb >> 4) ^ keyA0) + (cb & 15);
        aoff[i][1] = 256 * (hq + 4) + 16 * ((cb >> 4) ^ keyA1) + (cb & 15);
        const int cc = (32 * i + 16 * (g & 1) + 4 * pp) * 2;           // ... of its 4 ci columns, in the wave's tap image
        boff[i][0] = A_BYTES + wtap * B_TAP + 128 * hq + 16 * ((cc >> 4) ^ keyB) + (cc & 15);
        boff[i][1] = A_BYTES + wtap * B_TAP + 128 * (hq + 4) + 16 * ((cc >> 4) ^ keyB) + (cc & 15);
    }
    auto trfrag = [&](const unsigned char* lo, const unsigned char* hi) -> FragT {
        const s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lo));
        const s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(hi));
        const s16x8 r = {l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
        return __builtin_bit_cast(FragT, r);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (LW > 0) {
        if (loader) {
            if (kt_beg < kt_end) issue(kt_beg, 0);
            if (kt_beg + 1 < kt_end) issue(kt_beg + 1, 1);
            int slot = 0;
            for (int t = kt_beg; t < kt_end; ++t) {
                if (t + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                           // tile t published, tile t-1 released
                if (t + 2 < kt_end) issue(t + 2, slot == 0 ? 2 : slot - 1);
                slot = slot == 2 ? 0 : slot + 1;
            }
            return;                                                     // (the epilogue has no barrier)
        }
        int slot = 0;
        for (int t = kt_beg; t < kt_end; ++t) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* S = lds + slot * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                            // fragments per K sub-step: 16 live registers, not 64
                FragT a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    b[i] = trfrag(S + boff[i][0] + ks * 2048, S + boff[i][1] + ks * 2048);
                    a[i] = trfrag(S + aoff[i][0] + ks * 4096, S + aoff[i][1] + ks * 4096);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else if (kt_beg < kt_end) {
        issue(kt_beg, 0);
        if (kt_beg + 1 < kt_end) issue(kt_beg + 1, 1);
        int slot = 0;
        for (int t = kt_beg; t < kt_end; ++t) {
            if (t + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* S = lds + slot * STAGE;
            FragT a[4][2], b[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    b[ks][i] = trfrag(S + boff[i][0] + ks * 2048, S + boff[i][1] + ks * 2048);
                    a[ks][i] = trfrag(S + aoff[i][0] + ks * 4096, S + aoff[i][1] + ks * 4096);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < kt_end) issue(t + 2, slot == 0 ? 2 : slot - 1);     // the slot of tile t-1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma(a[ks][i], b[ks][j], acc[i][j]);
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    float* slab = static_cast<float*>(p.y) + (size_t)bz * p.Cout * 16 * p.Cin;
    const int tap = ky * 4 + wtap;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * i + crow(r, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                slab[((size_t)co * 16 + tap) * p.Cin + ci0 + 32 * j + (lane & 31)] = acc[i][j][r];
        }
#endif
}
template <typename T, int LW = 0>
__global__ __launch_bounds__(512 + LW * 64) void conv_wgrad_dma_kernel(ConvParams p) {
    wgrad_dma_body<T, LW>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
}
struct WgradBatch { ConvParams p[3]; int first[4]; int gx[3], gy[3], gz[3]; int n; };
template <typename T, int LW = 0>
__global__ __launch_bounds__(512 + LW * 64) void conv_wgrad_dma_batch_kernel(WgradBatch b) {
    const int L0 = blockIdx.x;
    const int l = (b.n > 2 && L0 >= b.first[2]) ? 2 : ((b.n > 1 && L0 >= b.first[1]) ? 1 : 0);     // (workgroup-uniform)
    const int L = L0 - b.first[l], gx = b.gx[l], gy = b.gy[l];
    if (l == 0) wgrad_dma_body<T, LW>(b.p[0], L % gx, (L / gx) % gy, L / (gx * gy), gx, gy, b.gz[0]);
    else if (l == 1) wgrad_dma_body<T, LW>(b.p[1], L % gx, (L / gx) % gy, L / (gx * gy), gx, gy, b.gz[1]);
    else wgrad_dma_body<T, LW>(b.p[2], L % gx, (L / gx) % gy, L / (gx * gy), gx, gy, b.gz[2]);
}

// sum the split-K slabs, apply the spectral-norm rank-1 corrections, write PyTorch layout
//   dw[co][ci][tap] (+)= sum_s slab[s][co][tap][ci]  - sum_k coef[k]*cscale[k] u_k[co] v_k[ci*16+tap]
// One workgroup = one co x 64 ci x 16 taps: slab reads are coalesced along ci, the [tap][ci] -> [ci][tap] transpose
// goes through LDS, and the 4-KB output chunk dw[co][ci0..ci0+63][0..15] is written contiguously.
// coef_rep (nullable): the coefficients' striped partial sums -- nrep replicas rep_stride floats apart (norm.hip
// replica_offset) -- which are added to coef[k] here, so that no separate fold launch has to run between the backward kernels
// that produce them and this reduction.
__device__ __forceinline__ void wgrad_reduce_body(float (*tile)[65], int co, int cx, int zi, int zn,
                                    const float* __restrict__ slab, int nsplit, float* __restrict__ dw,
                                    int Cout, int Cin, int Cin_real, const float* coef, const float* cscale,
                                    const float* u, int ustride, const float* v, int vstride, int nrank,
                                    int accumulate, const float* coef_rep = nullptr, int nrep = 0, int rep_stride = 0) {
    const int ci0 = cx * 64;
    const int cw = min(64, Cin - ci0);                       // channels in this chunk (8 for the padded first layer)
    const size_t total = (size_t)Cout * 16 * Cin;
    // slab group zi of zn owns a group of splits (accumulate == 2: dw was zeroed by the caller, groups add atomically)
    const int per = (nsplit + zn - 1) / zn;
    const int k0 = zi * per, k1 = min(nsplit, k0 + per);
    // the rank-1 correction's operands (this thread's OUTPUT element: ci, 4 consecutive taps) are fetched first, so that their
    // latency lies under the slab loads instead of behind the barrier
    const int ocil = threadIdx.x >> 2, otp = (threadIdx.x & 3) * 4, oci = ci0 + ocil;
    float uk[4];
    float4 vk[4];
    float crep[4] = {0.f, 0.f, 0.f, 0.f};
    if (coef_rep && zi == 0) {                               // (uniform) lane r of every wave fetches replica r; xor-shuffles sum them
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pv = 0.f;
            if (k < nrank) for (int r = lane; r < nrep; r += 64) pv += coef_rep[(size_t)r * rep_stride + k];
            crep[k] = wave_sum(pv);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool on = k < nrank && zi == 0;
        uk[k] = on ? (coef[k] + crep[k]) * (cscale ? cscale[k] : 1.f) * u[(size_t)k * ustride + co] : 0.f;
        vk[k] = (on && ocil < cw && oci < Cin_real) ? *reinterpret_cast<const float4*>(v + (size_t)k * vstride + oci * 16 + otp)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {   // 16-byte loads: thread -> (tap, 4 consecutive ci); 256 threads cover the 16 x 64 tile once
        const int tap = threadIdx.x >> 4, cil = (threadIdx.x & 15) * 4;
        if (cil < cw) {
            const size_t idx = ((size_t)co * 16 + tap) * Cin + ci0 + cil;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int k = k0; k < k1; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(slab + (size_t)k * total + idx);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            tile[tap][cil] = s.x; tile[tap][cil + 1] = s.y; tile[tap][cil + 2] = s.z; tile[tap][cil + 3] = s.w;
        }
    }
    __syncthreads();
    // output: thread -> (ci, 4 consecutive taps): the 64 ci x 16 taps of one co are 4 KB contiguous in dw
    const int cil = ocil, tp = otp;
    const int ci = oci;
    if (cil < cw && ci < Cin_real) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sv = tile[tp + j][cil];
#pragma unroll
            for (int k = 0; k < 4; ++k) sv -= uk[k] * (j == 0 ? vk[k].x : j == 1 ? vk[k].y : j == 2 ? vk[k].z : vk[k].w);
            o[j] = sv;
        }
        float* op = dw + ((size_t)co * Cin_real + ci) * 16 + tp;
        if (accumulate == 2 && zn > 1) {                  // several slab groups add into the zeroed gradient
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(op + j, o[j]);
        } else if (accumulate == 1) {
            float4 t = *reinterpret_cast<float4*>(op);
            *reinterpret_cast<float4*>(op) = make_float4(t.x + o[0], t.y + o[1], t.z + o[2], t.w + o[3]);
        } else {
            *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, float* __restrict__ dw,
                                    int Cout, int Cin, int Cin_real, const float* coef, const float* cscale,
                                    const float* u, int ustride, const float* v, int vstride, int nrank,
                                    int accumulate) {
    __shared__ float tile[16][65];
    wgrad_reduce_body(tile, blockIdx.y, blockIdx.x, blockIdx.z, gridDim.z, slab, nsplit, dw, Cout, Cin, Cin_real, coef, cscale,
                      u, ustride, v, vstride, nrank, accumulate);
}

// up to 8 layers in one launch: the reductions of a whole backward pass are independent of each other and of the
// dgrad chain, and the small ones (first layers) ride along with the big ones instead of paying a launch each
struct RedLayer { const float* slab; float* dw; const float* coef; const float* u; const float* v;
                  int nsplit, Cout, Cin, Cin_real, nrank, zg, blk0;
                  const float* coef_rep; const float* bias_rep; float* dbias; };   // striped sums (replica 0) of coef / of the bias gradient
struct RedBatch { RedLayer l[8]; int nl, ustride, vstride, accumulate, nrep, rep_stride, bias_blk0; };

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(RedBatch b) {
    __shared__ float tile[16][65];
    if (b.bias_blk0 >= 0 && (int)blockIdx.x >= b.bias_blk0) {
        // tail workgroups: dbias[l][c] = sum over the replicas of the backward kernels' striped bias-gradient sums
        int j = ((int)blockIdx.x - b.bias_blk0) * 256 + threadIdx.x;
        for (int i = 0; i < b.nl; ++i) {
            const RedLayer& L = b.l[i];
            if (!L.bias_rep) continue;
            if (j < L.Cout) {
                float sv = 0.f;
#pragma unroll 8
                for (int r = 0; r < b.nrep; ++r) sv += L.bias_rep[(size_t)r * b.rep_stride + j];
                L.dbias[j] = sv;
                return;
            }
            j -= L.Cout;
        }
        return;
    }
    int li = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < b.nl && (int)blockIdx.x >= b.l[i].blk0) li = i;
    const RedLayer& L = b.l[li];
    const int local = blockIdx.x - L.blk0, nch = (L.Cin + 63) / 64;
    const int cx = local % nch, co = (local / nch) % L.Cout, zi = local / (nch * L.Cout);
    wgrad_reduce_body(tile, co, cx, zi, L.zg, L.slab, L.nsplit, L.dw, L.Cout, L.Cin, L.Cin_real, L.coef, nullptr, L.u, b.ustride,
                      L.v, b.vstride, L.nrank, b.accumulate, L.coef_rep, b.nrep, b.rep_stride);
}

// fp32 PyTorch-layout weight [Cout][Cin][4][4] -> packed operand layouts in T
template <typename T>
__global__ void prep_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wt,
                                   int Cout, int Cin, int CinP, float scale) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [Cout][16][CinP]
    const size_t total = (size_t)Cout * 16 * CinP;
    if (idx >= total) return;
    const int ci = idx % CinP, tap = (idx / CinP) % 16, co = idx / ((size_t)CinP * 16);
    const float val = ci < Cin ? w[((size_t)co * Cin + ci) * 16 + tap] * scale : 0.f;
    if (wf) Elem<T>::st(wf + idx, val);
    if (wt) Elem<T>::st(wt + ((size_t)ci * 16 + tap) * Cout + co, val);
}


// ------------------------------------------------------------------------------------------
// 3x3 stride-1 pad-1 convolutions (GeneratorSimpleRegressor.features, cgan/models.py:163-193): the same implicit-GEMM
// kernels with Geo<3>.  Packed operands, both with rows of `wk` = 9*C rounded up to 64 elements (zero padded):
//   W3f[Cout][tap][CinP]                      forward
//   W3t[Cin][8 - tap][Cout]                   data gradient = the same stride-1 conv of dy with the taps rotated by 180
//                                             degrees and the channel roles swapped (no separate dgrad kernel)
// ------------------------------------------------------------------------------------------
struct Prep3Layer { const float* w; void* wf; void* wt; int Cout, Cin, CinP, wkf, wkt; };
struct Prep3Batch { Prep3Layer l[8]; float scale; };
template <typename T>
__global__ __launch_bounds__(256) void prep3_weight_batch_kernel(Prep3Batch b) {
    const Prep3Layer& L = b.l[blockIdx.y];
    T* wf = static_cast<T*>(L.wf);
    T* wt = static_cast<T*>(L.wt);
    const size_t nf = wf ? (size_t)L.Cout * L.wkf : 0, nt = wt ? (size_t)L.Cin * L.wkt : 0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nf + nt; idx += (size_t)gridDim.x * 256) {
        if (idx < nf) {
            const int co = (int)(idx / L.wkf), k = (int)(idx % L.wkf);
            const int tap = k / L.CinP, ci = k % L.CinP;
            Elem<T>::st(wf + idx, (tap < 9 && ci < L.Cin) ? L.w[((size_t)co * L.Cin + ci) * 9 + tap] * b.scale : 0.f);
        } else {
            const size_t j = idx - nf;
            const int ci = (int)(j / L.wkt), k = (int)(j % L.wkt);
            const int tap = k / L.Cout, co = k % L.Cout;
            Elem<T>::st(wt + j, tap < 9 ? L.w[((size_t)co * L.Cin + ci) * 9 + (8 - tap)] * b.scale : 0.f);
        }
    }
}

// dw[co][ci][tap] = sum_s slab[s][co][tap][ci] over taps 0..8 of the 16-tap slab layout; one workgroup = one co x 64 ci:
// coalesced slab reads along ci, LDS transpose, one contiguous 576-float run of dw written.  Up to 8 layers per launch.
struct Red3Layer { const float* slab; float* dw; int nsplit, Cout, Cin, Cin_real, blk0; };
struct Red3Batch { Red3Layer l[8]; int nl; };
__global__ __launch_bounds__(256) void wgrad3_reduce_batch_kernel(Red3Batch b) {
    __shared__ float tile[9][65];
    int li = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < b.nl && (int)blockIdx.x >= b.l[i].blk0) li = i;
    const Red3Layer& L = b.l[li];
    const int local = blockIdx.x - L.blk0, nch = (L.Cin + 63) / 64;
    const int ci0 = (local % nch) * 64, co = local / nch;
    const int cw = min(64, L.Cin - ci0), cr = min(cw, L.Cin_real - ci0);
    const size_t total = (size_t)L.Cout * 16 * L.Cin;
    if (threadIdx.x < 9 * 16) {                                  // 16-byte loads: thread -> (tap, 4 consecutive ci)
        const int tap = threadIdx.x >> 4, cil = (threadIdx.x & 15) * 4;
        if (cil < cw) {
            const float* src = L.slab + ((size_t)co * 16 + tap) * L.Cin + ci0 + cil;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int k = 0; k < L.nsplit; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(src + (size_t)k * total);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            tile[tap][cil] = s.x; tile[tap][cil + 1] = s.y; tile[tap][cil + 2] = s.z; tile[tap][cil + 3] = s.w;
        }
    }
    __syncthreads();
    float* out = L.dw + ((size_t)co * L.Cin_real + ci0) * 9;
    for (int e = threadIdx.x; e < cr * 9; e += 256) out[e] = tile[e % 9][e / 9];
}

bool use_dma() {
    static bool v = [] { const char* e = getenv("GCSSL_CONV_DMA"); return !(e && e[0] == '0'); }();
    return v;
}
// 8-wave workgroups on the 128-row tiles put 2-3 waves on every SIMD (the kernels are instruction-issue bound at
// 1-1.5 waves/SIMD: ~115 non-MFMA instructions per 8 MFMAs); GCSSL_DMA_WAVES=4 forces the 4-wave form for A/B runs.
int dma_waves() {
    static int v = [] { const char* e = getenv("GCSSL_DMA_WAVES"); return e ? atoi(e) : 8; }();
    return v;
}
template <typename T, int BM, int BN, int MODE> struct Dma8 {            // 8-wave form exists only for the 128-row tiles
    static bool launch(const ConvParams&, dim3, hipStream_t) { return false; }
};
template <typename T, int MODE> struct Dma8<T, 128, 128, MODE> {
    static bool launch(const ConvParams& p, dim3 grid, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_kernel<T, 128, 128, MODE, 2, 4, false>), grid, dim3(512), 0, st, p);
        return true;
    }
};
template <typename T, int MODE> struct Dma8<T, 128, 64, MODE> {
    static bool launch(const ConvParams& p, dim3 grid, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_kernel<T, 128, 64, MODE, 4, 2, false>), grid, dim3(512), 0, st, p);
        return true;
    }
};
// persistent form: only where there are clearly more tiles than resident workgroups (otherwise it IS the plain kernel)
int persist_mode() {
    static int v = [] { const char* e = getenv("GCSSL_PERSIST"); return e ? atoi(e) : 3; }();   // bit 0: Cin >= 64 layers, bit 1: 8-channel layers
    return v;
}
int cu_count() {
    static int v = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    return v;
}
template <typename T, int BM, int BN, int MODE> struct Persist {
    static bool launch(const ConvParams&, int, int, int, int, hipStream_t) { return false; }
};
template <typename T, int MODE> struct Persist<T, 128, 64, MODE> {
    static bool launch(const ConvParams& p, int g, int tm, int tn, int total, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_persist_kernel<T, 128, 64, MODE, 4, 2>), dim3(g), dim3(512), 0, st, p, tm, tn, total);
        return true;
    }
};
template <typename T, int MODE> struct Persist<T, 64, 64, MODE> {
    static bool launch(const ConvParams& p, int g, int tm, int tn, int total, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_persist_kernel<T, 64, 64, MODE, 2, 2>), dim3(g), dim3(256), 0, st, p, tm, tn, total);
        return true;
    }
};
template <typename T, int BM, int BN, int MODE> struct PersistSmallK {     // 8-channel first layers (K = 128: two K steps per tile)
    static bool launch(const ConvParams&, int, int, int, int, hipStream_t) { return false; }
};
template <typename T> struct PersistSmallK<T, 128, 64, 0> {
    static bool launch(const ConvParams& p, int g, int tm, int tn, int total, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_persist_kernel<T, 128, 64, 0, 2, 2, true>), dim3(g), dim3(256), 0, st, p, tm, tn, total);
        return true;
    }
};
template <typename T, int BM, int BN, int MODE>
void launch_dma(const ConvParams& p, dim3 grid, bool smallk, hipStream_t st) {
    if (persist_mode() && p.ksplit <= 1 && p.y_bytes && dma_waves() == 8) {
        const int nk = (MODE == 0 ? 16 * p.Cin : 4 * p.Cout) / 64;
        const int resident = (160 * 1024) / (3 * (BM + BN) * 128);
        const int slots = cu_count() * resident, total = (int)(grid.x * grid.y * grid.z);
        if (nk >= 2 && total > slots + slots / 4) {
            if (!smallk && Persist<T, BM, BN, MODE>::launch(p, slots, (int)grid.x, (int)grid.y, total, st)) return;
            if (smallk && (persist_mode() & 2) && PersistSmallK<T, BM, BN, MODE>::launch(p, slots, (int)grid.x, (int)grid.y, total, st)) return;
        }
    }
    if (dma_waves() == 8 && !smallk && Dma8<T, BM, BN, MODE>::launch(p, grid, st)) return;
    if (smallk) {
        GCSSL_LAUNCH((conv_dma_kernel<T, BM, BN, MODE, 2, 2, true>), grid, dim3(256), 0, st, p);
    } else {
        GCSSL_LAUNCH((conv_dma_kernel<T, BM, BN, MODE, 2, 2, false>), grid, dim3(256), 0, st, p);
    }
}

// several layers per launch (blockIdx.y = layer): the critic is re-packed after every optimiser step
struct PrepLayer { const float* w; void* wf; void* wt; int Cout, Cin, CinP; };
struct PrepBatch { PrepLayer l[8]; int nl; const float* w5; float* w5p; int C5; float scale; };   // w5: the critic head's [1][C5][4][4] -> fp32 [16][C5] (nullable)
// 16 consecutive fp32 values of an LDS row -> 16 consecutive T in global memory (32 or 64 bytes, vector stores)
template <typename T> __device__ __forceinline__ void store16(T* dst, const float* src);
template <> __device__ __forceinline__ void store16<float>(float* dst, const float* src) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        reinterpret_cast<float4*>(dst)[j] = make_float4(src[4 * j], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
}
template <typename T> __device__ __forceinline__ void store16_16(T* dst, const float* src) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint4 w;
        w.x = pack2<T>(src[8 * j + 0], src[8 * j + 1]);
        w.y = pack2<T>(src[8 * j + 2], src[8 * j + 3]);
        w.z = pack2<T>(src[8 * j + 4], src[8 * j + 5]);
        w.w = pack2<T>(src[8 * j + 6], src[8 * j + 7]);
        reinterpret_cast<uint4*>(dst)[j] = w;
    }
}
template <> __device__ __forceinline__ void store16<bf16_t>(bf16_t* dst, const float* src) { store16_16<bf16_t>(dst, src); }
template <> __device__ __forceinline__ void store16<f16_t>(f16_t* dst, const float* src) { store16_16<f16_t>(dst, src); }

// The re-pack is a pair of transposes of w[co][ci][tap]: Wf[co][tap][ci] makes ci the fast axis, Wt[ci][tap][co] makes
// co the fast axis.  Each goes through an LDS tile shaped so that BOTH the fp32 reads and the packed writes are full
// 128-byte runs (blockIdx.z = 0: 4 co x 64 ci x 16 taps for Wf; 1: 64 co x 4 ci x 16 taps for Wt); the element-wise
// form below (2-byte writes 16*Cout elements apart for Wt) ran at <1 TB/s.  Layers whose channel counts are not
// multiples of 64, or padded (the first layers), keep the element-wise form: they are tiny.
template <typename T>
__global__ __launch_bounds__(256) void prep_weight_batch_kernel(PrepBatch b, SnBatch sn, int sn_row, C5DgradRider c5r, int c5_row) {
    if ((int)blockIdx.y == sn_row) {                         // one more grid row: a spectral-norm chain's closing step (gcssl_sn_defer_finish)
        if (blockIdx.z == 0 && (int)blockIdx.x < sn.nl) sn_finish_body(sn, blockIdx.x, sn.nl);
        return;
    }
    if ((int)blockIdx.y == c5_row) {                         // ... and one for the head conv's constant-seed data gradient (gcssl_conv4x4s1_c1_dgrad_defer)
        if (blockIdx.z == 0) for (int blk = blockIdx.x; blk < c5r.nblk; blk += gridDim.x) c5_dgrad_rider_body(c5r, blk);
        return;
    }
    if ((int)blockIdx.y == b.nl) {                           // the extra grid row: the 512 -> 1 head conv's weight (gcssl_prep_c5_weight)
        if (blockIdx.z) return;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 16 * b.C5; idx += gridDim.x * 256)
            b.w5p[idx] = b.w5[(size_t)(idx % b.C5) * 16 + idx / b.C5];
        return;
    }
    const PrepLayer L = b.l[blockIdx.y];
    __shared__ float tile[64 * 65];
    const bool tiled = L.Cin % 64 == 0 && L.Cout % 64 == 0 && L.CinP == L.Cin;
    T* wf = static_cast<T*>(L.wf);
    T* wt = static_cast<T*>(L.wt);
    const int tid = threadIdx.x;
    if (!tiled) {
        if (blockIdx.z) return;
        const size_t total = (size_t)L.Cout * 16 * L.CinP;
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + tid; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
            const int ci = idx % L.CinP, tap = (idx / L.CinP) % 16, co = idx / ((size_t)L.CinP * 16);
            const float val = ci < L.Cin ? L.w[((size_t)co * L.Cin + ci) * 16 + tap] * b.scale : 0.f;
            if (wf) Elem<T>::st(wf + idx, val);
            if (wt) Elem<T>::st(wt + ((size_t)ci * 16 + tap) * L.Cout + co, val);
        }
        return;
    }
    if (blockIdx.z == 0) {
        if (!wf) return;
        const int nci = L.Cin / 64, ntiles = (L.Cout / 4) * nci;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int co0 = (t / nci) * 4, ci0 = (t % nci) * 64;
            const int ci = tid >> 2, tp = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {               // one co: 64 ci x 16 taps = 4 KB contiguous
                const float4 v = reinterpret_cast<const float4*>(L.w + ((size_t)(co0 + i) * L.Cin + ci0) * 16)[tid];
                float* row = tile + (i * 16 + tp) * 65 + ci;       // tile[(co, tap)][ci]
                row[0] = v.x * b.scale; row[65] = v.y * b.scale; row[130] = v.z * b.scale; row[195] = v.w * b.scale;
            }
            __syncthreads();
            const int r = tid >> 2, c0 = (tid & 3) * 16;            // row r = (co, tap): 64 ci = 128 B (bf16)
            store16<T>(wf + ((size_t)(co0 + (r >> 4)) * 16 + (r & 15)) * L.CinP + ci0 + c0, tile + r * 65 + c0);
            __syncthreads();
        }
    } else {
        if (!wt) return;
        const int nci = L.Cin / 4, ntiles = (L.Cout / 64) * nci;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int co0 = (t / nci) * 64, ci0 = (t % nci) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {               // per co: 4 ci x 16 taps = 256 B contiguous
                const int f = tid + 256 * i, co = f >> 4, piece = f & 15;
                const float4 v = reinterpret_cast<const float4*>(L.w + ((size_t)(co0 + co) * L.Cin + ci0) * 16)[piece];
                float* col = tile + (piece * 4) * 65 + co;          // tile[k = ci_local*16 + tap][co]
                col[0] = v.x * b.scale; col[65] = v.y * b.scale; col[130] = v.z * b.scale; col[195] = v.w * b.scale;
            }
            __syncthreads();
            const int r = tid >> 2, c0 = (tid & 3) * 16;            // row r = (ci_local, tap): 64 co = 128 B (bf16)
            store16<T>(wt + ((size_t)ci0 * 16 + r) * L.Cout + co0 + c0, tile + r * 65 + c0);
            __syncthreads();
        }
    }
}

int x3_waves() {       // waves per workgroup of the split-precision 128 x 64 tiles: 8 = forward 4 x 2 (default), 16 = the data gradient too, 4 = 2 x 2 (A/B)
    static const int v = [] { const char* e = getenv("GCSSL_X3_WAVES"); return e ? atoi(e) : 8; }();
    return v;
}
template <typename T, int BM, int BN, int MM = 0>
int launch_fwd(const ConvParams& p, hipStream_t st) {
    if (p.plan_out) { *p.plan_out = p.ksplit > 1 ? p.ksplit : 1; return GCSSL_OK; }
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.ksplit > 1 ? p.ksplit : 1);
    if constexpr (MM != 0 && BM == 128 && BN == 64) {
        if (x3_waves() >= 8) { GCSSL_LAUNCH((conv_fwd_kernel<T, BM, BN, 4, MM, 4, 2>), grid, dim3(512), 0, st, p); return gcssl_launch_status(); }
    }
    if (Is16<T>::v && use_dma()) launch_dma<typename Op16<T>::type, BM, BN, 0>(p, grid, p.Cin < 64, st);
    else GCSSL_LAUNCH((conv_fwd_kernel<T, BM, BN, 4, MM>), grid, dim3(NT), 0, st, p);
    return gcssl_launch_status();
}
template <typename T, int BM, int BN, int MM = 0>
int launch_dgrad(const ConvParams& p, hipStream_t st) {
    if (p.plan_out) { *p.plan_out = p.ksplit > 1 ? p.ksplit : 1; return GCSSL_OK; }
    dim3 grid((p.M + BM - 1) / BM, (p.Cin + BN - 1) / BN, 4 * (p.ksplit > 1 ? p.ksplit : 1));
    if constexpr (MM != 0 && BM == 128 && BN == 64) {
        if (x3_waves() == 16 && !p.ab_a) { GCSSL_LAUNCH((conv_dgrad_kernel<T, BM, BN, MM, 4, 2>), grid, dim3(512), 0, st, p); return gcssl_launch_status(); }   // (A/B only: neutral)
    }
    if constexpr (MM != 0) {
        if (p.ab_a) { GCSSL_LAUNCH((conv_dgrad_kernel<T, BM, BN, MM, 2, 2, true>), grid, dim3(NT), 0, st, p); return gcssl_launch_status(); }
    }
    if (Is16<T>::v && use_dma()) launch_dma<typename Op16<T>::type, BM, BN, 1>(p, grid, false, st);
    else GCSSL_LAUNCH((conv_dgrad_kernel<T, BM, BN, MM>), grid, dim3(NT), 0, st, p);
    return gcssl_launch_status();
}
// split-precision forms: the PACKED weights (B operand of the forward / data-gradient GEMMs) carry a factor 2^6 -- N(0, 0.02)
// entries land at O(1), where the fp16 lo half keeps all its bits (below 6e-5 it would be subnormal) -- put there by the pack
// kernels when they are called with a split dtype; the conv epilogues scale the accumulators back.  Exact (powers of two).
// GCSSL_X3_WSCALE: log2 of the factor (experiments).
float x3_wscale() {
    static const int lg = [] { const char* e = getenv("GCSSL_X3_WSCALE"); return e ? atoi(e) : 6; }();
    return (float)(1 << lg);
}
void set_mm_scales(ConvParams& p, bool weights_b) {
    p.mm_oscale = weights_b ? 1.f / x3_wscale() : 1.f;
    static const int dbg = [] { const char* e = getenv("GCSSL_X3_DEBUG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
}

int check_geom(int N, int Hi, int Wi, int Cin, int Cout) {
    if (N <= 0 || Hi < 2 || Wi < 2 || !is_pow2(Hi) || !is_pow2(Wi) || !is_pow2(Cin) || !is_pow2(Cout))
        return GCSSL_EBADSHAPE;
    if (Cin < 8 || Cout < 8) return GCSSL_EBADSHAPE;
    return GCSSL_OK;
}

// operand extents in bytes for the buffer-load bounds check; false if a tensor does not fit 31-bit byte offsets
bool fill_bytes(ConvParams& p, size_t x_elems, size_t w_elems, int es) {
    const size_t xb = x_elems * es, wb = w_elems * es;
    if (xb >= 0x7FFFFFFFull || wb >= 0x7FFFFFFFull) return false;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    return true;
}

void fill_geom(ConvParams& p, int N, int Hi, int Wi, int Cin, int Cout) {
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout;
    p.lgWo = ilog2(Wi / 2); p.lgHoWo = ilog2((Hi / 2) * (Wi / 2));
    p.lgCin = ilog2(Cin); p.lgCout = ilog2(Cout);
    p.M = N * (Hi / 2) * (Wi / 2);
    static const int epi = [] { const char* e = getenv("GCSSL_EPI_LDS"); return e ? atoi(e) : 1; }();
    p.epi_lds = epi;
}

// pick the workgroup tile so that the grid fills the 256 CUs when the problem allows it
// Small-M layers (G.down4, the GP chain at batch B, ...) would launch far fewer than 2 workgroups per CU with a
// 64..128-deep serial K loop each: they are latency-bound, not MFMA-bound.  When the output is fp32 and the epilogue
// linear, split K across workgroups and accumulate with fp32 atomics into a zeroed output (128-B contiguous per
// half-wave: the full-rate atomic shape of MI355X_MICROARCH.md "Global float atomics").
// a bigger tile is only chosen when it still yields this many workgroups (occupancy hides the gather latency)
long tile_threshold() {
    static long v = [] { const char* e = getenv("GCSSL_TILE_WGS"); return e ? atol(e) : 256L; }();
    return v;
}

// 128x128 (one workgroup per CU: 96 KB of LDS) lost to 128x64 (two per CU) on every layer of the forced-tile matrix
// (tools/archive/conv_matrix.sh: D.c2.fwd 37.6 -> 29.7 us, D.c3.dgrad 35.2 -> 27.6 us), so it needs many more tiles to be chosen
long tile128_threshold() {
    static long v = [] { const char* e = getenv("GCSSL_TILE128_WGS"); return e ? atol(e) : 2048L; }();
    return v;
}

int split_tile() {
    static int v = [] { const char* e = getenv("GCSSL_SPLIT_TILE"); return e ? atoi(e) : 64; }();
    return v;
}
int ksplit_max() {
    static int v = [] { const char* e = getenv("GCSSL_KSPLIT_MAX"); return e ? atoi(e) : 8; }();
    return v;
}
int sib_remap() {
    static int v = [] { const char* e = getenv("GCSSL_SIB_REMAP"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
int wgrad_xcd() {
    static int v = [] { const char* e = getenv("GCSSL_WGRAD_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
int pick_ksplit(long tiles, int nk, bool allowed) {
    static const long nosplit = [] { const char* e = getenv("GCSSL_SPLIT_TILES"); return e ? atol(e) : 384L; }();
    static const long target = [] { const char* e = getenv("GCSSL_SPLIT_TARGET"); return e ? atol(e) : 512L; }();
    if (!allowed || tiles >= nosplit || nk < 16 || ksplit_max() <= 1) return 1;
    int ks = (int)((target + tiles - 1) / tiles);
    if (ks > ksplit_max()) ks = ksplit_max();
    while (ks > 1 && nk / ks < 8) --ks;
    return ks;
}

// The loader / consumer ring (launch_ring: 128 x 128 tiles, one workgroup per CU) fills LDS at the same ~43 GB/s per CU as
// every other form, but does 65 FLOP per filled byte instead of 44: 1.5x per CU, measured.  It only pays when its workgroups
// fill ONE round of the chip -- 160..256 of them, if need be by a K split of >= 8 steps each (D.c3.fwd[n=768] 26.6 -> 23.6 us,
// D.c4.dgrad 27.7 -> 22.9 us with 192 tiles; D.c2.fwd 24.6 -> 33.6 us and D.c3.dgrad 26.2 -> 30.4 us with 384 tiles: a
// round and a half).  Returns the K split to use, 0 = not a ring shape.
int ring_ksplit(long tiles, int nk, bool split_allowed) {
    static const int on = [] { const char* e = getenv("GCSSL_RING"); return e ? atoi(e) : 1; }();
    if (!on || !use_dma() || tiles <= 0 || tiles > cu_count()) return 0;
    if (tiles >= 160) return 1;
    // (D.c4.fwd[n=768] unsplit on 96 CUs is 28.5 vs 30.6 us stand-alone, but 39 vs 32 us inside the iteration: kept in two halves)
    if (!split_allowed || ksplit_max() <= 1) return 0;
    // ... a split only in two halves of >= 32 K steps (D.c4.fwd[n=768], 96 tiles: 40.0 -> 32.5 us); deeper splits of the
    // B-sample layers lost to the 64 x 64 split-K form (G.up2.dgrad 23.4 -> 34.1 us, D.c3.gp_dgrad 20.3 -> 24.2 us)
    static const int deep = [] { const char* e = getenv("GCSSL_RING_SPLIT"); return e ? atoi(e) : 2; }();
    int ks = (int)(cu_count() / tiles);
    if (ks > ksplit_max()) ks = ksplit_max();
    if (ks > deep) return 0;
    return (ks >= 2 && tiles * ks >= 160 && nk / ks >= 32) ? ks : 0;
}

int zero_output(const ConvParams& p, long rows, int cols, hipStream_t st) {
    if (p.split_stride || p.plan_out) return GCSSL_OK;                     // slab mode: every split owns its own slab, nothing to zero
    gcssl_zero2d_async(static_cast<float*>(p.y), (size_t)p.ldy, cols, (size_t)rows, st);   // (a kernel, not a memset node: common.h)
    return gcssl_launch_status();
}

// 256-row tiles, bf16 LDS-DMA path only: 8 waves as 4 (M) x 2 (N), so a wave owns 64x64 (or 64x32) outputs and issues
// 16 (8) MFMAs per K step against a fixed per-step cost (barrier skew, DMA issue, fragment reads) of ~750 cycles --
// measured with s_memtime (tools/archive/trace_conv.py): with 32x32 wave tiles that fixed cost is 5x the MFMA time.
// 144 KB (120 KB) of LDS: one workgroup per CU, two waves per SIMD.
const char* forced_tile() {
    static const char* v = getenv("GCSSL_FORCE_TILE");       // "256x128", "256x64", "128x128", "128x64", "64x64": experiments
    return v;
}
template <typename T, int BM, int BN, int MODE> struct PersistBig {       // 256x128 has 64 epilogue stores per wave: beyond the 6-bit vmcnt
    static bool launch(const ConvParams&, int, int, int, int, hipStream_t) { return false; }
};
template <typename T, int MODE> struct PersistBig<T, 256, 64, MODE> {
    static bool launch(const ConvParams& p, int g, int tm, int tn, int total, hipStream_t st) {
        GCSSL_LAUNCH((conv_dma_persist_kernel<T, 256, 64, MODE, 4, 2>), dim3(g), dim3(512), 0, st, p, tm, tn, total);
        return true;
    }
};
// 128 x 128 tiles on the loader / consumer form of conv_dma_kernel: 8 consumer waves (32 x 64 wave tiles) + 4 loader waves,
// 4-slot ring (128 KB of LDS, one workgroup per CU), three K tiles in flight
template <typename T, int MODE>
int launch_ring(const ConvParams& p, hipStream_t st) {
    if (p.plan_out) { *p.plan_out = p.ksplit > 1 ? p.ksplit : 1; return GCSSL_OK; }
    const int ncols = MODE == 0 ? p.Cout : p.Cin;
    dim3 grid((p.M + 127) / 128, (ncols + 127) / 128, (MODE == 1 ? 4 : 1) * (p.ksplit > 1 ? p.ksplit : 1));
    static const int dbg = [] { const char* e = getenv("GCSSL_RING_DEBUG"); return e ? atoi(e) : 0; }();
    ConvParams q = p; q.dbg = dbg;
    // 8 loader waves (1024 threads, <= 128 registers: the consumers read -> wait -> MFMA inside one barrier interval) against 4
    // loader waves with consumers software-pipelined across the barrier: 20.9 vs 23.8 us (D.c3.fwd), 20.8 vs 23.9 (D.c4.dgrad).
    // The K loop of the 4-loader form ran at the loaders' issue rate (8 pieces x ~105 cycles per wave and step).
    static const int lw8 = [] { const char* e = getenv("GCSSL_RING_LW8"); return e ? atoi(e) : 1; }();
    if (lw8) { GCSSL_LAUNCH((conv_dma_kernel<T, 128, 128, MODE, 4, 2, false, 8, 4, false>), grid, dim3(1024), 0, st, q); return gcssl_launch_status(); }
    GCSSL_LAUNCH((conv_dma_kernel<T, 128, 128, MODE, 4, 2, false, 4, 4>), grid, dim3(768), 0, st, q);
    return gcssl_launch_status();
}
template <typename T, int BM, int BN, int MODE>
int launch_big(const ConvParams& p, hipStream_t st) {
    if (p.plan_out) { *p.plan_out = p.ksplit > 1 ? p.ksplit : 1; return GCSSL_OK; }
    const int ncols = MODE == 0 ? p.Cout : p.Cin;
    dim3 grid((p.M + BM - 1) / BM, (ncols + BN - 1) / BN, (MODE == 1 ? 4 : 1) * (p.ksplit > 1 ? p.ksplit : 1));
    const int total = (int)(grid.x * grid.y * grid.z), slots = cu_count();          // 120-144 KB of LDS: one per CU
    if (persist_mode() && p.ksplit <= 1 && p.y_bytes && total > slots + slots / 4 &&
        PersistBig<T, BM, BN, MODE>::launch(p, slots, (int)grid.x, (int)grid.y, total, st)) return gcssl_launch_status();
    GCSSL_LAUNCH((conv_dma_kernel<T, BM, BN, MODE, 4, 2, false>), grid, dim3(512), 0, st, p);
    return gcssl_launch_status();
}

// ------------------------------------------------------------------------------------------
// forward of the FIRST layers (Cin padded to 8, Cout = 64: D.c1, G.down1 and the reverse GP chain's c1): K = 16 taps x 8
// channels = 128, two K steps of the generic 64-deep ring -- a launch that is all prologue and epilogue (28-36 us for 38 MB
// at batch 768).  These layers are HBM-bound, so the kernel is built around its memory accesses:
//   * a workgroup owns 128 consecutive output pixels of one sample (R = 128 / Wo output rows) and loads the 2R + 2 input
//     rows they touch ONCE, with coalesced 16-byte loads (a pixel's 8 channels are 16 bytes), into an LDS image with a zero
//     pixel at both row ends and zero rows outside the map: no bounds logic after this point;
//   * an MFMA A fragment (lane = output pixel, one tap's 8 channels) is one ds_read_b128 of that image -- the im2col
//     happens in the LDS address;
//   * the 16 B fragments of a wave (64 output channels x K = 128) come straight from the 16-KB packed weight and stay in
//     registers;
//   * the 128 x 64 result goes through an LDS transpose so that the stores are full 128-byte (16-bit) / 256-byte (fp32)
//     runs per pixel.
// Geometry: Wo in {16, 32, 64} (32x32 ... 128x128 inputs), one sample's rows per tile; other shapes take the generic path.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_fwd_c8_kernel(ConvParams p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    constexpr int MT = 128, WROW = 17;                                   // weight rows of 16 chunks + 1 pad chunk (272 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char c8_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wo = p.Wi >> 1, Ho = p.Hi >> 1, R = MT >> p.lgWo, rows = 2 * R + 2, rowpx = p.Wi + 2;
    const int tiles_per_n = Ho / R;
    uint4* img = reinterpret_cast<uint4*>(c8_lds);                       // [rows][Wi + 2] pixels of 16 bytes
    unsigned char* outt = c8_lds + (size_t)rows * rowpx * 16;            // [128][64] result tile; first: the weight image
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    // input image: chunk c -> (row, px); rows outside the map read OOB = 0; the two pad pixels of a row are zeroed.  A tile's
    // chunks (<= 3 per thread) are fetched into registers one tile AHEAD: the loads of tile t+1 fly under the MFMAs, the LDS
    // transpose and the stores of tile t (fetch -> LDS -> MFMA -> LDS -> store in sequence left every phase exposed)
    constexpr int NC = 3;
    const int nchunk = rows * p.Wi;                                      // 576 / 640 / 768 for Wo = 16 / 32 / 64
    u32x4 pre[NC];
    auto fetch = [&](int tile) {
        const int n = tile / tiles_per_n, iy0 = 2 * (tile - n * tiles_per_n) * R - 1;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> (p.lgWo + 1), px = c & (p.Wi - 1), iy = iy0 + row;
            const bool ok = c < nchunk && (unsigned)iy < (unsigned)p.Hi;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((n * p.Hi + iy) * p.Wi + px) * p.ldx) * 2) : OOB, 0, 0);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);                     // (in flight under the weight prologue below)
    // The packed weight Wf[64 co][16 taps][8 ci] (16 KB) is fetched ONCE per workgroup with coalesced 16-byte loads and
    // passed through LDS (rows padded to 17 chunks: the fragment reads of 32 consecutive co are conflict-free).  Loading
    // the fragments straight from memory touches 32 different 128-byte lines per instruction, and L1 serves lines, not
    // bytes: that version ran at the generic kernel's 30 us.
    {
        uint4* wimg = reinterpret_cast<uint4*>(outt);
        for (int c = tid; c < 64 * 16; c += 256)
            wimg[(c >> 4) * WROW + (c & 15)] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 16u, 0, 0));
        __syncthreads();
    }
    FragT bf[2][8];                                                      // lane -> co = 32 j + (lane & 31), tap = 2 ks + (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bf[j][ks] = __builtin_bit_cast(FragT, reinterpret_cast<const uint4*>(outt)[(32 * j + (lane & 31)) * WROW + 2 * ks + (lane >> 5)]);
    const int pl = 32 * wave + (lane & 31), oyl = pl >> p.lgWo, ox = pl & (Wo - 1), h = lane >> 5;
    const bool f32out = p.out_f32;
    const bool actb = p.ab_a != nullptr;                                 // (the tile then holds fp32 values: host sizes it so)
    float dot_acc = 0.f; int nsat = 0;
    const int cpp = f32out ? 16 : 8, es = f32out ? 4 : 2;                // 16-byte chunks per output pixel (64 channels)
    unsigned char* yb = static_cast<unsigned char*>(p.y);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
        // (The barrier that follows also closes the previous tile's reads of `outt` / the weight image.)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            if (c < nchunk) img[(c >> (p.lgWo + 1)) * rowpx + (c & (p.Wi - 1)) + 1] = __builtin_bit_cast(uint4, pre[i]);
        }
        if (tid < 2 * rows) img[(tid >> 1) * rowpx + ((tid & 1) ? p.Wi + 1 : 0)] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        // A fragments: lane -> output pixel pl of the tile, tap = 2 ks + h: one ds_read_b128 of the image each
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        FragT af[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int tap = 2 * ks + h, ky = tap >> 2, kx = tap & 3;
            af[ks] = __builtin_bit_cast(FragT, img[(2 * oyl + ky) * rowpx + 2 * ox + kx]);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mfma(af[ks], bf[j][ks], acc[j]);
        // epilogue: scale, bias, activation; transpose through LDS; 16-byte stores
        const float gs = p.gscale ? p.gscale[n / p.group_n] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = 32 * j + (lane & 31);
            const float b = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = 32 * wave + crow(r, lane);
                float v = acc[j][r] * gs + b;
                if (p.act == 1) v = lrelu_f(v);
                if (f32out || actb) reinterpret_cast<float*>(outt)[prow * 64 + co] = v;
                else reinterpret_cast<unsigned short*>(outt)[prow * 64 + co] = (unsigned short)Bits16<T>::enc(v);
            }
        }
        __syncthreads();
        const size_t m0 = (size_t)(n * Ho + oy0) * Wo;                   // first output pixel of the tile
        if (actb) {
            // activation backward of the layer whose (reverse-chain) forward this is: y = lrelu'(a) * v in the compute dtype,
            // a = the stored activation at the same pixels; and the spectral-norm dot <dotx, v> of the same pass
            const unsigned char* ab = static_cast<const unsigned char*>(p.ab_a);
            const unsigned char* xb = static_cast<const unsigned char*>(p.ab_dotx);
            for (int c = tid; c < MT * 8; c += 256) {
                const int px = c >> 3, ch = c & 7;
                const float4 v0 = reinterpret_cast<const float4*>(outt)[px * 16 + ch * 2], v1 = reinterpret_cast<const float4*>(outt)[px * 16 + ch * 2 + 1];
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const uint4 aw = *reinterpret_cast<const uint4*>(ab + ((m0 + px) * p.ab_lda) * 2 + ch * 16);
                const unsigned awv[4] = {aw.x, aw.y, aw.z, aw.w};
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = Bits16<T>::dec(awv[e >> 1] >> (16 * (e & 1))) > 0.f ? v[e] : 0.2f * v[e];
                if (xb) {
                    const uint4 xw = *reinterpret_cast<const uint4*>(xb + ((m0 + px) * p.ab_lddot) * 2 + ch * 16);
                    const unsigned xwv[4] = {xw.x, xw.y, xw.z, xw.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot_acc += Bits16<T>::dec(xwv[e >> 1] >> (16 * (e & 1))) * v[e];
                }
                nsat += sat_hits<T>(o);
                uint4 w; w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
                *reinterpret_cast<uint4*>(yb + ((m0 + px) * p.ldy) * 2 + ch * 16) = w;
            }
            continue;
        }
        for (int c = tid; c < MT * cpp; c += 256) {
            const int px = f32out ? c >> 4 : c >> 3, ch = c & (cpp - 1);
            *reinterpret_cast<uint4*>(yb + ((m0 + px) * p.ldy) * es + ch * 16) = reinterpret_cast<const uint4*>(outt)[c];
        }
    }
    if (actb) {
        sat_commit(p.ab_sat, nsat);
        if (p.ab_dot_out) {
            __shared__ float dred[4];
            const float tot = block_sum<4>(dot_acc, dred);
            if (tid == 0) atomicAdd(p.ab_dot_out, tot);
        }
    }
#endif
}
// The same kernel for the split-precision modes (fp32 tensors, MFMA operands split into 16-bit halves: igemm.hip header): a
// pixel's 8 fp32 channels are 32 bytes (two 16-byte loads) and become one hi and one lo fragment chunk when the image is
// written to LDS -- two images; the fp32 packed weight (2^6-scaled, x3_wscale) is split once per workgroup into hi / lo
// fragment sets that stay in registers; three MFMAs per (tap pair, 32 output channels): A_lo B_hi + A_hi B_lo + A_hi B_hi.
// The generic split kernel spent 34.7 us on D.c1.fwd[n=768] (K = 128: four steps, all prologue / epilogue) against the
// 16-bit kernel's 16.
template <typename H>
__global__ __launch_bounds__(256) void conv_fwd_c8_x3_kernel(ConvParams p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<H>::type FragT;
    constexpr int MT = 128, WROW = 17;
    extern __shared__ __attribute__((aligned(16))) unsigned char c8_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wo = p.Wi >> 1, Ho = p.Hi >> 1, R = MT >> p.lgWo, rows = 2 * R + 2, rowpx = p.Wi + 2;
    const int tiles_per_n = Ho / R;
    const int npix = rows * rowpx;
    uint4* img_hi = reinterpret_cast<uint4*>(c8_lds);                    // [rows][Wi + 2] pixels of 8 hi halves
    uint4* img_lo = img_hi + npix;
    unsigned char* outt = c8_lds + (size_t)npix * 32;                    // [128][64] fp32 result tile; first: the two weight images
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    constexpr int NC = 3;
    const int nchunk = rows * p.Wi;                                      // pixels of the image (<= 3 per thread)
    u32x4 pre[NC][2];
    auto fetch = [&](int tile) {
        const int n = tile / tiles_per_n, iy0 = 2 * (tile - n * tiles_per_n) * R - 1;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> (p.lgWo + 1), px = c & (p.Wi - 1), iy = iy0 + row;
            const bool ok = c < nchunk && (unsigned)iy < (unsigned)p.Hi;
            const unsigned off = ok ? (unsigned)((((n * p.Hi + iy) * p.Wi + px) * p.ldx) * 4) : OOB;
            pre[i][0] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            pre[i][1] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? off + 16u : OOB, 0, 0);
        }
    };
    auto split8 = [](const u32x4& a, const u32x4& b, uint4& hi, uint4& lo) {
        uint2 h0, l0, h1, l1;
        split4<H>(__builtin_bit_cast(float4, a), h0, l0);
        split4<H>(__builtin_bit_cast(float4, b), h1, l1);
        hi = make_uint4(h0.x, h0.y, h1.x, h1.y); lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    {
        uint4* whi = reinterpret_cast<uint4*>(outt);
        uint4* wlo = whi + 64 * WROW;
        for (int c = tid; c < 64 * 16; c += 256) {                       // c -> (co, 8 consecutive K elements = tap pair half)
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 32u, 0, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 32u + 16u, 0, 0);
            uint4 hi, lo;
            split8(a, b, hi, lo);
            whi[(c >> 4) * WROW + (c & 15)] = hi; wlo[(c >> 4) * WROW + (c & 15)] = lo;
        }
        __syncthreads();
    }
    FragT bh[2][8], bl[2][8];                                            // lane -> co = 32 j + (lane & 31), tap = 2 ks + (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int e = (32 * j + (lane & 31)) * WROW + 2 * ks + (lane >> 5);
            bh[j][ks] = __builtin_bit_cast(FragT, reinterpret_cast<const uint4*>(outt)[e]);
            bl[j][ks] = __builtin_bit_cast(FragT, reinterpret_cast<const uint4*>(outt)[64 * WROW + e]);
        }
    const int pl = 32 * wave + (lane & 31), oyl = pl >> p.lgWo, ox = pl & (Wo - 1), h = lane >> 5;
    unsigned char* yb = static_cast<unsigned char*>(p.y);
    const bool actb = p.ab_a != nullptr;
    float dot_acc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            if (c < nchunk) {
                uint4 hi, lo;
                split8(pre[i][0], pre[i][1], hi, lo);
                const int e = (c >> (p.lgWo + 1)) * rowpx + (c & (p.Wi - 1)) + 1;
                img_hi[e] = hi; img_lo[e] = lo;
            }
        }
        if (tid < 2 * rows) {
            const int e = (tid >> 1) * rowpx + ((tid & 1) ? p.Wi + 1 : 0);
            img_hi[e] = make_uint4(0u, 0u, 0u, 0u); img_lo[e] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int tap = 2 * ks + h, ky = tap >> 2, kx = tap & 3;
            const int e = (2 * oyl + ky) * rowpx + 2 * ox + kx;
            const FragT ah = __builtin_bit_cast(FragT, img_hi[e]), al = __builtin_bit_cast(FragT, img_lo[e]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = mfma(al, bh[j][ks], acc[j]);
                acc[j] = mfma(ah, bl[j][ks], acc[j]);
                acc[j] = mfma(ah, bh[j][ks], acc[j]);
            }
        }
        const float gs = (p.gscale ? p.gscale[n / p.group_n] : 1.f) * p.mm_oscale;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = 32 * j + (lane & 31);
            const float b = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = 32 * wave + crow(r, lane);
                float v = acc[j][r] * gs + b;
                if (p.act == 1) v = lrelu_f(v);
                reinterpret_cast<float*>(outt)[prow * 64 + co] = v;
            }
        }
        __syncthreads();
        const size_t m0 = (size_t)(n * Ho + oy0) * Wo;
        if (actb) {
            // y = lrelu'(a) * v and the spectral-norm dot <dotx, v> of the same pass (conv_fwd_c8_kernel's ACTB form on fp32 tensors)
            const unsigned char* ab = static_cast<const unsigned char*>(p.ab_a);
            const unsigned char* xb = static_cast<const unsigned char*>(p.ab_dotx);
            for (int c = tid; c < MT * 16; c += 256) {
                const int px = c >> 4, ch = c & 15;
                const float4 v = reinterpret_cast<const float4*>(outt)[c];
                const float4 av = *reinterpret_cast<const float4*>(ab + ((m0 + px) * p.ab_lda) * 4 + ch * 16);
                float4 o;
                o.x = av.x > 0.f ? v.x : 0.2f * v.x; o.y = av.y > 0.f ? v.y : 0.2f * v.y;
                o.z = av.z > 0.f ? v.z : 0.2f * v.z; o.w = av.w > 0.f ? v.w : 0.2f * v.w;
                if (xb) {
                    const float4 xv = *reinterpret_cast<const float4*>(xb + ((m0 + px) * p.ab_lddot) * 4 + ch * 16);
                    dot_acc += (xv.x * v.x + xv.y * v.y) + (xv.z * v.z + xv.w * v.w);
                }
                *reinterpret_cast<float4*>(yb + ((m0 + px) * p.ldy) * 4 + ch * 16) = o;
            }
            continue;
        }
        for (int c = tid; c < MT * 16; c += 256)
            *reinterpret_cast<uint4*>(yb + ((m0 + (c >> 4)) * p.ldy) * 4 + (c & 15) * 16) = reinterpret_cast<const uint4*>(outt)[c];
    }
    if (actb && p.ab_dot_out) {
        __shared__ float dred[4];
        const float tot = block_sum<4>(dot_acc, dred);
        if (tid == 0) atomicAdd(p.ab_dot_out, tot);
    }
#endif
}
// (A/B knob: GCSSL_C8_FWD=0 sends these shapes back to the generic tiles)
bool c8_x3_on() { static const bool v = [] { const char* e = getenv("GCSSL_C8_X3"); return !(e && e[0] == '0'); }(); return v; }   // (A/B: the split-precision first-layer kernels)
bool c8_fwd_on() { static const bool v = [] { const char* e = getenv("GCSSL_C8_FWD"); return !(e && e[0] == '0'); }(); return v; }

// Tile plan of the split-precision forms.  Their operands are fp32 in memory (twice the bytes of a 16-bit tile per K element) and
// every kernel of this file is paced by its LDS fill rate, so FLOP per filled byte decides: 128x128 tiles (32 FLOP/B) where the
// column count allows, 128x64 (21) else, 64x64 (10.7) only for first layers; small-M launches fill the chip by splitting K (the
// outputs of this mode are fp32 and every epilogue but the first layers' is linear).  Returns the tile code (2 = 128x128,
// 1 = 128x64, 0 = 64x64) and sets *ks.  mult: 4 parity classes for the dgrad form.  GCSSL_X3_TILE / GCSSL_X3_KS: force (experiments).
int x3_plan(long M, int ncols, int nk, int mult, bool split_ok, int* ks) {
    static const int ftile = [] { const char* e = getenv("GCSSL_X3_TILE"); return e ? atoi(e) : -1; }();
    static const int fks = [] { const char* e = getenv("GCSSL_X3_KS"); return e ? atoi(e) : 0; }();
    static const long want = [] { const char* e = getenv("GCSSL_X3_WGS"); return e ? atol(e) : 384L; }();
    // measured (tools/x3_ab.sh, n = 768 critic shapes): 128x64 beats 128x128 on every forward (8 waves: 62.7 / 68.3 / 72.6 us against
    // 69.6 / 84.1 / 83.3 for c2 / c3 / c4) and ties on the data gradients -- the launches are issue- and skeleton-bound, not
    // fill-bound, so the extra workgroups in flight are worth more than the FLOP per byte; 128x128 stays behind GCSSL_X3_TILE=2
    int tile = (ncols >= 64 && M >= 128) ? 1 : 0;
    if (ftile >= 0 && (ftile <= tile || (ftile == 2 && ncols >= 128 && M >= 128))) tile = ftile;
    const int bm = tile ? 128 : 64, bn = tile == 2 ? 128 : 64;
    const long tiles = mult * ((M + bm - 1) / bm) * ((ncols + bn - 1) / bn);
    int k = 1;
    if (split_ok && ksplit_max() > 1 && tiles < want && nk >= 16) {
        k = (int)((want + tiles - 1) / tiles);
        if (k > ksplit_max()) k = ksplit_max();
        while (k > 1 && nk / k < 8) --k;
    }
    if (fks > 0 && split_ok) { k = fks; while (k > 1 && nk / k < 4) --k; }
    *ks = k;
    return tile;
}
template <typename T, int MM = 0>
int dispatch_fwd(ConvParams p, hipStream_t st) {
    if (MM) set_mm_scales(p, true);
    if constexpr (MM != 0) {
        const int nk = 16 * p.Cin / 32;
        int ks = 1;
        const int tile = p.Cin < 64 ? 0 : x3_plan(p.M, p.Cout, nk, 1, p.act == 0, &ks);
        {
            const int Wo = p.Wi / 2;
            if (p.Cin == 8 && p.Cout == 64 && (Wo == 16 || Wo == 32 || Wo == 64) && (p.Hi / 2) % (128 / Wo) == 0 && c8_fwd_on() && c8_x3_on() &&
                !p.split_stride && p.ldx % 4 == 0 && p.ldy % 4 == 0 && aligned16(p.x) && aligned16(p.w) && aligned16(p.y)) {
                if (p.plan_out) { *p.plan_out = 1; return GCSSL_OK; }
                const int R = 128 / Wo, rows = 2 * R + 2;
                const size_t tile_b = (size_t)2 * 64 * 17 * 16;                           // two weight images (> the 32-KB fp32 result tile)
                const size_t lds = (size_t)rows * (p.Wi + 2) * 32 + tile_b;
                static const int gmax = [] { const char* e = getenv("GCSSL_C8_GRID"); return e ? atoi(e) : 2 * cu_count(); }();
                const int ntiles = p.N * ((p.Hi / 2) / R), per = (ntiles + gmax - 1) / gmax, grid = (ntiles + per - 1) / per;
                GCSSL_LAUNCH((conv_fwd_c8_x3_kernel<typename SplitH<MM>::type>), dim3((unsigned)grid), dim3(256), lds, st, p, ntiles);
                return gcssl_launch_status();
            }
        }
        if (p.Cin >= 64) {
            p.ksplit = ks;
            if (ks > 1) {
                p.ktiles_per_split = (nk + ks - 1) / ks;
                int rc = zero_output(p, p.M, p.Cout, st);
                if (rc) return rc;
            }
            if (tile == 2) return launch_fwd<T, 128, 128, MM>(p, st);
            if (tile == 1) return launch_fwd<T, 128, 64, MM>(p, st);
            return launch_fwd<T, 64, 64, MM>(p, st);
        }
    }
    if constexpr (Is16<T>::v) {
        const int Wo = p.Wi / 2;
        if (p.Cin == 8 && p.Cout == 64 && (Wo == 16 || Wo == 32 || Wo == 64) && (p.Hi / 2) % (128 / Wo) == 0 && c8_fwd_on() &&
            p.ldx % 8 == 0 && p.ldy % (p.out_f32 ? 4 : 8) == 0 && aligned16(p.y)) {
            if (p.plan_out) { *p.plan_out = 1; return GCSSL_OK; }
            const int R = 128 / Wo, rows = 2 * R + 2;
            size_t tile_b = (size_t)128 * 64 * ((p.out_f32 || p.ab_a) ? 4 : 2);
            if (tile_b < (size_t)64 * 17 * 16) tile_b = (size_t)64 * 17 * 16;       // the region first holds the padded weight image
            const size_t lds = (size_t)rows * (p.Wi + 2) * 16 + tile_b;
            // two workgroups per CU, each walking its share of the tiles (768 x 32 x 32: 19.4 us with 768 workgroups of two tiles,
            // 16.2 us with 512 of three; GCSSL_C8_GRID: A/B)
            static const int gmax = [] { const char* e = getenv("GCSSL_C8_GRID"); return e ? atoi(e) : 2 * cu_count(); }();
            const int ntiles = p.N * ((p.Hi / 2) / R), per = (ntiles + gmax - 1) / gmax, grid = (ntiles + per - 1) / per;
            GCSSL_LAUNCH(conv_fwd_c8_kernel<T>, dim3((unsigned)grid), dim3(256), lds, st, p, ntiles);
            return gcssl_launch_status();
        }
    }
    if (Is16<T>::v && use_dma() && p.Cin >= 64) {
        const char* f = forced_tile();
        if (f && !strcmp(f, "ring") && p.Cout >= 128) return launch_ring<typename Op16<T>::type, 0>(p, st);
        if (f && !strcmp(f, "256x128") && p.Cout >= 128) return launch_big<typename Op16<T>::type, 256, 128, 0>(p, st);
        if (f && !strcmp(f, "256x64")) return launch_big<typename Op16<T>::type, 256, 64, 0>(p, st);
        if (f && !strcmp(f, "128x128") && p.Cout >= 128) return launch_fwd<T, 128, 128, MM>(p, st);
        if (f && !strcmp(f, "128x64")) return launch_fwd<T, 128, 64, MM>(p, st);
    }
    const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if constexpr (Is16<T>::v) {
        if (p.Cin >= 64 && p.Cout >= 128 && p.M >= 128 && !forced_tile()) {
            const bool lin = (p.out_f32 || std::is_same<T, float>::value) && p.act == 0;
            const int rk = ring_ksplit(t128, 16 * p.Cin / 64, lin);
            if (rk >= 1) {
                p.ksplit = rk;
                if (rk > 1) {
                    p.ktiles_per_split = (16 * p.Cin / 64 + rk - 1) / rk;
                    int rc = zero_output(p, p.M, p.Cout, st);
                    if (rc) return rc;
                }
                return launch_ring<typename Op16<T>::type, 0>(p, st);
            }
        }
    }
    if (p.Cout >= 128 && t128 >= tile128_threshold()) return launch_fwd<T, 128, 128, MM>(p, st);
    if (p.Cout >= 64 && (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64) >= tile_threshold()) return launch_fwd<T, 128, 64, MM>(p, st);
    const long t64 = (long)((p.M + 63) / 64) * ((p.Cout + 63) / 64);
    const int nk = 16 * p.Cin / BKOf<T>::v;
    const bool f32out = p.out_f32 || std::is_same<T, float>::value;
    const bool split_ok = f32out && p.act == 0;
    // small-M layers: split K.  With the LDS-DMA kernels a K step costs about the same for a 128x64 tile (8 waves) as
    // for a 64x64 one (4 waves), so prefer half as many, twice as deep... twice as WIDE workgroups with twice the split
    if (split_tile() == 128 && Is16<T>::v && use_dma() && p.Cin >= 64 && p.Cout >= 64 && p.M >= 128) {
        const long t = (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64);
        p.ksplit = pick_ksplit(t, nk, split_ok);
        if (p.ksplit > 1) {
            p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
            int rc = zero_output(p, p.M, p.Cout, st);
            if (rc) return rc;
            return launch_fwd<T, 128, 64, MM>(p, st);
        }
    }
    p.ksplit = pick_ksplit(t64, nk, split_ok);
    if (p.ksplit > 1) {
        p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
        int rc = zero_output(p, p.M, p.Cout, st);
        if (rc) return rc;
    }
    return launch_fwd<T, 64, 64, MM>(p, st);
}
// ---- forward conv + InstanceNorm + LeakyReLU in one launch (the FIN instantiations of conv_dma_kernel).
// Which form serves these shapes: 0 none (the caller keeps the conv -> fp32 z -> gcssl_in_act_fwd pair), 1 the loader /
// consumer ring (128 x 128 tiles, one round of the chip), 2 128 x 64 tiles, 3 64 x 64 tiles.  No K split: the statistics need
// the complete sums, so shapes that only fill the chip by splitting K (B-sample c4) stay on the unfused pair.
int fin_form(const ConvParams& p) {
    static const int on = [] { const char* e = getenv("GCSSL_FIN"); return e ? atoi(e) : 1; }();
    static const long ring_min = [] { const char* e = getenv("GCSSL_FIN_RING_MIN"); return e ? atol(e) : 96L; }();
    static const long min_wgs = [] { const char* e = getenv("GCSSL_FIN_WGS"); return e ? atol(e) : 192L; }();
    const int HW = (p.Hi / 2) * (p.Wi / 2);
    if (!on || !use_dma() || dma_waves() != 8 || HW > 64 || HW < 4 || p.Cin < 64 || p.Cout < 64) return 0;
    const long tm128 = (p.M + 127) / 128;
    const long t128 = tm128 * ((p.Cout + 127) / 128);
    // ring form with 64-row tiles (4 consumer + 8 loader waves): for shapes whose 128 x 128 tiles would leave most of the chip
    // idle (D.c4 / G.down4 at n = 768: 96 tiles) but whose 64 x 128 tiles fill one round (192)
    const long t64x128 = (long)((p.M + 63) / 64) * ((p.Cout + 127) / 128);
    if ((on & 8) == 0 && (on & 2) == 0 && p.Cout >= 128 && p.M >= 64 && HW <= 16 && t128 < 160 && t64x128 >= 160 && t64x128 <= cu_count()) return 5;
    if ((on & 2) == 0 && p.Cout >= 128 && p.M >= 128 && t128 <= cu_count() && t128 >= ring_min) return 1;
    // 8x8 maps with clearly more 128 x 64 tiles than resident workgroups: the persistent form (statistics in registers)
    if ((on & 4) == 0 && HW == 64 && !p.in_mask && !p.in_apre && persist_mode() && tm128 * ((p.Cout + 63) / 64) > 2 * cu_count() + cu_count() / 2 &&
        p.y_bytes) return 4;
    if (p.M >= 128 && tm128 * ((p.Cout + 63) / 64) >= min_wgs) return 2;
    if ((long)((p.M + 63) / 64) * ((p.Cout + 63) / 64) >= min_wgs) return 3;
    return 0;
}
template <typename T>
int dispatch_fwd_in(ConvParams p, hipStream_t st) {
    if constexpr (Is16<T>::v) {
        typedef typename Op16<T>::type O;
        const int form = fin_form(p);
        p.ksplit = 1;
        if (form == 1) {
            dim3 grid((p.M + 127) / 128, (p.Cout + 127) / 128, 1);
            GCSSL_LAUNCH((conv_dma_kernel<O, 128, 128, 0, 4, 2, false, 8, 4, false, true>), grid, dim3(1024), 0, st, p);
        } else if (form == 2) {
            dim3 grid((p.M + 127) / 128, (p.Cout + 63) / 64, 1);
            GCSSL_LAUNCH((conv_dma_kernel<O, 128, 64, 0, 4, 2, false, 0, 3, true, true>), grid, dim3(512), 0, st, p);
        } else if (form == 3) {
            dim3 grid((p.M + 63) / 64, (p.Cout + 63) / 64, 1);
            GCSSL_LAUNCH((conv_dma_kernel<O, 64, 64, 0, 2, 2, false, 0, 3, true, true>), grid, dim3(256), 0, st, p);
        } else if (form == 5) {
            dim3 grid((p.M + 63) / 64, (p.Cout + 127) / 128, 1);
            GCSSL_LAUNCH((conv_dma_kernel<O, 64, 128, 0, 2, 2, false, 8, 4, false, true>), grid, dim3(768), 0, st, p);
        } else if (form == 4) {
            const int tm = (p.M + 127) / 128, tn = (p.Cout + 63) / 64;
            GCSSL_LAUNCH((conv_dma_persist_kernel<O, 128, 64, 0, 4, 2, false, true>), dim3(2 * cu_count()), dim3(512), 0, st, p, tm, tn, tm * tn);
        } else {
            return GCSSL_EBADSHAPE;
        }
        return gcssl_launch_status();
    }
    return GCSSL_EBADDTYPE;
}
// ------------------------------------------------------------------------------------------
// data gradient of the first layer (Cin padded to 8, Cout = 64: D.c1 in the gradient-penalty chain; 0.8 GFLOP whose result
// is 8 channels wide), built like conv_fwd_c8_kernel: a workgroup owns R dy rows of one sample (R * Wo = 32), loads the
// R + 2 rows its outputs touch once with coalesced 16-byte loads into an LDS image (zero pixel at both row ends, zero rows
// outside the map; the 8 chunks of a 128-byte pixel XOR-swizzled by the pixel index so that fragment reads of consecutive
// pixels spread over the banks); wave = output parity class (py, px) = its 2 x 2 taps; an A fragment is one ds_read_b128;
// the wave's 16 B fragments (Wt rows of the 8 real input channels, columns 8..31 zero) come through LDS once per
// workgroup and stay in registers; the 2R x Wi x 8 result tile is assembled in LDS in memory order and leaves as full rows.
//   dx[n][2 jy + py][2 jx + px][ci] = gscale * sum_{a,b,co} dy[n][oy_a][ox_b][co] Wt[ci][ky_a * 4 + kx_b][co]
//   a = 0: ky = 1 + py, oy = jy;   a = 1: ky = 3 - 3 py, oy = jy - 1 + 2 py   (and the same in x)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_dgrad_c8_kernel(ConvParams p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    constexpr int WROW = 129;                                            // a ci row of Wt: 128 chunks + 1 pad chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char c8_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1, R = 32 >> p.lgWo, rows = R + 2, rowpx = Wo + 2;
    const int tiles_per_n = Ho / R;
    uint4* img = reinterpret_cast<uint4*>(c8_lds);                       // [rows][Wo + 2] pixels x 8 chunks
    unsigned char* outt = c8_lds + (size_t)rows * rowpx * 128;           // [2R][Wi][8] result tile; first: the weight image
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    // dy image: chunk c -> (row, pixel, 16-byte chunk); image row i = dy row oy0 - 1 + i.  Fetched into registers one tile
    // ahead (<= 3 chunks per thread), like conv_fwd_c8_kernel's input image.
    constexpr int NC = 3;
    const int nchunk = rows * Wo * 8;                                    // 512 / 768 for Wo = 16 / 32
    u32x4 pre[NC];
    auto fetch = [&](int tile) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            const int ch = c & 7, pxl = (c >> 3) & (Wo - 1), row = c >> (3 + p.lgWo), oy = oy0 - 1 + row;
            const bool ok = c < nchunk && (unsigned)oy < (unsigned)Ho;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((n * Ho + oy) * Wo + pxl) * p.ldx + ch * 8) * 2) : OOB, 0, 0);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);                     // (in flight under the weight prologue below)
    {
        uint4* wimg = reinterpret_cast<uint4*>(outt);
        for (int c = tid; c < 8 * 128; c += 256)
            wimg[(c >> 7) * WROW + (c & 127)] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 16u, 0, 0));
        __syncthreads();
    }
    const int h = lane >> 5, ci = lane & 31;
    FragT bf[2][2][4];                                                   // [a][b][ks]: channels 16 ks + 8 h .. + 7 of tap (a, b)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tap = (a ? 3 - 3 * py : 1 + py) * 4 + (b ? 3 - 3 * px : 1 + px);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 w = reinterpret_cast<const uint4*>(outt)[(ci < 8 ? ci : 0) * WROW + tap * 8 + 2 * ks + h];
                bf[a][b][ks] = __builtin_bit_cast(FragT, ci < 8 ? w : make_uint4(0u, 0u, 0u, 0u));
            }
        }
    const int r = lane & 31, jyl = r >> p.lgWo, jx = r & (Wo - 1);       // the lane's output pixel (of its class) in the tile
    const bool f32out = p.out_f32;
    const int es = f32out ? 4 : 2;
    unsigned char* yb = static_cast<unsigned char*>(p.y);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            if (c < nchunk) {
                const int ch = c & 7, pxl = (c >> 3) & (Wo - 1), row = c >> (3 + p.lgWo);
                const int pix = row * rowpx + pxl + 1;
                img[pix * 8 + (ch ^ (pix & 7))] = __builtin_bit_cast(uint4, pre[i]);
            }
        }
        if (tid < 2 * rows * 8) {                                        // the two pad pixels of every row
            const int pix = (tid >> 4) * rowpx + (((tid >> 3) & 1) ? Wo + 1 : 0);
            img[pix * 8 + (tid & 7)] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = (a ? jyl - 1 + 2 * py : jyl) + 1, col = (b ? jx - 1 + 2 * px : jx) + 1;
                const int pix = row * rowpx + col;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = mfma(__builtin_bit_cast(FragT, img[pix * 8 + ((2 * ks + h) ^ (pix & 7))]), bf[a][b][ks], acc);
            }
        // result tile in memory order: [2R rows][Wi][8 channels]
        const float gs = p.gscale ? p.gscale[n / p.group_n] : 1.f;
        if (ci < 8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rr = crow(q, lane), oyl = 2 * (rr >> p.lgWo) + py, oxl = 2 * (rr & (Wo - 1)) + px;
                const float v = acc[q] * gs;
                const int e = (oyl * p.Wi + oxl) * 8 + ci;
                if (f32out) reinterpret_cast<float*>(outt)[e] = v;
                else reinterpret_cast<unsigned short*>(outt)[e] = (unsigned short)Bits16<T>::enc(v);
            }
        }
        __syncthreads();
        const int cpp = f32out ? 2 : 1;                                  // 16-byte chunks per output pixel (8 channels)
        const size_t pix0 = (size_t)(n * p.Hi + 2 * oy0) * p.Wi;         // first output pixel of the tile (2R full rows)
        for (int c = tid; c < 2 * R * p.Wi * cpp; c += 256) {
            const int opx = f32out ? c >> 1 : c, chn = f32out ? c & 1 : 0;
            *reinterpret_cast<uint4*>(yb + ((pix0 + opx) * p.ldy) * es + chn * 16) = reinterpret_cast<const uint4*>(outt)[c];
        }
    }
#endif
}
// ... and its split-precision form (fp32 dy and Wt, hi / lo images, three MFMAs per step; see conv_fwd_c8_x3_kernel).  The
// generic split kernel pads the 8 result channels to a 64-column tile in 4096 workgroups: 47.5 us for 0.8 GFLOP.
template <typename H>
__global__ __launch_bounds__(256) void conv_dgrad_c8_x3_kernel(ConvParams p, int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<H>::type FragT;
    constexpr int WROW = 129;
    extern __shared__ __attribute__((aligned(16))) unsigned char c8_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1, R = 32 >> p.lgWo, rows = R + 2, rowpx = Wo + 2;
    const int tiles_per_n = Ho / R;
    const int npix = rows * rowpx;
    uint4* img_hi = reinterpret_cast<uint4*>(c8_lds);                    // [rows][Wo + 2] pixels x 8 chunks of 8 hi halves
    uint4* img_lo = img_hi + npix * 8;
    unsigned char* outt = c8_lds + (size_t)npix * 256;                   // [2R][Wi][8] fp32 result tile; first: the two weight images
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    constexpr int NC = 3;
    const int nchunk = rows * Wo * 8;                                    // 8-channel groups of the image (<= 3 per thread)
    u32x4 pre[NC][2];
    auto fetch = [&](int tile) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            const int ch = c & 7, pxl = (c >> 3) & (Wo - 1), row = c >> (3 + p.lgWo), oy = oy0 - 1 + row;
            const bool ok = c < nchunk && (unsigned)oy < (unsigned)Ho;
            const unsigned off = ok ? (unsigned)((((n * Ho + oy) * Wo + pxl) * p.ldx + ch * 8) * 4) : OOB;
            pre[i][0] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            pre[i][1] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? off + 16u : OOB, 0, 0);
        }
    };
    auto split8 = [](const u32x4& a, const u32x4& b, uint4& hi, uint4& lo) {
        uint2 h0, l0, h1, l1;
        split4<H>(__builtin_bit_cast(float4, a), h0, l0);
        split4<H>(__builtin_bit_cast(float4, b), h1, l1);
        hi = make_uint4(h0.x, h0.y, h1.x, h1.y); lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    {
        uint4* whi = reinterpret_cast<uint4*>(outt);
        uint4* wlo = whi + 8 * WROW;
        for (int c = tid; c < 8 * 128; c += 256) {                       // c -> (ci, 8 consecutive co of a tap)
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 32u, 0, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)c * 32u + 16u, 0, 0);
            uint4 hi, lo;
            split8(a, b, hi, lo);
            whi[(c >> 7) * WROW + (c & 127)] = hi; wlo[(c >> 7) * WROW + (c & 127)] = lo;
        }
        __syncthreads();
    }
    const int h = lane >> 5, ci = lane & 31;
    FragT bh[2][2][4], bl[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int tap = (a ? 3 - 3 * py : 1 + py) * 4 + (b ? 3 - 3 * px : 1 + px);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int e = (ci < 8 ? ci : 0) * WROW + tap * 8 + 2 * ks + h;
                const uint4 wh = reinterpret_cast<const uint4*>(outt)[e], wl = reinterpret_cast<const uint4*>(outt)[8 * WROW + e];
                bh[a][b][ks] = __builtin_bit_cast(FragT, ci < 8 ? wh : make_uint4(0u, 0u, 0u, 0u));
                bl[a][b][ks] = __builtin_bit_cast(FragT, ci < 8 ? wl : make_uint4(0u, 0u, 0u, 0u));
            }
        }
    const int r = lane & 31, jyl = r >> p.lgWo, jx = r & (Wo - 1);
    unsigned char* yb = static_cast<unsigned char*>(p.y);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_n, oy0 = (tile - n * tiles_per_n) * R;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = tid + 256 * i;
            if (c < nchunk) {
                const int ch = c & 7, pxl = (c >> 3) & (Wo - 1), row = c >> (3 + p.lgWo);
                const int pix = row * rowpx + pxl + 1;
                uint4 hi, lo;
                split8(pre[i][0], pre[i][1], hi, lo);
                img_hi[pix * 8 + (ch ^ (pix & 7))] = hi; img_lo[pix * 8 + (ch ^ (pix & 7))] = lo;
            }
        }
        if (tid < 2 * rows * 8) {
            const int pix = (tid >> 4) * rowpx + (((tid >> 3) & 1) ? Wo + 1 : 0);
            img_hi[pix * 8 + (tid & 7)] = make_uint4(0u, 0u, 0u, 0u); img_lo[pix * 8 + (tid & 7)] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = (a ? jyl - 1 + 2 * py : jyl) + 1, col = (b ? jx - 1 + 2 * px : jx) + 1;
                const int pix = row * rowpx + col;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int e = pix * 8 + ((2 * ks + h) ^ (pix & 7));
                    const FragT ah = __builtin_bit_cast(FragT, img_hi[e]), al = __builtin_bit_cast(FragT, img_lo[e]);
                    acc = mfma(al, bh[a][b][ks], acc);
                    acc = mfma(ah, bl[a][b][ks], acc);
                    acc = mfma(ah, bh[a][b][ks], acc);
                }
            }
        const float gs = (p.gscale ? p.gscale[n / p.group_n] : 1.f) * p.mm_oscale;
        if (ci < 8) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rr = crow(q, lane), oyl = 2 * (rr >> p.lgWo) + py, oxl = 2 * (rr & (Wo - 1)) + px;
                reinterpret_cast<float*>(outt)[(oyl * p.Wi + oxl) * 8 + ci] = acc[q] * gs;
            }
        }
        __syncthreads();
        const size_t pix0 = (size_t)(n * p.Hi + 2 * oy0) * p.Wi;
        for (int c = tid; c < 2 * R * p.Wi * 2; c += 256)
            *reinterpret_cast<uint4*>(yb + ((pix0 + (c >> 1)) * p.ldy) * 4 + (c & 1) * 16) = reinterpret_cast<const uint4*>(outt)[c];
    }
#endif
}
// (A/B knob: GCSSL_C8_DGRAD=0 sends the shape back to the generic tiles)
bool c8_dgrad_on() { static const bool v = [] { const char* e = getenv("GCSSL_C8_DGRAD"); return !(e && e[0] == '0'); }(); return v; }

int dgrad_img_on(const ConvParams& p);                                   // (dgrad_img_kernel, below)
template <typename O, int OUT> void launch_dgrad_img(const ConvParams& p, hipStream_t st);

template <typename T, int MM = 0>
int dispatch_dgrad(ConvParams p, hipStream_t st) {
    if (MM) set_mm_scales(p, true);
    if constexpr (MM != 0) {
        {
            const int Wo = p.Wi / 2;
            if (p.Cin == 8 && p.Cout == 64 && (Wo == 16 || Wo == 32) && (p.Hi / 2) % (32 / Wo) == 0 && c8_dgrad_on() && c8_x3_on() &&
                !p.split_stride && !p.ab_a && p.ldx % 4 == 0 && p.ldy % 4 == 0 && aligned16(p.x) && aligned16(p.w) && aligned16(p.y)) {
                if (p.plan_out) { *p.plan_out = 1; return GCSSL_OK; }
                const int R = 32 / Wo, rows = R + 2;
                size_t tile_b = (size_t)2 * R * p.Wi * 8 * 4;
                if (tile_b < (size_t)2 * 8 * 129 * 16) tile_b = (size_t)2 * 8 * 129 * 16;   // the region first holds the two padded weight images
                const size_t lds = (size_t)rows * (Wo + 2) * 256 + tile_b;
                static const int gmax = [] { const char* e = getenv("GCSSL_C8_DGRID"); return e ? atoi(e) : 2 * cu_count(); }();
                const int ntiles = p.N * ((p.Hi / 2) / R), per = (ntiles + gmax - 1) / gmax, grid = (ntiles + per - 1) / per;
                GCSSL_LAUNCH((conv_dgrad_c8_x3_kernel<typename SplitH<MM>::type>), dim3((unsigned)grid), dim3(256), lds, st, p, ntiles);
                return gcssl_launch_status();
            }
        }
        if (p.Cin >= 64) {
            const int nk = 4 * p.Cout / 32;
            int ks = 1;
            const int tile = x3_plan(p.M, p.Cin, nk, 4, !p.ab_a, &ks);              // (the activation-backward epilogue is not linear)
            p.ksplit = ks;
            if (ks > 1) {
                p.ktiles_per_split = (nk + ks - 1) / ks;
                int rc = zero_output(p, (long)p.N * p.Hi * p.Wi, p.Cin, st);
                if (rc) return rc;
            }
            if (tile == 2) return launch_dgrad<T, 128, 128, MM>(p, st);
            if (tile == 1) return launch_dgrad<T, 128, 64, MM>(p, st);
            return launch_dgrad<T, 64, 64, MM>(p, st);
        }
    }
    if constexpr (Is16<T>::v) {
        // the 64 <- 128 layer on 16 x 16 maps, fp32 dx, >= 3 samples per CU: dy maps resident in LDS (dgrad_img_kernel)
        if (p.out_f32 && !p.split_stride && p.ldy % 4 == 0 && aligned16(p.y) && p.y_bytes && !forced_tile() && dgrad_img_on(p)) {
            if (p.plan_out) { *p.plan_out = 1; return GCSSL_OK; }
            launch_dgrad_img<typename Op16<T>::type, 1>(p, st);
            return gcssl_launch_status();
        }
    }
    if constexpr (Is16<T>::v) {
        const int Wo = p.Wi / 2;
        if (p.Cin == 8 && p.Cout == 64 && (Wo == 16 || Wo == 32) && (p.Hi / 2) % (32 / Wo) == 0 && c8_dgrad_on() &&
            p.ldx % 8 == 0 && p.ldy % (p.out_f32 ? 4 : 8) == 0 && aligned16(p.y)) {
            if (p.plan_out) { *p.plan_out = 1; return GCSSL_OK; }
            const int R = 32 / Wo, rows = R + 2;
            size_t tile_b = (size_t)2 * R * p.Wi * 8 * (p.out_f32 ? 4 : 2);
            if (tile_b < (size_t)8 * 129 * 16) tile_b = (size_t)8 * 129 * 16;       // the region first holds the padded weight image
            const size_t lds = (size_t)rows * (Wo + 2) * 128 + tile_b;
            // (256 x 32 x 32, 2048 tiles: 12.9 us with 1024 workgroups, 11.3 us with 512; GCSSL_C8_DGRID: A/B)
            static const int gmax = [] { const char* e = getenv("GCSSL_C8_DGRID"); return e ? atoi(e) : 2 * cu_count(); }();
            const int ntiles = p.N * ((p.Hi / 2) / R), per = (ntiles + gmax - 1) / gmax, grid = (ntiles + per - 1) / per;
            GCSSL_LAUNCH(conv_dgrad_c8_kernel<T>, dim3((unsigned)grid), dim3(256), lds, st, p, ntiles);
            return gcssl_launch_status();
        }
    }
    if (Is16<T>::v && use_dma() && p.Cin >= 64) {
        const char* f = forced_tile();
        if (f && !strcmp(f, "ring") && p.Cin >= 128) return launch_ring<typename Op16<T>::type, 1>(p, st);
        if (f && !strcmp(f, "256x128") && p.Cin >= 128) return launch_big<typename Op16<T>::type, 256, 128, 1>(p, st);
        if (f && !strcmp(f, "256x64")) return launch_big<typename Op16<T>::type, 256, 64, 1>(p, st);
        if (f && !strcmp(f, "128x128") && p.Cin >= 128) return launch_dgrad<T, 128, 128, MM>(p, st);
        if (f && !strcmp(f, "128x64")) return launch_dgrad<T, 128, 64, MM>(p, st);
    }
    const long t128 = 4L * ((p.M + 127) / 128) * ((p.Cin + 127) / 128);
    if constexpr (Is16<T>::v) {
        if (p.Cin >= 128 && p.Cout >= 64 && p.M >= 128 && !forced_tile()) {
            const bool lin = p.out_f32 || std::is_same<T, float>::value;
            const int rk = ring_ksplit(t128, 4 * p.Cout / 64, lin);
            if (rk >= 1) {
                p.ksplit = rk;
                if (rk > 1) {
                    p.ktiles_per_split = (4 * p.Cout / 64 + rk - 1) / rk;
                    int rc = zero_output(p, (long)p.N * p.Hi * p.Wi, p.Cin, st);
                    if (rc) return rc;
                }
                return launch_ring<typename Op16<T>::type, 1>(p, st);
            }
        }
    }
    if (p.Cin >= 128 && t128 >= tile128_threshold()) return launch_dgrad<T, 128, 128, MM>(p, st);
    if (p.Cin >= 64 && 4L * ((p.M + 127) / 128) * ((p.Cin + 63) / 64) >= tile_threshold()) return launch_dgrad<T, 128, 64, MM>(p, st);
    const long t64 = 4L * ((p.M + 63) / 64) * ((p.Cin + 63) / 64);
    const int nk = 4 * p.Cout / BKOf<T>::v;
    const bool f32out = p.out_f32 || std::is_same<T, float>::value;
    if (split_tile() == 128 && Is16<T>::v && use_dma() && p.Cin >= 64 && p.Cout >= 64 && p.M >= 128) {
        const long t = 4L * ((p.M + 127) / 128) * ((p.Cin + 63) / 64);
        p.ksplit = pick_ksplit(t, nk, f32out);
        if (p.ksplit > 1) {
            p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
            int rc = zero_output(p, (long)p.N * p.Hi * p.Wi, p.Cin, st);
            if (rc) return rc;
            return launch_dgrad<T, 128, 64, MM>(p, st);
        }
    }
    p.ksplit = pick_ksplit(t64, nk, f32out);
    if (p.ksplit > 1) {
        p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
        int rc = zero_output(p, (long)p.N * p.Hi * p.Wi, p.Cin, st);
        if (rc) return rc;
    }
    return launch_dgrad<T, 64, 64, MM>(p, st);      // Cin < 64 (first layer, Cin padded to 8): masked columns
}

// ------------------------------------------------------------------------------------------
// Data gradient of the 64 -> 128 layer on 16 x 16 inputs (D.c2 of the 32 x 32 configurations; dy maps of 8 x 8 x 128) with the
// dy maps RESIDENT in LDS, for batches of at least three samples per CU.  A workgroup owns three whole samples (768 samples =
// 256 workgroups = one round of the chip), 12 waves:
//   * their dy maps go to LDS ONCE, as a zero-haloed 10 x 10 image per sample (256-byte pixels, the sixteen 16-byte chunks
//     XOR-swizzled by the pixel index); an MFMA A fragment of any (parity class, tap) is one ds_read_b128 at a lane-constant
//     pixel + a step-uniform offset -- the im2col happens in the LDS address, as in conv_dgrad_c8_kernel.  (The ring forms
//     refill a 128-row A tile for every (class, tap): each dy pixel crosses the CU's LDS-DMA path 16 times.)
//   * only the weights stream: 16 K steps (4 classes x 4 taps, all 128 dy channels: K = 128) of one 64 ci x 128 co tile (16 KB)
//     through a 3-slot LDS-DMA ring issued by waves 0-3 ALONE;
//   * a class (192 rows x 64 ci, fp32) leaves sample by sample through a 17-KB LDS tile in 8-channel chunks, and all of that
//     memory traffic belongs to waves 4-11: vmcnt retires in order, so a wave that waits for its next weight tile also waits
//     for every store it issued before -- waves that issue no DMA never wait for a store, and the 50 MB of the epilogue
//     (stored activation in, dzs out) flow under the next class's MFMAs.  OUT 0 = gcssl_act_bwd's elementwise part and sums
//     (dzs = lrelu'(a) dx gscale in the compute dtype; the activation chunks are fetched at the head of the class), OUT 1 =
//     fp32 dx * gscale.
// ------------------------------------------------------------------------------------------
template <typename T, int OUT>
__global__ __launch_bounds__(768) void dgrad_img_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Frag16<T>::type FragT;
    constexpr int SPW = 3, NW = 12, NCH = 2 * SPW;                        // NCH: chunks per epilogue thread and class
    constexpr int IMG = SPW * 100 * 256, SLOT = 16384, RS = 64 * 4 + 16;  // image bytes; ring slot; stage row stride (64 columns)
    __shared__ __attribute__((aligned(16))) unsigned char lds[IMG + 3 * SLOT + 64 * RS];
    unsigned char* ring = lds + IMG;
    unsigned char* stage = ring + 3 * SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * SPW;
    // roles: waves 0-5 run the MFMAs (one sample's 64 rows x 32 ci each), 6-7 issue the weight DMA, 8-11 own the epilogue
    const bool mm = wave < 6, dma = wave == 6 || wave == 7, epi = wave >= 8;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    // ---- the dy images: LDS slot g (16 bytes) = pixel g >> 4, physical chunk g & 15 <- logical chunk (g & 15) ^ (pixel & 15);
    // halo pixels and samples past N read OOB = 0.  75 wave-wide DMA instructions over the 12 waves.
    for (int wi = wave; wi < 25 * SPW; wi += NW) {
        const int g = wi * 64 + lane, pi = g >> 4, lc = (g & 15) ^ (pi & 15);
        const int s = pi / 100, rem = pi - s * 100, iy = rem / 10 - 1, ix = rem - (rem / 10) * 10 - 1;
        const bool ok = (unsigned)iy < 8u && (unsigned)ix < 8u && n0 + s < p.N;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(lds + wi * 1024), 16,
            ok ? (unsigned)(((((n0 + s) * 8 + iy) * 8 + ix) * p.ldx + lc * 8) * 2) : OOB, 0, 0, 0);
    }
    // ---- weight ring (waves 6-7): step q = (class, tap pair); 1024 chunks per tile: chunk c -> channel half c >> 9, ci row
    // (c >> 3) & 63, 16-byte piece c & 7 (swizzled); DMA thread td fetches chunks td + 128 i, i < 8
    int plain_slot = SLOT;                                               // (plain int: see conv_dma_kernel's note on the host pass)
    const int td = tid - 384;
    auto issue = [&](int q, int slot) {
        const int cls = q >> 2, t4 = q & 3, py = cls >> 1, px = cls & 1;
        const int tap = (1 - py + 2 * (t4 >> 1)) * 4 + (1 - px + 2 * (t4 & 1));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = td + 128 * i, row = (c >> 3) & 63, lcw = (c & 7) ^ ((row >> 1) & 7);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_p)(ring + slot * plain_slot + ((wave - 6) + 2 * i) * 1024), 16,
                (unsigned)(((row * 16 + tap) * 128 + (c >> 9) * 64 + lcw * 8) * 2), 0, 0, 0);
        }
    };
    if (dma) { issue(0, 0); issue(1, 1); }
    // ---- MFMA role: wave = (sample rb, ci half wc); lane -> rows 32 i + (lane & 31) of the sample's class block, i < 2
    const int rb = wave >> 1, wc = wave & 1, h = lane >> 5;
    int pix0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pix0[i] = rb * 100 + (4 * i + ((lane & 31) >> 3) + 1) * 10 + (lane & 7) + 1;
    const int brow = wc * 32 + (lane & 31);
    const unsigned char* bbase = ring + brow * 128;
    const int bsw = (brow >> 1) & 7;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // ---- epilogue role: thread te -> chunks (sample s, row (te >> 3) + 32 j) x channels 8 ech .. 8 ech + 7, s < 3, j < 2
    const int te = tid - 512, erow0 = (te >> 3) & 31, ech = te & 7;
    const __amdgpu_buffer_rsrc_t ar = make_rsrc(p.ab_a, OUT == 0 ? p.ab_bytes : 0u), yr = make_rsrc(p.y, p.y_bytes);
    float bias8[8], gsv[SPW];
    int grpv[SPW];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (OUT == 0 && epi && p.ab_bias) ? p.ab_bias[ech * 8 + e] : 0.f;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int n = n0 + s;
        grpv[s] = (epi && (p.gscale || (OUT == 0 && p.ab_cdot))) ? (int)(((float)n + 0.5f) * p.inv_group_n) : 0;
        gsv[s] = (epi && p.gscale && n < p.N) ? p.gscale[grpv[s]] : 1.f;
    }
    float sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sdg[4] = {0.f, 0.f, 0.f, 0.f};
    int nsat = 0;
    uint4 areg[NCH];                                                     // the finished class's activation chunks (fetched at its end)
    float vreg[NCH][8];                                                  // ... and its values of this thread's six chunks
    int pcls = -1;                                                       // class whose chunks still have to leave
    // one chunk (slot k = 2 s + j of class c: 8 channels of one output pixel) from registers to memory
    auto apply = [&](int c, int k) __attribute__((always_inline)) {      // (inlined with a constant k everywhere: the register arrays must
        const int s = k >> 1, j = k & 1;                                 //  not be indexed by a run-time value, they would go to scratch)
        const int n = n0 + s, py = c >> 1, px = c & 1, erow = erow0 + 32 * j;
        const bool live = n < p.N;
        const unsigned pix = (unsigned)((n * 16 + 2 * (erow >> 3) + py) * 16 + 2 * (erow & 7) + px);
        const float gs = gsv[s];
        if constexpr (OUT == 0) {
            const unsigned awv[4] = {areg[k].x, areg[k].y, areg[k].z, areg[k].w};
            float o[8], sd = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = Bits16<T>::dec(awv[e >> 1] >> (16 * (e & 1)));
                const float dz = av > 0.f ? vreg[k][e] : 0.2f * vreg[k][e];
                const float zv = av > 0.f ? av : 5.0f * av;              // invert LeakyReLU(0.2)
                o[e] = dz * gs;
                sb[e] += dz; sd += o[e] * (zv - bias8[e]);
            }
            const int grp = grpv[s];
            if (grp == 0) sdg[0] += sd; else if (grp == 1) sdg[1] += sd; else if (grp == 2) sdg[2] += sd; else sdg[3] += sd;
            nsat += sat_hits<T>(o);
            u32x4 w; w[0] = pack2<T>(o[0], o[1]); w[1] = pack2<T>(o[2], o[3]); w[2] = pack2<T>(o[4], o[5]); w[3] = pack2<T>(o[6], o[7]);
            __builtin_amdgcn_raw_buffer_store_b128(w, yr, live ? (pix * (unsigned)p.ldy + ech * 8u) * 2u : OOB, 0, 0);
        } else {
            const unsigned off = live ? (pix * (unsigned)p.ldy + ech * 8u) * 4u : OOB;
            u32x4 w0, w1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { w0[e] = __builtin_bit_cast(unsigned, vreg[k][e] * gs); w1[e] = __builtin_bit_cast(unsigned, vreg[k][4 + e] * gs); }
            __builtin_amdgcn_raw_buffer_store_b128(w0, yr, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(w1, yr, off == OOB ? OOB : off + 16u, 0, 0);
        }
    };
    // ---- one loop per role (the register sets of the roles are then allocated independently: acc + fragments for the MFMA
    // waves, the six staged chunks + their activation chunks for the epilogue waves); every role executes the same barrier
    // sequence: one per K step, and five around the three sample blocks when a class completes.
    if (mm) {
        int slot = 0;
        for (int q = 0; q < 16; ++q) {
            const int cls = q >> 2, t4 = q & 3, py = cls >> 1, px = cls & 1;
            if (q == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the image
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int doff = (py - (t4 >> 1)) * 10 + (px - (t4 & 1));
            const unsigned char* bt = bbase + slot * SLOT;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {                             // channel halves of the K = 128 step: 12 fragments in flight
                FragT a[2][4], b[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    b[kk] = __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(bt + hf * 8192 + (((2 * kk + h) ^ bsw) << 4)));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int pi = pix0[i] + doff;
                        a[i][kk] = __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(lds + pi * 256 + (((8 * hf + 2 * kk + h) ^ (pi & 15)) << 4)));
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = mfma(a[i][kk], b[kk], acc[i]);
            }
            slot = slot == 2 ? 0 : slot + 1;
            if (t4 != 3) continue;
            // the class is complete: this wave's sample block goes to the stage in its turn
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                if (s) __builtin_amdgcn_s_barrier();                     // the previous sample's block has been read
                if (rb == s) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            *reinterpret_cast<float*>(stage + (32 * i + crow(r, lane)) * RS + (wc * 32 + (lane & 31)) * 4) = acc[i][r];
                            acc[i][r] = 0.f;
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    } else if (dma) {
        int slot = 0;
        for (int q = 0; q < 16; ++q) {
            // tile q has landed once only tile q + 1's eight DMA instructions can still be outstanding (these waves issue no other
            // vector memory operation after the prologue)
            if (q == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // tile q is published; tile q - 1's slot is free
            if (q + 2 < 16) issue(q + 2, slot == 0 ? 2 : slot - 1);
            slot = slot == 2 ? 0 : slot + 1;
            if ((q & 3) != 3) continue;
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                if (s) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        for (int q = 0; q < 16; ++q) {
            const int cls = q >> 2, t4 = q & 3, py = cls >> 1, px = cls & 1;
            if (q == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the image
            __builtin_amdgcn_s_barrier();
            if (pcls >= 0) {
                // the PREVIOUS class's chunks leave under this class's MFMAs: two per step in steps 1, 2, 3 (their activation
                // chunks were fetched when the class completed)
                if (t4 == 1) { apply(pcls, 0); apply(pcls, 1); }
                else if (t4 == 2) { apply(pcls, 2); apply(pcls, 3); }
                else if (t4 == 3) { apply(pcls, 4); apply(pcls, 5); }
            }
            if (t4 != 3) continue;
            // the class is complete: its sample blocks, fp32, from the stage into this thread's six chunk slots
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                if (s) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 v0 = *reinterpret_cast<const float4*>(stage + (erow0 + 32 * j) * RS + ech * 32);
                    const float4 v1 = *reinterpret_cast<const float4*>(stage + (erow0 + 32 * j) * RS + ech * 32 + 16);
                    float* v = vreg[2 * s + j];
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // returned before the block is overwritten
            }
            pcls = cls;
            if constexpr (OUT == 0) {                                    // the finished class's activation chunks, in the order they are applied
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int n = n0 + (k >> 1), erow = erow0 + 32 * (k & 1);
                    const unsigned pix = (unsigned)((n * 16 + 2 * (erow >> 3) + py) * 16 + 2 * (erow & 7) + px);
                    areg[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ar,
                        n < p.N ? (pix * (unsigned)p.ab_lda + ech * 8u) * 2u : OOB, 0, 0));
                }
            }
        }
    }
    if (epi && pcls >= 0) {                                              // the last class's chunks
        apply(pcls, 0); apply(pcls, 1); apply(pcls, 2); apply(pcls, 3); apply(pcls, 4); apply(pcls, 5);
    }
    if constexpr (OUT == 0) {
        if (epi) sat_commit(p.ab_sat, nsat);
        if (p.ab_dbias || p.ab_cdot) {
            // the workgroup's sums -> one replica of the striped bias-gradient / spectral-norm sums (norm.hip replica_offset);
            // scratch: the idle ring ([256][8] column sums, [4][4] group sums, [4][64] second-level sums)
            float* red = reinterpret_cast<float*>(ring);
            float* redg = red + 256 * 8;
            float* red2 = redg + 4 * 4;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // every wave is done with the ring and the stage
            if (epi) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[te * 8 + e] = sb[e];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float t = wave_sum(sdg[g]);
                    if (lane == 0) redg[(wave - 8) * 4 + g] = t;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (epi) {                     // channel ch = te & 63 = 8 ce + e: its terms sit with the 32 threads te' = ce + 8 k; 4 threads share them
                const int ch = te & 63, part = te >> 6, ce = ch >> 3, e = ch & 7;
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += red[(ce + 8 * (part * 8 + j)) * 8 + e];
                red2[part * 64 + ch] = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int rep = p.ab_nrep > 1 ? (int)(blockIdx.x % (unsigned)p.ab_nrep) * p.ab_rep_stride : 0;
            if (p.ab_dbias && tid < 64) {
                float u = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) u += red2[w * 64 + tid];
                atomicAdd(p.ab_dbias + rep + tid, u);
            }
            if (p.ab_cdot && tid >= 64 && tid < 68) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += redg[w * 4 + (tid - 64)];
                if (t != 0.f) atomicAdd(p.ab_cdot + rep + (tid - 64), t);
            }
        }
    }
#endif
}
// shapes the image-resident form serves (GCSSL_DGRAD_IMG=0: A/B): three samples per workgroup, from one round of the chip up
// (fewer samples: a grid of a third of the CUs loses to the ring forms' 256 tiles)
int dgrad_img_on(const ConvParams& p) {
    static const bool on = [] { const char* e = getenv("GCSSL_DGRAD_IMG"); return !(e && e[0] == '0'); }();
    return on && use_dma() && p.Cin == 64 && p.Cout == 128 && p.Hi == 16 && p.Wi == 16 && p.ldx % 8 == 0 && p.N >= 3 * cu_count();
}
template <typename O, int OUT>
void launch_dgrad_img(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    GCSSL_LAUNCH((dgrad_img_kernel<O, OUT>), dim3((p.N + 2) / 3), dim3(768), 0, st, p);
}

// ---- dgrad + activation backward of the norm-less layer in front (ACTB forms).  0 = not served, 1 = persistent form (with the
// striped bias / spectral-norm sums), 2 = plain tiled form (elementwise part only).
int actb_form(const ConvParams& p, bool need_sums) {
    static const int on = [] { const char* e = getenv("GCSSL_ACTB"); return e ? atoi(e) : 1; }();
    if (!on || !use_dma() || dma_waves() != 8 || p.Cin != 64 || p.Cout < 64 || p.M < 128) return 0;
    if (dgrad_img_on(p) && p.ab_lda % 8 == 0 && p.ldy % 8 == 0) return 3;     // (_ok probes carry ab_lda = ldy = 0)
    const long total = 4L * ((p.M + 127) / 128);
    const long slots = 2L * cu_count();
    if (persist_mode() && p.y_bytes && total > slots + slots / 4) return 1;
    if (!need_sums && total >= tile_threshold()) return 2;
    return 0;
}
template <typename T>
int dispatch_dgrad_actb(ConvParams p, hipStream_t st) {
    if constexpr (Is16<T>::v) {
        typedef typename Op16<T>::type O;
        const int form = actb_form(p, p.ab_dbias || p.ab_cdot);
        p.ksplit = 1;
        const int tm = (p.M + 127) / 128;
        if (form == 1) {
            GCSSL_LAUNCH((conv_dma_persist_kernel<O, 128, 64, 1, 4, 2, false, false, true>), dim3(2 * cu_count()), dim3(512), 0, st, p, tm, 1, 4 * tm);
        } else if (form == 2) {
            GCSSL_LAUNCH((conv_dma_kernel<O, 128, 64, 1, 4, 2, false, 0, 3, true, false, true>), dim3(tm, 1, 4), dim3(512), 0, st, p);
        } else if (form == 3) {
            launch_dgrad_img<O, 0>(p, st);
        } else {
            return GCSSL_EBADSHAPE;
        }
        return gcssl_launch_status();
    }
    return GCSSL_EBADDTYPE;
}

}  // namespace

extern "C" {

// the K split the dispatcher picks for these shapes (1 = none): sizes the slab buffer of the split_stride form
int gcssl_conv4x4s2_fwd_splits(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int act, int out_f32) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    int ks = 1;
    ConvParams p{}; p.act = act; p.out_f32 = out_f32; p.plan_out = &ks; p.ldx = Cin; p.ldy = Cout; p.y_bytes = 1;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    GCSSL_DISPATCH_CONV(dtype, rc = (dispatch_fwd<T, MM>(p, nullptr)));
    return rc ? rc : ks;
}
int gcssl_conv4x4s2_dgrad_splits(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int out_f32) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    int ks = 1;
    ConvParams p{}; p.out_f32 = out_f32; p.plan_out = &ks; p.ldx = Cout; p.ldy = Cin; p.y_bytes = 1;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    GCSSL_DISPATCH_CONV(dtype, rc = (dispatch_dgrad<T, MM>(p, nullptr)));
    return rc ? rc : ks;
}

int gcssl_conv4x4s2_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias,
                        const float* gscale, int group_n, void* y, int ldy, int N, int Hi, int Wi,
                        int Cin, int Cout, int act, int out_f32, long split_stride, void* stream) {
    if (!x || !wf || !y) return GCSSL_ENULL;
    if (split_stride < 0) return GCSSL_EBADSHAPE;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (Cout < 64 || ldx < Cin || ldy < Cout || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    const int kv = gcssl_f32_storage(dtype) ? 4 : 8;
    if (ldx % kv || !aligned16(x) || !aligned16(wf)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = wf; p.y = y; p.bias = bias; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    p.ldx = ldx; p.ldy = ldy; p.act = act; p.out_f32 = out_f32; p.split_stride = split_stride;
    p.sib_remap = sib_remap();
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * Hi * Wi * ldx, (size_t)Cout * 16 * Cin, kv == 4 ? 4 : 2)) return GCSSL_EBADSHAPE;
    {   // output extent for buffer stores: the last pixel's Cout channels end it
        const size_t yb = (((size_t)N * (Hi / 2) * (Wi / 2) - 1) * ldy + Cout) * ((out_f32 || kv == 4) ? 4 : 2);
        p.y_bytes = yb < 0x7FFFFFFFull ? (unsigned)yb : 0u;
    }
    hipStream_t st = (hipStream_t)stream;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    GCSSL_DISPATCH_CONV(dtype, return (dispatch_fwd<T, MM>(p, st)));
    return GCSSL_EBADDTYPE;
}

// ---- split-precision modes: forward conv + InstanceNorm + LeakyReLU in one launch on fp32 tensors (conv_fwd_kernel's FIN form:
// 128 x 64 tiles, 8 waves).  Served when a 128-row tile holds whole samples (H*W/4 <= 64 output pixels), the tiles fill the chip
// without a K split (the statistics need complete sums) and Cin, Cout >= 64.  z (nullable) still receives the pre-norm values.
static bool fin_x3_shape(int N, int Hi, int Wi, int Cin, int Cout) {
    static const bool on = [] { const char* e = getenv("GCSSL_X3_FIN"); return !(e && e[0] == '0'); }();
    // (>= 384 workgroups: D.c4 at n = 768 would run 192 workgroups of 128 K steps each unsplit -- 93 us in the probe, 151 us beside
    //  the generator's chain, against 68 + 14 us for its two-way K split + norm launch: round 4, profiles/round4_x3_*)
    static const long min_wgs = [] { const char* e = getenv("GCSSL_X3_FIN_WGS"); return e ? atol(e) : 384L; }();
    const int HW = (Hi / 2) * (Wi / 2);
    const long M = (long)N * HW;
    return on && HW <= 64 && HW >= 4 && Cin >= 64 && Cout >= 64 && M >= 128 && ((M + 127) / 128) * (Cout / 64) >= min_wgs;
}
int gcssl_conv4x4s2_in_act_x3_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (dtype != GCSSL_F32_F16X3 && dtype != GCSSL_F32_BF16X3) return 0;
    return fin_x3_shape(N, Hi, Wi, Cin, Cout) ? 1 : 0;
}
int gcssl_conv4x4s2_in_act_x3_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias, const float* gscale, int group_n,
                                  float* z, int ldz, float* a, int lda, float* mean, float* rstd, int N, int Hi, int Wi, int Cin,
                                  int Cout, void* stream) {
    if (!x || !wf || !a || !mean || !rstd) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (dtype != GCSSL_F32_F16X3 && dtype != GCSSL_F32_BF16X3) return GCSSL_EBADDTYPE;
    if (!fin_x3_shape(N, Hi, Wi, Cin, Cout) || ldx < Cin || lda < Cout || (z && ldz < Cout) || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    if (ldx % 4 || lda % 4 || (z && ldz % 4) || !aligned16(x) || !aligned16(wf) || !aligned16(a) || (z && !aligned16(z))) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = wf; p.y = a; p.bias = bias; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    p.ldx = ldx; p.ldy = lda; p.act = 1; p.in_mean = mean; p.in_rstd = rstd; p.fin_z = z; p.ld_fin_z = ldz;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * Hi * Wi * ldx, (size_t)Cout * 16 * Cin, 4)) return GCSSL_EBADSHAPE;
    set_mm_scales(p, true);
    p.ksplit = 1;
    dim3 grid((p.M + 127) / 128, (Cout + 63) / 64, 1);
    if (dtype == GCSSL_F32_F16X3) GCSSL_LAUNCH((conv_fwd_kernel<float, 128, 64, 4, 1, 4, 2, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else GCSSL_LAUNCH((conv_fwd_kernel<float, 128, 64, 4, 2, 4, 2, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
    return gcssl_launch_status();
}

// 1 if gcssl_conv4x4s2_in_act_fwd serves these shapes (16-bit dtype, H*W/4 <= 64 output pixels per sample, enough tiles
// without a K split), 0 if the caller should use the conv -> gcssl_in_act_fwd pair, < 0 on bad geometry
int gcssl_conv4x4s2_in_act_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32) return 0;
    ConvParams p{}; p.y_bytes = 1;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    return fin_form(p) ? 1 : 0;
}

int gcssl_conv4x4s2_in_act_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias, const float* gscale,
                               int group_n, void* a, int lda, float* mean, float* rstd, const uint8_t* mask, void* apre,
                               int ld_apre, int apre_n0, int N, int Hi, int Wi, int Cin, int Cout, int act, void* stream) {
    if (!x || !wf || !a || !mean || !rstd) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32) return GCSSL_EBADDTYPE;                        // the fp32 parity mode keeps the fp32 z and the separate norm
    if (act != 1) return GCSSL_EBADSHAPE;                                  // LeakyReLU(0.2) only: the backward inverts it
    if (Cout < 64 || Cin < 64 || ldx < Cin || lda < Cout || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    if (apre && (ld_apre < Cout || ld_apre % 8 || apre_n0 < 0 || apre_n0 > N)) return GCSSL_EBADSHAPE;
    if (ldx % 8 || lda % 8 || !aligned16(x) || !aligned16(wf) || !aligned16(a) || (apre && !aligned16(apre)) ||
        (mask && (((uintptr_t)mask) & 7))) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = wf; p.y = a; p.bias = bias; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    p.ldx = ldx; p.ldy = lda; p.act = act;
    p.in_mean = mean; p.in_rstd = rstd; p.in_mask = mask; p.in_apre = apre; p.ld_apre = ld_apre; p.apre_n0 = apre_n0;
    p.sib_remap = sib_remap();
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * Hi * Wi * ldx, (size_t)Cout * 16 * Cin, 2)) return GCSSL_EBADSHAPE;
    {   // output extent for the buffer stores of the persistent form
        const size_t yb = (((size_t)N * (Hi / 2) * (Wi / 2) - 1) * lda + Cout) * 2;
        p.y_bytes = (yb < 0x7FFFFFFFull && (size_t)N * Cout * 4 < 0x7FFFFFFFull) ? (unsigned)yb : 0u;
    }
    if (!fin_form(p)) return GCSSL_EBADSHAPE;
    GCSSL_DISPATCH(dtype, return dispatch_fwd_in<T>(p, (hipStream_t)stream));
    return GCSSL_EBADDTYPE;
}

int gcssl_conv4x4s2_dgrad(int dtype, const void* dy, int lddy, const void* wt, const float* gscale,
                          int group_n, void* dx, int lddx, int N, int Hi, int Wi, int Cin, int Cout,
                          int out_f32, long split_stride, void* stream) {
    if (!dy || !wt || !dx) return GCSSL_ENULL;
    if (split_stride < 0) return GCSSL_EBADSHAPE;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (Cout < 8 || lddy < Cout || lddx < Cin || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    const int kv = gcssl_f32_storage(dtype) ? 4 : 8;
    if (lddy % kv || !aligned16(dy) || !aligned16(wt)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = dy; p.w = wt; p.y = dx; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    static const int class_major = [] { const char* e = getenv("GCSSL_DGRAD_ORDER"); return (e && e[0] == '1') ? 1 : 0; }();
    p.class_major = class_major;
    p.sib_remap = class_major ? 0 : sib_remap();
    p.ldx = lddy; p.ldy = lddx; p.out_f32 = out_f32; p.split_stride = split_stride;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * (Hi / 2) * (Wi / 2) * lddy, (size_t)Cin * 16 * Cout, kv == 4 ? 4 : 2)) return GCSSL_EBADSHAPE;
    {
        const size_t yb = (((size_t)N * Hi * Wi - 1) * lddx + Cin) * ((out_f32 || kv == 4) ? 4 : 2);
        p.y_bytes = yb < 0x7FFFFFFFull ? (unsigned)yb : 0u;
    }
    hipStream_t st = (hipStream_t)stream;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    GCSSL_DISPATCH_CONV(dtype, return (dispatch_dgrad<T, MM>(p, st)));
    return GCSSL_EBADDTYPE;
}

// the split-precision modes' form of gcssl_conv4x4s2_dgrad_act_bwd: conv_dgrad_kernel's own epilogue, any shape its 64-channel-wide
// tiles serve without a K split (GCSSL_X3_ACTB=0: off, A/B)
static bool actb_x3_shape(int N, int Hi, int Wi, int Cin, int Cout) {
    static const bool on = [] { const char* e = getenv("GCSSL_X3_ACTB"); return !(e && e[0] == '0'); }();
    return on && Cin >= 64 && Cout >= 64 && (long)N * (Hi / 2) * (Wi / 2) >= 128;
}
// 1 if gcssl_conv4x4s2_dgrad_act_bwd serves these shapes (with_sums: dbias / cdot requested), else 0
int gcssl_conv4x4s2_dgrad_act_bwd_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int with_sums) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32) return 0;
    if (gcssl_f32_storage(dtype)) return actb_x3_shape(N, Hi, Wi, Cin, Cout) ? 1 : 0;       // split-precision modes (fp32 tensors)
    ConvParams p{}; p.y_bytes = 1;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    return actb_form(p, with_sums != 0) ? 1 : 0;
}

int gcssl_conv4x4s2_dgrad_act_bwd(int dtype, const void* dy, int lddy, const void* wt, const void* a, int lda, const float* gscale,
                                  int group_n, const float* bias, void* dzs, int lddz, float* dbias, float* cdot, int nrep,
                                  int rep_stride, unsigned* sat, int N, int Hi, int Wi, int Cin, int Cout, void* stream) {
    if (!dy || !wt || !a || !dzs) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32) return GCSSL_EBADDTYPE;
    const bool x3 = gcssl_f32_storage(dtype);                            // split-precision modes: every tensor here is fp32
    if (lddy < Cout || lda < Cin || lddz < Cin || ((gscale || cdot) && group_n <= 0)) return GCSSL_EBADSHAPE;
    if (nrep < 1 || (nrep > 1 && rep_stride < Cin)) return GCSSL_EBADSHAPE;
    if ((gscale || cdot) && (group_n * (Hi / 2) * (Wi / 2)) % 128) return GCSSL_EBADSHAPE;      // a 128-row tile must not straddle groups
    const int kv = x3 ? 4 : 8;
    if (lddy % kv || lda % kv || lddz % kv || !aligned16(dy) || !aligned16(wt) || !aligned16(a) || !aligned16(dzs)) return GCSSL_EALIGN;
    if (x3) {
        if (!actb_x3_shape(N, Hi, Wi, Cin, Cout)) return GCSSL_EBADSHAPE;
        ConvParams q{}; q.x = dy; q.w = wt; q.y = dzs; q.gscale = gscale; q.group_n = group_n; q.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
        q.ldx = lddy; q.ldy = lddz;
        q.ab_a = a; q.ab_lda = lda; q.ab_bias = bias; q.ab_dbias = dbias; q.ab_cdot = cdot; q.ab_nrep = nrep; q.ab_rep_stride = rep_stride;
        fill_geom(q, N, Hi, Wi, Cin, Cout);
        if (!fill_bytes(q, (size_t)N * (Hi / 2) * (Wi / 2) * lddy, (size_t)Cin * 16 * Cout, 4)) return GCSSL_EBADSHAPE;
        q.y_bytes = 1;
        GCSSL_DISPATCH_CONV(dtype, return (dispatch_dgrad<T, MM>(q, (hipStream_t)stream)));
        return GCSSL_EBADDTYPE;
    }
    ConvParams p{}; p.x = dy; p.w = wt; p.y = dzs; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    p.ldx = lddy; p.ldy = lddz;
    p.ab_a = a; p.ab_lda = lda; p.ab_bias = bias; p.ab_dbias = dbias; p.ab_cdot = cdot; p.ab_nrep = nrep; p.ab_rep_stride = rep_stride;
    p.ab_sat = sat;
    p.sib_remap = sib_remap();
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * (Hi / 2) * (Wi / 2) * lddy, (size_t)Cin * 16 * Cout, 2)) return GCSSL_EBADSHAPE;
    {
        const size_t yb = (((size_t)N * Hi * Wi - 1) * lddz + Cin) * 2, ab = (((size_t)N * Hi * Wi - 1) * lda + Cin) * 2;
        if (yb >= 0x7FFFFFFFull || ab >= 0x7FFFFFFFull) return GCSSL_EBADSHAPE;
        p.y_bytes = (unsigned)yb; p.ab_bytes = (unsigned)ab;
    }
    if (!actb_form(p, dbias || cdot)) return GCSSL_EBADSHAPE;
    GCSSL_DISPATCH(dtype, return dispatch_dgrad_actb<T>(p, (hipStream_t)stream));
    return GCSSL_EBADDTYPE;
}

// The 8-channel first layer's conv as the FORWARD of the reverse (double-backward) gradient-penalty chain, with the layer's
// LeakyReLU backward and the <dotx, conv output> spectral-norm term in the epilogue: y = lrelu'(a) * (gscale * conv(x)) in the
// compute dtype.  The fp32 conv output never goes to memory.  Served by conv_fwd_c8_kernel's shapes only (Cin 8 -> Cout 64).
static bool fwd_actb_shape(int N, int Hi, int Wi, int Cin, int Cout) {
    static const bool on = [] { const char* e = getenv("GCSSL_FWD_ACTB"); return !(e && e[0] == '0'); }();
    const int Wo = Wi / 2;
    return on && Cin == 8 && Cout == 64 && (Wo == 16 || Wo == 32 || Wo == 64) && (Hi / 2) % (128 / Wo) == 0 && c8_fwd_on();
}
int gcssl_conv4x4s2_fwd_act_bwd_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout) {
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32_F16X3 || dtype == GCSSL_F32_BF16X3) return fwd_actb_shape(N, Hi, Wi, Cin, Cout) && c8_x3_on() ? 1 : 0;   // (fp32 tensors)
    return dtype != GCSSL_F32 && fwd_actb_shape(N, Hi, Wi, Cin, Cout) ? 1 : 0;
}

int gcssl_conv4x4s2_fwd_act_bwd(int dtype, const void* x, int ldx, const void* wf, const float* gscale, int group_n, const void* a,
                                int lda, void* y, int ldy, const void* dotx, int lddot, float* dot_out, unsigned* sat, int N, int Hi,
                                int Wi, int Cin, int Cout, void* stream) {
    if (!x || !wf || !a || !y) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (dtype == GCSSL_F32) return GCSSL_EBADDTYPE;
    const bool x3 = gcssl_f32_storage(dtype);                            // split-precision modes: every tensor here is fp32
    if (!fwd_actb_shape(N, Hi, Wi, Cin, Cout) || (x3 && !c8_x3_on())) return GCSSL_EBADSHAPE;
    if ((dotx != nullptr) != (dot_out != nullptr)) return GCSSL_EBADSHAPE;
    if (ldx < Cin || lda < Cout || ldy < Cout || (dotx && lddot < Cout) || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    const int kv = x3 ? 4 : 8;
    if (ldx % kv || lda % kv || ldy % kv || (dotx && lddot % kv) || !aligned16(x) || !aligned16(wf) || !aligned16(a) || !aligned16(y) ||
        (dotx && !aligned16(dotx))) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = wf; p.y = y; p.gscale = gscale; p.group_n = group_n; p.inv_group_n = group_n > 0 ? 1.0f / (float)group_n : 0.f;
    p.ldx = ldx; p.ldy = ldy;
    p.ab_a = a; p.ab_lda = lda; p.ab_dotx = dotx; p.ab_lddot = lddot; p.ab_dot_out = dot_out; p.ab_sat = sat;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * Hi * Wi * ldx, (size_t)Cout * 16 * Cin, x3 ? 4 : 2)) return GCSSL_EBADSHAPE;
    p.y_bytes = 1;
    GCSSL_DISPATCH_CONV(dtype, return (dispatch_fwd<T, MM>(p, (hipStream_t)stream)));
    return GCSSL_EBADDTYPE;
}

// number of split-K slabs gcssl_conv4x4s2_wgrad will write for this geometry (caller sizes the workspace)
// The filter-row LDS-DMA wgrad kernel serves the layers that give 256 workgroups >= 8 K steps each; with less work per
// workgroup its 128-KB slab tile costs more than the halved fills save (G.down2..4, G.up1 at batch 256: 17.6 vs 16.3 us).
static bool wgrad_dma_shape(int N, int Hi, int Wi, int Cin, int Cout) {
    static const int on = [] { const char* e = getenv("GCSSL_WGRAD_DMA"); return e ? atoi(e) : 1; }();   // A/B knob: 0 off, 2 always
    if (!on || !use_dma() || Cin % 64 || Cout % 128) return false;
    const long nkt = ((long)N * (Hi / 2) * (Wi / 2) + 63) / 64, tiles = (long)(Cout / 128) * 4 * (Cin / 64);
    return on == 2 || tiles * (nkt / 16) >= 128;
}

int gcssl_conv4x4s2_wgrad_splits(int N, int Hi, int Wi, int Cin, int Cout) {
    if (check_geom(N, Hi, Wi, Cin, Cout)) return GCSSL_EBADSHAPE;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : (Cin >= 64 ? 64 : (Cin == 8 ? 8 : 0));
    if (!bn) return GCSSL_EBADSHAPE;
    // the LDS-DMA kernel gives a workgroup a whole filter row of a 128 x 64 block and holds one workgroup per CU (144 KB of
    // LDS); the count depends on the shape only, so that one slab size serves every element type
    const bool row4 = wgrad_dma_shape(N, Hi, Wi, Cin, Cout);
    const long tiles = row4 ? (long)(Cout / 128) * 4 * (Cin / 64) : (long)(Cout / bm) * 16 * (Cin / bn) / (Cin == 8 ? 16 : 1);
    const int nkt = (N * (Hi / 2) * (Wi / 2) + 63) / 64;       // K granules of 64 output pixels (dtype independent)
    static const long target_e = [] { const char* e = getenv("GCSSL_WGRAD_WGS"); return e ? atol(e) : 0L; }();
    const long target = target_e ? target_e : (row4 ? 256L : 512L);
    long want = (target + tiles - 1) / tiles;              // ~2 workgroups per CU (1 for the filter-row kernel)
    // padded first layers have a single tile, so only K splits fill the chip: 128 / 256 / 512 splits of D.c1.wgrad (1024
    // samples) take 25.8 / 16.0 / 13.9 us -- a workgroup's time is its K steps -- and the 32-KB slabs stay cheap to reduce
    static const long cap = [] { const char* e = getenv("GCSSL_WGRAD_CAP"); return e ? atol(e) : 512L; }();
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > nkt) want = nkt;
    // ... but never fewer than `mink` K granules (64 pixels each) per split: below that a workgroup writes its fp32 slab
    // tile (64 KB) for a handful of MFMAs and the slabs, not the contraction, set the time
    static const int mink = [] { const char* e = getenv("GCSSL_WGRAD_MINK"); return e ? atoi(e) : 8; }();   // 1/8/16/32: 78.0/78.4/78.0/76.7k img/s
    int per = (nkt + (int)want - 1) / (int)want;
    if (per < mink) per = mink < nkt ? mink : nkt;
    return (nkt + per - 1) / per;
}

int gcssl_conv4x4s2_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* slab,
                          int N, int Hi, int Wi, int Cin, int Cout, void* stream) {
    if (!x || !dy || !slab) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if ((Cin < 64 && Cin != 8) || Cout < 64 || ldx < Cin || lddy < Cout) return GCSSL_EBADSHAPE;
    const int kv = gcssl_f32_storage(dtype) ? 4 : 8;
    if (ldx % kv || lddy % kv || !aligned16(x) || !aligned16(dy)) return GCSSL_EALIGN;
    const int nsplit = gcssl_conv4x4s2_wgrad_splits(N, Hi, Wi, Cin, Cout);
    if (nsplit <= 0) return GCSSL_EBADSHAPE;
    ConvParams p{}; p.x = x; p.w = dy; p.y = slab; p.ldx = ldx; p.ldw = lddy;
    p.xcd_remap = wgrad_xcd();
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    if (!fill_bytes(p, (size_t)N * Hi * Wi * ldx, (size_t)N * (Hi / 2) * (Wi / 2) * lddy, kv == 4 ? 4 : 2)) return GCSSL_EBADSHAPE;
    const int nkt = (p.M + 63) / 64;
    p.ktiles_per_split = ((nkt + nsplit - 1) / nsplit) * (64 / (gcssl_f32_storage(dtype) ? BKOf<float>::v : BKOf<bf16_t>::v));
    hipStream_t st = (hipStream_t)stream;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : 64;
    const bool smallc = Cin == 8;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    set_mm_scales(p, false);
    if (!gcssl_f32_storage(dtype) && wgrad_dma_shape(N, Hi, Wi, Cin, Cout)) {
        dim3 gd(Cout / 128, 4 * (Cin / 64), nsplit);                    // LDS-DMA ring, one filter row (4 taps) per workgroup
        static const int lw = [] { const char* e = getenv("GCSSL_WGRAD_LW"); return e ? atoi(e) : 8; }();    // A/B knob: 0 = all waves fetch
        if (lw) {
            if (dtype == GCSSL_F16) GCSSL_LAUNCH((conv_wgrad_dma_kernel<f16_t, 8>), gd, dim3(1024), 0, st, p);
            else GCSSL_LAUNCH((conv_wgrad_dma_kernel<bf16_t, 8>), gd, dim3(1024), 0, st, p);
            return gcssl_launch_status();
        }
        if (dtype == GCSSL_F16) GCSSL_LAUNCH((conv_wgrad_dma_kernel<f16_t, 0>), gd, dim3(512), 0, st, p);
        else GCSSL_LAUNCH((conv_wgrad_dma_kernel<bf16_t, 0>), gd, dim3(512), 0, st, p);
        return gcssl_launch_status();
    }
    dim3 grid(Cout / bm, smallc ? 1 : 16 * (Cin / bn), nsplit);
#define WG(T, A, B, S) GCSSL_LAUNCH((conv_wgrad_kernel<T, A, B, S, 4, MM>), grid, dim3(NT), 0, st, p)
#define WG8(T, A, B) GCSSL_LAUNCH((conv_wgrad_kernel<T, A, B, false, 4, MM, 4, 2>), grid, dim3(512), 0, st, p)
    static const int w8 = [] { const char* e = getenv("GCSSL_X3_WGRAD_WAVES"); return e ? atoi(e) : 8; }();   // split-precision forms: 8 waves (A/B: 4)
    GCSSL_DISPATCH_CONV(dtype,
        if (smallc) { if (bm == 128) WG(T, 128, 128, true); else WG(T, 64, 128, true); }
        else if (MM != 0 && w8 == 16 && bm == 128 && bn == 128) { if constexpr (MM != 0) WG8(T, 128, 128); }   // (A/B only: spills, 68.7 -> 75.8 us)
        else if (MM != 0 && w8 >= 8 && bm == 128 && bn == 64) { if constexpr (MM != 0) WG8(T, 128, 64); }       // 85.9 -> 81.9 us (D.c2 / G.up4 shapes)
        else if (bm == 128 && bn == 128) WG(T, 128, 128, false); else if (bm == 128) WG(T, 128, 64, false);
        else if (bn == 128) WG(T, 64, 128, false); else WG(T, 64, 64, false));
#undef WG
#undef WG8
    return gcssl_launch_status();
}

// The weight gradients of up to three layers in ONE launch (they do not depend on each other): served when every layer takes the
// filter-row LDS-DMA kernel (16-bit dtype, wgrad_dma_shape) -- then the workgroups of all layers are one grid; otherwise the
// layers are launched one by one exactly as gcssl_conv4x4s2_wgrad would.  Arguments per layer as gcssl_conv4x4s2_wgrad's.
int gcssl_conv4x4s2_wgrad_batch(int dtype, int nl, const void* const* x, const int* ldx, const void* const* dy, const int* lddy,
                                float* const* slab, const int* N, const int* Hi, const int* Wi, const int* Cin, const int* Cout,
                                void* stream) {
    if (!x || !ldx || !dy || !lddy || !slab || !N || !Hi || !Wi || !Cin || !Cout) return GCSSL_ENULL;
    if (nl < 1 || nl > 3) return GCSSL_EBADSHAPE;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    static const bool on = [] { const char* e = getenv("GCSSL_WGRAD_BATCH"); return !(e && e[0] == '0'); }();
    bool all_dma = on && !gcssl_f32_storage(dtype) && nl > 1;
    for (int i = 0; i < nl && all_dma; ++i)
        all_dma = x[i] && dy[i] && slab[i] && check_geom(N[i], Hi[i], Wi[i], Cin[i], Cout[i]) == 0 && wgrad_dma_shape(N[i], Hi[i], Wi[i], Cin[i], Cout[i]) &&
                  ldx[i] >= Cin[i] && lddy[i] >= Cout[i] && ldx[i] % 8 == 0 && lddy[i] % 8 == 0 && aligned16(x[i]) && aligned16(dy[i]);
    static const int lw = [] { const char* e = getenv("GCSSL_WGRAD_LW"); return e ? atoi(e) : 8; }();
    if (!all_dma || !lw) {
        for (int i = 0; i < nl; ++i) {
            const int rc = gcssl_conv4x4s2_wgrad(dtype, x[i], ldx[i], dy[i], lddy[i], slab[i], N[i], Hi[i], Wi[i], Cin[i], Cout[i], stream);
            if (rc) return rc;
        }
        return GCSSL_OK;
    }
    WgradBatch b{};
    int total = 0;
    for (int i = 0; i < nl; ++i) {
        const int nsplit = gcssl_conv4x4s2_wgrad_splits(N[i], Hi[i], Wi[i], Cin[i], Cout[i]);
        if (nsplit <= 0) return GCSSL_EBADSHAPE;
        ConvParams& p = b.p[i];
        p.x = x[i]; p.w = dy[i]; p.y = slab[i]; p.ldx = ldx[i]; p.ldw = lddy[i];
        p.xcd_remap = wgrad_xcd();
        fill_geom(p, N[i], Hi[i], Wi[i], Cin[i], Cout[i]);
        if (!fill_bytes(p, (size_t)N[i] * Hi[i] * Wi[i] * ldx[i], (size_t)N[i] * (Hi[i] / 2) * (Wi[i] / 2) * lddy[i], 2)) return GCSSL_EBADSHAPE;
        const int nkt = (p.M + 63) / 64;
        p.ktiles_per_split = ((nkt + nsplit - 1) / nsplit) * (64 / BKOf<bf16_t>::v);
        set_mm_scales(p, false);
        b.gx[i] = Cout[i] / 128; b.gy[i] = 4 * (Cin[i] / 64); b.gz[i] = nsplit;
        b.first[i] = total; total += b.gx[i] * b.gy[i] * b.gz[i];
    }
    b.first[nl] = total; b.n = nl;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GCSSL_F16) GCSSL_LAUNCH((conv_wgrad_dma_batch_kernel<f16_t, 8>), dim3((unsigned)total), dim3(1024), 0, st, b);
    else GCSSL_LAUNCH((conv_wgrad_dma_batch_kernel<bf16_t, 8>), dim3((unsigned)total), dim3(1024), 0, st, b);
    return gcssl_launch_status();
}

int gcssl_wgrad_reduce(const float* slab, int nsplit, float* dw, int Cout, int Cin, int Cin_real,
                       const float* coef, const float* cscale, const float* u, int ustride, const float* v,
                       int vstride, int nrank, int accumulate, void* stream) {
    if (!slab || !dw || (nrank > 0 && (!coef || !u || !v))) return GCSSL_ENULL;
    if (nsplit <= 0 || Cout <= 0 || Cin <= 0 || Cin_real <= 0 || Cin_real > Cin) return GCSSL_EBADSHAPE;
    if (nrank > 4 || (nrank > 0 && (ustride < Cout || vstride < Cin_real * 16))) return GCSSL_EBADSHAPE;
    if (nrank > 0 && (vstride % 4 || !aligned16(v))) return GCSSL_EALIGN;                 // 16-byte loads of the v rows
    int zg = 1;
    if (accumulate == 2) { zg = (nsplit + 15) / 16; if (zg > 16) zg = 16; }
    GCSSL_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((Cin + 63) / 64), (unsigned)Cout, (unsigned)zg), dim3(256), 0,
                       (hipStream_t)stream, slab, nsplit, dw, Cout, Cin, Cin_real, coef, cscale, u, ustride, v, vstride, nrank, accumulate);
    return gcssl_launch_status();
}

int gcssl_wgrad_reduce_batch(int nl, const float* const* slab, const int* nsplit, float* const* dw, const int* Cout,
                             const int* Cin, const int* Cin_real, const float* const* coef, const float* const* u,
                             const float* const* v, int ustride, int vstride, int nrank, int accumulate,
                             const float* const* coef_rep, const float* const* bias_rep, float* const* dbias, int nrep,
                             int rep_stride, void* stream) {
    if (!slab || !nsplit || !dw || !Cout || !Cin || !Cin_real) return GCSSL_ENULL;
    if (nl < 1 || nl > 8 || nrank < 0 || nrank > 4 || accumulate < 0 || accumulate > 2) return GCSSL_EBADSHAPE;
    if (nrank > 0 && (!coef || !u || !v)) return GCSSL_ENULL;
    if ((coef_rep || bias_rep) && (nrep < 1 || rep_stride < 1)) return GCSSL_EBADSHAPE;
    if (coef_rep && !nrank) return GCSSL_EBADSHAPE;
    if (bias_rep && !dbias) return GCSSL_ENULL;
    RedBatch b{};
    int blk = 0;
    for (int i = 0; i < nl; ++i) {
        if (!slab[i] || !dw[i] || (nrank > 0 && (!coef[i] || !u[i] || !v[i]))) return GCSSL_ENULL;
        if (nsplit[i] <= 0 || Cout[i] <= 0 || Cin[i] <= 0 || Cin_real[i] <= 0 || Cin_real[i] > Cin[i]) return GCSSL_EBADSHAPE;
        if (nrank > 0 && (ustride < Cout[i] || vstride < Cin_real[i] * 16)) return GCSSL_EBADSHAPE;
        if (nrank > 0 && (vstride % 4 || !aligned16(v[i]))) return GCSSL_EALIGN;      // 16-byte loads of the v rows
        int zg = 1;
        if (accumulate == 2) { zg = (nsplit[i] + 15) / 16; if (zg > 16) zg = 16; }
        b.l[i] = RedLayer{slab[i], dw[i], nrank ? coef[i] : nullptr, nrank ? u[i] : nullptr, nrank ? v[i] : nullptr,
                          nsplit[i], Cout[i], Cin[i], Cin_real[i], nrank, zg, blk,
                          coef_rep ? coef_rep[i] : nullptr, bias_rep ? bias_rep[i] : nullptr, (bias_rep && bias_rep[i]) ? dbias[i] : nullptr};
        if (bias_rep && bias_rep[i] && !dbias[i]) return GCSSL_ENULL;
        blk += ((Cin[i] + 63) / 64) * Cout[i] * zg;
    }
    b.nl = nl; b.ustride = ustride; b.vstride = vstride; b.accumulate = accumulate; b.nrep = nrep; b.rep_stride = rep_stride;
    b.bias_blk0 = -1;
    if (bias_rep) {
        int nb = 0;
        for (int i = 0; i < nl; ++i) if (bias_rep[i]) nb += Cout[i];
        if (nb > 0) { b.bias_blk0 = blk; blk += (nb + 255) / 256; }
    }
    GCSSL_LAUNCH(wgrad_reduce_batch_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, b);
    return gcssl_launch_status();
}

int gcssl_prep_conv_weights(int dtype, int nl, const float* const* w, void* const* wf, void* const* wt, const int* Cout,
                            const int* Cin, const int* CinP, const float* w5, float* w5p, int C5, void* stream) {
    if (!w || !wf || !wt || !Cout || !Cin || !CinP) return GCSSL_ENULL;
    if (nl < 1 || nl > 8) return GCSSL_EBADSHAPE;
    if ((w5 != nullptr) != (w5p != nullptr) || (w5 && C5 <= 0)) return GCSSL_EBADSHAPE;
    PrepBatch b{};
    b.nl = nl; b.w5 = w5; b.w5p = w5p; b.C5 = C5;
    b.scale = (dtype == GCSSL_F32_F16X3 || dtype == GCSSL_F32_BF16X3) ? x3_wscale() : 1.f;       // (the head conv's w5p is never scaled)
    size_t mx = 0;
    for (int i = 0; i < nl; ++i) {
        if (!w[i] || (!wf[i] && !wt[i])) return GCSSL_ENULL;
        if (Cout[i] <= 0 || Cin[i] <= 0 || CinP[i] < Cin[i]) return GCSSL_EBADSHAPE;
        b.l[i] = PrepLayer{w[i], wf[i], wt[i], Cout[i], Cin[i], CinP[i]};
        const size_t t = (size_t)Cout[i] * 16 * CinP[i];
        if (t > mx) mx = t;
    }
    unsigned gx = (unsigned)((mx + 4095) / 4096); if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;     // one 4096-element tile per pass
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    SnBatch sn{};
    int sn_row = -1;                                         // a spectral-norm closing step left pending for this launch?
    int rows = nl + (w5 ? 1 : 0);
    if (gcssl_take_pending_sn(&sn)) { sn_row = rows++; if (gx < (unsigned)sn.nl) gx = (unsigned)sn.nl; }
    C5DgradRider c5r{};
    int c5_row = -1;                                         // ... a deferred head-conv data gradient?
    if (gcssl_take_pending_c5(&c5r)) c5_row = rows++;
    dim3 grid(gx, rows, 2);
    GCSSL_DISPATCH_CONV(dtype, GCSSL_LAUNCH(prep_weight_batch_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, b, sn, sn_row, c5r, c5_row));
    return gcssl_launch_status();
}

int gcssl_prep_conv_weight(int dtype, const float* w, void* wf, void* wt, int Cout, int Cin, int CinP,
                           void* stream) {
    if (!w || (!wf && !wt)) return GCSSL_ENULL;
    if (Cout <= 0 || Cin <= 0 || CinP < Cin) return GCSSL_EBADSHAPE;
    const size_t total = (size_t)Cout * 16 * CinP;
    dim3 grid((unsigned)((total + 255) / 256));
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    const float wscale = (dtype == GCSSL_F32_F16X3 || dtype == GCSSL_F32_BF16X3) ? x3_wscale() : 1.f;
    GCSSL_DISPATCH_CONV(dtype, GCSSL_LAUNCH(prep_weight_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, w, (T*)wf, (T*)wt, Cout, Cin, CinP, wscale));
    return gcssl_launch_status();
}

// ---- 3x3 stride-1 pad-1 forms (GeneratorSimpleRegressor, cgan/models.py:163-193) ----
static int wk3(int c) { return (9 * c + 63) / 64 * 64; }

int gcssl_conv3x3_wk(int C) { return C > 0 ? wk3(C) : GCSSL_EBADSHAPE; }

int gcssl_conv3x3_fwd(int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                      int N, int H, int W, int Cin, int Cout, int out_f32, void* stream) {
    if (!x || !w || !y) return GCSSL_ENULL;
    int rc = check_geom(N, H, W, Cin, Cout);
    if (rc) return rc;
    if (Cout < 64 || ldx < Cin || ldy < Cout) return GCSSL_EBADSHAPE;
    const int kv = gcssl_f32_storage(dtype) ? 4 : 8;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (ldx % kv || !aligned16(x) || !aligned16(w)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = w; p.y = y; p.bias = bias; p.ldx = ldx; p.ldy = ldy; p.out_f32 = out_f32;
    p.N = N; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Cout = Cout; p.wk = wk3(Cin);
    static const int kcap = [] { const char* e = getenv("GCSSL_KCAP"); return e ? atoi(e) : 0; }();
    p.kcap = kcap;
    p.lgWo = ilog2(W); p.lgHoWo = ilog2(H * W); p.lgCin = ilog2(Cin); p.lgCout = ilog2(Cout);
    p.M = N * H * W;
    if (!fill_bytes(p, (size_t)N * H * W * ldx, (size_t)Cout * p.wk, kv == 4 ? 4 : 2)) return GCSSL_EBADSHAPE;
    {   // output extent for the buffer stores of the persistent kernel
        const size_t yb = (((size_t)N * H * W - 1) * ldy + Cout) * ((out_f32 || kv == 4) ? 4 : 2);
        p.y_bytes = yb < 0x7FFFFFFFull ? (unsigned)yb : 0u;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool big = (long)((p.M + 127) / 128) * (Cout / 64) >= 256;
    dim3 grid((p.M + (big ? 127 : 63)) / (big ? 128 : 64), Cout / 64, 1);
    static const bool dma3 = [] { const char* e = getenv("GCSSL_CONV3_DMA"); return !(e && e[0] == '0'); }();   // A/B knob
    if (!gcssl_f32_storage(dtype) && use_dma() && dma3) {             // LDS-DMA pipeline (MODE 2 of conv_dma_kernel)
#define DMA3(T) do { \
        if (Cin < 64) {                                        /* 8-channel first layer: a K tile spans several taps */ \
            if (big) GCSSL_LAUNCH((conv_dma_kernel<T, 128, 64, 2, 2, 2, true>), grid, dim3(256), 0, st, p); \
            else GCSSL_LAUNCH((conv_dma_kernel<T, 64, 64, 2, 2, 2, true>), grid, dim3(256), 0, st, p); \
        } else if (big) { \
            /* more tiles than resident workgroups: the persistent form (the DMA ring never drains between tiles) */ \
            const int slots = cu_count() * ((160 * 1024) / (3 * (128 + 64) * 128)), total = (int)(grid.x * grid.y); \
            if ((persist_mode() & 1) && p.y_bytes && total > slots + slots / 4) \
                GCSSL_LAUNCH((conv_dma_persist_kernel<T, 128, 64, 2, 4, 2>), dim3(slots), dim3(512), 0, st, p, (int)grid.x, (int)grid.y, total); \
            else GCSSL_LAUNCH((conv_dma_kernel<T, 128, 64, 2, 4, 2, false>), grid, dim3(512), 0, st, p); \
        } \
        else GCSSL_LAUNCH((conv_dma_kernel<T, 64, 64, 2, 2, 2, false>), grid, dim3(256), 0, st, p); } while (0)
        if (dtype == GCSSL_F16) DMA3(f16_t); else DMA3(bf16_t);
#undef DMA3
        return gcssl_launch_status();
    }
    set_mm_scales(p, true);
    GCSSL_DISPATCH_CONV(dtype, if (big) GCSSL_LAUNCH((conv_fwd_kernel<T, 128, 64, 3, MM>), grid, dim3(NT), 0, st, p);
                               else GCSSL_LAUNCH((conv_fwd_kernel<T, 64, 64, 3, MM>), grid, dim3(NT), 0, st, p));
    return gcssl_launch_status();
}

int gcssl_conv3x3_wgrad_splits(int N, int H, int W, int Cin, int Cout) {
    if (check_geom(N, H, W, Cin, Cout)) return GCSSL_EBADSHAPE;
    if ((Cin < 64 && Cin != 8) || Cout < 64) return GCSSL_EBADSHAPE;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : (Cin >= 64 ? 64 : 8);
    const long tiles = Cin == 8 ? Cout / bm : (long)(Cout / bm) * 9 * (Cin / bn);
    const int nkt = (N * H * W + 63) / 64;
    static const long target = [] { const char* e = getenv("GCSSL_WGRAD3_WGS"); return e ? atol(e) : 512L; }();
    long want = (target + tiles - 1) / tiles;
    if (want > 128) want = 128;
    if (want > nkt) want = nkt;
    int per = (nkt + (int)want - 1) / (int)want;
    if (per < 8) per = 8 < nkt ? 8 : nkt;
    return (nkt + per - 1) / per;
}

// slab: [splits][Cout][16][Cin] fp32 (the 16-tap layout of the 4x4 form; taps 9-15 are not written when Cin >= 64)
int gcssl_conv3x3_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* slab, int N, int H, int W,
                        int Cin, int Cout, void* stream) {
    if (!x || !dy || !slab) return GCSSL_ENULL;
    const int nsplit = gcssl_conv3x3_wgrad_splits(N, H, W, Cin, Cout);
    if (nsplit <= 0) return GCSSL_EBADSHAPE;
    if (ldx < Cin || lddy < Cout) return GCSSL_EBADSHAPE;
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    const int kv = gcssl_f32_storage(dtype) ? 4 : 8;
    if (ldx % kv || lddy % kv || !aligned16(x) || !aligned16(dy)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = dy; p.y = slab; p.ldx = ldx; p.ldw = lddy;
    p.xcd_remap = wgrad_xcd();
    p.N = N; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Cout = Cout;
    p.lgWo = ilog2(W); p.lgHoWo = ilog2(H * W); p.lgCin = ilog2(Cin); p.lgCout = ilog2(Cout);
    p.M = N * H * W;
    if (!fill_bytes(p, (size_t)N * H * W * ldx, (size_t)N * H * W * lddy, kv == 4 ? 4 : 2)) return GCSSL_EBADSHAPE;
    const int nkt = (p.M + 63) / 64;
    p.ktiles_per_split = ((nkt + nsplit - 1) / nsplit) * (64 / (gcssl_f32_storage(dtype) ? BKOf<float>::v : BKOf<bf16_t>::v));
    set_mm_scales(p, false);
    hipStream_t st = (hipStream_t)stream;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : 64;
    const bool smallc = Cin == 8;
    dim3 grid(Cout / bm, smallc ? 1 : 9 * (Cin / bn), nsplit);
#define WG(T, A, B, S) GCSSL_LAUNCH((conv_wgrad_kernel<T, A, B, S, 3, MM>), grid, dim3(NT), 0, st, p)
    GCSSL_DISPATCH_CONV(dtype,
        if (smallc) { if (bm == 128) WG(T, 128, 128, true); else WG(T, 64, 128, true); }
        else if (bm == 128 && bn == 128) WG(T, 128, 128, false); else if (bm == 128) WG(T, 128, 64, false);
        else if (bn == 128) WG(T, 64, 128, false); else WG(T, 64, 64, false));
#undef WG
    return gcssl_launch_status();
}

int gcssl_conv3x3_wgrad_reduce(int nl, const float* const* slab, const int* nsplit, float* const* dw, const int* Cout,
                               const int* Cin, const int* Cin_real, void* stream) {
    if (!slab || !nsplit || !dw || !Cout || !Cin || !Cin_real) return GCSSL_ENULL;
    if (nl < 1 || nl > 8) return GCSSL_EBADSHAPE;
    Red3Batch b{};
    int blk = 0;
    for (int i = 0; i < nl; ++i) {
        if (!slab[i] || !dw[i]) return GCSSL_ENULL;
        if (nsplit[i] <= 0 || Cout[i] <= 0 || Cin[i] <= 0 || Cin_real[i] <= 0 || Cin_real[i] > Cin[i]) return GCSSL_EBADSHAPE;
        b.l[i] = Red3Layer{slab[i], dw[i], nsplit[i], Cout[i], Cin[i], Cin_real[i], blk};
        blk += ((Cin[i] + 63) / 64) * Cout[i];
    }
    b.nl = nl;
    GCSSL_LAUNCH(wgrad3_reduce_batch_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, b);
    return gcssl_launch_status();
}

// w[i]: fp32 [Cout][Cin][3][3]; wf[i] (nullable): [Cout][wk(CinP)]; wt[i] (nullable): [Cin][wk(Cout)] in the compute dtype
int gcssl_conv3x3_prep_weights(int dtype, int nl, const float* const* w, void* const* wf, void* const* wt, const int* Cout,
                               const int* Cin, const int* CinP, void* stream) {
    if (!w || !wf || !wt || !Cout || !Cin || !CinP) return GCSSL_ENULL;
    if (nl < 1 || nl > 8) return GCSSL_EBADSHAPE;
    Prep3Batch b{};
    size_t mx = 0;
    for (int i = 0; i < nl; ++i) {
        if (!w[i] || (!wf[i] && !wt[i])) return GCSSL_ENULL;
        if (Cout[i] <= 0 || Cin[i] <= 0 || CinP[i] < Cin[i]) return GCSSL_EBADSHAPE;
        b.l[i] = Prep3Layer{w[i], wf[i], wt[i], Cout[i], Cin[i], CinP[i], wk3(CinP[i]), wk3(Cout[i])};
        const size_t t = (wf[i] ? (size_t)Cout[i] * wk3(CinP[i]) : 0) + (wt[i] ? (size_t)Cin[i] * wk3(Cout[i]) : 0);
        if (t > mx) mx = t;
    }
    unsigned gx = (unsigned)((mx + 1023) / 1024); if (gx > 2048) gx = 2048; if (gx < 1) gx = 1;
    dim3 grid(gx, nl);
    if (gcssl_bad_conv_dtype(dtype)) return GCSSL_EBADDTYPE;
    b.scale = (dtype == GCSSL_F32_F16X3 || dtype == GCSSL_F32_BF16X3) ? x3_wscale() : 1.f;
    GCSSL_DISPATCH_CONV(dtype, GCSSL_LAUNCH(prep3_weight_batch_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, b));
    return gcssl_launch_status();
}

}  // extern "C"
