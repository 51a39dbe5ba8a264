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
#include <type_traits>

namespace {

constexpr int BK = 32;       // K elements per LDS tile
constexpr int NT = 256;      // threads per workgroup: 4 waves as 2 (M) x 2 (N)

// ------------------------------------------------------------------------------------------
// LDS tiles.  KMajor: [row][k], filled by k-contiguous vectors (fwd, dgrad).
//             MMajor: [k][row], filled by row-contiguous vectors (wgrad; read transposed).
// frag(row0, ks, lane) returns the MFMA operand of lane `lane` for the 32 rows row0.. and
// k-step ks (2 k per step for f32 32x32x2, 16 per step for bf16 32x32x16).
// ------------------------------------------------------------------------------------------
template <typename T, int ROWS> struct KMajor;
template <int ROWS> struct KMajor<float, ROWS> {
    static constexpr int STRIDE = BK + 1;                     // conflict-free b32 fragment reads
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
template <int ROWS> struct KMajor<bf16_t, ROWS> {
    static constexpr int KSTEPS = BK / 16;
    typedef bf16x8 Frag;
    uint4 d[ROWS * 4];                                        // 64-byte rows, 16-byte chunks XOR-swizzled
    __device__ static int swz(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }
    __device__ void store_vec(int row, int chunk, const Vec16<bf16_t>& v) { d[swz(row, chunk)] = v.v; }
    __device__ Frag frag(int row0, int ks, int lane) const {
        return __builtin_bit_cast(bf16x8, d[swz(row0 + (lane & 31), ks * 2 + (lane >> 5))]);
    }
};

template <typename T, int ROWS> struct MMajor;
template <int ROWS> struct MMajor<float, ROWS> {
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
template <int ROWS> struct MMajor<bf16_t, ROWS> {
    static constexpr int KSTEPS = BK / 16;
    static constexpr int STRIDE = ROWS * 2 + 64;              // bytes; +64 B keeps the 4 rows of a tr block on distinct banks
    typedef bf16x8 Frag;
    __attribute__((aligned(16))) unsigned char d[BK * STRIDE];
    __device__ void store_vec(int k, int chunk, const Vec16<bf16_t>& v) {
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
        return __builtin_bit_cast(bf16x8, r);
    }
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// one BK slab: acc[i][j] += A(rows wm0+32i..) x B(rows wn0+32j..)^T
template <int TM, int TN, class TA, class TB>
__device__ __forceinline__ void mma_slab(const TA& As, const TB& Bs, int wm0, int wn0, int lane,
                                         f32x16 (&acc)[TM][TN]) {
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
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct ConvParams {
    const void* x;      // fwd: input act; dgrad: dy; wgrad: x (high-res input of the conv)
    const void* w;      // fwd: Wf; dgrad: Wt; wgrad: dy (low-res)
    void* y;            // fwd: output; dgrad: dx; wgrad: slab (fp32)
    const float* bias;  // fwd only, nullable
    const float* gscale;  // per-group multiplier, nullable
    int group_n;        // samples per group
    int ldx, ldw, ldy;  // pixel strides (elements); ldw = lddy for wgrad
    int N, Hi, Wi, Cin, Cout;   // conv geometry: x is [N][Hi][Wi][Cin], y is [N][Hi/2][Wi/2][Cout]
    int lgWo, lgHoWo, lgCin, lgCout;
    int M;              // GEMM rows
    int act;            // fwd epilogue: 1 = LeakyReLU(0.2)
    int out_f32;        // fwd/dgrad: write fp32 regardless of T
    int ktiles_per_split;   // wgrad; fwd/dgrad when ksplit > 1
    int ksplit;             // fwd/dgrad: K is split over ksplit workgroups that atomically add into a zeroed fp32 output
};

// ------------------------------------------------------------------------------------------
// forward: y[m][co] = act( gscale[g(m)] * sum_{tap,ci} x[n, 2oy-1+ky, 2ox-1+kx, ci] Wf[co][tap][ci] + bias[co] )
// GEMM M = N*Ho*Wo, N = Cout, K = 16*Cin
// ------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(ConvParams p) {
    constexpr int KV = Elem<T>::KV, CH = BK / KV, RPT = NT / CH;   // rows covered per pass
    constexpr int NVA = BM / RPT, NVB = BN / RPT;
    constexpr int TM = BM / 64, TN = BN / 64;
    __shared__ KMajor<T, BM> As[2];
    __shared__ KMajor<T, BN> Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const T* x = static_cast<const T*>(p.x);
    const T* w = static_cast<const T*>(p.w);
    const int chunk = tid % CH, row_t = tid / CH;
    const int K = 16 * p.Cin;
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1;

    int pixbase[NVA], iy0[NVA], ix0[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + row_t + i * RPT;
        if (m < p.M) {
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            const int oy = rem >> p.lgWo, ox = rem & (Wo - 1);
            pixbase[i] = n * p.Hi * p.Wi; iy0[i] = 2 * oy - 1; ix0[i] = 2 * ox - 1;
        } else { pixbase[i] = -1; iy0[i] = 0; ix0[i] = 0; }
    }
    (void)Ho;
    Vec16<T> ra[NVA], rb[NVB];
    auto gload = [&](int k0) {
        const int k = k0 + chunk * KV;
        const int tap = k >> p.lgCin, ci = k & (p.Cin - 1);
        const int ky = tap >> 2, kx = tap & 3;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = pixbase[i] >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            ra[i] = ok ? Vec16<T>::load(x + (size_t)(pixbase[i] + iy * p.Wi + ix) * p.ldx + ci) : Vec16<T>::zero();
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            const int co = n0 + row_t + j * RPT;
            rb[j] = co < p.Cout ? Vec16<T>::load(w + (size_t)co * K + k) : Vec16<T>::zero();
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) As[buf].store_vec(row_t + i * RPT, chunk, ra[i]);
#pragma unroll
        for (int j = 0; j < NVB; ++j) Bs[buf].store_vec(row_t + j * RPT, chunk, rb[j]);
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
    if (t_beg < t_end) {
        gload(t_beg * BK); lstore(0); __syncthreads();
        for (int t = t_beg; t < t_end; ++t) {
            const int b = (t - t_beg) & 1;
            if (t + 1 < t_end) gload((t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < t_end) lstore(b ^ 1);
            __syncthreads();
        }
    }
    // epilogue (out_f32: the pre-norm tensor z stays fp32 in bf16 mode, see norm.hip)
    T* y = static_cast<T*>(p.y);
    float* y32 = static_cast<float*>(p.y);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + 32 * i + crow(r, lane);
            if (m >= p.M) continue;
            float s = 1.f;
            if (p.gscale) s = p.gscale[(m >> p.lgHoWo) / p.group_n];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn0 + 32 * j + (lane & 31);
                if (co >= p.Cout) continue;
                float v = acc[i][j][r] * s;
                if (p.ksplit > 1) {                       // linear epilogue only: partial sums add up, bias once
                    if (p.bias && ks == 0) v += p.bias[co];
                    atomicAdd(y32 + (size_t)m * p.ldy + co, v);
                    continue;
                }
                if (p.bias) v += p.bias[co];
                if (p.act == 1) v = lrelu_f(v);
                if (p.out_f32) y32[(size_t)m * p.ldy + co] = v;
                else Elem<T>::st(y + (size_t)m * p.ldy + co, v);
            }
        }
}

// ------------------------------------------------------------------------------------------
// dgrad / transposed conv: dx[n,iy,ix,ci] = gscale * sum_{co, taps matching parity} dy[n,oy,ox,co] Wt[ci][tap][co]
// One launch z-slice per output parity class (py,px); per class M = N*Ho*Wo, N = Cin, K = 4*Cout.
//   ky = 1-py+2ty, oy = iy' + py - ty   (iy = 2 iy' + py), same in x.
// ------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(NT) void conv_dgrad_kernel(ConvParams p) {
    constexpr int KV = Elem<T>::KV, CH = BK / KV, RPT = NT / CH;
    constexpr int NVA = BM / RPT, NVB = BN / RPT;
    constexpr int TM = BM / 64, TN = BN / 64;
    __shared__ KMajor<T, BM> As[2];
    __shared__ KMajor<T, BN> Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int cls = p.ksplit > 1 ? (int)blockIdx.z / p.ksplit : (int)blockIdx.z;
    const int ks = p.ksplit > 1 ? (int)blockIdx.z % p.ksplit : 0;
    const int py = cls >> 1, px = cls & 1;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const T* dy = static_cast<const T*>(p.x);
    const T* wt = static_cast<const T*>(p.w);
    const int chunk = tid % CH, row_t = tid / CH;
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1;
    const int K = 4 * p.Cout;

    int pixbase[NVA], yy[NVA], xx[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + row_t + i * RPT;
        if (m < p.M) {
            const int n = m >> p.lgHoWo, rem = m & ((1 << p.lgHoWo) - 1);
            pixbase[i] = n * Ho * Wo; yy[i] = (rem >> p.lgWo) + py; xx[i] = (rem & (Wo - 1)) + px;
        } else { pixbase[i] = -1; yy[i] = 0; xx[i] = 0; }
    }
    Vec16<T> ra[NVA], rb[NVB];
    auto gload = [&](int k0) {
        const int k = k0 + chunk * KV;
        const int t = k >> p.lgCout, co = k & (p.Cout - 1);
        const int ty = t >> 1, tx = t & 1;
        const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int oy = yy[i] - ty, ox = xx[i] - tx;
            const bool ok = pixbase[i] >= 0 && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
            ra[i] = ok ? Vec16<T>::load(dy + (size_t)(pixbase[i] + oy * Wo + ox) * p.ldx + co) : Vec16<T>::zero();
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            const int ci = n0 + row_t + j * RPT;
            rb[j] = ci < p.Cin ? Vec16<T>::load(wt + ((size_t)ci * 16 + tap) * p.Cout + co) : Vec16<T>::zero();
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) As[buf].store_vec(row_t + i * RPT, chunk, ra[i]);
#pragma unroll
        for (int j = 0; j < NVB; ++j) Bs[buf].store_vec(row_t + j * RPT, chunk, rb[j]);
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
    if (t_beg < t_end) {
        gload(t_beg * BK); lstore(0); __syncthreads();
        for (int t = t_beg; t < t_end; ++t) {
            const int b = (t - t_beg) & 1;
            if (t + 1 < t_end) gload((t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < t_end) lstore(b ^ 1);
            __syncthreads();
        }
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
            float s = 1.f;
            if (p.gscale) s = p.gscale[n / p.group_n];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = n0 + wn0 + 32 * j + (lane & 31);
                if (ci >= p.Cin) continue;
                const float v = acc[i][j][r] * s;
                if (p.ksplit > 1) atomicAdd(static_cast<float*>(p.y) + pix * p.ldy + ci, v);
                else if (p.out_f32) static_cast<float*>(p.y)[pix * p.ldy + ci] = v;
                else Elem<T>::st(static_cast<T*>(p.y) + pix * p.ldy + ci, v);
            }
        }
}

// ------------------------------------------------------------------------------------------
// wgrad: slab[split][co][tap][ci] = sum_{k in split} dy[k][co] * x[n, 2oy-1+ky, 2ox-1+kx, ci],  k = (n,oy,ox)
// GEMM M = Cout, N = (tap, ci), K = N*Ho*Wo split over blockIdx.z.  blockIdx.y = tap * (Cin/BN) + ci-tile.
// Both operands arrive row(k)-major with channels contiguous -> MMajor tiles, transposed LDS reads.
// ------------------------------------------------------------------------------------------
// SMALLC (first layers, Cin padded to 8): the N tile is all 16 taps x 8 channels (BN must be 128).
template <typename T, int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(ConvParams p) {
    constexpr int KV = Elem<T>::KV;
    static_assert(!SMALLC || BN == 128, "SMALLC covers 16 taps x 8 channels");
    constexpr int CHA = BM / KV, CHB = BN / KV;               // vectors per k-row
    constexpr int NVA = BK * CHA / NT, NVB = BK * CHB / NT;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 threads");
    constexpr int TM = BM / 64, TN = BN / 64;
    __shared__ MMajor<T, BM> As[2];
    __shared__ MMajor<T, BN> Bs[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * BM;
    const int ntile_ci = SMALLC ? 1 : p.Cin / BN;
    const int tap = SMALLC ? 0 : blockIdx.y / ntile_ci, ci0 = SMALLC ? 0 : (blockIdx.y % ntile_ci) * BN;
    const int ky = tap >> 2, kx = tap & 3;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const T* x = static_cast<const T*>(p.x);
    const T* dy = static_cast<const T*>(p.w);
    const int Wo = p.Wi >> 1;
    const int Ktot = p.M;                                      // N*Ho*Wo
    const int kt_beg = blockIdx.z * p.ktiles_per_split;
    int kt_end = kt_beg + p.ktiles_per_split;
    const int nkt = (Ktot + BK - 1) / BK;
    if (kt_end > nkt) kt_end = nkt;

    Vec16<T> ra[NVA], rb[NVB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int v = tid + i * NT, kr = v / CHA, c = v % CHA;
            const int k = k0 + kr;
            ra[i] = k < Ktot ? Vec16<T>::load(dy + (size_t)k * p.ldw + co0 + c * KV) : Vec16<T>::zero();
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * NT, kr = v / CHB, c = v % CHB;
            const int k = k0 + kr;
            const int n = k >> p.lgHoWo, rem = k & ((1 << p.lgHoWo) - 1);
            int kyy = ky, kxx = kx, coff = ci0 + c * KV;
            if (SMALLC) { const int tp = (c * KV) >> 3; kyy = tp >> 2; kxx = tp & 3; coff = (c * KV) & 7; }
            const int iy = 2 * (rem >> p.lgWo) - 1 + kyy, ix = 2 * (rem & (Wo - 1)) - 1 + kxx;
            const bool ok = k < Ktot && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            rb[i] = ok ? Vec16<T>::load(x + ((size_t)(n * p.Hi + iy) * p.Wi + ix) * p.ldx + coff) : Vec16<T>::zero();
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) { const int v = tid + i * NT; As[buf].store_vec(v / CHA, v % CHA, ra[i]); }
#pragma unroll
        for (int i = 0; i < NVB; ++i) { const int v = tid + i * NT; Bs[buf].store_vec(v / CHB, v % CHB, rb[i]); }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kt_beg < kt_end) {
        gload(kt_beg * BK); lstore(0); __syncthreads();
        for (int t = kt_beg; t < kt_end; ++t) {
            const int b = (t - kt_beg) & 1;
            if (t + 1 < kt_end) gload((t + 1) * BK);
            mma_slab<TM, TN>(As[b], Bs[b], wm0, wn0, lane, acc);
            if (t + 1 < kt_end) lstore(b ^ 1);
            __syncthreads();
        }
    }
    float* slab = static_cast<float*>(p.y) + (size_t)blockIdx.z * p.Cout * 16 * p.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * i + crow(r, lane);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + wn0 + 32 * j + (lane & 31);   // SMALLC: ci is the packed (tap*8 + channel) column
                slab[((size_t)co * 16 + tap) * p.Cin + ci] = acc[i][j][r];
            }
        }
}

// sum the split-K slabs, apply the spectral-norm rank-1 corrections, write PyTorch layout
//   dw[co][ci][tap] (+)= sum_s slab[s][co][tap][ci]  - sum_k coef[k]*cscale[k] u_k[co] v_k[ci*16+tap]
// One workgroup = one co x 64 ci x 16 taps: slab reads are coalesced along ci, the [tap][ci] -> [ci][tap] transpose
// goes through LDS, and the 4-KB output chunk dw[co][ci0..ci0+63][0..15] is written contiguously.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, float* __restrict__ dw,
                                    int Cout, int Cin, int Cin_real, const float* coef, const float* cscale,
                                    const float* u, int ustride, const float* v, int vstride, int nrank,
                                    int accumulate) {
    __shared__ float tile[16][65];
    const int co = blockIdx.y, ci0 = blockIdx.x * 64;
    const int cw = min(64, Cin - ci0);                       // channels in this chunk (8 for the padded first layer)
    const size_t total = (size_t)Cout * 16 * Cin;
    // blockIdx.z owns a group of splits (accumulate == 2: dw was zeroed by the caller, groups add atomically)
    const int per = (nsplit + gridDim.z - 1) / gridDim.z;
    const int k0 = blockIdx.z * per, k1 = min(nsplit, k0 + per);
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int tap = e >> 6, cil = e & 63;
        if (cil >= cw) continue;
        const size_t idx = ((size_t)co * 16 + tap) * Cin + ci0 + cil;
        float s = 0.f;
        for (int k = k0; k < k1; ++k) s += slab[(size_t)k * total + idx];
        tile[tap][cil] = s;
    }
    __syncthreads();
    float uk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        uk[k] = (k < nrank && blockIdx.z == 0) ? coef[k] * (cscale ? cscale[k] : 1.f) * u[(size_t)k * ustride + co] : 0.f;
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int cil = e >> 4, tap = e & 15;
        const int ci = ci0 + cil;
        if (cil >= cw || ci >= Cin_real) continue;
        float s = tile[tap][cil];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nrank) s -= uk[k] * v[(size_t)k * vstride + ci * 16 + tap];
        float* o = dw + ((size_t)co * Cin_real + ci) * 16 + tap;
        if (accumulate == 2) atomicAdd(o, s);
        else *o = accumulate ? (*o + s) : s;
    }
}

// fp32 PyTorch-layout weight [Cout][Cin][4][4] -> packed operand layouts in T
template <typename T>
__global__ void prep_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wt,
                                   int Cout, int Cin, int CinP) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [Cout][16][CinP]
    const size_t total = (size_t)Cout * 16 * CinP;
    if (idx >= total) return;
    const int ci = idx % CinP, tap = (idx / CinP) % 16, co = idx / ((size_t)CinP * 16);
    const float val = ci < Cin ? w[((size_t)co * Cin + ci) * 16 + tap] : 0.f;
    if (wf) Elem<T>::st(wf + idx, val);
    if (wt) Elem<T>::st(wt + ((size_t)ci * 16 + tap) * Cout + co, val);
}

template <typename T, int BM, int BN>
int launch_fwd(const ConvParams& p, hipStream_t st) {
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_fwd_kernel<T, BM, BN>), grid, dim3(NT), 0, st, p);
    return gcssl_launch_status();
}
template <typename T, int BM, int BN>
int launch_dgrad(const ConvParams& p, hipStream_t st) {
    dim3 grid((p.M + BM - 1) / BM, (p.Cin + BN - 1) / BN, 4 * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL((conv_dgrad_kernel<T, BM, BN>), grid, dim3(NT), 0, st, p);
    return gcssl_launch_status();
}

int check_geom(int N, int Hi, int Wi, int Cin, int Cout) {
    if (N <= 0 || Hi < 2 || Wi < 2 || !is_pow2(Hi) || !is_pow2(Wi) || !is_pow2(Cin) || !is_pow2(Cout))
        return GCSSL_EBADSHAPE;
    if (Cin < 8 || Cout < 8) return GCSSL_EBADSHAPE;
    return GCSSL_OK;
}

void fill_geom(ConvParams& p, int N, int Hi, int Wi, int Cin, int Cout) {
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Cout = Cout;
    p.lgWo = ilog2(Wi / 2); p.lgHoWo = ilog2((Hi / 2) * (Wi / 2));
    p.lgCin = ilog2(Cin); p.lgCout = ilog2(Cout);
    p.M = N * (Hi / 2) * (Wi / 2);
}

// pick the workgroup tile so that the grid fills the 256 CUs when the problem allows it
// Small-M layers (G.down4, the GP chain at batch B, ...) would launch far fewer than 2 workgroups per CU with a
// 64..128-deep serial K loop each: they are latency-bound, not MFMA-bound.  When the output is fp32 and the epilogue
// linear, split K across workgroups and accumulate with fp32 atomics into a zeroed output (128-B contiguous per
// half-wave: the full-rate atomic shape of MI355X_MICROARCH.md "Global float atomics").
int pick_ksplit(long tiles, int nk, bool allowed) {
    if (!allowed || tiles >= 384 || nk < 16) return 1;
    int ks = (int)((512 + tiles - 1) / tiles);
    if (ks > 8) ks = 8;
    while (ks > 1 && nk / ks < 8) --ks;
    return ks;
}

int zero_output(const ConvParams& p, long rows, int cols, hipStream_t st) {
    hipError_t e = hipMemset2DAsync(p.y, (size_t)p.ldy * 4, 0, (size_t)cols * 4, (size_t)rows, st);
    return e == hipSuccess ? GCSSL_OK : (int)e;
}

template <typename T>
int dispatch_fwd(ConvParams p, hipStream_t st) {
    const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout >= 128 && t128 >= 256) return launch_fwd<T, 128, 128>(p, st);
    if (p.Cout >= 64 && (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64) >= 256) return launch_fwd<T, 128, 64>(p, st);
    const long t64 = (long)((p.M + 63) / 64) * ((p.Cout + 63) / 64);
    const int nk = 16 * p.Cin / BK;
    const bool f32out = p.out_f32 || std::is_same<T, float>::value;
    p.ksplit = pick_ksplit(t64, nk, f32out && p.act == 0);
    if (p.ksplit > 1) {
        p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
        int rc = zero_output(p, p.M, p.Cout, st);
        if (rc) return rc;
    }
    return launch_fwd<T, 64, 64>(p, st);
}
template <typename T>
int dispatch_dgrad(ConvParams p, hipStream_t st) {
    const long t128 = 4L * ((p.M + 127) / 128) * ((p.Cin + 127) / 128);
    if (p.Cin >= 128 && t128 >= 256) return launch_dgrad<T, 128, 128>(p, st);
    if (p.Cin >= 64 && 4L * ((p.M + 127) / 128) * ((p.Cin + 63) / 64) >= 256) return launch_dgrad<T, 128, 64>(p, st);
    const long t64 = 4L * ((p.M + 63) / 64) * ((p.Cin + 63) / 64);
    const int nk = 4 * p.Cout / BK;
    const bool f32out = p.out_f32 || std::is_same<T, float>::value;
    p.ksplit = pick_ksplit(t64, nk, f32out);
    if (p.ksplit > 1) {
        p.ktiles_per_split = (nk + p.ksplit - 1) / p.ksplit;
        int rc = zero_output(p, (long)p.N * p.Hi * p.Wi, p.Cin, st);
        if (rc) return rc;
    }
    return launch_dgrad<T, 64, 64>(p, st);      // Cin < 64 (first layer, Cin padded to 8): masked columns
}

}  // namespace

extern "C" {

int gcssl_conv4x4s2_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias,
                        const float* gscale, int group_n, void* y, int ldy, int N, int Hi, int Wi,
                        int Cin, int Cout, int act, int out_f32, void* stream) {
    if (!x || !wf || !y) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (Cout < 64 || ldx < Cin || ldy < Cout || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    const int kv = dtype == GCSSL_F32 ? 4 : 8;
    if (ldx % kv || !aligned16(x) || !aligned16(wf)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = x; p.w = wf; p.y = y; p.bias = bias; p.gscale = gscale; p.group_n = group_n;
    p.ldx = ldx; p.ldy = ldy; p.act = act; p.out_f32 = out_f32;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GCSSL_F32) return dispatch_fwd<float>(p, st);
    if (dtype == GCSSL_BF16) return dispatch_fwd<bf16_t>(p, st);
    return GCSSL_EBADDTYPE;
}

int gcssl_conv4x4s2_dgrad(int dtype, const void* dy, int lddy, const void* wt, const float* gscale,
                          int group_n, void* dx, int lddx, int N, int Hi, int Wi, int Cin, int Cout,
                          int out_f32, void* stream) {
    if (!dy || !wt || !dx) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if (Cout < 8 || lddy < Cout || lddx < Cin || (gscale && group_n <= 0)) return GCSSL_EBADSHAPE;
    const int kv = dtype == GCSSL_F32 ? 4 : 8;
    if (lddy % kv || !aligned16(dy) || !aligned16(wt)) return GCSSL_EALIGN;
    ConvParams p{}; p.x = dy; p.w = wt; p.y = dx; p.gscale = gscale; p.group_n = group_n;
    p.ldx = lddy; p.ldy = lddx; p.out_f32 = out_f32;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GCSSL_F32) return dispatch_dgrad<float>(p, st);
    if (dtype == GCSSL_BF16) return dispatch_dgrad<bf16_t>(p, st);
    return GCSSL_EBADDTYPE;
}

// number of split-K slabs gcssl_conv4x4s2_wgrad will write for this geometry (caller sizes the workspace)
int gcssl_conv4x4s2_wgrad_splits(int N, int Hi, int Wi, int Cin, int Cout) {
    if (check_geom(N, Hi, Wi, Cin, Cout)) return GCSSL_EBADSHAPE;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : (Cin >= 64 ? 64 : (Cin == 8 ? 8 : 0));
    if (!bn) return GCSSL_EBADSHAPE;
    const long tiles = (long)(Cout / bm) * 16 * (Cin / bn) / (Cin == 8 ? 16 : 1);
    const int nkt = (N * (Hi / 2) * (Wi / 2) + BK - 1) / BK;
    long want = (512 + tiles - 1) / tiles;                 // ~2 workgroups per CU
    if (want > 128) want = 128;                            // padded first layers have a single tile: bound the slab count
    if (want < 1) want = 1;
    if (want > nkt) want = nkt;
    const int per = (nkt + (int)want - 1) / (int)want;
    return (nkt + per - 1) / per;
}

int gcssl_conv4x4s2_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* slab,
                          int N, int Hi, int Wi, int Cin, int Cout, void* stream) {
    if (!x || !dy || !slab) return GCSSL_ENULL;
    int rc = check_geom(N, Hi, Wi, Cin, Cout);
    if (rc) return rc;
    if ((Cin < 64 && Cin != 8) || Cout < 64 || ldx < Cin || lddy < Cout) return GCSSL_EBADSHAPE;
    const int kv = dtype == GCSSL_F32 ? 4 : 8;
    if (ldx % kv || lddy % kv || !aligned16(x) || !aligned16(dy)) return GCSSL_EALIGN;
    const int nsplit = gcssl_conv4x4s2_wgrad_splits(N, Hi, Wi, Cin, Cout);
    if (nsplit <= 0) return GCSSL_EBADSHAPE;
    ConvParams p{}; p.x = x; p.w = dy; p.y = slab; p.ldx = ldx; p.ldw = lddy;
    fill_geom(p, N, Hi, Wi, Cin, Cout);
    const int nkt = (p.M + BK - 1) / BK;
    p.ktiles_per_split = (nkt + nsplit - 1) / nsplit;
    hipStream_t st = (hipStream_t)stream;
    const int bm = Cout >= 128 ? 128 : 64, bn = Cin >= 128 ? 128 : 64;
    const bool smallc = Cin == 8;
    dim3 grid(Cout / bm, smallc ? 1 : 16 * (Cin / bn), nsplit);
#define WG(T, A, B, S) hipLaunchKernelGGL((conv_wgrad_kernel<T, A, B, S>), grid, dim3(NT), 0, st, p)
    if (dtype == GCSSL_F32) {
        if (smallc) { if (bm == 128) WG(float, 128, 128, true); else WG(float, 64, 128, true); }
        else if (bm == 128 && bn == 128) WG(float, 128, 128, false); else if (bm == 128) WG(float, 128, 64, false);
        else if (bn == 128) WG(float, 64, 128, false); else WG(float, 64, 64, false);
    } else if (dtype == GCSSL_BF16) {
        if (smallc) { if (bm == 128) WG(bf16_t, 128, 128, true); else WG(bf16_t, 64, 128, true); }
        else if (bm == 128 && bn == 128) WG(bf16_t, 128, 128, false); else if (bm == 128) WG(bf16_t, 128, 64, false);
        else if (bn == 128) WG(bf16_t, 64, 128, false); else WG(bf16_t, 64, 64, false);
    } else return GCSSL_EBADDTYPE;
#undef WG
    return gcssl_launch_status();
}

int gcssl_wgrad_reduce(const float* slab, int nsplit, float* dw, int Cout, int Cin, int Cin_real,
                       const float* coef, const float* cscale, const float* u, int ustride, const float* v,
                       int vstride, int nrank, int accumulate, void* stream) {
    if (!slab || !dw || (nrank > 0 && (!coef || !u || !v))) return GCSSL_ENULL;
    if (nsplit <= 0 || Cout <= 0 || Cin <= 0 || Cin_real <= 0 || Cin_real > Cin) return GCSSL_EBADSHAPE;
    if (nrank > 4 || (nrank > 0 && (ustride < Cout || vstride < Cin_real * 16))) return GCSSL_EBADSHAPE;
    int zg = 1;
    if (accumulate == 2) { zg = (nsplit + 15) / 16; if (zg > 16) zg = 16; }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cin + 63) / 64), (unsigned)Cout, (unsigned)zg), dim3(256), 0,
                       (hipStream_t)stream, slab, nsplit, dw, Cout, Cin, Cin_real, coef, cscale, u, ustride, v, vstride, nrank, accumulate);
    return gcssl_launch_status();
}

int gcssl_prep_conv_weight(int dtype, const float* w, void* wf, void* wt, int Cout, int Cin, int CinP,
                           void* stream) {
    if (!w || (!wf && !wt)) return GCSSL_ENULL;
    if (Cout <= 0 || Cin <= 0 || CinP < Cin) return GCSSL_EBADSHAPE;
    const size_t total = (size_t)Cout * 16 * CinP;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == GCSSL_F32)
        hipLaunchKernelGGL(prep_weight_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (float*)wf, (float*)wt, Cout, Cin, CinP);
    else if (dtype == GCSSL_BF16)
        hipLaunchKernelGGL(prep_weight_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, w,
                           (bf16_t*)wf, (bf16_t*)wt, Cout, Cin, CinP);
    else return GCSSL_EBADDTYPE;
    return gcssl_launch_status();
}

}  // extern "C"
