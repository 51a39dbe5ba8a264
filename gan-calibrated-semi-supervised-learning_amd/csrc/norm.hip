// InstanceNorm2d (affine=False, eps 1e-5, biased variance) fused with its activation, dropout and
// the bookkeeping the WGAN-GP step needs -- forward, backward and DOUBLE backward -- for NHWC
// activations on gfx950.  These are HBM-bound streaming/reduction kernels: a workgroup owns one
// sample x 64 channels, lanes run along the contiguous channel axis (coalesced 128/256-byte rows)
// and the H*W reduction is done per thread + a 4-way LDS combine (no atomics on the statistics).
//
// Replaces: nn.InstanceNorm2d + LeakyReLU/ReLU (+ Dropout) at cgan/models.py:59-63,73-76,114,241-242
// and their autograd first/second-order backward used by cgan/losses.py:213-220 (create_graph=True).
// Formulas: oracle/manual_step.py (in_bwd, in_bwd_bwd), verified against autograd.
//
// The pre-norm tensor z is ALWAYS fp32 (also in bf16 mode): with only 4..64 elements per (n,c) plane at 32x32 inputs,
// z - mean(z) cancels most of a bf16 mantissa (measured: 18 % error on the gradient penalty with bf16 z).  For the
// same reason every INCOMING gradient of the backward kernels (da, da2, gb_a, qz, zt) is fp32 -- they come out of fp32
// MFMA accumulators anyway -- and only the tensors that feed the next MFMA (a, dzs, gt_a, gb_zs) are in `dtype`.
#include "common.h"
#include <algorithm>

namespace {

constexpr int CW = 64;     // channels per workgroup
constexpr int RG = 4;      // row groups (threads along H*W)
constexpr float IN_EPS = 1e-5f;

// combine RG partial sums per channel through LDS; every thread gets its channel's total
template <int NV>
__device__ __forceinline__ void combine(float (&v)[NV], float (*sm)[RG][CW], int tx, int ty) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[i][ty][tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) s += sm[i][g][tx];
        v[i] = s;
    }
}

__device__ __forceinline__ float act_fwd(float x, int act) { return act == 1 ? lrelu_f(x) : (x > 0.f ? x : 0.f); }
__device__ __forceinline__ float act_grad(float xhat, int act) { return xhat > 0.f ? 1.f : (act == 1 ? 0.2f : 0.f); }

// ------------------------------------------------------------------------------------------------------------
// Vectorised layout shared by the forward/backward kernels: a workgroup = 64 channels x 16 row groups, every lane
// owns 4 consecutive channels (one 16-byte load of the fp32 z / incoming gradient per row, 8-byte bf16 stores).
//   small maps (H*W <= 64): one fused kernel, the <=4 rows of a lane stay in registers -> z is read ONCE, two-pass
//                           exact statistics;
//   large maps:             stats (shifted sums, atomics) -> finalize -> apply, all fully parallel over H*W chunks.
// ------------------------------------------------------------------------------------------------------------
constexpr int VC = 4, CGN = 16, RGN = 16, MAXR = 4, SMALL_HW = RGN * MAXR;
constexpr int BIGR = 16, MID_HW = RGN * BIGR;   // 64 < H*W <= 256: same fused kernels with 16 rows per lane (forward, backward)

__device__ __forceinline__ int replica_offset(int nrep, int rep_stride) {
    if (nrep <= 1) return 0;
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    return (int)(wg % (unsigned)nrep) * rep_stride;
}

// Sum RG_ consecutive row groups per 4-channel column; every thread gets its segment's totals.  Threads are laid out
// threadIdx = ty * 16 + tx, so a wave holds the row groups 4w..4w+3 at lanes l, l+16, l+32, l+48: two xor-shuffles sum
// them in registers (a 4-group segment is done there), and only the four per-wave partials of a 16-group segment go
// through LDS.  (The first form -- every thread re-reading all 16 partials of its NV x 4 values from LDS, fully unrolled --
// cost 128 VGPRs in the backward kernel and held it at one wave per SIMD.)
template <int NV, int RG_>
__device__ __forceinline__ void combine_seg(float (&v)[NV][VC], float (*sm)[RGN][CW], int tx, int ty) {
    static_assert(RG_ == 1 || RG_ == 4 || RG_ == 16, "row groups per segment");
    static_assert(CGN == 16 && RGN == 16, "lane layout");
    if (RG_ == 1) return;                                  // the lane already holds its sample's full column sums
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VC; ++j) {
            v[i][j] += __shfl_xor(v[i][j], 16, 64);
            v[i][j] += __shfl_xor(v[i][j], 32, 64);
        }
    if (RG_ == 4) return;
    const int w = ty >> 2;
    __syncthreads();
    if ((ty & 3) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) sm[i][w][tx * VC + j] = v[i][j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VC; ++j)
            v[i][j] = (sm[i][0][tx * VC + j] + sm[i][1][tx * VC + j]) + (sm[i][2][tx * VC + j] + sm[i][3][tx * VC + j]);
}
// all 16 row groups of the workgroup
template <int NV>
__device__ __forceinline__ void combine16(float (&v)[NV][VC], float (*sm)[RGN][CW], int tx, int ty) {
    combine_seg<NV, 16>(v, sm, tx, ty);
}

// Small maps (H*W <= 64): a lane owns 4 rows x 4 channels, RG_ = ceil(HW/4) in {1,4,16} row-group lanes make up one
// sample and 16/RG_ samples share a 256-thread pass (so 2x2 and 4x4 maps still use every lane); a workgroup runs
// several passes and keeps its bias / spectral-norm partial sums in registers -> ONE set of atomics per workgroup
// (same-address float atomics serialise at ~12 ns each: 1536 workgroups on 3 addresses cost more than the data pass).

__device__ __forceinline__ void ld4(const float* p, float (&o)[VC]) {
    const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[VC]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[VC]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void st4_16(T* p, const float (&v)[VC]) {
    uint2 w;
    w.x = pack2<T>(v[0], v[1]);
    w.y = pack2<T>(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = w;
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[VC]) { st4_16<bf16_t>(p, v); }
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, const float (&v)[VC]) { st4_16<f16_t>(p, v); }
template <typename T> __device__ __forceinline__ void ldT4(const T* p, float (&o)[VC]);
template <> __device__ __forceinline__ void ldT4<float>(const float* p, float (&o)[VC]) { ld4(p, o); }
template <typename T> __device__ __forceinline__ void ldT4_16(const T* p, float (&o)[VC]) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    o[0] = Bits16<T>::dec(w.x); o[1] = Bits16<T>::dec(w.x >> 16);
    o[2] = Bits16<T>::dec(w.y); o[3] = Bits16<T>::dec(w.y >> 16);
}
template <> __device__ __forceinline__ void ldT4<bf16_t>(const bf16_t* p, float (&o)[VC]) { ldT4_16<bf16_t>(p, o); }
template <> __device__ __forceinline__ void ldT4<f16_t>(const f16_t* p, float (&o)[VC]) { ldT4_16<f16_t>(p, o); }
__device__ __forceinline__ void keep4(const uint8_t* mp, float (&k)[VC]) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(mp);
#pragma unroll
    for (int j = 0; j < VC; ++j) k[j] = ((w >> (8 * j)) & 0xFF) ? 2.f : 0.f;
}

// 16-byte row accesses through buffer descriptors (common.h): offsets that are OOB read 0 / drop the store
__device__ __forceinline__ void bld4(__amdgpu_buffer_rsrc_t r, unsigned off, float (&o)[VC]) {
    const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <typename T> __device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[VC]) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), r, off, 0, 0);
    } else {
        const u32x2 w = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(w, r, off, 0, 0);
    }
}
__device__ __forceinline__ unsigned rsrc_bytes(size_t b) { return b < 0x7fffffffu ? (unsigned)b : 0x7fffffffu; }

// ---- forward, small maps: a = act((z - mean) * rstd) [* keep * 2]; writes mean/rstd [N][C]
// Row accesses are branch-free buffer accesses (see in_bwd_small_kernel): dead rows read 0 and are left out of the variance.
template <typename T, int RG, int MR = MAXR, bool SLAB = false>
__global__ __launch_bounds__(CGN * RGN) void in_fwd_small_kernel(float* __restrict__ z, int ldz, T* __restrict__ a, int lda,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                const uint8_t* __restrict__ mask, int N, int HW, int C, int act,
                                                                int spb, int nslab, long slab_stride) {
    __shared__ float sm[1][RGN][CW];
    constexpr int SPP = RGN / RG;
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int slot = ty / RG, rg = ty % RG;
    const int c = blockIdx.x * CW + tx * VC;
    const size_t nhw = (size_t)N * HW;
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(z, rsrc_bytes(nhw * ldz * 4)), mkr = make_rsrc(mask, mask ? rsrc_bytes(nhw * C) : 0u),
                                 outr = make_rsrc(a, rsrc_bytes(nhw * lda * sizeof(T)));
    for (int n0 = blockIdx.y * spb; n0 < min(N, (int)(blockIdx.y + 1) * spb); n0 += SPP) {
        const int n = n0 + slot;
        const bool live = n < N;
        unsigned pix[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) { const int p = rg + RG * i; pix[i] = (live && p < HW) ? (unsigned)(n * HW + p) : OOB; }
        float v[MR][VC];
#pragma unroll
        for (int i = 0; i < MR; ++i) bld4(zr, pix[i] != OOB ? (pix[i] * ldz + c) * 4u : OOB, v[i]);
        if (SLAB) {                               // split-K partial sums of the producing conv: add the slabs, keep the total
            for (int k = 1; k < nslab; ++k) {
                const __amdgpu_buffer_rsrc_t sr = make_rsrc(z + (size_t)k * slab_stride, rsrc_bytes(nhw * ldz * 4));
                float t[MR][VC];
#pragma unroll
                for (int i = 0; i < MR; ++i) bld4(sr, pix[i] != OOB ? (pix[i] * ldz + c) * 4u : OOB, t[i]);
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < VC; ++j) v[i][j] += t[i][j];
            }
#pragma unroll
            for (int i = 0; i < MR; ++i) bst4<float>(zr, pix[i] != OOB ? (pix[i] * ldz + c) * 4u : OOB, v[i]);
        }
        float s[1][VC] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) s[0][j] += v[i][j];
        combine_seg<1, RG>(s, sm, tx, ty);
        float mu[VC], r[VC];
#pragma unroll
        for (int j = 0; j < VC; ++j) { mu[j] = s[0][j] / HW; s[0][j] = 0.f; }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) { const float d = v[i][j] - mu[j]; s[0][j] += pix[i] != OOB ? d * d : 0.f; }
        combine_seg<1, RG>(s, sm, tx, ty);
#pragma unroll
        for (int j = 0; j < VC; ++j) r[j] = 1.0f / sqrtf(s[0][j] / HW + IN_EPS);
        if (live && rg == 0) {
            st4<float>(mean + (size_t)n * C + c, mu);
            st4<float>(rstd + (size_t)n * C + c, r);
        }
        unsigned w[MR];
        if (mask) {
#pragma unroll
            for (int i = 0; i < MR; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b32(mkr, pix[i] != OOB ? pix[i] * C + c : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            float o[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                o[j] = act_fwd((v[i][j] - mu[j]) * r[j], act);
                if (mask) o[j] *= ((w[i] >> (8 * j)) & 0xFF) ? 2.f : 0.f;
            }
            bst4<T>(outr, pix[i] != OOB ? (pix[i] * lda + c) * (unsigned)sizeof(T) : OOB, o);
        }
    }
}

// ---- forward, maps of 257..1024 pixels: one workgroup owns (sample, 32 channels) and keeps its slab of z in LDS
// (<= 128 KB), so z is read from HBM ONCE, the statistics are the exact two-pass ones, and mean/rstd/pool have a single
// writer (no atomics, no fills, no finalize launch).  Replaces stats + finalize + apply (+2 fills) for G.up4 at 32x32.
constexpr int LDS_HW_MAX = 1024;
template <typename T, int LDS_CH>
__global__ __launch_bounds__(256) void in_fwd_lds_kernel(const float* __restrict__ z, int ldz, T* __restrict__ a, int lda,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         const uint8_t* __restrict__ mask, float* __restrict__ pool,
                                                         int HW, int C, int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* slab = reinterpret_cast<float*>(lds_raw);                    // [HW][32]
    constexpr int LPR = LDS_CH / VC, RGS = 256 / LPR;                    // lanes per pixel row, row groups
    __shared__ float part[RGS][LDS_CH + 1];
    __shared__ float stat[2][LDS_CH];
    const int tx = threadIdx.x % LPR, ty = threadIdx.x / LPR;            // LPR lanes x 4 channels, RGS row groups
    const int c = blockIdx.x * LDS_CH + tx * VC, n = blockIdx.y;
    const float* zp = z + (size_t)n * HW * ldz + c;
    float s[VC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = ty; p < HW; p += RGS) {
        float v[VC]; ld4(zp + (size_t)p * ldz, v);
        *reinterpret_cast<float4*>(slab + p * LDS_CH + tx * VC) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int j = 0; j < VC; ++j) s[j] += v[j];
    }
    auto combine = [&](float (&v)[VC], int which) {                     // sum over the 32 row groups -> stat[which][ch]
#pragma unroll
        for (int j = 0; j < VC; ++j) part[ty][tx * VC + j] = v[j];
        __syncthreads();
        if (threadIdx.x < LDS_CH) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < RGS; ++g) t += part[g][threadIdx.x];
            stat[which][threadIdx.x] = t;
        }
        __syncthreads();
    };
    combine(s, 0);
    float mu[VC], r[VC];
#pragma unroll
    for (int j = 0; j < VC; ++j) { mu[j] = stat[0][tx * VC + j] / HW; s[j] = 0.f; }
    for (int p = ty; p < HW; p += RGS) {
        const float4 v = *reinterpret_cast<const float4*>(slab + p * LDS_CH + tx * VC);
        const float d0 = v.x - mu[0], d1 = v.y - mu[1], d2 = v.z - mu[2], d3 = v.w - mu[3];
        s[0] += d0 * d0; s[1] += d1 * d1; s[2] += d2 * d2; s[3] += d3 * d3;
    }
    combine(s, 1);
#pragma unroll
    for (int j = 0; j < VC; ++j) { r[j] = 1.0f / sqrtf(stat[1][tx * VC + j] / HW + IN_EPS); s[j] = 0.f; }
    if (ty == 0) { st4<float>(mean + (size_t)n * C + c, mu); st4<float>(rstd + (size_t)n * C + c, r); }
    T* ap = a + (size_t)n * HW * lda + c;
    for (int p = ty; p < HW; p += RGS) {
        const float4 v4 = *reinterpret_cast<const float4*>(slab + p * LDS_CH + tx * VC);
        const float v[VC] = {v4.x, v4.y, v4.z, v4.w};
        float o[VC], k[VC] = {1.f, 1.f, 1.f, 1.f};
        if (mask) keep4(mask + ((size_t)n * HW + p) * C + c, k);
#pragma unroll
        for (int j = 0; j < VC; ++j) { o[j] = act_fwd((v[j] - mu[j]) * r[j], act) * k[j]; s[j] += o[j]; }
        if (a) st4<T>(ap + (size_t)p * lda, o);                           // (a == nullptr: only the statistics and the pool sums are wanted)
    }
    if (pool) {                                                         // sum over H*W of the activation (global average pool)
        combine(s, 0);
        if (threadIdx.x < LDS_CH) pool[(size_t)n * C + blockIdx.x * LDS_CH + threadIdx.x] += stat[0][threadIdx.x];
    }
}

// ---- forward, large maps.  (1) shifted sums: s1 += sum(z - z0), s2 += sum((z - z0)^2), z0 = z[n, 0, c]
__global__ __launch_bounds__(CGN * RGN) void in_stats_kernel(const float* __restrict__ z, int ldz, float* __restrict__ s1,
                                                            float* __restrict__ s2, int HW, int C) {
    __shared__ float sm[2][RGN][CW];
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int c = blockIdx.x * CW + tx * VC, n = blockIdx.y;
    const float* zp = z + (size_t)n * HW * ldz + c;
    float z0[VC]; ld4(zp, z0);
    float s[2][VC] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int p0 = blockIdx.z * SMALL_HW;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int p = p0 + ty + RGN * i;
        if (p >= HW) continue;
        float v[VC]; ld4(zp + (size_t)p * ldz, v);
#pragma unroll
        for (int j = 0; j < VC; ++j) { const float d = v[j] - z0[j]; s[0][j] += d; s[1][j] += d * d; }
    }
    combine16<2>(s, sm, tx, ty);
    if (ty == 0) {
#pragma unroll
        for (int j = 0; j < VC; ++j) { atomicAdd(s1 + (size_t)n * C + c + j, s[0][j]); atomicAdd(s2 + (size_t)n * C + c + j, s[1][j]); }
    }
}
// (2) mean = z0 + s1/HW; var = s2/HW - (s1/HW)^2
__global__ void in_finalize_kernel(const float* __restrict__ z, int ldz, float* __restrict__ mean, float* __restrict__ rstd,
                                   int N, int HW, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * C) return;
    const int n = i / C, c = i % C;
    const float z0 = z[(size_t)n * HW * ldz + c];
    const float m = mean[i] / HW;
    const float var = fmaxf(rstd[i] / HW - m * m, 0.f);
    mean[i] = z0 + m;
    rstd[i] = 1.0f / sqrtf(var + IN_EPS);
}
// (3) apply (+ optional global-average-pool partial sums: pool[n][c] += sum a)
template <typename T>
__global__ __launch_bounds__(CGN * RGN) void in_apply_kernel(const float* __restrict__ z, int ldz, T* __restrict__ a, int lda,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ pool,
                                                            int HW, int C, int act) {
    __shared__ float sm[1][RGN][CW];
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int c = blockIdx.x * CW + tx * VC, n = blockIdx.y;
    const float* zp = z + (size_t)n * HW * ldz + c;
    T* ap = a + (size_t)n * HW * lda + c;
    float mu[VC], r[VC];
    ld4(mean + (size_t)n * C + c, mu); ld4(rstd + (size_t)n * C + c, r);
    float ps[1][VC] = {{0.f, 0.f, 0.f, 0.f}};
    const int p0 = blockIdx.z * SMALL_HW;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int p = p0 + ty + RGN * i;
        if (p >= HW) continue;
        float v[VC], o[VC], k[VC] = {1.f, 1.f, 1.f, 1.f};
        ld4(zp + (size_t)p * ldz, v);
        if (mask) keep4(mask + ((size_t)n * HW + p) * C + c, k);
#pragma unroll
        for (int j = 0; j < VC; ++j) { o[j] = act_fwd((v[j] - mu[j]) * r[j], act) * k[j]; ps[0][j] += o[j]; }
        st4<T>(ap + (size_t)p * lda, o);
    }
    if (pool) {
        combine16<1>(ps, sm, tx, ty);
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < VC; ++j) atomicAdd(pool + (size_t)n * C + c + j, ps[0][j]);
        }
    }
}

// An incoming gradient that a K-split conv left as `nslab` partial-sum slabs: this thread's rows (rg, rg+RG, ...) of its
// 4-channel column are summed into slab 0, which the caller then reads like an ordinary tensor (same thread, same
// addresses: program order).  Kept OUT of the callers' row loops: a store inside them stops the compiler from hoisting the
// later rows' loads above it (measured: in_bwd_small 17.7 -> 27.9 us with the fold inline, slabs or not).
template <int RG, int MR>
__device__ __forceinline__ void fold_slabs(float* base, int ld, int nslab, long stride, int rg, int HW) {
    for (int i = 0; i < MR; ++i) {
        const int p = rg + RG * i;
        if (p >= HW) break;
        float t[VC]; ld4(base + (size_t)p * ld, t);
        for (int k = 1; k < nslab; ++k) {
            float u[VC]; ld4(base + (size_t)k * stride + (size_t)p * ld, u);
#pragma unroll
            for (int j = 0; j < VC; ++j) t[j] += u[j];
        }
        st4<float>(base + (size_t)p * ld, t);
    }
}

struct InBwdParams {
    const float* da; int ldda;         // incoming gradient of the activation output, fp32 (nullable if da_bcast)
    const float* da2; int ldda2;       // optional second gradient added to da (skip connection), fp32
    const float* da_bcast;             // optional [N][C] gradient broadcast over H*W (global-avg-pool backward)
    const void* z; int ldz;            // pre-norm conv output, fp32 -- or (AZ kernels) the 16-bit LeakyReLU activation it became
    const float* mean; const float* rstd;
    const uint8_t* mask;               // dropout keep mask [N][HW][C], nullable
    const float* zt; int zt_n0;        // optional fp32 double-backward term added to dz of samples n >= zt_n0 ([N-zt_n0][HW][C])
    const float* gscale; int group_n;  // output multiplier per sample group, nullable
    const float* bias;                 // conv bias (for the spectral-norm <dz, z-b> term), nullable
    void* dzs; int lddz;               // output: dz * gscale
    float* dbias;                      // [C] += sum dz (atomic), nullable
    float* cdot;                       // [ngroups] += sum dzs (z - bias) = <G_k, W_orig>/sigma_k^2 (atomic), nullable
    int nrep, rep_stride;              // dbias/cdot are nrep replicas rep_stride floats apart; a workgroup adds to one
    int da_nslab; long da_slab_stride; // da is the first of da_nslab split-K partial-sum slabs of the producing conv (floats apart)
    int HW, C, act;
    const float* pre_cnt; const float* pre_pos; float pre_pos_scale;   // optional precomputed sums (see gcssl_in_act_bwd)
    unsigned* sat;                     // += number of dzs values clipped by the fp16 store (nullable; common.h sat_hits)
};

// small maps: fused backward, several samples per pass and several passes per workgroup (see combine_seg)
// SLAB: the incoming gradient arrives as split-K slabs.  A separate instantiation, because the mere presence of the fold's
// store (taken or not) cost the plain kernel 17 -> 28 us: the compiler stops overlapping the samples' loads across it.
//
// The row loops are BRANCH-FREE: every row access is a raw buffer access whose offset is OOB for rows past H*W / samples
// past N (reads 0, store dropped), and an absent optional operand sits behind ONE uniform branch around its whole row set.
// With per-row `if (p >= HW) continue` / `if (da2p) ...` tests hipcc emitted one exec-mask branch and one vmcnt drain per
// load -- 4 rows x up to 5 operands of serial round trips at one wave per SIMD (256 VGPRs): D.c2's backward over 768
// samples took 52 us for 71 MB (1.4 TB/s).  The host refuses tensors of 2 GiB or more (32-bit byte offsets).
// AZ: q.z is not the fp32 pre-norm tensor but the 16-bit ACTIVATION lrelu(xhat) that gcssl_conv4x4s2_in_act_fwd stored (the
// fp32 z never went to memory): xhat = a > 0 ? a : 5 a, and z - bias = xhat / rstd + mean - bias for the spectral-norm dot.
template <typename T, int RG, int MR = MAXR, bool SLAB = false, bool AZ = false>
__global__ __launch_bounds__(CGN * RGN) void in_bwd_small_kernel(InBwdParams q, int N, int spb, int mixed_groups) {
    __shared__ float sm[2][RGN][CW];
    __shared__ float red[CGN * RGN / 64];
    constexpr int SPP = RGN / RG;
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int slot = ty / RG, rg = ty % RG;
    const int c = blockIdx.x * CW + tx * VC;
    const int HW = q.HW, C = q.C;
    const size_t nhw = (size_t)N * HW;
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(q.z, rsrc_bytes(nhw * q.ldz * (AZ ? sizeof(T) : 4))),
                                 dar = make_rsrc(q.da, q.da ? rsrc_bytes(nhw * q.ldda * 4) : 0u),
                                 da2r = make_rsrc(q.da2, q.da2 ? rsrc_bytes(nhw * q.ldda2 * 4) : 0u),
                                 mkr = make_rsrc(q.mask, q.mask ? rsrc_bytes(nhw * C) : 0u),
                                 ztr = make_rsrc(q.zt, q.zt ? rsrc_bytes((size_t)(N - q.zt_n0) * HW * C * 4) : 0u),
                                 outr = make_rsrc(q.dzs, rsrc_bytes(nhw * q.lddz * sizeof(T)));
    float b[VC] = {0.f, 0.f, 0.f, 0.f};
    if (q.bias) ld4(q.bias + c, b);
    float sb[1][VC] = {{0.f, 0.f, 0.f, 0.f}};
    float sd = 0.f;
    int nsat = 0;
    const int nb = blockIdx.y * spb;
    for (int n0 = nb; n0 < min(N, nb + spb); n0 += SPP) {
        const int n = n0 + slot;
        const bool live = n < N;
        float mu[VC] = {0.f, 0.f, 0.f, 0.f}, r[VC] = {0.f, 0.f, 0.f, 0.f}, dab[VC] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            ld4(q.mean + (size_t)n * C + c, mu); ld4(q.rstd + (size_t)n * C + c, r);
            if (q.da_bcast) ld4(q.da_bcast + (size_t)n * C + c, dab);
        }
        if (SLAB && q.da && live)                                    // split-K slabs of the producing conv
            fold_slabs<RG, MR>(const_cast<float*>(q.da) + (size_t)n * HW * q.ldda + c, q.ldda, q.da_nslab, q.da_slab_stride, rg, HW);
        unsigned pix[MR];                                            // pixel index of row i, or OOB
#pragma unroll
        for (int i = 0; i < MR; ++i) { const int p = rg + RG * i; pix[i] = (live && p < HW) ? (unsigned)(n * HW + p) : OOB; }
        float zv[MR][VC], dn[MR][VC];                                // zv: z (fp32 source) or xhat (AZ)
        if constexpr (AZ) {
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(zr, pix[i] != OOB ? (pix[i] * q.ldz + c) * (unsigned)sizeof(T) : OOB, 0, 0);
                zv[i][0] = Bits16<T>::dec(w[0]); zv[i][1] = Bits16<T>::dec(w[0] >> 16);
                zv[i][2] = Bits16<T>::dec(w[1]); zv[i][3] = Bits16<T>::dec(w[1] >> 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MR; ++i) bld4(zr, pix[i] != OOB ? (pix[i] * q.ldz + c) * 4u : OOB, zv[i]);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) bld4(dar, pix[i] != OOB ? (pix[i] * q.ldda + c) * 4u : OOB, dn[i]);
        if (q.da2) {
            float t[MR][VC];
#pragma unroll
            for (int i = 0; i < MR; ++i) bld4(da2r, pix[i] != OOB ? (pix[i] * q.ldda2 + c) * 4u : OOB, t[i]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < VC; ++j) dn[i][j] = (dab[j] + dn[i][j]) + t[i][j];
        } else {
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < VC; ++j) dn[i][j] = dab[j] + dn[i][j];
        }
        if (q.mask) {
            unsigned w[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b32(mkr, pix[i] != OOB ? pix[i] * C + c : OOB, 0, 0);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < VC; ++j) dn[i][j] *= ((w[i] >> (8 * j)) & 0xFF) ? 2.f : 0.f;
        }
        float s[2][VC] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                if constexpr (AZ) zv[i][j] = zv[i][j] > 0.f ? zv[i][j] : 5.0f * zv[i][j];   // invert LeakyReLU(0.2): xhat
                const float xh = AZ ? zv[i][j] : (zv[i][j] - mu[j]) * r[j];
                dn[i][j] = dn[i][j] * act_grad(xh, q.act);           // (dead rows: dn = 0)
                s[0][j] += dn[i][j]; s[1][j] += dn[i][j] * xh;
            }
        combine_seg<2, RG>(s, sm, tx, ty);
        const float gs = (q.gscale && live) ? q.gscale[n / q.group_n] : 1.f;
        float zt[MR][VC];
        if (q.zt) {
            const bool tn = live && n >= q.zt_n0;
#pragma unroll
            for (int i = 0; i < MR; ++i)
                bld4(ztr, (tn && pix[i] != OOB) ? ((pix[i] - (unsigned)(q.zt_n0 * HW)) * C + c) * 4u : OOB, zt[i]);
        } else {
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < VC; ++j) zt[i][j] = 0.f;
        }
        float m1[VC], m2[VC], sdev[VC], mb[VC];
#pragma unroll
        for (int j = 0; j < VC; ++j) {
            m1[j] = s[0][j] / HW; m2[j] = s[1][j] / HW;
            sdev[j] = (AZ && live) ? 1.0f / r[j] : 0.f; mb[j] = mu[j] - b[j];      // AZ: z - bias = xhat * sdev + (mean - bias)
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            float o[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                const float xh = AZ ? zv[i][j] : (zv[i][j] - mu[j]) * r[j];
                float dz = r[j] * (dn[i][j] - m1[j] - xh * m2[j]) + zt[i][j];
                dz = pix[i] != OOB ? dz : 0.f;                       // dead rows add nothing to the bias / SN sums
                sb[0][j] += dz; sd += dz * gs * (AZ ? xh * sdev[j] + mb[j] : zv[i][j] - b[j]);
                o[j] = dz * gs;
            }
            nsat += sat_hits<T>(o);
            bst4<T>(outr, pix[i] != OOB ? (pix[i] * q.lddz + c) * (unsigned)sizeof(T) : OOB, o);
        }
        if (mixed_groups && q.cdot) {          // tiny batches only: a pass may straddle sample groups
            if (live && sd != 0.f) atomicAdd(q.cdot + replica_offset(q.nrep, q.rep_stride) + n / q.group_n, sd);
            sd = 0.f;
        }
    }
    if (q.dbias) {
        combine16<1>(sb, reinterpret_cast<float(*)[RGN][CW]>(sm), tx, ty);
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < VC; ++j) atomicAdd(q.dbias + replica_offset(q.nrep, q.rep_stride) + c + j, sb[0][j]);
        }
    }
    if (q.cdot && !mixed_groups) {
        const float tot = block_sum<CGN * RGN / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot + replica_offset(q.nrep, q.rep_stride) + nb / q.group_n, tot);     // all samples of a workgroup share a group
    }
    sat_commit(q.sat, nsat);
}

// dn = act'(xhat) (da + da2 + da_bcast) [*2 keep];  dz = rstd (dn - mean(dn) - xhat mean(dn xhat)) [+ zt]
// MODE 0: fused small-map kernel (rows in registers); MODE 1: large maps, sums only (ws[n][c][0..1] += sum dn, sum dn xhat);
// MODE 2: large maps, apply (reads the sums from ws).
template <typename T, int MODE>
__global__ __launch_bounds__(CGN * RGN) void in_bwd_kernel(InBwdParams q, float* __restrict__ ws) {
    __shared__ float sm[2][RGN][CW];
    __shared__ float red[CGN * RGN / 64];
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int c = blockIdx.x * CW + tx * VC, n = blockIdx.y;
    const int HW = q.HW, C = q.C;
    const float* zp = static_cast<const float*>(q.z) + (size_t)n * HW * q.ldz + c;
    const float* dap = q.da ? q.da + (size_t)n * HW * q.ldda + c : nullptr;
    const float* da2p = q.da2 ? q.da2 + (size_t)n * HW * q.ldda2 + c : nullptr;
    const uint8_t* mp = q.mask ? q.mask + (size_t)n * HW * C + c : nullptr;
    float mu[VC], r[VC], dab[VC] = {0.f, 0.f, 0.f, 0.f};
    ld4(q.mean + (size_t)n * C + c, mu); ld4(q.rstd + (size_t)n * C + c, r);
    if (q.da_bcast) ld4(q.da_bcast + (size_t)n * C + c, dab);
    const int p0 = MODE == 0 ? 0 : blockIdx.z * SMALL_HW;
    float zv[MAXR][VC], dn[MAXR][VC];
    float s[2][VC] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int p = p0 + ty + RGN * i;
        if (p >= HW) continue;
        ld4(zp + (size_t)p * q.ldz, zv[i]);
        float d[VC] = {dab[0], dab[1], dab[2], dab[3]};
        if (dap) {
            float t[VC]; ld4(dap + (size_t)p * q.ldda, t);
#pragma unroll
            for (int j = 0; j < VC; ++j) d[j] += t[j];
        }
        if (da2p) {
            float t[VC]; ld4(da2p + (size_t)p * q.ldda2, t);
#pragma unroll
            for (int j = 0; j < VC; ++j) d[j] += t[j];
        }
        float k[VC] = {1.f, 1.f, 1.f, 1.f};
        if (mp) keep4(mp + (size_t)p * C, k);
#pragma unroll
        for (int j = 0; j < VC; ++j) {
            const float xh = (zv[i][j] - mu[j]) * r[j];
            dn[i][j] = d[j] * k[j] * act_grad(xh, q.act);
            s[0][j] += dn[i][j]; s[1][j] += dn[i][j] * xh;
        }
    }
    if (MODE == 1) {
        combine16<2>(s, sm, tx, ty);
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                atomicAdd(ws + ((size_t)n * C + c + j) * 2, s[0][j]);
                atomicAdd(ws + ((size_t)n * C + c + j) * 2 + 1, s[1][j]);
            }
        }
        return;
    }
    float m1[VC], m2[VC];
    if (MODE == 0) {
        combine16<2>(s, sm, tx, ty);
#pragma unroll
        for (int j = 0; j < VC; ++j) { m1[j] = s[0][j] / HW; m2[j] = s[1][j] / HW; }
    } else if (q.pre_cnt) {
        // the incoming gradient is the per-(n, c) constant dab and the activation is ReLU: sum dn = dab * #{xhat > 0} and
        // sum dn xhat = dab * sum relu(xhat), both of which the forward kernel already produced (count, pooled sum)
        float cn[VC], ps[VC];
        ld4(q.pre_cnt + (size_t)n * C + c, cn); ld4(q.pre_pos + (size_t)n * C + c, ps);
#pragma unroll
        for (int j = 0; j < VC; ++j) { m1[j] = dab[j] * cn[j] / HW; m2[j] = dab[j] * (ps[j] * q.pre_pos_scale) / HW; }
    } else {
#pragma unroll
        for (int j = 0; j < VC; ++j) { m1[j] = ws[((size_t)n * C + c + j) * 2] / HW; m2[j] = ws[((size_t)n * C + c + j) * 2 + 1] / HW; }
    }
    const float gs = q.gscale ? q.gscale[n / q.group_n] : 1.f;
    const float* ztp = (q.zt && n >= q.zt_n0) ? q.zt + (size_t)(n - q.zt_n0) * HW * C + c : nullptr;
    T* op = static_cast<T*>(q.dzs) + (size_t)n * HW * q.lddz + c;
    float b[VC] = {0.f, 0.f, 0.f, 0.f};
    if (q.bias) ld4(q.bias + c, b);
    float sb[1][VC] = {{0.f, 0.f, 0.f, 0.f}};
    float sd = 0.f;
    int nsat = 0;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int p = p0 + ty + RGN * i;
        if (p >= HW) continue;
        float t[VC] = {0.f, 0.f, 0.f, 0.f}, o[VC];
        if (ztp) ld4(ztp + (size_t)p * C, t);
#pragma unroll
        for (int j = 0; j < VC; ++j) {
            const float xh = (zv[i][j] - mu[j]) * r[j];
            const float dz = r[j] * (dn[i][j] - m1[j] - xh * m2[j]) + t[j];
            sb[0][j] += dz; sd += dz * gs * (zv[i][j] - b[j]);
            o[j] = dz * gs;
        }
        nsat += sat_hits<T>(o);
        st4<T>(op + (size_t)p * q.lddz, o);
    }
    sat_commit(q.sat, nsat);
    if (q.dbias) {
        combine16<1>(sb, reinterpret_cast<float(*)[RGN][CW]>(sm), tx, ty);
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < VC; ++j) atomicAdd(q.dbias + replica_offset(q.nrep, q.rep_stride) + c + j, sb[0][j]);
        }
    }
    if (q.cdot) {
        const float tot = block_sum<CGN * RGN / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot + replica_offset(q.nrep, q.rep_stride) + n / q.group_n, tot);
    }
}

struct InDblParams {
    const float* gb_a; int ldgb;       // first-order chain gradient wrt activation output (dn = act' * gb_a), fp32
    const float* qz; int ldq;          // adjoint of the first-order dz (gt_z), fp32
    const void* gb_zs; int ldgz;       // first-order dz * isig (for the spectral-norm dot), nullable
    const void* z; int ldz;            // fp32 pre-norm tensor -- or (AZ kernel) the 16-bit LeakyReLU activation
    const float* mean; const float* rstd;
    void* gt_a; int ldga;              // out: act'(xhat) * rstd * (q - mean(q) - xhat mean(q xhat))
    float* zt;                         // out: adjoint wrt z, dense fp32 [N][HW][C]
    float* cdot;                       // scalar += sum gb_zs * q (atomic), nullable
    int HW, C, act;
    int q_nslab; long q_slab_stride;   // qz is the first of q_nslab split-K slabs of the producing conv
    unsigned* sat;                     // += number of gt_a values clipped by the fp16 store (nullable)
};

// adjoint of dz = in_bwd(xhat(z), rstd(z), dn) for incoming adjoint q (see oracle/manual_step.py:in_bwd_bwd)
template <typename T>
__global__ __launch_bounds__(CW * RG) void in_dbl_bwd_kernel(InDblParams q) {
    __shared__ float sm[5][RG][CW];
    __shared__ float red[CW * RG / 64];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, n = blockIdx.y;
    const int HW = q.HW, C = q.C;
    const float* zp = static_cast<const float*>(q.z) + (size_t)n * HW * q.ldz + c;
    const float* gp = q.gb_a + (size_t)n * HW * q.ldgb + c;
    const float* qp = q.qz + (size_t)n * HW * q.ldq + c;
    const T* gzp = q.gb_zs ? static_cast<const T*>(q.gb_zs) + (size_t)n * HW * q.ldgz + c : nullptr;
    const float mu = q.mean[(size_t)n * C + c], r = q.rstd[(size_t)n * C + c];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float sd = 0.f;
    int nsat = 0;
    for (int p = ty; p < HW; p += RG) {
        const float xh = (zp[(size_t)p * q.ldz] - mu) * r;
        const float dn = act_grad(xh, q.act) * gp[(size_t)p * q.ldgb];
        const float qq = qp[(size_t)p * q.ldq];
        s[0] += dn; s[1] += dn * xh; s[2] += qq; s[3] += qq * xh; s[4] += qq * dn;
        if (gzp) sd += Elem<T>::ld(gzp + (size_t)p * q.ldgz) * qq;
    }
    combine<5>(s, sm, tx, ty);
    const float inv = 1.f / HW;
    const float m1 = s[0] * inv, m2 = s[1] * inv, mq = s[2] * inv, mqx = s[3] * inv, mqd = s[4] * inv;
    const float k0 = mqd - mq * m1 - 3.f * mqx * m2;
    T* gap = static_cast<T*>(q.gt_a) + (size_t)n * HW * q.ldga + c;
    float* ztp = q.zt + (size_t)n * HW * C + c;
    for (int p = ty; p < HW; p += RG) {
        const float xh = (zp[(size_t)p * q.ldz] - mu) * r;
        const float ag = act_grad(xh, q.act);
        const float dn = ag * gp[(size_t)p * q.ldgb];
        const float qq = qp[(size_t)p * q.ldq];
        const float ga = ag * r * (qq - mq - xh * mqx);
        nsat += sat_hit<T>(ga);
        Elem<T>::st(gap + (size_t)p * q.ldga, ga);
        ztp[(size_t)p * C] = r * r * (-xh * k0 - m2 * (qq - mq) - mqx * (dn - m1));
    }
    sat_commit(q.sat, nsat);
    if (q.cdot) {
        const float tot = block_sum<CW * RG / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot, tot);
    }
}

// small maps, vectorised double backward (same math as in_dbl_bwd_kernel)
template <typename T, int RG, bool SLAB = false, bool AZ = false>
__global__ __launch_bounds__(CGN * RGN) void in_dbl_small_kernel(InDblParams q, int N, int spb) {
    // (branch-free buffer row accesses, as in in_bwd_small_kernel)
    __shared__ float sm[5][RGN][CW];
    __shared__ float red[CGN * RGN / 64];
    constexpr int SPP = RGN / RG, MR = MAXR;
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int slot = ty / RG, rg = ty % RG;
    const int c = blockIdx.x * CW + tx * VC;
    const int HW = q.HW, C = q.C;
    const size_t nhw = (size_t)N * HW;
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(q.z, rsrc_bytes(nhw * q.ldz * (AZ ? sizeof(T) : 4))), gr = make_rsrc(q.gb_a, rsrc_bytes(nhw * q.ldgb * 4)),
                                 qr = make_rsrc(q.qz, rsrc_bytes(nhw * q.ldq * 4)),
                                 gzr = make_rsrc(q.gb_zs, q.gb_zs ? rsrc_bytes(nhw * q.ldgz * sizeof(T)) : 0u),
                                 gar = make_rsrc(q.gt_a, rsrc_bytes(nhw * q.ldga * sizeof(T))), ztr = make_rsrc(q.zt, rsrc_bytes(nhw * C * 4));
    float sd = 0.f;
    int nsat = 0;
    const int nb = blockIdx.y * spb;
    for (int n0 = nb; n0 < min(N, nb + spb); n0 += SPP) {
        const int n = n0 + slot;
        const bool live = n < N;
        float mu[VC] = {0.f, 0.f, 0.f, 0.f}, r[VC] = {0.f, 0.f, 0.f, 0.f};
        if (live) { ld4(q.mean + (size_t)n * C + c, mu); ld4(q.rstd + (size_t)n * C + c, r); }
        if (SLAB && live)                                             // split-K slabs of the producing conv (see fold_slabs)
            fold_slabs<RG, MAXR>(const_cast<float*>(q.qz) + (size_t)n * HW * q.ldq + c, q.ldq, q.q_nslab, q.q_slab_stride, rg, HW);
        unsigned pix[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) { const int p = rg + RG * i; pix[i] = (live && p < HW) ? (unsigned)(n * HW + p) : OOB; }
        float xh[MR][VC], dn[MR][VC], qq[MR][VC];
        if constexpr (AZ) {                                           // the 16-bit activation instead of the fp32 z (see in_bwd_small_kernel)
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(zr, pix[i] != OOB ? (pix[i] * q.ldz + c) * (unsigned)sizeof(T) : OOB, 0, 0);
                xh[i][0] = Bits16<T>::dec(w[0]); xh[i][1] = Bits16<T>::dec(w[0] >> 16);
                xh[i][2] = Bits16<T>::dec(w[1]); xh[i][3] = Bits16<T>::dec(w[1] >> 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MR; ++i) bld4(zr, pix[i] != OOB ? (pix[i] * q.ldz + c) * 4u : OOB, xh[i]);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) bld4(gr, pix[i] != OOB ? (pix[i] * q.ldgb + c) * 4u : OOB, dn[i]);
#pragma unroll
        for (int i = 0; i < MR; ++i) bld4(qr, pix[i] != OOB ? (pix[i] * q.ldq + c) * 4u : OOB, qq[i]);
        if (q.gb_zs) {
            float gz[MR][VC];
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const unsigned off = pix[i] != OOB ? (pix[i] * q.ldgz + c) * (unsigned)sizeof(T) : OOB;
                if constexpr (sizeof(T) == 4) bld4(gzr, off, gz[i]);
                else {
                    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(gzr, off, 0, 0);
                    gz[i][0] = Bits16<T>::dec(w[0]); gz[i][1] = Bits16<T>::dec(w[0] >> 16);
                    gz[i][2] = Bits16<T>::dec(w[1]); gz[i][3] = Bits16<T>::dec(w[1] >> 16);
                }
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < VC; ++j) sd += gz[i][j] * qq[i][j];
        }
        float s[5][VC];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) s[i][j] = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                if constexpr (AZ) xh[i][j] = xh[i][j] > 0.f ? xh[i][j] : 5.0f * xh[i][j];       // (dead rows read 0)
                else xh[i][j] = pix[i] != OOB ? (xh[i][j] - mu[j]) * r[j] : 0.f;
                dn[i][j] = act_grad(xh[i][j], q.act) * dn[i][j];              // (dead rows: gb_a = q = 0)
                s[0][j] += dn[i][j]; s[1][j] += dn[i][j] * xh[i][j]; s[2][j] += qq[i][j];
                s[3][j] += qq[i][j] * xh[i][j]; s[4][j] += qq[i][j] * dn[i][j];
            }
        combine_seg<5, RG>(s, sm, tx, ty);
        const float inv = 1.f / HW;
        float m1[VC], m2[VC], mq[VC], mqx[VC], mqd[VC];
#pragma unroll
        for (int j = 0; j < VC; ++j) { m1[j] = s[0][j] * inv; m2[j] = s[1][j] * inv; mq[j] = s[2][j] * inv; mqx[j] = s[3][j] * inv; mqd[j] = s[4][j] * inv; }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            float o[VC], zt[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                o[j] = act_grad(xh[i][j], q.act) * r[j] * (qq[i][j] - mq[j] - xh[i][j] * mqx[j]);
                zt[j] = r[j] * r[j] * (-xh[i][j] * (mqd[j] - mq[j] * m1[j] - 3.f * mqx[j] * m2[j]) - m2[j] * (qq[i][j] - mq[j]) - mqx[j] * (dn[i][j] - m1[j]));
            }
            nsat += sat_hits<T>(o);
            bst4<T>(gar, pix[i] != OOB ? (pix[i] * q.ldga + c) * (unsigned)sizeof(T) : OOB, o);
            bst4<float>(ztr, pix[i] != OOB ? (pix[i] * C + c) * 4u : OOB, zt);
        }
    }
    if (q.cdot) {
        const float tot = block_sum<CGN * RGN / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot, tot);
    }
    sat_commit(q.sat, nsat);
}

// ---- layers without a norm (D.c1, G.down1): a = lrelu(z) was fused in the conv epilogue; backward is elementwise
struct ActBwdParams {
    const float* da; int ldda; const float* da2; int ldda2;      // incoming gradients, fp32
    const void* a; int lda;
    const float* gscale; int group_n;
    const float* bias;
    void* dzs; int lddz;
    float* dbias; float* cdot;
    int HW, C;
    int nrep, rep_stride;
    unsigned* sat;                     // += number of dzs values clipped by the fp16 store (nullable)
    const void* dotx; int lddot; float* dot_out;   // optional: dot_out += sum dotx * da (dotx in the compute dtype): gcssl_dot_accum
                                                   // folded in for the reverse GP chain's norm-less layer (<gb_zs, gt_z>)
};
// grid: (C/64, sample blocks of `spb`, H*W chunks of `rows_per_chunk`); lanes own 4 channels x strided rows, the bias /
// spectral-norm partial sums stay in registers across the samples of a workgroup (one set of atomics per workgroup)
template <typename T>
__global__ __launch_bounds__(CGN * RGN) void act_bwd_kernel(ActBwdParams q, int N, int spb, int rows_per_chunk, int mixed_groups) {
    __shared__ float sm[1][RGN][CW];
    __shared__ float red[CGN * RGN / 64];
    const int tx = threadIdx.x % CGN, ty = threadIdx.x / CGN;
    const int c = blockIdx.x * CW + tx * VC;
    const int HW = q.HW;
    float b[VC] = {0.f, 0.f, 0.f, 0.f};
    if (q.bias) ld4(q.bias + c, b);
    float sb[1][VC] = {{0.f, 0.f, 0.f, 0.f}};
    float sd = 0.f, sdot = 0.f;
    int nsat = 0;
    const int p0 = blockIdx.z * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
    const int nb = blockIdx.y * spb;
    for (int n = nb; n < min(N, nb + spb); ++n) {
        const T* ap = static_cast<const T*>(q.a) + (size_t)n * HW * q.lda + c;
        const float* dap = q.da + (size_t)n * HW * q.ldda + c;
        const float* da2p = q.da2 ? q.da2 + (size_t)n * HW * q.ldda2 + c : nullptr;
        T* op = static_cast<T*>(q.dzs) + (size_t)n * HW * q.lddz + c;
        const T* xp = q.dotx ? static_cast<const T*>(q.dotx) + (size_t)n * HW * q.lddot + c : nullptr;
        const float gs = q.gscale ? q.gscale[n / q.group_n] : 1.f;
#pragma unroll 4
        for (int p = p0 + ty; p < p1; p += RGN) {           // unrolled: four rows' loads in flight per lane
            float av[VC], d[VC], o[VC];
            ldT4<T>(ap + (size_t)p * q.lda, av);
            ld4(dap + (size_t)p * q.ldda, d);
            if (xp) {
                float xv[VC]; ldT4<T>(xp + (size_t)p * q.lddot, xv);
                sdot += (xv[0] * d[0] + xv[1] * d[1]) + (xv[2] * d[2] + xv[3] * d[3]);
            }
            if (da2p) {
                float t[VC]; ld4(da2p + (size_t)p * q.ldda2, t);
#pragma unroll
                for (int j = 0; j < VC; ++j) d[j] += t[j];
            }
#pragma unroll
            for (int j = 0; j < VC; ++j) {
                const float dz = av[j] > 0.f ? d[j] : 0.2f * d[j];
                const float zv = av[j] > 0.f ? av[j] : av[j] * 5.0f;          // invert LeakyReLU(0.2)
                sb[0][j] += dz; sd += dz * gs * (zv - b[j]);
                o[j] = dz * gs;
            }
            nsat += sat_hits<T>(o);
            st4<T>(op + (size_t)p * q.lddz, o);
        }
        if (mixed_groups && q.cdot) {
            const float tot = block_sum<CGN * RGN / 64>(sd, red);
            if (threadIdx.x == 0) atomicAdd(q.cdot + replica_offset(q.nrep, q.rep_stride) + n / q.group_n, tot);
            sd = 0.f;
        }
    }
    if (q.dbias) {
        combine16<1>(sb, sm, tx, ty);
        if (ty == 0) {
#pragma unroll
            for (int j = 0; j < VC; ++j) atomicAdd(q.dbias + replica_offset(q.nrep, q.rep_stride) + c + j, sb[0][j]);
        }
    }
    if (q.cdot && !mixed_groups) {
        const float tot = block_sum<CGN * RGN / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot + replica_offset(q.nrep, q.rep_stride) + nb / q.group_n, tot);
    }
    sat_commit(q.sat, nsat);
    if (q.dot_out) {
        const float tot = block_sum<CGN * RGN / 64>(sdot, red);
        if (threadIdx.x == 0) atomicAdd(q.dot_out, tot);
    }
}

// out += sum x*y  (strided NHWC views), used for the <gb_zs, gt_z> spectral-norm term of the norm-less layer;
// 4 channels per lane and step (8-byte bf16 / 16-byte fp32 loads), C % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void dot_accum_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                       size_t pixels, int C, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    const int c4 = C / 4;
    const size_t total = pixels * c4;
    if (total < 0x7fffffffull && (c4 & (c4 - 1)) == 0) {     // 32-bit indices, shift / mask instead of a 64-bit division per element
        const unsigned tot32 = (unsigned)total, lg = 31 - __builtin_clz((unsigned)c4), stride = gridDim.x * 256u;
#pragma unroll 4
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < tot32; i += stride) {
            const unsigned pix = i >> lg, c = (i & (unsigned)(c4 - 1)) * 4u;
            float xv[VC], yv[VC];
            ldT4<T>(x + (size_t)pix * ldx + c, xv); ld4(y + (size_t)pix * ldy + c, yv);
            s += xv[0] * yv[0] + xv[1] * yv[1] + xv[2] * yv[2] + xv[3] * yv[3];
        }
    } else {
#pragma unroll 4
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t pix = i / c4; const int c = (int)(i % c4) * 4;
            float xv[VC], yv[VC];
            ldT4<T>(x + pix * ldx + c, xv); ld4(y + pix * ldy + c, yv);
            s += xv[0] * yv[0] + xv[1] * yv[1] + xv[2] * yv[2] + xv[3] * yv[3];
        }
    }
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) atomicAdd(out, tot);
}

// Same-address float atomics serialise (~12 ns each, and a 256-byte bias vector shares a few memory-side lines): 768
// workgroups adding 64 bias sums each cost 70 us on a 17-us pass (tools/archive/actbwd_bench.py).  The sums therefore go to one of
// nrep replicas chosen by workgroup index; gcssl_sum_replicas folds them afterwards.
bool bad_dtype(int dt) { return gcssl_bad_dtype(dt); }

// row-group lanes per sample for a small map, and samples per workgroup (a multiple of the samples per pass that keeps
// >= ~256 workgroups and, when per-group sums are accumulated, never straddles a sample group)
// the small-map kernels address rows with 32-bit byte offsets into buffer descriptors
bool fits_buffer(size_t n, size_t hw, int ld_max) { return n * hw * (size_t)ld_max * 4 < 0x80000000ull; }
int small_rg(int HW) { return HW <= 4 ? 1 : (HW <= 16 ? 4 : 16); }
int small_spb(int N, int C, int rg, int group_n) {
    // streaming passes want several workgroups per CU in flight (256 -> 1024: +2 % on the whole iteration); the sums they
    // add atomically are striped over replicas, so more workgroups no longer means more contention
    static const long min_wgs = [] { const char* e = getenv("GCSSL_IN_WGS"); return e ? atol(e) : 1024L; }();
    const int spp = RGN / rg;
    int spb = spp;
    while (spb * 2 <= 64 && (long)(C / CW) * ((N + spb * 2 - 1) / (spb * 2)) >= min_wgs &&
           (group_n <= 0 || group_n % (spb * 2) == 0)) spb *= 2;
    return spb;
}

}  // namespace

extern "C" {

// the LDS-resident forward needs the 128-KB dynamic-LDS opt-in; done once, outside any stream capture (gcssl_init)
int gcssl_init_norm() {
    const int bytes = (int)(LDS_HW_MAX * 32 * sizeof(float));
    hipError_t e = hipSuccess;
#define OPT_IN(T) do { \
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(in_fwd_lds_kernel<T, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(in_fwd_lds_kernel<T, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes / 2); } while (0)
    OPT_IN(bf16_t); OPT_IN(f16_t); OPT_IN(float);
#undef OPT_IN
    return e == hipSuccess ? GCSSL_OK : (int)e;
}

int gcssl_in_act_fwd(int dtype, float* z, int ldz, void* a, int lda, float* mean, float* rstd,
                     const uint8_t* mask, float* pool, int nslab, long slab_stride, int N, int HW, int C, int act, void* stream) {
    if (!z || !mean || !rstd) return GCSSL_ENULL;
    // a == NULL: statistics + pool sums only (the generator's last activation feeds nothing but its global average pool) --
    // served by the LDS-resident form, where it halves the launch's HBM traffic
    if (!a && !(pool && HW > MID_HW && HW <= LDS_HW_MAX && C % 32 == 0 && nslab == 1)) return GCSSL_ENULL;
    if (nslab < 1 || (nslab > 1 && (slab_stride <= 0 || slab_stride % 4 || HW > MID_HW || pool))) return GCSSL_EBADSHAPE;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || (a && (lda < C || lda % 4)) || ldz % 4 || (act != 1 && act != 2)) return GCSSL_EBADSHAPE;
    if (!aligned16(z) || (((uintptr_t)a) & 7)) return GCSSL_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (HW <= MID_HW && !pool) {
        if (!fits_buffer(N, HW, ldz > lda ? ldz : lda)) return GCSSL_EBADSHAPE;
        const int rg = small_rg(HW), spb = small_spb(N, C, rg, 0);
        dim3 grid(C / CW, (N + spb - 1) / spb);
#define FWD_SMALL(T, RG, MR) do { if (nslab > 1) hipLaunchKernelGGL((in_fwd_small_kernel<T, RG, MR, true>), grid, dim3(CGN * RGN), 0, st, z, ldz, (T*)a, lda, mean, rstd, mask, N, HW, C, act, spb, nslab, slab_stride); \
                                  else hipLaunchKernelGGL((in_fwd_small_kernel<T, RG, MR, false>), grid, dim3(CGN * RGN), 0, st, z, ldz, (T*)a, lda, mean, rstd, mask, N, HW, C, act, spb, nslab, slab_stride); } while (0)
        GCSSL_DISPATCH(dtype,
            if (rg == 1) FWD_SMALL(T, 1, MAXR); else if (rg == 4) FWD_SMALL(T, 4, MAXR);
            else if (HW <= SMALL_HW) FWD_SMALL(T, 16, MAXR); else FWD_SMALL(T, 16, BIGR));
#undef FWD_SMALL
        return gcssl_launch_status();
    }
    static const int lds_ch = [] { const char* e = getenv("GCSSL_IN_LDS_CH"); return (e && atoi(e) == 16) ? 16 : 32; }();
    if (HW <= LDS_HW_MAX && C % lds_ch == 0 && nslab == 1) {
        const size_t lds = (size_t)HW * lds_ch * sizeof(float);
        static bool inited = false;
        if (!inited) { int rc = gcssl_init_norm(); if (rc) return rc; inited = true; }
        dim3 g2(C / lds_ch, N);
#define LDS_FWD(T, CHN) hipLaunchKernelGGL((in_fwd_lds_kernel<T, CHN>), g2, dim3(256), lds, st, z, ldz, (T*)a, lda, mean, rstd, mask, pool, HW, C, act)
        GCSSL_DISPATCH(dtype, if (lds_ch == 16) LDS_FWD(T, 16); else LDS_FWD(T, 32));
#undef LDS_FWD
        return gcssl_launch_status();
    }
    dim3 grid(C / CW, N, (HW + SMALL_HW - 1) / SMALL_HW);
    if (rstd == mean + (size_t)N * C) {                      // back-to-back buffers (the engine's): one fill
        gcssl_zero_async(mean, 2 * (size_t)N * C, st);        // (kernels, not memset nodes: common.h)
    } else {
        gcssl_zero_async(mean, (size_t)N * C, st);
        gcssl_zero_async(rstd, (size_t)N * C, st);
    }
    hipLaunchKernelGGL(in_stats_kernel, grid, dim3(CGN * RGN), 0, st, z, ldz, mean, rstd, HW, C);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((unsigned)(((size_t)N * C + 255) / 256)), dim3(256), 0, st, z, ldz, mean, rstd, N, HW, C);
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(in_apply_kernel<T>, grid, dim3(CGN * RGN), 0, st, z, ldz, (T*)a, lda, mean, rstd, mask, pool, HW, C, act));
    return gcssl_launch_status();
}

int gcssl_in_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const float* da_bcast,
                     const void* z, int ldz, int z_kind, const float* mean, const float* rstd, const uint8_t* mask,
                     const float* zt, int zt_n0, const float* gscale, int group_n, const float* bias,
                     void* dzs, int lddz, float* dbias, float* cdot, int nrep, int rep_stride, int da_nslab, long da_slab_stride,
                     float* ws, const float* presum_cnt, const float* presum_pos, float presum_pos_scale, unsigned* sat,
                     int N, int HW, int C, int act, void* stream) {
    if ((!da && !da_bcast) || !z || !mean || !rstd || !dzs) return GCSSL_ENULL;
    if (nrep < 1 || (nrep > 1 && rep_stride < C)) return GCSSL_EBADSHAPE;
    if (da_nslab < 1 || (da_nslab > 1 && (!da || da_slab_stride <= 0 || da_slab_stride % 4 || HW > MID_HW))) return GCSSL_EBADSHAPE;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || lddz < C || ldz % 4 || lddz % 4 || (act != 1 && act != 2)) return GCSSL_EBADSHAPE;
    if ((da && ldda % 4) || (da2 && ldda2 % 4)) return GCSSL_EBADSHAPE;
    if ((gscale || cdot) && group_n <= 0) return GCSSL_EBADSHAPE;
    if (z_kind != 0 && z_kind != 1) return GCSSL_EBADSHAPE;
    if (z_kind == 1 && (dtype == GCSSL_F32 || HW > MID_HW || act != 1 || (((uintptr_t)z) & 7))) return GCSSL_EBADSHAPE;   // the activation form: 16-bit, small maps, LeakyReLU
    if (HW > MID_HW && !ws && !(presum_cnt && presum_pos)) return GCSSL_ENULL;
    InBwdParams q{da, ldda, da2, ldda2, da_bcast, z, ldz, mean, rstd, mask, zt, zt_n0, gscale,
                  group_n > 0 ? group_n : N, bias, dzs, lddz, dbias, cdot, nrep, rep_stride, da_nslab, da_slab_stride, HW, C, act};
    q.sat = sat;
    hipStream_t st = (hipStream_t)stream;
    if (HW <= MID_HW) {
        if (!fits_buffer(N, HW, std::max(std::max(ldz, lddz), std::max(da ? ldda : 0, da2 ? ldda2 : 0)))) return GCSSL_EBADSHAPE;
        const int rg = small_rg(HW), spb = small_spb(N, C, rg, cdot ? q.group_n : 0);
        const int mixed = (cdot && q.group_n % spb) ? 1 : 0;
        dim3 grid(C / CW, (N + spb - 1) / spb);
#define BWD_SMALL(T, RG, MR) do { if (z_kind == 1) { if constexpr (sizeof(T) == 2) { \
                                      if (da_nslab > 1) hipLaunchKernelGGL((in_bwd_small_kernel<T, RG, MR, true, true>), grid, dim3(CGN * RGN), 0, st, q, N, spb, mixed); \
                                      else hipLaunchKernelGGL((in_bwd_small_kernel<T, RG, MR, false, true>), grid, dim3(CGN * RGN), 0, st, q, N, spb, mixed); } } \
                                  else if (da_nslab > 1) hipLaunchKernelGGL((in_bwd_small_kernel<T, RG, MR, true>), grid, dim3(CGN * RGN), 0, st, q, N, spb, mixed); \
                                  else hipLaunchKernelGGL((in_bwd_small_kernel<T, RG, MR, false>), grid, dim3(CGN * RGN), 0, st, q, N, spb, mixed); } while (0)
        GCSSL_DISPATCH(dtype,
            if (rg == 1) BWD_SMALL(T, 1, MAXR); else if (rg == 4) BWD_SMALL(T, 4, MAXR);
            else if (HW <= SMALL_HW) BWD_SMALL(T, 16, MAXR); else BWD_SMALL(T, 16, BIGR));
#undef BWD_SMALL
        return gcssl_launch_status();
    }
    dim3 grid(C / CW, N, (HW + SMALL_HW - 1) / SMALL_HW);
    if (presum_cnt && presum_pos) {                          // statistics known from the forward pass: the apply pass alone
        if (da || da2 || mask || zt || !da_bcast || act != 2) return GCSSL_EBADSHAPE;
        q.pre_cnt = presum_cnt; q.pre_pos = presum_pos; q.pre_pos_scale = presum_pos_scale;
        GCSSL_DISPATCH(dtype, hipLaunchKernelGGL((in_bwd_kernel<T, 2>), grid, dim3(CGN * RGN), 0, st, q, ws));
        return gcssl_launch_status();
    }
    gcssl_zero_async(ws, 2 * (size_t)N * C, st);             // (a kernel, not a memset node: common.h)
    GCSSL_DISPATCH(dtype,
        hipLaunchKernelGGL((in_bwd_kernel<T, 1>), grid, dim3(CGN * RGN), 0, st, q, ws);
        hipLaunchKernelGGL((in_bwd_kernel<T, 2>), grid, dim3(CGN * RGN), 0, st, q, ws));
    return gcssl_launch_status();
}

int gcssl_in_dbl_bwd(int dtype, const float* gb_a, int ldgb, const float* qz, int ldq, const void* gb_zs, int ldgz,
                     const void* z, int ldz, int z_kind, const float* mean, const float* rstd, void* gt_a, int ldga,
                     float* zt, float* cdot, int q_nslab, long q_slab_stride, unsigned* sat, int N, int HW, int C, int act,
                     void* stream) {
    if (!gb_a || !qz || !z || !mean || !rstd || !gt_a || !zt) return GCSSL_ENULL;
    if (q_nslab < 1 || (q_nslab > 1 && (q_slab_stride <= 0 || q_slab_stride % 4 || HW > SMALL_HW))) return GCSSL_EBADSHAPE;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || ldgb < C || ldq < C || ldga < C) return GCSSL_EBADSHAPE;
    InDblParams q{gb_a, ldgb, qz, ldq, gb_zs, ldgz, z, ldz, mean, rstd, gt_a, ldga, zt, cdot, HW, C, act, q_nslab, q_slab_stride, sat};
    const bool small = HW <= SMALL_HW && !(ldgb % 4) && !(ldq % 4) && !(ldz % 4) && !(ldga % 4) && (!gb_zs || !(ldgz % 4));
    if (q_nslab > 1 && !small) return GCSSL_EBADSHAPE;      // slabs are summed by the fused small-map kernel only
    if (z_kind != 0 && z_kind != 1) return GCSSL_EBADSHAPE;
    if (z_kind == 1 && (!small || dtype == GCSSL_F32 || act != 1 || (((uintptr_t)z) & 7))) return GCSSL_EBADSHAPE;
    if (small) {
        if (!fits_buffer(N, HW, std::max(std::max(ldz, ldgb), std::max(std::max(ldq, ldga), std::max(ldgz, C))))) return GCSSL_EBADSHAPE;
        const int rg = small_rg(HW), spb = small_spb(N, C, rg, 0);
        dim3 sgrid(C / CW, (N + spb - 1) / spb);
        hipStream_t st = (hipStream_t)stream;
#define DBL_SMALL(T, RG) do { if (z_kind == 1) { if constexpr (sizeof(T) == 2) { \
                                  if (q_nslab > 1) hipLaunchKernelGGL((in_dbl_small_kernel<T, RG, true, true>), sgrid, dim3(CGN * RGN), 0, st, q, N, spb); \
                                  else hipLaunchKernelGGL((in_dbl_small_kernel<T, RG, false, true>), sgrid, dim3(CGN * RGN), 0, st, q, N, spb); } } \
                              else if (q_nslab > 1) hipLaunchKernelGGL((in_dbl_small_kernel<T, RG, true>), sgrid, dim3(CGN * RGN), 0, st, q, N, spb); \
                              else hipLaunchKernelGGL((in_dbl_small_kernel<T, RG, false>), sgrid, dim3(CGN * RGN), 0, st, q, N, spb); } while (0)
        GCSSL_DISPATCH(dtype, if (rg == 1) DBL_SMALL(T, 1); else if (rg == 4) DBL_SMALL(T, 4); else DBL_SMALL(T, 16));
#undef DBL_SMALL
        return gcssl_launch_status();
    }
    dim3 grid(C / CW, N);
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(in_dbl_bwd_kernel<T>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q));
    return gcssl_launch_status();
}

int gcssl_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const void* a, int lda,
                  const float* gscale, int group_n, const float* bias, void* dzs, int lddz, float* dbias,
                  float* cdot, int nrep, int rep_stride, unsigned* sat, const void* dotx, int lddot, float* dot_out,
                  int N, int HW, int C, void* stream) {
    if (!da || !a || !dzs) return GCSSL_ENULL;
    if ((dotx != nullptr) != (dot_out != nullptr) || (dotx && (lddot < C || lddot % 4 || da2))) return GCSSL_EBADSHAPE;
    if (nrep < 1 || (nrep > 1 && rep_stride < C)) return GCSSL_EBADSHAPE;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW <= 0 || C <= 0 || C % CW || lda < C || lddz < C || ldda < C) return GCSSL_EBADSHAPE;
    if (lda % 4 || lddz % 4 || ldda % 4 || (da2 && ldda2 % 4)) return GCSSL_EBADSHAPE;
    if ((gscale || cdot) && group_n <= 0) return GCSSL_EBADSHAPE;
    ActBwdParams q{da, ldda, da2, ldda2, a, lda, gscale, group_n > 0 ? group_n : N, bias, dzs, lddz, dbias, cdot, HW, C, nrep, rep_stride, sat, dotx, lddot, dot_out};
    // 64-row chunks of H*W x blocks of samples; grow the sample block while >= ~512 workgroups remain
    int rows = HW < 64 ? HW : 64;
    const int zc = (HW + rows - 1) / rows;
    int spb = 1;
    static const long act_wgs = [] { const char* e = getenv("GCSSL_ACT_WGS"); return e ? atol(e) : 512L; }();
    while (spb * 2 <= 32 && (long)(C / CW) * zc * ((N + spb * 2 - 1) / (spb * 2)) >= act_wgs &&
           (!cdot || q.group_n % (spb * 2) == 0)) spb *= 2;
    const int mixed = (cdot && q.group_n % spb) ? 1 : 0;
    dim3 grid(C / CW, (N + spb - 1) / spb, zc);
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, grid, dim3(CGN * RGN), 0, (hipStream_t)stream, q, N, spb, rows, mixed));
    return gcssl_launch_status();
}

int gcssl_dot_accum(int dtype, const void* x, int ldx, const float* y, int ldy, long pixels, int C, float* out,
                    void* stream) {
    if (!x || !y || !out) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (pixels <= 0 || C <= 0 || ldx < C || ldy < C) return GCSSL_EBADSHAPE;
    const size_t total = (size_t)pixels * C;
    if (C % 4 || ldx % 4 || ldy % 4) return GCSSL_EBADSHAPE;
    // one same-address atomic per block (~12 ns each, serialised): few, long blocks
    int blocks = (int)((total / 4 + 1023) / 1024); if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(dot_accum_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx,
                                             y, ldy, (size_t)pixels, C, out));
    return gcssl_launch_status();
}

}  // extern "C"
