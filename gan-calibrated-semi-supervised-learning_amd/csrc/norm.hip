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

// a = act((z - mean) * rstd) [* keep * 2];  writes mean/rstd [N][C]
template <typename T>
__global__ __launch_bounds__(CW * RG) void in_act_fwd_kernel(const float* __restrict__ z, int ldz, T* __restrict__ a, int lda,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            const uint8_t* __restrict__ mask, int HW, int C, int act) {
    __shared__ float sm[1][RG][CW];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, n = blockIdx.y;
    const float* zp = z + (size_t)n * HW * ldz + c;
    float s[1] = {0.f};
    for (int p = ty; p < HW; p += RG) s[0] += zp[(size_t)p * ldz];
    combine<1>(s, sm, tx, ty);
    const float mu = s[0] / HW;
    s[0] = 0.f;
    for (int p = ty; p < HW; p += RG) { const float d = zp[(size_t)p * ldz] - mu; s[0] += d * d; }
    combine<1>(s, sm, tx, ty);
    const float r = 1.0f / sqrtf(s[0] / HW + IN_EPS);
    if (ty == 0) { mean[(size_t)n * C + c] = mu; rstd[(size_t)n * C + c] = r; }
    T* ap = a + (size_t)n * HW * lda + c;
    const uint8_t* mp = mask ? mask + (size_t)n * HW * C + c : nullptr;
    for (int p = ty; p < HW; p += RG) {
        float v = act_fwd((zp[(size_t)p * ldz] - mu) * r, act);
        if (mp) v *= mp[(size_t)p * C] ? 2.f : 0.f;
        Elem<T>::st(ap + (size_t)p * lda, v);
    }
}

struct InBwdParams {
    const float* da; int ldda;         // incoming gradient of the activation output, fp32 (nullable if da_bcast)
    const float* da2; int ldda2;       // optional second gradient added to da (skip connection), fp32
    const float* da_bcast;             // optional [N][C] gradient broadcast over H*W (global-avg-pool backward)
    const float* z; int ldz;           // pre-norm conv output, always fp32
    const float* mean; const float* rstd;
    const uint8_t* mask;               // dropout keep mask [N][HW][C], nullable
    const float* zt; int zt_n0;        // optional fp32 double-backward term added to dz of samples n >= zt_n0 ([N-zt_n0][HW][C])
    const float* gscale; int group_n;  // output multiplier per sample group, nullable
    const float* bias;                 // conv bias (for the spectral-norm <dz, z-b> term), nullable
    void* dzs; int lddz;               // output: dz * gscale
    float* dbias;                      // [C] += sum dz (atomic), nullable
    float* cdot;                       // [ngroups] += sum dzs (z - bias) = <G_k, W_orig>/sigma_k^2 (atomic), nullable
    int HW, C, act;
};

// dn = act'(xhat) (da + da2) [*2 keep];  dz = rstd (dn - mean(dn) - xhat mean(dn xhat)) [+ zt]
template <typename T>
__global__ __launch_bounds__(CW * RG) void in_act_bwd_kernel(InBwdParams q) {
    __shared__ float sm[2][RG][CW];
    __shared__ float red[CW * RG / 64];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, n = blockIdx.y;
    const int HW = q.HW, C = q.C;
    const float* zp = q.z + (size_t)n * HW * q.ldz + c;
    const float* dap = q.da ? q.da + (size_t)n * HW * q.ldda + c : nullptr;
    const float* da2p = q.da2 ? q.da2 + (size_t)n * HW * q.ldda2 + c : nullptr;
    const uint8_t* mp = q.mask ? q.mask + (size_t)n * HW * C + c : nullptr;
    const float mu = q.mean[(size_t)n * C + c], r = q.rstd[(size_t)n * C + c];
    const float dab = q.da_bcast ? q.da_bcast[(size_t)n * C + c] : 0.f;
    auto dn_at = [&](int p, float xh) {
        float d = dab;
        if (dap) d += dap[(size_t)p * q.ldda];
        if (da2p) d += da2p[(size_t)p * q.ldda2];
        if (mp) d *= mp[(size_t)p * C] ? 2.f : 0.f;
        return d * act_grad(xh, q.act);
    };
    float s[2] = {0.f, 0.f};
    for (int p = ty; p < HW; p += RG) {
        const float xh = (zp[(size_t)p * q.ldz] - mu) * r;
        const float dn = dn_at(p, xh);
        s[0] += dn; s[1] += dn * xh;
    }
    combine<2>(s, sm, tx, ty);
    const float m1 = s[0] / HW, m2 = s[1] / HW;
    const float gs = q.gscale ? q.gscale[n / q.group_n] : 1.f;
    const float* ztp = (q.zt && n >= q.zt_n0) ? q.zt + (size_t)(n - q.zt_n0) * HW * C + c : nullptr;
    T* op = static_cast<T*>(q.dzs) + (size_t)n * HW * q.lddz + c;
    const float b = q.bias ? q.bias[c] : 0.f;
    float sb = 0.f, sd = 0.f;
    for (int p = ty; p < HW; p += RG) {
        const float zv = zp[(size_t)p * q.ldz];
        const float xh = (zv - mu) * r;
        float dz = r * (dn_at(p, xh) - m1 - xh * m2);
        if (ztp) dz += ztp[(size_t)p * C];
        sb += dz; sd += dz * gs * (zv - b);
        Elem<T>::st(op + (size_t)p * q.lddz, dz * gs);
    }
    if (q.dbias) {
        float v[1] = {sb};
        combine<1>(v, reinterpret_cast<float(*)[RG][CW]>(sm), tx, ty);
        if (ty == 0) atomicAdd(q.dbias + c, v[0]);
    }
    if (q.cdot) {
        const float tot = block_sum<CW * RG / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot + n / q.group_n, tot);
    }
}

struct InDblParams {
    const float* gb_a; int ldgb;       // first-order chain gradient wrt activation output (dn = act' * gb_a), fp32
    const float* qz; int ldq;          // adjoint of the first-order dz (gt_z), fp32
    const void* gb_zs; int ldgz;       // first-order dz * isig (for the spectral-norm dot), nullable
    const float* z; int ldz;           // always fp32
    const float* mean; const float* rstd;
    void* gt_a; int ldga;              // out: act'(xhat) * rstd * (q - mean(q) - xhat mean(q xhat))
    float* zt;                         // out: adjoint wrt z, dense fp32 [N][HW][C]
    float* cdot;                       // scalar += sum gb_zs * q (atomic), nullable
    int HW, C, act;
};

// adjoint of dz = in_bwd(xhat(z), rstd(z), dn) for incoming adjoint q (see oracle/manual_step.py:in_bwd_bwd)
template <typename T>
__global__ __launch_bounds__(CW * RG) void in_dbl_bwd_kernel(InDblParams q) {
    __shared__ float sm[5][RG][CW];
    __shared__ float red[CW * RG / 64];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, n = blockIdx.y;
    const int HW = q.HW, C = q.C;
    const float* zp = q.z + (size_t)n * HW * q.ldz + c;
    const float* gp = q.gb_a + (size_t)n * HW * q.ldgb + c;
    const float* qp = q.qz + (size_t)n * HW * q.ldq + c;
    const T* gzp = q.gb_zs ? static_cast<const T*>(q.gb_zs) + (size_t)n * HW * q.ldgz + c : nullptr;
    const float mu = q.mean[(size_t)n * C + c], r = q.rstd[(size_t)n * C + c];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float sd = 0.f;
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
        Elem<T>::st(gap + (size_t)p * q.ldga, ag * r * (qq - mq - xh * mqx));
        ztp[(size_t)p * C] = r * r * (-xh * k0 - m2 * (qq - mq) - mqx * (dn - m1));
    }
    if (q.cdot) {
        const float tot = block_sum<CW * RG / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot, tot);
    }
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
};
template <typename T>
__global__ __launch_bounds__(CW * RG) void act_bwd_kernel(ActBwdParams q) {
    __shared__ float sm[1][RG][CW];
    __shared__ float red[CW * RG / 64];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, n = blockIdx.y;
    const int HW = q.HW;
    const T* ap = static_cast<const T*>(q.a) + (size_t)n * HW * q.lda + c;
    const float* dap = q.da + (size_t)n * HW * q.ldda + c;
    const float* da2p = q.da2 ? q.da2 + (size_t)n * HW * q.ldda2 + c : nullptr;
    T* op = static_cast<T*>(q.dzs) + (size_t)n * HW * q.lddz + c;
    const float gs = q.gscale ? q.gscale[n / q.group_n] : 1.f;
    const float b = q.bias ? q.bias[c] : 0.f;
    float sb = 0.f, sd = 0.f;
    // HW is split over blockIdx.z chunks to keep enough workgroups in flight on big maps
    const int chunk = (HW + gridDim.z - 1) / gridDim.z;
    const int p0 = blockIdx.z * chunk, p1 = min(HW, p0 + chunk);
    for (int p = p0 + ty; p < p1; p += RG) {
        const float av = Elem<T>::ld(ap + (size_t)p * q.lda);
        float d = dap[(size_t)p * q.ldda];
        if (da2p) d += da2p[(size_t)p * q.ldda2];
        const float dz = av > 0.f ? d : 0.2f * d;
        const float zv = av > 0.f ? av : av * 5.0f;          // invert LeakyReLU(0.2)
        sb += dz; sd += dz * gs * (zv - b);
        Elem<T>::st(op + (size_t)p * q.lddz, dz * gs);
    }
    if (q.dbias) {
        float v[1] = {sb};
        combine<1>(v, sm, tx, ty);
        if (ty == 0) atomicAdd(q.dbias + c, v[0]);
    }
    if (q.cdot) {
        const float tot = block_sum<CW * RG / 64>(sd, red);
        if (threadIdx.x == 0) atomicAdd(q.cdot + n / q.group_n, tot);
    }
}

// out += sum x*y  (strided NHWC views), used for the <gb_zs, gt_z> spectral-norm term of the norm-less layer
template <typename T>
__global__ void dot_accum_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                 size_t pixels, int C, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t total = pixels * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C; const int c = i % C;
        s += Elem<T>::ld(x + pix * ldx + c) * y[pix * ldy + c];
    }
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) atomicAdd(out, tot);
}

bool bad_dtype(int dt) { return dt != GCSSL_F32 && dt != GCSSL_BF16; }

}  // namespace

extern "C" {

int gcssl_in_act_fwd(int dtype, const float* z, int ldz, void* a, int lda, float* mean, float* rstd,
                     const uint8_t* mask, int N, int HW, int C, int act, void* stream) {
    if (!z || !a || !mean || !rstd) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || lda < C || (act != 1 && act != 2)) return GCSSL_EBADSHAPE;
    dim3 grid(C / CW, N);
    if (dtype == GCSSL_F32)
        hipLaunchKernelGGL(in_act_fwd_kernel<float>, grid, dim3(CW * RG), 0, (hipStream_t)stream, z, ldz,
                           (float*)a, lda, mean, rstd, mask, HW, C, act);
    else
        hipLaunchKernelGGL(in_act_fwd_kernel<bf16_t>, grid, dim3(CW * RG), 0, (hipStream_t)stream, z, ldz,
                           (bf16_t*)a, lda, mean, rstd, mask, HW, C, act);
    return gcssl_launch_status();
}

int gcssl_in_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const float* da_bcast,
                     const float* z, int ldz, const float* mean, const float* rstd, const uint8_t* mask,
                     const float* zt, int zt_n0, const float* gscale, int group_n, const float* bias,
                     void* dzs, int lddz, float* dbias, float* cdot, int N, int HW, int C, int act, void* stream) {
    if ((!da && !da_bcast) || !z || !mean || !rstd || !dzs) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || lddz < C || (act != 1 && act != 2)) return GCSSL_EBADSHAPE;
    if ((gscale || cdot) && group_n <= 0) return GCSSL_EBADSHAPE;
    InBwdParams q{da, ldda, da2, ldda2, da_bcast, z, ldz, mean, rstd, mask, zt, zt_n0, gscale,
                  group_n > 0 ? group_n : N, bias, dzs, lddz, dbias, cdot, HW, C, act};
    dim3 grid(C / CW, N);
    if (dtype == GCSSL_F32) hipLaunchKernelGGL(in_act_bwd_kernel<float>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(in_act_bwd_kernel<bf16_t>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    return gcssl_launch_status();
}

int gcssl_in_dbl_bwd(int dtype, const float* gb_a, int ldgb, const float* qz, int ldq, const void* gb_zs, int ldgz,
                     const float* z, int ldz, const float* mean, const float* rstd, void* gt_a, int ldga,
                     float* zt, float* cdot, int N, int HW, int C, int act, void* stream) {
    if (!gb_a || !qz || !z || !mean || !rstd || !gt_a || !zt) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW < 2 || C <= 0 || C % CW || ldz < C || ldgb < C || ldq < C || ldga < C) return GCSSL_EBADSHAPE;
    InDblParams q{gb_a, ldgb, qz, ldq, gb_zs, ldgz, z, ldz, mean, rstd, gt_a, ldga, zt, cdot, HW, C, act};
    dim3 grid(C / CW, N);
    if (dtype == GCSSL_F32) hipLaunchKernelGGL(in_dbl_bwd_kernel<float>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(in_dbl_bwd_kernel<bf16_t>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    return gcssl_launch_status();
}

int gcssl_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const void* a, int lda,
                  const float* gscale, int group_n, const float* bias, void* dzs, int lddz, float* dbias,
                  float* cdot, int N, int HW, int C, void* stream) {
    if (!da || !a || !dzs) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (N <= 0 || HW <= 0 || C <= 0 || C % CW || lda < C || lddz < C || ldda < C) return GCSSL_EBADSHAPE;
    if ((gscale || cdot) && group_n <= 0) return GCSSL_EBADSHAPE;
    ActBwdParams q{da, ldda, da2, ldda2, a, lda, gscale, group_n > 0 ? group_n : N, bias, dzs, lddz, dbias, cdot, HW, C};
    int zsplit = HW / 64; if (zsplit < 1) zsplit = 1; if (zsplit > 16) zsplit = 16;
    dim3 grid(C / CW, N, zsplit);
    if (dtype == GCSSL_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    else hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid, dim3(CW * RG), 0, (hipStream_t)stream, q);
    return gcssl_launch_status();
}

int gcssl_dot_accum(int dtype, const void* x, int ldx, const float* y, int ldy, long pixels, int C, float* out,
                    void* stream) {
    if (!x || !y || !out) return GCSSL_ENULL;
    if (bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    if (pixels <= 0 || C <= 0 || ldx < C || ldy < C) return GCSSL_EBADSHAPE;
    const size_t total = (size_t)pixels * C;
    int blocks = (int)((total + 255) / 256); if (blocks > 1024) blocks = 1024;
    if (dtype == GCSSL_F32)
        hipLaunchKernelGGL(dot_accum_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx,
                           y, ldy, (size_t)pixels, C, out);
    else
        hipLaunchKernelGGL(dot_accum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           y, ldy, (size_t)pixels, C, out);
    return gcssl_launch_status();
}

}  // extern "C"
