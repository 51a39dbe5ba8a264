// The non-conv pieces of GeneratorSimpleRegressor (cgan/models.py:147-216) for gfx950:
//   nn.MaxPool2d(2, 2) forward / backward                        models.py:169,178,187,196
//   nn.AdaptiveAvgPool2d(1) + Flatten                              models.py:201-202
//   Linear(512,256)+ReLU+Dropout, Linear(256,64)+ReLU+Dropout, Linear(64,4)+Tanh, * delta_scale   models.py:203-216
// The 3x3 convolutions are the Geo<3> instantiations in igemm.hip; InstanceNorm + ReLU are the kernels of norm.hip.
// All of this is HBM/L2-bound byte shuffling or tiny dense algebra (147k weights): no MFMA here.  Activations are NHWC in
// the compute dtype T, gradients that feed a norm backward kernel are fp32 (norm.hip header), the head is fp32 throughout.
#include "common.h"

namespace {

constexpr int D0 = 512, D1 = 256, D2 = 64, D3 = 4;      // regressor widths (models.py:203-210)
constexpr int NS = 4;                                    // samples per workgroup of the head kernels

// ---- MaxPool2d(2,2): one thread = one output pixel x one 16-byte channel vector
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ a, int lda, T* __restrict__ o, int ldo,
                                                           int N, int H, int W, int C) {
    constexpr int KV = Elem<T>::KV;
    const int Ho = H >> 1, Wo = W >> 1, cv = C / KV;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * KV;
        const size_t pix = idx / cv;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        const T* p = a + (((size_t)n * H + 2 * oy) * W + 2 * ox) * lda + c;
        const Vec16<T> v00 = Vec16<T>::load(p), v01 = Vec16<T>::load(p + lda);
        const Vec16<T> v10 = Vec16<T>::load(p + (size_t)W * lda), v11 = Vec16<T>::load(p + (size_t)(W + 1) * lda);
        T* q = o + pix * ldo + c;
#pragma unroll
        for (int i = 0; i < KV; ++i)
            Elem<T>::st(q + i, fmaxf(fmaxf(v00.get(i), v01.get(i)), fmaxf(v10.get(i), v11.get(i))));
    }
}

// backward: the window's gradient goes to its FIRST maximum in row-major order (what max_pool2d's saved indices hold),
// zero elsewhere.  dpool is [N][H/2][W/2][C] fp32, or with bcast a per-sample vector [N][C] scaled by bscale (the
// AdaptiveAvgPool2d(1) backward folded in: every pooled pixel receives dfeat / (Ho*Wo)).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ dpool,
                                                           int ldd, int bcast, float bscale, float* __restrict__ da, int ldda,
                                                           int N, int H, int W, int C) {
    constexpr int KV = Elem<T>::KV;
    const int Ho = H >> 1, Wo = W >> 1, cv = C / KV;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * KV;
        const size_t pix = idx / cv;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        const size_t in0 = (((size_t)n * H + 2 * oy) * W + 2 * ox);
        const T* p = a + in0 * lda + c;
        const Vec16<T> v00 = Vec16<T>::load(p), v01 = Vec16<T>::load(p + lda);
        const Vec16<T> v10 = Vec16<T>::load(p + (size_t)W * lda), v11 = Vec16<T>::load(p + (size_t)(W + 1) * lda);
        const float* g = bcast ? dpool + (size_t)n * ldd + c : dpool + pix * ldd + c;
        float* d = da + in0 * ldda + c;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            const float x0 = v00.get(i), x1 = v01.get(i), x2 = v10.get(i), x3 = v11.get(i);
            int am = 0; float m = x0;
            if (x1 > m) { m = x1; am = 1; }
            if (x2 > m) { m = x2; am = 2; }
            if (x3 > m) { m = x3; am = 3; }
            const float gv = g[i] * bscale;
            d[i] = am == 0 ? gv : 0.f;
            d[ldda + i] = am == 1 ? gv : 0.f;
            d[(size_t)W * ldda + i] = am == 2 ? gv : 0.f;
            d[(size_t)(W + 1) * ldda + i] = am == 3 ? gv : 0.f;
        }
    }
}

// AdaptiveAvgPool2d(1): feat[n][c] = mean over the HW pixels
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, int ldx, float* __restrict__ feat, int N, int HW, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx % C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += Elem<T>::ld(x + ((size_t)n * HW + p) * ldx + c);
    feat[idx] = s / (float)HW;
}

// ---- regressor head, forward.  One workgroup = NS samples.  w1t / w2t are the TRANSPOSED weights [in][out] (re-packed
// once per optimiser step): thread o walks the inputs, a wave reads 256-byte runs of each row and the NS activations come
// from LDS broadcasts -- no cross-lane reduction.  (The first version gave a wave one output at a time and reduced over
// its lanes: 64 dependent global-load round trips per wave, 88 us per launch at B = 256.)
__global__ __launch_bounds__(1024) void mlp_head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w1t,
        const float* __restrict__ b1, const float* __restrict__ w2t, const float* __restrict__ b2, const float* __restrict__ w3,
        const float* __restrict__ b3, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2, float delta_scale,
        float* __restrict__ h1g, float* __restrict__ h2g, float* __restrict__ traw, float* __restrict__ delta, int N) {
    // 1024 threads: every layer splits its input range over 4 (16) thread groups and adds the partial sums through LDS.
    // Activations sit in LDS as [input][sample] so one 16-byte broadcast read feeds the NS = 4 FMAs of a weight: with one
    // 4-byte read per (input, sample) the kernel was bound by LDS instruction issue, not by the weight reads.
    __shared__ float4 f[D0], h1[D1];
    __shared__ float h2[NS][D2], part[16 * NS * D2];
    static_assert(NS == 4 && 4 * NS * D1 <= 16 * NS * D2 && D1 == 256 && D2 == 64 && NS * D1 == 1024, "thread mappings below");
    const int tid = threadIdx.x, n0 = blockIdx.x * NS;
    for (int e = tid; e < NS * D0; e += 1024) {
        const int s = e / D0, i = e % D0;
        reinterpret_cast<float*>(&f[i])[s] = n0 + s < N ? feat[(size_t)(n0 + s) * D0 + i] : 0.f;
    }
    __syncthreads();
    {   // Linear(512,256): 256 outputs x 4 quarters of the inputs
        const int q = tid >> 8, o = tid & 255;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int i = q * (D0 / 4); i < (q + 1) * (D0 / 4); ++i) {
            const float wv = w1t[i * D1 + o];
            const float4 x = f[i];
            acc.x += wv * x.x; acc.y += wv * x.y; acc.z += wv * x.z; acc.w += wv * x.w;
        }
        part[(q * NS + 0) * D1 + o] = acc.x; part[(q * NS + 1) * D1 + o] = acc.y;
        part[(q * NS + 2) * D1 + o] = acc.z; part[(q * NS + 3) * D1 + o] = acc.w;
    }
    __syncthreads();
    {   // + bias, ReLU, Dropout(0.5) (keep mask given; eval: m1 == null): thread = (sample, neuron)
        const int s = tid >> 8, o = tid & 255, n = n0 + s;
        float v = b1[o];
#pragma unroll
        for (int q = 0; q < 4; ++q) v += part[(q * NS + s) * D1 + o];
        v = fmaxf(v, 0.f);
        if (m1 && n < N) v *= m1[(size_t)n * D1 + o] ? 2.f : 0.f;
        reinterpret_cast<float*>(&h1[o])[s] = v;
        if (n < N) h1g[(size_t)n * D1 + o] = v;
    }
    __syncthreads();
    {   // Linear(256,64): 64 outputs x 16 slices of the inputs
        const int q = tid >> 6, o = tid & 63;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = q * (D1 / 16); i < (q + 1) * (D1 / 16); ++i) {
            const float wv = w2t[i * D2 + o];
            const float4 x = h1[i];
            acc.x += wv * x.x; acc.y += wv * x.y; acc.z += wv * x.z; acc.w += wv * x.w;
        }
        part[(q * NS + 0) * D2 + o] = acc.x; part[(q * NS + 1) * D2 + o] = acc.y;
        part[(q * NS + 2) * D2 + o] = acc.z; part[(q * NS + 3) * D2 + o] = acc.w;
    }
    __syncthreads();
    if (tid < NS * D2) {
        const int s = tid / D2, o = tid % D2, n = n0 + s;
        float v = b2[o];
#pragma unroll
        for (int q = 0; q < 16; ++q) v += part[(q * NS + s) * D2 + o];
        v = fmaxf(v, 0.f);
        if (m2 && n < N) v *= m2[(size_t)n * D2 + o] ? 2.f : 0.f;
        h2[s][o] = v;
        if (n < N) h2g[(size_t)n * D2 + o] = v;
    }
    __syncthreads();
    if (tid < NS * D3) {                                             // Linear(64,4) + Tanh, * delta_scale
        const int s = tid / D3, j = tid % D3, n = n0 + s;
        float v = b3[j];
        for (int i = 0; i < D2; ++i) v += w3[j * D2 + i] * h2[s][i];
        if (n < N) {
            const float th = tanhf(v);
            traw[(size_t)n * D3 + j] = th;
            delta[(size_t)n * D3 + j] = th * delta_scale;
        }
    }
}

// ---- backward, data path: per sample the pre-activation gradients dp3 [4], dp2 [64], dp1 [256] and dfeat [512].
// Column access w[o][i] for a fixed i: thread i walks o, so a wave reads 256-byte runs of each row.
__global__ __launch_bounds__(256) void mlp_head_bwd_data_kernel(const float* __restrict__ gdelta, const float* __restrict__ traw,
        const float* __restrict__ h1g, const float* __restrict__ h2g, const float* __restrict__ w1, const float* __restrict__ w2,
        const float* __restrict__ w3, float delta_scale, float drop_scale, float* __restrict__ dp1, float* __restrict__ dp2,
        float* __restrict__ dp3, float* __restrict__ dfeat, int N) {
    __shared__ float g3[NS][D3], g2[NS][D2], g1[NS][D1];
    const int tid = threadIdx.x, n0 = blockIdx.x * NS;
    if (tid < NS * D3) {
        const int s = tid / D3, j = tid % D3, n = n0 + s;
        float v = 0.f;
        if (n < N) {
            const float th = traw[(size_t)n * D3 + j];
            v = gdelta[(size_t)n * D3 + j] * delta_scale * (1.f - th * th);
            dp3[(size_t)n * D3 + j] = v;
        }
        g3[s][j] = v;
    }
    __syncthreads();
    for (int e = tid; e < NS * D2; e += 256) {
        const int s = e / D2, i = e % D2, n = n0 + s;
        float v = 0.f;
        if (n < N) {
#pragma unroll
            for (int o = 0; o < D3; ++o) v += g3[s][o] * w3[o * D2 + i];
            v = h2g[(size_t)n * D2 + i] > 0.f ? v * drop_scale : 0.f;      // Dropout keep * 2 and ReLU' in one test (h = relu*mask*2)
            dp2[(size_t)n * D2 + i] = v;
        }
        g2[s][i] = v;
    }
    __syncthreads();
    {
        const int i = tid;                                           // D1 == 256 threads
        float acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = 0.f;
        for (int o = 0; o < D2; ++o) {
            const float wv = w2[o * D1 + i];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] += g2[s][o] * wv;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = n0 + s;
            float v = 0.f;
            if (n < N) {
                v = h1g[(size_t)n * D1 + i] > 0.f ? acc[s] * drop_scale : 0.f;
                dp1[(size_t)n * D1 + i] = v;
            }
            g1[s][i] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < D0; i += 256) {
        float acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = 0.f;
        for (int o = 0; o < D1; ++o) {
            const float wv = w1[(size_t)o * D0 + i];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] += g1[s][o] * wv;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (n0 + s < N) dfeat[(size_t)(n0 + s) * D0 + i] = acc[s];
    }
}

// ---- backward, weights: dW[o][i] = sum_n dp[n][o] in[n][i], db[o] = sum_n dp[n][o].  One thread = four output rows
// of one input column (the input value is loaded once for four FMAs, the four dp values are one 16-byte broadcast load);
// the batch is walked serially, coalesced along i.
__global__ __launch_bounds__(256) void mlp_head_wgrad_kernel(const float* __restrict__ feat, const float* __restrict__ h1g,
        const float* __restrict__ h2g, const float* __restrict__ dp1, const float* __restrict__ dp2, const float* __restrict__ dp3,
        float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
        float* __restrict__ dw3, float* __restrict__ db3, int N) {
    constexpr int G1 = D1 / 4 * D0, G2 = D2 / 4 * D1, G3 = D3 / 4 * D2, GW = G1 + G2 + G3, EB = D1 + D2 + D3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= GW + EB) return;
    if (idx >= GW) {
        int e = idx - GW;
        const float* dp; float* out; int DO;
        if (e < D1) { dp = dp1; out = db1 + e; DO = D1; }
        else if (e < D1 + D2) { e -= D1; dp = dp2; out = db2 + e; DO = D2; }
        else { e -= D1 + D2; dp = dp3; out = db3 + e; DO = D3; }
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dp[(size_t)n * DO + e];
        *out = s;
        return;
    }
    const float* dp; const float* in; float* out; int og, i, DO, DI;
    if (idx < G1) { og = idx / D0; i = idx % D0; dp = dp1; in = feat; out = dw1; DO = D1; DI = D0; }
    else if (idx < G1 + G2) { const int e = idx - G1; og = e / D1; i = e % D1; dp = dp2; in = h1g; out = dw2; DO = D2; DI = D1; }
    else { const int e = idx - G1 - G2; og = e / D2; i = e % D2; dp = dp3; in = h2g; out = dw3; DO = D3; DI = D2; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
        const float x = in[(size_t)n * DI + i];
        const float4 d = *reinterpret_cast<const float4*>(dp + (size_t)n * DO + 4 * og);
        s0 += d.x * x; s1 += d.y * x; s2 += d.z * x; s3 += d.w * x;
    }
    float* o = out + (size_t)(4 * og) * DI + i;
    o[0] = s0; o[DI] = s1; o[2 * DI] = s2; o[3 * DI] = s3;
}

unsigned grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    return g ? (unsigned)g : 1u;
}

}  // namespace

extern "C" {

int gcssl_maxpool2_fwd(int dtype, const void* a, int lda, void* o, int ldo, int N, int H, int W, int C, void* stream) {
    if (!a || !o) return GCSSL_ENULL;
    if (gcssl_bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    const int kv = dtype == GCSSL_F32 ? 4 : 8;
    if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C % kv || lda < C || ldo < C) return GCSSL_EBADSHAPE;
    if (lda % kv || ldo % kv || !aligned16(a) || !aligned16(o)) return GCSSL_EALIGN;
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / kv);
    hipStream_t st = (hipStream_t)stream;
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(maxpool2_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)a, lda, (T*)o, ldo, N, H, W, C));
    return gcssl_launch_status();
}

int gcssl_maxpool2_bwd(int dtype, const void* a, int lda, const float* dpool, int ldd, int bcast, float bscale, float* da,
                       int ldda, int N, int H, int W, int C, void* stream) {
    if (!a || !dpool || !da) return GCSSL_ENULL;
    if (gcssl_bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    const int kv = dtype == GCSSL_F32 ? 4 : 8;
    if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || C % kv || lda < C || ldd < C || ldda < C) return GCSSL_EBADSHAPE;
    if (lda % kv || !aligned16(a)) return GCSSL_EALIGN;
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / kv);
    hipStream_t st = (hipStream_t)stream;
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(maxpool2_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)a, lda, dpool, ldd, bcast, bscale, da, ldda, N, H, W, C));
    return gcssl_launch_status();
}

int gcssl_avgpool_fwd(int dtype, const void* x, int ldx, float* feat, int N, int HW, int C, void* stream) {
    if (!x || !feat) return GCSSL_ENULL;
    if (N <= 0 || HW <= 0 || C <= 0 || ldx < C) return GCSSL_EBADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(((size_t)N * C + 255) / 256));
    if (gcssl_bad_dtype(dtype)) return GCSSL_EBADDTYPE;
    GCSSL_DISPATCH(dtype, hipLaunchKernelGGL(avgpool_kernel<T>, grid, dim3(256), 0, st, (const T*)x, ldx, feat, N, HW, C));
    return gcssl_launch_status();
}

/* regressor head (models.py:200-216): feat [N][512] -> h1 [N][256], h2 [N][64] (post ReLU+Dropout), traw = tanh [N][4],
 * delta = traw * delta_scale.  w1t [512][256] and w2t [256][64] are the TRANSPOSES of the nn.Linear weights (coalesced
 * reads with one thread per output neuron); w3 [4][64] is plain.  m1 [N][256] / m2 [N][64]: Dropout keep masks (bytes),
 * both NULL in eval mode. */
int gcssl_mlp_head_fwd(const float* feat, const float* w1t, const float* b1, const float* w2t, const float* b2, const float* w3,
                       const float* b3, const uint8_t* m1, const uint8_t* m2, float delta_scale, float* h1, float* h2,
                       float* traw, float* delta, int N, void* stream) {
    if (!feat || !w1t || !b1 || !w2t || !b2 || !w3 || !b3 || !h1 || !h2 || !traw || !delta) return GCSSL_ENULL;
    if (N <= 0 || (!m1) != (!m2)) return GCSSL_EBADSHAPE;
    hipLaunchKernelGGL(mlp_head_fwd_kernel, dim3((N + NS - 1) / NS), dim3(1024), 0, (hipStream_t)stream, feat, w1t, b1, w2t, b2, w3, b3,
                       m1, m2, delta_scale, h1, h2, traw, delta, N);
    return gcssl_launch_status();
}

/* backward of the head for d loss / d delta = gdelta [N][4]: writes the weight and bias gradients (plain stores), the
 * scratch dp1 [N][256], dp2 [N][64], dp3 [N][4] and dfeat [N][512] (gradient wrt the pooled features). */
int gcssl_mlp_head_bwd(const float* gdelta, const float* traw, const float* h1, const float* h2, const float* feat,
                       const float* w1, const float* w2, const float* w3, float delta_scale, int train, float* dp1, float* dp2,
                       float* dp3, float* dfeat, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, int N,
                       void* stream) {
    if (!gdelta || !traw || !h1 || !h2 || !feat || !w1 || !w2 || !w3 || !dp1 || !dp2 || !dp3 || !dfeat || !dw1 || !db1 || !dw2 ||
        !db2 || !dw3 || !db3) return GCSSL_ENULL;
    if (N <= 0) return GCSSL_EBADSHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_head_bwd_data_kernel, dim3((N + NS - 1) / NS), dim3(256), 0, st, gdelta, traw, h1, h2, w1, w2, w3,
                       delta_scale, train ? 2.f : 1.f, dp1, dp2, dp3, dfeat, N);
    constexpr int total = (D1 * D0 + D2 * D1 + D3 * D2) / 4 + D1 + D2 + D3;
    hipLaunchKernelGGL(mlp_head_wgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, st, feat, h1, h2, dp1, dp2, dp3, dw1, db1,
                       dw2, db2, dw3, db3, N);
    return gcssl_launch_status();
}

}  // extern "C"
