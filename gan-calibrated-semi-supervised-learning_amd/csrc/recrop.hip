// Re-crop stage of the training loop on the GPU: get_refined_patch_batch,
// cgan/cgan_train_enhanced.py:37-137, from a device-resident image atlas.  Per sample: clamp the refined box (:82-85),
// pixel rectangle + validity test with fallback to the predicted box (:88-104), integer-truncated crop (:101,104), grey
// padding to a square (:107-112), Pillow's BICUBIC resize to SxS (:115-116) and ToTensor + Normalize(0.5,0.5) (:46-49).
//
// The resize reproduces Pillow 12.2 (src/libImaging/Resample.c) bit for bit: coefficients in IEEE double with the same
// operation order (explicit _rn intrinsics: no FMA contraction), converted to 22-bit fixed point; a horizontal 8-bit pass
// whose result is rounded to uint8, then the vertical pass on those bytes.  Integer accumulation is order-independent,
// so the taps of one output are split over 8 lanes.  HBM-bound in the source crop (each source row is read once).
//
// One workgroup per (sample, chunk of 32 output columns): it walks the source rows of the padded square once; every
// row goes global -> LDS bytes -> horizontal taps -> uint8 -> scattered into the S x 32 vertical accumulators in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

constexpr int PREC = 22;           // PRECISION_BITS of Resample.c (32 - 8 - 2)
constexpr int CCH = 32;            // output columns per workgroup
constexpr int LANES = 8;           // lanes sharing the taps of one output column

__device__ __forceinline__ double bicubic(double x) {        // Resample.c bicubic_filter, a = -0.5
    if (x < 0.0) x = -x;
    if (x < 1.0) return __dadd_rn(__dmul_rn(__dmul_rn(__dsub_rn(__dmul_rn(1.5, x), 2.5), x), x), 1.0);
    if (x < 2.0) return __dmul_rn(__dsub_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dsub_rn(x, 5.0), x), 8.0), x), 4.0), -0.5);
    return 0.0;
}

__device__ __forceinline__ int clip8(int v) { v >>= PREC; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ToTensor (uint8 -> float / 255) then Normalize(0.5, 0.5) in float32, rounded like torch's two ops
__device__ __forceinline__ float norm_px(int v) { return __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), 0.5f), 0.5f); }

struct Rect { double x1, y1, x2, y2; };
__device__ __forceinline__ Rect pixel_rect(const float* box, int W, int H, bool clamp) {
    float cx = box[0], cy = box[1], w = box[2], h = box[3];
    if (clamp) {
        cx = fminf(fmaxf(cx, 0.1f), 0.9f); cy = fminf(fmaxf(cy, 0.1f), 0.9f);
        w = fminf(fmaxf(w, 0.05f), 0.8f);  h = fminf(fmaxf(h, 0.05f), 0.8f);
    }
    const double px = __dmul_rn((double)cx, (double)W), py = __dmul_rn((double)cy, (double)H);
    const double pw = __dmul_rn((double)w, (double)W), ph = __dmul_rn((double)h, (double)H);
    Rect r;
    r.x1 = fmax(0.0, __dsub_rn(px, pw * 0.5)); r.y1 = fmax(0.0, __dsub_rn(py, ph * 0.5));
    r.x2 = fmin((double)W, __dadd_rn(px, pw * 0.5)); r.y2 = fmin((double)H, __dadd_rn(py, ph * 0.5));
    return r;
}

struct RecropParams {
    const uint8_t* atlas; const long* img_off; const int* img_w; const int* img_h; const int* img_idx;
    const float* refined; const float* pred; const float* fallback; float* out; int* status; int* ws;
    int B, S, ksize_max, row_words, max_side, mode;
    unsigned atlas_bytes;
};

// what every kernel of the stage derives from a sample's boxes (cheap scalar math, recomputed instead of stored)
struct Crop { int W, H, l, t, cw, ch, q, pl, pt, status; const uint8_t* img; };
__device__ __forceinline__ Crop make_crop(const RecropParams& p, int n) {
    Crop c;
    const int ii = p.img_idx[n];
    c.W = p.img_w[ii]; c.H = p.img_h[ii]; c.img = p.atlas + p.img_off[ii];
    // mode 0: the training loop's re-crop (clamped box, validity test, fallback to the predicted box);
    // mode 1: the dataset's _letterbox (cgan/dataset.py:104-124): the box as given, no test, no fallback;
    // mode 2: inference.py's crop_patch + letterbox (cgan/inference.py:51-68): as mode 1, but Image.crop gets FLOAT
    //         coordinates there and rounds them to nearest-even instead of truncating
    Rect r = pixel_rect(p.refined + 4 * n, c.W, c.H, p.mode == 0);
    c.status = 0;
    if (p.mode == 0 && (r.x2 <= r.x1 || r.y2 <= r.y1 || (r.x2 - r.x1) < 10.0 || (r.y2 - r.y1) < 10.0)) {       // :95
        r = pixel_rect(p.pred + 4 * n, c.W, c.H, false);
        c.status = 1;
    }
    if (p.mode == 2) { c.l = (int)rint(r.x1); c.t = (int)rint(r.y1); c.cw = (int)rint(r.x2) - c.l; c.ch = (int)rint(r.y2) - c.t; }
    else { c.l = (int)r.x1; c.t = (int)r.y1; c.cw = (int)r.x2 - c.l; c.ch = (int)r.y2 - c.t; }
    if (c.cw < 0 || c.ch < 0) c.status = 2;                 // Image.crop raises -> the reference's except branch
    c.q = max(c.cw, c.ch);
    if (c.q > p.max_side) c.status = 2;                     // larger than the caller's bound: LDS and tables were sized for it
    c.pl = max(c.ch - c.cw, 0) / 2; c.pt = max(c.cw - c.ch, 0) / 2;
    return c;
}

// pixel (x, y) of the padded square: the crop at (pl, pt) of size cw x ch, grey elsewhere
__device__ __forceinline__ int sq_px(const Crop& c, int x, int y, int ch_) {
    const int xx = x - c.pl, yy = y - c.pt;
    if ((unsigned)xx >= (unsigned)c.cw || (unsigned)yy >= (unsigned)c.ch) return 128;
    return c.img[((size_t)(c.t + yy) * c.W + (c.l + xx)) * 3 + ch_];
}

// ---- kernel 1: coefficient tables.  One wave per (sample, output index): precompute_coeffs + normalize_coeffs_8bpc of
// Resample.c for the box (0, q) -> S.  The square crop makes the horizontal and the vertical pass share the table.
// ws[n][xx] = {xmin, xmax, k[0 .. ksize_max)}.  The weights are evaluated in parallel; their sum is taken left to right
// by one lane, as the C loop does (double addition is not associative).
__global__ __launch_bounds__(64) void recrop_coeff_kernel(RecropParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* wbuf = reinterpret_cast<double*>(smem);
    const int xx = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const Crop c = make_crop(p, n);
    int* row = p.ws + ((size_t)n * p.S + xx) * (p.ksize_max + 2);
    if (c.status == 2 || c.q == 0 || c.q == p.S) { if (lane == 0) { row[0] = 0; row[1] = 0; } return; }
    const int q = c.q, S = p.S;
    const double scale = __ddiv_rn((double)q, (double)S);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5); if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5); if (xmax > q) xmax = q;
    xmax -= xmin;
    for (int x = lane; x < xmax; x += 64)
        wbuf[x] = bicubic(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
    __syncthreads();
    double ww = 0.0;
    if (lane == 0) { for (int x = 0; x < xmax; ++x) ww = __dadd_rn(ww, wbuf[x]); wbuf[p.ksize_max] = ww; }
    __syncthreads();
    ww = wbuf[p.ksize_max];
    for (int x = lane; x < p.ksize_max; x += 64) {
        int k = 0;
        if (x < xmax) {
            double w = wbuf[x];
            if (ww != 0.0) w = __ddiv_rn(w, ww);
            k = w < 0.0 ? (int)__dadd_rn(-0.5, __dmul_rn(w, (double)(1 << PREC))) : (int)__dadd_rn(0.5, __dmul_rn(w, (double)(1 << PREC)));
        }
        row[2 + x] = k;
    }
    if (lane == 0) { row[0] = xmin; row[1] = xmax; }
}

// ---- kernel 2: the two 8-bit passes.  Workgroup = (sample, 32 output columns, JCH output rows); it needs the source rows
// of those output rows' windows only.  Its four waves take source rows round-robin and are independent of each other:
// a wave streams its next row into its own LDS buffer with LDS-DMA dword loads (aligned words around the crop's bytes)
// while it works on the current one, takes the horizontal taps (2 lanes per column; the grey padding is a per-column
// constant, not bytes), rounds to uint8 like Pillow's intermediate image, and adds into the shared vertical
// accumulators with LDS integer atomics.
constexpr int JCH = 8;
__global__ __launch_bounds__(256) void recrop_kernel(RecropParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y, i0 = blockIdx.x * CCH, j0 = blockIdx.z * JCH, S = p.S, tid = threadIdx.x;
    const int ncol = min(CCH, S - i0), nrow = min(JCH, S - j0);
    const int KS = p.ksize_max;
    // LDS: kh[CCH][KS] | kv[JCH][KS] | bh[CCH][2] | bv[JCH][2] | gsum[CCH] | acc[JCH][CCH][3] | rowbuf[4 waves][2][row_words]
    int* kh = reinterpret_cast<int*>(smem);
    int* kv = kh + (size_t)CCH * KS;
    int* bh = kv + (size_t)JCH * KS;
    int* bv = bh + 2 * CCH;
    int* gsum = bv + 2 * JCH;
    int* acc = gsum + CCH;
    unsigned* rowbuf = reinterpret_cast<unsigned*>(acc + JCH * CCH * 3);
    const Crop c = make_crop(p, n);
    float* out = p.out + (size_t)n * 3 * S * S;
    if (tid == 0 && blockIdx.x == 0 && blockIdx.z == 0 && p.status) p.status[n] = c.status;
    if (c.status == 2 || c.q == 0 || c.q == S) {            // no resize: failed / 0x0 crop (Pillow: zeros) / exact size
        for (int e = tid; e < 3 * nrow * ncol; e += 256) {
            const int ch_ = e / (nrow * ncol), j = j0 + (e / ncol) % nrow, i = i0 + e % ncol;
            float v;
            if (c.status == 2) v = p.fallback ? p.fallback[((size_t)n * 3 + ch_) * S * S + (size_t)j * S + i] : 0.f;
            else if (c.q == 0) v = norm_px(0);
            else v = norm_px(sq_px(c, i, j, ch_));
            out[(size_t)ch_ * S * S + (size_t)j * S + i] = v;
        }
        return;
    }
    // ---- this workgroup's slices of the coefficient table
    const int* wsn = p.ws + (size_t)n * S * (KS + 2);
    for (int e = tid; e < ncol * KS; e += 256) kh[e] = wsn[(size_t)(i0 + e / KS) * (KS + 2) + 2 + e % KS];
    for (int e = tid; e < nrow * KS; e += 256) kv[e] = wsn[(size_t)(j0 + e / KS) * (KS + 2) + 2 + e % KS];
    if (tid < 2 * ncol) bh[tid] = wsn[(size_t)(i0 + tid / 2) * (KS + 2) + (tid & 1)];
    if (tid >= 64 && tid < 64 + 2 * nrow) bv[tid - 64] = wsn[(size_t)(j0 + (tid - 64) / 2) * (KS + 2) + (tid & 1)];
    for (int e = tid; e < JCH * CCH * 3; e += 256) acc[e] = 1 << (PREC - 1);
    __syncthreads();
    // per-column sum of the taps that fall on the grey padding (same for every source row inside the crop) and of all taps
    if (tid < ncol) {
        const int xmin = bh[2 * tid], xmax = bh[2 * tid + 1];
        int g = 0;
        for (int x = 0; x < xmax; ++x) { const int xc = xmin + x - c.pl; if ((unsigned)xc >= (unsigned)c.cw) g += kh[tid * KS + x]; }
        gsum[tid] = g;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int col = lane >> 1, part = lane & 1;
    const bool live = col < ncol;
    const int cxmin = live ? bh[2 * col] : 0, cxmax = live ? bh[2 * col + 1] : 0;
    // taps of this column that read crop pixels: x in [ta, tb) (square coordinates xmin + x), crop x = xmin + x - pl
    const int ta = max(0, c.pl - cxmin), tb = min(cxmax, c.pl + c.cw - cxmin);
    int ksum = 0;
    if (live) for (int x = 0; x < cxmax; ++x) ksum += kh[col * KS + x];
    const int gs = live ? gsum[col] : 0;
    const int* ck = kh + col * KS;
    const int y_beg = bv[0], y_end = bv[2 * (nrow - 1)] + bv[2 * (nrow - 1) + 1];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.atlas), 0, p.atlas_bytes, 0x00020000);
    unsigned* mybuf = rowbuf + (size_t)wave * 2 * p.row_words;
    // source bytes of crop row yy: [rowaddr, rowaddr + cw*3); the aligned words around them go to LDS, phase = rowaddr & 3
    auto row_addr = [&](int y) -> long { return (long)(c.img - p.atlas) + ((long)(c.t + y - c.pt) * c.W + c.l) * 3; };
    auto issue = [&](int y, int slot) {
        if ((unsigned)(y - c.pt) >= (unsigned)c.ch) return;                       // padding row: nothing to load
        const long a = row_addr(y);
        const unsigned base = (unsigned)(a & ~3L);
        const int nwords = (int)(((a + (long)c.cw * 3 + 3) & ~3L) - (long)base) >> 2;
        unsigned* dst = mybuf + (size_t)slot * p.row_words;
        for (int w0 = 0; w0 < nwords; w0 += 64)                                   // 256 B per wave-instruction
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + w0), 4,
                                                     (w0 + lane < nwords) ? base + 4u * (unsigned)(w0 + lane) : 0x80000000u, 0, 0, 0);
    };
    int slot = 0;
    int y = y_beg + wave;
    if (y < y_end) issue(y, 0);
    for (; y < y_end; y += 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's row y has landed
        __builtin_amdgcn_wave_barrier();
        if (y + 4 < y_end) issue(y + 4, slot ^ 1);                                // next row streams in during the taps
        const bool row_in = (unsigned)(y - c.pt) < (unsigned)c.ch;
        int s0, s1, s2;
        if (!row_in) { s0 = s1 = s2 = part == 0 ? 128 * ksum : 0; }               // grey row
        else {
            s0 = s1 = s2 = part == 0 ? 128 * gs : 0;
            const uint8_t* rp = reinterpret_cast<const uint8_t*>(mybuf + (size_t)slot * p.row_words) + (row_addr(y) & 3)
                                + (cxmin - c.pl) * 3;
            for (int x = ta + part; x < tb; x += 2) {
                const int k = ck[x];
                s0 += rp[3 * x] * k; s1 += rp[3 * x + 1] * k; s2 += rp[3 * x + 2] * k;
            }
        }
        s0 += __shfl_xor(s0, 1, 64); s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
        if (live) {
            const int h0 = clip8(s0 + (1 << (PREC - 1))), h1 = clip8(s1 + (1 << (PREC - 1))), h2 = clip8(s2 + (1 << (PREC - 1)));
            for (int j = part; j < nrow; j += 2) {
                const int d = y - bv[2 * j];
                if ((unsigned)d >= (unsigned)bv[2 * j + 1]) continue;             // row y outside output row j's window
                const int kvv = kv[j * KS + d];
                int* a = acc + (j * CCH + col) * 3;
                atomicAdd(a, h0 * kvv); atomicAdd(a + 1, h1 * kvv); atomicAdd(a + 2, h2 * kvv);
            }
        }
        slot ^= 1;
    }
    __syncthreads();
    for (int e = tid; e < 3 * nrow * ncol; e += 256) {
        const int ch_ = e / (nrow * ncol), j = (e / ncol) % nrow, i = e % ncol;
        out[(size_t)ch_ * S * S + (size_t)(j0 + j) * S + i0 + i] = norm_px(clip8(acc[(j * CCH + i) * 3 + ch_]));
    }
}

// eval-mode box transform apply_delta_to_bbox(training=False), cgan/losses.py:108-150 (fp32 like the reference)
__global__ void apply_delta_eval_kernel(const float* __restrict__ box, const float* __restrict__ delta, float* __restrict__ out, int B) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    const float* b = box + 4 * n; const float* d = delta + 4 * n;
    float dc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dc[i] = fminf(fmaxf(d[i], -1.5f), 1.5f);
    const float cx = __fadd_rn(b[0], __fmul_rn(dc[0], b[2])), cy = __fadd_rn(b[1], __fmul_rn(dc[1], b[3]));
    const float w = __fmul_rn(b[2], expf(fminf(fmaxf(dc[2], -1.f), 1.f))), h = __fmul_rn(b[3], expf(fminf(fmaxf(dc[3], -1.f), 1.f)));
    out[4 * n + 0] = fminf(fmaxf(cx, 0.05f), 0.95f); out[4 * n + 1] = fminf(fmaxf(cy, 0.05f), 0.95f);
    out[4 * n + 2] = fminf(fmaxf(w, 0.02f), 0.8f);   out[4 * n + 3] = fminf(fmaxf(h, 0.02f), 0.8f);
}

}  // namespace

extern "C" {

int gcssl_init_recrop() {         // dynamic-LDS opt-in up to the 160 KB of a CU, once, outside any stream capture
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(recrop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e == hipSuccess ? GCSSL_OK : (int)e;
}

int gcssl_recrop_ws_ints(int B, int S, int max_side) {
    if (B <= 0 || S < 2 || S > 256 || max_side < 1) return GCSSL_EBADSHAPE;
    const double fs = (double)max_side / S > 1.0 ? (double)max_side / S : 1.0;
    const long n = (long)B * S * ((long)ceil(2.0 * fs) * 2 + 1 + 2);
    return n > 0x7FFFFFFF ? GCSSL_EBADSHAPE : (int)n;
}

int gcssl_recrop_patches(const uint8_t* atlas, long atlas_bytes, const long* img_off, const int* img_w, const int* img_h,
                         const int* img_idx, const float* refined_box, const float* pred_box, const float* fallback,
                         float* out, int* status, int* ws, int B, int S, int max_side, int mode, void* stream) {
    if (!atlas || !img_off || !img_w || !img_h || !img_idx || !refined_box || (mode == 0 && !pred_box) || !out || !ws) return GCSSL_ENULL;
    if (B <= 0 || S < 2 || S > 256 || max_side < 1 || atlas_bytes <= 0 || atlas_bytes >= 0x7FFFFFFFL || mode < 0 || mode > 2)
        return GCSSL_EBADSHAPE;
    // worst case over the launch: a crop side of max_side pixels
    const double fs = (double)max_side / S > 1.0 ? (double)max_side / S : 1.0;
    const int ksize_max = (int)ceil(2.0 * fs) * 2 + 1;
    const int row_words = ((max_side + 8) * 3 + 3) / 4 + 2;
    const size_t lds = ((size_t)(CCH + JCH) * ksize_max + 2 * CCH + 2 * JCH + CCH + (size_t)JCH * CCH * 3) * 4 + (size_t)4 * 2 * row_words * 4;
    if (lds > 160 * 1024) return GCSSL_EBADSHAPE;            // crop side too large for one workgroup's LDS
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(recrop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    RecropParams p{atlas, img_off, img_w, img_h, img_idx, refined_box, pred_box, fallback, out, status, ws, B, S, ksize_max, row_words,
                   max_side, mode, (unsigned)atlas_bytes};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(recrop_coeff_kernel, dim3(S, B), dim3(64), (size_t)(ksize_max + 2) * 8, st, p);
    hipLaunchKernelGGL(recrop_kernel, dim3((S + CCH - 1) / CCH, B, (S + JCH - 1) / JCH), dim3(256), lds, st, p);
    return gcssl_launch_status();
}

int gcssl_apply_delta_eval(const float* box, const float* delta, float* out, int B, void* stream) {
    if (!box || !delta || !out) return GCSSL_ENULL;
    if (B <= 0) return GCSSL_EBADSHAPE;
    hipLaunchKernelGGL(apply_delta_eval_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, box, delta, out, B);
    return gcssl_launch_status();
}

}  // extern "C"
