// Pixel-stationary transposed convolution (k4 s2 p1) with the InstanceNorm + ReLU (+ global-average-pool) epilogue fused in:
// the generator's up path  ConvTranspose2d -> InstanceNorm2d -> ReLU  (cgan/models.py:72-74,112-115) as ONE launch per layer.
//
// Why another conv kernel.  The implicit-GEMM kernels of igemm.hip re-gather every A row of every K step from L2: a
// 128x64 tile moves 24 KB into LDS per 1.05 MFLOP (44 FLOP/B), and LDS-DMA throughput per CU is bytes-in-flight / latency
// (~96 KB / 1.1 us = 82 GB/s measured, MI355X_MICROARCH.md "ldsdma-fill"), so their K loop tops out near 36 % of the MFMA
// rate whatever the tile.  A transposed conv reads each INPUT pixel 16 times (4 taps x 4 output-parity
// classes): here a workgroup keeps 256 input pixels in LDS ONCE per 32-channel chunk and computes all four parity classes
// of their 1024 output pixels from it; only the weight tiles stream (16 KB per K step for 4.2 MFLOP: 210 FLOP/B with the
// input).  The loop is MFMA-bound: per wave and K step 16 MFMAs (128 x 64 wave tile) against 12 ds_read_b128 and ~3 DMA
// issues, one barrier.
//
//   workgroup   = 256 input pixels = 1 sample of 16x16 (G.up4 at 32x32 images) or 4 samples of 8x8 (G.up3), all 64 output
//                 channels; persistent over pixel blocks (grid = #CUs).
//   wave (8)    = output-parity class (py,px) = wave>>1, pixel half = wave&1: 128 rows x 64 columns, acc 4x2 tiles of 32x32.
//   K loop      = for chunk q of 32 input channels: for tap t=(ty,tx) of the class's 4 taps: one step (BK = 32).
//   LDS         = A: 3 chunk buffers [256 px][64 B], B: 4-slot ring of [4 classes][64 co][64 B]; 16-byte chunks XOR-swizzled
//                 by ((row>>2)&3) on the DMA SOURCE side so that ds_read_b128 fragments are conflict-free; one zero row
//                 stands in for out-of-image taps.
//   epilogue    = whole samples live in the workgroup's accumulators: exact two-pass mean / variance per (sample, channel)
//                 across waves through LDS, then x^ = (z-mean)*rstd, ReLU, optional outputs: the activation (16-bit), the
//                 pre-norm z in fp32 for the samples whose backward needs it, the pooled sums for the head.
//
// Reference ops replaced: nn.ConvTranspose2d(k4,s2,p1,bias=False) + nn.InstanceNorm2d + nn.ReLU (+ AdaptiveAvgPool2d(1)
// for up4): cgan/models.py:72-74,112-118.
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lds_void_p;
constexpr float IN_EPS = 1e-5f;

template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { typedef bf16x8 type; };
template <> struct FragOf<f16_t> { typedef f16x8 type; };
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

struct CtParams {
    const void* x;      // input activations [N][H][H][ldx >= K] (16-bit)
    const void* wt;     // Wt[64 co][16 taps][K] (the dgrad pack of igemm.hip: gcssl_prep_conv_weight's `wt`)
    float* z32;         // nullable: pre-norm output [N][2H][2H][ldz] fp32, written for samples >= z_n0 only
    void* a;            // nullable: activation output [N][2H][2H][lda] (16-bit)
    float* mean; float* rstd;   // [N][64]
    float* pool;        // nullable: [N][64] sum over the output pixels of the activation (written, not accumulated)
    float* cnt;         // nullable (needs pool): [N][64] number of output pixels with a positive normalised value (written)
    int ldx, ldz, lda, z_n0;
    int N, lgH, K;      // H = 1 << lgH in {8, 16}; K input channels (multiple of 32)
    int nblocks;        // pixel blocks of 256 input pixels = N * H * H / 256
    int debug;          // timing experiments only (GCSSL_CT_DEBUG; results are wrong): 1 no DMA in the loop, 2 no barriers, 4 no LDS reads after the first double step
    int rotate;         // 1: workgroup b starts its K loop at chunk b % nq (all workgroups stream the SAME weights: without
                        // the rotation they all ask L2 for the same 16 KB tile at the same moment)
    unsigned x_bytes, w_bytes;
};

constexpr int NTH = 512, BSLOT = 16384, ABUF = 16384 + 64;      // an A buffer = 256 pixel rows + one zero row (out-of-image taps)
constexpr int NB = 6;                                            // weight ring slots: tiles are issued NB-1 K steps ahead
constexpr int LDS_B = 0, LDS_A = NB * BSLOT, LDS_RED = LDS_A + 3 * ABUF + 64;
constexpr int LDS_TOTAL = LDS_RED + 2 * 8 * 2 * 64 * 4;          // two [wave][slot][64] float arrays

// a row of 64 bytes (32 channels) holds 4 chunks of 16 B; 4 rows share a 256-B bank row.  Physical chunk = c ^ ((row>>2)&3):
// the 16 lanes of a ds_read_b128 group read 16 rows with 4 values of row&3 and 4 of (row>>2)&3 -> 16 distinct 16-B slots.
// The chunk of MFMA k-step kk is c = 2*kk + (lane>>5), so the kk = 1 address is the kk = 0 address ^ 32.
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// LDS writes -> visible to the other waves, WITHOUT draining the LDS-DMA prefetches in flight (a __syncthreads() here would
// emit s_waitcnt vmcnt(0): cdna_hip_programming.md "Pipelining across barriers")
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int LGH>
__global__ __launch_bounds__(NTH) void convt_in_relu_kernel(CtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename FragOf<T>::type FragT;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cls = wave >> 1, half = wave & 1, py = cls >> 1, px = cls & 1;
    constexpr int H = 1 << LGH, HW = H * H, lgHW = 2 * LGH;
    const int nq = p.K >> 5, nsteps = nq * 4;                          // 32-channel chunks; K steps per pixel block
    const int rq = p.rotate ? (int)(blockIdx.x % (unsigned)nq) : 0;     // chunk rotation of this workgroup's K order
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, p.w_bytes, 0x00020000);

    // the zero row behind each input buffer (no DMA ever writes it)
    if (tid < 48) reinterpret_cast<unsigned*>(lds + LDS_A + (tid >> 4) * ABUF + 16384)[tid & 15] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (ordered before the first reads by the K loop's first barrier)

    // ---- DMA issue: waves 4..7 are the producers (each shares a SIMD with one of waves 0..3, which start their MFMAs while
    // the producer is still issuing: the two waves of a SIMD leave the barrier together, and an in-order wave cannot issue
    // MFMAs while it issues DMA).  Per K step producer pw = wave-4 issues the 4 weight pieces of parity class pw (rows
    // 16*part + (lane>>2), physical chunk lane&3) and ONE input piece: fixed counts for the vmcnt bookkeeping.
    const bool prod = wave >= 4;
    const int pw = wave & 3;
    int boff[4];                                                        // byte offset of the lane's 16 bytes in Wt, without q and tap
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        const int row = 16 * part + (lane >> 2);                        // output channel
        const int lc = (lane & 3) ^ ((row >> 2) & 3);                   // logical chunk this lane fetches
        boff[part] = (row * 16 * p.K + lc * 8) * 2;
    }
    auto issue_b = [&](int s, int slot) {                               // s = K step within a block (tap t = s&3, chunk q = s>>2)
        int q = (s >> 2) + rq; if (q >= nq) q -= nq;
        const int t = s & 3, ty = t >> 1, tx = t & 1;
        const int tap = (1 - (pw >> 1) + 2 * ty) * 4 + (1 - (pw & 1) + 2 * tx);
#pragma unroll
        for (int part = 0; part < 4; ++part)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_p)(lds + LDS_B + slot * BSLOT + (4 * pw + part) * 1024), 16,
                                                     (unsigned)(boff[part] + (tap * p.K + q * 32) * 2), 0, 0, 0);
    };
    // input piece id (0..15) of chunk position qi of pixel block blk: pixels 16*id + (lane>>2), physical chunk lane&3.
    auto issue_a = [&](int blk, int qi, int id, int buf) {            // qi: position in this workgroup's chunk order
        int q = qi + rq; if (q >= nq) q -= nq;
        const int pix = 16 * id + (lane >> 2);
        const int lc = (lane & 3) ^ ((pix >> 2) & 3);
        const unsigned off = (unsigned)((((size_t)blk * 256 + pix) * p.ldx + q * 32 + lc * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(lds + LDS_A + buf * ABUF + id * 1024), 16,
                                                 blk < p.nblocks ? off : OOB, 0, 0, 0);
    };

    // The K loop advances in DOUBLE steps (two taps = 32 MFMAs per wave between barriers: a barrier + first-fragment latency
    // costs ~450 cycles, as much as the 16 MFMAs of a single tap).  Global double-step counter g over (block, double step):
    // its two weight tiles live in ring slots 2 (g % 3), +1 and are issued TWO double steps ahead; input chunk q of a block
    // lives in A buffer (chunk counter) % 3 and its 16 pieces are issued during the two double steps of the chunk two before it.
    // A producer wave issues 8 + 2 DMA instructions per double step; at the top of double step g everything it issued up to
    // g-2 must have landed, i.e. at most the 10 instructions of g-1 may be outstanding (also right after the prologue,
    // which issues the tiles of double steps 0 and 1: 8 younger instructions at g = 0).
    static_assert(NB == 6, "three double slots");
    int blk = blockIdx.x;
    if (blk >= p.nblocks) return;
    if (prod) {   // prologue: chunks 0 and 1 of the first block (8 pieces per producer), weight tiles of double steps 0 and 1
#pragma unroll
        for (int k = 0; k < 4; ++k) { issue_a(blk, 0, 4 * k + pw, 0); issue_a(blk, 1, 4 * k + pw, 1); }
#pragma unroll
        for (int k = 0; k < 4; ++k) issue_b(k, k);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // chunks 0, 1 and tiles 0, 1 landed; tiles 2, 3 may fly
    }
    int g = 0;                                                          // global double step
    int slot = 0;                                                       // ring slot of the double step's first tile: 2 (g % 3)
    int abuf = 0;                                                       // A buffer of the current chunk (cycles 0, 1, 2)
    while (true) {
        const int nblk = blk + (int)gridDim.x;
        // ---- per-lane LDS offsets: the wave's 4 row blocks are input positions pos = 128*half + 32*i + (lane&31) of the pixel
        // block; tap t = (ty,tx) reads the pixel shifted by (py-ty, px-tx), or the zero row (index 256) when that falls outside
        // the sample's map.  abase[t][i] / bbase[j]: kk = 0 addresses inside an input buffer / a weight slot (kk = 1: ^ 32).
        // Recomputed per block from an opaque copy of the lane id: 18 registers that need not live across the epilogue.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int abase[4][4], bbase[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = 128 * half + 32 * i + (ln & 31);
            const int rem = pos & (HW - 1), iy = rem >> LGH, ix = rem & (H - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
                const int yy = iy + py - ty, xx = ix + px - tx;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)H;
                abase[t][i] = swz(ok ? pos + (py - ty) * H + (px - tx) : 256, ln >> 5);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) bbase[j] = cls * 4096 + swz(32 * j + (ln & 31), ln >> 5);
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int qi = 0; qi < ((p.debug & 8) ? 0 : nq); ++qi) {
            const int abuf2 = abuf >= 1 ? abuf - 1 : 2;                 // (abuf + 2) % 3: where the chunk two ahead goes
#pragma unroll
            for (int dd = 0; dd < 2; ++dd, ++g) {
                const int t0 = 2 * dd, s0 = qi * 4 + t0;               // the double step's taps t0, t0+1 = K steps s0, s0+1
                if (prod && g > 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                if (!(p.debug & 2)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (prod && !(p.debug & 1)) {   // tiles of double step g+2 (their slots were read at g-1) and two pieces of the chunk two ahead
                    const int sn = s0 + 4, sl2 = slot >= 2 ? slot - 2 : 4;     // (slot + 4) % 6
                    issue_b(sn < nsteps ? sn : sn - nsteps, sl2);       // (same weights for every block: re-read from L2)
                    issue_b(sn + 1 < nsteps ? sn + 1 : sn + 1 - nsteps, sl2 + 1);
                    const int q2 = qi + 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int id = 8 * dd + 4 * k + pw;
                        if (q2 < nq) issue_a(blk, q2, id, abuf2);
                        else issue_a(nblk, q2 - nq, id, abuf2);         // (no next block: OOB offset, zeros, still counted)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // The instruction stream is pinned group by group (sched_barrier): only the first six fragment reads are exposed,
                // every other read rides between the MFMAs of the previous group.
                const unsigned char* Ab = lds + LDS_A + abuf * ABUF;
                const unsigned char* Bb = lds + LDS_B + slot * BSLOT;
                auto ldA = [&](int t, int kk, int i) { return __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(Ab + (abase[t][i] ^ (kk << 5)))); };
                auto ldB = [&](int u, int kk, int j) { return __builtin_bit_cast(FragT, *reinterpret_cast<const uint4*>(Bb + u * BSLOT + (bbase[j] ^ (kk << 5)))); };
                FragT ax[4], bx[2], ay[4], by[2];
                bx[0] = ldB(0, 0, 0); bx[1] = ldB(0, 0, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) ax[i] = ldA(t0, 0, i);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                // group = 8 MFMAs of fragment set (a, b) with the 6 reads of the next set (na, nb) issued between them
#define CT_GROUP(a, b, na, nb, T, U, KK)                                                                   \
                acc[0][0] = mma(a[0], b[0], acc[0][0]); acc[0][1] = mma(a[0], b[1], acc[0][1]);             \
                nb[0] = ldB(U, KK, 0); nb[1] = ldB(U, KK, 1);                                               \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                acc[1][0] = mma(a[1], b[0], acc[1][0]); acc[1][1] = mma(a[1], b[1], acc[1][1]);             \
                na[0] = ldA(T, KK, 0); na[1] = ldA(T, KK, 1);                                               \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                acc[2][0] = mma(a[2], b[0], acc[2][0]); acc[2][1] = mma(a[2], b[1], acc[2][1]);             \
                na[2] = ldA(T, KK, 2); na[3] = ldA(T, KK, 3);                                               \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                acc[3][0] = mma(a[3], b[0], acc[3][0]); acc[3][1] = mma(a[3], b[1], acc[3][1]);             \
                __builtin_amdgcn_sched_barrier(0);
                CT_GROUP(ax, bx, ay, by, t0, 0, 1)                      // tap t0 kk0   | reads tap t0   kk1
                CT_GROUP(ay, by, ax, bx, t0 + 1, 1, 0)                  // tap t0 kk1   | reads tap t0+1 kk0
                CT_GROUP(ax, bx, ay, by, t0 + 1, 1, 1)                  // tap t0+1 kk0 | reads tap t0+1 kk1
#undef CT_GROUP
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][0] = mma(ay[i], by[0], acc[i][0]); acc[i][1] = mma(ay[i], by[1], acc[i][1]); }
                __builtin_amdgcn_s_setprio(0);
                slot = slot == 4 ? 0 : slot + 2;
            }
            abuf = abuf == 2 ? 0 : abuf + 1;
        }
        // ---- epilogue: InstanceNorm statistics per (sample, channel) over the sample's 4*HW output pixels.
        // Sample slots of this wave: HW = 256 -> one (all four row blocks); HW = 64 -> two (row blocks 0-1 and 2-3).
        // red[wave][slot][64]; a sample's rows live in the waves of every class (HW = 256: of both halves as well).
        if (p.debug & 16) { if (nblk >= p.nblocks) break; blk = nblk; continue; }
        float* red = reinterpret_cast<float*>(lds + LDS_RED);
        constexpr int nslot = HW >= 128 ? 1 : 2, bps = 4 / nslot;        // row blocks per slot
        const float inv_cnt = 1.0f / (float)(4 * HW);
        float mu[nslot][2], rs[nslot][2];                                // [slot][col j]
        {   // Each wave reduces its own rows exactly (two passes over registers: mean_w, then M2_w = sum (z - mean_w)^2), the
            // waves' results are combined with Chan's formula: one LDS exchange instead of one per pass.
            float* red2 = red + 8 * 2 * 64;
            constexpr float cnt_w = (float)(bps * 32);                   // values per (wave, slot, column)
#pragma unroll
            for (int sl = 0; sl < nslot; ++sl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = sl * bps; i < (sl + 1) * bps; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v += acc[i][j][r];
                    v += __shfl_xor(v, 32, 64);
                    const float mw = v * (1.0f / cnt_w);
                    float m2 = 0.f;
#pragma unroll
                    for (int i = sl * bps; i < (sl + 1) * bps; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - mw; m2 += d * d; }
                    m2 += __shfl_xor(m2, 32, 64);
                    if (lane < 32) { red[(wave * 2 + sl) * 64 + 32 * j + lane] = mw; red2[(wave * 2 + sl) * 64 + 32 * j + lane] = m2; }
                }
            lds_barrier();
#pragma unroll
            for (int sl = 0; sl < nslot; ++sl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // waves that hold rows of this sample: HW=256: all 8; HW=64: the 4 with the same pixel half
                    float ms = 0.f, m2s = 0.f, mw[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        mw[w] = 0.f;
                        if (HW < 128 && (w & 1) != half) continue;
                        mw[w] = red[(w * 2 + sl) * 64 + 32 * j + (lane & 31)];
                        ms += mw[w]; m2s += red2[(w * 2 + sl) * 64 + 32 * j + (lane & 31)];
                    }
                    constexpr float nw = HW >= 128 ? 8.f : 4.f;
                    const float m = ms * (1.0f / nw);
                    float between = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        if (HW < 128 && (w & 1) != half) continue;
                        between += (mw[w] - m) * (mw[w] - m);
                    }
                    mu[sl][j] = m;
                    rs[sl][j] = 1.0f / sqrtf((m2s + cnt_w * between) * inv_cnt + IN_EPS);
                }
            lds_barrier();                                               // red[] is rewritten by the pool reduction
        }
        // outputs.  Row r of C fragment block i is input position 128*half + 32*i + crow -> output pixel (2iy+py, 2ix+px).
        constexpr int spw = 256 >> lgHW;                                 // samples per pixel block (1 or 4)
        auto out_pixel = [&](int i, int r, int& n) -> size_t {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int ps = 128 * half + 32 * i + row, rem = ps & (HW - 1);
            n = blk * spw + (ps >> lgHW);
            const int oy = 2 * (rem >> LGH) + py, ox = 2 * (rem & (H - 1)) + px;
            return ((size_t)n * 2 * H + oy) * 2 * H + ox;
        };
        if (p.z32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n0 = blk * spw + ((128 * half + 32 * i) >> lgHW);      // the block's sample (wave-uniform)
                if (n0 < p.z_n0) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int n; const size_t opix = out_pixel(i, r, n);
#pragma unroll
                    for (int j = 0; j < 2; ++j) p.z32[opix * p.ldz + 32 * j + (lane & 31)] = acc[i][j][r];
                }
            }
        }
        if (p.a) {
            T* ap = static_cast<T*>(p.a);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int n; const size_t opix = out_pixel(i, r, n);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        Elem<T>::st(ap + opix * p.lda + 32 * j + (lane & 31), fmaxf((acc[i][j][r] - mu[i / bps][j]) * rs[i / bps][j], 0.f));
                }
        }
        // mean / rstd (one writer per (sample, channel): class 0, and for HW=256 half 0)
        const bool writer = cls == 0 && (HW < 128 || half == 0) && lane < 32;
        if (writer) {
#pragma unroll
            for (int sl = 0; sl < nslot; ++sl) {
                const int n = blk * spw + (HW >= 128 ? 0 : 2 * half + sl);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    p.mean[(size_t)n * 64 + 32 * j + lane] = mu[sl][j];
                    p.rstd[(size_t)n * 64 + 32 * j + lane] = rs[sl][j];
                }
            }
        }
        if (p.pool) {                                                    // sum over the output pixels of the activation
#pragma unroll
            for (int sl = 0; sl < nslot; ++sl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = 0.f, c = 0.f;
#pragma unroll
                    for (int i = sl * bps; i < (sl + 1) * bps; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float xh = (acc[i][j][r] - mu[sl][j]) * rs[sl][j];
                            v += fmaxf(xh, 0.f); c += xh > 0.f ? 1.f : 0.f;
                        }
                    v += __shfl_xor(v, 32, 64); c += __shfl_xor(c, 32, 64);
                    if (lane < 32) { red[(wave * 2 + sl) * 64 + 32 * j + lane] = v; red[1024 + (wave * 2 + sl) * 64 + 32 * j + lane] = c; }
                }
            lds_barrier();
            if (writer) {
#pragma unroll
                for (int sl = 0; sl < nslot; ++sl) {
                    const int n = blk * spw + (HW >= 128 ? 0 : 2 * half + sl);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float tsum = 0.f, csum = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            if (HW < 128 && (w & 1) != half) continue;
                            tsum += red[(w * 2 + sl) * 64 + 32 * j + lane];
                            csum += red[1024 + (w * 2 + sl) * 64 + 32 * j + lane];
                        }
                        p.pool[(size_t)n * 64 + 32 * j + lane] = tsum;
                        if (p.cnt) p.cnt[(size_t)n * 64 + 32 * j + lane] = csum;
                    }
                }
            }
            lds_barrier();
        }
        if (nblk >= p.nblocks) break;
        blk = nblk;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the ring's last (discarded) prefetches
#endif
}

}  // namespace

extern "C" {

/* ConvTranspose2d(K -> 64, k4 s2 p1, no bias) + InstanceNorm2d + ReLU (+ the sums of AdaptiveAvgPool2d(1)) in one launch
 * (cgan/models.py:72-74,112-118), for inputs of 8x8 or 16x16 pixels.  wt: the dgrad pack Wt[64][16][K] of
 * gcssl_prep_conv_weight.  Outputs (each nullable except mean/rstd): a [N][2H][2H][lda] in `dtype`; z32 fp32 pre-norm values
 * for samples >= z_n0 (what gcssl_in_act_bwd reads); pool[N][64] = sum over pixels of the activation (written); cnt[N][64]
 * (needs pool) = number of output pixels whose normalised value is positive, i.e. sum of ReLU' (written): with pool it lets
 * gcssl_in_act_bwd skip its statistics pass when the incoming gradient is a per-(n, c) constant (presum_cnt / presum_pos). */
int gcssl_convT4x4s2_in_relu_fwd(int dtype, const void* x, int ldx, const void* wt, float* z32, int ldz, int z_n0, void* a,
                                 int lda, float* mean, float* rstd, float* pool, float* cnt, int N, int H, int K, int Cout,
                                 void* stream) {
    if (!x || !wt || !mean || !rstd || (cnt && !pool)) return GCSSL_ENULL;
    if (dtype != GCSSL_BF16 && dtype != GCSSL_F16) return GCSSL_EBADDTYPE;
    if (N <= 0 || (H != 8 && H != 16) || K < 64 || K % 32 || Cout != 64 || ldx < K || ldx % 8) return GCSSL_EBADSHAPE;   // (K >= 64: 8 K steps > ring depth)
    if ((N * H * H) % 256) return GCSSL_EBADSHAPE;                      // whole pixel blocks (H = 8: N a multiple of 4)
    if ((z32 && ldz < 64) || (a && (lda < 64))) return GCSSL_EBADSHAPE;
    if (!aligned16(x) || !aligned16(wt)) return GCSSL_EALIGN;
    const size_t xb = (size_t)N * H * H * ldx * 2, wb = (size_t)64 * 16 * K * 2;
    if (xb >= 0x7FFFFFFFull) return GCSSL_EBADSHAPE;
    CtParams p{};
    p.x = x; p.wt = wt; p.z32 = z32; p.a = a; p.mean = mean; p.rstd = rstd; p.pool = pool; p.cnt = cnt;
    p.ldx = ldx; p.ldz = ldz; p.lda = lda; p.z_n0 = z_n0; p.N = N; p.lgH = H == 8 ? 3 : 4; p.K = K;
    p.nblocks = N * H * H / 256; p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    static const int rot = [] { const char* e = getenv("GCSSL_CT_ROTATE"); return (e && e[0] == '0') ? 0 : 1; }();
    p.rotate = rot;
    static const int dbg = [] { const char* e = getenv("GCSSL_CT_DEBUG"); return e ? atoi(e) : 0; }();
    p.debug = dbg;
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    const int grid = p.nblocks < cus ? p.nblocks : cus;
#define CT(T) do { if (H == 16) GCSSL_LAUNCH((convt_in_relu_kernel<T, 4>), dim3(grid), dim3(NTH), 0, (hipStream_t)stream, p); \
                   else GCSSL_LAUNCH((convt_in_relu_kernel<T, 3>), dim3(grid), dim3(NTH), 0, (hipStream_t)stream, p); } while (0)
    if (dtype == GCSSL_F16) CT(f16_t); else CT(bf16_t);
#undef CT
    return gcssl_launch_status();
}

}  // extern "C"
