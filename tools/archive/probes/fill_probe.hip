// LDS-DMA fill-rate probe: how many bytes per second can a CU's loader waves bring into LDS, with nothing else going on?
// Loader waves only (no consumers, no barriers): each workgroup fills a ring of 32-KB slots with 1-KB `buffer_load ... lds`
// pieces, `depth` K tiles in flight behind a counted vmcnt.  Source patterns:
//   0  D.c3.fwd's gather: x[768][8][8][128] fp16 (A rows = output pixels at tap offsets, 128-byte pieces) + Wf[256][16][128]
//   1  the same bytes, contiguous 32-KB blocks from a small (1 MB) buffer: every fill an L2 hit
//   2  contiguous 32-KB blocks streaming through a 512-MB buffer: every fill from HBM
// build: hipcc --offload-arch=gfx950 -O3 -o fill_probe tools/probes/fill_probe.hip ; run: ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_void_p;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int NW, int DEPTH, int PATTERN>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char* x, unsigned xbytes, const char* w, unsigned wbytes, int steps,
                                                      unsigned long long* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NSLOT = DEPTH + 1, STAGE = 32 * 1024, PIECES = 32 / NW;    // 1-KB pieces per wave and K tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * STAGE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, xbytes), wr = make_rsrc(w, wbytes);
    const int tile = blockIdx.x;                     // 192 tiles: 96 M tiles x 2 N tiles (D.c3.fwd at n = 768)
    const int m0 = (tile >> 1) * 128, n0 = (tile & 1) * 128;
    auto issue = [&](int t, int slot) {
        unsigned char* base = lds + slot * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = wave + NW * i;            // 0..31: 16 A pieces (128 rows), 16 B pieces (128 rows)
            const int row = (piece & 15) * 8 + (lane >> 3), chunk = lane & 7;
            unsigned off;
            if (PATTERN == 0) {
                const int tap = t >> 1, ci0 = (t & 1) * 64;
                if (piece < 16) {
                    const int m = m0 + row, n = m >> 4, oy = (m >> 2) & 3, ox = m & 3;
                    const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
                    off = ((unsigned)iy < 8u && (unsigned)ix < 8u) ? (unsigned)((((n * 8 + iy) * 8 + ix) * 128 + ci0 + chunk * 8) * 2) : OOB;
                } else {
                    off = (unsigned)((((n0 + row) * 16 + tap) * 128 + ci0 + chunk * 8) * 2);
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(piece < 16 ? xr : wr, (lds_void_p)(base + i * NW * 1024), 16, off, 0, 0, 0);
            } else {
                // contiguous: K tile t of this workgroup is one 32-KB block
                const unsigned blk = PATTERN == 1 ? (unsigned)((tile * 7 + t) & 31) : (unsigned)(tile * steps + t);
                off = blk * 32768u + piece * 1024u + lane * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_p)(base + i * NW * 1024), 16, off < xbytes ? off : OOB, 0, 0, 0);
            }
        }
    };
    for (int q = 0; q < DEPTH && q < steps; ++q) issue(q, q);
    int slot = 0;
    for (int t = 0; t < steps; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES <= 63 ? (DEPTH - 1) * PIECES : 63) : "memory");
        if (t + DEPTH < steps) issue(t + DEPTH, slot);          // refill the slot that just landed (nobody reads it)
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = lds[0];
#endif
}

template <int NW, int DEPTH, int PATTERN>
void run(const char* name, const char* x, unsigned xb, const char* w, unsigned wb, int wgs, int steps, unsigned long long* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<NW, DEPTH, PATTERN>), dim3(wgs), dim3(NW * 64), 0, 0, x, xb, w, wb, steps, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill_kernel<NW, DEPTH, PATTERN>), dim3(wgs), dim3(NW * 64), 0, 0, x, xb, w, wb, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)wgs * steps * 32768.0;
    printf("%-44s waves %d depth %d wgs %4d steps %3d: %7.1f us  %6.1f GB/s per WG  %6.2f TB/s chip\n", name, NW, DEPTH, wgs, steps, us,
           bytes / wgs / us / 1e3, bytes / us / 1e6);
}

int main() {
    const size_t xb = (size_t)768 * 8 * 8 * 128 * 2, wb = (size_t)256 * 16 * 128 * 2, big = (size_t)512 << 20;
    char *x, *w, *stream; unsigned long long* sink;
    hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&stream, big); hipMalloc(&sink, 4096 * 8);
    hipMemset(x, 1, xb); hipMemset(w, 1, wb); hipMemset(stream, 1, big);
    for (int wgs : {192, 256, 512}) {
        run<4, 3, 0>("conv gather (D.c3.fwd)", x, (unsigned)xb, w, (unsigned)wb, wgs > 192 ? 192 : wgs, 32, sink);
        if (wgs == 192) {
            run<4, 2, 0>("conv gather (D.c3.fwd)", x, (unsigned)xb, w, (unsigned)wb, 192, 32, sink);
            run<4, 4, 0>("conv gather (D.c3.fwd)", x, (unsigned)xb, w, (unsigned)wb, 192, 32, sink);
            run<8, 3, 0>("conv gather (D.c3.fwd)", x, (unsigned)xb, w, (unsigned)wb, 192, 32, sink);
            run<2, 3, 0>("conv gather (D.c3.fwd)", x, (unsigned)xb, w, (unsigned)wb, 192, 32, sink);
        }
        run<4, 3, 1>("contiguous 32-KB blocks, 1-MB source (L2)", x, 1u << 20, w, (unsigned)wb, wgs, 32, sink);
        run<4, 3, 2>("contiguous 32-KB blocks, streamed (HBM)", stream, (unsigned)(big - 1 < 0x7fffffff ? big - 1 : 0x7fffffff), w, (unsigned)wb, wgs, 32, sink);
    }
    run<4, 4, 1>("contiguous 32-KB blocks, 1-MB source (L2)", x, 1u << 20, w, (unsigned)wb, 256, 32, sink);
    run<8, 3, 1>("contiguous 32-KB blocks, 1-MB source (L2)", x, 1u << 20, w, (unsigned)wb, 256, 32, sink);
    hipDeviceSynchronize();
    return 0;
}
