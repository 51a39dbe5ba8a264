/* gcssl.h -- C ABI of libgcssl_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the hot path of
 * 1213ray/GAN-Calibrated-Semi-Supervised-Learning, i.e. the WGAN-GP cGAN training step
 * (reference: cgan/cgan_train_enhanced.py:304-369, cgan/models.py, cgan/losses.py).
 *
 * The reference has NO native/FFI boundary: its hot path is stock torch.nn ops.  Each entry
 * point below therefore cites the reference op (file:line) it replaces.  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer owned by the caller
 *     (including all workspaces); the library keeps no device state.
 *   - `stream` is a hipStream_t; kernels are only enqueued on it, never synchronised (graph-capture safe).
 *   - return value: 0 ok; <0 argument error (GCSSL_E*); >0 a hipError_t from the launch.  Never throws.
 *   - dtype: element type of activations/operands (GCSSL_F32 exact-fp32 MFMA; GCSSL_BF16 / GCSSL_F16 bf16 / IEEE-half
 *     MFMA operands with fp32 accumulation -- the same kernels and rate, fp16 trades exponent range for 3 more mantissa
 *     bits: BASELINE configs[3] "fp16 with fp32 loss accum").  Statistics, losses, weight gradients, master weights and
 *     optimiser state are always fp32.
 *   - activations are NHWC: tensor[n][y][x][c] with an explicit pixel stride `ld*` (elements) so a kernel can
 *     read/write a channel slice of a concat buffer in place; spatial sizes and channel counts are powers of 2.
 */
#ifndef GCSSL_H
#define GCSSL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GCSSL_OK 0
#define GCSSL_EBADSHAPE (-1)
#define GCSSL_EBADDTYPE (-2)
#define GCSSL_EALIGN (-3)
#define GCSSL_ENULL (-4)
#define GCSSL_F32 0
#define GCSSL_BF16 1
#define GCSSL_F16 2
/* Split-precision conv modes (round 4): tensors are fp32 in memory exactly as with GCSSL_F32 -- every non-conv entry point
 * is called with GCSSL_F32 -- and only the conv contractions differ: operands are split hi + lo into 16-bit halves on their
 * way into LDS and contracted with THREE 16-bit MFMAs per K step (hi*hi + lo*hi + hi*lo, fp32 accumulate): 22 (fp16) / 16
 * (bf16) mantissa bits per operand on the 2.5-PFLOP/s pipe instead of the 157-TFLOP/s exact-fp32 MFMA.  Accepted by
 * gcssl_conv4x4s2_{fwd,dgrad,wgrad}, their *_splits probes, gcssl_conv3x3_{fwd,wgrad} and the weight-pack entry points
 * (gcssl_prep_conv_weight[s], gcssl_conv3x3_prep_weights: a pack made with a split dtype carries the modes' 2^6 weight
 * pre-scale, which the conv epilogues undo -- pack and conv must be called with the same dtype).  GCSSL_F32_F16X3 needs its
 * operands inside fp16's range (the engine's static loss scales see to the gradients); nothing is clamped: an overflow is a
 * NaN, not a saturated value.  Reference arithmetic these modes reproduce to fp32 grade: cgan/models.py:222-258,
 * cgan/losses.py:185-233 (pure fp32). */
#define GCSSL_F32_F16X3 3
#define GCSSL_F32_BF16X3 4

/* Revision of THIS header's contract: argument lists and constants.  gcssl_abi_revision() of the loaded library must equal
 * it (round 3 changed a dozen argument lists in place under an unchanged version string: ADVICE r3). */
#define GCSSL_ABI_REVISION 5
int gcssl_abi_revision(void);
const char* gcssl_version(void);
/* The kernel template expression the most recent conv entry point of THIS process launched (as written at its launch site:
 * e.g. "(conv_dma_kernel<O, 128, 128, 0, 4, 2, false, 8, 4, false, true>)"; O / T = the operand type of `dtype`).  The
 * dispatchers choose an instantiation per shape; bench.py uses this to name the rocprofv3 symbol of its dominant launch. */
const char* gcssl_last_kernel(void);
/* ... and the number of workgroups of that launch (tools/prof_labels.py matches a label to its rocprofv3 launches by kernel
 * name AND grid size). */
int gcssl_last_grid(void);
/* One-time device-side set-up (dynamic-LDS opt-ins of the kernels that use > 64 KB).  Call once per process with a GPU
 * present and before capturing entry points into a hipGraph (they also do it lazily on first use, which a capture in
 * progress may refuse). */
int gcssl_init(void);
/* gcssl_init also allocates the ONE piece of device memory the library owns: the 16-byte barrier block of the cooperative
 * spectral-norm chain (with GCSSL_SN_COOP=1 gcssl_sn_power_iter runs a whole chain of power iterations as one launch whose
 * workgroups synchronise through it; the default is the multi-launch form, which measured faster: csrc/misc.hip).
 * gcssl_sn_coop_status: 0 fine; 1 a workgroup once gave up waiting at a grid barrier (bounded spin: results of that chain are
 * invalid -- it never happened in testing and means the grid was not co-resident); -1 not initialised.  Host-synchronous. */
int gcssl_sn_coop_status(void);
int gcssl_init_norm(void);
int gcssl_init_recrop(void);

/* ---- boundary packing ------------------------------------------------------------------------------------
 * torch.cat([pred_patch, other_patch], 1) (cgan/models.py:257): two NCHW fp32 (B,3,S,S) tensors -> NHWC
 * [B][S*S][8] (channels 0-2 = a, 3-5 = b (zeros if b==NULL), 6-7 = 0). */
int gcssl_pack_pair(int dtype, const float* a, const float* b, void* out, int B, int S, int reps, void* stream);
/* reps >= 1: the packed batch is written reps times, B*S*S pixels apart (the generator's input for the n_critic + 1
 * forwards of an iteration run as one batch: cgan/cgan_train_enhanced.py:311-312 and :348 read the same pred_patch). */
/* interpolation alpha*real + (1-alpha)*fake of both halves (cgan/losses.py:203-204), alpha: [B]. */
int gcssl_pack_interp(int dtype, const float* pred, const float* gt, const float* refined, const float* alpha,
                      void* out, int B, int S, void* stream);
/* both non-real groups of a critic step in one pass: out_fake = (pred, refined), out_interp as gcssl_pack_interp;
 * alpha NULL: drawn per sample from the counter-based hash keyed by (seed, counter[0], n) like gcssl_uniform_gen. */
int gcssl_pack_fake_interp(int dtype, const float* pred, const float* gt, const float* refined, const float* alpha,
                           unsigned long long seed, const double* counter, void* out_fake, void* out_interp, int B, int S,
                           void* stream);
/* gcssl_pack_fake_interp + gcssl_pack_pair(pred, gt -> out_real) of the same critic step as ONE launch; out_real nullable. */
int gcssl_pack_groups(int dtype, const float* pred, const float* gt, const float* refined, const float* alpha,
                      unsigned long long seed, const double* counter, void* out_real, void* out_fake, void* out_interp, int B, int S,
                      void* stream);
/* NHWC8 fp32 input-gradient -> the two NCHW (B,3,S,S) gradients torch.autograd.grad returns (cgan/losses.py:213). */
int gcssl_unpack_grad(const float* g, float* ga, float* gb, int B, int S, void* stream);

/* ---- weights -----------------------------------------------------------------------------------------------
 * fp32 PyTorch-layout conv weight [Cout][Cin][4][4] -> packed MFMA operand layouts in `dtype`:
 * wf [Cout][16][CinP] (forward GEMM) and wt [CinP][16][Cout] (dgrad GEMM); channels Cin..CinP-1 are zero.
 * For a ConvTranspose2d weight [CinT][CoutT][4][4] pass Cout=CinT, Cin=CoutT.  Either output may be NULL. */
int gcssl_prep_conv_weight(int dtype, const float* w, void* wf, void* wt, int Cout, int Cin, int CinP, void* stream);
/* the same for nl <= 8 layers in ONE launch (host arrays of device pointers / sizes). */
int gcssl_prep_conv_weights(int dtype, int nl, const float* const* w, void* const* wf, void* const* wt, const int* Cout,
                            const int* Cin, const int* CinP, const float* w5, float* w5p, int C5, void* stream);
/* w5 / w5p (both or neither): the critic head's weight rides along (== gcssl_prep_c5_weight(w5, w5p, C5)). */
/* critic head weight [1][C][4][4] -> fp32 [16][C]. */
int gcssl_prep_c5_weight(const float* w, float* wp, int C, void* stream);

/* ---- Conv2d(k4,s2,p1): cgan/models.py:57 (G.down*), :236 (D.c1-c4) -------------------------------------------
 * y[n,oy,ox,co] = act( gscale[n/group_n] * conv(x, W)[...] + bias[co] );  x [N][Hi][Wi][ldx>=Cin], y [N][Hi/2][Wi/2][ldy>=Cout].
 * bias, gscale nullable; act: 0 none, 1 LeakyReLU(0.2) (cgan/models.py:60,242).  gscale carries 1/sigma of the
 * spectral norm (cgan/models.py:237-238) per sample group.  Also serves the data-gradient of ConvTranspose2d.
 * out_f32: write y as fp32 whatever dtype (pre-InstanceNorm tensors are always fp32, see gcssl_in_act_fwd). */
int gcssl_conv4x4s2_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias, const float* gscale,
                        int group_n, void* y, int ldy, int N, int Hi, int Wi, int Cin, int Cout, int act, int out_f32,
                        long split_stride, void* stream);
/* K split the dispatcher uses for these shapes (>= 1).  With split_stride > 0 the ks partial sums are stored plainly in
 * fp32 slabs y + k*split_stride (no memset, no atomics: float atomics run at 1.3 TB/s chip-wide) and the consumer
 * adds them (gcssl_in_act_fwd nslab); with split_stride = 0 they are added atomically into y, which the call zeroes. */
int gcssl_conv4x4s2_fwd_splits(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int act, int out_f32);
/* Split-precision modes (GCSSL_F32_F16X3 / GCSSL_F32_BF16X3; fp32 tensors): nn.Conv2d(k4,s2,p1) + InstanceNorm2d + LeakyReLU(0.2) of
 * cgan/models.py:236-242 / :57-60 in ONE launch -- a (fp32 activation), mean / rstd ([N][Cout] fp32) and, when z is not NULL, the
 * fp32 pre-norm values too (the fp32 backward kernels read them).  _ok: 1 if the shapes are served (whole samples per 128-row
 * tile: H*W/4 <= 64 output pixels; enough tiles without a K split), else 0 -- the caller then keeps gcssl_conv4x4s2_fwd +
 * gcssl_in_act_fwd. */
int gcssl_conv4x4s2_in_act_x3_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout);
int gcssl_conv4x4s2_in_act_x3_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias, const float* gscale, int group_n,
                                  float* z, int ldz, float* a, int lda, float* mean, float* rstd, int N, int Hi, int Wi, int Cin,
                                  int Cout, void* stream);
/* Conv2d(k4,s2,p1) + InstanceNorm2d + LeakyReLU(0.2) [+ Dropout(0.5)] as ONE launch -- the whole `conv_block` of
 * cgan/models.py:236-242 (D.c2-c4) / `UNetDown` of :54-66 (G.down2-4) -- for 16-bit dtypes and output maps of <= 64 pixels:
 * a GEMM tile then holds whole samples, the per-(n, c) statistics are taken from the fp32 accumulators inside the conv
 * epilogue and the fp32 pre-norm tensor never goes to memory.  a [N][Hi/2][Wi/2][lda>=Cout] = act((z - mean) * rstd)
 * [* 2 * mask]; mean, rstd [N][Cout] fp32.  apre (nullable): the activation without the dropout mask for samples
 * n >= apre_n0, [N - apre_n0][Hi/2 * Wi/2][ld_apre] -- what gcssl_in_act_bwd(z_kind = 1) rebuilds xhat from when `a` itself is
 * masked.  act must be 1.  gcssl_conv4x4s2_in_act_ok: 1 if these shapes are served, 0 if the caller should use
 * gcssl_conv4x4s2_fwd (out_f32) + gcssl_in_act_fwd instead (fp32 dtype, larger maps, too few tiles without a K split). */
int gcssl_conv4x4s2_in_act_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout);
int gcssl_conv4x4s2_in_act_fwd(int dtype, const void* x, int ldx, const void* wf, const float* bias, const float* gscale,
                               int group_n, void* a, int lda, float* mean, float* rstd, const uint8_t* mask, void* apre,
                               int ld_apre, int apre_n0, int N, int Hi, int Wi, int Cin, int Cout, int act, void* stream);
/* data gradient of the conv == ConvTranspose2d(k4,s2,p1) forward (cgan/models.py:72,113):
 * dx[N][Hi][Wi][lddx>=Cin] = gscale * convT(dy[N][Hi/2][Wi/2][lddy>=Cout], W).  out_f32: write fp32 whatever dtype. */
int gcssl_conv4x4s2_dgrad(int dtype, const void* dy, int lddy, const void* wt, const float* gscale, int group_n,
                          void* dx, int lddx, int N, int Hi, int Wi, int Cin, int Cout, int out_f32,
                          long split_stride, void* stream);
int gcssl_conv4x4s2_dgrad_splits(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int out_f32);
/* The same data gradient with the activation backward of the norm-less layer in FRONT of the conv (D.c1 / G.down1:
 * Conv + LeakyReLU without InstanceNorm, cgan/models.py:103,246) in its epilogue -- gcssl_conv4x4s2_dgrad (fp32 dx) followed by
 * gcssl_act_bwd as one launch, 16-bit dtypes, Cin == 64: dzs[N][Hi][Wi][lddz>=Cin] = lrelu'(a) * dx * gscale[n/group_n] in
 * `dtype`, with a [N][Hi][Wi][lda] that layer's stored activation; dbias / cdot / nrep / rep_stride / sat as in gcssl_act_bwd
 * (bias: that layer's conv bias).  _ok: 1 if the shapes are served (with_sums: dbias or cdot requested; only the persistent
 * form carries them), 0 if the caller should use the two launches. */
int gcssl_conv4x4s2_dgrad_act_bwd_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout, int with_sums);
int gcssl_conv4x4s2_dgrad_act_bwd(int dtype, const void* dy, int lddy, const void* wt, const void* a, int lda, const float* gscale,
                                  int group_n, const float* bias, void* dzs, int lddz, float* dbias, float* cdot, int nrep,
                                  int rep_stride, unsigned int* sat, int N, int Hi, int Wi, int Cin, int Cout, void* stream);
/* gcssl_conv4x4s2_fwd of the 8-channel first layer (Cin 8 -> Cout 64) as the forward of the double-backward gradient-penalty
 * chain (the create_graph=True part of d_loss.backward(), cgan/cgan_train_enhanced.py:330, through D.c1 = Conv + LeakyReLU,
 * cgan/models.py:246), with gcssl_act_bwd and gcssl_dot_accum in its epilogue, 16-bit dtypes: with v = gscale[n/group_n] *
 * conv(x, W) (fp32, never stored), y[N][Hi/2][Wi/2][ldy>=Cout] = lrelu'(a) * v in `dtype` (a: that layer's stored activation,
 * same layout, lda) and, if dotx is given, *dot_out += sum dotx * v.  sat as in gcssl_act_bwd.  _ok: 1 if served, 0 if the
 * caller should use the three launches. */
int gcssl_conv4x4s2_fwd_act_bwd_ok(int dtype, int N, int Hi, int Wi, int Cin, int Cout);
int gcssl_conv4x4s2_fwd_act_bwd(int dtype, const void* x, int ldx, const void* wf, const float* gscale, int group_n, const void* a,
                                int lda, void* y, int ldy, const void* dotx, int lddot, float* dot_out, unsigned int* sat, int N,
                                int Hi, int Wi, int Cin, int Cout, void* stream);
/* weight gradient: slab[s][Cout][16][Cin] (fp32, s < gcssl_conv4x4s2_wgrad_splits(...)) partial sums over the
 * s-th K range of sum_{n,oy,ox} dy[n,oy,ox,co] x[n,2oy-1+ky,2ox-1+kx,ci].  Cin is 8 (padded first layer) or >= 64. */
int gcssl_conv4x4s2_wgrad_splits(int N, int Hi, int Wi, int Cin, int Cout);
int gcssl_conv4x4s2_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* slab, int N, int Hi,
                          int Wi, int Cin, int Cout, void* stream);
/* Up to three layers' weight gradients as ONE launch (a backward pass's weight gradients do not depend on each other; arguments
 * per layer as gcssl_conv4x4s2_wgrad's, nl in 1..3).  One grid when every layer takes the filter-row LDS-DMA kernel (16-bit
 * dtypes, the critic's c2-c4 / the generator's up2-up4 at the bench batch); otherwise identical to nl single calls. */
int gcssl_conv4x4s2_wgrad_batch(int dtype, int nl, const void* const* x, const int* ldx, const void* const* dy, const int* lddy,
                                float* const* slab, const int* N, const int* Hi, const int* Wi, const int* Cin, const int* Cout,
                                void* stream);
/* dw[Cout][Cin_real][4][4] (=|+=) sum_s slab[s] - sum_k coef[k]*cscale[k] u_k[co] v_k[ci*16+tap]: split-K reduction
 * fused with the spectral-norm quotient rule d(W/sigma) (sigma = u^T W v; cgan/models.py:237-238).
 * u: nrank rows of stride ustride (>= Cout); v: nrank rows of stride vstride (>= Cin_real*16); cscale nullable.
 * accumulate: 0 dw = result; 1 dw += result; 2 dw was zeroed by the caller and may be accumulated atomically (lets the
 * reduction run in parallel over groups of slabs). */
int gcssl_wgrad_reduce(const float* slab, int nsplit, float* dw, int Cout, int Cin, int Cin_real, const float* coef,
                       const float* cscale, const float* u, int ustride, const float* v, int vstride, int nrank,
                       int accumulate, void* stream);
/* The same reduction for nl <= 8 layers in one launch (all weight gradients of one loss.backward(),
 * cgan/cgan_train_enhanced.py:330,366); arrays are indexed by layer, nrank/strides/accumulate are shared. */
int gcssl_wgrad_reduce_batch(int nl, const float* const* slab, const int* nsplit, float* const* dw, const int* Cout,
                             const int* Cin, const int* Cin_real, const float* const* coef, const float* const* u,
                             const float* const* v, int ustride, int vstride, int nrank, int accumulate,
                             const float* const* coef_rep, const float* const* bias_rep, float* const* dbias, int nrep,
                             int rep_stride, void* stream);
/* coef_rep / bias_rep (nullable arrays of nullable pointers): replica 0 of the striped partial sums the backward kernels
 * left (gcssl_in_act_bwd / gcssl_act_bwd with nrep > 1): coef[i][k] + sum_r coef_rep[i][k + r*rep_stride] is the coefficient
 * used, and dbias[i][c] = sum_r bias_rep[i][c + r*rep_stride] (c < Cout[i]) is stored by tail workgroups -- the fold of
 * gcssl_sum_replicas without a launch of its own. */

/* ---- critic head Conv2d(512,1,k4,s1,p1,bias=False): cgan/models.py:252 ------------------------------------------
 * out [N][Hi-1][Wi-1] fp32.  dgrad/wgrad take either a dout tensor or per-group constants g0,g1,g2 (dout==NULL):
 * the WGAN seeds -1/(B hw), +1/(B hw) of cgan/cgan_train_enhanced.py:327-328 and the ones of cgan/losses.py:216. */
int gcssl_conv4x4s1_c1_fwd(int dtype, const void* x, int ldx, const float* wp, float* out, float* group_mean, int groups,
                           int N, int Hi, int Wi, int C, void* stream);
/* group_mean (nullable, `groups` floats, zeroed by the caller): += the mean of each of `groups` equal chunks of `out` -- the
 * batch means of cgan/cgan_train_enhanced.py:327 / :362 without a launch of their own (== gcssl_group_mean afterwards). */
int gcssl_conv4x4s1_c1_dgrad(int dtype, const float* dout, float g0, float g1, float g2, float g3, int group_n, const float* wp,
                             void* dx, int lddx, int N, int Hi, int Wi, int C, void* stream);
int gcssl_conv4x4s1_c1_wgrad(int dtype, const void* x, int ldx, const float* dout, float g0, float g1, float g2, float g3, int group_n,
                             float* dw, int N, int Hi, int Wi, int C, void* stream);   /* dw[C][16] += (atomic) */

/* ---- InstanceNorm2d(affine=False, eps=1e-5) + activation (+Dropout): cgan/models.py:59-63,73-76,114,241-242 -----
 * act: 1 LeakyReLU(0.2), 2 ReLU.  mask: dropout keep mask [N][HW][C] (uint8) or NULL; kept values are scaled by 2.
 * The pre-norm tensor z and every incoming gradient (da, da2, gb_a, qz, zt) are ALWAYS fp32 (z - mean(z) and
 * dn - mean(dn) over 4..64 elements cancel a bf16 mantissa); tensors that feed an MFMA (a, dzs, gt_a, gb_zs) are `dtype`. */
int gcssl_in_act_fwd(int dtype, float* z, int ldz, void* a, int lda, float* mean, float* rstd, const uint8_t* mask,
                     float* pool, int nslab, long slab_stride, int N, int HW, int C, int act, void* stream);
/* pool (nullable): [N][C] fp32, += sum over H*W of the activation output (AdaptiveAvgPool2d(1) of cgan/models.py:118,
 * fused; caller zeroes it and divides by H*W).  a may be NULL when pool is given and 256 < HW <= 1024, C % 32 == 0, nslab == 1:
 * only mean / rstd / pool are produced (GCSSL_ENULL for a NULL a elsewhere). */
/* first-order backward: dn = act'(xhat) (da + da2 + da_bcast) [*2 keep]; dz = rstd (dn - mean dn - xhat mean(dn xhat))
 * (+ zt for samples n >= zt_n0: the double-backward term); dzs = dz * gscale[n/group_n];
 * dbias[c] += sum dz; cdot[n/group_n] += sum dzs (z - bias[c]) -- the coefficient <dW_sn, W_orig>/sigma^2 of the
 * spectral-norm quotient rule when gscale = 1/sigma.  Optional args nullable. */
int gcssl_in_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const float* da_bcast,
                     const void* z, int ldz, int z_kind, const float* mean, const float* rstd, const uint8_t* mask, const float* zt,
                     int zt_n0, const float* gscale, int group_n, const float* bias, void* dzs, int lddz, float* dbias,
                     float* cdot, int nrep, int rep_stride, int da_nslab, long da_slab_stride,
                     float* ws, const float* presum_cnt, const float* presum_pos, float presum_pos_scale, unsigned int* sat,
                     int N, int HW, int C, int act, void* stream);
/* sat (nullable, device): += the number of output values whose magnitude exceeded fp16's largest finite number and were
 * clipped by the store (dtype GCSSL_F16 only; the static loss scale is meant to keep this at zero -- the engine exposes the
 * count, bench.py reports it, the tests assert 0).  Same argument on gcssl_in_dbl_bwd (gt_a), gcssl_act_bwd and gcssl_gp_norm.
 * z_kind 0: z is the fp32 pre-norm tensor.  z_kind 1 (maps of <= 256 pixels, act = LeakyReLU): z is the 16-bit ACTIVATION
 * without dropout, in `dtype`, as gcssl_conv4x4s2_in_act_fwd left it (its `a`, or `apre` for a masked layer); xhat is rebuilt as
 * a > 0 ? a : 5 a and z - bias as xhat / rstd + mean - bias.
 * ws: caller-owned scratch of 2*N*C floats, required when H*W > 256 (two-kernel path), else may be NULL.
 * presum_cnt / presum_pos (nullable, [N][C] fp32; only with act = ReLU and da_bcast as the sole incoming gradient): the
 * number of positive normalised values and presum_pos_scale * presum_pos = their sum, as the forward pass left them
 * (gcssl_convT4x4s2_in_relu_fwd's cnt, and its pool or the head's pooled mean with scale H*W): maps of more than 256
 * pixels then skip the statistics pass over z and need no ws. */
/* second-order backward (create_graph=True, cgan/losses.py:213-220): adjoint of dz=IN_bwd(z, act'*gb_a) for an
 * incoming adjoint qz: gt_a = act'(xhat) * d/d(dn), zt = d/dz; cdot += sum gb_zs*qz. */
int gcssl_in_dbl_bwd(int dtype, const float* gb_a, int ldgb, const float* qz, int ldq, const void* gb_zs, int ldgz,
                     const void* z, int ldz, int z_kind, const float* mean, const float* rstd, void* gt_a, int ldga, float* zt,
                     float* cdot, int q_nslab, long q_slab_stride, unsigned int* sat, int N, int HW, int C, int act, void* stream);
/* z_kind as in gcssl_in_act_bwd (1: maps of <= 64 pixels).
 * da_nslab / q_nslab > 1: da / qz is the first of that many split-K partial-sum slabs (stride in floats) written by a
 * gcssl_conv4x4s2_* call with split_stride > 0; the kernel first folds them into slab 0 (maps up to 16x16 / 8x8), so
 * slab 0 holds the total afterwards (gcssl_in_dbl_bwd reads the same da again as gb_a). */
/* LeakyReLU backward for the norm-less layers (D.c1, G.down1; cgan/models.py:103,246), from the activation OUTPUT a. */
int gcssl_act_bwd(int dtype, const float* da, int ldda, const float* da2, int ldda2, const void* a, int lda,
                  const float* gscale, int group_n, const float* bias, void* dzs, int lddz, float* dbias, float* cdot,
                  int nrep, int rep_stride, unsigned int* sat, const void* dotx, int lddot, float* dot_out,
                  int N, int HW, int C, void* stream);
/* dotx / dot_out (both or neither; not with da2): dot_out += sum dotx * da with dotx [N][HW][lddot>=C] in `dtype` -- gcssl_dot_accum
 * of the same operands folded into this pass (the <gb_zs, gt_z> spectral-norm term of the norm-less first layer). */
/* dbias / cdot of the two backward entry points above may be striped: nrep replicas, rep_stride floats apart, each
 * workgroup adds to one of them (same-address float atomics serialise); nrep = 1 is the plain form.  This folds them:
 * dst[i][j] (=|+=) sum_r src[i][j + r*rep_stride] for nseg <= 8 segments of len[i] floats; bit i of accumulate
 * selects += for segment i. */
int gcssl_sum_replicas(int nseg, const float* const* src, float* const* dst, const int* len, int nrep, int rep_stride,
                       int accumulate, void* stream);
int gcssl_dot_accum(int dtype, const void* x, int ldx, const float* y, int ldy, long pixels, int C, float* out, void* stream);

/* ---- spectral norm power iteration (torch.nn.utils.spectral_norm, cgan/models.py:237-238) -----------------------
 * v <- normalize(W^T u), u <- normalize(W v), eps 1e-12; sigma = u.(W v).  nl <= 4 layers per call.  iterate = k >= 1 runs k
 * iterations back to back on the same weights (a critic step's real, fake and interpolated forwards each make one: :308,
 * :316, cgan/losses.py:210) and fills history slots slot .. slot+k-1 of u_hist/v_hist/sigma/isig; iterate = 0 only evaluates
 * sigma into `slot` (eval mode).  t[i] (2 * cols[i] floats) must be ZERO on entry and is left zero on exit; s[i] is rows[i]
 * floats of scratch.  zero / nzero (nullable): nzero floats the call also clears -- the engine's per-step scalar block rides
 * on the closing launch instead of a fill of its own. */
int gcssl_sn_power_iter(int nl, const float* const* w, float* const* u, float* const* v, float* const* t, float* const* s,
                        const int* rows, const int* cols, float* sigma, float* isig, float* u_hist, float* v_hist,
                        int hist_stride_u, int hist_stride_v, int slot, int nslots, int iterate, float* zero, long nzero,
                        void* stream);
/* One launch less per chain: gcssl_sn_defer_finish(1) asks the NEXT gcssl_sn_power_iter(iterate > 0) of this thread to leave its
 * closing step (u = s/|s|, sigma, the extra fill: 4 workgroups) pending; the NEXT gcssl_prep_conv_weights launch then carries it as
 * an extra grid row -- the re-pack reads nothing the chain writes, and both must finish before the first conv of the forward.
 * gcssl_sn_flush_finish launches a pending step on its own (a caller that deferred and does not re-pack after all). */
int gcssl_sn_defer_finish(int on);
int gcssl_sn_flush_finish(void* stream);
/* The same for gcssl_conv4x4s1_c1_dgrad's constant-dout form (the only one the step engine launches): recorded, not launched; the next
 * gcssl_prep_conv_weights launch of this thread carries it (it reads the RAW head weight w [1][512][4][4], not the packed copy that
 * launch writes; dx fp32, C = 512).  gcssl_sn_flush_finish launches a pending one on its own. */
int gcssl_conv4x4s1_c1_dgrad_defer(float g0, float g1, float g2, float g3, int group_n, const float* w, float* dx, int lddx,
                                   int N, int Hi, int Wi, int C);

/* ---- fused generator up-path layer (cgan/models.py:72-74,112-118) ---------------------------------------------------------
 * ConvTranspose2d(K -> 64, k4 s2 p1, bias=False) + InstanceNorm2d + ReLU (+ the sums AdaptiveAvgPool2d(1) needs) as ONE
 * launch: the pixel-stationary kernel of csrc/convt_fused.hip (inputs of 8x8 or 16x16 pixels; 16-bit dtypes only).
 * x [N][H][H][ldx>=K]; wt = the dgrad pack Wt[64][16][K] of gcssl_prep_conv_weight.  Outputs, each nullable except
 * mean/rstd [N][64]: a [N][2H][2H][lda] in `dtype`; z32 = the fp32 pre-norm values of samples >= z_n0 only (what
 * gcssl_in_act_bwd reads for the samples that have a backward pass); pool [N][64] = sum over the output pixels of the
 * activation (written, not accumulated); cnt [N][64] (needs pool) = how many of them are positive (written). */
int gcssl_convT4x4s2_in_relu_fwd(int dtype, const void* x, int ldx, const void* wt, float* z32, int ldz, int z_n0, void* a,
                                 int lda, float* mean, float* rstd, float* pool, float* cnt, int N, int H, int K, int Cout,
                                 void* stream);

/* ---- gradient penalty (cgan/losses.py:223-231) -------------------------------------------------------------------
 * nrm[b] = sqrt(sum g_b^2 + 1e-12); gp_sum += mean((nrm-1)^2); coef[b] = lambda_gp*2/B*(nrm-1)/nrm. */
int gcssl_gp_norm(const float* g, long per_sample, int B, float lambda_gp, float* nrm, float* coef, float* gp_sum,
                  int dtype, void* scaled, unsigned int* sat, void* stream);   /* scaled (nullable, `dtype`): g * coef[n], the reverse pass's seed */
int gcssl_scale_rows(int dtype, const float* x, const float* coef, void* y, long per_sample, int B, void* stream);

/* ---- clip_grad_norm_(1.0) + Adam (cgan/cgan_train_enhanced.py:256-257,331-332,368-369) over flat fp32 buffers ---
 * state: GCSSL_ADAM_STATE doubles, zeroed once by the caller: {step, -, last total norm, clip coef, lr/(1-b1^t),
 * sqrt(1-b2^t), -, lr override, then the call's <= 256 partial sums of squares}; the step is advanced on the device
 * (graph replay safe); a positive state[7] replaces `lr` (what ReduceLROnPlateau at
 * cgan/cgan_train_enhanced.py:260-261,427-428 changes).
 * write_clipped: 0 leave g, 1 store g*clip_coef (what clip_grad_norm_ leaves in .grad), 2 zero g (fused zero_grad).
 * grad_scale (> 0): the optimiser sees g * grad_scale -- 1/world_size after a data-parallel SUM all-reduce, 1 otherwise;
 * norm, clip coefficient and the written-back gradient are those of the scaled gradient.
 * p, g, m, v must be 16-byte aligned. */
#define GCSSL_ADAM_STATE 264
int gcssl_clip_adam(float* p, float* g, float* m, float* v, long n, double* state, double lr, double b1, double b2,
                    double eps, double max_norm, int write_clipped, double grad_scale, void* stream);

/* ---- generator head (cgan/models.py:118-123,139-141) and box/EIoU loss (cgan/losses.py:19-73,99-150) ------------- */
/* pool_sum (nullable): [B][64] sums over H*W already accumulated by gcssl_in_act_fwd(pool=...); then x is not read and
 * pool_sum is left ZERO for the next accumulation. */
int gcssl_pool_fc_tanh_fwd(int dtype, const void* x, int ldx, float* pool_sum, const float* w, const float* bias,
                           float scale, float* pooled, float* traw, float* delta, int B, int HW, int C, void* stream);
/* dw [4][64] and db [4] are accumulated atomically (caller zeroes them). */
int gcssl_head_bwd(const float* g_delta, const float* traw, const float* pooled, const float* w, float scale, int B,
                   int HW, float* dw, float* db, float* da_bcast, void* stream);
/* EIoU of apply_delta_to_bbox(pred_box, delta) against apply_delta_to_bbox(pred_box, delta_true) (training form): *loss_acc =
 * -mean(eiou) (STORED: one workgroup walks the batch, no atomics, no fill in front), calibrated = the boxes,
 * g_delta = lambda_iou * d(1 - mean eiou)/d delta. */
int gcssl_eiou_fwd_bwd(const float* pred_box, const float* delta, const float* delta_true, int B, float lambda_iou,
                       float* g_delta, float* calibrated, float* loss_acc, void* stream);

/* ---- re-crop stage (next row f1): get_refined_patch_batch, cgan/cgan_train_enhanced.py:37-137 ---------------------
 * atlas: device bytes of RGB (HWC, uint8) images; image i starts at img_off[i] and is img_w[i] x img_h[i]; sample n
 * uses image img_idx[n].  refined_box / pred_box: [B][4] normalised (cx,cy,w,h) fp32 (refined = apply_delta_to_bbox(
 * pred, delta, training=False), e.g. from gcssl_apply_delta_eval).  out: [B][3][S][S] fp32 in [-1,1].  Per sample:
 * clamp, crop (fallback to pred_box when the refined crop is invalid, :95-104), grey padding to a square, Pillow 12.2
 * BICUBIC resize (bit-exact), ToTensor + Normalize(0.5,0.5).  status (nullable) [B]: 0 ok, 1 predicted box used,
 * 2 failed (the reference's except branch: fallback[n] if given, else zeros).  max_side: upper bound of any crop side
 * in pixels (sizes the LDS and the coefficient workspace; the largest image side covers every box).  atlas_bytes: size
 * of the atlas (< 2^31).  ws: gcssl_recrop_ws_ints(B, S, max_side) ints of scratch (per-sample coefficient tables).
 * mode 1 = the dataset's CalibratorDataset._letterbox (cgan/dataset.py:104-124): crop refined_box as given (no clamp, no
 * validity test, no fallback; pred_box unused), pad, resize, normalise -- the pred/gt training patches themselves.
 * mode 2 = inference.py's crop_patch + letterbox (cgan/inference.py:51-68): as mode 1 with the crop edges rounded to
 * nearest (Image.crop on float coordinates) instead of truncated. */
int gcssl_recrop_ws_ints(int B, int S, int max_side);
int gcssl_recrop_patches(const uint8_t* atlas, long atlas_bytes, const long* img_off, const int* img_w, const int* img_h,
                         const int* img_idx, const float* refined_box, const float* pred_box, const float* fallback,
                         float* out, int* status, int* ws, int B, int S, int max_side, int mode, void* stream);
/* apply_delta_to_bbox(bbox, delta, training=False), cgan/losses.py:108-150, fp32. */
int gcssl_apply_delta_eval(const float* box, const float* delta, float* out, int B, void* stream);

/* ---- misc ---------------------------------------------------------------------------------------------------------- */
/* Bernoulli(0.5) keep-masks (nn.Dropout(0.5), cgan/models.py:106,109,110) and uniform [0,1) floats (torch.rand alpha,
 * cgan/losses.py:199) from a counter-based hash keyed by (seed, counter[0], index); counter lives on the device so a
 * replayed hipGraph draws fresh values; out of the mask generator must be 8-byte aligned. */
int gcssl_dropout_mask_gen(uint8_t* out, long n, unsigned long long seed, const double* counter, void* stream);
int gcssl_uniform_gen(float* out, long n, unsigned long long seed, const double* counter, void* stream);
int gcssl_group_mean(const float* x, int groups, int per_group, float* out, void* stream);
int gcssl_cast(int dtype, const float* x, void* y, long n, void* stream);
int gcssl_uncast(int dtype, const void* x, float* y, long n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GeneratorSimpleRegressor (generator_type "simple", cgan/models.py:147-216; selected at cgan/cgan_train_enhanced.py:26-31)
 * ------------------------------------------------------------------------------------------------------------------ */
/* nn.Conv2d(Cin, Cout, 3, padding=1) (models.py:163-193) as an implicit GEMM on MFMA: x [N][H][W][ldx>=Cin] in `dtype`,
 * w = packed rows of gcssl_conv3x3_wk(Cin) elements (gcssl_conv3x3_prep_weights), y [N][H][W][ldy>=Cout] (fp32 when
 * out_f32 or dtype is fp32).  The data gradient is this same call on dy with the transposed-and-rotated pack `wt` and the
 * channel counts swapped.  H, W, Cin (>= 8), Cout (>= 64) powers of two. */
int gcssl_conv3x3_wk(int C);
int gcssl_conv3x3_fwd(int dtype, const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                      int N, int H, int W, int Cin, int Cout, int out_f32, void* stream);
/* weight gradient: split-K fp32 slabs [splits][Cout][16][Cin] (taps 0..8 of the 16-tap slab layout are meaningful),
 * summed into the PyTorch layout dw[Cout][Cin_real][3][3] by gcssl_conv3x3_wgrad_reduce (up to 8 layers per launch). */
int gcssl_conv3x3_wgrad_splits(int N, int H, int W, int Cin, int Cout);
int gcssl_conv3x3_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* slab, int N, int H, int W,
                        int Cin, int Cout, void* stream);
int gcssl_conv3x3_wgrad_reduce(int nl, const float* const* slab, const int* nsplit, float* const* dw, const int* Cout,
                               const int* Cin, const int* Cin_real, void* stream);
/* w[i]: fp32 [Cout][Cin][3][3] -> wf[i] [Cout][wk(CinP)] (k = tap*CinP + ci) and wt[i] [Cin][wk(Cout)] (k = (8-tap)*Cout + co);
 * either may be NULL. */
int gcssl_conv3x3_prep_weights(int dtype, int nl, const float* const* w, void* const* wf, void* const* wt, const int* Cout,
                               const int* Cin, const int* CinP, void* stream);
/* nn.MaxPool2d(2,2) (models.py:169,178,187,196).  Backward routes each window's gradient to its first maximum; dpool is
 * [N][H/2][W/2][C] fp32, or with bcast != 0 a per-sample vector [N][C] multiplied by bscale (AdaptiveAvgPool2d backward). */
int gcssl_maxpool2_fwd(int dtype, const void* a, int lda, void* o, int ldo, int N, int H, int W, int C, void* stream);
int gcssl_maxpool2_bwd(int dtype, const void* a, int lda, const float* dpool, int ldd, int bcast, float bscale, float* da,
                       int ldda, int N, int H, int W, int C, void* stream);
/* nn.AdaptiveAvgPool2d(1) + Flatten (models.py:201-202): feat[N][C] fp32 */
int gcssl_avgpool_fwd(int dtype, const void* x, int ldx, float* feat, int N, int HW, int C, void* stream);
/* regressor (models.py:203-216): Linear(512,256)+ReLU+Dropout, Linear(256,64)+ReLU+Dropout, Linear(64,4), Tanh, * delta_scale.
 * The forward takes the first two weights TRANSPOSED (w1t [512][256], w2t [256][64]: one thread per output neuron,
 * coalesced rows), the backward the nn.Linear layouts.  m1 [N][256], m2 [N][64]: Dropout(0.5) keep masks (bytes), both
 * NULL in eval mode.  h1/h2: post-dropout activations kept for the backward; traw = tanh output. */
int gcssl_mlp_head_fwd(const float* feat, const float* w1t, const float* b1, const float* w2t, const float* b2, const float* w3,
                       const float* b3, const uint8_t* m1, const uint8_t* m2, float delta_scale, float* h1, float* h2,
                       float* traw, float* delta, int N, void* stream);
int gcssl_mlp_head_bwd(const float* gdelta, const float* traw, const float* h1, const float* h2, const float* feat,
                       const float* w1, const float* w2, const float* w3, float delta_scale, int train, float* dp1, float* dp2,
                       float* dp3, float* dfeat, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, int N,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
