/*
 * libmtdgan_hip.so -- C ABI of the MI355X (gfx950) kernels behind the MTD-GAN training hot path.
 *
 * The reference (babbu3682/MTD-GAN) is pure Python: it has no FFI of its own, every op below is an
 * ATen call made from the reference file:line cited at the entry point that replaces it.  This header
 * is therefore the boundary a maintainer binds with ctypes/cffi; the host-side
 * mirror of the reference's module surface lives in mtd-gan_amd/ (arch/Ours/networks.py etc.).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; activations are NHWC with an explicit
 *     leading dimension `ld` (floats between consecutive pixels, >= channels) so channel slices of a
 *     wider tensor can be read/written in place; weights keep PyTorch's OIHW (Conv2d) / IOHW
 *     (ConvTranspose2d) layout and are addressed through (w_sn, w_sc) element strides;
 *   - the library never allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*); workspaces are sized by the *_ws_bytes helpers;
 *   - every entry point returns 0 on success, a negative MTD_E* code for argument errors, or a
 *     positive hipError_t from the launch.
 */
#ifndef MTDGAN_HIP_H
#define MTDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTD_OK 0
#define MTD_EINVAL (-1)   /* bad shape / null pointer / unsupported combination */
#define MTD_EALIGN (-2)   /* pointer or leading dimension not 16-byte aligned where required */
#define MTD_EWS (-3)      /* workspace too small */

#define MTD_ACT_NONE 0
#define MTD_ACT_RELU 1
#define MTD_ACT_LRELU 2   /* LeakyReLU(0.2) -- arch/Ours/networks.py:182 etc. */
/* Round 4: relu(acc * scale + bias) + add1 + add2 -- the residual operands added AFTER the activation: x + relu(conv3x3(x) + b)
 * of FFT_ConvBlock.forward (networks.py:31-35) in one launch for whole-slice inference, where the block's third term then
 * needs one operand less.  Only where mtd_conv_relu_add_ok() says so (mtd_conv_igemm on the generator-shaped 32-channel 3x3
 * layers whose plan is the persistent kernel; no mask, no out2); every other entry point answers MTD_EINVAL to it. */
#define MTD_ACT_RELU_ADD 3

/* Gather geometry shared by conv forward, data-gradient and weight-gradient kernels.
 * For launch-grid pixel (b, oy, ox), tap (ty, tx):
 *     iy = oy*in_sy + off_y + ty*tap_dy        ix = ox*in_sx + off_x + tx*tap_dx
 *     kidx = (ky0 + ty*ky_step)*KW + (kx0 + tx*kx_step)          (index into the kh*kw plane)
 * and the result pixel is written at (oy*out_sy + out_oy, ox*out_sx + out_ox) of an OHF x OWF image.
 *   conv fwd  (k,s,p): in_s=s, off=-p, tap_d=+1, TH=TW=k, ky0=0, ky_step=1, out_s=1, out_o=0
 *   dgrad s=1 (k,p)  : in_s=1, off=+p, tap_d=-1, TH=TW=k               (also ConvTranspose2d fwd)
 *   dgrad s=2 k=4 p=1: one launch per output parity (py,px): TH=TW=2, ky0=(py+1)&1, ky_step=2,
 *                      off=(py+1-ky0)/2, tap_d=-1, out_s=2, out_o=py */
typedef struct {
    int B, IH, IW;          /* gathered tensor: batch, height, width                      */
    int OH, OW;             /* launch grid (pixels enumerated per image)                  */
    int in_sy, in_sx, off_y, off_x, tap_dy, tap_dx;
    int TH, TW, KW;
    int ky0, kx0, ky_step, kx_step;
    int OHF, OWF, out_sy, out_sx, out_oy, out_ox;
} mtd_geom;

/* Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   out[pix, n] = mask'( act( scale * sum_{tap,c} in[gather(pix,tap), c] * W(n,c,tap) + bias[n]
 *                              + add1[pix,n] + add2[pix,n] ) )
 * Replaces F.conv2d / F.conv_transpose2d / F.linear call sites: arch/Ours/networks.py:18-19,32
 * (block convs), :97-162 (generator encoder/decoder), :385-472 (discriminator), and their autograd
 * data-gradients.  Requires C % 32 == 0, N % 32 == 0 (other shapes: mtd_conv_direct) and a weight view that
 * is contiguous along c (w_sc == 1, 16-byte aligned rows): 1x1 convs / Linear are that natively, other views
 * are re-laid out once per optimizer step by mtd_pack_weights into [tap][n][c]. */
typedef struct {
    mtd_geom g;
    const float* in;  int in_ld;  int C;
    const float* w;   long long w_sn, w_sc, w_st; /* W(n,c,kidx) = w[n*w_sn + c*w_sc + kidx*w_st] */
    int N;
    float* out;       int out_ld;
    const float* scale;                          /* device scalar (1/sigma) or NULL        */
    const float* bias;                           /* [N] or NULL                            */
    const float* add1; int add1_ld;
    const float* add2; int add2_ld;
    int act;
    const float* mask; int mask_ld; float mask_slope;  /* v *= (mask>0 ? 1 : mask_slope)  */
    float* ws; size_t ws_bytes;                  /* split-K slabs (see mtd_conv_igemm_ws_bytes) */
    const float* scale2; int scale_split;        /* optional second scale: launch-grid pixels >= scale_split use *scale2.  Two
                                                    discriminator passes with their own spectral-norm sigma share one launch
                                                    (batch halves); 0 / NULL = one scale for all pixels */
    float* out2; int out2_ld;                    /* optional second output: the value BEFORE the mask factor (out gets it after).
                                                    A data-gradient launch then also hands the next layer's activation-masked
                                                    cotangent over.  Halo-tile kernel only (C == 32, 3x3, 64-pixel rows,
                                                    >= 32768 pixels); MTD_EINVAL on every other path */
    unsigned* tile_ctr; int tile_ctr_len;        /* LAB BUILDS ONLY since round 5 (ignored by the shipped library: the variant lost twice).
                                                    Optional arrival counters of split-K launches: tile_ctr_len zero-initialised
                                                    words that only these launches touch (they leave them zero).  With one word
                                                    per output tile the last slice to arrive at a tile sums the slabs, in slice
                                                    order, and runs the epilogue in the same kernel; NULL / too few = a second
                                                    launch (splitk_epilogue_kernel) does.  Same bits either way.  Launches that
                                                    may run concurrently (different streams) need different counters */
} mtd_conv_args;

size_t mtd_conv_igemm_ws_bytes(const mtd_conv_args* a);
int mtd_conv_igemm(const mtd_conv_args* a, void* stream);
/* ---- Winograd F(2x2, 3x3) form of the same contract (csrc/conv_winograd.hip) for the 3x3 stride-1 "same" layers
 * (arch/Ours/networks.py:181-221 conv{l}1/2, :230-301 *_dconv{l}1/2 and their data gradients): 16 instead of 36
 * multiplications per 2x2 output tile, fp32 MFMA, same epilogue, same split-K workspace contract.
 *   mtd_winograd_weights   transforms `count` weight views (once per optimizer step) into [xi 0..15][C/8][N][8] floats;
 *                          kmap[a*3+b] = index in the kh x kw plane of the filter entry that multiplies input offset
 *                          (-1+a, -1+b) (mtd_winograd_kmap derives it from a geometry: a forward conv and the data
 *                          gradient of the same layer use different maps); table in device memory + the same on the host.
 *   mtd_conv_winograd      a->w = the transformed weights (w_sn / w_sc ignored; w_st = 6 says they are the F(2x4, 3x3) form).  Requires 3x3, stride 1, OH x OW ==
 *                          IH x IW both even, C % 16 == 0, N % 64 == 0 (or C = N = 32 in the F(2x4, 3x3) form), no out2:
 *                          mtd_conv_winograd_ok().  The generator's 32 -> 32 channel layers (arch/Ours/networks.py:95-164) with one
 *                          residual operand at most, no scales and no mask run on a persistent kernel of their own
 *                          (csrc/conv_wino_c32.h; what whole-slice inference spends its conv time in), which also implements
 *                          MTD_ACT_RELU_ADD; with N % 64 == 0 that activation is refused as before. */
/* px: patch width of the transform along x -- 6: F(2x4, 3x3) (round 4: F(2,3) down the rows, F(4,3) along them, 24 positions,
 * 3 multiplications per output pixel; dst holds 24 N C floats), 0 or 4: F(2x2, 3x3) (16 N C floats).  A conv launch says which
 * weights it was given through a->w_st (6 or not); mtd_conv_winograd_patch_w tells the caller which form the library's plan
 * wants for a layer (0: not in the Winograd domain at all).
 * Round 5: px + 16 (20, 22) = the SPLIT-BF16 form of the same transform -- every transformed weight as three bf16 pieces whose sum
 * is the fp32 value exactly, dst = [xi][C/16][plane 0..2][N][16 c] bf16 (4 px N C x 6 bytes; C % 16 == 0): the operand of
 * csrc/conv_winograd_split.h, which forms each fp32 product from six bf16 MFMA products at fp32 accuracy.  The plan asks for it
 * on every layer with N % 64 == 0 after mtd_set_option("wino_split", 1) (off by default: see the option). */
typedef struct { const float* src; float* dst; long long sn, sc, st; int N, C; int kmap[9]; int px; } mtd_wino_weight_desc;
size_t mtd_winograd_weight_floats(int N, int C);      /* 36 N C: enough for every form */
int mtd_conv_winograd_patch_w(const mtd_conv_args* a);
int mtd_conv_winograd_f4_min_w(int min_w);      /* tuning / test hook: narrowest map width that takes F(2x4, 3x3); 0 = never, < 0 = query; returns the previous value */
int mtd_winograd_kmap(const mtd_geom* g, int* kmap9);
int mtd_winograd_weights(const mtd_wino_weight_desc* table_dev, const mtd_wino_weight_desc* table_host, int count, void* stream);
int mtd_conv_winograd_ok(const mtd_conv_args* a);
size_t mtd_conv_winograd_ws_bytes(const mtd_conv_args* a);
int mtd_conv_winograd(const mtd_conv_args* a, void* stream);
/* Two or three convs of ONE shape in one launch (round 6): the mirror layers of the discriminator's pixel-level and restoration
 * decoders (networks.py:420-467: s_dconv{l}k / r_dconv{l}k) and the data gradients of one layer in backward passes that are advanced
 * together.  a[0 .. count) as for mtd_conv_winograd, each with its own operands and workspace (mtd_conv_winograd_ws_bytes of any).
 * The split of K is planned for the group's whole grid (1 / count of the slices per problem): the results equal single calls of
 * mtd_conv_winograd up to the grouping of the K sum (a few 1e-6), bit for bit where neither splits K. */
int mtd_conv_winograd_group_ok(const mtd_conv_args* a, int count);
int mtd_conv_winograd_group(const mtd_conv_args* a, int count, void* stream);

/* ---- Winograd F(3x3, 2x2) for the 4x4 / stride-2 / padding-1 layers (csrc/conv_wino_s2.h; arch/Ours/networks.py:185-215 down1..3:
 * Conv2d(k4, s2, p1) forward and its four-parity data gradient): the forward conv is a 2x2 stride-1 conv over the 4 C channels of
 * the space-to-depth image of the padded input (never formed: each phase is read at pixel stride 2), each parity class of the
 * data gradient a 2x2 stride-1 conv over the cotangent; 16 multiplications per 3x3 output tile instead of 36.
 *   mtd_winograd_s2_kmap     for a forward geometry (TH = TW = 4, in_s = 2, tap_d = 1): groups = 4 and, per phase (py, px) = group
 *                            2 py + px, the filter entries at correlation positions (jy, jx): kmap16[4 group + 2 jy + jx];
 *                            for a class geometry (TH = TW = 2, in_s = 1, tap_d = +-1): groups = 1, kmap16[0..3].
 *   mtd_winograd_s2_weights  dst = [xi 0..15][groups C / 8][N][8] floats (16 groups N C), once per optimizer step.
 *   mtd_conv_winograd_s2     a[0 .. count) as for mtd_conv_igemm_multi (count = 1: a plain launch), a[i].w = set i's transformed
 *                            weights.  Requires C % 16 == 0, N % 64 == 0, no out2, not MTD_ACT_RELU_ADD; the sets share every operand
 *                            but w, ws and the geometry's offsets: mtd_conv_winograd_s2_ok() (0: no; 1: in the domain; 2: and the
 *                            plan expects it to beat mtd_conv_igemm / _multi on this shape).  Same epilogue, and with split-K the
 *                            same workspace contract: mtd_conv_winograd_s2_ws_bytes() bytes for EACH set. */
typedef struct { const float* src; float* dst; long long sn, sc, st; int N, C, groups; int kmap[16]; } mtd_wino_s2_weight_desc;
int mtd_winograd_s2_kmap(const mtd_geom* g, int* groups, int* kmap16);
int mtd_winograd_s2_weights(const mtd_wino_s2_weight_desc* table_dev, const mtd_wino_s2_weight_desc* table_host, int count, void* stream);
int mtd_conv_winograd_s2_ok(const mtd_conv_args* a, int count);
size_t mtd_conv_winograd_s2_ws_bytes(const mtd_conv_args* a, int count);
int mtd_conv_winograd_s2(const mtd_conv_args* a, int count, void* stream);

/* Up to four launches of ONE shape (same pixels, N, C, taps) as one grid -- the four input-parity classes of a stride-2
 * data gradient (arch/Ours/networks.py down{l}: Conv2d(k4, s2, p1) backward w.r.t. its input), each a 2x2-tap stride-1
 * gather that writes every other pixel of the same output.  a[0..count) are complete argument sets; with split-K each
 * needs its own workspace of mtd_conv_igemm_multi_ws_bytes(a, count) bytes.  No out2. */
size_t mtd_conv_igemm_multi_ws_bytes(const mtd_conv_args* a, int count);
int mtd_conv_igemm_multi(const mtd_conv_args* a, int count, void* stream);

/* dst[(t*N + n)*C + c] = src[n*sn + c*sc + t]  for `count` weight tensors in one launch (table in device
 * memory + the same table on the host for sizing).  N and C multiples of 32, T <= 16. */
typedef struct { const float* src; float* dst; int N, C, T; long long sn, sc; } mtd_pack_desc;
int mtd_pack_weights(const mtd_pack_desc* table_dev, const mtd_pack_desc* table_host, int count, void* stream);

/* Same contract on the vector ALU for degenerate channel counts (C==1, N==1, N or C not a multiple
 * of 32): generator encoder.0 / decoder.0 (networks.py:97,162), discriminator conv11, *_dconv61/62,
 * enc_out/dec_out/rec_out (networks.py:385,441-442,466-472). */
int mtd_conv_direct(const mtd_conv_args* a, void* stream);
int mtd_conv_relu_add_ok(const mtd_conv_args* a);      /* nonzero: mtd_conv_igemm takes these arguments with act = MTD_ACT_RELU_ADD */

/* Weight gradient:  dW(n,c,kidx) (+)= sum_pix p[pix,n] * q[gather(pix,tap),c]
 * (autograd of the conv call sites above).  dw is addressed like W.
 * db (optional, [N]) (+)= sum_pix p[pix,n].  accumulate: bit 0 adds into dw, bit 1 adds into db. */
typedef struct {
    mtd_geom g;
    const float* p; int p_ld; int N;
    const float* q; int q_ld; int C;
    float* dw; long long w_sn, w_sc;
    float* db;
    int accumulate;
    float* ws; size_t ws_bytes;
    /* optional (round 6): the two halves of a paired discriminator pass in ONE launch of the small-map kernels.  half_scale != NULL:
     * the cotangent is multiplied, as it is used, by *half_scale for pixels [0, m_first) and by *half_scale2 for the rest -- dw then
     * holds G_1 / sigma_1 + G_2 / sigma_2 (each half's raw gradient scaled by its own 1 / sigma: what the spectral-norm correction
     * adds up anyway, mtd_sn_grad_layer.prescaled).  db stays the unscaled sum.  m_first must be a multiple of 32 (a chunk of the K
     * loop never straddles the halves); only the register-operand kernels take it (mtd_conv_wgrad_half_scale_ok). */
    const float* half_scale; const float* half_scale2; int m_first;
} mtd_wgrad_args;

size_t mtd_conv_wgrad_ws_bytes(const mtd_wgrad_args* a);
int mtd_conv_wgrad(const mtd_wgrad_args* a, void* stream);
int mtd_conv_wgrad_half_scale_ok(const mtd_wgrad_args* a);      /* nonzero: mtd_conv_wgrad takes these arguments with half_scale set */
/* The kernel the plan picks for these arguments: index into the weight-gradient name table of the launch profiler
 * (16 = wgrad_wino_kernel, Winograd F(2x2,3x3): 4/9 of the layer's multiplications), -1 = the vector-ALU kernels of
 * mtd_conv_direct's domain, MTD_EINVAL = invalid arguments.  Nothing is launched.  (Host-side flop accounting of bench.py.) */
int mtd_conv_wgrad_plan_cfg(const mtd_wgrad_args* a);

/* The two image ranges [0, b_first), [b_first, B) of one batch, a weight gradient each (a->dw and dw2; both bias
 * gradients into a->db, the second accumulated) from ONE launch of the slab-producing kernel: the paired discriminator
 * passes of the D step (D(y) | D(fake) as one batch of 2B, networks.py:1959-1970) need the halves' gradients apart,
 * because each half has its own spectral-norm sigma, u, v.  Same arithmetic per range as mtd_conv_wgrad on that range
 * up to the order of the slab sums.  _ok: nonzero if the layer qualifies (B == 2 b_first, N and C multiples of 32, and a plan
 * whose kernel has the pair form: all but the row-window kernels of the 16-pixel-aligned 3x3 / 1x1 stride-1 layers below
 * 64 channels), else the caller runs mtd_conv_wgrad twice.  _ws_bytes: 0 if it does not qualify. */
int mtd_conv_wgrad_pair_ok(const mtd_wgrad_args* a, int b_first);
/* which plans pair: 0 none, 1 the default rule (Winograd and stride-2 halo-window kernels: the ones a pair launch is
 * faster for), 2 Winograd only, 3 every kernel that has the pair form (tests), -1 the MTD_WGRAD_PAIR environment
 * variable's choice (default 1).  Returns the previous mode. */
int mtd_conv_wgrad_pair_mode(int mode);
size_t mtd_conv_wgrad_pair_ws_bytes(const mtd_wgrad_args* a, int b_first);
int mtd_conv_wgrad_pair(const mtd_wgrad_args* a, float* dw2, int b_first, void* stream);
/* The same with the gradients taken from a->p + p_add, added as the operands are loaded (the decoders' weight gradients of
 * the D step are sums over two task passes' cotangents -- discriminator_path.cot -- and a weight gradient is linear in its
 * cotangent: no pass of its own for the sum).  p_add: shape, pixel stride and alignment of a->p.  Only where
 * mtd_conv_wgrad_pair_ok returns 2 (1: pair form without the second cotangent, 0: no pair form). */
int mtd_conv_wgrad_pair_sum(const mtd_wgrad_args* a, const float* p_add, float* dw2, int b_first, void* stream);

/* Deferred form for a backward pass with many small layers (the generator: 41 conv layers of 32 channels, each with
 * 256 partial-sum slabs).  mtd_conv_wgrad_slabs runs only the slab-producing kernel into a->ws, which must stay
 * untouched until the reduce; *nslab == 0 means the kernel wrote dw / db itself and nothing is left to do.
 * mtd_conv_wgrad_reduce_multi then sums the slabs of `count` layers and scatters into their dw / db in ONE launch
 * (same fixed two-level order as mtd_conv_wgrad's own reduce).  N and C must be multiples of 32. */
typedef struct {
    mtd_wgrad_args a;          /* as passed to mtd_conv_wgrad_slabs (a.ws = the slabs) */
    int T;                     /* taps */
    int nslab;
    long long slab_stride;     /* floats */
    int first_block;           /* filled by the caller: prefix sum of blocks (mtd_conv_wgrad_reduce_blocks) */
    int pad_;
} mtd_wgrad_reduce_desc;
int mtd_conv_wgrad_slabs(const mtd_wgrad_args* a, int* nslab, long long* slab_stride, void* stream);
/* mtd_conv_wgrad_slabs plus mtd_rfft_rows(x, x_ld, R, B, col_weight) -- in one launch when the layer runs on the row-window
 * kernel (3x3, stride 1, one 32 x 32 tile: the generator's blocks, whose weight gradient and spectral backward chain start from
 * the same cotangent), otherwise as two launches. */
int mtd_conv_wgrad_slabs_rfft(const mtd_wgrad_args* a, int* nslab, long long* slab_stride, const float* x, int x_ld, float* R,
                              int B, int col_weight, void* stream);
int mtd_conv_wgrad_reduce_blocks(const mtd_wgrad_reduce_desc* d);
int mtd_conv_wgrad_reduce_multi(const mtd_wgrad_reduce_desc* table_dev, const mtd_wgrad_reduce_desc* table_host, int count, void* stream);

/* ---- Res-FFT-Conv block spectral path (arch/Ours/networks.py:21-30), H = W = 64, C = 32 ------
 * Spectra are stored as [B][kw 0..32][h or kh 0..63][2 (re,im)][32 channels].                  */

/* rows: real FFT along W of x (NHWC, ld), ortho scale 1/8.  col_weight: 0 = none (rfft2 forward),
 * 1 = w(kw) in {1,2,...,2,1} (irfft2 backward). */
int mtd_rfft_rows(const float* x, int x_ld, float* R, int B, int col_weight, void* stream);

/* Tail of a Res-FFT-Conv block in ONE launch (arch/Ours/networks.py:32-36: `x + relu(img_conv(x)) + irfft2(...)`):
 *     out2 = act(conv3x3(in) + bias)        the spatial branch (kept: the backward pass uses it as its ReLU mask; may be NULL)
 *     out  = in + out2 + irfft_rows(T)      T = output of mtd_spec_mix_fwd / _fwd4
 * i.e. mtd_conv_igemm(out = out2) + mtd_irfft_rows(T, out, add1 = in, add2 = out2) without the second launch and its
 * re-reads: the halo-tile conv kernel's tiles are complete image rows, so the inverse row transform runs between its MFMA
 * loop and its epilogue and `in` is taken from the halo tile.  C = N = 32, 3x3 stride 1 "same", 64 x 64 maps, >= 32768
 * pixels, no add / mask / scale2 operands: mtd_resfft_block_tail_ok() says whether a->... qualifies (else MTD_EINVAL). */
int mtd_resfft_block_tail_ok(const mtd_conv_args* a);
int mtd_resfft_block_tail(const mtd_conv_args* a, const float* T, void* stream);

/* columns + channel mix + columns back, one kernel:
 *   S = FFT_H(R)/8 ; Z = W2 . [Re S; Im S] + b2 ; T = IFFT_H(relu(Z))/8
 * w2t is W2 transposed ([k][o], 64x64).  S_save / Z_save ([B][33][64][64]) may be NULL (inference). */
int mtd_spec_mix_fwd(const float* R, const float* w2t, const float* b2, float* T,
                     float* S_save, float* Z_save, int B, void* stream);

/* backward of the above given gR = weighted row-FFT of the upstream gradient:
 *   gZ = FFT_H(gR)/8 * (Z>0) ; gS = W2^T gZ ; gT = IFFT_H(gS)/8 (columns 1..31 halved: rfft2 backward)
 *   dW2 partials (slab per workgroup) and db2 partials go to ws; mtd_spec_mix_wgrad_reduce sums them. */
size_t mtd_spec_mix_bwd_ws_bytes(int B);
int mtd_spec_mix_bwd(const float* gR, const float* w2, const float* S_save, const float* Z_save,
                     float* gT, float* ws, int B, void* stream);
/* Four-wave forms of the two calls above (csrc/resfft4.hip): same mathematics, layouts and slab format; the saved
 * pre-activation is a SIGN MASK of mtd_spec_mix_zmask_bytes(B) bytes (128 64-bit words per patch and kw pair) instead of
 * a float tensor.  S_save and zmask may be NULL in the forward call (no tape). */
size_t mtd_spec_mix_zmask_bytes(int B);
int mtd_spec_mix_fwd4(const float* R, const float* w2t, const float* b2, float* T, float* S_save, void* zmask, int B, void* stream);
int mtd_spec_mix_bwd4(const float* gR, const float* w2, const float* S_save, const void* zmask, float* gT, float* ws, int B,
                      void* stream);
int mtd_spec_mix_wgrad_reduce(const float* ws, int B, float* dw2, float* db2, int accumulate, void* stream);
/* the same reduce for `count` blocks' slab sets in one launch (dw2 and ws 16-byte aligned) */
typedef struct { const float* ws; float* dw2; float* db2; int nslab; int accumulate; } mtd_mix_reduce_desc;
int mtd_spec_mix_wgrad_reduce_multi(const mtd_mix_reduce_desc* table_dev, const mtd_mix_reduce_desc* table_host, int count, void* stream);

/* rows back: c2r along W (uses only Re of columns 0 and 32), ortho 1/8, fused epilogue
 *   out = (y + add1 + add2) * (mask > 0 ? 1 : 0)     (add1/add2/mask optional, NHWC with own ld) */
int mtd_irfft_rows(const float* T, float* out, int out_ld, const float* add1, int add1_ld,
                   const float* add2, int add2_ld, const float* mask, int mask_ld, int B, void* stream);

/* The same three steps for square maps of side S = 128, 256 or 512 (whole-slice inference, reference engine.py:89,129:
 * the generator runs on 512 x 512 images and rfft2 becomes a 512-point transform).  Forward only; spectra are
 * [B][kw 0..S/2][h 0..S-1][Re 32 | Im 32], ortho scaling 1/sqrt(S) per dimension.  mtd_spec_mix_any writes zeros into the
 * imaginary halves of columns 0 and S/2 of T: mtd_irfft_rows_any does not read them (torch's c2r ignores them too), and the two
 * columns' real parts come out of one packed transform. */
int mtd_rfft_rows_any(const float* x, int x_ld, float* R, int B, int S, void* stream);
int mtd_spec_mix_any(const float* R, const float* w2t, const float* b2, float* T, int B, int S, void* stream);
int mtd_irfft_rows_any(const float* T, float* out, int out_ld, const float* add1, int add1_ld, const float* add2,
                       int add2_ld, int B, int S, void* stream);

/* 64x64 transpose of the 1x1 spectral conv weight (W2[o][k] -> W2T[k][o]). */
int mtd_transpose64(const float* src, float* dst, void* stream);
/* n transposes in one launch; ptrs_dev (device): [src_0, dst_0, src_1, dst_1, ...] (the 21 blocks' mix weights, once per step) */
int mtd_transpose64_multi(const float* const* ptrs_dev, int n, void* stream);

/* ---- element-wise helpers -------------------------------------------------------------------- */
/* out[p,c] = g[p,c] * (y[p,c] > 0 ? 1 : slope)  over npix x C, each tensor with its own ld. */
int mtd_act_grad(const float* g, int g_ld, const float* y, int y_ld, float* out, int out_ld,
                 long long npix, int C, float slope, void* stream);
/* out[i] = a[i] * b[i]  (Dropout mask at networks.py:417 and its gradient), n contiguous floats */
int mtd_mul(const float* a, const float* b, float* out, long long n, void* stream);
int mtd_add(const float* a, const float* b, float* out, long long n, void* stream);      /* out = a + b (sum of two tasks' cotangents before a shared weight gradient) */
/* Round 4: the iteration's remaining bookkeeping as library launches, so that the whole step is a list of C-ABI calls
 * (kernels.LaunchList records and replays it).  They replace torch ops of the host mirror, not reference call sites of their
 * own: nn.Dropout's multiplier (networks.py:313 c_drop; the uniform draws stay torch's generator), the upstream scalar of
 * g_loss.backward() (engine.py:52), the sums behind d_loss's stack (networks.py:1992) and the 17 logged values (engine.py:64-73),
 * and the zero fill of the gradient buffers' non-overwritten entries. */
int mtd_dropout_mask(const float* r, float p, float keep_scale, float* out, long long n, void* stream);   /* out = r >= p ? keep_scale : 0 */
int mtd_scale_by(const float* a, const float* s, float* out, long long n, void* stream);                  /* out = a * s[0], s in device memory */
typedef struct { const float* a; const float* b; int na, nb; } mtd_sum_desc;                              /* sum(a[0..na)) + sum(b[0..nb)) */
int mtd_scalar_sums(const mtd_sum_desc* table_dev, int count, float* out, void* stream);
typedef struct { float* p; long long n; } mtd_zero_desc;
int mtd_zero_multi(const mtd_zero_desc* table_dev, const mtd_zero_desc* table_host, int count, void* stream);
/* sums[t] = sum of the 32-bit patterns of tensor t modulo 2^64 (an order-independent integer checksum).  Data-parallel
 * replicas must hold bit-identical parameters after every update (train.py:93-98 relies on nn.DataParallel for that; here the
 * ranks compare these sums after the first replayed iterations: parallel.DataParallelSync.replicas_agree). */
int mtd_checksum_multi(const mtd_zero_desc* table_dev, const mtd_zero_desc* table_host, int count, unsigned long long* sums, void* stream);
/* dst[0..bytes) = src_pinned[0..bytes): src is page-locked host memory mapped into the device address space
 * (hipHostMalloc / torch pin_memory), read by a kernel on `stream`; bytes % 16 == 0, both 16-byte aligned.
 * Used for the descriptor tables (mtd_*_layer / mtd_loss_term / mtd_adamw_tensor arrays). */
int mtd_upload(const void* src_pinned, void* dst, size_t bytes, void* stream);
/* out[p, 0..C) = a[p, 0..C)  (strided copy: concat / slice), optional accumulate */
int mtd_copy_channels(const float* a, int a_ld, float* out, int out_ld, long long npix, int C,
                      int accumulate, void* stream);
/* bilinear x2, align_corners=False (nn.Upsample at networks.py:230-260) and its adjoint */
int mtd_upsample2x_fwd(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, void* stream);
int mtd_upsample2x_bwd(const float* gout, int gout_ld, float* gin, int gin_ld, int B, int H, int W, int C, void* stream);
/* same with four channels per thread and the LeakyReLU-gradient mask of the consumer fused in: gin = U^T(gout) * (y > 0 ? 1 : slope)
 * (y == NULL: no mask).  Needs C and the strides to be multiples of 4 and 16-byte aligned bases (MTD_EALIGN otherwise). */
int mtd_upsample2x_bwd_masked(const float* gout, int gout_ld, float* gin, int gin_ld, const float* y, int y_ld, float slope,
                              int B, int H, int W, int C, void* stream);
/* PixelShuffle(2) (networks.py:166-175): in [B,H,W,4C] -> out [B,2H,2W,C] and adjoint */
int mtd_pixel_shuffle2_fwd(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, void* stream);
int mtd_pixel_shuffle2_bwd(const float* gout, int gout_ld, float* gin, int gin_ld, int B, int H, int W, int C, void* stream);

/* ---- spectral norm (torch.nn.utils.spectral_norm at networks.py:181-300), batched over layers -- */
typedef struct {
    const float* w;     /* weight_orig viewed as [rows][cols]                                   */
    float* u;           /* [rows]  updated in place when train                                   */
    float* v;           /* [cols]  updated in place when train                                   */
    float* sigma;       /* out: sigma, inv_sigma = sigma[1]                                      */
    float* u_save;      /* out (optional): copies of the u, v used by this forward (for backward) */
    float* v_save;
    int rows, cols;
} mtd_sn_layer;
size_t mtd_sn_ws_bytes(const mtd_sn_layer* layers_host, int n_layers);
int mtd_sn_power_iter(const mtd_sn_layer* layers_dev, const mtd_sn_layer* layers_host, int n_layers,
                      int train, float* ws, void* stream);
/* nit train-mode power iterations on the same weights back to back (the discriminator step's four passes, networks.py:1957-1992 calls
 * D four times before any weight changes): the tables hold nit * n_layers entries, iteration-major; entry [i * n_layers + l] says where
 * iteration i leaves layer l's sigma / u_save / v_save (w, u, v, rows, cols as in entry l).  nit + 1 passes over the weights instead
 * of 2 nit: W v of one iteration and W^T (W v) of the next share a pass.  Same workspace as mtd_sn_power_iter; results differ from nit
 * calls of it by rounding (~1e-7). */
int mtd_sn_power_iter_multi(const mtd_sn_layer* layers_dev, const mtd_sn_layer* layers_host, int n_layers, int nit,
                            float* ws, void* stream);
/* g_orig (+)= G/sigma - <G,W>/sigma^2 * u v^T       */
typedef struct {
    const float* G; const float* w; const float* u; const float* v; const float* sigma;
    float* g_out; int rows, cols; int accumulate;
    /* optional second pass over the same weight (a batch-paired discriminator pass has one sigma, u, v per half):
     * g_out (+)= corr(G, u, v, sigma) and then += corr(G2, u2, v2, sigma2), in that order; G2 == NULL: single pass */
    const float* G2; const float* u2; const float* v2; const float* sigma2;
    /* prescaled != 0 (round 6): G already holds G_1 / sigma_1 + G_2 / sigma_2 (mtd_wgrad_args.half_scale) and G2 is NULL:
     * g_out (+)= G - <G_1,W>/sigma_1^2 u v^T - <G_2,W>/sigma_2^2 u2 v2^T; the two dot products must come from the activation-side
     * form below (the raw gradients no longer exist apart); u2 / v2 / sigma2 non-NULL says that there are two passes. */
    int prescaled;
    /* optional (round 6): <G, W> WITHOUT reading G or W.  With y = conv(x, W)/sigma + b and gy the cotangent of y,
     *   <G, W> = sum_pix gy . (W * x) = sigma * sum_{pix,n} gy[pix,n] (y[pix,n] - b[n]),
     * and y comes back from the saved activation a = LeakyReLU(y): y = a > 0 ? a : a * act_inv_slope.  act_gy != NULL selects this
     * form for the layer: a reduction over two [M][N] activation-sized tensors -- on the 4x4 ... 1x1 maps a few hundred KB where the
     * weight-side dot reads G, G2 and W (3 x 9.4 MB for a 512 -> 512 3x3 layer).  The caller picks it where M < cols.
     * act_gy2: a second cotangent added on load (NULL: none).  Pixels [0, act_M_first) belong to the first pass (sigma), the rest to
     * the second (sigma2); act_M_first == act_M for a single pass.  Needs rows % 4 == 0, 16-byte aligned bases and strides. */
    const float* act_gy; const float* act_gy2; const float* act_a; const float* act_bias;
    int act_gy_ld, act_gy2_ld, act_a_ld, act_M, act_M_first; float act_inv_slope;
} mtd_sn_grad_layer;
size_t mtd_sn_grad_ws_bytes(const mtd_sn_grad_layer* layers_host, int n_layers);
int mtd_sn_grad(const mtd_sn_grad_layer* layers_dev, const mtd_sn_grad_layer* layers_host, int n_layers,
                float* ws, void* stream);

/* ---- PCGrad (module/weight_methods.py:449-464) ---------------------------------------------- */
/* gram[a*T+b] = <g_a, g_b> over n elements, T <= 4 task vectors g0..g3 (flat, contiguous; unused = NULL) */
size_t mtd_pcgrad_ws_bytes(long long n, int T);
int mtd_pcgrad_gram(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n,
                    double* gram, void* ws, void* stream);
/* coefficients from the Gram matrix on the device (orders: T*T ints, shuffle order per i), then
 * merged = sum_k w_k g_k.  No host synchronisation.  coeff_out: T floats (device). */
int mtd_pcgrad_combine(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n,
                       const double* gram, const int* orders, float* merged, float* coeff_out, void* stream);

/* The two halves of mtd_pcgrad_combine on their own, for module/pcgrad.py::PCGrad (the optimizer wrapper,
 * pcgrad.py:50-69): the projections run on the flat gradient of ALL optimizer parameters, then parameters every
 * objective reaches get the mean of the projected gradients and the others their sum -- one coefficient replay, one
 * axpy (merged = scale * sum_k coeff[k] g_k over n elements) per run of parameters with the same reduction. */
int mtd_pcgrad_coeff(const double* gram, const int* orders, int T, float* coeff_out, void* stream);
int mtd_pcgrad_axpy(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n,
                    const float* coeff, float scale, float* merged, void* stream);

/* ---- fused multi-tensor AdamW (train.py:122-126; torch.optim.AdamW semantics) ---------------- */
typedef struct { float* p; const float* g; float* m; float* v; long long n; } mtd_adamw_tensor;
int mtd_adamw_multi(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count,
                    float lr, float beta1, float beta2, float eps, float wd, int step, void* stream);
/* same, with (1 - lr*wd, lr/bias_correction1, 1/sqrt(bias_correction2)) read from dyn[0..2] (device memory):
 * lets a captured hipGraph be replayed with a new step count */
int mtd_adamw_multi_pre(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count, float beta1,
                        float beta2, float eps, float decay, float step_size, float inv_sqrt_bc2, void* stream);
int mtd_adamw_multi_dyn(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count,
                        float beta1, float beta2, float eps, const float* dyn, void* stream);

/* ---- loss terms (losses.py:10-15,99-138; networks.py:1962-1977,1998-2002), batched by descriptor table --
 * kind 0: m*(a-t)^2 with t = b[i] or tconst, m = (mx[i]-my[i] != 0) or 1   (ls_gan / NDS_Loss / F.mse_loss)
 * kind 1: |a-b| (F.l1_loss)      kind 2: sqrt((a-b)^2 + eps^2) (CharbonnierLoss)
 * value[k] = scale * sum_i term;   grads: grad_out[i] (+)= coef * d term_i / d a_i                        */
typedef struct {
    int kind;
    const float* a; const float* b; float tconst;
    const float* mx; const float* my;
    long long n; float scale; float eps;
    float* grad_out; float coef; int accumulate;
} mtd_loss_term;
size_t mtd_loss_terms_ws_bytes(int nterms);
int mtd_loss_terms(const void* terms_dev, int nterms, float* out, void* ws, void* stream);
int mtd_loss_term_grads(const void* terms_dev, int nterms, void* stream);
/* x.clip(0,1) (networks.py:1969-1970) and its gradient mask (inclusive bounds, as torch.clamp) */
int mtd_clip01(const float* x, float* out, long long n, void* stream);
int mtd_clip01_bwd(const float* g, const float* x, float* out, long long n, void* stream);
/* EdgeLoss (losses.py:113-138) on B images of 64x64: out[0] = scale * sum sqrt(lap(a-b)^2 + eps^2);
 * grad_out (optional) (+)= coef * d(sum)/da */
size_t mtd_edge_loss_ws_bytes(int B);
int mtd_edge_loss(const float* a, const float* b, int B, float scale, float eps, float* out, float* grad_out,
                  float coef, int accumulate, void* ws, void* stream);

/* ---- pixel metrics of the evaluation loops (metrics.py:172-244; engine.py:78-183) -------------------------------
 * For a batch of B single-channel H x W images a, b (contiguous): out2[0] = sum (a' - b)^2, out2[1] = sum of the SSIM map
 * of (a', b) (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2), a' = clip(a, 0, 1) if clip_a.
 * RMSE = sqrt(out2[0] / n), PSNR = 10 log10(1 / (out2[0] / n + 1e-10)), SSIM = out2[1] / n with n = B*H*W. */
size_t mtd_image_metrics_ws_bytes(int B, int H, int W);
int mtd_image_metrics(const float* a, const float* b, int B, int H, int W, int clip_a, double* out2, void* ws, void* stream);

/* ---- training-patch front end (create_datasets/Mayo.py:117-136, the "window_patch" pipeline, on the device) ----------
 * Input: the two dose levels of one or more CT slices in Hounsfield units as the reference's get_pixels_hu produces them
 * (int16, Mayo.py:19-43), resident in HBM.  mtd_foreground_bbox = CropForegroundd(source_key = full dose, select x > 0
 * after windowing, i.e. HU > a_min): per slice [y0, y1, x0, x1), the whole slice when nothing is foreground.
 * mtd_window_patches = ScaleIntensityRanged(a_min, a_max -> 0..1, clip) + crop to the box + SpatialPadd(roi) (symmetric,
 * zeros) + one roi x roi sample per descriptor (RandSpatialCropSamplesd: origin = floor(u * (size - roi + 1))) followed by
 * RandRotate90d (rot_k quarter turns, np.rot90 on axes (H, W)), RandFlipd over both axes (flip != 0) and RandRotated
 * (angle radians about the patch centre, bilinear, border padding, align_corners = False; 0 = not applied), as ONE gather
 * per output pixel.  The random draws are the caller's (the reference's come from monai's RandomState): uy, ux in [0, 1).
 * Outputs: (n, 1, roi, roi) float32 for each dose level. */
typedef struct { int slice; float uy, ux; int rot_k; int flip; float angle; } mtd_patch_desc;
int mtd_foreground_bbox(const short* hu_full, int n_slices, int H, int W, float a_min, int* bbox, void* stream);
int mtd_window_patches(const short* hu_low, const short* hu_full, int n_slices, int H, int W, const int* bbox,
                       const mtd_patch_desc* descs, int n, float a_min, float a_max, int roi, float* out_low, float* out_full,
                       void* stream);
/* whole slices (valid / test pipelines, Mayo.py:150-157): out[i] = clip((hu[i] - a_min) / (a_max - a_min), 0, 1) */
int mtd_hu_window(const short* hu, long long n, float a_min, float a_max, float* out, void* stream);

/* ---- backward of one 32 -> 32 channel 3x3 generator layer in ONE launch (csrc/conv_c32_bwd.hip) ----------------------
 * d: the data gradient exactly as mtd_conv_igemm would take it (it must be a launch of the halo-tile kernel: C == N == 32,
 * 3x3, stride 1, 64-pixel rows, whole four-row tiles); w: the weight + bias gradient of the same layer exactly as
 * mtd_conv_wgrad_slabs would take it (reference: the backward of arch/Ours/networks.py:21-36 img_conv and :95-164
 * encoder / decoder layers, which autograd runs as two ATen kernels).  Slabs go to w->ws (mtd_conv_c32_bwd_ws_bytes),
 * to be summed by mtd_conv_wgrad_reduce_multi.  mtd_conv_c32_bwd_ok: 1 if the pair is eligible. */
int mtd_conv_c32_bwd_ok(const mtd_conv_args* d, const mtd_wgrad_args* w);
size_t mtd_conv_c32_bwd_ws_bytes(const mtd_conv_args* d, const mtd_wgrad_args* w);
int mtd_conv_c32_bwd(const mtd_conv_args* d, const mtd_wgrad_args* w, int* nslab, long long* slab_stride, void* stream);
/* The same launch closing the backward pass of a Res-FFT-Conv block (autograd of arch/Ours/networks.py:21-36):
 *     d->out = mask'( dgrad + add1 + add2 + irfft_rows(gT) ),   gT = output of mtd_spec_mix_bwd / _bwd4,
 * i.e. mtd_conv_c32_bwd followed by mtd_irfft_rows(gT, out, add1 = that result, mask) without the second launch: the
 * rfft2-backward row transform is 33 more MFMAs per 32-pixel block against an inverse-DFT matrix in LDS.  64 x 64 maps. */
int mtd_conv_c32_bwd_irfft(const mtd_conv_args* d, const mtd_wgrad_args* w, const float* gT, int* nslab, long long* slab_stride,
                           void* stream);
int mtd_conv_c32_bwd_stamps(unsigned long long* host256);   /* lab (MTD_C32F_STAMPS=1): clock stamps of the last launch's first 16 workgroups */

/* ---- launch profiler (bench.py's roofline leg) ---------------------------------------------------------------
 * When enabled, mtd_conv_igemm / mtd_conv_wgrad time their MAIN kernel (not the split-K / slab reductions that
 * follow it) with a pair of HIP events on the stream they were given (see mtd_prof_mode).  mtd_prof_collect synchronises those events
 * and returns one record per launch.  kernel: 0 = igemm_kernel, 1 = wgrad_kernel, 2 = the HBM-bound spectral kernels of the
 * whole-slice inference path (cfg 0 rfft_rows_any, 1 spec_mix_any, 2 irfft_rows_any: flops 0, bytes set); cfg = tile configuration index
 * (the template instantiation); flops = 2*M*N*C*taps (dense algorithmic count).
 * Not for use while a hipGraph is being captured. */
/* Tuning hook (tools/tune_igemm.py): force the tile configuration (0..8, -1 = automatic) and the split-K factor of
 * every following mtd_conv_igemm call in this process.  9 / 10 leave the plan automatic and pick the persistent / the
 * halo-tile kernel for the generator-shaped layers (C == 32, 3x3, M >= 32768). */
int mtd_conv_igemm_override(int cfg, int splitk);
int mtd_conv_wgrad_override(int cfg, int nsplit);      /* same for mtd_conv_wgrad: tile/tap-group config 0..6, pixel splits */

typedef struct mtd_prof_record {
    int kernel, cfg, splitk, N, C, taps;
    long long M;
    double flops;
    float ms;
    int _pad;
    double bytes;     /* algorithmic HBM bytes of the launch: every distinct operand element read once, every result written once */
} mtd_prof_record;
int mtd_prof_enable(int capacity);                       /* capacity <= 0 switches profiling off and frees events */
int mtd_prof_collect(mtd_prof_record* out, int max_records);   /* returns the number of completed records */
/* Timing mode of the profiler: 1 (default) = the events ride on the kernel's dispatch packet (duration = the dispatch's
 * own begin / end timestamps, what a kernel trace reports); 0 = hipEventRecord before / after the launch.  attach < 0
 * only queries.  Returns the mode in effect, MTD_EINVAL while records are pending. */
int mtd_prof_mode(int attach);

const char* mtd_version(void);

/* ---- run-time options.  The shipped library reads NO environment variable (round 5): the kernel-selection and tuning
 * switches of earlier rounds exist only in a build with -DMTD_LAB (mtd_lab_build() == 1; tools/ probes).  What may be changed
 * at run time goes through these two calls, by name; unknown names return MTD_EINVAL.
 *   "c32f_safe_wait"  0 (default) | 1: the fused 32-channel backward launch waits for ALL outstanding vector-memory operations
 *                     before it hands a halo buffer to the next DMA, instead of the counted wait (a checking mode:
 *                     tests/test_kernels_gpu.py::test_fused_c32_backward_counted_waits_same_bits).
 *   "wino_split"      0 (default) | 1: the 3x3 stride-1 layers with N % 64 == 0 run their Winograd products on the fp32 MFMA
 *                     (wino_conv_kernel) | on the bf16 matrix pipe from exact three-way bf16 splits of both operands (six
 *                     products per fp32 product, fp32 accumulation: fp32 accuracy at 2.7x the fp32 MFMA rate;
 *                     csrc/conv_winograd_split.h).  Measured (profiles/r5_wino3_parts.txt): the launches are bound by the
 *                     vector-memory path, not the matrix pipe -- 5-20 % per launch on hot inputs, nothing in the step -- so
 *                     the fp32 kernel stays the default.  Callers that cache mtd_conv_winograd_patch_w's answers drop them
 *                     after a change (tests/test_kernels_gpu.py::test_winograd_conv_vs_torch runs both).
 * Plan-level tuning hooks with their own entry points: mtd_conv_winograd_f4_min_w, mtd_conv_wgrad_plan_cfg, mtd_prof_mode. */
int mtd_set_option(const char* name, int value);
int mtd_get_option(const char* name, int* value);
int mtd_lab_build(void);

#ifdef __cplusplus
}
#endif
#endif
