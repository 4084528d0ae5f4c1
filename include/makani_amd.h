/*
 * makani_amd.h -- C ABI of the MI355X-native SFNO spectral hot path.
 *
 * One shared library (libmakani_amd.so, built by hipcc for gfx950) exports the
 * entry points below.  Plain pointers and sizes only; every device pointer is a
 * HIP device address, every `stream` is a hipStream_t passed as void*.  All
 * kernels are launched asynchronously on `stream` (graph-capturable: no
 * allocation, no host synchronisation inside).  Return value: 0 on success,
 * non-zero on error (mk_last_error() holds the message for the calling thread).
 *
 * The reference (choutilin/makani) has no FFI: its hot path is PyTorch ops
 * behind nn.Module interfaces.  Each entry point names the reference call it
 * replaces (file:line under the reference tree); INTEGRATION.md shows the
 * ctypes binding a maintainer would add.
 *
 * Private device layouts (chosen for coalesced HBM access on CDNA4):
 *   grid field   x  [BC][K][N]      real fp32, N contiguous      (= NCHW, B*C flattened)
 *   Fourier rows xf [M][K][BC]      complex64 interleaved, BC contiguous ("MKBC")
 *   spectrum     c  [L][M][BC]      complex64 interleaved, BC contiguous ("LMBC")
 *   dhconv weight w [L][I][O]       complex64 interleaved, O contiguous
 *   Legendre tab    [M][L][KP]      fp32, KP = K rounded up to 32, zero padded
 * The public torch layout [B,C,L,M] complex64 is converted with mk_spec_pack /
 * mk_spec_unpack.
 */
#ifndef MAKANI_AMD_H
#define MAKANI_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------- */
int mk_version(void);
const char* mk_last_error(void);

/* ---- host-side precompute (float64 arithmetic, no GPU needed) ---------- */
/* grid: 0 = "equiangular" (Clenshaw-Curtis), 1 = "legendre-gauss". */

/* Colatitudes (ascending from the north pole) and quadrature weights, nlat each.
 * Replaces torch_harmonics.quadrature.{clenshaw_curtiss,legendre_gauss}_weights
 * (reference call sites: makani/utils/grids.py:19,32,77,83; sfnonet.py:536-539). */
int mk_quadrature(int grid, int nlat, double* theta, double* weights);

/* Padded row length of the Legendre table for nlat latitudes. */
int mk_legendre_kpad(int nlat);

/* Orthonormal associated Legendre table with Condon-Shortley phase,
 * out[m][l][k] for m<mmax, l<lmax, k<kpad (zero for k>=nlat and l<m), fp32
 * rounded from float64.  with_quad_weights != 0 multiplies by w_k (forward
 * transform table, torch-harmonics RealSHT.weights); 0 gives the synthesis
 * table (InverseRealSHT.pct).  Replaces torch_harmonics.legendre._precompute_legpoly
 * as invoked by sfnonet.py:536-539. */
int mk_legendre_table(int grid, int nlat, int lmax, int mmax, int with_quad_weights, float* out);

/* Twiddle table for real FFTs of length nlon: out holds 2*(nlon/2) + 2*(nlon/2+1)
 * floats: exp(-2 pi i j/(nlon/2)) for j<nlon/2, then exp(-2 pi i m/nlon) for m<=nlon/2. */
int mk_fft_twiddle_len(int nlon);
int mk_fft_twiddles(int nlon, float* out);

/* ---- longitudinal real FFT (K1 / K4) ----------------------------------- */
/* xf[m][k][bc] = s_m * sum_n x[bc][k][n] exp(-2 pi i m n / nlon), m < mmax, with
 * s_0 = scale0, s_{nlon/2} = scale_h (Nyquist), s_m = scale_m otherwise.
 * Forward SHT uses all three = 2 pi / nlon:
 * `2*pi*torch.fft.rfft(x, dim=-1, norm="forward")[..., :mmax]` of torch-harmonics
 * RealSHT.forward (called at spectral_convolution.py:131, sfnonet.py:596).
 * With (1, 2, 1) it is the adjoint of mk_irfft(1, 1, 1) (backward of K4).
 * x_dtype: 0 = fp32, 1 = bf16 input rows. */
int mk_rfft(const void* x, int x_dtype, float* xf, const float* twiddles,
            int bc, int nlat, int nlon, int mmax, float scale0, float scale_m, float scale_h,
            void* stream);

/* x[bc][k][n] = s_0 Re(xf[0][k][bc]) + sum_{0<m<mmax, m != nlon/2} 2 s_m Re(xf[m][k][bc] exp(2 pi i m n/nlon))
 *               + s_h Re(xf[nlon/2][k][bc]) (-1)^n   (only if mmax == nlon/2+1).
 * (1, 1, 1) is `torch.fft.irfft(x, n=nlon, dim=-1, norm="forward")` of
 * InverseRealSHT.forward (spectral_convolution.py:133,141; sfnonet.py:598): modes
 * >= mmax are zero, imaginary parts of the zero and Nyquist modes are ignored.
 * With (2 pi/nlon, pi/nlon, 2 pi/nlon) it is the adjoint of mk_rfft (backward of K1).
 * x_dtype: 0 = fp32 rows out, 1 = bf16 rows out (fuses the `.to(dtype)` of
 * spectral_convolution.py:134,146; production lengths 480 / 1440 only). */
int mk_irfft(const float* xf, void* x, int x_dtype, const float* twiddles,
             int bc, int nlat, int nlon, int mmax, float scale0, float scale_m, float scale_h,
             void* stream);

/* ---- Legendre contraction on fp32 MFMA (K2 / K3) ------------------------ */
/* Analysis: c[l][m][n] = sum_k tab[m_off+m][l_off.. l][k] xf[m][k][n], n < 2*bc (re/im as
 * independent real columns), only l >= m (global indices); entries with l < m are not
 * written.  torch-harmonics `einsum('...kmr,mlk->...lmr', x, weights)`.
 * m_off: global index of local mode 0 (w-sharded tables); tab points at the
 * full [mmax_glob][lmax][kpad] table. */
int mk_legendre_fwd(const float* xf, const float* tab, float* c,
                    int bc, int nlat, int lmax, int mmax_loc, int m_off, int mmax_glob, void* stream);

/* Synthesis: xf[m][k][n] = sum_{l>=m} tab[m][l][k] c[l][m][n].
 * torch-harmonics `einsum('...lmr,mlk->...kmr', x, pct)`.  Also the backward of
 * mk_legendre_fwd (with the analysis table); mk_legendre_fwd with the synthesis
 * table is the backward of this one. */
int mk_legendre_inv(const float* c, const float* tab, float* xf,
                    int bc, int nlat, int lmax, int mmax_loc, int m_off, int mmax_glob, void* stream);

/* ---- the same contractions on the bf16 matrix cores, fp32-accurate ("bf16x3") ----------
 * Every fp32 operand is split exactly into three bf16 pieces and each product is evaluated
 * as the six leading piece products, accumulated in fp32 (error ~2^-22 relative, same
 * parity budget as above).  The table is pre-split once into the kernels' tile images:
 *   inverse = 0: for contractions over latitude  (mk_legendre_fwd_x3)
 *   inverse = 1: for contractions over degree    (mk_legendre_inv_x3)
 * mk_legendre_x3_bytes: size of the image; mk_legendre_x3_split: fp32 device table
 * [mmax][lmax][kpad] (mk_legendre_table) -> image (device). */
long long mk_legendre_x3_bytes(int nlat, int lmax, int mmax, int inverse);
int mk_legendre_x3_split(const float* tab, void* out, int nlat, int lmax, int mmax, int inverse, void* stream);
/* Same contracts as mk_legendre_fwd / mk_legendre_inv with `tab_x3` the matching image. */
int mk_legendre_fwd_x3(const float* xf, const void* tab_x3, float* c,
                       int bc, int nlat, int lmax, int mmax_loc, int m_off, int mmax_glob, void* stream);
int mk_legendre_inv_x3(const float* c, const void* tab_x3, float* xf,
                       int bc, int nlat, int lmax, int mmax_loc, int m_off, int mmax_glob, void* stream);

/* ---- latitude-major Fourier rows (distributed SHT) ---------------------------------------
 * `_ex` forms with xf_layout: 0 = xf[m][k][bc] (all calls above), 1 = xf[k][m][bc].  With latitude
 * outermost the two latitude all-to-alls of the distributed transform (makani/mpu/layers.py:38-169 pattern,
 * torch_harmonics distributed_transpose_polar) gather / split the OUTERMOST axis, so the received chunks are the
 * operand and the sent chunks lie back to back: no pack / concatenate copies on that side. */
int mk_rfft_ex(const void* x, int x_dtype, float* xf, const float* twiddles, int bc, int nlat, int nlon,
               int mmax, float scale0, float scale_m, float scale_h, int xf_layout, void* stream);
int mk_irfft_ex(const float* xf, void* x, int x_dtype, const float* twiddles, int bc, int nlat, int nlon,
                int mmax, float scale0, float scale_m, float scale_h, int xf_layout, void* stream);
int mk_legendre_fwd_x3_ex(const float* xf, const void* tab_x3, float* c, int bc, int nlat, int lmax,
                          int mmax_loc, int m_off, int mmax_glob, int xf_layout, void* stream);
int mk_legendre_inv_x3_ex(const float* c, const void* tab_x3, float* xf, int bc, int nlat, int lmax,
                          int mmax_loc, int m_off, int mmax_glob, int xf_layout, void* stream);

/* Peer-major Fourier rows for the distributed transform: bc = batch * chans rows, the channels cut into blocks of
 * chans_per_peer (a multiple of 24), xf = [chans / chans_per_peer][nlat][mmax][batch][chans_per_peer] -- exactly the send
 * (analysis) / receive (synthesis) buffer of the channel <-> latitude all-to-all, so that side of the transpose needs no
 * pack / concatenate copy either.  Production lengths only (nlon 480 / 1440, mmax <= 241). */
int mk_rfft_pm(const void* x, int x_dtype, float* xf, const float* twiddles, int bc, int nlat, int nlon, int mmax,
               float scale0, float scale_m, float scale_h, int chans, int chans_per_peer, void* stream);
int mk_irfft_pm(const float* xf, void* x, int x_dtype, const float* twiddles, int bc, int nlat, int nlon, int mmax,
                float scale0, float scale_m, float scale_h, int chans, int chans_per_peer, void* stream);

/* Inverse transform that also delivers the statistics of its output (round 3): rowsums[2 r], rowsums[2 r + 1] += sum and sum of
 * squares of output row r (= b * C + c) over this call's latitudes, on the values as stored; fp64 accumulators zeroed by the
 * caller.  It is the statistics pass of the instance norm that follows the inverse SHT in every FNO block
 * (sfnonet.py:239-253: norm0) without a second read of the field.  chans_per_peer > 0: peer-major rows as mk_irfft_pm, else
 * xf_layout as mk_irfft_ex.  Production lengths only (nlon 480 / 1440, mmax <= 241). */
int mk_irfft_sums(const float* xf, void* x, int x_dtype, const float* twiddles, int bc, int nlat, int nlon, int mmax,
                  float scale0, float scale_m, float scale_h, int xf_layout, int chans, int chans_per_peer, double* rowsums,
                  void* stream);

/* ---- spectral filter contraction (K5) ---------------------------------- */
/* y[l][m][b][o] = sum_i x[l][m][b][i] * w[l][i][o]  (complex), for global m <= l.
 * Replaces _contract_dhconv `einsum("bixy,iox->boxy")` (contractions.py:130-136,
 * dispatched by factorizations.py:167-200, called at spectral_convolution.py:137).
 * l_off / m_off: global indices of local l = 0 / m = 0 (h / w sharding). */
int mk_dhconv_fwd(const float* x, const float* w, float* y, int lloc, int mloc, int batch,
                  int cin, int cout, int l_off, int m_off, void* stream);
/* gx[l][m][b][i] = sum_o gy[l][m][b][o] * conj(w[l][i][o]) */
int mk_dhconv_dgrad(const float* gy, const float* w, float* gx, int lloc, int mloc, int batch,
                    int cin, int cout, int l_off, int m_off, void* stream);
/* gw[l][i][o] = sum_{m<=l, b} conj(x[l][m][b][i]) * gy[l][m][b][o] */
int mk_dhconv_wgrad(const float* x, const float* gy, float* gw, int lloc, int mloc, int batch,
                    int cin, int cout, int l_off, int m_off, void* stream);

/* bf16x3 variants of the three dhconv kernels (see the Legendre section): same contracts,
 * cin and cout must be even (returns an error otherwise; the fp32 kernels take any size). */
int mk_dhconv_fwd_x3(const float* x, const float* w, float* y, int lloc, int mloc, int batch,
                     int cin, int cout, int l_off, int m_off, void* stream);
int mk_dhconv_dgrad_x3(const float* gy, const float* w, float* gx, int lloc, int mloc, int batch,
                       int cin, int cout, int l_off, int m_off, void* stream);
int mk_dhconv_wgrad_x3(const float* x, const float* gy, float* gw, int lloc, int mloc, int batch,
                       int cin, int cout, int l_off, int m_off, void* stream);

/* ---- "diagonal" spectral filter: one complex weight per (l, m) ---------------------------
 * Public layout, P = L * M contiguous: x [B][I][P], w [I][O][P], y [B][O][P] complex64.
 *   y[b][o][p] = sum_i x[b][i][p] * w[i][o][p]
 * Replaces _contract_diagonal `einsum("bixy,ioxy->boxy")` (contractions.py:121-127) and its gradients
 *   gx[b][i][p] = sum_o gy[b][o][p] * conj(w[i][o][p]),   gw[i][o][p] = sum_b conj(x[b][i][p]) * gy[b][o][p].
 * Elementwise in p (1 flop per weight byte): HBM-streaming kernels, exact fp32 fma chains. */
int mk_diag_fwd(const float* x, const float* w, float* y, int batch, int cin, int cout, long long P, void* stream);
int mk_diag_dgrad(const float* gy, const float* w, float* gx, int batch, int cin, int cout, long long P, void* stream);
int mk_diag_wgrad(const float* x, const float* gy, float* gw, int batch, int cin, int cout, long long P, void* stream);

/* ---- layout conversion -------------------------------------------------- */
/* torch [BC][L][M] complex64  <->  private [L][M][BC] complex64.  unpack writes
 * exact zeros where global l < m (what the reference's zero table entries give). */
int mk_spec_pack(const float* c_std, float* c_prv, int bc, int lloc, int mloc, void* stream);
int mk_spec_unpack(const float* c_prv, float* c_std, int bc, int lloc, int mloc,
                   int l_off, int m_off, void* stream);

/* ---- fused pointwise ops of the FNO block (rows = B*C <= 65535, P = H*W multiple of 8) -------- */
/* dtype: 0 = fp32, 1 = bf16 storage; arithmetic is fp32.
 * y = gelu(x + bias[row % C]) (exact erf GELU).  Replaces the bias add of nn.Conv2d(.., 1) followed by
 * nn.GELU in MLP / EncoderDecoder (makani/models/common/layers.py:95-99,158-206). */
int mk_bias_gelu_fwd(const void* x, const float* bias, void* y, int dtype, int rows, int C, long long P,
                     void* stream);
/* gx = gy * gelu'(x + bias); gbias[c] += sum over (b, p) of gx (caller zeroes gbias; may be NULL). */
int mk_bias_gelu_bwd(const void* x, const float* bias, const void* gy, void* gx, float* gbias, int dtype,
                     int rows, int C, long long P, void* stream);
/* Instance norm over each row, y = act(((x - mean) * rstd) * weight[c] + bias[c]), biased variance,
 * act = GELU if fuse_gelu else identity.  stats[row] = (mean, rstd) is kept for the backward;
 * workspace: 2*rows doubles (zeroed inside).  Replaces nn.InstanceNorm2d(eps=1e-6, affine=True)
 * (+ act_layer0) of FourierNeuralOperatorBlock (makani/models/networks/sfnonet.py:239-253,375-380). */
int mk_instnorm_fwd(const void* x, const float* weight, const float* bias, void* y, float* stats,
                    double* workspace, int dtype, int rows, int C, long long P, float eps, int fuse_gelu,
                    void* stream);
/* gx of the above; on return workspace[row] = (sum g', sum g' * xhat) with g' = gy * act'(z), from which
 * the caller forms gbias[c] = sum_b workspace[b, c, 0], gweight[c] = sum_b workspace[b, c, 1]. */
int mk_instnorm_bwd(const void* x, const void* gy, const float* stats, const float* weight, const float* bias,
                    void* gx, double* workspace, int dtype, int rows, int C, long long P, int fuse_gelu,
                    void* stream);

/* Split-phase forms for rows sharded over ranks (DistributedInstanceNorm2d, makani/mpu/layer_norm.py:27-114):
 * phase 1 = local row sums into `workspace` ([rows][2] doubles: sum x, sum x^2 / sum g', sum g' xhat);
 * the caller all-reduces `workspace` over the ranks sharing the rows; phase 2 = apply with the reduced sums and
 * `count` = the global number of elements per row.  phase 0 = both (count = P): the single-GPU calls above. */
int mk_instnorm_fwd_ex(const void* x, const float* weight, const float* bias, void* y, float* stats,
                       double* workspace, int dtype, int rows, int C, long long P, long long count, float eps,
                       int fuse_gelu, int phase, void* stream);
int mk_instnorm_bwd_ex(const void* x, const void* gy, const float* stats, const float* weight, const float* bias,
                       void* gx, double* workspace, int dtype, int rows, int C, long long P, long long count,
                       int fuse_gelu, int phase, void* stream);
/* mk_instnorm_bwd for one sample (rows = C) that also writes the gradients of the affine parameters: gwb fp32 [2][C], row 0 =
 * weight gradient (sum g' xhat), row 1 = bias gradient (sum g'), the backward of `nn.InstanceNorm2d(affine=True)`'s parameters
 * (sfnonet.py:239-253) without a copy / cast launch behind the kernel. */
int mk_instnorm_bwd_wb(const void* x, const void* gy, const float* stats, const float* weight, const float* bias, void* gx,
                       double* workspace, float* gwb, int dtype, int C, long long P, int fuse_gelu, void* stream);

/* ---- 1x1 convolution weight gradient (bf16 MFMA) ------------------------------------------ */
/* Latitude-weighted squared error of the training harness (latitude weights as in
 * makani/utils/losses.py:149-271):  loss = scale * sum_{r,w} wrow[r % H] * (pred[r][w] - tar[r][w])^2 over rows
 * r = (b, c, h) of W points (W % 8 == 0); pred fp32 (dtype 0) or bf16 (1), tar fp32, loss one double (zeroed here).
 * Backward: gpred = 2 * scale * gloss[0] * wrow[r % H] * (pred - tar) in pred's dtype. */
int mk_wmse_fwd(const void* pred, int dtype, const float* tar, const float* wrow, double* loss, long long rows,
                int H, int W, float scale, void* stream);
int mk_wmse_bwd(const void* pred, int dtype, const float* tar, const float* wrow, const float* gloss, void* gpred,
                long long rows, int H, int W, float scale, void* stream);

/* The same convolutions on fp32 fields (no autocast), fp32-accurate on the bf16x3 engine of the spectral GEMMs (csrc/gemm_x3.hip):
 *   mode 0:  c[b] = a b[b]            a [M][K] row-major (lda a multiple of 4, rows zero-padded to K rounded up to 4), b[b] [K][N] = the NCHW field (N = H*W even)
 *   mode 1:  c[b] += a b[b]           (a skip connection folded into the GEMM: c holds the addend)
 *   mode 2:  c += sum_b a[b] b[b]^T   a[b] [M][K], b[b] [N][K], K = H*W the contraction: the weight gradient (c zeroed by the caller, fp32 atomics)
 * sa / sb / sc = batch strides in elements.  Replaces F.conv2d / its gradients behind nn.Conv2d(.., 1) in fp32 mode (layers.py:95-206). */
int mk_conv1x1_x3(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int K, long long N,
                  int batch, long long sa, long long sb, long long sc, int mode, void* stream);
/* c[b] = act(a b[b] + bias): mode 0 with the bias add (bias fp32 [M] or NULL) and, with act = 1, the exact (erf) GELU in the epilogue:
 * `nn.Conv2d(cin, cout, 1, bias=True)` followed by `nn.GELU()` (layers.py:95-99, 158-206) in one pass over the output. */
int mk_conv1x1_x3_bias_act(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int K,
                           long long N, int batch, long long sb, long long sc, const float* bias, int act, void* stream);

/* gw[o][i] += sum over (b, p) of gy[b][o][p] * x[b][i][p]; gy, x bf16 [B][C][P] (P multiple of 8), gw fp32
 * [cout][cin] accumulated with atomics (caller zeroes it).  The weight gradient of nn.Conv2d(cin, cout, 1)
 * in MLP / EncoderDecoder / skip connections (layers.py:95-128,158-183; sfnonet.py:207,463). */
int mk_conv1x1_wgrad(const void* gy, const void* x, float* gw, int batch, int cout, int cin, long long P,
                     void* stream);
/* The same with an activation applied to x while it is staged: x_act = 1 multiplies with GELU(x) (exact erf form,
 * rounded to bf16).  The weight gradient of the SECOND convolution of an MLP (layers.py:158-206) from the kept
 * pre-activation: with mk_pce_mlp the activated hidden field is never written.  cout <= 384. */
int mk_conv1x1_wgrad_act(const void* gy, const void* x, float* gw, int batch, int cout, int cin, long long P, int x_act,
                         void* stream);
/* Pixel-column engine (csrc/pce.hip): the same 1x1 convolutions as one persistent kernel per GEMM with the
 * pointwise passes of layers.py:86-216 / sfnonet.py:239-267 folded into the epilogue:
 *   acc[b][m][p] = sum_k A[m][k] * x[b][k][p] (+ bias[m]);   aux_out <- acc (bf16, optional: the pre-activation kept for
 *   the backward pass);   v = gelu ? GELU(acc) : acc;   v *= GELU'(aux_in[b][m][p]) (optional: backward of the activation);
 *   y = v (+ addend[b][m][p]).
 * A comes pre-packed (mk_pce_pack) as the MFMA fragment image of W [M][K] (forward) or of W^T (data gradient).
 * x, y, addend, aux_* are bf16 [B][C][P], P a multiple of 8; K <= 768, M <= 1536; bias is fp32 [M]
 * (or NULL); addend and aux_in are exclusive.
 * Replaces hipBLASLt's mm/addmm behind nn.Conv2d(.., 1) and the separate bias+GELU passes. */
long long mk_pce_image_bytes(int M, int K);
int mk_pce_pack(const void* w, int w_dtype /* 0 fp32, 1 bf16 */, int transpose, int M, int K, int ldw, void* img,
                void* stream);
/* All weight images of a net in ONE launch: desc_dev = n descriptors of 10 x int64 {weight pointer, dtype (0 fp32 / 1 bf16),
 * transpose, M, K, leading dimension, TH, steps per pass, core elements (the last three from mk_pce_pack_layout), first element in
 * the arena}, sorted by first element; image e occupies mk_pce_image_bytes(M, K) bytes from arena + 2 * first (the 64 zero bytes
 * included).  Replaces the per-call mk_pce_pack of every 1x1 convolution of a training step (56 launches at the SFNO config). */
int mk_pce_pack_layout(int M, int K, long long* out3);
int mk_pce_pack_batch(const void* desc_dev, int n, void* arena, long long total_elements, void* stream);
int mk_pce_gemm(const void* x, const void* wimg, void* y, const float* bias, const void* addend, const void* aux_in,
                void* aux_out, int gelu, int batch, int M, int K, long long P, void* stream);
/* The same with two more seams to the instance norms around the convolutions (sfnonet.py:262-267):
 *  - rowstats [B][M][2] (double, zeroed by the call; M <= 768) receives sum and sum of squares over the pixels of every
 *    stored y row: the statistics pass of the norm that follows an MLP, and the bias gradient of a convolution (sum over
 *    pixels of the output gradient), without another pass over the field;
 *  - addend_affine [B][M][2] (float) makes the addend enter as a * addend + b per row: the APPLY pass of the instance norm
 *    in front of a skip connection (coefficients from mk_instnorm_coeffs), folded into the skip convolution. */
int mk_pce_gemm_ex(const void* x, const void* wimg, void* y, const float* bias, const void* addend, const float* addend_affine,
                   const void* aux_in, void* aux_out, int gelu, double* rowstats, int batch, int M, int K, long long P,
                   void* stream);
/* Fused two-layer node (csrc/pce_mlp.hip): conv1x1 -> GELU -> conv1x1 of `MLP` / `EncoderDecoder` (layers.py:86-216; call
 * sites sfnonet.py:207,379,463) as ONE launch with the Hd-row hidden field kept on chip:
 *   mode 0 (forward):   mid_out = A1 x + b1  (bf16 [B][Hd][P], the pre-activation kept for backward);
 *                       y = A2 GELU(mid_out) (+ b2);   rowstats_y as in mk_pce_gemm_ex (or NULL)
 *   mode 1 (backward):  mid_out = (A1 x) * GELU'(mid_in)  (x = the output gradient, mid_in = the kept pre-activation:
 *                       mid_out is the gradient w.r.t. the pre-activation, operand of the first weight gradient);
 *                       y = A2 mid_out (the input gradient);   rowsum_mid [B][Hd] (double, zeroed by the call, or NULL) receives
 *                       the pixel sums of mid_out = the gradient of b1
 * A1 [Hd][K1] and A2 [M][Hd] come packed by mk_pce_mlp_pack (forward: W1, W2; backward: W2^T, W1^T -- `*_transposed` says
 * the array holds the transpose, i.e. a1 is [K1][Hd] / a2 is [Hd][M]).  K1 <= 384, Hd <= 768, M <= 384; x, y, mid_* bf16,
 * P a multiple of 8, biases fp32 or NULL. */
long long mk_pce_mlp_image_bytes(int M, int Hd, int K1);
int mk_pce_mlp_pack(const void* a1, int a1_transposed, int lda1, const void* a2, int a2_transposed, int lda2,
                    int w_dtype /* 0 fp32, 1 bf16 */, int M, int Hd, int K1, void* img, void* stream);
int mk_pce_mlp(const void* x, const void* wimg, void* y, void* mid_out, const void* mid_in, const float* b1, const float* b2,
               double* rowstats_y, double* rowsum_mid, int mode, int batch, int M, int Hd, int K1, long long P, void* stream);
/* Profiling aid (tools/mlp_stamps.py; build with -DMK_MLP_STAMPS, MK_MLP_DBG=1): s_memtime stamps of workgroup 0, 4 x 128. */
int mk_pce_mlp_debug_stamps(unsigned long long* out512);
/* Per-row coefficients of an instance norm from its row sums: stats[row] = (mean, rstd) (the form mk_instnorm_bwd takes),
 * affine[row] = (rstd * weight[c], bias[c] - mean * rstd * weight[c]); count = elements per row (global, when sharded). */
int mk_instnorm_coeffs(const double* sums, const float* weight, const float* bias, float* stats, float* affine, int rows,
                       int C, long long count, float eps, void* stream);
/* Profiling aid (tools/pce_stamps.py): with MK_PCE_DBG=1 in the environment the kernel records s_memtime stamps of
 * workgroup 0; this copies the 8 x 64 stamps of the last launch to the host. */
int mk_pce_debug_stamps(unsigned long long* out512);

/* One Adam step (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias-corrected moments, no
 * amsgrad) over n contiguous fp32 elements in one streaming pass; `step` is the 1-based step count.  The optimizer
 * step of the training harness (makani/utils/trainer.py:762-763); complex parameters are stepped as 2 n reals. */
int mk_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAKANI_AMD_H */
