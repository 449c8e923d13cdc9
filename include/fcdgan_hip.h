/* libfcdgan_hip.so -- C ABI of the MI355X (gfx950) FCD-GAN hot path.
 *
 * The reference (Cwuwhu/FCD-GAN-pytorch) has NO native/FFI layer: its only
 * boundary is the Python nn.Module surface of Module.py / Loss.py / ssim.py,
 * and every arithmetic op below is reached through torch.nn (ATen).  Each entry
 * point here therefore replaces one ATen operator *as invoked from* the cited
 * reference line; the Python host mirror (fcd_gan_pytorch_amd/) binds them with
 * ctypes.
 *
 * Conventions
 *  - plain C symbols, POD arguments, no torch types; all tensors fp32 NCHW
 *    contiguous device pointers owned by the CALLER (incl. workspaces);
 *  - return 0 on success, negative fcd_status otherwise (never throws);
 *    fcd_last_error_string() describes the last failure of the calling thread;
 *  - every launch is asynchronous on the hipStream_t passed last (as void*),
 *    no internal synchronisation, no persistent device allocation.
 */
#ifndef FCDGAN_HIP_H
#define FCDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fcd_status {
  FCD_OK = 0,
  FCD_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  FCD_ERR_LAUNCH = -2,      /* HIP launch error */
  FCD_ERR_WORKSPACE = -3    /* workspace too small */
} fcd_status;

enum { FCD_ACT_NONE = 0, FCD_ACT_RELU = 1, FCD_ACT_LEAKY = 2, FCD_ACT_PRELU = 3 };

int fcd_version(void);
/* sha256[:16] over the HIP sources + this header the loaded binary was built from (csrc/source_hash.py); measurements
 * committed under profiles/ are stamped with it and reported only by the binary they were taken on */
const char* fcd_build_hash(void);
const char* fcd_last_error_string(void);

/* ---- run-time switches (csrc/switches.h) -----------------------------------
 * ONE table of integer switches (which kernel / launch geometry serves a layer; A/B runs, tests, bench.py's
 * `fp32_mfma_only` pass).  It is filled once, when the library is loaded, from the environment variables FCD_<NAME>;
 * after that nothing on any call path reads the environment -- a run changes a switch through fcd_switch_set.  The
 * reference has no such knobs (its kernels are ATen's); the host mirror reads its own A/B switches from the same table.
 * Names are accepted with or without the FCD_ prefix.  fcd_switch_get: the value (never negative), FCD_ERR_INVALID
 * for an unknown name.  fcd_switch_set: returns the previous value; value < 0 restores the default. */
int fcd_switch_count(void);
const char* fcd_switch_name(int i);
const char* fcd_switch_help(int i);
int fcd_switch_default(int i);
int fcd_switch_get(const char* name);
int fcd_switch_set(const char* name, int value);

/* ---- convolution ---------------------------------------------------------
 * Replaces nn.Conv2d as used at Module.py:25,29 (3x3 p1), :146,158 (9x9 p4),
 * :85 (1x1), :196-205 (3x3 s2 p1), :212,214 (1x1 on 1x1 maps), the VGG16
 * 3x3 stack of Loss.py:25, and nn.ConvTranspose2d(k2,s2) at Module.py:63.
 *
 * desc: input (N,C,H,W), filter (K,C,R,S), stride (1|2), pad, output (N,K,P,Q)
 *   with P = (H+2*pad-R)/stride+1.  Supported filters: 3x3 s1/s2, 9x9 s1,
 *   1x1 s1, 2x2 s2 (the conv whose data-gradient is ConvTranspose2d k2 s2).
 */
typedef struct fcd_conv_desc {
  int32_t N, C, H, W;
  int32_t K, R, S;
  int32_t stride, pad;
  int32_t P, Q;
} fcd_conv_desc;

/* Packed ("GEMM-A") weight layouts consumed by the MFMA kernels.
 * mode 0 (forward):   wp[(c*R*S + r*S + s) * Kpad + k] = w[k][c][r][s]
 * mode 1 (data-grad): wp[(k*R*S + r*S + s) * Cpad2 + c] = w[k][c][R-1-r][S-1-s]
 * zero-filled up to the padded extents reported by fcd_conv_packed_elems(). */
int64_t fcd_conv_packed_elems(int K, int C, int R, int S, int mode);
int fcd_conv_pack_weights(const float* w, float* wp, int K, int C, int R, int S, int mode, void* stream);

/* y = conv(x, w) + bias (bias may be NULL), then ReLU when fuse_relu != 0 (the
 * conv+ReLU pairs of the VGG16 stack, Loss.py:25).  wp: mode-0 packed weights. */
int fcd_conv2d_fwd(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                   float* y, int fuse_relu, void* stream);
/* Forward with the general fused epilogue (inference paths):
 *   y = act(conv(x, w) + bias) + residual
 * act: FCD_ACT_*; slope from device memory (slope_ptr, the PReLU weight) or slope_imm;
 * residual: optional tensor of y's shape (ResidualBlock / block7 skip adds, Module.py:171,190). */
int fcd_conv2d_fwd_ex(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                      float* y, int act, const float* slope_ptr, float slope_imm,
                      const float* residual, void* stream);
/* Thin-channel 3x3 layers (<= 4 input channels, > 32 output channels, W % 4 == 0, K % 8 == 0; VGG conv1_1 on
 * single bands, Loss.py:52-58): fused-ReLU forward that also writes the ReLU mask as 4 bits per 1 x 4 pixel
 * strip (fcd_conv2d_relu_bits_bytes() bytes; 0 = the layer has no such path), and the data gradient that
 * consumes the bits instead of the fp32 activation. */
size_t fcd_conv2d_relu_bits_bytes(const fcd_conv_desc* d);
int fcd_conv2d_fwd_relu_bits(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                             float* y, unsigned char* bits, void* stream);
int fcd_conv2d_bwd_data_bits(const fcd_conv_desc* d, const float* dy, const unsigned char* bits,
                             const float* wp_bwd, float* dx, void* stream);
/* 1x1 convolution to ONE output channel (+ sigmoid): the Segmentor's change-density head, reference Module.py:82-90
 * (`OutConv`: nn.Conv2d(in, 1, kernel_size=1) followed by nn.Sigmoid), as streaming kernels -- one pass over the
 * (N, C, HW) activation per direction instead of a 1-row MFMA GEMM.  fcd_conv1x1_head_plan(): 1 when the shape is
 * supported (K == 1, HW % 4 == 0, HW >= 1024, C >= 8; FCD_CONV_HEAD=0 turns the path off).
 * forward:  y[n, p] = act(bias + sum_c w[c] x[n, c, p]),  act = sigmoid when `sigmoid` != 0.
 * backward: g = dy * y_sig * (1 - y_sig) with y_sig = the forward's sigmoid output (NULL: g = dy);
 *           dx[n, c, p] = w[c] g[n, p];  dw[c] = sum_{n,p} g x;  db = sum g.  dx or (dw, db) may be NULL.
 *           ws: fcd_conv1x1_head_bwd_ws_bytes(N, C) bytes (fp64 partial sums, reduced in a fixed order). */
int fcd_conv1x1_head_plan(int N, int C, int HW, int K);
int fcd_conv1x1_head_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int HW,
                         int sigmoid, void* stream);
size_t fcd_conv1x1_head_bwd_ws_bytes(int N, int C);
int fcd_conv1x1_head_bwd(const float* x, const float* w, const float* dy, const float* y_sig, float* dx, float* dw,
                         float* db, int N, int C, int HW, void* ws, size_t ws_bytes, void* stream);
/* [r5] The head behind a train-mode BatchNorm2d + ReLU (reference Module.py:25-31 -> :82-90: the decoder's last DoubleConv feeds OutConv
 * and nothing else): y = [sigmoid](b + sum_c w[c] relu(z[c] scale[c] + shift[c])) straight from the BatchNorm INPUT z, and the whole
 * backward tail (head data / weight gradient, BatchNorm reduce + apply) as two passes over z: the 128-channel activation and its
 * gradient are never tensors (10 -> 4 passes over the Segmentor's largest tensor).  scale / shift from fcd_bn_train_stats, mean /
 * invstd its saved statistics, all [groups][C]; workspace fcd_conv1x1_head_bn_bwd_ws_bytes. */
int fcd_conv1x1_head_bn_fwd(const float* z, const float* scale, const float* shift, int groups, const float* w, const float* bias,
                            float* y, int N, int C, int HW, int sigmoid, void* stream);
size_t fcd_conv1x1_head_bn_bwd_ws_bytes(int N, int C, int groups);
int fcd_conv1x1_head_bn_bwd(const float* z, const float* w, const float* dy, const float* y_sig, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int groups, float* dz, float* dw, float* db,
                            float* dgamma, float* dbeta, int N, int C, int HW, void* ws, size_t ws_bytes, void* stream);
/* ---- Winograd F(m x m, 3 x 3) path for wide 3x3 / stride-1 / pad-1 layers (m = 2 or 4) ------------
 * fcd_conv_wino_plan(): tile size the library uses for this layer and direction (mode 0 forward,
 * 1 data gradient), 0 = the layer runs on the direct kernels (then none of the *_wino calls apply).
 * Filters are transformed once per weight version with fcd_conv_wino_pack (opaque layout,
 * fcd_conv_wino_filter_elems floats); every call needs fcd_conv_wino_ws_bytes of caller workspace.
 * Replaces the same nn.Conv2d call sites as fcd_conv2d_fwd / _bwd_data (Module.py:25-31, Loss.py:25). */
int fcd_conv_wino_plan(const fcd_conv_desc* d, int mode);
/* tile-size policy: 0 = direct kernels only, 2 or 4 (default, switch WINO); returns the previous value */
int fcd_conv_wino_set(int m);
/* matrix pipe of the batched GEMM of the three-kernel path: 1 (default, switch WINO_SPLIT) = bf16 MFMA on exact
 * three-way bf16 splits of the fp32 operands, six partial products accumulated in fp32 (fp32-equivalent result);
 * 0 = v_mfma_f32_32x32x2_f32 (fcd_conv_wino_pack writes the operand form of the CURRENT setting -- fp32 U or its three
 * bf16 planes -- so filters packed before a change of this switch must be packed again); 2 = as 1 with the 256 x 256-tile kernel for every GEMM of >= 256 rows (tests: the policy
 * otherwise keeps it for launches that fill the chip).  on < 0 only queries.  Returns the previous value. */
int fcd_conv_wino_split_set(int on);
/* matrix pipe of the NCHW-direct 3x3 weight-gradient kernel (fcd_conv2d_bwd_weight*, layers that do not take the F(4x4)
 * form; replaces the weight gradient autograd derives for nn.Conv2d at Module.py:25-31,177-181,196-205): 1 (default, switch
 * WGRAD_SPLIT) = bf16 MFMA on exact three-way splits of x and dY, six products per multiply accumulated in fp32; 0 = v_mfma_f32_*
 * (stride-1 layers; the stride-2 form exists on the bf16 pipe only and then takes the channel-minor copies + fp32 kernel).
 * on < 0 only queries.  Returns the previous value. */
int fcd_conv_wgrad_split_set(int on);
size_t fcd_conv_wino_ws_bytes(const fcd_conv_desc* d, int mode);
int64_t fcd_conv_wino_filter_elems(int K, int C, int mode, int m);
int fcd_conv_wino_pack(const float* w, float* U, int K, int C, int mode, int m, void* stream);
/* (round 5) every F(4x4) pack of a net in ONE launch, e.g. right after an optimizer step (the Segmentor: 18 layers x {forward, data
 * gradient} = 36 launches otherwise).  items_dev: n x 8 int64 in DEVICE memory {w, U, K, C, mode, first block, blocks, split}: w / U as
 * fcd_conv_wino_pack takes them (U: fcd_conv_wino_filter_elems(K, C, mode, 4) floats), blocks = fcd_conv_wino_pack_blocks(K, C, mode),
 * first block = sum of the previous items' blocks, split = fcd_conv_wino_split_set(-1) != 0 && rows > 64 (the pack then holds the
 * three bf16 planes, else the fp32 U); total_blocks = sum of blocks; total_elems = sum of rows x Kc (profiling only).  Bit-identical to
 * the single calls. */
int fcd_conv_wino_pack_blocks(int K, int C, int mode);
int fcd_conv_wino_pack_multi(const long long* items_dev, int n, int total_blocks, double total_elems, void* stream);
/* y = [relu](conv + bias); with pool_y != NULL instead pool_y / code = maxpool2(relu(conv + bias)) */
int fcd_conv2d_fwd_wino(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                        int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                        void* stream);
/* ---- virtual channel concatenation (the U-Net decoder: Module.py:78 `torch.cat([x2, x1], dim=1)` feeding DoubleConv, and the
 * Siamese skip pairs Module.py:116-132) -- the convolution reads / its data gradient writes up to three (N, chans[i], H, W)
 * tensors as if they were ONE (N, C, H, W) tensor; no concatenated copy exists.  chans[i] % 32 == 0, sum == d->C.
 * fcd_conv_wino_cat_ok(): 1 when forward, data gradient and weight gradient of the layer all take tensor lists (wide 3x3
 * layers on the F(4x4) path); otherwise concatenate and use the plain calls.  Workspaces as for the plain calls
 * (fcd_conv_wino_ws_bytes / fcd_conv2d_bwd_weight_ws_bytes). */
int fcd_conv_wino_cat_ok(const fcd_conv_desc* d);
int fcd_conv2d_fwd_wino_cat(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc, const float* U,
                            const float* bias, float* y, int fuse_relu, void* ws, size_t ws_bytes, void* stream);
int fcd_conv2d_bwd_data_wino_cat(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* U,
                                 float* const* dsrc, const int* chans, int nsrc, void* ws, size_t ws_bytes, void* stream);
int fcd_conv2d_bwd_weight_bias_cat(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                   const float* dy, const float* relu_out, float* dw, float* db, void* ws, size_t ws_bytes,
                                   void* stream);
/* ---- the forward pass's transformed input kept for the weight gradient (round 3).  The weight gradient of a wide 3x3 layer
 * (Winograd form) needs V = B^T x B of the layer input -- exactly what the forward pass just computed.  With
 * fcd_conv_wino_keepv_bytes(d) > 0 the caller may hand the forward call a buffer of that size (v_keep); it receives V
 * [36][C/32][tiles][32], and fcd_conv2d_bwd_weight_bias_v() computes dw / db from it and dy alone (x is not read again:
 * the weight gradient's own input transform, 1.6 ms per Demo_RSSS step, disappears; the GEMM reads V transposed).  The
 * autograd tape of reference Module.py:25-31 keeps x for the weight gradient; here it keeps V instead.  Workspace of the
 * _v call: fcd_conv2d_bwd_weight_ws_bytes(d). */
/* Optional outputs of the F(4x4) forward calls (fcd_conv2d_fwd_wino_x / _cat_x; NULL pointers = not wanted):
 *   v_keep   the transformed input V for the weight gradient (above; fcd_conv_wino_keepv_bytes)
 *   bn_part  per-workgroup {sum y, sum y^2} of every output channel and BatchNorm sample group -- the statistics pass of the
 *            BatchNorm2d that follows the convolution (reference Module.py:25-31, 177-181) done by the output transform while
 *            y is in registers; hand it to fcd_bn_act_fwd_parts.  Layout [(g K + k) split + s][3] doubles with
 *            split = fcd_conv_wino_bn_split(d, groups) (0: not available for this layer / grouping), no fused ReLU.
 *   in_scale / in_shift / in_groups   [r5] x is the INPUT of a train-mode BatchNorm2d + ReLU whose only consumer is this convolution
 *            (the middle of reference Module.py:25-31 DoubleConv): the input transform computes relu(x * in_scale[(n / (N / in_groups)) C + c]
 *            + in_shift[...]) while loading -- the arithmetic of fcd_bn_act_fwd's apply pass, bit for bit -- and the activation is never
 *            written.  scale / shift come from fcd_bn_train_stats; allowed when fcd_conv_wino_in_affine_ok(d). */
typedef struct fcd_wino_fwd_extras {
  float* v_keep;
  double* bn_part;
  int32_t bn_groups;
  int32_t in_groups;
  const float* in_scale;
  const float* in_shift;
} fcd_wino_fwd_extras;
int fcd_conv_wino_in_affine_ok(const fcd_conv_desc* d);
int fcd_conv_wino_bn_split(const fcd_conv_desc* d, int groups);
size_t fcd_conv_wino_bn_part_bytes(const fcd_conv_desc* d, int groups);
int fcd_conv2d_fwd_wino_x(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y, int fuse_relu,
                          float* pool_y, unsigned char* code, void* ws, size_t ws_bytes, const fcd_wino_fwd_extras* ex,
                          void* stream);
int fcd_conv2d_fwd_wino_cat_x(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc, const float* U,
                              const float* bias, float* y, int fuse_relu, void* ws, size_t ws_bytes,
                              const fcd_wino_fwd_extras* ex, void* stream);
size_t fcd_conv_wino_keepv_bytes(const fcd_conv_desc* d);
int fcd_conv2d_fwd_wino_keepv(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                              int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes, float* v_keep,
                              void* stream);
int fcd_conv2d_fwd_wino_cat_keepv(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc, const float* U,
                                  const float* bias, float* y, int fuse_relu, void* ws, size_t ws_bytes, float* v_keep,
                                  void* stream);
int fcd_conv2d_bwd_weight_bias_v(const fcd_conv_desc* d, const float* v_fwd, const float* dy, const float* relu_out, float* dw,
                                 float* db, void* ws, size_t ws_bytes, void* stream);
/* ---- ReLU mask of a FROZEN F(4x4) layer as bits (round 3; the VGG16 stack of the perception term, reference Loss.py:25-27:
 * requires_grad = False, so the backward pass needs of a layer's output only its sign).  fcd_conv2d_fwd_wino_relu_bits writes
 * y = relu(conv + bias) AND 16 bits per (n, k, 4 x 4 output tile); fcd_conv2d_bwd_data_wino_bits gates dy with them instead
 * of re-reading y.  fcd_conv_wino_relu_bits_bytes(d) = size of the mask, 0 when the layer does not run this way. */
size_t fcd_conv_wino_relu_bits_bytes(const fcd_conv_desc* d);
int fcd_conv2d_fwd_wino_relu_bits(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                  unsigned short* relu_bits, void* ws, size_t ws_bytes, void* stream);
int fcd_conv2d_bwd_data_wino_bits(const fcd_conv_desc* d, const float* dy, const unsigned short* relu_bits, const float* U,
                                  float* dx, void* ws, size_t ws_bytes, void* stream);
/* ---- runs of FROZEN F(4x4) layers (round 5): the conv + ReLU pairs between two max-pools of the VGG16 stack (reference
 * Loss.py:25-36; torchvision vgg16.features[5..8], [10..15], [17..22], [24..29]).  d[0 .. n) in forward order, same N, H, W,
 * d[i].C == d[i-1].K.  Between two layers of a run ONE kernel turns the products of layer i into the transformed input of
 * layer i + 1 (bias, ReLU, sign bits on the way; the gradient's ReLU gate on the way back): the activation between them is
 * never written -- 4.5 instead of 6.5 tensor passes per layer boundary.  Results are bit-identical to the layer-by-layer calls.
 * fcd_conv_wino_chain_ok(d, n, mode): 1 when the run qualifies (mode 0 forward, 1 data gradient); the data gradient of a run
 * may cover a SUFFIX of the forward run (conv2_1's data gradient is a 64-row GEMM and runs on the fused F(2x2) kernel).
 * relu_bits[i]: sign bits of layer i's activation, fcd_conv_wino_chain_bits_bytes(&d[i]) bytes, written by the forward run
 * (entries / the array may be NULL when no backward pass follows; with pool_y the last entry is not used).
 * Backward: dy = gradient of the run's output, pooled when pool_code != NULL, else gated with relu_bits[n-1] if given;
 * gate_in (optional): sign bits of the run's INPUT activation -- dx is zeroed where it was <= 0 (for a producer whose own
 * data-gradient kernel then runs ungated). */
int fcd_conv_wino_chain_ok(const fcd_conv_desc* d, int n, int mode);
size_t fcd_conv_wino_chain_ws_bytes(const fcd_conv_desc* d, int n, int mode);
size_t fcd_conv_wino_chain_bits_bytes(const fcd_conv_desc* d);
int fcd_conv2d_fwd_wino_chain(const fcd_conv_desc* d, int n, const float* x, const float* const* U, const float* const* bias,
                              float* y, float* pool_y, unsigned char* code, unsigned short* const* relu_bits, void* ws,
                              size_t ws_bytes, void* stream);
int fcd_conv2d_bwd_data_wino_chain(const fcd_conv_desc* d, int n, const float* dy, const unsigned char* pool_code,
                                   const unsigned short* const* relu_bits, const unsigned short* gate_in,
                                   const float* const* U1, float* dx, void* ws, size_t ws_bytes, void* stream);
/* dx from dy, dy * [relu_out > 0] (relu_out != NULL) or the pooled gradient routed by pool_code */
int fcd_conv2d_bwd_data_wino(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                             const unsigned char* pool_code, const float* U, float* dx, void* ws,
                             size_t ws_bytes, void* stream);
/* dx = conv_transpose(dy', w): desc describes the FORWARD conv; wp_bwd: mode-1
 * packed weights.  dx has shape (N,C,H,W).  relu_out (optional, shape of dy): the
 * fused-ReLU forward output; dy' = dy * [relu_out > 0] is formed while staging. */
int fcd_conv2d_bwd_data(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                        const float* wp_bwd, float* dx, void* stream);
/* conv3x3(s1,p1) + bias + ReLU + MaxPool2d(2) in ONE kernel (the conv/ReLU/pool triples of
 * the VGG16 stack, Loss.py:25).  Writes y_pool (N,K,P/2,Q/2) and an argmax code byte per
 * pooled element (bits 0-1: window slot row*2+col of the first maximum, bit 2: max > 0); the
 * full-resolution activation never goes to memory.  Needs K > 32. */
int fcd_conv2d_fwd_relu_pool(const fcd_conv_desc* d, const float* x, const float* wp,
                             const float* bias, float* y_pool, unsigned char* code, void* stream);
/* Data gradient of the above: dy_pool (N,K,P/2,Q/2) + code -> dx (N,C,H,W); the max-pool and
 * ReLU backward are folded into the kernel's patch loader.  Needs C > 32. */
int fcd_conv2d_bwd_data_pooled(const fcd_conv_desc* d, const float* dy_pool,
                               const unsigned char* code, const float* wp_bwd, float* dx,
                               void* stream);
/* dw[k][c][r][s] = sum_{n,p,q} dy[n,k,p,q] * x[n,c,p*stride+r-pad,q*stride+s-pad]
 * (plain, unpacked layout).  Needs ws of fcd_conv2d_bwd_weight_ws_bytes(). */
size_t fcd_conv2d_bwd_weight_ws_bytes(const fcd_conv_desc* d);
int fcd_conv2d_bwd_weight(const fcd_conv_desc* d, const float* x, const float* dy,
                          const float* relu_out, float* dw, void* ws, size_t ws_bytes, void* stream);
/* Same, plus the bias gradient db[k] = sum_{n,p,q} dy' (NULL: skipped): the channel sums are a
 * by-product of the pass that re-lays dy out for the weight-gradient kernel, so dy is not read again. */
int fcd_conv2d_bwd_weight_bias(const fcd_conv_desc* d, const float* x, const float* dy,
                               const float* relu_out, float* dw, float* db, void* ws, size_t ws_bytes,
                               void* stream);
/* out[c] = sum over (n, hw) of x[n,c,hw] (* [relu_out > 0] when given) -- bias gradients. */
size_t fcd_channel_sum_ws_bytes(int C);
int fcd_channel_sum(const float* x, const float* relu_out, float* out, int N, int C, int HW,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm2d (+ fused activation) -------------------------------------
 * Replaces nn.BatchNorm2d + nn.ReLU / nn.LeakyReLU(0.2) / nn.PReLU chains at
 * Module.py:26-31,156,178-181,197-209 and the bare activations at :147,197.
 *
 * The batch is split into `groups` equal sample groups that are normalised
 * independently and update the running statistics one after the other, in
 * order -- this is how the Siamese encoder (Module.py:114-131) and the
 * Discriminator's shared net (Module.py:220-221) call one BN layer several
 * times per step; groups=1 is the ordinary layer.
 *  has_bn=0: pure activation (gamma..save_invstd ignored).
 *  training=1: batch statistics (biased var), running stats updated with
 *    `momentum` using the unbiased variance; save_mean/save_invstd [groups*C].
 *  training=0: running statistics.
 *  act: FCD_ACT_*; slope: device pointer to the scalar slope (PReLU weight) or
 *    NULL with slope_imm used instead (LeakyReLU).
 */
int fcd_bn_act_fwd(const float* x, float* y, int N, int C, int HW, int groups, int has_bn,
                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float momentum, float eps, int training, float* save_mean, float* save_invstd,
                   int act, const float* slope, float slope_imm, void* ws, size_t ws_bytes,
                   void* stream);
/* [r5] Train-mode BatchNorm + activation as fcd_bn_act_fwd (has_bn = 1, training = 1), but the running statistics take the
 * groups' momentum updates in the order order[0 .. norder) (HOST array of group indices; repeats allowed; groups, norder <= 16)
 * instead of 0 .. groups - 1 once each.  For a net the reference calls twice on the SAME samples in one step
 * (Discriminator_SRGAN_simple: x_mask is the first argument of both calls, Demo_RSSS.py:293,302 -> Module.py:220): the shared
 * samples are normalised once, the running statistics end up as after the reference's four calls. */
int fcd_bn_act_fwd_replay(const float* x, float* y, int N, int C, int HW, int groups, const int* order, int norder,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* save_mean, float* save_invstd, int act,
                          const float* slope, float slope_imm, void* ws, size_t ws_bytes, void* stream);

size_t fcd_bn_act_ws_bytes(int C, int groups);
/* Backward of the above.  dz: grad wrt the activated output.  Writes dx and
 * OVERWRITES dgamma/dbeta [C] (sum over groups) and dslope [1] when non-NULL. */
int fcd_bn_act_bwd(const float* dz, const float* x, float* dx, int N, int C, int HW, int groups,
                   int has_bn, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, int training, const float* save_mean,
                   const float* save_invstd, int act, const float* slope, float slope_imm,
                   float* dgamma, float* dbeta, float* dslope, void* ws, size_t ws_bytes,
                   void* stream);

/* Synchronised BatchNorm (optional; reproduces the reference's single-device large-batch
 * statistics under data parallelism): the forward / backward above split where the per-channel
 * sums exist so the host can all-reduce them (RCCL) in between.  Sums are fp64:
 *   stats  out[(g*C+c)*2 + {0,1}] = {sum x, sum x^2} of THIS rank's samples
 *   bwd    out[(g*C+c)*3 + {0,1,2}] = {sum dy', sum dy'*xhat, PReLU-slope term}
 * `count` = samples per group x HW summed over all ranks. */
int fcd_bn_partial_stats(const float* x, double* out, int N, int C, int HW, int groups, void* ws,
                         size_t ws_bytes, void* stream);
int fcd_bn_act_fwd_from_stats(const float* x, float* y, int N, int C, int HW, int groups,
                              const double* sums, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* save_mean, float* save_invstd,
                              int act, const float* slope, float slope_imm, void* ws,
                              size_t ws_bytes, void* stream);
/* train-mode BN + activation from the per-workgroup partial sums of the producing convolution (fcd_wino_fwd_extras.bn_part,
 * split = fcd_conv_wino_bn_split): no statistics pass over x */
int fcd_bn_act_fwd_parts(const float* x, float* y, int N, int C, int HW, int groups, const double* part, int split,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, float* save_mean, float* save_invstd, int act, const float* slope, float slope_imm,
                         void* ws, size_t ws_bytes, void* stream);
/* [r5] the statistics half of a train-mode BatchNorm2d whose normalise + ReLU pass is done by the NEXT convolution's loader
 * (fcd_wino_fwd_extras.in_scale / in_shift; reference Module.py:25-31): mean / invstd saved, running statistics updated, scale =
 * gamma * invstd and shift = beta - mean * scale written per (group, channel).  part / split as in fcd_bn_act_fwd_parts, or NULL / 0
 * (x is read once).  The backward pass is fcd_bn_act_bwd on the convolution's data gradient and the same x. */
int fcd_bn_train_stats(const float* x, int N, int C, int HW, int groups, const double* part, int split, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                       float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);
/* [r5] Encoder tail: train-mode BatchNorm2d + ReLU whose result feeds MaxPool2d(2) and a skip connection (reference Module.py:30-31 ->
 * :43-44 / :116-132).  _fwd writes a = relu(z * scale + shift) and p = maxpool2(a) in one pass (scale / shift from fcd_bn_train_stats);
 * _bwd takes the skip consumer's gradient of a and the gradient of p and writes the BatchNorm input gradient: the pooling argmax and the
 * ReLU gate are recomputed from z, neither a nor the summed gradient is read or written (5 instead of 8 passes).  _plan: shapes taken
 * (even H, W % 4 == 0); 16-B aligned tensors; workspace fcd_bn_act_ws_bytes(C, groups). */
int fcd_bn_relu_pool_plan(int N, int C, int H, int W, int groups);
int fcd_bn_relu_pool_fwd(const float* z, float* a, float* p, int N, int C, int H, int W, int groups, const float* scale,
                         const float* shift, void* stream);
int fcd_bn_relu_pool_bwd(const float* z, const float* dskip, const float* dpool, float* dz, int N, int C, int H, int W, int groups,
                         const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float* dgamma,
                         float* dbeta, void* ws, size_t ws_bytes, void* stream);
int fcd_bn_bwd_partial(const float* dz, const float* x, double* out, int N, int C, int HW,
                       int groups, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, int act, const float* slope, float slope_imm,
                       void* ws, size_t ws_bytes, void* stream);
int fcd_bn_bwd_from_sums(const float* dz, const float* x, float* dx, int N, int C, int HW,
                         int groups, const double* sums, double count, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd,
                         int act, const float* slope, float slope_imm, void* ws, size_t ws_bytes,
                         void* stream);

/* ---- pooling / resampling --------------------------------------------------
 * nn.MaxPool2d(2) Module.py:43 and VGG pools (Loss.py:25); nn.Upsample(x2,
 * bilinear, align_corners=True) Module.py:60; F.avg_pool2d(k2, padding=s%2)
 * ssim.py:213-215; nn.AdaptiveAvgPool2d(1) Module.py:211. */
int fcd_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, void* stream);
int fcd_maxpool2_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, void* stream);
/* [r5] dx = add + maxpool2 backward: x has a second consumer (the decoder's skip connection, reference Module.py:116-132) whose
 * gradient `add` is summed in the same pass (dx may alias add) */
int fcd_maxpool2_bwd_add(const float* x, const float* dy, const float* add, float* dx, int NC, int H, int W, void* stream);
int fcd_upsample2x_fwd(const float* x, float* y, int NC, int H, int W, void* stream);
int fcd_upsample2x_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream);
int fcd_avgpool2_pad_fwd(const float* x, float* y, int NC, int H, int W, void* stream);
int fcd_avgpool2_pad_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream);
/* Discriminator_SRGAN_simple.forward, Module.py:220-223: classifier's AdaptiveAvgPool2d(1) of net(x) - net(y) for the
 * batched feature tensor f = (2 * pairs groups of n samples, C, HW), pair i = (group 2i, group 2i + 1):
 *   d[i * n + s][c] = mean_p (f[(2i) n + s][c][p] - f[(2i + 1) n + s][c][p])   (difference first -- the reference's order --
 * accumulated in fp64).  bwd: df = +/- g / HW, every element of f written. */
int fcd_pair_gap_diff_fwd(const float* f, float* d, int pairs, int n, int C, int HW, void* stream);
int fcd_pair_gap_diff_bwd(const float* g, float* df, int pairs, int n, int C, int HW, void* stream);
/* `x * (1 - cmask).repeat(1, C, 1, 1)` of every tensor entering the Discriminator / perception VGG / SSIM (Demo_RSSS.py:290-300,
 * Demo_WSSS.py:264-277, Loss.py:78-79,111-112), written as ONE batch: z[i * N + n][c][p] = src_i[n][c][p] * (1 - cmask[n][p]),
 * i < k <= 4 sources of N*C*HW floats (unused source pointers NULL), cmask N*HW floats, z k*N*C*HW floats.  Forward values are
 * bit-identical to the ATen sequence.  bwd: dcmask[n][p] = -sum_i sum_c dz_i * src_i (fp64 accumulator; NULL: not wanted),
 * d_i = dz_i * (1 - cmask) for every non-NULL d_i. */
int fcd_masked_stack_fwd(const float* s0, const float* s1, const float* s2, const float* s3, int k, const float* cmask, float* z,
                         int N, int C, int HW, void* stream);
int fcd_masked_stack_bwd(const float* dz, const float* s0, const float* s1, const float* s2, const float* s3, int k,
                         const float* cmask, float* dcmask, float* d0, float* d1, float* d2, float* d3, int N, int C, int HW,
                         void* stream);

/* ---- raw-tile normalisation (NORMALIZE, CommonFunc.py:199-224, as GDALDataset applies it: data_utils.py:106-116)
 * out[n][c] = float((double(x[n][c]) - mean[c]) / std[c]) where valid[n] != 0, else 0.  x, out: N*C*HW floats;
 * valid: N*HW floats (the window of the patch that holds scene pixels) or NULL; mean, std: C doubles on the device.
 * fp64 per element: bit-identical to the reference's host-side float64 arithmetic followed by .float(). */
int fcd_normalize_tiles(const float* x, const float* valid, const double* mean, const double* stdv, float* out,
                        int N, int C, int HW, void* stream);

/* ---- loss terms ------------------------------------------------------------
 * Masked per-sample sums -- the reconstruction terms of Loss.py:76-84 (L1) and
 * :110-119 (MSE), and region_loss Loss.py:127-141:
 *   d = (a - b) * w,  w = complement ? (1 - m) : m   (m is (N,1,HW), broadcast
 *   over the C channels of a, b; b may be NULL = 0)
 *   num[n]  = sum_{c,p} |d|  (kind 0)   or   d^2  (kind 1)
 *   wsum[n] = sum_p w
 * out2[2*N] = {num[0..N), wsum[0..N)}.  The per-sample ratio / skip-if-zero
 * logic on these N numbers is host side. */
size_t fcd_masked_recon_ws_bytes(int N);
int fcd_masked_recon_fwd(const float* a, const float* b, const float* m, float* out2, int N, int C,
                         int HW, int kind, int complement, void* ws, size_t ws_bytes, void* stream);
/* Backward of  L = sum_n coef[n]*num[n] + cw[n]*wsum[n]  (coef, cw: device
 * arrays of N).  Writes (each optional, may be NULL) da, db (N,C,HW), dm (N,1,HW). */
int fcd_masked_recon_bwd(const float* a, const float* b, const float* m, const float* coef,
                         const float* cw, float* da, float* db, float* dm, int N, int C, int HW,
                         int kind, int complement, void* stream);
/* mean_n(num[n] * scale / wsum[n]) over the out2 = {num[N], wsum[N]} of fcd_masked_recon_fwd -- the tail of the per-sample loops
 * Loss.py:82-84,115-119,135-138 (skip_zero: samples with wsum == 0 are skipped, the reference's `continue`, but still counted in
 * the mean) -- and its adjoint in the form fcd_masked_recon_bwd takes: coef[n] = dL/dnum[n], cw[n] = dL/dwsum[n] for the upstream
 * gradient g[0]. */
int fcd_ratio_mean_fwd(const float* out2, int N, float scale, int skip_zero, float* loss, void* stream);
int fcd_ratio_mean_bwd(const float* g, const float* out2, int N, float scale, int skip_zero, float* coef, float* cw, void* stream);

/* SSIM level (ssim.py:55-92): 11-tap (win) separable VALID Gaussian window
 * statistics of X,Y -> per-(n,c) means of ssim_map and cs_map.
 * out[2*NC] = {ssim_mean[NC], cs_mean[NC]}. Requires H,W >= win_size. */
int fcd_ssim_level_fwd(const float* X, const float* Y, const float* win, int win_size, float* out,
                       int NC, int H, int W, float C1, float C2, void* ws, size_t ws_bytes,
                       void* stream);
size_t fcd_ssim_ws_bytes(int NC, int H, int W);
/* Backward: g_ssim[NC], g_cs[NC] are the upstream grads of the two means;
 * writes dX, dY (N*C*H*W each).  ws: fcd_ssim_ws_bytes() (adjoint maps). */
int fcd_ssim_level_bwd(const float* X, const float* Y, const float* win, int win_size,
                       const float* g_ssim, const float* g_cs, float* dX, float* dY, int NC, int H,
                       int W, float C1, float C2, void* ws, size_t ws_bytes, void* stream);

/* ---- data gradient of 3x3 / stride-2 / pad-1 convolutions by sub-pixel decomposition: one stride-1 pseudo-convolution with
 * 2x2 taps over dy producing the four phases of dx (16 instead of 36 multiplies per input-pixel quad), rows scattered to
 * their sub-pixel positions by the epilogue.  fcd_conv_s2_dgrad_plan: 1 when the layer takes this form; filters are then
 * packed with fcd_conv_s2_dgrad_pack (fcd_conv_s2_dgrad_packed_elems floats).  relu_out as in fcd_conv2d_bwd_data. */
int fcd_conv_s2_dgrad_plan(const fcd_conv_desc* d);
int64_t fcd_conv_s2_dgrad_packed_elems(int K, int C);
int fcd_conv_s2_dgrad_pack(const float* w, float* wp, int K, int C, void* stream);
int fcd_conv2d_bwd_data_s2(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* wp_s2, float* dx,
                           void* stream);

/* ---- fused Winograd F(2x2,3x3) convolution for 3x3 / stride-1 / pad-1 layers with 33..64 GEMM rows (output channels
 * forward, input channels for the data gradient) and >= 32 reduction channels: input transform, 16 batched MFMA GEMMs
 * and output transform in ONE kernel (csrc/conv_wino2.hip) -- 2.25x fewer multiplies than the direct kernel at the same
 * HBM traffic, results within direct-kernel rounding (F(2x2) transforms are exact in fp32 up to a few ulp).
 * fcd_conv_wino2_plan: 1 when the library runs the layer (mode 0 forward / 1 data gradient) on this kernel -- the
 * caller then packs the filters with fcd_conv_wino2_pack (fcd_conv_wino2_filter_elems floats) instead of
 * fcd_conv_pack_weights.  Forward: y = act(conv + bias) + residual (act / slope / residual as fcd_conv2d_fwd_ex), or,
 * with pool_y != NULL, pool_y / code = maxpool2(relu(conv + bias)) as fcd_conv2d_fwd_relu_pool.  Data gradient:
 * source gating as fcd_conv2d_bwd_data (relu_out) / fcd_conv2d_bwd_data_pooled (pool_code).  No workspace. */
int fcd_conv_wino2_plan(const fcd_conv_desc* d, int mode);
int64_t fcd_conv_wino2_filter_elems(int K, int C, int mode);
int fcd_conv_wino2_pack(const float* w, float* U, int K, int C, int mode, void* stream);
int fcd_conv2d_fwd_wino2(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y, int act,
                         const float* slope_ptr, float slope_imm, const float* residual, float* pool_y,
                         unsigned char* code, void* stream);
int fcd_conv2d_bwd_data_wino2(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                              const unsigned char* pool_code, const float* U, float* dx, void* stream);

/* ---- optimizers (torch.optim.Adam / RMSprop defaults; Demo_RSSS.py:151-158)
 * Flat fp32 buffers of n elements.  grad_scale multiplies the gradient first
 * (1/world_size after an all-reduce-sum).  step: 1-based step count. */
int fcd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale,
                  void* stream);
int fcd_rmsprop_step(float* p, const float* g, float* sq, int64_t n, float lr, float alpha,
                     float eps, float weight_decay, float grad_scale, void* stream);
/* The same updates with the per-step scalars in DEVICE memory (train steps replayed from a hipGraph: a replay re-issues the
 * recorded arguments, so the learning-rate schedule of CommonFunc.py:23-37 and Adam's bias correction reach the kernel through
 * memory).  hyper: floats {lr, 1 - beta1^t, sqrt(1 - beta2^t)} (Adam) / {lr} (RMSprop), written by the host before each step. */
int fcd_adam_step_h(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                    float beta2, float eps, float weight_decay, float grad_scale, void* stream);
int fcd_rmsprop_step_h(float* p, const float* g, float* sq, int64_t n, const float* hyper, float alpha,
                       float eps, float weight_decay, float grad_scale, void* stream);

/* ---- profiling -------------------------------------------------------------
 * When enabled every launch is bracketed by HIP events on its stream; read()
 * synchronises and returns per-family totals: out[f*4+0]=ms, +1=launches,
 * +2=algorithmic FLOPs (direct-convolution count, also for the Winograd families), +3=bytes
 * (f < fcd_prof_families()).  enable(2) additionally keeps a per-launch log (family, layer tag, ms,
 * FLOPs, bytes); fcd_prof_detail_read() returns it as tab-separated text lines (call fcd_prof_read() first:
 * it resolves the events), returns the buffer size needed incl. NUL, clears the log when reset. */
void fcd_prof_enable(int on);
int64_t fcd_prof_detail_read(char* buf, int64_t cap, int reset);
int fcd_prof_families(void);
int fcd_prof_read(double* out, int reset);
const char* fcd_prof_family_name(int f);

#ifdef __cplusplus
}
#endif
#endif
