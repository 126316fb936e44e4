/* sy11.h — C-ABI of libsy11.so: the MI355X (gfx950) kernels behind the Spectrogram-YOLOv11 hot path.
 *
 * The reference (a fork of Ultralytics 8.3.70) has NO native code and NO FFI on this path: every op is a
 * stock ATen call issued from Python modules.  Each entry point below therefore names the reference
 * *Python call site* whose arithmetic it replaces (file:line under /root/reference/ultralytics), which is
 * what a maintainer would re-bind (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - all pointers are DEVICE pointers borrowed from the caller (torch tensors); the library allocates
 *     nothing persistent and keeps no global state; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*), re-entrant per stream, and never synchronises the host;
 *   - activations are NHWC; a tensor view is (pointer to first used channel, `ld` = elements between
 *     consecutive pixels), so channel slices of a wider buffer ("concat by pointer") need no copy;
 *   - conv weights are [Cout][KH][KW][Cin/groups] (= torch OIHW tensor in channels_last memory format);
 *   - dtype codes: 0 = f32, 1 = f16, 2 = bf16.  Accumulation is always f32;
 *   - return value: 0 on success, negative sy11_status on error; sy11_last_error() gives the message
 *     (thread-local).  Shapes/strides are validated on the host BEFORE any launch.
 */
#ifndef SY11_H
#define SY11_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SY11_VERSION 100

enum sy11_status { SY11_OK = 0, SY11_EINVAL = -1, SY11_EUNSUPPORTED = -2, SY11_ELAUNCH = -3 };
enum sy11_dtype { SY11_F32 = 0, SY11_F16 = 1, SY11_BF16 = 2, SY11_U8 = 3 /* image entries only */ };

/* epilogue / behaviour flags for the conv entry points */
#define SY11_EPI_SILU 1u       /* y = silu(acc + bias)                                   */
#define SY11_EPI_ACCUM 2u      /* y += acc (gradient accumulation into an existing view)  */
#define SY11_EPI_OUT_F32 4u    /* y is f32 whatever desc.dtype says (Detect logits)       */

typedef struct sy11_conv_desc {
  int32_t dtype;               /* element type of x, w, (y)                                  */
  int32_t B, IH, IW, C;        /* input: batch, height, width, channels CONSUMED              */
  int32_t x_ld;                /* input pixel stride (elements), >= C                          */
  int32_t OH, OW, N;           /* output: height, width, channels PRODUCED                     */
  int32_t y_ld;                /* output pixel stride (elements), >= N                         */
  int32_t KH, KW, SH, SW, PH, PW, DH, DW;
  int32_t groups;              /* 1, or == C == N (depthwise)                                  */
  uint32_t flags;              /* SY11_EPI_*                                                   */
  int32_t stat_slots;          /* stat_sum / stat_sq are [stat_slots][N]; workgroup i adds into slot i % stat_slots
                                  (same-address f32 atomics retire at ~25 ns each: spreading them keeps the BN
                                  statistics off the critical path).  0 or 1 = a single [N] row.               */
} sy11_conv_desc;

int sy11_version(void);
const char* sy11_last_error(void);

/* ---- run-time options (no reference counterpart; "tune" stands where the reference has torch.backends.cudnn.benchmark,
 *      utils/torch_utils.py:488):
 *   "tune" 0|1            first-call tile autotuner (0: no NEW measurements; recorded / imported picks are still honoured)
 *   "tune_log" 0|1, "igemm_cfg" / "wgrad_cfg" (-1 = automatic, else force one tile configuration), "igemm_korder" 0|1,
 *   "igemm_deep" 0|1|2 (deeper LDS rings), "igemm_bpol" 0|1|2 (cache policy of the filter-row copies),
 *   "dgrad_s2_halo" 0|1 (3x3 stride-2 input gradient: four igemm launches, one per output parity / ONE fused-parity pass),
 *   "row_map" 0|1 (row walk of the BatchNorm and copy kernels: strided grid of r01 / contiguous chunk per workgroup),
 *   "deterministic" 0|1   ordered reductions (cfg/default.yaml:29 `deterministic: True`, utils/torch_utils.py:474-492): every sum
 *                         across workgroups — BatchNorm statistics and backward sums, filter / bias gradients, loss partials —
 *                         goes through one partial row per workgroup and a fixed-shape fold instead of f32 atomics; bit-identical
 *                         results run to run when the tile choice is pinned ("tune" 0 or an imported pick table).  The partial
 *                         rows live in a per-stream workspace the LIBRARY allocates (grown during eager warm-up, never under capture).
 *                         Cost on the headline workload: +3.8 ... +4.9 % per training step (r04).
 * Defaults come from the environment (SY11_TUNE, SY11_TUNE_LOG, SY11_IGEMM_CFG, SY11_WGRAD_CFG, SY11_IGEMM_KORDER,
 * SY11_IGEMM_DEEP, SY11_IGEMM_BPOL, SY11_DGRAD_S2_HALO, SY11_ROW_MAP, SY11_DETERMINISTIC).  Process-wide; set them between
 * launches, not concurrently with them.                                                                                */
int sy11_set_option(const char* name, int32_t value);
int sy11_get_option(const char* name, int32_t* value);
/* The autotuner's pick tables as a flat array of 16-byte records {u64 problem hash, i32 kind, i32 pick}: export on one
 * rank, broadcast, import on the others, so every rank of a data-parallel job (engine/trainer.py:217-228) runs the same
 * kernels.  sy11_tune_export returns the number of bytes needed (pass buf = NULL to size the buffer).                 */
int64_t sy11_tune_export(void* buf, int64_t capacity_bytes);
int sy11_tune_import(const void* buf, int64_t bytes);
int sy11_tune_clear(void);
/* Measurement helper (bench.py `peaks`): a pure-MFMA loop, `iters` x 8 v_mfma_f32_32x32x16_f16 per wave, 4 waves per
 * workgroup, no memory traffic.  FLOPs per launch = workgroups * 4 * iters * 8 * 32768; out: workgroups * 4 floats.     */
int sy11_peak_mfma_f16(int32_t workgroups, int32_t iters, float* out, void* stream);
/* Diagnostic (tools/igemm8_stamps.py, never on the product path): with SY11_IGEMM_DEBUG=9 the 8-wave convolution pipeline
 * (csrc/igemm8.hip) sums per-section s_memtime deltas in waves 0 / 4 of workgroup 0; this copies the 2 x 8 counters of the
 * last such launch to the HOST array out16 (synchronous).  With out16[0] == 1 on entry the array must hold 16 + 8 * 2048
 * elements: the per-workgroup records {entry, exit (100 MHz ticks), XCC id, cycles of the four kernel sections, 0} of the first 2048
 * workgroups follow.                                                                                                      */
int sy11_debug_stamps(uint64_t* out16);
/* ... and of its persistent form (csrc/igemm8p.hip): per workgroup (256) {cycles waiting for the first stages, main loops, next-tile
 * addresses + first copies, epilogues, tiles, 100 MHz ticks from start to end, 0, 0} -> 2048 values on the host.                  */
int sy11_debug_stamps_persistent(uint64_t* out2048);

/* ---- convolution (replaces nn.Conv2d inside Conv.forward / forward_fuse, nn/modules/conv.py:79-83,
 *      and the bare nn.Conv2d heads of Detect, nn/modules/head.py:44-55) -------------------------------- */

/* y[b,oy,ox,n] = sum_{r,s,c} x[b, oy*SH-PH+r*DH, ox*SW-PW+s*DW, c] * w[n,r,s,c]  (+bias[n]) (silu)
 * groups==1: implicit-GEMM on MFMA.  groups==C==N: direct depthwise kernel.
 * stat_sum/stat_sq (f32[N], may be NULL): per-channel sum and sum of squares of the f32 accumulators are
 * atomically ADDED (BatchNorm batch statistics of F.batch_norm(training=True), conv.py:81).               */
int sy11_conv2d_fwd(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                    float* stat_sum, float* stat_sq, void* stream);

/* Conv.forward in train mode (conv.py:81: bn(conv(x)) with batch statistics): the convolution above with its BatchNorm
 * statistics FINALISED in the kernel tail — the last workgroup to finish folds the statistic slots and writes mean / rstd
 * (saved for backward), scale = gamma*rstd, shift = beta - mean*scale, and the running-stat update, exactly what
 * sy11_bn_finalize computes.  ticket: zero-initialised int32 words (one per group for grouped convs), returned at zero.   */
typedef struct sy11_bn_tail {
  const float* gamma;
  const float* beta;
  float* running_mean;         /* both NULL: no running-stat update */
  float* running_var;
  float* mean;                 /* outputs, [N] each */
  float* rstd;
  float* scale;
  float* shift;
  int32_t* ticket;
  float eps, momentum;
  double count;                /* pixels per channel: B*OH*OW */
} sy11_bn_tail;
int sy11_conv2d_fwd_bn(const sy11_conv_desc* d, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq,
                       const sy11_bn_tail* bn, void* stream);

/* dx = conv_transpose(dy, w): gradient wrt the input of the conv described by `d`
 * (autograd of nn.Conv2d, fired by loss.backward() at engine/trainer.py:388).
 * `dy` has pixel stride dy_ld, `dx` pixel stride d->x_ld.  wt is the tap-transposed filter made by
 * sy11_weight_transpose: [C][KH*KW][N] for groups==1 (ignored for depthwise: pass w).
 * SY11_EPI_ACCUM in d->flags adds into dx instead of overwriting it.                                      */
int sy11_conv2d_dgrad(const sy11_conv_desc* d, const void* dy, int32_t dy_ld, const void* wt, void* dx, void* stream);

/* dw[n,r,s,c] += sum_{b,oy,ox} dy[b,oy,ox,n] * x[b, oy*SH-PH+r*DH, ox*SW-PW+s*DW, c]   (dw is ALWAYS f32 and is
 * accumulated into: zero it for a fresh gradient).                                                         */
int sy11_conv2d_wgrad(const sy11_conv_desc* d, const void* x, const void* dy, int32_t dy_ld, float* dw, void* stream);

/* wt[c][t][n] = w[n][t][c]  (t = r*KW+s), same dtype; feeds sy11_conv2d_dgrad                               */
int sy11_weight_transpose(int32_t dtype, int32_t N, int32_t T, int32_t C, const void* w, void* wt, void* stream);

/* every dgrad filter of a model in one launch.  desc: device array of nlayers records
 * {int32 N, T, C, pad; int64 src_off, dst_off (elements into src / dst); int32 first_tile, tiles_c, tiles_n, pad}
 * sorted by first_tile; a layer owns T * tiles_n * tiles_c tiles of 32x32 (tiles_x = ceil(x/32)).                */
int sy11_weight_transpose_multi(int32_t dtype, int32_t nlayers, int32_t total_tiles, const void* desc, const void* src,
                                void* dst, void* stream);

/* first layer: x is the caller's NCHW f32 image (B,3,H,W) in [0,1] (detect/train.py:59 output); y NHWC.    */
int sy11_stem_conv_fwd(const sy11_conv_desc* d, const float* x_nchw, const void* w, const float* bias, void* y,
                       float* stat_sum, float* stat_sq, void* stream);
int sy11_stem_conv_fwd_bn(const sy11_conv_desc* d, const float* x_nchw, const void* w, void* y, float* stat_sum,
                          float* stat_sq, const sy11_bn_tail* bn, void* stream);
int sy11_stem_conv_wgrad(const sy11_conv_desc* d, const float* x_nchw, const void* dy, int32_t dy_ld, float* dw,
                         void* stream);

/* ---- BatchNorm (+SiLU, + residual) around the conv: nn.BatchNorm2d + nn.SiLU in Conv.forward
 *      (nn/modules/conv.py:81), eps/momentum from utils/torch_utils.py:417-418, residual of Bottleneck
 *      (nn/modules/block.py:725) ------------------------------------------------------------------------- */

/* from sum/sumsq (summed over stat_slots rows of C) over `count` pixels: mean, rstd (saved for backward), scale = gamma*rstd,
 * shift = beta - mean*scale, and the running-stat update (unbiased var, momentum).                          */
int sy11_bn_finalize(int32_t C, int32_t stat_slots, double count, const float* stat_sum, const float* stat_sq, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                     float* mean, float* rstd, float* scale, float* shift, void* stream);

/* z = act(y*scale[c] + shift[c]) (+ res);  y,z,res: M pixels x C channels with their own pixel strides      */
int sy11_bn_act_fwd(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const float* scale,
                    const float* shift, int32_t silu, const void* res, int32_t res_ld, void* z, int32_t z_ld,
                    void* stream);

/* backward, pass 1: g = dz * act'(y*scale+shift);  sum_g[c] += sum g;  sum_gx[c] += sum g * xhat.
 * sum_g / sum_gx are [sum_slots][C] (workgroup i adds into slot i % sum_slots; pass 2 folds the slots).          */
int sy11_bn_act_bwd_reduce(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                           int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                           const float* shift, int32_t silu, float* sum_g, float* sum_gx, int32_t sum_slots,
                           void* stream);
/* backward, pass 2: dy = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M); also dgamma += sum_gx, dbeta += sum_g     */
int sy11_bn_act_bwd_apply(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                          int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                          const float* shift, const float* gamma, int32_t silu, const float* sum_g,
                          const float* sum_gx, int32_t sum_slots, void* dy, int32_t dy_ld, float* dgamma, float* dbeta,
                          void* stream);

/* pass 2 with the gradient of a RESIDUAL operand folded in (Bottleneck shortcut, nn/modules/block.py:725: z = x + act(bn(conv))):
 * res_grad[m, 0:C] (= | +=, f32 add and one rounding as sy11_copy2d) dz[m, 0:C], from the dz values the pass holds anyway —
 * instead of a separate three-pass copy launch.  res_grad == NULL: exactly sy11_bn_act_bwd_apply.                              */
int sy11_bn_act_bwd_apply_res(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                              int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                              const float* shift, const float* gamma, int32_t silu, const float* sum_g,
                              const float* sum_gx, int32_t sum_slots, void* dy, int32_t dy_ld, float* dgamma, float* dbeta,
                              void* res_grad, int32_t res_ld, int32_t res_accumulate, void* stream);

/* ---- data movement inside the graph --------------------------------------------------------------------- */
/* dst[m, 0:C] (= | +=) src[m, 0:C] with independent pixel strides: torch.cat / chunk (conv.py:1821,
 * block.py:466-468) and residual-gradient accumulation.                                                     */
int sy11_copy2d(int32_t dtype, int64_t M, int32_t C, const void* src, int32_t src_ld, void* dst, int32_t dst_ld,
                int32_t accumulate, void* stream);
/* nn.Upsample(None, 2, "nearest") (cfg/models/11/yolo11.yaml:34,38) written straight into a concat slice    */
int sy11_upsample2x_fwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x, int32_t x_ld,
                        void* y, int32_t y_ld, void* stream);
int sy11_upsample2x_bwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* dy, int32_t dy_ld,
                        void* dx, int32_t dx_ld, int32_t accumulate, void* stream);
/* pairwise IoU of the validator's _process_batch (utils/metrics.py:52-72; models/yolo/detect/val.py:205-232):
 * a (n,4), b (m,4) xyxy f32 -> out (n,m) f32, bit-identical to the reference's element-wise evaluation order      */
int sy11_box_iou(int32_t n, int32_t m, const float* a, const float* b, float eps, float* out, void* stream);

/* ---- Fusion('ESChannel') of the fusion model variant (cfg yolo11_fusion_sand3_new.yaml) ----------------------
 * Replaces Fusion.forward (nn/modules/conv.py:2087-2127) = GCT (conv.py:2284-2301) on the channel concat plus
 * WeightedSpatialAttention(3) (conv.py:1839-1852) per input:  out = sum_i x_i * (gate[b, i*C + c] + S_i[b, h, w]).
 * All views NHWC (pointer + pixel stride); C * sizeof(elem) / 16 must be a power of two <= 64 (C = 128 in the model). */
/* per input: mm[pix] = (mean_c x, max_c x), amax[pix] = first argmax channel, sq[b*sq_ld + c] += sum_hw x^2        */
int sy11_fusion_stats(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* x, int32_t x_ld, float* mm,
                      uint16_t* amax, float* sq, int32_t sq_ld, void* stream);
/* S = sigmoid(conv3x3(mm; w)), pad 1, no bias; w = cv1.weight[0] in [KH][KW][2] memory (18 floats)                 */
int sy11_sab_map_fwd(int32_t B, int32_t H, int32_t W, const float* mm, const float* w, float* S, void* stream);
int sy11_sab_map_bwd(int32_t B, int32_t H, int32_t W, const float* dS, const float* S, const float* mm, const float* w,
                     float* dmm, float* dw /* [18] accumulated */, void* stream);
/* gate[b][c] = 1 + tanh(e * gamma / sqrt(mean_c(e^2) + eps) + beta), e = sqrt(sq + eps) * alpha, c in [0, Ct)     */
int sy11_gct_gate_fwd(int32_t B, int32_t Ct, const float* sq, const float* alpha, const float* gamma, const float* beta,
                      float eps, float* G, void* stream);
/* from dG = dL/dgate: q[b][c] with dx += x * q, and dalpha / dgamma / dbeta accumulated over the batch             */
int sy11_gct_gate_bwd(int32_t B, int32_t Ct, const float* sq, const float* alpha, const float* gamma, const float* beta,
                      float eps, const float* dG, float* q, float* dalpha, float* dgamma, float* dbeta, void* stream);
int sy11_fusion_combine(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t n_in, const void* x0, int32_t ld0,
                        const float* S0, const void* x1, int32_t ld1, const float* S1, const void* x2, int32_t ld2,
                        const float* S2, const float* G, void* out, int32_t out_ld, void* stream);
/* per input: dG[b*dg_ld + c] += sum_pix dout * x ; dS[pix] = sum_c dout * x                                        */
int sy11_fusion_bwd_reduce(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* dout, int32_t dout_ld,
                           const void* x, int32_t x_ld, float* dG, int32_t dg_ld, float* dS, void* stream);
/* per input: dx (= | +=) dout * (G + S) + x * q + dmm[.,0] / C + [c == amax] dmm[.,1]                              */
int sy11_fusion_bwd_apply(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* dout, int32_t dout_ld,
                          const void* x, int32_t x_ld, const float* G, const float* q, int32_t g_ld, const float* S,
                          const float* dmm, const uint16_t* amax, void* dx, int32_t dx_ld, int32_t accumulate,
                          void* stream);

/* nn.MaxPool2d(5, 1, 2) (SPPF, nn/modules/block.py:192,197); idx[m,c] = window position (0..24) of the
 * first maximum in row-major scan order (ATen max_pool2d_with_indices tie rule).                            */
int sy11_maxpool5_fwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x, int32_t x_ld,
                      void* y, int32_t y_ld, uint8_t* idx, void* stream);
int sy11_maxpool5_bwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* dy, int32_t dy_ld,
                      const uint8_t* idx, void* dx, int32_t dx_ld, int32_t accumulate, void* stream);
/* flat dtype cast (f32 <-> f16 / bf16), n elements                                                          */
/* backward of a bias conv with f32 output (Detect's last 1x1 convs, nn/modules/head.py:44-55; autograd of nn.Conv2d bias +
 * autocast's cast of the incoming gradient): dz f32 (M x N, pixel stride dz_ld) -> dy (dtype, M x npad contiguous, channels >= N
 * zero) and dbias[c] += sum_m dz[m][c] (dbias may be NULL).  partials != NULL ([partial_rows][N] scratch): ordered reduction —
 * one partial row per workgroup, folded in row order by a second launch (bit-reproducible).                             */
int sy11_bias_grad_cast(int32_t dtype, int64_t M, int32_t N, int32_t npad, const float* dz, int32_t dz_ld, void* dy, float* dbias,
                        float* partials, int32_t partial_rows, void* stream);
int sy11_cast(int32_t src_dtype, int32_t dst_dtype, int64_t n, const void* src, void* dst, void* stream);

/* ---- C2PSA attention core: softmax(q^T k * scale) applied to v (nn/modules/block.py:1925-1931) ---------- */
/* qkv: (B, N, heads*(2*kd+hd)) NHWC pixels, per head [q(kd) k(kd) v(hd)];  o: (B, N, heads*hd);
 * p (f32, B*heads*N*N) receives the attention probabilities (kept for backward).                            */
int sy11_attention_fwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                       int32_t qkv_ld, void* o, int32_t o_ld, float* p, void* stream);
/* workspace: caller-allocated, sy11_attention_workspace_bytes(B, N, heads) bytes (f32 B*heads*N*N: the generic kernels
 * stage dS there; the MFMA path only uses its first B*heads*N floats for the per-query row sums).                    */
int sy11_attention_bwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                       int32_t qkv_ld, const float* p, const void* d_o, int32_t do_ld, void* dqkv, int32_t dqkv_ld,
                       float* workspace, void* stream);
/* the same with the forward output o (B, N, heads*hd) at hand: the per-query row sums of softmax's backward are dO . o, so the MFMA path
 * reads P once instead of twice (o == NULL: identical to sy11_attention_bwd).  autograd's SoftmaxBackward of
 * `attn.softmax(dim=-1)` (nn/modules/block.py:1928) — same sums, taken from the product instead of the factors.            */
int sy11_attention_bwd_o(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                         int32_t qkv_ld, const float* p, const void* o, int32_t o_ld, const void* d_o, int32_t do_ld, void* dqkv,
                         int32_t dqkv_ld, float* workspace, void* stream);
size_t sy11_attention_workspace_bytes(int32_t B, int32_t N, int32_t heads);

/* ---- Detect decode + NMS (nn/modules/head.py:100-131; utils/ops.py:181-332 + torchvision.ops.nms) ------- */
/* maps: 3 NHWC f32 maps (B,H_l,W_l,64+nc); out: (B, 4+nc, A) f32 exactly as Detect._inference returns it.   */
int sy11_detect_decode(int32_t B, int32_t nc, int32_t nl, const float* const* maps, const int32_t* hs,
                       const int32_t* ws, const float* strides, float* out, void* stream);
/* greedy NMS over n candidate boxes ALREADY sorted by (score desc, index asc): keep[i] in {0,1}.
 * boxes: (n,4) xyxy f32 (class offset already added).  workspace: n*ceil(n/64) uint64 words.               */
int sy11_nms_sorted(int32_t n, const float* boxes, float iou_thres, uint64_t* workspace, uint8_t* keep,
                    void* stream);
size_t sy11_nms_workspace_bytes(int32_t n);
/* The same for `nseg` images at once: image i owns counts[i] consecutive rows of boxes / keep (HOST array; rows of an image in
 * score-descending order).  Images run side by side; each sweep stops after `max_keep` survivors (ops.py:322 keeps at most max_det
 * rows per image) and clears the rest of its keep flags.  workspace: as many bytes as the ..._workspace_bytes query below returns.  */
int sy11_nms_sorted_batched(int32_t nseg, const int32_t* counts, const float* boxes, float iou_thres, int32_t max_keep,
                            uint64_t* workspace, uint8_t* keep, void* stream);
size_t sy11_nms_batched_workspace_bytes(int32_t nseg, const int32_t* counts);
/* Candidate selection of non_max_suppression (utils/ops.py:246-292: `x[xc]`, the multi-label `where(cls > conf_thres)` / the
 * best-class `max`, in the reference's candidate order image -> anchor -> class) straight from the (B, D = 4 + nc + nm, A) tensor
 * Detect returns.  Two calls: with blk_offset = NULL it COUNTS, blk_count[b * ceil(A / 256) + k] = candidates of anchors
 * [256 k, 256 k + 256) of image b; with blk_offset = the exclusive prefix sum of those counts it WRITES, per candidate,
 * key = segment << 32 | ~bits(score) (a stable ascending sort of the keys = per segment, score descending, ties in candidate order),
 * its anchor and its class; segment = image, or image * nc + class with segment_by_class.  conf_thres >= 0.          */
int sy11_nms_candidates(int32_t B, int32_t D, int32_t A, int32_t nc, float conf_thres, int32_t multi_label, int32_t segment_by_class,
                        const float* pred, const int32_t* blk_offset, int32_t* blk_count, int64_t* key, int32_t* anchor, int32_t* cls,
                        void* stream);
/* sy11_nms_sorted_batched with the segment list on the DEVICE: rows seg_start[s] .. seg_start[s + 1] (nseg + 1 entries) and first
 * mask word seg_ws[s] of segment s; max_n = the longest segment (host).  One segment per (image, class) is the reference's batched
 * NMS (boxes shifted by class * max_wh never intersect across classes, utils/ops.py:307-312) with n^2 / (2 nc) pair tests.      */
int sy11_nms_sorted_segments(int32_t nseg, const int32_t* seg_start, const int64_t* seg_ws, int32_t max_n, const float* boxes,
                             float iou_thres, int32_t max_keep, uint64_t* workspace, uint8_t* keep, void* stream);

/* ---- fused detection criterion (v8DetectionLoss.__call__, utils/loss.py:221-275; TaskAlignedAssigner, utils/tal.py:40-296;
 *      bbox_iou CIoU, utils/metrics.py:171-234).  maps: nl NHWC f32 head maps (B, H_l*W_l, 64+nc); gt: (B, G, 5) rows
 *      [cls, x1, y1, x2, y2] in pixels, all-zero rows are padding.  Workspaces are caller-allocated:
 *      pbox (B,A,4) f32, align/overlap (B,G,A) f32, topk (B,G,10) i32, assign (B,A) i32, pos (2,B,G) f32 zeroed,
 *      norm (B,A) f32, sums (64,4) f32 zeroed: per-slot partials of {sum(target_scores), box, cls, dfl}.            */
int sy11_det_loss_assign(int32_t B, int32_t nc, int32_t nl, const float* const* maps, const int32_t* hs, const int32_t* ws,
                         const float* strides, int32_t G, const float* gt, float* pbox, float* align, float* overlap,
                         int32_t* topk, int32_t* assign, float* pos, float* norm, float* sums, void* stream);
/* un-normalised loss sums: sums[.][1..3] += sum (1-CIoU)*w, sum BCE, sum DFL*w                                      */
int sy11_det_loss_terms(int32_t B, int32_t nc, int32_t nl, const float* const* maps, const int32_t* hs, const int32_t* ws,
                        const float* strides, int32_t G, const float* gt, const int32_t* assign, const float* norm,
                        float* sums, void* stream);
/* dmaps[l] = d(B * (gb*box + gc*cls + gd*dfl) / tss) / d maps[l], scaled by the DEVICE scalar *upstream and, when given, by the
 * second device scalar *inv_tss (out[4] of sy11_det_loss_finish); with inv_tss = NULL `upstream` must already carry
 * 1 / max(tss, 1).  No host synchronisation.                                                                       */
int sy11_det_loss_bwd(int32_t B, int32_t nc, int32_t nl, const float* const* maps, float* const* dmaps, const int32_t* hs,
                      const int32_t* ws, const float* strides, int32_t G, const float* gt, const int32_t* assign,
                      const float* norm, const float* upstream, const float* inv_tss, float gain_box, float gain_cls,
                      float gain_dfl, void* stream);
/* v8DetectionLoss.preprocess (utils/loss.py:194-207): n targets given as three strided f32 columns (image index, class, xywh
 * normalised; strides in elements) -> gt (B, G, 5) [cls, x1, y1, x2, y2] in pixels (scale_w / scale_h = image width / height),
 * an image's targets in their original order, the rest of its G rows zero.  G = the largest target count of one image.  */
int sy11_det_loss_pack_targets(int32_t n, int32_t B, int32_t G, const float* batch_idx, int32_t idx_stride, const float* cls,
                               int32_t cls_stride, const float* bboxes, int32_t box_stride, float scale_w, float scale_h, float* gt,
                               void* stream);
/* loss.py:268-275: out[0] = batch_size * sum_i gain_i * term_i / max(tss, 1), out[1..3] = the gained items, out[4] = 1 / max(tss, 1);
 * the 64 slots of `sums` are folded in index order.                                                                  */
int sy11_det_loss_finish(const float* sums, int32_t B, float gain_box, float gain_cls, float gain_dfl, float* out, void* stream);

/* ---- trainer step over flat buffers (engine/trainer.py:585-593 optimizer_step: GradScaler.unscale_ -> clip_grad_norm_(10.0) ->
 *      optimizer.step -> GradScaler.update -> zero_grad -> ModelEMA.update, utils/torch_utils.py:495-531; the optimizers are
 *      torch.optim.SGD(nesterov=True) / AdamW as built by build_optimizer, trainer.py:758-819).  Two launches:
 *      sy11_opt_grad_norm writes `nparts` ordered partial sums of (grad / scale)^2 plus snapshots of 1/scale and of the Adam
 *      step counter into ws (sy11_opt_workspace_floats(nparts) floats); sy11_opt_step folds them in a fixed order (bit-
 *      reproducible clip factor), applies the update to param / mom (/ sq), averages the UPDATED parameters and the float
 *      buffers into the EMA, zeroes grad, and updates the loss scale like GradScaler.update.  A non-finite norm with amp = 1
 *      skips the parameter update (GradScaler.step) but not the EMA.  Flat buffers: three consecutive groups ending at
 *      group_end[0..2] (multiples of 4), all 16-byte aligned.                                                            */
typedef struct sy11_opt_desc {
  int64_t n;                   /* elements of param / grad / mom / sq / ema                                      */
  int64_t n_buf;               /* elements of buf / ema_buf (BatchNorm running statistics), may be 0             */
  int64_t group_end[3];        /* end of each parameter group inside the flat buffers                            */
  float lr[3], momentum[3], weight_decay[3];    /* AdamW: momentum = beta1                                      */
  int32_t kind;                /* 0 SGD nesterov (dampening 0), 1 AdamW (decoupled decay), 2 Adam (L2 in the gradient) */
  float beta2, eps;            /* AdamW                                                                          */
  float max_norm;              /* clip_grad_norm_ threshold (10.0)                                               */
  float ema_decay;             /* d of this update: ema = d * ema + (1 - d) * value                              */
  int32_t amp;                 /* 1: gradients carry the loss scale; overflow skips the update                   */
  float growth_factor, backoff_factor; int32_t growth_interval;      /* GradScaler: 2.0, 0.5, 2000              */
  int32_t nparts;              /* partial sums written by sy11_opt_grad_norm (<= 4096)                           */
} sy11_opt_desc;
int sy11_opt_workspace_floats(int32_t nparts);
int sy11_opt_grad_norm(int64_t n, const float* grad, const float* scale, const float* adam_step, float* ws, int32_t nparts,
                       void* stream);
/* scale / growth_tracker: GradScaler's device scalars (updated in place; NULL with amp = 0); adam_step: device f32 count of
 * applied AdamW steps (updated in place; NULL for SGD); norm_out: optional [2] = {total gradient norm, update skipped}.  */
int sy11_opt_step(const sy11_opt_desc* d, float* param, float* grad, float* mom, float* sq, float* ema, const float* buf,
                  float* ema_buf, const float* ws, float* scale, int32_t* growth_tracker, float* adam_step, float* norm_out,
                  void* stream);

/* ---- IQ -> STFT -> power -> mel -> log producer (no reference code: README.md:7; spec in DESIGN.md) ------ */
/* iq: (B, L) interleaved complex64; db: (B, n_frames, n_mel) f32 dB (frame-major: coalesced stores);
 * minmax: (B,2) f32 [min,max] per image
 * (must be pre-filled with +inf/-inf by the caller or by sy11_stft_minmax_init).                            */
int sy11_stft_logmel(int32_t B, int32_t L, int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mel,
                     const float* iq, const float* window, const int32_t* mel_start, const float* mel_w,
                     int32_t mel_taps, float* db, float* minmax, void* stream);
int sy11_stft_minmax_init(int32_t B, float* minmax, void* stream);
/* img[b,c,f,t] = (db[b,t,f]-min)/(max-min), c = 0..2, NCHW f32 (what preprocess_batch hands the model)        */
int sy11_stft_normalize(int32_t B, int32_t n_mel, int32_t n_frames, const float* db, const float* minmax,
                        float* img_nchw, void* stream);

/* ---- image side of preprocess --------------------------------------------------------------
 * batch["img"].float() / 255 of DetectionTrainer.preprocess_batch (models/yolo/detect/train.py:59): n uint8 values
 * -> dtype (true division by 255.0f then rounding to dtype)                                                        */
int sy11_image_u8_to_float(int32_t dtype, int64_t n, const uint8_t* x, void* y, void* stream);
/* the multi_scale branch of preprocess_batch (models/yolo/detect/train.py:60-73): F.interpolate(imgs size=ns
 * mode="bilinear" align_corners=False) on `planes` = B*C planes of IH x IW -> OH x OW.  x_dtype may be SY11_U8:
 * the taps are then divided by 255 first (uint8 batch resized in the same pass)                                    */
int sy11_image_resize_bilinear(int32_t x_dtype, int32_t y_dtype, int32_t planes, int32_t IH, int32_t IW, int32_t OH,
                               int32_t OW, const void* x, void* y, void* stream);
/* LetterBox.__call__ (data/augment.py:1544-1591) fused with BasePredictor.preprocess (engine/predictor.py:118-136):
 * src (sh x sw x 3) uint8 HWC -> cv2.resize(INTER_LINEAR) to new_h x new_w placed at (top left) on an H x W canvas
 * filled with `fill` (114); reverse_c swaps channel 0 and 2 (BGR->RGB); chw=1 writes (3 H W) planes else (H W 3);
 * dtype SY11_U8 keeps bytes (the dataset stage) otherwise value / 255 in dtype (the predictor stage).
 * The 8-bit bilinear follows OpenCV imgproc/src/resize.cpp (opencv-python is a reference dependency and is not
 * vendored): 11-bit coefficients; the ((b*(S>>4))>>16 + ... + 2)>>2 vertical pass; exact 2x shrink = 2x2 box mean */
int sy11_image_letterbox(int32_t dtype, int32_t sh, int32_t sw, int32_t H, int32_t W, int32_t new_h, int32_t new_w,
                         int32_t top, int32_t left, int32_t fill, int32_t reverse_c, int32_t chw,
                         const uint8_t* src, void* dst, void* stream);
/* One training sample of the augmentation pipeline in a single launch (data/augment.py: Mosaic._mosaic4 :660-715 then
 * RandomPerspective.affine_transform :1000-1078 then RandomHSV :1303-1390 then RandomFlip :1393-1474 then the
 * transpose / channel flip of Format._format_img :2070-2107).  The mosaic canvas (canvas_h x canvas_w filled with
 * `fill`) is virtual: tile t (tile_src[t] = device pointer to an (h w 3) uint8 image) covers canvas
 * [x1 x2) x [y1 y2) and canvas pixel (x y) reads source pixel (x - padw  y - padh); tile_geom holds 8 int32 per
 * tile in the order h w x1 y1 x2 y2 padw padh.  minv = the 6 doubles of the INVERTED 2x3 map exactly as
 * cv::warpAffine computes them (NULL: no warp and the output is the canvas); hsv_lut = 768 bytes hue|sat|val
 * tables (NULL: no HSV step).  tile_src / tile_geom / minv / hsv_lut are HOST arrays (copied into the launch).
 * Output H x W written as with sy11_image_letterbox (dtype SY11_U8 or float / 255; chw; reverse_c).
 * cv2.warpAffine and cv2.cvtColor are restated from OpenCV 4.x (imgwarp.cpp fixed-point INTER_LINEAR path;
 * color_hsv.simd.hpp RGB2HSV_b / HSV2RGB_b): opencv-python is an un-vendored reference dependency              */
int sy11_image_mosaic_warp(int32_t dtype, int32_t n_tiles, const uint8_t* const* tile_src, const int32_t* tile_geom,
                           int32_t canvas_h, int32_t canvas_w, const double* minv, int32_t H, int32_t W,
                           const uint8_t* hsv_lut, int32_t flip_ud, int32_t flip_lr, int32_t fill,
                           int32_t reverse_c, int32_t chw, void* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SY11_H */
