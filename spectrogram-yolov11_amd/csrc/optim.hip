// optim.hip — the trainer's optimizer_step (engine/trainer.py:585-593 in the reference: GradScaler.unscale_ ->
// clip_grad_norm_(10.0) -> optimizer.step (SGD nesterov / AdamW, three parameter groups) -> GradScaler.update -> zero_grad ->
// ModelEMA.update, utils/torch_utils.py:495-531) over the FLAT parameter / gradient / momentum / EMA buffers of engine/flat.py,
// as TWO launches:
//   K1 grad_norm  per-workgroup partial sums of (g * inv_scale)^2 in a FIXED slot each (no atomics: the fold in K2 is ordered,
//                 so the clip factor — and with it every weight — is bit-reproducible), plus a snapshot of the loss scale and
//                 of the Adam step counter (K2's workgroup 0 updates both in place while other workgroups may still start);
//   K2 step       every workgroup folds the partials in the same order -> total norm, found_inf, clip factor; then per element:
//                 unscale, clip, SGD-nesterov or AdamW update of (p, momentum / moments), EMA of the updated parameter, gradient
//                 zeroed for the next window; extra workgroups average the float buffers (BN running statistics) into the EMA;
//                 workgroup 0 applies GradScaler.update (backoff on overflow, growth after `growth_interval` clean steps).
// All streaming: p, g, m, ema read once and written once (AdamW: v too).  HBM-bound: ~8 x 4 B per parameter.
#include "common.h"

#define OPT_HDR 4     // ws[0] = inv_scale snapshot, ws[1] = Adam step snapshot, ws[2..3] reserved; ws[OPT_HDR + i] = partial i

__global__ __launch_bounds__(256) void opt_grad_norm_kernel(long n4, const float4* __restrict__ g, const float* __restrict__ scale,
                                                            const float* __restrict__ adam_step, float* __restrict__ ws) {
  __shared__ float red[4];
  const float inv = scale ? 1.0f / scale[0] : 1.0f;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = g[i];
    const float a = v.x * inv, b = v.y * inv, c = v.z * inv, d = v.w * inv;
    s += a * a + b * b + c * c + d * d;
  }
  // ordered fold: lanes by xor-shuffle (a fixed tree), then the four waves in index order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[OPT_HDR + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    if (blockIdx.x == 0) {
      ws[0] = inv;
      ws[1] = adam_step ? adam_step[0] : 0.f;
    }
  }
}

struct OptGroup {
  long begin4, end4;          // float4 index range of the group inside the flat buffers
  float lr, momentum, weight_decay, pad;
};
struct OptArgs {
  long n4, nbuf;
  float4* p; float4* g; float4* m; float4* v; float4* ema;
  const float* buf; float* ema_buf;
  const float* ws; int nparts;
  OptGroup grp[3];
  int kind;                   // 0 = SGD (nesterov, dampening 0), 1 = AdamW (beta2 0.999-style: `beta2`, `eps`)
  float beta2, eps, max_norm, ema_decay;
  int amp;                    // 1: ws[0] holds 1/scale and a non-finite norm skips the update (GradScaler.step)
  float* scale; int* growth_tracker; float growth, backoff; int growth_interval;
  float* adam_step;
  float* norm_out;            // optional: [0] = total gradient norm (unscaled, before clipping), [1] = found_inf
};

__global__ __launch_bounds__(256) void opt_step_kernel(const OptArgs a, int param_blocks) {
  __shared__ float red[256];
  __shared__ float s_total;
  // every workgroup folds the K1 partials in the same fixed order: thread t takes partials t, t+256, ... then a binary tree
  float s = 0.f;
  for (int i = threadIdx.x; i < a.nparts; i += 256) s += a.ws[OPT_HDR + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) s_total = red[0];
  __syncthreads();
  const float total = s_total;
  const bool finite = (total == total) && (fabsf(total) <= 3.0e38f);
  const bool skip = a.amp && !finite;
  const float inv = a.ws[0];
  const float norm = sqrtf(total);
  float coef = a.max_norm / (norm + 1e-6f);               // clip_grad_norm_: ONE factor for all parameters, clamped to 1
  coef = coef > 1.f ? 1.f : coef;
  const float gs = inv * coef;
  const float d = a.ema_decay, od = 1.f - a.ema_decay;

  if ((int)blockIdx.x >= param_blocks) {                  // EMA of the float buffers (BN running statistics)
    for (long i = (long)(blockIdx.x - param_blocks) * 256 + threadIdx.x; i < a.nbuf; i += (long)(gridDim.x - param_blocks) * 256)
      a.ema_buf[i] = d * a.ema_buf[i] + od * a.buf[i];
    return;
  }
  // bias corrections (AdamW): the step this update is (count of non-skipped steps so far + 1)
  const float t = a.ws[1] + 1.f;
  float bc1[3], bc2s = 1.f;
  if (a.kind != 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) bc1[k] = 1.f - powf(a.grp[k].momentum, t);
    bc2s = sqrtf(1.f - powf(a.beta2, t));
  }
  const float4 zero = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (long)param_blocks * 256) {
    const int k = i < a.grp[0].end4 ? 0 : (i < a.grp[1].end4 ? 1 : 2);
    const float lr = a.grp[k].lr, mom = a.grp[k].momentum, wd = a.grp[k].weight_decay;
    float4 p = a.p[i];
    if (!skip) {
      const float4 g4 = a.g[i];
      float4 m = a.m[i];
      float pv[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g4.x * gs, g4.y * gs, g4.z * gs, g4.w * gs}, mv[4] = {m.x, m.y, m.z, m.w};
      if (a.kind == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float g = gv[q];
          if (wd != 0.f) g += wd * pv[q];                 // torch.optim.SGD: decay joins the gradient
          const float b = mom * mv[q] + g;                // zero-initialised buffer: first step gives b = g (torch's clone)
          mv[q] = b;
          pv[q] -= lr * (g + mom * b);                    // nesterov
        }
      } else {
        float4 v4 = a.v[i];
        float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float g = gv[q];
          if (a.kind == 1) pv[q] *= 1.f - lr * wd;         // AdamW: decoupled decay
          else if (wd != 0.f) g += wd * pv[q];             // torch.optim.Adam (kind 2): L2 joins the gradient before the moments
          const float mn = mom * mv[q] + (1.f - mom) * g;
          const float vn = a.beta2 * vv[q] + (1.f - a.beta2) * g * g;
          mv[q] = mn; vv[q] = vn;
          pv[q] -= (lr / bc1[k]) * mn / (sqrtf(vn) / bc2s + a.eps);
        }
        a.v[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
      p = make_float4(pv[0], pv[1], pv[2], pv[3]);
      a.p[i] = p;
      a.m[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    }
    a.g[i] = zero;                                         // zero_grad for the next accumulation window
    if (a.ema) {
      float4 e = a.ema[i];
      e.x = d * e.x + od * p.x; e.y = d * e.y + od * p.y; e.z = d * e.z + od * p.z; e.w = d * e.w + od * p.w;
      a.ema[i] = e;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.norm_out) { a.norm_out[0] = norm; a.norm_out[1] = skip ? 1.f : 0.f; }
    if (a.adam_step && !skip) a.adam_step[0] = t;
    if (a.amp && a.scale) {                                // GradScaler.update (amp_update_scale)
      if (!finite) { a.scale[0] *= a.backoff; a.growth_tracker[0] = 0; }
      else {
        const int ok = a.growth_tracker[0] + 1;
        if (ok == a.growth_interval) {
          const float ns = a.scale[0] * a.growth;
          if (fabsf(ns) <= 3.0e38f) a.scale[0] = ns;
          a.growth_tracker[0] = 0;
        } else a.growth_tracker[0] = ok;
      }
    }
  }
}

extern "C" int sy11_opt_workspace_floats(int32_t nparts) { return OPT_HDR + (nparts > 0 ? nparts : 0); }

extern "C" int sy11_opt_grad_norm(int64_t n, const float* grad, const float* scale, const float* adam_step, float* ws, int32_t nparts,
                                  void* stream) {
  SY11_REQUIRE(n > 0 && n % 4 == 0 && grad && ws && nparts > 0 && nparts <= 4096, "opt_grad_norm: bad argument (n must be a multiple of 4, 1 <= nparts <= 4096)");
  SY11_REQUIRE(((uintptr_t)grad & 15) == 0, "opt_grad_norm: the gradient buffer must be 16-byte aligned");
  hipLaunchKernelGGL(opt_grad_norm_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, (long)(n / 4), (const float4*)grad, scale, adam_step, ws);
  SY11_LAUNCH_CHECK("opt_grad_norm");
  return SY11_OK;
}

extern "C" int sy11_opt_step(const sy11_opt_desc* d, float* param, float* grad, float* mom, float* sq, float* ema, const float* buf,
                             float* ema_buf, const float* ws, float* scale, int32_t* growth_tracker, float* adam_step, float* norm_out,
                             void* stream) {
  SY11_REQUIRE(d && param && grad && mom && ws, "opt_step: null pointer");
  SY11_REQUIRE(d->n > 0 && d->n % 4 == 0 && d->nparts > 0 && d->nparts <= 4096, "opt_step: bad n / nparts");
  SY11_REQUIRE(d->kind == 0 || ((d->kind == 1 || d->kind == 2) && sq && adam_step),
               "opt_step: kind must be 0 (SGD), 1 (AdamW) or 2 (Adam, coupled L2); 1 and 2 need `sq` and `adam_step`");
  SY11_REQUIRE(d->n_buf == 0 || (buf && ema_buf && ema), "opt_step: buffers given without their EMA");
  SY11_REQUIRE(!d->amp || (scale && growth_tracker), "opt_step: amp needs the scale and growth-tracker scalars");
  for (const void* q : {(const void*)param, (const void*)grad, (const void*)mom, (const void*)ema, (const void*)sq})
    SY11_REQUIRE(((uintptr_t)q & 15) == 0, "opt_step: flat buffers must be 16-byte aligned");
  OptArgs a{};
  a.n4 = d->n / 4; a.nbuf = d->n_buf;
  a.p = (float4*)param; a.g = (float4*)grad; a.m = (float4*)mom; a.v = (float4*)sq; a.ema = (float4*)ema;
  a.buf = buf; a.ema_buf = ema_buf; a.ws = ws; a.nparts = d->nparts;
  long prev = 0;
  for (int k = 0; k < 3; ++k) {
    SY11_REQUIRE(d->group_end[k] % 4 == 0 && d->group_end[k] >= prev && d->group_end[k] <= d->n, "opt_step: group %d ends at %ld (groups are consecutive, multiples of 4)", k, (long)d->group_end[k]);
    a.grp[k].begin4 = prev / 4; a.grp[k].end4 = d->group_end[k] / 4;
    a.grp[k].lr = d->lr[k]; a.grp[k].momentum = d->momentum[k]; a.grp[k].weight_decay = d->weight_decay[k];
    prev = d->group_end[k];
  }
  SY11_REQUIRE(prev == d->n, "opt_step: the three groups must cover the flat buffer");
  a.kind = d->kind; a.beta2 = d->beta2; a.eps = d->eps; a.max_norm = d->max_norm; a.ema_decay = d->ema_decay; a.amp = d->amp;
  a.scale = scale; a.growth_tracker = growth_tracker; a.growth = d->growth_factor; a.backoff = d->backoff_factor; a.growth_interval = d->growth_interval;
  a.adam_step = adam_step; a.norm_out = norm_out;
  long pb = (a.n4 + 256L * 4 - 1) / (256L * 4);
  pb = pb < 1 ? 1 : (pb > 2048 ? 2048 : pb);
  long bb = d->n_buf > 0 ? (d->n_buf + 256L * 4 - 1) / (256L * 4) : 0;
  bb = bb > 64 ? 64 : bb;
  hipLaunchKernelGGL(opt_step_kernel, dim3((unsigned)(pb + bb)), dim3(256), 0, (hipStream_t)stream, a, (int)pb);
  SY11_LAUNCH_CHECK("opt_step");
  return SY11_OK;
}
