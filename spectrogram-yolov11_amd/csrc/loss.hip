// loss.hip — fused v8 detection criterion on the MI355X: box decode, task-aligned assignment, BCE + CIoU + DFL values
// AND their gradient w.r.t. the raw head maps, in 5 launches (the reference runs ~150 ATen kernels + autograd:
// utils/loss.py:172-275, utils/tal.py:14-296, utils/metrics.py:171-234).
//
// Head maps are the engine's NHWC f32 tensors (B, H_l*W_l, 64+nc): one anchor = one 576-byte row.  A group of 16 lanes owns
// an anchor: lane j holds bin j of the four DFL sides (softmax over the 16 lanes by shuffles) and classes j, j+16, ...
// so every global access of a wave is 4 contiguous rows.
//   K1 decode      pred box (grid units) per anchor
//   K2 tal_metrics per (image, gt): in-gt test, CIoU overlap, align = score^0.5 * overlap^6, top-10 (value desc, index asc)
//   K3 tal_resolve per anchor: positive mask, multi-gt conflicts by max overlap (over ALL gts, first max), per-gt maxima
//   K4 tal_norm    per anchor: normalised alignment score (= sum of its target_scores) and its global sum
//   K5 loss_grad   per anchor: loss terms (forward) or d loss / d logits (backward; upstream scalar read from device memory)
#include "common.h"
#include "det.h"

#define REG 16
#define TOPK 10

struct LossArgs {
  const float* maps[4];
  float* dmaps[4];
  int hs[4], ws[4], a0[4];
  float strides[4];
  int B, nc, nl, A, G, no;
  const float* gt;      // (B, G, 5): cls, x1, y1, x2, y2 (pixels); padded rows are all-zero
  float* pbox;          // (B, A, 4) predicted xyxy, grid units
  float* align;         // (B, G, A)
  float* overlap;       // (B, G, A)
  int* topk;            // (B, G, TOPK)
  int* assign;          // (B, A) gt index or -1
  float* pos_align;     // (B, G)
  float* pos_ov;        // (B, G)
  float* norm;          // (B, A)
  float* sums;          // (sum_slots, 4): tss, box, cls, dfl
  int sum_slots;        // 64 (the caller's buffer) or, ordered mode, one row per workgroup of the stream's workspace (det.h)
  const float* gscale;  // backward: upstream gradient (device scalar)
  const float* gscale2; // backward: optional second device factor (1 / max(tss, 1) from loss_finish), or NULL
  float gain_box, gain_cls, gain_dfl;
};

__device__ __forceinline__ void anchor_of(const LossArgs& a, int an, int& l, int& gx, int& gy) {
  l = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.nl && an >= a.a0[i]) l = i;
  const int loc = an - a.a0[l];
  gy = loc / a.ws[l];
  gx = loc - gy * a.ws[l];
}

// CIoU(box1, box2) exactly as metrics.py:199-228 (h gets +eps, w does not; union += eps)
__device__ __forceinline__ float ciou_f(float a1, float b1, float a2, float b2, float c1, float d1, float c2_, float d2) {
  // box1 = (a1,b1,a2,b2), box2 = (c1,d1,c2_,d2)
  const float eps = 1e-7f;
  const float w1 = a2 - a1, h1 = b2 - b1 + eps, w2 = c2_ - c1, h2 = d2 - d1 + eps;
  const float iw = fmaxf(fminf(a2, c2_) - fmaxf(a1, c1), 0.f), ih = fmaxf(fminf(b2, d2) - fmaxf(b1, d1), 0.f);
  const float inter = iw * ih;
  const float uni = w1 * h1 + w2 * h2 - inter + eps;
  const float iou = inter / uni;
  const float cw = fmaxf(a2, c2_) - fminf(a1, c1), ch = fmaxf(b2, d2) - fminf(b1, d1);
  const float cc = cw * cw + ch * ch + eps;
  const float rx = c1 + c2_ - a1 - a2, ry = d1 + d2 - b1 - b2;
  const float rho2 = (rx * rx + ry * ry) * 0.25f;
  const float dat = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = 0.40528473456935109f * dat * dat;     // 4 / pi^2
  const float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / cc + v * alpha);
}

// ---- K1: decode.  One lane per (anchor, side): its 16 bins are 64 contiguous bytes; the softmax expectation runs in registers with
// the SAME summation tree the 16-lane shuffle form had (pairs 8 apart, then 4, 2, 1 — lane 0's view of the xor butterfly), so the
// boxes — and with them the assignment — are bit-identical to r01-r02's kernel at 2.4x fewer vector instructions (r03: 79 -> ~35 us).
__device__ __forceinline__ float tree16(const float (&v)[16]) {
  float a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = v[i] + v[i + 8];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = a[i] + a[i + 4];
  return (b[0] + b[2]) + (b[1] + b[3]);
}
__global__ __launch_bounds__(256) void loss_decode_kernel(const LossArgs a) {
  const long gi4 = (long)blockIdx.x * 256 + threadIdx.x;      // (anchor, side)
  const long gi = gi4 >> 2;
  const int sd = (int)(gi4 & 3);
  if (gi >= (long)a.B * a.A) return;
  const int b = (int)(gi / a.A), an = (int)(gi - (long)b * a.A);
  int l, gx, gy;
  anchor_of(a, an, l, gx, gy);
  const float* row = a.maps[l] + ((long)b * a.hs[l] * a.ws[l] + (an - a.a0[l])) * a.no + sd * REG;
  float x[16];
  if ((a.no & 3) == 0) {                                       // rows of 64 + nc floats start 16-byte aligned when nc % 4 == 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *(const f32x4*)(row + 4 * i);
      x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = row[i];
  }
  float mx = x[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, x[i]);
  float e[16], ej[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    e[i] = expf(x[i] - mx);
    ej[i] = e[i] * (float)i;
    asm volatile("" : "+v"(ej[i]));                            // a rounded product, as in the shuffle form: no contraction into the tree's adds
  }
  const float d = tree16(ej) / tree16(e);
  const float ctr = ((sd & 1) ? gy : gx) + 0.5f;
  a.pbox[gi * 4 + sd] = sd < 2 ? ctr - d : ctr + d;
}

// ---- K2: per (b, g) alignment metrics + top-k
__global__ __launch_bounds__(256) void loss_tal_metrics_kernel(const LossArgs a) {
  extern __shared__ float s_al[];                  // [A]
  __shared__ float r_val[4];
  __shared__ int r_idx[4];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* gt = a.gt + ((long)b * a.G + g) * 5;
  const float gcls = gt[0], x1 = gt[1], y1 = gt[2], x2 = gt[3], y2 = gt[4];
  const bool valid = (x1 + y1 + x2 + y2) > 0.f;    // mask_gt (loss.py:243)
  const int cls = max((int)gcls, 0);
  float* al = a.align + ((long)b * a.G + g) * a.A;
  float* ov = a.overlap + ((long)b * a.G + g) * a.A;
  for (int an = tid; an < a.A; an += 256) {
    int l, gx, gy;
    anchor_of(a, an, l, gx, gy);
    const float st = a.strides[l];
    const float ax = (gx + 0.5f) * st, ay = (gy + 0.5f) * st;
    const float dmin = fminf(fminf(ax - x1, ay - y1), fminf(x2 - ax, y2 - ay));
    float al_v = 0.f, ov_v = 0.f;
    if (valid && dmin > 1e-9f) {
      const float* pb = a.pbox + ((long)b * a.A + an) * 4;
      ov_v = fmaxf(ciou_f(x1, y1, x2, y2, pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st), 0.f);
      const float logit = a.maps[l][((long)b * a.hs[l] * a.ws[l] + (an - a.a0[l])) * a.no + 4 * REG + cls];
      const float sc = 1.f / (1.f + expf(-logit));
      const float o2 = ov_v * ov_v;
      al_v = sqrtf(sc) * (o2 * o2 * o2);
    }
    al[an] = al_v;
    ov[an] = ov_v;
    s_al[an] = al_v;
  }
  __syncthreads();
  int* tk = a.topk + ((long)b * a.G + g) * TOPK;
  if (!valid) {                                    // padded gt: the reference zeroes its indices; it selects nothing
    if (tid < TOPK) tk[tid] = -1;
    return;
  }
  for (int k = 0; k < TOPK; ++k) {
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int an = tid; an < a.A; an += 256) {
      const float v = s_al[an];
      if (v > bv) { bv = v; bi = an; }             // strided scan keeps the lowest index among equals per thread
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ov2 = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
    }
    if ((tid & 63) == 0) { r_val[tid >> 6] = bv; r_idx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = r_val[0];
      int i = r_idx[0];
      for (int w = 1; w < 4; ++w)
        if (r_val[w] > v || (r_val[w] == v && r_idx[w] < i)) { v = r_val[w]; i = r_idx[w]; }
      tk[k] = i;
      s_al[i] = -2.f;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void atomic_max_pos(float* p, float v) { atomicMax((int*)p, __float_as_int(v)); }   // v >= 0

// ---- K3: per anchor resolve
__global__ __launch_bounds__(256) void loss_tal_resolve_kernel(const LossArgs a) {
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (long)a.B * a.A) return;
  const int b = (int)(gi / a.A), an = (int)(gi - (long)b * a.A);
  int l, gx, gy;
  anchor_of(a, an, l, gx, gy);
  const float st = a.strides[l];
  const float ax = (gx + 0.5f) * st, ay = (gy + 0.5f) * st;
  int count = 0, first = -1, best = 0;
  float best_ov = -1.f;
  for (int g = 0; g < a.G; ++g) {
    const float* gt = a.gt + ((long)b * a.G + g) * 5;
    const float x1 = gt[1], y1 = gt[2], x2 = gt[3], y2 = gt[4];
    const float o = a.overlap[((long)b * a.G + g) * a.A + an];
    if (o > best_ov) { best_ov = o; best = g; }    // argmax over ALL gts, first maximum (tal.py:285)
    const bool valid = (x1 + y1 + x2 + y2) > 0.f;
    const float dmin = fminf(fminf(ax - x1, ay - y1), fminf(x2 - ax, y2 - ay));
    if (!(valid && dmin > 1e-9f)) continue;
    const int* tk = a.topk + ((long)b * a.G + g) * TOPK;
    bool sel = false;
#pragma unroll
    for (int k = 0; k < TOPK; ++k) sel |= (tk[k] == an);
    if (sel) { if (first < 0) first = g; ++count; }
  }
  const int asg = count > 1 ? best : first;
  a.assign[gi] = asg;
  if (asg >= 0) {
    const long o = ((long)b * a.G + asg) * a.A + an;
    atomic_max_pos(a.pos_align + (long)b * a.G + asg, a.align[o]);
    atomic_max_pos(a.pos_ov + (long)b * a.G + asg, a.overlap[o]);
  }
}

// ---- K4: normalised alignment per anchor + its sum
__global__ __launch_bounds__(256) void loss_tal_norm_kernel(const LossArgs a) {
  __shared__ float red[4];
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  float nv = 0.f;
  if (gi < (long)a.B * a.A) {
    const int b = (int)(gi / a.A), an = (int)(gi - (long)b * a.A);
    const int asg = a.assign[gi];
    if (asg >= 0) {
      const long o = ((long)b * a.G + asg) * a.A + an;
      nv = a.align[o] * a.pos_ov[(long)b * a.G + asg] / (a.pos_align[(long)b * a.G + asg] + 1e-9f);
    }
    a.norm[gi] = nv;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nv += __shfl_xor(nv, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nv;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(a.sums + (long)(blockIdx.x % a.sum_slots) * 4, ((red[0] + red[1]) + red[2]) + red[3]);
}

// ---- K5: loss terms (BWD = false) or gradient w.r.t. the logits (BWD = true)
template <bool BWD>
__global__ __launch_bounds__(256) void loss_terms_kernel(const LossArgs a, float tss_inv) {
  __shared__ float red[3][16];
  const int grp = threadIdx.x >> 4, j = threadIdx.x & 15;
  float l_box = 0.f, l_cls = 0.f, l_dfl = 0.f;
  // grid-stride over groups of 16 anchors (forward: <= 2048 workgroups, so that the three partial sums of a workgroup end in 6 K
  // atomics per launch instead of 100 K onto the same 64 slots; backward: one group per workgroup, no sums)
  const long BA = (long)a.B * a.A;
  for (long base = (long)blockIdx.x * 16; base < BA; base += (long)gridDim.x * 16) {
  const long gi = base + grp;
  const bool live = gi < BA;
  if (live) {
    const int b = (int)(gi / a.A), an = (int)(gi - (long)b * a.A);
    int l, gx, gy;
    anchor_of(a, an, l, gx, gy);
    const long roff = ((long)b * a.hs[l] * a.ws[l] + (an - a.a0[l])) * a.no;
    const float* row = a.maps[l] + roff;
    float* drow = BWD ? a.dmaps[l] + roff : nullptr;
    const int asg = a.assign[gi];
    const float w = a.norm[gi];                                  // = sum_c target_scores
    float up = 1.f;
    if (BWD) up = a.gscale[0] * (a.gscale2 ? a.gscale2[0] : 1.f) * (float)a.B * tss_inv;            // d(total)/d(term sum): loss = B * sum(gain_i * term_i / tss)
    int label = -1;
    float tx1 = 0, ty1 = 0, tx2 = 0, ty2 = 0;
    const float st = a.strides[l];
    if (asg >= 0) {
      const float* gt = a.gt + ((long)b * a.G + asg) * 5;
      label = max((int)gt[0], 0);
      tx1 = gt[1] / st; ty1 = gt[2] / st; tx2 = gt[3] / st; ty2 = gt[4] / st;   // target_bboxes /= stride (loss.py:266)
    }
    // ---- classification: BCE with logits against onehot(label) * w.  Four classes per lane and access when the row allows it:
    // a group of 16 lanes then moves 256 contiguous bytes per instruction instead of 64 (the rows are 4 * (64 + nc) bytes apart)
    int c_tail = 0;                                              // classes below c_tail go four per lane, the rest one per lane
    if ((a.nc & 3) == 0 && (((uintptr_t)row | (uintptr_t)drow) & 15) == 0) {
      c_tail = a.nc & ~63;                                       // whole rounds of 16 lanes x 4 classes only: every lane stays busy
      for (int c4 = j; c4 < (c_tail >> 2); c4 += 16) {
        const f32x4 x4 = *(const f32x4*)(row + 4 * REG + 4 * c4);
        f32x4 d4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = x4[q];
          const float t = (4 * c4 + q == label) ? w : 0.f;
          if (!BWD) l_cls += fmaxf(x, 0.f) - x * t + __logf(1.f + __expf(-fabsf(x)));
          else d4[q] = (__builtin_amdgcn_rcpf(1.f + __expf(-x)) - t) * a.gain_cls * up;
        }
        if (BWD) *(f32x4*)(drow + 4 * REG + 4 * c4) = d4;
      }
    }
    for (int c = c_tail + j; c < a.nc; c += 16) {
      const float x = row[4 * REG + c];
      const float t = (c == label) ? w : 0.f;
      if (!BWD) {
        l_cls += fmaxf(x, 0.f) - x * t + __logf(1.f + __expf(-fabsf(x)));   // hardware exp2/log2 forms: |error| < 1e-7 per term
      } else {
        const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
        drow[4 * REG + c] = (s - t) * a.gain_cls * up;
      }
    }
    // ---- box branch (softmax statistics of the 4 sides are needed for both terms)
    float p[4], ex[4], lse[4], xs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float x = row[s * REG + j];
      xs[s] = x;
      float mx = x;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
      const float e = expf(x - mx);
      float den = e, num = e * (float)j;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) { den += __shfl_xor(den, o, 16); num += __shfl_xor(num, o, 16); }
      p[s] = e / den;
      ex[s] = num / den;
      lse[s] = mx + logf(den);
    }
    float dlog[4] = {0.f, 0.f, 0.f, 0.f};
    if (asg >= 0) {
      const float ax = gx + 0.5f, ay = gy + 0.5f;
      const float px1 = ax - ex[0], py1 = ay - ex[1], px2 = ax + ex[2], py2 = ay + ex[3];
      // CIoU(pred, target) and its gradient w.r.t. the pred corners (alpha held constant: metrics.py:225-227)
      const float eps = 1e-7f;
      const float w1 = px2 - px1, h1 = py2 - py1 + eps, w2 = tx2 - tx1, h2 = ty2 - ty1 + eps;
      const float iwr = fminf(px2, tx2) - fmaxf(px1, tx1), ihr = fminf(py2, ty2) - fmaxf(py1, ty1);
      const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
      const float inter = iw * ih, uni = w1 * h1 + w2 * h2 - inter + eps, iou = inter / uni;
      const float cw = fmaxf(px2, tx2) - fminf(px1, tx1), ch = fmaxf(py2, ty2) - fminf(py1, ty1);
      const float cc = cw * cw + ch * ch + eps;
      const float rx = tx1 + tx2 - px1 - px2, ry = ty1 + ty2 - py1 - py2;
      const float rho2 = (rx * rx + ry * ry) * 0.25f;
      const float at1 = atanf(w1 / h1), at2 = atanf(w2 / h2);
      const float dat = at2 - at1;
      const float v = 0.40528473456935109f * dat * dat;
      const float alpha = v / (v - iou + (1.f + eps));
      const float ciou = iou - (rho2 / cc + v * alpha);
      if (!BWD) {
        if (j == 0) l_box += (1.f - ciou) * w;
      } else {
        // derivatives w.r.t. (px1, py1, px2, py2)
        const float diw[4] = {(iwr > 0.f && px1 > tx1) ? -1.f : 0.f, 0.f, (iwr > 0.f && px2 < tx2) ? 1.f : 0.f, 0.f};
        const float dih[4] = {0.f, (ihr > 0.f && py1 > ty1) ? -1.f : 0.f, 0.f, (ihr > 0.f && py2 < ty2) ? 1.f : 0.f};
        const float dw1[4] = {-1.f, 0.f, 1.f, 0.f}, dh1[4] = {0.f, -1.f, 0.f, 1.f};
        const float dcw[4] = {(px1 < tx1) ? -1.f : 0.f, 0.f, (px2 > tx2) ? 1.f : 0.f, 0.f};
        const float dch[4] = {0.f, (py1 < ty1) ? -1.f : 0.f, 0.f, (py2 > ty2) ? 1.f : 0.f};
        const float drho[4] = {-rx * 0.5f, -ry * 0.5f, -rx * 0.5f, -ry * 0.5f};
        const float inv_wh = 1.f / (w1 * w1 + h1 * h1);
        float dc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dinter = ih * diw[q] + iw * dih[q];
          const float duni = h1 * dw1[q] + w1 * dh1[q] - dinter;
          const float diou = (dinter * uni - inter * duni) / (uni * uni);
          const float dcc = 2.f * cw * dcw[q] + 2.f * ch * dch[q];
          const float dpen = (drho[q] * cc - rho2 * dcc) / (cc * cc);
          const float dat1 = (h1 * dw1[q] - w1 * dh1[q]) * inv_wh;
          const float dv = 2.f * 0.40528473456935109f * dat * (-dat1);
          dc[q] = diou - dpen - alpha * dv;
        }
        // loss_box = (1 - ciou) * w  ->  d/d corner = -w * dc;  corners = anchor -/+ expectation
        const float gb = -w * a.gain_box * up;
        const float dex[4] = {-gb * dc[0], -gb * dc[1], gb * dc[2], gb * dc[3]};
#pragma unroll
        for (int s = 0; s < 4; ++s) dlog[s] += p[s] * ((float)j - ex[s]) * dex[s];
      }
      // DFL: target distances clamped to [0, reg_max - 1 - 0.01] (tal.py:364, loss.py:77)
      const float tdist[4] = {ax - tx1, ay - ty1, tx2 - ax, ty2 - ay};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float t = fminf(fmaxf(tdist[s], 0.f), (float)(REG - 1) - 0.01f);
        const int tl = (int)t;
        const float wl = (float)(tl + 1) - t, wr = 1.f - wl;
        if (!BWD) {
          float term = 0.f;
          if (j == tl) term += wl * (lse[s] - xs[s]);
          if (j == tl + 1) term += wr * (lse[s] - xs[s]);
          l_dfl += term * 0.25f * w;
        } else {
          const float gd = 0.25f * w * a.gain_dfl * up;
          dlog[s] += gd * (p[s] * (wl + wr) - (j == tl ? wl : 0.f) - (j == tl + 1 ? wr : 0.f));
        }
      }
    }
    if (BWD) {
#pragma unroll
      for (int s = 0; s < 4; ++s) drow[s * REG + j] = dlog[s];
    }
  }
  }
  if (!BWD) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { l_cls += __shfl_xor(l_cls, o, 16); l_dfl += __shfl_xor(l_dfl, o, 16); }
    if (j == 0) { red[0][grp] = l_box; red[1][grp] = l_cls; red[2][grp] = l_dfl; }
    __syncthreads();
    if (threadIdx.x < 3) {
      float s = 0.f;
      for (int q = 0; q < 16; ++q) s += red[threadIdx.x][q];
      atomicAdd(a.sums + (long)(blockIdx.x % a.sum_slots) * 4 + 1 + threadIdx.x, s);
    }
  }
}

static int fill_args(LossArgs& a, int B, int nc, int nl, const float* const* maps, float* const* dmaps, const int* hs, const int* ws,
                     const float* strides, int G, const float* gt, float* pbox, float* align, float* overlap, int* topk, int* assign,
                     float* pos, float* norm, float* sums) {
  SY11_REQUIRE(B > 0 && nc > 0 && nl > 0 && nl <= 4 && G >= 0, "det_loss: bad dims");
  SY11_REQUIRE(maps && hs && ws && strides && pbox && assign && norm && sums, "det_loss: null pointer");
  int A = 0;
  for (int i = 0; i < nl; ++i) {
    SY11_REQUIRE(maps[i] && hs[i] > 0 && ws[i] > 0, "det_loss: bad level %d", i);
    a.maps[i] = maps[i]; a.dmaps[i] = dmaps ? dmaps[i] : nullptr; a.hs[i] = hs[i]; a.ws[i] = ws[i]; a.a0[i] = A; a.strides[i] = strides[i];
    A += hs[i] * ws[i];
  }
  SY11_REQUIRE((long)B * A < (1L << 31), "det_loss: too many anchors");
  a.B = B; a.nc = nc; a.nl = nl; a.A = A; a.G = G; a.no = 4 * REG + nc;
  a.gt = gt; a.pbox = pbox; a.align = align; a.overlap = overlap; a.topk = topk; a.assign = assign;
  a.pos_align = pos; a.pos_ov = pos ? pos + (long)B * G : nullptr; a.norm = norm; a.sums = sums; a.sum_slots = 64;
  return SY11_OK;
}

// Forward: fills the assignment workspace and sums[64][4] = {tss, box, cls, dfl} partials (caller zeroes sums and pos).
// The loss terms need tss: pass 1 (assign=true) runs K1-K4; pass 2 (assign=false) runs K5 with 1/max(tss,1).
extern "C" int sy11_det_loss_assign(int32_t B, int32_t nc, int32_t nl, const float* const* maps, const int32_t* hs, const int32_t* ws,
                                    const float* strides, int32_t G, const float* gt, float* pbox, float* align, float* overlap,
                                    int32_t* topk, int32_t* assign, float* pos, float* norm, float* sums, void* stream) {
  LossArgs a{};
  int rc = fill_args(a, B, nc, nl, maps, nullptr, hs, ws, strides, G, gt, pbox, align, overlap, topk, assign, pos, norm, sums);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const long BA = (long)B * a.A;
  hipLaunchKernelGGL(loss_decode_kernel, dim3((unsigned)((BA * 4 + 255) / 256)), dim3(256), 0, st, a);
  if (G > 0) {
    SY11_REQUIRE(gt && align && overlap && topk && pos, "det_loss: null workspace");
    const size_t lds = (size_t)a.A * 4;
    SY11_REQUIRE(lds <= 150 * 1024, "det_loss: %d anchors exceed the LDS top-k buffer", a.A);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)loss_tal_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(loss_tal_metrics_kernel, dim3(G, B), dim3(256), lds, st, a);
  }
  hipLaunchKernelGGL(loss_tal_resolve_kernel, dim3((unsigned)((BA + 255) / 256)), dim3(256), 0, st, a);
  DetPartials dp;                                   // ordered mode: the tss partials as one row per workgroup, folded in index order
  const long nb = (BA + 255) / 256;
  if (sy11_det(16)) {
    if (!dp.acquire(st, 1, nb, 4)) SY11_FAIL(SY11_ELAUNCH, "det_loss_assign: ordered-reduction workspace unavailable");
    a.sums = dp.buf(0); a.sum_slots = (int)nb;
  }
  hipLaunchKernelGGL(loss_tal_norm_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
  SY11_LAUNCH_CHECK("det_loss_assign");
  return dp.base ? dp.fold(0, sums) : SY11_OK;
}

// terms: sums[.][1..3] += box / cls / dfl partial sums (un-normalised: divide by max(tss,1) on the device afterwards)
extern "C" int sy11_det_loss_terms(int32_t B, int32_t nc, int32_t nl, const float* const* maps, const int32_t* hs, const int32_t* ws,
                                   const float* strides, int32_t G, const float* gt, const int32_t* assign, const float* norm,
                                   float* sums, void* stream) {
  LossArgs a{};
  static float dummy;
  int rc = fill_args(a, B, nc, nl, maps, nullptr, hs, ws, strides, G, gt, &dummy, nullptr, nullptr, nullptr, (int*)assign, nullptr,
                     (float*)norm, sums);
  if (rc) return rc;
  const long BA = (long)B * a.A;
  DetPartials dp;
  long nb = (BA + 15) / 16;
  if (nb > 2048) nb = 2048;                          // grid-stride (see loss_terms_kernel)
  if (sy11_det(16)) {
    if (!dp.acquire((hipStream_t)stream, 1, nb, 4)) SY11_FAIL(SY11_ELAUNCH, "det_loss_terms: ordered-reduction workspace unavailable");
    a.sums = dp.buf(0); a.sum_slots = (int)nb;
  }
  hipLaunchKernelGGL((loss_terms_kernel<false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, 1.f);
  SY11_LAUNCH_CHECK("det_loss_terms");
  return dp.base ? dp.fold(0, sums) : SY11_OK;
}

// backward: dmaps[l] = d(loss)/d(maps[l]) where loss = B * (gain_box*box + gain_cls*cls + gain_dfl*dfl) / tss, times *gscale.
// tss_inv_dev: device scalar 1/max(tss,1) is folded on the host side into `tss_inv` via a tiny device op -> passed by pointer.
extern "C" int sy11_det_loss_bwd(int32_t B, int32_t nc, int32_t nl, const float* const* maps, float* const* dmaps, const int32_t* hs,
                                 const int32_t* ws, const float* strides, int32_t G, const float* gt, const int32_t* assign,
                                 const float* norm, const float* gscale_times_tssinv, const float* second_factor, float gain_box, float gain_cls,
                                 float gain_dfl, void* stream) {
  LossArgs a{};
  static float dummy;
  SY11_REQUIRE(dmaps && gscale_times_tssinv, "det_loss_bwd: null pointer");
  int rc = fill_args(a, B, nc, nl, maps, dmaps, hs, ws, strides, G, gt, &dummy, nullptr, nullptr, nullptr, (int*)assign, nullptr,
                     (float*)norm, &dummy);
  if (rc) return rc;
  a.gscale = gscale_times_tssinv;
  a.gscale2 = second_factor;
  a.gain_box = gain_box; a.gain_cls = gain_cls; a.gain_dfl = gain_dfl;
  const long BA = (long)B * a.A;
  hipLaunchKernelGGL((loss_terms_kernel<true>), dim3((unsigned)((BA + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a, 1.f);
  SY11_LAUNCH_CHECK("det_loss_bwd");
  return SY11_OK;
}


// ---- target packing (v8DetectionLoss.preprocess, utils/loss.py:194-207): the (n) targets [image, cls, xywh normalised] of a batch ->
// gt (B, G, 5) rows [cls, x1, y1, x2, y2] in pixels, an image's targets in their original order, zero rows as padding.  One wave
// per image: a ballot over 64 targets at a time gives every match its rank.  Replaces ~12 ATen launches (cat, bincount / scatter,
// argsort, cumsum, index_put, the xywh2xyxy slices) and the per-image host loop of the reference.
__global__ __launch_bounds__(64) void loss_pack_targets_kernel(int n, int G, const float* __restrict__ idx, int idx_st, const float* __restrict__ cls,
                                                               int cls_st, const float* __restrict__ box, int box_st, float sw, float sh,
                                                               float* __restrict__ gt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* out = gt + (long)b * G * 5;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool hit = i < n && (int)idx[(long)i * idx_st] == b;
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int r = base + __popcll(m & ((1ull << lane) - 1ull));
      if (r < G) {
        const float* bx = box + (long)i * box_st;
        // out[..., 1:5] * scale, then xywh2xyxy.  Every product goes through an empty asm so the backend cannot fuse it into the
        // following add / subtract (-ffp-contract=fast fuses at instruction selection): bit-exact against the tensor-op order
        float cx = bx[0] * sw, cy = bx[1] * sh, hw = bx[2] * sw, hh = bx[3] * sh;
        asm volatile("" : "+v"(cx), "+v"(cy), "+v"(hw), "+v"(hh));
        hw *= 0.5f; hh *= 0.5f;
        float* o = out + (long)r * 5;
        o[0] = cls[(long)i * cls_st];
        o[1] = cx - hw; o[2] = cy - hh; o[3] = cx + hw; o[4] = cy + hh;
      }
    }
    base += __popcll(m);
  }
  for (int r = (base < G ? base : G) + lane; r < G; r += 64) {
#pragma unroll
    for (int q = 0; q < 5; ++q) out[(long)r * 5 + q] = 0.f;
  }
}

extern "C" int sy11_det_loss_pack_targets(int32_t n, int32_t B, int32_t G, const float* batch_idx, int32_t idx_stride, const float* cls,
                                          int32_t cls_stride, const float* bboxes, int32_t box_stride, float scale_w, float scale_h, float* gt,
                                          void* stream) {
  SY11_REQUIRE(n >= 0 && B > 0 && G >= 0 && gt, "det_loss_pack_targets: bad argument");
  SY11_REQUIRE(n == 0 || (batch_idx && cls && bboxes && idx_stride >= 1 && cls_stride >= 1 && box_stride >= 4), "det_loss_pack_targets: null / short-stride target columns");
  if (G == 0) return SY11_OK;
  hipLaunchKernelGGL(loss_pack_targets_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, n, G, batch_idx, idx_stride, cls, cls_stride, bboxes, box_stride, scale_w,
                     scale_h, gt);
  SY11_LAUNCH_CHECK("det_loss_pack_targets");
  return SY11_OK;
}

// ---- finish: sums (64, 4) partials -> out[0] = loss = B * sum_i gain_i * term_i / max(tss, 1), out[1..3] = the three gained items,
// out[4] = 1 / max(tss, 1) (the backward launch multiplies the upstream gradient by it).  The 64 slots are folded in index order.
__global__ __launch_bounds__(64) void loss_finish_kernel(const float* __restrict__ sums, int B, float gb, float gc, float gd, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 64; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] += sums[s * 4 + q];
  const float tss = fmaxf(t[0], 1.f);
  const float i0 = t[1] / tss * gb, i1 = t[2] / tss * gc, i2 = t[3] / tss * gd;
  out[1] = i0; out[2] = i1; out[3] = i2;
  out[0] = ((i0 + i1) + i2) * (float)B;
  out[4] = 1.f / tss;
}
extern "C" int sy11_det_loss_finish(const float* sums, int32_t B, float gain_box, float gain_cls, float gain_dfl, float* out, void* stream) {
  SY11_REQUIRE(sums && out && B > 0, "det_loss_finish: bad argument");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, B, gain_box, gain_cls, gain_dfl, out);
  SY11_LAUNCH_CHECK("det_loss_finish");
  return SY11_OK;
}
