// bn_tail.h — BatchNorm statistics finalisation folded into the tail of the producing convolution.
//
// Every workgroup of a conv kernel adds its per-channel sum / sum-of-squares into the [slots][C] statistic rows, fences,
// and takes a ticket; the workgroup that draws the last ticket folds the slots (f64), writes mean / rstd / scale / shift for
// the BN apply kernel and the backward pass and updates the running statistics — what sy11_bn_finalize does as a separate
// launch (81 launches of ~5 us per training step of yolo11s).  The ticket word is left at zero for the next use.
#pragma once
#include "common.h"

struct BnTailDev {
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* mean;
  float* rstd;
  float* scale;
  float* shift;
  int* ticket;          // nullptr: no tail
  float eps, momentum;
  double count;
  int C;                // channels finalised by this launch
};

static inline BnTailDev bn_tail_dev(const sy11_bn_tail* t, int C, int ch_off, int ticket_off) {
  BnTailDev d{};
  if (!t) return d;
  d.gamma = t->gamma + ch_off; d.beta = t->beta + ch_off;
  d.running_mean = t->running_mean ? t->running_mean + ch_off : nullptr;
  d.running_var = t->running_var ? t->running_var + ch_off : nullptr;
  d.mean = t->mean + ch_off; d.rstd = t->rstd + ch_off; d.scale = t->scale + ch_off; d.shift = t->shift + ch_off;
  d.ticket = t->ticket + ticket_off;
  d.eps = t->eps; d.momentum = t->momentum; d.count = t->count; d.C = C;
  return d;
}

// call with ALL threads of the workgroup, after the workgroup's statistic atomics have been issued
__device__ __forceinline__ void bn_tail_run(const BnTailDev& t, const float* ssum, const float* ssq, int slots, int stride, unsigned nblocks) {
  __shared__ int s_last;
  // NOT __threadfence(): on the multi-XCD parts an agent-scope release writes back the XCD's whole L2 (buffer_wbl2) — with
  // one fence per workgroup that doubled the training step.  Everything the last workgroup reads was produced by
  // device-scope ATOMICS (performed memory-side, coherent across XCDs) and is read back with atomic loads, so it is enough
  // that this workgroup's own atomics have been performed before it draws its ticket: wait for them, then barrier.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(t.ticket, 1) == (int)nblocks - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  for (int c = threadIdx.x; c < t.C; c += blockDim.x) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < slots; ++k) {
      s1 += (double)__hip_atomic_load(ssum + (long)k * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += (double)__hip_atomic_load(ssq + (long)k * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double mu = s1 / t.count;
    double var = s2 / t.count - mu * mu;
    if (var < 0) var = 0;
    const float r = (float)(1.0 / sqrt(var + (double)t.eps));
    t.mean[c] = (float)mu;
    t.rstd[c] = r;
    const float sc = t.gamma[c] * r;
    t.scale[c] = sc;
    t.shift[c] = t.beta[c] - (float)mu * sc;
    if (t.running_mean) {
      const double unbiased = t.count > 1 ? var * t.count / (t.count - 1) : var;
      t.running_mean[c] = (1.f - t.momentum) * t.running_mean[c] + t.momentum * (float)mu;
      t.running_var[c] = (1.f - t.momentum) * t.running_var[c] + t.momentum * (float)unbiased;
    }
  }
  if (threadIdx.x == 0) *t.ticket = 0;
}
