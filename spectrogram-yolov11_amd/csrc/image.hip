// image.hip — the image side of preprocess: uint8 -> float/255, the multi-scale bilinear resize of preprocess_batch,
// and LetterBox (resize + 114 border + BGR->RGB + HWC->CHW + /255) in one pass.  All three are pure HBM streaming work:
// one thread per output pixel (all channels), x fastest so the plane writes coalesce; the source taps of neighbouring
// threads fall in the same or the next cache line.
//
// Reference behaviour restated (cited per kernel):
//   models/yolo/detect/train.py:57-74   preprocess_batch: .float() / 255, F.interpolate(bilinear, align_corners=False)
//   data/augment.py:1477-1600           LetterBox.__call__: cv2.resize(INTER_LINEAR) + cv2.copyMakeBorder(114)
//   engine/predictor.py:118-136         preprocess: BGR->RGB, HWC->CHW, .half()/.float(), /= 255
// cv2 itself is a third-party dependency of the reference (opencv-python >= 4.6, not vendored, not installed here); the
// resize follows OpenCV's published 8-bit INTER_LINEAR algorithm (imgproc/src/resize.cpp: 11-bit coefficients, the
// ((b*(S>>4))>>16 ... +2)>>2 vertical pass, and the exact-2x INTER_AREA shortcut).
#include "common.h"
#include <string.h>

namespace {

template <typename T> struct Out { static __device__ __forceinline__ T cvt(float v) { return ElemTraits<T>::from_f(v); } };

// ---------------------------------------------------------------------------------------------- u8 -> float / 255
template <typename T> struct alignas(4 * sizeof(T)) Quad { T v[4]; };

template <typename T>
__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, long n) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;          // 4 bytes in, one 8/16-byte store out per thread
  if (i0 + 4 <= n) {
    const unsigned w = *(const unsigned*)(x + i0);
    Quad<T> q;
#pragma unroll
    for (int j = 0; j < 4; ++j) q.v[j] = Out<T>::cvt(__fdiv_rn((float)((w >> (8 * j)) & 255u), 255.0f));
    *(Quad<T>*)(y + i0) = q;
  } else {
    for (long i = i0; i < n; ++i) y[i] = Out<T>::cvt(__fdiv_rn((float)x[i], 255.0f));
  }
}

// ---------------------------------------------------------------------------------------------- ATen bilinear (align_corners=False)
// src = scale*(dst+0.5)-0.5 clamped at 0, i1 = i0 + (i0 < in-1), val = h0*(w0*a + w1*b) + h1*(w0*c + w1*d): the
// arithmetic of aten/native/UpSample.h area_pixel_compute_source_index + the upsample_bilinear2d kernels.
template <typename TI> __device__ __forceinline__ float load_px(const TI* p) { return ElemTraits<TI>::to_f(*p); }
template <> __device__ __forceinline__ float load_px<uint8_t>(const uint8_t* p) { return __fdiv_rn((float)*p, 255.0f); }

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const TI* __restrict__ x, TO* __restrict__ y, int planes, int IH,
                                                              int IW, int OH, int OW, float sh, float sw) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (ox >= OW) return;
  float fy = sh * ((float)oy + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  float fx = sw * ((float)ox + 0.5f) - 0.5f;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int yp = y0 < IH - 1 ? 1 : 0, xp = x0 < IW - 1 ? 1 : 0;
  const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
  for (int p = blockIdx.z; p < planes; p += gridDim.z) {
    const TI* s = x + ((long)p * IH + y0) * IW + x0;
    const float a = load_px(s), b = load_px(s + xp), c = load_px(s + (long)yp * IW), d = load_px(s + (long)yp * IW + xp);
    y[((long)p * OH + oy) * OW + ox] = Out<TO>::cvt(h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d));
  }
}

// ---------------------------------------------------------------------------------------------- LetterBox
struct LbArgs {
  const uint8_t* src;  // (sh, sw, 3) HWC
  void* dst;           // (H, W, 3) u8 HWC or (3, H, W) planes
  int sh, sw, H, W, nh, nw, top, left, fill, reverse_c, chw, area2;
  double scale_x, scale_y;
};

template <bool PIN>
__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  // resize.cpp: fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx.  Columns past either border pin
  // fx = 0 on the clamped column; rows do NOT: they keep their fractional weights and only the row indices are clipped
  // (so a border row is blended with itself, which truncates differently from a single full-weight tap).
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (PIN) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  s0 = s < 0 ? 0 : (s < ssize ? s : ssize - 1);
  s1 = s + 1 < 0 ? 0 : (s + 1 < ssize ? s + 1 : ssize - 1);
  // saturate_cast<short>(cbuf[k] * INTER_RESIZE_COEF_SCALE): round-half-even of an exact float product
  a0 = __float2int_rn((1.f - f) * 2048.f);
  a1 = __float2int_rn(f * 2048.f);
}

template <typename T>
__global__ __launch_bounds__(256) void letterbox_kernel(const LbArgs a) {
  const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
  if (X >= a.W) return;
  int v[3] = {a.fill, a.fill, a.fill};
  const int dx = X - a.left, dy = Y - a.top;
  if ((unsigned)dx < (unsigned)a.nw && (unsigned)dy < (unsigned)a.nh) {
    if (a.nw == a.sw && a.nh == a.sh) {                       // LetterBox skips cv2.resize when the size already matches
      const uint8_t* s = a.src + ((long)dy * a.sw + dx) * 3;
      v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
    } else if (a.area2) {                                      // exact 2x shrink: cv2 swaps INTER_LINEAR for the 2x2 box mean
      const uint8_t* s = a.src + ((long)(2 * dy) * a.sw + 2 * dx) * 3;
      const uint8_t* n = s + (long)a.sw * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (s[c] + s[c + 3] + n[c] + n[c + 3] + 2) >> 2;
    } else {
      int x0, x1, ax0, ax1, y0, y1, by0, by1;
      lin_coef<true>(dx, a.scale_x, a.sw, x0, x1, ax0, ax1);
      lin_coef<false>(dy, a.scale_y, a.sh, y0, y1, by0, by1);
      const uint8_t* r0 = a.src + (long)y0 * a.sw * 3;
      const uint8_t* r1 = a.src + (long)y1 * a.sw * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;       // HResizeLinear: 11-bit products, no shift
        const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
        const int r = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;   // VResizeLinear 8u
        v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
      }
    }
  }
  if (a.reverse_c) { const int t = v[0]; v[0] = v[2]; v[2] = t; }
  if constexpr (sizeof(T) == 1) {
    uint8_t* d = (uint8_t*)a.dst;
    if (a.chw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d[((long)c * a.H + Y) * a.W + X] = (uint8_t)v[c];
    } else {
      uint8_t* p = d + ((long)Y * a.W + X) * 3;
      p[0] = (uint8_t)v[0]; p[1] = (uint8_t)v[1]; p[2] = (uint8_t)v[2];
    }
  } else {
    T* d = (T*)a.dst;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T o = Out<T>::cvt(__fdiv_rn((float)v[c], 255.0f));
      if (a.chw) d[((long)c * a.H + Y) * a.W + X] = o;
      else d[((long)Y * a.W + X) * 3 + c] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------- mosaic + affine + HSV + flip
// One launch renders a training sample: the (virtual) mosaic canvas is never materialised — every bilinear tap of
// cv2.warpAffine looks up which tile covers that canvas pixel — then RandomHSV's BGR->HSV->LUT->BGR and the two flips are
// applied in registers and the pixel is written once, already in the layout / dtype the batch tensor wants.
//   Mosaic._mosaic4         data/augment.py:660-715     tiles pasted on a 2s x 2s canvas filled with 114
//   RandomPerspective       data/augment.py:1000-1078   cv2.warpAffine(img, M[:2], dsize, borderValue=(114,114,114))
//   RandomHSV               data/augment.py:1303-1390   cv2.cvtColor(BGR2HSV) + 3 LUTs + cv2.cvtColor(HSV2BGR)
//   RandomFlip              data/augment.py:1393-1474   np.flipud / np.fliplr
// warpAffine follows OpenCV 4.x imgwarp.cpp's fixed-point path: source coordinates in 1/1024 px from
// rint(M*x*1024) tables, +16, >>5 -> 5 fractional bits that index a bilinear table of exact integer weights
// (32-fy)(32-fx)*32 ... (sum 32768), result (sum + 16384) >> 15, every tap outside the source = borderValue.
// cvtColor follows color_hsv.simd.hpp: integer RGB2HSV_b with the 12-bit division tables; float HSV2RGB_b.
struct WarpTile {
  const uint8_t* src;
  int h, w, x1, y1, x2, y2, padw, padh;
};
struct WarpArgs {
  WarpTile tile[4];
  double m[6];
  void* dst;
  int n_tiles, canvas_h, canvas_w, has_warp, H, W, has_hsv, flip_ud, flip_lr, fill, reverse_c, chw;
  unsigned lut[192];   // 3 x 256 bytes: hue, sat, val
};

__device__ __forceinline__ double nofma(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float nofma(float x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ void canvas_fetch(const WarpArgs& a, int cx, int cy, int* v) {
  v[0] = v[1] = v[2] = a.fill;
  if ((unsigned)cx >= (unsigned)a.canvas_w || (unsigned)cy >= (unsigned)a.canvas_h) return;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < a.n_tiles) {
      const WarpTile& k = a.tile[t];
      if (cx >= k.x1 && cx < k.x2 && cy >= k.y1 && cy < k.y2) {
        const uint8_t* p = k.src + ((long)(cy - k.padh) * k.w + (cx - k.padw)) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mosaic_warp_kernel(const WarpArgs a) {
  __shared__ unsigned lut32[192];
  if (a.has_hsv) {
    if (threadIdx.x < 192) lut32[threadIdx.x] = a.lut[threadIdx.x];
    __syncthreads();
  }
  const uint8_t* lut = (const uint8_t*)lut32;
  const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
  if (X >= a.W) return;
  const int wx = a.flip_lr ? a.W - 1 - X : X, wy = a.flip_ud ? a.H - 1 - Y : Y;
  int v[3];
  if (a.has_warp) {
    const int adelta = __double2int_rn(a.m[0] * (double)wx * 1024.0), bdelta = __double2int_rn(a.m[3] * (double)wx * 1024.0);
    const int X0 = __double2int_rn((nofma(a.m[1] * (double)wy) + a.m[2]) * 1024.0) + 16;
    const int Y0 = __double2int_rn((nofma(a.m[4] * (double)wy) + a.m[5]) * 1024.0) + 16;
    const int Xf = (X0 + adelta) >> 5, Yf = (Y0 + bdelta) >> 5;
    int sx = Xf >> 5, sy = Yf >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);      // saturate_cast<short>
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = Xf & 31, fy = Yf & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    int t00[3], t01[3], t10[3], t11[3];
    canvas_fetch(a, sx, sy, t00);
    canvas_fetch(a, sx + 1, sy, t01);
    canvas_fetch(a, sx, sy + 1, t10);
    canvas_fetch(a, sx + 1, sy + 1, t11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r = (t00[c] * w00 + t01[c] * w01 + t10[c] * w10 + t11[c] * w11 + 16384) >> 15;
      v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
    }
  } else {
    canvas_fetch(a, wx, wy, v);
  }
  if (a.has_hsv) {
    const int b = v[0], g = v[1], r = v[2];
    const int vmax = max(max(b, g), r), vmin = min(min(b, g), r), diff = vmax - vmin;
    // hdiv_table180[i] = cvRound((180 << 12) / (6. * i)); sdiv_table[i] = cvRound((255 << 12) / (1. * i)); [0] = 0
    const int sdiv = vmax ? __double2int_rn(1044480.0 / (double)vmax) : 0;
    const int hdiv = diff ? __double2int_rn(737280.0 / (6.0 * (double)diff)) : 0;
    int s = (diff * sdiv + 2048) >> 12;
    int h = vmax == r ? g - b : (vmax == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv + 2048) >> 12;
    h += h < 0 ? 180 : 0;
    h = h > 255 ? 255 : h;
    const int h2 = lut[h], s2 = lut[256 + (s & 255)], v2 = lut[512 + vmax];
    // HSV2RGB_b: floats, hscale = 6/180
    const float fs = (float)s2 * (1.f / 255.f), fv = (float)v2 * (1.f / 255.f);
    float fb = fv, fg = fv, fr = fv;
    if (s2 != 0) {
      float hh = nofma((float)h2 * (6.f / 180.f));        // keep the product rounded: hh - sector must not become an fma
      int sector = (int)floorf(hh);
      hh -= (float)sector;
      if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
      const float t1 = fv * (1.f - fs);
      const float t2 = fv * (1.f - nofma(fs * hh));
      const float t3 = fv * (1.f - nofma(fs * (1.f - hh)));
      // sector_data = {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} -> (b, g, r) picks from tab = {v, t1, t2, t3}
      switch (sector) {
        case 0: fb = t1; fg = t3; fr = fv; break;
        case 1: fb = t1; fg = fv; fr = t2; break;
        case 2: fb = t3; fg = fv; fr = t1; break;
        case 3: fb = fv; fg = t2; fr = t1; break;
        case 4: fb = fv; fg = t1; fr = t3; break;
        default: fb = t2; fg = t1; fr = fv; break;
      }
    }
    const int ob = __float2int_rn(fb * 255.f), og = __float2int_rn(fg * 255.f), orr = __float2int_rn(fr * 255.f);
    v[0] = ob < 0 ? 0 : (ob > 255 ? 255 : ob);
    v[1] = og < 0 ? 0 : (og > 255 ? 255 : og);
    v[2] = orr < 0 ? 0 : (orr > 255 ? 255 : orr);
  }
  if (a.reverse_c) { const int t = v[0]; v[0] = v[2]; v[2] = t; }
  if constexpr (sizeof(T) == 1) {
    uint8_t* d = (uint8_t*)a.dst;
    if (a.chw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d[((long)c * a.H + Y) * a.W + X] = (uint8_t)v[c];
    } else {
      uint8_t* p = d + ((long)Y * a.W + X) * 3;
      p[0] = (uint8_t)v[0]; p[1] = (uint8_t)v[1]; p[2] = (uint8_t)v[2];
    }
  } else {
    T* d = (T*)a.dst;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T o = Out<T>::cvt(__fdiv_rn((float)v[c], 255.0f));
      if (a.chw) d[((long)c * a.H + Y) * a.W + X] = o;
      else d[((long)Y * a.W + X) * 3 + c] = o;
    }
  }
}

}  // namespace

extern "C" int sy11_image_u8_to_float(int32_t dtype, int64_t n, const uint8_t* x, void* y, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && n >= 0, "image_u8_to_float: bad dtype %d or negative size", dtype);
  if (n == 0) return SY11_OK;
  SY11_REQUIRE(x && y, "image_u8_to_float: null buffer");
  SY11_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 15) == 0, "image_u8_to_float: x must be 4-byte and y 16-byte aligned");
  const dim3 grid(cdiv(n, 256 * 4));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SY11_F32) u8_to_float_kernel<float><<<grid, 256, 0, s>>>(x, (float*)y, n);
  else if (dtype == SY11_F16) u8_to_float_kernel<_Float16><<<grid, 256, 0, s>>>(x, (_Float16*)y, n);
  else u8_to_float_kernel<__bf16><<<grid, 256, 0, s>>>(x, (__bf16*)y, n);
  SY11_LAUNCH_CHECK("image_u8_to_float");
  return SY11_OK;
}

template <typename TI>
static int resize_dispatch(int y_dtype, const void* x, void* y, int planes, int IH, int IW, int OH, int OW, hipStream_t s) {
  // area_pixel_compute_scale<float>(in, out, align_corners=false, scales=nullopt) = (float)in / out
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  const dim3 grid(cdiv(OW, 256), OH, planes < 64 ? planes : 64);
  if (y_dtype == SY11_F32) resize_bilinear_kernel<TI, float><<<grid, 256, 0, s>>>((const TI*)x, (float*)y, planes, IH, IW, OH, OW, sh, sw);
  else if (y_dtype == SY11_F16) resize_bilinear_kernel<TI, _Float16><<<grid, 256, 0, s>>>((const TI*)x, (_Float16*)y, planes, IH, IW, OH, OW, sh, sw);
  else resize_bilinear_kernel<TI, __bf16><<<grid, 256, 0, s>>>((const TI*)x, (__bf16*)y, planes, IH, IW, OH, OW, sh, sw);
  SY11_LAUNCH_CHECK("image_resize_bilinear");
  return SY11_OK;
}

extern "C" int sy11_image_resize_bilinear(int32_t x_dtype, int32_t y_dtype, int32_t planes, int32_t IH, int32_t IW, int32_t OH,
                                          int32_t OW, const void* x, void* y, void* stream) {
  SY11_REQUIRE((dtype_ok(x_dtype) || x_dtype == SY11_U8) && dtype_ok(y_dtype), "image_resize_bilinear: bad dtype");
  SY11_REQUIRE(planes > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && OH <= 65535 && x && y, "image_resize_bilinear: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == SY11_U8) return resize_dispatch<uint8_t>(y_dtype, x, y, planes, IH, IW, OH, OW, s);
  if (x_dtype == SY11_F32) return resize_dispatch<float>(y_dtype, x, y, planes, IH, IW, OH, OW, s);
  if (x_dtype == SY11_F16) return resize_dispatch<_Float16>(y_dtype, x, y, planes, IH, IW, OH, OW, s);
  return resize_dispatch<__bf16>(y_dtype, x, y, planes, IH, IW, OH, OW, s);
}

extern "C" int sy11_image_letterbox(int32_t dtype, int32_t sh, int32_t sw, int32_t H, int32_t W, int32_t new_h, int32_t new_w,
                                    int32_t top, int32_t left, int32_t fill, int32_t reverse_c, int32_t chw,
                                    const uint8_t* src, void* dst, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) || dtype == SY11_U8, "image_letterbox: bad dtype %d", dtype);
  SY11_REQUIRE(sh > 0 && sw > 0 && H > 0 && W > 0 && new_h > 0 && new_w > 0 && H <= 65535 && src && dst, "image_letterbox: bad shape");
  SY11_REQUIRE(top >= 0 && left >= 0 && top + new_h <= H && left + new_w <= W,
               "image_letterbox: resized image (%d x %d at %d,%d) does not fit the %d x %d canvas", new_h, new_w, top, left, H, W);
  SY11_REQUIRE(fill >= 0 && fill <= 255, "image_letterbox: fill must be a byte");
  LbArgs a;
  a.src = src; a.dst = dst; a.sh = sh; a.sw = sw; a.H = H; a.W = W; a.nh = new_h; a.nw = new_w; a.top = top; a.left = left;
  a.fill = fill; a.reverse_c = reverse_c; a.chw = chw;
  // resize.cpp: inv_scale = dsize/ssize (double); scale = 1./inv_scale
  const double inv_x = (double)new_w / sw, inv_y = (double)new_h / sh;
  a.scale_x = 1. / inv_x;
  a.scale_y = 1. / inv_y;
  a.area2 = (sw == 2 * new_w && sh == 2 * new_h) ? 1 : 0;
  const dim3 grid(cdiv(W, 256), H);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SY11_U8) letterbox_kernel<uint8_t><<<grid, 256, 0, s>>>(a);
  else if (dtype == SY11_F32) letterbox_kernel<float><<<grid, 256, 0, s>>>(a);
  else if (dtype == SY11_F16) letterbox_kernel<_Float16><<<grid, 256, 0, s>>>(a);
  else letterbox_kernel<__bf16><<<grid, 256, 0, s>>>(a);
  SY11_LAUNCH_CHECK("image_letterbox");
  return SY11_OK;
}

extern "C" int sy11_image_mosaic_warp(int32_t dtype, int32_t n_tiles, const uint8_t* const* tile_src, const int32_t* tile_geom,
                                      int32_t canvas_h, int32_t canvas_w, const double* minv, int32_t H, int32_t W,
                                      const uint8_t* hsv_lut, int32_t flip_ud, int32_t flip_lr, int32_t fill,
                                      int32_t reverse_c, int32_t chw, void* dst, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) || dtype == SY11_U8, "image_mosaic_warp: bad dtype %d", dtype);
  SY11_REQUIRE(n_tiles >= 0 && n_tiles <= 4 && (n_tiles == 0 || (tile_src && tile_geom)), "image_mosaic_warp: 0..4 tiles");
  SY11_REQUIRE(canvas_h > 0 && canvas_w > 0 && H > 0 && W > 0 && H <= 65535 && dst, "image_mosaic_warp: bad shape");
  SY11_REQUIRE(minv || (H == canvas_h && W == canvas_w), "image_mosaic_warp: without a warp the output is the canvas (%d x %d)", canvas_h, canvas_w);
  SY11_REQUIRE(fill >= 0 && fill <= 255, "image_mosaic_warp: fill must be a byte");
  WarpArgs a;
  for (int t = 0; t < 4; ++t) a.tile[t] = WarpTile{nullptr, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < n_tiles; ++t) {
    const int32_t* g = tile_geom + 8 * t;
    WarpTile k{tile_src[t], g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]};
    SY11_REQUIRE(k.src && k.h > 0 && k.w > 0, "image_mosaic_warp: tile %d has no pixels", t);
    SY11_REQUIRE(k.x1 >= 0 && k.y1 >= 0 && k.x2 <= canvas_w && k.y2 <= canvas_h && k.x1 <= k.x2 && k.y1 <= k.y2,
                 "image_mosaic_warp: tile %d region [%d,%d)x[%d,%d) leaves the %d x %d canvas", t, k.x1, k.x2, k.y1, k.y2, canvas_w, canvas_h);
    SY11_REQUIRE(k.x1 - k.padw >= 0 && k.x2 - k.padw <= k.w && k.y1 - k.padh >= 0 && k.y2 - k.padh <= k.h,
                 "image_mosaic_warp: tile %d region reads outside its %d x %d source", t, k.h, k.w);
    a.tile[t] = k;
  }
  a.n_tiles = n_tiles; a.canvas_h = canvas_h; a.canvas_w = canvas_w; a.H = H; a.W = W; a.dst = dst;
  a.has_warp = minv ? 1 : 0;
  for (int i = 0; i < 6; ++i) a.m[i] = minv ? minv[i] : 0.0;
  a.has_hsv = hsv_lut ? 1 : 0;
  if (hsv_lut) memcpy(a.lut, hsv_lut, 768);
  else memset(a.lut, 0, 768);
  a.flip_ud = flip_ud; a.flip_lr = flip_lr; a.fill = fill; a.reverse_c = reverse_c; a.chw = chw;
  const dim3 grid(cdiv(W, 256), H);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SY11_U8) mosaic_warp_kernel<uint8_t><<<grid, 256, 0, s>>>(a);
  else if (dtype == SY11_F32) mosaic_warp_kernel<float><<<grid, 256, 0, s>>>(a);
  else if (dtype == SY11_F16) mosaic_warp_kernel<_Float16><<<grid, 256, 0, s>>>(a);
  else mosaic_warp_kernel<__bf16><<<grid, 256, 0, s>>>(a);
  SY11_LAUNCH_CHECK("image_mosaic_warp");
  return SY11_OK;
}
