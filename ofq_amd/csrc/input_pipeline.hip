// On-device input pipeline: what the reference's loader does to a decoded, cropped uint8 batch
// before the model sees it, in ONE pass over the batch (1-2 B read, 4 B written per element):
//   1. mixup / cutmix in uint8 space -- timm 0.5.4 FastCollateMixup._mix_batch_collate (train.py:590: the collate of the
//      prefetching loader): sample b is mixed with sample B-1-b; mixup: rint(lam * x + (1 - lam) * x') as float32 then
//      truncated to uint8; cutmix: the box [yl,yh) x [xl,xh) is copied from the partner
//   2. normalisation -- timm PrefetchLoader: x.float().sub_(mean * 255).div_(std * 255), per channel, fp32
//   3. random erasing -- timm RandomErasing(mode='pixel') inside PrefetchLoader: one rectangle per erased sample is
//      overwritten with standard-normal noise (supplied by the caller so that the host, the device and the oracle can
//      share one stream of numbers)
// The random DECISIONS (lam, cutmix or mixup, the box, which samples are erased and where) are host-side in timm (numpy /
// python `random`) and stay host-side here (ofq_amd/data.py); this kernel applies them.
#include "common.h"

struct InputPipeArgs {
  const unsigned char* in; float* out; const int* rects; const float* noise;
  float mean[4], std[4];
  float lam, oml;          // lam and 1 - lam as float32 (numpy rounds the float64 scalars to the array's dtype)
  int use_mix, use_cutmix, yl, yh, xl, xh;
  int B, C, H, W;
};

__global__ __launch_bounds__(256) void input_pipeline_kernel(InputPipeArgs a) {
  const int64_t plane = (int64_t)a.H * a.W;
  const int64_t per_img = plane * a.C;
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;         // W % 4 == 0: four pixels of one row
  if (i4 >= per_img * a.B) return;
  const int b = (int)(i4 / per_img);
  const int64_t r = i4 - (int64_t)b * per_img;
  const int c = (int)(r / plane);
  const int64_t p = r - (int64_t)c * plane;
  const int y = (int)(p / a.W), x0 = (int)(p - (int64_t)y * a.W);
  const uchar4 own = *reinterpret_cast<const uchar4*>(a.in + i4);
  unsigned char v[4] = {own.x, own.y, own.z, own.w};
  if (a.use_mix) {
    const int64_t j4 = (int64_t)(a.B - 1 - b) * per_img + r;
    const uchar4 oth = *reinterpret_cast<const uchar4*>(a.in + j4);
    const unsigned char o[4] = {oth.x, oth.y, oth.z, oth.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.use_cutmix) {
        if (y >= a.yl && y < a.yh && x0 + e >= a.xl && x0 + e < a.xh) v[e] = o[e];
      } else {
        // numpy: mixed = x.astype(float32) * lam + x'.astype(float32) * (1 - lam); rint; astype(uint8)
        const float m = __fadd_rn(__fmul_rn((float)v[e], a.lam), __fmul_rn((float)o[e], a.oml));
        v[e] = (unsigned char)(int)rintf(m);
      }
    }
  }
  float out[4];
  const float mean = a.mean[c], sd = a.std[c];
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = __fdiv_rn(__fsub_rn((float)v[e], mean), sd);
  if (a.rects) {
    const int top = a.rects[4 * b], left = a.rects[4 * b + 1], h = a.rects[4 * b + 2], w = a.rects[4 * b + 3];
    if (h > 0 && y >= top && y < top + h) {
      const float4 nz = *reinterpret_cast<const float4*>(a.noise + i4);
      const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x0 + e >= left && x0 + e < left + w) out[e] = nn[e];
    }
  }
  *reinterpret_cast<float4*>(a.out + i4) = make_float4(out[0], out[1], out[2], out[3]);
}

extern "C" int ofq_input_pipeline_u8(const uint8_t* in, float* out, int64_t B, int64_t C, int64_t H, int64_t W, const float* mean255,
                                     const float* std255, int use_mix, int use_cutmix, float lam, float one_minus_lam, int yl, int yh, int xl, int xh,
                                     const int32_t* rects, const float* noise, ofq_stream_t stream) {
  if (!in || !out || !mean255 || !std255 || B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0 || (W & 3)) return OFQ_EINVAL;
  if ((((uintptr_t)in) & 3) || (((uintptr_t)out) & 15) || (rects && (!noise || (((uintptr_t)noise) & 15)))) return OFQ_EINVAL;
  if (use_mix && (B & 1)) return OFQ_EINVAL;                     // timm: "Batch size should be even when using this"
  InputPipeArgs a = {};
  a.in = in; a.out = out; a.rects = rects; a.noise = noise;
  for (int c = 0; c < (int)C; ++c) { a.mean[c] = mean255[c]; a.std[c] = std255[c]; }     // HOST arrays
  a.lam = lam; a.oml = one_minus_lam; a.use_mix = use_mix; a.use_cutmix = use_cutmix; a.yl = yl; a.yh = yh; a.xl = xl; a.xh = xh;
  a.B = (int)B; a.C = (int)C; a.H = (int)H; a.W = (int)W;
  const int64_t n4 = B * C * H * W / 4;
  hipLaunchKernelGGL(input_pipeline_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
