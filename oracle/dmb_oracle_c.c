/*
 * CPU ORACLE, part 2 -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Plain C loop-nest restatement of the index arithmetic of every primitive on the path, independent of both
 * PyTorch and the HIP kernels: it is cross-checked against oracle/dmb_oracle.py (which is pinned to the real
 * reference through tests/golden/) on small shapes by tests/test_oracle_c.py.  Accumulation is in double so that
 * it can arbitrate between FP32 implementations.  Each function cites the reference lines it follows.
 * Build: `make -C oracle` -> oracle/_build/libdmb_oracle_c.so (gcc only).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static int keep(int x, int d, int W) { /* cat_fms.py:36-44 */
  if (d > 0) return x >= d;
  if (d < 0) return x < W + d;
  return 1;
}

/* cost_processors/utils/cat_fms.py:7-48 */
void oc_cat_fms(const float* L, const float* R, float* out, int B, int C, int H, int W, int D, const int* idx) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < D; ++k)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const int d = idx[k], ok = keep(x, d, W);
            const size_t o = ((((size_t)b * 2 * C + c) * D + k) * H + y) * W + x;
            const size_t o2 = ((((size_t)b * 2 * C + C + c) * D + k) * H + y) * W + x;
            const size_t i = (((size_t)b * C + c) * H + y) * W;
            out[o] = ok ? L[i + x] : 0.f;
            out[o2] = ok ? R[i + x - d] : 0.f;
          }
}

/* cost_processors/utils/dif_fms.py:7-46 */
void oc_dif_fms(const float* L, const float* R, float* out, int B, int C, int H, int W, int D, const int* idx) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < D; ++k)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const int d = idx[k];
            const size_t i = (((size_t)b * C + c) * H + y) * W;
            out[((((size_t)b * C + c) * D + k) * H + y) * W + x] = keep(x, d, W) ? L[i + x] - R[i + x - d] : 0.f;
          }
}

/* group-wise correlation: spec SURVEY 8-a4 (no reference implementation) */
void oc_gwc_fms(const float* L, const float* R, float* out, int B, int C, int G, int H, int W, int D, const int* idx) {
  const int cg = C / G;
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < G; ++g)
      for (int k = 0; k < D; ++k)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const int d = idx[k];
            double s = 0.0;
            if (keep(x, d, W))
              for (int c = g * cg; c < (g + 1) * cg; ++c) {
                const size_t i = (((size_t)b * C + c) * H + y) * W;
                s += (double)L[i + x] * (double)R[i + x - d];
              }
            out[((((size_t)b * G + g) * D + k) * H + y) * W + x] = (float)(s / cg);
          }
}

/* nn.Conv3d k3 p1 stride s (+ per-channel scale/shift, residual, relu): basic_layers.py:68-83,160-177 */
void oc_conv3d_k3(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y,
                  int B, int Ci, int Co, int D, int H, int W, int stride, int relu) {
  const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Co; ++co)
      for (int zo = 0; zo < Do; ++zo)
        for (int yo = 0; yo < Ho; ++yo)
          for (int xo = 0; xo < Wo; ++xo) {
            double acc = 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int z = zo * stride - 1 + kz, yy = yo * stride - 1 + ky, xx = xo * stride - 1 + kx;
                    if (z < 0 || z >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    acc += (double)x[((((size_t)b * Ci + ci) * D + z) * H + yy) * W + xx] *
                           (double)w[((((size_t)co * Ci + ci) * 3 + kz) * 3 + ky) * 3 + kx];
                  }
            const size_t o = ((((size_t)b * Co + co) * Do + zo) * Ho + yo) * Wo + xo;
            double v = acc * (scale ? scale[co] : 1.0) + (shift ? shift[co] : 0.0);
            if (res) v += res[o];
            if (relu && v < 0) v = 0;
            y[o] = (float)v;
          }
}

/* nn.ConvTranspose3d k3 s2 p1 op1: y[2i - 1 + k] += x[i] * w[ci][co][k]  (hourglass.py:52-60) */
void oc_deconv3d_k3s2(const float* x, const float* w, const float* scale, const float* shift, const float* res,
                      float* y, int B, int Ci, int Co, int D, int H, int W, int relu) {
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Co; ++co)
      for (int zo = 0; zo < Do; ++zo)
        for (int yo = 0; yo < Ho; ++yo)
          for (int xo = 0; xo < Wo; ++xo) {
            double acc = 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int tz = zo + 1 - kz, ty = yo + 1 - ky, tx = xo + 1 - kx;
                    if ((tz & 1) || (ty & 1) || (tx & 1)) continue;
                    const int z = tz / 2, yy = ty / 2, xx = tx / 2;
                    if (tz < 0 || z >= D || ty < 0 || yy >= H || tx < 0 || xx >= W) continue;
                    acc += (double)x[((((size_t)b * Ci + ci) * D + z) * H + yy) * W + xx] *
                           (double)w[((((size_t)ci * Co + co) * 3 + kz) * 3 + ky) * 3 + kx];
                  }
            const size_t o = ((((size_t)b * Co + co) * Do + zo) * Ho + yo) * Wo + xo;
            double v = acc * (scale ? scale[co] : 1.0) + (shift ? shift[co] : 0.0);
            if (res) v += res[o];
            if (relu && v < 0) v = 0;
            y[o] = (float)v;
          }
}

/* nn.ConvTranspose3d(1,1,8,4,2): y[4i - 2 + k] += x[i] * w[k]  (AcfNet.py:55-57,81-83) */
void oc_deconv3d_k8s4_c1(const float* x, const float* w, float* y, int B, int D, int H, int W) {
  const int Do = 4 * D, Ho = 4 * H, Wo = 4 * W;
  for (int b = 0; b < B; ++b)
    for (int zo = 0; zo < Do; ++zo)
      for (int yo = 0; yo < Ho; ++yo)
        for (int xo = 0; xo < Wo; ++xo) {
          double acc = 0.0;
          for (int kz = 0; kz < 8; ++kz)
            for (int ky = 0; ky < 8; ++ky)
              for (int kx = 0; kx < 8; ++kx) {
                const int tz = zo + 2 - kz, ty = yo + 2 - ky, tx = xo + 2 - kx;
                if ((tz & 3) || (ty & 3) || (tx & 3) || tz < 0 || ty < 0 || tx < 0) continue;
                const int z = tz / 4, yy = ty / 4, xx = tx / 4;
                if (z >= D || yy >= H || xx >= W) continue;
                acc += (double)x[(((size_t)b * D + z) * H + yy) * W + xx] * (double)w[(kz * 8 + ky) * 8 + kx];
              }
          y[(((size_t)b * Do + zo) * Ho + yo) * Wo + xo] = (float)acc;
        }
}

/* F.interpolate(trilinear, align_corners=True) (PSMNet.py:77-93): FP32 index arithmetic as ATen's CPU kernel
 * (src = scale * dst rounded to float, lambda = src - floor(src)), interpolation itself in double. */
static void lerp_idx(int dst, int in, int out, int* i0, int* i1, float* l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  *i0 = (int)src;
  if (*i0 > in - 1) *i0 = in - 1;
  *i1 = *i0 + ((*i0 < in - 1) ? 1 : 0);
  float l = src - (float)*i0;
  *l1 = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
}
void oc_trilinear_ac(const float* x, float* y, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  for (int b = 0; b < B; ++b)
    for (int zo = 0; zo < Do; ++zo)
      for (int yo = 0; yo < Ho; ++yo)
        for (int xo = 0; xo < Wo; ++xo) {
          int z0, z1, y0, y1, x0, x1;
          float lz, ly, lx;
          lerp_idx(zo, Di, Do, &z0, &z1, &lz);
          lerp_idx(yo, Hi, Ho, &y0, &y1, &ly);
          lerp_idx(xo, Wi, Wo, &x0, &x1, &lx);
          const float* p = x + (size_t)b * Di * Hi * Wi;
#define V(z, yy, xx) ((double)p[((size_t)(z) * Hi + (yy)) * Wi + (xx)])
          const double a00 = V(z0, y0, x0) * (1.f - lx) + V(z0, y0, x1) * lx, a01 = V(z0, y1, x0) * (1.f - lx) + V(z0, y1, x1) * lx;
          const double a10 = V(z1, y0, x0) * (1.f - lx) + V(z1, y0, x1) * lx, a11 = V(z1, y1, x0) * (1.f - lx) + V(z1, y1, x1) * lx;
#undef V
          const double h0 = a00 * (1.f - ly) + a01 * ly, h1 = a10 * (1.f - ly) + a11 * ly;
          y[(((size_t)b * Do + zo) * Ho + yo) * Wo + xo] = (float)(h0 * (1.f - lz) + h1 * lz);
        }
}

/* SoftArgmin (soft_argmin.py:45-75) in double */
void oc_soft_argmin(const float* cost, float* disp, int B, int D, int H, int W, float alpha, int normalize,
                    const float* samples) {
  const size_t HW = (size_t)H * W;
  for (int b = 0; b < B; ++b)
    for (size_t p = 0; p < HW; ++p) {
      const float* c = cost + (size_t)b * D * HW + p;
      double m = -INFINITY, s = 0.0, t = 0.0;
      if (normalize) {
        for (int k = 0; k < D; ++k) m = fmax(m, (double)(c[k * HW] * alpha));
        for (int k = 0; k < D; ++k) {
          const double e = exp((double)(c[k * HW] * alpha) - m);
          s += e;
          t += e * samples[k];
        }
        disp[b * HW + p] = (float)(t / s);
      } else {
        for (int k = 0; k < D; ++k) t += (double)(c[k * HW] * alpha) * samples[k];
        disp[b * HW + p] = (float)t;
      }
    }
}

/* LocalSoftArgmin (local_soft_argmin.py:48-105); argidx = first maximal index */
void oc_local_soft_argmin(const float* cost, float* disp, int64_t* argidx, int B, int D, int H, int W, int radius,
                          int rdil, int start, int dil, float alpha) {
  const size_t HW = (size_t)H * W;
  for (int b = 0; b < B; ++b)
    for (size_t p = 0; p < HW; ++p) {
      const float* c = cost + (size_t)b * D * HW + p;
      int bi = 0;
      for (int k = 1; k < D; ++k)
        if (c[k * HW] > c[bi * HW]) bi = k;
      if (argidx) argidx[b * HW + p] = bi;
      double mx = -INFINITY, s = 0.0, t = 0.0;
      for (int pass = 0; pass < 2; ++pass)
        for (int i = -radius; i <= radius; ++i) {
          const int raw = bi + i * rdil;
          const float mask = (raw >= 0 && raw <= D - 1) ? 1.f : 0.f;
          const int ci = raw < 0 ? 0 : (raw > D - 1 ? D - 1 : raw);
          const float l = c[ci * HW] * alpha * mask + (1.f - mask) * (-10000.0f * alpha);
          if (pass == 0) {
            mx = fmax(mx, (double)l);
          } else {
            const double e = exp((double)l - mx);
            s += e;
            t += e * ((double)start + (double)ci * dil);
          }
        }
      disp[b * HW + p] = (float)(t / s);
    }
}

/* pixel_error.py:6-73 for one image after eval.py:12-31 cropping; out = {epe, 1px, 2px, 3px, 5px} */
void oc_calc_error(const float* est, const float* gt, int Hp, int Wp, int H0, int W0, float lb, float ub, double* out) {
  const int crop = Hp - H0 >= 0, top = crop ? Hp - H0 : 0, cols = crop ? (W0 < Wp ? W0 : Wp) : Wp;
  double sum = 0, n = 0, c[4] = {0, 0, 0, 0};
  const float th[4] = {1, 2, 3, 5};
  for (int y = top; y < Hp; ++y)
    for (int x = 0; x < cols; ++x) {
      const float g = gt[(size_t)y * Wp + x];
      if (!(g > lb && g < ub)) continue;
      const float a = fabsf(g - est[(size_t)y * Wp + x]);
      sum += a;
      n += 1;
      for (int k = 0; k < 4; ++k) c[k] += a > th[k];
    }
  out[0] = n >= 1 ? sum / n : 0;
  for (int k = 0; k < 4; ++k) out[1 + k] = n >= 1 ? 100.0 * c[k] / n : 0;
}
