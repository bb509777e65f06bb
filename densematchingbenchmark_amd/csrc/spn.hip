// Spatial propagation scan (the reference's only native op): dmb/ops/spn/src/gaterecurrent2dnoind_kernel.cu, used by
// disp_refinement/AnyNet.py:54 as GateRecurrent2dnoind(horizontal=True, reverse=False).  (backlog.)
//
//   H[s, t] = (1 - g1 - g2 - g3) * X[s, t] + g1 * H[s', t - 1] + g2 * H[s', t] + g3 * H[s', t + 1]
//
// s runs along the scan axis (columns when `horizontal`, rows otherwise), s' = s - 1 (s + 1 when `reverse`) is the position
// scanned just before, t is the transverse coordinate; g_k = G_k[s, t] where the neighbour it multiplies lies inside the
// image and 0 where it does not (gaterecurrent2dnoind_kernel.cu:10-98: the gate of a link is stored at the link's LATER
// position, i.e. at [s, t] itself, and get_gate_sf returns 0 for links that leave the image), so the first scanned line is a
// copy of X.  FP32, operation for operation as forward_one_col_left_right (:130-166) with contraction off.
//
// The reference launches one kernel per scanned line (width or height launches of N*C*T threads each: :535-552).  Here one
// launch does the whole scan: a workgroup owns one (n, c) plane, a thread one transverse position (two when the line is longer
// than 1024), the previous line travels through a double-buffered LDS row (one barrier per line), and the next line's four
// operands are in flight while the current one is computed.  The backward scan (:288-345, run in the opposite direction) is
// the same structure with the next line's output gradient and gates in LDS.
#include "dmb_common.h"

namespace dmb {

constexpr int SPN_MAXT = 2046;   // longest transverse line: two positions per thread, and the backward's 8 LDS rows of T + 2 floats within 64 KiB -- ONE limit for both directions, so that a forward that succeeds under autograd can always run its backward

#pragma clang fp contract(off)
template <int TPT>
__global__ __launch_bounds__(1024) void spn_fwd_kernel(const float* __restrict__ X, const float* __restrict__ G1,
                                                       const float* __restrict__ G2, const float* __restrict__ G3,
                                                       float* __restrict__ Hout, int S, int T, long long ss, long long ts,
                                                       long long plane, int reverse) {
  extern __shared__ float row[];   // [2][T + 2]: the previous line with a zero on either side
  const size_t base = (size_t)blockIdx.x * plane;
  const int NT = blockDim.x;
  for (int i = threadIdx.x; i < 2 * (T + 2); i += NT) row[i] = 0.f;
  __syncthreads();
  float x[TPT], g1[TPT], g2[TPT], g3[TPT];
  auto fetch = [&](int s) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = threadIdx.x + k * NT;
      if (t < T) {
        const size_t o = base + (size_t)s * ss + (size_t)t * ts;
        x[k] = X[o];
        g1[k] = G1[o];
        g2[k] = G2[o];
        g3[k] = G3[o];
      }
    }
  };
  fetch(reverse ? S - 1 : 0);
  for (int i = 0; i < S; ++i) {
    const int s = reverse ? S - 1 - i : i;
    const float* prev = row + (i & 1) * (T + 2);
    float* cur = row + ((i + 1) & 1) * (T + 2);
    float xv[TPT], a1[TPT], a2[TPT], a3[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      xv[k] = x[k];
      a1[k] = g1[k];
      a2[k] = g2[k];
      a3[k] = g3[k];
    }
    if (i + 1 < S) fetch(reverse ? s - 1 : s + 1);   // lands while this line is computed
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = threadIdx.x + k * NT;
      if (t < T) {
        const bool has_prev = i > 0;
        const float q1 = (has_prev && t - 1 >= 0) ? a1[k] : 0.f;
        const float q2 = has_prev ? a2[k] : 0.f;
        const float q3 = (has_prev && t + 1 < T) ? a3[k] : 0.f;
        const float h1 = q1 * prev[t];          // prev[t + 1 - 1]
        const float h2 = q2 * prev[t + 1];
        const float h3 = q3 * prev[t + 2];
        const float h_hype = ((h1 + h2) + h3);
        const float x_hype = ((((1.f - q1) - q2) - q3) * xv[k]);
        const float h = (x_hype + h_hype);
        Hout[base + (size_t)s * ss + (size_t)t * ts] = h;
        cur[t + 1] = h;
      }
    }
    __syncthreads();
  }
}

// Backward: scanned in the OPPOSITE direction.  hd = d loss / d H[s, t] including what flows back from the line scanned after
// it (gaterecurrent2dnoind_kernel.cu:304-319); dX = (1 - g1 - g2 - g3) * hd; dG_k = hd * (H[s', t + dk] - X[s, t]) where the
// link exists, 0 elsewhere (:324-344; the reference leaves those entries at their zero initialisation).
#pragma clang fp contract(off)
template <int TPT>
__global__ __launch_bounds__(1024) void spn_bwd_kernel(const float* __restrict__ X, const float* __restrict__ G1,
                                                       const float* __restrict__ G2, const float* __restrict__ G3,
                                                       const float* __restrict__ Hf, const float* __restrict__ dH,
                                                       float* __restrict__ dX, float* __restrict__ dG1, float* __restrict__ dG2,
                                                       float* __restrict__ dG3, int S, int T, long long ss, long long ts,
                                                       long long plane, int reverse) {
  extern __shared__ float row[];   // [2][4][T + 2]: hd and the three gates of the line scanned AFTER the current one (forward order)
  const size_t base = (size_t)blockIdx.x * plane;
  const int NT = blockDim.x, P = T + 2;
  for (int i = threadIdx.x; i < 8 * P; i += NT) row[i] = 0.f;
  __syncthreads();
  for (int i = 0; i < S; ++i) {
    const int s = reverse ? i : S - 1 - i;            // forward order was the opposite
    const int sp = reverse ? s + 1 : s - 1;           // the line scanned before s in the forward pass
    const bool has_prev = sp >= 0 && sp < S;
    const float* nx = row + (i & 1) * 4 * P;          // next-in-forward line: [hd | g1 | g2 | g3]
    float* cu = row + ((i + 1) & 1) * 4 * P;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = threadIdx.x + k * NT;
      if (t < T) {
        const size_t o = base + (size_t)s * ss + (size_t)t * ts;
        const float x = X[o], r1 = G1[o], r2 = G2[o], r3 = G3[o];
        float hd = dH[o];
        // the line after s used H[s, t] through its gates g3 (at t - 1), g2 (at t), g1 (at t + 1); a link that leaves the image
        // has gate 0 there; beyond the last line everything in `nx` is 0
        const float a3 = (t - 1 >= 0) ? nx[P * 3 + t] : 0.f, a2 = nx[P * 2 + t + 1], a1 = (t + 1 < T) ? nx[P * 1 + t + 2] : 0.f;
        hd = (((hd + (nx[t] * a3)) + (nx[t + 1] * a2)) + (nx[t + 2] * a1));
        const float q1 = (has_prev && t - 1 >= 0) ? r1 : 0.f;
        const float q2 = has_prev ? r2 : 0.f;
        const float q3 = (has_prev && t + 1 < T) ? r3 : 0.f;
        dX[o] = ((((1.f - q1) - q2) - q3) * hd);
        const size_t op = base + (size_t)(has_prev ? sp : s) * ss + (size_t)t * ts;
        dG1[o] = (has_prev && t - 1 >= 0) ? (hd * (Hf[op - ts] - x)) : 0.f;
        dG2[o] = has_prev ? (hd * (Hf[op] - x)) : 0.f;
        dG3[o] = (has_prev && t + 1 < T) ? (hd * (Hf[op + ts] - x)) : 0.f;
        cu[t + 1] = hd;
        cu[P + t + 1] = q1;      // gates as the line scanned before sees them are the raw ones where the link exists
        cu[2 * P + t + 1] = q2;
        cu[3 * P + t + 1] = q3;
      }
    }
    __syncthreads();
  }
}

static int spn_dims(int N, int C, int H, int W, int horizontal, int& S, int& T, long long& ss, long long& ts) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(DMB_EINVAL, "spn: bad argument");
  S = horizontal ? W : H;
  T = horizontal ? H : W;
  ss = horizontal ? 1 : W;
  ts = horizontal ? W : 1;
  if (T > SPN_MAXT) return fail(DMB_EUNSUPPORTED, "spn: the line across the scan direction is limited to 2046 positions");
  if ((long long)N * C > 0x7fffffffLL) return fail(DMB_EUNSUPPORTED, "spn: too many planes");
  return DMB_OK;
}

}  // namespace dmb

using namespace dmb;

extern "C" int dmb_spn_gaterecurrent2d_f32(const float* X, const float* G1, const float* G2, const float* G3, float* Hout, int N,
                                           int C, int H, int W, int horizontal, int reverse, void* stream) {
  if (!X || !G1 || !G2 || !G3 || !Hout) return fail(DMB_EINVAL, "spn: bad argument");
  int S, T;
  long long ss, ts;
  if (int e = spn_dims(N, C, H, W, horizontal, S, T, ss, ts)) return e;
  const int tpt = T > 1024 ? 2 : 1;
  const int nt = cdiv(cdiv(T, tpt), 64) * 64;
  const size_t lds = (size_t)2 * (T + 2) * sizeof(float);
  const long long plane = (long long)H * W;
  if (tpt == 1)
    hipLaunchKernelGGL((spn_fwd_kernel<1>), dim3((unsigned)(N * C)), dim3(nt), lds, (hipStream_t)stream, X, G1, G2, G3, Hout, S, T, ss, ts, plane, reverse);
  else
    hipLaunchKernelGGL((spn_fwd_kernel<2>), dim3((unsigned)(N * C)), dim3(nt), lds, (hipStream_t)stream, X, G1, G2, G3, Hout, S, T, ss, ts, plane, reverse);
  return launch_status("spn forward launch failed");
}

extern "C" int dmb_spn_gaterecurrent2d_bwd_f32(const float* X, const float* G1, const float* G2, const float* G3, const float* Hfwd,
                                               const float* dH, float* dX, float* dG1, float* dG2, float* dG3, int N, int C, int H,
                                               int W, int horizontal, int reverse, void* stream) {
  if (!X || !G1 || !G2 || !G3 || !Hfwd || !dH || !dX || !dG1 || !dG2 || !dG3) return fail(DMB_EINVAL, "spn_bwd: bad argument");
  int S, T;
  long long ss, ts;
  if (int e = spn_dims(N, C, H, W, horizontal, S, T, ss, ts)) return e;
  const int tpt = T > 1024 ? 2 : 1;
  const int nt = cdiv(cdiv(T, tpt), 64) * 64;
  const size_t lds = (size_t)8 * (T + 2) * sizeof(float);
  if (lds > 64 * 1024) return fail(DMB_EUNSUPPORTED, "spn_bwd: line too long");
  const long long plane = (long long)H * W;
  if (tpt == 1)
    hipLaunchKernelGGL((spn_bwd_kernel<1>), dim3((unsigned)(N * C)), dim3(nt), lds, (hipStream_t)stream, X, G1, G2, G3, Hfwd, dH, dX, dG1, dG2, dG3, S, T, ss, ts, plane, reverse);
  else
    hipLaunchKernelGGL((spn_bwd_kernel<2>), dim3((unsigned)(N * C)), dim3(nt), lds, (hipStream_t)stream, X, G1, G2, G3, Hfwd, dH, dX, dG1, dG2, dG3, S, T, ss, ts, plane, reverse);
  return launch_status("spn backward launch failed");
}
