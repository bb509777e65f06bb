// The thin torch extension over the C ABI (north_star: "exposed to Python through a thin C-ABI torch extension").
//
// The reference's native-op pattern is a torch.utils.cpp_extension module whose functions take at::Tensor, check them and call the
// kernel launcher (dmb/ops/spn/setup.py:5-16, dmb/ops/spn/src/gaterecurrent2dnoind_cuda.cpp:20-89).  This translation unit is that
// shim for libdmb_hip.so: per entry point it unwraps the tensors (device, dtype, contiguity, the shapes the kernel will index with),
// takes the caller's CURRENT HIP stream from c10 (c10::hip::getCurrentHIPStream), allocates the output with the caching allocator
// when the caller passes none, calls the C-ABI function of include/dmb_hip.h and raises on a non-zero code.  No arithmetic, no
// kernels, no state: the product is the C ABI; this file only removes the interpreter from the launch path (measured,
// profiles/r06_binding_overhead.log: 12.6 us of host time per ops.conv3d_k3 call through ctypes -- 4.2 us of it the launch itself
// -- which made one 256x512 pair host-bound at 0.87 ms per step once its kernels took 0.93 ms).  The ctypes binding (_lib.py)
// stays: it binds EVERY entry point of the same library, this shim the ones on the per-pair launch path.
//
// Built by build.py::build_torch_shim with the host compiler (no device code): lib/_dmb_torch_shim.so, linked against
// lib/libdmb_hip.so ($ORIGIN rpath) and libtorch / libc10_hip.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <string>
#include <tuple>
#include <vector>

#include "dmb_hip.h"

namespace {

PyObject* g_error_class = nullptr;   // densematchingbenchmark_amd._lib.DmbLibraryError (set once by _lib.load_shim)

[[noreturn]] void raise(const std::string& msg) {
  PyErr_SetString(g_error_class ? g_error_class : PyExc_RuntimeError, msg.c_str());
  throw py::error_already_set();
}

void chk(int rc, const char* what) {
  if (rc != DMB_OK) raise(std::string(what) + " failed with code " + std::to_string(rc) + " (" + dmb_last_error() + ")");
}

// A contiguous FP32 tensor on the CURRENT HIP device (kernels launch on the calling thread's current device): anything else is
// refused -- there is no CPU path behind this boundary.
const at::Tensor& f32(const at::Tensor& t, const char* name) {
  if (!t.is_cuda())
    raise(std::string(name) + " lives on " + t.device().str() + ": the dmb HIP path only runs on a GPU (cuda/hip) tensor and has no CPU fallback");
  if (t.scalar_type() != at::kFloat) raise(std::string(name) + " must be float32");
  if (!t.is_contiguous()) raise(std::string(name) + " must be contiguous");
  if (t.get_device() != c10::hip::current_device())
    raise(std::string(name) + " lives on cuda:" + std::to_string(t.get_device()) + " but the current device is cuda:" + std::to_string(c10::hip::current_device()));
  return t;
}
const float* ptr(const at::Tensor& t, const char* name) { return f32(t, name).data_ptr<float>(); }
const float* optr(const c10::optional<at::Tensor>& t, const char* name) { return t.has_value() ? ptr(*t, name) : nullptr; }
void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void affine_ok(const c10::optional<at::Tensor>& s, const c10::optional<at::Tensor>& h, int64_t Co, const char* what) {
  if ((s.has_value() && s->numel() != Co) || (h.has_value() && h->numel() != Co))
    raise(std::string(what) + ": scale / shift must have one element per output channel");
}

at::Tensor out_or_new(const c10::optional<at::Tensor>& out, at::IntArrayRef shape, const at::Tensor& like, const char* what) {
  if (!out.has_value()) return at::empty(shape, like.options());
  if (out->sizes() != shape || out->device() != like.device()) raise(std::string(what) + ": out has the wrong shape or device");
  f32(*out, "out");
  return *out;
}

// ---- dmb_conv3d_k3_f32 (layers/basic_layers.py:68-100; hourglass.py:62-86) ------------------------------------------------------
at::Tensor conv3d_k3(const at::Tensor& x, const at::Tensor& wpack, int64_t Co, const c10::optional<at::Tensor>& scale,
                     const c10::optional<at::Tensor>& shift, const c10::optional<at::Tensor>& residual, int64_t stride, int64_t relu_flags,
                     const c10::optional<at::Tensor>& out) {
  f32(x, "x");
  if (x.dim() != 5) raise("conv3d_k3: x must be [B, Ci, D, H, W]");
  const int64_t B = x.size(0), Ci = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4);
  if (stride != 1 && stride != 2) raise("conv3d_k3: stride must be 1 or 2");
  const int64_t Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  at::Tensor y = out_or_new(out, {B, Co, Do, Ho, Wo}, x, "conv3d_k3");
  if (residual.has_value() && residual->sizes() != y.sizes()) raise("conv3d_k3: residual shape != output shape");
  affine_ok(scale, shift, Co, "conv3d_k3");
  if (wpack.numel() != dmb_conv3d_packed_floats((int)Co, (int)Ci)) raise("conv3d_k3: packed weights do not belong to these channel counts");
  chk(dmb_conv3d_k3_f32(ptr(x, "x"), ptr(wpack, "wpack"), optr(scale, "scale"), optr(shift, "shift"), optr(residual, "residual"),
                        y.data_ptr<float>(), (int)B, (int)Ci, (int)Co, (int)D, (int)H, (int)W, (int)stride, (int)relu_flags, stream_of(x)),
      "dmb_conv3d_k3_f32");
  return y;
}

// ---- dmb_deconv3d_k3s2_f32 (layers/basic_layers.py:160-177; hourglass.py:52-60,84-86) -------------------------------------------
at::Tensor deconv3d_k3s2(const at::Tensor& x, const at::Tensor& wpack, int64_t Co, const c10::optional<at::Tensor>& scale,
                         const c10::optional<at::Tensor>& shift, const c10::optional<at::Tensor>& residual, int64_t relu_flags,
                         const c10::optional<at::Tensor>& workspace, const c10::optional<at::Tensor>& out, int64_t Wout) {
  f32(x, "x");
  if (x.dim() != 5) raise("deconv3d_k3s2: x must be [B, Ci, D, H, W]");
  const int64_t B = x.size(0), Ci = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4);
  if (Wout <= 0) Wout = 2 * W;
  at::Tensor y = out_or_new(out, {B, Co, 2 * D, 2 * H, Wout}, x, "deconv3d_k3s2");
  if (residual.has_value() && residual->sizes() != y.sizes()) raise("deconv3d_k3s2: residual shape != output shape");
  affine_ok(scale, shift, Co, "deconv3d_k3s2");
  if (wpack.numel() != dmb_deconv3d_packed_floats((int)Ci, (int)Co)) raise("deconv3d_k3s2: packed weights do not belong to these channel counts");
  void* ws = nullptr;
  if (workspace.has_value()) {
    const at::Tensor& w = *workspace;
    if (!w.is_cuda() || w.scalar_type() != at::kInt || !w.is_contiguous() || w.numel() * 4 < DMB_DECONV3D_WORKSPACE_BYTES || w.device() != x.device())
      raise("deconv3d_k3s2: workspace must be a contiguous int32 tensor of DMB_DECONV3D_WORKSPACE_BYTES on the input's device");
    ws = w.data_ptr();
  }
  chk(dmb_deconv3d_k3s2_f32(ptr(x, "x"), ptr(wpack, "wpack"), optr(scale, "scale"), optr(shift, "shift"), optr(residual, "residual"),
                            y.data_ptr<float>(), (int)B, (int)Ci, (int)Co, (int)D, (int)H, (int)W, (int)Wout, (int)relu_flags, ws, stream_of(x)),
      "dmb_deconv3d_k3s2_f32");
  return y;
}

// ---- dmb_conv3d_k3_c1_f32 (aggregators/PSMNet.py:46-54,70-72) -------------------------------------------------------------------
at::Tensor conv3d_k3_c1(const at::Tensor& x, const at::Tensor& w, double bias, const c10::optional<at::Tensor>& residual, int64_t flags) {
  f32(x, "x");
  if (x.dim() != 5) raise("conv3d_k3_c1: x must be [B, Ci, D, H, W]");
  const int64_t B = x.size(0), Ci = x.size(1), D = x.size(2), H = x.size(3), W = x.size(4);
  if (w.numel() != Ci * 27) raise("conv3d_k3_c1: weight does not belong to this channel count");
  at::Tensor y = at::empty({B, 1, D, H, W}, x.options());
  if (residual.has_value() && residual->sizes() != y.sizes()) raise("conv3d_k3_c1 residual: shapes differ");
  chk(dmb_conv3d_k3_c1_f32(ptr(x, "x"), ptr(w, "weight"), (float)bias, optr(residual, "residual"), y.data_ptr<float>(), (int)B, (int)Ci,
                           (int)D, (int)H, (int)W, (int)flags, stream_of(x)),
      "dmb_conv3d_k3_c1_f32");
  return y;
}

// ---- dmb_trilinear_ac_soft_argmin_f32 (aggregators/PSMNet.py:74-93 + disp_predictors/faster_soft_argmin.py:46-71) -----------------
std::vector<at::Tensor> trilinear_ac_soft_argmin(const at::Tensor& x, int64_t Do, int64_t Ho, int64_t Wo, double alpha,
                                                 const std::vector<float>& disp_values) {
  f32(x, "x");
  if (x.dim() != 4) raise("trilinear_ac_soft_argmin: x must be [B, D, H, W]");
  if ((int64_t)disp_values.size() != Do) raise("trilinear_ac_soft_argmin: one disparity sample value per output plane");
  const int64_t B = x.size(0);
  at::Tensor y = at::empty({B, Do, Ho, Wo}, x.options()), disp = at::empty({B, 1, Ho, Wo}, x.options());
  chk(dmb_trilinear_ac_soft_argmin_f32(ptr(x, "x"), y.data_ptr<float>(), disp.data_ptr<float>(), (int)B, (int)x.size(1), (int)x.size(2),
                                       (int)x.size(3), (int)Do, (int)Ho, (int)Wo, (float)alpha, disp_values.data(), stream_of(x)),
      "dmb_trilinear_ac_soft_argmin_f32");
  return {y, disp};
}

// ---- dmb_conv2d_f32 (layers/basic_layers.py:12-66; backbones/PSMNet.py:8-129) ---------------------------------------------------
// x [B, Cx, H, W], channels [coff, coff + Ci) read; out [B, Ctot, Ho, Wo] written at channel out_off; residual read at res_off.
at::Tensor conv2d(const at::Tensor& x, int64_t coff, int64_t Ci, const at::Tensor& wpack, int64_t Co, int64_t ksize, int64_t stride,
                  int64_t dilation, const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
                  const c10::optional<at::Tensor>& residual, int64_t res_off, int64_t relu_flags, const c10::optional<at::Tensor>& out,
                  int64_t out_off) {
  f32(x, "x");
  if (x.dim() != 4) raise("conv2d: x must be [B, C, H, W]");
  const int64_t B = x.size(0), Cx = x.size(1), H = x.size(2), W = x.size(3);
  if (coff < 0 || Ci <= 0 || coff + Ci > Cx) raise("conv2d: input channel window outside the tensor");
  if (stride < 1) raise("conv2d: stride");
  const int64_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  at::Tensor y;
  if (out.has_value()) {
    y = f32(*out, "out");
    if (y.dim() != 4 || y.size(0) != B || y.size(2) != Ho || y.size(3) != Wo || out_off < 0 || out_off + Co > y.size(1) || y.device() != x.device())
      raise("conv2d: output tensor does not fit");
  } else {
    y = at::empty({B, Co, Ho, Wo}, x.options());
    out_off = 0;
  }
  int64_t Cres = 0;
  const float* rp = nullptr;
  if (residual.has_value()) {
    const at::Tensor& r = f32(*residual, "residual");
    if (r.dim() != 4 || r.size(0) != B || r.size(2) != Ho || r.size(3) != Wo || res_off < 0 || res_off + Co > r.size(1)) raise("conv2d: residual shape");
    Cres = r.size(1);
    rp = r.data_ptr<float>() + res_off * Ho * Wo;
  }
  affine_ok(scale, shift, Co, "conv2d");
  if (wpack.numel() != dmb_conv2d_packed_floats((int)Co, (int)Ci, (int)ksize)) raise("conv2d: packed weights do not belong to this layer");
  chk(dmb_conv2d_f32(x.data_ptr<float>() + coff * H * W, ptr(wpack, "wpack"), optr(scale, "scale"), optr(shift, "shift"), rp,
                     y.data_ptr<float>() + out_off * Ho * Wo, (int)B, (int)Ci, (int)Co, (int)H, (int)W, (int)ksize, (int)stride, (int)dilation,
                     (int)relu_flags, (int)Cx, (int)y.size(1), (int)Cres, stream_of(x)),
      "dmb_conv2d_f32");
  return y;
}

// ---- dmb_copy_window_f32 (zero-extended / cropped rows of the volume-free first layer, csrc/catconv.hip) --------------------------
at::Tensor copy_window(const at::Tensor& src, int64_t Wd, int64_t xs) {
  f32(src, "src");
  if (src.dim() < 1 || Wd <= 0) raise("copy_window: bad argument");
  const int64_t W = src.size(-1);
  std::vector<int64_t> shape(src.sizes().begin(), src.sizes().end());
  shape.back() = Wd;
  at::Tensor dst = at::empty(shape, src.options());
  chk(dmb_copy_window_f32(src.data_ptr<float>(), dst.data_ptr<float>(), (long long)(src.numel() / W), (int)W, (int)Wd, (int)xs, stream_of(src)),
      "dmb_copy_window_f32");
  return dst;
}

// ---- the per-unit launches of a training step (round 6: the backbone's 145 units make images -> loss host-bound) ---------------
// [B, C, *spatial] -> (B, C, S)
void bcs(const at::Tensor& t, const char* what, int64_t& B, int64_t& C, int64_t& S) {
  if (t.dim() < 3) raise(std::string(what) + ": [B, C, spatial ...] expected");
  B = t.size(0);
  C = t.size(1);
  S = B * C > 0 ? t.numel() / (B * C) : 0;
  if (B <= 0 || C <= 0 || S <= 0) raise(std::string(what) + ": empty tensor");
}
float* mptr(const c10::optional<at::Tensor>& t, const char* name) { return t.has_value() ? const_cast<float*>(ptr(*t, name)) : nullptr; }

// dmb_bn_train_fwd_f32 (layers/basic_layers.py:68-83 under train()): -> (y, stats [4, C] = mean, invstd, scale, shift)
std::tuple<at::Tensor, at::Tensor> bn_train_fwd(const at::Tensor& c, const c10::optional<at::Tensor>& gamma, const c10::optional<at::Tensor>& beta,
                                                const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                                                const c10::optional<at::Tensor>& num_batches_tracked, double momentum, double eps,
                                                const c10::optional<at::Tensor>& residual, int64_t relu) {
  f32(c, "c");
  int64_t B, C, S;
  bcs(c, "bn_train_fwd", B, C, S);
  if (residual.has_value() && residual->sizes() != c.sizes()) raise("bn_train_fwd: residual shape != input shape");
  for (const auto* t : {&gamma, &beta, &running_mean, &running_var})
    if (t->has_value() && (*t)->numel() != C) raise("bn_train_fwd: per-channel tensors must have C elements");
  long long* nbt = nullptr;
  if (num_batches_tracked.has_value()) {
    const at::Tensor& n = *num_batches_tracked;
    if (!n.is_cuda() || n.scalar_type() != at::kLong || n.numel() != 1 || n.device() != c.device()) raise("bn_train_fwd: num_batches_tracked must be an int64 scalar on the input's device");
    nbt = reinterpret_cast<long long*>(n.data_ptr<int64_t>());
  }
  at::Tensor y = at::empty_like(c);
  at::Tensor stats = at::empty({4, C}, c.options());
  at::Tensor ws = at::empty({dmb_bn_workspace_doubles((int)C, (long long)S)}, c.options().dtype(at::kDouble));
  float* st = stats.data_ptr<float>();
  chk(dmb_bn_train_fwd_f32(c.data_ptr<float>(), optr(gamma, "gamma"), optr(beta, "beta"), mptr(running_mean, "running_mean"),
                           mptr(running_var, "running_var"), nbt, (float)momentum, (float)eps, st, st + C, st + 2 * C, st + 3 * C,
                           optr(residual, "residual"), y.data_ptr<float>(), ws.data_ptr<double>(), (int)B, (int)C, (long long)S, (int)relu,
                           stream_of(c)),
      "dmb_bn_train_fwd_f32");
  // the kernel wrote these through their pointers: tell torch (the modules' folded-parameter caches key on the versions)
  for (const auto* t : {&running_mean, &running_var, &num_batches_tracked})
    if (t->has_value()) (*t)->unsafeGetTensorImpl()->bump_version();
  return std::make_tuple(y, stats);
}

// dmb_bn_act_bwd_f32: -> (dc, gb [2, C] = dgamma, dbeta, dres or an undefined tensor)
std::tuple<at::Tensor, at::Tensor, c10::optional<at::Tensor>> bn_act_bwd(const at::Tensor& dy, const at::Tensor& c, const c10::optional<at::Tensor>& y,
                                                                        const at::Tensor& scale, const at::Tensor& shift, const at::Tensor& mean,
                                                                        const at::Tensor& invstd, int64_t relu, bool training, bool want_dres,
                                                                        const c10::optional<at::Tensor>& dres_acc) {
  f32(dy, "dy");
  f32(c, "c");
  int64_t B, C, S;
  bcs(c, "bn_act_bwd", B, C, S);
  if (dy.sizes() != c.sizes()) raise("bn_act_bwd: dy shape != c shape");
  if (relu == 1 && !y.has_value()) raise("bn_act_bwd: the unit's output is needed for a ReLU after the skip add");
  if (y.has_value() && y->sizes() != c.sizes()) raise("bn_act_bwd: y shape != c shape");
  if (dres_acc.has_value() && dres_acc->sizes() != c.sizes()) raise("bn_act_bwd: dres_acc shape != c shape");
  for (const at::Tensor* t : {&scale, &shift, &mean, &invstd})
    if (t->numel() != C) raise("bn_act_bwd: per-channel tensors must have C elements");
  at::Tensor dc = at::empty_like(c);
  c10::optional<at::Tensor> dres;
  if (want_dres || dres_acc.has_value()) dres = at::empty_like(c);
  at::Tensor gb = at::empty({2, C}, c.options());
  at::Tensor ws = at::empty({dmb_bn_workspace_doubles((int)C, (long long)S)}, c.options().dtype(at::kDouble));
  chk(dmb_bn_act_bwd_f32(dy.data_ptr<float>(), c.data_ptr<float>(), relu == 1 ? ptr(*y, "y") : nullptr, ptr(scale, "scale"), ptr(shift, "shift"),
                         ptr(mean, "mean"), ptr(invstd, "invstd"), ws.data_ptr<double>(), gb.data_ptr<float>(), gb.data_ptr<float>() + C,
                         dc.data_ptr<float>(), dres.has_value() ? dres->data_ptr<float>() : nullptr, optr(dres_acc, "dres_acc"), (int)B, (int)C,
                         (long long)S, (int)relu, training ? 1 : 0, stream_of(c)),
      "dmb_bn_act_bwd_f32");
  return std::make_tuple(dc, gb, dres);
}

// dmb_conv2d_wgrad_f32 (autograd of the 2-D units w.r.t. their weights): x [B, Ci, H, W], dc [B, Co, H, W], W % 4 == 0
at::Tensor conv2d_wgrad(const at::Tensor& x, const at::Tensor& dc, int64_t ksize, int64_t dilation) {
  f32(x, "x");
  f32(dc, "dc");
  if (x.dim() != 4 || dc.dim() != 4 || dc.size(0) != x.size(0) || dc.size(2) != x.size(2) || dc.size(3) != x.size(3))
    raise("conv2d_wgrad: dc does not match x");
  const int64_t B = x.size(0), Ci = x.size(1), H = x.size(2), W = x.size(3), Co = dc.size(1);
  at::Tensor dw = at::empty({Co, Ci, ksize, ksize}, x.options());
  at::Tensor ws = at::empty({dmb_conv2d_wgrad_workspace_floats((int)Co, (int)Ci)}, x.options());
  chk(dmb_conv2d_wgrad_f32(x.data_ptr<float>(), dc.data_ptr<float>(), dw.data_ptr<float>(), ws.data_ptr<float>(), (int)B, (int)Ci, (int)Co, (int)H,
                           (int)W, (int)ksize, (int)dilation, stream_of(x)),
      "dmb_conv2d_wgrad_f32");
  return dw;
}

void set_error_class(py::object cls) {
  Py_XDECREF(g_error_class);
  g_error_class = cls.ptr();
  Py_XINCREF(g_error_class);
}

}  // namespace

#ifndef DMB_SHIM_ID
#define DMB_SHIM_ID "unknown"
#endif

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "thin torch extension over libdmb_hip.so's C ABI (include/dmb_hip.h): unwrap tensors, current HIP stream, raise on error";
  m.def("set_error_class", &set_error_class);
  m.def("build_id", []() { return std::string(DMB_SHIM_ID); });
  m.def("library_build_id", []() { return std::string(dmb_build_id()); });
  m.def("abi_version", []() { return dmb_abi_version(); });
  m.def("conv3d_k3", &conv3d_k3);
  m.def("deconv3d_k3s2", &deconv3d_k3s2);
  m.def("conv3d_k3_c1", &conv3d_k3_c1);
  m.def("trilinear_ac_soft_argmin", &trilinear_ac_soft_argmin);
  m.def("conv2d", &conv2d);
  m.def("copy_window", &copy_window);
  m.def("bn_train_fwd", &bn_train_fwd);
  m.def("bn_act_bwd", &bn_act_bwd);
  m.def("conv2d_wgrad", &conv2d_wgrad);
}
