// The reference's compiled operator module, rebuilt on the C ABI of include/boxattn.h.
//
// BoxeR's Functions import a pybind11 module `e2edet.ops` with four functions
// (e2edet/module/ops/src/vision.cpp:7-12; signatures box_attn/box_attn.h:29-83,
// instance_attn/instance_attn.h:32-92).  This file is that module for MI355X: the same four
// names, argument lists and return values on at::Tensor, and nothing inside but argument checks
// (CHECK_INPUT of box_attn.cu:9-11,24-28; the batch / im2col_step assertion of box_attn.cu:42)
// and pointer marshalling into libboxattn_hip.so -- the stub a maintainer of the reference
// would compile instead of its src/ directory.  Host code only: no
// kernels here, built with the host compiler against torch's headers
// (`python setup.py build_ext --inplace`, or __graft_entry__.build()).
//
// float32 / float64 as in the reference; bfloat16 `value` (with float32 locations / weights) is
// the storage mode this library adds.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "boxattn.h"

namespace e2edet {
namespace {

struct Dims {
    int B, S, H, C, L, Lq, P;
};

void check_input(const at::Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor: Not implemented on the CPU");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

void check_chunks(int64_t batch, int im2col_step)
{
    const int64_t step = std::min<int64_t>(batch, im2col_step);
    if (batch > 0)
        TORCH_CHECK(step > 0 && batch % step == 0, "batch(", batch, ") must divide im2col_step(",
                    step, ")");
}

// The reference's CHECK_INPUT block plus the shape relations its kernels silently assume.
Dims prepare(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi,
             const at::Tensor &loc, std::initializer_list<const at::Tensor *> weights,
             int im2col_step)
{
    check_input(value, "value");
    check_input(shapes, "spatial_shapes");
    check_input(lsi, "level_start_index");
    check_input(loc, "sampling_loc");
    for (const at::Tensor *w : weights) check_input(*w, "attn_weight");
    const auto vt = value.scalar_type();
    TORCH_CHECK(vt == at::kFloat || vt == at::kDouble || vt == at::kBFloat16,
                "box_attn: unsupported dtype (float32, float64, bfloat16)");
    TORCH_CHECK(shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong,
                "spatial_shapes / level_start_index must be int64");
    TORCH_CHECK(value.dim() == 4 && loc.dim() == 6 && loc.size(5) == 2,
                "expected value (B,S,H,C) and sampling_loc (B,Lq,H,L,P,2)");
    Dims d;
    d.B = (int)value.size(0); d.S = (int)value.size(1); d.H = (int)value.size(2);
    d.C = (int)value.size(3); d.L = (int)shapes.size(0);
    d.Lq = (int)loc.size(1); d.P = (int)loc.size(4);
    TORCH_CHECK(loc.size(0) == d.B && loc.size(2) == d.H && loc.size(3) == d.L &&
                    lsi.numel() == d.L,
                "sampling_loc / spatial_shapes do not match value");
    const auto wt = vt == at::kBFloat16 ? at::kFloat : vt;     // bf16 storage: fp32 geometry
    TORCH_CHECK(loc.scalar_type() == wt, "sampling_loc must be ",
                vt == at::kBFloat16 ? "float32 for bfloat16 value" : "of value's dtype");
    const int64_t n_w = (int64_t)d.B * d.Lq * d.H * d.L * d.P;
    for (const at::Tensor *w : weights) {
        TORCH_CHECK(w->numel() == n_w, "attention weights must have B*Lq*H*L*P elements");
        TORCH_CHECK(w->scalar_type() == wt, "attention weights must have sampling_loc's dtype");
    }
    check_chunks(d.B, im2col_step);
    return d;
}

void check_rc(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed with hipError ", rc);
}

void *current_stream(const at::Tensor &t)
{
    return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

// Host copies of the two level tables (a few int64 each); the workspace entry points want them.
struct HostTables {
    at::Tensor shapes, lsi;
    HostTables(const at::Tensor &s, const at::Tensor &l)
        : shapes(s.cpu().contiguous()), lsi(l.cpu().contiguous()) {}
    const int64_t *sh() const { return shapes.data_ptr<int64_t>(); }
    const int64_t *ls() const { return lsi.data_ptr<int64_t>(); }
};

at::Tensor workspace(const at::Tensor &value, const Dims &d, const HostTables &h)
{
    const size_t bytes = boxattn_bwd_workspace_bytes(value.scalar_type() == at::kBFloat16, d.B, d.S,
                                                     d.H, d.C, d.L, d.Lq, d.P, h.sh(), h.ls());
    return at::empty({(int64_t)std::max<size_t>(bytes, 256)}, value.options().dtype(at::kByte));
}

const uint16_t *bf(const at::Tensor &t) { return (const uint16_t *)t.data_ptr(); }
uint16_t *bf(at::Tensor &t) { return (uint16_t *)t.data_ptr(); }

}  // namespace

// box_attn.h:29-54 -> output (B, Lq, H*C)
at::Tensor box_attn_forward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                            const at::Tensor &level_start_index, const at::Tensor &sampling_loc,
                            const at::Tensor &attn_weight, const int im2col_step)
{
    const Dims d = prepare(value, spatial_shapes, level_start_index, sampling_loc, {&attn_weight},
                           im2col_step);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor out = at::empty({d.B, d.Lq, (int64_t)d.H * d.C}, value.options());
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    void *st = current_stream(value);
    int rc;
    if (value.scalar_type() == at::kDouble) {
        rc = boxattn_fwd_f64(value.data_ptr<double>(), sh, ls, sampling_loc.data_ptr<double>(),
                             attn_weight.data_ptr<double>(), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                             out.data_ptr<double>(), st);
    } else {
        const HostTables h(spatial_shapes, level_start_index);
        if (value.scalar_type() == at::kFloat)
            rc = boxattn_fwd_hl_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                    attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq,
                                    d.P, out.data_ptr<float>(), h.sh(), h.ls(), st);
        else
            rc = boxattn_fwd_hl_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                                     attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq,
                                     d.P, bf(out), h.sh(), h.ls(), st);
    }
    check_rc(rc, "boxattn_fwd");
    return out;
}

// box_attn.h:56-83 -> {grad_value, grad_sampling_loc, grad_attn_weight}
std::vector<at::Tensor> box_attn_backward(const at::Tensor &value, const at::Tensor &spatial_shapes,
                                          const at::Tensor &level_start_index,
                                          const at::Tensor &sampling_loc,
                                          const at::Tensor &attn_weight,
                                          const at::Tensor &grad_output, const int im2col_step)
{
    const Dims d = prepare(value, spatial_shapes, level_start_index, sampling_loc, {&attn_weight},
                           im2col_step);
    check_input(grad_output, "grad_output");
    TORCH_CHECK(grad_output.scalar_type() == value.scalar_type(),
                "grad_output must have the dtype of value");
    TORCH_CHECK(grad_output.numel() == (int64_t)d.B * d.Lq * d.H * d.C,
                "grad_output must have B*Lq*H*C elements");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    // the library defines every output element: no zero-fill
    at::Tensor grad_value = at::empty_like(value);
    at::Tensor grad_loc = at::empty_like(sampling_loc);
    at::Tensor grad_attn = at::empty_like(attn_weight);
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    void *st = current_stream(value);
    int rc;
    if (value.scalar_type() == at::kDouble) {
        rc = boxattn_bwd_f64(value.data_ptr<double>(), sh, ls, sampling_loc.data_ptr<double>(),
                             attn_weight.data_ptr<double>(), grad_output.data_ptr<double>(), d.B, d.S,
                             d.H, d.C, d.L, d.Lq, d.P, grad_value.data_ptr<double>(),
                             grad_loc.data_ptr<double>(), grad_attn.data_ptr<double>(), st);
    } else {
        const HostTables h(spatial_shapes, level_start_index);
        at::Tensor ws = workspace(value, d, h);
        if (value.scalar_type() == at::kFloat)
            rc = boxattn_bwd_ws_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                    attn_weight.data_ptr<float>(), grad_output.data_ptr<float>(),
                                    d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, grad_value.data_ptr<float>(),
                                    grad_loc.data_ptr<float>(), grad_attn.data_ptr<float>(), h.sh(),
                                    h.ls(), ws.data_ptr(), (size_t)ws.numel(), nullptr, 0, 0, st);
        else
            rc = boxattn_bwd_ws_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                                     attn_weight.data_ptr<float>(), bf(grad_output), d.B, d.S, d.H,
                                     d.C, d.L, d.Lq, d.P, bf(grad_value), grad_loc.data_ptr<float>(),
                                     grad_attn.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(),
                                     (size_t)ws.numel(), nullptr, 0, 0, st);
    }
    check_rc(rc, "boxattn_bwd");
    return {grad_value, grad_loc, grad_attn};
}

// instance_attn.h:32-59 -> {output (B,Lq,H*C), mask_output (B,Lq,P,H*C)}
std::vector<at::Tensor> instance_attn_forward(const at::Tensor &value,
                                              const at::Tensor &spatial_shapes,
                                              const at::Tensor &level_start_index,
                                              const at::Tensor &sampling_loc,
                                              const at::Tensor &spatial_attn_weight,
                                              const at::Tensor &level_attn_weight,
                                              const int im2col_step)
{
    const Dims d = prepare(value, spatial_shapes, level_start_index, sampling_loc,
                           {&spatial_attn_weight, &level_attn_weight}, im2col_step);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor out = at::empty({d.B, d.Lq, (int64_t)d.H * d.C}, value.options());
    at::Tensor mask = at::empty({d.B, d.Lq, d.P, (int64_t)d.H * d.C}, value.options());
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    void *st = current_stream(value);
    int rc;
    if (value.scalar_type() == at::kDouble)
        rc = instattn_fwd_f64(value.data_ptr<double>(), sh, ls, sampling_loc.data_ptr<double>(),
                              spatial_attn_weight.data_ptr<double>(),
                              level_attn_weight.data_ptr<double>(), d.B, d.S, d.H, d.C, d.L, d.Lq,
                              d.P, out.data_ptr<double>(), mask.data_ptr<double>(), st);
    else if (value.scalar_type() == at::kFloat)
        rc = instattn_fwd_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                              spatial_attn_weight.data_ptr<float>(),
                              level_attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                              out.data_ptr<float>(), mask.data_ptr<float>(), st);
    else
        rc = instattn_fwd_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                               spatial_attn_weight.data_ptr<float>(),
                               level_attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq,
                               d.P, bf(out), bf(mask), st);
    check_rc(rc, "instattn_fwd");
    return {out, mask};
}

// instance_attn.h:61-92 -> {grad_value, grad_sampling_loc, grad_spatial_attn_weight,
//                           grad_level_attn_weight}
std::vector<at::Tensor> instance_attn_backward(
    const at::Tensor &value, const at::Tensor &spatial_shapes, const at::Tensor &level_start_index,
    const at::Tensor &sampling_loc, const at::Tensor &spatial_attn_weight,
    const at::Tensor &level_attn_weight, const at::Tensor &grad_output,
    const at::Tensor &grad_mask_output, const int im2col_step)
{
    const Dims d = prepare(value, spatial_shapes, level_start_index, sampling_loc,
                           {&spatial_attn_weight, &level_attn_weight}, im2col_step);
    check_input(grad_output, "grad_output");
    check_input(grad_mask_output, "grad_mask_output");
    TORCH_CHECK(grad_output.scalar_type() == value.scalar_type() &&
                    grad_mask_output.scalar_type() == value.scalar_type(),
                "grad_output / grad_mask_output must have the dtype of value");
    TORCH_CHECK(grad_output.numel() == (int64_t)d.B * d.Lq * d.H * d.C &&
                    grad_mask_output.numel() == (int64_t)d.B * d.Lq * d.P * d.H * d.C,
                "grad_output / grad_mask_output have the wrong number of elements");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    at::Tensor grad_value = at::empty_like(value);
    at::Tensor grad_loc = at::empty_like(sampling_loc);
    at::Tensor grad_sw = at::empty_like(spatial_attn_weight);
    at::Tensor grad_lw = at::empty_like(level_attn_weight);
    const int64_t *sh = spatial_shapes.data_ptr<int64_t>(), *ls = level_start_index.data_ptr<int64_t>();
    void *st = current_stream(value);
    int rc;
    if (value.scalar_type() == at::kDouble) {
        rc = instattn_bwd_f64(value.data_ptr<double>(), sh, ls, sampling_loc.data_ptr<double>(),
                              spatial_attn_weight.data_ptr<double>(),
                              level_attn_weight.data_ptr<double>(), grad_output.data_ptr<double>(),
                              grad_mask_output.data_ptr<double>(), d.B, d.S, d.H, d.C, d.L, d.Lq,
                              d.P, grad_value.data_ptr<double>(), grad_loc.data_ptr<double>(),
                              grad_sw.data_ptr<double>(), grad_lw.data_ptr<double>(), st);
    } else {
        const HostTables h(spatial_shapes, level_start_index);
        at::Tensor ws = workspace(value, d, h);
        if (value.scalar_type() == at::kFloat)
            rc = instattn_bwd_ws_f32(
                value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                grad_output.data_ptr<float>(), grad_mask_output.data_ptr<float>(), d.B, d.S, d.H,
                d.C, d.L, d.Lq, d.P, grad_value.data_ptr<float>(), grad_loc.data_ptr<float>(),
                grad_sw.data_ptr<float>(), grad_lw.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(),
                (size_t)ws.numel(), nullptr, 0, 0, st);
        else
            rc = instattn_bwd_ws_bf16(
                bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                bf(grad_output), bf(grad_mask_output), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                bf(grad_value), grad_loc.data_ptr<float>(), grad_sw.data_ptr<float>(),
                grad_lw.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(), (size_t)ws.numel(), nullptr, 0, 0, st);
    }
    check_rc(rc, "instattn_bwd");
    return {grad_value, grad_loc, grad_sw, grad_lw};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("box_attn_forward", &box_attn_forward, "box_attn_forward");
    m.def("box_attn_backward", &box_attn_backward, "box_attn_backward");
    m.def("instance_attn_forward", &instance_attn_forward, "instance_attn_forward");
    m.def("instance_attn_backward", &instance_attn_backward, "instance_attn_backward");
    m.def("abi_version", &boxattn_abi_version, "ABI version of the linked libboxattn_hip.so");
    m.attr("compiled_abi_version") = BOXATTN_ABI_VERSION;      // of the header this module was compiled against
}

}  // namespace e2edet
