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
#include <c10/hip/HIPGraphsC10Utils.h>

#include <algorithm>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
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

// ---- what a training step must not pay for on every call (boxer_amd/ops.py keeps the same three things):
//  * host copies of the two level tables: one device -> host copy (a synchronisation) per TENSOR, not per call --
//    BoxeR hands the same two tensors to every layer of every step;
//  * the backward's scratch: one tensor per (device, stream), grown as needed (0.1-0.3 GB at BoxeR-R50 shapes);
//  * the plan: a forward whose inputs require a gradient runs the *_fwd_train_* entry and PARKS the plan; the
//    backward -- which the reference's Function calls with nothing but the saved tensors
//    (box_attention_func.py:24-64) -- finds it under (stream, dimensions, location / weight addresses + versions).
std::mutex g_mu;

struct TableEntry {
    c10::weak_intrusive_ptr<c10::TensorImpl> impl;     // the tensor object the copy was made of ...
    uint32_t version;                                  // ... and its version counter then
    std::shared_ptr<const std::vector<int64_t>> data;
};
std::map<const c10::TensorImpl *, TableEntry> g_tables;
constexpr size_t kTableCap = 64;

uint32_t version_of(const at::Tensor &t)
{
    return t.is_inference() ? 0u : (uint32_t)t._version();
}

// Host copy of a small int64 device table, cached on the tensor OBJECT: an address alone may be a new tensor's (the
// weak reference tells a live object from a recycled address), a version tells an in-place update.
std::shared_ptr<const std::vector<int64_t>> host_table(const at::Tensor &t)
{
    const c10::TensorImpl *key = t.unsafeGetTensorImpl();
    const uint32_t ver = version_of(t);
    {
        std::lock_guard<std::mutex> g(g_mu);
        const auto it = g_tables.find(key);
        if (it != g_tables.end()) {
            const auto alive = it->second.impl.lock();
            if (alive && alive.get() == key && it->second.version == ver) return it->second.data;
            g_tables.erase(it);
        }
    }
    const at::Tensor h = t.cpu().contiguous();         // (the one synchronising copy)
    auto data = std::make_shared<const std::vector<int64_t>>(h.data_ptr<int64_t>(), h.data_ptr<int64_t>() + h.numel());
    std::lock_guard<std::mutex> g(g_mu);
    if (g_tables.size() >= kTableCap) g_tables.clear();
    g_tables.erase(key);
    g_tables.emplace(key, TableEntry{c10::weak_intrusive_ptr<c10::TensorImpl>(t.getIntrusivePtr()), ver, data});
    return data;
}
struct HostTables {
    std::shared_ptr<const std::vector<int64_t>> shapes, lsi;
    HostTables(const at::Tensor &s, const at::Tensor &l) : shapes(host_table(s)), lsi(host_table(l)) {}
    const int64_t *sh() const { return shapes->data(); }
    const int64_t *ls() const { return lsi->data(); }
};

typedef std::pair<int, void *> StreamKey;               // (device index, stream handle)
std::map<StreamKey, at::Tensor> g_workspace;
// (device, stream, B, S, H, C, L, Lq, P, storage type, hash of the level shapes): ABI 8 keeps the record ranges of the
// backward's one-pass fill in the state buffer -- one buffer per geometry
typedef std::tuple<int, void *, int, int, int, int, int, int, int, int, uint64_t> StateKey;
struct StateEntry { at::Tensor buf; bool fresh; };      // fresh: zeroed, not yet handed to the library (BOXATTN_HINT_FRESH_STATE)
std::map<StateKey, StateEntry> g_state;
constexpr size_t kStateCap = 1024;

// a tensor created while the stream captures a graph lives in the graph's private pool: it serves the call, but must
// not outlive the graph in a process-wide table (boxer_amd/ops.py has the same guard)
bool capturing()
{
    return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None;
}

// the backward's scratch of this stream (contents only live inside one call; calls on a stream never overlap)
at::Tensor workspace(const at::Tensor &value, const Dims &d, const HostTables &h, void *stream)
{
    const size_t bytes = std::max<size_t>(
        boxattn_bwd_workspace_bytes(value.scalar_type() == at::kBFloat16, d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                                    h.sh(), h.ls()), 256);
    const StreamKey key(value.get_device(), stream);
    const bool keep = !capturing();
    std::lock_guard<std::mutex> g(g_mu);
    const auto it = g_workspace.find(key);
    if (it != g_workspace.end() && (size_t)it->second.numel() >= bytes) return it->second;
    at::Tensor ws = at::empty({(int64_t)bytes}, value.options().dtype(at::kByte));
    if (keep) g_workspace[key] = ws;
    return ws;
}
// the library's state buffer of this (stream, geometry): zeroed once; every call leaves its tickets zero and the
// record ranges of the next backward's one-pass fill
at::Tensor state(const at::Tensor &value, const Dims &d, const HostTables &h, void *stream, int *hints)
{
    const size_t bytes = boxattn_state_bytes(d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, h.sh(), h.ls());
    uint64_t geo = 1469598103934665603ull;
    for (int i = 0; i < 2 * d.L; ++i) geo = (geo ^ (uint64_t)h.sh()[i]) * 1099511628211ull;
    const StateKey key(value.get_device(), stream, d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, (int)value.scalar_type(), geo);
    const bool keep = !capturing();
    std::lock_guard<std::mutex> g(g_mu);
    const auto it = g_state.find(key);
    if (it != g_state.end() && (size_t)it->second.buf.numel() >= bytes) {
        if (it->second.fresh) *hints |= BOXATTN_HINT_FRESH_STATE;
        it->second.fresh = false;
        return it->second.buf;
    }
    at::Tensor st = at::zeros({(int64_t)bytes}, value.options().dtype(at::kByte));
    *hints |= BOXATTN_HINT_FRESH_STATE;
    if (keep) {
        if (g_state.size() >= kStateCap) g_state.clear();
        g_state[key] = StateEntry{st, false};
    }
    return st;
}

// ---- parked plans
typedef std::tuple<int, void *, int, int, int, int, int, int, int, int, int, const void *, uint32_t, const void *, uint32_t,
                   const void *, uint32_t> PlanKey;
struct Parked {
    PlanKey key;
    at::Tensor plan;
    std::vector<at::Tensor> keep;      // the tensors the key names: held, so that their addresses stay theirs
};
std::deque<Parked> g_parked;
constexpr size_t kParkCap = 32;

PlanKey plan_key(const at::Tensor &value, void *stream, const Dims &d, const at::Tensor &loc, const at::Tensor &w0,
                 const at::Tensor *w1)
{
    // (the storage type is part of the key: at 16 / 64 channels per head a bfloat16 plan bins contiguous query ranges,
    // a float32 one interleaved ones)
    return PlanKey(value.get_device(), stream, d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, boxattn_options_epoch(),
                   (int)value.scalar_type(), loc.data_ptr(), version_of(loc), w0.data_ptr(), version_of(w0), w1 ? w1->data_ptr() : nullptr,
                   w1 ? version_of(*w1) : 0u);
}
void park(PlanKey key, at::Tensor plan, std::vector<at::Tensor> keep)
{
    if (capturing()) return;       // (the plan tensor belongs to the graph's pool; the backward plans for itself)
    std::lock_guard<std::mutex> g(g_mu);
    for (auto it = g_parked.begin(); it != g_parked.end();)
        it = it->key == key ? g_parked.erase(it) : it + 1;
    while (g_parked.size() >= kParkCap) g_parked.pop_front();
    g_parked.push_back(Parked{key, std::move(plan), std::move(keep)});
}
at::Tensor take_parked(const PlanKey &key)
{
    std::lock_guard<std::mutex> g(g_mu);
    for (auto it = g_parked.begin(); it != g_parked.end(); ++it)
        if (it->key == key) {
            at::Tensor plan = it->plan;
            g_parked.erase(it);
            return plan;
        }
    return at::Tensor();
}
bool wants_plan(std::initializer_list<const at::Tensor *> ts)
{
    for (const at::Tensor *t : ts)
        if (t->requires_grad()) return true;
    return false;
}
at::Tensor plan_buffer(const at::Tensor &value, const Dims &d, const HostTables &h)
{
    const size_t bytes = boxattn_plan_bytes(value.scalar_type() == at::kBFloat16, d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                                            h.sh(), h.ls());
    return bytes ? at::empty({(int64_t)bytes}, value.options().dtype(at::kByte)) : at::Tensor();
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
        at::Tensor plan;
        if (wants_plan({&value, &sampling_loc, &attn_weight})) plan = plan_buffer(value, d, h);
        if (plan.defined()) {          // a backward will follow: the forward's launch prepares its plan
            int hints = 0;
            at::Tensor state_buf = state(value, d, h, st, &hints);
            int built = 0;
            if (value.scalar_type() == at::kFloat)
                rc = boxattn_fwd_train_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                           attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                                           out.data_ptr<float>(), h.sh(), h.ls(), plan.data_ptr(), (size_t)plan.numel(),
                                           state_buf.data_ptr(), (size_t)state_buf.numel(), hints, &built, st);
            else
                rc = boxattn_fwd_train_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                                            attn_weight.data_ptr<float>(), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, bf(out),
                                            h.sh(), h.ls(), plan.data_ptr(), (size_t)plan.numel(), state_buf.data_ptr(),
                                            (size_t)state_buf.numel(), hints, &built, st);
            if (rc == 0 && built)
                park(plan_key(value, st, d, sampling_loc, attn_weight, nullptr), plan, {sampling_loc, attn_weight});
        } else if (value.scalar_type() == at::kFloat)
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
        at::Tensor ws = workspace(value, d, h, st);
        int hints = 0;
            at::Tensor state_buf = state(value, d, h, st, &hints);
        const at::Tensor plan = take_parked(plan_key(value, st, d, sampling_loc, attn_weight, nullptr));
        const void *pp = plan.defined() ? plan.data_ptr() : nullptr;
        const size_t pn = plan.defined() ? (size_t)plan.numel() : 0;
        if (value.scalar_type() == at::kFloat)
            rc = boxattn_bwd_ws_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                    attn_weight.data_ptr<float>(), grad_output.data_ptr<float>(),
                                    d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, grad_value.data_ptr<float>(),
                                    grad_loc.data_ptr<float>(), grad_attn.data_ptr<float>(), h.sh(),
                                    h.ls(), ws.data_ptr(), (size_t)ws.numel(), pp, pn, state_buf.data_ptr(),
                                    (size_t)state_buf.numel(), hints, st);
        else
            rc = boxattn_bwd_ws_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                                     attn_weight.data_ptr<float>(), bf(grad_output), d.B, d.S, d.H,
                                     d.C, d.L, d.Lq, d.P, bf(grad_value), grad_loc.data_ptr<float>(),
                                     grad_attn.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(),
                                     (size_t)ws.numel(), pp, pn, state_buf.data_ptr(), (size_t)state_buf.numel(), hints, st);
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
    at::Tensor plan;
    if (value.scalar_type() != at::kDouble &&
        wants_plan({&value, &sampling_loc, &spatial_attn_weight, &level_attn_weight})) {
        const HostTables h(spatial_shapes, level_start_index);
        plan = plan_buffer(value, d, h);
        if (plan.defined()) {
            int hints = 0;
            at::Tensor state_buf = state(value, d, h, st, &hints);
            int built = 0;
            if (value.scalar_type() == at::kFloat)
                rc = instattn_fwd_train_f32(value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                                            spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                                            d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, out.data_ptr<float>(),
                                            mask.data_ptr<float>(), h.sh(), h.ls(), plan.data_ptr(), (size_t)plan.numel(),
                                            state_buf.data_ptr(), (size_t)state_buf.numel(), hints, &built, st);
            else
                rc = instattn_fwd_train_bf16(bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                                             spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                                             d.B, d.S, d.H, d.C, d.L, d.Lq, d.P, bf(out), bf(mask), h.sh(), h.ls(),
                                             plan.data_ptr(), (size_t)plan.numel(), state_buf.data_ptr(),
                                             (size_t)state_buf.numel(), hints, &built, st);
            if (rc == 0 && built)
                park(plan_key(value, st, d, sampling_loc, spatial_attn_weight, &level_attn_weight), plan,
                     {sampling_loc, spatial_attn_weight, level_attn_weight});
            check_rc(rc, "instattn_fwd");
            return {out, mask};
        }
    }
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
        at::Tensor ws = workspace(value, d, h, st);
        int hints = 0;
            at::Tensor state_buf = state(value, d, h, st, &hints);
        const at::Tensor plan =
            take_parked(plan_key(value, st, d, sampling_loc, spatial_attn_weight, &level_attn_weight));
        const void *pp = plan.defined() ? plan.data_ptr() : nullptr;
        const size_t pn = plan.defined() ? (size_t)plan.numel() : 0;
        if (value.scalar_type() == at::kFloat)
            rc = instattn_bwd_ws_f32(
                value.data_ptr<float>(), sh, ls, sampling_loc.data_ptr<float>(),
                spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                grad_output.data_ptr<float>(), grad_mask_output.data_ptr<float>(), d.B, d.S, d.H,
                d.C, d.L, d.Lq, d.P, grad_value.data_ptr<float>(), grad_loc.data_ptr<float>(),
                grad_sw.data_ptr<float>(), grad_lw.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(),
                (size_t)ws.numel(), pp, pn, state_buf.data_ptr(), (size_t)state_buf.numel(), hints, st);
        else
            rc = instattn_bwd_ws_bf16(
                bf(value), sh, ls, sampling_loc.data_ptr<float>(),
                spatial_attn_weight.data_ptr<float>(), level_attn_weight.data_ptr<float>(),
                bf(grad_output), bf(grad_mask_output), d.B, d.S, d.H, d.C, d.L, d.Lq, d.P,
                bf(grad_value), grad_loc.data_ptr<float>(), grad_sw.data_ptr<float>(),
                grad_lw.data_ptr<float>(), h.sh(), h.ls(), ws.data_ptr(), (size_t)ws.numel(), pp, pn,
                state_buf.data_ptr(), (size_t)state_buf.numel(), hints, st);
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
    m.def("parked_plans", []() { std::lock_guard<std::mutex> g(g_mu); return (int)g_parked.size(); },
          "plans parked by training forwards whose backward has not run yet");
    m.def("release_buffers", []() {
        std::lock_guard<std::mutex> g(g_mu);
        g_parked.clear(); g_workspace.clear(); g_state.clear(); g_tables.clear();
    }, "drop the cached scratch / state tensors, host tables and parked plans");
    m.attr("compiled_abi_version") = BOXATTN_ABI_VERSION;      // of the header this module was compiled against
}

}  // namespace e2edet
