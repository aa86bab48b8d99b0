"""Debug aid: the training forward's scan workgroups under HIP graph replay."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
from boxer_amd import ops, _lib
lib = _lib.load()
g = T._seeded([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 333, 4, seed=21)
dev_ = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a)).cuda()
dtype = torch.float32
value = dev_(g["value"], dtype); shapes, lsi = dev_(g["shapes"]), dev_(g["lsi"])
loc, attn = dev_(g["loc"], torch.float32), dev_(g["attn"], torch.float32)
gout = dev_(g["grad_out"], dtype)
plans = []
def run():
    out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
    plans.append(plan)
    return (out,) + tuple(ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan))
for zero_ws in (False, True):
    for _ in range(3):
        eager = [t.clone() for t in run()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            captured = run()
    ws = plans[-1].ws
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(4):
        if zero_ws:
            ws.zero_()
        graph.replay(); torch.cuda.synchronize()
        lib.boxattn_set_option(15, 1); want = run(); torch.cuda.synchronize(); lib.boxattn_set_option(15, 0)
        print("zero_ws", zero_ws, "replay", rep, [float((a.float() - b.float()).abs().max()) for a, b in zip(captured, want)])
        value.mul_(0.9); gout.mul_(1.1)
