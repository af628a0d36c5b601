"""Does a hipGraph captured with an EXTERNAL event-record node let another stream start work in the middle of a replay?
(The mechanism behind the data-parallel exchange overlapped with a REPLAYED backward: graph_step.GraphedTrainStep.)
torch.cuda.Event(external=True) is refused on ROCm builds of PyTorch, so the HIP calls are made directly (ctypes on the runtime
torch has already loaded): hipEventRecordWithFlags(ev, stream, hipEventRecordExternal) inside the capture, hipStreamWaitEvent on
the side stream after each launch.  Prints, per replay: the value the side stream saw behind the event (must be the replay's own),
and when the side stream's copy finished relative to the end of the replay (earlier = it overlapped the rest of the graph)."""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
side = torch.cuda.Stream()
a = torch.zeros(1 << 20, device="cuda")
b = torch.zeros(1 << 27, device="cuda")
out = torch.zeros(1 << 20, device="cuda")
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0            # hipEventDisableTiming
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a.add_(1)
    rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1)   # hipEventRecordExternal
    for _ in range(40):
        b.add_(1)                               # ~40 x 0.25 ms of streaming work behind the event
print("hipEventRecordWithFlags(external) inside the capture ->", rc)
t_side, t_end, t0 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for it in range(4):
    t0.record()
    g.replay()
    t_end.record()
    with torch.cuda.stream(side):
        rcw = hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0)
        out.copy_(a)
        t_side.record(side)
    torch.cuda.synchronize()
    print(f"replay {it}: wait rc {rcw}; side stream saw a = {out[0].item():.0f} (want {it + 1}); graph took {t0.elapsed_time(t_end):.2f} ms; "
          f"side copy finished {t0.elapsed_time(t_side):.2f} ms after the launch")
