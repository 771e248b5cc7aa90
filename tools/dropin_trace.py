# -*- coding: utf-8 -*-
"""The drop-in entry (rmnet_memory_read_f32) at the dense 480p T=5 one-object size: per-call time enqueued from Python,
enqueued with a preallocated output, and replayed from a HIP graph (device side only).  Under rocprofv3 --kernel-trace
--stats the per-kernel rows say where the device time goes."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rmnet_amd import ops
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
ev = bench.HipEvents(4)
g = torch.Generator().manual_seed(0)
mk = (torch.randn(1, 128, 5, 30, 54, generator=g) * 0.6).to(dev)
mv = torch.randn(1, 512, 5, 30, 54, generator=g).to(dev)
qk = (torch.randn(1, 128, 30, 54, generator=g) * 0.6).to(dev)
qv = torch.randn(1, 512, 30, 54, generator=g).to(dev)
flags = int(os.environ.get('FLAGS', '0'))
N = int(os.environ.get('N', '50'))
out = torch.empty(1, 1024, 30, 54, device=dev)
res = {}


def bracket(fn, n, stream):
    st = stream.cuda_stream
    floor = ev.floor_us(st)
    fn(); torch.cuda.synchronize()
    ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[0]), ctypes.c_void_p(st))
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) * 1e6 / n
    ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[1]), ctypes.c_void_p(st))
    torch.cuda.synchronize()
    return round((ev.elapsed_ms(ev.ev[0], ev.ev[1]) * 1e3 - floor) / n, 2), round(host, 2)


cur = torch.cuda.current_stream(dev)
for _ in range(3):
    ops.memory_read(mk, mv, qk, qv, flags=flags)
res['python_alloc_out'] = bracket(lambda: ops.memory_read(mk, mv, qk, qv, flags=flags), N, cur)
res['python_given_out'] = bracket(lambda: ops.memory_read(mk, mv, qk, qv, flags=flags, out=out), N, cur)
if not os.environ.get('NOGRAPH'):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        ops.memory_read(mk, mv, qk, qv, flags=flags, out=out)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            ops.memory_read(mk, mv, qk, qv, flags=flags, out=out)
        res['graph_replay'] = bracket(gr.replay, N, s)
print(json.dumps({'us_per_call_device, us_per_call_host_enqueue': res}))
