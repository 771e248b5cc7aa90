# -*- coding: utf-8 -*-
"""The hand-written kernels one by one (bench.kernel_figures plus the drop-in entry at the dense 480p T=5
size) -- run it under rocprofv3 --kernel-trace --stats for the per-kernel rows of profiles/."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rmnet_amd import ops
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
ev = bench.HipEvents(8)
out = bench.kernel_figures(dev, ev)
# drop-in MemoryReader entry (rmnet_memory_read_f32): dense 480p, T = 5, 1 object -- staging + bank read + combine
g = torch.Generator().manual_seed(0)
mk = (torch.randn(1, 128, 5, 30, 54, generator=g) * 0.6).to(dev)
mv = torch.randn(1, 512, 5, 30, 54, generator=g).to(dev)
qk = (torch.randn(1, 128, 30, 54, generator=g) * 0.6).to(dev)
qv = torch.randn(1, 512, 30, 54, generator=g).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
import ctypes
import numpy as np
floor = ev.floor_us(st)
for name, flags in (('dropin_dense_480p_T5_default', 0), ('dropin_dense_480p_T5_exact_fp32', ops.MR_EXACT_FP32)):
    for _ in range(3):
        ops.memory_read(mk, mv, qk, qv, flags=flags)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[0]), ctypes.c_void_p(st))
        ops.memory_read(mk, mv, qk, qv, flags=flags)
        ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[1]), ctypes.c_void_p(st))
        torch.cuda.synchronize()
        ts.append(ev.elapsed_ms(ev.ev[0], ev.ev[1]) * 1e3 - floor)
    ab = bench.algorithmic_bytes(1, 5, 30, 54)
    # pipelined: 20 calls enqueued back to back, one bracket (launch gaps overlap the previous call's kernels)
    ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[0]), ctypes.c_void_p(st))
    for _ in range(20):
        ops.memory_read(mk, mv, qk, qv, flags=flags)
    ev.hip.hipEventRecord(ctypes.c_void_p(ev.ev[1]), ctypes.c_void_p(st))
    torch.cuda.synchronize()
    piped = (ev.elapsed_ms(ev.ev[0], ev.ev[1]) * 1e3 - floor) / 20
    out[name] = {'us_whole_call_isolated': round(float(np.mean(ts)), 2), 'us_per_call_back_to_back': round(piped, 2),
                 'algorithmic_bytes': ab,
                 'hbm_frac_back_to_back': round(ab / piped / 1e3 / bench.HBM_PEAK_GBS, 4),
                 'note': 'isolated = one call between two host syncs (5 launches incl. their host-side gaps); '
                         'back to back = 20 calls in one bracket'}
print(json.dumps(out, indent=1))
