# -*- coding: utf-8 -*-
"""Time line of bk_main INSIDE the frame loop (needs a -DBK_CLK=1 library through RMNET_HIP_LIB): the bench workload
(8 clips of 480x854, 1 object, memory pinned at T = 5), the bank read given a persistent workspace so that the last
launch's per-workgroup stamps can be read back.   python tools/loop_clk.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rmnet_amd import networks, ops, _lib
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
from rmnet_amd.tiny_flownet import TinyFlowNet
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = os.environ.get('RMNET_BANK_PRECISION', 'f16')
dev = torch.device('cuda', 0)
torch.set_grad_enabled(False)
torch.backends.cudnn.benchmark = os.environ.get('FIND', '0') == '1'
H, W, K, T = bench.H, bench.W, bench.K_CH, bench.T_MEM
B = int(os.environ.get('CLIPS', bench.DEFAULT_CLIPS))   # clips per GPU (CLIPS=8: the launch of rounds 1-4)
net = networks.procedural_init_(RMNet(None, read_precision=prec)).to(dev).eval()
tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval()
net.fuse_epilogues(); tfn.fuse_epilogues()
n_clip = 12
clips = [synthetic_clip(n_clip, K, H, W, seed=c, size=2.1) for c in range(B)]
frames = torch.cat([c[0] for c in clips]).to(dev)
masks = torch.cat([c[1] for c in clips]).to(dev).float()
ctx = net._ClipContext(net, B, K, H, W, [K - 1] * B, dev)
bank = net.new_bank(ctx, T)
lib = _lib.load()
nb = lib.rmnet_bank_read_workspace_bytes(B, ctx.h, ctx.w)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
orig_read = bank.read
def read_ws(*a, **k):
    k['ws'] = ws
    return orig_read(*a, **k)
bank.read = read_ws
for t in range(1, T):
    net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], tfn._forward(frames[:, t], frames[:, t - 1]), commit=True)
ev = bench.HipEvents(3 * steps)
floor = ev.floor_us(torch.cuda.current_stream(dev).cuda_stream)
for i in range(3 + steps):
    t = T + (i % (n_clip - T))
    net._profile_events = tuple(ev.ev[3 * (i - 3):3 * (i - 3) + 3]) if i >= 3 else None
    net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], tfn._forward(frames[:, t], frames[:, t - 1]), commit=False)
torch.cuda.synchronize()
us_ = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 - floor for i in range(steps)]
print('in-loop bk_main (%s): avg %.2f min %.2f max %.2f us' % (prec, np.mean(us_), np.min(us_), np.max(us_)))
if os.environ.get('DIST') == '1':
    print('   per launch, sorted: ' + ' '.join('%.1f' % x for x in sorted(us_)))
no = B
plan_end = no * 12 * 4
base = nb - 16384 - ((plan_end + 255) // 256 * 256)
plan = ws[base:][:plan_end].view(torch.int32).cpu().numpy().reshape(no, 12)
print('plan records (Mq nqt njt M rect.. slot C):'); print(plan)
raw = ws[base + plan_end + 64:][:256 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
if raw[:, 1].max() > 0:
    us = raw / 100.0
    comp = us[:, 6] > 0
    def st(col, sel):
        v = us[sel, col]
        return '%.1f [%.1f..%.1f]' % (np.median(v), v.min(), v.max())
    print('plan inputs in LDS %s' % st(7, comp))
    print('compute WGs (%d): plan done %s; first walk starts %s; last walk over %s; compute part over %s; left %s'
          % (comp.sum(), st(4, comp), st(5, comp), st(6, comp), st(1, comp), st(3, comp)))
    aside = ~comp & (us[:, 3] > 0)
    if aside.any():
        print('set-aside WGs (%d): tickets %s; left %s' % (aside.sum(), sorted(raw[aside, 2].tolist()), st(3, aside)))
    print('tickets served by compute WGs: %d' % raw[comp, 2].sum())
    walk = us[:, 6] - us[:, 5]
    print('walk us by XCD: ' + '  '.join('%d: med %.1f max %.1f' % (x, np.median(walk[comp & (np.arange(256) % 8 == x)]), walk[comp & (np.arange(256) % 8 == x)].max()) for x in range(8)))
    cyc = raw[:, 0] / (raw[:, 1] * 10.0 + 1e-9)
    print('clock GHz: med %.2f min %.2f max %.2f' % (np.median(cyc[comp]), cyc[comp].min(), cyc[comp].max()))
    if os.environ.get('EPILOGUE') == '1':   # a build with the (uncommitted) epilogue stamps: col 0 = E2 passed / published, 2 = nsp*1000 + ticket*10 + role, 4 = merged, 7 = stores issued
        code = raw[:, 2]
        role = code % 10
        nsp = code // 1000
        for r, name in ((1, 'last arrivers'), (0, 'early arrivers'), (2, 'single segment')):
            sel = comp & (role == r) & (nsp > 0)
            if not sel.any():
                continue
            if r == 0:
                print('%s (%d): walk over %s; published %s; left %s' % (name, sel.sum(), st(6, sel), st(0, sel), st(3, sel)))
            else:
                print('%s (%d, nsp %s): walk over %s; partials there %s; merged %s; stores issued %s; left %s'
                      % (name, sel.sum(), np.bincount(nsp[sel]).tolist(), st(6, sel), st(0, sel), st(4, sel), st(7, sel), st(3, sel)))
                d = us[sel]
                print('   deltas: walk->there %.1f  there->merged %.1f  merged->stored %.1f  stored->left %.1f (medians)'
                      % (np.median(d[:, 0] - d[:, 6]), np.median(d[:, 4] - d[:, 0]), np.median(d[:, 7] - d[:, 4]), np.median(d[:, 3] - d[:, 7])))
                worst = np.argsort(-d[:, 3])[:6]
                for w in worst:
                    print('   late: nsp %d walk %.1f there %.1f merged %.1f stored %.1f left %.1f' % (nsp[sel][w], d[w, 6], d[w, 0], d[w, 4], d[w, 7], d[w, 3]))
