#!/bin/bash
# Socket power and clocks while bk_main runs back to back (evidence for DESIGN.md section 5's power-cap reading).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^$" | head -30
echo "---- under load (bk_main + mr_combine in a loop, 8 objects, T = 10)"
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rmnet_amd import ops
dev = torch.device('cuda', 0)
no, T, h, w = 8, 10, 30, 54
g = torch.Generator().manual_seed(0)
mk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, h, w, generator=g).to(dev)
r = torch.tensor([(3, 34, 2, 25)] * no, dtype=torch.int32, device=dev)
q = torch.tensor([(2, 25, 1, 21)] * no, dtype=torch.int32, device=dev)
bank = ops.MemoryBank(no, T, h, w, dev)
for t in range(T):
    bank.append(t, mk, mv, r)
t0 = time.time()
while time.time() - t0 < 8.0:
    for _ in range(200):
        bank.read(T, mk, mv, q)
    torch.cuda.synchronize()
PY
sleep 4
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | head -6; sleep 1; done
wait
