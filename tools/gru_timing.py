#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of one recurrence step (needs a build with -DSLU_GRU_TIMING,
which makes the forward kernel dump per-wave averages into the start of the reserve buffer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from slu_hip import ops
T, B, H = 300, 64, 128
gx = torch.randn(T, B, 6 * H, device="cuda")
wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
for _ in range(2):
    out, rsv = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, 2, True)
torch.cuda.synchronize()
t = rsv[:8 * 8 * 8].view(8, 8, 8)[:, :, :4].cpu()      # (wg, wave, phase)
names = ["lds-read wait", "mfma phase", "tail (tanh/blend/stores)", "barrier"]
print("s_memtime ticks per step (100 MHz constant clock?) averaged over waves, per workgroup:")
print(t.mean(1))
tot = t.sum(2).mean()
for k, n in enumerate(names):
    print("%-28s %8.1f ticks  %5.1f %%" % (n, t[:, :, k].mean().item(), 100 * t[:, :, k].mean().item() / tot.item()))
print("total ticks/step", tot.item())
print("per-wave detail of wg 0:\n", t[0])
