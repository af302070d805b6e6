"""Diagnostic: gradient w.r.t. every stage output of a full-size ASR step, HIP kernels vs the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch  # noqa: E402
from oracle import slu_oracle as O  # noqa: E402
import models  # noqa: E402

cfg = O.OracleConfig(pretraining_type=2)
cfg.folder = "/tmp"
torch.manual_seed(31)
pm = models.PretrainedModel(cfg)
sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in pm.state_dict().items()}
g = torch.Generator().manual_seed(32)
B, T = 64, 48000
x = 0.1 * torch.randn(B, T, generator=g)
Tp, Tw = -(-T // 640), -(-T // 2560)
yp = torch.randint(0, cfg.num_phonemes, (B, Tp), generator=g)
yw = torch.randint(0, cfg.vocabulary_size, (B, Tw), generator=g)
yp[torch.rand(B, Tp, generator=g) < 0.1] = -1
yw[torch.rand(B, Tw, generator=g) < 0.1] = -1
masks = O.draw_dropout_masks(cfg, x, seed=33, include_intent=False)
models.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
pm.train()
grads, outs = {}, {}
stages = pm._stages()
for i, st in enumerate(stages):
    run = st.run

    def wrapped(h, training, out_planes=False, _run=run, _i=i):
        y = _run(h, training, out_planes)
        outs[_i] = y
        y.register_hook(lambda gr, _i=_i: grads.__setitem__(_i, gr.detach().clone()))
        return y
    st.run = wrapped
from slu_hip import ops as _ops
rec = {}
_act, _data = _ops.wconv_bwd_act, _ops.wconv_bwd_data


def rec_act(dy, y, route, B_, l_conv, c_out, do_abs, pool, slope, tm):
    r = _act(dy, y, route, B_, l_conv, c_out, do_abs, pool, slope, tm)
    rec[("act", c_out, tm)] = (dy.detach().clone(), y.detach().clone(), r.detach().clone())
    return r


def rec_data(d_conv, weight, B_, l_in):
    r = _data(d_conv, weight, B_, l_in)
    rec[("data", weight.shape[0], weight.shape[1])] = (d_conv.detach().clone(), weight.detach().clone(), r.detach().clone())
    return r


_ops.wconv_bwd_act, _ops.wconv_bwd_data = rec_act, rec_data
pl, wl, pa, wa = pm(x, yp, yw)
(pl + wl).backward()
torch.cuda.synchronize()
models.set_dropout_masks(None)
torch.set_num_threads(64)
enc = {}
orig = O.encoder_stages


def hooked(*a, **k):
    st = orig(*a, **k)
    for n, t in st.items():
        if t.requires_grad:
            t.retain_grad()
    enc.update(st)
    return st


O.encoder_stages = hooked
rpl, rwl, _, _ = O.asr_forward(sd, x, yp, yw, cfg, masks, explicit_gru=False)
(rpl + rwl).backward()
names = ["cnn0", "cnn1", "cnn2", "phone_down0", "phone_down1", "word_down0", "word_down1"]
for i, n in enumerate(names):
    ref = enc[n].grad
    got = grads[i].cpu()
    if i < 2:
        got = got.permute(0, 2, 1)                    # (B, L, C) -> (B, C, L)
    elif i == 2:
        got = got.permute(1, 2, 0)                    # (L, B, C) -> (B, C, L)
    else:
        got = got.permute(1, 0, 2)                    # (T, B, C) -> (B, T, C)
    fwd = outs[i].detach().cpu()
    fwd = fwd.permute(0, 2, 1) if i < 2 else (fwd.permute(1, 2, 0) if i == 2 else fwd.permute(1, 0, 2))
    e = (got - ref).abs()
    idx = e.flatten().argmax().item()
    pos = []
    for d_ in reversed(ref.shape):
        pos.append(idx % d_)
        idx //= d_
    print("%-12s forward rel err %.2e | grad rel err %.2e at %s (|grad| max %.3e, #>1e-4: %d)"
          % (n, (fwd - enc[n].detach()).abs().max() / enc[n].detach().abs().max(), e.max() / ref.abs().max(), tuple(reversed(pos)),
             ref.abs().max(), int((e > 1e-4 * ref.abs().max()).sum())))
pre = enc["conv1"].detach()[33, :, 94]
c = pre.abs().argmin().item()
hip1 = outs[1].detach().cpu()[33, 94, c].item()
print("oracle conv1 pre-activation at (b=33, l=94): min |v| = %.3e at channel %d; HIP post-activation there %.3e (slope-0.2 branch iff negative)" % (pre.abs().min().item(), c, hip1))
print("oracle value %.6e" % pre[c].item())

for key, val in rec.items():
    print(key, [tuple(t.shape) for t in val])
dc1, w1, dx1 = rec[("data", 60, 80)]                   # conv1: weight (60, 80, 5); d_conv (B, L, 60) -> dx (B, L, 80)
ref_dc1 = enc["conv1"].grad.permute(0, 2, 1)            # oracle d(conv1 pre-activation) (B, L, 60)
e = (dc1.cpu() - ref_dc1).abs()
print("d_conv1 (output of wconv_bwd_act) vs oracle: rel %.2e at %s" % (e.max() / ref_dc1.abs().max(), (e == e.max()).nonzero()[0].tolist()))
import torch.nn.functional as F
ref_dx = F.conv_transpose1d(dc1.cpu().double().permute(0, 2, 1), w1.cpu().double(), padding=2).permute(0, 2, 1)
e = (dx1.cpu().double() - ref_dx).abs()
print("wconv_bwd_data(conv1) vs float64 conv_transpose of ITS OWN input: rel %.2e at %s; #>1e-4: %d"
      % (e.max() / ref_dx.abs().max(), (e == e.max()).nonzero()[0].tolist(), int((e > 1e-4 * ref_dx.abs().max()).sum())))
again = _data(dc1, w1, 64, 300)
print("re-run of wconv_bwd_data on the same input: max |diff| to the first run %.3e" % (again - dx1).abs().max().item())
print("input row (33, 94): %s ..." % dc1[33, 94, :6].tolist())
print("oracle conv1 pre-activation at (33, ch 1, l 95) = %.4e; HIP conv1 block output there = %.4e; dy there: HIP %.4e | d_conv HIP %.4e oracle %.4e"
      % (enc["conv1"].detach()[33, 1, 95].item(), outs[1].detach().cpu()[33, 95, 1].item(), rec[("act", 60, False)][0][33, 95, 1].item(),
         dc1[33, 95, 1].item(), ref_dc1[33, 95, 1].item()))
