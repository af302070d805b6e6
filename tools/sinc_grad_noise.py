"""Diagnostic: where does the float64 Sinc-parameter gradient of a full-size ASR step deviate?  Takes the fp32 oracle's
gradient at the Sinc convolution's output, recomputes the filter gradient and the parameter gradients in float64
("exact" given that upstream gradient) and compares both the oracle's own fp32 result and the HIP kernels' with it."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import slu_oracle as O  # noqa: E402
import models  # noqa: E402


def sinc_filters64(filt_b1, filt_band, N, fs):
    freq_scale = fs * 1.0
    half = int((N - 1) / 2)
    t_right = (torch.linspace(1, (N - 1) / 2, steps=half) / fs).double()
    beg = torch.abs(filt_b1) + 50.0 / freq_scale
    end = beg + (torch.abs(filt_band) + 50.0 / freq_scale)
    n = torch.linspace(0, N, steps=N)
    window = (0.54 - 0.46 * torch.cos(2 * math.pi * n / N)).double()

    def low_pass(f):
        arg = 2 * math.pi * (f * freq_scale).unsqueeze(1) * t_right.unsqueeze(0)
        y_right = torch.sin(arg) / arg
        y = torch.cat([torch.flip(y_right, dims=[1]), torch.ones(y_right.shape[0], 1, dtype=torch.float64), y_right], dim=1)
        return 2 * f.unsqueeze(1) * y
    bp = low_pass(end) - low_pass(beg)
    bp = bp / bp.max(dim=1, keepdim=True)[0]
    return bp * window.unsqueeze(0)


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
from slu_hip import ops as _ops
rec = {}
_w, _s = _ops.wconv_bwd_weight, _ops.sinc_filters_bwd


def rec_w(d_conv, xin, B_, l_in, c_in, c_out, k_t, stride, want_bias, out=None):
    r = _w(d_conv, xin, B_, l_in, c_in, c_out, k_t, stride, want_bias, out)
    if c_in == 1:
        rec["d_conv"], rec["dW"] = d_conv.detach().clone(), r[0].detach().clone()
    return r


def rec_s(b1_, band_, dF, filt_dim, fs):
    r = _s(b1_, band_, dF, filt_dim, fs)
    rec["dF"], rec["db1"] = dF.detach().clone(), r[0].detach().clone()
    return r


_ops.wconv_bwd_weight, _ops.sinc_filters_bwd = rec_w, rec_s
pm.train()
pl, wl, pa, wa = pm(x, yp, yw)
(pl + wl).backward()
torch.cuda.synchronize()
models.set_dropout_masks(None)
torch.set_num_threads(64)
# oracle with the conv0 output retained
st_holder = {}
orig = O.sinc_layer


def hooked(*a, **k):
    out = orig(*a, **k)
    out.retain_grad()
    st_holder["conv0"] = out
    return out


O.sinc_layer = hooked
rpl, rwl, _, _ = O.asr_forward(sd, x, yp, yw, cfg, masks, explicit_gru=False)
(rpl + rwl).backward()
d_conv = st_holder["conv0"].grad.double()                       # (B, 80, L)
b1, band = sd["phoneme_layers.0.filt_b1"], sd["phoneme_layers.0.filt_band"]
q1, q2 = b1.detach().clone().requires_grad_(), band.detach().clone().requires_grad_()
f64 = sinc_filters64(q1, q2, 401, 16000)
f64.retain_grad()
out64 = F.conv1d(x.double().unsqueeze(1), f64.view(80, 1, 401), stride=80, padding=200)
(out64 * d_conv).sum().backward()
for name, exact, orc, hip in (("filt_b1", q1.grad, b1.grad, pm.phoneme_layers[0].filt_b1.grad.cpu()),
                              ("filt_band", q2.grad, band.grad, pm.phoneme_layers[0].filt_band.grad.cpu())):
    s = exact.abs().max().item()
    print("%s: |grad| max %.3e; oracle fp32 vs float64-exact %.2e; HIP vs float64-exact %.2e; HIP vs oracle %.2e"
          % (name, s, (orc - exact).abs().max().item() / s, (hip - exact).abs().max().item() / s, (hip - orc).abs().max().item() / s))

dc_gpu = rec["d_conv"].cpu().double().permute(0, 2, 1)            # (B, L, C) -> (B, C, L)
print("d_conv: HIP vs oracle rel %.2e (max |d_conv| %.3e)" % ((dc_gpu - d_conv).abs().max().item() / d_conv.abs().max().item(), d_conv.abs().max().item()))
dW64 = f64.grad
print("dW: HIP vs float64 (from the oracle's d_conv) rel %.2e" % ((rec["dW"].cpu().double().view(80, 401) - dW64).abs().max().item() / dW64.abs().max().item()))
# the HIP filter-parameter backward on the float64 dW (rounded to fp32) alone
db1_k, dband_k = _s(pm.phoneme_layers[0].filt_b1.detach(), pm.phoneme_layers[0].filt_band.detach(), dW64.float().cuda().contiguous(), 401, 16000)
print("sinc_filters_bwd kernel on the exact dW: d_b1 rel %.2e, d_band rel %.2e"
      % ((db1_k.cpu() - q1.grad).abs().max().item() / q1.grad.abs().max().item(), (dband_k.cpu() - q2.grad).abs().max().item() / q2.grad.abs().max().item()))
e = (dc_gpu - d_conv).abs() / d_conv.abs().max()
bad = (e > 1e-4).nonzero()
print("elements of d_conv off by > 1e-4 of max: %d of %d" % (bad.shape[0], e.numel()))
if bad.shape[0]:
    print("  by frame l (histogram of l, top 10):", torch.bincount(bad[:, 2], minlength=d_conv.shape[2]).topk(10))
    print("  by channel (top 5):", torch.bincount(bad[:, 1], minlength=80).topk(5))
    for b_, c_, l_ in bad[:6].tolist():
        pair = st_holder["conv0"][b_, c_, (l_ // 2) * 2:(l_ // 2) * 2 + 2].detach()
        print("  (b=%d c=%d l=%d): HIP %.4e oracle %.4e; conv outputs of the pool pair: %s" % (b_, c_, l_, dc_gpu[b_, c_, l_], d_conv[b_, c_, l_], pair.tolist()))
