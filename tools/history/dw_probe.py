"""Diagnostic: slu_wconv_bwd_weight (Sinc geometry) against a float64 reference — overall error, error by tap, and
single-frame probes (one non-zero d_conv frame at a time) to find frames whose contribution is lost or misplaced."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from slu_hip import ops  # noqa: E402

torch.manual_seed(0)
B, T, C, K, S = int(os.environ.get("PB", "64")), int(os.environ.get("PT", "48000")), 80, 401, 80
x = (0.1 * torch.randn(B, T)).cuda()
L = ops.conv_out_len(T, K, S)


def ref_dw(dc):
    xs = F.pad(x.double(), (K // 2, K // 2)).unfold(1, K, S)[:, :L]          # (B, L, K)
    return torch.einsum("blc,blk->ck", dc.double(), xs)


dc = torch.randn(B, L, C, device="cuda") * torch.rand(B, L, 1, device="cuda")
dW, _ = ops.wconv_bwd_weight(dc, x, B, T, 1, C, K, S, False)
ref = ref_dw(dc)
err = (dW.view(C, K).double() - ref).abs()
print("random d_conv: max err %.3e of max|dW| %.3e (rel %.2e); worst tap %d, worst channel %d"
      % (err.max(), ref.abs().max(), err.max() / ref.abs().max(), int(err.max(0)[0].argmax()), int(err.max(1)[0].argmax())))
print("err by tap (max over channels), every 40th:", [float("%.2e" % v) for v in err.max(0)[0][::40].tolist()])
bad = []
for b in sorted({0, 1, B // 2, B - 1}):
    for l in sorted({0, 1, 2, 62, 63, 64, 65, 127, 128, L // 2, L - 66, L - 65, L - 64, L - 2, L - 1}):
        d1 = torch.zeros(B, L, C, device="cuda")
        d1[b, l] = torch.randn(C, device="cuda")
        got, _ = ops.wconv_bwd_weight(d1, x, B, T, 1, C, K, S, False)
        r = ref_dw(d1)
        e = ((got.view(C, K).double() - r).abs().max() / r.abs().max()).item()
        if e > 1e-5:
            bad.append((b, l, e))
print("single-frame probes with rel err > 1e-5:", bad if bad else "none")
