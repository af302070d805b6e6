import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
from slu_hip import ops
T, B, H, D, sub = 75, 128, 128, 2, 64
for nsplit in (2, 3):
    torch.manual_seed(T * 7 + B)
    gx = torch.randn(T, B, D * 3 * H, device="cuda")
    wf, bf = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1
    wr, br = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1
    raw, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit)
    off_dev = torch.tensor([5 * 16], dtype=torch.int64, device="cuda")
    for p, offset, odev in ((0.5, 7 * 16 + 3, None), (0.25, 3, off_dev), (0.75, 3, None)):
        keep = ops.dropout_bits(T, B, D * H, p, 1234, offset, odev, sub, gx.device)
        two_f = ops.dropout_pool_fwd(raw, None, p, 1234, offset, "avg", 2, odev, sub, keep_bits=keep)
        one_f = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, False)
        torch.cuda.synchronize()
        ne = (one_f != two_f)
        print("nsplit", nsplit, "p", p, "mismatches", int(ne.sum()), "of", ne.numel())
        if ne.any():
            idx = ne.nonzero()[:6]
            for i in idx:
                t, b, c = [int(v) for v in i]
                print("   (t=%d b=%d c=%d) fused %r two-launch %r  raw %r %r" % (t, b, c, float(one_f[t, b, c]), float(two_f[t, b, c]),
                      float(raw[2 * t, b, c]), float(raw[min(2 * t + 1, T - 1), b, c])))
            print("   by t:", ne.sum(dim=(1, 2)).tolist()[:40])
            print("   by c%4:", [int(ne[:, :, k::4].sum()) for k in range(4)], "dir halves:", int(ne[:, :, :H].sum()), int(ne[:, :, H:].sum()))
