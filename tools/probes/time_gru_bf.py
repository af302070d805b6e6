"""Times the frozen layers' split-precision recurrence in its product form (gx in, Dropout + avg-pool epilogue, planes out) on
the 160-CU look-ahead partition: T = 300 and 150, 1280 sequences, bf16x3.  SLU_HIP_LIB selects an A/B build (tools/build_alt.sh).
usage: python tools/probes/time_gru_bf.py [label]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
from slu_hip import ops, pipeline
dev = torch.device("cuda", 0)
n = pipeline.cu_split()
st = pipeline.cu_range_stream(dev, n, pipeline.n_compute_units(dev) - n)
B, H, D, ns = 1280, 128, 2, 3
torch.manual_seed(0)
wf, bf = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, device=dev) * 0.1
wr, br = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, device=dev) * 0.1
res = []
for T in (300, 150):
    gx = torch.randn(T, B, D * 3 * H, device=dev)
    keep = ops.dropout_bits(T, B, D * H, 0.5, 1234, 16, None, 64, dev)
    with torch.cuda.stream(st):
        for _ in range(3):
            out = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            out = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True)
        e1.record(st)
    torch.cuda.synchronize()
    res.append("T=%d %.1f us (checksum %d)" % (T, 100.0 * e0.elapsed_time(e1), int(out.planes.view(torch.int16).to(torch.int64).sum())))
print("%-70s %s" % (sys.argv[1] if len(sys.argv) > 1 else os.environ.get("SLU_HIP_LIB", "product build"), " | ".join(res)))
