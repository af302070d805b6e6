#!/usr/bin/env python3
"""profiles/inloop_kernel_us.json from a `rocprof_summary.py --by-shape` table of the default bench command: for every
kernel of bench.py's roofline table, the average duration of its launches INSIDE the real pipelined loop, averaged over
the launch shapes of one look-ahead cycle like `avg_us` of the table (each shape once).

    python tools/inloop_json.py profiles/r03_f_default_kernel_stats_by_shape.txt > profiles/inloop_kernel_us.json
"""
import json
import re
import sys

path = sys.argv[1]
rows = []
for line in open(path):
    m = re.match(r"^(.*?)\s+grid=(\d+)x(\d+)x(\d+) wg=(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s", line)
    if m:
        rows.append({"name": m.group(1).strip(), "grid": tuple(int(m.group(k)) for k in (2, 3, 4)), "wg": int(m.group(5)),
                     "calls": int(m.group(6)), "avg_us": float(m.group(8))})


def pick(prefix, n_shapes):
    """the n_shapes most frequently launched shapes of a kernel (the steady-state super-batch / step shapes)"""
    return sorted((r for r in rows if r["name"].startswith(prefix)), key=lambda r: -r["calls"])[:n_shapes]


spec = {  # bench.py table name -> (name prefix in the trace, launch shapes per look-ahead cycle)
    "gru_seq_fwd4_kernel<128>": ("void slu::gru_seq_fwd4_kernel<128>", 1),
    "gemm_f32_kernel<true,true,2>": ("void slu::gemm_f32_kernel<true, true, 2>", 1),
}
for ns in (2, 3):           # the split scheme of the frozen stages: f16x2 (default) / bf16x3
    spec.update({
        # round 4: <H, NS, KI, EPI> — the first layer's launch (fused input projection, plane output), the K = 256 layers
        # with plane output (two launches per cycle, one grid) and the last frozen layer (fp32 output): three instantiations
        "gru_bf_fwd_kernel<128,%d>" % ns: ("void slu::gru_bf_fwd_kernel<128, %d," % ns, 3),
        "dropout_bits_kernel": ("slu::dropout_bits_kernel", 1),
        "gemm_bf_panel96_kernel<%d,8>" % ns: ("void slu::gemm_bf_panel96_kernel<%d, 8>" % ns, 1),
        "gemm_bf_kernel<%d>" % ns: ("void slu::gemm_bf_kernel<%d>" % ns, 2),
        "gemm_bf_panel_kernel<%d,2>" % ns: ("void slu::gemm_bf_panel_kernel<%d, 2>" % ns, 1),
        "dropout_pool_fwd4_kernel<%d>" % ns: ("void slu::dropout_pool_fwd4_kernel<%d>" % ns, 3),
    })
    if any(r["name"].startswith("void slu::wconv_bf_fwd_kernel<") and (", %d, true>" % ns in r["name"] or ", %d, false>" % ns in r["name"])
           for r in rows):
        spec["wconv_bf_fwd_kernel<%d>" % ns] = ("void slu::wconv_bf_fwd_kernel<", 2)
out = {"_comment": "average kernel durations inside the real pipelined loop (rocprofv3 --kernel-trace of `python bench.py`), "
                   "per launch shape, from " + path + "; bench.py divides roofline.frac_isolated by avg_us(in loop) / avg_us(isolated)"}
for key, (prefix, n) in spec.items():
    sel = pick(prefix, n)
    if not sel:
        continue
    top = max(r["calls"] for r in sel)
    weights = [max(1, round(r["calls"] / min(s["calls"] for s in sel))) for r in sel]      # launches per cycle of each shape
    avg = sum(r["avg_us"] * w for r, w in zip(sel, weights)) / sum(weights)
    out[key] = {"avg_us": round(avg, 2), "source": path,
                "shapes": [{"grid": "x".join(map(str, r["grid"])), "calls": r["calls"], "avg_us": r["avg_us"]} for r in sel]}
print(json.dumps(out, indent=1))
