"""Instruction mix of the innermost (largest backward-branch) loop of a kernel in hipcc -S output.
usage: isa_loop_mix.py file.s <substring of the mangled kernel name> ...   (no GPU needed)"""
import collections
import re
import sys


def analyze(txt, sym):
    start = txt.index("\n" + sym + ":")
    end = txt.index("s_endpgm", start)
    lines = [l.strip() for l in txt[start:end].split("\n")]
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"^s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    loop = [l for l in lines[best[0]:best[1] + 1] if l and not l.startswith((".", ";")) and not l.endswith(":")]
    cnt = collections.Counter()
    for l in loop:
        op = l.split()[0]
        if op.startswith("v_mfma"):
            k = "mfma"
        elif op.startswith(("v_exp", "v_rcp", "v_sin", "v_cos", "v_log", "v_sqrt", "v_rsq")):
            k = "trans"
        elif op.startswith("v_cvt"):
            k = "cvt"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("ds_"):
            k = op
        elif op.startswith(("global_", "buffer_", "scratch_")):
            k = "_".join(op.split("_")[:3])
        elif op.startswith("s_waitcnt"):
            k = "s_waitcnt"
        elif op.startswith("s_"):
            k = "salu"
        else:
            k = op
        cnt[k] += 1
    return len(loop), dict(sorted(cnt.items()))


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    syms = re.findall(r"^(_Z\w+):", txt, re.M)
    for pat in sys.argv[2:]:
        for s in syms:
            if pat in s:
                print(s, *analyze(txt, s))
