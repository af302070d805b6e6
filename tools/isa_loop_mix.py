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
        mm = re.match(r"^s_c?branch\w* (\.LBB\d+_\d+)", l)
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


def loop_lines(txt, sym):
    """The loop's instructions in order (for reading the schedule: --dump)."""
    start = txt.index("\n" + sym + ":")
    end = txt.index("s_endpgm", start)
    lines = [l.strip() for l in txt[start:end].split("\n")]
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"^(\.LBB\d+_\d+):", l)}
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"^s_c?branch\w* (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return [l for l in lines[best[0]:best[1] + 1] if l and not l.startswith((".", ";"))]


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        txt = open(sys.argv[2]).read()
        syms = re.findall(r"^(_Z\w+):", txt, re.M)
        for s_ in syms:
            if sys.argv[3] in s_:
                print("\n".join(loop_lines(txt, s_)))
                break
        sys.exit(0)
    txt = open(sys.argv[1]).read()
    syms = re.findall(r"^(_Z\w+):", txt, re.M)
    for pat in sys.argv[2:]:
        for s in syms:
            if pat in s:
                print(s, *analyze(txt, s))
