#!/usr/bin/env python3
"""The default bench loop fed from pinned HOST batches (H2D inside the clock) for several placements of the copies:
SLU_COPY_CUS = 0 (on the slot's stream, serialised with its kernels), 16 / 32 / 160 (a copy stream confined to that many
CUs of the look-ahead partition), 999 (an unmasked copy stream); "default" leaves the variable alone.  An argument
"<cus>:<slots>" also sets SLU_LOOKAHEAD_SLOTS (super-batches in flight: 2 = the copy of group g + 2 starts when group g's
steps end, 3 = one group earlier, beside group g + 1's encoder)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "SLU_PROBE_CHILD" not in os.environ:
    for arg in sys.argv[1:] or ["0", "16", "32", "160", "999"]:
        cus, _, slots = arg.partition(":")
        env = dict(os.environ, SLU_PROBE_CHILD="1")
        if cus != "default":
            env["SLU_COPY_CUS"] = cus
        if slots:
            env["SLU_LOOKAHEAD_SLOTS"] = slots
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print("SLU_COPY_CUS=%s SLU_LOOKAHEAD_SLOTS=%s: %s" % (cus, slots or "default", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

torch.cuda.set_device(0)
config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 4)
dev = torch.device("cuda", 0)
batches = [tuple(t.to(dev) for t in b) for b in train_ds.loader]
model.train()
for _ in range(3):
    bench.run_steps(model, trainer, batches, 256)
torch.cuda.synchronize()
print(json.dumps(bench.host_inputs_point(model, trainer, batches, 256, False)))
