import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
import bench
n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 40
config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 4)
dev = next(model.parameters()).device
model.train()
from slu_hip import ops, pipeline
n = model.frozen_prefix_len()
slot = pipeline.PrefixSlot(dev)
model.pretrained_model.warm_weight_caches()
base = [torch.randn(64, 48000, device=dev) * 0.1 for _ in range(4)]
xs = [base[i % 4] for i in range(n_b)]
for it in range(4):
    print("run", it, "graphs", {k[:2]: (v is not None) for k, v in slot.graphs.items()}, flush=True)
    f, done, g = slot.run(model, xs, n, 1 + it * n_b, True)
    torch.cuda.synchronize()
    print("   ok", tuple(f.shape), float(f.abs().max()), "words", slot.words[:2].tolist(), slot.words[n_b - 1:n_b + 1].tolist(), slot.words[-1].item(), flush=True)
