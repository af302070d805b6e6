"""Worker of tests/test_hip_dp.py: one rank of a data-parallel run of the HIP model (or the single-process
reference run).  All ranks share the box's one GPU over gloo (SLU_DIST_BACKEND=gloo SLU_LOCAL_DEVICE=0): RCCL
refuses duplicate devices, everything else of the multi-process step path is the real one.
    python dp_hip_worker.py <out.pt> <world_size>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch  # noqa: E402

from oracle import slu_oracle as O  # noqa: E402  (config holder + seeded initial weights only)
import data  # noqa: E402
import models  # noqa: E402
import training  # noqa: E402
from slu_hip import dp  # noqa: E402

out, world = sys.argv[1], int(sys.argv[2])
rank, ws, local = dp.init_from_env()
assert ws == world
torch.cuda.set_device(local)
work = os.path.dirname(out)
cfg = O.OracleConfig(cnn_N_filt=[16, 12, 12], cnn_len_filt=[101, 5, 5], cnn_stride=[20, 1, 1],
                     phone_rnn_num_hidden=[32, 32], word_rnn_num_hidden=[32, 32], intent_rnn_num_hidden=[32],
                     phone_rnn_drop=[0.0, 0.0], word_rnn_drop=[0.0, 0.0], intent_rnn_drop=[0.0],
                     vocabulary_size=60, num_phonemes=20, pretraining_type=0)
cfg.folder = os.path.join(work, "exp%d_%d" % (world, rank))
os.makedirs(os.path.join(cfg.folder, "training"), exist_ok=True)
cfg.training_lr = 0.003
cfg.starting_unfreezing_index = 1
cfg.unfreezing_type = 0
cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
torch.manual_seed(11)
model = models.Model(cfg)                       # pretraining_type 0: every layer trainable (fp32 + float64 buckets)
trainer = training.Trainer(model, cfg)
g = torch.Generator().manual_seed(5)
batches = []
for _ in range(4):
    x = 0.1 * torch.randn(16, 6000, generator=g)
    y = torch.stack([torch.randint(0, n, (16,), generator=g) for n in cfg.values_per_slot], dim=1)
    n = 16 // ws
    batches.append((x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]))
model.train()
losses = [float(v[0]) for v, _ in trainer._iterate(batches, True, False)]
torch.cuda.synchronize()
torch.save({"losses": losses, "sd": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "payload": trainer.bucket.nbytes(), "dtypes": sorted(str(d) for d in trainer.bucket.flats)}, out)
trainer.close()
if ws > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
