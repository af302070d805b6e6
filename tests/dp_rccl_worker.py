"""Worker of tests/test_hip_multigpu.py: one rank of a data-parallel run (one process per GPU — or several ranks on one
GPU —, control plane gloo, gradient collective by SLU_COMM: the hand-written IPC all-reduce, RCCL through the C ABI, or
torch.distributed's own), or the single-process reference run on the full batches.
    python dp_rccl_worker.py <out.pt> <world_size>
Two epochs of Trainer.train on a frozen pre-trained encoder with gradual unfreezing (unfreezing_type 2): the set of
trainable parameters — and with it the flat gradient bucket — changes between the epochs."""
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
                     vocabulary_size=60, num_phonemes=20, pretraining_type=2)
cfg.folder = os.path.join(work, "exp%d_%d" % (world, rank))
os.makedirs(os.path.join(cfg.folder, "training"), exist_ok=True)
os.makedirs(os.path.join(cfg.folder, "pretraining"), exist_ok=True)
cfg.training_lr = 0.003
cfg.unfreezing_type = 2
cfg.starting_unfreezing_index = 1
cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
torch.manual_seed(10)
torch.save(O.init_pretrained_state_dict(cfg), os.path.join(cfg.folder, "pretraining", "model_state.pth"))
torch.manual_seed(11)
model = models.Model(cfg)
trainer = training.Trainer(model, cfg)
GLOBAL = 16
ds = data.SyntheticSLUDataset(4, GLOBAL, 6000, cfg.values_per_slot, seed=5)         # the same global batches on every rank
n = GLOBAL // ws
ds.batches = [(x[rank * n:(rank + 1) * n].contiguous(), y[rank * n:(rank + 1) * n].contiguous()) for x, y in ds.batches]
ds.loader = data._SyntheticLoader(ds.batches)
epochs, payloads, live, collective = [], [], [], []
_reset = trainer.bucket.reset


def reset_and_record():             # Trainer.train resets the bucket when unfreeze_one_layer() changed the trainable set
    payloads.append(trainer.bucket.nbytes())
    _reset()


trainer.bucket.reset = reset_and_record
for _ in range(3):
    acc, loss = trainer.train(ds, print_interval=1000)
    epochs.append((acc, loss))
    live.append(sum(1 for p in model.parameters() if p.requires_grad))
    collective.append(trainer.graph_stats().get("collective"))     # where the captured steps' all-reduce sits
assert len(payloads) == 3
torch.cuda.synchronize()
torch.save({"epochs": epochs, "payloads": payloads, "live": live, "collective": collective,
            "sd": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "comm": type(trainer.bucket.comm).__name__ if trainer.bucket.comm is not None else "torch.distributed",
            "ipc_status": trainer.bucket.comm.status() if hasattr(trainer.bucket.comm, "status") else 0,
            "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"}, out)
trainer.close()
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
