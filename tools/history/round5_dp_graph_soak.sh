#!/bin/bash
# Round-5 soak: N consecutive one-rank data-parallel training runs (SLU_DP_SINGLE=1, control plane gloo) with the gradient
# all-reduce captured as a node of the step's hipGraph — the configuration round 4 had to leave off because torch's NCCL
# watchdog aborted 2 of 18 such processes.  usage: tools/round5_dp_graph_soak.sh <runs per data plane> <parallel>
N=${1:-25}; P=${2:-5}
O=gpurun_out/r05_soak2; rm -rf $O; mkdir -p $O
for comm in rccl ipc; do
  ok=0; bad=0
  for ((i = 0; i < N; i += P)); do
    pids=()
    for ((j = i; j < i + P && j < N; j++)); do
      port=$((29600 + j))
      mkdir -p $O/${comm}_$j          # a directory per run: the worker writes its experiment folder beside its output
      RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port SLU_DP_SINGLE=1 SLU_COMM=$comm \
        python tests/dp_rccl_worker.py $O/${comm}_$j/out.pt 1 > $O/${comm}_$j.log 2>&1 &
      pids+=($!)
    done
    for pid in "${pids[@]}"; do if wait $pid; then ok=$((ok + 1)); else bad=$((bad + 1)); fi; done
  done
  nodes=$(python - <<PY
import glob, torch
n = 0
for f in glob.glob("$O/${comm}_*/out.pt"):
    d = torch.load(f)
    n += all(c == "a node of the step's hipGraph" for c in d["collective"])
print(n)
PY
)
  echo "data plane $comm: $ok of $N processes finished cleanly, $bad aborted; the collective was a graph node in all three epochs of $nodes runs"
done
