#!/bin/bash
# tools/multigpu_smoke.sh [N=2] -- run on a node with N >= 2 MI355X: correctness of every sharded mode over RCCL against the single-GPU
# plan, then one short bench.py line per shard mode.  Expected output: tools/multigpu_smoke.expected.txt.
N=${1:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1"
$RUN --master-port 29533 tools/multigpu_smoke.py || exit 1
$RUN --master-port 29534 bench.py --gpus $N --steps 20 --warmup 3 --workload control07 --shard blocks --no-cpu-baseline --no-other-configs
$RUN --master-port 29535 bench.py --gpus $N --steps 5 --warmup 2 --workload maxcut2000 --shard columns --no-cpu-baseline --no-other-configs
$RUN --master-port 29536 bench.py --gpus $N --steps 20 --warmup 3 --workload blockdiag
$RUN --master-port 29537 bench.py --gpus $N --steps 20 --warmup 3 --workload grid:120
$RUN --master-port 29538 bench.py --gpus $N --steps 20 --warmup 3 --workload control07 --shard replicas --no-cpu-baseline --no-other-configs
$RUN --master-port 29539 bench.py --gpus $N --steps 5 --warmup 2 --workload maxcut2000 --shard blockcyclic --no-cpu-baseline
