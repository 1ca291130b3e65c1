#!/bin/bash
# scratch/multirank_dryrun.sh -- bench.py with TWO (or N=$RANKS) ranks on ONE GPU (collectives over gloo through host memory):
# exercises the multi-rank control flow (pre-warm broadcast, per-step gather, strong-scaling shards incl. ragged and empty
# ones) that RCCL runs on a real node.  The numbers of these runs mean nothing.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
export TIMG_DIST_BACKEND=gloo
n=${RANKS:-2}
run() { timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $1 bench.py --gpus $n "${@:2}" 2>gpurun_out/dryrun_$1.err | tail -1 | cut -c1-200; grep -m1 "Error" gpurun_out/dryrun_$1.err; }
run 29541 --steps 3 --warmup 1 --cpu-seconds 2
run 29542 --config c4 --frames 80 --steps 2 --warmup 1
run 29543 --config c5 --frames 9 --steps 2 --warmup 1
run 29544 --config c3 --steps 2 --warmup 1
RANKSAVE=$n; n=4
run 29545 --config c5 --frames 3 --steps 2 --warmup 1
run 29546 --config c4 --frames 10 --chunk 2 --steps 2 --warmup 1
