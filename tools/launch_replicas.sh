#!/bin/bash
# One process per GPU, independent replicas, weights broadcast once over RCCL -- the launch the driver
# uses for the N-GPU bench, wrapped:   tools/launch_replicas.sh 8 [bench.py args...]
# N = 1 runs bench.py directly (no rendezvous).
set -e
N=${1:-1}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC only on this pool: RCCL / tensor sharing fail with the legacy mode
if [ "$N" -le 1 ]; then exec python "$ROOT/bench.py" --gpus 1 "$@"; fi
PORT=${MASTER_PORT:-$((20000 + RANDOM % 20000))}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
     "$ROOT/bench.py" --gpus "$N" "$@"
