#!/bin/bash
# determinism probe of the jellyfish script under a concurrent GPU load:  bash tools/jelly_det.sh [ENV=VAL ...]
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
A="inference/inference_2d_jellyfish.py --synthetic True --batch_size 3 --num_batches 1 --frames 4 --image_size 64 --timesteps 2"
run1() { python $A --inference_result_path $1 > $1.log 2>&1; }
run2() { DPC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 $A --inference_result_path $1 > $1.log 2>&1; }
rm -rf /tmp/jd; mkdir -p /tmp/jd
run1 /tmp/jd/a1
python bench.py --no-extras --steps 120 > /tmp/jd/load.log 2>&1 &
LOAD=$!
sleep 25
for i in 1 2 3 4 5 6 7 8 9 10; do if [ "${RANKS:-2}" = 1 ]; then run1 /tmp/jd/b$i; else run2 /tmp/jd/b$i $((29620+i)); fi; done
kill $LOAD 2>/dev/null; wait $LOAD 2>/dev/null
python - <<'PY'
import numpy as np
names=["b%d" % i for i in range(1, 11)]
def load(n): return [np.load(f"/tmp/jd/{n}/{sub}/{i}.npy") for sub in ("thetas","states") for i in range(3)]
ref=load("a1")
for n in names:
    print(n, max(np.abs(p-q).max() for p,q in zip(ref,load(n))))
PY
