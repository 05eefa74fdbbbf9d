#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/gpu_box_probe.py c2 c5 c4 c1 2>&1 | grep BOX
timeout ${T:-900} python -m pytest tests -m gpu -x -q ${K:+-k "$K"} -p no:cacheprovider 2>&1 | tail -${TAIL:-15}
