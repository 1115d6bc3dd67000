#!/bin/bash
# Stages the reference's Python package for ONE gpurun call (tools/run_reference_layers.py): the GPU box has no /root/reference.
# _refstage/ is git-ignored (never committed) and removed again by `tools/stage_reference.sh clean`.
cd "$(dirname "$0")/.."
DST="$PWD/_refstage"
if [ "$1" == "clean" ]; then rm -rf "$DST"; exit 0; fi
rm -rf "$DST" && mkdir -p "$DST"
(cd /root/reference && find litegs -name "*.py" -not -path "*/submodules/*" -exec cp --parents {} "$DST" \;)
cp /root/reference/example_train.py /root/reference/example_metrics.py /root/reference/full_eval.py "$DST"/
find "$DST" -name "*.py" | wc -l
