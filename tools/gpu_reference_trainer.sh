#!/bin/bash
# usage (after tools/stage_reference.sh): tools/gpu_reference_trainer.sh [ITER] -> profiles-ready log gpurun_out/reference_trainer.log
# 1. a synthetic COLMAP scene (tools/make_colmap_scene.py), 2. the reference's UNMODIFIED example_train.py + example_metrics.py from
# _refstage/ with litegs_fused / fused_ssim / simple_knn = this repository and compat/ for plyfile, cv2, torchmetrics,
# 3. this repository's train.py (litegs_amd.training.start, native executor) on the same files, 4. both evaluated by the reference's
# example_metrics.py.
ITER=${1:-3000}
R=$GRAFT_REPO_ROOT
LOG=$R/gpurun_out/reference_trainer.log
mkdir -p $R/gpurun_out
SCENE=/tmp/scene; OUT_REF=/tmp/out_ref; OUT_OURS=/tmp/out_ours
{
echo "== scene"; timeout -s KILL 120 python $R/tools/make_colmap_scene.py --out $SCENE --points 30000 --frames 40 --width 640 --height 400 --focal 560
COMMON="-s $SCENE --eval -r 1 --iterations $ITER --target_primitives 60000"
echo "== reference example_train.py (unmodified, from _refstage/) on litegs_fused = this repository"
cd $R/_refstage
T0=$(date +%s%N)
PYTHONPATH=$R:$R/compat timeout -s KILL 600 python example_train.py $COMMON -m $OUT_REF 2>&1 | grep -v "it/s\]" | tail -25
T1=$(date +%s%N)
echo "reference trainer wall: $(( (T1 - T0) / 1000000 )) ms"
PYTHONPATH=$R:$R/compat timeout -s KILL 300 python example_metrics.py $COMMON -m $OUT_REF 2>&1 | tail -12
echo "== litegs_amd train.py (native executor) on the same scene"
cd $R
T0=$(date +%s%N)
timeout -s KILL 600 python train.py $COMMON -m $OUT_OURS 2>&1 | tail -8
T1=$(date +%s%N)
echo "litegs_amd trainer wall: $(( (T1 - T0) / 1000000 )) ms"
cd $R/_refstage
PYTHONPATH=$R:$R/compat timeout -s KILL 300 python example_metrics.py $COMMON -m $OUT_OURS 2>&1 | tail -12
ls -la $OUT_REF/point_cloud/finish $OUT_OURS/point_cloud/finish
} > $LOG 2>&1
tail -60 $LOG
