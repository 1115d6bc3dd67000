#!/bin/bash
# One parameterised GPU-box script (replaces the per-session tools/gpu_rNN_*.sh of earlier rounds).
#   usage: tools/gpu_session.sh TAG STAGE [STAGE ...]      (run from the repository root on the GPU box; output under gpurun_out/)
# stages:
#   tests        the whole -m gpu suite (flip counts collected)           tests:<pytest args>  a targeted run
#   stochastic:<reps>  the noise-bounded tests repeated in one session (calibration of their bounds)
#   smoke        __graft_entry__.smoke()
#   benchdriver  bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)
#   benchab:<specs>  fresh / trained step time under lg_set_tuning variants
#   bench        the default bench line                                     bench10m / bench500k  the other BASELINE sizes
#   benchdp2     bench.py --gpus 2 with both ranks on this GPU over gloo (control-flow check of the N>1 path)
#   trace        rocprofv3 --kernel-trace --stats of the default bench -> step timeline + kernel stats
#   hbm          per-kernel HBM traffic (PMC FETCH_SIZE / WRITE_SIZE + durations) -> GB/s and fraction of 8 TB/s, fresh and training state
#   sq           SQ counters (VALU / SALU / LDS bank conflicts) of the bench workload, fresh and trained state
#   poison       short scenarios plain vs with poisoned per-frame buffers + validators (tools/poison_probe.py): must be indistinguishable
#   hunt:<runs>  the same loop in the configuration that faulted, with poisoned buffers + validators + breadcrumbs (see the stage)
#   conv:<runs>  tests/convergence_3m.py with <runs> executor trainers in ONE process (operator curve taken from profiles/), allocator
#                snapshots on, late-phase state saved to /tmp/late.pt at epoch 120 by the first trainer
#   bwdab        tools/bwd_ab.py: blend backward variants (lg_set_tuning key 5: generic / fast / splat-parallel) on the bench workload
#   late         tools/late_phase.py ab + parity on /tmp/late.pt            latetrace  kernel trace + LDS counters of late-phase steps
#   dpglue       tools/dp_glue_bench.py on the trained-state cloud
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for STAGE in "$@"; do
  echo "=== $STAGE"
  case $STAGE in
    tests)
      rm -f gpurun_out/flip_counts.jsonl
      LITEGS_COLLECT_FLIPS=1 timeout -s KILL 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_$TAG.log 2>&1
      grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -1; grep -E "^FAILED|^ERROR" gpurun_out/pytest_$TAG.log | head -20; grep -E "^E  " gpurun_out/pytest_$TAG.log | head -20 ;;
    tests:*)
      timeout -s KILL 600 python -m pytest ${STAGE#tests:} -m gpu -x -q > gpurun_out/pytest_quick_$TAG.log 2>&1
      grep -E "passed|failed|error" gpurun_out/pytest_quick_$TAG.log | tail -2; grep -E "^E  " gpurun_out/pytest_quick_$TAG.log | head -20 ;;
    stochastic:*)    # every noise-bounded test (pytest.mark.stochastic) <reps> times in this one session; the statistic each bounds goes to noise_stats.jsonl
      REPS=${STAGE#stochastic:}
      rm -f gpurun_out/noise_stats.jsonl
      for i in $(seq 1 $REPS); do
        timeout -s KILL 600 python -m pytest tests -m "gpu and stochastic" -q -p no:cacheprovider > gpurun_out/pytest_stochastic_${TAG}_$i.log 2>&1
        echo "rep $i: $(grep -E 'passed|failed' gpurun_out/pytest_stochastic_${TAG}_$i.log | tail -1)"; grep -E "^FAILED|^E  " gpurun_out/pytest_stochastic_${TAG}_$i.log | head -8
      done
      cp gpurun_out/noise_stats.jsonl gpurun_out/noise_stats_$TAG.jsonl 2>/dev/null ;;
    smoke)
      timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -1 gpurun_out/smoke_$TAG.log ;;
    bench)
      timeout -s KILL 600 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-900 ;;
    benchab:*)       # the bench's fresh and noise-trained states under lg_set_tuning variants: benchab:<key=value[+key=value]>[,<variant>...] ("-" = defaults)
      for SPEC in $(echo ${STAGE#benchab:} | tr ',' ' '); do
        T=$(echo $SPEC | tr '+' ','); [ "$T" = "-" ] && T=""
        LITEGS_TUNING=$T timeout -s KILL 300 python bench.py --no-cpu-baseline --no-operator-path --no-pmc --no-training-state > gpurun_out/benchab_${TAG}.log 2>&1
        python - <<PY
import json
d = json.loads(open("gpurun_out/benchab_${TAG}.log").read().strip().splitlines()[-1])
print(f"tuning '$T': fresh {d['ms_per_step']:.4f} ms (p50 {d['ms_p50']:.4f}, excl. replays {d['ms_per_step_excl_replays']:.4f})  trained {d['steady_state']['ms_per_step']:.4f} ms (p50 {d['steady_state']['ms_p50']:.4f})  sanitised {d.get('sanitised')}")
PY
      done ;;
    benchdriver)     # the command line the driver records (BENCH_rNN.json)
      timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$TAG.log 2>&1; tail -1 gpurun_out/bench_driver_$TAG.log | cut -c1-700 ;;
    bench10m)
      timeout -s KILL 300 python bench.py --config 10m_1600x1200 --frames 4 --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_10m_$TAG.log 2>&1; tail -1 gpurun_out/bench_10m_$TAG.log | cut -c1-300 ;;
    bench500k)
      timeout -s KILL 200 python bench.py --config 500k_1080p --no-cpu-baseline --no-operator-path --no-pmc --soak-steps 0 > gpurun_out/bench_500k_$TAG.log 2>&1; tail -1 gpurun_out/bench_500k_$TAG.log | cut -c1-300 ;;
    benchdp2)
      LITEGS_BENCH_ONE_GPU=1 LITEGS_HANG_DUMP=${LITEGS_HANG_DUMP:-150} timeout -s KILL 200 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline --no-operator-path --no-pmc > gpurun_out/bench_dp2_$TAG.log 2>&1; grep '^{' gpurun_out/bench_dp2_$TAG.log | cut -c1-600; tail -3 gpurun_out/bench_dp2_$TAG.log | cut -c1-300 ;;
    trace)
      (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --no-cpu-baseline --no-operator-path --no-pmc --no-training-state > $R/gpurun_out/rocprof_$TAG.log 2>&1)
      T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
      python tools/profile_r03.py $T > gpurun_out/step_timeline_$TAG.md 2> gpurun_out/step_timeline_$TAG.err; sed -n 3,32p gpurun_out/step_timeline_$TAG.md; tail -3 gpurun_out/step_timeline_$TAG.err
      S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$TAG.csv
      rm -rf gpurun_out/prof_$TAG ;;
    sq)
      timeout -s KILL 300 python tools/soaked_probe.py save /tmp/soaked.npz 1000 > gpurun_out/soaked_save_$TAG.log 2>&1; tail -1 gpurun_out/soaked_save_$TAG.log
      for STATE in fresh trained; do
        if [ $STATE = fresh ]; then CMD="python $R/bench.py --pmc-child --steps 8 --warmup 0"; else CMD="python $R/tools/soaked_probe.py run /tmp/soaked.npz 8"; fi
        for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
          NAME=$(echo $SET | cut -d' ' -f1)
          (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${STATE}_${NAME}_$TAG -o pmc -- $CMD > $R/gpurun_out/pmc_${STATE}_${NAME}_$TAG.log 2>&1)
          C=$(find gpurun_out/pmc_${STATE}_${NAME}_$TAG -name "*counter_collection.csv" | head -1)
          python tools/sq_summary.py $C "$STATE cloud, $NAME set" > gpurun_out/sq_${STATE}_${NAME}_$TAG.md 2>> gpurun_out/sq_$TAG.err; sed -n 5,16p gpurun_out/sq_${STATE}_${NAME}_$TAG.md
          rm -rf gpurun_out/pmc_${STATE}_${NAME}_$TAG
        done
      done ;;
    hbm)             # per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE passes with the kernel trace) in the fresh and the training state
      for STATE in fresh training; do
        if [ $STATE = fresh ]; then CMD="python $R/bench.py --pmc-child --steps 8 --warmup 0"; else CMD="python $R/tools/late_phase.py trace recipe default"; fi
        for C in FETCH_SIZE WRITE_SIZE; do
          (cd /tmp && LATE_TRACE_STEPS=8 timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/hbm_${STATE}_${C}_$TAG -o pmc -- $CMD > $R/gpurun_out/hbm_${STATE}_${C}_$TAG.log 2>&1)
        done
        FC=$(find gpurun_out/hbm_${STATE}_FETCH_SIZE_$TAG -name "*counter_collection.csv" | head -1); FT=$(find gpurun_out/hbm_${STATE}_FETCH_SIZE_$TAG -name "*kernel_trace.csv" | head -1)
        WC=$(find gpurun_out/hbm_${STATE}_WRITE_SIZE_$TAG -name "*counter_collection.csv" | head -1)
        LASTK=0; [ $STATE = training ] && LASTK=8
        python tools/hbm_summary.py $FC "$FT" $WC "$STATE state (3 M Gaussians @1080p)" $LASTK > gpurun_out/hbm_${STATE}_$TAG.md 2>> gpurun_out/hbm_$TAG.err; head -30 gpurun_out/hbm_${STATE}_$TAG.md | cut -c1-200
        rm -rf gpurun_out/hbm_${STATE}_FETCH_SIZE_$TAG gpurun_out/hbm_${STATE}_WRITE_SIZE_$TAG
      done ;;
    convown:*)       # whole-run A/B: the executor keeps its own schedule + depth-bound culling after the first statistics epoch
      RUNS=${STAGE#convown:}
      LITEGS_CONV_SNAPSHOT=/tmp/conv_snap_own_$TAG.jsonl LITEGS_CONV_PARTIAL=gpurun_out/convergence_3m_own_${TAG}_partial.json \
        LITEGS_CONV_SKIP_OPERATOR=profiles/r03_convergence_3m.json timeout -s KILL 2400 python -X faulthandler tests/convergence_3m.py --runs $RUNS --set stat_schedule_always=false$CONV_OWN_EXTRA \
        --out gpurun_out/convergence_3m_own_$TAG.md > gpurun_out/convergence_3m_own_$TAG.log 2>&1
      echo "exit $?"; grep -E "executor:|fault|Error|error" gpurun_out/convergence_3m_own_$TAG.log | head -20 ;;
    poison)          # tools/poison_probe.py through its test: plain vs poisoned-buffer runs of short scenarios; findings print as XFAIL lines
      LITEGS_POISON_PROBE=1 timeout -s KILL 600 python -m pytest tests/test_zz_gpu_poison.py -m gpu -q > gpurun_out/poison_$TAG.log 2>&1
      tail -5 gpurun_out/poison_$TAG.log | cut -c1-3000 ;;
    hunt:*)          # the fault hunt of profiles/r04_fault_attribution.md: <runs> trainers in one process with size predictions kept across
                     # densifications (the configuration that faulted), per-frame buffers poisoned (a word no kernel wrote is out of range
                     # wherever it is used), every table validator on, launch breadcrumbs on.  A validator report names the stage.
      RUNS=${STAGE#hunt:}
      LITEGS_GUARD_ALLOC=poison LITEGS_VALIDATE_TABLES=1 LITEGS_CRUMBS=1 LITEGS_CONV_VERBOSE=1 LITEGS_CONV_PARTIAL=gpurun_out/hunt_${TAG}_partial.json \
        LITEGS_CONV_SKIP_OPERATOR=profiles/r03_convergence_3m.json timeout -s KILL 2400 python -X faulthandler tests/convergence_3m.py --runs $RUNS \
        --set keep_size_predictions=true $CONV_ARGS --out gpurun_out/hunt_$TAG.md > gpurun_out/hunt_$TAG.log 2>&1
      echo "exit $?"; grep -a -E "executor:|fault|Error|error|crumbs|validate" gpurun_out/hunt_$TAG.log | cut -c1-600 | head -20; grep -a "epoch .* done" gpurun_out/hunt_$TAG.log | tail -2 ;;
    conv:*)
      RUNS=${STAGE#conv:}
      rm -f /tmp/late.pt /tmp/mid.pt gpurun_out/conv_snap_$TAG.jsonl
      LITEGS_CONV_SAVE=/tmp/late.pt:120,/tmp/mid.pt:27 LITEGS_CONV_SNAPSHOT=/tmp/conv_snap_$TAG.jsonl LITEGS_CONV_PARTIAL=gpurun_out/convergence_3m_${TAG}_partial.json \
        LITEGS_CONV_SKIP_OPERATOR=profiles/r03_convergence_3m.json timeout -s KILL 2400 python -X faulthandler tests/convergence_3m.py --runs $RUNS $CONV_ARGS \
        --out gpurun_out/convergence_3m_$TAG.md > gpurun_out/convergence_3m_$TAG.log 2>&1
      echo "exit $?"; grep -E "executor:|fault|Error|error" gpurun_out/convergence_3m_$TAG.log | head -20
      tail -c 20000000 /tmp/conv_snap_$TAG.jsonl | tail -n 3 > gpurun_out/conv_snap_tail_$TAG.jsonl ;;
    late)
      timeout -s KILL 900 python tools/late_phase.py ab /tmp/late.pt > gpurun_out/late_ab_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/late_ab_$TAG.log | tail -14
      if [ -f /tmp/mid.pt ]; then timeout -s KILL 600 python tools/late_phase.py ab /tmp/mid.pt default,global,own_schedule > gpurun_out/mid_ab_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/mid_ab_$TAG.log | tail -8; fi
      timeout -s KILL 600 python tools/late_phase.py parity /tmp/late.pt > gpurun_out/late_parity_$TAG.log 2>&1; grep -E "parity|PARITY" gpurun_out/late_parity_$TAG.log | tail -12 ;;
    critical)        # blend kernels: whole frame against the heaviest tiles alone (is the launch a critical path?)
      timeout -s KILL 300 python tools/late_phase.py critical recipe > gpurun_out/critical_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/critical_$TAG.log | tail -12 ;;
    projab)          # fused projection: SH loads behind / in front of the tile walk (lg_set_tuning key 12), fresh and after 1000 steps
      timeout -s KILL 300 python tools/proj_ab.py 3m_1080p 1000 > gpurun_out/proj_ab_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/proj_ab_$TAG.log | tail -14 ;;
    bwdab)           # blend backward in situ on the bench workload: generic / fast / splat-parallel kernels, fresh and after 1000 steps
      timeout -s KILL 400 python tools/bwd_ab.py 3m_1080p 1000 > gpurun_out/bwd_ab_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/bwd_ab_$TAG.log | tail -24 ;;
    recipeab:*)      # step / forward time of chosen variants on the training_state recipe
      timeout -s KILL 600 python tools/late_phase.py ab recipe ${STAGE#recipeab:} > gpurun_out/recipe_ab_$TAG.log 2>&1; grep -v "amdgpu.ids" gpurun_out/recipe_ab_$TAG.log | tail -12 ;;
    recipetrace:*)   # kernel trace of one named variant on the recipe state
      V=${STAGE#recipetrace:}
      (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rec_${V}_$TAG -o trace -- python $R/tools/late_phase.py trace recipe $V > $R/gpurun_out/recipe_trace_${V}_$TAG.log 2>&1)
      T=$(find gpurun_out/prof_rec_${V}_$TAG -name "*kernel_trace.csv" | head -1)
      python tools/late_timeline.py $T 8 > gpurun_out/recipe_timeline_${V}_$TAG.md 2>> gpurun_out/recipe_timeline_$TAG.err; grep -v "rocprim\|at::native" gpurun_out/recipe_timeline_${V}_$TAG.md | head -32
      rm -rf gpurun_out/prof_rec_${V}_$TAG ;;
    recipesq:*)      # SQ counter passes (issue shares, VALU / SALU / LDS instruction counts) of one variant on the training_state recipe
      V=${STAGE#recipesq:}
      I=0
      for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
        I=$((I+1))
        (cd /tmp && LATE_TRACE_STEPS=4 timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_rec_$TAG -o pmc -- python $R/tools/late_phase.py trace recipe $V > $R/gpurun_out/pmc_rec_${I}_$TAG.log 2>&1)
        C=$(find gpurun_out/pmc_rec_$TAG -name "*counter_collection.csv" | head -1)
        python tools/sq_summary.py $C "training_state recipe, variant $V, set $I" > gpurun_out/sq_recipe_${V}_${I}_$TAG.md 2>> gpurun_out/recipe_timeline_$TAG.err; sed -n 5,24p gpurun_out/sq_recipe_${V}_${I}_$TAG.md | cut -c1-330
        rm -rf gpurun_out/pmc_rec_$TAG
      done ;;
    recipetrace)     # the same traces on the bench's training_state recipe (no stored cloud needed: 1.6 s to reach the state)
      for V in default stat_epoch; do
        (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rec_${V}_$TAG -o trace -- python $R/tools/late_phase.py trace recipe $V > $R/gpurun_out/recipe_trace_${V}_$TAG.log 2>&1)
        T=$(find gpurun_out/prof_rec_${V}_$TAG -name "*kernel_trace.csv" | head -1)
        python tools/late_timeline.py $T 8 > gpurun_out/recipe_timeline_${V}_$TAG.md 2>> gpurun_out/recipe_timeline_$TAG.err; grep -v "rocprim\|at::native" gpurun_out/recipe_timeline_${V}_$TAG.md | head -48
        rm -rf gpurun_out/prof_rec_${V}_$TAG
      done ;;
    latetrace)
      for V in default stat_epoch; do
        (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_late_${V}_$TAG -o trace -- python $R/tools/late_phase.py trace /tmp/late.pt $V > $R/gpurun_out/late_trace_${V}_$TAG.log 2>&1)
        T=$(find gpurun_out/prof_late_${V}_$TAG -name "*kernel_trace.csv" | head -1)
        python tools/late_timeline.py $T 8 > gpurun_out/late_timeline_${V}_$TAG.md 2>> gpurun_out/late_timeline_$TAG.err; head -40 gpurun_out/late_timeline_${V}_$TAG.md
        rm -rf gpurun_out/prof_late_${V}_$TAG
      done
      for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
        (cd /tmp && LATE_TRACE_STEPS=4 timeout -s KILL 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_late_$TAG -o pmc -- python $R/tools/late_phase.py trace /tmp/late.pt default > $R/gpurun_out/pmc_late_$TAG.log 2>&1)
        C=$(find gpurun_out/pmc_late_$TAG -name "*counter_collection.csv" | head -1)
        python tools/sq_summary.py $C "late-phase cloud, LDS set" > gpurun_out/sq_late_lds_$TAG.md 2>> gpurun_out/late_timeline_$TAG.err; sed -n 5,20p gpurun_out/sq_late_lds_$TAG.md
        rm -rf gpurun_out/pmc_late_$TAG
      done ;;
    optrace)         # the operator surface (what the unmodified reference trainer drives): kernel time per step against wall time per step
      (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_op_$TAG -o trace -- python $R/bench.py --operator-path --steps 32 --warmup 8 --no-cpu-baseline --no-pmc --soak-steps 0 --no-training-state > $R/gpurun_out/optrace_$TAG.log 2>&1)
      tail -1 gpurun_out/optrace_$TAG.log | cut -c1-200
      S=$(find gpurun_out/prof_op_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_operator_$TAG.csv
      python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kernel_stats_operator_$TAG.csv")))
steps = 8 + 8 + 32 + 32          # setup + warm-up + timed + forward-only (forward kernels run in all of them)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / 1e6:.1f} ms over the process")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d}  total {float(r["TotalDurationNs"]) / 1e3:10.0f} us  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
      rm -rf gpurun_out/prof_op_$TAG ;;
    dptrace)         # kernel stats of the data-parallel step's glue (single-rank RCCL group), fresh and trained state
      for ST in fresh trained; do
        if [ $ST = fresh ]; then EXTRA=""; else EXTRA="--trained 1000"; fi
        (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dp_${ST}_$TAG -o trace -- python $R/tools/dp_glue_bench.py $EXTRA moments > $R/gpurun_out/dptrace_${ST}_$TAG.log 2>&1)
        S=$(find gpurun_out/prof_dp_${ST}_$TAG -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_dp_${ST}_$TAG.csv
        grep "ms/step" gpurun_out/dptrace_${ST}_$TAG.log
        python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kernel_stats_dp_${ST}_$TAG.csv")))
for r in sorted(rows, key=lambda r: -float(r["AverageNs"])):
    if any(k in r["Name"] for k in ("dp_", "compact", "mark", "slot", "Memset", "fill", "copyBuffer", "nccl", "rccl", "project_backward")):
        print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
        rm -rf gpurun_out/prof_dp_${ST}_$TAG
      done ;;
    dpglue)
      rm -f gpurun_out/dp_glue.log
      timeout -s KILL 300 python tools/dp_glue_bench.py > gpurun_out/dp_glue_$TAG.log 2>&1
      timeout -s KILL 400 python tools/dp_glue_bench.py --trained 1000 >> gpurun_out/dp_glue_$TAG.log 2>&1; grep "ms/step" gpurun_out/dp_glue_$TAG.log ;;
    spec)
      timeout -s KILL 400 python tools/spec_noise.py 3 > gpurun_out/spec_noise_$TAG.log 2>&1; grep -v amdgpu.ids gpurun_out/spec_noise_$TAG.log | cut -c1-220 ;;
    conv8)
      LITEGS_CONV_SKIP_OPERATOR=profiles/r03_convergence_3m.json timeout -s KILL 600 python tests/convergence_3m.py --runs 1 --frames 8 --iterations 1600 --eval-every 20 \
        --out gpurun_out/conv8_$TAG.md > gpurun_out/conv8_$TAG.log 2>&1; grep -E "executor:" gpurun_out/conv8_$TAG.log; grep -A30 "Cost per iteration" gpurun_out/conv8_$TAG.md | tail -26 ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
