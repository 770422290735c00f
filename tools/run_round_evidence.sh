#!/bin/bash
# GPU box: everything that ends up under profiles/ for a round, in one call (each step appends to gpurun_out/evidence_<tag>.log
# so that a long run never looks hung).   tools/run_round_evidence.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
LOG=$R/gpurun_out/evidence_$TAG.log
cd $R
echo "== bench UCI_13" >> $LOG
python bench.py > gpurun_out/${TAG}_bench_line.json 2>> $LOG
echo "== bench wikiv2" >> $LOG
python bench.py --shape wikiv2 > gpurun_out/${TAG}_bench_line_wikiv2.json 2>> $LOG
echo "== components" >> $LOG
python tools/bench_components.py > gpurun_out/${TAG}_components.jsonl 2>> $LOG
echo "== profile bench UCI_13" >> $LOG
bash tools/profile_bench.sh $TAG >> $LOG 2>&1
echo "== profile bench wikiv2" >> $LOG
R4D_PROFILE_SHAPE=wikiv2 bash tools/profile_bench.sh ${TAG}w >> $LOG 2>&1
echo "== profile scan" >> $LOG
bash tools/profile_scan.sh $TAG >> $LOG 2>&1
echo "== profile jaccard" >> $LOG
bash tools/profile_jaccard.sh $TAG >> $LOG 2>&1
echo "== profile training step" >> $LOG
bash tools/profile_training.sh gpurun_out/train_prof >> $LOG 2>&1
cp gpurun_out/train_prof/train_kernel_stats.csv gpurun_out/${TAG}_training_step_kernel_stats.csv
echo "== done" >> $LOG
