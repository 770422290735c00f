#!/bin/bash
# GPU box: everything that ends up under profiles/ for a round, in one call (each step appends to gpurun_out/evidence_<tag>.log
# so that a long run never looks hung).   tools/run_round_evidence.sh r02
# Order matters: the PMC passes come FIRST and their summaries are installed into profiles/ (on the box), so that the bench
# lines taken afterwards carry `traffic` measured on exactly these sources (bench.py drops it when the source stamp differs).
# Everything to commit is collected under gpurun_out/evidence_<tag>/ : copy that directory's files into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
LOG=$R/gpurun_out/evidence_$TAG.log
EV=$R/gpurun_out/evidence_$TAG
mkdir -p $EV
cd $R
echo "== profile bench UCI_13 (bf16x3 GEMMs: the headline)" >> $LOG
bash tools/profile_bench.sh $TAG >> $LOG 2>&1
echo "== profile bench UCI_13 (exact-f32 MFMA GEMMs)" >> $LOG
R4D_PROFILE_GEMM=f32 bash tools/profile_bench.sh ${TAG}f >> $LOG 2>&1
echo "== profile bench wikiv2" >> $LOG
R4D_PROFILE_SHAPE=wikiv2 bash tools/profile_bench.sh ${TAG}w >> $LOG 2>&1
echo "== profile scan" >> $LOG
bash tools/profile_scan.sh $TAG >> $LOG 2>&1
cp gpurun_out/profiles_$TAG/* gpurun_out/profiles_${TAG}f/* gpurun_out/profiles_${TAG}w/* $EV/
cp gpurun_out/profiles_$TAG/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/profiles_${TAG}w/${TAG}w_pmc_traffic.json profiles/pmc_traffic_wikiv2.json
cp gpurun_out/profiles_${TAG}f/${TAG}f_pmc_traffic.json profiles/pmc_traffic_f32.json
cp gpurun_out/prof_scan_$TAG/pmc_scan.json profiles/pmc_scan.json
cp profiles/pmc_traffic.json profiles/pmc_traffic_wikiv2.json profiles/pmc_traffic_f32.json profiles/pmc_scan.json $EV/
cp gpurun_out/prof_scan_$TAG/pmc.txt $EV/${TAG}_scan_pmc.txt
for f in gpurun_out/prof_scan_$TAG/kernel_stats_*.csv; do cp $f $EV/${TAG}_scan_$(basename $f); done
echo "== bench UCI_13" >> $LOG
python bench.py > $EV/${TAG}_bench_line.json 2>> $LOG
echo "== bench wikiv2" >> $LOG
python bench.py --shape wikiv2 > $EV/${TAG}_bench_line_wikiv2.json 2>> $LOG
echo "== bf16x3 acceptance table + end-to-end CLI timings" >> $LOG
python tools/s3_acceptance.py > $EV/${TAG}_s3_acceptance.md 2>> $LOG
python tools/annotation_e2e.py annotation retriever 2>> $LOG | grep "^{" > $EV/${TAG}_cli_end_to_end.jsonl
python tools/s3_bench.py auto 2>> $LOG | grep "^{" > $EV/${TAG}_s3_gemm_shapes.jsonl
echo "== components" >> $LOG
python tools/bench_components.py > $EV/${TAG}_components.jsonl 2>> $LOG
echo "== profile jaccard" >> $LOG
bash tools/profile_jaccard.sh $TAG >> $LOG 2>&1
cp gpurun_out/prof_jaccard_$TAG/pmc.txt $EV/${TAG}_jaccard_pmc.txt
cp gpurun_out/prof_jaccard_$TAG/kernel_stats.csv $EV/${TAG}_jaccard_kernel_stats.csv
echo "== profile training step" >> $LOG
bash tools/profile_training.sh gpurun_out/train_prof >> $LOG 2>&1
cp gpurun_out/train_prof/train_kernel_stats.csv $EV/${TAG}_training_step_kernel_stats.csv
tail -1 gpurun_out/train_prof/line.json > $EV/${TAG}_training_step.jsonl
echo "== profile decode step" >> $LOG
bash tools/profile_decode.sh gpurun_out/decode_prof >> $LOG 2>&1
cp gpurun_out/decode_prof/decode_kernel_stats.csv $EV/${TAG}_decode_step_kernel_stats.csv
echo "== done" >> $LOG
ls $EV >> $LOG
