REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prep_c2 -o prep -- python $REPO/tools/gpu_prepare_profile.py 200 3 1 20 incremental=0 > $OUT/prep_c2.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/prep_c2/prep_results.db | cut -c1-150 | head -20
python tools/rocpd_summary.py list $OUT/prep_c2/prep_results.db gpmpc | tail -12
rm -rf $OUT/prep_c2
