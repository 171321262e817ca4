cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|VALUBusy\|SALUBusy\|VALUUtilization\|MemUnitBusy\|LDSBankConflict\|MemUnitStalled\|WriteUnitStalled\|OccupancyPercent\|MeanOccupancyPerCU" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_list.txt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "VALUBusy VALUUtilization" "MemUnitStalled WriteUnitStalled LDSBankConflict"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmcX
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcX -o x -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --opt overlap_cameras=false > $R/gpurun_out/pmcX.log 2>&1
  f=$(find /tmp/pmcX -name "*counter_collection.csv" | head -1)
  echo "== $set" >> $R/gpurun_out/pmc_sq.txt
  if [ -n "$f" ]; then python $R/profiles/pmc_summary.py $f | grep "rasterize\|preprocess\|loss_\|adam" >> $R/gpurun_out/pmc_sq.txt; else tail -3 $R/gpurun_out/pmcX.log >> $R/gpurun_out/pmc_sq.txt; fi
done
