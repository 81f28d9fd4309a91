# SQ counter passes over tools/nt_pmc.py (separate --pmc passes, kernel trace only); summary -> gpurun_out/$1/summary.txt
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/nt_pmc.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -- python3 tools/nt_pmc.py > $OUT/b.log 2>&1
python3 tools/nt_pmc.py --summarize $OUT/a $OUT/b > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
