cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 300 python tools/r03_diag_saturating.py > $O/diag_saturating.log 2>&1
tail -40 $O/diag_saturating.log
CMD="python bench.py --steps 3 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/sqA /tmp/sqB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/sqA -- $CMD > $O/sqA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqB -- $CMD > $O/sqB.log 2>&1
python tools/pmc_sq.py /tmp/sqA /tmp/sqB --md $O/pmc_sq_all.md > /dev/null 2> $O/pmc_sq.err
python tools/pmc_sq.py /tmp/sqA /tmp/sqB --match edge_ 
tail -3 $O/pmc_sq.err $O/sqA.log
