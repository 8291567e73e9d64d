# SQ counters of the message-passing kernels on the round's last build (two PMC passes) -> gpurun_out/r03z/pmc_sq_last_build.md
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/sqA /tmp/sqB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/sqA -- $CMD > $O/sqA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqB -- $CMD > $O/sqB.log 2>&1
python tools/pmc_sq.py /tmp/sqA /tmp/sqB --md $O/pmc_sq_last_build.md > /dev/null 2> $O/pmc_sq.err
python tools/pmc_sq.py /tmp/sqA /tmp/sqB --match edge_
tail -n 2 $O/pmc_sq.err; tail -n 2 $O/sqA.log
