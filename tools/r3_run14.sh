cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
python $R/tools/stem_bench.py 20
MVF_STEM_TPW=1 python $R/tools/stem_bench.py 20
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/sq_stem -- python $R/tools/stem_bench.py 5 > /tmp/sq_stem.log 2>&1
python $R/tools/sq_summary.py $R/gpurun_out/r3_stem_sq.json $(find /tmp/sq_stem -name "*.db") | grep -i stem
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/sq_stem2 -- python $R/tools/stem_bench.py 5 > /tmp/sq_stem2.log 2>&1
python $R/tools/sq_summary.py $R/gpurun_out/r3_stem_sq2.json $(find /tmp/sq_stem2 -name "*.db") | grep -i stem
tail -3 /tmp/sq_stem2.log
