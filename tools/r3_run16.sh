cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
python $R/tools/c3_bench.py 20
for b in 1 2 7 14 28; do echo BPW=$b; MVF_CONV3X3_BPW=$b python $R/tools/c3_bench.py 20 2>&1 | grep -v amdgpu; done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/sq_c3 -- python $R/tools/c3_bench.py 5 > /tmp/sq_c3.log 2>&1
python $R/tools/sq_summary.py $R/gpurun_out/r3_c3_sq.json $(find /tmp/sq_c3 -name "*.db") | grep -i c64
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/sq_c32 -- python $R/tools/c3_bench.py 5 > /tmp/sq_c32.log 2>&1
python $R/tools/sq_summary.py $R/gpurun_out/r3_c3_sq2.json $(find /tmp/sq_c32 -name "*.db") | grep -i c64
