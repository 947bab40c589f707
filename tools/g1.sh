cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY_ORDER=1
mkdir -p $R/gpurun_out
for v in base a7 g1nb4a7; do
PDA_HIP_LIB=$R/pda_amd/csrc/variants/libpda_hip_$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$v -o p -- env PYTHONPATH=$R python $R/tools/time_v4.py c3 131072 1 v4 > $R/gpurun_out/pmc_$v.log 2>&1
tail -2 $R/gpurun_out/pmc_$v.log
(cd $R; python tools/pmc_summary.py gpurun_out/pmc_$v sweep4 2>&1 | grep -E "derived|kernel_trace|Error|error" | head -5)
done
