# rocprofv3 --pmc passes over the weight-gradient GEMM of one layer under several tile variants (DJ_CFG=cfg,splits)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for cfg in ${CFGS:-4,28 14,28 17,28 15,14}; do
  i=$((i+1))
  DJ_CFG=$cfg timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${OUT:-pmcw}_$i -- python tools/one_conv.py ${SHAPE:-32 38 38 256 256 3 1 same 1} wgrad pro > /dev/null 2>gpurun_out/${OUT:-pmcw}_$i.err
  rc=$?; echo "cfg $cfg rc $rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python tools/pmc_summary.py gpurun_out/${OUT:-pmcw}_*
