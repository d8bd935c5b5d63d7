cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for spec in "fwd x 11,1" "fwd x 9,1" "fwd x 10,1" "dgrad x 11,1"; do
  set -- $spec
  i=$((i+1))
  DJ_CFG=$3 timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4_$i -- python tools/one_conv.py 32 19 19 256 256 3 1 same 1 $1 $2 > /dev/null 2>gpurun_out/pmc4_$i.err || exit 1
done
echo done
