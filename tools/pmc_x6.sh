# rocprofv3 --pmc passes over one conv on the split-bf16 (float32x6) kernels (tools/one_conv32.py): issue / wait / LDS counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE=${SHAPE:-32 38 38 256 256 3}
MODE=${MODE:-float32x6}
i=0
for spec in ${SPECS:-fwd_plain:13 fwd_plain:9 dgrad:13 wgrad:13}; do
  set -- ${spec/:/ }
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcx_a$i -- python tools/one_conv32.py $SHAPE $1 $2 $MODE 14 > /dev/null 2>gpurun_out/pmcx_a$i.err || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmcx_b$i -- python tools/one_conv32.py $SHAPE $1 $2 $MODE 14 > /dev/null 2>gpurun_out/pmcx_b$i.err || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcx_c$i -- python tools/one_conv32.py $SHAPE $1 $2 $MODE 14 > /dev/null 2>gpurun_out/pmcx_c$i.err || exit 1
done
python - <<'PY'
import csv, glob
for i in range(1, 5):
    out = {}
    dur = 0; name = "?"; grid = 0
    for s in "abc":
        fs = glob.glob("gpurun_out/pmcx_%s%d/*/*_counter_collection.csv" % (s, i))
        if not fs: continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "dj_igemm" in r["Kernel_Name"]]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                out[r["Counter_Name"]] = float(r["Counter_Value"]); name = r["Kernel_Name"][:90]; grid = int(r["Grid_Size"])
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wc = out.get("SQ_WAVE_CYCLES", 1)
    print(name, "waves", grid // 64, "dur %.1f us" % dur)
    print("   " + "  ".join("%s=%.3g" % (k, v) for k, v in sorted(out.items())))
    print("   per wave-cycle: " + "  ".join("%s %.1f%%" % (k.replace("SQ_", ""), 100 * out[k] / wc) for k in out if k.startswith("SQ_") and ("WAIT" in k or "ACTIVE" in k or "CONFLICT" in k)))
PY
