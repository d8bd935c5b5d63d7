# rocprofv3 --pmc passes over one float16-mode conv on 16-bit tensors (tools/one_conv16.py): issue / wait / memory counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE=${SHAPE:-32 38 38 256 1024 1}
i=0
for spec in "fwd_plain 0" "fwd 0" "fwd_plain 9" "dgrad 2"; do
  set -- $spec
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmch_a$i -- python tools/one_conv16.py $SHAPE $1 $2 > /dev/null 2>gpurun_out/pmch_a$i.err || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmch_b$i -- python tools/one_conv16.py $SHAPE $1 $2 > /dev/null 2>gpurun_out/pmch_b$i.err || exit 1
  timeout -k 10 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr --kernel-trace --output-format csv -d gpurun_out/pmch_c$i -- python tools/one_conv16.py $SHAPE $1 $2 > /dev/null 2>gpurun_out/pmch_c$i.err || exit 1
done
python - <<'PY'
import csv, glob
for i in range(1, 5):
    out = {}
    dur = 0
    for s in "abc":
        fs = glob.glob("gpurun_out/pmch_%s%d/*/*_counter_collection.csv" % (s, i))
        if not fs: continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "dj_igemm" in r["Kernel_Name"]]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                out[r["Counter_Name"]] = float(r["Counter_Value"]); name = r["Kernel_Name"][:70]; grid = int(r["Grid_Size"])
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wc = out.get("SQ_WAVE_CYCLES", 1)
    print(name, "waves", grid // 64, "dur %.1f us" % dur)
    print("   " + "  ".join("%s=%.3g" % (k, v) for k, v in sorted(out.items())))
    print("   per wave-cycle: " + "  ".join("%s %.1f%%" % (k.replace("SQ_", ""), 100 * out[k] / wc) for k in out if k.startswith("SQ_") and ("WAIT" in k or "ACTIVE" in k)))
PY
