"""Summarise a rocprofv3 --pmc run of tools/one_conv.py: per igemm kernel, last dispatch."""
import csv, sys, glob, collections
for d in sys.argv[1:]:
    if not glob.glob(d + "/*/*_counter_collection.csv"):
        continue
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "dj_igemm" in r["Kernel_Name"] or "dj_wgrad" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    c = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}
    r0 = [r for r in rows if int(r["Dispatch_Id"]) == last][0]
    dur = (int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) / 1e3
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    name = r0["Kernel_Name"].split("(")[0].replace("void dj_igemm_fast_kernel", "")
    print("%-34s grid %6s dur %7.1f us clk %.2f GHz | MFMA busy %5.1f%% | wait_any %4.1f%% wait_inst %4.1f%% active %4.1f%% | VALU/wave %6.0f | vgpr %s agpr %s" % (
        name, int(r0["Grid_Size"]) // 256, dur, cyc / dur / 1e3, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
        100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"] / (int(r0["Grid_Size"]) / 64), r0["VGPR_Count"], r0["Accum_VGPR_Count"]))
