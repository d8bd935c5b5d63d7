"""(fetch.csv, write.csv of a tools/pmc_calibrate.py run) -> 'F32_FETCH F32_WRITE F16_FETCH F16_WRITE' scale factors."""
import csv, sys
def per_launch(path, counter, kernel):
    v = [float(r["Counter_Value"]) * 1024.0 for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return sum(v) / len(v)
f, w = sys.argv[1], sys.argv[2]
n32, n16 = (1 << 28) * 4.0, (1 << 28) * 2.0
print("%.4f %.4f %.4f %.4f" % (n32 / per_launch(f, "FETCH_SIZE", "dj_copy2d_kernel"), n32 / per_launch(w, "WRITE_SIZE", "dj_copy2d_kernel"),
                               n16 / per_launch(f, "FETCH_SIZE", "dj_copy2d_t_kernel"), n16 / per_launch(w, "WRITE_SIZE", "dj_copy2d_t_kernel")))
