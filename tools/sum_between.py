"""Sum of kernel durations between marker fills in a rocprofv3 kernel trace (development aid for tools/lfa_levels.py)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# markers: fill kernels with grid sizes derived from 12345+d / 54321+d are hard to spot; instead split on the 'vectorized_elementwise_kernel<4, at::native::FillFunctor<float>' calls
fills = [i for i, r in enumerate(rows) if "FillFunctor<float>" in r["Kernel_Name"]]
# the last 8 fills are the 4 (start, end) pairs
fills = fills[-8:]
for lv in range(4):
    a, b = fills[2 * lv], fills[2 * lv + 1]
    seg = rows[a + 1:b]
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 5e3
    print("level %d: %d kernels per block call, %.1f us of kernels per block call" % (lv, len(seg) // 5, tot))
