"""Print a rocprofv3 --kernel-trace --stats directory's kernel table (kernels with more than N calls):
    python scripts/print_kernel_stats.py gpurun_out/prof_x [min_calls]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
total = 0.0
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) > min_calls:
        print("%-72s %5s %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
        total += float(r["AverageNs"]) / 1e3
print("%-72s %5s %9.1f us" % ("sum of the averages", "", total))
