"""Per-launch durations of kernels matching a substring in the LAST step of a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 440
for r in rows[-n:]:
    k = r['Kernel_Name']
    if sys.argv[2] in k:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{d:8.1f} us grid {r['Grid_Size_X']:>9} z {r['Grid_Size_Z']:>3} wg {r['Workgroup_Size_X']} {k[6:70]}")
