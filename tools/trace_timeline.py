"""Dev tool: print the kernel timeline of the last batch step from a rocprofv3 kernel-trace csv (start, duration, gap, grid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (-12, 40)
def short(n):
    n = n.replace('void ', '')
    return n[:n.find('(')][:60] if '(' in n else n[:60]
idx = [i for i, r in enumerate(rows) if 'k_plain_alice' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = int(rows[max(0, i0 + lo)]['Start_Timestamp'])
for r in rows[max(0, i0 + lo): i0 + hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1e6:9.3f} ms  dur {(e-s)/1e3:9.1f} us  gap {(s-prev_end)/1e3:8.1f} us  grid {r['Grid_Size_X']:>8}  {short(r['Kernel_Name'])}")
    prev_end = e
