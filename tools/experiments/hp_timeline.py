"""print a window of the copy / kernel timeline of tools/hostpath_once.py under rocprofv3 --kernel-trace --memory-copy-trace --output-format csv
python tools/experiments/hp_timeline.py <dir with hp_kernel_trace.csv, hp_memory_copy_trace.csv> [rows]"""
import csv, re, sys
d = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mc = list(csv.DictReader(open(d + '/hp_memory_copy_trace.csv'))); kt = list(csv.DictReader(open(d + '/hp_kernel_trace.csv')))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'].replace('MEMORY_COPY_', '')) for r in mc]
for r in kt:
    m = re.search(r'(k_\w+)', r['Kernel_Name']); ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:24]))
ev.sort()
big = [e for e in ev if e[2] in ('HOST_TO_DEVICE', 'DEVICE_TO_HOST') and e[1] - e[0] > 300000]
t0 = big[len(big) // 2][0]
for e in [e for e in ev if e[0] >= t0 - 100000 and e[1] - e[0] > 40000][:rows]:
    print("%9.3f ms  +%7.3f ms  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
