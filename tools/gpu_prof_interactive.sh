# kernel trace of the one-iteration-per-call loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/iprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/iprof/trace -o t --output-format csv -- python3 tools/exp/interactive_latency.py > gpurun_out/iprof/trace.log 2>&1 || exit 1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/iprof/trace/t_kernel_trace.csv')))
# the last 60*... launches belong to the car scene; split by grid? just print stats of the second half
half = rows[len(rows)//2:]
d = collections.defaultdict(list)
for r in half:
    d[r['Kernel_Name'].split('(')[0][-28:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-30s n %6d  total %9.1f us  median %7.1f  p10 %7.1f  p90 %7.1f' % (k, len(v), sum(v), v[len(v)//2], v[len(v)//10], v[len(v)*9//10]))
t0 = min(int(r['Start_Timestamp']) for r in half); t1 = max(int(r['End_Timestamp']) for r in half)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in half)
print('span %.1f ms, kernels busy %.1f ms' % ((t1 - t0) / 1e6, busy / 1e6))
PY
