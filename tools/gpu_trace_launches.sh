# per-launch durations of one kernel in a short one-stream bench run (kernel trace, no counters):
#   bash tools/gpu_trace_launches.sh SCENE KERNEL_SUBSTRING [LIB]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
[ -n "$3" ] && export CASTER_LIB=$PWD/$3
rm -rf gpurun_out/ktrace; mkdir -p gpurun_out/ktrace
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/ktrace -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-exclusive --scene $1 --streams 1 > gpurun_out/ktrace/log 2>&1 || { tail -5 gpurun_out/ktrace/log; exit 1; }
python3 - "$2" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob('gpurun_out/ktrace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
rows.sort()
print(len(rows), 'launches, ms each in launch order:')
print(' '.join('%.2f' % d for _, d in rows))
PY
find gpurun_out/ktrace -name "*.csv" -size +1M -delete
