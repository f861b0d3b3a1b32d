# durations of the launches of one batch on one stream, in launch order
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/lt1; mkdir -p gpurun_out/lt1
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/lt1 -o t --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-exclusive --streams 1 > gpurun_out/lt1/log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/lt1/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name'].replace('caster::', '').replace('void ', '').split('(')[0]
    out.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
# the last batch: from the last k_generate on
last = max(i for i, (n, _) in enumerate(out) if n.startswith('k_generate'))
with open('gpurun_out/launches.txt', 'w') as g:
    for n, ms in out[last:]:
        if ms > 0.02: g.write('%-24s %8.3f ms\n' % (n, ms))
print(open('gpurun_out/launches.txt').read())
PY
rm -rf gpurun_out/lt1
