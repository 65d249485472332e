#!/bin/bash
# kernel-trace stats of the bench step under two environments (A = default, B = "$1"): per-kernel average durations side by side.
# usage: tools/prof_ab.sh "PPV_X=1 PPV_Y=2" outdir
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${2:-prof_ab}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense --no-roofline > $O/a.log 2>&1
export $1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense --no-roofline > $O/b.log 2>&1
python3 - <<PY
import csv,glob,re
def load(d):
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)[0]
    return {r['Name']:(int(r['Calls']),float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a,b=load('$O/a'),load('$O/b')
names=sorted(set(a)|set(b), key=lambda n:-(a.get(n,(0,0))[1]+b.get(n,(0,0))[1]))
ta=sum(v[1] for v in a.values()); tb=sum(v[1] for v in b.values())
print(f"total A {ta/7e6:.3f} ms/step  B {tb/7e6:.3f} ms/step (7 steps incl. warm-up; init kernels included)")
for n in names[:45]:
    ca,da=a.get(n,(0,0)); cb,db=b.get(n,(0,0))
    print(f"{ca:6d} {da/max(ca,1)/1e3:8.1f} us {da/7e6:7.3f} | {cb:6d} {db/max(cb,1)/1e3:8.1f} us {db/7e6:7.3f}  {re.sub(r'ppv::|void ','',n)[:90]}")
PY
