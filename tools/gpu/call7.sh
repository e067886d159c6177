#!/bin/bash
# per-epoch kernel breakdown of the emulated rank 0 of 8 (2x4 grid): difference of two rocprofv3 --stats runs (20 vs 10 steps)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
for S in 10 20; do
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_emu8_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --exchange 2x4 --no-interleave --no-cpu-baseline --primary-only --steps $S --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_emu8_s$S.json 2> $GRAFT_REPO_ROOT/$O/prof_emu8_s$S.log)
echo "rc=$?"
done
python - <<'PY'
import csv,glob,os,json
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02'
def load(S):
    f=glob.glob(f'{O}/prof_emu8_s{S}/**/*kernel_stats.csv',recursive=True)[-1]
    return {r['Name']:(int(r['Calls']),float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a,b=load(10),load(20)
rows=[]
for k,(c2,t2) in b.items():
    c1,t1=a.get(k,(0,0.0))
    if c2>c1: rows.append((k,(c2-c1)/10,(t2-t1)/10/1e6))
rows.sort(key=lambda r:-r[2])
tot=sum(r[2] for r in rows); n=sum(r[1] for r in rows)
print('per-epoch GPU ms',round(tot,3),'launches',n)
for k,c,t in rows[:45]: print(f"{k[:95]:95s} {c:5.1f} {t:7.3f}")
for S in (10,20):
    d=json.loads([l for l in open(f'{O}/prof_emu8_s{S}.json') if l.startswith('{')][-1]); print(S, d['ms_per_step'])
PY
