#!/usr/bin/env bash
# per-(kernel, grid) table of the HRNet step (configs[3] shapes) from a rocprofv3 kernel trace
cd "$(dirname "$0")/../.." || exit 1
ROOT=$PWD; O=$ROOT/gpurun_out/r02hr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $O/trace -o h -- python $ROOT/bench.py --config hrnet --steps 6 --warmup 2 > $O/trace.log 2>&1
python - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True): rows+=list(csv.DictReader(open(f)))
STEPS=9
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:70]
    grid=int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z'])
    d[(n,grid)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in d.values())/STEPS/1e3
print('serialised kernel ms/step %.2f, launches/step %.0f'%(tot, sum(len(v) for v in d.values())/STEPS))
with open('$O/r02_hrnet_per_shape.csv','w') as f:
    f.write('kernel,grid_threads,calls_per_step,avg_us,ms_per_step\n')
    for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
        f.write('"%s",%d,%.1f,%.2f,%.3f\n'%(k[0],k[1],len(v)/STEPS,sum(v)/len(v),sum(v)/STEPS/1e3))
fam=collections.defaultdict(lambda:[0,0.0])
for k,v in d.items():
    fam[k[0].split('<')[0]][0]+=len(v)/STEPS; fam[k[0].split('<')[0]][1]+=sum(v)/STEPS/1e3
for k,v in sorted(fam.items(), key=lambda kv:-kv[1][1])[:14]: print('%-30s calls %7.1f ms %7.3f'%(k,v[0],v[1]))
PY
head -30 $O/r02_hrnet_per_shape.csv
rm -rf $O/trace
