#!/usr/bin/env bash
# round 6 evidence: rocprofv3 kernel trace + per-shape table + PMC passes (tools/profile.sh r06, with the HRNet passes), then the what-if table
cd "$(dirname "$0")/../.." || exit 1
HRNET=1 timeout 1500 bash tools/profile.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -12 gpurun_out/r06_profile.log
O=gpurun_out/r06whatif_final; mkdir -p $O
run() {  # name, FPD_WHATIF value, env
  FPD_WHATIF="$2" timeout 200 env $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
run nowgrad nowgrad
run student_alone t_all
run t_nobig t_big
run t_bneck_big t_bneck_big
run t_head t_head
run t_plain_big t_plain_big
run nobigconv nobigconv
run no3x3big no3x3big
run no1x1big no1x1big
run nobig nobig
run nomid nomid
run nosmall nosmall
run noapply noapply
run noew noew
run base2 ""
} | tee $O/summary.txt
