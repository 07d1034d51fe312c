#!/usr/bin/env bash
# round 5: what each class of ops costs INSIDE the pipelined step at the end of the round (FPD_WHATIF lowers the class to no-ops:
# timing only, results are wrong), base and variants interleaved on ONE box.  Output: gpurun_out/r05whatif/summary.txt
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05whatif; mkdir -p $O
run() {  # name, FPD_WHATIF value
  FPD_WHATIF="$2" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phase-times > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
run nowgrad nowgrad
run nowgrad_small nowgrad_small
run nowgrad_big nowgrad_big
run student_alone t_all
run student_alone_nowgrad t_all,nowgrad
run t_nobig t_big
run t_nosmall t_small
run nobigconv nobigconv
run nobig nobig
run nomid nomid
run nosmall nosmall
run noapply noapply
run noew noew
run base2 ""
} | tee $O/summary.txt
